#!/bin/bash
# usage: tools/pmc_sq.sh <lib or ""> <tag>   -- SQ issue/wait counter passes over tools/prof_solve.py (solve kernel)
set -u
lib=$1; tag=$2
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export NEMO_HIP_LIB=$lib
p1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
p2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"
p3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"
p4="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
i=0
for ctrs in "$p1" "$p2" "$p3" "$p4"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/sq_${tag}_$i -- python3 tools/prof_solve.py --images 1024 --steps 100 --reps 1 > gpurun_out/sq_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/sq_${tag}_$i.log; exit 1; }
done
python3 tools/pmc_read.py gpurun_out/sq_${tag}_1 gpurun_out/sq_${tag}_2 gpurun_out/sq_${tag}_3 gpurun_out/sq_${tag}_4 | grep -A9 -E "solve_kernel|bank_table"
