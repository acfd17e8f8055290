#!/bin/bash
# Builds libnemo_hip.so for gfx950 (MI355X).  -ffp-contract=off is part of the
# numerical contract: coverage must not depend on FMA contraction.
# -fno-slp-vectorize: the compiler's own pairing of scalar fp32 operations into packed (v_pk_*) instructions costs the
# solve kernel 5 % (a packed fp32 instruction takes two passes of the vector ALU, and the pairs are assembled with
# moves): 8 % more vector instructions, 4 % fewer busy cycles without it (profiles/r03_solve_kernel_variants.md (m)).
# Explicit vector types (the f4 dot products of the streaming kernels, the matrix-core operands) are not affected.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$ROOT/nemo_amd/lib"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value \
  -I"$ROOT/include" -o "$ROOT/nemo_amd/lib/libnemo_hip.so" "$HERE/nemo_hip.hip" "$@"
echo "built $ROOT/nemo_amd/lib/libnemo_hip.so"
