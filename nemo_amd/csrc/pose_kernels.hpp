// pose_kernels.hpp -- kernels built on eval_pose: rasterise, score, pre-render, grid (list path), launch order, the persistent solve, module backward; interpolate
// Part of the single translation unit nemo_hip.hip (included from there, in order; not a stand-alone header).
#ifndef POSE_KERNELS_HPP
#define POSE_KERNELS_HPP

namespace {

// ---------------------------------------------------------------- kernels
extern __shared__ __align__(16) unsigned char smem[];

__global__ void campos_kernel(const float *C, const float *theta, const float *extra, int N, float *R, float *T) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  Cam k;
  float c[3] = {C[3 * n], C[3 * n + 1], C[3 * n + 2]};
  float ex[3] = {0.f, 0.f, 0.f};
  if (extra) { ex[0] = extra[3 * n]; ex[1] = extra[3 * n + 1]; ex[2] = extra[3 * n + 2]; }
  dev_camera(c, theta[n], extra ? ex : nullptr, k);
  for (int i = 0; i < 9; ++i) R[9 * n + i] = k.R[i];
  for (int i = 0; i < 3; ++i) T[3 * n + i] = k.T[i];
}

template <bool PERSP, bool BIG>
__global__ __launch_bounds__(NT, 4) void raster_kernel(KParams P, const MeshDesc *meshes, const int32_t *mesh_idx,
                                                       const float *Rs, const float *Ts, int32_t *p2f, float *bary,
                                                       float *zbuf) {
  const int n = blockIdx.x;
  const MeshDesc M = meshes[mesh_idx ? mesh_idx[n] : 0];
  const Lds L = lds_carve(smem, P);
  const size_t hw = (size_t)P.h * P.w;
  EmitOut emit{p2f + n * hw, bary + 3 * n * hw, zbuf ? zbuf + n * hw : nullptr};
  for (int i = threadIdx.x; i < (int)hw; i += NT) {
    emit.p2f[i] = -1;
    emit.bary[3 * i] = -1.0f; emit.bary[3 * i + 1] = -1.0f; emit.bary[3 * i + 2] = -1.0f;
    if (emit.zbuf) emit.zbuf[i] = -1.0f;
  }
  if (threadIdx.x < 9) L.ust[U_R + threadIdx.x] = Rs[9 * n + threadIdx.x];
  if (threadIdx.x < 3) L.ust[U_T + threadIdx.x] = Ts[3 * n + threadIdx.x];
  load_world_verts<BIG>(P, L, M);  // ends with a barrier
  eval_pose<128, PERSP, false, false, false, false, BIG>(P, L, M, nullptr, nullptr, emit);
}

// One evaluation per job at given poses (teacher-forced parity entry point).
template <int D, bool PERSP, bool BIG>
__global__ __launch_bounds__(NT, 4) void score_kernel(KParams P, const MeshDesc *meshes, const int32_t *img_idx,
                                                      const int32_t *mesh_idx, const float *feat, const float *clu,
                                                      const float *Cs, const float *thetas, float *loss, float *grad,
                                                      int32_t *ncov_out, u64 *counter) {
  const int n = blockIdx.x;
  const MeshDesc M = meshes[mesh_idx ? mesh_idx[n] : 0];
  const int img = img_idx ? img_idx[n] : n;
  const size_t hw = (size_t)P.h * P.w;
  const float *f = feat + (size_t)img * hw * D;
  const float *c = clu ? clu + (size_t)img * hw : nullptr;
  const Lds L = lds_carve(smem, P);
  load_world_verts<BIG>(P, L, M);
  loss_constant(P, L, c);
  if (threadIdx.x == 0) {
    const float Cp[3] = {Cs[3 * n], Cs[3 * n + 1], Cs[3 * n + 2]};
    Cam cam;
    ust_set_camera(L, Cp, thetas[n], cam);
  }
  __syncthreads();
  EmitOut none{nullptr, nullptr, nullptr};
  eval_pose<D, PERSP, true, true, false, false, BIG>(P, L, M, f, c, none);
  if (threadIdx.x == 0) {
    loss[n] = L.ust[U_K] - P.inv_hw * block_total(L, 0);
    const int nc = (int)(block_total(L, T_NCOV) + 0.5f);
    if (ncov_out) ncov_out[n] = nc;
    if (grad) {
      const float Cp[3] = {Cs[3 * n], Cs[3 * n + 1], Cs[3 * n + 2]};
      Cam cam;
      ust_get_camera(L, cam);
      float g[4];
      ust_totals_to_grad(L, cam, Cp, g);
      grad[4 * n] = g[0]; grad[4 * n + 1] = g[1]; grad[4 * n + 2] = g[2]; grad[4 * n + 3] = g[3];
    }
    if (counter) atomicAdd(counter, (u64)nc);
  }
}

// Pre-render (get_pre_render_samples, code/MeshPoseSolveAll.py:62-82): the coverage of a grid pose
// depends on the mesh and the pose only, so its face-major list is built once per (mesh, grid, pose)
// and every image's grid search consumes it.  One workgroup per (pair, pose).
template <bool PERSP, bool BIG>
__global__ __launch_bounds__(NT, 4) void prerender_kernel(KParams P, const MeshDesc *meshes, const int32_t *pair_mesh,
                                                          const int32_t *pair_grid, const float *gridC,
                                                          const float *gridTheta, int G, uint32_t *list, int32_t *cnt,
                                                          size_t stride) {
  const int sl = blockIdx.x / G, g = blockIdx.x % G;
  const MeshDesc M = meshes[pair_mesh[sl]];
  const int gi = pair_grid[sl];
  const Lds L = lds_carve(smem, P);
  load_world_verts<BIG>(P, L, M);
  if (threadIdx.x == 0) {
    const float *gc = gridC + ((size_t)gi * G + g) * 3;
    const float Cp[3] = {gc[0], gc[1], gc[2]};
    Cam cam;
    ust_set_camera(L, Cp, gridTheta[(size_t)gi * G + g], cam);
  }
  __syncthreads();
  EmitOut e{};
  e.list_out = list + (size_t)blockIdx.x * stride;
  eval_pose<128, PERSP, false, false, false, false, BIG>(P, L, M, nullptr, nullptr, e);
  if (threadIdx.x == 0) cnt[blockIdx.x] = (int)(block_total(L, T_NCOV) + 0.5f);
}

// Grid search: one workgroup per (job, grid pose).  Block ids are dealt so that
// all poses of one image run on the same XCD at about the same time (blocks b
// and b+8 share an XCD under round-robin dispatch; L2 affinity only, never
// correctness): the image's feature rows are then served from that XCD's L2.
template <int D, bool PERSP, bool BIG>
__global__ __launch_bounds__(NT, 4) void grid_kernel(KParams P, const MeshDesc *meshes, int N, const int32_t *img_idx,
                                                     const int32_t *mesh_idx, const int32_t *grid_idx,
                                                     const float *feat, const float *clu, const float *gridC,
                                                     const float *gridTheta, int G, float *losses, u64 *counter,
                                                     PreStore pre, const int32_t *work, unsigned block_base,
                                                     unsigned stride) {
  // work == nullptr: every job, one (job, pose block) item per workgroup (stride 0; large meshes: launches are
  // sliced by block_base and the scratch slice is blockIdx.x).  Otherwise work[0] jobs listed in work[1..] -- those
  // the matrix-core path did not take -- walked by a fixed number of workgroups with a stride; when that list is
  // empty the launch costs a few microseconds.
  const int GB = (G + GRID_POSES_PER_WG - 1) / GRID_POSES_PER_WG;
  const int nw = work ? work[0] : N;
  const unsigned total = 8u * (unsigned)((nw + 7) / 8) * (unsigned)GB;
  const Lds L = lds_carve(smem, P);
  for (unsigned bid = blockIdx.x + block_base; bid < total; bid += stride) {
  const int xcd = bid & 7, k = bid >> 3;
  const int wn = (k / GB) * 8 + xcd, g0 = (k % GB) * GRID_POSES_PER_WG;
  if (wn < nw) {
  const int n = work ? work[1 + wn] : wn;
  const int mi = mesh_idx ? mesh_idx[n] : 0;
  const MeshDesc M = meshes[mi];
  const int img = img_idx ? img_idx[n] : n;
  const int gi = grid_idx ? grid_idx[n] : 0;
  const int slot = (pre.slot && gi < MAX_GRIDS) ? pre.slot[mi * MAX_GRIDS + gi] : -1;
  const size_t hw = (size_t)P.h * P.w;
  const float *f = feat + (size_t)img * hw * D;
  const float *c = clu ? clu + (size_t)img * hw : nullptr;
  load_world_verts<BIG>(P, L, M);
  loss_constant(P, L, c);
  u64 ncov_sum = 0;
  // a workgroup walks a few consecutive poses of its image: the per-image set-up is paid once
  for (int g = g0; g < min(G, g0 + GRID_POSES_PER_WG); ++g) {
    if (threadIdx.x == 0) {
      const float *gc = gridC + ((size_t)gi * G + g) * 3;
      const float Cp[3] = {gc[0], gc[1], gc[2]};
      Cam cam;
      ust_set_camera(L, Cp, gridTheta[(size_t)gi * G + g], cam);
    }
    __syncthreads();
    if (slot >= 0) {  // coverage of this (mesh, pose) was rendered once by nemo_grid_prerender
      EmitOut e{};
      const size_t pg = (size_t)slot * G + g;
      e.pre_list = pre.list + pg * pre.stride;
      e.pre_cnt = pre.cnt[pg];
      eval_pose<D, PERSP, false, true, false, true, BIG>(P, L, M, f, c, e);
    } else {
      EmitOut none{};
      eval_pose<D, PERSP, false, true, false, false, BIG>(P, L, M, f, c, none);
    }
    if (threadIdx.x == 0) {
      losses[(size_t)n * G + g] = L.ust[U_K] - P.inv_hw * block_total(L, 0);
      ncov_sum += (u64)(block_total(L, T_NCOV) + 0.5f);
    }
  }
  if (threadIdx.x == 0 && counter) atomicAdd(counter, ncov_sum);
  }
  if (stride == 0) break;
  }
}


// argmin over the grid losses (first minimum, torch.min semantics at
// code/MeshPoseSolveAll.py:97) -> C0, theta0.
__global__ void grid_argmin_kernel(int N, int G, const float *losses, const int32_t *grid_idx, const float *gridC,
                                   const float *gridTheta, float *C0, float *theta0, int32_t *best) {
  const int n = blockIdx.x, lane = threadIdx.x;  // 64 threads
  float bl = INFINITY;
  int bi = 0x7fffffff;
  for (int g = lane; g < G; g += 64) {
    const float l = losses[(size_t)n * G + g];
    if (l < bl || (l == bl && g < bi)) { bl = l; bi = g; }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    const float ol = __shfl_xor(bl, o);
    const int oi = __shfl_xor(bi, o);
    if (ol < bl || (ol == bl && oi < bi)) { bl = ol; bi = oi; }
  }
  if (lane == 0) {
    if (bi == 0x7fffffff) bi = 0;  // all-NaN row: keep the first pose
    const int gi = grid_idx ? grid_idx[n] : 0;
    const float *gc = gridC + ((size_t)gi * G + bi) * 3;
    C0[3 * n] = gc[0]; C0[3 * n + 1] = gc[1]; C0[3 * n + 2] = gc[2];
    theta0[n] = gridTheta[(size_t)gi * G + bi];
    if (best) best[n] = bi;
  }
}

// Launch order of the solve (scheduling only; results do not depend on it).  A solve is one workgroup per job
// for all 300 steps, a CU holds one workgroup, and a job's cost grows with the pixels its object covers
// (1.2k .. 6k at the bench workload), so with only a few jobs per CU the launch ends when the unluckiest CU
// does.  Jobs are therefore started longest first: solve_cost_kernel counts the covered pixels at the
// initial pose (one rasterisation per job, 1/steps of the solve's work) and solve_order_kernel sorts the job
// ids by that count, descending (counting sort, 64-pixel buckets); workgroup b of the solve runs job order[b].
template <bool PERSP, bool BIG>
__global__ __launch_bounds__(NT, 4) void solve_cost_kernel(KParams P, const MeshDesc *meshes, const int32_t *mesh_idx,
                                                           const float *C0, const float *theta0, const float *trans0,
                                                           int32_t *cost) {
  const int n = blockIdx.x;
  const MeshDesc M = meshes[mesh_idx ? mesh_idx[n] : 0];
  const Lds L = lds_carve(smem, P);
  load_world_verts<BIG>(P, L, M);
  if (threadIdx.x == 0) {
    const float Cp[3] = {C0[3 * n], C0[3 * n + 1], C0[3 * n + 2]};
    const float ex[3] = {trans0 ? trans0[2 * n] : 0.f, trans0 ? trans0[2 * n + 1] : 0.f, 0.f};
    Cam cam;
    ust_set_camera(L, Cp, theta0[n], cam, trans0 ? ex : nullptr);
  }
  __syncthreads();
  EmitOut e{};
  eval_pose<128, PERSP, false, false, false, false, BIG>(P, L, M, nullptr, nullptr, e);
  if (threadIdx.x == 0) cost[n] = (int)(block_total(L, T_NCOV) + 0.5f);
}

constexpr int SO_BUCKETS = 1024;
__global__ __launch_bounds__(1024) void solve_order_kernel(int N, const int32_t *cost, int32_t *order) {
  __shared__ int cnt[SO_BUCKETS];
  auto bucket = [](int c) { return SO_BUCKETS - 1 - min(max(c, 0) >> 6, SO_BUCKETS - 1); };  // bucket 0 = most pixels
  for (int i = threadIdx.x; i < SO_BUCKETS; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) atomicAdd(&cnt[bucket(cost[n])], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < SO_BUCKETS; ++i) {
      const int c = cnt[i];
      cnt[i] = run;  // becomes the placement cursor
      run += c;
    }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) order[atomicAdd(&cnt[bucket(cost[n])], 1)] = n;
}

// XCD affinity of the launch order.  Workgroup b is dispatched to XCD b % 8, and each XCD has its own 4 MiB L2; with
// a dozen meshes in a batch their vertex banks (0.56 MB each) do not all stay resident in one L2.  Inside every
// window of 64 consecutive ranks of the cost-sorted order (8 slots per XCD; costs within a window are alike, so the
// longest-first property is kept) the jobs are dealt so that a mesh lands on its preferred XCD
// (mesh * 8 / n_meshes) while that XCD has slots left in the window, on the nearest XCD with a free slot otherwise.
// One thread per window.  Scheduling only.
__global__ void solve_affinity_kernel(int N, int n_meshes, const int32_t *mesh_idx, int32_t *order) {
  const int base = (int)(blockIdx.x * blockDim.x + threadIdx.x) * 64;
  if (base >= N) return;
  const int len = min(64, N - base);
  int32_t in[64], out[64];
  int used[8], cap[8];
  for (int x = 0; x < 8; ++x) { used[x] = 0; cap[x] = (len - x + 7) / 8; }  // slots i < len with i % 8 == x
  for (int i = 0; i < len; ++i) in[i] = order[base + i];
  unsigned long long left = 0;
  for (int i = 0; i < len; ++i) {
    const int x = max(0, min(7, (int)(((long long)mesh_idx[in[i]] * 8) / max(n_meshes, 1))));
    if (used[x] < cap[x]) out[x + 8 * used[x]++] = in[i]; else left |= 1ull << i;
  }
  for (int i = 0; i < len; ++i) {
    if (!((left >> i) & 1ull)) continue;
    const int x = max(0, min(7, (int)(((long long)mesh_idx[in[i]] * 8) / max(n_meshes, 1))));
    for (int d = 1; d < 8; ++d) {
      const int y = (x + ((d & 1) ? (d + 1) / 2 : 8 - d / 2)) & 7;  // x+1, x-1, x+2, x-2, ...
      if (used[y] < cap[y]) { out[y + 8 * used[y]++] = in[i]; break; }
    }
  }
  for (int i = 0; i < len; ++i) order[base + i] = out[i];
}

// The fused solve: code/MeshPoseSolveAll.py:227-250 in one launch.  Thread 0
// runs the serial part of every step (camera, camera backward, Adam) and hands
// R, T to the workgroup through LDS; (C, theta) and the Adam moments live in LDS.
// TAB: the per-job tables of feature-row x bank-row dot products exist (bank_table_kernel, launched just before);
// the table of this workgroup's job is slot blockIdx.x: tab + blockIdx.x * tab_stride floats.
#ifndef NEMO_RESORT_EVERY
#define NEMO_RESORT_EVERY 8   // steps between two face sorts of the ordered rasteriser pass (a power of two)
#endif
template <int D, bool PERSP, bool BIG, bool TAB = false, bool LAZY = false>
__global__ __launch_bounds__(NT, 4) void solve_kernel(KParams P, const MeshDesc *meshes, const int32_t *img_idx,
                                                      const int32_t *mesh_idx, const float *feat, const float *clu,
                                                      const float *C0, const float *theta0, const float *trans0,
                                                      int ndof, const float *adam_tab, int steps, float beta1,
                                                      float beta2, float eps, float *pose, float *loss_out,
                                                      float *trace, u64 *counter, int stagger,
                                                      const int32_t *order, int N, int32_t *status,
                                                      const float *tab = nullptr, size_t tab_stride = 0) {
  const int n = order ? order[blockIdx.x] : (int)blockIdx.x;  // longest job first (solve_order_kernel)
  // `order` is a permutation of [0, N) written by this call's own set-up kernels; a workgroup that reads anything
  // else (nemo_solve_static's shared array overwritten by an overlapping call) leaves instead of indexing with it
  if ((unsigned)n >= (unsigned)N) return;
  const MeshDesc M = meshes[mesh_idx ? mesh_idx[n] : 0];
  const int img = img_idx ? img_idx[n] : n;
  const size_t hw = (size_t)P.h * P.w;
  const float *f = feat + (size_t)img * hw * D;
  const float *c = clu ? clu + (size_t)img * hw : nullptr;
  const Lds L = lds_carve(smem, P);
  // One-off start stagger.  Every workgroup does the same amount of work per step, so
  // without it all resident workgroups run their HBM-heavy score phase at the same time
  // and their HBM-idle raster phase at the same time; spreading the start over about one
  // step keeps the chip-wide HBM demand flat for the whole launch.
  {
    const unsigned slot = (blockIdx.x * 0x9E3779B1u) >> 25;  // 0..127, well mixed
    for (unsigned i = 0; i < slot * stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
#ifdef NEMO_SOLVE_PRIO
  {
    const int w_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((NEMO_SOLVE_PRIO == 1 || NEMO_SOLVE_PRIO == 5) && w_ >= 8) __builtin_amdgcn_s_setprio(1);
    if (NEMO_SOLVE_PRIO == 2 && w_ >= 8) __builtin_amdgcn_s_setprio(3);
    if ((NEMO_SOLVE_PRIO == 3 || NEMO_SOLVE_PRIO == 5) && w_ == 0) __builtin_amdgcn_s_setprio(3);
    if (NEMO_SOLVE_PRIO == 4 && w_ >= 12) __builtin_amdgcn_s_setprio(1);
  }
#endif
#ifdef NEMO_STAMPS
  stamp_zero();
#endif
  load_world_verts<BIG>(P, L, M);
  if constexpr (TAB) {  // the ordered rasteriser pass (eval_pose) starts in face order
    for (int i = threadIdx.x; i < M.F; i += NT) ((unsigned short *)L.fc)[4 * i + 3] = (unsigned short)i;
    if constexpr (LAZY) {         // the job's table starts empty: no block exists (bitmap behind the 64 sort bins, eval_pose)
      const int n_tiles = (int)(table_ld((uint32_t)P.h, (uint32_t)P.w) >> 5);   // one 8-byte word of vertex-chunk bits per pixel patch
      for (int i = threadIdx.x; i < n_tiles; i += NT) ((u64 *)L.finfo)[32 + i] = 0ull;
    }
  }
  loss_constant(P, L, c);
  const bool lead = threadIdx.x == 0;
  if (lead) {
    L.ust[U_P] = C0[3 * n]; L.ust[U_P + 1] = C0[3 * n + 1]; L.ust[U_P + 2] = C0[3 * n + 2]; L.ust[U_P + 3] = theta0[n];
#pragma unroll
    for (int i = 0; i < 4; ++i) { L.ust[U_M + i] = 0.f; L.ust[U_V + i] = 0.f; }
    // 6-DoF (PredUnlabeledDataset.py:318-328): trans_2d enters as extra_trans = (t0, t1, 0) added to T
    L.ust[U_P6] = (ndof == 6 && trans0) ? trans0[2 * n] : 0.f;
    L.ust[U_P6 + 1] = (ndof == 6 && trans0) ? trans0[2 * n + 1] : 0.f;
    L.ust[U_M6] = L.ust[U_M6 + 1] = L.ust[U_V6] = L.ust[U_V6 + 1] = 0.f;
    *(unsigned int *)&L.ust[U_BLK] = 0u;
    const float ex[3] = {L.ust[U_P6], L.ust[U_P6 + 1], 0.f};
    Cam cam;
    ust_set_camera(L, L.ust + U_P, L.ust[U_P + 3], cam, ndof == 6 ? ex : nullptr);
  }
  __syncthreads();
  const float omb1 = (float)(1.0 - (double)beta1), omb2 = (float)(1.0 - (double)beta2);
  EmitOut none{nullptr, nullptr, nullptr};
  float last_loss = 0.f;
  u64 cov_total = 0;
  // The z-buffer of the next step is cleared while thread 0 runs the serial camera / Adam section.
  for (int i = threadIdx.x; i < P.cap; i += NT) L.zb[i] = ~0ull;
  __syncthreads();
  // This thread's first two vertices are fetched (L2) while thread 0 runs the serial section, not at the top of the step.
  float vreg[6];
  auto fetch_verts = [&]() {
    const gf32p gv = as_global(M.verts);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = min((int)threadIdx.x + k * NT, M.V - 1);
      vreg[3 * k] = gv[3 * v]; vreg[3 * k + 1] = gv[3 * v + 1]; vreg[3 * k + 2] = gv[3 * v + 2];
    }
  };
  fetch_verts();
  for (int t = 0; t < steps; ++t) {
    // no closing barrier inside: thread 0 leaves eval_pose with the totals and goes straight on to Adam, the other
    // waves clear the z-buffer meanwhile (the score loop's scratch in that region is dead behind the band barrier)
    eval_pose<D, PERSP, true, true, false, false, BIG, false, TAB, LAZY>(P, L, M, f, c, none, true, vreg,
                                                                     TAB ? tab + (size_t)blockIdx.x * tab_stride : nullptr,
                                                                     TAB && (t & (NEMO_RESORT_EVERY - 1)) == 0);
    fetch_verts();
    for (int i = threadIdx.x; i < P.cap; i += NT) L.zb[i] = ~0ull;
#ifdef NEMO_STAMPS
    const unsigned long long st_lead = (threadIdx.x == 0) ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
    if (lead) {
      float p[6] = {L.ust[U_P], L.ust[U_P + 1], L.ust[U_P + 2], L.ust[U_P + 3], L.ust[U_P6], L.ust[U_P6 + 1]};
      Cam cam;
      ust_get_camera(L, cam);
      float g[6];
      ust_totals_to_grad(L, cam, p, g);
      g[4] = block_total(L, T_GT);      // dL/d extra_trans = dL/dT
      g[5] = block_total(L, T_GT + 1);
      last_loss = L.ust[U_K] - P.inv_hw * block_total(L, 0);
      cov_total += (u64)(block_total(L, T_NCOV) + 0.5f);
      if (trace) {
        if (ndof == 6) {
          float *tr = trace + ((size_t)n * steps + t) * 13;
#pragma unroll
          for (int i = 0; i < 6; ++i) { tr[i] = p[i]; tr[7 + i] = g[i]; }
          tr[6] = last_loss;
        } else {
          float *tr = trace + ((size_t)n * steps + t) * 9;
          tr[0] = p[0]; tr[1] = p[1]; tr[2] = p[2]; tr[3] = p[3]; tr[4] = last_loss;
          tr[5] = g[0]; tr[6] = g[1]; tr[7] = g[2]; tr[8] = g[3];
        }
      }
      // torch.optim.Adam (1.5.1 form) with the host-prepared per-step scalars
      const float step_size = adam_tab[2 * t], bc2s = adam_tab[2 * t + 1];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (i >= ndof) break;
        const int um = i < 4 ? U_M + i : U_M6 + i - 4, uv = i < 4 ? U_V + i : U_V6 + i - 4, up = i < 4 ? U_P + i : U_P6 + i - 4;
        float m = L.ust[um], v = L.ust[uv];
        m = m * beta1 + omb1 * g[i];
        v = v * beta2 + (omb2 * g[i]) * g[i];
        const float denom = sqrtf(v) / bc2s + eps;
        p[i] = p[i] - step_size * (m / denom);
        L.ust[um] = m; L.ust[uv] = v; L.ust[up] = p[i];
      }
      const float ex[3] = {p[4], p[5], 0.f};
      ust_set_camera(L, p, p[3], cam, ndof == 6 ? ex : nullptr);
    }
    __syncthreads();
#ifdef NEMO_STAMPS
    if (threadIdx.x == 0) {
      const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();
      stamp_lds()[7] += now_ - st_lead;
    }
#endif
  }
#ifdef NEMO_STAMPS
  __syncthreads();
  stamp_flush();
#endif
  if (lead) {
    float *po = pose + (size_t)ndof * n;
    po[0] = L.ust[U_P]; po[1] = L.ust[U_P + 1]; po[2] = L.ust[U_P + 2]; po[3] = L.ust[U_P + 3];
    if (ndof == 6) { po[4] = L.ust[U_P6]; po[5] = L.ust[U_P6 + 1]; }
    if (loss_out) loss_out[n] = last_loss;
    if (status) {  // 1 = finite pose, 0 = not (scored pi/2 by the evaluation, code/MeshPoseSolveAll.py:262-263)
      bool ok = true;
      for (int i = 0; i < ndof; ++i) ok = ok && __builtin_isfinite(po[i]);
      status[n] = ok ? 1 : 0;
    }
    if (counter) atomicAdd(counter, cov_total);
#ifndef NEMO_LAZY_DISTINCT
    if (LAZY && counter) atomicAdd(counter + 1, (u64) * (unsigned int *)&L.ust[U_BLK]);
#endif
  }
#ifdef NEMO_LAZY_DISTINCT   // diagnostic build: report the blocks that EXIST at the end instead of the blocks computed (the difference = computed twice)
  if constexpr (LAZY) {
    const int n_tiles = (int)(table_ld((uint32_t)P.h, (uint32_t)P.w) >> 5);
    unsigned int c_ = 0;
    for (int i = threadIdx.x; i < n_tiles; i += NT) c_ += (unsigned int)__builtin_popcountll(((u64 *)L.finfo)[32 + i]);
    if (counter && c_) atomicAdd(counter + 1, (u64)c_);
  }
#endif
}

// Backward of rasterize + interpolate for the autograd-compatible module path:
// coverage is recomputed from (R, T) (deterministic, identical to the forward),
// grad_out plays the role of the feature map, and the block totals dL/dR, dL/dT
// are written out for torch autograd to chain into campos / theta.
template <int D, bool PERSP, bool BIG>
__global__ __launch_bounds__(NT, 4) void interp_bwd_kernel(KParams P, const MeshDesc *meshes, const int32_t *mesh_idx,
                                                           const float *Rs, const float *Ts, const float *grad_out,
                                                           float *grad_R, float *grad_T) {
  const int n = blockIdx.x;
  const MeshDesc M = meshes[mesh_idx ? mesh_idx[n] : 0];
  const Lds L = lds_carve(smem, P);
  const size_t hw = (size_t)P.h * P.w;
  if (threadIdx.x < 9) L.ust[U_R + threadIdx.x] = Rs[9 * n + threadIdx.x];
  if (threadIdx.x < 3) L.ust[U_T + threadIdx.x] = Ts[3 * n + threadIdx.x];
  load_world_verts<BIG>(P, L, M);
  EmitOut none{nullptr, nullptr, nullptr};
  eval_pose<D, PERSP, true, true, true, false, BIG>(P, L, M, grad_out + (size_t)n * hw * D, nullptr, none);
  if (threadIdx.x < 9) grad_R[9 * n + threadIdx.x] = block_total(L, T_GR + threadIdx.x);
  if (threadIdx.x < 3) grad_T[3 * n + threadIdx.x] = block_total(L, T_GT + threadIdx.x);
}

// interpolate_face_attributes (lib/MeshUtils.py:129): 32 lanes per pixel, 16 B each.
__global__ void interpolate_kernel(int D, size_t npix_total, size_t hw, const MeshDesc *meshes,
                                   const int32_t *mesh_idx, const int32_t *p2f, const float *bary, float *out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = D / 4;  // f4 per pixel
  const size_t pix = gid / per;
  const int c = (int)(gid % per);
  if (pix >= npix_total) return;
  const int n = (int)(pix / hw);
  const MeshDesc M = meshes[mesh_idx ? mesh_idx[n] : 0];
  const int f = p2f[pix];
  f4 o = {0.f, 0.f, 0.f, 0.f};
  if (f >= 0) {
    const float b0 = bary[3 * pix], b1 = bary[3 * pix + 1], b2 = bary[3 * pix + 2];
    const f4 m0 = ((const f4 *)(M.bank + (size_t)M.faces[3 * f] * D))[c];
    const f4 m1 = ((const f4 *)(M.bank + (size_t)M.faces[3 * f + 1] * D))[c];
    const f4 m2 = ((const f4 *)(M.bank + (size_t)M.faces[3 * f + 2] * D))[c];
    o = (b0 * m0 + b1 * m1) + b2 * m2;
  }
  ((f4 *)out)[pix * per + c] = o;
}

}  // namespace

#endif  // POSE_KERNELS_HPP
