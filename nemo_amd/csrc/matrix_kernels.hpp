// matrix_kernels.hpp -- kernels on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): dense grid search, backbone tail, clutter contraction; the streaming clutter kernel
// Part of the single translation unit nemo_hip.hip (included from there, in order; not a stand-alone header).
#ifndef MATRIX_KERNELS_HPP
#define MATRIX_KERNELS_HPP
#include <type_traits>

namespace {

// ---------------------------------------------------------------- grid search on the matrix cores
// get_init_pos_rendered (code/MeshPoseSolveAll.py:84-100) compares an image with the stored renders of all
// grid poses: object_score[g, p] = sum_d render[g, p, d] * feature[p, d].  With the renders of a (mesh, grid)
// pair kept as dense pixel-major maps (nemo_grid_prerender) that is, per pixel, a small GEMM
//   [jobs of the pair] x D  times  D x [G poses]
// whose operands are read once.  A workgroup owns (pair, chunk of GP_PIX pixels); wave w owns the 32 poses
// 32w..32w+31 and walks the pair's jobs 32*GP_TILES at a time with v_mfma_f32_32x32x2_f32 (exact fp32 FMA
// chains); the loss terms max(obj, clu) - max(0, clu) are summed over the chunk's pixels in registers and
// left as per-chunk partials, which grid_finish_kernel adds in chunk order (deterministic).
constexpr int GP_PIX = 64;
#ifndef NEMO_GP_TILES
#define NEMO_GP_TILES 2
#endif
constexpr int GP_TILES = NEMO_GP_TILES;
constexpr int GP_MAX_PAIRS = 2048;

__global__ void gather_pose_kernel(int n, const int32_t *gidx, const float *gridC, const float *gridTheta, float *C, float *theta) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int g = gidx[k];
  C[3 * k] = gridC[3 * g]; C[3 * k + 1] = gridC[3 * g + 1]; C[3 * k + 2] = gridC[3 * g + 2];
  theta[k] = gridTheta[g];
}

// Render jobs of one pose block: job (pair i, block pose b) -> its mesh and its index into the grid arrays.
__global__ void block_jobs_kernel(int n_pairs, int G, int g0, int gb, const int32_t *pair_mesh, const int32_t *pair_grid,
                                  int32_t *mesh_out, int32_t *gidx_out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_pairs * gb) return;
  const int i = k / gb, b = k - i * gb;
  mesh_out[k] = pair_mesh[i];
  gidx_out[k] = pair_grid[i] * G + g0 + b;
}

// chunk_live[pair][chunk] = 1 if any pose of the block [g0, g0 + gb) covers a pixel of the chunk (from the stored lists).
__global__ void chunk_live_kernel(int G, int g0, int gb, int n_chunks, size_t stride, const uint32_t *list,
                                  const int32_t *cnt, uint8_t *live) {
  const int sl = blockIdx.x / gb, g = g0 + blockIdx.x % gb;
  const size_t pg = (size_t)sl * G + g;
  const uint32_t *l = list + pg * stride;
  for (int e = threadIdx.x; e < cnt[pg]; e += blockDim.x) live[sl * n_chunks + (int)((l[e] & 0xffffu) / GP_PIX)] = 1;
}

// Jobs grouped by pre-render slot: order[] lists the job ids slot by slot, slot_start[] the ranges,
// job_slot[n] the slot of job n (-1: not pre-rendered, evaluated on the fly by grid_kernel).
__global__ __launch_bounds__(1024) void grid_group_kernel(int N, const int32_t *mesh_idx, const int32_t *grid_idx,
                                                          const int32_t *slot_tab, int n_pairs, int32_t *order,
                                                          int32_t *slot_start, int32_t *job_slot, int32_t *work) {
  __shared__ int cnt[GP_MAX_PAIRS + 1];
  __shared__ int n_work;
  for (int i = threadIdx.x; i <= n_pairs; i += blockDim.x) cnt[i] = 0;
  if (threadIdx.x == 0) n_work = 0;
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const int mi = mesh_idx ? mesh_idx[n] : 0, gi = grid_idx ? grid_idx[n] : 0;
    const int sl = gi < MAX_GRIDS ? slot_tab[mi * MAX_GRIDS + gi] : -1;
    job_slot[n] = sl;
    if (sl >= 0) atomicAdd(&cnt[sl], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < n_pairs; ++i) {
      const int c = cnt[i];
      slot_start[i] = run;
      cnt[i] = run;  // becomes the placement cursor
      run += c;
    }
    slot_start[n_pairs] = run;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const int sl = job_slot[n];
    if (sl >= 0) order[atomicAdd(&cnt[sl], 1)] = n;  // order inside a slot is irrelevant: jobs are independent
    else work[1 + atomicAdd(&n_work, 1)] = n;        // left to grid_kernel (rendered on the fly / list path)
  }
  __syncthreads();
  if (threadIdx.x == 0) work[0] = n_work;
}

// Both operands are staged through LDS.  Per pixel the workgroup needs the feature rows
// of the pass's jobs (A, JTW tiles of 32 rows) and the rendered rows of its poses (B, PTW tiles of 32 rows): each
// row is read from HBM once, coalesced, by the whole workgroup (one pixel ahead, through registers) instead of
// once per wave with 16-byte gathers, and wave w multiplies tile (jt, pt) = (w / PTW, w % PTW) out of LDS.  Rows
// are padded to 132 floats, which makes the 16-byte operand reads of a 16-lane group conflict-free.  For
// D = 256 a pixel is staged in two halves of 128 features.
// Pose blocks: `maps` holds the renders of Gb consecutive grid poses [g_off, g_off + Gb) of every pair (all G of them
// when the store fits its budget, otherwise nemo_grid_search renders block after block into the same buffer);
// `chunk_live` (optional) says per (pair, pixel chunk) whether ANY pose of the block covers a pixel of the chunk --
// if none does, every object score is 0, every loss term max(0, clu) - max(0, clu) = 0, and the chunk is skipped.
constexpr int GQ_ROW = 132;
template <int D>
__global__ __launch_bounds__(1024) void grid_mfma_lds_kernel(int hw, int N, int G, int Gb, int g_off, int n_chunks, int PTW,
                                                             int JTW, const int32_t *img_idx, const int32_t *order,
                                                             const int32_t *slot_start, const float *__restrict__ feat,
                                                             const float *__restrict__ clu,
                                                             const float *__restrict__ maps, float *__restrict__ partial,
                                                             const uint8_t *__restrict__ chunk_live, int clutter_on) {
  extern __shared__ __align__(16) float gq_sm[];
  constexpr int KH = D / 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int sl = blockIdx.x / n_chunks, c = blockIdx.x % n_chunks;
  const int js = slot_start[sl], ns = slot_start[sl + 1] - js;
  if (ns == 0) return;
  const int p0 = c * GP_PIX, np = min(GP_PIX, hw - p0);
  const int g0 = (int)blockIdx.y * PTW * 32;      // first pose of this workgroup, relative to the block
  if (chunk_live && !chunk_live[sl * n_chunks + c]) {
    const int ng = min(PTW * 32, Gb - g0);
    for (int k = tid; k < ns * ng; k += 1024)
      partial[((size_t)c * N + order[js + k / ng]) * G + g_off + g0 + k % ng] = 0.0f;
    return;
  }
  const int nrowsA = JTW * 32, NS = PTW + JTW;  // slab k of 32 rows: A for k < JTW, B after
  float *As = gq_sm, *Bs = gq_sm + nrowsA * GQ_ROW, *cls = gq_sm + NS * 32 * GQ_ROW;
  const int pt = wave % PTW, jt = wave / PTW;
  const bool active = jt < JTW;
  const int srow = tid >> 5, sc4 = tid & 31;  // staging: row within the slab, 16-byte column
  for (int tg = 0; tg < ns; tg += nrowsA) {
    const float *rp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      rp[k] = nullptr;
      if (k < JTW) {
        const int job = order[js + min(tg + k * 32 + srow, ns - 1)];
        const int img = img_idx ? img_idx[job] : job;
        rp[k] = feat + ((size_t)img * hw + p0) * D + sc4 * 4;
      } else if (k < NS) {
        const int g = min(g0 + (k - JTW) * 32 + srow, Gb - 1);
        rp[k] = maps + (((size_t)sl * Gb + g) * hw + p0) * D + sc4 * 4;
      }
    }
    const float *clp = nullptr;
    if (clutter_on && tid < nrowsA) {
      const int job = order[js + min(tg + tid, ns - 1)];
      clp = clu + (size_t)(img_idx ? img_idx[job] : job) * hw + p0;
    }
    f16v lsum;
#pragma unroll
    for (int r = 0; r < 16; ++r) lsum[r] = 0.0f;
    f4 st[8];
    float clv = 0.0f;
    auto fetch = [&](int p, int kh) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < NS) st[k] = __builtin_nontemporal_load((const f4 *)(rp[k] + (size_t)p * D + kh * 128));
      if (kh == 0 && clp) clv = clp[p];
    };
    fetch(0, 0);
    for (int p = 0; p < np; ++p) {
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        __syncthreads();  // the previous stage has been consumed
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < NS) *(f4 *)&gq_sm[(k * 32 + srow) * GQ_ROW + sc4 * 4] = st[k];
        if (kh == 0 && tid < nrowsA) cls[tid] = clv;
        __syncthreads();
        if (kh + 1 < KH) fetch(p, kh + 1);
        else if (p + 1 < np) fetch(p + 1, 0);
        if (active) {
          const float *ap = As + (jt * 32 + i) * GQ_ROW + 4 * half, *bp = Bs + (pt * 32 + i) * GQ_ROW + 4 * half;
#pragma unroll 4
          for (int q = 0; q < 16; ++q) {
            const f4 a = *(const f4 *)(ap + 8 * q), b = *(const f4 *)(bp + 8 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
          }
        }
      }
      // C/D layout: column = lane & 31 (pose), row = (r & 3) + 8 * (r >> 2) + 4 * half (job of the tile)
      if (active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f4 cv = *(const f4 *)&cls[jt * 32 + 8 * k + 4 * half];
          const float cc[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float o = acc[4 * k + r];
            lsum[4 * k + r] += clutter_on ? (fmaxf(o, cc[r]) - fmaxf(0.0f, cc[r])) : o;
          }
        }
      }
    }
    if (active) {
      const int g = g0 + pt * 32 + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jrow = tg + jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (jrow < ns && g < Gb) partial[((size_t)c * N + order[js + jrow]) * G + g_off + g] = lsum[r];
      }
    }
    __syncthreads();  // the next pass restages the LDS
  }
}

// ---------------------------------------------------------------- the same contraction without materialised renders
// A dense grid (BASELINE config 4: 9720 poses x 12 classes) would need 478 GB of rendered maps.  What a render holds
// per (pose, pixel) is three bank rows and three weights, so nemo_grid_prerender stores exactly that -- a 20-byte
// record: three 16-bit vertex ids + a covered flag, three barycentrics -- and the kernel below builds the pose
// operand of the per-pixel GEMM in LDS from the records and the (L2-resident) vertex bank:
//   row(g, p) = (b0 * bank[v0] + b1 * bank[v1]) + b2 * bank[v2]     (the arithmetic of interpolate_kernel, bit for bit)
// A workgroup owns (pair, 64 pixels, a block of GR_PR poses) and takes the pair's jobs 64 at a time.  It works in stages
// of (listed pixel, 64 features) through two LDS stage buffers with ONE barrier per stage, and its waves have two roles
// (grid_mfma_rec_kernel): matrix waves multiply stage s while staging waves build the pose rows of stage s + 1 and
// move the feature rows of stage s + 2.  Two 512-thread workgroups share a CU (76 KB of LDS and 128 VGPRs each): a
// stage lasts as long as the slowest of its waves, and the other workgroup's waves fill the pipes meanwhile.
// pack_rec_kernel turns the rasteriser's (pix_to_face, bary) of a block of poses into the records.
struct RecVid { uint32_t v01, v2c; };  // v0 | v1 << 16, v2 | covered << 16
__global__ void pack_rec_kernel(size_t npix, size_t hw, int G, int g0, int gb, const MeshDesc *meshes, const int32_t *mesh_of_job,
                                const int32_t *p2f, const float *bary, const int32_t *slot_of, RecVid *vid, float *bout) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (job = pair * gb + b, pixel)
  if (k >= npix) return;
  const size_t job = k / hw, pix = k - job * hw;
  const size_t pair = job / gb, b = job - pair * gb;
  const size_t dst = (pair * G + (size_t)slot_of[pair * G + g0 + b]) * hw + pix;   // the store is in slot order (rec_group_kernel)
  const int f = p2f[k];
  RecVid r{0u, 0u};
  float b0 = 0.f, b1 = 0.f, b2 = 0.f;
  if (f >= 0) {
    const int32_t *fc = meshes[mesh_of_job[job]].faces + 3 * (size_t)f;
    r.v01 = (uint32_t)fc[0] | ((uint32_t)fc[1] << 16);
    r.v2c = (uint32_t)fc[2] | (1u << 16);
    b0 = bary[3 * k]; b1 = bary[3 * k + 1]; b2 = bary[3 * k + 2];
  }
  vid[dst] = r;
  bout[3 * dst] = b0; bout[3 * dst + 1] = b1; bout[3 * dst + 2] = b2;
}

#ifndef NEMO_GR_SMALL
#define NEMO_GR_SMALL 1
#endif
// Shape.  Default: 64 poses x 64 jobs per workgroup, 512 threads, two workgroups per CU (their barriers are independent:
// 0.152 s against 0.186 s for the dense one-class grid).  NEMO_GR_SMALL=0: 128 poses, 1024 threads, one workgroup per CU.
constexpr int GR_T = NEMO_GR_SMALL ? 2 : 4;    // pose tiles per workgroup
constexpr int GR_JT = 2;                       // job tiles per pass (64 jobs)
constexpr int GR_MP = NEMO_GR_SMALL ? 2 : 4;   // listed pixels per staged group of records
constexpr int GR_PR = GR_T * 32;               // poses per workgroup
constexpr int GR_NT = GR_T * GR_JT * 128;      // one matrix wave per (job tile, pose tile) and as many staging waves
constexpr int GR_SR = GR_NT / 32;              // rows one pass of the staging threads covers (16 lanes per row)
constexpr int GR_KS = 64;  // features per stage: a stage is (listed pixel, 64-feature slice)
constexpr int GR_ROW = GR_KS + 4;  // padded LDS row: the 16-byte operand reads of a 16-lane group are conflict-free
constexpr int GR_BUF = (GR_T + GR_JT) * 32 * GR_ROW;  // floats of one stage buffer (job rows, then pose rows)
constexpr size_t GR_LDS = ((size_t)2 * GR_BUF + 2 * 2 * GR_JT * 32 /*cls*/ + 2 * GR_MP * GR_PR * 5 /*records*/ + 2 * GR_PR /*coverage*/ +
                           64 /*pixel list*/ + 2 * GR_JT * 32 /*job ids*/) * sizeof(float);
// ---- pose order of the record store.  The contraction skips a (32-pose tile, pixel) none of whose poses covers the
// pixel, and a pixel no pose of the workgroup's block covers; in grid order a tile is the in-plane rotations x distances
// of one viewpoint and 85 % of the (tile, pixel) pairs stay live at 44 % coverage.  Poses with alike silhouettes are
// therefore stored next to each other (56 % live, tools/pose_grouping_probe.py): every pose gets a signature -- one
// bit per B x B block of the window, set when the pose covers a pixel of the block -- and tiles are grown greedily:
// the free pose of largest covered area is the seed, the 31 free poses at the smallest Hamming distance join it (ties:
// lowest pose index).  Results do not depend on the order: a pose's losses are its own column of the products, and a
// skipped product is an exact zero.  perm[slot] = pose, slot_of[pose] = slot, per pair.
constexpr int RG_W = 16;   // signature words: at most 512 blocks
__global__ void iota_mod_kernel(size_t n, int G, int32_t *a, int32_t *b) {   // grid order: slot = pose
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) a[k] = b[k] = (int32_t)(k % (size_t)G);
}
__global__ void rec_sig_kernel(int win_w, int B, int blocks_per_row, size_t stride, const uint32_t *list, const int32_t *cnt,
                               uint32_t *sig) {
  __shared__ uint32_t bits[RG_W];
  const size_t pp = blockIdx.x;   // pair * G + pose
  if (threadIdx.x < RG_W) bits[threadIdx.x] = 0u;
  __syncthreads();
  const int n = cnt[pp];
  const uint32_t *l = list + pp * stride;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    const int pix = (int)(l[k] & 0xffffu), r = pix / win_w, c = pix - r * win_w;
    const int b = (r / B) * blocks_per_row + c / B;
    atomicOr(&bits[b >> 5], 1u << (b & 31));
  }
  __syncthreads();
  if (threadIdx.x < RG_W) sig[pp * RG_W + threadIdx.x] = bits[threadIdx.x];
}

// One workgroup per pair.  Thread t owns the contiguous poses [t q, (t + 1) q), q = ceil(G / 1024) <= 64 (G <= 65536),
// so "lowest pose index first" is thread order, then order inside the thread.
__global__ __launch_bounds__(1024) void rec_group_kernel(int G, const uint32_t *sig, const int32_t *cnt, int32_t *perm, int32_t *slot_of) {
  __shared__ uint32_t seed_sig[RG_W];
  __shared__ int hist[RG_W * 32 + 1];
  __shared__ int wtot[16];
  __shared__ long long wbest[16];
  __shared__ int s_t, s_quota;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  sig += (size_t)pair * G * RG_W; cnt += (size_t)pair * G; perm += (size_t)pair * G; slot_of += (size_t)pair * G;
  const int q = (G + 1023) / 1024;
  const int gb = min(G, tid * q), nown = min(G, gb + q) - gb;
  u64 freem = nown >= 64 ? ~0ull : ((1ull << nown) - 1ull);
  auto dist = [&](int g) {
    const uint4 *sp = (const uint4 *)(sig + (size_t)g * RG_W);
    int d = 0;
#pragma unroll
    for (int w4 = 0; w4 < RG_W / 4; ++w4) {
      const uint4 v = sp[w4];
      d += __builtin_popcount(v.x ^ seed_sig[4 * w4]) + __builtin_popcount(v.y ^ seed_sig[4 * w4 + 1]) +
           __builtin_popcount(v.z ^ seed_sig[4 * w4 + 2]) + __builtin_popcount(v.w ^ seed_sig[4 * w4 + 3]);
    }
    return d;
  };
  // block-wide exclusive prefix of one int per thread, in thread order (two barriers)
  auto block_excl = [&](int v) {
    const int inc = wave_scan_incl(v, lane);
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wtot[w];
    __syncthreads();
    return off + inc - v;
  };
  for (int base = 0; base < G; base += 32) {
    const int need = min(32, G - base);
    // ---- seed: the free pose of largest covered area, lowest index on ties
    long long best = -1;
    for (int j = 0; j < nown; ++j)
      if ((freem >> j) & 1ull) best = max(best, ((long long)cnt[gb + j] << 32) | (long long)(0x7fffffff - (gb + j)));
    for (int o = 32; o >= 1; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (lane == 0) wbest[wave] = best;
    for (int k = tid; k <= RG_W * 32; k += 1024) hist[k] = 0;
    __syncthreads();
    long long bb = wbest[0];
    for (int w = 1; w < 16; ++w) bb = max(bb, wbest[w]);
    const int seed = 0x7fffffff - (int)(bb & 0x7fffffffll);
    if (tid < RG_W) seed_sig[tid] = sig[(size_t)seed * RG_W + tid];
    __syncthreads();
    // ---- histogram of the distances of the free poses, then the distance t the tile fills up at
    for (int j = 0; j < nown; ++j)
      if ((freem >> j) & 1ull) atomicAdd(&hist[dist(gb + j)], 1);
    __syncthreads();
    if (tid == 0) {
      int cum = 0, t = 0;
      for (; t < RG_W * 32; ++t) {
        if (cum + hist[t] >= need) break;
        cum += hist[t];
      }
      s_t = t; s_quota = need - cum;   // poses nearer than t all join; s_quota of those AT t do, lowest index first
    }
    __syncthreads();
    const int t = s_t, quota = s_quota;
    int c_lt = 0, c_eq = 0;
    for (int j = 0; j < nown; ++j)
      if ((freem >> j) & 1ull) {
        const int d = dist(gb + j);
        c_lt += d < t; c_eq += d == t;
      }
    const int eq_before = block_excl(c_eq);
    const int a_eq = max(0, min(c_eq, quota - eq_before));
    int slot = base + block_excl(c_lt + a_eq);
    int eq_taken = 0;
    for (int j = 0; j < nown; ++j)
      if ((freem >> j) & 1ull) {
        const int d = dist(gb + j);
        const bool take = d < t || (d == t && eq_taken < a_eq);
        if (d == t && take) ++eq_taken;
        if (take) {
          perm[slot] = gb + j; slot_of[gb + j] = slot; ++slot;
          freem &= ~(1ull << j);
        }
      }
    __syncthreads();
  }
}

// Per (pair, block of GR_PR poses, chunk of 64 pixels): which pixels does ANY pose of the block cover.  The contraction
// visits only those (the others contribute nothing to any loss of the block).
__global__ void rec_mask_kernel(int G, int n_pb, int n_chunks, size_t hw, const RecVid *vid, u64 *mask) {
  const int c = blockIdx.x % n_chunks, pb = (blockIdx.x / n_chunks) % n_pb, sl = blockIdx.x / (n_chunks * n_pb);
  const size_t p = (size_t)c * GP_PIX + threadIdx.x;   // 64 threads: one per pixel of the chunk
  bool any = false;
  if (p < hw)
    for (int g = pb * GR_PR; g < min(G, (pb + 1) * GR_PR) && !any; ++g) any = (vid[((size_t)sl * G + g) * hw + p].v2c >> 16) != 0u;
  const u64 m = __builtin_amdgcn_ballot_w64(any);
  if (threadIdx.x == 0) mask[blockIdx.x] = m;
}

template <int D>
__global__ __launch_bounds__(GR_NT, 4) void grid_mfma_rec_kernel(int hw, int N, int G, int n_chunks, const int32_t *img_idx,
                                                              const int32_t *order, const int32_t *slot_start,
                                                              const float *__restrict__ feat, const float *__restrict__ clu,
                                                              const MeshDesc *meshes, const int32_t *pair_mesh,
                                                              const RecVid *__restrict__ vid, const float *__restrict__ bary,
                                                              float *__restrict__ partial, const u64 *__restrict__ pix_mask,
                                                              int n_pb, int clutter_on) {
  extern __shared__ __align__(16) float gq_sm[];
  constexpr int NKS = D / GR_KS, T = GR_T, JT = GR_JT, PR = GR_PR, MP = GR_MP, ROW = GR_ROW;
  constexpr int SR = GR_SR, NMW = T * JT;    // rows per staging pass; matrix waves
  constexpr int NRB = PR / SR, NRA = JT * 32 / SR;  // row fragments per staging thread (rows srow + SR k of the pose / the job slab)
  constexpr int GS = MP * NKS;              // stages per group of records
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, half = lane >> 5;
  // The pose block is the fastest-varying index: the workgroups that are resident together read the SAME feature rows
  // (one pixel chunk of one pair's jobs), so the feature tensor streams from HBM about once per sweep of the
  // chunks instead of once per pose block.
  const int pb = (int)(blockIdx.x % n_pb), xc = (int)(blockIdx.x / n_pb);
  const int sl = xc / n_chunks, c = xc % n_chunks;
  const int js = slot_start[sl], ns = slot_start[sl + 1] - js;
  if (ns == 0) return;
  const int p0 = c * GP_PIX;
  const int g0 = pb * PR;
  // the pixels of this chunk that some pose of this block covers, in ascending order
  const u64 pmask = pix_mask[((size_t)sl * n_pb + pb) * n_chunks + c];
  const int npl = __builtin_popcountll(pmask);
  if (npl == 0) {
    const int ng = min(PR, G - g0);
    for (int k = tid; k < ns * ng; k += GR_NT) partial[((size_t)c * N + order[js + k / ng]) * G + g0 + k % ng] = 0.0f;
    return;
  }
  const gf32p bank = as_global(meshes[pair_mesh[sl]].bank);
  // LDS: the pose rows of a stage (GR_PR rows of 64 features) and the job rows of a SUB-STAGE (64 rows) are both double
  // buffered.  A stage's pose rows serve NJ = 2 sub-stages -- two passes of 64 jobs -- so a pair with more than 64 jobs
  // pays for the building of its pose rows once per 128 jobs; sub-stage u + 1 is written while the matrix
  // instructions of sub-stage u read the other buffers: one barrier per sub-stage.
  constexpr int NJ = 2;
  constexpr int ABUF = JT * 32 * ROW, BBUF = PR * ROW;                  // floats of one job-row / pose-row buffer
  float *Asm = gq_sm, *Bsm = gq_sm + 2 * ABUF;
  float *cls = gq_sm + 2 * GR_BUF;                                      // [2][NJ][64] clutter scores of the jobs, by list parity and job pass
  uint32_t *mv01 = (uint32_t *)(cls + 2 * NJ * JT * 32), *mv2c = mv01 + 2 * MP * PR;   // [2][MP * PR] records, by group parity
  float *mb = (float *)(mv2c + 2 * MP * PR);                            // [2][MP * PR][3]
  int *bcov = (int *)(mb + 2 * MP * PR * 3);                            // [2][PR]: does pose row r cover the pixel (list parity)
  int *plist = bcov + 2 * PR;                                           // [64] listed pixels (offsets inside the chunk)
  int *ordl = plist + 64;                                               // [NJ][64] job ids of the sweep
  if (tid < 64 && ((pmask >> tid) & 1ull)) plist[__builtin_popcountll(pmask & ((1ull << tid) - 1ull))] = tid;
  // Wave roles (two separate loops with the same barrier count, so that the register allocation is the maximum of the
  // two roles and not their sum):
  //   the first T * JT waves, the MATRIX waves: wave w multiplies job tile w / T with pose tile w % T and, every MP
  //     pixels, moves its share of the next group of records HBM -> LDS;
  //   the other T * JT waves, the STAGING waves: build the GR_PR pose rows of the next stage from the records and the
  //     bank (half of their fragments in each sub-stage of the current one), and move the 64 feature rows of the
  //     sub-stage after next HBM -> registers -> LDS.  Their loads are unconditional and in a fixed order, so the in-order
  //     memory counter lets them wait for the bank rows (L2) without waiting for the feature rows (HBM) issued after.
  const bool matrix = wave < NMW;
  const int pt = wave % T, jt = (wave / T) % JT;
  const int t2 = tid - NMW * 64, srow = t2 >> 4, sc4 = t2 & 15;              // staging waves: rows srow + SR k of a slab
  const int n_stage = npl * NKS;                                         // stage s = (listed pixel s / NKS, slice s % NKS)
  const gf4p feat4 = (gf4p)as_global(feat);
#ifndef NEMO_GR_ABL
#define NEMO_GR_ABL 0
#endif
  // The staging waves are the younger half of the workgroup and would lose every issue arbitration to the matrix waves
  // of their SIMD; a static priority for them is worth 3-4.5 % (priority 3 the same, priority for the matrix waves
  // nothing -- profiles/r03_dense_grid_records.md).
#ifndef NEMO_GR_PRIO
#define NEMO_GR_PRIO 1
#endif
  if (NEMO_GR_PRIO == 1 && !matrix) __builtin_amdgcn_s_setprio(1);
  if (NEMO_GR_PRIO == 2 && !matrix) __builtin_amdgcn_s_setprio(3);
  if (NEMO_GR_PRIO == 3 && matrix) __builtin_amdgcn_s_setprio(1);
  __syncthreads();
  for (int tg = 0; tg < ns; tg += NJ * JT * 32) {
    const int njp = ns - tg > JT * 32 ? NJ : 1;      // job passes of this sweep (the last one may hold a single pass)
    const int n_sub = n_stage * njp;                 // sub-stage u = (stage u / njp, job pass u % njp)
    if (matrix) {
      // records are staged MP listed pixels at a time: matrix thread t owns item t = (pose row t / MP, list slot t % MP)
      typedef float f3v __attribute__((ext_vector_type(3), aligned(4)));
      RecVid mvr;
      f3v mbr;
      static_assert(MP * PR <= NMW * 64, "one record item per matrix thread");
      const bool mine = tid < MP * PR;
      const int mpxs = tid % MP, mlds = mpxs * PR + min(tid / MP, PR - 1);
      const size_t mbase = ((size_t)sl * G + min(g0 + min(tid / MP, PR - 1), G - 1)) * hw + p0;  // (the item's pose, pixel 0 of the chunk)
      auto meta_fetch = [&](int grp) {
        const size_t idx = mbase + plist[min(grp * MP + mpxs, npl - 1)];
        mvr = vid[idx];
        mbr = *(const f3v *)(bary + 3 * idx);
      };
      auto meta_store = [&](int grp) {
        const int o = (grp & 1) * MP * PR + mlds;
        if (mine) {
          mv01[o] = mvr.v01; mv2c[o] = mvr.v2c;
          mb[o * 3] = mbr.x; mb[o * 3 + 1] = mbr.y; mb[o * 3 + 2] = mbr.z;
        }
      };
      f16v lsum[NJ], acc[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) lsum[j][r] = acc[j][r] = 0.0f;
      // ---- prologue: the records of the first group
      meta_fetch(0);
      meta_store(0);
      __syncthreads();   // the staging waves read the records
      __syncthreads();   // ... and have written the rows of sub-stage 0
      // one sub-stage for job pass JP (a compile-time index: the accumulators of the two passes are separate registers)
      auto substage = [&](const int u, const int s, auto JPc) {
        constexpr int JP = decltype(JPc)::value;
        const int pos = s / NKS, ks = s - pos * NKS;
        if (ks == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[JP][r] = 0.0f;
        }
        // does any of this wave's 32 poses cover the pixel?  (wave-uniform; a dead tile contributes nothing)
        // So does a job tile beyond the end of a ragged pass.
        const bool live = !(NEMO_GR_ABL & 2) && tg + JP * JT * 32 + jt * 32 < ns &&
                          __builtin_amdgcn_ballot_w64(bcov[(pos & 1) * PR + pt * 32 + i] != 0) != 0ull;
        if (JP == 0) {
          // records: fetched four stages, stored two stages before the staging waves first read the group
          if ((s + 2) % GS == 0 && s + 2 < n_stage) meta_store((s + 2) / GS);
          if ((s + 4) % GS == 0 && s + 4 < n_stage) meta_fetch((s + 4) / GS);
        }
        if (live) {
          const float *ap = Asm + (u & 1) * ABUF + (jt * 32 + i) * ROW + 4 * half;
          const float *bp = Bsm + (s & 1) * BBUF + (pt * 32 + i) * ROW + 4 * half;
          // one 16-byte operand step = 4 of the sub-stage's 32 matrix instructions; the operands of the next step are
          // read from LDS before this step's instructions are issued
          f4 a0 = *(const f4 *)ap, bb = *(const f4 *)bp;
#pragma unroll
          for (int q = 0; q < GR_KS / 8; ++q) {
            const f4 x0 = a0, y = bb;
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < GR_KS / 8) { a0 = *(const f4 *)(ap + 8 * (q + 1)); bb = *(const f4 *)(bp + 8 * (q + 1)); }
            __builtin_amdgcn_sched_barrier(0);
            acc[JP] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, y.x, acc[JP], 0, 0, 0);
            acc[JP] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y.y, acc[JP], 0, 0, 0);
            acc[JP] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.z, y.z, acc[JP], 0, 0, 0);
            acc[JP] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.w, y.w, acc[JP], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          // C/D layout: column = lane & 31 (pose), row = (r & 3) + 8 * (r >> 2) + 4 * half (job of the tile)
          if (ks == NKS - 1) {
            const float *cl_ = cls + ((pos & 1) * NJ + JP) * JT * 32 + jt * 32 + 4 * half;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f4 cv = *(const f4 *)&cl_[8 * k];
              const float cc[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float o = acc[JP][4 * k + r];
                lsum[JP][4 * k + r] += clutter_on ? (fmaxf(o, cc[r]) - fmaxf(0.0f, cc[r])) : o;
              }
            }
          }
        }
      };
      if (njp == NJ) {
        for (int s = 0; s < n_stage; ++s) {
          substage(2 * s, s, std::integral_constant<int, 0>{});
          __syncthreads();  // sub-stage u has been consumed, sub-stage u + 1 is in place
          substage(2 * s + 1, s, std::integral_constant<int, 1>{});
          __syncthreads();
        }
      } else {
        for (int s = 0; s < n_stage; ++s) {
          substage(s, s, std::integral_constant<int, 0>{});
          __syncthreads();
        }
      }
      const int g = g0 + pt * 32 + i;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jr = j * JT * 32 + jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (tg + jr < ns && g < G) partial[((size_t)c * N + ordl[jr]) * G + g] = lsum[j][r];
        }
    } else {
      // ---- per-sweep offsets (in 16-byte units: the feature tensor may exceed 4 GiB, its 16-byte index does not)
      uint32_t ro[NJ][NRA];   // feature rows of the sweep's jobs at pixel 0 of the chunk, per job pass
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int k = 0; k < NRA; ++k) {
          const int job = order[js + min(tg + j * JT * 32 + srow + SR * k, ns - 1)];
          ro[j][k] = (uint32_t)((((size_t)(img_idx ? img_idx[job] : job) * hw + p0) * D) >> 2) + (uint32_t)sc4;
        }
      // the clutter score of job t2 at the stage's pixel (threads that stage none read a valid dummy: the load below is
      // unconditional so that the memory counter's arithmetic is exact)
      const bool has_cl = clutter_on && t2 < JT * 32;
      const float *clp[NJ] = {feat, feat};
      if (has_cl) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int job = order[js + min(tg + j * JT * 32 + t2, ns - 1)];
          clp[j] = clu + (size_t)(img_idx ? img_idx[job] : job) * hw + p0;
        }
      }
      f4 st[NRA];   // the feature-row fragments of the sub-stage after next travel through registers
      float clv = 0.0f;
      auto fetch = [&](int u) {
        const int s = u / njp, jp = u - s * njp;
        const int px = plist[s / NKS], ks = s % NKS;
#pragma unroll
        for (int k = 0; k < NRA; ++k)
          st[k] = (NEMO_GR_ABL & 8) ? f4{0.f, 0.f, 0.f, 0.f}
                                    : feat4[(jp ? ro[1][k] : ro[0][k]) + (uint32_t)(px * (D / 4) + ks * (GR_KS / 4))];
        clv = (NEMO_GR_ABL & 8) ? 0.f : (jp ? clp[1] : clp[0])[has_cl ? px : 0];
      };
      auto put_rows = [&](int u) {   // the fetched fragments of sub-stage u go to its buffer
        const int s = u / njp, jp = u - s * njp;
        float *dst = Asm + (u & 1) * ABUF + srow * ROW + sc4 * 4;
#pragma unroll
        for (int k = 0; k < NRA; ++k) *(f4 *)&dst[SR * k * ROW] = st[k];
        if (s % NKS == 0 && has_cl) cls[(((s / NKS) & 1) * NJ + jp) * JT * 32 + t2] = clv;
      };
      // The pose-row fragments of this thread for stage s = three bank rows each, weighted by the barycentrics; they are
      // built HB = NRB / 2 at a time (fragments K0 .. K0 + HB - 1), which is what is live in registers.
      static_assert(NRB % 2 == 0, "the fragments of a stage are built in two halves");
      constexpr int HB = NRB / 2;
      f4 m[3][HB];
      float bw[3][HB];
      auto build_issue = [&](int s, auto K0c) {
        constexpr int K0 = decltype(K0c)::value;
        const int pos = s / NKS, ks = s - pos * NKS;
        // one variable LDS index per stage; fragment k's row is a compile-time distance from it
        const int base = ((pos / MP) & 1) * MP * PR + (pos % MP) * PR + srow;
        const uint32_t *a_ = mv01 + base, *b_ = mv2c + base;
        const float *w_ = mb + 3 * base;
        int *c_ = bcov + (pos & 1) * PR + srow;
        const uint32_t col = (uint32_t)(ks * GR_KS + sc4 * 4);
        uint32_t ra[HB], rb[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) {   // every LDS read first ...
          ra[k] = a_[SR * (K0 + k)]; rb[k] = b_[SR * (K0 + k)];
#pragma unroll
          for (int v = 0; v < 3; ++v) bw[v][k] = w_[3 * SR * (K0 + k) + v];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < HB; ++k) {   // ... then the bank-row loads, none of them conditional: a row that does not
          const bool cov = (rb[k] >> 16) != 0u;   // cover the pixel reads bank row 0 with zero weights (records hold 0, 0, 0)
          const uint32_t vv[3] = {ra[k] & 0xffffu, ra[k] >> 16, rb[k] & 0xffffu};
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            m[v][k] = (NEMO_GR_ABL & 1) ? f4{0.f, 0.f, 0.f, 0.f} : *(gf4p)(bank + (vv[v] * (uint32_t)D + col));
          }
          if (ks == 0 && sc4 == 0) c_[SR * (K0 + k)] = cov ? 1 : 0;
        }
      };
      auto build_put = [&](int s, auto K0c) {
        constexpr int K0 = decltype(K0c)::value;
        float *dst = Bsm + (s & 1) * BBUF + srow * ROW + sc4 * 4;
#pragma unroll
        for (int k = 0; k < HB; ++k)   // (b0 * r0 + b1 * r1) + b2 * r2: interpolate_kernel's arithmetic
          *(f4 *)&dst[SR * (K0 + k) * ROW] = (bw[0][k] * m[0][k] + bw[1][k] * m[1][k]) + bw[2][k] * m[2][k];
      };
      using I0 = std::integral_constant<int, 0>;
      using IH = std::integral_constant<int, HB>;
      // one sub-stage of the staging side: half K0 of the next stage's pose rows (both halves when the sweep has a single
      // job pass), the job rows of sub-stage u + 1 into LDS, those of sub-stage u + 2 into registers
      auto stage_sub = [&](int u, int s, auto K0c, bool both) {
        if (s + 1 < n_stage && !(NEMO_GR_ABL & 4)) {   // ONE conditional around the whole sequence: every wait count is exact
          build_issue(s + 1, K0c);                       // bank-row loads (L2)
          __builtin_amdgcn_sched_barrier(0);
          put_rows(min(u + 1, n_sub - 1));               // the NRA + 1 loads issued one sub-stage ago
          __builtin_amdgcn_sched_barrier(0);
          fetch(min(u + 2, n_sub - 1));                  // NRA + 1 loads (HBM), younger than the bank rows build_put waits for
          __builtin_amdgcn_sched_barrier(0);
          build_put(s + 1, K0c);
          if (both) {
            __builtin_amdgcn_sched_barrier(0);
            build_issue(s + 1, IH{});
            __builtin_amdgcn_sched_barrier(0);
            build_put(s + 1, IH{});
          }
        } else {
          put_rows(min(u + 1, n_sub - 1));               // (the last stage: nothing left to build)
          __builtin_amdgcn_sched_barrier(0);
          fetch(min(u + 2, n_sub - 1));
        }
      };
      // ---- prologue: the job ids of the sweep, the feature rows of sub-stages 0 and 1, the pose rows of stage 0
      if (t2 < NJ * JT * 32) ordl[t2] = order[js + min(tg + t2, ns - 1)];
      fetch(0);
      put_rows(0);
      fetch(min(1, n_sub - 1));
      __syncthreads();   // the records of the first group are in LDS
      build_issue(0, I0{});
      build_put(0, I0{});
      build_issue(0, IH{});
      build_put(0, IH{});
      __syncthreads();
      if (njp == NJ) {
        for (int s = 0; s < n_stage; ++s) {
          stage_sub(2 * s, s, I0{}, false);
          __syncthreads();
          stage_sub(2 * s + 1, s, IH{}, false);
          __syncthreads();
        }
      } else {
        for (int s = 0; s < n_stage; ++s) {
          stage_sub(s, s, I0{}, true);
          __syncthreads();
        }
      }
    }
    __syncthreads();  // the next sweep restages the LDS
  }
}

// loss[n][g] = K_n - (1/hw) * sum over chunks (chunk order), K_n = 1 + mean(min(clu, 0)) (loss_constant).
__global__ __launch_bounds__(256) void grid_finish_kernel(int hw, float inv_hw, int N, int G, int n_chunks,
                                                          const int32_t *img_idx, const int32_t *job_slot,
                                                          const float *clu, const float *partial,
                                                          const int32_t *pre_cnt, float *losses, u64 *counter,
                                                          int clutter_on, const int32_t *perm) {
  __shared__ double red[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int sl = job_slot[n];
  if (sl < 0) return;
  double a = 0.0;
  if (clutter_on) {
    const float *c = clu + (size_t)(img_idx ? img_idx[n] : n) * hw;
    for (int k = tid; k < hw; k += 256) a += (double)fminf(c[k], 0.0f);
  }
  red[tid] = a;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float K = 1.0f + (float)red[0] * inv_hw;
  u64 cov = 0;
  for (int g = tid; g < G; g += 256) {
    float s = 0.0f;
    for (int c = 0; c < n_chunks; ++c) s += partial[((size_t)c * N + n) * G + g];
    // the record store keeps its poses in slot order (perm[slot] = pose); losses are reported in grid order
    losses[(size_t)n * G + (perm ? perm[(size_t)sl * G + g] : g)] = K - inv_hw * s;
    cov += (u64)pre_cnt[(size_t)sl * G + g];
  }
  if (counter && cov) atomicAdd(counter, cov);
}

// ---------------------------------------------------------------- backbone tail
// NetE2E.forward_test after the backbone (code/models/KeypointRepresentationNet.py:217-223): the 1x1 convolution
// with out_layer.weight [D][C] (no bias), F.normalize over the D channels (eps 1e-12), and -- fused -- the clutter
// score of code/MeshPoseSolveAll.py:220.  Input x is the backbone's [N][C][h*w] map; the output is written
// directly in the solver's pixel-major layout [N][h*w][D].
// out[d, p] = sum_c W[d, c] x[c, p] runs on v_mfma_f32_32x32x2_f32: A = 32 output channels (D/32 tiles kept in
// accumulators), B = 32 consecutive pixels of one image (coalesced dword loads per channel), the whole weight
// matrix staged once per workgroup in LDS in the order the lanes consume it (one 16-byte read per k-step and four
// output tiles).  Workgroups are persistent.  The LDS-resident weights leave room for one workgroup -- two waves
// per SIMD -- per CU, so little but the wave's own instruction stream hides latency: the channels go through in
// chunks of 2 * FH_KS, branch-free and fully unrolled; the B operands of the NEXT chunk (the next tile's first
// chunk after a tile's last) are loaded into the other half of a register ping-pong one by one between the groups
// of matrix instructions of this chunk (a whole chunk = FH_KS x 4 x 64 cycles ahead of their use), and the A
// operands are read from LDS one k-step ahead.  Measured (tools/prof_head.py, 512 images, C = 256 -> D = 128):
// 2.45 ms = 110 TFLOP/s, 0.70 of the fp32 matrix peak (the branchy per-k-step version it replaces: 66).
constexpr int FH_KS = 32;  // k-steps (channel pairs) per chunk
constexpr int FH_NT = 512, FH_NW = FH_NT / 64;  // two waves per SIMD, 256 registers each: accumulators 64 (D = 128), B ping-pong 64
template <int D>
__global__ __launch_bounds__(FH_NT) void feature_head_kernel(const float *__restrict__ x, int N, int C, int hw,
                                                             const float *__restrict__ weight,
                                                             const float *__restrict__ cbank, int K,
                                                             float *__restrict__ feat, float *__restrict__ clu) {
  constexpr int T = D / 32;                        // output-channel tiles
  extern __shared__ __align__(16) float wl[];      // [ceil(C/2 / FH_KS) * FH_KS][2][32][T]: (k-step, half, row in tile, tile); zero past C
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int nch = (C / 2 + FH_KS - 1) / FH_KS;     // chunks per tile
  {
    const int cpad = nch * FH_KS * 2;  // channels rounded up to whole chunks (zero weights past C)
    for (int k = tid; k < D * cpad; k += FH_NT) {  // coalesced along the channels of one output row
      const int d = k / cpad, c = k - d * cpad;
      wl[(((c >> 1) * 2 + (c & 1)) * 32 + (d & 31)) * T + (d >> 5)] = c < C ? weight[(size_t)d * C + c] : 0.0f;
    }
  }
  __syncthreads();
  const int tiles_per_img = (hw + 31) / 32;
  const long long total = (long long)N * tiles_per_img;
  const size_t hw2 = 2 * (size_t)hw;
  // address of this lane's B value of k-step 0 of chunk `ch` of tile `tile`: channel 2 (ch FH_KS) + half of pixel p0 + i
  auto b_ptr = [&](long long tile, int ch) {
    const int n = (int)(tile / tiles_per_img), p0 = (int)(tile - (long long)n * tiles_per_img) * 32;
    return x + ((size_t)n * C + half + 2 * ch * FH_KS) * hw + min(p0 + i, hw - 1);
  };
  const long long stride = (long long)gridDim.x * FH_NW;
  long long tile = (long long)blockIdx.x * FH_NW + wave;
  if (tile >= total) return;
  int ch = 0;
  f16v acc[T];
  float b0[FH_KS], b1[FH_KS];
  {
    const float *xq = b_ptr(tile, 0);
    const int nv = min(FH_KS, C / 2);
#pragma unroll
    for (int s2 = 0; s2 < FH_KS; ++s2) {
      b0[s2] = __builtin_nontemporal_load(xq);
      xq += s2 + 1 < nv ? hw2 : 0;
    }
  }
  // one chunk: FH_KS k-steps on `bc` with the next chunk's B values fetched into `bn` between them, and the tile's
  // epilogue after its last chunk
  auto chunk = [&](float (&bc)[FH_KS], float (&bn)[FH_KS]) {
    const bool last_ch = ch + 1 == nch;
    const long long ntile = last_ch ? tile + stride : tile;
    const int nchk = last_ch ? 0 : ch + 1;
    // source of the prefetch; after the wave's last chunk: a harmless re-read of the chunk in hand.  Past C / 2 the
    // last channel pair of the SAME pixel is re-read and meets zero weights (no select on the loaded value: it
    // would make the wave wait for the load here instead of a chunk later).
    const bool more = ntile < total;
    const float *xq = b_ptr(more ? ntile : tile, more ? nchk : ch);
    const int nv = min(FH_KS, C / 2 - (more ? nchk : ch) * FH_KS);
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    }
    const float *wp = wl + (size_t)(((ch * FH_KS) * 2 + half) * 32 + i) * T;
    // A operands run one k-step ahead of the matrix instructions that use them (few waves per SIMD: an LDS read
    // issued behind a group of matrix instructions would only start when the last of them issues)
    f4 ae[T / 4], ao[T / 4];  // even / odd k-steps: two live register sets, so a read can be issued a step early
    auto lds_a = [&](int s2, f4(&a)[T / 4]) {
#pragma unroll
      for (int t4 = 0; t4 < T / 4; ++t4) a[t4] = *(const f4 *)(wp + s2 * 64 * T + 4 * t4);
    };
    auto mfma_step = [&](const f4(&a)[T / 4], float bv) {
#pragma unroll
      for (int t4 = 0; t4 < T / 4; ++t4) {
        acc[4 * t4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t4].x, bv, acc[4 * t4], 0, 0, 0);
        acc[4 * t4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t4].y, bv, acc[4 * t4 + 1], 0, 0, 0);
        acc[4 * t4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t4].z, bv, acc[4 * t4 + 2], 0, 0, 0);
        acc[4 * t4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t4].w, bv, acc[4 * t4 + 3], 0, 0, 0);
      }
    };
    lds_a(0, ae);
#pragma unroll
    for (int s2 = 0; s2 < FH_KS; s2 += 2) {
      // full scheduling fences: left alone, the scheduler sinks each read to just above the instructions that use it
      lds_a(s2 + 1, ao);
      bn[s2] = __builtin_nontemporal_load(xq);
      xq += s2 + 1 < nv ? hw2 : 0;
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(ae, bc[s2]);
      __builtin_amdgcn_sched_barrier(0);
      if (s2 + 2 < FH_KS) lds_a(s2 + 2, ae);
      bn[s2 + 1] = __builtin_nontemporal_load(xq);
      xq += s2 + 2 < nv ? hw2 : 0;
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(ao, bc[s2 + 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (last_ch) {
      const int n = (int)(tile / tiles_per_img), p0 = (int)(tile - (long long)n * tiles_per_img) * 32;
      const int p = min(p0 + i, hw - 1);
      const bool pok = p0 + i < hw;
      // lane (pixel j = i, half) holds channels d = 32 t + 8 q + 4 half + c, q, c = 0..3  (acc[t][4 q + c])
      float n2 = 0.0f;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) n2 = fmaf(acc[t][r], acc[t][r], n2);
      n2 += __shfl_xor(n2, 32);
      const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);  // F.normalize(p=2, dim=1, eps=1e-12)
      float *op = feat + ((size_t)n * hw + p) * D + 4 * half;
      float best = -INFINITY;
      for (int k = 0; k < (clu ? K : 0); ++k) {
        const float *ck = cbank + (size_t)k * D + 4 * half;
        float dsum = 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f4 cv = *(const f4 *)(ck + 32 * t + 8 * q);
            dsum = fmaf(acc[t][4 * q] * inv, cv.x, dsum); dsum = fmaf(acc[t][4 * q + 1] * inv, cv.y, dsum);
            dsum = fmaf(acc[t][4 * q + 2] * inv, cv.z, dsum); dsum = fmaf(acc[t][4 * q + 3] * inv, cv.w, dsum);
          }
        dsum += __shfl_xor(dsum, 32);
        best = fmaxf(best, dsum);
      }
      if (pok) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(f4 *)(op + 32 * t + 8 * q) = f4{acc[t][4 * q] * inv, acc[t][4 * q + 1] * inv, acc[t][4 * q + 2] * inv, acc[t][4 * q + 3] * inv};
        if (clu && half == 0) clu[(size_t)n * hw + p] = best;
      }
    }
    tile = ntile;
    ch = nchk;
  };
  while (tile < total) {
    chunk(b0, b1);
    if (tile >= total) break;
    chunk(b1, b0);
  }
}

// Clutter score: out[p] = max_k <cbank[k], feat[p]> (code/MeshPoseSolveAll.py:220,
// lib/MeshUtils.py:273-275) on the fp32 MFMA (v_mfma_f32_32x32x2_f32, exact f32
// fmaf chain).  A = 32 clutter vectors, B = 32 pixels, so a lane owns one pixel
// column and the max over clutter vectors is a max over its accumulators.
// Lane (i = l & 31, half = l >> 5) feeds feature index 8q + 4*half + c as k-step
// 4q + c: both operands are then plain 16-byte reads.
// Persistent workgroups.  The bank goes through LDS in chunks of CK_ROWS vectors (rows padded by four floats: the
// 16-byte reads of a wave are conflict-free), double-buffered when there is more than one chunk, so the A operand
// never shares the in-order vector-memory counter with the feature rows; every wave keeps the 32 feature rows of
// its tile in registers and fetches the rows of its NEXT tile into a second register set under the matrix
// instructions of the last chunk (without it the waves of a SIMD load together and multiply together: time =
// stream time + matrix time, 65 TFLOP/s at K = 64; now 104, and 141 = 0.90 of the fp32 matrix peak at K = 2560,
// tools/time_clutter.py).  Rows past K repeat row K - 1, which leaves the max alone.
constexpr int CK_ROWS = 64;
template <int D> struct CkCfg { static constexpr int NT = D <= 128 ? 512 : 256; };  // registers per wave: 256 / 512
template <int D>
__global__ __launch_bounds__(CkCfg<D>::NT) void clutter_kernel(const float *__restrict__ feat, size_t npix,
                                                               const float *__restrict__ cbank, int K,
                                                               float *__restrict__ out) {
  constexpr int NT = CkCfg<D>::NT, NWV = NT / 64;
  constexpr int RS = D + 4;                       // padded row stride in floats
  constexpr int NS = CK_ROWS * (D / 4) / NT;      // 16-byte pieces of a chunk per thread
  extern __shared__ __align__(16) float cl[];     // [2][CK_ROWS][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int nchunks = (K + CK_ROWS - 1) / CK_ROWS;
  const size_t ntiles = (npix + 31) / 32;
  f4 st[NS];
  auto stage_load = [&](int c) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int idx = tid + j * NT, row = idx / (D / 4), col4 = idx - row * (D / 4);
      st[j] = ((const f4 *)cbank)[(size_t)min(c * CK_ROWS + row, K - 1) * (D / 4) + col4];
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int idx = tid + j * NT, row = idx / (D / 4), col4 = idx - row * (D / 4);
      *(f4 *)(cl + (size_t)buf * CK_ROWS * RS + row * RS + 4 * col4) = st[j];
    }
  };
  auto load_x = [&](size_t tile, f4(&x)[D / 8]) {
    const size_t pix = min(tile * 32 + (size_t)i, npix - 1);
    const f4 *xp = (const f4 *)(feat + pix * D) + half;
#pragma unroll
    for (int q = 0; q < D / 8; ++q) x[q] = __builtin_nontemporal_load(xp + 2 * q);
  };
  const size_t stride = (size_t)gridDim.x * NWV;
  size_t base = (size_t)blockIdx.x * NWV;  // the workgroup's first tile of the round; every wave takes part in every
  if (base >= ntiles) return;              // round (barriers), waves past the end recompute the last tile and write nothing
  f4 xa[D / 8], xb[D / 8];
  stage_load(0);
  load_x(min(base + wave, ntiles - 1), xa);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  auto round = [&](f4(&xc)[D / 8], f4(&xn)[D / 8]) {
    const size_t tile = base + wave;
    const size_t nbase = base + stride;
    float best = -INFINITY;
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 == nchunks) load_x(min((nbase < ntiles ? nbase : base) + wave, ntiles - 1), xn);
      if (nchunks > 1) stage_load(c + 1 == nchunks ? 0 : c + 1);
      const float *ap = cl + (size_t)buf * CK_ROWS * RS + i * RS + 4 * half;
      // rows i (and, when the chunk holds more than 32 vectors, 32 + i) of the chunk against the tile's 32 pixels;
      // the A operands are read one step ahead of the matrix instructions that use them
      auto contract = [&](auto two_tag) {
        constexpr bool TWO = decltype(two_tag)::value;
        f16v acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
        f4 e0, e1, o0, o1;
        e0 = *(const f4 *)(ap);
        if (TWO) e1 = *(const f4 *)(ap + 32 * RS);
        auto mm = [&](const f4 &a0, const f4 &a1, const f4 &xv) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, xv.x, acc0, 0, 0, 0);
          if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, xv.x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, xv.y, acc0, 0, 0, 0);
          if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, xv.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, xv.z, acc0, 0, 0, 0);
          if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, xv.z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, xv.w, acc0, 0, 0, 0);
          if (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, xv.w, acc1, 0, 0, 0);
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < D / 8; q += 2) {
          o0 = *(const f4 *)(ap + 8 * (q + 1));
          if (TWO) o1 = *(const f4 *)(ap + 32 * RS + 8 * (q + 1));
          __builtin_amdgcn_sched_barrier(0);
          mm(e0, e1, xc[q]);
          __builtin_amdgcn_sched_barrier(0);
          if (q + 2 < D / 8) {
            e0 = *(const f4 *)(ap + 8 * (q + 2));
            if (TWO) e1 = *(const f4 *)(ap + 32 * RS + 8 * (q + 2));
          }
          __builtin_amdgcn_sched_barrier(0);
          mm(o0, o1, xc[q + 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) best = fmaxf(best, TWO ? fmaxf(acc0[r], acc1[r]) : acc0[r]);
      };
      if (K - c * CK_ROWS > 32) contract(std::true_type{}); else contract(std::false_type{});
      if (nchunks > 1) {
        stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
      }
    }
    best = fmaxf(best, __shfl_xor(best, 32));
    if (half == 0 && tile < ntiles && tile * 32 + i < npix) out[tile * 32 + i] = best;
    base = nbase;
  };
  while (base < ntiles) {
    round(xa, xb);
    if (base >= ntiles) break;
    round(xb, xa);
  }
}

// Geometry of a padded table slot (see bank_table_kernel): row count (row length: table_ld, device_common.hpp).
__host__ __device__ inline uint32_t table_rows(uint32_t vmax) { return (vmax + (uint32_t)CK_ROWS - 1u) / (uint32_t)CK_ROWS * (uint32_t)CK_ROWS; }

// Per-job table of ALL feature-row x bank-row dot products: tab[n][v][p] = <bank_m(n)[v], feat_img(n)[p]>, f32
// [N][tab_stride / hw rows][hw].  The three dot products the solve needs per covered pixel and Adam step do not depend
// on the pose, so nemo_solve computes them once here -- a [V x D] . [D x hw] product per job on the fp32 matrix cores,
// clutter_kernel's structure with the accumulators stored instead of maximised -- and its 300 steps read three floats
// per covered pixel (eval_pose<TAB>).  Vertex-major rows: a lane owns a pixel column, so a stored accumulator row is the
// 32 pixels of one 4-row x 8-column patch of the window (128 bytes; column order: table_ld, device_common.hpp), and the
// pixels of a face and of its neighbours share the table's cache lines.
// Grid: (workgroups per job, jobs of the launch); slot y of the table belongs to the job workgroup y of the solve runs.
// A slot is PADDED -- rows of ld = 32 floats per pixel patch, a multiple of CK_ROWS rows (table_ld / table_rows) -- so
// that every store of the kernel is unconditional: a predicated store costs a branch around it and, behind a branch, the
// compiler can no longer count outstanding memory operations and waits for all of them.
template <int D>
__global__ __launch_bounds__(CkCfg<D>::NT) void bank_table_kernel(const float *__restrict__ feat, int hw, int ld, int w,
                                                                  const int32_t *img_idx, const int32_t *mesh_idx,
                                                                  const MeshDesc *meshes, float *__restrict__ tab,
                                                                  size_t tab_stride, const int32_t *order, int N) {
  constexpr int NT = CkCfg<D>::NT, NWV = NT / 64;
  constexpr int RS = D + 4;                       // padded row stride in floats
  constexpr int NS = CK_ROWS * (D / 4) / NT;      // 16-byte pieces of a chunk per thread
  extern __shared__ __align__(16) float cl[];     // [2][CK_ROWS][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  // table slot blockIdx.y holds the job the solve's workgroup blockIdx.y will run: order[blockIdx.y], or blockIdx.y itself
  const int n = order ? order[blockIdx.y] : (int)blockIdx.y;
  if ((unsigned)n >= (unsigned)N) return;   // (nemo_solve_static's shared order array overwritten by an overlapping call)
  const MeshDesc M = meshes[mesh_idx ? mesh_idx[n] : 0];
  const int K = M.V;
  const float *cbank = M.bank;
  const float *fimg = feat + (size_t)(img_idx ? img_idx[n] : n) * hw * D;
  __attribute__((address_space(1))) float *gout = (__attribute__((address_space(1))) float *)(tab + (size_t)blockIdx.y * tab_stride);
  const int nchunks = (K + CK_ROWS - 1) / CK_ROWS;
  const int ntiles = ld / 32;   // pixel patches (4 rows x 8 columns of the window), row-major
  const int tpr = (int)table_tpr((uint32_t)w), h = hw / w;
  f4 st[NS];
  auto stage_load = [&](int c) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int idx = tid + j * NT, row = idx / (D / 4), col4 = idx - row * (D / 4);
      st[j] = ((const f4 *)cbank)[(size_t)min(c * CK_ROWS + row, K - 1) * (D / 4) + col4];
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int idx = tid + j * NT, row = idx / (D / 4), col4 = idx - row * (D / 4);
      *(f4 *)(cl + (size_t)buf * CK_ROWS * RS + row * RS + 4 * col4) = st[j];
    }
  };
  auto load_x = [&](int tile, f4(&x)[D / 8]) {
    // lane i of a half = position i of the patch: row i >> 3, column i & 7 (columns past the window repeat its edge:
    // their table entries are padding)
    const int ti = tile / tpr, tj = tile - ti * tpr;
    const int pix = min(4 * ti + (i >> 3), h - 1) * w + min(8 * tj + (i & 7), w - 1);
    const f4 *xp = (const f4 *)(fimg + (size_t)pix * D) + half;
#pragma unroll
    for (int q = 0; q < D / 8; ++q) x[q] = __builtin_nontemporal_load(xp + 2 * q);
  };
  const int stride = (int)gridDim.x * NWV;
  int base = (int)blockIdx.x * NWV;  // the workgroup's first tile of the round; every wave takes part in every round
  if (base >= ntiles) return;         // (barriers), waves past the end recompute the last tile and write nothing
  // Two register sets of feature rows at D = 128: the next tile's 32 rows are fetched under the matrix instructions of
  // the round's last chunk.  (D = 256: 256 threads and 128 registers per set -- one set, the reload is exposed once per round.)
  constexpr bool TWOX = D <= 128;
  f4 xa[D / 8], xb[TWOX ? D / 8 : 1];
  stage_load(0);
  load_x(min(base + wave, ntiles - 1), xa);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  const uint32_t uld = (uint32_t)ld;
  typedef __attribute__((address_space(1))) char *gcharp;
  auto round = [&](f4(&xc)[D / 8], f4(&xn)[TWOX ? D / 8 : 1]) {
    // a wave past the end recomputes the last tile and stores the same values a second time
    const int tile = min(base + wave, ntiles - 1);
    const int nbase = base + stride;
    // C/D layout: column = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * half (bank row of the 32-row tile).
    // A row's address = wave-uniform row base + one 32-bit per-lane byte offset (a slot is far below 4 GiB).
    uint32_t boff = 4u * ((uint32_t)(tile * 32 + i) + (uint32_t)(4 * half) * uld);
    asm volatile("" : "+v"(boff));   // opaque: stays ONE 32-bit register next to the scalar row bases
    for (int c = 0; c < nchunks; ++c) {
      if constexpr (TWOX) {
        if (c + 1 == nchunks && nbase < ntiles) load_x(min(nbase + wave, ntiles - 1), xn);
      }
      if (nchunks > 1) stage_load(c + 1 == nchunks ? 0 : c + 1);
      const float *ap = cl + (size_t)buf * CK_ROWS * RS + i * RS + 4 * half;
      // rows i and 32 + i of the chunk against the tile's 32 pixels; the A operands are read one step ahead of the
      // matrix instructions that use them
      f16v acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
      f4 e0, e1, o0, o1;
      e0 = *(const f4 *)(ap);
      e1 = *(const f4 *)(ap + 32 * RS);
      auto mm = [&](const f4 &a0, const f4 &a1, const f4 &xv) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, xv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, xv.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, xv.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, xv.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, xv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, xv.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, xv.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, xv.w, acc1, 0, 0, 0);
      };
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < D / 8; q += 2) {
        o0 = *(const f4 *)(ap + 8 * (q + 1));
        o1 = *(const f4 *)(ap + 32 * RS + 8 * (q + 1));
        __builtin_amdgcn_sched_barrier(0);
        mm(e0, e1, xc[q]);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 2 < D / 8) {
          e0 = *(const f4 *)(ap + 8 * (q + 2));
          e1 = *(const f4 *)(ap + 32 * RS + 8 * (q + 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        mm(o0, o1, xc[q + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nchunks > 1) stage_store(buf ^ 1);   // before the stores: the wait for the staged rows must not wait for them
      __builtin_amdgcn_sched_barrier(0);
      {
        int crow = c * CK_ROWS;
        asm volatile("" : "+s"(crow));   // opaque and scalar: keeps the row bases out of the loop's induction expressions
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int r = j & 15, rr = (r & 3) + 8 * (r >> 2) + 32 * (j >> 4);
          const float val = (j >> 4) ? acc1[r] : acc0[r];
#ifndef NEMO_TAB_ABL
#define NEMO_TAB_ABL 0   // diagnostic builds: 1 = no table stores, 2 = plain (cacheable) stores
#endif
          const gcharp rowp = (gcharp)(gout + (size_t)(crow + rr) * uld);   // wave-uniform
          __attribute__((address_space(1))) float *dst = (__attribute__((address_space(1))) float *)(rowp + boff);
          if (NEMO_TAB_ABL == 0) __builtin_nontemporal_store(val, dst);
          else if (NEMO_TAB_ABL == 2) *dst = val;
          else if (val == 12345.678f) *dst = val;
        }
      }
      if (nchunks > 1) {
        __syncthreads();
        buf ^= 1;
      }
    }
    if constexpr (!TWOX) {
      if (nbase < ntiles) load_x(min(nbase + wave, ntiles - 1), xc);
    }
    base = nbase;
  };
  while (base < ntiles) {
    if constexpr (TWOX) {
      round(xa, xb);
      if (base >= ntiles) break;
      round(xb, xa);
    } else {
      round(xa, xb);
    }
  }
}

// Clutter score for a small bank (K <= 8; the reference's mean clutter vector is K = 1, code/MeshPoseSolveAll.py:207):
// a matrix-core tile would be all padding, and the pass is a pure stream of the feature maps.  D/8 lanes per pixel
// read the pixel's row as two coalesced 16-byte loads per lane (all rows of a wave's 8 iterations in flight at once),
// the clutter vectors come from L1, and a DPP tree reduces each dot product.
template <int D>
__global__ __launch_bounds__(256) void clutter_dot_kernel(const float *__restrict__ feat, size_t npix,
                                                          const float *__restrict__ cbank, int K,
                                                          float *__restrict__ out) {
  constexpr int LPP = D / 8, GPW = 64 / LPP, IT = 8;
  const int lane = threadIdx.x & 63, gl = lane & (LPP - 1), grp = lane / LPP;
  const size_t p0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (GPW * IT) + grp;
  f4 x0[IT], x1[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const size_t p = p0 + (size_t)it * GPW;
    const f4 *xp = (const f4 *)(feat + (p < npix ? p : npix - 1) * D) + gl;
    x0[it] = __builtin_nontemporal_load(xp);
    x1[it] = __builtin_nontemporal_load(xp + LPP);
  }
  float best[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) best[it] = -INFINITY;
  for (int k = 0; k < K; ++k) {
    const f4 *cp = (const f4 *)(cbank + (size_t)k * D) + gl;
    const f4 c0 = cp[0], c1 = cp[LPP];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const f4 a = __builtin_elementwise_fma(x1[it], c1, x0[it] * c0);
      best[it] = fmaxf(best[it], sum_group<LPP>((a.x + a.y) + (a.z + a.w)));
    }
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const size_t p = p0 + (size_t)it * GPW;
    if (gl == 0 && p < npix) out[p] = best[it];
  }
}

}  // namespace

#endif  // MATRIX_KERNELS_HPP
