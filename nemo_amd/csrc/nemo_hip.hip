// nemo_hip.hip -- hand-written CDNA4 (gfx950) kernels + C ABI of libnemo_hip.so.  One translation unit: the device
// code is in device_common.hpp, eval_pose.hpp, pose_kernels.hpp and matrix_kernels.hpp (included below, in that
// order), the context, launch logic and the extern "C" entry points are here.
//
// One workgroup (1024 threads = 16 wave64, the CU's whole LDS) owns one job (image, mesh) and runs the
// whole render-and-compare evaluation out of LDS (eval_pose):
//   camera (look-at, theta)  ->  project vertices into LDS
//   -> face-parallel rasterisation into a packed 64-bit LDS z-buffer
//      (key = pz bits << 32 | face id; ds_min_u64 == "nearest fragment, lowest
//      face index on ties", the exact outcome of PyTorch3D's naive K=1 loop)
//   -> face-major list of the covered pixels (deterministic order)
//   -> fused interpolate + score: every wave owns a contiguous run of the list; 16 lanes per pixel,
//      16-byte loads of the pixel's feature row and of the face's three vertex-bank rows (kept in
//      registers while consecutive entries share them), three 128-d dot products reduced with DPP,
//      then one lane per pixel for the loss term and dL/d(bary)
//   -> hand-derived backward to dL/dR, dL/dT (12 scalars, fixed reduction tree, no atomics)
//      -> dL/dC, dL/dtheta -> Adam, all inside the kernel.
// The solve kernel loops this 300 times without returning to the host.
//
// Build with -ffp-contract=off: coverage is decided by separately rounded fp32
// products (as the reference's CPU rasterizer does); the dot products use
// explicit fmaf.  Reference citations (file:line) are into Angtian/NeMo.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <vector>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nemo_hip.h"
#include "fastdiv.h"

#include "device_common.hpp"
#include "eval_pose.hpp"
#include "pose_kernels.hpp"
#include "matrix_kernels.hpp"


// ==================================================================== C ABI
struct nemo_ctx {
  nemo_config cfg;
  KParams P;
  size_t lds;
  MeshDesc *d_meshes;
  MeshDesc h_meshes[MAX_MESHES];
  int n_meshes;
  int vmax_seen, fmax_seen;
  float *d_adam;
  int adam_steps, adam_cap;
  float beta1, beta2, eps;
  u64 *d_counter;
  // pre-rendered grid (nemo_grid_prerender)
  int32_t *d_pre_slot, *d_pre_cnt, *d_pre_pairs;
  uint32_t *d_pre_list;
  RecVid *d_pre_vid;  // records [pairs][G][h*w] (vertex ids + covered flag) and their barycentrics [..][3] for the
  float *d_pre_bary;  // record-based matrix-core grid search (dense grids: no materialised renders), or null
  u64 *d_pre_mask;    // [pairs][pose blocks of GR_PR][pixel chunks]: pixels covered by any pose of the block
  int32_t *d_pre_perm, *d_pre_slotof;  // [pairs][G]: the record store's pose order (slot -> pose, pose -> slot)
  int env_grid_group;  // NEMO_GRID_GROUP=0: record store in grid order (no grouping of alike poses)
  float *d_pre_maps;  // dense renders [pairs][pre_block][h*w][D] for the MFMA grid search, or null (no store / large meshes)
  int pre_block;      // poses per pair held by d_pre_maps: pre_G (whole grid resident) or a block of it (rendered per call)
  // block render scratch (job = (pair, pose of the block)); in the resident case freed once the maps are rendered
  int32_t *d_blk_mesh, *d_blk_gidx, *d_blk_p2f;
  float *d_blk_C, *d_blk_t, *d_blk_R, *d_blk_T, *d_blk_bary;
  uint8_t *d_blk_live;  // [pairs][pixel chunks]: does any pose of the current block cover a pixel of the chunk
  const float *pre_gridC, *pre_gridT;
  int pre_G, pre_pairs;
  // grid-search workspace (grown on demand): job grouping and per-chunk partial sums
  int32_t *d_gs_order, *d_gs_start, *d_gs_jobslot;
  int32_t *d_gs_work;  // [0] = number of jobs without a pre-rendered pair, [1..] their ids (grid_kernel's work list)
  float *d_gs_partial;
  size_t gs_jobs_cap, gs_partial_cap;
  uint8_t *d_slot_order[MAX_MESHES];  // per mesh (MeshDesc::slot_order), owned by the context
  unsigned char *d_big;  // BIG-mesh table scratch
  size_t big_cap;
  int32_t *d_so;         // solve launch order workspace: cost[so_cap], order[so_cap]
  size_t so_cap;
  size_t big_blocks_req; // workgroups the BIG scratch was reserved for (re-applied when a mesh registration changes the slice size)
  int device;            // HIP device the context was created on
  int n_cu;              // its compute-unit count (cached: hipGetDeviceProperties is slow)
  // environment switches, read ONCE in nemo_ctx_create (hot calls never call getenv)
  int env_grid_no_mfma, env_solve_stagger, env_solve_order, env_solve_affinity, env_slot_identity;
  // nemo_solve's per-job tables of feature-row x bank-row dot products (bank_table_kernel): sized by
  // nemo_workspace_reserve for max_jobs jobs when that fits NEMO_SOLVE_TABLE_GB (default 64; 0 = never)
  float *d_tab;
  size_t tab_cap;            // floats
  double env_solve_table_gb;
  int env_table_wpj;         // NEMO_TABLE_WPJ: workgroups per job of bank_table_kernel
  int env_solve_lazy;        // NEMO_SOLVE_LAZY: 1 (default) the solve kernel fills its tables on demand; 0 bank_table_kernel fills them whole
  hipEvent_t ev_solve[3];    // recorded around the table kernel and the solve kernel of the last table-form call (first slice)
  int ev_solve_valid;
  int last_solve_mode;       // of the last solve call: 0 streaming, 1 tables filled by bank_table_kernel, 2 tables filled on demand
  int env_grid_records;  // NEMO_GRID_RECORDS: -1 (default) the record store when the dense renders do not fit or G >= 4096; 0 never; 1 whenever it fits
  double env_prerender_map_gb;
  char err[512];
};

static char g_err[512] = "";

static int fail(nemo_ctx *ctx, int code, const char *fmt, ...) {
  char *dst = ctx ? ctx->err : g_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  if (ctx) strncpy(g_err, ctx->err, 511);
  return code;
}
static int hip_fail(nemo_ctx *ctx, hipError_t e, const char *what) {
  return fail(ctx, NEMO_ERR_HIP_BASE - (int)e, "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(ctx, call)                                 \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
  } while (0)

// Sizes the LDS carve for the current largest mesh; called on registration.
// floats per job in the solve's dot-product tables (padded slot, matrix_kernels.hpp); 0 before any mesh is registered
static size_t table_stride(const nemo_ctx *ctx) {
  if (ctx->vmax_seen <= 0) return 0;
  return (size_t)table_rows((uint32_t)ctx->vmax_seen) * table_ld((uint32_t)ctx->cfg.win_h, (uint32_t)ctx->cfg.win_w);
}

static int plan_lds(nemo_ctx *ctx) {
  const nemo_config &c = ctx->cfg;
  const int vmax = ctx->vmax_seen > 0 ? ((ctx->vmax_seen + 63) & ~63) : 64;
  const int fmax = ctx->fmax_seen > 0 ? ((ctx->fmax_seen + 63) & ~63) : 64;
  const long hw = (long)c.win_h * c.win_w;
  if (hw > 65536)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "window %dx%d has more than 65536 pixels (16-bit pixel ids)", c.win_h, c.win_w);
  const long misc = (NW * RED_N + UST_N + c.raster_size + 12) * 4 + 128;
  const long fixed = (long)(NW * CH * (12 + 16) + 64 * (2 * NW + 24) + 192);  // score-phase scratch hosted by the z-buffer region
  auto cap_for = [&](long avail) {
    long cap = avail / (8 + 4);
    if (cap > hw) cap = hw;
    if (avail - fixed < cap * 4 + 64 * (cap / CH)) cap = (avail - fixed) / 5;
    return cap;
  };
  long cap = cap_for((long)LDS_BUDGET - (long)vmax * 16 - (long)fmax * FACE_BYTES - misc);
  int big = 0;
#ifdef NEMO_FORCE_BIG
  const bool force_big = true;
#else
  const bool force_big = false;
#endif
  if (force_big || cap < c.win_w || cap < (hw < 2048 ? hw : 2048)) {
    // the mesh tables would leave (almost) no z-buffer: keep them in global scratch instead (BIG kernels)
    big = 1;
    cap = cap_for((long)LDS_BUDGET - misc);
  }
  if (cap < c.win_w)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "window width %d does not fit the %d-byte LDS plan", c.win_w, LDS_BUDGET);
  ctx->P.cap = (int)cap;
  ctx->P.vmax = vmax;
  ctx->P.fmax = fmax;
  ctx->P.big = big;
  ctx->P.big_stride = (unsigned)(((size_t)vmax * 16 + (size_t)fmax * FACE_BYTES + 255) & ~(size_t)255);
  ctx->P.zb_bytes = (int)zb_region_bytes((int)cap);
  ctx->P.w_magic = (unsigned)(0x100000000ULL / (unsigned long long)c.win_w) + 1u;
  ctx->lds = lds_bytes((int)cap, big ? 0 : vmax, big ? 0 : fmax, c.win_w, c.raster_size);
  if (ctx->lds > (size_t)LDS_BUDGET) return fail(ctx, NEMO_ERR_UNSUPPORTED, "LDS plan %zu > budget", ctx->lds);
  return NEMO_OK;
}

// BIG meshes: one scratch slice (screen vertices, face ids, per-face fragment boxes) per workgroup of a launch.
constexpr unsigned BIG_MAX_BLOCKS = 2048;  // launches with more workgroups are issued in slices of this many
// Set-up side: (re)allocates the scratch for `nblocks` workgroups (nemo_workspace_reserve, nemo_mesh_register,
// nemo_grid_prerender -- all allowed to allocate and synchronise).
static int reserve_big_scratch(nemo_ctx *ctx, size_t nblocks) {
  if (nblocks > BIG_MAX_BLOCKS) nblocks = BIG_MAX_BLOCKS;
  if (nblocks > ctx->big_blocks_req) ctx->big_blocks_req = nblocks;
  if (!ctx->P.big) return NEMO_OK;
  const size_t need = ctx->big_blocks_req * (size_t)ctx->P.big_stride;
  if (need > ctx->big_cap) {  // the new block first: a failed allocation leaves the old scratch in place
    unsigned char *nb = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&nb, need));
    hipFree(ctx->d_big);
    ctx->d_big = nb;
    ctx->big_cap = need;
  }
  ctx->P.big_scratch = (unsigned long long)(uintptr_t)ctx->d_big;
  return NEMO_OK;
}
// Hot side: never allocates; a launch that needs more slices than were reserved is refused.
static int ensure_big_scratch(nemo_ctx *ctx, size_t nblocks) {
  if (!ctx->P.big) return NEMO_OK;
  if (nblocks > BIG_MAX_BLOCKS) nblocks = BIG_MAX_BLOCKS;
  const size_t need = nblocks * (size_t)ctx->P.big_stride;
  if (need > ctx->big_cap)
    return fail(ctx, NEMO_ERR_STATE,
                "large-mesh scratch holds %zu workgroup slices, this call needs %zu: call nemo_workspace_reserve first",
                ctx->P.big_stride ? ctx->big_cap / ctx->P.big_stride : 0, nblocks);
  ctx->P.big_scratch = (unsigned long long)(uintptr_t)ctx->d_big;
  return NEMO_OK;
}

// ---- nemo_solve_static: the solve's launch-order workspace in a file-scope device array instead of context
// memory, for callers that cannot allocate (HIP graph capture, serving loops).  Each GPU the library runs on holds
// its own copy of the array at its own address; the address is looked up once per device (static_ws_init, from
// nemo_ctx_create with that device current) and only that pointer -- never the array's name -- is used on the host.
constexpr int STATIC_MAX_JOBS = 16384;  // 2 x 16384 x 4 B = 128 KiB per GPU, held until the process ends
static __device__ __align__(256) int32_t scratch[2 * STATIC_MAX_JOBS];
constexpr int MAX_DEVICES = 64;
static int32_t *g_static_ws[MAX_DEVICES];
static std::mutex g_static_mu;
static int static_ws_init(nemo_ctx *ctx) {
  const int dev = ctx->device;
  if (dev < 0 || dev >= MAX_DEVICES) return NEMO_OK;  // nemo_solve_static refuses such a device
  std::lock_guard<std::mutex> lock(g_static_mu);
  if (!g_static_ws[dev]) {
    void *ptr = nullptr;
    HIP_TRY(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    g_static_ws[dev] = (int32_t *)ptr;
  }
  return NEMO_OK;
}

#define DISPATCH_DPB(kern, BIGV, grid, ...)                                                                       \
  do {                                                                                                            \
    const bool persp_ = ctx->P.persp != 0;                                                                        \
    if (ctx->cfg.feat_dim == 128) {                                                                               \
      if (persp_) hipLaunchKernelGGL((kern<128, true, BIGV>), grid, dim3(NT), ctx->lds, (hipStream_t)stream, __VA_ARGS__); \
      else hipLaunchKernelGGL((kern<128, false, BIGV>), grid, dim3(NT), ctx->lds, (hipStream_t)stream, __VA_ARGS__);      \
    } else {                                                                                                      \
      if (persp_) hipLaunchKernelGGL((kern<256, true, BIGV>), grid, dim3(NT), ctx->lds, (hipStream_t)stream, __VA_ARGS__); \
      else hipLaunchKernelGGL((kern<256, false, BIGV>), grid, dim3(NT), ctx->lds, (hipStream_t)stream, __VA_ARGS__);      \
    }                                                                                                             \
  } while (0)
// One workgroup per job; BIG meshes need a scratch slice per workgroup of the launch.
#define DISPATCH_DP(kern, grid, ...)                                                            \
  do {                                                                                          \
    if (ctx->P.big) {                                                                           \
      if ((grid).x > BIG_MAX_BLOCKS)                                                            \
        return fail(ctx, NEMO_ERR_UNSUPPORTED, "%u jobs in one call with a large mesh (limit %u)", (grid).x, BIG_MAX_BLOCKS); \
      int rc_ = ensure_big_scratch(ctx, (grid).x);                                              \
      if (rc_ != NEMO_OK) return rc_;                                                           \
      DISPATCH_DPB(kern, true, grid, __VA_ARGS__);                                              \
    } else {                                                                                    \
      DISPATCH_DPB(kern, false, grid, __VA_ARGS__);                                             \
    }                                                                                           \
  } while (0)
#define LAUNCH_PB(kern, grid, st, ...)                                                          \
  do {                                                                                          \
    if (ctx->P.big) {                                                                           \
      if (ctx->P.persp) hipLaunchKernelGGL((kern<true, true>), grid, dim3(NT), ctx->lds, st, __VA_ARGS__);   \
      else hipLaunchKernelGGL((kern<false, true>), grid, dim3(NT), ctx->lds, st, __VA_ARGS__);  \
    } else {                                                                                    \
      if (ctx->P.persp) hipLaunchKernelGGL((kern<true, false>), grid, dim3(NT), ctx->lds, st, __VA_ARGS__);  \
      else hipLaunchKernelGGL((kern<false, false>), grid, dim3(NT), ctx->lds, st, __VA_ARGS__); \
    }                                                                                           \
  } while (0)

// Renders grid poses [g0, g0 + gb) of every pre-render pair into ctx->d_pre_maps ([pair][gb][h*w][D], zero rows where
// nothing is covered) and refreshes the per-(pair, pixel chunk) coverage flags of the block.  Launches only.
static int render_pose_block(nemo_ctx *ctx, const float *d_grid_C, const float *d_grid_theta, int G, int g0, int gb,
                             hipStream_t st) {
  const int n_pairs = ctx->pre_pairs;
  const size_t jobs = (size_t)n_pairs * gb, stride = (size_t)ctx->cfg.win_h * ctx->cfg.win_w, D = (size_t)ctx->cfg.feat_dim;
  const int n_chunks = (int)((stride + GP_PIX - 1) / GP_PIX);
  hipLaunchKernelGGL(block_jobs_kernel, dim3((unsigned)((jobs + 255) / 256)), dim3(256), 0, st, n_pairs, G, g0, gb,
                     (const int32_t *)ctx->d_pre_pairs, (const int32_t *)(ctx->d_pre_pairs + n_pairs), ctx->d_blk_mesh,
                     ctx->d_blk_gidx);
  hipLaunchKernelGGL(gather_pose_kernel, dim3((unsigned)((jobs + 255) / 256)), dim3(256), 0, st, (int)jobs,
                     (const int32_t *)ctx->d_blk_gidx, d_grid_C, d_grid_theta, ctx->d_blk_C, ctx->d_blk_t);
  hipLaunchKernelGGL(campos_kernel, dim3((unsigned)((jobs + 63) / 64)), dim3(64), 0, st, (const float *)ctx->d_blk_C,
                     (const float *)ctx->d_blk_t, (const float *)nullptr, (int)jobs, ctx->d_blk_R, ctx->d_blk_T);
  LAUNCH_PB(raster_kernel, dim3((unsigned)jobs), st, ctx->P, ctx->d_meshes, (const int32_t *)ctx->d_blk_mesh,
            (const float *)ctx->d_blk_R, (const float *)ctx->d_blk_T, ctx->d_blk_p2f, ctx->d_blk_bary, (float *)nullptr);
  const size_t npix = jobs * stride, threads = npix * (D / 4);
  hipLaunchKernelGGL(interpolate_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (int)D, npix, stride,
                     ctx->d_meshes, (const int32_t *)ctx->d_blk_mesh, (const int32_t *)ctx->d_blk_p2f,
                     (const float *)ctx->d_blk_bary, ctx->d_pre_maps);
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_blk_live, 0, (size_t)n_pairs * n_chunks, st));
  hipLaunchKernelGGL(chunk_live_kernel, dim3((unsigned)jobs), dim3(256), 0, st, G, g0, gb, n_chunks, stride,
                     (const uint32_t *)ctx->d_pre_list, (const int32_t *)ctx->d_pre_cnt, ctx->d_blk_live);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

extern "C" {

const char *nemo_last_error(const nemo_ctx *ctx) { return ctx ? ctx->err : g_err; }

int nemo_ctx_create(const nemo_config *cfg, nemo_ctx **out) {
  if (!cfg || !out) return fail(nullptr, NEMO_ERR_ARG, "nemo_ctx_create: null argument");
  if (cfg->raster_size <= 0 || cfg->win_h <= 0 || cfg->win_w <= 0 || cfg->win_h > cfg->raster_size ||
      cfg->win_w > cfg->raster_size)
    return fail(nullptr, NEMO_ERR_ARG, "nemo_ctx_create: bad raster/window shape S=%d h=%d w=%d", cfg->raster_size,
                cfg->win_h, cfg->win_w);
  if (cfg->feat_dim != 128 && cfg->feat_dim != 256)
    return fail(nullptr, NEMO_ERR_UNSUPPORTED, "nemo_ctx_create: feat_dim %d (kernels are built for 128 and 256)",
                cfg->feat_dim);
  nemo_ctx *ctx = new nemo_ctx();
  memset(ctx, 0, sizeof(*ctx));
  ctx->cfg = *cfg;
  if (ctx->cfg.win_r0 < 0) ctx->cfg.win_r0 = cfg->raster_size / 2 - cfg->win_h / 2;
  if (ctx->cfg.win_c0 < 0) ctx->cfg.win_c0 = cfg->raster_size / 2 - cfg->win_w / 2;
  if (ctx->cfg.win_r0 + cfg->win_h > cfg->raster_size || ctx->cfg.win_c0 + cfg->win_w > cfg->raster_size) {
    delete ctx;
    return fail(nullptr, NEMO_ERR_ARG, "nemo_ctx_create: window leaves the raster");
  }
  KParams &P = ctx->P;
  P.S = cfg->raster_size; P.h = cfg->win_h; P.w = cfg->win_w; P.r0 = ctx->cfg.win_r0; P.c0 = ctx->cfg.win_c0;
  P.s = cfg->ndc_scale; P.persp = cfg->perspective_correct; P.clutter = cfg->clutter; P.keps = cfg->k_eps;
  P.inv_hw = (float)(1.0 / ((double)cfg->win_h * cfg->win_w));
  int rc = plan_lds(ctx);
  if (rc != NEMO_OK) { delete ctx; return rc; }
  hipError_t e = hipMalloc((void **)&ctx->d_meshes, sizeof(MeshDesc) * MAX_MESHES);
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_counter, 2 * sizeof(u64));   // [0] covered pixel-steps, [1] table blocks computed
  if (e == hipSuccess) e = hipMemset(ctx->d_counter, 0, 2 * sizeof(u64));
  ctx->adam_cap = 4096;
  for (int i = 0; i < 3; ++i) if (e == hipSuccess) e = hipEventCreate(&ctx->ev_solve[i]);
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_adam, sizeof(float) * 2 * ctx->adam_cap);
  if (e != hipSuccess) {
    int code = hip_fail(nullptr, e, "nemo_ctx_create: hipMalloc");
    delete ctx;
    return code;
  }
  // opt in to > 64 KiB dynamic LDS for every pose kernel instantiation
  const int maxlds = LDS_BUDGET;
#define SET_LDS(k) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds)
#define SET_LDS_B(k, ...) SET_LDS((k<__VA_ARGS__, false>)); SET_LDS((k<__VA_ARGS__, true>))
  SET_LDS_B(raster_kernel, false); SET_LDS_B(raster_kernel, true);
  SET_LDS_B(prerender_kernel, false); SET_LDS_B(prerender_kernel, true);
  SET_LDS_B(solve_cost_kernel, false); SET_LDS_B(solve_cost_kernel, true);
  SET_LDS(grid_mfma_lds_kernel<128>); SET_LDS(grid_mfma_lds_kernel<256>);
  SET_LDS_B(score_kernel, 128, false); SET_LDS_B(score_kernel, 128, true);
  SET_LDS_B(score_kernel, 256, false); SET_LDS_B(score_kernel, 256, true);
  SET_LDS_B(grid_kernel, 128, false); SET_LDS_B(grid_kernel, 128, true);
  SET_LDS_B(grid_kernel, 256, false); SET_LDS_B(grid_kernel, 256, true);
  SET_LDS_B(solve_kernel, 128, false); SET_LDS_B(solve_kernel, 128, true);
  SET_LDS_B(solve_kernel, 256, false); SET_LDS_B(solve_kernel, 256, true);
  SET_LDS((solve_kernel<128, false, false, true>)); SET_LDS((solve_kernel<128, true, false, true>));
  SET_LDS((solve_kernel<256, false, false, true>)); SET_LDS((solve_kernel<256, true, false, true>));
  SET_LDS((solve_kernel<128, false, false, true, true>)); SET_LDS((solve_kernel<128, true, false, true, true>));
  SET_LDS((solve_kernel<256, false, false, true, true>)); SET_LDS((solve_kernel<256, true, false, true, true>));
  SET_LDS_B(interp_bwd_kernel, 128, false); SET_LDS_B(interp_bwd_kernel, 128, true);
  SET_LDS_B(interp_bwd_kernel, 256, false); SET_LDS_B(interp_bwd_kernel, 256, true);
#undef SET_LDS_B
#undef SET_LDS
  hipFuncSetAttribute((const void *)grid_mfma_rec_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GR_LDS);
  hipFuncSetAttribute((const void *)grid_mfma_rec_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GR_LDS);
  hipFuncSetAttribute((const void *)feature_head_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void *)feature_head_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void *)bank_table_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CK_ROWS * (128 + 4) * 4);
  hipFuncSetAttribute((const void *)bank_table_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CK_ROWS * (256 + 4) * 4);
  hipFuncSetAttribute((const void *)clutter_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CK_ROWS * (128 + 4) * 4);
  hipFuncSetAttribute((const void *)clutter_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CK_ROWS * (256 + 4) * 4);
  // device facts and environment switches are read here, once; hot calls never query either
  ctx->n_cu = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDevice(&ctx->device) != hipSuccess) ctx->device = 0;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
  }
  auto env_int = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
  ctx->env_grid_no_mfma = env_int("NEMO_GRID_NO_MFMA", 0);
  ctx->env_solve_stagger = env_int("NEMO_SOLVE_STAGGER", 0);
  {
    const char *l_ = getenv("NEMO_SOLVE_LAZY");
    ctx->env_solve_lazy = l_ ? atoi(l_) : 1;
    const char *w_ = getenv("NEMO_TABLE_WPJ");
    ctx->env_table_wpj = w_ ? std::max(1, std::min(64, atoi(w_))) : 2;
    const char *e_ = getenv("NEMO_SOLVE_TABLE_GB");
    ctx->env_solve_table_gb = e_ ? atof(e_) : 64.0;
  }
  ctx->env_solve_order = env_int("NEMO_SOLVE_ORDER", 1);
  ctx->env_solve_affinity = env_int("NEMO_SOLVE_AFFINITY", 1);
  ctx->env_slot_identity = env_int("NEMO_SLOT_ORDER", 1) == 0;
  ctx->env_grid_records = env_int("NEMO_GRID_RECORDS", -1);
  ctx->env_grid_group = env_int("NEMO_GRID_GROUP", 1);
  {
    const char *v = getenv("NEMO_PRERENDER_MAP_GB");
    ctx->env_prerender_map_gb = v ? atof(v) : 48.0;
  }
  rc = static_ws_init(ctx);
  if (rc != NEMO_OK) { nemo_ctx_destroy(ctx); return rc; }
  *out = ctx;
  return NEMO_OK;
}

/* Set-up call: sizes every workspace the hot entry points use, so that none of them allocates, frees or
 * synchronises.  max_jobs = the largest N of any later call, max_grid_poses = the largest G of nemo_grid_search
 * (0: no grid search).  Growing only; may be called again.  Workspaces belong to the context, so a context
 * serves ONE stream at a time. */
int nemo_workspace_reserve(nemo_ctx *ctx, int32_t max_jobs, int32_t max_grid_poses) {
  if (!ctx || max_jobs < 0 || max_grid_poses < 0) return fail(ctx, NEMO_ERR_ARG, "nemo_workspace_reserve: bad argument");
  const size_t N = (size_t)max_jobs, G = (size_t)max_grid_poses;
  if (N > ctx->so_cap) {  // solve launch order: cost[N], order[N]
    hipFree(ctx->d_so);
    ctx->d_so = nullptr;
    ctx->so_cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_so, sizeof(int32_t) * 2 * N));
    ctx->so_cap = N;
  }
  if (N > ctx->gs_jobs_cap) {  // grid search: job grouping by pre-render slot
    hipFree(ctx->d_gs_order); hipFree(ctx->d_gs_jobslot); hipFree(ctx->d_gs_start); hipFree(ctx->d_gs_work);
    ctx->d_gs_order = ctx->d_gs_jobslot = ctx->d_gs_start = ctx->d_gs_work = nullptr;
    ctx->gs_jobs_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_gs_order, sizeof(int32_t) * N));
    HIP_TRY(ctx, hipMalloc(&ctx->d_gs_jobslot, sizeof(int32_t) * N));
    HIP_TRY(ctx, hipMalloc(&ctx->d_gs_start, sizeof(int32_t) * (GP_MAX_PAIRS + 1)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_gs_work, sizeof(int32_t) * (N + 1)));
    ctx->gs_jobs_cap = N;
  }
  const size_t n_chunks = ((size_t)ctx->cfg.win_h * ctx->cfg.win_w + GP_PIX - 1) / GP_PIX;
  const size_t need_partial = n_chunks * N * G;
  if (need_partial > ctx->gs_partial_cap) {  // matrix-core grid search: per-chunk partial sums
    hipFree(ctx->d_gs_partial);
    ctx->d_gs_partial = nullptr;
    ctx->gs_partial_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_gs_partial, sizeof(float) * need_partial));
    ctx->gs_partial_cap = need_partial;
  }
  // the solve's dot-product tables: as many jobs as fit NEMO_SOLVE_TABLE_GB, at most max_jobs (a call with more jobs goes
  // through in slices of that many; under 256 jobs' worth -- one per CU -- the solve streams the feature rows instead)
  {
    const size_t stride = table_stride(ctx);
    if (!ctx->P.big && stride > 0 && N > 0) {
      size_t jobs = (size_t)(ctx->env_solve_table_gb * 1073741824.0 / ((double)stride * 4.0));
      if (jobs > N) jobs = N;
      if (jobs < N && jobs < 256) jobs = 0;
      if (jobs * stride > ctx->tab_cap) {
        hipFree(ctx->d_tab);
        ctx->d_tab = nullptr;
        ctx->tab_cap = 0;
        // not getting the tables is not an error: the solve then streams the feature rows
        if (hipMalloc((void **)&ctx->d_tab, sizeof(float) * jobs * stride) == hipSuccess) ctx->tab_cap = jobs * stride;
        else { ctx->d_tab = nullptr; (void)hipGetLastError(); }
      }
    }
  }
  // large meshes: one table slice per workgroup of the widest launch (per-job kernels: N; grid search: sliced)
  size_t blocks = N;
  if (G > 0) {
    const size_t gb = 8 * ((N + 7) / 8) * ((G + GRID_POSES_PER_WG - 1) / GRID_POSES_PER_WG);
    if (gb > blocks) blocks = gb;
  }
  return reserve_big_scratch(ctx, blocks);
}

static void free_block_scratch(nemo_ctx *ctx, bool keep_live) {
  hipFree(ctx->d_blk_mesh); hipFree(ctx->d_blk_gidx); hipFree(ctx->d_blk_p2f); hipFree(ctx->d_blk_C); hipFree(ctx->d_blk_t);
  hipFree(ctx->d_blk_R); hipFree(ctx->d_blk_T); hipFree(ctx->d_blk_bary);
  ctx->d_blk_mesh = ctx->d_blk_gidx = ctx->d_blk_p2f = nullptr;
  ctx->d_blk_C = ctx->d_blk_t = ctx->d_blk_R = ctx->d_blk_T = ctx->d_blk_bary = nullptr;
  if (!keep_live) { hipFree(ctx->d_blk_live); ctx->d_blk_live = nullptr; }
}

int nemo_ctx_destroy(nemo_ctx *ctx) {
  if (!ctx) return NEMO_OK;
  free_block_scratch(ctx, false);
  hipFree(ctx->d_meshes);
  hipFree(ctx->d_counter);
  hipFree(ctx->d_adam);
  hipFree(ctx->d_pre_slot); hipFree(ctx->d_pre_cnt); hipFree(ctx->d_pre_pairs); hipFree(ctx->d_pre_list);
  hipFree(ctx->d_pre_vid); hipFree(ctx->d_pre_bary); hipFree(ctx->d_pre_mask); hipFree(ctx->d_pre_perm); hipFree(ctx->d_pre_slotof);
  hipFree(ctx->d_pre_maps); hipFree(ctx->d_gs_order); hipFree(ctx->d_gs_start); hipFree(ctx->d_gs_jobslot);
  hipFree(ctx->d_gs_partial); hipFree(ctx->d_gs_work); hipFree(ctx->d_big); hipFree(ctx->d_so); hipFree(ctx->d_tab);
  for (int i = 0; i < ctx->n_meshes; ++i) hipFree(ctx->d_slot_order[i]);
  for (int i = 0; i < 3; ++i) if (ctx->ev_solve[i]) hipEventDestroy(ctx->ev_solve[i]);
  delete ctx;
  return NEMO_OK;
}

int nemo_mesh_register(nemo_ctx *ctx, const float *d_verts, int32_t V, const int32_t *d_faces, int32_t F,
                       const float *d_bank) {
  if (!ctx) return fail(nullptr, NEMO_ERR_ARG, "nemo_mesh_register: null ctx");
  if (!d_verts || !d_faces || V <= 0 || F <= 0)
    return fail(ctx, NEMO_ERR_ARG, "nemo_mesh_register: bad mesh (V=%d F=%d)", V, F);
  if (ctx->n_meshes >= MAX_MESHES) return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_mesh_register: more than %d meshes", MAX_MESHES);
  if (V > 65535 || F > 65535)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_mesh_register: V=%d F=%d (vertex and face ids are 16-bit in LDS)", V, F);
  const int old_vmax = ctx->vmax_seen, old_fmax = ctx->fmax_seen;
  if (V > ctx->vmax_seen) ctx->vmax_seen = V;
  if (F > ctx->fmax_seen) ctx->fmax_seen = F;
  int rc = plan_lds(ctx);
  if (rc != NEMO_OK) { ctx->vmax_seen = old_vmax; ctx->fmax_seen = old_fmax; plan_lds(ctx); return rc; }
  const int id = ctx->n_meshes;
  // Slot order (MeshDesc::slot_order): walk the faces in index order -- the order of the score loop's face-major
  // list -- and give every face the permutation of its three vertices that leaves the most register sets of the
  // previous face untouched.  Results do not depend on it (the dot products are stored back by vertex slot).
  std::vector<int32_t> hf((size_t)3 * F);
  HIP_TRY(ctx, hipMemcpy(hf.data(), d_faces, sizeof(int32_t) * 3 * (size_t)F, hipMemcpyDeviceToHost));
  for (size_t k = 0; k < hf.size(); ++k)
    if (hf[k] < 0 || hf[k] >= V) {
      ctx->vmax_seen = old_vmax; ctx->fmax_seen = old_fmax; plan_lds(ctx);
      return fail(ctx, NEMO_ERR_ARG, "nemo_mesh_register: face %zu refers to vertex %d (V = %d)", k / 3, hf[k], V);
    }
  if ((rc = reserve_big_scratch(ctx, 0)) != NEMO_OK) {  // the slice size may have grown: re-apply the reservation
    ctx->vmax_seen = old_vmax; ctx->fmax_seen = old_fmax; plan_lds(ctx);  // as the error paths above: the mesh was not registered
    reserve_big_scratch(ctx, 0);
    return rc;
  }
  std::vector<uint8_t> order((size_t)F);
  {
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    int32_t prev[3] = {-1, -1, -1};
    const bool identity = ctx->env_slot_identity != 0;  // NEMO_SLOT_ORDER=0 (invariance test): every face keeps its own order
    for (int f = 0; f < F; ++f) {
      const int32_t *t = &hf[(size_t)3 * f];
      int best = -1, bp = 0;
      for (int p = 0; p < 6; ++p) {
        const int m = (t[perms[p][0]] == prev[0]) + (t[perms[p][1]] == prev[1]) + (t[perms[p][2]] == prev[2]);
        if (m > best) { best = m; bp = p; }
      }
      if (identity) bp = 0;
      for (int k = 0; k < 3; ++k) prev[k] = t[perms[bp][k]];
      order[f] = (uint8_t)(perms[bp][0] | (perms[bp][1] << 2) | (perms[bp][2] << 4));
    }
  }
  uint8_t *d_order = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)&d_order, (size_t)F));
  hipError_t ce = hipMemcpy(d_order, order.data(), (size_t)F, hipMemcpyHostToDevice);
  if (ce != hipSuccess) { hipFree(d_order); return hip_fail(ctx, ce, "nemo_mesh_register: slot order upload"); }
  ctx->d_slot_order[id] = d_order;
  ctx->h_meshes[id] = MeshDesc{d_verts, d_faces, d_bank, V, F, d_order};
  ce = hipMemcpy(ctx->d_meshes + id, &ctx->h_meshes[id], sizeof(MeshDesc), hipMemcpyHostToDevice);
  if (ce != hipSuccess) { hipFree(d_order); ctx->d_slot_order[id] = nullptr; return hip_fail(ctx, ce, "nemo_mesh_register: descriptor upload"); }
  ctx->n_meshes = id + 1;
  return id;
}

int nemo_set_schedule(nemo_ctx *ctx, const float *h_tab, int32_t steps, float beta1, float beta2, float eps) {
  if (!ctx || !h_tab || steps <= 0) return fail(ctx, NEMO_ERR_ARG, "nemo_set_schedule: bad argument");
  if (steps > ctx->adam_cap) return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_set_schedule: steps %d > %d", steps, ctx->adam_cap);
  HIP_TRY(ctx, hipMemcpy(ctx->d_adam, h_tab, sizeof(float) * 2 * steps, hipMemcpyHostToDevice));
  ctx->adam_steps = steps; ctx->beta1 = beta1; ctx->beta2 = beta2; ctx->eps = eps;
  return NEMO_OK;
}

int nemo_campos_to_R_T(nemo_ctx *ctx, const float *d_C, const float *d_theta, const float *d_extra, int32_t N,
                       float *d_R, float *d_T, void *stream) {
  if (!ctx || N < 0) return fail(ctx, NEMO_ERR_ARG, "nemo_campos_to_R_T: bad argument");
  if (N == 0) return NEMO_OK;
  if (!d_C || !d_theta || !d_R || !d_T) return fail(ctx, NEMO_ERR_ARG, "nemo_campos_to_R_T: null buffer");
  hipLaunchKernelGGL(campos_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_C, d_theta, d_extra, N, d_R, d_T);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

static int check_ready(nemo_ctx *ctx, const char *who, int32_t N) {
  if (!ctx) return fail(nullptr, NEMO_ERR_ARG, "%s: null ctx", who);
  if (N < 0) return fail(ctx, NEMO_ERR_ARG, "%s: negative job count", who);
  if (N == 0) return 1;  // empty batch: nothing to do (callers return NEMO_OK)
  if (ctx->n_meshes == 0) return fail(ctx, NEMO_ERR_STATE, "%s: no mesh registered", who);
  return NEMO_OK;
}

int nemo_rasterize(nemo_ctx *ctx, const int32_t *d_mesh_idx, const float *d_R, const float *d_T, int32_t N,
                   int32_t *d_pix_to_face, float *d_bary, float *d_zbuf, void *stream) {
  int rc = check_ready(ctx, "nemo_rasterize", N);
  if (rc) return rc > 0 ? NEMO_OK : rc;
  if (!d_R || !d_T || !d_pix_to_face || !d_bary) return fail(ctx, NEMO_ERR_ARG, "nemo_rasterize: null buffer");
  if (N == 0) return NEMO_OK;
  if (ctx->P.big && (unsigned)N > BIG_MAX_BLOCKS)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_rasterize: %d jobs in one call with a large mesh (limit %u)", N, BIG_MAX_BLOCKS);
  if ((rc = ensure_big_scratch(ctx, (size_t)N)) != NEMO_OK) return rc;
  LAUNCH_PB(raster_kernel, dim3(N), (hipStream_t)stream, ctx->P, ctx->d_meshes, d_mesh_idx, d_R, d_T, d_pix_to_face, d_bary,
            d_zbuf);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

int nemo_interpolate(nemo_ctx *ctx, const int32_t *d_mesh_idx, const int32_t *d_pix_to_face, const float *d_bary,
                     int32_t N, float *d_out, void *stream) {
  int rc = check_ready(ctx, "nemo_interpolate", N);
  if (rc) return rc > 0 ? NEMO_OK : rc;
  if (!d_pix_to_face || !d_bary || !d_out) return fail(ctx, NEMO_ERR_ARG, "nemo_interpolate: null buffer");
  if (N == 0) return NEMO_OK;
  const size_t hw = (size_t)ctx->P.h * ctx->P.w, npix = hw * N;
  const size_t threads = npix * (ctx->cfg.feat_dim / 4);
  hipLaunchKernelGGL(interpolate_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     ctx->cfg.feat_dim, npix, hw, ctx->d_meshes, d_mesh_idx, d_pix_to_face, d_bary, d_out);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

int nemo_score_fwd_bwd(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx,
                       const float *d_feat, const float *d_clu, const float *d_C, const float *d_theta, float *d_loss,
                       float *d_grad, int32_t *d_ncov, void *stream) {
  int rc = check_ready(ctx, "nemo_score_fwd_bwd", N);
  if (rc) return rc > 0 ? NEMO_OK : rc;
  if (!d_feat || !d_C || !d_theta || !d_loss) return fail(ctx, NEMO_ERR_ARG, "nemo_score_fwd_bwd: null buffer");
  if (ctx->cfg.clutter && !d_clu) return fail(ctx, NEMO_ERR_ARG, "nemo_score_fwd_bwd: clutter loss needs d_clu");
  if (N == 0) return NEMO_OK;
  DISPATCH_DP(score_kernel, dim3(N), ctx->P, ctx->d_meshes, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C, d_theta, d_loss,
              d_grad, d_ncov, ctx->d_counter);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

int nemo_grid_search(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx,
                     const int32_t *d_grid_idx, const float *d_feat, const float *d_clu, const float *d_grid_C,
                     const float *d_grid_theta, int32_t G, float *d_losses, float *d_C0, float *d_theta0,
                     int32_t *d_best, void *stream) {
  int rc = check_ready(ctx, "nemo_grid_search", N);
  if (rc) return rc > 0 ? NEMO_OK : rc;
  if (!d_feat || !d_grid_C || !d_grid_theta || !d_losses || !d_C0 || !d_theta0 || G <= 0)
    return fail(ctx, NEMO_ERR_ARG, "nemo_grid_search: bad argument");
  if (ctx->cfg.clutter && !d_clu) return fail(ctx, NEMO_ERR_ARG, "nemo_grid_search: clutter loss needs d_clu");
  if (N == 0) return NEMO_OK;
  const long long blocks = 8LL * ((N + 7) / 8) * ((G + GRID_POSES_PER_WG - 1) / GRID_POSES_PER_WG);
  if (blocks > 0x7fffffffLL) return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_grid_search: N*G too large for one launch");
  PreStore pre{nullptr, nullptr, nullptr, 0};
  const bool store_ok = ctx->pre_pairs > 0 && ctx->pre_G == G && ctx->pre_gridC == d_grid_C && ctx->pre_gridT == d_grid_theta;
  if (store_ok) pre = PreStore{ctx->d_pre_slot, ctx->d_pre_cnt, ctx->d_pre_list, (size_t)ctx->cfg.win_h * ctx->cfg.win_w};
  const bool mfma = store_ok && ctx->d_pre_maps && !ctx->env_grid_no_mfma;
  const bool rec = store_ok && ctx->d_pre_vid && !ctx->env_grid_no_mfma;
  const int32_t *d_work = nullptr;
  if (rec) {
    // dense grids: the pose operand is built from the record store, nothing is rendered or materialised per call
    const int hw = ctx->cfg.win_h * ctx->cfg.win_w, n_chunks = (hw + GP_PIX - 1) / GP_PIX;
    const size_t need_partial = (size_t)n_chunks * N * G;
    if ((size_t)N > ctx->gs_jobs_cap || need_partial > ctx->gs_partial_cap)
      return fail(ctx, NEMO_ERR_STATE, "nemo_grid_search: workspace reserved for %zu jobs / %zu partial sums, this call needs "
                  "%d / %zu: call nemo_workspace_reserve(ctx, >= %d, >= %d) first", ctx->gs_jobs_cap, ctx->gs_partial_cap, N,
                  need_partial, N, G);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grid_group_kernel, dim3(1), dim3(1024), 0, st, N, d_mesh_idx, d_grid_idx, ctx->d_pre_slot,
                       ctx->pre_pairs, ctx->d_gs_order, ctx->d_gs_start, ctx->d_gs_jobslot, ctx->d_gs_work);
    d_work = ctx->d_gs_work;
    const int con = (ctx->cfg.clutter && d_clu) ? 1 : 0;
    const int n_pb = (G + GR_PR - 1) / GR_PR;
    const dim3 grd((unsigned)((size_t)ctx->pre_pairs * n_chunks * n_pb)), blk(GR_NT);
    if (ctx->cfg.feat_dim == 128)
      hipLaunchKernelGGL((grid_mfma_rec_kernel<128>), grd, blk, GR_LDS, st, hw, N, G, n_chunks, d_img_idx, ctx->d_gs_order,
                         ctx->d_gs_start, d_feat, d_clu, (const MeshDesc *)ctx->d_meshes, (const int32_t *)ctx->d_pre_pairs,
                         (const RecVid *)ctx->d_pre_vid, (const float *)ctx->d_pre_bary, ctx->d_gs_partial, (const u64 *)ctx->d_pre_mask, n_pb, con);
    else
      hipLaunchKernelGGL((grid_mfma_rec_kernel<256>), grd, blk, GR_LDS, st, hw, N, G, n_chunks, d_img_idx, ctx->d_gs_order,
                         ctx->d_gs_start, d_feat, d_clu, (const MeshDesc *)ctx->d_meshes, (const int32_t *)ctx->d_pre_pairs,
                         (const RecVid *)ctx->d_pre_vid, (const float *)ctx->d_pre_bary, ctx->d_gs_partial, (const u64 *)ctx->d_pre_mask, n_pb, con);
    hipLaunchKernelGGL(grid_finish_kernel, dim3(N), dim3(256), 0, st, hw, ctx->P.inv_hw, N, G, n_chunks, d_img_idx,
                       ctx->d_gs_jobslot, d_clu, ctx->d_gs_partial, ctx->d_pre_cnt, d_losses, ctx->d_counter, con,
                       (const int32_t *)ctx->d_pre_perm);
    HIP_TRY(ctx, hipGetLastError());
  } else if (mfma) {
    // matrix-core path over the dense pre-rendered maps; jobs without a stored pair fall through to grid_kernel
    const int hw = ctx->cfg.win_h * ctx->cfg.win_w, n_chunks = (hw + GP_PIX - 1) / GP_PIX;
    const size_t need_partial = (size_t)n_chunks * N * G;
    if ((size_t)N > ctx->gs_jobs_cap || need_partial > ctx->gs_partial_cap)
      return fail(ctx, NEMO_ERR_STATE, "nemo_grid_search: workspace reserved for %zu jobs / %zu partial sums, this call needs "
                  "%d / %zu: call nemo_workspace_reserve(ctx, >= %d, >= %d) first", ctx->gs_jobs_cap, ctx->gs_partial_cap, N,
                  need_partial, N, G);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grid_group_kernel, dim3(1), dim3(1024), 0, st, N, d_mesh_idx, d_grid_idx, ctx->d_pre_slot,
                       ctx->pre_pairs, ctx->d_gs_order, ctx->d_gs_start, ctx->d_gs_jobslot, ctx->d_gs_work);
    d_work = ctx->d_gs_work;
    const int con = (ctx->cfg.clutter && d_clu) ? 1 : 0;
    // pose blocks: the whole grid when its renders are resident, otherwise block after block through the same buffer
    // (render -> contraction -> reuse; everything was allocated by nemo_grid_prerender)
    for (int g0 = 0; g0 < G; g0 += ctx->pre_block) {
      const int gb = G - g0 < ctx->pre_block ? G - g0 : ctx->pre_block;
      if (ctx->pre_block < G) {
        int rc_ = render_pose_block(ctx, d_grid_C, d_grid_theta, G, g0, gb, st);
        if (rc_ != NEMO_OK) return rc_;
      }
      // pose tiles per workgroup (PTW) and job tiles per pass (JTW): 16 waves, at most 8 staged tiles of 32 rows
      const int PT = (gb + 31) / 32, PTW = PT <= 5 ? PT : 4;
      const int JTW = (16 / PTW) < (8 - PTW) ? (16 / PTW) : (8 - PTW);
      const dim3 grd((unsigned)(ctx->pre_pairs * n_chunks), (unsigned)((PT + PTW - 1) / PTW)), blk(1024);
      const size_t sm = ((size_t)(PTW + JTW) * 32 * GQ_ROW + (size_t)JTW * 32) * sizeof(float);
      if (ctx->cfg.feat_dim == 128)
        hipLaunchKernelGGL((grid_mfma_lds_kernel<128>), grd, blk, sm, st, hw, N, G, gb, g0, n_chunks, PTW, JTW, d_img_idx,
                           ctx->d_gs_order, ctx->d_gs_start, d_feat, d_clu, ctx->d_pre_maps, ctx->d_gs_partial,
                           ctx->d_blk_live, con);
      else
        hipLaunchKernelGGL((grid_mfma_lds_kernel<256>), grd, blk, sm, st, hw, N, G, gb, g0, n_chunks, PTW, JTW, d_img_idx,
                           ctx->d_gs_order, ctx->d_gs_start, d_feat, d_clu, ctx->d_pre_maps, ctx->d_gs_partial,
                           ctx->d_blk_live, con);
    }
    hipLaunchKernelGGL(grid_finish_kernel, dim3(N), dim3(256), 0, st, hw, ctx->P.inv_hw, N, G, n_chunks, d_img_idx,
                       ctx->d_gs_jobslot, d_clu, ctx->d_gs_partial, ctx->d_pre_cnt, d_losses, ctx->d_counter, con,
                       (const int32_t *)nullptr);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (d_work) {
    // only the jobs the matrix-core path left over (device-side list): a fixed, small grid with a stride loop
    const unsigned nb = (unsigned)(blocks < 2048 ? blocks : 2048);  // a multiple of 8 either way
    int rc_ = ensure_big_scratch(ctx, nb);
    if (rc_ != NEMO_OK) return rc_;
    if (ctx->P.big)
      DISPATCH_DPB(grid_kernel, true, dim3(nb), ctx->P, ctx->d_meshes, N, d_img_idx, d_mesh_idx, d_grid_idx, d_feat, d_clu,
                   d_grid_C, d_grid_theta, G, d_losses, ctx->d_counter, pre, d_work, 0u, nb);
    else
      DISPATCH_DPB(grid_kernel, false, dim3(nb), ctx->P, ctx->d_meshes, N, d_img_idx, d_mesh_idx, d_grid_idx, d_feat, d_clu,
                   d_grid_C, d_grid_theta, G, d_losses, ctx->d_counter, pre, d_work, 0u, nb);
  } else if (ctx->P.big) {
    int rc_ = ensure_big_scratch(ctx, (size_t)blocks);
    if (rc_ != NEMO_OK) return rc_;
    for (long long b0 = 0; b0 < blocks; b0 += BIG_MAX_BLOCKS) {
      const unsigned nb = (unsigned)(blocks - b0 < (long long)BIG_MAX_BLOCKS ? blocks - b0 : (long long)BIG_MAX_BLOCKS);
      DISPATCH_DPB(grid_kernel, true, dim3(nb), ctx->P, ctx->d_meshes, N, d_img_idx, d_mesh_idx, d_grid_idx, d_feat, d_clu,
                   d_grid_C, d_grid_theta, G, d_losses, ctx->d_counter, pre, (const int32_t *)nullptr, (unsigned)b0, 0u);
    }
  } else {
    DISPATCH_DPB(grid_kernel, false, dim3((unsigned)blocks), ctx->P, ctx->d_meshes, N, d_img_idx, d_mesh_idx, d_grid_idx, d_feat,
                 d_clu, d_grid_C, d_grid_theta, G, d_losses, ctx->d_counter, pre, (const int32_t *)nullptr, 0u, 0u);
  }
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(grid_argmin_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, N, G, d_losses, d_grid_idx,
                     d_grid_C, d_grid_theta, d_C0, d_theta0, d_best);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

int nemo_grid_prerender(nemo_ctx *ctx, const float *d_grid_C, const float *d_grid_theta, int32_t n_grids, int32_t G,
                        const int32_t *h_pair_mesh, const int32_t *h_pair_grid, int32_t n_pairs, void *stream) {
  if (!ctx) return fail(nullptr, NEMO_ERR_ARG, "nemo_grid_prerender: null ctx");
  hipFree(ctx->d_pre_slot); hipFree(ctx->d_pre_cnt); hipFree(ctx->d_pre_pairs); hipFree(ctx->d_pre_list);
  hipFree(ctx->d_pre_maps); hipFree(ctx->d_pre_vid); hipFree(ctx->d_pre_bary); hipFree(ctx->d_pre_mask);
  hipFree(ctx->d_pre_perm); hipFree(ctx->d_pre_slotof);
  ctx->d_pre_mask = nullptr; ctx->d_pre_perm = ctx->d_pre_slotof = nullptr;
  free_block_scratch(ctx, false);
  ctx->pre_block = 0;
  ctx->d_pre_slot = ctx->d_pre_cnt = ctx->d_pre_pairs = nullptr;
  ctx->d_pre_list = nullptr;
  ctx->d_pre_maps = nullptr;
  ctx->d_pre_vid = nullptr;
  ctx->d_pre_bary = nullptr;
  ctx->pre_pairs = 0;
  if (n_pairs == 0) return NEMO_OK;  // store dropped
  if (!d_grid_C || !d_grid_theta || !h_pair_mesh || !h_pair_grid || n_pairs < 0 || G <= 0 || n_grids <= 0)
    return fail(ctx, NEMO_ERR_ARG, "nemo_grid_prerender: bad argument");
  if (n_grids > MAX_GRIDS) return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_grid_prerender: more than %d grids", MAX_GRIDS);
  std::vector<int32_t> h_slot((size_t)MAX_MESHES * MAX_GRIDS, -1);
  for (int i = 0; i < n_pairs; ++i) {
    const int m = h_pair_mesh[i], g = h_pair_grid[i];
    if (m < 0 || m >= ctx->n_meshes || g < 0 || g >= n_grids)
      return fail(ctx, NEMO_ERR_ARG, "nemo_grid_prerender: pair %d = (mesh %d, grid %d) out of range", i, m, g);
    h_slot[m * MAX_GRIDS + g] = i;
  }
  const size_t stride = (size_t)ctx->cfg.win_h * ctx->cfg.win_w;
  HIP_TRY(ctx, hipMalloc(&ctx->d_pre_slot, sizeof(int32_t) * h_slot.size()));
  HIP_TRY(ctx, hipMalloc(&ctx->d_pre_pairs, sizeof(int32_t) * 2 * n_pairs));
  HIP_TRY(ctx, hipMalloc(&ctx->d_pre_cnt, sizeof(int32_t) * (size_t)n_pairs * G));
  HIP_TRY(ctx, hipMalloc(&ctx->d_pre_list, sizeof(uint32_t) * (size_t)n_pairs * G * stride));
  HIP_TRY(ctx, hipMemcpy(ctx->d_pre_slot, h_slot.data(), sizeof(int32_t) * h_slot.size(), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_pre_pairs, h_pair_mesh, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_pre_pairs + n_pairs, h_pair_grid, sizeof(int32_t) * n_pairs, hipMemcpyHostToDevice));
  const dim3 grid((unsigned)(n_pairs * G));
  if (ctx->P.big && grid.x > BIG_MAX_BLOCKS)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_grid_prerender: %u (pair, pose) renders with a large mesh (limit %u per call)", grid.x, BIG_MAX_BLOCKS);
  {
    int rc_ = reserve_big_scratch(ctx, grid.x);
    if (rc_ != NEMO_OK) return rc_;
  }
  LAUNCH_PB(prerender_kernel, grid, (hipStream_t)stream, ctx->P, ctx->d_meshes, ctx->d_pre_pairs, ctx->d_pre_pairs + n_pairs,
            d_grid_C, d_grid_theta, G, ctx->d_pre_list, ctx->d_pre_cnt, stride);
  HIP_TRY(ctx, hipGetLastError());
  ctx->pre_gridC = d_grid_C; ctx->pre_gridT = d_grid_theta;
  ctx->pre_G = G; ctx->pre_pairs = n_pairs;
  // Dense renders for the MFMA grid search: campos -> rasterize -> interpolate with the standalone kernels
  // (uncovered pixels are zero rows).  Budget NEMO_PRERENDER_MAP_GB (default 48): when all G poses of every pair
  // fit, they are rendered here, once; otherwise the buffer holds a BLOCK of poses (a multiple of 32, the matrix-core
  // pose tile) and nemo_grid_search renders block after block into it.  Large (BIG) meshes render at most
  // BIG_MAX_BLOCKS (pair, pose) jobs per launch (one table slice per workgroup), which bounds their block.
  // Record store (dense grids): per (pair, pose, pixel) three vertex ids + covered flag and three barycentrics, 20 bytes,
  // from which grid_mfma_rec_kernel builds the rendered rows on the fly.  Rendered here in slices of poses through a
  // temporary pix_to_face / bary buffer (campos -> rasterize -> pack).  Ordinary (LDS-table) meshes only.
  // Chosen when the dense renders of all G poses exceed the budget and the records do not (measured, BASELINE config 4:
  // 12 classes x 9720 poses, 0.194 s against 0.455 s for block-after-block rendering), and for grids of 4096 poses or
  // more even when the renders would fit (one class x 9720 poses: 0.141 s and 1.6 GB against 0.198 s and 40 GB; at 144
  // poses the two are within 0.2 ms of each other and the renders stay the default) -- profiles/r03_dense_grid_records.md.
  const double map_budget = ctx->env_prerender_map_gb * 1073741824.0;
  const bool maps_fit = (double)n_pairs * G * stride * ctx->cfg.feat_dim * 4.0 <= map_budget;
  if (n_pairs <= GP_MAX_PAIRS && !ctx->P.big && !ctx->env_grid_no_mfma && (double)n_pairs * G * stride * 20.0 <= map_budget &&
      (ctx->env_grid_records > 0 || (ctx->env_grid_records < 0 && (!maps_fit || G >= 4096)))) {
    const size_t n_rec = (size_t)n_pairs * G * stride;
    const int n_chunks = (int)((stride + GP_PIX - 1) / GP_PIX);
    long gb = (long)((2048.0 * 1048576.0) / ((double)n_pairs * stride * 16.0));   // about 2 GiB of scratch per slice
    if (gb < 1) gb = 1;
    if (gb > G) gb = G;
    const size_t jobs = (size_t)n_pairs * gb;
    hipError_t e = hipMalloc((void **)&ctx->d_pre_vid, n_rec * sizeof(RecVid));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_pre_bary, n_rec * 12);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_mesh, jobs * 4);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_gidx, jobs * 4);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_C, jobs * 12);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_t, jobs * 4);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_R, jobs * 36);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_T, jobs * 12);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_p2f, jobs * stride * 4);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_bary, jobs * stride * 12);
    hipStream_t st = (hipStream_t)stream;
    // the store's pose order: alike silhouettes next to each other (rec_group_kernel), from the lists just rendered
    uint32_t *d_sig = nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_pre_perm, sizeof(int32_t) * (size_t)n_pairs * G);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_pre_slotof, sizeof(int32_t) * (size_t)n_pairs * G);
    if (e == hipSuccess) e = hipMalloc((void **)&d_sig, sizeof(uint32_t) * RG_W * (size_t)n_pairs * G);
    if (e == hipSuccess) {
      if (ctx->env_grid_group && G <= 65536) {
        int B = 8;
        while (((ctx->cfg.win_h + B - 1) / B) * ((ctx->cfg.win_w + B - 1) / B) > RG_W * 32) B *= 2;
        hipLaunchKernelGGL(rec_sig_kernel, dim3((unsigned)((size_t)n_pairs * G)), dim3(256), 0, st, ctx->cfg.win_w, B,
                           (ctx->cfg.win_w + B - 1) / B, stride, (const uint32_t *)ctx->d_pre_list, (const int32_t *)ctx->d_pre_cnt, d_sig);
        hipLaunchKernelGGL(rec_group_kernel, dim3((unsigned)n_pairs), dim3(1024), 0, st, (int)G, (const uint32_t *)d_sig,
                           (const int32_t *)ctx->d_pre_cnt, ctx->d_pre_perm, ctx->d_pre_slotof);
      } else {
        hipLaunchKernelGGL(iota_mod_kernel, dim3((unsigned)(((size_t)n_pairs * G + 255) / 256)), dim3(256), 0, st, (size_t)n_pairs * G,
                           (int)G, ctx->d_pre_perm, ctx->d_pre_slotof);
      }
      e = hipGetLastError();
    }
    for (long g0 = 0; e == hipSuccess && g0 < G; g0 += gb) {
      const int nb = (int)(G - g0 < gb ? G - g0 : gb);
      const size_t nj = (size_t)n_pairs * nb;
      hipLaunchKernelGGL(block_jobs_kernel, dim3((unsigned)((nj + 255) / 256)), dim3(256), 0, st, n_pairs, G, (int)g0, nb,
                         (const int32_t *)ctx->d_pre_pairs, (const int32_t *)(ctx->d_pre_pairs + n_pairs), ctx->d_blk_mesh, ctx->d_blk_gidx);
      hipLaunchKernelGGL(gather_pose_kernel, dim3((unsigned)((nj + 255) / 256)), dim3(256), 0, st, (int)nj,
                         (const int32_t *)ctx->d_blk_gidx, d_grid_C, d_grid_theta, ctx->d_blk_C, ctx->d_blk_t);
      hipLaunchKernelGGL(campos_kernel, dim3((unsigned)((nj + 63) / 64)), dim3(64), 0, st, (const float *)ctx->d_blk_C,
                         (const float *)ctx->d_blk_t, (const float *)nullptr, (int)nj, ctx->d_blk_R, ctx->d_blk_T);
      LAUNCH_PB(raster_kernel, dim3((unsigned)nj), st, ctx->P, ctx->d_meshes, (const int32_t *)ctx->d_blk_mesh,
                (const float *)ctx->d_blk_R, (const float *)ctx->d_blk_T, ctx->d_blk_p2f, ctx->d_blk_bary, (float *)nullptr);
      const size_t npix = nj * stride;
      hipLaunchKernelGGL(pack_rec_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, npix, stride, (int)G, (int)g0, nb,
                         (const MeshDesc *)ctx->d_meshes, (const int32_t *)ctx->d_blk_mesh, (const int32_t *)ctx->d_blk_p2f,
                         (const float *)ctx->d_blk_bary, (const int32_t *)ctx->d_pre_slotof, ctx->d_pre_vid, ctx->d_pre_bary);
      e = hipGetLastError();
    }
    const int n_pb = (int)((G + GR_PR - 1) / GR_PR);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_pre_mask, sizeof(u64) * (size_t)n_pairs * n_pb * n_chunks);
    if (e == hipSuccess) {  // per (pair, pose block, pixel chunk): the pixels any pose of the block covers
      hipLaunchKernelGGL(rec_mask_kernel, dim3((unsigned)((size_t)n_pairs * n_pb * n_chunks)), dim3(64), 0, st, (int)G, n_pb, n_chunks,
                         stride, (const RecVid *)ctx->d_pre_vid, ctx->d_pre_mask);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    free_block_scratch(ctx, false);
    hipFree(d_sig);
    if (e != hipSuccess) {
      hipFree(ctx->d_pre_vid); hipFree(ctx->d_pre_bary); hipFree(ctx->d_pre_mask); hipFree(ctx->d_pre_perm); hipFree(ctx->d_pre_slotof);
      ctx->d_pre_vid = nullptr; ctx->d_pre_bary = nullptr; ctx->d_pre_mask = nullptr; ctx->d_pre_perm = ctx->d_pre_slotof = nullptr;
      return hip_fail(ctx, e, "nemo_grid_prerender: record store");
    }
    return NEMO_OK;
  }
  if (n_pairs <= GP_MAX_PAIRS && !ctx->env_grid_no_mfma) {
    const double budget = ctx->env_prerender_map_gb * 1073741824.0;
    const size_t D = (size_t)ctx->cfg.feat_dim;
    const double per_pose = (double)n_pairs * stride * D * 4.0;  // one pose of every pair
    long block = (long)(budget / per_pose);
    if (ctx->P.big && block > (long)(BIG_MAX_BLOCKS / (unsigned)n_pairs)) block = (long)(BIG_MAX_BLOCKS / (unsigned)n_pairs);
    if (block >= G) block = G; else block = (block / 32) * 32;
    if (ctx->P.big && block > 0 && reserve_big_scratch(ctx, (size_t)n_pairs * block) != NEMO_OK) block = 0;
    if (block >= 32 || (block == G && block > 0)) {
      const size_t jobs = (size_t)n_pairs * block;
      const int n_chunks = (int)((stride + GP_PIX - 1) / GP_PIX);
      hipError_t e = hipMalloc(&ctx->d_blk_mesh, jobs * 4);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_gidx, jobs * 4);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_C, jobs * 12);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_t, jobs * 4);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_R, jobs * 36);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_T, jobs * 12);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_p2f, jobs * stride * 4);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_bary, jobs * stride * 12);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_blk_live, (size_t)n_pairs * n_chunks);
      if (e == hipSuccess) e = hipMalloc(&ctx->d_pre_maps, (size_t)(per_pose * (double)block));
      if (e != hipSuccess) {
        free_block_scratch(ctx, false);
        hipFree(ctx->d_pre_maps);
        ctx->d_pre_maps = nullptr;
        return hip_fail(ctx, e, "nemo_grid_prerender: dense maps");
      }
      ctx->pre_block = (int)block;
      if (block == G) {  // resident: render now, then drop the render scratch (the chunk flags stay)
        int rc_ = render_pose_block(ctx, d_grid_C, d_grid_theta, G, 0, G, (hipStream_t)stream);
        hipError_t se = hipStreamSynchronize((hipStream_t)stream);
        free_block_scratch(ctx, true);
        if (rc_ != NEMO_OK || se != hipSuccess) {
          hipFree(ctx->d_pre_maps);
          ctx->d_pre_maps = nullptr;
          return rc_ != NEMO_OK ? rc_ : hip_fail(ctx, se, "nemo_grid_prerender: render maps");
        }
      }
    }
  }
  return NEMO_OK;
}

int nemo_solve_last_timing(nemo_ctx *ctx, float *table_ms, float *loop_ms) {
  if (!ctx || !table_ms || !loop_ms) return fail(ctx, NEMO_ERR_ARG, "nemo_solve_last_timing: null argument");
  if (!ctx->ev_solve_valid) return fail(ctx, NEMO_ERR_STATE, "nemo_solve_last_timing: the last solve did not use tables");
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev_solve[2]));
  HIP_TRY(ctx, hipEventElapsedTime(table_ms, ctx->ev_solve[0], ctx->ev_solve[1]));
  HIP_TRY(ctx, hipEventElapsedTime(loop_ms, ctx->ev_solve[1], ctx->ev_solve[2]));
  return NEMO_OK;
}

int64_t nemo_solve_table_capacity(const nemo_ctx *ctx) {
  if (!ctx || !ctx->d_tab || ctx->P.big || ctx->vmax_seen <= 0) return 0;
  return (int64_t)(ctx->tab_cap / table_stride(ctx));
}

int nemo_grid_store_kind(const nemo_ctx *ctx) {
  if (!ctx) return 0;
  if (ctx->d_pre_vid) return 3;
  if (ctx->d_pre_maps) return ctx->pre_block == ctx->pre_G ? 1 : 2;
  return 0;
}

// Shared by nemo_solve / nemo_solve6 / nemo_solve_static: launch order (longest job first), then the fused solve.
// `ws` = int32 workspace of 2 * ws_cap entries (cost[ws_cap], order[ws_cap]) -- the context's reservation or the
// static device array; never allocated here.
static int launch_solve(nemo_ctx *ctx, const char *who, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx,
                        const float *d_feat, const float *d_clu, const float *d_C0, const float *d_theta0,
                        const float *d_trans0, int ndof, float *d_pose, float *d_loss, float *d_trace, int32_t *d_status,
                        void *stream, int32_t *ws, size_t ws_cap) {
  int rc = check_ready(ctx, who, N);
  if (rc) return rc > 0 ? NEMO_OK : rc;
  if (ctx->adam_steps <= 0) return fail(ctx, NEMO_ERR_STATE, "%s: call nemo_set_schedule first", who);
  if (!d_feat || !d_C0 || !d_theta0 || !d_pose) return fail(ctx, NEMO_ERR_ARG, "%s: null buffer", who);
  if (ctx->cfg.clutter && !d_clu) return fail(ctx, NEMO_ERR_ARG, "%s: clutter loss needs d_clu", who);
  if (ctx->P.big && (unsigned)N > BIG_MAX_BLOCKS)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "%s: %d jobs in one call with a large mesh (limit %u)", who, N, BIG_MAX_BLOCKS);
  // start stagger (NEMO_SOLVE_STAGGER, experiments) in units of (slot x 8128-cycle sleeps)
  const int stagger = ctx->env_solve_stagger;
  // launch order: only worth it when some CU gets more than one job (NEMO_SOLVE_ORDER=0 turns it off)
  const int32_t *d_order = nullptr;
  if (N > 64 && ctx->env_solve_order) {
    if ((size_t)N > ws_cap)
      return fail(ctx, NEMO_ERR_STATE, "%s: launch-order workspace holds %zu jobs, this call has %d: call "
                  "nemo_workspace_reserve(ctx, >= %d, ...) first", who, ws_cap, N, N);
    int32_t *d_cost = ws, *d_ord = ws + ws_cap;
    if ((rc = ensure_big_scratch(ctx, (size_t)N)) != NEMO_OK) return rc;
    LAUNCH_PB(solve_cost_kernel, dim3(N), (hipStream_t)stream, ctx->P, ctx->d_meshes, d_mesh_idx, d_C0, d_theta0,
              ndof == 6 ? d_trans0 : (const float *)nullptr, d_cost);
    hipLaunchKernelGGL(solve_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (int)N, (const int32_t *)d_cost, d_ord);
    HIP_TRY(ctx, hipGetLastError());
    if (d_mesh_idx && ctx->n_meshes > 1 && ctx->env_solve_affinity) {
      const int windows = (N + 63) / 64;
      hipLaunchKernelGGL(solve_affinity_kernel, dim3((unsigned)((windows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (int)N,
                         ctx->n_meshes, d_mesh_idx, d_ord);
    }
    d_order = d_ord;
  }
  // Tables of feature-row x bank-row dot products, when the reserved workspace holds them: the contraction is done once
  // here instead of once per Adam step.  Same poses up to the summation order of a dot product; never for
  // large (BIG) meshes.  A call with more jobs than the tables hold goes through in slices of the launch order (which
  // then exists: such a call has more than 64 jobs): tables for a slice, then its solve, on the one stream.
  const size_t hw = (size_t)ctx->cfg.win_h * ctx->cfg.win_w, tab_stride = table_stride(ctx);
  // (the tables form packs a covered pixel's window row and column into 8 bits each)
  const size_t tab_jobs = (!ctx->P.big && ctx->d_tab && tab_stride > 0 && ctx->cfg.win_h <= 256 && ctx->cfg.win_w <= 256) ? ctx->tab_cap / tab_stride : 0;
  if (tab_jobs >= (size_t)N || (tab_jobs >= 256 && d_order)) {
    const int D = ctx->cfg.feat_dim, nt = D == 128 ? CkCfg<128>::NT : CkCfg<256>::NT;
    const size_t lds = (size_t)2 * CK_ROWS * (D + 4) * 4;
    // lazy tables (NEMO_SOLVE_LAZY, default on): the block bitmap must fit the per-face record area of the LDS plan
    // behind the 64 sort bins
    // (one 8-byte word of vertex-chunk bits per pixel tile: at most 64 chunks of 32 = 2048 vertices)
    const uint32_t tab_ld = table_ld((uint32_t)ctx->cfg.win_h, (uint32_t)ctx->cfg.win_w);
    const size_t map_bytes = 256 + (size_t)(tab_ld / 32) * 8;
    const int lazy = (ctx->env_solve_lazy && table_rows((uint32_t)ctx->vmax_seen) <= 2048 && map_bytes <= (size_t)ctx->P.fmax * 16) ? 1 : 0;
    // workgroups per job: a workgroup's prologue (first bank chunk, first feature rows) is exposed, so few, long
    // workgroups -- NEMO_TABLE_WPJ (default 2) -- unless the call has too few jobs to fill the chip with them
    unsigned wpj = (unsigned)ctx->env_table_wpj;
    while ((size_t)wpj * (size_t)std::min<size_t>(tab_jobs, (size_t)N) < 512 && wpj < 8) wpj *= 2;
    wpj = (unsigned)std::min<size_t>(wpj, (tab_ld / 32 + (nt / 64) - 1) / (nt / 64));
    for (size_t k0 = 0; k0 < (size_t)N; k0 += tab_jobs) {
      const unsigned nk = (unsigned)std::min<size_t>(tab_jobs, (size_t)N - k0);
      const int32_t *ord = d_order ? d_order + k0 : nullptr;   // (no order: a single slice, slot y = job y)
      if (k0 == 0) hipEventRecord(ctx->ev_solve[0], (hipStream_t)stream);
      if (lazy) { /* the solve kernel fills its tables itself, block by block, as its steps ask for them */ }
      else if (D == 128)
        hipLaunchKernelGGL((bank_table_kernel<128>), dim3(wpj, nk), dim3(nt), lds, (hipStream_t)stream, d_feat, (int)hw,
                           (int)tab_ld, ctx->cfg.win_w, d_img_idx, d_mesh_idx, (const MeshDesc *)ctx->d_meshes, ctx->d_tab, tab_stride, ord, (int)N);
      else
        hipLaunchKernelGGL((bank_table_kernel<256>), dim3(wpj, nk), dim3(nt), lds, (hipStream_t)stream, d_feat, (int)hw,
                           (int)tab_ld, ctx->cfg.win_w, d_img_idx, d_mesh_idx, (const MeshDesc *)ctx->d_meshes, ctx->d_tab, tab_stride, ord, (int)N);
      if (k0 == 0) hipEventRecord(ctx->ev_solve[1], (hipStream_t)stream);
#define SOLVE_TAB(DD, PP, LZ)                                                                                               \
  hipLaunchKernelGGL((solve_kernel<DD, PP, false, true, LZ>), dim3(nk), dim3(NT), ctx->lds, (hipStream_t)stream, ctx->P,    \
                     ctx->d_meshes, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, d_trans0, ndof, ctx->d_adam,       \
                     ctx->adam_steps, ctx->beta1, ctx->beta2, ctx->eps, d_pose, d_loss, d_trace, ctx->d_counter, stagger,   \
                     ord, (int)N, d_status, (const float *)ctx->d_tab, tab_stride)
#define SOLVE_TAB_L(DD, PP) do { if (lazy) SOLVE_TAB(DD, PP, true); else SOLVE_TAB(DD, PP, false); } while (0)
      if (D == 128) { if (ctx->P.persp) SOLVE_TAB_L(128, true); else SOLVE_TAB_L(128, false); }
      else          { if (ctx->P.persp) SOLVE_TAB_L(256, true); else SOLVE_TAB_L(256, false); }
#undef SOLVE_TAB_L
#undef SOLVE_TAB
      if (k0 == 0) { hipEventRecord(ctx->ev_solve[2], (hipStream_t)stream); ctx->ev_solve_valid = 1; }
    }
    ctx->last_solve_mode = lazy ? 2 : 1;
    HIP_TRY(ctx, hipGetLastError());
    return NEMO_OK;
  }
  ctx->ev_solve_valid = 0;
  ctx->last_solve_mode = 0;
  DISPATCH_DP(solve_kernel, dim3(N), ctx->P, ctx->d_meshes, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, d_trans0,
              ndof, ctx->d_adam, ctx->adam_steps, ctx->beta1, ctx->beta2, ctx->eps, d_pose, d_loss, d_trace, ctx->d_counter,
              stagger, d_order, (int)N, d_status, (const float *)nullptr, (size_t)0);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

int nemo_solve(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx, const float *d_feat,
               const float *d_clu, const float *d_C0, const float *d_theta0, float *d_pose, float *d_loss,
               float *d_trace, int32_t *d_status, void *stream) {
  return launch_solve(ctx, "nemo_solve", N, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, nullptr, 4, d_pose,
                      d_loss, d_trace, d_status, stream, ctx ? ctx->d_so : nullptr, ctx ? ctx->so_cap : 0);
}

int nemo_solve_static(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx, const float *d_feat,
                      const float *d_clu, const float *d_C0, const float *d_theta0, float *d_pose, float *d_loss,
                      float *d_trace, int32_t *d_status, void *stream) {
  if (!ctx) return fail(nullptr, NEMO_ERR_ARG, "nemo_solve_static: null ctx");
  if (N > STATIC_MAX_JOBS)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_solve_static: %d jobs, the static workspace holds %d", N, STATIC_MAX_JOBS);
  if (ctx->P.big)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_solve_static: large meshes need per-workgroup table scratch "
                "(use nemo_workspace_reserve + nemo_solve)");
  int32_t *ws = (ctx->device >= 0 && ctx->device < MAX_DEVICES) ? g_static_ws[ctx->device] : nullptr;
  if (!ws) return fail(ctx, NEMO_ERR_STATE, "nemo_solve_static: no static workspace on device %d", ctx->device);
  return launch_solve(ctx, "nemo_solve_static", N, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, nullptr, 4, d_pose,
                      d_loss, d_trace, d_status, stream, ws, (size_t)STATIC_MAX_JOBS);
}

int nemo_solve6(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx, const float *d_feat,
                const float *d_clu, const float *d_C0, const float *d_theta0, const float *d_trans0, float *d_pose6,
                float *d_loss, float *d_trace, int32_t *d_status, void *stream) {
  return launch_solve(ctx, "nemo_solve6", N, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, d_trans0, 6, d_pose6,
                      d_loss, d_trace, d_status, stream, ctx ? ctx->d_so : nullptr, ctx ? ctx->so_cap : 0);
}

int nemo_interpolate_backward(nemo_ctx *ctx, const int32_t *d_mesh_idx, const float *d_R, const float *d_T,
                              const float *d_grad_out, int32_t N, float *d_grad_R, float *d_grad_T, void *stream) {
  int rc = check_ready(ctx, "nemo_interpolate_backward", N);
  if (rc) return rc > 0 ? NEMO_OK : rc;
  if (!d_R || !d_T || !d_grad_out || !d_grad_R || !d_grad_T)
    return fail(ctx, NEMO_ERR_ARG, "nemo_interpolate_backward: null buffer");
  if (N == 0) return NEMO_OK;
  DISPATCH_DP(interp_bwd_kernel, dim3(N), ctx->P, ctx->d_meshes, d_mesh_idx, d_R, d_T, d_grad_out, d_grad_R, d_grad_T);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

int nemo_clutter_score(nemo_ctx *ctx, const float *d_feat, int32_t n_img, const float *d_cbank, int32_t K,
                       float *d_out, void *stream) {
  if (!ctx) return fail(nullptr, NEMO_ERR_ARG, "nemo_clutter_score: null ctx");
  if (K <= 0 || n_img < 0) return fail(ctx, NEMO_ERR_ARG, "nemo_clutter_score: bad argument");
  if (n_img == 0) return NEMO_OK;
  if (!d_feat || !d_cbank || !d_out) return fail(ctx, NEMO_ERR_ARG, "nemo_clutter_score: null buffer");
  const size_t npix = (size_t)n_img * ctx->P.h * ctx->P.w;
  const size_t blocks = (npix + 127) / 128;
  if (blocks > 0x7fffffffULL) return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_clutter_score: too many pixels for one launch");
  if (K <= 8) {  // small bank: streaming dot products, no matrix-core tile
    if (ctx->cfg.feat_dim == 128) {
      const size_t nb = (npix + 4 * 32 - 1) / (4 * 32);  // 4 waves x (4 pixels x 8 iterations)
      hipLaunchKernelGGL((clutter_dot_kernel<128>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, d_feat, npix, d_cbank, K, d_out);
    } else {
      const size_t nb = (npix + 4 * 16 - 1) / (4 * 16);  // 4 waves x (2 pixels x 8 iterations)
      if (nb > 0x7fffffffULL) return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_clutter_score: too many pixels for one launch");
      hipLaunchKernelGGL((clutter_dot_kernel<256>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, d_feat, npix, d_cbank, K, d_out);
    }
    HIP_TRY(ctx, hipGetLastError());
    return NEMO_OK;
  }
  {
    // persistent workgroups, one per CU; LDS = two chunks of CK_ROWS padded bank rows (the kernels were opted in to
    // this much dynamic LDS by nemo_ctx_create)
    const int D = ctx->cfg.feat_dim, nt = D == 128 ? CkCfg<128>::NT : CkCfg<256>::NT;
    const size_t tiles = (npix + 31) / 32, per = (size_t)(nt / 64);
    const unsigned nb = (unsigned)std::min<size_t>((tiles + per - 1) / per, (size_t)ctx->n_cu);
    const size_t lds = 2 * (size_t)CK_ROWS * (D + 4) * sizeof(float);
    if (D == 128)
      hipLaunchKernelGGL((clutter_kernel<128>), dim3(nb), dim3(nt), lds, (hipStream_t)stream, d_feat, npix, d_cbank, K, d_out);
    else
      hipLaunchKernelGGL((clutter_kernel<256>), dim3(nb), dim3(nt), lds, (hipStream_t)stream, d_feat, npix, d_cbank, K, d_out);
  }
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

int nemo_feature_head(nemo_ctx *ctx, const float *d_x, int32_t N, int32_t C, const float *d_weight, const float *d_cbank,
                      int32_t K, float *d_feat, float *d_clu, void *stream) {
  if (!ctx) return fail(nullptr, NEMO_ERR_ARG, "nemo_feature_head: null ctx");
  if (N < 0 || C <= 0) return fail(ctx, NEMO_ERR_ARG, "nemo_feature_head: bad argument");
  if (N == 0) return NEMO_OK;
  if (!d_x || !d_weight || !d_feat) return fail(ctx, NEMO_ERR_ARG, "nemo_feature_head: null buffer");
  if (d_clu && (!d_cbank || K <= 0)) return fail(ctx, NEMO_ERR_ARG, "nemo_feature_head: clutter output needs a clutter bank");
  const int D = ctx->cfg.feat_dim, hw = ctx->cfg.win_h * ctx->cfg.win_w;
  // the weights are LDS-resident, padded with zeros to whole chunks of 2 * FH_KS channels
  const size_t lds = (size_t)D * 2 * FH_KS * ((C / 2 + FH_KS - 1) / FH_KS) * sizeof(float);
  if ((C & 1) || lds > 128 * 1024)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_feature_head: C = %d (even; D * C * 4 <= 128 KiB after rounding C up to a multiple of %d: the weights are LDS-resident)", C, 2 * FH_KS);
  const int n_cu = ctx->n_cu;  // cached by nemo_ctx_create, which also opted the kernel in to 128 KiB of LDS
  const long long tiles = (long long)N * ((hw + 31) / 32);
  const unsigned blocks = (unsigned)(tiles / FH_NW < n_cu ? (tiles + FH_NW - 1) / FH_NW : n_cu);
  if (D == 128) {
    hipLaunchKernelGGL((feature_head_kernel<128>), dim3(blocks), dim3(FH_NT), lds, (hipStream_t)stream, d_x, N, C, hw, d_weight,
                       d_cbank, K, d_feat, d_clu);
  } else {
    hipLaunchKernelGGL((feature_head_kernel<256>), dim3(blocks), dim3(FH_NT), lds, (hipStream_t)stream, d_x, N, C, hw, d_weight,
                       d_cbank, K, d_feat, d_clu);
  }
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

#ifdef NEMO_STAMPS
// diagnostic build only: read (and clear) the per-phase cycle sums
int nemo_debug_stamps(unsigned long long *out16) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

#ifdef NEMO_STAMPS
// diagnostic build only: read (and clear) the score-loop busy time summed per wave index
int nemo_debug_wave_busy(unsigned long long *out16) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wave_busy), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_busy), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

int64_t nemo_bytes_per_covered_pixel(const nemo_ctx *ctx) { return ctx ? 4LL * ctx->cfg.feat_dim + 4 : 0; }

int nemo_counter_reset(nemo_ctx *ctx, void *stream) {
  if (!ctx) return fail(nullptr, NEMO_ERR_ARG, "nemo_counter_reset: null ctx");
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_counter, 0, 2 * sizeof(u64), (hipStream_t)stream));
  return NEMO_OK;
}

int nemo_counter_read(nemo_ctx *ctx, uint64_t *out) {
  if (!ctx || !out) return fail(ctx, NEMO_ERR_ARG, "nemo_counter_read: bad argument");
  HIP_TRY(ctx, hipDeviceSynchronize());
  u64 v = 0;
  HIP_TRY(ctx, hipMemcpy(&v, ctx->d_counter, sizeof(u64), hipMemcpyDeviceToHost));
  *out = v;
  return NEMO_OK;
}

int nemo_solve_table_stats(nemo_ctx *ctx, int32_t *mode, uint64_t *blocks) {
  if (!ctx || !mode || !blocks) return fail(ctx, NEMO_ERR_ARG, "nemo_solve_table_stats: bad argument");
  HIP_TRY(ctx, hipDeviceSynchronize());
  u64 v = 0;
  HIP_TRY(ctx, hipMemcpy(&v, ctx->d_counter + 1, sizeof(u64), hipMemcpyDeviceToHost));
  *mode = ctx->last_solve_mode;
  *blocks = v;
  return NEMO_OK;
}

}  // extern "C"
