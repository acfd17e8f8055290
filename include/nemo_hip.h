/*
 * nemo_hip.h -- C ABI of libnemo_hip.so, the MI355X (gfx950) render-and-compare
 * pose solver that stands in for the PyTorch3D rasterizer + torch autograd +
 * torch.optim.Adam inside the reference's MeshPoseSolveAll inference path.
 *
 * Conventions
 *  - every pointer named d_* is a DEVICE pointer owned by the caller (for the
 *    Python host: tensor.data_ptr() of a contiguous torch tensor), 16-byte aligned;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - every call returns 0 on success or a negative code (NEMO_ERR_*, or
 *    -(1000 + hipError_t)); nemo_last_error() gives the text.  Nothing aborts;
 *  - no entry point below the "setup" group allocates, frees, synchronises or reads the environment: the
 *    workspaces they need are sized by nemo_workspace_reserve (set-up group), and a hot call whose batch does
 *    not fit the reservation returns NEMO_ERR_STATE instead of growing it.  The workspaces belong to the
 *    context, so a context serves ONE stream at a time (use one context per concurrent stream);
 *  - meshes hold PRE-PROCESSED vertices (x,y,z)->(x,z,-y) as the reference's
 *    pre_process_mesh_pascal does (code/lib/MeshUtils.py:115-117);
 *  - feature maps are PIXEL-MAJOR f32 [n_img][h][w][D] (the reference's
 *    [D][h][w] maps, code/MeshPoseSolveAll.py:218, transposed once on upload);
 *  - a "job" is one (image, mesh) pair; d_img_idx / d_mesh_idx (int32[N]) select
 *    them, NULL meaning identity / mesh 0.
 *
 * file:line citations are into /root/reference (Angtian/NeMo).
 */
#ifndef NEMO_HIP_H
#define NEMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEMO_OK 0
#define NEMO_ERR_ARG (-1)         /* bad argument / shape */
#define NEMO_ERR_UNSUPPORTED (-2) /* size outside what the kernels were built for */
#define NEMO_ERR_STATE (-3)       /* call order (e.g. schedule not set) */
#define NEMO_ERR_HIP_BASE (-1000) /* -(1000 + hipError_t) */

typedef struct nemo_ctx nemo_ctx;

/* Rasterizer / camera / loss configuration: RasterizationSettings(image_size=S,
 * blur_radius=0, faces_per_pixel=1, bin_size=0) + OpenGLPerspectiveCameras(fov)
 * + center_crop_fun(map_shape, (S,S)) (code/MeshPoseSolveAll.py:180-191,204). */
typedef struct {
  int32_t raster_size;         /* S = max(H,W)//8 (:171) */
  int32_t win_h, win_w;        /* feature-map shape h, w (:191) */
  int32_t win_r0, win_c0;      /* crop origin; <0 = centred as lib/MeshUtils.py:148-150 */
  int32_t feat_dim;            /* D: 128 or 256 */
  float ndc_scale;             /* 1/tan(fov/2) of OpenGLPerspectiveCameras (:180) */
  int32_t perspective_correct; /* PyTorch3D 0.2.0: 0; FoV cameras in >=0.4: 1 */
  int32_t clutter;             /* 1: loss_fun (:58-59); 0: loss_fun0 / turn_off_clutter (:50-56) */
  double k_eps;                /* PyTorch3D kEpsilon (1e-8 current sources) */
} nemo_config;

/* ---- setup (may allocate / synchronise) ---------------------------------- */

/* Replaces building cameras + RasterizationSettings + MeshRasterizer
 * (code/MeshPoseSolveAll.py:180-190). */
int nemo_ctx_create(const nemo_config *cfg, nemo_ctx **out);
int nemo_ctx_destroy(nemo_ctx *ctx);
const char *nemo_last_error(const nemo_ctx *ctx); /* ctx may be NULL (global slot) */

/* Replaces MeshInterpolateModule.__init__ / update_memory (lib/MeshUtils.py:154-193):
 * registers one mesh (device pointers stay owned by the caller and must outlive
 * the context); the vertex bank f32[V][D] is indexed through faces instead of
 * materialising face_memory.  The faces are read once here (synchronous copy) to choose, per face, the
 * order in which the score loop keeps its three bank rows in registers (a pure scheduling table of F bytes
 * owned by the context; results do not depend on it).  Returns the mesh id (>= 0) or an error. */
int nemo_mesh_register(nemo_ctx *ctx, const float *d_verts, int32_t V, const int32_t *d_faces,
                       int32_t F, const float *d_bank);

/* Adam / ExponentialLR schedule (code/MeshPoseSolveAll.py:230-231,249-250;
 * torch.optim.Adam): host table f32[steps][2] = {lr_t/(1-b1^t), sqrt(1-b2^t)},
 * prepared in double by the caller exactly like torch's Python side. */
int nemo_set_schedule(nemo_ctx *ctx, const float *h_tab, int32_t steps, float beta1, float beta2,
                      float eps);

/* Sizes every workspace of the hot entry points once (SURVEY 8b "no allocation in hot calls"; the reference
 * allocates freely per call: every torch op at code/MeshPoseSolveAll.py:236-245 creates its outputs).
 * max_jobs = largest N of any later call; max_grid_poses = largest G of nemo_grid_search (0: none).
 * Covers: the solve's launch-order arrays (int32[2 max_jobs]), the grid search's job grouping and per-chunk
 * partial sums (ceil(h w / 64) * max_jobs * max_grid_poses floats), and, for meshes whose tables do not fit the
 * LDS, one table slice per workgroup (at most 2048).  Grows only; may be called again (between hot calls).
 * Environment switches (NEMO_GRID_NO_MFMA, NEMO_SOLVE_ORDER, NEMO_SOLVE_AFFINITY, NEMO_SLOT_ORDER,
 * NEMO_PRERENDER_MAP_GB ...) are read once, in nemo_ctx_create. */
int nemo_workspace_reserve(nemo_ctx *ctx, int32_t max_jobs, int32_t max_grid_poses);

/* ---- hot path (asynchronous on `stream`, no allocation) ------------------- */

/* campos_to_R_T (lib/MeshUtils.py:104-111): look_at_rotation(C, at=0, up=+y) @
 * rotation_theta(theta), T = -R^T C (+ extra_trans, may be NULL).
 * d_C f32[N][3], d_theta f32[N] -> d_R f32[N][9] (row-major), d_T f32[N][3]. */
int nemo_campos_to_R_T(nemo_ctx *ctx, const float *d_C, const float *d_theta, const float *d_extra,
                       int32_t N, float *d_R, float *d_T, void *stream);

/* MeshRasterizer.forward -> rasterize_meshes naive, K=1 (lib/MeshUtils.py:83-85),
 * already cropped to the window (:226).  d_pix_to_face i32[N][h][w] (-1 empty,
 * face ids LOCAL to the job's mesh), d_bary f32[N][h][w][3] (-1 empty),
 * d_zbuf f32[N][h][w] or NULL. */
int nemo_rasterize(nemo_ctx *ctx, const int32_t *d_mesh_idx, const float *d_R, const float *d_T,
                   int32_t N, int32_t *d_pix_to_face, float *d_bary, float *d_zbuf, void *stream);

/* interpolate_face_attributes (lib/MeshUtils.py:129): d_out f32[N][h][w][D],
 * zero where pix_to_face == -1. */
int nemo_interpolate(nemo_ctx *ctx, const int32_t *d_mesh_idx, const int32_t *d_pix_to_face,
                     const float *d_bary, int32_t N, float *d_out, void *stream);

/* Backward of rasterize+interpolate for the autograd-compatible module path
 * (loss.backward() at code/MeshPoseSolveAll.py:240 through lib/MeshUtils.py:122-133):
 * d_grad_out f32[N][h][w][D] -> d_grad_R f32[N][9], d_grad_T f32[N][3].  Coverage is
 * recomputed from (R, T): deterministic, hence identical to nemo_rasterize's. */
int nemo_interpolate_backward(nemo_ctx *ctx, const int32_t *d_mesh_idx, const float *d_R,
                              const float *d_T, const float *d_grad_out, int32_t N,
                              float *d_grad_R, float *d_grad_T, void *stream);

/* Clutter score (code/MeshPoseSolveAll.py:207,220; K-vector variant
 * lib/MeshUtils.py:273-275): d_out[n][p] = max_k <d_cbank[k], feat[n][p]>,
 * d_cbank f32[K][D]; the pixels x clutter-bank contraction runs on MFMA. */
int nemo_clutter_score(nemo_ctx *ctx, const float *d_feat, int32_t n_img, const float *d_cbank,
                       int32_t K, float *d_out, void *stream);

/* Backbone tail + clutter score in one pass: NetE2E.forward_test after the backbone
 * (code/models/KeypointRepresentationNet.py:217-223) = 1x1 convolution with out_layer.weight f32[D][C] (no bias)
 * + F.normalize over channels (eps 1e-12), fused with the clutter score of code/MeshPoseSolveAll.py:220.
 * d_x f32[N][C][h*w] (the backbone's NCHW output at the feature-map resolution); d_feat f32[N][h][w][D] is written
 * pixel-major, ready for the calls below; d_clu f32[N][h][w] = max_k <d_cbank[k], feat> or NULL.  The contraction
 * runs on the fp32 matrix cores; D*C*4 bytes of weights must fit 128 KiB of LDS (C = 256 at D = 128). */
int nemo_feature_head(nemo_ctx *ctx, const float *d_x, int32_t N, int32_t C, const float *d_weight,
                      const float *d_cbank, int32_t K, float *d_feat, float *d_clu, void *stream);

/* One render-and-compare evaluation per job at given poses: object score
 * (:237), loss (:238), and dL/dC, dL/dtheta as loss.backward() (:240) would
 * leave them in C.grad / theta.grad.  d_loss f32[N], d_grad f32[N][4]
 * (gCx,gCy,gCz,gtheta; may be NULL), d_ncov i32[N] (may be NULL). */
int nemo_score_fwd_bwd(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx,
                       const int32_t *d_mesh_idx, const float *d_feat, const float *d_clu,
                       const float *d_C, const float *d_theta, float *d_loss, float *d_grad,
                       int32_t *d_ncov, void *stream);

/* get_pre_render_samples + get_init_pos_rendered (code/MeshPoseSolveAll.py:62-100):
 * loss of every grid pose for every job, first minimum wins.
 * d_grid_C f32[n_grids][G][3], d_grid_theta f32[n_grids][G]; d_grid_idx i32[N]
 * picks the grid per job (NULL = grid 0; the reference uses one distance per
 * category, :150-151,169).  Outputs: d_C0 f32[N][3], d_theta0 f32[N],
 * d_best i32[N] (may be NULL), d_losses f32[N][G] scratch/output (required). */
int nemo_grid_search(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx,
                     const int32_t *d_grid_idx, const float *d_feat, const float *d_clu,
                     const float *d_grid_C, const float *d_grid_theta, int32_t G, float *d_losses,
                     float *d_C0, float *d_theta0, int32_t *d_best, void *stream);

/* get_pre_render_samples (code/MeshPoseSolveAll.py:62-82): the reference renders every grid pose once
 * per category and compares each image against the stored renders.  Here the image-independent part
 * of a grid pose -- which pixels a mesh covers and with which face -- is rendered once per listed
 * (mesh, grid) pair and kept by the context; nemo_grid_search then skips projection-independent work
 * for jobs whose pair is stored (others are rendered on the fly; results are the same either way).
 * The store is tied to the two device pointers and G and is dropped by n_pairs == 0 or the next call.
 * h_pair_mesh / h_pair_grid are HOST arrays.  Setup call: allocates.
 * What is kept (all three give the same losses; the matrix-core forms bit for bit the same as each other):
 *   - always: per (pair, pose) the list of covered pixels with their faces, n_pairs*G*win_h*win_w*4 bytes;
 *   - the operand of the matrix-core grid search, within NEMO_PRERENDER_MAP_GB (default 48) of device memory:
 *     1 the dense renders f32[pair][G][win_h*win_w][D] when they fit and G < 4096, rendered here once;
 *     3 else, for meshes whose tables fit the LDS, a record per (pair, pose, pixel) -- three 16-bit vertex ids, a covered
 *       flag, three barycentrics, 20 bytes -- from which nemo_grid_search builds the rendered rows on the fly in LDS
 *       (BASELINE config 4, 12 classes x 9720 poses: 19 GB of records instead of 478 GB of renders).  The store keeps
 *       poses with alike silhouettes next to each other (tiles of 32 poses whose pixels nobody covers are skipped);
 *       losses are reported in grid order and do not depend on the store's order;
 *     2 else a buffer for a block of poses (a multiple of 32) that nemo_grid_search re-renders block after block.
 *   NEMO_GRID_RECORDS=1 in the environment (at nemo_ctx_create) prefers 3 whenever the records fit, =0 never uses it;
 *   NEMO_GRID_GROUP=0 keeps the record store in grid order; NEMO_GRID_NO_MFMA=1 keeps the list path only.
 *   nemo_grid_store_kind reports which form the context holds. */
int nemo_grid_prerender(nemo_ctx *ctx, const float *d_grid_C, const float *d_grid_theta, int32_t n_grids,
                        int32_t G, const int32_t *h_pair_mesh, const int32_t *h_pair_grid, int32_t n_pairs,
                        void *stream);
/* 0 = no matrix-core operand stored (lists only, or nothing pre-rendered), 1 = resident dense renders, 2 = pose-block
 * buffer, 3 = record store.  A query of this library's own state; the reference has no counterpart. */
int nemo_grid_store_kind(const nemo_ctx *ctx);

/* The 300-step solve (code/MeshPoseSolveAll.py:227-250): render, score, loss,
 * backward and Adam on (Cx,Cy,Cz,theta) fused in one persistent launch, one
 * workgroup per job; schedule from nemo_set_schedule.  d_pose f32[N][4]
 * = final (Cx,Cy,Cz,theta), d_loss f32[N] = loss at the last evaluated iterate,
 * d_trace f32[N][steps][9] = (C, theta, loss, gC, gtheta) at the start of every
 * step, or NULL.  d_status i32[N] or NULL: 1 = the job's final pose is finite, 0 = it is not (a non-finite start
 * pose stays non-finite; the launch completes and the other jobs are unaffected) -- the per-image flag behind the
 * reference's "NaN / Inf -> error pi/2" rule (code/MeshPoseSolveAll.py:262-263).
 * Scheduling (results do not depend on it): for N > 64 the jobs are started longest first -- one
 * rasterisation per job counts the pixels covered at (C0, theta0), a device counting sort orders the job
 * ids, workgroup b runs job order[b]; the int32[2N] workspace comes from nemo_workspace_reserve
 * (NEMO_ERR_STATE if N exceeds it).
 * Inside windows of 64 ranks of that order the jobs are dealt to the XCDs (workgroup b runs on XCD b % 8) by mesh,
 * so that an XCD's L2 keeps few vertex banks.  NEMO_SOLVE_ORDER=0 in the environment (at nemo_ctx_create) launches
 * in index order, NEMO_SOLVE_AFFINITY=0 skips the XCD pass. */
int nemo_solve(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx,
               const float *d_feat, const float *d_clu, const float *d_C0, const float *d_theta0,
               float *d_pose, float *d_loss, float *d_trace, int32_t *d_status, void *stream);

/* How nemo_solve / nemo_solve6 / nemo_solve_static spend their time -- the two forms give the same poses up to the
 * summation order inside a dot product (both are checked against the oracle to the same 1e-4):
 *  - TABLES (default when they fit): the object score needs, per covered pixel, the dot products of the pixel's feature
 *    row with the three bank rows of its face.  None of that depends on the pose, so the call first computes ALL
 *    feature-row x bank-row products of every job on the matrix cores (f32 [job][V][columns], 36 MB per job at
 *    V = 1088 and an 80 x 100 window; the columns of a row are 2-D patches of the window, 32 floats per 4-row x 8-column
 *    patch, so that the pixels of a face share cache lines in both image directions) and its Adam steps read three floats per covered pixel.  The tables live in the
 *    workspace sized by nemo_workspace_reserve(ctx, max_jobs, ...): max_jobs x (vertices of the largest registered mesh,
 *    rounded up to 64) x (ceil(win_h / 4) * ceil(win_w / 8) * 32) x 4 bytes, allocated only when that is within
 *    NEMO_SOLVE_TABLE_GB (environment, read at nemo_ctx_create; default 64, 0 = never); with room for at least 256 jobs
 *    a larger call goes through in slices.  Never for meshes whose tables do not fit the LDS, nor for windows of more
 *    than 256 rows or columns (a covered pixel's row and column are packed into 8 bits each).  NEMO_TABLE_WPJ
 *    (default 2): workgroups per job of the table kernel.
 *  - STREAMING (otherwise): every Adam step reads the covered pixels' feature rows (4*D + 4 bytes per covered pixel and
 *    step) and contracts them with the bank rows in the kernel.
 *    By default the tables are filled ON DEMAND (NEMO_SOLVE_LAZY, environment, default 1): the solve kernel keeps a bitmap
 *    of (32 vertices x one 4 x 8 pixel patch) blocks per job in LDS and a wave computes the blocks its pixels need and that do not
 *    exist yet, itself, on the matrix cores, before it scores them -- bit for bit the values the whole-table kernel
 *    writes; under a tenth of a job's blocks are ever computed.  Used when the largest registered mesh has at most 2048
 *    vertices; NEMO_SOLVE_LAZY=0: bank_table_kernel fills every job's whole table ahead of the solve kernel.
 * nemo_solve_table_capacity: the number of jobs the reserved tables hold for the meshes registered so far (0: streaming). */
int64_t nemo_solve_table_capacity(const nemo_ctx *ctx);
/* Measurement aid, no counterpart in the reference: device time of the two kernels of the last TABLES-form solve call of
 * this context (its first slice) -- bank_table_kernel and solve_kernel -- from events the call records on its stream.
 * SYNCHRONISES on the last of them.  NEMO_ERR_STATE if the last solve streamed. */
int nemo_solve_last_timing(nemo_ctx *ctx, float *table_ms, float *loop_ms);
/* Measurement aid, no counterpart in the reference.  mode: form of the last solve call of this context -- 0 streaming,
 * 1 tables filled whole by bank_table_kernel, 2 tables filled on demand inside the solve kernel (the default; see
 * NEMO_SOLVE_LAZY above); blocks: (32 vertices x one pixel patch) table blocks computed on demand since the last
 * nemo_counter_reset.  SYNCHRONISES the device. */
int nemo_solve_table_stats(nemo_ctx *ctx, int32_t *mode, uint64_t *blocks);

/* 6-DoF variant of the solve (code/PredUnlabeledDataset.py:318-347): Adam also moves trans_2d, which
 * enters the camera as extra_trans = trans_2d @ [[1,0,0],[0,1,0]] added to T (:328, lib/MeshUtils.py:107-108).
 * Same kernel as nemo_solve.  d_trans0 f32[N][2] (NULL = zeros, :318); d_pose6 f32[N][6] =
 * (Cx,Cy,Cz,theta,t0,t1); d_trace f32[N][steps][13] = (p6, loss, g6) at the start of every step, or NULL;
 * d_status as in nemo_solve (all six components finite). */
int nemo_solve6(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx,
                const float *d_feat, const float *d_clu, const float *d_C0, const float *d_theta0,
                const float *d_trans0, float *d_pose6, float *d_loss, float *d_trace, int32_t *d_status,
                void *stream);

/* nemo_solve for callers that cannot allocate at all (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has NO counterpart in the reference.  Same arguments and results as nemo_solve (bit
 * for bit); the launch-order workspace is a file-scope device array of the library (int32[2 * 16384] = 128 KiB per
 * GPU, held until the process ends) instead of the context's reservation, so neither this call nor any set-up for
 * it allocates.  Limits: N <= 16384 (NEMO_ERR_UNSUPPORTED beyond); meshes whose tables do not fit the LDS are
 * refused (their per-workgroup table scratch is far larger than a static array should be).  With N <= 64 the jobs
 * are launched in index order and the array is not touched at all.  Every call on a GPU shares that GPU's array:
 * calls that overlap on one GPU (two streams, two contexts) race on it, and the race corrupts RESULTS, not only the
 * schedule -- a launch order overwritten by another call makes some jobs run twice and others not at all (their
 * d_pose / d_loss / d_status rows keep whatever the buffers held); job ids outside [0, N) are dropped, so nothing is
 * written out of bounds.  Serialise such calls. */
int nemo_solve_static(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx,
                      const float *d_feat, const float *d_clu, const float *d_C0, const float *d_theta0,
                      float *d_pose, float *d_loss, float *d_trace, int32_t *d_status, void *stream);

/* Bytes the fused kernels read per covered pixel and step: 4*D + 4. */
int64_t nemo_bytes_per_covered_pixel(const nemo_ctx *ctx);

/* Covered-pixel-steps accumulated by nemo_solve / nemo_grid_search / score calls
 * since the counter was last reset (device counter, read back synchronously):
 * the n_cov sum that prices the algorithmic HBM traffic. */
int nemo_counter_reset(nemo_ctx *ctx, void *stream);
int nemo_counter_read(nemo_ctx *ctx, uint64_t *out_covered_pixel_steps);

#ifdef __cplusplus
}
#endif
#endif /* NEMO_HIP_H */
