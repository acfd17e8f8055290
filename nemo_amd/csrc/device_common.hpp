// device_common.hpp -- constants, vector types, small math, camera, raster math, cross-lane reductions, LDS layout
// Part of the single translation unit nemo_hip.hip (included from there, in order; not a stand-alone header).
#ifndef DEVICE_COMMON_HPP
#define DEVICE_COMMON_HPP

namespace {

#ifndef NEMO_NT
#define NEMO_NT 1024
#endif
constexpr int NT = NEMO_NT;    // threads per workgroup of the pose kernels
constexpr int NW = NT / 64;    // waves per workgroup
constexpr int RED_N = 16;      // floats per wave in the reduction scratch
#ifndef NEMO_CHUNK
#define NEMO_CHUNK 64
#endif
constexpr int CH = NEMO_CHUNK;  // covered pixels per score chunk (multiple of 64)
constexpr int UST_N = 96;      // floats of workgroup-uniform state kept in LDS
// uniform-state slots: R[0..8] T[9..11] p[12..15] m[16..19] v[20..23] K[24], block totals [32..45],
// camera intermediates of the current iterate [48..82] (kept for the camera backward)
constexpr int U_R = 0, U_T = 9, U_P = 12, U_M = 16, U_V = 20, U_K = 24, U_TOT = 32, U_CAM = 48;
constexpr int U_P6 = 84, U_M6 = 86, U_V6 = 88;  // 6-DoF variant: trans_2d and its Adam moments
constexpr int U_BLK = 30;                     // LAZY solve: table blocks this job computed (a uint32 in a float slot)
// block totals: [0] loss term, [1..9] dL/dR, [10..12] dL/dT, [13] covered pixels
constexpr int T_LOSS = 0, T_GR = 1, T_GT = 10, T_NCOV = 13;
constexpr int MAX_MESHES = 1024;
#ifdef NEMO_STAMPS
constexpr int STAMP_LDS = 512;  // the diagnostic build's static stamp counters come out of the dynamic budget
#else
constexpr int STAMP_LDS = 0;
#endif
constexpr int LDS_BUDGET = 160 * 1024 / (1024 / NT) - (NT < 1024 ? 512 : 0) - STAMP_LDS;  // the workgroups resident on a CU share its 160 KiB

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// Columns of a row of the solve's dot-product tables (bank_table_kernel, eval_pose<TAB>): the window is cut into
// patches of 4 rows x 8 columns; column c of a table row = patch (c >> 5), position (c & 31) = 8 (row & 3) + (col & 7)
// inside it; patches row-major, table_tpr(w) per patch row.  A (32 vertices x one patch) block is what the matrix cores
// produce at a time, and a face's pixels -- a compact 2-D neighbourhood -- fall into far fewer patches than into runs of
// 32 consecutive pixels of an image row (blocks computed on demand per job: 0.57 of the row-run layout).
__host__ __device__ inline uint32_t table_tpr(uint32_t w) { return (w + 7u) >> 3; }
__host__ __device__ inline uint32_t table_ld(uint32_t h, uint32_t w) { return ((h + 3u) >> 2) * table_tpr(w) * 32u; }
typedef unsigned long long u64;
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
// Pointers that come out of the mesh table are global memory; saying so lets the
// compiler emit global_load (vmcnt only) instead of flat_load.
#define GAS __attribute__((address_space(1)))
typedef const f4 GAS *gf4p;
typedef const float GAS *gf32p;
typedef const int32_t GAS *gi32p;
template <typename T>
__device__ __forceinline__ const T GAS *as_global(const T *p) { return (const T GAS *)(uintptr_t)p; }

struct MeshDesc {
  const float *verts;
  const int32_t *faces;
  const float *bank;
  int32_t V, F;
  // Per face, the order in which the score loop keeps the three bank rows in its registers: bits 2k..2k+1 = the
  // vertex slot held by register set k (identity 0x24).  Chosen at registration so that consecutive faces keep as
  // many rows in place as possible (the loop re-reads a row only when its register set changes vertex).
  const uint8_t *slot_order;
};

struct KParams {
  int S, h, w, r0, c0;
  float s;
  int persp, clutter;
  double keps;
  float inv_hw;
  int cap;   // z-buffer capacity in pixels
  int vmax;  // vertex capacity of the LDS arrays
  int fmax;  // face capacity of the LDS arrays
  unsigned w_magic;  // floor(2^32 / w) + 1: p / w == umulhi(p, w_magic) for p, w < 2^16
  int zb_bytes;      // z-buffer region: max(8*cap, chunk partials + phase A->B hand-off buffer)
  // Large meshes (their per-vertex / per-face tables do not fit the LDS next to a useful z-buffer): the tables
  // live in a per-workgroup slice of a global scratch buffer instead (L2-resident), selected by the BIG
  // template parameter of the kernels.
  int big;
  unsigned big_stride;            // bytes per workgroup slice
  unsigned long long big_scratch; // device address of slice 0
};

// ---------------------------------------------------------------- small math
// Fixed-order fp32 sin/cos; bit-identical twin of the oracle's orc_sincosf.
// Stands in for torch.cos / torch.sin in rotation_theta (lib/MeshUtils.py:74).
__device__ __forceinline__ void dev_sincosf(float x, float &sn, float &cs) {
  const float TWO_OVER_PI = 0.636619772367581343f;
  const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.54978995489188216e-8f;
  float kf = rintf(x * TWO_OVER_PI);
  int k = (int)kf;
  float r = x - kf * P1;
  r = r - kf * P2;
  r = r - kf * P3;
  float z = r * r;
  float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
  ps = ps * z - 1.6666654611e-1f;
  ps = ps * z;
  ps = ps * r + r;
  float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
  pc = pc * z + 4.166664568298827e-2f;
  pc = pc * z;
  pc = pc * z;
  pc = pc - 0.5f * z;
  pc = pc + 1.0f;
  switch (k & 3) {
    case 0: sn = ps; cs = pc; break;
    case 1: sn = pc; cs = -ps; break;
    case 2: sn = -ps; cs = -pc; break;
    default: sn = -pc; cs = ps; break;
  }
}

struct Cam {
  float d[3], nz, zax[3];
  float xr[3], nx, xax[3];
  float yr[3], ny, yax[3];
  float sn, cs;
  float R[9], T[3];
};

__device__ __forceinline__ float clamp_min(float v, float e) { return v > e ? v : e; }

// campos_to_R_T (lib/MeshUtils.py:104-111) with PyTorch3D look_at_rotation
// (at = 0, up = +y; F.normalize eps = 1e-5) and rotation_theta (:57-80).
__device__ __forceinline__ void dev_camera(const float C[3], float theta, const float *extra, Cam &k) {
#pragma unroll
  for (int i = 0; i < 3; ++i) k.d[i] = 0.0f - C[i];
  k.nz = sqrtf((k.d[0] * k.d[0] + k.d[1] * k.d[1]) + k.d[2] * k.d[2]);
  float dz = clamp_min(k.nz, 1e-5f);
#pragma unroll
  for (int i = 0; i < 3; ++i) k.zax[i] = k.d[i] / dz;
  k.xr[0] = k.zax[2]; k.xr[1] = 0.0f; k.xr[2] = 0.0f - k.zax[0];
  k.nx = sqrtf((k.xr[0] * k.xr[0] + k.xr[1] * k.xr[1]) + k.xr[2] * k.xr[2]);
  float dx = clamp_min(k.nx, 1e-5f);
#pragma unroll
  for (int i = 0; i < 3; ++i) k.xax[i] = k.xr[i] / dx;
  k.yr[0] = k.zax[1] * k.xax[2] - k.zax[2] * k.xax[1];
  k.yr[1] = k.zax[2] * k.xax[0] - k.zax[0] * k.xax[2];
  k.yr[2] = k.zax[0] * k.xax[1] - k.zax[1] * k.xax[0];
  k.ny = sqrtf((k.yr[0] * k.yr[0] + k.yr[1] * k.yr[1]) + k.yr[2] * k.yr[2]);
  float dy = clamp_min(k.ny, 1e-5f);
#pragma unroll
  for (int i = 0; i < 3; ++i) k.yax[i] = k.yr[i] / dy;
  dev_sincosf(theta, k.sn, k.cs);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    k.R[3 * i + 0] = k.xax[i] * k.cs + k.yax[i] * k.sn;
    k.R[3 * i + 1] = k.yax[i] * k.cs - k.xax[i] * k.sn;
    k.R[3 * i + 2] = k.zax[i];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float t = (k.R[0 + j] * C[0] + k.R[3 + j] * C[1]) + k.R[6 + j] * C[2];
    k.T[j] = 0.0f - t;
    if (extra) k.T[j] = k.T[j] + extra[j];
  }
}

__device__ __forceinline__ void normalize_bwd(const float g[3], const float raw[3], float n, float out[3]) {
  if (n > 1e-5f) {
    float inv = 1.0f / n;
    float h0 = raw[0] * inv, h1 = raw[1] * inv, h2 = raw[2] * inv;
    float dot = h0 * g[0] + h1 * g[1] + h2 * g[2];
    out[0] = (g[0] - h0 * dot) * inv;
    out[1] = (g[1] - h1 * dot) * inv;
    out[2] = (g[2] - h2 * dot) * inv;
  } else {
    out[0] = g[0] * 1e5f; out[1] = g[1] * 1e5f; out[2] = g[2] * 1e5f;
  }
}

// (dL/dR, dL/dT) -> (dL/dC, dL/dtheta): backward of dev_camera.
__device__ __forceinline__ void dev_camera_bwd(const Cam &k, const float C[3], float GR[9], const float GT[3],
                                               float gC[3], float &gtheta) {
  float gc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      GR[3 * i + j] += -C[i] * GT[j];
      gc[i] += -k.R[3 * i + j] * GT[j];
    }
  float gth = 0.f, gx[3], gy[3], gz[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gth += GR[3 * i] * k.R[3 * i + 1] - GR[3 * i + 1] * k.R[3 * i];
    gx[i] = GR[3 * i] * k.cs - GR[3 * i + 1] * k.sn;
    gy[i] = GR[3 * i] * k.sn + GR[3 * i + 1] * k.cs;
    gz[i] = GR[3 * i + 2];
  }
  float gyr[3], gxr[3], gd[3];
  normalize_bwd(gy, k.yr, k.ny, gyr);
  gz[0] += k.xax[1] * gyr[2] - k.xax[2] * gyr[1];
  gz[1] += k.xax[2] * gyr[0] - k.xax[0] * gyr[2];
  gz[2] += k.xax[0] * gyr[1] - k.xax[1] * gyr[0];
  gx[0] += gyr[1] * k.zax[2] - gyr[2] * k.zax[1];
  gx[1] += gyr[2] * k.zax[0] - gyr[0] * k.zax[2];
  gx[2] += gyr[0] * k.zax[1] - gyr[1] * k.zax[0];
  normalize_bwd(gx, k.xr, k.nx, gxr);
  gz[0] += -gxr[2];
  gz[2] += gxr[0];
  normalize_bwd(gz, k.d, k.nz, gd);
#pragma unroll
  for (int i = 0; i < 3; ++i) gC[i] = gc[i] - gd[i];
  gtheta = gth;
}

// ---------------------------------------------------------------- raster math
__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);  // PyTorch3D EdgeFunctionForward
}
__device__ __forceinline__ float ndc_of(int i, float Sf) {
  return -1.0f + (2.0f * (float)i + 1.0f) / Sf;  // PyTorch3D PixToNdc
}

struct Frag { float b0, b1, b2, pz; };

// One (pixel, face) test of the naive rasterizer, given the
// face's nine screen coordinates v = (x0,y0,z0,x1,...) and its `area`
// (= (float)((double)edge(v2;v0,v1) + kEps)).  The sign pre-test only skips
// divisions whose quotient cannot be > 0; every accepted fragment goes through
// the exact same operations as the oracle.
template <bool PERSP>
__device__ __forceinline__ bool frag_eval(float px, float py, const float *v, float area, Frag &o) {
  float e0 = edge_fn(px, py, v[3], v[4], v[6], v[7]);
  float e1 = edge_fn(px, py, v[6], v[7], v[0], v[1]);
  float e2 = edge_fn(px, py, v[0], v[1], v[3], v[4]);
  if (!PERSP) {
    bool ok = area > 0.0f ? (e0 > 0.0f && e1 > 0.0f && e2 > 0.0f)
                          : (area < 0.0f ? (e0 < 0.0f && e1 < 0.0f && e2 < 0.0f) : true);
    if (!ok) return false;
  }
  float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
  float b0 = w0, b1 = w1, b2 = w2;
  if (PERSP) {
    float t0 = (w0 * v[5]) * v[8];
    float t1 = (v[2] * w1) * v[8];
    float t2 = (v[2] * v[5]) * w2;
    float den = (t0 + t1) + t2;
    b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
  }
  float pz = (b0 * v[2] + b1 * v[5]) + b2 * v[8];
  if (pz < 0.0f) return false;
  if (!(b0 > 0.0f && b1 > 0.0f && b2 > 0.0f)) return false;
  o.b0 = b0; o.b1 = b1; o.b2 = b2; o.pz = pz + 0.0f;  // -0 -> +0 so the key order is the float order
  return true;
}

__device__ __forceinline__ float face_area_eps(const float *v, double keps) {
  return (float)((double)edge_fn(v[6], v[7], v[0], v[1], v[3], v[4]) + keps);
}
__device__ __forceinline__ bool face_degenerate(const float *v, double keps) {
  double fa = (double)edge_fn(v[0], v[1], v[3], v[4], v[6], v[7]);
  return fa <= keps && fa >= -keps;
}

// ---------------------------------------------------------------- cross-lane
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Sum over each aligned group of 8 lanes; every lane of the group gets the total.
__device__ __forceinline__ float sum8(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  return v;
}
// Sums of three values over each row of 16 lanes in nine DPP operations: the mirrored halves of a
// row are duplicates after the first step, so they are merged and reduced together.  On return the
// lanes of quad q (q = 0, 1, 2) of every row hold the row total of value q.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_merge(float keep, float take) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, take),
                                                               CTRL, 0xF, BANK_MASK, false));
}
__device__ __forceinline__ float sum3_row16(float a, float b, float c) {
  a += dpp_mov<0x140>(a);  // row_mirror
  b += dpp_mov<0x140>(b);
  c += dpp_mov<0x140>(c);
  float z = dpp_merge<0xE4, 0xC>(a, c);  // lanes 8-15 carry c
  z += dpp_mov<0x141>(z);                // row_half_mirror
  b += dpp_mov<0x141>(b);
  float w = dpp_merge<0xE4, 0xA>(z, b);  // quads 1 and 3 carry b
  w += dpp_mov<0x4E>(w);
  w += dpp_mov<0xB1>(w);
  return w;
}
template <int LPP>
__device__ __forceinline__ float sum_group(float v) {
  v = sum8(v);
  if (LPP >= 16) v += dpp_mov<0x140>(v);  // row_mirror: lanes 0-7 <-> 15-8 of each row of 16
  if (LPP >= 32) v += __shfl_xor(v, 16);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Sum over the 64 lanes with a fixed tree: DPP inside rows of 16, then the four row totals.
// The result is wave-uniform (built from readlane values).
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = sum8(v);
  v += dpp_mov<0x140>(v);  // row_mirror
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
// Min / max over the 64 lanes: DPP inside rows of 16 (no LDS round trips), then the four row results.
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_mov<0xB1>(v)); v = fminf(v, dpp_mov<0x4E>(v));
  v = fminf(v, dpp_mov<0x141>(v)); v = fminf(v, dpp_mov<0x140>(v));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v)); v = fmaxf(v, dpp_mov<0x140>(v));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// Inclusive prefix sum over the 64 lanes: Hillis-Steele inside rows of 16 with row_shr (lanes shifted in from outside
// the row read 0), then the three row totals are added to the rows after them.
template <int CTRL>
__device__ __forceinline__ int dpp_shr0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int wave_scan_incl(int v, int lane) {
  v += dpp_shr0<0x111>(v); v += dpp_shr0<0x112>(v); v += dpp_shr0<0x114>(v); v += dpp_shr0<0x118>(v);
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
  return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}

// ---------------------------------------------------------------- phase stamps
// Diagnostic build only (-DNEMO_STAMPS): wave 0 / lane 0 of every workgroup adds the
// cycles it spent per phase (barrier waits included) into g_stamps; the shipped
// library contains none of this.
#ifdef NEMO_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_wave_busy[16];  // score-loop busy time by wave index
// The stamps are summed in LDS (slots 0..15 = phases, 16..31 = score-loop busy time by wave) and flushed to the global
// counters once per workgroup (stamp_flush, end of solve_kernel): a global atomic per phase and step from 256 resident
// workgroups onto 16 addresses costs more than the shorter phases last.
__device__ __forceinline__ unsigned long long *stamp_lds() {
  __shared__ unsigned long long s_[32];
  return s_;
}
__device__ __forceinline__ void stamp_zero() { if (threadIdx.x < 32) stamp_lds()[threadIdx.x] = 0ull; }
__device__ __forceinline__ void stamp_flush() {
  if (threadIdx.x < 16) atomicAdd(&g_stamps[threadIdx.x], stamp_lds()[threadIdx.x]);
  else if (threadIdx.x < 32) atomicAdd(&g_wave_busy[threadIdx.x - 16], stamp_lds()[threadIdx.x]);
}
#define STAMP_DECL unsigned long long st_prev = (threadIdx.x == 0) ? __builtin_amdgcn_s_memrealtime() : 0ull
#define STAMP(i)                                                         \
  do {                                                                   \
    if (threadIdx.x == 0) {                                              \
      const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();      \
      stamp_lds()[i] += now_ - st_prev;                                  \
      st_prev = now_;                                                    \
    }                                                                    \
  } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#endif

// ---------------------------------------------------------------- LDS layout
constexpr int FACE_BYTES = 8 + 16;  // per face: vertex ids (us4) + FaceRec (uint4)
// FaceRec, written by the rasteriser for the list builder, per face and band:
//   x = a_lo | a_hi << 16, y = b_lo | b_hi << 16 | masked << 31, (z, w) = 64-bit mask.
// masked: the face's candidate box is at most 8 x 8 pixel centres; (a_lo.., b_lo..) is that box and bit
//   8 * (b - b_lo) + (a - a_lo) of the mask says that the face produced a fragment there.
// otherwise: the box is the bounding box of the face's fragments and the mask is unused.
struct Lds {
  u64 *zb;               // [cap] packed z-buffer
  f4 *sv;                // [vmax] screen verts (u, t, z_view, -)
  us4 *fc;               // [fmax] face vertex ids (i0, i1, i2, -) as 16-bit
  u4v *finfo;            // [fmax] per face, per band: FaceRec (candidate / fragment box + 8x8 fragment mask)
  __attribute__((address_space(1))) f4 *gsv;     // BIG: the same three tables in global scratch
  __attribute__((address_space(1))) us4 *gfc;
  __attribute__((address_space(1))) u4v *gfinfo;
  float *ndc;            // [S] pixel-centre NDC coordinate of flipped index a: PixToNdc(a, S)
  uint32_t *list;        // [cap] covered pixels, face-major: (face id << 16) | window pixel index
  float *red;            // [NW][RED_N]
  float *ust;            // [UST_N] workgroup-uniform state (camera, Adam)
};

__host__ __device__ inline size_t zb_region_bytes(int cap) {
  const size_t a = (size_t)cap * 8, b = (size_t)NW * CH * (12 + 16);  // z-buffer, later the score loop's per-wave scratch
  return ((a > b ? a : b) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t lds_bytes(int cap, int vmax, int fmax, int wmax, int S) {
  size_t b = zb_region_bytes(cap) + (size_t)cap * 4 + (size_t)vmax * 16 + (size_t)fmax * FACE_BYTES;
  b += (NW * RED_N + UST_N + ((S + 11) & ~3)) * 4 + 64;
  (void)wmax;
  return (b + 15) & ~(size_t)15;
}

__device__ __forceinline__ Lds lds_carve(unsigned char *base, const KParams &P) {
  Lds L;
  L.zb = (u64 *)base;
  L.sv = (f4 *)(base + (size_t)P.zb_bytes);
  const int vl = P.big ? 0 : P.vmax, fl = P.big ? 0 : P.fmax;  // table entries kept in LDS
  L.fc = (us4 *)(L.sv + vl);
  L.finfo = (u4v *)(L.fc + fl);
  L.list = (uint32_t *)(L.finfo + fl);
  L.red = (float *)(L.list + P.cap);
  L.ust = L.red + NW * RED_N;
  L.ndc = L.ust + UST_N + 2;  // indices -2 .. S+1
  unsigned char *g = (unsigned char *)(P.big_scratch + (unsigned long long)blockIdx.x * P.big_stride);
  L.gsv = (__attribute__((address_space(1))) f4 *)g;
  L.gfc = (__attribute__((address_space(1))) us4 *)(g + (size_t)P.vmax * 16);
  L.gfinfo = (__attribute__((address_space(1))) u4v *)(L.gfc + P.fmax);
  return L;
}
// Table accessors: LDS for ordinary meshes, the global scratch slice for BIG ones.
template <bool BIG> __device__ __forceinline__ f4 sv_at(const Lds &L, int i) { if constexpr (BIG) return L.gsv[i]; else return L.sv[i]; }
template <bool BIG> __device__ __forceinline__ void sv_put(const Lds &L, int i, f4 v) { if constexpr (BIG) L.gsv[i] = v; else L.sv[i] = v; }
template <bool BIG> __device__ __forceinline__ us4 fc_at(const Lds &L, int i) { if constexpr (BIG) return L.gfc[i]; else return L.fc[i]; }
template <bool BIG> __device__ __forceinline__ void fc_put(const Lds &L, int i, us4 v) { if constexpr (BIG) L.gfc[i] = v; else L.fc[i] = v; }
template <bool BIG> __device__ __forceinline__ u4v finfo_at(const Lds &L, int i) { if constexpr (BIG) return L.gfinfo[i]; else return L.finfo[i]; }
template <bool BIG> __device__ __forceinline__ void finfo_put(const Lds &L, int i, u4v v) { if constexpr (BIG) L.gfinfo[i] = v; else L.finfo[i] = v; }
// The winners' mask of a face replaces its fragment mask (second half of the record).
template <bool BIG> __device__ __forceinline__ void finfo_put_mask(const Lds &L, int i, u64 m) {
  const u2v v = u2v{(uint32_t)m, (uint32_t)(m >> 32)};
  if constexpr (BIG) *((__attribute__((address_space(1))) u2v *)(L.gfinfo + i) + 1) = v; else *((u2v *)(L.finfo + i) + 1) = v;
}

constexpr int MAX_GRIDS = 128;  // pose grids per context (MultiCuboid: one per subtype mesh, 79 in all)
constexpr int GRID_POSES_PER_WG = 4;
struct PreStore {  // pre-rendered grid poses (device pointers; slot == nullptr: none)
  const int32_t *slot;   // [MAX_MESHES][MAX_GRIDS] -> pair slot or -1
  const int32_t *cnt;    // [pairs][G] list lengths
  const uint32_t *list;  // [pairs][G][stride] face-major lists
  size_t stride;
};

struct EmitOut {  // standalone rasterizer outputs (window), may be null
  int32_t *p2f;
  float *bary;
  float *zbuf;
  uint32_t *list_out;        // pre-render: the face-major list itself
  const uint32_t *pre_list;  // PRE: list built earlier by prerender_kernel
  int pre_cnt;
};

template <bool BIG>
__device__ __forceinline__ void load_face_verts(const Lds &L, int i0, int i1, int i2, float *v) {
  const f4 s0 = sv_at<BIG>(L, i0), s1 = sv_at<BIG>(L, i1), s2 = sv_at<BIG>(L, i2);
  v[0] = s0.x; v[1] = s0.y; v[2] = s0.z;
  v[3] = s1.x; v[4] = s1.y; v[5] = s1.z;
  v[6] = s2.x; v[7] = s2.y; v[8] = s2.z;
}

// Block totals left in LDS by eval_pose (L.ust[U_TOT + i], i = T_LOSS, T_GR.., T_GT.., T_NCOV).
__device__ __forceinline__ float block_total(const Lds &L, int i) { return L.ust[U_TOT + i]; }

}  // namespace

#endif  // DEVICE_COMMON_HPP
