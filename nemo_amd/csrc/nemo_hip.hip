// nemo_hip.hip -- hand-written CDNA4 (gfx950) kernels + C ABI of libnemo_hip.so.
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
// block totals: [0] loss term, [1..9] dL/dR, [10..12] dL/dT, [13] covered pixels
constexpr int T_LOSS = 0, T_GR = 1, T_GT = 10, T_NCOV = 13;
constexpr int MAX_MESHES = 1024;
constexpr int LDS_BUDGET = 160 * 1024 / (1024 / NT) - (NT < 1024 ? 512 : 0);  // the workgroups resident on a CU share its 160 KiB

typedef float f4 __attribute__((ext_vector_type(4)));
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
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---------------------------------------------------------------- phase stamps
// Diagnostic build only (-DNEMO_STAMPS): wave 0 / lane 0 of every workgroup adds the
// cycles it spent per phase (barrier waits included) into g_stamps; the shipped
// library contains none of this.
#ifdef NEMO_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_wave_busy[16];  // score-loop busy time by wave index
#define STAMP_DECL unsigned long long st_prev = (threadIdx.x == 0) ? __builtin_amdgcn_s_memrealtime() : 0ull
#define STAMP(i)                                                         \
  do {                                                                   \
    if (threadIdx.x == 0) {                                              \
      const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();      \
      atomicAdd(&g_stamps[i], now_ - st_prev);                           \
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

// ---------------------------------------------------------------- the evaluation
// Whole-workgroup routine.  Camera R (row-major) and T are read from L.ust; the
// per-wave partial totals are left in L.red (see block_total) behind a barrier.
// RAW = true: `feat` holds an upstream gradient dL/d(out) [h][w][D] of the
// interpolated map and the routine back-propagates it (module path); otherwise
// `feat` is the image feature map and the loss of MeshPoseSolveAll.py:58-59 is used.
template <int D, bool PERSP, bool GRAD, bool SCORE, bool RAW = false, bool PRE = false, bool BIG = false, bool TAIL_SYNC = true>
__device__ void eval_pose(const KParams &P, const Lds &L, const MeshDesc &M, const float *__restrict__ feat,
                          const float *__restrict__ clu, EmitOut emit, bool zb_clean = false,
                          const float *__restrict__ vpre = nullptr) {
#ifndef NEMO_KIT
#define NEMO_KIT 2
#endif
  constexpr int KIT = NEMO_KIT;             // 16-byte loads per row per lane
  constexpr int LPP = D / (4 * KIT);        // lanes per pixel: 16 (D=128) or 32 (D=256)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: everything derived from it lives in SGPRs
  const int S = P.S;
  const float Sf = (float)S;
  const int V = M.V, F = M.F;
  const float s = P.s;
  const gf32p bank = as_global(M.bank);
  const gf32p gverts = as_global(M.verts);
  const gf32p gfeat = as_global(feat);
  const gf32p gclu = as_global(clu);

  STAMP_DECL;
  if (!SCORE && tid < 16) L.ust[U_TOT + tid] = 0.0f;  // ordered before any use by the bounding-box barrier
  // ---- 1. project vertices (MeshRasterizer.transform; order fixed in DESIGN.md)
  float umin = 3.0e38f, umax = -3.0e38f, tmin = 3.0e38f, tmax = -3.0e38f;
  {
    const float *R = L.ust + U_R, *T = L.ust + U_T;
    const float m00 = R[0] * s, m10 = R[3] * s, m20 = R[6] * s, m30 = T[0] * s;
    const float m01 = R[1] * s, m11 = R[4] * s, m21 = R[7] * s, m31 = T[1] * s;
    const float r02 = R[2], r12 = R[5], r22 = R[8], t2 = T[2];
    int slot = 0;
    for (int v = tid; v < V; v += NT, ++slot) {
      // vpre: the caller loaded this thread's first two vertices ahead of time (solve kernel: under the Adam section)
      const bool pre = vpre != nullptr && slot < 2;
      const float x = pre ? vpre[3 * slot] : gverts[3 * v], y = pre ? vpre[3 * slot + 1] : gverts[3 * v + 1],
                  z = pre ? vpre[3 * slot + 2] : gverts[3 * v + 2];
      const float xc = ((x * m00 + y * m10) + z * m20) + m30;
      const float yc = ((x * m01 + y * m11) + z * m21) + m31;
      const float zv = ((x * r02 + y * r12) + z * r22) + t2;
      const float u = xc / zv, t = yc / zv;
      sv_put<BIG>(L, v, f4{u, t, zv, 0.0f});
      umin = fminf(umin, u); umax = fmaxf(umax, u);
      tmin = fminf(tmin, t); tmax = fmaxf(tmax, t);
    }
  }
  umin = wave_min(umin); umax = wave_max(umax); tmin = wave_min(tmin); tmax = wave_max(tmax);
  if (lane == 0) {
    L.red[wave * RED_N + 0] = umin; L.red[wave * RED_N + 1] = umax;
    L.red[wave * RED_N + 2] = tmin; L.red[wave * RED_N + 3] = tmax;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    umin = fminf(umin, L.red[k * RED_N + 0]); umax = fmaxf(umax, L.red[k * RED_N + 1]);
    tmin = fminf(tmin, L.red[k * RED_N + 2]); tmax = fmaxf(tmax, L.red[k * RED_N + 3]);
  }
  // Flipped pixel indices a = S-1-col, b = S-1-row: pixel centre = ndc_of(a), ndc_of(b).
  // Conservative index range of the object's bounding box, clipped to the window.
  const int a_w0 = S - P.c0 - P.w, a_w1 = S - 1 - P.c0;  // window in flipped cols
  const int b_w0 = S - P.r0 - P.h, b_w1 = S - 1 - P.r0;  // window in flipped rows
  auto lo_idx = [&](float vmin) {
    float f = ((fmaxf(fminf(vmin, 4.0f), -4.0f) + 1.0f) * Sf - 1.0f) * 0.5f;
    return (int)floorf(f) - 1;
  };
  auto hi_idx = [&](float vmax_) {
    float f = ((fmaxf(fminf(vmax_, 4.0f), -4.0f) + 1.0f) * Sf - 1.0f) * 0.5f;
    return (int)ceilf(f) + 1;
  };
  // [lo, hi] = indices i in [clip_lo, clip_hi] with vmin <= ndc[i] <= vmax, exactly: ndc[] is monotone
  // and the closed-form estimates floor(f) / ceil(f) are within [0, +2] / [-2, 0] of the answer
  // (checked exhaustively in tests/test_host_cpu.py), so three table probes each settle it without a
  // loop.  The table carries two extra entries on either side (indices -2 .. S+1).
  auto exact_range = [&](float vmin, float vmax_, int clip_lo, int clip_hi, int &lo, int &hi) {
    const int e = lo_idx(vmin) + 1, ec = hi_idx(vmax_) - 1;  // floor(f(vmin)), ceil(f(vmax))
    auto at = [&](int i) { return L.ndc[min(max(i, -2), S + 1)]; };
    const float l0 = at(e - 1), l1 = at(e), l2 = at(e + 1), h0 = at(ec + 1), h1 = at(ec), h2 = at(ec - 1);
    lo = (e - 1) + (int)(l0 < vmin) + (int)(l1 < vmin) + (int)(l2 < vmin);
    hi = (ec + 1) - (int)(h0 > vmax_) - (int)(h1 > vmax_) - (int)(h2 > vmax_);
    lo = min(max(lo, clip_lo), clip_hi + 1);
    hi = max(min(hi, clip_hi), clip_lo - 1);
  };
  // workgroup-uniform values read back from LDS live in VGPRs unless told otherwise
  const int a_lo = __builtin_amdgcn_readfirstlane(max(lo_idx(umin), a_w0));
  const int a_hi = __builtin_amdgcn_readfirstlane(min(hi_idx(umax), a_w1));
  const int b_lo = __builtin_amdgcn_readfirstlane(max(lo_idx(tmin), b_w0));
  const int b_hi = __builtin_amdgcn_readfirstlane(min(hi_idx(tmax), b_w1));
  const int bw = a_hi - a_lo + 1;
  const bool empty_box = (bw <= 0) || (b_hi < b_lo);
  STAMP(0);

  int ncov = 0, ebase = 0;

  // ---- consumer of one band's face-major list (L.list[0..ntot)): emit or score; ends behind a barrier
#ifdef NEMO_STAMPS
  unsigned long long st_score0 = 0;
#endif
  float tot = 0.0f;  // wave 0, lane i < 13: running total i of the evaluation (T_LOSS, T_GR.., T_GT..)
  auto consume = [&](const int ntot, const int ebase, const bool last) {
    if (!SCORE) {
      // pre-render: keep the list itself; standalone rasterizer: emit pix_to_face / bary / zbuf
      if (emit.list_out)
        for (int e = tid; e < ntot; e += NT) emit.list_out[ebase + e] = L.list[e];
      for (int e = tid; emit.p2f && e < ntot; e += NT) {
        const uint32_t ent = L.list[e];
        const int p = (int)(ent & 0xffffu), f = (int)(ent >> 16);
        const int i = (int)__umulhi((unsigned)p, P.w_magic), j = p - i * P.w;
        const int a = S - 1 - (j + P.c0), b = S - 1 - (i + P.r0);
        const us4 fv_ = fc_at<BIG>(L, f);
        float v[9];
        load_face_verts<BIG>(L, fv_.x, fv_.y, fv_.z, v);
        Frag fr;
        fr.b0 = fr.b1 = fr.b2 = fr.pz = -1.0f;
        frag_eval<PERSP>(L.ndc[a], L.ndc[b], v, face_area_eps(v, P.keps), fr);
        emit.p2f[p] = f;
        emit.bary[3 * p] = fr.b0; emit.bary[3 * p + 1] = fr.b1; emit.bary[3 * p + 2] = fr.b2;
        if (emit.zbuf) emit.zbuf[p] = fr.pz;
      }
    } else {
      // ---- 3. fused interpolate + score (+ backward).  Each wave owns a contiguous
      // segment of the face-major list and works through it in chunks of 64 pixels:
      //  phase A (LPP lanes per pixel): 16-byte loads of the pixel's feature row and of
      //    the face's three vertex-bank rows, three D-long dot products reduced with
      //    DPP, results parked in a wave-private LDS buffer.  A lane group walks
      //    consecutive list entries and re-reads a bank row only when that vertex slot
      //    changes (the two triangles of a cuboid cell share two of their three slots).
      //    The next entry's feature row is always in flight (ping-pong registers); the
      //    prefetch is unconditional because a conditional load would make the compiler
      //    drain vmcnt to zero every iteration.
      //  phase B (one lane per pixel): barycentrics, object score, loss term and the
      //    hand-derived backward to dL/dR, dL/dT accumulators.
      // Both phases are wave-local (LDS operations of one wave complete in order), so
      // there is no workgroup barrier inside this loop.  A wave keeps its 13 sums in registers over
      // all its chunks, reduces them across lanes once (fixed DPP tree) and leaves them in L.red; the
      // split of the list over the waves is a function of its length only, so the totals are
      // reproducible bit for bit.
      constexpr int GPW = 64 / LPP;
      constexpr int NP = GRAD ? 13 : 1;  // partials per chunk: loss term, dL/dR (9), dL/dT (3)
      const int gl = lane & (LPP - 1), grp = lane / LPP;
#ifdef NEMO_STAMPS
      st_score0 = __builtin_amdgcn_s_memrealtime();
#endif
      uint4 *rec = (uint4 *)L.zb + wave * CH;                               // [NW][CH]   (the z-buffer is dead by now
      float *gbuf = (float *)((uint4 *)L.zb + NW * CH) + wave * (CH * 3);   // [NW][CH*3]  and hosts the wave-private scratch)
      // rows are addressed as uniform base + 32-bit byte offset (feature window and bank are < 4 GiB)
      typedef const __attribute__((address_space(1))) char *gbytep;
      const gbytep fbytes = (gbytep)gfeat, bbytes = (gbytep)bank;
      const uint32_t gl16 = (uint32_t)gl * 16u;
      // byte offset of the feature row of this lane group's first entry in the chunk [cb_, cb_ + cn_)
      auto first_off_at = [&](int cb_, int cn_) {
        const int per_ = (cn_ + GPW - 1) / GPW;
        return (L.list[cb_ + min(grp * per_, cn_ - 1)] & 0xffffu) * (uint32_t)(D * 4) + gl16;
      };
      f4 xa[KIT], xb[KIT];  // ping-pong: current / next feature-row fragments
      // Static split: wave w owns the contiguous entries [ntot w / NW, ntot (w + 1) / NW) and walks them in
      // chunks of CH; its 13 sums stay in registers across chunks and are reduced across lanes once.
      const int seg_b = (ntot * wave) / NW, seg_e = (ntot * (wave + 1)) / NW;
      float pl = 0.f;
      float GR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float GT[3] = {0.f, 0.f, 0.f};
      int cb = seg_b;
      if (cb < seg_e) {
        const gf4p Fp = (gf4p)(fbytes + first_off_at(cb, min(CH, seg_e - cb)));
#pragma unroll
        for (int k = 0; k < KIT; ++k) xa[k] = __builtin_nontemporal_load(Fp + LPP * k);
      }
      while (cb < seg_e) {
        const int cn = min(CH, seg_e - cb);
        const int cb_next = cb + CH;
        const int per = (cn + GPW - 1) / GPW;  // entries per lane group in this chunk
        // ---------------- records (one lane per entry): everything phase A needs to know about an entry
        // -- feature-row byte offset and the three bank rows -- is worked out once here, 64 entries per
        // instruction, instead of by every lane of a group inside the loop.  The same pass fetches this
        // lane's phase-B list entries and clutter scores (they land during phase A).
        uint32_t ent_bs[CH / 64];
        float cls[CH / 64];
#pragma unroll
        for (int u = 0; u < CH / 64; ++u) {
          const uint32_t ent = L.list[cb + min(u * 64 + lane, cn - 1)];
          ent_bs[u] = ent;
          cls[u] = (P.clutter && clu) ? gclu[ent & 0xffffu] : 0.0f;
          if (u * 64 < cn) {
            const us4 fv_ = fc_at<BIG>(L, ent >> 16);
            // register set k of phase A holds the bank row of vertex slot (code >> 2k) & 3 (MeshDesc::slot_order)
            const uint32_t code = fv_.w;
            auto slot_id = [&](uint32_t c) { return (uint32_t)(c == 0 ? fv_.x : (c == 1 ? fv_.y : fv_.z)); };
            rec[u * 64 + lane] = uint4{(ent & 0xffffu) * (uint32_t)(D * 4),
                                       slot_id(code & 3u) | (slot_id((code >> 2) & 3u) << 16), slot_id((code >> 4) & 3u), code};
          }
        }
        // ---------------- phase A
        {
          // first feature row of this group in the wave's next chunk (or a harmless reload at the very end)
          const int e0 = grp * per;
          uint4 R0 = rec[min(e0, cn - 1)], R1 = rec[min(e0 + 1, cn - 1)];  // records run two entries ahead
          const uint32_t off_next_chunk = cb_next < seg_e ? first_off_at(cb_next, min(CH, seg_e - cb_next)) : R0.x + gl16;
          int i0 = -1, i1 = -1, i2 = -1;
          f4 y0[KIT], y1[KIT], y2[KIT];
#pragma unroll
          for (int k = 0; k < KIT; ++k) y0[k] = y1[k] = y2[k] = f4{0.f, 0.f, 0.f, 0.f};
          auto body = [&](int j, f4(&xc)[KIT], f4(&xnext)[KIT]) {
            const int el = e0 + j;
            const uint4 R2 = rec[min(el + 2, cn - 1)];
            // 1st: the loads this entry depends on (bank rows of a changed vertex slot) ...
            const int n0 = (int)(R0.y & 0xffffu), n1 = (int)(R0.y >> 16), n2 = (int)R0.z;
            if (n0 != i0) {
              i0 = n0;
              const gf4p B = (gf4p)(bbytes + ((uint32_t)n0 * (uint32_t)(D * 4) + gl16));
#pragma unroll
              for (int k = 0; k < KIT; ++k) y0[k] = B[LPP * k];
            }
            if (n1 != i1) {
              i1 = n1;
              const gf4p B = (gf4p)(bbytes + ((uint32_t)n1 * (uint32_t)(D * 4) + gl16));
#pragma unroll
              for (int k = 0; k < KIT; ++k) y1[k] = B[LPP * k];
            }
            if (n2 != i2) {
              i2 = n2;
              const gf4p B = (gf4p)(bbytes + ((uint32_t)n2 * (uint32_t)(D * 4) + gl16));
#pragma unroll
              for (int k = 0; k < KIT; ++k) y2[k] = B[LPP * k];
            }
            // ... 2nd: the next entry's feature row, which stays in flight (vmcnt is in order)
            {
              const uint32_t offn = (j + 1 < per) ? R1.x + gl16 : off_next_chunk;
              const gf4p Fp = (gf4p)(fbytes + offn);
#pragma unroll
              for (int k = 0; k < KIT; ++k)
                xnext[k] = __builtin_nontemporal_load(Fp + LPP * k);
            }
            f4 A0 = xc[0] * y0[0], A1 = xc[0] * y1[0], A2 = xc[0] * y2[0];
#pragma unroll
            for (int k = 1; k < KIT; ++k) {
              A0 = __builtin_elementwise_fma(xc[k], y0[k], A0);
              A1 = __builtin_elementwise_fma(xc[k], y1[k], A1);
              A2 = __builtin_elementwise_fma(xc[k], y2[k], A2);
            }
            float s0 = (A0.x + A0.y) + (A0.z + A0.w), s1 = (A1.x + A1.y) + (A1.z + A1.w),
                  s2 = (A2.x + A2.y) + (A2.z + A2.w);
            if (LPP >= 16) {
              if (LPP >= 32) {
                s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
              }
              const float w = sum3_row16(s0, s1, s2);  // quad q of the row holds the total of value q (q < 3)
              // value q belongs to vertex slot (code >> 2q) & 3: phase B reads the hand-off buffer in slot order
              if (el < cn && (gl & 3) == 0 && gl < 12) gbuf[3 * el + ((R0.w >> ((gl >> 2) * 2)) & 3u)] = w;
            } else {
              const float g0 = sum8(s0), g1 = sum8(s1), g2 = sum8(s2);
              if (gl == 0 && el < cn) {
                gbuf[3 * el + (R0.w & 3u)] = g0; gbuf[3 * el + ((R0.w >> 2) & 3u)] = g1; gbuf[3 * el + ((R0.w >> 4) & 3u)] = g2;
              }
            }
            R0 = R1; R1 = R2;
          };
          int j = 0;
          for (; j + 1 < per; j += 2) {
            body(j, xa, xb);
            body(j + 1, xb, xa);
          }
          if (j < per) {  // odd tail: one more entry, then put the prefetched row back in xa
            body(j, xa, xb);
#pragma unroll
            for (int k = 0; k < KIT; ++k) xa[k] = xb[k];
          }
        }
        STAMP(8);
        // ---------------- phase B: this lane's pixels -> its 13 contributions
#pragma unroll
        for (int u = 0; u < CH / 64; ++u)
        if (u * 64 + lane < cn) {
          const uint32_t ent_b = ent_bs[u];
          const float cl = cls[u];
          const int lb = u * 64 + lane;
          const int p = (int)(ent_b & 0xffffu), f = (int)(ent_b >> 16);
          const int pi = (int)__umulhi((unsigned)p, P.w_magic), pj = p - pi * P.w;
          const int a = S - 1 - (pj + P.c0), b = S - 1 - (pi + P.r0);
          const us4 fv_ = fc_at<BIG>(L, f);
          const int i0 = fv_.x, i1 = fv_.y, i2 = fv_.z;
          const float g0 = gbuf[3 * lb], g1 = gbuf[3 * lb + 1], g2 = gbuf[3 * lb + 2];
          float v[9];
          load_face_verts<BIG>(L, i0, i1, i2, v);
          const float px = L.ndc[a], py = L.ndc[b];
          // coverage was decided (exactly) by the rasterizer; the quantities below only feed the
          // 1e-4-tolerance loss and gradient, so reciprocals replace IEEE divisions here
          const float ia = __builtin_amdgcn_rcpf(face_area_eps(v, P.keps));
          const float w0 = edge_fn(px, py, v[3], v[4], v[6], v[7]) * ia;
          const float w1 = edge_fn(px, py, v[6], v[7], v[0], v[1]) * ia;
          const float w2 = edge_fn(px, py, v[0], v[1], v[3], v[4]) * ia;
          float b0 = w0, b1 = w1, b2 = w2, iden = 1.0f;
          if (PERSP) {
            const float t0 = (w0 * v[5]) * v[8], t1 = (v[2] * w1) * v[8], t2 = (v[2] * v[5]) * w2;
            iden = __builtin_amdgcn_rcpf((t0 + t1) + t2);
            b0 = t0 * iden; b1 = t1 * iden; b2 = t2 * iden;
          }
          const float obj = (b0 * g0 + b1 * g1) + b2 * g2;
          float ind;
          if (RAW) {
            ind = 1.0f;
          } else if (P.clutter) {
            pl += fmaxf(obj, cl) - fmaxf(0.0f, cl);
            ind = obj > cl ? 1.0f : (obj == cl ? 0.5f : 0.0f);
          } else {
            pl += obj;
            ind = 1.0f;
          }
          if (GRAD && ind != 0.0f) {
            const float up = RAW ? 1.0f : -P.inv_hw * ind;
            const float gb0 = up * g0, gb1 = up * g1, gb2 = up * g2;
            float gw0 = gb0, gw1 = gb1, gw2 = gb2, gz0 = 0.f, gz1 = 0.f, gz2 = 0.f;
            if (PERSP) {
              const float z0 = v[2], z1 = v[5], z2 = v[8];
              const float dot = (gb0 * b0 + gb1 * b1) + gb2 * b2;
              const float gt0 = (gb0 - dot) * iden, gt1 = (gb1 - dot) * iden, gt2 = (gb2 - dot) * iden;
              gw0 = gt0 * z1 * z2; gw1 = gt1 * z0 * z2; gw2 = gt2 * z0 * z1;
              gz0 = gt1 * w1 * z2 + gt2 * z1 * w2;
              gz1 = gt0 * w0 * z2 + gt2 * z0 * w2;
              gz2 = gt0 * w0 * z1 + gt1 * z0 * w1;
            }
            const float gE0 = gw0 * ia, gE1 = gw1 * ia, gE2 = gw2 * ia;
            const float gA = -((gw0 * w0 + gw1 * w1) + gw2 * w2) * ia;
            float gv[9];
            gv[0] = gE1 * (v[7] - py) + gE2 * (py - v[4]) + gA * (v[7] - v[4]);
            gv[1] = gE1 * (px - v[6]) + gE2 * (v[3] - px) + gA * (v[3] - v[6]);
            gv[2] = gz0;
            gv[3] = gE0 * (py - v[7]) + gE2 * (v[1] - py) + gA * (v[1] - v[7]);
            gv[4] = gE0 * (v[6] - px) + gE2 * (px - v[0]) + gA * (v[6] - v[0]);
            gv[5] = gz1;
            gv[6] = gE0 * (v[4] - py) + gE1 * (py - v[1]) + gA * (v[4] - v[1]);
            gv[7] = gE0 * (px - v[3]) + gE1 * (v[0] - px) + gA * (v[0] - v[3]);
            gv[8] = gz2;
            const int vi[3] = {i0, i1, i2};
            const float inv_s = 1.0f / s, T0 = L.ust[U_T], T1 = L.ust[U_T + 1], T2 = L.ust[U_T + 2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float iz = __builtin_amdgcn_rcpf(v[3 * c + 2]);
              const float gxc = gv[3 * c] * iz, gyc = gv[3 * c + 1] * iz;
              const float gzv = gv[3 * c + 2] - (gv[3 * c] * v[3 * c] + gv[3 * c + 1] * v[3 * c + 1]) * iz;
              const float q0 = s * gxc, q1 = s * gyc, q2 = gzv;
              // world vertex = (view - T) R^T: accumulate against (view - T) and rotate the 3x3 total once
              // at the end (no world-space vertex table in LDS)
              const float zs = v[3 * c + 2] * inv_s;
              const float X0 = v[3 * c] * zs - T0, X1 = v[3 * c + 1] * zs - T1, X2 = v[3 * c + 2] - T2;
              (void)vi;
              GR[0] = fmaf(X0, q0, GR[0]); GR[1] = fmaf(X0, q1, GR[1]); GR[2] = fmaf(X0, q2, GR[2]);
              GR[3] = fmaf(X1, q0, GR[3]); GR[4] = fmaf(X1, q1, GR[4]); GR[5] = fmaf(X1, q2, GR[5]);
              GR[6] = fmaf(X2, q0, GR[6]); GR[7] = fmaf(X2, q1, GR[7]); GR[8] = fmaf(X2, q2, GR[8]);
              GT[0] += q0; GT[1] += q1; GT[2] += q2;
            }
          }
        }
        STAMP(9);
        cb = cb_next;
      }
      {  // the wave's partials (fixed DPP tree), left in L.red by one lane
        float pv[13];
        pv[0] = wave_sum_dpp(pl);
        if (GRAD) {
#pragma unroll
          for (int i = 0; i < 9; ++i) pv[1 + i] = wave_sum_dpp(GR[i]);
#pragma unroll
          for (int i = 0; i < 3; ++i) pv[10 + i] = wave_sum_dpp(GT[i]);
        }
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < NP; ++i) L.red[wave * RED_N + i] = pv[i];
        }
      }
    }
    STAMP(4);
#ifdef NEMO_STAMPS
    const unsigned long long st_arrive = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();  // all chunks of the band are done
#ifdef NEMO_STAMPS
    if (SCORE && (threadIdx.x & 63) == 0) {  // every wave: its wait at the band barrier, and its busy time in the score loop
      atomicAdd(&g_stamps[10], __builtin_amdgcn_s_memrealtime() - st_arrive);
      atomicAdd(&g_stamps[11], st_arrive - st_score0);
      atomicAdd(&g_wave_busy[(threadIdx.x >> 6) & 15], st_arrive - st_score0);
    }
#endif
    if (SCORE) {
      // the waves' partials -> running totals, in wave order: lane i of wave 0 keeps total i in a register
      constexpr int NPB = GRAD ? 13 : 1;
      if (wave == 0 && lane < NPB) {
#pragma unroll
        for (int k = 0; k < NW; ++k) tot += L.red[k * RED_N + lane];
      }
      // After the LAST band nothing of the band's LDS is reused before the caller's next barrier, and the totals
      // live in wave 0's registers: no second barrier.
      if (last) return;
    }
    __syncthreads();  // band's z-buffer / lists / L.red are reused by the next band
    STAMP(5);
  };


  if (PRE) {
    // pre-rendered pose (grid search): the face-major list was built once per (mesh, pose) by
    // prerender_kernel; it is consumed in pieces of at most `cap` entries
    for (int pb = 0; pb < emit.pre_cnt; pb += P.cap) {
      const int ntot = min(P.cap, emit.pre_cnt - pb);
      for (int i = tid; i < ntot; i += NT) L.list[i] = emit.pre_list[pb + i];
      ncov += (tid == 0) ? ntot : 0;
      __syncthreads();
      consume(ntot, ebase, pb + P.cap >= emit.pre_cnt);
      ebase += ntot;
    }
  } else if (!empty_box) {
    const int rows_per_band = P.cap / bw;  // >= 1 because cap >= w >= bw
    for (int bb = b_lo; bb <= b_hi; bb += rows_per_band) {
      const int nr = min(rows_per_band, b_hi - bb + 1);
      const int npx = nr * bw;
      // ---- 2a. clear the band's z-buffer (zb_clean: the caller left all `cap` keys cleared behind a barrier)
      if (!(zb_clean && bb == b_lo)) {
        for (int i = tid; i < npx; i += NT) L.zb[i] = ~0ull;
        __syncthreads();
      }
      STAMP(1);
      // ---- 2b. face-parallel rasterisation (one thread per face).  The candidate pixel centres of a face
      // (its exact bounding-box ranges) are walked as ONE flat loop, so a wave runs max-over-lanes of the
      // candidate COUNT, not of rows x columns; the sign pre-test is branch-free; the three quotients
      // e_k / area use the hoisted exact division (fastdiv.h), the IEEE expansion only for operands outside
      // its safe range.  Every accepted fragment sees the same operations, in the same order, as the oracle.
      for (int f = tid; f < F; f += NT) {
        const us4 fv_ = fc_at<BIG>(L, f);
        const int i0 = fv_.x, i1 = fv_.y, i2 = fv_.z;
        float v[9];
        load_face_verts<BIG>(L, i0, i1, i2, v);
        u4v rec = u4v{1u, 1u | (1u << 31), 0u, 0u};  // empty box, empty mask
        if (!face_degenerate(v, P.keps)) {
          const float xmin = fminf(fminf(v[0], v[3]), v[6]), xmax = fmaxf(fmaxf(v[0], v[3]), v[6]);
          const float ymin = fminf(fminf(v[1], v[4]), v[7]), ymax = fmaxf(fmaxf(v[1], v[4]), v[7]);
          // exact index ranges of the pixel centres inside the face's bounding box (the test
          // PyTorch3D applies per pixel), clipped to the object box / band
          int fa_lo, fa_hi, fb_lo, fb_hi;
          exact_range(xmin, xmax, a_lo, a_hi, fa_lo, fa_hi);
          exact_range(ymin, ymax, bb, bb + nr - 1, fb_lo, fb_hi);
          if (fa_lo <= fa_hi && fb_lo <= fb_hi) {
            const float area0 = face_area_eps(v, P.keps);
            const int na = fa_hi - fa_lo + 1, nb = fb_hi - fb_lo + 1, ncand = na * nb;
            const bool use_mask = na <= 8 && nb <= 8;
            // Orientation: with a negative area every edge value and the area are negated (exact, and
            // e / area == (-e) / (-area) bit for bit), so the pre-test below is "all edge values > 0"
            // for either winding.  odd: zero / NaN area, no pre-test possible.
            const bool odd = !(area0 > 0.0f || area0 < 0.0f);
            const float sg = area0 < 0.0f ? -1.0f : 1.0f;
            const float area = area0 * sg;
            // edge(p; a, b) = (p.x - a.x)(b.y - a.y) - (p.y - a.y)(b.x - a.x): the per-face and per-row factors
            const float dy0 = (v[7] - v[4]) * sg, dx0 = (v[6] - v[3]) * sg;
            const float dy1 = (v[1] - v[7]) * sg, dx1 = (v[0] - v[6]) * sg;
            const float dy2 = (v[4] - v[1]) * sg, dx2 = (v[3] - v[0]) * sg;
            const FastDiv fd(area);
            // |edge value| <= 4 M (1 + M) for pixel centres in [-1, 1] and |vertex x, y| <= M: with M <= 2^18
            // only the lower end of FastDiv's safe range is left to check per candidate.
            const float mabs = fmaxf(fmaxf(fabsf(xmin), fabsf(xmax)), fmaxf(fabsf(ymin), fabsf(ymax)));
            const float thr = (fd.d_safe && mabs <= 0x1p18f && !odd) ? 0x1p-40f : INFINITY;
            int a = fa_lo, b = fb_lo, bit = 0;
            float py = L.ndc[b];
            float r0 = (py - v[4]) * dx0, r1 = (py - v[7]) * dx1, r2 = (py - v[1]) * dx2;
            int ha_lo = 0x7fff, ha_hi = -1, hb_lo = 0x7fff, hb_hi = -1;
            u64 fmask = 0;
            u64 *zrow = L.zb + ((b - bb) * bw - a_lo);
            for (int c = 0; c < ncand; ++c) {
              const float px = L.ndc[a];
              const float e0 = (px - v[3]) * dy0 - r0;
              const float e1 = (px - v[6]) * dy1 - r1;
              const float e2 = (px - v[0]) * dy2 - r2;
              // the quotients can only all be > 0 if every (oriented) edge value is; min3 skips NaNs, which
              // the explicit comparisons on the quotients below reject like the reference does
              const float m = PERSP ? fminf(fminf(fabsf(e0), fabsf(e1)), fabsf(e2)) : fminf(fminf(e0, e1), e2);
              if (PERSP || odd || m > 0.0f) {
                float w0, w1, w2;
                if (m >= thr) {
                  w0 = fd.div(e0); w1 = fd.div(e1); w2 = fd.div(e2);
                } else {
                  w0 = e0 / area; w1 = e1 / area; w2 = e2 / area;
                }
                float b0 = w0, b1 = w1, b2 = w2;
                if (PERSP) {
                  const float t0 = (w0 * v[5]) * v[8];
                  const float t1 = (v[2] * w1) * v[8];
                  const float t2 = (v[2] * v[5]) * w2;
                  const float den = (t0 + t1) + t2;
                  b0 = t0 / den; b1 = t1 / den; b2 = t2 / den;
                }
                const float pz = (b0 * v[2] + b1 * v[5]) + b2 * v[8];
                if (!(pz < 0.0f) && b0 > 0.0f && b1 > 0.0f && b2 > 0.0f) {
                  const u64 key = ((u64)__float_as_uint(pz + 0.0f) << 32) | (u64)(uint32_t)f;  // -0 -> +0: key order = float order
                  atomicMin(&zrow[a], key);
                  fmask |= 1ull << (bit & 63);
                  if (__builtin_expect(!use_mask, 0)) {
                    ha_lo = min(ha_lo, a); ha_hi = max(ha_hi, a);
                    hb_lo = min(hb_lo, b); hb_hi = max(hb_hi, b);
                  }
                }
              }
              ++a; ++bit;
              if (a > fa_hi) {  // next row of the candidate box (the NDC table has two spare entries past S - 1)
                a = fa_lo; ++b; bit += 8 - na;
                zrow += bw;
                py = L.ndc[b];
                r0 = (py - v[4]) * dx0; r1 = (py - v[7]) * dx1; r2 = (py - v[1]) * dx2;
              }
            }
            if (use_mask) {
              rec = u4v{(uint32_t)fa_lo | ((uint32_t)fa_hi << 16), (uint32_t)fb_lo | ((uint32_t)fb_hi << 16) | (1u << 31),
                          (uint32_t)fmask, (uint32_t)(fmask >> 32)};
            } else {
              if (ha_hi < 0) { ha_lo = hb_lo = 1; ha_hi = hb_hi = 0; }
              rec = u4v{(uint32_t)ha_lo | ((uint32_t)ha_hi << 16), (uint32_t)hb_lo | ((uint32_t)hb_hi << 16), 0u, 0u};
            }
          }
        }
        finfo_put<BIG>(L, f, rec);
      }
      __syncthreads();
      STAMP(2);
      // ---- 2c. face-major list of the covered pixels: ordered by winning face, then
      // row-major.  Thread t owns the contiguous face run [t*fpl, (t+1)*fpl), so a
      // block-wide exclusive scan of the per-thread counts gives every thread its slot
      // range; the order (hence every later float sum) is deterministic.  An entry
      // packs (face id << 16) | window pixel index, all the score phases need.
      // Masked faces: the count pass visits only the pixels where the face produced a fragment and
      // leaves the mask of those it won in the record; the fill pass visits only the winners.
      const uint32_t *zlo = (const uint32_t *)L.zb;  // low dword of a key = face id
      const int fpl = (F + NT - 1) / NT;
      const int fbeg = min(F, tid * fpl), fend = min(F, fbeg + fpl);
      auto zface = [&](int a, int b) { return zlo[2 * ((b - bb) * bw + (a - a_lo))]; };
      auto pack = [&](int f, int a, int b) {
        return ((uint32_t)f << 16) | (uint32_t)((S - 1 - b - P.r0) * P.w + (S - 1 - a - P.c0));
      };
      int cnt = 0;
      for (int f = fbeg; f < fend; ++f) {
        const u4v fi = finfo_at<BIG>(L, f);
        const int fa_lo = fi.x & 0xffffu, fa_hi = fi.x >> 16, fb_lo = fi.y & 0xffffu, fb_hi = (fi.y >> 16) & 0x7fffu;
        if (fi.y >> 31) {
          u64 m = (u64)fi.z | ((u64)fi.w << 32), won = 0;
          if (m == 0) continue;
          while (m) {
            const int c = __builtin_ctzll(m);
            m &= m - 1;
            if (zface(fa_lo + (c & 7), fb_lo + (c >> 3)) == (uint32_t)f) won |= 1ull << c;
          }
          cnt += __builtin_popcountll(won);
          finfo_put_mask<BIG>(L, f, won);
        } else {
          for (int b = fb_lo; b <= fb_hi; ++b)
            for (int a = fa_lo; a <= fa_hi; ++a) cnt += (int)(zface(a, b) == (uint32_t)f);
        }
      }
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
      }
      int *ired = (int *)L.red;
      if (lane == 63) ired[wave * RED_N + 8] = incl;
      __syncthreads();
      int slot = incl - cnt, ntot = 0;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int t = ired[k * RED_N + 8];
        slot += (k < wave) ? t : 0;
        ntot += t;
      }
      ntot = __builtin_amdgcn_readfirstlane(ntot);
      if (cnt > 0)
        for (int f = fbeg; f < fend; ++f) {
          const u4v fi = finfo_at<BIG>(L, f);
          const int fa_lo = fi.x & 0xffffu, fa_hi = fi.x >> 16, fb_lo = fi.y & 0xffffu, fb_hi = (fi.y >> 16) & 0x7fffu;
          if (fi.y >> 31) {
            u64 m = (u64)fi.z | ((u64)fi.w << 32);  // winners, row-major (bit order = b, then a)
            while (m) {
              const int c = __builtin_ctzll(m);
              m &= m - 1;
              L.list[slot++] = pack(f, fa_lo + (c & 7), fb_lo + (c >> 3));
            }
          } else {
            for (int b = fb_lo; b <= fb_hi; ++b)
              for (int a = fa_lo; a <= fa_hi; ++a)
                if (zface(a, b) == (uint32_t)f) L.list[slot++] = pack(f, a, b);
          }
        }
      ncov += (tid == 0) ? ntot : 0;
      __syncthreads();
      STAMP(3);
      consume(ntot, ebase, bb + rows_per_band > b_hi);
      ebase += ntot;
    }
  }

  // ---- 4. totals -> L.ust[U_TOT..]: lane i of wave 0 holds total i; thread 0 collects them (readlane), rotates the
  // gradient (dL/dR = R * sum((view - T)^T q)) and publishes all 14.  TAIL_SYNC = false (solve kernel): no closing
  // barrier -- thread 0 itself goes on to the camera backward and Adam, the caller's own barrier publishes.
  if (SCORE) {
    if (wave == 0) {
      float tv[13];
#pragma unroll
      for (int i = 0; i < 13; ++i) tv[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), i));
      if (tid == 0) {
        L.ust[U_TOT + T_LOSS] = tv[0];
        L.ust[U_TOT + T_NCOV] = (float)ncov;
        if (GRAD) {
          const float *R = L.ust + U_R;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              L.ust[U_TOT + T_GR + 3 * i + j] = (R[3 * i] * tv[1 + j] + R[3 * i + 1] * tv[4 + j]) + R[3 * i + 2] * tv[7 + j];
#pragma unroll
          for (int i = 0; i < 3; ++i) L.ust[U_TOT + T_GT + i] = tv[10 + i];
        }
      }
    }
  } else if (tid == 0) {
    L.ust[U_TOT + T_NCOV] = (float)ncov;
  }
  if (TAIL_SYNC) __syncthreads();
  STAMP(6);
}

// Per-image loss constant: L = K - inv_hw * sum_cov(max(obj,clu) - max(0,clu)),
// K = 1 + mean(clu) - mean(max(0,clu))   (loss_fun, code/MeshPoseSolveAll.py:58-59).
// Thread 0 leaves it in L.ust[U_K].
__device__ void loss_constant(const KParams &P, const Lds &L, const float *__restrict__ clu) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!(P.clutter && clu)) {
    if (tid == 0) L.ust[U_K] = 1.0f;
    __syncthreads();
    return;
  }
  double a = 0.0;
  const gf32p gclu = as_global(clu);
  for (int i = tid; i < P.h * P.w; i += NT) a += (double)fminf(gclu[i], 0.0f);  // clu - max(0,clu)
  float d = wave_sum((float)a);
  if (lane == 0) L.red[wave * RED_N] = d;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) t += L.red[k * RED_N];
    L.ust[U_K] = 1.0f + t * P.inv_hw;
  }
  __syncthreads();
}

// Per-kernel LDS set-up: face vertex ids as 8-byte records and the table of pixel-centre
// NDC coordinates (the exact PixToNdc values, computed once instead of per pixel test).
template <bool BIG>
__device__ __forceinline__ void load_world_verts(const KParams &P, const Lds &L, const MeshDesc &M) {
  const gi32p gf = as_global(M.faces);
  const gf32p gv = as_global(M.verts);
  const uint8_t GAS *so = as_global(M.slot_order);
  for (int i = threadIdx.x; i < M.F; i += NT)
    fc_put<BIG>(L, i, us4{(unsigned short)gf[3 * i], (unsigned short)gf[3 * i + 1], (unsigned short)gf[3 * i + 2],
                          (unsigned short)(M.slot_order ? so[i] : 0x24)});
  for (int i = (int)threadIdx.x - 2; i < P.S + 2; i += NT) L.ndc[i] = ndc_of(i, (float)P.S);
  __syncthreads();
}

// Thread-0 helpers working on the uniform state in LDS.
static_assert(sizeof(Cam) == 35 * sizeof(float) && U_CAM + 35 <= UST_N, "camera slot");
__device__ __forceinline__ void ust_set_camera(const Lds &L, const float *Cp, float theta, Cam &cam,
                                               const float *extra = nullptr) {
  dev_camera(Cp, theta, extra, cam);
#pragma unroll
  for (int i = 0; i < 9; ++i) L.ust[U_R + i] = cam.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) L.ust[U_T + i] = cam.T[i];
  const float *cf = (const float *)&cam;  // kept for the camera backward of this iterate
#pragma unroll
  for (int i = 0; i < 35; ++i) L.ust[U_CAM + i] = cf[i];
}
__device__ __forceinline__ void ust_get_camera(const Lds &L, Cam &cam) {
  float *cf = (float *)&cam;
#pragma unroll
  for (int i = 0; i < 35; ++i) cf[i] = L.ust[U_CAM + i];
}
__device__ __forceinline__ void ust_totals_to_grad(const Lds &L, const Cam &cam, const float *Cp, float g[4]) {
  float GR[9], GT[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) GR[i] = block_total(L, T_GR + i);
#pragma unroll
  for (int i = 0; i < 3; ++i) GT[i] = block_total(L, T_GT + i);
  dev_camera_bwd(cam, Cp, GR, GT, g, g[3]);
}

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

typedef float f16v __attribute__((ext_vector_type(16)));
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

// loss[n][g] = K_n - (1/hw) * sum over chunks (chunk order), K_n = 1 + mean(min(clu, 0)) (loss_constant).
__global__ __launch_bounds__(256) void grid_finish_kernel(int hw, float inv_hw, int N, int G, int n_chunks,
                                                          const int32_t *img_idx, const int32_t *job_slot,
                                                          const float *clu, const float *partial,
                                                          const int32_t *pre_cnt, float *losses, u64 *counter,
                                                          int clutter_on) {
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
    losses[(size_t)n * G + g] = K - inv_hw * s;
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
template <int D, bool PERSP, bool BIG>
__global__ __launch_bounds__(NT, 4) void solve_kernel(KParams P, const MeshDesc *meshes, const int32_t *img_idx,
                                                      const int32_t *mesh_idx, const float *feat, const float *clu,
                                                      const float *C0, const float *theta0, const float *trans0,
                                                      int ndof, const float *adam_tab, int steps, float beta1,
                                                      float beta2, float eps, float *pose, float *loss_out,
                                                      float *trace, u64 *counter, int stagger,
                                                      const int32_t *order) {
  const int n = order ? order[blockIdx.x] : (int)blockIdx.x;  // longest job first (solve_order_kernel)
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
  load_world_verts<BIG>(P, L, M);
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
    eval_pose<D, PERSP, true, true, false, false, BIG, false>(P, L, M, f, c, none, true, vreg);
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
      atomicAdd(&g_stamps[7], now_ - st_lead);
    }
#endif
  }
  if (lead) {
    float *po = pose + (size_t)ndof * n;
    po[0] = L.ust[U_P]; po[1] = L.ust[U_P + 1]; po[2] = L.ust[U_P + 2]; po[3] = L.ust[U_P + 3];
    if (ndof == 6) { po[4] = L.ust[U_P6]; po[5] = L.ust[U_P6 + 1]; }
    if (loss_out) loss_out[n] = last_loss;
    if (counter) atomicAdd(counter, cov_total);
  }
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
  if (need > ctx->big_cap) {
    hipFree(ctx->d_big);
    ctx->d_big = nullptr;
    ctx->big_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_big, need));
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
  if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_counter, sizeof(u64));
  if (e == hipSuccess) e = hipMemset(ctx->d_counter, 0, sizeof(u64));
  ctx->adam_cap = 4096;
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
  SET_LDS_B(interp_bwd_kernel, 128, false); SET_LDS_B(interp_bwd_kernel, 128, true);
  SET_LDS_B(interp_bwd_kernel, 256, false); SET_LDS_B(interp_bwd_kernel, 256, true);
#undef SET_LDS_B
#undef SET_LDS
  hipFuncSetAttribute((const void *)feature_head_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void *)feature_head_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
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
  ctx->env_solve_order = env_int("NEMO_SOLVE_ORDER", 1);
  ctx->env_solve_affinity = env_int("NEMO_SOLVE_AFFINITY", 1);
  ctx->env_slot_identity = env_int("NEMO_SLOT_ORDER", 1) == 0;
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
  hipFree(ctx->d_pre_maps); hipFree(ctx->d_gs_order); hipFree(ctx->d_gs_start); hipFree(ctx->d_gs_jobslot);
  hipFree(ctx->d_gs_partial); hipFree(ctx->d_gs_work); hipFree(ctx->d_big); hipFree(ctx->d_so);
  for (int i = 0; i < ctx->n_meshes; ++i) hipFree(ctx->d_slot_order[i]);
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
  if ((rc = reserve_big_scratch(ctx, 0)) != NEMO_OK) return rc;  // the slice size may have grown: re-apply the reservation
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
  const int32_t *d_work = nullptr;
  if (mfma) {
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
                       ctx->d_gs_jobslot, d_clu, ctx->d_gs_partial, ctx->d_pre_cnt, d_losses, ctx->d_counter, con);
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
  hipFree(ctx->d_pre_maps);
  free_block_scratch(ctx, false);
  ctx->pre_block = 0;
  ctx->d_pre_slot = ctx->d_pre_cnt = ctx->d_pre_pairs = nullptr;
  ctx->d_pre_list = nullptr;
  ctx->d_pre_maps = nullptr;
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
  if (n_pairs <= GP_MAX_PAIRS) {
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

// Shared by nemo_solve / nemo_solve6 / nemo_solve_static: launch order (longest job first), then the fused solve.
// `ws` = int32 workspace of 2 * ws_cap entries (cost[ws_cap], order[ws_cap]) -- the context's reservation or the
// static device array; never allocated here.
static int launch_solve(nemo_ctx *ctx, const char *who, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx,
                        const float *d_feat, const float *d_clu, const float *d_C0, const float *d_theta0,
                        const float *d_trans0, int ndof, float *d_pose, float *d_loss, float *d_trace, void *stream,
                        int32_t *ws, size_t ws_cap) {
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
  DISPATCH_DP(solve_kernel, dim3(N), ctx->P, ctx->d_meshes, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, d_trans0,
              ndof, ctx->d_adam, ctx->adam_steps, ctx->beta1, ctx->beta2, ctx->eps, d_pose, d_loss, d_trace, ctx->d_counter,
              stagger, d_order);
  HIP_TRY(ctx, hipGetLastError());
  return NEMO_OK;
}

int nemo_solve(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx, const float *d_feat,
               const float *d_clu, const float *d_C0, const float *d_theta0, float *d_pose, float *d_loss,
               float *d_trace, void *stream) {
  return launch_solve(ctx, "nemo_solve", N, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, nullptr, 4, d_pose,
                      d_loss, d_trace, stream, ctx ? ctx->d_so : nullptr, ctx ? ctx->so_cap : 0);
}

int nemo_solve_static(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx, const float *d_feat,
                      const float *d_clu, const float *d_C0, const float *d_theta0, float *d_pose, float *d_loss,
                      float *d_trace, void *stream) {
  if (!ctx) return fail(nullptr, NEMO_ERR_ARG, "nemo_solve_static: null ctx");
  if (N > STATIC_MAX_JOBS)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_solve_static: %d jobs, the static workspace holds %d", N, STATIC_MAX_JOBS);
  if (ctx->P.big)
    return fail(ctx, NEMO_ERR_UNSUPPORTED, "nemo_solve_static: large meshes need per-workgroup table scratch "
                "(use nemo_workspace_reserve + nemo_solve)");
  int32_t *ws = (ctx->device >= 0 && ctx->device < MAX_DEVICES) ? g_static_ws[ctx->device] : nullptr;
  if (!ws) return fail(ctx, NEMO_ERR_STATE, "nemo_solve_static: no static workspace on device %d", ctx->device);
  return launch_solve(ctx, "nemo_solve_static", N, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, nullptr, 4, d_pose,
                      d_loss, d_trace, stream, ws, (size_t)STATIC_MAX_JOBS);
}

int nemo_solve6(nemo_ctx *ctx, int32_t N, const int32_t *d_img_idx, const int32_t *d_mesh_idx, const float *d_feat,
                const float *d_clu, const float *d_C0, const float *d_theta0, const float *d_trans0, float *d_pose6,
                float *d_loss, float *d_trace, void *stream) {
  return launch_solve(ctx, "nemo_solve6", N, d_img_idx, d_mesh_idx, d_feat, d_clu, d_C0, d_theta0, d_trans0, 6, d_pose6,
                      d_loss, d_trace, stream, ctx ? ctx->d_so : nullptr, ctx ? ctx->so_cap : 0);
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
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_counter, 0, sizeof(u64), (hipStream_t)stream));
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

}  // extern "C"
