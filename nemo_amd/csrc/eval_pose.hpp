// eval_pose.hpp -- the per-job evaluation (project -> rasterise -> face-major list -> fused interpolate + score -> backward) and its set-up helpers
// Part of the single translation unit nemo_hip.hip (included from there, in order; not a stand-alone header).
#ifndef EVAL_POSE_HPP
#define EVAL_POSE_HPP

namespace {

// ---------------------------------------------------------------- the evaluation
// Whole-workgroup routine.  Camera R (row-major) and T are read from L.ust; the
// per-wave partial totals are left in L.red (see block_total) behind a barrier.
// RAW = true: `feat` holds an upstream gradient dL/d(out) [h][w][D] of the
// interpolated map and the routine back-propagates it (module path); otherwise
// `feat` is the image feature map and the loss of MeshPoseSolveAll.py:58-59 is used.
// TAB (solve kernel only): `tab` = this job's table of ALL feature-row x bank-row dot products, f32 [V][h * w]
// (bank_table_kernel).  The three dot products of a covered pixel are then three 4-byte reads instead of a 512-byte
// feature row, three bank rows and 3 D multiply-adds: the contraction does not depend on the pose, so it is done once
// per solve instead of once per Adam step.
// LAZY (TAB): the table starts EMPTY and is filled on demand, in blocks of (32 consecutive vertices) x (32 consecutive
// window pixels = one 128-byte line per vertex row).  A bitmap in LDS says which blocks exist; before a wave scores its
// pixels it computes the blocks they need that are not there yet, itself, on the matrix cores -- the instruction
// sequence of bank_table_kernel for that block, so the same bits -- stores them, waits for the stores and sets the bits.
// A step touches about an eighth of a job's blocks and the pose moves slowly: the solve computes a fraction of the
// table, inside the one kernel.  Two waves may compute the same block at the same time; they store the same values.
// TAB also turns on the ORDERED rasteriser pass (F <= 2 NT): thread slot i rasterises face L.fc[i].w instead of face i.
// The caller starts with the identity; on a step with `resort` set the pass counting-sorts the faces by the size of
// their candidate box (64 bins, LDS atomics) and deals them out so that the 64 lanes of a wave walk candidate loops of
// (almost) the same length and every wave gets one long and one short group.  A wave runs the longest loop of its lanes:
// in face order the critical wave of a cuboid runs 31-34 iterations against a mean of 17-25.  Which thread rasterises a
// face is invisible in the results (ds_min on the z-buffer; records and list are indexed by face).
template <int D, bool PERSP, bool GRAD, bool SCORE, bool RAW = false, bool PRE = false, bool BIG = false, bool TAIL_SYNC = true,
          bool TAB = false, bool LAZY = false>
__device__ void eval_pose(const KParams &P, const Lds &L, const MeshDesc &M, const float *__restrict__ feat,
                          const float *__restrict__ clu, EmitOut emit, bool zb_clean = false,
                          const float *__restrict__ vpre = nullptr, const float *__restrict__ tab = nullptr,
                          const bool resort = false) {
#ifndef NEMO_KIT
#define NEMO_KIT 2
#endif
#ifndef NEMO_PHASEA_MFMA
#define NEMO_PHASEA_MFMA 0  // 1: phase A on v_mfma_f32_16x16x4_f32 (diagnostic build; measured slower, profiles/r03_solve_kernel_variants.md)
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
  // TAB keeps no workgroup-wide list: its area hosts the per-vertex backward table (when it is large enough)
  f4 *const xv = (f4 *)L.list;
  const bool use_xv = TAB && !BIG && P.cap >= 4 * P.vmax;
  const float xv_inv_s = 1.0f / s;
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
      if constexpr (TAB) {
        // what the backward needs per vertex (view-space position minus T, 1 / z): once per vertex here instead of once per
        // covered pixel and vertex in phase B -- the same expressions, so the same bits
        if (use_xv) {
          const float iz = __builtin_amdgcn_rcpf(zv), zs = zv * xv_inv_s;
          xv[v] = f4{u * zs - T[0], t * zs - T[1], zv - T[2], iz};
        }
      }
      umin = fminf(umin, u); umax = fmaxf(umax, u);
      tmin = fminf(tmin, t); tmax = fmaxf(tmax, t);
    }
  }
  const bool ordered = TAB && !BIG && F <= 2 * NT;
  // 64 bin counters of the face sort, in the per-face record area (TAB keeps no records; zeroed ahead of the barrier below)
  int *const sort_bins = (int *)L.finfo;
  auto bin_at = [&](int b) -> int & { return sort_bins[b]; };
  if (TAB && resort && tid < 64) bin_at(tid) = 0;
  umin = wave_min(umin); umax = wave_max(umax); tmin = wave_min(tmin); tmax = wave_max(tmax);
  if (lane == 0) {
    L.red[wave * RED_N + 0] = umin; L.red[wave * RED_N + 1] = umax;
    L.red[wave * RED_N + 2] = tmin; L.red[wave * RED_N + 3] = tmax;
  }
  __syncthreads();
  {
    // the NW waves' boxes -> the workgroup's: lane (row j = value, k = wave) reads one number, maxima negated so that the
    // four rows of 16 reduce with the same four DPP minimum steps (exact: negation and min / max commute)
    static_assert(NW <= 16, "one DPP row per value");
    const int j = lane >> 4, k = min(lane & 15, NW - 1);
    float r = L.red[k * RED_N + j];
    r = (j & 1) ? -r : r;
    r = fminf(r, dpp_mov<0xB1>(r)); r = fminf(r, dpp_mov<0x4E>(r));
    r = fminf(r, dpp_mov<0x141>(r)); r = fminf(r, dpp_mov<0x140>(r));
    umin = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0));
    umax = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 16));
    tmin = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 32));
    tmax = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 48));
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
    // floor(f(vmin)), ceil(f(vmax)), brought into [-1, S] ONCE so that the six probes need no clamp of their own: an
    // estimate outside that range means an answer outside [0, S - 1] on that side, and the clip below gives the same
    // range either way (lo_true <= -1 -> clip_lo; lo_true >= S -> clip_hi + 1; likewise for hi).
    const int e = min(max(lo_idx(vmin) + 1, -1), S), ec = min(max(hi_idx(vmax_) - 1, -1), S);
    auto at = [&](int i) { return L.ndc[i]; };
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
  int tab_bw = 1, tab_a_lo = 0, tab_bb = 0, tab_wcnt = 0;  // TAB: geometry of the band whose z-buffer `consume` compacts
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
      constexpr int NP = GRAD ? 13 : 1;  // partials per chunk: loss term, dL/dR (9), dL/dT (3)
#ifdef NEMO_STAMPS
      st_score0 = __builtin_amdgcn_s_memrealtime();
#endif
      // rows are addressed as uniform base + 32-bit byte offset (feature window and bank are < 4 GiB)
      typedef const __attribute__((address_space(1))) char *gbytep;
      const gbytep fbytes = (gbytep)gfeat, bbytes = (gbytep)bank;
      // Static split: wave w owns the contiguous entries [ntot w / NW, ntot (w + 1) / NW) and walks them in
      // chunks of CH; its 13 sums stay in registers across chunks and are reduced across lanes once.
      const int seg_b = (ntot * wave) / NW, seg_e = (ntot * (wave + 1)) / NW;
      float pl = 0.f;
      float GR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float GT[3] = {0.f, 0.f, 0.f};
      int cb = seg_b;
      // ---------------- phase B for one list entry: its 13 contributions (g0, g1, g2 = the pixel's feature row against the
      // face's three bank rows, by vertex slot)
      auto phase_b = [&](const uint32_t ent_b, const float cl, const float g0, const float g1, const float g2) {
          const int p = (int)(ent_b & 0xffffu), f = (int)(ent_b >> 16);
          // window row / column of the pixel (TAB: the list word holds them, 8 bits each)
          const int pi = TAB ? (p >> 8) : (int)__umulhi((unsigned)p, P.w_magic), pj = TAB ? (p & 0xff) : p - pi * P.w;
          const int a = S - 1 - (pj + P.c0), b = S - 1 - (pi + P.r0);
          const us4 fv_ = fc_at<BIG>(L, f);
          const int i0 = fv_.x, i1 = fv_.y, i2 = fv_.z;
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
              float iz, X0, X1, X2;
              // world vertex = (view - T) R^T: accumulate against (view - T) and rotate the 3x3 total once
              // at the end (no world-space vertex table in LDS)
              if (TAB && use_xv) {
                const f4 xx = xv[vi[c]];
                X0 = xx.x; X1 = xx.y; X2 = xx.z; iz = xx.w;
              } else {
                iz = __builtin_amdgcn_rcpf(v[3 * c + 2]);
                const float zs = v[3 * c + 2] * inv_s;
                X0 = v[3 * c] * zs - T0; X1 = v[3 * c + 1] * zs - T1; X2 = v[3 * c + 2] - T2;
              }
              const float gxc = gv[3 * c] * iz, gyc = gv[3 * c + 1] * iz;
              const float gzv = gv[3 * c + 2] - (gv[3 * c] * v[3 * c] + gv[3 * c + 1] * v[3 * c + 1]) * iz;
              const float q0 = s * gxc, q1 = s * gyc, q2 = gzv;
              GR[0] = fmaf(X0, q0, GR[0]); GR[1] = fmaf(X0, q1, GR[1]); GR[2] = fmaf(X0, q2, GR[2]);
              GR[3] = fmaf(X1, q0, GR[3]); GR[4] = fmaf(X1, q1, GR[4]); GR[5] = fmaf(X1, q2, GR[5]);
              GR[6] = fmaf(X2, q0, GR[6]); GR[7] = fmaf(X2, q1, GR[7]); GR[8] = fmaf(X2, q2, GR[8]);
              GT[0] += q0; GT[1] += q1; GT[2] += q2;
            }
          }
      };
      if constexpr (TAB) {
        // TAB: there is no workgroup-wide list.  `ntot` is the number of z-buffer entries of the band; a wave owns
        // contiguous runs of entries (below) and compacts the covered ones IN PLACE (64 keys are read,
        // then their list words -- face << 16 | window row << 8 | window column, in entry order -- overwrite the low end of the wave's
        // own segment: a write never reaches an entry that is still to be read, and the LDS operations of a wave
        // complete in order) and scores them straight away: no count pass, no scan, no fill, no barrier between the
        // rasteriser's and the end of the score.  Entry order is row-major over the object's box, so every float sum
        // is still a fixed function of the z-buffer contents.
        // Balance: coverage is low in the top and bottom rows of the object's box and high in the middle, so a wave owns
        // TWO segments, sub-bands w and w + NW of 2 NW; the list of both lives in the larger one ("home", compacted
        // first), which has room for both: 2 |home| words >= |home| + |other| entries.
        auto sub_b = [&](int k) { return (ntot * k) / (2 * NW); };
        const int sA_b = sub_b(wave), sA_e = sub_b(wave + 1), sB_b = sub_b(wave + NW), sB_e = sub_b(wave + NW + 1);
        const bool home_a = (sA_e - sA_b) >= (sB_e - sB_b);
        uint32_t *const zw = (uint32_t *)L.zb;   // entry i: face id = low dword = zw[2 i]
        uint32_t *const wl = zw + 2 * (home_a ? sA_b : sB_b);      // the wave's list
        int wcnt = 0;
        const float inv_bw = 1.0f / (float)tab_bw;
        const int q64 = 64 / tab_bw, r64 = 64 - q64 * tab_bw;   // 64 entries further = q64 rows and r64 columns further
        auto compact = [&](const int zs_b, const int zs_e) {
          // row / column of this lane's first entry in the band (estimate by reciprocal, corrected: exact), then carried
          // from one group of 64 entries to the next
          int row, col;
          {
            const int i = zs_b + lane;
            row = (int)((float)i * inv_bw); col = i - row * tab_bw;
            if (col < 0) { --row; col += tab_bw; }
            if (col >= tab_bw) { ++row; col -= tab_bw; }
          }
          for (int i0 = zs_b; i0 < zs_e; i0 += 64) {
            const int i = i0 + lane;
            const uint32_t f = (i < zs_e) ? zw[2 * i] : 0xffffffffu;
            const bool cov = f != 0xffffffffu;
            const u64 bal = __ballot(cov);
            const int pre = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            const int a = tab_a_lo + col, b = tab_bb + row;
            if (cov) wl[wcnt + pre] = (f << 16) | (uint32_t)((S - 1 - b - P.r0) << 8) | (uint32_t)(S - 1 - a - P.c0);
            wcnt += __builtin_popcountll(bal);
            row += q64; col += r64;
            if (col >= tab_bw) { col -= tab_bw; ++row; }
          }
        };
        if (home_a) { compact(sA_b, sA_e); compact(sB_b, sB_e); } else { compact(sB_b, sB_e); compact(sA_b, sA_e); }
        const uint32_t tpr = table_tpr((uint32_t)P.w), hw32 = table_ld((uint32_t)P.h, (uint32_t)P.w);  // pixel patches per patch row; row length of the table
        // column of a list word's pixel in a table row, and its patch (device_common.hpp, table_ld)
        auto patch_of = [&](uint32_t ent) { return ((ent >> 10) & 63u) * tpr + ((ent & 0xffu) >> 3); };
        auto col_of = [&](uint32_t ent) { return patch_of(ent) * 32u + ((ent >> 5) & 0x18u) + (ent & 7u); };
        // LAZY: the blocks an entry reads are made to exist when the entry is FETCHED (below): one lane per entry forms the
        // (at most three) vertex-chunk bits of its face, one 8-byte LDS read says which of them exist in its patch, and
        // only when some lane misses a block does the wave leave the fetch to compute it.
        u64 *const validt = (u64 *)L.finfo + 32;   // per pixel tile: bit c = block (vertex chunk c, tile) exists; the first
        const uint32_t n_tiles = hw32 >> 5;        // 256 bytes of the area are the sort bins
        __attribute__((address_space(1))) float *const wtab = (__attribute__((address_space(1))) float *)tab;
        const int bi = lane & 31, bhalf = lane >> 5;
        auto compute_block = [&](const uint32_t bc, const uint32_t bt, const uint32_t ti, const uint32_t tj) {   // all 64 lanes; chunk and patch bt = (ti, tj) wave-uniform
          // lane (i, half) feeds feature 8 q + 4 half + c as k-step 4 q + c of both operands (bank_table_kernel's order).
          // A block is 32 bank rows (one accumulator: the register budget of a 1024-thread workgroup, and finer blocks
          // waste less: a pixel tile needs some twenty vertices).  The chain of matrix instructions per accumulator is
          // bank_table_kernel's, so the stored values are bit for bit its values.
          // lane bi of a half = position bi of the patch (row bi >> 3, column bi & 7; past the window: its edge, padding)
          const int xpix = min((int)(4u * ti) + (bi >> 3), P.h - 1) * P.w + min((int)(8u * tj) + (bi & 7), P.w - 1);
          const gf4p xp = (gf4p)(gfeat + (size_t)xpix * D) + bhalf;
          const uint32_t col = bt * 32u + (uint32_t)bi;
          {
            const gf4p ap = (gf4p)(bank + (size_t)min((int)(bc * 32u) + bi, V - 1) * D) + bhalf;
            f16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            constexpr int PF = 6;   // 16-byte pieces in flight per operand
            f4 av[PF], xv_[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) { av[u] = ap[2 * u]; xv_[u] = xp[2 * u]; }
#pragma unroll
            for (int q = 0; q < D / 8; ++q) {
              const f4 a = av[q % PF], x = xv_[q % PF];
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, x.x, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, x.y, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, x.z, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, x.w, acc, 0, 0, 0);
              if (q + PF < D / 8) { av[q % PF] = ap[2 * (q + PF)]; xv_[q % PF] = xp[2 * (q + PF)]; }
              __builtin_amdgcn_sched_barrier(0);
            }
            // C/D layout: column = lane & 31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 half; rows past V exist (padded slot)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t row = bc * 32u + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * bhalf);
              wtab[row * hw32 + col] = acc[r];
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the stores are done before the bit says so
        };
        auto ensure = [&](const uint32_t ent, const us4 fv_) {
          const uint32_t tl = patch_of(ent);
          const u64 need = (1ull << (fv_.x >> 5)) | (1ull << (fv_.y >> 5)) | (1ull << (fv_.z >> 5));
          u64 missing = need & ~validt[tl];
          u64 bal = __ballot(missing != 0ull);
          while (__builtin_expect(bal != 0ull, 0)) {
            const int src = __builtin_ctzll(bal);
            const uint32_t t_s = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)tl, src));
            const uint32_t m_lo = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)(uint32_t)missing, src));
            const uint32_t m_hi = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)(uint32_t)(missing >> 32), src));
            const uint32_t c_s = m_lo ? (uint32_t)__builtin_ctz(m_lo) : 32u + (uint32_t)__builtin_ctz(m_hi);
            const uint32_t e_s = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)ent, src));
            compute_block(c_s, t_s, (e_s >> 10) & 63u, (e_s & 0xffu) >> 3);
            if (lane == 0) {
              atomicOr((unsigned long long *)&validt[t_s], 1ull << c_s);
              atomicAdd((unsigned int *)&L.ust[U_BLK], 1u);
            }
            missing = need & ~validt[tl];
            bal = __ballot(missing != 0ull);
          }
        };
        // The dot products come from the job's table: one lane per entry, the entry's list word, clutter score and three
        // table values are fetched one chunk ahead (unconditionally, with a clamped index: exact wait counts).
        const gf32p gtab = as_global(tab);
        uint32_t ent_c = 0u;
        float cl_c = 0.f, g_c[3] = {0.f, 0.f, 0.f};
        auto fetch = [&](int cb_, uint32_t &ent, float &cl, float (&g)[3]) {
          ent = wl[min(cb_ + lane, wcnt - 1)];
          const uint32_t px = col_of(ent);
          cl = (P.clutter && clu) ? gclu[((ent >> 8) & 0xffu) * (uint32_t)P.w + (ent & 0xffu)] : 0.0f;
          const us4 fv_ = fc_at<BIG>(L, ent >> 16);
          if constexpr (LAZY) ensure(ent, fv_);   // (a clamped index repeats a real entry of the wave: nothing is missing then)
          g[0] = gtab[(uint32_t)fv_.x * hw32 + px]; g[1] = gtab[(uint32_t)fv_.y * hw32 + px]; g[2] = gtab[(uint32_t)fv_.z * hw32 + px];
        };
        int cbt = 0;
        if (wcnt > 0) fetch(0, ent_c, cl_c, g_c);
        while (cbt < wcnt) {
          const int cn = min(CH, wcnt - cbt);
          uint32_t ent_n;
          float cl_n, g_n[3];
          fetch(cbt + CH, ent_n, cl_n, g_n);
          if (lane < cn) phase_b(ent_c, cl_c, g_c[0], g_c[1], g_c[2]);
          ent_c = ent_n; cl_c = cl_n; g_c[0] = g_n[0]; g_c[1] = g_n[1]; g_c[2] = g_n[2];
          cbt += CH;
        }
        tab_wcnt = wcnt;
      }
      if constexpr (!TAB) {
#if NEMO_PHASEA_MFMA
      // wave-private scratch in the (dead) z-buffer region, WS_BYTES per wave: per-entry words, per-tile run tables,
      // the phase A -> B hand-off buffer and a dump row for the lanes that have nothing to hand off
      static_assert(CH == 64, "the matrix-core phase A works on chunks of 64 entries (4 tiles = 4 DPP rows)");
      constexpr int WS_BYTES = CH * (12 + 16);
      unsigned char *wsb = (unsigned char *)L.zb + wave * WS_BYTES;
      uint32_t *pixw = (uint32_t *)wsb;                    // [64]  (window pixel << 4) | run index inside the tile
      us4 *trow = (us4 *)(wsb + 256);                      // [4][16] vertex ids of run r of tile t
      float *gbuf = (float *)(wsb + 768);                  // [64][3] dot products by vertex slot
      float *dump = (float *)(wsb + 1536);                 // [64]
      const int pj = lane & 15, kk = lane >> 4;            // operand roles: pixel / row slot pj of the tile, k-group kk
      const int slot_run = pj / 3, slot_v = pj - 3 * slot_run;  // row slot pj = vertex slot_v of the pass's run slot_run
#else
      constexpr int GPW = 64 / LPP;
      const int gl = lane & (LPP - 1), grp = lane / LPP;
      uint4 *rec = (uint4 *)L.zb + wave * CH;                               // [NW][CH]   (the z-buffer is dead by now
      float *gbuf = (float *)((uint4 *)L.zb + NW * CH) + wave * (CH * 3);   // [NW][CH*3]  and hosts the wave-private scratch)
      const uint32_t gl16 = (uint32_t)gl * 16u;
      // byte offset of the feature row of this lane group's first entry in the chunk [cb_, cb_ + cn_)
      auto first_off_at = [&](int cb_, int cn_) {
        const int per_ = (cn_ + GPW - 1) / GPW;
        return (L.list[cb_ + min(grp * per_, cn_ - 1)] & 0xffffu) * (uint32_t)(D * 4) + gl16;
      };
      f4 xa[KIT], xb[KIT];  // ping-pong: current / next feature-row fragments
      if (cb < seg_e) {
        const gf4p Fp = (gf4p)(fbytes + first_off_at(cb, min(CH, seg_e - cb)));
#pragma unroll
        for (int k = 0; k < KIT; ++k) xa[k] = __builtin_nontemporal_load(Fp + LPP * k);
      }
#endif
      while (cb < seg_e) {
        const int cn = min(CH, seg_e - cb);
        const int cb_next = cb + CH;
        uint32_t ent_bs[CH / 64];
        float cls[CH / 64];
#if NEMO_PHASEA_MFMA
        // ---------------- records (one lane per entry; a tile = 16 consecutive entries = one DPP row).  The list is
        // face-major, so a tile is a few runs of equal face: lane e finds the run it belongs to (head flags, row
        // scan), run heads publish their three vertex ids, every lane its pixel and run index.
        int nrun_t[4];  // runs per tile (wave-uniform)
        {
          const uint32_t ent = L.list[cb + min(lane, cn - 1)];  // lanes past the end repeat the last entry
          ent_bs[0] = ent;
          cls[0] = (P.clutter && clu) ? gclu[ent & 0xffffu] : 0.0f;
          const int f = (int)(ent >> 16);
          const us4 fv_ = fc_at<BIG>(L, f);
          const int fprev = __builtin_amdgcn_update_dpp(-1, f, 0x111, 0xF, 0xF, false);  // row_shr:1; lane 0 of a row keeps -1
          const int head = (int)(f != fprev);
          int rho = head;
          rho += dpp_shr0<0x111>(rho); rho += dpp_shr0<0x112>(rho); rho += dpp_shr0<0x114>(rho); rho += dpp_shr0<0x118>(rho);
          rho -= 1;  // run index inside the tile, 0 .. 15
          if (head) trow[(lane >> 4) * 16 + rho] = fv_;
          pixw[lane] = ((ent & 0xffffu) << 4) | (uint32_t)rho;
#pragma unroll
          for (int t = 0; t < 4; ++t) nrun_t[t] = __builtin_amdgcn_readlane(rho, 16 * t + 15) + 1;
        }
        // ---------------- phase A on the matrix cores: per tile, D[row slot][pixel] = bank rows x feature rows with
        // v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).  Lane (pj, kk) feeds feature 16 q + 4 kk + c as k-step
        // 4 q + c of BOTH operands, so both are plain 16-byte loads: 8 per feature row (all 16 rows of the tile in
        // flight at once) and 8 per bank row.  A pass holds the rows of 5 runs (3 vertices each; slot 15 idle); a
        // tile with more runs takes further passes over the same feature registers.  Lane (pj, kk) ends up with the
        // products of pixel pj against row slots 4 kk .. 4 kk + 3 and hands those its pixel needs to phase B.
        {
          const int ntile = (cn + 15) >> 4;
          for (int t = 0; t < ntile; ++t) {
            const uint32_t w = pixw[16 * t + pj];
            const int nr = t == 0 ? nrun_t[0] : (t == 1 ? nrun_t[1] : (t == 2 ? nrun_t[2] : nrun_t[3]));
#ifndef NEMO_MF_LINE
#define NEMO_MF_LINE 0
#endif
#ifndef NEMO_MF_NT
#define NEMO_MF_NT 1
#endif
#ifndef NEMO_MF_ABL
#define NEMO_MF_ABL 0
#endif
            // k-step 4 q + c of a lane is feature 16 q + 4 kk + c (LINE 0: the four k-groups of a row share 64-byte
            // segments) or 32 kk + 4 q + c (LINE 1: a lane walks one 128-byte line of its row)
            constexpr int KK_B = NEMO_MF_LINE ? 128 : 16, Q_F4 = NEMO_MF_LINE ? 1 : 4;
            const uint32_t foff = (w >> 4) * (uint32_t)(D * 4) + (uint32_t)kk * KK_B;
            const int my_run = (int)(w & 15u);
            const int ent_i = 16 * t + pj;
            f4 xf[8], yb[8];
            if (D == 128) {
#pragma unroll
              for (int q = 0; q < 8; ++q) xf[q] = (NEMO_MF_ABL == 2) ? f4{1.f, 1.f, 1.f, 1.f} : (NEMO_MF_NT ? __builtin_nontemporal_load((gf4p)(fbytes + foff) + Q_F4 * q) : ((gf4p)(fbytes + foff))[Q_F4 * q]);
            }
            for (int r0 = 0; r0 < nr; r0 += 5) {
              const us4 rv = trow[t * 16 + min(r0 + slot_run, nr - 1)];
              const uint32_t vid = slot_v == 0 ? rv.x : (slot_v == 1 ? rv.y : rv.z);
              const uint32_t boff = vid * (uint32_t)(D * 4) + (uint32_t)kk * KK_B;
              f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int hh = 0; hh < D / 128; ++hh) {
                if (D != 128) {
#pragma unroll
                  for (int q = 0; q < 8; ++q) xf[q] = __builtin_nontemporal_load((gf4p)(fbytes + foff) + (NEMO_MF_LINE ? 32 * hh + q : 32 * hh + 4 * q));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) yb[q] = (NEMO_MF_ABL == 3) ? f4{1.f, 1.f, 1.f, 1.f} : ((gf4p)(bbytes + boff))[NEMO_MF_LINE ? 32 * hh + q : 32 * hh + 4 * q];
                if (NEMO_MF_ABL == 1) {
#pragma unroll
                  for (int q = 0; q < 8; ++q) acc0 += yb[q] * xf[q];
                } else
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[q].x, xf[q].x, acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[q + 1].x, xf[q + 1].x, acc1, 0, 0, 0);
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[q].y, xf[q].y, acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[q + 1].y, xf[q + 1].y, acc1, 0, 0, 0);
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[q].z, xf[q].z, acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[q + 1].z, xf[q + 1].z, acc1, 0, 0, 0);
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[q].w, xf[q].w, acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[q + 1].w, xf[q + 1].w, acc1, 0, 0, 0);
                }
              }
              const f4 acc = acc0 + acc1;
              // hand-off: row slot 4 kk + r holds vertex (4 kk + r) - 3 rel of the pixel's run when that is 0, 1 or 2
              const int rel = my_run - r0;
              const bool mine = rel >= 0 && rel < 5 && ent_i < cn;
              const int v0 = 4 * kk - 3 * rel;  // vertex index of my register 0
              float *gb = gbuf + 3 * ent_i;
              *((mine && (unsigned)(v0 + 0) < 3u) ? gb + v0 + 0 : dump + lane) = acc.x;
              *((mine && (unsigned)(v0 + 1) < 3u) ? gb + v0 + 1 : dump + lane) = acc.y;
              *((mine && (unsigned)(v0 + 2) < 3u) ? gb + v0 + 2 : dump + lane) = acc.z;
              *((mine && (unsigned)(v0 + 3) < 3u) ? gb + v0 + 3 : dump + lane) = acc.w;
            }
          }
        }
#else
        const int per = (cn + GPW - 1) / GPW;  // entries per lane group in this chunk
        // ---------------- records (one lane per entry): everything phase A needs to know about an entry
        // -- feature-row byte offset and the three bank rows -- is worked out once here, 64 entries per
        // instruction, instead of by every lane of a group inside the loop.  The same pass fetches this
        // lane's phase-B list entries and clutter scores (they land during phase A).
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
#endif
        STAMP(8);
        // ---------------- phase B: this lane's pixels -> its 13 contributions
#pragma unroll
        for (int u = 0; u < CH / 64; ++u)
        if (u * 64 + lane < cn) {
          const int lb = u * 64 + lane;
          phase_b(ent_bs[u], cls[u], gbuf[3 * lb], gbuf[3 * lb + 1], gbuf[3 * lb + 2]);
        }
        STAMP(9);
        cb = cb_next;
      }
      }  // !TAB
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
          if (TAB) L.red[wave * RED_N + 13] = (float)tab_wcnt;   // the wave's covered pixels (exact: far below 2^24)
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
      atomicAdd(&stamp_lds()[10], __builtin_amdgcn_s_memrealtime() - st_arrive);
      atomicAdd(&stamp_lds()[11], st_arrive - st_score0);
      stamp_lds()[16 + ((threadIdx.x >> 6) & 15)] += st_arrive - st_score0;
    }
#endif
    if (SCORE) {
      // the waves' partials -> running totals, in wave order: lane i of wave 0 keeps total i in a register
      constexpr int NPB = TAB ? 14 : (GRAD ? 13 : 1);   // TAB: lane 13 totals the waves' covered-pixel counts
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
      const bool sort_now = TAB && ordered && resort && bb == b_lo;
      uint32_t sk0 = 0u, sk1 = 0u;  // sort_now: (bin << 16) | rank inside the bin, and the face, of the two rounds
      uint32_t sf0 = 0u, sf1 = 0u;
      for (int fs = tid; fs < F; fs += NT) {
        int f = fs;
        if constexpr (TAB) { if (ordered) f = (int)((const unsigned short *)L.fc)[4 * fs + 3]; }
        const us4 fv_ = fc_at<BIG>(L, f);
        const int i0 = fv_.x, i1 = fv_.y, i2 = fv_.z;
        float v[9];
        load_face_verts<BIG>(L, i0, i1, i2, v);
        int ncand_f = 0;
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
            ncand_f = ncand;
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
            // One candidate (pixel centre px of column a_ in the current row): PyTorch3D's three edge values and the rest.
            auto cand = [&](const float px, const int a_, const int bit_) {
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
                  atomicMin(&zrow[a_], key);
                  if constexpr (!TAB) {   // (TAB builds no list from per-face records)
                    fmask |= 1ull << (bit_ & 63);
                    if (__builtin_expect(!use_mask, 0)) {
                      ha_lo = min(ha_lo, a_); ha_hi = max(ha_hi, a_);
                      hb_lo = min(hb_lo, b); hb_hi = max(hb_hi, b);
                    }
                  }
                }
              }
            };
            // The candidates are walked in PAIRS of neighbouring columns of a row (a row of na columns is (na + 1) / 2 pairs;
            // the second candidate of a row's last pair is switched off when na is odd): loop control, the pixel-centre
            // read and the row advance -- more than half of a rejected candidate's instructions -- are paid once per pair.
            const int npr = (na + 1) >> 1, npairs = npr * nb;
            for (int c = 0; c < npairs; ++c) {
              const float px0 = L.ndc[a], px1 = L.ndc[a + 1];   // (the NDC table has two spare entries past S - 1)
              cand(px0, a, bit);
              if (a < fa_hi) cand(px1, a + 1, bit + 1);
              a += 2; bit += 2;
              if (a > fa_hi) {  // next row of the candidate box
                a = fa_lo; ++b; bit += 8 - 2 * npr;
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
        if constexpr (!TAB) finfo_put<BIG>(L, f, rec);
        if constexpr (TAB) {
          if (sort_now) {  // every face is visited exactly once: the ranks inside a bin are a permutation of it
            const int key = min(ncand_f, 63);
            const uint32_t kr = ((uint32_t)key << 16) | (uint32_t)atomicAdd(&bin_at(key), 1);
            if (fs < NT) { sk0 = kr; sf0 = (uint32_t)f; } else { sk1 = kr; sf1 = (uint32_t)f; }
          }
        }
      }
      __syncthreads();
      if constexpr (TAB) {
        if (sort_now) {
          // every wave: start of bin b in descending key order = number of faces in the bins above it
          const int c = bin_at(63 - lane);
          const int excl = wave_scan_incl(c, lane) - c;
          auto place = [&](uint32_t kr, uint32_t f_, bool valid) {
            const int key = (int)(kr >> 16);
            // (the cross-lane read runs with every lane active: a lane that is switched off reads as zero)
            const int pos = __shfl(excl, 63 - key) + (int)(kr & 0xffffu);   // position in the sorted order
            // round 0 (slots 0 .. NT-1) takes the NT largest in order, round 1 the rest in REVERSE: the wave with the
            // longest group of the first round gets the shortest (or none) of the second
            const int slot = pos < NT ? pos : NT + (F - 1 - pos);
            if (valid) ((unsigned short *)L.fc)[4 * slot + 3] = (unsigned short)f_;
          };
          place(sk0, sf0, tid < F);
          place(sk1, sf1, tid + NT < F);
        }
      }
      STAMP(2);
      if constexpr (TAB) {
        tab_bw = bw; tab_a_lo = a_lo; tab_bb = bb;
        STAMP(3);
        consume(npx, ebase, bb + rows_per_band > b_hi);
        continue;
      }
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
      const int incl = wave_scan_incl(cnt, lane);
      int *ired = (int *)L.red;
      if (lane == 63) ired[wave * RED_N + 8] = incl;
      __syncthreads();
      // the waves' totals: lane k < NW reads wave k's, a row scan gives every wave its base and the grand total
      int wt = (lane < NW) ? ired[lane * RED_N + 8] : 0;
      wt += dpp_shr0<0x111>(wt); wt += dpp_shr0<0x112>(wt); wt += dpp_shr0<0x114>(wt); wt += dpp_shr0<0x118>(wt);
      const int ntot = __builtin_amdgcn_readlane(wt, NW - 1);
      const int slot0 = wave > 0 ? __builtin_amdgcn_readlane(wt, max(wave - 1, 0)) : 0;
      int slot = incl - cnt + slot0;
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
      const float ncov_tab = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 13));
      if (tid == 0) {
        L.ust[U_TOT + T_LOSS] = tv[0];
        L.ust[U_TOT + T_NCOV] = TAB ? ncov_tab : (float)ncov;
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

}  // namespace

#endif  // EVAL_POSE_HPP
