// fastdiv.h -- exact fp32 division by a loop-invariant denominator.
//
// The rasteriser divides three edge values by the same face area for every candidate pixel
// (PyTorch3D: w_k = e_k / area) and the quotients must be the correctly rounded
// IEEE ones, because pz = sum w_k z_k decides pix_to_face.  The compiler's IEEE division is
//   v_div_scale x2, v_rcp, 2 fma (reciprocal refinement), mul, 4 fma (two quotient corrections),
//   v_div_fmas, v_div_fixup;
// when neither operand needs scaling (v_div_scale leaves both untouched) and no special value is
// involved, the result is exactly  fma(e1, r, q1)  of the chain below.  FastDiv hoists the
// reciprocal refinement out of the candidate loop and runs that same chain, which is therefore
// bit-identical to `n / d` on the range it declares safe; everything else takes the IEEE path.
// tools/fastdiv_check.hip compares the two on the GPU over billions of operand pairs.
#pragma once

struct FastDiv {
  float d, r;
  bool d_safe;
  // |x| in [2^-40, 2^40]: far inside the range where v_div_scale_f32 is the identity
  // (no denormal operand, reciprocal or quotient; numerator exponent well above 23).
  static __device__ __forceinline__ bool in_range(float ax) { return ax >= 0x1p-40f && ax <= 0x1p40f; }
  __device__ __forceinline__ explicit FastDiv(float d_) : d(d_) {
    d_safe = in_range(__builtin_fabsf(d_));
    const float r0 = __builtin_amdgcn_rcpf(d_);
    const float e = __builtin_fmaf(-d_, r0, 1.0f);
    r = __builtin_fmaf(e, r0, r0);
  }
  // n / d for |n| in the safe range (caller checks); same operation chain as the IEEE expansion.
  __device__ __forceinline__ float div(float n) const {
    const float q0 = n * r;
    const float e0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e1, r, q1);
  }
};
