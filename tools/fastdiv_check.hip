// Checks FastDiv (nemo_amd/csrc/fastdiv.h) against the compiler's IEEE fp32 division on the GPU:
// every thread walks a stream of (numerator, denominator) bit patterns inside the declared safe range
// (random mantissas and exponents, plus the mantissa corner cases 0x000000, 0x7fffff, 0x400000 +- 1 for
// either operand) and counts quotients that differ in any bit from the compiler's division or from the
// quotient computed in double precision and rounded to float.  Prints the number of pairs and mismatches;
// exit status 1 on any mismatch.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gpurun_out/fastdiv_check tools/fastdiv_check.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../nemo_amd/csrc/fastdiv.h"

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// a float with |x| in [2^-40, 2^40) from 32 random bits; `corner` selects special mantissas
__device__ __forceinline__ float make(uint32_t bits, uint32_t corner) {
  uint32_t man = bits & 0x7fffffu;
  switch (corner & 7u) {
    case 1: man = 0; break;
    case 2: man = 0x7fffffu; break;
    case 3: man = 0x400000u; break;
    case 4: man = 0x3fffffu; break;
    case 5: man = 0x400001u; break;
    case 6: man = 1u; break;
    default: break;
  }
  const uint32_t ex = 127u - 40u + ((bits >> 23) & 0xffu) % 80u;
  return __uint_as_float((bits & 0x80000000u) | (ex << 23) | man);
}

__global__ void check(unsigned long long *bad, uint32_t seed, int inner, int denoms) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t s = mix(tid * 0x9E3779B9u + seed);
  unsigned long long nbad = 0;
  for (int k = 0; k < denoms; ++k) {
    s = mix(s + 0x632BE5ABu);
    const uint32_t cs = mix(s ^ 0xdeadbeefu);
    const float d = make(s, (cs & 0x30u) ? 0u : cs);  // one denominator in four gets a corner mantissa
    const FastDiv fd(d);
    if (!fd.d_safe) { ++nbad; continue; }
    for (int i = 0; i < inner; ++i) {
      s = mix(s + 0x9E3779B9u);
      const uint32_t cn = mix(s ^ 0x12345678u);
      float n = make(s, (cn & 0x30u) ? 0u : cn);
      // half of the numerators are kept close to the denominator in magnitude (barycentric range)
      if (cn & 0x100u) n = __uint_as_float((__float_as_uint(n) & 0x807fffffu) | ((__float_as_uint(d) & 0x7f800000u) - (((cn >> 9) & 7u) << 23)));
      if (!FastDiv::in_range(__builtin_fabsf(n))) continue;
      // b: the compiler's IEEE expansion; c: the quotient rounded from double precision (53 >= 2 * 24 + 2
      // bits, so the second rounding cannot change the result): an independent statement of the correctly
      // rounded fp32 quotient
      const float a = fd.div(n), b = n / d, c = (float)((double)n / (double)d);
      if (__float_as_uint(a) != __float_as_uint(b) || __float_as_uint(a) != __float_as_uint(c)) ++nbad;
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char **argv) {
  const int blocks = 4096, threads = 256, inner = 64, denoms = argc > 1 ? atoi(argv[1]) : 64;
  unsigned long long *d_bad, h_bad = 0, total_bad = 0;
  if (hipMalloc(&d_bad, sizeof(h_bad)) != hipSuccess) return 2;
  unsigned long long pairs = 0;
  for (uint32_t rep = 0; rep < 8; ++rep) {
    hipMemset(d_bad, 0, sizeof(h_bad));
    hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, 0, d_bad, 0xC0FFEEu + rep * 977u, inner, denoms);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(&h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost);
    total_bad += h_bad;
    pairs += (unsigned long long)blocks * threads * inner * denoms;
  }
  printf("fastdiv_check: %llu operand pairs (upper bound), %llu mismatches\n", pairs, total_bad);
  return total_bad ? 1 : 0;
}
