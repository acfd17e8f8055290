// Micro-benchmark: cost of one vector-memory wave-instruction in the texture-address path of a
// gfx950 CU, for the access shapes the score loop uses (cache-resident data, 16 waves per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ta_microbench tools/ta_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f4 *gf4p;
typedef const __attribute__((address_space(1))) f2 *gf2p;
typedef const __attribute__((address_space(1))) float *gf1p;

constexpr int ROWB = 512;      // bytes per row
constexpr int NROWS = 64;      // rows in the resident table (32 KiB)

// MODE 0: 64 lanes x 16 B contiguous (two rows)
// MODE 1: 8 groups x 8 lanes, each group a 128-B segment of its own row (feature-row shape)
// MODE 2: as 1, one group active        MODE 3: as 1, two groups active     MODE 4: as 1, four groups
// MODE 5: 64 lanes x 8 B                MODE 6: 64 lanes x 4 B
// MODE 7: 4 groups x 16 lanes x 16 B    MODE 8: as 7, one group active
// MODE 9: 2 groups x 32 lanes x 16 B    MODE 10: as 9, one group active
template <int MODE>
__global__ __launch_bounds__(1024) void k(const float *tab, float *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char *base = (const char *)tab;
  float acc = 0.f;
  int r = (wave * 7 + blockIdx.x) & (NROWS - 1);
  for (int i = 0; i < iters; ++i) {
    r = (r + 5) & (NROWS - 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) {
        f4 v = *((gf4p)(base + ((r + 2 * u) & (NROWS - 2)) * ROWB) + lane);
        acc += v.x + v.w;
      } else if (MODE >= 1 && MODE <= 4) {
        const int g = lane >> 3, gl = lane & 7;
        const int act = MODE == 1 ? 8 : MODE == 2 ? 1 : MODE == 3 ? 2 : 4;
        if (g < act) {
          f4 v = *((gf4p)(base + ((r + g * 3) & (NROWS - 1)) * ROWB) + gl + 8 * u);
          acc += v.x + v.w;
        }
      } else if (MODE == 5) {
        f2 v = *((gf2p)(base + ((r + u) & (NROWS - 1)) * ROWB) + lane);
        acc += v.x + v.y;
      } else if (MODE == 6) {
        float v = *((gf1p)(base + ((r + u) & (NROWS - 1)) * ROWB) + lane);
        acc += v;
      } else if (MODE == 7 || MODE == 8) {
        const int g = lane >> 4, gl = lane & 15;
        if (MODE == 7 || g < 1) {
          f4 v = *((gf4p)(base + ((r + g * 3) & (NROWS - 1)) * ROWB) + gl + 16 * (u & 1));
          acc += v.x + v.w;
        }
      } else {
        const int g = lane >> 5, gl = lane & 31;
        if (MODE == 9 || g < 1) {
          f4 v = *((gf4p)(base + ((r + g * 3 + u) & (NROWS - 1)) * ROWB) + gl);
          acc += v.x + v.w;
        }
      }
    }
  }
  if (acc == 123.456f) out[threadIdx.x] = acc;
}

// Streaming shape of the score loop: 4 groups x 16 lanes, two 16-byte loads per 512-byte row, rows picked
// pseudo-randomly from a table of `nrows` rows (large table: HBM; small: L2), DEPTH rows in flight per wave.
template <int DEPTH>
__global__ __launch_bounds__(1024) void ks(const float *tab, float *out, int iters, unsigned nrows_mask) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, gl = lane & 15;
  const char *base = (const char *)tab;
  unsigned r = (blockIdx.x * 16 + wave) * 4 + g;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
    f4 v[DEPTH][2];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      r = r * 1664525u + 1013904223u;
      const gf4p p = (gf4p)(base + (size_t)((r >> 8) & nrows_mask) * ROWB) + gl;
      v[u][0] = __builtin_nontemporal_load(p);
      v[u][1] = __builtin_nontemporal_load(p + 16);
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) acc += v[u][0] * v[u][1];
  }
  if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int DEPTH>
void runs(const float *tab, float *out, unsigned nrows, const char *name, int grid = 256) {
  const int iters = 4000 / DEPTH;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  ks<DEPTH><<<grid, 1024>>>(tab, out, 4, nrows - 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  ks<DEPTH><<<grid, 1024>>>(tab, out, iters, nrows - 1);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double instr_per_cu = (double)iters * DEPTH * 2 * 16;
  const double bytes = (double)grid * instr_per_cu * 1024;
  printf("stream depth %d %-22s grid %3d %8.3f ms  %6.1f ns/instr/CU (%.1f cyc)  %.0f GB/s total, %.1f GB/s/CU\n", DEPTH, name, grid, ms,
         ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4, bytes / ms / 1e6, bytes / ms / 1e6 / grid);
}

template <int MODE>
void run(const float *tab, float *out, const char *name) {
  const int iters = 2000, grid = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<grid, 1024>>>(tab, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE><<<grid, 1024>>>(tab, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double instr_per_cu = (double)iters * 4 * 16;  // wave-instructions per CU
  printf("mode %2d %-44s %8.3f ms  %6.1f ns/instr/CU  (%.1f cycles @2.4GHz)\n", MODE, name, ms,
         ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
}

int main() {
  float *tab, *out;
  hipMalloc(&tab, NROWS * ROWB);
  hipMalloc(&out, 4096);
  hipMemset(tab, 0, NROWS * ROWB);
  run<0>(tab, out, "64 lanes x16B contiguous");
  run<1>(tab, out, "8 groups x 8 lanes x16B (8 rows)");
  run<2>(tab, out, "  same, 1 of 8 groups active");
  run<3>(tab, out, "  same, 2 of 8 groups active");
  run<4>(tab, out, "  same, 4 of 8 groups active");
  run<5>(tab, out, "64 lanes x8B contiguous");
  run<6>(tab, out, "64 lanes x4B contiguous");
  run<7>(tab, out, "4 groups x 16 lanes x16B");
  run<8>(tab, out, "  same, 1 of 4 groups active");
  run<9>(tab, out, "2 groups x 32 lanes x16B");
  run<10>(tab, out, "  same, 1 of 2 groups active");
  float *big;
  const unsigned big_rows = 1u << 20;  // 512 MiB
  hipMalloc(&big, (size_t)big_rows * ROWB);
  hipMemset(big, 0, (size_t)big_rows * ROWB);
  runs<1>(big, out, big_rows, "HBM (512 MiB)");
  runs<2>(big, out, big_rows, "HBM (512 MiB)");
  runs<4>(big, out, big_rows, "HBM (512 MiB)");
  runs<8>(big, out, big_rows, "HBM (512 MiB)");
  runs<2>(big, out, big_rows, "HBM (512 MiB)", 64);
  runs<8>(big, out, big_rows, "HBM (512 MiB)", 64);
  runs<1>(big, out, 1u << 15, "MALL/L2 (16 MiB)");
  runs<2>(big, out, 1u << 15, "MALL/L2 (16 MiB)");
  runs<8>(big, out, 1u << 15, "MALL/L2 (16 MiB)");
  runs<1>(big, out, 1u << 12, "L2 (2 MiB)");
  runs<2>(big, out, 1u << 12, "L2 (2 MiB)");
  runs<8>(big, out, 1u << 12, "L2 (2 MiB)");
  return 0;
}
