// Micro-benchmark (round 5): does the memory system take the SSG rows faster when a workgroup writes LONGER runs per
// row and hand-over?  The materialising k_s 49 row pass writes, per hand-over, one run per (pixel, image) into rows that
// are 9,604 bytes apart: 392-byte runs with 64 pixels per workgroup (shipped), 784-byte runs with 32 pixels.  Here:
// 2 x 262,144 rows of 2,401 floats (5.04 GB) written as 64-byte-aligned dwordx4 nt stores, 8 lanes per 128 bytes, by
// 512-thread workgroups that own PX consecutive rows and advance RUN floats per hand-over; optionally the same volume
// is read (coalesced 256-byte runs, nt) beside the writes as the kernel does.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_row_runs.hip -o /tmp/row_runs && /tmp/row_runs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int P = 2401;
template <int PX, int RUN, bool READ>
__global__ __launch_bounds__(512) void k(float *out_a, float *out_b, const float *src, int nrows) {
  const int row0 = blockIdx.x * PX;
  const int pair = threadIdx.x >> 3, sj = threadIdx.x & 7;      // 64 (row, image) pairs per pass
  float acc = 0.f;
  for (int q0 = 0; q0 < P; q0 += RUN) {
    if (READ) {   // the hand-over's input: PX * RUN * 2 floats, coalesced
      const float *s = src + ((size_t)blockIdx.x * P + q0) * PX * 2;
      for (int i = threadIdx.x; i < PX * RUN * 2; i += 512) acc += __builtin_nontemporal_load(s + i);
    }
    for (int pp = pair; pp < 2 * PX; pp += 64) {
      const int row = row0 + (pp >> 1);
      float *o = (pp & 1) ? out_b : out_a;
      const size_t base = (size_t)row * P + q0;
      const size_t a0 = (base + 15) & ~(size_t)15, a1 = (base + (q0 + RUN < P ? RUN : P - q0)) & ~(size_t)15;
      for (size_t g = a0 + 4 * sj; g + 4 <= a1; g += 32) {
        f4 v = {acc, acc, acc, acc};
        __builtin_nontemporal_store(v, (f4 *)(o + g));
      }
    }
  }
  if (acc == 123.456f) out_a[0] = acc;
}
template <int PX, int RUN, bool READ>
static float run(float *a, float *b, const float *src, int nrows) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<PX, RUN, READ><<<nrows / PX, 512>>>(a, b, src, nrows);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) k<PX, RUN, READ><<<nrows / PX, 512>>>(a, b, src, nrows);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}
int main() {
  const int nrows = 262144;
  float *a, *b, *src;
  const size_t n = (size_t)nrows * P + 64;
  if (hipMalloc(&a, n * 4) || hipMalloc(&b, n * 4) || hipMalloc(&src, 2 * n * 4)) return 1;
  hipMemset(src, 0, 2 * n * 4);
  printf("2 x %d rows x 2401 floats = %.2f GB written per launch (ms; write TB/s)\n", nrows, 2.0 * nrows * P * 4 / 1e9);
  const double gb = 2.0 * nrows * P * 4 / 1e9;
  float t;
  t = run<64, 98, false>(a, b, src, nrows);  printf("write only   64 px x  98 floats (392 B runs): %.3f (%.2f)\n", t, gb / t);
  t = run<32, 196, false>(a, b, src, nrows); printf("write only   32 px x 196 floats (784 B runs): %.3f (%.2f)\n", t, gb / t);
  t = run<16, 392, false>(a, b, src, nrows); printf("write only   16 px x 392 floats (1568 B runs): %.3f (%.2f)\n", t, gb / t);
  t = run<64, 98, true>(a, b, src, nrows);   printf("read + write 64 px x  98: %.3f (%.2f total)\n", t, 2 * gb / t);
  t = run<32, 196, true>(a, b, src, nrows);  printf("read + write 32 px x 196: %.3f (%.2f total)\n", t, 2 * gb / t);
  t = run<16, 392, true>(a, b, src, nrows);  printf("read + write 16 px x 392: %.3f (%.2f total)\n", t, 2 * gb / t);
  return 0;
}
