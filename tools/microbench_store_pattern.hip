// Micro-benchmark: how fast does the memory system absorb the SSG rows of a (49,13) forward, as a function of how
// many consecutive offsets a lane writes at a time?  480 workgroups x 192 lanes, every lane owns 6 rows of `pitch`
// floats (the strip kernel's ownership: 32 consecutive lanes = 32 consecutive rows) and walks q = 0 .. 2400, storing
// S floats per row every S steps; `spin` ALU iterations per step stand for the compute.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int S>
__global__ __launch_bounds__(192) void rows_kernel(float *out, int pitch, int spin, int nrows_wg) {
  const int tid = threadIdx.x;
  float acc = (float)tid;
  size_t row[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) row[j] = ((size_t)blockIdx.x * nrows_wg + j * 192 + tid) * pitch;
  for (int q0 = 0; q0 + S <= 2401; q0 += S) {
    for (int s = 0; s < S * spin; ++s) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float *o = out + row[j] + q0;
#pragma unroll
      for (int t = 0; t + 4 <= S; t += 4) {
        float4 v = make_float4(acc, acc + 1, acc + 2, acc + 3);
        __builtin_memcpy(o + t, &v, 16);
      }
#pragma unroll
      for (int t = S & ~3; t < S; ++t) o[t] = acc;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int S>
static float run(float *out, int pitch, int spin) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  rows_kernel<S><<<480, 192>>>(out, pitch, spin, 1152);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 3; ++i) rows_kernel<S><<<480, 192>>>(out, pitch, spin, 1152);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 3;
}

int main() {
  const size_t rows = 480 * 1152;
  float *out;
  if (hipMalloc(&out, rows * 2432 * sizeof(float)) != hipSuccess) return 1;
  const double gb = rows * 2401.0 * 4 / 1e9;
  printf("rows %zu, %.2f GB per launch\n", rows, gb);
  for (int pitch : {2401, 2432}) {
    for (int spin : {0, 160}) {
      printf("pitch %d spin %3d:", pitch, spin);
      printf("  S=4 %.3f ms", run<4>(out, pitch, spin));
      printf("  S=8 %.3f", run<8>(out, pitch, spin));
      printf("  S=16 %.3f", run<16>(out, pitch, spin));
      printf("  S=32 %.3f", run<32>(out, pitch, spin));
      printf("  S=49 %.3f\n", run<49>(out, pitch, spin));
    }
  }
  return 0;
}
