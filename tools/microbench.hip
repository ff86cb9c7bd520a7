// Hardware facts the SSG kernels are designed against (run on the MI355X box):
//   1. fp32 VALU throughput: v_fma_f32 vs v_pk_fma_f32 (is packed math a lever on gfx950?)
//   2. fp32 global atomic-add throughput for the backward's scatter patterns
//   3. ds_read_b32 LDS throughput next to FMAs
// hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float a, float b) {
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_pkfma(float *out, int iters, float a, float b) {
  float2v acc[NACC];
  const float2v av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = float2v{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// sub + fma pairs with one LDS dword per `reuse` pairs
template <int REUSE>
__global__ __launch_bounds__(256) void k_lds_fma(float *out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = i;
  int idx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const float v = lds[(idx + it * 67) & 4095];
#pragma unroll
    for (int r = 0; r < REUSE; ++r) {
      const float d = acc[(r + 1) & 7] * 0.5f - v;
      acc[r & 7] = __builtin_fmaf(d, d, acc[r & 7]);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 0: wave-contiguous atomics (lane i -> base+i); mode 1: 5-float row segments of a 25x25 tile
// scattered in a 256x256x3 image (the backward kernel's pattern); mode 2: plain stores (reference)
__global__ __launch_bounds__(128) void k_atomic(float *img, int mode, int reps, unsigned seed) {
  const int tid = threadIdx.x;
  unsigned h = (blockIdx.x * 2654435761u) ^ seed;
  for (int r = 0; r < reps; ++r) {
    h = h * 1664525u + 1013904223u;
    const int y0 = (h >> 8) % 230, x0 = (h >> 20) % 230, c = r % 3;
    if (mode == 0) {
      unsafeAtomicAdd(img + (c * 256 + y0 + (tid >> 6)) * 256 + (x0 & ~63) % 192 + (tid & 63), 1.0f);
    } else {
      const int m = tid % 25, by = m / 5, bx = m % 5, j = (tid / 25);
      const int yy = y0 + by * 5 + (r % 5), xx = x0 + bx * 5 + (j % 5);
      if (mode == 1) unsafeAtomicAdd(img + (c * 256 + yy) * 256 + xx, 1.0f);
      else img[(c * 256 + yy) * 256 + xx] = 1.0f;
    }
  }
}

template <class F>
static float timeit(F f, int n = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / n;
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 8192 * sizeof(float));
  const int grid = 256 * 8, iters = 4096;
  {
    float ms = timeit([&] { hipLaunchKernelGGL(k_fma<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
    printf("v_fma_f32     16 acc: %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * grid * 256 * 16.0 * iters / ms / 1e9);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(k_pkfma<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
    printf("v_pk_fma_f32   8x2 acc: %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * grid * 256 * 16.0 * iters / ms / 1e9);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(k_lds_fma<2>, dim3(grid), dim3(256), 0, 0, out, iters); });
    printf("lds+2x(sub,fma): %.3f ms  %.2f Tpairs/s  %.2f T lds-dwords/s\n", ms, grid * 256 * 2.0 * iters / ms / 1e9, grid * 256.0 * iters / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_lds_fma<8>, dim3(grid), dim3(256), 0, 0, out, iters); });
    printf("lds+8x(sub,fma): %.3f ms  %.2f Tpairs/s  %.2f T lds-dwords/s\n", ms, grid * 256 * 8.0 * iters / ms / 1e9, grid * 256.0 * iters / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_lds_fma<16>, dim3(grid), dim3(256), 0, 0, out, iters); });
    printf("lds+16x(sub,fma): %.3f ms  %.2f Tpairs/s  %.2f T lds-dwords/s\n", ms, grid * 256 * 16.0 * iters / ms / 1e9, grid * 256.0 * iters / ms / 1e9);
  }
  float *img;
  hipMalloc(&img, 16 * 3 * 256 * 256 * sizeof(float));
  hipMemset(img, 0, 16 * 3 * 256 * 256 * sizeof(float));
  for (int mode = 0; mode < 3; ++mode) {
    const int g2 = 16384, reps = 75;
    float ms = timeit([&] { hipLaunchKernelGGL(k_atomic, dim3(g2), dim3(128), 0, 0, img, mode, reps, 12345u); });
    printf("atomic mode %d: %.3f ms  %.1f G lane-ops/s\n", mode, ms, (double)g2 * 128 * reps / ms / 1e6);
  }
  return 0;
}
