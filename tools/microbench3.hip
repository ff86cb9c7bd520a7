// LDS fp32 atomic-add rate on gfx950 (ds_add_f32) vs plain LDS read-modify-write.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(128) void k(float *out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 128) lds[i] = 0.f;
  __syncthreads();
  const int t = threadIdx.x;
  float v = 1.0f + t * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 25; ++u) {
      int a;
      if (MODE == 0 || MODE == 3) a = (t + u * 131) & 4095;                       // conflict-free, distinct per lane
      else if (MODE == 1) a = ((t % 25) / 5 * 5 + u / 5) * 33 + (t % 25 % 5) * 5 + u % 5 + (t / 25) * 40;  // backward's pattern
      else a = (u * 7) & 4095;                                                    // all lanes same address
      if (MODE == 3) lds[a] += v;  // plain RMW (racy, rate only)
      else atomicAdd(&lds[a], v);
    }
  }
  __syncthreads();
  out[blockIdx.x * 128 + t] = lds[t];
}
template <class F> static float timeit(F f) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize(); (void)hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  float *out; (void)hipMalloc(&out, 4096 * 128 * 4);
  const int grid = 256 * 8, iters = 200;
  const double ops = (double)grid * 128 * iters * 25;
  float ms;
  ms = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(128), 0, 0, out, iters); }); printf("ds_add_f32 conflict-free : %.3f ms  %.1f G lane-ops/s\n", ms, ops / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(128), 0, 0, out, iters); }); printf("ds_add_f32 bwd pattern   : %.3f ms  %.1f G lane-ops/s\n", ms, ops / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(grid), dim3(128), 0, 0, out, iters); }); printf("ds_add_f32 same address  : %.3f ms  %.1f G lane-ops/s\n", ms, ops / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(grid), dim3(128), 0, 0, out, iters); }); printf("plain lds RMW            : %.3f ms  %.1f G lane-ops/s\n", ms, ops / ms / 1e6);
  return 0;
}
