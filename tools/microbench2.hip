// VALU issue-rate probes for the SSG inner loops (gfx950).  Every kernel runs NITER iterations of
// an unrolled body of independent operations on VGPR operands; results are kept live.
// Reports wave-instructions/s as "T lane-ops/s" (64 lanes per wave-instruction) and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define NACC 16
// A: v_fmac_f32 (VOP2), all-VGPR operands
__global__ __launch_bounds__(256) void kA(float *out, const float *in, int iters) {
  float acc[NACC], x[4];
  for (int i = 0; i < 4; ++i) x[i] = in[threadIdx.x + 256 * i];
  const float y = in[threadIdx.x + 1024];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(x[i & 3], y, acc[i]);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// B: (sub, fma) pairs like the forward kernel: d = a - b; acc = fma(d, d, acc)
__global__ __launch_bounds__(256) void kB(float *out, const float *in, int iters) {
  float acc[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) b[i] = in[threadIdx.x + 256 * i];
  float a = in[threadIdx.x + 5000];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const float d = a - b[i];
      acc[i] = __builtin_fmaf(d, d, acc[i]);
    }
    a += 1e-7f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// C: packed (sub, fma) pairs
__global__ __launch_bounds__(256) void kC(float *out, const float *in, int iters) {
  f2 acc[NACC / 2], b[NACC / 2];
  for (int i = 0; i < NACC / 2; ++i) b[i] = f2{in[threadIdx.x + 256 * i], in[threadIdx.x + 256 * i + 7]};
  f2 a = {in[threadIdx.x + 5000], in[threadIdx.x + 5001]};
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) acc[i] = f2{(float)i, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC / 2; ++i) {
      const f2 d = a - b[i];
      acc[i] = __builtin_elementwise_fma(d, d, acc[i]);
    }
    a += f2{1e-7f, 1e-7f};
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// C2: packed with a scalar `a` broadcast to both halves (op_sel)
__global__ __launch_bounds__(256) void kC2(float *out, const float *in, int iters) {
  f2 acc[NACC / 2], b[NACC / 2];
  for (int i = 0; i < NACC / 2; ++i) b[i] = f2{in[threadIdx.x + 256 * i], in[threadIdx.x + 256 * i + 7]};
  float a = in[threadIdx.x + 5000];
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) acc[i] = f2{(float)i, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC / 2; ++i) {
      const f2 d = f2{a, a} - b[i];
      acc[i] = __builtin_elementwise_fma(d, d, acc[i]);
    }
    a += 1e-7f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// D: packed fma only (correlation form): acc2 = fma(a2, b2, acc2)
__global__ __launch_bounds__(256) void kD(float *out, const float *in, int iters) {
  f2 acc[NACC / 2], b[NACC / 2];
  for (int i = 0; i < NACC / 2; ++i) b[i] = f2{in[threadIdx.x + 256 * i], in[threadIdx.x + 256 * i + 7]};
  f2 a = {in[threadIdx.x + 5000], in[threadIdx.x + 5001]};
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) acc[i] = f2{(float)i, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC / 2; ++i) acc[i] = __builtin_elementwise_fma(a, b[i], acc[i]);
    a += f2{1e-7f, 1e-7f};
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// E: packed (sub,fma) fed by ds_read_b64 every REUSE pairs
template <int REUSE>
__global__ __launch_bounds__(256) void kE(float *out, const float *in, int iters) {
  __shared__ f2 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = f2{in[i], in[i + 1]};
  __syncthreads();
  f2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f2{(float)i, (float)i};
  f2 a = {in[threadIdx.x + 5000], in[threadIdx.x + 5001]};
  for (int it = 0; it < iters; ++it) {
    const f2 v = lds[(threadIdx.x + it * 67) & 2047];
#pragma unroll
    for (int r = 0; r < REUSE; ++r) {
      const f2 d = a - v;
      acc[r & 7] = __builtin_elementwise_fma(d, d, acc[r & 7]);
      a += f2{0.f, 0.f};
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
static float timeit(F f, int n = 5) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < n; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / n;
}

int main() {
  float *out, *in;
  (void)hipMalloc(&out, 256 * 8192 * sizeof(float));
  (void)hipMalloc(&in, 65536 * sizeof(float));
  (void)hipMemset(in, 0, 65536 * sizeof(float));
  const int iters = 4096;
  for (int grid : {256 * 2, 256 * 8}) {  // 2 and 8 waves per SIMD
    printf("--- grid %d (%d waves/SIMD)\n", grid, grid / 256);
    const double lanes = (double)grid * 256 * iters;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(kA, dim3(grid), dim3(256), 0, 0, out, in, iters); });
    printf("A  v_fmac vgpr        : %.3f ms  %.1f T lane-ops/s  %.1f TFLOP/s\n", ms, lanes * NACC / ms / 1e9, 2 * lanes * NACC / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(kB, dim3(grid), dim3(256), 0, 0, out, in, iters); });
    printf("B  (sub,fma) scalar   : %.3f ms  %.1f T lane-ops/s  %.1f T terms/s\n", ms, 2 * lanes * NACC / ms / 1e9, lanes * NACC / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(kC, dim3(grid), dim3(256), 0, 0, out, in, iters); });
    printf("C  (sub,fma) packed   : %.3f ms  %.1f T pk-instr-lanes/s  %.1f T terms/s\n", ms, lanes * NACC / ms / 1e9, lanes * NACC / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(kC2, dim3(grid), dim3(256), 0, 0, out, in, iters); });
    printf("C2 packed, a broadcast: %.3f ms  %.1f T terms/s\n", ms, lanes * NACC / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(kD, dim3(grid), dim3(256), 0, 0, out, in, iters); });
    printf("D  pk_fma only        : %.3f ms  %.1f T terms/s  %.1f TFLOP/s\n", ms, lanes * NACC / ms / 1e9, 2 * lanes * NACC / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(kE<4>, dim3(grid), dim3(256), 0, 0, out, in, iters); });
    printf("E4 b64 load + 4 pk pairs : %.3f ms  %.1f T terms/s\n", ms, lanes * 8 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(kE<8>, dim3(grid), dim3(256), 0, 0, out, in, iters); });
    printf("E8 b64 load + 8 pk pairs : %.3f ms  %.1f T terms/s\n", ms, lanes * 16 / ms / 1e9);
  }
  return 0;
}
