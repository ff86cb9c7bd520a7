// Review item 8 (round 2): the 9-tap horizontal box sums of the (25,9) dense forward, H = E . Band, on the matrix cores?
// One wave per workgroup, NIT offset steps each:
//   valu : the kernel's scheme on a lane's 10 pixels -- prefix / suffix blocks + one DPP operand (~25 VALU instructions)
//   mfma : H(16 x 32) = E(16 x 40) . Band(40 x 32) as v_mfma_f32_16x16x4_f32 -- per 16-centre block only the 24 E columns
//          under its band: 6 instructions, 12 per step (exact fp32 products with the 0/1 band; E taken as already laid out
//          for the A operand, which the real kernel would still have to arrange through LDS)
// and the accuracy side: an MFMA accumulates the K blocks one after the other -- a chain of 6 dependent partial sums
// where the kernel uses a depth-4 tree; flat rows (E ~ 1 + 1e-3 noise) show what that costs against an fp64 sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NIT = 20000;

template <int D> __device__ __forceinline__ float quad_next(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xF9, 0xf, 0xf, true));
}

__global__ __launch_bounds__(64) void k_valu(const float *in, float *out) {
  float e[10];
  for (int i = 0; i < 10; ++i) e[i] = in[threadIdx.x * 10 + i];
  float acc = 0.f;
  for (int it = 0; it < NIT; ++it) {
    // prefix / suffix sums of the lane's 10 values from shared blocks, windows of 9 = own suffix + next lane's prefix
    const float p01 = e[0] + e[1], p23 = e[2] + e[3], p45 = e[4] + e[5], p67 = e[6] + e[7], p89 = e[8] + e[9];
    const float p03 = p01 + p23, p47 = p45 + p67, p07 = p03 + p47;
    float pf[8] = {e[0], p01, p01 + e[2], p03, p03 + e[4], p03 + p45, p03 + p45 + e[6], p07};
    const float s89 = p89, s69 = p67 + p89, s29 = p23 + (p45 + s69);
    float sf[10] = {p07 + p89, e[1] + s29, s29, e[3] + (p45 + s69), p45 + s69, e[5] + s69, s69, e[7] + s89, s89, e[9]};
    float h[10];
    h[0] = pf[7] + e[8];          // window 0..8
    h[1] = sf[1];                 // window 1..9
#pragma unroll
    for (int k = 2; k < 10; ++k) h[k] = sf[k] + quad_next<1>(pf[k - 2]);   // k..9 own + 0..k-2 of the next lane
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) t += h[k];
    acc += t;
#pragma unroll
    for (int i = 0; i < 10; ++i) e[i] = __builtin_fmaf(e[i], 0.999f, 1e-3f);   // next step's E (keeps the loop honest)
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(64) void k_mfma(const float *in, float *out) {
  // A operand of 16x16x4: lane (row = lane % 16, k = lane / 16) holds E[row][4 kb + k]; 6 K blocks per 16-centre block
  float a[12];
  for (int i = 0; i < 12; ++i) a[i] = in[threadIdx.x * 12 + i];
  // B operand: Band[4 kb + k][col = lane % 16] = 1 if 0 <= (4 kb + k) - col <= 8
  float bnd[6];
  for (int kb = 0; kb < 6; ++kb) {
    const int kk = 4 * kb + (int)threadIdx.x / 16, col = threadIdx.x % 16;
    bnd[kb] = (kk - col >= 0 && kk - col <= 8) ? 1.f : 0.f;
  }
  f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int it = 0; it < NIT; ++it) {
    f4 h0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 6; ++kb) {
      h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb], bnd[kb], h0, 0, 0, 0);
      h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[6 + kb], bnd[kb], h1, 0, 0, 0);
    }
    acc0 += h0;
    acc1 += h1;
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = __builtin_fmaf(a[i], 0.999f, 1e-3f);
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
}

// accuracy: one 16 x 24 block of flat values through the MFMA chain vs the kernel's tree vs fp64
__global__ __launch_bounds__(64) void k_acc(const float *E, float *Hm) {
  // E[16][24] row-major; output Hm[16][16] = sum_{t=0..8} E[row][col + t]
  f4 h = {0, 0, 0, 0};
  for (int kb = 0; kb < 6; ++kb) {
    const int kk = 4 * kb + (int)threadIdx.x / 16, col = threadIdx.x % 16, row = threadIdx.x % 16;
    const float b = (kk - col >= 0 && kk - col <= 8) ? 1.f : 0.f;
    h = __builtin_amdgcn_mfma_f32_16x16x4f32(E[row * 24 + kk], b, h, 0, 0, 0);
  }
  // D layout of 16x16x4: lane holds rows 4 (lane / 16) + i, column lane % 16
  for (int i = 0; i < 4; ++i) Hm[(4 * (threadIdx.x / 16) + i) * 16 + threadIdx.x % 16] = h[i];
}

int main() {
  const int NB = 256 * 8;   // two waves per SIMD over the chip
  float *in, *out;
  hipMalloc(&in, 64 * 12 * 4); hipMalloc(&out, NB * 64 * 4);
  std::vector<float> h(64 * 12);
  for (auto &v : h) v = 0.5f + 0.5f * rand() / RAND_MAX;
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2];
  for (int v = 0; v < 2; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (v == 0) k_valu<<<NB, 64>>>(in, out); else k_mfma<<<NB, 64>>>(in, out);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[v], e0, e1);
    }
  }
  // NB waves over 1024 SIMDs = 2 waves per SIMD: cycles per step per SIMD = time * 2.4e9 / NIT; per wave half of it
  printf("horizontal 9-tap sums of one wave-step (16 rows x 40 pixels -> 16 x 32..40 sums), 2 waves per SIMD resident:\n");
  printf("  VALU prefix/suffix scheme : %.3f ms -> %.0f SIMD cycles per wave-step (incl. the 10-FMA E update and 10 adds of the check sum)\n", ms[0], ms[0] * 1e-3 * 2.4e9 / NIT / 2);
  printf("  12 x v_mfma_f32_16x16x4_f32: %.3f ms -> %.0f SIMD cycles per wave-step (incl. the 12-FMA A update)\n", ms[1], ms[1] * 1e-3 * 2.4e9 / NIT / 2);
  // accuracy on flat rows
  std::vector<float> E(16 * 24); std::vector<float> Hm(256);
  srand(7);
  double worst_m = 0, worst_t = 0;
  float *dE, *dH; hipMalloc(&dE, E.size() * 4); hipMalloc(&dH, 256 * 4);
  for (int trial = 0; trial < 200; ++trial) {
    for (auto &v : E) v = 1.0f + 1e-3f * (rand() / (float)RAND_MAX - 0.5f);
    hipMemcpy(dE, E.data(), E.size() * 4, hipMemcpyHostToDevice);
    k_acc<<<1, 64>>>(dE, dH);
    hipMemcpy(Hm.data(), dH, 256 * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
      double ref = 0; for (int t = 0; t < 9; ++t) ref += E[r * 24 + c + t];
      const float *x = &E[r * 24 + c];
      const float tree = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7])) + x[8];
      worst_m = fmax(worst_m, fabs(Hm[r * 16 + c] - ref) / ref);
      worst_t = fmax(worst_t, fabs(tree - ref) / ref);
    }
  }
  printf("flat rows (E = 1 +- 5e-4), 9-tap sums vs fp64: MFMA K-block chain max rel err %.2e ; depth-4 tree %.2e (fp32 ulp 6e-8)\n", worst_m, worst_t);
  return 0;
}
