// accuracy of v_log_f32 / v_rcp_f32 near 1 (is the hardware log2 relative-accurate for x ~ 1?)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *x, float *y, float *z, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { y[i] = __builtin_amdgcn_logf(x[i]); z[i] = __builtin_amdgcn_rcpf(x[i]); }
}
int main() {
  std::vector<float> h;
  for (int e = -24; e <= -1; ++e) for (int s = -1; s <= 1; s += 2) for (int m = 0; m < 64; ++m)
    h.push_back(1.0f + s * ldexpf(1.0f + m / 64.0f, e));
  for (int i = 0; i < 100000; ++i) h.push_back(0.0078f + i * 1e-7f);
  int n = h.size(); float *dx, *dy, *dz; hipMalloc(&dx, 4 * n); hipMalloc(&dy, 4 * n); hipMalloc(&dz, 4 * n);
  hipMemcpy(dx, h.data(), 4 * n, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(dx, dy, dz, n);
  std::vector<float> y(n), z(n); hipMemcpy(y.data(), dy, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(z.data(), dz, 4 * n, hipMemcpyDeviceToHost);
  double wrel = 0, wabs = 0, wrel2 = 0, wrcp = 0; int n1 = 24 * 2 * 64;
  for (int i = 0; i < n; ++i) {
    double t = log2((double)h[i]); double e = fabs(y[i] - t);
    if (i < n1) { if (t != 0) wrel = fmax(wrel, e / fabs(t)); wabs = fmax(wabs, e); } else wrel2 = fmax(wrel2, e / fabs(t));
    wrcp = fmax(wrcp, fabs(z[i] * (double)h[i] - 1.0));
  }
  printf("near 1: max rel err %.3e  max abs err %.3e ; range [0.0078,0.0178]: max rel err %.3e (ulp=6e-8) ; rcp max rel %.3e\n", wrel, wabs, wrel2, wrcp);
  for (int e = -20; e <= -2; e += 3) { float x = 1.0f + ldexpf(1.0f, e); int i = 0; for (; i < n1; ++i) if (h[i] == x) break; printf("  x=1+2^%d: hw %.9e  true %.9e\n", e, y[i], log2((double)x)); }
}
