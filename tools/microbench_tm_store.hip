// Micro-benchmark (round 5): what the memory system takes from the k_s = 49 strip forward's STORES, by layout and store
// width.  480 workgroups x 192 lanes; a lane owns 6 centres of one column of a 36 x 32 strip (the edge role's map) and
// walks q = 0 .. 2400, one value per centre and offset -- 5.3 GB per launch, nothing else.
//   mode 0  tile-major [slot][q][128 px], one dword per centre and step (the shipped layout)
//   mode 1  the same, nontemporal
//   mode 2  [slot][q/4][128 px][4]: one dwordx4 per centre every 4 steps
//   mode 3  the same, nontemporal
//   mode 4  [slot][q/2][128 px][2]: one dwordx2 per centre every 2 steps
//   mode 5  reference: the same bytes as a linear dwordx4 stream (each workgroup a contiguous 11 MB)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_tm_store.hip -o /tmp/tm_store && /tmp/tm_store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int P = 2401, PG4 = 601, PG2 = 1201;

template <int MODE>
__global__ __launch_bounds__(192) void tm_store(float *tm, int spin) {
  const int tid = threadIdx.x, ecol = tid % 32, e0 = (tid / 32) * 6;
  float acc = (float)tid;
  size_t base[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int tq = (e0 + j) >> 2, ey = (e0 + j) & 3, px = 64 * (ey & 1) + 32 * (ey >> 1) + ecol;
    const size_t slot = (size_t)blockIdx.x * 9 + tq;
    if (MODE <= 1) base[j] = slot * P * 128 + px;
    else if (MODE <= 3) base[j] = slot * PG4 * 512 + px * 4;
    else base[j] = slot * PG2 * 256 + px * 2;
  }
  if (MODE == 5) {
    f4 *o = (f4 *)(tm + (size_t)blockIdx.x * 9 * P * 128);
    const int n4 = 9 * P * 128 / 4;
    for (int i = tid; i < n4; i += 192) {
      f4 v = {acc, acc, acc, acc};
      __builtin_nontemporal_store(v, o + i);
    }
    return;
  }
  for (int q = 0; q < P; ++q) {
    for (int s = 0; s < spin; ++s) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
    if (MODE <= 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (MODE == 0) tm[base[j] + (size_t)q * 128] = acc;
        else __builtin_nontemporal_store(acc, tm + base[j] + (size_t)q * 128);
      }
    } else if (MODE <= 3) {
      if ((q & 3) == 3 || q == P - 1) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          f4 v = {acc, acc + 1, acc + 2, acc + 3};
          f4 *o = (f4 *)(tm + base[j] + (size_t)(q >> 2) * 512);
          if (MODE == 2) *o = v;
          else __builtin_nontemporal_store(v, o);
        }
      }
    } else {
      if ((q & 1) == 1 || q == P - 1) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          f2 v = {acc, acc + 1};
          *(f2 *)(tm + base[j] + (size_t)(q >> 1) * 256) = v;
        }
      }
    }
  }
  if (acc == 123.456f) tm[0] = acc;
}

template <int MODE>
static float run(float *tm, int spin) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  tm_store<MODE><<<480, 192>>>(tm, spin);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 3; ++i) tm_store<MODE><<<480, 192>>>(tm, spin);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 3;
}

int main() {
  const size_t floats = (size_t)480 * 9 * 2404 * 128 + 1024;
  float *tm;
  if (hipMalloc(&tm, floats * sizeof(float)) != hipSuccess) return 1;
  const double gb = 480.0 * 9 * 2401 * 128 * 4 / 1e9;
  printf("%.2f GB per launch; ms per launch (TB/s)\n", gb);
  for (int spin : {0, 40}) {
    const float t[6] = {run<0>(tm, spin), run<1>(tm, spin), run<2>(tm, spin), run<3>(tm, spin), run<4>(tm, spin), run<5>(tm, spin)};
    const char *name[6] = {"dword [q][px]", "dword nt", "x4 [q/4][px][4]", "x4 nt", "x2 [q/2][px][2]", "linear x4 nt"};
    printf("spin %2d:", spin);
    for (int i = 0; i < 6; ++i) printf("  %s %.3f (%.2f)", name[i], t[i], gb / t[i]);
    printf("\n");
  }
  return 0;
}
