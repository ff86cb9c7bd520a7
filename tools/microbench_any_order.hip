// Does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let two kernels of ONE stream overlap on gfx950?
// Two spin kernels of 128 single-wave workgroups each (half the CUs): serial = 2 T, overlapped = T.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int *sink) {
  const long long t0 = wall_clock64();
  int k = 0;
  while (wall_clock64() - t0 < cycles) ++k;
  if (k == -1) *sink = k;
}
static double run(int mode, hipStream_t st, hipStream_t st2, hipEvent_t ev, hipEvent_t ev2, int *sink) {
  const long long cyc = 100000 * 100;   // 100 MHz wall clock: 100 ms?  (scaled below)
  (void)cyc;
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int rep = 0; rep < 20; ++rep) {
    if (mode == 2) { hipEventRecord(ev, st); hipStreamWaitEvent(st2, ev, 0); }   // fork BEFORE A, as the library does
    hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, st, 20000LL, sink);        // ~200 us at 100 MHz
    if (mode == 0) hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, st, 20000LL, sink);
    if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(128), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 20000LL, sink);
    if (mode == 2) {   // event fork / join on a second stream
      hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, st2, 20000LL, sink);
      hipEventRecord(ev2, st2); hipStreamWaitEvent(st, ev2, 0);
    }
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 100LL, sink);            // the "join": a normal launch behind both
  }
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
}
int main() {
  hipStream_t st, st2; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
  hipEvent_t ev, ev2; hipEventCreateWithFlags(&ev, hipEventDisableTiming); hipEventCreateWithFlags(&ev2, hipEventDisableTiming);
  int *sink; hipMalloc(&sink, 4);
  for (int w = 0; w < 2; ++w)
    for (int mode = 0; mode < 3; ++mode)
      printf("%s: %.1f us per (A, B, join) round\n", mode == 0 ? "plain launches (serial)      " : mode == 1 ? "B with hipExtAnyOrderLaunch  " : "B on a second stream (events)",
             run(mode, st, st2, ev, ev2, sink));
  return 0;
}
