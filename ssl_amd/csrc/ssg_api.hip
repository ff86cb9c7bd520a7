// C ABI of the gfx950 SSG engine: argument checking, workspace carving and
// kernel dispatch.  See include/ssg_hip.h for the contract of every entry point
// and the reference interface it replaces.
#include "../../include/ssg_hip.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

#include "ssg_common.hpp"

namespace ssg {
int launch_fwd(const FwdParams &p, hipStream_t st);
int launch_bwd(const BwdParams &p, hipStream_t st);
unsigned bwd_grid(const BwdParams &p);
size_t bwd_max_partials(int B, int H, int W, int n_rows);
int launch_loss_finalize(const float *partials, int nparts, const int *n_dev, int n_host, int P, float w_l1,
                         float w_kl, float *loss_out, int nan_on_overflow, hipStream_t st, int set_size = 0);
const char *fwd_kernel_name(int ks, int kw);
const char *bwd_kernel_name(int ks, int kw);
size_t edge_scratch_bytes(int B, int H, int W);
int launch_edge_list(const void *mask, int kind, int mask_channels, int B, int H, int W, int stride, float thr,
                     int *edges, int capacity, int *counts, int *rank, int *order, int *plan, int dense_thr,
                     int plan_tile_rows, void *scratch, void *zero_a, size_t zero_a_bytes, void *zero_b,
                     size_t zero_b_bytes, void *zero_c, size_t zero_c_bytes, hipStream_t st);
size_t fwd_plan_bytes(int B, int H, int W, int capacity);
int fwd_plan_order_offset(int B, int H, int W);
struct DenseParams {
  const float *img[2];
  float *out[2];
  int nimg;
  const int *rank;
  const int *n_dense;
  const int *tiles;
  int max_tiles;
  const int *n_dev;
  int n_host;
  int B, H, W;
  float sigma, eps;
  int generalization;
  int dbg;
  double *row_scale;
  float *tm[2];
  int tm_slots;
  int grid_tiles;
  const int *strips;
  int max_strips;
  int *status;
  int raw;
};
int fwd_plan_strip_offset(int B, int H, int W);
int dense_max_strips(int B, int H, int W, int ks);
bool dense_supported(int ks, int kw, int C);
int dense_max_tiles(int B, int H, int W, int ks);
int dense_tile_rows(int ks);
int launch_fwd_dense(const DenseParams &p, int ks, int kw, int C, hipStream_t st);
int launch_edge_mask(const float *gt, int B, int H, int W, float thr, int stride, uint8_t *out, hipStream_t st);
bool grow_supported(int ks, int kw);
unsigned grow_grid(int n_host);
int launch_grad_rows(const GrowParams &p, int ks, int kw, hipStream_t st);
int launch_rows_tm(const TmRowsParams &p, int ks, int kw, hipStream_t st);
int rows_tm_parts(int n_tiles);
int launch_pos_to_mask(const int *pos, int mc, int Hp, int Wp, uint8_t *mask, hipStream_t st);
int launch_pos_relabel(const int *pos, int mc, int Hp, int Wp, int *rank, int *perm, int *dup, int *ndup, int *plan,
                       int *order2, hipStream_t st);
size_t criteria_scratch_bytes();
int launch_criteria_sums(const float *a, const float *b, size_t n, void *scratch, float *sums_out, hipStream_t st);
int launch_criteria_grad(const float *a, const float *b, size_t n, const float *coef, float *g, hipStream_t st);
bool dense_bwd_supported(int ks, int kw, int C);
int launch_bwd_dense(const DenseBwdParams &p, int ks, int kw, int C, hipStream_t st);
int launch_augment_crop(const void *src, void *dst, int elem_bytes, int B, int C, int Hs, int Ws, int Ho, int Wo,
                        const int *params, hipStream_t st);
int launch_pool_swap(void *queue, void *batch, size_t sample_bytes, const int *slots, int b, hipStream_t st);
size_t usm_scratch_bytes(int B, int C, int H, int W);
int launch_jpeg(const float *img, float *out, int B, int H, int W, const float *quality_dev, float quality_host,
                hipStream_t st);
int launch_filter2d(const float *img, const float *kernels, float *out, int B, int C, int H, int W, int k, int nk,
                    hipStream_t st);
int launch_usm_sharp(const float *img, float *out, int B, int C, int H, int W, int ksize, float sigma, float weight,
                     float threshold, void *scratch, hipStream_t st);
int launch_grad_fix_flush(long long *gfix, float *grad, size_t n, int assign, hipStream_t st, const LossFinalize *fin = nullptr);
int launch_grad_fix_bound(const BwdParams &p, hipStream_t st);
int launch_grad_fix_reduce(const float *part, int n, long long *gfix, size_t n_fix, hipStream_t st);
bool tiny_edge_list_ok(int B, int H, int W);
int launch_tiny_edge_list(const void *mask, int kind, int mask_channels, int B, int H, int W, int stride, float thr,
                          int *edges, int capacity, int *counts, int *rank, void *zero_a, size_t zero_a_bytes,
                          void *zero_b, size_t zero_b_bytes, void *zero_c, size_t zero_c_bytes, hipStream_t st);
bool tiny_step_supported(int ks, int kw, int C, int capacity);
int launch_tiny_step(const TinyParams &p, int C, hipStream_t st);
}  // namespace ssg

using namespace ssg;

// Edge pixels per dense tile (8 x 32 for k_s 25, 4 x 32 for k_s 49) from which the shared-term kernels take the tile
// (0 = never).  The marginal costs of the two paths are equal between 16 and 28 pixels per tile; below ~16 the direct
// kernels win.  Round-4 sweeps (tools/r4_thr_sweep2.sh, r4_thr_sweep3.sh; same box): 20 against 28: C2
// 1.308 vs 1.318 ms, stride-3 C4 0.540 vs 0.560, Bernoulli 4 % 0.513 vs 0.512, C5 7.78 vs 7.78; again at the end of the
// round, after the direct forward had lost 9 % of its instructions (profiles/r4_dense_threshold_18_vs_20.txt, three
// alternations): 18 against 20: C2 1.2771 vs 1.2849, fused 1.2225 vs 1.2309, C4 0.5024 vs 0.5196, Bernoulli 4 % 0.4925
// vs 0.4868, 1 % equal (16 costs the Bernoulli masks 3 %, 24 and above cost C4 4 %).  Round 6, after the dense-tile
// kernels lost 6-7 % of their time (profiles/r6_dense_threshold_sweep.txt, three alternations): 16 against 18: C2 1.1936
// vs 1.2034, C4 0.446 vs 0.444, Bernoulli 4 % 0.456 vs 0.459, 1 % equal; 14: 1.1987 / 0.439; 12: 1.208 / 0.428 (C4's
// best); 10 costs Bernoulli 4 % 20 %.  Default 16.
// ssg_set_dense_threshold(n) overrides it (profiling build: also the environment variable SSG_DENSE_THR at first use).
constexpr int DENSE_THR_DEFAULT = 16;
static std::atomic<int> g_dense_thr{-1};   // (atomic: the ABI may be called from several host threads)
static int dense_threshold() {
  int v = g_dense_thr.load(std::memory_order_relaxed);
  if (v < 0) {
    v = env_int("SSG_DENSE_THR", DENSE_THR_DEFAULT);
    if (v < 0) v = 0;
    g_dense_thr.store(v, std::memory_order_relaxed);
  }
  return v;
}

// Small (11,5) steps in two launches (ssg_tiny.hip); on by default, ssg_set_tiny_step(0) keeps every call on the general path.
static std::atomic<int> g_tiny_step{1};
extern "C" int ssg_set_tiny_step(int on) { return g_tiny_step.exchange(on != 0 ? 1 : 0, std::memory_order_relaxed); }
extern "C" int ssg_set_operator_plan_threshold(int positions);
extern "C" int ssg_set_dense_threshold(int edge_pixels_per_tile) {
  const int prev = dense_threshold();
  g_dense_thr.store(edge_pixels_per_tile > 0 ? edge_pixels_per_tile : 0, std::memory_order_relaxed);
  return prev;
}

// Profiling build only (-DSSG_PROFILE -> libssg_hip_prof.so): SSG_DEBUG_SKIP=<bitmask> / ssg_set_profile_mask ablate
// kernel phases or skip whole launches (results are then WRONG).  Bits 0-7 direct forward phases, 8-15 backward
// phases, 16-23 dense forward phases, 24 no split backward; launches skipped: 25 dense forward, 26 direct forward,
// 27 dense backward, 28 direct backward (split mode), 29 G rows.  The product library has neither the symbol nor the
// environment variable: its mask is the constant 0 and every test on it folds away.
#ifdef SSG_PROFILE
static std::atomic<int> g_dbg{-1};
static int dbg_mask() {
  int v = g_dbg.load(std::memory_order_relaxed);
  if (v < 0) {
    v = env_int("SSG_DEBUG_SKIP", 0);
    if (v < 0) v = 0;
    g_dbg.store(v, std::memory_order_relaxed);
  }
  return v;
}
namespace ssg { int strip_occupancy(); int strip_times(unsigned long long *host, int n); }
// LDS poison (ssg_common.hpp): one workgroup per CU at a time (all 160 KB), several rounds so that every CU gets one
static std::atomic<int> g_lds_poison_on{0};
static std::atomic<unsigned> g_lds_poison_pat{0};
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pat, int words) {
  extern __shared__ unsigned s_poison[];
  volatile unsigned *w = s_poison;
  for (int i = threadIdx.x; i < words; i += 256) w[i] = pat;
}
namespace ssg {
void prof_poison_lds(hipStream_t st) {
  if (!g_lds_poison_on.load(std::memory_order_relaxed)) return;
  constexpr int BYTES = 160 * 1024;
  static std::atomic<unsigned long long> lds_set{0};
  if (ensure_dynamic_lds(lds_poison_kernel, BYTES, lds_set)) return;
  lds_poison_kernel<<<dim3(1024), dim3(256), BYTES, st>>>(g_lds_poison_pat.load(std::memory_order_relaxed), BYTES / 4);
}
}  // namespace ssg
extern "C" int ssg_prof_set_lds_poison(int on, unsigned pattern) {
  const int prev = g_lds_poison_on.load(std::memory_order_relaxed);
  g_lds_poison_pat.store(pattern, std::memory_order_relaxed);
  g_lds_poison_on.store(on ? 1 : 0, std::memory_order_relaxed);
  return prev;
}
extern "C" int ssg_prof_strip_times(unsigned long long *host, int n) { return ssg::strip_times(host, n); }
extern "C" int ssg_prof_occupancy(int which) { return which == 0 ? ssg::strip_occupancy() : -1; }
extern "C" int ssg_set_profile_mask(int mask) {
  const int prev = dbg_mask();
  g_dbg.store(mask > 0 ? mask : 0, std::memory_order_relaxed);
  return prev;
}
#else
static constexpr int dbg_mask() { return 0; }
#endif

// The dense-tile kernel and the direct kernel of a pass work on disjoint SSG rows (and add into the gradient with
// atomics), so the direct one runs on a side stream beside the dense one: fork = side waits for an event on the
// caller's stream, join = the caller's stream waits for the side's event.  Both are plain event edges, so a stream
// capture of the caller's stream (hipGraph) records the fork as two parallel branches.  One side stream and three
// events per (host thread, device); ssg_set_overlap(0) keeps every launch on the caller's stream.  Measured on MI355X:
// C2 (k_s 25) 1.541 -> 1.510 ms per step; C5 (k_s 49, every kernel already fills the chip for its whole run)
// 9.15 -> 9.64 ms -- so the fork is taken for k_s <= 25 only.
struct SideStream {
  hipStream_t side = nullptr;
  hipEvent_t forked = nullptr, joined = nullptr, gate = nullptr, gate2 = nullptr;
};
static std::atomic<int> g_overlap{-1};
// 0 off, 1 dense-tile kernel on the caller's stream, 2 direct kernel on the caller's stream, 3 (default) whichever of the
// two the previous plan built on this device makes the longer branch
static int overlap_mode() {
  int v = g_overlap.load(std::memory_order_relaxed);
  if (v < 0) {
    v = env_int("SSG_OVERLAP", 3);
    v = v < 0 ? 0 : (v > 3 ? 3 : v);
    g_overlap.store(v, std::memory_order_relaxed);
  }
  return v;
}
static bool overlap_enabled() { return overlap_mode() != 0; }
// Schedules of a fused step at k_s <= 25 (ForkChain below).  FREE-RUNNING chains pay when the direct chain carries the
// step (Bernoulli 4 %: 0.485 -> 0.463 ms) or the dense kernels are too few workgroups to keep the chip to themselves
// (C4, 330 tiles: 0.50 -> 0.466).  Where long dense kernels dominate (C2: 1,287 tiles, direct 0.38 ms of kernel time
// against 0.95) free-running direct kernels spend themselves beside the dense FORWARD and the dense backward runs alone
// with its tail unfilled (+8 %); there the chains are GATED -- the direct backward waits, one way, for the dense chain's
// row pass (C2 1.271 -> 1.248 against fork / join around each pass; a low-priority side stream: no effect).
// two_chains_wanted(): free-running (true) or gated (false), per call from the last plan's shape (PlanHint: the numbers
// the stream assignment uses); unknown = gated.
// (profiling build: SSG_TWO_CHAINS=0 always gated, 2 always free-running, -1 no chains: fork / join around each pass)
static bool two_chains_wanted();
static bool two_chains_allowed();

// What the last plan built on a device looked like: {rows left to the direct kernels, dense tiles}, written by the
// edge-list builder's scan kernel with a plain store into host-mapped pinned memory (one 64-byte block per device,
// allocated at the first build, never freed; a posted PCIe write, no stream operation) and read by the HOST at the next
// forked pass -- no synchronisation: whatever has landed is a hint, and both stream assignments give the same results.
struct PlanHint {
  int *host = nullptr, *dev = nullptr;
};
static PlanHint plan_hint(bool allocate_never = false) {
  constexpr int MAXDEV = 64;
  static std::mutex mu;
  static PlanHint tab[MAXDEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return PlanHint{};
  std::lock_guard<std::mutex> lk(mu);
  if (!tab[dev].host) {
    // hipHostMalloc is not allowed while the calling stream is being captured (it fails, and in the global / thread-local
    // capture modes invalidates the capture): the block is allocated by the first call per device made OUTSIDE capture;
    // a first call under capture runs without a hint (gated chains, dense kernels on the caller's stream).
    if (allocate_never) return PlanHint{};
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return PlanHint{};
    ((int *)h)[0] = ((int *)h)[1] = -1;   // nothing seen yet
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipHostFree(h);
      return PlanHint{};
    }
    tab[dev] = PlanHint{(int *)h, (int *)d};
  }
  return tab[dev];
}
// rows the last plan built on this device left to the direct kernels (-> FwdParams / BwdParams::rows_hint); 0 = unknown
static int hint_sparse_rows() {
  if (overlap_mode() != 3) return 0;
  const PlanHint h = plan_hint(true);
  if (!h.host) return 0;
  const int n = ((volatile int *)h.host)[0];
  return n > 0 ? n : 0;
}
namespace ssg {
int *plan_hint_device_word(hipStream_t st) {
  if (overlap_mode() != 3) return nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  return plan_hint(capturing).dev;
}
}
// the two streams of a forked pass: .dense takes the dense-tile kernel, .direct the direct one
struct StreamPair {
  hipStream_t dense, direct;
};
static thread_local int t_last_assignment = 0;
static StreamPair assign_streams(hipStream_t st, hipStream_t st2) {
  int mode = overlap_mode();
  if (mode == 3) {
    // whole-chip costs at (25,9), MI355X: a dense tile 0.74 us through forward + backward, a direct row 38 ns
    mode = 1;
    const PlanHint h = plan_hint(true);   // (readers never allocate: the builder's launch does, outside capture)
    if (h.host) {
      const int n_sparse = ((volatile int *)h.host)[0], n_tiles = ((volatile int *)h.host)[1];
      if (n_sparse >= 0 && n_tiles >= 0 && (long long)n_sparse * 38 > (long long)n_tiles * 740) mode = 2;
    }
  }
  t_last_assignment = st2 == st ? 0 : mode;
  return mode == 2 ? StreamPair{st2, st} : StreamPair{st, st2};
}
// Gated chains, round 6: the direct backward is released when the dense FORWARD is through (not, as in round 5, the dense
// chain's row pass): it then runs beside that memory-bound row pass -- C2 1.1996 -> 1.178 ms, three alternations
// (profiles/r6_schedule_ab.txt; holding the direct FORWARD until the dense forward is through as well: 1.184 alone,
// 1.20 together).  The loss finalize follows the direct backward on its stream, behind a second event for the dense
// chain's row pass.  (profiling build: SSG_SCHED=1 restores the round-5 gate.)
static int sched_mode() {
  static const int m = env_int("SSG_SCHED", 0);
  return m;
}
static bool two_chains_allowed() {   // (profiling build: SSG_TWO_CHAINS=-1 restores fork / join around forward and backward each)
  static const bool on = env_int("SSG_TWO_CHAINS", 1) >= 0;
  return on;
}
// (a stream that is being captured into a HIP graph: free-running chains replay well -- C4 0.452 ms as a graph, 0.495
// joined --, a GATED pair does not: C2 as a graph 1.44 ms gated, 1.26 with fork / join around each pass, which it keeps)
static bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
static bool two_chains_wanted() {    // free-running (true) or gated (false)
  static const int sw = env_int("SSG_TWO_CHAINS", 1);
  if (sw != 1) return sw == 2;
  if (overlap_mode() != 3) return false;
  const PlanHint h = plan_hint(true);
  if (!h.host) return false;
  const int n_sparse = ((volatile int *)h.host)[0], n_tiles = ((volatile int *)h.host)[1];
  if (n_sparse < 0 || n_tiles < 0) return false;
  // the direct chain carries the step, or the dense kernels' grids (two images per tile) are at most two resident rounds
  // of 512 workgroups: too short to keep the chip to themselves anyway
  return (long long)n_sparse * 38 > (long long)n_tiles * 740 || n_tiles <= 512;
}
// (diagnostics: the assignment the calling thread's last forked pass used -- 0 none, 1 dense-tile kernel on the caller's
// stream, 2 direct kernel on the caller's stream)
extern "C" int ssg_last_overlap_assignment(void) { return t_last_assignment; }
// ssg_set_overlap(0): every launch on the caller's stream (per-kernel rocprofv3 durations: profiles/*_kernel_stats.csv
// are taken that way; same results -- the two branches work on disjoint rows).  1 (default): the dense-tile kernel on
// the caller's stream, the direct one on the side stream -- right where dense tiles carry most rows (Laplacian masks:
// C2).  2: the other way round -- right for masks WITHOUT dense tiles (Bernoulli / thin strided masks), whose whole
// critical path then sits on one stream and whose join finds the side stream's empty launches finished.  Measured on one
// box (profiles/r5_ab_stream_assignment.txt): mode 2 against 1: Bernoulli 1 % 0.213 -> 0.182 ms, 4 % 0.491 -> 0.466,
// C4 0.498 -> 0.508, C2 1.275 -> 1.317 (the fork's latency lands on whichever kernel runs on the side stream).
// 3 (default since round 5): 1 or 2 per pass, from the shape of the last plan built on the device (PlanHint above): the
// branch expected to run longer stays on the caller's stream.  Steady streams of similar masks settle after one call.
// Returns the previous setting.
extern "C" int ssg_set_overlap(int mode) {
  const int prev = overlap_mode();
  g_overlap.store(mode < 0 ? 0 : (mode > 3 ? 3 : mode), std::memory_order_relaxed);
  return prev;
}
static SideStream *side_stream() {
  constexpr int MAXDEV = 64;
  thread_local SideStream tab[MAXDEV];
  int dev = 0;
  if (!overlap_enabled() || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  SideStream &s = tab[dev];
  if (!s.side) {
    if (hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s.forked, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.joined, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.gate, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.gate2, hipEventDisableTiming) != hipSuccess) {
      s.side = nullptr;
      return nullptr;
    }
  }
  return &s;
}
// stream for the second branch after everything queued on `st` so far (st itself when overlap is off)
static hipStream_t fork_from(hipStream_t st, int ks, SideStream *&s) {
  s = ks <= 25 ? side_stream() : nullptr;
  if (!s) return st;
  if (hipEventRecord(s->forked, st) != hipSuccess || hipStreamWaitEvent(s->side, s->forked, 0) != hipSuccess) {
    s = nullptr;
    return st;
  }
  return s->side;
}
static int join_to(hipStream_t st, SideStream *s) {
  if (!s) return 0;
  int rc = (int)hipEventRecord(s->joined, s->side);
  if (!rc) rc = (int)hipStreamWaitEvent(st, s->joined, 0);
  return rc;
}
// Two-chain step (the fused entry points at k_s <= 25, round 5): the forward's fork is NOT joined before the backward.
// The dense-tile chain (forward, its rows' ssg_grad_rows pass, backward) and the direct chain (the same three for the
// rows of the plan's sparse list) share nothing until the end: the row passes are per class anyway (split_backward), the
// fixed-point scale of the deterministic accumulation comes from the a-priori bound of |G| (no maximum over all rows
// to wait for), integer sums do not care who adds first, the criteria sums go to separate slots.  One join, then flush
// and finalize.  Same bits as the joined schedule -- only the streams differ.  Against fork / join around forward and
// backward each: one cross-stream round trip less, and the memory-bound row pass of one chain runs beside the VALU-bound
// kernels of the other.
// `gated`: the direct chain's backward is held (a one-way event: the dense chain never waits) until the dense chain's
// row pass is through -- the schedule for steps whose long dense kernels dominate (two_chains_wanted() says no): the
// direct backward then runs beside the dense backward and fills its tail instead of spending itself beside the dense
// forward.  The loss finalize rides on the direct chain's stream behind the gate, off the critical path.
struct ForkChain {
  SideStream *fk = nullptr;
  StreamPair sp{nullptr, nullptr};
  bool active = false;   // forked by the forward, to be joined by the backward
  bool gated = false;
};

// One 4-byte status word per device, owned by the library (allocated at the first call that can set it, never freed):
// kernels that refuse their input without a host-visible error -- a dense kernel handed a plan cut for another tile
// height -- set a bit in it; ssg_device_status() reads and clears it.
static int *device_status_word() {
  constexpr int MAXDEV = 64;
  static std::mutex mu;
  static int *tab[MAXDEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tab[dev]) {
    int *w = nullptr;
    if (hipMalloc(&w, sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(w, 0, sizeof(int)) != hipSuccess) {
      (void)hipFree(w);
      return nullptr;
    }
    tab[dev] = w;
  }
  return tab[dev];
}

static bool sizes_ok(int ks, int kw) { return ks > 0 && kw > 0 && (ks & 1) && (kw & 1) && kw <= ks; }

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Waves per tile of the dense-tile backward (each takes a contiguous range of offset rows): by default chosen on the
// device so that the launch fills the chip about once; SSG_BWD_QSPLIT fixes it in the profiling build (experiments).
static int bwd_qsplit() {   // 0 = chosen on the device from the number of dense tiles (DenseBwdParams::qsplit)
  static int v = -1;
  if (v < 0) {
    v = env_int("SSG_BWD_QSPLIT", 0);
    if (v < 0) v = 0;
    if (v > 25) v = 25;
  }
  return v;
}

// Scratch of the split backward (ssg_grad_rows -> dense-tile kernel + direct kernel): G (n, k_s^2), sum_b (n).
static size_t split_scratch_bytes(int n_rows, int ks) {
  const size_t n = (size_t)(n_rows > 0 ? n_rows : 1);
  return align_up(sizeof(float) * n * ks * ks, 256) + 3 * align_up(sizeof(float) * n, 256);   // G, sum_b, max|G| parts, dot
}

// Tile-major scratch rows of the fused step at k_s = 49 (TmRowsParams, ssg_common.hpp): the dense tiles in the first
// `slots` places of the plan keep their e rows in a second rows region, tile by tile, and the three kernels that touch
// them move whole 256-byte runs.  (Profiling build: SSG_TILE_MAJOR=0 keeps every row row-major, SSG_STRIPS=0 leaves
// every tile-major tile to the tile kernel -- A/B measurements; a product caller gets the row-major step by not giving
// the workspace the tile-major bytes, LossStep(tile_major=False).)
struct TileMajor {
  float *rows[2] = {nullptr, nullptr};
  int slots = 0;
};
static bool strips_enabled() {   // SSG_STRIPS=0: the tile kernel computes every tile-major tile (A/B measurements)
  static int v = -1;
  if (v < 0) {
    v = env_int("SSG_STRIPS", 1) != 0;
  }
  return v != 0;
}
static bool tile_major_enabled() {
  static int v = -1;
  if (v < 0) {
    v = env_int("SSG_TILE_MAJOR", 1) != 0;
  }
  return v != 0;
}

// Backward over a forward plan: G rows (+ criteria sums) by ssg_grad_rows, the dense tiles by the shared-term
// kernel, the remaining rows by the direct kernel in GRAD_D mode.  `p` carries the sources as for launch_bwd.
// the loss finalize of a GRAD_LOSS step: it needs ssg_grad_rows' partial sums only, so it is queued on the side stream
// ahead of the direct backward kernel instead of at the very end of the caller's stream (7 us off the critical path)
using FinalizeArgs = LossFinalize;   // {partials, nparts, n_dev, n_host, P, w_l1, w_kl, loss_out, nan_on_overflow}

// nparts of a split backward's criteria sums: ssg_grad_rows' workgroups, then ssg_rows_tm's
static int split_tm_tiles(const BwdParams &p, const TileMajor *tm) {
  if (!tm || tm->slots <= 0) return 0;
  const int mt = dense_max_tiles(p.B, p.H, p.W, p.ks);
  return tm->slots < mt ? tm->slots : mt;
}

// per-class row passes (split_backward): the criteria sums then take two sets of grow_grid(n_rows) slots
static bool split_row_classes(const BwdParams &p, int n_tm) {
  return p.mode == GRAD_LOSS && n_tm == 0 && p.row_scale && p.ks <= 25;
}

static int split_backward(BwdParams p, const int *rank, const int *plan, void *scratch, hipStream_t st,
                          const FinalizeArgs *fin = nullptr, bool *fin_done = nullptr, const TileMajor *tm = nullptr,
                          ForkChain *chain = nullptr) {
  float *G = (float *)scratch;
  const size_t nfl = align_up(sizeof(float) * (size_t)(p.n_host > 0 ? p.n_host : 1), 256);
  float *sum_b = (float *)((char *)scratch + align_up(sizeof(float) * (size_t)p.n_host * p.ks * p.ks, 256));
  float *gmax_part = (float *)((char *)sum_b + nfl);
  float *dot = (float *)((char *)gmax_part + nfl);
  const int n_tm = split_tm_tiles(p, tm);
  GrowParams g{};
  g.mode = p.mode;
  g.gin = p.gin;
  g.ssg = p.ssg;
  g.ssg2 = p.ssg2;
  g.row_scale = p.row_scale;
  g.rows_scratch = p.rows_scratch;
  g.n_dev = p.n_dev;
  g.n_host = p.n_host;
  g.C = p.C;
  g.sigma = p.sigma;
  g.generalization = p.generalization;
  g.w_l1 = p.w_l1;
  g.w_kl = p.w_kl;
  g.upstream = p.upstream;
  g.G = p.grad ? G : nullptr;
  g.sum_b = p.grad ? sum_b : nullptr;
  g.partials = p.partials;
  g.gmax_part = p.gfix ? gmax_part : nullptr;
  if (n_tm > 0) {   // (tile-major call: ssg_grad_rows walks the plan's list of sparse rows with a capped grid)
    g.grid_cap = 4096;
    g.tm_hdr = plan + 1;
    g.tm_slots = tm->slots;
    g.sparse_order = plan + fwd_plan_order_offset(p.B, p.H, p.W);
  }
  // GRAD_LOSS without tile-major rows (every k_s <= 25 call): the fixed-point scale comes from the a-priori bound of |G|
  // (ssg_grad_rows' first workgroup writes it: no maximum over the rows, no reduction launch); and with deferred row
  // scales the rows are passed over PER CLASS -- the dense-tile kernels' rows (non-zero scale), then the plan's sparse
  // list -- whatever the schedule, so that the criteria sums are grouped the same way on one stream and on two.
  const size_t n_fix = (size_t)p.B * p.C * p.H * p.W;
  const bool apriori = p.gfix && p.mode == GRAD_LOSS && n_tm == 0;
  const bool classes = split_row_classes(p, n_tm);
  // A backward on its own (ssg_loss_backward: the deferred loop's node, the module) forks HERE and runs the same two
  // chains from the row passes on: the sparse list's pass beside the dense-tile rows' instead of behind it.
  ForkChain local;
  // (the hint is written by the GPU asynchronously: ONE read per decision, or a flip between two reads could record a
  // gated pair into a capturing stream)
  const bool free_wanted = two_chains_wanted();
  if (!(chain && chain->active) && classes && p.grad && !(dbg_mask() & ((1 << 27) | (1 << 28) | (1 << 29))) &&
      two_chains_allowed() && (free_wanted || !stream_capturing(st))) {
    SideStream *fk0 = nullptr;
    hipStream_t st20 = fork_from(st, p.ks, fk0);
    if (fk0 && st20 != st) {
      local.fk = fk0;
      local.sp = assign_streams(st, st20);
      local.active = true;
      local.gated = !free_wanted;
      chain = &local;
    }
  }
  const bool chained = chain && chain->active;
  hipStream_t sd = chained ? chain->sp.dense : st, ss = chained ? chain->sp.direct : st;
  auto leave = [&](int rc0) {   // (a forward that left its streams forked is joined on every way out)
    if (!chained) return rc0;
    const int rcj = join_to(st, chain->fk);
    chain->active = false;
    return rc0 ? rc0 : rcj;
  };
  if (chained && !(classes && p.grad)) return leave(SSG_E_BADARG);
  if (apriori) {
    g.gmax_part = nullptr;
    g.fix_word = (unsigned *)(p.gfix + n_fix);
  }
  int rc = 0;
  if (dbg_mask() & (1 << 29)) {
  } else if (classes) {
    const unsigned gg = grow_grid(p.n_host);
    const bool early = chained && chain->gated && sd == st && ss != st && !(sched_mode() & 1);
    if (early) rc = (int)hipEventRecord(chain->fk->gate, sd);   // (behind the dense forward, in front of its row pass)
    GrowParams gd = g;   // the dense-tile rows
    gd.only = 1;
    if (!rc) rc = launch_grad_rows(gd, p.ks, p.kw, sd);
    GrowParams gs = g;   // the plan's sparse list; its criteria sums behind the first pass's
    gs.only = 2;
    gs.grid_cap = 4096;
    gs.tm_hdr = plan + 1;
    gs.tm_slots = 0;
    gs.sparse_order = plan + fwd_plan_order_offset(p.B, p.H, p.W);
    gs.partials = p.partials + 2 * (size_t)gg;
    if (!rc) rc = launch_grad_rows(gs, p.ks, p.kw, ss);
  } else {
    rc = launch_grad_rows(g, p.ks, p.kw, st);
  }
  if (!rc && n_tm > 0) {   // the rows of the tile-major tiles (ssg_grad_rows skipped them: negative row scale)
    TmRowsParams t{};
    t.tm[0] = tm->rows[0];
    t.tm[1] = tm->rows[1];
    t.row_scale = p.row_scale;
    t.rank = rank;
    t.n_dense = plan + 1;
    t.tiles = plan + 4;
    t.n_tiles = n_tm;
    t.tm_slots = tm->slots;
    t.n_dev = p.n_dev;
    t.n_host = p.n_host;
    t.B = p.B;
    t.H = p.H;
    t.W = p.W;
    t.C = p.C;
    t.sigma = p.sigma;
    t.w_l1 = p.w_l1;
    t.w_kl = p.w_kl;
    t.upstream = p.upstream;
    t.dot = dot;
    t.sum_b = sum_b;
    t.partials = p.partials + 2 * (size_t)grow_grid(p.n_host);
    t.gmax_part = p.gfix ? gmax_part + grow_grid(p.n_host) : nullptr;
    if (!p.rows_scratch) {   // materialising call: the normalised rows of the tile-major tiles are written here
      t.out[0] = p.ssg;
      t.out[1] = p.ssg2;
    }
    rc = (dbg_mask() & (1 << 29)) ? 0 : launch_rows_tm(t, p.ks, p.kw, st);
  }
  if (rc || !p.grad) return leave(rc);
  if (p.gfix && !apriori) {
    rc = launch_grad_fix_reduce(gmax_part, (int)grow_grid(p.n_host) + rows_tm_parts(n_tm), p.gfix, n_fix, st);
    if (rc) return leave(rc);
  }
  const float *grows = p.mode == GRAD_D ? p.gin : G;
  DenseBwdParams d{};
  d.img = p.img;
  d.grad = p.grad;
  d.gfix = p.gfix;
  d.G = grows;
  d.sum_b = sum_b;
  d.rank = rank;
  d.n_dense = plan + 1;
  d.tiles = plan + 4;
  d.max_tiles = dense_max_tiles(p.B, p.H, p.W, p.ks);
  d.n_dev = p.n_dev;
  d.n_host = p.n_host;
  d.B = p.B;
  d.H = p.H;
  d.W = p.W;
  d.qsplit = bwd_qsplit();
  d.dbg = p.dbg;
  d.status = device_status_word();
  if (n_tm > 0) {
    d.tm[0] = tm->rows[0];
    d.tm[1] = tm->rows[1];
    d.row_scale = p.row_scale;
    d.dot = dot;
    d.tm_slots = tm->slots;
    d.sigma = p.sigma;
    d.w_l1 = p.w_l1;
    d.w_kl = p.w_kl;
    d.upstream = p.upstream;
  }
  if (chained) {   // two-chain step: each backward kernel behind its own chain's row pass; then the one join
    const bool early = chain->gated && sd == st && ss != st && !(sched_mode() & 1);
    if (chain->gated && sd == st && ss != st && !early) {
      rc = (int)hipEventRecord(chain->fk->gate, sd);
      if (!rc) rc = (int)hipStreamWaitEvent(ss, chain->fk->gate, 0);
      if (!rc && fin) {   // (both passes' criteria sums are complete behind the gate)
        rc = launch_loss_finalize(fin->partials, fin->nparts, fin->n_dev, fin->n_host, fin->P, fin->w_l1, fin->w_kl,
                                  fin->loss_out, fin->nan_on_overflow, ss, fin->set_size);
        if (!rc && fin_done) *fin_done = true;
      }
      if (rc) return leave(rc);
    }
    if (early) {   // the direct backward waits for the dense FORWARD only; the finalize, behind it, for the dense row pass
      rc = (int)hipStreamWaitEvent(ss, chain->fk->gate, 0);
      if (!rc) rc = (int)hipEventRecord(chain->fk->gate2, sd);
      if (rc) return leave(rc);
    }
    rc = launch_bwd_dense(d, p.ks, p.kw, p.C, sd);
    if (!rc) {
      BwdParams s = p;
      s.mode = GRAD_D;
      s.gin = grows;
      s.order = plan + fwd_plan_order_offset(p.B, p.H, p.W);
      s.n_dev = plan;  // n_sparse
      s.partials = nullptr;
      s.rows_hint = hint_sparse_rows();
      rc = launch_bwd(s, ss);
    }
    if (!rc && early && fin) {
      rc = (int)hipStreamWaitEvent(ss, chain->fk->gate2, 0);
      if (!rc) rc = launch_loss_finalize(fin->partials, fin->nparts, fin->n_dev, fin->n_host, fin->P, fin->w_l1, fin->w_kl,
                                         fin->loss_out, fin->nan_on_overflow, ss, fin->set_size);
      if (!rc && fin_done) *fin_done = true;
    }
    return leave(rc);
  }
  SideStream *fk = nullptr;
  hipStream_t st2 = (dbg_mask() & ((1 << 27) | (1 << 28))) ? st : fork_from(st, p.ks, fk);
  // (which kernel runs on which stream: ssg_set_overlap.  With a side stream the loss finalize -- it needs ssg_grad_rows'
  // partial sums only -- is queued there ahead of that stream's kernel, off the critical path; on one stream it rides in
  // the last workgroup of grad_fix_flush (det_end, round 5) or, without a fixed-point buffer, follows the backward.)
  if (fin && fk && st2 != st) {
    rc = launch_loss_finalize(fin->partials, fin->nparts, fin->n_dev, fin->n_host, fin->P, fin->w_l1, fin->w_kl,
                              fin->loss_out, fin->nan_on_overflow, st2, fin->set_size);
    if (rc) return rc;
    if (fin_done) *fin_done = true;
  }
  const StreamPair sp = assign_streams(st, st2);
  rc = (dbg_mask() & (1 << 27)) ? 0 : launch_bwd_dense(d, p.ks, p.kw, p.C, sp.dense);
  if (!rc && !(dbg_mask() & (1 << 28))) {
    BwdParams s = p;
    s.mode = GRAD_D;
    s.gin = grows;
    s.order = plan + fwd_plan_order_offset(p.B, p.H, p.W);
    s.n_dev = plan;  // n_sparse
    s.partials = nullptr;
    s.rows_hint = hint_sparse_rows();
    rc = launch_bwd(s, sp.direct);
  }

  const int rcj = join_to(st, fk);
  return rc ? rc : rcj;
}

// Deterministic mode: the kernels add into the caller's zeroed fixed-point buffer; one flush folds it into grad.
static int det_begin(BwdParams &p, void *grad_fix, hipStream_t st, bool prezeroed = false) {
  p.gfix = nullptr;
  if (!grad_fix || !p.grad) return 0;
  p.gfix = (long long *)grad_fix;
  if (prezeroed) return 0;   // (the fused step: cleared by the edge-list builder's first kernel)
  return (int)hipMemsetAsync(grad_fix, 0, sizeof(long long) * ((size_t)p.B * p.C * p.H * p.W + 8), st);
}
static int det_end(const BwdParams &p, hipStream_t st, bool assign = false, const LossFinalize *fin = nullptr, bool *fin_done = nullptr) {
  if (!p.gfix) return 0;
  const int rc = launch_grad_fix_flush(p.gfix, p.grad, (size_t)p.B * p.C * p.H * p.W, assign ? 1 : 0, st, fin);
  if (!rc && fin && fin_done) *fin_done = true;
  return rc;
}

// ---- the reference operator with many positions: plan built inside the call -------------------------------------------
// similarity.h's interface has no workspace argument, so the plan's buffers come from a library-owned, stream-ordered
// memory pool (one per device, release threshold = never: after the first call an allocation is a free-list lookup).
// Measured (profiles/r4_operator_vs_plan.txt): the plan costs ~45 us per call and a handful of dense tiles are one
// resident round whatever their number, so the direct kernels in `pos` order win below ~10 k positions (0.134 vs 0.180
// ms forward at 4,820) and lose above (0.251 vs 0.197 ms at 18,417; forward + backward 0.733 vs 0.454).
constexpr int OP_PLAN_FROM_DEFAULT = 8192;
static std::atomic<int> g_op_plan_from{-1};
// positions from which a call takes the plan path (0x7fffffff = never); ssg_set_operator_plan_threshold
// (n <= 0: never)
static int op_plan_from() {
  int v = g_op_plan_from.load(std::memory_order_relaxed);
  if (v < 0) {
    v = env_int("SSG_OP_PLAN_FROM", OP_PLAN_FROM_DEFAULT);
    if (v <= 0) v = 0x7fffffff;
    g_op_plan_from.store(v, std::memory_order_relaxed);
  }
  return v;
}

static hipMemPool_t op_pool() {
  constexpr int MAXDEV = 64;
  static std::mutex mu;
  static hipMemPool_t tab[MAXDEV] = {};
  static bool failed[MAXDEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tab[dev] && !failed[dev]) {
    hipMemPoolProps props{};
    props.allocType = hipMemAllocationTypePinned;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    hipMemPool_t pool = nullptr;
    if (hipMemPoolCreate(&pool, &props) != hipSuccess) {
      failed[dev] = true;
      (void)hipGetLastError();
      return nullptr;
    }
    unsigned long long keep = ~0ull;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    tab[dev] = pool;
  }
  return tab[dev];
}

struct OpPlan {
  char *base = nullptr;
  uint8_t *mask = nullptr;
  int *edges = nullptr, *counts = nullptr, *rank = nullptr, *plan = nullptr, *perm = nullptr, *dup = nullptr, *ndup = nullptr;
  void *escratch = nullptr, *bscratch = nullptr;
};

// true: the call qualifies for the plan path (enough positions, a size with shared-term kernels, not inside a stream capture)
// (the forward gains less from the shared-term kernel than the backward -- measured at 18,417 positions: forward 0.196 ms
// direct / 0.231 with the plan, backward 0.462 / 0.258 -- so it takes the plan from three times as many positions)
static bool op_wants_plan(int mc, int ks, int kw, int C, hipStream_t st, bool forward) {
  const long from = (long)op_plan_from() * (forward && op_plan_from() > 1 ? 3 : 1);
  if (from >= 0x7fffffffL || (long)mc < from || !dense_supported(ks, kw, C) || !dense_bwd_supported(ks, kw, C) || !grow_supported(ks, kw)) return false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return false;
  }
  return true;
}

// rank map, plan and duplicate list of `pos` (padded coordinates on a (Hp, Wp) image), everything relabelled to the
// caller's row numbers.  rc != 0 or o.base == nullptr: fall back to the direct path.
static int op_plan_build(const int *pos, int mc, int ks, int Hp, int Wp, size_t bwd_scratch_bytes, hipStream_t st, OpPlan &o) {
  hipMemPool_t pool = op_pool();
  if (!pool) return 0;
  const size_t npix = (size_t)Hp * Wp;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += align_up(bytes, 256);
    return at;
  };
  const size_t o_mask = take(npix), o_edges = take(sizeof(int) * 3 * (size_t)mc), o_counts = take(sizeof(int) * 8),
               o_rank = take(sizeof(int) * npix), o_plan = take(fwd_plan_bytes(1, Hp, Wp, mc)),
               o_perm = take(sizeof(int) * (size_t)mc), o_dup = take(sizeof(int) * (size_t)mc), o_ndup = take(sizeof(int) * 4),
               o_es = take(edge_scratch_bytes(1, Hp, Wp)), o_bs = take(bwd_scratch_bytes);
  void *base = nullptr;
  if (hipMallocFromPoolAsync(&base, off, pool, st) != hipSuccess || !base) {
    (void)hipGetLastError();
    return 0;
  }
  o.base = (char *)base;
  o.mask = (uint8_t *)(o.base + o_mask);
  o.edges = (int *)(o.base + o_edges);
  o.counts = (int *)(o.base + o_counts);
  o.rank = (int *)(o.base + o_rank);
  o.plan = (int *)(o.base + o_plan);
  o.perm = (int *)(o.base + o_perm);
  o.dup = (int *)(o.base + o_dup);
  o.ndup = (int *)(o.base + o_ndup);
  o.escratch = o.base + o_es;
  o.bscratch = bwd_scratch_bytes ? o.base + o_bs : nullptr;
  int rc = (int)hipMemsetAsync(o.mask, 0, npix, st);
  if (!rc) rc = (int)hipMemsetAsync(o.perm, 0xff, sizeof(int) * (size_t)mc, st);
  if (!rc) rc = (int)hipMemsetAsync(o.ndup, 0, sizeof(int) * 4, st);
  if (!rc) rc = launch_pos_to_mask(pos, mc, Hp, Wp, o.mask, st);
  if (!rc)
    rc = launch_edge_list(o.mask, 1, 1, 1, Hp, Wp, 0, 0.f, o.edges, mc, o.counts, o.rank, nullptr, o.plan, dense_threshold(),
                          dense_tile_rows(ks), o.escratch, nullptr, 0, nullptr, 0, nullptr, 0, st);
  if (!rc) rc = launch_pos_relabel(pos, mc, Hp, Wp, o.rank, o.perm, o.dup, o.ndup, o.plan, o.plan + fwd_plan_order_offset(1, Hp, Wp), st);
  return rc;
}

static int op_plan_free(OpPlan &o, hipStream_t st) {
  if (!o.base) return 0;
  const int rc = (int)hipFreeAsync(o.base, st);
  o.base = nullptr;
  return rc;
}

static bool split_ok(int ks, int kw, int C, const int *rank, const int *plan, const void *scratch) {
  return rank && plan && scratch && grow_supported(ks, kw) && dense_bwd_supported(ks, kw, C) &&
         !(dbg_mask() & (1 << 24));
}

extern "C" {

int ssg_abi_version(void) { return 6; }

const char *ssg_status_string(int status) {
  switch (status) {
    case 0: return "ok";
    case SSG_E_BADARG: return "ssg: bad argument (null pointer, even/non-positive kernel size, bad kind)";
    case SSG_E_TOOLARGE: return "ssg: search tile does not fit the 160 KiB LDS of a CU";
    case SSG_E_WORKSPACE: return "ssg: workspace too small";
    case SSG_E_IMAGESMALL: return "ssg: image side <= k_s/2, reflect padding undefined";
    case SSG_E_ALIGN: return "ssg: workspace / grad_fix / grad_sr of the fused step must be 16-byte aligned";
    case SSG_E_PLAN: return "ssg: a dense kernel was handed a plan cut for another tile height (k_s 25 vs 49); that launch did nothing";
    default: return status > 0 ? hipGetErrorString((hipError_t)status) : "ssg: unknown status";
  }
}

int ssg_compute_similarity(const float *image, const int *pos, float *out, int mc, int psize, int ksize, int height,
                           int width, int channel, ssg_stream_t stream) {
  if (mc < 0 || !sizes_ok(psize, ksize) || channel <= 0 || height <= 0 || width <= 0) return SSG_E_BADARG;
  if (mc == 0) return 0;
  if (!image || !pos || !out) return SSG_E_BADARG;
  FwdParams p{};
  p.img[0] = image;
  p.out[0] = out;
  p.nimg = 1;
  p.edges = pos;
  p.estride = 2;
  p.n_dev = nullptr;
  p.n_host = mc;
  p.B = 1;
  p.C = channel;
  p.H = height;
  p.W = width;
  p.sigma = 1.f;
  p.eps = 0.f;
  p.generalization = 0;
  p.raw = 1;
  p.ks = psize;
  p.kw = ksize;
  p.dbg = dbg_mask() & 0xff;
  hipStream_t st = (hipStream_t)stream;
  if (op_wants_plan(mc, psize, ksize, channel, st, true)) {
    OpPlan o;
    int rc = op_plan_build(pos, mc, psize, height, width, 0, st, o);
    if (!rc && o.base) {
      // dense tiles -> shared-term kernel in raw mode, the rest (the plan's tile order, caller's row numbers) -> the
      // direct kernels, duplicates of a position -> a direct launch of their own
      DenseParams d{};
      d.img[0] = image;
      d.out[0] = out;
      d.nimg = 1;
      d.rank = o.rank;
      d.n_dense = o.plan + 1;
      d.tiles = o.plan + 4;
      d.max_tiles = dense_max_tiles(1, height, width, psize);
      d.n_host = mc;
      d.B = 1;
      d.H = height;
      d.W = width;
      d.sigma = 1.f;
      d.raw = 1;
      d.dbg = (dbg_mask() >> 16) & 0xff;
      d.status = device_status_word();
      SideStream *fk = nullptr;
      hipStream_t st2 = fork_from(st, psize, fk);
      const StreamPair sp = assign_streams(st, st2);
      rc = launch_fwd_dense(d, psize, ksize, channel, sp.dense);
      FwdParams q = p;
      q.order = o.plan + fwd_plan_order_offset(1, height, width);
      q.n_dev = o.plan;   // n_sparse
      if (!rc) rc = launch_fwd(q, sp.direct);
      const int rcj = join_to(st, fk);
      if (!rc) rc = rcj;
      q.order = o.dup;
      q.n_dev = o.ndup;
      if (!rc) rc = launch_fwd(q, st);
    }
    const bool used = o.base != nullptr;
    const int rcf = op_plan_free(o, st);
    if (used || rc) return rc ? rc : rcf;
  }
  return launch_fwd(p, st);
}

int ssg_compute_similarity_backward(const float *image, const float *grads, const int *pos, float *image_grads,
                                    int mc, int psize, int ksize, int height, int width, int channel,
                                    ssg_stream_t stream) {
  if (mc < 0 || !sizes_ok(psize, ksize) || channel <= 0 || height <= 0 || width <= 0) return SSG_E_BADARG;
  if (mc == 0) return 0;
  if (!image || !grads || !pos || !image_grads) return SSG_E_BADARG;
  BwdParams p{};
  p.img = image;
  p.grad = image_grads;
  p.edges = pos;
  p.estride = 2;
  p.n_dev = nullptr;
  p.n_host = mc;
  p.B = 1;
  p.C = channel;
  p.H = height;
  p.W = width;
  p.mode = GRAD_D;
  p.gin = grads;
  p.sigma = 1.f;
  p.ks = psize;
  p.kw = ksize;
  p.dbg = (dbg_mask() >> 8) & 0xff;
  hipStream_t st = (hipStream_t)stream;
  if (op_wants_plan(mc, psize, ksize, channel, st, false)) {
    OpPlan o;
    int rc = op_plan_build(pos, mc, psize, height, width, split_scratch_bytes(mc, psize), st, o);
    if (!rc && o.base) {
      // the split backward in GRAD_D mode (`grads` ARE the G rows): border sums by ssg_grad_rows, dense tiles by the
      // shared-term kernel, the plan's sparse rows by the direct one; then the duplicates of a position on their own
      rc = split_backward(p, o.rank, o.plan, o.bscratch, st);
      BwdParams q = p;
      q.order = o.dup;
      q.n_dev = o.ndup;
      if (!rc) rc = launch_bwd(q, st);
    }
    const bool used = o.base != nullptr;
    const int rcf = op_plan_free(o, st);
    if (used || rc) return rc ? rc : rcf;
  }
  return launch_bwd(p, st);
}

size_t ssg_edge_scratch_bytes(int B, int H, int W) { return edge_scratch_bytes(B, H, W); }

size_t ssg_forward_plan_bytes(int B, int H, int W, int capacity) { return fwd_plan_bytes(B, H, W, capacity); }

static int edge_list_impl(const void *mask, int mask_kind, int mask_channels, int B, int H, int W, int mask_stride,
                          float lap_threshold, int plan_ks, int *edges, int capacity, int *counts, int *rank_map,
                          int *tile_order, int *fwd_plan, void *scratch, void *zero_a, size_t zero_a_bytes, void *zero_b,
                          size_t zero_b_bytes, void *zero_c, size_t zero_c_bytes, ssg_stream_t stream) {
  if (!mask || !edges || !counts || !scratch || B <= 0 || H <= 0 || W <= 0 || capacity < 0 || mask_kind < 0 ||
      mask_kind > 2 || mask_channels <= 0 || ((tile_order || fwd_plan) && !rank_map))
    return SSG_E_BADARG;
  // (the plan is always built when asked for -- with threshold 0 it lists no dense tile and every row in its
  // direct order -- so that the kernels consuming it never depend on the process-wide threshold)
  return launch_edge_list(mask, mask_kind, mask_channels, B, H, W, mask_stride, lap_threshold, edges, capacity,
                          counts, rank_map, tile_order, fwd_plan, dense_threshold(), dense_tile_rows(plan_ks), scratch,
                          zero_a, zero_a_bytes, zero_b, zero_b_bytes, zero_c, zero_c_bytes, (hipStream_t)stream);
}

int ssg_edge_list(const void *mask, int mask_kind, int mask_channels, int B, int H, int W, int mask_stride,
                  float lap_threshold, int plan_ks, int *edges, int capacity, int *counts, int *rank_map,
                  int *tile_order, int *fwd_plan, void *scratch, ssg_stream_t stream) {
  return edge_list_impl(mask, mask_kind, mask_channels, B, H, W, mask_stride, lap_threshold, plan_ks, edges, capacity,
                        counts, rank_map, tile_order, fwd_plan, scratch, nullptr, 0, nullptr, 0, nullptr, 0, stream);
}

int ssg_edge_mask_laplacian(const float *gt, int B, int H, int W, float lap_threshold, int mask_stride,
                            uint8_t *mask_out, ssg_stream_t stream) {
  if (!gt || !mask_out || B <= 0 || H <= 0 || W <= 0) return SSG_E_BADARG;
  return launch_edge_mask(gt, B, H, W, lap_threshold, mask_stride, mask_out, (hipStream_t)stream);
}

static int map_forward_impl(const float *img, const float *img2, int B, int C, int H, int W, const int *edges,
                            const int *tile_order, const int *rank_map, const int *fwd_plan, const int *n_edges_dev,
                            int n_rows, int ks, int kw, float sigma, float eps, int generalization, float *ssg,
                            float *ssg2, double *row_scale, bool row_scale_zeroed, ssg_stream_t stream,
                            const TileMajor *tm = nullptr, ForkChain *chain = nullptr) {
  if (n_rows < 0 || !sizes_ok(ks, kw) || B <= 0 || C <= 0) return SSG_E_BADARG;
  if (H <= ks / 2 || W <= ks / 2) return SSG_E_IMAGESMALL;
  if (n_rows == 0) return 0;
  if (!img || !edges || !ssg || ((img2 != nullptr) != (ssg2 != nullptr))) return SSG_E_BADARG;
  FwdParams p{};
  p.img[0] = img;
  p.img[1] = img2;
  p.out[0] = ssg;
  p.out[1] = ssg2;
  p.nimg = img2 ? 2 : 1;
  p.edges = edges;
  p.estride = 3;
  p.order = tile_order;
  p.n_dev = n_edges_dev;
  p.n_host = n_rows;
  p.B = B;
  p.C = C;
  p.H = H;
  p.W = W;
  p.sigma = sigma;
  p.eps = eps;
  p.generalization = generalization;
  p.raw = 0;
  p.ks = ks;
  p.kw = kw;
  p.dbg = dbg_mask() & 0xff;
  if (fwd_plan && rank_map && dense_supported(ks, kw, C)) {
    // dense tiles -> shared-term kernel; the rest (plan's own tile-major order) -> direct kernels
    DenseParams d{};
    d.img[0] = img;
    d.img[1] = img2;
    d.out[0] = ssg;
    d.out[1] = ssg2;
    d.nimg = p.nimg;
    d.rank = rank_map;
    d.n_dense = fwd_plan + 1;
    d.tiles = fwd_plan + 4;
    d.max_tiles = dense_max_tiles(B, H, W, ks);
    d.n_dev = n_edges_dev;
    d.n_host = n_rows;
    d.B = B;
    d.H = H;
    d.W = W;
    d.sigma = sigma;
    d.eps = eps;
    d.generalization = generalization;
    d.dbg = (dbg_mask() >> 16) & 0xff;
    d.row_scale = row_scale;
    d.status = device_status_word();
    if (tm && tm->slots > 0 && row_scale && img2) {
      d.tm[0] = tm->rows[0];
      d.tm[1] = tm->rows[1];
      d.tm_slots = tm->slots;
      d.max_strips = strips_enabled() ? dense_max_strips(B, H, W, ks) : 0;   // (k_s 49: whole strips of heavy tiles)
      d.strips = d.max_strips ? fwd_plan + fwd_plan_strip_offset(B, H, W) : nullptr;
    }
    if (row_scale && !row_scale_zeroed) {   // 0 = "this row is already normalised" (the rows of the direct kernels)
      const int rc0 = (int)hipMemsetAsync(row_scale, 0, sizeof(double) * 2 * (size_t)n_rows, (hipStream_t)stream);
      if (rc0) return rc0;
    }
    SideStream *fk = nullptr;
    hipStream_t st = (hipStream_t)stream;
    hipStream_t st2 = (dbg_mask() & ((1 << 25) | (1 << 26))) ? st : fork_from(st, ks, fk);
    const StreamPair sp = assign_streams(st, st2);   // (ssg_set_overlap)
    int rc = (dbg_mask() & (1 << 25)) ? 0 : launch_fwd_dense(d, ks, kw, C, sp.dense);
    p.order = fwd_plan + fwd_plan_order_offset(B, H, W);
    p.n_dev = fwd_plan;  // n_sparse
    p.rows_hint = hint_sparse_rows();
    if (!rc && !(dbg_mask() & (1 << 26))) rc = launch_fwd(p, sp.direct);
    if (!rc && chain && fk && st2 != st) {   // two-chain step: the backward's kernels follow on the same two streams
      chain->fk = fk;
      chain->sp = sp;
      chain->active = true;
      return 0;
    }
    const int rcj = join_to(st, fk);
    return rc ? rc : rcj;
  }
  return launch_fwd(p, (hipStream_t)stream);
}

int ssg_map_forward(const float *img, const float *img2, int B, int C, int H, int W, const int *edges,
                    const int *tile_order, const int *rank_map, const int *fwd_plan, const int *n_edges_dev, int n_rows,
                    int ks, int kw, float sigma, float eps, int generalization, float *ssg, float *ssg2,
                    double *row_scale, ssg_stream_t stream) {
  return map_forward_impl(img, img2, B, C, H, W, edges, tile_order, rank_map, fwd_plan, n_edges_dev, n_rows, ks, kw,
                          sigma, eps, generalization, ssg, ssg2, row_scale, false, stream);
}

size_t ssg_backward_scratch_bytes(int n_rows, int ks) { return split_scratch_bytes(n_rows, ks); }

int ssg_map_backward(const float *img, int B, int C, int H, int W, const int *edges, const int *tile_order,
                     const int *rank_map, const int *fwd_plan, const int *n_edges_dev, int n_rows, int ks, int kw,
                     float sigma, int generalization, const float *ssg, const float *grad_ssg, float *grad_img,
                     void *scratch, void *grad_fix, ssg_stream_t stream) {
  if (n_rows < 0 || !sizes_ok(ks, kw) || B <= 0 || C <= 0) return SSG_E_BADARG;
  if (H <= ks / 2 || W <= ks / 2) return SSG_E_IMAGESMALL;
  if (n_rows == 0) return 0;
  if (!img || !edges || !ssg || !grad_ssg || !grad_img) return SSG_E_BADARG;
  BwdParams p{};
  p.img = img;
  p.grad = grad_img;
  p.edges = edges;
  p.estride = 3;
  p.n_dev = n_edges_dev;
  p.n_host = n_rows;
  p.B = B;
  p.C = C;
  p.H = H;
  p.W = W;
  p.order = tile_order;
  p.mode = GRAD_S;
  p.gin = grad_ssg;
  p.ssg = ssg;
  p.sigma = sigma;
  p.generalization = generalization;
  p.ks = ks;
  p.kw = kw;
  p.dbg = (dbg_mask() >> 8) & 0xff;
  int rc = det_begin(p, grad_fix, (hipStream_t)stream);
  if (rc) return rc;
  if (split_ok(ks, kw, C, rank_map, fwd_plan, scratch)) {
    rc = split_backward(p, rank_map, fwd_plan, scratch, (hipStream_t)stream);
  } else {
    if (p.gfix) rc = launch_grad_fix_bound(p, (hipStream_t)stream);
    if (!rc) rc = launch_bwd(p, (hipStream_t)stream);
  }
  return rc ? rc : det_end(p, (hipStream_t)stream);
}

size_t ssg_grad_fix_bytes(int B, int C, int H, int W) { return sizeof(long long) * ((size_t)B * C * H * W + 8); }

static size_t partials_bytes(int B, int H, int W, int n_rows) {
  return align_up(2 * sizeof(float) * bwd_max_partials(B, H, W, n_rows) + 64, 256);
}

size_t ssg_loss_scratch_bytes(int B, int H, int W, int n_rows, int ks) {
  return partials_bytes(B, H, W, n_rows) + split_scratch_bytes(n_rows, ks);
}

static int loss_backward(const float *sr, int B, int C, int H, int W, const int *edges, const int *tile_order,
                         const int *rank_map, const int *fwd_plan, const int *n_edges_dev, int n_rows, int ks, int kw,
                         float sigma, int generalization, float *ssg_sr, float *ssg_gt, float w_l1, float w_kl,
                         const float *upstream, float *loss_out, float *grad_sr, void *scratch, void *grad_fix,
                         const double *row_scale, bool rows_scratch, bool fix_zeroed, bool grad_is_output,
                         ssg_stream_t stream, const TileMajor *tm = nullptr, bool nan_on_overflow = false,
                         ForkChain *chain = nullptr) {
  if (n_rows < 0 || !sizes_ok(ks, kw) || B <= 0 || C <= 0 || !loss_out) return SSG_E_BADARG;
  if (H <= ks / 2 || W <= ks / 2) return SSG_E_IMAGESMALL;
  hipStream_t st = (hipStream_t)stream;
  // (a forward that left its streams forked is joined on every way out)
  auto leave = [&](int rc0) {
    if (chain && chain->active) {
      const int rcj = join_to(st, chain->fk);
      chain->active = false;
      return rc0 ? rc0 : rcj;
    }
    return rc0;
  };
  if (n_rows == 0) return leave((int)hipMemsetAsync(loss_out, 0, 2 * sizeof(float), st));
  if (!sr || !edges || !ssg_sr || !ssg_gt || !scratch) return leave(SSG_E_BADARG);
  BwdParams p{};
  p.img = sr;
  p.grad = grad_sr;
  p.edges = edges;
  p.estride = 3;
  p.n_dev = n_edges_dev;
  p.n_host = n_rows;
  p.B = B;
  p.C = C;
  p.H = H;
  p.W = W;
  p.order = tile_order;
  p.mode = GRAD_LOSS;
  p.ssg = ssg_sr;
  p.ssg2 = ssg_gt;
  p.sigma = sigma;
  p.generalization = generalization;
  p.w_l1 = w_l1;
  p.w_kl = w_kl;
  p.partials = (float *)scratch;
  p.upstream = upstream;
  p.ks = ks;
  p.kw = kw;
  p.dbg = (dbg_mask() >> 8) & 0xff;
  p.row_scale = row_scale;
  p.rows_scratch = rows_scratch ? 1 : 0;
  if (row_scale && !split_ok(ks, kw, C, rank_map, fwd_plan, scratch)) return leave(SSG_E_BADARG);  // only ssg_grad_rows rescales
  int rc = det_begin(p, grad_fix, st, fix_zeroed);
  if (rc) return leave(rc);
  int nparts;
  bool fin_done = false;
  const bool split = split_ok(ks, kw, C, rank_map, fwd_plan, scratch);
  // (per-class row passes: two sets of grow_grid(n_rows) slots of criteria sums)
  nparts = split ? (split_row_classes(p, split_tm_tiles(p, tm)) ? 2 : 1) * (int)grow_grid(n_rows) + rows_tm_parts(split_tm_tiles(p, tm)) : 0;
  // (ssg_grad_rows' slots come in sets of grow_grid(n_rows), live up to the device's row count: LossFinalize::set_size)
  const int set_size = split && split_tm_tiles(p, tm) == 0 ? (int)grow_grid(n_rows) : 0;
  if (split) {
    const FinalizeArgs fin{p.partials, nparts, n_edges_dev, n_rows, ks * ks, w_l1, w_kl, loss_out, nan_on_overflow ? 1 : 0, set_size};
    rc = split_backward(p, rank_map, fwd_plan, (char *)scratch + partials_bytes(B, H, W, n_rows), st, &fin, &fin_done, tm,
                        chain);
  } else {
    rc = leave(0);
    p.fix_inline = p.gfix ? 1 : 0;   // (GRAD_LOSS: the backward kernel derives the a-priori scale itself -- one launch less)
    if (!rc) rc = launch_bwd(p, st);
    nparts = (int)bwd_grid(p);
  }
  const FinalizeArgs fin{p.partials, nparts, n_edges_dev, n_rows, ks * ks, w_l1, w_kl, loss_out, nan_on_overflow ? 1 : 0, set_size};
  // (the finalize rides in the flush's last workgroup when its partial sums are few -- 256 threads play its 1,024 lanes: C5's
  // 70 k partials took 30 us there against 10 + 11 as two launches)
  // (sets of slots are read up to the device's row count only: what counts is their live prefix, bounded by the host's)
  const int fin_reads = set_size > 0 ? (nparts / set_size) * ((n_rows + 3) / 4) : nparts;
  if (!rc) rc = det_end(p, st, grad_is_output, (fin_done || fin_reads > 8192) ? nullptr : &fin, &fin_done);
  if (rc || fin_done) return rc;
  return launch_loss_finalize(p.partials, nparts, n_edges_dev, n_rows, ks * ks, w_l1, w_kl, loss_out, nan_on_overflow ? 1 : 0, st,
                              set_size);
}

int ssg_loss_backward(const float *sr, int B, int C, int H, int W, const int *edges, const int *tile_order,
                      const int *rank_map, const int *fwd_plan, const int *n_edges_dev, int n_rows, int ks, int kw,
                      float sigma, int generalization,
                      float *ssg_sr, float *ssg_gt, float w_l1, float w_kl, const float *upstream,
                      float *loss_out, float *grad_sr, void *scratch, void *grad_fix, const double *row_scale,
                      int rows_are_scratch, ssg_stream_t stream) {
  return loss_backward(sr, B, C, H, W, edges, tile_order, rank_map, fwd_plan, n_edges_dev, n_rows, ks, kw, sigma,
                       generalization, ssg_sr, ssg_gt, w_l1, w_kl, upstream, loss_out, grad_sr, scratch, grad_fix,
                       row_scale, rows_are_scratch != 0, false, false, stream);
}

// two row-major regions (sr, gt); at k_s = 49 two tile-major regions of the same size behind them
static size_t rows_region_bytes(int capacity, int ks) {
  return align_up(sizeof(float) * (size_t)(capacity > 0 ? capacity : 1) * ks * ks, 256);
}
// (a tile-major region: capacity / 128 slots and a spare one for the short strips of ssg_fwd_strip)
static size_t tm_region_bytes(int capacity, int ks) {
  return align_up(sizeof(float) * ((size_t)(capacity > 0 ? capacity : 1) / TM_PX + 1) * ks * ks * TM_PX, 256);
}
size_t ssg_loss_tm_bytes(int capacity, int ks) { return ks == 49 ? 2 * tm_region_bytes(capacity, ks) : 0; }
size_t ssg_loss_rows_bytes(int capacity, int ks) {
  return 2 * rows_region_bytes(capacity, ks) + ssg_loss_tm_bytes(capacity, ks);
}

size_t ssg_loss_workspace_bytes(int B, int H, int W, int capacity, int ks) {
  return align_up(sizeof(int) * 3 * (size_t)(capacity > 0 ? capacity : 1), 256) +
         align_up(2 * sizeof(double) * (size_t)(capacity > 0 ? capacity : 1), 256) +
         align_up(sizeof(int) * (size_t)B * H * W, 256) + align_up(sizeof(int) * (size_t)(capacity > 0 ? capacity : 1), 256) +
         align_up(fwd_plan_bytes(B, H, W, capacity), 256) + align_up(edge_scratch_bytes(B, H, W), 256) +
         align_up(ssg_loss_scratch_bytes(B, H, W, capacity, ks), 256);
}

// carving of the fused call's workspace (one place: loss_fwd_bwd_impl and ssg_loss_workspace_layout use it)
struct LossWorkspace {
  size_t edges, rank, order, plan, escratch, lscratch, row_scale, base_bytes, rows[2], tm[2];
  int tm_slots;
};
static LossWorkspace carve_workspace(int B, int H, int W, int capacity, int ks, bool fused = true) {
  LossWorkspace w{};
  size_t o = 0;
  w.edges = o;
  o += align_up(sizeof(int) * 3 * (size_t)capacity, 256);
  w.rank = o;
  o += align_up(sizeof(int) * (size_t)B * H * W, 256);
  w.order = o;
  o += align_up(sizeof(int) * (size_t)capacity, 256);
  w.plan = o;
  o += align_up(fwd_plan_bytes(B, H, W, capacity), 256);
  w.escratch = o;
  o += align_up(edge_scratch_bytes(B, H, W), 256);
  w.lscratch = o;
  o += align_up(ssg_loss_scratch_bytes(B, H, W, capacity, ks), 256);
  w.row_scale = o;
  w.base_bytes = ssg_loss_workspace_bytes(B, H, W, capacity, ks);
  const size_t region = rows_region_bytes(capacity, ks);
  w.rows[0] = w.base_bytes;
  w.rows[1] = w.rows[0] + region;
  if (ks == 49) {   // (a materialising call has no row-major scratch rows: the tile-major regions follow the base)
    w.tm[0] = fused ? w.rows[1] + region : w.base_bytes;
    w.tm[1] = w.tm[0] + tm_region_bytes(capacity, ks);
    w.tm_slots = capacity / TM_PX;
  }
  if (!fused) w.rows[0] = w.rows[1] = 0;
  return w;
}

int ssg_loss_workspace_layout(int B, int H, int W, int capacity, int ks, int fused, size_t out[9]) {
  if (!out || B <= 0 || H <= 0 || W <= 0 || capacity <= 0 || ks <= 0) return SSG_E_BADARG;
  const LossWorkspace w = carve_workspace(B, H, W, capacity, ks, fused != 0);
  out[0] = w.edges;
  out[1] = w.rank;
  out[2] = w.plan;
  out[3] = w.row_scale;
  out[4] = w.rows[0];
  out[5] = w.rows[1];
  out[6] = w.tm[0];
  out[7] = w.tm[1];
  out[8] = (size_t)w.tm_slots;
  return 0;
}

static int loss_fwd_bwd_impl(const float *sr, const float *gt, const void *mask, int mask_kind, int mask_channels, int B,
                             int C, int H, int W, int ks, int kw, float sigma, float eps, int generalization, float w_l1,
                             float w_kl, int mask_stride, float lap_threshold, int capacity, float *ssg_sr,
                             float *ssg_gt, int *counts, float *loss_out, float *grad_sr, void *workspace,
                             size_t workspace_bytes, void *grad_fix, bool grad_is_output, ssg_stream_t stream) {
  if (!sr || !gt || !counts || !loss_out || !workspace || capacity <= 0) return SSG_E_BADARG;
  if ((ssg_sr == nullptr) != (ssg_gt == nullptr)) return SSG_E_BADARG;
  if (mask_kind != 2 && !mask) return SSG_E_BADARG;
  // fused step (no SSG output): the rows are the engine's scratch, behind the regular workspace
  const bool fused = ssg_sr == nullptr;
  const size_t base_bytes = ssg_loss_workspace_bytes(B, H, W, capacity, ks);
  if (workspace_bytes < base_bytes + (fused ? ssg_loss_rows_bytes(capacity, ks) : 0)) return SSG_E_WORKSPACE;
  // the edge-list builder's first kernel clears the row scales, the fixed-point sums and (ssg_loss_step) the gradient
  // with 16-byte stores: offset views of a larger buffer must keep that alignment (include/ssg_hip.h)
  if ((((uintptr_t)workspace) | ((uintptr_t)grad_fix) | ((uintptr_t)grad_sr)) & 15) return SSG_E_ALIGN;
  const LossWorkspace lw = carve_workspace(B, H, W, capacity, ks, fused);
  char *ws = (char *)workspace;
  TileMajor tm;
  if (fused) {
    ssg_sr = (float *)(ws + lw.rows[0]);
    ssg_gt = (float *)(ws + lw.rows[1]);
  }
  // tile-major rows: always in the fused step; in a materialising call when the caller's workspace has room for the
  // two regions (ssg_loss_tm_bytes) -- ssg_rows_tm_mat then writes the normalised SSG rows from them
  // (ssg_fwd_strip addresses a region with 32-bit ELEMENT offsets: regions of 2^32 floats = 16 GB and more -- from 1.78 M
  // rows per call -- stay on row-major rows)
  if (ks == 49 && kw == 13 && C == 3 && generalization && tile_major_enabled() &&
      (size_t)(lw.tm_slots + 1) * (size_t)(ks * ks) * TM_PX < (1ull << 32) &&
      (fused || workspace_bytes >= base_bytes + ssg_loss_tm_bytes(capacity, ks))) {
    tm.rows[0] = (float *)(ws + lw.tm[0]);
    tm.rows[1] = (float *)(ws + lw.tm[1]);
    tm.slots = lw.tm_slots;
  }
  int *edges = (int *)(ws + lw.edges);
  int *rank = (int *)(ws + lw.rank);
  // small (11,5) steps: one workgroup builds the edge list, one workgroup per edge pixel does the rest (ssg_tiny.hip)
  if (g_tiny_step.load(std::memory_order_relaxed) && tiny_step_supported(ks, kw, C, capacity) && tiny_edge_list_ok(B, H, W) &&
      B > 0 && H > ks / 2 && W > ks / 2 && mask_kind >= 0 && mask_kind <= 2 && (mask_kind == 2 || mask_channels > 0)) {
    const bool fixed = grad_fix && grad_sr;
    int *ticket = (int *)(ws + lw.escratch);
    int rc = launch_tiny_edge_list(mask_kind == 2 ? (const void *)gt : mask, mask_kind, mask_kind == 2 ? 3 : mask_channels, B, H, W,
                                   mask_stride, lap_threshold, edges, capacity, counts, rank, ticket, 16,
                                   fixed ? grad_fix : nullptr, sizeof(long long) * ((size_t)B * C * H * W + 8),
                                   (grad_is_output && grad_sr && !fixed) ? (void *)grad_sr : nullptr,
                                   sizeof(float) * (size_t)B * C * H * W, (hipStream_t)stream);
    if (rc) return rc;
    TinyParams t{};
    t.img[0] = sr;
    t.img[1] = gt;
    t.out[0] = fused ? nullptr : ssg_sr;
    t.out[1] = fused ? nullptr : ssg_gt;
    t.edges = edges;
    t.n_dev = counts;
    t.n_host = capacity;
    t.B = B;
    t.H = H;
    t.W = W;
    t.sigma = sigma;
    t.eps = eps;
    t.generalization = generalization;
    t.w_l1 = w_l1;
    t.w_kl = w_kl;
    t.grad = grad_sr;
    t.gfix = fixed ? (long long *)grad_fix : nullptr;
    t.assign = grad_is_output ? 1 : 0;
    t.partials = (float *)(ws + lw.lscratch);
    t.loss_out = loss_out;
    t.ticket = ticket;
    t.nan_on_overflow = 1;
    t.dbg = env_int("SSG_TINY_DBG", 0);
    return launch_tiny_step(t, C, (hipStream_t)stream);
  }
  int *order = (int *)(ws + lw.order);
  int *plan = (int *)(ws + lw.plan);
  void *escratch = ws + lw.escratch;
  void *lscratch = ws + lw.lscratch;
  double *row_scale = (double *)(ws + lw.row_scale);
  // (deferred normalisation wherever the split backward -- whose ssg_grad_rows pass rescales -- follows)
  const bool defer = split_ok(ks, kw, C, rank, plan, lscratch) && dense_supported(ks, kw, C);
  if (!defer) tm.slots = 0;
  // with the plan in use every kernel takes its job order from it: the full tile-major order is not built (3 launches)
  if (defer) order = nullptr;
  // kernel sizes without shared-term kernels ((11,5), other channel counts) have no use for a plan: not built
  // (4 launches of 4-5 us each: BASELINE's C1 step is 14 dependent launches long)
  if (!dense_supported(ks, kw, C)) plan = nullptr;
  // the row scales and the fixed-point gradient sums start at zero: cleared by the edge-list builder's first kernel
  // (16-byte granules: both sizes are multiples of 16)
  const bool zero_fix = grad_fix && grad_sr;
  // two chains (ForkChain): sizes with a dense / direct split and a side stream to put one of them on
  // (under capture only the free-running chains: a GATED pair replays badly, see stream_capturing)
  const bool free_wanted = two_chains_wanted();   // (one read of the asynchronously written hint per decision)
  const bool two_chains = defer && grad_sr && ks <= 25 && overlap_enabled() && two_chains_allowed() &&
                          (free_wanted || !stream_capturing((hipStream_t)stream));
  const bool free_running = two_chains && free_wanted;
  const size_t fix_bytes = sizeof(long long) * ((size_t)B * C * H * W + 8), rs_bytes = 2 * sizeof(double) * (size_t)capacity;
  int rc = edge_list_impl(mask_kind == 2 ? (const void *)gt : mask, mask_kind, mask_kind == 2 ? 3 : mask_channels, B, H,
                          W, mask_stride, lap_threshold, ks, edges, capacity, counts, rank, order, plan, escratch,
                          defer ? (void *)row_scale : nullptr, rs_bytes, zero_fix ? grad_fix : nullptr, fix_bytes,
                          // a gradient that is an OUTPUT: the deterministic flush assigns it; with fp32 atomics it is
                          // cleared here (whole 16-byte granules; a tail of < 16 bytes by the memset below)
                          (grad_is_output && grad_sr && !zero_fix) ? (void *)grad_sr : nullptr,
                          sizeof(float) * (size_t)B * C * H * W, stream);
  if (!rc && grad_is_output && grad_sr && !zero_fix && ((sizeof(float) * (size_t)B * C * H * W) & 15))
    rc = (int)hipMemsetAsync((char *)grad_sr + ((sizeof(float) * (size_t)B * C * H * W) & ~(size_t)15), 0,
                             (sizeof(float) * (size_t)B * C * H * W) & 15, (hipStream_t)stream);
  if (rc) return rc;
  ForkChain chain;
  chain.gated = !free_running;
  rc = map_forward_impl(sr, gt, B, C, H, W, edges, order, rank, plan, counts, capacity, ks, kw, sigma, eps,
                        generalization, ssg_sr, ssg_gt, defer ? row_scale : nullptr, defer, stream, &tm,
                        two_chains ? &chain : nullptr);
  if (rc) return rc;
  return loss_backward(sr, B, C, H, W, edges, order, rank, plan, counts, capacity, ks, kw, sigma, generalization,
                       ssg_sr, ssg_gt, w_l1, w_kl, nullptr, loss_out, grad_sr, lscratch, grad_fix,
                       defer ? row_scale : nullptr, fused, zero_fix, grad_is_output && zero_fix, stream, &tm, true,
                       two_chains ? &chain : nullptr);
}

int ssg_loss_fwd_bwd(const float *sr, const float *gt, const void *mask, int mask_kind, int mask_channels, int B,
                     int C, int H, int W, int ks, int kw, float sigma, float eps, int generalization, float w_l1,
                     float w_kl, int mask_stride, float lap_threshold, int capacity, float *ssg_sr, float *ssg_gt,
                     int *counts, float *loss_out, float *grad_sr, void *workspace, size_t workspace_bytes,
                     void *grad_fix, ssg_stream_t stream) {
  return loss_fwd_bwd_impl(sr, gt, mask, mask_kind, mask_channels, B, C, H, W, ks, kw, sigma, eps, generalization, w_l1,
                           w_kl, mask_stride, lap_threshold, capacity, ssg_sr, ssg_gt, counts, loss_out, grad_sr,
                           workspace, workspace_bytes, grad_fix, false, stream);
}

int ssg_loss_step(const float *sr, const float *gt, const void *mask, int mask_kind, int mask_channels, int B, int C,
                  int H, int W, int ks, int kw, float sigma, float eps, int generalization, float w_l1, float w_kl,
                  int mask_stride, float lap_threshold, int capacity, float *ssg_sr, float *ssg_gt, int *counts,
                  float *loss_out, float *grad_sr, void *workspace, size_t workspace_bytes, void *grad_fix,
                  ssg_stream_t stream) {
  return loss_fwd_bwd_impl(sr, gt, mask, mask_kind, mask_channels, B, C, H, W, ks, kw, sigma, eps, generalization, w_l1,
                           w_kl, mask_stride, lap_threshold, capacity, ssg_sr, ssg_gt, counts, loss_out, grad_sr,
                           workspace, workspace_bytes, grad_fix, true, stream);
}

int ssg_augment_crop(const void *src, void *dst, int elem_bytes, int B, int C, int Hs, int Ws, int Ho, int Wo,
                     const int *params, ssg_stream_t stream) {
  if (!src || !dst || !params || B < 0 || C <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 ||
      (elem_bytes != 1 && elem_bytes != 4))
    return SSG_E_BADARG;
  return launch_augment_crop(src, dst, elem_bytes, B, C, Hs, Ws, Ho, Wo, params, (hipStream_t)stream);
}

int ssg_pool_swap(void *queue, void *batch, size_t sample_bytes, const int *slots, int b, ssg_stream_t stream) {
  if (!queue || !batch || !slots || b < 0) return SSG_E_BADARG;
  return launch_pool_swap(queue, batch, sample_bytes, slots, b, (hipStream_t)stream);
}

size_t ssg_usm_scratch_bytes(int B, int C, int H, int W) { return usm_scratch_bytes(B, C, H, W); }

int ssg_usm_sharp(const float *img, float *out, int B, int C, int H, int W, int radius, float sigma, float weight,
                  float threshold, void *scratch, size_t scratch_bytes, ssg_stream_t stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || radius < 0) return SSG_E_BADARG;
  if (B == 0) return 0;
  if (!img || !out || !scratch || img == out) return SSG_E_BADARG;
  if (scratch_bytes < usm_scratch_bytes(B, C, H, W)) return SSG_E_WORKSPACE;
  const int ksize = radius % 2 == 0 ? radius + 1 : radius;   // img_process_util.py:67-68
  const int rc = launch_usm_sharp(img, out, B, C, H, W, ksize, sigma, weight, threshold, scratch, (hipStream_t)stream);
  return rc == -1 ? SSG_E_BADARG : rc == -4 ? SSG_E_IMAGESMALL : rc;
}

int ssg_filter2d(const float *img, const float *kernels, float *out, int B, int C, int H, int W, int k, int n_kernels,
                 ssg_stream_t stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0) return SSG_E_BADARG;
  if (B == 0) return 0;
  if (!img || !kernels || !out || img == out) return SSG_E_BADARG;
  const int rc = launch_filter2d(img, kernels, out, B, C, H, W, k, n_kernels, (hipStream_t)stream);
  return rc == -1 ? SSG_E_BADARG : rc == -4 ? SSG_E_IMAGESMALL : rc;
}

int ssg_diffjpeg(const float *img, float *out, int B, int H, int W, const float *quality_dev, float quality,
                 ssg_stream_t stream) {
  if (B < 0 || H <= 0 || W <= 0) return SSG_E_BADARG;
  if (B == 0) return 0;
  if (!img || !out || (!quality_dev && !(quality > 0.f))) return SSG_E_BADARG;
  return launch_jpeg(img, out, B, H, W, quality_dev, quality, (hipStream_t)stream);
}

int ssg_operator_pool_trim(void) {
  hipMemPool_t pool = op_pool();
  return pool ? (int)hipMemPoolTrimTo(pool, 0) : 0;
}

int ssg_set_operator_plan_threshold(int positions) {
  const int prev = op_plan_from();
  g_op_plan_from.store(positions > 0 ? positions : 0x7fffffff, std::memory_order_relaxed);
  return prev;
}

size_t ssg_criteria_scratch_bytes(void) { return criteria_scratch_bytes(); }

int ssg_criteria_sums(const float *pred, const float *target, size_t n, void *scratch, float *sums_out,
                      ssg_stream_t stream) {
  if (!sums_out || !scratch) return SSG_E_BADARG;
  if (n == 0) return (int)hipMemsetAsync(sums_out, 0, 2 * sizeof(float), (hipStream_t)stream);
  if (!pred || !target) return SSG_E_BADARG;
  return launch_criteria_sums(pred, target, n, scratch, sums_out, (hipStream_t)stream);
}

int ssg_criteria_grad(const float *pred, const float *target, size_t n, const float *coef, float *grad_pred,
                      ssg_stream_t stream) {
  if (n == 0) return 0;
  if (!pred || !target || !coef || !grad_pred) return SSG_E_BADARG;
  return launch_criteria_grad(pred, target, n, coef, grad_pred, (hipStream_t)stream);
}

int ssg_device_status(ssg_stream_t stream) {
  int *w = device_status_word();
  if (!w) return 0;
  int v = 0;
  int rc = (int)hipMemcpyAsync(&v, w, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (!rc && v) rc = (int)hipMemsetAsync(w, 0, sizeof(int), (hipStream_t)stream);
  if (!rc) rc = (int)hipStreamSynchronize((hipStream_t)stream);
  if (rc) return rc;
  return (v & 1) ? SSG_E_PLAN : 0;
}

const char *ssg_kernel_name(int ks, int kw, int backward) {
  if (!backward && ks == 25 && kw == 9 && dense_threshold() > 0)
    return "ssg_fwd_dense<25,9,3>+ssg_fwd_tiled<Geo<25,9,5,128>,merged|single>";
  return backward ? bwd_kernel_name(ks, kw) : fwd_kernel_name(ks, kw);
}

}  // extern "C"
