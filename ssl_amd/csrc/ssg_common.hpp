// Shared device/host definitions of the gfx950 SSG engine (see include/ssg_hip.h).
//
// Work decomposition used by every tiled kernel ("job" = one edge pixel of one
// image): the k_s x k_s grid of search offsets is cut into NB x NB blocks of
// BS x BS offsets; one lane owns one block, so a job occupies LPJ = NB*NB lanes
// and a workgroup of WG lanes carries JOBS = WG / LPJ jobs.  For the paper's
// k_s = 25 that is 5x5 blocks of 5x5 offsets: 25 lanes per edge pixel, five
// edge pixels per 128-lane workgroup (125/128 lanes busy).  A lane walks the
// (BS+k_w-1)^2 input patch its block needs row by row; every LDS value it
// loads feeds up to BS*k_w FMAs, which keeps the kernel on the fp32 VALU
// roofline instead of LDS bandwidth (MI355X: 128 lane-ops vs 32 LDS dwords per
// clock per CU).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>
#include <utility>

// Profiling ablations (kernel phases switched off, results then WRONG) exist only in the profiling build
// (-DSSG_PROFILE -> libssg_hip_prof.so, loaded by bench.py's per-kernel timings and tools/); in the product
// library every SSG_DBG(...) is the constant 0 and the branches fold away.
#ifdef SSG_PROFILE
#define SSG_DBG(p, bits) ((p).dbg & (bits))
#else
#define SSG_DBG(p, bits) 0
#endif

namespace ssg {

// F.pad(mode='reflect') index map (border sample not duplicated).
__device__ __forceinline__ int reflect_idx(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// cv2 BORDER_REFLECT_101 (identical map; kept separate for the n == 1 corner).
__device__ __forceinline__ int reflect101_idx(int i, int n) {
  if (n == 1) return 0;
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

template <int KS_, int KW_, int BS_, int WG_>
struct Geo {
  static constexpr int KS = KS_, KW = KW_, BS = BS_, WG = WG_;
  static constexpr int HP = KS / 2, HK = KW / 2;
  static constexpr int P = KS * KS;
  static constexpr int NB = (KS + BS - 1) / BS;  // blocks per side
  static constexpr int LPJ = NB * NB;            // lanes per job
  static constexpr int JOBS = WG / LPJ;          // jobs per workgroup
  static constexpr int PW = BS + KW - 1;         // input patch side per lane
  static constexpr int S = KS + 1;               // LDS row stride of a tile (floats)
  static constexpr int CH = KS * S;              // LDS channel stride
  static constexpr int ZROW = NB * BS + 2 * HK;  // length of the all-zero row
  // Backward: row stride of a job's G tile (KS values, then >= HK zeros).  The 25 lanes of a job read rows 5 by + r at
  // columns 5 bx + j: with KS + HK = 29 two of them share a bank (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.49 for
  // ssg_bwd_tiled<Geo<25,9,5,128>>), with 37 the 25 addresses 185 by + 5 bx fall on 25 different banks.
  static constexpr int GS = (KS == 25 && KW == 9) ? 37 : KS + HK;
  static_assert(JOBS >= 1, "workgroup too small for one job");
  static_assert(KW <= KS && (KS & 1) && (KW & 1), "odd sizes, k_w <= k_s");
};

// Compiler fence for the fully unrolled row-streaming loops.  hipcc otherwise sinks
// the arithmetic of every patch row below all rows' LDS loads (sched_barrier does not
// order pure arithmetic), which keeps the whole (BS+k_w-1)^2 patch live and spills.
// Passing every accumulator through an empty asm with a memory clobber pins row r's
// FMAs above it and row r+2's loads below it, at zero instruction cost.
template <int N>
__device__ __forceinline__ void pin_row(float (&v)[N]) {
  if constexpr (N == 3) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])::"memory");
  } else if constexpr (N == 4) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
  } else if constexpr (N == 5) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4])::"memory");
  } else if constexpr (N == 6) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])::"memory");
  } else if constexpr (N == 7) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6])::"memory");
  } else if constexpr (N == 9) {
    asm volatile(""
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                   "+v"(v[8])::"memory");
  } else if constexpr (N == 10) {
    asm volatile(""
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                   "+v"(v[8]), "+v"(v[9])::"memory");
  } else if constexpr (N == 11) {
    asm volatile(""
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                   "+v"(v[8]), "+v"(v[9]), "+v"(v[10])::"memory");
  } else if constexpr (N == 13) {
    asm volatile(""
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                   "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12])::"memory");
  } else {
    static_assert(N == 3, "add a pin_row arm for this width");
  }
}

template <int R, int N>
__device__ __forceinline__ void pin_block(float (&v)[R][N]) {
#pragma unroll
  for (int i = 0; i < R; ++i) pin_row<N>(v[i]);
}

// `order` entries: bits 0..29 = row; bit 30 of the FIRST entry of every group of ORDER_GROUP
// consecutive entries = "these jobs are one image's edge pixels within 8 rows x 16 columns".
// Dense tile list of a forward plan: hdr = plan + 1 points at {n_heavy, tile rows, n_light}; the tiles with more than
// 64 edge pixels (two or more 64-lane chunks in the forward's edge stage, i.e. the longest workgroups) are appended
// from the front of the `n_super` slots, the others from the back, and slot t walks the heavy ones first: longest jobs
// first shortens the kernel's tail (list scheduling of ~2,000 workgroups on 512 slots: -12 % makespan in a model).
// k_s 49: a STRIP of STRIP_ROWS x 32 centres = nine 4-row tiles (fewer at the bottom of the image), all of them heavy,
// is listed in the plan's strip list with the position of its first tile in the dense list -- the strip's tiles occupy
// consecutive positions there, i.e. consecutive slots of the tile-major region -- and its tiles' ids carry
// TILE_IN_STRIP: in a tile-major call ssg_fwd_strip computes their forward rows (E/H shared by 36 centre rows instead
// of 4) and the tile kernel skips them; every other consumer masks the bit off (dense_tile_id).
constexpr int STRIP_ROWS = 36;
constexpr int TILE_IN_STRIP = 1 << 30;
// 8 x 32 tiles holding more than 128 edge pixels (three or four 64-slot chunks in the forward's edge stage): the dense
// forward runs them in its four-chunk instantiation, every other tile in the two-chunk one (ssg_dense.hip)
constexpr int TILE_HUGE = 1 << 29;
__device__ __forceinline__ int dense_tile_id(int listed) { return listed & ~(TILE_IN_STRIP | TILE_HUGE); }
__device__ __forceinline__ int dense_tile_count(const int *hdr) { return hdr[0] + hdr[2]; }
__device__ __forceinline__ int dense_tile_at(const int *hdr, const int *tiles, int n_super, int tslot) {
  const int nh = hdr[0];
  return tslot < nh ? tiles[tslot] : tiles[n_super - 1 - (tslot - nh)];
}

constexpr int ORDER_GROUP = 5;
// a group of jobs is "mergeable" (one shared LDS region / gradient window) when its edge pixels are one image's and lie
// within MERGE_ROWS x MERGE_COLS pixels
constexpr int MERGE_ROWS = 8, MERGE_COLS = 16;   // (16 x 16 measured: no change at C2)
constexpr int ORDER_FLAG = 1 << 30;
constexpr int ORDER_MASK = ORDER_FLAG - 1;

struct Edge {
  int b, y, x;
};

// edges rows are (b,y,x) (stride 3) or the reference op's (Y,X) (stride 2, b = 0).
__device__ __forceinline__ Edge load_edge(const int *edges, int stride, int n) {
  const int *e = edges + (size_t)n * stride;
  Edge r;
  r.b = stride == 3 ? e[0] : 0;
  r.y = e[stride - 2];
  r.x = e[stride - 1];
  return r;
}

// Parameters shared by the forward kernels.
struct FwdParams {
  const float *img[2];  // (B,C,H,W) each; img[1] may be null
  float *out[2];        // (n, KS*KS) each
  int nimg;
  const int *edges;
  int estride;       // 3: (b,y,x)   2: (Y,X)
  const int *order;  // nullable (n): tile-major permutation of the rows -> image-major job order
  const int *n_dev;  // nullable device row count
  int n_host;        // host bound on rows
  int B, C, H, W;
  float sigma, eps;
  int generalization;
  int raw;  // 1: out += D (reference operator)   0: SSG epilogue
  int ks, kw;  // used by the generic kernel only
  int dbg;     // profiling ablations (0 in production): bit0 skip fill, bit1 skip compute, bit2 skip epilogue/store
  int small;   // direct forward, set by its launcher: which variants of a (25,9) launch run (ssg_fwd.hip, small_call)
  // host-side only: rows the direct kernels are EXPECTED to get (0 = unknown) -- the sparse-row count of the last plan
  // built on the device (ssg_api.hip: PlanHint).  Sizes the main grid; the looping tail kernel behind it takes whatever
  // the hint missed, so a stale hint costs time, never rows.
  int rows_hint;
};

// How the backward kernel obtains G = dL/dD for a job.
enum GradMode : int {
  GRAD_D = 0,    // gin = dL/dD                       (reference operator backward)
  GRAD_S = 1,    // gin = dL/dS, ssg = S saved         (similarity_map autograd)
  GRAD_LOSS = 2  // ssg = S_sr, ssg2 = S_gt: L1 + KL   (fused loss)
};

struct BwdParams {
  const float *img;  // (B,C,H,W)
  float *grad;       // (B,C,H,W), accumulated with fp32 atomics (may be null in GRAD_LOSS: loss only)
  long long *gfix;   // nullable (B,C,H,W): deterministic mode, the same sums in 2^-40 fixed point (integer atomics)
  int fix_inline;    // GRAD_LOSS on the direct-only path: every workgroup derives the scale from loss_grad_bound() itself
                     // and the first one leaves the bound in the word behind the sums for the flush (no bound launch)
  const int *edges;
  int estride;
  const int *order;  // nullable (n) int32: job k works on row order[k] (tile-major permutation of the rows)
  const int *n_dev;
  int n_host;
  int B, C, H, W;
  int mode;
  const float *gin;   // GRAD_D / GRAD_S
  const float *ssg;   // GRAD_S / GRAD_LOSS
  const float *ssg2;  // GRAD_LOSS
  float sigma;
  int generalization;
  float w_l1, w_kl;
  const float *upstream;  // GRAD_LOSS, nullable: device {dL/dl1, dL/dkl}
  const double *row_scale;  // GRAD_LOSS, nullable: deferred normalisation (GrowParams), split backward only
  int rows_scratch;         // GRAD_LOSS: the rows are scratch, ssg_grad_rows does not write the rescaled rows back
  float *partials;  // GRAD_LOSS: (gridDim.x, 2) per-workgroup sums of |a-b| and t'(log t' - log s')
  int ks, kw;       // generic kernel only
  int dbg;          // profiling ablations (0 in production): bit0 skip prologue math, bit1 skip pass A, bit2 skip pass B, bit3 skip atomics
  int rows_hint;    // host-side only: see FwdParams::rows_hint
};

// dL/dS of the two criteria at one element (a = s_sr, b = s_gt), and the
// criteria's un-normalised terms.
__device__ __forceinline__ float criteria_elem(float a, float b, float w1m, float w2m, float &l1, float &kl) {
  const float cl = 1e-10f;
  const float ac = fmaxf(a, cl), bc = fmaxf(b, cl);
  l1 += fabsf(a - b);
  // t (log t - log s) is evaluated as t log(t/s).  Both arguments are clamped to [1e-10, 1], so the ratio is a
  // normal number and the hardware reciprocal / log2 apply without the range handling of logf() and operator/;
  // v_log_f32 is relative-accurate (<= 1e-7) also next to 1 (profiles/r1_microbench_log.txt).  The KL sum
  // cancels to second order where s ~ t, so a *systematic* relative error eps of the ratio would add eps to
  // every row (v_rcp_f32 alone: 2e-5 of the loss on fixture F1): one Newton step makes the quotient correctly
  // rounded, i.e. unbiased.
  const float rc = __builtin_amdgcn_rcpf(ac);
  const float r0 = bc * rc;
  const float ratio = __builtin_fmaf(__builtin_fmaf(-r0, ac, bc), rc, r0);
  kl += bc * (0.69314718056f * __builtin_amdgcn_logf(ratio));
  float g = a > b ? w1m : (a < b ? -w1m : 0.f);
  if (a >= cl && w2m != 0.f) g -= w2m * ratio;   // (an L1-only gradient never sees the ratio: it may overflow on foreign tensors)
  return g;
}

// The same on tensors the engine did not produce (L1Loss / KLDistanceLoss on arbitrary fp32 GPU tensors, basic_loss.py):
// a NaN in either operand reaches both sums and the gradient, as it does through torch.clamp / F.kl_div / sign
// (v_max_f32 would drop it: fmaxf(NaN, 1e-10) = 1e-10).
__device__ __forceinline__ float criteria_elem_any(float a, float b, float w1m, float w2m, float &l1, float &kl) {
  float g = criteria_elem(a, b, w1m, w2m, l1, kl);
  if (a != a || b != b) {
    const float qnan = __builtin_nanf("");
    kl = qnan;
    g = qnan;
  }
  return g;
}

// arguments of the loss finalize (ssg_bwd.hip: ssg_loss_finalize, or the tail of grad_fix_flush)
struct LossFinalize {
  const float *partials;
  int nparts;
  const int *n_dev;
  int n_host, P;
  float w_l1, w_kl;
  float *loss_out;
  int nan_on_overflow;
  // > 0: the slots are sets of `set_size` (ssg_grad_rows' passes: one slot per 4 rows, sized by the HOST's bound on the
  // rows), of which only the first ceil(rows / 4) by the DEVICE count can be non-zero: the rest is not read (a generous
  // capacity costs nothing here).  0: all nparts slots are read.
  int set_size;
};

// Environment switches exist in the PROFILING build only (libssg_hip_prof.so, -DSSG_PROFILE: the A/B measurements of
// tools/).  The product library never reads the environment: what it does is set through the C ABI
// (ssg_set_dense_threshold, ssg_set_operator_plan_threshold, ssg_set_overlap) or not at all, so that every path it can
// take is one a test can reach (tests/test_cpu_host.py checks the binary for "SSG_" strings).
inline int env_int(const char *name, int dflt) {
#ifdef SSG_PROFILE
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel,
// device) -- `done` is the caller's per-kernel bit set of devices -- from any host thread, and report a failure
// instead of launching into it.
template <class K>
inline int ensure_dynamic_lds(K kernel, int bytes, std::atomic<unsigned long long> &done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

// One contribution to the image gradient.  Default: hardware fp32 atomic (the order of the additions, hence the
// last bits of the result, varies from run to run -- like the reference's atomicAdd, similarity.cu:123-128).
// Deterministic mode (gfix != nullptr): the contribution is rounded to a multiple of 1/scale and added with a 64-bit
// INTEGER atomic; integer addition is associative, so the sum does not depend on the order and two runs agree bit
// for bit.  The scale is a power of two derived ON THE DEVICE from an upper bound of |G| = |dL/dD| (exact maximum
// from ssg_grad_rows for ssg_map_backward and the tile-major step; the a-priori bound of the loss steps otherwise), stored as float bits in the word after the
// n = B*C*H*W sums: |G| * scale < 2^36.  A pixel collects fewer than 2^21 terms 2 G d (2 k_w^2 k_s^2 = 8.1e5 for
// (49,13) under a dense mask, 1.0e5 for (25,9)), so the sum stays below 2^36 * 2^21 * 2 |d| = 2^58 |d|: no wrap for
// pixel differences up to 16 -- the loss is applied to the un-clamped generator output, whose differences exceed 1
// early in training but not by that much.  Resolution: 2^-35 of the largest |G|, 11 bits finer than an fp32 sum of
// the same terms.  (Round 2 used 2^-38, which left a factor 4 for |d| at (49,13).)
__device__ __forceinline__ float grad_fix_scale_of(unsigned bound_bits) {
  unsigned e = (bound_bits >> 23) & 0xffu;   // biased exponent of the bound: bound < 2^(e-126)
  e = e < 40u ? 40u : e;
  return __uint_as_float((289u - e) << 23);  // 2^(35 - (e - 127))
}
__device__ __forceinline__ float grad_fix_scale(const long long *gfix, size_t n) {
  if (!gfix) return 1.f;
  return grad_fix_scale_of(*(const unsigned *)(gfix + n));
}
// a-priori bound of |G| of a loss step (GRAD_LOSS): |s g| <= w1m + w2m (s, t <= 1) and |sum g s| likewise -> 4 kfac (w1m +
// w2m) with a factor 2 to spare; kfac = 1 / (sigma C k_w^2), w.m = |w.| u. / (n k_s^2)
__device__ __forceinline__ float loss_grad_bound(float sigma, int C, int kw, float w_l1, float w_kl, const float *upstream,
                                                 int nrows, int P) {
  const float kfac = 1.f / (sigma * (float)(C * kw * kw));
  const float invM = 1.f / ((float)(nrows > 0 ? nrows : 1) * (float)P);
  const float u1 = upstream ? fabsf(upstream[0]) : 1.f, u2 = upstream ? fabsf(upstream[1]) : 1.f;
  return 4.f * kfac * (fabsf(w_l1) * u1 + fabsf(w_kl) * u2) * invM;
}
// round(v * scale) as a 64-bit integer, round-half-even like __float2ll_rn(v * scale) (the product is exact: scale is a
// power of two), through the double "magic number": x + 1.5 * 2^52 leaves the integer in the low mantissa bits for
// |x| < 2^51 -- a contribution is below 2^36 * a few dozen terms.  Four instructions (two of them fp64) instead of the
// twelve of the fp32 -> i64 conversion sequence: the conversion was 2-3 % of both backward kernels' instructions.
__device__ __forceinline__ long long fix_round(float v, float scale) {
  const double x = __builtin_fma((double)v, (double)scale, 6755399441055744.0);   // 1.5 * 2^52
  return __double_as_longlong(x) - 0x4338000000000000LL;
}
__device__ __forceinline__ void grad_add(float *grad, long long *gfix, size_t idx, float v, float scale) {
  if (gfix)
    atomicAdd((unsigned long long *)gfix + idx, (unsigned long long)fix_round(v, scale));
  else
    unsafeAtomicAdd(grad + idx, v);
}
// bound of |G| -> the word behind the sums (float bits of non-negative floats order like unsigned integers)
// (the word only grows: a wave whose bound does not exceed the value it reads -- possibly a stale, smaller one --
// has nothing to add; without this test every row's atomic queues on one address, +0.8 ms at 76 k rows)
__device__ __forceinline__ void grad_fix_bound(long long *gfix, size_t n, float bound) {
  unsigned *w = (unsigned *)(gfix + n);
  const unsigned b = __float_as_uint(bound);
  if (b > __builtin_nontemporal_load(w)) atomicMax(w, b);
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>), fully inlined
// (a `#pragma unroll` over the dense kernels' offset loops was NOT honoured: hipcc kept them rolled and
// indexed the register arrays through M0, s_set_gpr_idx_on)
template <int... Is, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// sum over the 64 lanes, every lane gets the same bits (xor butterfly: a + b == b + a)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ssg_grad_rows (ssg_grow.hip): G = dL/dD rows for the backward kernels
struct GrowParams {
  int mode;           // GradMode
  const float *gin;   // GRAD_D / GRAD_S
  const float *ssg;   // GRAD_S / GRAD_LOSS
  const float *ssg2;  // GRAD_LOSS
  // nullable [2][n_host] (GRAD_LOSS): deferred normalisation of the dense-tile forward -- a non-zero entry means the
  // row still holds e = exp(-d/sigma) and is rescaled here (and written back, normalised, through ssg / ssg2)
  const double *row_scale;
  int rows_scratch;   // 1: ssg / ssg2 are the engine's own scratch (fused step, no SSG output): nothing is written back
  const int *n_dev;
  int n_host;
  int C;
  float sigma;
  int generalization;
  float w_l1, w_kl;
  const float *upstream;  // nullable device {dL/dl1, dL/dkl}
  float *G;               // (n, P) out; nullable (loss only, or GRAD_D)
  float *sum_b;           // (n) out; nullable: sum of G over the offsets with a truncated window
  float *partials;        // (grid, 2), GRAD_LOSS
  float *gmax_part;       // nullable (grid): deterministic mode, every workgroup's max|G| (reduced into the word
                          // behind the fixed-point sums by grad_fix_reduce: no atomics on one address)
  int ngroups;            // groups of 4 rows (set by launch_grad_rows)
  int grid_cap;           // > 0: at most this many workgroups, each walking several groups
  // a call with a tile-major region: plan + 1 (tm_active's header), the region's slots, and the plan's list of the
  // rows that are not in a dense tile (plan[0] of them)
  const int *tm_hdr;
  int tm_slots;
  const int *sparse_order;
  // two-chain step (ssg_api.hip): 0 every row; 1 the rows with a non-zero row scale only (the dense-tile kernels' rows,
  // un-normalised); 2 the rows of the plan's sparse list only (tm_hdr = plan + 1, sparse_order; the direct kernels' rows)
  int only;
  // nullable: the bound word behind the fixed-point sums -- the first workgroup writes the a-priori bound of |G| there
  // (GRAD_LOSS: 4 kfac (|w_l1| u_1 + |w_kl| u_2) / (n P), grad_fix_bound_kernel's) instead of every group's maximum
  unsigned *fix_word;
};

// Tile-major scratch rows (fused step at k_s = 49): the tile in slot t of the plan's dense list keeps
// e[n,q] = exp(-d/sigma) of its TM_PX = 4 x 32 pixels at  base + (t * P + q) * TM_PX + idx,  idx = 64 ck + lane with
// pixel (row 2 (lane / 32) + ck, column lane % 32) of the tile -- the dense forward's fixed pixel map -- so that every
// wave access of the three kernels that touch the rows is one aligned 256-byte run.
constexpr int TM_PX = 128;
// s = e * (1/(sum e + eps)) with the fp64 row scale as a float pair (hi, lo): e * lo brings
// back what the rounding of the scale to ONE float would lose for the whole row at once (a relative 6e-8 common to
// the 2,401 entries of a row shifts t/s - 1, i.e. the KL part of the gradient, by 1e-4 of itself at sigma = 1).
// ssg_rows_tm and the dense backward both use exactly this, so they see the same bits.
struct TmScale {
  float hi, lo;
};
__device__ __forceinline__ TmScale tm_scale(double neg_scale) {   // tile-major rows carry -1/(sum e + eps)
  const double sc = -neg_scale;
  TmScale r;
  r.hi = (float)sc;
  r.lo = (float)(sc - (double)r.hi);
  return r;
}
// (e * hi is formed exactly inside the fma, the small product joins it, ONE rounding: the correctly rounded e * scale)
__device__ __forceinline__ float tm_apply(float e, TmScale s) { return __builtin_fmaf(e, s.hi, e * s.lo); }
__device__ __forceinline__ int tm_pixel_row(int ck, int lane) { return 2 * (lane >> 5) + ck; }
__device__ __forceinline__ int tm_pixel_col(int lane) { return lane & 31; }

// A call keeps ALL its dense tiles tile-major or none (the same test in every kernel that touches the rows; the
// order of the plan's tile list varies from run to run -- atomic appends -- so a per-slot split would not be
// reproducible): when the tiles fit the region's slots and are at least 60 % full on average -- a tile-major row
// costs 128 pixels' worth of traffic whatever the number of edge pixels in the tile.
// hdr = plan + 1 (hdr[-1] = rows NOT in a dense tile), nrows = rows of the call.
__device__ __forceinline__ bool tm_active(const int *hdr, int tm_slots, int nrows) {
  const int nd = hdr[0] + hdr[2];
  return tm_slots > 0 && nd <= tm_slots && 5 * (nrows - hdr[-1]) >= 3 * TM_PX * nd;
}

// ssg_rows_tm (ssg_grow.hip): the row pass over tile-major rows -- criteria sums, sum_q g s, border sums, |G| bound
struct TmRowsParams {
  const float *tm[2];        // e rows of sr / gt
  const double *row_scale;   // [2][n_host]; tile-major rows carry -1/(sum e + eps)
  const int *rank;           // (B,H,W)
  const int *n_dense;        // plan + 1
  const int *tiles;          // plan + 4
  int n_tiles;               // launch bound = min(dense_max_tiles, tm_slots)
  int tm_slots;              // slots of the tile-major region (tm_active)
  const int *n_dev;
  int n_host;
  int B, H, W, C;
  float sigma;
  float w_l1, w_kl;
  const float *upstream;     // nullable device {dL/dl1, dL/dkl}
  float *dot;                // (n) out: sum_q g s of every tile-major row
  float *sum_b;              // (n) out: sum of G over the border offsets
  float *partials;           // (n_tiles, 2) out: criteria sums per workgroup
  float *gmax_part;          // nullable (n_tiles) out: upper bound of |G| per workgroup
  const float *out[2];       // nullable: the call's SSG tensors (n, k_s^2) -- ssg_rows_tm_mat writes the normalised rows
};

// ssg_bwd_dense (ssg_bwd_dense.hip)
struct DenseBwdParams {
  const float *img;    // (B,C,H,W)
  float *grad;         // (B,C,H,W), accumulated with fp32 atomics
  long long *gfix;     // nullable: deterministic mode (see grad_add)
  const float *G;      // (n, P) dL/dD rows
  const float *sum_b;  // (n)
  const int *rank;     // (B,H,W) row of every pixel, -1 if not an edge pixel
  const int *n_dense;  // device count of dense tiles
  const int *tiles;    // dense tile ids
  int max_tiles;       // launch bound
  const int *n_dev;    // rows computed at all (capacity clamp), nullable
  int n_host;
  int B, H, W;
  int qsplit;          // waves per tile, each taking a contiguous range of offset rows; 0 = chosen on the device from
                       // the number of dense tiles so that the launch fills the chip's wave slots about once (a wave's
                       // sweep of all k_s^2 offsets takes 0.2-0.3 ms however few tiles there are: sparse / strided
                       // masks -- BASELINE's C4 has 183 tiles -- would leave 80 % of the chip idle for that long)
  int auto_slots;      // qsplit == 0: wave slots of the device for this kernel (CUs x SIMDs x waves per SIMD)
  int dbg;
  // tile-major rows (TmRowsParams), when tm_active(): the TM variant forms G = -(s k)(g - dot) itself from the two e
  // rows, the row scales and ssg_rows_tm's dot; otherwise the other variant reads the row-major G rows
  const float *tm[2];
  const double *row_scale;
  const float *dot;
  int tm_slots;
  float sigma, w_l1, w_kl;
  const float *upstream;
  int *status;         // nullable: library-owned device status word (ssg_device_status)
};

// ssg_tiny_step (ssg_tiny.hip): the fused loss step of small (11,5) calls
struct TinyParams {
  const float *img[2];   // sr, gt (B,C,H,W)
  float *out[2];         // SSG rows (n, P) each
  const int *edges;      // (n,3)
  const int *n_dev;      // counts[0]
  int n_host;            // capacity
  int B, H, W;
  float sigma, eps;
  int generalization;
  float w_l1, w_kl;
  float *grad;           // nullable: loss only
  long long *gfix;       // nullable: fp32 atomics
  int assign;            // the gradient is an output: grad = sums
  float *partials;       // (capacity, 2)
  float *loss_out;
  int *ticket;           // zeroed by tiny_edge_list
  int nan_on_overflow;
  int dbg;               // profiling ablations (0 in production): 1 no forward sums, 2 no backward, 4 no atomics, 8 no fold, 16 no rows
};

__device__ __forceinline__ int rows_to_do(const int *n_dev, int n_host) {
  int n = n_host;
  if (n_dev) {
    const int d = *n_dev;
    n = d < n ? d : n;
  }
  return n;
}

}  // namespace ssg

#ifdef SSG_PROFILE
// Profiling build only: every launch of the library can be preceded by a kernel that fills the LDS of every CU with a
// pattern (ssg_prof_set_lds_poison; off by default).  LDS keeps what the previous workgroup on the CU left in it, so a
// kernel that reads a word it has not written normally sees plausible stale data and fails once in a blue moon (the merge
// flags of band_scatter, round 5); with the poison it sees the pattern every time.  tools/r5_poison_suite.sh runs the
// GPU suite that way.
namespace ssg { void prof_poison_lds(hipStream_t st); }
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                         \
  do {                                                                                                            \
    ::ssg::prof_poison_lds(streamId);                                                                             \
    kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                            \
  } while (0)
#endif
