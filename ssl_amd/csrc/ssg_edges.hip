// Edge list of a batch, built on device (no host round trip): which pixels of
// which images take part in the SSG loss, in the reference's order
// (image-major, then row-major == torch.where / torch.nonzero,
// loss_util.py:196, similaritywrapper.py:67), optionally generating the
// reference's offline Laplacian edge mask on the fly (generate_mask.py:22-31)
// and applying the mask_stride eye pattern (realesrganssl_model.py:64-72).
//
// Two builders.  The chunked one: per-1024-pixel chunk counts, an exclusive scan of the chunk counts (one workgroup), an
// order-preserving scatter, then the tile order / plan kernels on the rank map (7-12 launches; k_s 49 plans with their
// strips, callers that want the full order AND a plan).  The BANDED one (further down; every other call): 8-row x
// 256-column blocks give the row-segment counts and the tile counts in one pass -- three launches.
#include "ssg_common.hpp"

namespace ssg {

struct EdgeParams {
  const void *mask;
  int kind;  // 0 fp32 mask, 1 uint8 mask, 2 fp32 GT (Laplacian)
  int mask_channels;
  int B, H, W;
  int stride;
  float thr;
  int nblk_img;  // ceil(H*W / 1024)
};

constexpr int CHUNK = 1024;  // pixels per workgroup (256 lanes x 4 consecutive pixels)

// PIL convert('L') of round(255 * rgb): ITU-R 601-2 in 16.16 fixed point.
__device__ __forceinline__ int gray_l(const float *img, size_t plane, size_t off) {
  const float r = fminf(fmaxf(img[off] * 255.f, 0.f), 255.f);
  const float g = fminf(fmaxf(img[off + plane] * 255.f, 0.f), 255.f);
  const float b = fminf(fmaxf(img[off + 2 * plane] * 255.f, 0.f), 255.f);
  const unsigned ri = (unsigned)__float2int_rn(r), gi = (unsigned)__float2int_rn(g), bi = (unsigned)__float2int_rn(b);
  return (int)((ri * 19595u + gi * 38470u + bi * 7471u + 0x8000u) >> 16);
}

__device__ __forceinline__ bool edge_pred(const EdgeParams &p, int b, int y, int x) {
  if (p.stride > 1 && (y % p.stride) != (x % p.stride)) return false;
  const size_t plane = (size_t)p.H * p.W;
  if (p.kind == 0) {
    return ((const float *)p.mask)[(size_t)b * p.mask_channels * plane + (size_t)y * p.W + x] == 1.0f;
  } else if (p.kind == 1) {
    return ((const uint8_t *)p.mask)[(size_t)b * p.mask_channels * plane + (size_t)y * p.W + x] == 1;
  }
  // cv2.Laplacian(L, CV_8U): [[0,1,0],[1,-4,1],[0,1,0]], BORDER_REFLECT_101, saturate
  const float *img = (const float *)p.mask + (size_t)b * 3 * plane;
  const int ym = reflect101_idx(y - 1, p.H), yp = reflect101_idx(y + 1, p.H);
  const int xm = reflect101_idx(x - 1, p.W), xp = reflect101_idx(x + 1, p.W);
  int v = gray_l(img, plane, (size_t)ym * p.W + x) + gray_l(img, plane, (size_t)yp * p.W + x) +
          gray_l(img, plane, (size_t)y * p.W + xm) + gray_l(img, plane, (size_t)y * p.W + xp) -
          4 * gray_l(img, plane, (size_t)y * p.W + x);
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  return (float)v > p.thr;
}

// bit k of the result: pixel 4*tid+k of this chunk is an edge pixel
__device__ __forceinline__ unsigned chunk_bits(const EdgeParams &p, int &b, int &pix0) {
  const int blk = blockIdx.x;
  b = blk / p.nblk_img;
  pix0 = (blk - b * p.nblk_img) * CHUNK + 4 * threadIdx.x;
  unsigned bits = 0;
  const int HW = p.H * p.W;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pix = pix0 + k;
    if (pix < HW) {
      const int y = pix / p.W, x = pix - y * p.W;
      if (edge_pred(p, b, y, x)) bits |= 1u << k;
    }
  }
  return bits;
}

constexpr int OT = 8;  // order tile side (pixels)

// Buffers of the same call that must start zeroed (16-byte aligned, sizes multiples of 16): cleared by the first
// edge-list kernel's threads instead of by hipMemsetAsync launches of their own (the fused step's row scales and
// fixed-point gradient sums: two launches, 12 us at C2).
struct ZeroRanges {
  void *ptr[3];
  size_t bytes[3];
};

// (also zeroes the order tiles' counters that edge_scatter fills, and the caller's ZeroRanges)
__global__ __launch_bounds__(256) void edge_count(EdgeParams p, int *blockcnt, int *tcnt, int nt, ZeroRanges z,
                                                  uint8_t *bits_out) {
  __shared__ int wsum[4];
  if (tcnt)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nt; i += gridDim.x * 256) tcnt[i] = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint4 *q = (uint4 *)z.ptr[k];
    const size_t n = z.bytes[k] / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = make_uint4(0, 0, 0, 0);
  }
  int b, pix0;
  const unsigned bits = chunk_bits(p, b, pix0);
  // the predicate of the lane's four pixels, kept for edge_scatter (with the on-device Laplacian mask it is 15 gathers
  // and three fixed-point gray conversions per pixel: evaluated once instead of twice)
  bits_out[(size_t)blockIdx.x * 256 + threadIdx.x] = (uint8_t)bits;
  int c = __popc(bits);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of blockcnt[0..nblk) -> blockoff, plus counts (B+2):
// counts[0] = N, counts[1+b] = first row of image b, counts[1+B] = N.
__global__ __launch_bounds__(1024) void edge_scan(const int *blockcnt, int *blockoff, int nblk, int nblk_img, int B,
                                                  int *counts, int *plan_header) {
  __shared__ int buf[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (plan_header && tid < 4) plan_header[tid] = 0;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + tid;
    const int v = i < nblk ? blockcnt[i] : 0;
    buf[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = tid >= o ? buf[tid - o] : 0;
      __syncthreads();
      buf[tid] += t;
      __syncthreads();
    }
    const int excl = carry + buf[tid] - v;
    if (i < nblk) {
      blockoff[i] = excl;
      if (i % nblk_img == 0) counts[1 + i / nblk_img] = excl;
    }
    __syncthreads();
    if (tid == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (tid == 0) {
    counts[0] = carry;
    counts[1 + B] = carry;
  }
}

__global__ __launch_bounds__(256) void edge_scatter(EdgeParams p, const int *blockoff, int *edges, int capacity,
                                                    int *rank, int *tcnt, const uint8_t *bits_in) {
  __shared__ int wtot[4];
  const int b = blockIdx.x / p.nblk_img;
  const int pix0 = (blockIdx.x - b * p.nblk_img) * CHUNK + 4 * threadIdx.x;
  const unsigned bits = bits_in[(size_t)blockIdx.x * 256 + threadIdx.x];   // (edge_count's)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = __popc(bits);
  int incl = c;  // inclusive scan over the 256 lanes: shuffles inside the waves + the 4 wave totals
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  for (int k = 0; k < wv; ++k) incl += wtot[k];
  int pos = blockoff[blockIdx.x] + incl - c;
  const int HW = p.H * p.W;
  const int tx_n = (p.W + OT - 1) / OT, ty_n = (p.H + OT - 1) / OT;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pix = pix0 + k;
    const bool on = bits & (1u << k);
    if (on && pos < capacity) {
      const int y = pix / p.W, x = pix - y * p.W;
      edges[3 * (size_t)pos + 0] = b;
      edges[3 * (size_t)pos + 1] = y;
      edges[3 * (size_t)pos + 2] = x;
      if (tcnt) atomicAdd(&tcnt[(b * ty_n + y / OT) * tx_n + x / OT], 1);  // rows per 8x8 order tile
    }
    // rank map: row index of every pixel (-1: not an edge pixel / beyond capacity)
    if (rank && pix < HW) rank[(size_t)b * HW + pix] = (on && pos < capacity) ? pos : -1;
    if (on) ++pos;
  }
}

__global__ __launch_bounds__(256) void edge_mask_write(EdgeParams p, uint8_t *out) {
  int b, pix0;
  const unsigned bits = chunk_bits(p, b, pix0);
  const int HW = p.H * p.W;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (pix0 + k < HW) out[(size_t)b * HW + pix0 + k] = (bits >> k) & 1u;
}

// ---- tile-major job order for the backward kernel ----
// order[k] = row of `edges` that job k works on: rows grouped by 8x8 image tile (tiles in
// image-major, row-major order; row-major inside a tile), so consecutive jobs are spatial
// neighbours.  Built from the rank map: one wave per tile.

__device__ __forceinline__ int tile_rank(const int *rank, int B, int H, int W, int tile, int lane, int &n_valid) {
  const int tx_n = (W + OT - 1) / OT, ty_n = (H + OT - 1) / OT;
  const int b = tile / (tx_n * ty_n), t = tile - b * tx_n * ty_n;
  const int y = (t / tx_n) * OT + lane / OT, x = (t % tx_n) * OT + lane % OT;
  (void)B;
  n_valid = 0;
  return (y < H && x < W) ? rank[((size_t)b * H + y) * W + x] : -1;
}

// super-tile (sty x 32 centres: the dense kernels' tile, sty = 8 for k_w = 9, 4 for k_w = 13) of pixel (b,y,x)
__device__ __forceinline__ int super_tile_of(int b, int y, int x, int H, int W, int sty) {
  const int sx_n = (W + 31) / 32, sy_n = (H + sty - 1) / sty;
  return (b * sy_n + y / sty) * sx_n + x / 32;
}


// one wave per super-tile: count its edge pixels (rank map), flag it dense at >= thr and append it to the dense
// list (plan[1] / plan[3] = counts of heavy / light tiles, zeroed by edge_scan)
// `strips` (k_s 49, 4-row tiles): the lists are built later (strip_select, plan_append) -- only the count is left in
// dflag (0 = not dense) and the strip list's counter is cleared
__global__ __launch_bounds__(256) void plan_count(const int *rank, int B, int H, int W, int sty, int thr, int *dflag,
                                                  int *plan, int *dense_ids, int *tcnt, int *strips) {
  const int sx_n = (W + 31) / 32, sy_n = (H + sty - 1) / sty;
  const int st = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (st >= B * sy_n * sx_n) return;
  const int b = st / (sy_n * sx_n), t = st - b * sy_n * sx_n;
  const int y0 = (t / sx_n) * sty, x0 = (t % sx_n) * 32;
  int n = 0;
  if (sty == OT) {
    // 8-row super-tiles are four 8 x 8 order tiles side by side: chunk k of the wave IS order tile k, so the same
    // four loads (all in flight together) also give tile_count_sparse's counts -- one launch less
    bool on[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = y0 + lane / OT, x = x0 + OT * k + lane % OT;
      on[k] = y < H && x < W && rank[((size_t)b * H + y) * W + x] >= 0;
    }
    int cnt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cnt[k] = __popcll(__ballot(on[k]));
      n += cnt[k];
    }
    const int dense = thr > 0 && n >= thr;
    if (tcnt && lane < 4) {
      const int tx_n = (W + OT - 1) / OT, ty_n = (H + OT - 1) / OT, tx = x0 / OT + lane;
      if (tx < tx_n) tcnt[(b * ty_n + y0 / OT) * tx_n + tx] = dense ? 0 : cnt[lane];   // (cnt[lane]: 4 selects)
    }
  } else {
    for (int i = lane; i < sty * 32; i += 64) {
      const int y = y0 + i / 32, x = x0 + i % 32;
      const bool on = y < H && x < W && rank[((size_t)b * H + y) * W + x] >= 0;
      n += __popcll(__ballot(on));
    }
  }
  if (lane == 0) {
    const int dense = thr > 0 && n >= thr;
    if (st == 0) plan[2] = sty;
    if (strips) {
      dflag[st] = dense ? n : 0;
      if (st == 0) strips[0] = 0;
      return;
    }
    dflag[st] = dense;
    if (dense) {   // heavy tiles from the front, light ones from the back (dense_tile_at, ssg_common.hpp)
      if (n > 64) dense_ids[atomicAdd(&plan[1], 1)] = st | (sty == OT && n > 128 ? TILE_HUGE : 0);
      else dense_ids[B * sy_n * sx_n - 1 - atomicAdd(&plan[3], 1)] = st;
    }
  }
}

// k_s 49: a strip of STRIP_ROWS x 32 centres (nine 4-row tiles, fewer at the bottom of the image) is listed for
// ssg_fwd_strip when every tile it covers is heavy (> 64 edge pixels).  Its tiles take CONSECUTIVE places at the front
// of the dense list (one atomic for the whole strip: place = slot of the tile-major region) with TILE_IN_STRIP set, and
// their counts are negated so that plan_append leaves them alone.  One thread per strip; strips[0] = number of
// strips, then (strip id, first place) pairs.
__global__ __launch_bounds__(256) void strip_select(int B, int H, int W, int *dflag, int *plan, int *dense_ids, int *strips) {
  constexpr int TPS = STRIP_ROWS / 4;
  const int sx_n = (W + 31) / 32, sy_n = (H + 3) / 4, ss_n = (H + STRIP_ROWS - 1) / STRIP_ROWS;
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= B * ss_n * sx_n) return;
  const int b = s / (ss_n * sx_n), t = s - b * ss_n * sx_n, ssy = t / sx_n, sx = t - ssy * sx_n;
  const int r0 = ssy * TPS, r1 = r0 + TPS < sy_n ? r0 + TPS : sy_n;
  int *f = dflag + ((size_t)b * sy_n) * sx_n + sx;
  bool all = true;
  for (int rr = r0; rr < r1; ++rr) all = all && f[rr * sx_n] > 64;
  if (!all) return;
  const int first = atomicAdd(&plan[1], r1 - r0);
  for (int rr = r0; rr < r1; ++rr) {
    f[rr * sx_n] = -f[rr * sx_n];
    dense_ids[first + rr - r0] = ((b * sy_n + rr) * sx_n + sx) | TILE_IN_STRIP;
  }
  const int k = atomicAdd(&strips[0], 1);
  strips[1 + 2 * k] = s;
  strips[2 + 2 * k] = first;
}

// the rest of the dense tile list from the counts plan_count left (negative: listed by strip_select)
__global__ __launch_bounds__(256) void plan_append(int ns, const int *dflag, int *plan, int *dense_ids) {
  const int st = blockIdx.x * 256 + threadIdx.x;
  if (st >= ns) return;
  const int n = dflag[st];
  if (n <= 0) return;
  if (n > 64) dense_ids[atomicAdd(&plan[1], 1)] = st;
  else dense_ids[ns - 1 - atomicAdd(&plan[3], 1)] = st;
}

// plan_count for 8-row super-tiles without touching the rank map: a super-tile's edge-pixel count is the sum of the four
// 8 x 8 order tiles it consists of, which edge_scatter has already counted (rows below `capacity` only, like the rank
// map).  One THREAD per super-tile instead of one wave reading 256 rank entries: 15 -> 3 us at C2.
__global__ __launch_bounds__(256) void plan_from_tile_counts(int B, int H, int W, int thr, int *dflag, int *plan,
                                                             int *dense_ids, int *tcnt) {
  const int sx_n = (W + 31) / 32, sy_n = (H + OT - 1) / OT, tx_n = (W + OT - 1) / OT;
  const int st = blockIdx.x * 256 + threadIdx.x;
  if (st >= B * sy_n * sx_n) return;
  const int b = st / (sy_n * sx_n), t = st - b * sy_n * sx_n, sy = t / sx_n, sx = t - sy * sx_n;
  int *c = tcnt + ((size_t)b * sy_n + sy) * tx_n + 4 * sx;   // (the order tiles have the super-tiles' row pitch: ty_n = sy_n)
  const int nk = tx_n - 4 * sx < 4 ? tx_n - 4 * sx : 4;
  int n = 0;
  for (int k = 0; k < nk; ++k) n += c[k];
  const int dense = thr > 0 && n >= thr;
  dflag[st] = dense;
  if (dense) {   // heavy tiles from the front, light ones from the back (dense_tile_at, ssg_common.hpp)
    if (n > 64) dense_ids[atomicAdd(&plan[1], 1)] = st | (n > 128 ? TILE_HUGE : 0);
    else dense_ids[B * sy_n * sx_n - 1 - atomicAdd(&plan[3], 1)] = st;
    for (int k = 0; k < nk; ++k) c[k] = 0;   // its rows belong to the dense kernels: not in the sparse order
  }
  if (st == 0) plan[2] = OT;
}

// rows per 8x8 order tile that are NOT in a dense super-tile (one wave per order tile)
__global__ __launch_bounds__(256) void tile_count_sparse(const int *rank, int B, int H, int W, int ntiles, int sty,
                                                         const int *dflag, int *tcnt) {
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tile >= ntiles) return;
  const int tx_n = (W + OT - 1) / OT, ty_n = (H + OT - 1) / OT;
  const int b = tile / (tx_n * ty_n), t = tile - b * tx_n * ty_n;
  const int y = (t / tx_n) * OT + lane / OT, x = (t % tx_n) * OT + lane % OT;
  const bool on = y < H && x < W && rank[((size_t)b * H + y) * W + x] >= 0 && !dflag[super_tile_of(b, y, x, H, W, sty)];
  const unsigned long long bal = __ballot(on);
  if (lane == 0) tcnt[tile] = __popcll(bal);
}

// exclusive scan of cnt[0..n) by one workgroup, 16 Ki entries per round: every lane keeps a run of 16
// consecutive counts in registers (four 16-byte loads), the run sums are scanned with shuffles inside the waves
// and across the 16 wave totals (two barriers per round), and the offsets leave as four 16-byte stores.
constexpr int SCAN_RUN = 16, SCAN_ROUND = 1024 * SCAN_RUN;

// (block of 1024 lanes; wtot / wincl: 16 ints of LDS each; returns the total, identical in every lane)
__device__ __forceinline__ int block_scan_1024(const int *cnt, int *off, int n, int *wtot, int *wincl) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int carry = 0;  // kept identical in every lane
  for (int base = 0; base < n; base += SCAN_ROUND) {
    const int lo = base + tid * SCAN_RUN;
    int v[SCAN_RUN];
    if (lo + SCAN_RUN <= n && ((size_t)(cnt + lo) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < SCAN_RUN; k += 4) {
        const int4 q = *(const int4 *)(cnt + lo + k);
        v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < SCAN_RUN; ++k) v[k] = lo + k < n ? cnt[lo + k] : 0;
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_RUN; ++k) s += v[k];
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    if (tid < 64) {
      int w = tid < 16 ? wtot[tid] : 0;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const int t = __shfl_up(w, o, 64);
        if (tid >= o) w += t;
      }
      if (tid < 16) wincl[tid] = w;
    }
    __syncthreads();
    int acc = carry + (wv ? wincl[wv - 1] : 0) + incl - s;
    carry += wincl[15];
#pragma unroll
    for (int k = 0; k < SCAN_RUN; ++k) {
      const int t = v[k];
      v[k] = acc;
      acc += t;
    }
    if (lo + SCAN_RUN <= n && ((size_t)(off + lo) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < SCAN_RUN; k += 4) *(int4 *)(off + lo + k) = make_int4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    } else {
#pragma unroll
      for (int k = 0; k < SCAN_RUN; ++k)
        if (lo + k < n) off[lo + k] = v[k];
    }
    __syncthreads();  // wtot / wincl are rewritten by the next round
  }
  return carry;
}

// The pieces of one round of the scan above for callers that keep a lane's run in registers between load and store
// (band_scan's single-round path: both of its scans' loads are issued together, and the plan is cut from the loaded tile
// counts before they are scanned).
__device__ __forceinline__ void scan_run_load(const int *cnt, int n, int (&v)[SCAN_RUN]) {
  const int lo = threadIdx.x * SCAN_RUN;
  if (lo + SCAN_RUN <= n && ((size_t)(cnt + lo) & 15) == 0) {
#pragma unroll
    for (int k = 0; k < SCAN_RUN; k += 4) {
      const int4 q = *(const int4 *)(cnt + lo + k);
      v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_RUN; ++k) v[k] = lo + k < n ? cnt[lo + k] : 0;
  }
}
__device__ __forceinline__ void scan_run_store(int *off, int n, const int (&v)[SCAN_RUN]) {
  const int lo = threadIdx.x * SCAN_RUN;
  if (lo + SCAN_RUN <= n && ((size_t)(off + lo) & 15) == 0) {
#pragma unroll
    for (int k = 0; k < SCAN_RUN; k += 4) *(int4 *)(off + lo + k) = make_int4(v[k], v[k + 1], v[k + 2], v[k + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_RUN; ++k)
      if (lo + k < n) off[lo + k] = v[k];
  }
}
// counts -> exclusive offsets in place; returns the total (two barriers; wtot / wincl free again after a third)
__device__ __forceinline__ int scan_run_excl(int (&v)[SCAN_RUN], int *wtot, int *wincl) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_RUN; ++k) s += v[k];
  int incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  if (tid < 64) {
    int w = tid < 16 ? wtot[tid] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(w, o, 64);
      if (tid >= o) w += t;
    }
    if (tid < 16) wincl[tid] = w;
  }
  __syncthreads();
  int acc = (wv ? wincl[wv - 1] : 0) + incl - s;
#pragma unroll
  for (int k = 0; k < SCAN_RUN; ++k) {
    const int t = v[k];
    v[k] = acc;
    acc += t;
  }
  return wincl[15];
}

__global__ __launch_bounds__(1024) void tile_scan(const int *cnt, int *off, int n, int *total_out) {
  __shared__ int wtot[16], wincl[16];
  const int total = block_scan_1024(cnt, off, n, wtot, wincl);
  if (total_out && threadIdx.x == 0) *total_out = total;
}

__global__ __launch_bounds__(256) void tile_scatter(const int *rank, int B, int H, int W, int ntiles, const int *off,
                                                    int *order, int capacity, const int *skip, int sty) {
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tile >= ntiles) return;
  int nv;
  int r = tile_rank(rank, B, H, W, tile, lane, nv);
  if (skip && r >= 0) {  // rows of dense super-tiles belong to the dense kernels
    const int tx_n = (W + OT - 1) / OT, ty_n = (H + OT - 1) / OT;
    const int b = tile / (tx_n * ty_n), t = tile - b * tx_n * ty_n;
    if (skip[super_tile_of(b, (t / tx_n) * OT + lane / OT, (t % tx_n) * OT + lane % OT, H, W, sty)]) r = -1;
  }
  const unsigned long long bal = __ballot(r >= 0);
  if (r >= 0) {
    const int pos = off[tile] + __popcll(bal & ((1ull << lane) - 1ull));
    if (pos < capacity) order[pos] = r;
  }
}

// Marks the groups of ORDER_GROUP consecutive jobs whose edge pixels are one image's and lie within
// 8 rows x 16 columns (bit ORDER_FLAG of the group's first entry): the forward kernel variants pick
// their groups from this flag with a single load.
// (up to two orders per launch: blockIdx.y picks the (order, row count) pair)
__global__ __launch_bounds__(256) void tile_group_flags(int *order_a, const int *n_a, int *order_b, const int *n_b,
                                                        const int *edges, int capacity) {
  int *order = blockIdx.y ? order_b : order_a;
  const int *n_ptr = blockIdx.y ? n_b : n_a;
  const int g = blockIdx.x * 256 + threadIdx.x;
  int n = n_ptr[0];
  n = n < capacity ? n : capacity;
  const int k0 = g * ORDER_GROUP;
  if (k0 >= n) return;
  int y0 = 1 << 30, x0 = 1 << 30, y1 = -1, x1 = -1, b0 = -1;
  bool ok = true;
  int first = 0;
  for (int j = 0; j < ORDER_GROUP && k0 + j < n; ++j) {
    const int row = order[k0 + j] & ORDER_MASK;
    if (j == 0) first = row;
    const int b = edges[3 * (size_t)row], y = edges[3 * (size_t)row + 1], x = edges[3 * (size_t)row + 2];
    if (b0 < 0) b0 = b;
    ok = ok && b == b0;
    y0 = y < y0 ? y : y0;
    y1 = y > y1 ? y : y1;
    x0 = x < x0 ? x : x0;
    x1 = x > x1 ? x : x1;
  }
  ok = ok && (y1 - y0) <= MERGE_ROWS - 1 && (x1 - x0) <= MERGE_COLS - 1;
  order[k0] = first | (ok ? ORDER_FLAG : 0);
}

// ---- banded builder (round 4): edge list, rank map, dense list and sparse order in THREE launches instead of seven ----
// A workgroup owns a band of 8 image rows x 256 columns: 32 strips of 8 pixels per row, one lane per strip.  Such a block
// is a run of whole 8 x 8 order tiles (a strip IS a tile row) and of whole 8 x 32 super-tiles, and each of its row
// segments is a contiguous piece of the reference's row-major edge order -- so one pass over the mask yields the
// row-segment counts (the edge order's scan input) AND the tile counts (the plan's input), which the seven-launch
// builder only had after its scatter pass (atomics into the order tiles' counters):
//   band_count    bits of every strip, edge pixels per row segment, edge pixels per order tile (+ the caller's ZeroRanges)
//   band_scan     one workgroup: row-segment offsets and `counts`; tile counts corrected when N exceeds the capacity
//                 (rows from `capacity` on do not exist for the plan: same lists as the seven-launch builder); dense
//                 list and flags from the tile counts; offsets of the sparse order; the plan's header
//   band_scatter  edges, rank map and the tile-major order of the rows left to the direct kernels
//                 (+ the merge flags of the groups of ORDER_GROUP entries that lie inside the block)
constexpr int BAND_COLS = 256;

__device__ __forceinline__ unsigned strip_bits(const EdgeParams &p, int b, int y, int x0) {
  unsigned bits = 0;
  if (p.kind == 0 && x0 + 8 <= p.W) {
    const float *m = (const float *)p.mask + (size_t)b * p.mask_channels * p.H * p.W + (size_t)y * p.W + x0;
    float v[8];
    if (((size_t)m & 15) == 0) {
      const float4 a = *(const float4 *)m, c = *(const float4 *)(m + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = m[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (v[k] == 1.0f && (p.stride <= 1 || (y % p.stride) == ((x0 + k) % p.stride))) bits |= 1u << k;
    return bits;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (x0 + k < p.W && edge_pred(p, b, y, x0 + k)) bits |= 1u << k;
  return bits;
}

__global__ __launch_bounds__(256) void band_count(EdgeParams p, int nseg, int *segcnt, int *tcnt, uint8_t *bits_out,
                                                  ZeroRanges z) {
  __shared__ int s_c[8][33];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint4 *q = (uint4 *)z.ptr[k];
    const size_t n = z.bytes[k] / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = make_uint4(0, 0, 0, 0);
  }
  const int tx_n = (p.W + OT - 1) / OT, ty_n = (p.H + OT - 1) / OT;
  const int seg = blockIdx.x % nseg, band = (blockIdx.x / nseg) % ty_n, b = blockIdx.x / (nseg * ty_n);
  const int r = threadIdx.x >> 5, g = threadIdx.x & 31;
  const int y = band * OT + r, sx = seg * 32 + g;
  const bool in = y < p.H && sx < tx_n;
  unsigned bits = 0u;
  if (p.kind == 2) {
    // Laplacian of GT on the fly (generate_mask.py:22-31): the block's 'L' values once -- its 8 rows x 256 columns and a
    // one-pixel frame, BORDER_REFLECT_101 -- into LDS (10 per lane), then the 5-point stencil from there: a quarter of
    // the gathers and gray conversions of evaluating edge_pred per pixel (C2 step with mask=None: +0.045 -> +0.02 ms)
    __shared__ uint8_t s_l[OT + 2][BAND_COLS + 2];
    const float *img = (const float *)p.mask + (size_t)b * 3 * p.H * p.W;
    const size_t plane = (size_t)p.H * p.W;
    constexpr int NL = (OT + 2) * (BAND_COLS + 2), NIT = (NL + 255) / 256;
    float px[NIT][3];
    // (all loads of the lane first -- unconditional, from clamped coordinates -- then the conversions: one round of
    //  memory latency per workgroup instead of one per value)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + 256 * it < NL ? threadIdx.x + 256 * it : NL - 1;
      const int rr = i / (BAND_COLS + 2), cc = i - rr * (BAND_COLS + 2);
      int yy = band * OT + rr - 1, xx = seg * BAND_COLS + cc - 1;
      yy = yy > p.H ? p.H : yy;
      xx = xx > p.W ? p.W : xx;
      const size_t off = (size_t)reflect101_idx(yy, p.H) * p.W + reflect101_idx(xx, p.W);
#pragma unroll
      for (int c = 0; c < 3; ++c) px[it][c] = img[off + c * plane];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + 256 * it;
      if (i < NL) ((uint8_t *)s_l)[i] = (uint8_t)gray_l(px[it], 1, 0);
    }
    __syncthreads();
    if (in) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int x = sx * 8 + k, cx = g * 8 + k + 1;
        if (x >= p.W || (p.stride > 1 && (y % p.stride) != (x % p.stride))) continue;
        int v = (int)s_l[r][cx] + (int)s_l[r + 2][cx] + (int)s_l[r + 1][cx - 1] + (int)s_l[r + 1][cx + 1] - 4 * (int)s_l[r + 1][cx];
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        if ((float)v > p.thr) bits |= 1u << k;
      }
    }
  } else if (in) {
    bits = strip_bits(p, b, y, sx * 8);
  }
  if (in) bits_out[((size_t)b * p.H + y) * tx_n + sx] = (uint8_t)bits;
  const int c = __popc(bits);
  int rc = c;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) rc += __shfl_xor(rc, o, 64);   // (offsets < 32: the sum of the row's 32 lanes)
  if (g == 0 && y < p.H) segcnt[((size_t)b * p.H + y) * nseg + seg] = rc;
  s_c[r][g] = c;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int tx = seg * 32 + threadIdx.x;
    if (tx < tx_n) {
      int n = 0;
#pragma unroll
      for (int k = 0; k < OT; ++k) n += s_c[k][threadIdx.x];
      tcnt[((size_t)b * ty_n + band) * tx_n + tx] = n;
    }
  }
}

__global__ __launch_bounds__(1024) void band_scan(int B, int H, int W, int nseg, const int *segcnt, int *segoff, int *counts,
                                                  int capacity, const uint8_t *bits, int *tcnt, int *toff, int thr,
                                                  int *dflag, int *plan, int *hint) {
  __shared__ int wtot[16], wincl[16];
  __shared__ int s_heavy, s_light;
  const int tid = threadIdx.x;
  const int tx_n = (W + OT - 1) / OT, ty_n = (H + OT - 1) / OT;
  const int nrow = B * H * nseg, nt = B * ty_n * tx_n;
  if (tid == 0) s_heavy = s_light = 0;
  // Single-round path (both arrays fit one round of the scan, whole super-tiles per lane: C2, C4 and every per-image
  // call): the loads of BOTH scans are issued together, the plan is cut from the tile counts in registers -- a lane's 16
  // consecutive order tiles are four consecutive super-tiles when a tile row holds a multiple of four -- and the counts
  // are scanned from there: two memory round trips instead of five (15 -> 8 us at C2).
  if (plan && (tx_n & 3) == 0 && nrow <= SCAN_ROUND && nt <= SCAN_ROUND) {
    int sv[SCAN_RUN], tv[SCAN_RUN];
    scan_run_load(segcnt, nrow, sv);
    scan_run_load(tcnt, nt, tv);
    const int N = scan_run_excl(sv, wtot, wincl);
    if (N <= capacity) {   // (block-uniform; an overflowing call takes the general path below)
      scan_run_store(segoff, nrow, sv);
      const int ns = nt >> 2;
#pragma unroll
      for (int j = 0; j < SCAN_RUN / 4; ++j) {
        const int st = tid * (SCAN_RUN / 4) + j;
        if (st < ns) {
          const int n = tv[4 * j] + tv[4 * j + 1] + tv[4 * j + 2] + tv[4 * j + 3];
          const int dense = thr > 0 && n >= thr;
          dflag[st] = dense;
          if (dense) {   // heavy tiles from the front, light ones from the back (dense_tile_at, ssg_common.hpp)
            if (n > 64) plan[4 + atomicAdd(&s_heavy, 1)] = st | (n > 128 ? TILE_HUGE : 0);
            else plan[4 + ns - 1 - atomicAdd(&s_light, 1)] = st;
            tv[4 * j] = tv[4 * j + 1] = tv[4 * j + 2] = tv[4 * j + 3] = 0;   // its rows belong to the dense kernels
          }
        }
      }
      __syncthreads();   // (wtot / wincl free again; segoff visible)
      for (int b = tid; b < B; b += 1024) counts[1 + b] = segoff[(size_t)b * H * nseg];
      if (tid == 0) counts[0] = counts[1 + B] = N;
      const int n_sparse = scan_run_excl(tv, wtot, wincl);
      scan_run_store(toff, nt, tv);
      if (tid == 0) {
        plan[0] = n_sparse;
        plan[1] = s_heavy;
        plan[2] = OT;
        plan[3] = s_light;
        if (hint) {   // (host-mapped: what the next forked pass on this device assigns its streams by, ssg_api.hip)
          hint[0] = n_sparse;
          hint[1] = s_heavy + s_light;
        }
      }
      return;
    }
    __syncthreads();
  }
  const int N = block_scan_1024(segcnt, segoff, nrow, wtot, wincl);   // (ends with a barrier: segoff is visible)
  for (int b = tid; b < B; b += 1024) counts[1 + b] = segoff[(size_t)b * H * nseg];
  if (tid == 0) counts[0] = counts[1 + B] = N;
  if (N > capacity) {
    // rows from `capacity` on are not listed: the tiles count what is left of them.  A row segment lies before the cut,
    // behind it, or -- exactly one in the batch -- across it (its first capacity - offset edge pixels count)
    for (int t = tid; t < nt; t += 1024) {
      const int b = t / (ty_n * tx_n), rem = t - b * ty_n * tx_n, band = rem / tx_n, tx = rem - band * tx_n;
      const int seg = tx / 32;
      int n = 0;
      for (int r = 0; r < OT; ++r) {
        const int y = band * OT + r;
        if (y >= H) break;
        const size_t rs = ((size_t)b * H + y) * nseg + seg;
        const int so = segoff[rs];
        if (so >= capacity) continue;
        const uint8_t *row = bits + ((size_t)b * H + y) * tx_n;
        int c = __popc((unsigned)row[tx]);
        if (so + segcnt[rs] > capacity) {
          int before = so;
          for (int k = seg * 32; k < tx; ++k) before += __popc((unsigned)row[k]);
          const int left = capacity - before;
          c = left < 0 ? 0 : (left < c ? left : c);
        }
        n += c;
      }
      tcnt[t] = n;
    }
  }
  __syncthreads();
  if (plan) {
    const int sx_n = (W + 31) / 32, ns = B * ty_n * sx_n;
    for (int st = tid; st < ns; st += 1024) {
      const int b = st / (ty_n * sx_n), rem = st - b * ty_n * sx_n, sy = rem / sx_n, sx = rem - sy * sx_n;
      int *c = tcnt + ((size_t)b * ty_n + sy) * tx_n + 4 * sx;
      const int nk = tx_n - 4 * sx < 4 ? tx_n - 4 * sx : 4;
      int n = 0;
      for (int k = 0; k < nk; ++k) n += c[k];
      const int dense = thr > 0 && n >= thr;
      dflag[st] = dense;
      if (dense) {   // heavy tiles from the front, light ones from the back (dense_tile_at, ssg_common.hpp)
        if (n > 64) plan[4 + atomicAdd(&s_heavy, 1)] = st | (n > 128 ? TILE_HUGE : 0);
        else plan[4 + ns - 1 - atomicAdd(&s_light, 1)] = st;
        for (int k = 0; k < nk; ++k) c[k] = 0;   // its rows belong to the dense kernels: not in the sparse order
      }
    }
    __syncthreads();
  }
  const int n_sparse = block_scan_1024(tcnt, toff, nt, wtot, wincl);
  if (plan && tid == 0) {
    plan[0] = n_sparse;
    plan[1] = s_heavy;
    plan[2] = OT;
    plan[3] = s_light;
    if (hint) {
      hint[0] = n_sparse;
      hint[1] = s_heavy + s_light;
    }
  }
}

__global__ __launch_bounds__(256) void band_scatter(EdgeParams p, int nseg, const int *segoff, const uint8_t *bits_in,
                                                    int *edges, int capacity, int *rank, const int *toff,
                                                    const int *dflag, int *order_out, const int *n_order) {
  __shared__ int s_v[8][33];
  __shared__ unsigned short s_x[OT * BAND_COLS];   // column of every listed row of the block, by its place in the order
  const int tx_n = (p.W + OT - 1) / OT, ty_n = (p.H + OT - 1) / OT;
  const int seg = blockIdx.x % nseg, band = (blockIdx.x / nseg) % ty_n, b = blockIdx.x / (nseg * ty_n);
  const int r = threadIdx.x >> 5, g = threadIdx.x & 31;
  const int y = band * OT + r, sx = seg * 32 + g, x0 = sx * 8;
  const bool in = y < p.H && sx < tx_n;
  const unsigned bits = in ? bits_in[((size_t)b * p.H + y) * tx_n + sx] : 0u;
  // (issued with the loads above: the order's offsets)
  const size_t tile0 = ((size_t)b * ty_n + band) * tx_n;
  const int sx_n = (p.W + 31) / 32;
  const bool listed = in && order_out && !(dflag && dflag[((size_t)b * ty_n + band) * sx_n + sx / 4]);   // not a dense super-tile's
  const int t_off = listed ? toff[tile0 + sx] : 0;
  const int sx_last = seg * 32 + 31 < tx_n ? seg * 32 + 31 : tx_n - 1;
  const int blk_first = order_out ? toff[tile0 + seg * 32] : 0;
  // end of the block's piece of the order = the offset of the NEXT block's first order tile (tiles are numbered in block
  // order), the whole order's length behind the last one.  (Not toff + tcnt of the block's last tile: band_scan's
  // single-round path zeroes the counts of dense tiles in registers only, so that sum ran past the block's piece when its
  // last tile was a dense one -- a group reaching into the next band then read s_x entries nobody had written and could
  // be flagged mergeable on LDS garbage: its rows outside the first row's merge window came out wrong.)
  const size_t t_next = tile0 + sx_last + 1;
  int n_listed = order_out ? n_order[0] : 0;
  n_listed = n_listed < capacity ? n_listed : capacity;
  const int blk_end = !order_out ? 0 : (t_next < (size_t)p.B * ty_n * tx_n ? toff[t_next] : n_listed);
  const int c = __popc(bits);
  int incl = c;   // inclusive scan over the row's 32 lanes (a half-wave: lanes g >= o take from their own half)
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (g >= o) incl += t;
  }
  const int pos0 = (y < p.H ? segoff[((size_t)b * p.H + y) * nseg + seg] : 0) + incl - c;
  int valid = capacity - pos0;
  valid = valid < 0 ? 0 : (valid < c ? valid : c);
  s_v[r][g] = valid;
  __syncthreads();
  int idx = t_off;
  for (int k = 0; k < r; ++k) idx += s_v[k][g];
  if (in) {
    int rk[8];
    int pos = pos0, li = idx - blk_first;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool on = bits & (1u << k);
      const bool ok = on && pos < capacity;
      if (ok) {
        edges[3 * (size_t)pos + 0] = b;
        edges[3 * (size_t)pos + 1] = y;
        edges[3 * (size_t)pos + 2] = x0 + k;
        if (listed) s_x[li++] = (unsigned short)(x0 + k);   // (the block's listed rows: fewer than its 2,048 pixels)
      }
      rk[k] = ok ? pos : -1;   // rank map: row index of every pixel (-1: not an edge pixel / beyond capacity)
      if (on) ++pos;
    }
    int *rrow = rank + ((size_t)b * p.H + y) * p.W + x0;
    if (x0 + 8 <= p.W && ((size_t)rrow & 15) == 0) {
      *(int4 *)rrow = make_int4(rk[0], rk[1], rk[2], rk[3]);
      *(int4 *)(rrow + 4) = make_int4(rk[4], rk[5], rk[6], rk[7]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (x0 + k < p.W) rrow[k] = rk[k];
    }
  }
  if (!order_out) return;
  __syncthreads();
  // The order's entries with the merge flag of tile_group_flags on the first entry of every group of ORDER_GROUP that lies
  // inside this block: one image and 8 rows by construction, so the flag is "within MERGE_COLS columns".  A group that
  // runs across two blocks stays unflagged (the single variants take it: same values) -- across band rows it could
  // only qualify in an image of <= MERGE_COLS columns.
  if (listed && valid > 0) {
    const int n = n_listed;
    for (int j = 0; j < valid; ++j) {
      const int k0 = idx + j;
      if (k0 >= capacity) break;
      int e = pos0 + j;   // (a strip's rows are consecutive in the edge order)
      if (k0 % ORDER_GROUP == 0) {
        const int gend = k0 + ORDER_GROUP < n ? k0 + ORDER_GROUP : n;
        if (gend <= blk_end) {
          int lo = 1 << 30, hi = -1;
          for (int q = k0 - blk_first; q < gend - blk_first; ++q) {
            const int x = s_x[q];
            lo = x < lo ? x : lo;
            hi = x > hi ? x : hi;
          }
          if (hi - lo <= MERGE_COLS - 1) e |= ORDER_FLAG;
        }
      }
      order_out[k0] = e;
    }
  }
}

// ---- the reference operator's position list -> the engine's plan (ssg_compute_similarity[_backward] with many positions,
// ssg_api.hip).  The caller's `pos` (mc (Y,X) pairs in any order, duplicates allowed) becomes a uint8 mask; the regular
// builder makes edge list, rank map and plan from it in ITS row order (row-major); then everything that names a row is
// relabelled to the CALLER's row numbers, so that the kernels read and write the caller's (mc, k_s^2) rows in place:
//   pos_perm    perm[internal row] = the largest caller row at that pixel
//   pos_dups    caller rows that lost that election (duplicates of a position): listed for a direct launch of their own
//   pos_relabel rank map and the plan's sparse order: internal row -> caller row
__global__ __launch_bounds__(256) void pos_to_mask(const int *pos, int mc, int Hp, int Wp, uint8_t *mask) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= mc) return;
  const int Y = pos[2 * n], X = pos[2 * n + 1];
  if ((unsigned)Y < (unsigned)Hp && (unsigned)X < (unsigned)Wp) mask[(size_t)Y * Wp + X] = 1;
}

__global__ __launch_bounds__(256) void pos_perm(const int *pos, int mc, int Hp, int Wp, const int *rank, int *perm) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= mc) return;
  const int Y = pos[2 * n], X = pos[2 * n + 1];
  if ((unsigned)Y >= (unsigned)Hp || (unsigned)X >= (unsigned)Wp) return;
  const int r = rank[(size_t)Y * Wp + X];
  if (r >= 0) atomicMax(&perm[r], n);
}

__global__ __launch_bounds__(256) void pos_dups(const int *pos, int mc, int Hp, int Wp, const int *rank, const int *perm,
                                                int *dup, int *ndup) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= mc) return;
  const int Y = pos[2 * n], X = pos[2 * n + 1];
  if ((unsigned)Y >= (unsigned)Hp || (unsigned)X >= (unsigned)Wp) return;   // (a position outside the image: no row work)
  const int r = rank[(size_t)Y * Wp + X];
  if (r < 0 || perm[r] != n) dup[atomicAdd(ndup, 1)] = n;
}

__global__ __launch_bounds__(256) void pos_relabel(int *rank, size_t npix, const int *perm, int *order2, const int *plan,
                                                   int capacity) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < npix) {
    const int r = rank[i];
    if (r >= 0) rank[i] = perm[r];
  }
  int ns = plan[0];
  ns = ns < capacity ? ns : capacity;
  if (i < (size_t)ns) {
    const int e = order2[i];
    order2[i] = (e & ORDER_FLAG) | perm[e & ORDER_MASK];
  }
}

int launch_pos_to_mask(const int *pos, int mc, int Hp, int Wp, uint8_t *mask, hipStream_t st) {
  hipLaunchKernelGGL(pos_to_mask, dim3((mc + 255) / 256), dim3(256), 0, st, pos, mc, Hp, Wp, mask);
  return (int)hipGetLastError();
}

int launch_pos_relabel(const int *pos, int mc, int Hp, int Wp, int *rank, int *perm, int *dup, int *ndup, int *plan,
                       int *order2, hipStream_t st) {
  const unsigned g = (unsigned)((mc + 255) / 256);
  hipLaunchKernelGGL(pos_perm, dim3(g), dim3(256), 0, st, pos, mc, Hp, Wp, rank, perm);
  hipLaunchKernelGGL(pos_dups, dim3(g), dim3(256), 0, st, pos, mc, Hp, Wp, rank, perm, dup, ndup);
  const size_t npix = (size_t)Hp * Wp, nthr = npix > (size_t)mc ? npix : (size_t)mc;
  hipLaunchKernelGGL(pos_relabel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, rank, npix, perm, order2, plan, mc);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------ host ----
static size_t n_order_tiles(int B, int H, int W) { return (size_t)B * ((H + OT - 1) / OT) * ((W + OT - 1) / OT); }
// (sized for the smallest super-tile height, 4 rows)
static size_t n_super_tiles(int B, int H, int W) { return (size_t)B * ((H + 3) / 4) * ((W + 31) / 32); }

static size_t n_row_segments(int B, int H, int W) { return (size_t)B * H * ((W + BAND_COLS - 1) / BAND_COLS); }

size_t edge_scratch_bytes(int B, int H, int W) {
  const size_t nblk = (size_t)B * (((size_t)H * W + CHUNK - 1) / CHUNK);
  // (+ one predicate byte per lane of edge_count, behind the int arrays)
  const size_t chunked = (2 * nblk + 2 * n_order_tiles(B, H, W) + n_super_tiles(B, H, W)) * sizeof(int) + 64 + nblk * 256;
  // banded builder: row-segment counts and offsets instead of the chunks', one predicate byte per 8-pixel strip
  const size_t banded = (2 * n_row_segments(B, H, W) + 2 * n_order_tiles(B, H, W) + n_super_tiles(B, H, W)) * sizeof(int) + 64 +
                        (size_t)B * H * ((W + OT - 1) / OT);
  return chunked > banded ? chunked : banded;
}

int *plan_hint_device_word(hipStream_t st);   // ssg_api.hip: host-mapped {rows for the direct kernels, dense tiles} of the device's last plan

static bool banded_enabled() {
  static const bool on = env_int("SSG_EDGE_BANDED", 1) != 0;   // (profiling build only: the chunked builder for every call)
  return on;
}

static size_t n_strips(int B, int H, int W) { return (size_t)B * ((H + STRIP_ROWS - 1) / STRIP_ROWS) * ((W + 31) / 32); }

// forward plan: [0] n_sparse, [1] n_heavy, [2] tile rows, [3] n_light, [4, 4+ns) the dense kernels' super-tile ids (heavy from the front, light from the back; ns =
// n_super_tiles), then the k_s 49 forward's strip list (count, then n_strips (strip id, first place) pairs), then
// (capacity) the tile-major order of the rows left to the direct kernels
int fwd_plan_strip_offset(int B, int H, int W) { return 4 + (int)n_super_tiles(B, H, W); }
int fwd_plan_order_offset(int B, int H, int W) { return fwd_plan_strip_offset(B, H, W) + 1 + 2 * (int)n_strips(B, H, W); }

size_t fwd_plan_bytes(int B, int H, int W, int capacity) {
  return sizeof(int) * ((size_t)fwd_plan_order_offset(B, H, W) + (size_t)(capacity > 0 ? capacity : 1));
}

static void build_order(const int *rank, int B, int H, int W, int *order, int capacity, const int *edges,
                        const int *n_ptr, bool flags, int *tcnt, int *toff, hipStream_t st) {
  const int nt = (int)n_order_tiles(B, H, W);
  hipLaunchKernelGGL(tile_scan, dim3(1), dim3(1024), 0, st, tcnt, toff, nt, nullptr);
  hipLaunchKernelGGL(tile_scatter, dim3((nt + 3) / 4), dim3(256), 0, st, rank, B, H, W, nt, toff, order, capacity, nullptr, 8);
  const int ngroups = (capacity + ORDER_GROUP - 1) / ORDER_GROUP;
  if (flags && ngroups > 0)
    hipLaunchKernelGGL(tile_group_flags, dim3((ngroups + 255) / 256, 1), dim3(256), 0, st, order, n_ptr, nullptr, nullptr,
                       edges, capacity);
}

int launch_edge_list(const void *mask, int kind, int mask_channels, int B, int H, int W, int stride, float thr,
                     int *edges, int capacity, int *counts, int *rank, int *order, int *plan, int dense_thr,
                     int plan_tile_rows, void *scratch, void *zero_a, size_t zero_a_bytes, void *zero_b,
                     size_t zero_b_bytes, void *zero_c, size_t zero_c_bytes, hipStream_t st) {
  EdgeParams p{mask, kind, mask_channels, B, H, W, stride, thr, (int)(((size_t)H * W + CHUNK - 1) / CHUNK)};
  const int nblk = B * p.nblk_img;
  const int nt = (int)n_order_tiles(B, H, W);
  if (rank && banded_enabled() && ((plan && plan_tile_rows == OT && !order) || (order && !plan))) {
    const int nseg = (W + BAND_COLS - 1) / BAND_COLS, ty_n = (H + OT - 1) / OT;
    const size_t nrow = n_row_segments(B, H, W);
    int *segcnt = (int *)scratch, *segoff = segcnt + nrow, *tcnt = segoff + nrow, *toff = tcnt + nt, *dflag = toff + nt;
    uint8_t *bits = (uint8_t *)(dflag + n_super_tiles(B, H, W)) + 64;
    const ZeroRanges z{{zero_a, zero_b, zero_c}, {zero_a ? zero_a_bytes : 0, zero_b ? zero_b_bytes : 0, zero_c ? zero_c_bytes : 0}};
    const unsigned grid = (unsigned)(B * ty_n * nseg);
    int *order_out = plan ? plan + fwd_plan_order_offset(B, H, W) : order;
    hipLaunchKernelGGL(band_count, dim3(grid), dim3(256), 0, st, p, nseg, segcnt, tcnt, bits, z);
    hipLaunchKernelGGL(band_scan, dim3(1), dim3(1024), 0, st, B, H, W, nseg, segcnt, segoff, counts, capacity, bits, tcnt, toff,
                       dense_thr, plan ? dflag : nullptr, plan, plan ? plan_hint_device_word(st) : nullptr);
    // (the merge flags of the groups are set by the scatter pass itself: three launches)
    hipLaunchKernelGGL(band_scatter, dim3(grid), dim3(256), 0, st, p, nseg, segoff, bits, edges, capacity, rank, toff,
                       plan ? dflag : nullptr, order_out, plan ? plan : counts);
    return (int)hipGetLastError();
  }
  int *blockcnt = (int *)scratch, *blockoff = blockcnt + nblk;
  int *tcnt = blockoff + nblk, *toff = tcnt + nt;
  const bool need_tiles = order || plan;  // rows per 8x8 order tile, counted while the edges are scattered
  uint8_t *bits = (uint8_t *)(toff + nt + n_super_tiles(B, H, W)) + 64;
  const ZeroRanges z{{zero_a, zero_b, zero_c}, {zero_a ? zero_a_bytes : 0, zero_b ? zero_b_bytes : 0, zero_c ? zero_c_bytes : 0}};
  hipLaunchKernelGGL(edge_count, dim3(nblk), dim3(256), 0, st, p, blockcnt, need_tiles ? tcnt : nullptr, nt, z, bits);
  hipLaunchKernelGGL(edge_scan, dim3(1), dim3(1024), 0, st, blockcnt, blockoff, nblk, p.nblk_img, B, counts, plan);
  hipLaunchKernelGGL(edge_scatter, dim3(nblk), dim3(256), 0, st, p, blockoff, edges, capacity, rank,
                     need_tiles ? tcnt : nullptr, bits);
  // (with a forward plan the group flags of both orders are set by one launch at the end)
  if (order) build_order(rank, B, H, W, order, capacity, edges, counts, !plan, tcnt, toff, st);
  if (plan) {
    const int sty = plan_tile_rows;
    const int ns = B * ((H + sty - 1) / sty) * ((W + 31) / 32);
    int *dflag = toff + nt, *order2 = plan + fwd_plan_order_offset(B, H, W);
    int *strips = plan + fwd_plan_strip_offset(B, H, W);
    if (sty == OT) {   // 8-row super-tiles = four order tiles each: counts are already there
      hipLaunchKernelGGL(plan_from_tile_counts, dim3((ns + 255) / 256), dim3(256), 0, st, B, H, W, dense_thr, dflag, plan,
                         plan + 4, tcnt);
    } else {
      const bool with_strips = sty == 4;
      hipLaunchKernelGGL(plan_count, dim3((ns + 3) / 4), dim3(256), 0, st, rank, B, H, W, sty, dense_thr, dflag, plan,
                         plan + 4, nullptr, with_strips ? strips : nullptr);
      if (with_strips) {
        const int nstr = (int)n_strips(B, H, W);
        hipLaunchKernelGGL(strip_select, dim3((nstr + 255) / 256), dim3(256), 0, st, B, H, W, dflag, plan, plan + 4, strips);
        hipLaunchKernelGGL(plan_append, dim3((ns + 255) / 256), dim3(256), 0, st, ns, dflag, plan, plan + 4);
      }
      hipLaunchKernelGGL(tile_count_sparse, dim3((nt + 3) / 4), dim3(256), 0, st, rank, B, H, W, nt, sty, dflag, tcnt);
    }
    hipLaunchKernelGGL(tile_scan, dim3(1), dim3(1024), 0, st, tcnt, toff, nt, plan);
    hipLaunchKernelGGL(tile_scatter, dim3((nt + 3) / 4), dim3(256), 0, st, rank, B, H, W, nt, toff, order2, capacity, dflag,
                       sty);
    const int ngroups = (capacity + ORDER_GROUP - 1) / ORDER_GROUP;
    if (ngroups > 0)
      hipLaunchKernelGGL(tile_group_flags, dim3((ngroups + 255) / 256, order ? 2 : 1), dim3(256), 0, st, order2, plan, order,
                         counts, edges, capacity);
  }
  return (int)hipGetLastError();
}

// ---- small calls: the whole edge list by ONE workgroup, one launch (round 6) ----
// BASELINE's C1 (1 x 3 x 64 x 64, 209 edge pixels) spends 3 x 5 us of its 59 us step in the three launches of the banded
// builder.  Up to TINY_LIST_PIXELS pixels one 1,024-lane workgroup does it all: clears the caller's ranges (the
// fixed-point gradient sums, the gradient, the step's ticket word), evaluates the predicate of TINY_PPT consecutive
// pixels per lane (image-major, row-major: a lane's pixels are consecutive in the reference's torch.nonzero order,
// similaritywrapper.py:64-67), scans the counts and writes rows, rank map and counts exactly as the other builders do
// (counts[0] = counts[1+B] = N found, counts[1+b] = first row of image b; rows at and beyond `capacity` are dropped, their
// rank is -1).  No tile order, no plan: its caller (ssg_tiny.hip) works row by row.
constexpr int TINY_LIST_PIXELS = 16384, TINY_PPT_MAX = TINY_LIST_PIXELS / 1024;
static_assert(TINY_PPT_MAX <= 32, "a lane keeps its pixels' predicate bits in one word");
struct TinyZero {
  void *ptr[3];
  size_t bytes[3];   // multiples of 4
};
__global__ __launch_bounds__(1024) void tiny_edge_list(EdgeParams p, int *edges, int capacity, int *counts, int *rank,
                                                       TinyZero z) {
  __shared__ int wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // (workgroups 1 .. gridDim-1 clear the ranges beside the one that builds the list; launched alone, workgroup 0 does both)
  if (blockIdx.x > 0 || gridDim.x == 1) {
    const size_t t0 = gridDim.x == 1 ? tid : (size_t)(blockIdx.x - 1) * 1024 + tid, nt = gridDim.x == 1 ? 1024 : (size_t)(gridDim.x - 1) * 1024;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (!z.ptr[k]) continue;
      const size_t n16 = (((size_t)z.ptr[k] & 15) == 0) ? z.bytes[k] / 16 : 0;
      uint4 *q = (uint4 *)z.ptr[k];
      for (size_t i = t0; i < n16; i += nt) q[i] = make_uint4(0, 0, 0, 0);
      unsigned *w = (unsigned *)z.ptr[k];
      for (size_t i = 4 * n16 + t0; i < z.bytes[k] / 4; i += nt) w[i] = 0u;
    }
    if (blockIdx.x > 0) return;
  }
  const int HW = p.H * p.W, npix = p.B * HW;
  const int ppt = (npix + 1023) / 1024;            // <= TINY_PPT_MAX (the launcher's condition)
  const int i0 = tid * ppt;
  unsigned bits = 0;
  for (int k = 0; k < ppt; ++k) {
    const int i = i0 + k;
    if (i < npix) {
      const int b = i / HW, r = i - b * HW, y = r / p.W, x = r - y * p.W;
      if (edge_pred(p, b, y, x)) bits |= 1u << k;
    }
  }
  const int c = __popc(bits);
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wv] = incl;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int t = wtot[k];
    before += k < wv ? t : 0;
    total += t;
  }
  int pos = before + incl - c;
  for (int k = 0; k < ppt; ++k) {
    const int i = i0 + k;
    if (i >= npix) break;
    const int b = i / HW, r = i - b * HW;
    if (r == 0) counts[1 + b] = pos;
    const bool on = (bits >> k) & 1u, kept = on && pos < capacity;
    if (kept) {
      const int y = r / p.W, x = r - y * p.W;
      edges[3 * (size_t)pos + 0] = b;
      edges[3 * (size_t)pos + 1] = y;
      edges[3 * (size_t)pos + 2] = x;
    }
    if (rank) rank[i] = kept ? pos : -1;
    if (on) ++pos;
  }
  if (tid == 0) counts[0] = counts[1 + p.B] = total;
}

bool tiny_edge_list_ok(int B, int H, int W) { return (long)B * H * W <= TINY_LIST_PIXELS; }

int launch_tiny_edge_list(const void *mask, int kind, int mask_channels, int B, int H, int W, int stride, float thr,
                          int *edges, int capacity, int *counts, int *rank, void *zero_a, size_t zero_a_bytes,
                          void *zero_b, size_t zero_b_bytes, void *zero_c, size_t zero_c_bytes, hipStream_t st) {
  if (!tiny_edge_list_ok(B, H, W)) return -1;
  EdgeParams p{mask, kind, mask_channels, B, H, W, stride, thr, (int)(((size_t)H * W + CHUNK - 1) / CHUNK)};
  const TinyZero z{{zero_a, zero_b, zero_c}, {zero_a ? zero_a_bytes : 0, zero_b ? zero_b_bytes : 0, zero_c ? zero_c_bytes : 0}};
  const size_t zbytes = z.bytes[0] + z.bytes[1] + z.bytes[2];
  const unsigned grid = zbytes > 4096 ? 1u + (unsigned)((zbytes + 16383) / 16384 < 16 ? (zbytes + 16383) / 16384 : 16) : 1u;
  hipLaunchKernelGGL(tiny_edge_list, dim3(grid), dim3(1024), 0, st, p, edges, capacity, counts, rank, z);
  return (int)hipGetLastError();
}

int launch_edge_mask(const float *gt, int B, int H, int W, float thr, int stride, uint8_t *out, hipStream_t st) {
  EdgeParams p{gt, 2, 3, B, H, W, stride, thr, (int)(((size_t)H * W + CHUNK - 1) / CHUNK)};
  hipLaunchKernelGGL(edge_mask_write, dim3(B * p.nblk_img), dim3(256), 0, st, p, out);
  return (int)hipGetLastError();
}

}  // namespace ssg
