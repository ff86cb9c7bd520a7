// G rows for the backward kernels of gfx950: G[n,q] = dL/dD[n,q] of every edge pixel, one wave
// per SSG row, HBM-streaming (reads the row(s), writes one row).
//
//   GRAD_LOSS: from (S_sr, S_gt): g = d(l1 + kl)/dS (L1Loss basic_loss.py:66, KLDistanceLoss
//              basic_loss.py:281, also their un-normalised sums -> `partials`), then the
//              epilogue's backward (loss_util.py:224-227 differentiated):
//              G = -(s / (sigma C k_w^2)) (g - sum_p g s)        [no sum term without generalization]
//   GRAD_S   : g = dL/dS given (similarity_map autograd)
//   GRAD_D   : G given by the caller (operator interface); only the row's border sum is produced
//
// Besides G (row-major (n, k_s^2), the layout of the reference operator's `grads`,
// similarity.h:13-23) every row gets sum_b[n] = sum of G over the "border" offsets q whose window
// is truncated by the zero rule (similarity.cu:43-47): the dense-tile backward needs it for the
// |I|^2 part of the distance.  G at the centre offset multiplies (A - B) == 0 exactly and is written
// as 0 (it is the largest entry of a row by orders of magnitude at small sigma).
//
// Rows that live in the tile-major region of a k_s = 49 call (negative row scale) are not touched by ssg_grad_rows:
// ssg_rows_tm (fused step) / ssg_rows_tm_mat (materialising call: it also writes the normalised SSG rows) below walk
// them with lanes = pixels, and the dense backward forms their G itself.
#include "ssg_common.hpp"

namespace ssg {

template <int KS, int KW>
__global__ __launch_bounds__(256) void ssg_grad_rows(GrowParams p) {
  constexpr int P = KS * KS, HP = KS / 2, HK = KW / 2, EPL = (P + 63) / 64;
  __shared__ float sred[12];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  // one group of 4 rows per workgroup (grid == p.ngroups), or -- a call with a tile-major region, where most groups
  // only find out that their rows live there -- a capped grid that walks the groups; either way group g's sums go to
  // slot g of the partial arrays (same values, same order in the finalize)
  // (a call whose dense tiles are tile-major: only the rows NOT in a dense tile are left, taken from the plan's own
  // list of them; the groups behind the list write their zero sums without reading anything)
  if (p.fix_word && blockIdx.x == 0 && threadIdx.x == 0) {   // (same value from every pass of a call: a plain store)
    *p.fix_word = __float_as_uint(loss_grad_bound(p.sigma, p.C, KW, p.w_l1, p.w_kl, p.upstream, nrows, P));
  }
  const bool via_list = p.tm_hdr && p.sparse_order && (p.only == 2 || tm_active(p.tm_hdr, p.tm_slots, nrows));
  const int n_list = via_list ? (p.tm_hdr[-1] < nrows ? p.tm_hdr[-1] : nrows) : 0;
  // (groups = what the HOST's bound on the rows allows; the live ones follow from the DEVICE count: a generous capacity
  //  then costs the zero fill below, not a workgroup per dead group)
  const int live_rows = (nrows + 3) / 4 < p.ngroups ? (nrows + 3) / 4 : p.ngroups;
  const int live_groups = via_list ? (n_list + 3) / 4 : live_rows;
  // the groups behind the list: zero sums, one thread per group (no barrier, nothing read)
  for (int g = live_groups + blockIdx.x * 256 + threadIdx.x; g < p.ngroups; g += gridDim.x * 256) {
    if (p.gmax_part) p.gmax_part[g] = 0.f;
    if (p.mode == GRAD_LOSS) {
      p.partials[2 * g] = 0.f;
      p.partials[2 * g + 1] = 0.f;
    }
  }
  for (int grp = blockIdx.x; grp < live_groups; grp += gridDim.x) {
  const int k4 = grp * 4 + wv;
  const int n = via_list ? (k4 < n_list ? (p.sparse_order[k4] & ORDER_MASK) : nrows) : k4;
  float l1p = 0.f, klp = 0.f, gmax = 0.f;
  // a NEGATIVE row scale: the row lives in the tile-major region (fused step at k_s = 49); ssg_rows_tm and the dense
  // backward own it, nothing of it is read or written here
  const bool tm_row = n < nrows && p.row_scale && p.mode == GRAD_LOSS && p.row_scale[n] < 0.0;
  // (two-chain step, the dense chain's pass: a zero row scale marks a row of the direct kernels -- the other chain's pass
  // takes it; the scales were cleared before the chains forked, so the test does not race with the other stream)
  const bool other_chain = p.only == 1 && n < nrows && p.row_scale[n] == 0.0;
  if (n < nrows && !tm_row && !other_chain) {
    const size_t base = (size_t)n * P;
    float va[EPL], vg[EPL];
    const float *src_a = p.mode == GRAD_D ? p.gin : p.ssg;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + 64 * k;
      va[k] = e < P ? src_a[base + e] : 0.f;
    }
    if (p.mode != GRAD_D) {
      const float *src_b = p.mode == GRAD_S ? p.gin : p.ssg2;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int e = lane + 64 * k;
        vg[k] = e < P ? src_b[base + e] : 0.f;
      }
      if (p.row_scale && p.mode == GRAD_LOSS) {
        // rows of the dense-tile forward arrive un-normalised: s = e * 1/(sum e + eps), the product in fp64 and
        // rounded once (what the forward's own rescale pass does), written back so that the SSG tensors the caller
        // sees are the normalised ones (not in the fused step, where the rows are the engine's scratch)
        const double sa = p.row_scale[n], sb2 = p.row_scale[(size_t)p.n_host + n];
        if (sa != 0.0) {
#pragma unroll
          for (int k = 0; k < EPL; ++k) {
            const int e = lane + 64 * k;
            va[k] = (float)(sa * (double)va[k]);
            if (e < P && !p.rows_scratch) const_cast<float *>(p.ssg)[base + e] = va[k];
          }
        }
        if (sb2 != 0.0) {
#pragma unroll
          for (int k = 0; k < EPL; ++k) {
            const int e = lane + 64 * k;
            vg[k] = (float)(sb2 * (double)vg[k]);
            if (e < P && !p.rows_scratch) const_cast<float *>(p.ssg2)[base + e] = vg[k];
          }
        }
      }
      const float invM = 1.f / ((float)nrows * (float)P);
      const float u1 = p.upstream ? p.upstream[0] : 1.f, u2 = p.upstream ? p.upstream[1] : 1.f;
      const float w1m = p.w_l1 * invM * u1, w2m = p.w_kl * invM * u2;
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int e = lane + 64 * k;
        float g = 0.f;
        if (e < P) g = p.mode == GRAD_S ? vg[k] : criteria_elem(va[k], vg[k], w1m, w2m, l1p, klp);
        vg[k] = g;
        dot = __builtin_fmaf(g, va[k], dot);
      }
      dot = p.generalization ? wave_sum(dot) : 0.f;
      const float kfac = 1.f / (p.sigma * (float)(p.C * KW * KW));
#pragma unroll
      for (int k = 0; k < EPL; ++k) va[k] = -(va[k] * kfac) * (vg[k] - dot);
    }
    float sb = 0.f, gm = 0.f;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int e = lane + 64 * k;
      if (e < P) {
        const int py = e / KS, px = e - py * KS;
        if (e == HP * KS + HP) va[k] = 0.f;
        gm = fmaxf(gm, fabsf(va[k]));
        const bool border = py < HK || py > KS - 1 - HK || px < HK || px > KS - 1 - HK;
        if (border) sb += va[k];
        if (p.G && p.mode != GRAD_D) p.G[base + e] = va[k];
      }
    }
    if (p.sum_b) {
      sb = wave_sum(sb);
      if (lane == 0) p.sum_b[n] = sb;
    }
    gmax = gm;
  }
  if (p.gmax_part) {  // deterministic mode: the fixed-point scale follows from the largest |G| (ssg_common.hpp)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
    if (lane == 0) sred[8 + wv] = gmax;
    __syncthreads();
    if (threadIdx.x == 0) p.gmax_part[grp] = fmaxf(fmaxf(sred[8], sred[9]), fmaxf(sred[10], sred[11]));
  }
  if (p.mode == GRAD_LOSS) {
    l1p = wave_sum(l1p);
    klp = wave_sum(klp);
    if (lane == 0) {
      sred[wv] = l1p;
      sred[4 + wv] = klp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      p.partials[2 * grp] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
      p.partials[2 * grp + 1] = (sred[4] + sred[5]) + (sred[6] + sred[7]);
    }
  }
  if (gridDim.x < (unsigned)live_groups) __syncthreads();   // (sred is reused by the next group)
  }
}

// The row pass over TILE-MAJOR rows (TmRowsParams, ssg_common.hpp): one workgroup of 16 waves per tile; wave (ck, part)
// walks the offset rows of its eighth of the search area for the 64 pixels of chunk ck -- lane = pixel, so that every
// load is one aligned 256-byte run and nothing is exchanged between lanes until the end.  s = tm_apply(e, row scale)
// (the same operations the dense backward uses: both see the same bits), then per pixel
//   criteria sums (L1Loss basic_loss.py:66, KLDistanceLoss basic_loss.py:281; same operations as criteria_elem),
//   dot = sum_q g s (loss_util.py:224-227 differentiated: G = -(s k)(g - dot)),
//   sum_b = sum of G over the offsets with a truncated window = -k (sum_b g s - dot sum_b s),
//   an upper bound of |G| for the fixed-point scale of the deterministic accumulation.  The exact maximum would need
//   dot before the pass; the bound splits g = g_l1 + g_kl: |s (g_l1 - dot_l1)| <= s (|w1| + |dot_l1|), and with
//   s g_kl = -w2 t' (t' = the clamped s_gt, 0 where s < 1e-10), s (g_kl - dot_kl) = w2 (s - t') - s (dot_kl + w2): the
//   first term is tracked per element, the second vanishes as sum t -> 1.  At most ~2x above the true maximum whether
//   the L1 or the KL part dominates (a bit of the 11 spare bits of the fixed-point format).
template <int KS, int KW>
__global__ __launch_bounds__(1024) void ssg_rows_tm(TmRowsParams p) {
  constexpr int P = KS * KS, HP = KS / 2, HK = KW / 2, NQ = 8, TY = 4, TX = 32, UNR = 7;
  static_assert(KS % UNR == 0 && TY * TX == TM_PX, "offset rows in groups of 7; 4 x 32 tiles");
  __shared__ float red[8][NQ][TM_PX];
  __shared__ double redk[NQ][TM_PX];
  __shared__ float wred[3][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ck = wv & 1, part = wv >> 1;
  const int tslot = blockIdx.x;
  if (tslot >= dense_tile_count(p.n_dense) || p.n_dense[1] != TY ||
      !tm_active(p.n_dense, p.tm_slots, rows_to_do(p.n_dev, p.n_host))) {
    if (threadIdx.x < 2) {   // (two partial slots per tile: rows_tm_parts)
      p.partials[2 * (2 * tslot + threadIdx.x)] = 0.f;
      p.partials[2 * (2 * tslot + threadIdx.x) + 1] = 0.f;
      if (p.gmax_part) p.gmax_part[2 * tslot + threadIdx.x] = 0.f;
    }
    return;
  }
  const int H = p.H, W = p.W;
  const int tx_n = (W + TX - 1) / TX, ty_n = (H + TY - 1) / TY;
  const int tile = dense_tile_id(dense_tile_at(p.n_dense, p.tiles, p.B * ty_n * tx_n, tslot));
  const int b = tile / (tx_n * ty_n), tr = tile - b * tx_n * ty_n;
  const int ty0 = (tr / tx_n) * TY, tx0 = (tr % tx_n) * TX;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const int y = ty0 + tm_pixel_row(ck, lane), x = tx0 + tm_pixel_col(lane);
  int r = (y < H && x < W) ? p.rank[((size_t)b * H + y) * W + x] : -1;
  if (r >= nrows) r = -1;
  // (holes of the tile: scale 0 -> s = t = 0 -> every sum below gets an exact 0 from them)
  const TmScale sa = tm_scale(r >= 0 ? p.row_scale[r] : 0.0), sb = tm_scale(r >= 0 ? p.row_scale[(size_t)p.n_host + r] : 0.0);
  const float invM = 1.f / ((float)nrows * (float)P);
  const float u1 = p.upstream ? p.upstream[0] : 1.f, u2 = p.upstream ? p.upstream[1] : 1.f;
  const float w1m = p.w_l1 * invM * u1, w2m = p.w_kl * invM * u2;
  const float cl = 1e-10f;
  const float *pa = p.tm[0] + (size_t)tslot * P * TM_PX + ck * 64 + lane;
  const float *pb = p.tm[1] + (size_t)tslot * P * TM_PX + ck * 64 + lane;
  float l1 = 0.f, d1 = 0.f, d2 = 0.f, bs = 0.f, bg = 0.f, m1 = 0.f, m2 = 0.f;
  // KL: first-order terms t log(t/s) of either sign that cancel to second order over a row (sigma = 1: to 1e-3 of
  // their size) -- fp32 inside an offset row (49 terms), fp64 across the rows; a 300-term fp32 chain costs the
  // cancelled total 2e-5 (measured against the fp64 KL of the same SSGs)
  double kld = 0.0;
  const int qy0 = (KS * part) / NQ, qy1 = (KS * (part + 1)) / NQ;
  for (int qy = qy0; qy < qy1; ++qy) {
    const bool yb = qy < HK || qy > KS - 1 - HK, yc = qy == HP;
    float kl = 0.f;
#pragma unroll 1
    for (int g0 = 0; g0 < KS; g0 += UNR) {   // (rolled: unrolled, hipcc lifts all 98 loads of the offset row to its top and spills)
      float ea[UNR], eb[UNR];
#pragma unroll
      for (int j = 0; j < UNR; ++j) {
        ea[j] = __builtin_nontemporal_load(pa + (size_t)(qy * KS + g0 + j) * TM_PX);
        eb[j] = __builtin_nontemporal_load(pb + (size_t)(qy * KS + g0 + j) * TM_PX);
      }
#pragma unroll
      for (int j = 0; j < UNR; ++j) {
        const int qx = g0 + j;
        const float a = tm_apply(ea[j], sa), t = tm_apply(eb[j], sb);
        // criteria_elem's operations (ssg_common.hpp), with the two parts of g kept apart
        const float ac = fmaxf(a, cl), bc = fmaxf(t, cl);
        l1 += fabsf(a - t);
        const float rc = __builtin_amdgcn_rcpf(ac);
        const float r0 = bc * rc;
        const float ratio = __builtin_fmaf(__builtin_fmaf(-r0, ac, bc), rc, r0);
        kl += bc * (0.69314718056f * __builtin_amdgcn_logf(ratio));
        // s g in the form the dense backward uses (s * t'/s = t' without the division; sgn by scaling and clamping):
        // d1 = sum s sgn(s - t), d2 = sum t' -- the weights come in at the end
        const float sg = __builtin_amdgcn_fmed3f((a - t) * 0x1p126f, -1.f, 1.f);
        const float bz = a >= cl ? bc : 0.f;
        d1 = __builtin_fmaf(sg, a, d1);
        d2 += bz;
        const float gs = __builtin_fmaf(w1m * sg, a, -w2m * bz);
        // wave-uniform 0/1 factors instead of branches: border offset; not the centre offset (G there multiplies
        // A - B == 0 and is dropped: not part of the bound)
        const float bm = (yb || qx < HK || qx > KS - 1 - HK) ? 1.f : 0.f, cm = (qx == HP && yc) ? 0.f : 1.f;
        bs = __builtin_fmaf(a, bm, bs);
        bg = __builtin_fmaf(gs, bm, bg);
        m1 = fmaxf(m1, a * cm);
        m2 = fmaxf(m2, fabsf(a - bz) * cm);
      }
    }
    kld += (double)kl;
  }
  red[0][part][ck * 64 + lane] = l1;
  redk[part][ck * 64 + lane] = kld;
  red[2][part][ck * 64 + lane] = d1;
  red[3][part][ck * 64 + lane] = d2;
  red[4][part][ck * 64 + lane] = bs;
  red[5][part][ck * 64 + lane] = bg;
  red[6][part][ck * 64 + lane] = m1;
  red[7][part][ck * 64 + lane] = m2;
  __syncthreads();
  float l1w = 0.f, klw = 0.f, gb = 0.f;
  if (part == 0) {   // waves 0 and 1: one lane per pixel, the eight parts in a fixed order
    const int px = ck * 64 + lane;
    float v[8];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k == 1) continue;
      float t = red[k][0][px];
#pragma unroll
      for (int j = 1; j < NQ; ++j) t += red[k][j][px];
      v[k] = t;
    }
    {
      double t = redk[0][px];
#pragma unroll
      for (int j = 1; j < NQ; ++j) t += redk[j][px];
      v[1] = (float)t;   // (the pixel's KL: non-negative up to rounding, no cancellation left)
    }
#pragma unroll
    for (int k = 6; k < 8; ++k) {
      float t = red[k][0][px];
#pragma unroll
      for (int j = 1; j < NQ; ++j) t = fmaxf(t, red[k][j][px]);
      v[k] = t;
    }
    const float kfac = 1.f / (p.sigma * (float)(p.C * KW * KW));
    v[2] *= w1m;    // dot_l1 = w1 sum s sgn(s - t)
    v[3] *= -w2m;   // dot_kl = -w2 sum t'
    const float dot = v[2] + v[3];
    if (r >= 0) {
      p.dot[r] = dot;
      p.sum_b[r] = -kfac * (v[5] - dot * v[4]);
    }
    gb = kfac * (v[6] * (fabsf(w1m) + fabsf(v[2]) + fabsf(v[3] + w2m)) + fabsf(w2m) * v[7]) * 1.0001f;
    l1w = wave_sum(v[0]);
    klw = wave_sum(v[1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gb = fmaxf(gb, __shfl_xor(gb, o, 64));
    if (lane == 0) {
      wred[0][ck] = l1w;
      wred[1][ck] = klw;
      wred[2][ck] = gb;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2) {   // one partial slot per chunk, as the materialising kernel leaves them
    p.partials[2 * (2 * tslot + threadIdx.x)] = wred[0][threadIdx.x];
    p.partials[2 * (2 * tslot + threadIdx.x) + 1] = wred[1][threadIdx.x];
    if (p.gmax_part) p.gmax_part[2 * tslot + threadIdx.x] = wred[2][threadIdx.x];
  }
}

// The same pass when the call MATERIALISES its SSG tensors (ssg_sr / ssg_gt are the caller's (n, k_s^2) row-major rows):
// besides the sums above it writes the normalised rows -- a transpose from [offset][pixel] to [pixel][offset] through LDS.
// Round 5 layout (C5: 3.0 -> see profiles/EXPERIMENTS.md):
//   * one workgroup of 8 waves per 64-pixel CHUNK of a tile (two per tile, ~68 KB of LDS each: two workgroups per CU, so
//     that one's store phase runs beside the other's loads -- round 4's 1,024-thread workgroup held 140 KB and the CU alone);
//   * wave = one eighth of the 49 offsets of an offset row (6-7 columns), lane = pixel: every load is an aligned 256-byte
//     run, the next hand-over's loads are in flight while the current one is worked on;
//   * per (pixel, image) the LDS holds a RING of 128 floats indexed by the offset q (+ a per-row rotation that gives ring
//     and SSG tensor the same 16-byte phase).  After every NR = 2 offset rows the workgroup writes out, per pixel, every
//     64-byte-ALIGNED segment of the pixel's SSG row that is complete; the <= 15 floats behind it stay in the ring for the
//     next hand-over.  Every store is a whole aligned dwordx4, eight lanes = 128 contiguous bytes, and no 64-byte sector
//     of the tensors is written twice (round 4 wrote 392-byte runs at 4-byte alignment with scalar dword stores: both
//     ends of every run were partial sectors written twice; the first and last segment of a ROW still are -- 2 of 151).
template <int KS, int KW>
__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(4, 4))) void ssg_rows_tm_mat(TmRowsParams p) {
  constexpr int P = KS * KS, HP = KS / 2, HK = KW / 2, NQ = 8, TY = 4, TX = 32, QMAX = (KS + NQ - 1) / NQ;
  constexpr int CPX = 64, NR = 2, RING = 128, PITCH = RING + 4, SEG = 16, NIT = (KS + NR - 1) / NR;
  static_assert(TY * TX == TM_PX && QMAX == 7, "4 x 32 tiles; at most 7 offsets of a row per part");
  static_assert(NR * KS + SEG - 1 <= RING, "a hand-over's new offsets + the carried ones fit the ring");
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *stage = lds;                                   // [2 images][CPX][PITCH]
  int *prow = (int *)(stage + 2 * CPX * PITCH);         // [CPX] row of every pixel of the chunk (-1: hole)
  float *red = lds;                                     // after the main loop, over the stage: [8][NQ][CPX]
  double *redk = (double *)(red + 8 * NQ * CPX);        // [NQ][CPX]
  float *wred = (float *)(redk + NQ * CPX);             // [3]
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int slot = blockIdx.x;
  const int tslot = slot >> 1, ck = slot & 1;
  if (tslot >= dense_tile_count(p.n_dense) || p.n_dense[1] != TY ||
      !tm_active(p.n_dense, p.tm_slots, rows_to_do(p.n_dev, p.n_host))) {
    if (threadIdx.x == 0) {
      p.partials[2 * slot] = 0.f;
      p.partials[2 * slot + 1] = 0.f;
      if (p.gmax_part) p.gmax_part[slot] = 0.f;
    }
    return;
  }
  const int H = p.H, W = p.W;
  const int tx_n = (W + TX - 1) / TX, ty_n = (H + TY - 1) / TY;
  const int tile = dense_tile_id(dense_tile_at(p.n_dense, p.tiles, p.B * ty_n * tx_n, tslot));
  const int b = tile / (tx_n * ty_n), tr = tile - b * tx_n * ty_n;
  const int ty0 = (tr / tx_n) * TY, tx0 = (tr % tx_n) * TX;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const int y = ty0 + tm_pixel_row(ck, lane), x = tx0 + tm_pixel_col(lane);
  int r = (y < H && x < W) ? p.rank[((size_t)b * H + y) * W + x] : -1;
  if (r >= nrows) r = -1;
  if (part == 0) prow[lane] = r;
  const TmScale sa = tm_scale(r >= 0 ? p.row_scale[r] : 0.0), sb = tm_scale(r >= 0 ? p.row_scale[(size_t)p.n_host + r] : 0.0);
  const float invM = 1.f / ((float)nrows * (float)P);
  const float u1 = p.upstream ? p.upstream[0] : 1.f, u2 = p.upstream ? p.upstream[1] : 1.f;
  const float w1m = p.w_l1 * invM * u1, w2m = p.w_kl * invM * u2;
  const float cl = 1e-10f;
  // the tensors' own phase: out = aligned-down pointer + goff floats (a caller's pointer need not be 64-byte aligned)
  float *out_a = const_cast<float *>(p.out[0]), *out_b = const_cast<float *>(p.out[1]);
  const int goff_a = (int)(((uintptr_t)out_a >> 2) & (SEG - 1)), goff_b = (int)(((uintptr_t)out_b >> 2) & (SEG - 1));
  out_a -= goff_a;
  out_b -= goff_b;
  const int qx0 = (KS * part) / NQ, qn = (KS * (part + 1)) / NQ - qx0;   // this part's columns of every offset row
  const float *pa = p.tm[0] + (size_t)tslot * P * TM_PX + (size_t)qx0 * TM_PX + ck * 64 + lane;
  const float *pb = p.tm[1] + (size_t)tslot * P * TM_PX + (size_t)qx0 * TM_PX + ck * 64 + lane;
  // ring position of offset q of this lane's pixel: (q + rot) & 127 -- rot = the 16-byte phase of the pixel's row in the
  // tensor, so that a 4-float group aligned in the tensor is aligned (and unbroken by the wrap) in the ring
  const int rot_a = (int)(((unsigned)goff_a + (unsigned)(r > 0 ? r : 0) * (unsigned)(P & 3)) & 3u);
  const int rot_b = (int)(((unsigned)goff_b + (unsigned)(r > 0 ? r : 0) * (unsigned)(P & 3)) & 3u);
  float *ring_a = stage + lane * PITCH, *ring_b = ring_a + CPX * PITCH;
  float l1 = 0.f, d1 = 0.f, d2 = 0.f, bs = 0.f, bg = 0.f, m1 = 0.f, m2 = 0.f;
  double kld = 0.0;
  float na[NR][QMAX], nb[NR][QMAX];   // the next hand-over's values, in flight while the current one is worked on
#pragma unroll
  for (int rr = 0; rr < NR; ++rr)
#pragma unroll
    for (int j = 0; j < QMAX; ++j) {
      const int jj = j < qn ? j : 0;
      na[rr][j] = __builtin_nontemporal_load(pa + (size_t)(rr * KS + jj) * TM_PX);
      nb[rr][j] = __builtin_nontemporal_load(pb + (size_t)(rr * KS + jj) * TM_PX);
    }
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  lds_barrier();   // (prow)
  // the store phase's identity: eight lanes per pixel, each a 16-byte quarter of two consecutive 64-byte segments
  const int spx = threadIdx.x >> 3, sj = threadIdx.x & 7;
  const int srow = prow[spx];
  // the pixel's row in each tensor, taken from its own 64-byte boundary: `sbase` = floats between that boundary and the row's
  // first element (0..15), `rowp` = the boundary; everything below is 32-bit arithmetic relative to it (round 5 fix: row
  // index x k_s^2 as a 32-bit product overflowed from 894 k rows on -- four dense 512 x 512 images)
  const size_t srow0 = (size_t)(srow > 0 ? srow : 0) * P;
  const int sbase_a = (int)((srow0 + goff_a) & (SEG - 1)), sbase_b = (int)((srow0 + goff_b) & (SEG - 1));
  float *const rowp_a = out_a + (srow0 + goff_a - sbase_a), *const rowp_b = out_b + (srow0 + goff_b - sbase_b);
  int kc_a = 0, kc_b = 0;                                          // next segment of the row to write
  const float *sring_a = stage + spx * PITCH, *sring_b = sring_a + CPX * PITCH;
#pragma unroll 1
  for (int it = 0; it < NIT; ++it) {
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const int qy = NR * it + rr;
      if (qy < KS) {
        const bool yb = qy < HK || qy > KS - 1 - HK, yc = qy == HP;
        const int q0 = qy * KS + qx0;
        float kl = 0.f;
        // normalise first, then re-issue this offset row's registers for the same row of the NEXT hand-over: the loads
        // are a whole hand-over ahead of their use and only NR x 14 values are ever in flight per lane
        float av[QMAX], tv[QMAX];
#pragma unroll
        for (int j = 0; j < QMAX; ++j) {
          av[j] = tm_apply(na[rr][j], sa);
          tv[j] = tm_apply(nb[rr][j], sb);
        }
        if (qy + NR < KS) {
#pragma unroll
          for (int j = 0; j < QMAX; ++j) {
            const int jj = j < qn ? j : 0;
            na[rr][j] = __builtin_nontemporal_load(pa + (size_t)((qy + NR) * KS + jj) * TM_PX);
            nb[rr][j] = __builtin_nontemporal_load(pb + (size_t)((qy + NR) * KS + jj) * TM_PX);
          }
        }
#pragma unroll
        for (int j = 0; j < QMAX; ++j) {
          if (j < qn) {   // (wave-uniform: parts 0..6 have six columns, part 7 seven)
            const int qx = qx0 + j;
            const float a = av[j], t = tv[j];
            ring_a[(q0 + j + rot_a) & (RING - 1)] = a;
            ring_b[(q0 + j + rot_b) & (RING - 1)] = t;
            const float ac = fmaxf(a, cl), bc = fmaxf(t, cl);
            l1 += fabsf(a - t);
            const float rc = __builtin_amdgcn_rcpf(ac);
            const float r0 = bc * rc;
            const float ratio = __builtin_fmaf(__builtin_fmaf(-r0, ac, bc), rc, r0);
            kl += bc * (0.69314718056f * __builtin_amdgcn_logf(ratio));
            const float sg = __builtin_amdgcn_fmed3f((a - t) * 0x1p126f, -1.f, 1.f);
            const float bz = a >= cl ? bc : 0.f;
            d1 = __builtin_fmaf(sg, a, d1);
            d2 += bz;
            const float gs = __builtin_fmaf(w1m * sg, a, -w2m * bz);
            const float bm = (yb || qx < HK || qx > KS - 1 - HK) ? 1.f : 0.f, cm = (qx == HP && yc) ? 0.f : 1.f;
            bs = __builtin_fmaf(a, bm, bs);
            bg = __builtin_fmaf(gs, bm, bg);
            m1 = fmaxf(m1, a * cm);
            m2 = fmaxf(m2, fabsf(a - bz) * cm);
          }
        }
        kld += (double)kl;
      }
    }
    // (barriers for the LDS hand-over only: a __syncthreads() would also drain the stores of the last hand-over and
    // the next hand-over's loads -- two memory round trips per hand-over)
    lds_barrier();   // offsets < qa of every pixel of the chunk are in the rings
    {
      const int qa = NR * KS * (it + 1) < P ? NR * KS * (it + 1) : P;
#pragma unroll
      for (int img = 0; img < 2; ++img) {
        const int sbase = img ? sbase_b : sbase_a;
        int &kc = img ? kc_b : kc_a;
        const float *sring = img ? sring_b : sring_a;
        float *outp = img ? rowp_b : rowp_a;
        // complete segments: below the first float not yet in the ring; the row's last hand-over takes its partial tail
        const int kend = qa == P ? (sbase + P + SEG - 1) / SEG : (sbase + qa) / SEG;
        if (srow >= 0) {
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {   // (at most 7 segments per hand-over)
            const int seg = kc + 2 * s2 + (sj >> 2);
            if (seg < kend) {
              const int gi = seg * SEG + 4 * (sj & 3), q = gi - sbase;
              const f4 v = *(const f4 *)(sring + ((q + (sbase & 3)) & (RING - 1)));
              if (q >= 0 && q + 3 < P) {
                __builtin_nontemporal_store(v, (f4 *)(outp + (size_t)gi));   // (plain stores: 2.50 instead of 2.25 ms)
              } else {   // the row's first / last segment: the floats that belong to this row only
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (q + i >= 0 && q + i < P) outp[(size_t)gi + i] = v[i];
              }
            }
          }
        }
        kc = kend;
      }
    }
    lds_barrier();   // (the rings are free for the next hand-over)
  }
  red[(0 * NQ + part) * CPX + lane] = l1;
  redk[part * CPX + lane] = kld;
  red[(2 * NQ + part) * CPX + lane] = d1;
  red[(3 * NQ + part) * CPX + lane] = d2;
  red[(4 * NQ + part) * CPX + lane] = bs;
  red[(5 * NQ + part) * CPX + lane] = bg;
  red[(6 * NQ + part) * CPX + lane] = m1;
  red[(7 * NQ + part) * CPX + lane] = m2;
  __syncthreads();
  if (part == 0) {   // wave 0: one lane per pixel, the eight parts in a fixed order
    float v[8];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k == 1) continue;
      float t = red[(k * NQ) * CPX + lane];
#pragma unroll
      for (int j = 1; j < NQ; ++j) t += red[(k * NQ + j) * CPX + lane];
      v[k] = t;
    }
    {
      double t = redk[lane];
#pragma unroll
      for (int j = 1; j < NQ; ++j) t += redk[j * CPX + lane];
      v[1] = (float)t;
    }
#pragma unroll
    for (int k = 6; k < 8; ++k) {
      float t = red[(k * NQ) * CPX + lane];
#pragma unroll
      for (int j = 1; j < NQ; ++j) t = fmaxf(t, red[(k * NQ + j) * CPX + lane]);
      v[k] = t;
    }
    const float kfac = 1.f / (p.sigma * (float)(p.C * KW * KW));
    v[2] *= w1m;
    v[3] *= -w2m;
    const float dot = v[2] + v[3];
    if (r >= 0) {
      p.dot[r] = dot;
      p.sum_b[r] = -kfac * (v[5] - dot * v[4]);
    }
    float gb = kfac * (v[6] * (fabsf(w1m) + fabsf(v[2]) + fabsf(v[3] + w2m)) + fabsf(w2m) * v[7]) * 1.0001f;
    const float l1w = wave_sum(v[0]), klw = wave_sum(v[1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gb = fmaxf(gb, __shfl_xor(gb, o, 64));
    if (lane == 0) {
      p.partials[2 * slot] = l1w;
      p.partials[2 * slot + 1] = klw;
      if (p.gmax_part) p.gmax_part[slot] = gb;
    }
  }
  (void)wred;
}

// Both kernels leave TWO partial slots per tile (the materialising kernel runs one workgroup per 64-pixel chunk).
int rows_tm_parts(int n_tiles) { return 2 * n_tiles; }

int launch_rows_tm(const TmRowsParams &p, int ks, int kw, hipStream_t st) {
  if (p.n_tiles <= 0) return 0;
  if (ks != 49 || kw != 13) return -1;
  if (p.out[0] && p.out[1]) {
    const size_t lds0 = sizeof(float) * (size_t)(2 * 64 * (128 + 4)) + sizeof(int) * 64, lds1 = sizeof(float) * 8 * 8 * 64 + sizeof(double) * 8 * 64 + 64;
    const size_t lds = lds0 > lds1 ? lds0 : lds1;
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(ssg_rows_tm_mat<49, 13>, (int)lds, lds_set)) return rc;
    hipLaunchKernelGGL((ssg_rows_tm_mat<49, 13>), dim3(2u * (unsigned)p.n_tiles), dim3(512), lds, st, p);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL((ssg_rows_tm<49, 13>), dim3((unsigned)p.n_tiles), dim3(1024), 0, st, p);
  return (int)hipGetLastError();
}

// ---- the two criteria on plain tensors (rows a8 / a9 standalone: L1Loss, KLDistanceLoss of basic_loss.py:41-66,269-282
// applied to SSG tensors that already exist -- the eager `similarity_map` path and every fallback of the deferred
// handles).  One streaming pass gives both sums (fp32 inside a lane's ~100 elements, fp64 from there on, fixed order:
// bit-reproducible); one pass gives d(c1 sum|a-b| + c2 sum kl)/da with the two coefficients read on the device.
constexpr int CRIT_GRID = 2048;

__global__ __launch_bounds__(256) void criteria_sums_kernel(const float *a, const float *b, size_t n, double *part) {
  __shared__ double w1[4], w2[4];
  float l1 = 0.f, kl = 0.f;
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
  const float4 *a4 = (const float4 *)a, *b4 = (const float4 *)b;
  const bool vec = (((size_t)a | (size_t)b) & 15) == 0;
  double L1 = 0.0, KL = 0.0;
  if (vec) {
    int run = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 x = a4[i], y = b4[i];
      criteria_elem_any(x.x, y.x, 0.f, 0.f, l1, kl);
      criteria_elem_any(x.y, y.y, 0.f, 0.f, l1, kl);
      criteria_elem_any(x.z, y.z, 0.f, 0.f, l1, kl);
      criteria_elem_any(x.w, y.w, 0.f, 0.f, l1, kl);
      if (++run == 32) {   // (128 elements per fp32 partial sum)
        L1 += (double)l1; KL += (double)kl; l1 = kl = 0.f; run = 0;
      }
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) criteria_elem_any(a[i], b[i], 0.f, 0.f, l1, kl);
  } else {
    int run = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      criteria_elem_any(a[i], b[i], 0.f, 0.f, l1, kl);
      if (++run == 128) {
        L1 += (double)l1; KL += (double)kl; l1 = kl = 0.f; run = 0;
      }
    }
  }
  L1 += (double)l1;
  KL += (double)kl;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    L1 += __shfl_down(L1, o, 64);
    KL += __shfl_down(KL, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    w1[threadIdx.x >> 6] = L1;
    w2[threadIdx.x >> 6] = KL;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = (w1[0] + w1[1]) + (w1[2] + w1[3]);
    part[2 * blockIdx.x + 1] = (w2[0] + w2[1]) + (w2[2] + w2[3]);
  }
}

__global__ __launch_bounds__(256) void criteria_finish_kernel(const double *part, int nparts, float *out) {
  __shared__ double s1[256], s2[256];
  double x = 0.0, y = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    x += part[2 * i];
    y += part[2 * i + 1];
  }
  s1[threadIdx.x] = x;
  s2[threadIdx.x] = y;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s1[threadIdx.x] += s1[threadIdx.x + o];
      s2[threadIdx.x] += s2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)s1[0];
    out[1] = (float)s2[0];
  }
}

__global__ __launch_bounds__(256) void criteria_grad_kernel(const float *a, const float *b, size_t n, const float *coef,
                                                            float *g) {
  const float c1 = coef[0], c2 = coef[1];
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
  float d1 = 0.f, d2 = 0.f;
  if (((((size_t)a | (size_t)b | (size_t)g)) & 15) == 0) {
    const float4 *a4 = (const float4 *)a, *b4 = (const float4 *)b;
    float4 *g4 = (float4 *)g;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 x = a4[i], y = b4[i];
      float4 r;
      r.x = criteria_elem_any(x.x, y.x, c1, c2, d1, d2);
      r.y = criteria_elem_any(x.y, y.y, c1, c2, d1, d2);
      r.z = criteria_elem_any(x.z, y.z, c1, c2, d1, d2);
      r.w = criteria_elem_any(x.w, y.w, c1, c2, d1, d2);
      g4[i] = r;
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) g[i] = criteria_elem_any(a[i], b[i], c1, c2, d1, d2);
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) g[i] = criteria_elem_any(a[i], b[i], c1, c2, d1, d2);
  }
}

size_t criteria_scratch_bytes() { return sizeof(double) * 2 * CRIT_GRID; }

int launch_criteria_sums(const float *a, const float *b, size_t n, void *scratch, float *sums_out, hipStream_t st) {
  const size_t want = (n / 4 + 255) / 256;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > CRIT_GRID ? CRIT_GRID : want));
  hipLaunchKernelGGL(criteria_sums_kernel, dim3(grid), dim3(256), 0, st, a, b, n, (double *)scratch);
  hipLaunchKernelGGL(criteria_finish_kernel, dim3(1), dim3(256), 0, st, (const double *)scratch, (int)grid, sums_out);
  return (int)hipGetLastError();
}

int launch_criteria_grad(const float *a, const float *b, size_t n, const float *coef, float *g, hipStream_t st) {
  const size_t want = (n / 4 + 255) / 256;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 4 * CRIT_GRID ? 4 * CRIT_GRID : want));
  hipLaunchKernelGGL(criteria_grad_kernel, dim3(grid), dim3(256), 0, st, a, b, n, coef, g);
  return (int)hipGetLastError();
}

bool grow_supported(int ks, int kw) { return (ks == 25 && kw == 9) || (ks == 49 && kw == 13); }

unsigned grow_grid(int n_host) { return (unsigned)((n_host + 3) / 4); }
constexpr int GROW_GRID_MAX = 32768;

int launch_grad_rows(const GrowParams &p0, int ks, int kw, hipStream_t st) {
  if (p0.n_host <= 0) return 0;
  GrowParams p = p0;
  p.ngroups = (int)grow_grid(p.n_host);
  // (a call without its own cap: at most GROW_GRID_MAX workgroups -- beyond that the bound on the rows is far above any
  //  count the kernel will see at this image size, and the workgroups walk the groups)
  const int cap = p.grid_cap > 0 ? p.grid_cap : GROW_GRID_MAX;
  const unsigned grid = cap < p.ngroups ? (unsigned)cap : (unsigned)p.ngroups;
  if (ks == 25 && kw == 9)
    hipLaunchKernelGGL((ssg_grad_rows<25, 9>), dim3(grid), dim3(256), 0, st, p);
  else if (ks == 49 && kw == 13)
    hipLaunchKernelGGL((ssg_grad_rows<49, 13>), dim3(grid), dim3(256), 0, st, p);
  else
    return -1;
  return (int)hipGetLastError();
}

}  // namespace ssg
