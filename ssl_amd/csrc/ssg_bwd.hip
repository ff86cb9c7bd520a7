// Backward SSG kernels for gfx950: dL/dimg from G = dL/dD, per edge pixel on
// chip, one fp32 atomic per touched image pixel (the reference issues
// ~2*C*k_w^2 global atomics per (edge pixel, search offset), similarity.cu:123-128).
//
// For one edge pixel with centre window A (k_w x k_w) and search tile S (k_s x k_s)
// the reference's scatter is algebraically
//   gS[c,t] = -2 ( sum_k Gz[t-k] A[c,k]  -  S[c,t] * sum_k Gz[t-k] )      t in tile
//   gA[c,k] =  2 ( A[c,k] * sum_p G[p]   -  sum_t Gz[t-k] S[c,t] )        k in window
// (Gz = G zero-extended outside the k_s x k_s offsets; the out-of-area rule
// "B = 0" contributes exactly the A*sum G part).  Both sums are correlations
// of the k_s x k_s G tile with a k_w x k_w stencil and reuse the forward's
// decomposition: a lane owns a BS x BS block of tile positions t and streams the
// (BS+k_w-1)^2 patch of Gz it needs from LDS row by row (ssg_common.hpp).
//   pass A: acc[t]  += Gz[t+k'] * Aflip[c,k']      (25 accumulators, A in registers)
//   pass B: P[k']   += Gz[t+k'] * S[c,t]           (per-lane partials, reduced
//           across the job's 25 lanes through an LDS slice in a fixed order)
// G itself comes from one of three sources (GradMode): dL/dD directly (the
// reference operator's backward), dL/dS + saved S (similarity_map autograd), or
// S_sr/S_gt for the fused L1 + KL criteria, whose partial sums are also
// produced here (L1Loss basic_loss.py:66, KLDistanceLoss basic_loss.py:281).
#include "ssg_common.hpp"

namespace ssg {

// out[j] = sum_{k=0}^{KW-1} v[j+k] for j < BS, BS <= KW: the taps BS-1 .. KW-1 belong to every window and are
// summed once; window j adds the suffix v[j .. BS-2] and the prefix v[KW .. KW-1+j] (running sums).
template <int BS, int KW>
__device__ __forceinline__ void shared_window_sums(const float (&v)[BS + KW - 1], float (&out)[BS]) {
  static_assert(BS <= KW, "windows must overlap in at least one tap");
  float mid = v[BS - 1];
#pragma unroll
  for (int k = BS; k < KW; ++k) mid += v[k];
  float suf[BS], pre[BS];  // suf[j] = v[j] + ... + v[BS-2] (j < BS-1), pre[j] = v[KW] + ... + v[KW-1+j] (j >= 1)
  suf[BS - 1] = 0.f;
#pragma unroll
  for (int j = BS - 2; j >= 0; --j) suf[j] = (j == BS - 2) ? v[j] : v[j] + suf[j + 1];
  pre[0] = 0.f;
#pragma unroll
  for (int j = 1; j < BS; ++j) pre[j] = (j == 1) ? v[KW] : pre[j - 1] + v[KW - 1 + j];
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    float t = mid;
    if (j < BS - 1) t += suf[j];
    if (j > 0) t += pre[j];
    out[j] = t;
  }
}

// Row `ry` of a job's zero-extended G tile, columns cx0 .. cx0+PW-1.  Tile rows are KS values followed
// by HK zeros (stride KS+HK): the zeros after row r-1 are also the zeros before row r, so columns
// -HK .. KS+HK-1 need no masking; rows outside the tile come from the all-zero row.
template <class G>
__device__ __forceinline__ void load_grow(const float *tg, const float *zrow, int ry, int cx0, float (&out)[G::PW]) {
  const float *rowp = ((unsigned)ry < (unsigned)G::KS) ? (tg + ry * G::GS) : zrow;
#pragma unroll
  for (int j = 0; j < G::PW; ++j) out[j] = rowp[cx0 + j];
}

// Workgroup w takes JOBS consecutive entries of the job order: either the caller's row order
// (reference operator interface: arbitrary pos list) or, when `p.order` is given, the tile-major
// permutation the edge-list builder wrote (8x8 image tiles, row-major inside a tile).  In tile
// order a workgroup's JOBS edge pixels sit within a few pixels of each other, so their gradient
// tiles are first summed into ONE LDS window (plain read-modify-write, one job at a time: LDS
// fp32 atomics run at 0.4 lane-ops/clk/CU on gfx950, measured) and only that window goes to HBM
// with fp32 atomics -- 3.5x fewer global atomics, which is what bounds this kernel.
// One group of JOBS rows; `vb` plays blockIdx.x (the main kernel passes it, the tail kernel walks it).  Returns 1 when the
// group ran (its LDS is in use), -1 when vb maps behind the last group.
template <class G, int KHC>
__device__ __forceinline__ int bwd_tiled_group(const BwdParams &p, const int vb) {
  constexpr int KS = G::KS, KW = G::KW, BS = G::BS, WG = G::WG;
  constexpr int HP = G::HP, HK = G::HK, P = G::P, NB = G::NB, LPJ = G::LPJ;
  constexpr int JOBS = G::JOBS, PW = G::PW, S = G::GS, CHG = KS * S;  // G tile: row stride, size
  constexpr int PADF = (HK + 3) & ~3;
  constexpr int NCH = (KW + KHC - 1) / KHC;  // pass-B chunks of KHC stencil rows
  constexpr int SL = KHC * KW;               // partials per chunk
  constexpr int EPL = (P + LPJ - 1) / LPJ;   // row elements per lane
  constexpr int MH = KS + MERGE_ROWS - 1, MW = KS + MERGE_COLS - 1;   // merge window: centres within 8 rows x 16 columns

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = p.C, H = p.H, W = p.W;
  float *gt = smem + PADF;                    // [JOBS][KS][S]   G tiles, HK zeros after every row
  float *zero = gt + JOBS * CHG;              // zero row
  float *red = zero + ((G::ZROW + 3) & ~3);   // [JOBS][SL][LPJ] pass-B slice
  float *red2 = red + JOBS * SL * LPJ;        // [WG] scalar reductions
  float *jsc = red2 + WG;                     // [JOBS][4]: dot, sumG
  int *sh_edge = (int *)(jsc + JOBS * 4);     // [JOBS][4]: b, y, x, row
  float *at = (float *)(sh_edge + JOBS * 4);  // [JOBS][C][KW][KW] centre windows
  float *gwin = at + JOBS * C * KW * KW;      // [JOBS][KW][KW] window gradients of the current channel

  // (opaque to the optimiser: inside the tail kernel's group loop the lane constants derived from it would otherwise be
  // hoisted out of the loop and held across the whole body -- 256 VGPRs + 31 AGPRs at (25,9), scratch at (49,13); the
  // one-group kernels lose nothing by it)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  In tile order, job groups
  // g and g+1 update overlapping gradient pixels: give every XCD one contiguous range of groups so
  // that those atomics meet in one L2.
  int grp = vb;
  if (p.order && !SSG_DBG(p, 16)) {
    const int ng = (nrows + JOBS - 1) / JOBS, per = (ng + 7) >> 3;
    const int xcd = vb & 7, idx = vb >> 3;
    grp = idx < per ? xcd * per + idx : ng;
  }
  const int job0 = grp * JOBS;
  if (job0 >= nrows) {
    if (p.mode == GRAD_LOSS && tid == 0) {
      p.partials[2 * vb] = 0.f;
      p.partials[2 * vb + 1] = 0.f;
    }
    return -1;   // (every later workgroup index maps behind the last group as well)
  }
  if (tid < JOBS) {
    const int k = job0 + tid;
    const bool v = k < nrows;
    const int row = v ? (p.order ? (p.order[k] & ORDER_MASK) : k) : -1;
    const Edge e = load_edge(p.edges, p.estride, v ? row : 0);
    sh_edge[tid * 4 + 0] = e.b;
    sh_edge[tid * 4 + 1] = e.y;
    sh_edge[tid * 4 + 2] = e.x;
    sh_edge[tid * 4 + 3] = row;
  }
  for (int i = tid; i < G::ZROW + 4; i += WG) zero[i] = 0.f;
  for (int i = tid; i < JOBS * KS * (S - KS); i += WG) gt[(i / (S - KS)) * S + KS + i % (S - KS)] = 0.f;
  if (tid < PADF) smem[tid] = 0.f;
  __syncthreads();

  int jl = tid / LPJ;
  const int m = tid - jl * LPJ;
  const bool lane_on = jl < JOBS;
  if (!lane_on) jl = 0;
  float *tg = gt + jl * CHG;
  const float kfac = 1.f / (p.sigma * (float)(C * KW * KW));
  const float *zrow = zero + HK;
  float l1p = 0.f, klp = 0.f;
  const bool need_grad = p.grad != nullptr;
  float gsc;
  if (p.fix_inline && p.gfix) {   // (loss step on the direct-only path: no launch in front of this one computes the bound)
    const unsigned bb = __float_as_uint(loss_grad_bound(p.sigma, C, KW, p.w_l1, p.w_kl, p.upstream, rows_to_do(p.n_dev, p.n_host), P));
    gsc = grad_fix_scale_of(bb);
    if (vb == 0 && tid == 0) *(unsigned *)(p.gfix + (size_t)p.B * p.C * p.H * p.W) = bb;   // for grad_fix_flush
  } else {
    gsc = grad_fix_scale(p.gfix, (size_t)p.B * p.C * p.H * p.W);
  }

  // Barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its global
  // atomics (a __syncthreads() would drain vmcnt and stall every barrier behind the atomics'
  // L2 round trip -- 73 % of wave time was spent there).
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  {
    const int mo = m;
    const int by = mo / NB, bx = mo - by * NB;
    const int ry0 = BS * by - HK, cx0 = BS * bx - HK;
    const int eb = sh_edge[jl * 4 + 0], ey = sh_edge[jl * 4 + 1], ex = sh_edge[jl * 4 + 2];
    const int n = sh_edge[jl * 4 + 3];
    const bool job_on = lane_on && n >= 0;

    // ---- stage 1: G tile of each job.  Every global load of the row is issued before the first
    // use (EPL independent loads per source; a rolled loop paid one L2 latency per element) ----
    {
      const size_t base = (size_t)(job_on ? n : 0) * P;
      float va[EPL], vg[EPL];
      const bool direct = p.mode == GRAD_D || SSG_DBG(p, 1);
      const float *src_a = direct ? (p.mode == GRAD_D ? p.gin : p.ssg) : p.ssg;
      const float *src_b = p.mode == GRAD_S ? p.gin : p.ssg2;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int e = mo + k * LPJ;
        va[k] = (job_on && e < P) ? src_a[base + e] : 0.f;
      }
      if (!direct) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          const int e = mo + k * LPJ;
          vg[k] = (job_on && e < P) ? src_b[base + e] : 0.f;
        }
        const float invM = 1.f / ((float)nrows * (float)P);
        const float u1 = p.upstream ? p.upstream[0] : 1.f, u2 = p.upstream ? p.upstream[1] : 1.f;
        const float w1m = p.w_l1 * invM * u1, w2m = p.w_kl * invM * u2;
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          const int e = mo + k * LPJ;
          float g = 0.f;
          if (job_on && e < P) g = p.mode == GRAD_S ? vg[k] : criteria_elem(va[k], vg[k], w1m, w2m, l1p, klp);
          vg[k] = g;
          dot = __builtin_fmaf(g, va[k], dot);
        }
        // one barrier pair serves the per-job dot products and the workgroup's criteria partial sums
        // (`red` is free until pass B): lanes [8,16) / [16,24) each sum an eighth of the l1 / kl terms
        red2[tid] = dot;
        red[tid] = l1p;
        red[WG + tid] = klp;
        __syncthreads();
        if (tid < JOBS) {
          float t = 0.f;
          for (int k = 0; k < LPJ; ++k) t += red2[tid * LPJ + k];
          jsc[tid * 4 + 0] = p.generalization ? t : 0.f;
        } else if (tid >= 8 && tid < 24) {
          const int h = (tid - 8) >> 3, part = (tid - 8) & 7;
          float t = 0.f;
          for (int k = 0; k < WG / 8; ++k) t += red[h * WG + part * (WG / 8) + k];
          red[2 * WG + (tid - 8)] = t;
        }
        __syncthreads();
        if (p.mode == GRAD_LOSS && tid == 0) {
          float t1 = 0.f, t2 = 0.f;
          for (int k = 0; k < 8; ++k) {
            t1 += red[2 * WG + k];
            t2 += red[2 * WG + 8 + k];
          }
          p.partials[2 * vb] = t1;
          p.partials[2 * vb + 1] = t2;
        }
        dot = jsc[jl * 4 + 0];
#pragma unroll
        for (int k = 0; k < EPL; ++k) va[k] = -(va[k] * kfac) * (vg[k] - dot);
      }
      // G at the centre offset multiplies (A - B) == 0 exactly (B is the window itself there), but it is the
      // largest entry of the row by orders of magnitude when sigma is small; in the split sums below its two
      // copies would cancel only to fp32 round-off (measured 2.4e-4 of max|grad| at sigma = 0.004).  Dropping
      // it is exact.  The lane's share of sum_p G (without the centre) goes to the per-job reduction.
      float ls = 0.f;
      if (lane_on) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          const int e = mo + k * LPJ;
          if (e < P) {
            const int py = e / KS, px = e - py * KS;
            const float gv = (e == HP * KS + HP) ? 0.f : va[k];
            tg[py * S + px] = gv;
            ls += gv;
          }
        }
      }
      red2[tid] = ls;  // (the dot products were consumed before the barrier above)
    }
    if (p.mode == GRAD_LOSS && SSG_DBG(p, 1) && tid == 0) {  // (profiling ablation without criteria)
      p.partials[2 * vb] = 0.f;
      p.partials[2 * vb + 1] = 0.f;
    }
    if (need_grad) {
    // centre windows A (reflect by index mirroring); loads of all jobs in flight together
    {
      constexpr int APT = (3 * KW * KW + WG - 1) / WG;  // elements per thread per job when C <= 3
      if (C <= 3) {
        float v[JOBS][APT];
#pragma unroll
        for (int j = 0; j < JOBS; ++j) {
          const int b = sh_edge[j * 4 + 0], y = sh_edge[j * 4 + 1], x = sh_edge[j * 4 + 2];
#pragma unroll
          for (int k = 0; k < APT; ++k) {
            const int e = tid + k * WG;
            const int ec = e < C * KW * KW ? e : 0;
            const int c = ec / (KW * KW), r = ec - c * KW * KW, kh = r / KW, kx = r - kh * KW;
            v[j][k] =
                p.img[(((size_t)b * C + c) * H + reflect_idx(y - HK + kh, H)) * W + reflect_idx(x - HK + kx, W)];
          }
        }
#pragma unroll
        for (int j = 0; j < JOBS; ++j)
#pragma unroll
          for (int k = 0; k < APT; ++k) {
            const int e = tid + k * WG;
            if (e < C * KW * KW) at[(j * C) * KW * KW + e] = v[j][k];
          }
      } else {
        for (int j = 0; j < JOBS; ++j) {
          const int b = sh_edge[j * 4 + 0], y = sh_edge[j * 4 + 1], x = sh_edge[j * 4 + 2];
          for (int e = tid; e < C * KW * KW; e += WG) {
            const int c = e / (KW * KW), r = e - c * KW * KW, kh = r / KW, kx = r - kh * KW;
            at[(j * C) * KW * KW + e] =
                p.img[(((size_t)b * C + c) * H + reflect_idx(y - HK + kh, H)) * W + reflect_idx(x - HK + kx, W)];
          }
        }
      }
    }
    __syncthreads();  // G tiles, centre windows and the sum_p G shares are in LDS
    if (tid < JOBS) {
      float t = 0.f;
      for (int k = 0; k < LPJ; ++k) t += red2[tid * LPJ + k];
      jsc[tid * 4 + 1] = t;  // read in pass B's reduction, several barriers downstream
    }

    // window sum of Gz around every owned t (channel independent), streamed like pass A: per
    // patch row the k_w-tap horizontal sums, added to every block row the patch row pairs with
    // (the BS windows of a block share their middle taps: the shared part is summed once and each window
    // adds its own few taps on either side -- 2.3x fewer additions than BS independent k_w-tap sums)
    float box[BS][BS];
    {
      float hrow[PW][BS];  // horizontal k_w-tap sums of the PW patch rows
      float bn[PW];
      load_grow<G>(tg, zrow, ry0, cx0, bn);
#pragma unroll
      for (int r = 0; r < PW; ++r) {
        float bv[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) bv[j] = bn[j];
        if (r + 1 < PW) load_grow<G>(tg, zrow, ry0 + r + 1, cx0, bn);
        shared_window_sums<BS, KW>(bv, hrow[r]);
        pin_row<BS>(hrow[r]);
      }
      // vertical: box[i][j] = sum_{r=i}^{i+KW-1} hrow[r][j], same sharing along the rows
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        float col[PW], out[BS];
#pragma unroll
        for (int r = 0; r < PW; ++r) col[r] = hrow[r][j];
        shared_window_sums<BS, KW>(col, out);
#pragma unroll
        for (int i = 0; i < BS; ++i) box[i][j] = out[i];
      }
    }

    // image offset of tile position (ty,tx) of this job; callers guard ty,tx < KS
    auto img_off = [&](int ty, int tx) -> int {
      return reflect_idx(ey - HP + ty, H) * W + reflect_idx(ex - HP + tx, W);
    };
    // (`fence` is an opaque zero produced by an asm AFTER pass A: it pins the address arithmetic
    // and the loads below that point, so hipcc cannot keep 50 VGPRs of offsets and image values
    // live across the unrolled pass)
    auto load_sv = [&](int c, float (&sv)[BS][BS], int fence) {
      const size_t cb = ((size_t)eb * C + c) * H * W;
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const int ty = BS * by + i, tx = BS * bx + j;
          const bool in = ty < KS && tx < KS;
          sv[i][j] = in ? p.img[cb + fence + img_off(in ? ty : 0, in ? tx : 0)] : 0.f;
        }
    };

    float *gst = red + jl * SL * LPJ;  // job's gradient staging tile [KS][KS], aliases its slice
    static_assert(SL * LPJ >= P, "staging tile must fit the reduction slice");
    static_assert(JOBS <= 8 && WG >= 24 && WG % 8 == 0 && JOBS * SL * LPJ >= 2 * WG + 16,
                  "stage-1 reductions: lanes [0,JOBS) / [8,24) and their scratch in the pass-B slice");

#pragma unroll 1
    for (int c = 0; c < C; ++c) {
      const size_t cbase = ((size_t)eb * C + c) * H * W;
      const float *ac = at + (jl * C + c) * KW * KW;

      // ---- pass A: acc[t] = sum_k' Gz[t+k'] * A[c,-k'] ----
      float acc[BS][BS];
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) acc[i][j] = 0.f;
      if (SSG_DBG(p, 2)) {
      } else if constexpr (KW <= 9) {
        // flipped stencil, one row per patch row: row kh is live during r = kh .. kh + BS - 1 only (see ssg_fwd.hip)
        float af[KW][KW];
        auto load_af = [&](int kh) {
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) af[kh][kx] = ac[(KW - 1 - kh) * KW + (KW - 1 - kx)];
        };
        load_af(0);
        float bn[PW];
        load_grow<G>(tg, zrow, ry0, cx0, bn);
#pragma unroll
        for (int r = 0; r < PW; ++r) {
          float bv[PW];
#pragma unroll
          for (int j = 0; j < PW; ++j) bv[j] = bn[j];
          if (r + 1 < PW) load_grow<G>(tg, zrow, ry0 + r + 1, cx0, bn);
          if (r + 1 < KW) load_af(r + 1);
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const int kh = r - i;
            if (kh < 0 || kh >= KW) continue;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)  // tap-major: BS independent FMAs back to back
#pragma unroll
              for (int j = 0; j < BS; ++j) acc[i][j] = __builtin_fmaf(af[kh][kx], bv[j + kx], acc[i][j]);
          }
          pin_block<BS, BS>(acc);
        }
      } else {
        // large stencils: same fully unrolled row stream, the flipped stencil row is re-read from LDS
        // (broadcast) where it is used instead of living in k_w^2 registers
        float bn[PW];
        load_grow<G>(tg, zrow, ry0, cx0, bn);
#pragma unroll
        for (int r = 0; r < PW; ++r) {
          float bv[PW];
#pragma unroll
          for (int j = 0; j < PW; ++j) bv[j] = bn[j];
          if (r + 1 < PW) load_grow<G>(tg, zrow, ry0 + r + 1, cx0, bn);
#pragma unroll
          for (int i = 0; i < BS; ++i) {
            const int kh = r - i;
            if (kh < 0 || kh >= KW) continue;
            const float *ar = ac + (KW - 1 - kh) * KW;
            float av[KW];
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) av[kx] = ar[KW - 1 - kx];
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
#pragma unroll
              for (int j = 0; j < BS; ++j) acc[i][j] = __builtin_fmaf(av[kx], bv[j + kx], acc[i][j]);
          }
          pin_block<BS, BS>(acc);
        }
      }
      // S[c,t] of the owned block: loaded only now (pass A's register footprint is acc + stencil
      // + two patch rows + box)
      float sv[BS][BS];
      {
        int fz = 0;
        asm volatile("" : "+v"(fz)::"memory");
        load_sv(c, sv, fz);
      }
      // gS[c,t] = -2 (acc - S box)
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) acc[i][j] = -2.f * (acc[i][j] - sv[i][j] * box[i][j]);

      // ---- pass B: P[k'] = sum_t Gz[t+k'] * S[c,t], KHC stencil rows at a time ----
#pragma unroll 1
      for (int ch = 0; ch < (SSG_DBG(p, 4) ? 0 : NCH); ++ch) {
        const int kh0 = ch * KHC;
        float pp[KHC][KW];
#pragma unroll
        for (int a = 0; a < KHC; ++a)
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) pp[a][kx] = 0.f;
        // patch rows r = kh' + i, kh' in [kh0, kh0+KHC), i in [0,BS)
        float bn[PW];
        load_grow<G>(tg, zrow, ry0 + kh0, cx0, bn);
#pragma unroll
        for (int rr = 0; rr < KHC + BS - 1; ++rr) {
          float bv[PW];
#pragma unroll
          for (int j = 0; j < PW; ++j) bv[j] = bn[j];
          if (rr + 1 < KHC + BS - 1) load_grow<G>(tg, zrow, ry0 + kh0 + rr + 1, cx0, bn);
#pragma unroll
          for (int a = 0; a < KHC; ++a) {
            const int i = rr - a;  // block row paired with stencil row kh0+a on this patch row
            if (i < 0 || i >= BS) continue;
#pragma unroll
            for (int j = 0; j < BS; ++j)
#pragma unroll
              for (int kx = 0; kx < KW; ++kx) pp[a][kx] = __builtin_fmaf(bv[j + kx], sv[i][j], pp[a][kx]);
          }
          pin_block<KHC, KW>(pp);
        }
        // reduce the slice across the job's lanes (fixed order)
        if (lane_on) {
#pragma unroll
          for (int a = 0; a < KHC; ++a)
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) red[(jl * SL + a * KW + kx) * LPJ + m] = pp[a][kx];
        }
        lds_barrier();
        if (lane_on)
          for (int o = m; o < SL; o += LPJ) {
            const int khp = kh0 + o / KW;
            if (khp < KW) {
              const float *rp = red + (jl * SL + o) * LPJ;
              float t = 0.f;
              for (int k = 0; k < LPJ; ++k) t += rp[k];
              const int kh = KW - 1 - khp, kx = KW - 1 - (o % KW);  // k = -k'
              gwin[jl * KW * KW + kh * KW + kx] = 2.f * (ac[kh * KW + kx] * jsc[jl * 4 + 1] - t);  // unique owner
            }
          }
        lds_barrier();
      }

      // ---- stage the job's gradient tile of this channel in LDS, merge the window part ----
      if (lane_on) {
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
          for (int j = 0; j < BS; ++j) {
            const int ty = BS * by + i, tx = BS * bx + j;
            if (ty < KS && tx < KS) gst[ty * KS + tx] = acc[i][j];
          }
      }
      lds_barrier();
      if (lane_on)
        for (int o = m; o < KW * KW; o += LPJ) {
          const int kh = o / KW, kx = o - kh * KW;
          gst[(HP - HK + kh) * KS + (HP - HK + kx)] += gwin[jl * KW * KW + o];  // unique owner per (kh,kx)
        }
      lds_barrier();
      // Merge window: when the workgroup's jobs are one image's edge pixels within 8 rows x 16
      // columns (always, up to tile seams, in tile order), their tiles are summed in LDS first.
      int my0 = 1 << 30, mx0 = 1 << 30, my1 = -1, mx1 = -1, mb0 = -1;
      bool merge = p.order != nullptr;
#pragma unroll
      for (int j = 0; j < JOBS; ++j) {
        if (sh_edge[j * 4 + 3] >= 0) {
          const int b = sh_edge[j * 4 + 0], y = sh_edge[j * 4 + 1], x = sh_edge[j * 4 + 2];
          if (mb0 < 0) mb0 = b;
          merge = merge && b == mb0;
          my0 = y < my0 ? y : my0;
          my1 = y > my1 ? y : my1;
          mx0 = x < mx0 ? x : mx0;
          mx1 = x > mx1 ? x : mx1;
        }
      }
      merge = merge && (my1 - my0) <= MH - KS && (mx1 - mx0) <= MW - KS;
      if (merge) {
        // ---- gather-merge: every lane owns pixels of the jobs' common window and sums the
        // staged tiles that cover them (no LDS atomics, no per-job rounds), then issues ONE
        // fp32 atomic per touched pixel, row-contiguous ----
        if (!SSG_DBG(p, 8)) {
          const size_t cb0 = ((size_t)mb0 * C + c) * H * W;
          const int wh = my1 - my0 + KS, ww = mx1 - mx0 + KS;
          int dy[JOBS], dx[JOBS];
#pragma unroll
          for (int j = 0; j < JOBS; ++j) {
            const bool on = sh_edge[j * 4 + 3] >= 0;
            dy[j] = on ? sh_edge[j * 4 + 1] - my0 : -(1 << 20);
            dx[j] = on ? sh_edge[j * 4 + 2] - mx0 : -(1 << 20);
          }
          const float inv_ww = 1.f / (float)ww;  // i / ww for i < MH * MW, exact in fp32
          for (int i = tid; i < wh * ww; i += WG) {
            const int ry = (int)(((float)i + 0.5f) * inv_ww), rx = i - ry * ww;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < JOBS; ++j) {
              const int ty = ry - dy[j], tx = rx - dx[j];
              const bool in = (unsigned)ty < (unsigned)KS && (unsigned)tx < (unsigned)KS;
              const float g = red[j * SL * LPJ + (in ? ty * KS + tx : 0)];
              v += in ? g : 0.f;
            }
            if (v != 0.f && !SSG_DBG(p, 32))   // (profiling ablation 32: the gather without its atomics)
              grad_add(p.grad, p.gfix, cb0 + (size_t)reflect_idx(my0 - HP + ry, H) * W + reflect_idx(mx0 - HP + rx, W), v, gsc);
            if (SSG_DBG(p, 32) && v == 123456.f) red2[0] = v;   // (keeps the sums alive)
          }
        }
        lds_barrier();
      } else {
        // ---- flush per job: row-contiguous fp32 atomics, nothing waits for them ----
        if (job_on && !SSG_DBG(p, 8)) {
#pragma unroll 5
          for (int k = 0; k < EPL; ++k) {
            const int e = m + k * LPJ;
            if (e < P) {
              const int ty = e / KS, tx = e - ty * KS;
              grad_add(p.grad, p.gfix, cbase + img_off(ty, tx), gst[e], gsc);
            }
          }
        }
        lds_barrier();  // staging tile is the next channel's reduction slice
      }
    }
    }  // need_grad
  }

  return 1;
}

template <class G, int KHC>
__global__ __launch_bounds__(G::WG) void ssg_bwd_tiled(BwdParams p) {
  bwd_tiled_group<G, KHC>(p, (int)blockIdx.x);
}
// The workgroup indices from `first` on, walked by a small grid (see ssg_fwd_tiled_tail: the tail of a launch whose bound on
// the rows is far above what the main launch covers with one workgroup per group).
template <class G, int KHC>
__global__ __launch_bounds__(G::WG) void ssg_bwd_tiled_tail(BwdParams p, int first) {
#pragma unroll 1
  for (int vb = first + (int)blockIdx.x;; vb += (int)gridDim.x) {
    if (bwd_tiled_group<G, KHC>(p, vb) < 0) break;
    __syncthreads();   // the group's LDS is rewritten by the next one
  }
}

// Any odd (ks, kw): one 256-lane workgroup per edge pixel.
__global__ __launch_bounds__(256) void ssg_bwd_generic(BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ks = p.ks, kw = p.kw, hp = ks / 2, hk = kw / 2, P = ks * ks, K2 = kw * kw;
  const int C = p.C, H = p.H, W = p.W, tid = threadIdx.x;
  float *gt = smem;          // [P]
  float *tile = gt + P;      // [C][P]
  float *red = tile + C * P; // [256]
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const int n = blockIdx.x;
  if (n >= nrows) {
    if (p.mode == GRAD_LOSS && tid == 0) {
      p.partials[2 * n] = 0.f;
      p.partials[2 * n + 1] = 0.f;
    }
    return;
  }
  const int row = p.order ? (p.order[n] & ORDER_MASK) : n;
  const Edge e = load_edge(p.edges, p.estride, row);
  const size_t base = (size_t)row * P;
  const float kfac = 1.f / (p.sigma * (float)(C * K2));
  float l1p = 0.f, klp = 0.f;
  if (p.mode == GRAD_D) {
    for (int i = tid; i < P; i += 256) gt[i] = p.gin[base + i];
  } else {
    const float invM = 1.f / ((float)nrows * (float)P);
    const float u1 = p.upstream ? p.upstream[0] : 1.f, u2 = p.upstream ? p.upstream[1] : 1.f;
    const float w1m = p.w_l1 * invM * u1, w2m = p.w_kl * invM * u2;
    float dot = 0.f;
    for (int i = tid; i < P; i += 256) {
      const float s = p.ssg[base + i];
      const float g = p.mode == GRAD_S ? p.gin[base + i] : criteria_elem(s, p.ssg2[base + i], w1m, w2m, l1p, klp);
      dot = __builtin_fmaf(g, s, dot);
      gt[i] = g;
    }
    red[tid] = dot;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < 256; ++k) t += red[k];
    if (!p.generalization) t = 0.f;
    __syncthreads();
    for (int i = tid; i < P; i += 256) gt[i] = -(p.ssg[base + i] * kfac) * (gt[i] - t);
    if (p.mode == GRAD_LOSS) {
      red[tid] = l1p;
      __syncthreads();
      float t1 = 0.f, t2 = 0.f;
      if (tid == 0)
        for (int k = 0; k < 256; ++k) t1 += red[k];
      __syncthreads();
      red[tid] = klp;
      __syncthreads();
      if (tid == 0) {
        for (int k = 0; k < 256; ++k) t2 += red[k];
        p.partials[2 * n] = t1;
        p.partials[2 * n + 1] = t2;
      }
      if (p.grad == nullptr) return;
    }
  }
  __syncthreads();
  if (tid == 0) gt[hp * ks + hp] = 0.f;  // exact: the centre offset has A - B == 0 (see the tiled kernel)
  const size_t ibase = (size_t)e.b * C * H * W;
  float gsc;
  if (p.fix_inline && p.gfix) {
    const unsigned bb = __float_as_uint(loss_grad_bound(p.sigma, C, p.kw, p.w_l1, p.w_kl, p.upstream, rows_to_do(p.n_dev, p.n_host), P));
    gsc = grad_fix_scale_of(bb);
    if (blockIdx.x == 0 && tid == 0) *(unsigned *)(p.gfix + (size_t)p.B * p.C * p.H * p.W) = bb;
  } else {
    gsc = grad_fix_scale(p.gfix, (size_t)p.B * p.C * p.H * p.W);
  }
  for (int i = tid; i < C * P; i += 256) {
    const int c = i / P, r = i - c * P, ry = r / ks, rx = r - ry * ks;
    tile[i] = p.img[ibase + ((size_t)c * H + reflect_idx(e.y - hp + ry, H)) * W + reflect_idx(e.x - hp + rx, W)];
  }
  __syncthreads();
  float ls = 0.f;
  for (int i = tid; i < P; i += 256) ls += gt[i];
  red[tid] = ls;
  __syncthreads();
  float sumG = 0.f;
  for (int k = 0; k < 256; ++k) sumG += red[k];
  // tile positions: gS[c,t] = -2 sum_k Gz[t-k] (A[c,k] - S[c,t])
  for (int i = tid; i < C * P; i += 256) {
    const int c = i / P, r = i - c * P, ty = r / ks, tx = r - ty * ks;
    const float st = tile[i];
    float acc = 0.f;
    for (int kh = -hk; kh <= hk; ++kh)
      for (int kx = -hk; kx <= hk; ++kx) {
        const int py = ty - kh, px = tx - kx;
        if ((unsigned)py < (unsigned)ks && (unsigned)px < (unsigned)ks)
          acc = __builtin_fmaf(gt[py * ks + px], tile[(c * ks + hp + kh) * ks + hp + kx] - st, acc);
      }
    grad_add(p.grad, p.gfix, ibase + ((size_t)c * H + reflect_idx(e.y - hp + ty, H)) * W + reflect_idx(e.x - hp + tx, W),
             -2.f * acc, gsc);
  }
  // window positions: gA[c,k] = 2 ( A sum G - sum_p G[p] Sz[c,p+k] )
  for (int i = tid; i < C * K2; i += 256) {
    const int c = i / K2, r = i - c * K2, kh = r / kw - hk, kx = r - (r / kw) * kw - hk;
    const float a = tile[(c * ks + hp + kh) * ks + hp + kx];
    float acc = 0.f;
    for (int py = 0; py < ks; ++py) {
      const int yy = py + kh;
      if ((unsigned)yy >= (unsigned)ks) continue;
      for (int px = 0; px < ks; ++px) {
        const int xx = px + kx;
        if ((unsigned)xx < (unsigned)ks) acc = __builtin_fmaf(gt[py * ks + px], tile[(c * ks + yy) * ks + xx], acc);
      }
    }
    grad_add(p.grad, p.gfix, ibase + ((size_t)c * H + reflect_idx(e.y + kh, H)) * W + reflect_idx(e.x + kx, W),
             2.f * (a * sumG - acc), gsc);
  }
}

// loss_out[0] = w_l1 * sum|a-b| / M, loss_out[1] = w_kl * sum t'(log t' - log s') / M.
// One workgroup, fp64, fixed summation order: 1,024 lane-strided partial sums, then a binary tree over them -- written for
// NT threads that each play 1024 / NT of those lanes, so that the standalone kernel (NT = 1024) and the tail of
// grad_fix_flush (NT = 256, round 5: one launch less at the end of every deterministic step) give the same bits.
template <int NT>
__device__ __forceinline__ void loss_finalize_body(const LossFinalize &f, double *s1, double *s2) {
  // (sets of slots: only the live prefix of every set, see LossFinalize::set_size -- the dead slots hold zeros, so the
  //  sums are the same numbers in the same order of lanes either way)
  const int nsets = f.set_size > 0 ? f.nparts / f.set_size : 1;
  const int set_size = f.set_size > 0 ? f.set_size : f.nparts;
  int live = set_size;
  if (f.set_size > 0) {
    const int lr = (rows_to_do(f.n_dev, f.n_host) + 3) / 4;
    live = lr < set_size ? lr : set_size;
  }
  for (int vt = threadIdx.x; vt < 1024; vt += NT) {
    double a = 0, b = 0;
    for (int sset = 0; sset < nsets; ++sset) {
      const float2 *pp = (const float2 *)f.partials + (size_t)sset * set_size;
      int i = vt;
      for (; i + 3 * 1024 < live; i += 4 * 1024) {  // four independent loads in flight
        const float2 v0 = pp[i], v1 = pp[i + 1024], v2 = pp[i + 2048], v3 = pp[i + 3072];
        a += (double)v0.x + (double)v1.x + (double)v2.x + (double)v3.x;
        b += (double)v0.y + (double)v1.y + (double)v2.y + (double)v3.y;
      }
      for (; i < live; i += 1024) {
        const float2 v = pp[i];
        a += (double)v.x;
        b += (double)v.y;
      }
    }
    s1[vt] = a;
    s2[vt] = b;
  }
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    for (int vt = threadIdx.x; vt < o; vt += NT) {
      s1[vt] += s1[vt + o];
      s2[vt] += s2[vt + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int nrows = rows_to_do(f.n_dev, f.n_host);
    const double M = (double)nrows * (double)f.P;
    f.loss_out[0] = nrows > 0 ? (float)((double)f.w_l1 * s1[0] / M) : 0.f;
    f.loss_out[1] = nrows > 0 ? (float)((double)f.w_kl * s2[0] / M) : 0.f;
    // the fused entry points: a step that found more edge pixels than the caller's capacity has used the first
    // `capacity` of them only -- its losses are NaN, so that a truncated step cannot pass for a complete one
    if (f.nan_on_overflow && f.n_dev && *f.n_dev > f.n_host) f.loss_out[0] = f.loss_out[1] = __builtin_nanf("");
  }
}
__global__ __launch_bounds__(1024) void ssg_loss_finalize(LossFinalize f) {
  __shared__ double s1[1024], s2[1024];
  loss_finalize_body<1024>(f, s1, s2);
}

// deterministic mode: grad += fixed-point sums (one rounding per pixel); assign != 0: grad = the sums (ssg_loss_step:
// the gradient is an output, nobody has to clear it first)
// FIN: the step's loss finalize rides in the LAST workgroup (its partial sums were complete before this launch).
template <bool FIN>
__device__ __forceinline__ void grad_fix_flush_body(const long long *gfix, float *grad, size_t n, int assign);
template <bool FIN>
__global__ __launch_bounds__(256) void grad_fix_flush(const long long *gfix, float *grad, size_t n, int assign, LossFinalize f) {
  grad_fix_flush_body<FIN>(gfix, grad, n, assign);
  if constexpr (FIN) {
    if (blockIdx.x == gridDim.x - 1) {
      __shared__ double s1[1024], s2[1024];
      loss_finalize_body<256>(f, s1, s2);
    }
  }
}
template <bool FIN>
__device__ __forceinline__ void grad_fix_flush_body(const long long *gfix, float *grad, size_t n, int assign) {
  const double inv = 1.0 / (double)grad_fix_scale(gfix, n);
  if ((((size_t)gfix & 15) | ((size_t)grad & 7)) == 0) {
    // two sums per lane and iteration: one 16-byte load, one 8-byte store (the pass is 37 MB of traffic at C2, at the end
    // of every step's critical path)
    const size_t n2 = n / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
      const longlong2 v = ((const longlong2 *)gfix)[i];
      float2 *g = (float2 *)grad + i;
      if (assign) {
        *g = make_float2((float)((double)v.x * inv), (float)((double)v.y * inv));
      } else if (v.x | v.y) {
        float2 o = *g;
        if (v.x) o.x += (float)((double)v.x * inv);
        if (v.y) o.y += (float)((double)v.y * inv);
        *g = o;
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const long long v = gfix[n - 1];
      if (assign) grad[n - 1] = (float)((double)v * inv);
      else if (v) grad[n - 1] += (float)((double)v * inv);
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const long long v = gfix[i];
    if (assign) grad[i] = (float)((double)v * inv);
    else if (v) grad[i] += (float)((double)v * inv);
  }
}

// deterministic mode on the direct-only path (no ssg_grad_rows pass): an upper bound of |G| before the backward
// kernel runs.  GRAD_D: max|gin|; GRAD_S: G = -(s kfac)(g - sum g s) with 0 <= s <= 1, sum s <= 1 -> 2 kfac max|g|;
// GRAD_LOSS: |s g| <= w1m + w2m (s, t <= 1) -> 4 kfac (w1m + w2m).
__global__ __launch_bounds__(256) void grad_fix_bound_kernel(BwdParams p, size_t n_fix) {
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const int P = p.ks * p.ks;
  const float kfac = 1.f / (p.sigma * (float)(p.C * p.kw * p.kw));
  if (p.mode == GRAD_LOSS) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      grad_fix_bound(p.gfix, n_fix, loss_grad_bound(p.sigma, p.C, p.kw, p.w_l1, p.w_kl, p.upstream, nrows, P));
    }
    return;
  }
  const size_t n = (size_t)nrows * P;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(p.gin[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {   // one atomic per workgroup, at most 512 of them
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    grad_fix_bound(p.gfix, n_fix, p.mode == GRAD_S ? 2.f * kfac * m : m);
  }
}

// max over the per-workgroup maxima of ssg_grad_rows -> the bound word (one workgroup)
__global__ __launch_bounds__(1024) void grad_fix_reduce(const float *part, int n, long long *gfix, size_t n_fix) {
  __shared__ float sm[16];
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) m = fmaxf(m, sm[k]);
    *(unsigned *)(gfix + n_fix) = __float_as_uint(m);
  }
}

int launch_grad_fix_reduce(const float *part, int n, long long *gfix, size_t n_fix, hipStream_t st) {
  hipLaunchKernelGGL(grad_fix_reduce, dim3(1), dim3(1024), 0, st, part, n, gfix, n_fix);
  return (int)hipGetLastError();
}

int launch_grad_fix_bound(const BwdParams &p, hipStream_t st) {
  const size_t n = (size_t)p.n_host * p.ks * p.ks;
  const unsigned grid = p.mode == GRAD_LOSS ? 1u : (unsigned)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
  if (grid == 0) return 0;
  hipLaunchKernelGGL(grad_fix_bound_kernel, dim3(grid), dim3(256), 0, st, p, (size_t)p.B * p.C * p.H * p.W);
  return (int)hipGetLastError();
}

int launch_grad_fix_flush(long long *gfix, float *grad, size_t n, int assign, hipStream_t st, const LossFinalize *fin) {
  if (!n) return fin ? -1 : 0;
  const size_t units = (n + 1) / 2;   // (two sums per lane)
  const unsigned grid = (unsigned)((units + 255) / 256 < 16384 ? (units + 255) / 256 : 16384);
  if (fin) hipLaunchKernelGGL(grad_fix_flush<true>, dim3(grid), dim3(256), 0, st, gfix, grad, n, assign, *fin);
  else hipLaunchKernelGGL(grad_fix_flush<false>, dim3(grid), dim3(256), 0, st, gfix, grad, n, assign, LossFinalize{});
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------ host ----
template <class G, int KHC>
static size_t bwd_lds_bytes(int C) {
  constexpr int PADF = (G::HK + 3) & ~3;
  return sizeof(float) * (size_t)(PADF + G::JOBS * G::KS * G::GS + ((G::ZROW + 3) & ~3) + G::JOBS * KHC * G::KW * G::LPJ +
                                  G::WG + G::JOBS * 4 + G::JOBS * 4 + G::JOBS * (C + 1) * G::KW * G::KW + 8);
}

unsigned bwd_grid(const BwdParams &p);
constexpr unsigned BWD_MAIN_GROUPS = 20480, BWD_TAIL_GRID = 512;   // (multiples of 8: the XCD-contiguous group mapping)

template <class G, int KHC>
static int launch_bwd_tiled(const BwdParams &p, hipStream_t st) {
  const size_t lds = bwd_lds_bytes<G, KHC>(p.C);
  if (lds > 160 * 1024) return -2;
  static std::atomic<unsigned long long> lds_set{0};
  if (const int rc = ensure_dynamic_lds(ssg_bwd_tiled<G, KHC>, 160 * 1024, lds_set)) return rc;
  const unsigned grid = bwd_grid(p);
  if (p.n_host == 0) return 0;
  // (a bound on the rows far above BWD_MAIN_GROUPS x JOBS: one workgroup per group up to there, the looping tail behind --
  //  not where the workgroups' criteria sums have a slot each, GRAD_LOSS on the direct-only path)
  unsigned main_groups = BWD_MAIN_GROUPS;   // (with a hint of the rows to expect: 1.25 x that, see launch_fwd_tiled)
  if (p.rows_hint > 0) {
    const long want = ((long)p.rows_hint * 5 / 4 + G::JOBS - 1) / G::JOBS + 32;
    const long capped = want < 512 ? 512 : (want > (long)BWD_MAIN_GROUPS ? (long)BWD_MAIN_GROUPS : want);
    main_groups = (unsigned)((capped + 7) / 8 * 8);
  }
  if ((grid > main_groups + 4096 || grid > BWD_MAIN_GROUPS) && !p.partials) {   // (a tail launch costs ~5 us: see launch_fwd_tiled)
    static std::atomic<unsigned long long> lds_set_tail{0};
    if (const int rc = ensure_dynamic_lds(ssg_bwd_tiled_tail<G, KHC>, 160 * 1024, lds_set_tail)) return rc;
    hipLaunchKernelGGL((ssg_bwd_tiled<G, KHC>), dim3(main_groups), dim3(G::WG), lds, st, p);
    hipLaunchKernelGGL((ssg_bwd_tiled_tail<G, KHC>), dim3(BWD_TAIL_GRID), dim3(G::WG), lds, st, p, (int)main_groups);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL((ssg_bwd_tiled<G, KHC>), dim3(grid), dim3(G::WG), lds, st, p);
  return (int)hipGetLastError();
}

// Number of workgroups (= rows of `partials`) launch_bwd will use.
// Tiled variants: a multiple of 8 workgroups, so that the XCD-contiguous group mapping covers every group.
unsigned bwd_grid(const BwdParams &p) {
  auto tiled = [&](int jobs) { return (unsigned)(((p.n_host + jobs - 1) / jobs + 7) / 8 * 8); };
  if (p.ks == 25 && p.kw == 9) return tiled(Geo<25, 9, 5, 128>::JOBS);
  if (p.ks == 11 && p.kw == 5) return tiled(Geo<11, 5, 4, 64>::JOBS);
  if (p.ks == 49 && p.kw == 13 && bwd_lds_bytes<Geo<49, 13, 7, 256>, 4>(p.C) <= 160 * 1024)
    return tiled(Geo<49, 13, 7, 256>::JOBS);
  return (unsigned)p.n_host;
}

// Upper bound on bwd_grid() for scratch sizing.
size_t bwd_max_partials(int B, int H, int W, int n_rows) {
  (void)B;
  (void)H;
  (void)W;
  return (size_t)(n_rows > 0 ? n_rows : 1) + 8;
}

int launch_bwd(const BwdParams &p, hipStream_t st) {
  if (p.ks == 25 && p.kw == 9) return launch_bwd_tiled<Geo<25, 9, 5, 128>, 3>(p, st);
  if (p.ks == 11 && p.kw == 5) return launch_bwd_tiled<Geo<11, 5, 4, 64>, 5>(p, st);
  if (p.ks == 49 && p.kw == 13 && bwd_lds_bytes<Geo<49, 13, 7, 256>, 4>(p.C) <= 160 * 1024)
    return launch_bwd_tiled<Geo<49, 13, 7, 256>, 4>(p, st);
  const size_t lds = sizeof(float) * ((size_t)(p.C + 1) * p.ks * p.ks + 256);
  if (lds > 160 * 1024) return -2;
  static std::atomic<unsigned long long> lds_set{0};
  if (const int rc = ensure_dynamic_lds(ssg_bwd_generic, 160 * 1024, lds_set)) return rc;
  if (p.n_host == 0) return 0;
  hipLaunchKernelGGL(ssg_bwd_generic, dim3((unsigned)p.n_host), dim3(256), lds, st, p);
  return (int)hipGetLastError();
}

int launch_loss_finalize(const float *partials, int nparts, const int *n_dev, int n_host, int P, float w_l1,
                         float w_kl, float *loss_out, int nan_on_overflow, hipStream_t st, int set_size) {
  hipLaunchKernelGGL(ssg_loss_finalize, dim3(1), dim3(1024), 0, st,
                     LossFinalize{partials, nparts, n_dev, n_host, P, w_l1, w_kl, loss_out, nan_on_overflow, set_size});
  return (int)hipGetLastError();
}

const char *bwd_kernel_name(int ks, int kw) {
  if (ks == 25 && kw == 9) return "ssg_bwd_tiled<Geo<25,9,5,128>,3>";
  if (ks == 11 && kw == 5) return "ssg_bwd_tiled<Geo<11,5,4,64>,5>";
  if (ks == 49 && kw == 13) return "ssg_bwd_tiled<Geo<49,13,7,256>,4>";
  return "ssg_bwd_generic";
}

}  // namespace ssg
