// Forward SSG kernels for gfx950 (wave64, LDS-tiled, no MFMA).
//
// ssg_fwd_tiled<Geo>: compile-time (k_s, k_w, block) geometry.  Each job's
//   C x k_s x k_s search tile is staged once in LDS from the UNPADDED image
//   (reflect padding = index mirroring during the fill).  A lane owns a BS x BS
//   block of search offsets, keeps the k_w x k_w centre window of the current
//   channel in registers, and streams the (BS+k_w-1)^2 patch its block needs
//   from LDS one row at a time: 13 LDS dwords feed up to 5*5*9 (sub, fma)
//   pairs.  The reference's "B = 0 outside the search area" rule
//   (similarity.cu:43-47 == F.unfold zero padding, loss_util.py:208) is an
//   all-zero LDS row for rows and a per-lane column predicate for columns.
//   Epilogue (loss_util.py:224-227): q = D/(C k_w^2), e = exp(-q/sigma),
//   s = e / (sum_p e + eps); the row sum runs through LDS in a fixed order,
//   the SSG row is staged in LDS and written with coalesced dword stores.
//
// ssg_fwd_generic: any odd (k_s, k_w) at run time, one workgroup per job, one
//   search offset per lane-iteration.  Correct, not fast; keeps unusual sizes on
//   the GPU (there is no CPU fallback anywhere in the product path).
#include "ssg_common.hpp"

namespace ssg {

// One patch row (PW floats) of a lane's block from the LDS tile: rows outside the
// search area read the all-zero row, columns outside it are predicated to 0.
template <class G>
__device__ __forceinline__ void load_row(const float *tc, int rs, const float *zrow, int ry, int cx0,
                                         const bool (&colv)[G::PW], float (&out)[G::PW]) {
  const float *rowp = ((unsigned)ry < (unsigned)G::KS) ? (tc + ry * rs) : zrow;
#pragma unroll
  for (int j = 0; j < G::PW; ++j) {
    const float v = rowp[cx0 + j];
    out[j] = colv[j] ? v : 0.f;
  }
}

// MERGED = false: every job stages its own k_s x k_s search tile in LDS, ONE CHANNEL AT A TIME (the channel loop below
//                 refills the tiles: 13 KB instead of 40 KB per workgroup at k_s 25, i.e. the registers, not the LDS,
//                 bound the waves per SIMD -- measured on this variant: 1.2 waves per SIMD on average and 32 % VALU
//                 issue with all channels resident).
// MERGED = true : the workgroup's jobs are edge pixels of ONE image within 8 rows x 16 columns
//                 (the usual case in the tile-major job order): their search areas are read from
//                 one shared LDS region of at most 32 x 40 pixels per channel -- 2.4x fewer fill
//                 loads and 17 KB instead of 40 KB of LDS per workgroup (3 instead of 2 waves per
//                 SIMD).  Both variants are launched over the same job groups; a group runs in the
//                 variant its geometry selects and exits at once in the other.
// Small calls -- fewer jobs (rows x images) than fill the chip once at 5 x 5 offset blocks: Bernoulli-sparse and strided
// masks, one image of the reference's per-image loop -- run Geo<25,9,3,256>: 3 x 3 blocks, 81 lanes per job, 3 jobs per
// 256-lane workgroup, every group in tile order.  17 % more lane-operations per job (121 LDS dwords feed 9 x 81 pairs
// instead of 169 feeding 25 x 81) on 3.2 times the lanes: measured 0.105 -> 0.066 ms for 2 x 2,627 jobs (Bernoulli 1 %
// of 4 x 256 x 256), neutral at 20 k jobs, slower above.
constexpr int SMALL_CALL_JOBS = 8192;
__device__ __forceinline__ bool small_call(int njobs) { return njobs <= SMALL_CALL_JOBS; }

// returns 1: the group ran here (its LDS is in use: barrier before the next one), 0: not this variant's group, -1: the
// group lies behind the last job (so does every later one: a workgroup's loop ends)
template <class G, bool MERGED>
__device__ __forceinline__ int fwd_tiled_group(const FwdParams &p, int grp) {
  constexpr int KS = G::KS, KW = G::KW, BS = G::BS, WG = G::WG;
  constexpr int HP = G::HP, HK = G::HK, P = G::P, NB = G::NB, LPJ = G::LPJ;
  constexpr int JOBS = G::JOBS, PW = G::PW, S = G::S, CH = G::CH;
  constexpr int PADF = (HK + 3) & ~3;
  constexpr int MH = KS + MERGE_ROWS - 1, MW = KS + MERGE_COLS - 1, MS = MW + 1;  // merged region: rows, cols, row stride

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = p.C, H = p.H, W = p.W;
  float *tiles = smem + PADF;                                      // [JOBS][C][KS][S] or [C][MH][MS]
  // (merged: the shared region is reused to stage the JOBS output rows, whichever is larger)
  const int merged_floats = (C * MH * MS > JOBS * P ? C * MH * MS : JOBS * P + 1) & ~1;
  float *zero = tiles + (MERGED ? merged_floats : JOBS * CH);      // ZROW zeros (also absorbs tail over-reads)
  double *red = (double *)(zero + ((G::ZROW + 3) & ~3));           // [WG] row-sum scratch (8-byte aligned)
  int *sh_edge = (int *)(red + WG);                                // [JOBS][6]: b, y, x, row, which, pad

  int tid_ = threadIdx.x;
  // (inside a group loop -- the k_s = 49 kernel's, and every size's tail kernel: opaque to the optimiser, which would
  // otherwise hoist the lane constants derived from it out of the loop and hold them across the whole body)
  asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  // job numbering: row order -> q = row * nimg + image; tile order -> image-major, every image's
  // jobs padded to a multiple of JOBS so that groups coincide with the ORDER_GROUP groups
  const int npad = p.order ? (nrows + JOBS - 1) / JOBS * JOBS : nrows;
  const int njobs = npad * p.nimg;
  const int job0 = grp * JOBS;
  if (job0 >= njobs) return -1;
  int which0 = 0, k0 = 0;
  bool mergeable = false;
  if (p.order) {
    static_assert(!MERGED || JOBS == ORDER_GROUP, "order flags are computed for groups of ORDER_GROUP jobs");
    which0 = job0 / npad;
    k0 = job0 - which0 * npad;
    if (k0 >= nrows) return 0;                           // padding-only group
    mergeable = (p.order[k0] & ORDER_FLAG) != 0;         // one wave-uniform load decides the variant
  }
  // p.small: 0 = this launch is the only set of variants; 1 = this variant takes every group whatever its flag (the
  // small-call variant launched alone); 2 = regular and small-call variants were both launched and the device-side job
  // count picks (small_call() below)
  constexpr bool SMALLV = G::BS < 5 && G::KS == 25;
  if (p.small == 2 && small_call(nrows * p.nimg) != SMALLV) return -1;   // (the whole launch is the other class's)
  if (!(SMALLV && p.small != 0) && mergeable != MERGED) return 0;

  if (tid < JOBS) {
    int row = -1, which = 0;
    if (p.order) {
      which = which0;
      if (k0 + tid < nrows) row = p.order[k0 + tid] & ORDER_MASK;
    } else if (job0 + tid < njobs) {
      row = (job0 + tid) / p.nimg;
      which = (job0 + tid) - row * p.nimg;
    }
    const Edge e = load_edge(p.edges, p.estride, row < 0 ? 0 : row);
    sh_edge[tid * 6 + 0] = e.b;
    sh_edge[tid * 6 + 1] = e.y;
    sh_edge[tid * 6 + 2] = e.x;
    sh_edge[tid * 6 + 3] = row;
    sh_edge[tid * 6 + 4] = which;
  }
  for (int i = tid; i < G::ZROW + 4; i += WG) zero[i] = 0.f;
  if (tid < PADF) smem[tid] = 0.f;
  __syncthreads();

  // common window of the group's jobs (uniform across the workgroup; merged variant only)
  int my0 = 1 << 30, mx0 = 1 << 30, my1 = -1, mx1 = -1, mb0 = 0, mw0 = 0;
  if constexpr (MERGED) {
#pragma unroll
    for (int j = 0; j < JOBS; ++j)
      if (sh_edge[j * 6 + 3] >= 0) {
        const int y = sh_edge[j * 6 + 1], x = sh_edge[j * 6 + 2];
        mb0 = sh_edge[j * 6 + 0];
        mw0 = sh_edge[j * 6 + 4];
        my0 = y < my0 ? y : my0;
        my1 = y > my1 ? y : my1;
        mx0 = x < mx0 ? x : mx0;
        mx1 = x > mx1 ? x : mx1;
      }
  }

  if constexpr (MERGED) {
    // ---- fill the shared region: rows of up to 40 contiguous pixels, reflect by mirroring ----
    const int wh = my1 - my0 + KS, ww = mx1 - mx0 + KS;
    const float *src = p.img[mw0] + (size_t)mb0 * C * H * W;
    // 8 lanes per region row (5 columns each, stride 8), WG/8 rows per pass, two passes in flight
    constexpr int LPR = 8, CPL = (MW + LPR - 1) / LPR, RPP = WG / LPR;
    const int rows = SSG_DBG(p, 1) ? 0 : C * wh;
    const int lx = tid % LPR, lr = tid / LPR;
    for (int r0 = 0; r0 < rows; r0 += 2 * RPP) {
      float v[2][CPL];
      int d[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int R = r0 + h * RPP + lr;
        const bool on = R < rows;
        const int Rc = on ? R : 0;
        const int c = Rc / wh, ry = Rc - c * wh;
        const float *srow = src + ((size_t)c * H + reflect_idx(my0 - HP + ry, H)) * W;
        d[h] = on ? (c * MH + ry) * MS : -1;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int rx = lx + k * LPR;
          v[h][k] = srow[reflect_idx(mx0 - HP + (rx < ww ? rx : 0), W)];
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int rx = lx + k * LPR;
          if (d[h] >= 0 && rx < ww) tiles[d[h] + rx] = v[h][k];
        }
    }
  }
  // ---- single variant: the jobs' k_s x k_s tiles of ONE channel, reflect by index mirroring.  All global loads (JOBS x
  // EPT per thread) are issued before the first LDS store (a plain loop pays the full L2 latency per element) ----
  // Thread -> tile elements.  With whole tile rows per pass (thread = (row in pass, column): WG / k_s rows per pass) the
  // column's mirrored x is one value per job and a row's mirrored y one per (job, pass) -- no division and a third of the
  // index arithmetic of the linear map e = tid + k WG (20 VALU per load there: a fifth of the single variant's
  // instructions at (25,9)); taken when it needs no more passes than the linear map.
  constexpr int RPT = WG / KS, NPASS = (KS + RPT - 1) / RPT;
  constexpr bool ROWMAP = RPT > 0 && NPASS <= (P + WG - 1) / WG;
  const int f_row = tid / KS, f_col = tid - f_row * KS;
  auto fill_channel = [&](int c) {
    if constexpr (ROWMAP) {
      float v[JOBS][NPASS];
      const bool t_on = f_row < RPT;
#pragma unroll
      for (int j = 0; j < JOBS; ++j) {
        const int b = sh_edge[j * 6 + 0], y = sh_edge[j * 6 + 1], x = sh_edge[j * 6 + 2];
        const float *src = p.img[sh_edge[j * 6 + 4]] + ((size_t)b * C + c) * H * W + reflect_idx(x - HP + f_col, W);
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
          const int ry = f_row + k * RPT;
          v[j][k] = src[(size_t)reflect_idx(y - HP + (ry < KS ? ry : KS - 1), H) * W];
        }
      }
#pragma unroll
      for (int j = 0; j < JOBS; ++j)
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
          const int ry = f_row + k * RPT;
          if (t_on && ry < KS) tiles[j * CH + ry * S + f_col] = v[j][k];
        }
      return;
    }
    constexpr int EPT = (P + WG - 1) / WG;  // tile elements per thread
    float v[JOBS][EPT];
#pragma unroll
    for (int j = 0; j < JOBS; ++j) {
      const float *src = p.img[sh_edge[j * 6 + 4]];
      const int b = sh_edge[j * 6 + 0], y = sh_edge[j * 6 + 1], x = sh_edge[j * 6 + 2];
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * WG;
        const int ec = e < P ? e : P - 1;
        const int ry = ec / KS, rx = ec - ry * KS;
        v[j][k] = src[(((size_t)b * C + c) * H + reflect_idx(y - HP + ry, H)) * W + reflect_idx(x - HP + rx, W)];
      }
    }
#pragma unroll
    for (int j = 0; j < JOBS; ++j)
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * WG;
        if (e < P) tiles[j * CH + (e / KS) * S + e % KS] = v[j][k];
      }
  };
  __syncthreads();

  // ---- per-lane block ----
  int jl = tid / LPJ;
  const int m = tid - jl * LPJ;
  const bool lane_on = jl < JOBS;
  if (!lane_on) jl = 0;
  const int by = m / NB, bx = m - by * NB;
  const int ry0 = BS * by - HK, cx0 = BS * bx - HK;
  bool colv[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) colv[j] = (unsigned)(cx0 + j) < (unsigned)KS;

  float acc[BS][BS];
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) acc[i][j] = 0.f;

  const float *zrow = zero + HK;
  // per-job tile origin, row stride and channel stride inside LDS
  const int rs = MERGED ? MS : S, chs = MERGED ? MH * MS : CH;
  const float *tjob = MERGED ? tiles + (sh_edge[jl * 6 + 1] - my0) * MS + (sh_edge[jl * 6 + 2] - mx0) : tiles + jl * CH;
#pragma unroll 1
  for (int c = 0; c < (SSG_DBG(p, 2) ? 0 : C); ++c) {
    if constexpr (!MERGED) {
      if (c > 0) __syncthreads();               // every lane is done with the previous channel's tiles
      if (!SSG_DBG(p, 1)) fill_channel(c);
      __syncthreads();
    }
    const float *tc = MERGED ? tjob + c * chs : tjob;
    if constexpr (KW <= 9) {
      // centre window of this channel (uniform across the job's lanes: LDS broadcast reads).  Window row kh meets
      // patch row r for the block rows i = r - kh, i.e. during r = kh .. kh + BS - 1 only: it is fetched one patch row
      // ahead of its first use and dead BS rows later, so BS + 1 of the k_w rows are live at a time (54 instead of
      // 81 registers at k_w 9: the kernel drops under 128 VGPRs, 4 waves per SIMD instead of 3 -- a wave issues a
      // VALU instruction at most every ~4 cycles while the SIMD-32 takes one every 2, so the waves in flight are
      // what fills the pipe)
      float a[KW][KW];
      auto load_a = [&](int kh) {
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) a[kh][kx] = tc[(HP - HK + kh) * rs + (HP - HK + kx)];
      };
      load_a(0);
      // software pipeline: patch row r+1 is in flight while row r is consumed;
      // pin_block keeps hipcc from hoisting every row's loads to the top (which blew
      // the VGPR budget and spilled ~380 dwords per lane).
      float bn[PW];
      load_row<G>(tc, rs, zrow, ry0, cx0, colv, bn);
#pragma unroll
      for (int r = 0; r < PW; ++r) {
        float bv[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) bv[j] = bn[j];
        if (r + 1 < PW) load_row<G>(tc, rs, zrow, ry0 + r + 1, cx0, colv, bn);
        if (r + 1 < KW) load_a(r + 1);
#pragma unroll
        for (int i = 0; i < BS; ++i) {
          const int kh = r - i;
          if (kh < 0 || kh >= KW) continue;
          // tap-major order: BS independent (sub, fma) pairs back to back, so consecutive
          // instructions never wait on each other (accumulator-major order serialised every
          // FMA behind the previous one on the same accumulator)
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) {
            float d[BS];
#pragma unroll
            for (int j = 0; j < BS; ++j) d[j] = a[kh][kx] - bv[j + kx];
#pragma unroll
            for (int j = 0; j < BS; ++j) acc[i][j] = __builtin_fmaf(d[j], d[j], acc[i][j]);
          }
        }
        pin_block<BS, BS>(acc);
      }
    } else {
      // large windows: same fully unrolled, software-pipelined row stream; the centre-window row is re-read
      // from LDS (broadcast) where it is used instead of living in k_w^2 registers
      float bn[PW];
      load_row<G>(tc, rs, zrow, ry0, cx0, colv, bn);
#pragma unroll
      for (int r = 0; r < PW; ++r) {
        float bv[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) bv[j] = bn[j];
        if (r + 1 < PW) load_row<G>(tc, rs, zrow, ry0 + r + 1, cx0, colv, bn);
#pragma unroll
        for (int i = 0; i < BS; ++i) {
          const int kh = r - i;
          if (kh < 0 || kh >= KW) continue;
          const float *ar = tc + (HP - HK + kh) * rs + (HP - HK);
          float av[KW];
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) av[kx] = ar[kx];
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) {
            float d[BS];
#pragma unroll
            for (int j = 0; j < BS; ++j) d[j] = av[kx] - bv[j + kx];
#pragma unroll
            for (int j = 0; j < BS; ++j) acc[i][j] = __builtin_fmaf(d[j], d[j], acc[i][j]);
          }
        }
        pin_block<BS, BS>(acc);
      }
    }
  }

  const bool job_on = lane_on && sh_edge[jl * 6 + 3] >= 0;
  const int n = job_on ? sh_edge[jl * 6 + 3] : 0;
  const int which = job_on ? sh_edge[jl * 6 + 4] : 0;

  if (p.raw) {
    // reference operator: out[n,py,px] += D   (similarity.cu:49)
    if (job_on) {
      float *o = p.out[which] + (size_t)n * P;
#pragma unroll
      for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const int py = BS * by + i, px = BS * bx + j;
          if (py < KS && px < KS) o[py * KS + px] += acc[i][j];
        }
    }
    return true;
  }

  if (SSG_DBG(p, 4)) {
    if (acc[0][0] == 123.456f) p.out[0][0] = acc[1][1];
    return true;
  }
  // ---- epilogue: e = exp(-(D/den)/sigma), row sum, normalise ----
  // -(D/(C k_w^2))/sigma as one multiply by a host-rounded constant (|x| differs from the reference's two
  // divisions by <= 1.5 ulp).
  //
  // The row sum and the scale 1/(sum + eps) are carried in fp64 and every s = fl32(e * scale) is rounded on its
  // own.  With an fp32 scale all k_s^2 entries of a row share the scale's rounding error delta (~3e-8), and the
  // KL criterion sum t log(t/s) -- second order in (t - s) -- picks up the full (delta_t - delta_s) per row:
  // on flat rows (fixture F1) that is ~1.5e-5 of the loss, above the 1e-5 parity bar, for ANY fp32
  // normalisation (the reference's own fp32 run included).  ~100 fp64 ops per lane, 1 % of the kernel.
  // exp(x) = 2^(x log2 e) on v_exp_f32 with the constant folded on the host side of the multiply: the
  // argument's rounding (|x| * 6e-8) costs e a relative 1e-6 at |x| = 20 (e = 2e-9) and nothing near e = 1 --
  // far inside the 1e-5 budget on s in [0,1] -- and saves expf()'s ~12-instruction range reduction per offset.
  // Results below 2^-126 flush to 0 (expf would return denormals, i.e. < 1.2e-38).
  const float nk = (float)(-1.4426950408889634 / ((double)(C * KW * KW) * (double)p.sigma));
  double lsum = 0.0;
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int py = BS * by + i, px = BS * bx + j;
      const float e = (py < KS && px < KS) ? __builtin_amdgcn_exp2f(acc[i][j] * nk) : 0.f;
      acc[i][j] = e;
      lsum += (double)e;
    }
  red[tid] = lsum;
  __syncthreads();  // also: every lane is done reading the tiles -> reuse as staging
  if (p.generalization) {
    double tot = 0.0;
    for (int k = 0; k < LPJ; ++k) tot += red[jl * LPJ + k];
    const double scale = 1.0 / (tot + (double)p.eps);
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) acc[i][j] = (float)((double)acc[i][j] * scale);
  }
  float *stage = tiles + (MERGED ? jl * P : jl * CH);  // >= P floats per job
  if (lane_on) {
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int py = BS * by + i, px = BS * bx + j;
        if (py < KS && px < KS) stage[py * KS + px] = acc[i][j];
      }
  }
  __syncthreads();
  for (int j = 0; j < JOBS; ++j) {
    const int row = sh_edge[j * 6 + 3];
    if (row < 0) continue;
    float *o = p.out[sh_edge[j * 6 + 4]] + (size_t)row * P;
    const float *sj = tiles + (MERGED ? j * P : j * CH);
    for (int e = tid; e < P; e += WG) o[e] = sj[e];
  }
  return true;   // (LDS was used: the caller's loop needs a barrier before the next group)
}

// Job groups per workgroup.  k_s = 49 is the stress configuration, whose masks are dense: nearly every row belongs
// to the dense kernels and the host's bound (capacity) sizes a grid of 1e5 workgroups that start only to leave (4.2e5
// waves per launch, 40-70 us each at C5); there a workgroup walks 16 consecutive groups, so the empty grid is 16 times
// smaller.  k_s <= 25 keeps one group per workgroup (the loop and its register cost fold away).
template <class G>
constexpr int fwd_groups_per_wg() { return G::KS >= 49 ? 16 : 1; }
// k_s <= 25: one workgroup per group up to this many groups (C2's bench bound is 31 k), a looping tail behind them
constexpr unsigned FWD_MAIN_GROUPS = 40960, FWD_TAIL_GRID = 1024;

template <class G, bool MERGED>
__global__ __launch_bounds__(G::WG) __attribute__((amdgpu_waves_per_eu(2))) void ssg_fwd_tiled(FwdParams p) {
  constexpr int GPW = fwd_groups_per_wg<G>();
  if constexpr (GPW == 1) {
    fwd_tiled_group<G, MERGED>(p, (int)blockIdx.x);
  } else {
#pragma unroll 1
    for (int g = 0; g < GPW; ++g) {
      if (fwd_tiled_group<G, MERGED>(p, (int)blockIdx.x * GPW + g) > 0)
        __syncthreads();   // the group's LDS (job table, staging) is rewritten by the next one
    }
  }
}

// The groups from `first` on, walked by a small grid with stride gridDim: the tail of a launch whose bound on the rows
// (the caller's capacity) is far above what the main launch covers with one workgroup per group (FWD_MAIN_GROUPS).  Its
// workgroups leave at the first group behind the last job the DEVICE count knows, i.e. at once unless the step really has
// that many rows; the loop's registers (12 more: the merged variant would drop from 4 to 3 waves per SIMD) stay out of
// the main kernel.
template <class G, bool MERGED>
__global__ __launch_bounds__(G::WG) __attribute__((amdgpu_waves_per_eu(2))) void ssg_fwd_tiled_tail(FwdParams p, int first) {
#pragma unroll 1
  for (int grp = first + (int)blockIdx.x;; grp += (int)gridDim.x) {
    const int r = fwd_tiled_group<G, MERGED>(p, grp);
    if (r < 0) break;
    if (r > 0) __syncthreads();
  }
}

// Any odd (ks, kw): one workgroup (256 lanes) per job, tile in LDS.
__global__ __launch_bounds__(256) void ssg_fwd_generic(FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ks = p.ks, kw = p.kw, hp = ks / 2, hk = kw / 2, P = ks * ks;
  const int C = p.C, H = p.H, W = p.W, tid = threadIdx.x;
  float *tile = smem;        // [C][ks][ks]
  float *red = smem + ((C * P + 1) & ~1); // [256] doubles (8-byte aligned)
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const int q = blockIdx.x;
  if (q >= nrows * p.nimg) return;
  const int n = q / p.nimg, which = q % p.nimg;
  const Edge e = load_edge(p.edges, p.estride, n);
  const float *src = p.img[which] + (size_t)e.b * C * H * W;
  for (int i = tid; i < C * P; i += 256) {
    const int c = i / P, r = i - c * P, ry = r / ks, rx = r - ry * ks;
    tile[i] = src[((size_t)c * H + reflect_idx(e.y - hp + ry, H)) * W + reflect_idx(e.x - hp + rx, W)];
  }
  __syncthreads();
  const float den = (float)(C * kw * kw);
  float *o = p.out[which] + (size_t)n * P;
  double lsum = 0.0;
  for (int pidx = tid; pidx < P; pidx += 256) {
    const int py = pidx / ks, px = pidx - py * ks;
    float acc = 0.f;
    for (int c = 0; c < C; ++c)
      for (int kh = -hk; kh <= hk; ++kh)
        for (int kx = -hk; kx <= hk; ++kx) {
          const float a = tile[(c * ks + hp + kh) * ks + hp + kx];
          const int yy = py + kh, xx = px + kx;
          const bool in = (unsigned)yy < (unsigned)ks && (unsigned)xx < (unsigned)ks;
          const float d = in ? a - tile[(c * ks + yy) * ks + xx] : a;
          acc = __builtin_fmaf(d, d, acc);
        }
    if (p.raw) {
      o[pidx] += acc;
    } else {
      const float ev = expf(-1.f * (acc / den) / p.sigma);
      o[pidx] = ev;  // normalised below
      lsum += (double)ev;
    }
  }
  if (p.raw || !p.generalization) return;
  // fp64 row sum and scale, each entry rounded once (see the tiled kernel's epilogue)
  double *dred = (double *)red;
  dred[tid] = lsum;
  __syncthreads();
  double tot = 0.0;
  for (int k = 0; k < 256; ++k) tot += dred[k];
  const double scale = 1.0 / (tot + (double)p.eps);
  for (int pidx = tid; pidx < P; pidx += 256) o[pidx] = (float)(scale * (double)o[pidx]);
}

// ------------------------------------------------------------------ host ----
template <class G, bool MERGED>
static size_t fwd_lds_bytes(int C) {
  constexpr int PADF = (G::HK + 3) & ~3;
  constexpr int MH = G::KS + MERGE_ROWS - 1, MS = G::KS + MERGE_COLS;
  const size_t region = (size_t)C * MH * MS, rows = (size_t)G::JOBS * G::P;
  const size_t tiles = MERGED ? ((region > rows ? region : rows + 1) & ~(size_t)1) : (size_t)G::JOBS * G::CH;
  return sizeof(float) * (size_t)(PADF + tiles + ((G::ZROW + 3) & ~3) + 4 + 2 * G::WG) + sizeof(int) * 6 * G::JOBS;
}

template <class G, bool MERGED>
static int launch_fwd_tiled(const FwdParams &p, hipStream_t st) {
  const size_t lds = fwd_lds_bytes<G, MERGED>(p.C);
  if (lds > 160 * 1024) return -2;
  static std::atomic<unsigned long long> lds_set{0};
  if (const int rc = ensure_dynamic_lds(ssg_fwd_tiled<G, MERGED>, 160 * 1024, lds_set)) return rc;
  // tile order pads every image's jobs to a multiple of JOBS
  const long per_img = p.order ? ((long)p.n_host + G::JOBS - 1) / G::JOBS * G::JOBS : (long)p.n_host;
  const long njobs = per_img * p.nimg;
  if (njobs == 0) return 0;
  const long ngroups = (njobs + G::JOBS - 1) / G::JOBS;
  unsigned grid = (unsigned)((ngroups + fwd_groups_per_wg<G>() - 1) / fwd_groups_per_wg<G>());
  if constexpr (fwd_groups_per_wg<G>() == 1) {
    // the main grid: one workgroup per group up to FWD_MAIN_GROUPS -- or, with a hint of the rows to expect (the last
    // plan's), 1.25 x that: workgroups that start only to leave are dispatched at ~500 per us, which a short launch
    // (BASELINE's C4: 3.4 k direct rows under a capacity of every pixel) does not hide
    unsigned main_groups = FWD_MAIN_GROUPS;
    if (p.rows_hint > 0) {
      const long want = (((long)p.rows_hint * 5 / 4 + G::JOBS - 1) / G::JOBS + 32) * p.nimg;
      const long capped = want < 512 ? 512 : (want > (long)FWD_MAIN_GROUPS ? (long)FWD_MAIN_GROUPS : want);
      main_groups = (unsigned)((capped + 7) / 8 * 8);
    }
    // (a generous capacity: the groups behind the main grid go to the looping tail -- when there are enough of them to pay
    // for its launch: ~5 us against ~2 ns per workgroup that starts only to leave)
    if (grid > main_groups + 4096 || grid > FWD_MAIN_GROUPS) {
      static std::atomic<unsigned long long> lds_set_tail{0};
      if (const int rc = ensure_dynamic_lds(ssg_fwd_tiled_tail<G, MERGED>, 160 * 1024, lds_set_tail)) return rc;
      hipLaunchKernelGGL((ssg_fwd_tiled<G, MERGED>), dim3(main_groups), dim3(G::WG), lds, st, p);
      hipLaunchKernelGGL((ssg_fwd_tiled_tail<G, MERGED>), dim3(FWD_TAIL_GRID), dim3(G::WG), lds, st, p, (int)main_groups);
      return (int)hipGetLastError();
    }
  }
  hipLaunchKernelGGL((ssg_fwd_tiled<G, MERGED>), dim3(grid), dim3(G::WG), lds, st, p);
  return (int)hipGetLastError();
}

// Tile-ordered launch of both variants over the same job groups (merged groups run in the first, the rest in
// the second); without an order (reference operator, raw distances) the single variant alone in row order.
template <class G>
static int launch_fwd_pair(FwdParams p, hipStream_t st) {
  // (raw distances -- the reference operator -- take the same two variants when the caller brings a tile order: the
  // operator's plan path, ssg_api.hip; without an order the single variant walks the rows as they come)
  const bool can_merge = p.order && fwd_lds_bytes<G, true>(p.C) <= 160 * 1024;
  if (!can_merge) p.order = nullptr;
  p.small = 0;
  if constexpr (G::KS == 25 && G::KW == 9) {
    // the small-call variant (SMALL_CALL_JOBS): alone when the host's bound on the jobs says so, together with the
    // regular ones -- the device-side count picks -- while the bound is within 8x of it (a generous capacity), not at all
    // beyond (C2's 155 k jobs: no third launch)
    static const int mode = env_int("SSG_FWD_SMALL", 1) != 0;   // (profiling build: SSG_FWD_SMALL=0 = never, A/B measurements)
    // (with a hint of the rows to expect the variant is launched when THAT says "small" -- within 1.25 x, the device-side count
    // still picks -- whatever the host's bound: a generous capacity keeps it, a C2-sized call under a tight capacity does
    // not pay for its empty grid)
    const long bound = (long)p.n_host * p.nimg;
    const bool want = p.rows_hint > 0 ? 4L * p.rows_hint * p.nimg <= 5L * SMALL_CALL_JOBS : bound <= 8L * SMALL_CALL_JOBS;
    if (mode && want) {
      using GS3 = Geo<25, 9, 3, 256>;
      p.small = (long)p.n_host * p.nimg <= SMALL_CALL_JOBS ? 1 : 2;   // (alone only when the HOST's bound says so)
      const int rc = launch_fwd_tiled<GS3, false>(p, st);
      if (rc || p.small == 1) return rc;
    }
  }
  if (can_merge) {
    const int rc = launch_fwd_tiled<G, true>(p, st);
    if (rc) return rc;
  }
  return launch_fwd_tiled<G, false>(p, st);
}

int launch_fwd(const FwdParams &p_in, hipStream_t st) {
  FwdParams p = p_in;
  if (p.ks == 25 && p.kw == 9) return launch_fwd_pair<Geo<25, 9, 5, 128>>(p, st);
  // (49,13): 5 jobs x 49 lanes in 256-lane workgroups; the merged variant needs 48 KB of LDS (3 workgroups
  // per CU), the single one 147 KB (groups that straddle tiles or images only)
  if (p.ks == 49 && p.kw == 13 && p.order && fwd_lds_bytes<Geo<49, 13, 7, 256>, false>(p.C) <= 160 * 1024)
    return launch_fwd_pair<Geo<49, 13, 7, 256>>(p, st);
  p.order = nullptr;  // the other geometries run in row order
  if (p.ks == 11 && p.kw == 5) return launch_fwd_tiled<Geo<11, 5, 4, 64>, false>(p, st);
  if (p.ks == 49 && p.kw == 13 && fwd_lds_bytes<Geo<49, 13, 7, 128>, false>(p.C) <= 160 * 1024)
    return launch_fwd_tiled<Geo<49, 13, 7, 128>, false>(p, st);
  const size_t lds = sizeof(float) * ((size_t)p.C * p.ks * p.ks + 2 + 512);
  if (lds > 160 * 1024) return -2;
  static std::atomic<unsigned long long> lds_set{0};
  if (const int rc = ensure_dynamic_lds(ssg_fwd_generic, 160 * 1024, lds_set)) return rc;
  const long njobs = (long)p.n_host * p.nimg;
  if (njobs == 0) return 0;
  hipLaunchKernelGGL(ssg_fwd_generic, dim3((unsigned)njobs), dim3(256), lds, st, p);
  return (int)hipGetLastError();
}

const char *fwd_kernel_name(int ks, int kw) {
  if (ks == 25 && kw == 9) return "ssg_fwd_tiled<Geo<25,9,5,128>,merged|single>";
  if (ks == 11 && kw == 5) return "ssg_fwd_tiled<Geo<11,5,4,64>,single>";
  if (ks == 49 && kw == 13) return "ssg_fwd_tiled<Geo<49,13,7,256>,merged|single>";
  return "ssg_fwd_generic";
}

}  // namespace ssg
