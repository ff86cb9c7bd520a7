// Dense-tile ("shared-term") forward SSG kernel for gfx950.
//
// The direct kernels (ssg_fwd.hip) spend 2*C*k_w^2 VALU ops per (edge pixel, search offset).
// Where edge pixels are dense, most of that work is shared: for an offset q the per-pixel term
//     E_q[u] = sum_c ( I[c,u] - I[c,u+q] )^2
// is the same for every edge pixel whose k_w x k_w window covers u, and
//     D[n,q] = sum_{k in K(q)} E_q[x_n + k]  +  sum_{k in win \ K(q)} |I[x_n + k]|^2
// where K(q) = { k : |k + q|_inf <= k_s/2 } is the part of the window whose partner stays inside
// the search area (the reference's "B = 0 outside the area" rule, similarity.cu:43-47 ==
// F.unfold zero padding, loss_util.py:208); K(q) is a sub-rectangle [ylo,yhi] x [xlo,xhi] of the
// window that depends on q only.  Both parts are separable box sums of non-negative fields, so
// everything here is plain additions of non-negative numbers: no sliding subtraction, no prefix
// sums (both break the 1e-5 budget at sigma = 0.004).
//
// One workgroup (4 waves) owns a tile of 8 x 32 candidate centres.  Its (8+2*16) x (32+2*16) x C
// image region sits in LDS.  Wave w walks the offset rows q_y = w, w+4, ...; inside a row the 25
// (k_s) offsets q_x are fully unrolled: lane (r, g) keeps its own 16 pixels of U-row r (U = tile
// grown by the window halo) in registers, slides the 16-pixel window of I[u+q] by one pixel per
// step (3 LDS dwords), forms E, then the horizontal box sums of its 8 centre columns with
// compile-time truncation [xlo,xhi], and writes them to the wave's H buffer.  The tile's edge
// pixels (census from the rank map) then add the vertical taps [ylo,yhi], the |I|^2 complement,
// apply exp and store e[n,q]; the row sums are accumulated on the fly and the rows are rescaled
// by 1/(sum + eps) at the end (L2-resident re-read of what the workgroup just wrote).
//
// Cost: ~235 wave-instructions per (tile, offset) regardless of the number of edge pixels, against
// 25 * 12.6 k lane-instructions per edge pixel for the direct kernels: break-even at ~27 edge
// pixels per 256-pixel tile; the edge-list builder routes tiles above the threshold here.
#include "ssg_common.hpp"

namespace ssg {

struct DenseParams {
  const float *img[2];
  float *out[2];
  int nimg;
  const int *rank;      // (B,H,W) row of every pixel, -1 if not an edge pixel
  const int *n_dense;   // device count of dense tiles
  const int *tiles;     // dense tile ids
  int max_tiles;        // launch bound per image slot
  const int *n_dev;     // rows computed at all (capacity clamp), nullable
  int n_host;
  int B, H, W;
  float sigma, eps;
  int generalization;
  int dbg;  // profiling ablations: bit0 no stores, bit1 no edge stage, bit3 no rescale, bit6 no main loop
  double *row_scale;  // nullable [nimg][n_host]: deferred normalisation -- the rows stay e, 1/(sum e + eps) goes here
};

constexpr int DT_X = 32;  // centre columns per tile; rows: DT_Y = 16 - (k_w - 1) (8 for k_w = 9, 4 for k_w = 13)
// Row stride of the H buffers (horizontal sums on the 16 x 32 centre columns of U).  The tile's edge pixels read
// them at (row ey + k, column ex): with stride 32 every pixel of a VERTICAL edge line falls on one LDS bank (half of
// the kernel's LDS cycles were conflict replays, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.49); with 44 two
// pixels collide only if dx = -12 dy (mod 32), which no edge curve does inside an 8-row tile.
constexpr int DT_HS = 44;

// NW waves per workgroup share the tile's image region; wave w walks the offset rows q_y = w (mod NW).  (25,9): 4
// (47 KB of LDS, two workgroups per CU); (49,13): 8 -- its 71 KB region allows one workgroup per CU, and a lone
// wave per SIMD issues a VALU instruction only every ~4 cycles (5.4 -> 3.x ms at C5).
template <int KS, int KW, int C, int NW>
__global__ __launch_bounds__(64 * NW) void ssg_fwd_dense(DenseParams p) {
  constexpr int NT = 64 * NW;
  constexpr int HP = KS / 2, HK = KW / 2, P = KS * KS, HALO = HP + HK, DT_Y = 16 - 2 * HK;
  constexpr int RH = DT_Y + 2 * HALO, RWD = DT_X + 2 * HALO, RS = RWD + 1;  // image region
  constexpr int UH = DT_Y + 2 * HK, UW = DT_X + 2 * HK;                      // window halo U
  constexpr int LW = 8 + KW - 1;                                             // U columns per lane
  constexpr int NE_MAX = DT_Y * DT_X, NCHUNK = NE_MAX / 64;
  static_assert(UH == 16 && DT_X == 32, "lane map: 16 U-rows x 4 column groups of 8 centres");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *reg = smem;                       // [C][RH][RS]
  float *F = reg + C * RH * RS;            // [UH][UW]   sum_c I^2 on U
  float *HF = F + UH * UW;                 // [UH][DT_HS] full-window horizontal sums of F
  float *Hb = HF + UH * DT_HS;             // [NW][UH][DT_HS] per-wave horizontal sums of E_q
  // per-wave partial row sums (fp64, see ssg_fwd.hip): wave w's NE_MAX doubles reuse ITS OWN H buffer once its
  // offset rows are done (same size, wave-private, so no other wave is still reading it)
  double *rsum = (double *)Hb;                    // [NW][RSTR], RSTR = one H buffer in doubles
  constexpr int RSTR = UH * DT_HS / 2;
  static_assert(NE_MAX <= RSTR, "row sums alias the wave's H buffer");
  int *elist = (int *)(Hb + NW * UH * DT_HS);     // [NE_MAX][3] (ey, ex, row)
  int *misc = elist + NE_MAX * 3;          // [16]: wave counts, n_e (misc[NW])

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int which = blockIdx.x / p.max_tiles, tslot = blockIdx.x - which * p.max_tiles;
  if (tslot >= *p.n_dense) return;
  const int H = p.H, W = p.W;
  const int tx_n = (W + DT_X - 1) / DT_X, ty_n = (H + DT_Y - 1) / DT_Y;
  const int tile = p.tiles[tslot];
  const int b = tile / (tx_n * ty_n), tr = tile - b * tx_n * ty_n;
  const int ty0 = (tr / tx_n) * DT_Y, tx0 = (tr % tx_n) * DT_X;
  const int nrows = rows_to_do(p.n_dev, p.n_host);

  // ---- census of the tile's edge pixels (row-major inside the tile) ----
  {
    const int ey = tid / DT_X, ex = tid % DT_X;
    const int y = ty0 + ey, x = tx0 + ex;
    int r = (ey < DT_Y && y < H && x < W) ? p.rank[((size_t)b * H + y) * W + x] : -1;
    if (r >= nrows) r = -1;
    const unsigned long long bal = __ballot(r >= 0);
    if (lane == 0) misc[wv] = __popcll(bal);
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; ++k) base += misc[k];
    if (r >= 0) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      elist[3 * pos + 0] = ey;
      elist[3 * pos + 1] = ex;
      elist[3 * pos + 2] = r;
    }
    if (tid == 0) {
      int t = 0;
      for (int k = 0; k < NW; ++k) t += misc[k];
      misc[NW] = t;
    }
  }
  // ---- image region: C x RH x RWD, reflect by index mirroring (clamped: far corners of
  // tiles that overhang a small image are never used, but must stay in bounds) ----
  {
    const float *src = p.img[which] + (size_t)b * C * H * W;
    constexpr int CPLF = (RWD + 15) / 16;    // 16 lanes per row, CPLF consecutive pixels each
    const int lx = tid % 16, lr = tid / 16;
    for (int R0 = 0; R0 < C * RH; R0 += NT / 16) {
      const int R = R0 + lr;
      if (R < C * RH) {
        const int c = R / RH, ry = R - c * RH;
        int gy = reflect_idx(ty0 - HALO + ry, H);
        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        const float *srow = src + ((size_t)c * H + gy) * W;
        float v[CPLF];
#pragma unroll
        for (int k = 0; k < CPLF; ++k) {
          int gx = reflect_idx(tx0 - HALO + lx * CPLF + k, W);
          gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
          v[k] = srow[gx];
        }
#pragma unroll
        for (int k = 0; k < CPLF; ++k)
          if (lx * CPLF + k < RWD) reg[(c * RH + ry) * RS + lx * CPLF + k] = v[k];
      }
    }
  }
  __syncthreads();
  const int n_e = misc[NW];
  for (int i = tid; i < UH * UW; i += NT) {
    const int ur = i / UW, uc = i - ur * UW;
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float v = reg[(c * RH + ur + HP) * RS + uc + HP];
      t = __builtin_fmaf(v, v, t);
    }
    F[i] = t;
  }
  __syncthreads();
  for (int i = tid; i < UH * DT_X; i += NT) {
    const int ur = i / DT_X, tc = i - ur * DT_X;
    float t = 0.f;
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) t += F[ur * UW + tc + kx];
    HF[ur * DT_HS + tc] = t;
  }
  __syncthreads();

  // ---- main loop: wave wv takes offset rows qyi = wv, wv+4, ... ----
  const int r = lane >> 2, g = lane & 3;
  float iu[C][LW];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = 0; i < LW; ++i) iu[c][i] = reg[(c * RH + r + HP) * RS + 8 * g + HP + i];
  float Fr[LW];  // |I|^2 of the lane's LW pixels (complement of the columns that leave the area)
#pragma unroll
  for (int i = 0; i < LW; ++i) Fr[i] = F[r * UW + 8 * g + i];
  float *hb = Hb + wv * UH * DT_HS;
  float *hrow = hb + r * DT_HS + 8 * g;
  // exp(x) = 2^(x log2 e), constant folded (see ssg_fwd.hip)
  const float nk = (float)(-1.4426950408889634 / ((double)(C * KW * KW) * (double)p.sigma));
  double rs[NCHUNK];
#pragma unroll
  for (int k = 0; k < NCHUNK; ++k) rs[k] = 0.0;
  float *outp = p.out[which];
  // this lane's edge pixels (one per chunk of 64 list entries), hoisted out of the offset loops
  int hoff[NCHUNK];
  size_t orow[NCHUNK];
  bool eon[NCHUNK];
#pragma unroll
  for (int ck = 0; ck < NCHUNK; ++ck) {
    const int e = ck * 64 + lane;
    eon[ck] = e < n_e;
    const int ec = eon[ck] ? e : 0;
    const int ey = elist[3 * ec], ex = elist[3 * ec + 1];
    hoff[ck] = ey * DT_HS + ex;  // window row k of the centre is U-row ey + k
    orow[ck] = (size_t)elist[3 * ec + 2] * P;
  }

  const int n_e_stage = (p.dbg & 2) ? 0 : n_e;  // profiling ablations: 2 no edge stage, 1 no stores, 64 no main loop
  const bool do_store = !(p.dbg & 1);
#pragma unroll 1
  for (int qyi = wv; qyi < ((p.dbg & 64) ? 0 : KS); qyi += NW) {
    // D[n,q] = sum_{k in K(q)} E_q[x+k] + sum_{k not in K(q)} |I[x+k]|^2 with K(q) = rows [ylo,yhi] x columns
    // [xlo,xhi] of the window.  Rows: wave-uniform 0/1 weights.  Columns: the lane adds the |I|^2 of the
    // columns that left (compile-time set) to its horizontal sums, so H' rows carry E inside and |I|^2 outside
    // [xlo,xhi]; rows outside [ylo,yhi] contribute their full-width |I|^2 sums (HF), gathered once per q_y.
    const int ylo = (-HK > -qyi) ? -HK : -qyi, yhi = (HK < KS - 1 - qyi) ? HK : KS - 1 - qyi;
    float wgt[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) wgt[k] = (k - HK >= ylo && k - HK <= yhi) ? 1.f : 0.f;
    float av[NCHUNK];
#pragma unroll
    for (int ck = 0; ck < NCHUNK; ++ck) {
      av[ck] = 0.f;
      if (ck * 64 < n_e && (ylo > -HK || yhi < HK)) {
        const float *fc = HF + hoff[ck];
#pragma unroll
        for (int k = 0; k < KW; ++k) av[ck] = __builtin_fmaf(1.f - wgt[k], fc[k * DT_HS], av[ck]);
      }
    }
    const float *rq = reg + (r + qyi) * RS + 8 * g;  // + c*RH*RS + column (i + qxi)
    float w[C][LW];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int i = 0; i < LW; ++i) w[c][i] = rq[c * RH * RS + i];
    constexpr int SB = 13;  // consecutive offsets per edge pixel buffered in registers and stored together
    float evb[NCHUNK][SB];
    static_for(std::make_integer_sequence<int, KS>{}, [&](auto qc) {
      constexpr int qxi = decltype(qc)::value;
      constexpr int xlo = (-HK > -qxi) ? -HK : -qxi, xhi = (HK < KS - 1 - qxi) ? HK : KS - 1 - qxi;
      // E_q on the lane's LW pixels; the window of I[u+q] is a circular buffer whose slot index is
      // a compile-time function of the step: pixel i of step qxi lives in slot (i + qxi) % LW
      float E[LW];
#pragma unroll
      for (int i = 0; i < LW; ++i) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float d = iu[c][i] - w[c][(i + qxi) % LW];
          t = __builtin_fmaf(d, d, t);
        }
        // pixels whose column left the area enter the sums with |I|^2 instead of E_q: window tap kx of centre
        // column j is pixel i = j + HK + kx, i.e. kx = i - HK - j; only one-sided sets occur, so the swap can be
        // made per (i, j) at compile time below
        E[i] = t;
      }
      // horizontal sums for the 8 centre columns: E on taps [xlo, xhi], |I|^2 on the others
      float Hs[8];
      if constexpr (xlo == -HK && xhi == HK && KW == 9) {
        // full 9-tap windows (17 of the 25 steps): pair / quad / octet sums shared between the 8 outputs,
        // 44 additions instead of 64
        float p2[15], p4[13];
#pragma unroll
        for (int i = 0; i < 15; ++i) p2[i] = E[i] + E[i + 1];
#pragma unroll
        for (int i = 0; i < 13; ++i) p4[i] = p2[i] + p2[i + 2];
#pragma unroll
        for (int j = 0; j < 8; ++j) Hs[j] = (p4[j] + p4[j + 4]) + E[j + 8];
      } else if constexpr (xlo == -HK && xhi == HK && KW == 13) {
        // full 13-tap windows: 13 = 8 + 4 + 1 from shared pair / quad / octet sums (53 additions instead of 96)
        float p2[19], p4[17], p8[8];
#pragma unroll
        for (int i = 0; i < 19; ++i) p2[i] = E[i] + E[i + 1];
#pragma unroll
        for (int i = 0; i < 17; ++i) p4[i] = p2[i] + p2[i + 2];
#pragma unroll
        for (int j = 0; j < 8; ++j) p8[j] = p4[j] + p4[j + 4];
#pragma unroll
        for (int j = 0; j < 8; ++j) Hs[j] = (p8[j] + p4[j + 8]) + E[j + 12];
      } else
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = E[j + HK + xlo];
#pragma unroll
        for (int kx = xlo + 1; kx <= xhi; ++kx) t += E[j + HK + kx];
#pragma unroll
        for (int kx = -HK; kx < xlo; ++kx) t += Fr[j + HK + kx];
#pragma unroll
        for (int kx = xhi + 1; kx <= HK; ++kx) t += Fr[j + HK + kx];
        Hs[j] = t;
      }
      *(float4 *)(hrow) = make_float4(Hs[0], Hs[1], Hs[2], Hs[3]);
      *(float4 *)(hrow + 4) = make_float4(Hs[4], Hs[5], Hs[6], Hs[7]);
      // next q_x: the pixel that leaves the window (slot qxi % LW) is replaced by the one that enters
      if (qxi + 1 < KS) {
#pragma unroll
        for (int c = 0; c < C; ++c) w[c][qxi % LW] = rq[c * RH * RS + LW + qxi];
      }
      // (the H buffer is private to this wave and LDS operations of one wave execute in issue order: the reads
      // below see the writes above, and the next step's writes cannot overtake them -- no wait, no barrier)
      // ---- the tile's edge pixels: weighted vertical taps, row complement, exp, store ----
#pragma unroll
      for (int ck = 0; ck < NCHUNK; ++ck) {
        if (ck * 64 < n_e_stage) {
          const float *hc = hb + hoff[ck];
          float hv[KW];
#pragma unroll
          for (int k = 0; k < KW; ++k) hv[k] = hc[k * DT_HS];
          float d = av[ck];
#pragma unroll
          for (int k = 0; k < KW; ++k) d = __builtin_fmaf(wgt[k], hv[k], d);
          const float ev = __builtin_amdgcn_exp2f(d * nk);
          rs[ck] += (double)ev;
          // every lane writes its own SSG row: single dwords are one L2 request per lane and step (request-
          // rate bound at high density), so SB consecutive offsets leave together as 16-byte stores
          evb[ck][qxi % SB] = ev;
          if (eon[ck] && do_store) {
            float *o = outp + orow[ck] + qyi * KS;
            if constexpr (qxi % SB == SB - 1 || qxi == KS - 1) {
              constexpr int cnt = qxi % SB + 1, q0 = qxi - (cnt - 1);
#pragma unroll
              for (int t = 0; t + 4 <= cnt; t += 4) {
                float4 v4 = make_float4(evb[ck][t], evb[ck][t + 1], evb[ck][t + 2], evb[ck][t + 3]);
                __builtin_memcpy(o + q0 + t, &v4, 16);
              }
#pragma unroll
              for (int t = cnt & ~3; t < cnt; ++t) o[q0 + t] = evb[ck][t];
            }
          }
        }
      }
    });
  }

  // ---- row sums over the four waves, then rescale the rows this workgroup wrote ----
#pragma unroll
  for (int ck = 0; ck < NCHUNK; ++ck)
    if (ck * 64 + lane < n_e) rsum[wv * RSTR + ck * 64 + lane] = rs[ck];
  __threadfence_block();
  __syncthreads();
  if (!p.generalization || (p.dbg & 8)) return;
  if (p.row_scale) {
    // deferred normalisation: the consumer that streams the rows anyway (ssg_grad_rows) rescales them; saves this
    // kernel's second pass over its rows (one read + one write of every row)
    double *rsc = p.row_scale + (size_t)which * p.n_host;
    for (int e = tid; e < n_e; e += NT) {
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < NW; ++k) tot += rsum[k * RSTR + e];
      rsc[elist[3 * e + 2]] = 1.0 / (tot + (double)p.eps);
    }
    return;
  }
  // global stores of this workgroup must be visible to its own later loads: same CU, L1 is
  // write-through, the loads below are issued after the barrier + vmcnt drain
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // two rows per wave and iteration: both rows' loads are in flight before the first store (the pass is a chain of
  // L2 round trips otherwise: 0.12 of the kernel's 0.52 ms at C2)
  constexpr int RPL = (P + 63) / 64;  // row elements per lane
  for (int e0 = 2 * wv; e0 < n_e; e0 += 2 * NW) {
    float v[2][RPL];
    double scale[2];
    float *o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = e0 + j < n_e ? e0 + j : e0;
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < NW; ++k) tot += rsum[k * RSTR + e];
      scale[j] = 1.0 / (tot + (double)p.eps);
      o[j] = outp + (size_t)elist[3 * e + 2] * P;
#pragma unroll
      for (int k = 0; k < RPL; ++k) {
        const int q = lane + 64 * k;
        v[j][k] = q < P ? __builtin_nontemporal_load(o[j] + q) : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (e0 + j < n_e) {
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
          const int q = lane + 64 * k;
          if (q < P) o[j][q] = (float)(scale[j] * (double)v[j][k]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ host ----
template <int KS, int KW, int C, int NW>
static size_t dense_lds_bytes() {
  constexpr int DT_Y = 16 - 2 * (KW / 2);
  constexpr int HALO = KS / 2 + KW / 2, RH = DT_Y + 2 * HALO, RS = DT_X + 2 * HALO + 1;
  constexpr int UH = DT_Y + 2 * (KW / 2), UW = DT_X + 2 * (KW / 2), NE = DT_Y * DT_X;
  return sizeof(float) * (size_t)(C * RH * RS + UH * UW + UH * DT_HS + NW * UH * DT_HS) + sizeof(int) * (NE * 3 + 16);
}

bool dense_supported(int ks, int kw, int C) { return C == 3 && ((ks == 25 && kw == 9) || (ks == 49 && kw == 13)); }

// rows of the dense kernels' tiles for a search size (the plan of ssg_edge_list is built for it)
int dense_tile_rows(int ks) { return ks == 49 ? 4 : 8; }

int dense_max_tiles(int B, int H, int W, int ks) {
  const int ty = dense_tile_rows(ks);
  return B * ((H + ty - 1) / ty) * ((W + DT_X - 1) / DT_X);
}

template <int KS, int KW, int C, int NW>
static int launch_fwd_dense_t(const DenseParams &p, hipStream_t st) {
  const size_t lds = dense_lds_bytes<KS, KW, C, NW>();
  static std::atomic<unsigned long long> lds_set{0};
  if (const int rc = ensure_dynamic_lds(ssg_fwd_dense<KS, KW, C, NW>, (int)lds, lds_set)) return rc;
  hipLaunchKernelGGL((ssg_fwd_dense<KS, KW, C, NW>), dim3((unsigned)p.max_tiles * p.nimg), dim3(64 * NW), lds, st, p);
  return (int)hipGetLastError();
}

int launch_fwd_dense(const DenseParams &p, int ks, int kw, int C, hipStream_t st) {
  if (!dense_supported(ks, kw, C)) return -1;
  if (p.max_tiles == 0) return 0;
  return ks == 25 ? launch_fwd_dense_t<25, 9, 3, 4>(p, st) : launch_fwd_dense_t<49, 13, 3, 8>(p, st);
}

}  // namespace ssg
