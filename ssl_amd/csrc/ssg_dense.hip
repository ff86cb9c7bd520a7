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
// One workgroup (NW waves: 4 for (25,9), 7 for (49,13)) owns a tile of DT_Y x 32 candidate centres (DT_Y = 8 for
// k_w 9, 4 for k_w 13, so that the tile grown by the window halo, U, is always 16 rows).  Its (DT_Y + 2 HALO) x
// (32 + 2 HALO) x C image region sits in LDS.  Wave w walks the offset rows q_y = w (mod NW); inside a row the k_s
// offsets q_x are fully unrolled.  Lane (U-row r, column group g) -- the four groups of a row are one DPP quad -- owns
// L = 10 (12) pixels of the row WITHOUT overlap as L/2 packed-fp32 register pairs (pixel j, pixel j + L/2), keeps a
// circular window of I[u+q] that advances by one pixel per step (C LDS dwords), forms E with v_pk_add / v_pk_fma,
// then the horizontal box sums of its L centre columns: for full windows from shared prefix / suffix blocks plus the
// neighbouring lane's prefix through a quad_perm DPP operand, for truncated windows [xlo,xhi] (compile-time per
// step) from the row continued into the next lane, E inside the taps kept and |I|^2 outside.  The sums go to the
// wave's private H buffer in LDS; the tile's edge pixels (census from the rank map, counting-sorted by H-buffer bank
// for k_w 9) add the vertical taps [ylo,yhi] as a depth-5 tree with 0/1 weights, the |I|^2 complement, apply
// v_exp_f32 and store e[n,q] (13 buffered offsets per 16-byte store group).  Row sums are carried in fp64; the rows
// are either rescaled by 1/(sum + eps) at the end (L2-resident re-read of what the workgroup just wrote) or -- the
// fused step -- left un-normalised with 1/(sum + eps) in `row_scale` for ssg_grad_rows to apply (deferred
// normalisation: one pass over the rows less).
//
// Cost (C2, SQ counters): ~95 VALU (42 % of them packed) + 14 LDS instructions per wave and offset step regardless of
// the number of edge pixels, plus ~20 per 64 edge pixels of the tile; the direct kernels spend 2 C k_w^2 lane-ops per
// (edge pixel, offset).  Break-even is ~16-28 edge pixels per 256-pixel tile (flat: tools/thr_sweep.sh); the
// edge-list builder routes tiles at or above the threshold here.
//
// Two more things live in this file for k_s = 49 (round 3): the TM variant of the tile kernel, which leaves e in the
// TILE-MAJOR scratch region ([slot][offset][128 pixels]: every wave store one aligned 256-byte run, no store
// buffering) instead of the caller's row-major rows, and ssg_fwd_strip, which computes the forward rows of whole strips
// of nine heavy tiles at once for tile-major calls (E/H shared by 36 centre rows).  ssg_common.hpp (TM_PX, tm_active,
// TmRowsParams) and DESIGN.md sections 3-5 describe the layout and who reads it.
#include "ssg_common.hpp"

namespace ssg {

typedef float f2 __attribute__((ext_vector_type(2)));

// value of lane + D inside the lane's quad (the last lanes read the quad's last lane: their sums belong to centres
// outside the tile)
template <int D>
__device__ __forceinline__ float quad_next(float v) {
  constexpr int ctrl = D == 1 ? 0xF9 : 0xFE;  // quad_perm [1,2,3,3] / [2,3,3,3]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}

// Sums of consecutive entries of a register array as trees of power-of-two blocks (the compiler shares the blocks
// between the sums that contain them): depth log2 instead of the length -- a chain of 9 additions per window costs
// the KL of nearly flat rows its 1e-5 (tests: kl_conditioning_on_flat_rows).
template <int LO, int LEN, typename T, int N>
__device__ __forceinline__ T block_sum(const T (&e)[N]) {
  static_assert(LEN >= 1 && (LEN & (LEN - 1)) == 0 && LO >= 0 && LO + LEN <= N, "power-of-two block inside the array");
  if constexpr (LEN == 1) return e[LO];
  else return block_sum<LO, LEN / 2>(e) + block_sum<LO + LEN / 2, LEN / 2>(e);
}
constexpr int floor_pow2(int n) { return n < 2 ? 1 : 2 * floor_pow2(n / 2); }
// e[LO] + ... + e[LO+LEN-1], blocks aligned from LO (prefix sums share them)
template <int LO, int LEN, typename T, int N>
__device__ __forceinline__ T sum_from(const T (&e)[N]) {
  constexpr int B = floor_pow2(LEN);
  if constexpr (B == LEN) return block_sum<LO, LEN>(e);
  else return block_sum<LO, B>(e) + sum_from<LO + B, LEN - B>(e);
}
// e[HI-LEN] + ... + e[HI-1], blocks aligned from HI downwards (suffix sums share them)
template <int HI, int LEN, typename T, int N>
__device__ __forceinline__ T sum_upto(const T (&e)[N]) {
  constexpr int B = floor_pow2(LEN);
  if constexpr (B == LEN) return block_sum<HI - LEN, LEN>(e);
  else return block_sum<HI - B, B>(e) + sum_upto<HI - B, LEN - B>(e);
}

// sum_k w[k] v[k] + rest with 0/1 weights (the products are exact) as a balanced tree of packed operations: two
// accumulator pairs take the taps four at a time.  Depth 5 instead of a chain of KW dependent FMAs -- the chain's
// rounding costs the KL of nearly flat rows its 1e-5 (tests: kl_conditioning_on_flat_rows), and its latency the
// two waves of a SIMD cannot hide.
template <int KW>
__device__ __forceinline__ float tap_sum(const float (&v)[KW], const float (&w)[KW], float rest) {
  static_assert(KW % 2 == 1 && KW >= 5, "pairs of taps and a last one");
  constexpr int NP = KW / 2;
  f2 a = f2{v[0], v[1]} * f2{w[0], w[1]}, b = f2{v[2], v[3]} * f2{w[2], w[3]};
#pragma unroll
  for (int j = 2; j < NP; ++j) {
    const f2 vv = f2{v[2 * j], v[2 * j + 1]}, ww = f2{w[2 * j], w[2 * j + 1]};
    if (j % 2 == 0) a = __builtin_elementwise_fma(vv, ww, a);
    else b = __builtin_elementwise_fma(vv, ww, b);
  }
  const f2 r2 = a + b;
  return (r2.x + r2.y) + __builtin_fmaf(w[KW - 1], v[KW - 1], rest);
}

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
  int dbg;  // profiling ablations: bit0 no stores, bit1 no edge stage, bit2 (strips) no E/H stage, bit3 no rescale, bit6 no main loop
  double *row_scale;  // nullable [nimg][n_host]: deferred normalisation -- the rows stay e, 1/(sum e + eps) goes here
  // tile-major scratch rows (fused step at k_s = 49, ssg_api.hip; tm_active() in ssg_common.hpp decides per call):
  // the tile in plan slot t leaves its e values at tm[img] + t * P * 128 + q * 128 + (64 ck + lane) -- every wave
  // store is one aligned 256-byte run -- and marks its rows with a NEGATIVE row scale; nullptr / 0 = row-major rows only
  float *tm[2];
  int tm_slots;
  int grid_tiles;  // plan slots this launch covers (per image)
  const int *strips;  // k_s 49 tile-major calls: [0] number of strips, then (strip id, first slot) pairs; nullable
  int max_strips;     // launch bound per image
  int *status;        // nullable: library-owned device status word (ssg_device_status): bit 0 = plan of another tile height
  int raw;            // 1: the reference operator's output -- out[n, q] += D[n, q] (similarity.cu:49), no epilogue
};

constexpr int DT_X = 32;  // centre columns per tile; rows: DT_Y = 16 - (k_w - 1) (8 for k_w = 9, 4 for k_w = 13)
// Pixels of a U-row per lane: the 4 lanes of a quad cover the row without overlap (10 x 4 = 40 = U for k_w 9; 12 x 4
// = 48 >= 44 for k_w 13, even so that the pixels pair up for packed fp32 math).
constexpr int dense_lane_px(int kw) { return kw == 9 ? 10 : 12; }
// H buffers (horizontal sums on the 16 U-rows x 4L centre columns): lane (r, g) stores its L sums in row r, the
// tile's edge pixels gather them at (row ey + k, column of ex).  The LDS, not the VALU, bounds this kernel, so the
// layout is chosen per size for what its tiles do most:
//   k_w 9  (8 x 32 tiles, a few dozen edge pixels each): the lane's register pairs (centres k, k + L/2) go out as
//          8-byte stores at r * 40 + 10 g + 2 (k % 5).  A ds_write_b64 is served 16 lanes (4 rows x 4 groups) at a
//          time on 32 banks, and a row stride of 8 (mod 32) puts the 16 pairs on 16 different bank pairs.  Measured
//          (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per launch at C2): stride 44: 7.6e7 / 1.46e8; stride 40: 4.1e7 /
//          1.11e8; 16-byte stores at r * 48 + 12 g: 5.3e7 / 1.23e8.  The gathers are kept apart by the order of the
//          tile's edge list instead (census).
//   k_w 13 (4 x 32 tiles, in practice full ones -- C5 is a 100 % mask): gathers first -- 32 lanes read 32 consecutive
//          centres of one row, conflict-free only if column = ex; the stores are then single dwords (12 g + k at an
//          odd row stride: 2-way, the best a contiguous row allows).
constexpr bool dense_h_paired(int kw) { return kw == 9; }
constexpr int dense_h_group(int kw) { return kw == 9 ? 10 : 12; }
constexpr int dense_h_stride(int kw) { return kw == 9 ? 40 : 49; }
__device__ __forceinline__ int dense_h_col(int ex, int L, int G, bool paired) {
  const int k = ex % L;
  return paired ? G * (ex / L) + 2 * (k % (L / 2)) + k / (L / 2) : ex;
}

// NW waves per workgroup share the tile's image region; wave w walks the offset rows q_y = w (mod NW).  (25,9): 4
// (47 KB of LDS, two workgroups per CU, 2 waves per SIMD -- hence the register cap: with 14 AGPRs on top of 256 VGPRs
// only one workgroup fits and the kernel takes 0.63 instead of 0.45 ms); (49,13): its 71 KB region allows one
// workgroup per CU, and a lone wave per SIMD issues a VALU instruction only every ~4 cycles, so 7 waves -- 49 rows
// are 7 each, with 8 the workgroup waits for the one wave that has a 7th row.
// TM: the launch writes tile-major scratch rows (k_w 13 only: the fixed pixel map below is the layout's pixel index)
// RAW: raw squared distances ACCUMULATED into the rows (the reference operator, ssg_compute_similarity with a plan)
// NCH (k_w 9 only; 0 = every chunk): 64-slot chunks of the tile's edge list this instantiation carries.  The plan marks
// the 8 x 32 tiles with more than 128 edge pixels (TILE_HUGE): they run in the NCH = 4 instantiation, every other tile in
// NCH = 2, whose edge stage is SOFTWARE-PIPELINED -- the nine taps of an offset are gathered right behind the H stores
// and consumed one step later, behind the next offset's E / H arithmetic, so that no step waits for its own LDS round
// trip (18 registers; with four chunks it would be 36 more than the kernel has).  Measured (same-box A/B, C2): dense
// forward 0.396 -> 0.374 ms.
template <int KS, int KW, int C, int NW, bool TM = false, bool RAW = false, int NCH = 0>
__device__ __forceinline__ void fwd_dense_body(const DenseParams &p) {
  static_assert(!(TM && RAW), "raw distances go to the caller's row-major rows");
  constexpr int NT = 64 * NW;
  constexpr int HP = KS / 2, HK = KW / 2, P = KS * KS, HALO = HP + HK, DT_Y = 16 - 2 * HK;
  constexpr int RH = DT_Y + 2 * HALO, RWD = DT_X + 2 * HALO, RS = RWD + 1;  // image region
  constexpr int UH = DT_Y + 2 * HK, UW = DT_X + 2 * HK;                      // window halo U
  constexpr int L = dense_lane_px(KW), HL = L / 2;                           // U columns per lane (10 / 12), no overlap
  constexpr int NV = HL + KW - 1;                                            // pixel pairs the windows of a lane reach
  constexpr int DT_HS = dense_h_stride(KW);
  constexpr bool HPAIR = dense_h_paired(KW);
  constexpr int HG = dense_h_group(KW);
  constexpr int NE_MAX = DT_Y * DT_X, NCHUNK = NCH > 0 ? NCH : NE_MAX / 64;
  constexpr bool PIPE = dense_h_paired(KW) && NCH > 0 && NCH <= 2 && !TM;   // pipelined edge stage (see above)
  static_assert(NCH == 0 || (dense_h_paired(KW) && NCH * 64 <= NE_MAX), "chunk classes exist for the 8 x 32 tiles only");
  static_assert(!TM || !HPAIR, "tile-major rows use the k_w 13 pixel map");
  static_assert(UH == 16 && DT_X == 32 && 4 * L >= UW && 3 * HG + L <= DT_HS && HG >= L && (HPAIR || HG == L) && L % 2 == 0 && KW - 1 <= L,
                "lane map: 16 U-rows x 4 column groups (one quad) of L pixels; a window spans two lanes");

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
  int *misc = elist + NE_MAX * 3;          // [16 + 64 NW]: wave counts, n_e (misc[NW]); then the census' (wave, bank) counts and starts

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // tile-major launch order: (slot 0, image 0), (slot 0, image 1), (slot 1, image 0) ... -- the heavy tiles of BOTH images
  // start first (longest jobs first across the whole grid) and the slots behind the list's end drain last
  const int tslot = blockIdx.x / p.nimg, which = blockIdx.x - tslot * p.nimg;
  if (tslot >= dense_tile_count(p.n_dense)) return;
  // (both variants are launched over the whole tile list when the call has a tile-major region; one of them leaves)
  if (p.tm_slots > 0 && tm_active(p.n_dense, p.tm_slots, rows_to_do(p.n_dev, p.n_host)) != TM) return;
  if (TM && p.tm_slots <= 0) return;
  // the plan records the tile height it was cut for (ssg_edge_list's plan_ks): walking an 8-row plan with 4-row
  // tiles (or the reverse) would decode garbage tile ids -- leave, and say so in the device status word (the Python
  // host raises before it gets here, engine.check_plan; C callers ask ssg_device_status())
  if (p.n_dense[1] != DT_Y) {   // (the launch does nothing; ssg_device_status() reports SSG_E_PLAN)
    if (p.status && tid == 0) atomicOr(p.status, 1);
    return;
  }
  const int H = p.H, W = p.W;
  const int tx_n = (W + DT_X - 1) / DT_X, ty_n = (H + DT_Y - 1) / DT_Y;
  const int listed = dense_tile_at(p.n_dense, p.tiles, p.B * ty_n * tx_n, tslot);
  if (TM && p.strips && (listed & TILE_IN_STRIP)) return;   // a strip (ssg_fwd_strip) computes this tile's rows
  if constexpr (NCH > 0) {   // (both chunk classes are launched over the tile list: a tile runs in its own)
    if (((listed & TILE_HUGE) != 0) != (NCH > 2)) return;
  }
  const int tile = dense_tile_id(listed);
  const int b = tile / (tx_n * ty_n), tr = tile - b * tx_n * ty_n;
  const int ty0 = (tr / tx_n) * DT_Y, tx0 = (tr % tx_n) * DT_X;
  const int nrows = rows_to_do(p.n_dev, p.n_host);

  // ---- image region: C x RH x RWD, reflect by index mirroring (clamped: far corners of tiles that overhang a small
  // image are never used, but must stay in bounds).  16 lanes per region row, lane lx takes columns lx, lx + 16, ...
  // (neighbouring lanes read neighbouring pixels).  ALL row passes are loaded here, in front of the census -- its rank
  // load and ballots run while they are in flight -- and stored behind it (round 6; before: two passes at a time
  // behind the census, ~5 L2 round trips in a row at the head of every workgroup) ----
  constexpr int CPLF = (RWD + 15) / 16, RPP = NT / 16, NPASS = (C * RH + RPP - 1) / RPP;
  const int lx = tid % 16, lr = tid / 16;
  // (k_w 9: the thread's rank-map entry FIRST -- loads return in order, so the census below waits for this one only)
  int rank_px = -1;
  if constexpr (HPAIR) {
    const int y = ty0 + tid / DT_X, x = tx0 + tid % DT_X;
    const int v = p.rank[((size_t)b * H + (y < H ? y : H - 1)) * W + (x < W ? x : W - 1)];
    rank_px = (y < H && x < W) ? v : -1;
  }
  float rv[NPASS][CPLF];
  int rdst[NPASS];
  {
    const float *src = p.img[which] + (size_t)b * C * H * W;
#pragma unroll
    for (int h = 0; h < NPASS; ++h) {
      const int R = h * RPP + lr;
      const bool on = R < C * RH;
      const int Rc = on ? R : 0;
      const int c = Rc / RH, ry = Rc - c * RH;
      int gy = reflect_idx(ty0 - HALO + ry, H);
      gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
      const float *srow = src + ((size_t)c * H + gy) * W;
      rdst[h] = on ? (c * RH + ry) * RS : -1;
#pragma unroll
      for (int k = 0; k < CPLF; ++k) {
        int gx = reflect_idx(tx0 - HALO + lx + 16 * k, W);
        gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        rv[h][k] = srow[gx];
      }
    }
  }
  // ---- census of the tile's edge pixels ----
  // List slot e = (chunk of 64, lane) decides which lanes gather together from the H buffers.  k_w 13 (full tiles):
  // row-major, 32 consecutive centres of a row per half-wave, conflict-free.  k_w 9 (a few dozen pixels anywhere in the
  // tile): row-major order puts ~3 pixels of a half-wave on one LDS bank; instead the pixels are counting-sorted by
  // the bank of their H column (ballots per wave, one scan over banks x waves -- LDS atomics cost 0.03 ms here) and
  // dealt round-robin to the half-waves, so that a half-wave holds about one pixel per bank.  The list then has
  // holes (row -1) up to a multiple of 32.
  if constexpr (HPAIR) {
    static_assert(DT_Y * DT_X == NT, "one thread per tile pixel");
    int *wcnt = misc + 16, *bstart = wcnt + NW * 32;  // [NW][32] pixels per (wave, bank); starts in the sorted list
    const int ey = tid / DT_X, ex = tid % DT_X;
    int r = rank_px;
    if (r >= nrows) r = -1;
    elist[3 * tid + 2] = -1;
    const int bank = (DT_HS * ey + dense_h_col(ex, L, HG, true)) & 31;
    // per wave: how many of its pixels fall on each bank (kept by lane `bank`), and each pixel's place among them
    int pib = 0, mycnt = 0;
    for (int bk = 0; bk < 32; ++bk) {
      const unsigned long long m = __ballot(r >= 0 && bank == bk);
      if (bank == bk) pib = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == bk) mycnt = __popcll(m);
    }
    if (lane < 32) wcnt[wv * 32 + lane] = mycnt;
    __syncthreads();
    if (tid < 32) {  // exclusive scan over (bank, wave)
      int c[NW], tot = 0;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        c[k] = wcnt[k * 32 + tid];
        tot += c[k];
      }
      int incl = tot;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        const int t = __shfl_up(incl, o, 32);
        if (tid >= o) incl += t;
      }
      int run = incl - tot;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        bstart[k * 32 + tid] = run;
        run += c[k];
      }
      if (tid == 31) misc[NW] = 32 * ((incl + 31) / 32);   // slots: whole half-waves
    }
    __syncthreads();
    if (r >= 0) {
      const int ng = misc[NW] / 32, i = bstart[wv * 32 + bank] + pib;
      const int pos = (i % ng) * 32 + i / ng;
      elist[3 * pos + 0] = ey;
      elist[3 * pos + 1] = ex;
      elist[3 * pos + 2] = r;
    }
  } else {
    // k_w 13: a fixed map instead of a list -- slot e = 64 p + lane is pixel (ey = 2 (lane / 32) + p, ex = lane % 32),
    // row -1 where that pixel is no edge pixel: a lane's two pixels sit in one column, one row apart, so their
    // vertical windows share 12 of 13 taps and the edge stage reads 14 H values for both (26 before)
    static_assert(HPAIR || (DT_Y == 4 && NCHUNK == 2), "pair map: 4 x 32 tile, two pixels per lane");
    if (tid < NE_MAX) {
      const int pp = tid >> 6, ln = tid & 63;
      const int ey = 2 * (ln >> 5) + pp, ex = ln & 31;
      const int y = ty0 + ey, x = tx0 + ex;
      int r = (y < H && x < W) ? p.rank[((size_t)b * H + y) * W + x] : -1;
      if (r >= nrows) r = -1;
      elist[3 * tid + 0] = ey;
      elist[3 * tid + 1] = ex;
      elist[3 * tid + 2] = r;
    }
    if (tid == 0) misc[NW] = NE_MAX;
  }
  // ---- image region, second half: the values loaded in front of the census go to LDS ----
#pragma unroll
  for (int h = 0; h < NPASS; ++h)
#pragma unroll
    for (int k = 0; k < CPLF; ++k)
      if (rdst[h] >= 0 && lx + 16 * k < RWD) reg[rdst[h] + lx + 16 * k] = rv[h][k];
  __syncthreads();
  const int n_e = misc[NW];  // list slots (k_w 9: with holes, row -1)
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
    HF[ur * DT_HS + dense_h_col(tc, L, HG, HPAIR)] = t;
  }
  __syncthreads();

  // ---- main loop: wave wv takes offset rows qyi = wv, wv+4, ... ----
  const int r = lane >> 2, g = lane & 3;
  // Packed fp32: register pair j of the lane = its pixels (j, j + L/2), so that every operation below is one
  // v_pk_* on two pixels, and the circular window of I[u+q] pairs up the same way whatever the step (slot s and
  // slot s + L/2 always hold two pixels L/2 apart; for s >= L/2 in swapped order, which the packed operand select
  // absorbs).
  f2 iu[C][HL];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const float *q = reg + (c * RH + r + HP) * RS + L * g + HP + j;
      iu[c][j] = f2{q[0], q[HL]};
    }
  // |I|^2 on the pixels the lane's windows reach, paired like the window values V of a truncated step (below):
  // Fv[m] = (F[m], F[m + L/2]), m = 0 .. L/2 + KW - 2, columns counted from the lane's first pixel.  The last
  // group's columns past UW belong to centres outside the tile (never read): clamped.
  f2 Fv[NV];
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    const int c0 = L * g + m, c1 = c0 + HL;
    Fv[m] = f2{F[r * UW + (c0 < UW ? c0 : UW - 1)], F[r * UW + (c1 < UW ? c1 : UW - 1)]};
  }
  float *hb = Hb + wv * UH * DT_HS;
  float *hrow = hb + r * DT_HS + HG * g;  // the lane's centres L*g .. L*g + L-1 (those >= DT_X: written, never read)
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
    eon[ck] = e < n_e && elist[3 * (e < n_e ? e : 0) + 2] >= 0;
    // (k_w 13: the slot's position is the fixed map's whether or not the pixel is an edge pixel -- the lane's two
    // windows are read through hoff[0])
    const int ec = (eon[ck] || !HPAIR) ? e : 0;
    const int ey = elist[3 * ec], ex = elist[3 * ec + 1];
    hoff[ck] = ey * DT_HS + dense_h_col(ex, L, HG, HPAIR);  // window row k of the centre is U-row ey + k
    orow[ck] = eon[ck] ? (size_t)elist[3 * ec + 2] * P : 0;
  }

  const int n_e_stage = SSG_DBG(p, 2) ? 0 : n_e;  // profiling ablations: 2 no edge stage, 1 no stores, 64 no main loop
  const bool do_store = !SSG_DBG(p, 1);
  // Offset rows per wave: KS / NW whole rows each (q_y = wv, wv + NW, ...).  (49,13): 7 x 7.  (25,9): 4 x 6 and ONE row
  // left over, which used to be wave 0's seventh -- 175 steps on SIMD 0 of every CU against 150 on the other three.  The
  // left-over row (q_y = KS - 1) is SHARED instead: wave k takes the q_x steps [RUN[k], RUN[k+1]) of it, run bounds on the
  // bounds of the store groups halved, so that every run leaves as its own stores: 157 / 156 / 156 / 156 steps per wave.
  // A wave enters the unrolled row at its run (window filled for that step) and leaves after it; the shared row's steps
  // are a second instantiation of the step body (SPLIT), without the software pipeline across steps.
  constexpr int SB = (KS == 49 && !TM) ? 7 : 13;   // offsets buffered per store group (see the stores below)
  // (the four-chunk instantiation of the TILE_HUGE tiles is at its 256 registers: it keeps the left-over row on wave 0)
  constexpr int REM = (NCHUNK <= 2) ? KS % NW : 0;
  static_assert(REM == 0 || (REM == 1 && NW == 4 && KS > SB && KS <= 2 * SB), "one shared row, four runs: two store groups halved");
  constexpr int RUN1 = (SB + 1) / 2, RUN3 = SB + (KS - SB) / 2;   // runs [0, RUN1) [RUN1, SB) [SB, RUN3) [RUN3, KS)
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int run_lo = wvu == 0 ? 0 : wvu == 1 ? RUN1 : wvu == 2 ? SB : RUN3;
  const int run_hi = wvu == 0 ? RUN1 : wvu == 1 ? SB : wvu == 2 ? RUN3 : KS;
  // NACT (pipelined two-chunk instantiation of the 8 x 32 tiles only; -1 = decided per step from n_e): chunks of the edge
  // list that hold pixels -- the offset rows exist once per value and the (workgroup-uniform) choice is made ONCE, in front
  // of the row loop: the two `chunk present?` branches per offset step cost more than they look (the dense backward's
  // per-step branches around its lane exchange: 6 % of that kernel, profiles/r6_bwd_dense_ablation.txt)
  auto do_row = [&](const int qyi, auto shared_c, auto nact_c) {
    constexpr bool shared_row = decltype(shared_c)::value;
    constexpr int NACT = decltype(nact_c)::value;
    // D[n,q] = sum_{k in K(q)} E_q[x+k] + sum_{k not in K(q)} |I[x+k]|^2 with K(q) = rows [ylo,yhi] x columns
    // [xlo,xhi] of the window.  Rows: wave-uniform 0/1 weights.  Columns: the lane adds the |I|^2 of the
    // columns that left (compile-time set) to its horizontal sums, so H' rows carry E inside and |I|^2 outside
    // [xlo,xhi]; rows outside [ylo,yhi] contribute their full-width |I|^2 sums (HF), gathered once per q_y.
    const int ylo = (-HK > -qyi) ? -HK : -qyi, yhi = (HK < KS - 1 - qyi) ? HK : KS - 1 - qyi;
    float wgt[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) wgt[k] = (k - HK >= ylo && k - HK <= yhi) ? 1.f : 0.f;
    const bool interior = ylo == -HK && yhi == HK;
    float av[NCHUNK];
#pragma unroll
    for (int ck = 0; ck < NCHUNK; ++ck) {
      av[ck] = 0.f;
      if (ck * 64 < n_e && (ylo > -HK || yhi < HK)) {
        const float *fc = HF + hoff[ck];
        float fv[KW], cw[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) {
          fv[k] = fc[k * DT_HS];
          cw[k] = 1.f - wgt[k];
        }
        av[ck] = tap_sum<KW>(fv, cw, 0.f);
      }
    }
    const float *rq = reg + (r + qyi) * RS + L * g;  // + c*RH*RS + column (i + qxi)
    f2 w[C][HL];  // slots (t, t + L/2)
    // the window as step S finds it: slot a holds pixel (a - S) mod L of the lane, i.e. region column (a - S) mod L + S
    auto fill_window = [&](auto sc) {
      constexpr int S0 = decltype(sc)::value;
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int a = 0; a < L; ++a) {
          const float v = rq[c * RH * RS + ((a - S0) % L + L) % L + S0];
          if (a < HL) w[c][a].x = v;
          else w[c][a - HL].y = v;
        }
    };
    if constexpr (!shared_row) {
      fill_window(std::integral_constant<int, 0>{});
    } else {
      if (run_lo == 0) fill_window(std::integral_constant<int, 0>{});
      else if (run_lo == RUN1) fill_window(std::integral_constant<int, RUN1>{});
      else if (run_lo == SB) fill_window(std::integral_constant<int, SB>{});
      else fill_window(std::integral_constant<int, RUN3>{});
    }
    // consecutive offsets per edge pixel buffered in registers and stored together; the (49,13) row-major variant (the
    // fallback of tile-major calls: masks under 60 % tile fill) keeps 7 (seven groups per offset row) -- with 13 it spilled 10 values per lane, 44 B of scratch
    float evb[TM ? 1 : NCHUNK][TM ? 1 : SB];   // (tile-major rows: nothing is buffered)
    float *tmq = nullptr;                      // tile-major: this wave's offset row, + 64 ck + lane
    if constexpr (TM) tmq = p.tm[which] + ((size_t)tslot * P + (size_t)qyi * KS) * (size_t)NE_MAX + lane;
    float hvp[PIPE ? NCHUNK : 1][KW];   // pipelined edge stage: the taps gathered in the previous step
    auto step = [&](auto qc, auto split_c) {
      constexpr int qxi = decltype(qc)::value;
      constexpr bool SPLIT = decltype(split_c)::value;   // a step of the shared row: one of this wave's run
      constexpr int xlo = (-HK > -qxi) ? -HK : -qxi, xhi = (HK < KS - 1 - qxi) ? HK : KS - 1 - qxi;
      // E_q on the lane's L pixels, E[j] = (pixel j, pixel j + L/2); pixel i of step qxi lives in window slot
      // (i + qxi) % L -- a compile-time function of the step
      // (channel-major: consecutive packed operations belong to different pixel pairs -- with the pair loop outside,
      // every v_pk_fma waited for the one before it, an s_nop each: 14 idle issue slots of ~125 per step)
      // (the four-chunk instantiations are at their register budget: pair-major there, as before)
      f2 E[HL];
      if constexpr (NCHUNK <= 2) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int j = 0; j < HL; ++j) {
            const int a = (j + qxi) % L;  // slot of pixel j; pixel j + L/2 sits in slot (a + L/2) % L
            const f2 wv2 = a < HL ? w[c][a] : w[c][a - HL].yx;
            const f2 d = iu[c][j] - wv2;
            E[j] = c == 0 ? d * d : __builtin_elementwise_fma(d, d, E[j]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < HL; ++j) {
          const int a = (j + qxi) % L;
          f2 t = f2{0.f, 0.f};
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const f2 wv2 = a < HL ? w[c][a] : w[c][a - HL].yx;
            const f2 d = iu[c][j] - wv2;
            t = __builtin_elementwise_fma(d, d, t);
          }
          E[j] = t;
        }
      }
      // horizontal sums for the lane's L centre columns k (window = pixels k .. k+KW-1 of the row: the lane's own
      // from k to L-1, then the next lane's of the quad): E on taps [xlo, xhi], |I|^2 on the others.
      // Hs[k] = (centre k, centre k + L/2).
      f2 Hs[HL];
      if constexpr (xlo == -HK && xhi == HK) {
        // full windows (17 of 25 / 37 of 49 steps): own suffix sum + the next lane's prefix sum.  Prefix and suffix
        // sums of the two half rows come from shared power-of-two blocks of pairs (packed), the sums across the
        // halves and across the lanes are scalar: ~25 VALU instead of L (KW-1) = 80
        float Pf[L], Sf[L];  // Pf[m] = pixels 0..m, Sf[k] = pixels k..L-1 (only the ones used survive)
        static_for(std::make_integer_sequence<int, HL>{}, [&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const f2 pl = sum_from<0, m + 1>(E);    // (pixels 0..m, pixels L/2 .. L/2+m)
          const f2 su = sum_upto<HL, HL - m>(E);  // (pixels m..L/2-1, pixels L/2+m .. L-1)
          const f2 tot = sum_from<0, HL>(E);
          Pf[m] = pl.x;
          Pf[m + HL] = tot.x + pl.y;
          Sf[m + HL] = su.y;
          Sf[m] = su.x + tot.y;
        });
        float hs[L];
        if constexpr (KW == 9 && L == 10) {
          // windows 0 and 1 lie inside the lane; windows 2..9 end in the next lane: Sf[k] + (next lane's Pf[k-2]) as ONE
          // v_add_f32_dpp each (a DPP move and an addition before).  The hazard recogniser does not see into the
          // statement: the s_nop covers the VALU-write -> DPP-read wait states of the Pf registers.
          hs[0] = Pf[KW - 1];
#pragma unroll
          for (int k = 1; k < L; ++k) hs[k] = Sf[k];
          asm volatile("s_nop 1\n\t"
                       "v_add_f32_dpp %0, %8, %0 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_add_f32_dpp %1, %9, %1 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_add_f32_dpp %2, %10, %2 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_add_f32_dpp %3, %11, %3 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_add_f32_dpp %4, %12, %4 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_add_f32_dpp %5, %13, %5 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_add_f32_dpp %6, %14, %6 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                       "v_add_f32_dpp %7, %15, %7 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                       : "+v"(hs[2]), "+v"(hs[3]), "+v"(hs[4]), "+v"(hs[5]), "+v"(hs[6]), "+v"(hs[7]), "+v"(hs[8]), "+v"(hs[9])
                       : "v"(Pf[0]), "v"(Pf[1]), "v"(Pf[2]), "v"(Pf[3]), "v"(Pf[4]), "v"(Pf[5]), "v"(Pf[6]), "v"(Pf[7]));
        } else {
        static_for(std::make_integer_sequence<int, L>{}, [&](auto kc) {
          constexpr int k = decltype(kc)::value, last = k + KW - 1;  // window [k, last] in the lane's own coordinates
          if constexpr (last < L) {
            static_assert(k == 0 || last == L - 1, "in-lane windows: a prefix or a suffix");
            hs[k] = k == 0 ? Pf[last] : Sf[k];
          } else {
            hs[k] = Sf[k] + quad_next<1>(Pf[last - L]);
          }
        });
        }
#pragma unroll
        for (int k = 0; k < HL; ++k) Hs[k] = f2{hs[k], hs[k + HL]};
      } else {
        // truncated windows: V[m] = (value m, value m + L/2) of the row continued into the next lane, value = E on
        // the taps kept and |I|^2 (Fv) on the others; Hs[k] = sum_t V[k + t]
        f2 V[NV];
#pragma unroll
        for (int m = 0; m < NV; ++m) {
          if (m < HL) V[m] = E[m];
          else if (m < L) V[m] = f2{E[m - HL].y, quad_next<1>(E[m - HL].x)};
          else V[m] = f2{quad_next<1>(E[m - L].x), quad_next<1>(E[m - L].y)};
        }
#pragma unroll
        for (int k = 0; k < HL; ++k) {
          f2 t = f2{0.f, 0.f};
          bool first = true;
#pragma unroll
          for (int tp = 0; tp < KW; ++tp) {
            const bool kept = tp - HK >= xlo && tp - HK <= xhi;
            const f2 v = kept ? V[k + tp] : Fv[k + tp];
            t = first ? v : t + v;
            first = false;
          }
          Hs[k] = t;
        }
      }
      if constexpr (HPAIR) {
#pragma unroll
        for (int k = 0; k < HL; ++k) *(f2 *)(hrow + 2 * k) = Hs[k];
      } else {
#pragma unroll
        for (int k = 0; k < HL; ++k) {
          hrow[k] = Hs[k].x;
          hrow[k + HL] = Hs[k].y;
        }
      }
      // next q_x: the pixel that leaves the window (slot qxi % L) is replaced by the one that enters
      if (qxi + 1 < KS) {
        constexpr int sl = qxi % L;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float v = rq[c * RH * RS + L + qxi];
          if constexpr (sl < HL) w[c][sl].x = v;
          else w[c][sl - HL].y = v;
        }
      }
      // (the H buffer is private to this wave and LDS operations of one wave execute in issue order: the reads
      // below see the writes above, and the next step's writes cannot overtake them -- no wait, no barrier)
      // ---- the tile's edge pixels: weighted vertical taps, row complement, exp, store ----
      float dpair[2];
      if constexpr (!HPAIR) {
        if (n_e_stage > 0) {
          // the lane's two pixels (rows ey0, ey0 + 1 of one column): 14 H values serve both windows
          const float *hc = hb + hoff[0];
          float hv[KW + 1];
#pragma unroll
          for (int k = 0; k <= KW; ++k) hv[k] = hc[k * DT_HS];
          if (interior) {   // every tap kept, no complement: the 12 shared taps are summed once
            f2 t01 = f2{hv[2], hv[3]} + f2{hv[4], hv[5]}, t23 = f2{hv[6], hv[7]} + f2{hv[8], hv[9]};
            t01 = t01 + f2{hv[10], hv[11]};
            t01 = t01 + t23;
            const float common = (t01.x + t01.y) + (hv[1] + hv[KW - 1]);
            dpair[0] = common + hv[0];
            dpair[1] = common + hv[KW];
          } else {
            float h0[KW], h1[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) {
              h0[k] = hv[k];
              h1[k] = hv[k + 1];
            }
            dpair[0] = tap_sum<KW>(h0, wgt, av[0]);
            dpair[1] = tap_sum<KW>(h1, wgt, av[1]);
          }
        }
      }
      // (the taps of one chunk and what becomes of their sum; QE = the offset the value belongs to)
      auto edge_gather = [&](int ck, float (&hv)[KW]) {
        const float *hc = hb + hoff[ck];
#pragma unroll
        for (int k = 0; k < KW; ++k) hv[k] = hc[k * DT_HS];
      };
      auto edge_emit = [&](auto qe_c, int ck, float d) {
        constexpr int QE = decltype(qe_c)::value;
          float ev;
          if constexpr (RAW) {
            ev = d;
          } else {
            ev = __builtin_amdgcn_exp2f(d * nk);
            rs[ck] += (double)ev;
            asm volatile("" : "+v"(rs[ck]));   // (pinned: the sink pass otherwise moves these additions to the end of the row and keeps every e for them)
          }
          if constexpr (TM) {
            // tile-major scratch rows: the 64 lanes' values of one offset are one aligned 256-byte run (holes
            // included: the consumers skip them by the rank map)
            if (do_store) tmq[QE * NE_MAX + ck * 64] = ev;
          } else {
          // every lane writes its own SSG row: single dwords are one L2 request per lane and step (request-
          // rate bound at high density), so SB consecutive offsets leave together as 16-byte stores
          evb[ck][QE % SB] = ev;
          if (eon[ck] && do_store) {
            float *o = outp + orow[ck] + qyi * KS;
            // the group (regular rows) or the run (shared row) that ends with this offset: first offset q0, cnt of them
            constexpr bool ends = SPLIT ? (QE + 1 == RUN1 || QE + 1 == SB || QE + 1 == RUN3 || QE + 1 == KS)
                                        : (QE % SB == SB - 1 || QE == KS - 1);
            if constexpr (ends) {
              constexpr int q0 = !SPLIT ? QE - QE % SB : (QE + 1 == RUN1 ? 0 : QE + 1 == SB ? RUN1 : QE + 1 == RUN3 ? SB : RUN3);
              constexpr int cnt = QE + 1 - q0, i0 = q0 % SB;
#pragma unroll
              for (int t = 0; t + 4 <= cnt; t += 4) {
                float4 v4 = make_float4(evb[ck][i0 + t], evb[ck][i0 + t + 1], evb[ck][i0 + t + 2], evb[ck][i0 + t + 3]);
                if constexpr (RAW) {   // out += D: the row's own lane is the only writer of these words
                  float4 old;
                  __builtin_memcpy(&old, o + q0 + t, 16);
                  v4.x += old.x; v4.y += old.y; v4.z += old.z; v4.w += old.w;
                }
                __builtin_memcpy(o + q0 + t, &v4, 16);
              }
#pragma unroll
              for (int t = cnt & ~3; t < cnt; ++t) {
                if constexpr (RAW) o[q0 + t] += evb[ck][i0 + t];
                else o[q0 + t] = evb[ck][i0 + t];
              }
            }
          }
          }
      };
#pragma unroll
      for (int ck = 0; ck < (NACT >= 0 ? NACT : NCHUNK); ++ck) {
        if (NACT >= 0 || ck * 64 < n_e_stage) {
          if constexpr (PIPE && !SPLIT) {
            // consume the taps gathered a step ago (offset qxi - 1), then gather this step's behind its H stores
            if constexpr (qxi > 0) edge_emit(std::integral_constant<int, qxi - 1>{}, ck, tap_sum<KW>(hvp[ck], wgt, av[ck]));
            edge_gather(ck, hvp[ck]);
            if constexpr (qxi == KS - 1) edge_emit(std::integral_constant<int, KS - 1>{}, ck, tap_sum<KW>(hvp[ck], wgt, av[ck]));
          } else if constexpr (HPAIR) {
            float hv[KW];
            edge_gather(ck, hv);
            edge_emit(std::integral_constant<int, qxi>{}, ck, tap_sum<KW>(hv, wgt, av[ck]));
          } else {
            edge_emit(std::integral_constant<int, qxi>{}, ck, dpair[ck]);
          }
        }
      }
    };
    if constexpr (!shared_row) {
      static_for(std::make_integer_sequence<int, KS>{}, [&](auto qc) { step(qc, std::false_type{}); });
    } else {
      static_for(std::make_integer_sequence<int, KS>{}, [&](auto qc) {
        constexpr int qxi = decltype(qc)::value;
        if (qxi >= run_lo && qxi < run_hi) step(qc, std::true_type{});   // (wave-uniform: scalar branches)
      });
    }
  };
  if (!SSG_DBG(p, 64)) {
#define SSG_ALL_ROWS(NACT_)                                                                                    \
  do {                                                                                                         \
    _Pragma("unroll 1") for (int qyi = wv; qyi < KS - REM; qyi += NW)                                          \
        do_row(qyi, std::false_type{}, std::integral_constant<int, NACT_>{});                                  \
    if constexpr (REM > 0) do_row(KS - 1, std::true_type{}, std::integral_constant<int, NACT_>{});             \
  } while (0)
    if constexpr (PIPE && NCHUNK == 2 && !RAW) {   // (the raw-distance instantiation spills with two row bodies: per-step test)
      if (n_e_stage > 64) SSG_ALL_ROWS(2);
      else if (n_e_stage > 0) SSG_ALL_ROWS(1);
#ifdef SSG_PROFILE
      else SSG_ALL_ROWS(0);   // (ablation: no edge stage)
#endif
    } else {
      SSG_ALL_ROWS(-1);
    }
#undef SSG_ALL_ROWS
  }

  // ---- row sums over the four waves, then rescale the rows this workgroup wrote ----
#pragma unroll
  for (int ck = 0; ck < NCHUNK; ++ck)
    if (ck * 64 + lane < n_e) rsum[wv * RSTR + ck * 64 + lane] = rs[ck];
  __threadfence_block();
  __syncthreads();
  if (RAW || !p.generalization || SSG_DBG(p, 8)) return;
  if (p.row_scale) {
    // deferred normalisation: the consumer that streams the rows anyway (ssg_grad_rows) rescales them; saves this
    // kernel's second pass over its rows (one read + one write of every row)
    double *rsc = p.row_scale + (size_t)which * p.n_host;
    for (int e = tid; e < n_e; e += NT) {
      if (elist[3 * e + 2] < 0) continue;
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < NW; ++k) tot += rsum[k * RSTR + e];
      const double sc = 1.0 / (tot + (double)p.eps);
      rsc[elist[3 * e + 2]] = TM ? -sc : sc;   // (negative: "this row lives in the tile-major region")
    }
    return;
  }
  if constexpr (TM) return;   // (tile-major rows exist in the deferred form only)
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
    bool ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = e0 + j < n_e ? e0 + j : e0;
      ok[j] = e0 + j < n_e && elist[3 * e + 2] >= 0;
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < NW; ++k) tot += rsum[k * RSTR + e];
      scale[j] = 1.0 / (tot + (double)p.eps);
      o[j] = outp + (size_t)(ok[j] ? elist[3 * e + 2] : 0) * P;
#pragma unroll
      for (int k = 0; k < RPL; ++k) {
        const int q = lane + 64 * k;
        v[j][k] = q < P ? __builtin_nontemporal_load(o[j] + q) : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
          const int q = lane + 64 * k;
          if (q < P) o[j][q] = (float)(scale[j] * (double)v[j][k]);
        }
      }
    }
  }
}

template <int KS, int KW, int C, int NW, bool TM = false, bool RAW = false, int NCH = 0>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void ssg_fwd_dense(DenseParams p) {
  fwd_dense_body<KS, KW, C, NW, TM, RAW, NCH>(p);
}

// (25,9): BOTH chunk classes of the 8 x 32 tiles in one launch (round 6; see ssg_bwd_dense_classes): the workgroup reads its
// tile's class bit from the plan and runs that instantiation's body.  Both run at two waves per SIMD, so the merged kernel
// costs no occupancy; C2 / C4 hold no TILE_HUGE tile and lose a launch of workgroups that start only to leave.
template <int KS, int KW, int C, int NW, bool RAW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void ssg_fwd_dense_classes(DenseParams p) {
  const int tslot = blockIdx.x / p.nimg;
  if (tslot >= dense_tile_count(p.n_dense)) return;
  const int tx_n = (p.W + DT_X - 1) / DT_X, ty_n = (p.H + 7) / 8;
  const int listed = dense_tile_at(p.n_dense, p.tiles, p.B * ty_n * tx_n, tslot);
  if (__builtin_amdgcn_readfirstlane(listed) & TILE_HUGE) fwd_dense_body<KS, KW, C, NW, false, RAW, 4>(p);
  else fwd_dense_body<KS, KW, C, NW, false, RAW, 2>(p);
}


// ------------------------------------------------------------------ strips (k_s 49, tile-major rows) ----
// The tile kernel above computes E_q and its horizontal sums on 16 U-rows to serve DT_Y = 16 - (k_w - 1) centre rows:
// at k_w 13 that is 4 rows of 16, a four-fold overhead on the part of the kernel that does not depend on the number of
// edge pixels -- and C5 is a 100 % mask.  A strip is SY = 16 NW - (k_w - 1) centre rows x 32 columns (36 rows for NW =
// 3): NW waves own 16 U-rows each (the lane map, the window and the horizontal sums are the tile kernel's), walk ALL
// k_s^2 offsets in lockstep and exchange the horizontal sums through a double-buffered H array (one s_barrier per
// offset): the overhead falls to 48 / 36.  Every lane owns PX = 6 centres of one column (rows 6 seg .. 6 seg + 5) for
// the whole kernel: their 13-tap vertical windows share 18 H values (a tree of 48 additions for the six sums) and the
// row sums stay in the lane's registers (no reduction over waves).
// The rows go to the TILE-MAJOR region only (round 3 measured the same kernel on row-major rows: compute 2.0 ms against
// the tile kernel's 2.9 at C5, but 5.2 ms with 1,152 row-major rows open per workgroup): the strip's tiles hold
// consecutive slots of the region (strip_select, ssg_edges.hip), a half-wave's 32 centres of one tile row are one
// aligned 128-byte run of slot (first + row / 4) at every offset, holes included.
// The image region does not fit LDS for 36 + 2 HALO rows and is not needed at once: offset row q_y touches region
// rows q_y .. q_y + 47 only, a BAND of 48 rows kept as a ring (region row rho at slot rho % 48); one new row per q_y
// is fetched at the top of the q_y loop and stored after its last step.
//
// Round 5: the two stages of an offset run on DIFFERENT waves (the recipe of the role-split dense backward).  A workgroup
// is 2 NW waves: waves 0 .. NW-1 ("E/H role") own 16 U-rows each and do nothing but the E/H stage -- window, E, horizontal
// sums, H rows --, waves NW .. 2 NW - 1 ("edge role") own the 6 centres per lane and do nothing but the edge stage -- H
// gathers, vertical sums, exp, fp64 row sums, stores.  They meet at the one s_barrier per offset the kernel always had
// (E/H of offset t+1 beside the edge stage of t, double-buffered H).  Each role keeps only its own registers (the
// one-role-per-wave kernel needed 238 VGPRs -> 2 waves per SIMD, 1.3 resident: issue-limited at one instruction per ~5
// cycles and wave), so the kernel fits 3 waves per SIMD: two workgroups = 12 waves per CU instead of 6.
#ifdef SSG_PROFILE
__device__ unsigned long long g_strip_times[3 * 1024];   // per workgroup: start, end (s_memtime), XCC id << 8 | CU id
#endif
template <int KS, int KW, int C, int NW>
__global__ __launch_bounds__(256 * NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void ssg_fwd_strip(DenseParams p) {
#ifdef SSG_PROFILE
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_strip_times[3 * blockIdx.x] = __builtin_readcyclecounter();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_strip_times[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
  constexpr int NT = 128 * NW, NTR = 64 * NW;                     // threads of the workgroup / of one role
  constexpr int HP = KS / 2, HK = KW / 2, P = KS * KS, HALO = HP + HK;
  constexpr int UH = 16 * NW, SY = UH - 2 * HK;                  // U-rows and centre rows of a strip
  constexpr int RWD = DT_X + 2 * HALO, RS = RWD + 1;             // band row
  constexpr int UW = DT_X + 2 * HK;
  constexpr int L = dense_lane_px(KW), HL = L / 2, NV = HL + KW - 1;
  constexpr int HS = DT_X + 1;                                   // H row stride: gathers AND the dword stores conflict-free
  constexpr int PX = SY * DT_X / NTR, NHV = PX + KW - 1;          // centres per lane (one column), H values they share
  static_assert(SY == STRIP_ROWS && SY * DT_X == PX * NTR && NTR % DT_X == 0 && (NTR / DT_X) * PX == SY && 4 * L >= UW && L % 2 == 0 &&
                KW - 1 <= L && KW == 13 && PX == 6, "lane maps: 16 U-rows per wave, 6 centres of a column per lane");

  // TWO strips per workgroup (4 NW waves, each strip its own half of the LDS): measured (tools/r5_strip_times.py,
  // tools/microbench_lds_residency.hip) the CU does not admit a second 2 NW-wave workgroup of this kernel although LDS,
  // registers and the occupancy API allow it -- its 6 waves at 3 per SIMD only fit beside another 6 in one placement --
  // so the strips ran in two rounds, one workgroup per CU.  One 12-wave workgroup fills the four SIMDs with 3 waves each.
  constexpr int STRIP_LDS = C * UH * RS + 3 * UH * HS + UH * UW + 16;   // floats of one strip (launch_fwd_strip_t)
  extern __shared__ __attribute__((aligned(16))) float smem_wg[];
  const int half = threadIdx.x / NT;
  float *smem = smem_wg + half * STRIP_LDS;
  float *band = smem;                 // [C][UH][RS]  region rows q_y .. q_y + UH - 1, row rho at slot rho % UH
  float *HF = band + C * UH * RS;     // [UH][HS]     full-window horizontal sums of |I|^2
  float *Hb = HF + UH * HS;           // [2][UH][HS]  horizontal sums of E_q, double-buffered over the steps
  float *F = Hb + 2 * UH * HS;        // [UH][UW]     |I|^2 on U (the columns a truncated window leaves behind)

  const int tid = threadIdx.x - half * NT, lane = tid & 63, wv = tid >> 6;   // thread / wave within the strip
  // jobs = (image, strip) pairs of the call, two per workgroup; an odd last job is mirrored by the idle half (same
  // values to the same addresses) so that both halves run the same barriers
  const int n_act = p.strips[0] < p.max_strips ? p.strips[0] : p.max_strips, n_jobs = n_act * p.nimg;
  if (2 * (int)blockIdx.x >= n_jobs) return;
  const int job = 2 * (int)blockIdx.x + half < n_jobs ? 2 * (int)blockIdx.x + half : n_jobs - 1;
  const int which = job / n_act, slot = job - which * n_act;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  if (!tm_active(p.n_dense, p.tm_slots, nrows)) return;   // (row-major call: the tile kernel computes these tiles)
  const int H = p.H, W = p.W;
  const int tx_n = (W + DT_X - 1) / DT_X, sy_n = (H + SY - 1) / SY;
  const int strip = p.strips[1 + 2 * slot], first = p.strips[2 + 2 * slot];
  const int b = strip / (tx_n * sy_n), tr = strip - b * tx_n * sy_n;
  const int ty0 = (tr / tx_n) * SY, tx0 = (tr % tx_n) * DT_X;
  const float *src = p.img[which] + (size_t)b * C * H * W;
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // region rows rho0 .. rho0 + UH - 1 into the band (reflect by index mirroring, clamped like the tile kernel's)
  auto fill_band = [&](int rho0) {
    constexpr int CPLF = (RWD + 15) / 16, RPP = NT / 16;
    const int lx = tid % 16, lr = tid / 16;
    for (int R0 = 0; R0 < C * UH; R0 += RPP) {
      const int R = R0 + lr;
      const int c = R / UH, i = R - c * UH, rho = rho0 + i;
      int gy = reflect_idx(ty0 - HALO + rho, H);
      gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
      const float *srow = src + ((size_t)c * H + gy) * W;
      float v[CPLF];
#pragma unroll
      for (int k = 0; k < CPLF; ++k) {
        int gx = reflect_idx(tx0 - HALO + lx + 16 * k, W);
        gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        v[k] = srow[gx];
      }
      float *drow = band + (c * UH + rho % UH) * RS;
#pragma unroll
      for (int k = 0; k < CPLF; ++k)
        if (lx + 16 * k < RWD) drow[lx + 16 * k] = v[k];
    }
  };
  static_assert((C * UH) % (NT / 16) == 0, "whole row passes");

  // ---- set-up on the U rows themselves (region rows HP .. HP + UH - 1): own pixels, |I|^2 and its sums ----
  fill_band(HP);
  __syncthreads();
  const bool eh_role = wv < NW;                              // (wave-uniform)
  const int r = lane >> 2, g = lane & 3, R = 16 * (eh_role ? wv : 0) + r;   // E/H role: U-row of the lane
  f2 iu[C][HL];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const float *q = band + (c * UH + (R + HP) % UH) * RS + L * g + HP + j;
      iu[c][j] = f2{q[0], q[HL]};
    }
  for (int i = tid; i < UH * UW; i += NT) {
    const int ur = i / UW, uc = i - ur * UW;
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float v = band[(c * UH + (ur + HP) % UH) * RS + uc + HP];
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
    HF[ur * HS + tc] = t;
  }
  // |I|^2 on the pixels the lane's windows reach, Fv(m) = (F[m], F[m + L/2]) counted from the lane's first pixel
  // (the tile kernel keeps these 18 pairs in registers; here they are read when a truncated step needs them: 12 of
  // 49 steps).  The last group's columns past UW belong to centres outside the strip: clamped to the row.
  const int frow_off = R * UW + (g < 3 ? L * g : UW - NV - HL);
  __syncthreads();
  fill_band(0);

  // ---- the lane's PX centres: rows e0 .. e0 + PX - 1 of column ecol ----
  // Centre j lives in the strip's tile (e0 + j) / 4 = slot first + (e0 + j) / 4 of the tile-major region, at pixel
  // index 64 (ey & 1) + 32 (ey >> 1) + column, ey = (e0 + j) % 4 (tm_pixel_row / tm_pixel_col inverted): a uniform
  // base per offset + one 32-bit element offset per centre.
  // A strip at the bottom of the image has fewer than nine tiles: the centres below them go to the region's spare
  // slot (index tm_slots, behind the last real one: written by every short strip, read by nobody).
  const int tid_e = eh_role ? 0 : tid - NTR;                  // edge role: thread within the role
  const int ecol = tid_e % DT_X, e0 = (tid_e / DT_X) * PX;
  const int ty_n = (H + 3) / 4, n_tq = ty_n - ty0 / 4 < SY / 4 ? ty_n - ty0 / 4 : SY / 4;
  int orow[PX];
  unsigned coff[PX];   // ELEMENT offset of the centre's pixel in the tile-major region: (slots + 1) * P * 128 < 2^32 (16 GB per
                       // image: ssg_api.hip keeps larger calls on row-major rows)
  // (wave-uniform, and told so: `which` comes from the thread index through the strip pairing, and only a base the
  // compiler knows to live in SGPRs gives `global_store_dword v_off, v_data, s[base]`)
  float *const tmbase = __builtin_amdgcn_readfirstlane(which) ? p.tm[1] : p.tm[0];
#pragma unroll
  for (int j = 0; j < PX; ++j) {
    const int y = ty0 + e0 + j, x = tx0 + ecol;
    const int rk = (y < H && x < W) ? p.rank[((size_t)b * H + y) * W + x] : -1;
    orow[j] = rk >= nrows ? -1 : rk;
    const int tq = (e0 + j) >> 2, ey = (e0 + j) & 3;
    coff[j] = (unsigned)(tq < n_tq ? first + tq : p.tm_slots) * (unsigned)(P * TM_PX) + (unsigned)(64 * (ey & 1) + 32 * (ey >> 1) + ecol);
  }
  int pf_gx = reflect_idx(tx0 - HALO + (tid < RWD ? tid : 0), W);   // column of the band row this thread fetches
  pf_gx = pf_gx < 0 ? 0 : (pf_gx >= W ? W - 1 : pf_gx);
  const float nk = (float)(-1.4426950408889634 / ((double)(C * KW * KW) * (double)p.sigma));
  double rs[PX];
#pragma unroll
  for (int j = 0; j < PX; ++j) rs[j] = 0.0;
  float *hwrite = Hb + R * HS + L * g;          // the lane's centres L g .. L g + L - 1 (those < DT_X are stored)
  // centres k < HSPLIT of a lane are inside the strip for column groups g <= GA, the others for g <= GB
  constexpr int HSPLIT = DT_X - 2 * L, GA = 2, GB = 1;
  static_assert(L == 12 && DT_X == 32 && HSPLIT == 8, "groups 0, 1 whole, group 2 its first eight centres, group 3 none");
  const float *hread = Hb + e0 * HS + ecol;     // window row k of centre j is U-row e0 + j + k
  const float *hfread = HF + e0 * HS + ecol;
  __syncthreads();

  // The two roles run the same number of barriers per offset row: 1 (offset 0's H) + KS - 1 (one per further offset) +
  // 1 (band row stored; not after the last row).
  // (static priority for the E/H role, whose step is the longer one: -1.3 % same-box; the edge role at priority: +1.8 %)
  if (eh_role) __builtin_amdgcn_s_setprio(1);
  if (eh_role) {
#pragma unroll 1
  for (int qyi = 0; qyi < KS; ++qyi) {
    // the band row the NEXT q_y needs (region row qyi + UH), fetched now, stored after this row's last step
    float pf[C];
    if (qyi + 1 < KS && tid < RWD) {   // one column per thread, C uniform row pointers
      int gy = reflect_idx(ty0 - HALO + qyi + UH, H);
      gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
#pragma unroll
      for (int c = 0; c < C; ++c) pf[c] = src[((size_t)c * H + gy) * W + pf_gx];
    }
    const float *rq = band + ((R + qyi) % UH) * RS + L * g;  // + c*UH*RS + column
    f2 w[C][HL];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < HL; ++t) w[c][t] = f2{rq[c * UH * RS + t], rq[c * UH * RS + t + HL]};
    // step (qyi, qxi) writes H buffer (qyi + qxi) % 2 (k_s is odd: the parity alternates across rows too)
    float *hw_even = hwrite + (qyi & 1) * UH * HS, *hw_odd = hwrite + ((qyi & 1) ^ 1) * UH * HS;
    // One offset = an E/H stage (E_q on the lane's pixels, horizontal sums, H rows to this step's buffer, window moved
    // on) and an edge stage (the 18 H values of the lane's six centres, vertical sums, exp, row sums, stores).  They
    // are software-pipelined: a step runs the E/H stage of the NEXT offset, then the edge stage of its own -- whose H
    // values were requested right behind the previous barrier and arrive under the E/H arithmetic -- then the barrier
    // (everybody's H rows of the next offset are in LDS, everybody's reads of this offset's buffer are done) and the
    // requests for the next offset.  One s_barrier per offset as before; the LDS round trips leave the critical path.
    auto eh_stage = [&](auto qc) {
      constexpr int qxi = decltype(qc)::value;
      constexpr int xlo = (-HK > -qxi) ? -HK : -qxi, xhi = (HK < KS - 1 - qxi) ? HK : KS - 1 - qxi;
      f2 E[HL];
#pragma unroll
      for (int j = 0; j < HL; ++j) {
        const int a = (j + qxi) % L;
        f2 t = f2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const f2 wv2 = a < HL ? w[c][a] : w[c][a - HL].yx;
          const f2 d = iu[c][j] - wv2;
          t = __builtin_elementwise_fma(d, d, t);
        }
        E[j] = t;
      }
      float hs[L];
      if constexpr (xlo == -HK && xhi == HK) {
        float Pf[L], Sf[L];
        static_for(std::make_integer_sequence<int, HL>{}, [&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const f2 pl = sum_from<0, m + 1>(E);
          const f2 su = sum_upto<HL, HL - m>(E);
          const f2 tot = sum_from<0, HL>(E);
          Pf[m] = pl.x;
          Pf[m + HL] = tot.x + pl.y;
          Sf[m + HL] = su.y;
          Sf[m] = su.x + tot.y;
        });
        // every window of this lane map ends in the next lane (k + 12 >= L): hs[k] = Sf[k] + (next lane's Pf[k]) as ONE
        // v_add_f32_dpp each instead of a DPP move and an addition -- these waves pay per instruction (DESIGN section
        // 4).  The hazard recogniser does not see into the statement: the s_nop covers the VALU-write -> DPP-read wait
        // states of the Pf registers.
        static_assert(KW - 1 >= L && L == 12, "all twelve windows reach into the next lane");
#pragma unroll
        for (int k = 0; k < L; ++k) hs[k] = Sf[k];
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %6, %0 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %1, %7, %1 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %2, %8, %2 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %3, %9, %3 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %4, %10, %4 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %5, %11, %5 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(hs[0]), "+v"(hs[1]), "+v"(hs[2]), "+v"(hs[3]), "+v"(hs[4]), "+v"(hs[5])
                     : "v"(Pf[0]), "v"(Pf[1]), "v"(Pf[2]), "v"(Pf[3]), "v"(Pf[4]), "v"(Pf[5]));
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %6, %0 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %1, %7, %1 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %2, %8, %2 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %3, %9, %3 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %4, %10, %4 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_add_f32_dpp %5, %11, %5 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(hs[6]), "+v"(hs[7]), "+v"(hs[8]), "+v"(hs[9]), "+v"(hs[10]), "+v"(hs[11])
                     : "v"(Pf[6]), "v"(Pf[7]), "v"(Pf[8]), "v"(Pf[9]), "v"(Pf[10]), "v"(Pf[11]));
      } else {
        // (|I|^2 does not depend on q_y: left alone, the compiler hoists the sums of the taps left behind out of the
        // q_y loop -- 144 registers)
        int fo = frow_off;
        asm volatile("" : "+v"(fo));
        const float *frow = F + fo;
        f2 V[NV];
#pragma unroll
        for (int m = 0; m < NV; ++m) {
          if (m < HL) V[m] = E[m];
          else if (m < L) V[m] = f2{E[m - HL].y, quad_next<1>(E[m - HL].x)};
          else V[m] = f2{quad_next<1>(E[m - L].x), quad_next<1>(E[m - L].y)};
        }
#pragma unroll
        for (int k = 0; k < HL; ++k) {
          f2 t = f2{0.f, 0.f};
          bool first_tap = true;
#pragma unroll
          for (int tp = 0; tp < KW; ++tp) {
            const bool kept = tp - HK >= xlo && tp - HK <= xhi;
            const f2 v = kept ? V[k + tp] : f2{frow[k + tp], frow[k + tp + HL]};
            t = first_tap ? v : t + v;
            first_tap = false;
          }
          hs[k] = t.x;
          hs[k + HL] = t.y;
        }
      }
      float *hw = qxi % 2 == 0 ? hw_even : hw_odd;
      if (SSG_DBG(p, 32)) {   // (profiling: no H stores)
#pragma unroll
        for (int k = 0; k < L; ++k) asm volatile("" ::"v"(hs[k]));
      } else {
        // (round 5: predicated -- the lanes outside the strip used to store to a dummy word instead, and that word's bank
        // cost every store pass a conflict cycle: 72 of the 134 conflict cycles of a workgroup-step)
        if (g <= GA) {
#pragma unroll
          for (int k = 0; k < HSPLIT; ++k) hw[k] = hs[k];
        }
        if (g <= GB) {
#pragma unroll
          for (int k = HSPLIT; k < L; ++k) hw[k] = hs[k];
        }
      }
      if (qxi + 1 < KS && !SSG_DBG(p, 128)) {
        constexpr int sl = qxi % L;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float v = rq[c * UH * RS + L + qxi];
          if constexpr (sl < HL) w[c][sl].x = v;
          else w[c][sl - HL].y = v;
        }
      }
    };
    if (!SSG_DBG(p, 4)) eh_stage(std::integral_constant<int, 0>{});
    lds_barrier();
    static_for(std::make_integer_sequence<int, KS - 1>{}, [&](auto qc) {
      if (!SSG_DBG(p, 4)) eh_stage(std::integral_constant<int, decltype(qc)::value + 1>{});
      lds_barrier();
    });
    if (qyi + 1 < KS) {
      // (the last step's barrier is behind every wave's last read of region row qyi, whose slot this is)
      if (tid < RWD) {
#pragma unroll
        for (int c = 0; c < C; ++c) band[(c * UH + qyi % UH) * RS + tid] = pf[c];
      }
      lds_barrier();
    }
  }
  } else {
#pragma unroll 1
  for (int qyi = 0; qyi < KS; ++qyi) {
    const int ylo = (-HK > -qyi) ? -HK : -qyi, yhi = (HK < KS - 1 - qyi) ? HK : KS - 1 - qyi;
    float wgt[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k)
      wgt[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((k - HK >= ylo && k - HK <= yhi) ? 0x3f800000 : 0));
    const bool interior = ylo == -HK && yhi == HK;
    float av[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) av[j] = 0.f;
    if (!interior) {
      float hf[NHV], cw[KW];
#pragma unroll
      for (int t = 0; t < NHV; ++t) hf[t] = hfread[t * HS];
#pragma unroll
      for (int k = 0; k < KW; ++k) cw[k] = 1.f - wgt[k];
#pragma unroll
      for (int j = 0; j < PX; ++j) {
        float fv[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) fv[k] = hf[j + k];
        av[j] = tap_sum<KW>(fv, cw, 0.f);
      }
    }
    const float *hr_even = hread + (qyi & 1) * UH * HS, *hr_odd = hread + ((qyi & 1) ^ 1) * UH * HS;
    // the 18 H values of the lane's six centres as nine register pairs P[m] = (H[2m], H[2m+1]): the vertical sums
    // below are packed operations on them -- these waves are issue-limited (one VALU slot per 4 cycles, DESIGN
    // section 4), so two additions per slot count
    static_assert(NHV == 18, "nine pairs");
    f2 P[NHV / 2];
    auto h_requests = [&](auto qc) {
      constexpr int qxi = decltype(qc)::value;
      const float *hc = qxi % 2 == 0 ? hr_even : hr_odd;
      if (SSG_DBG(p, 16)) return;   // (profiling: no gathers)
#pragma unroll
      for (int m = 0; m < NHV / 2; ++m) P[m] = f2{hc[2 * m * HS], hc[(2 * m + 1) * HS]};
    };
    auto edge_stage = [&](auto qc) {
      constexpr int qxi = decltype(qc)::value;
      float d[PX];
      if (interior) {
        // all 13 taps kept.  Window j = H[j] + ... + H[j+12]; with T[m] = P[m] + P[m+1] and Q[a] = T[a] + T[a+2] + T[a+4]
        // (= P[a] + ... + P[a+5], packed), S[a] = Q[a].x + Q[a].y is H[2a] + ... + H[2a+11]:
        //   window 2a = S[a] + H[2a+12],  window 2a+1 = H[2a+1] + S[a+1]          (16 packed + 10 scalar additions)
        f2 T[8], Q[4];
#pragma unroll
        for (int m = 0; m < 8; ++m) T[m] = P[m] + P[m + 1];
#pragma unroll
        for (int a = 0; a < 4; ++a) Q[a] = (T[a] + T[a + 2]) + T[a + 4];
        float S[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) S[a] = Q[a].x + Q[a].y;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          d[2 * a] = S[a] + P[a + 6].x;
          d[2 * a + 1] = P[a].y + S[a + 1];
        }
      } else {
        // truncated rows: 0/1 weights (wave-uniform) as pairs; window 2a takes (w[2m], w[2m+1]) on P[a+m], m = 0..5,
        // and w[12] on H[2a+12]; window 2a+1 takes w[0] on H[2a+1] and (w[2m-1], w[2m]) on P[a+m], m = 1..6
        f2 we[6], wo[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          we[m] = f2{wgt[2 * m], wgt[2 * m + 1]};
          wo[m] = f2{wgt[2 * m + 1], wgt[2 * m + 2]};
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          f2 e0 = P[a] * we[0], e1 = P[a + 1] * we[1];
          f2 o0 = P[a + 1] * wo[0], o1 = P[a + 2] * wo[1];
#pragma unroll
          for (int m = 2; m < 6; ++m) {
            if (m % 2 == 0) {
              e0 = __builtin_elementwise_fma(P[a + m], we[m], e0);
              o0 = __builtin_elementwise_fma(P[a + 1 + m], wo[m], o0);
            } else {
              e1 = __builtin_elementwise_fma(P[a + m], we[m], e1);
              o1 = __builtin_elementwise_fma(P[a + 1 + m], wo[m], o1);
            }
          }
          const f2 e = e0 + e1, o = o0 + o1;
          d[2 * a] = (e.x + e.y) + __builtin_fmaf(wgt[12], P[a + 6].x, av[2 * a]);
          d[2 * a + 1] = (o.x + o.y) + __builtin_fmaf(wgt[0], P[a].y, av[2 * a + 1]);
        }
      }
      // e, row sums, stores: every half-wave's 32 values of a centre row are one aligned 128-byte run
      if (SSG_DBG(p, 8)) {   // (profiling: no exp / row sums / stores)
#pragma unroll
        for (int j = 0; j < PX; ++j) asm volatile("" ::"v"(d[j]));
      } else
#pragma unroll
      for (int j = 0; j < PX; ++j) {
        const float ev = __builtin_amdgcn_exp2f(d[j] * nk);
        // (row sums in fp64 from the first addition: an fp32 chain over one offset row leaves the sum -- hence the
        // whole row's scale -- a relative 1e-7 off, which the KL part of the gradient sees as 1e-4 of itself)
        rs[j] += (double)ev;
        asm volatile("" : "+v"(rs[j]));   // (or the additions sink to the end of the row, with every e kept for them)
        // (a wave-uniform base per offset + a 32-bit element offset per lane: one v_lshl_add_u64 per store instead of a
        // 64-bit add pair)
        if (!SSG_DBG(p, 1)) (tmbase + (size_t)(qyi * KS + qxi) * TM_PX)[coff[j]] = ev;
      }
    };
    lds_barrier();                                   // (offset 0's H rows are in LDS)
    h_requests(std::integral_constant<int, 0>{});
    static_for(std::make_integer_sequence<int, KS>{}, [&](auto qc) {
      constexpr int qxi = decltype(qc)::value;
      if (!SSG_DBG(p, 2)) edge_stage(qc);
      if constexpr (qxi + 1 < KS) {
        lds_barrier();
        if (!SSG_DBG(p, 2)) h_requests(std::integral_constant<int, qxi + 1>{});
      }
    });
    if (qyi + 1 < KS) lds_barrier();                 // (the E/H role stores the next band row)
  }
  }
#ifdef SSG_PROFILE
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_strip_times[3 * blockIdx.x + 1] = __builtin_readcyclecounter();
#endif
  // deferred normalisation, the tile-major mark (see the tile kernel)
  double *rsc = p.row_scale + (size_t)which * p.n_host;
  if (!eh_role) {
#pragma unroll
    for (int j = 0; j < PX; ++j)
      if (orow[j] >= 0) rsc[orow[j]] = -1.0 / (rs[j] + (double)p.eps);
  }
}

// ------------------------------------------------------------------ host ----
template <int KS, int KW, int C, int NW>
static size_t dense_lds_bytes() {
  constexpr int DT_Y = 16 - 2 * (KW / 2);
  constexpr int HALO = KS / 2 + KW / 2, RH = DT_Y + 2 * HALO, RS = DT_X + 2 * HALO + 1;
  constexpr int UH = DT_Y + 2 * (KW / 2), UW = DT_X + 2 * (KW / 2), NE = DT_Y * DT_X, DT_HS = dense_h_stride(KW);
  return sizeof(float) * (size_t)(C * RH * RS + UH * UW + UH * DT_HS + NW * UH * DT_HS) + sizeof(int) * (NE * 3 + 16 + 64 * NW);
}

bool dense_supported(int ks, int kw, int C) { return C == 3 && ((ks == 25 && kw == 9) || (ks == 49 && kw == 13)); }

// rows of the dense kernels' tiles for a search size (the plan of ssg_edge_list is built for it)
int dense_tile_rows(int ks) { return ks == 49 ? 4 : 8; }

int dense_max_tiles(int B, int H, int W, int ks) {
  const int ty = dense_tile_rows(ks);
  return B * ((H + ty - 1) / ty) * ((W + DT_X - 1) / DT_X);
}

template <int KS, int KW, int C, int NW, bool TM, bool RAW = false, int NCH = 0>
static int launch_fwd_dense_t(DenseParams p, int n_tiles, hipStream_t st) {
  if (n_tiles <= 0) return 0;
  const size_t lds = dense_lds_bytes<KS, KW, C, NW>();
  static std::atomic<unsigned long long> lds_set{0};
  if (const int rc = ensure_dynamic_lds(ssg_fwd_dense<KS, KW, C, NW, TM, RAW, NCH>, (int)lds, lds_set)) return rc;
  p.grid_tiles = n_tiles;
  hipLaunchKernelGGL((ssg_fwd_dense<KS, KW, C, NW, TM, RAW, NCH>), dim3((unsigned)n_tiles * p.nimg), dim3(64 * NW), lds, st, p);
  return (int)hipGetLastError();
}

// (25,9): the two chunk classes of the 8 x 32 tiles, both over the whole tile list (a tile runs in its own)
template <bool RAW>
static int launch_fwd_dense_25(const DenseParams &p0, hipStream_t st) {
  static const bool one_launch = env_int("SSG_DENSE_ONE_LAUNCH", 1) != 0;   // (profiling build: 0 = a launch per class, A/B)
  if (one_launch) {
    DenseParams p = p0;
    if (p.max_tiles <= 0) return 0;
    const size_t lds = dense_lds_bytes<25, 9, 3, 4>();
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(ssg_fwd_dense_classes<25, 9, 3, 4, RAW>, (int)lds, lds_set)) return rc;
    p.grid_tiles = p.max_tiles;
    hipLaunchKernelGGL((ssg_fwd_dense_classes<25, 9, 3, 4, RAW>), dim3((unsigned)p.max_tiles * p.nimg), dim3(256), lds, st, p);
    return (int)hipGetLastError();
  }
  const DenseParams &p = p0;
  int rc = launch_fwd_dense_t<25, 9, 3, 4, false, RAW, 2>(p, p.max_tiles, st);
  // TILE_HUGE tiles sit among the HEAVY ones, at the front of the list, and a heavy tile holds more than 64 of the call's
  // rows: slots from n_host / 65 on cannot be theirs (with a tight row capacity the launch -- empty at C2 / C4 -- shrinks
  // from 8,192 to 2,400 workgroups)
  const int heavy_max = p.n_host / 65 + 1;
  if (!rc) rc = launch_fwd_dense_t<25, 9, 3, 4, false, RAW, 4>(p, heavy_max < p.max_tiles ? heavy_max : p.max_tiles, st);
  return rc;
}

// strips of ssg_fwd_strip: STRIP_ROWS x 32 centres (k_s 49 only)
int dense_max_strips(int B, int H, int W, int ks) {
  return ks == 49 ? B * ((H + STRIP_ROWS - 1) / STRIP_ROWS) * ((W + DT_X - 1) / DT_X) : 0;
}

template <int KS, int KW, int C, int NW>
static int launch_fwd_strip_t(const DenseParams &p, hipStream_t st) {
  constexpr int UH = 16 * NW, HALO = KS / 2 + KW / 2, RS = DT_X + 2 * HALO + 1, HS = DT_X + 1;
  const size_t lds = sizeof(float) * (size_t)(C * UH * RS + 3 * UH * HS + UH * (DT_X + KW - 1) + 16);
  static std::atomic<unsigned long long> lds_set{0};
  if (const int rc = ensure_dynamic_lds(ssg_fwd_strip<KS, KW, C, NW>, 2 * (int)lds, lds_set)) return rc;
  hipLaunchKernelGGL((ssg_fwd_strip<KS, KW, C, NW>), dim3((unsigned)(p.max_strips * p.nimg + 1) / 2), dim3(256 * NW), 2 * lds, st, p);
  return (int)hipGetLastError();
}

#ifdef SSG_PROFILE
int strip_times(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_strip_times), sizeof(unsigned long long) * (size_t)(n < 3072 ? n : 3072));
}
// profiling build: workgroups of ssg_fwd_strip<49,13,3,3> the runtime says fit one CU with the launch's dynamic LDS
int strip_occupancy() {
  constexpr int KS = 49, KW = 13, C = 3, NW = 3, UH = 16 * NW, HALO = KS / 2 + KW / 2, RS = DT_X + 2 * HALO + 1, HS = DT_X + 1;
  const size_t lds = sizeof(float) * (size_t)(C * UH * RS + 3 * UH * HS + UH * (DT_X + KW - 1) + 16);
  static std::atomic<unsigned long long> lds_set{0};
  if (ensure_dynamic_lds(ssg_fwd_strip<KS, KW, C, NW>, 2 * (int)lds, lds_set)) return -1;
  int n = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ssg_fwd_strip<KS, KW, C, NW>, 256 * NW, 2 * lds) != hipSuccess) return -2;
  return n;
}
#endif

// a call with a tile-major region (k_s 49, fused step with a row-scale array) launches both variants over the tile
// list; tm_active() -- device-side, from the plan's counts -- lets one of them run
int launch_fwd_dense(const DenseParams &p0, int ks, int kw, int C, hipStream_t st) {
  if (!dense_supported(ks, kw, C)) return -1;
  if (p0.max_tiles == 0) return 0;
  DenseParams p = p0;
  const bool tm = ks == 49 && p.tm[0] && p.tm_slots > 0 && p.row_scale && p.generalization && (p.nimg == 1 || p.tm[1]);
  if (!tm) p.tm_slots = 0;
  if (p.raw) {   // the reference operator: raw distances accumulated into row-major rows, no row scales, no tile-major region
    p.tm_slots = 0;
    p.row_scale = nullptr;
    p.strips = nullptr;
    if (ks == 25) return launch_fwd_dense_25<true>(p, st);
    return launch_fwd_dense_t<49, 13, 3, 7, false, true>(p, p.max_tiles, st);
  }
  if (ks == 25) return launch_fwd_dense_25<false>(p, st);
  if (!tm) p.strips = nullptr;
  int rc = (tm && p.strips && p.max_strips > 0) ? launch_fwd_strip_t<49, 13, 3, 3>(p, st) : 0;
  if (!rc && tm) rc = launch_fwd_dense_t<49, 13, 3, 7, true>(p, p.tm_slots < p.max_tiles ? p.tm_slots : p.max_tiles, st);
  if (!rc) rc = launch_fwd_dense_t<49, 13, 3, 7, false>(p, p.max_tiles, st);
  return rc;
}

}  // namespace ssg
