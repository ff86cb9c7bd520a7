// Dense-tile ("shared-term") backward SSG kernel for gfx950.
//
// The direct backward (ssg_bwd.hip) spends ~2*C*k_w^2 FMAs per (edge pixel, search offset).  Where
// edge pixels are dense the work is shared, exactly as in the forward (ssg_dense.hip): with
//     D[n,q] = sum_{k in K(q)} sum_c (I[x_n+k] - I[x_n+k+q])^2 + sum_{k in win \ K(q)} sum_c I[x_n+k]^2
// (K(q) = the part of the k_w x k_w window whose partner stays inside the search area: the reference's
// "B = 0 outside the area" rule, similarity.cu:43-47 == F.unfold zero padding, loss_util.py:208) and
// G = dL/dD, the reference's per-term atomics (similarity.cu:113-128) sum to
//     dL/dI[c,v] = 2 sum_q { W_q[v] (I[v] - I[v+q])  -  W_q[v-q] (I[v-q] - I[v])  +  V_q[v] I[v] }
//     W_q[u] = sum_{n : u - x_n in K(q)} G[n,q]         (truncated box sum of the sparse field G[.,q])
//     V_q[u] = sum_{n : u - x_n in win \ K(q)} G[n,q]
// in padded coordinates (the reflect fold follows by index mirroring when the result is added to the
// image gradient).  sum_q V_q = Box(sum_b) - sum_{q border} W_q with sum_b[n] = sum of G[n,q] over the
// offsets whose window is truncated (from ssg_grad_rows), so V costs one extra box sum per tile.
//
// One WAVE owns a tile of TY x 32 candidate centres (TY + k_w - 1 = 16) and, for a range of offset
// rows q_y, walks the k_s offsets q_x fully unrolled.  Per offset:
//   1. the tile's edge pixels (census from the rank map) drop G[n,q] into a zero-padded LDS field;
//   2. W_q on the tile grown by the window halo (U, 16 x (32 + k_w - 1) pixels): horizontal truncated
//      sums in lanes (tile row, column group), vertical inclusive prefix across the tile rows with DPP
//      row shifts, prefix rows to LDS; lane (U-row r, column group g) reads two prefix rows: the
//      vertical window is their difference (at most TY terms deep);
//   3. lane (r, g) owns NPX consecutive pixels u of U-row r: I[c,u], a circular register window of
//      I[c,u+q] (one LDS dword per channel and step), accumulators for dL/dI[c,u] (live for the whole
//      sweep) and a circular window of accumulators for dL/dI[c,u+q]; the column of that window that
//      is complete after the step is added to a 16-row LDS band of the gradient region (plain
//      read-modify-write: one wave, in-order LDS, distinct addresses per lane).
// After an offset row the band row that is complete goes to HBM with one fp32 atomic per non-zero
// pixel and the image band advances by one row.  Cost per (tile, offset): ~200 wave instructions
// whatever the number of edge pixels, against ~19 k lane-instructions per edge pixel and offset row
// in the direct kernel.
#include "ssg_common.hpp"

namespace ssg {

template <int SH>
__device__ __forceinline__ float dpp_row_shr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SH, 0xf, 0xf, true));
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

// Window sums out[i] = v[lo + i] + ... + v[hi + i], i < N, of a register array: the first one explicitly, the others
// by sliding (out[i+1] = out[i] - v[lo+i] + v[hi+1+i]) when that is fewer operations.  G is signed, so unlike the
// forward's sums of non-negative terms nothing is lost by subtracting: the error stays a few ulps of the partial sums
// either way (measured by the F10 gradient tests).
template <int N, int LO, int HI, int NV>
__device__ __forceinline__ void window_sums(const float (&v)[NV], float (&out)[N]) {
  static_assert(LO >= 0 && HI >= LO && HI + N - 1 < NV, "windows inside the array");
  constexpr int Wd = HI - LO + 1;
  if constexpr (Wd + 2 * (N - 1) - 1 < N * (Wd - 1)) {
    float t = v[LO];
#pragma unroll
    for (int m = LO + 1; m <= HI; ++m) t += v[m];
    out[0] = t;
#pragma unroll
    for (int i = 1; i < N; ++i) out[i] = (out[i - 1] - v[LO + i - 1]) + v[HI + i];
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float t = v[LO + i];
#pragma unroll
      for (int m = LO + i + 1; m <= HI + i; ++m) t += v[m];
      out[i] = t;
    }
  }
}


// Inclusive prefix over 8 tile rows held 2 lanes apart in a 16-lane DPP row, for 5 values at once: 15 `v_add_f32_dpp`
// (row_shr 2 / 4 / 8 with bound_ctrl: a lane whose source lies outside its row adds 0).  Written with the DPP intrinsic
// -- hipcc folds move and addition into one instruction and places the wait states itself (round 6; the hand-written
// asm block of rounds 3-5 with its two s_nop measured the same: 0.413 ms either way).
__device__ __forceinline__ void dpp_prefix8x5(float (&v)[5]) {
#pragma unroll
  for (int i = 0; i < 5; ++i) v[i] += dpp_row_shr<2>(v[i]);
#pragma unroll
  for (int i = 0; i < 5; ++i) v[i] += dpp_row_shr<4>(v[i]);
#pragma unroll
  for (int i = 0; i < 5; ++i) v[i] += dpp_row_shr<8>(v[i]);
}
// Nothing moves across an offset step: without it hipcc hoists the address arithmetic and the LDS loads of all
// k_s unrolled steps to the top of the offset row (1,100 live values, 650 spilled).
__device__ __forceinline__ void step_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// RG = column groups of the lane map: 4 -> one wave covers the 16 U-rows (16 x 4 lanes); 8 -> a wave covers 8 U-rows
// (8 x 8 lanes) and TWO waves (blockIdx.z = which half) share a tile: half the bands (8 rows) per wave, so that 8
// waves fit a CU's LDS and every SIMD holds two -- a lone wave issues a VALU instruction only every ~4 cycles.
template <int KS, int KW, int C, int TY, int RG = 4>
struct DenseBwdGeo {
  static constexpr int TX = 32, HP = KS / 2, HK = KW / 2, HALO = HP + HK, P = KS * KS;
  static constexpr int UH = TY + 2 * HK, UW = TX + 2 * HK;    // tile grown by the window halo
  static constexpr int NPX = (UW + RG - 1) / RG;              // pixels per lane; RG*NPX >= UW columns are walked (the
                                                              // lanes past U see W = 0: their prefix entries are never written)
  static constexpr int UWP = RG * NPX;
  static constexpr bool REGW = RG == 8 && TY == 8 && UWP == UW;  // one role per lane: W from the lane's own prefix registers
  // (padded) pixels per lane in a prefix row; RG = 8: no pad and 40-float rows -- the half-wave's 4 rows x 8 groups
  // then fall on different banks (48-float rows of 6-float groups: 41 % of the LDS cycles were conflict replays)
  static constexpr int NPXP = RG == 8 ? NPX : ((NPX + 1) & ~1);
  static constexpr int RR = 64 / RG, NHALF = UH / RR;         // U-rows per wave, waves per tile
  static constexpr int RW = UWP + KS - 1, RH = UH + KS - 1;   // gradient / image region
  // channel stride inside a band row.  RG = 4 (k_s 49): RWS = 92, BRS = 276 = 20 (mod 32) -- the band accesses of a
  // step (lane (r, g): row slot (r + q_y) & 15, column 11 g + q_x) then fall on 32 different banks per half-wave; with
  // the odd stride 93 (BRS = 23 mod 32) every one of them was a two-way conflict (27 % of the kernel's LDS cycles).
  static constexpr int RWS = (RG == 4 && (3 * RW) % 32 == 20) ? RW : (RW | 1);
  // band row stride: RG = 4: C odd strides; RG = 8: = 4 (mod 32), the 8 slots x 4 groups (5 floats apart) of a
  // half-wave (lane = 8 * group + row) on 32 different banks
  static constexpr int BRS = REGW ? ((C * RWS - 4 + 31) / 32 * 32 + 4) : (RG == 8 ? ((C * RWS - 8 + 15) / 16 * 16 + 8) : C * RWS);
  static constexpr int BR = RR;                               // band rows
  static constexpr int HG = 64 / TY, HOUT = (UW + HG - 1) / HG;
  // padded G field row.  The horizontal-sum lanes (tile row hty, group hg) read dword hty*GQS + HOUT*hg + m: with
  // GQS = 20 (mod 32) for (TY 8, HOUT 5) / 8 (mod 32) for (TY 4, HOUT 3) the 32 lanes of a half-wave fall on 32
  // different banks
  static constexpr int GQS_MIN = (HG * HOUT > UW ? HG * HOUT : UW) + 2 * HK;
  static constexpr int GQS_RES = TY == 8 ? 20 : 8;
  static constexpr int GQS = GQS_MIN + ((GQS_RES - GQS_MIN % 32) + 32) % 32;
  static constexpr int PS = RG * NPXP;                        // W row
  // RG = 4: the rows a lane can need -- the vertical window of a U-row covers tile rows [a, bb] with a = 0 or bb = TY-1
  // (k_w - 1 >= 2 TY: it is at least TY rows long even when cut), so W is a prefix P_1 .. P_TY (rows 0 .. TY-1) or a
  // suffix S_a = P_TY - P_a (rows TY .. 2TY-2): ONE row per lane instead of the difference of two prefix rows
  // (round 4: -22 of a step's 73 LDS dwords per lane pair, -11 subtractions per role)
  static constexpr int WROWS = 2 * TY - 1;
  // one field + its prefix rows (two copies: steps alternate); RG = 8 keeps the prefix in registers: no rows
  // (+4: the copy's last word is a dummy that absorbs the writes of lanes without an edge pixel / past U)
  static constexpr int FSZ = TY * GQS + (REGW ? 0 : WROWS * PS) + 4;
  static constexpr int NE_MAX = TY * TX, NCHUNK = NE_MAX / 64;
  static constexpr int NG = KS / 4;                           // full groups of 4 offsets per offset row
  static constexpr int CPL = (RW + 63) / 64;                  // region columns per lane (row loads / flushes)
  static_assert(UH == 16 && (RG == 4 || RG == 8), "lane map: 16 x 4 or 2 x (8 x 8)");
  static_assert(KS % 4 == 1, "offset rows are consumed as groups of 4 + 1");
  static_assert(64 % TY == 0 && TY <= 8, "prefix lanes: tile row in the low lane bits");
  static constexpr size_t lds_bytes() {
    return sizeof(float) * (size_t)(2 * BR * BRS + 2 * FSZ + 4);
  }
};

// NCH = 64-pixel chunks of the tile's edge-pixel list this instantiation carries per offset step; a wave
// whose tile needs another count leaves at once (every instantiation is launched over the same tile list), so
// the offset loop is free of control flow.
// TM (k_s 49 only): the tile's rows are tile-major scratch rows of the fused step (TmRowsParams, ssg_common.hpp) -- the
// lanes keep the dense forward's fixed pixel map, read e_sr / e_gt of one offset as aligned 256-byte runs seven
// steps ahead, and form G = -(s k)(g - dot) themselves (criteria_elem's g, bit for bit what ssg_rows_tm summed into
// dot) instead of reading G rows that ssg_grad_rows would have had to write.
// SPL (tile-major k_s 49 only): TWO waves per tile in one workgroup, the lane map unchanged, the work split by ROLE --
//   role 0: channels 0 and 1 (the packed pair) of both ends of every pair, G formation (tile-major loads, field
//           writes), the border sums swb;
//   role 1: channel 2, the W stage (horizontal sums, vertical prefix, prefix rows to LDS).
// Every LDS array stays what it was (the bands are per channel, so each role flushes and refills its own); the two
// waves meet at ONE s_barrier per offset step: G[.,t] (role 0, end of step t-2) -> prefix rows (role 1, end of step
// t-1) -> W (both, top of step t).  Each role's registers are its own (≈ 190 / ≈ 110 instead of 256 + 138 parked in
// AGPRs), so a SIMD holds two waves.  ROLE 2 = one wave does everything (the other instantiations).
template <int KS, int KW, int C, int TY, int NCH, int RG, bool TM, bool SPL, int ROLE>
__device__ __forceinline__ void bwd_dense_body(const DenseBwdParams &p) {
  using G = DenseBwdGeo<KS, KW, C, TY, RG>;
#ifndef SSG_SWB_SKIP
#define SSG_SWB_SKIP 0
#endif
  constexpr int SWB_SKIP = SSG_SWB_SKIP;   // 1: role 0 carries the border sums, 0: role 1
  static_assert(SPL == (ROLE != 2) && (!SPL || (TM && RG == 4)), "roles: tile-major k_s 49 instantiation only");
  constexpr bool DO_A = ROLE != 1, DO_B = ROLE != 0;   // channels (0,1) / channel 2
  constexpr bool DO_G = ROLE != 1, DO_W = ROLE != 0;   // G formation / W stage
  constexpr bool DO_S = ROLE != SWB_SKIP;               // border sums swb (experiment knob: which role carries them)
  // LDS hand-over between the steps' stages: one wave = program order; two waves = LDS writes retired, then s_barrier
  // (not __syncthreads(): it would also drain the tile-major loads that are seven steps ahead)
  auto stage_sync = [] {
#ifdef SSG_TIMING_NO_BARRIER   // (timing experiment, wrong results: what the s_barrier itself costs)
    if constexpr (SPL) asm volatile("" ::: "memory");
#else
    if constexpr (SPL) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    // (nothing is scheduled across: without this the next step's body -- register operands only -- moves up in front of
    // the s_barrier, behind a wait for ALL of the step's LDS reads, and no longer covers their latency)
    if constexpr (SPL) __builtin_amdgcn_sched_barrier(0);
    else __builtin_amdgcn_wave_barrier();
  };
  auto chan_on = [](int c) { return c < 2 ? DO_A : DO_B; };
  constexpr int RR = G::RR, RMASK = RR - 1, BRS = G::BRS;
  constexpr int TX = G::TX, HP = G::HP, HK = G::HK, HALO = G::HALO, P = G::P;
  constexpr int UW = G::UW, NPX = G::NPX, NPXP = G::NPXP, RW = G::RW, RH = G::RH, RWS = G::RWS;
  constexpr int HOUT = G::HOUT, GQS = G::GQS, PS = G::PS, NCHUNK = G::NCHUNK, NG = G::NG, CPL = G::CPL;
  constexpr int FSZ = G::FSZ;
  constexpr bool REGW = G::REGW;
  constexpr int DUMMY = FSZ - 1;  // last word of a copy (offsets are relative to the copy's base)
  // instantiations: (25,9) 2 and 4 chunks -- the plan's TILE_HUGE tiles (more than 128 edge pixels) and all others;
  // (49,13) 2 chunks (the whole 4 x 32 tile)
  constexpr int NCH_LO = NCH <= 2 ? 0 : NCH / 2;
  static_assert(NG % 2 == 0 && NCH <= NCHUNK, "two G slots alternate over an even number of groups");
  constexpr int TMD = 7;   // tile-major rows: offsets in flight (ring slots; k_s % TMD == 0 keeps slot = q_x % TMD)
  static_assert(!TM || (RG == 4 && TY == 4 && NCH == NCHUNK && NCH * 64 == TM_PX && KS % TMD == 0), "tile-major rows: 4 x 32 tiles, one wave");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *imgb = smem;                       // [16][C][RWS] image band: region rows q_y .. q_y+15
  float *grb = imgb + RR * BRS;             // [RR][BRS >= C*RWS] gradient band, same rows
  // two copies (offset steps alternate) of { [TY][GQS] G[.,q] on the tile, 2*HK zeros left of every row;
  //                                          [TY+1][PS] vertical prefix of the horizontal sums (row 0 = 0) }
  float *fld = grb + RR * BRS;
  int *elist = (int *)imgb;  // [NE_MAX][2] (field offset, row): prologue only, the image band is filled afterwards
  static_assert(2 * G::NE_MAX <= 2 * RR * BRS, "edge list aliases the bands");

  const int lane = threadIdx.x & 63;
  const int tslot = blockIdx.x;
  if (tslot >= dense_tile_count(p.n_dense)) return;
  if (p.tm_slots > 0 && tm_active(p.n_dense, p.tm_slots, rows_to_do(p.n_dev, p.n_host)) != TM) return;
  if (TM && p.tm_slots <= 0) return;
  if (p.n_dense[1] != TY) {   // plan cut for another tile height (see ssg_fwd_dense): nothing is done, the status word says so
    if (p.status && lane == 0) atomicOr(p.status, 1);
    return;
  }
  const int H = p.H, W = p.W;
  const int tx_n = (W + TX - 1) / TX, ty_n = (H + TY - 1) / TY;
  const int listed = dense_tile_at(p.n_dense, p.tiles, p.B * ty_n * tx_n, tslot);
  if constexpr (TY == 8) {
    // the tile's class by the plan (TILE_HUGE marks the tiles with more than 128 edge pixels): one instantiation per
    // class is launched, the other leaves before its census.  (A capacity clamp can only lower a tile's count below its
    // class' range, never raise it.)  A one-chunk instantiation for the tiles with <= 64 edge pixels (69 % of C2's) was
    // measured: its launch and the two-chunk one each fill part of the chip for a whole sweep -- 0.446 -> 0.509 ms.
    const int cls = (listed & TILE_HUGE) ? 4 : 2;
    if (cls != NCH) return;
  }
  const int tile = dense_tile_id(listed);
  const int b = tile / (tx_n * ty_n), tr = tile - b * tx_n * ty_n;
  const int ty0 = (tr / tx_n) * TY, tx0 = (tr % tx_n) * TX;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const float gsc = grad_fix_scale(p.gfix, (size_t)p.B * C * H * W);
  // offset rows of this wave: a fixed split, or (qsplit 0) one chosen from the tile count -- wave-uniform scalars
  int qs = p.qsplit;
  if (qs <= 0) {
    const int want = p.auto_slots / (dense_tile_count(p.n_dense) * G::NHALF);
    qs = want < 1 ? 1 : (want > (int)gridDim.y ? (int)gridDim.y : want);
  }
  if ((int)blockIdx.y >= qs) return;
  const int qy0 = (KS * (int)blockIdx.y) / qs, qy1 = (KS * ((int)blockIdx.y + 1)) / qs;

  // ---- census of the tile's edge pixels (row-major inside the tile) ----
  int n_e = 0;
  int tm_pos[NCH], tm_row[NCH];   // tile-major rows: the lane's pixels by the fixed map, no list
  // (the rank map's loads of all chunks first -- clamped addresses, no exec-masked blocks: four serialised round trips
  // to memory otherwise stand at the head of every wave)
  int rk[NCHUNK];
#pragma unroll
  for (int k = 0; k < NCHUNK; ++k) {
    const int pos = lane + 64 * k;
    const int ey = TM ? tm_pixel_row(k, lane) : pos / TX, ex = TM ? tm_pixel_col(lane) : pos - (pos / TX) * TX;
    const int y = ty0 + ey, x = tx0 + ex;
    const int v = p.rank[((size_t)b * H + (y < H ? y : H - 1)) * W + (x < W ? x : W - 1)];
    rk[k] = (y < H && x < W) ? v : -1;
  }
#pragma unroll
  for (int k = 0; k < NCHUNK; ++k) {
    const int pos = lane + 64 * k;
    const int ey = TM ? tm_pixel_row(k, lane) : pos / TX, ex = TM ? tm_pixel_col(lane) : pos - (pos / TX) * TX;
    int r = rk[k];
    if (r >= nrows) r = -1;
    const unsigned long long bal = __ballot(r >= 0);
    if constexpr (TM) {
      tm_pos[k < NCH ? k : 0] = r >= 0 ? ey * GQS + 2 * HK + ex : DUMMY;
      tm_row[k < NCH ? k : 0] = r;
    } else if (r >= 0) {
      const int at = n_e + __popcll(bal & ((1ull << lane) - 1ull));
      elist[2 * at] = ey * GQS + 2 * HK + ex;
      elist[2 * at + 1] = r;
    }
    n_e += __popcll(bal);
  }
  if constexpr (TY == 8) {
    if (n_e == 0 || n_e > 64 * NCH) return;   // (the class was checked above; an emptied tile has nothing to add)
  } else {
    if (n_e <= 64 * NCH_LO || n_e > 64 * NCH) return;
  }
  if constexpr (DO_G) for (int i = lane; i < 2 * FSZ + 4; i += 64) fld[i] = 0.f;  // fields (pads stay 0) and prefix rows 0
  if constexpr (DO_W) for (int i = lane; i < RR * BRS; i += 64) grb[i] = 0.f;
  stage_sync();

  int epos[NCH];
  const float *gp[NCH];
  int erow[NCH];
  // tile-major rows: the two row scales (as the floats ssg_rows_tm used) and dot of the lane's pixels; 0 for a hole
  // (s = t = 0 -> G = 0, dropped into the dummy word)
  TmScale tsa[NCH], tsb[NCH];
  float tdot[NCH];
#pragma unroll
  for (int ck = 0; ck < NCH; ++ck) {
    if constexpr (!DO_G) {
      epos[ck] = DUMMY;
      erow[ck] = 0;
      gp[ck] = nullptr;
      tsa[ck] = tsb[ck] = TmScale{0.f, 0.f};
      tdot[ck] = 0.f;
    } else if constexpr (TM) {
      const int r = tm_row[ck];
      epos[ck] = tm_pos[ck];
      erow[ck] = r >= 0 ? r : 0;
      gp[ck] = nullptr;
      tsa[ck] = tm_scale(r >= 0 ? p.row_scale[r] : 0.0);
      tsb[ck] = tm_scale(r >= 0 ? p.row_scale[(size_t)p.n_host + r] : 0.0);
      tdot[ck] = r >= 0 ? p.dot[r] : 0.f;   // (times k: below)
    } else {
      const int e = ck * 64 + lane;
      const bool on = e < n_e;
      epos[ck] = on ? elist[2 * e] : DUMMY;
      erow[ck] = on ? elist[2 * e + 1] : elist[1];
      gp[ck] = p.G + (SSG_DBG(p, 16) ? (size_t)0 : (size_t)erow[ck] * P);   // (profiling: bit 4 = every lane streams row 0)
      tsa[ck] = tsb[ck] = TmScale{0.f, 0.f};
      tdot[ck] = 0.f;
    }
  }

  // lane roles: main (U-row r, column group g); prefix (tile row hty, column group hg)
  // RG = 4: main lane = (U-row lane/4, group lane%4), prefix lane = (tile row lane%TY, group lane/TY), W through LDS
  // prefix rows.  RG = 8 (TY = 8, HOUT = NPX): ONE role per lane -- lane = 8*group + j -- so that the vertical prefix a
  // lane computes IS the W of its own pixels: the first half (U-rows 0..7) sums tile rows 0..j for U-row j; the
  // second half walks the tile rows in reverse (lane j <-> tile row 7-j, U-row 15-j), so its prefix is the suffix
  // sum rows 7-j..7 it needs.  Only offset rows whose window is cut vertically fetch two other lanes' prefixes
  // (ds_bpermute inside the 8-lane group).
  const int half = REGW ? (int)blockIdx.z : 0;
  // (RG = 8 lane map: lane = 2*j + (g & 1) + 16*(g >> 1) -- the 8 tile rows of a group sit 2 lanes apart inside one
  // 16-lane DPP row, so that the prefix scan is three in-place `v_add_f32_dpp row_shr:2/4/8`: a lane whose source
  // falls off its DPP row is disabled (bound_ctrl off), which is exactly the scan's "j >= shift" condition.  A
  // half-wave still holds groups 0..3 x rows 0..7, so the bank maps below are unchanged)
  const int jrow = REGW ? (lane >> 1) & 7 : lane & 7;
  const int g = REGW ? ((lane & 1) | ((lane >> 4) << 1)) : lane % RG;
  const int r = REGW ? (half ? 15 - jrow : jrow) : RR * (int)blockIdx.z + lane / RG;  // U-row of the tile
  const int hty = REGW ? jrow : lane % TY, hg = REGW ? g : lane / TY;
  const float m1 = hty >= 1 ? 1.f : 0.f, m2 = hty >= 2 ? 1.f : 0.f, m4 = hty >= 4 ? 1.f : 0.f;
  const int hsrc = (REGW && half ? TY - 1 - hty : hty) * GQS + HOUT * hg;  // (offsets relative to a copy's base)
  int hdst[HOUT], hdst2[HOUT];   // prefix row hty, suffix row TY-1+hty (hty >= 1)
#pragma unroll
  for (int i = 0; i < HOUT; ++i) {
    const int uc = HOUT * hg + i;
    hdst[i] = uc < UW ? TY * GQS + hty * PS + (uc / NPX) * NPXP + uc % NPX : DUMMY;  // (columns past U: the dummy word)
    // (tile row 0 has no suffix row of its own: S_0 = P_TY, written once more into the last prefix row -- 16 lanes on
    // the dummy word would be a 16-way bank conflict per store)
    hdst2[i] = uc < UW ? TY * GQS + (TY - 1 + hty) * PS + (uc / NPX) * NPXP + uc % NPX : DUMMY;
  }
  // suffix S_hty = P_TY (the group's last tile row: quad lane 3 / lane TY-1 of the prefix group) - P_hty (the lane below)
  auto suffix_of = [&](float pfx) {
    static_assert(REGW || TY == 4, "suffix rows: the TY = 4 tile rows of a group are one DPP quad");
    const float tot = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, pfx), 0xff, 0xf, 0xf, true));  // quad_perm [3,3,3,3]
    return __builtin_fmaf(-m1, dpp_row_shr<1>(pfx), tot);   // (m1 = 0 for tile row 0; exact product: the same rounding as tot - P)
  };

  float wout[HOUT];  // RG = 8: the prefix of the offset prepared last (consumed by the next step)
  // The box sum W of the field in copy `f` on the lane's NPX pixels comes in two halves, so that consecutive
  // offset steps overlap (x_stage of step s+1 runs beside the body of step s):
  //   x_stage: horizontal sums over the column taps kept (XLO..XHI), vertical inclusive prefix, prefix rows to LDS;
  //   y_read : the lane's pixels = difference of the two prefix rows bounding the row taps kept.
  auto x_stage = [&](auto xlo_c, auto xhi_c, float *f) {
    constexpr int XLO = decltype(xlo_c)::value, XHI = decltype(xhi_c)::value;
    constexpr int M0 = HK - XHI, M1 = HOUT - 1 + HK - XLO;  // taps m of output i: [i + HK - XHI, i + HK - XLO]
    float v[HOUT + 2 * HK];
#pragma unroll
    for (int m = M0; m <= M1; ++m) v[m] = f[hsrc + m];
    float out[HOUT];
    window_sums<HOUT, HK - XHI, HK - XLO>(v, out);
    // inclusive prefix over the tile rows (adjacent lanes); the 0/1 factors stop a row group from reading
    // its neighbour's lanes
    if constexpr (REGW && HOUT == 5) {
      dpp_prefix8x5(out);
    } else {
#pragma unroll
      for (int i = 0; i < HOUT; ++i) {
        out[i] = __builtin_fmaf(dpp_row_shr<1>(out[i]), m1, out[i]);
        if constexpr (TY > 2) out[i] = __builtin_fmaf(dpp_row_shr<2>(out[i]), m2, out[i]);
        if constexpr (TY > 4) out[i] = __builtin_fmaf(dpp_row_shr<4>(out[i]), m4, out[i]);
      }
    }
    if constexpr (!REGW) {
#pragma unroll
      for (int i = 0; i < HOUT; ++i) {
        f[hdst[i]] = out[i];
        f[hdst2[i]] = suffix_of(out[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < HOUT; ++i) wout[i] = out[i];
    }
  };
  // (a lane's W row segment is NPXP = 12 floats on a 16-byte boundary -- so are the 12 zeros: three ds_read_b128)
  auto y_read = [&](const float *f, int pw, float (&Wv)[NPX]) {
    if constexpr (!REGW && NPXP % 4 == 0 && NPXP >= NPX) {
      // (the pad word is not loaded -- ds_read_b96 for the last three: a dead destination register would be reused at
      // once and make its next writer wait for the whole read)
      typedef float f4a __attribute__((ext_vector_type(4), aligned(16)));
      typedef float f3a __attribute__((ext_vector_type(3), aligned(16)));
      const float *q = (const float *)__builtin_assume_aligned(f + pw, 16);
#pragma unroll
      for (int k = 0; k < NPXP / 4; ++k) {
        if (4 * k + 4 <= NPX) {
          const f4a t = *(const f4a *)(q + 4 * k);
#pragma unroll
          for (int j = 0; j < 4; ++j) Wv[4 * k + j] = t[j];
        } else if (4 * k + 3 == NPX) {
          const f3a t = *(const f3a *)(q + 4 * k);
#pragma unroll
          for (int j = 0; j < 3; ++j) Wv[4 * k + j] = t[j];
        } else {
#pragma unroll
          for (int j = 0; 4 * k + j < NPX; ++j) Wv[4 * k + j] = q[4 * k + j];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPX; ++i) Wv[i] = f[pw + i];
    }
  };
  // prefix rows of U-row r for row taps [ylo, yhi]: tile rows [r-HK-yhi, r-HK-ylo] clipped to the tile
  // (no tile row in the window: the 2 HK zeros left of the field's first row)
  static_assert(REGW || (KW - 1 >= 2 * TY && 2 * HK >= NPX), "W rows: prefix or suffix; the zero pad serves a whole lane");
  auto prefix_rows = [&](int ylo, int yhi) {
    int a = r - HK - yhi, bb = r - HK - ylo;
    a = a < 0 ? 0 : a;
    bb = bb > TY - 1 ? TY - 1 : bb;
    const bool none = a > bb;
    return none ? 0 : TY * GQS + (a == 0 ? bb : TY - 1 + a) * PS + NPXP * g;
  };

  auto row_shift_of = [&](int qy) {   // (= row_shift below: the storage shift of the G field of offset row qy)
    if constexpr (!REGW) return 0;
    else return !half ? (qy < HK ? HK - qy : 0) : (qy > KS - 1 - HK ? (KS - 1 - HK) - qy : 0);
  };
  // RG = 8, offset rows with a vertically cut window [ylo, yhi]: W = (prefix of lane jp) - (prefix of lane jn) of the
  // lane's group.  The window is cut at ONE end only (k_s > 2 (k_w/2)), and then either the positive term is the lane's
  // OWN prefix or there is no negative term -- top half: yhi = HK gives a = 0 (no negative term), ylo = -HK gives bb = j
  // (own); bottom half (lane j holds rows 7-j..7): yhi = HK gives a = 7 - j (own), ylo = -HK gives bb = 7 (no negative
  // term).  So ONE ds_bpermute per value serves every case: W = c_own * own + c_oth * prefix(lane l_oth)
  // (round 6: the two-exchange form cost the kernel 14 % of its time in the 8 cut rows of 25 -- ablation
  // profiles/r6_bwd_dense_ablation.txt).  `l_oth` is a byte address for ds_bpermute.
  auto cut_rows = [&](int qy, int &l_oth, float &c_own, float &c_oth) {
    const int ylo = (-HK > -qy) ? -HK : -qy, yhi = (HK < KS - 1 - qy) ? HK : KS - 1 - qy;
    if (row_shift_of(qy) != 0) {   // the field of this row is stored shifted: every lane's own prefix is its window (see row_shift)
      c_own = 1.f;
      c_oth = 0.f;
      l_oth = 0;
      return;
    }
    int a = r - HK - yhi, bb = r - HK - ylo;  // tile rows [a, bb] feed U-row r
    a = a < 0 ? 0 : a;
    bb = bb > TY - 1 ? TY - 1 : bb;
    const bool none = a > bb;
    const int base = lane & ~14;  // row 0 of this lane's group (rows are 2 lanes apart)
    int jp, jn;
    if (!half) {  // lane k holds rows 0..k: rows [a,bb] = lane bb - lane (a-1)
      jp = bb;
      jn = a - 1;
    } else {      // lane k holds rows 7-k..7: rows [a,bb] = lane (7-a) - lane (6-bb)
      jp = TY - 1 - a;
      jn = TY - 2 - bb;
    }
    const bool own = !none && jp == jrow;
    const bool neg = !none && jn >= 0;
    // (own || !neg by the case analysis above; a window cut at both ends -- impossible for k_w <= k_s -- would need both)
    c_own = own ? 1.f : 0.f;
    c_oth = none ? 0.f : (own ? (neg ? -1.f : 0.f) : 1.f);
    l_oth = 4 * (base + 2 * (own ? (neg ? jn : 0) : (none ? 0 : jp)));
  };
  // the lane exchange of a cut row is ISSUED here (end of the step that produced `wout`) and consumed by w_finish behind
  // the next step's LDS reads and pixel differences, which cover its latency
  auto w_exchange = [&](int l_oth, float (&vx)[NPX]) {
#pragma unroll
    for (int i = 0; i < NPX; ++i)
      vx[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(l_oth, __builtin_bit_cast(int, wout[i < HOUT ? i : 0])));
  };
  auto w_finish = [&](bool full, const float (&vx)[NPX], float c_own, float c_oth, float (&Wv)[NPX]) {
    static_assert(!REGW || HOUT == NPX, "one role per lane");
    if (full) {
#pragma unroll
      for (int i = 0; i < NPX; ++i) Wv[i] = wout[i < HOUT ? i : 0];
    } else {
#pragma unroll
      // (the rows that take the exchange are the ones cut on the wave's OWN side: W = own prefix - another lane's, and
      // c_own == 1 in every lane -- see cut_rows; the rows cut on the other side are stored shifted, see row_shift)
      for (int i = 0; i < NPX; ++i) Wv[i] = __builtin_fmaf(c_oth, vx[i], wout[i < HOUT ? i : 0]);
    }
  };

  // ---- the lane's own pixels, and Box(sum_b) on them ----
  const float *src = p.img + (size_t)b * C * H * W;
  // (channels 0 and 1 of a pixel travel as one packed-fp32 register pair, channel 2 alone: the per-pixel work of a
  // step is then 1 v_pk + 1 scalar instruction instead of 3 -- the ring slots of the u+q window rotate by one pixel
  // per step, so pairing two PIXELS would alternate alignment, pairing two CHANNELS of one pixel never does)
  static_assert(C == 3, "channel pairing: (0,1) packed + 2");
  f2 iuA[NPX];
  float iuB[NPX];
  {
    int gy = reflect_idx(ty0 - HK + r, H);
    gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      int gx = reflect_idx(tx0 - HK + NPX * g + i, W);
      gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
      iuA[i] = DO_A ? f2{src[((size_t)0 * H + gy) * W + gx], src[((size_t)1 * H + gy) * W + gx]} : f2{0.f, 0.f};
      iuB[i] = DO_B ? src[((size_t)2 * H + gy) * W + gx] : 0.f;
    }
  }
  // image band rows: global -> registers -> LDS (lane = region column)
  auto load_img_row = [&](int rho, float (&v)[C][CPL]) {
    int gy = reflect_idx(ty0 - HALO + rho, H);
    gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int col = lane + 64 * k;
      int gx = reflect_idx(tx0 - HALO + (col < RW ? col : 0), W);
      gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
#pragma unroll
      for (int c = 0; c < C; ++c) v[c][k] = chan_on(c) ? src[((size_t)c * H + gy) * W + gx] : 0.f;
    }
  };
  auto store_img_row = [&](int rho, const float (&v)[C][CPL]) {
    float *dst = imgb + (rho & RMASK) * BRS;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int col = lane + 64 * k;
      if (col < RW) {
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (chan_on(c)) dst[c * RWS + col] = v[c][k];
      }
    }
  };
  // one finished row of the gradient band -> HBM (reflect fold by index mirroring), band row cleared
  auto flush_row = [&](int rho) {
    float *brow = grb + (rho & RMASK) * BRS;
    const int py = ty0 - HALO + rho;
    const int gy = reflect_idx(py, H);
    const bool yok = gy >= 0 && gy < H;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int col = lane + 64 * k;
      if (col < RW) {
        const int gx = reflect_idx(tx0 - HALO + col, W);
        const bool ok = yok && gx >= 0 && gx < W;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          if (!chan_on(c)) continue;
          const float v = brow[c * RWS + col];
          brow[c * RWS + col] = 0.f;
          if (ok && v != 0.f && !SSG_DBG(p, 8)) grad_add(p.grad, p.gfix, (((size_t)b * C + c) * H + gy) * W + gx, 2.f * v, gsc);
        }
      }
    }
  };

  const int r0 = RR * (int)blockIdx.z;  // first U-row of this wave: its bands hold region rows r0 + q_y .. + RR - 1
  for (int rho = r0 + qy0; rho < r0 + qy0 + RR; ++rho) {
    float v[C][CPL];
    load_img_row(rho, v);
    store_img_row(rho, v);
  }

  f2 guA[NPX];
  float guB[NPX], swb[NPX];
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    swb[i] = 0.f;
    guA[i] = f2{0.f, 0.f};
    guB[i] = 0.f;
  }

  // G values of the lane's edge pixels, 4 offsets per load, one group ahead (two register slots) + the row's
  // last offset on its own
  float4u gbuf[2][TM ? 1 : NCH];
  float gl[NCH];
  auto load_group = [&](int qyi, int k, float4u (&dst)[TM ? 1 : NCH]) {
    if constexpr (!TM) {
#pragma unroll
      for (int ck = 0; ck < NCH; ++ck) dst[ck] = *(const float4u *)(gp[ck] + qyi * KS + 4 * k);
    }
  };
  auto load_last = [&](int qyi) {
    if constexpr (!TM) {
#pragma unroll
      for (int ck = 0; ck < NCH; ++ck) gl[ck] = gp[ck][qyi * KS + KS - 1];
    }
  };
  // tile-major rows: ring of TMD offsets (slot = q_x % TMD), e_sr / e_gt of the lane's NCH pixels each
  float tea[TM ? TMD : 1][NCH], teb[TM ? TMD : 1][NCH];
  const float *tma = nullptr, *tmb = nullptr;
  float tw1 = 0.f, tw2 = 0.f, tkf = 0.f;
  if constexpr (TM && DO_G) {
    // (wave-uniform: scalar base + lane offset addressing; profiling bit 4: every tile streams slot 0 -- L2 hits)
    tma = p.tm[0] + (size_t)(SSG_DBG(p, 16) ? 0 : tslot) * P * TM_PX;
    tmb = p.tm[1] + (size_t)(SSG_DBG(p, 16) ? 0 : tslot) * P * TM_PX;
    const float invM = 1.f / ((float)nrows * (float)P);
    tkf = 1.f / (p.sigma * (float)(C * KW * KW));
    tw1 = -tkf * (p.w_l1 * invM * (p.upstream ? p.upstream[0] : 1.f));
    tw2 = tkf * (p.w_kl * invM * (p.upstream ? p.upstream[1] : 1.f));
#pragma unroll
    for (int ck = 0; ck < NCH; ++ck) tdot[ck] *= tkf;
  }
  auto tm_load = [&](auto slot_c, int q) {   // offset q (linear) -> ring slot
    constexpr int sl = decltype(slot_c)::value;
    if constexpr (TM && DO_G) {
#pragma unroll
      for (int ck = 0; ck < NCH; ++ck) {
        tea[sl][ck] = __builtin_nontemporal_load(tma + (size_t)q * TM_PX + ck * 64 + lane);
        teb[sl][ck] = __builtin_nontemporal_load(tmb + (size_t)q * TM_PX + ck * 64 + lane);
      }
    }
  };
  if constexpr (TM) {
    static_for(std::make_integer_sequence<int, TMD>{}, [&](auto sc) { tm_load(sc, qy0 * KS + decltype(sc)::value); });
  } else if constexpr (DO_G) {
    load_group(qy0, 0, gbuf[0]);
    load_group(qy0, 1, gbuf[1]);
    load_last(qy0);
  }
  // REGW: the G field of an offset row whose vertical window is cut on the side AWAY from the wave's half is stored
  // SHIFTED by whole rows, so that the lanes' plain prefixes are the cut windows' sums (no lane exchange in those rows):
  //   top half (U-row j = prefix over tile rows 0..j), q_y < HK: rows 0 .. j-(HK-q_y) are wanted = the prefix of the field
  //     moved DOWN by HK - q_y rows (rows pushed out of the tile are not wanted by any U-row of this half);
  //   bottom half (lane j' = suffix over rows 7-j'..7), q_y > k_s-1-HK: rows 7-j'+(q_y-(k_s-1-HK)) .. 7 = the suffix of the
  //     field moved UP by that many rows.
  // A shifted offset leaves its values at other positions than the offsets around it: the two steps of a row that write
  // the NEXT row's first offsets clear the positions they replace (x_clear), and the epilogue clears its copy.
  auto row_shift = [&](int qy) { return row_shift_of(qy); };
  auto shift_pos = [&](int e, int sh) {   // field position under a row shift; rows pushed out of the tile -> the dummy word
    const int ny = e / GQS + sh;          // (the dummy word itself lies in "row" TY)
    return (e != DUMMY && ny >= 0 && ny < TY) ? e + sh * GQS : DUMMY;
  };
  // G[.,q] of one offset into field copy f; x_put(.., qyi, qxi) reads the registers the schedule below filled.
  // (tile-major rows: G is formed here; `q_refill` = the linear offset TMD steps ahead that refills the ring slot)
  auto x_put = [&](auto qx_c, int qyi, float *f, const int (&pos)[NCH], int q_refill = 0) {
    constexpr int qxi = decltype(qx_c)::value, grp = qxi / 4, gj = qxi % 4;
    if constexpr (!DO_G) return;
#pragma unroll
    for (int ck = 0; ck < NCH; ++ck) {
      float gv;
      if constexpr (TM) {
        // G = -k s (g - dot) with s g = w1 s sgn(s - t) - w2 t' (t' = max(t, 1e-10) where s >= 1e-10, else 0: the
        // product s * t'/s of KLDistanceLoss' gradient without the division), the constants folded:
        // tw1 = -k w1, tw2 = k w2, tdot = k dot.  sgn by scaling and clamping: exact for |s - t| >= 2^-126.
        const float cl = 1e-10f;
        const float a = tm_apply(tea[qxi % TMD][ck], tsa[ck]), t = tm_apply(teb[qxi % TMD][ck], tsb[ck]);
        const float sg = __builtin_amdgcn_fmed3f((a - t) * 0x1p126f, -1.f, 1.f);
        const float bz = a >= cl ? fmaxf(t, cl) : 0.f;
        gv = __builtin_fmaf(a, __builtin_fmaf(sg, tw1, tdot[ck]), tw2 * bz);
      } else {
        gv = grp < NG ? gbuf[grp & 1][ck][gj] : gl[ck];
      }
      if constexpr (qxi == HP) gv = (qyi == HP) ? 0.f : gv;  // centre offset: A - B == 0 exactly
      f[pos[ck]] = gv;
    }
    if constexpr (TM) tm_load(std::integral_constant<int, qxi % TMD>{}, q_refill);
  };
  // Two waves (SPL): every hand-over crosses an s_barrier, and a read issued right behind the barrier would put its
  // whole LDS round trip on the step's critical path.  The stages therefore run ONE offset further ahead (LAG): G[., t]
  // at the end of step t-3, its prefix rows during step t-2, and both waves fetch W of offset t in the middle of step
  // t-1 (behind the body, consumed a step later from registers): no step waits for data written in the step before.
  // Two field copies still suffice (F_t: written end of t-3, read top of t-2; P_t: written end of t-2, read in t-1).
  constexpr int LAG = SPL ? 1 : 0;
  static_assert(!LAG || !REGW, "the lagged schedule reads W from LDS prefix rows");
  float Wv[NPX];   // (LAG: W of the offset the next step consumes)
  // The work of one offset t is spread over three steps so that a step waits for LDS ONCE:
  //   end of step t-2 : G[., t] into field copy t&1                       (x_put)
  //   step t-1        : top: read the field; horizontal sums + prefix; end: prefix rows to copy t&1   (x_stage)
  //   step t          : top: read the two prefix rows (W), the band column to flush, the next window
  //                     column; body; end: band column, next step's writes
  // Every LDS read of a step is issued at its top and touches only what earlier steps wrote.
  {  // first two offsets of the sweep (copy = parity of the offset's linear index; k_s is odd)
    float *f0 = fld + ((qy0 * KS) & 1) * FSZ, *f1 = fld + (((qy0 * KS) & 1) ^ 1) * FSZ;
    int epos0[NCH];
#pragma unroll
    for (int ck = 0; ck < NCH; ++ck) epos0[ck] = REGW ? shift_pos(epos[ck], row_shift(qy0)) : epos[ck];
    x_put(std::integral_constant<int, 0>{}, qy0, f0, epos0, qy0 * KS + TMD);
    x_put(std::integral_constant<int, 1>{}, qy0, f1, epos0, qy0 * KS + TMD + 1);
    stage_sync();
    if constexpr (DO_W) x_stage(std::integral_constant<int, (-HK > 0 ? -HK : 0)>{}, std::integral_constant<int, HK>{}, f0);
    if constexpr (LAG) {   // one offset more in flight (see the step schedule below): P(1), F(2), and W of offset 0
      if constexpr (DO_W) x_stage(std::integral_constant<int, (-HK > -1 ? -HK : -1)>{}, std::integral_constant<int, HK>{}, f1);
      stage_sync();
      x_put(std::integral_constant<int, 2>{}, qy0, f0, epos0, qy0 * KS + TMD + 2);
      const int ylo0 = (-HK > -qy0) ? -HK : -qy0, yhi0 = (HK < KS - 1 - qy0) ? HK : KS - 1 - qy0;
      y_read(f0, prefix_rows(ylo0, yhi0), Wv);
    }
    step_fence();
    if constexpr (SPL) stage_sync();
  }

  // One offset row q_y (its k_s steps unrolled).  REGW: the row exists in two instantiations -- CUT = the lane exchange
  // in every step, !CUT = none at all -- and offset_rows() below walks the rows in runs of one class: the wave-uniform
  // choice is made per ROW.  Per-step branches around the exchange cost the kernel 6 % of its time on top of the
  // exchange itself (profiles/r6_bwd_dense_ablation.txt: 0.482 ms with them, 0.427 without exchange and branches,
  // 0.507 with the exchange in every row and no branch).
  auto offset_row = [&](auto cut_c, const int qyi) {
    // (cut_c: 0 = no exchange, 1 = in every step, 2 = ONE row body with a wave-uniform test per step -- the four-chunk
    // instantiation of the TILE_HUGE tiles, which spills with two row bodies)
    constexpr int CUTM = decltype(cut_c)::value;
    constexpr bool CUT = CUTM == 1;
    // (next row's prefetches run unconditionally on clamped rows: the offset loop stays branch-free)
    const int qyn = qyi + 1 < KS ? qyi + 1 : KS - 1;
    // (first offsets of this and the next offset row for the tile-major loads, opaque to the optimiser: loop strength
    // reduction otherwise keeps one running pointer per unrolled load -- 196 of them -- across the offset rows)
    int rq0 = qyi * KS, rq1 = qyn * KS;
    if constexpr (TM) asm volatile("" : "+s"(rq0), "+s"(rq1));
    float nrow[C][CPL];
    load_img_row(r0 + qyi + RR < RH ? r0 + qyi + RR : RH - 1, nrow);
    const int ylo = (-HK > -qyi) ? -HK : -qyi, yhi = (HK < KS - 1 - qyi) ? HK : KS - 1 - qyi;
    const float ymask = (ylo > -HK || yhi < HK) ? 1.f : 0.f;
    int pw = 0, l_oth = 0;
    float c_own = 0.f, c_oth = 0.f;
    if constexpr (REGW) cut_rows(qyi, l_oth, c_own, c_oth);
    else pw = prefix_rows(ylo, yhi);
    const bool cut_now = CUTM == 2 && __ballot(c_own != 1.f || c_oth != 0.f) != 0ull;
    int eposc[NCH];   // where this row's offsets put G (REGW: under the row's shift)
#pragma unroll
    for (int ck = 0; ck < NCH; ++ck) eposc[ck] = REGW ? shift_pos(epos[ck], row_shift(qyi)) : epos[ck];
    int pwn = 0;   // LAG: the row's last step fetches W of the next row's first offset
    if constexpr (LAG) pwn = prefix_rows((-HK > -qyn) ? -HK : -qyn, (HK < KS - 1 - qyn) ? HK : KS - 1 - qyn);
    float *fe = fld + (qyi & 1) * FSZ, *fo = fld + ((qyi & 1) ^ 1) * FSZ;  // copies of the even / odd q_x of this row
    const int slr = (r + qyi) & RMASK;
    const float *ib = imgb + slr * BRS + NPX * g;
    float *gb = grb + slr * BRS + NPX * g;
    f2 wA[NPX], grA[NPX];
    float wB[NPX], grB[NPX];
    float vx[NPX];   // REGW, cut rows: the other lane's prefix (in flight from the end of a step to the next one's w_finish)
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      vx[i] = 0.f;
      wA[i] = DO_A ? f2{ib[i], ib[RWS + i]} : f2{0.f, 0.f};
      wB[i] = DO_B ? ib[2 * RWS + i] : 0.f;
      grA[i] = f2{0.f, 0.f};
      grB[i] = 0.f;
    }
    static_for(std::make_integer_sequence<int, KS>{}, [&](auto qc) {
      constexpr int qxi = decltype(qc)::value;
      constexpr int xlo = (-HK > -qxi) ? -HK : -qxi, xhi = (HK < KS - 1 - qxi) ? HK : KS - 1 - qxi;
      constexpr int grp = qxi / 4, gj = qxi % 4;
      constexpr int qxn = (qxi + 1 + LAG) % KS, qx2 = (qxi + 2 + LAG) % KS;  // offsets prepared during this step (P stage, G)
      constexpr int nlo = (-HK > -qxn) ? -HK : -qxn, nhi = (HK < KS - 1 - qxn) ? HK : KS - 1 - qxn;
      constexpr int M0 = HK - nhi, M1 = HOUT - 1 + HK - nlo;
      float *fc = (qxi & 1) ? fo : fe, *fn = (qxi & 1) ? fe : fo;  // (k_s odd: the next row's offsets continue the alternation)
      float *fP = LAG ? fc : fn, *fG = LAG ? fn : fc;              // copies of the offsets qxn / qx2
      // G prefetch, two groups ahead: group grp+2 (of the next row past the end) replaces group grp, whose last
      // value went to LDS a step ago; the row's last offset on its own
      if constexpr (gj == 2 && grp < NG) {
        if constexpr (grp + 2 < NG) load_group(qyi, grp + 2, gbuf[grp & 1]);
        else load_group(qyn, grp + 2 - NG, gbuf[grp & 1]);
      }
      if constexpr (qxi == KS - 2) load_last(qyn);
      // ---- top: every LDS read of the step ----
      float v[HOUT + 2 * HK], fl[C], wn[C];
      if constexpr (LAG) {
      } else if constexpr (REGW) {
        // (cut rows: the exchange for this step was issued at the end of the previous one, right behind the prefix --
        // only the row's first step issues its own)
        if constexpr (CUT && qxi == 0) w_exchange(l_oth, vx);
        if constexpr (CUTM == 2) {
          if (cut_now) w_exchange(l_oth, vx);
        }
      } else {
        y_read(fc, pw, Wv);
      }
      if constexpr (DO_W) {
#pragma unroll
        for (int m = M0; m <= M1; ++m) v[m] = fP[hsrc + m];
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        fl[c] = chan_on(c) ? gb[c * RWS + qxi] : 0.f;
        wn[c] = (qxi + 1 < KS && chan_on(c)) ? ib[c * RWS + qxi + NPX] : 0.f;
      }
      // LAG: W of the next offset (its rows were written a step ago), into registers of its own
      float Wn[NPX];
      if constexpr (LAG) y_read(fn, qxi + 1 < KS ? pw : pwn, Wn);
      // (two waves: the reads stay here, in front of the body that covers their latency -- the scheduler otherwise
      // sinks them to their first use behind it)
      // (the reads stay here, in front of the body that covers their latency -- the scheduler otherwise sinks them to
      // their first use behind it (SPL), or pulls the horizontal sums that consume v[] up in front of the body (REGW:
      // a wait for the whole read burst at the top of every step; round 6))
      if constexpr (SPL || REGW) __builtin_amdgcn_sched_barrier(0);
      // ---- both ends of every pair (u, u+q) ----
      constexpr bool xborder = xlo > -HK || xhi < HK;
      // the pixel differences first: they do not depend on W (REGW: they cover the cut rows' lane exchange)
      f2 dA[NPX];
      float dB[NPX];
      constexpr bool DIFF_FIRST = REGW && NCH <= 2;   // (the four-chunk instantiation is at its register budget: differences next to their use)
      if constexpr (DIFF_FIRST) {
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
          const int sl = (i + qxi) % NPX;
          dA[i] = iuA[i] - wA[sl];
          dB[i] = iuB[i] - wB[sl];
          // (pinned in front of the cut rows' branch: machine sinking otherwise moves them behind it, and the branch
          // then waits for the lane exchange with nothing in between)
          asm volatile("" : "+v"(dA[i]), "+v"(dB[i]));
        }
        w_finish(CUTM == 2 ? !cut_now : !CUT, vx, c_own, c_oth, Wv);
      } else if constexpr (REGW) {
        w_finish(CUTM == 2 ? !cut_now : !CUT, vx, c_own, c_oth, Wv);
      }
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const float Wi = Wv[i];
        if constexpr (DO_S) {
          if constexpr (xborder) swb[i] += Wi;
          else if constexpr (CUTM == 3) {}   // (a full offset row: its x-interior offsets are no border offsets)
          else swb[i] = __builtin_fmaf(Wi, ymask, swb[i]);
        }
        const int sl = (i + qxi) % NPX;
        if constexpr (!DIFF_FIRST) {   // (the other lane maps form the differences next to their use: fewer live registers)
          dA[i] = DO_A ? iuA[i] - wA[sl] : f2{0.f, 0.f};
          dB[i] = DO_B ? iuB[i] - wB[sl] : 0.f;
        }
        if constexpr (DO_B) {
          guB[i] = __builtin_fmaf(Wi, dB[i], guB[i]);
          grB[sl] = __builtin_fmaf(-Wi, dB[i], grB[sl]);
        }
        if constexpr (!DO_A) {
        } else if constexpr (RG == 4) {   // measured: packed 2.27 -> 2.06 ms for (49,13); 0.31 -> 0.33 ms for (25,9), which stays scalar
          const f2 Wi2 = f2{Wi, Wi};
          guA[i] = __builtin_elementwise_fma(Wi2, dA[i], guA[i]);
          grA[sl] = __builtin_elementwise_fma(-Wi2, dA[i], grA[sl]);
        } else {
          guA[i].x = __builtin_fmaf(Wi, dA[i].x, guA[i].x);
          guA[i].y = __builtin_fmaf(Wi, dA[i].y, guA[i].y);
          grA[sl].x = __builtin_fmaf(-Wi, dA[i].x, grA[sl].x);
          grA[sl].y = __builtin_fmaf(-Wi, dA[i].y, grA[sl].y);
        }
      }
      // ---- next offset: horizontal sums over its column taps, vertical prefix ----
      if constexpr (REGW) __builtin_amdgcn_sched_barrier(0);
      float out[HOUT];
      if constexpr (DO_W) {
        window_sums<HOUT, HK - nhi, HK - nlo>(v, out);
        if constexpr (REGW && HOUT == 5) {
          dpp_prefix8x5(out);
        } else {
#pragma unroll
          for (int i = 0; i < HOUT; ++i) {
            out[i] = __builtin_fmaf(dpp_row_shr<1>(out[i]), m1, out[i]);
            if constexpr (TY > 2) out[i] = __builtin_fmaf(dpp_row_shr<2>(out[i]), m2, out[i]);
            if constexpr (TY > 4) out[i] = __builtin_fmaf(dpp_row_shr<4>(out[i]), m4, out[i]);
          }
        }
      }
      // ---- end: every LDS write of the step ----
      if constexpr (!DO_W) {
      } else if constexpr (REGW) {
#pragma unroll
        for (int i = 0; i < HOUT; ++i) wout[i] = out[i];
        // cut rows: the next step's lane exchange starts here -- the step's LDS writes, the next step's reads and its
        // pixel differences run while it is in flight
        if constexpr (CUT && qxi + 1 < KS) w_exchange(l_oth, vx);
      } else {
#pragma unroll
        for (int i = 0; i < HOUT; ++i) {
          fP[hdst[i]] = out[i];
          fP[hdst2[i]] = suffix_of(out[i]);
        }
      }
      // region column NPX*g + qxi of this band row is complete: to the band; the window moves on
      {
        constexpr int s0 = qxi % NPX;
        if constexpr (DO_A) {
          gb[qxi] = fl[0] + grA[s0].x;
          gb[RWS + qxi] = fl[1] + grA[s0].y;
          grA[s0] = f2{0.f, 0.f};
          if constexpr (qxi + 1 < KS) wA[s0] = f2{wn[0], wn[1]};
        }
        if constexpr (DO_B) {
          gb[2 * RWS + qxi] = fl[2] + grB[s0];
          grB[s0] = 0.f;
          if constexpr (qxi + 1 < KS) wB[s0] = wn[2];
        }
      }
      if constexpr (REGW && qxi + 2 + LAG >= KS) {
        // the next row's first offsets replace this row's last ones in their copies: under another shift at other
        // positions -- the old ones are cleared first (same positions when the shifts agree: overwritten below)
        int eposn[NCH];
#pragma unroll
        for (int ck = 0; ck < NCH; ++ck) {
          fG[eposc[ck]] = 0.f;
          eposn[ck] = shift_pos(epos[ck], row_shift(qyn));
        }
        x_put(std::integral_constant<int, qx2>{}, qyn, fG, eposn);
      } else
      x_put(std::integral_constant<int, qx2>{}, qxi + 2 + LAG < KS ? qyi : qyn, fG, eposc,
            (qxi + 2 + LAG < KS && qx2 + TMD < KS) ? rq0 + qx2 + TMD : (qxi + 2 + LAG < KS ? rq1 + qx2 + TMD - KS : rq1 + qx2 + TMD));
      // (the accumulators of the lane's own pixels pass through an empty asm: they are not read again before
      // the end of the sweep, and hipcc otherwise sinks their FMAs below all k_s steps, keeping every step's
      // W and differences alive: 1,000 spilled registers)
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        if constexpr (DO_A) asm volatile("" : "+v"(guA[i]));
        if constexpr (DO_B) asm volatile("" : "+v"(guB[i]));
      }
      if constexpr (DO_S) pin_row<NPX>(swb);
      if constexpr (LAG) {
#pragma unroll
        for (int i = 0; i < NPX; ++i) Wv[i] = Wn[i];
      }
      step_fence();
      if constexpr (SPL) stage_sync();
    });
    // the row's last NPX-1 columns
#pragma unroll
    for (int i = 1; i < NPX; ++i)
    {
      if constexpr (DO_A) {
        gb[KS - 1 + i] += grA[(i + KS - 1) % NPX].x;
        gb[RWS + KS - 1 + i] += grA[(i + KS - 1) % NPX].y;
      }
      if constexpr (DO_B) gb[2 * RWS + KS - 1 + i] += grB[(i + KS - 1) % NPX];
    }
    __builtin_amdgcn_wave_barrier();
    flush_row(r0 + qyi);
    store_img_row(r0 + qyi + RR, nrow);
    __builtin_amdgcn_wave_barrier();
  };
  if constexpr (!REGW) {
#pragma unroll 1
    for (int qyi = qy0; qyi < qy1; ++qyi) offset_row(std::integral_constant<int, 0>{}, qyi);
  } else if constexpr (NCH > 2) {
#pragma unroll 1
    for (int qyi = qy0; qyi < qy1; ++qyi) offset_row(std::integral_constant<int, 2>{}, qyi);
  } else {
    // a row is SIMPLE when every lane's W is its own prefix: full windows, and the cut rows next to them whose cut
    // does not reach into the tile (q_y = 21 for the top half, 3 for the bottom half)
    auto row_simple = [&](int qyi) {
      int l_oth;
      float c_own, c_oth;
      cut_rows(qyi, l_oth, c_own, c_oth);
      return __ballot(c_own != 1.f || c_oth != 0.f) == 0ull;
    };
    // (a third row body for the FULL rows -- window not cut at all: no border sums in their x-interior steps, 5 VALU
    // instructions of ~97 in 17 of 25 steps of 17 of 25 rows)
    auto row_full = [&](int qyi) { return qyi >= HK && qyi <= KS - 1 - HK; };
    int qyi = qy0;
#pragma unroll 1
    while (qyi < qy1) {
      if (row_full(qyi)) {
#pragma unroll 1
        do {
          offset_row(std::integral_constant<int, 3>{}, qyi);
          ++qyi;
        } while (qyi < qy1 && row_full(qyi));
      } else if (row_simple(qyi)) {
#pragma unroll 1
        do {
          offset_row(std::integral_constant<int, 0>{}, qyi);
          ++qyi;
        } while (qyi < qy1 && row_simple(qyi) && !row_full(qyi));
      } else {
#pragma unroll 1
        do {
          offset_row(std::integral_constant<int, 1>{}, qyi);
          ++qyi;
        } while (qyi < qy1 && !row_simple(qyi));
      }
    }
  }
  for (int rho = r0 + qy1; rho < r0 + qy1 + RR - 1; ++rho) flush_row(rho);

  // ---- the lane's own pixels: + I * sum_q V_q (Box(sum_b) - sum of the border W_q), then to HBM ----
  float vbox[NPX];
  {
    if constexpr (DO_G) {
      // (REGW: the sweep's last offsets may have been stored under a row shift: the copy is cleared, pads included)
      if constexpr (REGW) {
        for (int i = lane; i < TY * GQS; i += 64) fld[i] = 0.f;
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int ck = 0; ck < NCH; ++ck) fld[epos[ck]] = (blockIdx.y == 0) ? p.sum_b[erow[ck]] : 0.f;
    }
    stage_sync();
    if constexpr (DO_W) x_stage(std::integral_constant<int, -HK>{}, std::integral_constant<int, HK>{}, fld);
    if constexpr (SPL && DO_S) {   // the border sums for the other role (the gradient band is flushed: scratch)
#pragma unroll
      for (int i = 0; i < NPX; ++i) grb[64 * i + lane] = swb[i];
    }
    stage_sync();
    if constexpr (SPL && !DO_S) {
#pragma unroll
      for (int i = 0; i < NPX; ++i) swb[i] = grb[64 * i + lane];
    }
    if constexpr (REGW) {
      w_finish(true, vbox, 0.f, 0.f, vbox);
    } else {
      y_read(fld, prefix_rows(-HK, HK), vbox);
    }
    step_fence();
  }

  {
    const int py = ty0 - HK + r;
    const int gy = reflect_idx(py, H);
    const bool yok = gy >= 0 && gy < H;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int gx = reflect_idx(tx0 - HK + NPX * g + i, W);
      const bool ok = yok && gx >= 0 && gx < W;
      const float vt = vbox[i] - swb[i];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (!chan_on(c)) continue;
        const float iuc = c == 0 ? iuA[i].x : c == 1 ? iuA[i].y : iuB[i], guc = c == 0 ? guA[i].x : c == 1 ? guA[i].y : guB[i];
        const float v = __builtin_fmaf(iuc, vt, guc);
        if (ok && v != 0.f && !SSG_DBG(p, 8)) grad_add(p.grad, p.gfix, (((size_t)b * C + c) * H + gy) * W + gx, 2.f * v, gsc);
      }
    }
  }
}

template <int KS, int KW, int C, int TY, int NCH, int RG, bool TM = false, bool SPL = false>
__global__ __launch_bounds__(SPL ? 128 : 64)
__attribute__((amdgpu_waves_per_eu((RG == 8 || SPL) ? 2 : 1, (RG == 8 || SPL) ? 2 : 1))) void ssg_bwd_dense(DenseBwdParams p) {
  if constexpr (!SPL) {
    bwd_dense_body<KS, KW, C, TY, NCH, RG, TM, false, 2>(p);
  } else if (threadIdx.x < 64) {
    bwd_dense_body<KS, KW, C, TY, NCH, RG, TM, true, 0>(p);
  } else {
    bwd_dense_body<KS, KW, C, TY, NCH, RG, TM, true, 1>(p);
  }
}

// (25,9): BOTH chunk classes of the 8 x 32 tiles in one launch (round 6).  The two instantiations used to be launched one
// after the other over the tile list, a tile running in its own; C2 and C4 hold no TILE_HUGE tile, so the second launch was
// ~5 us of workgroups that start only to leave, on the step's critical path.  Both bodies run at two waves per SIMD (236
// and 238 registers), so one kernel that picks the body by the plan's class bit costs neither occupancy.
template <int KS, int KW, int C, int TY, int RG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void ssg_bwd_dense_classes(DenseBwdParams p) {
  static_assert(TY == 8 && RG == 8, "the 8 x 32 tiles of k_s <= 25");
  const int tslot = blockIdx.x;
  if (tslot >= dense_tile_count(p.n_dense)) return;
  const int tx_n = (p.W + 31) / 32, ty_n = (p.H + TY - 1) / TY;
  const int listed = dense_tile_at(p.n_dense, p.tiles, p.B * ty_n * tx_n, tslot);
  if (__builtin_amdgcn_readfirstlane(listed) & TILE_HUGE) bwd_dense_body<KS, KW, C, TY, 4, RG, false, false, 2>(p);
  else bwd_dense_body<KS, KW, C, TY, 2, RG, false, false, 2>(p);
}

// ------------------------------------------------------------------ host ----
bool dense_bwd_supported(int ks, int kw, int C) { return C == 3 && ((ks == 25 && kw == 9) || (ks == 49 && kw == 13)); }

// qsplit 0: the grid carries this many offset-row parts per tile, the device uses 1 .. QSPLIT_AUTO_MAX of them
constexpr int QSPLIT_AUTO_MAX = 5;

template <int KS, int KW, int C, int TY, int NCH, int RG, bool TM = false, bool SPL = false>
static int launch_one(const DenseBwdParams &p, int n_tiles, hipStream_t st) {
  using G = DenseBwdGeo<KS, KW, C, TY, RG>;
  if (n_tiles <= 0) return 0;
  static std::atomic<unsigned long long> lds_set{0};
  if (const int rc = ensure_dynamic_lds(ssg_bwd_dense<KS, KW, C, TY, NCH, RG, TM, SPL>, (int)G::lds_bytes(), lds_set)) return rc;
  hipLaunchKernelGGL((ssg_bwd_dense<KS, KW, C, TY, NCH, RG, TM, SPL>),
                     dim3((unsigned)n_tiles, (unsigned)(p.qsplit > 0 ? p.qsplit : QSPLIT_AUTO_MAX), G::NHALF), dim3(SPL ? 128 : 64),
                     G::lds_bytes(), st, p);
  return (int)hipGetLastError();
}

int launch_bwd_dense(const DenseBwdParams &p0, int ks, int kw, int C, hipStream_t st) {
  if (!dense_bwd_supported(ks, kw, C)) return -1;
  if (p0.max_tiles == 0) return 0;
  DenseBwdParams p = p0;
  if (p.qsplit <= 0) {   // wave slots of the device for this kernel: (25,9) 2 waves per SIMD, (49,13) one
    static std::atomic<int> cus{0};
    int n = cus.load(std::memory_order_relaxed);
    if (n <= 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;
      cus.store(n, std::memory_order_relaxed);
    }
    p.qsplit = 0;
    p.auto_slots = n * 4 * (ks == 49 ? 1 : 2);
  }
  // 4 x 32 tiles: at most 128 edge pixels.  One wave per tile: the two-waves-per-tile layout (RG = 8, 48 padded
  // columns, prefix rows through LDS) was measured slower here, 3.16 vs 2.27 ms at C5 -- with 4 tile rows each half
  // repeats a W stage that is most of its work.  With a tile-major region (fused step) both variants are launched
  // and tm_active() lets one of them run.
  const bool tm = ks == 49 && p.tm[0] && p.tm[1] && p.row_scale && p.dot && p.tm_slots > 0;
  if (!tm) p.tm_slots = 0;
  if (ks == 49) {
    // (tile-major rows: the role-split two-wave instantiation; round 3's one-wave kernel -- 256 VGPR + 136 AGPR, 2.50 ms
    // against 1.98 at C5 -- is no longer instantiated)
    int rc = !tm ? 0 : launch_one<49, 13, 3, 4, 2, 4, true, true>(p, p.tm_slots < p.max_tiles ? p.tm_slots : p.max_tiles, st);
    if (!rc) rc = launch_one<49, 13, 3, 4, 2, 4, false>(p, p.max_tiles, st);
    return rc;
  }
  static const bool one_launch = env_int("SSG_DENSE_ONE_LAUNCH", 1) != 0;   // (profiling build: 0 = a launch per class, A/B)
  if (one_launch) {
    using G = DenseBwdGeo<25, 9, 3, 8, 8>;
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(ssg_bwd_dense_classes<25, 9, 3, 8, 8>, (int)G::lds_bytes(), lds_set)) return rc;
    hipLaunchKernelGGL((ssg_bwd_dense_classes<25, 9, 3, 8, 8>),
                       dim3((unsigned)p.max_tiles, (unsigned)(p.qsplit > 0 ? p.qsplit : QSPLIT_AUTO_MAX), G::NHALF), dim3(64),
                       G::lds_bytes(), st, p);
    return (int)hipGetLastError();
  }
  int rc = launch_one<25, 9, 3, 8, 2, 8>(p, p.max_tiles, st);
  // (TILE_HUGE tiles are heavy ones -- more than 64 rows each, at the front of the list: see launch_fwd_dense_25)
  const int heavy_max = p.n_host / 65 + 1;
  if (!rc) rc = launch_one<25, 9, 3, 8, 4, 8>(p, heavy_max < p.max_tiles ? heavy_max : p.max_tiles, st);
  return rc;
}

}  // namespace ssg
