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
__device__ __forceinline__ void load_row(const float *tc, const float *zrow, int ry, int cx0,
                                         const bool (&colv)[G::PW], float (&out)[G::PW]) {
  const float *rowp = ((unsigned)ry < (unsigned)G::KS) ? (tc + ry * G::S) : zrow;
#pragma unroll
  for (int j = 0; j < G::PW; ++j) {
    const float v = rowp[cx0 + j];
    out[j] = colv[j] ? v : 0.f;
  }
}

template <class G>
__global__ __launch_bounds__(G::WG) void ssg_fwd_tiled(FwdParams p) {
  constexpr int KS = G::KS, KW = G::KW, BS = G::BS, WG = G::WG;
  constexpr int HP = G::HP, HK = G::HK, P = G::P, NB = G::NB, LPJ = G::LPJ;
  constexpr int JOBS = G::JOBS, PW = G::PW, S = G::S, CH = G::CH;
  constexpr int PADF = (HK + 3) & ~3;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = p.C, H = p.H, W = p.W;
  float *tiles = smem + PADF;                  // [JOBS][C][KS][S]
  float *zero = tiles + JOBS * C * CH;         // ZROW zeros (also absorbs tail over-reads)
  float *red = zero + ((G::ZROW + 3) & ~3);    // [WG] row-sum scratch
  int *sh_edge = (int *)(red + WG);            // [JOBS][4]: b, y, x, valid

  const int tid = threadIdx.x;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const int njobs = nrows * p.nimg;
  const int job0 = blockIdx.x * JOBS;
  if (job0 >= njobs) return;

  if (tid < JOBS) {
    const int q = job0 + tid;
    const bool v = q < njobs;
    const int n = v ? q / p.nimg : 0;
    Edge e = load_edge(p.edges, p.estride, n);
    sh_edge[tid * 4 + 0] = e.b;
    sh_edge[tid * 4 + 1] = e.y;
    sh_edge[tid * 4 + 2] = e.x;
    sh_edge[tid * 4 + 3] = v ? 1 : 0;
  }
  for (int i = tid; i < G::ZROW + 4; i += WG) zero[i] = 0.f;
  if (tid < PADF) smem[tid] = 0.f;
  __syncthreads();

  // ---- fill: JOBS x C x KS x KS floats, reflect by index mirroring.  All global loads of a
  // job are issued before the first LDS store (a plain loop pays the full L2 latency per
  // element: the fill was 29 % of the kernel) ----
  constexpr int EPT = (P + WG - 1) / WG;  // tile elements per thread per channel
  for (int j = 0; j < ((p.dbg & 1) ? 0 : JOBS); ++j) {
    const int q = job0 + j;
    const int which = q < njobs ? q % p.nimg : 0;
    const float *src = p.img[which];
    const int b = sh_edge[j * 4 + 0], y = sh_edge[j * 4 + 1], x = sh_edge[j * 4 + 2];
    const float *s0[EPT];
    int d0[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * WG;
      const int ec = e < P ? e : P - 1;
      const int ry = ec / KS, rx = ec - ry * KS;
      s0[k] = src + ((size_t)b * C * H + reflect_idx(y - HP + ry, H)) * W + reflect_idx(x - HP + rx, W);
      d0[k] = e < P ? (j * C) * CH + ry * S + rx : -1;
    }
    if (C == 3) {
      float v[EPT][3];
#pragma unroll
      for (int k = 0; k < EPT; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[k][c] = s0[k][(size_t)c * H * W];
#pragma unroll
      for (int k = 0; k < EPT; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (d0[k] >= 0) tiles[d0[k] + c * CH] = v[k][c];
    } else {
      for (int c = 0; c < C; ++c) {
        float v[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) v[k] = s0[k][(size_t)c * H * W];
#pragma unroll
        for (int k = 0; k < EPT; ++k)
          if (d0[k] >= 0) tiles[d0[k] + c * CH] = v[k];
      }
    }
  }
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
#pragma unroll 1
  for (int c = 0; c < ((p.dbg & 2) ? 0 : C); ++c) {
    const float *tc = tiles + (jl * C + c) * CH;
    if constexpr (KW <= 9) {
      float a[KW][KW];  // centre window of this channel (uniform across the job's lanes)
#pragma unroll
      for (int kh = 0; kh < KW; ++kh)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) a[kh][kx] = tc[(HP - HK + kh) * S + (HP - HK + kx)];
      // software pipeline: patch row r+1 is in flight while row r is consumed;
      // pin_block keeps hipcc from hoisting every row's loads to the top (which blew
      // the VGPR budget and spilled ~380 dwords per lane).
      float bn[PW];
      load_row<G>(tc, zrow, ry0, cx0, colv, bn);
#pragma unroll
      for (int r = 0; r < PW; ++r) {
        float bv[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) bv[j] = bn[j];
        if (r + 1 < PW) load_row<G>(tc, zrow, ry0 + r + 1, cx0, colv, bn);
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
      // large windows: the centre row is re-read (LDS broadcast) per patch row
#pragma unroll 1
      for (int r = 0; r < PW; ++r) {
        const int ry = ry0 + r;
        const float *rowp = ((unsigned)ry < (unsigned)KS) ? (tc + ry * S) : zrow;
        float bv[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) {
          const float v = rowp[cx0 + j];
          bv[j] = colv[j] ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < BS; ++i) {
          const int kh = r - i;
          if (kh < 0 || kh >= KW) continue;  // runtime r: predicated
          const float *ar = tc + (HP - HK + kh) * S + (HP - HK);
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) {
            const float av = ar[kx];
#pragma unroll
            for (int j = 0; j < BS; ++j) {
              const float d = av - bv[j + kx];
              acc[i][j] = __builtin_fmaf(d, d, acc[i][j]);
            }
          }
        }
      }
    }
  }

  const int q = job0 + jl;
  const bool job_on = lane_on && q < njobs;
  const int n = job_on ? q / p.nimg : 0;
  const int which = job_on ? q % p.nimg : 0;

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
    return;
  }

  if (p.dbg & 4) {
    if (acc[0][0] == 123.456f) p.out[0][0] = acc[1][1];
    return;
  }
  // ---- epilogue: e = exp(-(D/den)/sigma), row sum, normalise ----
  // -(D/(C k_w^2))/sigma as one multiply by a host-rounded constant: |x| differs from the
  // reference's two divisions by <= 1.5 ulp, i.e. e by < 1e-5 relative even at e ~ 1e-38
  const float nk = -1.f / ((float)(C * KW * KW) * p.sigma);
  float lsum = 0.f;
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      const int py = BS * by + i, px = BS * bx + j;
      const float e = (py < KS && px < KS) ? expf(acc[i][j] * nk) : 0.f;
      acc[i][j] = e;
      lsum += e;
    }
  red[tid] = lsum;
  __syncthreads();  // also: every lane is done reading the tiles -> reuse as staging
  float scale = 1.f;
  if (p.generalization) {
    float tot = 0.f;
    for (int k = 0; k < LPJ; ++k) tot += red[jl * LPJ + k];
    scale = 1.f / (tot + p.eps);
  }
  float *stage = tiles + (jl * C) * CH;  // >= P floats per job
  if (lane_on) {
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const int py = BS * by + i, px = BS * bx + j;
        if (py < KS && px < KS) stage[py * KS + px] = scale * acc[i][j];
      }
  }
  __syncthreads();
  for (int j = 0; j < JOBS; ++j) {
    const int qq = job0 + j;
    if (qq >= njobs) break;
    float *o = p.out[qq % p.nimg] + (size_t)(qq / p.nimg) * P;
    const float *sj = tiles + (j * C) * CH;
    for (int e = tid; e < P; e += WG) o[e] = sj[e];
  }
}

// Any odd (ks, kw): one workgroup (256 lanes) per job, tile in LDS.
__global__ __launch_bounds__(256) void ssg_fwd_generic(FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ks = p.ks, kw = p.kw, hp = ks / 2, hk = kw / 2, P = ks * ks;
  const int C = p.C, H = p.H, W = p.W, tid = threadIdx.x;
  float *tile = smem;        // [C][ks][ks]
  float *red = smem + C * P; // [256]
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
  float lsum = 0.f;
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
      lsum += ev;
    }
  }
  if (p.raw || !p.generalization) return;
  red[tid] = lsum;
  __syncthreads();
  float tot = 0.f;
  for (int k = 0; k < 256; ++k) tot += red[k];
  const float scale = 1.f / (tot + p.eps);
  for (int pidx = tid; pidx < P; pidx += 256) o[pidx] = scale * o[pidx];
}

// ------------------------------------------------------------------ host ----
template <class G>
static size_t fwd_lds_bytes(int C) {
  constexpr int PADF = (G::HK + 3) & ~3;
  return sizeof(float) * (size_t)(PADF + G::JOBS * C * G::CH + ((G::ZROW + 3) & ~3) + 4 + G::WG) +
         sizeof(int) * 4 * G::JOBS;
}

template <class G>
static int launch_fwd_tiled(const FwdParams &p, hipStream_t st) {
  const size_t lds = fwd_lds_bytes<G>(p.C);
  if (lds > 160 * 1024) return -2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)ssg_fwd_tiled<G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  const long njobs = (long)p.n_host * p.nimg;
  if (njobs == 0) return 0;
  const unsigned grid = (unsigned)((njobs + G::JOBS - 1) / G::JOBS);
  hipLaunchKernelGGL(ssg_fwd_tiled<G>, dim3(grid), dim3(G::WG), lds, st, p);
  return (int)hipGetLastError();
}

int launch_fwd(const FwdParams &p, hipStream_t st) {
  if (p.ks == 25 && p.kw == 9) return launch_fwd_tiled<Geo<25, 9, 5, 128>>(p, st);
  if (p.ks == 11 && p.kw == 5) return launch_fwd_tiled<Geo<11, 5, 4, 64>>(p, st);
  if (p.ks == 49 && p.kw == 13 && fwd_lds_bytes<Geo<49, 13, 7, 128>>(p.C) <= 160 * 1024)
    return launch_fwd_tiled<Geo<49, 13, 7, 128>>(p, st);
  const size_t lds = sizeof(float) * ((size_t)p.C * p.ks * p.ks + 256);
  if (lds > 160 * 1024) return -2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)ssg_fwd_generic, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const long njobs = (long)p.n_host * p.nimg;
  if (njobs == 0) return 0;
  hipLaunchKernelGGL(ssg_fwd_generic, dim3((unsigned)njobs), dim3(256), lds, st, p);
  return (int)hipGetLastError();
}

const char *fwd_kernel_name(int ks, int kw) {
  if (ks == 25 && kw == 9) return "ssg_fwd_tiled<Geo<25,9,5,128>>";
  if (ks == 11 && kw == 5) return "ssg_fwd_tiled<Geo<11,5,4,64>>";
  if (ks == 49 && kw == 13) return "ssg_fwd_tiled<Geo<49,13,7,128>>";
  return "ssg_fwd_generic";
}

}  // namespace ssg
