// The fused loss step of SMALL calls at (k_s, k_w, C) = (11, 5, 3) in two launches (round 6, review item 5).
//
// BASELINE's configs[0] (C1: 1 x 3 x 64 x 64, 209 edge pixels -- the reference's own CPU-runnable case, loss_util.py:185-229
// + basic_loss.py:66,281 on a 64 x 64 crop) took 59 us as six dependent launches: three for the edge list, the direct
// forward (15 us), the direct backward (27 us), the flush.  Those kernels give an edge pixel 9 lanes and a workgroup 7 edge
// pixels, three channels in sequence: with 209 rows the chip holds 30 workgroups whose own latency chain is the step.
// Here an edge pixel gets a WHOLE 256-lane workgroup and the step two launches:
//   tiny_edge_list (ssg_edges.hip)  one workgroup: clears the fixed-point sums / the gradient / the ticket, builds rows,
//                                   rank map and counts;
//   ssg_tiny_step                   one workgroup per edge pixel (looping when there are more rows than workgroups):
//       forward      lane = (image, offset): 2 x 121 of 256 lanes, 75 (sub, fma) each on the two 3 x 11 x 11 search tiles in
//                    LDS (reflect padding = index mirroring in the fill; "B = 0 outside the search area",
//                    similarity.cu:43-47); e = exp2(D nk), fp64 row sum over two waves, s = fl32(e / (sum + eps))
//                    (loss_util.py:224-227) -- the arithmetic of ssg_fwd.hip's epilogue, operation by operation;
//       criteria     lane = offset: g = dL/ds of L1 + KL (criteria_elem, ssg_common.hpp), sum g s, G = dL/dD, the row's
//                    |a-b| and t log(t/s) sums into its slot of `partials` (ssg_bwd.hip's stage 1);
//       backward     gS[c,t] and gA[c,k] of ssg_bwd.hip's header in their direct form (what ssg_bwd_generic computes):
//                    363 tile positions x 25 taps, 75 window positions x 121 offsets split three ways; one atomic per
//                    touched pixel and channel (64-bit integer sums in deterministic mode, at the a-priori scale);
//       finish       the workgroup that draws the last ticket sums the criteria slots in fp64 in a fixed order
//                    (-> loss_out) and turns the fixed-point sums into the gradient (ssg_bwd.hip's flush + finalize).
// Same results as the six-launch path to rounding (different summation orders; parity against the oracle and the
// reference's fixture F1 in tests/test_gpu_parity.py / test_gpu_tiny.py, SSG rows bit-identical to the general path's in 9,000 fuzz
// cases; bit-reproducible in deterministic mode).  Taken by
// ssg_loss_fwd_bwd / ssg_loss_step when B H W <= 16,384 pixels and capacity <= 4,096 rows (ssg_api.hip); ssg_set_tiny_step(0)
// turns it off.
#include "ssg_common.hpp"

namespace ssg {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int KS, int KW, int C>
__global__ __launch_bounds__(256) void ssg_tiny_step(TinyParams p) {
  constexpr int HP = KS / 2, HK = KW / 2, P = KS * KS, K2 = KW * KW;
  constexpr int NPART = 3, ROWS_PER_PART = (KS + NPART - 1) / NPART;
  static_assert(P <= 128 && C * K2 * NPART <= 256 && C * K2 <= 128, "lane maps: two waves per image row, 225 lanes in the window pass");
  __shared__ float tile[2][C][P];      // search tiles of sr / gt
  __shared__ float sb[2][128];         // the two SSG rows
  __shared__ float Gt[P];              // G = dL/dD (centre offset 0)
  __shared__ float gst[C][P];          // gradient of the tile
  __shared__ float wpart[NPART][C * K2];
  __shared__ double dred[4];
  __shared__ float fred[2][4];
  __shared__ int s_last;
  __shared__ double s1[256], s2[256];

  const int tid = threadIdx.x, wv = tid >> 6;
  const int H = p.H, W = p.W;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const float nk = (float)(-1.4426950408889634 / ((double)(C * K2) * (double)p.sigma));
  const float kfac = 1.f / (p.sigma * (float)(C * K2));
  const float invM = 1.f / ((float)nrows * (float)P);
  const float w1m = p.w_l1 * invM, w2m = p.w_kl * invM;
  const bool fix = p.gfix != nullptr && p.grad != nullptr;
  const size_t n_grad = (size_t)p.B * C * H * W;
  float gsc = 1.f;
  if (fix) gsc = grad_fix_scale_of(__float_as_uint(loss_grad_bound(p.sigma, C, KW, p.w_l1, p.w_kl, nullptr, nrows, P)));

  // forward lane: image `which` (waves 0,1 / 2,3), offset pidx
  const int which = tid >> 7, pidx = tid & 127;
  const bool f_on = pidx < P;
  const int py = f_on ? pidx / KS : 0, px = f_on ? pidx - py * KS : 0;

#pragma unroll 1
  for (int n = blockIdx.x; n < (SSG_DBG(p, 16) ? 0 : nrows); n += gridDim.x) {
    const Edge e = load_edge(p.edges, 3, n);
    // ---- the two search tiles (reflect by index mirroring) ----
    {
      constexpr int NE = 2 * C * P, EPT = (NE + 255) / 256;
      float v[EPT];
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int i = tid + 256 * k, ic = i < NE ? i : 0;
        const int w = ic / (C * P), r0 = ic - w * (C * P), c = r0 / P, r = r0 - c * P, ry = r / KS, rx = r - ry * KS;
        v[k] = p.img[w][(((size_t)e.b * C + c) * H + reflect_idx(e.y - HP + ry, H)) * W + reflect_idx(e.x - HP + rx, W)];
      }
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int i = tid + 256 * k;
        if (i < NE) (&tile[0][0][0])[i] = v[k];
      }
    }
    __syncthreads();

    // ---- forward: D of the lane's offset, e, row sum, s ----
    float acc = 0.f;
    {
      const float *tw = &tile[which][0][0];
      if (!SSG_DBG(p, 1))
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int kh = 0; kh < KW; ++kh)
#pragma unroll
          for (int kx = 0; kx < KW; ++kx) {
            const float a = tw[c * P + (HP - HK + kh) * KS + (HP - HK + kx)];
            const int yy = py + kh - HK, xx = px + kx - HK;
            const bool in = (unsigned)yy < (unsigned)KS && (unsigned)xx < (unsigned)KS;
            const float bv = tw[c * P + (in ? yy * KS + xx : 0)];
            const float d = in ? a - bv : a;
            acc = __builtin_fmaf(d, d, acc);
          }
    }
    const float ev = f_on ? __builtin_amdgcn_exp2f(acc * nk) : 0.f;
    {
      const double ws = wave_sum_f64((double)ev);
      if ((tid & 63) == 0) dred[wv] = ws;
    }
    __syncthreads();
    float sv = ev;
    if (p.generalization) {
      const double tot = dred[2 * which] + dred[2 * which + 1];
      sv = (float)((double)ev * (1.0 / (tot + (double)p.eps)));
    }
    if (f_on) {
      sb[which][pidx] = sv;
      if (p.out[which]) p.out[which][(size_t)n * P + pidx] = sv;
    }
    __syncthreads();

    // ---- criteria and G (lanes 0 .. P-1: waves 0, 1) ----
    float g = 0.f, a = 0.f, l1p = 0.f, klp = 0.f;
    if (tid < P) {
      a = sb[0][tid];
      g = criteria_elem(a, sb[1][tid], w1m, w2m, l1p, klp);
    }
    if (wv < 2) {
      const float d = wave_sum(g * a), l1s = wave_sum(l1p), kls = wave_sum(klp);
      if ((tid & 63) == 0) {
        fred[wv][0] = d;
        fred[wv][1] = l1s;
        fred[wv][2] = kls;
      }
    }
    __syncthreads();
    float Gv = 0.f;
    if (tid < P) {
      const float dot = p.generalization ? fred[0][0] + fred[1][0] : 0.f;
      Gv = tid == HP * KS + HP ? 0.f : -(a * kfac) * (g - dot);   // (the centre offset multiplies A - B == 0: dropped, see ssg_bwd.hip)
      Gt[tid] = Gv;
    }
    if (tid == 0) {
      p.partials[2 * (size_t)n] = fred[0][1] + fred[1][1];
      p.partials[2 * (size_t)n + 1] = fred[0][2] + fred[1][2];
    }
    if (p.grad && !SSG_DBG(p, 2)) {
      if (wv < 2) {
        const float sg = wave_sum(Gv);
        if ((tid & 63) == 0) fred[wv][3] = sg;
      }
      __syncthreads();
      const float sumG = fred[0][3] + fred[1][3];
      const float *ts = &tile[0][0][0];
      // tile positions: gS[c,t] = -2 sum_k Gz[t-k] (A[c,k] - S[c,t])
      for (int i = tid; i < C * P; i += 256) {
        const int c = i / P, r = i - c * P, ty = r / KS, tx = r - ty * KS;
        const float st = ts[i];
        float t = 0.f;
#pragma unroll
        for (int kh = -HK; kh <= HK; ++kh)
#pragma unroll
          for (int kx = -HK; kx <= HK; ++kx) {
            const int qy = ty - kh, qx = tx - kx;
            const bool in = (unsigned)qy < (unsigned)KS && (unsigned)qx < (unsigned)KS;
            const float gv = Gt[in ? qy * KS + qx : 0];
            t = __builtin_fmaf(in ? gv : 0.f, ts[c * P + (HP + kh) * KS + HP + kx] - st, t);
          }
        gst[c][r] = -2.f * t;
      }
      // window positions: gA[c,k] = 2 (A sum G - sum_p G[p] Sz[c,p+k]), the offsets' rows in NPART shares
      if (tid < NPART * C * K2) {
        const int part = tid / (C * K2), it = tid - part * (C * K2);
        const int c = it / K2, r = it - c * K2, kh = r / KW - HK, kx = r - (r / KW) * KW - HK;
        float t = 0.f;
        const int q0 = part * ROWS_PER_PART, q1 = q0 + ROWS_PER_PART < KS ? q0 + ROWS_PER_PART : KS;
        for (int qy = q0; qy < q1; ++qy) {
          const int yy = qy + kh;
          if ((unsigned)yy >= (unsigned)KS) continue;
#pragma unroll
          for (int qx = 0; qx < KS; ++qx) {
            const int xx = qx + kx;
            const bool in = (unsigned)xx < (unsigned)KS;
            t = __builtin_fmaf(in ? Gt[qy * KS + qx] : 0.f, ts[c * P + yy * KS + (in ? xx : 0)], t);
          }
        }
        wpart[part][it] = t;
      }
      __syncthreads();
      if (tid < C * K2) {
        const int c = tid / K2, r = tid - c * K2, kh = r / KW, kx = r - kh * KW;
        const int at = (HP - HK + kh) * KS + (HP - HK + kx);
        float t = wpart[0][tid];
#pragma unroll
        for (int k = 1; k < NPART; ++k) t += wpart[k][tid];
        gst[c][at] += 2.f * (ts[c * P + at] * sumG - t);   // (unique owner per (c, k))
      }
      __syncthreads();
      for (int i = tid; i < C * P; i += 256) {
        const int c = i / P, r = i - c * P, ty = r / KS, tx = r - ty * KS;
        const float v = gst[c][r];
        if (v != 0.f && !SSG_DBG(p, 4)) {
          const size_t gi = (((size_t)e.b * C + c) * H + reflect_idx(e.y - HP + ty, H)) * W + reflect_idx(e.x - HP + tx, W);
          grad_add(p.grad, fix ? p.gfix : nullptr, gi, v, gsc);
        }
      }
    }
    __syncthreads();   // (the tiles, the rows and the staging are the next row's)
  }

  // ---- the last workgroup through finishes the step ----
  // Only the workgroups that had a row take part (with no row at all workgroup 0 stands in): 1,024 tickets on one address
  // were 6 us of the launch.  The hand-off is the grid-synchronisation idiom: every lane's writes are ordered in front of
  // the workgroup barrier, then ONE lane draws the ticket with an agent-scope RELEASE -- on gfx950 an L2 write-back
  // (buffer_wbl2 sc1) in front of the atomic; issued by every wave (a __threadfence() per lane: 1,024 workgroups x 4) it made
  // the first version of this kernel 74 us long, 4 per working workgroup 37 us, one per working workgroup 26 us.  It cannot
  // be dropped: with a relaxed ticket behind an s_waitcnt the last workgroup folded sums that lacked other XCDs' last
  // additions, once in a few hundred steps of a 2 x 10 x 6 pixel call (tools/r6_fuzz_tiny.py).
  const int n_work = nrows < (int)gridDim.x ? (nrows > 0 ? nrows : 1) : (int)gridDim.x;
  if ((int)blockIdx.x >= n_work) return;
  // (every wave's own atomics and stores acknowledged before the barrier: the workgroup-scope fence of __syncthreads() leaves
  //  vmcnt alone on this target -- the waves of a workgroup share their L1 -- and the ticket below is drawn by ONE wave)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_last = __hip_atomic_fetch_add(p.ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n_work - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();   // (every wave of the last workgroup: its cached lines are invalidated before it reads the others' sums)
  {
    double a1 = 0.0, a2 = 0.0;
    // (plain loads behind the fence above -- it invalidates this CU's and this XCD's cached lines; a load with agent scope
    //  of its own would make 2 x nrows / 256 dependent round trips to memory of this loop)
    const float2 *pp = (const float2 *)p.partials;
    for (int i = tid; i < nrows; i += 256) {
      const float2 v = pp[i];
      a1 += (double)v.x;
      a2 += (double)v.y;
    }
    s1[tid] = a1;
    s2[tid] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        s1[tid] += s1[tid + o];
        s2[tid] += s2[tid + o];
      }
      __syncthreads();
    }
    if (tid == 0) {
      const double M = (double)nrows * (double)P;
      p.loss_out[0] = nrows > 0 ? (float)((double)p.w_l1 * s1[0] / M) : 0.f;
      p.loss_out[1] = nrows > 0 ? (float)((double)p.w_kl * s2[0] / M) : 0.f;
      // (a step that found more edge pixels than the caller's capacity has used the first `capacity` only: NaN losses)
      if (p.nan_on_overflow && p.n_dev && *p.n_dev > p.n_host) p.loss_out[0] = p.loss_out[1] = __builtin_nanf("");
    }
  }
  if (fix && !SSG_DBG(p, 8)) {
    const double inv = 1.0 / (double)gsc;
    // (one workgroup, after everybody else: as few dependent round trips to memory as the registers allow -- 16-byte loads, 12
    //  in flight per lane: C1's 12,288 sums are two rounds)
    if (((((size_t)p.gfix) & 15) | (((size_t)p.grad) & 7)) == 0) {
      constexpr int UN = 12;
      const size_t n2 = n_grad / 2;
      for (size_t i0 = 0; i0 < n2; i0 += 256 * UN) {
        longlong2 v[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
          const size_t i = i0 + tid + 256 * (size_t)k;
          v[k] = i < n2 ? ((const longlong2 *)p.gfix)[i] : longlong2{0, 0};
        }
        if (p.assign) {
#pragma unroll
          for (int k = 0; k < UN; ++k) {
            const size_t i = i0 + tid + 256 * (size_t)k;
            if (i < n2) ((float2 *)p.grad)[i] = make_float2((float)((double)v[k].x * inv), (float)((double)v[k].y * inv));
          }
        } else {
#pragma unroll
          for (int k = 0; k < UN; ++k) {
            const size_t i = i0 + tid + 256 * (size_t)k;
            if (i < n2 && (v[k].x | v[k].y)) {
              float2 o = ((float2 *)p.grad)[i];
              o.x += (float)((double)v[k].x * inv);
              o.y += (float)((double)v[k].y * inv);
              ((float2 *)p.grad)[i] = o;
            }
          }
        }
      }
      if ((n_grad & 1) && tid == 0) {
        const long long v = p.gfix[n_grad - 1];
        if (p.assign) p.grad[n_grad - 1] = (float)((double)v * inv);
        else if (v) p.grad[n_grad - 1] += (float)((double)v * inv);
      }
    } else {
      for (size_t i = tid; i < n_grad; i += 256) {
        const long long v = p.gfix[i];
        if (p.assign) p.grad[i] = (float)((double)v * inv);
        else if (v) p.grad[i] += (float)((double)v * inv);
      }
    }
    // (the bound word behind the sums, as the six-launch path leaves it)
    if (tid == 0) *(unsigned *)(p.gfix + n_grad) = __float_as_uint(loss_grad_bound(p.sigma, C, KW, p.w_l1, p.w_kl, nullptr, nrows, P));
  }
}

constexpr int TINY_MAX_ROWS = 4096, TINY_GRID = 1024;

bool tiny_step_supported(int ks, int kw, int C, int capacity) { return ks == 11 && kw == 5 && C == 3 && capacity > 0 && capacity <= TINY_MAX_ROWS; }

int launch_tiny_step(const TinyParams &p, int C, hipStream_t st) {
  (void)C;
  const unsigned grid = (unsigned)(p.n_host < TINY_GRID ? p.n_host : TINY_GRID);
  hipLaunchKernelGGL((ssg_tiny_step<11, 5, 3>), dim3(grid), dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

}  // namespace ssg
