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
#include "ssg_common.hpp"

namespace ssg {

template <int KS, int KW>
__global__ __launch_bounds__(256) void ssg_grad_rows(GrowParams p) {
  constexpr int P = KS * KS, HP = KS / 2, HK = KW / 2, EPL = (P + 63) / 64;
  __shared__ float sred[12];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nrows = rows_to_do(p.n_dev, p.n_host);
  const int n = blockIdx.x * 4 + wv;
  float l1p = 0.f, klp = 0.f, gmax = 0.f;
  if (n < nrows) {
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
    if (threadIdx.x == 0) p.gmax_part[blockIdx.x] = fmaxf(fmaxf(sred[8], sred[9]), fmaxf(sred[10], sred[11]));
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
      p.partials[2 * blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
      p.partials[2 * blockIdx.x + 1] = (sred[4] + sred[5]) + (sred[6] + sred[7]);
    }
  }
}

bool grow_supported(int ks, int kw) { return (ks == 25 && kw == 9) || (ks == 49 && kw == 13); }

unsigned grow_grid(int n_host) { return (unsigned)((n_host + 3) / 4); }

int launch_grad_rows(const GrowParams &p, int ks, int kw, hipStream_t st) {
  if (p.n_host <= 0) return 0;
  const unsigned grid = grow_grid(p.n_host);
  if (ks == 25 && kw == 9)
    hipLaunchKernelGGL((ssg_grad_rows<25, 9>), dim3(grid), dim3(256), 0, st, p);
  else if (ks == 49 && kw == 13)
    hipLaunchKernelGGL((ssg_grad_rows<49, 13>), dim3(grid), dim3(256), 0, st, p);
  else
    return -1;
  return (int)hipGetLastError();
}

}  // namespace ssg
