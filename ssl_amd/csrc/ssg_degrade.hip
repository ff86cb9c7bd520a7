// The element-wise / gather stages of the Real-ESRGAN degradation chain between the blur, JPEG and USM kernels of
// ssg_datapath.hip (GAN-Based-SR/basicsr/models/realesrganssl_model.py:168-297):
//   resize_kernel<MODE>   F.interpolate(x, scale_factor= | size=, mode='area' | 'bilinear' | 'bicubic') as the model calls
//                         it (:185,203,224,255,280,293): align_corners=False, no antialias
//   gaussian_noise_kernel add_gaussian_noise_pt (basicsr/data/degradations.py:455-507) with the normal fields as inputs
//   poisson_*             add_poisson_noise_pt (:601-674): level census (torch.unique), rates for torch.poisson, and
//                         the arithmetic after the draw
//   clamp_round_kernel    torch.clamp((out * 255.0).round(), 0, 255) / 255.  (:206,297)
// The random DRAWS stay with torch's device generator (ssl_amd/datapath.py): a hand-written generator could not
// reproduce them, everything around them is deterministic and lives here.  All kernels are HBM-bound: one thread per
// output element, neighbouring lanes on neighbouring pixels; inputs are re-read through L1/L2 (a bicubic output
// touches 16 inputs, its neighbour 12 of the same).
//
// Floating point: every stage evaluates the reference's expression in its order of operations with individually
// rounded fp32 multiplies / adds (this file is compiled with -ffp-contract=off -- csrc/Makefile -- on top of the
// __fmul_rn / __fadd_rn spelling: hipcc otherwise fuses a multiply into the following add, which the reference's
// separate torch kernels never do; measured: 1-ulp differences in the gray Poisson path), so the noise and rounding stages are bit-identical to a CPU fp32 run of the reference; the
// resizes agree to fp32 rounding of the source coordinates (torch's own fp32 result is 3e-6 from its fp64 one).
#include "../../include/ssg_hip.h"

#include "ssg_common.hpp"

namespace ssg {

enum ResizeMode : int { RESIZE_AREA = 0, RESIZE_BILINEAR = 1, RESIZE_BICUBIC = 2 };

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

// cubic convolution coefficients, A = -0.75 (torch's get_cubic_upsample_coefficients)
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  auto cc1 = [&](float x) { return add(mul(mul(sub(mul(add(A, 2.f), x), add(A, 3.f)), x), x), 1.f); };
  auto cc2 = [&](float x) { return sub(mul(add(mul(sub(mul(A, x), mul(5.f, A)), x), mul(8.f, A)), x), mul(4.f, A)); };
  w[0] = cc2(add(t, 1.f));
  w[1] = cc1(t);
  w[2] = cc1(sub(1.f, t));
  w[3] = cc2(sub(2.f, t));
}

// x (planes, Hi, Wi) -> out (planes, Ho, Wo); scale_* = torch's area_pixel_compute_scale (host side)
template <int MODE>
__global__ __launch_bounds__(256) void resize_kernel(const float *x, float *out, int planes, int Hi, int Wi, int Ho,
                                                     int Wo, float scale_h, float scale_w) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= Wo || oy >= Ho) return;
  for (int p = blockIdx.z; p < planes; p += gridDim.z) {
    const float *src = x + (size_t)p * Hi * Wi;
    float v;
    if constexpr (MODE == RESIZE_AREA) {
      // adaptive_avg_pool2d: [floor(o in / out), ceil((o + 1) in / out)), running sum in row-major order, one division
      const int y0 = (int)floorf((float)(oy * Hi) / (float)Ho), y1 = (int)ceilf((float)((oy + 1) * Hi) / (float)Ho);
      const int x0 = (int)floorf((float)(ox * Wi) / (float)Wo), x1 = (int)ceilf((float)((ox + 1) * Wi) / (float)Wo);
      float acc = 0.f;
      for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) acc = add(acc, src[(size_t)yy * Wi + xx]);
      v = acc / (float)((y1 - y0) * (x1 - x0));
    } else if constexpr (MODE == RESIZE_BILINEAR) {
      float sy = sub(mul(scale_h, add((float)oy, 0.5f)), 0.5f), sx = sub(mul(scale_w, add((float)ox, 0.5f)), 0.5f);
      sy = sy < 0.f ? 0.f : sy;
      sx = sx < 0.f ? 0.f : sx;
      int y0 = (int)sy, x0 = (int)sx;
      y0 = y0 > Hi - 1 ? Hi - 1 : y0;
      x0 = x0 > Wi - 1 ? Wi - 1 : x0;
      const int y1 = y0 + (y0 < Hi - 1), x1 = x0 + (x0 < Wi - 1);
      float ty = sub(sy, (float)y0), tx = sub(sx, (float)x0);
      ty = fminf(fmaxf(ty, 0.f), 1.f);
      tx = fminf(fmaxf(tx, 0.f), 1.f);
      const float wy0 = sub(1.f, ty), wx0 = sub(1.f, tx);
      const float top = add(mul(src[(size_t)y0 * Wi + x0], wx0), mul(src[(size_t)y0 * Wi + x1], tx));
      const float bot = add(mul(src[(size_t)y1 * Wi + x0], wx0), mul(src[(size_t)y1 * Wi + x1], tx));
      v = add(mul(top, wy0), mul(bot, ty));
    } else {
      const float sy = sub(mul(scale_h, add((float)oy, 0.5f)), 0.5f), sx = sub(mul(scale_w, add((float)ox, 0.5f)), 0.5f);
      const float fy = floorf(sy), fx = floorf(sx);
      const int iy = (int)fy, ix = (int)fx;
      float wy[4], wx[4];
      cubic_coeffs(sub(sy, fy), wy);
      cubic_coeffs(sub(sx, fx), wx);
      v = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        int yy = iy - 1 + a;
        yy = yy < 0 ? 0 : (yy > Hi - 1 ? Hi - 1 : yy);
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          int xx = ix - 1 + b;
          xx = xx < 0 ? 0 : (xx > Wi - 1 ? Wi - 1 : xx);
          row = add(row, mul(src[(size_t)yy * Wi + xx], wx[b]));
        }
        v = add(v, mul(row, wy[a]));
      }
    }
    out[((size_t)p * Ho + oy) * Wo + ox] = v;
  }
}

__device__ __forceinline__ float clip_round(float v, int clip, int rounds) {
  if (clip && rounds) return fminf(fmaxf(rintf(mul(v, 255.f)), 0.f), 255.f) / 255.f;   // rintf = half to even = torch.round
  if (clip) return fminf(fmaxf(v, 0.f), 1.f);
  if (rounds) return rintf(mul(v, 255.f)) / 255.f;
  return v;
}

__global__ __launch_bounds__(256) void clamp_round_kernel(const float *x, float *out, size_t n, int clip, int rounds) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = clip_round(x[i], clip, rounds);
}

// out = clip_round(img + noise), noise = (fc * sigma / 255) (1 - g) + (fg * sigma / 255) g   (degradations.py:481-489);
// fg is ONE (H,W) field shared by the batch (the reference's broadcast), read only when some sample has g != 0
__global__ __launch_bounds__(256) void gaussian_noise_kernel(const float *img, float *out, const float *fc,
                                                             const float *fg, const float *sigma, const float *gray,
                                                             int C, int HW, size_t n, int any_gray, int clip, int rounds) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / ((size_t)C * HW)), px = (int)(i % HW);
    const float s = sigma[b];
    float noise = mul(fc[i], s) / 255.f;
    if (any_gray) {
      const float g = gray[b];
      const float ng = mul(fg[px], s) / 255.f;
      noise = add(mul(noise, sub(1.f, g)), mul(ng, g));
    }
    out[i] = clip_round(add(img[i], noise), clip, rounds);
  }
}

__device__ __forceinline__ int level_of(float v) { return (int)fminf(fmaxf(rintf(mul(v, 255.f)), 0.f), 255.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) {
  // torchvision rgb_to_grayscale: (0.2989 r + 0.587 g + 0.114 b), left to right
  return add(add(mul(0.2989f, r), mul(0.587f, g)), mul(0.114f, b));
}

// census of the distinct levels of every sample (torch.unique on the rounded image, degradations.py:626-628,635-637):
// bitmap[b][0..7] colour levels, bitmap[b][8..15] gray levels (zeroed by the caller)
__global__ __launch_bounds__(256) void poisson_census_kernel(const float *img, unsigned *bitmap, int C, int HW,
                                                             int want_gray) {
  __shared__ unsigned sb[16];
  const int b = blockIdx.y;
  if (threadIdx.x < 16) sb[threadIdx.x] = 0u;
  __syncthreads();
  const float *src = img + (size_t)b * C * HW;
  for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += gridDim.x * 256) {
    for (int c = 0; c < C; ++c) {
      const int k = level_of(src[(size_t)c * HW + px]);
      atomicOr(&sb[k >> 5], 1u << (k & 31));
    }
    if (want_gray && C >= 3) {
      const int k = level_of(gray_of(src[px], src[(size_t)HW + px], src[2 * (size_t)HW + px]));
      atomicOr(&sb[8 + (k >> 5)], 1u << (k & 31));
    }
  }
  __syncthreads();
  if (threadIdx.x < 16 && sb[threadIdx.x]) atomicOr(&bitmap[b * 16 + threadIdx.x], sb[threadIdx.x]);
}

// 2 ** ceil(log2(number of distinct levels))
__device__ __forceinline__ float vals_of(const unsigned *bm) {
  int n = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) n += __popc(bm[k]);
  int v = 1;
  while (v < n) v <<= 1;
  return (float)v;
}

// rates torch.poisson is drawn from: img_r * vals (colour, (B,C,H,W)) and gray_r * vals_gray ((B,1,H,W)); vals -> (B,2)
__global__ __launch_bounds__(256) void poisson_rates_kernel(const float *img, const unsigned *bitmap, float *rate,
                                                            float *rate_gray, float *vals, int C, int HW) {
  const int b = blockIdx.y;
  const float vc = vals_of(bitmap + b * 16), vg = rate_gray ? vals_of(bitmap + b * 16 + 8) : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    vals[2 * b] = vc;
    vals[2 * b + 1] = vg;
  }
  const float *src = img + (size_t)b * C * HW;
  for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += gridDim.x * 256) {
    for (int c = 0; c < C; ++c)
      rate[((size_t)b * C + c) * HW + px] = mul((float)level_of(src[(size_t)c * HW + px]) / 255.f, vc);
    if (rate_gray)
      rate_gray[(size_t)b * HW + px] =
          mul((float)level_of(gray_of(src[px], src[(size_t)HW + px], src[2 * (size_t)HW + px])) / 255.f, vg);
  }
}

// out = clip_round(img + ((dc / vals - img_r) (1 - g) + (dg / vals_g - gray_r) g) * scale)   (degradations.py:631-645)
__global__ __launch_bounds__(256) void poisson_noise_kernel(const float *img, float *out, const float *dc, const float *dg,
                                                            const float *vals, const float *scale, const float *gray,
                                                            int C, int HW, int clip, int rounds) {
  const int b = blockIdx.y;
  const float vc = vals[2 * b], vg = vals[2 * b + 1], sc = scale[b], g = dg ? gray[b] : 0.f;
  const float *src = img + (size_t)b * C * HW;
  for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += gridDim.x * 256) {
    float ng = 0.f;
    if (dg) {
      const float gr = (float)level_of(gray_of(src[px], src[(size_t)HW + px], src[2 * (size_t)HW + px])) / 255.f;
      ng = sub(dg[(size_t)b * HW + px] / vg, gr);
    }
    for (int c = 0; c < C; ++c) {
      const size_t i = ((size_t)b * C + c) * HW + px;
      const float ir = (float)level_of(src[(size_t)c * HW + px]) / 255.f;
      float noise = sub(dc[i] / vc, ir);
      if (dg) noise = add(mul(noise, sub(1.f, g)), mul(ng, g));
      out[i] = clip_round(add(src[(size_t)c * HW + px], mul(noise, sc)), clip, rounds);
    }
  }
}

static unsigned blocks_for(size_t n, unsigned cap = 8192) {
  const size_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

int launch_resize(const float *x, float *out, int planes, int Hi, int Wi, int Ho, int Wo, int mode, double sf_h,
                  double sf_w, hipStream_t st) {
  // area_pixel_compute_scale: 1 / scale_factor (double) rounded to fp32 when a scale_factor was given, in / out else
  const float sh = sf_h > 0 ? (float)(1.0 / sf_h) : (float)Hi / (float)Ho;
  const float sw = sf_w > 0 ? (float)(1.0 / sf_w) : (float)Wi / (float)Wo;
  const dim3 grid((unsigned)((Wo + 63) / 64), (unsigned)((Ho + 3) / 4), (unsigned)(planes < 1024 ? planes : 1024));
  if (mode == RESIZE_AREA)
    hipLaunchKernelGGL(resize_kernel<RESIZE_AREA>, grid, dim3(256), 0, st, x, out, planes, Hi, Wi, Ho, Wo, sh, sw);
  else if (mode == RESIZE_BILINEAR)
    hipLaunchKernelGGL(resize_kernel<RESIZE_BILINEAR>, grid, dim3(256), 0, st, x, out, planes, Hi, Wi, Ho, Wo, sh, sw);
  else if (mode == RESIZE_BICUBIC)
    hipLaunchKernelGGL(resize_kernel<RESIZE_BICUBIC>, grid, dim3(256), 0, st, x, out, planes, Hi, Wi, Ho, Wo, sh, sw);
  else
    return -1;
  return (int)hipGetLastError();
}

int launch_clamp_round(const float *x, float *out, size_t n, int clip, int rounds, hipStream_t st) {
  if (!n) return 0;
  hipLaunchKernelGGL(clamp_round_kernel, dim3(blocks_for(n)), dim3(256), 0, st, x, out, n, clip, rounds);
  return (int)hipGetLastError();
}

int launch_gaussian_noise(const float *img, float *out, const float *fc, const float *fg, const float *sigma,
                          const float *gray, int B, int C, int H, int W, int clip, int rounds, hipStream_t st) {
  const size_t n = (size_t)B * C * H * W;
  if (!n) return 0;
  hipLaunchKernelGGL(gaussian_noise_kernel, dim3(blocks_for(n)), dim3(256), 0, st, img, out, fc, fg, sigma, gray, C,
                     H * W, n, fg != nullptr, clip, rounds);
  return (int)hipGetLastError();
}

int launch_poisson_rates(const float *img, float *rate, float *rate_gray, float *vals, void *scratch, int B, int C,
                         int H, int W, hipStream_t st) {
  if ((size_t)B * C * H * W == 0) return 0;
  unsigned *bitmap = (unsigned *)scratch;
  int rc = (int)hipMemsetAsync(bitmap, 0, sizeof(unsigned) * 16 * (size_t)B, st);
  if (rc) return rc;
  const dim3 grid(blocks_for((size_t)H * W, 64), (unsigned)B);
  hipLaunchKernelGGL(poisson_census_kernel, grid, dim3(256), 0, st, img, bitmap, C, H * W, rate_gray != nullptr);
  hipLaunchKernelGGL(poisson_rates_kernel, dim3(blocks_for((size_t)H * W, 256), (unsigned)B), dim3(256), 0, st, img,
                     bitmap, rate, rate_gray, vals, C, H * W);
  return (int)hipGetLastError();
}

int launch_poisson_noise(const float *img, float *out, const float *dc, const float *dg, const float *vals,
                         const float *scale, const float *gray, int B, int C, int H, int W, int clip, int rounds,
                         hipStream_t st) {
  if ((size_t)B * C * H * W == 0) return 0;
  hipLaunchKernelGGL(poisson_noise_kernel, dim3(blocks_for((size_t)H * W, 256), (unsigned)B), dim3(256), 0, st, img, out,
                     dc, dg, vals, scale, gray, C, H * W, clip, rounds);
  return (int)hipGetLastError();
}

}  // namespace ssg

using namespace ssg;

extern "C" {

int ssg_resize(const float *img, float *out, int B, int C, int Hi, int Wi, int Ho, int Wo, int mode,
               double scale_factor_h, double scale_factor_w, ssg_stream_t stream) {
  if (B < 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || mode < 0 || mode > 2) return SSG_E_BADARG;
  if (B == 0) return 0;
  if (!img || !out || img == out) return SSG_E_BADARG;
  return launch_resize(img, out, B * C, Hi, Wi, Ho, Wo, mode, scale_factor_h, scale_factor_w, (hipStream_t)stream);
}

int ssg_clamp_round(const float *img, float *out, size_t n, int clip, int rounds, ssg_stream_t stream) {
  if (n && (!img || !out)) return SSG_E_BADARG;
  return launch_clamp_round(img, out, n, clip, rounds, (hipStream_t)stream);
}

int ssg_gaussian_noise(const float *img, float *out, const float *field_color, const float *field_gray,
                       const float *sigma, const float *gray, int B, int C, int H, int W, int clip, int rounds,
                       ssg_stream_t stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0) return SSG_E_BADARG;
  if (B == 0) return 0;
  if (!img || !out || !field_color || !sigma || (field_gray && !gray)) return SSG_E_BADARG;
  return launch_gaussian_noise(img, out, field_color, field_gray, sigma, gray, B, C, H, W, clip, rounds,
                               (hipStream_t)stream);
}

size_t ssg_poisson_scratch_bytes(int B) { return sizeof(unsigned) * 16 * (size_t)(B > 0 ? B : 1); }

int ssg_poisson_rates(const float *img, float *rate_color, float *rate_gray, float *vals, void *scratch, int B, int C,
                      int H, int W, ssg_stream_t stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || (rate_gray && C != 3)) return SSG_E_BADARG;
  if (B == 0) return 0;
  if (!img || !rate_color || !vals || !scratch) return SSG_E_BADARG;
  return launch_poisson_rates(img, rate_color, rate_gray, vals, scratch, B, C, H, W, (hipStream_t)stream);
}

int ssg_poisson_noise(const float *img, float *out, const float *draw_color, const float *draw_gray, const float *vals,
                      const float *scale, const float *gray, int B, int C, int H, int W, int clip, int rounds,
                      ssg_stream_t stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || (draw_gray && C != 3)) return SSG_E_BADARG;
  if (B == 0) return 0;
  if (!img || !out || !draw_color || !vals || !scale || (draw_gray && !gray)) return SSG_E_BADARG;
  return launch_poisson_noise(img, out, draw_color, draw_gray, vals, scale, gray, B, C, H, W, clip, rounds,
                              (hipStream_t)stream);
}

}  // extern "C"
