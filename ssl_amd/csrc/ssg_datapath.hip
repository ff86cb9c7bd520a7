// Step BEFORE the loss on the GPU (minimal slice): the index work that feeds (gt, mask)
// pairs to the SSG loss -- pure byte moves, HBM-bound, bit exact.
//
//   ssg_augment_crop : the dataset's joint flip / rot90 of image and mask (basicsr/data/transforms.py:152-219
//                      `augment`: hflip, then vflip, then transpose) followed by the joint random crop
//                      (transforms.py:93-149 `paired_random_crop_img_mask`), as ONE gather: no intermediate
//                      flipped copy, one read + one write per output element.
//   ssg_pool_swap    : the training pair pool's dequeue-and-enqueue (realesrganssl_model.py:327-367) for one
//                      tensor: the b samples at `slots` leave the queue and the current batch takes their place,
//                      in place (the reference permutes the whole queue with queue[idx] first; the host keeps that
//                      permutation as a slot table instead, ssl_amd/datapath.py).
//   ssg_usm_sharp    : USMSharp.forward (basicsr/utils/img_process_util.py:63-83), the sharpening that turns every GT
//                      batch into gt_usm (realesrganssl_model.py:165,315): blur = G * img (reflect padding), residual
//                      = img - blur, mask = |residual| * 255 > threshold, soft = G * mask, out = soft * clip(img +
//                      weight * residual, 0, 1) + (1 - soft) * img, with G the (radius x radius) Gaussian.  The
//                      reference convolves with the 51 x 51 outer product; here the two blurs are separable row /
//                      column passes through LDS tiles (2 x 51 instead of 2601 taps per pixel), the elementwise steps
//                      ride in the column passes' epilogues: four launches, ~12 floats of HBM traffic per element.
#include "ssg_common.hpp"

namespace ssg {

// out[b,c,y,x] = src[b,c,sy,sx]; params (B,5) int32: top, left (in the augmented image), hflip, vflip, rot90
template <class T>
__global__ __launch_bounds__(256) void augment_crop(const T *src, T *dst, int B, int C, int Hs, int Ws, int Ho,
                                                    int Wo, const int *params) {
  const size_t n = (size_t)B * C * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % Wo);
    size_t t = i / Wo;
    const int y = (int)(t % Ho);
    t /= Ho;
    const int c = (int)(t % C), b = (int)(t / C);
    const int *p = params + 5 * b;
    int ya = p[0] + y, xa = p[1] + x;        // augmented-image coordinates
    if (p[4]) {                              // undo the transpose
      const int s = ya;
      ya = xa;
      xa = s;
    }
    if (p[3]) ya = Hs - 1 - ya;              // undo the vertical flip
    if (p[2]) xa = Ws - 1 - xa;              // undo the horizontal flip
    dst[i] = src[(((size_t)b * C + c) * Hs + ya) * Ws + xa];
  }
}

// queue sample slots[k] <-> batch sample k, 16 bytes per lane where the sample size allows
template <class V>
__global__ __launch_bounds__(256) void pool_swap(V *queue, V *batch, size_t sample_v, const int *slots, int b) {
  const size_t n = sample_v * b;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i / sample_v);
    const size_t o = i - (size_t)k * sample_v;
    V *q = queue + (size_t)slots[k] * sample_v + o;
    const V a = *q, c = batch[i];
    *q = c;
    batch[i] = a;
  }
}

int launch_augment_crop(const void *src, void *dst, int elem_bytes, int B, int C, int Hs, int Ws, int Ho, int Wo,
                        const int *params, hipStream_t st) {
  const size_t n = (size_t)B * C * Ho * Wo;
  if (n == 0) return 0;
  const unsigned grid = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(augment_crop<uint32_t>, dim3(grid), dim3(256), 0, st, (const uint32_t *)src, (uint32_t *)dst, B,
                       C, Hs, Ws, Ho, Wo, params);
  else if (elem_bytes == 1)
    hipLaunchKernelGGL(augment_crop<uint8_t>, dim3(grid), dim3(256), 0, st, (const uint8_t *)src, (uint8_t *)dst, B, C,
                       Hs, Ws, Ho, Wo, params);
  else
    return -1;
  return (int)hipGetLastError();
}

int launch_pool_swap(void *queue, void *batch, size_t sample_bytes, const int *slots, int b, hipStream_t st) {
  if (sample_bytes == 0 || b == 0) return 0;
  const bool wide = sample_bytes % 16 == 0 && ((size_t)queue % 16) == 0 && ((size_t)batch % 16) == 0;
  const size_t sv = wide ? sample_bytes / 16 : sample_bytes, n = sv * b;
  const unsigned grid = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  if (wide)
    hipLaunchKernelGGL(pool_swap<uint4>, dim3(grid), dim3(256), 0, st, (uint4 *)queue, (uint4 *)batch, sv, slots, b);
  else
    hipLaunchKernelGGL(pool_swap<uint8_t>, dim3(grid), dim3(256), 0, st, (uint8_t *)queue, (uint8_t *)batch, sv, slots,
                       b);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- USM ----
constexpr int USM_MAX_TAPS = 63, USM_PAD = 15;
struct UsmTaps {
  float k[USM_PAD + USM_MAX_TAPS + USM_PAD + 3];   // taps at [USM_PAD, USM_PAD + n), zeros around them
  int n;
};

// One separable pass of the Gaussian over (planes, H, W) fp32, PyTorch 'reflect' padding (F.pad(..., 'reflect'),
// img_process_util.py:17), 256 threads per tile, the input tile with its halo along the pass direction in LDS:
//   column pass: 64 x 64 outputs (114 input rows for 51 taps: 1.8 x re-read), a thread owns 16 consecutive rows of one
//                column -- 64 lanes on 64 consecutive columns, one LDS dword per input row, 16 FMAs each;
//   row pass   : 16 rows x 128 columns, a thread owns 8 consecutive columns -- one 16-byte LDS read per 4 inputs.
// Every input is multiplied into all the outputs it reaches against the zero-padded tap array, so the loops carry no
// conditions.  NTAPS > 0: tap count at compile time (51 = the reference's USMSharp()): the loops unroll and the taps sit
// in SGPRs; 0: any odd count <= 63, taps fetched from the kernel arguments step by step.
//   EPI 0: dst = blur
//   EPI 1: (column pass of the image blur) res = img - blur -> dst, mask = |res| * 255 > threshold -> dst2 (0/1 floats)
//   EPI 2: (column pass of the mask blur) soft = blur; out = soft * clip(img + weight * res, 0, 1) + (1 - soft) * img
constexpr int USM_VTX = 64, USM_VTY = 64, USM_HTX = 128, USM_HTY = 16;
template <bool VERT, int EPI, int NTAPS>
__global__ __launch_bounds__(256) void usm_pass(const float *src, float *dst, float *dst2, const float *img,
                                                const float *res, int planes, int H, int W, UsmTaps taps, float weight,
                                                float threshold) {
  constexpr int TX = VERT ? USM_VTX : USM_HTX, TY = VERT ? USM_VTY : USM_HTY, RPT = VERT ? 16 : 8;
  static_assert(RPT - 1 <= USM_PAD, "tap array padding");
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int n = NTAPS > 0 ? NTAPS : taps.n, R = n / 2;
  const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TY;
  const size_t plane = (size_t)blockIdx.z * H * W;
  const float *sp = src + plane;
  const int tw = VERT ? TX : TX + 2 * R, th = VERT ? TY + 2 * R : TY;
  const int ts = VERT ? TX + 1 : ((tw + RPT + 3) & ~3);   // (row pass: 16-byte rows, zeros behind the window)
  // (all global loads of a batch are issued before the first LDS store: one load per loop trip pays the L2 latency
  // 28 times per thread -- the fill, not the 51 taps, was what the first version spent its 40 us per pass on)
  if (VERT) {
    const int lx = threadIdx.x & 63;
    int gx = tx0 + lx;
    gx = gx >= W ? W - 1 : gx;                 // (columns of a partial tile past the image: never written)
    constexpr int U = 8;
    for (int base = threadIdx.x >> 6; base < th; base += 4 * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int gy = reflect_idx(ty0 + base + 4 * u - R, H);
        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        v[u] = sp[(size_t)gy * W + gx];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (base + 4 * u < th) tile[(base + 4 * u) * ts + lx] = v[u];
    }
  } else {
    const int lx = threadIdx.x;                // tw <= 128 + 62 < 256
    int gx = reflect_idx(tx0 + lx - R, W);
    gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    float v[TY];
#pragma unroll
    for (int ly = 0; ly < TY; ++ly) {
      int gy = ty0 + ly;
      gy = gy >= H ? H - 1 : gy;
      v[ly] = sp[(size_t)gy * W + gx];
    }
#pragma unroll
    for (int ly = 0; ly < TY; ++ly)
      if (lx < ts) tile[ly * ts + lx] = lx < tw ? v[ly] : 0.f;
  }
  __syncthreads();
  float acc[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) acc[j] = 0.f;
  int x, y;   // first of the thread's RPT outputs
  if (VERT) {
    const int lx = threadIdx.x & 63, ly = RPT * (threadIdx.x >> 6);
    x = tx0 + lx;
    y = ty0 + ly;
    const float *col = tile + ly * ts + lx;
#pragma unroll
    for (int t = 0; t < n + RPT - 1; ++t) {     // input row ly + t feeds output j with tap t - j
      const float v = col[t * ts];
      const float *kp = taps.k + USM_PAD + t;
#pragma unroll
      for (int j = 0; j < RPT; ++j) acc[j] = __builtin_fmaf(kp[-j], v, acc[j]);
    }
  } else {
    const int lx = RPT * (threadIdx.x & 15), ly = threadIdx.x >> 4;
    x = tx0 + lx;
    y = ty0 + ly;
    const float4 *row = (const float4 *)(tile + ly * ts + lx);
#pragma unroll
    for (int t4 = 0; t4 < (n + RPT - 1 + 3) / 4; ++t4) {
      const float4 q = row[t4];
      const float v[4] = {q.x, q.y, q.z, q.w};
      const float *kp = taps.k + USM_PAD + 4 * t4;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < RPT; ++j) acc[j] = __builtin_fmaf(kp[u - j], v[u], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int yy = VERT ? y + j : y, xx = VERT ? x : x + j;
    if (yy >= H || xx >= W) continue;
    const size_t o = plane + (size_t)yy * W + xx;
    if (EPI == 0) {
      dst[o] = acc[j];
    } else if (EPI == 1) {
      const float r = img[o] - acc[j];
      dst[o] = r;
      dst2[o] = fabsf(r) * 255.f > threshold ? 1.f : 0.f;
    } else {
      const float v = img[o];
      float sharp = v + weight * res[o];
      sharp = fminf(fmaxf(sharp, 0.f), 1.f);
      dst[o] = acc[j] * sharp + (1.f - acc[j]) * v;
    }
  }
}

// cv2.getGaussianKernel(ksize, sigma) as documented (OpenCV imgproc, getGaussianKernel): fixed tables for ksize <= 7
// with sigma <= 0, else G_i = a exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)) normalised to sum 1, sigma <= 0 meaning
// 0.3 ((ksize-1) 0.5 - 1) + 0.8; computed in fp64 and rounded once (the reference rounds the fp64 outer product).
int usm_gaussian_taps(int ksize, double sigma, UsmTaps *out) {
  if (ksize < 1 || ksize > USM_MAX_TAPS || !(ksize & 1)) return -1;
  static const double tab1[] = {1.0}, tab3[] = {0.25, 0.5, 0.25}, tab5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625},
                      tab7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
  const double *fixed = nullptr;
  if (sigma <= 0 && ksize <= 7) fixed = ksize == 1 ? tab1 : ksize == 3 ? tab3 : ksize == 5 ? tab5 : tab7;
  double k[USM_MAX_TAPS], sum = 0.0;
  const double sg = sigma > 0 ? sigma : ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8, s2 = -0.5 / (sg * sg);
  for (int i = 0; i < ksize; ++i) {
    const double x = i - (ksize - 1) * 0.5;
    k[i] = fixed ? fixed[i] : exp(s2 * x * x);
    sum += k[i];
  }
  out->n = ksize;
  for (float &v : out->k) v = 0.f;
  for (int i = 0; i < ksize; ++i) out->k[USM_PAD + i] = (float)(fixed ? k[i] : k[i] / sum);
  return 0;
}

size_t usm_scratch_bytes(int B, int C, int H, int W) { return 3 * sizeof(float) * (size_t)B * C * H * W; }

int launch_usm_sharp(const float *img, float *out, int B, int C, int H, int W, int ksize, float sigma, float weight,
                     float threshold, void *scratch, hipStream_t st) {
  UsmTaps taps;
  if (usm_gaussian_taps(ksize, (double)sigma, &taps)) return -1;
  const int R = ksize / 2;
  if (H <= R || W <= R) return -4;   // reflect padding needs pad < size (PyTorch raises the same way)
  const size_t n = (size_t)B * C * H * W;
  if (n == 0) return 0;
  float *t1 = (float *)scratch, *res = t1 + n, *msk = res + n;
  const dim3 grid_h((unsigned)((W + USM_HTX - 1) / USM_HTX), (unsigned)((H + USM_HTY - 1) / USM_HTY), (unsigned)(B * C));
  const dim3 grid_v((unsigned)((W + USM_VTX - 1) / USM_VTX), (unsigned)((H + USM_VTY - 1) / USM_VTY), (unsigned)(B * C));
  const size_t lds_h = sizeof(float) * USM_HTY * ((USM_HTX + 2 * R + 8 + 3) & ~3);
  const size_t lds_v = sizeof(float) * (USM_VTY + 2 * R) * (USM_VTX + 1);
  auto run = [&](auto nt) {
    constexpr int NT = decltype(nt)::value;
    hipLaunchKernelGGL((usm_pass<false, 0, NT>), grid_h, dim3(256), lds_h, st, img, t1, nullptr, nullptr, nullptr, B * C,
                       H, W, taps, weight, threshold);
    hipLaunchKernelGGL((usm_pass<true, 1, NT>), grid_v, dim3(256), lds_v, st, t1, res, msk, img, nullptr, B * C, H, W,
                       taps, weight, threshold);
    hipLaunchKernelGGL((usm_pass<false, 0, NT>), grid_h, dim3(256), lds_h, st, msk, t1, nullptr, nullptr, nullptr, B * C,
                       H, W, taps, weight, threshold);
    hipLaunchKernelGGL((usm_pass<true, 2, NT>), grid_v, dim3(256), lds_v, st, t1, out, nullptr, img, res, B * C, H, W,
                       taps, weight, threshold);
  };
  if (ksize == 51) run(std::integral_constant<int, 51>{});
  else run(std::integral_constant<int, 0>{});
  return (int)hipGetLastError();
}

// ------------------------------------------------------------ filter2D ----
// img_process_util.py:7-31 `filter2D`: reflect-pad by k/2, correlate every (b, c) plane with kernel b (or the one shared
// kernel) -- the blur steps of the degradation chain (realesrganssl_model.py:173,212,245,281,294).  Tile of 16 x 64
// outputs per workgroup with its halo in LDS, the k x k taps (zero-padded to K x 24) beside it; a thread owns 4
// consecutive outputs of a row and per tap row reads its K + 3 inputs and the K taps as 16-byte LDS reads: 4 K FMAs
// per (K + 3) / 2 reads.  K = 9 or 21: a k x k kernel is centred in the next of the two (the extra taps are zero, so
// the wider reflect halo they see does not matter).
template <int K>
__global__ __launch_bounds__(256) void filter2d_kernel(const float *img, const float *kernels, float *out, int C, int H,
                                                       int W, int k, int nk) {
  constexpr int TX = 64, TY = 16, R = K / 2, TW = TX + 2 * R, TS = (TW + 3) & ~3, KP = (K + 3 + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float tile[(TY + 2 * R) * TS];
  __shared__ __attribute__((aligned(16))) float taps[K * KP];
  const int plane_i = blockIdx.z, b = plane_i / C;
  const size_t plane = (size_t)plane_i * H * W;
  const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TY;
  const float *kp = kernels + (size_t)(nk == 1 ? 0 : b) * k * k;
  const int off = (K - k) / 2;
  for (int i = threadIdx.x; i < K * KP; i += 256) {
    const int ky = i / KP - off, kx = i % KP - off;
    taps[i] = (ky >= 0 && ky < k && kx >= 0 && kx < k) ? kp[ky * k + kx] : 0.f;
  }
  for (int i = threadIdx.x; i < (TY + 2 * R) * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    int gy = reflect_idx(ty0 + ly - R, H), gx = reflect_idx(tx0 + lx - R, W);
    gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);   // (beyond one reflection: only under zero taps or unwritten outputs)
    gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    tile[i] = img[plane + (size_t)gy * W + gx];
  }
  __syncthreads();
  const int lx = 4 * (threadIdx.x & 15), ly = threadIdx.x >> 4;
  // packed fp32: outputs (0,1) and (2,3) are two register pairs; tap kx multiplies the input pair (kx, kx+1) [+2] --
  // even kx: the pairs as read, odd kx: the pairs shifted by one input, built once per tap row
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll 3
  for (int ky = 0; ky < K; ++ky) {
    f2v ve[KP / 2], vo[KP / 2];
    float t[KP];
    const float4 *rv = (const float4 *)(tile + (ly + ky) * TS + lx), *rt = (const float4 *)(taps + ky * KP);
#pragma unroll
    for (int q = 0; q < KP / 4; ++q) {
      const float4 a = rv[q], c = rt[q];
      ve[2 * q] = f2v{a.x, a.y}, ve[2 * q + 1] = f2v{a.z, a.w};
      t[4 * q] = c.x, t[4 * q + 1] = c.y, t[4 * q + 2] = c.z, t[4 * q + 3] = c.w;
    }
#pragma unroll
    for (int m = 0; m + 1 < KP / 2; ++m) vo[m] = f2v{ve[m].y, ve[m + 1].x};
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const f2v tt = {t[kx], t[kx]};
      if (kx % 2 == 0) {
        acc01 = __builtin_elementwise_fma(tt, ve[kx / 2], acc01);
        acc23 = __builtin_elementwise_fma(tt, ve[kx / 2 + 1], acc23);
      } else {
        acc01 = __builtin_elementwise_fma(tt, vo[kx / 2], acc01);
        acc23 = __builtin_elementwise_fma(tt, vo[kx / 2 + 1], acc23);
      }
    }
  }
  const float acc[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
  const int y = ty0 + ly, x = tx0 + lx;
  if (y < H)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x + j < W) out[plane + (size_t)y * W + x + j] = acc[j];
}

int launch_filter2d(const float *img, const float *kernels, float *out, int B, int C, int H, int W, int k, int nk,
                    hipStream_t st) {
  if (k < 1 || k > 21 || !(k & 1) || (nk != 1 && nk != B)) return -1;
  if (H <= k / 2 || W <= k / 2) return -4;
  if ((size_t)B * C * H * W == 0) return 0;
  const dim3 grid((unsigned)((W + 63) / 64), (unsigned)((H + 15) / 16), (unsigned)(B * C));
  if (k <= 9) hipLaunchKernelGGL((filter2d_kernel<9>), grid, dim3(256), 0, st, img, kernels, out, C, H, W, k, nk);
  else hipLaunchKernelGGL((filter2d_kernel<21>), grid, dim3(256), 0, st, img, kernels, out, C, H, W, k, nk);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- JPEG ----
// DiffJPEG(differentiable=False).forward (basicsr/utils/diffjpeg.py:449-487), the JPEG simulation of the degradation
// chain (realesrganssl_model.py:34,201,240,285,290): zero-pad to multiples of 16, x 255, RGB -> YCbCr (:49-70), 2 x 2
// chroma average (:73-95), per 8 x 8 block DCT (:121-145), divide by table * factor and torch.round (:148-205),
// multiply back (:247-294), IDCT (:297-321), chroma repeat (:348-375), YCbCr -> RGB (:378-398), clamp, / 255, crop.
// One wave per 16 x 16 macroblock (4 luma blocks + Cb + Cr), everything between the load and the store on chip: lane
// (u, v) holds the 64 basis products cos((2x+1)u pi/16) cos((2y+1)v pi/16) of its coefficient (and the transposed
// set of its pixel for the inverse) as the reference builds them -- fp64 product rounded to fp32 -- and walks the
// block through LDS broadcasts.  factor = quality_to_factor(quality[b]) (:32-46): fp32 for the tensor branch, host double for a scalar.
__device__ const double kJpegCos[8][8] = {   // np.cos((2x+1) u pi / 16) [x][u], repr of the doubles numpy computes (diffjpeg.py:127)
    {1.0, 0.9807852804032304, 0.9238795325112867, 0.8314696123025452, 0.7071067811865476, 0.5555702330196023, 0.38268343236508984, 0.19509032201612833},
    {1.0, 0.8314696123025452, 0.38268343236508984, -0.1950903220161282, -0.7071067811865475, -0.9807852804032304, -0.9238795325112868, -0.5555702330196022},
    {1.0, 0.5555702330196023, -0.3826834323650897, -0.9807852804032304, -0.7071067811865477, 0.1950903220161283, 0.9238795325112865, 0.8314696123025455},
    {1.0, 0.19509032201612833, -0.9238795325112867, -0.5555702330196022, 0.7071067811865474, 0.8314696123025455, -0.3826834323650899, -0.9807852804032307},
    {1.0, -0.1950903220161282, -0.9238795325112868, 0.5555702330196018, 0.7071067811865477, -0.8314696123025451, -0.38268343236509056, 0.9807852804032304},
    {1.0, -0.555570233019602, -0.38268343236509034, 0.9807852804032304, -0.7071067811865467, -0.19509032201612803, 0.9238795325112867, -0.831469612302545},
    {1.0, -0.8314696123025453, 0.38268343236509, 0.19509032201612878, -0.7071067811865471, 0.9807852804032307, -0.9238795325112864, 0.5555702330196015},
    {1.0, -0.9807852804032304, 0.9238795325112865, -0.8314696123025451, 0.7071067811865466, -0.5555702330196015, 0.38268343236508956, -0.19509032201612858}};
__device__ const float kJpegY[8][8] = {{16, 11, 10, 16, 24, 40, 51, 61},     {12, 12, 14, 19, 26, 58, 60, 55},
                                       {14, 13, 16, 24, 40, 57, 69, 56},     {14, 17, 22, 29, 51, 87, 80, 62},
                                       {18, 22, 37, 56, 68, 109, 103, 77},   {24, 35, 55, 64, 81, 104, 113, 92},
                                       {49, 64, 78, 87, 103, 121, 120, 101}, {72, 92, 95, 98, 112, 100, 103, 99}};
__device__ const float kJpegC[4][4] = {{17, 18, 24, 47}, {18, 21, 26, 66}, {24, 26, 56, 99}, {47, 66, 99, 99}};

__global__ __launch_bounds__(64) void jpeg_kernel(const float *img, float *out, int H, int W, const float *quality,
                                                  float factor_host) {
  __shared__ float blk[6][64];   // 0..3 luma blocks (row-major 2 x 2), 4 Cb, 5 Cr -- pixels, then coefficients, then pixels
  const int lane = threadIdx.x, u = lane >> 3, v = lane & 7;
  const int b = blockIdx.z, y0 = 16 * blockIdx.y, x0 = 16 * blockIdx.x;
  const size_t plane = (size_t)H * W;
  const float *src = img + (size_t)b * 3 * plane;
  // tensor branch (diffjpeg.py:475-476): quality_to_factor on fp32 tensor elements, two fp32 roundings.  Scalar
  // branch (:473-474): Python doubles, rounded to fp32 ONCE where the table is multiplied (:169,199) -- that factor
  // is formed on the host in double (launch_jpeg) and arrives as `factor_host`.
  float factor = factor_host;
  if (quality) {
    float q = quality[b];
    q = q < 50.f ? 5000.f / q : 200.f - q * 2.f;
    factor = q / 100.f;
  }
  // ---- load 2 x 2 pixels per lane (zero beyond the image: F.pad constant 0), x 255, RGB -> YCbCr, chroma average ----
  {
    float cbs = 0.f, crs = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int py = 2 * u + (d >> 1), px = 2 * v + (d & 1), gy = y0 + py, gx = x0 + px;
      float r = 0.f, g = 0.f, bl = 0.f;
      if (gy < H && gx < W) {
        const size_t o = (size_t)gy * W + gx;
        r = src[o] * 255.f, g = src[plane + o] * 255.f, bl = src[2 * plane + o] * 255.f;
      }
      const float yy = (r * 0.299f + g * 0.587f) + bl * 0.114f;   // tensordot over the 3 channels + shift (0,128,128)
      const float cb = ((r * -0.168736f + g * -0.331264f) + bl * 0.5f) + 128.f;
      const float cr = ((r * 0.5f + g * -0.418688f) + bl * -0.081312f) + 128.f;
      blk[2 * (py >> 3) + (px >> 3)][8 * (py & 7) + (px & 7)] = yy;
      cbs += cb;
      crs += cr;
    }
    blk[4][lane] = cbs * 0.25f;   // avg_pool2d 2 x 2 (exact division by 4)
    blk[5][lane] = crs * 0.25f;
  }
  // basis products of this lane: forward T[x][y] = C[x][u] C[y][v]; inverse T2[i][j] = C[u][i] C[v][j] (lane = pixel)
  float tf[64], ti[64];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      tf[8 * i + j] = (float)(kJpegCos[i][u] * kJpegCos[j][v]);
      ti[8 * i + j] = (float)(kJpegCos[u][i] * kJpegCos[v][j]);
    }
  const float au = u == 0 ? 0.70710678118654746f : 1.f, av = v == 0 ? 0.70710678118654746f : 1.f;
  const float scale = (float)((u == 0 ? 0.70710678118654746 : 1.0) * (v == 0 ? 0.70710678118654746 : 1.0) * 0.25);
  const float alpha = (float)((u == 0 ? 0.70710678118654746 : 1.0) * (v == 0 ? 0.70710678118654746 : 1.0));
  (void)au; (void)av;
  __syncthreads();
  float coef[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 64; ++e) acc = __builtin_fmaf(blk[k][e] - 128.f, tf[e], acc);
    const float tab = (k < 4 ? kJpegY[v][u] : (u < 4 && v < 4 ? kJpegC[v][u] : 99.f)) * factor;   // tables stored transposed (:19,23)
    const float qv = rintf((scale * acc) / tab);                                                  // torch.round: half to even
    coef[k] = (qv * tab) * alpha;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 6; ++k) blk[k][lane] = coef[k];
  __syncthreads();
  float pix[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 64; ++e) acc = __builtin_fmaf(blk[k][e], ti[e], acc);
    pix[k] = 0.25f * acc + 128.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 6; ++k) blk[k][lane] = pix[k];
  __syncthreads();
  // ---- chroma repeat, YCbCr -> RGB, clamp, / 255: lane writes its 2 x 2 pixels ----
  float *dst = out + (size_t)b * 3 * plane;
  const float cb = blk[4][lane] - 128.f, cr = blk[5][lane] - 128.f;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int py = 2 * u + (d >> 1), px = 2 * v + (d & 1), gy = y0 + py, gx = x0 + px;
    if (gy >= H || gx >= W) continue;
    const float yy = blk[2 * (py >> 3) + (px >> 3)][8 * (py & 7) + (px & 7)];
    float r = (yy * 1.f + cb * 0.f) + cr * 1.402f;
    float g = (yy * 1.f + cb * -0.344136f) + cr * -0.714136f;
    float bl = (yy * 1.f + cb * 1.772f) + cr * 0.f;
    r = fminf(255.f, fmaxf(0.f, r));
    g = fminf(255.f, fmaxf(0.f, g));
    bl = fminf(255.f, fmaxf(0.f, bl));
    const size_t o = (size_t)gy * W + gx;
    dst[o] = r / 255.f;
    dst[plane + o] = g / 255.f;
    dst[2 * plane + o] = bl / 255.f;
  }
}

int launch_jpeg(const float *img, float *out, int B, int H, int W, const float *quality_dev, float quality_host,
                hipStream_t st) {
  if ((size_t)B * H * W == 0) return 0;
  const dim3 grid((unsigned)((W + 15) / 16), (unsigned)((H + 15) / 16), (unsigned)B);
  // scalar quality: quality_to_factor (diffjpeg.py:32-45) in double like the reference's Python floats, one rounding
  double q = (double)quality_host;
  q = q < 50.0 ? 5000.0 / q : 200.0 - q * 2.0;
  const float factor_host = (float)(q / 100.0);
  hipLaunchKernelGGL(jpeg_kernel, grid, dim3(64), 0, st, img, out, H, W, quality_dev, factor_host);
  return (int)hipGetLastError();
}

}  // namespace ssg
