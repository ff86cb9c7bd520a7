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

}  // namespace ssg
