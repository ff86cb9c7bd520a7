/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See ssg_oracle_impl.h.
 *
 * Builds libssg_oracle.so with every function in two precisions
 * (suffix _f32 / _f64) plus the integer edge-mask restatement below.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define REAL float
#define SFX(x) x##_f32
#include "ssg_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL double
#define SFX(x) x##_f64
#include "ssg_oracle_impl.h"
#undef REAL
#undef SFX

/* cv2 BORDER_REFLECT_101 (the default border of cv2.Laplacian): -1 -> 1,
 * n -> n-2; n == 1 degenerates to 0. */
static inline int orc_reflect101(int i, int n)
{
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

/*
 * Offline edge mask of scripts/data_preparation/generate_mask.py:22-31:
 *   L    = PIL Image.convert("L")  (ITU-R 601-2 in 16.16 fixed point:
 *          (R*19595 + G*38470 + B*7471 + 0x8000) >> 16)
 *   lap  = cv2.Laplacian(L, cv2.CV_8U)   (ksize=1 -> [[0,1,0],[1,-4,1],[0,1,0]],
 *          BORDER_REFLECT_101, saturate_cast<uchar>)
 *   mask = lap > threshold (20.0)
 * rgb is HWC uint8 (as PIL/np.array give it).  OpenCV is not installed in the
 * build container, so the Laplacian step follows OpenCV's documented
 * semantics; the 'L' step is pinned against PIL in tests.
 */
void orc_edge_mask_rgb8(const uint8_t *rgb, int H, int W, float threshold,
                        uint8_t *mask, uint8_t *gray_out)
{
    uint8_t *L = (uint8_t *)malloc((size_t)H * W);
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        const uint32_t r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        L[i] = (uint8_t)((r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16);
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v = (int)L[(size_t)orc_reflect101(y - 1, H) * W + x] +
                    (int)L[(size_t)orc_reflect101(y + 1, H) * W + x] +
                    (int)L[(size_t)y * W + orc_reflect101(x - 1, W)] +
                    (int)L[(size_t)y * W + orc_reflect101(x + 1, W)] -
                    4 * (int)L[(size_t)y * W + x];
            if (v < 0) v = 0;
            if (v > 255) v = 255;
            mask[(size_t)y * W + x] = ((float)v > threshold) ? 1 : 0;
        }
    if (gray_out)
        for (size_t i = 0; i < (size_t)H * W; ++i) gray_out[i] = L[i];
    free(L);
}

/*
 * Same mask from a float CHW image in [0,1] as the training loop holds it
 * (img = uint8/255 in the dataset, my_realesrgan_image_mask_dataset.py:79-86):
 * round(255*x) recovers the 8-bit sample exactly for 8-bit sources.
 */
void orc_edge_mask_chw_f32(const float *img, int H, int W, float threshold,
                           uint8_t *mask)
{
    uint8_t *rgb = (uint8_t *)malloc((size_t)H * W * 3);
    for (int c = 0; c < 3; ++c)
        for (size_t i = 0; i < (size_t)H * W; ++i) {
            float v = img[(size_t)c * H * W + i] * 255.0f;
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
            rgb[3 * i + c] = (uint8_t)lrintf(v);
        }
    orc_edge_mask_rgb8(rgb, H, W, threshold, mask, NULL);
    free(rgb);
}

/* mask_stride eye pattern, realesrganssl_model.py:64-70: keep (y,x) iff
 * y % s == x % s. */
void orc_mask_stride(uint8_t *mask, int H, int W, int s)
{
    if (s <= 1) return;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if ((y % s) != (x % s)) mask[(size_t)y * W + x] = 0;
}
