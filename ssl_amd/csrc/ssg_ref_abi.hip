// The reference operator's own interface (include/similarity.h <- similarity.h:2-23): same names, same
// parameter lists, C++ linkage, legacy default stream, no status -- thin forwards to the C ABI.
#include "../../include/similarity.h"

#include "../../include/ssg_hip.h"

static thread_local int g_last_status = 0;

void _compute_similarity(const float *image, const int *pos, float *out, const int mc, const int psize,
                         const int ksize, const int height, const int width, const int channel) {
  g_last_status = ssg_compute_similarity(image, pos, out, mc, psize, ksize, height, width, channel, nullptr);
}

void _compute_similarity_backward(const float *image, const float *grads, const int *pos, float *image_grads,
                                  const int mc, const int psize, const int ksize, const int height,
                                  const int width, const int channel) {
  g_last_status =
      ssg_compute_similarity_backward(image, grads, pos, image_grads, mc, psize, ksize, height, width, channel, nullptr);
}

extern "C" {
void ssg_ref_compute_similarity(const float *image, const int *pos, float *out, int mc, int psize, int ksize,
                                int height, int width, int channel) {
  _compute_similarity(image, pos, out, mc, psize, ksize, height, width, channel);
}
void ssg_ref_compute_similarity_backward(const float *image, const float *grads, const int *pos,
                                         float *image_grads, int mc, int psize, int ksize, int height, int width,
                                         int channel) {
  _compute_similarity_backward(image, grads, pos, image_grads, mc, psize, ksize, height, width, channel);
}
int ssg_last_status(void) { return g_last_status; }
}
