/*
 * similarity.h -- the reference operator's own C++ interface, exported by libssg_hip.so.
 *
 * GAN-Based-SR/basicsr/losses/similarity/similarity.h:2-23 declares two free functions (C++ linkage, no
 * stream, no status) that similaritywrapper.cpp:27-37,59-70 calls; the reference implements them in
 * similarity.cu:56-70,133-148 with CUDA kernels on the legacy default stream.  libssg_hip.so exports the same
 * two symbols with the same parameter lists and the same C++ linkage, so the reference's pybind glue links
 * against it unchanged: build similaritywrapper.cpp alone (drop similarity.cu from `sources`) and add
 * -L<dir> -lssg_hip.  Argument meaning as in include/ssg_hip.h group (A): `image` is the reflect-padded
 * (channel, height, width) fp32 device image, `pos` holds mc int32 (Y, X) pairs in padded coordinates, `out`
 * (mc, psize, psize) / `image_grads` (channel, height, width) are pre-zeroed by the caller and accumulated
 * into.  Launches go to the legacy default stream (stream 0), like the reference's; errors cannot be returned
 * through a void function, the status of the last call on this thread is ssg_last_status().
 *
 * For callers that cannot link C++ symbols, the extern "C" aliases below take the same arguments.
 */
#ifndef SSG_REFERENCE_SIMILARITY_H
#define SSG_REFERENCE_SIMILARITY_H

#ifdef __cplusplus
void _compute_similarity(const float *image, const int *pos, float *out, const int mc, const int psize,
                         const int ksize, const int height, const int width, const int channel);

void _compute_similarity_backward(const float *image, const float *grads, const int *pos, float *image_grads,
                                  const int mc, const int psize, const int ksize, const int height,
                                  const int width, const int channel);
extern "C" {
#endif

void ssg_ref_compute_similarity(const float *image, const int *pos, float *out, int mc, int psize, int ksize,
                                int height, int width, int channel);
void ssg_ref_compute_similarity_backward(const float *image, const float *grads, const int *pos,
                                         float *image_grads, int mc, int psize, int ksize, int height, int width,
                                         int channel);
/* status (include/ssg_hip.h convention) of the most recent of the four functions above on this thread */
int ssg_last_status(void);

#ifdef __cplusplus
}
#endif
#endif /* SSG_REFERENCE_SIMILARITY_H */
