/*
 * ssg_hip.h -- C ABI of the MI355X (gfx950) Self-Similarity-Graph loss engine.
 *
 * This is the drop-in boundary for the ONE hot path of ChrisDud0257/SSL: the
 * SSG loss.  Everything below is `extern "C"`, takes
 * plain DEVICE pointers and sizes, allocates nothing, never synchronises the
 * host, and launches on the hipStream_t the caller passes (pass
 * torch.cuda.current_stream().cuda_stream from PyTorch).  All floating point
 * is IEEE fp32; indices are int32.
 *
 * Stream semantics: everything a call queues is ordered after the work already on
 * `stream` and before whatever the caller queues on `stream` next.  Inside a call
 * the direct kernel of a forward / backward pass (k_s <= 25) runs on a library-owned
 * side stream beside the dense-tile kernel, forked from and joined back into
 * `stream` with events -- invisible to the caller, and a stream capture of `stream`
 * records it as two parallel graph branches.  SSG_OVERLAP=0 (environment, read at
 * first use) keeps every launch on `stream` itself.
 *
 * Reference interface each group replaces (paths relative to
 * /root/reference/GAN-Based-SR/):
 *   (A) ssg_compute_similarity[_backward]   <- basicsr/losses/similarity/similarity.h:2-23
 *                                              (kernels similarity.cu:6-54, 74-131; pybind
 *                                              glue similaritywrapper.cpp:9-72)
 *   (B) ssg_edge_*                          <- scripts/data_preparation/generate_mask.py:22-31,
 *                                              torch.where / torch.nonzero in loss_util.py:196 and
 *                                              similaritywrapper.py:64-68, mask_stride pattern
 *                                              realesrganssl_model.py:64-72
 *   (C) ssg_map_forward / ssg_map_backward  <- similarity_map.ssl_pytorch / ssl_cuda,
 *                                              basicsr/losses/loss_util.py:182-244 (+ autograd)
 *   (D) ssg_loss_*                          <- the caller loop realesrganssl_model.py:379-430
 *                                              with L1Loss (basic_loss.py:41-66) and
 *                                              KLDistanceLoss (basic_loss.py:269-282)
 *   (E) ssg_augment_crop / ssg_pool_swap    <- basicsr/data/transforms.py:93-219 (joint flip/rot90 + crop of
 *                                              image and mask), realesrganssl_model.py:327-367 (pair pool)
 *
 * Return value of every int function: 0 on success, a positive hipError_t
 * from the launch, or a negative SSG_E_* code.  ssg_status_string() maps both.
 */
#ifndef SSG_HIP_H
#define SSG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ssg_stream_t; /* hipStream_t */

#define SSG_E_BADARG (-1)     /* null pointer, even / non-positive kernel size, k_w > k_s ...   */
#define SSG_E_TOOLARGE (-2)   /* search tile does not fit the 160 KiB LDS of a CU               */
#define SSG_E_WORKSPACE (-3)  /* caller's workspace is smaller than ssg_loss_workspace_bytes()   */
#define SSG_E_IMAGESMALL (-4) /* H or W <= k_s/2: reflect padding undefined (torch raises too)   */
#define SSG_E_ALIGN (-5)      /* fused step: workspace / grad_fix / grad_sr not 16-byte aligned      */
#define SSG_E_PLAN (-6)       /* ssg_device_status(): a plan cut for another tile height was used   */

/* 4 (round 4): + ssg_device_status, ssg_criteria_sums / _grad / _scratch_bytes, SSG_E_ALIGN, SSG_E_PLAN; a fused step
 * whose edge count exceeds its capacity returns NaN losses; a plan of the wrong tile height no longer traps. */
/* 5 (round 5): + ssg_set_overlap (modes 0-3, default 3 = per pass from the last plan's shape), ssg_last_overlap_assignment;
 * the product library no longer reads ANY environment variable (SSG_DENSE_THR, SSG_OVERLAP, SSG_OP_PLAN_FROM ... are
 * honoured by the profiling build only); the one-wave tile-major dense backward is gone. */
/* 6 (round 6): + ssg_set_tiny_step; default dense threshold 16. */
int ssg_abi_version(void);
const char *ssg_status_string(int status);
/* Device-side refusals that no return value can carry (everything is asynchronous): waits for `stream`, then returns
 * SSG_E_PLAN if, since the last call, a dense-tile kernel of ssg_map_forward / ssg_map_backward / ssg_loss_backward was
 * handed a fwd_plan cut for another tile height than its k_s uses (ssg_edge_list's plan_ks: 8-row tiles for k_s <= 25,
 * 4-row tiles for k_s = 49) -- such a launch leaves its rows / gradient untouched instead of decoding tile ids with
 * the wrong geometry -- else 0.  Clears the word.  The status word (4 bytes per device) is the one thing the library
 * allocates, at the first forward / backward call that takes a plan.  ssg_loss_fwd_bwd / ssg_loss_step build their own
 * plan and cannot set it. */
int ssg_device_status(ssg_stream_t stream);

/* ---------------------------------------------------------------- (A) ----
 * Reference operator, same argument meaning as similarity.h:2-11 plus a
 * stream and a status.  `image` is the REFLECT-PADDED (channel,height,width)
 * fp32 image, `pos` holds mc (Y,X) int32 pairs in PADDED coordinates, `out`
 * is (mc,psize,psize) and is ACCUMULATED into (the reference kernel does
 * `out[...] += d*d` on a zeroed buffer).  Raw squared patch distances, no
 * epilogue.  Asynchronous. */
/* Calls with many positions build the engine's work split inside the call: from ssg_set_operator_plan_threshold()
 * positions on for the backward and three times as many for the forward (default 8192 / 24576; n <= 0: never; 1:
 * always, both; environment SSG_OP_PLAN_FROM) and for the sizes that have shared-term
 * kernels ((25,9,3), (49,13,3)) the position list becomes a rank map and a plan -- the buffers come from a
 * library-owned, stream-ordered memory pool (hipMallocFromPoolAsync / hipFreeAsync on `stream`: the reference interface
 * has no workspace argument) -- and the tiles holding many positions go through the shared-term kernels, the others
 * through the direct kernels in tile order; `pos` may be in any order and may repeat positions (each row is still
 * computed).  Same results either way (raw distances rel 2e-6).  Below the threshold, and inside a stream capture,
 * the direct kernels walk `pos` as it comes.  Measured, launches only (profiles/r4_operator_plan_path.txt): 4,697
 * positions forward 0.056 ms direct / 0.206 with the plan, backward 0.158 / 0.167; 18,417 positions forward 0.196 /
 * 0.231, backward 0.462 / 0.258.  Returns the previous threshold. */
int ssg_set_operator_plan_threshold(int positions);
/* The pool keeps what the largest call needed (mask, rank map, plan: ~24 bytes per padded pixel + 20 per position; the
 * backward's scratch: 4 k_s^2 bytes per position) so that later calls allocate without a system call;
 * ssg_operator_pool_trim() hands the unused part back to the driver (after the work that used it has finished). */
int ssg_operator_pool_trim(void);
int ssg_compute_similarity(const float *image, const int *pos, float *out,
                           int mc, int psize, int ksize, int height, int width,
                           int channel, ssg_stream_t stream);

/* similarity.h:13-23: scatter of `grads` = dL/d(out) (mc,psize,psize) into
 * `image_grads` (channel,height,width, PADDED layout, accumulated into).  The
 * reference uses ~2*C*ksize^2 global atomics per (edge pixel, offset); this
 * one reduces per edge pixel on chip and issues one atomic per touched
 * pixel. */
int ssg_compute_similarity_backward(const float *image, const float *grads,
                                    const int *pos, float *image_grads, int mc,
                                    int psize, int ksize, int height, int width,
                                    int channel, ssg_stream_t stream);

/* ---------------------------------------------------------------- (B) ----
 * Edge list of a batch, built on device without a host round trip.
 *
 * mask_kind: 0 = fp32 mask (B,mask_channels,H,W), an edge pixel is
 *                `mask[b,0,y,x] == 1.0f` (loss_util.py:196 on channel 0, like
 *                ssl_cuda's mask[0,0], loss_util.py:233);
 *            1 = uint8 mask, same layout, `== 1` (a 0/255 PNG mask must be
 *                divided by 255 first, exactly as for the reference's
 *                `mask == 1`);
 *            2 = no mask tensor: `mask` is the fp32 GT batch (B,3,H,W) in
 *                [0,1] and the reference's offline mask is generated on the
 *                fly: u8 = round(255 x), L = PIL 'L' (ITU-R 601-2 16.16 fixed
 *                point), 4-neighbour Laplacian with BORDER_REFLECT_101
 *                saturated to uint8, `> lap_threshold` (generate_mask.py:22-31).
 * mask_stride > 1 additionally keeps only y % s == x % s
 * (realesrganssl_model.py:64-72).
 *
 * Outputs: edges  (capacity,3) int32 rows (b,y,x) in UNPADDED coordinates,
 *                 ordered by image then row-major -- the order of
 *                 torch.cat([...], dim=1) over torch.where;
 *          counts (B+2) int32: counts[0] = total N (NOT clamped to capacity;
 *                 N > capacity means overflow: re-run with a larger list),
 *                 counts[1+b] = first row of image b, counts[1+B] = N.
 *          rank_map (nullable) (B,H,W) int32: for every pixel the row of
 *                 `edges` that holds it, or -1.
 *          tile_order (nullable, needs rank_map) (capacity) int32: the rows
 *                 0..N-1 permuted tile-major (8x8 image tiles, row-major
 *                 inside a tile).  Passing it to the backward entry points
 *                 makes consecutive jobs spatial neighbours, whose gradient
 *                 tiles are then summed on chip before touching HBM (3.5x
 *                 fewer global atomics).  Rows of the SSG tensors keep the
 *                 reference's order either way.  Entries hold the row in bits
 *                 0..29; bit 30 of every 5th entry (first of a group of five
 *                 jobs) marks groups whose edge pixels share one image and an
 *                 8 x 16 pixel window (the kernels' on-chip sharing test).
 *          fwd_plan (nullable, needs rank_map) ssg_forward_plan_bytes() bytes:
 *                 the forward's work split -- 8x32-pixel tiles holding at
 *                 least ssg_set_dense_threshold() edge pixels go to the
 *                 shared-term ("dense") kernels, the remaining rows, in their
 *                 own tile-major order, to the direct kernels.  The plan records
 *                 its own split (with threshold 0: no dense tile, every row in
 *                 the direct order), so its consumers never read the process-wide
 *                 threshold.  Kernel sizes other than (25, 9, C=3) ignore it.
 * scratch: ssg_edge_scratch_bytes(B,H,W) bytes of device memory. */
size_t ssg_edge_scratch_bytes(int B, int H, int W);
size_t ssg_forward_plan_bytes(int B, int H, int W, int capacity);
/* Threshold (edge pixels per 8x32 tile) from which the forward routes a tile to the
 * shared-term kernel; 0 = never.  Default 16.  Results are the same either way (parity-tested
 * with every tile routed through it and with none).  Process-wide; takes effect at the
 * next ssg_edge_list().  Returns the previous value.  No reference counterpart: the
 * reference has one code path. */
int ssg_set_dense_threshold(int edge_pixels_per_tile);
/* Stream assignment of the two kernels of a pass for k_s <= 25 (a library-owned side stream, event fork / join on the
 * caller's stream; capturable).  0: every launch on the caller's stream (per-kernel profiling).  1: the dense-tile
 * kernel on the caller's stream, the direct kernel beside it on the side stream -- for masks whose dense tiles carry
 * most rows (Laplacian edge masks).  2: the other way round -- for masks without dense tiles (Bernoulli, thin strided
 * masks: the whole critical path on one stream; Bernoulli 1 % -15 %, C2 +3 %).  3 (default): 1 or 2 per pass, by the
 * shape of the last plan ssg_edge_list / the fused step built on the device -- its scan kernel leaves {rows for the
 * direct kernels, dense tiles} in host-mapped memory, the host reads it at the next pass without synchronising; the
 * branch expected to take longer (38 ns per direct row against 0.74 us per dense tile) stays on the caller's stream.
 * In mode 3 the fused steps (ssg_loss_fwd_bwd, ssg_loss_step) also run forward AND backward of the two branches as two
 * chains with ONE join at the end: free-running when the direct chain carries the step or the plan holds at most 512
 * dense tiles, otherwise with the direct backward held by a one-way event until the dense backward starts (C2 1.27 ->
 * 1.25 ms, C4 0.50 -> 0.46, Bernoulli 1 % 0.185 -> 0.175).
 * Same results, bit for bit in deterministic mode, in every mode and schedule (disjoint rows, integer sums at a scale that
 * does not depend on the schedule).  Process-wide; returns the previous setting.  No reference counterpart (the
 * reference launches on the legacy stream, similarity.cu:69,147). */
int ssg_set_overlap(int mode);
/* Diagnostics: the assignment the calling thread's last forked pass used (0 not forked, 1, 2 as above). */
int ssg_last_overlap_assignment(void);
/* Small fused steps.  ssg_loss_fwd_bwd / ssg_loss_step at (k_s, k_w, C) = (11, 5, 3) -- BASELINE's configs[0], the
 * reference's own CPU-runnable case (loss_util.py:185-229 on a 64 x 64 crop) -- with B*H*W <= 16,384 pixels and
 * capacity <= 4,096 rows run as TWO launches: one workgroup builds the edge list and clears the sums, then one workgroup
 * per edge pixel computes both SSG rows, the criteria and the row's gradient, and the last one through folds the step
 * (ssg_tiny.hip; C1: 59 -> 27 us).  Same outputs, workspace and error behaviour as the general path; results equal
 * to rounding (other summation orders), bit-reproducible in deterministic mode.  The workspace then holds the edge list and
 * the rank map but no tile order.  on = 0 keeps every call on the general path (default 1).  Process-wide; returns the
 * previous setting.  No reference counterpart. */
int ssg_set_tiny_step(int on);
int ssg_edge_list(const void *mask, int mask_kind, int mask_channels, int B,
                  int H, int W, int mask_stride, float lap_threshold,
                  int plan_ks /* k_s the fwd_plan is built for (tile rows: 8, or 4 for k_s = 49); 0 = 25.
                               * CONSTRAINT: a plan is only valid for calls whose k_s uses the same tile height --
                               * the plan records it (fwd_plan[2]) and the dense kernels of ssg_map_forward /
                               * ssg_map_backward / ssg_loss_backward do NOTHING when it differs from theirs
                               * (ssg_device_status() then returns SSG_E_PLAN), instead of decoding tile ids with
                               * the wrong geometry.  ssg_loss_fwd_bwd builds its own plan and cannot get this wrong. */,
                  int *edges, int capacity, int *counts,
                  int *rank_map /* nullable */, int *tile_order /* nullable */,
                  int *fwd_plan /* nullable */, void *scratch,
                  ssg_stream_t stream);

/* The mask itself (B,H,W) uint8 {0,1} from an fp32 GT batch (B,3,H,W):
 * mask_kind 2 above materialised (offline tool generate_mask.py). */
int ssg_edge_mask_laplacian(const float *gt, int B, int H, int W,
                            float lap_threshold, int mask_stride,
                            uint8_t *mask_out, ssg_stream_t stream);

/* ---------------------------------------------------------------- (C) ----
 * similarity_map forward for a batch of UNPADDED images (B,C,H,W) (reflect
 * padding is done by index mirroring, nothing is materialised):
 *   D  = patch distances, q = D/(C*k_w^2), e = exp(-1*q/sigma),
 *   s  = generalization ? 1/(sum_p e + eps) * e : e        (loss_util.py:224-227)
 * `edges` as produced by ssg_edge_list; n_edges_dev (nullable) points at the
 * device-side row count (counts[0]); at most n_rows rows are computed (the
 * host-known bound used to size the launch and `ssg`).  ssg is (n_rows,
 * k_s*k_s) fp32, row n <-> edges[n].  If img2/ssg2 are non-null the same
 * edge list is evaluated on a second batch in the same launch (SR and GT).
 * tile_order (from ssg_edge_list) only changes which workgroup computes which
 * row: neighbouring edge pixels then share one LDS search region.
 * With a fwd_plan, n_rows must cover the rows the plan was cut for: n_rows >= min(counts[0], capacity of the
 * ssg_edge_list call) -- the launches over the plan's heavy tiles are sized from n_rows (a heavy tile holds more than
 * 64 rows).  The same holds for ssg_map_backward / ssg_loss_backward. */
int ssg_map_forward(const float *img, const float *img2, int B, int C, int H,
                    int W, const int *edges, const int *tile_order /* nullable */,
                    const int *rank_map /* nullable */,
                    const int *fwd_plan /* nullable */,
                    const int *n_edges_dev, int n_rows,
                    int ks, int kw, float sigma, float eps, int generalization,
                    float *ssg, float *ssg2,
                    double *row_scale /* nullable, 2*n_rows doubles: see ssg_loss_backward */,
                    ssg_stream_t stream);

/* Backward of the above: grad_img (B,C,H,W) += d/d img of sum(grad_ssg * ssg)
 * (reflect fold included).  `ssg` is the forward output (saved).
 * With rank_map, fwd_plan (both from ssg_edge_list) and `scratch`
 * (ssg_backward_scratch_bytes(n_rows, ks) bytes) the backward is split like
 * the forward: dL/dD rows are formed once (ssg_grad_rows), the plan's dense
 * tiles go through the shared-term backward kernel and the remaining rows
 * through the direct one.  Any of the three null: direct kernel for every row. */
size_t ssg_backward_scratch_bytes(int n_rows, int ks);
int ssg_map_backward(const float *img, int B, int C, int H, int W,
                     const int *edges, const int *tile_order /* nullable */,
                     const int *rank_map /* nullable */,
                     const int *fwd_plan /* nullable */,
                     const int *n_edges_dev, int n_rows,
                     int ks, int kw, float sigma, int generalization,
                     const float *ssg, const float *grad_ssg, float *grad_img,
                     void *scratch /* nullable */, void *grad_fix /* nullable */,
                     ssg_stream_t stream);

/* Deterministic gradient accumulation.  By default the kernels add their per-pixel contributions to the image
 * gradient with hardware fp32 atomics: the order of the additions, hence the last bits of the result, varies
 * from run to run (as with the reference's atomicAdd, similarity.cu:123-128).  Passing `grad_fix` --
 * ssg_grad_fix_bytes(B,C,H,W) bytes of device memory, contents irrelevant -- to the backward entry points makes
 * the result bit-reproducible: contributions are rounded to multiples of a power of two chosen on the device
 * from an upper bound of |dL/dD| over the call (2^-35 of it; headroom for pixel differences up to 16) and summed
 * with 64-bit integer atomics (integer addition is associative), then folded into grad once per pixel.  The bound:
 * ssg_map_backward and the k_s = 49 tile-major step take the exact maximum (11 bits finer than an fp32 sum of the same
 * terms); the loss steps take the a-priori bound 4 (|w_l1| u_1 + |w_kl| u_2) / (sigma C k_w^2 n k_s^2) -- |s g| <= w_1 +
 * w_2 because s, t <= 1 -- which needs no pass over the rows and does not depend on how the step is scheduled
 * (ssg_set_overlap).  It lies further above the maximum the flatter the rows are; measured at sigma 0.004, 0.05 and 1,
 * (25,9) and (11,5): the deterministic gradient stays within 2e-7 max|grad| of the fp32-atomic one, which is the atomics'
 * own run-to-run spread (tools/r5_fix_resolution.py). */
size_t ssg_grad_fix_bytes(int B, int C, int H, int W);

/* ---------------------------------------------------------------- (D) ----
 * The whole loss step of the caller loop over a batch:
 *   l1 = w_l1 * mean |s_sr - s_gt|,  kl = w_kl * mean t'(log t' - log s')
 * with the mean over M = N * k_s^2 elements of the LOCAL batch (images with
 * an empty mask contribute nothing; N == 0 gives 0,0 like ddpmssl.py:492),
 * and grad_sr (B,C,H,W) += d(l1+kl)/d sr.
 *
 * ssg_loss_backward consumes SSGs already computed by ssg_map_forward
 * (API-compatible mode: SSG tensors are materialised once each); rank_map and
 * fwd_plan (nullable, from ssg_edge_list) select the split backward described
 * at ssg_map_backward (its scratch is part of ssg_loss_scratch_bytes).
 * Deferred normalisation (optional): when the SAME `row_scale` buffer (2*n_rows doubles) is passed to
 * ssg_map_forward and then here, the rows the dense-tile forward produced are left un-normalised
 * (e = exp(-d/sigma)) with their 1/(sum e + eps) in row_scale, and this call rescales them in place while it
 * streams them anyway -- same arithmetic, same bits, one pass over the SSG tensors less.  Between the two calls
 * ssg_sr / ssg_gt are NOT yet the SSG tensors.  Requires rank_map, fwd_plan and scratch (SSG_E_BADARG otherwise).
 * rows_are_scratch != 0: the caller will not look at ssg_sr / ssg_gt again (an autograd node that keeps only the
 * gradient): the rescaled rows are then not written back -- one write of every deferred row less.
 * `upstream`
 * (nullable) points at two DEVICE floats {dL/dl1, dL/dkl} that scale the two
 * criteria's gradients (autograd's incoming gradients, read on device so the
 * host never synchronises); null means {1,1}.
 * loss_out: 2 floats {l1, kl} on device.  scratch: ssg_loss_scratch_bytes(B,H,W,n_rows,ks).
 * PRECONDITION of the deterministic mode (grad_fix != NULL) at k_s <= 25: the rows are SSG rows -- every entry of
 * ssg_sr / ssg_gt in [0, 1] and every ssg_gt row summing to at most 1 (true for whatever ssg_map_forward produces,
 * normalised or not).  The fixed-point scale of the gradient sums is then taken from the a-priori bound
 * |dL/dD| <= 4 (|w_l1| u_1 + |w_kl| u_2) / (sigma C k_w^2 n k_s^2) instead of a maximum over the rows (no reduction
 * pass); rows outside those ranges (foreign tensors) can exceed the bound and wrap the 64-bit sums silently -- pass
 * grad_fix = NULL (fp32 atomics) for such input.  k_s = 49 with tile-major rows keeps the exact maximum. */
size_t ssg_loss_scratch_bytes(int B, int H, int W, int n_rows, int ks);
int ssg_loss_backward(const float *sr, int B, int C, int H, int W,
                      const int *edges, const int *tile_order /* nullable */,
                      const int *rank_map /* nullable */,
                      const int *fwd_plan /* nullable */,
                      const int *n_edges_dev, int n_rows,
                      int ks, int kw, float sigma, int generalization,
                      float *ssg_sr, float *ssg_gt, float w_l1,
                      float w_kl, const float *upstream /* nullable */,
                      float *loss_out, float *grad_sr /* nullable */,
                      void *scratch, void *grad_fix /* nullable: deterministic mode */,
                      const double *row_scale /* nullable */, int rows_are_scratch, ssg_stream_t stream);

/* The two criteria on tensors that already exist (L1Loss basic_loss.py:41-66 with l1_loss :14-16; KLDistanceLoss
 * basic_loss.py:269-282): sums_out[0] = sum |pred - target|, sums_out[1] = sum t' (log t' - log s') with
 * s' = max(pred, 1e-10), t' = max(target, 1e-10), over n fp32 elements, in one streaming pass (fp64 accumulation in a
 * fixed order: bit-reproducible); the caller applies reduction and loss_weight (mean = sum / n).  `scratch`:
 * ssg_criteria_scratch_bytes() bytes.  ssg_criteria_grad: grad_pred[i] = coef[0] * sign(pred - target) - coef[1] *
 * t'/s' (second term 0 where pred < 1e-10: the clamp's derivative), coef = 2 DEVICE floats (the upstream gradients
 * times weight / n: no host round trip).  The engine's fused entry points below compute the same terms inside their
 * row passes; these two serve SSG tensors that were materialised (eager `similarity_map`, INTEGRATION.md Level 1). */
size_t ssg_criteria_scratch_bytes(void);
int ssg_criteria_sums(const float *pred, const float *target, size_t n, void *scratch, float *sums_out,
                      ssg_stream_t stream);
int ssg_criteria_grad(const float *pred, const float *target, size_t n, const float *coef, float *grad_pred,
                      ssg_stream_t stream);

/* Everything in one call: edge list (from a mask or from GT's Laplacian),
 * SSG(sr), SSG(gt), both criteria and the gradient.  ssg_sr / ssg_gt
 * (capacity, k_s*k_s) receive the SSG tensors; `counts` as in ssg_edge_list;
 * workspace >= ssg_loss_workspace_bytes(B,H,W,capacity,ks).
 * ALIGNMENT: `workspace`, `grad_fix` and `grad_sr` must be 16-byte aligned (SSG_E_ALIGN otherwise): the edge-list
 * builder's first kernel clears the row scales, the fixed-point sums and (ssg_loss_step) the gradient with 16-byte
 * stores.  Fresh allocations are; an offset view into a larger buffer has to keep the alignment.
 * OVERFLOW: counts[0] > capacity means the step has used the first `capacity` edge pixels only; loss_out is then
 * {NaN, NaN}, so that a truncated step cannot pass for a complete one (grad_sr holds the truncated step's gradient).
 *
 * Fused step, no SSG output (SURVEY 8d's B_alg' mode): ssg_sr == ssg_gt == NULL.  The rows then live in the
 * workspace, which must hold ssg_loss_workspace_bytes(...) + ssg_loss_rows_bytes(capacity, ks); the dense-tile
 * forward leaves them un-normalised, ssg_grad_rows reads them once (rescaling in registers) and nothing normalised is
 * ever written -- one write and one read of every row instead of two writes and two reads (C5: 26 -> 21 GB per step).
 * Loss and gradient are bit-identical to the materialising call (row-major rows).  A kernel that kept a tile's rows
 * on chip through criteria and backward would need 64 edge pixels x k_s^2 x 2 images x 4 B = 320 KB at (25,9) --
 * twice the LDS of a CU -- or a second forward sweep; DESIGN.md section 8 has the measured costing.
 *
 * k_s = 49 (k_w 13, C 3, generalization): ssg_loss_rows_bytes() holds a second pair of regions for TILE-MAJOR rows,
 * [slot of the plan's dense-tile list][offset q][128 pixels of the 4 x 32 tile].  A call whose dense tiles fit the
 * region (count <= capacity / 128) and are >= 60 % full on average -- decided on the device from the plan's counts,
 * for all tiles of the call or none -- keeps the dense tiles' rows there: the forward stores whole 256-byte runs,
 * ssg_rows_tm reads them once (criteria, sum_q g s, border sums), and the dense backward forms G = dL/dD itself
 * from the two rows instead of reading G rows (C5: 21 -> 15.7 GB per step, 7.7 -> 5.7 ms); whole strips of nine heavy
 * tiles (36 x 32 pixels, consecutive slots) get their forward rows from ssg_fwd_strip.  Same loss and gradient as the
 * row-major step up to fp32 rounding: l1 <= 2e-7, kl <= 5e-6 relative; the gradient is within 5e-7 of the fp64
 * oracle's like the row-major step's, and within 4e-5 of its maximum of the row-major step's (the strip forward
 * rounds e differently, and a few L1 entries whose sign(s_sr - s_gt) fp32 does not decide flip).  Bit-reproducible
 * run to run in deterministic mode.  SSG_TILE_MAJOR=0 (environment) keeps every row row-major, SSG_STRIPS=0 leaves
 * the forward to the tile kernel.
 * A MATERIALISING k_s = 49 call (ssg_sr / ssg_gt given) uses the tile-major rows too when its workspace holds
 * ssg_loss_workspace_bytes() + ssg_loss_tm_bytes(): forward and backward as above, and the row pass
 * (ssg_rows_tm_mat) also writes the normalised SSG rows into the caller's tensors, 196-byte runs at a time.
 * ssg_loss_workspace_layout() reports where the pieces of the workspace live (byte offsets; `fused` = 1 for a call
 * without SSG output; tests and tools read the scratch rows of a finished call through it): out[0] edge list, [1]
 * rank map, [2] plan, [3] row scales (2 x capacity doubles, negative = tile-major row), [4] / [5] row-major scratch
 * rows of sr / gt (fused only), [6] / [7] tile-major regions of sr / gt (0 when the size has none), out[8] = slots
 * of a tile-major region. */
size_t ssg_loss_workspace_bytes(int B, int H, int W, int capacity, int ks);
size_t ssg_loss_rows_bytes(int capacity, int ks);
size_t ssg_loss_tm_bytes(int capacity, int ks);
int ssg_loss_workspace_layout(int B, int H, int W, int capacity, int ks, int fused, size_t out[9]);
int ssg_loss_fwd_bwd(const float *sr, const float *gt, const void *mask,
                     int mask_kind, int mask_channels, int B, int C, int H,
                     int W, int ks, int kw, float sigma, float eps,
                     int generalization, float w_l1, float w_kl, int mask_stride,
                     float lap_threshold, int capacity, float *ssg_sr /* nullable with ssg_gt: fused step */,
                     float *ssg_gt, int *counts, float *loss_out, float *grad_sr,
                     void *workspace, size_t workspace_bytes,
                     void *grad_fix /* nullable: deterministic mode */, ssg_stream_t stream);

/* ssg_loss_fwd_bwd with the gradient as an OUTPUT: grad_sr (B,C,H,W) is overwritten with d(l1+kl)/d sr instead of
 * accumulated into, so the caller does not clear it first (one fill kernel per step less: the deterministic mode's
 * final fold assigns it, the fp32-atomics mode has it cleared by the edge-list builder's first kernel).  Everything
 * else as ssg_loss_fwd_bwd.  N == 0 (every mask empty): grad_sr = 0. */
int ssg_loss_step(const float *sr, const float *gt, const void *mask,
                  int mask_kind, int mask_channels, int B, int C, int H,
                  int W, int ks, int kw, float sigma, float eps,
                  int generalization, float w_l1, float w_kl, int mask_stride,
                  float lap_threshold, int capacity, float *ssg_sr /* nullable with ssg_gt: fused step */,
                  float *ssg_gt, int *counts, float *loss_out, float *grad_sr,
                  void *workspace, size_t workspace_bytes,
                  void *grad_fix /* nullable: deterministic mode */, ssg_stream_t stream);

/* ---------------------------------------------------------------- (E) ----
 * The step before the loss, on the GPU (minimal slice): joint augmentation + crop of image and mask, and the
 * training pair pool.  Byte moves only, bit exact.
 *
 * ssg_augment_crop <- basicsr/data/transforms.py:152-219 `augment` (hflip, then vflip, then rot90 = transpose,
 * the same draw for image and mask) followed by transforms.py:93-149 `paired_random_crop_img_mask`: dst
 * (B,C,Ho,Wo) = crop at (top,left) of the augmented src (B,C,Hs,Ws); params (B,5) int32 on device:
 * top, left, hflip, vflip, rot90 per sample (top/left index the AUGMENTED image; the caller draws them like the
 * reference does and scales them for the GT side).  elem_bytes 4 (fp32 images / masks) or 1 (uint8 masks).
 *
 * ssg_pool_swap <- realesrganssl_model.py:327-367 `_dequeue_and_enqueue` for one tensor of the pool: samples
 * `slots[k]` of `queue` (Q samples of sample_bytes each) and sample k of `batch` change places, k < b. */
int ssg_augment_crop(const void *src, void *dst, int elem_bytes, int B, int C, int Hs, int Ws, int Ho, int Wo,
                     const int *params, ssg_stream_t stream);
int ssg_pool_swap(void *queue, void *batch, size_t sample_bytes, const int *slots, int b, ssg_stream_t stream);

/* USMSharp.forward (basicsr/utils/img_process_util.py:63-83; applied to every GT batch, realesrganssl_model.py:165,
 * 315): out = soft * clip(img + weight * (img - G*img), 0, 1) + (1 - soft) * img, soft = G * (|img - G*img| * 255 >
 * threshold), G = cv2.getGaussianKernel(radius | 1, sigma) x its transpose (sigma <= 0: OpenCV's rule 0.3 ((k-1)/2 -
 * 1) + 0.8), reflect padding.  img, out (B,C,H,W) fp32, out != img; H, W > radius / 2 (SSG_E_IMAGESMALL otherwise, as
 * F.pad raises); odd kernel size <= 63; scratch >= ssg_usm_scratch_bytes (3 planes of the batch).  Reference defaults:
 * radius 50, sigma 0, weight 0.5, threshold 10.  Floating point: within 2e-6 of the fp64 evaluation except where
 * |residual| * 255 is within rounding of `threshold` (the mask bit is then not determined at fp32). */
/* filter2D (basicsr/utils/img_process_util.py:7-31; the blur steps of the degradation chain,
 * realesrganssl_model.py:173,212,245,281,294): out[b,c] = reflect-padded img[b,c] correlated with kernels[b] (n_kernels
 * == B) or kernels[0] (n_kernels == 1); kernels (n_kernels, k, k) fp32, k odd <= 21 (ValueError for even k in the
 * reference: SSG_E_BADARG); H, W > k / 2; out != img.  Within 2e-6 of the fp64 evaluation for blur kernels (sum 1). */
int ssg_filter2d(const float *img, const float *kernels, float *out, int B, int C, int H, int W, int k, int n_kernels,
                 ssg_stream_t stream);
/* DiffJPEG(differentiable=False).forward (basicsr/utils/diffjpeg.py:449-487; realesrganssl_model.py:34,201,240,285,
 * 290): the JPEG simulation of the degradation chain on (B,3,H,W) fp32 RGB in [0,1] -- zero padding to multiples of
 * 16, YCbCr, 2x2 chroma average, 8x8 DCT, quantisation with torch.round at factor = quality_to_factor(quality),
 * inverse path, clamp, crop -- fused into one kernel (one wave per 16x16 macroblock).  quality_dev: B device floats
 * (the reference's tensor branch), or NULL to use the host scalar `quality` for every sample.  out may alias img.
 * Floating point: a DCT coefficient whose quotient lies within fp32 rounding of k + 1/2 may round the other way than in
 * another fp32 implementation (the reference's own included); everywhere else within 3e-6 of the fp64 evaluation. */
int ssg_diffjpeg(const float *img, float *out, int B, int H, int W, const float *quality_dev, float quality,
                 ssg_stream_t stream);
size_t ssg_usm_scratch_bytes(int B, int C, int H, int W);
int ssg_usm_sharp(const float *img, float *out, int B, int C, int H, int W, int radius, float sigma, float weight,
                  float threshold, void *scratch, size_t scratch_bytes, ssg_stream_t stream);

/* The element-wise / gather stages of the degradation chain (realesrganssl_model.py:168-297) between the kernels above
 * (ssl_amd/csrc/ssg_degrade.hip).  All take device pointers, launch on `stream`, never synchronise.
 *
 * ssg_resize: torch.nn.functional.interpolate(img, scale_factor=s | size=(Ho, Wo), mode) as the model calls it
 *   (:185,203,224,255,280,293; align_corners=False, antialias=False): mode 0 'area' (adaptive average pooling),
 *   1 'bilinear', 2 'bicubic' (A = -0.75).  scale_factor_h / _w: the scale_factor the caller would have passed to
 *   F.interpolate (the source coordinates then use 1 / scale_factor like torch, not Hi / Ho), or 0 for the size= form;
 *   Ho, Wo are always given (floor(Hi * scale_factor) for the scale_factor form).  out != img. */
int ssg_resize(const float *img, float *out, int B, int C, int Hi, int Wi, int Ho, int Wo, int mode,
               double scale_factor_h, double scale_factor_w, ssg_stream_t stream);
/* clip && rounds: torch.clamp((x * 255.0).round(), 0, 255) / 255. (realesrganssl_model.py:206,297; round half to even);
 * clip only: clamp(x, 0, 1); rounds only: (x * 255).round() / 255 -- the tail of add_*_noise_pt, degradations.py:501-507. */
int ssg_clamp_round(const float *img, float *out, size_t n, int clip, int rounds, ssg_stream_t stream);
/* add_gaussian_noise_pt (basicsr/data/degradations.py:455-507) with the random fields passed in: field_color (B,C,H,W) =
 * the torch.randn(b,c,h,w) draw, field_gray (H,W) = the torch.randn(h,w) draw (one field for the whole batch, as the
 * reference's broadcast has it) or NULL when no sample has gray noise; sigma (B), gray (B) in {0,1} on the device. */
int ssg_gaussian_noise(const float *img, float *out, const float *field_color, const float *field_gray,
                       const float *sigma, const float *gray, int B, int C, int H, int W, int clip, int rounds,
                       ssg_stream_t stream);
/* add_poisson_noise_pt (degradations.py:601-674) around the torch.poisson draws.  ssg_poisson_rates: the per-sample
 * level census (the reference's torch.unique, a host loop there), vals = 2^ceil(log2(levels)) -> vals (B,2) =
 * (colour, gray), and the rates img_r * vals the draws are taken from (rate_gray (B,1,H,W) nullable: only when a
 * sample has gray noise; C must be 3 then).  ssg_poisson_noise: the arithmetic after the draws. */
size_t ssg_poisson_scratch_bytes(int B);
int ssg_poisson_rates(const float *img, float *rate_color, float *rate_gray, float *vals, void *scratch, int B, int C,
                      int H, int W, ssg_stream_t stream);
int ssg_poisson_noise(const float *img, float *out, const float *draw_color, const float *draw_gray, const float *vals,
                      const float *scale, const float *gray, int B, int C, int H, int W, int clip, int rounds,
                      ssg_stream_t stream);

#ifdef SSG_PROFILE
/* PROFILING BUILD ONLY (libssg_hip_prof.so, compiled with -DSSG_PROFILE; the product library libssg_hip.so does not
 * export this symbol and has no code path that skips work).  Results are WRONG while a mask is set: skip kernel
 * phases or whole launches so that one kernel of a multi-kernel entry point can be timed with events on its stream.
 * Bits: 25 dense-tile forward, 26 direct forward, 27 dense-tile backward, 28 direct backward (split mode), 29 G rows;
 * lower bits ablate phases inside kernels (ssg_api.hip).  Returns the previous mask; 0 = production behaviour.  The
 * environment variable SSG_DEBUG_SKIP presets the mask, in this build only. */
int ssg_set_profile_mask(int mask);
/* workgroups per CU the HIP runtime reports for a kernel at its launch geometry (0: ssg_fwd_strip<49,13,3,3>) */
int ssg_prof_occupancy(int which);
/* ssg_fwd_strip's per-workgroup clocks of its last launch: host[3 i] = start, [3 i + 1] = end (s_memtime; comparable
 * within one XCD only), [3 i + 2] = XCC id << 32 | HW_ID, for the first n / 3 (<= 1024) workgroups */
int ssg_prof_strip_times(unsigned long long *host, int n);
/* on != 0: every launch of the library is preceded by a kernel that fills the LDS of every CU with `pattern` (a kernel
 * that reads an LDS word it has not written then sees the pattern, not a previous workgroup's data: audit of
 * uninitialised LDS reads, tools/r5_poison_suite.sh).  Results stay correct.  Returns the previous switch. */
int ssg_prof_set_lds_poison(int on, unsigned pattern);
#endif

/* Host helper for profiling builds: name of the HIP kernel a configuration
 * dispatches to ("ssg_fwd<25,9,5>", "ssg_fwd_generic", ...). */
const char *ssg_kernel_name(int ks, int kw, int backward);

#ifdef __cplusplus
}
#endif
#endif /* SSG_HIP_H */
