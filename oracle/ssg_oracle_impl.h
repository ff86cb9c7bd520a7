/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's Self-Similarity-Graph (SSG)
 * loss hot path.  This header is a poor man's template: it is included twice
 * by ssg_oracle.c, once with REAL=float / SFX(x)=x##_f32 and once with
 * REAL=double / SFX(x)=x##_f64.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.  The shipped path is the HIP library in ssl_amd/csrc.
 *
 * Every function cites the reference file:line (relative to
 * /root/reference/GAN-Based-SR/) whose behaviour it restates.  Parity of this
 * restatement is PINNED by the .npz fixtures under tests/golden, which were produced by
 * importing the reference's own basicsr/losses/loss_util.py (see
 * tests/golden/make_golden.py) -- the reference itself holds no test or
 * golden vector for this path.
 */

/* F.pad(mode='reflect') index map (loss_util.py:189-191,
 * similaritywrapper.py:65): the border sample itself is not duplicated.
 * Valid while the pad is < n (torch enforces the same). */
static inline int SFX(orc_reflect)(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

/*
 * Raw squared patch distances D[n, py, px] on the UNPADDED image with reflect
 * handled by index mirroring.
 *
 * Restates basicsr/losses/similarity/similarity.cu:6-54 (one (edge pixel,
 * search offset) pair sums (A-B)^2 over c,kh,kw; B is replaced by 0 when
 * (py+kh, px+kw) leaves the k_s x k_s search area, lines 43-47), which is the
 * same function as the unfold path loss_util.py:189-223 (second F.unfold uses
 * ZERO padding, line 208).  pos holds (y, x) of each edge pixel in UNPADDED
 * coordinates, row-major like torch.where / torch.nonzero (loss_util.py:196,
 * similaritywrapper.py:67).
 */
void SFX(orc_distance)(const REAL *img, int C, int H, int W, const int *pos,
                       int N, int ks, int kw, REAL *D)
{
    const int hp = ks / 2, hk = kw / 2;
#pragma omp parallel for schedule(dynamic, 8)
    for (int n = 0; n < N; ++n) {
        const int y = pos[2 * n], x = pos[2 * n + 1];
        for (int py = 0; py < ks; ++py)
            for (int px = 0; px < ks; ++px) {
                REAL acc = 0;
                for (int c = 0; c < C; ++c) {
                    const REAL *ch = img + (size_t)c * H * W;
                    for (int kh = -hk; kh <= hk; ++kh)
                        for (int kx = -hk; kx <= hk; ++kx) {
                            const REAL a =
                                ch[(size_t)SFX(orc_reflect)(y + kh, H) * W +
                                   SFX(orc_reflect)(x + kx, W)];
                            REAL t;
                            if (py + kh < 0 || py + kh >= ks || px + kx < 0 ||
                                px + kx >= ks) {
                                t = a; /* similarity.cu:43-44 */
                            } else {
                                const REAL b =
                                    ch[(size_t)SFX(orc_reflect)(y - hp + py + kh, H) * W +
                                       SFX(orc_reflect)(x - hp + px + kx, W)];
                                t = a - b; /* similarity.cu:46 */
                            }
                            acc += t * t; /* similarity.cu:49 */
                        }
                }
                D[((size_t)n * ks + py) * ks + px] = acc;
            }
    }
}

/*
 * Literal restatement of the reference operator's C interface
 * (similarity.h:2-11): image is ALREADY reflect-padded (C, Hp, Wp), pos holds
 * (Y, X) in padded coordinates, out is (mc, psize, psize) and is ACCUMULATED
 * into (the reference kernel does `out[...] += tmp*tmp`, similarity.cu:49, on
 * a torch.zeros buffer, similaritywrapper.py:29).
 */
void SFX(orc_compute_similarity)(const REAL *image, const int *pos, REAL *out,
                                 int mc, int psize, int ksize, int height,
                                 int width, int channel)
{
    const int hp = (psize - 1) / 2, hk = (ksize - 1) / 2;
#pragma omp parallel for schedule(dynamic, 8)
    for (int n = 0; n < mc; ++n) {
        const int Y = pos[2 * n], X = pos[2 * n + 1];
        for (int py = 0; py < psize; ++py)
            for (int px = 0; px < psize; ++px) {
                REAL acc = 0;
                for (int c = 0; c < channel; ++c) {
                    const REAL *ch = image + (size_t)c * height * width;
                    for (int kh = -hk; kh <= hk; ++kh)
                        for (int kx = -hk; kx <= hk; ++kx) {
                            const REAL a = ch[(size_t)(Y + kh) * width + X + kx];
                            REAL t = a;
                            if (!(py + kh < 0 || py + kh >= psize ||
                                  px + kx < 0 || px + kx >= psize))
                                t = a - ch[(size_t)(Y - hp + py + kh) * width +
                                           (X - hp + px + kx)];
                            acc += t * t;
                        }
                }
                out[((size_t)n * psize + py) * psize + px] += acc;
            }
    }
}

/*
 * Literal restatement of similarity.h:13-23 / similarity.cu:74-131: scatter of
 * dL/dD into the PADDED image gradient (accumulated; the reference uses
 * atomicAdd on a torch.zeros buffer, similaritywrapper.py:47).  Summation
 * order here is fixed (n, p, c, kh, kw); the reference's is not.
 */
void SFX(orc_compute_similarity_backward)(const REAL *image, const REAL *grads,
                                          const int *pos, REAL *image_grads,
                                          int mc, int psize, int ksize,
                                          int height, int width, int channel)
{
    const int hp = (psize - 1) / 2, hk = (ksize - 1) / 2;
    for (int n = 0; n < mc; ++n) {
        const int Y = pos[2 * n], X = pos[2 * n + 1];
        for (int py = 0; py < psize; ++py)
            for (int px = 0; px < psize; ++px) {
                const REAL g = grads[((size_t)n * psize + py) * psize + px];
                if (g == 0) continue;
                for (int c = 0; c < channel; ++c) {
                    const size_t co = (size_t)c * height * width;
                    for (int kh = -hk; kh <= hk; ++kh)
                        for (int kx = -hk; kx <= hk; ++kx) {
                            const size_t ia = co + (size_t)(Y + kh) * width + X + kx;
                            if (py + kh < 0 || py + kh >= psize || px + kx < 0 ||
                                px + kx >= psize) {
                                image_grads[ia] += 2 * image[ia] * g; /* :123 */
                            } else {
                                const size_t ib = co +
                                    (size_t)(Y - hp + py + kh) * width +
                                    (X - hp + px + kx);
                                const REAL t = 2 * (image[ia] - image[ib]) * g;
                                image_grads[ia] += t; /* :126-127 */
                                image_grads[ib] -= t; /* :128 */
                            }
                        }
                }
            }
    }
}

/*
 * dL/dimg on the UNPADDED image given dL/dD, i.e. the op backward above
 * followed by autograd's backward of F.pad(reflect) (border pixels collect
 * their mirrors' gradients).  gI (C,H,W) is accumulated into.
 */
void SFX(orc_distance_backward)(const REAL *img, int C, int H, int W,
                                const int *pos, int N, int ks, int kw,
                                const REAL *gD, REAL *gI)
{
    const int hp = ks / 2, hk = kw / 2;
    const size_t sz = (size_t)C * H * W;
#pragma omp parallel
    {
        /* private accumulation image per thread, merged at the end */
        REAL *acc = (REAL *)calloc(sz, sizeof(REAL));
#pragma omp for schedule(dynamic, 8)
        for (int n = 0; n < N; ++n) {
            const int y = pos[2 * n], x = pos[2 * n + 1];
            for (int py = 0; py < ks; ++py)
                for (int px = 0; px < ks; ++px) {
                    const REAL g = gD[((size_t)n * ks + py) * ks + px];
                    if (g == 0) continue;
                    for (int c = 0; c < C; ++c) {
                        const size_t co = (size_t)c * H * W;
                        for (int kh = -hk; kh <= hk; ++kh)
                            for (int kx = -hk; kx <= hk; ++kx) {
                                const size_t ia = co +
                                    (size_t)SFX(orc_reflect)(y + kh, H) * W +
                                    SFX(orc_reflect)(x + kx, W);
                                if (py + kh < 0 || py + kh >= ks ||
                                    px + kx < 0 || px + kx >= ks) {
                                    acc[ia] += 2 * img[ia] * g;
                                } else {
                                    const size_t ib = co +
                                        (size_t)SFX(orc_reflect)(y - hp + py + kh, H) * W +
                                        SFX(orc_reflect)(x - hp + px + kx, W);
                                    const REAL t = 2 * (img[ia] - img[ib]) * g;
                                    acc[ia] += t;
                                    acc[ib] -= t;
                                }
                            }
                    }
                }
        }
#pragma omp critical
        for (size_t i = 0; i < sz; ++i) gI[i] += acc[i];
        free(acc);
    }
}

/*
 * Epilogue of ssl_pytorch / ssl_cuda (loss_util.py:224-227 and :234-242):
 *   q = D / (C * k_w^2);  e = exp(-1 * q / sigma);
 *   if generalization:  s = 1 / (sum_p e + eps) * e     (eps = 1e-10 there,
 *   1e-20 / 1e-6 in the Diffusion-Based-SR fork, loss_util.py:1250,775).
 * E (optional, may be NULL) receives the un-normalised e.
 */
void SFX(orc_ssg_epilogue)(const REAL *D, int N, int ks, int kw, int C,
                           REAL sigma, int generalization, REAL eps, REAL *S,
                           REAL *E)
{
    const int P = ks * ks;
    const REAL den = (REAL)C * (REAL)kw * (REAL)kw;
    for (int n = 0; n < N; ++n) {
        REAL sum = 0;
        for (int p = 0; p < P; ++p) {
            REAL q = D[(size_t)n * P + p] / den;
            REAL e = (REAL)exp((double)(-1 * q / sigma));
            S[(size_t)n * P + p] = e;
            if (E) E[(size_t)n * P + p] = e;
            sum += e;
        }
        if (generalization) {
            const REAL r = 1 / (sum + eps);
            for (int p = 0; p < P; ++p) S[(size_t)n * P + p] *= r;
        }
    }
}

/*
 * dL/dD from dL/dS (chain rule through the epilogue above).  With
 * generalization: G = -(s / (sigma*C*kw^2)) * (g - sum_p g*s); without:
 * G = -(e / (sigma*C*kw^2)) * g.   S is the epilogue's output.
 */
void SFX(orc_ssg_epilogue_backward)(const REAL *S, const REAL *gS, int N,
                                    int ks, int kw, int C, REAL sigma,
                                    int generalization, REAL *gD)
{
    const int P = ks * ks;
    const REAL k = 1 / (sigma * (REAL)C * (REAL)kw * (REAL)kw);
    for (int n = 0; n < N; ++n) {
        REAL dot = 0;
        if (generalization)
            for (int p = 0; p < P; ++p)
                dot += gS[(size_t)n * P + p] * S[(size_t)n * P + p];
        for (int p = 0; p < P; ++p) {
            const size_t i = (size_t)n * P + p;
            gD[i] = -(S[i] * k) * (gS[i] - dot);
        }
    }
}

/*
 * The two criteria applied to (SSG_sr, SSG_gt) over M = sum_i N_i * k_s^2
 * elements (realesrganssl_model.py:413-426):
 *   L1Loss  (basic_loss.py:41-66, l1_loss :14-16, 'mean'):
 *       w_l1 * mean |a - b|
 *   KLDistanceLoss (basic_loss.py:269-282):
 *       w_kl * F.kl_div(log(clamp(a,1e-10)), clamp(b,1e-10), 'mean')
 *       = w_kl * mean  b' * (log b' - log a')
 * out2 = {l1, kl}.  g_sr (optional) receives d(l1+kl)/d s_sr:
 *   w_l1*sign(a-b)/M  -  w_kl * b'/(a'*M) * [a >= 1e-10]   (clamp passes the
 *   gradient where the input is >= min).
 */
void SFX(orc_criteria)(const REAL *s_sr, const REAL *s_gt, long M, REAL w_l1,
                       REAL w_kl, REAL *out2, REAL *g_sr)
{
    const REAL cl = (REAL)1e-10;
    double l1 = 0, kl = 0; /* wide accumulators: this is the checker */
    for (long i = 0; i < M; ++i) {
        const REAL a = s_sr[i], b = s_gt[i];
        const REAL ac = a < cl ? cl : a, bc = b < cl ? cl : b;
        l1 += fabs((double)(a - b));
        kl += (double)bc * (log((double)bc) - log((double)ac));
        if (g_sr) {
            REAL g = 0;
            if (a > b) g += w_l1 / (REAL)M;
            else if (a < b) g -= w_l1 / (REAL)M;
            if (a >= cl) g -= w_kl * bc / (ac * (REAL)M);
            g_sr[i] = g;
        }
    }
    out2[0] = (REAL)(w_l1 * l1 / (double)M);
    out2[1] = (REAL)(w_kl * kl / (double)M);
}
