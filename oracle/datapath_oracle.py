"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

numpy restatement of the data-path slice, each function citing the reference lines it
follows (GAN-Based-SR/).  Pinned by tests/golden/f11_datapath.npz, produced by running the reference's own
basicsr/data/transforms.py (cv2.flip stubbed with its numpy equivalent: cv2 is not installed) and the
`_dequeue_and_enqueue` method of basicsr/models/realesrganssl_model.py (tests/golden/make_golden.py f11).
"""
import numpy as np


def augment(img_hwc, hflip, vflip, rot90):
    """transforms.py:178-187 `_augment`: horizontal flip, vertical flip, transpose(1, 0, 2), in that order."""
    a = img_hwc
    if hflip:
        a = a[:, ::-1]
    if vflip:
        a = a[::-1]
    if rot90:
        a = a.transpose(1, 0, 2)
    return np.ascontiguousarray(a)


def crop_nchw(x, top, left, size):
    """transforms.py:127,134,140 (Tensor branch): v[:, :, top:top+size, left:left+size]."""
    return np.ascontiguousarray(x[:, :, top:top + size, left:left + size])


def augment_crop_nchw(x, top, left, out_hw, flips):
    """per-sample augment (on HWC views) then crop, returned NCHW: the composition ssg_augment_crop computes."""
    out = []
    for b in range(x.shape[0]):
        a = augment(x[b].transpose(1, 2, 0), *flips[b]).transpose(2, 0, 1)
        out.append(a[:, top:top + out_hw[0], left:left + out_hw[1]])
    return np.ascontiguousarray(np.stack(out))


class PairPool:
    """realesrganssl_model.py:327-367, literally (the queue IS permuted here, like the reference's queue[idx])."""

    def __init__(self, queue_size):
        self.queue_size, self.ptr, self.q = queue_size, 0, None

    def exchange(self, tensors, idx_fn):
        """tensors: list of (b, ...) arrays; idx_fn() -> the permutation drawn when the pool is full."""
        b = tensors[0].shape[0]
        if self.q is None:
            self.q = [np.zeros((self.queue_size,) + t.shape[1:], t.dtype) for t in tensors]
        if self.ptr == self.queue_size:
            idx = idx_fn()
            self.q = [q[idx] for q in self.q]
            out = [q[:b].copy() for q in self.q]
            for q, t in zip(self.q, tensors):
                q[:b] = t
            return out
        for q, t in zip(self.q, tensors):
            q[self.ptr:self.ptr + b] = t
        self.ptr += b
        return [t.copy() for t in tensors]


def gaussian_kernel_1d(ksize, sigma=0.0):
    """cv2.getGaussianKernel(ksize, sigma) (n,) float64 as OpenCV documents it (imgproc, getGaussianKernel): the fixed
    tables for ksize <= 7 with sigma <= 0, else exp(-(i-(ksize-1)/2)^2 / (2 sigma^2)) normalised, sigma <= 0 meaning
    0.3*((ksize-1)*0.5 - 1) + 0.8.  cv2 is not installed: this function is pinned by the documentation only (the
    callers below are pinned by the reference's own code, fixture F12)."""
    fixed = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in fixed:
        return np.asarray(fixed[ksize], np.float64)
    sg = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-0.5 * x * x / (sg * sg))
    return k / k.sum()


def filter2d_reflect(img, k1d):
    """img_process_util.py:7-31 `filter2D` for one kernel shared by the batch: reflect-pad by k//2 (F.pad 'reflect'),
    correlate every (b, c) plane with the k x k kernel -- here with the separable factors of USMSharp's outer product
    (img_process_util.py:71), in the dtype of `img`."""
    r = len(k1d) // 2
    p = np.pad(img, ((0, 0), (0, 0), (r, r), (r, r)), mode="reflect")
    H, W = img.shape[-2:]
    t = sum(k1d[i] * p[..., :, i:i + W] for i in range(len(k1d)))
    return sum(k1d[i] * t[..., i:i + H, :] for i in range(len(k1d)))


def usm_sharp(img, radius=50, sigma=0.0, weight=0.5, threshold=10.0, return_parts=False):
    """USMSharp.__init__ + forward (img_process_util.py:63-83) on (B,C,H,W) in [0,1], computed in float64."""
    img = np.asarray(img, np.float64)
    ksize = radius + 1 if radius % 2 == 0 else radius
    k = gaussian_kernel_1d(ksize, sigma)
    blur = filter2d_reflect(img, k)
    residual = img - blur
    mask = (np.abs(residual) * 255 > threshold).astype(np.float64)
    soft = filter2d_reflect(mask, k)
    sharp = np.clip(img + weight * residual, 0, 1)
    out = soft * sharp + (1 - soft) * img
    return (out, residual, mask) if return_parts else out


def filter2d(img, kernels):
    """img_process_util.py:7-31 `filter2D`: img (B,C,H,W), kernels (B,k,k) or (1,k,k), k odd; reflect-pad by k//2 and
    correlate (F.conv2d does not flip) every plane of sample b with kernel b (the shared one).  float64."""
    img = np.asarray(img, np.float64)
    kernels = np.asarray(kernels, np.float64)
    k = kernels.shape[-1]
    if k % 2 != 1:
        raise ValueError("Wrong kernel size")
    r = k // 2
    B, C, H, W = img.shape
    p = np.pad(img, ((0, 0), (0, 0), (r, r), (r, r)), mode="reflect")
    out = np.zeros_like(img)
    for b in range(B):
        kb = kernels[0 if kernels.shape[0] == 1 else b]
        for ky in range(k):
            for kx in range(k):
                out[b] += kb[ky, kx] * p[b, :, ky:ky + H, kx:kx + W]
    return out
