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
