"""On-disk formats of the edge masks.

The reference prepares masks offline (GAN-Based-SR/scripts/data_preparation/generate_mask.py:17-41,
Diffusion-Based-SR/scripts/data_preparation/generate_mask_simmatrix.py:22-92): per image a MATLAB v5
file `mat/<name>.mat` holding an integer (H,W) array {0,1} under the key 'mat' (zlib-compressed), a
preview `png/<name>.png` (0/255), and -- in the Diffusion fork -- a density report `statis.txt`.
The datasets read the .mat back as float32 (H,W,1) (my_realesrgan_image_mask_dataset.py:79-83,
paired_image_mask_dataset.py:67-78).

These helpers are host-side file I/O only; the masks themselves come from the HIP Laplacian kernel
(`ssl_amd.engine.edge_mask_laplacian`, C ABI `ssg_edge_mask_laplacian`) via scripts/generate_mask.py.
"""
import os

import numpy as np
import scipy.io as sio

MAT_KEY = "mat"


def save_mask_mat(path, mask):
    """(H,W) {0,1} -> MATLAB v5 file, key 'mat', platform int like the reference's np.zeros(dtype='int'),
    compressed (generate_mask.py:28-41)."""
    m = np.ascontiguousarray(np.asarray(mask)).astype(np.int64)
    if m.ndim != 2:
        raise ValueError(f"mask must be (H,W), got {m.shape}")
    sio.savemat(path, {MAT_KEY: m}, do_compression=True)


def load_mask_mat(path):
    """-> float32 (H,W,1) contiguous, as the reference datasets hand it to the augment/crop transforms."""
    m = np.ascontiguousarray(np.array(sio.loadmat(path)[MAT_KEY]).astype(np.float32))
    if m.ndim == 2:
        m = np.expand_dims(m, axis=2)
    return m


def save_mask_png(path, mask):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(np.clip(np.asarray(mask).astype(np.int64) * 255, 0, 255).astype(np.uint8))) \
        .save(path, "PNG")


class DensityReport:
    """`statis.txt` of generate_mask_simmatrix.py:22-92: one entry per image (pixels, Laplacian > 0 count,
    mask count with fractions) and the max / min / average lines.  Like the reference, the max/min fractions
    are taken relative to the LAST image's pixel count (its `h*w*c` after the loop) and two of the labels
    say 'grad' where the mask statistics are meant; both quirks are kept so the reports diff clean."""

    def __init__(self, path):
        self.path = path
        self.f = open(path, mode="w", encoding="utf-8")
        self.sum_img, self.sum_grad, self.sum_mask, self.last = [], [], [], 0

    def add(self, name, num_pixels, num_grad, num_mask):
        self.sum_img.append(int(num_pixels)); self.sum_grad.append(int(num_grad)); self.sum_mask.append(int(num_mask))
        self.last = int(num_pixels)
        self.f.write(f"{name}:\n")
        self.f.write(f"Image number-{num_pixels}, grad number-{num_grad}-{num_grad / num_pixels:.4f}, "
                     f"mask number-{num_mask}-{num_mask / num_pixels:.4f}\n\n")

    def summary_lines(self):
        g, m, n, hwc = self.sum_grad, self.sum_mask, self.sum_img, self.last
        return [f"Maximum of grad is {max(g):.2f}, percentage is {max(g) / hwc:4f}",
                f"Minium of grad is {min(g):.2f}, percentage is {min(g) / hwc:.4f}",
                f"Average of grad is {sum(g) / len(g):.2f}, percentage is {sum(g) / sum(n):.4f}",
                f"Maximum of mask is {max(m):.2f}, percentage is {max(m) / hwc:4f}",
                f"Minium of grad is {min(m):.2f}, percentage is {min(m) / hwc:.4f}",
                f"Average of grad is {sum(m) / len(m):.2f}, percentage is {sum(m) / sum(n):.4f}"]

    def close(self):
        lines = self.summary_lines() if self.sum_img else []
        for l in lines:
            self.f.write(l + "\n")
        self.f.close()
        return lines
