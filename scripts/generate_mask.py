#!/usr/bin/env python
"""Offline edge-mask preparation on the GPU: the reference's generate_mask.py (GAN-Based-SR/scripts/
data_preparation/generate_mask.py:17-41) and, with --statis, the Diffusion fork's
generate_mask_simmatrix.py (mode 'L', type 'Laplacian'; :22-92) as one batch tool.

    python scripts/generate_mask.py --input <dir of images> --save <out dir> [--threshold 20] [--statis]

For every image: PIL 'L' luma -> 3x3 Laplacian (BORDER_REFLECT_101, saturated to uint8) -> > threshold,
computed by the HIP kernel behind `ssg_edge_mask_laplacian` (no OpenCV); writes <save>/mat/<name>.mat
(key 'mat', int, compressed), <save>/png/<name>.png and optionally <save>/statis.txt.  Images of equal
size are sent to the device as one batch.  Needs the MI355X: there is no CPU path.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, maskio  # noqa: E402


def masks_of(rgb_u8_list, threshold, device):
    """list of (H,W,3) uint8 of one size -> (mask (B,H,W) uint8, grad>0 (B,H,W) uint8) from the HIP kernel."""
    x = torch.as_tensor(np.stack(rgb_u8_list), device=device).permute(0, 3, 1, 2).to(torch.float32).div_(255.0)
    x = x.contiguous()
    return (engine.edge_mask_laplacian(x, lap_threshold=threshold).cpu().numpy(),
            engine.edge_mask_laplacian(x, lap_threshold=0.0).cpu().numpy())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", required=True)
    ap.add_argument("--save", required=True)
    ap.add_argument("--threshold", type=float, default=20.0)
    ap.add_argument("--statis", action="store_true", help="also write statis.txt (density report)")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args(argv)
    from PIL import Image
    if not torch.cuda.is_available():
        raise RuntimeError("scripts/generate_mask.py runs the HIP Laplacian kernel: no GPU visible")
    dev = torch.device("cuda:0")
    os.makedirs(os.path.join(args.save, "mat"), exist_ok=True)
    os.makedirs(os.path.join(args.save, "png"), exist_ok=True)
    report = maskio.DensityReport(os.path.join(args.save, "statis.txt")) if args.statis else None
    files = list(os.listdir(args.input))          # the reference's order: os.listdir
    # the kernel takes RGB and forms PIL's 'L' itself ((R*19595 + G*38470 + B*7471 + 0x8000) >> 16); a
    # greyscale file is replicated, for which that formula is the identity
    imgs = [np.array(Image.open(os.path.join(args.input, f)).convert("RGB")) for f in files]
    done = {}
    by_shape = {}
    for i, im in enumerate(imgs):
        by_shape.setdefault(im.shape, []).append(i)
    for shape, idx in by_shape.items():
        for s in range(0, len(idx), args.batch):
            chunk = idx[s:s + args.batch]
            m, g = masks_of([imgs[i] for i in chunk], args.threshold, dev)
            for k, i in enumerate(chunk):
                done[i] = (m[k], g[k])
    for i, f in enumerate(files):
        name = os.path.splitext(f)[0]
        m, g = done[i]
        maskio.save_mask_png(os.path.join(args.save, "png", f"{name}.png"), m)
        maskio.save_mask_mat(os.path.join(args.save, "mat", f"{name}.mat"), m)
        if report:
            report.add(name, m.size, int(g.sum()), int(m.sum()))
    if report:
        for line in report.close():
            print(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
