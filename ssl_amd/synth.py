"""Deterministic synthetic inputs for the SSG-loss benchmark and parity tests.

There is no dataset on the build or GPU boxes, so BASELINE.json's configs are
driven by synthetic crops: "natural-like" fp32 images in
[0,1] quantised to k/255 -- a low-frequency colour field plus random step
edges (rectangles / half-planes) and a few textured patches -- tuned so that
the reference's offline edge mask (Laplacian of the PIL 'L' channel > 20,
GAN-Based-SR/scripts/data_preparation/generate_mask.py:22-31) has a density of
about 8 %.  GT = image, SR = clip(GT + 0.02 * blurred Gaussian noise).

Pure numpy: this is host-side data generation, not part of the compute path.
"""
import hashlib

import numpy as np


def _bilinear_up(a, H, W):
    """(C,h,w) -> (C,H,W) bilinear, align_corners=True style."""
    C, h, w = a.shape
    ys = np.linspace(0, h - 1, H)
    xs = np.linspace(0, w - 1, W)
    y0 = np.floor(ys).astype(int).clip(0, h - 2)
    x0 = np.floor(xs).astype(int).clip(0, w - 2)
    fy = (ys - y0)[None, :, None]
    fx = (xs - x0)[None, None, :]
    a00 = a[:, y0][:, :, x0]
    a01 = a[:, y0][:, :, x0 + 1]
    a10 = a[:, y0 + 1][:, :, x0]
    a11 = a[:, y0 + 1][:, :, x0 + 1]
    return (a00 * (1 - fy) * (1 - fx) + a01 * (1 - fy) * fx +
            a10 * fy * (1 - fx) + a11 * fy * fx)


def _box3(a):
    """3x3 box blur with edge replication on the last two axes."""
    p = np.pad(a, [(0, 0)] * (a.ndim - 2) + [(1, 1), (1, 1)], mode="edge")
    out = np.zeros_like(a)
    H, W = a.shape[-2:]
    for dy in range(3):
        for dx in range(3):
            out += p[..., dy:dy + H, dx:dx + W]
    return out / 9.0


def gray_l_u8(rgb_u8):
    """PIL Image.convert('L') on (..., 3) uint8: ITU-R 601-2, 16.16 fixed point."""
    r = rgb_u8[..., 0].astype(np.uint32)
    g = rgb_u8[..., 1].astype(np.uint32)
    b = rgb_u8[..., 2].astype(np.uint32)
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def laplacian_edge_mask(img_chw, threshold=20.0):
    """Edge mask of generate_mask.py:22-31 from a float (3,H,W) image in [0,1].

    u8 = round(255 x); L = PIL 'L'; lap = 4-neighbour Laplacian with
    BORDER_REFLECT_101 saturated to uint8; mask = lap > threshold.
    Returns (H,W) uint8 {0,1}.
    """
    u8 = np.rint(np.clip(np.asarray(img_chw, np.float32) * np.float32(255.0), 0, 255)).astype(np.uint8)
    L = gray_l_u8(np.moveaxis(u8, 0, -1)).astype(np.int32)
    p = np.pad(L, 1, mode="reflect")
    lap = p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4 * L
    lap = np.clip(lap, 0, 255)
    return (lap.astype(np.float32) > np.float32(threshold)).astype(np.uint8)


def mask_stride_pattern(H, W, s):
    """torch.eye(s).repeat(...)[:H,:W] (realesrganssl_model.py:64-70): y%s == x%s."""
    if s <= 1:
        return np.ones((H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    return ((yy % s) == (xx % s)).astype(np.uint8)


def natural_like(seed, H=256, W=256, target_density=0.08, tol=0.01):
    """One GT crop (3,H,W) float32 in {k/255} whose edge-mask density ~ target.

    Shapes are added one at a time (deterministically from `seed`) until the
    Laplacian mask density enters [target - tol, target + tol] or passes it.
    """
    rng = np.random.default_rng(seed)
    low = rng.random((3, max(H // 32, 2) + 1, max(W // 32, 2) + 1))
    img = 0.30 + 0.40 * _bilinear_up(low, H, W)
    yy, xx = np.mgrid[0:H, 0:W]
    best = None
    for _ in range(400):
        kind = rng.integers(0, 3)
        delta = rng.uniform(0.10, 0.35, size=(3, 1, 1)) * rng.choice([-1.0, 1.0], size=(3, 1, 1))
        if kind == 0:  # axis-aligned rectangle
            h = int(rng.integers(H // 16, H // 3))
            w = int(rng.integers(W // 16, W // 3))
            y0 = int(rng.integers(-h // 2, H - h // 2))
            x0 = int(rng.integers(-w // 2, W - w // 2))
            region = (yy >= y0) & (yy < y0 + h) & (xx >= x0) & (xx < x0 + w)
        elif kind == 1:  # disc
            r = rng.uniform(H / 24, H / 6)
            cy, cx = rng.uniform(0, H), rng.uniform(0, W)
            region = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        else:  # textured patch: fine noise inside a rectangle
            h = int(rng.integers(H // 16, H // 5))
            w = int(rng.integers(W // 16, W // 5))
            y0 = int(rng.integers(0, max(H - h, 1)))
            x0 = int(rng.integers(0, max(W - w, 1)))
            region = (yy >= y0) & (yy < y0 + h) & (xx >= x0) & (xx < x0 + w)
            delta = rng.normal(0.0, 0.06, size=(1, H, W)) * np.ones((3, 1, 1))
        trial = np.clip(img + delta * region[None], 0.0, 1.0)
        q = (np.rint(trial * 255.0) / 255.0).astype(np.float32)
        d = float(laplacian_edge_mask(q).mean())
        if d > target_density + tol:
            continue  # this shape overshoots: skip it, try another
        img = trial
        best = (q, d)
        if d >= target_density - tol:
            break
    if best is None:
        q = (np.rint(np.clip(img, 0, 1) * 255.0) / 255.0).astype(np.float32)
        best = (q, float(laplacian_edge_mask(q).mean()))
    return best[0]


def degrade(gt, seed, amp=0.02):
    """SR stand-in: clip(GT + amp * 3x3-blurred Gaussian noise, 0, 1), float32."""
    rng = np.random.default_rng(seed)
    noise = _box3(rng.normal(0.0, 1.0, size=gt.shape)) * 3.0  # unit-ish variance after blur
    return np.clip(gt.astype(np.float64) + amp * noise, 0.0, 1.0).astype(np.float32)


def make_batch(batch=16, H=256, W=256, seed0=100, target_density=0.08):
    """(sr, gt, mask): (B,3,H,W) f32, (B,3,H,W) f32, (B,1,H,W) f32 {0,1}.

    Image i uses seed seed0+i; the mask is the
    reference's Laplacian edge mask of GT.
    """
    gts, srs, masks = [], [], []
    for i in range(batch):
        gt = natural_like(seed0 + i, H, W, target_density)
        gts.append(gt)
        srs.append(degrade(gt, 10_000 + seed0 + i))
        masks.append(laplacian_edge_mask(gt)[None].astype(np.float32))
    return np.stack(srs), np.stack(gts), np.stack(masks)


def uniform_case(H=64, W=64, n_edges=205):
    """BASELINE.json configs[0]: uniform-noise 1x3xHxW pair + fixed 5 % mask.

    SR from default_rng(0), GT from default_rng(1); mask = n_edges coordinates
    from default_rng(2).choice(H*W) united with the 4 corners.
    """
    sr = np.random.default_rng(0).random((1, 3, H, W), dtype=np.float32)
    gt = np.random.default_rng(1).random((1, 3, H, W), dtype=np.float32)
    idx = np.random.default_rng(2).choice(H * W, n_edges, replace=False)
    m = np.zeros(H * W, np.float32)
    m[idx] = 1.0
    m = m.reshape(H, W)
    m[0, 0] = m[0, W - 1] = m[H - 1, 0] = m[H - 1, W - 1] = 1.0
    return sr, gt, m[None, None]


def checksum(*arrays):
    """Short sha256 over the raw bytes: detects silent generator drift."""
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]
