"""Torch-facing front end of the gfx950 SSG engine.

PyTorch is used for device memory, streams and autograd plumbing only; every
number is produced by the hand-written HIP kernels in ssl_amd/csrc through the
C ABI of include/ssg_hip.h.  Tensors must live on the GPU; there is no CPU
path (a CPU tensor raises, like the reference operator refuses non-CUDA
tensors, similaritywrapper.py:60-62 -- but with an exception instead of
sys.exit()).
"""
import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"ssl_amd: expected a GPU tensor, got device {t.device}; the SSG engine has no CPU path")


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def deterministic_default():
    """The Python host asks for the bit-reproducible gradient accumulation (include/ssg_hip.h, ssg_grad_fix_bytes)
    unless SSG_DETERMINISTIC=0 is set: measured +3 % on the benchmark step (1.64 vs 1.59 ms), for gradients that
    are equal bit for bit from run to run.  Every entry point also takes `deterministic=` explicitly."""
    import os
    return os.environ.get("SSG_DETERMINISTIC", "1") not in ("", "0")


def _grad_fix(det, x):
    """Fixed-point accumulation buffer of the deterministic mode for an image batch x, or None."""
    if det is None:
        det = deterministic_default()
    if not det:
        return None
    B, C, H, W = x.shape
    return torch.empty(_lib.lib().ssg_grad_fix_bytes(B, C, H, W), dtype=torch.uint8, device=x.device)


class FwdPlan(tuple):
    """(order, rank map, plan) as the forward / backward entry points take them, plus `.ks`: the search size the plan's
    dense tiles were cut for (8-row tiles for k_s <= 25, 4-row tiles for k_s = 49).  A plan built for one tile
    height must not be walked by the kernels of the other (they would decode tile ids with their own geometry):
    `check_plan` raises before anything is launched."""

    def __new__(cls, order, rank, plan, ks):
        self = super().__new__(cls, (order, rank, plan))
        self.ks = int(ks)
        return self


def _plan_rows(ks):
    return 4 if int(ks) == 49 else 8   # dense_tile_rows() of ssl_amd/csrc/ssg_dense.hip


def check_plan(fwd, ks):
    """ValueError when `fwd` (EdgeList.fwd) was built by edge_list(ks=...) for another dense-tile height than the
    kernels of this call's k_s use."""
    built = getattr(fwd, "ks", None)
    if fwd is not None and built is not None and _plan_rows(built) != _plan_rows(ks):
        raise ValueError(f"ssl_amd: this edge list's dense/direct plan was built for k_s = {built} "
                         f"({_plan_rows(built)}-row tiles) but the call uses k_s = {ks} ({_plan_rows(ks)}-row tiles); "
                         f"build it with edge_list(..., ks={ks})")


class EdgeList(tuple):
    """(edges, counts) -- unpacks like the pair it always was -- plus `.rank`, the (B,H,W) int32
    rank map (row of `edges` holding each pixel, -1 elsewhere), and `.order`, the tile-major
    permutation of the rows that the backward kernels use as their job order, and `.plan`, the
    forward's work split between the dense-tile kernel and the direct kernels.  `.fwd` bundles
    what the forward entry points take."""

    def __new__(cls, edges, counts, rank, order, plan, ks=25):
        self = super().__new__(cls, (edges, counts))
        self.edges, self.counts, self.rank, self.order, self.plan = edges, counts, rank, order, plan
        self.ks = int(ks)
        self.fwd = FwdPlan(order, rank, plan, ks) if plan is not None else None
        return self


def edge_list(mask=None, gt=None, mask_stride=0, lap_threshold=20.0, capacity=None, ks=25, order=True, plan=True):
    """Device-side edge list of a batch.

    mask: (B,c1,H,W) float32 or uint8 (channel 0 is used) -- or None with
    gt (B,3,H,W) float32 in [0,1] to generate the reference's Laplacian mask
    on the fly.  Returns (edges (capacity,3) int32 [b,y,x], counts (B+2) int32
    on device: counts[0] = N).  No host synchronisation.  `ks` is the search size the dense/direct
    work split (`.plan`) is built for: pass the k_s the list will be used with.  order=False leaves the
    separate tile-major order out (`.order` None, three launches less) for callers whose kernels all take
    their job order from the plan (sizes with shared-term kernels: (25,9,3), (49,13,3)).  plan=False leaves the
    dense/direct plan out instead (`.plan` None, `.fwd` None: direct kernels in tile order; four launches less).
    """
    L = _lib.lib()
    if mask is not None:
        _need_gpu(mask)
        if mask.dtype == torch.uint8 or mask.dtype == torch.bool:   # edge pixel <=> value 1 (True), like `mask == 1`
            src, kind = mask.contiguous().view(torch.uint8), 1
        else:
            src, kind = _f32c(mask), 0
        B, c1, H, W = src.shape
    else:
        _need_gpu(gt)
        src, kind = _f32c(gt), 2
        B, c1, H, W = src.shape
        if c1 != 3:
            raise ValueError("Laplacian edge mask needs a 3-channel image")
    if capacity is None:
        capacity = B * H * W
    dev = src.device
    edges = torch.empty((max(capacity, 1), 3), dtype=torch.int32, device=dev)
    counts = torch.empty(B + 2, dtype=torch.int32, device=dev)
    rank = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    order = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev) if order else None
    plan = torch.empty(L.ssg_forward_plan_bytes(B, H, W, capacity) // 4, dtype=torch.int32, device=dev) if plan else None
    scratch = torch.empty(L.ssg_edge_scratch_bytes(B, H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):   # launches go to the tensors' GPU, whatever the current device is
        _lib.check(L.ssg_edge_list(_ptr(src), kind, c1, B, H, W, int(mask_stride or 0), float(lap_threshold), int(ks),
                                   _ptr(edges), capacity, _ptr(counts), _ptr(rank), _ptr(order), _ptr(plan),
                                   _ptr(scratch), _stream()))
    return EdgeList(edges, counts, rank, order, plan, ks)


def set_overlap(mode):
    """Stream assignment of the dense-tile and the direct kernel of a pass (k_s <= 25): False / 0 = every launch on the
    caller's stream (per-kernel profiling); True / 1 = dense kernel on the caller's stream, direct kernel on the side
    stream (masks with dense tiles: Laplacian edges); 2 = the other way round (masks without dense tiles: Bernoulli /
    thin strided masks, -15 % at 1 % density; +3 % at C2); 3 (default) = 1 or 2 per pass, from the shape of the last
    plan built on the device (include/ssg_hip.h).  Same results.  Returns the previous mode."""
    return _lib.lib().ssg_set_overlap(int(mode))


def set_dense_threshold(edge_pixels_per_tile):
    """Route 8x32-pixel tiles holding at least this many edge pixels through the shared-term ("dense") forward
    kernel (0 = never; default 16).  Same results either way.  Returns the previous value."""
    return _lib.lib().ssg_set_dense_threshold(int(edge_pixels_per_tile))


def set_tiny_step(on):
    """Small (11,5) fused steps (B*H*W <= 16,384 pixels, capacity <= 4,096 rows: BASELINE's C1) in two launches -- one
    workgroup per edge pixel (ssg_tiny.hip); True by default, False keeps every call on the general path.  Same results
    to rounding.  Returns the previous setting."""
    return bool(_lib.lib().ssg_set_tiny_step(int(bool(on))))


def edge_mask_laplacian(gt, lap_threshold=20.0, mask_stride=0):
    """(B,3,H,W) float32 in [0,1] -> (B,H,W) uint8 {0,1}: generate_mask.py:22-31 on device."""
    _need_gpu(gt)
    g = _f32c(gt)
    B, C, H, W = g.shape
    if C != 3:
        raise ValueError("Laplacian edge mask needs a 3-channel image")
    out = torch.empty((B, H, W), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().ssg_edge_mask_laplacian(_ptr(g), B, H, W, float(lap_threshold), int(mask_stride or 0),
                                                      _ptr(out), _stream()))
    return out


class _SSGMapFn(torch.autograd.Function):
    """SSG rows of a batch for a given edge list (loss_util.py:182-244 + autograd)."""

    @staticmethod
    def forward(ctx, img, edges, counts, n_rows, ks, kw, sigma, eps, generalization, order, fwd, det):
        x = _f32c(img)
        ctx.det = det
        f_order, f_rank, f_plan = fwd if fwd is not None else (order, None, None)
        B, C, H, W = x.shape
        ssg = torch.empty((n_rows, ks * ks), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ssg_map_forward(_ptr(x), None, B, C, H, W, _ptr(edges), _ptr(f_order), _ptr(f_rank),
                                                  _ptr(f_plan), _ptr(counts), n_rows, ks, kw, float(sigma), float(eps),
                                                  int(bool(generalization)), _ptr(ssg), None, None, _stream()))
        ctx.save_for_backward(x, edges, counts, ssg)
        ctx.in_dtype = img.dtype
        ctx.order = order
        ctx.split = (f_rank, f_plan)
        ctx.cfg = (n_rows, ks, kw, float(sigma), int(bool(generalization)))
        return ssg

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_ssg):
        x, edges, counts, ssg = ctx.saved_tensors
        n_rows, ks, kw, sigma, gen = ctx.cfg
        B, C, H, W = x.shape
        g = _f32c(grad_ssg)
        grad = torch.zeros_like(x)
        L = _lib.lib()
        rank, plan = ctx.split
        scratch = None
        if rank is not None and plan is not None:
            scratch = torch.empty(L.ssg_backward_scratch_bytes(n_rows, ks), dtype=torch.uint8, device=x.device)
        fix = _grad_fix(ctx.det, x)
        with torch.cuda.device(x.device):
            _lib.check(L.ssg_map_backward(_ptr(x), B, C, H, W, _ptr(edges), _ptr(ctx.order), _ptr(rank), _ptr(plan),
                                          _ptr(counts), n_rows, ks, kw, sigma, gen, _ptr(ssg), _ptr(g), _ptr(grad),
                                          _ptr(scratch), _ptr(fix), _stream()))
        return grad.to(ctx.in_dtype), None, None, None, None, None, None, None, None, None, None, None


def ssg_map(img, edges, counts, n_rows, ks, kw, sigma, eps=1e-10, generalization=True, order=None, fwd=None,
            deterministic=None):
    """(n_rows, ks*ks) SSG rows of `img` (B,C,H,W) at `edges`; differentiable w.r.t. img.
    `order` (EdgeList.order) is the backward kernel's tile-major job order, `fwd` (EdgeList.fwd)
    the forward's (order, rank map, dense/direct plan)."""
    _need_gpu(img, edges, counts, order)
    check_plan(fwd, ks)
    return _SSGMapFn.apply(img, edges, counts, int(n_rows), int(ks), int(kw), sigma, eps, generalization, order, fwd,
                           deterministic)


class _SSGLossFn(torch.autograd.Function):
    """(l1, kl) of the caller loop realesrganssl_model.py:379-430 over a batch.

    The SSG tensors live only inside forward(): when `sr` needs a gradient, d(l1 + kl)/d sr is produced by the
    SAME launch sequence that produces the two losses (one ssg_loss_backward call) and is the only thing kept
    for backward(), which scales it by the incoming gradient.  That is exact whenever both losses receive the
    same upstream gradient (they are added into one total in every caller of the reference); if autograd hands
    two different tensors, backward() recomputes the step with them (still no host synchronisation)."""

    @staticmethod
    def _run(x, y, edges, counts, n_rows, ks, kw, sigma, eps, gen, w_l1, w_kl, order, fwd, upstream, want_grad,
             det=None):
        L = _lib.lib()
        f_order, f_rank, f_plan = fwd if fwd is not None else (order, None, None)
        B, C, H, W = x.shape
        dev = x.device
        P = ks * ks
        ssg_sr = torch.empty((max(n_rows, 1), P), dtype=torch.float32, device=dev)
        ssg_gt = torch.empty((max(n_rows, 1), P), dtype=torch.float32, device=dev)
        loss = torch.zeros(2, dtype=torch.float32, device=dev)
        grad = torch.zeros_like(x) if want_grad else None
        scratch = torch.empty(L.ssg_loss_scratch_bytes(B, H, W, n_rows, ks), dtype=torch.uint8, device=dev)
        # deferred normalisation (include/ssg_hip.h): the dense-tile forward leaves its rows un-normalised and the
        # backward's row pass rescales them -- only where that pass exists (split backward: plan + supported sizes)
        rsc = None
        if f_rank is not None and f_plan is not None and (ks, kw, C) in ((25, 9, 3), (49, 13, 3)):
            rsc = torch.empty(2 * max(n_rows, 1), dtype=torch.float64, device=dev)
        _lib.check(L.ssg_map_forward(_ptr(x), _ptr(y), B, C, H, W, _ptr(edges), _ptr(f_order), _ptr(f_rank), _ptr(f_plan),
                                     _ptr(counts), n_rows, ks, kw, sigma, eps, gen, _ptr(ssg_sr), _ptr(ssg_gt),
                                     _ptr(rsc), _stream()))
        fix = _grad_fix(det, x) if want_grad else None
        _lib.check(L.ssg_loss_backward(_ptr(x), B, C, H, W, _ptr(edges), _ptr(order), _ptr(f_rank), _ptr(f_plan),
                                       _ptr(counts), n_rows, ks, kw, sigma, gen, _ptr(ssg_sr), _ptr(ssg_gt), w_l1,
                                       w_kl, _ptr(upstream), _ptr(loss), _ptr(grad), _ptr(scratch), _ptr(fix), _ptr(rsc),
                                       1, _stream()))   # (the rows die with this call: no write-back)
        return loss, grad

    @staticmethod
    def forward(ctx, sr, gt, edges, counts, n_rows, ks, kw, sigma, eps, generalization, w_l1, w_kl, order, fwd, det):
        x, y = _f32c(sr), _f32c(gt)
        cfg = (n_rows, ks, kw, float(sigma), float(eps), int(bool(generalization)), float(w_l1), float(w_kl))
        want_grad = bool(ctx.needs_input_grad[0])
        with torch.cuda.device(x.device):
            loss, grad = _SSGLossFn._run(x, y, edges, counts, *cfg, order, fwd, None, want_grad, det)
        ctx.cfg, ctx.order, ctx.fwd, ctx.det = cfg, order, fwd, det
        ctx.in_dtype = sr.dtype
        if want_grad:
            ctx.save_for_backward(x, y, edges, counts, grad)
        return loss[0], loss[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_l1, g_kl):
        x, y, edges, counts, grad = ctx.saved_tensors
        same = (g_l1.data_ptr() == g_kl.data_ptr() and g_l1.numel() == 1 and g_kl.numel() == 1)
        if same:
            out = grad * g_l1.to(torch.float32).reshape(())
        else:
            up = torch.stack([g_l1.to(torch.float32).reshape(()), g_kl.to(torch.float32).reshape(())]).contiguous()
            with torch.cuda.device(x.device):
                _, out = _SSGLossFn._run(x, y, edges, counts, *ctx.cfg, ctx.order, ctx.fwd, up, True, ctx.det)
        return (out.to(ctx.in_dtype),) + (None,) * 14


def ssg_loss(sr, gt, edges, counts, n_rows, ks=25, kw=9, sigma=0.004, eps=1e-10, generalization=True, w_l1=1.0,
             w_kl=1.0, order=None, fwd=None, deterministic=None):
    """Differentiable (l1, kl) for a batch given a device edge list; n_rows bounds N."""
    _need_gpu(sr, gt, edges, counts, order)
    check_plan(fwd, ks)
    return _SSGLossFn.apply(sr, gt, edges, counts, int(n_rows), int(ks), int(kw), sigma, eps, generalization, w_l1,
                            w_kl, order, fwd, deterministic)


class _SSGFusedFn(torch.autograd.Function):
    """(l1, kl) of a batch straight from the mask: ONE C call (ssg_loss_fwd_bwd in its fused form, ssg_sr = ssg_gt =
    NULL) builds the edge list, both SSGs as scratch rows inside the workspace, the criteria and d(l1+kl)/d sr.  Only
    that gradient (and the inputs, for the rare case below) is kept for backward(), which scales it by the incoming
    gradient; if autograd hands two DIFFERENT gradients for l1 and kl the step is redone through _SSGLossFn with them.
    `counts` (B+2 int32, device) receives the edge counts of the call."""

    @staticmethod
    def forward(ctx, sr, gt, mask, counts, cap, ks, kw, sigma, eps, generalization, w_l1, w_kl, mask_stride, lap_threshold,
                det):
        L = _lib.lib()
        x, y = _f32c(sr), _f32c(gt)
        B, C, H, W = x.shape
        if mask is None:
            kind, mc, mp = 2, 3, None
            if C != 3:
                raise ValueError("Laplacian edge mask needs a 3-channel image")
        elif mask.dtype == torch.uint8 or mask.dtype == torch.bool:
            mp = mask.contiguous().view(torch.uint8)
            kind, mc = 1, mp.shape[1]
        else:
            mp = _f32c(mask)
            kind, mc = 0, mp.shape[1]
        want_grad = bool(ctx.needs_input_grad[0])
        dev = x.device
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        grad = torch.empty_like(x) if want_grad else None      # (an OUTPUT of ssg_loss_step: no fill kernel)
        nb = L.ssg_loss_workspace_bytes(B, H, W, cap, ks) + L.ssg_loss_rows_bytes(cap, ks)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        fix = _grad_fix(det, x) if want_grad else None
        with torch.cuda.device(dev):
            _lib.check(L.ssg_loss_step(_ptr(x), _ptr(y), _ptr(mp), kind, mc, B, C, H, W, ks, kw, float(sigma),
                                       float(eps), int(bool(generalization)), float(w_l1), float(w_kl),
                                       int(mask_stride or 0), float(lap_threshold), cap, None, None, _ptr(counts),
                                       _ptr(loss), _ptr(grad), _ptr(ws), nb, _ptr(fix), _stream()))
        ctx.cfg = (cap, ks, kw, sigma, eps, generalization, w_l1, w_kl, mask_stride, lap_threshold, det)
        ctx.in_dtype = sr.dtype
        ctx.has_mask = mask is not None
        if want_grad:
            ctx.save_for_backward(x, y, grad, *((mask,) if mask is not None else ()))
        return loss[0], loss[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_l1, g_kl):
        x, y, grad = ctx.saved_tensors[:3]
        mask = ctx.saved_tensors[3] if ctx.has_mask else None
        same = (g_l1.data_ptr() == g_kl.data_ptr() and g_l1.numel() == 1 and g_kl.numel() == 1)
        if same:
            out = grad * g_l1.to(torch.float32).reshape(())
        else:
            cap, ks, kw, sigma, eps, gen, w_l1, w_kl, stride, thr, det = ctx.cfg
            up = torch.stack([g_l1.to(torch.float32).reshape(()), g_kl.to(torch.float32).reshape(())]).contiguous()
            el = edge_list(mask=mask, gt=y if mask is None else None, mask_stride=stride, lap_threshold=thr,
                           capacity=cap, ks=ks)
            with torch.cuda.device(x.device):
                _, out = _SSGLossFn._run(x, y, el.edges, el.counts, cap, ks, kw, float(sigma), float(eps),
                                         int(bool(gen)), float(w_l1), float(w_kl), el.order, el.fwd, up, True, det)
        return (out.to(ctx.in_dtype),) + (None,) * 14


def ssg_loss_from_mask(sr, gt, mask, counts, capacity, ks=25, kw=9, sigma=0.004, eps=1e-10, generalization=True,
                       w_l1=1.0, w_kl=1.0, mask_stride=0, lap_threshold=20.0, deterministic=None):
    """Differentiable (l1, kl) of a batch from its mask (or from GT's Laplacian, mask=None) in one fused C call."""
    _need_gpu(sr, gt, mask, counts)
    return _SSGFusedFn.apply(sr, gt, mask, counts, int(capacity), int(ks), int(kw), sigma, eps, generalization, w_l1,
                             w_kl, mask_stride, lap_threshold, deterministic)


class LossStep:
    """The whole loss step in ONE C call (ssg_loss_step = ssg_loss_fwd_bwd with the gradient as an output): edge list, SSG(sr), SSG(gt),
    L1 + KL and d(l1+kl)/d sr, with persistent buffers sized for `capacity` edge pixels.

    This is the path bench.py times.  No host synchronisation happens inside; `counts[0]`
    (device) holds the edge-pixel count N of the last step, `loss` the two scalars.

    Memory (k_s = 49): besides the two SSG tensors (2 x capacity x k_s^2 x 4 bytes) a materialising step holds the two
    TILE-MAJOR regions the dense kernels work in (ssg_loss_tm_bytes: about the same again -- +5 GB at capacity 512 x 512),
    the fused step (materialise=False) four row regions in its workspace instead of two.  tile_major=False leaves the
    regions out of a materialising step's workspace: the row-major kernels run (C5: 8.4 instead of 7.3 ms per step).
    A step that finds more edge pixels than `capacity` returns NaN losses (it has used the first `capacity` only).
    The default capacity (every pixel, never overflows) costs the C2 step ~2 % and the C4 step ~5 % against a bound near
    the real count (bench.py: N + 1024) -- tools/r5_capacity_cost.py; the workspace grows with it.

    graph=True records the step's ~17 launches (memset, edge-list builder, two forward variants,
    backward, finalize) into a HIP graph on first use and replays it afterwards: nothing in the
    step depends on host-side values (the edge count stays on the device), so the recording is
    valid for any input CONTENT at the same addresses; a call with other tensors re-records.
    """

    def __init__(self, B, C, H, W, ks=25, kw=9, sigma=0.004, eps=1e-10, generalization=True, w_l1=1.0, w_kl=1.0,
                 mask_stride=0, lap_threshold=20.0, capacity=None, device="cuda", graph=False, deterministic=None,
                 materialise=True, tile_major=True):
        L = _lib.lib()
        self.shape = (B, C, H, W)
        self.cfg = (ks, kw, float(sigma), float(eps), int(bool(generalization)), float(w_l1), float(w_kl),
                    int(mask_stride or 0), float(lap_threshold))
        self.capacity = int(capacity if capacity is not None else B * H * W)
        P = ks * ks
        # materialise=False: the fused step of the C ABI (no SSG output) -- the rows are scratch inside the workspace
        self.materialise = bool(materialise)
        self.ssg_sr = torch.empty((self.capacity, P), dtype=torch.float32, device=device) if materialise else None
        self.ssg_gt = torch.empty((self.capacity, P), dtype=torch.float32, device=device) if materialise else None
        self.counts = torch.zeros(B + 2, dtype=torch.int32, device=device)
        self.loss = torch.zeros(2, dtype=torch.float32, device=device)
        self.grad = torch.zeros((B, C, H, W), dtype=torch.float32, device=device)
        self.ws_bytes = L.ssg_loss_workspace_bytes(B, H, W, self.capacity, ks)
        if not materialise:
            self.ws_bytes += L.ssg_loss_rows_bytes(self.capacity, ks)
        elif tile_major:   # (k_s 49: room for the tile-major regions, which a materialising call then uses as well; 0 otherwise)
            self.ws_bytes += L.ssg_loss_tm_bytes(self.capacity, ks)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.fix = _grad_fix(deterministic, self.grad)   # deterministic mode: fixed-point accumulation buffer
        self.use_graph = bool(graph)
        self._graph, self._graph_key, self._graph_refs = None, None, None

    def edges(self):
        """View of the workspace's edge list (capacity,3) int32."""
        return self.ws[: self.capacity * 12].view(torch.int32).view(self.capacity, 3)

    def __call__(self, sr, gt, mask=None):
        """mask (B,c1,H,W) float32/uint8, or None -> Laplacian mask of gt generated on device."""
        _need_gpu(sr, gt, mask)
        B, C, H, W = self.shape
        ks, kw, sigma, eps, gen, w_l1, w_kl, stride, thr = self.cfg
        assert tuple(sr.shape) == self.shape and tuple(gt.shape) == self.shape
        assert sr.dtype == torch.float32 and gt.dtype == torch.float32 and sr.is_contiguous() and gt.is_contiguous()
        if mask is None:
            kind, mc, mp = 2, 3, None
        elif mask.dtype == torch.uint8 or mask.dtype == torch.bool:
            kind, mc, mp = 1, mask.shape[1], mask.view(torch.uint8)
        else:
            kind, mc, mp = 0, mask.shape[1], mask
            assert mask.dtype == torch.float32
        if mp is not None:
            assert mp.is_contiguous()

        def launch():
            # (ssg_loss_step: the gradient is an output of the call -- no fill kernel in front of it)
            with torch.cuda.device(self.grad.device):
                _lib.check(_lib.lib().ssg_loss_step(_ptr(sr), _ptr(gt), _ptr(mp), kind, mc, B, C, H, W, ks, kw,
                                                       sigma, eps, gen, w_l1, w_kl, stride, thr, self.capacity,
                                                       _ptr(self.ssg_sr), _ptr(self.ssg_gt), _ptr(self.counts),
                                                       _ptr(self.loss), _ptr(self.grad), _ptr(self.ws), self.ws_bytes,
                                                       _ptr(self.fix), _stream()))

        if not self.use_graph:
            launch()
            return self.loss, self.grad
        key = (_ptr(sr), _ptr(gt), _ptr(mp), kind, mc)
        if self._graph is None or key != self._graph_key:
            launch()                      # eager once: function attributes are set outside the recording
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                launch()
            self._graph, self._graph_key, self._graph_refs = g, key, (sr, gt, mp)   # keep the recorded buffers alive
        self._graph.replay()
        return self.loss, self.grad
