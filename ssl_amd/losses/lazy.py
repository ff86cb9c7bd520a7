"""Deferred SSG rows: how the reference's UNCHANGED per-image caller loop reaches the batched engine.

The loop every training model of the reference runs (realesrganssl_model.py:379-430, ddpmssl.py:438-513, KAIR
model_ssl.py:285-340; restated in ssl_amd/reference_loop.py) builds `similarity_map(img_i, mask_i, ...)` twice per
image, `torch.cat`s the results and hands them to `L1Loss` / `KLDistanceLoss`.  Run eagerly that is 2 b latency-bound
single-image launches, each behind a host synchronisation (the row count N_i shapes the tensor), then 15-20
element-wise torch kernels over the concatenated (1, sum N, k_s^2) tensors, then b single-image backward passes: 5-8x
the batched step at BASELINE's C2 (bench.py `extra.ref_api`, eager lines).

With `SSG_LAZY` on (the default; `set_lazy(False)` / SSG_LAZY=0 restore eager tensors) `similarity_map(...).getitem()`
returns a `LazySSG` handle instead: (image, mask, kernel sizes), no launch, no synchronisation.  The handle is a
tensor-LIKE in torch's `__torch_function__` protocol, so the loop's own calls reach it unchanged:

    torch.cat([h_0, h_1, ...], dim=1)                        -> a LazySSG over all the images
    F.l1_loss(pred, target, reduction='none') ... .mean()    -> the fused step's L1 mean   (basic_loss.py:14-16,41-66)
    torch.clamp(input=x, min=1e-10).log(), torch.clamp(input=y, min=1e-10), F.kl_div(..., reduction='mean')
                                                             -> the fused step's KL mean   (basic_loss.py:269-282)

Both criteria of one (pred, target) pair are the two outputs of ONE autograd node: its forward builds the batch's edge
list, SSG(sr), SSG(gt) and the criteria sums in one pass over all images (ssg_edge_list, ssg_map_forward,
ssg_loss_backward without a gradient), its backward receives autograd's two incoming gradients on the device
(`upstream`: no host round trip) and runs the split backward once.  It works with the reference's OWN criterion
modules as well as with ssl_amd's mirrors -- both only call the torch functions above.

Anything else done to a handle (`.shape`, indexing, arithmetic, any other torch function, pairs whose settings or
image shapes differ, a `weight=` tensor, `softmax=True`, GT images that require a gradient) MATERIALISES it: the rows
are computed eagerly per image exactly as without SSG_LAZY and the call proceeds on real tensors -- same values, the
reference's eager cost.  A pair whose two masks differ (the reference would fail on the row counts or silently
broadcast) yields NaN losses.
"""
import os
import warnings

import torch
import torch.nn.functional as F

from .. import _lib, engine

_ptr, _stream, _f32c = engine._ptr, engine._stream, engine._f32c

_LAZY = os.environ.get("SSG_LAZY", "1") not in ("", "0")
# The eager-cliff warning (LazySSG._compute): rows computed image by image cost several times the batched step.  It is
# raised PER HANDLE -- when a handle that torch.cat made of >= 2 per-image parts materialises, or when a second distinct
# handle materialises before a batched step has run -- and rate-limited PER CALL SITE (the first frame outside this
# package), so an intended eager use (eval code reading .shape, 3-channel masks that differ) warns once where it
# happens and a later regression somewhere else in the program still gets reported.
_materialised_since_step = set()   # ids of the distinct handles materialised since the last batched step
_warned_sites = {}                 # (file, line) -> number of warnings suppressed there
_WARN_PER_SITE = 1


def _call_site():
    import sys
    f = sys._getframe(2)
    here = os.path.dirname(os.path.abspath(__file__))
    while f is not None and os.path.dirname(os.path.abspath(f.f_code.co_filename)).startswith(os.path.dirname(here)):
        f = f.f_back
    return (f.f_code.co_filename, f.f_lineno) if f is not None else ("?", 0)


def set_lazy(on):
    """Switch deferred SSG handles on / off process-wide; returns the previous setting."""
    global _LAZY
    prev, _LAZY = _LAZY, bool(on)
    return prev


def lazy_enabled():
    return _LAZY


def _materialised(a):
    if isinstance(a, _Lazy):
        return a.materialise()
    if isinstance(a, (list, tuple)):
        return type(a)(_materialised(x) for x in a)
    if isinstance(a, dict):
        return {k: _materialised(v) for k, v in a.items()}
    return a


class _Lazy:
    """A value that is computed when something the fused path does not know asks for it."""

    _t = None

    def _compute(self):
        raise NotImplementedError

    def materialise(self):
        if self._t is None:
            self._t = self._compute()
        return self._t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        h = _HANDLERS.get(func)
        if h is not None:
            r = h(*args, **kwargs)
            if r is not NotImplemented:
                return r
        return func(*_materialised(args), **_materialised(kwargs))

    # everything a caller may do to the tensor it expects: done to the materialised tensor
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialise(), name)

    def __repr__(self):
        return f"{type(self).__name__}({'materialised' if self._t is not None else 'deferred'})"

    # dim 0 of everything deferred here -- SSG rows (1, sum N, k_s^2) and their element-wise images -- is 1: the
    # reference's caller loop asks `len(b_sr_list) > 0` of the concatenated tensor before each criterion
    # (realesrganssl_model.py:413,419); answering must not compute anything
    _len0 = None

    def __len__(self):
        return self._len0 if self._len0 is not None and self._t is None else len(self.materialise())

    def __bool__(self):
        return bool(self.materialise())

    def __float__(self):
        return float(self.materialise())

    def __int__(self):
        return int(self.materialise())

    __hash__ = object.__hash__       # (__eq__ below is the tensor's element-wise comparison; handles are keyed by identity)

    def __getitem__(self, k):
        return self.materialise()[k]

    def __iter__(self):
        return iter(self.materialise())

    def __neg__(self):
        return -self.materialise()

    def __abs__(self):
        return abs(self.materialise())


def _binary(name):
    def op(self, other):
        return getattr(self.materialise(), name)(_materialised(other))
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "pow", "matmul", "floordiv", "mod"):
    setattr(_Lazy, f"__{_n}__", _binary(f"__{_n}__"))
    setattr(_Lazy, f"__r{_n}__", _binary(f"__r{_n}__"))
for _n in ("lt", "le", "gt", "ge", "eq", "ne"):
    setattr(_Lazy, f"__{_n}__", _binary(f"__{_n}__"))


class LazySSG(_Lazy):
    """SSG rows (1, sum_i N_i, k_s^2) of one or several images, not computed yet.

    parts: [(img (1,C,H,W), mask (1,c1,H,W), conv)] in row order; conv = 'ch0' (ssl_cuda / ssl_hip: channel 0 of the
    mask decides, loss_util.py:233) or 'all' (ssl_pytorch: torch.where over every mask channel, loss_util.py:195-198).
    cfg: (k_s, k_w, sigma, eps, generalization)."""

    _len0 = 1

    def __init__(self, parts, cfg, from_cat=False):
        self.parts = list(parts)
        self.cfg = tuple(cfg)
        self.from_cat = bool(from_cat)   # made by torch.cat of per-image handles (the reference loop's batch)
        self._pairs = {}     # id(target handle) -> (target handle, (l1 mean, kl mean)) of the fused step

    def _compute(self):
        from .loss_util import eager_rows
        _materialised_since_step.add(id(self))
        if (self.from_cat and len(self.parts) >= 2) or len(_materialised_since_step) >= 2:
            # a batch of per-image handles (or a second handle of a loop) is used in a way the batched step does not
            # cover: its rows are computed one image at a time (C2: 7 ms instead of 2.9 per loop)
            site = _call_site()
            seen = _warned_sites.get(site, 0)
            _warned_sites[site] = seen + 1
            if seen < _WARN_PER_SITE:
                warnings.warn_explicit("ssl_amd: deferred SSG rows of %d image(s) are being computed eagerly, one image at a "
                              "time (the handle was used in a way other than torch.cat / L1Loss / KLDistanceLoss on "
                              "equal settings) -- same values, several times the batched step's cost.  "
                              "set_lazy(False) / SSG_LAZY=0 returns plain tensors everywhere and silences this."
                              % len(self.parts), RuntimeWarning, site[0], site[1])
        rows = [eager_rows(img, mask, conv, *self.cfg) for img, mask, conv in self.parts]
        return rows[0] if len(rows) == 1 else torch.cat(rows, dim=1)

    # the two methods the criteria call on an SSG tensor
    def clamp(self, min=None, max=None):
        return _h_clamp(self, min, max)

    def abs(self):
        return self.materialise().abs()


class _L1Elem(_Lazy):
    """F.l1_loss(pred, target, reduction='none') of two handles: only ever reduced."""
    _len0 = 1

    def __init__(self, pred, target):
        self.pred, self.target = pred, target

    def _compute(self):
        return F.l1_loss(self.pred.materialise(), self.target.materialise(), reduction='none')

    def mean(self, *a, **k):
        if not a and not k:
            r = fused_losses(self.pred, self.target)
            if r is not None:
                return r[0]
        return self.materialise().mean(*a, **k)

    def sum(self, *a, **k):
        if not a and not k:
            r = fused_losses(self.pred, self.target)
            if r is not None:
                return r[0] * r[2]           # mean x element count (device scalar)
        return self.materialise().sum(*a, **k)


class _Clamped(_Lazy):
    """torch.clamp(input=handle, min=m), optionally .log() of it -- the operands of the reference's KL criterion."""
    _len0 = 1

    def __init__(self, src, lo, logged=False):
        self.src, self.lo, self.logged = src, lo, logged

    def _compute(self):
        t = torch.clamp(input=self.src.materialise(), min=self.lo)
        return t.log() if self.logged else t

    def log(self):
        if self.logged:
            return self.materialise().log()
        return _Clamped(self.src, self.lo, True)


_KL_CLAMP = 1e-10     # basic_loss.py:281; the row kernels' constant (ssg_grow.hip)


def _h_cat(tensors, dim=0, **kw):
    if kw or dim != 1 or not len(tensors) or not all(isinstance(t, LazySSG) and t._t is None for t in tensors):
        return NotImplemented
    if any(t.cfg != tensors[0].cfg for t in tensors):
        return NotImplemented
    return LazySSG([p for t in tensors for p in t.parts], tensors[0].cfg, from_cat=True)


def _h_l1(input, target, size_average=None, reduce=None, reduction='mean', **kw):
    if kw or size_average is not None or reduce is not None:
        return NotImplemented
    if not (isinstance(input, LazySSG) and isinstance(target, LazySSG)):
        return NotImplemented
    e = _L1Elem(input, target)
    return e if reduction == 'none' else e.mean() if reduction == 'mean' else e.sum() if reduction == 'sum' else NotImplemented


def _h_clamp(input, min=None, max=None, **kw):
    if kw or max is not None or not isinstance(input, LazySSG) or isinstance(min, torch.Tensor) or min is None:
        return NotImplemented
    return _Clamped(input, float(min))


def _h_log(input, **kw):
    if kw or not isinstance(input, _Clamped) or input.logged:
        return NotImplemented
    return input.log()


def _h_kl(input, target, size_average=None, reduce=None, reduction='mean', log_target=False, **kw):
    if kw or size_average is not None or reduce is not None or log_target or reduction != 'mean':
        return NotImplemented
    if not (isinstance(input, _Clamped) and input.logged and isinstance(target, _Clamped) and not target.logged):
        return NotImplemented
    if input.lo != _KL_CLAMP or target.lo != _KL_CLAMP:
        return NotImplemented
    r = fused_losses(input.src, target.src)
    return NotImplemented if r is None else r[1]


def _h_mean(input, *a, **k):
    return input.mean(*a, **k) if isinstance(input, _L1Elem) else NotImplemented


def _h_sum(input, *a, **k):
    return input.sum(*a, **k) if isinstance(input, _L1Elem) else NotImplemented


_HANDLERS = {torch.cat: _h_cat, torch.concat: _h_cat, F.l1_loss: _h_l1, torch.clamp: _h_clamp, torch.clip: _h_clamp,
             torch.log: _h_log, F.kl_div: _h_kl, torch.mean: _h_mean, torch.sum: _h_sum,
             torch.Tensor.mean: _h_mean, torch.Tensor.sum: _h_sum, torch.Tensor.log: _h_log,
             torch.Tensor.clamp: _h_clamp}


class _LazyStepFn(torch.autograd.Function):
    """(l1 mean, kl mean) of a batch as ONE autograd node; the SSG rows live between forward and backward."""

    @staticmethod
    def forward(ctx, x_in, y, mask, ks, kw, sigma, eps, gen, det):
        L = _lib.lib()
        x = _f32c(x_in)
        B, C, H, W = x.shape
        dev = x.device
        with torch.cuda.device(dev):
            dense = (ks, kw, C) in ((25, 9, 3), (49, 13, 3))      # sizes whose kernels all work from the plan
            el = engine.edge_list(mask=mask, capacity=B * H * W, ks=ks, order=not dense)
            n = int(el.counts[0])        # ONE host synchronisation per step (the loop itself has one per image)
            loss = torch.zeros(2, dtype=torch.float32, device=dev)
            ctx.n = n
            ctx.in_dtype = x_in.dtype
            if n == 0:
                return loss[0], loss[1], torch.zeros((), dtype=torch.float32, device=dev)
            P = ks * ks
            ssg_sr = torch.empty((n, P), dtype=torch.float32, device=dev)
            ssg_gt = torch.empty((n, P), dtype=torch.float32, device=dev)
            order, rank, plan = el.fwd
            rsc = None
            if dense:
                rsc = torch.empty(2 * n, dtype=torch.float64, device=dev)   # deferred normalisation (ssg_hip.h)
            else:
                rank = plan = None
            _lib.check(L.ssg_map_forward(_ptr(x), _ptr(y), B, C, H, W, _ptr(el.edges), _ptr(order), _ptr(rank),
                                         _ptr(plan), _ptr(el.counts), n, ks, kw, sigma, eps, gen, _ptr(ssg_sr),
                                         _ptr(ssg_gt), _ptr(rsc), _stream()))
            scratch = torch.empty(L.ssg_loss_scratch_bytes(B, H, W, n, ks), dtype=torch.uint8, device=dev)
            _lib.check(L.ssg_loss_backward(_ptr(x), B, C, H, W, _ptr(el.edges), _ptr(order), _ptr(rank), _ptr(plan),
                                           _ptr(el.counts), n, ks, kw, sigma, gen, _ptr(ssg_sr), _ptr(ssg_gt), 1.0,
                                           1.0, None, _ptr(loss), None, _ptr(scratch), None, _ptr(rsc), 1, _stream()))
        count = torch.full((), float(n * P), dtype=torch.float32, device=dev)
        ctx.mark_non_differentiable(count)
        ctx.save_for_backward(x, el.edges, el.counts, ssg_sr, ssg_gt)
        ctx.keep = (order, rank, plan, rsc)
        ctx.cfg = (ks, kw, sigma, gen, det)
        return loss[0], loss[1], count

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_l1, g_kl, _g_count):
        if ctx.n == 0:
            return (None,) * 9
        L = _lib.lib()
        x, edges, counts, ssg_sr, ssg_gt = ctx.saved_tensors
        order, rank, plan, rsc = ctx.keep
        ks, kw, sigma, gen, det = ctx.cfg
        B, C, H, W = x.shape
        dev, n = x.device, ctx.n
        with torch.cuda.device(dev):
            up = torch.stack([g_l1.to(torch.float32).reshape(()), g_kl.to(torch.float32).reshape(())]).contiguous()
            grad = torch.zeros_like(x)
            loss = torch.empty(2, dtype=torch.float32, device=dev)
            scratch = torch.empty(L.ssg_loss_scratch_bytes(B, H, W, n, ks), dtype=torch.uint8, device=dev)
            fix = engine._grad_fix(det, x)
            _lib.check(L.ssg_loss_backward(_ptr(x), B, C, H, W, _ptr(edges), _ptr(order), _ptr(rank), _ptr(plan),
                                           _ptr(counts), n, ks, kw, sigma, gen, _ptr(ssg_sr), _ptr(ssg_gt), 1.0, 1.0,
                                           _ptr(up), _ptr(loss), _ptr(grad), _ptr(scratch), _ptr(fix), _ptr(rsc), 1,
                                           _stream()))
        return (grad.to(ctx.in_dtype),) + (None,) * 8


def _on_gpu(t):
    return t.is_cuda      # (the engine has no CPU path; the CPU test-suite swaps this and _LazyStepFn for stand-ins)


def fused_losses(pred, target, deterministic=None):
    """(l1 mean, kl mean, element count) of a (pred, target) pair of handles through one batched step, or None when
    the pair does not qualify (the caller then materialises).  Cached on the pair: L1Loss and KLDistanceLoss of the
    same step share one evaluation and one autograd node."""
    if not (isinstance(pred, LazySSG) and isinstance(target, LazySSG)):
        return None
    hit = pred._pairs.get(id(target))
    if hit is not None and hit[0] is target:
        return hit[1]
    if pred._t is not None or target._t is not None or pred.cfg != target.cfg or len(pred.parts) != len(target.parts):
        return None
    img0, mask0, conv0 = pred.parts[0]
    if not _on_gpu(img0) or img0.dim() != 4 or not img0.is_floating_point():
        return None
    for (a, ma, ca), (b, mb, cb) in zip(pred.parts, target.parts):
        if (a.shape != img0.shape or b.shape != img0.shape or a.shape[0] != 1 or ma.shape != mask0.shape
                or mb.shape != mask0.shape or ca != conv0 or cb != conv0 or b.requires_grad
                or a.device != img0.device or b.device != img0.device or ma.dtype != mask0.dtype or mb.dtype != mask0.dtype):
            return None
    ks, kw, sigma, eps, gen = pred.cfg
    x = torch.cat([p[0] for p in pred.parts], 0)            # (autograd's cat: its backward hands every image its slice)
    y = torch.cat([p[0] for p in target.parts], 0).detach()
    m = torch.cat([p[1] for p in pred.parts], 0)
    m2 = torch.cat([p[1] for p in target.parts], 0)
    bad = (m != m2).any()
    if conv0 == 'all' and m.shape[1] > 1:
        # ssl_pytorch lists the rows once per mask channel: with equal channels (the pair pool's masks,
        # realesrganssl_model.py:339-341) every row is repeated c1 times, which leaves both means where they are
        if not bool((m == m[:, :1]).all()):        # (host synchronisation; channels that differ: eager rows)
            return None
    mult = m.shape[1] if conv0 == 'all' else 1
    l1u, klu, count = _LazyStepFn.apply(x, _f32c(y), m[:, :1].contiguous(), int(ks), int(kw), float(sigma), float(eps),
                                        int(bool(gen)), deterministic)
    nan = torch.full((), float("nan"), dtype=l1u.dtype, device=l1u.device)
    out = (torch.where(bad, nan, l1u), torch.where(bad, nan, klu), count * mult)
    pred._pairs[id(target)] = (target, out)
    _materialised_since_step.clear()
    return out
