"""The two criteria the reference applies to (SSG_sr, SSG_gt) and their fusion.

L1Loss / KLDistanceLoss mirror GAN-Based-SR/basicsr/losses/basic_loss.py:41-66
and :269-282 (same constructor arguments and semantics) for code that keeps
the reference's per-image `similarity_map` loop.  Given deferred SSG handles
(the default, losses/lazy.py) their torch calls are intercepted and the whole
loop runs as one batched step; given materialised fp32 GPU tensors they run the
engine's streaming criteria kernels (ssg_criteria_sums / ssg_criteria_grad: one
pass each way instead of 6-11 element-wise torch kernels); anything else (CPU
tensors, other dtypes, element-wise weights, reduction 'none', a target that
wants a gradient) takes the reference's torch expressions.  SSGLoss is the batched
replacement of the whole caller block realesrganssl_model.py:379-430 /
ddpmssl.py:438-513: one module call per step, no Python loop over images, no
host synchronisation, masks of 1 or 3 channels, optional mask_stride and
on-device Laplacian mask.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine

_reduction_modes = ['none', 'mean', 'sum']


_NATIVE_CRITERIA = True


def set_native_criteria(on):
    """Route L1Loss / KLDistanceLoss on materialised fp32 GPU tensors through the engine's streaming criteria kernels
    (default) or through the reference's torch expressions (needed for a second derivative through the criterion:
    the native autograd node is once-differentiable).  Returns the previous setting."""
    global _NATIVE_CRITERIA
    prev, _NATIVE_CRITERIA = _NATIVE_CRITERIA, bool(on)
    return prev


def _native_pair(pred, target):
    """Whether (pred, target) can go through the engine's streaming criteria kernels (ssg_criteria_sums / _grad): real
    fp32 tensors of one shape on ONE GPU, no gradient wanted for the target.  (The native node is once-differentiable:
    a caller that needs a second derivative through the criterion -- gradient penalties -- passes tensors under
    `torch.autograd.graph`'s usual rules and gets an error from autograd, not a wrong value; `set_native_criteria(False)`
    restores the reference's torch expressions.)  Deferred handles (losses/lazy.py) are not
    tensors: they take the torch calls below, which is where they are intercepted."""
    return (_NATIVE_CRITERIA and isinstance(pred, torch.Tensor) and isinstance(target, torch.Tensor)
            and pred.is_cuda and target.is_cuda and pred.device == target.device
            and pred.dtype == torch.float32 and target.dtype == torch.float32 and pred.shape == target.shape
            and pred.numel() > 0 and not target.requires_grad)


class _CriterionSum(torch.autograd.Function):
    """sum |pred - target| (which = 0) or sum t'(log t' - log s') (which = 1) of two fp32 GPU tensors in one streaming
    pass (fp64 accumulation, fixed order); backward = one pass writing the gradient, the incoming gradient read on the
    device.  Replaces the 3 + 3 (L1) / 5 + 6 (KL) element-wise torch kernels of the reference's criteria."""

    @staticmethod
    def forward(ctx, pred, target, which):
        from .. import _lib
        L = _lib.lib()
        a, b = pred.contiguous(), target.detach().contiguous()
        sums = torch.empty(2, dtype=torch.float32, device=a.device)
        scratch = torch.empty(L.ssg_criteria_scratch_bytes(), dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            _lib.check(L.ssg_criteria_sums(a.data_ptr(), b.data_ptr(), a.numel(), scratch.data_ptr(), sums.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(a, b)
        ctx.which = which
        return sums[which]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import _lib
        a, b = ctx.saved_tensors
        coef = torch.zeros(2, dtype=torch.float32, device=a.device)
        coef[ctx.which] = g.to(torch.float32).reshape(())
        grad = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.check(_lib.lib().ssg_criteria_grad(a.data_ptr(), b.data_ptr(), a.numel(), coef.data_ptr(), grad.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream))
        return grad, None, None


class L1Loss(nn.Module):
    """loss_weight * L1(pred, target) with 'none' | 'mean' | 'sum' reduction and an
    optional element-wise weight (basic_loss.py:41-66, loss_util.py:33-62)."""

    def __init__(self, loss_weight=1.0, reduction='mean'):
        super(L1Loss, self).__init__()
        if reduction not in _reduction_modes:
            raise ValueError(f'Unsupported reduction mode: {reduction}. Supported ones are: {_reduction_modes}')
        self.loss_weight = loss_weight
        self.reduction = reduction

    def forward(self, pred, target, weight=None, **kwargs):
        if weight is None and self.reduction in ('mean', 'sum') and _native_pair(pred, target):
            total = _CriterionSum.apply(pred, target, 0)
            return self.loss_weight * (total / pred.numel() if self.reduction == 'mean' else total)
        loss = F.l1_loss(pred, target, reduction='none')
        if weight is not None:
            loss = loss * weight
        if self.reduction == 'sum':
            loss = loss.sum()
        elif self.reduction == 'mean':
            if weight is None:
                loss = loss.mean()
            else:  # weight_reduce_loss: mean over the weighted region
                w = weight.sum() if weight.size(1) > 1 else weight.sum() * loss.size(1)
                loss = loss.sum() / w
        return self.loss_weight * loss


class KLDistanceLoss(nn.Module):
    """loss_weight * F.kl_div(log(clamp(x,1e-10)), clamp(y,1e-10)) (basic_loss.py:269-282)."""

    def __init__(self, loss_weight=0.1, reduction='mean', softmax=False):
        super(KLDistanceLoss, self).__init__()
        self.loss_weight = loss_weight
        self.reduction = reduction
        self.softmax = softmax

    def forward(self, x, y):
        if self.softmax:
            x = x.softmax(dim=-1)
            y = y.softmax(dim=-1)
        if self.reduction in ('mean', 'sum', 'batchmean') and _native_pair(x, y):
            total = _CriterionSum.apply(x, y, 1)
            div = x.numel() if self.reduction == 'mean' else (x.shape[0] if self.reduction == 'batchmean' else 1)
            return self.loss_weight * (total / div)
        return self.loss_weight * F.kl_div(torch.clamp(input=x, min=1e-10).log(), torch.clamp(input=y, min=1e-10),
                                           reduction=self.reduction)


class SSGLoss(nn.Module):
    """Batched Self-Similarity-Graph loss: returns (l_selfsim, l_selfsim_kl).

    forward(sr, gt, mask=None): sr, gt (B,C,H,W) on the GPU; mask (B,1|3,H,W)
    float {0,1} / uint8 {0,1} / bool, or None to generate the reference's offline Laplacian
    edge mask of `gt` on the device (generate_mask.py:22-31).  Semantics of the
    reference loop: images whose mask is empty are skipped, the means run over
    sum_i N_i * k_s^2 elements of the LOCAL batch, both terms are 0 when every
    mask is empty.

    Memory: one call holds 2 * capacity * k_s^2 * 4 bytes of SSG rows (+ the same again / 2 of
    backward scratch) while it runs -- 0.5 GB per 100 k edge pixels at k_s = 25 -- and keeps only
    the (B,C,H,W) gradient for backward.  At k_s = 49 the workspace holds FOUR row regions (two
    row-major, two tile-major: ssg_loss_rows_bytes in include/ssg_hip.h): 4 * capacity * 2401 * 4
    bytes = 10 GB at capacity 512 x 512; size `capacity` accordingly for dense masks at that size.  `capacity` bounds the number of edge pixels of a call
    without a host round trip.  Default (capacity=None): B*H*W / (4 * max(1, mask_stride)) -- a quarter of the call's pixels
    (edge masks are ~7 % dense), and the stride pattern keeps 1 / stride of those; computed per call, so a small first
    batch does not pin it -- or the largest count seen so far plus 1/8, whichever is larger.  The default therefore
    overflows on the FIRST call for any mask that is more than 25 % dense before striding: pass `capacity=B*H*W //
    max(1, mask_stride)` for dense strided masks and `capacity=B*H*W` for the 100 % stress mask -- or rely on the checks
    below (one host synchronisation plus a second full step for the first call, and again after each growth):
      * the first `sync_checks` calls (default 2), and the call after any overflow, read the edge
        count back in the same step (one host synchronisation of a 4-byte copy); a call found
        truncated is RECOMPUTED at the grown capacity before forward() returns, so its losses
        and gradient cover every edge pixel (on_overflow='grow', default, with a warning) or a
        RuntimeError is raised (on_overflow='raise');
      * later calls copy their count to pinned host memory asynchronously; it is looked at one or
        more calls later, so nothing stalls.  A call that overflowed then HAS used only the first
        `capacity` edge pixels in batch order: its two losses are NaN (set on the device by the step
        itself: a truncated step never passes for a complete one), every such call is reported
        (warning / RuntimeError), the capacity grows, and the next call is checked synchronously.
        `flush()` waits for the outstanding counts (call it after the last / a single forward, e.g.
        in validation); switching the module to eval() and its deletion do so too.

    deterministic=True makes the gradient bit-reproducible from run to run (fixed-point integer
    accumulation instead of fp32 atomics; include/ssg_hip.h `ssg_grad_fix_bytes`).
    """

    def __init__(self, kernel_size_search=25, kernel_size_window=9, sigma=0.004, generalization=True,
                 loss_weight_l1=1e3, loss_weight_kl=1e3, mask_stride=0, eps=1e-10, lap_threshold=20.0,
                 capacity=None, on_overflow='grow', deterministic=None, sync_checks=2):
        super().__init__()
        if on_overflow not in ('grow', 'raise'):
            raise ValueError(f"on_overflow must be 'grow' or 'raise', got {on_overflow!r}")
        self.ks, self.kw = kernel_size_search, kernel_size_window
        self.sigma, self.generalization, self.eps = sigma, generalization, eps
        self.w_l1, self.w_kl = loss_weight_l1, loss_weight_kl
        self.mask_stride, self.lap_threshold = mask_stride, lap_threshold
        self.capacity = capacity   # the caller's bound (None: per-call default); never overwritten by a per-call clamp
        self._grown = 0            # capacity learnt from overflows
        self.on_overflow = on_overflow
        self.deterministic = deterministic   # None: SSG_DETERMINISTIC env; True: bit-reproducible gradients
        self._sync_left = int(sync_checks)   # calls still to be checked in the same step
        self._pending = []         # (event, pinned count, capacity used) of earlier calls, oldest first
        self._free = {}            # device index -> (pinned count, event) pairs whose count has been read

    def _capacity_for(self, B, H, W):
        # (a strided mask keeps 1 / stride of the pixels: the default bound shrinks with it -- the direct kernels' grids and
        #  the workspace follow the bound, a 10 x generous one cost the C4 step 8 %)
        cap = self.capacity if self.capacity is not None else max(1024, (B * H * W) // (4 * max(1, int(self.mask_stride or 0))))
        return min(max(cap, self._grown), B * H * W)   # (clamped for THIS call only)

    def _grow(self, n, cap):
        self._grown = max(self._grown, 2 * cap, n + n // 8)
        self._sync_left = max(self._sync_left, 1)   # verify the next call in its own step

    def _report(self, n, cap, recomputed):
        msg = (f"SSGLoss: a step had {n} edge pixels but capacity {cap}; "
               + ("it was recomputed at the grown capacity. " if recomputed else f"it used the first {cap} only. ")
               + f"capacity is now {self._grown} (dense masks: pass capacity=B*H*W).")
        if self.on_overflow == 'raise':
            raise RuntimeError(msg)
        import warnings
        warnings.warn(msg)

    def _check_previous(self, wait=False):
        """Look at the edge counts of earlier calls whose copies have landed (never blocks unless wait); every
        overflowed call is reported."""
        first_error = None
        while self._pending:
            ev, host, cap = self._pending[0][:3]
            if not (wait or ev.query()):
                break
            if wait:
                ev.synchronize()
            done = self._pending.pop(0)
            n = int(host[0])
            if len(done) > 3:
                self._free.setdefault(done[3], []).append((host, ev))
            if n > cap:
                self._grow(n, cap)
                try:
                    self._report(n, cap, recomputed=False)
                except RuntimeError as e:      # look at the remaining counts before raising
                    first_error = first_error or e
        if first_error is not None:
            raise first_error

    def flush(self):
        """Wait for the outstanding edge counts and report any overflow (after the last or a single forward)."""
        self._check_previous(wait=True)

    def train(self, mode=True):
        if not mode and self._pending:      # leaving training: nothing may stay unreported
            self.flush()
        return super().train(mode)

    def __del__(self):
        try:
            if self._pending:
                self.flush()
        except Exception:                   # (interpreter shutdown, 'raise' mode inside a finaliser)
            pass

    def _run(self, sr, gt, mask, cap):
        B = sr.shape[0]
        counts = torch.empty(B + 2, dtype=torch.int32, device=sr.device)
        out = engine.ssg_loss_from_mask(sr, gt.detach(), mask, counts, cap, self.ks, self.kw, self.sigma, self.eps,
                                        self.generalization, self.w_l1, self.w_kl, self.mask_stride, self.lap_threshold,
                                        self.deterministic)
        return out, counts

    def forward(self, sr, gt, mask=None):
        B, C, H, W = sr.shape
        self._check_previous()
        cap = self._capacity_for(B, H, W)
        out, counts = self._run(sr, gt, mask, cap)
        if self._sync_left > 0:
            # same-step check: a truncated call never leaves forward() unnoticed
            self._sync_left -= 1
            n = int(counts[0])           # (host synchronisation)
            if n > cap:
                self._grow(n, cap)
                if self.on_overflow == 'raise':
                    self._report(n, cap, recomputed=False)
                out, counts = self._run(sr, gt, mask, self._capacity_for(B, H, W))
                self._report(n, cap, recomputed=True)
        else:
            with torch.cuda.device(sr.device):   # the copy and the event go to the stream of sr's device
                # (pinned words and events are recycled once their count has been read: no page-locking per call)
                # (one page-locked block of 64 words per device, cut into (word, event) pairs at first use: page-locking
                #  a fresh word per call cost ~1 ms of host time each until enough of them had come back -- the first
                #  dozens of steps of a loop that runs ahead of the GPU.  With all 64 in flight the oldest is waited for.)
                pool = self._free.get(sr.device.index)
                if pool is None:
                    block = torch.zeros(64, dtype=torch.int32).pin_memory()
                    pool = self._free[sr.device.index] = [(block[i:i + 1], torch.cuda.Event()) for i in range(64)]
                while not pool and self._pending:
                    self._pending[0][0].synchronize()
                    self._check_previous()
                host, ev = pool.pop()
                host.copy_(counts[:1], non_blocking=True)
                ev.record(torch.cuda.current_stream(sr.device))
            self._pending.append((ev, host, cap, sr.device.index))
        self.last_counts = counts
        return out
