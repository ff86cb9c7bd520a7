"""`compute_similarity` -- the reference's native operator, on MI355X.

Mirror of GAN-Based-SR/basicsr/losses/similarity/similaritywrapper.py:59-69
(same name, arguments and result: raw squared patch distances (N, psize,
psize), differentiable w.r.t. `image`), backed by ssg_compute_similarity /
ssg_compute_similarity_backward of include/ssg_hip.h instead of the JIT-built
CUDA extension.  Differences by design: tensors that are not on the GPU raise
RuntimeError (the reference logs and calls sys.exit(), :60-62); nothing is
compiled at import; kernels run on torch's current stream and the two
torch.cuda.synchronize() calls of the reference backward (:50,53) are gone.
"""
import torch
import torch.nn.functional as F

from ... import _lib


class _DistanceOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image_pad, pos, psize, ksize):
        channel, height, width = image_pad.shape
        mc = pos.shape[0]
        out = torch.zeros((mc, psize, psize), dtype=torch.float32, device=image_pad.device)
        _lib.check(_lib.lib().ssg_compute_similarity(
            image_pad.data_ptr(), pos.data_ptr(), out.data_ptr(), mc, psize, ksize, height, width, channel,
            torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(image_pad, pos)
        ctx.sizes = (psize, ksize)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        image_pad, pos = ctx.saved_tensors
        psize, ksize = ctx.sizes
        channel, height, width = image_pad.shape
        grads = grad_output.to(torch.float32).contiguous()
        image_grad = torch.zeros_like(image_pad)
        _lib.check(_lib.lib().ssg_compute_similarity_backward(
            image_pad.data_ptr(), grads.data_ptr(), pos.data_ptr(), image_grad.data_ptr(), pos.shape[0], psize, ksize,
            height, width, channel, torch.cuda.current_stream().cuda_stream))
        return image_grad, None, None, None


def compute_similarity(image, mask, psize=25, ksize=9):
    """image (C,H,W) fp32 on the GPU, mask (H,W) on the GPU -> (N, psize, psize)."""
    if not image.is_cuda or not mask.is_cuda:
        raise RuntimeError(
            f"compute_similarity only accepts tensors on GPU memory but image({image.device}), mask({mask.device})")
    plen = psize // 2
    image_pad = F.pad(image.to(torch.float32), (plen, plen, plen, plen), mode="reflect").contiguous()
    mask_pad = F.pad(mask, (plen, plen, plen, plen), mode="constant")
    pos = torch.nonzero(mask_pad == 1).to(torch.int32).contiguous()   # (N,2) (Y,X), row-major
    return _DistanceOp.apply(image_pad, pos, int(psize), int(ksize))
