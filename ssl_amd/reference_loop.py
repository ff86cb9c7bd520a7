"""The reference's OWN caller loop around `similarity_map`, kept so that it can be run and timed unchanged.

These two functions restate what the training models of ChrisDud0257/SSL do with the loss-module API every step --
they are the reference's side of the drop-in boundary, not part of the engine.  tests/test_gpu_ref_api.py runs them
against the caller-loop fixtures (F1, F2, F3, F9), bench.py times them (`extra.ref_api`), and INTEGRATION.md points at
them: a training model that keeps its loop and only changes its imports executes exactly this.

  gan_selfsim_block   <- GAN-Based-SR/basicsr/models/realesrganssl_model.py:379-430 (`train_net_g`, SSL block)
                         and its KAIR twin train_BSGRAN/models/model_ssl.py:285-340
  dm_issl             <- Diffusion-Based-SR/ldm/models/diffusion/ddpmssl.py:438-513 (`issl`)
  stride_pattern      <- realesrganssl_model.py:64-72 / ddpmssl.py:47-56 (the eye-tiled `self.mask_stride`)

What the loop does per image i of the batch: slice mask, optionally multiply with the stride pattern, one host
synchronisation (`mask.sum() == 0`), skip empty images, two clones and a `similarity_map` construction for SR, the same
for GT, append; afterwards `torch.cat(dim=1)` of both lists and the two criterion modules on the concatenated tensors.
"""
import math

import torch


def stride_pattern(size, stride, device):
    """(1,1,size,size) float32: 1 where y % stride == x % stride (an identity matrix tiled over the crop)."""
    eye = torch.eye(stride, stride, dtype=torch.float32)
    reps = math.ceil(size / stride)
    return eye.repeat(reps, reps)[:size, :size].unsqueeze(0).unsqueeze(0).to(device)


def gan_selfsim_block(similarity_map, cri_selfsim, cri_selfsim1, output, gt, gt_mask, ssl_setting, mask_stride=None):
    """One pass of the Real-ESRGAN model's SSL block.  `ssl_setting` is the YAML block of the same name
    (options/train/RealESRGANSSL/train_RealESRGANSSL_x4.yml:113-119); `mask_stride` the stride pattern tensor or
    None.  Returns (l_selfsim, l_selfsim_kl); a term is None when its criterion is None or every mask was empty
    (the model then adds nothing to `l_g_total`)."""
    sr_rows, gt_rows = [], []
    for i in range(gt.shape[0]):
        m = gt_mask[i, :].unsqueeze(0)
        if mask_stride is not None:
            m = mask_stride * m
        if m.sum() == 0:                     # (host synchronisation, as in the reference)
            continue
        kw = dict(ssl_mode=ssl_setting['ssl_mode'], kernel_size_search=ssl_setting['kernel_size_search'],
                  generalization=ssl_setting['generalization'], kernel_size_window=ssl_setting['kernel_size_window'],
                  sigma=ssl_setting['sigma'])
        s_sr = similarity_map(img=output[i, :].unsqueeze(0).clone(), mask=m.clone(), **kw).getitem()
        s_gt = similarity_map(img=gt[i, :].unsqueeze(0).clone(), mask=m.clone(), **kw).getitem()
        sr_rows.append(s_sr)
        gt_rows.append(s_gt)
    # the reference's own tests, literally: `len()` of the two LISTS before the cat, then `len()` of the concatenated
    # TENSORS (1, sum N, k_s^2) before each criterion (realesrganssl_model.py:409-411, 413, 419)
    if len(sr_rows) > 0 and len(gt_rows) > 0:
        sr_rows = torch.cat(sr_rows, dim=1)
        gt_rows = torch.cat(gt_rows, dim=1)
    l_selfsim = l_selfsim_kl = None
    if cri_selfsim is not None:
        if len(sr_rows) > 0 and len(gt_rows) > 0:
            l_selfsim = cri_selfsim(sr_rows, gt_rows)
    if cri_selfsim1 is not None:
        if len(sr_rows) > 0 and len(gt_rows) > 0:
            l_selfsim_kl = cri_selfsim1(sr_rows, gt_rows)
    return l_selfsim, l_selfsim_kl


def dm_issl(similarity_map, cri_selfsim, cri_selfsim1, sr, gt, mask, sslopt, mask_stride=None):
    """`issl` of the diffusion fork: the same loop with the fork's constructor arguments (`simself_strategy`,
    `kernel_size`, `scaling_factor`, `softmax_sr` / `softmax_gt`, `kernel_size_center`, ...).  All masks empty:
    (0.0, 0.0) like ddpmssl.py:492-493."""
    sr_rows, gt_rows = [], []
    for i in range(gt.shape[0]):
        m = mask[i, :].unsqueeze(0)
        if sslopt.get('mask_stride', 0) > 1:
            m = mask_stride * m
        if m.sum() == 0:
            continue
        common = dict(simself_strategy=sslopt['simself_strategy'], dh=sslopt.get('simself_dh', 16),
                      dw=sslopt.get('simself_dw', 16), kernel_size=sslopt['kernel_size'],
                      scaling_factor=sslopt['scaling_factor'], temperature=sslopt.get('temperature', 0),
                      crossentropy=sslopt.get('crossentropy', False), rearrange_back=sslopt.get('rearrange_back', True),
                      stride=1, pix_num=1, index=None, kernel_size_center=sslopt.get('kernel_size_center', 9),
                      mean=sslopt.get('mean', False), var=sslopt.get('var', False),
                      gene_type=sslopt.get('gene_type', "sum"), largest_k=sslopt.get('largest_k', 0))
        s_sr = similarity_map(img=sr[i, :].unsqueeze(0).clone(), mask=m.clone(),
                              softmax=sslopt.get('softmax_sr', False), **common).getitem()
        s_gt = similarity_map(img=gt[i, :].unsqueeze(0).clone(), mask=m.clone(),
                              softmax=sslopt.get('softmax_gt', False), **common).getitem()
        sr_rows.append(s_sr)
        gt_rows.append(s_gt)
    if len(sr_rows) <= 0 or len(gt_rows) <= 0:          # (ddpmssl.py:492-493)
        return 0.0, 0.0
    sr_rows = torch.cat(sr_rows, dim=1)
    gt_rows = torch.cat(gt_rows, dim=1)
    l_selfsim = cri_selfsim(sr_rows, gt_rows) if cri_selfsim is not None else None
    l_selfsim_kl = cri_selfsim1(sr_rows, gt_rows) if cri_selfsim1 is not None else None
    return l_selfsim, l_selfsim_kl
