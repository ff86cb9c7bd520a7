"""Shared body of the Diffusion-fork strategy tests: runs every operator-based `simself_strategy` of
`ssl_amd.losses.dm_loss_util` on the inputs of fixture F8 and compares with what the reference's own
epilogues produced there (tests/golden/make_golden.py:f8_dm_strategies).  Used on the GPU through the
C-ABI operator (test_gpu_parity.py) and on the CPU with the operator replaced by the oracle's distance
(test_cpu_host.py), which checks the host-side epilogue table without a device."""
import os

import numpy as np
import torch

F8 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f8_dm_strategies.npz")
P = "areaarea_mask_"


def run_all(dm, device, tol_map=1e-5, skip=()):
    """dm: the module ssl_amd.losses.dm_loss_util (possibly with `compute_similarity` patched)."""
    g = np.load(F8)
    ks, kc, sigma, raw = int(g["ks"]), int(g["kc"]), float(g["sigma"]), float(g["sigma_raw"])
    step = int(g["row_step"])
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    G, S = t(g["gt"]), t(g["sr"])
    M1, M3 = t(g["mask"][None, None]), t(g["mask3"][None])
    worst = {}

    def check(name, got, key=None):
        ref = g[key or name]
        got = got.detach().cpu().numpy()[0][::step]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = float(np.abs(got - ref).max())
        worst[key or name] = err
        assert err <= tol_map, (name, err)

    def smap(strategy, img=G, **kw):
        a = dict(mask=M1, simself_strategy=P + strategy, kernel_size=ks, scaling_factor=sigma, softmax=True,
                 kernel_size_center=kc, dh=int(g["dh"]), dw=int(g["dw"]))
        a.update(kw)
        return dm.similarity_map(img, **a)

    if "nonlocal" not in skip:
        check("nonlocal", smap("nonlocal", scaling_factor=raw).getitem())
    check("nonlocal_cuda_v1", smap("nonlocal_cuda_v1", scaling_factor=raw).getitem())
    check("nonlocal_cuda_v1_patch", smap("nonlocal_cuda_v1_patch", scaling_factor=raw).getitem())
    check("nonlocal_cuda_v2", smap("nonlocal_cuda_v2", scaling_factor=0.5).getitem())
    for name in ("nonlocalavg_cuda_v1", "nonlocalavg_cuda_v2", "eulardistanceavg_cuda_v1", "nonlocalavg_cuda_v3"):
        if name not in skip:
            check(name, smap(name).getitem())
    if "nonlocalavg_cuda_v1" not in skip:
        check("nonlocalavg_cuda_v1", smap("nonlocalavg_cuda_v1", softmax=False).getitem(),
              "nonlocalavg_cuda_v1_nosoftmax")
    check("nonlocalavg_cuda_v4", smap("nonlocalavg_cuda_v4", kernel_size_center=[int(k) for k in g["kc_list"]]).getitem())
    check("nonlocalavg_cuda_v1RGB", smap("nonlocalavg_cuda_v1RGB", mask=M3).getitem())
    check("nonlocalavg_cuda_v5", smap("nonlocalavg_cuda_v5", gene_type="sum").getitem(), "nonlocalavg_cuda_v5_sum")
    check("nonlocalavg_cuda_v5", smap("nonlocalavg_cuda_v5", gene_type="softmax", largest_k=int(g["largest_k"])).getitem(),
          "nonlocalavg_cuda_v5_softmax_top16")
    a, b = smap("nonlocalavg_cuda_maxh_v1", img_sr=S).getitem_simmutual()
    # the shared bandwidth (q_gt - q_sr) / log(q_gt^2 / q_sr^2) cancels where the two distances are close, so
    # the map amplifies the fp32 rounding of the distances themselves (the reference's epilogue run in fp32 on
    # exactly-rounded distances is already 1e-4 off its fp64 run; the fp32 log of a ratio near 1 adds its own).  Check (i) the epilogue: the fork's formula
    # in fp64 on the operator's own fp32 distances, (ii) the fixture within the conditioning of the map.
    c = G.shape[1]
    d = [dm.compute_similarity(image=x[0], mask=M1[0, 0], psize=ks, ksize=kc).detach().double().reshape(1, -1, ks * ks)
         / (c * kc * kc) for x in (G, S)]
    h = (d[0] - d[1] + 1e-20) / (((d[0].pow(2) + 1e-20) / (d[1].pow(2) + 1e-20)).log() + 1e-20)
    for nm, v, q in (("maxh_gt", a, d[0]), ("maxh_sr", b, d[1])):
        e = torch.exp(-q / h)
        same_d = (e / (e.sum(-1, keepdim=True) + 1e-20)).cpu().numpy()
        got = v.detach().cpu().numpy()
        worst[nm + "_epilogue"] = float(np.abs(got - same_d).max())
        assert worst[nm + "_epilogue"] <= 1e-3, (nm, worst[nm + "_epilogue"])
        worst[nm] = float(np.abs(got[0][::step] - g[nm]).max())
        assert worst[nm] <= 2e-3, (nm, worst[nm])

    # learnable bandwidth: maps, loss, d/dsigma, d/dsr
    tm = dm.trainable_similarity_map(scaling_factor=sigma).to(device)
    Sg = S.clone().requires_grad_(True)
    s0, s1 = tm(G, Sg, M1, kernel_size_search=ks, kernel_size_center=kc, softmax=True)
    check("trainable", s0, "trainable_s_gt")
    check("trainable", s1, "trainable_s_sr")
    loss = (s0 - s1).pow(2).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["trainable_loss"])) <= 2e-4 * abs(float(g["trainable_loss"]))
    ds, ds_ref = float(tm.sigma.grad[0]), float(g["trainable_dsigma"][0])
    assert abs(ds - ds_ref) <= 2e-3 * abs(ds_ref), (ds, ds_ref)
    dsr, dsr_ref = Sg.grad.cpu().numpy(), g["trainable_dsr"]
    worst["trainable_dsr"] = float(np.abs(dsr - dsr_ref).max() / np.abs(dsr_ref).max())
    assert worst["trainable_dsr"] <= 1e-4
    assert abs(tm.getitem_h() - sigma) < 1e-7

    # gradient through an epilogue variant (centre removed + 1e-6 normaliser) under a fixed cotangent
    if "nonlocalavg_cuda_v2" not in skip:
        Gg = G.clone().requires_grad_(True)
        s = smap("nonlocalavg_cuda_v2", img=Gg).getitem()
        cot = np.random.default_rng(int(g["v2_cotangent_seed"])).standard_normal(tuple(s.shape))
        (s * t(cot)).sum().backward()
        ref = g["v2_dimg"]
        worst["v2_dimg"] = float(np.abs(Gg.grad.cpu().numpy() - ref).max() / np.abs(ref).max())
        assert worst["v2_dimg"] <= 1e-4
    return worst
