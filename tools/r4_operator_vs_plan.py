"""One image of the C2 batch (4,820 edge pixels): the reference operator's path (direct kernels in `pos` order) against
the engine's own per-image path (device edge list + plan -> dense tiles + merged direct kernels), forward and
forward + backward, HIP-event time of the launches alone and wall clock of the whole Python call.  Decides whether
building a plan inside ssg_compute_similarity[_backward] would pay."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth, compute_similarity
from ssl_amd.losses import set_lazy
from ssl_amd.losses.loss_util import eager_rows
dev = torch.device("cuda:0")
sizes = [(256, 256, 100), (512, 512, 2000)]
for H, W, seed in sizes:
    gt = synth.natural_like(seed, H, W)
    mask = synth.laplacian_edge_mask(gt).astype(np.float32)
    img = torch.as_tensor(gt, device=dev)
    m = torch.as_tensor(mask, device=dev)
    n = int(mask.sum())
    cot = torch.rand(n, 25, 25, device=dev)
    x = img.clone().requires_grad_(True)

    def op_fwd():
        return compute_similarity(image=x, mask=m, psize=25, ksize=9)

    def op_both():
        x.grad = None
        op_fwd().backward(cot)

    x4 = img[None].clone().requires_grad_(True)
    m4 = m[None, None]
    cot2 = cot.reshape(1, n, 625)

    def plan_fwd():
        return eager_rows(x4, m4, 'ch0', 25, 9, 1.0, 1e-10, True)

    def plan_both():
        x4.grad = None
        plan_fwd().backward(cot2)

    def wall(fn, it=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(it):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / it * 1e3

    print(f"{H}x{W}: N = {n}")
    print(f"  operator (pad + nonzero + direct kernels, pos order): fwd {wall(op_fwd):.4f} ms, fwd+bwd {wall(op_both):.4f} ms")
    print(f"  engine per-image path (edge list + plan + sync + dense/merged kernels + epilogue): fwd {wall(plan_fwd):.4f} ms, fwd+bwd {wall(plan_both):.4f} ms")
    el = engine.edge_list(mask=m4, ks=25, capacity=n + 64)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def ev(fn, it=50):
        fn(); torch.cuda.synchronize(); st.record()
        for _ in range(it):
            fn()
        en.record(); en.synchronize()
        return st.elapsed_time(en) / it
    print(f"  edge list + plan alone (events): {ev(lambda: engine.edge_list(mask=m4, ks=25, capacity=n + 64)):.4f} ms")
    print(f"  forward launches alone, with plan: {ev(lambda: engine.ssg_map(x4.detach(), el.edges, el.counts, n, 25, 9, 1.0, fwd=el.fwd, order=el.order)):.4f} ms;"
          f" without plan (direct kernels, tile order): {ev(lambda: engine.ssg_map(x4.detach(), el.edges, el.counts, n, 25, 9, 1.0, order=el.order)):.4f} ms")
