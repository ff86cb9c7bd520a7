#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/dd.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from ssl_amd import engine, synth, _lib
dev = torch.device("cuda:0")
sr_np, gt_np, _ = synth.make_batch(4, 256, 256)
rng = np.random.default_rng(0)
for dens in (0.16, 0.3, 0.5, 1.0):
    m = (rng.random((4, 1, 256, 256)) < dens).astype(np.float32)
    sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, m))
    el = engine.edge_list(mask=mask)
    n = int(el.counts[0]); nd = int(el.plan[1]) + int(el.plan[3])
    L = _lib.lib(); P = engine._ptr
    s1 = torch.empty((n, 625), device=dev); s2 = torch.empty((n, 625), device=dev)
    def f():
        _lib.check(L.ssg_map_forward(P(sr), P(gt), 4, 3, 256, 256, P(el.edges), P(el.order), P(el.rank), P(el.plan), P(el.counts), n, 25, 9, 1.0, 1e-10, 1, P(s1), P(s2), None, torch.cuda.current_stream().cuda_stream))
    f(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(5): f()
    en.record(); en.synchronize()
    print(f"density {dens:4.2f}: N={n} dense tiles={nd}  fwd {st.elapsed_time(en)/5:.3f} ms")
PY
for thr in 0 64 100 140 200; do echo "== SSG_DENSE_THR=$thr"; SSG_DENSE_THR=$thr python /tmp/dd.py 2>&1 | grep density; done
