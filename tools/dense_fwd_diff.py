"""Debug: dense-tile forward vs direct forward, element by element (same inputs, thresholds 28 and 0)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
ks, kw, sigma = 25, 9, float(sys.argv[1]) if len(sys.argv) > 1 else 0.004
gt = synth.natural_like(400, 96, 96)[None]
sr = synth.degrade(gt[0], 401)[None]
mask = synth.laplacian_edge_mask(gt[0])[None]
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
out = {}
for thr in (28, 0):
    engine.set_dense_threshold(thr)
    st = engine.LossStep(1, 3, 96, 96, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev)
    loss, grad = st(T(sr), T(gt), T(mask[:, None].astype(np.float32)))
    n = int(st.counts[0])
    out[thr] = (st.ssg_sr[:n].clone(), st.ssg_gt[:n].clone(), grad.clone(), st.edges()[:n].cpu().numpy(), loss.clone())
a, b = out[28], out[0]
print("n", a[0].shape, "loss", a[4].tolist(), b[4].tolist())
for nm, x, y in (("ssg_sr", a[0], b[0]), ("ssg_gt", a[1], b[1])):
    d = (x - y).abs()
    rel = d / y.abs().clamp(min=1e-30)
    print(nm, "max abs", float(d.max()), "max rel (y>1e-12)", float(rel[y > 1e-12].max()))
    bad = (d > 1e-6).nonzero()
    print(" n bad", bad.shape[0])
    if bad.shape[0]:
        rows = bad[:, 0].unique()
        print(" bad rows", rows[:20].tolist(), "edges", a[3][rows[:10].cpu().numpy()].tolist())
        r0 = int(rows[0]); qs = bad[bad[:, 0] == r0][:, 1]
        print(" row", r0, "bad q (qy,qx):", [(int(q) // ks, int(q) % ks) for q in qs[:30]])
        print(" vals", x[r0, qs[:5]].tolist(), y[r0, qs[:5]].tolist())
print("grad max diff", float((a[2] - b[2]).abs().max()), "max", float(b[2].abs().max()))
# dense forward + direct backward (profile bit 24: no split backward)
from ssl_amd import _lib
engine.set_dense_threshold(28)
_lib.lib().ssg_set_profile_mask(1 << 24)
st = engine.LossStep(1, 3, 96, 96, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev)
loss, grad = st(T(sr), T(gt), T(mask[:, None].astype(np.float32)))
print("dense fwd + direct bwd: grad max diff vs all-direct", float((grad - b[2]).abs().max()))
_lib.lib().ssg_set_profile_mask(0)
import os
os.environ["SSG_DETERMINISTIC"] = "0"
st = engine.LossStep(1, 3, 96, 96, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev, deterministic=False)
loss, grad = st(T(sr), T(gt), T(mask[:, None].astype(np.float32)))
print("dense fwd + dense bwd, float atomics: grad max diff vs all-direct", float((grad - b[2]).abs().max()))
d = (grad - b[2]).abs()
idx = (d > 5e-7).nonzero()
print("bad px", idx.shape[0], idx[:12].tolist())
e = a[3]
r = [i for i in range(e.shape[0]) if e[i, 1] == 54 and e[i, 2] == 72]
print("row of (54,72):", r)
if r:
    r = r[0]
    for nm, x, y in (("sr", a[0], b[0]), ("gt", a[1], b[1])):
        rel = ((x[r] - y[r]).abs() / y[r].clamp(min=1e-12))
        top = rel.argsort(descending=True)[:12]
        print(nm, [(int(q) // ks, int(q) % ks, f"{float(rel[q]):.1e}", f"{float(y[r, q]):.2e}") for q in top])
ys, xs = idx[:, 2], idx[:, 3]
print("bad bbox y", int(ys.min()), int(ys.max()), "x", int(xs.min()), int(xs.max()))
top = d.flatten().argsort(descending=True)[:5]
print("worst px", [(int(t) // (96 * 96), (int(t) // 96) % 96, int(t) % 96, float(d.flatten()[t])) for t in top])
near = [i for i in range(e.shape[0]) if 40 <= e[i, 1] <= 70 and 56 <= e[i, 2] <= 90]
print("edge px near:", [(int(e[i, 1]), int(e[i, 2])) for i in near][:60])
for r in near[:400]:
    for nm, x, y in (("sr", a[0], b[0]), ("gt", a[1], b[1])):
        rel = ((x[r] - y[r]).abs() / y[r].clamp(min=1e-9))
        if float(rel.max()) > 3e-5:
            top = rel.argsort(descending=True)[:6]
            print(nm, (int(e[r, 1]), int(e[r, 2])), [(int(q) // ks, int(q) % ks, f"{float(rel[q]):.1e}", f"{float(y[r, q]):.2e}") for q in top])
from oracle import ssg_oracle as orc
rsel = [i for i in range(e.shape[0]) if e[i, 1] == 52 and 65 <= e[i, 2] <= 90]
pos = e[rsel][:, 1:]
for nm, img, x, y in (("sr", sr, a[0], b[0]), ("gt", gt, a[1], b[1])):
    ref = orc.ssg_epilogue(orc.distance(img[0].astype(np.float64), pos, ks, kw), kw, 3, sigma, True).reshape(len(rsel), -1)
    ref = torch.as_tensor(ref, device=dev)
    for lab, z in (("dense", x), ("direct", y)):
        zz = z[torch.as_tensor(rsel, device=dev)].double()
        rel = ((zz - ref).abs() / ref.clamp(min=1e-9))
        print(nm, lab, "max abs err", float((zz - ref).abs().max()), "max rel err (ref>1e-9)", float(rel[ref > 1e-9].max()), "mean rel", float(rel[ref > 1e-9].mean()))
pos = e[:, 1:]
for nm, img, x, y in (("sr", sr, a[0], b[0]), ("gt", gt, a[1], b[1])):
    ref = orc.ssg_epilogue(orc.distance(img[0].astype(np.float64), pos, ks, kw), kw, 3, sigma, True).reshape(len(pos), -1)
    ref = torch.as_tensor(ref, device=dev)
    for lab, z in (("dense", x), ("direct", y)):
        rel = ((z.double() - ref).abs() / ref.clamp(min=1e-9)) * (ref > 1e-9)
        w = rel.max(dim=1).values
        top = w.argsort(descending=True)[:6]
        print(nm, lab, "worst rows", [(int(e[t, 1]), int(e[t, 2]), f"{float(w[t]):.1e}") for t in top])
refs = {}
for nm, img in (("sr", sr), ("gt", gt)):
    refs[nm] = torch.as_tensor(orc.ssg_epilogue(orc.distance(img[0].astype(np.float64), pos, ks, kw), kw, 3, sigma, True).reshape(len(pos), -1), device=dev)
dref = refs["sr"] - refs["gt"]
for lab, o in (("dense", a), ("direct", b)):
    dd = (o[0] - o[1]).double()
    flip = (torch.sign(dd) != torch.sign(dref))
    wgt_ = torch.maximum(refs["sr"], refs["gt"]) * flip
    top = wgt_.flatten().argsort(descending=True)[:6]
    print(lab, "sign flips", int(flip.sum()), [(int(e[int(t) // 625, 1]), int(e[int(t) // 625, 2]), (int(t) % 625) // ks, (int(t) % 625) % ks,
          f"s={float(refs['sr'].flatten()[t]):.3e}", f"dref={float(dref.flatten()[t]):.2e}", f"d32={float(dd.flatten()[t]):.2e}") for t in top])
