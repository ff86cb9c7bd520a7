import torch, numpy as np
from ssl_amd import engine, synth
dev = torch.device('cuda:0')
ks, kw = 49, 13
H = W = 128
gt = torch.from_numpy(synth.natural_like(1400, H, W)[None]).float().to(dev)
sr = torch.from_numpy(synth.degrade(gt[0].cpu().numpy(), 1401)[None]).float().to(dev)
dense = torch.ones((1, 1, H, W), device=dev)
step = engine.LossStep(1, 3, H, W, ks, kw, 1.0, 1e-10, True, 1e3, 1e3, device=dev)
loss, grad = step(sr, gt, dense)
n = int(step.counts[0])
S = step.ssg_sr[:n].clone()
# reference rows from the direct kernels (no plan)
el = engine.edge_list(mask=dense, ks=ks)
ref = engine.ssg_map(sr, el.edges, el.counts, H * W, ks, kw, 1.0, 1e-10, True, order=el.order, fwd=None)
print('n', n, 'ref', None if ref is None else ref.shape)
d = (S - ref[:n]).abs()
bad_rows = (d.max(1).values > 1e-5).nonzero().flatten()
print('bad rows', bad_rows.numel(), 'of', n)
if bad_rows.numel():
    br = bad_rows.cpu().numpy()
    ys, xs = br // W, br % W
    print('y range', ys.min(), ys.max(), 'unique y', np.unique(ys)[:60])
    print('x unique', np.unique(xs)[:64])
    r0 = int(br[0])
    dd = d[r0].view(ks, ks)
    badq = (dd > 1e-5).nonzero()
    print('row', r0, 'bad q count', badq.shape[0], 'qy uniq', badq[:, 0].unique().tolist(), 'qx uniq', badq[:, 1].unique().tolist())
    print('S', S[r0].view(ks, ks)[badq[0, 0], :8].tolist(), 'ref', ref[r0].view(ks, ks)[badq[0, 0], :8].tolist())
    print('sum S', float(S[r0].sum()), 'sum ref', float(ref[r0].sum()))
