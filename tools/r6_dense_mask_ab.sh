#!/bin/bash
# rocprofv3 kernel stats of the 4 x 3x256x256 step under a 100 % mask (every tile TILE_HUGE: the four-chunk instantiations)
# for library builds gpurun_ab/<tag>/ ("work" = the working tree's):  tools/r6_dense_mask_ab.sh sb work
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp; R=$PWD
mkdir -p gpurun_ab/work; cp ssl_amd/csrc/libssg_hip*.so gpurun_ab/work/
cat > /tmp/dense_step.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
sr, gt, _ = synth.make_batch(4, 256, 256)
m = np.ones((4, 1, 256, 256), np.float32)
a, b, mm = (torch.as_tensor(x, device=dev) for x in (sr, gt, m))
step = engine.LossStep(4, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=4 * 256 * 256)
for _ in range(30): step(a, b, mm)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): step(a, b, mm)
e1.record(); e1.synchronize()
print("ms/step %.4f" % (e0.elapsed_time(e1) / 30))
PY
for t in "$@"; do
  cp gpurun_ab/$t/libssg_hip.so gpurun_ab/$t/libssg_hip_prof.so ssl_amd/csrc/
  python /tmp/dense_step.py 2>/dev/null | sed "s/^/$t /"
  O=$R/gpurun_out/r6dm_$t; rm -rf $O
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python /tmp/dense_step.py > /dev/null 2>&1)
  f=$(find $O -name "*kernel_stats.csv" | head -1); head -7 $f | cut -d, -f1-4 | cut -c1-110 | sed "s/^/$t  /"
  find $O -name "*kernel_trace.csv" -delete
done
cp gpurun_ab/work/libssg_hip*.so ssl_amd/csrc/
