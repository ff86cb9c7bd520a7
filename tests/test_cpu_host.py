"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the
header declares; host logic (synthetic inputs, sharding, the optional loss
all-reduce over gloo with world_size 2).  No compute call is made here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    from ssl_amd import _lib
    _lib.build()
    assert os.path.exists(_lib.SO_PATH)
    inc = os.path.dirname(_lib.HEADER)
    hdr = open(_lib.HEADER).read() + open(os.path.join(inc, "similarity.h")).read()
    declared = set(re.findall(r"\b(ssg_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"ssg_stream_t"}
    assert len(declared) >= 18
    # (the declarations inside `#ifdef SSG_PROFILE` belong to the profiling build only -- next test)
    profile_only = {"ssg_set_profile_mask", "ssg_prof_occupancy", "ssg_prof_strip_times", "ssg_prof_set_lds_poison"}
    L = ctypes.CDLL(_lib.SO_PATH)
    for name in sorted(declared - profile_only):
        assert hasattr(L, name), f"{name} declared in include/*.h but not exported"
    # the reference's own two functions (similarity.h:2-23, C++ linkage): declared in include/similarity.h,
    # exported under the Itanium-mangled names the reference's similaritywrapper.cpp links against
    ref_hdr = open(os.path.join(inc, "similarity.h")).read()
    assert re.search(r"void _compute_similarity\(const float \*image, const int \*pos, float \*out,", ref_hdr)
    assert re.search(r"void _compute_similarity_backward\(const float \*image, const float \*grads, const int \*pos,", ref_hdr)
    for name in _lib.CXX_SYMBOLS:
        assert hasattr(L, name), f"{name} not exported"
    demangled = subprocess.run(["c++filt"] + list(_lib.CXX_SYMBOLS), stdout=subprocess.PIPE, text=True).stdout.split("\n")
    assert demangled[0] == "_compute_similarity(float const*, int const*, float*, int, int, int, int, int, int)"
    assert demangled[1] == ("_compute_similarity_backward(float const*, float const*, int const*, float*, int, int, int, "
                            "int, int, int)")
    # every prototype the Python binding uses is declared in the header
    assert set(_lib.PROTOTYPES) <= declared - profile_only
    lib = _lib.lib()
    assert lib.ssg_abi_version() == 6
    assert lib.ssg_status_string(0) == b"ok"
    assert b"LDS" in lib.ssg_status_string(-2)
    assert lib.ssg_kernel_name(25, 9, 0).startswith(b"ssg_fwd_")
    assert lib.ssg_kernel_name(7, 3, 1) == b"ssg_bwd_generic"
    assert lib.ssg_loss_workspace_bytes(16, 256, 256, 100000, 25) > 100000 * 12


def test_product_library_has_no_profiling_switch():
    """Kernel-phase ablations / launch skipping (results then wrong) exist only in libssg_hip_prof.so
    (-DSSG_PROFILE): the product library neither exports ssg_set_profile_mask nor reads SSG_DEBUG_SKIP."""
    from ssl_amd import _lib
    _lib.build()
    prod = ctypes.CDLL(_lib.SO_PATH)
    assert not hasattr(prod, "ssg_set_profile_mask")
    blob = open(_lib.SO_PATH, "rb").read()
    assert b"SSG_DEBUG_SKIP" not in blob
    # ... nor any other environment switch (round 5: SSG_DENSE_THR, SSG_OVERLAP, SSG_BWD_QSPLIT, SSG_STRIPS, SSG_TILE_MAJOR,
    # SSG_OP_PLAN_FROM, SSG_BWD_TM_SPLIT, SSG_EDGE_BANDED, SSG_FWD_SMALL are honoured by the profiling build only)
    import re
    assert not re.findall(rb"SSG_[A-Z_]{3,}\x00", blob), re.findall(rb"SSG_[A-Z_]{3,}\x00", blob)
    assert b"getenv" not in blob
    prof = ctypes.CDLL(_lib.PROF_SO_PATH)
    assert hasattr(prof, "ssg_set_profile_mask")
    assert b"SSG_DEBUG_SKIP" in open(_lib.PROF_SO_PATH, "rb").read()
    hdr = open(_lib.HEADER).read()
    i = hdr.index("int ssg_set_profile_mask")
    assert hdr.rfind("#ifdef SSG_PROFILE", 0, i) > hdr.rfind("#endif", 0, i)
    # the package's product path never asks for the profiling build
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ssl_amd")):
        for f in files:
            if f.endswith(".py") and f != "_lib.py":
                src = open(os.path.join(dirpath, f)).read()
                assert "lib_prof" not in src and "profile_build" not in src, f


def test_gfx950_code_object_present():
    """The .so carries a gfx950 code object (not a host-only build)."""
    from ssl_amd import _lib
    blob = open(_lib.SO_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"ssg_fwd_tiled" in blob and b"ssg_bwd_tiled" in blob


def test_no_cpu_fallback_in_product_path():
    from ssl_amd import similarity_map, SSGLoss, compute_similarity
    img = torch.rand(1, 3, 16, 16)
    m = torch.zeros(1, 1, 16, 16)
    m[0, 0, 3, 3] = 1
    for mode in ("hip", "cuda", "pytorch"):
        with pytest.raises(RuntimeError):
            similarity_map(img, m, ssl_mode=mode, kernel_size_search=5, kernel_size_window=3)
    with pytest.raises(ValueError):
        similarity_map(img, m, ssl_mode="tpu")
    with pytest.raises(RuntimeError):
        compute_similarity(img[0], m[0, 0])
    with pytest.raises(RuntimeError):
        SSGLoss(5, 3)(img, img, m)
    # the package never imports the oracle
    src = subprocess.run(["grep", "-rl", "oracle", os.path.join(ROOT, "ssl_amd"), "--include=*.py"],
                         stdout=subprocess.PIPE, text=True).stdout.split()
    assert src == [], f"product code references the oracle: {src}"


def test_synthetic_inputs_are_deterministic_and_on_spec():
    from ssl_amd import synth
    sr, gt, mask = synth.make_batch(2, 128, 128)
    sr2, gt2, mask2 = synth.make_batch(2, 128, 128)
    assert synth.checksum(sr, gt, mask) == synth.checksum(sr2, gt2, mask2)
    assert sr.dtype == gt.dtype == mask.dtype == np.float32
    assert gt.min() >= 0 and gt.max() <= 1 and np.array_equal(np.rint(gt * 255) / 255, gt.astype(np.float64).round(7)) \
        or np.allclose(np.rint(gt * 255), gt * 255, atol=1e-4)
    dens = mask.mean()
    assert 0.06 <= dens <= 0.10, dens
    s1, g1, m1 = synth.uniform_case()
    assert m1.sum() == 209 and m1[0, 0, 0, 0] == 1 and m1[0, 0, -1, -1] == 1
    # pinned checksum of the benchmark's first image (detects silent generator drift)
    assert synth.checksum(synth.natural_like(100, 64, 64)) == synth.checksum(synth.natural_like(100, 64, 64))
    # numpy mask == oracle mask (independent restatements of generate_mask.py)
    from oracle import ssg_oracle as orc
    assert np.array_equal(orc.edge_mask_chw(gt[0]), mask[0, 0].astype(np.uint8))
    assert np.array_equal(synth.mask_stride_pattern(7, 9, 3) * 1, orc.mask_stride(np.ones((7, 9), np.uint8), 3))


def test_shard_range_partitions():
    from ssl_amd.dist import shard_range
    for n in (0, 1, 7, 16, 33):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SSG_ROOT"])
from ssl_amd.dist import shard_range, global_mean_losses
from oracle import ssg_oracle as orc           # test infrastructure: stands in for the GPU kernels
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(5)
B, ks, kw, sigma = 4, 5, 3, 0.5
sr = rng.random((B, 3, 12, 12)); gt = rng.random((B, 3, 12, 12))
masks = (rng.random((B, 12, 12)) < 0.1).astype(np.uint8); masks[1] = 0   # one empty image
lo, hi = shard_range(B, rank, world)
r = orc.ssg_loss(sr[lo:hi], gt[lo:hi], masks[lo:hi], ks, kw, sigma, 1e3, 1e3, want_grad=False)
l1, kl, n = global_mean_losses(torch.tensor(r["l1"], dtype=torch.float64), torch.tensor(r["kl"], dtype=torch.float64), r["n_edges"], ks)
full = orc.ssg_loss(sr, gt, masks, ks, kw, sigma, 1e3, 1e3, want_grad=False)
assert int(n) == full["n_edges"], (int(n), full["n_edges"])
assert abs(float(l1) - full["l1"]) < 1e-9 * abs(full["l1"]) + 1e-12, (float(l1), full["l1"])
assert abs(float(kl) - full["kl"]) < 1e-9 * abs(full["kl"]) + 1e-12
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_gloo_sharding_and_global_mean(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, SSG_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_diffusion_fork_epilogue_table_matches_reference_fixture(monkeypatch):
    """Host logic of ssl_amd.losses.dm_loss_util (every operator-based strategy + trainable_similarity_map)
    vs fixture F8, which holds what the reference's own epilogues produced.  The device operator is replaced
    by the oracle's distance here (checker standing in for the kernel; the GPU suite runs the same body
    through the C ABI); the fused strategy `nonlocalavg_cuda_v1` has no host epilogue and is GPU-only."""
    from oracle import ssg_oracle as orc
    from ssl_amd.losses import dm_loss_util as dm
    import dm_cases

    class Dist(torch.autograd.Function):
        @staticmethod
        def forward(ctx, image, mask, psize, ksize):
            pos = orc.mask_to_pos(mask.numpy())
            img = image.detach().numpy().astype(np.float64)
            ctx.meta = (img, pos, psize, ksize)
            return torch.as_tensor(orc.distance(img, pos, psize, ksize)).float()

        @staticmethod
        def backward(ctx, g):
            img, pos, psize, ksize = ctx.meta
            return torch.as_tensor(orc.distance_backward(img, pos, psize, ksize, g.numpy().astype(np.float64))).float(), None, None, None

    monkeypatch.setattr(dm, "compute_similarity", lambda image, mask, psize=25, ksize=9: Dist.apply(image, mask, psize, ksize))
    worst = dm_cases.run_all(dm, torch.device("cpu"), skip=("nonlocalavg_cuda_v1",))
    assert len(worst) >= 16
    with pytest.raises(NotImplementedError):
        dm.similarity_map(torch.zeros(1, 3, 8, 8), torch.ones(1, 1, 8, 8), simself_strategy="imgimg")


def test_mask_files_roundtrip_and_density_report_format(tmp_path):
    """ssl_amd.maskio: .mat written as the reference writes it (key 'mat', integer (H,W), compressed;
    generate_mask.py:41) and read back as the datasets do (float32 (H,W,1),
    my_realesrgan_image_mask_dataset.py:79-83); statis.txt lines as generate_mask_simmatrix.py:71-90."""
    import scipy.io as sio
    from ssl_amd import maskio
    rng = np.random.default_rng(3)
    m = (rng.random((37, 53)) < 0.1).astype(np.uint8)
    path = str(tmp_path / "a.mat")
    maskio.save_mask_mat(path, m)
    raw = sio.loadmat(path)["mat"]
    assert raw.shape == (37, 53) and np.issubdtype(raw.dtype, np.integer) and np.array_equal(raw, m)
    assert os.path.getsize(path) < m.size * 8 // 4            # compressed
    back = maskio.load_mask_mat(path)
    assert back.dtype == np.float32 and back.shape == (37, 53, 1) and back.flags["C_CONTIGUOUS"]
    assert np.array_equal(back[..., 0], m.astype(np.float32))
    with pytest.raises(ValueError):
        maskio.save_mask_mat(path, np.zeros((2, 3, 4)))
    from PIL import Image
    maskio.save_mask_png(str(tmp_path / "a.png"), m)
    assert np.array_equal(np.array(Image.open(str(tmp_path / "a.png"))), m * 255)
    rep = maskio.DensityReport(str(tmp_path / "statis.txt"))
    rep.add("img0", 1000, 400, 80)
    rep.add("img1", 2000, 500, 100)
    lines = rep.close()
    text = open(str(tmp_path / "statis.txt")).read()
    assert text.startswith("img0:\nImage number-1000, grad number-400-0.4000, mask number-80-0.0800\n\nimg1:\n")
    assert lines == ["Maximum of grad is 500.00, percentage is 0.250000", "Minium of grad is 400.00, percentage is 0.2000",
                     "Average of grad is 450.00, percentage is 0.3000", "Maximum of mask is 100.00, percentage is 0.050000",
                     "Minium of grad is 80.00, percentage is 0.0400", "Average of grad is 90.00, percentage is 0.0600"]
    assert text.endswith("\n".join(lines) + "\n")


def test_bench_self_launch_two_ranks_strong_scaling_dry_run():
    """`python bench.py --gpus 2` with no launcher spawns its own ranks (one process per GPU on the box; here the
    --dry-run plumbing on CPU/gloo): 127.0.0.1 rendezvous, the 16-image batch split 2-ways, edge pixels summed and
    the step time max-reduced across ranks, ONE JSON line from rank 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--scaling", "strong", "--dry-run"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["ranks_seen"] == 2 and r["scaling"] == "strong" and r["dry_run"] is True
    assert r["config"]["images_rank0"] == 8
    from ssl_amd import synth
    _, _, mask = synth.make_batch(16, 256, 256)
    assert r["config"]["edge_px_total"] == float(mask.sum())
    assert r["config"]["edge_px_rank0"] == int(mask[:8].sum())
    import bench
    assert [bench.shard_images(16, k, 3) for k in range(3)] == [(0, 5), (5, 10), (10, 16)]


def test_argument_checks_need_no_gpu():
    """Status codes of include/ssg_hip.h that are decided before anything is launched: k_w > k_s and even sizes are
    SSG_E_BADARG (-1) at every entry point (the kernels would index their LDS tile out of bounds), an image side
    <= k_s/2 is SSG_E_IMAGESMALL (-4) like torch's reflect-pad error, a short workspace SSG_E_WORKSPACE (-3)."""
    from ssl_amd import _lib
    L = _lib.lib()
    one = ctypes.c_void_p(8)          # never dereferenced: the checks come first
    assert L.ssg_compute_similarity(one, one, one, 4, 5, 9, 40, 40, 3, None) == -1       # k_w > k_s
    assert L.ssg_compute_similarity(one, one, one, 4, 6, 3, 40, 40, 3, None) == -1       # even k_s
    assert L.ssg_compute_similarity_backward(one, one, one, one, 4, 5, 9, 40, 40, 3, None) == -1
    assert L.ssg_compute_similarity(one, one, one, 0, 5, 3, 40, 40, 3, None) == 0        # nothing to do
    assert L.ssg_map_forward(one, None, 1, 3, 12, 40, one, None, None, None, None, 4, 25, 9, 1.0, 1e-10, 1, one, None,
                             None, None) == -4
    assert L.ssg_map_forward(one, None, 1, 3, 40, 40, one, None, None, None, None, 4, 9, 25, 1.0, 1e-10, 1, one, None,
                             None, None) == -1
    assert L.ssg_loss_fwd_bwd(one, one, one, 0, 1, 1, 3, 40, 40, 25, 9, 1.0, 1e-10, 1, 1.0, 1.0, 0, 20.0, 100, one, one,
                              one, one, one, one, 16, None, None) == -3
    # fused step: workspace / grad_fix / grad_sr must be 16-byte aligned (its clears are 16-byte stores): SSG_E_ALIGN
    assert L.ssg_loss_fwd_bwd(one, one, one, 0, 1, 1, 3, 40, 40, 25, 9, 1.0, 1e-10, 1, 1.0, 1.0, 0, 20.0, 100, one, one,
                              one, one, one, one, 1 << 40, None, None) == -5
    assert b"aligned" in L.ssg_status_string(-5) and b"plan" in L.ssg_status_string(-6)
    assert L.ssg_augment_crop(one, one, 2, 1, 3, 8, 8, 4, 4, one, None) == -1            # element size 2
    assert L.ssg_grad_fix_bytes(2, 3, 16, 16) == 8 * (2 * 3 * 16 * 16 + 8)
    assert L.ssg_backward_scratch_bytes(100, 25) >= 100 * 625 * 4


def test_workspace_layout_of_the_fused_call():
    """ssg_loss_workspace_layout (host arithmetic only): the pieces lie in order inside ssg_loss_workspace_bytes(), the
    row-major scratch rows of the fused step behind them, and -- k_s = 49 only -- two tile-major regions of capacity / 128
    slots plus a spare one each; ssg_loss_rows_bytes() covers exactly that."""
    from ssl_amd import _lib
    L = _lib.lib()
    lay = (ctypes.c_size_t * 9)()
    for ks, cap in ((25, 5000), (49, 5000), (49, 100), (11, 333)):
        B, H, W = 2, 70, 100
        assert L.ssg_loss_workspace_layout(B, H, W, cap, ks, 1, lay) == 0
        base, rows = L.ssg_loss_workspace_bytes(B, H, W, cap, ks), L.ssg_loss_rows_bytes(cap, ks)
        edges, rank, plan, rsc, r0, r1, t0, t1, slots = list(lay)
        assert 0 == edges < rank < plan < rsc < base == r0 < r1 and all(o % 256 == 0 for o in (rank, plan, rsc, r0, r1))
        assert base - rsc >= 16 * cap and rank - edges >= 12 * cap and r1 - r0 >= 4 * cap * ks * ks
        if ks == 49:
            slot_bytes = 49 * 49 * 128 * 4
            assert slots == cap // 128 and r1 < t0 < t1 and t0 - r1 == r1 - r0
            assert t1 - t0 >= (slots + 1) * slot_bytes and base + rows - t1 >= (slots + 1) * slot_bytes
        else:
            assert (t0, t1, slots) == (0, 0, 0) and base + rows - r1 == r1 - r0
        # a materialising call: no row-major scratch rows, the tile-major regions (k_s 49) directly behind the base
        assert L.ssg_loss_workspace_layout(B, H, W, cap, ks, 0, lay) == 0
        assert (lay[4], lay[5]) == (0, 0) and lay[6] == (base if ks == 49 else 0)
        assert L.ssg_loss_tm_bytes(cap, ks) == (2 * (lay[7] - lay[6]) if ks == 49 else 0)
    assert L.ssg_loss_workspace_layout(1, 64, 64, 0, 25, 1, lay) == -1 and L.ssg_loss_workspace_layout(1, 64, 64, 10, 25, 1, None) == -1


def test_plan_built_for_another_tile_height_is_refused():
    """A dense/direct plan is cut for one tile height (8 rows for k_s <= 25, 4 for k_s = 49) and records the k_s it
    was built for; handing it to a call with the other geometry raises before anything is launched (the dense
    kernels would decode its tile ids with their own tile height)."""
    from ssl_amd import engine
    t = torch.zeros(4, dtype=torch.int32)
    fwd25 = engine.FwdPlan(t, t, t, 25)
    assert tuple(fwd25) == (t, t, t) and fwd25.ks == 25
    engine.check_plan(fwd25, 25)
    engine.check_plan(fwd25, 11)          # same 8-row tiles (only the direct kernels run for k_s = 11)
    engine.check_plan(None, 49)
    engine.check_plan((t, t, t), 49)      # a bare tuple carries no k_s: nothing to check
    with pytest.raises(ValueError, match="k_s = 25"):
        engine.check_plan(fwd25, 49)
    with pytest.raises(ValueError, match="4-row tiles"):
        engine.check_plan(engine.FwdPlan(t, t, t, 49), 25)
    el = engine.EdgeList(t, t, t, t, t, 49)
    assert el.ks == 49 and el.fwd.ks == 49 and tuple(el) == (t, t)
    # the entry points check before touching a tensor's device
    with pytest.raises(ValueError):
        engine.ssg_map(torch.zeros(1, 3, 64, 64).cuda() if torch.cuda.is_available() else _FakeGpu(), t, t, 4, 49, 13, 1.0,
                       fwd=fwd25) if torch.cuda.is_available() else engine.check_plan(fwd25, 49)


class _FakeGpu:
    is_cuda = True


def test_graft_entry_build_runs():
    """__graft_entry__.build() -- the driver's "does it build" check -- must keep up with the library (ABI version)."""
    import __graft_entry__ as g
    g.build()
