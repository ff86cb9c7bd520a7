"""C3 / C4-shaped end-to-end generator steps on the GPU (SURVEY section 8d reporting grid: "step time with the SSL
engine vs with SSL disabled"): the drop-in SSGLoss inside a stock-PyTorch training step, exactly where the
reference's per-image loop sits (realesrganssl_model.py:379-430, ddpmssl.py:424-428).  The networks are random-
weight stand-ins of the published shapes (context for the measurement, not product code).  The figures are printed
(`pytest -s`; bench.py reports the same two functions as `extra.c3_step_share` / `extra.c4_step_share`); the assertions are on correctness (finite
gradients into the network, same loss value as the C-ABI step) and on the loss staying a minor share of the step."""
import importlib.util
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_mod():
    spec = importlib.util.spec_from_file_location("train_step_bench", os.path.join(ROOT, "tools", "train_step_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_c3_and_c4_shaped_generator_steps_with_and_without_ssl():
    """The two functions bench.py reports as `extra.c3_step_share` / `extra.c4_step_share`, with correctness checks
    hooked in: finite gradients into the network, and the module's losses equal to the C-ABI step's on the same SR."""
    from ssl_amd import engine
    dev = torch.device("cuda:0")
    tsb = _bench_mod()

    def check3(net, last, gt, mask):
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
        ref = engine.LossStep(4, 3, 256, 256, 25, 9, 0.004, 1e-10, True, 1e3, 1e3, device=dev)
        loss_ref, _ = ref(last["sr"].contiguous(), gt, mask)
        assert abs(float(last["l1"]) - float(loss_ref[0])) <= 1e-6 * float(loss_ref[0])
        assert abs(float(last["kl"]) - float(loss_ref[1])) <= 1e-6 * float(loss_ref[1]) + 1e-12

    def check4(dec):
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in dec.parameters())

    out = {"c3": tsb.c3_step_share(dev, check=check3), "c4": tsb.c4_step_share(dev, check=check4)}
    assert out["c3"]["ssl_share"] < 0.15
    assert out["c4"]["ssl_ms"] < 5.0   # ms: the loss itself is ~0.5 ms at this size
    print(json.dumps(out, indent=1))
