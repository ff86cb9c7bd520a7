import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(autouse=True)
def _poisoned_allocator(request):
    """Audit mode (SSL_AMD_TEST_POISON=int:3 | nan | f32:1e30 | int:-1 ...): before every GPU test the caching
    allocator's free blocks are filled with the pattern, so that whatever a kernel reads without having written it is
    that pattern instead of a previous test's (often harmless) values.  Off by default; tools/r5_poison_suite.sh runs
    the GPU suite under several patterns."""
    spec = os.environ.get("SSL_AMD_TEST_POISON")
    if spec and request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            kind, _, val = spec.partition(":")
            torch.cuda.synchronize()
            n, blocks = 64 * 1024 * 1024, int(os.environ.get("SSL_AMD_TEST_POISON_BLOCKS", "16"))
            if kind == "int":
                junk = [torch.full((n,), int(val), dtype=torch.int32, device="cuda:0") for _ in range(blocks)]
            else:
                junk = [torch.full((n,), float("nan") if kind == "nan" else float(val), device="cuda:0") for _ in range(blocks)]
            # small-pool blocks as well (allocations below 1 MB come from 2 MB segments of their own)
            small = [torch.full((1 << 16,), junk[0][0].item(), dtype=junk[0].dtype, device="cuda:0") for _ in range(256)]
            del junk, small
    yield


@pytest.fixture(scope="session", autouse=True)
def _poisoned_lds():
    """Audit mode (SSL_AMD_TEST_LDS_POISON=<hex word>): the whole session runs on the PROFILING build with its LDS poison
    on -- every launch of the library is preceded by a kernel that fills the LDS of every CU with the word, so a kernel
    that reads LDS it has not written sees that word every time (include/ssg_hip.h: ssg_prof_set_lds_poison)."""
    spec = os.environ.get("SSL_AMD_TEST_LDS_POISON")
    if not spec:
        yield
        return
    from ssl_amd import _lib
    with _lib.profile_build() as L:
        L.ssg_prof_set_lds_poison(1, int(spec, 16))
        try:
            yield
        finally:
            L.ssg_prof_set_lds_poison(0, 0)
