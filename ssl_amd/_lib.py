"""ctypes binding of libssg_hip.so (include/ssg_hip.h).

The shared library is built in-tree by ``ssl_amd._lib.build()`` (hipcc,
--offload-arch=gfx950; cross-compiles without a GPU) and travels with the
source tree.  There is NO fallback: if the library is missing or a symbol is
absent, importing the compute path raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libssg_hip.so")
# the same sources with -DSSG_PROFILE: kernel-phase ablations + ssg_set_profile_mask (bench.py's per-kernel timings,
# tools/); the product library above has neither
PROF_SO_PATH = os.path.join(CSRC, "libssg_hip_prof.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "ssg_hip.h")

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/ssg_hip.h one to one
PROTOTYPES = {
    "ssg_abi_version": (_i, []),
    "ssg_status_string": (ctypes.c_char_p, [_i]),
    "ssg_compute_similarity": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ssg_compute_similarity_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ssg_edge_scratch_bytes": (_sz, [_i, _i, _i]),
    "ssg_forward_plan_bytes": (_sz, [_i, _i, _i, _i]),
    "ssg_set_dense_threshold": (_i, [_i]),
    "ssg_set_overlap": (_i, [_i]),
    "ssg_last_overlap_assignment": (_i, []),
    "ssg_set_tiny_step": (_i, [_i]),
    "ssg_edge_list": (_i, [_vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssg_edge_mask_laplacian": (_i, [_vp, _i, _i, _i, _f, _i, _vp, _vp]),
    "ssg_map_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp,
                             _vp, _vp]),
    "ssg_backward_scratch_bytes": (_sz, [_i, _i]),
    "ssg_map_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp,
                              _vp, _vp]),
    "ssg_grad_fix_bytes": (_sz, [_i, _i, _i, _i]),
    "ssg_loss_scratch_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ssg_loss_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _f, _f,
                               _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ssg_loss_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ssg_loss_rows_bytes": (_sz, [_i, _i]),
    "ssg_loss_tm_bytes": (_sz, [_i, _i]),
    "ssg_loss_workspace_layout": (_i, [_i, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_size_t)]),
    "ssg_filter2d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ssg_diffjpeg": (_i, [_vp, _vp, _i, _i, _i, _vp, _f, _vp]),
    "ssg_usm_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "ssg_usm_sharp": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _sz, _vp]),
    "ssg_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _f, _f, _i, _f, _i, _vp, _vp,
                              _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ssg_loss_step": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _f, _f, _i, _f, _i, _vp, _vp,
                           _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ssg_augment_crop": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ssg_pool_swap": (_i, [_vp, _vp, _sz, _vp, _i, _vp]),
    "ssg_resize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, ctypes.c_double, ctypes.c_double, _vp]),
    "ssg_clamp_round": (_i, [_vp, _vp, _sz, _i, _i, _vp]),
    "ssg_gaussian_noise": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ssg_poisson_scratch_bytes": (_sz, [_i]),
    "ssg_poisson_rates": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ssg_poisson_noise": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ssg_kernel_name": (ctypes.c_char_p, [_i, _i, _i]),
    # include/similarity.h: the reference operator's own (void, stream-less) interface
    "ssg_ref_compute_similarity": (None, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "ssg_ref_compute_similarity_backward": (None, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "ssg_last_status": (_i, []),
    "ssg_device_status": (_i, [_vp]),
    "ssg_set_operator_plan_threshold": (_i, [_i]),
    "ssg_operator_pool_trim": (_i, []),
    "ssg_criteria_scratch_bytes": (_sz, []),
    "ssg_criteria_sums": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "ssg_criteria_grad": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
}
PROF_PROTOTYPES = {"ssg_set_profile_mask": (_i, [_i]), "ssg_prof_occupancy": (_i, [_i]),
                   "ssg_prof_set_lds_poison": (_i, [_i, ctypes.c_uint])}   # libssg_hip_prof.so only
# C++-linkage symbols of include/similarity.h (Itanium mangling of the reference's declarations, similarity.h:2-23)
CXX_SYMBOLS = ("_Z19_compute_similarityPKfPKiPfiiiiii", "_Z28_compute_similarity_backwardPKfS0_PKiPfiiiiii")


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into csrc/libssg_hip.so (+ the profiling build libssg_hip_prof.so)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode:
        print(out.stdout)
    if out.returncode:
        raise RuntimeError("building libssg_hip.so failed (see output above)")
    return SO_PATH


_lib = None
_prof = None


def _load(path, prototypes):
    # torch bundles its own HIP runtime (libamdhip64 of its ROCm build).  It must be the one
    # already mapped when libssg_hip.so resolves its libamdhip64 dependency, otherwise the
    # process ends up with two runtimes and our launches see "no ROCm-capable device".
    import torch  # noqa: F401
    if torch.cuda.is_available():
        torch.cuda.init()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ssl_amd/csrc`).  ssl_amd has no CPU / PyTorch fallback for the SSG kernels.")
    L = ctypes.CDLL(path)
    for name, (res, args) in prototypes.items():
        fn = getattr(L, name)  # AttributeError if the symbol is absent: loud by design
        fn.restype = res
        fn.argtypes = args
    for name in CXX_SYMBOLS:
        getattr(L, name)
    return L


def lib():
    """The loaded library with typed prototypes.  Raises if it is not built."""
    global _lib
    if _lib is None:
        _lib = _load(SO_PATH, PROTOTYPES)
    return _lib


def lib_prof():
    """The PROFILING build (libssg_hip_prof.so): same entry points + ssg_set_profile_mask.  For timing tools only."""
    global _prof
    if _prof is None:
        _prof = _load(PROF_SO_PATH, dict(PROTOTYPES, **PROF_PROTOTYPES))
    return _prof


class profile_build:
    """`with _lib.profile_build() as L:` -- inside the block lib() returns the profiling build, so that the engine's
    host code (which calls lib()) runs on it; the product library is restored on exit."""

    def __enter__(self):
        global _lib
        self._saved = _lib
        _lib = lib_prof()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib.ssg_set_profile_mask(0)
        _lib = self._saved
        return False


def check(status):
    if status != 0:
        raise RuntimeError(f"libssg_hip: {lib().ssg_status_string(status).decode()} (status {status})")
