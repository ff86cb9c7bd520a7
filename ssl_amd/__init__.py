"""ssl_amd -- MI355X-native Self-Similarity-Graph (SSG) loss engine.

Drop-in for the SSG loss hot path of ChrisDud0257/SSL: `similarity_map`,
`compute_similarity`, `L1Loss`, `KLDistanceLoss` keep the reference's API;
`SSGLoss` is the batched replacement of the callers' per-image loop.  The
numbers come from hand-written HIP kernels (ssl_amd/csrc) behind the C ABI in
include/ssg_hip.h.  Importing this package does not need a GPU; computing does.
"""
__all__ = ["similarity_map", "compute_similarity", "L1Loss", "KLDistanceLoss", "SSGLoss", "engine", "synth"]


def __getattr__(name):
    # lazy: `import ssl_amd.synth` must work without torch side effects
    if name in ("similarity_map", "L1Loss", "KLDistanceLoss", "SSGLoss"):
        from . import losses
        return getattr(losses, name)
    if name == "compute_similarity":
        from .losses.similarity.similaritywrapper import compute_similarity
        return compute_similarity
    if name in ("engine", "synth", "_lib"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
