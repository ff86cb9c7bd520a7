from .basic_loss import KLDistanceLoss, L1Loss, SSGLoss, set_native_criteria  # noqa: F401
from .loss_util import similarity_map  # noqa: F401
from .lazy import LazySSG, lazy_enabled, set_lazy  # noqa: F401
