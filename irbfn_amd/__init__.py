"""irbfn_amd -- MI355X-native (gfx950) implementation of the IRBFN hot path of hzheng40/irbfn.

Host side is Python (like the reference) over a C-ABI shared library of hand-written HIP kernels
(``libirbfn_hip.so``, ``include/irbfn_hip.h``).  The module names mirror the reference's:

* ``irbfn_amd.flax_rbf``      basis-function tokens (``gaussian``, ``inverse_quadratic`` ...)
* ``irbfn_amd.model``         ``WCRBFNet(**model_card).apply(params, x)``
* ``irbfn_amd.dynamics``      ``integrate_st_mult``, ``dynamic_st_onestep_aux``, ``integrate_frenet_mult``
* ``irbfn_amd.planner_utils`` ``integrate_path_mult``
* ``irbfn_amd.autograd``      ``torch.autograd`` wrappers (the ``jax.grad`` surface)
* ``irbfn_amd.distributed``   one-process-per-GPU sharding, RCCL broadcast of the parameters

There is no CPU fallback: every entry point raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"

from . import flax_rbf  # noqa: F401
from .model import WCRBFNet, pred_step  # noqa: F401
