"""lisp-mcmc_amd: MI355X-native drop-in for the walker-adaptive-steps path of afranson/Lisp-MCMC.

The package directory name carries a hyphen (it is the name the build contract fixes);
import it as `import lisp_mcmc_amd` (alias module at the repository root) or with
importlib.import_module("lisp-mcmc_amd").

  _capi    ctypes binding of libmhx.so (the C ABI of include/mhx.h)
  engine   Engine: batched chains, one engine per GPU
  walker   the reference's own names: walker_create, walker_adaptive_steps, walker_get, ...
  models   model designators that stand in for the reference's :function closures
"""
from . import _capi as capi  # noqa: F401
from ._capi import MhxError  # noqa: F401
from .engine import Engine, Group, comm_unique_id, partition  # noqa: F401
from . import models  # noqa: F401
from . import sexpr  # noqa: F401
from . import distributed  # noqa: F401
from . import ingest  # noqa: F401
from .ingest import read_file_to_data, create_walker_data  # noqa: F401
from .saveload import walker_save, walker_load  # noqa: F401
from .walker import (  # noqa: F401
    Walker, WalkerStep, walker_create, mcmc_fit, walker_adaptive_steps,
    walker_adaptive_steps_full, walker_many_steps, walker_take_step, walker_get,
    walker_modify, prior_bounds, log_prior_flat, request_stop, create_log_liklihood_function,
)

__all__ = ["capi", "MhxError", "Engine", "Group", "comm_unique_id", "partition", "models", "Walker", "WalkerStep", "walker_create",
           "mcmc_fit", "walker_adaptive_steps", "walker_adaptive_steps_full",
           "walker_many_steps", "walker_take_step", "walker_get", "walker_modify",
           "prior_bounds", "log_prior_flat", "request_stop", "create_log_liklihood_function"]
