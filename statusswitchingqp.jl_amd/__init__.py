"""statusswitchingqp.jl_amd -- MI355X (gfx950) backend for the active-set inner
loop of PharosAbad/StatusSwitchingQP.jl, behind the reference's own surface:

    Status, IN, DN, UP, OE, EO, Settings, QP, solveQP

The directory name contains a dot, so it is loaded through
`__graft_entry__.load_package()` (importlib) under the module name `ssqp_amd`.
"""
from . import _capi, dist
from ._capi import LIB_PATH, NoDeviceError, SSQPError
from .solver import (BASE_SEED, CONFIGS, Context, DeviceBatch, GenConfig, ResidentBatch, default_context, generate_batch,
                     phase1_batch, solveQP, solveQP_batch, solveQP_batch_multi)
from .types import DN, EO, IN, OE, UP, QP, DimensionMismatch, Settings, Status
from .optimizer import (Optimizer, ResultStatus, TerminationStatus, UnsupportedConstraint, UnsupportedModel, get_constraints,
                        moi_to_qp)

__all__ = ["Status", "IN", "DN", "UP", "OE", "EO", "Settings", "QP", "solveQP", "solveQP_batch", "solveQP_batch_multi", "Context",
           "DeviceBatch", "ResidentBatch", "GenConfig", "CONFIGS", "BASE_SEED", "generate_batch", "phase1_batch", "NoDeviceError",
           "SSQPError", "DimensionMismatch", "LIB_PATH", "default_context", "Optimizer", "TerminationStatus", "ResultStatus",
           "moi_to_qp", "get_constraints", "UnsupportedConstraint", "UnsupportedModel"]
