"""ezkl_amd -- MI355X (gfx950) backend for the `ezkl prove` hot path: BN254 G1 MSM, Fr NTT, quotient sweep.

The product is `libezkl_hip.so` (C ABI in include/ezkl_hip.h).  This package is the thin Python host
mirror of the reference interface (ParamsKZG / EvaluationDomain / GraphEvaluator semantics) used by the
tests and bench; it never falls back to a CPU path: without the HIP library or a GPU every op raises."""
from .lib import EzklHipError, load, lib_path  # noqa: F401
from . import codecs  # noqa: F401
from .backend import (  # noqa: F401
    ParamsKZG, EvaluationDomain, DeviceBuffer, GraphProgram, msm_g1, ntt, vec_op, device_count, init,
)
