"""ezkl_amd -- MI355X (gfx950) backend for the `ezkl prove` hot path: BN254 G1 MSM, Fr NTT, quotient sweep.

The product is `libezkl_hip.so` (C ABI in include/ezkl_hip.h).  This package is the thin Python host
mirror of the reference interface (ParamsKZG / EvaluationDomain / GraphEvaluator semantics) used by the
tests and bench; it never falls back to a CPU path: without the HIP library or a GPU every op raises."""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before any HIP runtime initialises (torch included): see ctx_init in csrc/capi.hip
from .lib import EzklHipError, load, lib_path  # noqa: F401
from . import codecs  # noqa: F401


def enabled(k):
    """the runtime gate (ENABLE_HIP_GPU set and k > HIP_SMALL_K, default 8): the role of ENABLE_ICICLE_GPU / ICICLE_SMALL_K"""
    import ctypes
    return bool(load().ezkl_hip_enabled(ctypes.c_uint32(k)))


from .backend import (  # noqa: F401
    ParamsKZG, EvaluationDomain, DeviceBuffer, GraphProgram, msm_g1, ntt, vec_op, device_count, init,
)
