"""osmo_trx_amd -- MI355X (gfx950) receive-side burst DSP for osmo-trx.

The product is the HIP library osmo_trx_amd/lib/libtrxhip.so behind the C ABI of include/trxhip.h
(kernels: csrc/trx_kernels.hip, csrc/trx_aux_kernels.hip) plus the C++ host shim that keeps the
reference's sigProcLib.h signatures (host/).  This Python package is plumbing only: a ctypes
binding that hands torch device pointers to the C ABI (trxhip.py), the synthetic workload
generator (synth.py), batch sharding + RCCL table broadcast (shard.py) and the build driver
(build.py).  There is no CPU fallback: without the built library or without a GPU every
compute entry point raises.
"""
from .trxhip import (TrxHip, TrxHipError, PARAMS_DTYPE, RESULT_DTYPE, lib_path, load_library,  # noqa: F401
                     OFF, TSC, EXT_RACH, RACH, SCH, EDGE, IDLE)

__all__ = ["TrxHip", "TrxHipError", "PARAMS_DTYPE", "RESULT_DTYPE", "lib_path", "load_library",
           "OFF", "TSC", "EXT_RACH", "RACH", "SCH", "EDGE", "IDLE"]
