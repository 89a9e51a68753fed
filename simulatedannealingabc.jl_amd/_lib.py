"""ctypes binding of libsabc_hip.so (include/sabc_hip.h) and its in-tree build.

There is no CPU path: if the library is missing or no gfx950 device is usable, every
compute call raises.  torch is imported first on purpose -- it maps its own HIP runtime
(libamdhip64.so.7) and the library must bind to that one copy, not to a second one.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libsabc_hip.so")
HEADER = os.path.normpath(os.path.join(_HERE, "..", "include", "sabc_hip.h"))

ABI_VERSION = 6
P2P_DESC_BYTES, P2P_MAX_WORLD = 512, 8
MAX_PARA, MAX_STATS, MAX_MODEL_PARAMS = 16, 64, 32
MAX_SOURCE_STATS = 16     # simulators compiled from source (SABC_MODEL_USER)
MAX_JOINT_PARA = 8
MODEL_HOST, MODEL_GAUSS_IID, MODEL_GAUSS2D, MODEL_GK, MODEL_LV, MODEL_USER = 0, 1, 2, 3, 4, 5
PRIOR_NORMAL, PRIOR_UNIFORM, PRIOR_EXPONENTIAL, PRIOR_LOGNORMAL, PRIOR_GAMMA, PRIOR_BETA, PRIOR_TRUNCNORMAL = 0, 1, 2, 3, 4, 5, 6
PROP_RANDOMWALK, PROP_DIFFEVO, PROP_STRETCH = 0, 1, 2
ALG_SINGLE_EPS, ALG_MULTI_EPS = 0, 1
KERNEL_UPDATE, KERNEL_REDUCE, KERNEL_RESAMPLE, KERNEL_INIT, KERNEL_COLLECTIVE = 0, 1, 2, 3, 4

ERR_NAMES = {
    -1: "NSIM_TOO_SMALL", -2: "NEG_DISTANCE", -3: "BAD_V", -4: "BAD_DELTA", -5: "BAD_ALGORITHM", -6: "BAD_BETA",
    -7: "ZERO_MEAN_U", -8: "BAD_CONFIG", -9: "NOT_POSDEF", -10: "EMPTY_CDF", -11: "ROOT", -20: "NO_DEVICE",
    -21: "HIP", -22: "COMM", -23: "STATE", -24: "CALLBACK",
}


class SABCError(RuntimeError):
    """Counterpart of the reference's `error(...)` (ErrorException) sites."""

    def __init__(self, code, msg):
        super().__init__(f"{msg} [SABC_ERR_{ERR_NAMES.get(code, code)}]")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("n_particles", C.c_int64),
        ("n_para", C.c_int32), ("n_stats", C.c_int32), ("model_id", C.c_int32), ("n_model_params", C.c_int32),
        ("model_params", C.c_double * MAX_MODEL_PARAMS),
        ("prior_kind", C.c_int32 * MAX_PARA), ("prior_a", C.c_double * MAX_PARA), ("prior_b", C.c_double * MAX_PARA),
        ("prior_c", C.c_double * MAX_PARA), ("prior_d", C.c_double * MAX_PARA),
        ("prior_joint", C.c_int32), ("reserved2", C.c_int32), ("prior_chol", C.c_double * (MAX_PARA * MAX_PARA)),
        ("algorithm", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("reserved", C.c_int32),
        ("v", C.c_double), ("delta", C.c_double), ("seed", C.c_uint64),
    ]


class UpdateArgs(C.Structure):
    _fields_ = [
        ("n_simulation", C.c_int64), ("v", C.c_double), ("delta", C.c_double), ("resample", C.c_double),
        ("checkpoint_history", C.c_int64), ("proposal_kind", C.c_int32), ("more_chunks_follow", C.c_int32),
        ("proposal_p0", C.c_double), ("proposal_p1", C.c_double), ("history_phase", C.c_int64),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                           C.c_int32, C.c_void_p)
SIMULATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int64, C.c_uint64,
                          C.POINTER(C.c_double))
PRIOR_SAMPLE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double))
PRIOR_LOGPDF_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double))


def sources_newer_than_lib() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp", "Makefile"))]
    srcs.append(HEADER)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of every HIP translation unit (csrc/Makefile).  Safe when several processes ask at
    once (the ranks of a multi-process run start together): the staleness check happens UNDER the lock -- a rank that arrives
    while another is linking waits instead of loading a half-written file --, the linker writes a temporary name and the
    finished library is renamed into place, so a library some process has already mapped is never rewritten."""
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or sources_newer_than_lib():
            tmp = f"libsabc_hip.{os.getpid()}.tmp.so"
            cmd = ["make", "-C", CSRC, "-j4", f"OUT=../{tmp}"] + (["-B"] if force else [])
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if verbose or r.returncode:
                print(r.stdout)
            tmp_path = os.path.join(_HERE, tmp)
            if r.returncode:
                if os.path.exists(tmp_path):
                    os.remove(tmp_path)
                raise RuntimeError("building libsabc_hip.so failed")
            os.replace(tmp_path, LIB_PATH)
    return LIB_PATH


_lib = None


def lib():
    """Load libsabc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps the HIP runtime first; see module docstring)

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the SABC update loop."
        )
    L = bind(C.CDLL(LIB_PATH), strict=True)
    if L.sabc_abi_version() != ABI_VERSION:
        raise ImportError("libsabc_hip.so ABI version mismatch")
    _lib = L
    return L


def bind(L, strict=True):
    """Attach the include/sabc_hip.h signatures to a loaded library.  strict: every declared symbol
    must be there (the product library); the CPU engine harness of tests/ exports a subset."""
    dp, ip64 = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    vp = C.c_void_p
    sig = {
        "sabc_abi_version": ([], C.c_int),
        "sabc_last_global_error": ([], C.c_char_p),
        "sabc_device_count": ([], C.c_int),
        "sabc_create": ([C.POINTER(Config), C.POINTER(vp)], C.c_int),
        "sabc_destroy": ([vp], None),
        "sabc_last_error": ([vp], C.c_char_p),
        "sabc_set_stream": ([vp, vp], C.c_int),
        "sabc_set_collectives": ([vp, ALLREDUCE_FN, ALLGATHER_FN, vp, C.c_int], C.c_int),
        "sabc_set_alltoallv": ([vp, ALLTOALLV_FN], C.c_int),
        "sabc_comm_bytes": ([vp], C.c_int64),
        "sabc_set_host_simulator": ([vp, SIMULATE_FN, vp], C.c_int),
        "sabc_set_host_prior": ([vp, PRIOR_SAMPLE_FN, PRIOR_LOGPDF_FN, vp], C.c_int),
        "sabc_register_device_simulator": ([vp, C.c_char_p], C.c_int),
        "sabc_op_compile_device_simulator": ([C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int64], C.c_int),
        "sabc_op_compile_device_simulator_with_prior": ([C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int64], C.c_int),
        "sabc_comm_init_rccl": ([vp, vp], C.c_int),
        "sabc_comm_unique_id": ([vp], C.c_int),
        "sabc_comm_selftest": ([vp], C.c_int),
        "sabc_initialize": ([vp, C.c_int64], C.c_int),
        "sabc_update": ([vp, C.POINTER(UpdateArgs)], C.c_int),
        "sabc_n_global": ([vp], C.c_int64),
        "sabc_n_local": ([vp], C.c_int64),
        "sabc_local_offset": ([vp], C.c_int64),
        "sabc_get_population": ([vp, dp, dp, dp], C.c_int),
        "sabc_set_population": ([vp, dp, dp, dp], C.c_int),
        "sabc_get_counters": ([vp, ip64], C.c_int),
        "sabc_set_counters": ([vp, ip64], C.c_int),
        "sabc_get_epsilon": ([vp, dp, C.POINTER(C.c_int32)], C.c_int),
        "sabc_set_epsilon": ([vp, dp, C.c_int32], C.c_int),
        "sabc_history_len": ([vp], C.c_int64),
        "sabc_get_history": ([vp, dp, dp, dp], C.c_int),
        "sabc_clear_history": ([vp], C.c_int),
        "sabc_cdf_len": ([vp, C.c_int32], C.c_int64),
        "sabc_get_cdf_knots": ([vp, C.c_int32, dp], C.c_int),
        "sabc_set_cdf_knots": ([vp, C.c_int32, dp, C.c_int64], C.c_int),
        "sabc_cdf_apply": ([vp, dp, C.c_int64, dp], C.c_int),
        "sabc_get_proposal_sigma": ([vp, dp], C.c_int),
        "sabc_last_ess": ([vp], C.c_double),
        "sabc_op_build_cdf": ([C.c_int32, dp, C.c_int64, dp, ip64], C.c_int),
        "sabc_op_sort": ([C.c_int32, dp, C.c_int64, dp], C.c_int),
        "sabc_op_cdf_eval": ([C.c_int32, dp, C.c_int64, dp, C.c_int64, dp], C.c_int),
        "sabc_op_eps_single": ([C.c_double, C.c_double, dp], C.c_int),
        "sabc_op_eps_multi": ([dp, C.c_int32, C.c_double, dp], C.c_int),
        "sabc_op_simulate": ([vp, dp, C.c_int64, C.c_uint64, C.c_uint64, dp], C.c_int),
        "sabc_op_prior": ([vp, C.c_uint64, C.c_int64, dp, dp], C.c_int),
        "sabc_op_philox": ([C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32,
                            C.POINTER(C.c_uint32), dp], C.c_int),
        "sabc_op_normal_pairs": ([C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int64, dp],
                                 C.c_int),
        "sabc_op_rng_peak": ([C.c_int32, C.c_int64, C.c_int32, C.c_int32, dp], C.c_int),
        "sabc_profile_enable": ([vp, C.c_int32], C.c_int),
        "sabc_profile_get": ([vp, C.c_int32, dp, ip64], C.c_int),
        "sabc_profile_noops": ([vp, C.c_int32], C.c_int64),
        "sabc_set_host_chunk": ([vp, C.c_int64], C.c_int),
        "sabc_host_callback_seconds": ([vp], C.c_double),
        "sabc_host_callback_calls": ([vp], C.c_int64),
        "sabc_host_syncs": ([vp], C.c_int64),
        "sabc_kernel_launches": ([vp], C.c_int64),
        "sabc_persistent_launches": ([vp], C.c_int64),
        "sabc_persistent_lanes": ([vp], C.c_int32),
        "sabc_persistent_fallbacks": ([vp], C.c_int64),
        "sabc_collective_calls": ([vp], C.c_int64),
        "sabc_comm_p2p_setup": ([vp], C.c_int),
        "sabc_comm_p2p_descriptor": ([vp, vp], C.c_int),
        "sabc_comm_p2p_init": ([vp, vp], C.c_int),
        "sabc_comm_p2p_selftest": ([vp], C.c_int),
        "sabc_comm_p2p_set_timeout": ([vp, C.c_double], C.c_int),
        "sabc_comm_p2p_disable": ([vp], C.c_int),
        "sabc_comm_p2p_active": ([vp], C.c_int),
        "sabc_comm_p2p_fallbacks": ([vp], C.c_int64),
        "sabc_comm_p2p_inject_silence": ([vp, C.c_int32], C.c_int),
        "sabc_comm_p2p_inject_stale": ([vp, C.c_int32], C.c_int),
        "sabc_comm_p2p_inject_loss": ([vp, C.c_int32], C.c_int),
        "sabc_comm_p2p_set_destroy_wait": ([vp, C.c_double], C.c_int),
        "sabc_comm_p2p_parked_bytes": ([], C.c_int64),
    }
    for name, (args, res) in sig.items():
        if not strict and not hasattr(L, name):
            continue
        fn = getattr(L, name)   # AttributeError here = a symbol the header declares is missing
        fn.argtypes, fn.restype = args, res
    return L


def global_error() -> str:
    return lib().sabc_last_global_error().decode("utf-8", "replace")
