"""CPU harness for the product's host engine (TEST INFRASTRUCTURE; see ref_backend.cpp).
Builds tests/cpu_engine/libsabc_cpu_engine.so with g++ from the product's engine.cpp +
control.hpp and a Backend made of oracle calls, and exposes it through the same Python handle
class the product uses."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_HERE))
LIB = os.path.join(_HERE, "libsabc_cpu_engine.so")
_SRCS = [
    os.path.join(_HERE, "ref_backend.cpp"),
    os.path.join(ROOT, "simulatedannealingabc.jl_amd", "csrc", "engine.cpp"),
    os.path.join(ROOT, "oracle", "sabc_oracle.c"),
]
_DEPS = _SRCS + [os.path.join(ROOT, "simulatedannealingabc.jl_amd", "csrc", f)
                 for f in ("engine.hpp", "control.hpp", "host_math.hpp", "sabc_types.hpp", "p2p.hpp", "p2p_setup.hpp")] + \
        [os.path.join(ROOT, "include", "sabc_hip.h"), os.path.join(ROOT, "oracle", "sabc_oracle.h")]


def build():
    """Compile the harness if it is older than its sources.  Several processes may ask at once (the ranks of a multi-process
    test start together): one builds -- into a temporary file, renamed into place -- the others wait on the lock."""
    import fcntl

    def fresh():
        return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(p) for p in _DEPS)
    if fresh():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if fresh():
            return LIB
        obj, tmp = os.path.join(_HERE, f"sabc_oracle.{os.getpid()}.o"), LIB + f".{os.getpid()}.tmp"
        try:
            subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-c", _SRCS[2], "-o", obj])
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-ffp-contract=off", "-pthread",
                                   "-o", tmp, _SRCS[0], _SRCS[1], obj, "-lm"])
            os.replace(tmp, LIB)
        finally:
            for f in (obj, tmp):
                if os.path.exists(f):
                    os.remove(f)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        import sabc_amd
        _lib = sabc_amd._lib.bind(C.CDLL(build()), strict=False)
    return _lib


def handle_class():
    import sabc_amd

    class CpuEngineHandle(sabc_amd.SabcHandle):
        """Same host engine, oracle-backed kernels; for tests only."""

        def _load_library(self):
            return lib()

    return CpuEngineHandle
