"""ctypes binding of the CPU oracle (oracle/sabc_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  PARITY UNPINNED at
bit level (see sabc_oracle.h): the reference is Julia and cannot run here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsabc_oracle.so")

MAX_PARA, MAX_STATS, MAX_MODEL_PARAMS = 16, 64, 32
MODEL_HOST, MODEL_GAUSS_IID, MODEL_GAUSS2D, MODEL_GK, MODEL_LV = 0, 1, 2, 3, 4
SIMULATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int64, C.c_uint64,
                          C.POINTER(C.c_double))
PRIOR_NORMAL, PRIOR_UNIFORM, PRIOR_EXPONENTIAL, PRIOR_LOGNORMAL, PRIOR_GAMMA, PRIOR_BETA, PRIOR_TRUNCNORMAL = 0, 1, 2, 3, 4, 5, 6
PROP_RANDOMWALK, PROP_DIFFEVO, PROP_STRETCH = 0, 1, 2
ALG_SINGLE_EPS, ALG_MULTI_EPS = 0, 1
PURPOSE_PRIOR, PURPOSE_SIM, PURPOSE_PROP, PURPOSE_PROP2, PURPOSE_ACCEPT, PURPOSE_RESAMPLE = range(6)

ERRORS = {
    -1: "NSIM_TOO_SMALL", -2: "NEG_DISTANCE", -3: "BAD_V", -4: "BAD_DELTA", -5: "BAD_ALGORITHM",
    -6: "BAD_BETA", -7: "ZERO_MEAN_U", -8: "BAD_CONFIG", -9: "NOT_POSDEF", -10: "EMPTY_CDF", -11: "ROOT",
}


class OracleError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"oracle error {code} ({ERRORS.get(code, '?')}): {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("n_particles", C.c_int64),
        ("n_para", C.c_int32),
        ("n_stats", C.c_int32),
        ("model_id", C.c_int32),
        ("n_model_params", C.c_int32),
        ("model_params", C.c_double * MAX_MODEL_PARAMS),
        ("prior_kind", C.c_int32 * MAX_PARA),
        ("prior_a", C.c_double * MAX_PARA),
        ("prior_b", C.c_double * MAX_PARA),
        ("prior_c", C.c_double * MAX_PARA),
        ("prior_d", C.c_double * MAX_PARA),
        ("prior_joint", C.c_int32),
        ("_pad2", C.c_int32),
        ("prior_chol", C.c_double * (MAX_PARA * MAX_PARA)),
        ("algorithm", C.c_int32),
        ("_pad", C.c_int32),
        ("v", C.c_double),
        ("delta", C.c_double),
        ("seed", C.c_uint64),
        ("host_fn", SIMULATE_FN),
        ("host_ctx", C.c_void_p),
    ]


class UpdateArgs(C.Structure):
    _fields_ = [
        ("n_simulation", C.c_int64),
        ("v", C.c_double),
        ("delta", C.c_double),
        ("resample", C.c_double),
        ("checkpoint_history", C.c_int64),
        ("proposal_kind", C.c_int32),
        ("_pad", C.c_int32),
        ("proposal_p0", C.c_double),
        ("proposal_p1", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Safe when several processes ask at once: the staleness check happens
    under the lock, the compiler writes a temporary name and the finished library is renamed into place (a library some
    process has already mapped is never rewritten)."""
    import fcntl
    src = os.path.join(_HERE, "sabc_oracle.c")

    def stale():
        return not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "sabc_oracle.h")))
    with open(_LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or stale():
            tmp = os.path.join(_HERE, f"libsabc_oracle.{os.getpid()}.tmp.so")
            try:
                subprocess.check_call(["make", "-C", _HERE, "-B", "libsabc_oracle.so", f"OUT={os.path.basename(tmp)}"],
                                      stdout=subprocess.DEVNULL)
                os.replace(tmp, _LIB_PATH)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.orc_create.restype = C.c_int
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_initialize.argtypes = [C.c_void_p, C.c_int64]
        L.orc_update.argtypes = [C.c_void_p, C.POINTER(UpdateArgs)]
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        for f in ("orc_theta", "orc_u", "orc_rho"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = dp
        L.orc_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_epsilon.argtypes = [C.c_void_p, dp]
        L.orc_history_len.argtypes = [C.c_void_p]
        L.orc_history_len.restype = C.c_int64
        L.orc_history.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_cdf_len.argtypes = [C.c_void_p, C.c_int]
        L.orc_cdf_len.restype = C.c_int64
        L.orc_cdf_knots.argtypes = [C.c_void_p, C.c_int]
        L.orc_cdf_knots.restype = dp
        L.orc_last_ess.argtypes = [C.c_void_p]
        L.orc_last_ess.restype = C.c_double
        L.orc_proposal_sigma.argtypes = [C.c_void_p, dp]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_literal.argtypes = [C.c_int]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_stream_block.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_u52.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_u52.restype = C.c_double
        L.orc_normal_pair.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, dp]
        L.orc_build_cdf.argtypes = [dp, C.c_int64, dp]
        L.orc_build_cdf.restype = C.c_int64
        L.orc_cdf_apply.argtypes = [dp, C.c_int64, C.c_double]
        L.orc_cdf_apply.restype = C.c_double
        L.orc_eps_single.argtypes = [C.c_double, C.c_double]
        L.orc_eps_single.restype = C.c_double
        L.orc_eps_multi.argtypes = [dp, C.c_int, C.c_double, dp]
        L.orc_multi_eps_beta.argtypes = [C.c_double]
        L.orc_multi_eps_beta.restype = C.c_double
        L.orc_prior_logpdf.argtypes = [C.POINTER(Config), dp]
        L.orc_prior_logpdf.restype = C.c_double
        L.orc_prior_sample.argtypes = [C.POINTER(Config), C.c_uint64, dp]
        L.orc_simulate.argtypes = [C.POINTER(Config), dp, C.c_uint64, C.c_uint64, dp]
        L.orc_cholesky.argtypes = [dp, C.c_int, dp]
        L.orc_scan_chunks.argtypes = [C.c_int64]
        L.orc_scan_chunks.restype = C.c_int64
        L.orc_weight_scan.argtypes = [dp, C.c_int64, dp, dp, dp]
        L.orc_weight_scan.restype = None
        L.orc_resample_index.argtypes = [dp, dp, C.c_int64, C.c_double]
        L.orc_resample_index.restype = C.c_int64
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def host_simulator(fn, d, s):
    """Wrap fn(theta[d], particle_id, iter) -> rho[s] as the oracle's ORC_MODEL_HOST callback."""
    def cb(ctx, theta, ids, m, it, rho_out):
        try:
            for i in range(m):
                r = np.atleast_1d(np.asarray(fn(np.array([theta[k * m + i] for k in range(d)]), int(ids[i]), int(it)), dtype=float))
                for j in range(s):
                    rho_out[j * m + i] = r[j]
            return 0
        except Exception as e:   # never raise through the C frame
            print(f"[oracle] host simulator failed: {e!r}", flush=True)
            return -1
    return SIMULATE_FN(cb)


def make_config(*, n_particles, n_para, n_stats, model_id, model_params, prior, algorithm=ALG_SINGLE_EPS,
                v=1.0, delta=0.1, seed=20241220, host_fn=None, prior_chol=None) -> Config:
    """prior: list of (kind, a, b) or (kind, a, b, c, d) per dimension."""
    cfg = Config()
    cfg.n_particles, cfg.n_para, cfg.n_stats = int(n_particles), int(n_para), int(n_stats)
    cfg.model_id, cfg.n_model_params = int(model_id), len(model_params)
    for i, p in enumerate(model_params):
        cfg.model_params[i] = float(p)
    assert len(prior) == n_para
    for k, desc in enumerate(prior):
        kind, a, b, c, d = (tuple(desc) + (0.0, 0.0))[:5]
        cfg.prior_kind[k], cfg.prior_a[k], cfg.prior_b[k] = int(kind), float(a), float(b)
        cfg.prior_c[k], cfg.prior_d[k] = float(c), float(d)
    cfg.algorithm, cfg.v, cfg.delta, cfg.seed = int(algorithm), float(v), float(delta), int(seed)
    if prior_chol is not None:      # MvNormal(mu = the a's of `prior`, Sigma = L L'), L lower triangular d x d
        L = np.asarray(prior_chol, dtype=np.float64).reshape(n_para, n_para)
        cfg.prior_joint = 1
        for k in range(n_para):
            for l in range(n_para):
                cfg.prior_chol[k * n_para + l] = float(L[k, l]) if l <= k else 0.0
    if host_fn is not None:
        cfg.host_fn = host_fn      # keep a reference to the CFUNCTYPE object alive in the caller
    return cfg


def make_update_args(*, n_simulation, proposal=(PROP_DIFFEVO, None, 1e-5), n_para=1, n_particles=100, v=1.0,
                     delta=0.1, resample=None, checkpoint_history=1) -> UpdateArgs:
    a = UpdateArgs()
    a.n_simulation, a.v, a.delta = int(n_simulation), float(v), float(delta)
    a.resample = float(2 * n_particles if resample is None else resample)
    a.checkpoint_history = int(checkpoint_history)
    kind, p0, p1 = proposal
    if kind == PROP_DIFFEVO and p0 is None:
        p0 = 2.38 / np.sqrt(2 * n_para)         # proposals.jl:93
    a.proposal_kind, a.proposal_p0, a.proposal_p1 = int(kind), float(p0), float(p1 if p1 is not None else 0.0)
    return a


@dataclass
class OracleRun:
    """Owns one orc_state; mirrors SABCresult/SABCstate field names where it can."""
    cfg: Config
    _h: C.c_void_p = field(default=None, repr=False)

    def __post_init__(self):
        h = C.c_void_p()
        rc = lib().orc_create(C.byref(self.cfg), C.byref(h))
        if rc:
            raise OracleError(rc, "orc_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def _check(self, rc):
        if rc:
            raise OracleError(rc, lib().orc_last_error(self._h).decode("utf-8", "replace"))

    def initialize(self, n_simulation):
        self._check(lib().orc_initialize(self._h, int(n_simulation)))

    def update(self, args: UpdateArgs):
        self._check(lib().orc_update(self._h, C.byref(args)))

    # ---- state ----
    def _arr(self, fn, rows):
        n = self.cfg.n_particles
        p = fn(self._h)
        return np.ctypeslib.as_array(p, shape=(rows, n)).copy()

    @property
    def theta(self):  # [d][n]
        return self._arr(lib().orc_theta, self.cfg.n_para)

    @property
    def u(self):
        return self._arr(lib().orc_u, self.cfg.n_stats)

    @property
    def rho(self):
        return self._arr(lib().orc_rho, self.cfg.n_stats)

    @property
    def counters(self):
        out = (C.c_int64 * 4)()
        lib().orc_counters(self._h, out)
        return dict(n_simulation=out[0], n_accept=out[1], n_resampling=out[2], n_population_updates=out[3])

    @property
    def eps(self):
        out = np.zeros(MAX_STATS)
        k = lib().orc_epsilon(self._h, _dp(out))
        return out[:k].copy()

    @property
    def history(self):
        m = lib().orc_history_len(self._h)
        le = 1 if self.cfg.algorithm == ALG_SINGLE_EPS else self.cfg.n_stats
        e, u, r = np.zeros((m, le)), np.zeros((m, self.cfg.n_stats)), np.zeros((m, self.cfg.n_stats))
        if m:
            lib().orc_history(self._h, _dp(e), _dp(u), _dp(r))
        return e, u, r

    def cdf_knots(self, j):
        m = lib().orc_cdf_len(self._h, j)
        return np.ctypeslib.as_array(lib().orc_cdf_knots(self._h, j), shape=(m,)).copy()

    @property
    def ess(self):
        return lib().orc_last_ess(self._h)

    @property
    def sigma(self):
        d = self.cfg.n_para
        out = np.zeros((d, d))
        lib().orc_proposal_sigma(self._h, _dp(out))
        return out


# ---- unit helpers ----
def philox(key, ctr):
    k = (C.c_uint32 * 2)(*key)
    c = (C.c_uint32 * 4)(*ctr)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(k, c, o)
    return [int(x) for x in o]


def stream_block(seed, pid, purpose, it, k):
    o = (C.c_uint32 * 4)()
    lib().orc_stream_block(seed, pid, purpose, it, k, o)
    return [int(x) for x in o]


def u52(hi, lo):
    L = lib()
    L.orc_u52.argtypes, L.orc_u52.restype = [C.c_uint32, C.c_uint32], C.c_double
    return float(L.orc_u52(hi, lo))


def normal_pair(seed, pid, purpose, it, k):
    z = np.zeros(2)
    lib().orc_normal_pair(seed, pid, purpose, it, k, _dp(z))
    return z


def build_cdf(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    kn = np.zeros(len(x) + 2)
    m = lib().orc_build_cdf(_dp(x), len(x), _dp(kn))
    if m < 0:
        raise OracleError(int(m), "build_cdf")
    return kn[:m].copy()


def cdf_apply(knots, x):
    knots = np.ascontiguousarray(knots, dtype=np.float64)
    xs = np.atleast_1d(np.asarray(x, dtype=np.float64))
    out = np.array([lib().orc_cdf_apply(_dp(knots), len(knots), float(v)) for v in xs])
    return out if np.ndim(x) else float(out[0])


def eps_single(ubar, v):
    return lib().orc_eps_single(float(ubar), float(v))


def eps_multi(ubar, v):
    ubar = np.ascontiguousarray(ubar, dtype=np.float64)
    out = np.zeros(len(ubar))
    rc = lib().orc_eps_multi(_dp(ubar), len(ubar), float(v), _dp(out))
    if rc:
        raise OracleError(rc, "eps_multi")
    return out


def weight_scan(w):
    """(cum, chunk offsets, (sum w, sum w^2)) in the specified blocked order (sabc_oracle.c: orc_weight_scan)."""
    w = np.ascontiguousarray(w, dtype=np.float64)
    n = len(w)
    cum, bs, tot = np.zeros(n), np.zeros(lib().orc_scan_chunks(n)), np.zeros(2)
    lib().orc_weight_scan(_dp(w), n, _dp(cum), _dp(bs), _dp(tot))
    return cum, bs, tot


def resample_index(cum, bs, t):
    return int(lib().orc_resample_index(_dp(cum), _dp(bs), len(cum), float(t)))


def simulate(cfg: Config, theta, pid, it):
    th = np.ascontiguousarray(theta, dtype=np.float64)
    out = np.zeros(cfg.n_stats)
    rc = lib().orc_simulate(C.byref(cfg), _dp(th), int(pid), int(it), _dp(out))
    if rc:
        raise OracleError(rc, "simulate")
    return out


def prior_logpdf(cfg: Config, theta):
    th = np.ascontiguousarray(theta, dtype=np.float64)
    return lib().orc_prior_logpdf(C.byref(cfg), _dp(th))


def prior_sample(cfg: Config, pid):
    out = np.zeros(cfg.n_para)
    lib().orc_prior_sample(C.byref(cfg), int(pid), _dp(out))
    return out


def set_threads(n):
    lib().orc_set_threads(int(n))


def set_literal(on):
    """Literal expressions instead of the rewritten ones (see sabc_oracle.c: orc_set_literal)."""
    lib().orc_set_literal(int(bool(on)))
