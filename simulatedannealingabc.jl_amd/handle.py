"""SabcHandle: a thin Python owner of one `sabc_handle*` (the same calls a Julia wrapper makes
with `ccall`).  All state lives on the GPU behind the handle; arrays cross only on request."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, SABCError, UpdateArgs


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class SabcHandle:
    def __init__(self, *, n_particles, model, prior, algorithm=_lib.ALG_SINGLE_EPS, v=1.0, delta=0.1,
                 seed=20241220, device=0, rank=0, world=1):
        """model: object with model_id / params / n_stats; prior: distributions.Prior."""
        cfg = Config()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.device = int(device)
        cfg.n_particles = int(n_particles)
        cfg.n_para = len(prior)
        cfg.n_stats = int(model.n_stats)
        cfg.model_id = int(model.model_id)
        params = list(model.params)
        cfg.n_model_params = len(params)
        for i, p in enumerate(params):
            cfg.model_params[i] = float(p)
        for k, desc in enumerate(prior.descriptors()):
            kind, a, b, c, d = (tuple(desc) + (0.0, 0.0))[:5]
            cfg.prior_kind[k], cfg.prior_a[k], cfg.prior_b[k] = int(kind), float(a), float(b)
            cfg.prior_c[k], cfg.prior_d[k] = float(c), float(d)
        L = getattr(prior, "chol", None)          # MvNormal: lower Cholesky factor of Sigma
        if L is not None:
            d = len(prior)
            cfg.prior_joint = 1
            for k in range(d):
                for l in range(k + 1):
                    cfg.prior_chol[k * d + l] = float(L[k, l])
        if getattr(prior, "host_prior", False):     # any prior, as host callbacks (sabc_set_host_prior)
            cfg.prior_joint = 2
        if getattr(prior, "source_prior", False):   # any prior, as device code in the simulator's HIP source
            cfg.prior_joint = 3
        cfg.algorithm = int(algorithm)
        cfg.rank, cfg.world = int(rank), int(world)
        cfg.v, cfg.delta, cfg.seed = float(v), float(delta), int(seed)
        self.cfg = cfg
        self.d, self.s = cfg.n_para, cfg.n_stats
        self._L = self._load_library()
        h = C.c_void_p()
        rc = self._L.sabc_create(C.byref(cfg), C.byref(h))
        if rc:
            raise SABCError(rc, self._L.sabc_last_global_error().decode("utf-8", "replace"))
        self._h = h
        self._keep = []   # ctypes callbacks must outlive the handle
        self._host_model = None
        if getattr(model, "model_id", None) == _lib.MODEL_USER:
            rc = self._L.sabc_register_device_simulator(self._h, model.source.encode())
            if rc:
                msg = self._L.sabc_last_error(self._h).decode("utf-8", "replace")
                self.close()
                raise SABCError(rc, msg)
        self._host_prior = None
        if getattr(prior, "host_prior", False):
            cbs = prior.callbacks()
            self._keep.extend(cbs)
            self._host_prior = prior
            rc = self._L.sabc_set_host_prior(self._h, cbs[0], cbs[1], None)
            if rc:
                raise SABCError(rc, self._L.sabc_last_error(self._h).decode("utf-8", "replace"))
        if getattr(model, "model_id", None) == _lib.MODEL_HOST:
            cb = model.callback()
            self._keep.append(cb)
            self._host_model = model
            rc = self._L.sabc_set_host_simulator(self._h, cb, None)
            if rc:
                raise SABCError(rc, self._L.sabc_last_error(self._h).decode("utf-8", "replace"))

    def _load_library(self):
        return _lib.lib()          # libsabc_hip.so; raises if it is missing (no fallback)

    # ---- lifetime ----
    def close(self):
        """sabc_destroy.  On the peer-to-peer transport the shard LEAVES its group first (the peers are told, nothing of theirs
        stays mapped) and what they had mapped is freed once they have acknowledged -- bounded; otherwise it is parked until
        the process exits (include/sabc_hip.h, LEAVING).  No barrier with the other ranks is needed."""
        if getattr(self, "_h", None):
            self._L.sabc_destroy(self._h)
            self._h = None

    def __del__(self):
        # a finalizer runs at an arbitrary time on each rank and must not wait for peers: what a peer has not released by now
        # is parked at once
        try:
            if getattr(self, "_h", None) and hasattr(self._L, "sabc_comm_p2p_set_destroy_wait"):
                self._L.sabc_comm_p2p_set_destroy_wait(self._h, 0.0)
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc and self._host_prior is not None and self._host_prior.error is not None:
            err, self._host_prior.error = self._host_prior.error, None
            raise err                     # the exception the prior's sample / logpdf raised inside the host callback
        if rc and self._host_model is not None and self._host_model.error is not None:
            err, self._host_model.error = self._host_model.error, None
            raise err                     # the exception f_dist raised inside the host callback
        if rc:
            raise SABCError(rc, self._L.sabc_last_error(self._h).decode("utf-8", "replace"))

    # ---- hot path ----
    def initialize(self, n_simulation):
        self._check(self._L.sabc_initialize(self._h, int(n_simulation)))

    def update(self, *, n_simulation, proposal, v=1.0, delta=0.1, resample=None, checkpoint_history=1, history_phase=0,
               more_chunks_follow=False):
        """history_phase / more_chunks_follow: this call is one chunk of an update_population! call the wrapper has cut up for
        its progress lines -- population updates already done by the earlier chunks; whether another chunk follows (then the
        final history push of :378-382 is not this call's)."""
        a = UpdateArgs()
        a.n_simulation, a.v, a.delta = int(n_simulation), float(v), float(delta)
        a.resample = float(2 * self.cfg.n_particles if resample is None else resample)
        a.checkpoint_history = int(checkpoint_history)
        a.history_phase, a.more_chunks_follow = int(history_phase), int(bool(more_chunks_follow))
        a.proposal_kind, a.proposal_p0, a.proposal_p1 = proposal.descriptor()
        self._check(self._L.sabc_update(self._h, C.byref(a)))

    def set_stream(self, hip_stream: int):
        self._check(self._L.sabc_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_collectives(self, allreduce, allgather, device_buffers: bool):
        ar, ag = _lib.ALLREDUCE_FN(allreduce), _lib.ALLGATHER_FN(allgather)
        self._keep += [ar, ag]
        self._check(self._L.sabc_set_collectives(self._h, ar, ag, None, int(device_buffers)))

    def set_alltoallv(self, alltoallv):
        fn = _lib.ALLTOALLV_FN(alltoallv)
        self._keep.append(fn)
        self._check(self._L.sabc_set_alltoallv(self._h, fn))

    @property
    def comm_bytes(self):
        """Bytes that landed in this shard's receive buffers through the collectives so far."""
        return int(self._L.sabc_comm_bytes(self._h))

    def comm_init_rccl(self, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self._L.sabc_comm_init_rccl(self._h, C.cast(buf, C.c_void_p)))

    def comm_selftest(self):
        self._check(self._L.sabc_comm_selftest(self._h))

    # ---- peer-to-peer transport: the shards of one node exchange through each other's HBM (include/sabc_hip.h) ----
    def p2p_descriptor(self) -> bytes:
        """What the other shards need to map this shard's slot area, populations and rho (IPC handles + pid + pointers)."""
        buf = C.create_string_buffer(_lib.P2P_DESC_BYTES)
        self._check(self._L.sabc_comm_p2p_descriptor(self._h, C.cast(buf, C.c_void_p)))
        return buf.raw

    def p2p_init(self, all_descriptors=None):
        """all_descriptors: the descriptors of all shards in rank order (bytes, or a list of bytes); None lets the library
        exchange them over the collectives already installed."""
        if all_descriptors is None:
            self._check(self._L.sabc_comm_p2p_init(self._h, None))
            return
        blob = b"".join(all_descriptors) if not isinstance(all_descriptors, (bytes, bytearray)) else bytes(all_descriptors)
        if len(blob) != self.cfg.world * _lib.P2P_DESC_BYTES:
            raise ValueError("one descriptor per shard, in rank order")
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self._L.sabc_comm_p2p_init(self._h, C.cast(buf, C.c_void_p)))

    def p2p_selftest(self):
        self._check(self._L.sabc_comm_p2p_selftest(self._h))

    def p2p_setup(self) -> bool:
        """The whole set-up in one collective call (sabc_comm_p2p_setup): descriptors over the installed collectives -> map
        -> agreement -> self-test -> agreement.  True: every shard now runs peer to peer; False: every shard stays on the
        collectives (`p2p_setup_note` says why)."""
        rc = self._L.sabc_comm_p2p_setup(self._h)
        if rc < 0:
            self._check(rc)
        self.p2p_setup_note = self._L.sabc_last_error(self._h).decode("utf-8", "replace") if rc == 0 else ""
        return rc == 1

    def p2p_set_destroy_wait(self, milliseconds):
        self._check(self._L.sabc_comm_p2p_set_destroy_wait(self._h, float(milliseconds)))

    def p2p_parked_bytes(self):
        return int(self._L.sabc_comm_p2p_parked_bytes())

    def p2p_inject_loss(self, n=0):
        """Test hook: n more posts go out, then one reaches only this shard's own slots (lost on the wire)."""
        self._check(self._L.sabc_comm_p2p_inject_loss(self._h, int(n)))

    def p2p_inject_stale(self, n=1):
        """Test hook: this shard's next self-test reads its peers' memory as if a stale line had been served."""
        self._check(self._L.sabc_comm_p2p_inject_stale(self._h, int(n)))

    def p2p_set_timeout(self, milliseconds):
        self._check(self._L.sabc_comm_p2p_set_timeout(self._h, float(milliseconds)))

    def p2p_disable(self):
        self._check(self._L.sabc_comm_p2p_disable(self._h))

    @property
    def p2p_active(self):
        return bool(self._L.sabc_comm_p2p_active(self._h))

    @property
    def p2p_fallbacks(self):
        """Calls that were put back and finished over the collectives underneath because a peer-to-peer wait gave up."""
        return int(self._L.sabc_comm_p2p_fallbacks(self._h))

    def p2p_inject_silence(self, n=1):
        """Test hook: this shard skips its next n posts, so that its peers run into the bound of their waits."""
        self._check(self._L.sabc_comm_p2p_inject_silence(self._h, int(n)))

    # ---- state ----
    @property
    def n_local(self):
        return int(self._L.sabc_n_local(self._h))

    @property
    def local_offset(self):
        return int(self._L.sabc_local_offset(self._h))

    def get_population(self, theta=True, u=True, rho=True):
        """Local shard as (theta [d][n_local], u [s][n_local], rho [s][n_local]) = column-major n x k."""
        n = self.n_local
        th = np.empty((self.d, n)) if theta else None
        uu = np.empty((self.s, n)) if u else None
        rr = np.empty((self.s, n)) if rho else None
        self._check(self._L.sabc_get_population(self._h, _dp(th), _dp(uu), _dp(rr)))
        return th, uu, rr

    def set_population(self, theta=None, u=None, rho=None):
        n = self.n_local
        def prep(a, rows):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float64)
            if a.shape != (rows, n):
                raise ValueError(f"expected shape {(rows, n)}, got {a.shape}")
            return a
        th, uu, rr = prep(theta, self.d), prep(u, self.s), prep(rho, self.s)
        self._check(self._L.sabc_set_population(self._h, _dp(th), _dp(uu), _dp(rr)))

    @property
    def counters(self):
        out = (C.c_int64 * 4)()
        self._check(self._L.sabc_get_counters(self._h, out))
        return dict(n_simulation=out[0], n_accept=out[1], n_resampling=out[2], n_population_updates=out[3])

    def set_counters(self, n_simulation, n_accept, n_resampling, n_population_updates):
        arr = (C.c_int64 * 4)(n_simulation, n_accept, n_resampling, n_population_updates)
        self._check(self._L.sabc_set_counters(self._h, arr))

    @property
    def eps(self):
        out = np.zeros(_lib.MAX_STATS)
        ln = C.c_int32()
        self._check(self._L.sabc_get_epsilon(self._h, _dp(out), C.byref(ln)))
        return out[: ln.value].copy()

    def set_eps(self, eps):
        e = np.ascontiguousarray(eps, dtype=np.float64)
        self._check(self._L.sabc_set_epsilon(self._h, _dp(e), len(e)))

    @property
    def history(self):
        m = int(self._L.sabc_history_len(self._h))
        le = self.s if self.cfg.algorithm == _lib.ALG_MULTI_EPS else 1
        e, u, r = np.zeros((m, le)), np.zeros((m, self.s)), np.zeros((m, self.s))
        if m:
            self._check(self._L.sabc_get_history(self._h, _dp(e), _dp(u), _dp(r)))
        return e, u, r

    def clear_history(self):
        self._check(self._L.sabc_clear_history(self._h))

    def cdf_knots(self, stat):
        m = int(self._L.sabc_cdf_len(self._h, stat))
        out = np.zeros(m)
        if m:
            self._check(self._L.sabc_get_cdf_knots(self._h, stat, _dp(out)))
        return out

    def set_cdf_knots(self, stat, knots):
        k = np.ascontiguousarray(knots, dtype=np.float64)
        self._check(self._L.sabc_set_cdf_knots(self._h, stat, _dp(k), len(k)))

    def cdf_apply(self, rho):
        """rho: [s][m] -> u [s][m] on the device (cdf_estimators.jl:68-70)."""
        r = np.ascontiguousarray(rho, dtype=np.float64).reshape(self.s, -1)
        out = np.empty_like(r)
        self._check(self._L.sabc_cdf_apply(self._h, _dp(r), r.shape[1], _dp(out)))
        return out

    @property
    def proposal_sigma(self):
        out = np.zeros((self.d, self.d))
        self._check(self._L.sabc_get_proposal_sigma(self._h, _dp(out)))
        return out

    @property
    def ess(self):
        return float(self._L.sabc_last_ess(self._h))

    def simulate(self, theta, pid0, it):
        """theta [d][m] -> rho [s][m] with the RNG streams of particles pid0.. at iteration `it`."""
        th = np.ascontiguousarray(theta, dtype=np.float64).reshape(self.d, -1)
        out = np.empty((self.s, th.shape[1]))
        self._check(self._L.sabc_op_simulate(self._h, _dp(th), th.shape[1], int(pid0), int(it), _dp(out)))
        return out

    def prior(self, pid0, m):
        """rand(prior) for particle ids pid0.. and the log density of each draw, on the device: (theta [d][m], logpdf [m])."""
        th, lp = np.empty((self.d, int(m))), np.empty(int(m))
        self._check(self._L.sabc_op_prior(self._h, int(pid0), int(m), _dp(th), _dp(lp)))
        return th, lp

    # ---- measurement ----
    @property
    def host_syncs(self):
        return int(self._L.sabc_host_syncs(self._h))

    def set_host_chunk(self, particles):
        """Host-callback f_dist: proposals per callback (0 = automatic)."""
        self._check(self._L.sabc_set_host_chunk(self._h, int(particles)))

    @property
    def host_callback_seconds(self):
        """Seconds spent inside the caller's callbacks (f_dist, host prior) since the handle was created."""
        return float(self._L.sabc_host_callback_seconds(self._h))

    @property
    def host_callback_calls(self):
        return int(self._L.sabc_host_callback_calls(self._h))

    @property
    def kernel_launches(self):
        return int(self._L.sabc_kernel_launches(self._h))

    @property
    def collective_calls(self):
        return int(self._L.sabc_collective_calls(self._h))

    @property
    def persistent_launches(self):
        """Launches of the one-launch form of small shards (k_update_persistent) so far; 0 = the launch chain per update."""
        fn = getattr(self._L, "sabc_persistent_launches", None)
        return int(fn(self._h)) if fn is not None else 0

    @property
    def persistent_lanes(self):
        """Lanes per particle of the last one-launch update: 16 | 4 = a row | a quad of lanes runs a particle and shares its
        generator blocks (<= 2048 | 16 384 particles per launch), 1 = a lane per particle, 0 = none yet."""
        return int(self._L.sabc_persistent_lanes(self._h))

    @property
    def persistent_fallbacks(self):
        """One-launch updates that found the device too full for all their workgroups at once and handed the rest of their
        call to the launch chain (nothing touched, no error)."""
        return int(self._L.sabc_persistent_fallbacks(self._h))

    def profile_enable(self, on=True):
        self._check(self._L.sabc_profile_enable(self._h, int(on)))

    def profile_noops(self, kernel):
        """Timed launches of `kernel` that were no-ops behind a fired resample test (left out of profile_get)."""
        return int(self._L.sabc_profile_noops(self._h, int(kernel)))

    def profile_get(self, kernel):
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._L.sabc_profile_get(self._h, int(kernel), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


# ---- stand-alone operators ----
def op_build_cdf(x, device=0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(len(x) + 2)
    ln = C.c_int64()
    rc = _lib.lib().sabc_op_build_cdf(device, _dp(x), len(x), _dp(out), C.byref(ln))
    if rc:
        raise SABCError(rc, _lib.global_error())
    return out[: ln.value].copy()


def op_sort(x, device=0):
    """Ascending sort on the device (the library's own radix sort; what build_cdf uses)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    rc = _lib.lib().sabc_op_sort(device, _dp(x), len(x), _dp(out))
    if rc:
        raise SABCError(rc, _lib.global_error())
    return out


def op_cdf_eval(knots, q, device=0):
    k = np.ascontiguousarray(knots, dtype=np.float64)
    qq = np.atleast_1d(np.ascontiguousarray(q, dtype=np.float64))
    out = np.empty_like(qq)
    rc = _lib.lib().sabc_op_cdf_eval(device, _dp(k), len(k), _dp(qq), len(qq), _dp(out))
    if rc:
        raise SABCError(rc, _lib.global_error())
    return out if np.ndim(q) else float(out[0])


def op_eps_single(ubar, v):
    out = C.c_double()
    _lib.lib().sabc_op_eps_single(float(ubar), float(v), C.byref(out))
    return out.value


def op_eps_multi(ubar, v):
    ub = np.ascontiguousarray(ubar, dtype=np.float64)
    out = np.zeros(len(ub))
    rc = _lib.lib().sabc_op_eps_multi(_dp(ub), len(ub), float(v), _dp(out))
    if rc:
        raise SABCError(rc, "Division by zero - Mean u for a statistic is <= eps()")
    return out


def op_philox(seed, pid, purpose, it, k, device=0):
    w = (C.c_uint32 * 4)()
    z = np.zeros(2)
    rc = _lib.lib().sabc_op_philox(device, seed, pid, purpose, it, k, w, _dp(z))
    if rc:
        raise SABCError(rc, _lib.global_error())
    return [int(x) for x in w], z


def op_normal_pairs(seed, pid0, m, purpose=1, it=0, k=0, device=0):
    out = np.empty((int(m), 2))
    rc = _lib.lib().sabc_op_normal_pairs(device, int(seed), int(pid0), int(purpose), int(it), int(k), int(m), _dp(out))
    if rc:
        raise SABCError(rc, _lib.global_error())
    return out


def rccl_unique_id():
    """ncclGetUniqueId through the library's own RCCL binding (rank 0 calls it, then broadcasts)."""
    buf = C.create_string_buffer(128)
    rc = _lib.lib().sabc_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc:
        raise SABCError(rc, _lib.global_error())
    return buf.raw


def op_rng_peak(n_lanes=1_000_000, pairs_per_lane=50, repeats=20, device=0):
    """normals/s of the bare Philox + Box-Muller loop on this GPU (the VALU ceiling of the simulators)."""
    out = C.c_double()
    rc = _lib.lib().sabc_op_rng_peak(device, int(n_lanes), int(pairs_per_lane), int(repeats), C.byref(out))
    if rc:
        raise SABCError(rc, _lib.global_error())
    return out.value
