/*
 * sabc_hip.h -- C-ABI of libsabc_hip.so, the MI355X-native (gfx950) engine for the
 * particle-population update loop of SimulatedAnnealingABC.jl.
 *
 * The reference has no FFI: its hot path is reached through the
 * Julia functions `sabc` and `update_population!`.  The entry points below are what a
 * Julia wrapper binds with `ccall` in place of those function bodies (INTEGRATION.md
 * shows the binding); the Python host mirror in simulatedannealingabc.jl_amd/ binds the
 * same symbols with ctypes.  Citations are file:line under the reference checkout.
 *
 * Conventions
 *   - return 0 on success, a negative SABC_ERR_* otherwise; never throws across the ABI;
 *     sabc_last_error(h) holds the message the reference would have raised.
 *   - the caller owns every host buffer it passes; the library owns all device memory
 *     behind the opaque handle.  One host thread drives a handle; calls are not re-entrant.
 *   - matrices are column-major n x k (Julia layout) == SoA [k][n]: theta[k*n+i].
 *   - the product has no CPU path: every compute entry point fails with
 *     SABC_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef SABC_HIP_H
#define SABC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SABC_ABI_VERSION 6      /* 2: prior_c / prior_d (Gamma, Beta, truncated Normal priors); 3: prior_joint / prior_chol (MvNormal);
                                   4: peer-to-peer transport (sabc_comm_p2p_*), launch counters, SABC_MAX_PARA / _STATS 8 -> 16, prior_joint = 3;
                                   5: SABC_MAX_STATS 16 -> 64 (the buffers of sabc_get_epsilon / sabc_set_epsilon / sabc_cdf_apply callers);
                                   6: peer-to-peer life cycle (sabc_comm_p2p_setup, leaving handshake, descriptor 384 -> 512 bytes),
                                      sabc_update_args::history_phase / final_push */
#define SABC_MAX_PARA 16        /* host-callback and source-compiled simulators: any d up to this (the reference takes any length(prior),      */
#define SABC_MAX_STATS 64       /* SimulatedAnnealingABC.jl:163) and -- host-callback simulators -- any number of distances up to this (:164-167,181: */
                                /* summaries of a time series easily number dozens)                                                            */
#define SABC_MAX_SOURCE_STATS 16 /* a simulator compiled from source: its ECDF index lives in the fused kernel's LDS, s <= 16 */
#define SABC_MAX_JOINT_PARA 8   /* an MvNormal prior as data: d <= 8 (its Cholesky factor travels in kernel arguments)       */
#define SABC_MAX_MODEL_PARAMS 32

#if defined(__GNUC__)
#define SABC_API __attribute__((visibility("default")))
#else
#define SABC_API
#endif

/* device-coded simulators: the `f_dist` argument of sabc() (SimulatedAnnealingABC.jl:451)
   as data; definitions in DESIGN.md "Simulators" */
enum { SABC_MODEL_HOST = 0,   /* f_dist stays a host callable (sabc_set_host_simulator): any d, s within the maxima */
       SABC_MODEL_GAUSS_IID = 1, SABC_MODEL_GAUSS2D = 2, SABC_MODEL_GK = 3, SABC_MODEL_LV = 4,
       SABC_MODEL_USER = 5    /* f_dist as HIP source, compiled at run time into the fused update kernel
                                 (sabc_register_device_simulator): any d, s within the maxima */ };
/* `prior` argument as data: product of univariate families (Distributions.jl parametrisation):
   Normal(mu, sigma), Uniform(a, b), Exponential(theta = scale; second parameter unused), LogNormal(mu, sigma),
   Gamma(alpha = shape, theta = scale), Beta(alpha, beta), truncated(Normal(mu, sigma), lower, upper) */
enum { SABC_PRIOR_NORMAL = 0, SABC_PRIOR_UNIFORM = 1, SABC_PRIOR_EXPONENTIAL = 2, SABC_PRIOR_LOGNORMAL = 3,
       SABC_PRIOR_GAMMA = 4, SABC_PRIOR_BETA = 5, SABC_PRIOR_TRUNCNORMAL = 6 };
/* `proposal` argument (proposals.jl:24 RandomWalk, :85 DifferentialEvolution, :132 StretchMove) */
enum { SABC_PROP_RANDOMWALK = 0, SABC_PROP_DIFFEVO = 1, SABC_PROP_STRETCH = 2 };
/* `algorithm` argument (SimulatedAnnealingABC.jl:453,462) */
enum { SABC_ALG_SINGLE_EPS = 0, SABC_ALG_MULTI_EPS = 1 };

enum {
  SABC_OK = 0,
  SABC_ERR_NSIM_TOO_SMALL = -1,  /* SimulatedAnnealingABC.jl:155-156 */
  SABC_ERR_NEG_DISTANCE = -2,    /* :185 */
  SABC_ERR_BAD_V = -3,           /* :261 */
  SABC_ERR_BAD_DELTA = -4,       /* :262 */
  SABC_ERR_BAD_ALGORITHM = -5,   /* :462-464 */
  SABC_ERR_BAD_BETA = -6,        /* proposals.jl:30 */
  SABC_ERR_ZERO_MEAN_U = -7,     /* SimulatedAnnealingABC.jl:107-109 */
  SABC_ERR_BAD_CONFIG = -8,
  SABC_ERR_NOT_POSDEF = -9,      /* Cholesky inside MvNormal(...), proposals.jl:42 */
  SABC_ERR_EMPTY_CDF = -10,      /* maximum() of empty collection, cdf_estimators.jl:33 */
  SABC_ERR_ROOT = -11,
  SABC_ERR_NO_DEVICE = -20,
  SABC_ERR_HIP = -21,
  SABC_ERR_COMM = -22,
  SABC_ERR_STATE = -23,
  SABC_ERR_CALLBACK = -24        /* the host simulator returned non-zero */
};

typedef struct sabc_handle sabc_handle;

/* Everything sabc() fixes for the lifetime of a result (SimulatedAnnealingABC.jl:451-460). */
typedef struct {
  int32_t abi_version;            /* SABC_ABI_VERSION */
  int32_t device;                 /* HIP device ordinal */
  int64_t n_particles;            /* GLOBAL particle count (kw n_particles) */
  int32_t n_para;                 /* d = length(prior) */
  int32_t n_stats;                /* s = length(f_dist(theta)) */
  int32_t model_id;               /* SABC_MODEL_* */
  int32_t n_model_params;
  double  model_params[SABC_MAX_MODEL_PARAMS];
  int32_t prior_kind[SABC_MAX_PARA];
  double  prior_a[SABC_MAX_PARA]; /* Normal, LogNormal, truncated Normal: mu    | Uniform: lower | Exponential: scale theta |
                                     Gamma: shape alpha | Beta: alpha */
  double  prior_b[SABC_MAX_PARA]; /* Normal, LogNormal, truncated Normal: sigma | Uniform: upper | Exponential: unused |
                                     Gamma: scale theta | Beta: beta */
  double  prior_c[SABC_MAX_PARA]; /* truncated Normal: lower bound (others: unused) */
  double  prior_d[SABC_MAX_PARA]; /* truncated Normal: upper bound (others: unused) */
  int32_t prior_joint;            /* 0: product of the univariate families above | 1: MvNormal(mu, Sigma) over all n_para
                                     dimensions (d <= SABC_MAX_JOINT_PARA): mu = prior_a[0..d), Sigma = L L' with L = prior_chol | 2: host callbacks
                                     (sabc_set_host_prior; next to any simulator; prior_kind etc. unused) | 3: device code
                                     in the simulator's HIP source (SABC_MODEL_USER only: sabc_user_prior_sample /
                                     sabc_user_prior_logpdf next to sabc_user_simulate; prior_kind etc. unused) */
  int32_t reserved2;
  double  prior_chol[SABC_MAX_PARA * SABC_MAX_PARA];   /* prior_joint = 1: lower Cholesky factor of Sigma, row-major d x d */
  int32_t algorithm;              /* SABC_ALG_* */
  int32_t rank;                   /* this process' shard (0 when world == 1) */
  int32_t world;                  /* number of shards (GPUs) */
  int32_t reserved;
  double  v;                      /* kw v, used for eps_0 (:200-204) */
  double  delta;                  /* kw delta, used by the initial resample (:197) */
  uint64_t seed;                  /* Philox key */
} sabc_config;

/* Keyword arguments of update_population! (SimulatedAnnealingABC.jl:251-259). */
typedef struct {
  int64_t n_simulation;           /* budget; n_population_updates = n_simulation / n_particles (:275) */
  double  v;
  double  delta;
  double  resample;               /* kw resample (default 2 n_particles) */
  int64_t checkpoint_history;     /* >= 1 (`ix % checkpoint_history`, :367, is a DivideError for 0 in the reference) */
  int32_t proposal_kind;          /* SABC_PROP_* */
  int32_t more_chunks_follow;     /* 0: this call ends an update_population! call: the final history push of :378-382 applies.
                                     1: the wrapper has cut one update_population! call into several sabc_update calls (to print
                                     its progress lines, :359-364) and this is not the last of them: no final push */
  double  proposal_p0;            /* RandomWalk beta | DifferentialEvolution gamma0 | StretchMove a */
  double  proposal_p1;            /* DifferentialEvolution sigma_gamma */
  int64_t history_phase;          /* population updates of the SAME update_population! call done by earlier sabc_update calls:
                                     update ix of this call is update history_phase + ix of the loop at :294, and that number is
                                     what `% checkpoint_history` (:367) and the final push (:378) see -- a call cut into chunks
                                     of any length leaves the histories of the uncut call */
} sabc_update_args;

/* Collective hooks for world > 1 (one process per GPU).  `buf` is a device pointer when
   device_buffers != 0 (RCCL / torch.distributed "nccl"), a host pointer otherwise ("gloo").
   `stream` is the hipStream_t the library enqueues its kernels on. */
typedef int (*sabc_allreduce_fn)(void *ctx, void *buf, int64_t count_f64, void *stream);
typedef int (*sabc_allgather_fn)(void *ctx, const void *send, void *recv, int64_t count_f64_per_rank, void *stream);
/* personalised exchange (optional; the sharded resample uses it to fetch only the rows it drew, :129-132):
   send_counts[p] doubles go to rank p, taken from consecutive segments of `send`; recv_counts[p] arrive from rank p into
   consecutive segments of `recv`.  The count arrays (length `world`) are host memory. */
typedef int (*sabc_alltoallv_fn)(void *ctx, const void *send, const int64_t *send_counts, void *recv,
                                 const int64_t *recv_counts, int32_t world, void *stream);

/* The user's f_dist (SimulatedAnnealingABC.jl:164,175,315) as a host callback, for models that are not
   device-coded: called with the m proposals that passed the prior gate (theta column-major m x d,
   ids = their global particle ids, iter = population-update index, 0 during initialization); writes rho
   column-major m x s (non-negative).  Everything else of the update stays on the device. */
typedef int (*sabc_simulate_fn)(void *ctx, const double *theta, const int64_t *ids, int64_t m, uint64_t iter,
                                double *rho_out);

/* SimulatedAnnealingABC.jl:151 takes ANY Distributions.Distribution as prior.  One that is not among the families of
   sabc_config can be supplied as two host callbacks -- sabc_config::prior_joint = 2.  Next to SABC_MODEL_HOST the per-particle
   body is already cut at the host; next to a device-coded simulator (built in or from source) it is cut there for the
   log density alone: proposal kernel -> logpdf on the host -> the simulator as its own launch over the proposals inside the
   support -> accept kernel (same Philox streams as the fused kernel: the same run as with the prior as data):
     sample:  rand(prior) for the m particles `ids` (:174), theta_out column-major m x d;
     logpdf:  logpdf(prior, theta) for m parameter vectors (:314, :318), theta column-major m x d; -inf outside the support.
   Both return 0 on success.  The library calls `logpdf` once per (half-)batch of an update with the proposals followed by
   the current particles (m = 2 x batch). */
typedef int (*sabc_prior_sample_fn)(void *ctx, int64_t m, const int64_t *ids, double *theta_out);
typedef int (*sabc_prior_logpdf_fn)(void *ctx, int64_t m, const double *theta, double *logpdf_out);

SABC_API int         sabc_abi_version(void);
SABC_API const char *sabc_last_global_error(void);          /* for failures with no handle */
SABC_API int         sabc_device_count(void);

/* ---- lifetime ---- */
SABC_API int         sabc_create(const sabc_config *cfg, sabc_handle **out);
SABC_API void        sabc_destroy(sabc_handle *h);
SABC_API const char *sabc_last_error(const sabc_handle *h);
SABC_API int         sabc_set_stream(sabc_handle *h, void *hip_stream);
SABC_API int         sabc_set_collectives(sabc_handle *h, sabc_allreduce_fn ar, sabc_allgather_fn ag, void *ctx,
                                          int device_buffers);
/* after sabc_set_collectives: same ctx and buffer kind.  Without it a resample allgathers the whole population. */
SABC_API int         sabc_set_alltoallv(sabc_handle *h, sabc_alltoallv_fn fn);
/* bytes that landed in this shard's receive buffers through the collectives since sabc_create */
SABC_API int64_t     sabc_comm_bytes(const sabc_handle *h);
SABC_API int         sabc_set_host_simulator(sabc_handle *h, sabc_simulate_fn fn, void *ctx);   /* SABC_MODEL_HOST */
SABC_API int         sabc_set_host_prior(sabc_handle *h, sabc_prior_sample_fn sample, sabc_prior_logpdf_fn logpdf,
                                         void *ctx);                                          /* prior_joint = 2 */
/* SABC_MODEL_HOST: a half batch reaches f_dist in chunks of `particles` proposals (0 = automatic: an eighth of the half
   batch, at least 4096), so that the callback of one chunk overlaps the device's work on its neighbours */
SABC_API int         sabc_set_host_chunk(sabc_handle *h, int64_t particles);
/* seconds spent inside the caller's callbacks (f_dist, host prior) since sabc_create, and calls of f_dist: what is left
   of a call's wall time is the library's */
SABC_API double      sabc_host_callback_seconds(const sabc_handle *h);
SABC_API int64_t     sabc_host_callback_calls(const sabc_handle *h);
/* SABC_MODEL_USER: the user's f_dist (SimulatedAnnealingABC.jl:164,175,315) as device code.  `hip_source` is HIP C++
   defining, at global scope,
       __device__ void sabc_user_simulate(const double *theta,        // the d parameters
                                          const double *params,       // sabc_config::model_params
                                          sabc::NormalStream &rng,    // rng.next() / rng.pair(z0, z1): N(0,1) draws;
                                                                      // rng.uniform_pair(u0, u1): U(0,1) draws;
                                                                      // rng.for_pairs(n, [&](double z0, double z1) {...}):
                                                                      // the next n pairs, in stream order
                                          double *rho_out);           // the s non-negative distances
   (a function of its arguments and its draws alone.  for_pairs is the loop to draw the bulk of a simulation with: a small
   population runs a call's updates in one launch with a TEAM of 4 or 16 lanes per particle, which then generate 16 or 64
   pairs at a time, four blocks per lane -- csrc/device_rng.hpp; the stream is the same stream however it is drawn.)
   With sabc_config::prior_joint = 3 the same source also defines the PRIOR -- any distribution, evaluated inside the fused
   kernel (SimulatedAnnealingABC.jl:151 takes any Distributions.Distribution):
       __device__ void   sabc_user_prior_sample(const double *params, sabc::NormalStream &rng, double *theta_out);
       __device__ double sabc_user_prior_logpdf(const double *theta, const double *params);   // -INFINITY outside the support
   (`rng`: the particle's prior stream.)
   It is compiled with hipRTC for gfx950 against csrc/update_kernel.hpp -- the same propose -> simulate -> ECDF -> accept
   kernel, reductions and Philox streams as the built-in simulators -- and must be registered before sabc_initialize.
   On failure sabc_last_error(h) holds the compiler log. */
SABC_API int         sabc_register_device_simulator(sabc_handle *h, const char *hip_source);
/* the compiler stage alone (needs hipRTC but no device): 0 if `hip_source` compiles into the update kernels for (d, s);
   the compiler log is copied into log_out (may be NULL) */
SABC_API int         sabc_op_compile_device_simulator(const char *hip_source, int32_t d, int32_t s, char *log_out,
                                                      int64_t log_cap);
/* the same for a source that also carries the prior (prior_joint = 3) */
SABC_API int         sabc_op_compile_device_simulator_with_prior(const char *hip_source, int32_t d, int32_t s, char *log_out,
                                                                 int64_t log_cap);
SABC_API int         sabc_comm_init_rccl(sabc_handle *h, const void *unique_id_128b);
SABC_API int         sabc_comm_unique_id(void *out_128b);
/* one allreduce + one allgather through the installed collectives, checked on the host */
SABC_API int         sabc_comm_selftest(sabc_handle *h);

/* ---- peer-to-peer transport: the shards of ONE node exchange through each other's HBM (xGMI) ----
   Replaces, per population update, k_reduce_partials -> ncclAllReduce -> k_control by one launch that stores this shard's
   row of fused sums into slots every peer has mapped, waits (bounded) for the peers' rows, adds them in rank order and
   runs the control step (SimulatedAnnealingABC.jl:334,348-354); DifferentialEvolution / StretchMove partners
   (proposals.jl:105-106,141) and the rows a resample draws (:129-132) are read from their owner's memory directly.
   Set-up: every shard fills a descriptor (IPC handles of its slot area, both population buffers and rho, + its pid and
   raw pointers for shards living in the same process); the descriptors of all shards, in rank order, go to
   sabc_comm_p2p_init -- exchanged by the caller, or (all_descs == NULL) by the library over the collectives already
   installed (sabc_set_collectives / sabc_comm_init_rccl), which also stay as the fallback transport.
   Every wait is bounded (sabc_comm_p2p_set_timeout, default 5000 ms).  A shard that gives up fails the call on EVERY shard
   (the end-of-call status exchange) and switches the handle back to the collectives underneath; with such collectives
   installed sabc_update then puts the particles back and repeats the call over them (sabc_comm_p2p_fallbacks counts),
   without them it returns SABC_ERR_COMM per its error contract.

   LEAVING.  No shard ever frees memory a peer may still read, and no caller has to arrange that with barriers:
   sabc_comm_p2p_disable, a failed call, a new set-up and sabc_destroy all LEAVE the group -- the shard marks itself as
   leaving (a host page every peer has mapped + a word in every peer's slots: their waits for this shard give up at once
   instead of running into the bound), drains its stream, unmaps every peer and records that it has.  A peer notices at the
   entry of its next sabc_update / sabc_initialize (before it launches anything) or inside the call it is in, leaves as
   well, and carries on over the collectives underneath or returns SABC_ERR_COMM.  sabc_destroy then waits (up to
   sabc_comm_p2p_set_destroy_wait, default: the bound of the waits) until every peer has recorded that it unmapped this
   shard's memory; memory a peer has not released by then is PARKED -- kept until the process exits -- never freed under
   a reader (sabc_comm_p2p_parked_bytes).  The reference never corrupts state on the way out either
   (SimulatedAnnealingABC.jl:264-267,387-397). */
#define SABC_P2P_DESC_BYTES 512
#define SABC_P2P_MAX_WORLD 8
/* ONE call that sets the transport up on every shard, or leaves every shard on the collectives -- the same answer
   everywhere, nobody waiting out a bound because the others decided differently: descriptors exchanged over the
   collectives already installed -> peers mapped -> agreement (an allreduce of an ok flag) -> self-test -> agreement.
   Returns 1 when the handle now runs peer to peer, 0 when every shard stays on the collectives (sabc_last_error says why),
   < 0 when the collectives themselves failed.  Collective: every shard calls it. */
SABC_API int         sabc_comm_p2p_setup(sabc_handle *h);
/* the pieces of the above, for callers that move the descriptors themselves (shards in one process; tests) */
SABC_API int         sabc_comm_p2p_descriptor(sabc_handle *h, void *out_desc);
SABC_API int         sabc_comm_p2p_init(sabc_handle *h, const void *all_descs);
/* first contact, checked on the device and the host, SABC_ERR_COMM within the bound: a row of known values through the
   slots + a barrier; then what the transport READS -- every shard writes a rank- and round-tagged pattern into lines spread
   over both of its population buffers and rho (plain device memory), a barrier, every shard reads every peer's lines
   through its mappings and compares; a second round with another pattern (a line kept from the first would show); the
   lines are put back.  Every shard has to call it the same number of times. */
SABC_API int         sabc_comm_p2p_selftest(sabc_handle *h);
SABC_API int         sabc_comm_p2p_set_timeout(sabc_handle *h, double milliseconds);
/* leave the group (see LEAVING above); the handle keeps its particles and continues over the collectives underneath */
SABC_API int         sabc_comm_p2p_disable(sabc_handle *h);
/* how long sabc_destroy waits for the peers' acknowledgement; 0 = not at all (for finalizers: what a peer has not
   released is parked at once) */
SABC_API int         sabc_comm_p2p_set_destroy_wait(sabc_handle *h, double milliseconds);
/* bytes of device memory this process has parked so far (0 in a run whose shards all left in order) */
SABC_API int64_t     sabc_comm_p2p_parked_bytes(void);
/* 1 while the handle runs over the peer-to-peer transport */
SABC_API int         sabc_comm_p2p_active(const sabc_handle *h);
/* sabc_update calls in which a peer-to-peer wait gave up and that were put back (device-side copy of the particles taken
   at entry) and finished over the collectives installed underneath -- the caller sees a successful call */
SABC_API int64_t     sabc_comm_p2p_fallbacks(const sabc_handle *h);
/* test hooks.  inject_silence: n > 0: this shard skips its next n posts (rows of sums / barrier flags / call status), so
   that its peers run into the bound; n < 0: -n more posts go out first, then one is skipped.  inject_loss: n more posts
   go out, then one reaches only this shard's own slots -- a post lost on the wire: the shard itself carries on with its
   peers' rows (and may flip its population buffers in a resample they never reach).  inject_stale: the next self-test's
   pattern check on this shard reports a mismatch (what a stale line would look like) */
SABC_API int         sabc_comm_p2p_inject_silence(sabc_handle *h, int32_t n);
SABC_API int         sabc_comm_p2p_inject_loss(sabc_handle *h, int32_t n);
SABC_API int         sabc_comm_p2p_inject_stale(sabc_handle *h, int32_t n);

/* ---- the hot path ---- */
/* initialization(), SimulatedAnnealingABC.jl:151-227.  n_simulation is sabc()'s budget (:155). */
SABC_API int sabc_initialize(sabc_handle *h, int64_t n_simulation);
/* update_population!(), SimulatedAnnealingABC.jl:251-402: proposals.jl call + update_proposal!,
   per-particle body :308-331, resample :124-137, eps schedules :92-117, histories :367-382. */
SABC_API int sabc_update(sabc_handle *h, const sabc_update_args *args);
/* Error contract of sabc_update: argument errors (SABC_ERR_BAD_V / _BAD_DELTA / _BAD_BETA / _BAD_CONFIG) leave the
   handle untouched.  A failure inside the loop (device-side SABC_ERR_ZERO_MEAN_U / _NOT_POSDEF, a failing callback or
   collective) drains the queue, puts counters, epsilon and histories back to their values at entry -- the reference
   works on copies and leaves its state untouched when it throws (:264-267, :387-397) -- and, because the particles were
   updated in place on the device, the handle then refuses sabc_update (SABC_ERR_STATE) until sabc_set_population has
   restored them. */

/* ---- result / state (SABCresult :55-60, SABCstate :28-42) ---- */
SABC_API int64_t sabc_n_global(const sabc_handle *h);         /* n_particles of the whole population (all shards) */
SABC_API int64_t sabc_n_local(const sabc_handle *h);          /* particles held by this shard */
SABC_API int64_t sabc_local_offset(const sabc_handle *h);     /* global id of the first local particle */
/* local shard, column-major: theta n_local x d, u n_local x s, rho n_local x s; NULL skips */
SABC_API int sabc_get_population(sabc_handle *h, double *theta, double *u, double *rho);
SABC_API int sabc_set_population(sabc_handle *h, const double *theta, const double *u, const double *rho);
/* out[4] = n_simulation, n_accept, n_resampling, n_population_updates */
SABC_API int sabc_get_counters(const sabc_handle *h, int64_t out[4]);
SABC_API int sabc_set_counters(sabc_handle *h, const int64_t in[4]);
SABC_API int sabc_get_epsilon(const sabc_handle *h, double *eps, int32_t *len);
SABC_API int sabc_set_epsilon(sabc_handle *h, const double *eps, int32_t len);
SABC_API int64_t sabc_history_len(const sabc_handle *h);
/* row-major [len][eps_len], [len][s], [len][s] */
SABC_API int sabc_get_history(const sabc_handle *h, double *eps_hist, double *u_hist, double *rho_hist);
SABC_API int sabc_clear_history(sabc_handle *h);
/* state.cdfs_dist_prior (SimulatedAnnealingABC.jl:37): knots of the prior-predictive ECDF per
   statistic (cdf_estimators.jl:33) and its evaluation on device (cdf_estimators.jl:68-70) */
SABC_API int64_t sabc_cdf_len(const sabc_handle *h, int32_t stat);
SABC_API int sabc_get_cdf_knots(sabc_handle *h, int32_t stat, double *out);
SABC_API int sabc_set_cdf_knots(sabc_handle *h, int32_t stat, const double *knots, int64_t len);
/* rho: column-major m x s; u_out likewise */
SABC_API int sabc_cdf_apply(sabc_handle *h, const double *rho, int64_t m, double *u_out);
SABC_API int sabc_get_proposal_sigma(const sabc_handle *h, double *sigma_dxd);
SABC_API double sabc_last_ess(const sabc_handle *h);

/* ---- operators of the path, callable on their own (mirrors of the reference's internal
        functions so that its unit tests can be restated against the device code) ---- */
/* build_cdf(x)(q): cdf_estimators.jl:23-44; test/runtests.jl:9-29 */
SABC_API int sabc_op_build_cdf(int32_t device, const double *x, int64_t n, double *knots_out, int64_t *len_out);
SABC_API int sabc_op_cdf_eval(int32_t device, const double *knots, int64_t len, const double *q, int64_t m,
                              double *out);
/* the ascending sort behind build_cdf (cdf_estimators.jl:33 `sort(x)`): the library's own radix sort, any doubles */
SABC_API int sabc_op_sort(int32_t device, const double *x, int64_t n, double *out);
/* update_epsilon_single_eps / update_epsilon_multi_eps: SimulatedAnnealingABC.jl:92-117 (host code of the engine) */
SABC_API int sabc_op_eps_single(double ubar, double v, double *eps_out);
SABC_API int sabc_op_eps_multi(const double *ubar, int32_t s, double v, double *eps_out);
/* f_dist(theta) on device for m parameter vectors (column-major m x d -> m x s), RNG stream of
   particle ids pid0..pid0+m-1 at iteration `iter` */
SABC_API int sabc_op_simulate(sabc_handle *h, const double *theta, int64_t m, uint64_t pid0, uint64_t iter,
                              double *rho_out);
/* rand(prior) and logpdf(prior, .) on device (SimulatedAnnealingABC.jl:174,314,318) for particle ids pid0..pid0+m-1:
   theta_out column-major m x d, logpdf_out m values (the log density of each draw) */
SABC_API int sabc_op_prior(sabc_handle *h, uint64_t pid0, int64_t m, double *theta_out, double *logpdf_out);
/* Philox4x32-10 block and the Box-Muller pair derived from it, evaluated on device */
SABC_API int sabc_op_philox(int32_t device, uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k,
                            uint32_t out_words[4], double out_normals[2]);

/* Box-Muller pairs of block k for particles pid0..pid0+m-1 (out: z0,z1 interleaved): bulk accuracy
   check of the device log / sqrt / sincos against the oracle's libm */
SABC_API int sabc_op_normal_pairs(int32_t device, uint64_t seed, uint64_t pid0, uint32_t purpose, uint64_t iter,
                                  uint32_t k, int64_t m, double *out_2m);

/* Rate (normals/s) of the bare generator loop -- Philox4x32-10 block + Box-Muller pair, nothing else --
   on n_lanes lanes: the VALU ceiling of every simulator that draws from it */
SABC_API int sabc_op_rng_peak(int32_t device, int64_t n_lanes, int32_t pairs_per_lane, int32_t repeats,
                              double *normals_per_s);

/* ---- measurement ---- */
enum { SABC_KERNEL_UPDATE = 0, SABC_KERNEL_REDUCE = 1, SABC_KERNEL_RESAMPLE = 2, SABC_KERNEL_INIT = 3,
       SABC_KERNEL_COLLECTIVE = 4,   /* the allreduce of the fused sums on the collectives transport (level 2 only; on the
                                        peer-to-peer transport the exchange is inside SABC_KERNEL_REDUCE's one launch) */
       SABC_KERNEL_COUNT = 5 };
/* HIP-event timing of kernels on the library's stream, accumulated since enable.
   level 0 off, 1 = every second launch of SABC_KERNEL_UPDATE (the events ride on the kernel's dispatch packet and cost
   ~4 us of queue time each: sampling halves what the measurement adds), 2 = every kernel, 3 = every launch of
   SABC_KERNEL_UPDATE.  sabc_profile_get returns the time and the number of the launches that were timed. */
SABC_API int sabc_profile_enable(sabc_handle *h, int32_t level);
SABC_API int sabc_profile_get(sabc_handle *h, int32_t kernel, double *total_ms, int64_t *launches);
/* timed launches of `kernel` that turned out to be no-ops (queued ahead of a resample test that fired: they return at the
   halt flag in ~2 us) -- left out of sabc_profile_get's time and count */
SABC_API int64_t sabc_profile_noops(sabc_handle *h, int32_t kernel);
/* how many times update()/initialize() had to wait for the device so far (run-ahead windows) */
SABC_API int64_t sabc_host_syncs(const sabc_handle *h);
/* kernels the library has launched on its stream so far, and collective calls it has issued (hooks / RCCL): per
   population update at world > 1 the RCCL transport takes k_update + k_reduce_partials + ncclAllReduce + k_control,
   the peer-to-peer transport k_update + ONE launch */
SABC_API int64_t sabc_kernel_launches(const sabc_handle *h);
/* launches of the one-launch form of small shards so far (k_update_persistent: one shard of <= 65 536 particles with a
   device-coded simulator runs the population updates between two resamples in ONE launch; SABC_PERSISTENT=0 switches it
   off): 0 on a handle that takes the launch chain per update */
SABC_API int64_t sabc_persistent_launches(const sabc_handle *h);
/* lanes per particle of the handle's last such launch: a TEAM of 16 (a row of the wave; <= 2048 particles per launch) or 4 (a
   quad; <= 16 384) while that many times the workgroups fit the launch -- the device is then so empty that a particle's chain
   of generator blocks is what an update waits for: the lanes of a team run the particle side by side and share the blocks,
   same streams --, else 1; 0 before the first.  SABC_PERSISTENT_LANES = 1 | 4 | 16 overrides, SABC_PERSISTENT_LANES4_MAX /
   SABC_PERSISTENT_LANES16_MAX move the bounds */
SABC_API int32_t sabc_persistent_lanes(const sabc_handle *h);
/* one-launch updates that found the device too full for all their workgroups to be resident at once (other handles' or
   processes' kernels on it) and left without touching anything -- within SABC_PERSISTENT_RENDEZVOUS_MS, 20 ms --: the rest of
   that call ran as the launch chain, no error */
SABC_API int64_t sabc_persistent_fallbacks(const sabc_handle *h);
SABC_API int64_t sabc_collective_calls(const sabc_handle *h);

#ifdef __cplusplus
}
#endif
#endif
