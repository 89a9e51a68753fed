/*
 * sabc_oracle.h -- CPU oracle for the SABC particle-population update loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm in
 * Eawag-SIAM/SimulatedAnnealingABC.jl v0.4.0 (src/SimulatedAnnealingABC.jl,
 * src/proposals.jl, src/cdf_estimators.jl).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library
 * (libsabc_hip.so) never links, loads or falls back to it.
 *
 * PARITY UNPINNED (bit level): the reference is Julia, Julia is not installed
 * here, the reference's tests hold no golden values and its RNG streams are
 * unseeded.  The oracle is pinned by (i) the reference's own property tests
 * (test/runtests.jl:9-29 CDF properties; :62-78 counters; :140,179 eps<1),
 * (ii) the equations written in the reference source (eps equations
 * SimulatedAnnealingABC.jl:93,113-114), and (iii) analytic posteriors.
 * Third-party arithmetic restated from published semantics:
 *   Interpolations.jl ^0.15 LinearMonotonicInterpolation + Flat()  -> orc_cdf_apply
 *   Roots.jl ^2.1 find_zero                                        -> orc_eps_*
 *   StatsBase.jl ^0.34 sample(weights, replace=true), cov, mean    -> orc_resample, orc_cov
 *   Distributions.jl ^0.25 Normal/Uniform/MvNormal rand, logpdf    -> orc_prior_*
 */
#ifndef SABC_ORACLE_H
#define SABC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_PARA 16
#define ORC_MAX_STATS 64
#define ORC_MAX_MODEL_PARAMS 32

/* model ids (device-coded simulators; see DESIGN.md "Simulators") */
enum { ORC_MODEL_HOST = 0, ORC_MODEL_GAUSS_IID = 1, ORC_MODEL_GAUSS2D = 2, ORC_MODEL_GK = 3, ORC_MODEL_LV = 4 };
/* prior kinds (per dimension; product distribution) */
enum { ORC_PRIOR_NORMAL = 0, ORC_PRIOR_UNIFORM = 1, ORC_PRIOR_EXPONENTIAL = 2, ORC_PRIOR_LOGNORMAL = 3,
       ORC_PRIOR_GAMMA = 4, ORC_PRIOR_BETA = 5, ORC_PRIOR_TRUNCNORMAL = 6 };
/* proposal kinds (proposals.jl:24,85,132) */
enum { ORC_PROP_RANDOMWALK = 0, ORC_PROP_DIFFEVO = 1, ORC_PROP_STRETCH = 2 };
/* algorithm (SimulatedAnnealingABC.jl:462) */
enum { ORC_ALG_SINGLE_EPS = 0, ORC_ALG_MULTI_EPS = 1 };
/* Philox stream purposes */
enum { ORC_PURPOSE_PRIOR = 0, ORC_PURPOSE_SIM = 1, ORC_PURPOSE_PROP = 2,
       ORC_PURPOSE_PROP2 = 3, ORC_PURPOSE_ACCEPT = 4, ORC_PURPOSE_RESAMPLE = 5 };

/* error codes (negative) */
enum {
  ORC_OK = 0,
  ORC_ERR_NSIM_TOO_SMALL = -1,   /* SimulatedAnnealingABC.jl:155-156 */
  ORC_ERR_NEG_DISTANCE = -2,     /* :185 */
  ORC_ERR_BAD_V = -3,            /* :261 */
  ORC_ERR_BAD_DELTA = -4,        /* :262 */
  ORC_ERR_BAD_ALGORITHM = -5,    /* :462-464 */
  ORC_ERR_BAD_BETA = -6,         /* proposals.jl:30 */
  ORC_ERR_ZERO_MEAN_U = -7,      /* :107-109 */
  ORC_ERR_BAD_CONFIG = -8,
  ORC_ERR_NOT_POSDEF = -9,       /* MvNormal(Sigma) Cholesky failure, proposals.jl:42 */
  ORC_ERR_EMPTY_CDF = -10,       /* maximum(x) of empty collection, cdf_estimators.jl:33 */
  ORC_ERR_ROOT = -11
};

typedef struct {
  int64_t n_particles;
  int32_t n_para;                 /* d */
  int32_t n_stats;                /* s */
  int32_t model_id;
  int32_t n_model_params;
  double  model_params[ORC_MAX_MODEL_PARAMS];
  int32_t prior_kind[ORC_MAX_PARA];
  double  prior_a[ORC_MAX_PARA];  /* Normal: mu    | Uniform: lower */
  double  prior_b[ORC_MAX_PARA];  /* Normal: sigma | Uniform: upper | Gamma: scale | Beta: beta */
  double  prior_c[ORC_MAX_PARA];  /* truncated Normal: lower */
  double  prior_d[ORC_MAX_PARA];  /* truncated Normal: upper */
  int32_t prior_joint;            /* 1: MvNormal(prior_a, L L') with L = prior_chol (row-major d x d lower) */
  int32_t _pad2;
  double  prior_chol[ORC_MAX_PARA * ORC_MAX_PARA];
  int32_t algorithm;
  int32_t _pad;
  double  v;                      /* used by initialization for eps_0 */
  double  delta;                  /* used by initialization's resample */
  uint64_t seed;
  /* ORC_MODEL_HOST: f_dist as a host callback, one particle per call (m = 1) */
  int (*host_fn)(void *ctx, const double *theta, const int64_t *ids, int64_t m, uint64_t iter, double *rho_out);
  void *host_ctx;
} orc_config;

typedef struct {
  int64_t n_simulation;           /* budget for this call (update_population! kw) */
  double  v;
  double  delta;
  double  resample;               /* default 2*n_particles */
  int64_t checkpoint_history;     /* default 1 */
  int32_t proposal_kind;
  int32_t _pad;
  double  proposal_p0;            /* RW: beta | DE: gamma0 | Stretch: a */
  double  proposal_p1;            /* DE: sigma_gamma */
} orc_update_args;

typedef struct orc_state orc_state;

/* ---- whole-algorithm entry points ---- */
int  orc_create(const orc_config *cfg, orc_state **out);
void orc_destroy(orc_state *st);
/* initialization(), SimulatedAnnealingABC.jl:151-227; n_simulation is the sabc() budget */
int  orc_initialize(orc_state *st, int64_t n_simulation);
/* update_population!(), SimulatedAnnealingABC.jl:251-402 */
int  orc_update(orc_state *st, const orc_update_args *args);
const char *orc_last_error(const orc_state *st);

/* state access; theta is SoA [d][n] (= Julia column-major n x d), u and rho are [s][n] */
const double *orc_theta(const orc_state *st);
const double *orc_u(const orc_state *st);
const double *orc_rho(const orc_state *st);
void orc_counters(const orc_state *st, int64_t out[4]); /* n_simulation, n_accept, n_resampling, n_population_updates */
int  orc_epsilon(const orc_state *st, double *out);     /* returns length (1 or s) */
int64_t orc_history_len(const orc_state *st);
/* each row: eps (len_eps), u means (s), rho means (s); rho_history has the same number of rows */
void orc_history(const orc_state *st, double *eps_hist, double *u_hist, double *rho_hist);
int64_t orc_cdf_len(const orc_state *st, int stat);
const double *orc_cdf_knots(const orc_state *st, int stat);
double orc_last_ess(const orc_state *st);
void orc_proposal_sigma(const orc_state *st, double *out /* d*d */);
void orc_set_threads(int nthreads);
/* 1: literal expressions (unfactored Lotka-Volterra step, slope-form interpolation) instead of the rewritten ones */
void orc_set_literal(int on);

/* ---- unit functions (each cites the reference lines it restates) ---- */
void   orc_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]);
void   orc_stream_block(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k, uint32_t out[4]);
double orc_u52(uint32_t hi, uint32_t lo);
void   orc_normal_pair(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k, double z[2]);
/* cdf_estimators.jl:23-44; knots_out needs n+2 doubles; returns knot count or error */
int64_t orc_build_cdf(const double *x, int64_t n, double *knots_out);
double orc_cdf_apply(const double *knots, int64_t len, double x);
/* SimulatedAnnealingABC.jl:92-95 */
double orc_eps_single(double ubar, double v);
/* SimulatedAnnealingABC.jl:100-117; ubar[s] column means */
int    orc_eps_multi(const double *ubar, int s, double v, double *eps_out);
double orc_multi_eps_beta(double ubar_i);
/* standard normal quantile (Wichura, AS 241 PPND16) */
double orc_norm_quantile(double p);
/* prior */
double orc_prior_logpdf(const orc_config *cfg, const double *theta);
void   orc_prior_sample(const orc_config *cfg, uint64_t pid, double *theta);
/* simulators: rho_out[s] */
int    orc_simulate(const orc_config *cfg, const double *theta, uint64_t pid, uint64_t iter, double *rho_out);
/* running sum of the resample weights in the specified blocked order (sabc_oracle.c) and the draw on it (:129);
   bs needs orc_scan_chunks(n) doubles */
int64_t orc_scan_chunks(int64_t n);
void   orc_weight_scan(const double *w, int64_t n, double *cum, double *bs, double totals[2]);
int64_t orc_resample_index(const double *cum, const double *bs, int64_t n, double t);
/* lower Cholesky of d x d row-major; returns 0 or ORC_ERR_NOT_POSDEF */
int    orc_cholesky(const double *a, int d, double *l);

#ifdef __cplusplus
}
#endif
#endif
