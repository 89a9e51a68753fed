/*
 * sabc_oracle.c -- CPU oracle (plain C) for the SABC particle-population update
 * loop.  TEST INFRASTRUCTURE ONLY; see sabc_oracle.h for the scope note and
 * the "PARITY UNPINNED" statement.
 *
 * Every function cites the reference lines it restates, relative to
 * /root/reference (Eawag-SIAM/SimulatedAnnealingABC.jl v0.4.0).  Nothing here
 * is copied: the reference is Julia, this is a from-scratch C restatement
 * with its own counter-based RNG stream layout (DESIGN.md "RNG streams").
 */
#include "sabc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846
#define ORC_LOG2PI 1.8378770664093454835606594728112

struct orc_state {
  orc_config cfg;
  int64_t n;
  int d, s;
  double *theta;   /* [d][n] */
  double *u;       /* [s][n] */
  double *rho;     /* [s][n] */
  double *knots;   /* [s][n+2] */
  int64_t cdf_len[ORC_MAX_STATS];
  double eps[ORC_MAX_STATS];
  int eps_len;
  int initialized;
  int64_t n_simulation, n_accept, n_resampling, n_population_updates;
  /* histories (SABCstate fields, SimulatedAnnealingABC.jl:33-35) */
  double *eps_hist, *u_hist, *rho_hist;
  int64_t hist_len, hist_cap;
  /* proposal state (RandomWalk.Sigma, proposals.jl:24-27) */
  double sigma[ORC_MAX_PARA * ORC_MAX_PARA];
  double chol[ORC_MAX_PARA * ORC_MAX_PARA];
  double last_ess;
  char err[256];
};

static int g_threads = 1;
void orc_set_threads(int nthreads) { g_threads = nthreads < 1 ? 1 : nthreads; }

/* Literal mode: where the oracle's default arithmetic was rewritten into an algebraically equal form together with the
   device code (the factored Lotka-Volterra step, the weight form of the ECDF interpolant), switch back to the expression
   exactly as one writes it down from the model / from Interpolations' linear interpolant.  tests/test_oracle_units.py
   runs both and bounds the difference, so a mistake made identically in both rewrites cannot hide. */
static int g_literal = 0;
void orc_set_literal(int on) { g_literal = on ? 1 : 0; }

static int fail(orc_state *st, int code, const char *msg) {
  if (st) { strncpy(st->err, msg, sizeof(st->err) - 1); st->err[sizeof(st->err) - 1] = 0; }
  return code;
}
const char *orc_last_error(const orc_state *st) { return st ? st->err : "null state"; }

/* ------------------------------------------------------------------ */
/* RNG: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).  Replaces   */
/* Julia's task-local Xoshiro at every rand/randn site                 */
/* (SimulatedAnnealingABC.jl:163,174,324; proposals.jl:42,54,105,110,  */
/* 141,144).  Stream parity with Julia is impossible by construction   */
/* (SURVEY.md section 5 RNG), so the layout is the build's own.        */
/* ------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* counter = (particle id, block index, iteration, purpose | pid_hi<<8) */
void orc_stream_block(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k, uint32_t out[4]) {
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  uint32_t ctr[4] = { (uint32_t)pid, k, (uint32_t)iter,
                      (purpose & 0xFFu) | ((uint32_t)(pid >> 32) << 8) | ((uint32_t)((iter >> 32) & 0xFFu) << 24) };
  orc_philox4x32_10(key, ctr, out);
}

/* 52-bit uniform in (0,1): x = (low 20 bits of `hi`) : (all 32 bits of `lo`); (x + 1/2) * 2^-52 is exact
   in binary64.  (These are the 52 bits that drop into a double's mantissa without a shift.) */
double orc_u52(uint32_t hi, uint32_t lo) {
  uint64_t x = ((uint64_t)(hi & 0xFFFFFu) << 32) | lo;
  return ((double)x + 0.5) * 0x1.0p-52;
}

/* Box-Muller pair from one Philox block */
void orc_normal_pair(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k, double z[2]) {
  uint32_t w[4];
  orc_stream_block(seed, pid, purpose, iter, k, w);
  double ua = orc_u52(w[0], w[1]);
  double ub = orc_u52(w[2], w[3]);
  double r = sqrt(-2.0 * log(ua));
  double ang = 2.0 * ORC_PI * ub;
  z[0] = r * cos(ang);
  z[1] = r * sin(ang);
}

typedef struct { uint64_t seed, pid, iter; uint32_t purpose, k; int have; double spare; } nstream;
static nstream ns_open(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter) {
  nstream s = { seed, pid, iter, purpose, 0, 0, 0.0 };
  return s;
}
static double ns_next(nstream *s) {
  if (s->have) { s->have = 0; return s->spare; }
  double z[2];
  orc_normal_pair(s->seed, s->pid, s->purpose, s->iter, s->k++, z);
  s->spare = z[1]; s->have = 1;
  return z[0];
}

static uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

/* ------------------------------------------------------------------ */
/* Prior: product of Normal / Uniform.  Distributions.jl rand / logpdf */
/* as used at SimulatedAnnealingABC.jl:163,174,314,318.                */
/* ------------------------------------------------------------------ */
/* standard normal quantile: Wichura, Algorithm AS 241 (PPND16), Applied Statistics 37 (1988) */
double orc_norm_quantile(double p) {
  static const double a[8] = { 3.3871328727963666080e0, 1.3314166789178437745e2, 1.9715909503065514427e3, 1.3731693765509461125e4,
                               4.5921953931549871457e4, 6.7265770927008700853e4, 3.3430575583588128105e4, 2.5090809287301226727e3 };
  static const double b[8] = { 1.0, 4.2313330701600911252e1, 6.8718700749205790830e2, 5.3941960214247511077e3,
                               2.1213794301586595867e4, 3.9307895800092710610e4, 2.8729085735721942674e4, 5.2264952788528545610e3 };
  static const double c[8] = { 1.42343711074968357734e0, 4.63033784615654529590e0, 5.76949722146069140550e0, 3.64784832476320460504e0,
                               1.27045825245236838258e0, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4 };
  static const double d[8] = { 1.0, 2.05319162663775882187e0, 1.67638483018380384940e0, 6.89767334985100004550e-1,
                               1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9 };
  static const double e[8] = { 6.65790464350110377720e0, 5.46378491116411436990e0, 1.78482653991729133580e0, 2.96560571828504891230e-1,
                               2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7 };
  static const double f[8] = { 1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2,
                               7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15 };
  const double q = p - 0.5;
  double num = 0.0, den = 0.0, r;
  if (fabs(q) <= 0.425) {
    r = 0.180625 - q * q;
    for (int i = 7; i >= 0; --i) { num = num * r + a[i]; den = den * r + b[i]; }
    return q * num / den;
  }
  r = q < 0.0 ? p : 1.0 - p;
  if (!(r > 0.0)) return q < 0.0 ? -INFINITY : INFINITY;
  r = sqrt(-log(r));
  if (r <= 5.0) {
    r -= 1.6;
    for (int i = 7; i >= 0; --i) { num = num * r + c[i]; den = den * r + d[i]; }
  } else {
    r -= 5.0;
    for (int i = 7; i >= 0; --i) { num = num * r + e[i]; den = den * r + f[i]; }
  }
  return q < 0.0 ? -(num / den) : num / den;
}

static double norm_cdf(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

/* mass of the standard normal between the standardised bounds (complement form in the upper tail) */
static void truncnormal_mass(const orc_config *cfg, int k, double *p_lo, double *mass) {
  const double lo = (cfg->prior_c[k] - cfg->prior_a[k]) / cfg->prior_b[k], hi = (cfg->prior_d[k] - cfg->prior_a[k]) / cfg->prior_b[k];
  *p_lo = lo > 0 ? -norm_cdf(-lo) : norm_cdf(lo);       /* negative: bounds in the upper tail, drawn in the mirrored lower tail */
  *mass = lo > 0 ? norm_cdf(-lo) - norm_cdf(-hi) : norm_cdf(hi) - norm_cdf(lo);
}

/* Gamma(shape, 1): Marsaglia & Tsang (2000); attempt t: normal from block base + 8 (2t), uniform from base + 8 (2t+1) */
static double gamma_sample(uint64_t seed, uint64_t pid, uint32_t base, double shape) {
  const double al = shape < 1.0 ? shape + 1.0 : shape;
  const double dd = al - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * dd);
  double g = dd, boost = 1.0;
  for (uint32_t t = 0; t < 64u; ++t) {
    double z[2];
    uint32_t w[4];
    orc_normal_pair(seed, pid, ORC_PURPOSE_PRIOR, 0, base + 8u * (2u * t), z);
    orc_stream_block(seed, pid, ORC_PURPOSE_PRIOR, 0, base + 8u * (2u * t + 1u), w);
    if (t == 0) boost = orc_u52(w[2], w[3]);
    const double v1 = 1.0 + cc * z[0];
    if (!(v1 > 0.0)) continue;
    const double v = v1 * v1 * v1;
    g = dd * v;
    if (log(orc_u52(w[0], w[1])) < 0.5 * z[0] * z[0] + dd - dd * v + dd * log(v)) break;
  }
  return shape < 1.0 ? g * pow(boost, 1.0 / shape) : g;
}

void orc_prior_sample(const orc_config *cfg, uint64_t pid, double *theta) {
  if (cfg->prior_joint) {               /* MvNormal(mu, L L'): mu + L z, z_k = first normal of block k (rand(MvNormal), :174) */
    const int d = cfg->n_para;
    double z[ORC_MAX_PARA];
    for (int k = 0; k < d; ++k) { double zz[2]; orc_normal_pair(cfg->seed, pid, ORC_PURPOSE_PRIOR, 0, (uint32_t)k, zz); z[k] = zz[0]; }
    for (int k = 0; k < d; ++k) {
      double x = cfg->prior_a[k];
      for (int l = 0; l <= k; ++l) x += cfg->prior_chol[k * d + l] * z[l];
      theta[k] = x;
    }
    return;
  }
  for (int k = 0; k < cfg->n_para; ++k) {
    const int kind = cfg->prior_kind[k];
    if (kind == ORC_PRIOR_GAMMA) { theta[k] = cfg->prior_b[k] * gamma_sample(cfg->seed, pid, (uint32_t)k, cfg->prior_a[k]); continue; }
    if (kind == ORC_PRIOR_BETA) {
      const double x = gamma_sample(cfg->seed, pid, (uint32_t)k, cfg->prior_a[k]);
      const double y = gamma_sample(cfg->seed, pid, (uint32_t)k + (1u << 16), cfg->prior_b[k]);
      theta[k] = x / (x + y);
      continue;
    }
    uint32_t w[4];
    orc_stream_block(cfg->seed, pid, ORC_PURPOSE_PRIOR, 0, (uint32_t)k, w);
    double ua = orc_u52(w[0], w[1]);
    if (kind == ORC_PRIOR_NORMAL || kind == ORC_PRIOR_LOGNORMAL) {
      double ub = orc_u52(w[2], w[3]);
      double z0 = sqrt(-2.0 * log(ua)) * cos(2.0 * ORC_PI * ub);
      double x = cfg->prior_a[k] + cfg->prior_b[k] * z0;
      theta[k] = kind == ORC_PRIOR_LOGNORMAL ? exp(x) : x;
    } else if (kind == ORC_PRIOR_EXPONENTIAL) {
      theta[k] = -cfg->prior_a[k] * log(ua);               /* inverse CDF, scale parametrisation */
    } else if (kind == ORC_PRIOR_TRUNCNORMAL) {            /* inverse CDF on [Phi(lo'), Phi(hi')] */
      double p_lo, mass;
      truncnormal_mass(cfg, k, &p_lo, &mass);
      double x = p_lo < 0.0 ? cfg->prior_a[k] - cfg->prior_b[k] * orc_norm_quantile(-p_lo - ua * mass)
                            : cfg->prior_a[k] + cfg->prior_b[k] * orc_norm_quantile(p_lo + ua * mass);
      theta[k] = fmin(fmax(x, cfg->prior_c[k]), cfg->prior_d[k]);
    } else {
      theta[k] = cfg->prior_a[k] + (cfg->prior_b[k] - cfg->prior_a[k]) * ua;
    }
  }
}

double orc_prior_logpdf(const orc_config *cfg, const double *theta) {
  if (cfg->prior_joint) {               /* logpdf(MvNormal(mu, L L'), x) = -1/2 |L^-1 (x - mu)|^2 - d/2 log 2 pi - sum log L_kk */
    const int d = cfg->n_para;
    double y[ORC_MAX_PARA], q = 0.0, logdet = 0.0;
    for (int k = 0; k < d; ++k) {
      double r = theta[k] - cfg->prior_a[k];
      for (int l = 0; l < k; ++l) r -= cfg->prior_chol[k * d + l] * y[l];
      y[k] = r / cfg->prior_chol[k * d + k];
      q += y[k] * y[k];
      logdet += log(cfg->prior_chol[k * d + k]);
    }
    return -0.5 * q - (0.5 * d * ORC_LOG2PI + logdet);
  }
  double lp = 0.0;
  for (int k = 0; k < cfg->n_para; ++k) {
    double x = theta[k];
    const double a = cfg->prior_a[k], b = cfg->prior_b[k];
    if (cfg->prior_kind[k] == ORC_PRIOR_NORMAL) {
      double z = (x - a) / b;
      lp += -(z * z + ORC_LOG2PI) / 2.0 - log(b);
    } else if (cfg->prior_kind[k] == ORC_PRIOR_EXPONENTIAL) {      /* Distributions.Exponential(theta): support x >= 0 */
      if (x >= 0.0) lp += -x / a - log(a);
      else return -INFINITY;
    } else if (cfg->prior_kind[k] == ORC_PRIOR_LOGNORMAL) {        /* support x > 0 */
      if (x > 0.0) {
        double lx = log(x), z = (lx - a) / b;
        lp += -(z * z + ORC_LOG2PI) / 2.0 - log(b) - lx;
      } else return -INFINITY;
    } else if (cfg->prior_kind[k] == ORC_PRIOR_GAMMA) {            /* Gamma(shape a, scale b) */
      if (x > 0.0) lp += (a - 1.0) * log(x) - x / b - (lgamma(a) + a * log(b));
      else return -INFINITY;
    } else if (cfg->prior_kind[k] == ORC_PRIOR_BETA) {
      if (x > 0.0 && x < 1.0) lp += (a - 1.0) * log(x) + (b - 1.0) * log1p(-x) - (lgamma(a) + lgamma(b) - lgamma(a + b));
      else return -INFINITY;
    } else if (cfg->prior_kind[k] == ORC_PRIOR_TRUNCNORMAL) {      /* truncated(Normal(a, b), c, d) */
      if (x >= cfg->prior_c[k] && x <= cfg->prior_d[k]) {
        double p_lo, mass, z = (x - a) / b;
        truncnormal_mass(cfg, k, &p_lo, &mass);
        lp += -(z * z + ORC_LOG2PI) / 2.0 - (log(b) + log(mass));
      } else return -INFINITY;
    } else {
      if (x >= a && x <= b) lp += -log(b - a);
      else return -INFINITY;
    }
  }
  return lp;
}

/* ------------------------------------------------------------------ */
/* Simulators (the user's f_dist, SimulatedAnnealingABC.jl:164,175,315) */
/* shipped as data-described models.  Definitions in DESIGN.md.        */
/* ------------------------------------------------------------------ */
static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

int orc_simulate(const orc_config *cfg, const double *th, uint64_t pid, uint64_t iter, double *rho) {
  const double *p = cfg->model_params;
  const int s = cfg->n_stats, d = cfg->n_para;
  nstream ns = ns_open(cfg->seed, pid, ORC_PURPOSE_SIM, iter);
  switch (cfg->model_id) {
  case ORC_MODEL_HOST: {
    /* the user's f_dist itself (SimulatedAnnealingABC.jl:175,315) */
    int64_t id = (int64_t)pid;
    if (!cfg->host_fn) return -1;
    return cfg->host_fn(cfg->host_ctx, th, &id, 1, iter, rho);
  }
  case ORC_MODEL_GAUSS_IID: {
    /* test/runtests.jl:35,86,128-131,167-170: y ~ Normal(theta1, sd)^n_obs;
       rho1 = |obs_mean - mean(y)|, rho2 = |obs_m2 - mean(y.^2)| */
    int n_obs = (int)p[0];
    double sd = (d >= 2) ? th[1] : p[1];
    double sum = 0.0, sum2 = 0.0;
    for (int k = 0; k < n_obs; ++k) {
      double x = th[0] + sd * ns_next(&ns);
      sum += x; sum2 += x * x;
    }
    rho[0] = fabs(p[2] - sum / n_obs);
    if (s >= 2) rho[1] = fabs(p[3] - sum2 / n_obs);
    return 0;
  }
  case ORC_MODEL_GAUSS2D: {
    int n_obs = (int)p[0];
    double r = p[1], c = sqrt(1.0 - r * r);
    double S1 = 0, S2 = 0, Q11 = 0, Q22 = 0, Q12 = 0;
    for (int k = 0; k < n_obs; ++k) {
      double za = ns_next(&ns), zb = ns_next(&ns);
      double e1 = za, e2 = r * za + c * zb;
      S1 += e1; S2 += e2; Q11 += e1 * e1; Q22 += e2 * e2; Q12 += e1 * e2;
    }
    double m1 = S1 / n_obs, m2 = S2 / n_obs;
    double var1 = (Q11 - S1 * m1) / (n_obs - 1), var2 = (Q22 - S2 * m2) / (n_obs - 1);
    double cov = (Q12 - S1 * m2) / (n_obs - 1);
    double d1 = th[0] + m1 - p[2], d2 = th[1] + m2 - p[3];
    rho[0] = sqrt(d1 * d1 + d2 * d2);
    rho[1] = fabs(var1 + var2 - p[4]);
    rho[2] = fabs(cov - p[5]);
    return 0;
  }
  case ORC_MODEL_GK: {
    int n_draws = (int)p[0];
    double c = p[1];
    double x[256];
    if (n_draws > 256) return -1;
    for (int k = 0; k < n_draws; ++k) {
      double z = ns_next(&ns);
      x[k] = th[0] + th[1] * (1.0 + c * tanh(th[2] * z / 2.0)) * pow(1.0 + z * z, th[3]) * z;
    }
    qsort(x, (size_t)n_draws, sizeof(double), cmp_double);
    for (int j = 0; j < s; ++j) {
      int rank = (int)p[2 + j];          /* 1-based order statistic */
      double v = fabs(x[rank - 1] - p[2 + s + j]);
      rho[j] = isfinite(v) ? v : 1e30;
    }
    return 0;
  }
  case ORC_MODEL_LV: {
    int n_steps = (int)p[0];
    double dt = p[1], sg = p[2], X = p[3], Y = p[4];
    double sq = sqrt(dt);
    double SX = 0, QX = 0, SY = 0, QY = 0;
    for (int t = 0; t < n_steps; ++t) {
      double z1 = ns_next(&ns), z2 = ns_next(&ns);
      /* dX = (aX - bXY) dt + sigma X sqrt(dt) z1 = X ((a - bY) dt + sigma sqrt(dt) z1): the factored form, both
         species from the OLD state */
      if (g_literal) {         /* Euler-Maruyama as written: X += (aX - bXY) dt + sigma X dW1,  Y += (bXY - cY) dt + sigma Y dW2 */
        double dW1 = sq * z1, dW2 = sq * z2;
        double nX = X + (th[0] * X - th[1] * X * Y) * dt + sg * X * dW1;
        double nY = Y + (th[1] * X * Y - th[2] * Y) * dt + sg * Y * dW2;
        X = fmax(nX, 0.0); Y = fmax(nY, 0.0);
      } else {
      double fx = (th[0] - th[1] * Y) * dt + sg * sq * z1;
      double fy = (th[1] * X - th[2]) * dt + sg * sq * z2;
      X = fmax(X + X * fx, 0.0); Y = fmax(Y + Y * fy, 0.0);
      }
      SX += X; QX += X * X; SY += Y; QY += Y * Y;
    }
    double mX = SX / n_steps, mY = SY / n_steps;
    double vX = (QX - SX * mX) / (n_steps - 1), vY = (QY - SY * mY) / (n_steps - 1);
    double st[4] = { mX, sqrt(fmax(vX, 0.0)), mY, sqrt(fmax(vY, 0.0)) };
    for (int j = 0; j < s; ++j) {
      double v = fabs(st[j] - p[5 + j]);
      rho[j] = isfinite(v) ? v : 1e30;
    }
    return 0;
  }
  default:
    return -1;
  }
}

/* ------------------------------------------------------------------ */
/* cdf_estimators.jl:23-44  build_cdf(x::AbstractVector)               */
/* ------------------------------------------------------------------ */
int64_t orc_build_cdf(const double *x, int64_t n, double *knots) {
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i)          /* :29 filter(e -> e > 0, x) */
    if (x[i] > 0) knots[1 + m++] = x[i];
  if (m == 0) return ORC_ERR_EMPTY_CDF;    /* :33 maximum(x) of empty -> throws */
  qsort(knots + 1, (size_t)m, sizeof(double), cmp_double);
  knots[0] = 0.0;                          /* :33 [0; sort(x); maximum(x)*a] */
  knots[m + 1] = knots[m] * 1.5;           /* :32 a = 1.5 */
  return m + 2;
}

/* cdf_estimators.jl:36-42: ordinates range(0,1,length), LinearMonotonicInterpolation
   (Interpolations.jl ^0.15: piecewise linear, interval picked as
   searchsortedfirst(knots,x)-1 clamped to the first knot), Flat() extrapolation. */
double orc_cdf_apply(const double *knots, int64_t len, double x) {
  if (!(x >= knots[0])) return (x != x) ? x : 0.0;       /* Flat below; NaN propagates */
  if (x > knots[len - 1]) return 1.0;                    /* Flat above */
  int64_t lo = 0, hi = len;                              /* j = #knots < x */
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (knots[mid] < x) lo = mid + 1; else hi = mid;
  }
  int64_t i0 = lo > 0 ? lo - 1 : 0;
  double L1 = (double)(len - 1);
  double y0 = (double)i0 / L1, y1 = (double)(i0 + 1) / L1;
  /* weight form t in [0, 1] (what a linear interpolant evaluates); the slope form overflows to inf * 0 = NaN when two
     knots are closer than ~1e-308 / len */
  if (g_literal) return y0 + (x - knots[i0]) * ((y1 - y0) / (knots[i0 + 1] - knots[i0]));   /* slope form */
  double t = (x - knots[i0]) / (knots[i0 + 1] - knots[i0]);
  return y0 + t * (y1 - y0);
}

/* ------------------------------------------------------------------ */
/* SimulatedAnnealingABC.jl:92-95  update_epsilon_single_eps           */
/* root of eps^2 + v eps^1.5 - ubar^2 on (0, ubar); Roots.find_zero    */
/* bracketing -> any convergent bracketing method gives the same root. */
/* ------------------------------------------------------------------ */
double orc_eps_single(double ubar, double v) {
  if (ubar <= DBL_EPSILON) return 0.0;                   /* :93 ubar <= eps() */
  double u2 = ubar * ubar;
  double lo = 0.0, hi = ubar;                            /* f(lo)<0<f(hi) */
  double e = ubar;
  for (int it = 0; it < 200; ++it) {
    double se = sqrt(e);
    double f = e * e + v * e * se - u2;
    if (f > 0) hi = e; else lo = e;
    double fp = 2.0 * e + 1.5 * v * se;
    double en = e - f / fp;
    if (!(en > lo && en < hi)) en = 0.5 * (lo + hi);     /* safeguard */
    if (en == e || fabs(en - e) <= 2.0 * DBL_EPSILON * fabs(en)) { e = en; break; }
    e = en;
  }
  return e;
}

/* mean of the density ~ exp(-beta u) on [0,1]: (1-e^-b(1+b))/(b(1-e^-b)), :113 */
static double trunc_exp_mean(double b) {
  if (fabs(b) < 1e-3) { double b2 = b * b; return 0.5 - b / 12.0 + b * b2 / 720.0 - b * b2 * b2 / 30240.0; }
  return 1.0 / b - 1.0 / expm1(b);
}
static double trunc_exp_mean_deriv(double b) {
  if (fabs(b) < 1e-3) { double b2 = b * b; return -1.0 / 12.0 + b2 / 240.0 - b2 * b2 / 6048.0; }
  double em = expm1(b);
  return -1.0 / (b * b) + (em + 1.0) / (em * em);
}

/* root beta of trunc_exp_mean(beta) = ubar_i (SimulatedAnnealingABC.jl:113, start 1/ubar_i) */
double orc_multi_eps_beta(double ub) {
  if (ub == 0.5) return 0.0;
  if (ub > 0.5) return -orc_multi_eps_beta(1.0 - ub);    /* m(-b) = 1 - m(b) */
  double lo = 0.0, hi = 1.0 / ub;                        /* m(lo)=.5>ub, m(hi)<1/hi=ub */
  double b = hi;
  for (int it = 0; it < 300; ++it) {
    double f = trunc_exp_mean(b) - ub;
    if (f > 0) lo = b; else hi = b;
    double bn = b - f / trunc_exp_mean_deriv(b);
    if (!(bn > lo && bn < hi)) bn = 0.5 * (lo + hi);
    if (bn == b || fabs(bn - b) <= 2.0 * DBL_EPSILON * fabs(bn)) { b = bn; break; }
    b = bn;
  }
  return b;
}

/* SimulatedAnnealingABC.jl:100-117  update_epsilon_multi_eps */
int orc_eps_multi(const double *ubar, int s, double v, double *eps_out) {
  /* :103 cn = (2n+2)!/((n+1)!(n+2)!) */
  double cn = 1.0;
  for (int k = 1; k <= s + 1; ++k) cn = cn * (double)(s + 1 + k) / (double)k;  /* C(2s+2, s+1) */
  cn /= (double)(s + 2);
  for (int i = 0; i < s; ++i) {
    double ui = ubar[i];
    if (ui <= DBL_EPSILON) return ORC_ERR_ZERO_MEAN_U;   /* :107-109 */
    double num = 1.0, prodq = 1.0;
    for (int j = 0; j < s; ++j) {                        /* :110-112 */
      double q = ubar[j] / ui;
      num += pow(q, s / 2.0);
      prodq *= q;
    }
    double den = cn * (s + 1) * pow(ui, 1.0 + s / 2.0) * prodq;
    double beta = orc_multi_eps_beta(ui);                /* :113 */
    eps_out[i] = 1.0 / (beta + v * num / den);           /* :114 */
  }
  return 0;
}

/* Cholesky (implicit in MvNormal(zeros, Sigma), proposals.jl:42); row-major lower */
int orc_cholesky(const double *a, int d, double *l) {
  memset(l, 0, sizeof(double) * (size_t)(d * d));
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      double sum = a[i * d + j];
      for (int k = 0; k < j; ++k) sum -= l[i * d + k] * l[j * d + k];
      if (i == j) {
        if (!(sum > 0.0)) return ORC_ERR_NOT_POSDEF;
        l[i * d + i] = sqrt(sum);
      } else l[i * d + j] = sum / l[j * d + j];
    }
  return 0;
}

/* ------------------------------------------------------------------ */
/* state plumbing                                                      */
/* ------------------------------------------------------------------ */
int orc_create(const orc_config *cfg, orc_state **out) {
  *out = NULL;
  if (cfg->n_particles < 1 || cfg->n_para < 1 || cfg->n_para > ORC_MAX_PARA ||
      cfg->n_stats < 1 || cfg->n_stats > ORC_MAX_STATS) return ORC_ERR_BAD_CONFIG;
  if (!(cfg->algorithm == ORC_ALG_SINGLE_EPS || cfg->algorithm == ORC_ALG_MULTI_EPS))
    return ORC_ERR_BAD_ALGORITHM;                        /* :462-464 */
  orc_state *st = (orc_state *)calloc(1, sizeof(orc_state));
  st->cfg = *cfg;
  st->n = cfg->n_particles; st->d = cfg->n_para; st->s = cfg->n_stats;
  size_t n = (size_t)st->n;
  st->theta = (double *)calloc(n * st->d, sizeof(double));
  st->u = (double *)calloc(n * st->s, sizeof(double));
  st->rho = (double *)calloc(n * st->s, sizeof(double));
  st->knots = (double *)calloc((n + 2) * st->s, sizeof(double));
  st->eps_len = cfg->algorithm == ORC_ALG_MULTI_EPS ? st->s : 1;
  for (int i = 0; i < st->d * st->d; ++i) st->sigma[i] = -1.0;  /* proposals.jl:32,34 sentinel */
  *out = st;
  return 0;
}

void orc_destroy(orc_state *st) {
  if (!st) return;
  free(st->theta); free(st->u); free(st->rho); free(st->knots);
  free(st->eps_hist); free(st->u_hist); free(st->rho_hist);
  free(st);
}

const double *orc_theta(const orc_state *st) { return st->theta; }
const double *orc_u(const orc_state *st) { return st->u; }
const double *orc_rho(const orc_state *st) { return st->rho; }
void orc_counters(const orc_state *st, int64_t out[4]) {
  out[0] = st->n_simulation; out[1] = st->n_accept; out[2] = st->n_resampling; out[3] = st->n_population_updates;
}
int orc_epsilon(const orc_state *st, double *out) {
  for (int i = 0; i < st->eps_len; ++i) out[i] = st->eps[i];
  return st->eps_len;
}
int64_t orc_history_len(const orc_state *st) { return st->hist_len; }
void orc_history(const orc_state *st, double *e, double *u, double *r) {
  memcpy(e, st->eps_hist, sizeof(double) * (size_t)(st->hist_len * st->eps_len));
  memcpy(u, st->u_hist, sizeof(double) * (size_t)(st->hist_len * st->s));
  memcpy(r, st->rho_hist, sizeof(double) * (size_t)(st->hist_len * st->s));
}
int64_t orc_cdf_len(const orc_state *st, int j) { return st->cdf_len[j]; }
const double *orc_cdf_knots(const orc_state *st, int j) { return st->knots + (size_t)j * (size_t)(st->n + 2); }
double orc_last_ess(const orc_state *st) { return st->last_ess; }
void orc_proposal_sigma(const orc_state *st, double *out) { memcpy(out, st->sigma, sizeof(double) * (size_t)(st->d * st->d)); }

static void col_means(const double *a, int64_t n, int s, double *out) {
  for (int j = 0; j < s; ++j) {
    double acc = 0.0;
    const double *col = a + (size_t)j * (size_t)n;
    for (int64_t i = 0; i < n; ++i) acc += col[i];
    out[j] = acc / (double)n;
  }
}

/* push!(eps_history, ...), push!(u_history, ...), push!(rho_history, ...) :367-372 */
static void push_history(orc_state *st, int push_rho_only) {
  if (st->hist_len == st->hist_cap) {
    st->hist_cap = st->hist_cap ? st->hist_cap * 2 : 16;
    st->eps_hist = (double *)realloc(st->eps_hist, sizeof(double) * (size_t)(st->hist_cap * st->eps_len));
    st->u_hist = (double *)realloc(st->u_hist, sizeof(double) * (size_t)(st->hist_cap * st->s));
    st->rho_hist = (double *)realloc(st->rho_hist, sizeof(double) * (size_t)(st->hist_cap * st->s));
  }
  (void)push_rho_only;
  memcpy(st->eps_hist + st->hist_len * st->eps_len, st->eps, sizeof(double) * (size_t)st->eps_len);
  col_means(st->u, st->n, st->s, st->u_hist + st->hist_len * st->s);
  col_means(st->rho, st->n, st->s, st->rho_hist + st->hist_len * st->s);
  st->hist_len++;
}

/* ------------------------------------------------------------------ */
/* SimulatedAnnealingABC.jl:124-137  resample_population               */
/* StatsBase.sample(1:n, weights(w), n, replace=true) = n iid          */
/* categorical draws; realised by inverse-CDF on the running sum.      */
/* rho is NOT permuted (:131-132 permute population and u only).       */
/* ------------------------------------------------------------------ */
/* ------------------------------------------------------------------ */
/* Running sum of the resample weights.  sample(1:n, weights(w), n) (:129) is restated as n inverse-CDF draws  */
/* on the running weight sum; WHICH floating-point running sum is part of the shared stream specification      */
/* (DESIGN.md "RNG streams"), like the Philox counters: a sequential sum and a blocked one differ in the last    */
/* bits, and at n ~ 5e6 that moves a few draws per resample to the neighbouring particle.  The order below is    */
/* the one a 3-pass blocked scan produces (chunks of 1024 weights, 256 "threads" x 4 weights, 64-wide trees):    */
/*   pass 1  chunk sums:   per thread ((w0+w1)+w2)+w3; per 64 threads the tree r[l] += r[l+off], off = 32..1;    */
/*                         per chunk ((t0+t1)+t2)+t3                                                             */
/*   pass 2  chunk offsets: 1024 owners of ceil(nb/1024) consecutive chunks each, sequential inside an owner,    */
/*                         Hillis-Steele inclusive scan (offsets 1,2,..,512) across the owners; total = last     */
/*   pass 3  inside a chunk: exclusive sequential scan of the 256 thread sums, then offset + running sum of 4    */
/* ------------------------------------------------------------------ */
#define ORC_SCAN_CHUNK 1024
int64_t orc_scan_chunks(int64_t n) { return (n + ORC_SCAN_CHUNK - 1) / ORC_SCAN_CHUNK; }

void orc_weight_scan(const double *w, int64_t n, double *cum, double *bs, double totals[2]) {
  const int64_t nb = orc_scan_chunks(n);
  double *raw = (double *)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
  double W2 = 0.0;
  for (int64_t b = 0; b < nb; ++b) {                                   /* pass 1 */
    double th[256];
    for (int t = 0; t < 256; ++t) {
      double sacc = 0.0;
      for (int e = 0; e < 4; ++e) {
        const int64_t i = b * ORC_SCAN_CHUNK + 4 * t + e;
        const double wi = i < n ? w[i] : 0.0;
        sacc += wi; W2 += wi * wi;
      }
      th[t] = sacc;
    }
    double wave[4];
    for (int v = 0; v < 4; ++v) {
      double r[64];
      for (int l = 0; l < 64; ++l) r[l] = th[64 * v + l];
      for (int off = 32; off >= 1; off >>= 1)
        for (int l = 0; l < off; ++l) r[l] += r[l + off];
      wave[v] = r[0];
    }
    raw[b] = ((wave[0] + wave[1]) + wave[2]) + wave[3];
  }
  {                                                                    /* pass 2 */
    const int64_t per = (nb + 1023) / 1024;
    double a[2][1024];
    for (int t = 0; t < 1024; ++t) {
      const int64_t lo = (int64_t)t * per, hi = lo + per < nb ? lo + per : nb;
      double sacc = 0.0;
      for (int64_t b = lo; b < hi; ++b) sacc += raw[b];
      a[0][t] = sacc;
    }
    int cur = 0;
    for (int off = 1; off < 1024; off <<= 1) {
      for (int t = 0; t < 1024; ++t) a[1 - cur][t] = t >= off ? a[cur][t] + a[cur][t - off] : a[cur][t];
      cur = 1 - cur;
    }
    totals[0] = a[cur][1023];
    totals[1] = W2;
    for (int t = 0; t < 1024; ++t) {
      const int64_t lo = (int64_t)t * per, hi = lo + per < nb ? lo + per : nb;
      double run = t > 0 ? a[cur][t - 1] : 0.0;
      for (int64_t b = lo; b < hi; ++b) { const double v = raw[b]; bs[b] = run; run += v; }
    }
  }
  for (int64_t b = 0; b < nb; ++b) {                                   /* pass 3 */
    double th[256], ex[256];
    for (int t = 0; t < 256; ++t) {
      double sacc = 0.0;
      for (int e = 0; e < 4; ++e) { const int64_t i = b * ORC_SCAN_CHUNK + 4 * t + e; sacc += i < n ? w[i] : 0.0; }
      th[t] = sacc;
    }
    double run = 0.0;
    for (int t = 0; t < 256; ++t) { ex[t] = run; run += th[t]; }
    for (int t = 0; t < 256; ++t) {
      double r = bs[b] + ex[t];
      for (int e = 0; e < 4; ++e) {
        const int64_t i = b * ORC_SCAN_CHUNK + 4 * t + e;
        r += i < n ? w[i] : 0.0;
        if (i < n) cum[i] = r;
      }
    }
  }
  free(raw);
}

/* first k with cum[k] > t, searched as the scan is laid out: the chunk by its offset, then inside the chunk */
int64_t orc_resample_index(const double *cum, const double *bs, int64_t n, double t) {
  const int64_t nb = orc_scan_chunks(n);
  int64_t blo = 0, bhi = nb;
  while (blo < bhi) { const int64_t mid = blo + ((bhi - blo) >> 1); if (bs[mid] > t) bhi = mid; else blo = mid + 1; }
  const int64_t chunk = blo - 1;
  int64_t lo = chunk * ORC_SCAN_CHUNK, hi = lo + ORC_SCAN_CHUNK;
  if (hi > n) hi = n;
  while (lo < hi) { const int64_t mid = lo + ((hi - lo) >> 1); if (cum[mid] > t) hi = mid; else lo = mid + 1; }
  return lo < n ? lo : n - 1;
}

static int resample_population(orc_state *st, double delta, uint64_t iter) {
  const int64_t n = st->n; const int s = st->s, d = st->d;
  double ubar[ORC_MAX_STATS];
  col_means(st->u, n, s, ubar);                                      /* :126 */
  double *w = (double *)malloc(sizeof(double) * (size_t)n);
  double *cum = (double *)malloc(sizeof(double) * (size_t)n);
  double *bs = (double *)malloc(sizeof(double) * (size_t)orc_scan_chunks(n));
  for (int64_t i = 0; i < n; ++i) {                                  /* :127 */
    double acc = 0.0;
    for (int j = 0; j < s; ++j) acc += st->u[(size_t)j * n + i] * delta / ubar[j];
    w[i] = exp(-acc);
  }
  double totals[2];
  orc_weight_scan(w, n, cum, bs, totals);
  double *th2 = (double *)malloc(sizeof(double) * (size_t)(n * d));
  double *u2 = (double *)malloc(sizeof(double) * (size_t)(n * s));
  for (int64_t i = 0; i < n; ++i) {                                  /* :129 */
    uint32_t w4[4];
    orc_stream_block(st->cfg.seed, (uint64_t)i, ORC_PURPOSE_RESAMPLE, iter, 0, w4);
    const double t = orc_u52(w4[0], w4[1]) * totals[0];
    const int64_t idx = orc_resample_index(cum, bs, n, t);
    for (int k = 0; k < d; ++k) th2[(size_t)k * n + i] = st->theta[(size_t)k * n + idx];   /* :131 */
    for (int j = 0; j < s; ++j) u2[(size_t)j * n + i] = st->u[(size_t)j * n + idx];       /* :132 */
  }
  memcpy(st->theta, th2, sizeof(double) * (size_t)(n * d));
  memcpy(st->u, u2, sizeof(double) * (size_t)(n * s));
  st->last_ess = totals[0] * totals[0] / totals[1];                  /* :134 */
  free(w); free(cum); free(bs); free(th2); free(u2);
  return 0;
}

/* eps update dispatch, :200-204 and :350-354 */
static int update_epsilon(orc_state *st, double v) {
  const int64_t n = st->n; const int s = st->s;
  if (st->cfg.algorithm == ORC_ALG_MULTI_EPS) {
    double ubar[ORC_MAX_STATS];
    col_means(st->u, n, s, ubar);
    int rc = orc_eps_multi(ubar, s, v, st->eps);
    if (rc) return fail(st, rc, "Division by zero - Mean u for a statistic is <= eps()");
  } else {
    double acc = 0.0;                                                /* mean(u) over all n*s entries */
    for (size_t i = 0; i < (size_t)n * (size_t)s; ++i) acc += st->u[i];
    st->eps[0] = orc_eps_single(acc / ((double)n * (double)s), v);
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* SimulatedAnnealingABC.jl:151-227  initialization                    */
/* ------------------------------------------------------------------ */
int orc_initialize(orc_state *st, int64_t n_simulation) {
  const int64_t n = st->n; const int s = st->s, d = st->d;
  if (n_simulation < n) return fail(st, ORC_ERR_NSIM_TOO_SMALL, "`n_simulation` is too small for the number of particles.");  /* :155 */
  int bad = 0;
  /* :172-179 prior sample + simulate (the :163-165 shape probe is not needed: s is data) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) reduction(|:bad)
#endif
  for (int64_t i = 0; i < n; ++i) {
    double th[ORC_MAX_PARA], r[ORC_MAX_STATS];
    orc_prior_sample(&st->cfg, (uint64_t)i, th);
    if (orc_simulate(&st->cfg, th, (uint64_t)i, 0, r)) bad |= 2;
    for (int k = 0; k < d; ++k) st->theta[(size_t)k * n + i] = th[k];
    for (int j = 0; j < s; ++j) { st->rho[(size_t)j * n + i] = r[j]; if (r[j] < 0) bad |= 1; }
  }
  if (bad & 2) return fail(st, ORC_ERR_BAD_CONFIG, "unknown model or bad model parameters");
  if (bad & 1) return fail(st, ORC_ERR_NEG_DISTANCE, "Negative distances are not allowed!");   /* :185 */
  /* :187 build_cdf per column, :190-192 u = cdf(rho) */
  for (int j = 0; j < s; ++j) {
    double *kn = st->knots + (size_t)j * (size_t)(n + 2);
    int64_t len = orc_build_cdf(st->rho + (size_t)j * n, n, kn);
    if (len < 0) return fail(st, (int)len, "all prior distances of one statistic are zero");
    st->cdf_len[j] = len;
    for (int64_t i = 0; i < n; ++i) st->u[(size_t)j * n + i] = orc_cdf_apply(kn, len, st->rho[(size_t)j * n + i]);
  }
  /* :180 rho_history first row is taken before resampling; rho is never permuted,
     so taking it in push_history below gives the same numbers. */
  resample_population(st, st->cfg.delta, 0);                         /* :197 */
  int rc = update_epsilon(st, st->cfg.v);                            /* :200-204 */
  if (rc) return rc;
  st->hist_len = 0;
  push_history(st, 0);                                               /* :180,207-208 */
  st->n_simulation = n; st->n_accept = 0; st->n_resampling = 1; st->n_population_updates = 0;  /* :213,223 */
  st->initialized = 1;
  return 0;
}

/* ------------------------------------------------------------------ */
/* proposals.jl:46-48,58-60  update_proposal!(::RandomWalk)            */
/* ------------------------------------------------------------------ */
static int update_proposal(orc_state *st, const orc_update_args *a) {
  if (a->proposal_kind != ORC_PROP_RANDOMWALK) return 0;             /* :116,150 nothing */
  const int64_t n = st->n; const int d = st->d;
  double mean[ORC_MAX_PARA];
  col_means(st->theta, n, d, mean);
  for (int k = 0; k < d; ++k)
    for (int l = 0; l <= k; ++l) {
      double acc = 0.0;
      const double *ck = st->theta + (size_t)k * n, *cl = st->theta + (size_t)l * n;
      for (int64_t i = 0; i < n; ++i) acc += (ck[i] - mean[k]) * (cl[i] - mean[l]);
      double c = acc / (double)(n - 1);                              /* cov, corrected */
      st->sigma[k * d + l] = st->sigma[l * d + k] = c;
    }
  if (d == 1) {
    st->sigma[0] = a->proposal_p0 * st->sigma[0];                    /* :59 beta*cov(population) */
    st->chol[0] = sqrt(st->sigma[0]);                                /* :54 sqrt(rw.Sigma) */
    return 0;
  }
  for (int k = 0; k < d; ++k)
    for (int l = 0; l < d; ++l)
      st->sigma[k * d + l] = a->proposal_p0 * (st->sigma[k * d + l] + (k == l ? 1e-8 : 0.0));  /* :47 */
  if (orc_cholesky(st->sigma, d, st->chol)) return fail(st, ORC_ERR_NOT_POSDEF, "proposal covariance is not positive definite");
  return 0;
}

/* ------------------------------------------------------------------ */
/* proposal call: proposals.jl:40-43,52-55 (RandomWalk),               */
/* :101-114 (DifferentialEvolution), :137-148 (StretchMove)            */
/* ------------------------------------------------------------------ */
static double propose(const orc_state *st, const orc_update_args *a, int64_t i, int64_t inact_lo, int64_t inact_n,
                      uint64_t iter, double *thp) {
  const int64_t n = st->n; const int d = st->d;
  const uint64_t seed = st->cfg.seed;
  double th[ORC_MAX_PARA];
  for (int k = 0; k < d; ++k) th[k] = st->theta[(size_t)k * n + i];
  switch (a->proposal_kind) {
  case ORC_PROP_RANDOMWALK: {
    nstream ns = ns_open(seed, (uint64_t)i, ORC_PURPOSE_PROP, iter);
    double z[ORC_MAX_PARA];
    for (int k = 0; k < d; ++k) z[k] = ns_next(&ns);
    for (int k = 0; k < d; ++k) {                                    /* theta + L z */
      double acc = 0.0;
      for (int l = 0; l <= k; ++l) acc += st->chol[k * d + l] * z[l];
      thp[k] = th[k] + acc;
    }
    return 0.0;
  }
  case ORC_PROP_DIFFEVO: {
    uint64_t i1 = 0, i2 = 0;
    for (uint32_t t = 0;; ++t) {                                     /* :103-107 */
      uint32_t w[4];
      orc_stream_block(seed, (uint64_t)i, ORC_PURPOSE_PROP, iter, t, w);
      i1 = mulhi64(((uint64_t)w[0] << 32) | w[1], (uint64_t)inact_n);
      i2 = mulhi64(((uint64_t)w[2] << 32) | w[3], (uint64_t)inact_n);
      if (i1 != i2) break;
    }
    double z[2];
    orc_normal_pair(seed, (uint64_t)i, ORC_PURPOSE_PROP2, iter, 0, z);
    double gamma = a->proposal_p0 * (1.0 + a->proposal_p1 * z[0]);  /* :110 */
    for (int k = 0; k < d; ++k) {
      const double *col = st->theta + (size_t)k * n + inact_lo;
      thp[k] = th[k] + gamma * (col[i1] - col[i2]);                  /* :113 */
    }
    return 0.0;
  }
  case ORC_PROP_STRETCH: {
    uint32_t w[4];
    orc_stream_block(seed, (uint64_t)i, ORC_PURPOSE_PROP, iter, 0, w);
    uint64_t ip = mulhi64(((uint64_t)w[0] << 32) | w[1], (uint64_t)inact_n);   /* :141 */
    double U = orc_u52(w[2], w[3]);
    double aa = a->proposal_p0;
    double t = (aa - 1.0) * U + 1.0;
    double z = t * t / aa;                                           /* :144 */
    for (int k = 0; k < d; ++k) {
      double pk = st->theta[(size_t)k * n + inact_lo + (int64_t)ip];
      thp[k] = pk + z * (th[k] - pk);                                /* :147 */
    }
    return log(z) * (double)(d - 1);                                 /* :146 */
  }
  default:
    for (int k = 0; k < d; ++k) thp[k] = th[k];
    return 0.0;
  }
}

/* SimulatedAnnealingABC.jl:308-331, one particle of the active batch */
static int particle_body(orc_state *st, const orc_update_args *a, int64_t i, int64_t inact_lo, int64_t inact_n,
                         uint64_t iter) {
  const int64_t n = st->n; const int d = st->d, s = st->s;
  double thp[ORC_MAX_PARA] = {0}, th[ORC_MAX_PARA], rp[ORC_MAX_STATS] = {0}, up[ORC_MAX_STATS] = {0};
  double logf = propose(st, a, i, inact_lo, inact_n, iter, thp);    /* :311 */
  double lpp = orc_prior_logpdf(&st->cfg, thp);
  double log_accept;
  if (lpp > -INFINITY) {                                             /* :314 */
    orc_simulate(&st->cfg, thp, (uint64_t)i, iter, rp);              /* :315 */
    for (int k = 0; k < d; ++k) th[k] = st->theta[(size_t)k * n + i];
    double acc = 0.0;
    for (int j = 0; j < s; ++j) {
      up[j] = orc_cdf_apply(st->knots + (size_t)j * (size_t)(n + 2), st->cdf_len[j], rp[j]);   /* :316 */
      double e = st->eps_len == 1 ? st->eps[0] : st->eps[j];
      acc += (st->u[(size_t)j * n + i] - up[j]) / e;                 /* :319 */
    }
    log_accept = lpp - orc_prior_logpdf(&st->cfg, th) + acc + logf; /* :318-319 */
  } else {
    log_accept = -INFINITY;                                          /* :321 */
  }
  uint32_t w[4];
  orc_stream_block(st->cfg.seed, (uint64_t)i, ORC_PURPOSE_ACCEPT, iter, 0, w);
  if (log(orc_u52(w[0], w[1])) < log_accept) {                       /* :324 */
    for (int k = 0; k < d; ++k) st->theta[(size_t)k * n + i] = thp[k];      /* :325 */
    for (int j = 0; j < s; ++j) { st->u[(size_t)j * n + i] = up[j]; st->rho[(size_t)j * n + i] = rp[j]; }  /* :326-327 */
    return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* SimulatedAnnealingABC.jl:251-402  update_population!                */
/* ------------------------------------------------------------------ */
int orc_update(orc_state *st, const orc_update_args *a) {
  if (!st->initialized) return fail(st, ORC_ERR_BAD_CONFIG, "not initialized");
  if (!(a->v > 0)) return fail(st, ORC_ERR_BAD_V, "Annealing speed `v` must be positive.");              /* :261 */
  if (!(a->delta > 0)) return fail(st, ORC_ERR_BAD_DELTA, "Resamping intensity `δ` must be positive.");  /* :262 */
  if (a->proposal_kind == ORC_PROP_RANDOMWALK && !(a->proposal_p0 > 0 && a->proposal_p0 <= 1))
    return fail(st, ORC_ERR_BAD_BETA, "Mixing parameter `β` must be between zero and one.");           /* proposals.jl:30 */
  const int64_t n = st->n;
  const int64_t n_pop = a->n_simulation / n;                         /* :275 */
  const int64_t n_updates = n_pop * n;                               /* :276 */
  int64_t last_checkpoint = 0;                                       /* :277 */
  /* `ix % checkpoint_history` (:367) is a DivideError for 0; Julia's rem by a negative interval is that by its magnitude */
  if (n_pop > 0 && a->checkpoint_history == 0)
    return fail(st, ORC_ERR_BAD_CONFIG, "DivideError: integer division error (`checkpoint_history` must not be zero)");
  const int64_t cph = a->checkpoint_history < 0 ? -a->checkpoint_history : (a->checkpoint_history > 0 ? a->checkpoint_history : 1);
  int rc = update_proposal(st, a);                                   /* :284 */
  if (rc) return rc;
  const int64_t half = n / 2;
  if (n_pop > 0 && a->proposal_kind == ORC_PROP_DIFFEVO && half < 2)
    return fail(st, ORC_ERR_BAD_CONFIG, "DifferentialEvolution needs at least two particles per half");
  if (n_pop > 0 && a->proposal_kind == ORC_PROP_STRETCH && half < 1)
    return fail(st, ORC_ERR_BAD_CONFIG, "StretchMove needs at least one particle per half");
  int64_t n_accept = st->n_accept, n_resampling = st->n_resampling;
  for (int64_t ix = 1; ix <= n_pop; ++ix) {                          /* :294 */
    const uint64_t iter = (uint64_t)(st->n_population_updates + ix);
    int64_t acc_tmp = 0;
    const int64_t lo[2] = { 0, half }, cnt[2] = { half, n - half }; /* :300-301 */
    for (int b = 0; b < 2; ++b) {                                    /* :304 */
      const int64_t a_lo = lo[b], a_n = cnt[b], i_lo = lo[1 - b], i_n = cnt[1 - b];
      int64_t acc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) reduction(+:acc)
#endif
      for (int64_t i = a_lo; i < a_lo + a_n; ++i)                    /* :308 */
        acc += particle_body(st, a, i, i_lo, i_n, iter);
      acc_tmp += acc;
    }
    n_accept += acc_tmp;                                             /* :334 */
    if ((double)n_accept >= (double)(n_resampling + 1) * a->resample) {   /* :340 */
      resample_population(st, a->delta, iter);                       /* :341 */
      n_resampling += 1;                                             /* :342 */
    }
    rc = update_proposal(st, a);                                     /* :348 */
    if (rc) return rc;
    rc = update_epsilon(st, a->v);                                   /* :350-354 */
    if (rc) return rc;
    if (ix % cph == 0) {                                             /* :367-372 */
      push_history(st, 0);
      last_checkpoint = ix;
    }
  }
  if (last_checkpoint != n_pop) push_history(st, 0);                 /* :378-382 */
  st->n_simulation += n_updates;                                     /* :391 */
  st->n_accept = n_accept;                                           /* :392 */
  st->n_resampling = n_resampling;                                   /* :393 */
  st->n_population_updates += n_pop;                                 /* :394 */
  return 0;
}
