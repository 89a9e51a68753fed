/* demo.c -- a plain-C client of libsabc_hip.so: no Python, no torch, nothing but include/sabc_hip.h.
 * It is what a host in any language with a C FFI does (Julia's ccall in INTEGRATION.md): sabc() as
 * create + initialize + update, then read the result back.  tests/test_c_abi.py compiles it with gcc,
 * checks that it links against the library (CPU) and, on the GPU box, runs it and compares its output
 * with the oracle on the same seed.
 *
 * usage: demo n_particles n_simulation seed          (BASELINE configs[0]/[1]: 1-D Gaussian mean, RandomWalk)
 * prints one line:  n_accept n_resampling n_population_updates eps mean var
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sabc_hip.h"

static int fail(sabc_handle *h, const char *what, int rc) {
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, h ? sabc_last_error(h) : sabc_last_global_error());
  if (h) sabc_destroy(h);
  return 1;
}

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: demo n_particles n_simulation seed obs_mean\n"); return 2; }
  const int64_t n = atoll(argv[1]), n_simulation = atoll(argv[2]);
  if (sabc_abi_version() != SABC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }

  sabc_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.abi_version = SABC_ABI_VERSION;
  cfg.device = 0;
  cfg.n_particles = n;
  cfg.n_para = 1;
  cfg.n_stats = 1;
  cfg.model_id = SABC_MODEL_GAUSS_IID;          /* f_dist: |mean(y_obs) - mean(x)|, x_1..100 ~ N(theta, 1) */
  cfg.n_model_params = 4;
  cfg.model_params[0] = 100; cfg.model_params[1] = 1.0; cfg.model_params[2] = atof(argv[4]); cfg.model_params[3] = 0.0;
  cfg.prior_kind[0] = SABC_PRIOR_NORMAL; cfg.prior_a[0] = 0.0; cfg.prior_b[0] = 2.0;
  cfg.algorithm = SABC_ALG_SINGLE_EPS;
  cfg.rank = 0; cfg.world = 1;
  cfg.v = 1.0; cfg.delta = 0.1;
  cfg.seed = strtoull(argv[3], NULL, 10);

  sabc_handle *h = NULL;
  int rc = sabc_create(&cfg, &h);
  if (rc) return fail(NULL, "sabc_create", rc);
  if ((rc = sabc_initialize(h, n_simulation))) return fail(h, "sabc_initialize", rc);

  sabc_update_args up;
  memset(&up, 0, sizeof(up));
  up.n_simulation = n_simulation - n;            /* sabc(): n_sim_remaining, SimulatedAnnealingABC.jl:476 */
  up.v = 1.0; up.delta = 0.1;
  up.resample = 2.0 * (double)n;
  up.checkpoint_history = 1;
  up.proposal_kind = SABC_PROP_RANDOMWALK;
  up.proposal_p0 = 0.8;
  if ((rc = sabc_update(h, &up))) return fail(h, "sabc_update", rc);

  double *theta = malloc(sizeof(double) * (size_t)n), *u = malloc(sizeof(double) * (size_t)n), *rho = malloc(sizeof(double) * (size_t)n);
  if ((rc = sabc_get_population(h, theta, u, rho))) return fail(h, "sabc_get_population", rc);
  int64_t c[4];
  double eps[SABC_MAX_STATS];
  int32_t eps_len = 0;
  if ((rc = sabc_get_counters(h, c))) return fail(h, "sabc_get_counters", rc);
  if ((rc = sabc_get_epsilon(h, eps, &eps_len))) return fail(h, "sabc_get_epsilon", rc);
  double m = 0.0, q = 0.0;
  for (int64_t i = 0; i < n; ++i) m += theta[i];
  m /= (double)n;
  for (int64_t i = 0; i < n; ++i) q += (theta[i] - m) * (theta[i] - m);
  printf("%lld %lld %lld %.17g %.17g %.17g\n", (long long)c[1], (long long)c[2], (long long)c[3], eps[0], m, q / (double)n);
  free(theta); free(u); free(rho);
  sabc_destroy(h);
  return 0;
}
