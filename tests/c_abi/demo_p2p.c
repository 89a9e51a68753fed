/* demo_p2p.c -- two shards in two PROCESSES from plain C: nothing but include/sabc_hip.h and a socket pair.
 * What a multi-GPU host in any language with a C FFI does with the peer-to-peer transport (include/sabc_hip.h
 * "sabc_comm_p2p_*"): every rank creates its handle and installs whatever collectives the host has (here: an
 * allreduce / allgather of host buffers over a UNIX socket pair -- sabc_set_collectives), then ONE call,
 * sabc_comm_p2p_setup, exchanges the descriptors over them, maps the peer's slot area / populations / rho (hipIpc), runs
 * the self-test and makes the ranks AGREE: either both run peer to peer (sabc_initialize / sabc_update then take ONE launch
 * between two update kernels and never touch the collectives) or both stay on the collectives -- no rank waits out a bound
 * because the other decided differently.  Tear-down needs no barrier either: sabc_destroy leaves the group in order.
 *
 * The process forks BEFORE anything touches HIP (no handle exists yet); the child is rank 1.  Both ranks use device 0 (the
 * test box has one GPU; on a node each rank would pass its own ordinal).
 *
 * usage: demo_p2p n_particles n_updates seed obs_mean proposal(0 RandomWalk | 1 DifferentialEvolution) [stale]
 *        "stale": rank 1's self-test is told to read a stale line -- first contact fails on ONE rank, both stay on the socket
 * rank 0 prints one line:  n_accept n_resampling n_population_updates eps mean_of_all_particles sum_of_squares_about_it
 *                          collective_calls kernel_launches p2p_active setup_seconds
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include "sabc_hip.h"

static int xfer(int fd, void *buf, size_t n, int out) {
  char *p = (char *)buf;
  while (n) {
    const ssize_t k = out ? write(fd, p, n) : read(fd, p, n);
    if (k <= 0) return -1;
    p += k; n -= (size_t)k;
  }
  return 0;
}

/* the host's own collectives for two ranks: each sends its block and reads the peer's; the sum is taken in rank order, so
   both ranks get the same bits (host pointers: sabc_set_collectives(..., device_buffers = 0)) */
typedef struct { int fd, rank; } sock_ctx;
/* rank 0 writes first and reads second, rank 1 the other way round: no block size can fill both socket buffers at once */
static int swap(const sock_ctx *c, const double *out, double *in, int64_t count) {
  const size_t bytes = (size_t)count * sizeof(double);
  if (c->rank == 0) return xfer(c->fd, (void *)out, bytes, 1) || xfer(c->fd, in, bytes, 0);
  return xfer(c->fd, in, bytes, 0) || xfer(c->fd, (void *)out, bytes, 1);
}
static int sock_allgather(void *ctx, const void *send, void *recv, int64_t count, void *stream) {
  (void)stream;
  sock_ctx *c = (sock_ctx *)ctx;
  double *out = (double *)recv;
  memcpy(out + (size_t)c->rank * (size_t)count, send, (size_t)count * sizeof(double));
  return swap(c, (const double *)send, out + (size_t)(1 - c->rank) * (size_t)count, count) ? -1 : 0;
}
static int sock_allreduce(void *ctx, void *buf, int64_t count, void *stream) {
  (void)stream;
  sock_ctx *c = (sock_ctx *)ctx;
  double *mine = (double *)buf, *theirs = malloc((size_t)count * sizeof(double));
  int rc = !theirs || swap(c, mine, theirs, count);
  for (int64_t i = 0; !rc && i < count; ++i) mine[i] = c->rank == 0 ? mine[i] + theirs[i] : theirs[i] + mine[i];
  free(theirs);
  return rc ? -1 : 0;
}

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int fail(sabc_handle *h, const char *what, int rc, int rank) {
  fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, what, rc, h ? sabc_last_error(h) : sabc_last_global_error());
  return 1;
}

int main(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "usage: demo_p2p n_particles n_updates seed obs_mean proposal\n"); return 2; }
  const int64_t n = atoll(argv[1]), updates = atoll(argv[2]);
  const int prop = atoi(argv[5]);
  const int stale = argc > 6 && strcmp(argv[6], "stale") == 0;
  int sv[2];
  if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 1; }
  const pid_t child = fork();                       /* before any HIP call */
  if (child < 0) { perror("fork"); return 1; }
  const int rank = child == 0 ? 1 : 0, fd = sv[rank];
  close(sv[1 - rank]);

  sabc_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.abi_version = SABC_ABI_VERSION;
  cfg.device = 0;
  cfg.n_particles = n;                              /* GLOBAL particle count: each rank holds n / 2 */
  cfg.n_para = 1; cfg.n_stats = 1;
  cfg.model_id = SABC_MODEL_GAUSS_IID;
  cfg.n_model_params = 4;
  cfg.model_params[0] = 100; cfg.model_params[1] = 1.0; cfg.model_params[2] = atof(argv[4]); cfg.model_params[3] = 0.0;
  cfg.prior_kind[0] = SABC_PRIOR_NORMAL; cfg.prior_a[0] = 0.0; cfg.prior_b[0] = 2.0;
  cfg.algorithm = SABC_ALG_SINGLE_EPS;
  cfg.rank = rank; cfg.world = 2;
  cfg.v = 1.0; cfg.delta = 0.1;
  cfg.seed = strtoull(argv[3], NULL, 10);

  sabc_handle *h = NULL;
  int rc = sabc_create(&cfg, &h);
  if (rc) return fail(NULL, "sabc_create", rc, rank);

  /* the host's collectives, then the whole peer-to-peer set-up in one collective call */
  sock_ctx sc = {fd, rank};
  if ((rc = sabc_set_collectives(h, sock_allreduce, sock_allgather, &sc, 0))) return fail(h, "sabc_set_collectives", rc, rank);
  if (stale && rank == 1) sabc_comm_p2p_inject_stale(h, 1);
  const double t0 = now_s();
  const int on = sabc_comm_p2p_setup(h);
  const double setup_seconds = now_s() - t0;
  if (on < 0) return fail(h, "sabc_comm_p2p_setup", on, rank);
  if (on != sabc_comm_p2p_active(h) || on != (stale ? 0 : 1)) {
    fprintf(stderr, "rank %d: set-up said %d, transport active %d: %s\n", rank, on, sabc_comm_p2p_active(h), sabc_last_error(h));
    return 1;
  }

  if ((rc = sabc_initialize(h, (updates + 1) * n))) return fail(h, "sabc_initialize", rc, rank);
  sabc_update_args up;
  memset(&up, 0, sizeof(up));
  up.n_simulation = updates * n;
  up.v = 1.0; up.delta = 0.1;
  up.resample = (double)n / 4.0;
  up.checkpoint_history = 1;
  up.proposal_kind = prop == 1 ? SABC_PROP_DIFFEVO : SABC_PROP_RANDOMWALK;
  up.proposal_p0 = prop == 1 ? 2.38 / 1.4142135623730951 : 0.8;      /* gamma0 = 2.38 / sqrt(2 d) | beta */
  up.proposal_p1 = prop == 1 ? 1e-5 : 0.0;
  if ((rc = sabc_update(h, &up))) return fail(h, "sabc_update", rc, rank);

  const int64_t nl = sabc_n_local(h);
  double *theta = malloc(sizeof(double) * (size_t)nl);
  if ((rc = sabc_get_population(h, theta, NULL, NULL))) return fail(h, "sabc_get_population", rc, rank);
  double part[2] = {0.0, 0.0};                      /* sum, sum of squares of this shard */
  for (int64_t i = 0; i < nl; ++i) { part[0] += theta[i]; part[1] += theta[i] * theta[i]; }
  free(theta);
  int status = 0;
  if (rank == 1) {
    if (xfer(fd, part, sizeof(part), 1)) return 1;
  } else {
    double other[2];
    if (xfer(fd, other, sizeof(other), 0)) return 1;
    int64_t c[4];
    double eps[SABC_MAX_STATS];
    int32_t eps_len = 0;
    sabc_get_counters(h, c);
    sabc_get_epsilon(h, eps, &eps_len);
    const double m = (part[0] + other[0]) / (double)n;
    printf("%lld %lld %lld %.17g %.17g %.17g %lld %lld %d %.3f\n", (long long)c[1], (long long)c[2], (long long)c[3], eps[0], m,
           part[1] + other[1] - (double)n * m * m, (long long)sabc_collective_calls(h), (long long)sabc_kernel_launches(h),
           sabc_comm_p2p_active(h), setup_seconds);
  }
  /* no barrier before tearing down: sabc_destroy leaves the group in order (the peer is told, nothing is freed under it) */
  sabc_destroy(h);
  if (sabc_comm_p2p_parked_bytes() != 0) { fprintf(stderr, "rank %d: %lld bytes parked\n", rank, (long long)sabc_comm_p2p_parked_bytes()); return 1; }
  if (rank == 0) { waitpid(child, &status, 0); if (!WIFEXITED(status) || WEXITSTATUS(status)) return 1; }
  return 0;
}
