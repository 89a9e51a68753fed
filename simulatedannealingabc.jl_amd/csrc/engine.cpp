// engine.cpp -- see engine.hpp.  Line references are to the reference checkout
// (src/SimulatedAnnealingABC.jl unless a file is named).
#include "engine.hpp"
#include <chrono>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "control.hpp"
#include "prior_math.hpp"

namespace sabc {

Engine::Engine(const sabc_config &cfg, Backend *backend, Collectives *coll) : cfg_(cfg), be_(backend), coll_(coll) {
  std::memset(&m_, 0, sizeof(m_));
  m_.model_id = cfg.model_id;
  m_.d = cfg.n_para;
  m_.s = cfg.n_stats;
  m_.n_model_params = cfg.n_model_params;
  for (int i = 0; i < SABC_MAX_MODEL_PARAMS; ++i) m_.p[i] = cfg.model_params[i];
  for (int k = 0; k < kMaxPara; ++k) {
    m_.prior_kind[k] = cfg.prior_kind[k];
    m_.prior_a[k] = cfg.prior_a[k];
    m_.prior_b[k] = cfg.prior_b[k];
    m_.prior_c[k] = cfg.prior_c[k];
    m_.prior_d[k] = cfg.prior_d[k];
    const double a = cfg.prior_a[k], b = cfg.prior_b[k];
    // the normalising constant of each family's logpdf, once, on the host
    if (cfg.prior_kind[k] == SABC_PRIOR_GAMMA) {
      m_.prior_logc[k] = (a > 0 && b > 0) ? std::lgamma(a) + a * std::log(b) : 0.0;
    } else if (cfg.prior_kind[k] == SABC_PRIOR_BETA) {
      m_.prior_logc[k] = (a > 0 && b > 0) ? std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b) : 0.0;
    } else if (cfg.prior_kind[k] == SABC_PRIOR_TRUNCNORMAL) {
      const double lo = b > 0 ? (cfg.prior_c[k] - a) / b : 0.0, hi = b > 0 ? (cfg.prior_d[k] - a) / b : 0.0;
      // lo' > 0: the draw runs in the mirrored lower tail (device_models.hpp); the sign of k0 says so
      m_.prior_k0[k] = lo > 0 ? -hostmath::norm_cdf(-lo) : hostmath::norm_cdf(lo);
      // the mass between the bounds; for bounds in the upper tail the complement form keeps its digits
      m_.prior_k1[k] = lo > 0 ? hostmath::norm_cdf(-lo) - hostmath::norm_cdf(-hi) : hostmath::norm_cdf(hi) - hostmath::norm_cdf(lo);
      m_.prior_logc[k] = (b > 0 && m_.prior_k1[k] > 0) ? std::log(b) + std::log(m_.prior_k1[k]) : 0.0;
    } else {
      const double scale = cfg.prior_kind[k] == SABC_PRIOR_UNIFORM ? b - a : cfg.prior_kind[k] == SABC_PRIOR_EXPONENTIAL ? a : b;
      m_.prior_logc[k] = scale > 0 ? std::log(scale) : 0.0;
    }
  }
  m_.prior_joint = cfg.prior_joint;
  m_.prior_joint_logc = 0.5 * (double)cfg.n_para * 1.8378770664093454835606594728112;   // d/2 log(2 pi)
  for (int i = 0; i < kMaxJointPara * kMaxJointPara; ++i) m_.prior_L[i] = 0.0;
  if (cfg.prior_joint == 1 && cfg.n_para >= 1 && cfg.n_para <= kMaxJointPara)
    for (int k = 0; k < cfg.n_para; ++k) {
      for (int l = 0; l <= k; ++l) m_.prior_L[k * cfg.n_para + l] = cfg.prior_chol[k * cfg.n_para + l];
      const double lkk = cfg.prior_chol[k * cfg.n_para + k];
      m_.prior_joint_logc += lkk > 0 ? std::log(lkk) : 0.0;
    }
  m_.seed = cfg.seed;
  const int world = cfg.world < 1 ? 1 : cfg.world;
  sh_ = make_shard(cfg.n_particles, cfg.rank, world);
  std::memset(&cb_, 0, sizeof(cb_));
  for (int i = 0; i < kMaxPara * kMaxPara; ++i) cb_.sigma[i] = -1.0;   // proposals.jl:32,34 sentinel
}

int Engine::validate() {
  if (cfg_.abi_version != SABC_ABI_VERSION) return fail(SABC_ERR_BAD_CONFIG, "sabc_config.abi_version mismatch");
  if (!(cfg_.algorithm == SABC_ALG_SINGLE_EPS || cfg_.algorithm == SABC_ALG_MULTI_EPS))
    return fail(SABC_ERR_BAD_ALGORITHM, "Argument `algorithm` must be :multi_eps or :single_eps!");   // :462-464
  if (cfg_.n_particles < 1) return fail(SABC_ERR_BAD_CONFIG, "n_particles must be positive");
  const int d = cfg_.n_para, s = cfg_.n_stats;
  if (d < 1 || d > kMaxPara || s < 1 || s > kMaxStats) return fail(SABC_ERR_BAD_CONFIG, "n_para / n_stats out of range");
  if (cfg_.model_id == SABC_MODEL_USER && s > SABC_MAX_SOURCE_STATS)
    return fail(SABC_ERR_BAD_CONFIG, "a simulator compiled from source takes at most 16 statistics (its ECDF index lives in the kernel's LDS)");
  if (cfg_.world < 1 || cfg_.rank < 0 || cfg_.rank >= cfg_.world) return fail(SABC_ERR_BAD_CONFIG, "bad rank / world");
  if (cfg_.world > 1)
    for (int r = 0; r < cfg_.world; ++r)
      if (shard_n_local(sh_, r) < 2) return fail(SABC_ERR_BAD_CONFIG, "every shard needs at least two particles");
  const double *p = cfg_.model_params;
  bool ok = false;
  host_mode_ = cfg_.model_id == SABC_MODEL_HOST || cfg_.prior_joint == 2;    // the per-particle body is cut at the host
  switch (cfg_.model_id) {
    case SABC_MODEL_HOST:
    case SABC_MODEL_USER:
      ok = true;                                   // any d, s within the maxima; f_dist is the caller's
      break;
    case SABC_MODEL_GAUSS_IID:
      ok = (d == 1 || d == 2) && (s == 1 || s == 2) && cfg_.n_model_params >= 4 && p[0] >= 1;
      break;
    case SABC_MODEL_GAUSS2D:
      ok = d == 2 && s == 3 && cfg_.n_model_params >= 6 && p[0] >= 2 && std::fabs(p[1]) < 1.0;
      break;
    case SABC_MODEL_GK:
      ok = d == 4 && s == 4 && cfg_.n_model_params >= 10 && p[0] >= 1 && p[0] <= 128;
      for (int j = 0; ok && j < 4; ++j) ok = p[2 + j] >= 1 && p[2 + j] <= p[0];
      break;
    case SABC_MODEL_LV:
      ok = d == 3 && s == 4 && cfg_.n_model_params >= 9 && p[0] >= 2 && p[1] > 0;
      break;
    default:
      ok = false;
  }
  if (!ok) return fail(SABC_ERR_BAD_CONFIG, "unknown model id or model parameters inconsistent with n_para / n_stats");
  if (cfg_.prior_joint < 0 || cfg_.prior_joint > 3) return fail(SABC_ERR_BAD_CONFIG, "unknown joint prior");
  if (cfg_.prior_joint == 3 && cfg_.model_id != SABC_MODEL_USER)
    return fail(SABC_ERR_BAD_CONFIG, "a prior from device source travels in the simulator's source (SABC_MODEL_USER)");
  if (cfg_.prior_joint == 1) {
    if (d > kMaxJointPara) return fail(SABC_ERR_BAD_CONFIG, "an MvNormal prior as data takes at most 8 parameters");
    for (int k = 0; k < d; ++k)
      if (!(cfg_.prior_chol[k * d + k] > 0)) return fail(SABC_ERR_BAD_CONFIG, "MvNormal prior needs a Cholesky factor with a positive diagonal");
  }
  for (int k = 0; k < d && cfg_.prior_joint == 0; ++k) {          // (joint priors carry no per-dimension descriptors)
    if (cfg_.prior_kind[k] == SABC_PRIOR_NORMAL) {
      if (!(cfg_.prior_b[k] > 0)) return fail(SABC_ERR_BAD_CONFIG, "Normal prior needs sigma > 0");
    } else if (cfg_.prior_kind[k] == SABC_PRIOR_UNIFORM) {
      if (!(cfg_.prior_b[k] > cfg_.prior_a[k])) return fail(SABC_ERR_BAD_CONFIG, "Uniform prior needs upper > lower");
    } else if (cfg_.prior_kind[k] == SABC_PRIOR_EXPONENTIAL) {
      if (!(cfg_.prior_a[k] > 0)) return fail(SABC_ERR_BAD_CONFIG, "Exponential prior needs scale > 0");
    } else if (cfg_.prior_kind[k] == SABC_PRIOR_LOGNORMAL) {
      if (!(cfg_.prior_b[k] > 0)) return fail(SABC_ERR_BAD_CONFIG, "LogNormal prior needs sigma > 0");
    } else if (cfg_.prior_kind[k] == SABC_PRIOR_GAMMA) {
      if (!(cfg_.prior_a[k] > 0 && cfg_.prior_b[k] > 0)) return fail(SABC_ERR_BAD_CONFIG, "Gamma prior needs shape > 0 and scale > 0");
    } else if (cfg_.prior_kind[k] == SABC_PRIOR_BETA) {
      if (!(cfg_.prior_a[k] > 0 && cfg_.prior_b[k] > 0)) return fail(SABC_ERR_BAD_CONFIG, "Beta prior needs alpha > 0 and beta > 0");
    } else if (cfg_.prior_kind[k] == SABC_PRIOR_TRUNCNORMAL) {
      if (!(cfg_.prior_b[k] > 0 && cfg_.prior_d[k] > cfg_.prior_c[k]))
        return fail(SABC_ERR_BAD_CONFIG, "truncated Normal prior needs sigma > 0 and upper > lower");
      if (!(m_.prior_k1[k] > 1e-300))
        return fail(SABC_ERR_BAD_CONFIG, "truncated Normal prior: no probability mass between the bounds");
    } else {
      return fail(SABC_ERR_BAD_CONFIG, "unknown prior kind");
    }
  }
  np_ = n_partials(d, s);
  eps_len_ = cfg_.algorithm == SABC_ALG_MULTI_EPS ? s : 1;
  cb_.eps_len = eps_len_;
  return 0;
}

void Engine::counters(int64_t out[4]) const {
  out[0] = n_simulation_; out[1] = cb_.n_accept; out[2] = n_resampling_; out[3] = n_population_updates_;
}
void Engine::set_counters(const int64_t in[4]) {
  n_simulation_ = in[0]; cb_.n_accept = in[1]; n_resampling_ = in[2]; n_population_updates_ = in[3];
}
int Engine::set_eps(const double *e, int len) {
  if (len != eps_len_) return fail(SABC_ERR_BAD_CONFIG, "epsilon length does not match the algorithm");
  for (int i = 0; i < len; ++i) cb_.eps[i] = e[i];
  return 0;
}
void Engine::history(double *e, double *u, double *r) const {
  if (e) std::memcpy(e, eps_hist_.data(), eps_hist_.size() * sizeof(double));
  if (u) std::memcpy(u, u_hist_.data(), u_hist_.size() * sizeof(double));
  if (r) std::memcpy(r, rho_hist_.data(), rho_hist_.size() * sizeof(double));
}

// block partials -> shard sums (ControlBlock::sums) -> allreduce over shards; nothing is read back
int Engine::global_reduce(int64_t rows, bool guarded) {
  if (be_->reduce_partials(rows, guarded)) return fail(SABC_ERR_HIP, "reduce_partials failed");
  if (sh_.world > 1 && p2p()) {
    // the sum over the shards happens inside the launch of the control step that follows (p2p.hpp)
    if (be_->p2p_exchange_pending()) return fail(SABC_ERR_COMM, "peer-to-peer exchange could not be queued");
    comm_bytes_ += np_ * (int64_t)sizeof(double);
    return 0;
  }
  if (sh_.world > 1) {
    double *sums = be_->sums_buffer();
    be_->prof_begin(SABC_KERNEL_COLLECTIVE);
    const int rc = allreduce(sums, np_);
    be_->prof_end(SABC_KERNEL_COLLECTIVE);
    if (rc) return fail(SABC_ERR_COMM, "allreduce of the population sums failed");
  }
  return 0;
}

// the collectives, with the bytes that land in this shard's receive buffers counted
int Engine::allreduce(double *buf, int64_t count) {
  comm_bytes_ += count * (int64_t)sizeof(double);
  collective_calls_ += 1;
  return coll_->allreduce_sum(buf, count);
}
int Engine::allgather(const double *send, double *recv, int64_t count_per_rank) {
  comm_bytes_ += (int64_t)sh_.world * count_per_rank * (int64_t)sizeof(double);
  collective_calls_ += 1;
  return coll_->allgather(send, recv, count_per_rank);
}
int Engine::alltoallv(const double *send, const int64_t *sc, double *recv, const int64_t *rc) {
  for (int p = 0; p < sh_.world; ++p) comm_bytes_ += rc[p] * (int64_t)sizeof(double);
  collective_calls_ += 1;
  return coll_->alltoallv(send, sc, recv, rc);
}

// Peer-to-peer transport, end of a call: every shard tells the others how the call went and, when it went well, waits for
// the same from them -- a shard whose peer gave up in the call's LAST exchange must not return success on its own.  The
// status exchange is queued BEFORE the call's final read of the control block, which then carries its outcome.
int Engine::p2p_commit_ok() {
  if (!p2p()) return 0;
  if (be_->p2p_commit(0, true)) return fail(SABC_ERR_COMM, "peer-to-peer status exchange could not be queued");
  return 0;
}
// ... and after a failure: tell the peers (no wait) and go back to the installed Collectives -- the wrapper restores the
// particles and repeats the call over them.
void Engine::p2p_abort(int rc) {
  if (!p2p()) return;
  (void)be_->p2p_commit(1, false);
  ControlBlock scratch;
  (void)be_->read_control(&scratch);      // the post has left before the call returns (a later set-up wipes the slots)
  // an error of the algorithm (zero mean u, a covariance that is not positive definite, a failing callback) is the same
  // on every shard and says nothing about the transport: only a wait that gave up switches it off
  if (rc == SABC_ERR_COMM || rc == SABC_ERR_HIP) be_->p2p_disable();
}

// Entry of sabc_initialize / sabc_update on the peer-to-peer transport, before anything is launched: has a peer left the
// group (its host page says so: p2p.hpp)?  Then this shard leaves as well -- nothing of the peer's is read again -- and the
// call runs over the collectives underneath; without any it fails here, the handle untouched.
int Engine::p2p_check_peers() {
  if (!p2p() || be_->p2p_peers_present()) return 0;
  be_->p2p_disable();
  if (coll_->usable()) return 0;
  return fail(SABC_ERR_COMM, "a shard has left the peer-to-peer group and no collectives are installed underneath");
}

int Engine::stats_reduce() {
  int64_t rows = 0;
  if (be_->stats(&rows)) return fail(SABC_ERR_HIP, "stats kernel failed");
  return global_reduce(rows);
}

ControlArgs Engine::control_args(int32_t mode, const sabc_update_args *a, double v, double threshold) const {
  ControlArgs c;
  std::memset(&c, 0, sizeof(c));
  c.mode = mode;
  c.d = m_.d; c.s = m_.s;
  c.algorithm = cfg_.algorithm;
  c.prop_kind = a ? a->proposal_kind : -1;
  c.prop_p0 = a ? a->proposal_p0 : 0.0;
  c.n_global = (double)sh_.n_global;
  c.v = v;
  c.hist_capacity = hist_capacity_;
  c.resample_threshold = threshold;
  // update steps of the device-coded simulators report the change of sum(rho); stats passes (and the host-callback
  // mode, whose sums come from a stats pass) the sum itself
  c.rho_is_delta = ((mode & CTRL_ACCUMULATE) && !host_mode_) ? 1 : 0;
  return c;
}

int Engine::control(int32_t mode, const sabc_update_args *a, double v, bool notify, double threshold, int64_t *seq_out) {
  ControlArgs c = control_args(mode, a, v, threshold);
  c.notify_seq = notify ? ++notify_seq_ : 0;
  if (seq_out) *seq_out = c.notify_seq;
  if (be_->control(c)) return fail(SABC_ERR_HIP, "control kernel failed");
  return 0;
}

int Engine::wait_step(int64_t seq, int64_t *n_accept, int *halted) {
  int err = 0;
  if (be_->wait_notify(seq, n_accept, &err, halted)) {
    // the stream drained without the step reporting: a peer-to-peer wait that gave up leaves its error in the control block
    if (p2p() && be_->read_control(&cb_) == 0 && cb_.error == SABC_ERR_COMM) return sync_control();
    return fail(SABC_ERR_HIP, "waiting for the control step failed");
  }
  host_syncs_ += 1;
  if (std::getenv("SABC_DEBUG_SYNCS")) std::fprintf(stderr, "[sabc rank %d] wait_step seq %lld halted %d err %d -> %lld\n", sh_.rank, (long long)seq, *halted, err, (long long)host_syncs_);
  if (err) { cb_.error = err; return sync_control(); }
  return 0;
}

int Engine::sync_control() {
  if (be_->read_control(&cb_)) return fail(SABC_ERR_HIP, "reading the control block failed");
  host_syncs_ += 1;
  if (std::getenv("SABC_DEBUG_SYNCS")) std::fprintf(stderr, "[sabc rank %d] sync_control error %d -> %lld\n", sh_.rank, cb_.error, (long long)host_syncs_);
  switch (cb_.error) {
    case 0: return 0;
    case SABC_ERR_ZERO_MEAN_U: return fail(SABC_ERR_ZERO_MEAN_U, "Division by zero - Mean u for a statistic is <= eps()");   // :107-109
    case SABC_ERR_NOT_POSDEF: return fail(SABC_ERR_NOT_POSDEF, "RandomWalk covariance is not positive definite");
    case SABC_ERR_COMM: {
      static const char *kind[4] = {"?", "sums exchange", "barrier", "end-of-call status"};
      const int k = (cb_.comm_where >> 24) & 7;
      char buf[200];
      std::snprintf(buf, sizeof(buf), k & 4 ? "peer-to-peer %s %d: shard %d has left the group"
                                            : "peer-to-peer %s %d: shard %d did not post within the bound (or reported a failed call)",
                    kind[k & 3], cb_.comm_where & 0xFFFFF, (cb_.comm_where >> 20) & 15);
      return fail(SABC_ERR_COMM, buf);
    }
    case SABC_ERR_HIP: return fail(SABC_ERR_HIP, "a wait inside the persistent update kernel (row exchange / grid barrier) ran into its bound: a workgroup "
                                                 "that was resident at the launch's rendezvous stopped taking part (SABC_PERSISTENT=0 takes the launch chain)");
    default: return fail(cb_.error, "error raised by the device-side control step");
  }
}

// the rows the control kernel appended on the device -> the host-side histories (:33-35)
int Engine::drain_history() {
  const int s = m_.s, row_len = eps_len_ + 2 * s;
  const int64_t rows = cb_.hist_rows;
  if (rows > 0) {
    std::vector<double> buf((size_t)(rows * row_len));
    if (be_->read_history(buf.data(), rows, row_len)) return fail(SABC_ERR_HIP, "reading the history rows failed");
    for (int64_t r = 0; r < rows; ++r) {
      const double *row = &buf[(size_t)(r * row_len)];
      eps_hist_.insert(eps_hist_.end(), row, row + eps_len_);
      u_hist_.insert(u_hist_.end(), row + eps_len_, row + eps_len_ + s);
      rho_hist_.insert(rho_hist_.end(), row + eps_len_ + s, row + row_len);
    }
  }
  cb_.hist_rows = 0;
  return 0;
}

// resample_population, :124-137.  ControlBlock::sums must hold the CURRENT sum of u (the weights
// kernel takes ubar from there); leaves the sums of the resampled population in its place.
int Engine::resample(double delta, uint64_t iter) {
  const int d = m_.d, s = m_.s;
  if (sh_.world == 1) {
    int64_t rows = -1;
    if (be_->resample_local(delta, iter, &rows)) return fail(SABC_ERR_HIP, "resample kernels failed");   // :126-132
    return rows >= 0 ? global_reduce(rows) : stats_reduce();
  }
  if (p2p()) {
    // weights -> barrier -> every shard scans the owners' weight rows and reads the rows it drew from their owners: nothing
    // is gathered, no host round trip (host_syncs unchanged)
    if (be_->resample_p2p(delta, iter)) return fail(SABC_ERR_HIP, "peer-to-peer resample kernels failed");
    // what crosses: the other shards' weights (read once: the scan's first pass parks them locally) and the drawn rows
    // that live elsewhere
    comm_bytes_ += ((sh_.n_global - sh_.n_local) + sh_.n_local * (int64_t)(d + s) * (sh_.world - 1) / sh_.world) * (int64_t)sizeof(double);
    return stats_reduce();
  }
  if (be_->resample_weights(delta)) return fail(SABC_ERR_HIP, "resample weights kernel failed");   // :126-127
  const int64_t rows = d + s + 1;
  if (sh_.world > 1 && coll_->has_alltoallv()) {
    const int rc = resample_exchange(iter);
    if (rc) return rc;
    return stats_reduce();
  }
  const double *gathered = be_->pop_block();
  if (sh_.world > 1) {          // transport without a personalised exchange: every shard takes the whole population
    double *g = be_->gather_buffer((int64_t)sh_.world * rows * sh_.cap);
    if (!g) return fail(SABC_ERR_HIP, "out of memory for the resample gather buffer");
    if (allgather(be_->pop_block(), g, rows * sh_.cap)) return fail(SABC_ERR_COMM, "allgather of the population failed");
    gathered = g;
  }
  if (be_->resample_draw(gathered, iter)) return fail(SABC_ERR_HIP, "resample draw kernel failed");   // :129-132
  return stats_reduce();
}

// The sharded resample without moving the population: only the WEIGHT row is gathered (n doubles instead of
// (d + s + 1) n); every shard runs the same scan over it (bitwise identical running sums), draws the sources of its own
// n_local offspring from the global categorical (:129), asks each owner for the rows it drew (offsets inside the owner)
// and gets exactly those rows back (:131-132): n_local (1 + d + s) doubles cross instead of n (d + s + 1).
int Engine::resample_exchange(uint64_t iter) {
  const int d = m_.d, s = m_.s, W = sh_.world;
  const int64_t row_len = d + s;
  double *gw = be_->gather_buffer((int64_t)W * sh_.cap);
  if (!gw) return fail(SABC_ERR_HIP, "out of memory for the weight gather buffer");
  if (allgather(be_->pop_block() + (int64_t)row_len * sh_.cap, gw, sh_.cap)) return fail(SABC_ERR_COMM, "allgather of the weights failed");
  if (be_->resample_select(gw, iter)) return fail(SABC_ERR_HIP, "resample select kernel failed");
  std::vector<int64_t> sendc((size_t)W, 0), recvc((size_t)W, 0), sendr((size_t)W), recvr((size_t)W);
  double *req = be_->scratch_buffer(0, sh_.n_local > 0 ? sh_.n_local : 1);
  if (!req) return fail(SABC_ERR_HIP, "out of memory for the resample request buffer");
  if (be_->resample_bucket(sendc.data(), req)) return fail(SABC_ERR_HIP, "resample bucket kernel failed");
  host_syncs_ += 1;
  // who asks whom for how many rows: allgather of the W counts of every shard (exact as doubles)
  std::vector<double> cnt((size_t)W * (size_t)(W + 1), 0.0);
  for (int p = 0; p < W; ++p) cnt[(size_t)p] = (double)sendc[(size_t)p];
  double *cdev = be_->scratch_buffer(1, (int64_t)W * (W + 1));
  if (!cdev) return fail(SABC_ERR_HIP, "out of memory for the resample count buffer");
  if (be_->to_backend(cdev, cnt.data(), W)) return fail(SABC_ERR_HIP, "uploading the request counts failed");
  if (allgather(cdev, cdev + W, W)) return fail(SABC_ERR_COMM, "allgather of the request counts failed");
  if (be_->to_host(cnt.data() + W, cdev + W, (int64_t)W * W)) return fail(SABC_ERR_HIP, "reading the request counts failed");
  host_syncs_ += 1;
  int64_t R = 0;
  for (int p = 0; p < W; ++p) {
    recvc[(size_t)p] = (int64_t)cnt[(size_t)W + (size_t)p * W + (size_t)sh_.rank];   // what shard p asks of this one
    if (recvc[(size_t)p] < 0 || recvc[(size_t)p] > sh_.n_global) return fail(SABC_ERR_COMM, "corrupt request count in the resample exchange");
    R += recvc[(size_t)p];
    sendr[(size_t)p] = sendc[(size_t)p] * row_len;
    recvr[(size_t)p] = recvc[(size_t)p] * row_len;
  }
  double *req_in = be_->scratch_buffer(1, R > 0 ? R : 1);
  double *rows_out = be_->scratch_buffer(2, R > 0 ? R * row_len : 1);
  double *rows_in = be_->scratch_buffer(3, sh_.n_local > 0 ? sh_.n_local * row_len : 1);
  if (!req_in || !rows_out || !rows_in) return fail(SABC_ERR_HIP, "out of memory for the resample exchange buffers");
  if (alltoallv(req, sendc.data(), req_in, recvc.data())) return fail(SABC_ERR_COMM, "exchange of the resample requests failed");
  if (be_->resample_serve(req_in, R, rows_out)) return fail(SABC_ERR_HIP, "resample serve kernel failed");
  if (alltoallv(rows_out, recvr.data(), rows_in, sendr.data())) return fail(SABC_ERR_COMM, "exchange of the resampled rows failed");
  if (be_->resample_scatter(rows_in)) return fail(SABC_ERR_HIP, "resample scatter kernel failed");
  return 0;
}

// Partners of DifferentialEvolution / StretchMove come from the inactive halves of ALL shards
// (proposals.jl:105-106,141).  One shard reads them in place; several gather ONLY the inactive halves
// (d * ceil(cap / 2) doubles per shard, not the whole theta block) into [world][d][hcap].
int Engine::partner_source(int inactive_half, PartnerView *out) {
  if (sh_.world == 1) {
    *out = partner_view(be_->pop_block(), 0, inactive_half);
    return 0;
  }
  const int d = m_.d;
  if (p2p()) {
    // partners are read where they live: two (DE) or one (Stretch) random rows of d doubles per proposal, from the inactive
    // half of whichever shard owns them -- no gather; the caller puts a barrier between the two half batches
    PartnerView pv = partner_view(nullptr, 0, inactive_half);
    if (be_->partner_view_p2p(&pv)) return fail(SABC_ERR_COMM, "peer-mapped populations are not available");
    *out = pv;
    return 0;
  }
  const int64_t hcap = sh_.cap - sh_.cap / 2;                 // the larger (second) half of a full shard
  const int64_t h = sh_.n_local / 2;
  const int64_t off = inactive_half == 1 ? h : 0, cnt = inactive_half == 1 ? sh_.n_local - h : h;
  double *g = be_->gather_buffer((int64_t)(sh_.world + 1) * d * hcap);
  if (!g) return fail(SABC_ERR_HIP, "out of memory for the partner gather buffer");
  double *send = g + (int64_t)sh_.world * d * hcap;
  if (be_->copy_rows(be_->pop_block() + off, sh_.cap, send, hcap, d, cnt)) return fail(SABC_ERR_HIP, "packing the inactive half failed");
  if (allgather(send, g, (int64_t)d * hcap)) return fail(SABC_ERR_COMM, "allgather of the inactive halves failed");
  PartnerView pv = partner_view(g, (int64_t)d * hcap, inactive_half);
  pv.cap = hcap;                                              // row stride inside a gathered block
  pv.off_full = pv.off_last = 0;                              // the blocks hold the inactive half only
  *out = pv;
  return 0;
}

PartnerView Engine::partner_view(const double *base, int64_t rank_stride, int inactive_half) const {
  PartnerView pv;
  std::memset(&pv, 0, sizeof(pv));
  const int64_t n_full = sh_.cap, n_last = shard_n_local(sh_, sh_.world - 1);
  const int64_t h_full = n_full / 2, h_last = n_last / 2;
  pv.base = base;
  pv.rank_stride = rank_stride;
  pv.cap = sh_.cap;
  pv.world = sh_.world;
  if (inactive_half == 1) {        // second halves are frozen
    pv.off_full = h_full; pv.m_full = n_full - h_full;
    pv.off_last = h_last; pv.m_last = n_last - h_last;
  } else {
    pv.off_full = 0; pv.m_full = h_full;
    pv.off_last = 0; pv.m_last = h_last;
  }
  if (sh_.world == 1) { pv.m_full = pv.m_last; pv.off_full = pv.off_last; }
  pv.m_total = (int64_t)(sh_.world - 1) * pv.m_full + pv.m_last;
  if (pv.m_full < 1) pv.m_full = 1;
  return pv;
}

// ------------------------------------------------------------------------------------------
// initialization(), :151-227
// ------------------------------------------------------------------------------------------
int Engine::initialize(int64_t n_simulation) {
  if (n_simulation < sh_.n_global) {                                        // :155-156
    char buf[160];
    std::snprintf(buf, sizeof(buf), "`n_simulation = %lld` is too small for %lld particles.", (long long)n_simulation,
                  (long long)sh_.n_global);
    return fail(SABC_ERR_NSIM_TOO_SMALL, buf);
  }
  if (int rc0 = p2p_check_peers()) return rc0;
  const bool can_retry = p2p() && coll_->usable();
  int rc = initialize_body();
  if (rc) { const std::string why = err_; p2p_abort(rc); err_ = why; }
  if (rc == SABC_ERR_COMM && can_retry && !p2p()) {
    // a peer-to-peer wait gave up (on every shard: the status exchange sees to that): initialization starts from nothing,
    // so it is simply run again over the collectives underneath
    const std::string why = err_;
    p2p_fallbacks_ += 1;
    rc = initialize_body();
    if (rc) err_ = "after falling back from the peer-to-peer transport (" + why + "): " + err_;
  }
  be_->end_of_call();
  return rc;
}

int Engine::initialize_body() {
  const int s = m_.s;
  for (int k = 0; k < kMaxPara; ++k) cb_.pivot[k] = 0.0;
  cb_.n_accept = 0; cb_.hist_rows = 0; cb_.error = 0; cb_.halt = 0; cb_.eps_len = eps_len_;
  hist_capacity_ = 4;
  if (be_->history_reserve(hist_capacity_)) return fail(SABC_ERR_HIP, "history buffer allocation failed");
  if (be_->write_control(cb_)) return fail(SABC_ERR_HIP, "writing the control block failed");
  if (host_mode_) {
    if (be_->host_prior_simulate()) return fail(SABC_ERR_CALLBACK, "prior sample / host simulator failed");   // :172-179
  } else if (be_->prior_simulate()) {
    return fail(SABC_ERR_HIP, "prior sample / simulate kernel failed");   // :172-179
  }
  const double *gathered_rho = be_->rho_block();
  int any_negative = 0;
  if (p2p()) {
    // every shard sorts the full columns, read from their owners' rho blocks (identical ECDF tables everywhere)
    comm_bytes_ += (sh_.n_global - sh_.n_local) * (int64_t)s * (int64_t)sizeof(double);
    if (be_->build_cdf_p2p(cdf_len_, &any_negative)) return fail(SABC_ERR_HIP, "ECDF build failed");          // :187
  } else {
    if (sh_.world > 1) {
      double *g = be_->gather_buffer((int64_t)sh_.world * s * sh_.cap);
      if (!g) return fail(SABC_ERR_HIP, "out of memory for the rho gather buffer");
      if (allgather(be_->rho_block(), g, (int64_t)s * sh_.cap)) return fail(SABC_ERR_COMM, "allgather of rho failed");
      gathered_rho = g;
    }
    if (be_->build_cdf(gathered_rho, cdf_len_, &any_negative)) return fail(SABC_ERR_HIP, "ECDF build failed");   // :187
  }
  if (any_negative) return fail(SABC_ERR_NEG_DISTANCE, "Negative distances are not allowed!");             // :185
  for (int j = 0; j < s; ++j)
    if (cdf_len_[j] < 3) return fail(SABC_ERR_EMPTY_CDF, "all prior distances of one statistic are zero");
  if (be_->cdf_population()) return fail(SABC_ERR_HIP, "ECDF transform kernel failed");                   // :190-192
  int rc = stats_reduce();
  if (rc) return rc;
  if ((rc = control(0, nullptr, cfg_.v))) return rc;                        // takes the sums over (ubar for the weights)
  if ((rc = resample(cfg_.delta, 0))) return rc;                            // :197
  if ((rc = control(CTRL_EPSILON | CTRL_HISTORY | CTRL_PIVOT, nullptr, cfg_.v))) return rc;   // :200-208
  if ((rc = p2p_commit_ok())) return rc;
  if ((rc = sync_control())) return rc;
  clear_history();
  if ((rc = drain_history())) return rc;                                    // :180,207-208
  n_simulation_ = sh_.n_global;                                             // :213
  cb_.n_accept = 0; n_resampling_ = 1; n_population_updates_ = 0;           // :223
  initialized_ = true;
  population_replaced_ = true;                                              // (a handle may be initialised again: like a fresh one)
  return 0;
}

// one population update, enqueued only: the per-particle kernels (:304-331), the fused sums and
// their allreduce.  Nothing is read back here.
int Engine::enqueue_update(const sabc_update_args &a, uint64_t iter, bool guarded) {
  StepArgs c;
  std::memset(&c, 0, sizeof(c));
  c.iter = iter;
  c.prop_kind = a.proposal_kind; c.prop_p0 = a.proposal_p0; c.prop_p1 = a.proposal_p1;
  int64_t rows = 0, r = 0;
  if (host_mode_) {
    // f_dist on the host: propose on the device, simulate in the callback, accept on the device; the
    // sums come from a stats pass afterwards.  Synchronous by nature (the callback sits in the middle).
    if (a.proposal_kind == SABC_PROP_RANDOMWALK) {
      PartnerView none;
      std::memset(&none, 0, sizeof(none));
      if (be_->host_update_range(c, none, 0, sh_.n_local)) return fail(SABC_ERR_CALLBACK, "host-simulator update failed");
    } else {
      const int64_t h = sh_.n_local / 2;
      for (int half = 0; half < 2; ++half) {
        const int64_t lo = half == 0 ? 0 : h, cnt = half == 0 ? h : sh_.n_local - h;
        PartnerView pv;
        const int rc = partner_source(1 - half, &pv);
        if (rc) return rc;
        if (be_->host_update_range(c, pv, lo, cnt)) return fail(SABC_ERR_CALLBACK, "host-simulator update failed");
      }
    }
    if (be_->host_stats(&rows)) return fail(SABC_ERR_HIP, "stats kernel failed");
    return global_reduce(rows, false);
  }
  if (a.proposal_kind == SABC_PROP_RANDOMWALK) {
    // RandomWalk ignores the inactive half (proposals.jl:40,52), so both half batches of :304
    // are independent given eps and Sigma: one launch over the whole shard is the same update.
    PartnerView none;
    std::memset(&none, 0, sizeof(none));
    if (be_->update_range(c, none, 0, sh_.n_local, 0, &r)) return fail(SABC_ERR_HIP, "update kernel failed");
    rows = r;
  } else {
    const int64_t h = sh_.n_local / 2;
    for (int half = 0; half < 2; ++half) {                                  // :300-304
      const int64_t lo = half == 0 ? 0 : h, cnt = half == 0 ? h : sh_.n_local - h;
      PartnerView pv;                                                       // partners: the inactive halves of ALL shards
      const int rc = partner_source(1 - half, &pv);
      if (rc) return rc;
      if (p2p()) {
        // half batch B reads what half batch A wrote on EVERY shard: one flag barrier between the two launches.  (A needs
        // none: the exchange that ended the previous update was one.)
        if (half == 1 && be_->p2p_barrier(guarded)) return fail(SABC_ERR_COMM, "peer-to-peer barrier could not be queued");
        // rows actually read from other shards: 2 (DE) or 1 (Stretch) partners of d doubles per proposal, a share
        // (world - 1) / world of them remote
        const int64_t partners = a.proposal_kind == SABC_PROP_DIFFEVO ? 2 : 1;
        comm_bytes_ += cnt * partners * m_.d * (int64_t)sizeof(double) * (sh_.world - 1) / sh_.world;
      }
      if (be_->update_range(c, pv, lo, cnt, rows, &r)) return fail(SABC_ERR_HIP, "update kernel failed");
      rows += r;
    }
  }
  return global_reduce(rows, guarded);
}

// The loop of :294-375 on a small shard: the backend runs the updates in one launch (body -> sums -> control step, per update,
// inside the kernel) until the resample test of :340 fires; the host then resamples (:341), finishes that update's control
// step and launches the rest.  Same kernels' arithmetic, same control step, same history cadence as the chain below.
// *next_ix: the first update of the call this loop has NOT done -- n_pop + 1, or the update from which the launch chain has to
// take over: a launch whose workgroups did not all become resident within the rendezvous' bound (a device shared with other
// handles' or processes' kernels) has touched nothing and says so (done = -1)
int Engine::update_loop_persistent(const sabc_update_args &a, int64_t n_pop, int64_t cph, int64_t phase, int64_t *next_ix) {
  const int32_t after_update = CTRL_PROPOSAL | CTRL_EPSILON | CTRL_PIVOT;   // :348-354
  StepArgs c;
  std::memset(&c, 0, sizeof(c));
  c.prop_kind = a.proposal_kind; c.prop_p0 = a.proposal_p0; c.prop_p1 = a.proposal_p1;
  PartnerView pv_a, pv_b;                                                   // partners of half batch A: the second halves; of B: the first
  std::memset(&pv_a, 0, sizeof(pv_a));
  std::memset(&pv_b, 0, sizeof(pv_b));
  int rc;
  int64_t ix = 1;
  while (ix <= n_pop) {
    if (a.proposal_kind != SABC_PROP_RANDOMWALK) {                          // (the population buffers flip on a resample)
      if ((rc = partner_source(1, &pv_a))) return rc;
      if ((rc = partner_source(0, &pv_b))) return rc;
    }
    c.iter = (uint64_t)(n_population_updates_ + ix);
    const double threshold = (double)(n_resampling_ + 1) * a.resample;      // :340
    const ControlArgs ctrl = control_args(CTRL_ACCUMULATE | CTRL_CHECK | after_update, &a, a.v, threshold);
    int64_t done = 0;
    int halted = 0, error = 0;
    if (be_->update_persistent(c, ctrl, pv_a, pv_b, ix, phase, cph, n_pop - ix + 1, &done, &halted, &error))
      return fail(SABC_ERR_HIP, "persistent update kernel failed");
    host_syncs_ += 1;
    if (done == -1) {
      // (a launch that left at its rendezvous has kept the workgroups that did arrive waiting for the bound: on a device that
      // stays full, calls of a few updates each -- a wrapper's progress chunks -- would pay it every time.  Ten bounds' pause.)
      persistent_fallbacks_ += 1;
      persistent_retry_at_ = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
      *next_ix = ix;
      return 0;
    }
    persistent_launches_ += 1;
    if (error) { cb_.error = error; return sync_control(); }
    if (done < 1 || done > n_pop - ix + 1) return fail(SABC_ERR_HIP, "persistent update kernel reported an impossible update count");
    const int64_t last = ix + done - 1;
    if (halted) {
      // n_accept >= (n_resampling + 1) * resample after update `last` (:340): resample, then the part of its control step
      // that was skipped
      if ((rc = resample(a.delta, (uint64_t)(n_population_updates_ + last)))) return rc;          // :341
      n_resampling_ += 1;                                                                         // :342
      const int32_t hist = ((phase + last) % cph == 0) ? (int32_t)CTRL_HISTORY : 0;
      if ((rc = control(CTRL_CLEAR_HALT | after_update | hist, &a, a.v))) return rc;
    }
    ix = last + 1;
  }
  *next_ix = n_pop + 1;
  return 0;
}

// ------------------------------------------------------------------------------------------
// update_population!(), :251-402
// ------------------------------------------------------------------------------------------
// Error contract (include/sabc_hip.h): the reference works on copies and leaves its state untouched when it throws
// (:264-267, :387-397).  Here the population lives on the device and is updated in place, so after a failure inside
// the loop the queue is drained, the counters and eps are put back to their values at entry, and the handle refuses
// further updates until the caller has restored the particles with sabc_set_population().
int Engine::update(const sabc_update_args &a) {
  if (!initialized_) return fail(SABC_ERR_STATE, "population is not initialized (or a failed update left it half-updated: "
                                                 "restore it with sabc_set_population)");
  // Peer-to-peer transport with Collectives installed underneath: keep a device-side copy of the particles, so that a
  // call in which a peer-to-peer wait gave up can be put back and finished over the Collectives (every shard's call fails
  // with SABC_ERR_COMM then -- the end-of-call status exchange sees to that -- and every shard repeats it).
  if (int rc0 = p2p_check_peers()) return rc0;
  const bool can_retry = p2p() && coll_->usable() && be_->snapshot() == 0;
  const bool replaced_at_entry = population_replaced_;
  int rc = update_once(a);
  if (rc == SABC_ERR_COMM && can_retry && !p2p()) {
    const std::string why = err_;
    if (be_->restore_snapshot() == 0) {
      initialized_ = true;
      population_replaced_ = replaced_at_entry;
      p2p_fallbacks_ += 1;
      rc = update_once(a);
      if (rc) err_ = "after falling back from the peer-to-peer transport (" + why + "): " + err_;
    }
  }
  if (rc != SABC_ERR_STATE) be_->end_of_call();
  return rc;
}

int Engine::update_once(const sabc_update_args &a) {
  const ControlBlock at_entry = cb_;
  const int64_t resampling_at_entry = n_resampling_;
  const size_t hist_at_entry[3] = {eps_hist_.size(), u_hist_.size(), rho_hist_.size()};
  const int rc = update_loop(a);
  if (rc && rc != SABC_ERR_BAD_V && rc != SABC_ERR_BAD_DELTA && rc != SABC_ERR_BAD_BETA && rc != SABC_ERR_BAD_CONFIG) {
    const std::string why = err_;
    ControlBlock scratch;
    (void)be_->read_control(&scratch);                  // blocks until everything queued ahead has run
    cb_ = at_entry;
    cb_.halt = 0; cb_.error = 0; cb_.hist_rows = 0;
    n_resampling_ = resampling_at_entry;
    eps_hist_.resize(hist_at_entry[0]); u_hist_.resize(hist_at_entry[1]); rho_hist_.resize(hist_at_entry[2]);
    (void)be_->write_control(cb_);
    initialized_ = false;
    p2p_abort(rc);
    err_ = why;
  }
  return rc;
}

int Engine::update_loop(const sabc_update_args &a) {
  if (!(a.v > 0)) return fail(SABC_ERR_BAD_V, "Annealing speed `v` must be positive.");                 // :261
  if (!(a.delta > 0)) return fail(SABC_ERR_BAD_DELTA, "Resamping intensity `δ` must be positive.");     // :262
  if (a.proposal_kind < SABC_PROP_RANDOMWALK || a.proposal_kind > SABC_PROP_STRETCH)
    return fail(SABC_ERR_BAD_CONFIG, "unknown proposal kind");
  if (a.proposal_kind == SABC_PROP_RANDOMWALK && !(a.proposal_p0 > 0 && a.proposal_p0 <= 1))
    return fail(SABC_ERR_BAD_BETA, "Mixing parameter `β` must be between zero and one.");              // proposals.jl:30
  if (a.n_simulation < 0) return fail(SABC_ERR_BAD_CONFIG, "n_simulation must not be negative");
  const int64_t N = sh_.n_global;
  const int64_t n_pop = a.n_simulation / N;                                 // :275
  const int64_t n_updates = n_pop * N;                                      // :276
  // `ix % checkpoint_history` (:367): a DivideError for 0 in the reference; a negative interval divides like its magnitude
  if (n_pop > 0 && a.checkpoint_history == 0)
    return fail(SABC_ERR_BAD_CONFIG, "DivideError: integer division error (`checkpoint_history` must not be zero)");
  const int64_t cph = a.checkpoint_history < 0 ? -a.checkpoint_history : (a.checkpoint_history > 0 ? a.checkpoint_history : 1);
  // a wrapper may cut one update_population! call into several of these calls (progress lines, :359-364): update ix here is
  // update phase + ix of the loop at :294, and the histories follow THAT number
  if (a.history_phase < 0) return fail(SABC_ERR_BAD_CONFIG, "history_phase must not be negative");
  const int64_t phase = a.history_phase;
  int rc;
  // the host mirror is the truth between calls (state setters write into it)
  cb_.hist_rows = 0; cb_.error = 0; cb_.eps_len = eps_len_;
  hist_capacity_ = n_pop / cph + 3;
  if (be_->history_reserve(hist_capacity_)) return fail(SABC_ERR_HIP, "history buffer allocation failed");
  if (be_->write_control(cb_)) return fail(SABC_ERR_HIP, "writing the control block failed");

  // A chunk that CONTINUES an update_population! call (history_phase > 0, the same proposal, the particles untouched since
  // the previous chunk) starts from the control block as that chunk's last update left it -- Sigma and its factor from the
  // fused sums of that update (:348), the running sum of rho -- exactly what the uncut call works with at this point.
  // Recomputing them from a stats pass here would give the same numbers up to the summation order, i.e. an ulp, and the
  // cut call would no longer be the uncut call bit for bit.
  const bool continuation = a.history_phase > 0 && !population_replaced_ && last_prop_kind_ == a.proposal_kind &&
                            last_prop_p0_ == a.proposal_p0 && last_prop_p1_ == a.proposal_p1;
  if (a.proposal_kind == SABC_PROP_RANDOMWALK && !continuation) {           // update_proposal!, :284
    if (population_replaced_) {                                             // the pivot of the last call may be far off
      if ((rc = stats_reduce())) return rc;
      if ((rc = control(CTRL_PIVOT, &a, a.v))) return rc;                   // centre the moment sums first
    }
    if ((rc = stats_reduce())) return rc;
    if ((rc = control(CTRL_PROPOSAL, &a, a.v))) return rc;
  }
  if (n_pop > 0 && a.proposal_kind != SABC_PROP_RANDOMWALK && !host_mode_ && !continuation) {
    // the update steps only report the CHANGE of sum(rho): start the running sum from the population as it stands (the
    // caller may have replaced it with sabc_set_population since the last call)
    if ((rc = stats_reduce())) return rc;
    if ((rc = control(0, &a, a.v))) return rc;
  }
  if (n_pop > 0 && a.proposal_kind != SABC_PROP_RANDOMWALK) {
    const int64_t need = a.proposal_kind == SABC_PROP_DIFFEVO ? 2 : 1;
    for (int half = 0; half < 2; ++half)
      if (partner_view(nullptr, 0, half).m_total < need)
        return fail(SABC_ERR_BAD_CONFIG, "too few particles in a half batch for this proposal");
  }

  // Small shards with a device-coded simulator: the whole loop of :294-375 in ONE launch per stretch between two resamples
  // (kernels.hip: k_update_persistent) instead of a chain of launches per update
  int64_t first_ix = 1;                                                     // the launch chain's first update of the call
  if (n_pop > 0 && !host_mode_ && sh_.world == 1 && be_->persistent_supported(a.proposal_kind) &&
      std::chrono::steady_clock::now() >= persistent_retry_at_) {
    if ((rc = update_loop_persistent(a, n_pop, cph, phase, &first_ix))) return rc;
  }
  if (first_ix <= n_pop) {
  // The loop of :294-375 with the host two updates ahead of the device.  Every update is enqueued as
  //   k_update (x1 or x2) -> k_reduce_partials -> [allreduce] -> k_control(ACCUMULATE | CHECK | ...)
  // where the control step evaluates the resample test of :340 ON THE DEVICE.  If it does not fire, the
  // control step goes on to Sigma / eps / pivot / history and the next update -- already queued -- runs
  // without the GPU ever waiting for the host.  If it fires, it sets ControlBlock::halt; the kernels of
  // the update queued behind it see the flag and do nothing; the host learns it from the mailbox,
  // runs the resample (:341), finishes the control step, clears the flag and re-enqueues.
  const int32_t after_update = CTRL_PROPOSAL | CTRL_EPSILON | CTRL_PIVOT;   // :348-354
  constexpr int kMaxDepth = 2;
  const int kDepth = host_mode_ ? 1 : kMaxDepth;     // a host simulator leaves nothing to queue ahead
  int64_t seqs[kMaxDepth + 1] = {0};
  int64_t next_enqueue = first_ix, next_confirm = first_ix;
  int64_t known_accept = cb_.n_accept, last_delta = 0;   // as of the last confirmed update
  auto hist_flag = [&](int64_t ix) { return ((phase + ix) % cph == 0) ? (int32_t)CTRL_HISTORY : 0; };   // :367
  // Queueing ahead of an update that then triggers the resample costs three no-op launches; the accept
  // count moves slowly from one update to the next, so the host only queues ahead when the update in
  // flight is not expected to reach the threshold (a wrong guess costs one ~10 us gap, never correctness).
  auto resample_expected = [&]() {
    const double threshold = (double)(n_resampling_ + 1) * a.resample;
    return (double)known_accept + 1.25 * (double)last_delta >= threshold;
  };
  while (next_confirm <= n_pop) {                                           // :294
    while (next_enqueue <= n_pop && next_enqueue - next_confirm < kDepth &&
           (next_enqueue == next_confirm || !resample_expected())) {
      const int64_t ix = next_enqueue;
      if ((rc = enqueue_update(a, (uint64_t)(n_population_updates_ + ix), /*guarded=*/true))) return rc;
      const double threshold = (double)(n_resampling_ + 1) * a.resample;    // :340
      if ((rc = control(CTRL_GUARDED | CTRL_ACCUMULATE | CTRL_CHECK | after_update | hist_flag(ix), &a, a.v,
                        /*notify=*/true, threshold, &seqs[ix % (kMaxDepth + 1)])))
        return rc;
      ++next_enqueue;
    }
    const int64_t ix = next_confirm;
    int64_t n_accept_now = 0;
    int halted = 0;
    if ((rc = wait_step(seqs[ix % (kMaxDepth + 1)], &n_accept_now, &halted))) return rc;
    last_delta = n_accept_now - known_accept;
    known_accept = n_accept_now;
    if (halted) {
      // n_accept >= (n_resampling + 1) * resample after update ix (:340): resample, then the part of
      // the control step that was skipped; updates queued behind ix were no-ops and are enqueued again
      const uint64_t iter = (uint64_t)(n_population_updates_ + ix);
      if ((rc = resample(a.delta, iter))) return rc;                        // :341
      n_resampling_ += 1;                                                   // :342
      if ((rc = control(CTRL_CLEAR_HALT | after_update | hist_flag(ix), &a, a.v))) return rc;
      next_enqueue = ix + 1;
    }
    ++next_confirm;
  }
  }
  // :378-382 -- `last_checkpoint_epsilon != n_population_updates` is `the loop's last update was not a checkpoint`; only the
  // call that ends the loop stores the last value
  if (!a.more_chunks_follow && (phase + n_pop) % cph != 0) {
    if ((rc = control(CTRL_HISTORY | CTRL_KEEP_SUMS, &a, a.v))) return rc;
  }
  if ((rc = p2p_commit_ok())) return rc;
  if ((rc = sync_control())) return rc;
  if ((rc = drain_history())) return rc;
  population_replaced_ = false;                                             // the pivot now follows the population
  if (n_pop > 0) { last_prop_kind_ = a.proposal_kind; last_prop_p0_ = a.proposal_p0; last_prop_p1_ = a.proposal_p1; }
  n_simulation_ += n_updates;                                               // :391
  n_population_updates_ += n_pop;                                           // :394
  return 0;
}

}  // namespace sabc
