// ref_backend.cpp -- TEST INFRASTRUCTURE (never shipped, never loaded by the product package).
//
// Runs the product's host engine (simulatedannealingabc.jl_amd/csrc/engine.cpp + control.hpp:
// sharding arithmetic, run-ahead windows, resample decisions, collectives sequencing, history)
// on a CPU with a Backend whose "kernels" are loops over the oracle's per-particle functions
// (oracle/sabc_oracle.c).  Purpose: exercise the world > 1 code path with torch.distributed
// "gloo" where no GPU exists, and give the sharded HIP runs a reference with the SAME shard
// colouring.  It exports the subset of include/sabc_hip.h the tests need, under the same names.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sabc_hip.h"
#include "../../oracle/sabc_oracle.h"
#include "../../simulatedannealingabc.jl_amd/csrc/control.hpp"
#include "../../simulatedannealingabc.jl_amd/csrc/engine.hpp"
#include "../../simulatedannealingabc.jl_amd/csrc/p2p.hpp"
#include "../../simulatedannealingabc.jl_amd/csrc/p2p_setup.hpp"

using namespace sabc;

namespace {

constexpr int kBlockRows = 256;   // same partial-row granularity as the device kernels

class RefBackend : public Backend {
 public:
  int allocate(const ModelDesc &m, const Shard &sh) override {
    m_ = m; sh_ = sh; np_ = n_partials(m.d, m.s);
    rows_ = m.d + m.s + 1;
    for (auto &p : pop_) p.assign((size_t)rows_ * sh.cap, 0.0);
    rho_.assign((size_t)m.s * sh.cap, 0.0);
    knots_.assign((size_t)m.s * (sh.n_global + 2), 0.0);
    partials_.assign((size_t)(2 * ((sh.cap + kBlockRows - 1) / kBlockRows) + 4) * np_, 0.0);
    std::memset(&cb_, 0, sizeof(cb_));
    std::memset(&oc_, 0, sizeof(oc_));
    oc_.n_particles = sh.n_global; oc_.n_para = m.d; oc_.n_stats = m.s;
    oc_.model_id = m.model_id; oc_.n_model_params = m.n_model_params;
    for (int i = 0; i < ORC_MAX_MODEL_PARAMS; ++i) oc_.model_params[i] = m.p[i];
    for (int k = 0; k < ORC_MAX_PARA; ++k) { oc_.prior_kind[k] = m.prior_kind[k]; oc_.prior_a[k] = m.prior_a[k]; oc_.prior_b[k] = m.prior_b[k];
                                             oc_.prior_c[k] = m.prior_c[k]; oc_.prior_d[k] = m.prior_d[k]; }
    oc_.prior_joint = m.prior_joint;
    for (int k = 0; k < m.d; ++k)
      for (int l = 0; l < m.d; ++l) oc_.prior_chol[k * m.d + l] = m.prior_L[k * m.d + l];
    oc_.seed = m.seed;
    return 0;
  }
  double *pop_block() override { return pop_[cur_].data(); }
  double *rho_block() override { return rho_.data(); }
  double *sums_buffer() override { return stage_; }
  double *gather_buffer(int64_t doubles) override {
    if ((int64_t)gather_.size() < doubles) gather_.resize((size_t)doubles);
    return gather_.data();
  }

  double *scratch_buffer(int which, int64_t doubles) override {
    if (which < 0 || which >= 4) return nullptr;
    if ((int64_t)scratch_[which].size() < doubles) scratch_[which].resize((size_t)doubles);
    return scratch_[which].data();
  }
  int copy_rows(const double *src, int64_t sp, double *dst, int64_t dp, int rows, int64_t count) override {
    for (int r = 0; r < rows; ++r) std::memcpy(dst + (size_t)r * dp, src + (size_t)r * sp, (size_t)count * sizeof(double));
    return 0;
  }
  int to_backend(double *dst, const double *src, int64_t n) override { std::memcpy(dst, src, (size_t)n * sizeof(double)); return 0; }
  int to_host(double *dst, const double *src, int64_t n) override { std::memcpy(dst, src, (size_t)n * sizeof(double)); return 0; }

  int prior_simulate() override {
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap;
    for (int64_t li = 0; li < sh_.n_local; ++li) {
      double th[ORC_MAX_PARA], r[ORC_MAX_STATS];
      const uint64_t gid = (uint64_t)(sh_.gid0 + li);
      orc_prior_sample(&oc_, gid, th);
      if (orc_simulate(&oc_, th, gid, 0, r)) return -1;
      for (int k = 0; k < d; ++k) pop_[cur_][(size_t)k * cap + li] = th[k];
      for (int j = 0; j < s; ++j) rho_[(size_t)j * cap + li] = r[j];
    }
    return 0;
  }

  int build_cdf(const double *g, int64_t *len_out, int *any_negative) override {
    const int s = m_.s; const int64_t cap = sh_.cap, N = sh_.n_global;
    *any_negative = 0;
    std::vector<double> col((size_t)N);
    for (int j = 0; j < s; ++j) {
      for (int64_t gid = 0; gid < N; ++gid) {
        const int64_t r = gid / cap, o = gid - r * cap;
        col[(size_t)gid] = g[(r * s + j) * cap + o];
        if (col[(size_t)gid] < 0) *any_negative = 1;
      }
      const int64_t len = orc_build_cdf(col.data(), N, knots_.data() + (size_t)j * (N + 2));
      cdf_len_[j] = len > 0 ? len : 0;
      len_out[j] = cdf_len_[j];
    }
    return 0;
  }

  int cdf_population() override {
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap, N = sh_.n_global;
    for (int j = 0; j < s; ++j)
      for (int64_t li = 0; li < sh_.n_local; ++li)
        pop_[cur_][(size_t)(d + j) * cap + li] =
            orc_cdf_apply(knots_.data() + (size_t)j * (N + 2), cdf_len_[j], rho_[(size_t)j * cap + li]);
    return 0;
  }

  static const double *partner(const PartnerView &pv, uint64_t j) {
    int64_t r = (int64_t)(j / (uint64_t)pv.m_full);
    if (r > pv.world - 1) r = pv.world - 1;
    const int64_t o = (int64_t)j - r * pv.m_full;
    const int64_t off = (r == pv.world - 1) ? pv.off_last : pv.off_full;
    return (pv.direct ? pv.peer[r] : pv.base + r * pv.rank_stride) + off + o;
  }
  static uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

  // rho = the particle's distances (stats pass) or their change on acceptance (update step: ControlArgs::rho_is_delta)
  void moments(bool acc, const double *th, const double *u, const double *rho, double *row) const {
    const int d = m_.d, s = m_.s;
    row[0] += acc ? 1.0 : 0.0;
    for (int j = 0; j < s; ++j) { row[1 + j] += u[j]; row[1 + s + j] += rho[j]; }
    double dk[ORC_MAX_PARA];
    for (int k = 0; k < d; ++k) { dk[k] = th[k] - cb_.pivot[k]; row[1 + 2 * s + k] += dk[k]; }
    int q = 1 + 2 * s + d;
    for (int k = 0; k < d; ++k) for (int l = 0; l <= k; ++l) row[q++] += dk[k] * dk[l];
  }

  // the per-particle body, SimulatedAnnealingABC.jl:308-331, with the engine's conventions
  int update_range(const StepArgs &c, const PartnerView &pv, int64_t lo, int64_t cnt, int64_t row0,
                   int64_t *rows_out) override {
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap, N = sh_.n_global;
    double *pop = pop_[cur_].data();
    const int64_t rows = (cnt + kBlockRows - 1) / kBlockRows;
    *rows_out = rows;
    if (cb_.halt) return 0;          // queued ahead of a resample decision that fired
    for (int64_t b = 0; b < rows; ++b) {
      double *row = &partials_[(size_t)(row0 + b) * np_];
      for (int q = 0; q < np_; ++q) row[q] = 0.0;
      for (int64_t t = b * kBlockRows; t < cnt && t < (b + 1) * kBlockRows; ++t) {
        const int64_t li = lo + t;
        const uint64_t gid = (uint64_t)(sh_.gid0 + li);
        double th[ORC_MAX_PARA], u[ORC_MAX_STATS], rho[ORC_MAX_STATS], thp[ORC_MAX_PARA], up[ORC_MAX_STATS] = {0}, rp[ORC_MAX_STATS] = {0};
        for (int k = 0; k < d; ++k) th[k] = pop[(size_t)k * cap + li];
        for (int j = 0; j < s; ++j) { u[j] = pop[(size_t)(d + j) * cap + li]; rho[j] = rho_[(size_t)j * cap + li]; }
        double logf = 0.0;
        if (c.prop_kind == SABC_PROP_RANDOMWALK) {
          double z[ORC_MAX_PARA + 1];
          for (int k = 0; k < d; k += 2) orc_normal_pair(m_.seed, gid, ORC_PURPOSE_PROP, c.iter, (uint32_t)(k / 2), &z[k]);
          for (int k = 0; k < d; ++k) {
            double a = 0.0;
            for (int l = 0; l <= k; ++l) a += cb_.chol[k * d + l] * z[l];
            thp[k] = th[k] + a;
          }
        } else if (c.prop_kind == SABC_PROP_DIFFEVO) {
          uint64_t i1 = 0, i2 = 0;
          for (uint32_t a = 0;; ++a) {
            uint32_t w[4];
            orc_stream_block(m_.seed, gid, ORC_PURPOSE_PROP, c.iter, a, w);
            i1 = mulhi64(((uint64_t)w[0] << 32) | w[1], (uint64_t)pv.m_total);
            i2 = mulhi64(((uint64_t)w[2] << 32) | w[3], (uint64_t)pv.m_total);
            if (i1 != i2 || a > 64u) break;
          }
          double z[2];
          orc_normal_pair(m_.seed, gid, ORC_PURPOSE_PROP2, c.iter, 0, z);
          const double gamma = c.prop_p0 * (1.0 + c.prop_p1 * z[0]);
          const double *p1 = partner(pv, i1), *p2 = partner(pv, i2);
          for (int k = 0; k < d; ++k) thp[k] = th[k] + gamma * (p1[(int64_t)k * pv.cap] - p2[(int64_t)k * pv.cap]);
        } else {
          uint32_t w[4];
          orc_stream_block(m_.seed, gid, ORC_PURPOSE_PROP, c.iter, 0, w);
          const uint64_t ip = mulhi64(((uint64_t)w[0] << 32) | w[1], (uint64_t)pv.m_total);
          const double U = orc_u52(w[2], w[3]);
          const double tt = (c.prop_p0 - 1.0) * U + 1.0, z = tt * tt / c.prop_p0;
          const double *p = partner(pv, ip);
          for (int k = 0; k < d; ++k) { const double pk = p[(int64_t)k * pv.cap]; thp[k] = pk + z * (th[k] - pk); }
          logf = std::log(z) * (double)(d - 1);
        }
        const double lpp = orc_prior_logpdf(&oc_, thp);
        double log_accept = -INFINITY;
        if (lpp > -INFINITY) {
          orc_simulate(&oc_, thp, gid, c.iter, rp);
          double a = 0.0;
          for (int j = 0; j < s; ++j) {
            up[j] = orc_cdf_apply(knots_.data() + (size_t)j * (N + 2), cdf_len_[j], rp[j]);
            a += (u[j] - up[j]) / (cb_.eps_len == 1 ? cb_.eps[0] : cb_.eps[j]);
          }
          log_accept = lpp - orc_prior_logpdf(&oc_, th) + a + logf;
        }
        uint32_t wa[4];
        orc_stream_block(m_.seed, gid, ORC_PURPOSE_ACCEPT, c.iter, 0, wa);
        const bool accepted = std::log(orc_u52(wa[0], wa[1])) < log_accept;
        double drho[ORC_MAX_STATS] = {0};
        if (accepted) {
          for (int k = 0; k < d; ++k) { th[k] = thp[k]; pop[(size_t)k * cap + li] = thp[k]; }
          for (int j = 0; j < s; ++j) { u[j] = up[j]; drho[j] = rp[j] - rho[j]; pop[(size_t)(d + j) * cap + li] = up[j]; rho_[(size_t)j * cap + li] = rp[j]; }
        }
        moments(accepted, th, u, drho, row);
      }
    }
    *rows_out = rows;
    return 0;
  }

  int stats(int64_t *rows_out) override {
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap;
    const double *pop = pop_[cur_].data();
    const int64_t rows = (sh_.n_local + kBlockRows - 1) / kBlockRows;
    for (int64_t b = 0; b < rows; ++b) {
      double *row = &partials_[(size_t)b * np_];
      for (int q = 0; q < np_; ++q) row[q] = 0.0;
      for (int64_t li = b * kBlockRows; li < sh_.n_local && li < (b + 1) * kBlockRows; ++li) {
        double th[ORC_MAX_PARA], u[ORC_MAX_STATS], rho[ORC_MAX_STATS];
        for (int k = 0; k < d; ++k) th[k] = pop[(size_t)k * cap + li];
        for (int j = 0; j < s; ++j) { u[j] = pop[(size_t)(d + j) * cap + li]; rho[j] = rho_[(size_t)j * cap + li]; }
        moments(false, th, u, rho, row);
      }
    }
    *rows_out = rows;
    return 0;
  }

  int reduce_partials(int64_t rows, bool guarded) override {
    if (guarded && cb_.halt) return 0;
    for (int c = 0; c < np_; ++c) {
      double v = 0.0;
      for (int64_t r = 0; r < rows; ++r) v += partials_[(size_t)r * np_ + c];
      stage_[c] = v;
    }
    return 0;
  }

  int control(const ControlArgs &a) override {
    if (xchg_pending_) {
      // the peer-to-peer exchange of the product's k_reduce_control<true>, between host threads: a step that is a no-op
      // posts nothing on any shard; a wait that gives up leaves SABC_ERR_COMM + the halt flag and tells the "host"
      xchg_pending_ = false;
      const bool noop = ((a.mode & CTRL_GUARDED) && cb_.halt) || cb_.error == SABC_ERR_COMM;
      if (noop) { post_error_if_any(a); return 0; }
      if (!p2p_sum_rows(++xseq_)) { post_error_if_any(a); return 0; }
    }
    if (!control_step(cb_, a, hist_.data(), stage_)) { post_error_if_any(a); return 0; }
    if (a.notify_seq) {
      Mailbox &mb = ring_[a.notify_seq % kMailboxRing];
      uint64_t w0, w1;
      mailbox_pack(a.notify_seq, cb_.n_accept, cb_.error, cb_.halt, &w0, &w1);     // the product's packing, round trip
      mb.w0 = w0; mb.w1 = w1;
    }
    return 0;
  }
  int wait_notify(int64_t seq, int64_t *n_accept, int *error, int *halted) override {
    const Mailbox &mb = ring_[seq % kMailboxRing];
    int32_t e = 0, hl = 0;
    if (!mailbox_unpack(mb.w0, mb.w1, seq, n_accept, &e, &hl)) return -1;    // the synchronous backend must already have posted it
    *error = e; *halted = hl;
    return 0;
  }
  int read_control(ControlBlock *out) override { *out = cb_; return 0; }
  int write_control(const ControlBlock &in) override { cb_ = in; return 0; }
  int history_reserve(int64_t rows) override {
    if ((int64_t)hist_.size() < rows * 3 * kMaxStats) hist_.resize((size_t)(rows * 3 * kMaxStats));
    return 0;
  }
  int read_history(double *out, int64_t rows, int row_len) override {
    std::memcpy(out, hist_.data(), (size_t)(rows * row_len) * sizeof(double));
    return 0;
  }

  int resample_weights(double delta) override {
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap;
    double *pop = pop_[cur_].data();
    for (int64_t li = 0; li < sh_.n_local; ++li) {
      double a = 0.0;
      for (int j = 0; j < s; ++j) a += pop[(size_t)(d + j) * cap + li] * delta / (cb_.sums[1 + j] / (double)sh_.n_global);
      pop[(size_t)(d + s) * cap + li] = std::exp(-a);
    }
    return 0;
  }

  int resample_draw(const double *g, uint64_t iter) override {
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap, N = sh_.n_global;
    std::vector<double> w((size_t)N), cum((size_t)N), bs((size_t)orc_scan_chunks(N));
    for (int64_t gid = 0; gid < N; ++gid) {
      const int64_t r = gid / cap, o = gid - r * cap;
      w[(size_t)gid] = g[(r * rows_ + (rows_ - 1)) * cap + o];
    }
    double totals[2];
    orc_weight_scan(w.data(), N, cum.data(), bs.data(), totals);
    ess_ = totals[1] > 0 ? totals[0] * totals[0] / totals[1] : 0.0;
    std::vector<double> &dst = pop_[1 - cur_];
    for (int64_t li = 0; li < sh_.n_local; ++li) {
      uint32_t w4[4];
      orc_stream_block(m_.seed, (uint64_t)(sh_.gid0 + li), ORC_PURPOSE_RESAMPLE, iter, 0, w4);
      const double t = orc_u52(w4[0], w4[1]) * totals[0];
      const int64_t idx = orc_resample_index(cum.data(), bs.data(), N, t);
      const int64_t r = idx / cap, o = idx - r * cap;
      for (int row = 0; row < d + s; ++row) dst[(size_t)row * cap + li] = g[(r * rows_ + row) * cap + o];
    }
    flip_cur();
    return 0;
  }
  // the sharded resample (engine.cpp: resample_exchange), same protocol as HipBackend
  int resample_select(const double *gw, uint64_t iter) override {
    const int64_t cap = sh_.cap, N = sh_.n_global;
    std::vector<double> w((size_t)N), cum((size_t)N), bs((size_t)orc_scan_chunks(N));
    for (int64_t gid = 0; gid < N; ++gid) { const int64_t r = gid / cap, o = gid - r * cap; w[(size_t)gid] = gw[r * cap + o]; }
    double totals[2];
    orc_weight_scan(w.data(), N, cum.data(), bs.data(), totals);
    ess_ = totals[1] > 0 ? totals[0] * totals[0] / totals[1] : 0.0;
    idx_.assign((size_t)sh_.n_local, 0);
    for (int64_t li = 0; li < sh_.n_local; ++li) {
      uint32_t w4[4];
      orc_stream_block(m_.seed, (uint64_t)(sh_.gid0 + li), ORC_PURPOSE_RESAMPLE, iter, 0, w4);
      idx_[(size_t)li] = orc_resample_index(cum.data(), bs.data(), N, orc_u52(w4[0], w4[1]) * totals[0]);
    }
    return 0;
  }
  int resample_bucket(int64_t *counts, double *req) override {
    const int W = sh_.world; const int64_t cap = sh_.cap;
    std::vector<int64_t> cur((size_t)W, 0);
    for (int r = 0; r < W; ++r) counts[r] = 0;
    for (int64_t li = 0; li < sh_.n_local; ++li) counts[idx_[(size_t)li] / cap] += 1;
    for (int r = 1; r < W; ++r) cur[(size_t)r] = cur[(size_t)r - 1] + counts[r - 1];
    slot_.assign((size_t)sh_.n_local, 0);
    for (int64_t li = 0; li < sh_.n_local; ++li) {
      const int64_t r = idx_[(size_t)li] / cap, pos = cur[(size_t)r]++;
      req[pos] = (double)(idx_[(size_t)li] - r * cap);
      slot_[(size_t)pos] = li;
    }
    return 0;
  }
  int resample_serve(const double *req, int64_t m, double *rows_out) override {
    const int rl = m_.d + m_.s; const int64_t cap = sh_.cap;
    for (int64_t q = 0; q < m; ++q)
      for (int row = 0; row < rl; ++row) rows_out[q * rl + row] = pop_[cur_][(size_t)row * cap + (int64_t)req[q]];
    return 0;
  }
  int resample_scatter(const double *rows_in) override {
    const int rl = m_.d + m_.s; const int64_t cap = sh_.cap;
    std::vector<double> &dst = pop_[1 - cur_];
    for (int64_t pos = 0; pos < sh_.n_local; ++pos)
      for (int row = 0; row < rl; ++row) dst[(size_t)row * cap + slot_[(size_t)pos]] = rows_in[pos * rl + row];
    flip_cur();
    return 0;
  }
  double last_ess() override { return ess_; }

  int download(double *theta, double *u, double *rho) override {
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap, n = sh_.n_local;
    for (int k = 0; theta && k < d; ++k) std::memcpy(theta + (size_t)k * n, &pop_[cur_][(size_t)k * cap], (size_t)n * sizeof(double));
    for (int j = 0; u && j < s; ++j) std::memcpy(u + (size_t)j * n, &pop_[cur_][(size_t)(d + j) * cap], (size_t)n * sizeof(double));
    for (int j = 0; rho && j < s; ++j) std::memcpy(rho + (size_t)j * n, &rho_[(size_t)j * cap], (size_t)n * sizeof(double));
    return 0;
  }
  int upload(const double *theta, const double *u, const double *rho) override {
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap, n = sh_.n_local;
    for (int k = 0; theta && k < d; ++k) std::memcpy(&pop_[cur_][(size_t)k * cap], theta + (size_t)k * n, (size_t)n * sizeof(double));
    for (int j = 0; u && j < s; ++j) std::memcpy(&pop_[cur_][(size_t)(d + j) * cap], u + (size_t)j * n, (size_t)n * sizeof(double));
    for (int j = 0; rho && j < s; ++j) std::memcpy(&rho_[(size_t)j * cap], rho + (size_t)j * n, (size_t)n * sizeof(double));
    return 0;
  }
  int get_knots(int stat, double *out, int64_t len) override {
    std::memcpy(out, knots_.data() + (size_t)stat * (sh_.n_global + 2), (size_t)len * sizeof(double));
    return 0;
  }
  int set_knots(int stat, const double *k, int64_t len) override {
    std::memcpy(knots_.data() + (size_t)stat * (sh_.n_global + 2), k, (size_t)len * sizeof(double));
    cdf_len_[stat] = len;
    return 0;
  }

  // ---- the peer-to-peer transport of csrc/p2p.hpp between shards living in ONE process (one host thread each): the peers'
  //      memory is the pointer itself, the slots are atomics, every wait is bounded.  Same protocol as HipBackend's kernels
  //      -- generation tags, leave words, the host page, the owner's buffer parity -- so that engine.cpp's peer-to-peer paths
  //      (exchange inside the control step, barrier between the half batches, resample and ECDF build over the owners'
  //      memory, status exchange, abort, fallback, a shard that leaves) run where no GPU exists.
  bool p2p_active() const override { return p2p_on_; }
  int p2p_exchange_pending() override { xchg_pending_ = true; return 0; }
  void p2p_disable() override { p2p_leave(); }
  const std::string &error() const { return err_; }
  int p2p_descriptor(P2PDesc *out) {
    std::memset(out, 0, sizeof(*out));
    p2p_leave();
    if (fail_export_) { err_ = "test hook: this shard cannot export"; return -1; }
    for (auto &row : slot_seq_) for (auto &x : row) x.store(0);
    for (auto &row : bar_seq_) for (auto &x : row) x.store(0);
    for (auto &x : commit_) x.store(0);
    for (auto &x : leave_) x.store(0);
    xseq_ = bseq_ = call_ = 0;
    out->magic = kP2PMagic; out->rank = sh_.rank; out->world = sh_.world; out->cap = sh_.cap; out->n_global = sh_.n_global;
    out->d = m_.d; out->s = m_.s;
    out->ptr_slots = (uint64_t)(uintptr_t)this;
    out->ptr_page = (uint64_t)(uintptr_t)&page_;
    out->cur = cur_;
    out->gen_proposal = gen_ >= kP2PMaxGen ? 1u : gen_ + 1u;
    exported_ = true;
    return 0;
  }
  int p2p_init(const P2PDesc *all) {
    uint32_t proposals[kMaxPeers] = {0};
    for (int r = 0; r < sh_.world; ++r) {
      if (all[r].magic != kP2PMagic || all[r].rank != r || all[r].world != sh_.world) { err_ = "descriptors do not match"; return -1; }
      proposals[r] = all[r].gen_proposal;
    }
    gen_ = p2p_agree_gen(proposals, sh_.world);
    flips_ = 0;
    page_.gen.store(gen_); page_.state.store(kP2PNone);
    mapped_ = true;
    if (fail_map_) { p2p_leave(); err_ = "test hook: this shard cannot map its peers"; return -1; }
    for (int r = 0; r < sh_.world; ++r) {
      peers_[r] = (RefBackend *)(uintptr_t)all[r].ptr_slots;
      peer_page_[r] = (const P2PHostPage *)(uintptr_t)all[r].ptr_page;
      peer_cur0_[r] = all[r].cur & 1;
    }
    page_.state.store(kP2PActive, std::memory_order_release);
    p2p_on_ = true;
    return 0;
  }
  // p2p.hpp "LEAVES": leaving -> leave words -> (the stream is synchronous here) -> unmap -> released
  void p2p_leave() {
    xchg_pending_ = false;
    p2p_on_ = false;
    if (!mapped_) return;
    page_.state.store(kP2PLeaving, std::memory_order_release);
    for (int r = 0; r < sh_.world; ++r)
      if (peers_[r]) peers_[r]->leave_[sh_.rank].store(((uint64_t)gen_ << 32) | 1u, std::memory_order_release);
    for (int r = 0; r < kMaxPeers; ++r) { peers_[r] = nullptr; page_.released[r].store(gen_, std::memory_order_release); }
    mapped_ = false;
  }
  bool p2p_peers_present() override {
    if (!mapped_ || !p2p_on_) return true;
    for (int r = 0; r < sh_.world; ++r) {
      if (r == sh_.rank || !peer_page_[r]) continue;
      if (peer_page_[r]->gen.load(std::memory_order_acquire) != gen_ || peer_page_[r]->state.load(std::memory_order_acquire) != kP2PActive) return false;
    }
    return true;
  }
  // sabc_destroy: leave, then wait (bounded) for every peer's `released`; false = this object must be parked (a peer may
  // still hold the pointer)
  bool p2p_finish() {
    const P2PHostPage *pages[kMaxPeers];
    for (int r = 0; r < kMaxPeers; ++r) pages[r] = peer_page_[r];
    p2p_leave();
    bool ok = true;
    if (exported_) {
      const double wait_ms = destroy_wait_ms_ < 0 ? timeout_ms_ : destroy_wait_ms_;
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < sh_.world; ++r) {
        if (r == sh_.rank) continue;
        if (!pages[r] || gen_ == 0) { ok = false; continue; }
        while (pages[r]->released[sh_.rank].load(std::memory_order_acquire) != gen_ && pages[r]->gen.load(std::memory_order_acquire) <= gen_) {
          if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > wait_ms) { ok = false; break; }
          std::this_thread::yield();
        }
      }
    }
    page_.state.store(kP2PGone, std::memory_order_release);
    return ok;
  }
  void p2p_forget_export() { if (!mapped_) exported_ = false; }
  void p2p_set_destroy_wait(double ms) { destroy_wait_ms_ = ms; }
  void p2p_set_timeout(double ms) { timeout_ms_ = ms; }
  void p2p_inject_silence(int n) { loss_ = false; if (n >= 0) { skip_ = 0; silent_ = n; } else { skip_ = -n; silent_ = 1; } }
  void p2p_inject_loss(int n) { loss_ = true; skip_ = n > 0 ? n : 0; silent_ = 1; }
  void p2p_inject_stale(int n) { stale_ = n > 0 ? n : 0; }
  void p2p_inject_setup_failure(int what) { fail_export_ = what == 1; fail_map_ = what == 2; }
  int cur_parity() const { return cur_; }
  int take_silence() {                               // 0 | 1 the post is skipped | 2 it reaches this shard's own slots only
    if (skip_ > 0) { --skip_; return 0; }
    if (silent_ > 0) { --silent_; return loss_ ? 2 : 1; }
    return 0;
  }
  uint32_t tag(uint32_t seq) const { return p2p_tag(gen_, seq); }
  // 0: the word arrived; 1 + r: shard r did not post within the bound; 17 + r: shard r has left the group
  template <class A> int wait_for(A &word, uint64_t want, int r) {
    const auto t0 = std::chrono::steady_clock::now();
    while (word.load(std::memory_order_acquire) != want) {
      if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms_) return 1 + r;
      if (leave_[r].load(std::memory_order_acquire) == (((uint64_t)gen_ << 32) | 1u))
        return word.load(std::memory_order_acquire) == want ? 0 : 17 + r;
      std::this_thread::yield();
    }
    return 0;
  }
  void comm_fail(int kind, int failed, uint32_t seq) {
    cb_.error = SABC_ERR_COMM; cb_.halt = 1;
    cb_.comm_where = ((kind + (failed > 16 ? 4 : 0)) << 24) | (((failed - 1) & 15) << 20) | (int)(seq & kP2PSeqMask);
  }
  bool p2p_sum_rows(uint32_t seq) {
    const int W = sh_.world, ring = (int)(seq % kP2PRing);
    const uint32_t t = tag(seq);
    const int sil = take_silence();
    if (sil != 1)
      for (int p = 0; p < W; ++p) {
        if (sil == 2 && p != sh_.rank) continue;
        RefBackend *q = peers_[p];
        for (int c = 0; c < np_; ++c) q->slot_row_[ring][sh_.rank][c] = stage_[c];
        q->slot_seq_[ring][sh_.rank].store(t, std::memory_order_release);
      }
    for (int r = 0; r < W; ++r)
      if (const int f = wait_for(slot_seq_[ring][r], (uint64_t)t, r)) { comm_fail(1, f, seq); return false; }
    for (int c = 0; c < np_; ++c) {
      double a = slot_row_[ring][0][c];
      for (int r = 1; r < W; ++r) a += slot_row_[ring][r][c];               // rank order, like the kernel
      stage_[c] = a;
    }
    return true;
  }
  void post_error_if_any(const ControlArgs &a) {
    if (a.notify_seq && cb_.error == SABC_ERR_COMM) {
      Mailbox &mb = ring_[a.notify_seq % kMailboxRing];
      uint64_t w0, w1;
      mailbox_pack(a.notify_seq, cb_.n_accept, cb_.error, cb_.halt, &w0, &w1);
      mb.w0 = w0; mb.w1 = w1;
    }
  }
  int p2p_barrier(bool guarded) override {
    const uint32_t seq = ++bseq_;
    if ((guarded && cb_.halt) || cb_.error == SABC_ERR_COMM) return 0;
    const int W = sh_.world, ring = (int)(seq % kP2PRing);
    const uint32_t t = tag(seq);
    const int sil = take_silence();
    if (sil != 1)
      for (int p = 0; p < W; ++p)
        if (sil != 2 || p == sh_.rank) peers_[p]->bar_seq_[ring][sh_.rank].store(t, std::memory_order_release);
    for (int r = 0; r < W; ++r)
      if (const int f = wait_for(bar_seq_[ring][r], (uint64_t)t, r)) { comm_fail(2, f, seq); return 0; }
    return 0;
  }
  int p2p_commit(int status, bool wait) override {
    const uint32_t seq = ++call_;
    const uint64_t call = tag(seq);
    const uint64_t mine = (status != 0 || cb_.error != 0) ? 1 : 0;
    const int sil = take_silence();
    if (sil != 1)
      for (int p = 0; p < sh_.world; ++p)
        if (peers_[p] && (sil != 2 || p == sh_.rank)) peers_[p]->commit_[sh_.rank].store((call << 8) | mine, std::memory_order_release);
    if (!wait) return 0;
    int failed = 0;
    for (int r = 0; r < sh_.world && !failed; ++r) {
      const auto t0 = std::chrono::steady_clock::now();
      uint64_t w;
      while (((w = commit_[r].load(std::memory_order_acquire)) >> 8) != call) {
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms_) { failed = 1 + r; break; }
        if (leave_[r].load(std::memory_order_acquire) == (((uint64_t)gen_ << 32) | 1u) && (commit_[r].load(std::memory_order_acquire) >> 8) != call) { failed = 17 + r; break; }
        std::this_thread::yield();
      }
      if (!failed && (w & 0xFF) != 0) failed = 1 + r;
    }
    if (failed && cb_.error == 0) comm_fail(3, failed, seq);
    return 0;
  }
  // sabc_comm_p2p_selftest: a row through the slots, a barrier, and what the transport reads -- a tagged word in each of
  // this shard's three buffers, read by every peer through its "mapping", two rounds, the words put back
  int p2p_selftest() {
    if (!p2p_on_) { err_ = "the peer-to-peer transport is not initialised"; return -1; }
    const double keep[3] = {stage_[0], stage_[1], stage_[2]};
    for (int q = 0; q < 3; ++q) stage_[q] = (double)(sh_.rank + 1) * (q + 1);
    const int np_keep = np_;
    np_ = 3;
    const bool ok = p2p_sum_rows(++xseq_);
    np_ = np_keep;
    bool sums_ok = ok;
    for (int q = 0; ok && q < 3; ++q) sums_ok = sums_ok && stage_[q] == 0.5 * sh_.world * (sh_.world + 1) * (q + 1);
    for (int q = 0; q < 3; ++q) stage_[q] = keep[q];
    p2p_barrier(false);
    double *own[3] = {pop_[0].data(), pop_[1].data(), rho_.data()};
    double saved[3];
    int mismatches = 0;
    for (int round = 1; round <= 2; ++round) {
      for (int b = 0; b < 3; ++b) {
        if (round == 1) saved[b] = own[b][0];
        own[b][0] = pattern(round, sh_.rank, b);
      }
      p2p_barrier(false);
      const int expect = (round == 2 && stale_ > 0) ? 3 : round;
      for (int r = 0; r < sh_.world && cb_.error == 0; ++r) {
        const double *theirs[3] = {peers_[r]->pop_[0].data(), peers_[r]->pop_[1].data(), peers_[r]->rho_.data()};
        for (int b = 0; b < 3; ++b) mismatches += theirs[b][0] != pattern(expect, r, b);
      }
      p2p_barrier(false);
    }
    if (stale_ > 0) --stale_;
    for (int b = 0; b < 3; ++b) own[b][0] = saved[b];
    if (!ok || cb_.error == SABC_ERR_COMM) { err_ = "peer-to-peer self-test: a shard did not post within the bound"; cb_.error = 0; cb_.halt = 0; p2p_on_ = false; return -1; }
    if (!sums_ok) { err_ = "peer-to-peer self-test: wrong sums came back through the slots"; p2p_on_ = false; return -1; }
    if (mismatches) { err_ = "peer-to-peer self-test: words read from the shards' memory were not what their owners wrote"; p2p_on_ = false; return -1; }
    return 0;
  }
  double pattern(int round, int rank, int b) const { return 1e6 * gen_ + 1e4 * round + 100.0 * rank + b + 0.5; }
  // a peer's CURRENT population: the owner's parity at set-up + the flips since (p2p.hpp: P2PDesc::cur)
  const std::vector<double> &peer_pop_cur(int r) const { return peers_[r]->pop_[(peer_cur0_[r] ^ (int)(flips_ & 1u)) & 1]; }
  void flip_cur() { cur_ = 1 - cur_; ++flips_; }
  int build_cdf_p2p(int64_t *len_out, int *any_negative) override {
    if (p2p_barrier(false)) return -1;
    const int s = m_.s; const int64_t cap = sh_.cap;
    std::vector<double> g((size_t)sh_.world * s * cap);
    for (int r = 0; r < sh_.world; ++r) std::memcpy(&g[(size_t)r * s * cap], peers_[r]->rho_.data(), (size_t)s * cap * sizeof(double));
    return build_cdf(g.data(), len_out, any_negative);
  }
  int partner_view_p2p(PartnerView *pv) override {
    pv->direct = 1; pv->base = nullptr; pv->rank_stride = 0; pv->cap = sh_.cap;
    for (int r = 0; r < kMaxPeers; ++r) pv->peer[r] = r < sh_.world ? peer_pop_cur(r).data() : nullptr;
    return 0;
  }
  int resample_p2p(double delta, uint64_t iter) override {
    resample_weights(delta);
    if (p2p_barrier(false)) return -1;
    const int d = m_.d, s = m_.s; const int64_t cap = sh_.cap, N = sh_.n_global;
    std::vector<double> w((size_t)N), cum((size_t)N), bs((size_t)orc_scan_chunks(N));
    for (int64_t gid = 0; gid < N; ++gid) {
      const int64_t r = gid / cap, o = gid - r * cap;
      w[(size_t)gid] = peer_pop_cur((int)r)[(size_t)(d + s) * cap + o];
    }
    double totals[2];
    orc_weight_scan(w.data(), N, cum.data(), bs.data(), totals);
    ess_ = totals[1] > 0 ? totals[0] * totals[0] / totals[1] : 0.0;
    std::vector<double> &dst = pop_[1 - cur_];
    for (int64_t li = 0; li < sh_.n_local; ++li) {
      uint32_t w4[4];
      orc_stream_block(m_.seed, (uint64_t)(sh_.gid0 + li), ORC_PURPOSE_RESAMPLE, iter, 0, w4);
      const int64_t idx = orc_resample_index(cum.data(), bs.data(), N, orc_u52(w4[0], w4[1]) * totals[0]);
      const int64_t r = idx / cap, o = idx - r * cap;
      for (int row = 0; row < d + s; ++row) dst[(size_t)row * cap + li] = peer_pop_cur((int)r)[(size_t)row * cap + o];
    }
    flip_cur();
    return 0;
  }
  int snapshot() override { snap_pop_ = pop_[cur_]; snap_rho_ = rho_; return 0; }
  int restore_snapshot() override {
    if (snap_pop_.empty()) return -1;
    pop_[cur_] = snap_pop_; rho_ = snap_rho_; xchg_pending_ = false;
    return 0;
  }

 private:
  RefBackend *peers_[kMaxPeers] = {nullptr};
  const P2PHostPage *peer_page_[kMaxPeers] = {nullptr};
  int peer_cur0_[kMaxPeers] = {0};
  uint32_t flips_ = 0, gen_ = 0;
  // this shard's host page.  The shards share the process, so "a mapping of the reader's own" is modelled by never freeing it:
  // a peer may still be polling it for `released` when this object is deleted
  P2PHostPage &page_ = *new P2PHostPage();
  bool mapped_ = false, exported_ = false;
  double slot_row_[kP2PRing][kMaxPeers][kMaxPartials] = {};
  std::atomic<uint64_t> slot_seq_[kP2PRing][kMaxPeers] = {};
  std::atomic<uint64_t> bar_seq_[kP2PRing][kMaxPeers] = {};
  std::atomic<uint64_t> commit_[kMaxPeers] = {};
  std::atomic<uint64_t> leave_[kMaxPeers] = {};
  bool p2p_on_ = false, xchg_pending_ = false;
  uint32_t xseq_ = 0, bseq_ = 0, call_ = 0;
  double timeout_ms_ = 5000.0, destroy_wait_ms_ = -1.0;
  int silent_ = 0, skip_ = 0, stale_ = 0;
  bool fail_export_ = false, fail_map_ = false, loss_ = false;
  std::string err_;
  std::vector<double> snap_pop_, snap_rho_;
  ModelDesc m_{};
  Shard sh_{};
  orc_config oc_{};
  int np_ = 0, rows_ = 0, cur_ = 0;
  std::vector<double> pop_[2], rho_, knots_, partials_, gather_, hist_, scratch_[4];
  std::vector<int64_t> idx_, slot_;
  int64_t cdf_len_[kMaxStats] = {0};
  ControlBlock cb_{};
  Mailbox ring_[kMailboxRing] = {{kMailboxEmpty, kMailboxEmpty}, {kMailboxEmpty, kMailboxEmpty}, {kMailboxEmpty, kMailboxEmpty}, {kMailboxEmpty, kMailboxEmpty},
                                 {kMailboxEmpty, kMailboxEmpty}, {kMailboxEmpty, kMailboxEmpty}, {kMailboxEmpty, kMailboxEmpty}, {kMailboxEmpty, kMailboxEmpty}};
  static_assert(kMailboxRing == 8, "initialiser above");
  double stage_[kMaxPartials] = {0};
  double ess_ = 0.0;
};

class NoColl : public Collectives {
 public:
  int allreduce_sum(double *, int64_t) override { return 0; }
  int allgather(const double *, double *, int64_t) override { return -1; }
  bool usable() const override { return false; }
};

class HookColl : public Collectives {
 public:
  HookColl(sabc_allreduce_fn ar, sabc_allgather_fn ag, void *ctx, int world) : ar_(ar), ag_(ag), ctx_(ctx), world_(world) {}
  void set_alltoallv(sabc_alltoallv_fn fn) { a2a_ = fn; }
  bool has_alltoallv() const override { return a2a_ != nullptr; }
  int alltoallv(const double *send, const int64_t *sc, double *recv, const int64_t *rc) override {
    return a2a_(ctx_, send, sc, recv, rc, world_, nullptr);
  }
  int allreduce_sum(double *buf, int64_t count) override { return ar_(ctx_, buf, count, nullptr); }
  int allgather(const double *send, double *recv, int64_t count) override { return ag_(ctx_, send, recv, count, nullptr); }
 private:
  sabc_allreduce_fn ar_; sabc_allgather_fn ag_; void *ctx_; int world_;
  sabc_alltoallv_fn a2a_ = nullptr;
};

thread_local std::string g_err;

}  // namespace

struct sabc_handle {
  Engine *eng = nullptr;
  RefBackend *be = nullptr;
  Collectives *coll = nullptr;
  std::string err;
};

extern "C" {

int sabc_abi_version(void) { return SABC_ABI_VERSION; }
const char *sabc_last_global_error(void) { return g_err.c_str(); }
int sabc_device_count(void) { return 0; }

static std::atomic<int64_t> g_parked{0};
void sabc_destroy(sabc_handle *h) {
  if (!h) return;
  // the life cycle of p2p.hpp: leave, wait for the peers' `released`; a backend a peer may still point to is parked
  const bool free_it = h->be->p2p_finish();
  delete h->eng; delete h->coll;
  if (free_it) delete h->be; else g_parked += 1;
  delete h;
}

int sabc_create(const sabc_config *cfg, sabc_handle **out) {
  *out = nullptr;
  sabc_handle *h = new sabc_handle();
  h->be = new RefBackend();
  h->coll = new NoColl();
  h->eng = new Engine(*cfg, h->be, h->coll);
  const int rc = h->eng->validate();
  if (rc) { g_err = h->eng->error(); sabc_destroy(h); return rc; }
  h->be->allocate(h->eng->model(), h->eng->shard());
  *out = h;
  return 0;
}

const char *sabc_last_error(const sabc_handle *h) { return h ? h->err.c_str() : g_err.c_str(); }

int sabc_set_collectives(sabc_handle *h, sabc_allreduce_fn ar, sabc_allgather_fn ag, void *ctx, int) {
  delete h->coll;
  h->coll = new HookColl(ar, ag, ctx, h->eng->shard().world);
  h->eng->set_collectives(h->coll);
  return 0;
}

int sabc_set_alltoallv(sabc_handle *h, sabc_alltoallv_fn fn) {
  HookColl *c = dynamic_cast<HookColl *>(h->coll);
  if (!c || !fn) { h->err = "sabc_set_alltoallv needs sabc_set_collectives first"; return SABC_ERR_COMM; }
  c->set_alltoallv(fn);
  return 0;
}

int64_t sabc_comm_bytes(const sabc_handle *h) { return h->eng->comm_bytes(); }

int sabc_comm_selftest(sabc_handle *h) {
  const Shard &sh = h->eng->shard();
  const int world = sh.world;
  if (world == 1) return 0;
  double *g = h->be->gather_buffer((int64_t)(world + 1) * 4);
  g[0] = sh.rank + 1; g[1] = 2.0; g[2] = 3.0; g[3] = 100 + sh.rank;
  if (h->coll->allgather(g, g + 4, 4) || h->coll->allreduce_sum(g, 4)) { h->err = "self-test collective failed"; return SABC_ERR_COMM; }
  bool ok = g[0] == 0.5 * world * (world + 1) && g[1] == 2.0 * world;
  for (int r = 0; r < world; ++r) ok = ok && g[4 + 4 * r] == (double)(r + 1) && g[4 + 4 * r + 3] == (double)(100 + r);
  if (!ok) { h->err = "self-test collective gave wrong values"; return SABC_ERR_COMM; }
  if (h->coll->has_alltoallv()) {
    std::vector<int64_t> sc((size_t)world), rc((size_t)world);
    int64_t ns = 0, nr = 0;
    for (int p = 0; p < world; ++p) { sc[(size_t)p] = rc[(size_t)p] = sh.rank + p + 1; ns += sc[(size_t)p]; nr += rc[(size_t)p]; }
    std::vector<double> out((size_t)ns), in((size_t)nr, 0.0);
    int64_t o = 0;
    for (int p = 0; p < world; ++p)
      for (int64_t k = 0; k < sc[(size_t)p]; ++k) out[(size_t)o++] = 1000.0 * sh.rank + p;
    if (h->coll->alltoallv(out.data(), sc.data(), in.data(), rc.data())) { h->err = "self-test alltoallv failed"; return SABC_ERR_COMM; }
    o = 0;
    for (int p = 0; p < world; ++p)
      for (int64_t k = 0; k < rc[(size_t)p]; ++k)
        if (in[(size_t)o++] != 1000.0 * p + sh.rank) { h->err = "self-test alltoallv gave wrong words"; return SABC_ERR_COMM; }
  }
  return 0;
}

int sabc_initialize(sabc_handle *h, int64_t n_simulation) {
  const int rc = h->eng->initialize(n_simulation);
  if (rc) h->err = h->eng->error();
  return rc;
}
int sabc_update(sabc_handle *h, const sabc_update_args *a) {
  const int rc = h->eng->update(*a);
  if (rc) h->err = h->eng->error();
  return rc;
}
int64_t sabc_n_global(const sabc_handle *h) { return h ? h->eng->shard().n_global : 0; }
int64_t sabc_n_local(const sabc_handle *h) { return h->eng->shard().n_local; }
int64_t sabc_local_offset(const sabc_handle *h) { return h->eng->shard().gid0; }
int sabc_get_population(sabc_handle *h, double *t, double *u, double *r) { return h->be->download(t, u, r); }
int sabc_set_population(sabc_handle *h, const double *t, const double *u, const double *r) {
  h->be->upload(t, u, r); h->eng->mark_initialized(); return 0;
}
int sabc_get_counters(const sabc_handle *h, int64_t out[4]) { h->eng->counters(out); return 0; }
int sabc_set_counters(sabc_handle *h, const int64_t in[4]) { h->eng->set_counters(in); return 0; }
int sabc_get_epsilon(const sabc_handle *h, double *eps, int32_t *len) {
  for (int i = 0; i < h->eng->eps_len(); ++i) eps[i] = h->eng->eps()[i];
  if (len) *len = h->eng->eps_len();
  return 0;
}
int sabc_set_epsilon(sabc_handle *h, const double *eps, int32_t len) { return h->eng->set_eps(eps, len); }
int64_t sabc_history_len(const sabc_handle *h) { return h->eng->history_len(); }
int sabc_get_history(const sabc_handle *h, double *e, double *u, double *r) { h->eng->history(e, u, r); return 0; }
int sabc_clear_history(sabc_handle *h) { h->eng->clear_history(); return 0; }
int64_t sabc_cdf_len(const sabc_handle *h, int32_t stat) { return h->eng->cdf_len()[stat]; }
int sabc_get_cdf_knots(sabc_handle *h, int32_t stat, double *out) { return h->be->get_knots(stat, out, h->eng->cdf_len()[stat]); }
int sabc_get_proposal_sigma(const sabc_handle *h, double *sigma) {
  const int d = h->eng->model().d;
  for (int i = 0; i < d * d; ++i) sigma[i] = h->eng->sigma()[i];
  return 0;
}
double sabc_last_ess(const sabc_handle *h) { return h->be->last_ess(); }
int64_t sabc_host_syncs(const sabc_handle *h) { return h->eng->host_syncs(); }
int64_t sabc_collective_calls(const sabc_handle *h) { return h->eng->collective_calls(); }
int64_t sabc_kernel_launches(const sabc_handle *) { return 0; }
int64_t sabc_persistent_launches(const sabc_handle *h) { return h->eng->persistent_launches(); }
int32_t sabc_persistent_lanes(const sabc_handle *h) { return h->eng->persistent_lanes(); }
int64_t sabc_persistent_fallbacks(const sabc_handle *h) { return h->eng->persistent_fallbacks(); }

// the peer-to-peer entry points, over the in-process emulation above (shards = host threads of this process)
int sabc_comm_p2p_descriptor(sabc_handle *h, void *out) {
  if (h->be->p2p_descriptor((P2PDesc *)out)) { h->err = h->be->error(); return SABC_ERR_COMM; }
  return 0;
}
int sabc_comm_p2p_init(sabc_handle *h, const void *all) {
  if (!all || h->be->p2p_init((const P2PDesc *)all)) { h->err = "peer-to-peer descriptors do not match"; return SABC_ERR_COMM; }
  return 0;
}
int sabc_comm_p2p_selftest(sabc_handle *h) {
  if (h->be->p2p_selftest()) { h->err = h->be->error(); return SABC_ERR_COMM; }
  return 0;
}
// the product's own set-up sequence (csrc/p2p_setup.hpp) over this backend
int sabc_comm_p2p_setup(sabc_handle *h) {
  std::string note;
  const int rc = p2p_setup_sequence(h->be, h->coll, h->eng->shard(), h->eng->host_mode(), &note);
  h->err = note;
  return rc == 1 ? (h->eng->p2p() ? 1 : 0) : rc;
}
int sabc_comm_p2p_set_destroy_wait(sabc_handle *h, double ms) { h->be->p2p_set_destroy_wait(ms); return 0; }
int64_t sabc_comm_p2p_parked_bytes(void) { return g_parked.load(); }          // (here: parked backends)
int sabc_comm_p2p_inject_stale(sabc_handle *h, int32_t n) { h->be->p2p_inject_stale(n); return 0; }
int sabc_comm_p2p_inject_loss(sabc_handle *h, int32_t n) { h->be->p2p_inject_loss(n); return 0; }
// harness only: 1 = this shard's next descriptors cannot be exported, 2 = it cannot map its peers, 0 = back to normal
int sabc_test_p2p_inject_setup_failure(sabc_handle *h, int32_t what) { h->be->p2p_inject_setup_failure(what); return 0; }
// harness only: which of its two population buffers the shard stands on
int sabc_test_cur_parity(const sabc_handle *h) { return h->be->cur_parity(); }
int sabc_comm_p2p_set_timeout(sabc_handle *h, double ms) { h->be->p2p_set_timeout(ms); return 0; }
int sabc_comm_p2p_disable(sabc_handle *h) { h->be->p2p_disable(); return 0; }
int sabc_comm_p2p_active(const sabc_handle *h) { return h->eng->p2p() ? 1 : 0; }
int64_t sabc_comm_p2p_fallbacks(const sabc_handle *h) { return h->eng->p2p_fallbacks(); }
int sabc_comm_p2p_inject_silence(sabc_handle *h, int32_t n) { h->be->p2p_inject_silence(n); return 0; }

}  // extern "C"
