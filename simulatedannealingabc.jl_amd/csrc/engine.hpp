// engine.hpp -- host side of the population update loop: what `sabc`, `initialization`,
// `update_population!`, `resample_population`, `update_epsilon_*` and `update_proposal!`
// (SimulatedAnnealingABC.jl:92-137,151-227,251-402; proposals.jl:46-60) do BETWEEN the
// per-particle kernels.  Pure C++ against two small interfaces:
//   Backend      -- where the particles live and the kernels that touch them (HipBackend in
//                   hip_backend.hip is the product; tests/ plugs a CPU stand-in built on the oracle
//                   to exercise this file without a GPU);
//   Collectives  -- allreduce / allgather across shards (one process per GPU).
#pragma once
#include <chrono>
#include <cstdint>
#include <string>
#include <vector>
#include "sabc_types.hpp"

namespace sabc {

class Collectives {
 public:
  virtual ~Collectives() = default;
  // pointers are in the Backend's memory space (device memory for HipBackend)
  virtual int allreduce_sum(double *buf, int64_t count) = 0;
  virtual int allgather(const double *send, double *recv, int64_t count_per_rank) = 0;
  // personalised exchange: send_counts[p] doubles go to rank p (consecutive segments of `send`), recv_counts[p] arrive from
  // rank p (consecutive segments of `recv`); the count arrays are host memory.  Optional: without it the resample falls
  // back to an allgather of the whole population.
  virtual bool usable() const { return true; }      // false: the placeholder of a handle nobody gave a transport
  virtual bool has_alltoallv() const { return false; }
  virtual int alltoallv(const double *, const int64_t *, double *, const int64_t *) { return -1; }
};

class Backend {
 public:
  virtual ~Backend() = default;
  virtual int allocate(const ModelDesc &m, const Shard &sh) = 0;
  // blocks in backend memory; pop_block() may change after resample_draw()
  virtual double *pop_block() = 0;                  // [(d+s+1)][cap]: theta rows, u rows, weight row
  virtual double *rho_block() = 0;                  // [s][cap]
  virtual double *sums_buffer() = 0;                // staging for the fused sums: reduction and allreduce target
  virtual double *gather_buffer(int64_t doubles) = 0;
  // growable scratch areas in backend memory (which = 0..3), contents not preserved across a growth
  virtual double *scratch_buffer(int which, int64_t doubles) = 0;
  // rows x count doubles from src (row pitch src_pitch) to dst (row pitch dst_pitch), both in backend memory, in stream order
  virtual int copy_rows(const double *src, int64_t src_pitch, double *dst, int64_t dst_pitch, int rows, int64_t count) = 0;
  // small host <-> backend-memory transfers that complete before they return (counts of the resample exchange)
  virtual int to_backend(double *dst, const double *src_host, int64_t n) = 0;
  virtual int to_host(double *dst_host, const double *src, int64_t n) = 0;
  // host-simulator mode (SABC_MODEL_HOST): f_dist is a host callback; a backend without it says so
  virtual int set_host_simulator(sabc_simulate_fn, void *) { return -1; }
  virtual int set_host_prior(sabc_prior_sample_fn, sabc_prior_logpdf_fn, void *) { return -1; }   // prior_joint = 2
  virtual int host_prior_simulate() { return -1; }                      // :172-179 with f_dist on the host
  virtual int host_update_range(const StepArgs &, const PartnerView &, int64_t, int64_t) { return -1; }   // :308-331
  virtual int host_stats(int64_t *) { return -1; }                      // moment sums + the update's accept count
  // K1
  virtual int prior_simulate() = 0;
  // K2: gathered_rho is [world][s][cap]; len_out[j] = knot count (<= 0: no positive entry)
  virtual int build_cdf(const double *gathered_rho, int64_t *len_out, int *any_negative) = 0;
  // K3
  virtual int cdf_population() = 0;
  // K4 on local particles [lo, lo+cnt); partial rows start at row0; returns rows written.
  // eps, the Cholesky factor and the pivot are read from the control block.
  virtual int update_range(const StepArgs &c, const PartnerView &pv, int64_t lo, int64_t cnt, int64_t row0,
                           int64_t *rows_out) = 0;
  // K4 for small shards: `count` population updates in ONE launch, starting with update ix0 of the call (global index
  // c.iter) -- per update the body, the sums and the control step `ctrl` (+ a history row by the cadence (phase + ix) % cph),
  // until the resample test fires or an error is raised.  Blocks until the launch has ended; done = updates completed (the
  // last of them the one that halted), halted / error from the control block.  persistent_supported: this backend has such a
  // form for this proposal on its shard (the engine then takes it instead of a launch chain per update).
  virtual bool persistent_supported(int /*prop_kind*/) const { return false; }
  virtual int persistent_lanes() const { return 0; }   // lanes per particle of the last such launch (1 | 4), 0: none yet
  virtual int update_persistent(const StepArgs &, const ControlArgs &, const PartnerView &, const PartnerView &, int64_t /*ix0*/,
                                int64_t /*phase*/, int64_t /*cph*/, int64_t /*count*/, int64_t * /*done*/, int * /*halted*/,
                                int * /*error*/) { return -1; }
  virtual int stats(int64_t *rows_out) = 0;
  // -> ControlBlock::sums; guarded: part of a queued-ahead step (no-op while ControlBlock::halt is set)
  virtual int reduce_partials(int64_t rows, bool guarded) = 0;
  // state hand-over between updates (control.hpp), enqueued like a kernel
  virtual int control(const ControlArgs &a) = 0;
  // wait until the control step enqueued with notify_seq == seq has run; cheap (mailbox poll)
  virtual int wait_notify(int64_t seq, int64_t *n_accept, int *error, int *halted) = 0;
  virtual int read_control(ControlBlock *out) = 0;  // blocks until the stream has drained
  virtual int write_control(const ControlBlock &in) = 0;
  virtual int history_reserve(int64_t rows) = 0;    // capacity of the device-side history buffer
  virtual int read_history(double *out, int64_t rows, int row_len) = 0;
  // K5
  virtual int resample_weights(double delta) = 0;   // ubar comes from ControlBlock::sums
  virtual int resample_draw(const double *gathered_pop, uint64_t iter) = 0;
  // K5 on shards without moving the whole population (SimulatedAnnealingABC.jl:129-132):
  //  select : running sum of the gathered weights [world][cap] + this shard's n_local draws -> global source indices
  //  bucket : group the draws by owner shard; counts_host[r] = draws owned by shard r; req_out (n_local doubles) = their
  //           offsets inside the owner, bucket after bucket; returns after the counts are known on the host
  //  serve  : rows (theta, u) of the CURRENT population at the m requested offsets -> rows_out, one row of d + s per request
  //  scatter: rows_in (bucket order of `bucket`) -> the resampled population (rho is NOT permuted, :131-132)
  virtual int resample_select(const double *gathered_w, uint64_t iter) = 0;
  virtual int resample_bucket(int64_t *counts_host, double *req_out) = 0;
  virtual int resample_serve(const double *req_in, int64_t m, double *rows_out) = 0;
  virtual int resample_scatter(const double *rows_in) = 0;
  // K5 on ONE shard, everything in one call: weights, scan, draws, gather.  A backend that also leaves the moment sums of
  // the resampled population in its partial rows says how many through stats_rows (the engine then skips the stats pass);
  // the default runs the two steps above and leaves the sums to the caller (*stats_rows = -1).
  virtual int resample_local(double delta, uint64_t iter, int64_t *stats_rows) {
    *stats_rows = -1;
    if (resample_weights(delta)) return -1;
    return resample_draw(pop_block(), iter);
  }
  virtual double last_ess() = 0;
  // measurement: bracket what is enqueued between the two calls with timing events (kernel = SABC_KERNEL_*)
  virtual void prof_begin(int) {}
  virtual void prof_end(int) {}
  // the call's last exchange has completed on every shard: housekeeping that must not run while a peer may be waiting
  virtual void end_of_call() {}
  // ---- peer-to-peer transport (p2p.hpp): the shards of one node exchange through each other's mapped memory.  A backend
  //      without it never reports p2p_active() and the engine keeps to the Collectives.
  virtual bool p2p_active() const { return false; }
  // the reduction queued by reduce_partials() is to be summed over the shards inside the launch of the next control()
  // (or of sums_buffer()): reduce -> exchange -> control step in one kernel
  virtual int p2p_exchange_pending() { return -1; }
  // flag barrier between the shards' streams (guarded: a no-op while ControlBlock::halt is set)
  virtual int p2p_barrier(bool) { return -1; }
  // end of a call: post this shard's status (0 = fine), and on the success path wait for everyone's
  virtual int p2p_commit(int, bool) { return -1; }
  // leave the group (p2p.hpp): the peers learn of it, this shard's stream is drained, the peers are unmapped
  virtual void p2p_disable() {}
  // host-side look at the peers' pages: false when one of them has left the group (nothing is launched then)
  virtual bool p2p_peers_present() { return true; }
  // device-side copy of the particles (theta, u, rho) as they stand / put it back: what lets a call that failed over the
  // peer-to-peer transport be repeated over the Collectives without the caller noticing
  virtual int snapshot() { return -1; }
  virtual int restore_snapshot() { return -1; }
  // K2 with every shard's rho block read from its owner (barrier included)
  virtual int build_cdf_p2p(int64_t *, int *) { return -1; }
  // fills pv->peer[] with every shard's theta block in its owner's memory
  virtual int partner_view_p2p(PartnerView *) { return -1; }
  // K5 on shards with nothing gathered and no host round trip: weights -> barrier -> scan over the owners' weight rows ->
  // draws + rows read from their owners (:124-137)
  virtual int resample_p2p(double, uint64_t) { return -1; }
  // state import/export (host buffers, column-major n_local x k)
  virtual int download(double *theta, double *u, double *rho) = 0;
  virtual int upload(const double *theta, const double *u, const double *rho) = 0;
  virtual int get_knots(int stat, double *out, int64_t len) = 0;
  virtual int set_knots(int stat, const double *knots, int64_t len) = 0;
};

class Engine {
 public:
  Engine(const sabc_config &cfg, Backend *backend, Collectives *coll);

  int validate();                                   // config errors found at create time
  int initialize(int64_t n_simulation);             // initialization(), :151-227
  int update(const sabc_update_args &a);            // update_population!(), :251-402

  const std::string &error() const { return err_; }
  const Shard &shard() const { return sh_; }
  const ModelDesc &model() const { return m_; }
  int eps_len() const { return eps_len_; }
  const double *eps() const { return cb_.eps; }
  int set_eps(const double *e, int len);
  void counters(int64_t out[4]) const;
  void set_counters(const int64_t in[4]);
  int64_t history_len() const { return (int64_t)(eps_hist_.size() / (size_t)eps_len_); }
  void history(double *e, double *u, double *r) const;
  void clear_history() { eps_hist_.clear(); u_hist_.clear(); rho_hist_.clear(); }
  const int64_t *cdf_len() const { return cdf_len_; }
  void set_cdf_len(int stat, int64_t len) { cdf_len_[stat] = len; }
  const double *sigma() const { return cb_.sigma; }
  bool initialized() const { return initialized_; }
  void mark_initialized() { initialized_ = true; population_replaced_ = true; }   // sabc_set_population
  void set_collectives(Collectives *c) { coll_ = c; }
  // how often update() had to wait for the device (one per run-ahead window), for measurement
  int64_t host_syncs() const { return host_syncs_; }
  int64_t persistent_launches() const { return persistent_launches_; }
  int64_t persistent_fallbacks() const { return persistent_fallbacks_; }
  int persistent_lanes() const { return be_->persistent_lanes(); }
  // bytes that landed in this shard's receive buffers through collectives so far (allreduce: the vector; allgather: all
  // blocks; alltoallv: what arrived)
  int64_t comm_bytes() const { return comm_bytes_; }
  int64_t collective_calls() const { return collective_calls_; }
  // calls that were put back and repeated over the Collectives because a peer-to-peer wait gave up
  int64_t p2p_fallbacks() const { return p2p_fallbacks_; }
  // the peer-to-peer transport carries this handle (the backend has it mapped and the simulator is device code: a host
  // callback's duration differs from shard to shard by more than any sensible wait bound)
  bool p2p() const { return be_->p2p_active() && !host_mode_ && sh_.world > 1; }
  bool host_mode() const { return host_mode_; }

 private:
  int fail(int code, const std::string &msg) { err_ = msg; return code; }
  int global_reduce(int64_t rows, bool guarded = false);   // block partials -> shard sums -> allreduce
  int stats_reduce();                               // sums of the population as it stands
  // enqueue a control step; returns its mailbox sequence number through seq_out when notify is set
  int control(int32_t mode, const sabc_update_args *a, double v, bool notify = false, double threshold = 0.0,
              int64_t *seq_out = nullptr);
  int wait_step(int64_t seq, int64_t *n_accept, int *halted);   // poll the mailbox ring
  int sync_control();                               // device -> cb_, raises device-side errors
  int resample(double delta, uint64_t iter);        // :124-137
  int resample_exchange(uint64_t iter);             // the sharded form: weights allgather + exchange of the drawn rows
  int allreduce(double *buf, int64_t count);        // coll_ + byte accounting
  int allgather(const double *send, double *recv, int64_t count_per_rank);
  int alltoallv(const double *send, const int64_t *sc, double *recv, const int64_t *rc);
  // where the partners of the half batch `1 - inactive_half` are read from (all shards' inactive halves)
  int partner_source(int inactive_half, PartnerView *pv);
  int enqueue_update(const sabc_update_args &a, uint64_t iter, bool guarded);
  int update_loop(const sabc_update_args &a);       // update() minus the error contract
  int update_loop_persistent(const sabc_update_args &a, int64_t n_pop, int64_t cph, int64_t phase, int64_t *next_ix);   // small shards: one launch
  int update_once(const sabc_update_args &a);       // update_loop + the error contract; update() adds the transport fallback
  int drain_history();
  PartnerView partner_view(const double *base, int64_t rank_stride, int inactive_half) const;

  sabc_config cfg_;
  ModelDesc m_;
  Shard sh_;
  Backend *be_;
  Collectives *coll_;
  std::string err_;
  bool initialized_ = false;
  bool population_replaced_ = true;                 // the particles may lie anywhere relative to ControlBlock::pivot
  int last_prop_kind_ = -1;                         // the proposal of the last call that ran an update (a chunk that continues it
  double last_prop_p0_ = 0.0, last_prop_p1_ = 0.0;  // starts from the control block as it stands: update_loop)

  int np_ = 0, eps_len_ = 1;
  bool host_mode_ = false;                          // f_dist (SABC_MODEL_HOST) and / or the prior (prior_joint = 2) are host callbacks
  ControlBlock cb_;                                 // host mirror, current after every public call
  int64_t hist_capacity_ = 0;
  int64_t cdf_len_[kMaxStats] = {0};
  int64_t n_simulation_ = 0, n_resampling_ = 0, n_population_updates_ = 0;
  int64_t host_syncs_ = 0, notify_seq_ = 0, comm_bytes_ = 0, collective_calls_ = 0, p2p_fallbacks_ = 0, persistent_launches_ = 0, persistent_fallbacks_ = 0;
  std::chrono::steady_clock::time_point persistent_retry_at_{};    // no one-launch update before this (after one that found the device too full)
  ControlArgs control_args(int32_t mode, const sabc_update_args *a, double v, double threshold) const;
  int p2p_check_peers();                            // p2p: a peer has left the group? (entry of a call, nothing launched yet)
  int p2p_commit_ok();                              // p2p: the end-of-call status exchange (success path)
  void p2p_abort(int rc);                           // ... and after a failure (rc: what the call returns)
  int initialize_body();
  std::vector<double> eps_hist_, u_hist_, rho_hist_;
};

}  // namespace sabc
