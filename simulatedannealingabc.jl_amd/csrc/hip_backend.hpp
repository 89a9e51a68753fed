// hip_backend.hpp -- the product Backend: particle shard resident in HBM, gfx950 kernels,
// one HIP stream.  There is no CPU path behind it.
#pragma once
#include <hip/hip_runtime_api.h>
#include <string>
#include <mutex>
#include <utility>
#include <vector>
#include "engine.hpp"
#include "kernels.hpp"
#include "p2p_page.hpp"

namespace sabc {

class HipBackend : public Backend {
 public:
  explicit HipBackend(int device);
  ~HipBackend() override;

  int allocate(const ModelDesc &m, const Shard &sh) override;
  double *pop_block() override { return pop_[cur_]; }
  double *rho_block() override { return rho_; }
  double *sums_buffer() override;
  double *gather_buffer(int64_t doubles) override;
  double *scratch_buffer(int which, int64_t doubles) override;
  int copy_rows(const double *src, int64_t src_pitch, double *dst, int64_t dst_pitch, int rows, int64_t count) override;
  int to_backend(double *dst, const double *src_host, int64_t n) override;
  int to_host(double *dst_host, const double *src, int64_t n) override;
  int set_host_simulator(sabc_simulate_fn fn, void *ctx) override { host_fn_ = fn; host_ctx_ = ctx; return 0; }
  int set_host_prior(sabc_prior_sample_fn sample, sabc_prior_logpdf_fn logpdf, void *ctx) override {
    prior_sample_fn_ = sample; prior_logpdf_fn_ = logpdf; prior_ctx_ = ctx;
    return 0;
  }
  // SABC_MODEL_USER: compile the simulator source into the update kernels (rtc.hpp); the compiler log goes to error()
  int register_device_simulator(const char *hip_source);
  int host_prior_simulate() override;
  int host_update_range(const StepArgs &c, const PartnerView &pv, int64_t lo, int64_t cnt) override;
  int host_stats(int64_t *rows_out) override;
  int prior_simulate() override;
  int build_cdf(const double *gathered_rho, int64_t *len_out, int *any_negative) override;
  int cdf_population() override;
  int update_range(const StepArgs &c, const PartnerView &pv, int64_t lo, int64_t cnt, int64_t row0,
                   int64_t *rows_out) override;
  bool persistent_supported(int prop_kind) const override;
  int persistent_lanes() const override { return persist_lanes_; }
  int update_persistent(const StepArgs &c, const ControlArgs &ctrl, const PartnerView &pv_a, const PartnerView &pv_b, int64_t ix0,
                        int64_t phase, int64_t cph, int64_t count, int64_t *done, int *halted, int *error) override;
  int stats(int64_t *rows_out) override;
  int reduce_partials(int64_t rows, bool guarded) override;
  int control(const ControlArgs &a) override;
  int wait_notify(int64_t seq, int64_t *n_accept, int *error, int *halted) override;
  int read_control(ControlBlock *out) override;
  int write_control(const ControlBlock &in) override;
  int history_reserve(int64_t rows) override;
  int read_history(double *out, int64_t rows, int row_len) override;
  int resample_weights(double delta) override;
  int resample_draw(const double *gathered_pop, uint64_t iter) override;
  int resample_local(double delta, uint64_t iter, int64_t *stats_rows) override;
  int resample_select(const double *gathered_w, uint64_t iter) override;
  int resample_bucket(int64_t *counts_host, double *req_out) override;
  int resample_serve(const double *req_in, int64_t m, double *rows_out) override;
  int resample_scatter(const double *rows_in) override;
  double last_ess() override;
  void end_of_call() override;
  void prof_begin(int kernel) override;
  void prof_end(int kernel) override;
  int64_t profile_noops(int kernel) const { return kernel >= 0 && kernel < SABC_KERNEL_COUNT ? prof_noop_[kernel] : 0; }
  int download(double *theta, double *u, double *rho) override;
  int upload(const double *theta, const double *u, const double *rho) override;
  int get_knots(int stat, double *out, int64_t len) override;
  int set_knots(int stat, const double *knots, int64_t len) override;

  // peer-to-peer transport (p2p.hpp)
  bool p2p_active() const override { return p2p_on_; }
  int p2p_exchange_pending() override { pending_xchg_ = true; return 0; }
  int p2p_barrier(bool guarded) override;
  int p2p_commit(int status, bool wait) override;
  void p2p_disable() override { (void)p2p_leave(); }
  bool p2p_peers_present() override;                // every peer's host page still says `active` for this generation
  int build_cdf_p2p(int64_t *len_out, int *any_negative) override;
  int partner_view_p2p(PartnerView *pv) override;
  int resample_p2p(double delta, uint64_t iter) override;
  int snapshot() override;
  int restore_snapshot() override;
  int p2p_descriptor(P2PDesc *out);                 // allocates the slot area on first use
  int p2p_init(const P2PDesc *all);                 // maps every peer's slots, populations and rho; switches the transport on
  int p2p_selftest();
  int p2p_leave();                                  // p2p.hpp "LEAVES": leaving -> leave words -> drain -> unmap -> released
  // the group has agreed that every shard has left and unmapped (p2p_setup.hpp): nothing exported is mapped anywhere
  void p2p_forget_export() { if (!mapped_) exported_ = false; }
  void p2p_set_destroy_wait(double ms) { destroy_wait_ms_ = ms; }
  void p2p_inject_stale(int n) { p2p_stale_ = n > 0 ? n : 0; }
  static int64_t parked_bytes();
  void p2p_set_timeout(double ms) { if (ms > 0) p2p_timeout_ms_ = ms; }
  // test hook: n > 0: the next n posts are skipped; n < 0: -n more posts go out, then one is skipped
  void p2p_inject_silence(int n) { p2p_loss_ = false; if (n >= 0) { p2p_skip_ = 0; p2p_silent_ = n; } else { p2p_skip_ = -n; p2p_silent_ = 1; } }
  // test hook: n more posts go out, then one reaches only this shard's OWN slots (a post lost on the wire: the shard itself
  // carries on with its peers' rows, they run into the bound)
  void p2p_inject_loss(int n) { p2p_loss_ = true; p2p_skip_ = n > 0 ? n : 0; p2p_silent_ = 1; }
  int64_t kernel_launches() const { return launches_; }
  // host-simulator mode: seconds spent inside the caller's callbacks so far / calls of f_dist; particles per chunk
  double host_callback_seconds() const { return host_cb_seconds_; }
  int64_t host_callback_calls() const { return host_cb_calls_; }
  void set_host_chunk(int64_t particles) { host_chunk_ = particles > 0 ? particles : 0; }

  // extras used by the C-ABI layer
  int set_stream(hipStream_t s);
  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }
  const std::string &error() const { return err_; }
  CdfPtrs cdf_ptrs() const;
  void set_cdf_len(int stat, int64_t len) { cdf_len_[stat] = len; }
  int cdf_apply_host(const double *rho, int64_t m, double *u_out);
  int simulate_host(const double *theta, int64_t n, uint64_t pid0, uint64_t iter, double *rho_out);
  int prior_host(uint64_t pid0, int64_t n, double *theta_out, double *logpdf_out);   // sabc_op_prior
  void profile_enable(int level);
  int profile_get(int kernel, double *total_ms, int64_t *launches);
  // host staging for collectives that cannot take device pointers
  double *host_stage(int64_t doubles);

 private:
  int check(hipError_t e, const char *what);
  PopPtrs pop_ptrs(int which) const;

  int device_ = 0;
  hipStream_t stream_ = nullptr;
  bool own_stream_ = false;
  char *pinned_block_ = nullptr;                          // one pinned, mapped allocation behind cb_host_, mbox_host_, totals_host_
  std::string err_;
  ModelDesc m_{};
  Shard sh_{};
  int np_ = 0;
  double *pop_[2] = {nullptr, nullptr};
  int cur_ = 0;
  double *rho_ = nullptr, *knots_ = nullptr, *coarse_ = nullptr, *mid_ = nullptr;
  int64_t mid_stride_ = 0;
  int32_t cdf_shift_[kMaxStats] = {0};
  int build_coarse(int stat);
  int64_t knot_stride_ = 0;
  int64_t cdf_len_[kMaxStats] = {0};
  double *partials_ = nullptr;
  int64_t partial_rows_ = 0;
  ControlBlock *cb_dev_ = nullptr, *cb_host_ = nullptr;   // device block + pinned staging copy
  Mailbox *mbox_host_ = nullptr, *mbox_dev_ = nullptr;    // pinned + mapped: device posts, host polls
  double *hist_dev_ = nullptr;
  int flush_reduce();                                     // launch a deferred k_reduce_partials
  int64_t pending_rows_ = -1;                             // >= 0: a reduction waits to be fused into k_control
  int64_t fuse_reduce_max_ = kFuseReduceMaxDoubles;       // partial-row matrices up to this many doubles: reduced inside the control launch
  bool pending_guarded_ = false;
  double *sums_stage_ = nullptr;                          // reduction / allreduce target, taken over by k_control
  int64_t hist_cap_ = 0;
  double *gather_ = nullptr;
  int64_t gather_cap_ = 0;
  double *scratch_[4] = {nullptr, nullptr, nullptr, nullptr};
  int64_t scratch_cap_[4] = {0, 0, 0, 0};
  int64_t *idx_dev_ = nullptr, *slot_dev_ = nullptr;      // sharded resample: drawn source indices, reply -> destination
  unsigned long long *bucket_dev_ = nullptr;              // [2][world]: counts, cursors
  unsigned long long *bucket_host_ = nullptr;             // pinned staging of the same
  double *cum_ = nullptr, *block_sums_ = nullptr, *totals_dev_ = nullptr;
  double *totals_host_ = nullptr, *totals_host_dev_ = nullptr;   // pinned + mapped: k_scan_offsets posts (sum w, sum w^2) there
  double *pack_dev_ = nullptr;                                   // one shard: packed resample lines (kernels.hpp: launch_resample_local)
  double *col_a_ = nullptr, *col_b_ = nullptr;
  void *sort_tmp_ = nullptr;
  size_t sort_tmp_bytes_ = 0;
  int64_t *meta_dev_ = nullptr;
  std::vector<double> stage_;
  // host-simulator mode
  sabc_simulate_fn host_fn_ = nullptr;
  sabc_prior_sample_fn prior_sample_fn_ = nullptr;      // prior_joint = 2: rand(prior) / logpdf(prior, .) on the host
  sabc_prior_logpdf_fn prior_logpdf_fn_ = nullptr;
  void *prior_ctx_ = nullptr;
  void *host_ctx_ = nullptr;
  // staging of a half batch: pinned host arrays (host_x_) mapped into the device (host_x_dev_), allocated once
  double *host_thp_ = nullptr, *host_rho_ = nullptr, *host_cur_ = nullptr, *host_lp2_ = nullptr;
  double *host_thp_dev_ = nullptr, *host_rho_dev_ = nullptr, *host_cur_dev_ = nullptr, *host_lp2_dev_ = nullptr;
  unsigned char *host_gate_ = nullptr, *host_gate_dev_ = nullptr;      // one byte per proposal: inside the prior's support?
  double *dev_thp_ = nullptr, *dev_aux_ = nullptr;                     // device memory: proposals, (log prior, log factor)
  double *dev_rho_prop_ = nullptr;                                     // ... and, for a device-coded simulator next to a host prior, their distances
  static constexpr int kHostMaxChunks = 64;
  unsigned long long *host_flag_ = nullptr, *host_flag_dev_ = nullptr;   // per chunk: the proposal kernel posts, the host polls
  unsigned int *host_done_dev_ = nullptr;
  unsigned long long host_seq_ = 0;
  int64_t host_chunk_ = 0;                                // particles per chunk; 0 = automatic (host_chunk_size)
  std::vector<int64_t> host_ids_, host_where_;
  std::vector<double> host_thv_, host_rhov_, host_both_, host_lp_;
  double host_cb_seconds_ = 0.0;                          // time spent inside the caller's callbacks
  int64_t host_cb_calls_ = 0;
  unsigned long long *host_acc_dev_ = nullptr;
  int ensure_host_buffers();
  int64_t host_chunk_size(int64_t cnt) const;
  int wait_host_flag(int ch, unsigned long long seq);
  RtcKernels rtc_;                                        // SABC_MODEL_USER: kernels compiled from the user's source
  const RtcKernels *rtc() const { return rtc_.module ? &rtc_ : nullptr; }
  void free_later(void *p);                               // device memory released by end_of_call(), never inside a call
  std::vector<void *> deferred_free_;
  // peer-to-peer transport
  int build_cdf_blocks(const ShardBlocks &rho_blocks, int64_t *len_out, int *any_negative);
  P2PView p2p_view() const;
  int take_silence() {                                    // 0 | 1 skipped | 2 own slots only
    if (p2p_skip_ > 0) { --p2p_skip_; return 0; }
    if (p2p_silent_ > 0) { --p2p_silent_; return p2p_loss_ ? 2 : 1; }
    return 0;
  }
  bool p2p_finish();                                      // destructor: leave, wait for the peers' `released`; false = park
  void flip_cur() { cur_ = 1 - cur_; ++flips_; if (page_) page_->cur_parity.store((uint32_t)cur_, std::memory_order_relaxed); }
  // a peer's CURRENT population: indexed by the owner's parity at set-up + the flips since (p2p.hpp: P2PDesc::cur)
  double *peer_pop_cur(int r) const { return peer_pop_[(peer_cur0_[r] ^ (int)(flips_ & 1u)) & 1][r]; }
  uint32_t tag(uint32_t seq) const { return p2p_tag(gen_, seq); }
  int selftest_patterns(const P2PView &pv);
  bool open_peer_page(int r, const P2PDesc &d);
  uint64_t *slots_ = nullptr;                             // this shard's slot area (fine-grained device memory)
  uint64_t *peer_slots_[kMaxPeers] = {nullptr};
  double *peer_pop_[2][kMaxPeers] = {{nullptr}};
  double *peer_rho_[kMaxPeers] = {nullptr};
  int peer_cur0_[kMaxPeers] = {0};                        // the owner's parity when the descriptors were written
  uint32_t flips_ = 0;                                    // buffer flips of THIS shard since then (in step on all shards)
  std::vector<void *> ipc_opened_;                        // what hipIpcOpenMemHandle returned (closed when this shard leaves)
  P2PHostPage *page_ = nullptr;                           // this shard's host page (POSIX shared memory)
  char page_name_[48] = {0};
  const P2PHostPage *peer_page_[kMaxPeers] = {nullptr};
  bool peer_page_shm_[kMaxPeers] = {false};               // opened by name (to be unmapped), not a pointer of this process
  uint32_t gen_ = 0;                                      // generation of the current (or last) set-up
  bool mapped_ = false;                                   // the peers' memory is mapped
  bool exported_ = false;                                 // a descriptor has left: peers may have mapped this shard's memory
  double destroy_wait_ms_ = -1.0;                         // < 0: the bound of the waits
  int p2p_stale_ = 0;
  bool p2p_on_ = false, pending_xchg_ = false;
  uint32_t xseq_ = 0, bseq_ = 0, call_ = 0;               // exchange / barrier / call sequence numbers (the same on every shard)
  double p2p_timeout_ms_ = 5000.0;
  int wall_clock_khz_ = 100000;                           // s_memrealtime: 100 MHz unless the device says otherwise
  int p2p_silent_ = 0, p2p_skip_ = 0;
  bool p2p_loss_ = false;
  double *p2p_test_dev_ = nullptr;
  double *snap_pop_ = nullptr, *snap_rho_ = nullptr;     // device-side copy of the particles at the entry of a call
  int64_t launches_ = 0;
  int64_t persist_max_ = 65536;                           // shards up to this many particles run their updates in one launch (0: never)
  unsigned long long *persist_sync_ = nullptr;            // the grid barrier's counter and abort flag
  unsigned long long *persist_rows_ = nullptr;            // the workgroups' partial rows as tagged words, two parities (persistent_kernel.hpp)
  int64_t persist_rows_wg_ = 0;                           // workgroups it holds rows for
  int persist_lanes_ = 0;                                 // lanes per particle of the last one-launch update (0: none yet)
  int prof_ = 0, prof_open_ = -1;
  unsigned prof_tick_ = 0;
  struct EvPair { hipEvent_t a, b; };
  std::vector<EvPair> ev_[SABC_KERNEL_COUNT];
  std::vector<EvPair> ev_pool_;
  double prof_ms_[SABC_KERNEL_COUNT] = {0};
  int64_t prof_n_[SABC_KERNEL_COUNT] = {0};
  int64_t prof_noop_[SABC_KERNEL_COUNT] = {0};
};

}  // namespace sabc
