// hip_backend.hip -- HipBackend: owns the shard in HBM and launches the gfx950 kernels.
#include "hip_backend.hpp"

#include <hip/hip_runtime.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace sabc {

namespace {
std::atomic<int64_t> g_parked_bytes{0};       // device memory kept because a peer had not released it when its owner went away
}

#define HB_CHECK(expr, what)                       \
  do {                                             \
    const int rc_ = check((expr), (what));         \
    if (rc_) return rc_;                           \
  } while (0)

// launchers report hipGetLastError(), which is sticky: clear whatever an earlier, unrelated HIP call left
#define HB_LAUNCH(expr, what)                      \
  do {                                             \
    (void)hipGetLastError();                       \
    const int e_ = (expr);                         \
    launches_ += 1;                                \
    if (e_) return check((hipError_t)e_, (what));  \
  } while (0)

static double persist_timeout_ms() {                 // (read per launch: tests change it)
  const char *e = std::getenv("SABC_PERSISTENT_TIMEOUT_MS");
  const double v = e ? std::atof(e) : 0.0;
  return v > 0 ? v : 2000.0;
}

// Streams of closed handles are kept for the next handle on the same device: creating one costs ~0.4 ms, destroying one as much
// -- a tenth of a whole sabc() call at the sizes the reference's documentation works with (17 ms for n_particles = 1000,
// n_simulation = 1e6).  A stream goes back only after it has drained; streams the caller supplied (sabc_set_stream) are never
// pooled.  (The pool is never torn down: at process exit the runtime may already be gone.)
namespace {
std::mutex g_stream_pool_mutex;
std::vector<std::pair<int, hipStream_t>> *g_stream_pool = nullptr;       // (device, idle stream)
constexpr size_t kStreamPoolMax = 32;

hipStream_t pooled_stream_take(int device) {
  std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
  if (!g_stream_pool) return nullptr;
  for (size_t i = 0; i < g_stream_pool->size(); ++i)
    if ((*g_stream_pool)[i].first == device) {
      hipStream_t s = (*g_stream_pool)[i].second;
      g_stream_pool->erase(g_stream_pool->begin() + (long)i);
      return s;
    }
  return nullptr;
}
void pooled_stream_give(int device, hipStream_t s) {
  if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamDestroy(s); return; }
  std::lock_guard<std::mutex> lock(g_stream_pool_mutex);
  if (!g_stream_pool) g_stream_pool = new std::vector<std::pair<int, hipStream_t>>();
  if (g_stream_pool->size() >= kStreamPoolMax) { (void)hipStreamDestroy(s); return; }
  g_stream_pool->emplace_back(device, s);
}
}  // namespace

static double persist_rendezvous_ms() {              // (the wait for every workgroup of a one-launch update to be resident)
  const char *e = std::getenv("SABC_PERSISTENT_RENDEZVOUS_MS");
  const double v = e ? std::atof(e) : 0.0;
  return v > 0 ? v : 20.0;
}

HipBackend::HipBackend(int device) : device_(device) {
  // shards up to this many particles run the population updates of a call in ONE launch (kernels.hip: k_update_persistent);
  // SABC_PERSISTENT=0 (or SABC_PERSISTENT_MAX=0) keeps the launch chain per update at every size
  if (const char *e = std::getenv("SABC_PERSISTENT_MAX")) persist_max_ = std::atoll(e);
  if (const char *e = std::getenv("SABC_PERSISTENT")) { if (e[0] == '0') persist_max_ = 0; }
  // (tests lower the limit to reach the two-launch form -- k_reduce_partials, then the control / exchange launch -- at small n)
  if (const char *e = std::getenv("SABC_FUSE_REDUCE_MAX")) {
    const long long v = std::atoll(e);
    if (v >= 0) fuse_reduce_max_ = v;
  }
}

// hipFree waits for EVERY stream of the process.  With several shards in one process (tests; a Julia host driving the GPUs
// of a node from threads) a peer's kernel may be spinning for this shard's next post, which the host cannot enqueue while
// it sits in hipFree: nothing is freed inside a call.  end_of_call() runs after the call's last exchange.
void HipBackend::free_later(void *p) {
  if (p) deferred_free_.push_back(p);
}

void HipBackend::end_of_call() {
  for (void *p : deferred_free_) (void)hipFree(p);
  deferred_free_.clear();
}

HipBackend::~HipBackend() {
  if (!stream_ && !pop_[0]) return;                // never allocated (e.g. create failed on a bad device ordinal)
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  // the peer-to-peer group first (p2p.hpp "LEAVES"): what peers may have mapped -- both population buffers, rho, the slot
  // area -- is freed only when every one of them has recorded that it unmapped it; otherwise it is parked until the process
  // exits: a late reader meets stale particles, never an unmapped page
  const size_t pop_bytes = (size_t)(m_.d + m_.s + 1) * (size_t)sh_.cap * sizeof(double), rho_bytes = (size_t)m_.s * (size_t)sh_.cap * sizeof(double);
  if (!p2p_finish()) {
    g_parked_bytes += (int64_t)((pop_[0] ? pop_bytes : 0) + (pop_[1] ? pop_bytes : 0) + (rho_ ? rho_bytes : 0) + (slots_ ? (size_t)kP2PSlotWords * 8 : 0));
    pop_[0] = pop_[1] = rho_ = nullptr;
    slots_ = nullptr;
  }
  end_of_call();
  for (auto &v : ev_)
    for (auto &e : v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto &e : ev_pool_) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  double *dev[] = {pop_[0], pop_[1], rho_, knots_, coarse_, mid_, partials_, hist_dev_, sums_stage_, gather_, cum_, block_sums_, totals_dev_, col_a_, col_b_, pack_dev_};
  for (double *p : dev)
    if (p) (void)hipFree(p);
  for (double *p : scratch_)
    if (p) (void)hipFree(p);
  if (idx_dev_) (void)hipFree(idx_dev_);
  if (slot_dev_) (void)hipFree(slot_dev_);
  if (bucket_dev_) (void)hipFree(bucket_dev_);
  if (bucket_host_) (void)hipHostFree(bucket_host_);
  double *staged[] = {host_thp_, host_rho_, host_cur_, host_lp2_};
  for (double *p : staged)
    if (p) (void)hipHostFree(p);
  if (host_gate_) (void)hipHostFree(host_gate_);
  if (dev_thp_) (void)hipFree(dev_thp_);
  if (dev_aux_) (void)hipFree(dev_aux_);
  if (dev_rho_prop_) (void)hipFree(dev_rho_prop_);
  if (host_flag_) (void)hipHostFree(host_flag_);
  if (host_done_dev_) (void)hipFree(host_done_dev_);
  if (host_acc_dev_) (void)hipFree(host_acc_dev_);
  if (sort_tmp_) (void)hipFree(sort_tmp_);
  if (meta_dev_) (void)hipFree(meta_dev_);
  if (cb_dev_) (void)hipFree(cb_dev_);
  if (pinned_block_) (void)hipHostFree(pinned_block_);     // (the control block's staging copy, the mailbox ring, the totals)
  rtc_release(&rtc_);
  if (persist_sync_) (void)hipFree(persist_sync_);
  if (persist_rows_) (void)hipFree(persist_rows_);
  if (slots_) (void)hipFree(slots_);
  if (p2p_test_dev_) (void)hipFree(p2p_test_dev_);
  if (snap_pop_) (void)hipFree(snap_pop_);
  if (snap_rho_) (void)hipFree(snap_rho_);
  if (own_stream_ && stream_) pooled_stream_give(device_, stream_);
}

int HipBackend::check(hipError_t e, const char *what) {
  if (e == hipSuccess) return 0;
  char buf[256];
  std::snprintf(buf, sizeof(buf), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  err_ = buf;
  return -1;
}

int HipBackend::set_stream(hipStream_t s) {
  HB_CHECK(hipSetDevice(device_), "hipSetDevice");
  if (stream_) HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  if (own_stream_ && stream_) pooled_stream_give(device_, stream_);
  stream_ = s;
  own_stream_ = false;
  return 0;
}

PopPtrs HipBackend::pop_ptrs(int which) const {
  PopPtrs pp;
  pp.pop = pop_[which];
  pp.rho = rho_;
  pp.cap = sh_.cap;
  pp.n_local = sh_.n_local;
  pp.gid0 = sh_.gid0;
  return pp;
}

CdfPtrs HipBackend::cdf_ptrs() const {
  CdfPtrs c;
  c.knots = knots_;
  c.stride = knot_stride_;
  c.coarse = coarse_;
  c.mid = mid_;
  c.mid_stride = mid_stride_;
  for (int j = 0; j < kMaxStats; ++j) { c.len[j] = cdf_len_[j]; c.shift[j] = cdf_shift_[j]; }
  return c;
}

int HipBackend::allocate(const ModelDesc &m, const Shard &sh) {
  m_ = m;
  sh_ = sh;
  np_ = n_partials(m.d, m.s);
  HB_CHECK(hipSetDevice(device_), "hipSetDevice");
  if (!stream_) {
    stream_ = pooled_stream_take(device_);
    if (!stream_) HB_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    own_stream_ = true;
  }
  const size_t cap = (size_t)sh.cap, N = (size_t)sh.n_global;
  const size_t rows = (size_t)(m.d + m.s + 1);
  // What peers read over the peer-to-peer transport (p2p.hpp): both population buffers and rho.  On a sharded handle they
  // live in FINE-GRAINED device memory -- coherent across devices at every access, so a peer's read never depends on what a
  // kernel boundary does to its caches.  On one GPU that costs nothing measurable (two processes sharing an MI355X, n = 1e6:
  // 5.48 / 4.99e9 sims/s RandomWalk / DE with plain hipMalloc, 5.49 / 5.03e9 fine-grained); what it
  // costs a REMOTE reader is unmeasured, its reads (partners, drawn rows: random; the weight rows: once) have little reuse to
  // lose.  SABC_P2P_FINEGRAINED=0 goes back to plain device memory, visible across devices at kernel boundaries only;
  // either way sabc_comm_p2p_selftest checks on first contact that peers read what the owners' kernels wrote.
  static const bool fine = [] { const char *e = std::getenv("SABC_P2P_FINEGRAINED"); return !(e && e[0] == '0'); }();
  auto pop_alloc = [&](double **p, size_t bytes) {
    return fine && sh.world > 1 ? hipExtMallocWithFlags((void **)p, bytes, hipDeviceMallocFinegrained) : hipMalloc((void **)p, bytes);
  };
  for (int b = 0; b < 2; ++b) {
    HB_CHECK(pop_alloc(&pop_[b], rows * cap * sizeof(double)), "hipMalloc(pop)");
    HB_CHECK(hipMemsetAsync(pop_[b], 0, rows * cap * sizeof(double), stream_), "hipMemset(pop)");
  }
  HB_CHECK(pop_alloc(&rho_, (size_t)m.s * cap * sizeof(double)), "hipMalloc(rho)");
  HB_CHECK(hipMemsetAsync(rho_, 0, (size_t)m.s * cap * sizeof(double), stream_), "hipMemset(rho)");
  HB_CHECK(hipMalloc((void **)&coarse_, (size_t)m.s * cdf_coarse_entries(m.s) * sizeof(double)), "hipMalloc(coarse)");
  knot_stride_ = (((int64_t)N + 2 + 15) / 16) * 16;       // every table starts on a 128-byte line
  HB_CHECK(hipMalloc((void **)&knots_, (size_t)m.s * (size_t)knot_stride_ * sizeof(double)), "hipMalloc(knots)");
  mid_stride_ = cdf_mid_stride(knot_stride_);
  HB_CHECK(hipMalloc((void **)&mid_, (size_t)m.s * (size_t)mid_stride_ * sizeof(double)), "hipMalloc(mid)");
  {   // k_update writes one row per workgroup; its granularity depends on the model's kernel
    const int64_t per_half = update_rows(m, (sh.cap + 1) / 2) + 1, whole = update_rows(m, sh.cap);
    partial_rows_ = 2 * per_half > whole ? 2 * per_half : whole;
    if (partial_rows_ < n_blocks(sh.cap)) partial_rows_ = n_blocks(sh.cap);
    // (k_update_persistent double-buffers one row per workgroup by the update's parity)
    if (2 * persistent_workgroups_bound(m, sh.cap) > partial_rows_) partial_rows_ = 2 * persistent_workgroups_bound(m, sh.cap);
    partial_rows_ += 4;
  }
  HB_CHECK(hipMalloc((void **)&partials_, (size_t)partial_rows_ * np_ * sizeof(double)), "hipMalloc(partials)");
  HB_CHECK(hipMalloc((void **)&cb_dev_, sizeof(ControlBlock)), "hipMalloc(control block)");
  HB_CHECK(hipMemsetAsync(cb_dev_, 0, sizeof(ControlBlock), stream_), "hipMemset(control block)");
  {   // ONE pinned block for the control block's staging copy, the mailbox ring and the totals (hipHostFree is 0.2 ms apiece)
    const size_t off_mbox = (sizeof(ControlBlock) + 255) / 256 * 256, off_totals = off_mbox + (kMailboxRing * sizeof(Mailbox) + 255) / 256 * 256;
    HB_CHECK(hipHostMalloc((void **)&pinned_block_, off_totals + 256, hipHostMallocMapped), "hipHostMalloc(control block, mailbox, totals)");
    char *pinned_dev = nullptr;
    HB_CHECK(hipHostGetDevicePointer((void **)&pinned_dev, pinned_block_, 0), "hipHostGetDevicePointer(pinned block)");
    cb_host_ = reinterpret_cast<ControlBlock *>(pinned_block_);
    mbox_host_ = reinterpret_cast<Mailbox *>(pinned_block_ + off_mbox);
    mbox_dev_ = reinterpret_cast<Mailbox *>(pinned_dev + off_mbox);
    totals_host_ = reinterpret_cast<double *>(pinned_block_ + off_totals);
    totals_host_dev_ = reinterpret_cast<double *>(pinned_dev + off_totals);
  }
  HB_CHECK(hipMalloc((void **)&sums_stage_, kMaxPartials * sizeof(double)), "hipMalloc(sums staging)");
  HB_CHECK(hipMemsetAsync(sums_stage_, 0, kMaxPartials * sizeof(double), stream_), "hipMemset(sums staging)");
  for (int i = 0; i < kMailboxRing; ++i) { mbox_host_[i].w0 = kMailboxEmpty; mbox_host_[i].w1 = kMailboxEmpty; }
  HB_CHECK(hipMalloc((void **)&cum_, N * sizeof(double)), "hipMalloc(cum)");
  HB_CHECK(hipMalloc((void **)&block_sums_, (size_t)weight_scan_doubles((int64_t)N) * sizeof(double)), "hipMalloc(block_sums)");
  HB_CHECK(hipMalloc((void **)&totals_dev_, 2 * sizeof(double)), "hipMalloc(totals)");
  totals_host_[0] = totals_host_[1] = 0.0;
  HB_CHECK(hipMalloc((void **)&meta_dev_, 2 * kMaxStats * sizeof(int64_t)), "hipMalloc(meta)");
  int khz = 0;                                          // rate of the constant wall clock every bounded wait counts in
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_) == hipSuccess && khz > 0) wall_clock_khz_ = khz;
  else (void)hipGetLastError();
  return 0;
}

double *HipBackend::gather_buffer(int64_t doubles) {
  if (doubles > gather_cap_) {
    if (stream_) (void)hipStreamSynchronize(stream_);
    free_later(gather_);
    gather_ = nullptr;
    gather_cap_ = 0;
    if (hipMalloc((void **)&gather_, (size_t)doubles * sizeof(double)) != hipSuccess) return nullptr;
    gather_cap_ = doubles;
  }
  return gather_;
}

double *HipBackend::scratch_buffer(int which, int64_t doubles) {
  if (which < 0 || which >= 4) return nullptr;
  if (doubles > scratch_cap_[which]) {
    if (stream_) (void)hipStreamSynchronize(stream_);
    free_later(scratch_[which]);
    scratch_[which] = nullptr;
    scratch_cap_[which] = 0;
    const int64_t want = doubles + doubles / 4 + 64;       // head room: the request count of a resample varies from one to the next
    if (hipMalloc((void **)&scratch_[which], (size_t)want * sizeof(double)) != hipSuccess) return nullptr;
    scratch_cap_[which] = want;
  }
  return scratch_[which];
}

int HipBackend::copy_rows(const double *src, int64_t src_pitch, double *dst, int64_t dst_pitch, int rows, int64_t count) {
  if (rows <= 0 || count <= 0) return 0;
  HB_CHECK(hipMemcpy2DAsync(dst, (size_t)dst_pitch * sizeof(double), src, (size_t)src_pitch * sizeof(double),
                            (size_t)count * sizeof(double), (size_t)rows, hipMemcpyDeviceToDevice, stream_), "copy_rows");
  return 0;
}

int HipBackend::to_backend(double *dst, const double *src_host, int64_t n) {
  if (n <= 0) return 0;
  HB_CHECK(hipMemcpyAsync(dst, src_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream_), "to_backend");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  return 0;
}

int HipBackend::to_host(double *dst_host, const double *src, int64_t n) {
  if (n <= 0) return 0;
  HB_CHECK(hipMemcpyAsync(dst_host, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream_), "to_host");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  return 0;
}

double *HipBackend::host_stage(int64_t doubles) {
  if ((int64_t)stage_.size() < doubles) stage_.resize((size_t)doubles);
  return stage_.data();
}

// level 1: bracket only the dominant kernel (k_update) -- every hipEventRecord is a marker packet the
// queue has to drain, ~4 us of GPU time each, so the other kernels are bracketed only at level 2
void HipBackend::profile_enable(int level) {
  prof_ = level;
  prof_tick_ = 0;
  if (level)
    for (int k = 0; k < SABC_KERNEL_COUNT; ++k) { prof_ms_[k] = 0.0; prof_n_[k] = 0; prof_noop_[k] = 0; }
  while (level && ev_pool_.size() < 256) {       // created outside the timed region
    EvPair e;
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) break;
    ev_pool_.push_back(e);
  }
}

void HipBackend::prof_begin(int kernel) {
  if (!prof_ || (prof_ < 2 && kernel != SABC_KERNEL_UPDATE)) return;
  EvPair e;
  if (!ev_pool_.empty()) { e = ev_pool_.back(); ev_pool_.pop_back(); }
  else if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
  (void)hipEventRecord(e.a, stream_);
  ev_[kernel].push_back(e);
  prof_open_ = kernel;
}

void HipBackend::prof_end(int kernel) {
  if (prof_open_ != kernel || ev_[kernel].empty()) return;
  (void)hipEventRecord(ev_[kernel].back().b, stream_);
  prof_open_ = -1;
}

int HipBackend::profile_get(int kernel, double *total_ms, int64_t *launches) {
  if (kernel < 0 || kernel >= SABC_KERNEL_COUNT) return -1;
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  for (auto &e : ev_[kernel]) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      // an update launch queued ahead of a resample test that fired returns at `if (cb->halt)`: not a sample of the kernel
      if (kernel == SABC_KERNEL_UPDATE && ms < 0.004f && sh_.n_local >= 4096) prof_noop_[kernel] += 1;
      else { prof_ms_[kernel] += ms; prof_n_[kernel] += 1; }
    }
    ev_pool_.push_back(e);
  }
  ev_[kernel].clear();
  if (total_ms) *total_ms = prof_ms_[kernel];
  if (launches) *launches = prof_n_[kernel];
  return 0;
}

// ---- host mode: f_dist (SABC_MODEL_HOST) and / or the prior (prior_joint = 2) are host callbacks ----
// A prior that lives in host callbacks next to a DEVICE-coded simulator (any Distribution of the reference next to a built-in
// or source-compiled f_dist) takes the same cut -- k_host_propose -> logpdf(prior, .) on the host -> the simulator as its own
// launch over the gated proposals (k_simulate_batch, the fused kernel's streams) -> k_host_accept -- with one chunk per half
// batch: the callback is the log density alone, there is no host simulation to overlap.
// f_dist is the caller's function (SimulatedAnnealingABC.jl:315), so every half batch is cut at the host:
//   k_host_propose (device) -> f_dist on the proposals inside the prior's support (host) -> k_host_accept (device).
// What the library adds around the callback is kept off the critical path:
//  * staging arrays are PINNED host memory MAPPED into the device, allocated once: the kernels write proposals and read
//    distances in place -- no hipMemcpy call, no pageable staging, no allocation per half batch; and only what the host
//    needs crosses PCIe: the proposals and ONE BYTE of prior gate go down, the distances come up; the proposals' second copy
//    and the log densities stay in device memory for the accept step;
//  * the propose kernel signals completion CHUNK by chunk into a pinned flag word the host polls (no stream sync): the
//    callback for chunk c runs while the accept kernel of chunk c - 1 executes and later chunks are still being proposed;
//  * nothing waits at the end of a half batch: the next kernel on the stream is ordered behind the accept kernels.
int HipBackend::ensure_host_buffers() {
  if (host_thp_) return 0;
  const size_t cap = (size_t)(sh_.cap > 0 ? sh_.cap : 1);
  auto mapped = [&](double **host, double **dev, size_t doubles) -> int {
    HB_CHECK(hipHostMalloc((void **)host, doubles * sizeof(double), hipHostMallocMapped), "hipHostMalloc(host-mode staging)");
    HB_CHECK(hipHostGetDevicePointer((void **)dev, *host, 0), "hipHostGetDevicePointer(host-mode staging)");
    return 0;
  };
  if (mapped(&host_thp_, &host_thp_dev_, (size_t)m_.d * cap)) return -1;
  if (mapped(&host_rho_, &host_rho_dev_, (size_t)m_.s * cap)) return -1;
  HB_CHECK(hipHostMalloc((void **)&host_gate_, cap, hipHostMallocMapped), "hipHostMalloc(prior gate)");
  HB_CHECK(hipHostGetDevicePointer((void **)&host_gate_dev_, host_gate_, 0), "hipHostGetDevicePointer(prior gate)");
  if (m_.prior_joint == 2) {
    if (mapped(&host_cur_, &host_cur_dev_, (size_t)m_.d * cap)) return -1;
    if (mapped(&host_lp2_, &host_lp2_dev_, 2 * cap)) return -1;
  }
  // what only the device reads again: the proposals and (log prior, log factor) of the half batch in flight
  HB_CHECK(hipMalloc((void **)&dev_thp_, (size_t)m_.d * cap * sizeof(double)), "hipMalloc(proposals)");
  HB_CHECK(hipMalloc((void **)&dev_aux_, 2 * cap * sizeof(double)), "hipMalloc(log prior, log factor)");
  if (m_.model_id != SABC_MODEL_HOST)                    // a device-coded simulator next to a host prior: its distances stay on the device
    HB_CHECK(hipMalloc((void **)&dev_rho_prop_, (size_t)m_.s * cap * sizeof(double)), "hipMalloc(proposals' distances)");
  HB_CHECK(hipHostMalloc((void **)&host_flag_, kHostMaxChunks * sizeof(unsigned long long), hipHostMallocMapped), "hipHostMalloc(chunk flags)");
  for (int i = 0; i < kHostMaxChunks; ++i) host_flag_[i] = 0ull;
  HB_CHECK(hipHostGetDevicePointer((void **)&host_flag_dev_, host_flag_, 0), "hipHostGetDevicePointer(chunk flags)");
  HB_CHECK(hipMalloc((void **)&host_done_dev_, kHostMaxChunks * sizeof(unsigned int)), "hipMalloc(chunk counters)");
  HB_CHECK(hipMemsetAsync(host_done_dev_, 0, kHostMaxChunks * sizeof(unsigned int), stream_), "hipMemset");
  HB_CHECK(hipMalloc((void **)&host_acc_dev_, sizeof(unsigned long long)), "hipMalloc(host accept counter)");
  HB_CHECK(hipMemsetAsync(host_acc_dev_, 0, sizeof(unsigned long long), stream_), "hipMemset");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  host_ids_.reserve(cap); host_where_.reserve(cap);
  return 0;
}

// particles per chunk of a half batch of cnt: whole workgroups, at most kHostMaxChunks chunks, and not so small that the
// fixed cost of one callback (a ctypes / ccall transition, ~10-50 us from Python) shows: >= 4096 unless asked otherwise
int64_t HipBackend::host_chunk_size(int64_t cnt) const {
  int64_t chunk = host_chunk_;
  if (chunk <= 0) {
    int64_t pieces = cnt / 4096;                         // automatic: equal pieces of >= 4096, at most 8
    pieces = pieces < 1 ? 1 : (pieces > 8 ? 8 : pieces);
    chunk = (cnt + pieces - 1) / pieces;
  }
  const int64_t least = (cnt + kHostMaxChunks - 1) / kHostMaxChunks;
  if (chunk < least) chunk = least;
  chunk = ((chunk + kBlock - 1) / kBlock) * kBlock;
  return chunk;
}

int HipBackend::wait_host_flag(int ch, unsigned long long seq) {
  volatile unsigned long long *f = host_flag_ + ch;
  for (uint64_t spins = 1; *f != seq; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 0x3FFF) == 0) {
      const hipError_t q = hipStreamQuery(stream_);
      if (q == hipSuccess) {
        if (*f == seq) break;
        err_ = "the proposal kernel did not signal a chunk although the stream is idle";
        return -1;
      }
      if (q != hipErrorNotReady) return check(q, "hipStreamQuery");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return 0;
}

int HipBackend::host_prior_simulate() {
  const bool device_sim = m_.model_id != SABC_MODEL_HOST;
  if (!device_sim && !host_fn_) { err_ = "no host simulator set (sabc_set_host_simulator)"; return -1; }
  if (ensure_host_buffers()) return -1;
  const int d = m_.d, s = m_.s;
  const int64_t n = sh_.n_local;
  // the staging arrays double as theta [d][n] / rho [s][n] here (one-time, synchronous: n simulations on the host follow)
  double *th = host_thp_, *rho = host_rho_;
  host_ids_.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) host_ids_[(size_t)i] = sh_.gid0 + i;
  const size_t w = (size_t)n * sizeof(double), pitch = (size_t)sh_.cap * sizeof(double);
  if (m_.prior_joint == 2) {                            // rand(prior) on the host (:174), theta uploaded
    if (!prior_sample_fn_) { err_ = "no host prior set (sabc_set_host_prior)"; return -1; }
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = n > 0 ? prior_sample_fn_(prior_ctx_, n, host_ids_.data(), th) : 0;
    host_cb_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc) { err_ = "the host prior's sample callback failed"; return -1; }
    if (n > 0) HB_CHECK(hipMemcpy2DAsync(pop_[cur_], pitch, th, w, w, (size_t)d, hipMemcpyHostToDevice, stream_), "upload theta");
    if (device_sim) {
      // f_dist on the device (:175), the streams of the fused initialisation kernel (particle id, iteration 0); the pinned
      // staging array is mapped into the device: the simulator reads theta [d][n] straight from it
      if (n > 0) {
        HB_LAUNCH(launch_simulate_batch(m_, host_thp_dev_, n, (uint64_t)sh_.gid0, 0, dev_rho_prop_, stream_, rtc()), "k_simulate_batch");
        HB_CHECK(hipMemcpy2DAsync(rho_, pitch, dev_rho_prop_, w, w, (size_t)s, hipMemcpyDeviceToDevice, stream_), "rho");
      }
      HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
      return 0;
    }
  } else {
    HB_LAUNCH(launch_host_prior(m_, pop_ptrs(cur_), stream_), "k_host_prior");
    if (n > 0) HB_CHECK(hipMemcpy2DAsync(th, w, pop_[cur_], pitch, w, (size_t)d, hipMemcpyDeviceToHost, stream_), "download theta");
  }
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  for (int64_t i = 0; i < (int64_t)s * n; ++i) rho[i] = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = n > 0 ? host_fn_(host_ctx_, th, host_ids_.data(), n, 0, rho) : 0;
  host_cb_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  host_cb_calls_ += 1;
  if (rc) { err_ = "the host simulator (f_dist) failed"; return -1; }
  if (n > 0) HB_CHECK(hipMemcpy2DAsync(rho_, pitch, rho, w, w, (size_t)s, hipMemcpyHostToDevice, stream_), "upload rho");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  return 0;
}

int HipBackend::host_update_range(const StepArgs &c, const PartnerView &pv, int64_t lo, int64_t cnt) {
  const bool device_sim = m_.model_id != SABC_MODEL_HOST;
  if (!device_sim && !host_fn_) { err_ = "no host simulator set (sabc_set_host_simulator)"; return -1; }
  if (lo < 0 || cnt < 0 || lo + cnt > sh_.n_local) { err_ = "host_update_range: range outside the shard"; return -1; }
  if (cnt == 0) return 0;
  if (ensure_host_buffers()) return -1;
  const int d = m_.d, s = m_.s;
  const bool host_prior = m_.prior_joint == 2;
  if (host_prior && !prior_logpdf_fn_) { err_ = "no host prior set (sabc_set_host_prior)"; return -1; }
  const int64_t chunk = device_sim ? ((cnt + kBlock - 1) / kBlock) * kBlock : host_chunk_size(cnt);
  const int n_chunks = (int)((cnt + chunk - 1) / chunk);
  const unsigned long long seq = ++host_seq_;
  // ONE launch proposes the whole half batch (:311-314); it signals its chunks as they complete
  HB_LAUNCH(launch_host_propose(m_, c, cb_dev_, pop_ptrs(cur_), pv, lo, cnt, dev_thp_, dev_aux_, host_thp_dev_, host_gate_dev_,
                                host_prior ? host_cur_dev_ : nullptr, host_done_dev_, host_flag_dev_, seq, chunk, stream_),
            "k_host_propose");
  for (int ch = 0; ch < n_chunks; ++ch) {
    const int64_t t0 = (int64_t)ch * chunk, tn = (t0 + chunk < cnt ? t0 + chunk : cnt) - t0;
    if (wait_host_flag(ch, seq)) return -1;
    if (host_prior) {
      // logpdf(prior, .) on the host (:314, :318): one call for the chunk's proposals followed by its current particles
      host_both_.resize((size_t)(2 * tn * d));
      host_lp_.assign((size_t)(2 * tn), -INFINITY);
      for (int k = 0; k < d; ++k)
        for (int64_t t = 0; t < tn; ++t) {
          host_both_[(size_t)(k * 2 * tn + t)] = host_thp_[(size_t)(k * cnt + t0 + t)];
          host_both_[(size_t)(k * 2 * tn + tn + t)] = host_cur_[(size_t)(k * cnt + t0 + t)];
        }
      const auto c0 = std::chrono::steady_clock::now();
      const int rc = prior_logpdf_fn_(prior_ctx_, 2 * tn, host_both_.data(), host_lp_.data());
      host_cb_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
      if (rc) { err_ = "the host prior's logpdf callback failed"; return -1; }
      for (int64_t t = 0; t < tn; ++t) {
        const double l = host_lp_[(size_t)t];
        host_lp2_[(size_t)(t0 + t)] = l == l ? l : -INFINITY;             // NaN: outside the support
        host_lp2_[(size_t)(cnt + t0 + t)] = host_lp_[(size_t)(tn + t)];
        host_gate_[(size_t)(t0 + t)] = host_lp2_[(size_t)(t0 + t)] > -INFINITY ? 1 : 0;
      }
    }
    if (device_sim) {
      // the simulator on the device, over the proposals the host's gate bytes let through (mapped memory: read in place)
      prof_begin(SABC_KERNEL_UPDATE);
      HB_LAUNCH(launch_simulate_batch(m_, dev_thp_, cnt, (uint64_t)(sh_.gid0 + lo), c.iter, dev_rho_prop_, stream_, rtc(), host_gate_dev_),
                "k_simulate_batch");
      HB_LAUNCH(launch_host_accept(m_, c, cb_dev_, pop_ptrs(cur_), cdf_ptrs(), lo, cnt, 0, cnt, dev_thp_, dev_aux_, dev_rho_prop_,
                                   host_lp2_dev_, host_acc_dev_, stream_), "k_host_accept");
      prof_end(SABC_KERNEL_UPDATE);
      return 0;      // (nothing to wait for: the next half batch's proposal kernel is ordered behind these on the stream)
    }
    // only proposals inside the prior's support are simulated (:314-315): compact them for the callback
    host_ids_.resize((size_t)tn); host_where_.resize((size_t)tn);
    int64_t mv = 0;
    {
      const unsigned char *gate = host_gate_ + t0;
      int64_t *ids = host_ids_.data(), *where = host_where_.data();
      const int64_t gid_first = sh_.gid0 + lo + t0;
      for (int64_t t = 0; t < tn; ++t)
        if (gate[t]) { ids[mv] = gid_first + t; where[mv] = t0 + t; ++mv; }
    }
    // every proposal of the chunk passed and the chunk's rows are contiguous (one parameter / statistic, or the chunk is the
    // whole half batch): f_dist reads the proposals and writes the distances IN the staging arrays, nothing is copied
    const bool direct = mv == tn && (d == 1 || tn == cnt) && (s == 1 || tn == cnt);
    const double *th_arg = host_thp_ + t0;
    double *rho_arg = host_rho_ + t0;
    if (!direct) {
      host_thv_.resize((size_t)(d * mv)); host_rhov_.assign((size_t)(s * mv), 0.0);
      for (int k = 0; k < d; ++k) {
        const double *src = host_thp_ + (size_t)k * cnt;
        double *dst = host_thv_.data() + (size_t)k * mv;
        for (int64_t i = 0; i < mv; ++i) dst[i] = src[host_where_[(size_t)i]];
      }
      th_arg = host_thv_.data(); rho_arg = host_rhov_.data();
    }
    if (mv > 0) {
      const auto c0 = std::chrono::steady_clock::now();
      const int rc = host_fn_(host_ctx_, th_arg, host_ids_.data(), mv, c.iter, rho_arg);
      host_cb_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
      host_cb_calls_ += 1;
      if (rc) { err_ = "the host simulator (f_dist) failed"; return -1; }
    }
    if (!direct)
      for (int j = 0; j < s; ++j) {
        double *dst = host_rho_ + (size_t)j * cnt;
        for (int64_t t = 0; t < tn; ++t) dst[t0 + t] = 0.0;
        const double *src = host_rhov_.data() + (size_t)j * mv;
        for (int64_t i = 0; i < mv; ++i) dst[host_where_[(size_t)i]] = src[i];
      }
    // the accept step of this chunk (:316-329) reads the distances in place; it runs while the host is in the next
    // chunk's callback.  (The launch orders the host's stores above before the kernel's loads.)
    if (ch == 0) prof_begin(SABC_KERNEL_UPDATE);
    HB_LAUNCH(launch_host_accept(m_, c, cb_dev_, pop_ptrs(cur_), cdf_ptrs(), lo, cnt, t0, tn, dev_thp_, dev_aux_, host_rho_dev_,
                                 host_prior ? host_lp2_dev_ : nullptr, host_acc_dev_, stream_), "k_host_accept");
    if (ch == n_chunks - 1) prof_end(SABC_KERNEL_UPDATE);
  }
  return 0;
}

int HipBackend::host_stats(int64_t *rows_out) {
  if (ensure_host_buffers()) return -1;
  HB_LAUNCH(launch_stats_rt(m_, cb_dev_, pop_ptrs(cur_), partials_, host_acc_dev_, stream_), "k_stats_rt");
  *rows_out = n_blocks(sh_.n_local);
  return 0;
}

int HipBackend::register_device_simulator(const char *hip_source) {
  if (m_.model_id != SABC_MODEL_USER) { err_ = "the handle was not created with SABC_MODEL_USER"; return -1; }
  HB_CHECK(hipSetDevice(device_), "hipSetDevice");
  if (stream_) HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  rtc_release(&rtc_);
  std::string log;
  // (the one-launch form of small shards -- k_update_persistent -- is compiled only for handles that can take it)
  const bool small = persist_max_ > 0 && sh_.world == 1 && sh_.n_local <= persist_max_ && persistent_fits(m_.d, m_.s);
  if (rtc_build(hip_source, m_.d, m_.s, rtc_default_csrc_dir(), &rtc_, &log, m_.prior_joint == 3, small)) {
    // the one-launch kernels are the largest of the unit (three team widths x three proposals): should the compiler give up on
    // them for this simulator, the launch chain alone still runs it -- only a source that fails there too is an error
    std::string log_chain;
    if (!small || rtc_build(hip_source, m_.d, m_.s, rtc_default_csrc_dir(), &rtc_, &log_chain, m_.prior_joint == 3, false)) {
      err_ = "compiling the device simulator failed:\n" + log;
      return -1;
    }
  }
  return 0;
}

int HipBackend::prior_simulate() {
  if (m_.model_id == SABC_MODEL_USER && !rtc()) { err_ = "no device simulator registered (sabc_register_device_simulator)"; return -1; }
  prof_begin(SABC_KERNEL_INIT);
  HB_LAUNCH(launch_prior_simulate(m_, pop_ptrs(cur_), stream_, rtc()), "k_prior_simulate");
  prof_end(SABC_KERNEL_INIT);
  return 0;
}

int HipBackend::build_cdf(const double *gathered_rho, int64_t *len_out, int *any_negative) {
  return build_cdf_blocks(flat_blocks(gathered_rho, m_.s, sh_.cap, sh_.world), len_out, any_negative);
}

int HipBackend::build_cdf_blocks(const ShardBlocks &rho_blocks, int64_t *len_out, int *any_negative) {
  const int64_t N = sh_.n_global;
  if (!col_a_) {
    HB_CHECK(hipMalloc((void **)&col_a_, (size_t)N * sizeof(double)), "hipMalloc(col_a)");
    HB_CHECK(hipMalloc((void **)&col_b_, (size_t)N * sizeof(double)), "hipMalloc(col_b)");
    size_t bytes = 0;
    HB_LAUNCH(sort_f64(col_a_, col_b_, N, nullptr, &bytes, stream_), "radix sort size query");
    sort_tmp_bytes_ = bytes ? bytes : 16;
    HB_CHECK(hipMalloc(&sort_tmp_, sort_tmp_bytes_), "hipMalloc(sort_tmp)");
  }
  for (int j = 0; j < m_.s; ++j) {
    HB_LAUNCH(launch_compact_column(rho_blocks, j, N, col_a_, stream_), "k_compact_column");
    size_t bytes = sort_tmp_bytes_;
    HB_LAUNCH(sort_f64(col_a_, col_b_, N, sort_tmp_, &bytes, stream_), "radix sort");
    launches_ += 31;                                   // 8 passes of 4 kernels
    HB_LAUNCH(launch_cdf_knots(col_b_, N, knots_ + (int64_t)j * knot_stride_, meta_dev_ + 2 * j, stream_), "k_cdf_knots");
    launches_ += 1;
  }
  int64_t meta[2 * kMaxStats];
  HB_CHECK(hipMemcpyAsync(meta, meta_dev_, 2 * (size_t)m_.s * sizeof(int64_t), hipMemcpyDeviceToHost, stream_), "memcpy(meta)");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  *any_negative = 0;
  for (int j = 0; j < m_.s; ++j) {
    const int64_t mpos = N - meta[2 * j];
    cdf_len_[j] = mpos > 0 ? mpos + 2 : 0;
    len_out[j] = cdf_len_[j];
    if (meta[2 * j + 1]) *any_negative = 1;
    if (cdf_len_[j] > 0 && build_coarse(j)) return -1;
  }
  // the sort scratch is only needed once per result
  free_later(col_a_); free_later(col_b_); free_later(sort_tmp_);
  col_a_ = col_b_ = nullptr; sort_tmp_ = nullptr;
  return 0;
}

int HipBackend::cdf_population() {
  HB_LAUNCH(launch_cdf_population(m_, pop_ptrs(cur_), cdf_ptrs(), stream_), "k_cdf_population");
  return 0;
}

int HipBackend::update_range(const StepArgs &c, const PartnerView &pv, int64_t lo, int64_t cnt, int64_t row0,
                             int64_t *rows_out) {
  const int64_t rows = update_rows(m_, cnt);
  if (lo < 0 || cnt < 0 || lo + cnt > sh_.n_local || row0 + rows > partial_rows_) {
    err_ = "update_range: range outside the shard";
    return -1;
  }
  // the timing events ride on the kernel's own dispatch packet (no marker packets around it)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // level 1 times every SECOND launch of the update kernel (a timed dispatch packet costs ~4 us of queue time: sampling
  // halves what the measurement adds to the step); levels 2 and 3 time every launch
  if (prof_ && (prof_ != 1 || (prof_tick_++ & 1) == 0)) {
    EvPair e{nullptr, nullptr};
    if (!ev_pool_.empty()) { e = ev_pool_.back(); ev_pool_.pop_back(); }
    else if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) e = EvPair{nullptr, nullptr};
    if (e.a && e.b) { ev_[SABC_KERNEL_UPDATE].push_back(e); ev0 = e.a; ev1 = e.b; }
  }
  HB_LAUNCH(launch_update(m_, c, cb_dev_, pop_ptrs(cur_), cdf_ptrs(), pv, lo, cnt, partials_, row0, stream_, ev0, ev1, rtc()), "k_update");
  *rows_out = rows;
  return 0;
}

// Small shards: the updates of a call in one launch (persistent_kernel.hpp: k_update_persistent; for a simulator from source
// the run-time compiled instantiation).  Not under a profile level that wants every kernel bracketed (the launch chain is
// what such a run measures).
bool HipBackend::persistent_supported(int prop_kind) const {
  if (persist_max_ <= 0 || sh_.world != 1 || sh_.n_local > persist_max_ || prof_ >= 2) return false;
  const int64_t wg = persistent_workgroups(m_, prop_kind, sh_.n_local, rtc());
  return wg > 0 && 2 * wg <= partial_rows_;
}

int HipBackend::update_persistent(const StepArgs &c, const ControlArgs &ctrl, const PartnerView &pv_a, const PartnerView &pv_b, int64_t ix0,
                                  int64_t phase, int64_t cph, int64_t count, int64_t *done, int *halted, int *error) {
  if (pending_rows_ >= 0 && flush_reduce()) return -1;
  if (!persist_sync_) HB_CHECK(hipMalloc((void **)&persist_sync_, 4 * sizeof(unsigned long long)), "hipMalloc(grid barrier)");
  HB_CHECK(hipMemsetAsync(persist_sync_, 0, 4 * sizeof(unsigned long long), stream_), "hipMemset(grid barrier)");
  PersistArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.iter0 = c.iter;
  pa.ix0 = ix0; pa.phase = phase; pa.cph = cph;
  pa.act_n = sh_.n_local; pa.half = sh_.n_local / 2;
  pa.count = (int32_t)(count > (int64_t)1 << 30 ? (int64_t)1 << 30 : count);
  pa.prop_p0 = c.prop_p0; pa.prop_p1 = c.prop_p1;
  pa.ctrl = ctrl;
  pa.sync = persist_sync_;
  pa.timeout_ticks = (uint64_t)(persist_timeout_ms() * (double)wall_clock_khz_);
  pa.rendezvous_ticks = (uint64_t)(persist_rendezvous_ms() * (double)wall_clock_khz_);
  if (const char *e = std::getenv("SABC_PERSISTENT_TEST_ABSENT_WG")) pa.test_absent_wg = std::atoi(e);     // (tests/test_persistent.py)
  const int64_t wg = persistent_workgroups(m_, c.prop_kind, pa.act_n, rtc(), &persist_lanes_);
  if (wg <= 0) return check(hipErrorInvalidValue, "k_update_persistent: no one-launch form for this shard");
  // the rows travel as tagged words (two per value, two parities), zeroed before every launch: its tags start at 1
  const size_t row_bytes = (size_t)np_ * 2 * sizeof(unsigned long long);
  if (!persist_rows_ || persist_rows_wg_ < wg) {
    if (persist_rows_) (void)hipFree(persist_rows_);
    persist_rows_ = nullptr;
    persist_rows_wg_ = persistent_workgroups_bound(m_, sh_.cap) > wg ? persistent_workgroups_bound(m_, sh_.cap) : wg;
    HB_CHECK(hipMalloc((void **)&persist_rows_, 2 * (size_t)persist_rows_wg_ * row_bytes), "hipMalloc(partial rows of the one-launch form)");
  }
  HB_CHECK(hipMemsetAsync(persist_rows_, 0, 2 * (size_t)wg * row_bytes, stream_), "hipMemset(partial rows of the one-launch form)");
  prof_begin(SABC_KERNEL_UPDATE);
  HB_LAUNCH(launch_update_persistent(m_, c.prop_kind, pa, cb_dev_, pop_ptrs(cur_), cdf_ptrs(), pv_a, pv_b, reinterpret_cast<double *>(persist_rows_),
                                     hist_dev_, mbox_dev_, sums_stage_, stream_, rtc()), "k_update_persistent");
  prof_end(SABC_KERNEL_UPDATE);
  ControlBlock cb;
  if (read_control(&cb)) return -1;
  *done = cb.persist_done;
  *halted = cb.halt;
  *error = cb.error;
  return 0;
}

int HipBackend::stats(int64_t *rows_out) {
  HB_LAUNCH(launch_stats(m_, cb_dev_, pop_ptrs(cur_), partials_, stream_, rtc()), "k_stats");
  *rows_out = n_blocks(sh_.n_local);
  return 0;
}

// The reduction is deferred: if the next thing is the control step (one shard: no allreduce in
// between), both run as one launch (k_reduce_control); anything that needs the staged sums earlier
// (sums_buffer() for the allreduce) flushes it as its own kernel.
int HipBackend::reduce_partials(int64_t rows, bool guarded) {
  if (pending_rows_ >= 0 && flush_reduce()) return -1;
  pending_rows_ = rows;
  pending_guarded_ = guarded;
  return 0;
}

int HipBackend::flush_reduce() {
  if (pending_rows_ < 0) return 0;
  const int64_t rows = pending_rows_;
  pending_rows_ = -1;
  prof_begin(SABC_KERNEL_REDUCE);
  HB_LAUNCH(launch_reduce_partials(partials_, rows, np_, sums_stage_, pending_guarded_ ? &cb_dev_->halt : nullptr, stream_),
            "k_reduce_partials");
  if (pending_xchg_) {          // somebody wants the GLOBAL sums in the staging buffer: the exchange without the control step
    pending_xchg_ = false;
    ControlArgs none;
    std::memset(&none, 0, sizeof(none));
    none.mode = pending_guarded_ ? CTRL_GUARDED : 0;
    const P2PView pv = p2p_view();
    HB_LAUNCH(launch_reduce_control(partials_, -1, np_, sums_stage_, pending_guarded_, cb_dev_, none, hist_dev_, mbox_dev_, stream_, &pv,
                                    tag(++xseq_), /*do_control=*/false, take_silence()), "k_reduce_control (exchange)");
  }
  prof_end(SABC_KERNEL_REDUCE);
  return 0;
}

double *HipBackend::sums_buffer() {
  (void)flush_reduce();
  return sums_stage_;
}

int HipBackend::control(const ControlArgs &a) {
  const bool xchg = pending_xchg_ && pending_rows_ >= 0;
  const P2PView pv = xchg ? p2p_view() : P2PView();
  if (pending_rows_ >= 0 && np_ <= 64 && pending_rows_ * np_ <= fuse_reduce_max_) {
    const int64_t rows = pending_rows_;
    pending_rows_ = -1;
    pending_xchg_ = false;
    prof_begin(SABC_KERNEL_REDUCE);
    // several shards over the peer-to-peer slots: reduce -> exchange -> control step, ONE launch
    HB_LAUNCH(launch_reduce_control(partials_, rows, np_, sums_stage_, pending_guarded_, cb_dev_, a, hist_dev_, mbox_dev_, stream_,
                                    xchg ? &pv : nullptr, xchg ? tag(++xseq_) : 0, true, xchg ? take_silence() : 0),
              "k_reduce_control");
    prof_end(SABC_KERNEL_REDUCE);
    return 0;
  }
  if (xchg) {                   // a partial-row matrix too large for one workgroup: np workgroups reduce it first
    pending_xchg_ = false;
    if (flush_reduce()) return -1;
    prof_begin(SABC_KERNEL_REDUCE);
    HB_LAUNCH(launch_reduce_control(partials_, -1, np_, sums_stage_, pending_guarded_, cb_dev_, a, hist_dev_, mbox_dev_, stream_, &pv,
                                    tag(++xseq_), true, take_silence()), "k_reduce_control (exchange)");
    prof_end(SABC_KERNEL_REDUCE);
    return 0;
  }
  if (flush_reduce()) return -1;
  HB_LAUNCH(launch_control(cb_dev_, a, hist_dev_, mbox_dev_, sums_stage_, stream_), "k_control");
  return 0;
}

// Poll the mailbox.  A stream that has drained without the sequence word arriving means the
// control kernel never ran (a fault upstream): report instead of spinning forever.
int HipBackend::wait_notify(int64_t seq, int64_t *n_accept, int *error, int *halted) {
  Mailbox *mb = mbox_host_ + (seq % kMailboxRing);
  int32_t e = 0, hl = 0;
  for (uint64_t spins = 1;; ++spins) {
    if (mailbox_unpack(mb->w0, mb->w1, seq, n_accept, &e, &hl)) break;
    __builtin_ia32_pause();
    if ((spins & 0x3FFF) == 0) {
      const hipError_t q = hipStreamQuery(stream_);
      if (q == hipSuccess) {
        if (mailbox_unpack(mb->w0, mb->w1, seq, n_accept, &e, &hl)) break;
        err_ = "control step did not report back although the stream is idle";
        return -1;
      }
      if (q != hipErrorNotReady) return check(q, "hipStreamQuery");
    }
  }
  *error = (int)e;
  *halted = (int)hl;
  return 0;
}

int HipBackend::read_control(ControlBlock *out) {
  HB_CHECK(hipMemcpyAsync(cb_host_, cb_dev_, sizeof(ControlBlock), hipMemcpyDeviceToHost, stream_), "memcpy(control block)");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  std::memcpy(out, cb_host_, sizeof(ControlBlock));
  return 0;
}

int HipBackend::write_control(const ControlBlock &in) {
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");     // the staging copy is reused
  std::memcpy(cb_host_, &in, sizeof(ControlBlock));
  HB_CHECK(hipMemcpyAsync(cb_dev_, cb_host_, sizeof(ControlBlock), hipMemcpyHostToDevice, stream_), "memcpy(control block)");
  return 0;
}

int HipBackend::history_reserve(int64_t rows) {
  const int row_len = kMaxStats * 3;
  if (rows > hist_cap_) {
    // grow geometrically from 4096 rows (0.8 MB): a typical call never pays hipFree + hipMalloc inside update_population!
    int64_t cap = hist_cap_ > 0 ? 2 * hist_cap_ : 4096;
    if (cap < rows) cap = rows;
    HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    free_later(hist_dev_);
    hist_dev_ = nullptr;
    HB_CHECK(hipMalloc((void **)&hist_dev_, (size_t)cap * row_len * sizeof(double)), "hipMalloc(history)");
    hist_cap_ = cap;
  }
  return 0;
}

int HipBackend::read_history(double *out, int64_t rows, int row_len) {
  if (rows > hist_cap_) { err_ = "read_history: more rows than reserved"; return -1; }
  HB_CHECK(hipMemcpyAsync(out, hist_dev_, (size_t)rows * row_len * sizeof(double), hipMemcpyDeviceToHost, stream_), "memcpy(history)");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  return 0;
}

int HipBackend::resample_weights(double delta) {
  HB_LAUNCH(launch_resample_weights(m_, pop_ptrs(cur_), cb_dev_, (double)sh_.n_global, delta, stream_), "k_resample_weights");
  return 0;
}

int HipBackend::resample_draw(const double *gathered_pop, uint64_t iter) {
  const int rows = m_.d + m_.s + 1;
  prof_begin(SABC_KERNEL_RESAMPLE);
  const ShardBlocks blocks = flat_blocks(gathered_pop, rows, sh_.cap, sh_.world);
  HB_LAUNCH(launch_weight_scan(blocks, sh_.n_global, block_sums_, cum_, totals_dev_, totals_host_dev_, stream_), "weight scan");
  launches_ += 2;
  const int nxt = 1 - cur_;
  HB_LAUNCH(launch_resample_gather(m_, blocks, sh_.n_global, cum_, block_sums_, totals_dev_, iter, pop_ptrs(nxt), stream_),
            "k_resample_gather");
  prof_end(SABC_KERNEL_RESAMPLE);
  flip_cur();
  return 0;
}

// one shard: the whole of :124-137 in four launches (kernels.hpp: launch_resample_local)
int HipBackend::resample_local(double delta, uint64_t iter, int64_t *stats_rows) {
  if (!pack_dev_) {
    const int64_t doubles = resample_pack_doubles(m_.d + m_.s, sh_.cap > 0 ? sh_.cap : 1);
    if (doubles > 0) HB_CHECK(hipMalloc((void **)&pack_dev_, (size_t)doubles * sizeof(double)), "hipMalloc(packed resample lines)");
  }
  if (pending_rows_ >= 0 && flush_reduce()) return -1;      // the partial rows are about to be overwritten
  const int nxt = 1 - cur_;
  prof_begin(SABC_KERNEL_RESAMPLE);
  HB_LAUNCH(launch_resample_local(m_, pop_ptrs(cur_), pop_ptrs(nxt), cb_dev_, delta, iter, block_sums_, cum_, totals_dev_, totals_host_dev_,
                                  pack_dev_, partials_, stats_rows, stream_), "resample kernels");
  launches_ += 3;
  prof_end(SABC_KERNEL_RESAMPLE);
  flip_cur();
  return 0;
}

// ---- the sharded resample (engine.cpp: resample_exchange) -------------------------------------
int HipBackend::resample_select(const double *gathered_w, uint64_t iter) {
  if (!idx_dev_) {
    const size_t cap = (size_t)(sh_.cap > 0 ? sh_.cap : 1);
    HB_CHECK(hipMalloc((void **)&idx_dev_, cap * sizeof(int64_t)), "hipMalloc(resample indices)");
    HB_CHECK(hipMalloc((void **)&slot_dev_, cap * sizeof(int64_t)), "hipMalloc(resample slots)");
    HB_CHECK(hipMalloc((void **)&bucket_dev_, 2 * (size_t)sh_.world * sizeof(unsigned long long)), "hipMalloc(buckets)");
    HB_CHECK(hipHostMalloc((void **)&bucket_host_, 2 * (size_t)sh_.world * sizeof(unsigned long long)), "hipHostMalloc(buckets)");
  }
  prof_begin(SABC_KERNEL_RESAMPLE);
  HB_LAUNCH(launch_weight_scan(flat_blocks(gathered_w, 1, sh_.cap, sh_.world), sh_.n_global, block_sums_, cum_, totals_dev_, totals_host_dev_, stream_),
            "weight scan");
  launches_ += 2;
  HB_LAUNCH(launch_resample_select(m_, sh_.cap, sh_.n_global, cum_, block_sums_, totals_dev_, iter, pop_ptrs(cur_), idx_dev_, stream_),
            "k_resample_select");
  prof_end(SABC_KERNEL_RESAMPLE);
  return 0;
}

int HipBackend::resample_bucket(int64_t *counts_host, double *req_out) {
  const int W = sh_.world;
  const size_t bytes = (size_t)W * sizeof(unsigned long long);
  HB_CHECK(hipMemsetAsync(bucket_dev_, 0, 2 * bytes, stream_), "hipMemset(buckets)");
  HB_LAUNCH(launch_bucket_count(idx_dev_, sh_.n_local, sh_.cap, bucket_dev_, stream_), "k_bucket_count");
  HB_CHECK(hipMemcpyAsync(bucket_host_, bucket_dev_, bytes, hipMemcpyDeviceToHost, stream_), "memcpy(bucket counts)");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  unsigned long long run = 0;
  for (int r = 0; r < W; ++r) {
    counts_host[r] = (int64_t)bucket_host_[r];
    bucket_host_[W + r] = run;                       // exclusive offsets = where each bucket's cursor starts
    run += bucket_host_[r];
  }
  if ((int64_t)run != sh_.n_local) { err_ = "resample_bucket: the bucket counts do not add up to n_local"; return -1; }
  HB_CHECK(hipMemcpyAsync(bucket_dev_ + W, bucket_host_ + W, bytes, hipMemcpyHostToDevice, stream_), "memcpy(bucket cursors)");
  HB_LAUNCH(launch_bucket_scatter(idx_dev_, sh_.n_local, sh_.cap, bucket_dev_ + W, req_out, slot_dev_, stream_), "k_bucket_scatter");
  return 0;
}

int HipBackend::resample_serve(const double *req_in, int64_t m, double *rows_out) {
  HB_LAUNCH(launch_resample_serve(req_in, m, m_.d + m_.s, pop_ptrs(cur_), rows_out, stream_), "k_resample_serve");
  return 0;
}

int HipBackend::resample_scatter(const double *rows_in) {
  const int nxt = 1 - cur_;
  HB_LAUNCH(launch_resample_scatter(rows_in, slot_dev_, sh_.n_local, m_.d + m_.s, pop_ptrs(nxt), stream_), "k_resample_scatter");
  flip_cur();
  return 0;
}

double HipBackend::last_ess() {
  if (stream_) (void)hipStreamSynchronize(stream_);
  return totals_host_ && totals_host_[1] > 0 ? totals_host_[0] * totals_host_[0] / totals_host_[1] : 0.0;   // :134
}

int HipBackend::download(double *theta, double *u, double *rho) {
  const size_t w = (size_t)sh_.n_local * sizeof(double), pitch = (size_t)sh_.cap * sizeof(double);
  if (sh_.n_local > 0) {
    if (theta) HB_CHECK(hipMemcpy2DAsync(theta, w, pop_[cur_], pitch, w, (size_t)m_.d, hipMemcpyDeviceToHost, stream_), "download theta");
    if (u) HB_CHECK(hipMemcpy2DAsync(u, w, pop_[cur_] + (size_t)m_.d * sh_.cap, pitch, w, (size_t)m_.s, hipMemcpyDeviceToHost, stream_), "download u");
    if (rho) HB_CHECK(hipMemcpy2DAsync(rho, w, rho_, pitch, w, (size_t)m_.s, hipMemcpyDeviceToHost, stream_), "download rho");
  }
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  return 0;
}

int HipBackend::upload(const double *theta, const double *u, const double *rho) {
  const size_t w = (size_t)sh_.n_local * sizeof(double), pitch = (size_t)sh_.cap * sizeof(double);
  if (sh_.n_local > 0) {
    if (theta) HB_CHECK(hipMemcpy2DAsync(pop_[cur_], pitch, theta, w, w, (size_t)m_.d, hipMemcpyHostToDevice, stream_), "upload theta");
    if (u) HB_CHECK(hipMemcpy2DAsync(pop_[cur_] + (size_t)m_.d * sh_.cap, pitch, u, w, w, (size_t)m_.s, hipMemcpyHostToDevice, stream_), "upload u");
    if (rho) HB_CHECK(hipMemcpy2DAsync(rho_, pitch, rho, w, w, (size_t)m_.s, hipMemcpyHostToDevice, stream_), "upload rho");
  }
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  return 0;
}

int HipBackend::get_knots(int stat, double *out, int64_t len) {
  if (stat < 0 || stat >= m_.s || len > cdf_len_[stat]) { err_ = "get_knots: bad statistic index or length"; return -1; }
  HB_CHECK(hipMemcpyAsync(out, knots_ + (int64_t)stat * knot_stride_, (size_t)len * sizeof(double), hipMemcpyDeviceToHost, stream_), "memcpy(knots)");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  return 0;
}

int HipBackend::set_knots(int stat, const double *knots, int64_t len) {
  if (stat < 0 || stat >= m_.s || len < 3 || len > knot_stride_) { err_ = "set_knots: bad statistic index or length"; return -1; }
  HB_CHECK(hipMemcpyAsync(knots_ + (int64_t)stat * knot_stride_, knots, (size_t)len * sizeof(double), hipMemcpyHostToDevice, stream_), "memcpy(knots)");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  cdf_len_[stat] = len;
  return build_coarse(stat);
}

// index levels of the ECDF search (device_models.hpp): coarse = the smallest shift with ceil(len / 2^shift) <= cdf_coarse_entries(s)
int HipBackend::build_coarse(int stat) {
  int shift = 0;
  const int nc = cdf_coarse_entries(m_.s);
  while ((((int64_t)cdf_len_[stat] + ((int64_t)1 << shift) - 1) >> shift) > nc) ++shift;
  cdf_shift_[stat] = shift;
  HB_LAUNCH(launch_cdf_index(knots_ + (int64_t)stat * knot_stride_, cdf_len_[stat], knot_stride_, shift,
                             coarse_ + (int64_t)stat * nc, nc, mid_ + (int64_t)stat * mid_stride_, mid_stride_, stream_),
            "k_cdf_index");
  return 0;
}

int HipBackend::cdf_apply_host(const double *rho, int64_t m, double *u_out) {
  if (m <= 0) return 0;
  double *d_in = nullptr, *d_out = nullptr;
  const size_t bytes = (size_t)m * m_.s * sizeof(double);
  HB_CHECK(hipMalloc((void **)&d_in, bytes), "hipMalloc");
  HB_CHECK(hipMalloc((void **)&d_out, bytes), "hipMalloc");
  int rc = check(hipMemcpyAsync(d_in, rho, bytes, hipMemcpyHostToDevice, stream_), "memcpy");
  if (!rc) rc = check((hipError_t)launch_cdf_apply_matrix(cdf_ptrs(), m_.s, d_in, m, d_out, stream_), "k_cdf_apply_matrix");
  if (!rc) rc = check(hipMemcpyAsync(u_out, d_out, bytes, hipMemcpyDeviceToHost, stream_), "memcpy");
  if (!rc) rc = check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  (void)hipFree(d_in); (void)hipFree(d_out);
  return rc;
}

int HipBackend::prior_host(uint64_t pid0, int64_t n, double *theta_out, double *logpdf_out) {
  if (m_.prior_joint == 2) { err_ = "sabc_op_prior: the prior of this handle lives in host callbacks"; return -1; }
  if (m_.prior_joint == 3 && !(rtc() && rtc()->prior_op)) { err_ = "sabc_op_prior: no device simulator source (with its prior) registered"; return -1; }
  if (n <= 0) return 0;
  double *d_th = nullptr, *d_lp = nullptr;
  HB_CHECK(hipSetDevice(device_), "hipSetDevice");
  HB_CHECK(hipMalloc((void **)&d_th, (size_t)n * m_.d * sizeof(double)), "hipMalloc");
  HB_CHECK(hipMalloc((void **)&d_lp, (size_t)n * sizeof(double)), "hipMalloc");
  int rc = check((hipError_t)launch_prior_op(m_, pid0, n, d_th, d_lp, stream_, m_.prior_joint == 3 ? rtc() : nullptr), "k_prior_op");
  if (!rc) rc = check(hipMemcpyAsync(theta_out, d_th, (size_t)n * m_.d * sizeof(double), hipMemcpyDeviceToHost, stream_), "memcpy");
  if (!rc) rc = check(hipMemcpyAsync(logpdf_out, d_lp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream_), "memcpy");
  if (!rc) rc = check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  (void)hipFree(d_th); (void)hipFree(d_lp);
  return rc;
}

int HipBackend::simulate_host(const double *theta, int64_t n, uint64_t pid0, uint64_t iter, double *rho_out) {
  if (n <= 0) return 0;
  double *d_in = nullptr, *d_out = nullptr;
  HB_CHECK(hipMalloc((void **)&d_in, (size_t)n * m_.d * sizeof(double)), "hipMalloc");
  HB_CHECK(hipMalloc((void **)&d_out, (size_t)n * m_.s * sizeof(double)), "hipMalloc");
  int rc = check(hipMemcpyAsync(d_in, theta, (size_t)n * m_.d * sizeof(double), hipMemcpyHostToDevice, stream_), "memcpy");
  if (!rc) rc = check((hipError_t)launch_simulate_batch(m_, d_in, n, pid0, iter, d_out, stream_, rtc()), "k_simulate_batch");
  if (!rc) rc = check(hipMemcpyAsync(rho_out, d_out, (size_t)n * m_.s * sizeof(double), hipMemcpyDeviceToHost, stream_), "memcpy");
  if (!rc) rc = check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  (void)hipFree(d_in); (void)hipFree(d_out);
  return rc;
}

// ---- peer-to-peer transport (p2p.hpp) ----------------------------------------------------------
int64_t HipBackend::parked_bytes() { return g_parked_bytes.load(); }

P2PView HipBackend::p2p_view() const {
  P2PView v;
  std::memset(&v, 0, sizeof(v));
  for (int r = 0; r < kMaxPeers; ++r) v.slots[r] = peer_slots_[r];
  v.rank = sh_.rank;
  v.world = sh_.world;
  v.timeout_ticks = (uint64_t)(p2p_timeout_ms_ * (double)wall_clock_khz_);
  return v;
}

int HipBackend::p2p_descriptor(P2PDesc *out) {
  std::memset(out, 0, sizeof(*out));
  if (sh_.world < 2 || sh_.world > kMaxPeers) { err_ = "the peer-to-peer transport takes 2..8 shards (one node)"; return -1; }
  HB_CHECK(hipSetDevice(device_), "hipSetDevice");
  // a new set-up (after a failed call switched the transport off, or on top of a live one): this shard LEAVES the old group
  // first -- its peers are unmapped and told so -- before anything of the new one is exported
  if (p2p_leave()) return -1;
  if (!page_) {
    page_ = p2p_page_create(page_name_);
    if (!page_) { err_ = "the peer-to-peer transport needs POSIX shared memory for its host page (shm_open failed)"; return -1; }
  }
  if (!slots_) {
    // fine-grained, uncached device memory: a peer's store is visible to this device's loads without a cache to go through
    hipError_t e = hipExtMallocWithFlags((void **)&slots_, (size_t)kP2PSlotWords * 8, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags((void **)&slots_, (size_t)kP2PSlotWords * 8, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) { slots_ = nullptr; return check(e, "hipExtMallocWithFlags(slot area)"); }
    HB_CHECK(hipMalloc((void **)&p2p_test_dev_, (size_t)(2 * kMaxPartials + 2 + p2p_pattern_save_words() + 2) * sizeof(double)), "hipMalloc(self-test)");
  }
  // the slots are wiped and the running numbers start over.  (Correctness does not rest on the wipe: every word carries the
  // set-up generation, and a word of an earlier generation -- a status post still in flight from an old peer -- matches nothing.)
  HB_CHECK(hipMemsetAsync(slots_, 0, (size_t)kP2PSlotWords * 8, stream_), "hipMemset(slot area)");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  xseq_ = bseq_ = call_ = 0;
  out->magic = kP2PMagic;
  out->pid = (int32_t)getpid();
  out->device = device_;
  out->rank = sh_.rank; out->world = sh_.world;
  out->cap = sh_.cap; out->n_global = sh_.n_global;
  out->d = m_.d; out->s = m_.s;
  out->ptr_slots = (uint64_t)(uintptr_t)slots_;
  out->ptr_pop[0] = (uint64_t)(uintptr_t)pop_[0]; out->ptr_pop[1] = (uint64_t)(uintptr_t)pop_[1];
  out->ptr_rho = (uint64_t)(uintptr_t)rho_;
  out->cur = cur_;
  out->gen_proposal = gen_ >= kP2PMaxGen ? 1u : gen_ + 1u;
  out->ptr_page = (uint64_t)(uintptr_t)page_;
  std::memcpy(out->page_name, page_name_, sizeof(out->page_name));
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "P2PDesc holds 64-byte IPC handles");
  // the handles are only needed by shards in OTHER processes; a failure here surfaces there (all-zero handle)
  hipIpcMemHandle_t hd;
  void *what[4] = {slots_, pop_[0], pop_[1], rho_};
  unsigned char *where[4] = {out->ipc_slots, out->ipc_pop[0], out->ipc_pop[1], out->ipc_rho};
  for (int i = 0; i < 4; ++i) {
    if (hipIpcGetMemHandle(&hd, what[i]) == hipSuccess) std::memcpy(where[i], &hd, 64);
    else (void)hipGetLastError();
  }
  exported_ = true;                                     // from here on a peer may hold a mapping of this shard's memory
  return 0;
}

int HipBackend::p2p_init(const P2PDesc *all) {
  if (!slots_ || !page_) { err_ = "sabc_comm_p2p_descriptor has to be called first"; return -1; }
  HB_CHECK(hipSetDevice(device_), "hipSetDevice");
  const int W = sh_.world;
  if (mapped_ && p2p_leave()) return -1;                // (init twice without a new descriptor)
  p2p_on_ = false;
  int khz = 0;                                          // rate of the constant wall clock the waits are bounded by
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_) == hipSuccess && khz > 0) wall_clock_khz_ = khz;
  else (void)hipGetLastError();
  uint32_t proposals[kMaxPeers] = {0};
  for (int r = 0; r < W; ++r) {
    const P2PDesc &d = all[r];
    if (d.magic != kP2PMagic || d.rank != r || d.world != W || d.cap != sh_.cap || d.n_global != sh_.n_global || d.d != m_.d || d.s != m_.s) {
      err_ = "peer-to-peer descriptor of a shard does not match this handle's configuration";
      return -1;
    }
    proposals[r] = d.gen_proposal;
  }
  // the group's generation: above every member's last one; from here on this shard counts as mapped -- whatever goes wrong
  // below is undone by p2p_leave(), which also tells the peers (through the host page) that nothing of theirs stays mapped
  gen_ = p2p_agree_gen(proposals, W);
  flips_ = 0;
  page_->gen.store(gen_, std::memory_order_relaxed);
  page_->cur_parity.store((uint32_t)cur_, std::memory_order_relaxed);
  page_->state.store(kP2PNone, std::memory_order_release);
  mapped_ = true;
  auto fail = [&](const std::string &why) { (void)p2p_leave(); err_ = why; return -1; };
  for (int r = 0; r < W; ++r) {
    const P2PDesc &d = all[r];
    peer_cur0_[r] = d.cur & 1;
    if (r == sh_.rank) {
      peer_slots_[r] = slots_; peer_pop_[0][r] = pop_[0]; peer_pop_[1][r] = pop_[1]; peer_rho_[r] = rho_;
      peer_page_[r] = page_; peer_page_shm_[r] = false;
      continue;
    }
    if (d.pid == (int32_t)getpid()) {                   // same process: the pointers themselves
      if (d.device != device_) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device_, d.device) != hipSuccess || !can) return fail("no peer access between the devices of two shards");
        const hipError_t e = hipDeviceEnablePeerAccess(d.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)check(e, "hipDeviceEnablePeerAccess"); return fail(err_); }
        (void)hipGetLastError();
      }
      peer_slots_[r] = (uint64_t *)(uintptr_t)d.ptr_slots;
      peer_pop_[0][r] = (double *)(uintptr_t)d.ptr_pop[0]; peer_pop_[1][r] = (double *)(uintptr_t)d.ptr_pop[1];
      peer_rho_[r] = (double *)(uintptr_t)d.ptr_rho;
      // (the host page is opened by NAME even here: a mapping of this shard's own, which stays readable after the peer
      // has destroyed its handle and unmapped its side -- this shard may be polling it for `released` at that moment)
      if (!open_peer_page(r, d)) return fail("a peer shard's host page could not be opened (POSIX shared memory)");
      continue;
    }
    if (d.device != device_) {                          // another GPU of the node: kernels here must be able to reach it
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, device_, d.device) != hipSuccess || !can) {
        (void)hipGetLastError();
        return fail("no peer access between the devices of two shards (is the peer on this node?)");
      }
    }
    if (!open_peer_page(r, d)) return fail("a peer shard's host page could not be opened (POSIX shared memory; is the peer on this node?)");
    const unsigned char *from[4] = {d.ipc_slots, d.ipc_pop[0], d.ipc_pop[1], d.ipc_rho};
    void *got[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i) {
      hipIpcMemHandle_t hd;
      std::memcpy(&hd, from[i], 64);
      const hipError_t e = hipIpcOpenMemHandle(&got[i], hd, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { (void)check(e, "hipIpcOpenMemHandle (a peer shard's memory)"); return fail(err_); }
      ipc_opened_.push_back(got[i]);
    }
    peer_slots_[r] = (uint64_t *)got[0];
    peer_pop_[0][r] = (double *)got[1]; peer_pop_[1][r] = (double *)got[2];
    peer_rho_[r] = (double *)got[3];
  }
  page_->state.store(kP2PActive, std::memory_order_release);
  p2p_on_ = true;
  return 0;
}

bool HipBackend::open_peer_page(int r, const P2PDesc &d) {
  if (peer_page_shm_[r]) p2p_page_unmap(peer_page_[r]);           // (a page kept from an earlier set-up)
  char name[sizeof(d.page_name) + 1];
  std::memcpy(name, d.page_name, sizeof(d.page_name)); name[sizeof(d.page_name)] = 0;
  peer_page_[r] = p2p_page_open(name);
  peer_page_shm_[r] = peer_page_[r] != nullptr;
  return peer_page_[r] != nullptr;
}

// p2p.hpp "LEAVES".  Safe to call in any state and more than once; never frees anything a peer may have mapped.
int HipBackend::p2p_leave() {
  pending_xchg_ = false;
  p2p_on_ = false;
  if (!mapped_) return 0;
  (void)hipSetDevice(device_);
  page_->state.store(kP2PLeaving, std::memory_order_release);
  if (stream_) {
    // the peers' waits for this shard give up at once; then everything this shard has in flight -- it may be reading the
    // peers' populations -- is drained before their memory is unmapped
    const P2PView pv = p2p_view();
    (void)hipGetLastError();
    (void)launch_p2p_leave(pv, gen_, stream_);
    launches_ += 1;
    (void)hipStreamSynchronize(stream_);
    (void)hipGetLastError();
  }
  for (void *p : ipc_opened_) (void)hipIpcCloseMemHandle(p);
  ipc_opened_.clear();
  (void)hipGetLastError();
  for (int r = 0; r < kMaxPeers; ++r) {
    peer_slots_[r] = nullptr; peer_pop_[0][r] = peer_pop_[1][r] = nullptr; peer_rho_[r] = nullptr;
    page_->released[r].store(gen_, std::memory_order_release);          // "nothing of shard r's generation-gen_ memory is mapped here"
  }
  mapped_ = false;
  return 0;
}

bool HipBackend::p2p_peers_present() {
  if (!mapped_ || !p2p_on_) return true;
  for (int r = 0; r < sh_.world; ++r) {
    const P2PHostPage *pg = peer_page_[r];
    if (r == sh_.rank || !pg) continue;
    if (pg->gen.load(std::memory_order_acquire) != gen_ || pg->state.load(std::memory_order_acquire) != kP2PActive) return false;
  }
  return true;
}

// Destructor: leave, then wait (bounded) until every peer has recorded that it unmapped this shard's memory.  true: the
// memory peers could map may be freed; false: it has to be parked.
bool HipBackend::p2p_finish() {
  const P2PHostPage *pages[kMaxPeers];
  for (int r = 0; r < kMaxPeers; ++r) pages[r] = peer_page_[r];
  (void)p2p_leave();
  bool ok = true;
  if (exported_) {
    const double wait_ms = destroy_wait_ms_ < 0 ? p2p_timeout_ms_ : destroy_wait_ms_;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < sh_.world && r < kMaxPeers; ++r) {
      if (r == sh_.rank) continue;
      // a peer this shard never got to know (set-up stopped before or inside sabc_comm_p2p_init) cannot acknowledge
      if (!pages[r] || gen_ == 0) { ok = false; continue; }
      // acknowledged: the peer has unmapped this generation -- or has moved on to a later set-up, which begins by leaving
      while (pages[r]->released[sh_.rank].load(std::memory_order_acquire) != gen_ && pages[r]->gen.load(std::memory_order_acquire) <= gen_) {
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > wait_ms) { ok = false; break; }
        usleep(50);
      }
    }
  }
  for (int r = 0; r < kMaxPeers; ++r) {
    if (peer_page_shm_[r]) p2p_page_unmap(peer_page_[r]);
    peer_page_[r] = nullptr; peer_page_shm_[r] = false;
  }
  if (page_) {
    page_->state.store(kP2PGone, std::memory_order_release);
    p2p_page_destroy(page_, page_name_);
    page_ = nullptr;
  }
  return ok;
}

// First contact.  (1) a row of known values through the slots + one barrier, the host checks the sums; (2) what the
// transport READS: selftest_patterns().  Sequence numbers advance exactly as in a real exchange, so every shard has to call
// it the same number of times.
int HipBackend::p2p_selftest() {
  if (!p2p_on_) { err_ = "the peer-to-peer transport is not initialised"; return -1; }
  HB_CHECK(hipSetDevice(device_), "hipSetDevice");
  const int np = 7, W = sh_.world;
  double in[np], out[np];
  for (int q = 0; q < np; ++q) in[q] = (double)(sh_.rank + 1) * (q + 1) + (q == 3 ? 0.1 : 0.0);
  double *d_in = p2p_test_dev_, *d_out = p2p_test_dev_ + kMaxPartials;
  int *d_failed = (int *)(p2p_test_dev_ + 2 * kMaxPartials);
  HB_CHECK(hipMemcpyAsync(d_in, in, sizeof(in), hipMemcpyHostToDevice, stream_), "memcpy");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  const P2PView pv = p2p_view();
  HB_LAUNCH(launch_p2p_selftest(pv, tag(++xseq_), np, d_in, d_out, d_failed, take_silence(), stream_), "k_p2p_selftest");
  HB_LAUNCH(launch_p2p_barrier(pv, tag(++bseq_), cb_dev_, false, take_silence(), stream_), "k_p2p_barrier");
  int failed = 1;
  HB_CHECK(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, stream_), "memcpy");
  HB_CHECK(hipMemcpyAsync(&failed, d_failed, sizeof(int), hipMemcpyDeviceToHost, stream_), "memcpy");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  ControlBlock cb;
  if (read_control(&cb)) return -1;
  bool slots_ok = !(failed || cb.error == SABC_ERR_COMM);
  std::string why = slots_ok ? "" : "peer-to-peer self-test: a shard did not post within the bound";
  for (int q = 0; slots_ok && q < np; ++q) {
    double want = 0.0;
    for (int r = 0; r < W; ++r) { const double x = (double)(r + 1) * (q + 1) + (q == 3 ? 0.1 : 0.0); want = r == 0 ? x : want + x; }
    if (out[q] != want) { slots_ok = false; why = "peer-to-peer self-test: wrong sums came back through the slots"; }
  }
  // the second half runs whatever the first said: the shards stay in step (its barriers return at once behind an error)
  const int prc = selftest_patterns(pv);
  if (!slots_ok) { err_ = why; p2p_on_ = false; return -1; }
  if (prc) { p2p_on_ = false; return -1; }
  return 0;
}

// What the transport reads (kernels.hip: k_p2p_pattern_*): two rounds of write -> barrier -> read every shard's samples ->
// barrier, then the parked values go back.  Works on live populations (a set-up after sabc_initialize).
int HipBackend::selftest_patterns(const P2PView &pv) {
  const int64_t len[3] = {(int64_t)(m_.d + m_.s + 1) * sh_.cap, (int64_t)(m_.d + m_.s + 1) * sh_.cap, (int64_t)m_.s * sh_.cap};
  double *own[3] = {pop_[0], pop_[1], rho_};
  const double *peers[3][kMaxPeers];
  for (int r = 0; r < kMaxPeers; ++r) { peers[0][r] = peer_pop_[0][r]; peers[1][r] = peer_pop_[1][r]; peers[2][r] = peer_rho_[r]; }
  double *save = p2p_test_dev_ + 2 * kMaxPartials + 2;
  unsigned int *d_out = (unsigned int *)(save + p2p_pattern_save_words());
  unsigned int res[2][2] = {{0, 0}, {0, 0}};
  for (int round = 1; round <= 2; ++round) {
    HB_LAUNCH(launch_p2p_pattern_write(own, len, save, gen_, round, sh_.rank, round == 1 ? 0 : 1, stream_), "k_p2p_pattern_write");
    HB_LAUNCH(launch_p2p_barrier(pv, tag(++bseq_), cb_dev_, false, take_silence(), stream_), "k_p2p_barrier");   // every shard's pattern is written
    HB_CHECK(hipMemsetAsync(d_out, 0, 2 * sizeof(unsigned int), stream_), "memset");
    // (test hook: a shard told to see stale data compares the second round against a pattern nobody wrote)
    const int expect = (round == 2 && p2p_stale_ > 0) ? 3 : round;
    HB_LAUNCH(launch_p2p_pattern_check(peers, len, gen_, expect, sh_.world, d_out, stream_), "k_p2p_pattern_check");
    HB_CHECK(hipMemcpyAsync(res[round - 1], d_out, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, stream_), "memcpy");
    HB_LAUNCH(launch_p2p_barrier(pv, tag(++bseq_), cb_dev_, false, take_silence(), stream_), "k_p2p_barrier");   // every shard has read
  }
  if (p2p_stale_ > 0) --p2p_stale_;
  HB_LAUNCH(launch_p2p_pattern_write(own, len, save, gen_, 0, sh_.rank, 2, stream_), "k_p2p_pattern_write (restore)");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  ControlBlock cb;
  if (read_control(&cb)) return -1;
  if (cb.error == SABC_ERR_COMM) { err_ = "peer-to-peer self-test: a shard did not reach a barrier within the bound"; return -1; }
  for (int round = 1; round <= 2; ++round)
    if (res[round - 1][0]) {
      static const char *what[3] = {"population buffer 0", "population buffer 1", "rho"};
      const unsigned w = res[round - 1][1];
      char buf[256];
      std::snprintf(buf, sizeof(buf), "peer-to-peer self-test: %u of the words read from the shards' memory were not what their owners wrote "
                    "(round %d; first: shard %u, %s, sample %u) -- a kernel boundary does not make a peer's plain device memory visible here",
                    res[round - 1][0], round, w >> 28, what[((w >> 24) & 15) % 3], w & 0xFFFFFFu);
      err_ = buf;
      return -1;
    }
  return 0;
}

int HipBackend::snapshot() {
  const size_t pop_bytes = (size_t)(m_.d + m_.s + 1) * (size_t)sh_.cap * sizeof(double), rho_bytes = (size_t)m_.s * (size_t)sh_.cap * sizeof(double);
  if (!snap_pop_) {
    HB_CHECK(hipMalloc((void **)&snap_pop_, pop_bytes), "hipMalloc(snapshot)");
    HB_CHECK(hipMalloc((void **)&snap_rho_, rho_bytes), "hipMalloc(snapshot)");
  }
  HB_CHECK(hipMemcpyAsync(snap_pop_, pop_[cur_], pop_bytes, hipMemcpyDeviceToDevice, stream_), "snapshot");
  HB_CHECK(hipMemcpyAsync(snap_rho_, rho_, rho_bytes, hipMemcpyDeviceToDevice, stream_), "snapshot");
  return 0;
}

int HipBackend::restore_snapshot() {
  if (!snap_pop_) { err_ = "no snapshot of the particles"; return -1; }
  const size_t pop_bytes = (size_t)(m_.d + m_.s + 1) * (size_t)sh_.cap * sizeof(double), rho_bytes = (size_t)m_.s * (size_t)sh_.cap * sizeof(double);
  pending_rows_ = -1;
  pending_xchg_ = false;
  HB_CHECK(hipMemcpyAsync(pop_[cur_], snap_pop_, pop_bytes, hipMemcpyDeviceToDevice, stream_), "restore");
  HB_CHECK(hipMemcpyAsync(rho_, snap_rho_, rho_bytes, hipMemcpyDeviceToDevice, stream_), "restore");
  HB_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  return 0;
}

int HipBackend::p2p_barrier(bool guarded) {
  if (!p2p_on_) { err_ = "the peer-to-peer transport is not initialised"; return -1; }
  HB_LAUNCH(launch_p2p_barrier(p2p_view(), tag(++bseq_), cb_dev_, guarded, take_silence(), stream_), "k_p2p_barrier");
  return 0;
}

int HipBackend::p2p_commit(int status, bool wait) {
  if (!p2p_on_) { err_ = "the peer-to-peer transport is not initialised"; return -1; }
  if (pending_rows_ >= 0 && flush_reduce()) return -1;
  HB_LAUNCH(launch_p2p_commit(p2p_view(), tag(++call_), status, wait, cb_dev_, take_silence(), stream_), "k_p2p_commit");
  return 0;
}

int HipBackend::build_cdf_p2p(int64_t *len_out, int *any_negative) {
  if (p2p_barrier(false)) return -1;                     // every shard's prior simulations are done
  ShardBlocks b = flat_blocks(nullptr, m_.s, sh_.cap, sh_.world);
  b.direct = 1;
  for (int r = 0; r < sh_.world; ++r) b.peer[r] = peer_rho_[r];
  return build_cdf_blocks(b, len_out, any_negative);
}

int HipBackend::partner_view_p2p(PartnerView *pv) {
  if (!p2p_on_) { err_ = "the peer-to-peer transport is not initialised"; return -1; }
  pv->direct = 1;
  pv->base = nullptr;
  pv->rank_stride = 0;
  pv->cap = sh_.cap;
  for (int r = 0; r < kMaxPeers; ++r) pv->peer[r] = r < sh_.world ? peer_pop_cur(r) : nullptr;   // the OWNER's current buffer
  return 0;
}

int HipBackend::resample_p2p(double delta, uint64_t iter) {
  if (!p2p_on_) { err_ = "the peer-to-peer transport is not initialised"; return -1; }
  if (pending_rows_ >= 0 && flush_reduce()) return -1;
  const int rows = m_.d + m_.s + 1;
  prof_begin(SABC_KERNEL_RESAMPLE);
  HB_LAUNCH(launch_resample_weights(m_, pop_ptrs(cur_), cb_dev_, (double)sh_.n_global, delta, stream_), "k_resample_weights");   // :126-127
  if (p2p_barrier(false)) return -1;                     // every shard's weight row is written
  ShardBlocks b = flat_blocks(nullptr, rows, sh_.cap, sh_.world);
  b.direct = 1;
  for (int r = 0; r < sh_.world; ++r) b.peer[r] = peer_pop_cur(r);
  HB_LAUNCH(launch_weight_scan(b, sh_.n_global, block_sums_, cum_, totals_dev_, totals_host_dev_, stream_), "weight scan");
  launches_ += 2;
  const int nxt = 1 - cur_;
  HB_LAUNCH(launch_resample_gather(m_, b, sh_.n_global, cum_, block_sums_, totals_dev_, iter, pop_ptrs(nxt), stream_), "k_resample_gather");   // :129-132
  prof_end(SABC_KERNEL_RESAMPLE);
  flip_cur();
  return 0;
}

}  // namespace sabc
