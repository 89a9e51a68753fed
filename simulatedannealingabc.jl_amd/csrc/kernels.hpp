// kernels.hpp -- host-callable launchers of the gfx950 kernels (defined in kernels.hip, sort.hip).
// Every launcher only enqueues work on `stream` and returns the hipError_t of the launch as int.
#pragma once
#include "sabc_types.hpp"
#include "p2p.hpp"
#if !defined(__HIPCC_RTC__)     // hipRTC (rtc.cpp) compiles the device half of this header only
#include <hip/hip_runtime_api.h>
#include "rtc.hpp"
#endif

namespace sabc {

// Local shard in HBM.  pop = [(d + s + 1)][cap] doubles: rows 0..d-1 theta, d..d+s-1 u, last row
// resample weights.  rho = [s][cap].  SoA so that lane i touches address base + i (coalesced).
struct PopPtrs {
  double *pop;
  double *rho;
  int64_t cap, n_local, gid0;
};

// coarse level of the ECDF search: every 2^shift-th knot, cdf_coarse_entries(s) per statistic (padded with +inf), in LDS
// for the lifetime of a workgroup of update_block_threads(s) threads.  What a lookup costs is the number of DISTINCT
// LINES it touches behind the LDS level (each a trip to the L2 for every lane), not its number of loads: with 1024
// entries per statistic a block of the table is 1024 knots = 64 entries of the mid level = 4 lines, of which the
// bisection touches 2-3; with 2048 entries it is 2 lines (1-2 touched).  The LDS copy is per workgroup, so models with
// several statistics pay for the finer index with a larger workgroup -- and need 6 waves per SIMD (<= 80 VGPRs) to keep
// three of them on a CU.  Same-box A/B, k_update at n = 1e6, cfg3 | cfg5: 256 threads + 1024 entries 239.5 | 855.6 us;
// 512 + 2048 at 4 waves per SIMD 242.5 | 873.8; 512 + 2048 at 6 waves 234.7 | 845.6 (kept); 1024 + 4096 269 | 935;
// 256 + 2048 244 | 893.  (Cutting the mid level's line-crossing steps out altogether, a wrong-result experiment, gave
// 214 us for cfg3: the lines are worth more than this arrangement recovers.)
#ifndef SABC_CDF_COARSE
#define SABC_CDF_COARSE 1024          // one statistic
#endif
#ifndef SABC_CDF_COARSE_MS
#define SABC_CDF_COARSE_MS 2048       // several statistics
#endif
#ifndef SABC_UPDATE_BLOCK_MS
#define SABC_UPDATE_BLOCK_MS 512
#endif
constexpr int kCdfCoarseMax = SABC_CDF_COARSE > SABC_CDF_COARSE_MS ? SABC_CDF_COARSE : SABC_CDF_COARSE_MS;
// (more than 8 statistics -- source-compiled simulators only --: half the entries, or the index alone would not fit the LDS)
constexpr int cdf_coarse_entries(int s) { return s <= 1 ? SABC_CDF_COARSE : s <= 8 ? SABC_CDF_COARSE_MS : SABC_CDF_COARSE_MS / 2; }

struct CdfPtrs {
  const double *knots;   // [s][stride]; stride is a multiple of 16 (each table starts on a 128-byte line), +inf behind len
  int64_t stride;
  int64_t len[kMaxStats];
  const double *coarse;  // [s][cdf_coarse_entries(s)]
  const double *mid;     // [s][mid_stride]: every 16th knot (the first knot of each line), +inf padded
  int64_t mid_stride;
  int32_t shift[kMaxStats];
};
// entries of the mid level that a search can touch for a table of `stride` knots: (coarse << shift) >> 4 + 1 with
// (coarse << shift) < 2 * stride + 2 * coarse
inline int64_t cdf_mid_stride(int64_t stride) { return ((2 * stride + 2 * kCdfCoarseMax) >> 4) + 2; }

constexpr int kBlock = 256;        // 4 wavefronts of 64
// threads per workgroup of k_update (thread-per-particle form): the grid's tail -- the last workgroups of every CU run
// with idle neighbours -- shrinks with the workgroup, the per-workgroup LDS tables and partial rows grow in number
#ifndef SABC_UPDATE_BLOCK
#define SABC_UPDATE_BLOCK 256
#endif
constexpr int update_block_threads(int s) { return s <= 1 ? SABC_UPDATE_BLOCK : SABC_UPDATE_BLOCK_MS; }
// second argument of __launch_bounds__ for k_update: waves per SIMD the register allocation has to leave room for
#ifndef SABC_UPDATE_MIN_WAVES
#define SABC_UPDATE_MIN_WAVES 4
#endif
#ifndef SABC_UPDATE_MIN_WAVES_MS
#define SABC_UPDATE_MIN_WAVES_MS 6
#endif
// (four or more statistics: the LDS index alone holds a CU to two workgroups = 4 waves per SIMD, so the registers may be used)
constexpr int update_min_waves(int s) {
  return s <= 1 ? SABC_UPDATE_MIN_WAVES : 3 * (cdf_coarse_entries(s) * s * 8 + 4096) <= 160 * 1024 ? SABC_UPDATE_MIN_WAVES_MS : SABC_UPDATE_MIN_WAVES;
}
constexpr int kScanChunk = 1024;   // elements per scan block (4 per thread)
// (a grid-stride variant of k_update with <= 1024 workgroups cost 20 more VGPRs and 11 % of its speed:
// one workgroup per 256 (several statistics: 512) particles and dynamic workgroup scheduling stay)
// partial-row matrices up to this many doubles are summed inside the control kernel (one launch,
// one CU); larger ones by the np-workgroup reduction first
#ifndef SABC_FUSE_REDUCE_MAX
#define SABC_FUSE_REDUCE_MAX 32768
#endif
// (round 2, same-box A/B at n = 1e6 = 19.5 k doubles, four runs each: step minus update kernel 31.2 us with the two
// launches, 30.2 us fused; round 1 had measured the fused form 6 us slower, before the control step worked on an LDS copy)
constexpr int64_t kFuseReduceMaxDoubles = SABC_FUSE_REDUCE_MAX;

inline int64_t n_blocks(int64_t n) { return (n + kBlock - 1) / kBlock; }

// K4 for SMALL shards, the updates of a call in ONE launch (kernels.hip: k_update_persistent): what one launch is asked to do
struct PersistArgs {
  uint64_t iter0;                // global index of the launch's first population update (RNG counter word)
  int64_t ix0, phase, cph;       // ... which is update ix0 of the call; update ix appends a history row iff (phase + ix) % cph == 0
  int64_t act_n, half;           // particles of the shard; size of the first half batch (DifferentialEvolution / StretchMove)
  int32_t count;                 // updates to run (the launch stops early when the resample test fires or an error is raised)
  int32_t test_absent_wg;        // test hook: k > 0: workgroup k - 1 is lost after the rendezvous; k < 0: workgroup -k - 1 never becomes resident
  int32_t active;                // threads of a workgroup that carry particles (a multiple of 64 <= the block: persistent_workgroups)
  int32_t ctrl_wave;             // the wave without particles whose first lane runs the control step beside the next update's drafts
                                 // (= active / 64), or -1: every wave carries particles, wave 0 runs the step between two updates
  double prop_p0, prop_p1;
  ControlArgs ctrl;              // the control step of every update: ACCUMULATE | CHECK | PROPOSAL | EPSILON | PIVOT (history by cadence)
  unsigned long long *sync;      // [0] arrivals at the grid barrier (monotone), [1] abort flag, [2] arrivals at the rendezvous, [3] its decision
                                 // (1 commit | 2 abort); zeroed before the launch
  uint64_t timeout_ticks;        // bound of a wait at the grid barrier / for a row (wall clock)
  uint64_t rendezvous_ticks;     // bound of the wait for everybody at the start of the launch
};
// does k_update_persistent<.., D, S, ..> fit the 160 KB of LDS of a CU?  (its static LDS: the ECDF coarse index, the generator
// tables, the block reduction, a copy of the control block, the sums)  Shapes that do not keep the launch chain.
constexpr bool persistent_fits(int d, int s) {
  return (long)s * cdf_coarse_entries(s) * 8 + 3072 + (long)(update_block_threads(s) / 64 + 2) * n_partials(d, s) * 8 +
         (long)update_block_threads(s) * 8 + (long)sizeof(ControlBlock) + (long)kMaxPartials * 8 + 2048 <= 156 * 1024;
}

#if !defined(__HIPCC_RTC__)
// K1: theta_i ~ prior, rho_i = f_dist(theta_i)                       SimulatedAnnealingABC.jl:172-179
int launch_prior_simulate(const ModelDesc &m, PopPtrs pp, hipStream_t stream, const RtcKernels *rtc = nullptr);
// K3: u_ij = cdf_j(rho_ij) for the whole shard                        :190-192
int launch_cdf_population(const ModelDesc &m, PopPtrs pp, CdfPtrs cdf, hipStream_t stream);
// K4: the per-particle body for `act_n` particles starting at local index act_lo;  :308-331
// writes one partial row per block at partials[(row0 + blockIdx) * np]
// ev0 / ev1: optional timing events carried by the kernel's own dispatch packet
int launch_update(const ModelDesc &m, const StepArgs &c, const ControlBlock *cb, PopPtrs pp, CdfPtrs cdf, PartnerView pv,
                  int64_t act_lo, int64_t act_n, double *partials, int64_t row0, hipStream_t stream,
                  hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const RtcKernels *rtc = nullptr);
// workgroups of the launch (all of them have to be resident at once) or 0 when this model / shape has no persistent form
// (*lanes_out: lanes per particle of that launch, 1, 4 or 16; *active_out: threads of a workgroup that carry particles, fewer than
// the block where the launch can spare a wave for the control step -- kernels.hip: persistent_workgroups)
int64_t persistent_workgroups(const ModelDesc &m, int prop_kind, int64_t act_n, const RtcKernels *rtc = nullptr, int *lanes_out = nullptr,
                              int *active_out = nullptr);
int64_t persistent_workgroups_bound(const ModelDesc &m, int64_t cap);
int launch_update_persistent(const ModelDesc &m, int prop_kind, const PersistArgs &pa, ControlBlock *cb, PopPtrs pp, CdfPtrs cdf,
                             PartnerView pv_a, PartnerView pv_b, double *partials, double *hist, Mailbox *mbox, double *stage,
                             hipStream_t stream, const RtcKernels *rtc = nullptr);
// number of partial rows launch_update() writes for act_n particles (depends on the kernel's granularity)
int64_t update_rows(const ModelDesc &m, int64_t act_n);
// moment sums of the current shard (no update): same partial layout, n_accept = 0
int launch_stats(const ModelDesc &m, const ControlBlock *cb, PopPtrs pp, double *partials, hipStream_t stream,
                 const RtcKernels *rtc = nullptr);
// sums[c] = sum over rows of partials[row][c] in a fixed order
// halt != nullptr: guarded (no-op while *halt is set)
int launch_reduce_partials(const double *partials, int64_t rows, int np, double *sums, const int *halt,
                           hipStream_t stream);
// single-lane state hand-over (control.hpp): n_accept, Sigma / Cholesky, eps, pivot, history row
int launch_control(ControlBlock *cb, const ControlArgs &a, double *hist, Mailbox *mbox, const double *sums_in,
                   hipStream_t stream);
// both of the above in one launch: no collective in between (pv == nullptr), or -- several shards -- with the sum over the
// shards taken through the peer-to-peer slots inside the same launch (p2p.hpp; seq = the exchange's sequence number).
// rows < 0: the shard's own sums are already in `stage`.  do_control = false: the global sums into `stage`, nothing else.
int launch_reduce_control(const double *partials, int64_t rows, int np, double *stage, bool reduce_guarded,
                          ControlBlock *cb, const ControlArgs &a, double *hist, Mailbox *mbox, hipStream_t stream,
                          const P2PView *pv = nullptr, uint32_t seq = 0, bool do_control = true, int silent = 0);
// flag barrier between the shards' streams / end-of-call status exchange / a row of known values through the slots
// (silent: test hook -- 1 = this post is skipped, 2 = it reaches this shard's own slots only)
int launch_p2p_barrier(const P2PView &pv, uint32_t seq, ControlBlock *cb, bool guarded, int silent, hipStream_t stream);
int launch_p2p_commit(const P2PView &pv, uint32_t call, int status, bool wait, ControlBlock *cb, int silent, hipStream_t stream);
int launch_p2p_selftest(const P2PView &pv, uint32_t seq, int np, const double *in, double *out, int *failed, int silent,
                        hipStream_t stream);
// this shard leaves the group of generation `gen`: a word into every peer's slots
int launch_p2p_leave(const P2PView &pv, uint32_t gen, hipStream_t stream);
// self-test of what the transport reads (kernels.hip: k_p2p_pattern_*): buf / len = population buffer 0, 1, rho of this shard
// (write: mode 0 park + pattern, 1 pattern, 2 put back) or of every shard as mapped here (check: out[0] = mismatches)
int p2p_pattern_save_words();
int launch_p2p_pattern_write(double *const buf[3], const int64_t len[3], double *save, uint32_t gen, int round, int rank, int mode,
                             hipStream_t stream);
int launch_p2p_pattern_check(const double *const peer_buf[3][kMaxPeers], const int64_t len[3], uint32_t gen, int round, int world,
                             unsigned int *out, hipStream_t stream);
// K5a: w_i = exp(-sum_j u_ij delta / ubar_j) into the weight row       :126-127
int launch_resample_weights(const ModelDesc &m, PopPtrs pp, const ControlBlock *cb, double n_global, double delta,
                            hipStream_t stream);
// block_sums: weight_scan_doubles(n_global) doubles of scratch shared by K5b and K5c (chunk sums, chunk sums of squares,
// group-end values)
inline int64_t weight_scan_doubles(int64_t n_global) {
  const int64_t nb = (n_global + kScanChunk - 1) / kScanChunk;
  return 2 * nb + nb * (kScanChunk / 16);
}
// K5b: inclusive scan of the global weight vector (last row of every shard's block: a gathered copy [world][rows][cap] or
// the owners' own memory, ShardBlocks) -> cum[n_global];
// totals[0] = sum w, totals[1] = sum w^2                               :129,134
int launch_weight_scan(const ShardBlocks &gathered, int64_t n_global, double *block_sums,
                       double *cum, double *totals, double *totals_host, hipStream_t stream);
// K5 on one shard in four launches: weights fused into the first scan pass; the last pass also packs every particle's
// running sum together with its (theta, u) row, 4 / 2 / 1 particles to a 128-byte line; draw + gather read ONE random line
// per draw, with the moment sums of the resampled population fused in (when (d, s) is an instantiated combination;
// *stats_rows = -1 otherwise: the caller runs the stats pass).  pack: resample_pack_doubles(d + s, n) doubles of scratch.
int launch_resample_local(const ModelDesc &m, PopPtrs src, PopPtrs dst, const ControlBlock *cb, double delta, uint64_t iter,
                          double *block_sums, double *cum, double *totals, double *totals_host, double *pack,
                          double *partials, int64_t *stats_rows, hipStream_t stream);
int64_t resample_pack_doubles(int row_len, int64_t n);
// K5c: n_local categorical draws by inverse CDF + gather of theta and u (not rho)   :129-132
int launch_resample_gather(const ModelDesc &m, const ShardBlocks &gathered, int64_t n_global,
                           const double *cum, const double *block_sums, const double *totals, uint64_t iter, PopPtrs dst,
                           hipStream_t stream);
// K5 on shards: the draws as global source indices (no gather); requests grouped by owner shard; rows served by the
// owner (one contiguous row of d + s doubles per request) and scattered into the requester's next population
int launch_resample_select(const ModelDesc &m, int64_t cap, int64_t n_global, const double *cum, const double *block_sums,
                           const double *totals, uint64_t iter, PopPtrs dst, int64_t *idx_out, hipStream_t stream);
int launch_bucket_count(const int64_t *idx, int64_t n_local, int64_t cap, unsigned long long *counts, hipStream_t stream);
int launch_bucket_scatter(const int64_t *idx, int64_t n_local, int64_t cap, unsigned long long *cursor, double *req,
                          int64_t *slot, hipStream_t stream);
int launch_resample_serve(const double *req, int64_t m, int row_len, PopPtrs src, double *rows_out, hipStream_t stream);
int launch_resample_scatter(const double *rows_in, const int64_t *slot, int64_t n_local, int row_len, PopPtrs dst,
                            hipStream_t stream);
// K2: knots = [0; sorted positives; 1.5 max] from an ascending-sorted column         cdf_estimators.jl:29-33
// meta[0] = number of non-positive entries, meta[1] = 1 if any entry is negative
int launch_cdf_knots(const double *sorted, int64_t n, double *knots, int64_t *meta, hipStream_t stream);
// the index levels of one table: coarse[k] = knots[k << shift], mid[m] = knots[m << 4] (+inf beyond len), and
// +inf written into knots[len, stride)
int launch_cdf_index(double *knots, int64_t len, int64_t stride, int shift, double *coarse, int n_coarse, double *mid,
                     int64_t mid_len, hipStream_t stream);
// compact one statistic's column out of the gathered rho blocks [world][s][cap] into out[n_global]
int launch_compact_column(const ShardBlocks &gathered, int stat, int64_t n_global, double *out, hipStream_t stream);
// ascending sort of n doubles (sort.hip: LSD radix sort, 8 passes of 8 bits); query tmp size with tmp == nullptr
int sort_f64(const double *in, double *out, int64_t n, void *tmp, size_t *tmp_bytes, hipStream_t stream);
// host-simulator mode (SABC_MODEL_HOST): the per-particle body cut at f_dist
int launch_host_prior(const ModelDesc &m, PopPtrs pp, hipStream_t stream);
// thp / aux: device memory (proposals, log prior + log factor), written by the proposal step, read by the accept step.
// thp_host / gate_host / cur_out (optional) / rho_prop / lp_host (optional) are pinned host arrays mapped into the device
// (zero copy): what the host needs of a proposal (the proposal itself, one byte of prior gate, for a host-callback prior
// the current particle) and what it hands back (the distances; for a host-callback prior the two log densities).  The
// proposals of a half batch are signalled chunk by chunk: flag[ch] = seq once chunk ch (`chunk` particles, a multiple of
// kBlock) is complete (`done`: device counters, zero between launches).  The accept step runs per chunk [t_lo, t_lo + t_n).
int launch_host_propose(const ModelDesc &m, const StepArgs &c, const ControlBlock *cb, PopPtrs pp, PartnerView pv,
                        int64_t act_lo, int64_t act_n, double *thp, double *aux, double *thp_host, unsigned char *gate_host,
                        double *cur_out, unsigned int *done, unsigned long long *flag, unsigned long long seq, int64_t chunk,
                        hipStream_t stream);
int launch_host_accept(const ModelDesc &m, const StepArgs &c, const ControlBlock *cb, PopPtrs pp, CdfPtrs cdf, int64_t act_lo,
                       int64_t act_n, int64_t t_lo, int64_t t_n, const double *thp, const double *aux, const double *rho_prop,
                       const double *lp_host, unsigned long long *n_accept, hipStream_t stream);
int launch_stats_rt(const ModelDesc &m, const ControlBlock *cb, PopPtrs pp, double *partials, unsigned long long *n_accept,
                    hipStream_t stream);
// operators
int launch_cdf_eval(const double *knots, int64_t len, const double *q, int64_t m, double *out, hipStream_t stream);
int launch_cdf_apply_matrix(CdfPtrs cdf, int s, const double *rho, int64_t m, double *u_out, hipStream_t stream);
int launch_simulate_batch(const ModelDesc &m, const double *theta, int64_t n, uint64_t pid0, uint64_t iter,
                          double *rho_out, hipStream_t stream, const RtcKernels *rtc = nullptr, const unsigned char *gate = nullptr);
int launch_prior_op(const ModelDesc &m, uint64_t pid0, int64_t n, double *theta, double *lp, hipStream_t stream,
                    const RtcKernels *rtc = nullptr);
int launch_normal_pairs(uint64_t seed, uint64_t pid0, uint32_t purpose, uint64_t iter, uint32_t k, int64_t m, double *out,
                        hipStream_t stream);
int launch_rng_peak(uint64_t seed, int pairs, int64_t n, double *out, hipStream_t stream);
int launch_philox_debug(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k, uint32_t *words,
                        double *normals, hipStream_t stream);

#endif  // !__HIPCC_RTC__

}  // namespace sabc
