// sabc_types.hpp -- plain structs shared by the host engine (pure C++) and the HIP kernels.
// No HIP types here: this header also compiles with g++ for the CPU-side engine tests.
#pragma once
#include <stdint.h>
#include "../../include/sabc_hip.h"

#if defined(__HIPCC__)
#define SABC_TYPES_HD __host__ __device__
#else
#define SABC_TYPES_HD
#endif

namespace sabc {

constexpr int kMaxPara = SABC_MAX_PARA;
constexpr int kMaxStats = SABC_MAX_STATS;
constexpr int kMaxJointPara = SABC_MAX_JOINT_PARA;   // MvNormal prior as data

// What f_dist and prior are, as data (the sabc() arguments of SimulatedAnnealingABC.jl:451).
struct ModelDesc {
  int32_t model_id, d, s, n_model_params;
  double p[SABC_MAX_MODEL_PARAMS];
  int32_t prior_kind[kMaxPara];
  double prior_a[kMaxPara], prior_b[kMaxPara], prior_c[kMaxPara], prior_d[kMaxPara];
  // the normalising constant of logpdf, once on the host: log sigma | log(b - a) | log theta | lgamma(alpha) + alpha log theta
  // | log B(alpha, beta) | log sigma + log(Phi(hi') - Phi(lo'))
  double prior_logc[kMaxPara];
  double prior_k0[kMaxPara], prior_k1[kMaxPara];   // truncated Normal: Phi(lo') (lo' > 0: -Phi(-lo'), mirrored draw), Phi(hi') - Phi(lo') of the standardised bounds
  // prior_joint = 1: MvNormal(prior_a, L L'), L row-major d x d lower; logc = d/2 log(2 pi) + sum log L_kk
  int32_t prior_joint, prior_pad;
  double prior_L[kMaxJointPara * kMaxJointPara];
  double prior_joint_logc;
  uint64_t seed;
};

// Per-launch arguments of the update kernel that do not live in the control block.
struct StepArgs {
  uint64_t iter;                 // global population-update index (RNG counter word)
  int32_t prop_kind, reserved;
  double prop_p0, prop_p1;       // RandomWalk beta | DE gamma0, sigma_gamma | Stretch a
};

// Device-resident control block: everything a population update hands to the next one.  Written
// by the single-lane control kernel (and by the host through the state setters), read by the
// update kernel through scalar loads.  Keeping it in HBM is what lets the host enqueue several
// population updates back to back without reading anything back.
struct ControlBlock {
  double eps[kMaxStats];                 // state.eps (SimulatedAnnealingABC.jl:29)
  double beta[kMaxStats];                // unused since round 4 (the multi-eps root is a function of mean u alone: host_math.hpp); keeps the layout
  double chol[kMaxPara * kMaxPara];      // row-major lower Cholesky factor of Sigma (1-D: sqrt(Sigma))
  double sigma[kMaxPara * kMaxPara];     // RandomWalk.Sigma (proposals.jl:26)
  double pivot[kMaxPara];                // shift of the fused moment sums
  double sums[1 + 2 * kMaxStats + kMaxPara + kMaxPara * (kMaxPara + 1) / 2];   // last global sums
  int64_t n_accept;                      // state.n_accept
  int64_t hist_rows;                     // rows appended to the device history buffer
  int64_t persist_done;                  // k_update_persistent: population updates this launch has completed (-1: it never got there)
  int32_t error;                         // 0 or a SABC_ERR_* raised on the device
  int32_t eps_len;
  int32_t halt;                          // set when the resample test (:340) fires: queued-ahead kernels become no-ops
  int32_t comm_where;                    // SABC_ERR_COMM: which peer-to-peer wait gave up (kind << 24 | peer << 20 | seq & 0xFFFFF)
};

// what the control kernel is asked to do after a reduction
enum : int32_t {
  CTRL_ACCUMULATE = 1,    // n_accept += sums[0]                          (:334)
  CTRL_PROPOSAL = 2,      // Sigma, chol from sums                        (:348, proposals.jl:46-60)
  CTRL_EPSILON = 4,       // eps from sums                                (:350-354)
  CTRL_PIVOT = 8,         // pivot += S / n
  CTRL_HISTORY = 16,      // append (eps, mean u, mean rho)               (:367-372)
  CTRL_CHECK = 32,        // n_accept >= threshold ? set halt and stop here  (:340)
  CTRL_CLEAR_HALT = 64,   // after the host-driven resample
  CTRL_GUARDED = 128,     // no-op while halt is set (a step that was queued ahead of the decision)
  CTRL_KEEP_SUMS = 256    // no reduction preceded this step: work on ControlBlock::sums as they stand (the staging buffer
                          // still holds the LAST step's sums -- for an update step the change of sum(rho), not the sum)
};

struct ControlArgs {
  int32_t mode, d, s, algorithm, prop_kind;
  int32_t rho_is_delta;                  // the rho block of the sums holds the CHANGE of sum(rho) (an update step), not the sum
  double n_global, v, prop_p0;
  int64_t hist_capacity;
  int64_t notify_seq;                    // != 0: post (n_accept, error, halted, seq) to the host mailbox
  double resample_threshold;             // (n_resampling + 1) * resample for CTRL_CHECK
};

// Pinned, host-visible words the control kernel posts to so that the host can learn n_accept
// (the resample test of :340) by polling instead of draining the stream.  Two 8-byte words, each carrying the low 32 bits
// of the step's sequence number in its upper half and written with ONE store: a reader that finds the number in both has
// the payload -- no fence between payload and flag (a system-scope fence in the control kernel is a write-back of the L2:
// 2.6 us of a 9.7 us launch, tools/rc_timing.py).
//   w0 = seq32 << 32 | n_accept bits 0..31
//   w1 = seq32 << 32 | halted << 31 | (-error) << 23 | n_accept bits 32..54
struct Mailbox {
  volatile uint64_t w0, w1;
};
inline constexpr uint64_t kMailboxEmpty = ~0ull;        // what the host initialises the ring with
SABC_TYPES_HD inline void mailbox_pack(int64_t seq, int64_t n_accept, int32_t error, int32_t halted, uint64_t *w0, uint64_t *w1) {
  const uint64_t s32 = (uint64_t)(uint32_t)seq << 32, na = (uint64_t)n_accept;
  *w0 = s32 | (na & 0xFFFFFFFFull);
  *w1 = s32 | ((uint64_t)(halted ? 1 : 0) << 31) | ((uint64_t)((uint32_t)(-error) & 0xFFu) << 23) | ((na >> 32) & 0x7FFFFFull);
}
// false while the step's words have not both arrived
SABC_TYPES_HD inline bool mailbox_unpack(uint64_t w0, uint64_t w1, int64_t seq, int64_t *n_accept, int32_t *error, int32_t *halted) {
  const uint32_t s32 = (uint32_t)seq;
  if ((uint32_t)(w0 >> 32) != s32 || (uint32_t)(w1 >> 32) != s32) return false;
  *n_accept = (int64_t)((w0 & 0xFFFFFFFFull) | ((w1 & 0x7FFFFFull) << 32));
  *error = -(int32_t)((w1 >> 23) & 0xFFu);
  *halted = (int32_t)((w1 >> 31) & 1u);
  return true;
}
constexpr int kMailboxRing = 8;          // slot = seq % kMailboxRing; the host lags by at most 2 steps

// Layout of the fused per-update sums ("partials"): one row of `np` doubles.
//   [0]                n_accept
//   [1, 1+s)           sum_i u_ij
//   [1+s, 1+2s)        sum_i rho_ij  (stats passes)  |  sum over accepted i of rho'_ij - rho_ij  (update steps)
//   [1+2s, 1+2s+d)     sum_i (theta_ik - pivot_k)
//   [.., +d(d+1)/2)    sum_i (theta_ik - pivot_k)(theta_il - pivot_l), l <= k, row-major lower
inline constexpr int n_partials(int d, int s) { return 1 + 2 * s + d + d * (d + 1) / 2; }
constexpr int kMaxPartials = 1 + 2 * kMaxStats + kMaxPara + kMaxPara * (kMaxPara + 1) / 2;

// shards whose memory a kernel can address directly: the GPUs of one node (peer-mapped over xGMI, csrc/p2p.hpp)
constexpr int kMaxPeers = 8;

// Where a DifferentialEvolution / StretchMove partner (proposals.jl:105-106,141) is read from:
// the inactive halves of all shards.  direct == 0: one gathered copy, laid out [world][rows][cap] from `base`;
// direct != 0: shard r's own theta block in the OWNER's HBM, peer[r] (row stride cap, the inactive half at off_*).
struct PartnerView {
  const double *base;
  int64_t rank_stride;           // doubles between consecutive shards
  int64_t cap;                   // row stride inside a shard
  int64_t m_full, m_last;        // inactive-half size of a full shard / of the last shard
  int64_t off_full, off_last;    // where the inactive half starts inside a shard
  int64_t m_total;
  int32_t world, direct;
  const double *peer[kMaxPeers];
};

// Where shard r's block [rows][cap] of an array that exists once per shard (population, rho) is read from:
// a gathered copy [world][rows][cap] (direct == 0) or every owner's own memory (direct != 0, world <= kMaxPeers).
struct ShardBlocks {
  const double *flat;
  const double *peer[kMaxPeers];
  int64_t cap;
  int32_t rows, world, direct, reserved;
};
inline ShardBlocks flat_blocks(const double *g, int rows, int64_t cap, int world) {
  ShardBlocks b;
  b.flat = g;
  for (int r = 0; r < kMaxPeers; ++r) b.peer[r] = nullptr;
  b.cap = cap; b.rows = rows; b.world = world; b.direct = 0; b.reserved = 0;
  return b;
}

// Shard geometry: contiguous blocks of `cap` global ids per rank.
struct Shard {
  int64_t n_global, cap, n_local, gid0;
  int32_t rank, world;
};

inline Shard make_shard(int64_t n_global, int rank, int world) {
  Shard sh;
  sh.n_global = n_global; sh.rank = rank; sh.world = world;
  sh.cap = (n_global + world - 1) / world;
  sh.gid0 = (int64_t)rank * sh.cap;
  int64_t hi = sh.gid0 + sh.cap;
  if (hi > n_global) hi = n_global;
  sh.n_local = hi > sh.gid0 ? hi - sh.gid0 : 0;
  return sh;
}
inline int64_t shard_n_local(const Shard &sh, int r) {
  int64_t lo = (int64_t)r * sh.cap, hi = lo + sh.cap;
  if (hi > sh.n_global) hi = sh.n_global;
  return hi > lo ? hi - lo : 0;
}

}  // namespace sabc
