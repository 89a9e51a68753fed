// sabc_types.hpp -- plain structs shared by the host engine (pure C++) and the HIP kernels.
// No HIP types here: this header also compiles with g++ for the CPU-side engine tests.
#pragma once
#include <stdint.h>
#include "../../include/sabc_hip.h"

namespace sabc {

constexpr int kMaxPara = SABC_MAX_PARA;
constexpr int kMaxStats = SABC_MAX_STATS;

// What f_dist and prior are, as data (the sabc() arguments of SimulatedAnnealingABC.jl:451).
struct ModelDesc {
  int32_t model_id, d, s, n_model_params;
  double p[SABC_MAX_MODEL_PARAMS];
  int32_t prior_kind[kMaxPara];
  double prior_a[kMaxPara], prior_b[kMaxPara];
  uint64_t seed;
};

// Everything that is constant during one population update (SimulatedAnnealingABC.jl:294-354):
// eps (:350-354), the proposal's parameters and, for RandomWalk, the Cholesky factor of
// Sigma (proposals.jl:42,47) hoisted from per-particle to per-update.
struct StepCtrl {
  uint64_t iter;                 // global population-update index (RNG counter word)
  int32_t eps_len;               // 1 (:single_eps) or s (:multi_eps)
  int32_t prop_kind;
  double eps[kMaxStats];
  double prop_p0, prop_p1;
  double chol[kMaxPara * kMaxPara];   // row-major lower factor (1-D: sqrt(Sigma))
  double pivot[kMaxPara];             // shift used by the fused moment sums
};

// Layout of the fused per-update sums ("partials"): one row of `np` doubles.
//   [0]                n_accept
//   [1, 1+s)           sum_i u_ij
//   [1+s, 1+2s)        sum_i rho_ij
//   [1+2s, 1+2s+d)     sum_i (theta_ik - pivot_k)
//   [.., +d(d+1)/2)    sum_i (theta_ik - pivot_k)(theta_il - pivot_l), l <= k, row-major lower
inline constexpr int n_partials(int d, int s) { return 1 + 2 * s + d + d * (d + 1) / 2; }
constexpr int kMaxPartials = 1 + 2 * kMaxStats + kMaxPara + kMaxPara * (kMaxPara + 1) / 2;

// Where a DifferentialEvolution / StretchMove partner (proposals.jl:105-106,141) is read from:
// the inactive halves of all shards, laid out [world][rows][cap].
struct PartnerView {
  const double *base;
  int64_t rank_stride;           // doubles between consecutive shards
  int64_t cap;                   // row stride inside a shard
  int64_t m_full, m_last;        // inactive-half size of a full shard / of the last shard
  int64_t off_full, off_last;    // where the inactive half starts inside a shard
  int64_t m_total;
  int32_t world, reserved;
};

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
