// update_kernel.hpp -- the per-particle kernels as templates over the simulator: K4 `k_update` (propose -> prior gate
// -> simulate -> distance -> ECDF -> annealed MH accept -> store + fused block sums, SimulatedAnnealingABC.jl:308-331),
// K1 `k_prior_simulate` (:172-179), `k_simulate_batch` and `k_stats`.  Device code only, no launchers: kernels.hip
// instantiates them for the built-in simulators at build time, and rtc.cpp compiles this very header with hipRTC for a
// simulator the user supplies as HIP source (SABC_MODEL_USER) -- same kernel, same reductions, same RNG streams.
#pragma once
#include "device_models.hpp"
#include "kernels.hpp"

namespace sabc {

// ------------------------------------------------------------------------------------------
// block reduction of NP per-lane values: wave shuffles, then LDS across the 4 wavefronts
// ------------------------------------------------------------------------------------------
// FIRST: the smallest shuffle distance that carries anything (k_update_persistent with a team of lanes per particle: only every
// FIRST-th lane holds terms, the others exact zeros -- adding them changes no bit)
template <int NP, int BLOCK = kBlock, int FIRST = 1>
__device__ __forceinline__ void block_reduce_store(const double (&acc)[NP], double *__restrict__ out) {
  constexpr int NW = BLOCK / 64;
  __shared__ double sm[NW][NP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // step-major order: the NP shuffles of one step are independent and go out back to back (one LDS round trip per
  // step instead of one per step AND column -- 6 instead of 6 NP dependent trips at the end of every wave's life)
  double v[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) v[c] = acc[c];
#pragma unroll
  for (int off = 32; off >= FIRST; off >>= 1) {
    double t[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) t[c] = __shfl_down(v[c], off, 64);
#pragma unroll
    for (int c = 0; c < NP; ++c) v[c] += t[c];
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NP; ++c) sm[wave][c] = v[c];
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    const int c = threadIdx.x;
    double a = sm[0][c];
#pragma unroll
    for (int w = 1; w < NW; ++w) a += sm[w][c];
    out[c] = a;
  }
}

template <int D, int S>
__device__ __forceinline__ void moment_terms(const double *__restrict__ pivot, bool accepted, const double *th,
                                             const double *u, const double *rho, double (&acc)[n_partials(D, S)]) {
  acc[0] = accepted ? 1.0 : 0.0;
#pragma unroll
  for (int j = 0; j < S; ++j) { acc[1 + j] = u[j]; acc[1 + S + j] = rho[j]; }
  double dk[D];
#pragma unroll
  for (int k = 0; k < D; ++k) { dk[k] = th[k] - pivot[k]; acc[1 + 2 * S + k] = dk[k]; }
  int q = 1 + 2 * S + D;
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int l = 0; l <= k; ++l) acc[q++] = dk[k] * dk[l];
}

__device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b) { return __umul64hi(a, b); }

// (gid / cap, gid % cap) for 0 <= gid < 2^53 without a 64-bit integer division (~100 instructions on this ISA): one
// shard needs none (gid < cap), several take the f64 quotient, which is off by at most one, and correct it
__device__ __forceinline__ void split_index(int64_t gid, int64_t cap, int64_t &r, int64_t &o) {
  if (gid < cap) { r = 0; o = gid; return; }
  r = (int64_t)((double)gid / (double)cap);
  if (r * cap > gid) r -= 1;
  else if ((r + 1) * cap <= gid) r += 1;
  o = gid - r * cap;
}

__device__ __forceinline__ const double *partner_ptr(const PartnerView &pv, uint64_t j) {
  if (pv.world == 1) return pv.base + pv.off_last + (int64_t)j;        // one shard: no 64-bit division (uniform branch)
  // shard index j / m_full without a 64-bit integer division: j < 2^53, so the f64 quotient is off by at most one
  int64_t r = (int64_t)((double)j / (double)pv.m_full);
  if (r * pv.m_full > (int64_t)j) r -= 1;
  else if ((r + 1) * pv.m_full <= (int64_t)j) r += 1;
  if (r > pv.world - 1) r = pv.world - 1;
  const int64_t o = (int64_t)j - r * pv.m_full;
  const int64_t off = (r == pv.world - 1) ? pv.off_last : pv.off_full;
  // direct: shard r's theta block in its owner's HBM (peer-mapped, p2p.hpp); else its part of the gathered copy
  const double *b = pv.direct ? pv.peer[r] : pv.base + r * pv.rank_stride;
  return b + off + o;
}

// block [rows][cap] of shard r
__device__ __forceinline__ const double *shard_block(const ShardBlocks &b, int64_t r) {
  return b.direct ? b.peer[r] : b.flat + r * (int64_t)b.rows * b.cap;
}

// ------------------------------------------------------------------------------------------
// K4: propose -> prior gate -> simulate -> distance -> ECDF -> annealed MH accept -> store,
//     + fused block partials.   SimulatedAnnealingABC.jl:308-331
// ------------------------------------------------------------------------------------------
// the coarse level of the ECDF tables into LDS: all reads first, then the LDS writes -- one trip to memory at the front of
// the workgroup's life instead of one per pass
template <int S>
__device__ __forceinline__ void load_coarse_index(const CdfPtrs &cdf, double (&cidx)[S][cdf_coarse_entries(S)]) {
  constexpr int kUpdateBlock = update_block_threads(S), kCoarse = cdf_coarse_entries(S);
  static_assert((S * kCoarse / 2) % kUpdateBlock == 0, "whole passes of 16 bytes per thread");
  constexpr int kPasses = S * kCoarse / 2 / kUpdateBlock;
  const double2 *src = reinterpret_cast<const double2 *>(cdf.coarse);
  double2 *dst = reinterpret_cast<double2 *>(&cidx[0][0]);
  double2 tmp[kPasses];
#pragma unroll
  for (int q = 0; q < kPasses; ++q) tmp[q] = src[threadIdx.x + q * kUpdateBlock];
#pragma unroll
  for (int q = 0; q < kPasses; ++q) dst[threadIdx.x + q * kUpdateBlock] = tmp[q];
}

// -DSABC_PERSIST_TRACE (tools/persist_trace.py, a variant library -- not the product): workgroup 0's first lane stamps the
// phases of a population update (row iter % 64) with the 100 MHz wall clock
#ifdef SABC_PERSIST_TRACE
__device__ unsigned long long g_persist_trace[64 * 16];
#define SABC_TRACE(iter, i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_persist_trace[((iter) & 63) * 16 + (i)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define SABC_TRACE(iter, i) do { } while (0)
#endif

// The per-particle body, :308-331, for local particle li (global id gid) at population update `iter`; eps, the Cholesky
// factor and the pivot come from *cb -- the control block in memory (k_update: scalar loads) or a workgroup's LDS copy of it
// (k_update_persistent); the particle's moment terms go to acc.
// PAST_CACHES (k_update_persistent with DifferentialEvolution / StretchMove): the workgroups of ONE launch read each other's
// particles update after update, and the per-XCD L2s are not coherent with each other -- partners are read and accepted
// particles written with agent-scope accesses, which go past the caches, instead of a write-back and an invalidate of the L2
// at every grid barrier.  (A workgroup's own particles are only ever written by itself: plain loads.)
template <bool PAST_CACHES>
__device__ __forceinline__ double particle_load(const double *p) {
  return PAST_CACHES ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool PAST_CACHES>
__device__ __forceinline__ void particle_store(double *p, double v) {
  if (PAST_CACHES) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// LANES = 4 | 16 (k_update_persistent on an under-filled device): the lanes of a team -- a quad, a row of 16 -- run THIS particle
// together -- the same loads, the same arithmetic, the same decisions -- and share the generator's work (device_rng.hpp:
// NormalStream, coop); the team's first lane alone writes the particle back and reports its moment terms.
// LATENCY (k_update_persistent): the wave has its SIMD to itself -- the ECDF lookups of the S statistics step together
// (device_models.hpp: cdf_apply_3level_lockstep) instead of one after the other.
// The body in two halves, so that k_update_persistent can run the first of the NEXT update while one lane still solves for this
// update's epsilon (persistent_kernel.hpp: the control wave): what a particle's proposal and simulation need of the algorithm's
// state is the Cholesky factor alone (RandomWalk) -- epsilon and the pivot only enter where the acceptance is decided.
template <int D, int S>
struct ParticleDraft {
  double th[D], u[S];            // the particle as it stands
  double thp[D], rp[S];          // its proposal and the proposal's simulated distances (0 outside the prior's support)
  double logf, lpp;              // log of the proposal's density ratio (StretchMove), log prior of the proposal
};

// loads, proposal (:311), prior gate and simulation (:314-315)
template <int MODEL, int D, int S, int PROP, bool PAST_CACHES, int LANES, class CB>
__device__ __forceinline__ void update_particle_draft(const ModelDesc &m, const uint64_t iter, const double prop_p0, const double prop_p1,
                                                      const CB *__restrict__ cb, const PopPtrs &pp, const PartnerView &pv, const int64_t li,
                                                      const uint64_t gid, ParticleDraft<D, S> &q) {
  // rho is NOT read here: an update step reports the CHANGE of sum(rho) (rho' - rho of the accepted particles, read where
  // they are overwritten); the control step adds it to the running sum (ControlArgs::rho_is_delta).  8 s n bytes less
  // read per launch: the old distance of a particle that is not accepted is never needed.
#pragma unroll
  for (int k = 0; k < D; ++k) q.th[k] = pp.pop[(int64_t)k * pp.cap + li];
#pragma unroll
  for (int j = 0; j < S; ++j) q.u[j] = pp.pop[(int64_t)(D + j) * pp.cap + li];

  SABC_TRACE(iter, 6);
  // ---- proposal (:311) ----
  q.logf = 0.0;
  if (PROP == SABC_PROP_RANDOMWALK) {            // proposals.jl:40-43,52-55: theta + L z
    NormalStream ns(m.seed, gid, PURPOSE_PROP, iter, LANES > 1 ? LANES : 0);
    double z[D];
#pragma unroll
    for (int k = 0; k < D; ++k) z[k] = ns.next();
#pragma unroll
    for (int k = 0; k < D; ++k) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l <= k; ++l) a += cb->chol[k * D + l] * z[l];
      q.thp[k] = q.th[k] + a;
    }
  } else if (PROP == SABC_PROP_DIFFEVO) {        // proposals.jl:101-114
    uint64_t i1 = 0, i2 = 0;
    for (uint32_t a = 0;; ++a) {                 // :103-107, redraw both until distinct
      const u32x4 w = stream_block(m.seed, gid, PURPOSE_PROP, iter, a);
      i1 = mulhi64(pack64(w.x, w.y), (uint64_t)pv.m_total);
      i2 = mulhi64(pack64(w.z, w.w), (uint64_t)pv.m_total);
      if (i1 != i2 || a > 64u) break;
    }
    double z0, z1;
    box_muller(stream_block(m.seed, gid, PURPOSE_PROP2, iter, 0), z0, z1);
    const double gamma = prop_p0 * (1.0 + prop_p1 * z0);      // :110
    const double *p1 = partner_ptr(pv, i1), *p2 = partner_ptr(pv, i2);
#pragma unroll
    for (int k = 0; k < D; ++k)
      q.thp[k] = q.th[k] + gamma * (particle_load<PAST_CACHES>(p1 + (int64_t)k * pv.cap) - particle_load<PAST_CACHES>(p2 + (int64_t)k * pv.cap));
  } else {                                       // StretchMove, proposals.jl:137-148
    const u32x4 w = stream_block(m.seed, gid, PURPOSE_PROP, iter, 0);
    const uint64_t ip = mulhi64(pack64(w.x, w.y), (uint64_t)pv.m_total);   // :141
    const double U = u52(w.z, w.w);
    const double a = prop_p0;
    const double tt = (a - 1.0) * U + 1.0;
    const double z = tt * tt / a;                                          // :144
    const double *p = partner_ptr(pv, ip);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double pk = particle_load<PAST_CACHES>(p + (int64_t)k * pv.cap);
      q.thp[k] = pk + z * (q.th[k] - pk);                                  // :147
    }
    q.logf = log(z) * (double)(D - 1);                                     // :146
  }

  SABC_TRACE(iter, 7);
  // ---- the prior's gate and the simulation (:314-315) ----
  q.lpp = prior_logpdf<D>(m, q.thp);
#pragma unroll
  for (int j = 0; j < S; ++j) q.rp[j] = 0.0;
  if (q.lpp > -INFINITY) {
    if (LANES > 1) Sim<MODEL, D, S>::run(m, q.thp, gid, iter, q.rp, LANES);    // :315
    else Sim<MODEL, D, S>::run(m, q.thp, gid, iter, q.rp);
  }
  SABC_TRACE(iter, 8);
}

// ECDF transform, acceptance probability (:316-322), accept / store (:324-329), the particle's moment terms
template <int D, int S, bool PAST_CACHES, int LANES, bool LATENCY, class CB>
__device__ __forceinline__ void update_particle_decide(const ModelDesc &m, const uint64_t iter, const CB *__restrict__ cb, const PopPtrs &pp,
                                                       const CdfPtrs &cdf, const double (&cidx)[S][cdf_coarse_entries(S)], const int64_t li,
                                                       const uint64_t gid, ParticleDraft<D, S> &q, double (&acc)[n_partials(D, S)]) {
  constexpr int kCoarse = cdf_coarse_entries(S);
  double log_accept = -INFINITY;
  double up[S], drho[S];
#pragma unroll
  for (int j = 0; j < S; ++j) { up[j] = 0.0; drho[j] = 0.0; }
  if (q.lpp > -INFINITY) {
    double a = 0.0;
    if (LATENCY && S >= 2) cdf_apply_3level_lockstep<S, kCoarse>(cdf, cidx, q.rp, up);                       // :316
#pragma unroll
    for (int j = 0; j < S; ++j) {
      if (LATENCY && S == 1 && cdf.shift[0] == 0)
        up[j] = cdf_apply_lds<kCoarse>(cidx[j], cdf.len[j], q.rp[j]);                                         // :316
      else if (!(LATENCY && S >= 2))
        up[j] = cdf_apply_3level<kCoarse>(cdf.knots + (int64_t)j * cdf.stride, cdf.len[j], cdf.shift[j], cidx[j],
                                 cdf.mid + (int64_t)j * cdf.mid_stride, q.rp[j]);                             // :316
      const double e = (cb->eps_len == 1) ? cb->eps[0] : cb->eps[j];
      a += (q.u[j] - up[j]) / e;                                           // :319
    }
    log_accept = q.lpp - prior_logpdf<D>(m, q.th) + a + q.logf;            // :318-319
  }

  SABC_TRACE(iter, 9);
  // ---- accept / store (:324-329) ----
  const u32x4 wa = stream_block(m.seed, gid, PURPOSE_ACCEPT, iter, 0);
  const bool accepted = -0.5 * neg2_log_tab(u52(wa.x, wa.y)) < log_accept;      // log(U) < log alpha, :324
  const bool writer = LANES == 1 || (threadIdx.x & (LANES - 1)) == 0;
  if (accepted) {
#pragma unroll
    for (int j = 0; j < S; ++j) drho[j] = q.rp[j] - pp.rho[(int64_t)j * pp.cap + li];
#pragma unroll
    for (int k = 0; k < D; ++k) q.th[k] = q.thp[k];
#pragma unroll
    for (int j = 0; j < S; ++j) q.u[j] = up[j];
    if (writer) {
#pragma unroll
      for (int k = 0; k < D; ++k) particle_store<PAST_CACHES>(pp.pop + (int64_t)k * pp.cap + li, q.thp[k]);   // (partners read theta)
#pragma unroll
      for (int j = 0; j < S; ++j) {
        pp.pop[(int64_t)(D + j) * pp.cap + li] = up[j];
        pp.rho[(int64_t)j * pp.cap + li] = q.rp[j];
      }
    }
  }
  SABC_TRACE(iter, 10);
  moment_terms<D, S>(cb->pivot, accepted, q.th, q.u, drho, acc);
  if (!writer) {
#pragma unroll
    for (int q2 = 0; q2 < n_partials(D, S); ++q2) acc[q2] = 0.0;
  }
}

template <int MODEL, int D, int S, int PROP, bool PAST_CACHES = false, int LANES = 1, bool LATENCY = false, class CB>
__device__ __forceinline__ void update_particle(const ModelDesc &m, const uint64_t iter, const double prop_p0, const double prop_p1,
                                                const CB *__restrict__ cb, const PopPtrs &pp, const CdfPtrs &cdf, const PartnerView &pv,
                                                const double (&cidx)[S][cdf_coarse_entries(S)], const int64_t li, const uint64_t gid,
                                                double (&acc)[n_partials(D, S)]) {
  ParticleDraft<D, S> q;
  update_particle_draft<MODEL, D, S, PROP, PAST_CACHES, LANES>(m, iter, prop_p0, prop_p1, cb, pp, pv, li, gid, q);
  update_particle_decide<D, S, PAST_CACHES, LANES, LATENCY>(m, iter, cb, pp, cdf, cidx, li, gid, q, acc);
}

template <int MODEL, int D, int S, int PROP>
__global__ void __launch_bounds__(update_block_threads(S), update_min_waves(S))
k_update(const ModelDesc m, const StepArgs c, const ControlBlock *__restrict__ cb, const PopPtrs pp, const CdfPtrs cdf,
         const PartnerView pv, const int64_t act_lo, const int64_t act_n, double *__restrict__ partials) {
  constexpr int NP = n_partials(D, S);
  if (cb->halt) return;                    // queued ahead of a resample decision that fired (uniform)
  rng_tables_load();
  constexpr int kUpdateBlock = update_block_threads(S), kCoarse = cdf_coarse_entries(S);
  __shared__ double cidx[S][kCoarse];      // coarse level of the ECDF tables, 8 or 16 KB per statistic
  load_coarse_index<S>(cdf, cidx);
  __syncthreads();                         // publishes both the generator tables and the index
  double acc[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) acc[q] = 0.0;

  const int64_t t = (int64_t)blockIdx.x * kUpdateBlock + threadIdx.x;
  if (t < act_n) {
    const int64_t li = act_lo + t;
    update_particle<MODEL, D, S, PROP>(m, c.iter, c.prop_p0, c.prop_p1, cb, pp, cdf, pv, cidx, li, (uint64_t)(pp.gid0 + li), acc);
  }
  block_reduce_store<NP, kUpdateBlock>(acc, partials + (int64_t)blockIdx.x * NP);
}

// moment sums of the shard as it stands (after a resample, or at update_population! entry :284)
template <int D, int S>
__global__ void __launch_bounds__(kBlock)
k_stats(const ControlBlock *__restrict__ cb, const PopPtrs pp, double *__restrict__ partials) {
  constexpr int NP = n_partials(D, S);
  double acc[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) acc[q] = 0.0;
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (li < pp.n_local) {
    double th[D], u[S], rho[S];
#pragma unroll
    for (int k = 0; k < D; ++k) th[k] = pp.pop[(int64_t)k * pp.cap + li];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      u[j] = pp.pop[(int64_t)(D + j) * pp.cap + li];
      rho[j] = pp.rho[(int64_t)j * pp.cap + li];
    }
    moment_terms<D, S>(cb->pivot, false, th, u, rho, acc);
  }
  block_reduce_store<NP>(acc, partials + (int64_t)blockIdx.x * NP);
}

// ------------------------------------------------------------------------------------------
// K1: prior sample + simulate (SimulatedAnnealingABC.jl:172-179), iteration 0
// ------------------------------------------------------------------------------------------
template <int MODEL, int D, int S>
__global__ void __launch_bounds__(kBlock) k_prior_simulate(const ModelDesc m, const PopPtrs pp) {
  rng_tables_init();
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (li >= pp.n_local) return;
  const uint64_t gid = (uint64_t)(pp.gid0 + li);
  double th[D], rho[S];
  prior_sample<D>(m, gid, th);
  Sim<MODEL, D, S>::run(m, th, gid, 0, rho);
#pragma unroll
  for (int k = 0; k < D; ++k) pp.pop[(int64_t)k * pp.cap + li] = th[k];
#pragma unroll
  for (int j = 0; j < S; ++j) pp.rho[(int64_t)j * pp.cap + li] = rho[j];
}

// rand(prior) and its log density for particle ids pid0.. (sabc_op_prior), for a prior that lives in the run-time compiled
// unit (prior_joint = 3); the built-in families go through kernels.hip's run-time-d version
template <int D>
__global__ void __launch_bounds__(kBlock)
k_prior_op_t(const ModelDesc m, const uint64_t pid0, const int64_t n, double *__restrict__ theta, double *__restrict__ lp) {
  rng_tables_init();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double th[D];
  prior_sample<D>(m, pid0 + (uint64_t)i, th);
#pragma unroll
  for (int k = 0; k < D; ++k) theta[(int64_t)k * n + i] = th[k];
  lp[i] = prior_logpdf<D>(m, th);
}

// gate (optional, one byte per row): 0 = this theta lies outside the prior's support (SimulatedAnnealingABC.jl:314-315): it is
// not simulated, its distances are written as 0 and never looked at
template <int MODEL, int D, int S>
__global__ void __launch_bounds__(kBlock)
k_simulate_batch(const ModelDesc m, const double *__restrict__ theta, const int64_t n, const uint64_t pid0,
                 const uint64_t iter, double *__restrict__ rho_out, const unsigned char *__restrict__ gate) {
  rng_tables_init();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double th[D], rho[S];
#pragma unroll
  for (int j = 0; j < S; ++j) rho[j] = 0.0;
  if (!gate || gate[i]) {
#pragma unroll
    for (int k = 0; k < D; ++k) th[k] = theta[(int64_t)k * n + i];
    Sim<MODEL, D, S>::run(m, th, pid0 + (uint64_t)i, iter, rho);
  }
#pragma unroll
  for (int j = 0; j < S; ++j) rho_out[(int64_t)j * n + i] = rho[j];
}

}  // namespace sabc
