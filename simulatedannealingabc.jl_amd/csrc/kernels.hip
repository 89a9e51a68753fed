// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the SABC particle-population update loop.
//
// Design notes (DESIGN.md has the long form):
//  * one wavefront lane per particle; particle state is SoA ([row][cap]) so that the 64 lanes of a
//    wave read/write 512 contiguous bytes per row;
//  * no dense contraction anywhere -> MFMA deliberately unused; the kernel is bound by Philox
//    integer multiplies and f64 log/sqrt/sincospi, not by HBM (40 algorithmic bytes per
//    particle-simulation for d = s = 1);
//  * the reductions the reference does in separate passes (n_accept :334, mean(u) :353, column
//    means :369-370, cov(population) proposals.jl:47,59) are fused into the update kernel:
//    wave shuffles -> LDS across the 4 waves of a block -> one partial row per block, summed
//    later in a fixed order (bitwise reproducible for a given grid).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdlib>
#include <cstring>
#ifdef SABC_RC_TIMING
// A/B instrumentation (tools/rc_timing.py; never in the shipped build): where the time of a reduce-and-control launch goes,
// in ticks of the constant 100 MHz clock, summed over launches.  [0] launches, [1..] phases; marks are set by thread 0.
__device__ unsigned long long g_rc_ticks[16];
__device__ __forceinline__ void rc_mark(int i) {
  static __shared__ unsigned long long last;
  if (threadIdx.x != 0) return;
  const unsigned long long t = wall_clock64();
  if (i == 0) atomicAdd(&g_rc_ticks[0], 1ull); else atomicAdd(&g_rc_ticks[i], t - last);
  last = t;
}
#define SABC_CTRL_MARK(i) rc_mark(i)
#endif
#include "control.hpp"
#include "p2p.hpp"
#include "persistent_kernel.hpp"
#include "update_kernel.hpp"

namespace sabc {

// ------------------------------------------------------------------------------------------
// g-and-k (BASELINE config 4): the SIMULATION is wave-per-particle (128 draws sorted across the lanes); a wave owns
// kGkParticlesPerWave consecutive particles (a block of 4 waves 4 x that), does their proposals, prior gates, ECDF
// lookups and accept steps one particle per lane, and simulates them one after the other in between.
// ------------------------------------------------------------------------------------------
constexpr int kGkD = 4, kGkS = 4;
constexpr int kGkPerBlock = (kBlock / 64) * kGkParticlesPerWave;
// k_update_gk: every wave takes kGkReps groups of kGkParticlesPerWave particles in turn and the workgroup writes ONE partial
// row for all of them.  Measured at n = 1e6 with groups of 16 (tools/exp_ab2.sh, three runs each): 1 group 663 us, 2 groups
// 786 us, 4 groups 812 us -- the loop around the phases costs the register allocator 240 more bytes of scratch and 16 more
// SGPR reloads per particle.  So: one group, and the group itself grew to 64 (device_models.hpp) -- the lane-parallel
// phases then run with all lanes busy instead of being repeated.
#ifndef SABC_GK_REPS
#define SABC_GK_REPS 1
#endif
constexpr int kGkReps = SABC_GK_REPS;
#ifndef SABC_GK_PAIR
#define SABC_GK_PAIR 1
#endif
#ifndef SABC_GK_ROWS4
#define SABC_GK_ROWS4 1
#endif
constexpr int kGkUpdatePerBlock = kGkPerBlock * kGkReps;

// per-wave staging of what phase 1 (propose + simulate) hands to phase 2 (ECDF) and 3 (accept)
struct GkStage {
  double thp[kGkParticlesPerWave][kGkD];
  double rp[kGkParticlesPerWave][kGkS];
  double up[kGkParticlesPerWave][kGkS];
  double lpp[kGkParticlesPerWave];
  double logf[kGkParticlesPerWave];
};

// 4 workgroups per CU (<= 128 VGPRs, 68 B of scratch outside the sort): 850 -> 755 us at n = 1e6 against 3 per CU (144 VGPRs,
// no scratch) -- the sort waits on lane exchanges, so the extra wave pays; 5 per CU spills inside the loop (1030 us)
// ROWS4: every wanted rank is a multiple of 16 (the host looks: launch_update) -- the simulations run four particles at a time
// on the network of gk_simulate_rows4 only; the two-values-per-lane network stays out of this instantiation (and its
// registers with it: SABC_GK_ROWS4_WAVES workgroups per CU)
#ifndef SABC_GK_ROWS4_WAVES
#define SABC_GK_ROWS4_WAVES 4
#endif
template <int PROP, bool ROWS4>
__global__ void __launch_bounds__(kBlock, ROWS4 ? SABC_GK_ROWS4_WAVES : 4)
k_update_gk(const ModelDesc m, const StepArgs c, const ControlBlock *__restrict__ cb, const PopPtrs pp, const CdfPtrs cdf,
            const PartnerView pv, const int64_t act_lo, const int64_t act_n, double *__restrict__ partials) {
  constexpr int D = kGkD, S = kGkS, NP = n_partials(D, S), PW = kGkParticlesPerWave;
  static_assert(S == 4 && (PW * S) % 64 == 0 && PW <= 64, "phase 2 maps one (particle, statistic) pair to each lane, PW S / 64 times");
  if (cb->halt) return;                    // queued ahead of a resample decision that fired (uniform)
  rng_tables_init();
  __shared__ GkStage stage[kBlock / 64];
  __shared__ double red[kBlock / 64][NP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GkStage &st = stage[wave];
  if (lane < NP) red[wave][lane] = 0.0;
  for (int rep = 0; rep < kGkReps; ++rep) {
  const int64_t t0 = (int64_t)blockIdx.x * kGkUpdatePerBlock + (wave * kGkReps + rep) * PW;
  if (t0 >= act_n) break;                  // uniform over the wave
  // the wave owns particles t0 .. t0+PW-1; in the scalar phases (1a, 3) lane i < PW handles particle t0+i
  const int64_t t_mine = t0 + lane;
  const bool mine = lane < PW && t_mine < act_n;
  const int64_t li = act_lo + t_mine;
  const uint64_t gid = (uint64_t)(pp.gid0 + li);

  // ---- phase 1a, lane-parallel over the wave's particles: proposal (:311) and prior gate (:314)
  if (mine) {
    double th[D], thp[D];
#pragma unroll
    for (int k = 0; k < D; ++k) th[k] = pp.pop[(int64_t)k * pp.cap + li];
    double logf = 0.0;
    if (PROP == SABC_PROP_RANDOMWALK) {
      NormalStream ns(m.seed, gid, PURPOSE_PROP, c.iter);
      double z[D];
#pragma unroll
      for (int k = 0; k < D; ++k) z[k] = ns.next();
#pragma unroll
      for (int k = 0; k < D; ++k) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l <= k; ++l) a += cb->chol[k * D + l] * z[l];
        thp[k] = th[k] + a;
      }
    } else if (PROP == SABC_PROP_DIFFEVO) {
      uint64_t i1 = 0, i2 = 0;
      for (uint32_t a = 0;; ++a) {
        const u32x4 w = stream_block(m.seed, gid, PURPOSE_PROP, c.iter, a);
        i1 = mulhi64(pack64(w.x, w.y), (uint64_t)pv.m_total);
        i2 = mulhi64(pack64(w.z, w.w), (uint64_t)pv.m_total);
        if (i1 != i2 || a > 64u) break;
      }
      double z0, z1;
      box_muller(stream_block(m.seed, gid, PURPOSE_PROP2, c.iter, 0), z0, z1);
      const double gamma = c.prop_p0 * (1.0 + c.prop_p1 * z0);
      const double *p1 = partner_ptr(pv, i1), *p2 = partner_ptr(pv, i2);
#pragma unroll
      for (int k = 0; k < D; ++k) thp[k] = th[k] + gamma * (p1[(int64_t)k * pv.cap] - p2[(int64_t)k * pv.cap]);
    } else {
      const u32x4 w = stream_block(m.seed, gid, PURPOSE_PROP, c.iter, 0);
      const uint64_t ip = mulhi64(pack64(w.x, w.y), (uint64_t)pv.m_total);
      const double U = u52(w.z, w.w);
      const double a = c.prop_p0;
      const double tt = (a - 1.0) * U + 1.0;
      const double z = tt * tt / a;
      const double *p = partner_ptr(pv, ip);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double pk = p[(int64_t)k * pv.cap];
        thp[k] = pk + z * (th[k] - pk);
      }
      logf = log(z) * (double)(D - 1);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) st.thp[lane][k] = thp[k];
    st.lpp[lane] = prior_logpdf<D>(m, thp);
    st.logf[lane] = logf;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- phase 1b, the whole wave on the simulations (:315), lane l draws 2 of a particle's 128; TWO particles at a time
  // (SABC_GK_PAIR): their sorting networks are independent, so the lane exchanges of one overlap the selects of the other.
  // Particles outside the prior's support are not simulated (:314): the wave walks the set bits of `todo`.
  {
    unsigned long long todo = __ballot(mine && st.lpp[lane] > -INFINITY);
    // wanted ranks that are all multiples of 16 (BASELINE config 4): FOUR particles at a time, one per row of 16 lanes, eight
    // values per lane -- 15 of the network's 24 steps stay inside the lane (device_models.hpp: gk_simulate_rows4)
    while (ROWS4 && todo) {
      int idx[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (todo) { idx[q] = __ffsll((long long)todo) - 1; todo &= todo - 1; }
        else idx[q] = idx[q - 1 < 0 ? 0 : q - 1];        // fewer than four left: the last one again (it writes the same values)
      }
      const int row = lane >> 4;
      const int my = row == 0 ? idx[0] : row == 1 ? idx[1] : row == 2 ? idx[2] : idx[3];
      gk_simulate_rows4<S>(m, st.thp, st.rp, my, (uint64_t)(pp.gid0 + act_lo + t0 + my), c.iter);
    }
    while (!ROWS4 && todo) {                             // uniform over the wave
      const int ia = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
#if SABC_GK_PAIR
      int ib = ia;                                       // an odd one out is paired with itself
      if (todo) { ib = __ffsll((long long)todo) - 1; todo &= todo - 1; }
      double tha[D], thb[D], ra[S], rb[S];
#pragma unroll
      for (int k = 0; k < D; ++k) { tha[k] = st.thp[ia][k]; thb[k] = st.thp[ib][k]; }
      // the wanted order statistics of the normals where the quantile function is increasing (phase 2 maps them), else rho
      gk_simulate_wave_ranks_x2<S>(m, tha, thb, (uint64_t)(pp.gid0 + act_lo + t0 + ia), (uint64_t)(pp.gid0 + act_lo + t0 + ib),
                                   c.iter, ra, rb);
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < S; ++j) { st.rp[ia][j] = ra[j]; st.rp[ib][j] = rb[j]; }
      }
#else
      double thp[D], rp[S];
#pragma unroll
      for (int k = 0; k < D; ++k) thp[k] = st.thp[ia][k];
      gk_simulate_wave_ranks<S>(m, thp, (uint64_t)(pp.gid0 + act_lo + t0 + ia), c.iter, rp);
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < S; ++j) st.rp[ia][j] = rp[j];
      }
#endif
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- phase 2: the PW x 4 (particle, statistic) pairs of the wave, PW S / 64 per lane -- all with the lane's statistic
  // j = lane & 3: quantile function of the order statistic of the normals -> distance (device_models.hpp: gk_increasing),
  // then the lane's ECDF lookups (:316) in lockstep on the one table they share
  {
    constexpr int NPASS = PW * S / 64;
    const int j = lane & 3;
    int64_t len = cdf.len[0];
    double obs = m.p[2 + S];
#pragma unroll
    for (int q = 1; q < S; ++q)
      if (j == q) { len = cdf.len[q]; obs = m.p[2 + S + q]; }
    double r[NPASS], upv[NPASS];
    bool live[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int it = pass * (64 / S) + (lane >> 2);
      live[pass] = t0 + it < act_n && st.lpp[it] > -INFINITY;
      r[pass] = 0.0;
      if (live[pass]) {
        double thp[D];
#pragma unroll
        for (int k = 0; k < D; ++k) thp[k] = st.thp[it][k];
        r[pass] = st.rp[it][j];
        if (gk_increasing(thp, m.p[1])) {
          r[pass] = gk_rho_of_normal(thp, m.p[1], r[pass], obs);
          st.rp[it][j] = r[pass];
        }
      }
    }
    cdf_apply_mid_lockstep<NPASS>(cdf.knots + (int64_t)j * cdf.stride, len, cdf.mid + (int64_t)j * cdf.mid_stride, r, upv);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) st.up[pass * (64 / S) + (lane >> 2)][j] = live[pass] ? upv[pass] : 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- phase 3, lane-parallel again: acceptance (:318-329), store, and the particle's moment terms
  double term[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) term[q] = 0.0;
  if (mine) {
    double th[D], u[S], drho[S], thp[D], up[S], rp[S];
#pragma unroll
    for (int k = 0; k < D; ++k) { th[k] = pp.pop[(int64_t)k * pp.cap + li]; thp[k] = st.thp[lane][k]; }
#pragma unroll
    for (int j = 0; j < S; ++j) {
      u[j] = pp.pop[(int64_t)(D + j) * pp.cap + li];
      drho[j] = 0.0;                                         // the change of sum(rho), see k_update
      up[j] = st.up[lane][j];
      rp[j] = st.rp[lane][j];
    }
    const double lpp = st.lpp[lane];
    double log_accept = -INFINITY;
    if (lpp > -INFINITY) {
      double a = 0.0;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const double e = (cb->eps_len == 1) ? cb->eps[0] : cb->eps[j];
        a += (u[j] - up[j]) / e;
      }
      log_accept = lpp - prior_logpdf<D>(m, th) + a + st.logf[lane];
    }
    const u32x4 wa = stream_block(m.seed, gid, PURPOSE_ACCEPT, c.iter, 0);
    const bool accepted = -0.5 * neg2_log_tab(u52(wa.x, wa.y)) < log_accept;      // log(U) < log alpha, :324
    if (accepted) {
#pragma unroll
      for (int k = 0; k < D; ++k) { th[k] = thp[k]; pp.pop[(int64_t)k * pp.cap + li] = thp[k]; }
#pragma unroll
      for (int j = 0; j < S; ++j) {
        u[j] = up[j];
        drho[j] = rp[j] - pp.rho[(int64_t)j * pp.cap + li];
        pp.pop[(int64_t)(D + j) * pp.cap + li] = up[j];
        pp.rho[(int64_t)j * pp.cap + li] = rp[j];
      }
    }
    moment_terms<D, S>(cb->pivot, accepted, th, u, drho, term);
  }
  // sum the moment terms over the wave's PW particle lanes (lanes >= PW hold zeros) into the wave's running row, ...
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    double v = term[q];
#pragma unroll
    for (int off = PW / 2; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][q] += v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();         // the staging arrays are reused by the next group
  }
  // ... then over the 4 waves
  __syncthreads();
  if (threadIdx.x < NP) {
    const int q = threadIdx.x;
    partials[(int64_t)blockIdx.x * NP + q] = ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q];
  }
}

// one wave per row of theta: used for the prior sample at initialization and for sabc_op_simulate.
// Same phase structure as k_update_gk: one lane per particle draws / loads the parameters of the wave's particles in parallel
// (the prior draw is four Philox blocks + Box-Muller pairs per particle: done by all 64 lanes for one particle at a
// time it cost more than the simulation itself -- 1.8 ms for the 1e6 simulations k_update_gk does in 0.7 ms), the whole
// wave then simulates them one after the other, the particles' lanes store.
__global__ void __launch_bounds__(kBlock)
k_simulate_gk(const ModelDesc m, const double *__restrict__ theta_in, const int64_t n, const int64_t stride,
              const uint64_t pid0, const uint64_t iter, const int sample_prior, double *__restrict__ theta_out,
              double *__restrict__ rho_out, const int64_t out_stride) {
  constexpr int D = kGkD, S = kGkS, PW = kGkParticlesPerWave;
  rng_tables_init();
  __shared__ double sth[kBlock / 64][PW][D];
  __shared__ double srho[kBlock / 64][PW][S];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * kGkPerBlock + wave * PW;
  const int64_t i_mine = i0 + lane;
  const bool mine = lane < PW && i_mine < n;
  if (mine) {
    double th[D];
    if (sample_prior) {
      prior_sample<D>(m, pid0 + (uint64_t)i_mine, th);
    } else {
#pragma unroll
      for (int k = 0; k < D; ++k) th[k] = theta_in[(int64_t)k * stride + i_mine];
    }
#pragma unroll
    for (int k = 0; k < D; ++k) sth[wave][lane][k] = th[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int it = 0; it < PW; ++it) {
    if (i0 + it >= n) break;                          // uniform over the wave
    double th[D], rho[S];
#pragma unroll
    for (int k = 0; k < D; ++k) th[k] = sth[wave][it][k];
    gk_simulate_wave_ranks<S>(m, th, pid0 + (uint64_t)(i0 + it), iter, rho);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < S; ++j) srho[wave][it][j] = rho[j];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // order statistics of the normals -> distances, 16 x 4 (particle, statistic) pairs of the wave at once (gk_increasing)
  static_assert(S == 4 && (PW * S) % 64 == 0 && PW <= 64, "one (particle, statistic) pair per lane, PW S / 64 times");
  for (int pass = 0; pass < PW * S / 64; ++pass) {
    const int it = pass * (64 / S) + (lane >> 2), j = lane & 3;
    if (i0 + it < n) {
      double th[D];
#pragma unroll
      for (int k = 0; k < D; ++k) th[k] = sth[wave][it][k];
      double obs = m.p[2 + S];
#pragma unroll
      for (int q = 1; q < S; ++q)
        if (j == q) obs = m.p[2 + S + q];
      if (gk_increasing(th, m.p[1])) srho[wave][it][j] = gk_rho_of_normal(th, m.p[1], srho[wave][it][j], obs);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (mine) {
    if (theta_out) {
#pragma unroll
      for (int k = 0; k < D; ++k) theta_out[(int64_t)k * out_stride + i_mine] = sth[wave][lane][k];
    }
#pragma unroll
    for (int j = 0; j < S; ++j) rho_out[(int64_t)j * out_stride + i_mine] = srho[wave][lane][j];
  }
}

// ------------------------------------------------------------------------------------------
// Host-simulator mode (SABC_MODEL_HOST): f_dist is a host callable, so the per-particle
// body (:308-331) is cut at the simulator.  k_host_propose does :311-314 (proposal, prior gate),
// the host evaluates f_dist for the proposals that passed the gate, k_host_accept does :316-329
// (ECDF, annealed MH test, store).  d and s are run-time values here (any model within the
// maxima); these kernels are host-bound by construction, so they are written for generality.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double prior_logpdf_rt(const ModelDesc &m, const double *th) {
  if (m.prior_joint == 2) return 0.0;                  // host-callback prior: the host overwrites this (hip_backend.hip)
  if (m.prior_joint) return mvnormal_logpdf(m, m.d, th);
  double lp = 0.0;
  for (int k = 0; k < m.d; ++k) {
    const double l = prior_logpdf_dim(m, k, th[k]);
    lp = (l > -INFINITY && lp > -INFINITY) ? lp + l : -INFINITY;
  }
  return lp;
}

// rand(prior) for the shard (:174); theta goes to the population rows
__global__ void __launch_bounds__(kBlock) k_host_prior(const ModelDesc m, const PopPtrs pp) {
  rng_tables_init();
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (li >= pp.n_local) return;
  const uint64_t gid = (uint64_t)(pp.gid0 + li);
  if (m.prior_joint) {
    double th[kMaxPara];
    mvnormal_sample(m, m.d, gid, th);
    for (int k = 0; k < m.d; ++k) pp.pop[(int64_t)k * pp.cap + li] = th[k];
    return;
  }
  for (int k = 0; k < m.d; ++k)
    pp.pop[(int64_t)k * pp.cap + li] = prior_sample_dim(m, k, gid);
}

// one particle of the host-mode proposal step (:311-314)
__device__ __forceinline__ void host_propose_one(const ModelDesc &m, const StepArgs &c, const ControlBlock *__restrict__ cb,
                                                 const PopPtrs &pp, const PartnerView &pv, const int64_t act_lo, const int64_t act_n,
                                                 double *__restrict__ thp_out, double *__restrict__ aux,
                                                 double *__restrict__ thp_host, unsigned char *__restrict__ gate_host,
                                                 double *__restrict__ cur_out, const int64_t t) {
  const int d = m.d;
  const int64_t li = act_lo + t;
  const uint64_t gid = (uint64_t)(pp.gid0 + li);
  double th[kMaxPara], thp[kMaxPara];
  for (int k = 0; k < d; ++k) th[k] = pp.pop[(int64_t)k * pp.cap + li];
  double logf = 0.0;
  if (c.prop_kind == SABC_PROP_RANDOMWALK) {
    NormalStream ns(m.seed, gid, PURPOSE_PROP, c.iter);
    double z[kMaxPara];
    for (int k = 0; k < d; ++k) z[k] = ns.next();
    for (int k = 0; k < d; ++k) {
      double a = 0.0;
      for (int l = 0; l <= k; ++l) a += cb->chol[k * d + l] * z[l];
      thp[k] = th[k] + a;
    }
  } else if (c.prop_kind == SABC_PROP_DIFFEVO) {
    uint64_t i1 = 0, i2 = 0;
    for (uint32_t a = 0;; ++a) {
      const u32x4 w = stream_block(m.seed, gid, PURPOSE_PROP, c.iter, a);
      i1 = mulhi64(pack64(w.x, w.y), (uint64_t)pv.m_total);
      i2 = mulhi64(pack64(w.z, w.w), (uint64_t)pv.m_total);
      if (i1 != i2 || a > 64u) break;
    }
    double z0, z1;
    box_muller(stream_block(m.seed, gid, PURPOSE_PROP2, c.iter, 0), z0, z1);
    const double gamma = c.prop_p0 * (1.0 + c.prop_p1 * z0);
    const double *p1 = partner_ptr(pv, i1), *p2 = partner_ptr(pv, i2);
    for (int k = 0; k < d; ++k) thp[k] = th[k] + gamma * (p1[(int64_t)k * pv.cap] - p2[(int64_t)k * pv.cap]);
  } else {
    const u32x4 w = stream_block(m.seed, gid, PURPOSE_PROP, c.iter, 0);
    const uint64_t ip = mulhi64(pack64(w.x, w.y), (uint64_t)pv.m_total);
    const double U = u52(w.z, w.w);
    const double a = c.prop_p0;
    const double tt = (a - 1.0) * U + 1.0;
    const double z = tt * tt / a;
    const double *p = partner_ptr(pv, ip);
    for (int k = 0; k < d; ++k) {
      const double pk = p[(int64_t)k * pv.cap];
      thp[k] = pk + z * (th[k] - pk);
    }
    logf = log(z) * (double)(d - 1);
  }
  // the proposal stays in device memory for the accept step AND goes to the host for f_dist; of the prior gate the host
  // needs one byte (simulate or not), the log densities stay on the device
  const double lpp = prior_logpdf_rt(m, thp);
  for (int k = 0; k < d; ++k) { thp_out[(int64_t)k * act_n + t] = thp[k]; thp_host[(int64_t)k * act_n + t] = thp[k]; }
  aux[t] = lpp;
  aux[act_n + t] = logf;
  gate_host[t] = lpp > -INFINITY ? 1 : 0;
  if (cur_out)
    for (int k = 0; k < d; ++k) cur_out[(int64_t)k * act_n + t] = th[k];
}

// thp [d][act_n] = proposals, aux [2][act_n] = (log prior of the proposal or -inf, log_factor): device memory, read again by
// k_host_accept.  What the HOST needs goes to pinned host memory mapped into the device (zero copy, no D2H call follows):
// thp_host = the proposals, gate_host [act_n] = one byte per proposal (inside the prior's support?), cur_out (optional,
// [d][act_n]) = the current particles (a host-callback prior needs their log density too).  The half batch is cut into chunks of `sig.chunk`
// particles; the LAST workgroup of a chunk to finish posts `sig.seq` into the chunk's flag word in host memory, which the
// host polls -- it starts f_dist on chunk c while the later chunks are still being proposed, without a stream sync.
struct HostSignal {
  unsigned int *done;            // device: workgroups of each chunk that have finished
  unsigned long long *flag;      // mapped host memory: one word per chunk
  unsigned long long seq;
  int64_t chunk;                 // particles per chunk (a multiple of kBlock)
};

__global__ void __launch_bounds__(kBlock)
k_host_propose(const ModelDesc m, const StepArgs c, const ControlBlock *__restrict__ cb, const PopPtrs pp,
               const PartnerView pv, const int64_t act_lo, const int64_t act_n, double *__restrict__ thp_out,
               double *__restrict__ aux, double *__restrict__ thp_host, unsigned char *__restrict__ gate_host,
               double *__restrict__ cur_out, const HostSignal sig) {
  rng_tables_init();
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < act_n) host_propose_one(m, c, cb, pp, pv, act_lo, act_n, thp_out, aux, thp_host, gate_host, cur_out, t);
  __threadfence_system();                     // this lane's stores to host memory are out ...
  __syncthreads();                            // ... for every lane of the workgroup
  if (threadIdx.x == 0) {
    const int64_t ch = ((int64_t)blockIdx.x * kBlock) / sig.chunk;
    const int64_t first = ch * sig.chunk, last = first + sig.chunk < act_n ? first + sig.chunk : act_n;
    const unsigned int groups = (unsigned int)((last - first + kBlock - 1) / kBlock);
    if (atomicAdd(&sig.done[ch], 1u) == groups - 1u) {
      sig.done[ch] = 0u;                      // ready for the next half batch
      __threadfence_system();
      __hip_atomic_store(&sig.flag[ch], sig.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// rho_prop [s][act_n] from the host; n_accept is counted with an integer atomic (exact, order-free)
__global__ void __launch_bounds__(kBlock)
k_host_accept(const ModelDesc m, const StepArgs c, const ControlBlock *__restrict__ cb, const PopPtrs pp, const CdfPtrs cdf,
              const int64_t act_lo, const int64_t act_n, const int64_t t_lo, const int64_t t_n,
              const double *__restrict__ thp_in,
              const double *__restrict__ aux, const double *__restrict__ rho_prop, const double *__restrict__ lp_host,
              unsigned long long *n_accept) {
  // one chunk [t_lo, t_lo + t_n) of the half batch; thp / aux: device memory (k_host_propose); rho_prop: the host's mapped
  // staging array; lp_host (a host-callback prior only, mapped): [2][act_n] = log prior of the proposals | of the current particles
  const int64_t t = t_lo + (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool accepted = false;
  if (t < t_lo + t_n) {
    const int d = m.d, s = m.s;
    const int64_t li = act_lo + t;
    const uint64_t gid = (uint64_t)(pp.gid0 + li);
    const double lpp = lp_host ? lp_host[t] : aux[t], logf = aux[act_n + t];
    double log_accept = -INFINITY;
    double up[kMaxStats];
    if (lpp > -INFINITY) {
      double th[kMaxPara];
      for (int k = 0; k < d; ++k) th[k] = pp.pop[(int64_t)k * pp.cap + li];
      double a = 0.0;
      for (int j = 0; j < s; ++j) {
        up[j] = cdf_apply_mid(cdf.knots + (int64_t)j * cdf.stride, cdf.len[j], cdf.mid + (int64_t)j * cdf.mid_stride,
                              rho_prop[(int64_t)j * act_n + t]);
        const double e = (cb->eps_len == 1) ? cb->eps[0] : cb->eps[j];
        a += (pp.pop[(int64_t)(d + j) * pp.cap + li] - up[j]) / e;
      }
      log_accept = lpp - (lp_host ? lp_host[act_n + t] : prior_logpdf_rt(m, th)) + a + logf;   // (lp_host: host-callback prior)
    }
    const u32x4 wa = stream_block(m.seed, gid, PURPOSE_ACCEPT, c.iter, 0);
    accepted = log_fast(u52(wa.x, wa.y)) < log_accept;
    if (accepted) {
      for (int k = 0; k < d; ++k) pp.pop[(int64_t)k * pp.cap + li] = thp_in[(int64_t)k * act_n + t];
      for (int j = 0; j < s; ++j) {
        pp.pop[(int64_t)(d + j) * pp.cap + li] = up[j];
        pp.rho[(int64_t)j * pp.cap + li] = rho_prop[(int64_t)j * act_n + t];
      }
    }
  }
  const unsigned long long votes = __ballot(accepted);
  if ((threadIdx.x & 63) == 0 && votes) atomicAdd(n_accept, (unsigned long long)__popcll(votes));
}

// moment sums with run-time d and s (same partial-row layout as k_stats); block 0 also folds the
// accept counter of the host-mode update into component 0 and clears it
__global__ void __launch_bounds__(kBlock)
k_stats_rt(const int d, const int s, const ControlBlock *__restrict__ cb, const PopPtrs pp, double *__restrict__ partials,
           unsigned long long *n_accept) {
  __shared__ double sm[kBlock / 64];
  const int np = n_partials(d, s);
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = li < pp.n_local;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double dk[kMaxPara];
  for (int k = 0; k < d; ++k) dk[k] = live ? pp.pop[(int64_t)k * pp.cap + li] - cb->pivot[k] : 0.0;
  for (int q = 0; q < np; ++q) {
    double v = 0.0;
    if (live) {
      if (q == 0) v = 0.0;
      else if (q < 1 + s) v = pp.pop[(int64_t)(d + q - 1) * pp.cap + li];
      else if (q < 1 + 2 * s) v = pp.rho[(int64_t)(q - 1 - s) * pp.cap + li];
      else if (q < 1 + 2 * s + d) v = dk[q - 1 - 2 * s];
      else {
        int r = q - (1 + 2 * s + d), kk = 0;          // row-major lower index -> (kk, ll)
        while (r > kk) { r -= kk + 1; ++kk; }
        v = dk[kk] * dk[r];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = ((sm[0] + sm[1]) + sm[2]) + sm[3];
      if (q == 0 && blockIdx.x == 0 && n_accept) { tot = (double)*n_accept; *n_accept = 0ull; }
      partials[(int64_t)blockIdx.x * np + q] = tot;
    }
    __syncthreads();
  }
}

// fixed-order sum of the per-block partial rows: block c reduces component c
__global__ void __launch_bounds__(kBlock)
k_reduce_partials(const double *__restrict__ partials, const int64_t rows, const int np, double *__restrict__ sums,
                  const int *__restrict__ halt) {
  __shared__ double sm[kBlock / 64];
  if (halt && *halt) return;               // guarded: part of a step queued ahead of a fired resample test
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x;
  double v = 0.0;
#pragma unroll 4
  for (int64_t r = threadIdx.x; r < rows; r += kBlock) v += partials[r * np + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) sums[c] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

__device__ __forceinline__ void control_on_copy(ControlBlock &lcb, int &ran, ControlBlock *cb, const ControlArgs &a,
                                                double *hist, Mailbox *ring, const double *sums, double *stage) {
  // the multi-eps schedule (:100-117): one lane per statistic computes its epsilon from the sums the step is about to take
  // over -- s^2 divisions and square roots plus s root solves on ONE lane are 12 us per update at s = 3 and over a
  // millisecond at s = 48; lane 0 then applies the candidates inside control_step(), in order
  __shared__ EpsCandidates cand;
  __shared__ double ubar_s[kMaxStats];
  const bool noop = (a.mode & CTRL_GUARDED) && lcb.halt;                  // uniform; nobody has written lcb.halt yet
  ControlArgs a_step = a;
  if (!noop && !(a.mode & CTRL_KEEP_SUMS)) {
    // the sums are taken over by one lane per component; the step then works on them as they stand
    for (int q = threadIdx.x; q < n_partials(a.d, a.s); q += blockDim.x) control_take_sum(lcb, a, sums, q);
    a_step.mode |= CTRL_KEEP_SUMS;
    __syncthreads();
  }
  const bool multi = !noop && (a.mode & CTRL_EPSILON) && a.algorithm == SABC_ALG_MULTI_EPS;
  if (multi) {
    if ((int)threadIdx.x < a.s) ubar_s[threadIdx.x] = lcb.sums[1 + threadIdx.x] / a.n_global;
    __syncthreads();
    if ((int)threadIdx.x < a.s) {
      const int i = threadIdx.x;
      cand.ok[i] = hostmath::eps_multi_one(ubar_s, a.s, a.v, hostmath::eps_multi_cn(a.s), i, &cand.eps[i]) ? 1 : 0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) ran = control_step(lcb, a_step, hist, sums, &cand, multi) ? 1 : 0;
  __syncthreads();
  if (!ran) {                               // guarded and halted: nothing changed
    // ... and nothing is posted, unless the halt is a peer-to-peer wait that gave up (p2p.hpp): the host is waiting for
    // this step's sequence word and has to learn of the error
    if (threadIdx.x == 0 && a.notify_seq != 0 && lcb.error == SABC_ERR_COMM) mailbox_post(ring, a, lcb);
    return;
  }
  for (int i = threadIdx.x; i < kControlWords; i += blockDim.x)
    reinterpret_cast<uint64_t *>(cb)[i] = reinterpret_cast<const uint64_t *>(&lcb)[i];
  if (stage)
    for (int q = threadIdx.x; q < n_partials(a.d, a.s); q += blockDim.x) stage[q] = sums[q];
  if (threadIdx.x == 0 && a.notify_seq != 0) mailbox_post(ring, a, lcb);
}

__global__ void __launch_bounds__(64)
k_control(ControlBlock *cb, const ControlArgs a, double *hist, Mailbox *ring, const double *sums_in) {
  __shared__ ControlBlock lcb;
  __shared__ int ran;
  __shared__ double sums[kMaxPartials];
  const int np = n_partials(a.d, a.s);
  control_load(lcb, cb);
  for (int i = threadIdx.x; i < np; i += blockDim.x) sums[i] = sums_in[i];
  __syncthreads();
  control_on_copy(lcb, ran, cb, a, hist, ring, sums, nullptr);
}

// ------------------------------------------------------------------------------------------
// peer-to-peer exchange over mapped slots (p2p.hpp)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void p2p_store(uint64_t *p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);        // one 8-byte store, past the caches
}
__device__ __forceinline__ uint64_t p2p_load(const uint64_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t p2p_clock() { return (uint64_t)wall_clock64(); }

// spin until the word's upper half is the tag `seq`; gives up after `ticks`, as soon as another lane of the workgroup has,
// or as soon as the awaited shard has LEFT the group (its leave word in this shard's slots carries the tag's generation:
// p2p.hpp) -- *failed = 1 + peer, + 16 when the peer left
__device__ __forceinline__ uint64_t p2p_wait_word(const uint64_t *src, uint32_t seq, uint64_t t0, uint64_t ticks, volatile int *failed,
                                                  int peer, const uint64_t *my_slots) {
  uint64_t w = p2p_load(src);
  for (uint32_t polls = 1; (uint32_t)(w >> 32) != seq; ++polls) {
    if ((polls & 15u) == 0) {
      if (*failed) break;
      if (p2p_clock() - t0 > ticks) { *failed = 1 + peer; break; }
      const uint64_t lw = p2p_load(my_slots + kP2PLeaveOff + peer);
      if ((uint32_t)(lw >> 32) == p2p_tag_gen(seq) && (uint32_t)lw == 1u) {
        w = p2p_load(src);                                   // (what it posted before it left still counts)
        if ((uint32_t)(w >> 32) != seq) *failed = 17 + peer;
        break;
      }
    }
    __builtin_amdgcn_s_sleep(2);
    w = p2p_load(src);
  }
  return w;
}

// a wait gave up: the error goes into the control block together with the halt flag (everything queued behind becomes a
// no-op) and, if the host is waiting for this step, into the mailbox
__device__ __forceinline__ void p2p_fail(ControlBlock *cb, ControlBlock *lcb, const ControlArgs *a, Mailbox *ring, int kind, int failed,
                                         uint32_t seq) {
  cb->error = SABC_ERR_COMM;
  cb->halt = 1;
  // kind: 1 sums exchange | 2 barrier | 3 end-of-call status; + 4 when the shard waited for has left the group
  cb->comm_where = ((kind + (failed > 16 ? 4 : 0)) << 24) | (((failed - 1) & 15) << 20) | (int)(seq & kP2PSeqMask);
  if (lcb) { lcb->error = SABC_ERR_COMM; lcb->halt = 1; lcb->comm_where = cb->comm_where; }
  __threadfence();
  if (lcb && a && ring && a->notify_seq != 0) mailbox_post(ring, *a, *lcb);
}

// sum of the shards' rows of `np` doubles: `mine` (LDS) goes to every peer's slots in the LL form, the peers' rows are
// awaited in this shard's slots, and the rows are added in RANK order (every shard gets bitwise the same sums).
// All threads of the workgroup call it; returns false when a wait gave up.  words: LDS, world * 2 np.
__device__ __forceinline__ bool p2p_allreduce_rows(const P2PView &pv, const uint32_t seq, const int np, double *mine,
                                                   uint32_t *words, volatile int *failed, const int silent) {
  const int W = pv.world, nw = 2 * np, ring = (int)(seq % kP2PRing);
  const uint32_t *half = reinterpret_cast<const uint32_t *>(mine);
  if (silent != 1)                                   // test hook: 1 = nothing is posted, 2 = the post reaches this shard's own slots only
    for (int i = threadIdx.x; i < W * nw; i += blockDim.x) {
      const int p = i / nw, t = i - p * nw;
      if (silent == 2 && p != pv.rank) continue;
      p2p_store(pv.slots[p] + kP2PSumsOff + ((int64_t)ring * kMaxPeers + pv.rank) * kP2PWords + t, ((uint64_t)seq << 32) | half[t]);
    }
  const uint64_t t0 = p2p_clock();
  for (int i = threadIdx.x; i < W * nw; i += blockDim.x) {
    const int r = i / nw, t = i - r * nw;
    const uint64_t w = p2p_wait_word(pv.slots[pv.rank] + kP2PSumsOff + ((int64_t)ring * kMaxPeers + r) * kP2PWords + t, seq, t0,
                                     pv.timeout_ticks, failed, r, pv.slots[pv.rank]);
    words[i] = (uint32_t)w;
  }
  __syncthreads();
  if (*failed) return false;
  if ((int)threadIdx.x < np) {
    const int q = threadIdx.x;
    double a = 0.0;
    for (int r = 0; r < W; ++r) {
      const double x = __hiloint2double((int)words[r * nw + 2 * q + 1], (int)words[r * nw + 2 * q]);
      a = r == 0 ? x : a + x;
    }
    mine[q] = a;
  }
  __syncthreads();
  return true;
}

// k_reduce_partials + [the sum over the shards] + k_control in ONE launch.
//  XCHG = false: one shard, no collective in between.
//  XCHG = true : several shards over the peer-to-peer slots -- what was k_reduce_partials -> ncclAllReduce -> k_control.
// 1024 threads: thread (g, c) sums rows g, g+G, ... of column c (consecutive threads read consecutive addresses), LDS
// combines the G row groups in a fixed order, lane 0 runs the control step on the sums.  The loads of the control block
// and of the partial rows are issued together (one round trip); the staging buffer is written only by a step that runs.
// rows < 0: the shard's sums are already in `stage` (k_reduce_partials ran: a partial matrix too large for one workgroup).
// do_control == 0: only the (global) sums, into `stage` (whoever asked for sums_buffer()).
struct XchgArgs {
  P2PView pv;
  uint32_t seq;
  int32_t do_control, silent, reserved;
};

// lane i of every row of 16 receives the value of lane i - k of its row (0.0 where there is none): v_mov_b32 dpp row_shr:k x 2
template <int CTRL>
__device__ __forceinline__ double dpp_row_shr(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

#ifdef SABC_RC_TIMING
#define RC_MARK(i) rc_mark(i)
#else
#define RC_MARK(i) do { } while (0)
#endif

template <bool XCHG>
__global__ void __launch_bounds__(1024)
k_reduce_control(const double *__restrict__ partials, const int64_t rows, const int np, double *__restrict__ stage,
                 ControlBlock *cb, const ControlArgs a, double *hist, Mailbox *ring, const XchgArgs x) {
  __shared__ ControlBlock lcb;
  __shared__ int ran;
  __shared__ int failed;
  __shared__ double sm[1024];
  __shared__ double sums[kMaxPartials];
  __shared__ uint32_t words[XCHG ? kMaxPeers * kP2PWords : 1];
  RC_MARK(0);
  const int B = blockDim.x;                         // 1024, or 256 for a short matrix of partial rows (launch_reduce_control)
  const int G = B / np;
  const int g = threadIdx.x / np, c = threadIdx.x - g * np;
  // the rows first (they come from the other XCDs' blocks, i.e. from memory: the longest latency of this launch), then the
  // control block; all of a lane's rows in ONE round trip where they fit (20 at n = 1e6: 3906 rows over 204 row groups),
  // masked so that there is no tail of dependent single loads (each a trip to the L2: 3-4 of them were ~3 us of this
  // kernel); the additions stay in row order, a masked slot adds +0
  constexpr int kInFlight = 24;
  double xx[kInFlight];
  if (rows >= 0 && g < G) {
#pragma unroll
    for (int e = 0; e < kInFlight; ++e) {
      const int64_t r = g + (int64_t)e * G;
      xx[e] = r < rows ? partials[r * np + c] : 0.0;
    }
  }
  control_load(lcb, cb);
#ifdef SABC_RC_TIMING
  __builtin_amdgcn_s_waitcnt(0);                    // thread 0's loads have arrived
  RC_MARK(7);
#endif
  if (threadIdx.x == 0) failed = 0;
  if (rows >= 0) {
    double v = 0.0;
    if (g < G) {
#pragma unroll
      for (int e = 0; e < kInFlight; ++e) v += xx[e];
      for (int64_t r0 = g + (int64_t)kInFlight * G; r0 < rows; r0 += (int64_t)kInFlight * G) {
#pragma unroll
        for (int e = 0; e < kInFlight; ++e) {
          const int64_t r = r0 + (int64_t)e * G;
          xx[e] = r < rows ? partials[r * np + c] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < kInFlight; ++e) v += xx[e];
      }
    }
    sm[threadIdx.x] = v;
    __syncthreads();
    RC_MARK(1);                                     // control block + partial rows loaded
    const int n_waves = B >> 6;
    if (np <= 16) {
      // one WAVE per column: lane l adds the groups l, l + 64, ... (<= 4 LDS reads), the 64 lane sums are added inside the
      // wave -- DPP row shifts, then the four row totals in order -- without another barrier or LDS round (the 8-level LDS
      // tree below was 1.2 us of this launch, tools/rc_timing.py)
      const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
      for (int col = w; col < np; col += n_waves) {
        double t = 0.0;
        for (int gg = l; gg < G; gg += 64) t += sm[gg * np + col];
        t += dpp_row_shr<0x111>(t);
        t += dpp_row_shr<0x112>(t);
        t += dpp_row_shr<0x114>(t);
        t += dpp_row_shr<0x118>(t);                 // lane 16 r + 15 now holds the sum of row r
        const double total = ((read_lane(t, 15) + read_lane(t, 31)) + read_lane(t, 47)) + read_lane(t, 63);
        if (l == 0) sums[col] = total;
      }
    } else {
      // fixed-shape tree over the G row groups (a serial sum by np lanes would be G dependent LDS reads: 6 us at G = 204)
      int top = 1;
      while (top * 2 < G) top *= 2;
      for (int stride = top; stride >= 1; stride >>= 1) {
        if (g < stride && g + stride < G) sm[threadIdx.x] += sm[threadIdx.x + stride * np];
        __syncthreads();
      }
      if ((int)threadIdx.x < np) sums[threadIdx.x] = sm[threadIdx.x];
    }
  } else if ((int)threadIdx.x < np) {
    sums[threadIdx.x] = stage[threadIdx.x];
  }
  __syncthreads();
  RC_MARK(2);                                       // tree
  if (XCHG) {
    // every shard takes the same decision here (the halt flag follows from sums all shards share), so a step that is a
    // no-op posts nothing on ANY shard and nobody waits for it
    const bool noop = ((a.mode & CTRL_GUARDED) && lcb.halt) || lcb.error == SABC_ERR_COMM;
    if (noop) {
      if (threadIdx.x == 0 && x.do_control && a.notify_seq != 0 && lcb.error == SABC_ERR_COMM) mailbox_post(ring, a, lcb);
      return;
    }
    if (!p2p_allreduce_rows(x.pv, x.seq, np, sums, words, &failed, x.silent)) {
      if (threadIdx.x == 0) p2p_fail(cb, &lcb, x.do_control ? &a : nullptr, ring, 1, failed, x.seq);
      return;
    }
    if (!x.do_control) {
      if ((int)threadIdx.x < np) stage[threadIdx.x] = sums[threadIdx.x];
      return;
    }
  }
  RC_MARK(3);                                       // exchange
#ifdef SABC_RC_TIMING
  if (threadIdx.x == 0) ran = control_step(lcb, a, hist, sums) ? 1 : 0;
  __syncthreads();
  RC_MARK(4);                                       // control step (one lane)
  if (!ran) return;
  for (int i = threadIdx.x; i < kControlWords; i += blockDim.x)
    reinterpret_cast<uint64_t *>(cb)[i] = reinterpret_cast<const uint64_t *>(&lcb)[i];
  if (stage && (int)threadIdx.x < n_partials(a.d, a.s)) stage[threadIdx.x] = sums[threadIdx.x];
  __syncthreads();
  RC_MARK(5);                                       // write back issued
  if (threadIdx.x == 0 && a.notify_seq != 0) mailbox_post(ring, a, lcb);
  RC_MARK(6);                                       // mailbox
#else
  control_on_copy(lcb, ran, cb, a, hist, ring, sums, stage);
#endif
}

#ifdef SABC_RC_TIMING
extern "C" __attribute__((visibility("default"))) int sabc_debug_rc_ticks(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rc_ticks), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rc_ticks), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

// Flag barrier between the shards' streams: everything every shard has enqueued before its barrier `seq` has completed
// (kernel boundary) before anything enqueued behind it starts.  Lane r posts to / waits for shard r.
__global__ void __launch_bounds__(64)
k_p2p_barrier(const P2PView pv, const uint32_t seq, ControlBlock *cb, const int guarded, const int silent) {
  __shared__ int failed;
  if (threadIdx.x == 0) failed = 0;
  __syncthreads();
  if ((guarded && cb->halt) || cb->error == SABC_ERR_COMM) return;        // the same on every shard (see k_reduce_control)
  const int r = threadIdx.x, ring = (int)(seq % kP2PRing);
  __threadfence_system();
  if (r < pv.world && silent != 1 && (silent != 2 || r == pv.rank)) p2p_store(pv.slots[r] + kP2PBarOff + (int64_t)ring * kMaxPeers + pv.rank, ((uint64_t)seq << 32) | 1u);
  if (r < pv.world)
    (void)p2p_wait_word(pv.slots[pv.rank] + kP2PBarOff + (int64_t)ring * kMaxPeers + r, seq, p2p_clock(), pv.timeout_ticks, &failed, r,
                        pv.slots[pv.rank]);
  __syncthreads();
  if (threadIdx.x == 0 && failed) p2p_fail(cb, nullptr, nullptr, nullptr, 2, failed, seq);
  __threadfence_system();
}

// End of a sabc_initialize / sabc_update call over the peer-to-peer transport: every shard tells the others how the call
// went (status 0 = fine) and -- on the success path -- learns the same of them, so that a shard whose peer gave up in the
// call's LAST exchange does not return success on its own.  A shard that failed posts without waiting.
__global__ void __launch_bounds__(64)
k_p2p_commit(const P2PView pv, const uint32_t call, const int status, const int wait, ControlBlock *cb, const int silent) {
  __shared__ int failed;
  if (threadIdx.x == 0) failed = 0;
  __syncthreads();
  const int r = threadIdx.x;
  const int mine = (status != 0 || cb->error != 0) ? 1 : 0;
  if (r < pv.world && silent != 1 && (silent != 2 || r == pv.rank)) p2p_store(pv.slots[r] + kP2PCommitOff + pv.rank, ((uint64_t)call << 32) | (uint32_t)mine);
  if (!wait) return;
  if (r < pv.world) {
    const uint64_t w = p2p_wait_word(pv.slots[pv.rank] + kP2PCommitOff + r, call, p2p_clock(), pv.timeout_ticks, &failed, r,
                                     pv.slots[pv.rank]);
    if ((uint32_t)w != 0u && !failed) failed = 1 + r;                     // the peer's call failed
  }
  __syncthreads();
  if (threadIdx.x == 0 && failed && cb->error == 0) p2p_fail(cb, nullptr, nullptr, nullptr, 3, failed, call);
}

// rows of known values through the slots, for sabc_comm_p2p_selftest: out[q] = sum over shards of in[q]
__global__ void __launch_bounds__(1024)
k_p2p_selftest(const P2PView pv, const uint32_t seq, const int np, const double *__restrict__ in, double *__restrict__ out,
               int *__restrict__ failed_out, const int silent) {
  __shared__ int failed;
  __shared__ double sums[kMaxPartials];
  __shared__ uint32_t words[kMaxPeers * kP2PWords];
  if (threadIdx.x == 0) failed = 0;
  if ((int)threadIdx.x < np) sums[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  const bool ok = p2p_allreduce_rows(pv, seq, np, sums, words, &failed, silent);
  if (ok && (int)threadIdx.x < np) out[threadIdx.x] = sums[threadIdx.x];
  if (threadIdx.x == 0) *failed_out = ok ? 0 : failed;
}

// This shard leaves the group of generation `gen`: one word into every peer's slots (p2p.hpp); lane r tells shard r.
__global__ void __launch_bounds__(64) k_p2p_leave(const P2PView pv, const uint32_t gen) {
  const int r = threadIdx.x;
  if (r < pv.world && pv.slots[r]) p2p_store(pv.slots[r] + kP2PLeaveOff + pv.rank, ((uint64_t)gen << 32) | 1u);
}

// First contact, second half (sabc_comm_p2p_selftest): what the transport READS.  Partners, resampled rows and the ECDF
// build read a peer's populations and rho -- plain device memory, written by the owner's kernels, made visible by nothing but
// a kernel boundary on each side of a flag (p2p.hpp).  Every shard writes a pattern tagged with (generation, round, rank,
// buffer, sample) into `count` doubles spread evenly over each of its three buffers (mode 0: after parking what was there
// in `save`; mode 1: the second round), a barrier, every shard reads every shard's samples through its mappings and counts
// what is not the pattern; mode 2 puts the parked values back.
struct PatternBufs {
  uint64_t *buf[3];              // population buffer 0, population buffer 1, rho (as 64-bit words)
  int64_t len[3];                // doubles in each
  int32_t count[3];              // samples in each (<= kPatternSamples)
};
constexpr int kPatternSamples = 1024;
__device__ __forceinline__ uint64_t pattern_word(uint32_t gen, int round, int rank, int b, int k) {
  return 0x5AB0000000000000ull | ((uint64_t)(gen & 0xFFFu) << 40) | ((uint64_t)(round & 0xFF) << 32) | ((uint64_t)(rank & 0xFF) << 24) |
         ((uint64_t)(b & 0xF) << 20) | (uint64_t)(k & 0xFFFFF);
}
__device__ __forceinline__ int64_t pattern_index(int64_t len, int count, int k) { return (int64_t)k * (len / count); }

__global__ void __launch_bounds__(256)
k_p2p_pattern_write(const PatternBufs own, uint64_t *__restrict__ save, const uint32_t gen, const int round, const int rank, const int mode) {
  const int b = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < own.count[b]; k += gridDim.x * blockDim.x) {
    uint64_t *p = own.buf[b] + pattern_index(own.len[b], own.count[b], k);
    if (mode == 2) { *p = save[b * kPatternSamples + k]; continue; }
    if (mode == 0) save[b * kPatternSamples + k] = *p;
    *p = pattern_word(gen, round, rank, b, k);
  }
}

struct PatternPeers {
  const uint64_t *buf[3][kMaxPeers];
};
// out[0] = mismatches, out[1] = first mismatch as rank << 28 | buffer << 24 | sample (valid when out[0] > 0)
__global__ void __launch_bounds__(256)
k_p2p_pattern_check(const PatternPeers peers, const PatternBufs geo, const uint32_t gen, const int round, const int world,
                    unsigned int *__restrict__ out) {
  const int b = blockIdx.y, r = blockIdx.z;
  if (r >= world || !peers.buf[b][r]) return;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < geo.count[b]; k += gridDim.x * blockDim.x) {
    const uint64_t got = peers.buf[b][r][pattern_index(geo.len[b], geo.count[b], k)];    // a plain load, like the transport's
    if (got != pattern_word(gen, round, r, b, k)) {
      if (atomicAdd(&out[0], 1u) == 0u) out[1] = ((unsigned)r << 28) | ((unsigned)b << 24) | (unsigned)k;
    }
  }
}

// K3 over the shard: u = cdf(rho)  (:190-192)
__global__ void __launch_bounds__(kBlock) k_cdf_population(const int d, const int s, const PopPtrs pp, const CdfPtrs cdf) {
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (li >= pp.n_local) return;
  // mid level (every 16th knot, L2-resident) -> one line of the table: same rank as the plain search, ~7 instead of ~20
  // distinct lines per lookup
  for (int j = 0; j < s; ++j)
    pp.pop[(int64_t)(d + j) * pp.cap + li] = cdf_apply_mid(cdf.knots + (int64_t)j * cdf.stride, cdf.len[j],
                                                           cdf.mid + (int64_t)j * cdf.mid_stride, pp.rho[(int64_t)j * pp.cap + li]);
}

// ------------------------------------------------------------------------------------------
// K5: resample (SimulatedAnnealingABC.jl:124-137)
// ------------------------------------------------------------------------------------------
// w_i = exp(-sum_j u_ij delta / ubar_j), :126-127
__device__ __forceinline__ double particle_weight(const int d, const int s, const PopPtrs &pp, const ControlBlock *__restrict__ cb,
                                                  const double n_global, const double delta, const int64_t li) {
  double a = 0.0;
  for (int j = 0; j < s; ++j) {
    const double ubar = cb->sums[1 + j] / n_global;                                         // :126
    a += pp.pop[(int64_t)(d + j) * pp.cap + li] * delta / ubar;                             // :127
  }
  return exp(-a);
}

__global__ void __launch_bounds__(kBlock)
k_resample_weights(const int d, const int s, const PopPtrs pp, const ControlBlock *__restrict__ cb,
                   const double n_global, const double delta) {
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (li >= pp.n_local) return;
  pp.pop[(int64_t)(d + s) * pp.cap + li] = particle_weight(d, s, pp, cb, n_global, delta, li);
}

__device__ __forceinline__ double gathered_weight(const ShardBlocks &g, int64_t gid) {
  int64_t r, o;
  split_index(gid, g.cap, r, o);
  return shard_block(g, r)[(int64_t)(g.rows - 1) * g.cap + o];
}

// pass 1: per-chunk sums of w and w^2.  One shard (wargs.fused): the weights are computed here from the u rows and
// written to the weight row on the way (no separate k_resample_weights launch); same arithmetic, same values.
struct WeightArgs {
  int fused, d, s, reserved;
  PopPtrs pp;
  const ControlBlock *cb;
  double n_global, delta;
};

__global__ void __launch_bounds__(kBlock)
k_scan_sums(const ShardBlocks g, const int64_t n, double *__restrict__ bs, double *__restrict__ bq, const WeightArgs wa,
            double *__restrict__ wcopy) {
  __shared__ double sm[2][kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 4;
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t i = base + e;
    double w = 0.0;
    if (i < n) {
      if (wa.fused) {
        w = particle_weight(wa.d, wa.s, wa.pp, wa.cb, wa.n_global, wa.delta, i);
        wa.pp.pop[(int64_t)(wa.d + wa.s) * wa.pp.cap + i] = w;
      } else {
        w = gathered_weight(g, i);
        if (wcopy) wcopy[i] = w;       // weights read from their owners (peer-mapped): the last pass finds them here, not over xGMI again
      }
    }
    s += w; q += w * w;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
  if (lane == 0) { sm[0][wave] = s; sm[1][wave] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    bs[blockIdx.x] = ((sm[0][0] + sm[0][1]) + sm[0][2]) + sm[0][3];
    bq[blockIdx.x] = ((sm[1][0] + sm[1][1]) + sm[1][2]) + sm[1][3];
  }
}

// pass 2 (single block of 1024): exclusive scan of the chunk sums in place; totals.  Thread t owns
// `per` consecutive chunks; the 1024 thread totals are scanned in LDS by a fixed-shape
// Hillis-Steele network (same result on every run and every shard).
__global__ void __launch_bounds__(1024)
k_scan_offsets(double *__restrict__ bs, const double *__restrict__ bq, const int64_t nb, double *__restrict__ totals,
               double *__restrict__ totals_host) {
  __shared__ double sa[2][1024];
  __shared__ double sq[1024];
  const int t = threadIdx.x;
  const int64_t per = (nb + 1023) / 1024;
  const int64_t lo = (int64_t)t * per, hi = (lo + per < nb) ? lo + per : nb;
  double s = 0.0, q = 0.0;
  for (int64_t b = lo; b < hi; ++b) { s += bs[b]; q += bq[b]; }
  sa[0][t] = s; sq[t] = q;
  __syncthreads();
  int cur = 0;
  for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan of the thread totals
    sa[1 - cur][t] = t >= off ? sa[cur][t] + sa[cur][t - off] : sa[cur][t];
    cur = 1 - cur;
    __syncthreads();
  }
  for (int off = 512; off > 0; off >>= 1) {           // tree sum of the squares
    if (t < off) sq[t] += sq[t + off];
    __syncthreads();
  }
  if (t == 0) {
    totals[0] = sa[cur][1023]; totals[1] = sq[0];
    if (totals_host) { totals_host[0] = sa[cur][1023]; totals_host[1] = sq[0]; }   // pinned + mapped: the ESS of :134, no memcpy
  }
  double run = t > 0 ? sa[cur][t - 1] : 0.0;          // exclusive offset of this thread's first chunk
  for (int64_t b = lo; b < hi; ++b) { const double v = bs[b]; bs[b] = run; run += v; }
}

// pass 3: inclusive scan inside each chunk + chunk offset
// Packed lines (one shard): particle i's running sum AND its (theta, u) row sit together, `pg` particles to a 128-byte
// line (pg = 4, 2 or 1: the largest power of two with pg (1 + row_len) <= 16, so that a line never straddles a scan chunk
// and every particle's slot starts on a 16 / pg-double boundary):
//   pk[(i / pg) * 16 + (i % pg) * (16 / pg)] = { cum_i, theta_i..., u_i... }
//   ge[i / pg]  = cum at the line's last particle                      (kScanChunk / pg per chunk)
//   guide[b]    = a line whose running sums reach bucket b of [0, total), n_lines + 2 buckets (guide_bucket)
// A draw then costs THREE dependent fetches (the guide entry, two line ends, the packed line) instead of ~10 (binary
// search through `cm` and `cum`, one line per gathered row).  The running sums are the same numbers, so the drawn index
// is the same.
struct PackArgs {
  double *pk, *ge;       // pk == nullptr: no packing (the sharded path gathers rows by request)
  int32_t *guide;        // guide[b]: a line whose running sums reach bucket b of [0, total) -- where a draw starts looking
  const double *totals;  // totals[0] = sum of the weights (written by k_scan_offsets)
  int row_len, pg;
};

// bucket of a running sum t: n_lines equal buckets over [0, total).  The SAME expression places the lines in
// k_scan_final and the draws in the gather; it only has to be monotone -- the guide is a starting point, the search
// around it decides (packed_search).
__device__ __forceinline__ int64_t guide_bucket(const double t, const double total, const int64_t n_lines) {
  const double x = t * ((double)n_lines / total);
  const int64_t b = x > 0.0 ? (int64_t)x : 0;             // (NaN -> 0)
  return b <= n_lines + 1 ? b : n_lines + 1;
}

// pass 3: inclusive scan inside each chunk + chunk offset
__global__ void __launch_bounds__(kBlock)
k_scan_final(const ShardBlocks g, const int64_t n, const double *__restrict__ bs, double *__restrict__ cum,
             double *__restrict__ cm, const PackArgs pa, const int w_in_cum) {
  __shared__ double sm[kBlock];
  __shared__ double scum[kScanChunk];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 4;
  double w[4];
  double s = 0.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t i = base + e;
    // w_in_cum: the first pass left the weights in `cum` (each element is read here before this thread overwrites it below)
    w[e] = i < n ? (w_in_cum ? cum[i] : gathered_weight(g, i)) : 0.0;
    s += w[e];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  // Exclusive scan of the 256 thread totals in the FIXED sequential order 0, 1, 2, ... (part of the summation order the
  // oracle shares).  One lane walking the LDS array paid a dependent LDS round trip per element (~12 us of the kernel);
  // here the first wave holds the totals in registers (4 per lane) and the running sum visits them in the same order
  // through v_readlane: the same 256 additions, in registers.
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const double a[4] = {sm[4 * lane], sm[4 * lane + 1], sm[4 * lane + 2], sm[4 * lane + 3]};
    double ex[4] = {0.0, 0.0, 0.0, 0.0};
    double run = 0.0;
    for (int l = 0; l < 64; ++l) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[e]), l),
                                          __builtin_amdgcn_readlane(__double2loint(a[e]), l));
        if (lane == l) ex[e] = run;
        run += b;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[4 * lane + e] = ex[e];
  }
  __syncthreads();
  double run = bs[blockIdx.x] + sm[threadIdx.x];
  const bool packed = pa.pk != nullptr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t i = base + e;
    run += w[e];
    if (packed) { scum[threadIdx.x * 4 + e] = run; continue; }       // the packed gather reads neither cum nor cm
    if (i < n) cum[i] = run;
    // mid level of the resample search: cm[g] = cum at the end of 16-element group g (one 128-byte line of `cum`);
    // weights behind n are 0, so `run` is the total there; groups entirely behind n get +inf
    if ((i & 15) == 15) cm[i >> 4] = (i - 15 < n) ? run : INFINITY;
  }
  if (!packed) return;
  __syncthreads();
  // consecutive lanes take consecutive particles here (not 4 each, as in the scan): coalesced row reads, and the pg lanes
  // of a line write its 128 bytes with 16-byte stores
  const int stride = 16 / pa.pg;
  const int64_t n_lines = (n + pa.pg - 1) / pa.pg;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int loc = threadIdx.x + e * kBlock;
    const int64_t i = (int64_t)blockIdx.x * kScanChunk + loc;
    const int64_t line = i / pa.pg;
    if (line >= n_lines) continue;
    const int slot = (int)(i - line * pa.pg);
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = 0.0;
    v[0] = i < n ? scum[loc] : INFINITY;                             // empty slots of the last line never win a search
    if (i < n)
      for (int row = 0; row < pa.row_len; ++row) v[1 + row] = g.flat[(int64_t)row * g.cap + i];   // packing: one shard, its own block
    double2 *dst = reinterpret_cast<double2 *>(pa.pk + line * 16 + slot * stride);
    // (only the slot's used part: the padding behind 1 + row_len doubles is never read)
    for (int q = 0; 2 * q < stride && 2 * q < 1 + pa.row_len; ++q) dst[q] = make_double2(v[2 * q], v[2 * q + 1]);
    // the line's end value: its last particle, or the last particle of the population (scum is flat behind n)
    if (i < n && (slot == pa.pg - 1 || i == n - 1)) {
      const double e1 = scum[loc];
      pa.ge[line] = e1;
      // the buckets this line's running sums reach: from the end of the line before it (the chunk's offset for the chunk's
      // first line, inclusive there so that rounding between the offset and the previous chunk's end leaves no bucket
      // unwritten) to its own end; the population's last line takes the rest
      const int first_loc = loc - slot;                              // the line's first particle, inside this chunk
      const double total = pa.totals[0];
      int64_t b0 = first_loc > 0 ? guide_bucket(scum[first_loc - 1], total, n_lines) + 1 : guide_bucket(bs[blockIdx.x], total, n_lines);
      int64_t b1 = i == n - 1 ? n_lines + 1 : guide_bucket(e1, total, n_lines);
      for (int64_t b = b0; b <= b1; ++b) pa.guide[b] = (int32_t)line;
      if (i == n - 1) { pa.ge[line + 1] = INFINITY; pa.ge[line + 2] = INFINITY; }
    }
  }
}

// n_local categorical draws + gather of theta and u rows (rho is NOT permuted, :131-132).
// Inverse CDF by a three-level search, one line of `cum` per draw: the exclusive chunk offsets `bs` of the
// weight scan (one per 1024 weights, in LDS) -> `cm`, the running sum at the end of every 16-element group
// (64 per chunk, 0.5 MB at n = 1e6: L2-resident) -> the 16 elements of that group (one 128-byte line).
constexpr int kGatherCoarseMax = 4096;     // chunks held in LDS (n <= 4.2e6); beyond that bs is searched in global memory
constexpr int kGroupsPerChunk = kScanChunk / 16;
// the chunk offsets into LDS, four reads in flight per thread (nb <= 4096 and 256 threads: at most 4 trips to memory at the
// front of every workgroup of a latency-bound kernel instead of 16); the caller's __syncthreads() publishes them
__device__ __forceinline__ void stage_chunk_offsets(const double *__restrict__ bs, const int64_t nb, double *lds) {
  for (int64_t i0 = threadIdx.x; i0 < nb; i0 += 4 * kBlock) {
    double t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int64_t i = i0 + (int64_t)e * kBlock; t[e] = i < nb ? bs[i] : 0.0; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int64_t i = i0 + (int64_t)e * kBlock; if (i < nb) lds[i] = t[e]; }
  }
}
// first index k with cum[k] > t, by the three levels described above (B = the chunk offsets, in LDS or global memory)
__device__ __forceinline__ int64_t resample_search(const double t, const double *B, const int64_t nb,
                                                   const double *__restrict__ cm, const double *__restrict__ cum,
                                                   const int64_t n) {
  int64_t blo = 0, bhi = nb;              // first chunk whose offset exceeds t; bs[0] = 0 <= t
  while (blo < bhi) {
    const int64_t mid = blo + ((bhi - blo) >> 1);
    if (B[mid] > t) bhi = mid; else blo = mid + 1;
  }
  const int64_t chunk = blo - 1;
  // first group of the chunk whose end value exceeds t (count form over the chunk's 64 group ends)
  int64_t grp = chunk * kGroupsPerChunk;
#pragma unroll
  for (int step = kGroupsPerChunk >> 1; step >= 1; step >>= 1)
    if (cm[grp + step - 1] <= t) grp += step;
  if (cm[grp] <= t) grp += 1;             // 64 of 64: t is not below the chunk's own end (rounding of bs vs cum)
  int64_t lo = grp << 4, hi = lo + 16;    // first k in the group with cum[k] > t
  if (grp == (chunk + 1) * kGroupsPerChunk) hi = lo;
  if (hi > n) hi = n;
  if (lo > n) lo = n;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (cum[mid] > t) hi = mid; else lo = mid + 1;
  }
  return lo < n ? lo : n - 1;
}

// One shard: the draw and the gather of its rows in one kernel.  Several shards (k_resample_select): the draws only,
// as global source indices; the rows are fetched from their owners afterwards (k_resample_serve / _scatter).
template <bool GATHER>
__global__ void __launch_bounds__(kBlock)
k_resample_gather(const uint64_t seed, const int d, const int s, const ShardBlocks g, const int64_t n,
                  const double *__restrict__ cum, const double *__restrict__ bs,
                  const double *__restrict__ cm, const int64_t nb, const double *__restrict__ totals, const uint64_t iter,
                  const PopPtrs dst, int64_t *__restrict__ idx_out) {
  extern __shared__ double bs_lds[];
  const bool in_lds = nb <= kGatherCoarseMax;
  if (in_lds) {
    stage_chunk_offsets(bs, nb, bs_lds);
    __syncthreads();
  }
  const double *B = in_lds ? bs_lds : bs;
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (li >= dst.n_local) return;
  const uint64_t gid = (uint64_t)(dst.gid0 + li);
  const u32x4 w = stream_block(seed, gid, PURPOSE_RESAMPLE, iter, 0);
  const double t = u52(w.x, w.y) * totals[0];
  const int64_t idx = resample_search(t, B, nb, cm, cum, n);
  if (!GATHER) { idx_out[li] = idx; return; }
  int64_t r, o;
  split_index(idx, g.cap, r, o);
  const double *src = shard_block(g, r) + o;               // the drawn particle, in a gathered copy or in its owner's HBM
  for (int row = 0; row < d + s; ++row)
    dst.pop[(int64_t)row * dst.cap + li] = src[(int64_t)row * g.cap];
}

// The draw on packed lines.  Chunk by the offsets `B` (as resample_search: the oracle's answer is defined per chunk); inside
// the chunk the first line whose end value exceeds t is found AROUND a guess: guide[bucket of t] (k_scan_final) is a line
// whose running sums reach t's bucket -- with n_lines buckets usually the line itself or a neighbour --, the ends of that
// line and of the one before it are read together, and the search walks from there in whichever direction they say
// (the ends are non-decreasing inside a chunk, so the walk is the search).  Then the packed line: the running sums of its
// first PG - 1 slots pick the slot, and the caller reads that slot's row.  Per draw: the guide entry, two line ends, PG - 1
// running sums, the row -- ~8 load instructions in 4 dependent trips, the last two to one line (the binary search through
// two index levels was ~27 in ~11; the kernel is bound by the number of divergent-address loads).  Same decisions as resample_search on the same
// numbers: if no running sum of the chunk exceeds t (rounding of the offsets against the sums) the next chunk's first
// particle is taken, the last particle at the end of the population.  `row` receives the address of the drawn particle's row.
template <int PG>
__device__ __forceinline__ int64_t packed_search(const double t, const double total, const double *B, const int64_t nb,
                                                 const int32_t *__restrict__ guide, const double *__restrict__ ge,
                                                 const double *__restrict__ pk, const int64_t n, const double *&row) {
  constexpr int kStride = 16 / PG;
  constexpr int64_t kLinesPerChunk = kScanChunk / PG;
  const int64_t n_lines = (n + PG - 1) / PG;
  int64_t s = guide[guide_bucket(t, total, n_lines)];     // in flight during the search of the offsets
  int64_t blo = 0, bhi = nb;
  while (blo < bhi) {
    const int64_t mid = blo + ((bhi - blo) >> 1);
    if (B[mid] > t) bhi = mid; else blo = mid + 1;
  }
  const int64_t chunk = blo - 1;
  const int64_t l0 = chunk * kLinesPerChunk;
  int64_t end = l0 + kLinesPerChunk;
  if (end > n_lines) end = n_lines;
  s = s < l0 ? l0 : (s > end - 1 ? end - 1 : s);
  const double e_prev = s > l0 ? ge[s - 1] : -INFINITY;
  double e_s = ge[s];
  if (e_prev > t) {                                       // the guess lies behind the line: walk back
    s -= 1;
    while (s > l0 && ge[s - 1] > t) s -= 1;
  } else {
    int walked = 0;
    while (!(e_s > t) && s + 1 < end) {
      if (++walked > 8) {                                 // a bucket full of all-but-weightless lines: bisect the rest of the chunk
        int64_t lo = s + 1, hi = end;                     // first line in [lo, hi) whose end exceeds t, or `end`
        while (lo < hi) {
          const int64_t mid = lo + ((hi - lo) >> 1);
          if (ge[mid] > t) hi = mid; else lo = mid + 1;
        }
        s = lo;
        e_s = s < end ? INFINITY : -INFINITY;
        break;
      }
      s += 1;
      e_s = ge[s];
    }
    if (!(e_s > t)) s = end;                              // no line of the chunk exceeds t
  }
  int64_t idx;
  if (s < end) {
    // the slot: the COUNT of the running sums of the line's first PG - 1 slots that do not exceed t (independent reads;
    // the line's end exceeds t, so its last slot needs no test)
    double cw[PG > 1 ? PG - 1 : 1];
#pragma unroll
    for (int q = 0; q < PG - 1; ++q) cw[q] = pk[s * 16 + q * kStride];
    int slot = 0;
#pragma unroll
    for (int q = 0; q < PG - 1; ++q) slot += !(cw[q] > t) ? 1 : 0;
    idx = s * PG + slot;
    if (idx >= n) idx = n - 1;                            // (the last line; its empty slots hold +inf)
  } else {
    idx = (chunk + 1) * (int64_t)kScanChunk;
    if (idx >= n) idx = n - 1;
  }
  const int64_t line = idx / PG;
  row = pk + line * 16 + (idx - line * PG) * kStride + 1;
  return idx;
}

constexpr int packed_per_line(int row_len) {
  return 16 / (1 + row_len) >= 4 ? 4 : 16 / (1 + row_len) >= 2 ? 2 : 1;
}

// One shard, packed: the draw and its (theta, u) row come from one line, and -- with D, S known at compile time -- the
// moment sums of the RESAMPLED population (what k_stats would compute in a pass of its own: Sigma, eps and the history
// row are taken from the resampled population, :348-353) come out of the same kernel, in the same per-workgroup order.
template <int D, int S>
__global__ void __launch_bounds__(kBlock)
k_resample_gather_stats(const uint64_t seed, const double *__restrict__ pk, const double *__restrict__ ge,
                        const int32_t *__restrict__ guide, const int pg, const int64_t n, const double *__restrict__ bs, const int64_t nb, const double *__restrict__ totals,
                        const uint64_t iter, const PopPtrs dst, const ControlBlock *__restrict__ cb,
                        double *__restrict__ partials) {
  constexpr int NP = n_partials(D, S);
  extern __shared__ double bs_lds[];
  const bool in_lds = nb <= kGatherCoarseMax;
  if (in_lds) {
    stage_chunk_offsets(bs, nb, bs_lds);
    __syncthreads();
  }
  const double *B = in_lds ? bs_lds : bs;
  double acc[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) acc[q] = 0.0;
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (li < dst.n_local) {
    const uint64_t gid = (uint64_t)(dst.gid0 + li);
    const u32x4 w = stream_block(seed, gid, PURPOSE_RESAMPLE, iter, 0);
    const double total = totals[0];
    const double t = u52(w.x, w.y) * total;
    constexpr int PG = packed_per_line(D + S);
    const double *row;
    (void)packed_search<PG>(t, total, B, nb, guide, ge, pk, n, row);
    double th[D], u[S], rho[S];
#pragma unroll
    for (int k = 0; k < D; ++k) { th[k] = row[k]; dst.pop[(int64_t)k * dst.cap + li] = th[k]; }
#pragma unroll
    for (int j = 0; j < S; ++j) {
      u[j] = row[D + j];
      dst.pop[(int64_t)(D + j) * dst.cap + li] = u[j];
      rho[j] = dst.rho[(int64_t)j * dst.cap + li];                        // rho stays where it is (:131-132)
    }
    moment_terms<D, S>(cb->pivot, false, th, u, rho, acc);
  }
  block_reduce_store<NP>(acc, partials + (int64_t)blockIdx.x * NP);
}

// the same without the sums, d and s at run time (host-callback and source-compiled simulators)
__global__ void __launch_bounds__(kBlock)
k_resample_gather_packed(const uint64_t seed, const int row_len, const double *__restrict__ pk, const double *__restrict__ ge,
                         const int32_t *__restrict__ guide, const int pg, const int64_t n, const double *__restrict__ bs, const int64_t nb,
                         const double *__restrict__ totals, const uint64_t iter, const PopPtrs dst) {
  extern __shared__ double bs_lds[];
  const bool in_lds = nb <= kGatherCoarseMax;
  if (in_lds) {
    stage_chunk_offsets(bs, nb, bs_lds);
    __syncthreads();
  }
  const double *B = in_lds ? bs_lds : bs;
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (li >= dst.n_local) return;
  const u32x4 w = stream_block(seed, (uint64_t)(dst.gid0 + li), PURPOSE_RESAMPLE, iter, 0);
  const double total = totals[0];
  const double t = u52(w.x, w.y) * total;
  const double *row;
  if (pg == 4) (void)packed_search<4>(t, total, B, nb, guide, ge, pk, n, row);
  else if (pg == 2) (void)packed_search<2>(t, total, B, nb, guide, ge, pk, n, row);
  else (void)packed_search<1>(t, total, B, nb, guide, ge, pk, n, row);
  for (int r = 0; r < row_len; ++r) dst.pop[(int64_t)r * dst.cap + li] = row[r];
}

// ---- the sharded resample: requests grouped by owner, served by the owner, scattered by the requester ----
// counts[r] += number of draws whose source lives on shard r (wave-aggregated integer atomics)
__global__ void __launch_bounds__(kBlock)
k_bucket_count(const int64_t *__restrict__ idx, const int64_t n_local, const int64_t cap, unsigned long long *counts) {
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = li < n_local;
  int64_t r = 0, o = 0;
  if (live) split_index(idx[li], cap, r, o);
  unsigned long long todo = __ballot(live);
  const int lane = threadIdx.x & 63;
  while (todo) {                                   // one trip per distinct owner in the wave (<= world)
    const int leader = __ffsll((long long)todo) - 1;
    const int64_t r0 = __shfl(r, leader, 64);
    const unsigned long long same = __ballot(live && r == r0);
    if (lane == leader) atomicAdd(&counts[r0], (unsigned long long)__popcll(same));
    todo &= ~same;
  }
}

// cursor[r] starts at the exclusive offset of bucket r; req[pos] = offset inside the owner (exact as a double),
// slot[pos] = the local destination the reply belongs to.  The order inside a bucket is arbitrary (atomics); it only
// pairs a request with its reply.
__global__ void __launch_bounds__(kBlock)
k_bucket_scatter(const int64_t *__restrict__ idx, const int64_t n_local, const int64_t cap, unsigned long long *cursor,
                 double *__restrict__ req, int64_t *__restrict__ slot) {
  const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = li < n_local;
  int64_t r = 0, o = 0;
  if (live) split_index(idx[li], cap, r, o);
  unsigned long long todo = __ballot(live);
  const int lane = threadIdx.x & 63;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int64_t r0 = __shfl(r, leader, 64);
    const unsigned long long same = __ballot(live && r == r0);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(&cursor[r0], (unsigned long long)__popcll(same));
    base = __shfl(base, leader, 64);
    if (live && r == r0) {
      const int64_t pos = (int64_t)base + __popcll(same & below);
      req[pos] = (double)o;
      slot[pos] = li;
    }
    todo &= ~same;
  }
}

// owner side: rows_out[q][row] = pop[row][offset_q] for the m requested offsets (AoS: one contiguous row per request)
__global__ void __launch_bounds__(kBlock)
k_resample_serve(const double *__restrict__ req, const int64_t m, const int row_len, const PopPtrs src,
                 double *__restrict__ rows_out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= m * row_len) return;
  const int64_t q = e / row_len;
  const int row = (int)(e - q * row_len);
  int64_t o = (int64_t)req[q];
  o = o < 0 ? 0 : (o >= src.n_local ? src.n_local - 1 : o);      // a corrupt request must not fault
  rows_out[e] = src.pop[(int64_t)row * src.cap + o];
}

// requester side: rows_in is in the bucket order of k_bucket_scatter
__global__ void __launch_bounds__(kBlock)
k_resample_scatter(const double *__restrict__ rows_in, const int64_t *__restrict__ slot, const int64_t n_local,
                   const int row_len, const PopPtrs dst) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n_local * row_len) return;
  const int64_t pos = e / row_len;
  const int row = (int)(e - pos * row_len);
  dst.pop[(int64_t)row * dst.cap + slot[pos]] = rows_in[e];
}

// ------------------------------------------------------------------------------------------
// K2: ECDF knots from a sorted column (cdf_estimators.jl:29-33)
// ------------------------------------------------------------------------------------------
__global__ void k_cdf_meta(const double *__restrict__ sorted, const int64_t n, int64_t *__restrict__ meta) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t lo = 0, hi = n;             // first index with sorted[i] > 0
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (sorted[mid] > 0.0) hi = mid; else lo = mid + 1;
  }
  meta[0] = lo;
  meta[1] = (n > 0 && sorted[0] < 0.0) ? 1 : 0;
}

__global__ void __launch_bounds__(kBlock)
k_cdf_fill(const double *__restrict__ sorted, const int64_t n, const int64_t *__restrict__ meta,
           double *__restrict__ knots) {
  const int64_t z = meta[0], mpos = n - z;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < mpos) knots[1 + i] = sorted[z + i];
  if (i == 0) {
    knots[0] = 0.0;
    if (mpos > 0) knots[mpos + 1] = sorted[n - 1] * 1.5;
  }
}

__global__ void __launch_bounds__(kBlock)
k_cdf_index(double *__restrict__ knots, const int64_t len, const int64_t stride, const int shift, double *__restrict__ coarse,
            const int n_coarse, double *__restrict__ mid, const int64_t mid_len) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k < n_coarse) {
    const int64_t p = k << shift;
    coarse[k] = p < len ? knots[p] : INFINITY;
  }
  if (k < mid_len) {
    const int64_t p = k << kCdfLineShift;
    mid[k] = p < len ? knots[p] : INFINITY;
  }
  if (len + k < stride) knots[len + k] = INFINITY;       // the searches read up to 15 knots past the last one
}

__global__ void __launch_bounds__(kBlock)
k_compact_column(const ShardBlocks g, const int stat, const int64_t n, double *__restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= n) return;
  int64_t r, o;
  split_index(gid, g.cap, r, o);
  out[gid] = shard_block(g, r)[(int64_t)stat * g.cap + o];
}

// ------------------------------------------------------------------------------------------
// operators exposed on their own
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_cdf_eval(const double *__restrict__ knots, const int64_t len, const double *__restrict__ q, const int64_t m,
           double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < m) out[i] = cdf_apply(knots, len, q[i]);
}

__global__ void __launch_bounds__(kBlock)
k_cdf_apply_matrix(const CdfPtrs cdf, const int s, const double *__restrict__ rho, const int64_t m,
                   double *__restrict__ u) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  for (int j = 0; j < s; ++j)
    u[(int64_t)j * m + i] = cdf_apply(cdf.knots + (int64_t)j * cdf.stride, cdf.len[j], rho[(int64_t)j * m + i]);
}

// rand(prior) and its log density for particle ids pid0.. (sabc_op_prior)
__global__ void __launch_bounds__(kBlock)
k_prior_op(const ModelDesc m, const uint64_t pid0, const int64_t n, double *__restrict__ theta, double *__restrict__ lp) {
  rng_tables_init();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double th[kMaxPara];
  if (m.prior_joint) mvnormal_sample(m, m.d, pid0 + (uint64_t)i, th);
  else
    for (int k = 0; k < m.d; ++k) th[k] = prior_sample_dim(m, k, pid0 + (uint64_t)i);
  for (int k = 0; k < m.d; ++k) theta[(int64_t)k * n + i] = th[k];
  lp[i] = prior_logpdf_rt(m, th);
}

__global__ void k_philox_debug(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k,
                               uint32_t *words, double *normals) {
  rng_tables_init();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const u32x4 w = stream_block(seed, pid, purpose, iter, k);
  words[0] = w.x; words[1] = w.y; words[2] = w.z; words[3] = w.w;
  box_muller(w, normals[0], normals[1]);
}

// Pure generator loop: `pairs` Philox blocks + Box-Muller pairs per lane, nothing else (one store at the
// end keeps it alive).  Its rate is the VALU ceiling for any simulator that consumes normals from this
// generator; bench.py quotes k_update's in-kernel normal rate against it.
__global__ void __launch_bounds__(kBlock)
k_rng_peak(const uint64_t seed, const int pairs, const int64_t n, double *__restrict__ out) {
  rng_tables_init();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
#pragma unroll SABC_SIM_UNROLL
  for (int k = 0; k < pairs; ++k) {
    double z0, z1;
    box_muller(stream_block(seed, (uint64_t)i, PURPOSE_SIM, 0, (uint32_t)k), z0, z1);
    acc += z0;
    acc += z1;
  }
  out[i] = acc;
}

__global__ void __launch_bounds__(kBlock)
k_normal_pairs(uint64_t seed, uint64_t pid0, uint32_t purpose, uint64_t iter, uint32_t k, int64_t m, double *out) {
  rng_tables_init();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  double z0, z1;
  box_muller(stream_block(seed, pid0 + (uint64_t)i, purpose, iter, k), z0, z1);
  out[2 * i] = z0;
  out[2 * i + 1] = z1;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
#define SABC_LAUNCH_RC() ((int)hipGetLastError())

// launch of a kernel from the run-time compiled module of a user simulator (rtc.hpp): same argument list as the
// template it was instantiated from; timing events ride on the dispatch packet like hipExtLaunchKernelGGL's
template <class... A>
static int module_launch(hipFunction_t f, unsigned grid, unsigned block, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                         A... a) {
  void *args[] = {(void *)&a...};
  if (ev0)
    return (int)hipExtModuleLaunchKernel(f, grid * block, 1, 1, block, 1, 1, 0, stream, args, nullptr, ev0, ev1, 0);
  return (int)hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, stream, args, nullptr);
}

// dispatch on the (model, d, s) combinations that exist
#define SABC_DISPATCH_MODEL(m, CALL)                                                              \
  do {                                                                                            \
    const int key_ = (m).model_id * 100 + (m).d * 10 + (m).s;                                     \
    switch (key_) {                                                                               \
      case SABC_MODEL_GAUSS_IID * 100 + 11: { CALL(SABC_MODEL_GAUSS_IID, 1, 1); break; }          \
      case SABC_MODEL_GAUSS_IID * 100 + 12: { CALL(SABC_MODEL_GAUSS_IID, 1, 2); break; }          \
      case SABC_MODEL_GAUSS_IID * 100 + 21: { CALL(SABC_MODEL_GAUSS_IID, 2, 1); break; }          \
      case SABC_MODEL_GAUSS_IID * 100 + 22: { CALL(SABC_MODEL_GAUSS_IID, 2, 2); break; }          \
      case SABC_MODEL_GAUSS2D * 100 + 23: { CALL(SABC_MODEL_GAUSS2D, 2, 3); break; }              \
      case SABC_MODEL_LV * 100 + 34: { CALL(SABC_MODEL_LV, 3, 4); break; }                        \
      default: return (int)hipErrorInvalidValue;                                                  \
    }                                                                                             \
  } while (0)

inline unsigned gk_blocks(int64_t n) { return (unsigned)((n + kGkPerBlock - 1) / kGkPerBlock); }                       // k_simulate_gk
inline unsigned gk_update_blocks(int64_t n) { return (unsigned)((n + kGkUpdatePerBlock - 1) / kGkUpdatePerBlock); }   // k_update_gk

int launch_prior_simulate(const ModelDesc &m, PopPtrs pp, hipStream_t stream, const RtcKernels *rtc) {
  if (pp.n_local <= 0) return 0;
  if (m.model_id == SABC_MODEL_USER) {
    if (!rtc || !rtc->prior_simulate) return (int)hipErrorInvalidValue;
    return module_launch(rtc->prior_simulate, (unsigned)n_blocks(pp.n_local), kBlock, stream, nullptr, nullptr, m, pp);
  }
  if (m.model_id == SABC_MODEL_GK) {
    hipLaunchKernelGGL(k_simulate_gk, dim3(gk_blocks(pp.n_local)), dim3(kBlock), 0, stream, m, (const double *)nullptr,
                       pp.n_local, pp.cap, (uint64_t)pp.gid0, (uint64_t)0, 1, pp.pop, pp.rho, pp.cap);
    return SABC_LAUNCH_RC();
  }
  const dim3 grid((unsigned)n_blocks(pp.n_local)), block(kBlock);
#define CALL(M, D, S) hipLaunchKernelGGL((k_prior_simulate<M, D, S>), grid, block, 0, stream, m, pp)
  SABC_DISPATCH_MODEL(m, CALL);
#undef CALL
  return SABC_LAUNCH_RC();
}

int launch_cdf_population(const ModelDesc &m, PopPtrs pp, CdfPtrs cdf, hipStream_t stream) {
  if (pp.n_local <= 0) return 0;
  hipLaunchKernelGGL(k_cdf_population, dim3((unsigned)n_blocks(pp.n_local)), dim3(kBlock), 0, stream, m.d, m.s, pp, cdf);
  return SABC_LAUNCH_RC();
}

// workgroups (= partial rows) of one k_update launch over act_n particles
int64_t update_rows(const ModelDesc &m, int64_t act_n) {
  if (act_n <= 0) return 0;
  return m.model_id == SABC_MODEL_GK ? (int64_t)gk_update_blocks(act_n)   // 4 waves x kGkParticlesPerWave particles per workgroup
                                     : (act_n + update_block_threads(m.s) - 1) / update_block_threads(m.s);   // one thread per particle
}

// ev0 / ev1 (optional): timing events attached to the dispatch packet itself (hipExtLaunchKernel), so that
// measuring the kernel does not put separate marker packets into the queue
#define SABC_LAUNCH_UPDATE(KERNEL, GRID)                                                                         \
  do {                                                                                                           \
    if (ev0) hipExtLaunchKernelGGL((KERNEL), (GRID), block, 0, stream, ev0, ev1, 0, m, c, cb, pp, cdf, pv, act_lo, act_n, out); \
    else hipLaunchKernelGGL((KERNEL), (GRID), block, 0, stream, m, c, cb, pp, cdf, pv, act_lo, act_n, out);      \
  } while (0)

// the persistent form exists for the built-in simulators with one lane per particle.  Its workgroups must all be resident at
// once: at most 256 of them, one per CU (SABC_PERSISTENT_WG lowers that).  Measured against the launch chain, cfg2, us per
// population update (tools/sweep_small.sh): RandomWalk n = 1000: 16.3 | 22.6, 5000: 17.0 | 23.4, 10 000: 17.7 | 23.5, 16 384:
// 18.3 | 23.5, 32 768: 19.2 | 23.7, 62 500: 22.0 | 23.8; DifferentialEvolution 1000: 29.9 | 39.5, 16 384: 33.6 | 41.1, 62 500:
// 38.4 | 41.5.  (A first version fenced every barrier -- a write-back and an invalidate of the L2 per update -- and lost from
// 64 workgroups on: 25.8 us at n = 16 384, 50.6 at 62 500; what crosses between workgroups now goes past the caches.)
static int64_t persist_max_workgroups() {
  static const int64_t v = [] { const char *e = std::getenv("SABC_PERSISTENT_WG"); const long long x = e ? std::atoll(e) : 256; return x < 0 ? 0 : x > 256 ? 256 : x; }();
  return v;
}
// lanes per particle of the persistent form: a TEAM of 16 (a row of the wave) or 4 (a quad) shares a particle's generator work
// (update_kernel.hpp, LANES) while that many times the workgroups still fit the launch -- the device is then so empty that the
// extra waves run on idle SIMDs and a particle's serial chain is what an update waits for --, else 1.
// SABC_PERSISTENT_LANES = 1 | 4 | 16 overrides (a team: where it fits, else the next smaller).
static int persist_lanes_env() {                     // (read at every call: a process may run the forms side by side)
  const char *e = std::getenv("SABC_PERSISTENT_LANES");
  const int x = e ? std::atoi(e) : 0;
  return x == 1 || x == 4 || x == 16 ? x : 0;
}
static int64_t persist_team_max_particles(int lanes) {   // per launch (a half batch for DifferentialEvolution / StretchMove)
  const char *e = std::getenv(lanes == 16 ? "SABC_PERSISTENT_LANES16_MAX" : "SABC_PERSISTENT_LANES4_MAX");
  const long long x = e ? std::atoll(e) : (lanes == 16 ? 2048 : 16384);
  return x < 0 ? 0 : x;
}
int64_t persistent_workgroups(const ModelDesc &m, int prop_kind, int64_t act_n, const RtcKernels *rtc, int *lanes_out, int *active_out) {
  const int64_t B = update_block_threads(m.s);
  if (lanes_out) *lanes_out = 1;
  if (active_out) *active_out = (int)B;
  if (prop_kind < 0 || prop_kind > 2 || act_n < 2) return 0;
  bool have4 = true, have16 = true;
  if (m.model_id == SABC_MODEL_USER) {                 // a simulator from source: compiled with it (rtc.cpp), where its shape fits
    if (!rtc || !rtc->persistent[prop_kind]) return 0;
    have4 = rtc->persistent4[prop_kind] != nullptr;
    have16 = rtc->persistent16[prop_kind] != nullptr;
  } else if (!(m.model_id == SABC_MODEL_GAUSS_IID || m.model_id == SABC_MODEL_GAUSS2D || m.model_id == SABC_MODEL_LV)) {
    return 0;
  }
  if (!persistent_fits(m.d, m.s)) return 0;
  const int64_t per_launch = prop_kind == SABC_PROP_RANDOMWALK ? act_n : act_n - act_n / 2;     // the larger half batch
  // a wave per SIMD at most, and one wave of the workgroup without particles: the control wave (persistent_kernel.hpp)
  const int64_t thin = B - 64 < 256 ? B - 64 : 256;
  const int want = persist_lanes_env();
  const bool may16 = have16 && (want == 16 || (want == 0 && per_launch <= persist_team_max_particles(16)));
  const bool may4 = have4 && (want == 4 || want == 16 || (want == 0 && per_launch <= persist_team_max_particles(4)));
  // the thinnest spread whose workgroups fit the launch: a row per particle before a quad before a lane, a wave per SIMD before
  // the whole block
  const int64_t lanes_try[6] = {16, 16, 4, 4, 1, 1}, active_try[6] = {thin, B, thin, B, thin, B};
  for (int i = 0; i < 6; ++i) {
    if ((lanes_try[i] == 16 && !may16) || (lanes_try[i] == 4 && !may4)) continue;
    const int64_t wg = (lanes_try[i] * per_launch + active_try[i] - 1) / active_try[i];
    if (wg > persist_max_workgroups()) continue;
    if (lanes_out) *lanes_out = (int)lanes_try[i];
    if (active_out) *active_out = (int)active_try[i];
    return wg;
  }
  return 0;
}

#ifdef SABC_PERSIST_TRACE
extern "C" __attribute__((visibility("default"))) int sabc_debug_persist_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_persist_trace), sizeof(unsigned long long) * 64 * 16);
}
#endif

// the most workgroups a persistent launch over a shard of at most `cap` particles can have (sizes the partial rows)
int64_t persistent_workgroups_bound(const ModelDesc &m, int64_t cap) {
  if (!persistent_fits(m.d, m.s)) return 0;
  const int64_t A = update_block_threads(m.s) - 64 < 256 ? update_block_threads(m.s) - 64 : 256, wg16 = (16 * cap + A - 1) / A;
  return wg16 < persist_max_workgroups() ? wg16 : persist_max_workgroups();
}

int launch_update_persistent(const ModelDesc &m, int prop_kind, const PersistArgs &pa_in, ControlBlock *cb, PopPtrs pp, CdfPtrs cdf,
                             PartnerView pv_a, PartnerView pv_b, double *partials, double *hist, Mailbox *mbox, double *stage,
                             hipStream_t stream, const RtcKernels *rtc) {
  int lanes = 1, active = 0;
  const int64_t wg = persistent_workgroups(m, prop_kind, pa_in.act_n, rtc, &lanes, &active);
  if (wg <= 0) return (int)hipErrorInvalidValue;
  PersistArgs pa = pa_in;
  pa.active = active;
  pa.ctrl_wave = active < (int)update_block_threads(m.s) ? active / 64 : -1;
  const dim3 grid((unsigned)wg), block((unsigned)update_block_threads(m.s));
  if (m.model_id == SABC_MODEL_USER)
    return module_launch(lanes == 16 ? rtc->persistent16[prop_kind] : lanes == 4 ? rtc->persistent4[prop_kind] : rtc->persistent[prop_kind], grid.x, block.x, stream, nullptr, nullptr, m, pa,
                         cb, pp, cdf, pv_a, pv_b, partials, hist, mbox, stage);
#define PCALLL(M, D, S, P, L) hipLaunchKernelGGL((k_update_persistent<M, D, S, P, L>), grid, block, 0, stream, m, pa, cb, pp, cdf, pv_a, pv_b, partials, hist, mbox, stage)
#define PCALLP(M, D, S, P)                      \
  if (lanes == 16) PCALLL(M, D, S, P, 16);      \
  else if (lanes == 4) PCALLL(M, D, S, P, 4);   \
  else PCALLL(M, D, S, P, 1)
#define PCALL(M, D, S)                                                          \
  switch (prop_kind) {                                                          \
    case SABC_PROP_RANDOMWALK: PCALLP(M, D, S, SABC_PROP_RANDOMWALK); break;     \
    case SABC_PROP_DIFFEVO: PCALLP(M, D, S, SABC_PROP_DIFFEVO); break;           \
    default: PCALLP(M, D, S, SABC_PROP_STRETCH); break;                          \
  }
  if (m.model_id == SABC_MODEL_GAUSS_IID) {
    if (m.d == 1 && m.s == 1) { PCALL(SABC_MODEL_GAUSS_IID, 1, 1); }
    else if (m.d == 1 && m.s == 2) { PCALL(SABC_MODEL_GAUSS_IID, 1, 2); }
    else if (m.d == 2 && m.s == 1) { PCALL(SABC_MODEL_GAUSS_IID, 2, 1); }
    else { PCALL(SABC_MODEL_GAUSS_IID, 2, 2); }
  } else if (m.model_id == SABC_MODEL_GAUSS2D) {
    PCALL(SABC_MODEL_GAUSS2D, 2, 3);
  } else {
    PCALL(SABC_MODEL_LV, 3, 4);
  }
#undef PCALL
#undef PCALLP
#undef PCALLL
  return SABC_LAUNCH_RC();
}

int launch_update(const ModelDesc &m, const StepArgs &c, const ControlBlock *cb, PopPtrs pp, CdfPtrs cdf, PartnerView pv,
                  int64_t act_lo, int64_t act_n, double *partials, int64_t row0, hipStream_t stream, hipEvent_t ev0,
                  hipEvent_t ev1, const RtcKernels *rtc) {
  if (act_n <= 0) return 0;
  const dim3 grid((unsigned)update_rows(m, act_n)), block(m.model_id == SABC_MODEL_GK ? kBlock : update_block_threads(m.s));
  double *out = partials + row0 * n_partials(m.d, m.s);
  if (m.model_id == SABC_MODEL_USER) {
    if (!rtc || c.prop_kind < 0 || c.prop_kind > 2 || !rtc->update[c.prop_kind]) return (int)hipErrorInvalidValue;
    return module_launch(rtc->update[c.prop_kind], grid.x, update_block_threads(m.s), stream, ev0, ev1, m, c, cb, pp, cdf, pv, act_lo, act_n, out);
  }
  if (m.model_id == SABC_MODEL_GK) {
    const dim3 g((unsigned)update_rows(m, act_n));
    // wanted ranks that are all multiples of 16 (BASELINE config 4): FOUR particles at a time, one per row of 16 lanes, eight
    // values per lane -- 15 of the network's 24 steps stay inside the lane (device_models.hpp: gk_simulate_rows4)
    bool rows4 = SABC_GK_ROWS4 != 0;
    for (int j = 0; j < kGkS; ++j) rows4 = rows4 && (((int)m.p[2 + j]) & 15) == 0 && (int)m.p[2 + j] >= 16 && (int)m.p[2 + j] <= 128;
    switch (c.prop_kind * 2 + (rows4 ? 1 : 0)) {
      case SABC_PROP_RANDOMWALK * 2: SABC_LAUNCH_UPDATE((k_update_gk<SABC_PROP_RANDOMWALK, false>), g); break;
      case SABC_PROP_RANDOMWALK * 2 + 1: SABC_LAUNCH_UPDATE((k_update_gk<SABC_PROP_RANDOMWALK, true>), g); break;
      case SABC_PROP_DIFFEVO * 2: SABC_LAUNCH_UPDATE((k_update_gk<SABC_PROP_DIFFEVO, false>), g); break;
      case SABC_PROP_DIFFEVO * 2 + 1: SABC_LAUNCH_UPDATE((k_update_gk<SABC_PROP_DIFFEVO, true>), g); break;
      case SABC_PROP_STRETCH * 2: SABC_LAUNCH_UPDATE((k_update_gk<SABC_PROP_STRETCH, false>), g); break;
      case SABC_PROP_STRETCH * 2 + 1: SABC_LAUNCH_UPDATE((k_update_gk<SABC_PROP_STRETCH, true>), g); break;
      default: return (int)hipErrorInvalidValue;
    }
    return SABC_LAUNCH_RC();
  }
#define CALLP(M, D, S, P) SABC_LAUNCH_UPDATE((k_update<M, D, S, P>), grid)
#define CALL(M, D, S)                                                           \
  switch (c.prop_kind) {                                                        \
    case SABC_PROP_RANDOMWALK: CALLP(M, D, S, SABC_PROP_RANDOMWALK); break;     \
    case SABC_PROP_DIFFEVO: CALLP(M, D, S, SABC_PROP_DIFFEVO); break;           \
    case SABC_PROP_STRETCH: CALLP(M, D, S, SABC_PROP_STRETCH); break;           \
    default: return (int)hipErrorInvalidValue;                                  \
  }
  SABC_DISPATCH_MODEL(m, CALL);
#undef CALL
#undef CALLP
  return SABC_LAUNCH_RC();
}

int launch_host_prior(const ModelDesc &m, PopPtrs pp, hipStream_t stream) {
  if (pp.n_local <= 0) return 0;
  hipLaunchKernelGGL(k_host_prior, dim3((unsigned)n_blocks(pp.n_local)), dim3(kBlock), 0, stream, m, pp);
  return SABC_LAUNCH_RC();
}

int launch_host_propose(const ModelDesc &m, const StepArgs &c, const ControlBlock *cb, PopPtrs pp, PartnerView pv,
                        int64_t act_lo, int64_t act_n, double *thp, double *aux, double *thp_host, unsigned char *gate_host,
                        double *cur_out, unsigned int *done, unsigned long long *flag, unsigned long long seq, int64_t chunk,
                        hipStream_t stream) {
  if (act_n <= 0) return 0;
  HostSignal sig;
  sig.done = done; sig.flag = flag; sig.seq = seq; sig.chunk = chunk;
  hipLaunchKernelGGL(k_host_propose, dim3((unsigned)n_blocks(act_n)), dim3(kBlock), 0, stream, m, c, cb, pp, pv, act_lo, act_n,
                     thp, aux, thp_host, gate_host, cur_out, sig);
  return SABC_LAUNCH_RC();
}

int launch_host_accept(const ModelDesc &m, const StepArgs &c, const ControlBlock *cb, PopPtrs pp, CdfPtrs cdf, int64_t act_lo,
                       int64_t act_n, int64_t t_lo, int64_t t_n, const double *thp, const double *aux, const double *rho_prop,
                       const double *lp_host, unsigned long long *n_accept, hipStream_t stream) {
  if (t_n <= 0) return 0;
  hipLaunchKernelGGL(k_host_accept, dim3((unsigned)n_blocks(t_n)), dim3(kBlock), 0, stream, m, c, cb, pp, cdf, act_lo, act_n,
                     t_lo, t_n, thp, aux, rho_prop, lp_host, n_accept);
  return SABC_LAUNCH_RC();
}

int launch_stats_rt(const ModelDesc &m, const ControlBlock *cb, PopPtrs pp, double *partials, unsigned long long *n_accept,
                    hipStream_t stream) {
  if (pp.n_local <= 0) return 0;
  hipLaunchKernelGGL(k_stats_rt, dim3((unsigned)n_blocks(pp.n_local)), dim3(kBlock), 0, stream, m.d, m.s, cb, pp, partials,
                     n_accept);
  return SABC_LAUNCH_RC();
}

int launch_stats(const ModelDesc &m, const ControlBlock *cb, PopPtrs pp, double *partials, hipStream_t stream,
                 const RtcKernels *rtc) {
  if (pp.n_local <= 0) return 0;
  if (m.model_id == SABC_MODEL_HOST) return launch_stats_rt(m, cb, pp, partials, nullptr, stream);
  if (m.model_id == SABC_MODEL_USER) {
    if (!rtc || !rtc->stats) return (int)hipErrorInvalidValue;
    return module_launch(rtc->stats, (unsigned)n_blocks(pp.n_local), kBlock, stream, nullptr, nullptr, cb, pp, partials);
  }
  const dim3 grid((unsigned)n_blocks(pp.n_local)), block(kBlock);
#define CALL(M, D, S) hipLaunchKernelGGL((k_stats<D, S>), grid, block, 0, stream, cb, pp, partials)
  if (m.model_id == SABC_MODEL_GK) { CALL(SABC_MODEL_GK, 4, 4); return SABC_LAUNCH_RC(); }
  SABC_DISPATCH_MODEL(m, CALL);
#undef CALL
  return SABC_LAUNCH_RC();
}

int launch_reduce_partials(const double *partials, int64_t rows, int np, double *sums, const int *halt,
                           hipStream_t stream) {
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)np), dim3(kBlock), 0, stream, partials, rows, np, sums, halt);
  return SABC_LAUNCH_RC();
}

int launch_reduce_control(const double *partials, int64_t rows, int np, double *stage, bool reduce_guarded,
                          ControlBlock *cb, const ControlArgs &a, double *hist, Mailbox *mbox, hipStream_t stream,
                          const P2PView *pv, uint32_t seq, bool do_control, int silent) {
  (void)reduce_guarded;   // a guarded reduction is always paired with a guarded control step, which is what decides
  XchgArgs x;
  std::memset(&x, 0, sizeof(x));
  x.do_control = do_control ? 1 : 0;
  // a short matrix of partial rows (a shard of an 8-GPU run: 489 rows at n = 1e6) takes 4 waves instead of 16: the waves of
  // one workgroup start one after the other and the launch waits for the last one's loads (tools/rc_timing.py)
  static const int forced = [] { const char *e = std::getenv("SABC_RC_BLOCK"); const int v = e ? std::atoi(e) : 0; return (v == 256 || v == 1024) ? v : 0; }();
  // (np <= 256 threads' worth: the kernel gives every component of the row a lane)
  const int block = forced && np <= 256 ? forced : ((rows < 0 && np <= 256) || (np <= 64 && rows >= 0 && rows <= (int64_t)24 * (256 / np))) ? 256 : 1024;
  if (pv) {
    x.pv = *pv; x.seq = seq; x.silent = silent;
    hipLaunchKernelGGL(k_reduce_control<true>, dim3(1), dim3(block), 0, stream, partials, rows, np, stage, cb, a, hist, mbox, x);
  } else {
    hipLaunchKernelGGL(k_reduce_control<false>, dim3(1), dim3(block), 0, stream, partials, rows, np, stage, cb, a, hist, mbox, x);
  }
  return SABC_LAUNCH_RC();
}

int launch_p2p_barrier(const P2PView &pv, uint32_t seq, ControlBlock *cb, bool guarded, int silent, hipStream_t stream) {
  hipLaunchKernelGGL(k_p2p_barrier, dim3(1), dim3(64), 0, stream, pv, seq, cb, guarded ? 1 : 0, silent);
  return SABC_LAUNCH_RC();
}

int launch_p2p_commit(const P2PView &pv, uint32_t call, int status, bool wait, ControlBlock *cb, int silent, hipStream_t stream) {
  hipLaunchKernelGGL(k_p2p_commit, dim3(1), dim3(64), 0, stream, pv, call, status, wait ? 1 : 0, cb, silent);
  return SABC_LAUNCH_RC();
}

int launch_p2p_selftest(const P2PView &pv, uint32_t seq, int np, const double *in, double *out, int *failed, int silent,
                        hipStream_t stream) {
  hipLaunchKernelGGL(k_p2p_selftest, dim3(1), dim3(1024), 0, stream, pv, seq, np, in, out, failed, silent);
  return SABC_LAUNCH_RC();
}

int launch_p2p_leave(const P2PView &pv, uint32_t gen, hipStream_t stream) {
  hipLaunchKernelGGL(k_p2p_leave, dim3(1), dim3(64), 0, stream, pv, gen);
  return SABC_LAUNCH_RC();
}

static PatternBufs pattern_bufs(double *const buf[3], const int64_t len[3]) {
  PatternBufs g;
  for (int b = 0; b < 3; ++b) {
    g.buf[b] = reinterpret_cast<uint64_t *>(buf[b]);
    g.len[b] = len[b];
    g.count[b] = (int32_t)(len[b] < kPatternSamples ? len[b] : kPatternSamples);
  }
  return g;
}
int p2p_pattern_save_words() { return 3 * kPatternSamples; }
int launch_p2p_pattern_write(double *const buf[3], const int64_t len[3], double *save, uint32_t gen, int round, int rank, int mode,
                             hipStream_t stream) {
  hipLaunchKernelGGL(k_p2p_pattern_write, dim3(kPatternSamples / 256, 3), dim3(256), 0, stream, pattern_bufs(buf, len),
                     reinterpret_cast<uint64_t *>(save), gen, round, rank, mode);
  return SABC_LAUNCH_RC();
}
int launch_p2p_pattern_check(const double *const peer_buf[3][kMaxPeers], const int64_t len[3], uint32_t gen, int round, int world,
                             unsigned int *out, hipStream_t stream) {
  PatternPeers pp;
  for (int b = 0; b < 3; ++b)
    for (int r = 0; r < kMaxPeers; ++r) pp.buf[b][r] = reinterpret_cast<const uint64_t *>(peer_buf[b][r]);
  double *none[3] = {nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(k_p2p_pattern_check, dim3(kPatternSamples / 256, 3, world), dim3(256), 0, stream, pp, pattern_bufs(none, len), gen,
                     round, world, out);
  return SABC_LAUNCH_RC();
}

int launch_control(ControlBlock *cb, const ControlArgs &a, double *hist, Mailbox *mbox, const double *sums_in,
                   hipStream_t stream) {
  hipLaunchKernelGGL(k_control, dim3(1), dim3(64), 0, stream, cb, a, hist, mbox, sums_in);
  return SABC_LAUNCH_RC();
}

int launch_resample_weights(const ModelDesc &m, PopPtrs pp, const ControlBlock *cb, double n_global, double delta,
                            hipStream_t stream) {
  if (pp.n_local <= 0) return 0;
  hipLaunchKernelGGL(k_resample_weights, dim3((unsigned)n_blocks(pp.n_local)), dim3(kBlock), 0, stream, m.d, m.s, pp,
                     cb, n_global, delta);
  return SABC_LAUNCH_RC();
}

int launch_weight_scan(const ShardBlocks &gathered, int64_t n_global, double *block_sums,
                       double *cum, double *totals, double *totals_host, hipStream_t stream) {
  const int64_t nb = (n_global + kScanChunk - 1) / kScanChunk;
  double *bs = block_sums, *bq = block_sums + nb, *cm = block_sums + 2 * nb;
  WeightArgs wa;
  std::memset(&wa, 0, sizeof(wa));
  // blocks that live in other shards' memory are read ONCE: the first pass parks the weights in `cum`
  const int park = gathered.direct ? 1 : 0;
  hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(kBlock), 0, stream, gathered, n_global, bs, bq, wa, park ? cum : (double *)nullptr);
  hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(1024), 0, stream, bs, bq, nb, totals, totals_host);
  PackArgs none;
  std::memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(kBlock), 0, stream, gathered, n_global, bs, cum, cm, none, park);
  return SABC_LAUNCH_RC();
}

// doubles of the packed-line scratch of launch_resample_local for a shard of n particles:
// packed lines | line ends (+ two of +inf, rounded up to a line) | the guide (n_lines + 2 int32)
static inline int64_t pack_ge_doubles(int64_t lines) { return ((lines + 8 + 15) / 16) * 16; }

int64_t resample_pack_doubles(int row_len, int64_t n) {
  const int per = 16 / (1 + row_len);
  const int pg = per >= 4 ? 4 : per >= 2 ? 2 : per >= 1 ? 1 : 0;
  if (pg == 0) return 0;
  const int64_t lines = (n + pg - 1) / pg;
  return lines * 16 + pack_ge_doubles(lines) + (lines + 2 + 1) / 2 + 32;
}

// One shard: weights (fused into the first scan pass), scan, staging copy, draw + gather (+ the moment sums of the
// resampled population when the model's (d, s) is one the kernels are instantiated for): 4 launches.
// *stats_rows = partial rows written, or -1 when the caller still has to run the stats pass.
int launch_resample_local(const ModelDesc &m, PopPtrs src, PopPtrs dst, const ControlBlock *cb, double delta, uint64_t iter,
                          double *block_sums, double *cum, double *totals, double *totals_host, double *pack,
                          double *partials, int64_t *stats_rows, hipStream_t stream) {
  const int64_t n = src.n_local, cap = src.cap;
  *stats_rows = -1;
  if (n <= 0) return 0;
  const int rows = m.d + m.s + 1, rl = m.d + m.s;
  const int per = 16 / (1 + rl);
  const int pg = per >= 4 ? 4 : per >= 2 ? 2 : per >= 1 ? 1 : 0;
  const int64_t nb = (n + kScanChunk - 1) / kScanChunk;
  double *bs = block_sums, *bq = block_sums + nb, *cm = block_sums + 2 * nb;
  WeightArgs wa;
  std::memset(&wa, 0, sizeof(wa));
  wa.fused = 1; wa.d = m.d; wa.s = m.s; wa.pp = src; wa.cb = cb; wa.n_global = (double)n; wa.delta = delta;
  PackArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  if (pg > 0 && pack) {
    const int64_t lines = (n + pg - 1) / pg;
    pa.pk = pack; pa.ge = pack + lines * 16; pa.guide = reinterpret_cast<int32_t *>(pa.ge + pack_ge_doubles(lines));
    pa.totals = totals; pa.row_len = rl; pa.pg = pg;
  }
  const ShardBlocks own = flat_blocks(src.pop, rows, cap, 1);
  hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(kBlock), 0, stream, own, n, bs, bq, wa, (double *)nullptr);
  hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(1024), 0, stream, bs, bq, nb, totals, totals_host);
  hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(kBlock), 0, stream, own, n, bs, cum, cm, pa, 0);
  const size_t lds = nb <= kGatherCoarseMax ? (size_t)nb * sizeof(double) : 0;
  const dim3 grid((unsigned)n_blocks(n)), block(kBlock);
  if (!pa.pk) {                      // a row does not fit a 128-byte line (d + s > 15): the unpacked gather, sums by the caller
    hipLaunchKernelGGL(k_resample_gather<true>, grid, block, lds, stream, m.seed, m.d, m.s, own, n,
                       (const double *)cum, (const double *)bs, (const double *)cm, nb, (const double *)totals, iter, dst,
                       (int64_t *)nullptr);
    return SABC_LAUNCH_RC();
  }
#define CALL(M, D, S)                                                                                                    \
  do {                                                                                                                   \
    hipLaunchKernelGGL((k_resample_gather_stats<D, S>), grid, block, lds, stream, m.seed, (const double *)pa.pk,         \
                       (const double *)pa.ge, (const int32_t *)pa.guide, pg, n, (const double *)bs, nb,                  \
                       (const double *)totals, iter, dst, cb, partials);                                                 \
    *stats_rows = n_blocks(n);                                                                                           \
  } while (0)
  if (m.model_id == SABC_MODEL_GK) { CALL(SABC_MODEL_GK, 4, 4); return SABC_LAUNCH_RC(); }
  if (m.model_id == SABC_MODEL_HOST || m.model_id == SABC_MODEL_USER) {
    hipLaunchKernelGGL(k_resample_gather_packed, grid, block, lds, stream, m.seed, rl, (const double *)pa.pk,
                       (const double *)pa.ge, (const int32_t *)pa.guide, pg, n, (const double *)bs, nb, (const double *)totals, iter,
                       dst);
    return SABC_LAUNCH_RC();
  }
  SABC_DISPATCH_MODEL(m, CALL);
#undef CALL
  return SABC_LAUNCH_RC();
}

int launch_resample_gather(const ModelDesc &m, const ShardBlocks &gathered, int64_t n_global,
                           const double *cum, const double *block_sums, const double *totals, uint64_t iter, PopPtrs dst,
                           hipStream_t stream) {
  if (dst.n_local <= 0) return 0;
  const int64_t nb = (n_global + kScanChunk - 1) / kScanChunk;
  const size_t lds = nb <= kGatherCoarseMax ? (size_t)nb * sizeof(double) : 0;
  hipLaunchKernelGGL(k_resample_gather<true>, dim3((unsigned)n_blocks(dst.n_local)), dim3(kBlock), lds, stream, m.seed, m.d,
                     m.s, gathered, n_global, cum, block_sums, block_sums + 2 * nb, nb, totals, iter, dst,
                     (int64_t *)nullptr);
  return SABC_LAUNCH_RC();
}

int launch_resample_select(const ModelDesc &m, int64_t cap, int64_t n_global, const double *cum, const double *block_sums,
                           const double *totals, uint64_t iter, PopPtrs dst, int64_t *idx_out, hipStream_t stream) {
  if (dst.n_local <= 0) return 0;
  const int64_t nb = (n_global + kScanChunk - 1) / kScanChunk;
  const size_t lds = nb <= kGatherCoarseMax ? (size_t)nb * sizeof(double) : 0;
  hipLaunchKernelGGL(k_resample_gather<false>, dim3((unsigned)n_blocks(dst.n_local)), dim3(kBlock), lds, stream, m.seed, m.d,
                     m.s, flat_blocks(nullptr, 0, cap, 1), n_global, cum, block_sums, block_sums + 2 * nb, nb, totals, iter,
                     dst, idx_out);
  return SABC_LAUNCH_RC();
}

int launch_bucket_count(const int64_t *idx, int64_t n_local, int64_t cap, unsigned long long *counts, hipStream_t stream) {
  if (n_local <= 0) return 0;
  hipLaunchKernelGGL(k_bucket_count, dim3((unsigned)n_blocks(n_local)), dim3(kBlock), 0, stream, idx, n_local, cap, counts);
  return SABC_LAUNCH_RC();
}

int launch_bucket_scatter(const int64_t *idx, int64_t n_local, int64_t cap, unsigned long long *cursor, double *req,
                          int64_t *slot, hipStream_t stream) {
  if (n_local <= 0) return 0;
  hipLaunchKernelGGL(k_bucket_scatter, dim3((unsigned)n_blocks(n_local)), dim3(kBlock), 0, stream, idx, n_local, cap, cursor,
                     req, slot);
  return SABC_LAUNCH_RC();
}

int launch_resample_serve(const double *req, int64_t m, int row_len, PopPtrs src, double *rows_out, hipStream_t stream) {
  if (m <= 0) return 0;
  hipLaunchKernelGGL(k_resample_serve, dim3((unsigned)n_blocks(m * row_len)), dim3(kBlock), 0, stream, req, m, row_len, src,
                     rows_out);
  return SABC_LAUNCH_RC();
}

int launch_resample_scatter(const double *rows_in, const int64_t *slot, int64_t n_local, int row_len, PopPtrs dst,
                            hipStream_t stream) {
  if (n_local <= 0) return 0;
  hipLaunchKernelGGL(k_resample_scatter, dim3((unsigned)n_blocks(n_local * row_len)), dim3(kBlock), 0, stream, rows_in, slot,
                     n_local, row_len, dst);
  return SABC_LAUNCH_RC();
}

int launch_cdf_knots(const double *sorted, int64_t n, double *knots, int64_t *meta, hipStream_t stream) {
  hipLaunchKernelGGL(k_cdf_meta, dim3(1), dim3(64), 0, stream, sorted, n, meta);
  hipLaunchKernelGGL(k_cdf_fill, dim3((unsigned)n_blocks(n > 0 ? n : 1)), dim3(kBlock), 0, stream, sorted, n, meta, knots);
  return SABC_LAUNCH_RC();
}

int launch_cdf_index(double *knots, int64_t len, int64_t stride, int shift, double *coarse, int n_coarse, double *mid,
                     int64_t mid_len, hipStream_t stream) {
  int64_t work = n_coarse;
  if (mid_len > work) work = mid_len;
  if (stride - len > work) work = stride - len;
  hipLaunchKernelGGL(k_cdf_index, dim3((unsigned)n_blocks(work)), dim3(kBlock), 0, stream, knots, len, stride, shift, coarse,
                     n_coarse, mid, mid_len);
  return SABC_LAUNCH_RC();
}

int launch_compact_column(const ShardBlocks &gathered, int stat, int64_t n_global, double *out, hipStream_t stream) {
  hipLaunchKernelGGL(k_compact_column, dim3((unsigned)n_blocks(n_global)), dim3(kBlock), 0, stream, gathered, stat, n_global, out);
  return SABC_LAUNCH_RC();
}

int launch_cdf_eval(const double *knots, int64_t len, const double *q, int64_t m, double *out, hipStream_t stream) {
  if (m <= 0) return 0;
  hipLaunchKernelGGL(k_cdf_eval, dim3((unsigned)n_blocks(m)), dim3(kBlock), 0, stream, knots, len, q, m, out);
  return SABC_LAUNCH_RC();
}

int launch_cdf_apply_matrix(CdfPtrs cdf, int s, const double *rho, int64_t m, double *u_out, hipStream_t stream) {
  if (m <= 0) return 0;
  hipLaunchKernelGGL(k_cdf_apply_matrix, dim3((unsigned)n_blocks(m)), dim3(kBlock), 0, stream, cdf, s, rho, m, u_out);
  return SABC_LAUNCH_RC();
}

int launch_simulate_batch(const ModelDesc &m, const double *theta, int64_t n, uint64_t pid0, uint64_t iter,
                          double *rho_out, hipStream_t stream, const RtcKernels *rtc, const unsigned char *gate) {
  if (n <= 0) return 0;
  if (m.model_id == SABC_MODEL_USER) {
    if (!rtc || !rtc->simulate_batch) return (int)hipErrorInvalidValue;
    return module_launch(rtc->simulate_batch, (unsigned)n_blocks(n), kBlock, stream, nullptr, nullptr, m, theta, n, pid0, iter,
                         rho_out, gate);
  }
  if (m.model_id == SABC_MODEL_GK) {                  // (the gate is not looked at: the wave-cooperative simulator is a fixed
                                                      // amount of arithmetic for any theta, and gated-out rows are never read)
    hipLaunchKernelGGL(k_simulate_gk, dim3(gk_blocks(n)), dim3(kBlock), 0, stream, m, theta, n, n, pid0, iter, 0,
                       (double *)nullptr, rho_out, n);
    return SABC_LAUNCH_RC();
  }
  const dim3 grid((unsigned)n_blocks(n)), block(kBlock);
#define CALL(M, D, S) hipLaunchKernelGGL((k_simulate_batch<M, D, S>), grid, block, 0, stream, m, theta, n, pid0, iter, rho_out, gate)
  SABC_DISPATCH_MODEL(m, CALL);
#undef CALL
  return SABC_LAUNCH_RC();
}

int launch_prior_op(const ModelDesc &m, uint64_t pid0, int64_t n, double *theta, double *lp, hipStream_t stream,
                    const RtcKernels *rtc) {
  if (n <= 0) return 0;
  if (rtc) {                                   // a prior that lives in the run-time compiled unit
    if (!rtc->prior_op) return (int)hipErrorInvalidValue;
    return module_launch(rtc->prior_op, (unsigned)n_blocks(n), kBlock, stream, nullptr, nullptr, m, pid0, n, theta, lp);
  }
  hipLaunchKernelGGL(k_prior_op, dim3((unsigned)n_blocks(n)), dim3(kBlock), 0, stream, m, pid0, n, theta, lp);
  return SABC_LAUNCH_RC();
}

int launch_normal_pairs(uint64_t seed, uint64_t pid0, uint32_t purpose, uint64_t iter, uint32_t k, int64_t m, double *out,
                        hipStream_t stream) {
  if (m <= 0) return 0;
  hipLaunchKernelGGL(k_normal_pairs, dim3((unsigned)n_blocks(m)), dim3(kBlock), 0, stream, seed, pid0, purpose, iter, k, m, out);
  return SABC_LAUNCH_RC();
}

int launch_rng_peak(uint64_t seed, int pairs, int64_t n, double *out, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_rng_peak, dim3((unsigned)n_blocks(n)), dim3(kBlock), 0, stream, seed, pairs, n, out);
  return SABC_LAUNCH_RC();
}

int launch_philox_debug(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k, uint32_t *words,
                        double *normals, hipStream_t stream) {
  hipLaunchKernelGGL(k_philox_debug, dim3(1), dim3(64), 0, stream, seed, pid, purpose, iter, k, words, normals);
  return SABC_LAUNCH_RC();
}

}  // namespace sabc
