// control.hpp -- what happens to the algorithm state between two population updates
// (SimulatedAnnealingABC.jl:334,348-354,367-372; proposals.jl:46-60), as ONE function over the
// ControlBlock.  It is compiled as device code and run by a single lane (k_control in kernels.hip)
// so that the host does not have to read the sums back after every update; the CPU engine tests
// call the very same function on the host.
#pragma once
#include "host_math.hpp"
#include "sabc_types.hpp"

#ifndef SABC_CTRL_MARK
#define SABC_CTRL_MARK(i) do { } while (0)      // (timing instrumentation of an A/B build, kernels.hip)
#endif

namespace sabc {

// the multi-eps schedule's epsilons, computed ahead of the step by one lane per statistic (kernels.hip: control_on_copy) from
// the sums the step is about to take over; the step applies them in order, exactly as it would have computed them
struct EpsCandidates {
  double eps[kMaxStats];
  int32_t ok[kMaxStats];                  // 0: mean u of this statistic <= eps() (:107-109)
};

// component q of the sums, taken over from the staging buffer (the control kernels do this with one lane per component and
// then run the step with CTRL_KEEP_SUMS: 1 + 2s + d + d(d+1)/2 dependent LDS round trips on one lane are 10 us at s = 48)
SABC_HD inline void control_take_sum(ControlBlock &cb, const ControlArgs &a, const double *sums_in, int q) {
  const bool delta = a.rho_is_delta && q >= 1 + a.s && q < 1 + 2 * a.s;            // running sum(rho) += its change
  cb.sums[q] = delta ? cb.sums[q] + sums_in[q] : sums_in[q];
}

// The step in two parts, so that k_update_persistent can start the next update's proposals and simulations as soon as the FIRST
// is done (what they need of the state is the Cholesky factor) while one lane still works on the SECOND (persistent_kernel.hpp:
// the control wave).  control_step() below is the two in sequence: the launch chain, the host engine and the tests call that.
//
// DD, SS: the shape as compile-time constants where the caller has them (k_update_persistent: the offsets into the sums are then
// constants, the loops straight-line code and the d == 1 branch the only one compiled -- the step is a chain of dependent LDS
// round trips on one lane, and every address the compiler can resolve is one it can batch); 0: from the arguments
enum ControlFirst : int { CONTROL_NOOP = 0, CONTROL_HALTED = 1, CONTROL_GOES_ON = 2 };

// the sums taken over, the accept count (:334), the resample test (:340), the proposal's covariance (update_proposal!, :348)
template <int DD = 0, int SS = 0>
SABC_HD inline ControlFirst control_step_first(ControlBlock &cb, const ControlArgs &a, const double *sums_in) {
  if ((a.mode & CTRL_GUARDED) && cb.halt) return CONTROL_NOOP;
  if (a.mode & CTRL_CLEAR_HALT) cb.halt = 0;
  const int d = DD > 0 ? DD : a.d, s = SS > 0 ? SS : a.s;
  if (!(a.mode & CTRL_KEEP_SUMS))
    for (int q = 0; q < n_partials(d, s); ++q) control_take_sum(cb, a, sums_in, q);
  SABC_CTRL_MARK(9);
  const double n = a.n_global;
  const double *S = &cb.sums[1 + 2 * s], *Q = &cb.sums[1 + 2 * s + d];

  if (a.mode & CTRL_ACCUMULATE) cb.n_accept += (int64_t)(cb.sums[0] + 0.5);          // :334

  // the resample test of :340, on the device: when it fires, eps / Sigma must come from the RESAMPLED
  // population, so stop here; everything queued behind this step sees `halt` and does nothing until
  // the host has run the resample and cleared the flag
  if ((a.mode & CTRL_CHECK) && (double)cb.n_accept >= a.resample_threshold) {
    cb.halt = 1;
    return CONTROL_HALTED;
  }

  if ((a.mode & CTRL_PROPOSAL) && a.prop_kind == SABC_PROP_RANDOMWALK && d >= 2 && d <= 4) {
    const bool ok = d == 2 ? hostmath::rw_proposal_from_sums<2>(S, Q, n, a.prop_p0, cb.sigma, cb.chol)
                  : d == 3 ? hostmath::rw_proposal_from_sums<3>(S, Q, n, a.prop_p0, cb.sigma, cb.chol)
                           : hostmath::rw_proposal_from_sums<4>(S, Q, n, a.prop_p0, cb.sigma, cb.chol);
    if (!ok) cb.error = SABC_ERR_NOT_POSDEF;                                         // MvNormal(...), :42
  } else if ((a.mode & CTRL_PROPOSAL) && a.prop_kind == SABC_PROP_RANDOMWALK) {     // update_proposal!
    // (no local d x d array: on the single control lane it would live in scratch memory, every element a trip to the L2)
    if (d == 1) {
      const double c = (Q[0] - S[0] * S[0] / n) / (n - 1.0);                         // hostmath::cov_from_sums, d = 1
      // a variance is >= 0; the one-pass formula can round a population of identical particles (possible
      // after a resample of a tiny population) to -1e-17, where the reference's two-pass cov gives 0
      cb.sigma[0] = a.prop_p0 * (c > 0.0 ? c : 0.0);                                 // proposals.jl:59
      cb.chol[0] = sqrt(cb.sigma[0]);                                                // proposals.jl:54
    } else {
      hostmath::cov_from_sums(S, Q, d, n, cb.sigma);
      for (int k = 0; k < d; ++k)
        for (int l = 0; l < d; ++l)
          cb.sigma[k * d + l] = a.prop_p0 * (cb.sigma[k * d + l] + (k == l ? 1e-8 : 0.0));   // proposals.jl:47
      if (!hostmath::cholesky(cb.sigma, d, cb.chol)) cb.error = SABC_ERR_NOT_POSDEF;   // MvNormal(...), :42
    }
  }
  SABC_CTRL_MARK(10);
  return CONTROL_GOES_ON;
}

// epsilon (:350-354), the history row (:367-372), the pivot of the moment sums.  pre (valid iff pre_valid): the multi-eps
// schedule's candidates, computed ahead by one lane per statistic
template <int DD = 0, int SS = 0>
SABC_HD inline void control_step_second(ControlBlock &cb, const ControlArgs &a, double *hist, const EpsCandidates *pre, const bool pre_valid) {
  const int d = DD > 0 ? DD : a.d, s = SS > 0 ? SS : a.s;
  const double n = a.n_global;
  const double *S = &cb.sums[1 + 2 * s];
  if (a.mode & CTRL_EPSILON) {                                                       // :350-354
    if (a.algorithm == SABC_ALG_MULTI_EPS && pre_valid) {
      for (int j = 0; j < s; ++j) {                  // (a failing statistic stops the schedule where eps_multi would have)
        if (!pre->ok[j]) { cb.error = SABC_ERR_ZERO_MEAN_U; break; }                 // :107-109
        cb.eps[j] = pre->eps[j];
      }
    } else if (a.algorithm == SABC_ALG_MULTI_EPS) {
#if defined(__HIP_DEVICE_COMPILE__)
      __shared__ double ubar[kMaxStats];             // (a local array indexed at run time would live in scratch memory)
#else
      double ubar[kMaxStats];
#endif
      for (int j = 0; j < s; ++j) ubar[j] = cb.sums[1 + j] / n;
      if (!hostmath::eps_multi(ubar, s, a.v, cb.eps)) cb.error = SABC_ERR_ZERO_MEAN_U;   // :107-109
    } else {
      double tot = 0.0;
      for (int j = 0; j < s; ++j) tot += cb.sums[1 + j];
      cb.eps[0] = hostmath::eps_single(tot / (n * (double)s), a.v, cb.eps[0]);       // mean(u), :353
    }
  }

  SABC_CTRL_MARK(11);
  if (a.mode & CTRL_HISTORY) {                                                       // :367-372
    if (cb.hist_rows < a.hist_capacity) {
      if (hist) {                                    // (nullptr: a workgroup of k_update_persistent that only keeps the count)
        double *row = hist + cb.hist_rows * (cb.eps_len + 2 * s);
        for (int i = 0; i < cb.eps_len; ++i) row[i] = cb.eps[i];
        for (int j = 0; j < s; ++j) row[cb.eps_len + j] = cb.sums[1 + j] / n;
        for (int j = 0; j < s; ++j) row[cb.eps_len + s + j] = cb.sums[1 + s + j] / n;
      }
      cb.hist_rows += 1;
    } else {
      cb.error = SABC_ERR_STATE;
    }
  }

  SABC_CTRL_MARK(12);
  // keep the moment sums centred: the sums in hand are relative to the old pivot, so this goes last
  if (a.mode & CTRL_PIVOT)
    for (int k = 0; k < d; ++k) cb.pivot[k] += S[k] / n;
}

// returns false when the step was a no-op (guarded and halted): nothing must be posted then
// `sums_in` is the staging buffer the reduction (and the allreduce) wrote; it is taken over into the
// control block only by a step that really runs, so the collectives of an aborted step cannot touch state.
template <int DD = 0, int SS = 0>
SABC_HD inline bool control_step(ControlBlock &cb, const ControlArgs &a, double *hist, const double *sums_in,
                                 const EpsCandidates *pre = nullptr, const bool pre_valid = true) {
  const ControlFirst r = control_step_first<DD, SS>(cb, a, sums_in);
  if (r == CONTROL_NOOP) return false;
  if (r == CONTROL_HALTED) return true;
  control_step_second<DD, SS>(cb, a, hist, pre, pre != nullptr && pre_valid);
  return true;
}

}  // namespace sabc
