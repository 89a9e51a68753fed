// host_math.hpp -- the O(s) / O(d^3) scalar pieces of a population update: both epsilon schedules
// (SimulatedAnnealingABC.jl:92-117), the covariance from fused moment sums (proposals.jl:47,59)
// and its Cholesky factor (implicit in MvNormal(...), proposals.jl:42).  Compiled twice: as device
// code for the single-lane control kernel (k_control in kernels.hip, so that a population update
// needs no host round trip) and as plain C++ for the operator entry points and the CPU engine tests.
#pragma once
#if !defined(__HIPCC_RTC__)                // (hipRTC: no standard headers, the HIP runtime is pre-included)
#include <cfloat>
#include <cmath>
#include <cstdint>
#endif
#ifndef DBL_EPSILON
#define DBL_EPSILON 2.2204460492503131e-16
#endif

#if defined(__HIPCC_RTC__)
#define SABC_HD __host__ __device__
#elif defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SABC_HD __host__ __device__
#else
#define SABC_HD
#endif
#include "prior_math.hpp"

namespace sabc {
namespace hostmath {

// update_epsilon_single_eps (:92-95): root of e^2 + v e^1.5 - ubar^2 on (0, ubar).
// With e = x^2 this is g(x) = x^4 + v x^3 - ubar^2, convex and increasing for x > 0 with
// g(sqrt(ubar)) > 0, so Newton from sqrt(ubar) decreases monotonically onto the root (no sqrt in
// the loop; a handful of iterations instead of ~70 bisection steps on the single control lane).
SABC_HD inline double eps_single(double ubar, double v, double eps_hint = -1.0) {
  if (ubar <= DBL_EPSILON) return 0.0;                      // :93 ubar <= eps()
  const double u2 = ubar * ubar;
  double x = sqrt(ubar);
  // warm start: the previous eps is usually just right of the new root (eps falls while annealing);
  // it is a valid monotone start whenever g(sqrt(hint)) >= 0 and it lies inside the bracket
  if (eps_hint > 0.0 && eps_hint < ubar) {
    const double xh = sqrt(eps_hint), xh2 = xh * xh;
    if (xh2 * xh2 + v * xh2 * xh - u2 >= 0.0) x = xh;
  }
  for (int it = 0; it < 200; ++it) {
    const double x2 = x * x;
    const double g = x2 * x2 + v * x2 * x - u2;
    const double gp = 4.0 * x2 * x + 3.0 * v * x2;
    const double xn = x - g / gp;
    if (!(xn < x)) break;                                   // stopped decreasing: converged to rounding
    x = xn;
  }
  return x * x;
}

// (1 - e^-b (1 + b)) / (b (1 - e^-b)) of :113, i.e. the mean of the density ~exp(-b u) on [0,1],
// written as 1/b - 1/(e^b - 1) to avoid the cancellation of the literal form.
SABC_HD inline double tilted_mean(double b) {
  if (fabs(b) < 1e-2) {
    const double b2 = b * b;
    return 0.5 - b / 12.0 + b * b2 / 720.0 - b * b2 * b2 / 30240.0 + b * b2 * b2 * b2 / 1209600.0;
  }
  return 1.0 / b - 1.0 / expm1(b);
}

// d/db of tilted_mean
SABC_HD inline double tilted_mean_deriv(double b) {
  if (fabs(b) < 1e-2) { const double b2 = b * b; return -1.0 / 12.0 + b2 / 240.0 - b2 * b2 / 6048.0; }
  const double em = expm1(b);
  return -1.0 / (b * b) + (em + 1.0) / (em * em);
}

// beta_i of :113: tilted_mean(beta) = ubar_i.  Decreasing in beta, 1/2 at 0, < 1/beta for beta > 0: the root lies in
// (0, 1/ubar_i].
//
// A FUNCTION OF ubar_i ALONE.  Round 3 started the iteration from the previous update's root and stopped it at a relative
// step of 4 eps: a warm and a cold start could end on neighbouring doubles, so epsilon depended on the call history (a
// resumed run, a fresh handle and the oracle start cold) -- an ulp that can flip an accept at the boundary.  Now:
//   * mean u <= 1/44 (the late stage of every chain): the root is 1 / mean u -- e^-beta is below the last bit of 1/beta, the
//     equation reads 1/beta = ubar in double precision;
//   * otherwise a fixed start -- (1 - 2u)/u * p4(u), exact at both ends of (0, 1/2) and within 1e-3 in between -- and a FIXED
//     number of Newton steps (three: 1e-3 -> 1e-6 -> 1e-12 -> the noise of evaluating the equation itself), each one expm1.
// No data-dependent stopping, no hint: the same bits wherever and whenever it is evaluated, and as few expm1 as the warm start
// took at its best (a step is ~1 us on a control lane).  The derivative is written so that it stays finite when expm1
// overflows (the literal (e + 1) / e^2 is inf / inf there).
SABC_HD inline double multi_eps_beta(double ub) {
  if (ub == 0.5) return 0.0;
  const bool mirror = ub > 0.5;                      // tilted_mean(-b) = 1 - tilted_mean(b)
  if (mirror) ub = 1.0 - ub;
  double b;
  if (ub <= 1.0 / 44.0) {
    b = 1.0 / ub;
  } else {
    const double g = (((18.08509524 * ub - 19.10561105) * ub + 9.73841092) * ub + 1.63676858) * ub + 1.00652501;
    b = (1.0 - 2.0 * ub) / ub * g;
    for (int it = 0; it < 3; ++it) {
      // tilted_mean(b) and tilted_mean_deriv(b) from ONE expm1 (the same expressions as the two functions above)
      double tm, td;
      if (fabs(b) < 1e-2) {
        tm = tilted_mean(b);
        td = tilted_mean_deriv(b);
      } else {
        const double r = 1.0 / expm1(b);
        tm = 1.0 / b - r;
        td = -1.0 / (b * b) + r * (1.0 + r);         // = (e + 1) / e^2 with e = expm1(b)
      }
      b -= (tm - ub) / td;
    }
  }
  return mirror ? -b : b;
}

// x^(s/2) for x > 0 and a small integer s: sqrt and s - 1 multiplications (pow() is several hundred dependent instructions
// on the control lane and :110-112 need s^2 + s of them per update; the result differs from pow's in the last digits)
SABC_HD inline double pow_half_int(double x, int s) {
  const double r = sqrt(x);
  double p = 1.0;
  for (int i = 0; i < s; ++i) p *= r;
  return p;
}

// update_epsilon_multi_eps (:100-117), one statistic at a time: the s epsilons do not depend on each other, so the control
// kernels give each its own lane (kernels.hip: control_on_copy) -- on ONE lane the schedule is s^2 divisions and square
// roots plus s root solves per population update: 12 us at s = 3, over a millisecond at s = 48.
SABC_HD inline double eps_multi_cn(int s) {          // (2s+2)! / ((s+1)! (s+2)!)  (:103)
  double cn = 1.0;
  for (int k = 1; k <= s + 1; ++k) cn = cn * (double)(s + 1 + k) / (double)k;
  return cn / (double)(s + 2);
}
// false when ubar_i <= eps() (:107-109)
SABC_HD inline bool eps_multi_one(const double *ubar, int s, double v, double cn, int i, double *eps_i) {
  const double ui = ubar[i];
  if (ui <= DBL_EPSILON) return false;
  double num = 1.0, prodq = 1.0;
  for (int j = 0; j < s; ++j) {
    const double q = ubar[j] / ui;                   // :110
    num += pow_half_int(q, s);                       // :111  q^(s/2)
    prodq *= q;
  }
  const double den = cn * (s + 1) * (ui * pow_half_int(ui, s)) * prodq;   // :112  ui^(1 + s/2)
  const double beta = multi_eps_beta(ui);
  *eps_i = 1.0 / (beta + v * num / den);                                   // :113-114
  return true;
}

// update_epsilon_multi_eps (:100-117); returns false when some ubar_i <= eps() (:107-109)
SABC_HD inline bool eps_multi(const double *ubar, int s, double v, double *eps_out) {
  const double cn = eps_multi_cn(s);
  for (int i = 0; i < s; ++i)
    if (!eps_multi_one(ubar, s, v, cn, i, &eps_out[i])) return false;
  return true;
}

// row-major lower Cholesky; false if not positive definite
SABC_HD inline bool cholesky(const double *a, int d, double *l) {
  for (int i = 0; i < d * d; ++i) l[i] = 0.0;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      double sum = a[i * d + j];
      for (int k = 0; k < j; ++k) sum -= l[i * d + k] * l[j * d + k];
      if (i == j) {
        if (!(sum > 0.0)) return false;
        l[i * d + i] = sqrt(sum);
      } else {
        l[i * d + j] = sum / l[j * d + j];
      }
    }
  return true;
}

// sample covariance (n-1 denominator, StatsBase.cov) from pivot-shifted sums:
//   S_k = sum (x_k - c_k),  Q_kl = sum (x_k - c_k)(x_l - c_l) (row-major lower, l <= k)
SABC_HD inline void cov_from_sums(const double *S, const double *Q, int d, double n, double *cov) {
  int q = 0;
  for (int k = 0; k < d; ++k)
    for (int l = 0; l <= k; ++l, ++q) {
      const double c = (Q[q] - S[k] * S[l] / n) / (n - 1.0);
      cov[k * d + l] = cov[l * d + k] = c;
    }
}

// update_proposal!(::RandomWalk) for d > 1 (proposals.jl:46-48 + the Cholesky factor inside MvNormal(...), :42) with the
// dimension at compile time: the sums are read once, everything in between lives in registers, Sigma and its factor are
// written once.  The single control lane otherwise walks d x d arrays it cannot keep in registers (run-time d), every
// element a dependent trip to scratch or LDS -- 18 us per update at d = 4.  Same expressions in the same order as
// cov_from_sums / cholesky: the same numbers.
template <int D>
SABC_HD inline bool rw_proposal_from_sums(const double *S_in, const double *Q_in, double n, double beta, double *sigma_out,
                                          double *chol_out) {
  double S[D], Q[D * (D + 1) / 2], sg[D * D], l[D * D];
#pragma unroll
  for (int k = 0; k < D; ++k) S[k] = S_in[k];
#pragma unroll
  for (int q = 0; q < D * (D + 1) / 2; ++q) Q[q] = Q_in[q];
  {
    int q = 0;
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
      for (int m = 0; m <= k; ++m, ++q) {
        const double c = (Q[q] - S[k] * S[m] / n) / (n - 1.0);
        sg[k * D + m] = beta * (c + (k == m ? 1e-8 : 0.0));                  // proposals.jl:47
        sg[m * D + k] = sg[k * D + m];
      }
  }
  bool ok = true;
#pragma unroll
  for (int i = 0; i < D * D; ++i) l[i] = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double sum = sg[i * D + j];
#pragma unroll
      for (int k = 0; k < j; ++k) sum -= l[i * D + k] * l[j * D + k];
      if (i == j) {
        if (!(sum > 0.0)) ok = false;
        l[i * D + i] = sqrt(sum);
      } else {
        l[i * D + j] = sum / l[j * D + j];
      }
    }
#pragma unroll
  for (int i = 0; i < D * D; ++i) sigma_out[i] = sg[i];
  if (ok) {
#pragma unroll
    for (int i = 0; i < D * D; ++i) chol_out[i] = l[i];
  }
  return ok;
}

}  // namespace hostmath
}  // namespace sabc
