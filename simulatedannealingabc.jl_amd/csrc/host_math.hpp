// host_math.hpp -- the O(s) / O(d^3) scalar pieces of a population update that run on the host
// between kernels: both epsilon schedules (SimulatedAnnealingABC.jl:92-117), the covariance
// from fused moment sums (proposals.jl:47,59) and its Cholesky factor (implicit in
// MvNormal(...), proposals.jl:42).  Pure C++: shared by libsabc_hip.so and the CPU engine tests.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace sabc {
namespace hostmath {

// update_epsilon_single_eps (:92-95): root of e^2 + v e^1.5 - ubar^2 on (0, ubar).
// Roots.find_zero with a bracket is bisection down to adjacent floats; same here.
inline double eps_single(double ubar, double v) {
  if (ubar <= DBL_EPSILON) return 0.0;
  const double u2 = ubar * ubar;
  double lo = 0.0, hi = ubar;   // f(lo) < 0 < f(hi)
  for (int it = 0; it < 2200; ++it) {
    const double mid = lo + 0.5 * (hi - lo);
    if (!(mid > lo && mid < hi)) break;
    const double f = mid * mid + v * mid * std::sqrt(mid) - u2;
    if (f > 0.0) hi = mid; else lo = mid;
  }
  // the bracket is now two neighbouring doubles; return the end with the smaller residual
  const double fl = std::fabs(lo * lo + v * lo * std::sqrt(lo) - u2);
  const double fh = std::fabs(hi * hi + v * hi * std::sqrt(hi) - u2);
  return fl <= fh ? lo : hi;
}

// (1 - e^-b (1 + b)) / (b (1 - e^-b)) of :113, i.e. the mean of the density ~exp(-b u) on [0,1],
// written as 1/b - 1/(e^b - 1) to avoid the cancellation of the literal form.
inline double tilted_mean(double b) {
  if (std::fabs(b) < 1e-2) {
    const double b2 = b * b;
    return 0.5 - b / 12.0 + b * b2 / 720.0 - b * b2 * b2 / 30240.0 + b * b2 * b2 * b2 / 1209600.0;
  }
  return 1.0 / b - 1.0 / std::expm1(b);
}

// beta_i of :113: tilted_mean(beta) = ubar_i.  Decreasing in beta, 1/2 at 0, < 1/beta for beta > 0.
inline double multi_eps_beta(double ub) {
  if (ub == 0.5) return 0.0;
  if (ub > 0.5) return -multi_eps_beta(1.0 - ub);   // tilted_mean(-b) = 1 - tilted_mean(b)
  double lo = 0.0, hi = 1.0 / ub;                   // f(lo) > 0 > f(hi)
  for (int it = 0; it < 2200; ++it) {
    const double mid = lo + 0.5 * (hi - lo);
    if (!(mid > lo && mid < hi)) break;
    if (tilted_mean(mid) - ub > 0.0) lo = mid; else hi = mid;
  }
  return std::fabs(tilted_mean(lo) - ub) <= std::fabs(tilted_mean(hi) - ub) ? lo : hi;
}

// update_epsilon_multi_eps (:100-117); returns false when some ubar_i <= eps() (:107-109)
inline bool eps_multi(const double *ubar, int s, double v, double *eps_out) {
  double cn = 1.0;                                   // (2s+2)! / ((s+1)! (s+2)!)  (:103)
  for (int k = 1; k <= s + 1; ++k) cn = cn * (double)(s + 1 + k) / (double)k;
  cn /= (double)(s + 2);
  for (int i = 0; i < s; ++i) {
    const double ui = ubar[i];
    if (ui <= DBL_EPSILON) return false;
    double num = 1.0, prodq = 1.0;
    for (int j = 0; j < s; ++j) {
      const double q = ubar[j] / ui;                 // :110
      num += std::pow(q, s / 2.0);                   // :111
      prodq *= q;
    }
    const double den = cn * (s + 1) * std::pow(ui, 1.0 + s / 2.0) * prodq;   // :112
    eps_out[i] = 1.0 / (multi_eps_beta(ui) + v * num / den);                 // :113-114
  }
  return true;
}

// row-major lower Cholesky; false if not positive definite
inline bool cholesky(const double *a, int d, double *l) {
  for (int i = 0; i < d * d; ++i) l[i] = 0.0;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      double sum = a[i * d + j];
      for (int k = 0; k < j; ++k) sum -= l[i * d + k] * l[j * d + k];
      if (i == j) {
        if (!(sum > 0.0)) return false;
        l[i * d + i] = std::sqrt(sum);
      } else {
        l[i * d + j] = sum / l[j * d + j];
      }
    }
  return true;
}

// sample covariance (n-1 denominator, StatsBase.cov) from pivot-shifted sums:
//   S_k = sum (x_k - c_k),  Q_kl = sum (x_k - c_k)(x_l - c_l) (row-major lower, l <= k)
inline void cov_from_sums(const double *S, const double *Q, int d, double n, double *cov) {
  int q = 0;
  for (int k = 0; k < d; ++k)
    for (int l = 0; l <= k; ++l, ++q) {
      const double c = (Q[q] - S[k] * S[l] / n) / (n - 1.0);
      cov[k * d + l] = cov[l * d + k] = c;
    }
}

}  // namespace hostmath
}  // namespace sabc
