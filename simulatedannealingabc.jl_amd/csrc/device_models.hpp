// device_models.hpp -- priors, device-coded simulators (the user's f_dist of
// SimulatedAnnealingABC.jl:164,175,315 shipped as data) and the empirical-CDF lookup
// (cdf_estimators.jl:39-42,68-70) as gfx950 device functions.
#pragma once
#include "device_rng.hpp"
#include "prior_math.hpp"
#include "kernels.hpp"
#include "sabc_types.hpp"

// g-and-k: order statistics at multiples of 16 from block maxima instead of the last four steps of the sort (A/B switch)
#ifndef SABC_GK_BLOCKMAX
#define SABC_GK_BLOCKMAX 1
#endif

namespace sabc {

#define SABC_LOG2PI 1.8378770664093454835606594728112

// ---- prior: product of univariate Normal / Uniform / Exponential / LogNormal / Gamma / Beta / truncated Normal ----
// one dimension of Distributions.logpdf (:314,318); -inf outside the support.
// ModelDesc::prior_logc holds the family's normalising constant, computed once on the host (engine.cpp).
__device__ __forceinline__ double prior_logpdf_dim(const ModelDesc &m, int k, double x) {
  const int kind = m.prior_kind[k];
  const double a = m.prior_a[k], b = m.prior_b[k], lc = m.prior_logc[k];
  if (kind == SABC_PRIOR_NORMAL) {
    const double z = (x - a) / b;
    return -(z * z + SABC_LOG2PI) / 2.0 - lc;
  }
  if (kind == SABC_PRIOR_UNIFORM) return (x >= a && x <= b) ? -lc : -INFINITY;
  if (kind == SABC_PRIOR_EXPONENTIAL) return x >= 0.0 ? -x / a - lc : -INFINITY;
  if (kind == SABC_PRIOR_LOGNORMAL) {                           // LogNormal(mu = a, sigma = b)
    if (!(x > 0.0)) return -INFINITY;
    const double lx = log(x), z = (lx - a) / b;
    return -(z * z + SABC_LOG2PI) / 2.0 - lc - lx;
  }
  if (kind == SABC_PRIOR_GAMMA)                                 // Gamma(shape a, scale b)
    return x > 0.0 ? (a - 1.0) * log(x) - x / b - lc : -INFINITY;
  if (kind == SABC_PRIOR_BETA)                                  // Beta(a, b)
    return (x > 0.0 && x < 1.0) ? (a - 1.0) * log(x) + (b - 1.0) * log1p(-x) - lc : -INFINITY;
  if (kind == SABC_PRIOR_TRUNCNORMAL) {                         // truncated(Normal(a, b), c, d)
    if (!(x >= m.prior_c[k] && x <= m.prior_d[k])) return -INFINITY;
    const double z = (x - a) / b;
    return -(z * z + SABC_LOG2PI) / 2.0 - lc;
  }
  return -INFINITY;
}

// Gamma(shape, 1) by Marsaglia & Tsang (ACM TOMS 26, 2000); attempt t takes its normal from block base + 8 (2t) and its
// uniform from block base + 8 (2t + 1) of the particle's PRIOR stream (the stride 8 keeps the dimensions apart); shape < 1
// is drawn as Gamma(shape + 1) U^(1/shape) with U the second uniform of the first uniform block
__device__ __forceinline__ double gamma_sample(uint64_t seed, uint64_t pid, uint32_t base, double shape) {
  const double al = shape < 1.0 ? shape + 1.0 : shape;
  const double dd = al - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * dd);
  double g = dd, boost = 1.0;
  for (uint32_t t = 0; t < 64u; ++t) {
    double z0, z1;
    box_muller(stream_block(seed, pid, PURPOSE_PRIOR, 0, base + 8u * (2u * t)), z0, z1);
    const u32x4 w = stream_block(seed, pid, PURPOSE_PRIOR, 0, base + 8u * (2u * t + 1u));
    if (t == 0) boost = u52(w.z, w.w);
    const double v1 = 1.0 + cc * z0;
    if (!(v1 > 0.0)) continue;
    const double v = v1 * v1 * v1;
    g = dd * v;
    if (log(u52(w.x, w.y)) < 0.5 * z0 * z0 + dd - dd * v + dd * log(v)) break;
  }
  return shape < 1.0 ? g * pow(boost, 1.0 / shape) : g;
}

// one dimension of rand(prior) (:174): dimension k draws from blocks k, k + 8, k + 16, ... of the PRIOR stream
__device__ __forceinline__ double prior_sample_dim(const ModelDesc &m, int k, uint64_t pid) {
  const int kind = m.prior_kind[k];
  const double a = m.prior_a[k], b = m.prior_b[k];
  if (kind == SABC_PRIOR_GAMMA) return b * gamma_sample(m.seed, pid, (uint32_t)k, a);
  if (kind == SABC_PRIOR_BETA) {
    const double x = gamma_sample(m.seed, pid, (uint32_t)k, a), y = gamma_sample(m.seed, pid, (uint32_t)k + (1u << 16), b);
    return x / (x + y);
  }
  const u32x4 w = stream_block(m.seed, pid, PURPOSE_PRIOR, 0, (uint32_t)k);
  if (kind == SABC_PRIOR_NORMAL || kind == SABC_PRIOR_LOGNORMAL) {
    double z0, z1;
    box_muller(w, z0, z1);
    const double x = a + b * z0;
    return kind == SABC_PRIOR_LOGNORMAL ? exp(x) : x;
  }
  const double ua = u52(w.x, w.y);
  if (kind == SABC_PRIOR_EXPONENTIAL) return -a * log(ua);
  if (kind == SABC_PRIOR_TRUNCNORMAL) {                         // inverse CDF on [Phi(lo'), Phi(hi')]
    // bounds in the upper tail (lo' > 0, k0 stored negative = -Phi(-lo')): drawn in the mirrored lower tail, where the
    // CDF keeps its digits (Phi(lo') rounds to 1 beyond ~8 sigma and every draw would land on the bound)
    const double k0 = m.prior_k0[k];
    const double x = k0 < 0.0 ? a - b * hostmath::norm_quantile(-k0 - ua * m.prior_k1[k])
                              : a + b * hostmath::norm_quantile(k0 + ua * m.prior_k1[k]);
    return fmin(fmax(x, m.prior_c[k]), m.prior_d[k]);
  }
  return a + (b - a) * ua;
}

// MvNormal(mu, L L'): logpdf = -1/2 |L^-1 (x - mu)|^2 - (d/2 log 2 pi + sum log L_kk) by forward substitution;
// rand = mu + L z with z_k the first normal of block k of the PRIOR stream (a diagonal L gives the product of Normals)
__device__ __forceinline__ double mvnormal_logpdf(const ModelDesc &m, int d, const double *th) {
  double y[kMaxJointPara], q = 0.0;
  for (int k = 0; k < d; ++k) {
    double r = th[k] - m.prior_a[k];
    for (int l = 0; l < k; ++l) r -= m.prior_L[k * d + l] * y[l];
    y[k] = r / m.prior_L[k * d + k];
    q += y[k] * y[k];
  }
  return -0.5 * q - m.prior_joint_logc;
}

__device__ __forceinline__ void mvnormal_sample(const ModelDesc &m, int d, uint64_t pid, double *th) {
  double z[kMaxJointPara];
  for (int k = 0; k < d; ++k) {
    double z1;
    box_muller(stream_block(m.seed, pid, PURPOSE_PRIOR, 0, (uint32_t)k), z[k], z1);
  }
  for (int k = 0; k < d; ++k) {
    double x = m.prior_a[k];
    for (int l = 0; l <= k; ++l) x += m.prior_L[k * d + l] * z[l];
    th[k] = x;
  }
}

}  // namespace sabc
// SABC_USER_PRIOR (set by rtc.cpp when the handle's prior_joint is 3): the prior comes with the user's simulator source --
// ANY distribution next to a device-coded f_dist, evaluated inside the fused kernel like the built-in families (the
// reference takes any Distributions.Distribution at SimulatedAnnealingABC.jl:151,174,314,318).  The source defines, at
// global scope, next to sabc_user_simulate:
//   __device__ void   sabc_user_prior_sample(const double *params, sabc::NormalStream &rng, double *theta_out);   // rand(prior)
//   __device__ double sabc_user_prior_logpdf(const double *theta, const double *params);    // -INFINITY outside the support
// `rng` is the particle's PRIOR stream (rng.next(): N(0,1); rng.uniform_pair(u0, u1): U(0,1)).
#if defined(SABC_USER_PRIOR)
__device__ void sabc_user_prior_sample(const double *params, sabc::NormalStream &rng, double *theta_out);
__device__ double sabc_user_prior_logpdf(const double *theta, const double *params);
#endif
namespace sabc {

template <int D>
__device__ __forceinline__ double prior_logpdf(const ModelDesc &m, const double *th) {
#if defined(SABC_USER_PRIOR)
  if (m.prior_joint == 3) return ::sabc_user_prior_logpdf(th, m.p);
#endif
  if (D > 1 && m.prior_joint == 1) return mvnormal_logpdf(m, D, th);
  double lp = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double l = prior_logpdf_dim(m, k, th[k]);
    lp = (l > -INFINITY && lp > -INFINITY) ? lp + l : -INFINITY;
  }
  return lp;
}

// rand(prior) at :174
template <int D>
__device__ __forceinline__ void prior_sample(const ModelDesc &m, uint64_t pid, double *th) {
#if defined(SABC_USER_PRIOR)
  if (m.prior_joint == 3) {
    NormalStream ns(m.seed, pid, PURPOSE_PRIOR, 0);
    ::sabc_user_prior_sample(m.p, ns, th);
    return;
  }
#endif
  if (D > 1 && m.prior_joint == 1) { mvnormal_sample(m, D, pid, th); return; }
#pragma unroll
  for (int k = 0; k < D; ++k) th[k] = prior_sample_dim(m, k, pid);
}

// ---- empirical CDF: knots T[0..len), ordinates k/(len-1), piecewise linear, flat outside ----
__device__ __forceinline__ double cdf_apply(const double *__restrict__ T, int64_t len, double x) {
  if (!(x >= T[0])) return (x != x) ? x : 0.0;
  if (x > T[len - 1]) return 1.0;
  int64_t lo = 0, hi = len;               // lo = #knots < x
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (T[mid] < x) lo = mid + 1; else hi = mid;
  }
  const int64_t i0 = lo > 0 ? lo - 1 : 0;
  const double L1 = (double)(len - 1);
  const double y0 = (double)i0 / L1, y1 = (double)(i0 + 1) / L1;
  const double t = (x - T[i0]) / (T[i0 + 1] - T[i0]);     // weight form, t in [0, 1]: no overflow for close knots
  return y0 + t * (y1 - y0);
}

// ---- the same function with an index over the knot table -------------------------------------
// A plain binary search into an 8 MB table (n = 1e6) touches ~7 distinct 128-byte lines per lookup, most of
// them outside the 4 MB L2 of the XCD.  Index levels (built once per table, k_cdf_index):
//   C  coarse, LDS:    C[k] = T[k << shift], COARSE entries (+inf padded)              -> 10-11 LDS steps
//   M  mid, L2-hot:    M[m] = T[m << 4] (every 16th knot = first knot of each line)    -> shift - 4 steps in <= 0.5 MB
//   T  the table:      the 15 knots behind T[m << 4], all in ONE line (+inf padded)     -> 4 steps in one line
// so a lookup costs one line of the big table.  Every step tests the same monotone predicate T[i] < x as the
// plain search, so the rank -- and therefore u -- is bit-identical (duplicated knots included).
constexpr int kCdfLineShift = 4;           // 16 doubles = one 128-byte line

__device__ __forceinline__ double cdf_interp(const double *__restrict__ T, int64_t len, int64_t lo, double x) {
  const int64_t i0 = lo > 0 ? lo - 1 : 0;
  const double L1 = (double)(len - 1);
  const double y0 = (double)i0 / L1, y1 = (double)(i0 + 1) / L1;
  const double t = (x - T[i0]) / (T[i0 + 1] - T[i0]);     // weight form, t in [0, 1]: no overflow for close knots
  return y0 + t * (y1 - y0);
}

// p = last index in [a, a + 2^steps) with T[p] < x, given T[a] < x (branchless count form)
__device__ __forceinline__ int64_t cdf_advance(const double *__restrict__ T, int64_t a, int steps, double x) {
  for (int step = 1 << steps >> 1; step >= 1; step >>= 1)
    if (T[a + step] < x) a += step;
  return a;
}

// coarse (LDS, COARSE entries) -> mid -> line
template <int COARSE>
__device__ __forceinline__ double cdf_apply_3level(const double *__restrict__ T, int64_t len, int shift, const double *C,
                                                   const double *__restrict__ M, double x) {
  if (!(x >= T[0])) return (x != x) ? x : 0.0;
  if (x > T[len - 1]) return 1.0;
  int c = 0;                               // c = #coarse entries < x
#pragma unroll
  for (int step = COARSE >> 1; step >= 1; step >>= 1)
    if (C[c + step - 1] < x) c += step;
  if (C[c] < x) c += 1;                    // the last entry (index COARSE-1) is only reachable here
  int64_t lo = 0;                          // lo = #knots < x
  if (c > 0) {
    int64_t a = (int64_t)(c - 1) << shift; // T[a] < x, and T[a + 2^shift] >= x or beyond the table (+inf padding)
    if (shift > kCdfLineShift) {
      const int64_t m = cdf_advance(M, a >> kCdfLineShift, shift - kCdfLineShift, x);
      a = cdf_advance(T, m << kCdfLineShift, kCdfLineShift, x);
    } else {
      a = cdf_advance(T, a, shift, x);
    }
    lo = a + 1;
  }
  return cdf_interp(T, len, lo, x);
}

// A table that fits the coarse level whole (shift = 0: C[k] = T[k], +inf behind len -- populations of up to ~1000 particles with
// one statistic, ~2000 with more): the lookup never leaves LDS.  The same steps as cdf_apply_3level at shift = 0 (its second and
// third level are empty there), the same interpolation on the same knots: the same u.  (k_update_persistent only.)
// (the table is addressed as LDS explicitly: where the optimiser merges this lookup's tail with cdf_apply_3level's -- the same
// interpolation on knots from memory -- a pointer that is "LDS or memory" would need an aperture test per load)
typedef const __attribute__((address_space(3))) double *lds_knots_ptr;
template <int COARSE>
__device__ __forceinline__ double cdf_apply_lds(const double *C_generic, const int64_t len, const double x) {
  const lds_knots_ptr C = (lds_knots_ptr)C_generic;
  const double first = C[0], last = C[len - 1];
  int c = 0;                               // c = #knots < x
#pragma unroll
  for (int step = COARSE >> 1; step >= 1; step >>= 1) c += C[c + step - 1] < x ? step : 0;
  c += C[c] < x ? 1 : 0;
  const int i0 = c > 0 ? c - 1 : 0, i1 = i0 + 1 < COARSE ? i0 + 1 : COARSE - 1;
  const double k0 = C[i0], k1 = C[i1];
  const double L1 = (double)(len - 1);
  const double y0 = (double)i0 / L1, y1 = (double)(i0 + 1) / L1;
  const double t = (x - k0) / (k1 - k0);
  const double v = y0 + t * (y1 - y0);
  return !(x >= first) ? ((x != x) ? x : 0.0) : (x > last ? 1.0 : v);
}

// The S lookups of ONE particle (one per statistic, each into its own table), step by step together: where a wave has the SIMD
// to itself (k_update_persistent: small shards) a lookup is a chain of ~11 LDS and 2-6 memory round trips that nothing hides,
// and the S chains are independent -- in flight together they cost one chain, not S.  Every step tests the predicate of
// cdf_apply_3level: the same ranks, the same u.
template <int S, int COARSE>
__device__ __forceinline__ void cdf_apply_3level_lockstep(const CdfPtrs &cdf, const double (&C)[S][COARSE], const double *x, double *u) {
  const double *T[S], *M[S];
  double first[S], last[S];
  int c[S];
  bool same = true;
#pragma unroll
  for (int j = 0; j < S; ++j) {
    T[j] = cdf.knots + (int64_t)j * cdf.stride;
    M[j] = cdf.mid + (int64_t)j * cdf.mid_stride;
    c[j] = 0;
    same = same && cdf.shift[j] == cdf.shift[0];
  }
  const bool lds_knots = same && cdf.shift[0] == 0;                   // (the tables fit the coarse level whole: cdf_apply_lds)
  if (lds_knots) {                                                    // (two loops, not a select: no LDS-or-memory pointer)
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const lds_knots_ptr Cj = (lds_knots_ptr)&C[j][0];
      first[j] = Cj[0];
      last[j] = Cj[cdf.len[j] - 1];
    }
  } else {
#pragma unroll
    for (int j = 0; j < S; ++j) { first[j] = T[j][0]; last[j] = T[j][cdf.len[j] - 1]; }
  }
#pragma unroll
  for (int step = COARSE >> 1; step >= 1; step >>= 1) {
#pragma unroll
    for (int j = 0; j < S; ++j) c[j] += C[j][c[j] + step - 1] < x[j] ? step : 0;
  }
  int64_t a[S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    c[j] += C[j][c[j]] < x[j] ? 1 : 0;
    a[j] = c[j] > 0 ? (int64_t)(c[j] - 1) << cdf.shift[j] : 0;       // (c = 0: nothing below x, the steps below are not taken)
  }
  if (same) {                                                         // (uniform: the tables of a population have one length)
    const int sh = cdf.shift[0];
    if (sh > kCdfLineShift) {
      int64_t mm[S];
#pragma unroll
      for (int j = 0; j < S; ++j) mm[j] = a[j] >> kCdfLineShift;
      for (int step = 1 << (sh - kCdfLineShift) >> 1; step >= 1; step >>= 1) {
#pragma unroll
        for (int j = 0; j < S; ++j) mm[j] += (c[j] > 0 && M[j][mm[j] + step] < x[j]) ? step : 0;
      }
#pragma unroll
      for (int j = 0; j < S; ++j) a[j] = mm[j] << kCdfLineShift;
#pragma unroll
      for (int step = 1 << kCdfLineShift >> 1; step >= 1; step >>= 1) {
#pragma unroll
        for (int j = 0; j < S; ++j) a[j] += (c[j] > 0 && T[j][a[j] + step] < x[j]) ? step : 0;
      }
    } else {
      for (int step = 1 << sh >> 1; step >= 1; step >>= 1) {
#pragma unroll
        for (int j = 0; j < S; ++j) a[j] += (c[j] > 0 && T[j][a[j] + step] < x[j]) ? step : 0;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      if (c[j] > 0) {
        if (cdf.shift[j] > kCdfLineShift) {
          const int64_t m = cdf_advance(M[j], a[j] >> kCdfLineShift, cdf.shift[j] - kCdfLineShift, x[j]);
          a[j] = cdf_advance(T[j], m << kCdfLineShift, kCdfLineShift, x[j]);
        } else {
          a[j] = cdf_advance(T[j], a[j], cdf.shift[j], x[j]);
        }
      }
    }
  }
  double k0[S], k1[S];
  int64_t i0[S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const int64_t lo = c[j] > 0 ? a[j] + 1 : 0;
    i0[j] = lo > 0 ? lo - 1 : 0;
    if (lds_knots) {
      const lds_knots_ptr Cj = (lds_knots_ptr)&C[j][0];
      k0[j] = Cj[i0[j]];
      k1[j] = Cj[i0[j] + 1 < COARSE ? i0[j] + 1 : COARSE - 1];
    } else {
      k0[j] = T[j][i0[j]];
      k1[j] = T[j][i0[j] + 1];
    }
  }
#pragma unroll
  for (int j = 0; j < S; ++j) {                                        // cdf_interp, on the knots in hand
    const double L1 = (double)(cdf.len[j] - 1);
    const double y0 = (double)i0[j] / L1, y1 = (double)(i0[j] + 1) / L1;
    const double t = (x[j] - k0[j]) / (k1[j] - k0[j]);
    const double v = y0 + t * (y1 - y0);
    u[j] = !(x[j] >= first[j]) ? ((x[j] != x[j]) ? x[j] : 0.0) : (x[j] > last[j] ? 1.0 : v);
  }
}

// mid -> line, for callers without the LDS copy of the coarse level (one lookup per lane, g-and-k / host mode)
__device__ __forceinline__ double cdf_apply_mid(const double *__restrict__ T, int64_t len, const double *__restrict__ M,
                                                double x) {
  if (!(x >= T[0])) return (x != x) ? x : 0.0;
  if (x > T[len - 1]) return 1.0;
  int64_t lo = 0;
  if (T[0] < x) {
    int64_t mlo = 0, mhi = (len + 15) >> kCdfLineShift;   // last m in [0, mhi) with M[m] < x; M[0] = T[0] < x
    while (mhi - mlo > 1) {
      const int64_t mid = mlo + ((mhi - mlo) >> 1);
      if (M[mid] < x) mlo = mid; else mhi = mid;
    }
    lo = cdf_advance(T, mlo << kCdfLineShift, kCdfLineShift, x) + 1;
  }
  return cdf_interp(T, len, lo, x);
}

// N lookups into the SAME table, step by step together: every step of the search is a dependent read (16 in the mid
// level + 4 in the line + the interpolation's pair at n = 1e6) and the N values are independent, so the N reads of a step
// are in flight together -- a chain of ~22 round trips instead of ~22 N.  The mid level is walked in the power-of-two
// "advance" form (uniform trip count); it tests the same monotone predicate as cdf_apply_mid's bisection, so the rank --
// and u -- is bit-identical.  A lane with nothing to look up passes any finite x and ignores u.
template <int N>
__device__ __forceinline__ void cdf_apply_mid_lockstep(const double *__restrict__ T, int64_t len, const double *__restrict__ M,
                                                       const double *x, double *u) {
  const int64_t mcount = (len + 15) >> kCdfLineShift;     // entries of M that lie inside the table
  int64_t top = 1;
  while (top < mcount) top <<= 1;
  int64_t a[N];
#pragma unroll
  for (int p = 0; p < N; ++p) a[p] = 0;                   // last m with M[m] < x, given M[0] = T[0] < x (else discarded)
  for (int64_t step = top >> 1; step >= 1; step >>= 1) {
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const int64_t q = a[p] + step;
      const int64_t qs = q < mcount ? q : mcount - 1;      // a probe behind the table reads its last entry and is refused
      const double v = M[qs];
      a[p] = (q < mcount && v < x[p]) ? q : a[p];
    }
  }
#pragma unroll
  for (int p = 0; p < N; ++p) a[p] <<= kCdfLineShift;
#pragma unroll
  for (int step = 1 << kCdfLineShift >> 1; step >= 1; step >>= 1) {
#pragma unroll
    for (int p = 0; p < N; ++p) a[p] += (T[a[p] + step] < x[p]) ? step : 0;      // (+inf behind len)
  }
  const double first = T[0], last = T[len - 1], L1 = (double)(len - 1);
  double t0[N], t1[N];
  int64_t i0[N];
#pragma unroll
  for (int p = 0; p < N; ++p) {
    const int64_t lo = first < x[p] ? a[p] + 1 : 0;        // lo = #knots < x
    i0[p] = lo > 0 ? lo - 1 : 0;
    t0[p] = T[i0[p]]; t1[p] = T[i0[p] + 1];
  }
#pragma unroll
  for (int p = 0; p < N; ++p) {
    const double y0 = (double)i0[p] / L1, y1 = (double)(i0[p] + 1) / L1;
    const double t = (x[p] - t0[p]) / (t1[p] - t0[p]);     // cdf_interp
    const double v = y0 + t * (y1 - y0);
    u[p] = !(x[p] >= first) ? ((x[p] != x[p]) ? x[p] : 0.0) : (x[p] > last ? 1.0 : v);
  }
}

__device__ __forceinline__ double finite_or_big(double v) { return isfinite(v) ? v : 1e30; }

// ---- simulators ----
template <int MODEL, int D, int S>
struct Sim;

// y_1..n_obs ~ Normal(theta1, sd); rho1 = |obs_mean - mean(y)|, rho2 = |obs_m2 - mean(y^2)|
// (test/runtests.jl:35,86,128-131,167-170; BASELINE configs 1-2)
template <int D, int S>
struct Sim<SABC_MODEL_GAUSS_IID, D, S> {
  static __device__ __forceinline__ void run(const ModelDesc &m, const double *th, uint64_t pid, uint64_t iter,
                                             double *rho, int coop = 0) {
    const int n_obs = (int)m.p[0];
    const double mu = th[0];
    const double sd = (D >= 2) ? th[1] : m.p[1];
    NormalStream ns(m.seed, pid, PURPOSE_SIM, iter, coop);
    // x_i = mu + sd z_i: the loop sums z and z^2 only; sum x = n mu + sd sum z, sum x^2 = n mu^2 + 2 mu sd sum z + sd^2 sum z^2
    double sz = 0.0, szz = 0.0;
    const int n_pairs = n_obs >> 1;
    ns.for_pairs(n_pairs, [&](const double z0, const double z1) {
      sz += z0; if (S >= 2) szz = fma(z0, z0, szz);
      sz += z1; if (S >= 2) szz = fma(z1, z1, szz);
    });
    if (n_obs & 1) {
      double z0, z1;
      ns.pair(z0, z1);
      sz += z0; if (S >= 2) szz = fma(z0, z0, szz);
    }
    const double n = (double)n_obs;
    rho[0] = fabs(m.p[2] - (mu + sd * sz / n));
    if (S >= 2) rho[1] = fabs(m.p[3] - (mu * mu + (2.0 * mu * sd * sz + sd * sd * szz) / n));
  }
};

// x_1..n_obs ~ N(theta, [[1,r],[r,1]]); rho = (||mean - obs||, |var1+var2 - obs|, |cov12 - obs|)
template <int D, int S>
struct Sim<SABC_MODEL_GAUSS2D, D, S> {
  static __device__ __forceinline__ void run(const ModelDesc &m, const double *th, uint64_t pid, uint64_t iter,
                                             double *rho, int coop = 0) {
    const int n_obs = (int)m.p[0];
    const double r = m.p[1], c = sqrt(1.0 - r * r);
    NormalStream ns(m.seed, pid, PURPOSE_SIM, iter, coop);
    // e1 = za, e2 = r za + c zb: the loop sums the raw moments of (za, zb); the correlated ones follow by linearity
    double A = 0, B = 0, AA = 0, BB = 0, AB = 0;
    ns.for_pairs(n_obs, [&](const double za, const double zb) {
      A += za; B += zb; AA = fma(za, za, AA); BB = fma(zb, zb, BB); AB = fma(za, zb, AB);
    });
    const double S1 = A, S2 = r * A + c * B;
    const double Q11 = AA, Q22 = r * r * AA + 2.0 * r * c * AB + c * c * BB, Q12 = r * AA + c * AB;
    const double m1 = S1 / n_obs, m2 = S2 / n_obs;
    const double var1 = (Q11 - S1 * m1) / (n_obs - 1), var2 = (Q22 - S2 * m2) / (n_obs - 1);
    const double cov = (Q12 - S1 * m2) / (n_obs - 1);
    const double d1 = th[0] + m1 - m.p[2], d2 = th[1] + m2 - m.p[3];
    rho[0] = sqrt(d1 * d1 + d2 * d2);
    if (S >= 2) rho[1] = fabs(var1 + var2 - m.p[4]);
    if (S >= 3) rho[2] = fabs(cov - m.p[5]);
  }
};

#if !defined(__HIPCC_RTC__)   // the wave-cooperative g-and-k simulator has its own kernel (kernels.hip); a run-time compiled
                              // user simulator never needs it, and older hipRTC compilers lack some of its builtins
// g-and-k: x = A + B (1 + c tanh(g z / 2)) (1 + z^2)^k z; rho_j = |x_(rank_j) - obs_j|.
// Wave-cooperative: ONE WAVEFRONT PER PARTICLE.  Lane l draws Philox block l of the particle's
// SIM stream = normals 2l and 2l+1 = draws 2l and 2l+1 (n_draws <= 128 = 64 lanes x 2); the 128
// values are sorted in registers by a bitonic network across the wave (cross-lane exchanges via
// __shfl_xor) and the requested order statistics are read out of the owning lanes.  All 64 lanes
// must call this with the same (th, pid, iter).
constexpr int kGkMaxDraws = 128;
// particles a wave takes through its lane-parallel phases (proposal / prior gate, ECDF, accept) between the simulations,
// which it does one particle at a time: the lane-parallel phases cost the same for 16 busy lanes as for 64.
// Measured on cfg4 at n = 1e6 (tools/exp_ab2.sh, two runs each): 16 per wave 610 us, 32: 544 us, 64: 540 us.
#ifndef SABC_GK_PW
#define SABC_GK_PW 64
#endif
constexpr int kGkParticlesPerWave = SABC_GK_PW;

// (1 + z^2)^k as exp(k log(1 + z^2)): the argument of the log is >= 1 and normal, so the
// table-driven log of device_rng.hpp applies (relative error ~ k log(1+z^2) * 2e-16)
__device__ __forceinline__ double gk_quantile(const double *th, double c, double z) {
  const double w = exp_tab(-0.5 * th[3] * neg2_log_tab(fma(z, z, 1.0)));
  return th[0] + th[1] * (1.0 + c * tanh_abs_tab(th[2] * z / 2.0)) * w * z;
}

// v from lane ^ DIST, DIST a compile-time power of two: DPP quad_perm for 1 and 2 (register crossbar, no
// LDS hardware, no address VGPR), ds_swizzle bit-mask mode for 4, 8, 16 (no address VGPR), ds_bpermute for 32.
// Measured on cfg4 (k_update_gk, n = 1e6): everything through ds_swizzle 910 us; DPP for 1, 2 (this) 863 us;
// DPP also for 4 (half_mirror + quad_perm) and 8 (row_ror:8) 871 us.
template <int DIST>
__device__ __forceinline__ double xor_lane(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (DIST == 1) {
    lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);           // quad_perm [1,0,3,2]; every lane is written
    hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  } else if constexpr (DIST == 2) {
    lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);           // quad_perm [2,3,0,1]
    hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
  } else if constexpr (DIST <= 16) {
    constexpr int pattern = 0x1F | (DIST << 10);                        // and_mask 0x1f, or 0, xor DIST
    lo = __builtin_amdgcn_ds_swizzle(lo, pattern);
    hi = __builtin_amdgcn_ds_swizzle(hi, pattern);
  } else {
    return __shfl_xor(v, DIST, 64);
  }
  return __hiloint2double(hi, lo);
}

// One compare-exchange step of the bitonic network.  Written with a compare and selects instead of
// fmin / fmax: the values are never NaN, and min/max would each be preceded by a canonicalising
// v_max_f64 x, x (a third of the sort's instructions).  Equal values may be taken from either side.
// Lane distances 16 and 32 without the LDS crossbar: v_permlane16_swap / v_permlane32_swap (new on gfx950) exchange
// the upper half of one register with the lower half of another.  Swapping v0's upper half (odd rows) with v1's
// lower half (even rows) leaves every lane with BOTH elements of one pair -- the lower lanes hold the v0 pair, the
// upper lanes the v1 pair, a = the lower lane's element, b = the upper lane's --, the pair is ordered locally, and
// the same swap puts the elements back.  One compare per lane instead of two, no ds_bpermute / ds_swizzle round trip.
template <int DIST>
__device__ __forceinline__ void half_swap(double &a, double &b) {
  static_assert(DIST == 16 || DIST == 32, "row-pair or half-wave swap");
  unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  if constexpr (DIST == 32) {
    const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
  } else {
    const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
  }
  a = __hiloint2double((int)ahi, (int)alo);
  b = __hiloint2double((int)bhi, (int)blo);
}

// Lane-index bits as wave masks (bit l of lane_bit(i) = bit i of lane l): compile-time constants.  Every direction /
// side predicate of the network is an XOR of two of them, so the selects below take their condition from a scalar
// mask (v_cmp -> s_xor with a constant -> v_cndmask) instead of comparing a materialised 0/1 value with a per-lane 0/1
// vector (v_cmp -> v_cndmask 0,1 -> v_cmp_eq_u32 -> v_cndmask): 2 VALU instructions and their hazard nops less per
// element and cross-lane step, and no registers held for the predicates.  The boolean algebra is the one of the plain
// form ((p < v) == keep_min), so ties behave identically.
constexpr uint64_t lane_bit(int i) {
  return i == 0 ? 0xAAAAAAAAAAAAAAAAull : i == 1 ? 0xCCCCCCCCCCCCCCCCull : i == 2 ? 0xF0F0F0F0F0F0F0F0ull
       : i == 3 ? 0xFF00FF00FF00FF00ull : i == 4 ? 0xFFFF0000FFFF0000ull : 0xFFFFFFFF00000000ull;
}
constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }
// lanes that sort DESCENDING in the stage that builds runs of K elements: bit log2(K) - 1 of the lane (none for K = 128)
constexpr uint64_t descending_lanes(int K) { return K >= 128 ? 0ull : lane_bit(ilog2(K) - 1); }

// One compare-exchange step of the bitonic network.  Written with a compare and selects instead of
// fmin / fmax: the values are never NaN, and min/max would each be preceded by a canonicalising
// v_max_f64 x, x (a third of the sort's instructions).  Equal values may be taken from either side.
template <int K, int J>
__device__ __forceinline__ void bitonic_step(double &v0, double &v1) {
  constexpr int kOGT = 2, kOLT = 4;                    // LLVM FCmp predicates of __builtin_amdgcn_fcmp
  if constexpr (J == 1 || (J >> 1) >= 16) {
    // the lane holds both elements of a pair: its own (J == 1), or after v_permlane{16,32}_swap (distance 16, 32)
    if constexpr (J > 1) half_swap<(J >> 1)>(v0, v1);
    // swap iff (v0 > v1) == ascending; (both lanes of a swapped pair see the same direction: K > J)
    const bool swap = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_fcmp(v0, v1, kOGT) ^ descending_lanes(K));
    const double a = swap ? v1 : v0, b = swap ? v0 : v1;
    v0 = a; v1 = b;
    if constexpr (J > 1) half_swap<(J >> 1)>(v0, v1);
  } else {
    constexpr int dist = J >> 1;
    const double p0 = xor_lane<dist>(v0), p1 = xor_lane<dist>(v1);
    // keep_min = ((lane & dist) == 0) == ascending; take the partner's value iff (p < v) == keep_min, i.e. iff
    // (p < v) XOR bit_dist(lane) XOR descending(lane); the partner evaluates the mirrored test, so the pair {v, p} is preserved
    constexpr uint64_t m = lane_bit(ilog2(dist)) ^ descending_lanes(K);
    const bool t0 = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_fcmp(p0, v0, kOLT) ^ m);
    const bool t1 = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_fcmp(p1, v1, kOLT) ^ m);
    v0 = t0 ? p0 : v0;
    v1 = t1 ? p1 : v1;
  }
}

template <int K, int J>
__device__ __forceinline__ void bitonic_merge(double &v0, double &v1) {
  bitonic_step<K, J>(v0, v1);
  if constexpr (J > 1) bitonic_merge<K, J / 2>(v0, v1);
}

template <int K>
__device__ __forceinline__ void bitonic_sort128(double &v0, double &v1) {
  if constexpr (K > 2) bitonic_sort128<K / 2>(v0, v1);
  bitonic_merge<K, K / 2>(v0, v1);
}

// the same network over TWO independent sets of 128 values, step by step: the exchanges of one set (a trip through the LDS
// crossbar or a permlane swap) are in flight while the other set's compare and selects issue
template <int K, int J>
__device__ __forceinline__ void bitonic_merge_x2(double &a0, double &a1, double &b0, double &b1) {
  bitonic_step<K, J>(a0, a1);
  bitonic_step<K, J>(b0, b1);
  if constexpr (J > 1) bitonic_merge_x2<K, J / 2>(a0, a1, b0, b1);
}

template <int K>
__device__ __forceinline__ void bitonic_sort128_x2(double &a0, double &a1, double &b0, double &b1) {
  if constexpr (K > 2) bitonic_sort128_x2<K / 2>(a0, a1, b0, b1);
  bitonic_merge_x2<K, K / 2>(a0, a1, b0, b1);
}

// Only ORDER STATISTICS are wanted, and when every wanted rank is a multiple of 16 (BASELINE config 4: 16, 48, 80, 112 of
// 128) the last four steps of the final merge can go: after its steps at element distance 64, 32 and 16 every aligned block
// of 16 elements (8 lanes) holds exactly the ranks 16 b + 1 .. 16 b + 16 (as a bitonic sequence), so the order statistic
// of rank 16 (b + 1) is the block's MAXIMUM -- one value per lane through three exchanges (lane distance 1, 2, 4) instead
// of two values through three exchanges and a local step, each with a compare and two selects.  Bit-identical: the
// maximum of the block is the element the full sort would have put at its end.
template <int K, int J, int JMIN>
__device__ __forceinline__ void bitonic_merge_until_x2(double &a0, double &a1, double &b0, double &b1) {
  bitonic_step<K, J>(a0, a1);
  bitonic_step<K, J>(b0, b1);
  if constexpr (J > JMIN) bitonic_merge_until_x2<K, J / 2, JMIN>(a0, a1, b0, b1);
}
__device__ __forceinline__ double block16_max(double v0, double v1) {
  double v = v1 > v0 ? v1 : v0;
  double p = xor_lane<1>(v); v = p > v ? p : v;
  p = xor_lane<2>(v); v = p > v ? p : v;
  p = xor_lane<4>(v); v = p > v ? p : v;
  return v;
}
// value held by lane `l` (uniform over the wave), as a uniform value: v_readlane_b32 x 2, no trip through the LDS crossbar
__device__ __forceinline__ double read_lane(double v, int l) {
  const int ul = __builtin_amdgcn_readfirstlane(l);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), ul), __builtin_amdgcn_readlane(__double2loint(v), ul));
}

// The quantile function is INCREASING in z when B > 0, k >= 0 and 0 <= c <= 0.83: d/dz of z (1 + c tanh(g z / 2)) (1 + z^2)^k
// is (1 + z^2)^k [(1 + c t)(1 + 2 k z^2 / (1 + z^2)) + c w sech^2 w] with w = g z / 2, t = tanh w, and
// tanh a + a sech^2 a <= 1.1997, so the bracket is >= 1 - 1.1997 c > 0 for every g.  Then the order statistics of the
// simulated data are the quantile function OF the order statistics of the normals: sort(x)[r] = Q(sort(z)[r]) -- the
// 128 normals are sorted and Q is evaluated at the S wanted ranks only (by the caller, for 16 particles x 4 ranks per
// pass over the lanes) instead of 128 times per particle.  Outside that region the data themselves are sorted, as the
// reference does (the oracle always does).  Rounding can order two normals closer than ~1e-14 differently from their
// images; the order statistic then differs in its last digits only.
__device__ __forceinline__ bool gk_increasing(const double *th, double c) {
  return th[1] > 0.0 && th[3] >= 0.0 && c >= 0.0 && c <= 0.83;
}

// |Q(z) - obs| for an order statistic z of the normals (+inf: a rank behind n_draws)
__device__ __forceinline__ double gk_rho_of_normal(const double *th, double c, double z, double obs) {
  const double x = z < INFINITY ? gk_quantile(th, c, z) : INFINITY;
  return finite_or_big(fabs(x - obs));
}

// out[j]: gk_increasing(th, c) ? the normal of rank j (the caller applies gk_rho_of_normal) : rho_j itself.
// All 64 lanes must call this with the same (th, pid, iter).
template <int S>
__device__ __forceinline__ void gk_simulate_wave_ranks(const ModelDesc &m, const double *th, uint64_t pid, uint64_t iter,
                                                       double *out) {
  const int lane = threadIdx.x & 63;
  const int n_draws = (int)m.p[0];
  const double c = m.p[1];
  const bool inc = gk_increasing(th, c);                // uniform over the wave
  double z0, z1;
  box_muller(stream_block(m.seed, pid, PURPOSE_SIM, iter, (uint32_t)lane), z0, z1);
  if (!inc) { z0 = gk_quantile(th, c, z0); z1 = gk_quantile(th, c, z1); }
  const int i0 = 2 * lane, i1 = 2 * lane + 1;
  // bitonic sorting network over the 128 values, two per lane (element index = 2*lane + slot):
  // 28 compare-exchange steps, 7 of them inside the lane, 21 with the lane at distance j/2
  double v0 = i0 < n_draws ? z0 : INFINITY, v1 = i1 < n_draws ? z1 : INFINITY;
  bitonic_sort128<kGkMaxDraws>(v0, v1);
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const int want = (int)m.p[2 + j] - 1;               // 1-based order statistic -> sorted index (uniform)
    const double v = read_lane((want & 1) ? v1 : v0, want >> 1);
    out[j] = inc ? v : finite_or_big(fabs(v - m.p[2 + S + j]));
  }
}

// two particles at a time (A and B may be the same particle: an odd one out is simulated against itself)
template <int S>
__device__ __forceinline__ void gk_simulate_wave_ranks_x2(const ModelDesc &m, const double *thA, const double *thB, uint64_t pidA,
                                                          uint64_t pidB, uint64_t iter, double *outA, double *outB) {
  const int lane = threadIdx.x & 63;
  const int n_draws = (int)m.p[0];
  const double c = m.p[1];
  const bool incA = gk_increasing(thA, c), incB = gk_increasing(thB, c);     // uniform over the wave
  double a0, a1, b0, b1;
  box_muller(stream_block(m.seed, pidA, PURPOSE_SIM, iter, (uint32_t)lane), a0, a1);
  box_muller(stream_block(m.seed, pidB, PURPOSE_SIM, iter, (uint32_t)lane), b0, b1);
  if (!incA) { a0 = gk_quantile(thA, c, a0); a1 = gk_quantile(thA, c, a1); }
  if (!incB) { b0 = gk_quantile(thB, c, b0); b1 = gk_quantile(thB, c, b1); }
  const bool in0 = 2 * lane < n_draws, in1 = 2 * lane + 1 < n_draws;
  a0 = in0 ? a0 : INFINITY; a1 = in1 ? a1 : INFINITY;
  b0 = in0 ? b0 : INFINITY; b1 = in1 ? b1 : INFINITY;
  bool ranks_are_block_ends = true;                     // uniform: every wanted rank is a multiple of 16
#pragma unroll
  for (int j = 0; j < S; ++j) ranks_are_block_ends = ranks_are_block_ends && (((int)m.p[2 + j]) & 15) == 0;
  if (SABC_GK_BLOCKMAX && ranks_are_block_ends) {
    static_assert(kGkMaxDraws == 128, "the final merge is the one over 128 elements");
    bitonic_sort128_x2<kGkMaxDraws / 2>(a0, a1, b0, b1);                    // runs of 64, ascending | descending
    bitonic_merge_until_x2<kGkMaxDraws, kGkMaxDraws / 2, 16>(a0, a1, b0, b1);   // element distance 64, 32, 16
    const double ma = block16_max(a0, a1), mb = block16_max(b0, b1);
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int lane_of_block = (((int)m.p[2 + j] >> 4) - 1) * 8;           // rank 16 (b + 1): block b = lanes 8 b .. 8 b + 7
      const double va = read_lane(ma, lane_of_block), vb = read_lane(mb, lane_of_block);
      outA[j] = incA ? va : finite_or_big(fabs(va - m.p[2 + S + j]));
      outB[j] = incB ? vb : finite_or_big(fabs(vb - m.p[2 + S + j]));
    }
    return;
  }
  bitonic_sort128_x2<kGkMaxDraws>(a0, a1, b0, b1);
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const int want = (int)m.p[2 + j] - 1;               // 1-based order statistic -> sorted index (uniform)
    const double va = read_lane((want & 1) ? a1 : a0, want >> 1), vb = read_lane((want & 1) ? b1 : b0, want >> 1);
    outA[j] = incA ? va : finite_or_big(fabs(va - m.p[2 + S + j]));
    outB[j] = incB ? vb : finite_or_big(fabs(vb - m.p[2 + S + j]));
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 4: FOUR particles per wave at a time, one per row of 16 lanes, 8 of a particle's 128 values per lane.
//
// What the counters said of the two-values-per-lane network above (profiles/r03_pmc_sq_cfg4.csv): 82 % VALU busy, 44 % of the
// wave-cycles parked -- and per simulated particle ~3 x as many instructions in the sort as in the generator: 18 of its 24
// compare-exchange steps cross lanes, each one two moves, a compare and two selects PER ELEMENT.  With element e = 8 l + r
// (l = lane in the row, r = register) the three lowest distances are between REGISTERS -- 15 of the 24 steps never leave the
// lane, and a local compare-exchange is a v_min_f64 + a v_max_f64 -- and the remaining 9 pair lanes at distance 1, 2, 4, 8
// inside a row of 16: DPP quad_perm / ds_swizzle, no v_permlane, no second LDS round trip.  A cross-lane step is ONE v_min_f64
// per element by keeping the values SIGNED: with y = x in ascending and -x in descending lanes (the direction of a stage
// is a lane bit from runs of 8 on) every step of a stage orders ascending; and with t = y in the lane that keeps the
// minimum and -y in the lane that keeps the maximum both lanes evaluate min(t_own, -t_partner) (the negation is a source
// modifier): min(y_a, y_b) below, -max(y_a, y_b) above -- already in the sign the upper lane stores.  Between two steps the
// sign pattern changes by the XOR of two lane bits: one v_xor_b32 on the high word per element (13 such flips in all).
// Per 4 particles: 120 (local) + 168 (cross-lane, + 48 ds_swizzle) + 104 (flips) + 10 (block maxima) VALU instructions, i.e.
// ~100 per particle where the network above takes ~250.  The order statistics are the same numbers: a sorting network
// permutes, and min / max of doubles that are never NaN select.
//
// The wanted ranks have to be multiples of 16 (BASELINE config 4: 16, 48, 80, 112): after the final merge's steps at element
// distance 64, 32, 16 every pair of lanes (2 b, 2 b + 1) holds the ranks 16 b + 1 .. 16 b + 16, whose maximum is the order
// statistic of rank 16 (b + 1).  Other ranks take the network above.
__device__ __forceinline__ double gk_min_neg(double a, double b) {            // min(a, -b); (inline asm: a __builtin_fmin on
  double r;                                                                   // a shuffled value is preceded by a canonicalising
  asm("v_min_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));                     // v_max_f64 x, x)
  return r;
}
__device__ __forceinline__ double gk_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double gk_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// compare-exchange between the registers at distance 2^J of one lane; DESC_BIT >= 0: pairs whose lower index has that bit set
// order descending (the first two stages, whose direction is a register bit), -1: all ascending (on the signed values)
template <int J, int DESC_BIT>
__device__ __forceinline__ void gk_local_step(double (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i & (1 << J)) continue;
    const double a = v[i], b = v[i | (1 << J)];
    const double lo = gk_min(a, b), hi = gk_max(a, b);
    const bool desc = DESC_BIT >= 0 && ((i >> (DESC_BIT < 0 ? 0 : DESC_BIT)) & 1);
    v[i] = desc ? hi : lo;
    v[i | (1 << J)] = desc ? lo : hi;
  }
}
// ... with the lane at distance DIST of the row: every element becomes min(own, -partner's) (see above)
template <int DIST>
__device__ __forceinline__ void gk_cross_step(double (&v)[8]) {
  double p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = xor_lane<DIST>(v[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = gk_min_neg(v[i], p[i]);
}
__device__ __forceinline__ void gk_flip(double (&v)[8], int sign_mask) {      // sign_mask: 0x80000000 in the lanes that change sign
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __hiloint2double(__double2hiint(v[i]) ^ sign_mask, __double2loint(v[i]));
}

// Particle `my` (one per row of 16 lanes; the rows of a wave may hold the same particle) of the wave's staging area: its 128
// draws, the wanted order statistics -- ranks 16 (b + 1) -- to rp_lds[my][j]: the order statistic of the NORMALS where the
// quantile function is increasing (the caller maps it: gk_rho_of_normal), else rho_j itself.
template <int S>
__device__ __forceinline__ void gk_simulate_rows4(const ModelDesc &m, const double (*thp_lds)[4], double (*rp_lds)[S], const int my,
                                                  const uint64_t pid, const uint64_t iter) {
  const int lane = threadIdx.x & 63, rl = lane & 15;
  const int n_draws = (int)m.p[0];
  const double c = m.p[1];
  double th[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) th[k] = thp_lds[my][k];
  const bool inc = gk_increasing(th, c);                // uniform over the row
  double v[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) box_muller(stream_block(m.seed, pid, PURPOSE_SIM, iter, (uint32_t)(4 * rl + t)), v[2 * t], v[2 * t + 1]);
  if (!inc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = gk_quantile(th, c, v[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 8 * rl + i < n_draws ? v[i] : INFINITY;
  // lane bits of the row as sign masks
  const int b0 = (lane & 1) << 31, b1 = (lane & 2) << 30, b2 = (lane & 4) << 29, b3 = (lane & 8) << 28;
  // runs of 2 and 4: direction = a register bit
  gk_local_step<0, 1>(v);
  gk_local_step<1, 2>(v); gk_local_step<0, 2>(v);
  // runs of 8: descending in the lanes with bit 0 -- stored negated from here on (sign pattern in brackets)
  gk_flip(v, b0);                                                                             // [b0]
  gk_local_step<2, -1>(v); gk_local_step<1, -1>(v); gk_local_step<0, -1>(v);
  // runs of 16
  gk_flip(v, b1); gk_cross_step<1>(v);                                                        // [b1 ^ b0]
  gk_flip(v, b0); gk_local_step<2, -1>(v); gk_local_step<1, -1>(v); gk_local_step<0, -1>(v);  // [b1]
  // runs of 32
  gk_flip(v, b2); gk_cross_step<2>(v);                                                        // [b2 ^ b1]
  gk_flip(v, b1 ^ b0); gk_cross_step<1>(v);                                                   // [b2 ^ b0]
  gk_flip(v, b0); gk_local_step<2, -1>(v); gk_local_step<1, -1>(v); gk_local_step<0, -1>(v);  // [b2]
  // runs of 64
  gk_flip(v, b3); gk_cross_step<4>(v);                                                        // [b3 ^ b2]
  gk_flip(v, b2 ^ b1); gk_cross_step<2>(v);                                                   // [b3 ^ b1]
  gk_flip(v, b1 ^ b0); gk_cross_step<1>(v);                                                   // [b3 ^ b0]
  gk_flip(v, b0); gk_local_step<2, -1>(v); gk_local_step<1, -1>(v); gk_local_step<0, -1>(v);  // [b3]
  // the final merge (all ascending), element distance 64, 32, 16 only
  gk_cross_step<8>(v);                                                                        // [b3]
  gk_flip(v, b3 ^ b2); gk_cross_step<4>(v);                                                   // [b2]
  gk_flip(v, b2 ^ b1); gk_cross_step<2>(v);                                                   // [b1]
  gk_flip(v, b1);                                                                             // [0]: the values themselves
  // lanes (2 b, 2 b + 1) hold the ranks 16 b + 1 .. 16 b + 16: their maximum is the order statistic of rank 16 (b + 1)
  double bm = gk_max(gk_max(gk_max(v[0], v[1]), gk_max(v[2], v[3])), gk_max(gk_max(v[4], v[5]), gk_max(v[6], v[7])));
  bm = gk_max(bm, xor_lane<1>(bm));
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const int lane_of_block = (((int)m.p[2 + j] >> 4) - 1) * 2;             // (uniform)
    if (rl == lane_of_block) rp_lds[my][j] = inc ? bm : finite_or_big(fabs(bm - m.p[2 + S + j]));
  }
}

#endif  // !__HIPCC_RTC__

// stochastic Lotka-Volterra, Euler-Maruyama; rho = |mean/sd of prey and predator paths - obs|
template <int D, int S>
struct Sim<SABC_MODEL_LV, D, S> {
  static __device__ __forceinline__ void run(const ModelDesc &m, const double *th, uint64_t pid, uint64_t iter,
                                             double *rho, int coop = 0) {
    const int n_steps = (int)m.p[0];
    const double dt = m.p[1], sg = m.p[2];
    double X = m.p[3], Y = m.p[4];
    const double sdt = sg * sqrt(dt);
    NormalStream ns(m.seed, pid, PURPOSE_SIM, iter, coop);
    double SX = 0, QX = 0, SY = 0, QY = 0;
    ns.for_pairs(n_steps, [&](const double z1, const double z2) {
      // dX = (aX - bXY) dt + sigma X sqrt(dt) z1 = X ((a - bY) dt + sigma sqrt(dt) z1): the factored form (14 instead
      // of 21 operations per step), both species from the OLD state
      const double fx = fma(sdt, z1, fma(-th[1], Y, th[0]) * dt);
      const double fy = fma(sdt, z2, fma(th[1], X, -th[2]) * dt);
      X = fmax(fma(X, fx, X), 0.0);
      Y = fmax(fma(Y, fy, Y), 0.0);
      SX += X; QX += X * X; SY += Y; QY += Y * Y;
    });
    const double mX = SX / n_steps, mY = SY / n_steps;
    const double vX = (QX - SX * mX) / (n_steps - 1), vY = (QY - SY * mY) / (n_steps - 1);
    const double st[4] = {mX, sqrt(fmax(vX, 0.0)), mY, sqrt(fmax(vY, 0.0))};
#pragma unroll
    for (int j = 0; j < S; ++j) rho[j] = finite_or_big(fabs(st[j < 4 ? j : 3] - m.p[5 + j]));
  }
};

}  // namespace sabc
