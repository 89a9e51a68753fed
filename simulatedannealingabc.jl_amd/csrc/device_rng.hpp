// device_rng.hpp -- counter-based Philox4x32-10 per wavefront lane, 52-bit uniforms and
// Box-Muller pairs in f64 for gfx950.  Replaces every rand()/randn() site of the reference
// (SimulatedAnnealingABC.jl:163,174,324; proposals.jl:42,54,105-106,110,141,144).
//
// Stream layout (DESIGN.md "RNG streams"): key = seed; counter = (particle id, block index,
// iteration, purpose).  One block = 128 bits = two 52-bit uniforms = one Box-Muller pair.
// Keyed by GLOBAL particle id, so a run does not depend on how particles are sharded.
#pragma once
#if !defined(__HIPCC_RTC__)     // hipRTC pre-includes the HIP runtime
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#ifndef INFINITY                 // hipRTC has no <math.h>
#define INFINITY (__builtin_huge_val())
#endif

namespace sabc {

enum : uint32_t { PURPOSE_PRIOR = 0, PURPOSE_SIM = 1, PURPOSE_PROP = 2, PURPOSE_PROP2 = 3, PURPOSE_ACCEPT = 4,
                  PURPOSE_RESAMPLE = 5 };

struct u32x4 { uint32_t x, y, z, w; };

// a ^ b ^ c: one v_bitop3_b32 (truth table 0x96, new on gfx950) where the compiler in use knows the builtin -- hipcc of
// ROCm 7.x does; a hipRTC whose clang is older (the GPU box's run-time compiler once lacked a builtin hipcc had) falls
// back to two v_xor_b32.  Same bits either way (the Random123 known answers run through both, tests/test_user_simulator.py).
#if defined(__has_builtin)
#if __has_builtin(__builtin_amdgcn_bitop3_b32) && !defined(SABC_NO_BITOP3)
#define SABC_HAVE_BITOP3 1
#endif
#endif
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(SABC_HAVE_BITOP3)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                               uint32_t c3) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one v_mad_u64_u32 per product: hi and lo halves come from the same instruction
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    // three-input XOR in one instruction where available (xor3)
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ u32x4 stream_block(uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter,
                                              uint32_t k) {
  const uint32_t c3 = (purpose & 0xFFu) | ((uint32_t)(pid >> 32) << 8) | ((uint32_t)((iter >> 32) & 0xFFu) << 24);
  return philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)pid, k, (uint32_t)iter, c3);
}

// (x + 1/2) * 2^-52 with x = (low 20 bits of hi):(all of lo): exact in binary64, in (0,1).
// Built without a shift or an int->fp conversion: those 52 bits ARE the mantissa of a double in [1,2)
// (one v_and_or on the high word), and (1 - 2^-53) is subtracted, which is exact.
__device__ __forceinline__ double u52(uint32_t hi, uint32_t lo) {
  const double d = __hiloint2double((int)(0x3FF00000u | (hi & 0xFFFFFu)), (int)lo);
  return d - 0x1.fffffffffffffp-1;
}

__device__ __forceinline__ uint64_t pack64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }

// ---- f64 elementary functions specialised to the Box-Muller input ranges -------------------
// ocml's log / sincospi / sqrt are correctly-rounded-grade general routines built on double-double
// arithmetic (52 v_add_f64 per pair in the ISA).  The inputs here are known: u in [2^-53, 1), so
// no special values, no denormals, no range reduction beyond one table-free step.  Each routine
// below stays within ~1.5 ulp (tests/test_gpu_parity.py::test_normal_pairs_accuracy).

// q = a / b for finite, normal a, b of moderate magnitude: v_rcp_f64 + two Newton steps + one
// residual correction (what the compiler emits for '/', minus the scale/fixup for special values)
__device__ __forceinline__ double div_fast(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// sqrt(x) for x in [1e-300, 1e300]: v_rsq_f64 (~2^-27), one coupled Newton step, one residual correction
// (what the compiler emits for sqrt(), minus its second correction and the scaling for tiny / huge x)
__device__ __forceinline__ double sqrt_fast(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  return fma(fma(-g, g, x), h, g);
}

// log(x) for normal x > 0: x = 2^e m with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s),
// s = (m-1)/(m+1), degree-7 minimax polynomial in s^2 (the classic fdlibm scheme and constants)
__device__ __forceinline__ double log_fast(double x) {
  const uint64_t ix = (uint64_t)__double_as_longlong(x);
  int hx = (int)(ix >> 32);
  hx += 0x3ff00000 - 0x3fe6a09e;
  const int e = (hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffff) + 0x3fe6a09e;
  const double m = __longlong_as_double((long long)(((uint64_t)(uint32_t)hx << 32) | (ix & 0xffffffffull)));
  const double f = m - 1.0;
  const double hfsq = 0.5 * f * f;
  const double s = div_fast(f, 2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                   2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double dk = (double)e;
  return fma(dk, 6.93147180369123816490e-01, (fma(s, hfsq + R, dk * 1.90821492927058770002e-10) - hfsq) + f);
}

// ---- table-driven log and sin/cos for the Box-Muller pair ------------------------------------
// The pair costs one log and one sin/cos; with two small tables in LDS (2.5 KB, rng_tables.inc, generated by
// tools/gen_rng_tables.py) both shrink to short polynomials on a tiny argument: log 33 -> 19 VALU instructions
// (no division), sin/cos 34 -> 23 (no quadrant selects).  Every kernel that draws normals calls
// rng_tables_init() first (all threads of the workgroup, before any divergent return).
#include "rng_tables.inc"

struct RngTables {
  double2 logt[128];     // -2 x { 1/c_i rounded, -log of that }: c_i = 1 + i/128 (i < 53) or (1 + i/128)/2 (i >= 53); c_0 = 1
  double2 sct[33];       // { sin, cos } of 2 pi k / 32, k = 0..32 (row 32 = row 0)
  double exp2t[32];      // 2^(j/32)
};

__device__ __forceinline__ RngTables &rng_tables() {
  __shared__ RngTables t;
  return t;
}

// the copies only; the caller's next __syncthreads() publishes them (k_update shares that barrier with its ECDF index)
__device__ __forceinline__ void rng_tables_load() {
  RngTables &t = rng_tables();
  const int i = threadIdx.x;
  if (blockDim.x >= 128) {
    // every table entry has a thread of its own: the three reads go out together and the workgroup starts after ONE trip
    // to memory (three loops, each waiting for its own read, were three -- at the front of every workgroup's life)
    double2 a = make_double2(0.0, 0.0), b = make_double2(0.0, 0.0);
    double c = 0.0;
    if (i < 128) a = make_double2(kLogTab[i][0], kLogTab[i][1]);
    if (i < 33) b = make_double2(kSinCosTab[i & 31][0], kSinCosTab[i & 31][1]);
    if (i < 32) c = kExp2Tab[i];
    if (i < 128) t.logt[i] = a;
    if (i < 33) t.sct[i] = b;
    if (i < 32) t.exp2t[i] = c;
    return;
  }
  for (int k = i; k < 128; k += blockDim.x) t.logt[k] = make_double2(kLogTab[k][0], kLogTab[k][1]);
  for (int k = i; k < 33; k += blockDim.x) t.sct[k] = make_double2(kSinCosTab[k & 31][0], kSinCosTab[k & 31][1]);
  for (int k = i; k < 32; k += blockDim.x) t.exp2t[k] = kExp2Tab[k];
}

__device__ __forceinline__ void rng_tables_init() {
  rng_tables_load();
  __syncthreads();
}

// -2 log(x), normal x > 0 (the squared Box-Muller radius).  x = 2^E m; the 7 leading mantissa bits (rounded to
// nearest, so that 1 is a bin CENTRE) pick c_i with |m / c_i - 1| <= 2^-8; bins above sqrt(2) are taken as c_i / 2
// with E + 1, so m / c stays in [0.707, 1.414) and there is no cancellation against E ln 2 for x near 1 (x -> 1
// from below lands in bin 0 of E = 0: c = 1, full relative accuracy).  log(m) = logc_i + log1p(r), r = m inv_i - 1.
// With s = -2 r (one fma against the table's -2 inv_i): -2 log1p(r) = s + s^2/4 + s^3/12 + ... + s^7/448
// (remainder 2^-59 relative on |s| <= 2^-7).
__device__ __forceinline__ double neg2_log_tab(double x) {
  const uint32_t hi = (uint32_t)__double2hiint(x);
  const uint32_t t = hi + 0x800u;                       // round the mantissa to 7 bits (may carry into the exponent)
  const uint32_t tp = t + (75u << 13);                  // ... and carry when that rounded mantissa is >= 53/128
  const int nE = 1023 - (int)(tp >> 20);                // -E
  const double m = __hiloint2double((int)(hi + ((uint32_t)nE << 20)), __double2loint(x));   // x 2^-E, exact
  const double2 e = rng_tables().logt[(t >> 13) & 127u];
  const double s = fma(m, e.x, 2.0);                    // -2 (m inv - 1)
  double p = 1.0 / 448.0;
  p = fma(p, s, 1.0 / 192.0);
  p = fma(p, s, 1.0 / 80.0);
  p = fma(p, s, 1.0 / 32.0);
  p = fma(p, s, 1.0 / 12.0);
  p = fma(p, s, 0.25);
  const double l = fma(s * s, p, s);
  const double nEd = (double)nE;
  // the product and the polynomial have the same sign (both >= 0 for x < 1, both <= 0 for x > 1) and the table entry is
  // small, so one rounded product is accurate to about half an ulp of the sum: no hi/lo split of ln 2 needed
  return fma(nEd, 2.0 * 6.93147180559945309417e-01, e.y) + l;
}

// sin and cos of 2 pi u, u in (0,1): k = rint(32 u), f = 32 u - k exact, r = (pi/16) f, |r| <= pi/32;
// sin(a_k + r) = S + (S q + C sin r), cos(a_k + r) = C + (C q - S sin r) with q = cos r - 1 and (S, C) from the table.
__device__ __forceinline__ void sincos_2pi_tab(double u, double &sn, double &cs) {
  // k = rint(32 u) by the add-a-big-number trick: the sum's low mantissa word IS k (0..32; the table has 33 rows),
  // which saves the conversion and the wrap-around mask
  const double tm = fma(u, 32.0, 0x1.8p52);
  const double kf = tm - 0x1.8p52;
  const double r = 1.96349540849362077404e-01 * fma(u, 32.0, -kf);       // (pi / 16) (32 u - k), the difference is exact
  const double2 sc = rng_tables().sct[__double2loint(tm)];
  const double r2 = r * r;
  double p = 1.0 / 362880.0;
  p = fma(p, r2, -1.0 / 5040.0);
  p = fma(p, r2, 1.0 / 120.0);
  p = fma(p, r2, -1.0 / 6.0);
  const double sr = fma(r * r2, p, r);                                    // sin r
  double q = 1.0 / 40320.0;
  q = fma(q, r2, -1.0 / 720.0);
  q = fma(q, r2, 1.0 / 24.0);
  q = fma(q, r2, -0.5);
  q *= r2;                                                                // cos r - 1
  sn = sc.x + fma(sc.x, q, sc.y * sr);
  cs = sc.y + fma(sc.y, q, -(sc.x * sr));
}

// exp(x), |x| <= 700 (clamped): x = (32 e + j) ln2/32 + r with |r| <= ln2/64, exp(x) = 2^e 2^(j/32) exp(r); the integer
// 32 e + j by the add-a-big-number trick (two's complement in the sum's low word), 2^(j/32) from the table, exp(r) by
// its Taylor polynomial of degree 6 (remainder 4e-18).  15 VALU instructions against 21 for a table-free version with a degree-13 polynomial, and 7 polynomial
// constants instead of 13 (the g-and-k kernel, which calls it four times per draw pair, is short of scalar registers).
__device__ __forceinline__ double exp_tab(double x) {
  x = fmin(fmax(x, -700.0), 700.0);
  const double tm = fma(x, 0x1.71547652b82fep+5, 0x1.8p52);             // 32 / ln 2
  const double nf = tm - 0x1.8p52;
  const int n = __double2loint(tm);
  double r = fma(-nf, 0x1.62e42feep-6, x);                                // ln2/32, 32 significant bits: exact product
  r = fma(-nf, 0x1.a39ef35793c76p-38, r);
  const double t = rng_tables().exp2t[n & 31];
  double p = 1.0 / 720.0;
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(t * p, n >> 5);
}

// tanh(y) = sign(y) (1 - 2 / (exp(2|y|) + 1)): absolute error ~1e-16 (the relative error near 0 is not controlled,
// which is fine where it is used: inside 1 + c tanh(.))
__device__ __forceinline__ double tanh_abs_tab(double y) {
  const double a = fmin(fabs(y), 20.0);
  const double t = 1.0 - div_fast(2.0, exp_tab(2.0 * a) + 1.0);
  return copysign(t, y);
}

__device__ __forceinline__ void box_muller(const u32x4 w, double &z0, double &z1) {
  const double ua = u52(w.x, w.y);
  const double ub = u52(w.z, w.w);
  const double r = sqrt_fast(neg2_log_tab(ua));
  double sn, cs;
  sincos_2pi_tab(ub, sn, cs);
  z0 = r * cs;
  z1 = r * sn;
}

// A TEAM of W = 4 or 16 lanes (a quad | a row of the wave) that run one particle side by side: the value lane (team base + J)
// holds, on every lane of the team -- one DPP move per 32-bit half, no LDS (quad_perm [J, J, J, J] | row_newbcast:J).
template <int W, int J>
__device__ __forceinline__ double team_pick_ct(const double v) {
  static_assert((W == 4 || W == 16) && J >= 0 && J < W, "a quad or a row of 16 lanes");
  constexpr int ctrl = W == 4 ? J * 0x55 : 0x150 + J;
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// ... with the lane chosen at run time (uniform over the team): a branch per bit of j
template <int W, int J0 = 0, int N = W>
__device__ __forceinline__ double team_pick(const double v, const uint32_t j) {
  if constexpr (N == 1) {
    return team_pick_ct<W, J0>(v);
  } else {
    return (j & (uint32_t)(N / 2)) ? team_pick<W, J0 + N / 2, N / 2>(v, j) : team_pick<W, J0, N / 2>(v, j);
  }
}

// simulated pairs per loop trip of NormalStream::for_pairs with a lane per particle: independent Philox / Box-Muller chains for
// the scheduler (k_update<1,1,1,0>: 92 -> 62 VGPRs, 246 -> ~240 us at n = 1e6); the draws are still handed out in stream order
#ifndef SABC_SIM_UNROLL
#define SABC_SIM_UNROLL 2
#endif

template <int... J> struct lane_seq {};
template <int N, int... J> struct make_lane_seq : make_lane_seq<N - 1, N - 1, J...> {};
template <int... J> struct make_lane_seq<0, J...> { using type = lane_seq<J...>; };

// Sequential N(0,1) stream of one (particle, purpose, iteration): block k yields normals 2k, 2k+1.
//
// coop = 4 | 16 (k_update_persistent on small shards, persistent_kernel.hpp): the lanes of a TEAM -- a quad, or a row of 16 --
// run the same particle -- the same code on the same data -- and share the generator: each lane turns ITS blocks into pairs
// (the 99 of a simulator's ~105 instructions per pair that are Philox + Box-Muller), and the team consumes the pairs in stream
// order through DPP broadcasts.  The stream a simulator sees is the same stream, pair for pair and bit for bit (counter-based:
// block k is block k whoever computes it); a particle's chain of draws is a quarter | a sixteenth as long.  (Worth it only
// while the device is under-filled: the team issues more instructions per particle than one lane.)  All lanes of a team must
// make the same requests -- they do: their control flow depends on the particle's data only.
//
// for_pairs(n, f) -- f(z0, z1) for the next n pairs of the stream, in stream order -- is the loop a simulator should draw its
// bulk with.  A lane per particle: the plain loop over pair(), two trips unrolled.  A team per particle: 4 W pairs at a time,
// FOUR blocks per lane side by side (a single block per lane is a chain of ~100 dependent instructions: with one wave on the
// SIMD its latency, not its issue, is what the team would wait for), handed out through DPP broadcasts whose lane is a
// compile-time constant; the rest -- fewer than 4 W pairs -- in one more group of as many blocks per lane as it takes.  pair() / uniform_pair() / next() keep working in either mode, one
// group of W at a time, the lane picked by branches on the stream position.
struct NormalStream {
  uint64_t seed, pid, iter;
  uint32_t purpose, k;
  double spare;
  bool have;
  int coop;                      // 0: every lane its own stream | 4, 16: the lanes of a team share one (see above)
  int buf_block;                 // coop: first block of the group of W this lane holds one block of (-1: none) ...
  bool buf_uniform;              // ... as uniforms (uniform_pair) or as a Box-Muller pair
  double b0, b1;
  __device__ __forceinline__ NormalStream(uint64_t seed_, uint64_t pid_, uint32_t purpose_, uint64_t iter_, int coop_ = 0)
      : seed(seed_), pid(pid_), iter(iter_), purpose(purpose_), k(0), spare(0.0), have(false), coop(coop_), buf_block(-1),
        buf_uniform(false), b0(0.0), b1(0.0) {}
  __device__ __forceinline__ void pair(double &z0, double &z1) {  // consumes one whole block
    if (coop == 4) { fetch<4>(k++, false, z0, z1); return; }
    if (coop == 16) { fetch<16>(k++, false, z0, z1); return; }
    box_muller(stream_block(seed, pid, purpose, iter, k++), z0, z1);
  }
  // two U(0,1) draws from one whole block (the 52-bit uniforms the Box-Muller pair would have been made of)
  __device__ __forceinline__ void uniform_pair(double &u0, double &u1) {
    if (coop == 4) { fetch<4>(k++, true, u0, u1); return; }
    if (coop == 16) { fetch<16>(k++, true, u0, u1); return; }
    const u32x4 w = stream_block(seed, pid, purpose, iter, k++);
    u0 = u52(w.x, w.y);
    u1 = u52(w.z, w.w);
  }
  template <class F>
  __device__ __forceinline__ void for_pairs(const int n, F &&f) {
    if (coop == 4) { for_pairs_team<4>(n, f); return; }
    if (coop == 16) { for_pairs_team<16>(n, f); return; }
#pragma unroll SABC_SIM_UNROLL
    for (int i = 0; i < n; ++i) {
      double z0, z1;
      pair(z0, z1);
      f(z0, z1);
    }
  }
  __device__ __forceinline__ double next() {
    if (have) { have = false; return spare; }
    double z0;
    pair(z0, spare);
    have = true;
    return z0;
  }

 private:
  template <int W, class F, int... J>
  __device__ __forceinline__ void hand_out(const double g0, const double g1, F &f, lane_seq<J...>) {
    (f(team_pick_ct<W, J>(g0), team_pick_ct<W, J>(g1)), ...);
  }
  template <int W, class F, int... J>
  __device__ __forceinline__ void hand_out_first(const int m, const double g0, const double g1, F &f, lane_seq<J...>) {   // the first m < W pairs
    ((J < m ? f(team_pick_ct<W, J>(g0), team_pick_ct<W, J>(g1)) : (void)0), ...);      // (m = W: all of them)
  }
  template <int W, int NB, class F>
  __device__ __forceinline__ void tail(const int c, const int r, F &f) {      // (NB - 1) W < r <= NB W pairs from block k + c on
    using lanes = typename make_lane_seq<W>::type;
    const uint32_t q = threadIdx.x & (uint32_t)(W - 1);
    double g0[NB], g1[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) box_muller(stream_block(seed, pid, purpose, iter, k + (uint32_t)(c + W * a) + q), g0[a], g1[a]);
#pragma unroll
    for (int a = 0; a < NB - 1; ++a) hand_out<W>(g0[a], g1[a], f, lanes{});
    hand_out_first<W>(r - (NB - 1) * W, g0[NB - 1], g1[NB - 1], f, lanes{});
  }
  template <int W, class F>
  __device__ __forceinline__ void for_pairs_team(const int n, F &f) {
    using lanes = typename make_lane_seq<W>::type;
    const uint32_t q = threadIdx.x & (uint32_t)(W - 1);
    int c = 0;
    for (; c + 4 * W <= n; c += 4 * W) {               // whole groups of 4 W: blocks k + c + W a + q, a = 0..3, on lane q
      double g0[4], g1[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) box_muller(stream_block(seed, pid, purpose, iter, k + (uint32_t)(c + W * a) + q), g0[a], g1[a]);
      hand_out<W>(g0[0], g1[0], f, lanes{});
      hand_out<W>(g0[1], g1[1], f, lanes{});
      hand_out<W>(g0[2], g1[2], f, lanes{});
      hand_out<W>(g0[3], g1[3], f, lanes{});
    }
    const int r = n - c;                               // the rest, < 4 W pairs: ceil(r / W) blocks per lane, again side by side
    if (r > 3 * W) tail<W, 4>(c, r, f);
    else if (r > 2 * W) tail<W, 3>(c, r, f);
    else if (r > W) tail<W, 2>(c, r, f);
    else if (r > 0) tail<W, 1>(c, r, f);
    k += (uint32_t)n;
  }
  // coop: block kk of the stream, from the lane of the team that holds it; a group of W blocks is (re)generated -- one block
  // per lane -- when the request leaves the buffered group or asks for the other kind (a simulator that mixes pair() and
  // uniform_pair() inside a group pays a refill for it, the stream it sees is still the stream)
  template <int W>
  __device__ __forceinline__ void fetch(uint32_t kk, bool uniform, double &x0, double &x1) {
    const int base = (int)(kk & ~(uint32_t)(W - 1));
    if (buf_block != base || buf_uniform != uniform) {
      const u32x4 w = stream_block(seed, pid, purpose, iter, (uint32_t)base + (threadIdx.x & (uint32_t)(W - 1)));
      if (uniform) { b0 = u52(w.x, w.y); b1 = u52(w.z, w.w); }
      else box_muller(w, b0, b1);
      buf_block = base;
      buf_uniform = uniform;
    }
    x0 = team_pick<W>(b0, kk);
    x1 = team_pick<W>(b1, kk);
  }
};

}  // namespace sabc
