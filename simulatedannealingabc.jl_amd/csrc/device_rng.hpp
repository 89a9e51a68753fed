// device_rng.hpp -- counter-based Philox4x32-10 per wavefront lane, 52-bit uniforms and
// Box-Muller pairs in f64 for gfx950.  Replaces every rand()/randn() site of the reference
// (SimulatedAnnealingABC.jl:163,174,324; proposals.jl:42,54,105-106,110,141,144).
//
// Stream layout (DESIGN.md "RNG streams"): key = seed; counter = (particle id, block index,
// iteration, purpose).  One block = 128 bits = two 52-bit uniforms = one Box-Muller pair.
// Keyed by GLOBAL particle id, so a run does not depend on how particles are sharded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sabc {

enum : uint32_t { PURPOSE_PRIOR = 0, PURPOSE_SIM = 1, PURPOSE_PROP = 2, PURPOSE_PROP2 = 3, PURPOSE_ACCEPT = 4,
                  PURPOSE_RESAMPLE = 5 };

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                               uint32_t c3) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one v_mad_u64_u32 per product: hi and lo halves come from the same instruction
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
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

// (x + 1/2) * 2^-52 with x the top 52 bits: exact in binary64, in (0,1)
__device__ __forceinline__ double u52(uint32_t hi, uint32_t lo) {
  const uint64_t x = (((uint64_t)hi << 32) | lo) >> 12;
  return ((double)x + 0.5) * 0x1.0p-52;
}

__device__ __forceinline__ uint64_t pack64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }

__device__ __forceinline__ void box_muller(const u32x4 w, double &z0, double &z1) {
  const double ua = u52(w.x, w.y);
  const double ub = u52(w.z, w.w);
  const double r = sqrt(-2.0 * log(ua));
  double sn, cs;
  sincospi(2.0 * ub, &sn, &cs);   // angle 2*pi*ub without a range reduction
  z0 = r * cs;
  z1 = r * sn;
}

// Sequential N(0,1) stream of one (particle, purpose, iteration): block k yields normals 2k, 2k+1.
struct NormalStream {
  uint64_t seed, pid, iter;
  uint32_t purpose, k;
  double spare;
  bool have;
  __device__ __forceinline__ NormalStream(uint64_t seed_, uint64_t pid_, uint32_t purpose_, uint64_t iter_)
      : seed(seed_), pid(pid_), iter(iter_), purpose(purpose_), k(0), spare(0.0), have(false) {}
  __device__ __forceinline__ void pair(double &z0, double &z1) {  // consumes one whole block
    box_muller(stream_block(seed, pid, purpose, iter, k++), z0, z1);
  }
  __device__ __forceinline__ double next() {
    if (have) { have = false; return spare; }
    double z0;
    pair(z0, spare);
    have = true;
    return z0;
  }
};

}  // namespace sabc
