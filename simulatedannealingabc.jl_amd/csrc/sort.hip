// sort.hip -- ascending sort of one distance column (cdf_estimators.jl:33 `sort(x)`), once per statistic at
// initialization: a hand-written least-significant-digit radix sort for gfx950 (8 bits per pass, 8 passes over the
// order-preserving 64-bit image of a double).  Per pass four launches:
//   k_radix_hist    every workgroup counts the digits of its tile of 1024 keys (LDS atomics) -> hist[digit][tile]
//   k_radix_rowsum, k_radix_rowscan   exclusive prefix over hist in (digit, tile) order: where each tile's keys of each
//                   digit go (one workgroup per digit, coalesced)
//   k_radix_scatter every workgroup ranks its keys STABLY (wave-level match by ballots, waves and quarter-tiles in order)
//                   and writes them to their places
// The first pass reads doubles and maps them to keys, the last maps back.  The result of a sort is unique, so nothing here
// is part of the specification shared with the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"

namespace sabc {

namespace {

constexpr int kSortBlock = 256;                 // 4 wavefronts
constexpr int kSortPerThread = 4;
constexpr int kSortTile = kSortBlock * kSortPerThread;
constexpr int kRadix = 256;

// order-preserving image: negative doubles (sign bit set) are complemented, the others get the sign bit set
// (-inf < ... < -0.0 < +0.0 < ... < +inf < NaN with a clear sign bit)
__device__ __forceinline__ uint64_t key_of(double x) {
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(uint64_t k) {
  const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

template <bool FIRST>
__device__ __forceinline__ uint64_t load_key(const void *src, int64_t i) {
  if (FIRST) return key_of(static_cast<const double *>(src)[i]);
  return static_cast<const uint64_t *>(src)[i];
}

template <bool FIRST>
__global__ void __launch_bounds__(kSortBlock)
k_radix_hist(const void *__restrict__ src, const int64_t n, const int shift, const int64_t n_tiles, uint64_t *__restrict__ hist) {
  __shared__ unsigned cnt[kRadix];
  cnt[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int e = 0; e < kSortPerThread; ++e) {
    const int64_t i = base + threadIdx.x + (int64_t)e * kSortBlock;
    if (i < n) atomicAdd(&cnt[(unsigned)(load_key<FIRST>(src, i) >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * n_tiles + blockIdx.x] = cnt[threadIdx.x];
}

// Exclusive prefix over hist in (digit, tile) order, in two launches of 256 workgroups (one per digit, coalesced row
// accesses): the row totals, then every row scanned in place on top of the total of the rows before it.
__global__ void __launch_bounds__(kSortBlock)
k_radix_rowsum(const uint64_t *__restrict__ hist, const int64_t n_tiles, uint64_t *__restrict__ totals) {
  __shared__ uint64_t sm[kSortBlock / 64];
  const uint64_t *row = hist + (int64_t)blockIdx.x * n_tiles;
  uint64_t s = 0;
  for (int64_t i = threadIdx.x; i < n_tiles; i += kSortBlock) s += row[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) totals[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void __launch_bounds__(kSortBlock)
k_radix_rowscan(uint64_t *__restrict__ hist, const int64_t n_tiles, const uint64_t *__restrict__ totals) {
  __shared__ uint64_t part[2][kSortBlock];
  __shared__ uint64_t carry;
  const int t = threadIdx.x;
  // where this digit starts: the totals of the digits before it (tree sum over the 256 totals, masked)
  part[0][t] = t < (int)blockIdx.x ? totals[t] : 0;
  __syncthreads();
  for (int off = kSortBlock / 2; off > 0; off >>= 1) {
    if (t < off) part[0][t] += part[0][t + off];
    __syncthreads();
  }
  if (t == 0) carry = part[0][0];
  __syncthreads();
  uint64_t *row = hist + (int64_t)blockIdx.x * n_tiles;
  for (int64_t base = 0; base < n_tiles; base += kSortBlock) {     // 256 counters per trip: inclusive scan in LDS
    const int64_t i = base + t;
    const uint64_t v = i < n_tiles ? row[i] : 0;
    part[0][t] = v;
    __syncthreads();
    int cur = 0;
    for (int off = 1; off < kSortBlock; off <<= 1) {
      part[1 - cur][t] = t >= off ? part[cur][t] + part[cur][t - off] : part[cur][t];
      cur = 1 - cur;
      __syncthreads();
    }
    if (i < n_tiles) row[i] = carry + part[cur][t] - v;            // exclusive
    __syncthreads();
    if (t == kSortBlock - 1) carry += part[cur][t];
    __syncthreads();
  }
}

template <bool FIRST, bool LAST>
__global__ void __launch_bounds__(kSortBlock)
k_radix_scatter(const void *__restrict__ src, void *__restrict__ dst, const int64_t n, const int shift, const int64_t n_tiles,
                const uint64_t *__restrict__ hist) {
  __shared__ uint64_t gbase[kRadix];             // where this tile's keys of each digit start in dst
  __shared__ unsigned run[kRadix];               // keys of each digit already placed by earlier quarter-tiles
  __shared__ unsigned cnt[kSortBlock / 64][kRadix];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  gbase[threadIdx.x] = hist[(int64_t)threadIdx.x * n_tiles + blockIdx.x];
  run[threadIdx.x] = 0u;
#pragma unroll
  for (int w = 0; w < kSortBlock / 64; ++w) cnt[w][threadIdx.x] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  // quarter-tile e holds keys base + 256 e .. + 255 in thread order: (e, wave, lane) is the stable order
  for (int e = 0; e < kSortPerThread; ++e) {
    const int64_t i = base + threadIdx.x + (int64_t)e * kSortBlock;
    const bool live = i < n;
    const uint64_t key = live ? load_key<FIRST>(src, i) : 0ull;
    const unsigned d = (unsigned)(key >> shift) & 255u;
    uint64_t peers = __ballot(live);             // lanes of this wave holding the same digit
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned rank = (unsigned)__popcll(peers & below);
    if (live && rank == 0u) cnt[wave][d] = (unsigned)__popcll(peers);
    __syncthreads();
    if (live) {
      unsigned off = run[d] + rank;
      for (int w = 0; w < wave; ++w) off += cnt[w][d];
      const int64_t pos = (int64_t)(gbase[d] + off);
      if (LAST) static_cast<double *>(dst)[pos] = value_of(key);
      else static_cast<uint64_t *>(dst)[pos] = key;
    }
    __syncthreads();
    {                                            // thread t closes digit t for this quarter-tile
      unsigned tot = 0u;
#pragma unroll
      for (int w = 0; w < kSortBlock / 64; ++w) { tot += cnt[w][threadIdx.x]; cnt[w][threadIdx.x] = 0u; }
      run[threadIdx.x] += tot;
    }
    __syncthreads();
  }
}

}  // namespace

// ascending sort of n doubles.  tmp == nullptr: *tmp_bytes receives the scratch size (digit counters + one key buffer).
int sort_f64(const double *in, double *out, int64_t n, void *tmp, size_t *tmp_bytes, hipStream_t stream) {
  const int64_t n_tiles = n > 0 ? (n + kSortTile - 1) / kSortTile : 1;
  const size_t hist_bytes = ((size_t)kRadix * (size_t)n_tiles + kRadix) * sizeof(uint64_t);   // counters + row totals
  const size_t need = hist_bytes + (size_t)(n > 0 ? n : 1) * sizeof(uint64_t);
  if (!tmp) { *tmp_bytes = need; return 0; }
  if (*tmp_bytes < need) return (int)hipErrorInvalidValue;
  if (n <= 0) return 0;
  uint64_t *hist = static_cast<uint64_t *>(tmp);
  void *alt = static_cast<char *>(tmp) + hist_bytes;
  const dim3 grid((unsigned)n_tiles), block(kSortBlock);
  const void *src = in;
  for (int pass = 0; pass < 8; ++pass) {
    void *dst = (pass & 1) ? static_cast<void *>(out) : alt;      // pass 0 -> alt, 1 -> out, ..., 7 -> out
    const int shift = 8 * pass;
    if (pass == 0) hipLaunchKernelGGL(k_radix_hist<true>, grid, block, 0, stream, src, n, shift, n_tiles, hist);
    else hipLaunchKernelGGL(k_radix_hist<false>, grid, block, 0, stream, src, n, shift, n_tiles, hist);
    uint64_t *totals = hist + (size_t)kRadix * (size_t)n_tiles;
    hipLaunchKernelGGL(k_radix_rowsum, dim3(kRadix), block, 0, stream, (const uint64_t *)hist, n_tiles, totals);
    hipLaunchKernelGGL(k_radix_rowscan, dim3(kRadix), block, 0, stream, hist, n_tiles, (const uint64_t *)totals);
    if (pass == 0) hipLaunchKernelGGL((k_radix_scatter<true, false>), grid, block, 0, stream, src, dst, n, shift, n_tiles, (const uint64_t *)hist);
    else if (pass == 7) hipLaunchKernelGGL((k_radix_scatter<false, true>), grid, block, 0, stream, src, dst, n, shift, n_tiles, (const uint64_t *)hist);
    else hipLaunchKernelGGL((k_radix_scatter<false, false>), grid, block, 0, stream, src, dst, n, shift, n_tiles, (const uint64_t *)hist);
    src = dst;
  }
  return (int)hipGetLastError();
}

}  // namespace sabc
