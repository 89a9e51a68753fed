// sort.hip -- ascending sort of one distance column (cdf_estimators.jl:33 `sort(x)`), once per
// statistic at initialization.  A device radix sort is a commodity primitive: rocPRIM's is used
// (header-only, ships with ROCm); the result of a sort is unique, so it does not affect parity.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "kernels.hpp"

namespace sabc {

int sort_f64(const double *in, double *out, int64_t n, void *tmp, size_t *tmp_bytes, hipStream_t stream) {
  size_t bytes = tmp ? *tmp_bytes : 0;
  hipError_t e = rocprim::radix_sort_keys(tmp, bytes, in, out, (size_t)n, 0, 64, stream);
  if (!tmp) *tmp_bytes = bytes;
  return (int)e;
}

}  // namespace sabc
