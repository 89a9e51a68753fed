// rtc.hpp -- a simulator supplied as HIP source (SABC_MODEL_USER): compiled at run time with hipRTC into the same
// fused update kernel the built-in simulators use (update_kernel.hpp), loaded as a code-object module.
#pragma once
#include <hip/hip_runtime_api.h>
#include <string>

namespace sabc {

struct RtcKernels {
  hipModule_t module = nullptr;
  hipFunction_t prior_simulate = nullptr;      // k_prior_simulate<USER, D, S>
  hipFunction_t update[3] = {nullptr, nullptr, nullptr};   // k_update<USER, D, S, PROP>
  hipFunction_t simulate_batch = nullptr;      // k_simulate_batch<USER, D, S>
  hipFunction_t stats = nullptr;               // k_stats<D, S>
  hipFunction_t prior_op = nullptr;            // k_prior_op_t<D>
  hipFunction_t persistent[3] = {nullptr, nullptr, nullptr};   // k_update_persistent<USER, D, S, PROP> (persistent_kernel.hpp)
  hipFunction_t persistent4[3] = {nullptr, nullptr, nullptr};  // ... <USER, D, S, PROP, 4>: a particle per quad of lanes
  hipFunction_t persistent16[3] = {nullptr, nullptr, nullptr}; // ... <USER, D, S, PROP, 16>: a particle per row of 16 lanes
  int d = 0, s = 0;
};

// Compiles `user_source` (it must define
//     __device__ void sabc_user_simulate(const double *theta, const double *params, sabc::NormalStream &rng, double *rho_out);
// ) for gfx950 and loads the kernels for (d, s).  `csrc_dir` holds update_kernel.hpp and what it includes.
// Returns 0 or -1 with the compiler log / error text in `log`.
// user_prior: the source also defines sabc_user_prior_sample / sabc_user_prior_logpdf (sabc_config::prior_joint = 3).
// with_persistent: also compile k_update_persistent<USER, D, S, PROP> (small shards: the updates of a call in one launch;
// it roughly doubles the compile time, so only handles that can take the form ask for it)
int rtc_build(const char *user_source, int d, int s, const std::string &csrc_dir, RtcKernels *out, std::string *log,
              bool user_prior = false, bool with_persistent = false);
// the same with out == nullptr stopping after the compiler (no device needed): a syntax / interface check of a source
int rtc_compile(const char *user_source, int d, int s, const std::string &csrc_dir, RtcKernels *out, std::string *log,
                size_t *code_size, bool user_prior = false, bool with_persistent = false);
void rtc_release(RtcKernels *k);
// directory of this shared library + "/csrc" (the headers ship next to the library)
std::string rtc_default_csrc_dir();

}  // namespace sabc
