// rtc.cpp -- see rtc.hpp.  hipRTC is bound with dlopen at first use: a host without libhiprtc can still load the
// library and run the built-in simulators.
#include "rtc.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/sabc_hip.h"
#include "kernels.hpp"

namespace sabc {

namespace {

struct HiprtcApi {
  void *lib = nullptr;
  int (*CreateProgram)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*DestroyProgram)(void **) = nullptr;
  int (*CompileProgram)(void *, int, const char **) = nullptr;
  int (*GetProgramLogSize)(void *, size_t *) = nullptr;
  int (*GetProgramLog)(void *, char *) = nullptr;
  int (*GetCodeSize)(void *, size_t *) = nullptr;
  int (*GetCode)(void *, char *) = nullptr;
  int (*AddNameExpression)(void *, const char *) = nullptr;
  int (*GetLoweredName)(void *, const char *, const char **) = nullptr;
};

HiprtcApi *hiprtc_api() {
  static HiprtcApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    for (const char *n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
#define SABC_RTC_SYM(field, sym) api.field = (decltype(api.field))dlsym(api.lib, sym)
      SABC_RTC_SYM(CreateProgram, "hiprtcCreateProgram");
      SABC_RTC_SYM(DestroyProgram, "hiprtcDestroyProgram");
      SABC_RTC_SYM(CompileProgram, "hiprtcCompileProgram");
      SABC_RTC_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
      SABC_RTC_SYM(GetProgramLog, "hiprtcGetProgramLog");
      SABC_RTC_SYM(GetCodeSize, "hiprtcGetCodeSize");
      SABC_RTC_SYM(GetCode, "hiprtcGetCode");
      SABC_RTC_SYM(AddNameExpression, "hiprtcAddNameExpression");
      SABC_RTC_SYM(GetLoweredName, "hiprtcGetLoweredName");
#undef SABC_RTC_SYM
    }
  });
  const bool ok = api.lib && api.CreateProgram && api.DestroyProgram && api.CompileProgram && api.GetProgramLogSize &&
                  api.GetProgramLog && api.GetCodeSize && api.GetCode && api.AddNameExpression && api.GetLoweredName;
  return ok ? &api : nullptr;
}

void anchor() {}

// compiled code objects of this process, keyed by (source, d, s): a second handle with the same simulator (every
// sabc() call makes one) loads the module without paying the ~2 s of compilation again
struct CachedModule {
  std::vector<char> code;
  std::string lowered[10];
};
std::mutex g_cache_mutex;
std::map<std::string, CachedModule> g_cache;

}  // namespace

std::string rtc_default_csrc_dir() {
  Dl_info info;
  if (dladdr((void *)&anchor, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t slash = p.rfind('/');
    return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/csrc";
  }
  return "csrc";
}

void rtc_release(RtcKernels *k) {
  if (k && k->module) (void)hipModuleUnload(k->module);
  if (k) *k = RtcKernels();
}

int rtc_build(const char *user_source, int d, int s, const std::string &csrc_dir, RtcKernels *out, std::string *log,
              bool user_prior, bool with_persistent) {
  return rtc_compile(user_source, d, s, csrc_dir, out, log, nullptr, user_prior, with_persistent);
}

// out == nullptr: compile only (needs no device); code_size (optional) receives the size of the code object
int rtc_compile(const char *user_source, int d, int s, const std::string &csrc_dir, RtcKernels *out, std::string *log,
                size_t *code_size, bool user_prior, bool with_persistent) {
  constexpr int kMaxKernels = 10;
  const int kKernels = with_persistent ? 10 : 7;
  HiprtcApi *api = hiprtc_api();
  if (!api) { *log = "libhiprtc.so could not be loaded: simulators from source need the hipRTC of ROCm"; return -1; }
  if (d < 1 || d > SABC_MAX_PARA || s < 1 || s > SABC_MAX_SOURCE_STATS) { *log = "n_para / n_stats out of range (a simulator from source: d <= 16, s <= 16)"; return -1; }
  char tail[2048];
  std::snprintf(tail, sizeof(tail),
                "\nnamespace sabc {\n"
                "template <int D, int S>\n"
                "struct Sim<SABC_MODEL_USER, D, S> {\n"
                "  static __device__ __forceinline__ void run(const ModelDesc &m, const double *th, uint64_t pid, uint64_t iter,\n"
                "                                             double *rho) {\n"
                "    NormalStream ns(m.seed, pid, PURPOSE_SIM, iter);\n"
                "    ::sabc_user_simulate(th, m.p, ns, rho);\n"
                "  }\n"
                "};\n"
                "}  // namespace sabc\n");
  std::string src = std::string("#include \"update_kernel.hpp\"\n") + (with_persistent ? "#include \"persistent_kernel.hpp\"\n" : "") + "#line 1 \"f_dist.hip\"\n";
  src += user_source;
  src += tail;

  char name[10][112];   // kKernels <= 10
  std::snprintf(name[0], sizeof(name[0]), "sabc::k_prior_simulate<%d, %d, %d>", SABC_MODEL_USER, d, s);
  for (int p = 0; p < 3; ++p) std::snprintf(name[1 + p], sizeof(name[1 + p]), "sabc::k_update<%d, %d, %d, %d>", SABC_MODEL_USER, d, s, p);
  std::snprintf(name[4], sizeof(name[4]), "sabc::k_simulate_batch<%d, %d, %d>", SABC_MODEL_USER, d, s);
  std::snprintf(name[5], sizeof(name[5]), "sabc::k_stats<%d, %d>", d, s);
  std::snprintf(name[6], sizeof(name[6]), "sabc::k_prior_op_t<%d>", d);
  for (int p = 0; p < 3; ++p) std::snprintf(name[7 + p], sizeof(name[7 + p]), "sabc::k_update_persistent<%d, %d, %d, %d>", SABC_MODEL_USER, d, s, p);
  // extra compiler flags (e.g. -DSABC_NO_BITOP3: the two-instruction form of the Philox round's three-input XOR)
  const char *extra_env = std::getenv("SABC_RTC_EXTRA_FLAGS");
  const std::string extra = extra_env ? extra_env : "";

  const std::string cache_key = std::to_string(d) + "," + std::to_string(s) + (user_prior ? ",P" : "") + (with_persistent ? ",1L" : "") + "," + extra + "\n" + user_source;
  if (out) {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto hit = g_cache.find(cache_key);
    if (hit != g_cache.end()) {
      RtcKernels k;
      k.d = d; k.s = s;
      if (hipModuleLoadData(&k.module, hit->second.code.data()) != hipSuccess) { *log = "hipModuleLoadData of the cached simulator failed"; return -1; }
      hipFunction_t *slots[kMaxKernels] = {&k.prior_simulate, &k.update[0], &k.update[1], &k.update[2], &k.simulate_batch, &k.stats, &k.prior_op,
                                        &k.persistent[0], &k.persistent[1], &k.persistent[2]};
      for (int i = 0; i < kKernels; ++i)
        if (hipModuleGetFunction(slots[i], k.module, hit->second.lowered[i].c_str()) != hipSuccess) {
          *log = std::string("kernel not found in the cached module: ") + name[i];
          rtc_release(&k);
          return -1;
        }
      if (code_size) *code_size = hit->second.code.size();
      *out = k;
      return 0;
    }
  }
  void *prog = nullptr;
  if (api->CreateProgram(&prog, src.c_str(), "sabc_user_simulator.hip", 0, nullptr, nullptr)) { *log = "hiprtcCreateProgram failed"; return -1; }
  for (int i = 0; i < kKernels; ++i) api->AddNameExpression(prog, name[i]);
  const char *rocm = std::getenv("ROCM_PATH");
  const std::string inc_rocm = std::string("-I") + (rocm && *rocm ? rocm : "/opt/rocm") + "/include";
  const std::string inc_csrc = "-I" + csrc_dir;
  // the launch geometry the host library was built with (launch_update, build_coarse) must be the one the kernels are
  // compiled for: a variant library (tools/build_variants.sh) forwards its -D overrides to the run-time compiler
#define SABC_RTC_STR2(x) #x
#define SABC_RTC_STR(x) SABC_RTC_STR2(x)
  std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", inc_csrc.c_str(), inc_rocm.c_str(),
                        "-DSABC_UPDATE_BLOCK=" SABC_RTC_STR(SABC_UPDATE_BLOCK), "-DSABC_UPDATE_BLOCK_MS=" SABC_RTC_STR(SABC_UPDATE_BLOCK_MS),
                        "-DSABC_CDF_COARSE=" SABC_RTC_STR(SABC_CDF_COARSE), "-DSABC_CDF_COARSE_MS=" SABC_RTC_STR(SABC_CDF_COARSE_MS),
                        "-DSABC_UPDATE_MIN_WAVES=" SABC_RTC_STR(SABC_UPDATE_MIN_WAVES),
                        "-DSABC_UPDATE_MIN_WAVES_MS=" SABC_RTC_STR(SABC_UPDATE_MIN_WAVES_MS)};
#undef SABC_RTC_STR
#undef SABC_RTC_STR2
  if (user_prior) opts.push_back("-DSABC_USER_PRIOR=1");
  std::vector<std::string> extra_words;
  for (size_t i = 0; i < extra.size();) {
    while (i < extra.size() && extra[i] == ' ') ++i;
    size_t j = i;
    while (j < extra.size() && extra[j] != ' ') ++j;
    if (j > i) extra_words.push_back(extra.substr(i, j - i));
    i = j;
  }
  for (const std::string &w : extra_words) opts.push_back(w.c_str());
  const int rc = api->CompileProgram(prog, (int)opts.size(), opts.data());
  size_t ls = 0;
  api->GetProgramLogSize(prog, &ls);
  if (ls > 1) { log->assign(ls, '\0'); api->GetProgramLog(prog, &(*log)[0]); }
  if (rc) {
    if (log->empty()) *log = "hiprtcCompileProgram failed";
    api->DestroyProgram(&prog);
    return -1;
  }
  size_t cs = 0;
  api->GetCodeSize(prog, &cs);
  if (code_size) *code_size = cs;
  if (!out) {                                         // compile-only check: every kernel must be there by name
    for (int i = 0; i < kKernels; ++i) {
      const char *lowered = nullptr;
      if (api->GetLoweredName(prog, name[i], &lowered) || !lowered) {
        *log = std::string("kernel missing from the compiled module: ") + name[i];
        api->DestroyProgram(&prog);
        return -1;
      }
    }
    api->DestroyProgram(&prog);
    return 0;
  }
  std::vector<char> code(cs);
  api->GetCode(prog, code.data());
  RtcKernels k;
  k.d = d; k.s = s;
  if (hipModuleLoadData(&k.module, code.data()) != hipSuccess) {
    *log = "hipModuleLoadData of the compiled simulator failed";
    api->DestroyProgram(&prog);
    return -1;
  }
  hipFunction_t *slots[kMaxKernels] = {&k.prior_simulate, &k.update[0], &k.update[1], &k.update[2], &k.simulate_batch, &k.stats, &k.prior_op,
                                        &k.persistent[0], &k.persistent[1], &k.persistent[2]};
  CachedModule entry;
  for (int i = 0; i < kKernels; ++i) {
    const char *lowered = nullptr;
    if (api->GetLoweredName(prog, name[i], &lowered) || !lowered ||
        hipModuleGetFunction(slots[i], k.module, lowered) != hipSuccess) {
      *log = std::string("kernel not found in the compiled module: ") + name[i];
      api->DestroyProgram(&prog);
      rtc_release(&k);
      return -1;
    }
    entry.lowered[i] = lowered;
  }
  api->DestroyProgram(&prog);
  entry.code = std::move(code);
  {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    g_cache.emplace(cache_key, std::move(entry));
  }
  *out = k;
  return 0;
}

}  // namespace sabc
