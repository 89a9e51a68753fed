// rtc.cpp -- see rtc.hpp.  hipRTC is bound with dlopen at first use: a host without libhiprtc can still load the
// library and run the built-in simulators.
#include "rtc.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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
  std::string lowered[16];
};
std::mutex g_cache_mutex;
std::map<std::string, CachedModule> g_cache;

// ---- the same across processes: code objects on disk ----
// Compiling a simulator costs 2-5 s (the fused kernels with the user's code inlined, x 3 proposals, + the one-launch form of
// small shards) -- every sabc() of every process paid it again.  The code object and the kernels' lowered names are kept
// under $SABC_RTC_CACHE_DIR | $XDG_CACHE_HOME/sabc_hip | ~/.cache/sabc_hip, keyed by a hash of EVERYTHING that goes into the
// compilation: the in-process cache key (shape, flags, the user's source), the text of every header of csrc/ the unit
// includes (a rebuilt or different library never finds another one's code) and this library's ABI version.
// SABC_RTC_CACHE=0 switches it off.  A file that does not parse (truncated, foreign) is ignored and overwritten.
uint64_t fnv1a(const void *data, size_t n, uint64_t h) {
  const unsigned char *p = static_cast<const unsigned char *>(data);
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

uint64_t headers_hash(const std::string &csrc_dir) {
  static std::mutex m;
  static std::map<std::string, uint64_t> memo;
  std::lock_guard<std::mutex> lock(m);
  auto hit = memo.find(csrc_dir);
  if (hit != memo.end()) return hit->second;
  static const char *files[] = {"update_kernel.hpp", "persistent_kernel.hpp", "device_models.hpp", "device_rng.hpp", "rng_tables.inc",
                                "kernels.hpp", "control.hpp", "host_math.hpp", "prior_math.hpp", "sabc_types.hpp", "p2p.hpp",
                                "../../include/sabc_hip.h"};
  uint64_t h = 14695981039346656037ull;
  for (const char *f : files) {
    const std::string path = csrc_dir + "/" + f;
    FILE *fp = std::fopen(path.c_str(), "rb");
    h = fnv1a(f, std::strlen(f), h);
    if (!fp) continue;
    char buf[1 << 14];
    size_t k;
    while ((k = std::fread(buf, 1, sizeof(buf), fp)) > 0) h = fnv1a(buf, k, h);
    std::fclose(fp);
  }
  memo[csrc_dir] = h;
  return h;
}

std::string disk_cache_dir() {
  if (const char *e = std::getenv("SABC_RTC_CACHE")) { if (e[0] == '0') return std::string(); }
  std::string dir;
  if (const char *e = std::getenv("SABC_RTC_CACHE_DIR")) dir = e;
  else if (const char *x = std::getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/sabc_hip";
  else if (const char *home = std::getenv("HOME")) dir = std::string(home) + "/.cache/sabc_hip";
  if (dir.empty()) return dir;
  for (size_t i = 1; i <= dir.size(); ++i)             // mkdir -p
    if (i == dir.size() || dir[i] == '/') (void)::mkdir(dir.substr(0, i).c_str(), 0700);
  return dir;
}

std::string disk_cache_path(const std::string &cache_key, const std::string &csrc_dir) {
  const std::string dir = disk_cache_dir();
  if (dir.empty()) return dir;
  uint64_t h = fnv1a(cache_key.data(), cache_key.size(), 14695981039346656037ull);
  const uint64_t hh = headers_hash(csrc_dir);
  const int abi = SABC_ABI_VERSION;
  h = fnv1a(&hh, sizeof(hh), h);
  h = fnv1a(&abi, sizeof(abi), h);
  uint64_t h2 = fnv1a(cache_key.data(), cache_key.size(), 0x9E3779B97F4A7C15ull ^ hh);      // 128 bits in the name
  char name[64];
  std::snprintf(name, sizeof(name), "/%016llx%016llx.sabcrtc", (unsigned long long)h, (unsigned long long)h2);
  return dir + name;
}

constexpr uint64_t kDiskMagic = 0x5341424352544331ull;   // "SABCRTC1"

bool disk_cache_load(const std::string &path, int n_kernels, CachedModule *out) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) return false;
  bool ok = false;
  uint64_t head[3] = {0, 0, 0};                        // magic, kernels, code bytes
  if (std::fread(head, sizeof(head), 1, fp) == 1 && head[0] == kDiskMagic && head[1] == (uint64_t)n_kernels && head[2] > 0 &&
      head[2] < ((uint64_t)1 << 30)) {
    ok = true;
    for (int i = 0; ok && i < n_kernels; ++i) {
      uint32_t len = 0;
      ok = std::fread(&len, sizeof(len), 1, fp) == 1 && len > 0 && len < 4096;
      if (ok) { out->lowered[i].assign(len, '\0'); ok = std::fread(&out->lowered[i][0], 1, len, fp) == len; }
    }
    if (ok) { out->code.resize((size_t)head[2]); ok = std::fread(out->code.data(), 1, out->code.size(), fp) == out->code.size(); }
    uint64_t tail = 0;                                 // the file was written to its end
    ok = ok && std::fread(&tail, sizeof(tail), 1, fp) == 1 && tail == (kDiskMagic ^ head[2]);
  }
  std::fclose(fp);
  return ok;
}

void disk_cache_store(const std::string &path, int n_kernels, const CachedModule &m) {
  char tmp[32];
  std::snprintf(tmp, sizeof(tmp), ".%d.tmp", (int)getpid());
  const std::string t = path + tmp;
  FILE *fp = std::fopen(t.c_str(), "wb");
  if (!fp) return;
  const uint64_t head[3] = {kDiskMagic, (uint64_t)n_kernels, (uint64_t)m.code.size()};
  bool ok = std::fwrite(head, sizeof(head), 1, fp) == 1;
  for (int i = 0; ok && i < n_kernels; ++i) {
    const uint32_t len = (uint32_t)m.lowered[i].size();
    ok = std::fwrite(&len, sizeof(len), 1, fp) == 1 && std::fwrite(m.lowered[i].data(), 1, len, fp) == len;
  }
  ok = ok && std::fwrite(m.code.data(), 1, m.code.size(), fp) == m.code.size();
  const uint64_t tail = kDiskMagic ^ head[2];
  ok = ok && std::fwrite(&tail, sizeof(tail), 1, fp) == 1;
  ok = (std::fclose(fp) == 0) && ok;
  if (ok) ok = std::rename(t.c_str(), path.c_str()) == 0;       // (a reader never sees a partial file)
  if (!ok) (void)std::remove(t.c_str());
}

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
  constexpr int kMaxKernels = 16;
  const int kKernels = with_persistent ? 16 : 7;
  HiprtcApi *api = hiprtc_api();
  if (!api) { *log = "libhiprtc.so could not be loaded: simulators from source need the hipRTC of ROCm"; return -1; }
  if (d < 1 || d > SABC_MAX_PARA || s < 1 || s > SABC_MAX_SOURCE_STATS) { *log = "n_para / n_stats out of range (a simulator from source: d <= 16, s <= 16)"; return -1; }
  char tail[2048];
  std::snprintf(tail, sizeof(tail),
                "\nnamespace sabc {\n"
                "template <int D, int S>\n"
                "struct Sim<SABC_MODEL_USER, D, S> {\n"
                "  static __device__ __forceinline__ void run(const ModelDesc &m, const double *th, uint64_t pid, uint64_t iter,\n"
                "                                             double *rho, int coop = 0) {\n"
                "    NormalStream ns(m.seed, pid, PURPOSE_SIM, iter, coop);\n"
                "    ::sabc_user_simulate(th, m.p, ns, rho);\n"
                "  }\n"
                "};\n"
                "}  // namespace sabc\n");
  std::string src = std::string("#include \"update_kernel.hpp\"\n") + (with_persistent ? "#include \"persistent_kernel.hpp\"\n" : "") + "#line 1 \"f_dist.hip\"\n";
  src += user_source;
  src += tail;

  char name[kMaxKernels][112];
  std::snprintf(name[0], sizeof(name[0]), "sabc::k_prior_simulate<%d, %d, %d>", SABC_MODEL_USER, d, s);
  for (int p = 0; p < 3; ++p) std::snprintf(name[1 + p], sizeof(name[1 + p]), "sabc::k_update<%d, %d, %d, %d>", SABC_MODEL_USER, d, s, p);
  std::snprintf(name[4], sizeof(name[4]), "sabc::k_simulate_batch<%d, %d, %d>", SABC_MODEL_USER, d, s);
  std::snprintf(name[5], sizeof(name[5]), "sabc::k_stats<%d, %d>", d, s);
  std::snprintf(name[6], sizeof(name[6]), "sabc::k_prior_op_t<%d>", d);
  for (int p = 0; p < 3; ++p) std::snprintf(name[7 + p], sizeof(name[7 + p]), "sabc::k_update_persistent<%d, %d, %d, %d>", SABC_MODEL_USER, d, s, p);
  for (int p = 0; p < 3; ++p) std::snprintf(name[10 + p], sizeof(name[10 + p]), "sabc::k_update_persistent<%d, %d, %d, %d, 4>", SABC_MODEL_USER, d, s, p);
  for (int p = 0; p < 3; ++p) std::snprintf(name[13 + p], sizeof(name[13 + p]), "sabc::k_update_persistent<%d, %d, %d, %d, 16>", SABC_MODEL_USER, d, s, p);
  // extra compiler flags (e.g. -DSABC_NO_BITOP3: the two-instruction form of the Philox round's three-input XOR)
  const char *extra_env = std::getenv("SABC_RTC_EXTRA_FLAGS");
  const std::string extra = extra_env ? extra_env : "";

  // (the launch geometry the library was built with is part of what is compiled: a variant library must not find another's code)
#define SABC_RTC_STR2(x) #x
#define SABC_RTC_STR(x) SABC_RTC_STR2(x)
  static const char *geometry = SABC_RTC_STR(SABC_UPDATE_BLOCK) "," SABC_RTC_STR(SABC_UPDATE_BLOCK_MS) "," SABC_RTC_STR(SABC_CDF_COARSE) ","
                                SABC_RTC_STR(SABC_CDF_COARSE_MS) "," SABC_RTC_STR(SABC_UPDATE_MIN_WAVES) "," SABC_RTC_STR(SABC_UPDATE_MIN_WAVES_MS);
#undef SABC_RTC_STR
#undef SABC_RTC_STR2
  const std::string cache_key = std::to_string(d) + "," + std::to_string(s) + (user_prior ? ",P" : "") + (with_persistent ? ",1L4L16" : "") + "," + extra + "," +
                                geometry + "\n" + user_source;
  if (out) {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto hit = g_cache.find(cache_key);
    if (hit != g_cache.end()) {
      RtcKernels k;
      k.d = d; k.s = s;
      if (hipModuleLoadData(&k.module, hit->second.code.data()) != hipSuccess) { *log = "hipModuleLoadData of the cached simulator failed"; return -1; }
      hipFunction_t *slots[kMaxKernels] = {&k.prior_simulate, &k.update[0], &k.update[1], &k.update[2], &k.simulate_batch, &k.stats, &k.prior_op,
                                        &k.persistent[0], &k.persistent[1], &k.persistent[2], &k.persistent4[0], &k.persistent4[1], &k.persistent4[2],
                                        &k.persistent16[0], &k.persistent16[1], &k.persistent16[2]};
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
  const std::string disk_path = out ? disk_cache_path(cache_key, csrc_dir) : std::string();
  if (out && !disk_path.empty()) {
    CachedModule entry;
    if (disk_cache_load(disk_path, kKernels, &entry)) {
      RtcKernels k;
      k.d = d; k.s = s;
      bool ok = hipModuleLoadData(&k.module, entry.code.data()) == hipSuccess;
      hipFunction_t *slots[kMaxKernels] = {&k.prior_simulate, &k.update[0], &k.update[1], &k.update[2], &k.simulate_batch, &k.stats, &k.prior_op,
                                           &k.persistent[0], &k.persistent[1], &k.persistent[2], &k.persistent4[0], &k.persistent4[1], &k.persistent4[2],
                                        &k.persistent16[0], &k.persistent16[1], &k.persistent16[2]};
      for (int i = 0; ok && i < kKernels; ++i) ok = hipModuleGetFunction(slots[i], k.module, entry.lowered[i].c_str()) == hipSuccess;
      if (ok) {
        if (code_size) *code_size = entry.code.size();
        {
          std::lock_guard<std::mutex> lock(g_cache_mutex);
          g_cache.emplace(cache_key, std::move(entry));
        }
        *out = k;
        return 0;
      }
      (void)hipGetLastError();                         // a code object this device does not take: compile afresh, overwrite
      rtc_release(&k);
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
                                        &k.persistent[0], &k.persistent[1], &k.persistent[2], &k.persistent4[0], &k.persistent4[1], &k.persistent4[2],
                                        &k.persistent16[0], &k.persistent16[1], &k.persistent16[2]};
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
  if (!disk_path.empty()) disk_cache_store(disk_path, kKernels, entry);
  {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    g_cache.emplace(cache_key, std::move(entry));
  }
  *out = k;
  return 0;
}

}  // namespace sabc
