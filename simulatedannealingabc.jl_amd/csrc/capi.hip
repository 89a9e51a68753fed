// capi.hip -- the extern "C" surface of libsabc_hip.so (include/sabc_hip.h).
// Plain pointers and sizes only; no exceptions cross this boundary.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/sabc_hip.h"
#include "engine.hpp"
#include "hip_backend.hpp"
#include "host_math.hpp"
#include "kernels.hpp"
#include "p2p.hpp"
#include "p2p_setup.hpp"
#include "rtc.hpp"

using namespace sabc;

namespace {

thread_local std::string g_err;

// world == 1
class NoCollectives : public Collectives {
 public:
  // only ever called with world > 1 (the engine skips collectives on one shard): no transport was installed
  int allreduce_sum(double *, int64_t) override { return -1; }
  int allgather(const double *, double *, int64_t) override { return -1; }
  bool usable() const override { return false; }
};

// user-supplied hooks (torch.distributed from Python, MPI/RCCL from Julia)
class CallbackCollectives : public Collectives {
 public:
  CallbackCollectives(HipBackend *be, int world, sabc_allreduce_fn ar, sabc_allgather_fn ag, void *ctx, bool dev)
      : be_(be), world_(world), ar_(ar), ag_(ag), ctx_(ctx), dev_(dev) {}
  void set_alltoallv(sabc_alltoallv_fn fn) { a2a_ = fn; }
  bool has_alltoallv() const override { return a2a_ != nullptr; }
  int alltoallv(const double *send, const int64_t *sc, double *recv, const int64_t *rc) override {
    if (!a2a_) return -1;
    if (dev_) return a2a_(ctx_, send, sc, recv, rc, world_, (void *)be_->stream());
    int64_t ns = 0, nr = 0;
    for (int p = 0; p < world_; ++p) { ns += sc[p]; nr += rc[p]; }
    double *h = be_->host_stage(ns + nr + 2);
    if (ns > 0 && hipMemcpyAsync(h, send, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost, be_->stream()) != hipSuccess) return -1;
    if (hipStreamSynchronize(be_->stream()) != hipSuccess) return -1;
    if (a2a_(ctx_, h, sc, h + ns, rc, world_, nullptr)) return -1;
    if (nr > 0 && hipMemcpyAsync(recv, h + ns, (size_t)nr * sizeof(double), hipMemcpyHostToDevice, be_->stream()) != hipSuccess) return -1;
    return hipStreamSynchronize(be_->stream()) == hipSuccess ? 0 : -1;
  }
  int allreduce_sum(double *buf, int64_t count) override {
    if (dev_) return ar_(ctx_, buf, count, (void *)be_->stream());
    double *h = be_->host_stage(count);
    if (hipMemcpyAsync(h, buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, be_->stream()) != hipSuccess) return -1;
    if (hipStreamSynchronize(be_->stream()) != hipSuccess) return -1;
    if (ar_(ctx_, h, count, nullptr)) return -1;
    if (hipMemcpyAsync(buf, h, (size_t)count * sizeof(double), hipMemcpyHostToDevice, be_->stream()) != hipSuccess) return -1;
    return hipStreamSynchronize(be_->stream()) == hipSuccess ? 0 : -1;
  }
  int allgather(const double *send, double *recv, int64_t count) override {
    if (dev_) return ag_(ctx_, send, recv, count, (void *)be_->stream());
    double *h = be_->host_stage(count * (world_ + 1));
    if (hipMemcpyAsync(h, send, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, be_->stream()) != hipSuccess) return -1;
    if (hipStreamSynchronize(be_->stream()) != hipSuccess) return -1;
    if (ag_(ctx_, h, h + count, count, nullptr)) return -1;
    if (hipMemcpyAsync(recv, h + count, (size_t)count * world_ * sizeof(double), hipMemcpyHostToDevice, be_->stream()) != hipSuccess) return -1;
    return hipStreamSynchronize(be_->stream()) == hipSuccess ? 0 : -1;
  }

 private:
  HipBackend *be_;
  int world_;
  sabc_allreduce_fn ar_;
  sabc_allgather_fn ag_;
  sabc_alltoallv_fn a2a_ = nullptr;
  void *ctx_;
  bool dev_;
};

// RCCL over xGMI, bound at run time so that the library loads on hosts without librccl
// (and picks up the copy torch already mapped when there is one).
struct Id128 { char b[128]; };   // ncclUniqueId, passed by value
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
};

RcclApi *rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      api.GetUniqueId = (int (*)(void *))dlsym(api.lib, "ncclGetUniqueId");
      api.CommInitRank = (int (*)(void **, int, Id128, int))dlsym(api.lib, "ncclCommInitRank");
      api.CommDestroy = (int (*)(void *))dlsym(api.lib, "ncclCommDestroy");
      api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(api.lib, "ncclAllReduce");
      api.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(api.lib, "ncclAllGather");
      api.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(api.lib, "ncclSend");
      api.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(api.lib, "ncclRecv");
      api.GroupStart = (int (*)())dlsym(api.lib, "ncclGroupStart");
      api.GroupEnd = (int (*)())dlsym(api.lib, "ncclGroupEnd");
    }
  });
  return (api.lib && api.GetUniqueId && api.CommInitRank && api.AllReduce && api.AllGather) ? &api : nullptr;
}

class RcclCollectives : public Collectives {
 public:
  RcclCollectives(HipBackend *be, void *comm, int world) : be_(be), comm_(comm), world_(world) {}
  // The grouped send / recv exchange has not run between two GPUs yet (every box so far had one): it is OFF unless
  // SABC_RCCL_ALLTOALLV=1, and the sharded resample on this transport allgathers the whole population -- more bytes, but
  // nothing beyond ncclAllGather.  (Inside a node the peer-to-peer transport carries the resample anyway.)
  bool has_alltoallv() const override {
    static const bool allowed = [] { const char *e = std::getenv("SABC_RCCL_ALLTOALLV"); return e && e[0] == '1'; }();
    RcclApi *a = rccl_api();
    return allowed && a && a->Send && a->Recv && a->GroupStart && a->GroupEnd;
  }
  // grouped ncclSend / ncclRecv, one pair per peer with a non-empty segment (ncclFloat64 = 8)
  int alltoallv(const double *send, const int64_t *sc, double *recv, const int64_t *rc) override {
    RcclApi *a = rccl_api();
    if (a->GroupStart()) return -1;
    int bad = 0;
    int64_t so = 0, ro = 0;
    for (int p = 0; p < world_; ++p) {
      if (sc[p] > 0) bad |= a->Send(send + so, (size_t)sc[p], 8, p, comm_, be_->stream());
      if (rc[p] > 0) bad |= a->Recv(recv + ro, (size_t)rc[p], 8, p, comm_, be_->stream());
      so += sc[p]; ro += rc[p];
    }
    const int end = a->GroupEnd();
    return (bad || end) ? -1 : 0;
  }
  ~RcclCollectives() override {
    RcclApi *a = rccl_api();
    if (a && a->CommDestroy && comm_) a->CommDestroy(comm_);
  }
  int allreduce_sum(double *buf, int64_t count) override {   // ncclFloat64 = 8, ncclSum = 0
    return rccl_api()->AllReduce(buf, buf, (size_t)count, 8, 0, comm_, be_->stream());
  }
  int allgather(const double *send, double *recv, int64_t count) override {
    return rccl_api()->AllGather(send, recv, (size_t)count, 8, comm_, be_->stream());
  }

 private:
  HipBackend *be_;
  void *comm_;
  int world_;
};

int usable_device(int device, std::string &why) {
  (void)hipGetLastError();   // sticky: do not let an earlier failure leak into the launch checks below
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    why = "no usable HIP device (libsabc_hip.so has no CPU path): ";
    why += hipGetErrorString(e);
    return SABC_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    why = "device ordinal out of range";
    return SABC_ERR_NO_DEVICE;
  }
  return 0;
}

}  // namespace

struct sabc_handle {
  Engine *eng = nullptr;
  HipBackend *be = nullptr;
  Collectives *coll = nullptr;
  std::string err;
};

namespace {
int hfail(sabc_handle *h, int code) {
  if (!h) return code;
  h->err = h->eng && !h->eng->error().empty() ? h->eng->error() : h->err;
  if (h->be && !h->be->error().empty()) { h->err += h->err.empty() ? "" : " | "; h->err += h->be->error(); }
  return code;
}
int hset(sabc_handle *h, int code, const char *msg) {
  if (h) h->err = msg;
  return code;
}
}  // namespace

extern "C" {

int sabc_abi_version(void) { return SABC_ABI_VERSION; }
const char *sabc_last_global_error(void) { return g_err.c_str(); }

int sabc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sabc_create(const sabc_config *cfg, sabc_handle **out) {
  if (!cfg || !out) { g_err = "null argument"; return SABC_ERR_BAD_CONFIG; }
  *out = nullptr;
  sabc_handle *h = new (std::nothrow) sabc_handle();
  if (!h) { g_err = "out of memory"; return SABC_ERR_BAD_CONFIG; }
  // configuration errors are reported even without a GPU (they are the reference's own errors)
  h->be = new HipBackend(cfg->device);
  h->coll = new NoCollectives();
  h->eng = new Engine(*cfg, h->be, h->coll);
  int rc = h->eng->validate();
  if (rc) { g_err = h->eng->error(); sabc_destroy(h); return rc; }
  std::string why;
  rc = usable_device(cfg->device, why);
  if (rc) { g_err = why; sabc_destroy(h); return rc; }
  if (h->be->allocate(h->eng->model(), h->eng->shard())) {
    g_err = h->be->error();
    sabc_destroy(h);
    return SABC_ERR_HIP;
  }
  *out = h;
  return 0;
}

void sabc_destroy(sabc_handle *h) {
  if (!h) return;
  delete h->eng;
  delete h->coll;
  delete h->be;
  delete h;
}

const char *sabc_last_error(const sabc_handle *h) { return h ? h->err.c_str() : g_err.c_str(); }

int sabc_set_stream(sabc_handle *h, void *hip_stream) {
  if (!h) return SABC_ERR_STATE;
  return h->be->set_stream((hipStream_t)hip_stream) ? hfail(h, SABC_ERR_HIP) : 0;
}

int sabc_set_collectives(sabc_handle *h, sabc_allreduce_fn ar, sabc_allgather_fn ag, void *ctx, int device_buffers) {
  if (!h || !ar || !ag) return hset(h, SABC_ERR_COMM, "null collective hook");
  Collectives *c = new CallbackCollectives(h->be, h->eng->shard().world, ar, ag, ctx, device_buffers != 0);
  delete h->coll;
  h->coll = c;
  h->eng->set_collectives(c);
  return 0;
}

int sabc_set_alltoallv(sabc_handle *h, sabc_alltoallv_fn fn) {
  if (!h || !fn) return hset(h, SABC_ERR_COMM, "null collective hook");
  CallbackCollectives *c = dynamic_cast<CallbackCollectives *>(h->coll);
  if (!c) return hset(h, SABC_ERR_COMM, "sabc_set_alltoallv needs the hooks of sabc_set_collectives to be installed first");
  c->set_alltoallv(fn);
  return 0;
}

int64_t sabc_comm_bytes(const sabc_handle *h) { return h ? h->eng->comm_bytes() : 0; }

static int compile_only(const char *hip_source, int32_t d, int32_t s, char *log_out, int64_t log_cap, bool user_prior);

int sabc_op_compile_device_simulator(const char *hip_source, int32_t d, int32_t s, char *log_out, int64_t log_cap) {
  return compile_only(hip_source, d, s, log_out, log_cap, false);
}
int sabc_op_compile_device_simulator_with_prior(const char *hip_source, int32_t d, int32_t s, char *log_out, int64_t log_cap) {
  return compile_only(hip_source, d, s, log_out, log_cap, true);
}

static int compile_only(const char *hip_source, int32_t d, int32_t s, char *log_out, int64_t log_cap, bool user_prior) {
  if (!hip_source) { g_err = "null device simulator source"; return SABC_ERR_BAD_CONFIG; }
  std::string log;
  size_t cs = 0;
  const int rc = rtc_compile(hip_source, d, s, rtc_default_csrc_dir(), nullptr, &log, &cs, user_prior, /*with_persistent=*/persistent_fits(d, s));
  if (log_out && log_cap > 0) { std::snprintf(log_out, (size_t)log_cap, "%s", log.c_str()); }
  if (rc) { g_err = "compiling the device simulator failed:\n" + log; return SABC_ERR_BAD_CONFIG; }
  return 0;
}

int sabc_register_device_simulator(sabc_handle *h, const char *hip_source) {
  if (!h || !hip_source) return hset(h, SABC_ERR_BAD_CONFIG, "null device simulator source");
  if (h->eng->model().model_id != SABC_MODEL_USER) return hset(h, SABC_ERR_BAD_CONFIG, "the handle was not created with SABC_MODEL_USER");
  h->err.clear();
  return h->be->register_device_simulator(hip_source) ? hfail(h, SABC_ERR_BAD_CONFIG) : 0;
}

int sabc_set_host_simulator(sabc_handle *h, sabc_simulate_fn fn, void *ctx) {
  if (!h || !fn) return hset(h, SABC_ERR_BAD_CONFIG, "null host simulator");
  if (h->eng->model().model_id != SABC_MODEL_HOST) return hset(h, SABC_ERR_BAD_CONFIG, "the handle was not created with SABC_MODEL_HOST");
  h->be->set_host_simulator(fn, ctx);
  return 0;
}

int sabc_set_host_prior(sabc_handle *h, sabc_prior_sample_fn sample, sabc_prior_logpdf_fn logpdf, void *ctx) {
  if (!h || !sample || !logpdf) return hset(h, SABC_ERR_BAD_CONFIG, "null host prior callback");
  if (h->eng->model().prior_joint != 2) return hset(h, SABC_ERR_BAD_CONFIG, "the handle was not created with prior_joint = 2");
  h->be->set_host_prior(sample, logpdf, ctx);
  return 0;
}

int sabc_set_host_chunk(sabc_handle *h, int64_t particles) {
  if (!h) return SABC_ERR_STATE;
  h->be->set_host_chunk(particles);
  return 0;
}
double sabc_host_callback_seconds(const sabc_handle *h) { return h ? h->be->host_callback_seconds() : 0.0; }
int64_t sabc_host_callback_calls(const sabc_handle *h) { return h ? h->be->host_callback_calls() : 0; }

int sabc_comm_unique_id(void *out_128b) {
  RcclApi *a = rccl_api();
  if (!a) { g_err = "librccl.so could not be loaded"; return SABC_ERR_COMM; }
  return a->GetUniqueId(out_128b) == 0 ? 0 : SABC_ERR_COMM;
}

int sabc_comm_init_rccl(sabc_handle *h, const void *unique_id_128b) {
  if (!h) return SABC_ERR_STATE;
  RcclApi *a = rccl_api();
  if (!a) return hset(h, SABC_ERR_COMM, "librccl.so could not be loaded");
  if (hipSetDevice(h->be->device()) != hipSuccess) return hset(h, SABC_ERR_HIP, "hipSetDevice failed");
  Id128 id;
  std::memcpy(id.b, unique_id_128b, 128);
  void *comm = nullptr;
  const Shard &sh = h->eng->shard();
  if (a->CommInitRank(&comm, sh.world, id, sh.rank) != 0) return hset(h, SABC_ERR_COMM, "ncclCommInitRank failed");
  delete h->coll;
  h->coll = new RcclCollectives(h->be, comm, sh.world);
  h->eng->set_collectives(h->coll);
  return 0;
}

// Exercise the installed collectives once and check the result on the host: an allgather of an odd-sized block (5
// doubles: not a multiple of any vector width), an allreduce, and -- when the transport has one -- a personalised
// exchange with a different segment length for every (sender, receiver) pair, as the resample issues it.
// Validates the transport before the first population update.
int sabc_comm_selftest(sabc_handle *h) {
  if (!h) return SABC_ERR_STATE;
  const Shard &sh = h->eng->shard();
  if (hipSetDevice(h->be->device()) != hipSuccess) return hset(h, SABC_ERR_HIP, "hipSetDevice failed");
  const int world = sh.world, B = 5;
  if (world == 1) return 0;
  double *g = h->be->gather_buffer((int64_t)(world + 1) * B);
  if (!g) return hset(h, SABC_ERR_HIP, "out of memory for the self-test buffer");
  double send[B] = {(double)(sh.rank + 1), 2.0, 3.0, (double)(100 + sh.rank), -1.5};
  std::vector<double> back((size_t)(world + 1) * B, 0.0);
  if (h->be->to_backend(g, send, B)) return hfail(h, SABC_ERR_HIP);
  if (h->coll->allgather(g, g + B, B)) return hset(h, SABC_ERR_COMM, "self-test allgather failed");
  if (h->coll->allreduce_sum(g, B)) return hset(h, SABC_ERR_COMM, "self-test allreduce failed");
  if (h->be->to_host(back.data(), g, (int64_t)back.size())) return hfail(h, SABC_ERR_HIP);
  if (back[0] != 0.5 * world * (world + 1) || back[1] != 2.0 * world || back[4] != -1.5 * world)
    return hset(h, SABC_ERR_COMM, "self-test allreduce gave a wrong sum");
  for (int r = 0; r < world; ++r)
    if (back[B + B * r] != (double)(r + 1) || back[B + B * r + 3] != (double)(100 + r) || back[B + B * r + 4] != -1.5)
      return hset(h, SABC_ERR_COMM, "self-test allgather gave wrong words");
  if (h->coll->has_alltoallv()) {
    // rank r sends (r + p + 1) doubles of value 1000 r + p to rank p
    std::vector<int64_t> sc((size_t)world), rc((size_t)world);
    int64_t ns = 0, nr = 0;
    for (int p = 0; p < world; ++p) { sc[(size_t)p] = rc[(size_t)p] = sh.rank + p + 1; ns += sc[(size_t)p]; nr += rc[(size_t)p]; }
    std::vector<double> out((size_t)ns), in((size_t)nr, 0.0);
    int64_t o = 0;
    for (int p = 0; p < world; ++p)
      for (int64_t k = 0; k < sc[(size_t)p]; ++k) out[(size_t)o++] = 1000.0 * sh.rank + p;
    double *ds = h->be->scratch_buffer(0, ns), *dr = h->be->scratch_buffer(1, nr);
    if (!ds || !dr) return hset(h, SABC_ERR_HIP, "out of memory for the self-test buffer");
    if (h->be->to_backend(ds, out.data(), ns)) return hfail(h, SABC_ERR_HIP);
    if (h->coll->alltoallv(ds, sc.data(), dr, rc.data())) return hset(h, SABC_ERR_COMM, "self-test alltoallv failed");
    if (h->be->to_host(in.data(), dr, nr)) return hfail(h, SABC_ERR_HIP);
    o = 0;
    for (int p = 0; p < world; ++p)
      for (int64_t k = 0; k < rc[(size_t)p]; ++k)
        if (in[(size_t)o++] != 1000.0 * p + sh.rank) return hset(h, SABC_ERR_COMM, "self-test alltoallv gave wrong words");
  }
  return 0;
}

// ---- peer-to-peer transport (csrc/p2p.hpp) ----
int sabc_comm_p2p_descriptor(sabc_handle *h, void *out_desc) {
  if (!h || !out_desc) return SABC_ERR_STATE;
  return h->be->p2p_descriptor(reinterpret_cast<P2PDesc *>(out_desc)) ? hfail(h, SABC_ERR_COMM) : 0;
}

int sabc_comm_p2p_init(sabc_handle *h, const void *all_descs) {
  if (!h) return SABC_ERR_STATE;
  const Shard &sh = h->eng->shard();
  if (sh.world < 2 || sh.world > SABC_P2P_MAX_WORLD) return hset(h, SABC_ERR_COMM, "the peer-to-peer transport takes 2..8 shards (one node)");
  if (hipSetDevice(h->be->device()) != hipSuccess) return hset(h, SABC_ERR_HIP, "hipSetDevice failed");
  std::vector<P2PDesc> all((size_t)sh.world);
  if (all_descs) {
    std::memcpy(all.data(), all_descs, all.size() * sizeof(P2PDesc));
  } else {
    P2PDesc mine;
    if (h->be->p2p_descriptor(&mine)) return hfail(h, SABC_ERR_COMM);
    std::string why;
    if (int rc = p2p_gather_descriptors(h->be, h->coll, sh.world, mine, all, &why)) return hset(h, rc, why.c_str());
  }
  return h->be->p2p_init(all.data()) ? hfail(h, SABC_ERR_COMM) : 0;
}

// The whole set-up, with the shards agreeing after every step (csrc/p2p_setup.hpp).
int sabc_comm_p2p_setup(sabc_handle *h) {
  if (!h) return SABC_ERR_STATE;
  if (hipSetDevice(h->be->device()) != hipSuccess) return hset(h, SABC_ERR_HIP, "hipSetDevice failed");
  std::string note;
  const int rc = p2p_setup_sequence(h->be, h->coll, h->eng->shard(), h->eng->host_mode(), &note);
  h->err = note;
  return rc == 1 ? (h->eng->p2p() ? 1 : 0) : rc;
}

int sabc_comm_p2p_selftest(sabc_handle *h) {
  if (!h) return SABC_ERR_STATE;
  return h->be->p2p_selftest() ? hfail(h, SABC_ERR_COMM) : 0;
}

int sabc_comm_p2p_set_timeout(sabc_handle *h, double milliseconds) {
  if (!h || !(milliseconds > 0)) return hset(h, SABC_ERR_BAD_CONFIG, "the bound of a peer-to-peer wait must be positive");
  h->be->p2p_set_timeout(milliseconds);
  return 0;
}

int sabc_comm_p2p_disable(sabc_handle *h) {
  if (!h) return SABC_ERR_STATE;
  h->be->p2p_disable();
  return 0;
}

int sabc_comm_p2p_active(const sabc_handle *h) { return h && h->eng->p2p() ? 1 : 0; }
int64_t sabc_comm_p2p_fallbacks(const sabc_handle *h) { return h ? h->eng->p2p_fallbacks() : 0; }

int sabc_comm_p2p_inject_silence(sabc_handle *h, int32_t n) {
  if (!h) return SABC_ERR_STATE;
  h->be->p2p_inject_silence(n);
  return 0;
}

int sabc_comm_p2p_inject_loss(sabc_handle *h, int32_t n) {
  if (!h) return SABC_ERR_STATE;
  h->be->p2p_inject_loss(n);
  return 0;
}

int sabc_comm_p2p_inject_stale(sabc_handle *h, int32_t n) {
  if (!h) return SABC_ERR_STATE;
  h->be->p2p_inject_stale(n);
  return 0;
}

int sabc_comm_p2p_set_destroy_wait(sabc_handle *h, double milliseconds) {
  if (!h || !(milliseconds >= 0)) return hset(h, SABC_ERR_BAD_CONFIG, "the wait of sabc_destroy must not be negative");
  h->be->p2p_set_destroy_wait(milliseconds);
  return 0;
}

int64_t sabc_comm_p2p_parked_bytes(void) { return HipBackend::parked_bytes(); }
int64_t sabc_persistent_launches(const sabc_handle *h) { return h ? h->eng->persistent_launches() : 0; }
int32_t sabc_persistent_lanes(const sabc_handle *h) { return h ? h->eng->persistent_lanes() : 0; }
int64_t sabc_persistent_fallbacks(const sabc_handle *h) { return h ? h->eng->persistent_fallbacks() : 0; }

int sabc_initialize(sabc_handle *h, int64_t n_simulation) {
  if (!h) return SABC_ERR_STATE;
  if (hipSetDevice(h->be->device()) != hipSuccess) return hset(h, SABC_ERR_HIP, "hipSetDevice failed");
  const int rc = h->eng->initialize(n_simulation);
  return rc ? hfail(h, rc) : 0;
}

int sabc_update(sabc_handle *h, const sabc_update_args *args) {
  if (!h || !args) return SABC_ERR_STATE;
  if (hipSetDevice(h->be->device()) != hipSuccess) return hset(h, SABC_ERR_HIP, "hipSetDevice failed");
  const int rc = h->eng->update(*args);
  return rc ? hfail(h, rc) : 0;
}

int64_t sabc_n_global(const sabc_handle *h) { return h ? h->eng->shard().n_global : 0; }
int64_t sabc_n_local(const sabc_handle *h) { return h ? h->eng->shard().n_local : 0; }
int64_t sabc_local_offset(const sabc_handle *h) { return h ? h->eng->shard().gid0 : 0; }

int sabc_get_population(sabc_handle *h, double *theta, double *u, double *rho) {
  if (!h) return SABC_ERR_STATE;
  return h->be->download(theta, u, rho) ? hfail(h, SABC_ERR_HIP) : 0;
}

int sabc_set_population(sabc_handle *h, const double *theta, const double *u, const double *rho) {
  if (!h) return SABC_ERR_STATE;
  if (h->be->upload(theta, u, rho)) return hfail(h, SABC_ERR_HIP);
  h->eng->mark_initialized();
  return 0;
}

int sabc_get_counters(const sabc_handle *h, int64_t out[4]) {
  if (!h) return SABC_ERR_STATE;
  h->eng->counters(out);
  return 0;
}
int sabc_set_counters(sabc_handle *h, const int64_t in[4]) {
  if (!h) return SABC_ERR_STATE;
  h->eng->set_counters(in);
  return 0;
}
int sabc_get_epsilon(const sabc_handle *h, double *eps, int32_t *len) {
  if (!h) return SABC_ERR_STATE;
  for (int i = 0; i < h->eng->eps_len(); ++i) eps[i] = h->eng->eps()[i];
  if (len) *len = h->eng->eps_len();
  return 0;
}
int sabc_set_epsilon(sabc_handle *h, const double *eps, int32_t len) {
  if (!h) return SABC_ERR_STATE;
  const int rc = h->eng->set_eps(eps, len);
  return rc ? hfail(h, rc) : 0;
}
int64_t sabc_history_len(const sabc_handle *h) { return h ? h->eng->history_len() : 0; }
int sabc_get_history(const sabc_handle *h, double *e, double *u, double *r) {
  if (!h) return SABC_ERR_STATE;
  h->eng->history(e, u, r);
  return 0;
}
int sabc_clear_history(sabc_handle *h) {
  if (!h) return SABC_ERR_STATE;
  h->eng->clear_history();
  return 0;
}

int64_t sabc_cdf_len(const sabc_handle *h, int32_t stat) {
  if (!h || stat < 0 || stat >= h->eng->model().s) return 0;
  return h->eng->cdf_len()[stat];
}
int sabc_get_cdf_knots(sabc_handle *h, int32_t stat, double *out) {
  if (!h) return SABC_ERR_STATE;
  return h->be->get_knots(stat, out, sabc_cdf_len(h, stat)) ? hfail(h, SABC_ERR_HIP) : 0;
}
int sabc_set_cdf_knots(sabc_handle *h, int32_t stat, const double *knots, int64_t len) {
  if (!h) return SABC_ERR_STATE;
  if (h->be->set_knots(stat, knots, len)) return hfail(h, SABC_ERR_BAD_CONFIG);
  h->eng->set_cdf_len(stat, len);
  return 0;
}
int sabc_cdf_apply(sabc_handle *h, const double *rho, int64_t m, double *u_out) {
  if (!h) return SABC_ERR_STATE;
  for (int j = 0; j < h->eng->model().s; ++j)
    if (h->eng->cdf_len()[j] < 3) return hset(h, SABC_ERR_STATE, "ECDF tables are not built yet");
  return h->be->cdf_apply_host(rho, m, u_out) ? hfail(h, SABC_ERR_HIP) : 0;
}
int sabc_get_proposal_sigma(const sabc_handle *h, double *sigma) {
  if (!h) return SABC_ERR_STATE;
  const int d = h->eng->model().d;
  for (int i = 0; i < d * d; ++i) sigma[i] = h->eng->sigma()[i];
  return 0;
}
double sabc_last_ess(const sabc_handle *h) { return h ? h->be->last_ess() : 0.0; }

// ---- stand-alone operators -----------------------------------------------------------------
int sabc_op_build_cdf(int32_t device, const double *x, int64_t n, double *knots_out, int64_t *len_out) {
  std::string why;
  int rc = usable_device(device, why);
  if (rc) { g_err = why; return rc; }
  if (n < 1) { g_err = "empty input"; return SABC_ERR_EMPTY_CDF; }
  if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return SABC_ERR_HIP; }
  double *a = nullptr, *b = nullptr, *k = nullptr;
  int64_t *meta = nullptr;
  void *tmp = nullptr;
  size_t bytes = 0;
  hipError_t e = hipMalloc((void **)&a, (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&b, (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&k, (size_t)(n + 2) * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&meta, 16);
  if (e == hipSuccess) e = (hipError_t)sort_f64(a, b, n, nullptr, &bytes, nullptr);
  if (e == hipSuccess) e = hipMalloc(&tmp, bytes ? bytes : 16);
  if (e == hipSuccess) e = hipMemcpy(a, x, (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = (hipError_t)sort_f64(a, b, n, tmp, &bytes, nullptr);
  if (e == hipSuccess) e = (hipError_t)launch_cdf_knots(b, n, k, meta, nullptr);
  int64_t hm[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(hm, meta, 16, hipMemcpyDeviceToHost);
  int64_t len = 0;
  if (e == hipSuccess) {
    const int64_t mpos = n - hm[0];
    len = mpos > 0 ? mpos + 2 : 0;
    if (len) e = hipMemcpy(knots_out, k, (size_t)len * 8, hipMemcpyDeviceToHost);
  }
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(k); (void)hipFree(meta); (void)hipFree(tmp);
  if (e != hipSuccess) { g_err = hipGetErrorString(e); return SABC_ERR_HIP; }
  if (hm[1]) { g_err = "Negative distances are not allowed!"; return SABC_ERR_NEG_DISTANCE; }
  if (!len) { g_err = "no positive entry"; return SABC_ERR_EMPTY_CDF; }
  *len_out = len;
  return 0;
}

int sabc_op_sort(int32_t device, const double *x, int64_t n, double *out) {
  std::string why;
  int rc = usable_device(device, why);
  if (rc) { g_err = why; return rc; }
  if (n <= 0) return 0;
  if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return SABC_ERR_HIP; }
  double *a = nullptr, *b = nullptr;
  void *tmp = nullptr;
  size_t bytes = 0;
  hipError_t e = hipMalloc((void **)&a, (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&b, (size_t)n * 8);
  if (e == hipSuccess) e = (hipError_t)sort_f64(a, b, n, nullptr, &bytes, nullptr);
  if (e == hipSuccess) e = hipMalloc(&tmp, bytes ? bytes : 16);
  if (e == hipSuccess) e = hipMemcpy(a, x, (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = (hipError_t)sort_f64(a, b, n, tmp, &bytes, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, b, (size_t)n * 8, hipMemcpyDeviceToHost);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(tmp);
  if (e != hipSuccess) { g_err = hipGetErrorString(e); return SABC_ERR_HIP; }
  return 0;
}

int sabc_op_cdf_eval(int32_t device, const double *knots, int64_t len, const double *q, int64_t m, double *out) {
  std::string why;
  int rc = usable_device(device, why);
  if (rc) { g_err = why; return rc; }
  if (len < 2 || m < 0) { g_err = "bad length"; return SABC_ERR_BAD_CONFIG; }
  if (m == 0) return 0;
  if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return SABC_ERR_HIP; }
  double *k = nullptr, *dq = nullptr, *dout = nullptr;
  hipError_t e = hipMalloc((void **)&k, (size_t)len * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&dq, (size_t)m * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&dout, (size_t)m * 8);
  if (e == hipSuccess) e = hipMemcpy(k, knots, (size_t)len * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dq, q, (size_t)m * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = (hipError_t)launch_cdf_eval(k, len, dq, m, dout, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)m * 8, hipMemcpyDeviceToHost);
  (void)hipFree(k); (void)hipFree(dq); (void)hipFree(dout);
  if (e != hipSuccess) { g_err = hipGetErrorString(e); return SABC_ERR_HIP; }
  return 0;
}

int sabc_op_eps_single(double ubar, double v, double *eps_out) {
  *eps_out = hostmath::eps_single(ubar, v);
  return 0;
}

int sabc_op_eps_multi(const double *ubar, int32_t s, double v, double *eps_out) {
  if (s < 1 || s > SABC_MAX_STATS) return SABC_ERR_BAD_CONFIG;
  return hostmath::eps_multi(ubar, s, v, eps_out) ? 0 : SABC_ERR_ZERO_MEAN_U;
}

int sabc_op_simulate(sabc_handle *h, const double *theta, int64_t m, uint64_t pid0, uint64_t iter, double *rho_out) {
  if (!h) return SABC_ERR_STATE;
  return h->be->simulate_host(theta, m, pid0, iter, rho_out) ? hfail(h, SABC_ERR_HIP) : 0;
}

int sabc_op_prior(sabc_handle *h, uint64_t pid0, int64_t m, double *theta_out, double *logpdf_out) {
  if (!h || !theta_out || !logpdf_out) return SABC_ERR_STATE;
  return h->be->prior_host(pid0, m, theta_out, logpdf_out) ? hfail(h, SABC_ERR_HIP) : 0;
}

int sabc_op_philox(int32_t device, uint64_t seed, uint64_t pid, uint32_t purpose, uint64_t iter, uint32_t k,
                   uint32_t out_words[4], double out_normals[2]) {
  std::string why;
  int rc = usable_device(device, why);
  if (rc) { g_err = why; return rc; }
  if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return SABC_ERR_HIP; }
  uint32_t *w = nullptr;
  double *z = nullptr;
  hipError_t e = hipMalloc((void **)&w, 16);
  if (e == hipSuccess) e = hipMalloc((void **)&z, 16);
  if (e == hipSuccess) e = (hipError_t)launch_philox_debug(seed, pid, purpose, iter, k, w, z, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out_words, w, 16, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_normals, z, 16, hipMemcpyDeviceToHost);
  (void)hipFree(w); (void)hipFree(z);
  if (e != hipSuccess) { g_err = hipGetErrorString(e); return SABC_ERR_HIP; }
  return 0;
}

int sabc_op_normal_pairs(int32_t device, uint64_t seed, uint64_t pid0, uint32_t purpose, uint64_t iter, uint32_t k,
                         int64_t m, double *out_2m) {
  std::string why;
  int rc = usable_device(device, why);
  if (rc) { g_err = why; return rc; }
  if (m <= 0) return 0;
  if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return SABC_ERR_HIP; }
  double *d = nullptr;
  hipError_t e = hipMalloc((void **)&d, (size_t)m * 16);
  if (e == hipSuccess) e = (hipError_t)launch_normal_pairs(seed, pid0, purpose, iter, k, m, d, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out_2m, d, (size_t)m * 16, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { g_err = hipGetErrorString(e); return SABC_ERR_HIP; }
  return 0;
}

int sabc_op_rng_peak(int32_t device, int64_t n_lanes, int32_t pairs_per_lane, int32_t repeats, double *normals_per_s) {
  std::string why;
  int rc = usable_device(device, why);
  if (rc) { g_err = why; return rc; }
  if (n_lanes < 1 || pairs_per_lane < 1 || repeats < 1) { g_err = "bad arguments"; return SABC_ERR_BAD_CONFIG; }
  if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return SABC_ERR_HIP; }
  double *d = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t e = hipMalloc((void **)&d, (size_t)n_lanes * 8);
  if (e == hipSuccess) e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  if (e == hipSuccess) e = (hipError_t)launch_rng_peak(1, pairs_per_lane, n_lanes, d, nullptr);   // warm-up
  if (e == hipSuccess) e = hipEventRecord(a, nullptr);
  for (int r = 0; r < repeats && e == hipSuccess; ++r) e = (hipError_t)launch_rng_peak(2 + r, pairs_per_lane, n_lanes, d, nullptr);
  if (e == hipSuccess) e = hipEventRecord(b, nullptr);
  if (e == hipSuccess) e = hipEventSynchronize(b);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  (void)hipFree(d);
  if (e != hipSuccess) { g_err = hipGetErrorString(e); return SABC_ERR_HIP; }
  *normals_per_s = 2.0 * (double)pairs_per_lane * (double)n_lanes * (double)repeats / ((double)ms * 1e-3);
  return 0;
}

int64_t sabc_host_syncs(const sabc_handle *h) { return h ? h->eng->host_syncs() : 0; }
int64_t sabc_kernel_launches(const sabc_handle *h) { return h ? h->be->kernel_launches() : 0; }
int64_t sabc_collective_calls(const sabc_handle *h) { return h ? h->eng->collective_calls() : 0; }

int sabc_profile_enable(sabc_handle *h, int32_t on) {
  if (!h) return SABC_ERR_STATE;
  h->be->profile_enable(on);
  return 0;
}

int64_t sabc_profile_noops(sabc_handle *h, int32_t kernel) { return h ? h->be->profile_noops(kernel) : 0; }

int sabc_profile_get(sabc_handle *h, int32_t kernel, double *total_ms, int64_t *launches) {
  if (!h) return SABC_ERR_STATE;
  return h->be->profile_get(kernel, total_ms, launches) ? hfail(h, SABC_ERR_HIP) : 0;
}

}  // extern "C"
