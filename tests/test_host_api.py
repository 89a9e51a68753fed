"""Host logic that needs no GPU: the C-ABI library loads and exports every symbol the header
declares, configuration errors carry the reference's messages, and the product path fails
loudly (no CPU fallback) when no device is usable."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sabc_hip.h")).read()
    return re.findall(r"SABC_API\s+[\w\s\*]+?\b(sabc_\w+)\s*\(", src)


def test_library_exports_every_declared_symbol(S):
    L = C.CDLL(os.path.join(ROOT, "simulatedannealingabc.jl_amd", "libsabc_hip.so"))
    names = declared_symbols()
    assert len(names) >= 37
    for name in names:
        assert hasattr(L, name), f"{name} is declared in include/sabc_hip.h but not exported"
    assert L.sabc_abi_version() == 6


def test_host_side_eps_operators_match_oracle(S, O):
    for ub in [0.5, 0.31, 0.02, 1e-6]:
        assert S.op_eps_single(ub, 1.0) == pytest.approx(O.eps_single(ub, 1.0), rel=1e-13)
    assert S.op_eps_single(1e-17, 1.0) == 0.0
    ub = np.array([0.3, 0.21, 0.45, 0.6])
    np.testing.assert_allclose(S.op_eps_multi(ub, 0.7), O.eps_multi(ub, 0.7), rtol=1e-11)
    np.testing.assert_allclose(S.op_eps_multi([0.5], 1.0), O.eps_multi([0.5], 1.0), rtol=1e-12)
    with pytest.raises(S.SABCError):
        S.op_eps_multi([0.3, 0.0], 1.0)


def test_reference_error_behaviour(S):
    f, prior = S.GaussianIID(), S.Uniform(-10, 10)
    with pytest.raises(S.SABCError, match="too small"):                       # runtests.jl:39-40
        S.sabc(f, prior, n_particles=100, n_simulation=10)
    for kw in (dict(v=-0.1), dict(δ=-0.1)):                                   # runtests.jl:43-54
        with pytest.raises(S.SABCError):
            S.sabc(f, prior, n_particles=100, n_simulation=10, **kw)
    with pytest.raises(S.SABCError, match="algorithm"):                       # :462-464
        S.sabc(f, prior, n_particles=100, n_simulation=1000, algorithm="both_eps")
    with pytest.raises(S.SABCError, match="between zero and one"):            # proposals.jl:30
        S.RandomWalk(β=1.1, n_para=1)
    with pytest.raises(S.SABCError):
        S.RandomWalk(β=-0.1, n_para=1)


def test_proposal_constructors(S):
    with pytest.raises(TypeError):                                            # runtests.jl:204-205 (MethodError)
        S.DifferentialEvolution(1, 0.1)
    with pytest.raises(TypeError):
        S.DifferentialEvolution(1)
    with pytest.raises(ValueError):                                           # runtests.jl:207-208 (ArgumentError)
        S.DifferentialEvolution(γ0=1, n_para=5)
    with pytest.raises(ValueError):
        S.DifferentialEvolution(γ0=1, n_para=5, σ_gamma=1.4)
    assert S.DifferentialEvolution(n_para=2).γ0 == pytest.approx(2.38 / 2)    # proposals.jl:93
    assert S.StretchMove().a == 2.0 and S.RandomWalk(n_para=3).β == 0.8


def test_config_validation_happens_before_the_device_is_touched(S):
    with pytest.raises(ValueError):
        S.sabc(S.Gaussian2D(), S.Normal(0, 1), n_particles=100, n_simulation=1000)     # needs a 2-D prior
    with pytest.raises(TypeError):
        S.sabc("not callable", S.Normal(0, 1), n_particles=100, n_simulation=1000)
    with pytest.raises(S.SABCError) as e:
        S.SabcHandle(n_particles=100, model=S.GandK(n_draws=500), prior=S.product_distribution([S.Uniform(0, 10)] * 4))
    assert e.value.code == -8


def test_no_cpu_fallback(S):
    if S.lib().sabc_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(S.SABCError) as e:
        S.sabc(S.GaussianIID(), S.Uniform(-10, 10), n_particles=100, n_simulation=1000)
    assert e.value.code == -20 and "no CPU path" in str(e.value)
    with pytest.raises(S.SABCError):
        S.op_build_cdf([1.0, 2.0, 3.0])


def test_rng_tables_are_what_the_generator_script_writes(tmp_path, monkeypatch):
    """csrc/rng_tables.inc (the log / sin-cos tables behind the device Box-Muller) is generated, not edited:
    re-running tools/gen_rng_tables.py reproduces the committed file bit for bit, and the table entries satisfy
    their defining identities in binary64."""
    import importlib.util
    import math
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "simulatedannealingabc.jl_amd", "csrc", "rng_tables.inc")
    committed = open(inc).read()
    spec = importlib.util.spec_from_file_location("gen_rng_tables", os.path.join(root, "tools", "gen_rng_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    fake_tools = tmp_path / "tools"
    (tmp_path / "simulatedannealingabc.jl_amd" / "csrc").mkdir(parents=True)
    fake_tools.mkdir()
    monkeypatch.setattr(gen, "__file__", str(fake_tools / "gen_rng_tables.py"))
    gen.main()
    assert open(tmp_path / "simulatedannealingabc.jl_amd" / "csrc" / "rng_tables.inc").read() == committed
    rows = [[float.fromhex(x) for x in re.findall(r"-?0x[0-9a-f.]+p[-+]\d+", ln)] for ln in committed.splitlines() if ln.strip().startswith("{")]
    logt, sct = rows[:128], rows[128:]
    assert len(sct) == 32 and logt[0] == [-2.0, 0.0]
    for i, (m2inv, m2logc) in enumerate(logt):
        c = (1 + i / 128) if i < 53 else (1 + i / 128) / 2
        assert abs(-0.5 * m2inv * c - 1) < 2e-16 and abs(-0.5 * m2logc - math.log(c)) < 3e-16
    for k, (s, c) in enumerate(sct):
        assert abs(s * s + c * c - 1) < 3e-16 and abs(s - math.sin(math.pi * k / 16)) < 1e-15


def test_bench_refuses_to_run_without_a_device():
    """bench.py parses, imports the package and stops with a clear message when there is no GPU (no CPU path to time)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)


def test_star_import_exposes_the_public_names():
    """`from sabc_amd import *` (`from_scipy` was listed in __all__ but never imported)."""
    ns = {}
    exec("from sabc_amd import *", ns)
    import sabc_amd
    for name in sabc_amd.__all__:
        assert name in ns, name
    for name in ("sabc", "update_population_", "from_scipy", "SourcePrior", "HostPrior", "DeviceSource", "RandomWalk"):
        assert callable(ns[name])


def test_p2p_descriptor_size_matches_the_header():
    import re
    import sabc_amd
    hdr = open(sabc_amd._lib.HEADER).read()
    assert int(re.search(r"#define SABC_P2P_DESC_BYTES (\d+)", hdr).group(1)) == sabc_amd._lib.P2P_DESC_BYTES
    assert int(re.search(r"#define SABC_P2P_MAX_WORLD (\d+)", hdr).group(1)) == sabc_amd._lib.P2P_MAX_WORLD
    assert int(re.search(r"#define SABC_MAX_PARA (\d+)", hdr).group(1)) == sabc_amd._lib.MAX_PARA
    assert int(re.search(r"#define SABC_MAX_JOINT_PARA (\d+)", hdr).group(1)) == sabc_amd._lib.MAX_JOINT_PARA


def test_mailbox_words_round_trip(tmp_path):
    """The two 8-byte words the control kernel posts to the host (csrc/sabc_types.hpp: mailbox_pack / mailbox_unpack): every
    field survives, including n_accept beyond 2^32, every SABC_ERR_* a device step can raise, sequence numbers beyond 2^32;
    a word pair whose halves belong to different steps (the host read between the two stores) does not unpack."""
    import subprocess
    src = tmp_path / "mbox.cpp"
    src.write_text(r'''
#include <cstdio>
#include "simulatedannealingabc.jl_amd/csrc/sabc_types.hpp"
using namespace sabc;
int main() {
  const int64_t seqs[] = {1, 2, 7, 4294967295ll, 4294967296ll + 5, 123456789012ll};
  const int64_t accs[] = {0, 1, 4294967295ll, 4294967296ll, 5000000000000ll, (1ll << 54) - 1};
  const int32_t errs[] = {0, SABC_ERR_ZERO_MEAN_U, SABC_ERR_NOT_POSDEF, SABC_ERR_COMM, SABC_ERR_STATE, SABC_ERR_CALLBACK};
  int bad = 0;
  for (int64_t seq : seqs) for (int64_t na : accs) for (int32_t e : errs) for (int32_t h = 0; h < 2; ++h) {
    uint64_t w0, w1, v0, v1;
    mailbox_pack(seq, na, e, h, &w0, &w1);
    int64_t na2 = -1; int32_t e2 = 1, h2 = -1;
    if (!mailbox_unpack(w0, w1, seq, &na2, &e2, &h2) || na2 != na || e2 != e || h2 != h) ++bad;
    if (mailbox_unpack(w0, w1, seq + 1, &na2, &e2, &h2)) ++bad;                 // another step's number
    mailbox_pack(seq + 8, na, e, h, &v0, &v1);                                  // the slot's next occupant (ring of 8)
    if (mailbox_unpack(v0, w1, seq + 8, &na2, &e2, &h2) || mailbox_unpack(w0, v1, seq, &na2, &e2, &h2)) ++bad;   // torn
    if (mailbox_unpack(kMailboxEmpty, kMailboxEmpty, seq, &na2, &e2, &h2) && (uint32_t)seq != 0xFFFFFFFFu) ++bad;
  }
  std::printf("%d\n", bad);
  return bad != 0;
}
''')
    exe = tmp_path / "mbox"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", ROOT, "-o", str(exe), str(src)])
    assert subprocess.check_output([str(exe)], text=True).strip() == "0"


def test_multi_eps_beta_is_a_function_of_mean_u_alone(tmp_path):
    """The root beta_i of the multi-eps schedule (SimulatedAnnealingABC.jl:113; csrc/host_math.hpp: multi_eps_beta) is computed
    from mean u alone -- a fixed start and a fixed number of Newton steps, no hint from the previous update, no data-dependent
    stopping --, so epsilon cannot depend on the call history (a warm and a cold start could end an ulp apart).
    Against a bracketed solve to the last bit it is the root to 1e-12; where e^-beta is below the last bit of 1/beta (mean
    u <= 1/44, the late stage of every chain; expm1 overflows further out) the root is 1 / mean u; the equation holds."""
    import subprocess
    src = tmp_path / "beta.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include "simulatedannealingabc.jl_amd/csrc/host_math.hpp"
using namespace sabc::hostmath;
static double bracketed(double ub) {                                        // bisection on the equation, to the last bit
  if (ub > 0.5) return -bracketed(1.0 - ub);
  double lo = 0.0, hi = 1.0 / ub;
  for (int it = 0; it < 200 && lo < hi; ++it) {
    const double mid = lo + 0.5 * (hi - lo);
    if (mid <= lo || mid >= hi) break;
    if (tilted_mean(mid) - ub > 0.0) lo = mid; else hi = mid;
  }
  return 0.5 * (lo + hi);
}
int main() {
  double worst = 0.0, worst_tail = 0.0, worst_res = 0.0;
  srand(1);
  for (int t = 0; t < 100000; ++t) {
    double ub = pow(10.0, -7.0 * rand() / RAND_MAX) * 0.45;                 // 4.5e-8 .. 0.45
    if (t % 7 == 0) ub = 0.55 + 0.44 * rand() / RAND_MAX;                   // the mirrored branch (negative beta)
    if (t % 11 == 0) ub = 1.0 / 44.0 * (1.0 + 1e-6 * (rand() / (double)RAND_MAX - 0.5));   // around the switch to the closed form
    const double b = multi_eps_beta(ub), ref = bracketed(ub);
    const double d = fabs(b - ref) / fabs(ref);
    if (d > worst) worst = d;
    if (ub < 1e-3) { const double e = fabs(b * ub - 1.0); if (e > worst_tail) worst_tail = e; }
    const double tm = 1.0 / b - 1.0 / expm1(b);                             // the equation of :113, literally
    const double r = fabs(tm - ub) / ub;
    if (r > worst_res) worst_res = r;
  }
  std::printf("%.3g %.3g %.3g\n", worst, worst_tail, worst_res);
  return 0;
}
''')
    exe = tmp_path / "beta"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, "-o", str(exe), str(src)])
    worst, tail, res = (float(x) for x in subprocess.check_output([str(exe)], text=True).split())
    assert worst < 1e-12 and tail < 1e-12 and res < 1e-12, (worst, tail, res)
