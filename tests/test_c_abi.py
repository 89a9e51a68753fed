"""The C-ABI from plain C: tests/c_abi/demo.c is compiled with gcc against include/sabc_hip.h and linked with
libsabc_hip.so -- no Python, no torch in the client.  CPU: it builds and links, and without a device it fails
loudly (no CPU fallback).  GPU: its result equals the oracle's on the same seed."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "simulatedannealingabc.jl_amd")


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    if not shutil.which("gcc"):
        pytest.skip("gcc not found")
    lib = os.path.join(PKG, "libsabc_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    out = str(tmp_path_factory.mktemp("c_abi") / "demo")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "demo.c"), "-o", out, "-L", PKG, "-lsabc_hip", f"-Wl,-rpath,{PKG}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


def test_plain_c_client_builds_and_links(demo):
    r = subprocess.run([demo], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr         # it loaded (the dynamic linker found every symbol) and parsed argv


def test_plain_c_client_fails_loudly_without_a_device(demo):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    r = subprocess.run([demo, "100", "1000", "1", "1.5"], capture_output=True, text=True)
    assert r.returncode == 1 and "sabc_create failed (-20)" in r.stderr      # SABC_ERR_NO_DEVICE: there is no CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("n,updates", [(100, 99), (20_000, 12)])
def test_plain_c_client_matches_oracle(demo, O, gpu, n, updates):
    from tests.cases import SEED, oracle_config, y_obs_mean
    r = subprocess.run([demo, str(n), str((updates + 1) * n), str(SEED), repr(y_obs_mean())], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    f = r.stdout.split()
    run = O.OracleRun(oracle_config(O, "gauss1_cfg2", n))
    run.initialize((updates + 1) * n)
    run.update(O.make_update_args(n_simulation=updates * n, proposal=(O.PROP_RANDOMWALK, 0.8, 0.0), n_particles=n))
    c = run.counters
    assert [int(f[0]), int(f[1]), int(f[2])] == [c["n_accept"], c["n_resampling"], c["n_population_updates"]]
    th = run.theta[0]
    np.testing.assert_allclose([float(f[3]), float(f[4]), float(f[5])], [run.eps[0], th.mean(), th.var()], rtol=1e-9)


@pytest.fixture(scope="module")
def demo_p2p(tmp_path_factory):
    if not shutil.which("gcc"):
        pytest.skip("gcc not found")
    out = str(tmp_path_factory.mktemp("c_abi_p2p") / "demo_p2p")
    cmd = ["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "demo_p2p.c"), "-o", out, "-L", PKG, "-lsabc_hip", f"-Wl,-rpath,{PKG}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


def test_plain_c_p2p_client_builds_and_links(demo_p2p):
    r = subprocess.run([demo_p2p], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prop", [0, 1])
def test_plain_c_p2p_client_matches_the_cpu_engine(demo_p2p, S, gpu, tmp_path, prop):
    """Two processes from plain C (fork before any HIP call, descriptors over a socket pair, hipIpc mapping, no collective
    library): the peer-to-peer transport through nothing but the C-ABI.  Same counters and moments as the CPU engine with
    the same sharding."""
    from tests.cases import SEED, y_obs_mean
    from tests.test_distributed import launch
    n, k = 20_000, 10
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([demo_p2p, str(n), str(k), str(SEED), repr(y_obs_mean()), str(prop)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    f = r.stdout.split()
    ref = launch(2, str(tmp_path / "cpu.npz"), engine="cpu", backend="gloo", case="gauss1_cfg2", alg="single_eps",
                 prop="de" if prop else "rw", n=n, updates=k, resample=n // 4)
    assert [n * (k + 1), int(f[0]), int(f[1]), int(f[2])] == list(ref["counters"]) and int(f[1]) >= 2
    th = ref["theta"][0]
    tol = 1e-6 if prop else 1e-9
    np.testing.assert_allclose([float(f[3]), float(f[4])], [ref["eps"][0], th.mean()], rtol=tol)
    np.testing.assert_allclose(float(f[5]), ((th - th.mean()) ** 2).sum(), rtol=1e-6)
    assert int(f[6]) == 0 and int(f[7]) > 0 and int(f[8]) == 1     # no collective call by the engine; kernels were launched; p2p on


@pytest.mark.gpu
def test_plain_c_p2p_setup_that_fails_on_one_rank_leaves_both_on_the_collectives(demo_p2p, S, gpu, tmp_path):
    """First contact goes wrong on ONE rank (its self-test is told to read a stale line).  sabc_comm_p2p_setup makes both
    ranks agree inside the library -- from plain C, no Python around: both stay on the host's socket collectives, nobody
    waits out the 5 s bound of a first exchange, and the run is the reference run."""
    from tests.cases import SEED, y_obs_mean
    from tests.test_distributed import launch
    n, k = 8000, 6
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([demo_p2p, str(n), str(k), str(SEED), repr(y_obs_mean()), "1", "stale"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    f = r.stdout.split()
    ref = launch(2, str(tmp_path / "cpu.npz"), engine="cpu", backend="gloo", case="gauss1_cfg2", alg="single_eps", prop="de", n=n,
                 updates=k, resample=n // 4)
    assert [n * (k + 1), int(f[0]), int(f[1]), int(f[2])] == list(ref["counters"])
    np.testing.assert_allclose([float(f[3]), float(f[4])], [ref["eps"][0], ref["theta"][0].mean()], rtol=1e-6)
    assert int(f[8]) == 0 and int(f[6]) > k            # not peer to peer; the engine's collectives went over the socket
    assert float(f[9]) < 3.0                           # the set-up did not wait out a bound
