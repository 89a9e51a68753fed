"""`python bench.py --gpus N` without torch.distributed.run: bench.py starts its own ranks (before it imports torch or
touches HIP), relays rank 0's JSON line, and leaves no process behind -- whether the ranks succeed, fail or hang."""
import json
import os
import subprocess
import sys
import time

import psutil
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _ranks_alive(marker):
    out = []
    for p in psutil.process_iter(["cmdline"]):
        try:
            if marker in (p.info["cmdline"] or []):
                out.append(p.pid)
        except (psutil.NoSuchProcess, psutil.AccessDenied):
            pass
    return out


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_launcher_without_a_device_exits_cleanly():
    """No GPU here: every rank refuses (the engine has no CPU path); the launcher reports it and exits non-zero."""
    import torch
    if torch.cuda.is_available():            # (checked first: with a GPU the command below would be a full two-rank bench)
        pytest.skip("a GPU is visible: covered by test_launcher_two_ranks_on_one_gpu")
    marker = "31337"
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0",
                        "--cpu-updates", marker], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode not in (0, 124), (r.returncode, r.stderr[-2000:])
    assert "needs an MI355X" in r.stderr
    assert not r.stdout.strip().startswith("{")
    assert time.time() - t0 < 300
    assert _ranks_alive(marker) == []


def test_launcher_kills_ranks_at_the_deadline():
    """Ranks still running at the deadline are killed by exact PID and the exit code is 124."""
    marker = "31338"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--dist-backend", "gloo", "--launch-timeout", "0.05",
                        "--cpu-updates", marker], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert "killing them" in r.stderr
    time.sleep(0.2)
    assert _ranks_alive(marker) == []


def test_a_rank_is_not_relaunched():
    """With WORLD_SIZE set (torch.distributed.run, or our own launcher) bench.py is a rank: a mismatch with --gpus is an
    error, not another launch."""
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_launcher_two_ranks_on_one_gpu(gpu):
    """The gloo rehearsal of the N > 1 path on a one-GPU box: one JSON line with n_gpus = 2."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--steps", "4", "--warmup", "2",
                        "--n-particles", "200000", "--no-cpu-baseline"], env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0
    assert out["config"]["collectives"] == "hooks-gloo"
    assert out["comm_bytes_per_step"] >= 5 * 8


@pytest.mark.gpu
def test_launcher_two_ranks_peer_to_peer(gpu):
    """The same with the peer-to-peer transport on top of the gloo hooks (two processes mapping each other's memory with
    hipIpc): the line says so, counts no collective call in the timed region and one reduce-exchange-control launch per update
    -- and VALIDATES ITSELF: the accept and resample counts equal the oracle's for the same seed and calls
    (`posterior_vs_cpu`, shard 0's moments against the oracle's same slice) and the one-shard run's on the same GPU
    (`n1_equivalent`); after the timed region every rank leaves the peer-to-peer group and the collectives underneath get
    their own numbers (`exchange_rccl`)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--p2p", "on", "--steps", "6", "--warmup", "2",
                        "--n-particles", "200000", "--repeats", "2"], env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["repeats"] == 2 and out["value_min"] <= out["value"] <= out["value_max"]
    assert out["config"]["collectives"] == "p2p" and out["config"]["collectives_fallback"] == "hooks-gloo"
    assert out["collective_calls_per_update"] == 0 and not out["transport_degraded"]
    assert out["p2p_fallbacks"] == 0 and out["p2p_active_at_end"]
    ex = out["exchange"]
    assert ex["collective_calls_per_update"] == 0 and ex["reduce_control_launches"] >= 10 and ex["collectives_timed"] == 0
    assert 2.0 <= ex["launches_per_update"] <= 4.0            # k_update + ONE reduce-exchange-control launch (+ a resample's share)
    pv = out["posterior_vs_cpu"]
    assert pv["n_accept_equal"] and pv["n_resampling_equal"] and pv["updates"] == 8
    assert pv["rel_err_mean"] < 1e-9 and pv["rel_err_var"] < 1e-8 and pv["eps_rel_err"] < 1e-9
    n1 = out["n1_equivalent"]
    assert n1["n_accept_equal"] and n1["n_resampling_equal"] and n1["eps_rel_err"] < 1e-9 and n1["n_accept"] == out["state"]["n_accept"]
    xr = out["exchange_rccl"]
    assert xr["transport"] == "hooks-gloo" and not xr["p2p_active"]
    assert xr["collective_calls_per_update"] >= 1.0 and xr["allreduces_timed"] >= 10 and xr["allreduce_us"] > 0
    assert xr["launches_per_update"] >= 3.0                   # k_update + k_reduce_partials + k_control (+ resamples)
