#!/usr/bin/env python3
"""bench.py -- particle-simulations/sec of the SABC population update loop on MI355X.

One "step" = one population update (SimulatedAnnealingABC.jl:294-375): every particle is
proposed, simulated, ECDF-transformed and MH-accepted once, then the fused sums give the new
epsilon / proposal covariance (and a resample when it triggers).  Workload = BASELINE.json
configs[1]: 1-D Gaussian-mean ABC, n_particles = 1e6, f_dist as a device-side kernel,
RandomWalk proposal.  The population is resident in HBM before the timed region; the timed
region is ONE call of the C-ABI entry point sabc_update() for K population updates.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   (one rank per GPU)

For N > 1 the n_particles = 1e6 population is sharded over the ranks (strong scaling, the
north-star target is quoted at n_particles = 1e6 on 8 GPUs); --particles-per-gpu switches to weak.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SEED = 20241220
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def observed_mean():
    """y_obs: 100 draws of N(1.5, 1) with the fixed seed; only its mean is used."""
    return float(np.random.default_rng(SEED).normal(1.5, 1.0, 100).mean())


def cpu_baseline(n, updates, threads, warmup=0, steps=0, proposal="randomwalk", shard0=None):
    """The oracle (CPU restatement, NOT the Julia reference: Julia is not installed) on the host
    cores of this box, same workload, bounded sample; update loop only, like `value`.
    The first `warmup` + `steps` updates are issued exactly like the GPU's (same seed, same two calls), so that the
    posterior moments of the two runs can be compared (BASELINE metric: "posterior-mean L2 vs ref"); shard0 = number of
    particles of the first shard of a sharded run: its moments are those of the oracle's same slice (global ids 0 ..)."""
    from oracle import oracle as O
    O.set_threads(threads)
    cfg = O.make_config(n_particles=n, n_para=1, n_stats=1, model_id=O.MODEL_GAUSS_IID,
                        model_params=[100, 1.0, observed_mean(), 0.0], prior=[(O.PRIOR_NORMAL, 0.0, 2.0)], seed=SEED)
    run = O.OracleRun(cfg)
    run.initialize(n)
    prop = {"randomwalk": (O.PROP_RANDOMWALK, 0.8, 0.0), "de": (O.PROP_DIFFEVO, None, 1e-5), "stretch": (O.PROP_STRETCH, 2.0, 0.0)}[proposal]
    args = lambda k: O.make_update_args(n_simulation=k * n, proposal=prop, n_para=1, n_particles=n)
    t0 = time.perf_counter()
    same = None
    done = 0
    for k in (warmup, steps):
        if k > 0 and done + k <= updates:
            run.update(args(k))
            done += k
    if done == warmup + steps and done > 0:
        th = run.theta[0]
        same = {"mean": float(th.mean()), "var": float(th.var()), "n_accept": int(run.counters["n_accept"]), "updates": done,
                "n_resampling": int(run.counters["n_resampling"]), "eps": [float(x) for x in np.atleast_1d(run.eps)]}
        if shard0:
            same.update(shard0_mean=float(th[:shard0].mean()), shard0_var=float(th[:shard0].var()))
    if updates > done:
        run.update(args(updates - done))
    dt = time.perf_counter() - t0
    out = {"value": updates * n / dt, "unit": "particle-simulations/s", "cores": threads, "kind": "port",
           "sample": f"oracle/sabc_oracle.c (OpenMP), cfg2 n_particles={n}, {updates} population updates, "
                     f"{dt:.1f} s; CPU restatement, not the Julia reference; {threads} threads = the CPU share of a one-GPU box "
                     f"of this pool ({os.cpu_count()} logical CPUs visible; SABC_CPU_THREADS overrides)"}
    return out, same


def load_traffic(workload_n, config="cfg2"):
    if config != "cfg2":
        return None
    return _load_traffic(workload_n)


def _load_traffic(workload_n):
    """HBM bytes per launch of the update kernel from a committed rocprofv3 --pmc pass, if any."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
        if int(t.get("n_particles", -1)) == int(workload_n):
            return t.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def load_from_profiles(workload_n, config="cfg2"):
    """Numbers that do NOT come from this run: read from the rocprofv3 passes committed under profiles/ (same command, an
    earlier box).  Kept under one key of the JSON line so that they cannot be mistaken for live measurements."""
    out = {}
    if int(workload_n) != 1_000_000:
        return out
    try:
        import csv
        import subprocess
        for tag in ("r04", "r03", "r02"):
            name = f"{tag}_pmc_sq_{config}.csv"
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                break
        rows = {r["name"]: float(r["value"]) for r in csv.DictReader(open(path))}
        out.update({"pmc_valu_busy_fraction": rows["valu_busy_fraction"], "pmc_effective_clock_ghz": rows["effective_clock_ghz"],
                    "pmc_cycles_per_valu_instruction": rows["cycles_per_valu_instruction"],
                    "pmc_wave_cycles_parked": rows["wave_cycles_parked_waitcnt_barrier"], "source": f"profiles/{name}",
                    "note": "rocprofv3 --pmc SQ_* passes of the same command, committed; NOT measured in this run"})
        try:
            out["commit"] = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", path], capture_output=True,
                                           text=True, timeout=5).stdout.strip() or None
        except Exception:
            out["commit"] = None
    except Exception:
        pass
    t = load_traffic(workload_n, config)
    if t is not None:
        out["hbm_bytes_per_launch"] = t
        out["traffic_source"] = "profiles/pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes)"
    return out


def bench_host_path(args):
    """`--config host`: the path every model that is not device code takes -- f_dist as a host callable
    (SimulatedAnnealingABC.jl:315).  Per population update: callback_us = time inside the caller's
    function(s), library_us = everything else (proposal and accept kernels, zero-copy staging over PCIe, fused sums,
    control step, the mailbox wait).  Rows: BASELINE config 2's simulator as a vectorised NumPy callable at n = 100 (the
    reference's default), 5 000 (the size of its documentation example, docs/src/example.md:190-198) and 1e6; and the
    documentation's Gillespie SIR model itself (one Python call per particle) at n = 5 000."""
    import torch
    import sabc_amd as S
    from sabc_amd.examples import gaussian_mean_batched, sir_gillespie
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the SABC engine has no CPU path")
    rows = []

    def run(label, model, prior, n, proposal, warm, steps, reps=3):
        h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=SEED)
        h.initialize(n)
        if warm:
            h.update(n_simulation=warm * n, proposal=proposal)
        torch.cuda.synchronize()
        # three timed calls of `steps` updates each, the median one reported (a single hiccup of the host -- a page fault, the
        # garbage collector inside the Python callback -- is tens of milliseconds, i.e. everything at n = 100)
        samples = []
        for _ in range(reps):
            cb0, calls0, l0 = h.host_callback_seconds, h.host_callback_calls, h.kernel_launches
            t0 = time.perf_counter()
            h.update(n_simulation=steps * n, proposal=proposal)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            samples.append((dt, h.host_callback_seconds - cb0, h.host_callback_calls - calls0, h.kernel_launches - l0))
        samples.sort(key=lambda x: x[0] - x[1])
        dt, cb, calls, launches = samples[len(samples) // 2]
        calls0, l0 = h.host_callback_calls - calls, h.kernel_launches - launches
        rows.append({"model": label, "n_particles": n, "proposal": type(proposal).__name__, "updates": steps, "timed_calls": reps,
                     "per_update_us": dt / steps * 1e6, "callback_us": cb / steps * 1e6, "library_us": (dt - cb) / steps * 1e6,
                     "library_over_callback": (dt - cb) / cb if cb > 0 else None,
                     "f_dist_calls_per_update": (h.host_callback_calls - calls0) / steps,
                     "kernel_launches_per_update": (h.kernel_launches - l0) / steps,
                     "particle_sims_per_s": steps * n / dt})
        h.close()

    yb = observed_mean()
    for n, warm, steps in ((100, 20, 200), (5000, 10, 100), (1_000_000, 2, 6)):
        fn = gaussian_mean_batched(yb, 100, seed=1)
        model = S.HostDistance(fn, n_stats=1, n_para=1, univariate=True, batched=True)
        for prop in (S.RandomWalk(n_para=1), S.DifferentialEvolution(n_para=1)):
            run("cfg2 simulator, vectorised NumPy", model, S.Normal(0.0, 2.0), n, prop, warm, steps)
    simulate, f_sir = sir_gillespie(seed=11)
    data = simulate(0.6, 0.15)
    model = S.HostDistance(f_sir, n_stats=1, n_para=2, univariate=False, args=(data,))
    prior = S.product_distribution([S.Uniform(0.1, 1), S.Uniform(0.05, 0.5)])
    run("docs SIR (Gillespie), one Python call per particle", model, prior, 5000, S.DifferentialEvolution(n_para=2), 0, args.host_sir_updates, reps=1)
    out = {"metric": "host-callback f_dist path: library microseconds per population update", "unit": "us", "n_gpus": 1,
           "higher_is_better": False, "data": "synthetic", "dtype": "f64",
           "config": {"workload": "f_dist as a host callable (SABC_MODEL_HOST): proposal / ECDF / accept / sums on the device, "
                                  "simulator in the caller's Python function"},
           "value": rows[0]["library_us"], "rows": rows,
           "note": "value = library_us at n = 100 (RandomWalk); not the headline metric -- `python bench.py` without --config is"}
    print(json.dumps(out), flush=True)


def launch_ranks(n, argv, timeout_s):
    """Start `n` fresh rank processes of this script (one per GPU, LOCAL_RANK = RANK = 0..n-1, rendezvous on 127.0.0.1),
    relay rank 0's JSON line, and return the exit code.  A rank that fails takes the others down; ranks still running at
    the deadline are killed (exact PIDs) and the code is 124.  The parent never imports torch or touches HIP."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # dmabuf IPC: this pool's host driver supports nothing else (the image exports the variable already; hipIpcGetMemHandle
        # fails with "invalid argument" without it -- tests/test_p2p.py::test_p2p_two_processes_over_hip_ipc maps a peer's memory
        # with it on the GPU box, profiles/README.md has the run with it switched off)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "2")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    deadline = time.monotonic() + timeout_s
    code = 0
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [c for c in states if c not in (None, 0)]
            if bad:
                code = bad[0] if bad[0] > 0 else 1
                print(f"[bench] a rank exited with code {bad[0]}; stopping the others", file=sys.stderr, flush=True)
                break
            if all(c == 0 for c in states):
                break
            if time.monotonic() > deadline:
                code = 124
                print(f"[bench] ranks still running after {timeout_s:.0f} s; killing them", file=sys.stderr, flush=True)
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        out0 = ""
        for r, p in enumerate(procs):
            try:
                o, _ = p.communicate(timeout=30)
            except Exception:
                o = ""
            if r == 0:
                out0 = o or ""
    if code == 0:
        lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
        if not lines:
            print("[bench] rank 0 printed no JSON line", file=sys.stderr, flush=True)
            return 1
        print(lines[-1], flush=True)
    elif out0.strip():
        print(out0, file=sys.stderr, flush=True)
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-particles", type=int, default=1_000_000)
    ap.add_argument("--particles-per-gpu", type=int, default=0, help="weak scaling: this many particles per rank")
    ap.add_argument("--proposal", default="randomwalk", choices=["randomwalk", "de", "stretch"])
    ap.add_argument("--host-sir-updates", type=int, default=2, help="--config host: population updates of the Gillespie SIR row (~2 s each)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5", "host"],
                    help="cfg2 is the headline workload (BASELINE configs[1]); the others are secondary measurements")
    ap.add_argument("--algorithm", default="single_eps", choices=["single_eps", "multi_eps"])
    ap.add_argument("--n-obs", type=int, default=100, help="cfg2 only: draws per simulation (100 is the BASELINE workload)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the product path; gloo lets several ranks share one GPU to rehearse the N > 1 code path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions per process; `value` is their median")
    ap.add_argument("--preheat-seconds", type=float, default=1.0,
                    help="untimed launches of the bare generator loop (k_rng_peak) before the first timed region, until this much wall time has "
                         "passed (the GPU's clocks ramp over the first ~100 ms of load: five cold regions of 4 ms each measured 4.65 -> 5.46e9 in run order)")
    ap.add_argument("--p2p", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: the peer-to-peer transport on top of the collectives (auto: on for --dist-backend nccl)")
    ap.add_argument("--all-kernel-events", action="store_true", help="bracket every kernel, not only k_update (adds ~15 us/step)")
    ap.add_argument("--time-every-launch", action="store_true", help="HIP events on every k_update launch instead of every second one")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events (A/B of the measurement overhead)")
    ap.add_argument("--cpu-updates", type=int, default=100, help="population updates of the CPU baseline sample (~15 s on 16 cores)")
    ap.add_argument("--allow-hooks", action="store_true",
                    help="accepted for round-1 command lines; falling back to hooks-nccl no longer stops the run (see --strict-transport)")
    ap.add_argument("--strict-transport", action="store_true",
                    help="exit 3 also when RCCL could not be bound inside the library and the collectives fell back to "
                         "torch.distributed's RCCL on device pointers ('hooks-nccl': same wires, Python in the per-update path)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="--gpus N > 1 without a launcher: deadline for the ranks (s)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without torch.distributed.run: start the N ranks ourselves -- BEFORE anything in this
    # process imports torch or touches HIP (a process that has initialised the GPU must not fork/exec workers)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))

    if args.config == "host":
        return bench_host_path(args)

    import torch
    import torch.distributed as dist
    import sabc_amd as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or let bench.py start them: "
                         f"run it without a launcher)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the SABC engine has no CPU path")
    device = local_rank % torch.cuda.device_count()      # == local_rank on a node with one GPU per rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    weak = args.particles_per_gpu > 0
    n = args.particles_per_gpu * world if weak else args.n_particles
    K, W = args.steps, args.warmup
    if args.config == "cfg2":
        model, prior = S.GaussianIID(n_obs=args.n_obs, sd=1.0, obs_mean=observed_mean()), S.Normal(0.0, 2.0)
        normals_per_sim = args.n_obs
    elif args.config == "cfg3":       # 2-D correlated Gaussian, 3 statistics
        model = S.Gaussian2D(n_obs=50, r=0.6, obs_mean=(1.2, -0.7), obs_varsum=2.1, obs_cov=0.55)
        prior = S.product_distribution([S.Normal(0, 3), S.Normal(0, 3)])
        normals_per_sim = 100
    elif args.config == "cfg4":       # g-and-k, 4 order statistics; truth (3, 1, 2, 0.5)
        model = S.GandK(n_draws=128, c=0.8, ranks=(16, 48, 80, 112), obs=(1.9, 2.7, 3.6, 6.4))
        prior = S.product_distribution([S.Uniform(0, 10)] * 4)
        normals_per_sim = 128
    else:                             # stochastic Lotka-Volterra, 256 Euler-Maruyama steps
        model = S.LotkaVolterra(n_steps=256, dt=0.05, σ=0.1, x0=50.0, y0=50.0, obs=(18.0, 17.0, 14.0, 12.0))
        prior = S.product_distribution([S.Uniform(0, 2), S.Uniform(0, 0.1), S.Uniform(0, 2)])
        normals_per_sim = 512
    d, s = len(prior), model.n_stats
    proposal = {"randomwalk": S.RandomWalk(n_para=d), "de": S.DifferentialEvolution(n_para=d),
                "stretch": S.StretchMove()}[args.proposal]
    alg = S._lib.ALG_MULTI_EPS if args.algorithm == "multi_eps" else S._lib.ALG_SINGLE_EPS

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region is repeated R times in this process -- the population initialised again (sabc_initialize on the
    # same handle: same seed, the very same trajectory as a fresh handle's), the same warm-up, the same K updates -- and
    # `value` is the MEDIAN: one 4 ms sample says little (boxes and runs differ by a few per cent).  steps / warmup mean what
    # they always meant.  The handle and its transport are set up ONCE: N ranks that map each other's memory (and an RCCL
    # communicator each) are not torn down and rebuilt between the samples.
    samples = []
    trace = os.environ.get("SABC_BENCH_TRACE")

    def say(what):
        if trace:
            print(f"[bench trace] rank {rank} {what}", file=sys.stderr, flush=True)

    h = S.SabcHandle(n_particles=n, model=model, prior=prior, algorithm=alg, seed=SEED, device=device, rank=rank, world=world)
    transport, fallback = "none", None
    preheat = {"seconds": 0.0, "kernel": "k_rng_peak (1e6 lanes x 50 Philox + Box-Muller pairs)", "launches": 0}
    for rep in range(max(args.repeats, 1)):
        say(f"repeat {rep}")
        if world > 1 and rep == 0:
            from sabc_amd.dist import install_collectives
            transport = install_collectives(h, device, p2p=None if args.p2p == "auto" else args.p2p == "on")
            fallback = getattr(h, "fallback_transport", None)
            # every rank takes the same branch (install_collectives agrees on the transport across ranks)
            # (--dist-backend gloo is an explicit request for the host-staged rehearsal transport and says so in the line)
            degraded = args.dist_backend == "nccl" and transport not in ("rccl", "p2p")
            if degraded and args.strict_transport:
                print(f"[bench] rank {rank}: RCCL could not be bound inside the library, the collectives fell back to "
                      f"'{transport}': --strict-transport refuses to measure that", file=sys.stderr, flush=True)
                h.close()
                dist.destroy_process_group()
                raise SystemExit(3)
            if degraded:
                print(f"[bench] WARNING rank {rank}: RCCL could not be bound inside the library; measuring over '{transport}' "
                      "(torch.distributed on device pointers, a Python callback per collective): the line says so in "
                      "config.collectives and transport_degraded", file=sys.stderr, flush=True)
        if rep == 0 and args.preheat_seconds > 0:
            # bring the device to its sustained clocks before anything is timed: the bare Philox + Box-Muller loop (k_rng_peak,
            # the instruction mix of the simulators) for about a second.  Not the workload's own kernels: the chain's state is not
            # advanced, and a rocprofv3 pass of this command averages k_update over warm-up and timed regions only.
            say("pre-heat")
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < args.preheat_seconds:
                S.op_rng_peak(n_lanes=1_000_000, pairs_per_lane=50, repeats=20, device=device)
                preheat["launches"] += 20
            preheat["seconds"] = time.perf_counter() - t_pre
        say(f"repeat {rep}: transport {transport}, initialize")
        t_init0 = time.perf_counter()
        h.initialize(n)
        torch.cuda.synchronize()
        t_init = time.perf_counter() - t_init0
        say(f"repeat {rep}: warm-up")
        if W > 0:
            h.update(n_simulation=W * n, proposal=proposal)
        say(f"repeat {rep}: timed region")
        h.profile_enable(0 if args.no_kernel_events else (2 if args.all_kernel_events else (3 if args.time_every_launch else 1)))
        barrier()
        syncs0, comm0, launches0, coll0 = h.host_syncs, h.comm_bytes, h.kernel_launches, h.collective_calls
        resampling0 = h.counters["n_resampling"]
        t0 = time.perf_counter()
        h.update(n_simulation=K * n, proposal=proposal)          # exactly K population updates
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        say(f"repeat {rep}: done, {dt / K * 1e6:.1f} us per update")
        kern_ms, launches = h.profile_get(S._lib.KERNEL_UPDATE)
        samples.append({"dt": dt, "kern_ms": kern_ms, "launches": launches, "noops": h.profile_noops(S._lib.KERNEL_UPDATE),
                        "syncs": h.host_syncs - syncs0, "comm": h.comm_bytes - comm0,
                        "kernel_launches": h.kernel_launches - launches0, "collective_calls": h.collective_calls - coll0,
                        "red": h.profile_get(S._lib.KERNEL_REDUCE), "res": h.profile_get(S._lib.KERNEL_RESAMPLE), "t_init": t_init})
    order = sorted(range(len(samples)), key=lambda i: samples[i]["dt"])
    med = samples[order[len(order) // 2]]
    dt, kern_ms, launches = med["dt"], med["kern_ms"], med["launches"]
    syncs, comm_bytes, t_init = med["syncs"], med["comm"], med["t_init"]
    red_ms, red_n = med["red"]
    res_ms, res_n = med["res"]
    c = h.counters
    th, _, _ = h.get_population(u=False, rho=False)
    state_after_timed = (dict(c), h.eps.copy())          # the last timed region's: the extra updates below move on
    # N > 1: what one update costs between the update kernels, measured on 10 MORE updates with every kernel bracketed
    # (level 2 adds ~4 us per bracket, so this is not part of the timed region): the reduce[-exchange]-control launch and,
    # on the collectives transport, the allreduce between k_reduce_partials and k_control
    exchange = None
    p2p_active_after_timed = bool(h.p2p_active) if world > 1 else False
    if world > 1 and not args.no_kernel_events:
        say("extra updates with every kernel bracketed")
        h.profile_enable(2)
        l0, c0 = h.kernel_launches, h.collective_calls
        h.update(n_simulation=10 * n, proposal=proposal)
        r_ms, r_n = h.profile_get(S._lib.KERNEL_REDUCE)
        x_ms, x_n = h.profile_get(S._lib.KERNEL_COLLECTIVE)
        exchange = {"reduce_control_us": r_ms / max(r_n, 1) * 1e3, "reduce_control_launches": r_n,
                    "collective_us": (x_ms / x_n * 1e3) if x_n else None, "collectives_timed": x_n,
                    "launches_per_update": (h.kernel_launches - l0) / 10.0, "collective_calls_per_update": (h.collective_calls - c0) / 10.0,
                    "note": "10 extra updates after the timed region, every kernel bracketed by HIP events (not part of `value`)"}

    # ... and the same 10 bracketed updates over the COLLECTIVES underneath (north star: "RCCL allreduce over xGMI for the global
    # acceptance count, epsilon-schedule update and population covariance"): every rank leaves the peer-to-peer group -- no
    # barrier needed, a rank that is late finds out at the entry of its call (include/sabc_hip.h, LEAVING) -- and the run goes
    # on over what install_collectives put underneath.  Otherwise the collectives never get a number on a node where the
    # peer-to-peer transport comes up.
    exchange_rccl = None
    if world > 1 and transport == "p2p" and not args.no_kernel_events:
        say("leaving the peer-to-peer group: 10 bracketed updates over the collectives underneath")
        try:                                    # (a secondary measurement must not cost the run its line)
            h.p2p_disable()
            h.profile_enable(2)
            l0, c0 = h.kernel_launches, h.collective_calls
            t_x = time.perf_counter()
            h.update(n_simulation=10 * n, proposal=proposal)
            torch.cuda.synchronize()
            t_x = time.perf_counter() - t_x
            r_ms, r_n = h.profile_get(S._lib.KERNEL_REDUCE)
            x_ms, x_n = h.profile_get(S._lib.KERNEL_COLLECTIVE)
        except Exception as e:
            exchange_rccl = {"transport": fallback, "error": repr(e)}
            r_n = None
    if exchange_rccl is None and world > 1 and transport == "p2p" and not args.no_kernel_events:
        exchange_rccl = {"transport": fallback, "allreduce_us": (x_ms / x_n * 1e3) if x_n else None, "allreduces_timed": x_n,
                         "reduce_plus_control_us": r_ms / max(r_n, 1) * 1e3, "launches_per_update": (h.kernel_launches - l0) / 10.0,
                         "collective_calls_per_update": (h.collective_calls - c0) / 10.0, "us_per_update": t_x / 10.0 * 1e6,
                         "p2p_active": bool(h.p2p_active),
                         "note": "10 more updates after every rank left the peer-to-peer group (sabc_comm_p2p_disable): k_update -> "
                                 "k_reduce_partials -> allreduce -> k_control, resamples over allgathers; every kernel bracketed by "
                                 "HIP events (not part of `value`)"}
    # N > 1: the same seed and calls on ONE shard of this rank's GPU.  The sharded run is shard-count independent by design
    # (Philox streams keyed by global particle id, rank-order sums): its accept count has to be this one's
    n1 = None
    if world > 1 and rank == 0:
        say("n1_equivalent: the same calls on one shard")
        try:
            h1 = S.SabcHandle(n_particles=n, model=model, prior=prior, algorithm=alg, seed=SEED, device=device)
            h1.initialize(n)
            if W > 0:
                h1.update(n_simulation=W * n, proposal=proposal)
            h1.update(n_simulation=K * n, proposal=proposal)
            c1 = h1.counters
            e1 = h1.eps
        except Exception as e:
            n1, c1 = {"error": repr(e)}, None
    if world > 1 and rank == 0 and n1 is None:
        n1 = {"expected_equal": args.proposal == "randomwalk",
              "n_accept": c1["n_accept"], "n_resampling": c1["n_resampling"],
              "n_accept_equal": c1["n_accept"] == state_after_timed[0]["n_accept"],
              "n_resampling_equal": c1["n_resampling"] == state_after_timed[0]["n_resampling"],
              "eps_rel_err": float(np.max(np.abs(state_after_timed[1] / e1 - 1.0))),
              "note": "one shard on rank 0's GPU, same seed, same initialize + warm-up + K updates as the sharded timed region"
                      + ("" if args.proposal == "randomwalk" else
                         "; DifferentialEvolution / StretchMove take their partners from the OTHER colour of a two-colouring that is local "
                         "to each shard (DESIGN.md section 5): a sharded run is a different, equally valid chain than the one-shard run "
                         "-- agreement is statistical, not particle for particle")}
        h1.close()

    if rank == 0:
        bytes_per_sim = 8 * (2 * d + 3 * s)                 # 40 B for d = s = 1
        # launches of the update kernel in the timed region: K (RandomWalk: one per update) or 2K; every second one carries
        # timing events (sabc_profile_enable level 1), `launches` of them came back
        real_launches = K if args.proposal == "randomwalk" else 2 * K
        dts = sorted(x["dt"] for x in samples)
        if h.persistent_launches > 0:
            # a small shard: the updates of the call ran in ONE launch (k_update_persistent) -- there is no per-update kernel
            # duration to price against a roofline; the line carries the step time only
            launches = 0
        avg_launch_s = (kern_ms / launches) * 1e-3 if launches else float("nan")
        sims_per_launch = h.n_local if args.proposal == "randomwalk" else h.n_local / 2
        achieved = bytes_per_sim * sims_per_launch / avg_launch_s / 1e9 if launches else float("nan")
        valu = None
        if launches and world == 1:
            peak = S.op_rng_peak(n_lanes=int(sims_per_launch), pairs_per_lane=max(normals_per_sim // 2, 1), repeats=10, device=device)
            in_kernel = normals_per_sim * sims_per_launch / avg_launch_s
            valu = {"bound": "valu", "achieved": in_kernel, "peak": peak, "unit": "normals/s", "frac": in_kernel / peak,
                    "note": "live: peak = rate of k_rng_peak (generator only) measured in this run; the rest of k_update is "
                            "proposal, ECDF search, accept, sums"}
        yb = observed_mean()
        post_var = 1.0 / (1.0 / 4.0 + 100.0)
        analytic = {"analytic_posterior_mean": post_var * 100.0 * yb, "analytic_posterior_var": post_var,
                    "analytic_note": "the mean is the anchor; the population VARIANCE of this algorithm passes through the analytic value "
                                     "around update 60-90 and settles ~20 % (RandomWalk) below it -- also in an independent NumPy restatement "
                                     "of the reference (DESIGN.md section 7)"} \
            if args.config == "cfg2" else {}
        out = {
            "metric": "particle-simulations/sec at n_particles=1e6" if args.config == "cfg2" else f"particle-simulations/sec ({args.config})",
            "value": K * n / dt,
            "value_min": K * n / dts[-1], "value_max": K * n / dts[0], "repeats": len(samples),
            "preheat": dict(preheat, note="untimed launches of the bare generator loop before the first timed region (the device's clocks ramp over the first ~100 ms of load); --preheat-seconds 0 turns it off"),
            "values_in_run_order": [K * n / x["dt"] for x in samples],
            "kernel_us_in_run_order": [(x["kern_ms"] / x["launches"] * 1e3) if x["launches"] else None for x in samples],
            "value_note": "median of `repeats` timed regions in this process (population initialised again, same warm-up, same K updates each)",
            "unit": "particle-simulations/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[1]: 1-D Gaussian-mean ABC (Normal(0,2) prior, |mean(x)-mean(y_obs)|, "
                             f"100 draws per simulation), n_particles={n}, proposal={args.proposal}, {args.algorithm}")
                if args.config == "cfg2" else f"{args.config} ({type(model).__name__}, d={d}, s={s}), n_particles={n}, "
                                              f"proposal={args.proposal}, {args.algorithm}",
                "n_particles": n, "proposal": args.proposal, "algorithm": args.algorithm,
                "particles_per_gpu": h.n_local, "seed": SEED, "collectives": transport, "collectives_fallback": fallback,
            },
            # True: the library's own RCCL binding failed and torch.distributed's RCCL carried the collectives (device
            # pointers, same wires, one Python callback per collective) -- a measured but pessimistic number
            "transport_degraded": bool(world > 1 and args.dist_backend == "nccl" and transport not in ("rccl", "p2p")),
            # peer-to-peer transport: calls in which a wait ran into its bound and that the engine finished over the
            # collectives underneath (0 = the whole run went peer to peer), and whether it is still on at the end
            "p2p_fallbacks": h.p2p_fallbacks if world > 1 else 0,
            "p2p_active_at_end": p2p_active_after_timed,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if launches else None,
                # HBM bytes per launch from the committed FETCH_SIZE / WRITE_SIZE passes (profiles/, a 1e6-particle launch;
                # NOT collected in this run -- see from_profiles)
                "traffic": load_traffic(n, args.config) if world == 1 else None,
                "kernel": f"k_update<{type(model).__name__},{d},{s},{args.proposal}>",
                "avg_launch_us": avg_launch_s * 1e6 if launches else None,
                "launches": real_launches, "timed_launches": launches, "noop_launches_excluded": med["noops"],
                "algorithmic_bytes_per_sim": bytes_per_sim,
                "note": f"not HBM-bound by construction: {normals_per_sim} f64 normals ({normals_per_sim // 2} Philox4x32-10 "
                        f"blocks + Box-Muller log/sqrt/sincos) per {bytes_per_sim} algorithmic bytes; the binding resource is VALU issue "
                        f"(valu_roofline: SQ counters show the VALU executing in ~95 % of the busy cycles); see normals_per_s",
            },
            "normals_per_s": float(normals_per_sim) * K * n / dt,
            # the bound that actually binds: k_update's in-kernel normal rate against the bare Philox + Box-Muller
            # loop measured on this GPU right now (same lane count, same pairs per lane, nothing else in the loop)
            "valu_roofline": valu,
            # bytes landing in one shard's receive buffers per population update (allreduce of the fused sums; DE / Stretch:
            # two allgathers of the inactive halves; on resamples the weight row and the rows the shard drew)
            "comm_bytes_per_step": comm_bytes / K,
            # kernels the library launched / collective calls it issued per population update in the timed region
            # (resamples included), and -- N > 1 -- what the step between two update kernels costs
            "launches_per_update": med["kernel_launches"] / K,
            "persistent_launches": h.persistent_launches,      # > 0: small shard, the updates of a call in one launch
            "persistent_lanes": h.persistent_lanes,            # ... with 16 | 4 | 1 lanes per particle (a team shares the generator's blocks)
            "collective_calls_per_update": med["collective_calls"] / K,
            "exchange": exchange,
            "exchange_rccl": exchange_rccl,
            "n1_equivalent": n1,
            "from_profiles": load_from_profiles(n, args.config) if world == 1 else {},
            "kernel_time_frac": (avg_launch_s * real_launches) / dt if launches else None,
            "reduce_us_per_step": red_ms / max(red_n, 1) * 1e3,
            "resamples_in_timed_region": c["n_resampling"] - resampling0,
            "host_syncs_in_timed_region": syncs,
            "init_s": t_init,
            "state": {"n_accept": c["n_accept"], "n_resampling": c["n_resampling"],
                      "n_population_updates": c["n_population_updates"], "eps": state_after_timed[1].tolist(),
                      "shard0_mean": th.mean(1).tolist(), "shard0_var": th.var(1).tolist(), **analytic},
        }
        comparable = args.config == "cfg2" and args.algorithm == "single_eps" and args.n_obs == 100
        if not args.no_cpu_baseline and world == 1 and args.config == "cfg2":
            # the GPU box shows 256 logical CPUs but a one-GPU job's CPU share is 16: use that many threads
            threads = min(os.cpu_count() or 1, int(os.environ.get("SABC_CPU_THREADS", "16")))
            out["cpu_baseline"], same = cpu_baseline(n, max(args.cpu_updates, W + K), threads, W, K, proposal=args.proposal)
            if same and comparable:
                gm, gv = float(th.mean()), float(th.var())
                out["posterior_vs_cpu"] = {
                    "gpu_mean": gm, "gpu_var": gv, "cpu_mean": same["mean"], "cpu_var": same["var"],
                    "rel_err_mean": abs(gm / same["mean"] - 1.0), "rel_err_var": abs(gv / same["var"] - 1.0),
                    "n_accept_equal": same["n_accept"] == c["n_accept"], "updates": same["updates"],
                    "note": "same seed, same calls on the CPU restatement (oracle); the north star asks for moments within 1 %"}
        elif not args.no_cpu_baseline and world > 1 and comparable:
            # N > 1: the line validates itself.  The oracle runs the SAME initialize + warm-up + K updates (a bounded leg: W + K
            # updates, a few seconds on rank 0's CPU share while the other ranks wait) -- the accept and resample counts have
            # to be equal, epsilon and the moments of shard 0 (global particles 0 .. n_local - 1) those of the oracle's slice
            threads = min(os.cpu_count() or 1, int(os.environ.get("SABC_CPU_THREADS", "16")))
            try:
                cpu, same = cpu_baseline(n, W + K, threads, W, K, proposal=args.proposal, shard0=h.n_local)
            except Exception as e:              # (a secondary leg must not cost the run its line)
                cpu, same = None, None
                out["posterior_vs_cpu"] = {"error": repr(e)}
            gm, gv = float(th.mean()), float(th.var())
            if same:
                out["posterior_vs_cpu"] = {
                    "expected_equal": args.proposal == "randomwalk",      # (DE / Stretch: the shard-local two-colouring, see n1_equivalent)
                    "shard0_gpu_mean": gm, "shard0_gpu_var": gv, "shard0_cpu_mean": same["shard0_mean"], "shard0_cpu_var": same["shard0_var"],
                    "rel_err_mean": abs(gm / same["shard0_mean"] - 1.0), "rel_err_var": abs(gv / same["shard0_var"] - 1.0),
                    "eps_rel_err": float(np.max(np.abs(state_after_timed[1] / np.array(same["eps"]) - 1.0))),
                    "n_accept_equal": same["n_accept"] == c["n_accept"], "n_resampling_equal": same["n_resampling"] == c["n_resampling"],
                    "updates": same["updates"], "cpu_sims_per_s": cpu["value"], "cpu_cores": cpu["cores"],
                    "note": "same seed, same calls on the CPU restatement (oracle), whole population; moments of shard 0 against the "
                            "oracle's same slice; the north star asks for moments within 1 %"}
        print(json.dumps(out), flush=True)
    h.close()                         # (no barrier: sabc_destroy leaves the peer-to-peer group in order)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
