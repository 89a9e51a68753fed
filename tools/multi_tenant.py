"""Several independent small-population chains on ONE device at once (one process each): their one-launch updates compete for
the compute units.  Every workgroup of such a launch has to be resident at the same time; when the device is too full for that
the launch leaves at its rendezvous without touching anything and the call goes on as the launch chain
(sabc_persistent_fallbacks) -- every chain must finish, with the counts of a chain that had the device to itself.
usage (GPU box): python tools/multi_tenant.py [processes=6] [n=4096] [updates=3000] [proposal=rw]"""
import json, os, subprocess, sys, time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)


def worker(n, updates, prop, tag, meet=None, procs=1):
    import sabc_amd as S
    from tests.cases import hip_model_prior, hip_proposal
    model, prior = hip_model_prior(S, "gauss1_cfg2")
    h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=7)
    h.initialize(n)
    if meet:                                            # start together: a file per tenant that is ready
        open(os.path.join(meet, tag), "w").close()
        while len(os.listdir(meet)) < procs:
            time.sleep(0.001)
    t0 = time.perf_counter()
    calls = 6
    for _ in range(calls):
        h.update(n_simulation=(updates // calls) * n, proposal=hip_proposal(S, prop, 1), resample=n)
    dt = time.perf_counter() - t0
    c = dict(h.counters)
    print(json.dumps(dict(tag=tag, seconds=round(dt, 3), us_per_update=round(dt / (updates // calls * calls) * 1e6, 2), n_accept=c["n_accept"],
                          n_resampling=c["n_resampling"], launches=h.persistent_launches, fallbacks=h.persistent_fallbacks,
                          lanes=h.persistent_lanes, eps=float(h.eps[0]))), flush=True)
    h.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6] if len(sys.argv) > 6 else None,
               int(sys.argv[7]) if len(sys.argv) > 7 else 1)
        sys.exit(0)
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    updates = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    prop = sys.argv[4] if len(sys.argv) > 4 else "rw"
    env = dict(os.environ, PYTHONPATH=root)
    alone = subprocess.run([sys.executable, __file__, "--worker", str(n), str(updates), prop, "alone"], env=env, capture_output=True, text=True, timeout=300)
    print(alone.stdout.strip() or alone.stderr[-400:])
    import tempfile
    meet = tempfile.mkdtemp(prefix="sabc_tenants_")
    ps = [subprocess.Popen([sys.executable, __file__, "--worker", str(n), str(updates), prop, f"tenant{i}", meet, str(procs)], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True) for i in range(procs)]
    ok = True
    for p in ps:
        try:
            out, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            ok = False
        print(out.strip() or ("FAILED: " + err[-400:]))
        ok = ok and p.returncode == 0
    print("all tenants finished" if ok else "A TENANT FAILED")
    sys.exit(0 if ok else 1)
