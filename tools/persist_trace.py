"""Where the time of an update goes inside the one-launch form (k_update_persistent), from a variant library built with
-DSABC_PERSIST_TRACE (tools/build_variants.sh trace="-DSABC_PERSIST_TRACE"): workgroup 0 stamps the phases of its first 64
updates with the 100 MHz wall clock.  usage (GPU box): python tools/persist_trace.py [n] [proposal] [config]"""
import ctypes, os, shutil, sys
import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
variant = os.path.join(root, "tools", "exp_libs", "lib_trace.so")
product = os.path.join(root, "simulatedannealingabc.jl_amd", "libsabc_hip.so")
keep = product + ".keep"
os.replace(product, keep)                      # (renames, not overwrites: a mapped library must not change under its process)
shutil.copy(variant, product)
try:
    import sabc_amd as S
    from tests.cases import hip_model_prior, hip_proposal, MODELS
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    prop = sys.argv[2] if len(sys.argv) > 2 else "rw"
    name = sys.argv[3] if len(sys.argv) > 3 else "gauss1_cfg2"
    lib = S._lib.lib()
    for lanes in ("1", "4", "16"):
        os.environ["SABC_PERSISTENT"] = "1"; os.environ["SABC_PERSISTENT_LANES"] = lanes
        model, prior = hip_model_prior(S, name)
        h = S.SabcHandle(n_particles=n, model=model, prior=prior, seed=7)
        h.initialize(n)
        h.update(n_simulation=40 * n, proposal=hip_proposal(S, prop, len(MODELS[name]["prior"])), resample=10 ** 9)
        buf = (ctypes.c_ulonglong * 1024)()
        rc = lib.sabc_debug_persist_trace(buf)
        t = np.array(buf, dtype=np.float64).reshape(64, 16) * 0.01      # us, row = iter % 64
        t = t[np.argsort(t[:, 0])][-30:]                                # the last 30 updates, in time order
        # (with a control wave the next update's proposal and simulation are drafted beside the step's second part: stamps 6-8 of
        # an update fall into the previous one's tail, 12-13 are the control wave's -- not thread 0's)
        seq = [0, 9, 10, 1, 2, 4, 11, 5]
        names = ["ecdf+log alpha", "accept+store", "moments", "reduce", "row exchange", "control, first part", "draft of the next || second part"]
        d = np.diff(t[:, seq], axis=1)
        med = np.median(d, axis=0)
        print(f"n {n} {prop} {name} lanes {lanes}: " + "  ".join(f"{nm} {v:.2f}" for nm, v in zip(names, med)) +
              f"  | update {np.median(np.diff(t[:, 0])):.2f} us (rc {rc})")
        h.close()
finally:
    os.replace(keep, product)
