import os, time, sys
sys.path.insert(0, os.getcwd())
os.environ["SABC_RTC_CACHE"] = "0"
import sabc_amd as S
from tests.test_user_simulator import GAUSS_IID_FOR_PAIRS_SRC, DECAY_SRC, DECAY_OBS
for name, src, d, s, p, prior in (("gauss (1,1)", GAUSS_IID_FOR_PAIRS_SRC, 1, 1, [100, 1.0, 1.4, 0.0], S.Normal(0.0, 2.0)),):
    for n in (1000, 200000):
        t = time.time()
        r = S.sabc(S.DeviceSource(src, d, s, p), prior, n_particles=n, n_simulation=3 * n, seed=3)
        print(name, "n", n, "first sabc() incl. compilation %.1f s" % (time.time() - t))
