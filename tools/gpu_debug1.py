import faulthandler, sys, time
faulthandler.enable()
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, ".")
import numpy as np
import astar_pairwise_aligner_amd as pa
import oracle
from tests.util_seq import rand_seq
print("devices", pa.capi.load().pa_device_count(), flush=True)
def one(n, w, exact):
    a, b = rand_seq(n, seed=n), rand_seq(64 * w, seed=w)
    oa, ob = oracle.bitprofile_build(a, b)
    h = np.zeros((n, 2), np.uint64); h[:, 0] = 1
    v = np.zeros((w, 2), np.uint64); v[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    ho, vo = h.copy().view(oracle.H_DTYPE).reshape(n), v.copy().view(oracle.V_DTYPE).reshape(w)
    want = oracle.simd_compute(oa, ob, ho, vo, True)
    t = time.time()
    print("call", n, w, exact, flush=True)
    a2 = np.ascontiguousarray(oa).view(np.uint64).reshape(n, 2); b2 = np.ascontiguousarray(ob).view(np.uint64).reshape(w, 2)
    got = pa.compute(a2, b2, h, v, exact)
    ok = got == want and np.array_equal(v, vo.view(np.uint64).reshape(w, 2)) and (not exact or np.array_equal(h, ho.view(np.uint64).reshape(n, 2)))
    print("  ->", got, want, "OK" if ok else "MISMATCH", f"{time.time()-t:.3f}s", flush=True)
for args in [(1, 1, True), (16, 1, True), (100, 3, False), (256, 32, True), (256, 33, True), (256, 100, False), (1000, 70, True)]:
    one(*args)
