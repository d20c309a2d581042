"""kernel times of the engine's scan and sort primitives at several sizes (run under rocprofv3 --kernel-trace --stats)"""
import sys
import numpy as np
from longqc_amd import api
lib = api.load_library()
p = api.Params(); lib.lqcov_params_default(p)
eng = api.Engine(p, 0, lib=lib)
rng = np.random.default_rng(3)
for n in [int(a) for a in sys.argv[1:]] or [1 << 20, 1 << 24, 1 << 26, 1 << 28]:
    c = rng.integers(0, 100, size=n, dtype=np.uint64).astype(np.uint32)
    out = eng.debug_scan(c)
    assert out[-1] == int(c[:-1].astype(np.uint64).sum())
    print("scan", n, "ok", flush=True)
