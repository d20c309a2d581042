#!/usr/bin/env python3
"""Differential fuzz of the engine's device-wide primitives (longqc_amd/csrc/kernels_isort.hpp) on the GPU: random sizes (1 .. 6 M
pairs, biased to tile and range boundaries), key widths (1..32 bits in 4-byte keys, 33..64 in 8-byte keys), key distributions
(uniform, minimizer-like skew, few values, sorted, reversed) against numpy's stable sort / cumsum.
    python tools/fuzz_primitives.py [--cases 300] [--seed 1] [--emu]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longqc_amd import api


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--emu", action="store_true", help="the test emulator build instead of the device (small sizes)")
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = api.load_library(os.path.join(root, "tests", "emu", "liblqcov_emu.so")) if a.emu else api.load_library()
    p = api.Params(); lib.lqcov_params_default(p)
    eng = api.Engine(p, 0, lib=lib)
    rng = np.random.default_rng(a.seed)
    t0 = time.time(); pairs = 0
    for c in range(a.cases):
        kind = rng.integers(0, 5)
        top = 30000 if a.emu else 6_000_000
        n = int(rng.choice([rng.integers(1, 70), rng.integers(5000, 5300), rng.integers(1, top), 5120 * int(rng.integers(1, 70)) + int(rng.integers(-2, 3))]))
        n = max(1, min(n, top))
        wide = rng.random() < 0.25
        bits = int(rng.integers(33, 65)) if wide else int(rng.integers(1, 33))
        if kind == 0: keys = rng.integers(0, 1 << min(bits, 63), size=n, dtype=np.uint64)
        elif kind == 1: keys = rng.integers(0, 1 << min(bits, 63), size=(n, 5), dtype=np.uint64).min(axis=1)
        elif kind == 2: keys = rng.integers(0, 3, size=n, dtype=np.uint64) << np.uint64(max(bits - 2, 0))
        elif kind == 3: keys = np.sort(rng.integers(0, 1 << min(bits, 63), size=n, dtype=np.uint64))
        else: keys = np.sort(rng.integers(0, 1 << min(bits, 63), size=n, dtype=np.uint64))[::-1].copy()
        if bits < 64: keys &= np.uint64((1 << bits) - 1)        # (the sort looks at the low `bits` bits only)
        vals = np.arange(n, dtype=np.uint64)
        k, v = eng.debug_sort_pairs(keys, vals, bits, 8 if wide else 4)
        o = np.argsort(keys, kind="stable")
        if not (np.array_equal(k, keys[o]) and np.array_equal(v, vals[o])):
            print("MISMATCH: sort case %d n %d bits %d kind %d" % (c, n, bits, kind)); sys.exit(1)
        cnt = rng.integers(0, 1 << int(rng.integers(1, 33)), size=n, dtype=np.uint64).astype(np.uint32)
        want = np.zeros(n, dtype=np.uint64); want[1:] = np.cumsum(cnt.astype(np.uint64))[:-1]   # (all in uint64: a list in np.concatenate would make it float64)
        got = eng.debug_scan(cnt)
        if not np.array_equal(got, want):
            bad = np.nonzero(got != want)[0]
            print("MISMATCH: scan case %d n %d: %d wrong, first at %d (tile %d, slot %d): got %d want %d (diff %d), max count %d; last wrong %d" % (
                c, n, bad.size, bad[0], bad[0] // 4096, bad[0] % 4096, got[bad[0]], want[bad[0]], int(got[bad[0]]) - int(want[bad[0]]), int(cnt.max()), bad[-1]))
            again = eng.debug_scan(cnt)
            print("  the same call again: %s" % ("identical to numpy" if np.array_equal(again, want) else "wrong again, %d places, same as before: %s" % (int((again != want).sum()), np.array_equal(again, got))))
            sys.exit(1)
        pairs += n
    print("fuzz_primitives: %d cases (%d pairs sorted and scanned), all identical to numpy, %.1f s" % (a.cases, pairs, time.time() - t0))


if __name__ == "__main__":
    main()
