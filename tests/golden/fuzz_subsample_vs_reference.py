#!/usr/bin/env python3
"""Build container only (imports the REFERENCE's lq_utils from /root/reference, like make_subsample_golden.py): random chunk
sizes, sample sizes, fractions and exclusion lists through longqc_amd.sampleqc.subsample_from_chunk and through the reference's
lq_utils.subsample_from_chunk (lq_utils.py:371-411), slot by slot.  Nothing of this runs on the GPU box or in the test suite --
the committed vectors (subsample.json) do; this is how they were widened at the end of round 5.
    python tests/golden/fuzz_subsample_vs_reference.py [n_cases] [seed]"""
import copy, os, sys, types
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for m in ("pysam", "h5py", "edlib"):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import lq_utils  # noqa: E402
sys.path.insert(0, ROOT)
from longqc_amd import sampleqc  # noqa: E402


def chunk(lo, hi):
    return [["r%06d" % i, "ACGT" * (1 + i % 7), "IIII" * (1 + i % 7)] for i in range(lo, hi)]


def names(s):
    return [r[0] if r else 0 for r in s]


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for it in range(n_cases):
        kind = int(rng.integers(0, 4))
        sizes = [int(rng.integers(1, 800)) for _ in range(int(rng.integers(1, 5)))]
        num = int(rng.integers(1, 120)) if kind != 3 else float(rng.choice([0.01, 0.1, 0.5, 0.9]))
        el = None
        if kind == 2:
            el = set("r%06d" % i for i in rng.integers(0, sum(sizes), size=int(rng.integers(1, sum(sizes)))))
        got, want, cum, lo = ([0] * num if el is not None else []), ([0] * num if el is not None else []), 0, 0
        for sz in sizes:
            a = sampleqc.subsample_from_chunk(chunk(lo, lo + sz), cum, copy.deepcopy(got), num, elist=el) if el is not None else sampleqc.subsample_from_chunk(chunk(lo, lo + sz), cum, copy.deepcopy(got), num)
            b = lq_utils.subsample_from_chunk(chunk(lo, lo + sz), cum, copy.deepcopy(want), num, elist=el) if el is not None else lq_utils.subsample_from_chunk(chunk(lo, lo + sz), cum, copy.deepcopy(want), num)
            got, want = a, b
            cum += sz; lo += sz
            if kind == 3:
                break                               # (fraction mode: one chunk, as LongQC uses it)
        if names(got) != names(want):
            bad += 1
            print("case %d differs: kind %d sizes %s num %s elist %s\n  ours %s\n  ref  %s" % (it, kind, sizes, num, None if el is None else len(el), names(got)[:12], names(want)[:12]))
    print("%d cases, %d differ" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
