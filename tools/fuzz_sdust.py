#!/usr/bin/env python3
"""The second binary, `sdust` (sdust.c:136-222, run per chunk by lq_mask.py:17-23): random reads -- random bases, homopolymers,
short tandem repeats, N runs, lower case, very short reads, FASTA and FASTQ -- and random `-w` / `-t` through the emulator build
of k_sdust (tests/emu) and through the reference binary (oracle/_ref/sdust); the two tables must be the same bytes (-w up to 66:
the engine refuses wider windows).
    python tools/fuzz_sdust.py [--n 300] [--seed 1]"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lib", default=os.environ.get("LQCOV_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "liblqcov_emu.so"))
    args = ap.parse_args()
    from longqc_amd import api
    from tests import oracle_bind
    from tests.test_sdust import run_sdust_main
    lib = api.load_library(args.lib)
    ref = os.path.join(os.path.dirname(oracle_bind.REF_BIN), "sdust")
    rng = np.random.default_rng(args.seed)
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for it in range(args.n):
            reads = []
            for r in range(int(rng.integers(1, 12))):
                parts = []
                for _ in range(int(rng.integers(1, 8))):
                    kind = int(rng.integers(0, 6))
                    L = int(rng.integers(1, 400))
                    if kind == 0:
                        parts.append(A[rng.integers(0, 4, L)])
                    elif kind == 1:
                        parts.append(np.full(L, A[rng.integers(0, 4)], dtype=np.uint8))
                    elif kind == 2:
                        u = A[rng.integers(0, 4, int(rng.integers(2, 7)))]
                        parts.append(np.tile(u, L // len(u) + 1)[:L])
                    elif kind == 3:
                        parts.append(np.full(int(rng.integers(1, 30)), ord("N"), dtype=np.uint8))
                    elif kind == 4:
                        parts.append(np.frombuffer(bytes(A[rng.integers(0, 4, L)]).lower(), dtype=np.uint8))
                    else:
                        u = A[rng.integers(0, 4, int(rng.integers(8, 40)))]
                        parts.append(np.tile(u, int(rng.integers(2, 6))))
                s = np.concatenate(parts)
                if rng.random() < 0.1:
                    s = s[:int(rng.integers(1, 5))]
                reads.append(s)
            fasta = rng.random() < 0.3
            fn = os.path.join(d, "r.fa" if fasta else "r.fq")
            with open(fn, "wb") as f:
                for i, s in enumerate(reads):
                    if fasta:
                        f.write(b">s%d\n" % i + s.tobytes() + b"\n")
                    else:
                        f.write(b"@s%d\n" % i + s.tobytes() + b"\n+\n" + (33 + rng.integers(0, 60, len(s))).astype(np.uint8).tobytes() + b"\n")
            extra = []
            if rng.random() < 0.5:
                extra += ["-w", str(int(rng.choice([3, 8, 16, 20, 32, 50, 64, 66])))]
            if rng.random() < 0.5:
                extra += ["-t", str(int(rng.choice([5, 12, 20, 40])))]
            a = subprocess.run([ref] + extra + [fn], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            rc, out, err = run_sdust_main(lib, extra + [fn], tmp=d)
            if a.returncode != 0 and rc != 0:
                continue                            # (both refuse)
            if a.returncode != 0 or rc != 0 or a.stdout.decode() != out:
                bad += 1
                print("case %d differs (%s, ref rc %d, engine rc %d): %s" % (it, " ".join(extra), a.returncode, rc, err[-200:]))
                al, bl = a.stdout.decode().splitlines(), out.splitlines()
                for i in range(max(len(al), len(bl))):
                    if i >= len(al) or i >= len(bl) or al[i] != bl[i]:
                        print("  row %d: reference %r engine %r (read of %d bases)" % (i, al[i] if i < len(al) else None, bl[i] if i < len(bl) else None, len(reads[i]) if i < len(reads) else -1))
                        break
                if bad >= 5:
                    break
    print("%d inputs, %d differ" % (it + 1, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
