#!/usr/bin/env python3
"""The two FASTA/FASTQ readers on random text: the streaming reader (csrc/fastx.hpp, kseq's grammar, kseq.h:179-224) and the
mapped-file reader (csrc/fastx_mem.hpp: pieces parsed by several threads from guessed record starts) must report the same
records -- count, bases, and the two digests of names+sequences and of qualities -- on every input, however malformed.
    python tools/fuzz_reader.py [--n 20000] [--seed 1]        (host code only: any build of the library will do)"""
import argparse
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lib", default=os.environ.get("LQCOV_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "liblqcov_emu.so"))
    args = ap.parse_args()
    from longqc_amd import api
    lib = api.load_library(args.lib)
    rng = np.random.default_rng(args.seed)
    toks = [b">", b"@", b"+", b"\n", b"\n", b"\n", b"\r\n", b"\r", b" ", b"\t", b"ACGT", b"acgtnN", b"U", b"!!!!", b"IIII", b"@@", b">>", b"+\n", b"name", b"x y", b""]
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "f.txt")
        for it in range(args.n):
            kind = rng.integers(0, 3)
            if kind == 0:                           # token soup
                data = b"".join(toks[i] for i in rng.integers(0, len(toks), size=int(rng.integers(0, 60))))
            else:                                   # mostly well-formed records with a few damaged bytes
                recs = []
                for r in range(int(rng.integers(1, 8))):
                    L = int(rng.integers(0, 40))
                    s = bytes(rng.choice(list(b"ACGTN"), size=L).astype(np.uint8)) if L else b""
                    eol = b"\r\n" if rng.random() < 0.2 else b"\n"
                    if rng.random() < 0.5:
                        w = int(rng.integers(1, 30))
                        body = eol.join(s[i:i + w] for i in range(0, len(s), w)) if rng.random() < 0.5 and L else s
                        recs.append(b">r%d c" % r + eol + body + eol)
                    else:
                        q = bytes(rng.integers(33, 74, size=L).astype(np.uint8)) if L else b""   # (qualities may hold '@', '>' and '+')
                        recs.append(b"@r%d" % r + eol + s + eol + b"+" + eol + q + eol)
                data = bytearray(b"".join(recs))
                for _ in range(int(rng.integers(0, 3)) if kind == 2 else 0):
                    if data:
                        data[int(rng.integers(0, len(data)))] = int(rng.choice(list(b">@+\n\r A!")))
                if rng.random() < 0.2 and data:
                    data = data[:int(rng.integers(0, len(data)))]      # truncated
                data = bytes(data)
            open(fn, "wb").write(data)
            got = []
            for mode, th, piece in ((0, 1, 0), (1, 4, 16), (1, 3, 7)):
                out = (C.c_uint64 * 5)()
                rc = lib.lqcov_fastx_digest(fn.encode(), mode, th, piece, out)
                got.append((rc,) + tuple(int(x) for x in out)[:4])        # (out[4]: pieces parsed again, the parallel reader's own business)
            if len(set(got)) != 1:
                bad += 1
                print("case %d differs: %r\n  streaming %s\n  mapped/4/16 %s\n  mapped/3/7 %s" % (it, data, got[0], got[1], got[2]))
                if bad >= 5:
                    break
    print("%d inputs, %d on which the readers disagree" % (it + 1, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
