import ctypes as C
import os
import tempfile
from typing import List, Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Tests that take a good part of a minute each on the test emulator (every wave collective is 64 fiber switches; the default path
# runs two passes) or on three gloo ranks.  Each has a twin in tests/test_gpu_parity.py (the same check through the real library,
# `-m gpu`) or a smaller sibling that stays in the default run; LQCOV_SLOW_TESTS=1 runs them here too.
import pytest  # noqa: E402
slow_emu = pytest.mark.skipif(os.environ.get("LQCOV_SLOW_TESTS") != "1", reason="slow on the CPU test emulator; its GPU twin runs in -m gpu (LQCOV_SLOW_TESTS=1 runs it here)")

ONT = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-t", "4"]


def run_main(lib, argv: List[str], cwd: Optional[str] = None):
    """lqcov_main through ctypes -> (rc, stdout table, stderr log)."""
    full = [b"minimap2-coverage"] + [str(a).encode() for a in argv]
    arr = (C.c_char_p * len(full))(*full)
    old = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        out, err = os.path.join(d, "o"), os.path.join(d, "e")
        try:
            if cwd:
                os.chdir(cwd)
            rc = lib.lqcov_main(len(full), arr, out.encode(), err.encode(), 0)
        finally:
            os.chdir(old)
        return rc, (open(out).read() if os.path.exists(out) else ""), (open(err).read() if os.path.exists(err) else "")


def parse_sketch_dump(text: str):
    """ref_harness / oracle `sketch` dump -> list of (name, len, [(x,y)...])"""
    reads = []
    for line in text.splitlines():
        f = line.split("\t")
        if f[0] == "R":
            reads.append((f[1], int(f[2]), []))
        elif f[0] == "M":
            reads[-1][2].append((int(f[1], 16), int(f[2], 16)))
    return reads


def parse_chain_dump(text: str):
    """`chains` dump -> (mid_occ, {query index: dict(name, qlen, lambda, lambda2, chains=set(...), ivl=sorted, cnt={idx:n})})"""
    mid = None
    qs = {}
    cur = None
    for line in text.splitlines():
        f = line.split("\t")
        if f[0] == "I":
            mid = int(f[2])
        elif f[0] == "Q":
            cur = dict(name=f[2], qlen=int(f[3]), n_regs=int(f[4]), lambda_=int(f[5]), lambda2=int(f[6]), chains=[], ivl=[], cnt={})
            qs[int(f[1])] = cur
        elif f[0] == "C":
            cur["chains"].append(tuple(int(x) for x in f[1:9]))
        elif f[0] == "V":
            cur["ivl"].append((int(f[1]), int(f[2])))
        elif f[0] == "N":
            cur["cnt"][int(f[1])] = int(f[2])
    return mid, qs


def read_fastx(path):
    """minimal FASTA/Q reader for tests (names, uint8 seq arrays, qual arrays or None)"""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    names, seqs, quals = [], [], []
    with op(path, "rb") as f:
        data = f.read().replace(b"\r\n", b"\n").split(b"\n")
    i = 0
    while i < len(data):
        l = data[i]
        if l.startswith(b"@"):
            names.append(l[1:].split()[0].decode()); seqs.append(np.frombuffer(data[i + 1], dtype=np.uint8)); quals.append(np.frombuffer(data[i + 3], dtype=np.uint8)); i += 4
        elif l.startswith(b">"):
            j = i + 1; parts = []
            while j < len(data) and not data[j].startswith(b">"):
                parts.append(data[j]); j += 1
            names.append(l[1:].split()[0].decode()); seqs.append(np.frombuffer(b"".join(parts), dtype=np.uint8)); quals.append(None); i = j
        else:
            i += 1
    return names, seqs, (quals if all(q is not None for q in quals) else None)
