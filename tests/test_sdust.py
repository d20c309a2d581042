"""the low-complexity table of the reference's second binary `sdust` (sdust.c).  Oracle
restatement vs fixtures written by the reference binary (tests/golden/make_sdust_golden.py), the product's kernel under
the serial HIP stand-in vs the same fixtures.  GPU versions: test_gpu_parity.py."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_bind
from tests.conftest import GOLDEN, read_gz
from tests.helpers import read_fastx

CASES = json.load(open(os.path.join(GOLDEN, "sdust_cases.json")))


def run_sdust_main(lib, argv, cwd=None, tmp=None):
    full = [b"sdust"] + [str(a).encode() for a in argv]
    arr = (C.c_char_p * len(full))(*full)
    lib.lqsdust_main.restype = C.c_int
    lib.lqsdust_main.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.c_int]
    out, err = os.path.join(str(tmp), "o"), os.path.join(str(tmp), "e")
    old = os.getcwd()
    try:
        if cwd:
            os.chdir(cwd)
        rc = lib.lqsdust_main(len(full), arr, out.encode(), err.encode(), 0)
    finally:
        os.chdir(old)
    return rc, (open(out).read() if os.path.exists(out) else ""), (open(err).read() if os.path.exists(err) else "")


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_oracle_sdust_equals_the_reference_table(case, tmp_path):
    exe = oracle_bind.ensure_oracle()
    r = subprocess.run([exe, "sdust"] + case["argv"], cwd=GOLDEN, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode() == read_gz(case["expect"])
    assert case["masked_total"] > 0                               # the fixture does exercise the masking


def test_oracle_sdust_vs_reference_binary_on_synthetic_reads(datasets):
    ref = os.path.join(os.path.dirname(oracle_bind.REF_BIN), "sdust")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/sdust not built (needs /root/reference)")
    tf, _ = datasets("small")
    for extra in ([], ["-w", "20"], ["-t", "12"]):
        a = subprocess.run([ref] + extra + [tf], stdout=subprocess.PIPE, check=True).stdout
        b = subprocess.run([oracle_bind.ensure_oracle(), "sdust"] + extra + [tf], stdout=subprocess.PIPE, check=True).stdout
        assert a == b


def check_main_equals_fixture(lib, case, tmp_path):
    rc, out, err = run_sdust_main(lib, case["argv"], cwd=GOLDEN, tmp=tmp_path)
    assert rc == 0, err
    assert out == read_gz(case["expect"])


def check_in_memory_rows(lib, tmp_path):
    from longqc_amd import sdust
    for fn, exp in (("adv_sub.fq.gz", "adv_sub.sdust.gz"), ("adv_all.fa.gz", "adv_all.sdust.gz")):
        names, seqs, quals = read_fastx(os.path.join(GOLDEN, fn))
        rows = sdust.sdust_rows(names, seqs, quals, lib=lib)
        assert "\n".join(rows) + "\n" == read_gz(exp)
    out = str(tmp_path / "t.txt")
    sdust.run_sdust(os.path.join(GOLDEN, "tiny_all.fq.gz"), out, lib=lib)
    assert open(out).read() == read_gz("tiny_all.sdust.gz")
    with pytest.raises(Exception):
        sdust.run_sdust(os.path.join(GOLDEN, "tiny_all.fq.gz"), out, w=200, lib=lib)      # window beyond the supported range
    with pytest.raises(Exception):
        sdust.run_sdust(os.path.join(GOLDEN, "no_such_file.fq"), out, lib=lib)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_emulated_sdust_main_equals_the_reference_table(emu_lib, case, tmp_path):
    check_main_equals_fixture(emu_lib, case, tmp_path)


def test_emulated_sdust_in_memory_rows_and_errors(emu_lib, tmp_path):
    check_in_memory_rows(emu_lib, tmp_path)


def test_emulated_sdust_edge_reads(emu_lib, tmp_path):
    """empty read, read shorter than a word, all-N, a homopolymer, N in the middle of a low-complexity run"""
    p = str(tmp_path / "e.fq")
    recs = [("empty", ""), ("two", "AC"), ("alln", "N" * 70), ("polya", "A" * 200), ("split", "AT" * 40 + "N" + "AT" * 40),
            ("mixed", "ACGTTGCA" * 5 + "a" * 30 + "ACGGTCAGTC" * 4)]
    with open(p, "w") as f:
        for n, s in recs:
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "5" * len(s)))
    exe = oracle_bind.ensure_oracle()
    want = subprocess.run([exe, "sdust", p], stdout=subprocess.PIPE, check=True).stdout.decode()
    rc, out, err = run_sdust_main(emu_lib, [p], tmp=tmp_path)
    assert rc == 0, err
    assert out == want
    ref = os.path.join(os.path.dirname(oracle_bind.REF_BIN), "sdust")
    if os.path.exists(ref):
        assert subprocess.run([ref, p], stdout=subprocess.PIPE, check=True).stdout.decode() == want


def write_repeat_rich_reads(path, n=150, seed=11):
    """reads made mostly of short tandem repeats (unit 1..6, 3 % substitutions), lower-case two-letter stretches and N
    runs between random sequence: most bases end up masked, the perfect-interval list gets long"""
    rng = np.random.default_rng(seed)
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "w") as f:
        for i in range(n):
            parts = []
            for _ in range(int(rng.integers(1, 12))):
                k = rng.random()
                if k < 0.35:
                    parts.append(A[rng.integers(0, 4, size=int(rng.integers(20, 800)))].tobytes().decode())
                elif k < 0.85:
                    u = A[rng.integers(0, 4, size=int(rng.integers(1, 7)))].tobytes().decode()
                    rep = list(u * int(rng.integers(5, 400)))
                    for j in range(len(rep)):
                        if rng.random() < 0.03:
                            rep[j] = "ACGT"[int(rng.integers(0, 4))]
                    parts.append("".join(rep))
                elif k < 0.93:
                    parts.append("N" * int(rng.integers(1, 5)))
                else:
                    parts.append(A[rng.integers(0, 2, size=int(rng.integers(30, 300)))].tobytes().decode().lower())
            s = "".join(parts)
            f.write("@s%d\n%s\n+\n%s\n" % (i, s, "".join(chr(33 + int(x)) for x in rng.integers(2, 45, size=len(s)))))


def check_repeat_rich(lib, tmp_path, variants=(("64", "20"), ("64", "8"), ("20", "15"), ("66", "30"))):
    p = str(tmp_path / "stress.fq")
    write_repeat_rich_reads(p)
    exe = oracle_bind.ensure_oracle()
    ref = os.path.join(os.path.dirname(oracle_bind.REF_BIN), "sdust")
    for w, t in variants:
        want = subprocess.run([exe, "sdust", "-w", w, "-t", t, p], stdout=subprocess.PIPE, check=True).stdout.decode()
        if os.path.exists(ref):
            assert subprocess.run([ref, "-w", w, "-t", t, p], stdout=subprocess.PIPE, check=True).stdout.decode() == want
        rc, out, err = run_sdust_main(lib, ["-w", w, "-t", t, p], tmp=tmp_path)
        assert rc == 0, err
        assert out == want
        tot = sum(int(l.split("\t")[1]) for l in out.splitlines()); ln = sum(int(l.split("\t")[2]) for l in out.splitlines())
        assert tot > 0.3 * ln


def test_emulated_sdust_repeat_rich_reads(emu_lib, tmp_path):
    check_repeat_rich(emu_lib, tmp_path, variants=(("64", "20"), ("20", "15")))      # (the GPU test runs all four)


def test_emulated_lqmask_counterpart(emu_lib, tmp_path):
    """LqMaskMI355X: the chunk loop of lq_mask.LqMask (submit_sdust / close_pool / get_outfile_path) without temporary files"""
    from longqc_amd import sdust
    names, seqs, quals = read_fastx(os.path.join(GOLDEN, "adv_sub.fq.gz"))
    reads = [[n, bytes(s).decode(), bytes(q).decode()] for n, s, q in zip(names, seqs, quals)]
    lm = sdust.LqMaskMI355X(str(tmp_path / "out"), suffix="x", lib=emu_lib)
    lm.submit_sdust(reads[:7], 0)
    lm.submit_sdust(reads[7:], 1)
    lm.close_pool()
    assert lm.get_outfile_path().endswith("longqc_sdust_x.txt")
    assert open(lm.get_outfile_path()).read() == read_gz("adv_sub.sdust.gz")
