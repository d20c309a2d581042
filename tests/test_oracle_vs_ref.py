"""Pins the oracle against the reference itself, live: runs oracle/_ref (the reference's own sources
compiled by oracle/Makefile) on seeded synthetic sets and diffs tables and function-level dumps.
Needs oracle/_ref, i.e. the build container (skipped where /root/reference was never available)."""
import os

import pytest

from tests import oracle_bind
from tests.helpers import ONT

pytestmark = pytest.mark.skipif(not oracle_bind.have_ref(), reason="oracle/_ref not built")

VARIANTS = {
    "ont": ONT,
    "pb": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "80", "-t", "4"],
    "fast_k15": ["-Y", "-l", "0", "-q", "160", "-k", "15", "-w", "5", "-I", "4G", "-p", "160", "-t", "4"],
    "hifi_k19w10": ["-Y", "-l", "0", "-q", "160", "-k", "19", "-w", "10", "-I", "4G", "-p", "160", "-t", "4"],
    "parts": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "500K", "-p", "160", "-t", "4"],
    "spike_hpc": ["-Y", "-Hk15", "-w", "10", "-c", "1", "-l", "0", "--filter", "-t", "4"],
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("name", ["tiny", "small"])
def test_table_identical_to_reference(datasets, name, variant):
    tf, qf = datasets(name)
    argv = VARIANTS[variant] + [tf, qf]
    assert oracle_bind.table(argv) == oracle_bind.ref_table(argv)


@pytest.mark.parametrize("kwh", [("12", "5", "0"), ("15", "5", "0"), ("19", "10", "0"), ("15", "10", "1"), ("7", "40", "0")])
def test_sketch_identical_to_mm_sketch(datasets, kwh):
    tf, _ = datasets("tiny")
    k, w, h = kwh
    want = oracle_bind.ref_dump(["sketch", k, w, h, tf])
    got = oracle_bind.dump("sketch", ["-k", k, "-w", w] + (["-H"] if h == "1" else []), [tf])
    assert got == want


def test_index_parts_and_mid_occ(datasets):
    tf, _ = datasets("small")
    assert oracle_bind.dump("index", ["-k", "12", "-w", "5", "-I", "500K"], [tf]) == oracle_bind.ref_dump(["index", "12", "5", "0", "500K", tf])


def test_chains_intervals_counters(datasets):
    tf, qf = datasets("small")
    want = oracle_bind.ref_dump(["chains", "12", "5", "0", "4G", "40", "160", "160", tf, qf])
    got = oracle_bind.dump("chains", ["-k", "12", "-w", "5", "-m", "40", "-p", "160", "-q", "160", "-l", "0"], [tf, qf])
    assert got == want


def test_klib_order_matters_stable_sort_is_not_exact(tmp_path):
    """Documents why the GPU sort reproduces klib's unstable order (kernels_sort.hpp): on reads full of repeats (queries
    that carry the same minimizer several times: anchors with equal x) a stable (x, emission) anchor order changes rows; the
    grouped DP does not.  (The same holds for cfg1 split into 1-Mbase parts: round 1's form of this test.)"""
    from tests.test_emu_pipeline import _repeat_rich_dataset
    from tests.helpers import ONT
    tf, qf = _repeat_rich_dataset(tmp_path, 0)
    argv = ONT + [tf, qf]
    ref = oracle_bind.ref_table(argv)
    assert oracle_bind.table(argv) == ref
    assert oracle_bind.table(argv, ["--grouped"]) == ref
    assert oracle_bind.table(argv, ["--stable-sort"]) != ref


def test_q2p_table_regenerated_equals_reference_literals():
    """meanQ's lookup table is rebuilt numerically by the oracle and the product (10^(-q/10) to 15 decimals, eight
    entries one unit higher); compare that description with the literals in the reference source (read as data)."""
    import re
    from decimal import Decimal
    src = "/root/reference/minimap2-coverage/lqutils.c"
    if not os.path.exists(src):
        pytest.skip("reference sources not present")
    txt = open(src).read()
    block = txt[txt.rindex("double q2p[] = {"):]
    block = block[:block.index("};")]
    lits = re.findall(r"\d\.\d{15}", block)
    assert len(lits) == 127
    up = {34, 39, 58, 62, 67, 71, 72, 82}
    for q, v in enumerate(lits):
        mine = Decimal("%.15f" % (10.0 ** (-q / 10.0))) + (Decimal("0.000000000000001") if q in up else 0)
        assert mine == Decimal(v), q


def test_meanq_high_qualities_match_reference(tmp_path):
    """qualities up to Q93 ('~'), including the eight adjusted table entries"""
    import numpy as np
    from longqc_amd import synth
    rng = np.random.default_rng(3)
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [A[rng.integers(0, 4, 400)] for _ in range(30)]
    quals = [(33 + rng.integers(0, 94, 400)).astype(np.uint8) for _ in range(30)]
    for i, q0 in enumerate([34, 39, 58, 62, 67, 71, 72, 82]):
        quals[i][:] = 33 + q0
    rs = synth.ReadSet(["q%d" % i for i in range(30)], seqs, quals)
    fn = str(tmp_path / "q.fq")
    synth.write_fastq(fn, rs)
    argv = ONT + [fn, fn]
    assert oracle_bind.table(argv) == oracle_bind.ref_table(argv)
