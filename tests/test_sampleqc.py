"""Query-set construction and argv building against vectors captured from the
reference's own lq_utils.subsample_from_chunk (tests/golden/make_subsample_golden.py), and the in-memory
coverage path against the file-based one (emulator build; the GPU variant lives in test_gpu_parity.py)."""
import json
import os

import numpy as np
import pytest

from longqc_amd import sampleqc, synth
from tests.conftest import GOLDEN, read_gz
from tests.helpers import read_fastx


def _chunk(lo, hi):
    return [["r%06d" % i, "ACGT" * (1 + i % 7), "IIII" * (1 + i % 7)] for i in range(lo, hi)]


def _names(s):
    return [r[0] if r else 0 for r in s]


CASES = json.load(open(os.path.join(GOLDEN, "subsample.json")))


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["kind"] + str(c.get("n", c.get("sizes"))))
def test_subsample_matches_reference_lq_utils(case):
    if case["kind"] == "single":
        s = sampleqc.subsample_from_chunk(_chunk(0, case["n"]), 0, [], case["num"])
    elif case["kind"] == "multi":
        s, cum, lo = [], 0, 0
        for sz in case["sizes"]:
            s = sampleqc.subsample_from_chunk(_chunk(lo, lo + sz), cum, s, case["num"])
            cum += sz; lo += sz
    elif case["kind"] == "elist":
        s = sampleqc.subsample_from_chunk(_chunk(0, case["n"]), 0, [0] * case["num"], case["num"], elist=set(case["elist"]))
    else:
        s = sampleqc.subsample_from_chunk(_chunk(0, case["n"]), 0, [], case["frac"])
    assert _names(s) == case["expect"]


def test_synth_reservoir_is_the_same_draw():
    n, num = 1000, 37
    want = [c for c in CASES if c["kind"] == "single" and c["n"] == n][0]["expect"]
    assert ["r%06d" % i for i in synth.reservoir_subsample(n, num)] == want


def test_argv_table_matches_longqc():
    a = sampleqc.coverage_argv("ont-ligation", "in.fq", "sub.fq", ncpu=8)
    assert a == "-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 160 -t 8 in.fq sub.fq".split()
    assert sampleqc.coverage_argv("pb-sequel", "a", "b", ncpu=4, fast=True) == "-Y -l 0 -q 160 -k 15 -w 5 -I 4G -p 80 -t 4 a b".split()
    assert sampleqc.coverage_argv("pb-hifi", "a", "b", fast=True)[5:9] == ["-k", "19", "-w", "10"]
    assert sampleqc.coverage_argv("ont-rapid", "a", "b", short=True) == "-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 140 -t 4 a b".split()
    assert sampleqc.spikein_argv("refs/Sequel_control_reference.fasta", "sub.fq", 4) == \
        "-Y -Hk15 -w 10 -c 1 -l 0 --filter -t 4 refs/Sequel_control_reference.fasta sub.fq".split()
    # sampleqc --db (longQC.py:266-277, 440-442): index first, then map against the prebuilt index without -k/-w
    assert sampleqc.db_build_argv("ont-ligation", "t_db", "in.fq") == "-k 12 -w 5 -I 4G -d t_db in.fq".split()
    assert sampleqc.db_build_argv("pb-hifi", "t_db", "in.fq", fast=True) == "-k 19 -w 10 -I 4G -d t_db in.fq".split()
    assert sampleqc.db_build_argv("pb-sequel", "t_db_s", "in.fq", short=True) == "-k 12 -w 5 -I 4G -d t_db_s in.fq".split()
    assert sampleqc.db_coverage_argv("ont-ligation", "t_db", "sub.fq", ncpu=8) == "-Y -l 0 -q 160 -p 160 -t 8 t_db sub.fq".split()


def test_replace_masked_and_write_fastq(tmp_path):
    reads = _chunk(0, 200)
    s = sampleqc.subsample_from_chunk(reads, 0, [], 20)
    bad = [s[3][0], s[11][0]]
    s2 = sampleqc.replace_masked(list(s), bad, iter([(reads, len(reads), 0)]))
    assert len(s2) == 20 and not (set(bad) & set(_names(s2))) and len(set(_names(s2))) == 20
    assert [a for a, b in zip(_names(s), _names(s2)) if a != b] == bad      # only the masked slots changed
    s3 = sampleqc.replace_masked(list(s), _names(reads), iter([(reads, len(reads), 0)]))   # nothing left to draw
    assert s3 == []
    fn = str(tmp_path / "subsample.fastq")
    assert sampleqc.write_fastq(fn, s2) is True
    assert sampleqc.write_fastq(fn, s2) is None                              # exists (lq_utils.py:353-355)
    assert open(fn).read().startswith("@%s\n%s\n+\n%s\n" % tuple(s2[0]))
    assert sampleqc.write_fastq(str(tmp_path / "e.fq"), []) is None


def _as_chunks(path, chunk_reads):
    n, s, q = read_fastx(path)
    reads = [[a, b.tobytes().decode(), (c.tobytes().decode() if c is not None else "")] for a, b, c in zip(n, s, q or [None] * len(n))]
    for i in range(0, len(reads), chunk_reads):
        c = reads[i:i + chunk_reads]
        yield c, len(c), sum(len(r[1]) for r in c)


def check_in_memory_equals_file_path(lib):
    from longqc_amd import api
    argv = sampleqc.coverage_argv("ont-ligation", "x", "y", inds="100K")
    p, _, _ = api.parse_args(argv)
    eng = api.Engine(p, 0, lib=lib)
    qn, qs, qq = read_fastx(os.path.join(GOLDEN, "adv_sub.fq.gz"))
    s_reads = [[a, b.tobytes().decode(), c.tobytes().decode()] for a, b, c in zip(qn, qs, qq)]
    text = sampleqc.coverage_in_memory(_as_chunks(os.path.join(GOLDEN, "adv_all.fa.gz"), 37), s_reads, inds=100000, engine=eng)
    eng.close()
    assert text == read_gz("adv_parts.table.gz")          # same 10 parts, same COVT behaviour as the file-based run


def test_in_memory_coverage_equals_file_based(emu_lib):
    check_in_memory_equals_file_path(emu_lib)


# ---- the consumer's reading of the table ----
def _consumer_cases():
    import json
    return json.load(open(os.path.join(GOLDEN, "consumer.json")))


@pytest.mark.parametrize("case", _consumer_cases(), ids=lambda c: c["table"] + ("+ctl" if c["control"] else ""))
def test_table_reads_like_the_reference_consumer(case, tmp_path):
    """tests/golden/consumer.json = what the reference's LqCoverage computes from each golden table
    (make_consumer_golden.py); CoverageTable must give the same figures from the same bytes."""
    import gzip
    from longqc_amd.covtable import CoverageTable
    t = tmp_path / "t.txt"; t.write_bytes(gzip.open(os.path.join(GOLDEN, case["table"])).read())
    c = None
    if case["control"]:
        c = tmp_path / "c.txt"; c.write_bytes(gzip.open(os.path.join(GOLDEN, case["control"])).read())
    ct = CoverageTable(str(t), control_filtering=str(c) if c else None)
    assert len(ct) == case["n_rows"] and ct.control_reads == case["control_reads"]
    assert ct.unmapped_frac_trimmed == case["unmapped_frac_trimmed"]
    assert ct.unmapped_frac_untrimmed == case["unmapped_frac_untrimmed"]
    assert ct.get_unmapped_med_frac() == case["unmapped_med_frac"]
    assert ct.get_high_div_frac() == case["high_div_frac"]
    assert ct.get_control_num() == case["control_num"] and ct.get_control_frac() == case["control_frac"]
