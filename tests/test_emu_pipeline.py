"""Kernel-logic tests without a GPU: the product's kernel sources, compiled as plain C++ against the
serial HIP stand-in (tests/emu), must reproduce the reference's golden vectors and agree with the
oracle bit for bit.  This validates indexing, the sketch state machine with its halo warm-up, the
klib-order sort walk, chaining and coverage arithmetic -- not the GPU execution itself (the -m gpu
tests do that through the real liblqcov.so)."""
import json
import os
import re

import numpy as np
import pytest

from longqc_amd import api
from tests import oracle_bind
from tests.conftest import GOLDEN, read_gz
from tests.helpers import ONT, parse_chain_dump, parse_sketch_dump, read_fastx, run_main, slow_emu


def _cases(kind):
    return [c for c in json.load(open(os.path.join(GOLDEN, "cases.json"))) if c["kind"] == kind]


@pytest.mark.parametrize("case", [pytest.param(c, marks=slow_emu) if c["name"] in ("adv_defaults", "adv_fasta_query") else c for c in _cases("table")], ids=lambda c: c["name"])
def test_emulated_pipeline_reproduces_reference_tables(emu_lib, case):
    rc, out, err = run_main(emu_lib, case["argv"], cwd=GOLDEN)
    assert rc == 0, err
    assert out == read_gz(case["expect"])


def test_emulated_tables_read_like_the_reference_consumer(emu_lib, tmp_path):
    """the table the engine writes (main call + spike-in call), read by the consumer's parser, gives the
    figures the reference's LqCoverage computed from the reference's table (tests/golden/consumer.json)."""
    from longqc_amd.covtable import CoverageTable
    by_name = {c["name"]: c for c in _cases("table")}
    want = [c for c in json.load(open(os.path.join(GOLDEN, "consumer.json"))) if c["control"]]
    assert want
    for w in want:
        paths = []
        for fixture in (w["table"], w["control"]):
            case = by_name[fixture[:-len(".table.gz")]]
            rc, out, err = run_main(emu_lib, case["argv"], cwd=GOLDEN)
            assert rc == 0, err
            fn = tmp_path / fixture[:-3]; fn.write_text(out); paths.append(str(fn))
        ct = CoverageTable(paths[0], control_filtering=paths[1])
        assert (len(ct), ct.control_reads, ct.unmapped_frac_trimmed, ct.unmapped_frac_untrimmed, ct.get_unmapped_med_frac(), ct.get_high_div_frac(),
                ct.get_control_frac()) == (w["n_rows"], w["control_reads"], w["unmapped_frac_trimmed"], w["unmapped_frac_untrimmed"], w["unmapped_med_frac"],
                                           w["high_div_frac"], w["control_frac"])


@pytest.mark.parametrize("case", [pytest.param(c, marks=slow_emu) if c["name"] == "adv_ont" else c for c in _cases("table") if c["name"] in ("tiny_ont", "adv_ont", "adv_parts")], ids=lambda c: c["name"])
@pytest.mark.parametrize("tile", ["64", pytest.param("1000", marks=slow_emu)])
def test_emulated_every_query_through_klib_passes(emu_lib, case, tile, monkeypatch):
    """LQCOV_SORT=klib: every query goes through klib's passes as 8-byte records, no bucket leaves them early (the passes on
    the bytes of the position run too, reading their digits from the originals); streaming kernels over many small tiles"""
    monkeypatch.setenv("LQCOV_SORT_TILE", tile); monkeypatch.setenv("LQCOV_SORT", "klib")
    rc, out, err = run_main(emu_lib, case["argv"], cwd=GOLDEN)
    assert rc == 0, err
    assert out == read_gz(case["expect"])


def _engine(lib, **kw):
    p = api.Params()
    lib.lqcov_params_default(p)
    p.no_self = 1; p.min_ovlp = 0; p.min_score_med = 160; p.min_score_good = 160
    for k, v in kw.items():
        setattr(p, k, v)
    return api.Engine(p, 0, lib=lib)


@pytest.mark.parametrize("name,k,w,hpc,fn", [
    ("tiny_sketch_k12w5", 12, 5, 0, "tiny_sub.fq.gz"),
    ("adv_sketch_k12w5", 12, 5, 0, "adv_sub.fq.gz"),
    ("adv_sketch_k15w10hpc", 15, 10, 1, "adv_sub.fq.gz"),
    ("adv_sketch_k19w10", 19, 10, 0, "adv_sub.fq.gz"),
    ("adv_sketch_k6w30", 6, 30, 0, "adv_sub.fq.gz"),
])
def test_emulated_sketch_equals_mm_sketch_fixture(emu_lib, name, k, w, hpc, fn):
    want = parse_sketch_dump(read_gz(name + ".dump.gz"))
    names, seqs, quals = read_fastx(os.path.join(GOLDEN, fn))
    eng = _engine(emu_lib, k=k, w=w, hpc=hpc, min_score_med=40, min_score_good=40)
    eng.set_queries(names, seqs, quals)
    xy, off = eng.query_minimizers()
    assert len(want) == len(names)
    for i, (nm, ln, mm) in enumerate(want):
        got = [(int(x), int(y)) for x, y in xy[int(off[i]):int(off[i + 1])]]
        # the harness sketches read i with rid=i; queries are sketched with rid 0 (lqmap.c:131)
        exp = [(x, y & 0xffffffff) for x, y in mm]
        assert got == exp, (nm, ln)
    eng.close()


@pytest.mark.parametrize("name,k,w,fn", [("tiny_sketch_k12w5", 12, 5, "tiny_sub.fq.gz"), ("adv_sketch_k12w5", 12, 5, "adv_sub.fq.gz"), ("adv_sketch_k19w10", 19, 10, "adv_sub.fq.gz")])
def test_emulated_state_machine_decides_every_chunk(emu_lib, monkeypatch, name, k, w, fn):
    """LQCOV_SKETCH=machine: no chunk is decided by the data-parallel kernel (k_sketch_dp_mask: sliding-window minimum over the
    non-palindromic positions with minimap2's tie rules, the default wherever the machine is memoryless): the reference's list"""
    monkeypatch.setenv("LQCOV_SKETCH", "machine")
    test_emulated_sketch_equals_mm_sketch_fixture(emu_lib, name, k, w, 0, fn)


def test_emulated_state_machine_only_adversarial(emu_lib, monkeypatch):
    monkeypatch.setenv("LQCOV_SKETCH", "machine")
    test_emulated_sketch_halo_adversarial(emu_lib)


def test_emulated_data_parallel_sketch_fuzz(emu_lib, monkeypatch):
    """random reads with sparse Ns, AT stretches (palindromic k-mers), short-period repeats and tandem copies (ties), a
    two-letter alphabet; random (k, w): the data-parallel kernel and the state machine give the same list"""
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    fixed = [(12, 5), (12, 10), (19, 5), (19, 10)]       # the windows k_sketch_dp_mask knows at compile time, 32- and 64-bit hashes
    for it in range(10 + len(fixed)):
        rng = np.random.default_rng(100 + it)
        k = int(rng.choice([4, 6, 10, 12, 15, 19, 24, 28])); w = int(rng.choice([1, 2, 3, 5, 10, 16]))
        if it >= 10:
            k, w = fixed[it - 10]
        seqs = []
        for j in range(6):
            L = int(rng.integers(130, 3000)) if j < 4 else int(rng.integers(3000, 9000))    # (a tile of the data-parallel kernel is 1536 bases)
            s = A[rng.integers(0, 4, L)].copy()
            mode = int(rng.integers(0, 6))
            if mode == 1:
                s[rng.integers(0, L, max(1, L // 300))] = ord("N")
            elif mode == 2:
                a = int(rng.integers(0, L - 100)); n = int(rng.integers(20, 100)); s[a:a + n] = np.frombuffer((b"AT" * 60)[:n], np.uint8)
            elif mode == 3:
                u = int(rng.integers(1, 7)); a = int(rng.integers(0, L - 120)); s[a:a + 120] = np.tile(s[a:a + u], 120 // u + 1)[:120]
            elif mode == 4:
                s = np.frombuffer(b"AC", np.uint8)[rng.integers(0, 2, L)].copy()
            elif mode == 5:
                a = int(rng.integers(0, L - 125)); s[a:a + 120] = np.tile(s[a:a + 40], 3)
            seqs.append(s)
        names = ["s%d" % i for i in range(len(seqs))]
        res = []
        for mode, wgen in (("machine", "0"), ("default", "0")) + ((("default", "1"),) if w in (5, 10) else ()):   # (the same window read at run time)
            monkeypatch.setenv("LQCOV_SKETCH", mode); monkeypatch.setenv("LQCOV_SKETCH_WGEN", wgen)
            eng = _engine(emu_lib, k=k, w=w, hpc=0, min_score_med=40, min_score_good=40)
            eng.set_queries(names, seqs, None)
            xy, off = eng.query_minimizers()
            res.append((np.array(xy).copy(), np.array(off).copy()))
            eng.close()
        for other in res[1:]:
            assert np.array_equal(res[0][0], other[0]) and np.array_equal(res[0][1], other[1]), (it, k, w)


def test_emulated_constant_k_sketch_fuzz(emu_lib, monkeypatch):
    """k_sketch_dp_fast (-k 12 with -w 5 / 10: k and w as constants, straight-line steps for threads without palindromes and ties,
    the literal rules for the others) against the state machine and against the general kernel (LQCOV_SKETCH_FAST=0): reads of
    several tiles with tandem repeats of short periods (tied minima inside a window), two- and three-letter alphabets, AT and
    GC stretches (palindromes), Ns, reads that end inside and right after a tile"""
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    for it in range(12):
        rng = np.random.default_rng(700 + it)
        w = 5 if it % 3 else 10
        seqs = []
        for j in range(8):
            L = [1536 + 128, 1536 * 2 + 128 - 1, 1536 + 128 + 5, 1664 + 64][j] if j < 4 else int(rng.integers(2000, 12000))
            s = A[rng.integers(0, 4, L)].copy()
            for _ in range(int(rng.integers(0, 12))):
                mode = int(rng.integers(0, 7))
                a = int(rng.integers(0, L - 200)); n = int(rng.integers(20, 200))
                if mode == 0:
                    u = int(rng.integers(1, 9)); s[a:a + n] = np.tile(s[a:a + u], n // u + 1)[:n]          # tandem repeat, period 1..8
                elif mode == 1:
                    s[a:a + n] = np.frombuffer((b"AT" * 100)[:n], np.uint8)
                elif mode == 2:
                    s[a:a + n] = np.frombuffer(b"AC", np.uint8)[rng.integers(0, 2, n)]
                elif mode == 3:
                    s[a:a + n] = np.tile(s[a:a + 13], n // 13 + 1)[:n]                                    # period k + 1
                elif mode == 4:
                    s[a] = ord("N")
                elif mode == 5:
                    s[a:a + n] = np.frombuffer((b"GC" * 100)[:n], np.uint8)
                else:
                    h = n // 2; s[a + h:a + 2 * h] = (3 - (s[a:a + h][::-1] == A[:, None]).argmax(0)).astype(np.uint8).choose(A)   # a reverse-complement copy next to its source
            seqs.append(s)
        names = ["s%d" % i for i in range(len(seqs))]
        res = []
        for mode, fast in (("machine", "1"), ("default", "1"), ("default", "0")):
            monkeypatch.setenv("LQCOV_SKETCH", mode); monkeypatch.setenv("LQCOV_SKETCH_FAST", fast)
            eng = _engine(emu_lib, k=12, w=w, hpc=0, min_score_med=40, min_score_good=40)
            eng.set_queries(names, seqs, None)
            xy, off = eng.query_minimizers()
            res.append((np.array(xy).copy(), np.array(off).copy()))
            eng.close()
        for other in res[1:]:
            assert np.array_equal(res[0][0], other[0]) and np.array_equal(res[0][1], other[1]), (it, w)


def test_emulated_sketch_halo_adversarial(emu_lib):
    """palindromic / N-rich / homopolymer contexts around every chunk boundary: the warm-up must widen its halo"""
    rng = np.random.default_rng(5)
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    seqs.append(np.frombuffer(b"AT" * 700, dtype=np.uint8))                                  # all 12-mers palindromic
    seqs.append(np.concatenate([A[rng.integers(0, 4, 100)], np.frombuffer(b"AT" * 400, np.uint8), A[rng.integers(0, 4, 600)]]))
    s = A[rng.integers(0, 4, 2000)].copy(); s[::37] = ord("N"); seqs.append(s)                # N every 37 bases
    s = A[rng.integers(0, 4, 1500)].copy(); s[120:135] = ord("N"); s[250:390] = ord("N"); seqs.append(s)
    seqs.append(np.repeat(A[rng.integers(0, 4, 120)], rng.integers(1, 40, 120)))              # long homopolymers
    seqs.append(np.frombuffer(b"A" * 1000, dtype=np.uint8))
    seqs.append(np.frombuffer(b"ACG" * 500, dtype=np.uint8))
    seqs.append(np.concatenate([np.frombuffer(b"G" * 300, np.uint8), A[rng.integers(0, 4, 300)], np.frombuffer(b"TA" * 200, np.uint8)]))
    names = ["s%d" % i for i in range(len(seqs))]
    import tempfile
    from longqc_amd import synth
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "x.fa")
        synth.write_fastq(fn, synth.ReadSet(names, seqs, [None] * len(seqs)), fasta=True)
        for k, w, hpc in [(12, 5, 0), (15, 10, 1), (4, 3, 0), (28, 50, 0), (10, 1, 0), (15, 5, 1)]:
            want = parse_sketch_dump(oracle_bind.dump("sketch", ["-k", str(k), "-w", str(w)] + (["-H"] if hpc else []), [fn]))
            eng = _engine(emu_lib, k=k, w=w, hpc=hpc, min_score_med=40, min_score_good=40)
            eng.set_queries(names, seqs, None)
            xy, off = eng.query_minimizers()
            for i, (nm, ln, mm) in enumerate(want):
                got = [(int(x), int(y)) for x, y in xy[int(off[i]):int(off[i + 1])]]
                assert got == [(x, y & 0xffffffff) for x, y in mm], (k, w, hpc, nm)
            eng.close()


def _canon_chains(arr, q):
    return sorted(tuple(int(v) for v in r[1:]) for r in arr if r[0] == q)


@pytest.mark.parametrize("name,tfn,qfn", [("tiny_chains", "tiny_all.fq.gz", "tiny_sub.fq.gz"), ("adv_chains", "adv_all.fa.gz", "adv_sub.fq.gz")])
def test_emulated_chains_mid_occ_and_accumulators(emu_lib, name, tfn, qfn):
    mid, want = parse_chain_dump(read_gz(name + ".dump.gz"))
    tn, ts, _ = read_fastx(os.path.join(GOLDEN, tfn))
    qn, qs, qq = read_fastx(os.path.join(GOLDEN, qfn))
    eng = _engine(emu_lib)
    eng.set_debug(1)
    eng.set_queries(qn, qs, qq)
    pt = eng.part_begin()
    half = len(tn) // 2
    eng.part_add_targets(pt, tn[:half], ts[:half])             # two uploads, one part (mini-batches of mm_idx_gen)
    eng.part_add_targets(pt, tn[half:], ts[half:])
    eng.part_build(pt)
    assert eng.mid_occ == mid
    eng.part_map(pt)
    ch = eng.chains()
    eng.finish()
    rows = eng.rows()
    for qi, w in want.items():
        # (rid, rev, score, cnt, qs, qe, rs, re)
        assert _canon_chains(ch, qi) == sorted(w["chains"]), (qi, w["name"])
        assert rows[qi]["lambda_"] == w["lambda_"] and rows[qi]["lambda2"] == w["lambda2"], w["name"]
    eng.close()


@pytest.mark.parametrize("lanes", ["1", pytest.param("3", marks=slow_emu)])
def test_emulated_query_batching_is_invisible(emu_lib, datasets, monkeypatch, lanes):
    """many small query batches dealt to 1..3 mapping lanes (own stream + work space each; concurrent threads on the GPU,
    round-robin in the emulator): same table"""
    tf, qf = datasets("small")
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "700K", "-p", "160", tf, qf]
    want = oracle_bind.table(argv)
    monkeypatch.setenv("LQCOV_ANCHOR_BUDGET", "3000")
    monkeypatch.setenv("LQCOV_LANES", lanes)
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == want


def check_reset_and_rerun(emu_lib, packed):
    tn, ts, _ = read_fastx(os.path.join(GOLDEN, "tiny_all.fq.gz"))
    qn, qs, qq = read_fastx(os.path.join(GOLDEN, "tiny_sub.fq.gz"))
    if packed:      # (packed reads go up in slices with the data-parallel sketch kernel under the upload: the second build finds them resident)
        os.environ["LQCOV_UPLOAD_MIN_CHUNKS"] = "1"; os.environ["LQCOV_UPLOAD_SLICES"] = "3"
    try:
        eng = _engine(emu_lib)
    finally:
        os.environ.pop("LQCOV_UPLOAD_MIN_CHUNKS", None); os.environ.pop("LQCOV_UPLOAD_SLICES", None)
    eng.set_queries(qn, qs, qq)
    pt = eng.part_begin()
    if packed:
        flat = np.concatenate(ts); off = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.uint64)
        eng.part_add_packed(pt, api.PackedReads(flat, off, list(tn), lib=emu_lib))
    else:
        eng.part_add_targets(pt, tn, ts)
    tables = []
    for _ in range(2):
        eng.reset()
        eng.part_build(pt); eng.part_map(pt); eng.finish()
        tables.append(eng.table_text())
    assert tables[0] == tables[1] == read_gz("tiny_ont.table.gz")
    eng.close()


@pytest.mark.parametrize("packed", [False, True], ids=["ascii", "packed_in_slices"])
def test_emulated_reset_and_rerun(emu_lib, packed):
    check_reset_and_rerun(emu_lib, packed)


@pytest.mark.parametrize("slices", ["1", "3"])
def test_emulated_packed_upload_without_ambiguity_words(emu_lib, monkeypatch, slices):
    """lqcov_part_add_packed with amb == NULL (round 6: a range of reads without an ambiguous base goes up as codes alone, the device
    makes the bits past the reads' ends itself): the adversarial set -- reads with N runs among reads without -- added read range by
    read range, so that some ranges carry their ambiguity words and others do not; the reference's table.  Also: the flags say what
    the reads hold, and forcing the words up (LQCOV_UPLOAD_AMB=1) changes nothing."""
    tn, ts, _ = read_fastx(os.path.join(GOLDEN, "adv_all.fa.gz"))
    qn, qs, qq = read_fastx(os.path.join(GOLDEN, "adv_sub.fq.gz"))
    flat = np.concatenate(ts); off = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.uint64)
    P = api.PackedReads(flat, off, list(tn), lib=emu_lib)
    acgt = np.zeros(256, bool); acgt[[ord(c) for c in "ACGTUacgtu"]] = True
    want_flags = np.array([0 if acgt[x].all() else 1 for x in ts], dtype=np.uint8)
    assert np.array_equal(P.has_amb[:len(ts)], want_flags) and 0 < want_flags.sum() < len(ts)
    want = read_gz("adv_ont.table.gz")
    for force in ("0", "1"):
        monkeypatch.setenv("LQCOV_UPLOAD_AMB", force)
        monkeypatch.setenv("LQCOV_UPLOAD_MIN_CHUNKS", "1"); monkeypatch.setenv("LQCOV_UPLOAD_SLICES", slices)
        eng = _engine(emu_lib)
        eng.set_queries(qn, qs, qq)
        pt = eng.part_begin()
        cuts = sorted(set([0, len(tn)] + [int(i) for i in np.flatnonzero(np.diff(want_flags.astype(np.int8)))[:6] + 1]))
        sent = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sent.append(P.any_ambiguous(lo, hi))
            eng.part_add_packed(pt, P, lo, hi)
        assert True in sent and False in sent
        eng.reset(); eng.part_build(pt); eng.part_map(pt); eng.finish()
        assert eng.table_text() == want, force
        eng.close()


@pytest.mark.parametrize("order", ["queries_after_add", "two_parts_added_first", "clear_and_add_again"])
def test_emulated_packed_upload_mask_belongs_to_one_read_set(emu_lib, order):
    """The data-parallel sketch kernel runs under the upload of packed reads into mask buffers the handle shares between read sets
    (round-5 advice): whatever is sketched between a part's add_packed and its build -- the queries, another part -- must not leave
    the part with somebody else's mask.  Legal API orders the in-repo drivers never use."""
    tn, ts, _ = read_fastx(os.path.join(GOLDEN, "tiny_all.fq.gz"))
    qn, qs, qq = read_fastx(os.path.join(GOLDEN, "tiny_sub.fq.gz"))
    os.environ["LQCOV_UPLOAD_MIN_CHUNKS"] = "1"; os.environ["LQCOV_UPLOAD_SLICES"] = "3"
    try:
        eng = _engine(emu_lib)
    finally:
        os.environ.pop("LQCOV_UPLOAD_MIN_CHUNKS", None); os.environ.pop("LQCOV_UPLOAD_SLICES", None)
    flat = np.concatenate(ts); off = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.uint64)
    P = api.PackedReads(flat, off, list(tn), lib=emu_lib)
    want = read_gz("tiny_ont.table.gz")
    if order == "queries_after_add":
        pt = eng.part_begin()
        eng.part_add_packed(pt, P)
        eng.set_queries(qn, qs, qq)                       # (sketches the queries: the shared buffers change hands)
        eng.reset(); eng.part_build(pt); eng.part_map(pt); eng.finish()
        assert eng.table_text() == want
    elif order == "two_parts_added_first":
        eng.set_queries(qn, qs, qq)
        half = len(tn) // 2
        a, b = eng.part_begin(), eng.part_begin()
        eng.part_add_packed(a, P, 0, len(tn))             # the whole set in a ...
        eng.part_add_packed(b, P, 0, half)                # ... and another read set added before a is built
        eng.reset(); eng.part_build(a); eng.part_map(a); eng.finish()
        assert eng.table_text() == want
    else:
        eng.set_queries(qn, qs, qq)
        pt = eng.part_begin()
        eng.part_add_packed(pt, P, 0, len(tn) // 2)
        eng.part_clear(pt)
        eng.part_add_packed(pt, P)                        # the same part object, other reads
        eng.reset(); eng.part_build(pt); eng.part_map(pt); eng.finish()
        assert eng.table_text() == want
    eng.close()


@pytest.mark.parametrize("shift", ["4", pytest.param("7", marks=slow_emu), pytest.param("12", marks=slow_emu)])
def test_emulated_every_walk_size_class(emu_lib, datasets, monkeypatch, shift):
    """shrinks the LDS-window thresholds of the klib-order sort so that small inputs exercise every size class
    of the digit walk, including the global-memory lane walker used when a sub-array exceeds 156 KiB"""
    tf, qf = datasets("small")
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "2M", "-p", "160", tf, qf]
    want = oracle_bind.table(argv)
    monkeypatch.setenv("LQCOV_WALK_SHIFT", shift)
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == want


@pytest.mark.parametrize("case", [pytest.param(c, marks=slow_emu) if c["name"] == "adv_ont" else c for c in _cases("table") if c["name"] in ("tiny_ont", "adv_ont", "adv_parts", "tiny_spike")], ids=lambda c: c["name"])
def test_emulated_wave_chain_kernel_on_every_run(emu_lib, case, monkeypatch):
    """LQCOV_CHAIN_WAVE_MIN=3 sends every viable run through the cooperative (64 candidates per step) chain kernel"""
    monkeypatch.setenv("LQCOV_CHAIN_WAVE_MIN", "3")
    rc, out, err = run_main(emu_lib, case["argv"], cwd=GOLDEN)
    assert rc == 0, err
    assert out == read_gz(case["expect"])


@pytest.mark.parametrize("case", [pytest.param(c, marks=slow_emu) if c["name"] == "adv_ont" else c for c in _cases("table") if c["name"] in ("tiny_ont", "adv_ont", "adv_parts", "tiny_spike")], ids=lambda c: c["name"])
def test_emulated_chain_lds_budget_overflow(emu_lib, case, monkeypatch):
    """LQCOV_CHAIN_CAP=64 with LQCOV_CHAIN_WAVE_MIN=200: the smallest LDS budget of k_chain (64 anchors per wave) with every run
    up to 64 anchors in that kernel: the runs of a wave are chained in several rounds"""
    monkeypatch.setenv("LQCOV_CHAIN_CAP", "64")
    monkeypatch.setenv("LQCOV_CHAIN_WAVE_MIN", "200")
    rc, out, err = run_main(emu_lib, case["argv"], cwd=GOLDEN)
    assert rc == 0, err
    assert out == read_gz(case["expect"])


@slow_emu
def test_emulated_constant_digit_levels_can_be_walked_or_skipped(emu_lib, datasets, monkeypatch):
    """levels of the klib-order sort whose key byte is the same in every anchor of a part are stepped over by default;
    LQCOV_NO_LEVEL_SKIP=1 runs them as identity passes like klib does: same table"""
    tf, qf = datasets("small")
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "700K", "-p", "160", tf, qf]
    want = oracle_bind.table(argv)
    monkeypatch.setenv("LQCOV_NO_LEVEL_SKIP", "1")
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == want


@pytest.mark.parametrize("order", ["file", "striped"])
@pytest.mark.parametrize("case", [c for c in _cases("table") if c["name"] in ("adv_parts", "tiny_ont")], ids=lambda c: c["name"])
def test_emulated_query_order_is_internal_only(emu_lib, case, monkeypatch, order):
    """the engine holds the queries longest first; LQCOV_QUERY_ORDER=file keeps the caller's order inside as well, =striped deals the
    sorted queries to the lanes' stripes: same rows, in the caller's order, either way"""
    monkeypatch.setenv("LQCOV_QUERY_ORDER", order)
    rc, out, err = run_main(emu_lib, case["argv"], cwd=GOLDEN)
    assert rc == 0, err
    assert out == read_gz(case["expect"])


def _few_targets_dataset(tmp_path, n_targets=14, tlen=25000, n_queries=6, qlen=20000, glen=30000, seed=11):
    """few, long targets over a small genome: the pass on the rid byte has <= 16 buckets and every (query, strand)
    sub-array holds tens of thousands of anchors -- the shape of the top pass of a 4-Gbase part (kernels_ckpt.hpp)"""
    from longqc_amd import synth
    rng = np.random.default_rng(seed)
    A = synth._ACGT
    g = A[rng.integers(0, 4, size=glen, dtype=np.uint8)]
    g[5000:5400] = np.tile(g[4900:5000], 4)                              # a tandem repeat: repeated minimizers inside queries
    names, seqs, quals = [], [], []

    def read(L):
        st = int(rng.integers(0, glen - L + 1)) if L < glen else 0
        s = g[st:st + L]
        if rng.random() < 0.5:
            s = synth._COMP[s[::-1]]
        return synth._mutate(s, rng, 0.06, (3, 3, 4))
    for i in range(n_targets):
        s = read(min(tlen, glen)); names.append("t%02d" % i); seqs.append(s); quals.append((33 + rng.integers(3, 30, size=s.shape[0])).astype(np.uint8))
    T = synth.ReadSet(names, seqs, quals)
    qn, qs, qq = [], [], []
    for i in range(n_queries):
        s = read(min(qlen, glen)); qn.append("q%02d" % i); qs.append(s); qq.append((33 + rng.integers(3, 30, size=s.shape[0])).astype(np.uint8))
    Q = synth.ReadSet(qn, qs, qq)
    tf, qf = str(tmp_path / "ft_all.fq"), str(tmp_path / "ft_sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    return tf, qf


@pytest.mark.parametrize("variant", ["ckpt", pytest.param("ckpt_all_klib", marks=slow_emu), "plain"])
def test_emulated_checkpointed_walks(emu_lib, tmp_path, monkeypatch, variant):
    """long sub-arrays of a few-bucket pass are walked in pieces from computed checkpoint states (kernels_ckpt.hpp):
    same table as the oracle, with the size classes shrunk so that this small input reaches them"""
    tf, qf = _few_targets_dataset(tmp_path, n_queries=3, qlen=17000)
    argv = ONT + [tf, qf]
    want = oracle_bind.table(argv)
    monkeypatch.setenv("LQCOV_WALK_SHIFT", "4")
    if variant == "ckpt_all_klib":
        monkeypatch.setenv("LQCOV_SORT", "klib")
    if variant == "plain":
        monkeypatch.setenv("LQCOV_CKPT", "0")
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == want
    assert sum(1 for l in out.splitlines() if l.split("\t")[2] != "0") >= 2


def _many_short_queries_dataset(tmp_path, n_targets=60, n_queries=500, seed=5):
    """hundreds of queries of 150-500 bases: a few dozen anchors each, so that dozens of queries start inside one
    4096-anchor tile of the run list (kernels_chain.hpp: k_run_list), some of them without any anchor"""
    from longqc_amd import synth
    rng = np.random.default_rng(seed)
    A = synth._ACGT
    g = A[rng.integers(0, 4, size=30000, dtype=np.uint8)]

    def read(lo, hi, err):
        L = int(rng.integers(lo, hi)); st = int(rng.integers(0, g.shape[0] - L))
        s = g[st:st + L]
        if rng.random() < 0.5:
            s = synth._COMP[s[::-1]]
        return synth._mutate(s, rng, err, (3, 3, 4))

    def readset(prefix, seqs):
        return synth.ReadSet(["%s%04d" % (prefix, i) for i in range(len(seqs))], seqs, [(33 + rng.integers(3, 30, size=x.shape[0])).astype(np.uint8) for x in seqs])
    T = readset("t", [read(2000, 6000, 0.05) for _ in range(n_targets)])
    qs = [read(150, 500, 0.03) for _ in range(n_queries)]
    for i in range(0, n_queries, 17):
        qs[i] = A[rng.integers(0, 4, size=int(rng.integers(30, 200)), dtype=np.uint8)]      # unrelated: no anchors at all
    Q = readset("q", qs)
    tf, qf = str(tmp_path / "msq_all.fq"), str(tmp_path / "msq_sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    return tf, qf


def check_run_list_variants(lib, tmp_path, monkeypatch):
    tf, qf = _many_short_queries_dataset(tmp_path)
    argv = ["-Y", "-l", "0", "-q", "40", "-k", "12", "-w", "5", "-I", "4G", "-p", "40", "-m", "20", "-t", "4", tf, qf]
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    assert sum(1 for l in want.splitlines() if l.split("\t")[2] != "0") >= 100
    for budget in ("", "3000"):                    # one batch | batches of a few queries (run lists of a few tiles)
        monkeypatch.setenv("LQCOV_ANCHOR_BUDGET", budget) if budget else monkeypatch.delenv("LQCOV_ANCHOR_BUDGET", raising=False)
        rc, out, err = run_main(lib, argv)
        assert rc == 0, err
        assert out == want, budget
    monkeypatch.delenv("LQCOV_ANCHOR_BUDGET", raising=False)
    # a block walks all the tiles | a third of them (the tile's query from the previous tile, entries of several tiles staged
    # together) | grids that leave the last blocks without a tile; the emulator's threads once in descending order (a block
    # without tiles read the LDS counter before thread 0 had cleared it: on the GPU, whatever another kernel left there)
    for grid, order in (("1", ""), ("3", ""), ("7", "reverse"), ("10", "reverse"), ("13", ""), ("29", "reverse")):
        monkeypatch.setenv("LQCOV_RUN_GRID", grid)
        monkeypatch.setenv("LQ_EMU_ORDER", order)
        rc, out, err = run_main(lib, argv)
        assert rc == 0, err
        assert out == want, grid
    monkeypatch.delenv("LQ_EMU_ORDER", raising=False)
    # the list's LDS staging (k_run_list): room for 256 entries (the least it takes), so that a block's tiles flush it again
    # and again, row by row where a tile may hold more entries than that; -n 1 -m 10: every run of one anchor is listed
    monkeypatch.setenv("LQCOV_RUN_STAGE", "256")
    rc, out, err = run_main(lib, argv)
    assert rc == 0, err
    assert out == want
    argv1 = ["-Y", "-l", "0", "-q", "40", "-k", "12", "-w", "5", "-I", "4G", "-p", "40", "-m", "10", "-n", "1", "-t", "4", tf, qf]
    want1 = oracle_bind.ref_table(argv1) if oracle_bind.have_ref() else oracle_bind.table(argv1)
    for stage in ("256", ""):                      # (still with three blocks)
        monkeypatch.setenv("LQCOV_RUN_STAGE", stage) if stage else monkeypatch.delenv("LQCOV_RUN_STAGE", raising=False)
        rc, out, err = run_main(lib, argv1)
        assert rc == 0, err
        assert out == want1, stage


@slow_emu
def test_emulated_run_list_variants(emu_lib, tmp_path, monkeypatch):
    check_run_list_variants(emu_lib, tmp_path, monkeypatch)


def _repeat_rich_dataset(tmp_path, seed, n_targets=60, n_queries=8, glen=40000):
    """a genome full of tandem repeats (unit 30-300, 3-12 copies) and dispersed copies of 300-1500-base segments, reads with
    few errors: most queries carry the same (minimizer, strand) several times, so anchors with equal x -- whose final order is
    klib's and decides chains and rows -- are everywhere"""
    from longqc_amd import synth
    rng = np.random.default_rng(1000 + seed)
    A = synth._ACGT
    g = A[rng.integers(0, 4, size=glen, dtype=np.uint8)]
    for _ in range(12):
        u = int(rng.integers(30, 300)); c = int(rng.integers(3, 12)); at = int(rng.integers(0, glen - u * c))
        g[at:at + u * c] = np.tile(g[at:at + u], c)
    for _ in range(10):
        L = int(rng.integers(300, 1500)); src = int(rng.integers(0, glen - L)); dst = int(rng.integers(0, glen - L))
        seg = g[src:src + L].copy()
        g[dst:dst + L] = synth._COMP[seg[::-1]] if rng.random() < 0.5 else seg
    err = float(rng.choice([0.01, 0.04, 0.08]))

    def read(lo, hi):
        L = int(rng.integers(lo, hi)); st = int(rng.integers(0, glen - L))
        s = g[st:st + L]
        if rng.random() < 0.5:
            s = synth._COMP[s[::-1]]
        return synth._mutate(s, rng, err, (3, 3, 4))

    def readset(prefix, n, lo, hi):
        seqs = [read(lo, hi) for _ in range(n)]
        return synth.ReadSet(["%s%03d" % (prefix, i) for i in range(n)], seqs, [(33 + rng.integers(3, 30, size=x.shape[0])).astype(np.uint8) for x in seqs])
    T, Q = readset("t", n_targets, 3000, 12000), readset("q", n_queries, 6000, 20000)
    tf, qf = str(tmp_path / ("rr%d_all.fq" % seed)), str(tmp_path / ("rr%d_sub.fq" % seed))
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    return tf, qf


def check_repeat_rich_randomised(lib, tmp_path, monkeypatch, seed):
    tf, qf = _repeat_rich_dataset(tmp_path, seed)
    argv = ONT + [tf, qf]
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    # the size classes of the walks and of the parallel sort move with the seed, so that small inputs reach each of them
    monkeypatch.setenv("LQCOV_WALK_SHIFT", str([0, 4, 7, 10][seed % 4]))
    monkeypatch.setenv("LQCOV_PS_SHIFT", str([0, 3, 7][seed % 3]))
    monkeypatch.setenv("LQCOV_SORT_TILE", str([8192, 64, 1000][seed % 3]))      # tiles of the streaming kernels: one per sub-array | many
    monkeypatch.setenv("LQCOV_CKPT3", str(seed & 1))          # odd seeds: checkpoints for the second-longest class of many-bucket passes too
    rc, out, err = run_main(lib, argv)
    assert rc == 0, err
    assert out == want
    return argv, want


@pytest.mark.parametrize("seed", [pytest.param(0, marks=slow_emu), pytest.param(1, marks=slow_emu)])
def test_emulated_repeat_rich_randomised(emu_lib, tmp_path, monkeypatch, seed):
    """equal-x anchors everywhere (repeats inside the queries): rows equal the reference binary's, whichever path the sub-arrays
    take (parallel passes for those without a tie, klib's walk for the others); and the input has teeth: a stable sort by x
    gives a different table"""
    argv, want = check_repeat_rich_randomised(emu_lib, tmp_path, monkeypatch, seed)
    assert sum(1 for l in want.splitlines() if l.split("\t")[2] != "0") >= 4
    if seed == 0:
        assert oracle_bind.table(argv, ["--stable-sort"]) != want


@slow_emu
def test_emulated_ultra_long_reads(emu_lib, tmp_path):
    """reads of 100-350 kb at 30x (the shape of BASELINE configs[4]): (query, strand) sub-arrays of 10^5 anchors and runs
    of thousands per target; the three longest as queries, against the reference binary (or the oracle)"""
    import dataclasses
    from longqc_amd import synth
    cfg = dataclasses.replace(synth.CONFIGS["cfg5"], n_reads=200, nsample=200, depth=30.0)
    genome = synth.make_genome(cfg)
    T = synth.make_reads(cfg, genome)
    order = sorted(range(len(T)), key=lambda i: -int(T.seqs[i].shape[0]))[:3]
    Q = T.subset(order)
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    argv = ONT + [tf, qf]
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    assert out == want


def _one_long_pair_among_many_targets(tmp_path, n_small=6000, seed=3):
    """6000 random 200-base targets plus one 30-kb read; the query is a 99.9 % copy of it: one tie-free (strand, rid) sub-array of
    more than 8192 anchors whose compact key agrees in its top 1 + 13 bits -- the parallel sort has to step over the constant bits
    (round-2 advisor: the fixed six partition passes never reached the position bits and the run ended with rc -5)"""
    from longqc_amd import synth
    rng = np.random.default_rng(seed)
    A = synth._ACGT
    names, seqs = [], []
    for i in range(n_small):
        names.append("s%05d" % i); seqs.append(A[rng.integers(0, 4, size=200, dtype=np.uint8)])
    long_read = A[rng.integers(0, 4, size=30000, dtype=np.uint8)]
    names.insert(n_small // 2, "long"); seqs.insert(n_small // 2, long_read)
    q = long_read.copy()
    for p in rng.choice(q.shape[0], size=30, replace=False):
        q[p] = A[(int(np.where(A == q[p])[0][0]) + 1) % 4]
    T = synth.ReadSet(names, seqs, [np.full(s.shape[0], 40 + 33, dtype=np.uint8) for s in seqs])
    Q = synth.ReadSet(["qcopy"], [q], [np.full(q.shape[0], 40 + 33, dtype=np.uint8)])
    tf, qf = str(tmp_path / "many_all.fq"), str(tmp_path / "many_sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    return tf, qf


def check_long_pair_among_many_targets(lib, tmp_path):
    tf, qf = _one_long_pair_among_many_targets(tmp_path)
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "15", "-w", "5", "-I", "4G", "-p", "160", "-t", "4", tf, qf]
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    assert int(want.split("\t")[2]) > 0
    rc, out, err = run_main(lib, argv)
    assert rc == 0, err
    assert out == want


def test_emulated_long_pair_among_many_targets(emu_lib, tmp_path):
    check_long_pair_among_many_targets(emu_lib, tmp_path)


@slow_emu
def test_emulated_parallel_sort_size_classes(emu_lib, datasets, monkeypatch):
    """the parallel sort with its size classes shrunk 128-fold: more partition passes than are issued without looking
    (the tail with its look at the counter), segments whose keys agree in the bits of a pass (stepped over, unless still named
    by records: those must have left the originals before any pass writes to B)"""
    tf, qf = datasets("small")
    argv = ONT + [tf, qf]
    want = oracle_bind.table(argv)
    for passes, key64 in (("2", "0"), ("2", "1")):   # (no blind passes at all -- the tail alone: the descending-order test below); the finishing kernels' 64-bit key shape
        monkeypatch.setenv("LQCOV_PS_SHIFT", "7"); monkeypatch.setenv("LQCOV_PS_PASSES", passes); monkeypatch.setenv("LQCOV_PS_KEY64", key64)
        rc, out, err = run_main(emu_lib, argv)
        assert rc == 0, err
        assert out == want, passes


@pytest.mark.parametrize("case", [pytest.param(c, marks=slow_emu) if c["name"] == "adv_parts" else c for c in _cases("table") if c["name"] in ("tiny_ont", "adv_parts")] or _cases("table")[:2], ids=lambda c: c["name"])
def test_emulated_threads_in_descending_order_and_the_sort_checked(emu_lib, case, monkeypatch):
    """Two of the emulator's and the engine's self-checks at once.
    LQ_EMU_ORDER=reverse: the emulator runs a block's threads lowest first by default and fills the kernels' LDS with a pattern
    before every block (tests/emu/hipemu.hpp).  With the highest thread first, "thread 0 initialises, the others read" without
    a barrier in between reads the pattern -- on the GPU: whatever another lane's kernel left in the CU's LDS (round 3's
    k_run_list fault) -- and an order that only the atomics of a partition pass gave shows (round 3: tied anchors of a one-x
    bucket).  LQCOV_DEBUG_SORT: after every batch's sort the engine checks on the host that the anchors are a permutation of the
    emitted ones (the finishing kernels rebuild x from the compact key: a bit of x outside the key would be lost), that every
    query ascends in x and that anchors with equal x both carry the tie mark (an unmarked tie could have met the parallel
    sort, whose order among equal keys is arbitrary); a violation is an error.
    The shrunken size classes send the sort through every kernel family on these small inputs."""
    monkeypatch.setenv("LQCOV_DEBUG_SORT", "1")
    for order in ("reverse",) + (("random:5",) if case["name"] == "tiny_ont" else ()):   # (random: whenever a thread waits, any ready one goes on)
        monkeypatch.setenv("LQ_EMU_ORDER", order)
        for env in ({}, {"LQCOV_PS_SHIFT": "7", "LQCOV_PS_PASSES": "0", "LQCOV_RUN_GRID": "7", "LQCOV_RUN_STAGE": "256"}, {"LQCOV_WALK_SHIFT": "6", "LQCOV_SORT": "klib"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            rc, out, err = run_main(emu_lib, case["argv"], cwd=GOLDEN)
            assert rc == 0, err
            assert out == read_gz(case["expect"]), (order, env)
            for k in env:
                monkeypatch.delenv(k)


# ---- klib's order only where it can be observed; seed hits that cannot reach a chain never written (map_batch) -------------------
def check_observable_ties_scheme(lib, tmp_path, monkeypatch, seed, env, small=False):
    """repeat-rich reads (equal-x anchors everywhere, rows depend on klib's order): first pass with the counting filter and any
    sort, second pass in klib's order for the runs the chain kernels listed; the table equals the reference's under every
    shape of the filter (table shrunk: rid slices, aliased counters; no filter) and of the second pass (tiny work space: several
    sub-batches), and with the scheme switched off (LQCOV_TIES=klib: rounds 1-3's path)"""
    tf, qf = _repeat_rich_dataset(tmp_path, seed, **(dict(n_targets=12, n_queries=2, glen=21000) if small else {}))   # (small: what the test emulator chews in seconds)
    argv = ONT + [tf, qf]
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rc, out, err = run_main(lib, argv)
    assert rc == 0, err
    assert out == want
    return err


# shapes of the seed filter (kernels_seed.hpp): tiny buckets (dozens of slices per query), one-minimizer-wide segments of 7
# (many pieces per bucket), a record buffer of 1024 (many chunks), counters so few that pairs and bins alias; no filter at all
OBS_ENVS = [{}, {"LQCOV_SEED_BUCKET": "64", "LQCOV_SEED_SEGL": "7"}, {"LQCOV_SEED_BUCKET": "300", "LQCOV_SEED_CHUNK": "1024", "LQCOV_SEED_PAIR_BITS": "3", "LQCOV_SEED_HWORDS": "40"}, {"LQCOV_FILTER": "0"},
            {"LQCOV_ANCHOR_BUDGET": "20000", "LQCOV_LANES": "2"}, {"LQCOV_TIES": "klib"}, {"LQCOV_DEBUG_SORT": "1", "LQCOV_PS_SHIFT": "5"},
            {"LQCOV_CHAIN_WAVE_MIN": "3"},             # (every run through the wave kernel: its own copy of the tie rule)
            {"LQCOV_PLAN_AHEAD": "0", "LQCOV_SEED_BUCKET": "3000", "LQCOV_SEED_DCAP": "500", "LQCOV_SEED_SEGL": "100"},   # (the seed plan made when the part is mapped, not with its index; buckets decided in passes over stretches of their targets)
            {"LQCOV_SEED_BUCKET": "3000", "LQCOV_SEED_DCAP": "300", "LQCOV_SEED_BIGCAP": "1500", "LQCOV_SEED_PAIR_BITS": "4"}]   # (... and buckets beyond that: by pairs only; few pair counters: more passes)
# the same switches in four runs for the test emulator (a minute each on repeat-rich reads); the GPU suite takes them one by one
OBS_ENVS_EMU = [{"LQCOV_SEED_BUCKET": "3000", "LQCOV_SEED_DCAP": "300", "LQCOV_SEED_BIGCAP": "1500"}, {"LQCOV_SEED_BUCKET": "64", "LQCOV_SEED_SEGL": "7", "LQCOV_SEED_CHUNK": "1024", "LQCOV_SEED_PAIR_BITS": "3", "LQCOV_SEED_HWORDS": "40", "LQCOV_CHAIN_WAVE_MIN": "3"},
                {"LQCOV_FILTER": "0", "LQCOV_PLAN_AHEAD": "0", "LQCOV_ANCHOR_BUDGET": "20000", "LQCOV_DEBUG_SORT": "1", "LQCOV_PS_SHIFT": "5"}, {"LQCOV_TIES": "klib"},
                {"LQCOV_SEED_SURV_MAX": "1000", "LQCOV_SEED_CHUNK": "4000"}]      # (the plan holds a chunk or two at a time: the part's queries are mapped in groups)


@pytest.mark.parametrize("env", [pytest.param(e, marks=slow_emu) if "LQCOV_TIES" in e or "LQCOV_SEED_BIGCAP" in e else e for e in OBS_ENVS_EMU], ids=lambda e: "+".join("%s=%s" % kv for kv in e.items()) or "default")
def test_emulated_observable_ties_scheme(emu_lib, tmp_path, monkeypatch, env):
    err = check_observable_ties_scheme(emu_lib, tmp_path, monkeypatch, 2 + len(env), env, small=True)
    if "LQCOV_TIES" not in env:
        assert "queries chained in klib's order" in err and " 0 runs of 0 queries" not in err      # the second pass ran


def check_filter_drops_chance_hits(lib, tmp_path):
    """many targets, few of them overlapping a query: most seed hits are lone chance hits and are never written, few queries need
    klib's order at all; rows equal the reference's"""
    import dataclasses
    from longqc_amd import synth
    cfg = dataclasses.replace(synth.CONFIGS["cfg3"], n_reads=1500, nsample=40, depth=3.0, mean_len=3000)
    T, Q = synth.make_dataset(cfg)
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "80", tf, qf]
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    p, _, _ = api.parse_args(argv)
    eng = api.Engine(p, 0, lib=lib)
    out = str(tmp_path / "o.tsv")
    eng.run_files(tf, qf, out=out, err=str(tmp_path / "e.log"))
    st, emitted = eng.map_stats(), eng.last_n_anchors
    eng.close()
    assert open(out).read() == want
    assert 0 < st["last_written"] < 0.7 * emitted, (st, emitted)
    assert st["klib_queries"] < 20, st


def test_emulated_filter_drops_chance_hits(emu_lib, tmp_path):
    check_filter_drops_chance_hits(emu_lib, tmp_path)


def check_filter_thresholds(lib, tmp_path, monkeypatch, capfd):
    """the filter's threshold n_min = max(-n, ceil(-m / k)) from 2 to 15 (pair counters and 4-bit bins saturate at 15; the seven-bin
    window is exact up to 4 and lets a stretch that reaches its edge pass beyond), 16 and 1 (no filter possible): the reference's rows
    each time, with the default geometry and with few pair counters / histogram words (pairs that alias, pairs left without bins)"""
    import dataclasses
    from longqc_amd import synth
    cfg = dataclasses.replace(synth.CONFIGS["cfg3"], n_reads=500, nsample=20, depth=8.0, mean_len=2500)
    T, Q = synth.make_dataset(cfg)
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    seen = set()
    big = {"LQCOV_SEED_BUCKET": "3000", "LQCOV_SEED_DCAP": "300", "LQCOV_SEED_BIGCAP": "1500", "LQCOV_SEED_PAIR_BITS": "4"}   # buckets decided in passes, the largest by pairs only
    for n, m, filtered in ((2, 20, True), (3, 40, True), (5, 40, True), (3, 100, True), (15, 40, True), (16, 40, False), (1, 10, False)):
        argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-n", str(n), "-m", str(m), tf, qf]
        want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
        seen.add(want)
        for env in ({}, {"LQCOV_SEED_BUCKET": "300", "LQCOV_SEED_PAIR_BITS": "3", "LQCOV_SEED_HWORDS": "40", "LQCOV_SEED_SEGL": "50"}, big):
            if env is big:
                monkeypatch.setenv("LQCOV_SEED_STATS", "1")
                capfd.readouterr()
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            p, _, _ = api.parse_args(argv)
            eng = api.Engine(p, 0, lib=lib)
            out = str(tmp_path / "o.tsv")
            eng.run_files(tf, qf, out=out, err=str(tmp_path / "e.log"))
            st, emitted = eng.map_stats(), eng.last_n_anchors
            eng.close()
            for k in env:
                monkeypatch.delenv(k)
            assert open(out).read() == want, (n, m, env)
            if env is big:
                monkeypatch.delenv("LQCOV_SEED_STATS")
                if filtered:                        # "... N buckets beyond the block (M of them by pairs only) ..."
                    log = capfd.readouterr().err   # (the counters go to the process's stderr, not to the call's log)
                    mt = re.search(r"(\d+) buckets beyond the block \((\d+) of them by pairs only\)", log)
                    assert mt and int(mt.group(1)) > int(mt.group(2)) > 0, log[-600:]
            if not env:
                assert (st["last_written"] < emitted) == filtered, (n, m, st, emitted)
    assert len(seen) >= 4                           # (the thresholds do change the rows: the comparison is not vacuous)


def test_emulated_filter_thresholds(emu_lib, tmp_path, monkeypatch, capfd):
    check_filter_thresholds(emu_lib, tmp_path, monkeypatch, capfd)


def _pileup_dataset(tmp_path, n_targets, seed=7, n_hot=2, qlen=900):
    """`n_hot` queries buried under `n_targets` pieces of themselves (half of the pieces of the first query start at one and
    the same base, so that the counter of one minimizer climbs fastest), plus one query nothing matches"""
    rng = np.random.default_rng(seed)
    B = np.array(list("ACGT"))
    qs = [rng.integers(0, 4, qlen) for _ in range(n_hot + 1)]
    tf, qf = str(tmp_path / "pile_t.fa"), str(tmp_path / "pile_q.fa")
    with open(tf, "w") as f:
        for i in range(n_targets):
            q = qs[i % n_hot]
            if i % n_hot == 0 and rng.random() < 0.5:
                s, L = qlen // 3, int(rng.integers(150, 600))
            else:
                L = int(rng.integers(150, 700)); s = int(rng.integers(0, qlen - L + 1))
            seg = q[s:s + L].copy()
            for _ in range(int(rng.integers(0, 4))):
                at = int(rng.integers(20, L)); seg[at] = (seg[at] + 1 + rng.integers(0, 3)) % 4
            if rng.random() < 0.5:
                seg = (3 - seg)[::-1]
            f.write(">t%d\n%s\n" % (i, "".join(B[seg])))
    with open(qf, "w") as f:
        for i, q in enumerate(qs):
            f.write(">q%d\n%s\n" % (i, "".join(B[q])))
    return tf, qf


PILEUP16 = dict(n_targets=130000, f_first=0.85, seed=11, qlen=900)
PILEUP16_ARGV = ["-Y", "-l", "0", "-q", "40", "-k", "12", "-w", "5", "-I", "4G", "-p", "40", "-m", "20", "-t", "4"]


def _pileup16_dataset(tmp_path, n_targets, f_first, seed, qlen):
    """A pile-up that fills a real uint16 counter: 85 % of 130 000 targets are prefixes of one 900-base query, so the counter of
    the query's first minimizer reaches 65 535 about half way through its chains and the chains that come later (in mm_gen_regs'
    order) no longer count (esterr.c:136) -- which ones those are decides the row.  tests/golden/pileup16_rows.json holds what
    the reference binary printed for it (tests/golden/make_pileup_golden.py)."""
    rng = np.random.default_rng(seed)
    B = np.array(list("ACGT"))
    q = rng.integers(0, 4, qlen)
    tf, qf = str(tmp_path / "pile16_t.fa"), str(tmp_path / "pile16_q.fa")
    with open(tf, "w") as f:
        for i in range(n_targets):
            if rng.random() < f_first:
                s, L = 0, int(rng.integers(150, 900))
            else:
                L = int(rng.integers(150, 700)); s = int(rng.integers(0, qlen - L + 1))
            seg = q[s:s + L].copy()
            for _ in range(int(rng.integers(0, 3))):
                at = int(rng.integers(20, L)); seg[at] = (seg[at] + 1 + rng.integers(0, 3)) % 4
            if rng.random() < 0.5:
                seg = (3 - seg)[::-1]
            f.write(">t%d\n%s\n" % (i, "".join(B[seg])))
    with open(qf, "w") as f:
        f.write(">q0\n" + "".join(B[q]) + "\n")
        f.write(">q1\n" + "".join(B[rng.integers(0, 4, 700)]) + "\n")
    return tf, qf


def check_saturated_counters_are_replayed(lib, tmp_path, monkeypatch, bits, n_targets, parts, env=""):
    """esterr.c:130,136: the match counters are uint16 and their saturation test reads the counter of the chain's first minimizer,
    so once a counter is full the others depend on the order in which lq_cnt_match met the chains (hit.c:52-88).  With the
    counters narrowed to `bits` bits on both sides (a test hook) a few hundred overlaps get there: the rows must equal the
    oracle's, which walks the chains serially in the reference's order -- and not the rows of the opposite order."""
    tf, qf = _pileup_dataset(tmp_path, n_targets)
    argv = ["-Y", "-l", "0", "-q", "40", "-k", "12", "-w", "5", "-I", parts, "-p", "40", "-m", "20", "-t", "4", tf, qf]
    monkeypatch.setenv("LQO_CNT_BITS", str(bits))
    want = oracle_bind.table(argv)
    monkeypatch.setenv("LQO_REGS_ASCENDING", "1")
    assert oracle_bind.table(argv) != want                       # the input can tell the order
    monkeypatch.delenv("LQO_REGS_ASCENDING")
    monkeypatch.setenv("LQO_CNT_BITS", "16")
    assert oracle_bind.table(argv) != want                       # ... and the width
    monkeypatch.setenv("LQCOV_TEST_CNT_BITS", str(bits))
    for kv in env.split("+"):
        if kv:
            monkeypatch.setenv(*kv.split("=", 1))
    rc, out, err = run_main(lib, argv)
    assert rc == 0, err[-2000:]
    assert out == want
    assert "chains replayed in the reference's order" in err


@pytest.mark.parametrize("bits,n_targets,parts,env", [(5, 300, "4G", ""), (5, 400, "40K", "LQCOV_TIES=klib")], ids=["one_part", "parts_all_klib"])
def test_emulated_saturated_counters_are_replayed(emu_lib, tmp_path, monkeypatch, bits, n_targets, parts, env):
    check_saturated_counters_are_replayed(emu_lib, tmp_path, monkeypatch, bits, n_targets, parts, env)


def test_fuzz_tool_runs_a_case(emu_lib):
    """tools/fuzz_emu.py (random read sets, argv and knobs; the emulator build's table against the reference binary's): one seed of
    the campaign recorded in profiles/README.md still comes out identical"""
    import subprocess
    import sys
    if not oracle_bind.have_ref():
        pytest.skip("needs the reference binary (oracle/_ref)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_emu.py"), "--one", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "identical (20 rows" in r.stdout, r.stdout[-2000:]
