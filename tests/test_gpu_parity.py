"""-m gpu: the real thing.  Everything here goes through liblqcov.so (hipcc, gfx950) on cuda:0 via the
C ABI and is compared bit for bit with (a) the reference's golden vectors, (b) the oracle on seeded
inputs, (c) size-independent properties at larger sizes."""
import json
import os

import numpy as np
import pytest

import tests.test_emu_pipeline as E
import tests.test_host as H
from longqc_amd import api, synth
from tests import oracle_bind
from tests.conftest import GOLDEN, ROOT, read_gz
from tests.helpers import ONT, read_fastx, run_main

pytestmark = pytest.mark.gpu


def _cases(kind):
    return [c for c in json.load(open(os.path.join(GOLDEN, "cases.json"))) if c["kind"] == kind]


@pytest.mark.parametrize("case", _cases("table"), ids=lambda c: c["name"])
def test_gpu_tables_equal_reference_fixtures(gpu_lib, case):
    E.test_emulated_pipeline_reproduces_reference_tables(gpu_lib, case)


def test_gpu_tables_read_like_the_reference_consumer(gpu_lib, tmp_path):
    E.test_emulated_tables_read_like_the_reference_consumer(gpu_lib, tmp_path)


@pytest.mark.parametrize("name,k,w,hpc,fn", [
    ("tiny_sketch_k12w5", 12, 5, 0, "tiny_sub.fq.gz"),
    ("adv_sketch_k12w5", 12, 5, 0, "adv_sub.fq.gz"),
    ("adv_sketch_k15w10hpc", 15, 10, 1, "adv_sub.fq.gz"),
    ("adv_sketch_k19w10", 19, 10, 0, "adv_sub.fq.gz"),
    ("adv_sketch_k6w30", 6, 30, 0, "adv_sub.fq.gz"),
])
def test_gpu_sketch_equals_mm_sketch_fixture(gpu_lib, name, k, w, hpc, fn):
    E.test_emulated_sketch_equals_mm_sketch_fixture(gpu_lib, name, k, w, hpc, fn)


def test_gpu_sketch_halo_adversarial(gpu_lib):
    E.test_emulated_sketch_halo_adversarial(gpu_lib)


@pytest.mark.parametrize("name,tfn,qfn", [("tiny_chains", "tiny_all.fq.gz", "tiny_sub.fq.gz"), ("adv_chains", "adv_all.fa.gz", "adv_sub.fq.gz")])
def test_gpu_chains_mid_occ_and_accumulators(gpu_lib, name, tfn, qfn):
    E.test_emulated_chains_mid_occ_and_accumulators(gpu_lib, name, tfn, qfn)


def test_gpu_reset_and_rerun(gpu_lib):
    E.check_reset_and_rerun(gpu_lib, False)
    E.check_reset_and_rerun(gpu_lib, True)


def test_gpu_constant_k_sketch_fuzz(gpu_lib, monkeypatch):
    """k_sketch_dp_fast (round 6) against the state machine and the general data-parallel kernel, on the device"""
    E.test_emulated_constant_k_sketch_fuzz(gpu_lib, monkeypatch)


def test_gpu_data_parallel_sketch_fuzz(gpu_lib, monkeypatch):
    E.test_emulated_data_parallel_sketch_fuzz(gpu_lib, monkeypatch)


@pytest.mark.parametrize("slices", ["1", "3"])
def test_gpu_packed_upload_without_ambiguity_words(gpu_lib, monkeypatch, slices):
    """lqcov_part_add_packed with amb == NULL for the read ranges without an N (round 6), on the device"""
    E.test_emulated_packed_upload_without_ambiguity_words(gpu_lib, monkeypatch, slices)


def test_gpu_input_dialects(gpu_lib, tmp_path):
    H.test_input_dialects_agree_with_reference_semantics(gpu_lib, tmp_path)


def test_gpu_empty_inputs(gpu_lib, tmp_path):
    H.test_empty_query_file_and_empty_target_file(gpu_lib, tmp_path)


VARIANTS = {
    "ont": ONT,
    "pb": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "80", "-t", "4"],
    "fast_k15": ["-Y", "-l", "0", "-q", "160", "-k", "15", "-w", "5", "-I", "4G", "-p", "160", "-t", "4"],
    "hifi_k19w10": ["-Y", "-l", "0", "-q", "160", "-k", "19", "-w", "10", "-I", "4G", "-p", "160", "-t", "4"],
    "parts": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "1M", "-p", "160", "-t", "4"],
    "spike_hpc": ["-Y", "-Hk15", "-w", "10", "-c", "1", "-l", "0", "--filter", "-t", "4"],
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("name", ["small", "cfg1"])
def test_gpu_table_equals_oracle_on_synthetic_sets(gpu_lib, datasets, name, variant):
    """BASELINE.json configs[0] (cfg1: 1k ONT reads ~10 kb 5x, all reads as queries) and a smaller set"""
    tf, qf = datasets(name)
    argv = VARIANTS[variant] + [tf, qf]
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    assert out == oracle_bind.table(argv)


def test_gpu_query_batching_is_invisible(gpu_lib, datasets, monkeypatch):
    E.test_emulated_query_batching_is_invisible(gpu_lib, datasets, monkeypatch, "2")


def test_gpu_midsize_slice_of_cfg2_vs_reference_or_oracle(gpu_lib, tmp_path):
    """a 6 % slice of BASELINE.json configs[1] (3000 of the 50k ONT reads ~15 kb, 600 queries): big enough that
    per-query anchor arrays reach 10^4..10^5 and the klib walk recurses several levels"""
    import dataclasses
    cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=3000, nsample=600, depth=15.0)
    T, Q = synth.make_dataset(cfg)
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    argv = ONT + [tf, qf]
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    assert out == want


def test_gpu_properties_at_scale(gpu_lib, tmp_path):
    """Size-independent properties at the size of BASELINE configs[1] (50k ONT reads ~15 kb, 744 Mbases; LQCOV_QUICK_SCALE=1
    runs 20k): determinism, query independence (a query's row does not depend on which other queries ride
    along or on the anchor batching), row sanity, and spot rows against the reference binary."""
    import dataclasses
    n = 20000 if os.environ.get("LQCOV_QUICK_SCALE") else 50000        # all of configs[1] unless asked otherwise
    cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=n, nsample=2000)
    genome = synth.make_genome(cfg)
    T = synth.make_reads(cfg, genome)
    qi = synth.reservoir_subsample(len(T), cfg.nsample)
    Q = T.subset(qi)
    p = api.default_params(no_self=1, min_ovlp=0, min_score_med=160, min_score_good=160)
    eng = api.Engine(p, 0, lib=gpu_lib)
    eng.set_queries(Q.names, Q.seqs, Q.quals)
    pt = eng.part_begin()
    step = 2500
    for i in range(0, len(T), step):
        eng.part_add_targets(pt, T.names[i:i + step], T.seqs[i:i + step])
    eng.part_build(pt); eng.part_map(pt); eng.finish()
    t1 = eng.table_text()
    eng.reset(); eng.part_build(pt); eng.part_map(pt); eng.finish()
    assert eng.table_text() == t1                                     # deterministic
    rows = t1.splitlines()
    assert len(rows) == len(Q)
    for r, nm, s in zip(rows, Q.names, Q.seqs):
        f = r.split("\t")
        assert f[0] == nm and int(f[1]) == s.shape[0]
        if f[3] != "0":
            for seg in f[3].split(","):
                a, b = seg.split("-"); assert 0 <= int(a) < int(b) <= s.shape[0]
    assert sum(1 for r in rows if r.split("\t")[2] != "0") > 0.9 * len(rows)
    # query independence: 40 of the queries alone, tiny anchor budget, same rows
    sel = list(range(0, len(Q), len(Q) // 40))[:40]
    os.environ["LQCOV_ANCHOR_BUDGET"] = "200000"
    try:
        eng2 = api.Engine(p, 0, lib=gpu_lib)
    finally:
        del os.environ["LQCOV_ANCHOR_BUDGET"]
    sub = Q.subset(sel)
    eng2.set_queries(sub.names, sub.seqs, sub.quals)
    pt2 = eng2.part_begin()
    for i in range(0, len(T), step):
        eng2.part_add_targets(pt2, T.names[i:i + step], T.seqs[i:i + step])
    eng2.part_build(pt2); eng2.part_map(pt2); eng2.finish()
    t2 = eng2.table_text().splitlines()
    assert t2 == [rows[i] for i in sel]
    eng2.close(); eng.close()
    # spot rows against the reference binary itself (it is deterministic across -t)
    if oracle_bind.have_ref():
        tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
        synth.write_fastq(tf, T); synth.write_fastq(qf, sub)
        want = oracle_bind.ref_table(["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-t", str(os.cpu_count() or 4), tf, qf]).splitlines()
        assert t2 == want


def test_gpu_slice_of_configs2_vs_reference(gpu_lib, tmp_path):
    """1 % of BASELINE configs[2] (5000 of the 500k PacBio CLR reads ~10 kb, pb-sequel preset -p 80, 100 queries) against
    the reference binary"""
    import dataclasses
    cfg = dataclasses.replace(synth.CONFIGS["cfg3"], n_reads=5000, nsample=100)
    T, Q = synth.make_dataset(cfg)
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "80", "-t", "8", tf, qf]
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    assert out == want


def test_gpu_full_configs2_rows_equal_the_reference_fixture(gpu_lib):
    """ALL of BASELINE configs[2] as targets (500k reads, 5.2 Gbases: a 4.0-Gbase and a 1.2-Gbase index part by the rule of
    index.c:244) and the 40 queries for which the reference binary's rows were recorded in the build container
    (tests/golden/make_scale_golden.py -> cfg3_rows.json): byte-identical rows.  LQCOV_SKIP_FULL_CFG3=1 skips it."""
    if os.environ.get("LQCOV_SKIP_FULL_CFG3"):
        pytest.skip("asked to skip the full-size run")
    from longqc_amd import multigpu
    g = json.load(open(os.path.join(GOLDEN, "cfg3_rows.json")))
    cfg = synth.CONFIGS["cfg3"]
    genome = synth.make_genome(cfg)
    F = synth.make_reads_flat(cfg, genome)
    Q = synth.make_reads(cfg, genome, indices=g["read_indices"])
    assert [int(s.shape[0]) for s in Q.seqs] == g["query_lengths"]
    p, _, _ = api.parse_args(g["argv"] + ["t", "q"])
    lens = np.diff(F.off).astype(np.int64)
    parts = multigpu.split_parts(lens, int(p.batch_size), int(p.idx_mini_batch))
    assert len(parts) == 2
    P = api.PackedReads(F.flat, F.off, F.names(), lib=gpu_lib)
    eng = api.Engine(p, 0, lib=gpu_lib)
    eng.set_queries(Q.names, Q.seqs, Q.quals)
    pt = eng.part_begin()
    for lo, hi in parts:
        eng.part_clear(pt)
        eng.part_add_packed(pt, P, lo, hi)
        eng.part_build(pt); eng.part_map(pt)
    eng.finish()
    rows = eng.table_text().splitlines()
    eng.close(); P.close()
    assert rows == g["rows"]


def test_gpu_ultra_long_reads_vs_reference(gpu_lib, tmp_path):
    """the shape of BASELINE configs[4]: reads of 100-600 kb (N50 ~100 kb) at 30x over a small genome -- every query meets
    every target, (query, strand, target) runs of 10^4+ anchors, sub-arrays of 10^6 -- against the reference binary"""
    import dataclasses
    cfg = dataclasses.replace(synth.CONFIGS["cfg5"], n_reads=200, nsample=200, depth=30.0)
    genome = synth.make_genome(cfg)
    T = synth.make_reads(cfg, genome)
    order = sorted(range(len(T)), key=lambda i: -int(T.seqs[i].shape[0]))[:24]
    Q = T.subset(order)
    assert max(int(s.shape[0]) for s in Q.seqs) > 200000
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    argv = ONT + [tf, qf]
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    assert out == want


def test_gpu_ultra_long_reads_2000_vs_reference(gpu_lib, tmp_path):
    """a larger slice of the shape of BASELINE configs[4]: 2 000 reads of N50 ~100 kb (120 Mbases, 30x) and the 200-read
    reservoir subsample as queries -- (query, target) runs of 10^4 anchors that the parallel sort has to order by position
    after stepping over 11 constant bits of rid -- against the reference binary"""
    import dataclasses
    cfg = dataclasses.replace(synth.CONFIGS["cfg5"], n_reads=2000, nsample=200, depth=30.0)
    T, Q = synth.make_dataset(cfg)
    assert len(Q) == 200 and max(int(s.shape[0]) for s in T.seqs) > 300000
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    argv = ONT[:-2] + ["-t", str(os.cpu_count() or 4), tf, qf]
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    assert out == want


def test_gpu_many_parts_at_real_read_lengths_vs_reference(gpu_lib, tmp_path):
    """the shape of BASELINE configs[3] (5 M ONT reads ~20 kb 40x, ont-rapid preset, 25 index parts of 4 Gbases): 0.25 % of the
    reads (12 500, 0.25 Gbases) with -I scaled likewise (20M), so that the reference's rule (index.c:244,311-316) cuts a dozen
    parts at real read lengths: the long part loop (minimap2-coverage.c:449-458), mid_occ frozen by the first part (map.c:50),
    COVT and avg_k carried across the parts (esterr.c:87-97); 200 queries, table against the reference binary"""
    import dataclasses
    cfg = dataclasses.replace(synth.CONFIGS["cfg4"], n_reads=12500, nsample=200)
    T, Q = synth.make_dataset(cfg)
    assert len(Q) == 200
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "20M", "-p", "160", "-t", str(os.cpu_count() or 4), tf, qf]
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    assert sum(1 for l in err.splitlines() if "mapped" in l) >= 8, err
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    assert out == want
    rows = out.splitlines()
    assert len(rows) == 200 and sum(1 for r in rows if r.split("\t")[2] != "0") > 150


def test_gpu_parts_built_while_the_previous_one_is_mapped(gpu_lib):
    """the pipeline of bench.py at the level of the C ABI: two part objects, a host thread uploads, sketches and indexes part
    i + 1 (the engine's build stream) while part i is mapped; the 10 parts of the COVT fixture, twice, the reference's table"""
    import threading
    from longqc_amd import multigpu
    tn, ts, _ = read_fastx(os.path.join(GOLDEN, "adv_all.fa.gz"))
    qn, qs, qq = read_fastx(os.path.join(GOLDEN, "adv_sub.fq.gz"))
    p = api.Params(); gpu_lib.lqcov_params_default(p)
    p.no_self = 1; p.min_ovlp = 0; p.min_score_med = 160; p.min_score_good = 160; p.batch_size = 100000
    eng = api.Engine(p, 0, lib=gpu_lib)
    eng.set_queries(qn, qs, qq)
    eng.reserve_hbm(1 << 30)
    parts = multigpu.split_parts([int(s.shape[0]) for s in ts], 100000)
    assert len(parts) == 10
    pts = [eng.part_begin(), eng.part_begin()]

    def build(i):
        lo, hi = parts[i]
        eng.part_clear(pts[i % 2]); eng.part_add_targets(pts[i % 2], tn[lo:hi], ts[lo:hi]); eng.part_build(pts[i % 2])

    for rep in range(2):
        eng.reset()
        build(0)
        for i in range(len(parts)):
            th = threading.Thread(target=build, args=(i + 1,)) if i + 1 < len(parts) else None
            if th:
                th.start()
            eng.part_map(pts[i % 2])
            if th:
                th.join()
        eng.finish()
        assert eng.table_text() == read_gz("adv_parts.table.gz"), rep
    eng.close()


def test_gpu_long_pair_among_many_targets(gpu_lib, tmp_path):
    E.check_long_pair_among_many_targets(gpu_lib, tmp_path)


@pytest.mark.parametrize("env", [{"LQCOV_CKPT3": "1"}, {"LQCOV_WALK": "solo"}, {"LQCOV_CKPT": "0"}, {"LQCOV_SORT": "klib"}, {"LQCOV_SKETCH_KPT": "1", "LQCOV_SKETCH": "machine"},
                                 {"LQCOV_LANES": "2", "LQCOV_ANCHOR_BUDGET": "400000"},
                                 {"LQCOV_DEBUG_SORT": "1"}, {"LQCOV_DEBUG_SORT": "1", "LQCOV_PS_SHIFT": "5", "LQCOV_RUN_GRID": "5", "LQCOV_RUN_STAGE": "256"},
                                 {"LQCOV_DEBUG_SORT": "1", "LQCOV_PS_KEY64": "1"}], ids=lambda e: "+".join("%s=%s" % kv for kv in e.items()))
def test_gpu_every_shipped_switch_on_a_midsize_slice(gpu_lib, tmp_path, monkeypatch, env):
    """every switch that selects other kernels than the defaults (checkpoints for the second-longest class, solo walkers only,
    no checkpoints, every query through klib's passes, the sketch state machine alone with one chunk per thread, two lanes over small batches) on 1 500 ONT
    reads ~15 kb with 300 queries: no kernel of liblqcov.so stays unexecuted by the suite.  LQCOV_DEBUG_SORT: the engine checks every
    batch's sort on the host (a permutation of the emitted anchors, ascending, ties marked) and fails if not; with the parallel
    sort's classes shrunk and a run list of five blocks with a small LDS stage besides."""
    import dataclasses
    cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=1500, nsample=300, depth=15.0)
    T, Q = synth.make_dataset(cfg)
    tf, qf = str(tmp_path / "all.fq"), str(tmp_path / "sub.fq")
    synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
    argv = ONT + [tf, qf]
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    assert out == want


@pytest.mark.parametrize("shift", ["3", "7"])
def test_gpu_parallel_sort_size_classes(gpu_lib, datasets, monkeypatch, shift):
    """the parallel sort (kernels_psort.hpp) with its size classes shrunk: several partition passes on small inputs"""
    tf, qf = datasets("cfg1")
    argv = ONT + [tf, qf]
    want = oracle_bind.table(argv)
    monkeypatch.setenv("LQCOV_PS_SHIFT", shift)
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    assert out == want


def test_gpu_run_list_variants(gpu_lib, tmp_path, monkeypatch):
    E.check_run_list_variants(gpu_lib, tmp_path, monkeypatch)


@pytest.mark.parametrize("seed", list(range(3, 11)))
def test_gpu_repeat_rich_randomised(gpu_lib, tmp_path, monkeypatch, seed):
    E.check_repeat_rich_randomised(gpu_lib, tmp_path, monkeypatch, seed)


@pytest.mark.parametrize("variant", ["ckpt", "ckpt_all_klib", "plain"])
def test_gpu_checkpointed_walks(gpu_lib, tmp_path, monkeypatch, variant):
    E.test_emulated_checkpointed_walks(gpu_lib, tmp_path, monkeypatch, variant)


@pytest.mark.parametrize("shift", ["4", "7", "12"])
def test_gpu_every_walk_size_class(gpu_lib, datasets, monkeypatch, shift):
    E.test_emulated_every_walk_size_class(gpu_lib, datasets, monkeypatch, shift)


def test_gpu_two_ranks_share_the_device_exchange_accumulators(gpu_lib, tmp_path):
    """the N > 1 driver with real device pointers: two ranks (gloo, both on cuda:0 -- a 1-GPU box cannot host two
    RCCL ranks) split the 10 parts of the COVT fixture and must reproduce the reference's sequential table"""
    import torch.multiprocessing as mp
    import tests.test_multigpu_cpu as M
    out = str(tmp_path / "t.tsv")
    mp.spawn(M._worker, args=(2, M._free_port(), 100000, out, True), nprocs=2, join=True)
    assert open(out).read() == read_gz("adv_parts.table.gz")


def test_gpu_two_ranks_replay_saturated_counters_across_parts(gpu_lib, tmp_path, monkeypatch):
    """index parts spread over two ranks (gloo, both on cuda:0), 5-bit match counters that fill up in the merged sums: the ranks
    exchange the flagged queries' chains and replay them in the reference's order (esterr.c:127-138)"""
    import tests.test_multigpu_cpu as M
    M.check_ranks_replay_saturated_counters(tmp_path, monkeypatch, 2, True)


@pytest.mark.parametrize("I,expect", [(100000, "adv_parts.table.gz"), (4000000000, "adv_ont.table.gz")])
def test_gpu_two_ranks_share_the_device_query_sharded_replicated_index(gpu_lib, tmp_path, I, expect):
    """the north-star split with real device pointers: two ranks (gloo, both on cuda:0) each sketch half of every part,
    all-gather the minimizers, build the same index and map half of the queries; rank 0 gathers the rows"""
    import torch.multiprocessing as mp
    import tests.test_multigpu_cpu as M
    out = str(tmp_path / "t.tsv")
    mp.spawn(M._worker_qshard, args=(2, M._free_port(), I, out, True), nprocs=2, join=True)
    assert open(out).read() == read_gz(expect)


def test_gpu_in_memory_sampleqc_path(gpu_lib):
    """longqc_amd.sampleqc.coverage_in_memory: subsample handed over in memory, input streamed in chunks, parts cut
    by the reference's rule -- same table as the file-based reference run"""
    import tests.test_sampleqc as S
    S.check_in_memory_equals_file_path(gpu_lib)


@pytest.mark.parametrize("case", [c for c in _cases("table") if c["name"] in ("tiny_ont", "adv_ont", "adv_parts", "tiny_spike")], ids=lambda c: c["name"])
def test_gpu_wave_chain_kernel_on_every_run(gpu_lib, case, monkeypatch):
    E.test_emulated_wave_chain_kernel_on_every_run(gpu_lib, case, monkeypatch)


def test_gpu_wave_chain_and_lane_walker_variants_on_cfg1(gpu_lib, datasets, monkeypatch):
    """cfg1 in 1-Mbase parts (the klib-order-sensitive case) with every run in the cooperative chain kernel and the
    lane walker instead of the solo walker"""
    tf, qf = datasets("cfg1")
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "1M", "-p", "160", tf, qf]
    want = oracle_bind.table(argv)
    monkeypatch.setenv("LQCOV_CHAIN_WAVE_MIN", "3")
    monkeypatch.setenv("LQCOV_WALK", "lane")
    rc, out, err = run_main(gpu_lib, argv)
    assert rc == 0, err
    assert out == want


@pytest.mark.parametrize("lanes", ["1", "3"])
def test_gpu_concurrent_mapping_lanes(gpu_lib, datasets, monkeypatch, lanes):
    """small query batches on 1 and 3 concurrent mapping lanes (threads + streams) against the oracle"""
    E.test_emulated_query_batching_is_invisible(gpu_lib, datasets, monkeypatch, lanes)


@pytest.mark.parametrize("case", [c for c in _cases("table") if c["name"] in ("tiny_ont", "adv_parts")], ids=lambda c: c["name"])
def test_gpu_chain_lds_budget_overflow(gpu_lib, case, monkeypatch):
    E.test_emulated_chain_lds_budget_overflow(gpu_lib, case, monkeypatch)


# ---- the reference's index files ----
import tests.test_mmi as MMI  # noqa: E402


@pytest.mark.parametrize("case", MMI.CASES, ids=lambda c: c["name"])
def test_gpu_maps_from_the_reference_index_file(gpu_lib, case, tmp_path):
    MMI.check_maps_from_reference_index(gpu_lib, case, tmp_path)


@pytest.mark.parametrize("case", MMI.CASES, ids=lambda c: c["name"])
def test_gpu_index_dump_equals_the_reference_dump_and_loads_back(gpu_lib, case, tmp_path):
    MMI.test_emulated_index_dump_equals_the_reference_dump_and_loads_back(gpu_lib, case, tmp_path)


def test_gpu_dump_while_mapping_and_part_level_calls(gpu_lib, tmp_path):
    MMI.test_emulated_dump_while_mapping_and_part_level_calls(gpu_lib, tmp_path)


def test_gpu_index_with_more_minimizers_than_counters_is_refused(gpu_lib, tmp_path):
    MMI.test_emulated_index_with_more_minimizers_than_counters_is_refused(gpu_lib, tmp_path)


# ---- the low-complexity table (sdust) ----
import tests.test_sdust as SD  # noqa: E402


@pytest.mark.parametrize("case", SD.CASES, ids=lambda c: c["name"])
def test_gpu_sdust_main_equals_the_reference_table(gpu_lib, case, tmp_path):
    SD.check_main_equals_fixture(gpu_lib, case, tmp_path)


def test_gpu_sdust_in_memory_rows_and_errors(gpu_lib, tmp_path):
    SD.check_in_memory_rows(gpu_lib, tmp_path)


def test_gpu_lqmask_counterpart(gpu_lib, tmp_path):
    SD.test_emulated_lqmask_counterpart(gpu_lib, tmp_path)


def test_gpu_sdust_edge_reads(gpu_lib, tmp_path):
    SD.test_emulated_sdust_edge_reads(gpu_lib, tmp_path)


def test_gpu_sdust_repeat_rich_reads(gpu_lib, tmp_path):
    SD.check_repeat_rich(gpu_lib, tmp_path)


def test_gpu_sdust_midsize_vs_oracle(gpu_lib, tmp_path):
    """3 000 synthetic ONT reads (44 Mbases) with planted low-complexity stretches: the GPU table vs the oracle's"""
    import dataclasses
    import subprocess
    cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=3000, nsample=10)
    T, _ = synth.make_dataset(cfg)
    rng = np.random.default_rng(5)
    for i in range(0, len(T), 7):                                  # poly-A, dinucleotide and triplet repeats, N runs
        s = T.seqs[i].copy()
        if s.shape[0] > 900:
            a = int(rng.integers(0, s.shape[0] - 800))
            s[a:a + 200] = ord("A"); s[a + 300:a + 500] = np.frombuffer(b"AT" * 100, dtype=np.uint8)
            s[a + 520:a + 526] = ord("N"); s[a + 600:a + 780] = np.frombuffer(b"CAG" * 60, dtype=np.uint8)
            T.seqs[i] = s
    fq = str(tmp_path / "t.fq")
    synth.write_fastq(fq, T)
    want = subprocess.run([oracle_bind.ensure_oracle(), "sdust", fq], stdout=subprocess.PIPE, check=True).stdout.decode()
    rc, out, err = SD.run_sdust_main(gpu_lib, [fq], tmp=tmp_path)
    assert rc == 0, err
    assert out == want
    assert sum(int(l.split("\t")[1]) for l in out.splitlines()) > 100000


@pytest.mark.parametrize("env", E.OBS_ENVS, ids=lambda e: "+".join("%s=%s" % kv for kv in e.items()) or "default")
@pytest.mark.parametrize("seed", [11, 12])
def test_gpu_observable_ties_scheme(gpu_lib, tmp_path, monkeypatch, seed, env):
    E.check_observable_ties_scheme(gpu_lib, tmp_path, monkeypatch, seed, env)


def test_gpu_filter_drops_chance_hits(gpu_lib, tmp_path):
    E.check_filter_drops_chance_hits(gpu_lib, tmp_path)


@pytest.mark.parametrize("bits,n_targets,parts,env", [(5, 300, "4G", ""), (5, 400, "40K", "LQCOV_TIES=klib"), (6, 1500, "40K", ""), (7, 3000, "4G", "LQCOV_CHAIN_WAVE_MIN=1000")],
                         ids=["one_part", "parts_all_klib", "parts_until_the_coverage_cap", "thread_kernel_only"])
def test_gpu_saturated_counters_are_replayed(gpu_lib, tmp_path, monkeypatch, bits, n_targets, parts, env):
    E.check_saturated_counters_are_replayed(gpu_lib, tmp_path, monkeypatch, bits, n_targets, parts, env)


def test_gpu_a_full_uint16_counter_equals_the_reference_fixture(gpu_lib, tmp_path):
    """the real thing: 110 000 good overlaps on the first minimizer of one query inside one index part.  The counter stops at
    65 535, the chains that mm_gen_regs' order puts after that point no longer count (esterr.c:136); the row is what the reference
    binary printed (tests/golden/pileup16_rows.json) -- the oracle with the chains in the opposite order prints another one"""
    fx = json.load(open(os.path.join(GOLDEN, "pileup16_rows.json")))
    tf, qf = E._pileup16_dataset(tmp_path, **fx["dataset"])
    p, _, _ = api.parse_args(fx["argv"] + [tf, qf])
    eng = api.Engine(p, 0, lib=gpu_lib)
    out = str(tmp_path / "o.tsv")
    eng.run_files(tf, qf, out=out, err=str(tmp_path / "e.log"))
    rows = eng.rows()
    eng.close()
    assert open(out).read() == fx["table"]
    assert rows[0]["flags"] & 5 == 5 and rows[1]["flags"] == 0      # LQCOV_ROW_SATURATED | LQCOV_ROW_REPLAYED
    assert "chains replayed in the reference's order" in open(str(tmp_path / "e.log")).read()


def test_gpu_run_files_pipeline_on_a_plain_file_in_many_parts(gpu_lib, tmp_path, monkeypatch):
    """lqcov_run_files' own pipeline on the GPU: a plain FASTQ parsed from the mapping in tiny pieces, four index parts, part k + 1
    uploaded, sketched and indexed by a second host thread while part k is mapped; against the oracle on the gzip of the same reads"""
    H.test_run_files_parses_plain_targets_from_the_mapping(gpu_lib, tmp_path, monkeypatch)


@pytest.mark.parametrize("name", ["cfg4s", "cfg5s"])
def test_gpu_real_size_parts_rows_equal_the_reference_fixture(gpu_lib, name):
    """index parts of REAL size (-I 4G): cfg4s = 600 000 ONT reads ~20 kb of configs[3] (12 Gbases: three 4-Gbase parts and a rest;
    mid_occ frozen from part 0, map.c:50; COVT across real parts, esterr.c:87), cfg5s = 75 000 ultra-long reads of configs[4]
    (4.5 Gbases, 5000 queries ~ 300 Mbases: near the 500-Mbase limit of one query mini-batch, bseq.c:86-98).  40 rows each as the
    reference binary printed them in the build container (tests/golden/make_scale_golden.py)."""
    fx = os.path.join(GOLDEN, name + "_rows.json")
    if not os.path.exists(fx):
        pytest.skip("fixture not made")
    # In this process, like every other test (rounds 4-5 ran these two in a process of their own: late in a long pytest process the HIP
    # runtime had been seen to fault inside its allocator -- ROCm 7.2's stream-ordered pool handing out blocks still in use, explained
    # and removed from the lanes' path in round 5; the switch that could bring it back is gone since round 6).
    from tests import real_size_runner
    real_size_runner.main(name, lib=gpu_lib)


# ---- the process-level boundary (lq_exec.py:13-38,70-71; longQC.py:438-446,520-526): both back ends of LqCovExec and the two
# argv-compatible executables as the subprocesses longQC.py would spawn ---------------------------------------------------------
TINY_ARGV = ONT + [os.path.join(GOLDEN, "tiny_all.fq.gz"), os.path.join(GOLDEN, "tiny_sub.fq.gz")]


@pytest.mark.gpu
@pytest.mark.parametrize("subprocess_mode", [False, True], ids=["in_process", "subprocess"])
def test_gpu_lqcovexec_runs_like_lqexec(gpu_lib, tmp_path, subprocess_mode):
    """exec(*argv, out=, err=) returns at once, get_poll() is None while the run lasts and the exit status afterwards; the file
    named by out= holds the reference's table"""
    import time
    from longqc_amd import LqCovExec
    le = LqCovExec(subprocess_mode=subprocess_mode)
    out, err = str(tmp_path / "coverage_out.txt"), str(tmp_path / "coverage_err.txt")
    le.exec(*TINY_ARGV, out=out, err=err)
    t0 = time.time()
    while le.get_poll() is None:                               # (longQC.py:520-526 polls every 10 s)
        assert time.time() - t0 < 300
        time.sleep(0.05)
    assert le.get_poll() == 0, open(err).read()[-2000:]
    assert open(out).read() == read_gz("tiny_ont.table.gz")
    assert le.get_bin_path().endswith("minimap2-coverage-mi355x") and le.get_pid().isdigit()
    le.close()


@pytest.mark.gpu
def test_gpu_executables_as_subprocesses(gpu_lib, tmp_path):
    """minimap2-coverage-mi355x: table on stdout, log on stderr, LQCOV_DEVICE names the device, exit codes 0 / 1 (bad flags,
    unopenable target: minimap2-coverage.c:229-234,276-279) / 3 (no such device); sdust-mi355x: the reference's sdust table"""
    import subprocess
    exe = os.path.join(ROOT, "longqc_amd", "minimap2-coverage-mi355x")
    env = dict(os.environ, LQCOV_DEVICE="0")
    r = subprocess.run([exe] + TINY_ARGV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode() == read_gz("tiny_ont.table.gz")
    assert b"Real time" in r.stderr or b"[lqcov]" in r.stderr or len(r.stderr) > 0      # (free-form, never parsed: lq_exec.py:30-38)
    r = subprocess.run([exe, "-k", "12", "a", "b"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 1 and r.stdout == b""
    r = subprocess.run([exe] + ONT + [str(tmp_path / "missing.fq"), TINY_ARGV[-1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 1 and b"failed to open file" in r.stderr
    r = subprocess.run([exe] + TINY_ARGV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, LQCOV_DEVICE="63"), timeout=300)
    assert r.returncode == 3 and r.stdout == b"" and b"device" in r.stderr.lower()
    sd = os.path.join(ROOT, "longqc_amd", "sdust-mi355x")
    fq = str(tmp_path / "tiny_all.fq")
    with open(fq, "wb") as f:
        import gzip
        f.write(gzip.open(TINY_ARGV[-2]).read())
    r = subprocess.run([sd, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rows = r.stdout.decode().splitlines()
    names, _, _ = read_fastx(TINY_ARGV[-2])
    assert [l.split("\t")[0] for l in rows] == list(names)
    assert r.stdout.decode() == read_gz("tiny_all.sdust.gz")       # (what the reference's sdust printed: tests/golden/make_sdust_golden.py)
