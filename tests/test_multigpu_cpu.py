"""The N > 1 path on CPU: world_size-2 `gloo` process group, each rank driving the emulator build of the
engine (tests/emu) -- checks the part planning, the COVT / avg_k replay and the RCCL-shaped exchange of
longqc_amd/multigpu.py against the reference's own multi-part semantics (golden table + oracle)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from longqc_amd import api, multigpu
from tests import oracle_bind
from tests.conftest import GOLDEN, ROOT, read_gz
from tests.helpers import read_fastx, slow_emu


def test_split_parts_matches_reference_rule():
    rng = np.random.default_rng(1)
    lens = rng.integers(500, 3000, size=700).tolist()
    # oracle's `index` dump prints n_seq per part for the same rule
    import tempfile
    from longqc_amd import synth
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    rs = synth.ReadSet(["r%d" % i for i in range(len(lens))], [A[rng.integers(0, 4, l)] for l in lens], [None] * len(lens))
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "t.fa")
        synth.write_fastq(fn, rs, fasta=True)
        for I in ["100K", "333K", "2M"]:
            dump = oracle_bind.dump("index", ["-k", "12", "-w", "5", "-I", I], [fn])
            want = [int(l.split("\t")[2]) for l in dump.splitlines()]
            b = int(float(I[:-1]) * (1e3 if I[-1] == "K" else 1e6) + .499)
            got = [e - s for s, e in multigpu.split_parts(lens, b)]
            assert got == want, I


def test_covt_replay_logic():
    lam = torch.tensor([[100, 2000, 0], [100, 2000, 50], [100, 2000, 70]], dtype=torch.int64)
    avgk = torch.tensor([[12., 12., 0.], [12., 12., 12.], [12., 12., 12.]])
    inc, tot, k = multigpu.covt_replay(lam, avgk, torch.tensor([10, 10, 10]))
    assert inc.tolist() == [[True, True, False], [True, False, True], [True, False, True]]   # 2000/10 > 150 caps query 1 after part 0
    assert tot.tolist() == [300, 2000, 120] and k.tolist() == [12., 12., 12.]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, argv_I, out_path, use_gpu=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = api.load_library() if use_gpu else api.load_library(os.path.join(ROOT, "tests", "emu", "liblqcov_emu.so"))
        tn, ts, _ = read_fastx(os.path.join(GOLDEN, "adv_all.fa.gz"))
        qn, qs, qq = read_fastx(os.path.join(GOLDEN, "adv_sub.fq.gz"))
        p = api.Params(); lib.lqcov_params_default(p)
        p.no_self = 1; p.min_ovlp = 0; p.min_score_med = 160; p.min_score_good = 160; p.batch_size = argv_I
        eng = api.Engine(p, 0, lib=lib)
        eng.set_queries(qn, qs, qq)
        parts = multigpu.split_parts([int(s.shape[0]) for s in ts], argv_I)
        runner = multigpu.PartRunner(eng, world, rank, torch.device("cuda", 0) if use_gpu else torch.device("cpu"), [int(s.shape[0]) for s in qs])
        runner.begin()
        for base in range(0, len(parts), world):
            mine = base + rank
            pid = None
            if mine < len(parts):
                s, e = parts[mine]
                pid = eng.part_begin()
                eng.part_add_targets(pid, tn[s:e], ts[s:e])
                eng.part_build(pid)
            runner.map_and_combine(pid, part_index=mine, mid_occ_owner=0, share_mid_occ=(base == 0))
            if pid is not None:
                eng.part_release(pid)
        eng.finish()
        if rank == 0:
            eng.write_table(out_path)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, pytest.param(3, marks=slow_emu)])
def test_two_ranks_gloo_equal_reference_multipart_table(emu_lib, tmp_path, world):
    """adv_parts fixture: 10 parts at -I 100K, the pile-up queries hit the COVT cap after part 1 -- the
    distributed run must reproduce the reference's sequential multi-part table byte for byte."""
    out = str(tmp_path / "t.tsv")
    mp.spawn(_worker, args=(world, _free_port(), 100000, out), nprocs=world, join=True)
    assert open(out).read() == read_gz("adv_parts.table.gz")


def _worker_saturated(rank, world, port, tf, qf, argv, out_path, bits, use_gpu=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["LQCOV_TEST_CNT_BITS"] = str(bits)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        lib = api.load_library() if use_gpu else api.load_library(os.path.join(ROOT, "tests", "emu", "liblqcov_emu.so"))
        full = [b"minimap2-coverage"] + [a.encode() for a in argv] + [tf.encode(), qf.encode()]
        arr = (C.c_char_p * len(full))(*full)
        p = api.Params()
        t, q, d = C.c_char_p(), C.c_char_p(), C.c_char_p()
        err = C.create_string_buffer(256)
        assert lib.lqcov_parse_args(len(full), arr, C.byref(p), C.byref(t), C.byref(q), C.byref(d), err, 256) == 0, err.value
        tn, ts, _ = read_fastx(tf)
        qn, qs, qq = read_fastx(qf)
        eng = api.Engine(p, 0, lib=lib)
        eng.set_queries(qn, qs, qq)
        parts = multigpu.split_parts([int(x.shape[0]) for x in ts], int(p.batch_size))
        assert len(parts) >= 2 * world                       # several rounds: a query flagged in one round is replayed in the next ones too
        runner = multigpu.PartRunner(eng, world, rank, torch.device("cuda", 0) if use_gpu else torch.device("cpu"), [int(x.shape[0]) for x in qs])
        runner.begin()
        for base in range(0, len(parts), world):
            mine = base + rank
            pid = None
            if mine < len(parts):
                s, e = parts[mine]
                pid = eng.part_begin()
                eng.part_add_targets(pid, tn[s:e], ts[s:e])
                eng.part_build(pid)
            runner.map_and_combine(pid, part_index=mine, mid_occ_owner=0, share_mid_occ=(base == 0))
            if pid is not None:
                eng.part_release(pid)
        eng.finish()
        if rank == 0:
            eng.write_table(out_path)
            flags = [r["flags"] for r in eng.rows()]
            open(out_path + ".flags", "w").write(" ".join(str(f) for f in flags))
            open(out_path + ".sat", "w").write(" ".join(str(k) for k in sorted(runner.sat)))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, pytest.param(3, marks=slow_emu)])
def test_ranks_replay_saturated_counters_across_parts(emu_lib, tmp_path, monkeypatch, world):
    check_ranks_replay_saturated_counters(tmp_path, monkeypatch, world, False)


def check_ranks_replay_saturated_counters(tmp_path, monkeypatch, world, use_gpu):
    """esterr.c:127-138 with the index parts spread over ranks: match counters narrowed to 5 bits on both sides (the test hook of
    tests/test_emu_pipeline.py's one-handle case) fill up in the MERGED sums; the ranks exchange the chains of the flagged queries
    (lqcov_part_sat_records -> broadcast -> lqcov_sat_replay) and the table equals the oracle's, which walks the parts and their
    chains serially in the reference's order.  Rounds 1-5 refused this case (rows flagged SATURATED without REPLAYED)."""
    from tests.test_emu_pipeline import _pileup_dataset
    tf, qf = _pileup_dataset(tmp_path, 400)
    argv = ["-Y", "-l", "0", "-q", "40", "-k", "12", "-w", "5", "-I", "25K", "-p", "40", "-m", "20", "-t", "4"]
    monkeypatch.setenv("LQO_CNT_BITS", "5")
    want = oracle_bind.table(argv + [tf, qf])
    monkeypatch.setenv("LQO_CNT_BITS", "16")
    assert oracle_bind.table(argv + [tf, qf]) != want           # the width shows in the table
    out = str(tmp_path / "sat.tsv")
    mp.spawn(_worker_saturated, args=(world, _free_port(), tf, qf, argv, out, 5, use_gpu), nprocs=world, join=True)
    assert open(out).read() == want
    flags = [int(f) for f in open(out + ".flags").read().split()]
    assert any(f & 1 for f in flags) and all((f & 4) for f in flags if f & 1)     # LQCOV_ROW_SATURATED rows all carry LQCOV_ROW_REPLAYED
    assert open(out + ".sat").read().split()


# ---- queries sharded, index replicated (BASELINE.json's north star) ----
def _worker_qshard(rank, world, port, argv_I, out_path, use_gpu=False, parts_api=False, packed=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = api.load_library() if use_gpu else api.load_library(os.path.join(ROOT, "tests", "emu", "liblqcov_emu.so"))
        tn, ts, _ = read_fastx(os.path.join(GOLDEN, "adv_all.fa.gz"))
        qn, qs, qq = read_fastx(os.path.join(GOLDEN, "adv_sub.fq.gz"))
        p = api.Params(); lib.lqcov_params_default(p)
        p.no_self = 1; p.min_ovlp = 0; p.min_score_med = 160; p.min_score_good = 160; p.batch_size = argv_I
        eng = api.Engine(p, 0, lib=lib)
        runner = multigpu.QueryShardRunner(eng, world, rank, torch.device("cuda", 0) if use_gpu else torch.device("cpu"))
        runner.set_queries(qn, qs, qq)
        lens = [int(s.shape[0]) for s in ts]
        if parts_api:
            # the parts in a pipeline: two part objects, the front of part i + 1 (upload, sketch, all-gather, index) on a host
            # thread under the mapping of part i (on a GPU; one after the other on the test emulator), persistent exchange buffers
            plan = []
            P = None
            if packed:      # every rank packs the reads on the host (bench.py: only its own shares), the runner all-gathers the packed shares
                flat = np.concatenate(ts); off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
                P = api.PackedReads(flat, off, list(tn), lib=lib)
            for (s, e) in multigpu.split_parts(lens, argv_I):
                lo, hi = multigpu.balanced_ranges(lens[s:e], world)[rank]
                if packed:
                    plan.append(((P, s + lo, s + hi), lo, tn[s:e], lens[s:e]))
                    continue

                def add(pt, s=s, lo=lo, hi=hi):
                    if hi > lo:
                        eng.part_add_targets(pt, tn[s + lo:s + hi], ts[s + lo:s + hi])
                plan.append((add, lo, tn[s:e], lens[s:e]))
            runner.map_parts([eng.part_begin(), eng.part_begin()], plan)
            table = runner.gather_table()
            if rank == 0:
                open(out_path, "w").write(table)
            eng.close()
            return
        pid = eng.part_begin()
        for (s, e) in multigpu.split_parts(lens, argv_I):
            lo, hi = multigpu.balanced_ranges(lens[s:e], world)[rank]
            eng.part_clear(pid)
            if hi > lo:
                eng.part_add_targets(pid, tn[s + lo:s + hi], ts[s + lo:s + hi])
            runner.map_part(pid, lo, tn[s:e], lens[s:e])
            assert len(runner.last_sizes) == world and runner.last_exchange_bytes == multigpu.exchange_peak_bytes(runner.last_sizes, world)
        table = runner.gather_table()
        if rank == 0:
            open(out_path, "w").write(table)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,I,expect", [(2, 100000, "adv_parts.table.gz"), pytest.param(3, 100000, "adv_parts.table.gz", marks=slow_emu), (2, 4000000000, "adv_ont.table.gz")])
def test_query_sharded_replicated_index_equals_reference_table(emu_lib, tmp_path, world, I, expect):
    """the north-star split: every rank sketches a share of each part, the minimizers are all-gathered, every rank builds the
    same index and maps its share of the queries; rows gathered on rank 0.  10 parts with the COVT cap, and one part."""
    out = str(tmp_path / "t.tsv")
    mp.spawn(_worker_qshard, args=(world, _free_port(), I, out), nprocs=world, join=True)
    assert open(out).read() == read_gz(expect)


@pytest.mark.parametrize("world", [2, pytest.param(3, marks=slow_emu)])
def test_query_sharded_parts_in_a_pipeline_equal_reference_table(emu_lib, tmp_path, world):
    """QueryShardRunner.map_parts: the same split with two part objects and persistent exchange buffers (the fronts run under the
    mappings on a GPU; here one after the other), 10 parts with the COVT cap"""
    out = str(tmp_path / "t.tsv")
    mp.spawn(_worker_qshard, args=(world, _free_port(), 100000, out, False, True), nprocs=world, join=True)
    assert open(out).read() == read_gz("adv_parts.table.gz")


@pytest.mark.parametrize("world", [2, pytest.param(3, marks=slow_emu)])
def test_query_sharded_packed_reads_exchange_equals_reference_table(emu_lib, tmp_path, world):
    """round 6's front of the north-star split: every rank uploads 1 / N of a part's 2-bit packed reads, the packed reads are
    all-gathered (0.375 B per base instead of 16 B per minimizer), every rank sketches and indexes the whole part
    (lqcov_part_add_packed_shares_dev); 10 parts with the COVT cap, byte for byte the reference's table"""
    out = str(tmp_path / "t.tsv")
    mp.spawn(_worker_qshard, args=(world, _free_port(), 100000, out, False, True, True), nprocs=world, join=True)
    assert open(out).read() == read_gz("adv_parts.table.gz")


def test_scaling_model_of_the_pipelined_parts():
    """configs[3] (25 parts of 4 Gbases) with the stage rates of the end of round 6: the front of a part (all-gather of the packed reads
    without ambiguity words at one xGMI link, 6 ms, + the replicated sketch and index build: 0.07 s in all) does not shrink with N and is
    longer than the part's mapping on ONE GPU's eighth of the queries (0.22 s / 8), so sharded queries stop near 3x at configs[2]'s
    mapping rate (1.6x with the minimizers all-gathered, round 5); putting the fronts under the mappings still helps, index parts
    across the GPUs help more when there are parts enough"""
    parts = [4.0e9] * 25
    t1 = multigpu.QueryShardRunner.scaling_model(1, parts)
    q8, q8_serial = multigpu.QueryShardRunner.scaling_model(8, parts, pipelined=True), multigpu.QueryShardRunner.scaling_model(8, parts, pipelined=False)
    assert q8 < 0.9 * q8_serial and 2.5 < t1 / q8 < 3.5              # bound by the replicated front: all-gather + sketch + index build per part
    p8 = multigpu.PartRunner.scaling_model(8, parts)
    assert t1 / p8 > 4.0 and p8 < q8                                  # 25 parts over 8 GPUs: four rounds -- the split bench.py then picks (4.5x of the pipelined single-GPU job)
    two = [4.0e9, 1.18e9]                                             # configs[2]: two parts cannot fill eight GPUs, queries are sharded
    assert multigpu.QueryShardRunner.scaling_model(8, two) < multigpu.PartRunner.scaling_model(8, two)
    assert multigpu.QueryShardRunner.scaling_model(1, two) / multigpu.QueryShardRunner.scaling_model(8, two) > 3.0
    assert abs(multigpu.QueryShardRunner.scaling_model(1, two) - 0.357) < 0.04   # the single-GPU step the rates were read from (measured: 0.357 s)
    # the mapping of a 40x ONT set costs three times as much per base (the full-size configs[3] run: 19.5 s for 99 Gbases): sharded queries then scale too
    assert multigpu.QueryShardRunner.scaling_model(1, parts, map_s_per_gbase=0.17) / multigpu.QueryShardRunner.scaling_model(8, parts, map_s_per_gbase=0.17) > 5.0


def test_balanced_ranges_and_query_shards():
    lens = [5, 1, 1, 1, 8, 2, 2]
    r = multigpu.balanced_ranges(lens, 3)
    assert r[0][0] == 0 and r[-1][1] == len(lens) and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    sh = multigpu.shard_queries(lens, 2)
    assert sorted(sum(sh, [])) == list(range(len(lens)))
    assert abs(sum(lens[i] for i in sh[0]) - sum(lens[i] for i in sh[1])) <= 2
    assert multigpu.balanced_ranges([3, 3], 4)[-1][1] == 2            # more ranks than reads: empty shares


def test_exchange_memory_bound_at_eight_ranks():
    """the minimizer exchange of a 4-Gbase part (1.34 G minimizers) cut by bases into 8 shares within a percent of each other:
    one send and one receive buffer per array + the part's own copy stay below 2.2 x 16 B per minimizer"""
    M = 1340000000
    rng = np.random.default_rng(3)
    sizes = [int(M / 8 * (1 + 0.01 * (rng.random() - 0.5))) for _ in range(8)]
    assert multigpu.exchange_peak_bytes(sizes, 8) <= 2.2 * 16 * sum(sizes)
    assert multigpu.exchange_peak_bytes([M], 1) == 2 * 16 * M
