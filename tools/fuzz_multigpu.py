#!/usr/bin/env python3
"""The N > 1 splits of longqc_amd/multigpu.py on random inputs, without a GPU: 2 or 3 `gloo` ranks, each driving the emulator build
of the engine (tests/emu), on a random read set with a random -I (1 to ~12 index parts) -- index parts across the ranks (PartRunner:
mid_occ from part 0, the COVT cap and avg_k replayed in part order), queries sharded over a replicated index (QueryShardRunner: the
minimizer all-gather), and the same with the parts in a pipeline -- against the table the reference binary prints for the same files.
    python tools/fuzz_multigpu.py --seeds 0:20"""
import argparse
import dataclasses
import os
import socket
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, mode, argv, tf, qf, out_path):
    import torch
    import torch.distributed as dist
    from longqc_amd import api, multigpu
    from tests.helpers import read_fastx
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = api.load_library(os.environ.get("LQCOV_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "liblqcov_emu.so"))
        tn, ts, _ = read_fastx(tf)
        qn, qs, qq = read_fastx(qf)
        p, _, _ = api.parse_args(argv + [tf, qf])
        I = int(p.batch_size)
        eng = api.Engine(p, 0, lib=lib)
        lens = [int(s.shape[0]) for s in ts]
        parts = multigpu.split_parts(lens, I, int(p.idx_mini_batch))
        dev = torch.device("cpu")
        if mode == "parts":
            eng.set_queries(qn, qs, qq)
            runner = multigpu.PartRunner(eng, world, rank, dev, [int(s.shape[0]) for s in qs])
            runner.begin()
            for base in range(0, len(parts), world):
                mine, pid = base + rank, None
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
        else:
            runner = multigpu.QueryShardRunner(eng, world, rank, dev)
            runner.set_queries(qn, qs, qq)
            if mode == "queries_pipeline":
                plan = []
                for (s, e) in parts:
                    lo, hi = multigpu.balanced_ranges(lens[s:e], world)[rank]

                    def add(pt, s=s, lo=lo, hi=hi):
                        if hi > lo:
                            eng.part_add_targets(pt, tn[s + lo:s + hi], ts[s + lo:s + hi])
                    plan.append((add, lo, tn[s:e], lens[s:e]))
                runner.map_parts([eng.part_begin(), eng.part_begin()], plan)
            else:
                pid = eng.part_begin()
                for (s, e) in parts:
                    lo, hi = multigpu.balanced_ranges(lens[s:e], world)[rank]
                    eng.part_clear(pid)
                    if hi > lo:
                        eng.part_add_targets(pid, tn[s + lo:s + hi], ts[s + lo:s + hi])
                    runner.map_part(pid, lo, tn[s:e], lens[s:e])
            table = runner.gather_table()
            if rank == 0:
                open(out_path, "w").write(table)
        eng.close()
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:12")
    ap.add_argument("--limit", type=int, default=600, help="seconds per case")
    ap.add_argument("--cnt-bits", type=int, default=0, help="match counters narrowed to this many bits on both sides (LQCOV_TEST_CNT_BITS / LQO_CNT_BITS): deep read sets "
                    "then fill them in the sums merged over the ranks, and PartRunner must replay the flagged queries' chains across ranks (esterr.c:127-138); "
                    "index parts across the ranks only, against the oracle (the reference binary's counters are 16 bits wide)")
    args = ap.parse_args()
    import numpy as np
    import torch.multiprocessing as mp
    from longqc_amd import synth
    from tests import oracle_bind
    lo, hi = (int(v) for v in args.seeds.split(":"))
    bad, t0 = [], time.time()
    for seed in range(lo, hi):
        rng = np.random.default_rng([seed, 77])
        cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=int(rng.integers(40, 260)), mean_len=int(rng.integers(400, 2500)), min_len=int(rng.integers(50, 300)),
                                  depth=float(rng.choice([3, 8, 20, 60])), err=float(rng.choice([0.02, 0.08, 0.13])), seed=9000 + seed, nsample=int(rng.integers(4, 25)),
                                  qual=str(rng.choice(["ont", "none"])), junk_frac=float(rng.choice([0.0, 0.05])))
        if args.cnt_bits:
            cfg = dataclasses.replace(cfg, depth=float(rng.choice([40, 80, 160])), n_reads=int(rng.integers(150, 400)))
        if cfg.genome_len < 12000:                 # (a genome of a few kb is one tandem repeat away from a mid_occ of thousands: millions of seed hits,
            cfg = dataclasses.replace(cfg, depth=max(1.5, cfg.n_reads * cfg.mean_len / 12000.0))   # minutes per case on the emulator; tools/fuzz_emu.py has those)
        T, Q = synth.make_dataset(cfg)
        world = int(rng.choice([2, 2, 3]))
        mode = str(rng.choice(["parts", "queries", "queries_pipeline"]))
        if args.cnt_bits:
            mode = "parts"
        total = T.n_bases
        I = str(rng.choice(["4G", "%dK" % max(5, total // 1000 // int(rng.integers(2, 12)))]))
        if args.cnt_bits:
            I = "%dK" % max(5, total // 1000 // int(rng.integers(3, 12)))
        k, w = int(rng.choice([10, 12, 12, 15])), int(rng.choice([5, 5, 10]))
        m = int(rng.choice([20, 40, 40, 60]))
        argv = ["-Y", "-l", "0", "-k", str(k), "-w", str(w), "-I", I, "-m", str(m), "-p", str(m + int(rng.choice([0, 40, 120]))), "-q", str(m + 120), "-t", "2"]
        if rng.random() < 0.2:
            argv += ["-c", str(rng.choice([1, 2]))]
        with tempfile.TemporaryDirectory() as d:
            tf, qf, out = os.path.join(d, "all.fq"), os.path.join(d, "sub.fq"), os.path.join(d, "t.tsv")
            synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
            if args.cnt_bits:
                os.environ["LQO_CNT_BITS"] = str(args.cnt_bits); os.environ["LQCOV_TEST_CNT_BITS"] = str(args.cnt_bits)
                want = oracle_bind.table(argv + [tf, qf])
                os.environ["LQO_CNT_BITS"] = "16"
                wide = oracle_bind.table(argv + [tf, qf])
                os.environ["LQO_CNT_BITS"] = str(args.cnt_bits)
            else:
                want = oracle_bind.ref_table(argv + [tf, qf])
            try:
                ctx = mp.spawn(worker, args=(world, free_port(), mode, argv, tf, qf, out), nprocs=world, join=False)
                deadline = time.time() + args.limit
                while not ctx.join(timeout=1.0):
                    if time.time() > deadline:
                        for pr in ctx.processes:
                            pr.kill()
                        raise TimeoutError("not finished within %d s" % args.limit)
                got = open(out).read()
            except Exception as e:
                got = "EXCEPTION %r" % (e,)
            ok = got == want
            print("seed %d: %s, %d ranks, %s: %s (%d rows, %d with coverage%s)" % (seed, mode, world, " ".join(argv), "identical" if ok else "DIFFERS", len(want.splitlines()),
                                                                                  sum(1 for l in want.splitlines() if l.split("\t")[2] != "0"),
                                                                                  ", the counters' width shows in the table" if args.cnt_bits and wide != want else ""), flush=True)
            if not ok:
                bad.append(seed)
                print(got[:600])
    print("%d cases in %.0f s, findings: %s" % (hi - lo, time.time() - t0, bad or "none"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
