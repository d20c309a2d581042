#!/usr/bin/env python3
"""bench.py -- throughput of the sampleqc hot path (all reads -> index parts, subsample -> coverage table)
on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload at N=1 (BASELINE.json configs[2], the largest single-GPU configuration): 500k synthetic PacBio
Sequel CLR reads ~10 kb, 30x, pb-sequel preset (-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 80, longQC.py:171-220);
the reference's -I rule (index.c:244,311-316) cuts them into two index parts (4.0 + 1.0 Gbases); query set
= LongQC's seed-7 subsample of 5000 reads.  `--config cfg2` selects configs[1] (50k ONT reads ~15 kb, 15x).

One "step" = one whole job by SURVEY.md 8(d)'s clock: reset accumulators -> for every index part: H2D of
the 2-bit packed reads from page-locked host memory -> sketch -> index (+ mid_occ) -> seed / klib-order sort
/ chain / coverage for every query -> rows (D2H).  What happens before the clock: FASTQ parse (here: the
synthetic generator) and the host-side 2-bit packing that the parser thread does (reported as host_pack_s).
`value` = target bases / step time.  `hbm_resident_value` leaves the H2D out (reads already packed in HBM).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

# HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); kernels of streams that share a queue run
# one after the other.  The engine drives 3 streams per mapping lane + 1: ask for eight queues before the runtime starts
# (liblqcov.so does the same in lqcov_create when it is the first HIP user of the process; measured: 4 queues 1.74-1.77 s
# per step at configs[2], 8: 1.71-1.73, 16: 1.73).
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("LQCOV_HW_QUEUES", "8"))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

_ONT = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160"]
PRESET = {   # the argv longQC.py issues for the config's platform (longQC.py:177-231,440-445)
    "cfg2": ("ont-ligation", _ONT),
    "cfg3": ("pb-sequel", ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "80"]),
    "cfg4": ("ont-rapid", _ONT), "cfg4s": ("ont-rapid", _ONT),          # (ont-rapid and ont-ligation issue the same minimap parameters)
    "cfg5": ("ont-ligation", _ONT), "cfg5s": ("ont-ligation", _ONT),
}
CONFIG_LABEL = {
    "cfg2": "BASELINE configs[1]: %d synthetic ONT reads ~%d kb %gx, ont-ligation preset (-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 160)",
    "cfg3": "BASELINE configs[2]: %d synthetic PacBio Sequel CLR reads ~%d kb %gx, pb-sequel preset (-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 80)",
    "cfg4": "BASELINE configs[3]: %d synthetic ONT reads ~%d kb %gx, ont-rapid preset (-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 160)",
    "cfg4s": "a slice of BASELINE configs[3] with index parts of real size: %d synthetic ONT reads ~%d kb %gx, ont-rapid preset (-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 160)",
    "cfg5": "BASELINE configs[4]: %d synthetic ultra-long ONT reads (mean ~%d kb, N50 ~100 kb) %gx, ont-ligation preset (-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 160)",
    "cfg5s": "a slice of BASELINE configs[4] with index parts of real size: %d synthetic ultra-long ONT reads (mean ~%d kb, N50 ~100 kb) %gx, ont-ligation preset (-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 160)",
}


def cpu_baseline(cfg_name, F, Q, n_t, n_q):
    """The reference's own minimap2-coverage (oracle/_ref, kind "reference") or the C restatement
    (kind "port") on a bounded sample of the same workload, timed on this host's cores."""
    from longqc_amd import synth
    from tests import oracle_bind
    cores = os.cpu_count() or 1
    n_t = min(n_t, len(F)); n_q = min(n_q, len(Q))
    sub_q = Q.subset(range(n_q))
    with tempfile.TemporaryDirectory() as d:
        tf, qf = os.path.join(d, "t.fa"), os.path.join(d, "q.fq")
        sub_t = synth.FlatReads(F.first, F.flat[:int(F.off[n_t])], F.off[:n_t + 1])
        synth.write_flat_fasta(tf, sub_t)
        synth.write_fastq(qf, sub_q)
        argv = PRESET[cfg_name][1]
        if oracle_bind.have_ref():
            kind, used = "reference", cores
            cmd = [oracle_bind.REF_BIN] + argv + ["-t", str(cores), tf, qf]
        else:
            kind, used = "port", 1
            cmd = [oracle_bind.ensure_oracle(), "table"] + argv + [tf, qf]
        t0 = time.time()
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        if r.returncode != 0:
            return None
    return {"value": round(sub_t.n_bases / dt / 1e6, 3), "unit": "Mbases/s", "cores": used, "kind": kind,
            "seconds": round(dt, 2),
            "sample": ("all %d target reads (%.1f Mbases, FASTA) + all %d subsample reads as queries, same argv, file parse included" % (n_t, sub_t.n_bases / 1e6, n_q))
                      if n_t >= len(F) and n_q >= len(Q) else
                      ("first %d target reads (%.1f Mbases, FASTA) + first %d subsample reads as queries, same argv, "
                       "file parse included; the reference's sketch/index step is ~serial (3 fixed threads, index.c:293-300) and seed hits "
                       "per query grow with the index: the whole job is slower per base (full_job, if present: measured once on all reads)"
                       % (n_t, sub_t.n_bases / 1e6, n_q))}


def copy_ceiling(local):
    """What a plain device copy reaches on this box ("also report against a measured device-copy ceiling"): bytes read +
    written per second of a 2-GiB torch copy on cuda:<local>, best of five, with nothing else on the device.  None if it cannot run."""
    try:
        import torch
        n = 1 << 31
        dev = torch.device("cuda", local)
        a = torch.zeros(n, dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize(dev)
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        del a, b
        torch.cuda.empty_cache()
        return round(2.0 * n / (best / 1e3) / 1e9, 1) if best and best > 0 else None
    except Exception:
        return None


def golden_check(cfg_name, table_text):
    """rows of the reference itself for 40 of the queries, made on the whole read set in the build container
    (tests/golden/make_scale_golden.py); None if the fixture is not there or the run is not the full config"""
    fn = os.path.join(ROOT, "tests", "golden", cfg_name + "_rows.json")
    if not os.path.exists(fn):
        return None
    g = json.load(open(fn))
    lines = table_text.splitlines()
    bad = [s for s, row in zip(g["subsample_slots"], g["rows"]) if s >= len(lines) or lines[s] != row]
    return {"rows_checked": len(g["rows"]), "rows_identical": len(g["rows"]) - len(bad), "first_mismatch_slot": bad[0] if bad else None,
            "fixture": os.path.basename(fn), "note": "rows printed by the reference binary for these queries against the whole read set"}


def table_properties(table_text, names, lengths):
    """Size-independent checks of a coverage table (configs too large for reference rows): one row per query in the queries' order, the
    name and length columns, regions inside [0, length], ascending and disjoint, the coverage columns finite and equal to lambda / covered
    bases (minimap2-coverage.c:563-605).  Returns counts; "ok" only if every row passes."""
    lines = table_text.splitlines()
    bad = []
    n_reg = 0; cov = []
    if len(lines) != len(names):
        bad.append("rows %d != queries %d" % (len(lines), len(names)))
    for i, (l, nm, ln) in enumerate(zip(lines, names, lengths)):
        f = l.split("\t")
        try:
            assert len(f) == 9, "columns"
            assert f[0] == nm, "name"
            assert int(f[1]) == int(ln), "length"
            lam = int(f[2]); assert lam >= 0
            tot = 0
            for col in (3, 4):
                if f[col] == "0":
                    continue
                prev = -1
                for r in f[col].split(","):
                    a, b = (int(x) for x in r.split("-"))
                    assert 0 <= a < b <= int(ln) and a >= prev, "region"
                    prev = b
                    if col == 3:
                        tot += b - a
            if f[3] != "0":
                n_reg += 1
                assert abs(float(f[5]) - lam / tot) < 6e-4 + 1e-6 * lam / tot, "coverage column"
                cov.append(lam / tot)
            for col in (5, 6, 7, 8):
                v = float(f[col]); assert v == v or col == 6, "nan"     # (meanQ of a FASTA query prints -nan, lqutils.c:57)
        except Exception as e:
            bad.append("row %d: %s" % (i, e))
            if len(bad) > 5:
                break
    return {"ok": not bad, "rows": len(lines), "rows_with_regions": n_reg, "mean_coverage_of_covered": round(float(np.mean(cov)), 3) if cov else None,
            "max_coverage": round(float(np.max(cov)), 3) if cov else None, "problems": bad[:5]}


def sdust_bench(args):
    """`bench.py --config sdust`: the low-complexity table of every read of configs[1] (50 000 synthetic ONT reads ~15 kb, 744 Mbases) --
    what lq_mask.py gets from the reference's `sdust` binary chunk by chunk (lq_mask.py:17-23,99-121; sdust.c:136-222).  A step is one call
    of lqsdust_main on the FASTQ file (tmpfs): parse + upload + k_sdust + meanQ + rows written; value = bases / step.  cpu_baseline: the
    reference's own `sdust` (oracle/_ref) on the same reads, cut into one chunk file per worker as lq_mask.py cuts them, all workers at
    once on the host's cores; its concatenated output must equal the GPU's table."""
    import dataclasses, shutil
    import torch
    from longqc_amd import synth, sdust
    from tests import oracle_bind
    n = args.reads or 50000
    cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=n, nsample=10)
    t0 = time.time()
    T, _ = synth.make_dataset(cfg)
    gen_s = time.time() - t0
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 6 * T.n_bases else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        fq, out = os.path.join(d, "all.fq"), os.path.join(d, "sdust.tsv")
        synth.write_fastq(fq, T)
        for _ in range(args.warmup):
            sdust.run_sdust(fq, out)
        ts = []
        for _ in range(args.steps):
            torch.cuda.synchronize()
            t0 = time.time(); sdust.run_sdust(fq, out); ts.append(time.time() - t0)
        dt = sum(ts) / len(ts)
        table = open(out).read()
        line = {"metric": "Mbases/sec sdust low-complexity table (sampleqc, lq_mask.py)", "value": round(T.n_bases / dt / 1e6, 3), "unit": "Mbases/s", "n_gpus": 1,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u8", "data": "synthetic",
                "config": {"workload": "SURVEY 8(f)-4: `sdust <all reads>` on BASELINE configs[1]'s reads (%d synthetic ONT reads ~15 kb, %d bases, FASTQ on %s): masked bases, length, fraction, meanQ, "
                                       "QV7 count per read" % (len(T), T.n_bases, base or "the temp dir"),
                           "clock": "one lqsdust_main call: FASTQ parse + H2D + k_sdust (one read per thread, sdust.c:70-134) + meanQ / QV7 + rows written; wall clock", "parallelism": "single GPU"},
                "rows": table.count("\n"), "synth_gen_s": round(gen_s, 1)}
        ref = os.path.join(os.path.dirname(oracle_bind.REF_BIN), "sdust")
        if os.path.exists(ref) and not args.no_cpu_baseline:
            cores = min(os.cpu_count() or 1, 64)
            cuts = [len(T) * i // cores for i in range(cores + 1)]
            files = []
            for i in range(cores):
                if cuts[i + 1] > cuts[i]:
                    f = os.path.join(d, "chunk%d.fq" % i); synth.write_fastq(f, T.subset(range(cuts[i], cuts[i + 1]))); files.append(f)
            t0 = time.time()
            procs = [subprocess.Popen([ref, f], stdout=open(f + ".out", "w"), stderr=subprocess.DEVNULL) for f in files]
            rcs = [pr.wait() for pr in procs]
            dtr = time.time() - t0
            ref_table = "".join(open(f + ".out").read() for f in files)
            line["cpu_baseline"] = {"value": round(T.n_bases / dtr / 1e6, 3), "unit": "Mbases/s", "seconds": round(dtr, 2), "cores": len(files), "kind": "reference",
                                    "sample": "ALL reads of the workload in %d chunk files, one reference `sdust` process per chunk, all at once (lq_mask.py's pool)" % len(files),
                                    "table_identical_to_gpu": all(r == 0 for r in rcs) and ref_table == table}
            t0 = time.time(); subprocess.run([ref, files[0]], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); d1 = time.time() - t0
            nb1 = sum(int(x.shape[0]) for x in T.seqs[cuts[0]:cuts[1]])
            line["cpu_baseline"]["one_process"] = {"value": round(nb1 / d1 / 1e6, 3), "unit": "Mbases/s", "cores": 1, "sample": "the first chunk alone"}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=sorted(PRESET) + ["sdust"], help="sdust: the other native binary of sampleqc (SURVEY 8(f)-4), a bench line of its own")
    ap.add_argument("--reads", type=int, default=0, help="override the number of reads (default: the config's)")
    ap.add_argument("--nsample", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split", default="auto", choices=["auto", "queries", "parts"], help="N > 1: queries sharded over a replicated index | index parts across the GPUs | whichever the stage-rate model predicts faster")
    ap.add_argument("--sharded-reads", action="store_true", help="N > 1: every rank generates only the reads of its shares (automatic for configs above 20 Gbases)")
    ap.add_argument("--no-pipeline", action="store_true", help="N > 1: the front of part i + 1 (upload, sketch, all-gather, index) after the mapping of part i, not under it")
    ap.add_argument("--no-north-star", action="store_true", help="skip the extra pass that times the sketch and seed kernels alone on the device")
    ap.add_argument("--gz-end-to-end", action="store_true", help="end_to_end once more with the targets as one gzip stream (LongQC's usual .fastq.gz input): reported as end_to_end.gz")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the files-in / table-out call (lqcov_run_files on the workload written to tmpfs)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="target reads in the CPU baseline sample (~20-30 s of reference time)")
    ap.add_argument("--cache", default="", help="directory for the generated reads (re-used by later runs of the same config on this machine)")
    ap.add_argument("--index-size", default="", help="override the preset's -I (e.g. 2M: many index parts on a small input; tests)")
    ap.add_argument("--workers", type=int, default=0, help="processes of the synthetic generator (0: one per core, at most 64)")
    ap.add_argument("--front-only", action="store_true", help="not the bench: upload + sketch + index of the first part alone, --steps times, nothing else on the device; "
                                                                "prints the per-kernel HIP-event times of the build side undisturbed (development aid)")
    args = ap.parse_args()
    if args.config == "sdust":
        return sdust_bench(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # test hook for 1-GPU boxes: LQCOV_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo (RCCL needs one GPU per rank)
    one_dev = os.environ.get("LQCOV_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
    have_cuda = torch.cuda.is_available()        # False only in the CPU dry run of this script's logic (LQCOV_LIBRARY = the test emulator)
    if have_cuda:
        torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("gloo" if one_dev else "nccl", rank=rank, world_size=world)

    import dataclasses
    from longqc_amd import api, synth, multigpu
    cfg = synth.SCALE_SLICES[args.config] if args.config in synth.SCALE_SLICES else synth.CONFIGS[args.config]
    full_config = not args.reads and not args.nsample
    if args.reads:
        cfg = dataclasses.replace(cfg, n_reads=args.reads)
    if args.nsample:
        cfg = dataclasses.replace(cfg, nsample=args.nsample)
    argv = list(PRESET[args.config][1])
    if args.index_size:
        argv[argv.index("-I") + 1] = args.index_size
        full_config = False
    p, _, _ = api.parse_args(argv + ["t", "q"])

    t0 = time.time()
    genome = synth.make_genome(cfg)
    F = None
    cache = None
    sharded = world > 1 and (args.sharded_reads or cfg.n_reads * cfg.mean_len > 2e10)      # N GPUs on a big config: no rank holds all reads
    if not sharded:
        if args.cache:                                                            # A/B loops on one box: generate the reads once
            cache = os.path.join(args.cache, "lqcov_%s_%d_seed%d" % (args.config, cfg.n_reads, cfg.seed))
            if os.path.exists(cache + ".off.npy"):
                F = synth.FlatReads(0, np.load(cache + ".flat.npy"), np.load(cache + ".off.npy"))
        if F is None:
            F = synth.make_reads_flat(cfg, genome, workers=args.workers)         # every read of the config, flat ASCII
            if cache and rank == 0:
                os.makedirs(args.cache, exist_ok=True)
                np.save(cache + ".flat.npy", F.flat); np.save(cache + ".off.tmp.npy", F.off); os.replace(cache + ".off.tmp.npy", cache + ".off.npy")
        lens = np.diff(F.off).astype(np.int64)
        total_bases = float(F.n_bases)
    else:
        # Every rank generates a contiguous 1/N of the reads to learn their lengths (a read's length is only known once its errors
        # are drawn), the lengths are all-gathered, the parts and every rank's shares of them follow from the lengths, and a rank
        # then generates exactly the reads of its shares (every read has its own RNG stream: synth._one_read).
        wk = args.workers or max(1, min(64, (os.cpu_count() or 1) // world))
        n = cfg.n_reads
        a0, b0 = n * rank // world, n * (rank + 1) // world
        mine = np.diff(synth.make_reads_flat(cfg, genome, n_reads=b0 - a0, read_offset=a0, workers=wk).off).astype(np.int64)
        got = [None] * world
        dist.all_gather_object(got, mine)
        lens = np.concatenate(got)
        total_bases = float(lens.sum())
    qidx = synth.reservoir_subsample(cfg.n_reads, cfg.nsample)                   # LongQC's seed-7 subsample (lq_utils.py:371-411)
    Q = synth.make_reads(cfg, genome, indices=qidx)
    parts = multigpu.split_parts(lens, int(p.batch_size), int(p.idx_mini_batch))  # index.c:244,311-316
    # N GPUs, two exact splits (longqc_amd/multigpu.py): queries sharded over a replicated index (the north star; its front --
    # all-gather + index build -- does not shrink with N) or index parts across the GPUs (needs >= N parts to fill them).  The
    # one the model of the measured single-GPU stage rates predicts to be faster runs, unless --split says otherwise.
    part_bases = [int(lens[lo:hi].sum()) for lo, hi in parts]
    split = "none"
    if world > 1:
        tq, tp = multigpu.QueryShardRunner.scaling_model(world, part_bases), multigpu.PartRunner.scaling_model(world, part_bases)
        split = args.split if args.split != "auto" else ("parts" if tp < tq else "queries")
    if split == "parts":
        shares = [(0, hi - lo) if i % world == rank else (0, 0) for i, (lo, hi) in enumerate(parts)]      # whole parts, round robin
    else:
        shares = [multigpu.balanced_ranges(lens[lo:hi], world)[rank] for lo, hi in parts] if world > 1 else None
    t_gen = time.time() - t0
    t0 = time.time()
    if not sharded:
        P = api.PackedReads(F.flat, F.off, F.names())                            # what the parser thread does: 2-bit pack into pinned memory
        share_reads = [(P, lo + sa, lo + sb) for (lo, hi), (sa, sb) in zip(parts, shares)] if world > 1 else None
    else:
        share_reads = []
        for (lo, hi), (sa, sb) in zip(parts, shares):
            Fs = synth.make_reads_flat(cfg, genome, n_reads=sb - sa, read_offset=lo + sa, workers=wk)
            share_reads.append((api.PackedReads(Fs.flat, Fs.off, Fs.names()), 0, sb - sa))
            del Fs
        P = None
    t_pack = time.time() - t0

    def names_of(lo, hi):
        return ["r%07d" % i for i in range(lo, hi)]                               # (synth.FlatReads.names)

    eng = api.Engine(p, device=local)
    anchors = [0]
    written = [0]                                                                 # anchors the first pass of the mapping wrote (k_seed_count's survivors), per step
    mstats = [None]
    if world == 1:
        eng.set_queries(Q.names, Q.seqs, Q.quals)
        pts = [eng.part_begin(), eng.part_begin()] if len(parts) > 1 else [eng.part_begin()]
        pt = pts[0]
        if len(parts) > 1:
            # part i + 1 is uploaded, sketched and indexed (a host thread of its own, the engine's build stream) while part i is
            # mapped: the lanes leave room for the second part object (0.375 B per base of reads, 24 B per minimizer ~ 3 bases, tables)
            eng.reserve_hbm(int(max(int(lens[lo:hi].sum()) for lo, hi in parts[1:]) * 9.5))
        import threading

        def build(i):
            eng.part_clear(pts[i % 2])
            eng.part_add_packed(pts[i % 2], P, parts[i][0], parts[i][1])
            eng.part_build(pts[i % 2])

        def step(h2d=True):
            eng.reset()
            a = w = 0
            if h2d or len(parts) > 1:
                build(0)
            else:
                eng.part_build(pts[0])
            for i in range(len(parts)):
                th = None
                err = []
                if i + 1 < len(parts) and have_cuda:
                    def guarded(j=i + 1):
                        try:
                            build(j)
                        except BaseException as e:                # (a thread's exception would vanish: carried to the caller)
                            err.append(e)
                    th = threading.Thread(target=guarded); th.start()
                eng.part_map(pts[i % 2])
                a += eng.last_n_anchors
                w += eng.map_stats()["last_written"]
                if th is not None:
                    th.join()
                    if err:
                        raise err[0]
                elif i + 1 < len(parts):
                    build(i + 1)                                  # (the CPU dry run of this script: the test emulator runs one kernel at a time)
            eng.finish()
            anchors[0] = a; written[0] = w; mstats[0] = eng.map_stats()
    elif split == "parts":
        # N GPUs, index parts across them: rounds of N consecutive parts, every rank uploads, sketches, indexes and maps ALL queries
        # against its own part; mid_occ from part 0's owner (map.c:50), the COVT cap and avg_k replayed in part order from the
        # all-gathered per-part lambdas (esterr.c:87,93-97), sums and counters all-reduced, intervals all-gathered.
        dev = torch.device("cuda", local) if have_cuda else torch.device("cpu")
        eng.set_queries(Q.names, Q.seqs, Q.quals)
        prunner = multigpu.PartRunner(eng, world, rank, dev, [int(x.shape[0]) for x in Q.seqs])
        pt = eng.part_begin()
        table = [None]

        def step(h2d=True):
            prunner.begin()
            a = 0
            for base in range(0, len(parts), world):
                mine = base + rank
                pid = None
                if mine < len(parts):
                    Ps, s0, s1 = share_reads[mine]
                    eng.part_clear(pt); eng.part_add_packed(pt, Ps, s0, s1); eng.part_build(pt)
                    pid = pt
                prunner.map_and_combine(pid, part_index=mine, mid_occ_owner=0, share_mid_occ=(base == 0))
                if pid is not None:
                    a += eng.last_n_anchors
            eng.finish()
            table[0] = eng.table_text() if rank == 0 else None
            anchors[0] = a; mstats[0] = eng.map_stats()
    else:
        # N GPUs, the north-star split (longqc_amd/multigpu.py): every rank sketches 1/N of each part, the minimizers are
        # all-gathered over RCCL, every rank builds the same index and maps its 1/N of the queries; rows gathered on rank 0.
        # The same job on N GPUs: strong scaling.
        dev = torch.device("cuda", local) if have_cuda else torch.device("cpu")
        runner = multigpu.QueryShardRunner(eng, world, rank, dev)
        runner.set_queries(Q.names, Q.seqs, Q.quals)
        pts = [eng.part_begin(), eng.part_begin()]
        plan = []
        for (lo, hi), (sa, sb), (Ps, s0, s1) in zip(parts, shares, share_reads):
            # (this rank's share of the part as packed reads on the host: the runner all-gathers the packed reads -- 0.375 B per base --
            # and every rank sketches and indexes the whole part; LQCOV_EXCHANGE=minimizers: rounds 4-5's all-gather of per-rank sketches)
            if os.environ.get("LQCOV_EXCHANGE") == "minimizers":
                def add(pt, Ps=Ps, s0=s0, s1=s1):
                    if s1 > s0:
                        eng.part_add_packed(pt, Ps, s0, s1)
                plan.append((add, sa, api.encode_names(names_of(lo, hi)), lens[lo:hi].astype(np.uint32)))
            else:
                plan.append(((Ps, s0, s1), sa, api.encode_names(names_of(lo, hi)), lens[lo:hi].astype(np.uint32)))
        reserve = 0
        if len(parts) > 1 and have_cuda:
            # room for the part whose front runs under the mapping (a second part object: 24 B per minimizer ~ 8 B per base, its
            # tables) and for the exchange buffers (packed reads: (N + 1) x 0.4 B per base; minimizers: 2.2 x 16 B per minimizer)
            reserve = int(max(int(lens[lo:hi].sum()) for lo, hi in parts[1:]) * (9.5 + (12.0 if os.environ.get("LQCOV_EXCHANGE") == "minimizers" else 0.42 * (world + 1) / 3.0)))
        if one_dev and have_cuda:
            # (test mode: the ranks share one device -- each leaves the others their share of what is free now; the lanes size
            # their work space from what is free when the first part stands, and two ranks doing that at once ran the device dry)
            reserve += int(torch.cuda.mem_get_info(local)[0] * (world - 1) / world)
        if reserve:
            eng.reserve_hbm(reserve)
        table = [None]

        def step(h2d=True):
            eng.reset()
            anchors[0] = runner.map_parts(pts, plan, pipeline=have_cuda and not args.no_pipeline)
            table[0] = runner.gather_table()
            mstats[0] = eng.map_stats()

    if args.front_only and world == 1:
        best = {}
        for _ in range(max(args.steps, 1)):
            eng.set_profiling(0); eng.set_profiling(2)       # (clears the stage table)
            eng.reset()
            t0 = time.time()
            build(0)
            eng.sync()
            wall = (time.time() - t0) * 1e3
            st = {s["name"]: round(s["total_ms"], 2) for s in eng.stage_times() if s["total_ms"] > 0.05}
            if not best or wall < best["wall_ms"]:
                best = {"wall_ms": round(wall, 2), "stages_ms": st}
        print(json.dumps({"front_only": True, "part_bases": int(lens[parts[0][0]:parts[0][1]].sum()), "best_of": max(args.steps, 1), **best}))
        return

    def barrier():
        if world > 1:
            dist.barrier()
        if have_cuda:
            torch.cuda.synchronize()

    def timed(n, **kw):
        barrier()
        t0 = time.time()
        for _ in range(n):
            step(**kw)
        eng.sync()
        barrier()
        return time.time() - t0

    for _ in range(args.warmup):
        step()
    # one untimed step with HIP events around every launch (level 2: events only, nothing waits for them): the per-kernel
    # table and the name of the dominant kernel; the timed steps then carry events around that kernel only
    eng.set_profiling(2)
    step()
    all_st = [s for s in eng.stage_times() if s["name"].startswith("k_") or s["name"].startswith("index_") or s["name"].startswith("h2d_")]
    kern = [s for s in all_st if not s["name"].startswith("h2d_")]
    prof_stats = eng.map_stats()                                                  # (of the profiled step: the second pass's anchors)
    # What bounds the step: the kernel -- or the group of kernels that together do one line of SURVEY 8d's byte table -- with the
    # largest device time per step (HIP events around every launch of one untimed step).  Nothing is left out (round 5 left out the
    # serial token walks and the build side, and reported the 4th-largest entry):
    #   * "anchor sort, second pass": every kernel that reproduces klib's radix sort for the queries that own an observable tie
    #     (records, histograms, two-bucket passes, checkpoint solvers, token walkers, scatters, children).  SURVEY's line: 32 B per
    #     sorted anchor (read 16, write 16), A2 = the second pass's anchors.  Eight levels of an MSD sort behind one figure.
    #   * "chain": k_run_list + k_chain + k_chain_wave, 16 B per anchor the chain stage reads (what the first pass wrote + A2);
    #     SURVEY's 16 A' + 8 C for the chained anchors and the chains are not known per step and left out (a lower bound on bytes).
    #   * every other kernel by its own algorithmic bytes (StageTimer).
    SORT2 = ("k_sort_init", "k_rs_hist", "k_sort_two", "k_ck_", "k_sort_walk", "k_rs_scatter", "k_rs_children")
    CHAIN = ("k_run_list", "k_chain")
    A2 = float((prof_stats or {}).get("klib_anchors", 0))
    groups = {}
    def add(label, s, bytes_):
        g = groups.setdefault(label, {"name": label, "members": [], "total_ms": 0.0, "launches": 0, "algo_bytes": 0.0})
        g["members"].append(s["name"]); g["total_ms"] += s["total_ms"]; g["launches"] += s["launches"]; g["algo_bytes"] += bytes_
    for s_ in kern:
        if s_["name"].startswith(SORT2):
            add("anchor sort, second pass (klib order: k_rs_*, k_sort_*, k_ck_*)", s_, 0.0)
        elif s_["name"].startswith(CHAIN):
            add("chain (k_run_list, k_chain, k_chain_wave)", s_, 0.0)
        else:
            add(s_["name"], s_, float(s_["algo_bytes"]))
    for lab, g in groups.items():
        if lab.startswith("anchor sort"):
            g["algo_bytes"] = 32.0 * A2; g["byte_model"] = "32 B per second-pass anchor (SURVEY 8d 'anchor sort': 16 read + 16 written), A2 = %d" % int(A2)
        elif lab.startswith("chain"):
            g["algo_bytes"] = 16.0 * (float(written[0]) + A2); g["byte_model"] = "16 B per anchor the chain stage reads (first-pass survivors + second-pass anchors); chained anchors and chains not counted"
    build_side = ("k_sketch", "k_mask", "index_", "k_mark", "k_fill", "k_table", "k_sort_keys", "k_head")
    cand = [g for g in groups.values() if g["algo_bytes"] > 0]
    dom_g = max(cand or list(groups.values()), key=lambda g: g["total_ms"]) if groups else None
    dom_name = dom_g["name"] if dom_g else None
    eng.set_profiling(0)
    eng.set_profiling(2, only="|".join(dom_g["members"]) if dom_g else None)
    dt = timed(args.steps)
    ms_per_step = dt / max(args.steps, 1) * 1e3
    value = total_bases / (ms_per_step / 1e3) / 1e6
    timed_stats = eng.map_stats()
    st_m = [s for s in eng.stage_times() if dom_g and s["name"] in dom_g["members"]]
    st = None
    if st_m:
        st = {"name": dom_name, "total_ms": sum(s["total_ms"] for s in st_m), "launches": sum(s["launches"] for s in st_m), "members": sorted(set(dom_g["members"])),
              "algo_bytes": (32.0 * float((timed_stats or {}).get("klib_anchors", 0)) * 1.0 if dom_name.startswith("anchor sort") else
                             16.0 * (float(written[0]) + float((timed_stats or {}).get("klib_anchors", 0))) if dom_name.startswith("chain") else
                             float(sum(s["algo_bytes"] for s in st_m))),
              "byte_model": dom_g.get("byte_model", "the kernel's own algorithmic bytes (DESIGN.md section 3)")}
        if dom_name.startswith(("anchor sort", "chain")):
            st["algo_bytes"] *= 1.0                                               # (klib_anchors counts since the last reset(): one step)
            st["per_step"] = True
    eng.set_profiling(0)
    n_anchors = anchors[0]
    if world > 1:
        ta = torch.tensor([float(n_anchors)], dtype=torch.float64, device="cuda" if have_cuda else "cpu")
        dist.all_reduce(ta, op=dist.ReduceOp.SUM)
        tm = torch.tensor([dt], dtype=torch.float64, device="cuda" if have_cuda else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        n_anchors = int(ta.item()); dt = float(tm.item())
        ms_per_step = dt / max(args.steps, 1) * 1e3
        value = total_bases / (ms_per_step / 1e3) / 1e6
        table = table[0] or ""
    else:
        table = eng.table_text()
    n_qm_counted = None
    if world == 1:
        try:
            n_qm_counted = float(sum(r["n_mini"] for r in eng.rows()))          # the queries' minimizers, counted (the rows carry them)
        except Exception:
            n_qm_counted = None
    # the same job with the reads already resident in HBM (single-part workloads only: a multi-part job re-uses the part's buffers)
    resident = None
    if len(parts) == 1 and world == 1:
        dt_r = timed(max(1, args.steps), h2d=False)
        resident = total_bases / (dt_r / max(1, args.steps)) / 1e6

    # BASELINE.json's north star asks for the HBM fraction of the sketch and seed kernels: one extra pass over the first part with
    # every kernel alone on the device (profiling level 1: each launch waited for, lanes one after the other), so that the
    # times are the kernels' own and not what they stretch to beside the other streams
    north = None
    if world == 1 and not args.no_north_star:
        eng.set_profiling(0); eng.set_profiling(1)
        eng.reset(); build(0); eng.part_map(pts[0]); eng.sync()
        u = {x["name"]: x for x in eng.stage_times()}
        eng.set_profiling(0)
        B0 = float(part_bases[0]); M0 = float(eng.part_n_minimizers(pts[0])); A0 = float(eng.last_n_anchors); W0 = float(eng.map_stats()["last_written"])
        def grp(names, algo):
            ms = sum(u[n]["total_ms"] for n in names if n in u)
            return {"kernels": {n: round(u[n]["total_ms"], 3) for n in names if n in u}, "ms": round(ms, 3), "algo_GB": round(algo / 1e9, 2),
                    "GB_per_s": round(algo / 1e9 / (ms / 1e3), 1) if ms > 0 else None, "frac_of_hbm_peak": round(algo / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4) if ms > 0 else None}
        n_qm = n_qm_counted if n_qm_counted else float(sum(int(x.shape[0]) for x in Q.seqs)) / 3.0   # (fallback: ~ one minimizer per 3 bases at w = 5)
        sk = grp(["k_sketch_dp_mask", "k_sketch_mask", "k_sketch_count", "k_sketch_emit_mask", "k_sketch_emit", "scan"], 0.25 * B0 + 16.0 * M0)
        sd = grp(["k_seed_probe", "k_dup_mark", "k_seed_count", "k_seed_scatter", "k_seed_decide", "k_seed_emit_s", "k_seed_emit"], 32.0 * n_qm + 8.0 * A0 + 16.0 * A0)
        both = {"ms": round(sk["ms"] + sd["ms"], 3), "algo_GB": round(sk["algo_GB"] + sd["algo_GB"], 2)}
        both["GB_per_s"] = round(both["algo_GB"] / (both["ms"] / 1e3), 1) if both["ms"] > 0 else None
        both["frac_of_hbm_peak"] = round(both["GB_per_s"] / HBM_PEAK_GBS, 4) if both["GB_per_s"] else None
        north = {"part_bases": int(B0), "part_minimizers": int(M0), "seed_hits": int(A0), "anchors_written": int(W0), "sketch": sk, "seed": sd, "combined": both,
                 "bytes": "SURVEY 8d: sketch 0.25 B + 16 M; seed 32 M_q + 8 A + 16 A with A = every seed hit below mid_occ (what the reference writes and sorts). "
                          "This engine streams the occurrence lists twice (count, scatter), moves every hit once as an 8-byte record to its (query, slice of targets) bucket, "
                          "decides per bucket which hits can reach a chain and writes only those as anchors (anchors_written): on its own "
                          "bytes (2 x 8 A lists + 8 A records out + 8 A records in + 8 + 16 per survivor) the seed stage moves %.1f GB" % ((32.0 * A0 + 24.0 * W0 + 32.0 * n_qm) / 1e9),
                 "instruction_ceiling": "60 % of 8 TB/s on 5.6 B per base is 0.86 Tbases/s: 256 CUs x 4 SIMDs at 2.4 GHz issue ~2.5 T wave64 instructions/s = 157 T lane-ops/s, "
                                        "i.e. ~180 lane-instructions per base at the very most, before any stall.  Per k-mer the sketch needs the 2-bit window update and its "
                                        "reverse complement (~10), the 64-bit invertible hash (7 rounds of shift/add/xor on two 32-bit halves: ~30 for k <= 16), the "
                                        "palindrome test, the window minimum with minimap2's tie rules (~15 per position at w = 5 after the round-3 unrolling) and the "
                                        "mask write: ~140 instructions per base measured in round 3 -- the decision pass is issue-bound near that ceiling, not HBM-bound; "
                                        "the list from the mask (k_sketch_emit_mask) is the bandwidth part and runs at 1.4-1.8 TB/s"}
    roof = None
    if st:
        dom = st
        steps_t = max(args.steps, 1)
        if dom.get("per_step"):
            # a group of kernels: bytes and device time per step (the group's launches of one step, one after the other on their lanes)
            per_launch_ms = dom["total_ms"] / steps_t
            per_launch_bytes = dom["algo_bytes"]
        else:
            per_launch_ms = dom["total_ms"] / max(dom["launches"], 1)
            per_launch_bytes = dom["algo_bytes"] / max(dom["launches"], 1)
        ach = per_launch_bytes / (per_launch_ms / 1e3) / 1e9 if per_launch_ms > 0 else 0.0
        roof = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                "what_is_timed": ("the group's kernels: %s" % ", ".join(dom["members"])) if len(dom["members"]) > 1 else "one kernel",
                "byte_model": dom["byte_model"],
                "launches": dom["launches"], ("device_ms_per_step" if dom.get("per_step") else "avg_launch_ms"): round(per_launch_ms, 4),
                ("algo_bytes_per_step" if dom.get("per_step") else "algo_bytes_per_launch"): int(per_launch_bytes),
                "selection": "the entry with the largest device time in one step among every kernel / kernel group with a byte model -- nothing excluded",
                "timing": "HIP events around every launch of these kernels inside the timed steps, on the launching streams (lanes run side by side: "
                          "a group's device time can exceed its share of the wall clock); kernel_ms_one_step: the same for every kernel in one extra untimed step",
                "kernel_ms_one_step": {s["name"]: round(s["total_ms"], 3) for s in sorted(all_st, key=lambda s: -s["total_ms"])},
                "groups_ms_one_step": {g["name"]: {"ms": round(g["total_ms"], 2), "algo_GB": round(g["algo_bytes"] / 1e9, 2),
                                                   "GB_per_s": round(g["algo_bytes"] / 1e9 / (g["total_ms"] / 1e3), 1) if g["total_ms"] > 0 else None}
                                       for g in sorted(groups.values(), key=lambda g: -g["total_ms"])[:12]}}
        # The kernel with the most device time may be a latency-bound one (serial token walks, checkpoint solvers: a few bytes per
        # anchor) or one whose launches wait for room beside the other lanes' kernels; the same figures for every kernel whose
        # algorithmic bytes are known, from the one untimed step with events around every launch (4 lanes: durations include
        # the time a launch shares the device with up to seven other kernels)
        roof["per_kernel_one_step"] = {s["name"]: {"ms": round(s["total_ms"], 2), "launches": s["launches"], "algo_GB": round(s["algo_bytes"] / 1e9, 2),
                                                   "GB_per_s": round(s["algo_bytes"] / 1e9 / (s["total_ms"] / 1e3), 1) if s["total_ms"] > 0 else None}
                                       for s in sorted(all_st, key=lambda s: -s["total_ms"]) if s["algo_bytes"]}
    if rank == 0:
        # HBM traffic of the dominant kernel from the PMC passes of this workload, if they were collected
        # (tools/gpu_evidence.sh -> profiles/*pmc_traffic.json; separate rocprofv3 --pmc runs, never inside a timed run)
        if roof and full_config:
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*pmc_traffic*.json")), reverse=True):
                try:
                    js = json.load(open(fn))
                    kk = js["kernels"]
                except Exception:
                    continue
                if args.config not in js.get("workload", ""):
                    continue
                # stage names -> kernel names where they differ; a group: every kernel of its members
                alias = {"k_rs_hist": ["k_rs_hist<false>"], "k_rs_hist<first>": ["k_rs_hist<true>"], "k_ck_prefix": ["k_ck_tilehist", "k_ck_tilescan"], "k_ck_solve": ["k_ck_phases", "k_ck_solve"],
                         "k_sort_two": ["k_two_tiles", "k_sort_two_tiled", "k_sort_two_scan"], "k_sort_walk_solo_ck": ["k_sort_walk_solo"], "k_sort_walk_reg<1>ck": ["k_sort_walk_reg<1>"]}
                names = []
                for m in (st["members"] if st else [roof["kernel"]]):
                    names += alias.get(m, [m])
                hit = {}
                for roof_k in names:
                    want = roof_k.replace(">", "")                       # "k_ps_finish<8192" matches "k_ps_finish<8192, 1024, 10, unsigned int>"
                    for k, v in kk.items():
                        if k == roof_k or k.startswith(want + ",") or k.startswith(want + ">") or (("<" not in want) and k.split("<")[0] == want):
                            hit[k] = v
                if hit:
                    nd = sum(v["dispatches"] for v in hit.values())          # (a kernel with several shapes: all its launches)
                    tot = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in hit.values())
                    steps_pmc = max(int(js.get("steps", 1)), 1)
                    if st and st.get("per_step"):
                        roof["traffic"] = round(tot / steps_pmc / 1e9, 3)
                        roof["traffic_unit"] = ("GB per step over the group's %d launches (PMC: (2*FETCH_SIZE+WRITE_SIZE)*1024, from the committed %s: separate rocprofv3 --pmc passes "
                                                "of one step of this workload, commit %s -- not from this run)" % (nd, os.path.basename(fn), js.get("commit", "not recorded")))
                    else:
                        roof["traffic"] = round(tot / max(nd, 1) / 1e9, 3)
                        roof["traffic_unit"] = ("GB per launch (PMC: (2*FETCH_SIZE+WRITE_SIZE)*1024 over %d launches, from the committed %s: separate rocprofv3 --pmc passes "
                                                "of this workload, commit %s -- not from this run)" % (nd, os.path.basename(fn), js.get("commit", "not recorded")))
                    break
        line = {
            "metric": "Mbases/sec all-vs-all overlap coverage (sampleqc hot path)", "value": round(value, 3), "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": (CONFIG_LABEL[args.config] % (cfg.n_reads, cfg.mean_len // 1000, cfg.depth)) + ", %d subsample queries" % len(Q)
                                   + (" [-I overridden: %s]" % args.index_size if args.index_size else ""),
                       "index_parts": [int(lens[lo:hi].sum()) for lo, hi in parts],
                       "target_bases": int(total_bases), "query_bases": int(Q.n_bases), "anchors_per_step": int(n_anchors),
                       "anchors_written_per_step": int(written[0]) if world == 1 else None,      # seed hits whose (strand, target) can reach a chain: the others are never written
                       "klib_order": {k: v for k, v in (mstats[0] or {}).items() if k != "last_written"},   # runs / queries / anchors that needed klib's own order of equal-x anchors (second pass)
                       "clock": "H2D of the packed reads (pinned) -> sketch -> index -> seed -> sort -> chain -> coverage -> rows D2H, all parts (SURVEY 8d), "
                                + ("part i+1's upload + sketch + index under part i's mapping; " if world == 1 and len(parts) > 1 else "") +
                                "before the clock: the synthetic generator (= FASTQ parse) and the host-side 2-bit packing (host_pack_s) -- value_incl_host_pack adds the latter, unoverlapped",
                       "parallelism": "single GPU" if world == 1 else
                                      ("%d GPUs (torch.distributed world size %d, backend %s): index parts across the GPUs in rounds of %d, every rank maps all queries against its part; mid_occ broadcast from part 0, COVT cap and "
                                       "avg_k replayed in part order, sums all-reduced, intervals all-gathered (RCCL)" % (world, dist.get_world_size(), dist.get_backend(), world)) if split == "parts" else
                                      ("%d GPUs (torch.distributed world size %d, backend %s): queries sharded 1/N, index replicated (each rank uploads 1/N of a part's packed reads, the packed reads are "
                                       "all-gathered over RCCL -- 0.375 B per base --, every rank sketches the part and builds the identical index; "
                                       "the front of part i + 1 under the mapping of part i), rows gathered on rank 0" % (world, dist.get_world_size(), dist.get_backend())),
                       "scaling_model_s": None if world == 1 else {"queries_sharded": round(tq, 3), "parts_across_gpus": round(tp, 3), "single_gpu": round(multigpu.QueryShardRunner.scaling_model(1, part_bases), 3),
                                                                   "note": "predicted seconds per job from the single-GPU stage rates of round 4 (longqc_amd/multigpu.py); the faster split runs"}},
            "roofline": roof, "north_star": north,
            "host_pack_s": round(t_pack, 3), "synth_gen_s": round(t_gen, 2),
            "value_incl_host_pack": round(total_bases / (ms_per_step / 1e3 + t_pack) / 1e6, 3),
            "hbm_resident_value": round(resident, 3) if resident else None,
            "anchors_per_s": round(n_anchors / (ms_per_step / 1e3), 1),
        }
        if full_config:
            line["golden_rows"] = golden_check(args.config, table)
        line["table_properties"] = table_properties(table, Q.names, [int(x.shape[0]) for x in Q.seqs])
        if world == 1 and not args.no_cpu_baseline:
            n_t = args.cpu_sample or {"cfg3": 25000, "cfg2": 35000, "cfg4": 15000, "cfg4s": 15000, "cfg5": 5000, "cfg5s": 5000}.get(args.config, 25000)
            line["cpu_baseline"] = cpu_baseline(args.config, F, Q, n_t, len(Q) if n_t >= len(F) else max(50, n_t // 100))
        if one_dev:
            line["note"] = "LQCOV_BENCH_ONE_DEVICE test mode: all ranks share cuda:0 over gloo; not a scaling measurement"
    eng.close()
    if rank == 0 and world == 1 and have_cuda and line.get("roofline"):
        cc = copy_ceiling(local)                                                  # (the engine is closed: the device is idle)
        if cc:
            line["roofline"]["measured_copy_ceiling"] = {"GB/s": cc, "frac_of_it": round(line["roofline"]["achieved"] / cc, 4),
                                                         "what": "bytes read + written per second of a 2-GiB device-to-device copy on this box, best of 5"}
    if rank == 0 and world == 1 and not args.no_end_to_end:
        # The drop-in call as LongQC issues it (INTEGRATION.md level 1 / 2): files in, table out -- FASTA/Q parse (targets: plain
        # FASTA on tmpfs, parsed from the mapping by the host's cores; queries: FASTQ), 2-bit packing, upload, every part, rows,
        # formatting, with the parts in run_files' own pipeline.  SURVEY 8d: "once including it, to compare like-for-like with
        # the oracle's Real time" (minimap2-coverage.c:732).  The table must equal the one of the timed steps.
        del P
        import shutil
        base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2.5 * total_bases + (1 << 30) else None
        with tempfile.TemporaryDirectory(dir=base) as d:
            tf, qf, of = os.path.join(d, "all.fa"), os.path.join(d, "sub.fq"), os.path.join(d, "out.tsv")
            t0 = time.time()
            synth.write_flat_fasta(tf, F)
            synth.write_fastq(qf, Q)
            t_write = time.time() - t0
            # as LongQC would: the executable as a subprocess (process start, HIP start-up and code load included), same argv as
            # longQC.py:440-445.  The device has to be idle first: memory this process has just freed is still being reclaimed by
            # the driver for a few seconds, and an allocation that needs it waits (measured: 5 s for the first 20 GB buffer).
            exe = os.path.join(ROOT, "longqc_amd", "minimap2-coverage-mi355x")
            best = None; call_s = None; log_tail = []
            for _ in range(2):
                if have_cuda:
                    torch.cuda.empty_cache()
                time.sleep(8.0 if have_cuda else 0.0)
                t0 = time.time()
                if os.path.exists(exe) and have_cuda:
                    env = dict(os.environ); env["LQCOV_DEVICE"] = str(local)
                    r = subprocess.run([exe] + list(PRESET[args.config][1]) + ["-t", "8", tf, qf], stdout=open(of, "w"), stderr=subprocess.PIPE, env=env)
                    dt2 = time.time() - t0
                    elog = r.stderr.decode(errors="replace").splitlines()
                    if r.returncode != 0:
                        elog.append("exit status %d" % r.returncode)
                else:                                                             # (the CPU dry run of this script: in process, the test emulator)
                    e2 = api.Engine(p, device=local)
                    e2.run_files(tf, qf, out=of, err=os.path.join(d, "err.log"))
                    dt2 = time.time() - t0
                    e2.close()
                    elog = open(os.path.join(d, "err.log")).read().splitlines()
                if best is None or dt2 < best:
                    best = dt2; log_tail = [l for l in elog if l.startswith("[lqcov]")][-8:]
                    cs = [l for l in elog if "the whole call" in l]
                    call_s = float(cs[-1].split("the whole call")[1].split()[0]) if cs else None
            same = open(of).read() == table
            gz = None
            if args.gz_end_to_end and os.path.exists(exe) and have_cuda:
                # the same call with the targets as ONE gzip stream (LongQC's inputs are usually .fastq.gz): the streaming reader inflates on one
                # host thread (fastx.hpp, zlib), so the call is bound by it
                tg = tf + ".gz"
                t0 = time.time(); subprocess.run("pigz -1 -k -c %s > %s 2>/dev/null || gzip -1 -c %s > %s" % (tf, tg, tf, tg), shell=True, check=True); t_gz = time.time() - t0
                time.sleep(8.0)
                t0 = time.time()
                r = subprocess.run([exe] + list(PRESET[args.config][1]) + ["-t", "8", tg, qf], stdout=open(of + ".gz.tsv", "w"), stderr=subprocess.PIPE, env=env)
                dtg = time.time() - t0
                gz = {"value": round(total_bases / dtg / 1e6, 3), "unit": "Mbases/s", "seconds": round(dtg, 3), "table_identical_to_timed_steps": r.returncode == 0 and open(of + ".gz.tsv").read() == table,
                      "files": "targets: the same FASTA as one gzip -1 stream (%.2f GB, made in %.0f s)" % (os.path.getsize(tg) / 1e9, t_gz),
                      "log_tail": [l for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("[lqcov]")][-6:]}
                os.remove(tg)
            line["end_to_end"] = {"value": round(total_bases / best / 1e6, 3), "unit": "Mbases/s", "seconds": round(best, 3), "seconds_inside_the_call": call_s, "table_identical_to_timed_steps": same,
                                  "files": "targets: plain FASTA (%.1f GB) on %s, queries: FASTQ; best of 2 runs, device idle for 8 s before each" % (os.path.getsize(tf) / 1e9, base or "the temp dir"),
                                  "what": "the executable minimap2-coverage-mi355x as a subprocess, LongQC's argv: process start + HIP start-up + parse (mapped file, %d host threads) + 2-bit pack + H2D + sketch + index + map "
                                          "of every part in run_files' pipeline + rows + table text; wall clock around the process" % min(64, os.cpu_count() or 1),
                                  "log_tail": log_tail, "file_write_s": round(t_write, 1)}
            if gz:
                line["end_to_end"]["gz"] = gz
            # the reference on the same files, all reads and all queries, when the host has the cores for it (measured, not quoted)
            cores = os.cpu_count() or 1
            from tests import oracle_bind
            if cores >= 64 and oracle_bind.have_ref() and not args.no_cpu_baseline and line.get("cpu_baseline"):
                t0 = time.time()
                try:
                    r = subprocess.run([oracle_bind.REF_BIN] + list(PRESET[args.config][1]) + ["-t", str(cores), tf, qf], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
                    dtr = time.time() - t0
                    if r.returncode == 0:
                        # the whole job is the baseline; the bounded sample (faster per base: a smaller index, fewer hits per query) stays beside it
                        sample = line["cpu_baseline"]
                        line["cpu_baseline"] = {"value": round(total_bases / dtr / 1e6, 3), "unit": "Mbases/s", "seconds": round(dtr, 1), "cores": cores, "kind": "reference",
                                                "sample": "ALL target reads and ALL queries of the workload, same files as end_to_end, measured in this run",
                                                "table_identical_to_gpu": r.stdout.decode() == table, "bounded_sample": sample}
                        line["cpu_baseline"]["full_job"] = {k: line["cpu_baseline"][k] for k in ("value", "unit", "seconds", "cores", "kind", "sample", "table_identical_to_gpu")}
                except subprocess.TimeoutExpired:
                    pass
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
