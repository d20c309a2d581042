#!/usr/bin/env python3
"""bench.py -- throughput of the sampleqc hot path (all reads -> index, subsample -> coverage table)
on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 50k synthetic ONT reads ~15 kb, 15x, ont-ligation preset
(-Y -l 0 -q 160 -k 12 -w 5 -p 160), query set = LongQC's seed-7 subsample of 5000 reads.  One "step" =
one pass of the hot path over the resident reads: reset accumulators -> sketch + index the targets
-> seed / klib-order sort / chain / coverage for every query -> rows (D2H).  Reads are 2-bit packed
in HBM before the timed region starts.  At N > 1 every rank owns one index part of that size (the
reference's own -I partitioning of a N-times larger read set, DESIGN.md section "multi-GPU"), the
per-part accumulators are combined over RCCL inside the timed region, and value = all ranks'
target bases / max-over-ranks time (weak scaling).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(T, Q, n_t, n_q):
    """The reference's own minimap2-coverage (oracle/_ref, kind "reference") or the C restatement
    (kind "port") on a bounded sample of the same workload, timed on this host's cores."""
    from longqc_amd import synth
    from tests import oracle_bind
    cores = os.cpu_count() or 1
    sub_t = T.subset(range(min(n_t, len(T))))
    sub_q = Q.subset(range(min(n_q, len(Q))))
    with tempfile.TemporaryDirectory() as d:
        tf, qf = os.path.join(d, "t.fq"), os.path.join(d, "q.fq")
        synth.write_fastq(tf, sub_t); synth.write_fastq(qf, sub_q)
        argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160"]
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
            "sample": "first %d target reads (%.1f Mbases) + first %d subsample reads as queries, same argv, "
                      "FASTQ parse included; index build is ~serial in the reference (3 fixed threads)" % (len(sub_t), sub_t.n_bases / 1e6, len(sub_q))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="override reads per GPU (default: the config's 50000)")
    ap.add_argument("--nsample", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=35000, help="target reads in the CPU baseline sample (~20-30 s of reference time)")
    args = ap.parse_args()

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
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("gloo" if one_dev else "nccl", rank=rank, world_size=world)

    import dataclasses
    from longqc_amd import api, synth, multigpu
    cfg = synth.CONFIGS["cfg2"]
    if args.reads:
        cfg = dataclasses.replace(cfg, n_reads=args.reads)
    if args.nsample:
        cfg = dataclasses.replace(cfg, nsample=args.nsample)
    per_gpu = cfg.n_reads
    t0 = time.time()
    genome = synth.make_genome(cfg)
    T = synth.make_reads(cfg, genome, n_reads=per_gpu, read_offset=rank * per_gpu)          # this rank's index part
    qidx = synth.reservoir_subsample(per_gpu, cfg.nsample)                                   # subsample of part 0
    Q = T.subset(qidx) if rank == 0 else synth.make_reads(cfg, genome, indices=qidx)
    t_gen = time.time() - t0

    p = api.default_params(no_self=1, min_ovlp=0, min_score_med=160, min_score_good=160, k=12, w=5)
    eng = api.Engine(p, device=local)
    t0 = time.time()
    eng.set_queries(Q.names, Q.seqs, Q.quals)
    pt = eng.part_begin()
    # upload in 50-Mbase mini-batches like mm_idx_gen (index.c:246); packed 2-bit in HBM afterwards
    i = 0
    while i < len(T):
        j, b = i, 0
        while j < len(T) and b < 50000000:
            b += int(T.seqs[j].shape[0]); j += 1
        eng.part_add_targets(pt, T.names[i:j], T.seqs[i:j])
        i = j
    eng.sync()
    t_upload = time.time() - t0
    my_bases = T.n_bases
    runner = multigpu.PartRunner(eng, world=world, rank=rank, device=torch.device("cuda", local), query_lengths=[int(s.shape[0]) for s in Q.seqs]) if world > 1 else None

    def step():
        if runner is None:
            eng.reset()
            eng.part_build(pt)
            eng.part_map(pt)
        else:
            runner.begin()
            eng.part_build(pt)
            runner.map_and_combine(pt, part_index=rank)
        eng.finish()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # one untimed step with HIP events around every launch (level 2: events only, nothing waits for them): the per-kernel
    # table and the name of the dominant kernel; the timed steps then carry events around that kernel only
    eng.set_profiling(2)
    step()
    all_st = [s for s in eng.stage_times() if s["name"].startswith("k_") or s["name"].startswith("index_")]
    dom_name = max(all_st, key=lambda s: s["total_ms"])["name"] if all_st else None
    eng.set_profiling(0)
    eng.set_profiling(2, only=dom_name)
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    eng.sync()
    barrier()
    dt = time.time() - t0
    tt = torch.tensor([dt, float(my_bases)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, total_bases = float(tmax[0]), float(tsum[1])
    else:
        total_bases = float(my_bases)
    ms_per_step = dt / max(args.steps, 1) * 1e3
    value = total_bases / (ms_per_step / 1e3) / 1e6

    st = [s for s in eng.stage_times() if s["name"].startswith("k_") or s["name"].startswith("index_")]
    eng.set_profiling(0)
    n_anchors = eng.last_n_anchors
    roof = None
    if st:
        dom = max(st, key=lambda s: s["total_ms"])
        per_launch_ms = dom["total_ms"] / max(dom["launches"], 1)
        per_launch_bytes = dom["algo_bytes"] / max(dom["launches"], 1)
        ach = per_launch_bytes / (per_launch_ms / 1e3) / 1e9 if per_launch_ms > 0 else 0.0
        roof = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                "launches": dom["launches"], "avg_launch_ms": round(per_launch_ms, 4), "algo_bytes_per_launch": int(per_launch_bytes),
                "timing": "HIP events around every launch of this kernel inside the timed steps, on the launching stream; "
                          "kernel_ms_one_step: the same for every kernel in one extra untimed step (mapping lanes overlap, "
                          "so those add up to more than ms_per_step)",
                "kernel_ms_one_step": {s["name"]: round(s["total_ms"], 3) for s in sorted(all_st, key=lambda s: -s["total_ms"])}}
    if world > 1:
        barrier()
    if rank == 0:
        # HBM traffic of the dominant kernel from the PMC passes of this workload, if they were collected
        # (tools/gpu_round.sh -> profiles/*pmc_traffic.json; separate rocprofv3 --pmc runs, never inside a timed run)
        if roof and not args.reads and not args.nsample:
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
                try:
                    kk = json.load(open(fn))["kernels"]
                except Exception:
                    continue
                hit = [v for k, v in kk.items() if k.split("<")[0] == roof["kernel"].split("<")[0]]
                if hit:
                    roof["traffic"] = round(hit[0]["hbm_bytes_per_launch"] / 1e9, 3)
                    roof["traffic_unit"] = "GB per launch (PMC: (2*FETCH_SIZE+WRITE_SIZE)*1024, %s)" % os.path.basename(fn)
                    break
        line = {
            "metric": "Mbases/sec all-vs-all overlap coverage (sampleqc hot path)", "value": round(value, 3), "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%d synthetic ONT reads ~%d kb %gx per GPU (BASELINE configs[1]), ont-ligation preset "
                                   "(-Y -l 0 -q 160 -k 12 -w 5 -p 160), %d subsample queries" % (per_gpu, cfg.mean_len // 1000, cfg.depth, len(Q)),
                       "target_bases_per_gpu": int(my_bases), "query_bases": int(Q.n_bases), "anchors_per_step": int(n_anchors),
                       "parallelism": "1 index part per GPU, accumulators combined over RCCL" if world > 1 else "single GPU"},
            "roofline": roof,
            "upload_pack_s": round(t_upload, 3), "synth_gen_s": round(t_gen, 2),
            "pcie_inclusive_value": round(total_bases / (ms_per_step / 1e3 + t_upload) / 1e6, 3),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(T, Q, args.cpu_sample, max(50, args.cpu_sample // 10))
        if one_dev:
            line["note"] = "LQCOV_BENCH_ONE_DEVICE test mode: all ranks share cuda:0 over gloo; not a scaling measurement"
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
