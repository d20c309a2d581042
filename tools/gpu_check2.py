#!/usr/bin/env python3
"""GPU check: parity + per-stage device times on growing slices of configs[1]. Writes gpurun_out/check2.log"""
import dataclasses, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from longqc_amd import api, synth
from tests import oracle_bind

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "check2.log"), "a")
def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); LOG.write(s + "\n"); LOG.flush()

def run(n_reads, nsample, check_ref, tmp="/tmp"):
    cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=n_reads, nsample=nsample)
    t0 = time.time(); T, Q = synth.make_dataset(cfg); tg = time.time() - t0
    p = api.default_params(no_self=1, min_ovlp=0, min_score_med=160, min_score_good=160)
    eng = api.Engine(p, 0)
    t0 = time.time()
    eng.set_queries(Q.names, Q.seqs, Q.quals)
    pt = eng.part_begin()
    step = 3000
    for i in range(0, len(T), step):
        eng.part_add_targets(pt, T.names[i:i + step], T.seqs[i:i + step])
    tu = time.time() - t0
    times = []
    for it in range(2):
        t0 = time.time(); eng.reset(); eng.part_build(pt); t1 = time.time(); eng.part_map(pt); t2 = time.time(); eng.finish(); t3 = time.time()
        times.append((t1 - t0, t2 - t1, t3 - t2))
    log("reads", n_reads, "bases", T.n_bases, "queries", len(Q), "anchors", eng.last_n_anchors, "gen_s %.1f upload_s %.2f" % (tg, tu),
        "build/map/finish s:", ["%.3f/%.3f/%.3f" % t for t in times], "Mbases/s %.1f" % (T.n_bases / sum(times[1]) / 1e6))
    eng.set_profiling(True)
    eng.reset(); eng.part_build(pt); eng.part_map(pt); eng.finish()
    st = eng.stage_times()
    eng.set_profiling(False)
    for s in sorted(st, key=lambda s: -s["total_ms"])[:14]:
        gbs = s["algo_bytes"] / max(s["total_ms"], 1e-9) / 1e6
        log("   %-22s %10.3f ms  x%-5d  %8.1f GB/s algo" % (s["name"], s["total_ms"], s["launches"], gbs))
    table = eng.table_text()
    eng.close()
    if check_ref:
        tf, qf = os.path.join(tmp, "t.fq"), os.path.join(tmp, "q.fq")
        synth.write_fastq(tf, T); synth.write_fastq(qf, Q)
        t0 = time.time()
        want = oracle_bind.ref_table(["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-t", "64", tf, qf])
        log("   reference -t 64: %.2f s -> %.1f Mbases/s ; parity %s" % (time.time() - t0, T.n_bases / (time.time() - t0) / 1e6, "OK" if want == table else "FAIL"))
        os.remove(tf); os.remove(qf)

if __name__ == "__main__":
    sizes = [(1000, 300, True), (5000, 1000, True), (20000, 2000, False)]
    if len(sys.argv) > 1:
        sizes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
        sizes = [(a, b, bool(c)) for a, b, c in sizes]
    for n, q, c in sizes:
        run(n, q, c)
