#!/usr/bin/env python3
"""GPU sweep: one synthetic slice of configs[1], several environment-knob variants of the engine.
usage: gpu_sweep.py N_READS,N_QUERIES [--top K] "K=V K2=V2" "K=V" ...   ("-" = defaults)
Every variant must produce the same table as the first one (printed as same/DIFF)."""
import dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longqc_amd import api, synth

def main():
    args = sys.argv[1:]
    n_reads, nsample = (int(x) for x in args.pop(0).split(","))
    top = 10
    if args and args[0] == "--top":
        args.pop(0); top = int(args.pop(0))
    variants = args or ["-"]
    cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=n_reads, nsample=nsample)
    T, Q = synth.make_dataset(cfg)
    first = None
    for v in variants:
        env = dict(kv.split("=", 1) for kv in v.split()) if v != "-" else {}
        for k, val in env.items(): os.environ[k] = val
        eng = api.Engine(api.default_params(no_self=1, min_ovlp=0, min_score_med=160, min_score_good=160), 0)
        eng.set_queries(Q.names, Q.seqs, Q.quals)
        pt = eng.part_begin()
        for i in range(0, len(T), 3000):
            eng.part_add_targets(pt, T.names[i:i + 3000], T.seqs[i:i + 3000])
        times = []
        for it in range(3):
            t0 = time.time(); eng.reset(); eng.part_build(pt); eng.part_map(pt); eng.finish(); times.append(time.time() - t0)
        print("== %-60s step s: %s  -> %.1f Mbases/s" % (v, " ".join("%.3f" % t for t in times), T.n_bases / min(times[1:]) / 1e6), flush=True)
        if top:
            eng.set_profiling(True)
            eng.reset(); eng.part_build(pt); eng.part_map(pt); eng.finish()
            st = eng.stage_times()
            eng.set_profiling(False)
            print("   " + "  ".join("%s %.1f" % (s["name"].replace("k_sort_", "s:").replace("k_", ""), s["total_ms"]) for s in sorted(st, key=lambda s: -s["total_ms"])[:top]), flush=True)
        table = eng.table_text()
        eng.close()
        if first is None: first = table
        print("   table", "same" if table == first else "DIFF", flush=True)
        for k in env: del os.environ[k]

if __name__ == "__main__":
    main()
