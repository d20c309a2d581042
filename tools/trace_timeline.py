#!/usr/bin/env python3
"""When did what run during the timed step of a traced bench run?  From tools/gpu_trace.sh's trace_small.csv (kernel name, stream,
start ns, end ns, grid): the build stream's kernels in order (upload, sketch, index, seed plan of every part) and, per mapping lane
(a main stream and the three streams after it), every batch from its first kernel to its last with the start of its second pass
and its largest kernels.  The bench traces warm-up, timed and profiling steps one after the other; `--step k` picks the k-th part-1
sketch launch from the end as the start (default 2: the last timed step; the very last step runs with the lanes serialized for the
per-kernel times).
    python tools/trace_timeline.py gpurun_out/trace_small.csv [--step 2]"""
import collections
import sys


def main():
    fn = sys.argv[1]
    k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 2
    rows = []
    for line in open(fn):
        f = line.rstrip("\n").split(",")
        rows.append((f[0].replace("void ", ""), f[1], int(f[2]), int(f[3]), f[4]))
    rows.sort(key=lambda r: r[2])
    dp = [r for r in rows if r[0].startswith("k_sketch_dp_mask")]
    big = max(int(r[4]) for r in dp)
    firsts = [r[2] for r in dp if int(r[4]) == big]                       # part 1 of every step (the largest grid)
    t0 = firsts[-k] - 30_000_000                                          # (the part's upload comes before its sketch)
    t1 = firsts[-k + 1] - 30_000_000 if k > 1 else rows[-1][3]
    step = [r for r in rows if t0 <= r[2] < t1]
    streams = sorted({r[1] for r in step}, key=int)
    build = max(streams, key=lambda s: sum(r[3] - r[2] for r in step if r[1] == s and ("sketch" in r[0] or "rocprim" in r[0] or "k_seed_count" in r[0])))
    print("step of %.0f ms (%d kernels); build stream %s:" % ((max(r[3] for r in step) - t0) / 1e6, len(step), build))
    for r in step:
        if r[1] == build and r[3] - r[2] > 4_000_000:
            print("  %6.0f ms  +%5.0f  %s" % ((r[2] - t0) / 1e6, (r[3] - r[2]) / 1e6, r[0][:44]))
    mains = sorted({r[1] for r in step if r[0].startswith("k_seed_emit_s")}, key=int)
    for m in mains:
        group = [str(int(m) + d) for d in range(4)]
        ev = [r for r in step if r[1] in group]
        batches = []
        for r in ev:
            if r[0].startswith("k_seed_emit_s"):
                batches.append([])
            if batches:
                batches[-1].append(r)
        for b in batches:
            s, e = b[0][2], max(x[3] for x in b)
            acc = collections.Counter()
            for x in b:
                acc[x[0][:18]] += (x[3] - x[2]) / 1e6
            p2 = [x[2] for x in b if x[0] == "k_seed_emit"]
            print("lane (streams %s..%s): batch %4.0f -> %4.0f ms (%3.0f ms), second pass from %s; %s" % (
                group[0], group[-1], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, "%.0f" % ((p2[0] - t0) / 1e6) if p2 else "-",
                ", ".join("%s %.0f" % kv for kv in acc.most_common(5))))


if __name__ == "__main__":
    main()
