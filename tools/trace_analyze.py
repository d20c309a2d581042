#!/usr/bin/env python3
"""What the GPU did during one step, from tools/gpu_trace.sh's trace_small.csv (name, stream, start ns, end ns, grid):
the step = the last stretch of the trace that begins with the sketch kernels; per kernel: time summed, time during which it
was the only kernel running, average number of kernels running beside it; overall: wall time, time with 0 / 1 / 2 / ... kernels
running.  Usage: python tools/trace_analyze.py gpurun_out/trace_small.csv [n_steps_in_trace]"""
import collections
import sys


def main():
    fn = sys.argv[1]
    rows = []
    for line in open(fn):
        f = line.rstrip("\n").split(",")
        rows.append((f[0], f[1], int(f[2]), int(f[3])))
    rows.sort(key=lambda r: r[2])
    # the last step: from the last launch of the first sketch kernel of a job's first part
    starts = [r[2] for r in rows if r[0].startswith("void k_sketch_dp_mask") or r[0].startswith("k_sketch_dp_mask")]
    if not starts:
        starts = [rows[0][2]]
    # two parts per step at configs[2]: the step starts at the second-to-last sketch launch
    t_begin = starts[-2] if len(starts) >= 2 else starts[-1]
    step = [r for r in rows if r[2] >= t_begin]
    t0 = min(r[2] for r in step); t1 = max(r[3] for r in step)
    print("step: %d kernels, %.1f ms of wall time" % (len(step), (t1 - t0) / 1e6))
    ev = []
    for i, r in enumerate(step):
        ev.append((r[2], 1, i)); ev.append((r[3], -1, i))
    ev.sort()
    running = set()
    last = t0
    conc_time = collections.Counter()
    alone = collections.Counter(); weighted = collections.Counter(); total = collections.Counter(); n = collections.Counter()
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            c = len(running)
            conc_time[min(c, 8)] += dt
            for j in running:
                nm = step[j][0]
                weighted[nm] += dt * c
                if c == 1:
                    alone[nm] += dt
        last = t
        if d == 1:
            running.add(i)
        else:
            running.discard(i)
    for r in step:
        total[r[0]] += r[3] - r[2]; n[r[0]] += 1
    print("kernels running at once: " + "  ".join("%d: %.0f ms" % (c, conc_time[c] / 1e6) for c in sorted(conc_time)))
    print("%-44s %7s %9s %9s %8s" % ("kernel", "calls", "total ms", "alone ms", "avg conc"))
    for nm, tt in sorted(total.items(), key=lambda kv: -kv[1])[:40]:
        print("%-44s %7d %9.1f %9.1f %8.2f" % (nm[:44], n[nm], tt / 1e6, alone[nm] / 1e6, weighted[nm] / max(tt, 1)))
    # streams: busy time and the longest gaps
    by_stream = collections.defaultdict(list)
    for r in step:
        by_stream[r[1]].append(r)
    print("streams: " + "  ".join("%s: %.0f ms busy in %d kernels" % (s, sum(r[3] - r[2] for r in v) / 1e6, len(v)) for s, v in sorted(by_stream.items())))


if __name__ == "__main__":
    main()
