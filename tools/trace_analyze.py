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


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "lanes"):
    main()


def lanes_report(fn):
    """per mapping lane (a main stream = the one that runs k_seed_emit, the stream of its walkers, the stream of its parallel
    sort): how the time between the lane's first and last kernel of the step splits into: main stream busy, main stream idle
    while the lane's walkers run (the level waits for them), idle while only its parallel-sort stream runs, idle with nothing
    of the lane on the device (the host thread is between a read-back and the next launch)"""
    rows = []
    for line in open(fn):
        f = line.rstrip("\n").split(",")
        rows.append((f[0], f[1], int(f[2]), int(f[3])))
    rows.sort(key=lambda r: r[2])
    starts = [r[2] for r in rows if "k_sketch_dp_mask" in r[0]]
    t_begin = starts[-2] if len(starts) >= 2 else rows[0][2]
    step = [r for r in rows if r[2] >= t_begin]
    by = collections.defaultdict(list)
    for r in step:
        by[r[1]].append(r)
    mains = [s for s, v in by.items() if any("k_seed_emit" in r[0] for r in v)]
    walkers = [s for s, v in by.items() if any("k_sort_walk" in r[0] or "k_ck_" in r[0] for r in v)]
    others = [s for s in by if s not in mains and s not in walkers]
    print("main streams %s, walker streams %s, other streams %s" % (sorted(mains), sorted(walkers), sorted(others)))

    def union(iv):
        iv = sorted(iv); out = []
        for a, b in iv:
            if out and a <= out[-1][1]:
                out[-1][1] = max(out[-1][1], b)
            else:
                out.append([a, b])
        return out

    def length(iv):
        return sum(b - a for a, b in iv)

    def intersect(x, y):
        i = j = 0; out = []
        while i < len(x) and j < len(y):
            a = max(x[i][0], y[j][0]); b = min(x[i][1], y[j][1])
            if a < b:
                out.append([a, b])
            if x[i][1] < y[j][1]:
                i += 1
            else:
                j += 1
        return out

    allw = union([(r[2], r[3]) for s in walkers for r in by[s]])
    for m in sorted(mains):
        v = by[m]
        a0, a1 = min(r[2] for r in v), max(r[3] for r in v)
        busy = union([(r[2], r[3]) for r in v])
        idle = []
        last = a0
        for a, b in busy:
            if a > last:
                idle.append([last, a])
            last = b
        print("lane on stream %s: span %.0f ms, busy %.0f ms, idle %.0f ms (of which some walker kernel runs: %.0f ms), %d kernels, longest gaps: %s" % (
            m, (a1 - a0) / 1e6, length(busy) / 1e6, length(idle) / 1e6, length(intersect(idle, allw)) / 1e6, len(v),
            ", ".join("%.1f" % (g / 1e6) for g in sorted((b - a for a, b in idle), reverse=True)[:6])))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "lanes":
    lanes_report(sys.argv[1])
