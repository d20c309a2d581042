#!/usr/bin/env python3
"""Every kernel of one mapping lane in the last timed step of a traced bench run (tools/gpu_trace.sh's trace_small.csv), in start order:
    python tools/trace_lane.py gpurun_out/trace_small.csv [first stream of the lane, default 4] [min ms, default 0.3] [--step k]"""
import sys


def main():
    fn = sys.argv[1]
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 4
    mn = float(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else 0.3
    k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 2
    rows = []
    for line in open(fn):
        f = line.rstrip("\n").split(",")
        rows.append((f[0].replace("void ", ""), int(f[1]), int(f[2]), int(f[3]), int(f[4])))
    rows.sort(key=lambda r: r[2])
    dp = [r for r in rows if r[0].startswith("k_sketch_dp")]
    big = max(r[4] for r in dp)
    firsts = [r[2] for r in dp if r[4] == big]
    t0 = firsts[-k] - 30_000_000
    t1 = firsts[-k + 1] - 30_000_000 if k > 1 else rows[-1][3]
    last = None
    for r in rows:
        if not (t0 <= r[2] < t1) or r[1] not in range(s0, s0 + 4):
            continue
        s, e = (r[2] - t0) / 1e6, (r[3] - t0) / 1e6
        if last is not None and s - last > 1.0:
            print("        -- nothing on the lane for %.1f ms --" % (s - last))
        last = max(last or 0, e)
        if e - s >= mn:
            print("%7.2f %7.2f %6.2f s%-2d g%-9d %s" % (s, e, e - s, r[1], r[4], r[0][:44]))


if __name__ == "__main__":
    main()
