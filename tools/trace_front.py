#!/usr/bin/env python3
"""every kernel of the build stream during the first part's front and seed plan, with the gaps between them
    python tools/trace_front.py gpurun_out/trace_small.csv [--step 2] [--ms 240]"""
import sys


def main():
    fn = sys.argv[1]
    k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 2
    span = float(sys.argv[sys.argv.index("--ms") + 1]) if "--ms" in sys.argv else 240.0
    rows = []
    for line in open(fn):
        f = line.rstrip("\n").split(",")
        rows.append((f[0].replace("void ", ""), f[1], int(f[2]), int(f[3]), f[4]))
    rows.sort(key=lambda r: r[2])
    dp = [r for r in rows if r[0].startswith("k_sketch_dp_mask")]
    big = max(int(r[4]) for r in dp)
    firsts = [r for r in dp if int(r[4]) == big]
    t0 = firsts[-k][2] - 30_000_000
    bs = firsts[-k][1]
    step = [r for r in rows if t0 <= r[2] < t0 + span * 1e6 and r[1] == bs]
    prev = None
    for r in step:
        gap = (r[2] - prev) / 1e6 if prev else 0.0
        d = (r[3] - r[2]) / 1e6
        if d > 0.25 or gap > 0.4:
            print("%7.2f  +%6.2f  gap %5.2f  %s" % ((r[2] - t0) / 1e6, d, gap, r[0][:70]))
        prev = r[3]


if __name__ == "__main__":
    main()
