#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel HBM traffic.

    python tools/pmc_to_json.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json> [workload tag]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE / WRITE_SIZE are in
KiB of fabric requests; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream, so reads
are doubled; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  (Narrow / scattered accesses are uncalibrated there;
the raw values are kept next to the corrected one.)"""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    if not f:
        return agg
    for row in csv.DictReader(open(f[0])):
        if row.get("Counter_Name") != counter:
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    return agg


def main():
    fd, wd, out = sys.argv[1], sys.argv[2], sys.argv[3]
    tag = sys.argv[4] if len(sys.argv) > 4 else ""
    F, W = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    commit = ""
    try:
        commit = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_commit")).read().strip()
    except Exception:
        pass
    res = {"workload": tag, "commit": commit, "steps": 1, "formula": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 read correction)", "kernels": {}}
    for k in sorted(set(F) | set(W)):
        fs, fn = F.get(k, [0.0, 0]); ws, wn = W.get(k, [0.0, 0])
        n = max(fn, wn, 1)
        res["kernels"][k] = {"dispatches": n, "fetch_kib_per_launch": fs / max(fn, 1), "write_kib_per_launch": ws / max(wn, 1),
                             "hbm_bytes_per_launch": (2 * fs / max(fn, 1) + ws / max(wn, 1)) * 1024}
    json.dump(res, open(out, "w"), indent=1)
    top = sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:12]
    for k, v in top:
        print("%-40s x%-4d %10.3f GB/launch" % (k[:40], v["dispatches"], v["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
