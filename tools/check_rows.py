#!/usr/bin/env python3
"""the rows the reference binary printed for 40 queries of a workload (tests/golden/<cfg>_rows.json) against a table file
    python tools/check_rows.py cfg3 /dev/shm/out_1.tsv"""
import hashlib, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = json.load(open(os.path.join(root, "tests", "golden", sys.argv[1] + "_rows.json")))
data = open(sys.argv[2], "rb").read()
lines = data.decode().splitlines()
bad = [s for s, row in zip(g["subsample_slots"], g["rows"]) if s >= len(lines) or lines[s] != row]
print("%s: %d rows, md5 %s, golden rows %d/%d" % (sys.argv[2], len(lines), hashlib.md5(data).hexdigest(), len(g["rows"]) - len(bad), len(g["rows"])))
