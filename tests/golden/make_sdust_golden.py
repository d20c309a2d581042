#!/usr/bin/env python3
"""Fixtures for SURVEY.md section 8(f)-4, written by the REFERENCE's own `sdust` binary (oracle/_ref/sdust, compiled from
/root/reference/minimap2-coverage/sdust.c by oracle/Makefile):

    python tests/golden/make_sdust_golden.py      (build container only; after make_golden.py)

One table per (input, -w, -t): name, masked bases, length, masked/length, meanQ, #qualities > Q7 (sdust.c:207-214).
"""
import gzip
import hashlib
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "sdust")


def main():
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/sdust is not built (make -C oracle ref)")
    os.chdir(HERE)
    cases = []
    for name, argv in [
        ("tiny_all", ["tiny_all.fq.gz"]),
        ("adv_all", ["adv_all.fa.gz"]),                       # FASTA: meanQ = -nan, CRLF multi-line records, N runs, poly-AT, 'u' bases
        ("adv_sub", ["adv_sub.fq.gz"]),
        ("adv_sub_w32_t10", ["-w", "32", "-t", "10", "adv_sub.fq.gz"]),
        ("adv_all_w16", ["-w16", "adv_all.fa.gz"]),
    ]:
        r = subprocess.run([REF] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        with gzip.GzipFile(name + ".sdust.gz", "wb", mtime=0) as f:
            f.write(r.stdout)
        cases.append(dict(name=name, argv=argv, expect=name + ".sdust.gz", md5=hashlib.md5(r.stdout).hexdigest(),
                          masked_total=sum(int(l.split(b"\t")[1]) for l in r.stdout.splitlines())))
    with open("sdust_cases.json", "w") as f:
        json.dump(cases, f, indent=1)
    for c in cases:
        print(c["name"], c["masked_total"])


if __name__ == "__main__":
    main()
