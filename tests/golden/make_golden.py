#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REFERENCE ITSELF (oracle/_ref, compiled from
/root/reference/minimap2-coverage by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Fixtures are data: input read files (small, gz) and the reference's outputs for them --
the 9-column table (minimap2-coverage.c:545-617) for several argv, and function-level dumps
(mm_sketch lists, per-part mid_occ, chains) from oracle/ref_harness.c.  cases.json records the
argv of every expected file.  Nothing here contains reference source text.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from longqc_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "minimap2-coverage")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def adversarial():
    """A hand-built read set that walks the edge cases of SURVEY.md section 8(c)."""
    rng = np.random.default_rng(424242)
    A = synth._ACGT
    g = A[rng.integers(0, 4, size=30000, dtype=np.uint8)]
    g[3000:3300] = np.frombuffer(b"AT" * 150, dtype=np.uint8)          # palindromic 12-mers for hundreds of bases
    g[7000:7300] = np.frombuffer(b"ACG" * 100, dtype=np.uint8)         # tandem repeat: equal-hash window ties
    g[9000:9060] = ord("A")                                            # homopolymer run (HPC)
    g[12000:12400] = np.tile(g[11600:11800], 2)                        # 200-bp unit twice
    names, seqs, quals = [], [], []

    def add(name, s, q=None):
        s = np.ascontiguousarray(s, dtype=np.uint8)
        names.append(name); seqs.append(s)
        quals.append(q if q is not None else (33 + rng.integers(3, 40, size=s.shape[0])).astype(np.uint8))

    for i in range(420):                                               # pile-up: > 150x over one locus (COVT cap)
        s = synth._mutate(g[20000:21000], rng, 0.03, (3, 3, 4))
        add("pile%04d" % i, s)
    for i in range(260):                                               # ordinary noisy reads, ~12x
        L = int(rng.integers(800, 3000)); st = int(rng.integers(0, 30000 - L))
        s = g[st:st + L]
        if rng.random() < 0.5:
            s = synth._COMP[s[::-1]]
        add("adv%04d" % i, synth._mutate(s, rng, 0.08, (3, 3, 4)))
    s = g[5000:6500].copy(); s[400:410] = ord("N"); s[900] = ord("N"); add("with_N", s)
    add("lower_U", np.frombuffer(g[6000:7600].tobytes().lower().replace(b"t", b"u"), dtype=np.uint8))
    add("tiny", np.frombuffer(b"ACGTA", dtype=np.uint8))
    add("eleven", g[100:111])                                          # shorter than k=12
    add("exactk", g[200:212])
    add("polyAT", g[2900:3400])
    add("dup", g[15000:16200]); add("dup", g[16000:17400])              # duplicate names
    add("selfsame", g[22000:23500])                                    # also a query, identical
    add("allN", np.full(300, ord("N"), dtype=np.uint8))
    targets = synth.ReadSet(names, seqs, quals)
    qidx = [420, 425, 437, 520, 679, 0, 40, 419, 680, 681, 682, 683, 684, 685, 686, 687, 688, 689]
    queries = targets.subset(qidx)
    # a query that shares a target's name but not its sequence, and one that exists nowhere
    queries.names.append("adv0003"); queries.seqs.append(synth._mutate(g[25000:26500], rng, 0.05, (3, 3, 4)))
    queries.quals.append((33 + rng.integers(3, 40, size=queries.seqs[-1].shape[0])).astype(np.uint8))
    queries.names.append("junkq"); queries.seqs.append(A[rng.integers(0, 4, size=1500, dtype=np.uint8)])
    queries.quals.append(np.full(1500, ord("I"), dtype=np.uint8))
    return targets, queries


def run(cmd, out_path):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise SystemExit("failed: %s\n%s" % (" ".join(cmd), r.stderr.decode()[-2000:]))
    with gzip.GzipFile(out_path, "wb", mtime=0) as f:
        f.write(r.stdout)
    return hashlib.md5(r.stdout).hexdigest()


def main():
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref is not built (make -C oracle ref)")
    cases = []
    # --- inputs ---
    T, Q = synth.make_dataset(synth.CONFIGS["tiny"])
    synth.write_fastq(os.path.join(HERE, "tiny_all.fq.gz"), T)
    synth.write_fastq(os.path.join(HERE, "tiny_sub.fq.gz"), Q)
    aT, aQ = adversarial()
    synth.write_fastq(os.path.join(HERE, "adv_all.fa.gz"), aT, fasta=True, line_width=70, crlf=True)
    synth.write_fastq(os.path.join(HERE, "adv_sub.fq.gz"), aQ)
    synth.write_fastq(os.path.join(HERE, "adv_sub.fa.gz"), aQ, fasta=True)
    tables = {
        "tiny_ont": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-t", "4", "tiny_all.fq.gz", "tiny_sub.fq.gz"],
        "tiny_pb": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "80", "-t", "4", "tiny_all.fq.gz", "tiny_sub.fq.gz"],
        "tiny_k15": ["-Y", "-l", "0", "-q", "160", "-k", "15", "-w", "5", "-I", "4G", "-p", "160", "-t", "4", "tiny_all.fq.gz", "tiny_sub.fq.gz"],
        "tiny_k19w10": ["-Y", "-l", "0", "-q", "160", "-k", "19", "-w", "10", "-I", "4G", "-p", "160", "-t", "4", "tiny_all.fq.gz", "tiny_sub.fq.gz"],
        "tiny_parts": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "100K", "-p", "160", "-t", "4", "tiny_all.fq.gz", "tiny_sub.fq.gz"],
        "tiny_spike": ["-Y", "-Hk15", "-w", "10", "-c", "1", "-l", "0", "--filter", "-t", "4", "tiny_all.fq.gz", "tiny_sub.fq.gz"],
        "adv_ont": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-t", "4", "adv_all.fa.gz", "adv_sub.fq.gz"],
        "adv_parts": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "100K", "-p", "160", "-t", "4", "adv_all.fa.gz", "adv_sub.fq.gz"],
        "adv_fasta_query": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-t", "4", "adv_all.fa.gz", "adv_sub.fa.gz"],
        "adv_spike": ["-Y", "-Hk15", "-w", "10", "-c", "1", "-l", "0", "--filter", "-t", "4", "adv_all.fa.gz", "adv_sub.fq.gz"],
        "adv_defaults": ["-Y", "-t", "2", "adv_all.fa.gz", "adv_sub.fq.gz"],
        "adv_w20": ["-Y", "-l", "0", "-k", "11", "-w", "20", "-m", "30", "-n", "2", "-t", "2", "adv_all.fa.gz", "adv_sub.fq.gz"],
        # -X (MM_F_AVA): hits on targets whose name sorts before the query's are dropped (lqmap.c:187)
        "tiny_ava": ["-X", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-t", "4", "tiny_all.fq.gz", "tiny_sub.fq.gz"],
        "adv_ava_parts": ["-X", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "100K", "-p", "160", "-t", "4", "adv_all.fa.gz", "adv_sub.fq.gz"],
    }
    os.chdir(HERE)
    for name, argv in tables.items():
        md5 = run([REF] + argv, name + ".table.gz")
        cases.append(dict(kind="table", name=name, argv=argv, expect=name + ".table.gz", md5=md5))
    dumps = {
        "tiny_sketch_k12w5": ["sketch", "12", "5", "0", "tiny_sub.fq.gz"],
        "adv_sketch_k12w5": ["sketch", "12", "5", "0", "adv_sub.fq.gz"],
        "adv_sketch_k15w10hpc": ["sketch", "15", "10", "1", "adv_sub.fq.gz"],
        "adv_sketch_k19w10": ["sketch", "19", "10", "0", "adv_sub.fq.gz"],
        "adv_sketch_k6w30": ["sketch", "6", "30", "0", "adv_sub.fq.gz"],
        "adv_index_100K": ["index", "12", "5", "0", "100K", "adv_all.fa.gz"],
        "tiny_index_100K": ["index", "12", "5", "0", "100K", "tiny_all.fq.gz"],
        "tiny_chains": ["chains", "12", "5", "0", "4G", "40", "160", "160", "tiny_all.fq.gz", "tiny_sub.fq.gz"],
        "adv_chains": ["chains", "12", "5", "0", "4G", "40", "160", "160", "adv_all.fa.gz", "adv_sub.fq.gz"],
    }
    for name, args in dumps.items():
        md5 = run([HARNESS] + args, name + ".dump.gz")
        cases.append(dict(kind="dump", name=name, args=args, expect=name + ".dump.gz", md5=md5))
    with open("cases.json", "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote %d cases" % len(cases))
    os.system("du -sh .")


if __name__ == "__main__":
    main()
