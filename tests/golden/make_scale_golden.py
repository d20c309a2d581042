#!/usr/bin/env python3
"""Golden rows for the FULL-SIZE bench workloads, made by the REFERENCE ITSELF (oracle/_ref) in the
build container:

    python tests/golden/make_scale_golden.py cfg3      # BASELINE configs[2]: 500k PacBio CLR reads, 2 index parts
    python tests/golden/make_scale_golden.py cfg2      # BASELINE configs[1]: 50k ONT reads

The whole read set of the config is regenerated (synth.make_reads_flat: seeded, per-read RNG
streams, so the GPU box regenerates the same bytes), written as FASTA, and the reference maps a
small query set against it: the first N_HEAD reads of the seed-7 subsample plus the N_LONG longest
ones (the longest queries own the longest klib walks and chains).  A row depends only on its own
query and on the targets, so these rows are exactly the rows the 5000-query run prints for those
reads.  Output: tests/golden/<cfg>_rows.json = {argv, query indices into the config, rows (text)}.
Takes ~10 minutes and ~35 GB of RAM for cfg3 (the reference sketches single-threaded).
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from longqc_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "minimap2-coverage")
N_HEAD, N_LONG = 30, 10
ARGV = {
    "cfg3": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "80"],    # pb-sequel (longQC.py:171-220)
    "cfg2": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160"],   # ont-ligation
    "cfg4s": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160"],  # ont-rapid: the same minimap parameters (longQC.py:177-231)
    "cfg5s": ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160"],
}
# Slices of the 8-GPU configs with index parts of REAL size (-I 4G): SCALE_SLICES in longqc_amd/synth.py --
#   cfg4s: the first 600 000 reads' worth of BASELINE configs[3] (ONT ~20 kb, 40x): 12 Gbases = three 4-Gbase parts + a rest;
#   cfg5s: 75 000 reads of configs[4] (N50 100 kb): 4.5 Gbases = two parts, and a query set of 5000 reads ~ 300 Mbases, near
#          the 500-Mbase limit of one query mini-batch (bseq.c:86-98).


def main():
    name = sys.argv[1]
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/scale_golden"
    os.makedirs(work, exist_ok=True)
    cfg = synth.SCALE_SLICES[name] if name in synth.SCALE_SLICES else synth.CONFIGS[name]
    t0 = time.time()
    genome = synth.make_genome(cfg)
    tf = os.path.join(work, name + "_all.fa")
    if not os.path.exists(tf):
        F = synth.make_reads_flat(cfg, genome)
        synth.write_flat_fasta(tf + ".tmp", F)
        os.rename(tf + ".tmp", tf)
        del F
    print("targets written", time.time() - t0, flush=True)
    sub = synth.reservoir_subsample(cfg.n_reads, cfg.nsample)
    Q = synth.make_reads(cfg, genome, indices=sub)
    lens = np.array([s.shape[0] for s in Q.seqs])
    pick = list(range(N_HEAD))
    for j in np.argsort(-lens, kind="stable"):
        if len(pick) >= N_HEAD + N_LONG:
            break
        if int(j) not in pick:
            pick.append(int(j))
    qs = Q.subset(pick)
    qf = os.path.join(work, name + "_q.fq")
    synth.write_fastq(qf, qs)
    argv = ARGV[name] + ["-t", str(os.cpu_count() or 1), tf, qf]
    t0 = time.time()
    r = subprocess.run([REF] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    dt = time.time() - t0
    if r.returncode != 0:
        raise SystemExit("reference failed: " + r.stderr[-2000:])
    rows = r.stdout.splitlines()
    assert len(rows) == len(pick)
    out = {"config": name, "argv": ARGV[name], "subsample_slots": pick, "read_indices": [int(sub[j]) for j in pick],
           "query_lengths": [int(lens[j]) for j in pick], "rows": rows,
           "reference_seconds": round(dt, 1), "reference_threads": os.cpu_count(),
           "made_by": "tests/golden/make_scale_golden.py (oracle/_ref/minimap2-coverage on the whole synthetic set)"}
    with open(os.path.join(HERE, name + "_rows.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("reference took", dt, "s;", len(rows), "rows", flush=True)
    print(r.stderr[-1500:])


if __name__ == "__main__":
    main()
