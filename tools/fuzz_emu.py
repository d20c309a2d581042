#!/usr/bin/env python3
"""Differential fuzz without a GPU: random small read sets, random minimap2-coverage argv, random engine knobs -- the table of the
emulator build of the kernels (tests/emu) against the table the reference binary (oracle/_ref) prints for the same files and argv.
Every case runs in a child process (a crash is a finding, not the end of the run) and is fully described by its seed.

    python tools/fuzz_emu.py --seeds 0:200 --jobs 8          # a campaign; findings are listed at the end
    python tools/fuzz_emu.py --one 17 [--keep DIR]           # one case, verbose: argv, knobs, first differing row

The kernels are the product's (compiled as C++ for the CPU); what this cannot see is anything that only shows on the device
(races between waves the emulator's schedule does not produce, HIP runtime behaviour).  Findings so far: see profiles/README.md."""
import argparse
import dataclasses
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_case(seed: int, d: str):
    """-> (argv without the two files, env knobs, target path, query path, description)"""
    import numpy as np
    from longqc_amd import synth
    rng = np.random.default_rng([seed, 20260928])
    kind = int(rng.integers(0, 4))
    if kind <= 1:                                   # reads of a random genome: coverage, errors and lengths all over the place
        cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=int(rng.integers(30, 350)), mean_len=int(rng.integers(300, 4000)), min_len=int(rng.integers(20, 400)),
                                  depth=float(rng.choice([1.5, 3, 8, 20, 45])), err=float(rng.choice([0.0, 0.02, 0.08, 0.13])), seed=5000 + seed,
                                  nsample=int(rng.integers(3, 30)), n_frac=float(rng.choice([0.0, 0.0, 0.2])), qual=str(rng.choice(["ont", "none"])),
                                  junk_frac=float(rng.choice([0.0, 0.02, 0.2])))
        T, Q = synth.make_dataset(cfg)
        what = "synth %s" % (dataclasses.asdict(cfg),)
    elif kind == 2:                                 # tandem repeats and dispersed copies: equal-x anchors everywhere (klib's order matters)
        import tests.test_emu_pipeline as E
        import pathlib
        tf, qf = E._repeat_rich_dataset(pathlib.Path(d), seed, n_targets=int(rng.integers(20, 70)), n_queries=int(rng.integers(3, 9)), glen=int(rng.integers(20000, 50000)))
        T = Q = None
        what = "repeat-rich"
    else:                                           # a pile-up: a few queries under many pieces of themselves
        import tests.test_emu_pipeline as E
        import pathlib
        tf, qf = E._pileup_dataset(pathlib.Path(d), int(rng.integers(200, 3000)), seed=seed, n_hot=int(rng.integers(1, 4)), qlen=int(rng.integers(900, 1500)))
        T = Q = None
        what = "pile-up"
    if T is not None:
        fasta_t, fasta_q = bool(rng.random() < 0.3), bool(rng.random() < 0.2)
        tf, qf = os.path.join(d, "all." + ("fa" if fasta_t else "fq")), os.path.join(d, "sub." + ("fa" if fasta_q else "fq"))
        synth.write_fastq(tf, T, fasta=fasta_t, line_width=int(rng.choice([0, 0, 60])) if fasta_t else 0, crlf=bool(rng.random() < 0.1))
        synth.write_fastq(qf, Q, fasta=fasta_q)
    hpc = rng.random() < 0.15
    k = int(rng.choice([8, 10, 11, 12, 12, 12, 13, 15, 15, 19]))
    w = int(rng.choice([1, 3, 5, 5, 5, 8, 10, 12]))
    m = int(rng.choice([10, 20, 40, 40, 60, 100]))
    n = int(rng.choice([1, 2, 3, 3, 3, 5, 8]))
    p = m + int(rng.choice([0, 20, 40, 120]))
    q = p + int(rng.choice([0, 40, 120]))                   # (-q below -p or -m is refused by both binaries)
    argv = ["-Y", "-l", "0", "-k", str(k), "-w", str(w), "-m", str(m), "-n", str(n), "-p", str(p), "-q", str(q), "-t", "3"]
    if hpc:
        argv.append("-H")
    if rng.random() < 0.5:
        argv += ["-I", str(rng.choice(["15K", "40K", "150K", "600K"]))]
    if rng.random() < 0.25:
        argv += ["-c", str(rng.choice([1, 2, 5]))]
    if rng.random() < 0.15:
        argv += ["-g", str(rng.choice([200, 2000, 5000]))]
    if rng.random() < 0.15:
        argv += ["-s", str(rng.choice([1, 5, 50]))]
    if rng.random() < 0.15:
        argv += ["-a", str(rng.choice([0, 100, 5000]))]
    if rng.random() < 0.15:
        argv += ["-r", str(rng.choice([0.1, 0.7, 0.95]))]
    if rng.random() < 0.1:
        argv.append("--filter")
    if rng.random() < 0.1:
        argv[0] = "-X"                              # (instead of -Y: the two exclude each other)
    env = {}
    knobs = [("LQCOV_SEED_BUCKET", ["64", "300", "3000"]), ("LQCOV_SEED_SEGL", ["7", "50", "256"]), ("LQCOV_SEED_CHUNK", ["1024", "4000"]), ("LQCOV_SEED_PAIR_BITS", ["3", "4", "8"]),
             ("LQCOV_SEED_HWORDS", ["40", "200"]), ("LQCOV_SEED_DCAP", ["300", "500"]), ("LQCOV_SEED_BIGCAP", ["1500"]), ("LQCOV_SEED_SURV_MAX", ["1000", "20000"]),
             ("LQCOV_LANES", ["1", "2", "3"]), ("LQCOV_ANCHOR_BUDGET", ["3000", "20000"]), ("LQCOV_CHAIN_WAVE_MIN", ["3", "48", "1000"]), ("LQCOV_PLAN_AHEAD", ["0"]),
             ("LQCOV_UPLOAD_SLICES", ["1", "3", "8"]), ("LQCOV_UPLOAD_MIN_CHUNKS", ["1"]), ("LQCOV_PIPELINE", ["0"]), ("LQCOV_PARSE_THREADS", ["1", "5"]), ("LQCOV_PARSE_PIECE", ["4096"]),
             ("LQCOV_RUN_GRID", ["1", "3", "7"]), ("LQCOV_TILE_GRID", ["2"]), ("LQCOV_SORT_TILE", ["64"]), ("LQCOV_PS_SHIFT", ["5"]), ("LQCOV_WALK_SHIFT", ["4", "7"]),
             ("LQCOV_CKPT3", ["1"]), ("LQCOV_SKETCH", ["machine"]), ("LQCOV_QUERY_ORDER", ["file"]), ("LQ_EMU_ORDER", ["reverse", "random:%d" % seed])]
    for name, vals in knobs:
        if rng.random() < 0.18:
            env[name] = str(rng.choice(vals))
    return argv, env, tf, qf, what


def run_one(seed: int, keep: str = "", verbose: bool = False) -> int:
    from tests import oracle_bind
    from tests.helpers import run_main
    from longqc_amd import api
    emu = os.environ.get("LQCOV_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "liblqcov_emu.so")
    with tempfile.TemporaryDirectory() as tmp:
        d = keep or tmp
        os.makedirs(d, exist_ok=True)
        argv, env, tf, qf, what = make_case(seed, d)
        full = argv + [tf, qf]
        if verbose:
            print("seed %d: %s\n  argv: %s\n  knobs: %s" % (seed, what, " ".join(full), env), flush=True)
        try:
            want = oracle_bind.ref_table(full)
        except RuntimeError as e:                  # the reference refuses or crashes on this argv: not a case
            if verbose:
                print("  reference: %s" % str(e)[:300])
            return 2
        if os.environ.get("FUZZ_KNOBS"):             # bisecting a finding: FUZZ_KNOBS="A=1,B=2" replaces the case's knobs ("none": no knobs)
            env = dict(kv.split("=", 1) for kv in os.environ["FUZZ_KNOBS"].split(",") if "=" in kv)
            print("  knobs replaced: %s" % env, flush=True)
        os.environ.update(env)
        lib = api.load_library(emu)
        # a third of the cases go through an index file (sampleqc --db, longQC.py:266-277): the engine writes one with -d and maps from
        # it, or maps from the one the reference wrote; the reference's table from its own file is the same `want` (checked)
        via = seed % 3 if "-I" in argv or seed % 2 else 0
        if via:
            import subprocess
            kw = [a for i, a in enumerate(argv) if a in ("-k", "-w", "-H", "-I") or (i and argv[i - 1] in ("-k", "-w", "-I"))]
            # (the mapping call keeps -k / -w / -H: the reference sizes its match counters from the command line's sketch,
            # minimap2-coverage.c:419-422, and writes past them when the index file's parameters yield more minimizers)
            rest = [a for i, a in enumerate(argv) if not (a == "-I" or (i and argv[i - 1] == "-I"))]
            mmi_ref, mmi_own = os.path.join(d, "ref.mmi"), os.path.join(d, "own.mmi")
            r = subprocess.run([oracle_bind.REF_BIN] + kw + ["-d", mmi_ref, tf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode != 0:
                return 2
            want2 = oracle_bind.ref_table(rest + [mmi_ref, qf])
            if want2 != want:                       # (the reference itself: mapping from its dump = mapping from the reads)
                print("seed %d: the reference's table from its own index file differs from its table from the reads" % seed)
                return 1
            if via == 1:
                rc, out, err = run_main(lib, kw + ["-d", mmi_own, tf])
                if rc != 0:
                    print("seed %d DIFFERS: -d failed (rc %d): %s" % (seed, rc, err[-500:]))
                    return 1
            full = rest + [mmi_own if via == 1 else mmi_ref, qf]
            if verbose:
                print("  through an index file written by %s" % ("the engine" if via == 1 else "the reference"), flush=True)
        rc, out, err = run_main(lib, full)
        if rc == 0 and out == want:
            if verbose:
                print("  identical (%d rows, %d with coverage)" % (len(want.splitlines()), sum(1 for l in want.splitlines() if l.split("\t")[2] != "0")))
            return 0
        print("seed %d DIFFERS (rc %d): %s\n  argv: %s\n  knobs: %s" % (seed, rc, what, " ".join(full), env))
        a, b = want.splitlines(), out.splitlines()
        for i in range(max(len(a), len(b))):
            if i >= len(a) or i >= len(b) or a[i] != b[i]:
                print("  first differing row %d:\n    reference: %s\n    engine:    %s" % (i, a[i] if i < len(a) else "<none>", b[i] if i < len(b) else "<none>"))
                break
        print("  log tail: %s" % " | ".join(err.splitlines()[-3:]))
        return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:40", help="lo:hi")
    ap.add_argument("--jobs", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--one", type=int, default=-1)
    ap.add_argument("--keep", default="", help="--one: directory that keeps the case's files")
    ap.add_argument("--limit", type=int, default=600, help="seconds per case")
    args = ap.parse_args()
    if args.one >= 0:
        sys.exit(run_one(args.one, args.keep, True))
    lo, hi = (int(v) for v in args.seeds.split(":"))
    todo, running, res = list(range(lo, hi)), {}, {}
    t0 = time.time()
    while todo or running:
        while todo and len(running) < args.jobs:
            s = todo.pop(0)
            running[s] = (subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", str(s)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT), time.time())
        time.sleep(0.2)
        for s, (pr, ts) in list(running.items()):
            rc = pr.poll()
            if rc is None and time.time() - ts > args.limit:
                pr.kill()
                rc = -9
            if rc is not None:
                res[s] = (rc, pr.stdout.read() if pr.stdout else "")
                del running[s]
                if rc == -9:                          # (mid_occ of a tiny read set is "the largest count + 1": one tandem repeat shared by query and
                    print("---- seed %d: not finished within %d s\n%s" % (s, args.limit, res[s][1][-1500:]), flush=True)   # targets is tens of millions of seed hits, minutes on the emulator)
                elif rc not in (0, 2):
                    print("---- seed %d: exit %d\n%s" % (s, rc, res[s][1][-3000:]), flush=True)
    ok = sum(1 for r in res.values() if r[0] == 0)
    skipped = sum(1 for r in res.values() if r[0] == 2)
    slow = sorted(s for s, r in res.items() if r[0] == -9)
    bad = sorted(s for s, r in res.items() if r[0] not in (0, 2, -9))
    print("%d cases in %.0f s: %d identical, %d not a case for the reference, %d not finished in time %s, findings: %s" % (len(res), time.time() - t0, ok, skipped, len(slow), slow or "", bad or "none"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
