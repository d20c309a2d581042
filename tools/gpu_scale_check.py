#!/usr/bin/env python3
"""Scale check on the GPU box: N synthetic ONT reads (~15 kb) in several index parts (-I), table through the drop-in
executable vs the reference binary (oracle/_ref) on all host threads.  usage: gpu_scale_check.py N_READS N_QUERIES -I"""
import dataclasses, hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longqc_amd import synth
from tests import oracle_bind

n, nq, I = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=n, nsample=nq)
t0 = time.time(); T, Q = synth.make_dataset(cfg); print("generated %d reads, %.2f Gbases, %d queries in %.0f s" % (len(T), T.n_bases / 1e9, len(Q), time.time() - t0), flush=True)
tf, qf = "/tmp/sc_all.fq", "/tmp/sc_sub.fq"
t0 = time.time(); synth.write_fastq(tf, T); synth.write_fastq(qf, Q); print("fastq written in %.0f s" % (time.time() - t0), flush=True)
argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", I, "-p", "160"]
exe = os.path.join(ROOT, "longqc_amd", "minimap2-coverage-mi355x")
t0 = time.time()
a = subprocess.run([exe] + argv + ["-t", "8", tf, qf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
dt = time.time() - t0
log = a.stderr.decode()
print("MI355X executable: rc %d, %.1f s wall (parse + upload + compute) -> %.1f Mbases/s; parts: %d" % (a.returncode, dt, T.n_bases / dt / 1e6, log.count("target sequence(s)")), flush=True)
print("\n".join(l for l in log.splitlines() if l.startswith("[lqcov] part"))[:1500])
if oracle_bind.have_ref():
    t0 = time.time()
    b = subprocess.run([oracle_bind.REF_BIN] + argv + ["-t", str(os.cpu_count()), tf, qf], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    dt = time.time() - t0
    print("reference binary, %d threads: %.1f s -> %.1f Mbases/s" % (os.cpu_count(), dt, T.n_bases / dt / 1e6))
    print("tables identical:", a.stdout == b.stdout, "rows", a.stdout.count(b"\n"), "md5", hashlib.md5(a.stdout).hexdigest())
