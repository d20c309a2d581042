#!/usr/bin/env python3
"""GPU check of the sdust path: configs[1] reads (50k x ~15 kb) through lqsdust_main; time vs the reference binary on a sample."""
import dataclasses, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longqc_amd import synth, sdust
from tests import oracle_bind

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=n, nsample=10)
T, _ = synth.make_dataset(cfg)
fq = "/tmp/sd_all.fq"
synth.write_fastq(fq, T)
for it in range(2):
    t0 = time.time(); sdust.run_sdust(fq, "/tmp/sd_gpu.txt"); dt = time.time() - t0
    print("GPU sdust (parse + upload + kernel + rows): %.2f s -> %.1f Mbases/s" % (dt, T.n_bases / dt / 1e6), flush=True)
ref = os.path.join(os.path.dirname(oracle_bind.REF_BIN), "sdust")
sub = "/tmp/sd_sub.fq"
synth.write_fastq(sub, T.subset(range(min(2000, len(T)))))
if os.path.exists(ref):
    t0 = time.time(); a = subprocess.run([ref, sub], stdout=subprocess.PIPE, check=True).stdout.decode(); dt = time.time() - t0
    nb = sum(int(s.shape[0]) for s in T.seqs[:2000])
    print("reference sdust, 1 thread, first 2000 reads: %.2f s -> %.1f Mbases/s" % (dt, nb / dt / 1e6))
    sdust.run_sdust(sub, "/tmp/sd_gpu_sub.txt")
    print("parity on the sample:", "OK" if open("/tmp/sd_gpu_sub.txt").read() == a else "FAIL")
