"""development aid: the at-scale scenario of tests/test_gpu_parity.py (two maps of the same part) in a child process per
environment variant, to bisect a crash"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, dataclasses, os
sys.path.insert(0, %r)
from longqc_amd import api, synth
n, nq = int(sys.argv[1]), int(sys.argv[2])
cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=n, nsample=nq)
genome = synth.make_genome(cfg)
F = synth.make_reads_flat(cfg, genome)
qi = synth.reservoir_subsample(n, cfg.nsample)
Q = synth.make_reads(cfg, genome, indices=qi)
p = api.default_params(no_self=1, min_ovlp=0, min_score_med=160, min_score_good=160)
eng = api.Engine(p, 0)
eng.set_queries(Q.names, Q.seqs, Q.quals)
pt = eng.part_begin()
P = api.PackedReads(F.flat, F.off, F.names())
eng.part_add_packed(pt, P)
t = None
for rep in range(int(sys.argv[3])):
    eng.reset(); eng.part_build(pt); eng.part_map(pt); eng.finish()
    t2 = eng.table_text()
    print("map", rep, "ok", eng.last_n_anchors, t is None or t2 == t, flush=True)
    t = t2
''' % ROOT
variants = [a.split(",") for a in sys.argv[4:]] or [[]]
for v in variants:
    env = dict(os.environ)
    for kv in v:
        if kv:
            k, _, val = kv.partition("=")
            env[k] = val
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, sys.argv[1], sys.argv[2], sys.argv[3]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=90)
    except subprocess.TimeoutExpired as e:
        print("=== variant", v, "TIMEOUT", (e.stdout or b"")[-600:]); continue
    print("=== variant", v, "rc", r.returncode, "%.1f s" % (time.time() - t0))
    print("\n".join(l for l in r.stdout.splitlines() if "amdgpu.ids" not in l)[-1500:], flush=True)
