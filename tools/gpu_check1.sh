#!/bin/bash
# first contact with the GPU: CLI vs oracle/_ref on small synthetic sets, timings, rocprof.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w && W=gpurun_out/w
python - <<'PY'
import sys, time
sys.path.insert(0,'.')
from longqc_amd import synth
import dataclasses
for c in ['tiny','small','cfg1']:
    T,Q = synth.make_dataset(synth.CONFIGS[c])
    synth.write_fastq(f'gpurun_out/w/{c}_all.fq',T); synth.write_fastq(f'gpurun_out/w/{c}_sub.fq',Q)
cfg = dataclasses.replace(synth.CONFIGS['cfg2'], n_reads=5000, nsample=1000, name='mid')
t=time.time(); T,Q = synth.make_dataset(cfg); print('gen mid', time.time()-t, T.n_bases, Q.n_bases)
synth.write_fastq('gpurun_out/w/mid_all.fq',T); synth.write_fastq('gpurun_out/w/mid_sub.fq',Q)
PY
rocminfo | grep -E "Marketing|gfx9" | head -4
nproc; lscpu | grep "Model name"
for c in tiny small cfg1 mid; do
  A="-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 160 -t $(nproc) $W/${c}_all.fq $W/${c}_sub.fq"
  /usr/bin/env time -f "ref %e s" oracle/_ref/minimap2-coverage $A > $W/${c}_ref.txt 2> $W/${c}_ref.err; tail -1 $W/${c}_ref.err
  /usr/bin/env time -f "gpu %e s" timeout 600 longqc_amd/minimap2-coverage-mi355x $A > $W/${c}_gpu.txt 2> $W/${c}_gpu.err; echo "rc=$?"; tail -4 $W/${c}_gpu.err
  cmp $W/${c}_ref.txt $W/${c}_gpu.txt && echo "PARITY OK $c" || { echo "PARITY FAIL $c"; diff $W/${c}_ref.txt $W/${c}_gpu.txt | head -6; }
done
A="-Y -l 0 -q 160 -k 12 -w 5 -I 1M -p 160 -t 8 $W/cfg1_all.fq $W/cfg1_sub.fq"
oracle/_ref/minimap2-coverage $A > $W/mp_ref.txt 2>/dev/null; timeout 600 longqc_amd/minimap2-coverage-mi355x $A > $W/mp_gpu.txt 2>$W/mp_gpu.err; cmp $W/mp_ref.txt $W/mp_gpu.txt && echo "PARITY OK multipart" || echo "PARITY FAIL multipart"
A="-Y -Hk15 -w 10 -c 1 -l 0 --filter -t 4 $W/small_all.fq $W/small_sub.fq"
oracle/_ref/minimap2-coverage $A > $W/h_ref.txt 2>/dev/null; timeout 600 longqc_amd/minimap2-coverage-mi355x $A > $W/h_gpu.txt 2>$W/h_gpu.err; cmp $W/h_ref.txt $W/h_gpu.txt && echo "PARITY OK hpc" || echo "PARITY FAIL hpc"
cd /tmp && export TMPDIR=/tmp
A="-Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 160 -t 8 $GRAFT_REPO_ROOT/$W/mid_all.fq $GRAFT_REPO_ROOT/$W/mid_sub.fq"
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o mid -- $GRAFT_REPO_ROOT/longqc_amd/minimap2-coverage-mi355x $A > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof1.err
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head; 
F=$(find $GRAFT_REPO_ROOT/gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); head -25 $F
rm -f $GRAFT_REPO_ROOT/$W/*.fq
