#!/bin/bash
# the executable alone on the bench workload's files (written once to /dev/shm): stderr tail, exit status, table md5.
# Every run under its own short time limit (a wedged GPU must not eat the budget).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python - <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from longqc_amd import synth
cfg = synth.CONFIGS[os.environ.get("CFG", "cfg3")]
g = synth.make_genome(cfg)
F = synth.make_reads_flat(cfg, g)
Q = synth.make_reads(cfg, g, indices=synth.reservoir_subsample(cfg.n_reads, cfg.nsample))
synth.write_flat_fasta("/dev/shm/all.fa", F); synth.write_fastq("/dev/shm/sub.fq", Q)
PY
N=0
IFS='|' read -ra VS <<< "${VARIANTS:-base}"
for V in "${VS[@]}"; do
  N=$((N+1)); E="$V"; [ "$V" = base ] && E="X=1"
  ( time env $E LQCOV_DEVICE=0 timeout ${LIMIT:-60} longqc_amd/minimap2-coverage-mi355x -Y -l 0 -q 160 -k 12 -w 5 -I 4G -p 80 -t 8 /dev/shm/all.fa /dev/shm/sub.fq > /dev/shm/out_$N.tsv 2> gpurun_out/exe_$N.err ) 2> gpurun_out/exe_$N.time
  echo "== $V: exit $?"; python tools/check_rows.py ${CFG:-cfg3} /dev/shm/out_$N.tsv; grep -v "^\[lqcov\] launch" gpurun_out/exe_$N.err | tail -${TAIL:-6}; grep "launch" gpurun_out/exe_$N.err | tail -4; grep real gpurun_out/exe_$N.time
done
