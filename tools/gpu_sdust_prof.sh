#!/bin/bash
# per-kernel time of the sdust path on a slice of configs[1]
cd $GRAFT_REPO_ROOT
python - <<'PY'
import dataclasses, sys
sys.path.insert(0, ".")
from longqc_amd import synth
cfg = dataclasses.replace(synth.CONFIGS["cfg2"], n_reads=5000, nsample=10)
T, _ = synth.make_dataset(cfg)
synth.write_fastq("/tmp/sd5k.fq", T)
print("bases", T.n_bases)
PY
cd /tmp && export TMPDIR=/tmp
time ( $GRAFT_REPO_ROOT/longqc_amd/sdust-mi355x /tmp/sd5k.fq > /tmp/o.txt )
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sdp -o s -- $GRAFT_REPO_ROOT/longqc_amd/sdust-mi355x /tmp/sd5k.fq > /tmp/o2.txt 2>/tmp/e2.txt
cut -c1-150 /tmp/sdp/s_kernel_stats.csv | head -5
