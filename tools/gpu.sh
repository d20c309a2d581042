#!/bin/bash
# local helper: rebuild the HIP library (never ship a stale .so), then run a command on the GPU box.
#   tools/gpu.sh <timeout_s> '<command>' > log
set -e
cd /root/repo
make -C longqc_amd/csrc all 2>&1 | grep -E "error" && exit 1
rm -rf gpurun_out
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
