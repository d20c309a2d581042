#!/bin/bash
# The emulator build of the kernels (tests/emu) under AddressSanitizer: every out-of-bounds access of a kernel to a "device"
# buffer (heap) or to LDS (static arrays) stops the run with a report.  No GPU needed.
#   bash tools/emu_asan.sh [pytest args, default: the whole emulator suite]
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/liblqcov_emu_asan.so
( cd longqc_amd/csrc && g++ -DLQ_EMU -include ../../tests/emu/hipemu.hpp -DLQ_EXACT_ALLOC -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC \
    -Wno-unused-function -Wno-unknown-pragmas engine.cpp api.cpp dust.cpp -shared -o "$OUT" -lz ) || exit 1
# (libstdc++ preloaded too: the sanitizer's __cxa_throw interceptor needs it at start-up, and the tests of refused inputs throw)
LQCOV_EMU_LIB="$OUT" LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_emu_pipeline.py tests/test_mmi.py tests/test_sdust.py -x -q "${@:--n 6}"
