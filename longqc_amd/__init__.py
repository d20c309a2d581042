"""longqc_amd -- MI355X-native stand-in for LongQC's `minimap2-coverage` hot path.

Python host over the C ABI of liblqcov.so (include/lqcov.h; hand-written HIP kernels for gfx950).
    LqCovExec   drop-in for lq_exec.LqExec on this path (same exec()/get_poll() shape)
    Engine      buffer-level API (queries, index parts, rows)
    synth       seeded synthetic read sets (tests, bench)
"""
from .api import Engine, Params, LqcovError, load_library, library_path  # noqa: F401
from .exec import LqCovExec, run_argv  # noqa: F401

__all__ = ["Engine", "Params", "LqcovError", "LqCovExec", "run_argv", "load_library", "library_path"]
