"""LqCovExec -- the counterpart of the reference's process boundary for this path.

Reference: lq_exec.py:13-38 (`LqExec.exec(*args, out=, err=)` -> Popen with stdout/stderr to
files), lq_exec.py:70-71 (`get_poll()`), driven by longQC.py:438-446 and polled at :520-526.
Same names, same argument meaning: exec(*argv, out=path, err=path) starts the run asynchronously,
get_poll() returns None while it runs and the exit status afterwards (the reference ignores the
status; here a failed run is 1 / negative and get_error() carries the message).

Two back ends, both the HIP build: in-process (ctypes -> lqcov_main on a worker thread; ctypes
releases the GIL) or the argv-compatible executable `minimap2-coverage-mi355x` as a subprocess
(what longQC.py would spawn if its binary path pointed here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile
import threading
from typing import Optional, Sequence

from . import api

_HERE = os.path.dirname(os.path.abspath(__file__))
# in-process runs on one device take turns: every handle sizes its work space from the HBM that is free when it is created
_DEVICE_LOCKS = {}
_DEVICE_LOCKS_GUARD = threading.Lock()


def _device_lock(device: int) -> threading.Lock:
    with _DEVICE_LOCKS_GUARD:
        return _DEVICE_LOCKS.setdefault(device, threading.Lock())


def run_argv(argv: Sequence[str], out: Optional[str] = None, err: Optional[str] = None, device: int = 0) -> int:
    """Blocking `minimap2-coverage argv...` (argv without the program name)."""
    lib = api.load_library()
    full = [b"minimap2-coverage"] + [str(a).encode() for a in argv]
    arr = (C.c_char_p * len(full))(*full)
    return int(lib.lqcov_main(len(full), arr, out.encode() if out else None, err.encode() if err else None, device))


class LqCovExec:
    def __init__(self, bin_path: Optional[str] = None, ncpu: int = 1, device: int = 0, subprocess_mode: bool = False):
        self.bin_path = bin_path or os.path.join(_HERE, "minimap2-coverage-mi355x")
        self.device = device
        self.subprocess_mode = subprocess_mode
        self.proc = None
        self._thread = None
        self._rc = None
        self._err_path = None
        self.out_path = None
        self._tmp = []                    # files made for output the caller did not name: removed by close() / on collection

    def close(self):
        """remove the files that stood in for pipes the caller never named (the reference drops that output too)"""
        for fn in self._tmp:
            try:
                os.unlink(fn)
            except OSError:
                pass
        self._tmp = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exec(self, *args, out=None, err=None):
        self.close()
        self._err_path = err
        self._rc = None
        if self.subprocess_mode:
            fout = open(out, "w") if out else subprocess.PIPE
            ferr = open(err, "w") if err else subprocess.PIPE
            env = dict(os.environ, LQCOV_DEVICE=str(self.device))
            try:
                self.proc = subprocess.Popen([self.bin_path] + [str(a) for a in args], stdout=fout, stderr=ferr, env=env)
            finally:
                if out:
                    fout.close()
                if err:
                    ferr.close()
            return

        # the reference pipes what the caller does not name (and never reads the pipes): here such output goes to files of
        # its own instead of the host process's stdout / stderr
        if out is None:
            out = self.out_path = tempfile.NamedTemporaryFile(prefix="lqcov_out_", suffix=".tsv", delete=False).name
            self._tmp.append(out)
        if err is None:
            err = self._err_path = tempfile.NamedTemporaryFile(prefix="lqcov_err_", suffix=".log", delete=False).name
            self._tmp.append(err)

        def work():
            try:
                with _device_lock(self.device):
                    self._rc = run_argv(args, out=out, err=err, device=self.device)
            except Exception as e:  # library missing etc.: surface it through get_poll()/get_error()
                self._exc = e
                self._rc = -3

        self._exc = None
        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def get_poll(self):
        if self.subprocess_mode:
            return self.proc.poll() if self.proc else None
        if self._thread is None:
            return None
        if self._thread.is_alive():
            return None
        return self._rc

    def wait(self) -> int:
        if self.subprocess_mode:
            return self.proc.wait()
        self._thread.join()
        if getattr(self, "_exc", None):
            raise self._exc
        return self._rc

    def get_error(self) -> str:
        if self._err_path and os.path.exists(self._err_path):
            lines = [l for l in open(self._err_path).read().splitlines() if l.startswith("ERROR") or l.startswith("Error")]
            return "\n".join(lines)
        return ""

    def get_pid(self):
        return str(self.proc.pid) if self.proc else str(os.getpid())

    def get_bin_path(self):
        return self.bin_path
