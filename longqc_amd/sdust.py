"""Host side of the low-complexity table: the counterpart of lq_mask.py's use of the
reference's `sdust` binary (lq_mask.py:17-23 `_sdust`, :99-121 `submit_sdust` / `close_pool`), over the C ABI of
include/lqcov.h (lqsdust_main, lqsdust_reads).  No CPU fallback: without liblqcov.so or a HIP device the calls raise."""
import ctypes as C
import math
import os
from typing import List, Optional, Sequence

import numpy as np

from . import api


def _lib(lib=None):
    lib = lib or api.load_library()
    if not getattr(lib, "_lqsdust_bound", False):
        lib.lqsdust_main.restype = C.c_int
        lib.lqsdust_main.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.c_int]
        lib.lqsdust_reads.restype = C.c_int
        lib.lqsdust_reads.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        lib._lqsdust_bound = True
    return lib


def run_sdust(fin: str, fout: str, device: int = 0, w: Optional[int] = None, t: Optional[int] = None, lib=None) -> None:
    """== lq_mask._sdust(psdust, fin, fout): `sdust <fin>` with stdout -> fout (lq_mask.py:17-23)"""
    lib = _lib(lib)
    argv = [b"sdust"]
    if w is not None:
        argv += [b"-w", str(w).encode()]
    if t is not None:
        argv += [b"-t", str(t).encode()]
    argv.append(fin.encode())
    arr = (C.c_char_p * len(argv))(*argv)
    err = fout + ".stderr"
    rc = lib.lqsdust_main(len(argv), arr, fout.encode(), err.encode(), device)
    msg = open(err).read() if os.path.exists(err) else ""
    if os.path.exists(err):
        os.remove(err)
    if rc == 0 and msg and os.environ.get("LQCOV_TIMING"):
        import sys
        sys.stderr.write(msg)
    if rc != 0:
        raise api.LqcovError(rc if rc < 0 else -2, msg.strip() or "sdust failed")


def sdust_rows(names: Sequence[str], seqs: Sequence[np.ndarray], quals: Optional[Sequence[Optional[np.ndarray]]] = None,
               device: int = 0, w: int = 64, t: int = 20, lib=None) -> List[str]:
    """Reads in memory (no temporary FASTQ as in lq_mask.py:108-114) -> the rows `sdust` would print for them."""
    lib = _lib(lib)
    n = len(seqs)
    off = np.zeros(n + 1, dtype=np.uint64)
    for i, s in enumerate(seqs):
        off[i + 1] = off[i] + s.shape[0]
    flat = np.concatenate([np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]) if n else np.zeros(0, np.uint8)
    qflat = None
    if quals is not None and any(q is not None and q.shape[0] for q in quals):
        qflat = np.zeros(int(off[n]), dtype=np.uint8)
        for i, q in enumerate(quals):
            if q is not None and q.shape[0]:
                qflat[int(off[i]):int(off[i + 1])] = q
    masked = np.zeros(max(n, 1), dtype=np.uint32)
    psum = np.zeros(max(n, 1), dtype=np.float64)
    qv = np.zeros(max(n, 1), dtype=np.uint32)
    err = C.create_string_buffer(512)
    rc = lib.lqsdust_reads(device, n, flat.ctypes.data if n else None, off.ctypes.data, qflat.ctypes.data if qflat is not None else None,
                           w, t, masked.ctypes.data, psum.ctypes.data, qv.ctypes.data, err, 512)
    if rc != 0:
        raise api.LqcovError(rc, err.value.decode())
    rows = []
    for i in range(n):
        ln = int(off[i + 1] - off[i])
        has_q = qflat is not None and ln > 0 and qflat[int(off[i])] != 0
        frac = _c_div(float(masked[i]), ln)
        mq = -10 * _c_log10(_c_div(float(psum[i]) if has_q else 0.0, ln if has_q else 0))
        rows.append("%s\t%d\t%d\t%s\t%s\t%d" % (names[i], int(masked[i]), ln, _c_fmt3(frac), _c_fmt3(mq), int(qv[i])))
    return rows


class LqMaskMI355X:
    """Counterpart of lq_mask.LqMask for the chunk loop of longQC.py (lq_mask.py:25-41, 99-121): submit_sdust(reads,
    chunk_n) per chunk, close_pool() at the end, get_outfile_path() -> analysis table `longqc_sdust<suffix>.txt` with one
    row per read in submission order.  Reads are LongQC's [name, seq, qual, ...] records (str or bytes); no temporary
    FASTQ files, no process pool: every chunk is one call into the device.  The plots of LqMask are out of scope."""

    def __init__(self, work_dir: str, suffix: Optional[str] = None, device: int = 0, lib=None):
        self.suffix = "_" + suffix if suffix else ""
        os.makedirs(work_dir, exist_ok=True)
        self.wdir, self.device, self.lib = work_dir, device, lib
        self.outf = os.path.join(work_dir, "longqc_sdust" + self.suffix + ".txt")
        self._chunks = {}

    @staticmethod
    def _arr(v):
        if v is None:
            return None
        if isinstance(v, str):
            v = v.encode()
        return np.frombuffer(bytes(v), dtype=np.uint8)

    def submit_sdust(self, reads, chunk_n):
        names = [r[0].decode() if isinstance(r[0], (bytes, bytearray)) else str(r[0]) for r in reads]
        seqs = [self._arr(r[1]) for r in reads]
        quals = [self._arr(r[2]) if len(r) > 2 and r[2] else None for r in reads]
        self._chunks[chunk_n] = sdust_rows(names, seqs, quals, device=self.device, lib=self.lib)

    def close_pool(self):
        with open(self.outf, "w") as out:                      # lq_mask.py:83-88 concatenates the chunk tables in submission order
            for k in self._chunks:
                for row in self._chunks[k]:
                    out.write(row + "\n")
        self._chunks = {}

    def get_outfile_path(self):
        return self.outf


def _c_div(a: float, b: int) -> float:
    return a / b if b else math.nan                       # C: 0.0 / 0 (x86: the default NaN, printed "-nan")


def _c_log10(x: float) -> float:
    if x != x:
        return x
    return math.log10(x) if x > 0 else (-math.inf if x == 0 else math.nan)


def _c_fmt3(x: float) -> str:
    if x != x:
        return "-nan"                                     # x86 keeps the sign of the default NaN through log10 and the product
    if x in (math.inf, -math.inf):
        return "inf" if x > 0 else "-inf"
    return "%.3f" % x
