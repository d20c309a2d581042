"""ctypes binding of include/lqcov.h.

The product path: there is no CPU fallback -- if liblqcov.so (the hipcc/gfx950 build) is missing
or no HIP device is present, loading / Engine() raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LqcovError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("lqcov error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    """mirror of lqcov_params (include/lqcov.h)"""
    _fields_ = [
        ("k", C.c_int32), ("w", C.c_int32), ("hpc", C.c_int32),
        ("batch_size", C.c_uint64), ("idx_mini_batch", C.c_int32),
        ("max_gap", C.c_int32), ("min_cnt", C.c_int32), ("min_chain_score", C.c_int32),
        ("min_score_med", C.c_int32), ("min_score_good", C.c_int32), ("max_chain_skip", C.c_int32),
        ("bw", C.c_int32), ("max_overhang", C.c_int32), ("min_ovlp", C.c_int32), ("min_coverage", C.c_int32),
        ("min_ratio", C.c_double), ("mid_occ_frac", C.c_float),
        ("no_self", C.c_int32), ("ava", C.c_int32), ("filter_flag", C.c_int32), ("n_threads", C.c_int32),
    ]


class Row(C.Structure):
    _fields_ = [
        ("lambda_", C.c_uint64), ("lambda2", C.c_uint64), ("qual_psum", C.c_double),
        ("qlen", C.c_uint32), ("n_mini", C.c_uint32), ("n_match", C.c_uint32), ("avg_k", C.c_float),
        ("reg_off", C.c_uint32), ("n_reg", C.c_uint32), ("mreg_off", C.c_uint32), ("n_mreg", C.c_uint32),
        ("has_qual", C.c_uint32), ("flags", C.c_uint32),
    ]


class StageTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("total_ms", C.c_double), ("launches", C.c_uint64), ("algo_bytes", C.c_uint64)]


def library_path() -> str:
    return os.environ.get("LQCOV_LIBRARY", os.path.join(_HERE, "liblqcov.so"))


LQCOV_EOF = -100


def load_library(path: Optional[str] = None):
    """Load liblqcov.so (the gfx950 build).  Raises OSError if it has not been built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or library_path()
    if not os.path.exists(p):
        raise OSError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    lib = C.CDLL(p)
    H = C.c_void_p
    u8p, u64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
    sig = {
        "lqcov_abi_version": (C.c_int, []),
        "lqcov_params_default": (None, [C.POINTER(Params)]),
        "lqcov_parse_args": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(Params), C.POINTER(C.c_char_p),
                                       C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t]),
        "lqcov_main": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.c_int]),
        "lqcov_create": (H, [C.POINTER(Params), C.c_int]),
        "lqcov_destroy": (None, [H]),
        "lqcov_last_error": (C.c_char_p, [H]),
        "lqcov_set_profiling": (C.c_int, [H, C.c_int]),
        "lqcov_set_profiling_only": (C.c_int, [H, C.c_char_p]),
        "lqcov_set_debug": (C.c_int, [H, C.c_uint]),
        "lqcov_get_stage_times": (C.c_int, [H, C.POINTER(StageTime), C.c_int]),
        "lqcov_set_queries": (C.c_int, [H, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "lqcov_part_begin": (C.c_int, [H]),
        "lqcov_part_add_targets": (C.c_int, [H, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "lqcov_packed_chunks": (C.c_uint64, [C.c_uint32, C.c_void_p]),
        "lqcov_pack_reads": (C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
        "lqcov_host_alloc": (C.c_void_p, [C.c_size_t]),
        "lqcov_host_free": (None, [C.c_void_p]),
        "lqcov_part_add_packed": (C.c_int, [H, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "lqcov_part_add_packed_shares_dev": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
        "lqcov_part_clear": (C.c_int, [H, C.c_int]),
        "lqcov_part_build": (C.c_int, [H, C.c_int]),
        "lqcov_part_sketch": (C.c_int, [H, C.c_int]),
        "lqcov_part_map": (C.c_int, [H, C.c_int]),
        "lqcov_part_release": (C.c_int, [H, C.c_int]),
        "lqcov_reset": (C.c_int, [H]),
        "lqcov_sync": (C.c_int, [H]),
        "lqcov_workspace_trim": (C.c_int, [H]),
        "lqcov_finish": (C.c_int, [H]),
        "lqcov_n_queries": (C.c_int, [H]),
        "lqcov_query_order": (C.c_int, [H, C.c_void_p, C.c_uint32]),
        "lqcov_get_rows": (C.c_int, [H, C.POINTER(Row), C.c_uint32]),
        "lqcov_get_regions": (C.c_int, [H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
        "lqcov_reserve_hbm": (C.c_int, [H, C.c_uint64]),
        "lqcov_format_rows": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_char_p]),
        "lqcov_write_table": (C.c_int, [H, C.c_char_p]),
        "lqcov_run_files": (C.c_int, [H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
        "lqcov_run_files_ex": (C.c_int, [H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
        "lqcov_part_dump": (C.c_int, [H, C.c_int, C.c_char_p, C.c_int]),
        "lqcov_part_load": (C.c_int, [H, C.c_char_p, u64p]),
        "lqcov_mid_occ": (C.c_int32, [H]),
        "lqcov_part_n_minimizers": (C.c_uint64, [H, C.c_int]),
        "lqcov_part_n_keys": (C.c_uint64, [H, C.c_int]),
        "lqcov_last_n_anchors": (C.c_uint64, [H]),
        "lqcov_map_stats": (None, [H, u64p]),
        "lqcov_packed_ambiguous_reads": (C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
        "lqcov_tie_reasons": (None, [H, u64p]),
        "lqcov_fastx_digest": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_uint64, u64p]),
        "lqcov_get_query_minimizers": (C.c_int, [H, C.c_void_p, C.c_void_p, u64p]),
        "lqcov_get_part_minimizers": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, u64p]),
        "lqcov_get_chains": (C.c_int, [H, C.c_void_p, C.c_uint64, u64p]),
        "lqcov_sat_record_bytes": (C.c_uint32, []),
        "lqcov_counter_max": (C.c_uint32, [H]),
        "lqcov_counter_offsets": (C.c_int, [H, C.c_void_p]),
        "lqcov_part_sat_records": (C.c_int, [H, C.c_int, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p]),
        "lqcov_sat_replay": (C.c_int, [H, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
        "lqcov_accum_set_replayed": (C.c_int, [H, C.c_uint32, C.c_void_p, C.c_uint64]),
        "lqcov_debug_sort_pairs": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_int]),
        "lqcov_debug_scan": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_uint64]),
        "lqcov_part_minimizers_dev": (C.c_int, [H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), u64p]),
        "lqcov_part_minimizers_export_dev": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]),
        "lqcov_set_distributed": (C.c_int, [H, C.c_int]),
        "lqcov_set_mid_occ": (C.c_int, [H, C.c_int32]),
        "lqcov_accum_sizes": (C.c_int, [H, C.POINTER(C.c_uint32), u64p, C.POINTER(C.c_uint32)]),
        "lqcov_accum_export_dev": (C.c_int, [H] + [C.c_void_p] * 7),
        "lqcov_accum_import_dev": (C.c_int, [H] + [C.c_void_p] * 6 + [C.c_uint32]),
        "lqcov_part_build_from_minimizers_dev": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                                           C.c_void_p, C.c_void_p, C.c_void_p]),
        "lqcov_part_build_from_minimizer_shares_dev": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                                                 C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here == the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    if lib.lqcov_abi_version() != 1:
        raise OSError("liblqcov.so ABI version mismatch")
    lib._sig_names = sorted(sig)
    if path is None:
        _LIB = lib
    return lib


def default_params(**kw) -> Params:
    lib = load_library()
    p = Params()
    lib.lqcov_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def parse_args(argv: Sequence[str]) -> Tuple[Params, Optional[str], Optional[str]]:
    """argv WITHOUT the program name, e.g. ['-Y','-l','0',...,'all.fq','sub.fq'] (longQC.py:440-445)."""
    lib = load_library()
    full = [b"minimap2-coverage"] + [str(a).encode() for a in argv]
    arr = (C.c_char_p * len(full))(*full)
    p = Params()
    t, q, d = C.c_char_p(), C.c_char_p(), C.c_char_p()
    err = C.create_string_buffer(256)
    rc = lib.lqcov_parse_args(len(full), arr, C.byref(p), C.byref(t), C.byref(q), C.byref(d), err, 256)
    if rc:
        raise LqcovError(rc, err.value.decode())
    return p, (t.value.decode() if t.value else None), (q.value.decode() if q.value else None)


def format_rows(lib, filter_flag: int, rows: np.ndarray, regs: np.ndarray, mregs: np.ndarray, names: Sequence[str]) -> str:
    """the table text of binary rows (uint8[n, sizeof(lqcov_row)], reg_off / mreg_off pointing into regs / mregs): lqcov_format_rows"""
    import tempfile
    rows = np.ascontiguousarray(rows, dtype=np.uint8); regs = np.ascontiguousarray(regs, dtype=np.uint32); mregs = np.ascontiguousarray(mregs, dtype=np.uint32)
    nb, noff = _names(names)
    with tempfile.NamedTemporaryFile("r", suffix=".tsv") as f:
        rc = lib.lqcov_format_rows(int(filter_flag), rows.ctypes.data, rows.shape[0], regs.ctypes.data, mregs.ctypes.data, nb, noff.ctypes.data, f.name.encode())
        if rc != 0:
            raise LqcovError(rc, "lqcov_format_rows failed")
        return open(f.name).read()


def _flat(seqs: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([int(s.shape[0]) for s in seqs], dtype=np.uint64)
    flat = np.concatenate(seqs).astype(np.uint8, copy=False) if len(seqs) else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(flat), off


def _names(names: Sequence[str]) -> Tuple[bytes, np.ndarray]:
    off = np.zeros(len(names) + 1, dtype=np.uint64)
    parts = []
    pos = 0
    for i, n in enumerate(names):
        b = n.encode() + b"\0"
        parts.append(b)
        pos += len(b)
        off[i + 1] = pos
    return b"".join(parts), off


def encode_names(names: Sequence[str]):
    """names as the C ABI takes them (NUL-terminated blob + offsets): encode once when the same list is passed repeatedly"""
    return _names(names)


class PackedReads:
    """Reads 2-bit packed on the host into page-locked buffers (lqcov_pack_reads / lqcov_host_alloc): what the parser
    thread of lqcov_run_files produces, for callers that hold the reads in memory (bench.py, in-memory sampleqc)."""

    def __init__(self, flat: np.ndarray, off: np.ndarray, names: Sequence[str], threads: int = 0, lib=None):
        self.lib = lib or load_library()
        off = np.ascontiguousarray(off, dtype=np.uint64)
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        n = int(off.shape[0] - 1)
        self.n = n
        self.lens = np.ascontiguousarray(np.diff(off).astype(np.uint32))
        self.coff = np.zeros(n + 1, dtype=np.uint64)
        self.coff[1:] = np.cumsum((self.lens.astype(np.uint64) + 127) // 128, dtype=np.uint64)
        self.n_chunks = int(self.coff[-1])
        assert self.n_chunks == int(self.lib.lqcov_packed_chunks(n, off.ctypes.data))
        self.n_bases = int(off[-1] - off[0])
        self.codes_ptr = self.lib.lqcov_host_alloc(max(self.n_chunks, 1) * 32)
        self.amb_ptr = self.lib.lqcov_host_alloc(max(self.n_chunks, 1) * 16)
        if not self.codes_ptr or not self.amb_ptr:
            raise MemoryError("lqcov_host_alloc failed")
        rc = self.lib.lqcov_pack_reads(n, flat.ctypes.data, off.ctypes.data, self.codes_ptr, self.amb_ptr, threads)
        if rc:
            raise LqcovError(rc, "lqcov_pack_reads failed")
        self.names_blob, self.name_off = _names(names)
        # which reads hold an ambiguous base: a range without one is uploaded as codes alone (lqcov_part_add_packed with amb == NULL)
        self.has_amb = np.zeros(max(n, 1), dtype=np.uint8)
        rc = self.lib.lqcov_packed_ambiguous_reads(n, self.amb_ptr, self.lens.ctypes.data, self.has_amb.ctypes.data)
        if rc:
            raise LqcovError(rc, "lqcov_packed_ambiguous_reads failed")
        self.cum_amb = np.concatenate([[0], np.cumsum(self.has_amb[:n], dtype=np.int64)])

    def any_ambiguous(self, lo: int, hi: int) -> bool:
        return bool(self.cum_amb[hi] - self.cum_amb[lo])

    def __len__(self):
        return self.n

    def close(self):
        if getattr(self, "codes_ptr", None):
            self.lib.lqcov_host_free(self.codes_ptr); self.codes_ptr = None
        if getattr(self, "amb_ptr", None):
            self.lib.lqcov_host_free(self.amb_ptr); self.amb_ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One handle == one HIP device + stream (include/lqcov.h level 2)."""

    def __init__(self, params: Optional[Params] = None, device: int = 0, lib=None):
        self.lib = lib or load_library()
        self.params = params or default_params(no_self=1)
        self.h = self.lib.lqcov_create(C.byref(self.params), device)
        if not self.h:
            raise LqcovError(-3, self.lib.lqcov_last_error(None).decode() or "lqcov_create failed (no HIP device?)")
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.lqcov_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int) -> int:
        if rc < 0:
            raise LqcovError(rc, self.lib.lqcov_last_error(self.h).decode())
        return rc

    # -- data in --
    def set_queries(self, names: Sequence[str], seqs: Sequence[np.ndarray], quals: Optional[Sequence[np.ndarray]] = None):
        flat, off = _flat(seqs)
        nb, noff = _names(names)
        q = None
        if quals is not None:
            q, _ = _flat(quals)
        self._ck(self.lib.lqcov_set_queries(self.h, len(seqs), flat.ctypes.data, off.ctypes.data,
                                            q.ctypes.data if q is not None else None, nb, noff.ctypes.data))

    def part_begin(self) -> int:
        return self._ck(self.lib.lqcov_part_begin(self.h))

    def part_add_targets(self, part: int, names: Sequence[str], seqs: Sequence[np.ndarray]):
        flat, off = _flat(seqs)
        nb, noff = _names(names)
        self._ck(self.lib.lqcov_part_add_targets(self.h, part, len(seqs), flat.ctypes.data, off.ctypes.data, nb, noff.ctypes.data))

    def part_add_packed(self, part: int, packed: "PackedReads", lo: int = 0, hi: Optional[int] = None):
        """reads [lo, hi) of a PackedReads (2-bit packed on the host, page-locked): H2D only, no device-side packing"""
        hi = len(packed) if hi is None else hi
        c0 = int(packed.coff[lo])
        lens = np.ascontiguousarray(packed.lens[lo:hi])
        noff = np.ascontiguousarray(packed.name_off[lo:hi + 1] - packed.name_off[lo])
        blob = packed.names_blob[int(packed.name_off[lo]):int(packed.name_off[hi])]
        send_amb = packed.any_ambiguous(lo, hi) or os.environ.get("LQCOV_UPLOAD_AMB") == "1"     # (no N in these reads: 0.25 instead of 0.375 B per base cross PCIe)
        self._ck(self.lib.lqcov_part_add_packed(self.h, part, hi - lo, packed.codes_ptr + c0 * 32, (packed.amb_ptr + c0 * 16) if send_amb else None,
                                                lens.ctypes.data, blob, noff.ctypes.data))

    def part_add_packed_shares_dev(self, part: int, codes_ptr: int, amb_ptr: int, stride_chunks: int, share_chunks, lens: np.ndarray, names: Sequence[str]):
        """the packed reads of a whole part from device memory: share i (share_chunks[i] chunks of 128 bases) at chunk i * stride_chunks of
        codes_ptr (32 B per chunk) / amb_ptr (16 B per chunk), copied back to back; lens / names: every read of the part in order"""
        sc = np.ascontiguousarray(share_chunks, dtype=np.uint64)
        ln = np.ascontiguousarray(lens, dtype=np.uint32)
        nb, noff = names if isinstance(names, tuple) else _names(names)
        self._ck(self.lib.lqcov_part_add_packed_shares_dev(self.h, part, codes_ptr, amb_ptr, int(stride_chunks), len(sc), sc.ctypes.data, len(ln), ln.ctypes.data, nb, noff.ctypes.data))

    def part_clear(self, part: int):
        self._ck(self.lib.lqcov_part_clear(self.h, part))

    def part_build(self, part: int):
        self._ck(self.lib.lqcov_part_build(self.h, part))

    def part_sketch(self, part: int):
        self._ck(self.lib.lqcov_part_sketch(self.h, part))

    def part_map(self, part: int):
        self._ck(self.lib.lqcov_part_map(self.h, part))

    def part_release(self, part: int):
        self._ck(self.lib.lqcov_part_release(self.h, part))

    def reset(self):
        self._ck(self.lib.lqcov_reset(self.h))

    def sync(self):
        self._ck(self.lib.lqcov_sync(self.h))

    def reserve_hbm(self, n_bytes: int):
        """HBM the mapping lanes leave free when they size their work space (a second part that is built while one is mapped)"""
        self._ck(self.lib.lqcov_reserve_hbm(self.h, int(n_bytes)))

    def workspace_trim(self):
        self._ck(self.lib.lqcov_workspace_trim(self.h))

    def finish(self):
        self._ck(self.lib.lqcov_finish(self.h))

    def run_files(self, target: str, query: Optional[str], out: Optional[str] = None, err: Optional[str] = None,
                  dump: Optional[str] = None):
        """target: FASTA/Q(.gz) or an .mmi index (the reference's or ours); dump: also write the index there (-d)"""
        enc = lambda v: v.encode() if v else None
        self._ck(self.lib.lqcov_run_files_ex(self.h, target.encode(), enc(query), enc(dump), enc(out), enc(err)))

    def part_dump(self, part: int, path: str, append: bool = False):
        self._ck(self.lib.lqcov_part_dump(self.h, part, path.encode(), 1 if append else 0))

    def part_load(self, path: str, offset: int = 0) -> Tuple[Optional[int], int]:
        """-> (part id or None at the end of the file, offset of the next part)"""
        off = C.c_uint64(offset)
        rc = self.lib.lqcov_part_load(self.h, path.encode(), C.byref(off))
        if rc == LQCOV_EOF:
            return None, off.value
        return self._ck(rc), off.value

    # -- results --
    def rows(self) -> List[dict]:
        n = self.lib.lqcov_n_queries(self.h)
        arr = (Row * max(n, 1))()
        self._ck(self.lib.lqcov_get_rows(self.h, arr, n))
        rp, mp = C.c_void_p(), C.c_void_p()
        nr, nm = C.c_uint32(), C.c_uint32()
        self._ck(self.lib.lqcov_get_regions(self.h, C.byref(rp), C.byref(nr), C.byref(mp), C.byref(nm)))
        regs = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint32)), shape=(nr.value, 2)).copy() if nr.value else np.zeros((0, 2), np.uint32)
        mregs = np.ctypeslib.as_array(C.cast(mp, C.POINTER(C.c_uint32)), shape=(nm.value, 2)).copy() if nm.value else np.zeros((0, 2), np.uint32)
        out = []
        for i in range(n):
            r = arr[i]
            out.append(dict(lambda_=r.lambda_, lambda2=r.lambda2, qual_psum=r.qual_psum, qlen=r.qlen, n_mini=r.n_mini,
                            n_match=r.n_match, avg_k=r.avg_k, flags=r.flags,
                            regs=regs[r.reg_off:r.reg_off + r.n_reg].tolist(),
                            mregs=mregs[r.mreg_off:r.mreg_off + r.n_mreg].tolist()))
        return out

    def rows_binary(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """the finished rows as they are (one `lqcov_row` per query, caller's order, as uint8[n, sizeof(lqcov_row)]) and the two
        region pools (uint32[n, 2]): what the ranks of a multi-GPU run exchange"""
        n = self.lib.lqcov_n_queries(self.h)
        arr = (Row * max(n, 1))()
        self._ck(self.lib.lqcov_get_rows(self.h, arr, n))
        rows = np.frombuffer(arr, dtype=np.uint8).reshape(max(n, 1), C.sizeof(Row))[:n].copy()
        rp, mp = C.c_void_p(), C.c_void_p()
        nr, nm = C.c_uint32(), C.c_uint32()
        self._ck(self.lib.lqcov_get_regions(self.h, C.byref(rp), C.byref(nr), C.byref(mp), C.byref(nm)))
        regs = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint32)), shape=(nr.value, 2)).copy() if nr.value else np.zeros((0, 2), np.uint32)
        mregs = np.ctypeslib.as_array(C.cast(mp, C.POINTER(C.c_uint32)), shape=(nm.value, 2)).copy() if nm.value else np.zeros((0, 2), np.uint32)
        return rows, regs, mregs

    def write_table(self, path: str):
        self._ck(self.lib.lqcov_write_table(self.h, path.encode()))

    def table_text(self) -> str:
        import tempfile
        with tempfile.NamedTemporaryFile("r", suffix=".tsv") as f:
            self.write_table(f.name)
            return open(f.name).read()

    # -- inspection --
    @property
    def mid_occ(self) -> int:
        return int(self.lib.lqcov_mid_occ(self.h))

    @property
    def last_n_anchors(self) -> int:
        return int(self.lib.lqcov_last_n_anchors(self.h))

    def map_stats(self) -> dict:
        """anchors written against the last part; since reset(): runs / queries / anchors that needed klib's own order"""
        a = (C.c_uint64 * 4)()
        self.lib.lqcov_map_stats(self.h, a)
        r = (C.c_uint64 * 6)()
        self.lib.lqcov_tie_reasons(self.h, r)
        why = dict(zip(("skip_pending_at_group", "member_counts_as_skip", "top_score_twice", "scan_broke_off", "equal_peaks_in_backtrack", "other"), (int(v) for v in r)))
        return {"last_written": int(a[0]), "klib_runs": int(a[1]), "klib_queries": int(a[2]), "klib_anchors": int(a[3]), "reasons": why}

    def part_n_minimizers(self, part: int) -> int:
        return int(self.lib.lqcov_part_n_minimizers(self.h, part))

    def part_n_keys(self, part: int) -> int:
        return int(self.lib.lqcov_part_n_keys(self.h, part))

    def _minimizers(self, fn, *pre) -> Tuple[np.ndarray, np.ndarray]:
        n = C.c_uint64()
        self._ck(fn(self.h, *pre, None, None, C.byref(n)))
        nq = self.lib.lqcov_n_queries(self.h) if not pre else None
        xy = np.zeros((n.value, 2), dtype=np.uint64)
        return xy, n

    def query_minimizers(self) -> Tuple[np.ndarray, np.ndarray]:
        n = C.c_uint64()
        self._ck(self.lib.lqcov_get_query_minimizers(self.h, None, None, C.byref(n)))
        xy = np.zeros((n.value, 2), dtype=np.uint64)
        off = np.zeros(self.lib.lqcov_n_queries(self.h) + 1, dtype=np.uint64)
        self._ck(self.lib.lqcov_get_query_minimizers(self.h, xy.ctypes.data, off.ctypes.data, C.byref(n)))
        return xy, off

    def part_minimizers(self, part: int, n_reads: int) -> Tuple[np.ndarray, np.ndarray]:
        n = C.c_uint64()
        self._ck(self.lib.lqcov_get_part_minimizers(self.h, part, None, None, C.byref(n)))
        xy = np.zeros((n.value, 2), dtype=np.uint64)
        off = np.zeros(n_reads + 1, dtype=np.uint64)
        self._ck(self.lib.lqcov_get_part_minimizers(self.h, part, xy.ctypes.data, off.ctypes.data, C.byref(n)))
        return xy, off

    def set_debug(self, flags: int):
        self._ck(self.lib.lqcov_set_debug(self.h, flags))

    def chains(self) -> np.ndarray:
        """(n,9) int32: query, rid, rev, score, cnt, qs, qe, rs, re of the last part_map (needs set_debug(1))."""
        n = C.c_uint64()
        self._ck(self.lib.lqcov_get_chains(self.h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 9), dtype=np.int32)
        self._ck(self.lib.lqcov_get_chains(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out

    # -- saturated counters with the parts spread over ranks (multigpu.PartRunner) --
    def counter_max(self) -> int:
        return int(self.lib.lqcov_counter_max(self.h))

    def counter_offsets(self) -> np.ndarray:
        """off[q] .. off[q + 1]: the counters of query q (engine order) in the exported counter array"""
        off = np.zeros(self.lib.lqcov_n_queries(self.h) + 1, dtype=np.uint64)
        self._ck(self.lib.lqcov_counter_offsets(self.h, off.ctypes.data))
        return off

    def part_sat_records(self, part: int, query: int) -> Tuple[np.ndarray, np.ndarray]:
        """every kept chain of `query` (engine order) against the mapped part: (records as uint8[n, record bytes], counter index pool)"""
        n = (C.c_uint64 * 2)()
        self._ck(self.lib.lqcov_part_sat_records(self.h, part, query, None, 0, None, 0, n))
        rb = int(self.lib.lqcov_sat_record_bytes())
        recs = np.zeros((max(int(n[0]), 1), rb), dtype=np.uint8); at = np.zeros(max(int(n[1]), 1), dtype=np.uint32)
        self._ck(self.lib.lqcov_part_sat_records(self.h, part, query, recs.ctypes.data, int(n[0]), at.ctypes.data, int(n[1]), n))
        return recs[:int(n[0])], at[:int(n[1])]

    def sat_replay(self, query: int, recs: np.ndarray, at: np.ndarray, counters: np.ndarray) -> np.ndarray:
        recs = np.ascontiguousarray(recs, dtype=np.uint8); at = np.ascontiguousarray(at, dtype=np.uint32)
        c = np.ascontiguousarray(counters, dtype=np.uint32).copy()
        self._ck(self.lib.lqcov_sat_replay(self.h, query, recs.ctypes.data, recs.shape[0], at.ctypes.data, at.shape[0], c.ctypes.data, c.shape[0]))
        return c

    def accum_set_replayed(self, query: int, counters: np.ndarray):
        c = np.ascontiguousarray(counters, dtype=np.uint32)
        self._ck(self.lib.lqcov_accum_set_replayed(self.h, query, c.ctypes.data, c.shape[0]))

    def debug_sort_pairs(self, keys: np.ndarray, vals: Optional[np.ndarray], bits: int, key_bytes: int = 4):
        """tests: the engine's stable radix sort (kernels_isort.hpp) on host arrays; returns (keys, vals) sorted"""
        k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
        v = None if vals is None else np.ascontiguousarray(vals, dtype=np.uint64).copy()
        self._ck(self.lib.lqcov_debug_sort_pairs(self.h, k.ctypes.data, None if v is None else v.ctypes.data, k.shape[0], bits, key_bytes))
        return k, v

    def debug_scan(self, counts: np.ndarray) -> np.ndarray:
        """tests: the engine's exclusive scan of u32 counts (u64 sums)"""
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        out = np.zeros(c.shape[0], dtype=np.uint64)
        self._ck(self.lib.lqcov_debug_scan(self.h, c.ctypes.data, out.ctypes.data, c.shape[0]))
        return out

    def set_profiling(self, on, only: Optional[str] = None):
        """0 off; 1 waits for every kernel (exclusive times, lanes in turn); 2 events only.  only: time just that stage"""
        self._ck(self.lib.lqcov_set_profiling_only(self.h, only.encode() if only else None))
        self._ck(self.lib.lqcov_set_profiling(self.h, int(on)))

    def stage_times(self) -> List[dict]:
        arr = (StageTime * 160)()
        n = self._ck(self.lib.lqcov_get_stage_times(self.h, arr, 160))
        return [dict(name=arr[i].name.decode(), total_ms=arr[i].total_ms, launches=arr[i].launches, algo_bytes=arr[i].algo_bytes)
                for i in range(n)]

    # -- multi-GPU plumbing (pointers are device pointers, e.g. torch tensors' data_ptr()) --
    def set_distributed(self, on: bool):
        self._ck(self.lib.lqcov_set_distributed(self.h, 1 if on else 0))

    def set_mid_occ(self, v: int):
        self._ck(self.lib.lqcov_set_mid_occ(self.h, int(v)))

    def query_order(self) -> np.ndarray:
        """perm[i] = the caller's index of the i-th query in the engine's own (longest first) order"""
        n = self.lib.lqcov_n_queries(self.h)
        perm = np.zeros(max(n, 1), dtype=np.uint32)
        self._ck(self.lib.lqcov_query_order(self.h, perm.ctypes.data, n))
        return perm[:n]

    def accum_sizes(self) -> Tuple[int, int, int]:
        a, b, c = C.c_uint32(), C.c_uint64(), C.c_uint32()
        self._ck(self.lib.lqcov_accum_sizes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def accum_export(self, lam, lam2, avgk, flags, cnts, owner, ivl):
        self._ck(self.lib.lqcov_accum_export_dev(self.h, lam, lam2, avgk, flags, cnts, owner, ivl))

    def accum_import(self, lam, lam2, avgk, flags, cnts, ivl, n_ivl: int):
        self._ck(self.lib.lqcov_accum_import_dev(self.h, lam, lam2, avgk, flags, cnts, ivl, n_ivl))

    def part_minimizers_dev(self, part: int) -> Tuple[int, int, int]:
        x, y, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._ck(self.lib.lqcov_part_minimizers_dev(self.h, part, C.byref(x), C.byref(y), C.byref(n)))
        return x.value or 0, y.value or 0, n.value

    def part_minimizers_export(self, part: int, x_ptr: int, y_ptr: int, cap: int, rid_base: int):
        self._ck(self.lib.lqcov_part_minimizers_export_dev(self.h, part, x_ptr, y_ptr, cap, rid_base))

    def part_build_from_minimizer_shares_dev(self, part: int, x_ptr: int, y_ptr: int, stride: int, share_n, target_len: np.ndarray, names: Sequence[str]):
        """the receive buffers of an all-gather: share i = share_n[i] entries from word i * stride"""
        tl = np.ascontiguousarray(target_len, dtype=np.uint32)
        sn = np.ascontiguousarray(share_n, dtype=np.uint64)
        nb, noff = names if isinstance(names, tuple) else _names(names)
        self._ck(self.lib.lqcov_part_build_from_minimizer_shares_dev(self.h, part, x_ptr, y_ptr, stride, len(sn), sn.ctypes.data, len(tl), tl.ctypes.data, nb, noff.ctypes.data))

    def part_build_from_minimizers_dev(self, part: int, x_ptr: int, y_ptr: int, n: int, target_len: np.ndarray, names: Sequence[str]):
        tl = np.ascontiguousarray(target_len, dtype=np.uint32)
        nb, noff = names if isinstance(names, tuple) else _names(names)      # (blob, offsets) from encode_names() skips the encoding
        self._ck(self.lib.lqcov_part_build_from_minimizers_dev(self.h, part, x_ptr, y_ptr, n, len(tl), tl.ctypes.data, nb, noff.ctypes.data))
