"""Host logic (FASTA/Q stream, part boundaries, error behaviour, LqExec counterpart) -- exercised
through the emulator build so it runs without a GPU."""
import gzip
import os

import numpy as np
import pytest

from tests import oracle_bind
from tests.conftest import GOLDEN, ROOT, read_gz
from tests.helpers import ONT, run_main


def _write(path, data: bytes):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wb") as f:
        f.write(data)


def test_input_dialects_agree_with_reference_semantics(emu_lib, tmp_path):
    """gz, CRLF, multi-line FASTA, lower case, U, comments in headers, blank lines (kseq.h:179-224, bseq.c:61-63)"""
    from tests.helpers import read_fastx
    tn, ts, tq = read_fastx(os.path.join(GOLDEN, "tiny_all.fq.gz"))
    qn, qs, qq = read_fastx(os.path.join(GOLDEN, "tiny_sub.fq.gz"))
    want = read_gz("tiny_ont.table.gz")
    t1 = str(tmp_path / "t.fa")
    rec = []
    for i, (n, s) in enumerate(zip(tn, ts)):
        sb = s.tobytes()
        if i % 3 == 0:
            sb = sb.lower()
        if i % 5 == 0:
            sb = sb.replace(b"T", b"U").replace(b"t", b"u")
        lines = [sb[j:j + 61] for j in range(0, len(sb), 61)]
        rec.append(b">" + n.encode() + b" some comment\tmore\r\n" + b"\r\n".join(lines) + b"\r\n" + (b"\r\n" if i % 7 == 0 else b""))
    _write(t1, b"".join(rec))
    q1 = str(tmp_path / "q.fq.gz")
    _write(q1, b"".join(b"@" + n.encode() + b" c\n" + s.tobytes() + b"\n+" + n.encode() + b"\n" + q.tobytes() + b"\n" for n, s, q in zip(qn, qs, qq)))
    argv = ONT + [t1, q1]
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == want
    assert out == oracle_bind.table(argv)


def test_truncated_quality_ends_the_stream_like_kseq(emu_lib, tmp_path):
    q = str(tmp_path / "q.fq")
    _write(q, b"@a\nACGTACGTACGTAAGGCCTTACGATCGATCGACTAGCTAGCATCGA\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n@b\nACGTACGTAC\n+\nIII\n")
    argv = ONT + [os.path.join(GOLDEN, "tiny_all.fq.gz"), q]
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == oracle_bind.table(argv)
    assert [l.split("\t")[0] for l in out.splitlines()] == ["a"]


def test_lone_header_character_at_the_end_is_no_record(emu_lib, tmp_path):
    """a stream that ends right after a '@' / '>' yields no further record (kseq.h: ks_getuntil2 returns -1 there),
    neither as a query nor as a target"""
    seq = b"ACGTACGTACGTAAGGCCTTACGATCGATCGACTAGCTAGCATCGA"
    q = str(tmp_path / "q.fq")
    _write(q, b"@a\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n@")
    t = str(tmp_path / "t.fa")
    _write(t, b">t0\n" + seq + b"\n>")
    argv = ONT + [t, q]
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert [l.split("\t")[0] for l in out.splitlines()] == ["a"]
    if oracle_bind.have_ref():
        assert out == oracle_bind.ref_table(argv)


def test_unopenable_inputs(emu_lib, tmp_path):
    rc, out, err = run_main(emu_lib, ONT + [str(tmp_path / "missing.fq"), os.path.join(GOLDEN, "tiny_sub.fq.gz")])
    assert rc == 1 and "failed to open file" in err        # minimap2-coverage.c:276-279
    rc, out, err = run_main(emu_lib, ONT + [os.path.join(GOLDEN, "tiny_all.fq.gz"), str(tmp_path / "missing.fq")])
    assert rc != 0 and "failed to open file" in err


def test_bad_flags_exit_one(emu_lib):
    rc, out, err = run_main(emu_lib, ["-k", "12", "a", "b"])
    assert rc == 1 and "Choose either -X" in err            # minimap2-coverage.c:232-234
    rc, out, err = run_main(emu_lib, ["-X", "a", "b"])
    assert rc != 0


def test_empty_query_file_and_empty_target_file(emu_lib, tmp_path):
    e = str(tmp_path / "empty.fq")
    _write(e, b"")
    rc, out, err = run_main(emu_lib, ONT + [os.path.join(GOLDEN, "tiny_all.fq.gz"), e])
    assert rc == 0 and out == ""
    argv = ONT + [e, os.path.join(GOLDEN, "tiny_sub.fq.gz")]
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == oracle_bind.table(argv)


def test_lqexec_counterpart_shape(tmp_path):
    """same call shape as lq_exec.LqExec (exec(*args, out=, err=) / get_poll()); without a GPU the run fails loudly"""
    import torch
    from longqc_amd import LqCovExec
    le = LqCovExec()
    out, err = str(tmp_path / "coverage_out.txt"), str(tmp_path / "coverage_err.txt")
    le.exec(*(ONT + [os.path.join(GOLDEN, "tiny_all.fq.gz"), os.path.join(GOLDEN, "tiny_sub.fq.gz")]), out=out, err=err)
    rc = le.wait()
    assert le.get_poll() == rc
    if torch.cuda.is_available():
        assert rc == 0 and open(out).read() == read_gz("tiny_ont.table.gz")
    else:
        assert rc == -3 and "HIP device" in open(err).read()


# ---- the reader over the mapped file (fastx_mem.hpp) against the streaming reader (fastx.hpp), record for record -----------------
def _digest(lib, path, mode, threads=4, piece=64):
    import ctypes as C
    out = (C.c_uint64 * 5)()
    rc = lib.lqcov_fastx_digest(path.encode(), mode, threads, piece, out)
    assert rc == 0, rc
    return tuple(int(x) for x in out)


def _dialects():
    rng = np.random.default_rng(5)
    A = np.frombuffer(b"ACGTNacgtnUu", dtype=np.uint8)

    def seq(n):
        return A[rng.integers(0, len(A), n)].tobytes()

    def qual(n, first=None):
        q = (33 + rng.integers(0, 60, n)).astype(np.uint8).tobytes()      # '@' (Q31), '+' (Q10), '>' (Q29) all occur, also at line starts
        return (first + q[1:]) if first and n else q
    recs = {}
    # plain four-line FASTQ, quality lines that start with '@', '+' and '>'
    recs["fastq"] = b"".join(b"@r%d desc\n%s\n+\n%s\n" % (i, s, qual(len(s), [b"@", b"+", b">", None][i % 4])) for i, s in ((i, seq(int(rng.integers(1, 300)))) for i in range(400)))
    # CRLF, '+' line repeating the name, blank lines between records, no newline at the end
    recs["crlf"] = b"".join(b"@r%d\tc\r\n%s\r\n+r%d\r\n%s\r\n%s" % (i, s, i, qual(len(s)), b"\r\n" if i % 5 == 0 else b"") for i, s in ((i, seq(int(rng.integers(2, 200)))) for i in range(300)))[:-2]
    # wrapped FASTQ (sequence and quality over several lines): the guessed starts do not hold, the stitcher parses in order
    def wrap(b, w):
        return b"\n".join(b[j:j + w] for j in range(0, len(b), w))
    recs["wrapped"] = b"".join(b"@w%d\n%s\n+\n%s\n" % (i, wrap(s, 50), wrap(qual(len(s), b"@" if i % 3 == 0 else None), 50)) for i, s in ((i, seq(int(rng.integers(1, 400)))) for i in range(200)))
    # FASTA: one line, wrapped, CRLF, comments, empty sequences, a lone '>' at the end
    recs["fasta"] = b"".join(b">f%d some comment\n%s\n" % (i, wrap(seq(int(rng.integers(0, 500))), [60, 1000, 7][i % 3])) for i in range(300)) + b">"
    recs["fasta_crlf"] = b"".join(b">f%d\r\n%s\r\n" % (i, wrap(seq(int(rng.integers(1, 300))), 61).replace(b"\n", b"\r\n")) for i in range(200))
    # mixed FASTA / FASTQ, garbage before the first record
    recs["mixed"] = b"garbage line\n\n" + b"".join((b">m%d\n%s\n" % (i, seq(40))) if i % 2 else (b"@m%d\n%s\n+\n%s\n" % (i, seq(40), b"I" * 40)) for i in range(100))
    # a truncated quality string ends the stream (kseq returns -2)
    recs["truncated"] = b"".join(b"@t%d\n%s\n+\n%s\n" % (i, seq(30), b"I" * 30) for i in range(50)) + b"@bad\nACGTACGT\n+\nIII\n" + b"".join(b"@u%d\n%s\n+\n%s\n" % (i, seq(30), b"I" * 30) for i in range(50))
    recs["empty"] = b""
    recs["one"] = b"@a\nACGT\n+\nIIII"
    # no '>' anywhere (PacBio FASTQ with all-'!' qualities) / no '@' anywhere: the header search must not run to the end of the file per record
    recs["fastq_no_gt"] = b"".join(b"@p%d\n%s\n+\n%s\n" % (i, s.replace(b"N", b"A").replace(b"n", b"a"), b"!" * len(s)) for i, s in ((i, seq(int(rng.integers(1, 300)))) for i in range(400)))
    recs["fasta_no_at"] = b"".join(b">g%d\n%s\n" % (i, wrap(seq(int(rng.integers(1, 300))), 80)) for i in range(300))
    # a sequence line that is a single '\r' and the file's last byte: kseq keeps it (ks_getuntil2 returns at end of file before the strip, kseq.h:98)
    recs["cr_last_byte"] = b">r1\na\n\r"
    recs["cr_last_byte_fq"] = b"@q0\nACGT\n+\nIIII\n>r1\nacg\nt\n\r"
    # kseq drops one trailing '\r' of the WHOLE quality string after every line it appends (kseq.h:98-99), so an empty line after a
    # line that ends in "\r\r" drops the second one: here the quality comes out one short and the stream ends / exactly long enough
    # (found by tools/fuzz_reader.py; the expected counts are the reference's own kseq_read, oracle/_ref/ref_harness)
    recs["qual_crcr_then_empty_short"] = b"@q0\nACGT\n+\nIIII\n>\r\n!!!!@@U\n\nU x y\nACGT@\n+>\r\nIIII+>@!!!!>\t\t\r\r\n\nU+\n@q9\nAC\n+\nII\n"
    recs["qual_crcr_then_empty_fits"] = b"name+\nACGT+@\rx y\r\nacgtnN+\nIIII\n+\n>\r\r\n\nx y\r@>ACGT"
    return recs


KSEQ_COUNTS = {"qual_crcr_then_empty_short": (1, 4), "qual_crcr_then_empty_fits": (1, 11)}     # (records, bases) as kseq_read reports them


@pytest.mark.parametrize("name", sorted(_dialects()))
def test_mapped_reader_equals_streaming_reader(emu_lib, tmp_path, name):
    data = _dialects()[name]
    fn = str(tmp_path / (name + ".fx"))
    _write(fn, data)
    want = _digest(emu_lib, fn, 0)
    assert want[0] > 0 or name == "empty"
    if name in KSEQ_COUNTS:
        assert want[:2] == KSEQ_COUNTS[name]
    for piece in (64, 300, 5000, 1 << 20):
        for threads in (1, 4):
            got = _digest(emu_lib, fn, 1, threads, piece)
            assert got[:4] == want[:4], (name, piece, threads)
    if name == "fastq":
        assert _digest(emu_lib, fn, 1, 4, 300)[4] == 0          # the guess holds on plain FASTQ even with '@' / '+' / '>' opening quality lines: nothing parsed twice
    if name == "wrapped":
        assert _digest(emu_lib, fn, 1, 4, 300)[4] > 0           # (and where it cannot hold, the stitcher notices)


def test_mapped_reader_is_linear_on_a_fastq_without_any_gt(emu_lib, tmp_path):
    """round 4's reader searched the rest of the whole file for '>' after every FASTQ record: 40 s for 80 MB.  20 MB here must take well under that."""
    import time
    rng = np.random.default_rng(11)
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2000)].tobytes()
    rec = b"@p\n" + s + b"\n+\n" + b"!" * len(s) + b"\n"
    fn = str(tmp_path / "big.fq")
    _write(fn, rec * (20_000_000 // len(rec)))
    t0 = time.time()
    got = _digest(emu_lib, fn, 1, 4, 1 << 20)
    dt = time.time() - t0
    assert got[0] == 20_000_000 // len(rec)
    assert dt < 5.0, dt


def test_run_files_parses_plain_targets_from_the_mapping(emu_lib, tmp_path, monkeypatch):
    """the whole path on a plain FASTQ cut into tiny pieces and several index parts, against the gzip of the same file (streaming reader)"""
    from tests.helpers import read_fastx
    tn, ts, tq = read_fastx(os.path.join(GOLDEN, "tiny_all.fq.gz"))
    plain = str(tmp_path / "all.fq")
    _write(plain, b"".join(b"@" + n.encode() + b"\n" + s.tobytes() + b"\n+\n" + q.tobytes() + b"\n" for n, s, q in zip(tn, ts, tq)))
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "100K", "-p", "160"]
    q = os.path.join(GOLDEN, "tiny_sub.fq.gz")
    want = read_gz("tiny_parts.table.gz") if False else oracle_bind.table(argv + [os.path.join(GOLDEN, "tiny_all.fq.gz"), q])
    monkeypatch.setenv("LQCOV_PARSE_PIECE", "3000"); monkeypatch.setenv("LQCOV_PARSE_THREADS", "3")
    rc, out, err = run_main(emu_lib, argv + [plain, q])
    assert rc == 0, err
    assert out == want
    assert "from the mapped file" in err and "part 3" in err
    # the packed reads of a part in three slices, the data-parallel sketch kernel taking a slice while the next one goes up
    monkeypatch.setenv("LQCOV_UPLOAD_MIN_CHUNKS", "1"); monkeypatch.setenv("LQCOV_UPLOAD_SLICES", "3")
    rc, out, err = run_main(emu_lib, argv + [plain, q])
    assert rc == 0 and out == want, err


def test_run_files_streams_a_target_of_wrapped_records(emu_lib, tmp_path, monkeypatch):
    """records of several lines are copied out of the mapping by the parallel reader; beyond LQCOV_PARSE_SIDE bytes of such copies
    run_files gives the mapped reader up and streams the file, one part in memory at a time: same table"""
    from tests.helpers import read_fastx
    tn, ts, tq = read_fastx(os.path.join(GOLDEN, "tiny_all.fq.gz"))
    wrapped = str(tmp_path / "all.fa")
    _write(wrapped, b"".join(b">" + n.encode() + b"\n" + b"\n".join(s.tobytes()[j:j + 60] for j in range(0, len(s), 60)) + b"\n" for n, s in zip(tn, ts)))
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "100K", "-p", "160"]
    q = os.path.join(GOLDEN, "tiny_sub.fq.gz")
    want = oracle_bind.table(argv + [os.path.join(GOLDEN, "tiny_all.fq.gz"), q])
    monkeypatch.setenv("LQCOV_PARSE_PIECE", "3000"); monkeypatch.setenv("LQCOV_PARSE_THREADS", "3")
    rc, out, err = run_main(emu_lib, argv + [wrapped, q])
    assert rc == 0 and out == want and "from the mapped file" in err
    monkeypatch.setenv("LQCOV_PARSE_SIDE", "20000")
    rc, out, err = run_main(emu_lib, argv + [wrapped, q])
    assert rc == 0 and out == want and "streaming reader" in err and "from the mapped file" not in err


def test_replay_side_sort_equals_klibs_order_ties_included(tmp_path):
    """longqc_amd/csrc/sat_replay.hpp re-sorts the chains of a saturated query the way the reference does (chain.c:139-146,
    hit.c:60-70: radix_sort_128x, klib's unstable in-place MSD sort).  Its order of EQUAL keys is what has to be right, and the
    pile-up tests hardly ever produce equal keys: here the host routine is held against the oracle's restatement of klib's sort
    (pinned by the golden anchor orders) on arrays full of ties, of every size class (insertion sort up to 64, bucket recursion
    above)."""
    import ctypes as C
    import subprocess
    so = str(tmp_path / "sat_shim.so")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "emu", "sat_replay_shim.cpp"), "-o", so],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    shim = C.CDLL(so)
    oracle_bind.ensure_oracle()
    ora = C.CDLL(os.path.join(ROOT, "oracle", "liblqcov_oracle.so"))
    ora.lqo_sort_128x.argtypes = [C.c_void_p, C.c_size_t]
    shim.shim_klib_sort_128x.argtypes = [C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(3)
    for n in [0, 1, 2, 63, 64, 65, 66, 200, 1000, 5000, 40000]:
        for shape in range(6):
            if shape == 0:
                x = rng.integers(0, 7, size=n, dtype=np.uint64)                                  # few distinct keys, low byte only
            elif shape == 1:
                x = rng.integers(0, 5, size=n, dtype=np.uint64) << np.uint64(56)                 # ... top byte only
            elif shape == 2:
                x = (rng.integers(0, 3, size=n, dtype=np.uint64) << np.uint64(40)) | rng.integers(0, 4, size=n, dtype=np.uint64)
            elif shape == 3:
                x = rng.integers(0, 1 << 62, size=n, dtype=np.uint64); x[rng.integers(0, max(n, 1), size=n // 2)] = x[0] if n else 0   # half of them one key
            elif shape == 4:
                x = (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)) | (rng.integers(0, 300, size=n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 50, size=n, dtype=np.uint64)   # anchors' shape
            else:
                x = np.sort(rng.integers(0, 100, size=n, dtype=np.uint64))[::-1].copy()            # descending, many ties
            a = np.empty(2 * n, dtype=np.uint64)
            a[0::2] = x; a[1::2] = np.arange(n, dtype=np.uint64)
            b = a.copy()
            shim.shim_klib_sort_128x(a.ctypes.data_as(C.c_void_p), n)
            ora.lqo_sort_128x(b.ctypes.data_as(C.c_void_p), n)
            assert np.array_equal(a, b), (n, shape)
            assert np.all(a[0::2][1:] >= a[0::2][:-1])
