"""Host logic (FASTA/Q stream, part boundaries, error behaviour, LqExec counterpart) -- exercised
through the emulator build so it runs without a GPU."""
import gzip
import os

import numpy as np
import pytest

from tests import oracle_bind
from tests.conftest import GOLDEN, read_gz
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
