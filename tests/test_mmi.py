"""the reference's index files (.mmi; mm_idx_dump / mm_idx_load, index.c:390-479) through the
product's kernel sources under the serial HIP stand-in.  Fixtures: tests/golden/make_mmi_golden.py (index files and tables
written by the reference binary).  The GPU versions of these tests are in test_gpu_parity.py."""
import gzip
import json
import os
import struct
import subprocess

import pytest

from tests import oracle_bind
from tests.conftest import GOLDEN, read_gz
from tests.helpers import run_main, slow_emu

CASES = json.load(open(os.path.join(GOLDEN, "mmi_cases.json")))


def gunzip_to(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read())
    return dst


def parse_mmi(data: bytes):
    """-> list of parts: dict(header=(w,k,b,n_seq,flag), seqs=[(name,len)], index={minimizer: [y...]}, S=bytes)"""
    parts, o = [], 0
    while o < len(data):
        assert data[o:o + 4] == b"MMI\x02"
        w, k, b, n_seq, flag = struct.unpack_from("<5I", data, o + 4); o += 24
        seqs, sum_len = [], 0
        for _ in range(n_seq):
            l = data[o]; name = data[o + 1:o + 1 + l]; (ln,) = struct.unpack_from("<I", data, o + 1 + l); o += 5 + l
            seqs.append((name, ln)); sum_len += ln
        index = {}
        for bi in range(1 << b):
            (n_p,) = struct.unpack_from("<i", data, o); o += 4
            p = struct.unpack_from("<%dQ" % n_p, data, o); o += 8 * n_p
            (size,) = struct.unpack_from("<I", data, o); o += 4
            for _ in range(size):
                key, val = struct.unpack_from("<2Q", data, o); o += 16
                minier = (key >> 1) << b | bi
                assert minier not in index
                index[minier] = [val] if key & 1 else list(p[val >> 32:(val >> 32) + (val & 0xffffffff)])
                assert (key & 1) or (val & 0xffffffff) > 1
        S = b""
        if not flag & 2:
            nb = (sum_len + 7) // 8 * 4
            S = data[o:o + nb]; o += nb
        parts.append(dict(header=(w, k, b, n_seq, flag), seqs=seqs, index=index, S=S))
    return parts


def check_maps_from_reference_index(lib, case, tmp_path):
    mmi = gunzip_to(os.path.join(GOLDEN, case["index"]), str(tmp_path / (case["name"] + ".mmi")))
    argv = case["map_argv"][:-2] + [mmi, os.path.join(GOLDEN, "adv_sub.fq.gz")]
    rc, out, err = run_main(lib, argv)
    assert rc == 0, err
    assert out == read_gz(case["expect"])
    if case["name"] == "adv_k19w10":
        assert "overridden by parameters used in the prebuilt index" in err


def check_dump_equals_reference_dump(lib, case, tmp_path):
    """our -d file == the reference's -d file up to the order of the (key, value) pairs inside a bucket"""
    ours = str(tmp_path / "ours.mmi")
    argv = case["build_argv"][:-3] + ["-d", ours, os.path.join(GOLDEN, "adv_all.fa.gz")]
    rc, out, err = run_main(lib, argv)
    assert rc == 0, err
    assert out == ""                                          # index only: nothing on stdout (minimap2-coverage.c:460-468)
    mine = open(ours, "rb").read()
    ref = gzip.open(os.path.join(GOLDEN, case["index"]), "rb").read()
    assert len(mine) == len(ref)
    pm, pr = parse_mmi(mine), parse_mmi(ref)
    assert len(pm) == len(pr) and len(pm) == (1 if case["name"] == "adv_k19w10" else len(pr))
    for a, b in zip(pm, pr):
        assert a["header"] == b["header"] and a["seqs"] == b["seqs"]
        assert a["S"] == b["S"]
        assert a["index"] == b["index"]
    return ours


@pytest.mark.parametrize("case", [pytest.param(c, marks=slow_emu) if c["name"] == "adv_k12w5_300K" else c for c in CASES], ids=lambda c: c["name"])
def test_emulated_maps_from_the_reference_index_file(emu_lib, case, tmp_path):
    check_maps_from_reference_index(emu_lib, case, tmp_path)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_emulated_index_dump_equals_the_reference_dump_and_loads_back(emu_lib, case, tmp_path):
    ours = check_dump_equals_reference_dump(emu_lib, case, tmp_path)
    argv = case["map_argv"][:-2] + [ours, os.path.join(GOLDEN, "adv_sub.fq.gz")]
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == read_gz(case["expect"])
    if oracle_bind.have_ref():                                # the reference binary reads our file (build container only)
        r = subprocess.run([oracle_bind.REF_BIN] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout.decode() == out


@slow_emu
def test_emulated_dump_while_mapping_and_part_level_calls(emu_lib, tmp_path):
    """-d together with a query file maps as usual and writes the index; lqcov_part_dump / lqcov_part_load"""
    from longqc_amd import api
    d1 = str(tmp_path / "d1.mmi")
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "300K", "-p", "160", "-d", d1,
            os.path.join(GOLDEN, "adv_all.fa.gz"), os.path.join(GOLDEN, "adv_sub.fq.gz")]
    rc, out, err = run_main(emu_lib, argv)
    assert rc == 0, err
    assert out == read_gz("adv_mmi_k12w5.table.gz")           # same parts as the index-file case
    prm = api.Params()
    emu_lib.lqcov_params_default(prm)
    prm.no_self = 1; prm.min_ovlp = 0; prm.min_score_med = 160; prm.min_score_good = 160
    eng = api.Engine(prm, 0, lib=emu_lib)
    from tests.helpers import read_fastx
    qn, qs, qq = read_fastx(os.path.join(GOLDEN, "adv_sub.fq.gz"))
    eng.set_queries(qn, qs, qq)
    off, n_parts = 0, 0
    d2 = str(tmp_path / "d2.mmi")
    while True:
        pt, off = eng.part_load(d1, off)
        if pt is None:
            break
        eng.part_map(pt)
        with pytest.raises(api.LqcovError):                   # a loaded part carries no sequences to write back
            eng.part_dump(pt, d2, append=n_parts > 0)
        eng.part_release(pt)
        n_parts += 1
    eng.finish()
    assert n_parts == len(parse_mmi(open(d1, "rb").read())) and n_parts > 1
    assert eng.table_text() == read_gz("adv_mmi_k12w5.table.gz")
    eng.close()


def test_emulated_index_with_more_minimizers_than_counters_is_refused(emu_lib, tmp_path):
    """k=15 w=5 index, default -k 12 -w 5 on the command line: queries have more minimizers under the index's values than
    the counters the reference allocated (minimap2-coverage.c:422) and good chains match the surplus ones: the reference
    overruns its arrays (glibc aborts it on this input)"""
    mmi = str(tmp_path / "k15.mmi")
    rc, out, err = run_main(emu_lib, ["-k", "15", "-w", "5", "-d", mmi, os.path.join(GOLDEN, "adv_all.fa.gz")])
    assert rc == 0, err
    rc, out, err = run_main(emu_lib, ["-Y", "-l", "0", "-q", "160", "-p", "160", mmi, os.path.join(GOLDEN, "adv_sub.fq.gz")])
    assert rc != 0 and "overruns its counter array" in err
    rc, out, err = run_main(emu_lib, ["-Y", "-l", "0", "-q", "160", "-p", "160", "-k", "15", "-w", "5", mmi, os.path.join(GOLDEN, "adv_sub.fq.gz")])
    assert rc == 0, err
    want = oracle_bind.table(["-Y", "-l", "0", "-q", "160", "-p", "160", "-k", "15", "-w", "5", "-I", "4G",
                              os.path.join(GOLDEN, "adv_all.fa.gz"), os.path.join(GOLDEN, "adv_sub.fq.gz")])
    assert out == want


def test_emulated_rejected_combinations(emu_lib, tmp_path):
    """-d with an index file as the target, a truncated index file, neither query nor -d"""
    mmi = gunzip_to(os.path.join(GOLDEN, "adv_k19w10.mmi.gz"), str(tmp_path / "a.mmi"))
    rc, out, err = run_main(emu_lib, ["-d", str(tmp_path / "b.mmi"), mmi])
    assert rc != 0 and "not supported" in err
    cut = str(tmp_path / "cut.mmi")
    open(cut, "wb").write(open(mmi, "rb").read()[:100000])
    rc, out, err = run_main(emu_lib, ["-Y", "-l", "0", "-q", "160", "-p", "160", "-k", "19", "-w", "10", cut, os.path.join(GOLDEN, "adv_sub.fq.gz")])
    assert rc != 0 and "truncated" in err
    rc, out, err = run_main(emu_lib, ["-Y", mmi])
    assert rc != 0
