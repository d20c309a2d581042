"""The C-ABI library loads without a GPU, exports every symbol include/lqcov.h declares, parses the
reference's option table, and refuses -- loudly -- to compute without a HIP device."""
import ctypes as C
import os
import re

import pytest

from longqc_amd import api
from tests.conftest import ROOT


def _declared():
    hdr = open(os.path.join(ROOT, "include", "lqcov.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lqcov_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(gpu_lib):
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(gpu_lib, n), n
    assert gpu_lib.lqcov_abi_version() == 1
    assert set(gpu_lib._sig_names) == set(names), set(names) ^ set(gpu_lib._sig_names)


def test_library_is_the_hip_build(gpu_lib):
    """liblqcov.so carries gfx950 code objects (hipcc --offload-arch=gfx950), i.e. it is not a CPU stand-in."""
    blob = open(api.library_path(), "rb").read()
    assert b"gfx950" in blob
    assert b"k_sketch" in blob and b"k_chain" in blob and b"k_sort_walk" in blob


def test_defaults_match_reference_binary(gpu_lib):
    p = api.default_params()
    assert (p.k, p.w, p.hpc, p.batch_size, p.idx_mini_batch) == (12, 5, 0, 4000000000, 50000000)
    assert (p.max_gap, p.min_cnt, p.min_chain_score, p.max_chain_skip, p.bw) == (10000, 3, 40, 25, 500)
    assert (p.max_overhang, p.min_ovlp, p.min_coverage, p.min_ratio) == (2000, 1000, 3, 0.4)
    assert abs(p.mid_occ_frac - 2e-4) < 1e-9


def test_parse_sampleqc_argv(gpu_lib):
    # longQC.py:440-445 (ONT) and :555-556 (PacBio spike-in)
    p, t, q = api.parse_args(["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "4G", "-p", "160", "-t", "8", "all.fq.gz", "sub.fastq"])
    assert (p.k, p.w, p.batch_size, p.min_score_med, p.min_score_good, p.min_ovlp, p.no_self, p.ava, p.n_threads) == (12, 5, 4000000000, 160, 160, 0, 1, 0, 8)
    assert (t, q) == ("all.fq.gz", "sub.fastq")
    p, t, q = api.parse_args(["-Y", "-Hk15", "-w", "10", "-c", "1", "-l", "0", "--filter", "-t", "4", "ref.fasta", "sub.fastq"])
    assert (p.hpc, p.k, p.w, p.min_coverage, p.filter_flag, p.min_score_med, p.min_score_good) == (1, 15, 10, 1, 1, 40, 40)
    p, _, _ = api.parse_args(["-Y", "-I", "500K", "-m", "50", "a", "b"])
    assert (p.batch_size, p.min_chain_score, p.min_score_med, p.min_score_good) == (500000, 50, 50, 50)


@pytest.mark.parametrize("argv,msg", [
    (["a", "b"], "Choose either -X"),
    (["-X", "-Y", "a", "b"], "mutually exclusive"),
    (["-Y", "-m", "50", "-p", "40", "a", "b"], "-p must be larger"),
    (["-Y", "-p", "80", "-q", "60", "a", "b"], "-q must be larger"),
    (["-Y", "a"], "not enough arguments"),
    (["-Y", "--nope", "a", "b"], "unrecognized option"),
])
def test_parse_errors_like_reference(gpu_lib, argv, msg):
    with pytest.raises(api.LqcovError) as e:
        api.parse_args(argv)
    assert msg in str(e.value)


def test_no_gpu_fails_loudly(gpu_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.LqcovError) as e:
        api.Engine()
    assert "HIP device" in str(e.value)
    from longqc_amd import exec as lqexec
    rc = lqexec.run_argv(["-Y", os.path.join(ROOT, "tests", "golden", "tiny_all.fq.gz"), os.path.join(ROOT, "tests", "golden", "tiny_sub.fq.gz")],
                         out=os.devnull, err=os.devnull)
    assert rc == -3


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(OSError):
        api.load_library(str(tmp_path / "liblqcov.so"))


def test_product_never_touches_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/"""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "longqc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle_bind|lqcov_oracle|liblqcov_emu|/oracle/|oracle\.", txt) and f != "Makefile":
                    bad.append(os.path.join(base, f))
    assert not bad, bad
