import gzip
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_cases():
    return json.load(open(os.path.join(GOLDEN, "cases.json")))


def read_gz(name):
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read().decode()


@pytest.fixture(scope="session")
def emu_lib():
    """tests/emu/liblqcov_emu.so: the product's kernel sources compiled as plain C++ against the serial
    HIP stand-in (tests/emu/hipemu.hpp).  Test-only; checks kernel *logic* without a GPU."""
    from longqc_amd import api
    if os.environ.get("LQCOV_EMU_LIB"):                  # e.g. an AddressSanitizer build of the same sources (tools/emu_asan.sh)
        return api.load_library(os.environ["LQCOV_EMU_LIB"])
    csrc = os.path.join(ROOT, "longqc_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "emu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return api.load_library(os.path.join(ROOT, "tests", "emu", "liblqcov_emu.so"))


@pytest.fixture(scope="session")
def gpu_lib():
    """longqc_amd/liblqcov.so, the hipcc/gfx950 build (make is a no-op when it is up to date)"""
    csrc = os.path.join(ROOT, "longqc_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "all"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    from longqc_amd import api
    return api.load_library()


@pytest.fixture(scope="session")
def datasets(tmp_path_factory):
    """seeded synthetic sets written once per session: name -> (targets path, queries path)"""
    from longqc_amd import synth
    d = tmp_path_factory.mktemp("data")
    out = {}

    def get(name):
        if name not in out:
            T, Q = synth.make_dataset(synth.CONFIGS[name])
            tf, qf = str(d / (name + "_all.fq")), str(d / (name + "_sub.fq"))
            synth.write_fastq(tf, T)
            synth.write_fastq(qf, Q)
            out[name] = (tf, qf)
        return out[name]
    return get
