"""The seed filter (kernels_seed.hpp) at every threshold it supports, through the real library on the GPU: the twin of
tests/test_emu_pipeline.py::test_emulated_filter_thresholds (added at the end of round 5, after the round's last GPU run)."""
import pytest

import tests.test_emu_pipeline as E

pytestmark = pytest.mark.gpu


def test_gpu_filter_thresholds(gpu_lib, tmp_path, monkeypatch, capfd):
    """-n / -m from a filter threshold of 2 to 15, 16 and 1 (no filter): the reference's rows each time; oversized buckets in
    passes over their targets and by pairs only"""
    E.check_filter_thresholds(gpu_lib, tmp_path, monkeypatch, capfd)
