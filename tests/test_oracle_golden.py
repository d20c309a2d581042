"""The oracle (oracle/lqcov_oracle.c, the CPU restatement) against the committed golden vectors,
which were produced by the reference itself (tests/golden/make_golden.py -> oracle/_ref)."""
import os

import pytest

from tests import oracle_bind
from tests.conftest import GOLDEN, read_gz


def _cases(kind):
    import json
    return [c for c in json.load(open(os.path.join(GOLDEN, "cases.json"))) if c["kind"] == kind]


@pytest.mark.parametrize("case", _cases("table"), ids=lambda c: c["name"])
def test_oracle_table_matches_reference_fixture(case):
    argv = list(case["argv"])
    argv[-2] = os.path.join(GOLDEN, argv[-2]); argv[-1] = os.path.join(GOLDEN, argv[-1])
    assert oracle_bind.table(argv) == read_gz(case["expect"])


def _oracle_dump_args(args):
    what = args[0]
    k, w, hpc = args[1], args[2], args[3]
    opts = ["-k", k, "-w", w] + (["-H"] if hpc == "1" else [])
    if what == "sketch":
        return what, opts, [os.path.join(GOLDEN, args[4])]
    if what == "index":
        return what, opts + ["-I", args[4]], [os.path.join(GOLDEN, args[5])]
    return what, opts + ["-I", args[4], "-m", args[5], "-p", args[6], "-q", args[7], "-l", "0"], [os.path.join(GOLDEN, args[8]), os.path.join(GOLDEN, args[9])]


@pytest.mark.parametrize("case", _cases("dump"), ids=lambda c: c["name"])
def test_oracle_function_dumps_match_reference_fixture(case):
    what, opts, files = _oracle_dump_args(case["args"])
    assert oracle_bind.dump(what, opts, files) == read_gz(case["expect"])


def test_grouped_chaining_and_stable_sort_modes_on_fixture():
    """The GPU formulation (per (strand,rid) DP, runs < min_cnt dropped) is exact; a stable anchor order is
    NOT (it is only checked to agree on this small input -- see test_oracle_vs_ref for a counterexample)."""
    argv = ["-Y", "-l", "0", "-q", "160", "-k", "12", "-w", "5", "-I", "100K", "-p", "160", os.path.join(GOLDEN, "adv_all.fa.gz"), os.path.join(GOLDEN, "adv_sub.fq.gz")]
    want = read_gz("adv_parts.table.gz")
    assert oracle_bind.table(argv, ["--grouped"]) == want


def test_oracle_fills_a_uint16_counter_like_the_reference(tmp_path):
    """esterr.c:130,136 on a pile-up that fills a real uint16 counter: the oracle prints what the reference binary printed
    (tests/golden/pileup16_rows.json, made by make_pileup_golden.py), and another table with the chains in the opposite order"""
    import json
    from tests.test_emu_pipeline import _pileup16_dataset
    fx = json.load(open(os.path.join(GOLDEN, "pileup16_rows.json")))
    tf, qf = _pileup16_dataset(tmp_path, **fx["dataset"])
    assert oracle_bind.table(fx["argv"] + [tf, qf]) == fx["table"]
