"""Where can klib's order of equal-x anchors be observed?  (lqmap.c:238 sorts with an unstable radix sort and chain.c:69-76
breaks score ties by array order.)  The engine sorts with any correct sort and asks for klib's own order only in the
(strand, rid) runs where two anchors of equal x are inside the band of one scan (chain.c:52-56: a candidate outside the band
is stepped over before any state changes) or are both peaks of equal score (chain.c:102-125).  The oracle's sort modes 2 / 3
state that rule on the CPU: equal-x anchors in reverse / hashed order everywhere, klib's order only in the runs the rule
names; every other run with equal x is chained in both orders and compared (exit status 3 on a difference)."""
import subprocess

import pytest

from tests import oracle_bind
from tests.helpers import ONT


def _run(argv, mode, env=None):
    import os
    oracle_bind.ensure_oracle()
    e = dict(os.environ); e.update(env or {})
    return subprocess.run([oracle_bind.ORACLE_CLI, "table", mode] + [str(a) for a in argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


def _stats(err):
    line = [l for l in err.splitlines() if l.startswith("[ties]")][-1]
    import re
    return dict(with_ties=int(re.search(r"with ties (\d+)", line).group(1)), observable=int(re.search(r"observable (\d+)", line).group(1)),
                mismatch=int(re.search(r"MISMATCH (\d+)", line).group(1)))


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("mode", ["--ties-reversed", "--ties-hashed"])
def test_runs_the_rule_calls_order_free_chain_alike_in_any_order(tmp_path, seed, mode):
    from tests.test_emu_pipeline import _repeat_rich_dataset
    tf, qf = _repeat_rich_dataset(tmp_path, seed)
    argv = ONT + [tf, qf]
    want = oracle_bind.ref_table(argv) if oracle_bind.have_ref() else oracle_bind.table(argv)
    r = _run(argv, mode)
    assert r.returncode == 0, r.stderr[-2000:]
    st = _stats(r.stderr)
    assert st["mismatch"] == 0 and st["with_ties"] > 50      # the input has teeth: plenty of runs hold equal x
    assert r.stdout == want


def test_the_rule_is_needed(tmp_path):
    """with the rule switched off (every run in the other order) some run chains differently: the check has power"""
    from tests.test_emu_pipeline import _repeat_rich_dataset
    bad = 0
    for seed in range(3):
        tf, qf = _repeat_rich_dataset(tmp_path, seed)
        r = _run(ONT + [tf, qf], "--ties-reversed", {"LQO_TIE_NONE": "1"})
        bad += _stats(r.stderr)["mismatch"]
    assert bad > 0


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_config_shaped_sets(datasets, name):
    tf, qf = datasets(name)
    argv = ONT + [tf, qf]
    r = _run(argv, "--ties-hashed")
    assert r.returncode == 0 and r.stdout == oracle_bind.table(argv)
