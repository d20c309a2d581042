"""Makes tests/golden/pileup16_rows.json: the table the reference binary (oracle/_ref/minimap2-coverage, built from the reference's
own sources by oracle/Makefile) prints for the seeded pile-up of tests/test_emu_pipeline.py::_pileup16_dataset -- the one input of
the suite on which a real uint16 match counter fills up (esterr.c:130,136).  Run in the build container:
    python tests/golden/make_pileup_golden.py
"""
import json
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import oracle_bind                                                     # noqa: E402
from tests.test_emu_pipeline import PILEUP16, PILEUP16_ARGV, _pileup16_dataset     # noqa: E402

with tempfile.TemporaryDirectory() as d:
    tf, qf = _pileup16_dataset(pathlib.Path(d), **PILEUP16)
    table = oracle_bind.ref_table(PILEUP16_ARGV + [tf, qf])
    os.environ["LQO_REGS_ASCENDING"] = "1"
    other = oracle_bind.table(PILEUP16_ARGV + [tf, qf])
assert table != other, "the input does not tell the order of the chains"
out = os.path.join(ROOT, "tests", "golden", "pileup16_rows.json")
json.dump({"dataset": PILEUP16, "argv": PILEUP16_ARGV, "table": table, "generator": "tests/golden/make_pileup_golden.py",
           "note": "table = stdout of the reference binary; the oracle with the chains in the opposite order prints another one"}, open(out, "w"), indent=1)
print(table)
