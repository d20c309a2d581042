#!/usr/bin/env python3
"""Consumer acceptance vectors: what the REFERENCE's LqCoverage (/root/reference/lq_coverage.py)
reads out of each golden coverage table -- the exact functions of the table it computes before its (randomly
initialised, not a parity target) mixture fits: unmapped_frac_trimmed / _untrimmed / _med, high_div_frac
(lq_coverage.py:211-224), the row count after the spike-in control filter (:104-107) and get_control_num / get_control_frac
(:193-203).  Run in the build container only (imports the reference).  Output: tests/golden/consumer.json."""
import gzip, json, os, sys, tempfile, logging, warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
logging.disable(logging.CRITICAL)
warnings.simplefilter("ignore")
import pandas as pd                      # noqa: E402
from lq_coverage import LqCoverage      # noqa: E402


def facts(table_gz, control_gz=None):
    with tempfile.TemporaryDirectory() as d:
        t = os.path.join(d, "t.txt"); open(t, "wb").write(gzip.open(os.path.join(HERE, table_gz)).read())
        c = None
        if control_gz:
            c = os.path.join(d, "c.txt"); open(c, "wb").write(gzip.open(os.path.join(HERE, control_gz)).read())
        # the constructor's own statements up to the estimate (lq_coverage.py:87-107), then the estimate itself; the
        # mixture fit that follows may fail on a tiny table -- the fractions are set before it starts
        o = object.__new__(LqCoverage)
        o.df = pd.read_table(t, sep='\t', header=None, dtype={3: str, 4: str})
        o.control_reads = None
        o.isTranscript = False
        o.warnings, o.errors = [], []
        if c:
            o.df_control = pd.read_table(c, sep='\t', header=None)
            o.control_reads = o.df_control[o.df_control[LqCoverage.T1_COVERAGE_COLUMN] >= 0.5][0].tolist()
            o.df = o.df[~o.df[LqCoverage.READ_NAME_COLUMN].isin(o.control_reads)]
        try:
            o._LqCoverage__est_coverage()
        except Exception:
            pass
        return dict(table=table_gz, control=control_gz, n_rows=int(o.df.shape[0]),
                    control_reads=list(o.control_reads) if o.control_reads is not None else None,
                    unmapped_frac_trimmed=o.unmapped_frac_trimmed, unmapped_frac_untrimmed=o.unmapped_frac_untrimmed,
                    unmapped_med_frac=o.get_unmapped_med_frac(), high_div_frac=o.get_high_div_frac(),
                    control_num=o.get_control_num(), control_frac=o.get_control_frac())


tables = sorted(f for f in os.listdir(HERE) if f.endswith(".table.gz") and "spike" not in f)
cases = [facts(t) for t in tables]
cases.append(facts("adv_ont.table.gz", "adv_spike.table.gz"))
cases.append(facts("tiny_ont.table.gz", "tiny_spike.table.gz"))
json.dump(cases, open(os.path.join(HERE, "consumer.json"), "w"), indent=1)
print("wrote", len(cases), "cases")
