"""The output contract of the path as its consumer reads it: the 9-column table
`minimap2-coverage` prints, parsed the way LongQC's `LqCoverage` parses it (lq_coverage.py:87-107: tab-separated,
no header, columns 3 and 4 kept as text, spike-in control reads dropped by name), and the figures that are exact
functions of the table (lq_coverage.py:211-224).  The mixture fits that follow in the reference are CPU
post-processing of <= 10 k rows with a random start and are not part of this build.

Same names as the reference where one exists: get_unmapped_frac(), get_unmapped_med_frac(), get_high_div_frac(),
get_control_num(), get_control_frac() (lq_coverage.py:160-203).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np


class CoverageTable:
    DIV_SCORE_THRESHOLD = 0.25         # lq_coverage.py:72
    COV_THRESHOLD_FOR_DIV_SC = 25      # lq_coverage.py:73
    # column numbers: lq_coverage.py:76-84
    READ_NAME_COLUMN, QLENGTH_COLUMN, N_MBASE_COLUMN, MED_READ_COV_CORS, T1_COVERAGE_COLUMN, QV_COLUMN, DIV_COLUMN, COVERAGE_COLUMN = 0, 1, 2, 4, 5, 6, 7, 8

    @staticmethod
    def _rows(path) -> List[List[str]]:
        with open(path) as f:
            return [l.rstrip("\n").split("\t") for l in f if l.strip()]

    def __init__(self, table_path, control_filtering: Optional[str] = None):
        rows = self._rows(table_path)
        self.control_reads = None
        if control_filtering is not None:
            self.control_reads = [r[0] for r in self._rows(control_filtering) if float(r[self.T1_COVERAGE_COLUMN]) >= 0.5]
            drop = set(self.control_reads)
            rows = [r for r in rows if r[0] not in drop]
        if not rows:
            raise ValueError("coverage table %s has no rows" % table_path)     # the reference divides by zero here
        if any(len(r) != 9 for r in rows):
            raise ValueError("coverage table %s: expected 9 tab-separated columns" % table_path)
        self.names = [r[0] for r in rows]
        self.qlen = np.array([int(r[1]) for r in rows], dtype=np.int64)
        self.n_mbase = np.array([int(r[2]) for r in rows], dtype=np.int64)
        self.good_coords = [r[3] for r in rows]
        self.med_coords = [r[4] for r in rows]
        self.t1_cov, self.qv, self.div, self.cov = (np.array([float(r[c]) for r in rows]) for c in (5, 6, 7, 8))
        n = len(rows)
        med0 = np.array([c == "0" for c in self.med_coords])
        self.unmapped_frac_trimmed = int((self.t1_cov == 0.0).sum()) / n
        self.unmapped_frac_untrimmed = int((self.n_mbase == 0).sum()) / n
        self.unmapped_frac_med = int(med0.sum()) / n
        self.high_div_frac = int(((self.div >= self.DIV_SCORE_THRESHOLD) & (self.t1_cov >= self.COV_THRESHOLD_FOR_DIV_SC) & ~med0).sum()) / n

    def __len__(self):
        return len(self.names)

    def get_unmapped_frac(self):
        return self.unmapped_frac_trimmed

    def get_unmapped_med_frac(self):
        return self.unmapped_frac_med

    def get_high_div_frac(self):
        return self.high_div_frac

    def get_control_num(self):
        return len(self.control_reads) if self.control_reads else 0

    def get_control_frac(self):
        if self.control_reads:
            return len(self.control_reads) / (len(self.control_reads) + len(self.names))
        return 0.0
