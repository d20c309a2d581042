"""One real-size parity case (tests/test_gpu_parity.py::test_gpu_real_size_parts_rows_equal_the_reference_fixture calls main() in the
test process; on its own:)
    python -m tests.real_size_runner cfg4s|cfg5s
Index parts of REAL size (-I 4G) through the part-level C ABI, 40 rows compared with what the reference binary printed in the
build container (tests/golden/<name>_rows.json, made by tests/golden/make_scale_golden.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from longqc_amd import api, multigpu, synth  # noqa: E402


def main(name, lib=None):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", name + "_rows.json")))
    cfg = synth.SCALE_SLICES[name]
    genome = synth.make_genome(cfg)
    F = synth.make_reads_flat(cfg, genome)
    Q = synth.make_reads(cfg, genome, indices=synth.reservoir_subsample(cfg.n_reads, cfg.nsample))
    p, _, _ = api.parse_args(g["argv"] + ["t", "q"])
    lens = np.diff(F.off).astype(np.int64)
    parts = multigpu.split_parts(lens, int(p.batch_size), int(p.idx_mini_batch))
    assert len(parts) >= (3 if name == "cfg4s" else 2)
    P = api.PackedReads(F.flat, F.off, F.names())
    eng = api.Engine(p, device=0, lib=lib or api.load_library())
    eng.set_queries(Q.names, Q.seqs, Q.quals)
    pt = eng.part_begin()
    for lo, hi in parts:
        eng.part_clear(pt); eng.part_add_packed(pt, P, lo, hi); eng.part_build(pt); eng.part_map(pt)
    eng.finish()
    lines = eng.table_text().splitlines()
    st = eng.map_stats()
    eng.close()
    bad = [s for s, row in zip(g["subsample_slots"], g["rows"]) if lines[s] != row]
    assert not bad, (bad[:5], st)
    print("real-size rows identical: %s, %d of %d; %s" % (name, len(g["rows"]), len(g["rows"]), st))


if __name__ == "__main__":
    main(sys.argv[1])
