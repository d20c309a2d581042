"""Host-side counterpart of the slice of `longQC.py sampleqc` that feeds and drives the coverage path
(SURVEY.md section 8(f)-2): building the query set (LongQC's seed-7 chunk reservoir), writing it as FASTQ,
the preset -> argv table, and the launch itself -- through the argv-compatible boundary or, without the
temporary FASTQ round trip, straight from memory through the C ABI.

Reference behaviour mirrored here (same names, argument meaning and order where a function exists there):
  subsample_from_chunk      lq_utils.py:371-411
  write_fastq               lq_utils.py:352-369
  replace_masked            longQC.py:369-406   (re-draw replacements for heavily masked subsample reads)
  coverage_argv             longQC.py:171-233 (presets), :438-446 (main call), :555-556 (spike-in call)
Reads are [name, seq, qual] lists like the reference's chunks (lq_utils.py:211-289).
"""
from __future__ import annotations

import os
import shlex
from typing import Iterable, List, Optional, Sequence

import numpy as np

# minimap2 parameters per preset: (-p for normal reads, -p for --short reads) -- longQC.py:171-220
PRESET_MED_SCORE = {
    "pb-rs2": (80, 60), "pb-sequel": (80, 60), "pb-hifi": (80, None),
    "ont-ligation": (160, 140), "ont-rapid": (160, 140), "ont-1dsq": (160, 140),
}
MINIMAP2_PARAMS = "-Y -l 0 -q 160"                                      # longQC.py:177,186,194,202,209,217
MINIMAP2_FILTERING_PARAMS = "-Y -Hk15 -w 10 -c 1 -l 0 --filter"        # longQC.py:255 (spike-in control)


def db_params(preset: str, fast: bool = False, inds: str = "4G") -> str:
    """longQC.py:222-231"""
    if preset == "pb-hifi":
        return ("-k 19 -w 10 -I %s" if fast else "-k 15 -w 5 -I %s") % inds
    return ("-k 15 -w 5 -I %s" if fast else "-k 12 -w 5 -I %s") % inds


def coverage_argv(preset: str, fastx_path: str, sample_path: str, ncpu: int = 4, fast: bool = False, inds: str = "4G",
                  short: bool = False) -> List[str]:
    """The argv of the main minimap2-coverage call (longQC.py:443-445); short=True gives the --short variant
    (-k 12 -w 5 and the lower -p, longQC.py:447-... )."""
    if preset not in PRESET_MED_SCORE:
        raise ValueError("unknown preset %r" % preset)
    med, med_short = PRESET_MED_SCORE[preset]
    if short:
        if med_short is None:
            raise ValueError("--short is not defined for %s" % preset)
        return shlex.split("%s %s -p %d -t %d %s %s" % (MINIMAP2_PARAMS, "-k 12 -w 5 -I %s" % inds, med_short, ncpu, fastx_path, sample_path))
    return shlex.split("%s %s -p %d -t %d %s %s" % (MINIMAP2_PARAMS, db_params(preset, fast, inds), med, ncpu, fastx_path, sample_path))


def db_build_argv(preset: str, tempdb_path: str, fastx_path: str, fast: bool = False, inds: str = "4G", short: bool = False) -> List[str]:
    """`sampleqc --db`, first call: index every read into a prebuilt .mmi (longQC.py:266-277); short=True gives the
    t_db_minimap2_short variant (-k 12 -w 5)."""
    params = "-k 12 -w 5 -I %s" % inds if short else db_params(preset, fast, inds)
    return shlex.split("%s -d %s %s" % (params, tempdb_path, fastx_path))


def db_coverage_argv(preset: str, tempdb_path: str, sample_path: str, ncpu: int = 4, short: bool = False) -> List[str]:
    """`sampleqc --db`, second call: map the subsample against the prebuilt index; no -k/-w on this command line
    (longQC.py:440-442, 531-533), so the binary sizes its counters with its defaults (k=12, w=5) while the index's
    own k / w drive the mapping."""
    if preset not in PRESET_MED_SCORE:
        raise ValueError("unknown preset %r" % preset)
    med, med_short = PRESET_MED_SCORE[preset]
    if short and med_short is None:
        raise ValueError("--short is not defined for %s" % preset)
    return shlex.split("%s -p %d -t %d %s %s" % (MINIMAP2_PARAMS, med_short if short else med, ncpu, tempdb_path, sample_path))


def spikein_argv(filter_ref: str, sample_path: str, ncpu: int = 4) -> List[str]:
    """longQC.py:555-556"""
    return shlex.split("%s -t %d %s %s" % (MINIMAP2_FILTERING_PARAMS, ncpu, filter_ref, sample_path))


def subsample_from_chunk(chunk, cum_n_seq, s_reads, param, s_seed=7, elist=None):
    """The draw of lq_utils.py:371-411 over one chunk of reads, all reads of the chunk at once.

    The reference's generator is re-seeded per chunk (seed 7) and draws len(chunk) + 1 uniforms; the k-th read that is not on
    `elist` uses the k-th of them.  param >= 1 keeps a reservoir of int(param) reads: the n-th read seen so far (over all
    chunks, n counted from 1) goes to slot n - 1 while the reservoir fills and to slot floor(u * n) afterwards, if that slot
    exists; of several reads aimed at one slot the last one stays.  param < 1 keeps a read when its uniform is below param."""
    take = [r for r in chunk if not (elist and r[0] in elist)]
    u = np.random.RandomState(s_seed).uniform(size=len(chunk) + 1)[:len(take)]
    if param < 1.:
        kept = np.flatnonzero(u < param)
        return s_reads + [[take[i][0], take[i][1], take[i][2]] for i in kept]
    size = int(param)
    if not s_reads:
        s_reads = [0] * size
    nth = cum_n_seq + 1 + np.arange(len(take), dtype=np.int64)
    slot = np.where(nth - 1 < size, nth - 1, (u * nth).astype(np.int64))
    hit = np.flatnonzero(slot < size)
    # the last read aimed at a slot wins: walk the hits backwards and keep the first sight of every slot
    _, first_from_back = np.unique(slot[hit][::-1], return_index=True)
    for i in hit[::-1][first_from_back]:
        s_reads[int(slot[i])] = [take[i][0], take[i][1], take[i][2]]
    return s_reads


def replace_masked(s_reads, exclude_seqs: Sequence[str], chunks: Iterable, logger=None):
    """longQC.py:369-406.  Reads of the subsample that sit on the highly-masked list are swapped for a second reservoir draw
    over the input -- one slot per masked read, skipping every read that was already picked or is masked itself.  The draw stops
    at the first chunk after which every slot is taken; if the input runs out first, the masked reads are dropped instead.
    `chunks` yields (reads, n_seqs, n_bases)."""
    picked = [r for r in s_reads if r != 0]
    masked = set(exclude_seqs)
    where = [i for i, r in enumerate(picked) if r[0] in masked]
    if not where:
        return picked
    avoid = masked | {r[0] for r in picked}
    spare = [0] * len(where)
    seen = 0
    for reads, n_seqs, _n_bases in chunks:
        subsample_from_chunk(reads, seen, spare, len(where), elist=avoid)
        seen += n_seqs
        if all(spare):
            break
    if not all(spare):
        gone = set(where)
        return [r for i, r in enumerate(picked) if i not in gone]
    for i, r in zip(where, spare):
        picked[i] = r
    return picked


def write_fastq(fn, reads, is_chunk=False):
    """lq_utils.py:352-369 (returns True on success, None when the file exists or there is nothing to write)"""
    if not is_chunk and os.path.isfile(fn):
        return None
    if len(reads) == 0:
        return None
    with open(fn, "a" if is_chunk else "w") as fq:
        for r in reads:
            fq.write("@%s\n%s\n+\n%s\n" % tuple(r))
    return True


def _to_arrays(reads):
    names = [r[0] for r in reads]
    seqs = [np.frombuffer(r[1].encode() if isinstance(r[1], str) else bytes(r[1]), dtype=np.uint8) for r in reads]
    quals = None
    if all(len(r) > 2 and r[2] for r in reads):
        quals = [np.frombuffer(r[2].encode() if isinstance(r[2], str) else bytes(r[2]), dtype=np.uint8) for r in reads]
    return names, seqs, quals


def coverage_in_memory(chunks: Iterable, s_reads, preset: str = "ont-ligation", fast: bool = False, inds: int = 4000000000,
                       out: Optional[str] = None, device: int = 0, engine=None):
    """The same computation as `LqExec(minimap2-coverage).exec(*coverage_argv(...))` without writing
    subsample.fastq and re-parsing both files (minimap2-coverage.c:408,471 parse the query file twice): the
    subsample goes to the device once, the input chunks (`chunks` yields (reads, n_seqs, n_bases), reads =
    [name, seq, qual]) are streamed into index parts cut by the reference's rule (index.c:244,311-316).
    Returns the table text (also written to `out` if given)."""
    from . import api, multigpu
    argv = coverage_argv(preset, "-", "-", fast=fast, inds=str(inds))
    p, _, _ = api.parse_args(argv)
    eng = engine or api.Engine(p, device=device)
    try:
        qn, qs, qq = _to_arrays([r for r in s_reads if r])
        eng.set_queries(qn, qs, qq)
        batch = int(p.batch_size)
        mini = min(int(p.idx_mini_batch), batch)
        part, part_bases, pend, pend_bases = None, 0, [], 0

        def flush_minibatch():
            nonlocal part, part_bases, pend, pend_bases
            if not pend:
                return
            if part is None:
                part = eng.part_begin(); part_bases = 0
            tn, ts, _ = _to_arrays(pend)
            eng.part_add_targets(part, tn, ts)
            part_bases += pend_bases
            pend, pend_bases = [], 0

        def close_part():
            nonlocal part, part_bases
            if part is not None:
                eng.part_build(part); eng.part_map(part); eng.part_release(part)
            part, part_bases = None, 0

        for (reads, n_seqs, n_bases) in chunks:
            for r in reads:
                if not pend and part is not None and part_bases > batch:      # checked before each mini-batch (index.c:244)
                    close_part()
                pend.append(r); pend_bases += len(r[1])
                if pend_bases >= mini:                                          # a mini-batch ends with the read that reaches the size
                    flush_minibatch()
        flush_minibatch()
        close_part()
        eng.finish()
        text = eng.table_text()
        if out:
            with open(out, "w") as f:
                f.write(text)
        return text
    finally:
        if engine is None:
            eng.close()
