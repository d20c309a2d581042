"""Seeded synthetic long-read sets for parity tests and bench.py.

No reference counterpart (the reference ships no test data); the *shape* of the data follows the
configs in BASELINE.json.  Everything is a pure function of the config and its seed, so the same
reads are regenerated bit-identically on the GPU box (no data files travel).

genome : i.i.d. uniform ACGT of length G, then ~1 % of G overwritten by exact 5-kb duplicates of
         other loci and ~0.1 % by short tandem repeats (exercises mid_occ, the equal-hash window
         ties of sketch.c:116-136 and repeated anchors).
reads  : start ~ U, strand ~ Bernoulli(.5), length ~ Gamma(shape, mean/shape) clipped to
         [min_len, 10*mean]; per-base independent errors (sub:ins:del as configured); a `junk`
         fraction of reads is random sequence.  Qualities uniform Q3..Q25, names r%07d.
query  : LongQC's seed-7 reservoir subsample (lq_utils.py:371-411) of `nsample` reads, applied to
         the whole set as one chunk; all reads if there are no more than nsample.
"""
from __future__ import annotations

import dataclasses
import gzip
from typing import List, Optional, Sequence, Tuple

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[ord("A")] = ord("T")
_COMP[ord("C")] = ord("G")
_COMP[ord("G")] = ord("C")
_COMP[ord("T")] = ord("A")
_COMP[ord("N")] = ord("N")


@dataclasses.dataclass
class SynthConfig:
    name: str
    n_reads: int
    mean_len: int
    depth: float
    seed: int
    gamma_shape: float = 2.0
    min_len: int = 500
    err: float = 0.10
    err_mix: Tuple[float, float, float] = (3.0, 3.0, 4.0)  # sub:ins:del
    junk_frac: float = 0.02
    nsample: int = 5000
    qual: str = "ont"  # "ont": uniform Q3..Q25 ; "none": '!' everywhere (Sequel CLR, lq_utils.py:248)
    n_frac: float = 0.0  # fraction of reads that get a short run of N (adversarial tests)

    @property
    def genome_len(self) -> int:
        return max(2000, int(self.n_reads * self.mean_len / self.depth))


# BASELINE.json configs[0..2]; 3/4 are 8-GPU throughput configs instantiated by bench.py per rank.
CONFIGS = {
    "cfg1": SynthConfig("cfg1", n_reads=1000, mean_len=10000, depth=5.0, seed=1001),
    "cfg2": SynthConfig("cfg2", n_reads=50000, mean_len=15000, depth=15.0, seed=1002),
    "cfg3": SynthConfig("cfg3", n_reads=500000, mean_len=10000, depth=30.0, seed=1003, err=0.13,
                        err_mix=(1.0, 6.0, 3.0), qual="none"),
    "cfg4": SynthConfig("cfg4", n_reads=5000000, mean_len=20000, depth=40.0, seed=1004),
    "cfg5": SynthConfig("cfg5", n_reads=1000000, mean_len=60000, depth=30.0, seed=1005, gamma_shape=1.2),
    # small ones for unit tests
    "tiny": SynthConfig("tiny", n_reads=120, mean_len=3000, depth=8.0, seed=7001, nsample=40, n_frac=0.1),
    "small": SynthConfig("small", n_reads=600, mean_len=6000, depth=10.0, seed=7002, nsample=150, n_frac=0.02),
}


# slices of the 8-GPU configs whose index parts have REAL size (-I 4G): what one GPU can be checked on against rows the
# reference printed in the build container (tests/golden/make_scale_golden.py -> tests/golden/<name>_rows.json)
SCALE_SLICES = {
    "cfg4s": dataclasses.replace(CONFIGS["cfg4"], name="cfg4s", n_reads=600000),    # 12 Gbases: three 4-Gbase parts and a rest
    "cfg5s": dataclasses.replace(CONFIGS["cfg5"], name="cfg5s", n_reads=75000),     # 4.5 Gbases of ~60-kb reads (N50 ~100 kb), 5000 queries ~ 300 Mbases
}


def make_genome(cfg: SynthConfig) -> np.ndarray:
    rng = np.random.default_rng(cfg.seed)
    G = cfg.genome_len
    g = _ACGT[rng.integers(0, 4, size=G, dtype=np.uint8)]
    # exact 5-kb duplicates (~1 % of G)
    dup = 5000
    if G > 4 * dup:
        for _ in range(max(1, int(0.01 * G / dup))):
            s = int(rng.integers(0, G - dup))
            d = int(rng.integers(0, G - dup))
            g[d:d + dup] = g[s:s + dup].copy()
    # short tandem repeats (~0.1 % of G): unit 1..6 bp, 100..400 bp long
    n_tr = max(1, int(0.001 * G / 200))
    for _ in range(n_tr):
        unit = _ACGT[rng.integers(0, 4, size=int(rng.integers(1, 7)), dtype=np.uint8)]
        ln = int(rng.integers(100, 401))
        d = int(rng.integers(0, max(1, G - ln)))
        rep = np.tile(unit, ln // len(unit) + 1)[:ln]
        g[d:d + len(rep)] = rep[: max(0, min(len(rep), G - d))]
    return g


def _mutate(seq: np.ndarray, rng: np.random.Generator, err: float, mix) -> np.ndarray:
    if err <= 0:
        return seq.copy()
    tot = float(sum(mix))
    p_sub, p_ins, p_del = (err * m / tot for m in mix)
    u = rng.random(seq.shape[0])
    is_sub = u < p_sub
    is_ins = (u >= p_sub) & (u < p_sub + p_ins)
    is_del = (u >= p_sub + p_ins) & (u < p_sub + p_ins + p_del)
    s = seq.copy()
    nsub = int(is_sub.sum())
    if nsub:
        # substitute by a *different* base: rotate within ACGT by 1..3
        code = np.searchsorted(_ACGT, s[is_sub])  # ACGT is sorted ascending in ASCII
        code = (code + rng.integers(1, 4, size=nsub)) & 3
        s[is_sub] = _ACGT[code]
    counts = np.ones(seq.shape[0], dtype=np.int64)
    counts[is_del] = 0
    counts[is_ins] = 2
    out = np.repeat(s, counts)
    nins = int(is_ins.sum())
    if nins:
        # the second copy of an inserted position becomes a random base
        ends = np.cumsum(counts)[is_ins] - 1
        out[ends] = _ACGT[rng.integers(0, 4, size=nins, dtype=np.uint8)]
    return out


@dataclasses.dataclass
class ReadSet:
    names: List[str]
    seqs: List[np.ndarray]  # uint8 ASCII
    quals: List[np.ndarray]  # uint8 ASCII (phred+33)

    def __len__(self) -> int:
        return len(self.names)

    @property
    def n_bases(self) -> int:
        return int(sum(int(s.shape[0]) for s in self.seqs))

    def subset(self, idx: Sequence[int]) -> "ReadSet":
        return ReadSet([self.names[i] for i in idx], [self.seqs[i] for i in idx], [self.quals[i] for i in idx])


def _one_read(cfg: SynthConfig, genome: np.ndarray, i: int, want_qual: bool = True):
    """read i of the config: its own RNG stream, so any slice of the set can be regenerated independently"""
    G = genome.shape[0]
    scale = cfg.mean_len / cfg.gamma_shape
    rng = np.random.default_rng([cfg.seed, 77, i])
    L = int(rng.gamma(cfg.gamma_shape, scale))
    L = max(cfg.min_len, min(L, 10 * cfg.mean_len, G))
    if rng.random() < cfg.junk_frac:
        s = _ACGT[rng.integers(0, 4, size=L, dtype=np.uint8)]
    else:
        st = int(rng.integers(0, G - L + 1))
        s = genome[st:st + L]
        if rng.random() < 0.5:
            s = _COMP[s[::-1]]
        s = _mutate(s, rng, cfg.err, cfg.err_mix)
    if cfg.n_frac > 0 and rng.random() < cfg.n_frac and s.shape[0] > 200:
        s = s.copy()
        p = int(rng.integers(0, s.shape[0] - 20))
        s[p:p + int(rng.integers(1, 20))] = ord("N")
    if not want_qual:
        q = None
    elif cfg.qual == "none":
        q = np.full(s.shape[0], ord("!"), dtype=np.uint8)
    else:
        q = (33 + rng.integers(3, 26, size=s.shape[0])).astype(np.uint8)
    return np.ascontiguousarray(s), q


def make_reads(cfg: SynthConfig, genome: Optional[np.ndarray] = None, n_reads: Optional[int] = None,
               read_offset: int = 0, indices: Optional[Sequence[int]] = None) -> ReadSet:
    """Generate reads [read_offset, read_offset + n_reads) of the config, or exactly `indices` (each
    read has its own RNG stream so that any slice of the set can be regenerated independently, e.g.
    one shard per rank, or just the query subsample)."""
    if genome is None:
        genome = make_genome(cfg)
    n = cfg.n_reads if n_reads is None else n_reads
    names, seqs, quals = [], [], []
    for i in (indices if indices is not None else range(read_offset, read_offset + n)):
        s, q = _one_read(cfg, genome, i)
        names.append("r%07d" % i)
        seqs.append(s)
        quals.append(q)
    return ReadSet(names, seqs, quals)


@dataclasses.dataclass
class FlatReads:
    """the same reads as one flat base array (the layout the C ABI takes); no qualities (index targets)"""
    first: int                 # index of the first read in the config
    flat: np.ndarray           # uint8 ASCII, all reads back to back
    off: np.ndarray            # uint64 [n+1]

    def __len__(self) -> int:
        return int(self.off.shape[0] - 1)

    @property
    def n_bases(self) -> int:
        return int(self.off[-1])

    def names(self) -> List[str]:
        return ["r%07d" % (self.first + i) for i in range(len(self))]

    def seq(self, i: int) -> np.ndarray:
        return self.flat[int(self.off[i]):int(self.off[i + 1])]


_W_CFG = None
_W_GENOME = None


def _flat_chunk(rng_):
    lo, hi = rng_
    seqs = [_one_read(_W_CFG, _W_GENOME, i, want_qual=False)[0] for i in range(lo, hi)]
    lens = np.array([s.shape[0] for s in seqs], dtype=np.uint64)
    return lo, (np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)), lens


def make_reads_flat(cfg: SynthConfig, genome: Optional[np.ndarray] = None, n_reads: Optional[int] = None,
                    read_offset: int = 0, workers: int = 0, chunk: int = 2000) -> FlatReads:
    """reads [read_offset, read_offset + n_reads) -- bit-identical to make_reads -- generated by `workers`
    forked processes (0: one per core, at most 64) straight into one flat array"""
    global _W_CFG, _W_GENOME
    import multiprocessing as mp
    import os
    if genome is None:
        genome = make_genome(cfg)
    n = cfg.n_reads if n_reads is None else n_reads
    if workers <= 0:
        workers = min(64, os.cpu_count() or 1)
    _W_CFG, _W_GENOME = cfg, genome
    jobs = [(lo, min(lo + chunk, read_offset + n)) for lo in range(read_offset, read_offset + n, chunk)]
    res = {}
    if workers > 1 and len(jobs) > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            for lo, flat, lens in pool.imap_unordered(_flat_chunk, jobs):
                res[lo] = (flat, lens)
    else:
        for j in jobs:
            lo, flat, lens = _flat_chunk(j)
            res[lo] = (flat, lens)
    lens = np.concatenate([res[lo][1] for lo, _ in jobs]) if jobs else np.zeros(0, np.uint64)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens, dtype=np.uint64)
    flat = np.empty(int(off[-1]), dtype=np.uint8)
    pos = 0
    for lo, _ in jobs:
        f = res.pop(lo)[0]
        flat[pos:pos + f.shape[0]] = f
        pos += f.shape[0]
    return FlatReads(read_offset, flat, off)


def write_flat_fasta(path: str, fr: FlatReads) -> None:
    with open(path, "wb") as f:
        for i in range(len(fr)):
            f.write(b">r%07d\n" % (fr.first + i))
            f.write(fr.seq(i).tobytes())
            f.write(b"\n")


def reservoir_subsample(n_total: int, num: int, s_seed: int = 7) -> List[int]:
    """Indices picked by LongQC's chunk reservoir (lq_utils.py:371-411) when the whole input is one
    chunk: h = uniform(size=n+1) after np.random.seed(7); slot d = n_seqs-1 while filling, else
    int(h[k]*n_seqs); replace iff d < num.  Returned in slot order (the order the FASTQ is written,
    longQC.py:417)."""
    if n_total <= num:
        return list(range(n_total))
    rs = np.random.RandomState(s_seed)
    h = rs.uniform(size=n_total + 1)
    slots = [0] * num
    n_seqs = 0
    for k in range(n_total):
        n_seqs += 1
        d = n_seqs - 1 if n_seqs - 1 < num else int(h[k] * n_seqs)
        if d < num:
            slots[d] = k
    return slots


def make_dataset(cfg: SynthConfig) -> Tuple[ReadSet, ReadSet]:
    """(targets = all reads, queries = seed-7 subsample)."""
    reads = make_reads(cfg)
    q = reads.subset(reservoir_subsample(len(reads), cfg.nsample))
    return reads, q


def write_fastq(path: str, rs: ReadSet, fasta: bool = False, line_width: int = 0, crlf: bool = False) -> None:
    eol = b"\r\n" if crlf else b"\n"
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wb") as f:
        for nm, s, q in zip(rs.names, rs.seqs, rs.quals):
            sb = s.tobytes()
            if fasta:
                f.write(b">" + nm.encode() + eol)
                if line_width > 0:
                    for i in range(0, len(sb), line_width):
                        f.write(sb[i:i + line_width] + eol)
                else:
                    f.write(sb + eol)
            else:
                f.write(b"@" + nm.encode() + eol + sb + eol + b"+" + eol + q.tobytes() + eol)
