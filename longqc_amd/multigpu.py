"""Two exact ways to put the path on N GPUs, one process per GPU, `torch.distributed` for the exchange:

1. QueryShardRunner (the default of bench.py --gpus N; BASELINE.json's north star): queries are split across the ranks,
   the index of every part is replicated.  Per part, rank r uploads a contiguous 1/N of the part's 2-bit packed reads, the
   ranks all-gather the packed reads (RCCL over xGMI: the only data-path collective, 0.375 B per base), every rank sketches the
   whole part and builds the same index -- hence the same mid_occ -- and maps its own queries; after the last part rank 0 gathers
   the rows.  (Rounds 4-5 all-gathered the minimizers of per-rank sketches: 16 B each, 21.5 GB per 4-Gbase part.)  Exact because every
   query owns its accumulators (minimap2-coverage.c:434, lqmap.c:788): nothing else crosses queries.
2. PartRunner: index parts across GPUs (below), for inputs of at least N parts.

Index parts across GPUs: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on
the GPU box, "gloo" in CPU tests) moves the per-part accumulators; every byte of compute stays in the HIP
engine behind the C ABI.

Why this sharding is exact.  The reference already splits the target set into index parts (-I,
index.c:244,311-316) and runs every query against every part in file order with persistent accumulators
(minimap2-coverage.c:449-458).  Within a part nothing depends on other parts except
  (i)   mid_occ, computed from part 0 only (map.c:50)                      -> broadcast from part 0's owner;
  (ii)  the COVT cap: part p is skipped for a query once lambda/qlen > 150 (esterr.c:87), lambda being the
        sum over the parts < p that were not skipped                       -> replayed here in part order
        from the all-gathered per-part lambdas (a few KB per part);
  (iii) avg_k, taken from the first part that sees the query (esterr.c:93-97) -> same replay.
Everything else a part contributes -- lambda, lambda2, the uint16 match counters (indexed by that part's
own filtered minimizer list, esterr.c:131-137) and the intervals that survive filter_redundant_coords for
that (query, part) (lqmap.c:287) -- is additive / concatenative, so parts can run concurrently on
different GPUs and be reduced: all_reduce(SUM) for the sums and counters, all_gather for the intervals.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

COVT = 150  # minimap2-coverage.h:20


# Collectives.  On the GPU box the backend is "nccl" (= RCCL over xGMI) and tensors stay on the device; the
# gloo backend (CPU tests, or two test ranks sharing one GPU) gets host staging for device tensors.
def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather(t: torch.Tensor, world: int, group) -> List[torch.Tensor]:
    if _staged(t, group):
        h = t.cpu()
        out = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(out, h, group=group)
        return [o.to(t.device) for o in out]
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return out


def _all_gather_into(t: torch.Tensor, world: int, group) -> torch.Tensor:
    """every rank's `t` (equal sizes) side by side in one tensor of world * len(t) entries: one collective, one receive buffer"""
    if _staged(t, group) or dist.get_backend(group) == "gloo":
        h = t.cpu()
        out = torch.empty(world * h.numel(), dtype=h.dtype)
        dist.all_gather(list(out.view(world, -1).unbind(0)), h, group=group)
        return out.to(t.device)
    out = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def _gather(t: torch.Tensor, world: int, rank: int, group) -> Optional[List[torch.Tensor]]:
    """rank 0 receives every rank's `t` (equal sizes); the others get None"""
    if _staged(t, group):
        h = t.cpu()
        out = [torch.empty_like(h) for _ in range(world)] if rank == 0 else None
        dist.gather(h, out, dst=0, group=group)
        return out
    out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, out, dst=0, group=group)
    return out


def _all_reduce(t: torch.Tensor, group, op=None) -> None:
    op = op if op is not None else dist.ReduceOp.SUM
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _broadcast(t: torch.Tensor, src: int, group) -> None:
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def covt_replay(lam_parts: torch.Tensor, avgk_parts: torch.Tensor, qlen: torch.Tensor,
                lam0: Optional[torch.Tensor] = None, avgk0: Optional[torch.Tensor] = None):
    """Replays lq_cnt_match's prologue (esterr.c:85-97) over parts in order.
    lam_parts [P, n_q] int64: lambda each part would add if it is not skipped; avgk_parts [P, n_q] float32:
    the part's avg_k candidate, 0 where the part has no usable minimizer for the query (n == 0).
    Returns (include [P, n_q] bool, lam_total, avg_k)."""
    P, n_q = lam_parts.shape
    lam = torch.zeros(n_q, dtype=torch.int64, device=lam_parts.device) if lam0 is None else lam0.clone()
    avgk = torch.zeros(n_q, dtype=torch.float32, device=lam_parts.device) if avgk0 is None else avgk0.clone()
    ql = qlen.to(torch.int64).clamp(min=1)
    inc = torch.zeros((P, n_q), dtype=torch.bool, device=lam_parts.device)
    for p in range(P):
        seen = avgk_parts[p] != 0                                 # n > 0 (esterr.c:85)
        capped = (torch.div(lam, ql, rounding_mode="floor") > COVT) & (avgk != 0)   # esterr.c:87 (integer division)
        use = seen & ~capped
        avgk = torch.where(use & (avgk == 0), avgk_parts[p], avgk)                 # esterr.c:93-97
        lam = lam + torch.where(use, lam_parts[p], torch.zeros_like(lam))
        inc[p] = use
    return inc, lam, avgk


def split_parts(lengths, batch_size: int, mini_batch: int = 50000000):
    """Read-index ranges [(start, end), ...] of the reference's index parts: mini-batches of min(50M, -I) bases
    (a mini-batch ends with the read that reaches the size, bseq.c:86-98) are appended while the running total
    is <= -I, checked before each mini-batch (index.c:244,311-316)."""
    chunk = min(int(mini_batch), int(batch_size))
    parts, i, n = [], 0, len(lengths)
    while i < n:
        start, total = i, 0
        while i < n and not total > batch_size:
            size = 0
            while i < n:
                size += int(lengths[i]); i += 1
                if size >= chunk:
                    break
            total += size
        parts.append((start, i))
    return parts


class PartRunner:
    """Drives one handle per rank through rounds of `world` concurrent parts and keeps the combined
    accumulators; `finalize()` imports them into the handle so that lqcov_finish() produces the rows."""

    @staticmethod
    def scaling_model(world: int, part_bases, upload_s_per_gbase=0.0056, sketch_s_per_gbase=0.0068, index_s_per_gbase=0.0083, map_s_per_gbase=0.0555) -> float:
        """seconds per job predicted for `world` GPUs: rounds of `world` consecutive parts, a round lasts as long as its largest
        part takes on one GPU (front + mapping of every query; the exchange is a few KB per part).  Same MI355X figures as
        QueryShardRunner.scaling_model (round 6)."""
        per = [(upload_s_per_gbase + sketch_s_per_gbase + index_s_per_gbase + map_s_per_gbase) * b / 1e9 for b in part_bases]
        return sum(max(per[i:i + world]) for i in range(0, len(per), world))

    def __init__(self, eng, world: int, rank: int, device: torch.device, query_lengths, group=None):
        self.eng, self.world, self.rank, self.dev, self.group = eng, world, rank, device, group
        eng.set_distributed(True)
        n_q, n_cnt, _ = eng.accum_sizes()
        self.n_q, self.n_cnt = n_q, n_cnt
        # the exchanged per-query arrays are in the engine's own query order (longest first): lengths in that order too
        ql = np.asarray(query_lengths, dtype=np.int64)
        assert ql.shape[0] == n_q
        self.qlen = torch.as_tensor(ql[eng.query_order().astype(np.int64)], device=self.dev)
        self.begin()

    def begin(self):
        """start a new job (== the reference binary starting): running totals to zero, mid_occ unset"""
        self.eng.set_mid_occ(-1)
        z = lambda n, dt: torch.zeros(max(n, 1), dtype=dt, device=self.dev)
        self.lam, self.lam2 = z(self.n_q, torch.int64), z(self.n_q, torch.int64)
        self.avgk, self.flags = z(self.n_q, torch.float32), z(self.n_q, torch.int32)
        self.cnts = z(self.n_cnt, torch.int32)
        self.ivl: List[torch.Tensor] = []
        self.sat = {}                                        # query (engine order) -> its counters, replayed in the reference's chain order (uint32 numpy)
        self.cnt_off = self.eng.counter_offsets().astype(np.int64) if self.n_q else np.zeros(1, np.int64)

    def share_mid_occ(self, owner_rank: int = 0):
        """mid_occ comes from part 0 (map.c:50): its owner broadcasts it after building."""
        t = torch.tensor([self.eng.mid_occ], dtype=torch.int32, device=self.dev)
        if self.world > 1:
            _broadcast(t, owner_rank, self.group)
        self.eng.set_mid_occ(int(t.item()))

    def map_and_combine(self, part: Optional[int], part_index: int, mid_occ_owner: int = 0, share_mid_occ: bool = True):
        """One round: this rank maps `part` (its handle-local id; None if it has no part this round), which is part
        `part_index` of the global order; all ranks then combine the round.  Parts of a round are consecutive:
        rank r holds part round_base + r."""
        eng, dev = self.eng, self.dev
        if share_mid_occ:
            self.share_mid_occ(mid_occ_owner)
        n_q, n_cnt = self.n_q, self.n_cnt
        lam, lam2 = torch.zeros(max(n_q, 1), dtype=torch.int64, device=dev), torch.zeros(max(n_q, 1), dtype=torch.int64, device=dev)
        avgk, flags = torch.zeros(max(n_q, 1), dtype=torch.float32, device=dev), torch.zeros(max(n_q, 1), dtype=torch.int32, device=dev)
        cnts, owner = torch.zeros(max(n_cnt, 1), dtype=torch.int32, device=dev), torch.zeros(max(n_cnt, 1), dtype=torch.int32, device=dev)
        ivl = torch.zeros((0, 3), dtype=torch.int32, device=dev)
        if part is not None:
            eng.reset()                                           # per-part accumulators (mid_occ survives in distributed mode)
            eng.part_map(part)
            _, _, n_ivl = eng.accum_sizes()
            ivl = torch.zeros((max(n_ivl, 1), 3), dtype=torch.int32, device=dev)
            eng.accum_export(lam.data_ptr(), lam2.data_ptr(), avgk.data_ptr(), flags.data_ptr(), cnts.data_ptr(), owner.data_ptr(), ivl.data_ptr())
            ivl = ivl[:n_ivl]
        # (ii)/(iii): replay the cap and avg_k over this round's parts, in part order
        if self.world > 1:
            lam_parts = torch.stack(_all_gather(lam, self.world, self.group))
            avgk_parts = torch.stack(_all_gather(avgk, self.world, self.group))
        else:
            lam_parts, avgk_parts = lam[None], avgk[None]
        inc, lam_tot, avgk_tot = covt_replay(lam_parts[:, :n_q], avgk_parts[:, :n_q], self.qlen, self.lam[:n_q], self.avgk[:n_q])
        mine = inc[self.rank] if part is not None else torch.zeros(n_q, dtype=torch.bool, device=dev)
        lam2c = torch.where(mine, lam2[:n_q], torch.zeros_like(lam2[:n_q]))
        cntc = torch.where(mine[owner[:n_cnt].long()], cnts[:n_cnt], torch.zeros_like(cnts[:n_cnt])) if n_cnt else cnts[:0]
        flagc = torch.where(mine, flags[:n_q], torch.zeros_like(flags[:n_q]))
        ivlc = ivl[mine[ivl[:, 0].long()]] if ivl.shape[0] else ivl
        if self.world > 1:
            lam2c = lam2c.contiguous(); _all_reduce(lam2c, self.group)
            if n_cnt:
                cntc = cntc.contiguous(); _all_reduce(cntc, self.group)
            flagc = flagc.contiguous(); _all_reduce(flagc, self.group, dist.ReduceOp.MAX)
            sizes = _all_gather(torch.tensor([ivlc.shape[0]], dtype=torch.int64, device=dev), self.world, self.group)
            mx = int(max(int(s.item()) for s in sizes))
            pad = torch.zeros((max(mx, 1), 3), dtype=torch.int32, device=dev)
            pad[:ivlc.shape[0]] = ivlc
            got = _all_gather(pad, self.world, self.group)
            ivlc = torch.cat([g[:int(s.item())] for g, s in zip(got, sizes)], dim=0)
        if n_cnt:
            self._replay_saturated(part, inc, cntc)
        self.lam[:n_q], self.avgk[:n_q] = lam_tot, avgk_tot
        self.lam2[:n_q] += lam2c
        if n_cnt:
            self.cnts[:n_cnt] += cntc
        self.flags[:n_q] = torch.maximum(self.flags[:n_q], flagc)
        self.ivl.append(ivlc)
        self.finalize()

    def _replay_saturated(self, part: Optional[int], inc: torch.Tensor, cntc: torch.Tensor):
        """esterr.c:127-138: the match counters are uint16 and the saturation tests read the counter of the chain's FIRST minimizer,
        so once a counter of a query is at 65535 its other counters depend on the order in which lq_cnt_match met the chains
        (hit.c:52-88).  Until then the 32-bit sums over the ranks are the reference's values.  A query one of whose merged counters
        reaches the limit in this round (or did in an earlier one) is chained once more against every included part of the round by
        the rank that holds the part, every kept chain recorded (lqcov_part_sat_records); the records go to all ranks (broadcast
        from the part's rank: the one exchange of this corner) and every rank replays them, part by part in part order, on the
        counters the query had before the round (lqcov_sat_replay) -- what one handle does by itself."""
        eng, dev, n_cnt = self.eng, self.dev, self.n_cnt
        cm = eng.counter_max()
        hot = torch.nonzero((self.cnts[:n_cnt].to(torch.int64) + cntc.to(torch.int64)) >= cm).flatten()
        new = set()
        if hot.numel():
            new = set(int(v) for v in np.unique(np.searchsorted(self.cnt_off[1:], hot.cpu().numpy(), side="right")))
        inc_any = inc.any(dim=0).cpu().numpy()
        todo = sorted(q for q in (new | set(self.sat)) if inc_any[q])
        if not todo:
            return
        if self.world > 1:
            have = [int(t.item()) for t in _all_gather(torch.tensor([1 if part is not None else 0], dtype=torch.int32, device=dev), self.world, self.group)]
        else:
            have = [1 if part is not None else 0]
        inc_h = inc.cpu().numpy()
        rb = int(eng.lib.lqcov_sat_record_bytes())
        for q in todo:
            lo, hi = int(self.cnt_off[q]), int(self.cnt_off[q + 1])
            c = self.sat[q] if q in self.sat else self.cnts[lo:hi].cpu().numpy().astype(np.uint32)
            for r in range(self.world):
                if not have[r] or not inc_h[r][q]:
                    continue
                if r == self.rank:
                    recs, at = eng.part_sat_records(part, q)
                    size = torch.tensor([recs.shape[0], at.shape[0]], dtype=torch.int64, device=dev)
                else:
                    recs = at = None
                    size = torch.zeros(2, dtype=torch.int64, device=dev)
                if self.world > 1:
                    _broadcast(size, r, self.group)
                    nr, na = int(size[0].item()), int(size[1].item())
                    tr = torch.zeros(max(nr * rb, 1), dtype=torch.uint8, device=dev)
                    ta = torch.zeros(max(na, 1), dtype=torch.int32, device=dev)
                    if r == self.rank:
                        tr[:nr * rb] = torch.from_numpy(recs.reshape(-1)).to(dev)
                        ta[:na] = torch.from_numpy(at.view(np.int32)).to(dev)
                    _broadcast(tr, r, self.group); _broadcast(ta, r, self.group)
                    recs = tr[:nr * rb].cpu().numpy().reshape(nr, rb)
                    at = ta[:na].cpu().numpy().view(np.uint32)
                c = eng.sat_replay(q, recs, at, c)
            self.sat[q] = c

    def finalize(self):
        ivl = torch.cat(self.ivl, dim=0).contiguous() if self.ivl else torch.zeros((0, 3), dtype=torch.int32, device=self.dev)
        self._ivl_keep = ivl if ivl.shape[0] else torch.zeros((1, 3), dtype=torch.int32, device=self.dev)
        self.eng.accum_import(self.lam.data_ptr(), self.lam2.data_ptr(), self.avgk.data_ptr(), self.flags.data_ptr(),
                              self.cnts.data_ptr(), self._ivl_keep.data_ptr(), int(ivl.shape[0]))
        for q, c in self.sat.items():                             # (replayed counters take the place of the merged sums; the rows carry LQCOV_ROW_REPLAYED)
            self.eng.accum_set_replayed(q, c)


# ---- queries sharded, index replicated (the north-star split) --------------------------------------------------
def balanced_ranges(lengths, world: int) -> List[tuple]:
    """world contiguous index ranges of about equal base totals (the ranks' shares of a part's reads)"""
    lens = np.asarray(lengths, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r // world, side="left")))
    cuts.append(len(lens))
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_queries(lengths, world: int) -> List[List[int]]:
    """query indices per rank: longest first, each to the rank with the fewest bases so far (anchors, and with them the
    mapping time, grow with the query's length); every rank's list ascending"""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i); load[r] += int(lengths[i])
    return [sorted(v) for v in out]


def exchange_peak_bytes(sizes, world: int) -> int:
    """HBM held at the peak of one part's minimizer exchange on a rank: per array (x and y) one send buffer of cap = max(sizes)
    words and one receive buffer of world * cap words, plus the part's own copy the engine makes from the shares (16 B per
    minimizer).  With shares cut by bases (within a percent of each other) that is (1 + 1/world) * 1.01 + 1 <= 2.2 times
    16 * M_t at world = 8."""
    cap = max(max(sizes), 1)
    ex = (world + 1) * cap * 16 if world > 1 else cap * 16
    return ex + 16 * int(sum(sizes))


class QueryShardRunner:
    """One engine handle per rank holding this rank's queries; map_part() replicates the index of one part and maps them."""

    def __init__(self, eng, world: int, rank: int, device: torch.device, group=None):
        self.eng, self.world, self.rank, self.dev, self.group = eng, world, rank, device, group
        self.my_queries: List[int] = []
        self.n_queries = 0

    def set_queries(self, names, seqs, quals=None):
        """every rank passes the whole query set; the handle receives this rank's share"""
        self.n_queries = len(names)
        self.names = list(names)
        self.my_queries = shard_queries([int(s.shape[0]) for s in seqs], self.world)[self.rank]
        self.eng.set_queries([names[i] for i in self.my_queries], [seqs[i] for i in self.my_queries],
                             [quals[i] for i in self.my_queries] if quals is not None else None)

    # ---- the parts in a pipeline (the reference's part loop, minimap2-coverage.c:449-458, overlapped) --------------------------
    # On N GPUs the front of a part -- upload and sketch of this rank's 1/N share, the all-gather, the replicated index build --
    # does not shrink with N the way the mapping does (queries are independent, minimap2-coverage.c:434: the mapping is 1/N).  So the
    # front of part i + 1 runs on a host thread of its own (the engine's build stream, a second part object, persistent
    # exchange buffers) while part i is mapped; a part then costs max(front, mapping) instead of their sum:
    #     t(N) ~ front(part 0) + sum over parts of max(front_i(N), map_i / N),   front_i(N) = upload_i / N + sketch_i / N + gather_i(N) + index_i
    # (scaling_model below).  Every rank's front thread issues the same collectives in the same order, the mapping issues none.
    def front(self, part: int, share, rid_base: int, all_names, all_lens):
        """the front of one part into part object `part`.  share = (PackedReads, lo, hi): this rank's contiguous share of the part's reads
        (reads [rid_base, ...) of the part, possibly none), 2-bit packed on the host -- the ranks all-gather the PACKED READS (0.375 B per
        base: 1.5 GB per 4-Gbase part), every rank then holds the whole part, sketches it and builds the same index (round 6).  share =
        a callable add_share(part): rounds 4-5's exchange -- the rank uploads and sketches its share, the MINIMIZERS are all-gathered
        (16 B each: 21.5 GB per part, 0.12 s on a ring of xGMI links to save a 53-ms sketch) -- kept for hosts that hold no packed reads."""
        eng = self.eng
        eng.part_clear(part)
        if callable(share):
            share(part)
            self._exchange_and_build(part, rid_base, all_names, all_lens, trim=False)
        else:
            self._exchange_packed_and_build(part, share, all_names, all_lens)

    def map_parts(self, parts, shares, pipeline: Optional[bool] = None):
        """parts: two part objects; shares: one (add_share, rid_base, all_names, all_lens) per index part, in file order.
        pipeline (default: on a GPU): the front of part i + 1 under the mapping of part i.  Returns the anchors of all parts."""
        import threading
        if pipeline is None:
            pipeline = self.dev.type == "cuda"
        shares = list(shares)
        anchors = 0
        if not shares:
            return 0
        self.front(parts[0], *shares[0])
        for i in range(len(shares)):
            th, err = None, []
            if i + 1 < len(shares):
                if pipeline:
                    def run(j=i + 1):
                        try:
                            if self.dev.type == "cuda":
                                torch.cuda.set_device(self.dev)
                            self.front(parts[j & 1], *shares[j])
                        except BaseException as e:       # (a thread's exception would vanish: carried to the caller)
                            err.append(e)
                    th = threading.Thread(target=run); th.start()
            self.eng.part_map(parts[i & 1])
            anchors += self.eng.last_n_anchors
            if th is not None:
                th.join()
                if err:
                    raise err[0]
            elif i + 1 < len(shares):
                self.front(parts[(i + 1) & 1], *shares[i + 1])
        return anchors

    @staticmethod
    def scaling_model(world: int, part_bases, upload_s_per_gbase=0.0056, sketch_s_per_gbase=0.0068, index_s_per_gbase=0.0083, map_s_per_gbase=0.0555,
                      link_gbytes_per_s=153.0, packed_bytes_per_base=0.25, pipelined=True) -> float:
        """seconds per job predicted for `world` GPUs (the defaults: MI355X figures measured at configs[2] at the end of round 6 -- upload
        22.5 ms (codes alone: reads without an N), sketch 27 ms, index build incl. sort, run heads, table and name work 39 ms, seed plan +
        mapping 0.22 s per 4.0 Gbases and 5000 queries; ring all-gather of the packed reads, 0.25 B per base without ambiguity words (0.375
        with), bound by one xGMI link).  The front of a part: this rank's 1 / N of the upload, the all-gather, then sketch and index of the
        WHOLE part on every rank.  The mapping's rate is configs[2]'s; a 40x ONT set takes three times as long per base (profiles/README.md,
        the full-size configs[3] run), which only helps the split that shares the mapping.  tools/scale.sh prints it beside what it measures."""
        t, prev_map = 0.0, None
        for b in part_bases:
            g = b / 1e9
            gather = 0.0 if world == 1 else (world - 1) / world * (g * 1e9 * packed_bytes_per_base) / (link_gbytes_per_s * 1e9)
            front = upload_s_per_gbase * g / world + gather + (sketch_s_per_gbase + index_s_per_gbase) * g
            mp = map_s_per_gbase * g / world
            if prev_map is None or not pipelined:
                t += front if prev_map is None else front + prev_map
            else:
                t += max(front, prev_map)
            prev_map = mp
        return t + (prev_map or 0.0)

    def _ensure_exchange(self, cap: int):
        """send (cap words) and receive (world * cap words) buffers of the minimizer exchange, kept from part to part"""
        have = getattr(self, "_ex_cap", 0)
        if cap > have:
            self._ex = None
            cap2 = cap + cap // 16 + 1024
            self._ex = [torch.empty(cap2, dtype=torch.int64, device=self.dev) for _ in range(2)] + \
                       ([torch.empty(self.world * cap2, dtype=torch.int64, device=self.dev) for _ in range(2)] if self.world > 1 else [])
            self._ex_cap = cap2
        return self._ex

    def _exchange_packed_and_build(self, part: int, share, all_names, all_lens):
        """all-gather of the ranks' packed shares (codes: 4 x u64 per 128-base chunk, ambiguity bits: 4 x u32), the part from the receive
        buffers (lqcov_part_add_packed_shares_dev: the shares back to back = the reads in file order, every read starts on a chunk), then
        the ordinary build: sketch + index (+ mid_occ on the first part) -- the same reads on every rank, hence the same index"""
        import ctypes
        eng, dev, world = self.eng, self.dev, self.world
        packed, lo, hi = share
        c0, c1 = int(packed.coff[lo]), int(packed.coff[hi])
        n_ch = c1 - c0
        mine_amb = 1 if (n_ch and packed.any_ambiguous(lo, hi)) else 0      # (a part without an N: the codes alone are exchanged, 0.25 B per base)
        if world > 1:
            got = [t.cpu().tolist() for t in _all_gather(torch.tensor([n_ch, mine_amb], dtype=torch.int64, device=dev), world, self.group)]
            sizes = [int(g[0]) for g in got]
            any_amb = any(int(g[1]) for g in got)
        else:
            sizes = [n_ch]
            any_amb = bool(mine_amb)
        cap = max(max(sizes), 1)
        have = getattr(self, "_pk_cap", 0)
        if cap > have:                                            # send and receive buffers, kept from part to part
            self._pk = None
            cap2 = cap + cap // 16 + 64
            self._pk = [torch.empty(cap2 * 4, dtype=torch.int64, device=dev), torch.empty(cap2 * 4, dtype=torch.int32, device=dev)] + \
                       ([torch.empty(world * cap2 * 4, dtype=torch.int64, device=dev), torch.empty(world * cap2 * 4, dtype=torch.int32, device=dev)] if world > 1 else [])
            self._pk_cap = cap2
        stride = self._pk_cap
        sc, sa = self._pk[0], self._pk[1]
        if n_ch:
            hc = np.ctypeslib.as_array(ctypes.cast(packed.codes_ptr + c0 * 32, ctypes.POINTER(ctypes.c_int64)), shape=(n_ch * 4,))
            ha = np.ctypeslib.as_array(ctypes.cast(packed.amb_ptr + c0 * 16, ctypes.POINTER(ctypes.c_int32)), shape=(n_ch * 4,))
            sc[:n_ch * 4].copy_(torch.from_numpy(hc))                                                  # this rank's 1 / N of the upload
            if any_amb:
                sa[:n_ch * 4].copy_(torch.from_numpy(ha))
        self.last_sizes = sizes
        self.last_any_amb = any_amb
        self.last_exchange_bytes = (world + 1) * stride * 48 if world > 1 else stride * 48           # (tests assert the bound: 0.375 B per base and buffer)
        if world > 1:
            gc, ga = self._pk[2], self._pk[3]
            if dist.get_backend(self.group) == "gloo":
                gc.copy_(_all_gather_into(sc, world, self.group))
                if any_amb:
                    ga.copy_(_all_gather_into(sa, world, self.group))
            else:
                dist.all_gather_into_tensor(gc, sc, group=self.group)                                # RCCL all-gather over xGMI
                if any_amb:
                    dist.all_gather_into_tensor(ga, sa, group=self.group)
                torch.cuda.current_stream(dev).synchronize()
        else:
            gc, ga = sc, sa
        eng.part_add_packed_shares_dev(part, gc.data_ptr(), ga.data_ptr() if any_amb else 0, stride, sizes, np.asarray(all_lens, dtype=np.uint32), all_names)
        eng.part_build(part)

    def map_part(self, part: int, rid_base: int, all_names, all_lens):
        """`part` holds this rank's contiguous share of the index part's reads (reads [rid_base, ...) of the part, possibly
        none); all_names / all_lens describe every read of the part in order.  Sketch the share, all-gather the minimizers,
        build the replicated index, map this rank's queries.  The part can be cleared / released afterwards."""
        self._exchange_and_build(part, rid_base, all_names, all_lens, trim=True)
        self.eng.part_map(part)

    def _exchange_and_build(self, part: int, rid_base: int, all_names, all_lens, trim: bool):
        eng, dev, world = self.eng, self.dev, self.world
        eng.part_sketch(part)
        n_mine = eng.part_n_minimizers(part)
        if world > 1:
            sizes = _all_gather(torch.tensor([n_mine], dtype=torch.int64, device=dev), world, self.group)
            sizes = [int(t.item()) for t in sizes]
        else:
            sizes = [n_mine]
        # Equal send buffers of `cap` words (the shares are cut by bases, so they differ by a percent), one receive buffer of
        # world * cap words per array, filled in place by all_gather_into_tensor; the engine copies the shares back to back
        # into the part (lqcov_part_build_from_minimizer_shares_dev): no concatenated copy here.  Peak of the exchange:
        # (1 + 1 / world) * 16 * M_t * (cap * world / M_t) bytes here + the part's own 16 * M_t.
        cap = max(max(sizes), 1)
        self.last_sizes = sizes
        self.last_exchange_bytes = exchange_peak_bytes(sizes, world)      # (tests assert the bound)
        if trim:
            if dev.type == "cuda" and world > 1:
                # the lanes' work space of the previous part is still held by the engine's pool, which torch's allocator cannot
                # draw from: give it back first if what is free would not do
                need = self.last_exchange_bytes
                if torch.cuda.mem_get_info(dev)[0] < need + need // 4:
                    eng.workspace_trim()
            xs = torch.empty(cap, dtype=torch.int64, device=dev); ys = torch.empty(cap, dtype=torch.int64, device=dev)
            eng.part_minimizers_export(part, xs.data_ptr(), ys.data_ptr(), cap, rid_base)
            if world > 1:
                gx = _all_gather_into(xs, world, self.group); gy = _all_gather_into(ys, world, self.group)   # RCCL all-gather over xGMI
                del xs, ys
            else:
                gx, gy = xs, ys
            stride = cap
        else:
            # pipelined: buffers kept from part to part (no allocation, trim or cache flush beside a running mapping), the
            # gather straight into the persistent receive buffers
            ex = self._ensure_exchange(cap)
            stride = self._ex_cap
            xs, ys = ex[0], ex[1]
            eng.part_minimizers_export(part, xs.data_ptr(), ys.data_ptr(), stride, rid_base)
            if world > 1:
                gx, gy = ex[2], ex[3]
                if dist.get_backend(self.group) == "gloo":
                    gx.copy_(_all_gather_into(xs, world, self.group)); gy.copy_(_all_gather_into(ys, world, self.group))
                else:
                    dist.all_gather_into_tensor(gx, xs, group=self.group); dist.all_gather_into_tensor(gy, ys, group=self.group)   # RCCL all-gather over xGMI
                    torch.cuda.current_stream(dev).synchronize()
            else:
                gx, gy = xs, ys
        eng.part_clear(part)                                      # the share's reads are no longer needed: the part becomes the whole index part
        eng.part_build_from_minimizer_shares_dev(part, gx.data_ptr(), gy.data_ptr(), stride, sizes, np.asarray(all_lens, dtype=np.uint32), all_names)
        if trim:
            del gx, gy
            if dev.type == "cuda":
                torch.cuda.empty_cache()                          # the exchange buffers go back to the device: the mapping sizes its work space from what is free

    def gather_table(self) -> Optional[str]:
        """finish on every rank; rank 0 returns the table of all queries in the caller's order, the others None.  What travels
        is binary: every rank's `lqcov_row` array, its two region pools and its query indices, as padded uint8 tensors through
        one gather each (RCCL on the GPU box); rank 0 rebases the region offsets, orders the rows and prints them with the
        engine's own formatter (lqcov_format_rows: the reference's printf arithmetic in one place)."""
        from . import api as _api
        self.eng.finish()
        if self.world == 1:
            return self.eng.table_text()
        rows, regs, mregs = self.eng.rows_binary()
        assert rows.shape[0] == len(self.my_queries)
        idx = np.asarray(self.my_queries, dtype=np.int64)
        dev = self.dev
        parts = [rows.reshape(-1), regs.reshape(-1).view(np.uint8), mregs.reshape(-1).view(np.uint8), idx.view(np.uint8)]
        sizes = torch.tensor([p.shape[0] for p in parts], dtype=torch.int64, device=dev)
        all_sizes = torch.stack(_all_gather(sizes, self.world, self.group)).cpu().numpy()        # [world, 4]
        got = []
        for j, p in enumerate(parts):
            cap = int(all_sizes[:, j].max())
            buf = torch.zeros(max(cap, 1), dtype=torch.uint8, device=dev)
            if p.shape[0]:
                buf[:p.shape[0]] = torch.from_numpy(np.ascontiguousarray(p)).to(dev)
            got.append(_gather(buf, self.world, self.rank, self.group))
        if self.rank != 0:
            return None
        row_sz = rows.shape[1] if rows.ndim == 2 and rows.shape[0] else _api.C.sizeof(_api.Row)
        all_rows, all_regs, all_mregs, all_idx = [], [], [], []
        reg_base = mreg_base = 0
        for r in range(self.world):
            nb = [int(v) for v in all_sizes[r]]
            rr = got[0][r][:nb[0]].cpu().numpy().reshape(-1, row_sz).copy()
            rg = got[1][r][:nb[1]].cpu().numpy().view(np.uint32).reshape(-1, 2)
            mg = got[2][r][:nb[2]].cpu().numpy().view(np.uint32).reshape(-1, 2)
            ix = got[3][r][:nb[3]].cpu().numpy().view(np.int64)
            if rr.shape[0]:
                st = np.frombuffer(rr.tobytes(), dtype=_ROW_DTYPE).copy()
                st["reg_off"] += reg_base; st["mreg_off"] += mreg_base
                rr = np.frombuffer(st.tobytes(), dtype=np.uint8).reshape(-1, row_sz)
            all_rows.append(rr); all_regs.append(rg); all_mregs.append(mg); all_idx.append(ix)
            reg_base += rg.shape[0]; mreg_base += mg.shape[0]
        rows_cat = np.concatenate(all_rows) if all_rows else np.zeros((0, row_sz), np.uint8)
        idx_cat = np.concatenate(all_idx)
        order = np.argsort(idx_cat, kind="stable")
        assert idx_cat.shape[0] == self.n_queries and np.array_equal(idx_cat[order], np.arange(self.n_queries))
        return _api.format_rows(self.eng.lib, int(self.eng.params.filter_flag), rows_cat[order], np.concatenate(all_regs), np.concatenate(all_mregs),
                                [self.names[i] for i in range(self.n_queries)])


# numpy view of lqcov_row (include/lqcov.h): the gatherer rebases the two region offsets
_ROW_DTYPE = np.dtype([("lambda_", "<u8"), ("lambda2", "<u8"), ("qual_psum", "<f8"), ("qlen", "<u4"), ("n_mini", "<u4"), ("n_match", "<u4"), ("avg_k", "<f4"),
                       ("reg_off", "<u4"), ("n_reg", "<u4"), ("mreg_off", "<u4"), ("n_mreg", "<u4"), ("has_qual", "<u4"), ("flags", "<u4")])
