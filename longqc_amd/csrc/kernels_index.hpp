// longqc_amd/csrc/kernels_index.hpp -- GPU-resident minimizer index of one part and seed collection.
//
// Index (reference index.c:150-236, semantic result "hash -> occurrences sorted by y"): the part's
// minimizers are produced in ascending y by k_sketch, a stable LSD radix sort on the 2k-bit hash
// groups them (rocPRIM device radix sort), run heads become the distinct keys, and an
// open-addressed, linearly probed table (16-B aligned key / start / count arrays, capacity a power
// of two >= 2*K_t) maps hash -> (start, n) into the sorted position array: one 8-byte probe per
// lookup in the common case, neighbouring lanes probing neighbouring query minimizers.
//
// Seeds (reference lqmap.c:140-205): per query minimizer one probe; occurrences >= mid_occ are
// skipped; survivors yield mini_pos entries and anchors in (minimizer, hit) order, minus the self
// diagonal.  Count, exclusive scan, fill -> anchors land in the reference's emission order.
#pragma once
#include "lq_common.hpp"
#include "kernels_sketch.hpp"

#define LQ_EMPTY_KEY LQ_U64MAX

__device__ __forceinline__ u64 lq_slot_of(u64 key, u32 cap_bits)
{	// Fibonacci hashing of the (already well mixed) minimizer hash
	return (key * 0x9E3779B97F4A7C15ULL) >> (64 - cap_bits);
}

// the sort key of a minimizer: its hash (x >> 8).  KT = u32 when the hash has at most 32 bits (k <= 16): the index sort then
// moves 12 instead of 16 bytes per minimizer and pass
template <class KT>
__global__ void k_sort_keys(const u64 *x, u64 n, KT *key)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) key[i] = (KT)(x[i] >> 8);
}

// y of a rank's share of a part, rid made part-global (multi-GPU: ranks sketch contiguous read ranges)
__global__ void k_rebase_y(const u64 *y, u64 n, u64 add, u64 *out)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = y[i] + add;
}

template <class KT>
__global__ void k_mark_heads(const KT *key, u64 n, u32 *head)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

// uidx = exclusive scan of head; every head writes its key and start
template <class KT>
__global__ void k_fill_unique(const KT *key, const u32 *head, const u64 *uidx, u64 n, u64 *ukey, u64 *ustart)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (head[i]) { ukey[uidx[i]] = (u64)key[i]; ustart[uidx[i]] = i; }
}

__global__ void k_unique_counts(const u64 *ustart, u64 n_keys, u64 n_mini, u32 *ucnt)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_keys) return;
	u64 e = i + 1 < n_keys ? ustart[i + 1] : n_mini;
	ucnt[i] = (u32)(e - ustart[i]);
}

__global__ void k_table_insert(const u64 *ukey, const u64 *ustart, const u32 *ucnt, u64 n_keys,
                               u64 *tkey, u64 *tstart, u32 *tcnt, u32 cap_bits)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_keys) return;
	u64 key = ukey[i], mask = ((u64)1 << cap_bits) - 1;
	u64 h = lq_slot_of(key, cap_bits);
	for (;;) {
		unsigned long long old = atomicCAS((unsigned long long*)&tkey[h], (unsigned long long)LQ_EMPTY_KEY, (unsigned long long)key);
		if (old == LQ_EMPTY_KEY) { tstart[h] = ustart[i]; tcnt[h] = ucnt[i]; return; }
		h = (h + 1) & mask;
	}
}

// mm_idx_get (index.c:69-86)
__device__ __forceinline__ u32 lq_table_get(const u64 *tkey, const u64 *tstart, const u32 *tcnt, u32 cap_bits, u64 key, u64 &start)
{
	u64 mask = ((u64)1 << cap_bits) - 1;
	u64 h = lq_slot_of(key, cap_bits);
	for (;;) {
		u64 kk = tkey[h];
		if (kk == key) { start = tstart[h]; return tcnt[h]; }
		if (kk == LQ_EMPTY_KEY) { start = 0; return 0; }
		h = (h + 1) & mask;
	}
}

// query of every query minimizer (qmoff has n_q+1 entries)
__global__ void k_minimizer_owner(const u64 *qmoff, u32 n_q, u64 n_qm, u32 *owner)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n_qm) owner[j] = lq_find_seg(qmoff, n_q, j);
}

// is target `rid` one of query q's same-name targets of this part? (lqmap.c:180-186: strcmp == 0)
__device__ __forceinline__ bool lq_is_self(const u32 *self_off, const u32 *self_rid, u32 q, u32 rid)
{
	for (u32 s = self_off[q]; s < self_off[q + 1]; ++s) if (self_rid[s] == rid) return true;
	return false;
}

// -X (MM_F_AVA, lqmap.c:187): a hit is dropped when strcmp(qname, tname) > 0.  t_rank[rid] = rank of the target's name
// among the part's distinct names, q_lo[q] = number of distinct target names below the query's: tname < qname <=> rank < q_lo.
struct AvaView { const u32 *t_rank, *q_lo; };     // both null without -X

// is y = (rid, pos, either strand) in the ascending occurrence list pos[st .. st+n)?  (index.c:188 keeps every list
// ascending in y = rid<<32 | pos<<1 | strand, and one (rid, pos) holds one minimizer: at most one entry matches)
__device__ __forceinline__ bool lq_list_has(const u64 *pos, u64 st, u32 n, u32 rid, u32 rpos)
{
	const u64 want = (u64)rid << 32 | (u64)rpos << 1;
	u32 lo = 0, hi = n;                                     // first entry >= want
	while (lo < hi) { const u32 mid = lo + ((hi - lo) >> 1); if (pos[st + mid] < want) lo = mid + 1; else hi = mid; }
	return lo < n && (pos[st + lo] >> 1) == (want >> 1);
}

// pass A of collect_seed_hits: probe, apply mid_occ, count surviving hits
__global__ void k_seed_probe(const u64 *qx, const u64 *qy, const u32 *owner, u64 n_qm,
                             const u64 *tkey, const u64 *tstart, const u32 *tcnt, u32 cap_bits, const u64 *pos,
                             i32 mid_occ, int no_self, const u32 *self_off, const u32 *self_rid, AvaView ava,
                             u64 *hit_start, u32 *hit_n, u32 *a_cnt, u32 *keep)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_qm) return;
	u64 st;
	u32 n = lq_table_get(tkey, tstart, tcnt, cap_bits, qx[j] >> 8, st);
	hit_start[j] = st; hit_n[j] = n;
	if ((i64)n >= (i64)mid_occ) { a_cnt[j] = 0; keep[j] = 0; return; }     // lqmap.c:166-173
	u32 c = n;
	const u32 q = owner[j];
	const u32 qpos = (u32)qy[j] >> 1;
	if (ava.t_rank) {                                       // -X: every hit has to be looked at (lqmap.c:187)
		const bool check_self = no_self && self_off[q] != self_off[q + 1];
		const u32 qlo = ava.q_lo[q];
		for (u32 t = 0; t < n; ++t) {
			u64 r = pos[st + t];
			if (check_self && ((u32)r >> 1) == qpos && lq_is_self(self_off, self_rid, q, (u32)(r >> 32))) --c;
			else if (ava.t_rank[(u32)(r >> 32)] < qlo) --c;
		}
	} else if (no_self && n) {
		// the self diagonal (lqmap.c:180-186): a hit on a same-name target at the query's own position.  The list is
		// ascending in (rid, pos), so each same-name target costs one binary search instead of a scan of the list.
		for (u32 s = self_off[q]; s < self_off[q + 1]; ++s) if (lq_list_has(pos, st, n, self_rid[s], qpos)) --c;
	}
	a_cnt[j] = c; keep[j] = 1;
}

// Query minimizers that share (hash, strand) with another anchor-bearing minimizer of the same query.  Two anchors of a
// query can only have the same x = (rev, rid, rpos) when two of its minimizers hit the same target occurrence, i.e.
// carry the same hash, and the same rev needs the same query strand: anchors of unmarked minimizers are unique in x,
// so any correct sort puts them where klib's unstable radix sort does (kernels_sort.hpp), and a query without a
// marked minimizer needs no klib-order walk at all.  One open-addressed table for the whole query set: slot -> j + 1.
__global__ void k_dup_mark(const u64 *qx, const u64 *qy, const u32 *owner, const u32 *a_cnt, u64 n_qm,
                           u32 *table, u32 tbits, u32 *dup, u32 *qdirty)
{
	const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_qm || a_cnt[j] == 0) return;
	const u64 key = qx[j] >> 8;
	const u32 strand = (u32)qy[j] & 1, q = owner[j];
	const u32 mask = (1u << tbits) - 1;
	u32 h = (u32)((((key << 1 | strand) ^ (u64)q * 0xD6E8FEB86659FD93ULL) * 0x9E3779B97F4A7C15ULL) >> (64 - tbits));
	for (;;) {
		const u32 old = atomicCAS(&table[h], 0u, (u32)j + 1);
		if (old == 0) return;
		const u32 jo = old - 1;
		if (owner[jo] == q && (qx[jo] >> 8) == key && ((u32)qy[jo] & 1) == strand) { dup[j] = 1; dup[jo] = 1; qdirty[q] = 1; return; }
		h = (h + 1) & mask;
	}
}

// pass B: anchors (lqmap.c:175-200) and mini_pos (lqmap.c:174)
// (a batch of queries = minimizers [j0, j0+nj); anchors are written relative to a_base)
// One query minimizer per lane for the set-up, which is parked in LDS; then the wave emits the anchors of one minimizer
// at a time, a hit per lane: the occurrence list is read and the anchors are written as contiguous runs (a thread
// walking its own list wrote 16-byte pieces 1 KiB apart: rocprofv3 counted 108 GB of HBM traffic per launch for 19 GB
// of anchors).  Anchors of minimizers marked by k_dup_mark carry LQ_TIE_MARK in y (never read downstream, like
// MM_SEED_TANDEM): the sort counts them to tell where klib's order can matter.  The anchors of a query that goes through
// klib's passes (qklib) are written to `originals` (the lane's second buffer) instead of `anchors`: the passes permute
// 8-byte records and gather every anchor once, into `anchors`, when its bucket is finished (kernels_rsort.hpp).
// A subset of the queries (the second pass of map_batch: the queries with a run in which klib's order of equal-x anchors can
// be observed): blockIdx.y = index into `sub`, the block's minimizers are those of query sub.q[blockIdx.y], the query's anchors
// go to sub.off[blockIdx.y] + (their place inside the query), to `originals` when sub.klib[...] says so; mini_pos is not
// written again.
#define LQ_EMIT_THREADS 256
struct EmitSetup { u64 st, out0; u32 n, q, span, qp, flags; i32 ql; };
struct EmitSub { const u32 *q; const u64 *off; const u32 *klib; };     // q == nullptr: all queries of the minimizer range [j0, j0 + nj)
__global__ void __launch_bounds__(LQ_EMIT_THREADS)
k_seed_emit(const u64 *qx, const u64 *qy, const u32 *owner, const u64 *qmoff, u64 j0, u64 nj,
            const u64 *pos, const u64 *hit_start, const u32 *hit_n, const u32 *keep, const u32 *dup,
            const u64 *a_off, u64 a_base, const u64 *mp_off, const u32 *qlen,
            int no_self, const u32 *self_off, const u32 *self_rid, AvaView ava,
            const u32 *qklib, mm128 *anchors, mm128 *originals, u64 *mini_pos, EmitSub sub)
{
	__shared__ EmitSetup su[LQ_EMIT_THREADS];
	u64 jt = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u32 lane = threadIdx.x & 63, w0 = threadIdx.x & ~63u;
	u64 sub_shift = 0; u32 sub_klib = 0;
	if (sub.q) {                                              // (block-uniform)
		const u32 sq = sub.q[blockIdx.y];
		j0 = qmoff[sq]; nj = qmoff[sq + 1] - j0;
		if ((u64)blockIdx.x * blockDim.x >= nj) return;
		a_base = 0;
		sub_shift = a_off[j0] - sub.off[blockIdx.y];          // the query's first anchor goes to sub.off[...]
		sub_klib = sub.klib[blockIdx.y];
	}
	const u64 j = j0 + (jt < nj ? jt : 0);
	const bool act = jt < nj && keep[j];
	EmitSetup e; e.st = 0; e.out0 = 0; e.n = 0; e.q = 0; e.span = 0; e.qp = 0; e.flags = 0; e.ql = 0;
	if (act) {
		e.q = owner[j];
		const u64 x = qx[j];
		e.span = (u32)(x & 0xff); e.qp = (u32)qy[j];
		if (!sub.q) mini_pos[mp_off[j]] = (u64)e.span << 32 | (e.qp >> 1);
		if (j > qmoff[e.q] && (qx[j - 1] >> 8) == (x >> 8)) e.flags |= 1;            // tandem
		if (j + 1 < qmoff[e.q + 1] && (qx[j + 1] >> 8) == (x >> 8)) e.flags |= 1;
		if (no_self && self_off[e.q] != self_off[e.q + 1]) e.flags |= 2;             // some target carries this query's name
		if (dup[j]) e.flags |= 4;
		if (sub.q ? sub_klib : qklib[e.q]) e.flags |= 8;                             // this query goes through klib's passes: its anchors are "originals"
		e.n = hit_n[j]; e.st = hit_start[j]; e.out0 = a_off[j] - a_base - sub_shift; e.ql = (i32)qlen[e.q];
	}
	su[threadIdx.x] = e;
	__syncthreads();
	for (u32 f = 0; f < 64; ++f) {
		const EmitSetup b = su[w0 + f];                         // the same entry in every lane of the wave
		if (b.n == 0) continue;
		const u32 b_qpos = b.qp >> 1;
		const u64 ybits = ((b.flags & 1) ? LQ_SEED_TANDEM : 0) | ((b.flags & 4) ? LQ_TIE_MARK : 0);
		const u64 y_same = (u64)b.span << 32 | b_qpos | ybits;
		const u64 y_rev = (u64)b.span << 32 | (u32)(b.ql - (i32)(b_qpos + 1 - b.span) - 1) | ybits;
		const bool filter = (b.flags & 2) || ava.t_rank;        // wave-uniform
		mm128 *out = (b.flags & 8) ? originals : anchors;
		u32 skipped = 0;
		for (u32 t0 = 0; t0 < b.n; t0 += 64) {
			const u32 t = t0 + lane;
			const bool valid = t < b.n;
			const u64 r = valid ? pos[b.st + t] : 0;
			const u32 rpos = (u32)r >> 1;
			u32 before = 0;
			bool skip = false;
			if (filter) {
				skip = valid && (((b.flags & 2) && rpos == b_qpos && lq_is_self(self_off, self_rid, b.q, (u32)(r >> 32))) ||
				                 (ava.t_rank && ava.t_rank[(u32)(r >> 32)] < ava.q_lo[b.q]));
				const u64 sm = __ballot(skip);
				before = skipped + (u32)__popcll(sm & ((1ULL << lane) - 1));
				skipped += (u32)__popcll(sm);
			}
			if (valid && !skip) {
				mm128 a;
				if ((r & 1) == (b.qp & 1)) { a.x = (r & 0xffffffff00000000ULL) | rpos; a.y = y_same; }
				else { a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | rpos; a.y = y_rev; }
				out[b.out0 + t - before] = a;
			}
		}
	}
}

// ---- anchors that cannot be part of a chain are never written --------------------------------------------------------------
// Against a 4-Gbase part a 10-kb query collects about a million seed hits, four fifths of them lone chance hits; they used to
// be written, sorted and read again only to be dropped.  Which hits can matter is decided first, exactly:
//   * mm_chain_dp never lets anchors of different (strand, rid) interact (kernels_chain.hpp), and inside one (strand, rid) run
//     anchor j is looked at by the scan of anchor i only when 0 < dq, dr <= max_gap and |dr - dq| <= bw (the `continue`s of
//     chain.c:52-56 come before any state changes).  dr - dq is the difference of the two anchors' diagonals d = x - y, so two
//     anchors that can interact lie at most bw apart in d, and the anchors of one connected component of "can interact" fill a
//     gap-free stretch of diagonal bins of width D > bw.
//   * a chain lives inside one component, needs min_cnt anchors and scores at most the sum of their spans (chain.c:57-67,
//     119-121): a component of fewer than n_min = max(min_cnt, ceil(min_sc / span_max)) anchors yields nothing and, being
//     invisible to the scans of every other component, can be left out without changing f, p, v of anything else.
//   So a hit survives iff the gap-free stretch of non-empty bins around its bin (strand, rid, d / D) holds at least n_min hits
//   (a stretch of n_min bins does by itself).  No false negatives; false positives (counters saturate at 3: a saturated bin is
//   taken as enough; bins alias when the diagonals of a pair outnumber the bins it gets) only cost what every hit used to cost.
// One block per query holds the 2-bit counters in 128 KiB of LDS and takes the rid range in slices of R targets x 2 strands x
// NB bins (NB >= 8 a power of two; slices so that a counter expects well under one hit).  Every occurrence list is ascending
// in rid (index.c:188), so a slice's hits are one contiguous piece of every list: a cursor per minimizer (global scratch)
// walks forward from slice to slice, and every hit is read twice in all -- once to count, once to decide -- eight lanes to a
// minimizer, 64 bytes a step.  Survivors are recorded as one bit per hit (a byte per step, written by the one group that owns
// the minimizer) and counted per minimizer; k_seed_emit_f then writes them, dense, in (query, minimizer, hit) order.
// avg_qspan, mini_pos and the lq_cnt_match prologue keep using the unfiltered totals (chain.c:37-38, lqmap.c:174).  Exact only
// together with a sort that does not need klib's walk over the *whole* query: map_batch's first pass.
#define LQ_FT_WORDS 32768u                  // 128 KiB of LDS: 524288 two-bit counters
#define LQ_FC_THREADS 1024
#ifndef LQ_FC_GROUP
#define LQ_FC_GROUP 8                        // lanes to a minimizer (8 or 16): one or two 64-byte lines of its occurrence list a step
#endif
#define LQ_FC_GMASK ((1u << LQ_FC_GROUP) - 1u)
#define LQ_FC_UNROLL 4                       // minimizers a group walks at a time
struct FiltParams { u32 n_min /* 0 or 1: no filter */, n_targets, keys_cap /* counters in use: a power of two in [256, 16 * LQ_FT_WORDS] (tests shrink it) */, a_cap /* hits per slice aimed at */, dshift /* log2 D, D > bw */,
                    split_strands /* 1: a set of bins per (target, strand); 0: the two strands of a target share its bins (twice the chance hits per bin, half the slices) */; };

__device__ __forceinline__ u32 lq_ft_get(const u32 *tab, u32 key) { return tab[key >> 4] >> ((key & 15) << 1) & 3u; }
__device__ __forceinline__ void lq_ft_inc(u32 *tab, u32 key)
{
	const u32 w = key >> 4, sh = (key & 15) << 1;
	u32 old = tab[w];
	for (;;) {
		if ((old >> sh & 3u) == 3u) return;                   // saturated
		const u32 seen = atomicCAS(&tab[w], old, old + (1u << sh));
		if (seen == old) return;
		old = seen;
	}
}
// does the gap-free stretch of non-empty bins around bin b of the NB bins at `base` hold n_min hits?  (bins wrap: aliasing only adds)
__device__ __forceinline__ bool lq_ft_alive(const u32 *tab, u32 base, u32 b, u32 nb_mask, u32 n_min)
{
	const u32 own = lq_ft_get(tab, base + b);
	if (own >= 3u || own >= n_min) return true;
	u32 side = n_min - 1;                                      // bins to look at on either side
	if (side > (nb_mask >> 1)) return true;                    // (a pair has too few bins to tell: keep)
	u32 tot = own;
	for (u32 k = 1; k <= side; ++k) { const u32 c = lq_ft_get(tab, base + ((b + k) & nb_mask)); if (c == 0) break; if (c >= 3u) return true; tot += c; if (k == side) return true; }
	for (u32 k = 1; k <= side; ++k) { const u32 c = lq_ft_get(tab, base + ((b - k) & nb_mask)); if (c == 0) break; if (c >= 3u) return true; tot += c; if (k == side) return true; }
	return tot >= n_min;
}

// the same for a pair with exactly 8 bins (16 bits of one table word) and n_min <= 4 -- the shape of every preset: no loops
__device__ __forceinline__ bool lq_ft_alive8(const u32 *tab, u32 base, u32 b, u32 n_min)
{
	const u32 f = tab[base >> 4] >> ((base & 15u) << 1) & 0xffffu;
	const u32 sh = ((b + 5u) & 7u) << 1;                       // rotate: bins b-3 .. b+3 to positions 0 .. 6
	const u32 g = (f >> sh | f << (16u - sh)) & 0xffffu;
	const u32 own = g >> 6 & 3u;
	const u32 r1 = g >> 8 & 3u, r2 = r1 ? g >> 10 & 3u : 0u, r3 = r2 ? g >> 12 & 3u : 0u;
	const u32 l1 = g >> 4 & 3u, l2 = l1 ? g >> 2 & 3u : 0u, l3 = l2 ? g & 3u : 0u;
	const u32 side = n_min - 1u;                               // 1 .. 3
	const u32 tot = own + r1 + l1 + (side > 1u ? r2 + l2 : 0u) + (side > 2u ? r3 + l3 : 0u);
	const bool sat = own == 3u || r1 == 3u || l1 == 3u || (side > 1u && (r2 == 3u || l2 == 3u)) || (side > 2u && (r3 == 3u || l3 == 3u));
	const bool reach = side == 1u ? (r1 | l1) != 0u : side == 2u ? (r2 | l2) != 0u : (r3 | l3) != 0u;   // n_min bins in a row
	return sat || reach || tot >= n_min;
}

// a query minimizer's occurrence list (start in pos[], length; 0: not kept), its y (position << 1 | strand), where the list goes
// on in the next slice of targets (k_seed_count's cursor), the byte its survivor bits start at
struct alignas(16) FMeta { u64 st; u32 n, qp; u64 fm_byte; u32 cursor, pad; };

// One sweep of one slice: the piece [cursor, first hit of a later slice) of every list of the query.  COUNT: the hits are
// counted per (rid, relative strand, diagonal bin); else: which of them survive (all of them without a filter), minus the self
// diagonal and -X, and the cursors move on.  A group of LQ_FC_GROUP lanes walks one piece at a time, LQ_FC_UNROLL pieces per
// group in flight ("workers": group x stream).  A worker takes the minimizers wid, wid + W, ... of the query and moves on to
// its next one as soon as a piece ends -- nobody waits for the longest piece of a turn (list lengths differ by two orders of
// magnitude); only the sweeps are separated by barriers.
struct FSlice { u32 r_lo, r_hi, bpp_log, nb_mask, rs_off; i32 ql; u32 q, qlo; bool filt, self_q; };
template <bool COUNT>
__device__ __forceinline__ void lq_seed_sweep(u32 *tab, FMeta *meta, const u64 *qx, u64 j0, u64 j1, const u64 *pos, const FSlice S, const FiltParams fp, u32 span_const,
                                              const u32 *self_off, const u32 *self_rid, AvaView ava, u8 *fmask, u32 *cntf)
{
	const u32 t = threadIdx.x, lane = t & 63, gl = t & (LQ_FC_GROUP - 1), gsh = lane & ~(u32)(LQ_FC_GROUP - 1);
	const u32 grp = t / LQ_FC_GROUP, n_grp = blockDim.x / LQ_FC_GROUP, W = n_grp * LQ_FC_UNROLL;
	const bool rare = S.self_q || ava.t_rank != nullptr;      // hits to be looked at one by one (lqmap.c:180-187)
	u64 jn[LQ_FC_UNROLL], st[LQ_FC_UNROLL];                   // next minimizer of the worker, its list
	u32 n[LQ_FC_UNROLL], qp[LQ_FC_UNROLL], c0[LQ_FC_UNROLL], c[LQ_FC_UNROLL], cnt[LQ_FC_UNROLL]; i32 ys[LQ_FC_UNROLL], yr[LQ_FC_UNROLL];
	u8 *fm[LQ_FC_UNROLL];
	bool more[LQ_FC_UNROLL], any = false;
#pragma unroll
	for (int u = 0; u < LQ_FC_UNROLL; ++u) { jn[u] = j0 + (u64)u * n_grp + grp; more[u] = false; n[u] = 0; qp[u] = 0; c0[u] = 0; c[u] = 0; cnt[u] = 0; st[u] = 0; ys[u] = 0; yr[u] = 0; fm[u] = fmask; any = any || jn[u] < j1; }
	while (__ballot(any)) {                                   // (every lane of the wave goes round until every worker of the wave is done)
		u64 r[LQ_FC_UNROLL];
#pragma unroll
		for (int u = 0; u < LQ_FC_UNROLL; ++u) {
			if (!more[u] && jn[u] < j1) {                         // the worker's next minimizer (uniform over the group)
				const FMeta m = meta[jn[u]];
				n[u] = m.n; st[u] = m.st; qp[u] = m.qp; c0[u] = m.cursor;
				const u32 span = span_const ? span_const : (u32)(qx[jn[u]] & 0xff);
				ys[u] = (i32)(m.qp >> 1) - S.ql - 256; yr[u] = S.ql - (i32)((m.qp >> 1) + 1 - span) - 1 - S.ql - 256;   // (the diagonal is taken relative to -qlen - 256: never negative)
				c[u] = m.cursor & ~(u32)(LQ_FC_GROUP - 1);        // steps are line-aligned in the list; hits before the cursor belong to earlier slices
				if (!COUNT) { cnt[u] = 0; fm[u] = fmask + m.fm_byte; }
				more[u] = c[u] < m.n;
				if (!more[u]) jn[u] += W;                         // nothing of it left for this slice
			}
			r[u] = more[u] && c[u] + gl < n[u] ? pos[st[u] + c[u] + gl] : ~0ULL;
		}
		any = false;
#pragma unroll
		for (int u = 0; u < LQ_FC_UNROLL; ++u) {
			const u32 tt = c[u] + gl, rid = (u32)(r[u] >> 32), rpos = (u32)r[u] >> 1;
			const bool valid = more[u] && tt < n[u];
			bool pass = valid && tt >= c0[u] && rid < S.r_hi;
			u32 key = 0, bin = 0;
			if (S.filt) {
				const u32 rs = ((u32)r[u] & 1u) ^ (qp[u] & 1u);
				const u32 d = (u32)((i32)rpos - (rs ? yr[u] : ys[u]));
				key = ((rid - S.r_lo) << S.bpp_log) + (rs ? S.rs_off : 0u); bin = (d >> fp.dshift) & S.nb_mask;
			}
			const u32 pastb = (u32)(__ballot(valid && rid >= S.r_hi) >> gsh) & LQ_FC_GMASK;   // hits of later slices
			const bool ends = more[u] && (pastb || c[u] + LQ_FC_GROUP >= n[u]);
			if (COUNT) {
				if (pass) lq_ft_inc(tab, key + bin);
			} else {
				if (pass && S.filt) pass = S.nb_mask == 7u && fp.n_min <= 4u ? lq_ft_alive8(tab, key, bin, fp.n_min) : lq_ft_alive(tab, key, bin, S.nb_mask, fp.n_min);
				if (rare && pass) {
					if (S.self_q && rpos == (qp[u] >> 1) && lq_is_self(self_off, self_rid, S.q, rid)) pass = false;   // lqmap.c:180-186
					if (pass && ava.t_rank && ava.t_rank[rid] < S.qlo) pass = false;                                      // lqmap.c:187
				}
				const u32 bits = (u32)(__ballot(pass) >> gsh) & LQ_FC_GMASK;
				if (gl == 0 && more[u]) {
					if (bits) {
						if (LQ_FC_GROUP == 8) fm[u][c[u] >> 3] |= (u8)bits; else *(u16*)(fm[u] + (c[u] >> 3)) |= (u16)bits;   // (c is a multiple of the group size; the bitmap of a minimizer starts on 8 bytes)
						cnt[u] += (u32)__popc(bits);
					}
					if (ends) {                                       // the piece is done: survivors of this slice, and where the next slice goes on
						if (cnt[u]) cntf[jn[u]] += cnt[u];
						meta[jn[u]].cursor = pastb ? c[u] + (u32)__ffs(pastb) - 1 : n[u];
					}
				}
			}
			if (ends) { more[u] = false; jn[u] += W; }
			c[u] += LQ_FC_GROUP;
			any = any || more[u] || jn[u] < j1;
		}
	}
}

// One block per query.
__global__ void __launch_bounds__(LQ_FC_THREADS)
k_seed_count(FMeta *meta, const u64 *qx, const u64 *qmoff, u32 q_lo, u32 q_hi, const u32 *qlen, const u64 *pos, const u64 *aq_off,
             int no_self, const u32 *self_off, const u32 *self_rid, AvaView ava, FiltParams fp, u32 span_const /* 0: from qx (-H) */,
             u8 *fmask, u32 *cntf)
{
	__shared__ u32 tab[LQ_FT_WORDS];
	const u32 t = threadIdx.x;
	const bool filt = fp.n_min >= 2;
	for (u32 q = q_lo + blockIdx.x; q < q_hi; q += gridDim.x) {
		const u64 j0 = qmoff[q], j1 = qmoff[q + 1];
		const u64 Aq = aq_off[q + 1] - aq_off[q];
		if (Aq == 0) continue;                                    // (block-uniform)
		// slices: R targets each, BPP bins per target (NB per strand, or NB shared by the two)
		u32 n_sl = 1, R = fp.n_targets, bpp_log = 0;
		if (filt) {
			const u64 by_bins = ((fp.split_strands ? 16ULL : 8ULL) * fp.n_targets + fp.keys_cap - 1) / fp.keys_cap, by_load = (Aq + fp.a_cap - 1) / fp.a_cap;
			u64 s = by_bins > by_load ? by_bins : by_load;
			if (s > fp.n_targets) s = fp.n_targets;
			if (s == 0) s = 1;
			n_sl = (u32)s;
			R = (fp.n_targets + n_sl - 1) / n_sl;
			while (bpp_log < (fp.split_strands ? 4u : 3u) || (bpp_log < 12 && ((u64)R << (bpp_log + 1)) <= fp.keys_cap)) ++bpp_log;
			while (((u64)R << bpp_log) > fp.keys_cap && R > 1) R = (R + 1) / 2;    // (more targets than the table has room for at 8 bins each: cannot happen after by_bins, kept as a guard)
			n_sl = (fp.n_targets + R - 1) / R;
		}
		FSlice S;
		S.bpp_log = bpp_log; S.filt = filt;
		S.nb_mask = filt ? (1u << (bpp_log - (fp.split_strands ? 1 : 0))) - 1 : 0;      // at least 8 bins
		S.rs_off = fp.split_strands ? S.nb_mask + 1 : 0;
		S.self_q = no_self && self_off[q] != self_off[q + 1];
		S.qlo = ava.q_lo ? ava.q_lo[q] : 0; S.q = q;
		S.ql = (i32)qlen[q];
		__syncthreads();
		for (u32 s = 0; s < n_sl; ++s) {
			S.r_lo = s * R; S.r_hi = s + 1 == n_sl ? 0xffffffffu : S.r_lo + R;
			if (filt) {
				const u32 words = (u32)((((u64)R << bpp_log) + 15) >> 4);
				for (u32 i = t; i < words; i += blockDim.x) tab[i] = 0;
				__syncthreads();
				lq_seed_sweep<true>(tab, meta, qx, j0, j1, pos, S, fp, span_const, self_off, self_rid, ava, fmask, cntf);
				__syncthreads();
			}
			lq_seed_sweep<false>(tab, meta, qx, j0, j1, pos, S, fp, span_const, self_off, self_rid, ava, fmask, cntf);
			__syncthreads();                                         // (the next slice clears the table; a cursor is read by its own group only)
		}
	}
}

// words of the survivor bitmap per query minimizer (scanned into fm_off)
__global__ void k_fmask_words(const u32 *hit_n, const u32 *keep, u64 n_qm, u32 *words)
{
	const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n_qm) words[j] = keep[j] ? (hit_n[j] + 63) >> 6 : 0;
}
__global__ void k_fmeta(const u32 *hit_n, const u32 *keep, const u64 *hit_start, const u64 *qy, const u64 *fm_off, u64 n_qm, FMeta *meta)
{
	const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_qm) return;
	FMeta m; m.st = hit_start[j]; m.n = keep[j] ? hit_n[j] : 0; m.qp = (u32)qy[j]; m.fm_byte = fm_off[j] * 8; m.cursor = 0; m.pad = 0;
	meta[j] = m;
}
// per query: where its surviving anchors start (af_off = exclusive scan of cntf)
// (for the queries [q_lo, q_hi] of a chunk whose minimizers end at j_end; n_total: the survivors up to and including the chunk)
__global__ void k_query_foff(const u64 *qmoff, const u64 *af_off, u32 q_lo, u32 q_hi, u64 j_end, u64 n_total, u64 *aqf_off)
{
	const u32 q = q_lo + blockIdx.x * blockDim.x + threadIdx.x;
	if (q > q_hi) return;
	const u64 j0 = qmoff[q];
	aqf_off[q] = j0 < j_end ? af_off[j0] : n_total;
}

// the surviving anchors (lqmap.c:175-200) of the minimizers [j0, j0 + nj), dense; and mini_pos (lqmap.c:174) of every kept
// minimizer.  Same shape as k_seed_emit: set-up per lane in LDS, then the wave writes one minimizer's survivors at a time.
__global__ void __launch_bounds__(LQ_EMIT_THREADS)
k_seed_emit_f(const u64 *qx, const u64 *qy, const u32 *owner, const u64 *qmoff, u64 j0, u64 nj,
              const u64 *pos, const u64 *hit_start, const u32 *hit_n, const u32 *keep, const u32 *dup,
              const u64 *fm_off, const u64 *fmask, const u32 *cntf, const u64 *af_off, u64 a_base, const u64 *mp_off, const u32 *qlen,
              mm128 *anchors, u64 *mini_pos)
{
	__shared__ EmitSetup su[LQ_EMIT_THREADS];
	__shared__ u64 fo[LQ_EMIT_THREADS];
	const u64 jt = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u32 lane = threadIdx.x & 63, w0 = threadIdx.x & ~63u;
	const u64 j = j0 + (jt < nj ? jt : 0);
	const bool act = jt < nj && keep[j];
	EmitSetup e; e.st = 0; e.out0 = 0; e.n = 0; e.q = 0; e.span = 0; e.qp = 0; e.flags = 0; e.ql = 0;
	u64 f0 = 0;
	if (act) {
		e.q = owner[j];
		const u64 x = qx[j];
		e.span = (u32)(x & 0xff); e.qp = (u32)qy[j];
		mini_pos[mp_off[j]] = (u64)e.span << 32 | (e.qp >> 1);
		if (cntf[j]) {
			if (j > qmoff[e.q] && (qx[j - 1] >> 8) == (x >> 8)) e.flags |= 1;            // tandem
			if (j + 1 < qmoff[e.q + 1] && (qx[j + 1] >> 8) == (x >> 8)) e.flags |= 1;
			if (dup[j]) e.flags |= 4;
			e.n = hit_n[j]; e.st = hit_start[j]; e.out0 = af_off[j] - a_base; e.ql = (i32)qlen[e.q];
			f0 = fm_off[j];
		}
	}
	su[threadIdx.x] = e; fo[threadIdx.x] = f0;
	__syncthreads();
	for (u32 f = 0; f < 64; ++f) {
		const EmitSetup b = su[w0 + f];                         // the same entry in every lane of the wave
		if (b.n == 0) continue;
		const u64 *fm = fmask + fo[w0 + f];
		const u32 b_qpos = b.qp >> 1;
		const u64 ybits = ((b.flags & 1) ? LQ_SEED_TANDEM : 0) | ((b.flags & 4) ? LQ_TIE_MARK : 0);
		const u64 y_same = (u64)b.span << 32 | b_qpos | ybits;
		const u64 y_rev = (u64)b.span << 32 | (u32)(b.ql - (i32)(b_qpos + 1 - b.span) - 1) | ybits;
		u32 done = 0;
		for (u32 t0 = 0; t0 < b.n; t0 += 64) {
			const u64 sv = fm[t0 >> 6];                             // (wave-uniform)
			if (sv == 0) continue;
			if (sv >> lane & 1) {
				const u64 r = pos[b.st + t0 + lane];
				const u32 rpos = (u32)r >> 1;
				mm128 a;
				if ((r & 1) == (b.qp & 1)) { a.x = (r & 0xffffffff00000000ULL) | rpos; a.y = y_same; }
				else { a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | rpos; a.y = y_rev; }
				anchors[b.out0 + done + (u32)__popcll(sv & ((1ULL << lane) - 1))] = a;
			}
			done += (u32)__popcll(sv);
		}
	}
}

// per query: anchor range, mini_pos range, avg_qspan (chain.c:37-38), lq_cnt_match prologue
// (esterr.c:85-97): skip flag and avg_k.
__global__ void __launch_bounds__(256)
k_query_prep(const u64 *qmoff, const u64 *a_off, const u64 *mp_off, u64 n_qm, u64 n_anchor_total, u64 n_mp_total, u32 n_q,
             const u64 *qx, const u32 *a_cnt, const u32 *keep, const u32 *qlen,
             u64 *aq_off, u64 *mpq_off, float *avg_qspan, const u64 *lambda, float *avg_k, u32 *skip, int covt_on, int mode /* 0: all; 1: offsets and avg_qspan only (the part's plan); 2: the skip verdict and avg_k only (when the part is mapped: they depend on the parts before) */)
{
	// one wave per query (the sums are integers: any order)
	const u32 q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (q > n_q) return;
	const u64 j0 = qmoff[q];
	if (lane == 0 && mode != 2) {
		aq_off[q] = j0 < n_qm ? a_off[j0] : n_anchor_total;
		mpq_off[q] = j0 < n_qm ? mp_off[j0] : n_mp_total;
	}
	if (q == n_q) return;
	const u64 j1 = qmoff[q + 1];
	u64 sum_span = 0, n_a = 0, sum_k = 0, n_mp = 0;
	for (u64 j = j0 + lane; j < j1; j += 64) {
		const u64 span = qx[j] & 0xff;
		sum_span += span * a_cnt[j]; n_a += a_cnt[j];
		if (keep[j]) { sum_k += span; ++n_mp; }
	}
	for (int o = 32; o > 0; o >>= 1) {
		sum_span += __shfl_xor(sum_span, o); n_a += __shfl_xor(n_a, o); sum_k += __shfl_xor(sum_k, o); n_mp += __shfl_xor(n_mp, o);
	}
	if (lane != 0) return;
	if (mode != 2) avg_qspan[q] = n_a ? __fdiv_rn((float)sum_span, (float)(i64)n_a) : 0.0f;
	if (mode == 1) return;
	u32 sk = 0;
	if (n_mp == 0) sk = 1;                                             // esterr.c:85
	else if (covt_on && lambda[q] / (u64)qlen[q] > LQ_COVT && avg_k[q] != 0.0f) sk = 1;   // esterr.c:87
	else if (avg_k[q] == 0.0f) avg_k[q] = __fdiv_rn((float)sum_k, (float)(i32)n_mp);   // esterr.c:93-97
	skip[q] = sk;
}

// mi->S of an index dump (index.c:278-284, mmpriv.h mm_seq4_set): 4 bits per base, 8 bases per word, all reads back to
// back (read r starts at base boff[r]); code = nt4 (0..3, 4 for anything else).  One thread per output word.
__global__ void k_seq4(const u64 *codes, const u32 *amb, const u64 *coff, const u64 *boff, u32 n_reads, u64 n_words, u32 *out)
{
	u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_words) return;
	const u64 total = boff[n_reads];
	u64 o = g * 8;
	u32 r = lq_find_seg(boff, n_reads, o);
	u32 w = 0;
	for (int j = 0; j < 8 && o < total; ++j, ++o) {
		while (o >= boff[r + 1]) ++r;                        // (empty reads are stepped over)
		const u64 p = o - boff[r];
		const u64 wi = coff[r] * LQ_CHUNK_WORDS + (p >> 5);
		const u32 b = (u32)(p & 31);
		const u32 c = (amb[wi] >> b & 1) ? 4u : (u32)(codes[wi] >> (2 * b) & 3);
		w |= c << (4 * j);
	}
	out[g] = w;
}
