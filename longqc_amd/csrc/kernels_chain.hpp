// longqc_amd/csrc/kernels_chain.hpp -- co-linear chaining and coverage accumulation.
//
// mm_chain_dp (reference chain.c:22-157) never lets anchors of different (strand, target) interact:
// x carries rev and rid in its high 32 bits, so the `st` window (chain.c:47), the DP, the peak
// bookkeeping and the backtrack are independent per (strand, rid) run of the sorted anchors; only
// avg_qspan is global to the query (chain.c:37-38) and is computed from the unfiltered totals in
// k_query_prep.  Runs shorter than min_cnt can never yield a chain (chain.c:119-121) and are
// skipped.  One thread owns one run and executes the reference's loop order literally (the
// accept / skip / break logic of chain.c:48-77 is order dependent), then turns each chain into
// the reg coordinates of mm_reg_set_coor (hit.c:23-38) and applies lq_cnt_match (esterr.c:99-138)
// with atomics into the per-query accumulators -- sums and counters commute, so the unspecified
// order of chains is unobservable (the uint16 saturation quirk of esterr.c:130,136 is detected
// and flagged instead).
#pragma once
#include "lq_common.hpp"
#include "kernels_sketch.hpp"
#include "kernels_sort.hpp"   // LQ_BLOCK_LOOP / LQ_BLOCK_SYNC / LQ_SHARED

// ---- the list of the (strand, rid) runs that can hold a chain, in one pass over the sorted anchors ------------------------------
// A run starts at the first anchor of the batch, at the first anchor of every query and wherever the high word of x (strand,
// rid) changes.  Against half a million targets most runs are one or two chance hits; a chain needs min_cnt anchors
// (chain.c:119-121) and scores at most the sum of its anchors' spans (chain.c:57-67), each at most k (255 with -H), so a run
// of fewer than n_min = max(min_cnt, ceil(min_sc / span_max)) anchors is never looked at again.  Per 4096-anchor tile: heads
// by ballots, the heads as a bitmap in LDS, a head's length = distance to the next head (the last run of a tile, when it
// goes on past the tile: to the end of its query or to the first different high word -- the query's anchors are sorted),
// entries start | length << 32.  The list is in no particular order; nothing downstream depends on it (sums, counters, a
// pool of intervals that is sorted before use).  k_run_list below has the details.
#define LQ_RUN_TILE 4096
#define LQ_RUN_THREADS 256
#define LQ_RUN_ROWS (LQ_RUN_TILE / LQ_RUN_THREADS)
#define LQ_RUN_WAVES (LQ_RUN_THREADS / 64)     // (k_sel_write's rows and waves)

// exclusive scan of n counts in place, the total in cnt[n]; one block
#define LQ_TSCAN_THREADS 1024
__global__ void __launch_bounds__(LQ_TSCAN_THREADS)
k_tile_scan(u32 *cnt, u32 n)
{
	__shared__ u32 wsum[LQ_TSCAN_THREADS / 64];
	const u32 t = threadIdx.x, lane = t & 63, w = t >> 6;
	const u32 per = (n + LQ_TSCAN_THREADS - 1) / LQ_TSCAN_THREADS;
	const u32 a = (u64)t * per < n ? t * per : n, b = a + per < n ? a + per : n;
	u32 sum = 0;
	for (u32 x = a; x < b; ++x) sum += cnt[x];
	u32 inc = sum;
	for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
	if (lane == 63) wsum[w] = inc;
	__syncthreads();
	if (t == 0) { u32 run = 0; for (u32 x = 0; x < LQ_TSCAN_THREADS / 64; ++x) { const u32 v = wsum[x]; wsum[x] = run; run += v; } cnt[n] = run; }
	__syncthreads();
	u32 run = wsum[w] + inc - sum;
	for (u32 x = a; x < b; ++x) { const u32 v = cnt[x]; cnt[x] = run; run += v; }
}

// strand and rid of an anchor: the high word of x, loaded alone
__device__ __forceinline__ u32 lq_hi32(const mm128 *a) { return ((const u32*)a)[1]; }
#define LQ_RUN_START(e) ((u64)(u32)(e))
#define LQ_RUN_LEN(e) ((i64)((e) >> 32))
#define LQ_RUN_STAGE 2048         // entries a block collects in LDS before it reserves their place in the list
#define LQ_RUN_PEEK 64            // anchors after the tile that the block looks at for the end of the tile's last run
// A block takes a contiguous stretch of tiles and collects the entries in LDS: one atomic on the list's counter per ~1500
// entries (half a dozen tiles at configs[2]).  What a tile costs is its chain of dependent loads and the waves a CU can hold,
// not its bytes: the tile's first query comes from the previous tile (a bisection for the block's first tile only), the end
// of the tile's last run from LQ_RUN_PEEK anchors loaded with the tile (a bisection inside the query only for a run that goes
// on beyond those); heads and entries go through LDS (a bitmap, a staging area filled through an LDS counter), so that a
// thread holds sixteen high words and little else (a first version kept ballots and lengths of all rows in registers: 184
// VGPRs, two waves per SIMD, 5 ms per launch against 0.9 ms for reading the anchors).
__global__ void __launch_bounds__(LQ_RUN_THREADS)
k_run_list(const mm128 *A, u64 n, const u64 *aq_off, u64 a_base, u32 n_q, u32 n_tiles, u32 n_min, u32 stage /* 256 .. LQ_RUN_STAGE (tests shrink it) */,
           u32 *n_runs, u64 *runs)
{
	__shared__ u32 qbits[LQ_RUN_TILE / 32];                       // query starts inside the tile
	__shared__ u32 hb[LQ_RUN_TILE / 32 + 2];                      // run starts inside the tile
	__shared__ u32 nx[LQ_RUN_PEEK];
	__shared__ u32 slot0, fill_s;
	__shared__ u64 stg[LQ_RUN_STAGE];
	const u32 t = threadIdx.x, lane = t & 63;
	const u32 per = (n_tiles + gridDim.x - 1) / gridDim.x;
	const u32 T0 = blockIdx.x * per < n_tiles ? blockIdx.x * per : n_tiles, T1 = T0 + per < n_tiles ? T0 + per : n_tiles;
	const u64 near_mask = n_min > 1 ? (n_min > 33 ? ~0ULL >> 31 : (1ULL << (n_min - 1)) - 1) : 0;   // the next n_min - 1 positions (up to 33 of them)
	u32 qlo = ~0u;                                                // the query that holds the tile's first anchor
	if (t < 2) hb[LQ_RUN_TILE / 32 + t] = 0;
	if (t == 0) fill_s = 0;
	__syncthreads();                                              // (a block without tiles reads fill_s right away, at the end)
	// the staged entries go to the list (block-uniform call; f = fill_s read between two barriers)
#define LQ_RUN_FLUSH(f) do { \
		if (t == 0) { slot0 = atomicAdd(n_runs, (f)); } \
		__syncthreads(); \
		for (u32 i_ = t; i_ < (f); i_ += LQ_RUN_THREADS) runs[slot0 + i_] = stg[i_]; \
		if (t == 0) { fill_s = 0; } \
		__syncthreads(); } while (0)
	for (u32 T = T0; T < T1; ++T) {
		const u64 base = (u64)T * LQ_RUN_TILE;
		const u32 tl = n - base < LQ_RUN_TILE ? (u32)(n - base) : LQ_RUN_TILE;
		if (t < LQ_RUN_PEEK) nx[t] = base + tl + t < n ? lq_hi32(A + base + tl + t) : 0u;
		if (t < LQ_RUN_TILE / 32) qbits[t] = 0;
		__syncthreads();
		{	// queries that start in [base, base + tile): the first candidate is the query that holds `base`
			u32 lo = qlo, hi = n_q;                               // invariant: aq_off[lo] - a_base <= base
			if (lo == ~0u) { lo = 0; while (hi - lo > 1) { const u32 mid = lo + ((hi - lo) >> 1); if (aq_off[mid] - a_base <= base) lo = mid; else hi = mid; } }
			else while (lo + 1 < n_q && aq_off[lo + 1] - a_base <= base) ++lo;
			qlo = lo;
			for (u32 q = lo + t; q < n_q; q += LQ_RUN_THREADS) {
				const u64 p = aq_off[q] - a_base;
				if (p >= base + LQ_RUN_TILE) break;
				if (p >= base && aq_off[q] < aq_off[q + 1]) atomicOr(&qbits[(u32)(p - base) >> 5], 1u << ((u32)(p - base) & 31));
			}
		}
		__syncthreads();
		// a run starts where the high word of x changes or a query starts (eight rows of loads in flight per thread)
		const u32 *hp = (const u32*)(A + base) + 1;               // high word of x of the tile's anchor o: hp[4 * o]
#pragma unroll 1
		for (u32 j0 = 0; j0 < LQ_RUN_ROWS; j0 += 8) {
			u32 h[8];
#pragma unroll
			for (u32 j = 0; j < 8; ++j) { const u32 o = (j0 + j) * LQ_RUN_THREADS + t; h[j] = o < tl ? hp[4 * o] : 0u; }
#pragma unroll
			for (u32 j = 0; j < 8; ++j) {
				const u32 o = (j0 + j) * LQ_RUN_THREADS + t;
				u32 prev = __shfl_up(h[j], 1);
				if (lane == 0) prev = (o < tl && base + o > 0) ? hp[4 * (i64)o - 4] : ~h[j];
				const u64 bal = __ballot(o < tl && (base + o == 0 || h[j] != prev || (qbits[o >> 5] >> (o & 31) & 1)));
				if (lane == 0) { hb[(o >> 5)] = (u32)bal; hb[(o >> 5) + 1] = (u32)(bal >> 32); }
			}
		}
		__syncthreads();
		// room for the tile's entries (at most one per n_min anchors); when even an empty stage cannot promise that, per row
		const u32 worst = tl / n_min + 1;
		const bool per_row = worst > stage;
		{
			const u32 f = fill_s;
			__syncthreads();
			if (!per_row && f + worst > stage) LQ_RUN_FLUSH(f);
		}
		for (u32 j = 0; j < LQ_RUN_ROWS; ++j) {
			if (per_row) {
				__syncthreads();
				const u32 f = fill_s;
				__syncthreads();
				if (f + LQ_RUN_THREADS > stage) LQ_RUN_FLUSH(f);
			}
			const u32 o = j * LQ_RUN_THREADS + t;
			u32 len = 0;
			if (hb[o >> 5] >> (o & 31) & 1) {
				const u32 w0 = (o + 1) >> 5, sh = (o + 1) & 31;
				const u64 win = (((u64)hb[w0 + 1] << 32 | hb[w0]) >> sh) | (sh ? (u64)hb[w0 + 2] << (64 - sh) : 0);   // heads at o + 1, o + 2, ...
				if ((win & near_mask) == 0) {                    // no head among the next n_min - 1 anchors of the tile: find the end
					u32 nxt = LQ_RUN_TILE;                       // the next head of the tile
					if (win) nxt = o + 1 + (u32)__builtin_ctzll(win);
					else {
						for (u32 wi = w0 + 2; wi < LQ_RUN_TILE / 32; ++wi) { const u32 m = hb[wi]; if (m) { nxt = wi * 32 + (u32)__builtin_ctz(m); break; } }
					}
					if (nxt < tl) len = nxt - o;
					else if (base + tl >= n) len = tl - o;
					else {                                       // goes on past the tile: to the end of its query or to the first other high word
						u32 q = qlo;
						while (aq_off[q + 1] - a_base <= base + o) ++q;
						const u64 qend = aq_off[q + 1] - a_base;
						const u32 h = lq_hi32(A + base + o);
						u64 lo = base + tl, hi = qend;           // first index in [lo, hi) whose high word differs, or hi
						const u32 pk = hi - lo < LQ_RUN_PEEK ? (u32)(hi - lo) : LQ_RUN_PEEK;
						u32 i = 0;
						while (i < pk && nx[i] == h) ++i;
						lo += i;
						if (i == LQ_RUN_PEEK)
							while (lo < hi) { const u64 mid = lo + ((hi - lo) >> 1); if (lq_hi32(A + mid) == h) lo = mid + 1; else hi = mid; }
						len = (u32)(lo - (base + o));
					}
				}
			}
			const bool viable = len >= n_min && len != 0;
			const u64 vb = __ballot(viable);
			if (vb) {
				u32 b = 0;
				if (lane == 0) b = atomicAdd(&fill_s, (u32)__popcll(vb));
				b = __shfl(b, 0);
				if (viable) stg[b + (u32)__popcll(vb & ((1ULL << lane) - 1))] = (base + o) | (u64)len << 32;
			}
		}
		__syncthreads();
	}
	{
		const u32 f = fill_s;
		__syncthreads();
		if (f) LQ_RUN_FLUSH(f);
	}
#undef LQ_RUN_FLUSH
}

__device__ __forceinline__ int lq_ilog2_32(u32 v) { return 31 - __clz(v); }   // chain.c:15-20 for v > 0

template <class UP>
__device__ __forceinline__ void lq_heapsort_u64(UP a, i64 n)
{
	if (n < 2) return;
	for (i64 start = n / 2 - 1; start >= 0; --start) {
		i64 root = start; u64 v = a[root];
		for (;;) {
			i64 ch = 2 * root + 1;
			if (ch >= n) break;
			if (ch + 1 < n && a[ch + 1] > a[ch]) ++ch;
			if (a[ch] <= v) break;
			a[root] = a[ch]; root = ch;
		}
		a[root] = v;
	}
	for (i64 end = n - 1; end > 0; --end) {
		u64 v = a[end]; a[end] = a[0];
		i64 root = 0;
		for (;;) {
			i64 ch = 2 * root + 1;
			if (ch >= end) break;
			if (ch + 1 < end && a[ch + 1] > a[ch]) ++ch;
			if (a[ch] <= v) break;
			a[root] = a[ch]; root = ch;
		}
		a[root] = v;
	}
}

__device__ __forceinline__ void lq_heapsort_u32(u32 *a, i64 n)
{
	if (n < 2) return;
	for (i64 start = n / 2 - 1; start >= 0; --start) {
		i64 root = start; u32 v = a[root];
		for (;;) {
			i64 ch = 2 * root + 1;
			if (ch >= n) break;
			if (ch + 1 < n && a[ch + 1] > a[ch]) ++ch;
			if (a[ch] <= v) break;
			a[root] = a[ch]; root = ch;
		}
		a[root] = v;
	}
	for (i64 end = n - 1; end > 0; --end) {
		u32 v = a[end]; a[end] = a[0];
		i64 root = 0;
		for (;;) {
			i64 ch = 2 * root + 1;
			if (ch >= end) break;
			if (ch + 1 < end && a[ch + 1] > a[ch]) ++ch;
			if (a[ch] <= v) break;
			a[root] = a[ch]; root = ch;
		}
		a[root] = v;
	}
}

// esterr.c:17-24
__device__ __forceinline__ i32 lq_fwd_qpos(i32 qlen, const mm128 &a)
{
	i32 x = (i32)a.y, q_span = (i32)(a.y >> 32 & 0xff);
	if (a.x >> 63) x = qlen - 1 - (x + 1 - q_span);
	return x;
}

struct ChainBufs {
	i32 *f, *p, *t, *v;        // per anchor scratch (chain.c:31-35)
	u64 *u;                    // per anchor scratch: chain ends / chains
};

struct CovState {              // per-query accumulators (minimap2-coverage.c:435-444)
	unsigned long long *lambda, *lambda2;
	u32 *cnts;                 // match counters, one per unfiltered query minimizer (uint16 in the reference)
	u32 *qflags;               // LQCOV_ROW_* bits
	const u32 *skip;           // esterr.c:85-91 verdict of this (query, part)
	const u64 *qmoff;          // counter array offsets
	const u64 *mini_pos;       // this part's filtered minimizer list (lqmap.c:174)
	const u64 *mpq_off;
	const u32 *qlen;
	const u32 *tlen;           // target lengths of this part
	Ivl *ivl; u32 *n_ivl; u32 ivl_cap;             // this part's intervals (esterr.c:122-126)
	ChainRec *dbg; unsigned long long *n_dbg; u64 dbg_cap;   // optional chain dump
	// Equal-x anchors (one target minimizer hit by two minimizers of the query) reach mm_chain_dp in the order klib's unstable
	// sort leaves them (lqmap.c:238), and the DP can tell (chain.c:69-76,102-125).  tie_mode 0: the array order IS klib's, chain
	// every run.  1: the anchors were sorted by some other correct sort; a run in which the order of equal-x anchors can be
	// observed (lq_chain_fill / lq_chain_finish say when) is not chained but listed in `sens` as q << 32 | high word of x.
	// 2: klib's order again, only the runs listed in `want` (sorted) are chained -- the second pass over the listed queries.
	int tie_mode;
	unsigned long long *sens; u32 *n_sens; u32 sens_cap;
	const unsigned long long *want; u32 n_want;
	const u32 *qmap;           // not null: the batch's i-th query is query qmap[i] (a subset of the queries); else q0 + i
	// The reference's counters are uint16 and the test that should saturate them reads a[st], not a[j] (esterr.c:130,136): once a
	// counter of a query reaches cnt_max (65535) its final counters depend on the order in which the chains were processed.
	// The kernels count in 32 bits and flag the query; a flagged query is chained once more with `rec` set: nothing is
	// accumulated, every kept chain is written out, and the host replays them in mm_gen_regs' order (sat_replay.hpp).
	u32 cnt_max;
	SatRec *rec; unsigned long long *n_rec; u64 rec_cap; u32 *rec_at; unsigned long long *n_at; u64 at_cap;
};

__device__ __forceinline__ bool lq_tie_wanted(const CovState &C, u32 q, u32 hi)
{
	const unsigned long long key = (unsigned long long)q << 32 | hi;
	u32 lo = 0, n = C.n_want;
	while (lo < n) { const u32 mid = lo + ((n - lo) >> 1); if (C.want[mid] < key) lo = mid + 1; else n = mid; }
	return lo < C.n_want && C.want[lo] == key;
}
// why: 1 a skip was pending when the group began, 2 a member of the group counts as a skip, 3 the group's top score is reached twice,
// 4 the scan broke off before a tie partner that would have raised the best score, 5 two equal-x peaks of one score in the backtrack
// order (chain.c:102-108); n_sens[why] counts the listed runs by their first reason (bench.py: klib_order.reasons)
#define LQ_TIE_WHY_BREAK 4u
#define LQ_TIE_WHY_PEAK 5u
__device__ __forceinline__ void lq_tie_list(const CovState &C, u32 q, u32 hi, u32 why)
{
	atomicAdd(C.n_sens + (why >= 1u && why <= 5u ? why : 6u), 1u);
	const u32 s = atomicAdd(C.n_sens, 1u);
	if (s < C.sens_cap) C.sens[s] = (unsigned long long)q << 32 | hi;
}

// runs of at least min_cnt anchors as a dense work list; key = ~size so that an ascending radix sort yields longest-first
// (long runs start first): two light passes over the run starts (count per tile, scan of the tile counts, write)
__global__ void __launch_bounds__(LQ_RUN_THREADS)
k_sel_count(const u64 *gstart, u64 n_groups, i32 min_cnt, i32 max_cnt, u32 n_tiles, u32 *tile_cnt)
{
	__shared__ u32 tot;
	for (u32 T = blockIdx.x; T < n_tiles; T += gridDim.x) {
		if (threadIdx.x == 0) tot = 0;
		__syncthreads();
		u32 c = 0;
		for (int j = 0; j < LQ_RUN_ROWS; ++j) {
			const u64 g = (u64)T * LQ_RUN_TILE + (u32)j * LQ_RUN_THREADS + threadIdx.x;
			i64 l = 0;
			if (g < n_groups) l = LQ_RUN_LEN(gstart[g]);
			c += (u32)__popcll(__ballot(g < n_groups && l >= (i64)min_cnt && l <= (i64)max_cnt));
		}
		if ((threadIdx.x & 63) == 0) atomicAdd(&tot, c);
		__syncthreads();
		if (threadIdx.x == 0) tile_cnt[T] = tot;
		__syncthreads();
	}
}

__global__ void __launch_bounds__(LQ_RUN_THREADS)
k_sel_write(const u64 *gstart, u64 n_groups, i32 min_cnt, i32 max_cnt, u32 n_tiles, const u32 *tile_off, u32 *sel, u32 *key)
{
	__shared__ u32 pre[LQ_RUN_ROWS * LQ_RUN_WAVES];
	const u32 t = threadIdx.x, lane = t & 63, w = t >> 6;
	for (u32 T = blockIdx.x; T < n_tiles; T += gridDim.x) {
		u64 bal[LQ_RUN_ROWS];
		u32 len[LQ_RUN_ROWS];
#pragma unroll
		for (int j = 0; j < LQ_RUN_ROWS; ++j) {
			const u64 g = (u64)T * LQ_RUN_TILE + (u32)j * LQ_RUN_THREADS + t;
			const u64 l = g < n_groups ? (u64)LQ_RUN_LEN(gstart[g]) : 0;
			len[j] = (u32)l;
			bal[j] = __ballot(g < n_groups && (i64)l >= (i64)min_cnt && (i64)l <= (i64)max_cnt);
		}
		if (lane == 0) {
#pragma unroll
			for (int j = 0; j < LQ_RUN_ROWS; ++j) pre[j * LQ_RUN_WAVES + w] = (u32)__popcll(bal[j]);
		}
		__syncthreads();
		if (t < 64) {
			const u32 v = pre[t];
			u32 inc = v;
			for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
			pre[t] = inc - v;
		}
		__syncthreads();
		const u32 off = tile_off[T];
#pragma unroll
		for (int j = 0; j < LQ_RUN_ROWS; ++j)
			if (bal[j] >> lane & 1) {
				const u32 r = off + pre[j * LQ_RUN_WAVES + w] + (u32)__popcll(bal[j] & ((1ULL << lane) - 1));
				sel[r] = (u32)((u64)T * LQ_RUN_TILE + (u32)j * LQ_RUN_THREADS + t);
				if (key) key[r] = 0xffffffffu - len[j];
			}
		__syncthreads();
	}
}

// ---- when does the order of equal-x anchors matter to the scores? ---------------------------------------------------------------
// Anchors of equal x never chain to each other (dr == 0, chain.c:52) and are neighbours in the sorted array.  A candidate outside
// the band is stepped over before any state changes (the `continue`s of chain.c:52-56 precede chain.c:69-76), so only the
// members of a tie group that are inside the band of the same scan (a "group" below) can be told apart by their order.  What
// a scanned candidate does: raise the best score (sc > max_f: max_f, max_j, one skip forgiven, chain.c:69-71), or count as a
// skip (t[j] == i, chain.c:72-74), or neither ("quiet") -- and leave its mark (chain.c:76), which every scanned candidate does
// whatever the order.  A quiet member (sc <= the best score before the group -- it only grows --, t[j] != i) commutes with
// everything.  A group none of whose members raises the best score commutes as a whole: its marked members count as skips one
// by one in any order, the scan ends -- if it does -- at the same count, and nothing that could have raised the score is left
// unscanned (on a scaled-down configs[2] that is 92 % of the groups with two or more loud members; without this case 29 % more
// anchors went through klib's passes).  With a member that raises it, two or more loud members
// still commute when no skip is pending before the group and none of them counts as one
// (then n_skip stays 0 in any order) and the highest score among them is reached by one member only (then max_f and max_j end
// the same).  A scan that breaks off (chain.c:73) does so at a loud member that counts as a skip: its tie partners not yet
// scanned would, in another order, have come first -- if one of them would have raised the best score, the order matters.  With every group of
// every scan order-free, f, p, v and the marks are the same per anchor in any order of the ties (oracle: sort modes 2 / 3;
// tests/test_tie_order.py chains every run the rule calls order-free in two orders and compares).
struct TieGroup {
	u32 x; i32 m, top;
	u32 st;                    // bit0: open, bit1: no skip pending at its start, bit2: a member counts as a skip, bit3: the top score reached twice, bit4: a member raises the best score, bits 5..: loud members
	__device__ __forceinline__ bool bad() const { return (st & 1u) && (st & 16u) && (st >> 5) >= 2u && (!(st & 2u) || (st & 12u)); }
	__device__ __forceinline__ u32 why() const { return !bad() ? 0u : !(st & 2u) ? 1u : (st & 4u) ? 2u : 3u; }   // (lq_tie_list)
	__device__ __forceinline__ bool raises(i32 sc) const { return sc > m; }
	// candidate (low word of x, score, counts as a skip) meets the scan's state (max_f, n_skip) as it is before it; not 0: the group that closes here was order-dependent (why())
	__device__ __forceinline__ u32 see(u32 xj, i32 sc, bool tmark, i32 max_f, i32 n_skip)
	{
		u32 r = 0;
		if (!((st & 1u) && xj == x)) { r = why(); x = xj; m = max_f; top = (i32)0x80000000; st = 1u | (n_skip == 0 ? 2u : 0u); }
		if (!(sc <= m && !tmark)) {
			st += 32u;
			if (sc > m) st |= 16u;
			if (tmark) st |= 4u;
			if (sc > top) { top = sc; st &= ~8u; } else if (sc == top) st |= 8u;
		}
		return r;
	}
};

// mm_chain_dp, first half (chain.c:41-81): scores f, predecessors p, peak scores v of one run, serially.
// Returns why (not 0) when the scores may depend on the order of equal-x anchors (watch_ties only; TieGroup above, lq_tie_list).
template <class AP, class IP>
__device__ __forceinline__ u32 lq_chain_fill(AP a, const i64 n, IP f, IP p, IP t, IP v, const float avg_qspan, const MapParams &P, const bool watch_ties)
{
	const i32 max_dist = P.max_gap, bw = P.bw, max_skip = P.max_skip;
	i64 st = 0;
	bool band_tie = false;
	TieGroup tg; tg.x = 0; tg.m = 0; tg.top = 0; tg.st = 0;
	for (i64 i = 0; i < n; ++i) t[i] = 0;
	// fill the score and backtrack arrays (chain.c:41-81).  Flat form: one candidate predecessor per loop trip,
	// so that the lanes of a wave (different runs) do not wait for each other's inner loops to finish.
	{
		i64 i = 0, j = -1, max_j = -1;
		u64 ri = 0;
		i32 qi = 0, q_span = 0, max_f = 0, n_skip = 0;
		bool setup = true;
		while (i < n) {
			if (setup) {
				ri = a[i].x; qi = (i32)a[i].y; q_span = (i32)(a[i].y >> 32 & 0xff);
				max_f = q_span; max_j = -1; n_skip = 0;
				while (st < i && ri - a[st].x > (u64)max_dist) ++st;
				j = i - 1; setup = false; tg.st = 0;
			}
			if (j >= st) {
				const i64 dr = (i64)(ri - a[j].x);
				const i32 dq = qi - (i32)a[j].y;
				if (!(dr == 0 || dq <= 0 || dq > max_dist)) {
					const i32 dd = dr > dq ? (i32)(dr - dq) : (i32)(dq - dr);
					if (dd <= bw) {
						const i32 min_d = dq < dr ? dq : (i32)dr;
						i32 sc = min_d > q_span ? q_span : min_d;
						const i32 log_dd = dd ? lq_ilog2_32((u32)dd) : 0;
						sc -= (i32)((double)dd * .01 * (double)avg_qspan) + (log_dd >> 1);     // chain.c:67
						sc += f[j];
						bool brk = false;
						if (watch_ties) { const u32 why = tg.see((u32)a[j].x, sc, t[j] == (i32)i, max_f, n_skip); if (why) return why; }   // (the run is listed, not chained: nothing it would compute from here on is used)
						if (sc > max_f) {
							max_f = sc; max_j = j;
							if (n_skip > 0) --n_skip;
						} else if (t[j] == (i32)i) {
							if (++n_skip > max_skip) brk = true;                      // chain.c:72-73
						}
						if (brk) {
							if (watch_ties)
								for (i64 jj = j - 1; jj >= st && a[jj].x == a[j].x; --jj) {   // tie partners the scan no longer reaches
									const i32 dq2 = qi - (i32)a[jj].y;
									if (dq2 <= 0 || dq2 > max_dist) continue;
									const i32 dd2 = dr > dq2 ? (i32)(dr - dq2) : (i32)(dq2 - dr);
									if (dd2 > bw) continue;
									const i32 md2 = dq2 < dr ? dq2 : (i32)dr;
									const i32 sc2 = (md2 > q_span ? q_span : md2) - ((i32)((double)dd2 * .01 * (double)avg_qspan) + ((dd2 ? lq_ilog2_32((u32)dd2) : 0) >> 1)) + f[jj];
									if (tg.raises(sc2)) return LQ_TIE_WHY_BREAK;
								}
							j = st;                                                   // leave the candidate loop
						}
						else if (p[j] >= 0) t[p[j]] = (i32)i;
					}
				}
				--j;
			} else {
				if (watch_ties && tg.bad()) return tg.why();                      // the scan's last group
				f[i] = max_f; p[i] = (i32)max_j;
				v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
				++i; setup = true;
			}
		}
	}
	return band_tie ? 1u : 0u;
}

// mm_chain_dp, second half (chain.c:84-137) + mm_reg_set_coor (hit.c:23-38) + lq_cnt_match (esterr.c:99-138)
// Returns true -- before anything is accumulated -- when two anchors of equal x are both peaks of the same score
// (watch_ties only): the backtracks run in (score, array index) order (chain.c:102-108), the one place after the scores
// where the order inside a tie group still counts.
template <class AP, class IP, class UP>
__device__ __forceinline__ bool lq_chain_finish(AP a, const i64 n, IP f, IP p, IP t, IP v, UP u,
                                                const u32 q, const bool accumulate, const MapParams &P, const CovState &C, const bool watch_ties)
{
	const i32 min_sc = P.min_sc;
	// chain ends (chain.c:84-101)
	for (i64 i = 0; i < n; ++i) t[i] = 0;
	for (i64 i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
	i64 n_u = 0;
	for (i64 i = 0; i < n; ++i) {
		if (t[i] == 0 && v[i] >= min_sc) {
			i64 j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];
			if (j < 0) j = i;
			u[n_u++] = (u64)(u32)f[j] << 32 | (u64)j;
		}
	}
	if (n_u == 0) return false;
	lq_heapsort_u64(u, n_u);                                   // keys are distinct: any sort == radix_sort_64 (chain.c:102)
	if (watch_ties)
		for (i64 i = 1; i < n_u; ++i) {
			const u64 u0 = u[i - 1], u1 = u[i];
			if ((u0 >> 32) == (u1 >> 32) && (u32)u0 != (u32)u1 && a[(i64)(i32)u0].x == a[(i64)(i32)u1].x) return true;
		}
	// backtrack from the best end (chain.c:108-125); u[] is ascending, so walk it from the top
	for (i64 i = 0; i < n; ++i) t[i] = 0;
	i64 n_v = 0;
	u32 n_rec_run = 0;
	const i32 qlen = (i32)C.qlen[q];
	const u64 *mp = C.mini_pos + C.mpq_off[q];
	const i32 n_mp = (i32)(C.mpq_off[q + 1] - C.mpq_off[q]);
	for (i64 ui = n_u - 1; ui >= 0; --ui) {
		const i64 n_v0 = n_v;
		const u64 ue = u[ui];
		i64 j = (i64)(i32)ue;
		do { v[n_v++] = (i32)j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
		const i32 cnt = (i32)(n_v - n_v0);
		i32 score;
		bool keep = false;
		if (j < 0) { score = (i32)(ue >> 32); keep = cnt >= P.min_cnt; }
		else { score = (i32)(ue >> 32) - f[j]; keep = score >= min_sc && cnt >= P.min_cnt; }
		if (!keep) { n_v = n_v0; continue; }
		// the chain's anchors in ascending x: a[v[n_v-1]], ..., a[v[n_v0]]   (chain.c:131-137)
		const mm128 first = a[v[n_v - 1]], last = a[v[n_v0]];
		// mm_reg_set_coor (hit.c:23-38)
		const i32 q_span = (i32)(first.y >> 32 & 0xff);
		const u32 rev = (u32)(first.x >> 63);
		const i32 rid = (i32)(first.x << 1 >> 33);
		const i32 rs = (i32)first.x + 1 > q_span ? (i32)first.x + 1 - q_span : 0;
		const i32 re = (i32)last.x + 1;
		i32 qs, qe;
		if (!rev) { qs = (i32)first.y + 1 - q_span; qe = (i32)last.y + 1; }
		else { qs = qlen - ((i32)last.y + 1); qe = qlen - ((i32)first.y + 1 - q_span); }
		if (C.dbg) {
			unsigned long long d = atomicAdd(C.n_dbg, 1ULL);
			if (d < C.dbg_cap) { ChainRec r; r.q = (i32)q; r.rid = rid; r.rev = (i32)rev; r.score = score; r.cnt = cnt; r.qs = qs; r.qe = qe; r.rs = rs; r.re = re; C.dbg[d] = r; }
		}
		if (!accumulate) continue;
		// lq_cnt_match for this reg (esterr.c:99-138)
		const i32 x0 = lq_fwd_qpos(qlen, rev ? last : first);
		i32 L = 0, R = n_mp - 1, sti = -1;
		while (L <= R) {                                           // get_mini_idx (esterr.c:26-38)
			const i32 m = (i32)(((u64)L + (u64)R) >> 1), y = (i32)mp[m];
			if (y < x0) L = m + 1; else if (y > x0) R = m - 1; else { sti = m; break; }
		}
		const u32 rl = C.tlen[rid];
		const u32 uqs = (u32)qs, uqe = (u32)qe, urs = (u32)rs, ure = (u32)re;
		const u32 hang5 = uqs < urs ? uqs : urs;
		const u32 hang3 = (u32)qlen - uqe < rl - ure ? (u32)qlen - uqe : rl - ure;
		const bool pass = sti >= 0 && !((double)(uqe - uqs) < (double)(uqe - uqs + hang5 + hang3) * P.min_ratio || hang5 > (u32)P.max_overhang || hang3 > (u32)P.max_overhang);
		if (C.rec) {                                               // replay of a saturated query: the chain goes to the host as it is
			const bool good = pass && score >= (i32)(u16)P.min_sc_good;
			SatRec sr;
			sr.first_x = first.x; sr.first_y = first.y & ~LQ_TIE_MARK; sr.f_peak = (u32)(ue >> 32); sr.run_hi = (u32)(first.x >> 32); sr.peak_j = (u32)ue; sr.seq = n_rec_run++;
			sr.score = (u32)score; sr.cnt = (u32)cnt; sr.sti = sti; sr.n_at = 0; sr.at_off = 0; sr.good = good ? 1u : 0u; sr.span = uqe - uqs + 1;
			if (good && cnt > 1) {
				sr.at_off = atomicAdd(C.n_at, (unsigned long long)(cnt - 1));
				const bool room = sr.at_off + (u64)(cnt - 1) <= C.at_cap;
				i32 k = 1;
				for (i32 jj = sti + 1; jj < n_mp && k < cnt; ++jj) {
					const mm128 ak = rev ? a[v[n_v0 + k]] : a[v[n_v - 1 - k]];
					if (lq_fwd_qpos(qlen, ak) == (i32)mp[jj]) { ++k; if (room) C.rec_at[sr.at_off + sr.n_at] = (u32)jj; ++sr.n_at; }
				}
			}
			const unsigned long long r = atomicAdd(C.n_rec, 1ULL);
			if (r < C.rec_cap) C.rec[r] = sr;
			continue;
		}
		if (!pass) continue;
		atomicAdd(&C.lambda[q], (unsigned long long)(u32)(uqe - uqs + 1));
		u32 flag = score >= (i32)(u16)P.min_sc_med ? 2u : 0u;
		{
			u32 s = atomicAdd(C.n_ivl, 1u);
			if (s < C.ivl_cap) { Ivl iv; iv.q = q; iv.start = uqs << 3 | flag; iv.end = uqe << 3 | flag | 1u; C.ivl[s] = iv; }
		}
		if (score < (i32)(u16)P.min_sc_good) continue;
		atomicAdd(&C.lambda2[q], (unsigned long long)(u32)(uqe - uqs + 1));
		u32 *cn = C.cnts + C.qmoff[q];
		u32 old = atomicAdd(&cn[sti], 1u);
		if (old + 1 >= C.cnt_max) atomicOr(&C.qflags[q], 1u);       // esterr.c:130: saturation regime, the order matters (replayed)
		i32 k = 1;
		for (i32 jj = sti + 1; jj < n_mp && k < cnt; ++jj) {
			const mm128 ak = rev ? a[v[n_v0 + k]] : a[v[n_v - 1 - k]];
			if (lq_fwd_qpos(qlen, ak) == (i32)mp[jj]) {
				++k;
				u32 o2 = atomicAdd(&cn[jj], 1u);
				if (o2 + 1 >= C.cnt_max) atomicOr(&C.qflags[q], 1u);
			}
		}
	}
	return false;
}

// The second pass chains a few thousand runs out of the tens of millions its queries have: their list entries come straight
// from their keys -- want[w] = query << 32 | high word of x (strand, target) -- by two bisections in the query's anchors (klib's
// order: ascending x), instead of a run list of everything (k_run_list) that a full-grid k_chain then searches the keys for
// (configs[2]: 28 + 4 ms of a lane's second pass).  qmap: the subset's queries, ascending; a key of a query that is not in
// this subset gets length 0.
__global__ void k_want_runs(const mm128 *A, const u64 *aq_off, u32 n_q, const u32 *qmap, const unsigned long long *want, u32 n_want, u64 *gstart)
{
	const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_want) return;
	const u32 q = (u32)(want[w] >> 32), hi = (u32)want[w];
	u32 lo = 0, n = n_q;
	while (lo < n) { const u32 mid = lo + ((n - lo) >> 1); if (qmap[mid] < q) lo = mid + 1; else n = mid; }
	u64 e = 0;
	if (lo < n_q && qmap[lo] == q) {
		const u64 s1 = aq_off[lo + 1];
		u64 a = aq_off[lo], b = s1;                            // the first anchor whose high word is >= hi
		while (a < b) { const u64 m = a + ((b - a) >> 1); if (lq_hi32(A + m) < hi) a = m + 1; else b = m; }
		u64 c = a, d = s1;                                     // ... > hi
		while (c < d) { const u64 m = c + ((d - c) >> 1); if (lq_hi32(A + m) <= hi) c = m + 1; else d = m; }
		e = a | (c - a) << 32;
	}
	gstart[w] = e;
}

// One (strand, rid) run of a query: mm_chain_dp on a[0..n), then per chain mm_reg_set_coor and lq_cnt_match.
// a, f, p, t, v, u may live in global memory (long runs) or in the calling thread's private arrays (short runs).
template <class AP, class IP, class UP>
__device__ __forceinline__ void lq_chain_run(AP a, const i64 n, IP f, IP p, IP t, IP v, UP u,
                                             const u32 q, const bool accumulate, const float *avg_qspan_q, const MapParams &P, const CovState &C)
{
	const bool watch = C.tie_mode == 1;
	u32 why = lq_chain_fill(a, n, f, p, t, v, avg_qspan_q[q], P, watch);
	if (!why && lq_chain_finish(a, n, f, p, t, v, u, q, accumulate, P, C, watch)) why = LQ_TIE_WHY_PEAK;
	if (why) lq_tie_list(C, q, (u32)(a[0].x >> 32), why);   // the order of its equal-x anchors can be observed: left to the second pass
}

// does the run have any chance to yield a chain?  A chain of c anchors scores at most the sum of their spans
// (every step adds min(dq, dr, span) <= span minus a non-negative gap cost, chain.c:57-67), and chains below
// min_sc are dropped (chain.c:86-101,119-121); so runs with fewer than min_cnt anchors or with a span total
// below min_sc can be skipped without looking at them.
template <class AP>
__device__ __forceinline__ bool lq_run_viable(AP a, i64 n, const MapParams &P)
{
	if (n < P.min_cnt) return false;
	if (n * 255 < P.min_sc) return false;
	i64 tot = 0;
	for (i64 i = 0; i < n && tot < P.min_sc; ++i) tot += (i64)(a[i].y >> 32 & 0xff);
	return tot >= P.min_sc;
}

// aq_off: the batch's view of the per-query anchor offsets (n_q+1 entries, absolute; the batch's
// anchors start at a_base); q0: global index of the batch's first query.
// One thread per run of min_len..max_len anchors, in array order.  The serial DP is a chain of dependent, scattered
// accesses to the run's anchors and DP arrays; with 64 unrelated working sets per wave in global memory those are L2
// round trips and partial-line writes (rocprofv3: 18 GB written per launch for 8 GB of anchors).  So the wave packs the
// runs it will actually chain (most runs are too short or cannot reach min_sc) into LDS -- anchors plus f/p/t/v/u, 40 B
// per anchor, offsets from a scan of the run lengths -- and each lane works on its slice there.  When the runs of a wave
// exceed the LDS budget they are taken in rounds: the runs that fit go now, the others wait for the next round.
// Measured on MI355X at configs[1] (k_chain per step): DP state in global scratch 113 ms; first version of the LDS
// packing, overflow runs in global scratch: budget of 384 anchors per wave 127 ms (10 waves per CU), 256: 107, 128: 97.
template <int LQ_CHAIN_LDS_CAP>
__global__ void __launch_bounds__(64)
k_chain(const mm128 *A, const u64 *gstart, const u32 *glist, u32 n_list, const u64 *aq_off, u64 a_base, u32 n_q, u32 q0,
        const float *avg_qspan_q, MapParams P, CovState C, i32 min_len, i32 max_len)
{
	LQ_SHARED mm128 s_a[LQ_CHAIN_LDS_CAP];
	LQ_SHARED i32 s_f[LQ_CHAIN_LDS_CAP], s_p[LQ_CHAIN_LDS_CAP], s_t[LQ_CHAIN_LDS_CAP], s_v[LQ_CHAIN_LDS_CAP];
	LQ_SHARED u64 s_u[LQ_CHAIN_LDS_CAP];
	LQ_SHARED u32 s_n[64], s_q[64], s_done[64];
	const u32 gi0 = blockIdx.x * blockDim.x;
	LQ_BLOCK_LOOP(ln) {
		u32 todo = 0;
		const u32 gi = gi0 + ln;
		if (gi < n_list) {
			const u32 g = glist ? glist[gi] : gi;
			const u64 gs = LQ_RUN_START(gstart[g]);
			const i64 n = LQ_RUN_LEN(gstart[g]);
			if (n >= min_len && n <= max_len && n <= LQ_CHAIN_LDS_CAP && lq_run_viable(A + gs, n, P)) {
				const u32 qi = lq_find_seg(aq_off, n_q, gs + a_base);
				const u32 q = C.qmap ? C.qmap[qi] : q0 + qi;
				if ((!C.skip[q] || C.dbg) && (C.tie_mode != 2 || lq_tie_wanted(C, q, lq_hi32(A + gs)))) { todo = (u32)n; s_q[ln] = q; }
			}
		}
		s_n[ln] = todo; s_done[ln] = 0;
	}
	LQ_BLOCK_SYNC();
	for (;;) {
		u32 left = 0;                                        // block-uniform: runs still waiting
		for (u32 z = 0; z < 64; ++z) left += s_n[z] != 0;
		if (left == 0) break;
		LQ_BLOCK_LOOP(ln) {
			const u32 n32 = s_n[ln];
			if (n32) {
				u32 off = 0;
				for (u32 z = 0; z < ln; ++z) off += s_n[z];
				if (off + n32 <= LQ_CHAIN_LDS_CAP) {             // the first waiting run always fits
					const u32 gi = gi0 + ln;
					const u32 g = glist ? glist[gi] : gi;
					const u64 gs = LQ_RUN_START(gstart[g]);
					const i64 n = (i64)n32;
					const u32 q = s_q[ln];
					mm128 *la = s_a + off;
					for (i64 i = 0; i < n; ++i) la[i] = A[gs + i];
					lq_chain_run(la, n, s_f + off, s_p + off, s_t + off, s_v + off, s_u + off, q, !C.skip[q], avg_qspan_q, P, C);
					s_done[ln] = 1;
				}
			}
		}
		LQ_BLOCK_SYNC();
		LQ_BLOCK_LOOP(ln) { if (s_done[ln]) { s_n[ln] = 0; s_done[ln] = 0; } }
		LQ_BLOCK_SYNC();
	}
}

// Long runs, one wave per run.  A chain kernel ends when its longest run ends, and the serial DP of a long run (a true
// overlap: hundreds to tens of thousands of anchors, tens of candidate predecessors each) is that tail.  The 64 lanes score
// the 64 nearest candidate predecessors j = i-1 .. i-64 of anchor i at once.
//   * Those candidates are, but for one, the candidates of the step before: every lane keeps its candidate's x, y, f, p, v in
//     registers, the window moves by one lane per anchor (__shfl_up) and lane 0 takes the anchor just finished -- no load of
//     a[j], f[j], p[j] per step (round 3: three dependent global round trips per anchor, ~4 us an anchor; 15 s of a 24-s step
//     on the ultra-long slice of configs[4]).
//   * The marks t[p[j]] = i (chain.c:76) of candidates inside the window live in a 64-entry LDS array of stamps (slot = index
//     mod 64, one anchor per slot while it is in the window); a mark for an older anchor goes to t[] in global memory, where the
//     rare chunks beyond the window look for it.  Marks are written for the whole chunk before the stamps are read: a mark can
//     only concern a later-scanned candidate (p[j] < j), and marks of candidates beyond the break are never looked at.
//   * The order-dependent part of chain.c:48-77 -- strict '>' keeps the nearest best, the skip counter counts candidates that
//     an already-scanned anchor chose as predecessor (t[j] == i) and forgives one per new best, the scan breaks after max_skip
//     of them -- is a pair of scans over the wave: "new best" is a prefix maximum, the counter a composition of maps
//     x -> max(x + a, b) (new best: (-1, 0); skip: (+1, -inf); neither: (0, -inf)), closed under composition.  Six shuffle
//     steps each instead of a loop of 64 on lane 0.  Where the order of equal-x anchors might show (tie_mode 1: two candidates
//     of equal x inside the band of this scan) and for the rare candidates beyond the window, lane 0 replays the rules one by
//     one as before (TieGroup keeps its state across chunks).
// f, p, v are still written to global memory (the second half reads them); __syncthreads() in a block of one wave is a wait
// for the wave's own memory operations.  The second half (chain ends, backtrack, regs, coverage) is the serial
// lq_chain_finish on lane 0.
// The wave's scans and the one-lane shift of the window as DPP operations (round 6).  __shfl_up is a ds_bpermute -- a trip through the
// LDS crossbar, ~100 cycles when the next step waits for it -- and an anchor's step was two dependent scans of six such steps plus five
// shifts: ~2000 cycles, a 30 000-anchor run 30 ms on one wave.  A DPP operand (row_shr within a row of 16, row_bcast:15 / :31 across
// rows, wave_shr:1 -- gfx9's cross-lane modes) is read by the ALU instruction itself.  The emulator keeps the shuffles.
#ifndef LQ_EMU
#define LQ_DPP(fill, v, ctrl, rows, banks) __builtin_amdgcn_update_dpp((int)(fill), (int)(v), (ctrl), (rows), (banks), false)
// lane l takes lane l - 1's value, lane 0 takes `fill`
__device__ __forceinline__ i32 lq_wave_shr1(i32 v, i32 fill) { return LQ_DPP(fill, v, 0x138, 0xf, 0xf); }
// inclusive prefix maximum over the wave
__device__ __forceinline__ i32 lq_wave_scan_max(i32 v)
{
	const i32 NEG = (i32)0x80000000;
	i32 o;
	o = LQ_DPP(NEG, v, 0x111, 0xf, 0xf); v = o > v ? o : v;    // row_shr:1
	o = LQ_DPP(NEG, v, 0x112, 0xf, 0xf); v = o > v ? o : v;    // row_shr:2
	o = LQ_DPP(NEG, v, 0x114, 0xf, 0xf); v = o > v ? o : v;    // row_shr:4
	o = LQ_DPP(NEG, v, 0x118, 0xf, 0xf); v = o > v ? o : v;    // row_shr:8
	o = LQ_DPP(NEG, v, 0x142, 0xa, 0xf); v = o > v ? o : v;    // row_bcast:15 -> rows 1 and 3
	o = LQ_DPP(NEG, v, 0x143, 0xc, 0xf); v = o > v ? o : v;    // row_bcast:31 -> rows 2 and 3
	return v;
}
// inclusive scan of the maps x -> max(x + a, b) under composition (the earlier map first).  A lane without a source composes with
// the identity (0, -2^29): "minus infinity" may come out as -2^29 + a few -- it is only ever compared with counts of at most 64
#define LQ_WAVE_COMPOSE_STEP(ctrl, rows) do { \
		const i32 oa_ = LQ_DPP(0, fa, ctrl, rows, 0xf), ob_ = LQ_DPP(-(1 << 29), fb, ctrl, rows, 0xf); \
		const i32 nb_ = ob_ + fa; fb = nb_ > fb ? nb_ : fb; fa = oa_ + fa; } while (0)
__device__ __forceinline__ void lq_wave_scan_compose(i32 &fa, i32 &fb)
{
	LQ_WAVE_COMPOSE_STEP(0x111, 0xf); LQ_WAVE_COMPOSE_STEP(0x112, 0xf); LQ_WAVE_COMPOSE_STEP(0x114, 0xf); LQ_WAVE_COMPOSE_STEP(0x118, 0xf);
	LQ_WAVE_COMPOSE_STEP(0x142, 0xa); LQ_WAVE_COMPOSE_STEP(0x143, 0xc);
}
#define LQ_WAVE_LANE(v, l) ((i32)__builtin_amdgcn_readlane((int)(v), (int)(l)))     // l uniform
// What one wave's lanes tell each other through LDS needs the LDS operations done, nothing else.  __syncthreads() also waits for
// every global access in flight -- in k_chain_wave's loop that was the load of the next anchor and the stores of f, p, v: a round
// trip to memory per anchor, most of the microsecond an anchor took.
#define LQ_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define LQ_WAVE_SYNC() __syncthreads()
static inline i32 lq_wave_shr1(i32 v, i32 fill) { const i32 o = __shfl_up(v, 1); return threadIdx.x == 0 ? fill : o; }
static inline i32 lq_wave_scan_max(i32 v) { for (int d = 1; d < 64; d <<= 1) { const i32 o = __shfl_up(v, d); if ((int)threadIdx.x >= d && o > v) v = o; } return v; }
static inline void lq_wave_scan_compose(i32 &fa, i32 &fb)
{
	for (int d = 1; d < 64; d <<= 1) {
		const i32 oa = __shfl_up(fa, d), ob = __shfl_up(fb, d);
		if ((int)threadIdx.x >= d) { const i32 nb = ob + fa; fb = nb > fb ? nb : fb; fa = oa + fa; }
	}
}
#define LQ_WAVE_LANE(v, l) __shfl((v), (int)(l))
#endif
#define LQ_CHAIN_WAVE_MIN 48      // measured on MI355X at configs[1]: 192 -> 206 ms, 96 -> 193, 48 -> 184, 24 -> 197 (k_chain + k_chain_wave)
struct WaveCand { i32 sc, j, flags; u32 x32; };             // flags: bit0 = passes the filters, bit1 = t[j] == i; x32: low word of the candidate's x

// mm_chain_dp's second half for one wave (all 64 lanes; arrays in global memory): what lq_chain_finish does on one lane, with
// the passes that are independent per anchor spread over the lanes -- has-a-successor marks, chain ends and their peaks
// (chain.c:84-101), the per-anchor counter increments of lq_cnt_match (esterr.c:131-137: a chain's anchors ascend in query
// position and every one of them is a kept minimizer of the query, so the reference's merge loop finds anchor k at exactly the
// index a binary search of mini_pos gives) -- and the order-dependent ones (the sort of the ends when there are more than 64,
// the backtrack with its used-marks, chain.c:102-125) on lane 0.  For a run of n anchors the serial version was ~8 n dependent
// global loads on one lane: more than the scoring itself since that moved into registers.  Same return value as lq_chain_finish.
__device__ __forceinline__ bool lq_chain_finish_wave(const mm128 *a, const i64 n, i32 *f, i32 *p, i32 *t, i32 *v, u64 *u,
                                                     const u32 q, const bool accumulate, const MapParams &P, const CovState &C, const bool watch, i32 *sh /* 8 ints of LDS */)
{
	const u32 ln = threadIdx.x;
	const i32 min_sc = P.min_sc;
	for (i64 i = ln; i < n; i += 64) t[i] = 0;
	LQ_BLOCK_SYNC();
	for (i64 i = ln; i < n; i += 64) { const i32 pi = p[i]; if (pi >= 0) t[pi] = 1; }
	LQ_BLOCK_SYNC();
	i64 n_u = 0;
	for (i64 base = 0; base < n; base += 64) {                  // chain ends, in no particular order (they are sorted below)
		const i64 i = base + ln;
		bool e = false; u64 val = 0;
		if (i < n && t[i] == 0 && v[i] >= min_sc) {
			i64 j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];
			if (j < 0) j = i;
			val = (u64)(u32)f[j] << 32 | (u64)j; e = true;
		}
		const u64 bal = __ballot(e);
		if (e) u[n_u + __popcll(bal & ((1ULL << ln) - 1))] = val;
		n_u += __popcll(bal);
	}
	if (n_u == 0) return false;
	LQ_BLOCK_SYNC();
	if (n_u <= 64) {                                             // sorted by rank: the keys are distinct unless two ends share a peak (equal values: any order)
		const u64 mine = ln < n_u ? u[ln] : 0;
		u32 r = 0;
		for (u32 k = 0; k < (u32)n_u; ++k) { const u64 o = __shfl(mine, (int)k); r += (o < mine) || (o == mine && k < ln); }
		LQ_BLOCK_SYNC();
		if (ln < n_u) u[r] = mine;
	} else if (ln == 0) lq_heapsort_u64(u, n_u);
	LQ_BLOCK_SYNC();
	if (watch) {
		bool tie = false;
		for (i64 i = 1 + ln; i < n_u; i += 64) { const u64 u0 = u[i - 1], u1 = u[i]; if ((u0 >> 32) == (u1 >> 32) && (u32)u0 != (u32)u1 && a[(i64)(i32)u0].x == a[(i64)(i32)u1].x) tie = true; }
		if (__ballot(tie)) return true;
	}
	for (i64 i = ln; i < n; i += 64) t[i] = 0;
	LQ_BLOCK_SYNC();
	// backtrack from the best end (chain.c:108-125) on lane 0: the kept chains' anchors in v[], chain k as (first slot, count, score) in u[k], f[k] (f is not needed once a chain's score is known ... it is: f[j] of a chain cut at a used anchor -- so the slots go to the tail of u)
	i64 n_keep = 0;
	if (ln == 0) {
		i64 n_v = 0;
		for (i64 ui = n_u - 1; ui >= 0; --ui) {
			const i64 n_v0 = n_v;
			const u64 ue = u[ui];
			i64 j = (i64)(i32)ue;
			do { v[n_v++] = (i32)j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
			const i32 cnt = (i32)(n_v - n_v0);
			i32 score; bool keep = false;
			if (j < 0) { score = (i32)(ue >> 32); keep = cnt >= P.min_cnt; }
			else { score = (i32)(ue >> 32) - f[j]; keep = score >= min_sc && cnt >= P.min_cnt; }
			if (!keep) { n_v = n_v0; continue; }
			// (entries ui .. n_u-1 of u are done with, and n_keep <= n_u - 1 - ui: the kept chains are listed from the top of u down)
			u[n_u - 1 - n_keep] = (u64)(u32)score << 32 | (u64)(u32)cnt;
			++n_keep;
		}
		sh[0] = (i32)n_keep;
	}
	LQ_BLOCK_SYNC();
	n_keep = sh[0];
	const i32 qlen = (i32)C.qlen[q];
	const u64 *mp = C.mini_pos + C.mpq_off[q];
	const i32 n_mp = (i32)(C.mpq_off[q + 1] - C.mpq_off[q]);
	i64 n_v0 = 0;
	for (i64 k = 0; k < n_keep; ++k) {                          // every kept chain: regs and lq_cnt_match, the wave together
		const u64 e = u[n_u - 1 - k];
		const i32 score = (i32)(e >> 32), cnt = (i32)(u32)e;
		const i64 n_v = n_v0 + cnt;
		// the chain's anchors in ascending x: a[v[n_v-1]], ..., a[v[n_v0]]   (chain.c:131-137)
		const mm128 first = a[v[n_v - 1]], last = a[v[n_v0]];
		const i32 q_span = (i32)(first.y >> 32 & 0xff);
		const u32 rev = (u32)(first.x >> 63);
		const i32 rid = (i32)(first.x << 1 >> 33);
		const i32 rs = (i32)first.x + 1 > q_span ? (i32)first.x + 1 - q_span : 0;
		const i32 re = (i32)last.x + 1;
		i32 qs, qe;
		if (!rev) { qs = (i32)first.y + 1 - q_span; qe = (i32)last.y + 1; }
		else { qs = qlen - ((i32)last.y + 1); qe = qlen - ((i32)first.y + 1 - q_span); }
		if (C.dbg && ln == 0) {
			unsigned long long d = atomicAdd(C.n_dbg, 1ULL);
			if (d < C.dbg_cap) { ChainRec r; r.q = (i32)q; r.rid = rid; r.rev = (i32)rev; r.score = score; r.cnt = cnt; r.qs = qs; r.qe = qe; r.rs = rs; r.re = re; C.dbg[d] = r; }
		}
		const i64 c0 = n_v0;
		n_v0 = n_v;
		if (!accumulate) continue;
		// lq_cnt_match for this reg (esterr.c:99-138); everything up to the counters is uniform over the wave
		const i32 x0 = lq_fwd_qpos(qlen, rev ? last : first);
		i32 L = 0, R = n_mp - 1, sti = -1;
		while (L <= R) {                                           // get_mini_idx (esterr.c:26-38)
			const i32 m = (i32)(((u64)L + (u64)R) >> 1), y = (i32)mp[m];
			if (y < x0) L = m + 1; else if (y > x0) R = m - 1; else { sti = m; break; }
		}
		if (sti < 0) continue;
		const u32 rl = C.tlen[rid];
		const u32 uqs = (u32)qs, uqe = (u32)qe, urs = (u32)rs, ure = (u32)re;
		const u32 hang5 = uqs < urs ? uqs : urs;
		const u32 hang3 = (u32)qlen - uqe < rl - ure ? (u32)qlen - uqe : rl - ure;
		if ((double)(uqe - uqs) < (double)(uqe - uqs + hang5 + hang3) * P.min_ratio || hang5 > (u32)P.max_overhang || hang3 > (u32)P.max_overhang)
			continue;
		const u32 flag = score >= (i32)(u16)P.min_sc_med ? 2u : 0u;
		if (ln == 0) {
			atomicAdd(&C.lambda[q], (unsigned long long)(u32)(uqe - uqs + 1));
			u32 s = atomicAdd(C.n_ivl, 1u);
			if (s < C.ivl_cap) { Ivl iv; iv.q = q; iv.start = uqs << 3 | flag; iv.end = uqe << 3 | flag | 1u; C.ivl[s] = iv; }
		}
		if (score < (i32)(u16)P.min_sc_good) continue;
		u32 *cn = C.cnts + C.qmoff[q];
		if (ln == 0) {
			atomicAdd(&C.lambda2[q], (unsigned long long)(u32)(uqe - uqs + 1));
			const u32 old = atomicAdd(&cn[sti], 1u);
			if (old + 1 >= C.cnt_max) atomicOr(&C.qflags[q], 1u);     // esterr.c:130: saturation regime, the order matters (replayed)
		}
		// anchors 1 .. cnt-1 in query order: each is found by the reference's merge loop at its own place in mini_pos, beyond sti
		for (i32 kk = 1 + (i32)ln; kk < cnt; kk += 64) {
			const mm128 ak = rev ? a[v[c0 + kk]] : a[v[n_v - 1 - kk]];
			const i32 xk = lq_fwd_qpos(qlen, ak);
			i32 lo = sti + 1, hi = n_mp - 1, at = -1;
			while (lo <= hi) { const i32 m = (i32)(((u64)lo + (u64)hi) >> 1), y = (i32)mp[m]; if (y < xk) lo = m + 1; else if (y > xk) hi = m - 1; else { at = m; break; } }
			if (at >= 0) {
				const u32 o2 = atomicAdd(&cn[at], 1u);
				if (o2 + 1 >= C.cnt_max) atomicOr(&C.qflags[q], 1u);
			}
		}
	}
	return false;
}

// lane 0: the scan-order rules over the `cnt` staged candidates of one chunk; state in st_sh ([0] max_f, [1] max_j, [2] n_skip,
// [3] done, [4] the order of equal-x anchors may matter, [5..8] the scan's open TieGroup)
__device__ __forceinline__ void lq_wave_replay(const WaveCand *cand, i32 *st_sh, i64 cnt, bool more_beyond, bool watch, i32 max_skip)
{
	i32 max_f = st_sh[0], max_j = st_sh[1], n_skip = st_sh[2], done = 0;
	TieGroup tg; tg.x = (u32)st_sh[5]; tg.m = st_sh[6]; tg.top = st_sh[7]; tg.st = (u32)st_sh[8];
	for (i64 c = 0; c < cnt; ++c) {
		const WaveCand w = cand[c];
		if (!(w.flags & 1)) continue;
		if (watch) { const u32 why = tg.see(w.x32, w.sc, (w.flags & 2) != 0, max_f, n_skip); if (why && !st_sh[4]) st_sh[4] = (i32)why; }   // (TieGroup, above lq_chain_fill)
		if (w.sc > max_f) { max_f = w.sc; max_j = w.j; if (n_skip > 0) --n_skip; }
		else if (w.flags & 2) {
			if (++n_skip > max_skip) {                                                  // chain.c:72-73
				if (watch) {	// tie partners the scan no longer reaches (if they go on into the next 64: assume the worst)
					i64 c2 = c + 1;
					for (; c2 < cnt && cand[c2].x32 == w.x32; ++c2) if ((cand[c2].flags & 1) && tg.raises(cand[c2].sc) && !st_sh[4]) st_sh[4] = (i32)LQ_TIE_WHY_BREAK;
					if (c2 == cnt && more_beyond && !st_sh[4]) st_sh[4] = (i32)LQ_TIE_WHY_BREAK;
				}
				done = 1; break;
			}
		}
	}
	if (watch && (done || !more_beyond) && tg.bad() && !st_sh[4]) st_sh[4] = (i32)tg.why();   // the scan's last group
	st_sh[0] = max_f; st_sh[1] = max_j; st_sh[2] = n_skip; st_sh[3] = done;
	st_sh[5] = (i32)tg.x; st_sh[6] = tg.m; st_sh[7] = tg.top; st_sh[8] = (i32)tg.st;
}

__global__ void __launch_bounds__(64)
k_chain_wave(const mm128 *A, const u64 *gstart, const u32 *glist, u32 n_list, const u64 *aq_off, u64 a_base, u32 n_q, u32 q0,
             const float *avg_qspan_q, MapParams P, ChainBufs B, CovState C)
{
	LQ_SHARED WaveCand cand[64];
	LQ_SHARED i32 st_sh[10];
	LQ_SHARED i32 stamp[64];                                 // stamp[j & 63] == i  <=>  t[j] == i, for the anchors j of the window [i - 64, i - 1]
	if (blockIdx.x >= n_list) return;
	const u32 g = glist[blockIdx.x];
	const u64 gs = LQ_RUN_START(gstart[g]);
	const i64 n = LQ_RUN_LEN(gstart[g]);
	const mm128 *a = A + gs;
	if (!lq_run_viable(a, n, P)) return;
	const u32 qi_ = lq_find_seg(aq_off, n_q, gs + a_base);
	const u32 q = C.qmap ? C.qmap[qi_] : q0 + qi_;
	const bool accumulate = !C.skip[q];
	if (!accumulate && !C.dbg) return;
	if (C.tie_mode == 2 && !lq_tie_wanted(C, q, lq_hi32(a))) return;
	const bool watch = C.tie_mode == 1;
	const u32 ln = threadIdx.x;
	if (ln == 0) st_sh[4] = 0;
	stamp[ln] = -1;
	i32 *f = B.f + gs, *p = B.p + gs, *t = B.t + gs, *v = B.v + gs;
	u64 *u = B.u + gs;
	const float avg_qspan = avg_qspan_q[q];
	const i32 max_dist = P.max_gap, bw = P.bw, max_skip = P.max_skip;
	for (i64 i = ln; i < n; i += 64) t[i] = 0;
	LQ_BLOCK_SYNC();
	// Round 6: nothing in the loop waits for global memory.  (Rounds 3-5: the next anchor came as "a[i + 1] or else ai" -- choosing
	// between a load and a local made the compiler keep ai in scratch memory and fetch through a flat pointer it waited for at once --
	// and f, p, v went out one anchor at a time from lane 0, stores the next iteration's wait counted too: a memory round trip or two
	// per anchor.)  Anchors: every lane holds one of the 64 at hand in registers (and one of the next 64, asked for 64 steps before
	// their turn), the anchor of a step is a v_readlane of its lane.  f, p, v: the window holds them of the last 64 anchors -- every 64
	// steps the lanes write theirs, 64 consecutive words each.  An anchor older than the window has been written by then: the chunks
	// beyond the window and the second half read it from memory after a full wait.
	const u32 n32 = (u32)n, x_hi = lq_hi32(a);
	u32 st = 0;
	u32 wx = 0; i32 wy = 0, wf = 0, wp = -1, wv = 0;         // lane c: x (low word), y, f, p, v of anchor i - 1 - c
	mm128 cur = a[ln < n32 ? ln : n32 - 1], nxt = a[64 + ln < n32 ? 64 + ln : n32 - 1];
	for (u32 i = 0; i < n32; ++i) {
		const u32 bi = i & 63;
		if (bi == 0 && i) { cur = nxt; const u32 k2 = i + 64 + ln; nxt = a[k2 < n32 ? k2 : n32 - 1]; }   // (uniform)
		const u32 rx = (u32)LQ_WAVE_LANE((u32)cur.x, bi);       // the low word of this anchor's x (the high words are equal inside a run)
		const i32 qi = LQ_WAVE_LANE((u32)cur.y, bi), q_span = LQ_WAVE_LANE((u32)(cur.y >> 32), bi) & 0xff;
		const u64 ri = (u64)x_hi << 32 | rx;
		const i32 j = (i32)i - 1 - (i32)ln;
		{	// st: the first anchor within max_dist of this one (chain.c:47).  x ascends, so the anchors out of reach are a prefix of the
			// run: among the 64 anchors of the window the nearest one out of reach says where it ends -- no load; only when the whole
			// window is within reach (more than 64 anchors inside max_dist) are older anchors looked at in memory
			const u64 far = __ballot(j >= 0 && rx - wx > (u32)max_dist);
			if (far) { const u32 s2 = i - (u32)__builtin_ctzll(far); if (s2 > st) st = s2; }
			else while (st + 64 < i && ri - a[st].x > (u64)max_dist) ++st;   // uniform: every lane computes the same st
		}
		i32 max_f = q_span, max_j = -1, n_skip = 0, done = 0;
		// ---- the 64 nearest candidates, from the window ----
		bool active = false; i32 sc = 0;
		if (j >= (i32)st) {
			const i64 dr = (i64)(rx - wx);
			const i32 dq = qi - wy;
			if (!(dr == 0 || dq <= 0 || dq > max_dist)) {
				const i32 dd = dr > dq ? (i32)(dr - dq) : (i32)(dq - dr);
				if (dd <= bw) {
					const i32 min_d = dq < dr ? dq : (i32)dr;
					sc = min_d > q_span ? q_span : min_d;
					const i32 log_dd = dd ? lq_ilog2_32((u32)dd) : 0;
					sc -= (i32)((double)dd * .01 * (double)avg_qspan) + (log_dd >> 1);     // chain.c:67
					sc += wf;
					active = true;
					if (wp >= 0) { if (wp + 64 >= (i32)i) stamp[wp & 63] = (i32)i; else t[wp] = (i32)i; }   // chain.c:76
				}
			}
		}
		LQ_WAVE_SYNC();
		const bool tm = active && stamp[(u32)j & 63u] == (i32)i;
		const bool more = i >= 65 + st;                          // candidates beyond the window are within reach
		// Where the order of equal-x anchors may show in this scan (first pass only): a candidate that is one of two or more of
		// equal x inside the band, or the window's last one when its tie partners may lie beyond.  The scan by the wave's two
		// prefix scans below is exact up to the first such candidate -- and most scans end (chain.c:72-73) before they get there.
		u64 ties = 0;
		if (watch) {
			const u32 px = (u32)lq_wave_shr1((i32)wx, 0);
			const bool head = ln == 0 || wx != px;
			const u64 H = __ballot(head);
			if (~H) {                                              // (uniform) some x is repeated in the window
				const u64 act = __ballot(active);
				const u32 lo = 63u - (u32)__clzll(H & (~0ULL >> (63 - ln)));                           // my run of equal x starts at lane lo ...
				const u64 above = ln == 63 ? 0 : (H >> (ln + 1)) << (ln + 1);
				const u32 hi = above ? (u32)__builtin_ctzll(above) - 1 : 63u;                         // ... and ends at lane hi
				const u64 gm = (~0ULL >> (63 - hi)) & (~0ULL << lo);
				ties = __ballot(active && (__popcll(act & gm) >= 2 || (hi == 63 && more)));
			} else ties = __ballot(active && ln == 63 && more);
		}
		bool serial = false;
		{
			// new bests: candidates whose score beats everything scanned before them
			const i32 inc = lq_wave_scan_max(active ? sc : (i32)0x80000000);
			i32 exc = lq_wave_shr1(inc, (i32)0x80000000);
			if (exc < max_f) exc = max_f;
			const bool rec = active && sc > exc;
			const bool skp = active && !rec && tm;
			// the skip counter after every candidate: x -> max(x + fa, fb), composed in scan order
			i32 fa = rec ? -1 : skp ? 1 : 0, fb = rec ? 0 : -(1 << 29);
			lq_wave_scan_compose(fa, fb);
			const i32 x_after = fa > fb ? fa : fb;               // (the counter starts at 0 with every anchor)
			const u64 brk = __ballot(skp && x_after > max_skip);
			const u32 cb = brk ? (u32)__builtin_ctzll(brk) : 64u;   // the scan ends at candidate cb (chain.c:72-73)
			serial = ties && (u32)__builtin_ctzll(ties) <= cb;      // (uniform) a candidate whose tie order may show is scanned
			if (!serial) {
				const u64 recm = __ballot(rec) & (cb >= 63 ? ~0ULL : (2ULL << cb) - 1);
				if (recm) { const int last = 63 - __clzll(recm); max_f = LQ_WAVE_LANE(sc, last); max_j = (i32)i - 1 - last; }
				done = cb < 64 ? 1 : 0;
				n_skip = LQ_WAVE_LANE(x_after, 63);
			}
		}
		if (serial) {                                              // one by one, with the rule of TieGroup
			cand[ln].sc = sc; cand[ln].j = j; cand[ln].flags = (active ? 1 : 0) | (tm ? 2 : 0); cand[ln].x32 = wx;
			if (ln == 0) { st_sh[0] = max_f; st_sh[1] = max_j; st_sh[2] = 0; st_sh[3] = 0; st_sh[8] = 0; }
			LQ_WAVE_SYNC();
			if (ln == 0) lq_wave_replay(cand, st_sh, i - st < 64 ? i - st : 64, more, watch, max_skip);
			LQ_WAVE_SYNC();
			max_f = st_sh[0]; max_j = st_sh[1]; n_skip = st_sh[2]; done = st_sh[3];
		}
		// ---- candidates beyond the window (rare: 64 of them scanned without the break) ----
		bool beyond = false;
		if (!done && more) {
			LQ_BLOCK_SYNC();
			if (ln == 0) { st_sh[0] = max_f; st_sh[1] = max_j; st_sh[2] = n_skip; st_sh[3] = 0; if (!serial) st_sh[8] = 0; }
			LQ_BLOCK_SYNC();
			for (i64 top = (i64)i - 65; top >= (i64)st; top -= 64) {
				const i64 jj = top - (i64)ln;
				i32 c_sc = 0, c_flags = 0; u32 c_x32 = 0;
				if (jj >= (i64)st) {
					const mm128 aj = a[jj];
					c_x32 = (u32)aj.x;
					const i64 dr = (i64)(ri - aj.x);
					const i32 dq = qi - (i32)aj.y;
					if (!(dr == 0 || dq <= 0 || dq > max_dist)) {
						const i32 dd = dr > dq ? (i32)(dr - dq) : (i32)(dq - dr);
						if (dd <= bw) {
							const i32 min_d = dq < dr ? dq : (i32)dr;
							i32 s2 = min_d > q_span ? q_span : min_d;
							const i32 log_dd = dd ? lq_ilog2_32((u32)dd) : 0;
							s2 -= (i32)((double)dd * .01 * (double)avg_qspan) + (log_dd >> 1);     // chain.c:67
							c_sc = s2 + f[jj];
							c_flags = 1;
							const i32 pj = p[jj];
							if (pj >= 0) t[pj] = (i32)i;                  // chain.c:76
						}
					}
				}
				cand[ln].sc = c_sc; cand[ln].j = (i32)jj; cand[ln].flags = c_flags; cand[ln].x32 = c_x32;
				LQ_BLOCK_SYNC();
				if (cand[ln].flags & 1) { if (t[cand[ln].j] == (i32)i) cand[ln].flags |= 2; }
				LQ_BLOCK_SYNC();
				if (ln == 0) lq_wave_replay(cand, st_sh, top - (i64)st + 1 < 64 ? top - (i64)st + 1 : 64, top - 64 >= (i64)st, watch, max_skip);
				LQ_BLOCK_SYNC();
				if (st_sh[3]) break;
			}
			max_f = st_sh[0]; max_j = st_sh[1];
			LQ_BLOCK_SYNC();
			beyond = true;
		}
		// the run is known to be listed (first pass): it is not chained, nothing it would compute from here on is used -- and the
		// longest runs of all are pile-ups of one repeated minimizer, where that is known after a few anchors (uniform: LDS)
		if (watch && (serial || beyond) && st_sh[4]) break;
		// ---- f, p, v of anchor i; the window moves on ----
		i32 vi = max_f;
		if (max_j >= 0) {
			const u32 back = i - 1 - (u32)max_j;
			if (back >= 64) LQ_BLOCK_SYNC();                         // (v[] of an anchor that has left the window: from memory, where its lane stored it)
			const i32 vj = back < 64 ? LQ_WAVE_LANE(wv, back) : v[max_j];      // (uniform branch: max_j is)
			if (vj > max_f) vi = vj;
		}
		wx = (u32)lq_wave_shr1((i32)wx, (i32)rx); wy = lq_wave_shr1(wy, qi); wf = lq_wave_shr1(wf, max_f); wp = lq_wave_shr1(wp, max_j); wv = lq_wave_shr1(wv, vi);
		if (bi == 63 || i + 1 == n32) {                          // (uniform) lane c holds anchor i - c: the anchors of this block of 64 go to memory
			if (ln <= bi) { const u32 k = i - ln; f[k] = wf; p[k] = wp; v[k] = wv; }
		}
	}
	LQ_BLOCK_SYNC();
	const bool listed = st_sh[4] != 0;                          // (uniform)
	LQ_BLOCK_SYNC();
	if (C.rec) {                                                 // replay of a saturated query (rare): the serial second half records the chains
		if (ln == 0) lq_chain_finish(a, n, f, p, t, v, u, q, accumulate, P, C, false);
		return;
	}
	const bool peak_tie = !listed && lq_chain_finish_wave(a, n, f, p, t, v, u, q, accumulate, P, C, watch, st_sh);
	if (ln == 0 && (listed || peak_tie)) lq_tie_list(C, q, lq_hi32(a), listed ? (u32)st_sh[4] : LQ_TIE_WHY_PEAK);
}

// ---- filter_redundant_coords (lqmap.c:25-100), one thread per query, on this part's intervals ----
// ivl is sorted by q; ivq_off[q] = first interval of query q.
__global__ void k_ivl_offsets(const u32 *qkey, u32 n_ivl, u32 n_q, u32 *ivq_off)
{
	u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q > n_q) return;
	u32 lo = 0, hi = n_ivl;                                 // first index with qkey >= q
	while (lo < hi) { u32 mid = lo + ((hi - lo) >> 1); if (qkey[mid] < q) lo = mid + 1; else hi = mid; }
	ivq_off[q] = lo;
}

__global__ void k_filter_redundant(const u64 *se /* start | end<<32, sorted by query */, const u32 *ivq_off, u32 n_q, u32 min_cov,
                                   u32 *scratch /* 4 u32 per interval */, Ivl *pv, u32 *n_pv, u32 pv_cap)
{
	u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_q) return;
	const u32 lo = ivq_off[q], n = ivq_off[q + 1] - lo;
	if (n == 0) return;
	const u64 *cv = se + lo;
	u32 *vc = scratch + (u64)lo * 4, *mc = vc + 2 * (u64)n;
	for (u32 i = 0; i < n; ++i) { vc[2 * i] = (u32)cv[i]; vc[2 * i + 1] = (u32)(cv[i] >> 32); }
	lq_heapsort_u32(vc, 2 * (i64)n);
	u32 med_start = 0, med_cov = 0, n_mc = 0;
	for (u32 j = 0; j < 2 * n; ++j) {
		const u32 e = vc[j], old = med_cov;
		if (e & 2) {
			if (e & 1) { if (e & 4) med_cov -= min_cov; else --med_cov; }
			else       { if (e & 4) med_cov += min_cov; else ++med_cov; }
		}
		if (old < min_cov && med_cov >= min_cov) med_start = e;
		else if (old >= min_cov && med_cov < min_cov) {
			const u32 mlen = (e >> 3) - med_start;                // sic (lqmap.c:63): decoded minus encoded
			if (mlen > 0) {
				mc[2 * n_mc] = med_start; mc[2 * n_mc + 1] = e; ++n_mc;
				u32 s = atomicAdd(n_pv, 1u);
				if (s < pv_cap) { Ivl m; m.q = q; m.start = med_start | 4u; m.end = e | 4u; pv[s] = m; }
			}
		}
	}
	for (u32 i = 0; i < n; ++i) {
		const u32 s0 = (u32)cv[i], e0 = (u32)(cv[i] >> 32);
		bool inside = false;
		for (u32 j = 0; j < n_mc; ++j) if (s0 >= mc[2 * j] && e0 <= mc[2 * j + 1]) { inside = true; break; }
		if (!inside) {
			u32 s = atomicAdd(n_pv, 1u);
			if (s < pv_cap) { Ivl m; m.q = q; m.start = s0; m.end = e0; pv[s] = m; }
		}
	}
}

__global__ void k_split_ivl(const Ivl *iv, u32 n, u32 *qkey, u64 *se)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	qkey[i] = iv[i].q; se[i] = (u64)iv[i].start | (u64)iv[i].end << 32;
}

// ---- pass 2 (minimap2-coverage.c:545-566): compute_reliable_region (lqutils.c:83-155) ----------
// one thread per query over its persisted intervals (sorted by query); two sweeps: count, write.
struct RegionT { u32 start, end; };

template <bool WRITE>
__device__ __forceinline__ void lq_sweep(const u32 *vc, u32 n2, u32 min_cov, RegionT *regs, RegionT *mregs, u32 &n_reg, u32 &n_mreg)
{
	u32 start = 0, cov = 0, med_start = 0, med_cov = 0;
	n_reg = n_mreg = 0;
	for (u32 j = 0; j < n2; ++j) {
		const u32 e = vc[j], old_cov = cov, old_med = med_cov;
		if (e & 1) {
			--cov;
			if (e & 2) { if (e & 4) { med_cov -= min_cov; cov -= (min_cov - 1); } else --med_cov; }
		} else {
			++cov;
			if (e & 2) { if (e & 4) { med_cov += min_cov; cov += (min_cov - 1); } else ++med_cov; }
		}
		if (old_cov < min_cov && cov >= min_cov) {
			start = e >> 3;
			if (old_med < min_cov && med_cov >= min_cov) med_start = e >> 3;
		} else if (old_cov >= min_cov && cov < min_cov) {
			if ((e >> 3) - start > 0) { if (WRITE) { regs[n_reg].start = start; regs[n_reg].end = e >> 3; } ++n_reg; }
			if (old_med >= min_cov && med_cov < min_cov)
				if ((e >> 3) - med_start > 0) { if (WRITE) { mregs[n_mreg].start = med_start; mregs[n_mreg].end = e >> 3; } ++n_mreg; }
		} else if (old_med < min_cov && med_cov >= min_cov) {
			med_start = e >> 3;
		} else if (old_med >= min_cov && med_cov < min_cov) {
			if ((e >> 3) - med_start > 0) { if (WRITE) { mregs[n_mreg].start = med_start; mregs[n_mreg].end = e >> 3; } ++n_mreg; }
		}
	}
}

struct RowDev {               // device-side mirror of lqcov_row's integer part
	u32 n_match, reg_off, n_reg, mreg_off, n_mreg;
};

__global__ void k_reliable(const u64 *se, const u32 *pvq_off, u32 n_q, u32 min_cov, u32 *scratch /* 2 u32 per interval */,
                           RegionT *regs, u32 *n_regs, RegionT *mregs, u32 *n_mregs, RowDev *rows)
{
	u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_q) return;
	const u32 lo = pvq_off[q], n = pvq_off[q + 1] - lo;
	u32 *vc = scratch + (u64)lo * 2;
	for (u32 i = 0; i < n; ++i) { vc[2 * i] = (u32)se[lo + i]; vc[2 * i + 1] = (u32)(se[lo + i] >> 32); }
	lq_heapsort_u32(vc, 2 * (i64)n);
	u32 nr, nm;
	lq_sweep<false>(vc, 2 * n, min_cov, nullptr, nullptr, nr, nm);
	u32 ro = nr ? atomicAdd(n_regs, nr) : 0, mo = nm ? atomicAdd(n_mregs, nm) : 0;
	lq_sweep<true>(vc, 2 * n, min_cov, regs + ro, mregs + mo, nr, nm);
	rows[q].reg_off = ro; rows[q].n_reg = nr; rows[q].mreg_off = mo; rows[q].n_mreg = nm;
}

// minimap2-coverage.c:552-561: integer mean of the uint16 counters, count of those above it
__global__ void k_cnt_stats(const u32 *cnts, const u64 *cnt_off, const u32 *nsize, u32 n_q, RowDev *rows, u32 *qflags, u32 cnt_max)
{
	u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_q) return;
	const u64 lo = cnt_off[q], hi = nsize ? lo + nsize[q] : cnt_off[q + 1];   // mv.n counters (minimap2-coverage.c:552-561)
	u32 sum = 0, n = (u32)(hi - lo), nm = 0;
	bool sat = false;
	for (u64 j = lo; j < hi; ++j) { sum += cnts[j] & cnt_max; if (cnts[j] >= cnt_max) sat = true; }   // uint16 storage wraps (esterr.c:136); a flagged query's counters are the replayed ones by now
	if (sat) atomicOr(&qflags[q], 1u);
	if (nsize) {                                            // counters the reference never allocated (see adopt_index_params)
		bool over = false;
		for (u64 j = hi; j < cnt_off[q + 1]; ++j) if (cnts[j]) over = true;
		if (over) atomicOr(&qflags[q], 2u);
	}
	if (n) sum /= n;
	for (u64 j = lo; j < hi; ++j) if ((cnts[j] & cnt_max) > sum) ++nm;
	rows[q].n_match = nm;
}

// meanQ's accumulation (lqutils.c:51-56): a sequential double sum per read, in read order
__global__ void k_qual_sum(const u8 *qual, const u64 *seq_off, u32 n_q, const double *q2p, double *psum)
{
	u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_q) return;
	double s = 0.0;
	for (u64 i = seq_off[q]; i < seq_off[q + 1]; ++i) {
		int v = (int)(signed char)qual[i] - 33;
		v = v < 0 ? 0 : v > 126 ? 126 : v;                         // the reference indexes out of bounds outside Q0..Q126
		s += q2p[v];
	}
	psum[q] = s;
}
