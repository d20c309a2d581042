// longqc_amd/csrc/kernels_seed.hpp -- which seed hits can be part of a chain at all, decided from a bucketed stream.
//
// Reference: collect_seed_hits (lqmap.c:140-205) emits every occurrence of every query minimizer; mm_chain_dp (chain.c:41-137)
// then drops almost all of them.  Against a 4-Gbase part a 10-kb query collects about a million seed hits of which a few
// thousand lie on true overlaps.  Which hits can matter is decided first, exactly (no false negatives):
//   * mm_chain_dp never lets anchors of different (strand, rid) interact (chain.c:47: x carries both), and inside one
//     (strand, rid) run anchor j is looked at by the scan of anchor i only when 0 < dq, dr <= max_gap and |dr - dq| <= bw (the
//     `continue`s of chain.c:52-56 come before any state changes).  dr - dq is the difference of the two anchors' diagonals
//     d = x - y, so two anchors that can interact lie at most bw apart in d, and the anchors of one connected component of "can
//     interact" fill a gap-free stretch of diagonal bins of width D > bw.
//   * a chain lives inside one component, needs min_cnt anchors and scores at most the sum of their spans (chain.c:57-67,
//     119-121): a component of fewer than n_min = max(min_cnt, ceil(min_sc / span_max)) anchors yields nothing and, being
//     invisible to the scans of every other component, can be left out without changing f, p, v of anything else.
//   So a hit survives iff (a) its (query, target, strand) pair holds n_min hits and (b) the gap-free stretch of non-empty
//   diagonal bins around its own bin holds n_min hits.  Counters that saturate or alias only ever add: false positives cost
//   what every hit used to cost, and the run list (kernels_chain.hpp) still decides exactly.
//
// Round 4 counted in 128 KiB of LDS per query and walked every occurrence list in ~22-hit pieces, once per slice of targets
// and twice per slice: 3 % of the HBM roofline, 5-10x its bytes in traffic.  Here the lists are streamed:
//   k_seed_count    per segment (a run of one query's minimizers): every list read once, front to back; hits per slice of
//                   targets counted (the lists ascend in rid, index.c:188: a slice is one stretch of every list)
//   (scan)          -> where every (query, slice, segment) piece starts: the buckets (query, slice) are contiguous
//   k_seed_scatter  the lists once more; every hit becomes an 8-byte record {rid, relative strand, diagonal, minimizer} and
//                   goes to its bucket -- a tile of records is sorted by slice in LDS and leaves as runs
//   k_seed_decide   per bucket: the records in registers, pair counters and hashed diagonal-bin counters in LDS (one fresh
//                   table per bucket: no aliasing inside a pair, bins as wide as the band), survivors compacted in place
//   k_seed_collect  survivors of all buckets, dense, in (query, slice) order: the part's seed plan keeps them
//   k_seed_emit_s   (when a batch is mapped) survivors -> anchors (lqmap.c:190-197)
// Everything is deterministic: ranks come from wave-private counts scanned in (slice, wave) order, never from the order in
// which atomics happen to land.
// avg_qspan, mini_pos and the lq_cnt_match prologue keep using the unfiltered totals (chain.c:37-38, lqmap.c:174).  Exact only
// together with a sort that does not need klib's walk over the *whole* query: map_batch's first pass.
#pragma once
#include "lq_common.hpp"
#include "kernels_index.hpp"

#define LQ_SD_SEGL 2048u                    // most minimizers of a segment (their hit offsets sit in LDS)
#define LQ_SD_THREADS 512
#define LQ_SD_WAVES (LQ_SD_THREADS / 64)
#define LQ_SD_RPT 8                         // hits per thread and tile of the count / scatter kernels
#define LQ_SD_TILE (LQ_SD_THREADS * LQ_SD_RPT)
#define LQ_SD_SL_SMALL 256u                 // slices per query: the two shapes of the scatter kernel
#define LQ_SD_SL_BIG 1024u

// record = rid << (jb + db + 1) | relative strand << (jb + db) | diagonal << jb | minimizer (index inside its query)
struct SeedBits { u32 jb, db; };
__device__ __forceinline__ u32 sd_rid(u64 r, SeedBits b) { return (u32)(r >> (b.jb + b.db + 1)); }
__device__ __forceinline__ u32 sd_rs(u64 r, SeedBits b) { return (u32)(r >> (b.jb + b.db)) & 1u; }
__device__ __forceinline__ u32 sd_diag(u64 r, SeedBits b) { return (u32)(r >> b.jb) & ((1u << b.db) - 1u); }
__device__ __forceinline__ u32 sd_jl(u64 r, SeedBits b) { return (u32)r & ((1u << b.jb) - 1u); }

// a query's slices of targets: slice of rid = (rid * mul) >> 32, monotone, below nsl for every rid of the part
// (mul = floor(2^32 * nsl / n_targets)); a segment = minimizers [j0, j1) of one query; cb: the query's first entry in the
// chunk's piece table, laid out [slice][segment]
struct alignas(16) SeedQ { u32 mul, nsl, nseg, seg0; u64 cb; u64 bk; };   // bk: the query's first bucket in the chunk
struct alignas(16) SeedSeg { u64 j0, j1; u32 q, ord; u32 pad0, pad1; };  // ord: the segment's number inside its query
__device__ __forceinline__ u32 sd_slice(u32 rid, u32 mul) { return (u32)(((u64)rid * mul) >> 32); }
// first rid of slice s (the smallest rid with sd_slice(rid) == s)
__device__ __forceinline__ u32 sd_slice_first(u32 s, u32 mul) { return s == 0 ? 0u : (u32)((((u64)s << 32) + mul - 1) / mul); }

// last i in [0, n) with off[i] <= v (off[0] <= v): entries that start where the next one starts are stepped over
__device__ __forceinline__ u32 lq_find_seg32(const u32 *off, u32 n, u32 v)
{
	u32 lo = 0, hi = n;
	while (hi - lo > 1) { const u32 mid = lo + ((hi - lo) >> 1); if (off[mid] <= v) lo = mid; else hi = mid; }
	return lo;
}

// exclusive prefix sum over the block (THREADS a multiple of 64, at most 1024); ws: 17 words of LDS; *total: sum of all
__device__ __forceinline__ u32 sd_block_exscan(u32 v, u32 *ws, u32 *total)
{
	const u32 t = threadIdx.x, lane = t & 63, w = t >> 6, nw = blockDim.x >> 6;
	u32 x = v;
	for (u32 o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o); if (lane >= o) x += y; }
	__syncthreads();                                          // (ws may still be read from the call before)
	if (lane == 63) ws[w] = x;
	__syncthreads();
	if (t == 0) { u32 acc = 0; for (u32 i = 0; i < nw; ++i) { const u32 c = ws[i]; ws[i] = acc; acc += c; } ws[16] = acc; }
	__syncthreads();
	*total = ws[16];
	return ws[w] + x - v;
}

// last i in [0, n) with off[i] <= g (off[0] = 0 <= g): the minimizer whose list holds hit g of the segment; lists of no hits
// share their offset with the next one and are stepped over
__device__ __forceinline__ u32 sd_owner(const u32 *off, u32 n, u32 g)
{
	u32 lo = 0, hi = n;
	while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (off[mid] <= g) lo = mid; else hi = mid; }
	return lo;
}

// One step of a wave over 64 consecutive hits of the segment (valid lanes are a prefix): counts them per slice in the wave's
// own column of `hist` (hist[d * LQ_SD_WAVES + wave]) and returns every hit's rank among the wave's hits of its slice so far.
// Hits of one list ascend in rid, so inside a list a slice is one run: its first lane adds the run's length.  Lists are
// taken one after the other (a step rarely touches more than two): no two lanes of one instruction ever add to one counter,
// and the counts a lane sees do not depend on how the hardware orders atomics.
__device__ __forceinline__ u32 sd_rank_step(u32 *hist, u32 wave, u32 d, u32 jl, bool valid, u32 lane)
{
	const u32 jp = __shfl_up(jl, 1), dp = __shfl_up(d, 1);
	const bool lhead = valid && (lane == 0 || jl != jp);
	const bool rhead = valid && (lhead || d != dp);
	u64 Lm = __ballot(lhead);
	const u64 Rm = __ballot(rhead);
	const u32 nvalid = (u32)__popcll(__ballot(valid));
	const u64 upto = (2ULL << lane) - 1ULL;                   // lanes 0 .. lane
	const u64 below = Rm & upto;
	const u32 hp = below ? 63u - (u32)__clzll(below) : 0u;    // my run's first lane
	const u64 above = Rm & ~upto;
	const u32 nxt = above ? (u32)__ffsll((unsigned long long)above) - 1u : 64u;
	const u32 len = (nxt < nvalid ? nxt : nvalid) - lane;     // (for a run's first lane: the run's length)
	u32 rank = 0;
	while (Lm) {                                              // (wave-uniform)
		const u32 lo = (u32)__ffsll((unsigned long long)Lm) - 1u;
		Lm &= Lm - 1;
		const u32 hi = Lm ? (u32)__ffsll((unsigned long long)Lm) - 1u : 64u;
		const bool in = valid && lane >= lo && lane < hi;
		u32 base = 0;
		if (in && rhead) base = atomicAdd(&hist[d * LQ_SD_WAVES + wave], len);
		const u32 b = __shfl(base, (int)hp);
		if (in) rank = b + (lane - hp);
	}
	return rank;
}

struct SeedIn {                             // what the count / scatter kernels read
	const SeedSeg *segs; const SeedQ *qg;
	const u64 *h_off;                       // exclusive scan of the kept minimizers' list lengths (n_qm + 1 entries)
	const u64 *hit_start; const u64 *pos;   // k_seed_probe's list starts; the part's occurrence lists
	const u64 *qx, *qy, *qmoff; const u32 *qlen;
};

// hit g of the segment (offsets relative to its first hit): which minimizer, which occurrence
__device__ __forceinline__ u64 sd_load_hit(const SeedIn &in, const u32 *loff, u32 nj, u64 j0, u32 g, u32 &jl)
{
	jl = sd_owner(loff, nj, g);
	return in.pos[in.hit_start[j0 + jl] + (g - loff[jl])];
}

// ---- hits per (query, slice, segment) ---------------------------------------------------------------------------------------
// One block per segment.  cnt[cb + slice * nseg + ord] = the segment's hits in that slice.
__global__ void __launch_bounds__(LQ_SD_THREADS)
k_seed_count(SeedIn in, u32 g_lo, u32 *cnt)
{
	__shared__ u32 loff[LQ_SD_SEGL + 1];
	__shared__ u32 hist[LQ_SD_SL_BIG * LQ_SD_WAVES];
	const SeedSeg sg = in.segs[g_lo + blockIdx.x];
	const SeedQ Q = in.qg[sg.q];
	const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const u32 nj = (u32)(sg.j1 - sg.j0);
	const u64 H0 = in.h_off[sg.j0];
	const u32 nH = (u32)(in.h_off[sg.j1] - H0);
	for (u32 i = t; i <= nj; i += LQ_SD_THREADS) loff[i] = (u32)(in.h_off[sg.j0 + i] - H0);
	for (u32 i = t; i < Q.nsl * LQ_SD_WAVES; i += LQ_SD_THREADS) hist[i] = 0;
	__syncthreads();
	for (u32 base = 0; base < nH; base += LQ_SD_TILE) {
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = base + (u32)k * LQ_SD_THREADS + t;
			const bool valid = g < nH;
			u32 jl = 0, d = 0;
			if (valid) { const u64 r = sd_load_hit(in, loff, nj, sg.j0, g, jl); d = sd_slice((u32)(r >> 32), Q.mul); }
			if (base + (u32)k * LQ_SD_THREADS + (t & ~63u) < nH) sd_rank_step(hist, wave, d, jl, valid, lane);   // (wave-uniform)
		}
	}
	__syncthreads();
	for (u32 s = t; s < Q.nsl; s += LQ_SD_THREADS) {
		u32 c = 0;
		for (u32 w = 0; w < LQ_SD_WAVES; ++w) c += hist[s * LQ_SD_WAVES + w];
		cnt[Q.cb + (u64)s * Q.nseg + sg.ord] = c;
	}
}

// ---- records to their buckets -----------------------------------------------------------------------------------------------
// One block per segment; off = exclusive scan of cnt: where the segment's piece of every bucket starts in `rec`.
template <u32 MAXSL>
__global__ void __launch_bounds__(LQ_SD_THREADS)
k_seed_scatter(SeedIn in, u32 g_lo, const u32 *off, SeedBits bits, u32 span_const /* 0: from qx (-H) */, u64 *rec)
{
	__shared__ u32 loff[LQ_SD_SEGL + 1];
	__shared__ u32 hist[MAXSL * LQ_SD_WAVES];                // per tile: counts, then first places, per (slice, wave)
	__shared__ u32 cursor[MAXSL];                            // where the segment's piece of every bucket goes on
	__shared__ u32 dbase[MAXSL + 1];                         // first place of every slice in the sorted tile
	__shared__ u64 sbuf[LQ_SD_TILE];
	__shared__ u32 ws[17];
	const SeedSeg sg = in.segs[g_lo + blockIdx.x];
	const SeedQ Q = in.qg[sg.q];
	const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const u32 nj = (u32)(sg.j1 - sg.j0);
	const u64 H0 = in.h_off[sg.j0];
	const u32 nH = (u32)(in.h_off[sg.j1] - H0);
	const u64 jq0 = in.qmoff[sg.q];
	const i32 ql = (i32)in.qlen[sg.q];
	for (u32 i = t; i <= nj; i += LQ_SD_THREADS) loff[i] = (u32)(in.h_off[sg.j0 + i] - H0);
	for (u32 s = t; s < Q.nsl; s += LQ_SD_THREADS) cursor[s] = off[Q.cb + (u64)s * Q.nseg + sg.ord];
	const u32 nE = Q.nsl * LQ_SD_WAVES;                      // entries of hist in use
	const u32 per = (nE + LQ_SD_THREADS - 1) / LQ_SD_THREADS;
	for (u32 base = 0; base < nH; base += LQ_SD_TILE) {
		for (u32 i = t; i < nE; i += LQ_SD_THREADS) hist[i] = 0;
		__syncthreads();
		u64 rc[LQ_SD_RPT]; u32 dg[LQ_SD_RPT], rk[LQ_SD_RPT];
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = base + (u32)k * LQ_SD_THREADS + t;
			const bool valid = g < nH;
			u32 jl = 0; dg[k] = 0; rc[k] = 0; rk[k] = 0;
			if (valid) {
				const u64 r = sd_load_hit(in, loff, nj, sg.j0, g, jl);
				const u64 j = sg.j0 + jl;
				const u32 qyv = (u32)in.qy[j], qpos = qyv >> 1;
				const u32 span = span_const ? span_const : (u32)(in.qx[j] & 0xff);
				const u32 rid = (u32)(r >> 32), rpos = (u32)r >> 1, rs = ((u32)r & 1u) ^ (qyv & 1u);
				const i32 ypos = rs ? ql - (i32)(qpos + 1 - span) - 1 : (i32)qpos;     // the anchor's query coordinate (lqmap.c:191-197)
				const u32 diag = (u32)((i32)rpos - ypos + ql + 256);                     // (never negative: ypos <= qlen)
				dg[k] = sd_slice(rid, Q.mul);
				rc[k] = (u64)rid << (bits.jb + bits.db + 1) | (u64)rs << (bits.jb + bits.db) | (u64)diag << bits.jb | (u64)(j - jq0);
			}
			if (base + (u32)k * LQ_SD_THREADS + (t & ~63u) < nH) rk[k] = sd_rank_step(hist, wave, dg[k], jl, valid, lane);
		}
		__syncthreads();
		// counts -> first places, in (slice, wave) order: a thread sums a stretch of `per` entries, the block scans the sums
		u32 mine = 0;
		for (u32 i = 0; i < per; ++i) { const u32 e = t * per + i; if (e < nE) mine += hist[e]; }
		u32 total = 0;
		u32 run = sd_block_exscan(mine, ws, &total);
		for (u32 i = 0; i < per; ++i) {
			const u32 e = t * per + i;
			if (e < nE) { const u32 c = hist[e]; hist[e] = run; if (e % LQ_SD_WAVES == 0) dbase[e / LQ_SD_WAVES] = run; run += c; }
		}
		if (t == 0) dbase[Q.nsl] = total;
		__syncthreads();
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = base + (u32)k * LQ_SD_THREADS + t;
			if (g < nH) sbuf[hist[dg[k] * LQ_SD_WAVES + wave] + rk[k]] = rc[k];
		}
		__syncthreads();
		for (u32 i = t; i < total; i += LQ_SD_THREADS) {
			const u64 r = sbuf[i];
			const u32 d = sd_slice(sd_rid(r, bits), Q.mul);
			rec[cursor[d] + (i - dbase[d])] = r;
		}
		__syncthreads();
		for (u32 s = t; s < Q.nsl; s += LQ_SD_THREADS) cursor[s] += dbase[s + 1] - dbase[s];
		// (the next tile's first barrier comes before anything reads cursor again)
	}
}

// ---- which records survive --------------------------------------------------------------------------------------------------
struct SeedDecide {
	u32 n_min;                              // hits a component needs (run_n_min; >= 2 here: without a filter nothing is bucketed)
	u32 dshift;                             // log2 of the bin width, D > bw
	u32 pair_bits, bin_bits;                // counters in use (powers of two, at most the tables' sizes; tests shrink them)
	int no_self;
};
#define LQ_SD_DTHREADS 512
#define LQ_SD_DRPT 16                       // records a thread holds: a bucket of up to 8192 records is read once
#define LQ_SD_DCAP (LQ_SD_DTHREADS * LQ_SD_DRPT)
#define LQ_SD_PAIR_BITS 13                  // 8192 pair counters (16 bits each)
#define LQ_SD_BIN_BITS 15                   // 32768 bin counters (4 bits each)

// 16-bit counters, two to a word; they stop near 0x8000 (a count that large is "enough" for every n_min in use)
__device__ __forceinline__ void sd_pair_inc(u32 *tab, u32 p)
{
	const u32 w = p >> 1, sh = (p & 1u) << 4;
	if ((tab[w] >> sh & 0xffffu) < 0x8000u) atomicAdd(&tab[w], 1u << sh);     // (at most blockDim more adds can slip past the test: no carry)
}
__device__ __forceinline__ u32 sd_pair_get(const u32 *tab, u32 p) { return tab[p >> 1] >> ((p & 1u) << 4) & 0xffffu; }
// 4-bit counters, eight to a word, saturating at 15
__device__ __forceinline__ void sd_bin_inc(u32 *tab, u32 b)
{
	const u32 w = b >> 3, sh = (b & 7u) << 2;
	u32 old = tab[w];
	for (;;) {
		if ((old >> sh & 15u) == 15u) return;
		const u32 seen = atomicCAS(&tab[w], old, old + (1u << sh));
		if (seen == old) return;
		old = seen;
	}
}
__device__ __forceinline__ u32 sd_bin_get(const u32 *tab, u32 b) { return tab[b >> 3] >> ((b & 7u) << 2) & 15u; }
// does the gap-free stretch of non-empty bins around bin b hold n_min hits?  (a saturated bin counts as enough)
__device__ __forceinline__ bool sd_alive(const u32 *tab, u32 b, u32 mask, u32 n_min)
{
	const u32 own = sd_bin_get(tab, b);
	if (own >= 15u || own >= n_min) return true;
	const u32 side = n_min - 1;                               // with `side` non-empty neighbours in a row on one side there are n_min hits
	u32 tot = own;
	for (u32 k = 1; k <= side; ++k) { const u32 c = sd_bin_get(tab, (b + k) & mask); if (c == 0) break; if (c >= 15u) return true; tot += c; if (k == side) return true; }
	for (u32 k = 1; k <= side; ++k) { const u32 c = sd_bin_get(tab, (b - k) & mask); if (c == 0) break; if (c >= 15u) return true; tot += c; if (k == side) return true; }
	return tot >= n_min;
}

struct SeedDecIn {
	const SeedQ *qg; const u32 *bq;         // bq: first bucket of every query of the chunk (n_qc + 1 entries), q_lo: its first query
	u32 q_lo, n_qc;
	const u32 *off;                         // piece table (exclusive scan of cnt), + 1 sentinel
	const u64 *qx, *qy, *qmoff; const u32 *qlen;
	const u32 *self_off, *self_rid; AvaView ava;
};

// what deciding one record needs (block-uniform)
struct SeedCtx {
	const u32 *pairs, *bins;
	SeedBits bits; u32 rid0, pmask, bmask, bin_bits, dshift, n_min, span_const, q, qlo;
	i32 ql; u64 jq0; bool self_q, rare;
};
__device__ __forceinline__ u32 sd_pair_of(u64 r, const SeedCtx &c) { return (((sd_rid(r, c.bits) - c.rid0) << 1) | sd_rs(r, c.bits)) & c.pmask; }
// A pair's bins start at a hashed place and follow each other: neighbours inside a pair are neighbours in the table, different
// pairs meet only by chance (and then only add)
__device__ __forceinline__ u32 sd_bin_of(u64 r, u32 p, const SeedCtx &c) { return (((p * 0x9E3779B1u) >> (32u - c.bin_bits)) + (sd_diag(r, c.bits) >> c.dshift)) & c.bmask; }
__device__ __forceinline__ bool sd_keep(u64 r, const SeedCtx &c, const SeedDecIn &in)
{
	const u32 p = sd_pair_of(r, c);
	if (sd_pair_get(c.pairs, p) < c.n_min) return false;
	if (!sd_alive(c.bins, sd_bin_of(r, p, c), c.bmask, c.n_min)) return false;
	if (c.rare) {                                             // the self diagonal and -X (lqmap.c:180-187)
		const u32 rid = sd_rid(r, c.bits);
		if (c.self_q) {
			const u64 j = c.jq0 + sd_jl(r, c.bits);
			const u32 qpos = (u32)in.qy[j] >> 1, span = c.span_const ? c.span_const : (u32)(in.qx[j] & 0xff);
			const i32 ypos = sd_rs(r, c.bits) ? c.ql - (i32)(qpos + 1 - span) - 1 : (i32)qpos;
			const u32 rpos = (u32)((i32)sd_diag(r, c.bits) + ypos - c.ql - 256);
			if (rpos == qpos && lq_is_self(in.self_off, in.self_rid, c.q, rid)) return false;
		}
		if (in.ava.t_rank && in.ava.t_rank[rid] < c.qlo) return false;
	}
	return true;
}

// One block per bucket (query, slice): records [off(q, s, 0), off(q, s + 1, 0)) of `rec`.  Survivors are moved to the front of
// the bucket, in a fixed order; scnt[bucket] = how many.
__global__ void __launch_bounds__(LQ_SD_DTHREADS)
k_seed_decide(SeedDecIn in, SeedDecide dp, SeedBits bits, u32 span_const, u64 *rec, u32 *scnt)
{
	__shared__ u32 pairs[1u << (LQ_SD_PAIR_BITS - 1)];
	__shared__ u32 bins[1u << (LQ_SD_BIN_BITS - 3)];
	__shared__ u32 ws[17];
	const u32 t = threadIdx.x;
	const u32 bk = blockIdx.x;
	const u32 qi = lq_find_seg32(in.bq, in.n_qc, bk);
	const u32 q = in.q_lo + qi, s = bk - in.bq[qi];
	const SeedQ Q = in.qg[q];
	const u64 e0 = Q.cb + (u64)s * Q.nseg;
	const u32 b0 = in.off[e0], n = in.off[e0 + Q.nseg] - b0;     // (the entry after a query's last one is the next query's first, or the sentinel)
	if (n == 0) { if (t == 0) scnt[bk] = 0; return; }
	SeedCtx c;
	c.pairs = pairs; c.bins = bins; c.bits = bits; c.rid0 = sd_slice_first(s, Q.mul);
	c.pmask = (1u << dp.pair_bits) - 1u; c.bmask = (1u << dp.bin_bits) - 1u; c.bin_bits = dp.bin_bits; c.dshift = dp.dshift; c.n_min = dp.n_min;
	c.span_const = span_const; c.q = q; c.qlo = in.ava.q_lo ? in.ava.q_lo[q] : 0;
	c.ql = (i32)in.qlen[q]; c.jq0 = in.qmoff[q];
	c.self_q = dp.no_self && in.self_off[q] != in.self_off[q + 1];
	c.rare = c.self_q || in.ava.t_rank != nullptr;
	for (u32 i = t; i < (1u << dp.pair_bits) / 2; i += LQ_SD_DTHREADS) pairs[i] = 0;      // (pair_bits >= 1, bin_bits >= 3: at least a word each)
	for (u32 i = t; i < (1u << dp.bin_bits) / 8; i += LQ_SD_DTHREADS) bins[i] = 0;
	u64 *R = rec + b0;
	if (n <= LQ_SD_DCAP) {
		// the bucket is read once: its records stay in registers through the three phases
		u64 rc[LQ_SD_DRPT];
#pragma unroll
		for (int k = 0; k < LQ_SD_DRPT; ++k) { const u32 i = (u32)k * LQ_SD_DTHREADS + t; rc[k] = i < n ? R[i] : ~0ULL; }
		__syncthreads();
		// (1) hits per (target, relative strand) pair
#pragma unroll
		for (int k = 0; k < LQ_SD_DRPT; ++k) if ((u32)k * LQ_SD_DTHREADS + t < n) sd_pair_inc(pairs, sd_pair_of(rc[k], c));
		__syncthreads();
		// (2) hits per diagonal bin, for the pairs that hold enough
#pragma unroll
		for (int k = 0; k < LQ_SD_DRPT; ++k) if ((u32)k * LQ_SD_DTHREADS + t < n) { const u32 p = sd_pair_of(rc[k], c); if (sd_pair_get(pairs, p) >= dp.n_min) sd_bin_inc(bins, sd_bin_of(rc[k], p, c)); }
		__syncthreads();
		// (3) decide, compact.  Survivors are ordered (thread, k): any fixed order will do
		u32 al = 0, mine = 0;
#pragma unroll
		for (int k = 0; k < LQ_SD_DRPT; ++k) if ((u32)k * LQ_SD_DTHREADS + t < n && sd_keep(rc[k], c, in)) { al |= 1u << k; ++mine; }
		u32 total = 0;
		u32 at = sd_block_exscan(mine, ws, &total);               // (every record was read before the scan's barriers)
#pragma unroll
		for (int k = 0; k < LQ_SD_DRPT; ++k) if (al >> k & 1u) R[at++] = rc[k];
		if (t == 0) scnt[bk] = total;
		return;
	}
	// a bucket beyond that (a query with more hits than slices can divide, targets of very different lengths): read once per phase
	__syncthreads();
	for (u32 i = t; i < n; i += LQ_SD_DTHREADS) sd_pair_inc(pairs, sd_pair_of(R[i], c));
	__syncthreads();
	for (u32 i = t; i < n; i += LQ_SD_DTHREADS) { const u64 r = R[i]; const u32 p = sd_pair_of(r, c); if (sd_pair_get(pairs, p) >= dp.n_min) sd_bin_inc(bins, sd_bin_of(r, p, c)); }
	__syncthreads();
	u32 done = 0;                                             // survivors written so far (block-uniform)
	for (u32 base = 0; base < n; base += LQ_SD_DTHREADS) {
		const u32 i = base + t;
		const u64 r = i < n ? R[i] : 0;
		const bool a = i < n && sd_keep(r, c, in);
		u32 total = 0;
		const u32 at = sd_block_exscan(a ? 1u : 0u, ws, &total);  // (its barriers stand between this round's reads and writes)
		if (a) R[done + at] = r;                                  // done + at <= i: never ahead of what is still to be read
		done += total;
	}
	if (t == 0) scnt[bk] = done;
}

// survivors of every bucket of the chunk, dense: surv[base + soff[bucket] ...] (soff = exclusive scan of scnt); and where
// every query's survivors start (aqf_off[q], for the chunk's queries)
__global__ void k_seed_collect(SeedDecIn in, const u64 *rec, const u32 *scnt, const u32 *soff, u64 base, u64 *surv, u64 *aqf_off)
{
	const u32 bk = blockIdx.x;
	const u32 qi = lq_find_seg32(in.bq, in.n_qc, bk);
	const u32 q = in.q_lo + qi, s = bk - in.bq[qi];
	const SeedQ Q = in.qg[q];
	if (s == 0 && threadIdx.x == 0) aqf_off[q] = base + soff[bk];
	const u32 n = scnt[bk];
	if (n == 0) return;
	const u64 *R = rec + in.off[Q.cb + (u64)s * Q.nseg];
	u64 *out = surv + base + soff[bk];
	for (u32 i = threadIdx.x; i < n; i += blockDim.x) out[i] = R[i];
}
// queries without a bucket in any chunk (no hits) and the end: aqf_off[q] = the next query's, filled from the back on the host side's order
__global__ void k_seed_fill_off(const u32 *has, u32 n_q, u64 total, u64 *aqf_off)
{
	// one thread: n_q is a few thousand
	if (blockIdx.x || threadIdx.x) return;
	u64 nxt = total;
	aqf_off[n_q] = total;
	for (u32 q = n_q; q-- > 0; ) { if (has[q]) nxt = aqf_off[q]; else aqf_off[q] = nxt; }
}

// ---- survivors -> anchors (lqmap.c:175-200) ---------------------------------------------------------------------------------
// the survivors [i0, i0 + n) of the plan, i.e. of the queries [q0, q1) of a batch; anchors[i - i0]
__global__ void __launch_bounds__(256)
k_seed_emit_s(const u64 *surv, u64 i0, u64 n, const u64 *aqf_off, u32 q0, u32 q1, SeedBits bits,
              const u64 *qx, const u64 *qy, const u64 *qmoff, const u32 *qlen, const u32 *dup, mm128 *anchors)
{
	const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u64 r = surv[i0 + i];
	const u32 q = q0 + lq_find_seg(aqf_off + q0, q1 - q0, i0 + i);
	const u64 j = qmoff[q] + sd_jl(r, bits);
	const u64 x = qx[j];
	const u32 span = (u32)(x & 0xff), qyv = (u32)qy[j], qpos = qyv >> 1;
	const i32 ql = (i32)qlen[q];
	const u32 rs = sd_rs(r, bits);
	const i32 ypos = rs ? ql - (i32)(qpos + 1 - span) - 1 : (i32)qpos;
	const u32 rpos = (u32)((i32)sd_diag(r, bits) + ypos - ql - 256);
	u64 ybits = dup[j] ? LQ_TIE_MARK : 0;
	if ((j > qmoff[q] && (qx[j - 1] >> 8) == (x >> 8)) || (j + 1 < qmoff[q + 1] && (qx[j + 1] >> 8) == (x >> 8))) ybits |= LQ_SEED_TANDEM;
	mm128 a;
	a.x = (u64)rs << 63 | (u64)sd_rid(r, bits) << 32 | rpos;
	a.y = (u64)span << 32 | (u32)ypos | ybits;
	anchors[i] = a;
}

// mini_pos (lqmap.c:174) of every kept minimizer
__global__ void k_mini_pos(const u64 *qx, const u64 *qy, const u32 *keep, const u64 *mp_off, u64 n_qm, u64 *mini_pos)
{
	const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n_qm && keep[j]) mini_pos[mp_off[j]] = (qx[j] & 0xff) << 32 | ((u32)qy[j] >> 1);
}
// list length of every kept minimizer (scanned into h_off)
__global__ void k_hit_len(const u32 *hit_n, const u32 *keep, u64 n_qm, u32 *len)
{
	const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n_qm) len[j] = keep[j] ? hit_n[j] : 0;
}
__global__ void k_query_hoff(const u64 *qmoff, const u64 *h_off, u32 n_q, u64 *hq_off)
{
	const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q <= n_q) hq_off[q] = h_off[qmoff[q]];
}
