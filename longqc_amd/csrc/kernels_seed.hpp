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

#ifndef LQ_SD_SEGL
#define LQ_SD_SEGL 256u                     // most minimizers of a segment (their hit offsets sit in LDS)
#endif
#ifndef LQ_SD_THREADS
#define LQ_SD_THREADS 512
#endif
#define LQ_SD_WAVES (LQ_SD_THREADS / 64)
#ifndef LQ_SD_RPT
#define LQ_SD_RPT 8                         // hits per thread and tile of the count / scatter kernels
#endif
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
// own column of `hist` (hist[d * stride + wave]; stride 1, wave 0: one column for the block, when only totals matter) and returns every hit's rank among the wave's hits of its slice so far.
// Hits of one list ascend in rid, so inside a list a slice is one run: its first lane adds the run's length.  Lists are
// taken one after the other (a step rarely touches more than two): no two lanes of one instruction ever add to one counter,
// and the counts a lane sees do not depend on how the hardware orders atomics.
__device__ __forceinline__ u32 sd_rank_step(u32 *hist, u32 wave, u32 d, u32 jl, bool valid, u32 lane, u32 stride)
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
		if (in && rhead) base = atomicAdd(&hist[d * stride + wave], len);
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

// ---- hits per (query, slice, segment) ---------------------------------------------------------------------------------------
// One block per segment.  cnt[cb + slice * nseg + ord] = the segment's hits in that slice.
// The segment's lists sit in LDS as (first hit's number, where the list lies in pos[] minus that number): hit g of the segment
// is pos[la[owner(g)] + g].  A thread first finds the owners of its LQ_SD_RPT hits and issues all their loads, then counts:
// the kernel lives on loads in flight, not on arithmetic.
__global__ void __launch_bounds__(LQ_SD_THREADS)
k_seed_count(SeedIn in, u32 g_lo, u32 *cnt)
{
	__shared__ u32 loff[LQ_SD_SEGL + 1];
	__shared__ u64 la[LQ_SD_SEGL];
	__shared__ u32 hist[LQ_SD_SL_BIG];                       // (shared by the waves: only the totals matter here)
	const SeedSeg sg = in.segs[g_lo + blockIdx.x];
	const SeedQ Q = in.qg[sg.q];
	const u32 t = threadIdx.x, lane = t & 63;
	const u32 nj = (u32)(sg.j1 - sg.j0);
	const u64 H0 = in.h_off[sg.j0];
	const u32 nH = (u32)(in.h_off[sg.j1] - H0);
	for (u32 i = t; i <= nj; i += LQ_SD_THREADS) {
		const u32 o = (u32)(in.h_off[sg.j0 + i] - H0);
		loff[i] = o;
		if (i < nj) la[i] = in.hit_start[sg.j0 + i] - o;
	}
	for (u32 i = t; i < Q.nsl; i += LQ_SD_THREADS) hist[i] = 0;
	__syncthreads();
	for (u32 base = 0; base < nH; base += LQ_SD_TILE) {
		u32 jl[LQ_SD_RPT]; u64 r[LQ_SD_RPT];
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = base + (u32)k * LQ_SD_THREADS + t;
			jl[k] = 0; r[k] = 0;
			if (g < nH) { jl[k] = sd_owner(loff, nj, g); r[k] = in.pos[la[jl[k]] + g]; }
		}
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = base + (u32)k * LQ_SD_THREADS + t;
			if (base + (u32)k * LQ_SD_THREADS + (t & ~63u) < nH)                      // (wave-uniform)
				sd_rank_step(hist, 0, sd_slice((u32)(r[k] >> 32), Q.mul), jl[k], g < nH, lane, 1);
		}
	}
	__syncthreads();
	for (u32 s = t; s < Q.nsl; s += LQ_SD_THREADS) cnt[Q.cb + (u64)s * Q.nseg + sg.ord] = hist[s];
}

// ---- records to their buckets -----------------------------------------------------------------------------------------------
// One block per segment; off = exclusive scan of cnt: where the segment's piece of every bucket starts in `rec`.
template <u32 MAXSL>
__global__ void __launch_bounds__(LQ_SD_THREADS)
k_seed_scatter(SeedIn in, u32 g_lo, const u32 *off, SeedBits bits, u32 span_const /* 0: from qx (-H) */, u64 *rec)
{
	__shared__ u32 loff[LQ_SD_SEGL + 1];
	__shared__ u64 la[LQ_SD_SEGL];
	__shared__ u32 lqy[LQ_SD_SEGL];                          // the minimizers' position << 1 | strand
	__shared__ u32 hist[MAXSL * LQ_SD_WAVES + 1];            // per tile: counts, then first places, per (slice, wave); the last entry: the tile's total
	__shared__ u32 cursor[MAXSL];                            // where the segment's piece of every bucket goes on
	__shared__ u64 sbuf[LQ_SD_TILE];
	__shared__ u32 ws[17];
	const SeedSeg sg = in.segs[g_lo + blockIdx.x];
	const SeedQ Q = in.qg[sg.q];
	const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const u32 nj = (u32)(sg.j1 - sg.j0);
	const u64 H0 = in.h_off[sg.j0];
	const u32 nH = (u32)(in.h_off[sg.j1] - H0);
	const u32 jrel = (u32)(sg.j0 - in.qmoff[sg.q]);          // the segment's first minimizer, counted inside its query
	const i32 ql = (i32)in.qlen[sg.q];
	for (u32 i = t; i <= nj; i += LQ_SD_THREADS) {
		const u32 o = (u32)(in.h_off[sg.j0 + i] - H0);
		loff[i] = o;
		if (i < nj) { la[i] = in.hit_start[sg.j0 + i] - o; lqy[i] = (u32)in.qy[sg.j0 + i]; }
	}
	for (u32 s = t; s < Q.nsl; s += LQ_SD_THREADS) cursor[s] = off[Q.cb + (u64)s * Q.nseg + sg.ord];
	const u32 nE = Q.nsl * LQ_SD_WAVES;                      // entries of hist in use
	const u32 per = (nE + LQ_SD_THREADS - 1) / LQ_SD_THREADS;
	for (u32 base = 0; base < nH; base += LQ_SD_TILE) {
		for (u32 i = t; i < nE; i += LQ_SD_THREADS) hist[i] = 0;
		__syncthreads();
		u64 rc[LQ_SD_RPT]; u32 jl[LQ_SD_RPT], rk[LQ_SD_RPT];
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {                    // owners, then every load in flight
			const u32 g = base + (u32)k * LQ_SD_THREADS + t;
			jl[k] = 0; rc[k] = 0;
			if (g < nH) { jl[k] = sd_owner(loff, nj, g); rc[k] = in.pos[la[jl[k]] + g]; }
		}
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = base + (u32)k * LQ_SD_THREADS + t;
			const bool valid = g < nH;
			u32 d = 0;
			if (valid) {
				const u64 r = rc[k];
				const u32 qyv = lqy[jl[k]], qpos = qyv >> 1;
				const u32 span = span_const ? span_const : (u32)(in.qx[sg.j0 + jl[k]] & 0xff);
				const u32 rid = (u32)(r >> 32), rpos = (u32)r >> 1, rs = ((u32)r & 1u) ^ (qyv & 1u);
				const i32 ypos = rs ? ql - (i32)(qpos + 1 - span) - 1 : (i32)qpos;     // the anchor's query coordinate (lqmap.c:191-197)
				const u32 diag = (u32)((i32)rpos - ypos + ql + 256);                     // (never negative: ypos <= qlen)
				d = sd_slice(rid, Q.mul);
				rc[k] = (u64)rid << (bits.jb + bits.db + 1) | (u64)rs << (bits.jb + bits.db) | (u64)diag << bits.jb | (u64)(jrel + jl[k]);
			}
			rk[k] = 0;
			if (base + (u32)k * LQ_SD_THREADS + (t & ~63u) < nH) rk[k] = sd_rank_step(hist, wave, d, jl[k], valid, lane, LQ_SD_WAVES);
			jl[k] = d;                                             // (from here on: the hit's slice)
		}
		__syncthreads();
		// counts -> first places, in (slice, wave) order: a thread sums a stretch of `per` entries, the block scans the sums
		u32 mine = 0;
		for (u32 i = 0; i < per; ++i) { const u32 e = t * per + i; if (e < nE) mine += hist[e]; }
		u32 total = 0;
		u32 run = sd_block_exscan(mine, ws, &total);
		for (u32 i = 0; i < per; ++i) {
			const u32 e = t * per + i;
			if (e < nE) { const u32 c = hist[e]; hist[e] = run; run += c; }
		}
		if (t == 0) hist[nE] = total;
		__syncthreads();
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = base + (u32)k * LQ_SD_THREADS + t;
			if (g < nH) sbuf[hist[jl[k] * LQ_SD_WAVES + wave] + rk[k]] = rc[k];
		}
		__syncthreads();
		for (u32 i = t; i < total; i += LQ_SD_THREADS) {
			const u64 r = sbuf[i];
			const u32 d = sd_slice(sd_rid(r, bits), Q.mul);
			rec[cursor[d] + (i - hist[d * LQ_SD_WAVES])] = r;        // (a slice's first place in the tile: that of its wave 0)
		}
		__syncthreads();
		for (u32 s = t; s < Q.nsl; s += LQ_SD_THREADS) cursor[s] += hist[(s + 1) * LQ_SD_WAVES] - hist[s * LQ_SD_WAVES];
		__syncthreads();                                          // (the next tile clears hist)
	}
}

// ---- which records survive --------------------------------------------------------------------------------------------------
struct SeedDecide {
	u32 n_min;                              // hits a component needs (run_n_min; >= 2 here: without a filter nothing is bucketed)
	u32 dshift;                             // log2 of the bin width, D > bw
	u32 pair_bits;                          // pair counters in use (a power of two, at most 2^LQ_SD_PAIR_BITS; tests shrink it: pairs alias)
	u32 big_pair;                           // a pair with that many hits is kept without looking at its diagonals (at most LQ_SD_BIG_PAIR)
	int no_self;
	unsigned long long *stats;              // LQCOV_SEED_STATS: {records, records whose pair holds n_min, survivors, buckets beyond the LDS path} summed; else null
};
#ifndef LQ_SD_DTHREADS
#define LQ_SD_DTHREADS 1024
#endif
#define LQ_SD_DRPT 8                        // records a thread holds: a bucket of up to 8192 records is read once
#define LQ_SD_DCAP (LQ_SD_DTHREADS * LQ_SD_DRPT)
#define LQ_SD_PAIR_BITS 13                  // 8192 pair counters (16 bits each)
#define LQ_SD_NPAIR (1u << LQ_SD_PAIR_BITS)
#define LQ_SD_BIG_PAIR 32u
#define LQ_SD_KEY_PASS (1u << 31)           // key bits: the record's pair holds n_min hits / the record survives
#define LQ_SD_KEY_LIVE (1u << 30)

// a bucket (query, slice) of the chunk: records [b0, b0 + n) of the record buffer; rid0: the slice's first target
struct alignas(16) SeedBk { u32 b0, n, q, rid0; };
struct SeedDecIn {
	const SeedQ *qg; const u32 *bq;         // bq: first bucket of every query of the chunk (n_qc + 1 entries), q_lo: its first query
	u32 q_lo, n_qc;
	const u32 *off;                         // piece table (exclusive scan of cnt), + 1 sentinel
	const u64 *qx, *qy, *qmoff; const u32 *qlen;
	const u32 *self_off, *self_rid; AvaView ava;
};
__global__ void k_seed_bdesc(SeedDecIn in, u32 n_bk, SeedBk *bd)
{
	const u32 bk = blockIdx.x * blockDim.x + threadIdx.x;
	if (bk >= n_bk) return;
	const u32 qi = lq_find_seg32(in.bq, in.n_qc, bk);
	const u32 q = in.q_lo + qi, s = bk - in.bq[qi];
	const SeedQ Q = in.qg[q];
	const u64 e0 = Q.cb + (u64)s * Q.nseg;
	SeedBk b; b.b0 = in.off[e0]; b.n = in.off[e0 + Q.nseg] - b.b0; b.q = q; b.rid0 = sd_slice_first(s, Q.mul);   // (the entry after a query's last one is the next query's first, or the sentinel)
	bd[bk] = b;
}

// 16-bit values, two to a word
__device__ __forceinline__ u32 sd_h16_get(const u32 *tab, u32 p) { return tab[p >> 1] >> ((p & 1u) << 4) & 0xffffu; }
// (both return the old value; a value never leaves its 16 bits here: sums are bounded by the bucket, and nothing is taken from a zero)
__device__ __forceinline__ u32 sd_h16_add(u32 *tab, u32 p, u32 v) { const u32 sh = (p & 1u) << 4; return atomicAdd(&tab[p >> 1], v << sh) >> sh & 0xffffu; }
__device__ __forceinline__ u32 sd_h16_dec(u32 *tab, u32 p) { const u32 sh = (p & 1u) << 4; return atomicSub(&tab[p >> 1], 1u << sh) >> sh & 0xffffu; }

// does the record whose pair's diagonal bins are ent[st .. en) and whose own bin is `mine` lie in a gap-free stretch of
// non-empty bins that holds n_min hits?  Bins are compared modulo 2^16 (a wrap only adds).  The window is seven bins: a
// stretch that reaches its edge is taken as long enough (exact for n_min <= 4, the presets'; generous beyond).
__device__ __forceinline__ bool sd_stretch(const u16 *ent, u32 st, u32 en, u32 mine, u32 n_min)
{
	u64 occ = 0;                                              // hits per bin mine - 3 .. mine + 3, a byte each (fewer than LQ_SD_BIG_PAIR <= 255 entries)
	for (u32 e = st; e < en; ++e) {
		const i32 d = (i32)(int16_t)(u16)(ent[e] - (u16)mine) + 3;
		if ((u32)d <= 6u) occ += 1ULL << (d << 3);
	}
	const u32 side = n_min - 1;
	u32 tot = (u32)(occ >> 24) & 0xffu;                       // (the record itself is one of the entries)
	if (tot >= n_min) return true;
	for (u32 k = 1; k <= 3; ++k) { const u32 c = (u32)(occ >> ((3 + k) << 3)) & 0xffu; if (c == 0) break; tot += c; if (k == side || k == 3) return true; }
	for (u32 k = 1; k <= 3; ++k) { const u32 c = (u32)(occ >> ((3 - k) << 3)) & 0xffu; if (c == 0) break; tot += c; if (k == side || k == 3) return true; }
	return tot >= n_min;
}

// the self diagonal and -X (lqmap.c:180-187) for a record of query q that would otherwise survive
__device__ __forceinline__ bool sd_rare_drop(u64 r, u32 q, const SeedDecIn &in, SeedBits bits, u32 span_const, bool self_q)
{
	const u32 rid = sd_rid(r, bits);
	if (self_q) {
		const i32 ql = (i32)in.qlen[q];
		const u64 j = in.qmoff[q] + sd_jl(r, bits);
		const u32 qpos = (u32)in.qy[j] >> 1, span = span_const ? span_const : (u32)(in.qx[j] & 0xff);
		const i32 ypos = sd_rs(r, bits) ? ql - (i32)(qpos + 1 - span) - 1 : (i32)qpos;
		const u32 rpos = (u32)((i32)sd_diag(r, bits) + ypos - ql - 256);
		if (rpos == qpos && lq_is_self(in.self_off, in.self_rid, q, rid)) return true;
	}
	return in.ava.t_rank && in.ava.t_rank[rid] < in.ava.q_lo[q];
}

// One block per bucket.  The records are read once and stay in registers for the final write; what the phases in between need
// of a record -- its pair and its diagonal bin -- sits in LDS as a 32-bit key, so those phases are short rolled loops:
//   pairs   hits per (target, relative strand), 16-bit counters
//   ends    inclusive scan of the counts of the pairs that hold n_min hits: where each such pair's bins end in `ent`
//   group   every record of such a pair drops its bin into its pair's stretch of `ent` (filled from the end: afterwards the
//           pair's stretch is [ends[p], ends[p + 1]))
//   decide  a record looks at its own pair's bins only: no other pair can add to them (sd_stretch)
//   write   survivors to the front of the bucket, ordered (thread, k); scnt[bucket] = how many
// A bucket beyond LQ_SD_DCAP records (a query with more hits than slices can divide): pairs only, the records read twice.
__global__ void __launch_bounds__(LQ_SD_DTHREADS)
k_seed_decide(SeedDecIn in, const SeedBk *bd, SeedDecide dp, SeedBits bits, u32 span_const, u64 *rec, u32 *scnt)
{
	__shared__ u32 key[LQ_SD_DCAP];
	__shared__ u32 ends[LQ_SD_NPAIR / 2 + 1];                // 16-bit halves; entry NPAIR: the total
	__shared__ u16 ent[LQ_SD_DCAP];
	__shared__ u32 ws[17];
	const u32 t = threadIdx.x;
	const u32 bk = blockIdx.x;
	const SeedBk B = bd[bk];
	const u32 n = B.n, q = B.q;
	if (n == 0) { if (t == 0) scnt[bk] = 0; return; }
	const u32 pmask = (1u << dp.pair_bits) - 1u;
	u64 *R = rec + B.b0;
	const bool self_q = dp.no_self && in.self_off[q] != in.self_off[q + 1];
	const bool rare = self_q || in.ava.t_rank != nullptr;
	for (u32 i = t; i < LQ_SD_NPAIR / 2 + 1; i += LQ_SD_DTHREADS) ends[i] = 0;
	if (n > LQ_SD_DCAP) {
		// ---- beyond what the block holds: pairs only ----
		__syncthreads();
		for (u32 i = t; i < n; i += LQ_SD_DTHREADS) {
			const u64 r = R[i];
			const u32 p = (((sd_rid(r, bits) - B.rid0) << 1) | sd_rs(r, bits)) & pmask;
			if (sd_h16_get(ends, p) < 0x8000u) sd_h16_add(ends, p, 1);           // (at most blockDim more adds can slip past the test: no carry)
		}
		__syncthreads();
		u32 done = 0;                                             // survivors written so far (block-uniform)
		for (u32 base = 0; base < n; base += LQ_SD_DTHREADS) {
			const u32 i = base + t;
			const u64 r = i < n ? R[i] : 0;
			bool a = i < n && sd_h16_get(ends, (((sd_rid(r, bits) - B.rid0) << 1) | sd_rs(r, bits)) & pmask) >= dp.n_min;
			if (a && rare && sd_rare_drop(r, q, in, bits, span_const, self_q)) a = false;
			u32 total = 0;
			const u32 at = sd_block_exscan(a ? 1u : 0u, ws, &total);  // (its barriers stand between this round's reads and writes)
			if (a) R[done + at] = r;                                  // done + at <= i: never ahead of what is still to be read
			done += total;
		}
		if (t == 0) { scnt[bk] = done; if (dp.stats) { atomicAdd(&dp.stats[0], (unsigned long long)n); atomicAdd(&dp.stats[1], (unsigned long long)done); atomicAdd(&dp.stats[2], (unsigned long long)done); atomicAdd(&dp.stats[3], 1ULL); } }
		return;
	}
	u64 rc[LQ_SD_DRPT];
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) { const u32 i = (u32)k * LQ_SD_DTHREADS + t; rc[k] = i < n ? R[i] : 0; }
	__syncthreads();                                          // (ends is clear)
	// pairs
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) {
		const u32 i = (u32)k * LQ_SD_DTHREADS + t;
		if (i < n) {
			const u32 p = (((sd_rid(rc[k], bits) - B.rid0) << 1) | sd_rs(rc[k], bits)) & pmask;
			key[i] = p | ((sd_diag(rc[k], bits) >> dp.dshift) & 0xffffu) << LQ_SD_PAIR_BITS;
			sd_h16_add(ends, p, 1);
		}
	}
	__syncthreads();
	// ends: a thread takes NPAIR / THREADS pairs in a row
	{
		constexpr u32 PER = LQ_SD_NPAIR / LQ_SD_DTHREADS;         // 8 (even: whole words)
		u32 c[PER], mine = 0;
#pragma unroll
		for (u32 i = 0; i < PER; i += 2) {
			const u32 w = ends[(t * PER + i) >> 1];
			c[i] = (w & 0xffffu) >= dp.n_min ? (w & 0xffffu) : 0u; c[i + 1] = (w >> 16) >= dp.n_min ? (w >> 16) : 0u;
			mine += c[i] + c[i + 1];
		}
		u32 total = 0;
		u32 run = sd_block_exscan(mine, ws, &total);            // (its first barrier: every count is read before any end is written)
#pragma unroll
		for (u32 i = 0; i < PER; i += 2) { const u32 e0 = run + c[i], e1 = e0 + c[i + 1]; ends[(t * PER + i) >> 1] = e0 | e1 << 16; run = e1; }
		if (t == 0) { ends[LQ_SD_NPAIR / 2] = total; if (dp.stats) { atomicAdd(&dp.stats[0], (unsigned long long)n); atomicAdd(&dp.stats[1], (unsigned long long)total); } }
	}
	__syncthreads();
	// which records belong to a pair that holds enough (the ends still stand)
	for (u32 i = t; i < n; i += LQ_SD_DTHREADS) {
		const u32 kk = key[i], p = kk & (LQ_SD_NPAIR - 1u);
		if (sd_h16_get(ends, p) != (p ? sd_h16_get(ends, p - 1) : 0u)) key[i] = kk | LQ_SD_KEY_PASS;
	}
	__syncthreads();
	// group: a pair's stretch fills from its end downwards
	for (u32 i = t; i < n; i += LQ_SD_DTHREADS) {
		const u32 kk = key[i];
		if (kk & LQ_SD_KEY_PASS) ent[sd_h16_dec(ends, kk & (LQ_SD_NPAIR - 1u)) - 1u] = (u16)(kk >> LQ_SD_PAIR_BITS);
	}
	__syncthreads();
	// decide
	for (u32 i = t; i < n; i += LQ_SD_DTHREADS) {
		const u32 kk = key[i];
		if (!(kk & LQ_SD_KEY_PASS)) continue;
		const u32 p = kk & (LQ_SD_NPAIR - 1u);
		const u32 st = sd_h16_get(ends, p), en = sd_h16_get(ends, p + 1);
		bool a = en - st >= dp.big_pair || sd_stretch(ent, st, en, kk >> LQ_SD_PAIR_BITS & 0xffffu, dp.n_min);
		if (a && rare) a = !sd_rare_drop(R[i], q, in, bits, span_const, self_q);   // (the bucket is still as it was read: nothing is written before the last phase)
		if (a) key[i] = kk | LQ_SD_KEY_LIVE;
	}
	__syncthreads();
	// write
	u32 al = 0, mine = 0;
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) {
		const u32 i = (u32)k * LQ_SD_DTHREADS + t;
		if (i < n && (key[i] & LQ_SD_KEY_LIVE)) { al |= 1u << k; ++mine; }
	}
	u32 total = 0;
	u32 at = sd_block_exscan(mine, ws, &total);                // (every record was read long before)
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) if (al >> k & 1u) R[at++] = rc[k];
	if (t == 0) { scnt[bk] = total; if (dp.stats) atomicAdd(&dp.stats[2], (unsigned long long)total); }
}

// survivors of every bucket of the chunk, dense: surv[base + soff[bucket] ...] (soff = exclusive scan of scnt); and where
// every query's survivors start (aqf_off[q], for the chunk's queries)
__global__ void k_seed_collect(const SeedBk *bd, const u64 *rec, const u32 *scnt, const u32 *soff, u64 base, u64 *surv, u64 *aqf_off)
{
	const u32 bk = blockIdx.x;
	const SeedBk B = bd[bk];
	if (B.rid0 == 0 && threadIdx.x == 0) aqf_off[B.q] = base + soff[bk];     // (a query's first slice starts at target 0, and only that one)
	const u32 n = scnt[bk];
	if (n == 0) return;
	const u64 *R = rec + B.b0;
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
