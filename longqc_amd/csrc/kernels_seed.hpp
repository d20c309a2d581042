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
//   k_seed_decide   per bucket: the records in registers, 16-bit pair counters in LDS; every pair that holds n_min hits gets a
//                   histogram of 4-bit diagonal bins as long as its diagonals can be, private to the pair (nothing aliases;
//                   bins wider than the band); a record counts itself in, then reads the bins around its own; survivors
//                   compacted in place.  Buckets beyond a block's registers: k_seed_decide_big, in passes over their targets
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
#define LQ_SD_SL_SMALL 512u                 // slices per query: the two shapes of the scatter kernel
#define LQ_SD_SL_BIG 2048u

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

// The owners of 64 consecutive hits, from the owner `ow` of a hit at or before the first of them (wave-uniform): a lane steps over
// the list ends between that hit and its own -- a step of 64 hits rarely touches more than two lists, where the binary search
// above is eight dependent LDS reads per hit (round 6: ~40 % of k_seed_count's instructions).  loff[n] = the segment's hits > g.
__device__ __forceinline__ u32 sd_owner_from(const u32 *off, u32 ow, u32 g, bool valid)
{
	u32 o = ow;
	if (valid) while (off[o + 1] <= g) ++o;
	return o;
}
#define LQ_SD_WAVE_HITS (LQ_SD_RPT * 64u)                   // a wave's hits of a tile: consecutive, 64 per step

// One step of a wave over 64 consecutive hits of the segment (valid lanes are a prefix): counts them per slice in the wave's
// own 16-bit counter (two waves to a word: hist[d * LQ_SD_WAVES / 2 + wave / 2]; a tile holds fewer than 65536 hits) and returns
// every hit's rank among the wave's hits of its slice so far.  Hits of one list ascend in rid, so inside a list a slice is one
// run: its first lane adds the run's length.  Lists are taken one after the other (a step rarely touches more than two): no two
// lanes of one instruction ever add to one counter, and the counts a lane sees do not depend on how the hardware orders atomics.
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
	const u32 sh = (wave & 1u) << 4;
	u32 *hw = hist + d * (LQ_SD_WAVES / 2) + (wave >> 1);
	u32 rank = 0;
	while (Lm) {                                              // (wave-uniform)
		const u32 lo = (u32)__ffsll((unsigned long long)Lm) - 1u;
		Lm &= Lm - 1;
		const u32 hi = Lm ? (u32)__ffsll((unsigned long long)Lm) - 1u : 64u;
		const bool in = valid && lane >= lo && lane < hi;
		u32 base = 0;
		if (in && rhead) base = atomicAdd(hw, len << sh) >> sh & 0xffffu;
		const u32 b = __shfl(base, (int)hp);
		if (in) rank = b + (lane - hp);
	}
	return rank;
}

// The same step when only the totals per slice matter (one counter per slice for the whole block): the first lane of a run
// adds what is left of the step from there on and takes the same amount back from the slice of the run before it -- a run of
// lanes [h, e) ends up with e - h, without anybody looking for e.
__device__ __forceinline__ void sd_count_step(u32 *hist, u32 d, bool valid, u32 lane)
{
	const u32 dp = __shfl_up(d, 1);
	const u32 left = (u32)__popcll(__ballot(valid)) - lane;
	if (valid && (lane == 0 || d != dp)) {
		atomicAdd(&hist[d], left);
		if (lane) atomicSub(&hist[dp], left);
	}
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
	const u32 wave = t >> 6;
	for (u32 base = 0; base < nH; base += LQ_SD_TILE) {
		u64 r[LQ_SD_RPT];
		const u32 gw = base + wave * LQ_SD_WAVE_HITS;            // the wave's first hit of the tile
		u32 ow = sd_owner(loff, nj, gw < nH ? gw : nH - 1u);     // (wave-uniform)
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = gw + (u32)k * 64u + lane;
			const u32 o = sd_owner_from(loff, ow, g, g < nH);
			r[k] = 0;
			if (g < nH) r[k] = in.pos[la[o] + g];
			ow = (u32)__shfl((int)o, 63);
		}
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = gw + (u32)k * 64u + lane;
			if (gw + (u32)k * 64u < nH)                              // (wave-uniform)
				sd_count_step(hist, sd_slice((u32)(r[k] >> 32), Q.mul), g < nH, lane);
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
	__shared__ u32 hist[MAXSL * (LQ_SD_WAVES / 2) + 1];      // per tile: counts, then first places, per (slice, wave), 16 bits each; the last word: the tile's total
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
	const u32 nE = Q.nsl * (LQ_SD_WAVES / 2);                // words of hist in use
	const u32 per = (nE + LQ_SD_THREADS - 1) / LQ_SD_THREADS;
	for (u32 base = 0; base < nH; base += LQ_SD_TILE) {
		for (u32 i = t; i < nE; i += LQ_SD_THREADS) hist[i] = 0;
		__syncthreads();
		u64 rc[LQ_SD_RPT]; u32 jl[LQ_SD_RPT], rk[LQ_SD_RPT];
		const u32 gw = base + wave * LQ_SD_WAVE_HITS;            // the wave's first hit of the tile: a wave takes consecutive hits, 64 per step
		u32 ow = sd_owner(loff, nj, gw < nH ? gw : nH - 1u);     // (wave-uniform)
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {                    // owners, then every load in flight
			const u32 g = gw + (u32)k * 64u + lane;
			jl[k] = sd_owner_from(loff, ow, g, g < nH); rc[k] = 0;
			if (g < nH) rc[k] = in.pos[la[jl[k]] + g];
			ow = (u32)__shfl((int)jl[k], 63);
		}
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = gw + (u32)k * 64u + lane;
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
			if (gw + (u32)k * 64u < nH) rk[k] = sd_rank_step(hist, wave, d, jl[k], valid, lane);
			jl[k] = d;                                             // (from here on: the hit's slice)
		}
		__syncthreads();
		// counts -> first places, in (slice, wave) order: a thread sums a stretch of `per` words (two counts each, the even wave's
		// in the low half), the block scans the sums
		u32 mine = 0;
		for (u32 i = 0; i < per; ++i) { const u32 e = t * per + i; if (e < nE) { const u32 c = hist[e]; mine += (c & 0xffffu) + (c >> 16); } }
		u32 total = 0;
		u32 run = sd_block_exscan(mine, ws, &total);
		for (u32 i = 0; i < per; ++i) {
			const u32 e = t * per + i;
			if (e < nE) { const u32 c = hist[e]; const u32 mid = run + (c & 0xffffu); hist[e] = run | mid << 16; run = mid + (c >> 16); }
		}
		if (t == 0) hist[nE] = total;
		__syncthreads();
#pragma unroll
		for (int k = 0; k < LQ_SD_RPT; ++k) {
			const u32 g = gw + (u32)k * 64u + lane;
			if (g < nH) sbuf[(hist[jl[k] * (LQ_SD_WAVES / 2) + (wave >> 1)] >> ((wave & 1u) << 4) & 0xffffu) + rk[k]] = rc[k];
		}
		__syncthreads();
		for (u32 i = t; i < total; i += LQ_SD_THREADS) {
			const u64 r = sbuf[i];
			const u32 d = sd_slice(sd_rid(r, bits), Q.mul);
			rec[cursor[d] + (i - (hist[d * (LQ_SD_WAVES / 2)] & 0xffffu))] = r;      // (a slice's first place in the tile: that of its wave 0)
		}
		__syncthreads();
		for (u32 s = t; s < Q.nsl; s += LQ_SD_THREADS) cursor[s] += (hist[(s + 1) * (LQ_SD_WAVES / 2)] & 0xffffu) - (hist[s * (LQ_SD_WAVES / 2)] & 0xffffu);   // (after the last slice: the word that holds the total)
		__syncthreads();                                          // (the next tile clears hist)
	}
}

// ---- which records survive --------------------------------------------------------------------------------------------------
struct SeedDecide {
	u32 n_min;                              // hits a component needs (run_n_min; 2 .. 15 here: else nothing is bucketed)
	u32 dshift;                             // log2 of the bin width, D > bw
	u32 pair_bits;                          // pair counters in use (a power of two, at most 2^LQ_SD_PAIR_BITS; tests shrink it: pairs alias)
	u32 max_tlen;                           // longest target of the part (the bins of pairs that share a counter)
	u32 hwords;                             // histogram space in use, in words of eight bins (at most LQ_SD_HWORDS; tests shrink it)
	u32 dcap, bigcap;                       // records of a bucket decided from registers / in passes (at most LQ_SD_DCAP / LQ_SD_BIGCAP; tests shrink them)
	int no_self;
	unsigned long long *stats;              // LQCOV_SEED_STATS: {records, records whose pair holds n_min, survivors, buckets beyond the block, records of pairs left without a histogram} summed; else null
};
#ifndef LQ_SD_DTHREADS
#define LQ_SD_DTHREADS 1024
#endif
#define LQ_SD_DRPT 8                        // records a thread holds: a bucket of up to 8192 records is read once
#define LQ_SD_DCAP (LQ_SD_DTHREADS * LQ_SD_DRPT)
#define LQ_SD_PAIR_BITS 13                  // 8192 pair counters (16 bits each)
#define LQ_SD_NPAIR (1u << LQ_SD_PAIR_BITS)
#define LQ_SD_HWORDS 12288u                 // histogram space of a bucket: words of eight 4-bit bins (48 KiB)
#define LQ_SD_HBINS_MAX 8192u               // a pair whose diagonals take more bins than that (reads of megabases) is kept as it is

// a bucket (query, slice) of the chunk: records [b0, b0 + n) of the record buffer; rid0: the slice's first target
struct alignas(32) SeedBk { u32 b0, n, q, rid0, nt, pad0, pad1, pad2; };   // nt: targets in the slice
struct SeedDecIn {
	const SeedQ *qg; const u32 *bq;         // bq: first bucket of every query of the chunk (n_qc + 1 entries), q_lo: its first query
	u32 q_lo, n_qc;
	const u32 *off;                         // piece table (exclusive scan of cnt), + 1 sentinel
	const u64 *qx, *qy, *qmoff; const u32 *qlen;
	const u32 *self_off, *self_rid; AvaView ava;
	const u32 *tlen; u32 n_targets;         // the part's target lengths
};
// (buckets of more than dcap records are listed for k_seed_decide_big: big[0] = how many, big[1 ..] = which)
__global__ void k_seed_bdesc(SeedDecIn in, u32 n_bk, u32 dcap, SeedBk *bd, u32 *big)
{
	const u32 bk = blockIdx.x * blockDim.x + threadIdx.x;
	if (bk >= n_bk) return;
	const u32 qi = lq_find_seg32(in.bq, in.n_qc, bk);
	const u32 q = in.q_lo + qi, s = bk - in.bq[qi];
	const SeedQ Q = in.qg[q];
	const u64 e0 = Q.cb + (u64)s * Q.nseg;
	SeedBk b; b.b0 = in.off[e0]; b.n = in.off[e0 + Q.nseg] - b.b0; b.q = q; b.rid0 = sd_slice_first(s, Q.mul);   // (the entry after a query's last one is the next query's first, or the sentinel)
	b.nt = (s + 1 < Q.nsl ? sd_slice_first(s + 1, Q.mul) : in.n_targets) - b.rid0; b.pad0 = b.pad1 = b.pad2 = 0;
	bd[bk] = b;
	if (b.n > dcap) big[1u + atomicAdd(&big[0], 1u)] = bk;
}

// 16-bit values, two to a word
__device__ __forceinline__ u32 sd_h16_get(const u32 *tab, u32 p) { return tab[p >> 1] >> ((p & 1u) << 4) & 0xffffu; }
// (returns the old value; a value never leaves its 16 bits here: sums are bounded by the bucket)
__device__ __forceinline__ u32 sd_h16_add(u32 *tab, u32 p, u32 v) { const u32 sh = (p & 1u) << 4; return atomicAdd(&tab[p >> 1], v << sh) >> sh & 0xffffu; }

// Hits per diagonal bin of one pair: 4 bits a bin, eight bins to a word, one bin per D of the pair's diagonals -- a record's
// diagonal (target position - query coordinate + query length + 256) lies in [0, query + target length + 256), so a pair needs
// (that >> dshift) + 1 bins, three more on either side for the windows below, and nothing ever wraps.  A bin counts to 15 and
// stays there: every n_min in use is below that, so "15" is as good as the number.
__device__ __forceinline__ void sd_bin_inc(u32 *w, u32 field)
{
	const u32 sh = field << 2;
	u32 old = *w;
	for (;;) {
		if ((old >> sh & 15u) == 15u) return;
		const u32 seen = atomicCAS(w, old, old + (1u << sh));
		if (seen == old) return;
		old = seen;
	}
}
__device__ __forceinline__ u32 sd_hist_words(u32 nb) { return (nb + 21u) >> 3; }   // bins 0 .. nb - 1 stand at places 3 .. nb + 2; the two-word window of the last one ends inside
// Does the gap-free stretch of non-empty bins around bin `dbin` hold n_min hits?  The seven bins dbin - 3 .. dbin + 3 (places
// dbin .. dbin + 6) are cut out of two neighbouring words; a stretch that reaches the window's edge is taken as long enough
// (exact for n_min <= 4, the presets'; generous beyond).
__device__ __forceinline__ bool sd_window_alive(const u32 *h, u32 dbin, u32 n_min)
{
	const u32 w = dbin >> 3;
	const u64 two = (u64)h[w + 1u] << 32 | h[w];
	const u32 g = (u32)(two >> ((dbin & 7u) << 2));            // fields 0 .. 6 = bins dbin - 3 .. dbin + 3
	const u32 own = g >> 12 & 15u;
	const u32 r1 = g >> 16 & 15u, r2 = r1 ? g >> 20 & 15u : 0u, r3 = r2 ? g >> 24 & 15u : 0u;
	const u32 l1 = g >> 8 & 15u, l2 = l1 ? g >> 4 & 15u : 0u, l3 = l2 ? g & 15u : 0u;
	const u32 side = n_min - 1u;
	const u32 reach = side <= 1u ? (r1 | l1) : side == 2u ? (r2 | l2) : (r3 | l3);      // n_min bins in a row (or the edge of the window)
	const u32 tot = own + r1 + l1 + (side > 1u ? r2 + l2 : 0u) + (side > 2u ? r3 + l3 : 0u);
	return reach != 0u || tot >= n_min;
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

// The pairs that hold n_min hits get a histogram of their diagonal bins each, as long as the pair's diagonals can be (query +
// target length), handed out by a scan over the pair counters, eight to a thread; pairs that find no room (or whose diagonals
// take more than LQ_SD_HBINS_MAX bins) are kept as they are.  ends[]: counts in, 0 (too few) / 0xffff (kept as it is) / the
// histogram's first word + 1 out.  rid_base: the target of pair 0; shared: a counter may stand for several targets (more
// pairs than counters in use).  nbmax: bins of the longest pair of the part (a record's bin is never beyond that).
__device__ __forceinline__ void sd_rank_pairs(u32 *ends, u32 *ws, const SeedDecIn &in, const SeedDecide &dp, u32 ql, u32 rid_base, bool shared, u32 t)
{
	constexpr u32 PER = LQ_SD_NPAIR / LQ_SD_DTHREADS;             // 8 (even: whole words)
	u32 c[PER], mine = 0, held = 0;
#pragma unroll
	for (u32 i = 0; i < PER; i += 2) { const u32 w = ends[(t * PER + i) >> 1]; c[i] = w & 0xffffu; c[i + 1] = w >> 16; }
#pragma unroll
	for (u32 i = 0; i < PER; ++i) {
		if (c[i] >= dp.n_min) {
			held += c[i];
			const u32 rid = rid_base + ((t * PER + i) >> 1);
			const u32 tl = shared || rid >= in.n_targets ? dp.max_tlen : in.tlen[rid];
			const u32 nb = ((ql + tl + 256u) >> dp.dshift) + 1u;  // bins its diagonals can take
			c[i] = nb <= LQ_SD_HBINS_MAX ? sd_hist_words(nb) : 0xffffu;
			if (c[i] != 0xffffu) mine += c[i];
		} else c[i] = 0;
	}
	u32 total = 0;
	u32 at = sd_block_exscan(mine, ws, &total);                 // (its first barrier: every count is read before any number is written)
#pragma unroll
	for (u32 i = 0; i < PER; ++i) {
		u32 v = c[i];
		if (v && v != 0xffffu) {
			const u32 nw = v;
			if (at + nw <= dp.hwords) v = at + 1u; else { v = 0xffffu; if (dp.stats) atomicAdd(&dp.stats[4], 1ULL); }
			at += nw;
		}
		c[i] = v;
	}
#pragma unroll
	for (u32 i = 0; i < PER; i += 2) ends[(t * PER + i) >> 1] = c[i] | c[i + 1] << 16;
	if (dp.stats && held) atomicAdd(&dp.stats[1], (unsigned long long)held);
}
// a record of a pair numbered o (see above) and diagonal bin dbin: count it / decide it
__device__ __forceinline__ void sd_count_bin(u32 *hist, u32 o, u32 dbin, u32 hwords)
{
	if (o && o != 0xffffu) { const u32 place = dbin + 3u, w = o - 1u + (place >> 3); sd_bin_inc(&hist[w < hwords ? w : hwords - 1u], place & 7u); }   // (the clamp never bites: a diagonal is below query + target length + 256)
}
__device__ __forceinline__ bool sd_decide_bin(const u32 *hist, u32 o, u32 dbin, u32 n_min)
{
	return o == 0xffffu || (o && sd_window_alive(hist + (o - 1u), dbin, n_min));
}

#define LQ_SD_BIGCAP 65536u                 // records of a bucket that is decided in passes (a bit each in LDS)

// One block per bucket.  The records are read once and stay in registers (eight per thread):
//   pairs   hits per (target, relative strand), 16-bit counters
//   rank    sd_rank_pairs
//   bins    every record of a pair with a histogram counts itself there: no other pair can add to it
//   decide  sd_window_alive; the self diagonal and -X (lqmap.c:180-187) for the queries that have any
//   write   survivors to the front of the bucket, ordered (thread, k); scnt[bucket] = how many
// (a bucket beyond LQ_SD_DCAP records -- a query with more hits than slices can divide --: k_seed_decide_big)
__global__ void __launch_bounds__(LQ_SD_DTHREADS, 8)          // (eight waves per SIMD: two blocks on a CU)
k_seed_decide(SeedDecIn in, const SeedBk *bd, SeedDecide dp, SeedBits bits, u32 span_const, u64 *rec, u32 *scnt)
{
	__shared__ u32 ends[LQ_SD_NPAIR / 2];                    // 16-bit halves (sd_rank_pairs)
	__shared__ u32 hist[LQ_SD_HWORDS + 1];
	__shared__ u32 ws[17];
	const u32 t = threadIdx.x;
	const u32 bk = blockIdx.x;
	const SeedBk B = bd[bk];
	const u32 n = B.n, q = B.q;
	if (n == 0) { if (t == 0) scnt[bk] = 0; return; }
	const u32 pmask = (1u << dp.pair_bits) - 1u;
	const u32 sh_p = bits.jb + bits.db, sh_d = bits.jb + dp.dshift, p0 = B.rid0 << 1;   // (rid << 1 | strand) = record >> sh_p
	const u32 dmask = (1u << (bits.db > dp.dshift ? bits.db - dp.dshift : 0u)) - 1u;     // the diagonal's bin: record >> sh_d & dmask
	u64 *R = rec + B.b0;
	const bool self_q = dp.no_self && in.self_off[q] != in.self_off[q + 1];
	const bool rare = self_q || in.ava.t_rank != nullptr;
	const u32 ql = in.qlen[q];
	if (n > dp.dcap) return;                                  // (k_seed_decide_big's)
	for (u32 i = t; i < LQ_SD_NPAIR / 2; i += LQ_SD_DTHREADS) ends[i] = 0;
	for (u32 i = t; i < dp.hwords + 1u; i += LQ_SD_DTHREADS) hist[i] = 0;
	u64 rc[LQ_SD_DRPT];
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) { const u32 i = (u32)k * LQ_SD_DTHREADS + t; rc[k] = i < n ? R[i] : 0; }
	__syncthreads();                                          // (ends and hist are clear)
	// pairs
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) if ((u32)k * LQ_SD_DTHREADS + t < n) sd_h16_add(ends, ((u32)(rc[k] >> sh_p) - p0) & pmask, 1);
	__syncthreads();
	sd_rank_pairs(ends, ws, in, dp, ql, B.rid0, B.nt * 2u > pmask + 1u, t);
	if (dp.stats && t == 0) atomicAdd(&dp.stats[0], (unsigned long long)n);
	__syncthreads();
	// bins
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) if ((u32)k * LQ_SD_DTHREADS + t < n) sd_count_bin(hist, sd_h16_get(ends, ((u32)(rc[k] >> sh_p) - p0) & pmask), (u32)(rc[k] >> sh_d) & dmask, dp.hwords);
	__syncthreads();
	// decide
	u32 al = 0;
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) if ((u32)k * LQ_SD_DTHREADS + t < n && sd_decide_bin(hist, sd_h16_get(ends, ((u32)(rc[k] >> sh_p) - p0) & pmask), (u32)(rc[k] >> sh_d) & dmask, dp.n_min)) al |= 1u << k;
	if (rare) for (u32 k = 0; k < LQ_SD_DRPT; ++k)                // (rolled, the records read again: rare)
		if ((al >> k & 1u) && sd_rare_drop(R[k * LQ_SD_DTHREADS + t], q, in, bits, span_const, self_q)) al &= ~(1u << k);
	// write
	u32 total = 0;
	u32 at = sd_block_exscan((u32)__popc(al), ws, &total);     // (every record was read long before)
#pragma unroll
	for (int k = 0; k < LQ_SD_DRPT; ++k) if (al >> k & 1u) R[at++] = rc[k];
	if (t == 0) { scnt[bk] = total; if (dp.stats) atomicAdd(&dp.stats[2], (unsigned long long)total); }
}

// The buckets beyond LQ_SD_DCAP records (listed by k_seed_bdesc), a few blocks striding over the list: the same phases once per
// stretch of the bucket's targets, the records read from memory in every phase and the verdicts kept as a bit per record;
// beyond LQ_SD_BIGCAP records: pairs only.
__global__ void __launch_bounds__(LQ_SD_DTHREADS)
k_seed_decide_big(SeedDecIn in, const SeedBk *bd, const u32 *biglist, SeedDecide dp, SeedBits bits, u32 span_const, u64 *rec, u32 *scnt)
{
	__shared__ u32 ends[LQ_SD_NPAIR / 2];                    // 16-bit halves (sd_rank_pairs)
	__shared__ u32 hist[LQ_SD_HWORDS + 1];
	__shared__ u32 live[LQ_SD_BIGCAP / 32];
	__shared__ u32 ws[17];
	const u32 t = threadIdx.x;
	const u32 n_big = biglist[0];
	const u32 pmask = (1u << dp.pair_bits) - 1u;
	const u32 sh_p = bits.jb + bits.db, sh_d = bits.jb + dp.dshift;
	const u32 dmask = (1u << (bits.db > dp.dshift ? bits.db - dp.dshift : 0u)) - 1u;     // the diagonal's bin: record >> sh_d & dmask
	for (u32 bi = blockIdx.x; bi < n_big; bi += gridDim.x) {
		const u32 bk = biglist[1u + bi];
		const SeedBk B = bd[bk];
		const u32 n = B.n, q = B.q, p0 = B.rid0 << 1;             // (rid << 1 | strand) = record >> sh_p
		u64 *R = rec + B.b0;
		const bool self_q = dp.no_self && in.self_off[q] != in.self_off[q + 1];
		const bool rare = self_q || in.ava.t_rank != nullptr;
		const u32 ql = in.qlen[q];
		const bool pairs_only = n > dp.bigcap;
		u32 width = B.nt, npass = 1;                              // targets per pass
		if (!pairs_only) {
			npass = (n + n / 4 + dp.dcap - 1) / dp.dcap;
			const u32 by_pairs = (2u * B.nt + pmask) / (pmask + 1u);
			if (npass < by_pairs && dp.pair_bits == LQ_SD_PAIR_BITS) npass = by_pairs;   // (every target a counter of its own, unless a test asked for few)
			if (npass > B.nt) npass = B.nt ? B.nt : 1;
			width = (B.nt + npass - 1) / npass;
			for (u32 i = t; i < (n + 31) / 32; i += LQ_SD_DTHREADS) live[i] = 0;
		}
		for (u32 pass = 0; pass < npass; ++pass) {
			const u32 lo = pass * width * 2u, span = width * 2u;   // the pass's pairs: [lo, lo + span) of the slice's
			__syncthreads();
			for (u32 i = t; i < LQ_SD_NPAIR / 2; i += LQ_SD_DTHREADS) ends[i] = 0;
			if (!pairs_only) for (u32 i = t; i < dp.hwords + 1u; i += LQ_SD_DTHREADS) hist[i] = 0;
			__syncthreads();
			for (u32 i = t; i < n; i += LQ_SD_DTHREADS) {
				const u32 lp = (u32)(R[i] >> sh_p) - p0 - lo;
				if (lp < span && sd_h16_get(ends, lp & pmask) < 0x8000u) sd_h16_add(ends, lp & pmask, 1);   // (at most blockDim more adds can slip past the test: no carry)
			}
			__syncthreads();
			if (pairs_only) break;
			sd_rank_pairs(ends, ws, in, dp, ql, B.rid0 + pass * width, span > pmask + 1u, t);
			__syncthreads();
			for (u32 i = t; i < n; i += LQ_SD_DTHREADS) {
				const u64 r = R[i];
				const u32 lp = (u32)(r >> sh_p) - p0 - lo;
				if (lp < span) sd_count_bin(hist, sd_h16_get(ends, lp & pmask), (u32)(r >> sh_d) & dmask, dp.hwords);
			}
			__syncthreads();
			for (u32 i = t; i < n; i += LQ_SD_DTHREADS) {
				const u64 r = R[i];
				const u32 lp = (u32)(r >> sh_p) - p0 - lo;
				if (lp < span && sd_decide_bin(hist, sd_h16_get(ends, lp & pmask), (u32)(r >> sh_d) & dmask, dp.n_min)) atomicOr(&live[i >> 5], 1u << (i & 31u));
			}
		}
		__syncthreads();
		u32 done = 0;                                             // survivors written so far (block-uniform)
		for (u32 base = 0; base < n; base += LQ_SD_DTHREADS) {
			const u32 i = base + t;
			const u64 r = i < n ? R[i] : 0;
			bool a = i < n && (pairs_only ? sd_h16_get(ends, ((u32)(r >> sh_p) - p0) & pmask) >= dp.n_min : (live[i >> 5] >> (i & 31u) & 1u) != 0u);
			if (a && rare && sd_rare_drop(r, q, in, bits, span_const, self_q)) a = false;
			u32 total = 0;
			const u32 at = sd_block_exscan(a ? 1u : 0u, ws, &total);  // (its barriers stand between this round's reads and writes)
			if (a) R[done + at] = r;                                  // done + at <= i: never ahead of what is still to be read
			done += total;
		}
		if (t == 0) { scnt[bk] = done; if (dp.stats) { atomicAdd(&dp.stats[0], (unsigned long long)n); atomicAdd(&dp.stats[2], (unsigned long long)done); atomicAdd(&dp.stats[3], 1ULL); if (pairs_only) atomicAdd(&dp.stats[5], 1ULL); } }
		__syncthreads();                                          // (the next bucket clears what this one still reads)
	}
}

// survivors of every bucket of the chunk, dense: surv[base + soff[bucket] ...] (soff = exclusive scan of scnt); and where
// every query's survivors start (aqf_off[q], for the chunk's queries)
// (surv[i] is survivor number `first` + i of the part)
__global__ void k_seed_collect(const SeedBk *bd, const u64 *rec, const u32 *scnt, const u32 *soff, u64 base, u64 first, u64 *surv, u64 *aqf_off)
{
	const u32 bk = blockIdx.x;
	const SeedBk B = bd[bk];
	if (B.rid0 == 0 && threadIdx.x == 0) aqf_off[B.q] = first + base + soff[bk];     // (a query's first slice starts at target 0, and only that one)
	const u32 n = scnt[bk];
	if (n == 0) return;
	const u64 *R = rec + B.b0;
	u64 *out = surv + base + soff[bk];
	for (u32 i = threadIdx.x; i < n; i += blockDim.x) out[i] = R[i];
}
// the queries [q_lo, q_hi) of a group of chunks: those without a bucket (no hits) start where the next one starts; the end
__global__ void k_seed_fill_off(const u32 *has, u32 q_lo, u32 q_hi, u64 total, u64 *aqf_off)
{
	// one thread: a few thousand queries
	if (blockIdx.x || threadIdx.x) return;
	u64 nxt = total;
	aqf_off[q_hi] = total;
	for (u32 q = q_hi; q-- > q_lo; ) { if (has[q]) nxt = aqf_off[q]; else aqf_off[q] = nxt; }
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
