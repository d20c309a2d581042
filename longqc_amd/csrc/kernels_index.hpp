// longqc_amd/csrc/kernels_index.hpp -- GPU-resident minimizer index of one part and seed collection.
//
// Index (reference index.c:150-236, semantic result "hash -> occurrences sorted by y"): the part's
// minimizers are produced in ascending y by k_sketch, a stable LSD radix sort on the 2k-bit hash
// groups them (kernels_isort.hpp), run heads become the distinct keys, and an
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
#include "kernels_isort.hpp"

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

// The distinct keys of the sorted key array and where each one's run starts (ukey, ustart), in two passes over the keys and no
// array as long as the keys in between: a tile of LQ_HEAD_TILE keys counts its run heads (key[i] != key[i - 1]); the tile counts
// are scanned; the same tile then ranks its heads (block scan) and writes them.  (Round 4 wrote a flag per minimizer, scanned
// the flags into 8-byte indices and read both back: 13.6 ms per 4-Gbase part against 4.)
#define LQ_HEAD_THREADS 256
#define LQ_HEAD_PER 8
#define LQ_HEAD_TILE (LQ_HEAD_THREADS * LQ_HEAD_PER)
template <class KT>
__device__ __forceinline__ u32 lq_head_bits(const KT *key, u64 n, u64 i0, KT *mine)
{
	u32 bits = 0;
	const u32 lane = threadIdx.x & 63;
	if (i0 - (u64)lane * LQ_HEAD_PER + 64 * LQ_HEAD_PER <= n) {   // the whole wave's 512 keys exist (the same answer in every lane: the shuffle below is wave-uniform)
		// the thread's 8 keys as 16-byte loads (i0 is a multiple of 8: aligned): a wave reads 2 or 4 KB in one piece; the key before them
		// is the last key of the lane before (only the wave's first lane reads it from memory)
		constexpr int NV = LQ_HEAD_PER * (int)sizeof(KT) / 16;
		const uint4 *src = reinterpret_cast<const uint4*>(key + i0);
		uint4 q[NV];
#pragma unroll
		for (int j = 0; j < NV; ++j) q[j] = src[j];
		memcpy(mine, q, sizeof(q));
		KT prev = __shfl_up(mine[LQ_HEAD_PER - 1], 1);
		if (lane == 0) prev = i0 > 0 ? key[i0 - 1] : (KT)0;
#pragma unroll
		for (int k = 0; k < LQ_HEAD_PER; ++k) { if ((i0 == 0 && k == 0) || mine[k] != prev) bits |= 1u << k; prev = mine[k]; }
		return bits;
	}
	KT prev = i0 > 0 && i0 <= n ? key[i0 - 1] : (KT)0;
#pragma unroll
	for (int k = 0; k < LQ_HEAD_PER; ++k) {
		const u64 i = i0 + (u64)k;
		if (i < n) { const KT v = key[i]; mine[k] = v; if (i == 0 || v != prev) bits |= 1u << k; prev = v; }
	}
	return bits;
}
template <class KT>
__global__ void __launch_bounds__(LQ_HEAD_THREADS)
k_head_count(const KT *key, u64 n, u32 *tile_cnt)
{
	__shared__ u32 acc;
	if (threadIdx.x == 0) acc = 0;
	__syncthreads();
	KT mine[LQ_HEAD_PER];
	const u32 c = (u32)__popc(lq_head_bits(key, n, (u64)blockIdx.x * LQ_HEAD_TILE + (u64)threadIdx.x * LQ_HEAD_PER, mine));
	u32 w = c;
	for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
	if ((threadIdx.x & 63) == 0) atomicAdd(&acc, w);
	__syncthreads();
	if (threadIdx.x == 0) tile_cnt[blockIdx.x] = acc;
}
// tile_off = exclusive scan of tile_cnt
template <class KT>
__global__ void __launch_bounds__(LQ_HEAD_THREADS)
k_head_fill(const KT *key, u64 n, const u64 *tile_off, u64 *ukey, u64 *ustart)
{
	__shared__ u32 wsum[LQ_HEAD_THREADS / 64 + 1];
	KT mine[LQ_HEAD_PER];
	const u64 i0 = (u64)blockIdx.x * LQ_HEAD_TILE + (u64)threadIdx.x * LQ_HEAD_PER;
	const u32 bits = lq_head_bits(key, n, i0, mine);
	const u32 c = (u32)__popc(bits), lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	u32 x = c;
	for (u32 o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o); if (lane >= o) x += y; }
	if (lane == 63) wsum[wv] = x;
	__syncthreads();
	u32 before = 0;
	for (u32 w = 0; w < wv; ++w) before += wsum[w];
	u64 at = tile_off[blockIdx.x] + before + x - c;
#pragma unroll
	for (int k = 0; k < LQ_HEAD_PER; ++k) if (bits >> k & 1u) { ukey[at] = (u64)mine[k]; ustart[at] = i0 + (u64)k; ++at; }
}

// The two kernels above in one pass over the keys (round 6, keys of at most 24 bits: at most 2^24 distinct ones, so ukey / ustart can
// be sized before their number is known): a block takes LQ_HEADLB_SUB tiles of keys in a row (one ticket and one granule per 16 384 keys: 5.0 ms for 1.34 G keys; per 2048 keys
// the kernel ran at the pace of the atomic on the ticket, 7.7 ms; per 32 768 keys 182 registers and 6.1 ms), counts their run heads, looks back
// over the blocks before it (one granule per block, kernels_isort.hpp) and writes its heads where they belong; the last block
// leaves the number of distinct keys in *n_keys.
#define LQ_HEADLB_SUB 8
template <class KT>
__global__ void __launch_bounds__(LQ_HEAD_THREADS)
k_head_lookback(const KT *key, u64 n, u64 *status, u32 *ticket, u64 *ukey, u64 *ustart, u64 *n_keys)
{
	__shared__ u32 wsum[LQ_HEADLB_SUB][LQ_HEAD_THREADS / 64];
	__shared__ u64 s_excl;
	__shared__ u32 s_tile;
	if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
	__syncthreads();
	const u32 tile = s_tile, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const u64 b0 = (u64)tile * (LQ_HEADLB_SUB * LQ_HEAD_TILE);
	u32 bits[LQ_HEADLB_SUB], x[LQ_HEADLB_SUB];
#pragma unroll
	for (int j = 0; j < LQ_HEADLB_SUB; ++j) {
		KT mine[LQ_HEAD_PER];
		bits[j] = lq_head_bits(key, n, b0 + (u64)j * LQ_HEAD_TILE + (u64)threadIdx.x * LQ_HEAD_PER, mine);
		u32 v = (u32)__popc(bits[j]);
		for (u32 o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(v, o); if (lane >= o) v += y; }
		x[j] = v;                                                  // heads of this sub-tile up to and including this thread, within its wave
		if (lane == 63) wsum[j][wv] = v;
	}
	__syncthreads();
	u32 total = 0, before[LQ_HEADLB_SUB];
#pragma unroll
	for (int j = 0; j < LQ_HEADLB_SUB; ++j) {
		before[j] = total;
		for (u32 w = 0; w < LQ_HEAD_THREADS / 64; ++w) { if (w < wv) before[j] += wsum[j][w]; total += wsum[j][w]; }
	}
	if (wv == 0) {
		const u64 excl = lq_tile_lookback(status, tile, (u64)total, lane);
		if (lane == 0) { s_excl = excl; if (b0 + (u64)LQ_HEADLB_SUB * LQ_HEAD_TILE >= n) *n_keys = excl + total; }
	}
	__syncthreads();
#pragma unroll
	for (int j = 0; j < LQ_HEADLB_SUB; ++j) {
		if (!bits[j]) continue;
		const u64 i0 = b0 + (u64)j * LQ_HEAD_TILE + (u64)threadIdx.x * LQ_HEAD_PER;
		u64 at = s_excl + before[j] + x[j] - (u32)__popc(bits[j]);
#pragma unroll
		for (int k = 0; k < LQ_HEAD_PER; ++k) if (bits[j] >> k & 1u) { ukey[at] = (u64)key[i0 + (u64)k]; ustart[at] = i0 + (u64)k; ++at; }   // (the few keys that are heads are read again: they are in L2)
	}
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
