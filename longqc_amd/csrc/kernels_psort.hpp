// longqc_amd/csrc/kernels_psort.hpp -- sort by x where klib's order cannot be observed.
//
// The reference sorts a query's anchors with klib's unstable in-place radix sort (lqmap.c:238, ksort.h:99-134), and
// the chaining DP sees the array order of anchors with equal x.  kernels_sort.hpp reproduces that order with serial
// token walks.  But a sub-array whose anchors all differ in x has exactly one sorted arrangement -- klib's, and
// every other correct sort's.  Equal x needs two minimizers of the query with the same (hash, strand)
// (k_dup_mark, kernels_index.hpp): sub-arrays holding fewer than two anchors of such minimizers (a whole query
// without marked minimizers, or a bucket of a klib pass that received at most one marked anchor) come here and are
// sorted with plain parallel passes: no klib levels, no walks, free choice of digits.
//
// Key: only the bits of x that vary inside the part count -- strand (bit 63), the low `rbits` of rid (bits 32..),
// the low `pbits` of the position; lq_ckey packs them into a compact key of K = 1 + rbits + pbits bits.  A segment
// carries `rem`, the number of low key bits it still has to be sorted by (the bits above are equal inside it).
//   * segments of more than LQ_PS_FIN_BIG elements: one partition pass on the top nbits <= 8 of the remaining bits
//     (per 2048-element tile: histogram -> one atomic range reservation per digit -> LDS-staged, run-contiguous
//     writes into the other buffer; the order inside a bucket is whatever the atomics give, which is fine: the
//     keys are distinct and the finish below looks at all remaining bits);
//   * segments up to LQ_PS_FIN_BIG (8192) / LQ_PS_FIN_SMALL (1024) elements: finished by one block: elements in
//     registers, keys in LDS, split by the next 10 / 8 key bits, then every element ranks itself among the (few) elements
//     of its sub-bucket by the full remaining key; written to A whatever buffer the segment was in.
// All list lengths live on the device; kernels take upper-bound grids and stride over the lists.
//
// Where a segment's anchors are: in A, in B, or -- a bucket that has just left klib's passes (kernels_rsort.hpp) -- still
// spread over the originals and named by the records of its range (buf 2 / 3: record array 0 / 1): the first kernel that
// touches such a segment gathers the anchors through the indices, so leaving klib costs no pass of its own.  A gathered
// segment always goes to A (the originals live in B until every such segment has been read: psort_run finishes them and
// runs their first partition pass before any pass writes to B).
#pragma once
#include "lq_common.hpp"
#include "kernels_sort.hpp"
#include "kernels_rsort.hpp"

struct KeyMap { u32 pbits, rbits; };                 // varying low bits of the position and of rid in this part
struct alignas(16) PSeg { u64 off; u32 len; u8 rem; u8 buf; u8 nbits; u8 pad; };   // buf: 0 = data in A, 1 = in B, 2 / 3 = originals named by record array 0 / 1
struct PPlan { u32 tile0, cnt0; };                   // first tile / first counter of a big segment
struct PsData { mm128 *A, *B; const RRec *R[2]; };   // the lane's buffers; B holds the originals of the klib queries
#define LQ_PS_SKIP 0xffu                             // PSeg.nbits of a big segment whose pass found every key in one bucket: nothing to move

__device__ __forceinline__ mm128 lq_ps_load(const PSeg &sg, const PsData &P, u32 i)
{
	if (sg.buf < 2) return (sg.buf ? P.B : P.A)[sg.off + i];
	return P.B[LQ_R_IDX(P.R[sg.buf - 2][sg.off + i].im)];
}
__device__ __forceinline__ u64 lq_ps_load_y(const PSeg &sg, const PsData &P, u32 i)
{
	if (sg.buf < 2) return (sg.buf ? P.B : P.A)[sg.off + i].y;
	return P.B[LQ_R_IDX(P.R[sg.buf - 2][sg.off + i].im)].y;
}
__device__ __forceinline__ u64 lq_ps_load_x(const PSeg &sg, const PsData &P, u32 i)
{
	if (sg.buf < 2) return (sg.buf ? P.B : P.A)[sg.off + i].x;
	return P.B[LQ_R_IDX(P.R[sg.buf - 2][sg.off + i].im)].x;
}

#define LQ_PS_FIN_SMALL 1024
#define LQ_PS_FIN_BIG   8192
#ifndef LQ_PS_TILE
#define LQ_PS_TILE      2048
#endif
#define LQ_PS_CHILD     4096      // aimed child size of a partition pass: half of what the finish takes
#define LQ_PS_THREADS   256

// counters of one batch's sort (device): indices into L.sort_cnt.  Two sets of psort lists: set 0 takes whole queries
// (k_sort_init) and is sorted on its own stream while klib's passes run; set 1 collects the buckets that leave them.
enum { LQ_C_KLIB0 = 0, LQ_C_KLIB1, LQ_C_TWO, LQ_C_WALK0, LQ_C_WALK1, LQ_C_WALK2, LQ_C_WALK3, LQ_C_WALK4, LQ_C_OVERFLOW, LQ_C_TILES,
       LQ_C_HIST0 = 10,        // (64-bit) anchors the first level's histogram kernel turned into records
       LQ_C_TWO_TILES = 12,    // tiles of the level's two-bucket sub-arrays
       LQ_C_PS0 = 16, LQ_C_PS1 = 32,
       // 64-bit tallies of the elements each kind of kernel really moved (algorithmic bytes of the stage times)
       LQ_C_HIST = 48, LQ_C_SCATTERED = 50, LQ_C_PART0 = 52, LQ_C_FINS0 = 54, LQ_C_FINB0 = 56, LQ_C_PART1 = 58, LQ_C_FINS1 = 60, LQ_C_FINB1 = 62,
       LQ_C_LEN0 = 64,         // [5] sub-arrays of the level to come by walk size class (by length alone: an upper bound of the class lists)
       LQ_C_N = 72 };
enum { LQ_P_BIG0 = 0, LQ_P_BIG1, LQ_P_FIN_S, LQ_P_FIN_B, LQ_P_TILES, LQ_P_CNT, LQ_P_OVERFLOW };   // offsets inside a set's counters
struct PsLists { struct PSeg *big[2], *fin_s, *fin_b; u32 *cnt; u32 cap_big, cap_fin;
                 u32 fin_s_max, fin_b_max, child_target; };   // size limits of the two finishing kernels, aimed child size of a pass (tests shrink them)

__device__ __forceinline__ u64 lq_ckey(u64 x, const KeyMap km)
{
	return (x & ((1ULL << km.pbits) - 1)) | ((x >> 32) & ((1ULL << km.rbits) - 1)) << km.pbits | (x >> 63) << (km.pbits + km.rbits);
}
// key bits below bit `shift` of x (a klib bucket made by the pass on the byte at `shift` still differs there only)
__device__ __forceinline__ u32 lq_rem_below(u32 shift, const KeyMap km)
{
	if (shift <= 32) return shift < km.pbits ? shift : km.pbits;
	const u32 r = shift - 32;
	return km.pbits + (r < km.rbits ? r : km.rbits) + (shift > 63 ? 1u : 0u);
}

// append a segment to the list its size asks for (big_slot: LQ_P_BIG0 or LQ_P_BIG1)
__device__ __forceinline__ void lq_ps_route(PSeg sg, const PsLists L, u32 big_slot)
{
	// (rem == 0: nothing to sort, only to bring home to A *in the order it is in* -- a bucket of klib's last pass holds one x, and
	// the order of its anchors is klib's; the finishing kernels copy such a segment in order whatever its length, a partition
	// pass would move it in the order its atomics happen to give)
	if (sg.len <= L.fin_s_max || sg.rem == 0) { const u32 s = atomicAdd(&L.cnt[LQ_P_FIN_S], 1u); if (s < L.cap_fin) L.fin_s[s] = sg; else atomicOr(&L.cnt[LQ_P_OVERFLOW], 1u); }
	else if (sg.len <= L.fin_b_max) { const u32 s = atomicAdd(&L.cnt[LQ_P_FIN_B], 1u); if (s < L.cap_fin) L.fin_b[s] = sg; else atomicOr(&L.cnt[LQ_P_OVERFLOW], 1u); }
	else { const u32 s = atomicAdd(&L.cnt[big_slot], 1u); if (s < L.cap_big) L.big[big_slot][s] = sg; else atomicOr(&L.cnt[LQ_P_OVERFLOW], 1u); }
}

// ---- plan of one partition pass: digits, tiles and counters of every big segment (one block) ----------------
__global__ void __launch_bounds__(256)
k_ps_plan(PSeg *segs, const u32 *n_p, PPlan *plan, u32 *cnt, u32 cap_cnt, u32 cap_tiles, u32 child_target, u32 *n_next_zero)
{
	__shared__ u32 st[256], sc[256], tmp[256], tot_t, tot_c, base_t, base_c;
	const u32 n = *n_p, t = threadIdx.x;
	if (t == 0) { base_t = 0; base_c = 0; *n_next_zero = 0; }    // (the pass appends its children to the other big list)
	__syncthreads();
	for (u32 s0 = 0; s0 < n; s0 += 256) {
		const u32 s = s0 + t;
		u32 nt = 0, nc = 0;
		if (s < n) {
			PSeg sg = segs[s];
			u32 nb = 1;
			while (nb < 8 && ((u64)child_target << nb) < sg.len) ++nb;   // children of about one tile
			if (nb > sg.rem) nb = sg.rem;
			sg.nbits = (u8)nb;
			segs[s] = sg;
			nt = (sg.len + LQ_PS_TILE - 1) / LQ_PS_TILE; nc = 1u << nb;
		}
		st[t] = nt; sc[t] = nc;
		__syncthreads();
		lq_scan256(st, tmp, &tot_t);
		lq_scan256(sc, tmp, &tot_c);
		if (s < n) { PPlan p; p.tile0 = base_t + st[t]; p.cnt0 = base_c + sc[t]; plan[s] = p; }
		__syncthreads();
		if (t == 0) { base_t += tot_t; base_c += tot_c; }
		__syncthreads();
	}
	if (t == 0) {
		PPlan e; e.tile0 = base_t; e.cnt0 = base_c; plan[n] = e;    // sentinel: totals
		cnt[LQ_P_TILES] = base_t; cnt[LQ_P_CNT] = base_c;
		if (base_c > cap_cnt || base_t > cap_tiles) atomicOr(&cnt[LQ_P_OVERFLOW], 2u);
	}
}

// segment of every tile of the pass (a tile's block then finds its segment with one load instead of a binary search over the
// plan -- a dozen dependent loads before it could touch an anchor); the pass's bucket counters and varying-bit words start
// from zero.  One block per segment (strided), its threads over the segment's tiles and counters.
__global__ void __launch_bounds__(256)
k_ps_tilemap(const PPlan *plan, const u32 *n_p, const u32 *cnt, u32 *tmap, u32 cap_tiles, u32 *gcnt, unsigned long long *gdiff)
{
	const u32 n = *n_p;
	if (cnt[LQ_P_OVERFLOW] & 2u) return;
	for (u32 s = blockIdx.x; s < n; s += gridDim.x) {
		const PPlan a = plan[s], b = plan[s + 1];
		for (u32 t = a.tile0 + threadIdx.x; t < b.tile0 && t < cap_tiles; t += blockDim.x) tmap[t] = s;
		for (u32 c = a.cnt0 + threadIdx.x; c < b.cnt0; c += blockDim.x) gcnt[c] = 0;
		if (threadIdx.x == 0) gdiff[s] = 0;
	}
}

// ---- histogram of the pass's digit, per big segment (tiles stride over the grid) ------------------------------
// Also: gdiff[segment] |= key ^ (key of the segment's first element) -- the key bits that vary inside the segment.  A
// segment whose keys agree in all the bits of this pass's digit (e.g. 30 000 anchors of an ultra-long query on one target:
// strand and rid are the top 1 + rbits bits of the key) is not moved at all: k_ps_scan re-lists it with `rem` cut down to
// its highest varying bit.
__global__ void __launch_bounds__(LQ_PS_THREADS)
k_ps_hist(const PSeg *segs, const u32 *n_p, const PPlan *plan, const u32 *tmap, const u32 *cnt, PsData P, KeyMap km, u32 *gcnt, unsigned long long *gdiff)
{
	__shared__ u32 lh[256];
	__shared__ unsigned long long ldiff[LQ_PS_THREADS / 64];
	const u32 n = *n_p;
	if (n == 0 || (cnt[LQ_P_OVERFLOW] & 2u)) return;
	const u32 n_tiles = cnt[LQ_P_TILES], t = threadIdx.x;
	for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const u32 s = tmap[tile];
		const PSeg sg = segs[s];
		const PPlan pl = plan[s];
		const u32 i0 = (tile - pl.tile0) * LQ_PS_TILE, i1 = i0 + LQ_PS_TILE < sg.len ? i0 + LQ_PS_TILE : sg.len;
		const u32 sh = sg.rem - sg.nbits, dm = (1u << sg.nbits) - 1;
		const u64 key0 = lq_ckey(lq_ps_load_x(sg, P, 0), km);
		u64 diff = 0;
		lh[t] = 0;
		__syncthreads();
		for (u32 i = i0 + t; i < i1; i += LQ_PS_THREADS) { const u64 key = lq_ckey(lq_ps_load_x(sg, P, i), km); diff |= key ^ key0; atomicAdd(&lh[(u32)(key >> sh) & dm], 1u); }
		for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o);
		if ((t & 63) == 0) ldiff[t >> 6] = diff;
		__syncthreads();
		if (t <= dm && lh[t]) atomicAdd(&gcnt[pl.cnt0 + t], lh[t]);
		if (t == 0) { unsigned long long dd = 0; for (u32 w = 0; w < LQ_PS_THREADS / 64; ++w) dd |= ldiff[w]; if (dd) atomicOr(&gdiff[s], dd); }
		__syncthreads();
	}
}

// ---- bucket offsets and children of every big segment (one block per segment, strided) ----------------------
__global__ void __launch_bounds__(256)
k_ps_scan(PSeg *segs, const u32 *n_p, const PPlan *plan, const u32 *gcnt, u32 *gcur, const unsigned long long *gdiff, PsLists L, u32 big_next_slot, unsigned long long *tally)
{
	__shared__ u32 v[256], tmp[256];
	const u32 n = *n_p, t = threadIdx.x;
	if (L.cnt[LQ_P_OVERFLOW] & 2u) return;
	for (u32 s = blockIdx.x; s < n; s += gridDim.x) {
		const PSeg sg = segs[s];
		const PPlan pl = plan[s];
		const u32 nb = 1u << sg.nbits;
		{	// every key in one bucket of this pass: the segment stays where it is and is listed again with the bits that do vary
			const u64 low = sg.rem >= 64 ? ~0ULL : ((1ULL << sg.rem) - 1);
			const u64 df = (u64)gdiff[s] & low;
			// (all keys equal and not in A yet: no short cut -- the pass below is then the copy that brings the segment home.  A segment
			// still named by records neither: its pass gathers it into A, and every such segment must have left B by the end of the
			// set's first pass -- later passes use B as their other buffer.)
			if ((df >> (sg.rem - sg.nbits)) == 0 && sg.buf < 2 && (df != 0 || sg.buf == 0)) {   // (uniform over the block)
				if (t == 0) {
					PSeg ch = sg; ch.nbits = 0;
					ch.rem = df ? (u8)(64 - __builtin_clzll(df)) : 0;
					if (ch.rem) lq_ps_route(ch, L, big_next_slot);
					PSeg mk = sg; mk.nbits = LQ_PS_SKIP; segs[s] = mk;          // k_ps_scatter leaves it alone
				}
				__syncthreads();
				continue;
			}
		}
		if (t == 0 && tally) atomicAdd(tally, (unsigned long long)sg.len);
		// the children differ in the key bits below this pass's digit that vary inside the segment, at most
		u32 crem = sg.rem - sg.nbits;
		{
			const u64 dl = (u64)gdiff[s] & (crem >= 64 ? ~0ULL : ((1ULL << crem) - 1));
			crem = dl ? 64 - (u32)__builtin_clzll(dl) : 0;
		}
		const u32 c = t < nb ? gcnt[pl.cnt0 + t] : 0;
		v[t] = c;
		__syncthreads();
		lq_scan256(v, tmp, nullptr);
		if (t < nb) {
			gcur[pl.cnt0 + t] = v[t];
			if (c) {
				PSeg ch; ch.off = sg.off + v[t]; ch.len = c; ch.rem = (u8)crem; ch.buf = sg.buf >= 2 ? 0 : sg.buf ^ 1; ch.nbits = 0; ch.pad = 0;
				if (ch.rem == 0 || c == 1) { if (ch.buf) { ch.rem = 0; lq_ps_route(ch, L, big_next_slot); } }   // nothing left to sort: only bring it home to A
				else lq_ps_route(ch, L, big_next_slot);
			}
		}
		__syncthreads();
	}
}

// ---- the partition pass: every tile moves its elements into their buckets in the other buffer ---------------
__global__ void __launch_bounds__(LQ_PS_THREADS)
k_ps_scatter(const PSeg *segs, const u32 *n_p, const PPlan *plan, const u32 *tmap, const u32 *cnt, PsData P, KeyMap km, u32 *gcur)
{
	__shared__ mm128 stage[LQ_PS_TILE];
	__shared__ u32 lh[256], lo[256], fill[256], gb[256], tmp[256];
	const u32 n = *n_p;
	if (n == 0 || (cnt[LQ_P_OVERFLOW] & 2u)) return;
	const u32 n_tiles = cnt[LQ_P_TILES], t = threadIdx.x;
	constexpr u32 PER = LQ_PS_TILE / LQ_PS_THREADS;
	for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const u32 s = tmap[tile];
		const PSeg sg = segs[s];
		const PPlan pl = plan[s];
		if (sg.nbits == LQ_PS_SKIP) continue;                     // (uniform) every key in one bucket: k_ps_scan listed the segment again
		const u32 i0 = (tile - pl.tile0) * LQ_PS_TILE, i1 = i0 + LQ_PS_TILE < sg.len ? i0 + LQ_PS_TILE : sg.len;
		mm128 *dst = (sg.buf == 1 || sg.buf >= 2 ? P.A : P.B) + sg.off;
		const u32 sh = sg.rem - sg.nbits, dm = (1u << sg.nbits) - 1;
		lh[t] = 0; fill[t] = 0;
		__syncthreads();
		mm128 e[PER];
		for (u32 k = 0; k < PER; ++k) {
			const u32 i = i0 + t + k * LQ_PS_THREADS;
			if (i < i1) { e[k] = lq_ps_load(sg, P, i); atomicAdd(&lh[(u32)(lq_ckey(e[k].x, km) >> sh) & dm], 1u); }
		}
		__syncthreads();
		const u32 mine = lh[t];
		lo[t] = mine;
		if (mine) gb[t] = atomicAdd(&gcur[pl.cnt0 + t], mine);  // this tile's run inside bucket t
		__syncthreads();
		lq_scan256(lo, tmp, nullptr);
		for (u32 k = 0; k < PER; ++k) {
			const u32 i = i0 + t + k * LQ_PS_THREADS;
			if (i < i1) { const u32 d = (u32)(lq_ckey(e[k].x, km) >> sh) & dm; stage[lo[d] + atomicAdd(&fill[d], 1u)] = e[k]; }
		}
		__syncthreads();
		for (u32 p = t; p < i1 - i0; p += LQ_PS_THREADS) {       // run-contiguous writes
			const mm128 a = stage[p];
			const u32 d = (u32)(lq_ckey(a.x, km) >> sh) & dm;
			dst[gb[d] + (p - lo[d])] = a;
		}
		__syncthreads();
	}
}

// ---- finish: one block sorts a segment by all its remaining key bits and writes it to A -------------------------
// LDS holds every element's key, split in two: the sub-bucket digit (the top SB bits of the range the segment's keys really
// span) and the bits below it, plus the sub-bucket histogram and the elements' indices grouped by sub-bucket.  An element's
// place = start of its sub-bucket + the number of smaller keys in it (a handful of elements: about n / 2^SB).  A thread keeps
// only y of the elements it loaded: x is the key -- the compact key holds every bit of x that is not zero in the whole part
// (lq_ckey), so the x that belongs at a place is rebuilt from the key that landed there.  (Holding the anchors, 4 registers
// each, took 128 registers and 27 spilled ones per lane in the 1024-thread shape: one block per CU.  Now two fit.)
// All loads are done before the first barrier, so the segment may be sorted in place.
// KEY = u32 when the key bits below the sub-bucket digit fit 32 bits for every segment of the part (K - SB <= 32, the
// usual case), u64 otherwise.
__device__ __forceinline__ u64 lq_ckey_inv(u64 ck, const KeyMap km)
{
	return (ck & ((1ULL << km.pbits) - 1)) | ((ck >> km.pbits) & ((1ULL << km.rbits) - 1)) << 32 | ((ck >> (km.pbits + km.rbits)) & 1) << 63;
}
template <int CAP, int THREADS, int SB, class KEY>
__global__ void __launch_bounds__(THREADS, (sizeof(KEY) == 4 || THREADS < 1024 ? 8 : 4))
k_ps_finish(const PSeg *segs, const u32 *n_p, PsData P, KeyMap km, unsigned long long *tally)
{
	constexpr int NSB = 1 << SB, PER = CAP / THREADS, SPT = NSB / THREADS > 0 ? NSB / THREADS : 1;
	static_assert(CAP % THREADS == 0 && (NSB % THREADS == 0 || NSB < THREADS), "shape");
	__shared__ KEY keys[CAP];                                  // first the low 32 (64) bits of the remaining key, then the bits below the digit
	__shared__ u16 dgs[CAP];                                   // first the bits above those (u32 keys), then the digit
	__shared__ u16 perm[CAP];
	__shared__ u32 hist[NSB], beg[NSB], fill[NSB], wsum[THREADS / 64 + 1];
	__shared__ u64 rmin[THREADS / 64], rmax[THREADS / 64], chigh_s;
	const u32 n_seg = *n_p, t = threadIdx.x;
	for (u32 s = blockIdx.x; s < n_seg; s += gridDim.x) {
		const PSeg sg = segs[s];
		const u32 n = sg.len;
		mm128 *out = P.A + sg.off;
		if (t == 0 && tally) atomicAdd(tally, (unsigned long long)n);
		if (sg.rem == 0 || n == 1) {                             // already in order: home to A
			if (sg.buf) for (u32 i = t; i < n; i += THREADS) out[i] = lq_ps_load(sg, P, i);
			continue;
		}
		const u64 km_mask = sg.rem >= 64 ? ~0ULL : ((1ULL << sg.rem) - 1);
		for (u32 c = t; c < NSB; c += THREADS) { hist[c] = 0; fill[c] = 0; }
		u64 y[PER];
		u64 kmin = ~0ULL, kmax = 0;
		// PER > 4 (the 1024-thread shape): x and y in two loads -- x is dead once its key is in LDS, so the loads in flight need half
		// the registers (64 in all: two blocks per CU).  PER <= 4: one 16-byte load per anchor; most of these segments are read
		// through records, and two 8-byte gathers cost twice the sectors of one 16-byte gather (PMC: 8.6 vs 6.2 GB per launch)
		constexpr bool SPLIT = PER > 4;
		u64 xk[SPLIT ? 1 : PER];
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const u32 i = t + (u32)k * THREADS, ic = i < n ? i : n - 1;   // (no branch: y[] stays in plain registers)
			if (SPLIT) y[k] = lq_ps_load_y(sg, P, ic);
			else { const mm128 e = lq_ps_load(sg, P, ic); y[k] = e.y; xk[k] = e.x; }
		}
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const u32 i = t + (u32)k * THREADS;
			if (i < n) {
				const u64 ck = lq_ckey(SPLIT ? lq_ps_load_x(sg, P, i) : xk[SPLIT ? 0 : k], km), kf = ck & km_mask;
				if (i == 0) chigh_s = ck & ~km_mask;             // the key bits the whole segment shares
				keys[i] = (KEY)kf;
				if (sizeof(KEY) == 4) dgs[i] = (u16)(kf >> 32);
				kmin = kf < kmin ? kf : kmin; kmax = kf > kmax ? kf : kmax;
			}
		}
		// The sub-bucket digit is taken from the range the segment's keys really span, not from the top of the bits they might
		// differ in: a child of a partition pass holds a few dozen targets and a stretch of positions, and the top SB bits of its
		// `rem` say little -- a few crowded sub-buckets, and the rank loop below is quadratic in a sub-bucket's size.
		for (int o = 32; o > 0; o >>= 1) { const u64 a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o); kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; }
		if ((t & 63) == 0) { rmin[t >> 6] = kmin; rmax[t >> 6] = kmax; }
		__syncthreads();
		for (u32 w = 0; w < THREADS / 64; ++w) { kmin = rmin[w] < kmin ? rmin[w] : kmin; kmax = rmax[w] > kmax ? rmax[w] : kmax; }
		const u64 chigh = chigh_s;
		const u64 range = kmax - kmin;
		const u32 bits = range ? 64 - (u32)__builtin_clzll(range) : 0, sh = bits > (u32)SB ? bits - (u32)SB : 0;
		const u64 lo_mask = ((u64)1 << sh) - 1;
#pragma unroll
		for (int k = 0; k < PER; ++k) {                          // (a thread rewrites the entries of its own elements only)
			const u32 i = t + (u32)k * THREADS;
			if (i < n) {
				const u64 kf = sizeof(KEY) == 4 ? ((u64)dgs[i] << 32 | (u64)keys[i]) : (u64)keys[i];
				const u64 key = kf - kmin;
				const u32 d = (u32)(key >> sh);
				keys[i] = (KEY)(key & lo_mask); dgs[i] = (u16)d;
				atomicAdd(&hist[d], 1u);
			}
		}
		__syncthreads();
		{	// exclusive scan of hist -> beg: SPT counters per thread, wave scan, wave totals through LDS
			u32 v[SPT], sum = 0;
			for (int q = 0; q < SPT; ++q) { const u32 c = t * SPT + q; v[q] = c < (u32)NSB ? hist[c] : 0; sum += v[q]; }
			u32 inc = sum;
			for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if ((int)(t & 63) >= d) inc += o; }
			if ((t & 63) == 63) wsum[t >> 6] = inc;
			__syncthreads();
			u32 base = 0;
			for (u32 w = 0; w < (t >> 6); ++w) base += wsum[w];
			u32 run = base + inc - sum;
			for (int q = 0; q < SPT; ++q) { const u32 c = t * SPT + q; if (c < (u32)NSB) beg[c] = run; run += v[q]; }
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const u32 i = t + (u32)k * THREADS;
			if (i < n) { const u32 d = dgs[i]; perm[beg[d] + atomicAdd(&fill[d], 1u)] = (u16)i; }
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const u32 i = t + (u32)k * THREADS;
			if (i < n) {
				const u32 d = dgs[i];
				const KEY key = keys[i];
				const u32 b0 = beg[d], b1 = b0 + hist[d];
				// (equal keys = equal x, when the sort is not asked for klib's order: any order among them, but a place each -- ties by
				// the index in the segment, appended to the key where the key leaves room: one comparison)
				u32 r = 0;
				if (sizeof(KEY) == 4) {
					static_assert(CAP <= 8192, "13 bits of index");
					const u64 me = (u64)key << 13 | i;
					for (u32 j = b0; j < b1; ++j) { const u32 pj = perm[j]; r += ((u64)keys[pj] << 13 | pj) < me; }
				} else
					for (u32 j = b0; j < b1; ++j) { const u32 pj = perm[j]; const KEY kj = keys[pj]; r += (kj < key) || (kj == key && pj < i); }
				mm128 e;
				e.x = lq_ckey_inv(chigh | (kmin + ((u64)d << sh | (u64)key)), km);
				e.y = y[k];
				out[b0 + r] = e;
			}
		}
		__syncthreads();
	}
}

// ---- the two entrances from the klib side -----------------------------------------------------------------
// radix_sort_128x entry (ksort.h:130-134) for every query of the batch: arrays of <= 64 elements are insertion sorted;
// a query with marked minimizers (qklib: its anchors were emitted into B, the originals) starts klib's passes at the top
// byte; every other query is free of equal x.
__global__ void k_sort_init(const u64 *aq_off, u64 a_base, u32 n_q, const u32 *qklib, mm128 *A, SortSeg *klib, u32 *cnt,
                            PsLists L, KeyMap km, WalkCaps caps)
{
	const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_q) return;
	const u64 off = aq_off[q] - a_base, len = aq_off[q + 1] - aq_off[q];
	if (len <= LQ_RS_MIN) { if (len > 1) lq_insertion_sort_x(A + off, (u32)len); return; }
	if (qklib[q]) {
		const u32 s = atomicAdd(&cnt[LQ_C_KLIB0], 1u);
		SortSeg sg; sg.off = off; sg.len = (u32)len; sg.shift = 56;
		klib[s] = sg;
		atomicAdd(&cnt[LQ_C_LEN0 + lq_walk_class((u32)len, caps)], 1u);
	} else {
		PSeg sg; sg.off = off; sg.len = (u32)len; sg.rem = (u8)(km.pbits + km.rbits + 1); sg.buf = 0; sg.nbits = 0; sg.pad = 0;
		lq_ps_route(sg, L, LQ_P_BIG0);
	}
}

// qklib[q] = 1: query q goes through klib's passes (marked minimizers -- or every query, LQCOV_SORT=klib -- and more than 64 anchors)
__global__ void k_query_klib(const u64 *aq_off, const u32 *qdirty, u32 n_q, int all_klib, u32 *qklib)
{
	const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q < n_q) qklib[q] = (qdirty[q] || all_klib) && aq_off[q + 1] - aq_off[q] > LQ_RS_MIN ? 1u : 0u;
}

// one wave per 64 buckets of a finished klib pass over records (Rn = the arrangement the pass left): recurse, hand over,
// or finish (ksort.h:121-128).
// Buckets of > 64 elements that received fewer than two marked anchors hold no equal x: they leave klib's passes for the
// parallel sort above, which gathers their anchors through the records (buf = 2 + rb); the others become next-level
// sub-arrays.  Buckets of <= 64 elements are finished by klib's insertion sort (ksort.h:87-97), which is stable, so its
// result is the unique stable order by x: the wave gathers the anchors of whole buckets (adjacent in memory, packed into
// chunks of <= 64 elements, one element per lane), every lane ranks its element among the elements of its own bucket (ties
// by position in the arrangement) and writes it to its final place in A.  After the pass on byte 0 everything is final.
// Round 6, second pass only (want != null): klib's order is needed in the listed runs and nowhere else -- the second pass chains
// nothing but those -- so a bucket of a level on strand / rid (shift >= 32: its anchors share the bits of x >> 32 above the level's
// byte... and that byte) that holds no listed run's (strand, rid) is DROPPED: no next level, no parallel sort, no insertion sort.
// What the walk of ITS level needed -- every element of the parent sub-array -- has been walked; nothing below depends on a dropped
// bucket (a level's walk only reads its own sub-array, ksort.h:99-129).  At configs[2] a query's ~40 listed runs sit in ~40 of the
// ~4000 buckets of the rid >> 8 level: the levels below shrink a hundredfold.  The dropped bucket's place in A is filled with its
// smallest possible key, so that A stays ascending in x >> 32 per query and k_want_runs' bisections still find the listed runs.
struct PruneWant { const unsigned long long *want; u32 n_want; const u64 *sub_off; const u32 *sub_q; u32 n_sub; };
#define LQ_CHILD_THREADS 64
__global__ void __launch_bounds__(LQ_CHILD_THREADS)
k_rs_children(const SortSeg *segs, const u32 *n_segs_p, const RRec *Rn, u32 rb, const mm128 *O, mm128 *A, const u32 *hist, const u32 *mhist, const u32 *begs,
              SortSeg *next, u32 *n_next, u32 const_levels, PsLists L, KeyMap km, int all_klib, u32 *n_tiles_zero, u32 *len_cnt, WalkCaps caps, PruneWant pw)
{
	__shared__ u64 xs[64];
	__shared__ u32 flag[64];
	const u32 n_segs = *n_segs_p;
	const u32 lane = threadIdx.x;
	if (blockIdx.x == 0 && lane == 0) *n_tiles_zero = 0;          // the tile list is done with for this level: the next k_sort_tiles counts from zero
	for (u64 w = blockIdx.x; w < (u64)n_segs * 4; w += gridDim.x) {
		const u64 t = w * 64 + lane;
		const u32 sgi = (u32)(t >> 8);
		const SortSeg sg = segs[sgi];
		u32 n = hist[t];
		const u32 bg = begs[t];
		if (pw.want && sg.shift >= 32) {                          // (uniform: a wave's 64 buckets belong to one sub-array)
			u32 qlo = 0, qhi = pw.n_sub;                          // the sub-array's query: the last one whose anchors start at or before it
			while (qhi - qlo > 1) { const u32 mid = qlo + ((qhi - qlo) >> 1); if (pw.sub_off[mid] <= sg.off) qlo = mid; else qhi = mid; }
			const unsigned long long qk = (unsigned long long)pw.sub_q[qlo] << 32;
			bool drop = false; u32 klo = 0;
			if (n) {
				const u32 key = Rn[sg.off + bg].key, m = sg.shift > 32 ? (1u << (sg.shift - 32)) - 1 : 0u;
				klo = key & ~m;
				const unsigned long long a = qk | klo, b = qk | (key | m);
				u32 lo = 0, hi = pw.n_want;
				while (lo < hi) { const u32 mid = lo + ((hi - lo) >> 1); if (pw.want[mid] < a) lo = mid + 1; else hi = mid; }
				drop = !(lo < pw.n_want && pw.want[lo] <= b);
			}
			u64 dm = __ballot(drop);
			while (dm) {                                              // the wave fills a dropped bucket's place together
				const u32 f = (u32)__builtin_ctzll(dm);
				const u32 fb = (u32)__builtin_amdgcn_readlane((int)bg, (int)f), fn = (u32)__builtin_amdgcn_readlane((int)n, (int)f), fk = (u32)__builtin_amdgcn_readlane((int)klo, (int)f);
				mm128 e; e.x = (u64)fk << 32; e.y = 0;
				for (u32 i = lane; i < fn; i += 64) A[sg.off + fb + i] = e;
				dm &= dm - 1;
			}
			if (drop) n = 0;
		}
		if (n > LQ_RS_MIN) {
			if (sg.shift == 0 || (mhist[t] < 2 && !all_klib)) {
				// (after the pass on byte 0 a bucket holds one x: nothing left to sort, the parallel sort's finish brings it home)
				PSeg ch; ch.off = sg.off + bg; ch.len = n; ch.rem = sg.shift == 0 ? 0 : (u8)lq_rem_below(sg.shift, km); ch.buf = (u8)(2 + rb); ch.nbits = 0; ch.pad = 0;
				lq_ps_route(ch, L, LQ_P_BIG0);
			} else {
				const u32 s = atomicAdd(n_next, 1u);
				// the next digit that can differ: levels whose byte is the same in every anchor of the part (bits of rid above the
				// target count, bits of the position above the longest target) are identity passes in klib (one bucket holds the
				// whole sub-array, which is recursed into unchanged: ksort.h:121-128) and are stepped over
				u32 sh = sg.shift - 8;
				while (sh > 0 && (const_levels >> (sh >> 3) & 1)) sh -= 8;
				SortSeg c; c.off = sg.off + bg; c.len = n; c.shift = sh;
				next[s] = c;
				atomicAdd(&len_cnt[lq_walk_class(n, caps)], 1u);       // (the host sizes the next level's walker launches by these, and skips the empty ones)
			}
		}
		const RRec *rseg = Rn + sg.off;
		mm128 *seg = A + sg.off;
		u64 todo = __ballot(n >= 1 && n <= LQ_RS_MIN);
		while (todo) {                                            // uniform: one chunk of whole buckets per turn
			const u32 f = (u32)__builtin_ctzll(todo);             // first bucket still to finish
			const u32 base = __builtin_amdgcn_readlane(bg, f);
			const u64 fit = __ballot(lane >= f && bg + n - base <= 64);   // bg + n grows with the lane: a run of lanes starting at f
			const u32 e = 64 - (u32)__builtin_clzll(fit);
			const u32 total = __builtin_amdgcn_readlane(bg + n, e - 1) - base;
			const bool member = (fit >> lane) & 1;
			flag[lane] = 0;
			__syncthreads();
			if (member && n) flag[bg - base] = 1;
			__syncthreads();
			const u64 M = __ballot(flag[lane] != 0);              // bit i: a bucket starts at element i of the chunk
			const u32 i = lane;
			const bool act = i < total;
			const u32 lo = 63 - (u32)__builtin_clzll((M & (~0ULL >> (63 - i))) | 1ULL);
			const u64 above = i == 63 ? 0 : (M >> (i + 1)) << (i + 1);
			const u32 hi = above ? (u32)__builtin_ctzll(above) : total;
			const u32 myn = act ? hi - lo : 0;
			mm128 el; el.x = 0; el.y = 0;
			if (act) { el = O[LQ_R_IDX(rseg[base + i].im)]; xs[i] = el.x; }
			__syncthreads();
			u32 rk = 0;
			for (u32 jj = 0; jj < myn; ++jj) {
				const u32 j = lo + jj;
				const u64 xj = xs[j];
				rk += j < i ? (xj <= el.x) : (xj < el.x);
			}
			__syncthreads();
			if (act) seg[base + lo + rk] = el;
			todo &= ~fit;
		}
	}
}
