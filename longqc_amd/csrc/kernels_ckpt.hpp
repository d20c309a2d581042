// longqc_amd/csrc/kernels_ckpt.hpp -- cutting one long klib token walk into segments that run side by side.
//
// A general pass of klib's radix sort (ksort.h:99-129) over one sub-array is a serial token walk (kernels_sort.hpp,
// kernels_walk.hpp), ~120 ns per element: the (query, strand) sub-arrays of the longest queries hold millions of anchors
// and their top pass alone would take most of a second.  The walk cannot be split by looking at the elements -- but its
// state at chosen moments can be computed without walking:
//
//   The outer loop of the pass fills the slots in ascending order; consider the moment it is about to look at slot s of
//   bucket k (buckets < k full, the token at rest).  Let A_c be the cursor of bucket c then.  Exactly the first
//   A_c - beg_c elements of every region R_c have been picked up, and every picked-up element sits in its bucket, so
//       A_c - beg_c = #{picked-up elements with digit c} = sum over regions l of n_{l,c}(A_l - beg_l)   for c > k,
//   with A_c = end_c for c < k and A_k = s, where n_{l,c}(m) counts digit c among the first m elements of R_l.  The
//   right-hand side is monotone in A.  The walk's own state solves the system, and the walk can never get ahead of ANY
//   solution A' that lies above an earlier state of it: look at the first moment a cursor would pass A'_c -- the element
//   arriving at c was picked up from the first A'_l - beg_l elements of some R_l and has digit c, and A'_c already counts
//   all of those.  Hence the state is the LEAST solution above the state at the end of the previous phase (bucket k-1
//   full) -- and, by the same argument, above the pass's start state: a state can be found from scratch, without the
//   states before it (k_ck_phases, k_ck_solve: every phase end and every checkpoint is found independently).  One caveat: only slots the outer loop really looks at name a
//   state.  Arrivals fill the head of bucket k before its phase begins; a slot s below the cursor that bucket k has
//   when the buckets before it are full is never looked at, and "k held at s, or further if its arrivals say so"
//   (A_k >= s and A_k - beg_k >= arrivals) then yields the state at the bucket's first look.
//   Monotone iteration from a state below (A <- max(A, F(A))) finds the least solution: no walking, only prefix counts of digits
//   (per-tile histograms, scanned) and a few dozen rounds -- the increments shrink by about (B-1)/B per round, so this is
//   for passes with few buckets (B <= 16: the byte of rid above 65536 targets), which is where the giant walks are.
//
// So: k_ck_tilehist + k_ck_tilescan build the prefix counts of every long sub-array, k_ck_phases finds the state at the
// end of every phase, k_ck_solve the state at evenly weighted checkpoints in between, and k_sort_walk_ck runs one walker
// per checkpoint (kernels_walk.hpp) from its state up to the next checkpoint's slot.  Every element is moved by exactly
// one walker, with the destination the single serial walk gives it.
#pragma once
#include "lq_common.hpp"
#include "kernels_sort.hpp"

#define LQ_CK_B 16                 // buckets a checkpointed pass may have (digits 0..15)
#define LQ_CK_TILE 1024            // elements per prefix-count tile


// The level's checkpointed sub-arrays (the longest size classes of k_sort_classify's lists), their prefix tiles and their
// checkpoints, laid out by one block on the device: ckn = [checkpoints in all, sub-arrays, prefix tiles].  The kernels below
// are launched with grids sized by upper bounds and read the real counts here -- no host round trip inside a level.
__global__ void __launch_bounds__(256)
k_ck_plan(const SortSeg *segs, const u32 *walk_list, u32 list_cap, const u32 *n_walk, int use3, u32 unit, u32 max_ck, CkSeg *out, u32 cap_cks, u32 *ckn)
{
	__shared__ u32 st[256], sc[256], tmp[256], tot_t, tot_c, base_t, base_c;
	const u32 t = threadIdx.x;
	const u32 n3 = use3 ? n_walk[3] : 0, n4 = n_walk[4];
	const u32 n = n3 + n4 < cap_cks ? n3 + n4 : cap_cks;        // (the bound holds by construction: every listed sub-array is longer than the class limit)
	if (t == 0) { base_t = 0; base_c = 0; }
	__syncthreads();
	for (u32 s0 = 0; s0 < n; s0 += 256) {
		const u32 i = s0 + t;
		u32 id = 0, nt = 0, nc = 0;
		if (i < n) {
			id = i < n3 ? walk_list[(u64)3 * list_cap + i] : walk_list[(u64)4 * list_cap + (i - n3)];
			const u32 len = segs[id].len;
			nc = len / unit; if (nc < 2) nc = 2; if (nc > max_ck) nc = max_ck;
			nt = len / LQ_CK_TILE + 1;
		}
		st[t] = nt; sc[t] = nc;
		__syncthreads();
		lq_scan256(st, tmp, &tot_t);
		lq_scan256(sc, tmp, &tot_c);
		if (i < n) { CkSeg c; c.sgi = id; c.tile0 = base_t + st[t]; c.ck0 = base_c + sc[t]; c.n_ck = nc; out[i] = c; }
		__syncthreads();
		if (t == 0) { base_t += tot_t; base_c += tot_c; }
		__syncthreads();
	}
	if (t == 0) { ckn[0] = base_c; ckn[1] = n; ckn[2] = base_t; }
}

// digit counts of every tile (raw), strided over all tiles of all listed sub-arrays
__global__ void __launch_bounds__(256)
k_ck_tilehist(const CkSeg *cks, const u32 *ckn, const SortSeg *segs, const u8 *D, u32 *T)
{
	__shared__ u32 lh[LQ_CK_B];
	const u32 t = threadIdx.x;
	const u32 n_cks = ckn[1], n_tiles = ckn[2];
	for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		u32 lo = 0, hi = n_cks;
		while (hi - lo > 1) { const u32 mid = lo + ((hi - lo) >> 1); if (cks[mid].tile0 <= tile) lo = mid; else hi = mid; }
		const CkSeg ck = cks[lo];
		const SortSeg sg = segs[ck.sgi];
		const u32 i0 = (tile - ck.tile0) * LQ_CK_TILE, i1 = i0 + LQ_CK_TILE < sg.len ? i0 + LQ_CK_TILE : sg.len;
		if (t < LQ_CK_B) lh[t] = 0;
		__syncthreads();
		const u8 *d = D + sg.off;
		for (u32 i = i0 + t; i < i1; i += 256) atomicAdd(&lh[d[i] & (LQ_CK_B - 1)], 1u);
		__syncthreads();
		if (t < LQ_CK_B) T[(u64)tile * LQ_CK_B + t] = lh[t];
		__syncthreads();
	}
}

// exclusive scan of the tile counts along each sub-array: T[tile][d] = count of digit d before the tile
__global__ void __launch_bounds__(256)
k_ck_tilescan(const CkSeg *cks, const u32 *ckn, const SortSeg *segs, u32 *T)
{
	__shared__ u32 part[256][LQ_CK_B + 1];
	const u32 t = threadIdx.x;
	const u32 n_cks = ckn[1];
	for (u32 j = blockIdx.x; j < n_cks; j += gridDim.x) {
		const CkSeg ck = cks[j];
		const u32 nt = segs[ck.sgi].len / LQ_CK_TILE + 1;        // one more than needed: the last entry holds the totals
		const u32 per = (nt + 255) / 256, a = t * per < nt ? t * per : nt, b = a + per < nt ? a + per : nt;
		u32 acc[LQ_CK_B];
		for (int d = 0; d < LQ_CK_B; ++d) acc[d] = 0;
		for (u32 x = a; x < b; ++x) for (int d = 0; d < LQ_CK_B; ++d) acc[d] += T[(u64)(ck.tile0 + x) * LQ_CK_B + d];
		for (int d = 0; d < LQ_CK_B; ++d) part[t][d] = acc[d];
		__syncthreads();
		if (t < LQ_CK_B) { u32 run = 0; for (u32 x = 0; x < 256; ++x) { const u32 v = part[x][t]; part[x][t] = run; run += v; } }
		__syncthreads();
		for (int d = 0; d < LQ_CK_B; ++d) acc[d] = part[t][d];
		for (u32 x = a; x < b; ++x) for (int d = 0; d < LQ_CK_B; ++d) { u32 *p = &T[(u64)(ck.tile0 + x) * LQ_CK_B + d]; const u32 v = *p; *p = acc[d]; acc[d] += v; }
		__syncthreads();
	}
}

// One wave solves for states; lane c < 16 owns bucket c.  lq_ck_prefix: counts of digit `lane` in the first x elements of
// the sub-array = tile table + the partial tile.  The partial tile (< 1024 digits) is read as aligned 16-byte words, one per
// lane (one more for lane 0 if the alignment asks for a 65th), counted into sixteen 16-bit fields in registers and summed
// over the wave by shuffles: no LDS, no barrier -- the look-ups of one round are independent loads.
__device__ __forceinline__ u32 lq_ck_prefix(const u8 *d, const u32 *T, u32 x, u32 lane)
{
	const u32 tile = x / LQ_CK_TILE, r0 = tile * LQ_CK_TILE;
	const u8 *p_lo = d + r0, *p_hi = d + x;
	const u8 *a0 = (const u8*)((size_t)p_lo & ~(size_t)15);
	const u32 n_words = (u32)((size_t)(p_hi - a0 + 15) >> 4);
	u64 f0 = 0, f1 = 0, f2 = 0, f3 = 0;
	for (u32 w = lane; w < n_words; w += 64) {
		const u8 *wa = a0 + (size_t)w * 16;
		const uint4 W = *(const uint4*)wa;
		const u32 ww[4] = { W.x, W.y, W.z, W.w };
#pragma unroll
		for (u32 k = 0; k < 16; ++k) {
			const u8 *g = wa + k;
			if (g >= p_lo && g < p_hi) {
				const u32 dg = (ww[k >> 2] >> ((k & 3) * 8)) & (LQ_CK_B - 1);
				const u64 inc = 1ULL << ((dg & 3) * 16);
				const u32 q = dg >> 2;
				f0 += q == 0 ? inc : 0; f1 += q == 1 ? inc : 0; f2 += q == 2 ? inc : 0; f3 += q == 3 ? inc : 0;
			}
		}
	}
	for (int o = 32; o > 0; o >>= 1) { f0 += __shfl_xor(f0, o); f1 += __shfl_xor(f1, o); f2 += __shfl_xor(f2, o); f3 += __shfl_xor(f3, o); }
	u32 v = 0;
	if (lane < LQ_CK_B) {
		const u32 q = lane >> 2;
		const u64 f = q == 0 ? f0 : q == 1 ? f1 : q == 2 ? f2 : f3;
		v = T[(u64)tile * LQ_CK_B + lane] + (u32)((f >> ((lane & 3) * 16)) & 0xffff);
	}
	return v;
}

// least solution above the state in `A` (lane c: cursor of bucket c, absolute slot index in the sub-array) with the
// buckets below k full and bucket k held at its value; pbeg[l] = prefix counts (of this lane's digit) at beg[l]
__device__ __forceinline__ u32 lq_ck_iterate(const u8 *d, const u32 *T, u32 k, u32 nb, u32 A, const u32 (&pbeg)[LQ_CK_B], const u32 (&pend)[LQ_CK_B],
                                            u32 my_beg, u32 lane)
{
	for (;;) {
		u32 acc = 0;
#pragma unroll
		for (u32 l = 0; l < LQ_CK_B; ++l) {                     // uniform loop over the regions
			if (l >= nb) break;
			const u32 x = __shfl(A, (int)l);
			if (l < k) acc += pend[l] - pbeg[l];                    // a full region has given everything it has of this digit
			else acc += lq_ck_prefix(d, T, x, lane) - pbeg[l];
		}
		u32 nA = A;
		if (lane < nb && lane > k) { const u32 f = my_beg + acc; if (f > A) nA = f; }
		const u64 changed = __ballot(nA != A);
		A = nA;
		if (!changed) break;
	}
	return A;
}

// state at the end of every phase: E[j][k][c] = cursors when bucket k has just become full.  One wave per (sub-array, phase):
// every phase end is found from scratch (the least solution above the start state with the buckets up to k full) instead
// of from the phase before -- the phases of a sub-array then cost the time of the slowest, not their sum.
__global__ void __launch_bounds__(64)
k_ck_phases(const CkSeg *cks, const u32 *ckn, const SortSeg *segs, const u8 *D, const u32 *hist, const u32 *begs, const u32 *T, u32 *E)
{
	const u32 lane = threadIdx.x;
	const u32 n_cks = ckn[1];
	for (u32 wi = blockIdx.x; wi < n_cks * LQ_CK_B; wi += gridDim.x) {
		const u32 j = wi / LQ_CK_B, k = wi % LQ_CK_B;
		const CkSeg ck = cks[j];
		const SortSeg sg = segs[ck.sgi];
		const u8 *d = D + sg.off;
		const u32 *Tj = T + (u64)ck.tile0 * LQ_CK_B;
		const u32 *bg = begs + (u64)ck.sgi * 256, *cn = hist + (u64)ck.sgi * 256;
		const u32 my_beg = lane < LQ_CK_B ? bg[lane] : 0, my_end = lane < LQ_CK_B ? my_beg + cn[lane] : 0;
		u32 nb = 1;
		for (u32 c = 0; c < LQ_CK_B; ++c) if (cn[c]) nb = c + 1;      // buckets in use
		u32 A = lane <= k ? my_end : my_beg;                        // the outer loop has filled the buckets up to k
		if (k < nb) {
			u32 pbeg[LQ_CK_B], pend[LQ_CK_B];
			for (u32 l = 0; l < LQ_CK_B; ++l) { pbeg[l] = 0; pend[l] = 0; }
#pragma unroll
			for (u32 l = 0; l < LQ_CK_B; ++l) if (l < nb) { pbeg[l] = lq_ck_prefix(d, Tj, bg[l], lane); pend[l] = lq_ck_prefix(d, Tj, bg[l] + cn[l], lane); }
			A = lq_ck_iterate(d, Tj, k, nb, A, pbeg, pend, my_beg, lane);
		} else A = my_end;                                          // past the last bucket in use: the final state
		if (lane < LQ_CK_B) E[((u64)j * LQ_CK_B + k) * LQ_CK_B + lane] = A;
	}
}

// state at every checkpoint (one wave each): checkpoint i of sub-array j sits at slot s of bucket k
__global__ void __launch_bounds__(64)
k_ck_solve(const CkSeg *cks, const u32 *ckn, const SortSeg *segs, const u8 *D, const u32 *hist, const u32 *begs, const u32 *T,
           const u32 *E, u32 *S, u32 *CKS)
{
	const u32 lane = threadIdx.x;
	const u32 n_ck_total = ckn[0], n_cks = ckn[1];
	for (u32 ci = blockIdx.x; ci < n_ck_total; ci += gridDim.x) {
		u32 lo = 0, hi = n_cks;
		while (hi - lo > 1) { const u32 mid = lo + ((hi - lo) >> 1); if (cks[mid].ck0 <= ci) lo = mid; else hi = mid; }
		const u32 j = lo;
		const CkSeg ck = cks[j];
		const u32 i = ci - ck.ck0;
		const SortSeg sg = segs[ck.sgi];
		const u8 *d = D + sg.off;
		const u32 *Tj = T + (u64)ck.tile0 * LQ_CK_B;
		const u32 *bg = begs + (u64)ck.sgi * 256, *cn = hist + (u64)ck.sgi * 256;
		const u32 my_beg = lane < LQ_CK_B ? bg[lane] : 0;
		// slot of this checkpoint: the checkpoints are shared out over the phases by the work each phase does (the elements
		// picked up during it: known from the phase-end states) and spread evenly over the slots the phase looks at
		u64 wtot = 0, wk_[LQ_CK_B];
		{
			u32 prev = 0;
			for (u32 c = 0; c < LQ_CK_B; ++c) prev += bg[c];         // (sum of the starting cursors)
			u64 before = prev;
			for (u32 q = 0; q < LQ_CK_B; ++q) {
				u64 sum = 0;
				for (u32 c = 0; c < LQ_CK_B; ++c) sum += E[((u64)j * LQ_CK_B + q) * LQ_CK_B + c];
				wk_[q] = sum - before; before = sum; wtot += wk_[q];
			}
		}
		u64 want = wtot / ck.n_ck * i + wtot % ck.n_ck * i / ck.n_ck;
		u32 k = 0, s = 0;
		for (k = 0; k < LQ_CK_B; ++k) {
			if (want < wk_[k] || k == LQ_CK_B - 1) {
				const u32 a0 = k ? E[((u64)j * LQ_CK_B + (k - 1)) * LQ_CK_B + k] : bg[k], e0 = bg[k] + cn[k];   // slots the phase looks at: [cursor at its start, end)
				const u64 span = e0 > a0 ? e0 - a0 : 0;
				s = a0 + (u32)(wk_[k] ? span * want / wk_[k] : 0);
				if (s > e0) s = e0;
				break;
			}
			want -= wk_[k];
		}
		if (i == 0) { k = 0; s = 0; }
		u32 A;
		if (k == 0) A = my_beg;
		else A = lane < LQ_CK_B ? E[((u64)j * LQ_CK_B + (k - 1)) * LQ_CK_B + lane] : 0;
		// the outer loop only ever looks at slots from bucket k's cursor on: an earlier target slot means "its first look"
		const u32 ak = (u32)__builtin_amdgcn_readlane((int)A, (int)k);
		if (s < ak) s = ak;
		if (lane == k) A = s;
		u32 nb = 1;
		for (u32 c = 0; c < LQ_CK_B; ++c) if (cn[c]) nb = c + 1;
		u32 pbeg[LQ_CK_B], pend[LQ_CK_B];
		for (u32 l = 0; l < LQ_CK_B; ++l) { pbeg[l] = 0; pend[l] = 0; }
#pragma unroll
		for (u32 l = 0; l < LQ_CK_B; ++l) if (l < nb) { pbeg[l] = lq_ck_prefix(d, Tj, bg[l], lane); pend[l] = lq_ck_prefix(d, Tj, bg[l] + cn[l], lane); }
		if (k < nb) A = lq_ck_iterate(d, Tj, k, nb, A, pbeg, pend, my_beg, lane);
		if (lane < LQ_CK_B) S[(u64)ci * LQ_CK_B + lane] = A;
		if (lane == 0) CKS[ci] = s;
	}
}

// ---- passes with many buckets (up to 256): the same states by following the elements in bulk ----------------------
// With B buckets the rounds of the iteration above number about B ln(mass) and each would cost B prefix look-ups.  Here
// the solver keeps, per bucket, the cursor A_c up to which the region has been read and the number of elements seen so
// far with digit c (arr[c], in LDS); "bucket c has A_c - beg_c < arr[c]" means arrivals are waiting: the lane that owns c
// reads the next elements of R_c and counts their digits.  Any order of doing that ends in the same least fixed point
// (chaotic iteration of a monotone system), and every element is read once per sub-array.  One wave walks the outer
// loop's slots in steps sized by the yield of the previous step (slots of the phase's bucket worth about len / n_ck
// picked-up elements: a bigger step costs only logarithmically more rounds), and writes a checkpoint -- all 256 cursors and
// the slot -- after each: the pieces come out balanced whatever the data looks like.
#define LQ_CKW_STEP 16
// arr[digit] += 1 for the elements [lo, hi) of the sub-array, read as aligned 16-byte words (a lane's backlog of hundreds of
// elements costs a sixteenth of the load latencies it would byte by byte).  The last word read stays in registers (cw, tag
// cwa = its address): in the long tail of a fixed point a bucket takes in an element or two per round, and sixteen of them
// then cost one load.
__device__ __forceinline__ void lq_ck_count_range(const u8 *d, u32 lo, u32 hi, u32 *arr, uint4 &cw, size_t &cwa)
{
	const u8 *p_lo = d + lo, *p_hi = d + hi;
	for (const u8 *wa = (const u8*)((size_t)p_lo & ~(size_t)15); wa < p_hi; wa += 16) {
		if ((size_t)wa != cwa) { cw = *(const uint4*)wa; cwa = (size_t)wa; }
		const u32 ww[4] = { cw.x, cw.y, cw.z, cw.w };
#pragma unroll
		for (u32 k = 0; k < 16; ++k) {
			const u8 *g = wa + k;
			if (g >= p_lo && g < p_hi) atomicAdd(&arr[(ww[k >> 2] >> ((k & 3) * 8)) & 0xff], 1u);
		}
	}
}
// (Measured and dropped, round 3: the chain cut into coarse parts that start from states found from scratch -- at configs[2]
// the solver took 818 ms per step against 384 ms for the serial chain: every part reads the buckets before its slot whole.)
__global__ void __launch_bounds__(64)
k_ck_chain256(const CkSeg *cks, const u32 *ckn, const SortSeg *segs, const u8 *D, const u32 *hist, const u32 *begs, u32 *S, u32 *CKS)
{
	__shared__ u32 arr[256];
	__shared__ u32 red[64];
	const u32 lane = threadIdx.x;
	const u32 n_cks = ckn[1];
	for (u32 j = blockIdx.x; j < n_cks; j += gridDim.x) {
		const CkSeg ck = cks[j];
		const u32 my_ck0 = ck.ck0, my_n = ck.n_ck;
		const SortSeg sg = segs[ck.sgi];
		const u8 *d = D + sg.off;
		const u32 *bg = begs + (u64)ck.sgi * 256, *cn = hist + (u64)ck.sgi * 256;
		u32 A[4], B0[4], E0[4];
		uint4 cw[4]; size_t cwa[4];
		for (int g = 0; g < 4; ++g) { cw[g] = uint4{0, 0, 0, 0}; cwa[g] = 0; }
		for (int g = 0; g < 4; ++g) { const u32 c = (u32)g * 64 + lane; B0[g] = bg[c]; E0[g] = B0[g] + cn[c]; A[g] = B0[g]; arr[c] = 0; }
		__syncthreads();
		// fixed point with the buckets before k full and bucket k held at s
#define LQ_CK_FIXED_POINT(k, s) \
		for (;;) { \
			bool pending = false; \
			for (int g = 0; g < 4; ++g) { \
				const u32 c = (u32)g * 64 + lane; \
				u32 need = c < (k) ? E0[g] : B0[g] + arr[c]; \
				if (c == (k) && need < (s)) need = (s);            /* held at s -- or further, where arrivals filled the bucket beyond s before its phase began */ \
				if (need > E0[g]) need = E0[g]; \
				if (A[g] < need) { lq_ck_count_range(d, A[g], need, arr, cw[g], cwa[g]); A[g] = need; pending = true; } \
			} \
			__syncthreads(); \
			if (!__ballot(pending)) break; \
		}
#define LQ_CK_PICKED(out) do { \
			u32 mine_ = 0; \
			for (int g = 0; g < 4; ++g) mine_ += A[g] - B0[g]; \
			red[lane] = mine_; \
			__syncthreads(); \
			u64 p_ = 0; \
			for (u32 x = 0; x < 64; ++x) p_ += red[x]; \
			__syncthreads(); \
			(out) = p_; \
		} while (0)
		const u64 target = (u64)sg.len / ck.n_ck + 1;
		u64 picked_at_last = 0, picked_before = 0;
		LQ_CK_PICKED(picked_before);
		picked_at_last = picked_before;
		u32 n_out = 0, est = LQ_CKW_STEP;                       // slots of the phase's bucket that are worth about `target` picked-up elements
		// first checkpoint: the start state
		for (int g = 0; g < 4; ++g) S[((u64)my_ck0 + 0) * 256 + (u32)g * 64 + lane] = A[g];
		if (lane == 0) CKS[my_ck0] = 0;
		n_out = 1;
		for (u32 k = 0; k < 256; ++k) {                          // phases of the outer loop
			const u32 kl = k & 63, kg = k >> 6;
			u32 ek = 0, ak = 0;
			for (int g = 0; g < 4; ++g) if ((u32)g == kg) { ek = (u32)__builtin_amdgcn_readlane((int)E0[g], (int)kl); ak = (u32)__builtin_amdgcn_readlane((int)A[g], (int)kl); }
			while (ak < ek) {
				const u32 step = est;
				u32 s = ek - ak > step ? ak + step : ek;                // the outer loop reaches slot s of bucket k
				LQ_CK_FIXED_POINT(k, s)
				// bucket k's cursor: slots filled = held value, unless arrivals already pushed it further
				for (int g = 0; g < 4; ++g) if ((u32)g == kg) ak = (u32)__builtin_amdgcn_readlane((int)A[g], (int)kl);
				if (ak < s) ak = s;
				u64 picked;
				LQ_CK_PICKED(picked);
				{	// one step per checkpoint: the next step covers the slots that the last one's yield says are worth `target`
					const u64 got = picked - picked_before;
					picked_before = picked;
					u64 e2 = got ? (u64)step * target / got : (u64)step * 4;
					if (e2 > (u64)step * 4) e2 = (u64)step * 4;
					est = (u32)(e2 < LQ_CKW_STEP ? LQ_CKW_STEP : e2 > (1u << 24) ? (1u << 24) : e2);
				}
				if (picked - picked_at_last >= target / 2 && n_out < my_n && ak < ek) {
					for (int g = 0; g < 4; ++g) S[((u64)my_ck0 + n_out) * 256 + (u32)g * 64 + lane] = A[g];
					if (lane == 0) CKS[my_ck0 + n_out] = ak;
					++n_out; picked_at_last = picked;
				}
			}
		}
#undef LQ_CK_FIXED_POINT
#undef LQ_CK_PICKED
		// unused checkpoints: the final state (their walkers find nothing to do)
		for (; n_out < my_n; ++n_out) {
			for (int g = 0; g < 4; ++g) S[((u64)my_ck0 + n_out) * 256 + (u32)g * 64 + lane] = E0[g];
			if (lane == 0) CKS[my_ck0 + n_out] = sg.len;
		}
		__syncthreads();
	}
}
