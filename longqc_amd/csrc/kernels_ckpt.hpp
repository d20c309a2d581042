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
k_ck_plan(const SortSeg *segs, const u32 *walk_list, u32 list_cap, const u32 *n_walk, int use3, u32 unit, u32 max_ck, u32 quantum /* checkpoints per sub-array: a multiple of it */, CkSeg *out, u32 cap_cks, u32 *ckn)
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
			nc = (nc + quantum - 1) / quantum * quantum;
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
// far with digit c (arr[c], in LDS); "bucket c has A_c - beg_c < arr[c]" means arrivals are waiting: the thread that owns c
// reads the next elements of R_c and counts their digits.  Any order of doing that ends in the same least fixed point
// (chaotic iteration of a monotone system).
//
// What that costs is not bytes but depth: one look of the outer loop at a slot of bucket k sets off a cascade that dies out
// only when elements of digit k turn up -- about (256 - k) ln(mass) rounds, whatever the step.  Round 5's form (one wave per
// sub-array, every phase of the outer loop solved after the one before: 256 cascades at the very least) took 25-78 ms per
// launch for 8 % of a step's seed hits.  Three things cut the depth (round 6):
//   * A state can be found from scratch (the least solution above the START state with the buckets before k full and k held
//     at s -- kernels_ckpt.hpp's opening comment), so a sub-array's checkpoints are shared out over SUB-CHAINS, one block
//     each: sub-chain i finds its first state from scratch -- the regions before its bucket counted by the whole block, then
//     one cascade -- and follows the walk from there to the first state of sub-chain i + 1, which it reaches exactly (the
//     same least solution), writing up to LQ_CKM_Q checkpoints on the way; quota it does not use repeats that last state
//     (a walker that starts there stops at once).
//   * Nearly all of a pass's work is done while the outer loop looks at the FIRST bucket in use, k0 (every cycle runs until
//     it finds one of its rare digits; when its last one is placed most regions are read to their tails): the sub-chains
//     start at evenly spaced slots of k0, and a few at the first look at buckets k0 + 1, k0 + 2, k0 + 4, ...
//   * After k0 a phase does little: the chain does not stop at every bucket but asks for "the buckets before k + jump full"
//     in one cascade, jump doubling while the yield stays below a checkpoint's worth of elements.
// One thread per bucket (256), so a round is one LDS look, at most a few counted elements and a barrier.
#define LQ_CKW_STEP 16
#define LQ_CKM_THREADS 256
#define LQ_CKM_Q 8                 // checkpoints per sub-chain: CkSeg.n_ck of a many-bucket pass is a multiple of it.  A sub-array gets twice the
                                   // checkpoints its length asks for (len / unit): a sub-chain whose stretch of the walk holds twice the average still
                                   // cuts it into pieces of the aimed-at size, and quota not used costs a walker that starts and stops
#ifndef LQ_CKM_SPINS
#define LQ_CKM_SPINS 8             // looks at its bucket a thread takes between two barriers of the fixed point (4: 15.2 ms, 8: 14.3 ms for 400 sub-arrays of 440 k)
#endif
// A round of the fixed point must not wait for global memory (a first form did: 2400 cycles per round, an L2 round trip -- with 64
// buckets per wave some lane crosses a 16-byte word in nearly every round, and a wave's loads are waited for together, so a word
// asked for ahead of time by one lane is only as early as the latest request of any lane).  So every bucket has a WINDOW of its
// digit stream in LDS -- the LQ_CKM_WINB bytes from its cursor on, filled by its thread when a fixed point begins (its loads in
// flight together, one wait per solve, nobody else reads them: no barrier) -- and the rounds read that; what a fixed point takes in beyond
// the window comes from global memory, 64 bytes per trip.  The block's barrier is the bare instruction behind a wait for LDS only.
#ifndef LQ_CKM_WINW
#define LQ_CKM_WINW 4              // (64 bytes: a fixed point takes in ~32 elements per bucket.  With 128 the LDS of a block allowed four blocks per CU instead of eight: 19.9 against 14.3 ms)
#endif
#define LQ_CKM_WINB (LQ_CKM_WINW * 16)
#ifdef LQ_EMU
#define LQ_CKM_BARRIER() __syncthreads()
#else
#define LQ_CKM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// (Two more things the first forms paid for, seen in the ISA: a `volatile` read of arr[] through a generic pointer is a FLAT load with
// system scope -- hundreds of cycles per look --, and HIP's uint4 is a union with an array, so picking a digit of it with a run-time
// index moved the word to scratch memory.  Hence the workgroup-scope atomic load below and the two 64-bit halves.)
#ifdef LQ_EMU
#define LQ_LDS_LOOK(p) (*(volatile u32*)(p))
#else
#define LQ_LDS_LOOK(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif
struct CkmWin { u64 lo, hi; u32 cwi, wsi; };      // the 16-byte word the cursor is in (two halves), its index; the index of the window's first word.
                                                  // Word i = the 16 bytes at dA + 16 i, dA = the sub-array's digits rounded down to 16 bytes
__device__ __forceinline__ void lq_ckm_count16(u64 lo, u64 hi, u32 k0, u32 k1, u32 *arr)   // digits k0 .. k1 - 1 of a word
{
	if (k0 == 0 && k1 == 16) {
#pragma unroll
		for (u32 k = 0; k < 8; ++k) atomicAdd(&arr[(u32)(lo >> (8 * k)) & 0xff], 1u);
#pragma unroll
		for (u32 k = 0; k < 8; ++k) atomicAdd(&arr[(u32)(hi >> (8 * k)) & 0xff], 1u);
	} else {
		for (u32 k = k0; k < k1; ++k) atomicAdd(&arr[(u32)((k < 8 ? lo : hi) >> (8 * (k & 7))) & 0xff], 1u);
	}
}
__device__ __forceinline__ void lq_ckm_refill(const u8 *dA, u32 pos, uint4 (*win)[LQ_CKM_THREADS], u32 c, CkmWin &W)
{
	W.wsi = pos >> 4;
	const uint4 *g = (const uint4*)dA + W.wsi;
	uint4 t[LQ_CKM_WINW];
#pragma unroll
	for (u32 j = 0; j < LQ_CKM_WINW; ++j) t[j] = g[j];         // (the digit array has 256 bytes of slack behind the last sub-array)
#pragma unroll
	for (u32 j = 0; j < LQ_CKM_WINW; ++j) win[j][c] = t[j];
	W.cwi = 0xffffffffu;
}
// arr[digit] += 1 for the elements at positions [lo, hi) (position = index in the sub-array + its misalignment): the owner of a
// bucket taking in its arrivals
__device__ __forceinline__ void lq_ckm_take(const u8 *dA, u32 lo, u32 hi, u32 *arr, const uint4 (*win)[LQ_CKM_THREADS], u32 c, CkmWin &W)
{
	while (lo < hi) {
		const u32 wi = lo >> 4, k0 = lo & 15;
		if (wi != W.cwi) {
			const u32 off = wi - W.wsi;
			uint4 t;
			if (off < LQ_CKM_WINW) t = win[off][c];
			else if (hi - lo >= 64 + 16) {                         // far beyond the window: four words per trip
				const uint4 *g = (const uint4*)dA + wi;
				const uint4 a0 = g[0], a1 = g[1], a2 = g[2], a3 = g[3];
				lq_ckm_count16(a0.x | (u64)a0.y << 32, a0.z | (u64)a0.w << 32, k0, 16, arr);
				lq_ckm_count16(a1.x | (u64)a1.y << 32, a1.z | (u64)a1.w << 32, 0, 16, arr);
				lq_ckm_count16(a2.x | (u64)a2.y << 32, a2.z | (u64)a2.w << 32, 0, 16, arr);
				lq_ckm_count16(a3.x | (u64)a3.y << 32, a3.z | (u64)a3.w << 32, 0, 16, arr);
				lo += 64 - k0;
				continue;
			}
			else t = ((const uint4*)dA)[wi];
			W.lo = t.x | (u64)t.y << 32; W.hi = t.z | (u64)t.w << 32;
			W.cwi = wi;
		}
		const u32 n = hi - lo < 16 - k0 ? hi - lo : 16 - k0;
		lq_ckm_count16(W.lo, W.hi, k0, k0 + n, arr);
		lo += n;
	}
}
// where sub-chain i of n_sub starts: bucket k, held at slot s (absolute; 0: not held -- the outer loop's first look at k);
// i == n_sub: the end of the pass (k = 256).  k0 / kl: the first / last bucket in use, beg0 / cnt0: k0's region.
__device__ __forceinline__ void lq_ckm_start(u32 i, u32 n_sub, u32 k0, u32 kl, u32 beg0, u32 cnt0, u32 &k, u32 &s)
{
	u32 T = 0;                                                 // starts after k0: at k0 + 1, k0 + 2, k0 + 4, ... <= kl
	while (T < 8 && k0 + (1u << T) <= kl) ++T;
	const u32 cap = n_sub / 4 ? n_sub / 4 : (n_sub > 1 ? 1u : 0u);
	if (T > cap) T = cap;
	const u32 P0 = n_sub - T;
	if (i >= n_sub) { k = 256; s = 0; }
	else if (i < P0) { k = k0; s = i ? beg0 + (u32)((u64)i * cnt0 / P0) : 0; }
	else { k = k0 + (1u << (i - P0)); s = 0; }
}
// (Measured and dropped, round 3: the chain cut into coarse parts that start from states found from scratch BY ONE WAVE -- at
// configs[2] the solver took 818 ms per step against 384 ms for the serial chain: every part read the buckets before its slot
// whole, element by element.  Here a block counts them in one sweep.)
#ifdef LQ_CKM_STATS
__device__ unsigned long long lq_ckm_stats[8];   // (tools/microbench: rounds in all, rounds of the from-scratch states, blocks, cycles, cycles from scratch, most rounds of a block, solves)
#endif
__global__ void __launch_bounds__(LQ_CKM_THREADS)
k_ck_chain256(const CkSeg *cks, const u32 *ckn, const SortSeg *segs, const u8 *D, const u32 *hist, const u32 *begs, u32 *S, u32 *CKS)
{
	__shared__ u32 arr[256];
	__shared__ uint4 win[LQ_CKM_WINW][LQ_CKM_THREADS];
	__shared__ u32 pend[3];
	__shared__ u32 red[LQ_CKM_THREADS / 64];
	__shared__ unsigned long long bal[LQ_CKM_THREADS / 64];
	const u32 c = threadIdx.x, lane = c & 63, wv = c >> 6;
	const u32 n_cks = ckn[1], n_sub_total = ckn[0] / LQ_CKM_Q;
	for (u32 g = blockIdx.x; g < n_sub_total; g += gridDim.x) {
		u32 lo = 0, hi = n_cks;
		while (hi - lo > 1) { const u32 mid = lo + ((hi - lo) >> 1); if (cks[mid].ck0 <= g * LQ_CKM_Q) lo = mid; else hi = mid; }
		const CkSeg ck = cks[lo];
		const u32 n_sub = ck.n_ck / LQ_CKM_Q, sub = g - ck.ck0 / LQ_CKM_Q;
		const u32 ckb = ck.ck0 + sub * LQ_CKM_Q;                  // this sub-chain's first checkpoint
		const SortSeg sg = segs[ck.sgi];
		const u8 *d = D + sg.off;
		const u32 B0 = begs[(u64)ck.sgi * 256 + c], E0 = B0 + hist[(u64)ck.sgi * 256 + c];
		u32 A = B0;
		CkmWin W; W.lo = W.hi = 0; W.cwi = 0xffffffffu; W.wsi = 0;
		const u32 mis = (u32)((size_t)d & 15);                    // positions = indices + mis, counted from dA
		const u8 *dA = d - mis;
		arr[c] = 0;
		if (c < 3) pend[c] = 0;
		u32 it = 0;                                               // rounds so far (which pend[] slot is whose)
#ifdef LQ_CKM_STATS
		const long long t_begin = clock64(); long long t_scratch = t_begin; u32 it_scratch = 0, n_solves = 0;
#endif
		// the first and the last bucket in use
		u32 k0, kl;
		{
			const unsigned long long b = __ballot(E0 > B0);
			if (lane == 0) bal[wv] = b;
			__syncthreads();
			k0 = 256; kl = 0;
			for (u32 w = 0; w < LQ_CKM_THREADS / 64; ++w) if (bal[w]) { if (k0 == 256) k0 = w * 64 + (u32)__builtin_ctzll(bal[w]); kl = w * 64 + 63 - (u32)__builtin_clzll(bal[w]); }
			__syncthreads();
		}
		// block-uniform helpers -------------------------------------------------------------------------------------------
		// the least solution above the state at hand with the buckets before kt full and kt held at st
#define LQ_CKM_SOLVE(kt, st) \
		lq_ckm_refill(dA, mis + A, win, c, W); \
		for (;; ++it) { \
			if (c == 0) pend[(it + 1) % 3] = 0; \
			bool did_ = false; \
			for (int sp_ = 0; sp_ < LQ_CKM_SPINS; ++sp_) { \
				u32 need = c < (kt) ? E0 : B0 + LQ_LDS_LOOK(&arr[c]); \
				if (c == (kt) && need < (st)) need = (st);         /* held at st -- or further, where arrivals filled the bucket beyond st before its phase began */ \
				if (need > E0) need = E0; \
				if (A < need) { lq_ckm_take(dA, mis + A, mis + need, arr, win, c, W); A = need; did_ = true; } \
			} \
			if (did_) pend[it % 3] = 1; \
			LQ_CKM_BARRIER(); \
			if (!pend[it % 3]) { ++it; break; } \
		}
		// where the outer loop is in the state at hand: the first bucket from kt on with unread slots and its cursor (256, len: none)
#define LQ_CKM_WHERE(kt, kc, ac) do { \
			const unsigned long long b_ = __ballot(c >= (kt) && A < E0); \
			if (lane == 0) bal[wv] = b_; \
			__syncthreads(); \
			(kc) = 256; \
			for (u32 w_ = 0; w_ < LQ_CKM_THREADS / 64; ++w_) if (bal[w_]) { (kc) = w_ * 64 + (u32)__builtin_ctzll(bal[w_]); break; } \
			if (c == (kc)) red[0] = A; \
			__syncthreads(); \
			(ac) = (kc) < 256 ? red[0] : sg.len; \
			__syncthreads(); \
		} while (0)
		// elements picked up so far
#define LQ_CKM_PICKED(out) do { \
			u32 m_ = A - B0; \
			for (int o_ = 32; o_ > 0; o_ >>= 1) m_ += __shfl_xor(m_, o_); \
			if (lane == 0) red[wv] = m_; \
			__syncthreads(); \
			u64 p_ = 0; \
			for (u32 w_ = 0; w_ < LQ_CKM_THREADS / 64; ++w_) p_ += red[w_]; \
			__syncthreads(); \
			(out) = p_; \
		} while (0)
#define LQ_CKM_WRITE(x, slot) do { S[((u64)ckb + (x)) * 256 + c] = A; if (c == 0) CKS[ckb + (x)] = (slot); } while (0)
		u32 ks, ss, ke, se;
		lq_ckm_start(sub, n_sub, k0, kl, k0 < 256 ? begs[(u64)ck.sgi * 256 + k0] : 0, k0 < 256 ? hist[(u64)ck.sgi * 256 + k0] : 0, ks, ss);
		lq_ckm_start(sub + 1, n_sub, k0, kl, k0 < 256 ? begs[(u64)ck.sgi * 256 + k0] : 0, k0 < 256 ? hist[(u64)ck.sgi * 256 + k0] : 0, ke, se);
		if (sub) {
			// this sub-chain's first state from scratch: the regions of the buckets before ks, and the slots of ks up to ss, are read
			// whole -- counted by all threads, 16 bytes at a time --, then the cascade
			const u32 full_hi = begs[(u64)ck.sgi * 256 + ks];     // (the regions lie one after the other: [0, beg of ks) is what the buckets before ks own)
			const u32 hold_hi = ss > full_hi ? ss : full_hi;
			{
				const u8 *p_lo = d, *p_hi = d + hold_hi;
				for (const u8 *wa = (const u8*)((size_t)p_lo & ~(size_t)15) + (size_t)c * 16; wa < p_hi; wa += (size_t)LQ_CKM_THREADS * 16) {
					const uint4 w4 = *(const uint4*)wa;
					const u32 ww[4] = { w4.x, w4.y, w4.z, w4.w };
#pragma unroll
					for (u32 k = 0; k < 16; ++k) {
						const u8 *gq = wa + k;
						if (gq >= p_lo && gq < p_hi) atomicAdd(&arr[(ww[k >> 2] >> ((k & 3) * 8)) & 0xff], 1u);
					}
				}
			}
			if (c < ks) A = E0; else if (c == ks && A < hold_hi) A = hold_hi < E0 ? hold_hi : E0;
			__syncthreads();
			LQ_CKM_SOLVE(ks, ss)
#ifdef LQ_CKM_STATS
			t_scratch = clock64(); it_scratch = it;
#endif
		}
		u32 kc, ac;
		LQ_CKM_WHERE(ks, kc, ac);
		const u64 target = 2 * ((u64)sg.len / ck.n_ck) + 1;       // elements a walker's piece should hold (half the quota is slack, LQ_CKM_Q)
		u64 picked_before = 0, picked_at_last = 0;
		LQ_CKM_PICKED(picked_before);
		picked_at_last = picked_before;
		LQ_CKM_WRITE(0, ac);
		u32 n_out = 1, est = LQ_CKW_STEP, jump = 1;
		// follow the walk to the first state of the next sub-chain
		while (kc < ke || (kc == ke && ac < se)) {
			const u32 ek = begs[(u64)ck.sgi * 256 + kc] + hist[(u64)ck.sgi * 256 + kc];
			u32 kt, st; bool by_slots;
			if (ek - ac > est) { kt = kc; st = ac + est; by_slots = true; }
			else { kt = kc + jump < 256 ? kc + jump : 256; st = 0; by_slots = false; }
			// never beyond the next sub-chain's start: (ke, se) -- a first look (st == 0) at ke comes before any of its slots
			if (kt > ke || (kt == ke && st != 0 && st >= se)) { kt = ke; st = se; }
			const u32 step = by_slots ? st - ac : 0;
			LQ_CKM_SOLVE(kt, st)
#ifdef LQ_CKM_STATS
			++n_solves;
#endif
			LQ_CKM_WHERE(kt, kc, ac);
			u64 picked;
			LQ_CKM_PICKED(picked);
			{	// the next step covers what the last one's yield says is worth `target`
				const u64 got = picked - picked_before;
				picked_before = picked;
				if (by_slots && step) {
					u64 e2 = got ? (u64)step * target / got : (u64)step * 4;
					if (e2 > (u64)step * 4) e2 = (u64)step * 4;
					est = (u32)(e2 < LQ_CKW_STEP ? LQ_CKW_STEP : e2 > (1u << 24) ? (1u << 24) : e2);
				} else if (!by_slots) {
					if (got < target / 2) jump = jump < 128 ? jump * 2 : 256;
					else if (got > target && jump > 1) jump /= 2;
				}
			}
			if (picked - picked_at_last >= target / 2 && n_out < LQ_CKM_Q && (kc < ke || (kc == ke && ac < se))) {
				LQ_CKM_WRITE(n_out, ac);
				++n_out; picked_at_last = picked;
			}
		}
		// quota not used: the state at hand -- the next sub-chain's first, or the pass's last (a walker finds nothing to do)
		for (; n_out < LQ_CKM_Q; ++n_out) LQ_CKM_WRITE(n_out, ac);
#ifdef LQ_CKM_STATS
		if (c == 0) {
			atomicAdd(&lq_ckm_stats[0], (unsigned long long)it); atomicAdd(&lq_ckm_stats[1], (unsigned long long)it_scratch); atomicAdd(&lq_ckm_stats[2], 1ULL);
			atomicAdd(&lq_ckm_stats[3], (unsigned long long)(clock64() - t_begin)); atomicAdd(&lq_ckm_stats[4], (unsigned long long)(t_scratch - t_begin));
			atomicMax(&lq_ckm_stats[5], (unsigned long long)it); atomicAdd(&lq_ckm_stats[6], (unsigned long long)n_solves);
		}
#endif
#undef LQ_CKM_SOLVE
#undef LQ_CKM_WHERE
#undef LQ_CKM_PICKED
#undef LQ_CKM_WRITE
		__syncthreads();
	}
}
