// longqc_amd/csrc/kernels_sort.hpp -- klib-order sort of every query's anchors by x.
//
// Why not a stock stable radix sort: the reference sorts anchors with klib's *in-place, unstable*
// MSD radix sort (lqmap.c:238 -> ksort.h:99-134) and mm_chain_dp then breaks score ties by array
// order (chain.c:69-76).  Anchors with equal x (one target minimizer hit by a repeated query
// minimizer) therefore reach the DP in klib's order, and a stable order changes real rows (1 of
// 10,000 (query,part) pairs in the cfg1 fixture).  So this sort reproduces klib's permutation
// exactly.
//
// klib's pass over a sub-array is a token walk: the token sits on a bucket, takes that bucket's
// next unread element (original slot order) and jumps to the bucket the element belongs to; an
// element is written to the next free slot of its own bucket when it is taken; the outer bucket
// only advances when full.  The walk is inherently sequential, but sub-arrays are independent,
// so the GPU runs it level-synchronously: per level (byte 7 .. byte 0) and per live sub-array,
//   k_sort_copy_hist : cooperative copy A->B (the pristine source) + 256-bin histogram,
//   k_sort_walk      : one lane per sub-array walks B and scatters into A; the 256 bucket
//                      cursors of each lane live in LDS ([256][64] u32 = 64 KiB per wave,
//                      lane-minor so that the 32-lane halves never bank-conflict),
//   k_sort_children  : buckets > 64 elements become next-level sub-arrays, smaller ones are
//                      finished with the (stable) insertion sort klib uses (ksort.h:87-97).
#pragma once
#include "lq_common.hpp"

#ifdef LQ_EMU
#define LQ_SHARED static
#else
#define LQ_SHARED __shared__
#endif

__device__ __forceinline__ void lq_insertion_sort_x(mm128 *a, u32 n)
{
	for (u32 i = 1; i < n; ++i) {
		if (a[i].x < a[i - 1].x) {
			mm128 t = a[i];
			u32 j = i;
			for (; j > 0 && t.x < a[j - 1].x; --j) a[j] = a[j - 1];
			a[j] = t;
		}
	}
}

// radix_sort_128x entry (ksort.h:130-134): arrays of <= 64 elements are insertion sorted
__global__ void k_sort_init(const u64 *aq_off, u64 a_base, u32 n_q, mm128 *A, SortSeg *segs, u32 *n_segs)
{
	u32 q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_q) return;
	u64 off = aq_off[q] - a_base, len = aq_off[q + 1] - aq_off[q];
	if (len > LQ_RS_MIN) {
		u32 s = atomicAdd(n_segs, 1u);
		SortSeg sg; sg.off = off; sg.len = (u32)len; sg.shift = 56;
		segs[s] = sg;
	} else if (len > 1) lq_insertion_sort_x(A + off, (u32)len);
}

// one block per sub-array: B <- A, hist[seg][digit]++
__global__ void k_sort_copy_hist(const SortSeg *segs, u32 n_segs, const mm128 *A, mm128 *B, u32 *hist)
{
	u32 sgi = blockIdx.x;
	if (sgi >= n_segs) return;
	SortSeg sg = segs[sgi];
	const mm128 *a = A + sg.off;
	mm128 *b = B + sg.off;
	u32 *h = hist + (u64)sgi * 256;
	for (u32 i = threadIdx.x; i < sg.len; i += blockDim.x) {
		mm128 e = a[i];
		b[i] = e;
		atomicAdd(&h[(e.x >> sg.shift) & 0xff], 1u);
	}
}

#define LQ_WALK_LANES 64
// one lane per sub-array (grid-stride over the level's work list)
__global__ void __launch_bounds__(LQ_WALK_LANES)
k_sort_walk(const SortSeg *segs, u32 n_segs, mm128 *A, const mm128 *B, const u32 *hist, u32 *begs)
{
	LQ_SHARED u32 nxt[256][LQ_WALK_LANES];
	const u32 lane = threadIdx.x;
	for (u64 sgi = (u64)blockIdx.x * LQ_WALK_LANES + lane; sgi < n_segs; sgi += (u64)gridDim.x * LQ_WALK_LANES) {
		SortSeg sg = segs[sgi];
		const u32 *cnt = hist + sgi * 256;
		u32 *bg = begs + sgi * 256;
		u32 acc = 0;
		bool single = false;
		for (int c = 0; c < 256; ++c) {
			u32 n = cnt[c];
			nxt[c][lane] = acc; bg[c] = acc;
			if (n == sg.len) single = true;
			acc += n;
		}
		if (single) continue;                               // one bucket holds everything: the pass is the identity
		mm128 *a = A + sg.off;
		const mm128 *b = B + sg.off;
		const u32 sh = sg.shift;
		u32 endk = 0;
		for (u32 k = 0; k < 256; ++k) {
			endk += cnt[k];                                 // = beg[k] + cnt[k]
			while (nxt[k][lane] < endk) {
				mm128 e = b[nxt[k][lane]];
				u32 l = (u32)(e.x >> sh) & 0xff;
				while (l != k) {
					u32 s = nxt[l][lane]++;
					mm128 t = b[s];
					a[s] = e;
					e = t;
					l = (u32)(e.x >> sh) & 0xff;
				}
				a[nxt[k][lane]++] = e;
			}
		}
	}
}

// one thread per (sub-array, bucket): recurse or finish (ksort.h:121-128)
__global__ void k_sort_children(const SortSeg *segs, u32 n_segs, mm128 *A, const u32 *hist, const u32 *begs,
                                SortSeg *next, u32 *n_next)
{
	u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (u64)n_segs * 256) return;
	u32 sgi = (u32)(t >> 8);
	SortSeg sg = segs[sgi];
	if (sg.shift == 0) return;
	u32 n = hist[t], bg = begs[t];
	if (n > LQ_RS_MIN) {
		u32 s = atomicAdd(n_next, 1u);
		SortSeg c; c.off = sg.off + bg; c.len = n; c.shift = sg.shift > 8 ? sg.shift - 8 : 0;
		next[s] = c;
	} else if (n > 1) lq_insertion_sort_x(A + sg.off + bg, n);
}
