// longqc_amd/csrc/kernels_rsort.hpp -- the streaming side of klib's passes on 8-byte records.
//
// klib's radix sort (ksort.h:99-129) moves 16-byte anchors at every level; a level's token walk only reads the digit
// bytes, and the next level only needs the next digits in the new arrangement.  The anchors of a query that goes through
// klib's passes therefore stay in the buffer the seed stage wrote them to (the *originals*, never modified), and the
// levels permute one record per anchor:
//     RRec { key = x >> 32 = strand:1 | rid:31,   im = tie mark:1 | index of the original:31 }
// key holds every digit of the levels at shift >= 32 (strand, the three bytes of rid); the levels below (bytes of the
// target position, reached only by the few sub-arrays that still hold two marked anchors after the rid levels) read
// their digit from the original through the index.  Per level and anchor the streaming kernels move 9 B (records in,
// digit out) + 20 B (destination + record in, record out) instead of 33 + 36 B of anchors, and the scattered side of the
// move is an 8-byte store: sub-arrays of up to half a million anchors keep their destination lines in one XCD's L2.
// An anchor itself is read and written exactly once more: gathered through the index when its bucket is finished
// (k_rs_children: buckets of <= 64; the parallel sort's kernels for the buckets that leave klib's passes, kernels_psort.hpp).
#pragma once
#include "lq_common.hpp"
#include "kernels_sort.hpp"

struct alignas(8) RRec { u32 key, im; };
#define LQ_R_MARK 0x80000000u
#define LQ_R_IDX(im) ((im) & 0x7fffffffu)

// digit of a record at `shift` (the byte of x at that bit position); O = the originals
__device__ __forceinline__ u32 lq_r_digit(const RRec e, u32 shift, const mm128 *O)
{
	return shift >= 32 ? (e.key >> (shift - 32)) & 0xffu : (u32)(O[LQ_R_IDX(e.im)].x >> shift) & 0xffu;
}

// One block per tile of a sub-array: D <- digit, hist[seg][*] += digit histogram, mhist[seg][*] += marked anchors per bucket.
// FIRST (the level that k_sort_init opens): the records are built from the originals on the way.
// The top levels have two to a handful of buckets (strand; the byte of rid above 65536 targets): a wave's 64 digits are
// then a few distinct values and 64 LDS atomics on one address would serialise.  Up to LQ_RS_PEEL distinct values of a wave
// are counted by ballots (one atomic per value and wave); what is left after that goes the ordinary way.
#define LQ_RS_PEEL 4
template <bool FIRST>
__global__ void __launch_bounds__(256)
k_rs_hist(const SortSeg *segs, const SortTile *tiles, const u32 *n_tiles_p, u32 tile, int xcd, const mm128 *O, RRec *R, u8 *D, u32 *hist, u32 *mhist,
          unsigned long long *tally)
{
	__shared__ u32 lh[256], lm[256];
	const u32 n_tiles = *n_tiles_p;
	const u32 lane = threadIdx.x & 63;
	LQ_TILE_LOOP(ti, n_tiles, xcd) {
		const SortTile tl = tiles[ti];
		const SortSeg sg = segs[tl.sgi];
		const u32 i0 = tl.tile * tile, i1 = sg.len - i0 < tile ? sg.len : i0 + tile;
		RRec *r = R + sg.off;
		u8 *d = D + sg.off;
		if (tl.tile == 0 && threadIdx.x == 0 && tally) atomicAdd(tally, (unsigned long long)sg.len);
		for (u32 c = threadIdx.x; c < 256; c += blockDim.x) { lh[c] = 0; lm[c] = 0; }
		__syncthreads();
		for (u32 ib = i0; ib < i1; ib += 256) {                 // (uniform trip count: the ballots below need whole waves)
			const u32 i = ib + threadIdx.x;
			const bool act = i < i1;
			u32 dg = 0, mk = 0;
			if (act) {
				RRec e;
				if (FIRST) {
					const mm128 a = O[sg.off + i];
					e.key = (u32)(a.x >> 32); e.im = (u32)(sg.off + i) | ((a.y & LQ_TIE_MARK) ? LQ_R_MARK : 0u);
					r[i] = e;
					dg = (u32)(a.x >> sg.shift) & 0xffu;
				} else {
					e = r[i];
					dg = lq_r_digit(e, sg.shift, O);
				}
				d[i] = (u8)dg;
				mk = e.im >> 31;
			}
			u64 todo = __ballot(act);
			const u64 marked = __ballot(act && mk);
			for (int p = 0; p < LQ_RS_PEEL && todo; ++p) {
				const u32 f = (u32)__builtin_ctzll(todo);
				const u32 v = (u32)__builtin_amdgcn_readlane((int)dg, (int)f);
				const u64 same = __ballot(act && dg == v) & todo;
				if (lane == f) { atomicAdd(&lh[v], (u32)__popcll(same)); const u32 nm = (u32)__popcll(same & marked); if (nm) atomicAdd(&lm[v], nm); }
				todo &= ~same;
			}
			if ((todo >> lane) & 1) { atomicAdd(&lh[dg], 1u); if (mk) atomicAdd(&lm[dg], 1u); }
		}
		__syncthreads();
		u32 *hrow = hist + (u64)tl.sgi * 256, *mrow = mhist + (u64)tl.sgi * 256;
		if (sg.len <= tile) {
			for (u32 c = threadIdx.x; c < 256; c += blockDim.x) { hrow[c] = lh[c]; mrow[c] = lm[c]; }
		} else {
			for (u32 c = threadIdx.x; c < 256; c += blockDim.x) { if (lh[c]) atomicAdd(&hrow[c], lh[c]); if (lm[c]) atomicAdd(&mrow[c], lm[c]); }
		}
		__syncthreads();
	}
}

// One block per tile: Rn[dst[i]] = Rc[i] (an identity pass -- one bucket holds the whole sub-array -- is a plain copy: the
// next level reads the other array; a two-bucket pass has moved its records itself).  Four independent loads in flight per thread.
__global__ void __launch_bounds__(256)
k_rs_scatter(const SortSeg *segs, const SegInfo *info, const SortTile *tiles, const u32 *n_tiles_p, u32 tile, int xcd, const RRec *Rc, RRec *Rn, const u32 *dst,
             unsigned long long *tally)
{
	const u32 n_tiles = *n_tiles_p;
	LQ_TILE_LOOP(ti, n_tiles, xcd) {
		const SortTile tl = tiles[ti];
		const SortSeg sg = segs[tl.sgi];
		const u32 i0 = tl.tile * tile, i1 = sg.len - i0 < tile ? sg.len : i0 + tile;
		const u64 *rc = (const u64*)(Rc + sg.off);              // (a record as one 8-byte word)
		u64 *rn = (u64*)(Rn + sg.off);
		const u32 *ds = dst + sg.off;
		u32 i = i0 + threadIdx.x;
		const u32 kind = info[tl.sgi].kind;
		if (kind == LQ_SEG_TWO) continue;                           // (k_sort_two_tiled<2> has moved these records)
		if (tl.tile == 0 && threadIdx.x == 0 && tally) atomicAdd(tally, (unsigned long long)sg.len);
		if (kind == LQ_SEG_IDENTITY) { for (; i < i1; i += 256) rn[i] = rc[i]; continue; }
		for (; i + 3 * 256 < i1; i += 4 * 256) {
			const u32 d0 = ds[i], d1 = ds[i + 256], d2 = ds[i + 512], d3 = ds[i + 768];
			const u64 e0 = rc[i], e1 = rc[i + 256], e2 = rc[i + 512], e3 = rc[i + 768];
			rn[d0] = e0; rn[d1] = e1; rn[d2] = e2; rn[d3] = e3;
		}
		for (; i < i1; i += 256) rn[ds[i]] = rc[i];
	}
}
