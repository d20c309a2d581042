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
// only advances when full.  Only the *digits* (one byte per element) drive the walk, and the walk of
// the next level only needs the digits in the arrangement this level leaves behind.  So the 16-byte
// anchors do not travel through the levels at all: the anchors of a query that needs klib's order
// stay where the seed stage wrote them (the lane's second buffer, "the originals"), and what the
// levels permute is an 8-byte record per anchor -- RRec {key = x >> 32 (strand | rid), index of the
// original | tie mark} -- ping-ponged between two record arrays.  An anchor is written once, to
// its final place in A, when its bucket is done with klib (kernels_rsort.hpp).  Each level (byte 7
// .. byte 0) runs, for all live sub-arrays at once:
//   k_rs_hist        : digit byte array D + histogram + count of marked anchors per bucket, over
//                      tiles of the sub-arrays (the first level builds the records on the way);
//   k_sort_classify  : per sub-array: bucket offsets; identity (one bucket), two-bucket, or general;
//   two-bucket passes (the top level: strand bit) have a closed form -- every element's destination
//                      follows from two prefix counts -- and run fully parallel, over tiles (k_sort_two_tiled), moving their records themselves;
//   k_sort_walk_*    : general passes: a token walk over the digit bytes that records dst[src]
//                      (here, kernels_walk.hpp; long walks cut at computed states: kernels_ckpt.hpp);
//   k_rs_scatter     : R'[dst[i]] = R[i], over tiles;
//   k_rs_children    : buckets > 64 elements become next-level sub-arrays or leave for the parallel
//                      sort (fewer than two marked anchors: no equal x inside), smaller ones are
//                      finished with the (stable, hence unique) order of the insertion sort klib uses
//                      (ksort.h:87-97) and written to A.
#pragma once
#include "lq_common.hpp"
struct CkSeg { u32 sgi, tile0, ck0, n_ck; };      // one long sub-array of a checkpointed pass (kernels_ckpt.hpp): its segment, first prefix tile, first checkpoint, checkpoints

// Block-cooperative kernels are written in phases: LQ_BLOCK_LOOP(t) { ... } runs its body once per thread of
// the block (t = thread index), LQ_BLOCK_SYNC() separates phases.
#define LQ_SHARED __shared__
#define LQ_BLOCK_LOOP(t) for (u32 t = threadIdx.x, lq_once_ = 1; lq_once_; lq_once_ = 0)
#define LQ_BLOCK_SYNC() __syncthreads()

#define LQ_SEG_GENERAL  0
#define LQ_SEG_IDENTITY 1
#define LQ_SEG_TWO      2

struct SegInfo { u32 kind, c0, c1, cnt0; };

// block-wide exclusive scan of v[0..256) in LDS (256 or more threads; returns with the result in v, total in *tot)
__device__ __forceinline__ void lq_scan256(u32 *v, u32 *tmp, u32 *tot)
{
	const u32 t = threadIdx.x;
	for (u32 d = 1; d < 256; d <<= 1) {
		u32 a = 0;
		if (t < 256) a = v[t] + (t >= d ? v[t - d] : 0);
		__syncthreads();
		if (t < 256) v[t] = a;
		__syncthreads();
	}
	if (t < 256) tmp[t] = t ? v[t - 1] : 0;
	if (t == 255 && tot) *tot = v[255];
	__syncthreads();
	if (t < 256) v[t] = tmp[t];
	__syncthreads();
}


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

// ---- tiles of the level's sub-arrays for the streaming kernels -------------------------------------------------------
// The top passes of a batch have few, long sub-arrays (one per (query, strand) at first: a few hundred of ~10^5..10^6 anchors);
// one block per sub-array leaves most of the chip idle.  The level's sub-arrays are cut into tiles of LQ_SORT_TILE anchors
// (a launch parameter: tests shrink it) (k_sort_tiles: a device-side list, one atomic per wave) and the histogram and the
// scatter run one block per tile; a sub-array of several tiles adds its tile histograms with atomics into rows zeroed by
// k_sort_tiles, a sub-array of one tile stores its row.
#define LQ_SORT_TILE 8192
struct SortTile { u32 sgi, tile; };
// The tile list in XCD-major order.  Blocks are dealt to the 8 XCDs round-robin (observed, not promised: block b runs on XCD
// b % 8 -- a pure speed assumption), and every XCD has its own L2: XCD x takes the x-th eighth of the list, so the tiles of one
// sub-array (adjacent in the list) write their destination lines through the same L2.  Needs gridDim.x % 8 == 0; every tile is
// visited exactly once whatever the placement really is.  xcd == 0: plain block-strided order.
#define LQ_XCDS 8
#define LQ_TILE_LOOP(ti, n_tiles, xcd) \
	for (u32 per_ = (xcd) ? ((n_tiles) + LQ_XCDS - 1) / LQ_XCDS : (n_tiles), x_ = (xcd) ? blockIdx.x % LQ_XCDS : 0, st_ = (xcd) ? gridDim.x / LQ_XCDS : gridDim.x, \
	         j_ = (xcd) ? blockIdx.x / LQ_XCDS : blockIdx.x, ti = x_ * per_ + j_; j_ < per_; j_ += st_, ti = x_ * per_ + j_) if (ti < (n_tiles))

__global__ void __launch_bounds__(256)
k_sort_tiles(const SortSeg *segs, const u32 *n_segs_p, u32 tile, SortTile *tiles, u32 *n_tiles, u32 *hist, u32 *mhist, u32 *zero0, u32 *zero1, u32 n_zero1, u32 *zero2, u32 *zero3, u32 n_zero3)
{
	const u32 n_segs = *n_segs_p;
	const u32 lane = threadIdx.x & 63;
	// the level's other counters start from zero (kernels later in the stream count into them; none of this kernel's blocks reads them)
	if (blockIdx.x == 0 && threadIdx.x == 0) { *zero0 = 0; for (u32 i = 0; i < n_zero1; ++i) zero1[i] = 0; *zero2 = 0; for (u32 i = 0; i < n_zero3; ++i) zero3[i] = 0; }
	for (u32 base = blockIdx.x * blockDim.x; base < n_segs; base += gridDim.x * blockDim.x) {
		const u32 sgi = base + threadIdx.x;
		u32 nt = 0;
		if (sgi < n_segs) { nt = (segs[sgi].len + tile - 1) / tile; if (nt == 0) nt = 1; }
		u32 inc = nt;
		for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
		u32 wbase = 0;
		if (lane == 63 && inc) wbase = atomicAdd(n_tiles, inc);
		wbase = __shfl(wbase, 63);
		const u32 at = wbase + inc - nt;
		for (u32 t = 0; t < nt; ++t) { SortTile e; e.sgi = sgi; e.tile = t; tiles[at + t] = e; }
		if (nt > 1) for (u32 c = 0; c < 256; ++c) { hist[(u64)sgi * 256 + c] = 0; mhist[(u64)sgi * 256 + c] = 0; }
	}
}

// one thread per sub-array: bucket offsets and the kind of pass; lists of general / two-bucket sub-arrays
// size classes of general passes: digits of the sub-array fit a 4 / 16 / 64 / 156 KiB LDS window, or not at all
#define LQ_WALK_CLASSES 5
struct WalkCaps { u32 c[4]; };        // default {4096, 16384, 65536, 159744}; tests shrink them to reach every class
__device__ __forceinline__ u32 lq_walk_class(u32 len, const WalkCaps &w)
{
	return len <= w.c[0] ? 0u : len <= w.c[1] ? 1u : len <= w.c[2] ? 2u : len <= w.c[3] ? 3u : 4u;
}

// walk_list holds LQ_WALK_CLASSES lists of n_segs entries each; counters = [n_two, n_walk[0..4]]
// One wave per sub-array: bucket offsets (exclusive scan of the 256 counts, four per lane) and the kind of pass.
#define LQ_CLASSIFY_THREADS 64
__global__ void __launch_bounds__(LQ_CLASSIFY_THREADS)
k_sort_classify(const SortSeg *segs, const u32 *n_segs_p, u32 list_cap, const u32 *hist, u32 *begs, SegInfo *info,
                u32 *walk_list, u32 *two_list, u32 *n_two, u32 *n_walk, WalkCaps caps)
{
	const u32 n_segs = *n_segs_p;
	const u32 lane = threadIdx.x;
	for (u32 sgi = blockIdx.x; sgi < n_segs; sgi += gridDim.x) {
		const u32 *cnt = hist + (u64)sgi * 256;
		u32 *bg = begs + (u64)sgi * 256;
		u32 nz = 0, c0 = 0, c1 = 0;
		const uint4 v = *(const uint4*)(cnt + 4 * lane);
		const u32 s1 = v.x, s2 = s1 + v.y, s3 = s2 + v.z, s4 = s3 + v.w;
		u32 inc = s4;                                             // inclusive scan of the lane sums
		for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
		const u32 ex = inc - s4;
		uint4 o4; o4.x = ex; o4.y = ex + s1; o4.z = ex + s2; o4.w = ex + s3;
		*(uint4*)(bg + 4 * lane) = o4;
		const u32 m4 = (v.x != 0) | (v.y != 0) << 1 | (v.z != 0) << 2 | (v.w != 0) << 3;
		const u32 nzl = __popc(m4);
		nz = __popcll(__ballot(nzl & 1)) + 2 * __popcll(__ballot(nzl & 2)) + 4 * __popcll(__ballot(nzl & 4));
		const u64 any = __ballot(m4 != 0);
		if (any) {
			const u32 f = (u32)__builtin_ctzll(any);
			const u32 mf = (u32)__builtin_amdgcn_readlane((int)m4, (int)f);
			c0 = 4 * f + (u32)__builtin_ctz(mf);
			const u32 rest = mf & (mf - 1);
			const u64 any2 = any & (any - 1);
			const u32 g = any2 ? (u32)__builtin_ctzll(any2) : 0;
			const u32 mg = (u32)__builtin_amdgcn_readlane((int)m4, (int)g);
			if (rest) c1 = 4 * f + (u32)__builtin_ctz(rest);
			else if (any2) c1 = 4 * g + (u32)__builtin_ctz(mg);
		}
		if (lane == 0) {
			SegInfo si; si.c0 = c0; si.c1 = c1; si.cnt0 = cnt[c0];
			if (nz <= 1) si.kind = LQ_SEG_IDENTITY;                  // one bucket holds everything: the pass is the identity
			else if (nz == 2) { si.kind = LQ_SEG_TWO; two_list[atomicAdd(n_two, 1u)] = sgi; }
			else {
				si.kind = LQ_SEG_GENERAL;
				const u32 wc = lq_walk_class(segs[sgi].len, caps);
				walk_list[(u64)wc * list_cap + atomicAdd(&n_walk[wc], 1u)] = sgi;
			}
			info[sgi] = si;
		}
	}
}

// ---- two-bucket pass, closed form ------------------------------------------------------------
// Regions R0 = [0,cnt0) and R1 = [cnt0,len).  X_t = t-th element of R0 that belongs to bucket c1,
// Y_t = t-th element of R1 that belongs to c0 (same count m).  klib's walk leaves bucket-c0 elements
// of R0 in place and drops Y_t into the hole of X_t; in R1 it cuts the slots into runs that end at
// Y_0, Y_1, ...: X_t lands on the first slot of run t and the run's own elements shift right by one;
// everything after Y_{m-1} stays.
struct alignas(16) TwoW16 { u32 w[4]; };
// digit k of the 16 loaded at a 16-byte aligned position
#define LQ_TWO_DIGIT(W, k) (((W).w[(k) >> 2] >> (((k) & 3) * 8)) & 0xffu)
// The strand pass at the top of every (query) array is a two-bucket pass over the whole array (a few hundred sub-arrays of
// ~10^5..10^6 anchors per batch), so it runs over tiles: k_two_tiles lists the tiles of the two-bucket sub-arrays (contiguous
// per sub-array, first index in tile0[]), <0> counts the X / Y elements of every tile, one wave per sub-array scans its tile
// counts (k_sort_two_scan; the total is m), <1> writes the position lists HX / PY at tile base + rank, <2> computes every
// element's destination by the formulas above (ranks from the scanned tile counts) and moves its record there -- no destination
// array for these passes (rocprofv3 counted 15 B of HBM writes per 4-byte destination stored 64 bytes apart per lane).
__global__ void __launch_bounds__(256)
k_two_tiles(const SortSeg *segs, const u32 *two_list, const u32 *n_two_p, u32 tile, SortTile *tiles, u32 *n_tiles, u32 *tile0)
{
	const u32 n_two = *n_two_p;
	const u32 lane = threadIdx.x & 63;
	for (u32 base = blockIdx.x * blockDim.x; base < n_two; base += gridDim.x * blockDim.x) {
		const u32 li = base + threadIdx.x;
		u32 nt = 0, sgi = 0;
		if (li < n_two) { sgi = two_list[li]; nt = (segs[sgi].len + tile - 1) / tile; }
		u32 inc = nt;
		for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
		u32 wbase = 0;
		if (lane == 63 && inc) wbase = atomicAdd(n_tiles, inc);
		wbase = __shfl(wbase, 63);
		const u32 at = wbase + inc - nt;
		if (li < n_two) tile0[sgi] = at;
		for (u32 t = 0; t < nt; ++t) { SortTile e; e.sgi = sgi; e.tile = t; tiles[at + t] = e; }
	}
}

// exclusive scan of one sub-array's tile counts (X, Y interleaved), one wave per sub-array; two_m[sgi] = number of X (= of Y)
__global__ void __launch_bounds__(64)
k_sort_two_scan(const SortSeg *segs, const u32 *two_list, const u32 *n_two_p, u32 tile, const u32 *tile0, u32 *tcnt, u32 *two_m)
{
	const u32 n_two = *n_two_p, lane = threadIdx.x;
	for (u32 li = blockIdx.x; li < n_two; li += gridDim.x) {
		const u32 sgi = two_list[li];
		const u32 nt = (segs[sgi].len + tile - 1) / tile;
		u32 *c = tcnt + 2 * (u64)tile0[sgi];
		const u32 per = (nt + 63) / 64, a = lane * per < nt ? lane * per : nt, b = a + per < nt ? a + per : nt;
		u32 sx = 0, sy = 0;
		for (u32 x = a; x < b; ++x) { sx += c[2 * x]; sy += c[2 * x + 1]; }
		u32 ix = sx, iy = sy;
		for (int d = 1; d < 64; d <<= 1) { const u32 ox = __shfl_up(ix, d), oy = __shfl_up(iy, d); if ((int)lane >= d) { ix += ox; iy += oy; } }
		u32 rx = ix - sx, ry = iy - sy;
		for (u32 x = a; x < b; ++x) { const u32 vx = c[2 * x], vy = c[2 * x + 1]; c[2 * x] = rx; c[2 * x + 1] = ry; rx += vx; ry += vy; }
		if (lane == 63) two_m[sgi] = ix;
	}
}

template <int MODE>   // 0: count, 1: position lists, 2: destinations
__global__ void __launch_bounds__(256)
k_sort_two_tiled(const SortSeg *segs, const SegInfo *info, const SortTile *tiles, const u32 *n_tiles_p, u32 tile, const u8 *D,
                 u32 *tcnt, const u32 *two_m, u32 *HX, u32 *PY, const u64 *Rc, u64 *Rn)
{
	__shared__ u32 wx[4], wy[4];
	__shared__ u32 sd[MODE == 2 ? 256 * 16 : 1];                // <2>: the destinations of the 4096 elements at hand
	const u32 t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const u32 n_tiles = *n_tiles_p;
	for (u32 ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
		const SortTile tl = tiles[ti];
		const SortSeg sg = segs[tl.sgi];
		const SegInfo si = info[tl.sgi];
		const u64 off = sg.off;
		const u32 i0 = tl.tile * tile, i1 = sg.len - i0 < tile ? sg.len : i0 + tile;
		const u64 lo = off + i0, hi = off + i1, a0 = lo & ~(u64)15;
		const u32 n_words = (u32)((hi - a0 + 15) >> 4);
		const u32 cnt0 = si.cnt0, c1 = si.c1;
		u32 bx = 0, by = 0, m = 0;                              // X / Y elements before the words at hand
		if (MODE) { bx = tcnt[2 * (u64)ti]; by = tcnt[2 * (u64)ti + 1]; }
		if (MODE == 2) m = two_m[tl.sgi];
		u32 *hx = HX + off, *py = PY + off;
		for (u32 w0 = 0; w0 < n_words; w0 += 256) {
			const u32 wi = w0 + t;
			const u64 p = a0 + (u64)wi * 16;
			TwoW16 W; W.w[0] = W.w[1] = W.w[2] = W.w[3] = 0;
			u32 cx = 0, cy = 0;
			if (wi < n_words) {
				W = *(const TwoW16*)(D + p);
				for (u32 k = 0; k < 16; ++k) {
					const u64 g = p + k;
					if (g >= lo && g < hi) { const u32 i = (u32)(g - off); const bool in0 = i < cnt0, is1 = LQ_TWO_DIGIT(W, k) == c1; cx += in0 && is1; cy += !in0 && !is1; }
				}
			}
			// exclusive ranks of this thread's word among the 256 words at hand, and their totals
			u32 ix = cx, iy = cy;
			for (int d = 1; d < 64; d <<= 1) { const u32 ox = __shfl_up(ix, d), oy = __shfl_up(iy, d); if ((int)lane >= d) { ix += ox; iy += oy; } }
			if (lane == 63) { wx[wv] = ix; wy[wv] = iy; }
			__syncthreads();
			u32 rx = bx + ix - cx, ry = by + iy - cy, tx = 0, ty = 0;
			for (u32 q = 0; q < 4; ++q) { if (q < wv) { rx += wx[q]; ry += wy[q]; } tx += wx[q]; ty += wy[q]; }
			if (MODE && wi < n_words) {
				for (u32 k = 0; k < 16; ++k) {
					const u64 g = p + k;
					if (g >= lo && g < hi) {
						const u32 i = (u32)(g - off); const bool in0 = i < cnt0, is1 = LQ_TWO_DIGIT(W, k) == c1;
						if (MODE == 1) {
							if (in0 && is1) hx[rx++] = i;
							else if (!in0 && !is1) py[ry++] = i;
						} else {
							u32 d;
							if (in0) {
								if (!is1) d = i;
								else { d = rx == 0 ? cnt0 : py[rx - 1] + 1; ++rx; }      // X_t takes the first slot of run t
							} else {
								if (!is1) { d = hx[ry]; ++ry; }                           // Y_t drops into the hole of X_t
								else d = ry < m ? i + 1 : i;                              // run elements shift right by one
							}
							sd[t * 16 + k] = d;
						}
					}
				}
			}
			bx += tx; by += ty;
			__syncthreads();
			if (MODE == 2) {
				// the move itself with the block's threads across the elements: 512 contiguous bytes of records per wave-load (a thread
				// moving its own 16 elements read 8 bytes every 128 and took twice the time of the destination array it saved)
				const u64 g0 = a0 + (u64)w0 * 16;
				for (u32 e = t; e < 256 * 16; e += 256) {
					const u64 g = g0 + e;
					if (g >= lo && g < hi) Rn[off + sd[e]] = Rc[g];               // (a record as one 8-byte word)
				}
				__syncthreads();
			}
		}
		if (MODE == 0 && t == 0) { tcnt[2 * (u64)ti] = bx; tcnt[2 * (u64)ti + 1] = by; }
	}
}

// ---- general pass: the token walk over digit bytes ---------------------------------------------
// The same walk with the sub-array's digits staged in LDS: one block per sub-array; all threads load the
// digit bytes, then one lane walks.  Each bucket keeps {cursor (24 bit), digit of the element under the cursor
// (8 bit)} in a single LDS word, so the serial chain is one LDS round trip per element (~50 ns) instead of a
// global-memory round trip (microseconds under load); the refill of the word is off the critical path.
template <int CAP>
__global__ void k_sort_walk_lds(const SortSeg *segs, const u32 *list, const u32 *n_list_p, const u8 *D, const u32 *hist, const u32 *begs, u32 *dst)
{
	LQ_SHARED u8 dig[CAP + 16];
	LQ_SHARED u32 entry[256];
	LQ_SHARED u32 endb[256];
	const u32 n_list = *n_list_p;
	for (u32 li = blockIdx.x; li < n_list; li += gridDim.x) {
	const u32 sgi = list[li];
	const SortSeg sg = segs[sgi];
	const u32 *cnt = hist + (u64)sgi * 256, *bg = begs + (u64)sgi * 256;
	const u8 *d = D + sg.off;
	u32 *ds = dst + sg.off;
	const u32 len = sg.len;
	LQ_BLOCK_LOOP(t) {
		for (u32 i = t; i < len; i += blockDim.x) dig[i] = d[i];
		if (t == 0) dig[len] = 0;
	}
	LQ_BLOCK_SYNC();
	LQ_BLOCK_LOOP(t) {
		for (u32 c = t; c < 256; c += blockDim.x) {
			const u32 b = bg[c];
			endb[c] = b + cnt[c];
			entry[c] = b | (u32)dig[b < len ? b : len] << 24;
		}
	}
	LQ_BLOCK_SYNC();
	if (threadIdx.x == 0) {
		u32 k = 0, src = 0, l = 0;
		bool carrying = false;
		for (;;) {
			if (!carrying) {
				while (k < 256 && (entry[k] & 0xffffffu) >= endb[k]) ++k;
				if (k >= 256) break;
				const u32 e = entry[k];
				src = e & 0xffffffu; l = e >> 24;                 // the hole this cycle leaves in bucket k
				carrying = true;
			} else {
				const u32 e = entry[l];
				const u32 t = e & 0xffffffu;                      // slot the carried element takes ...
				ds[src] = t;
				src = t;
				entry[l] = (t + 1) | (u32)dig[t + 1] << 24;
				l = e >> 24;                                      // ... and its occupant is carried on
			}
			if (l == k) {
				const u32 c = entry[k] & 0xffffffu;
				ds[src] = c;
				entry[k] = (c + 1) | (u32)dig[c + 1] << 24;
				carrying = false;
			}
		}
	}
	LQ_BLOCK_SYNC();
	}
}

// ---- general pass, long sub-arrays: one walker per wave ("solo"), digit windows in LDS -------------------
// A walk kernel ends when its longest walk ends (measured: 158k trips x ~2.4 us in the 64-walks-per-wave form,
// where every trip waits for some lane's global digit load).  Here a wave runs ONE walk on lane 0: per bucket LDS
// holds the cursor and a 16-byte window of the bucket's digit stream; when the cursor enters the next window,
// lane 0 starts an LDS-DMA (global_load_lds_dwordx4 straight into that bucket's window -- with a single active
// lane the wave-uniform LDS address the instruction needs is simply this lane's) and walks on; the window is only
// waited for if the token returns to that bucket before the data has landed.  The serial chain of a trip is then
// LDS-only.  ~6.5 KiB of LDS per walk lets ~24 walks share a CU.
#ifdef LQ_EMU
#define LQ_DMA_WIN16(gptr, ldsptr) memcpy((ldsptr), (gptr), 16)
#define LQ_WAIT_VM0() ((void)0)
#define LQ_LDS_U8(p) (*(p))
#define LQ_LDS_U32(p) (*(const u32*)(p))
#define LQ_LDS_U64(p) (*(p))
#define LQ_UNI(v) (v)
#else
#define LQ_DMA_WIN16(gptr, ldsptr) \
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)
#define LQ_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// digit read from a DMA-filled window after our own wait (an ordinary read would make hipcc wait for every DMA in flight)
__device__ __forceinline__ u32 lq_lds_u8(const u8 *p)
{
	u32 v;
	const u32 a = (u32)(size_t)(__attribute__((address_space(3))) const void*)p;
	asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
	return v;
}
#define LQ_LDS_U8(p) lq_lds_u8(p)
__device__ __forceinline__ u32 lq_lds_u32(const u8 *p)
{
	u32 v;
	const u32 a = (u32)(size_t)(__attribute__((address_space(3))) const void*)p;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
	return v;
}
#define LQ_LDS_U32(p) lq_lds_u32(p)
// The compiler cannot tell an LDS-DMA landing zone from any other LDS array and drains vmcnt (i.e. waits for every
// outstanding dst store) before each ordinary LDS read in the loop; the walk reads its entries with raw ds_read instead.
__device__ __forceinline__ u64 lq_lds_u64(const u64 *p)
{
	u64 v;
	const u32 a = (u32)(size_t)(__attribute__((address_space(3))) const void*)p;
	asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
	return v;
}
#define LQ_LDS_U64(p) lq_lds_u64(p)
// only lane 0 walks: moving what it reads from LDS into scalar registers makes the walk's control flow scalar branches
// instead of exec-mask sequences
#define LQ_UNI(v) ((u32)__builtin_amdgcn_readfirstlane((int)(v)))
#endif
#define LQ_SOLO_PEND 0x80000000u
__global__ void __launch_bounds__(64)
k_sort_walk_solo(const SortSeg *segs, const u32 *list, const u32 *n_list_p, const u8 *D, const u32 *hist, const u32 *begs, u32 *dst,
                 const CkSeg *cks, const u32 *ckn, const u32 *ck_S, const u32 *ck_slot)
{
	LQ_SHARED __attribute__((aligned(16))) u8 win[256][16];   // DMA landing windows: 16 digits of each bucket's stream
	LQ_SHARED u64 ent[256];                                   // low: cursor | PEND, high: the digits from the cursor to the next 4-byte boundary
	LQ_SHARED u32 endb[256];
	const u32 n_list = *n_list_p;
	for (u32 li = blockIdx.x; li < n_list; li += gridDim.x) {
	// work item: a whole sub-array of the list, or (cks != null) one checkpoint of kernels_ckpt.hpp: start from its cursors,
	// stop when the outer loop reaches the next checkpoint's slot
	u32 sgi, s_end = 0xffffffffu;
	const u32 *start = nullptr;
	if (cks) {
		u32 lo = 0, hi = ckn[1];
		while (hi - lo > 1) { const u32 mid = lo + ((hi - lo) >> 1); if (cks[mid].ck0 <= li) lo = mid; else hi = mid; }
		sgi = cks[lo].sgi;
		start = ck_S + (u64)li * 256;
		if (li + 1 < cks[lo].ck0 + cks[lo].n_ck) s_end = ck_slot[li + 1];
	} else sgi = list[li];
	const SortSeg sg = segs[sgi];
	const u32 *cnt = hist + (u64)sgi * 256, *bg = begs + (u64)sgi * 256;
	const u64 base = sg.off;                                  // D is 16-byte aligned; this sub-array's digits start at D[base]
	const u32 b15 = (u32)(base & 15);
	u32 *ds = dst + sg.off;
	LQ_BLOCK_LOOP(t) {
		for (u32 c = t; c < 256; c += blockDim.x) {
			const u32 b = start ? start[c] : bg[c];
			endb[c] = bg[c] + cnt[c];
			const u8 *w = D + ((base + b) & ~(u64)15);
			for (int i = 0; i < 16; ++i) win[c][i] = w[i];
			const u32 o = (b15 + b) & 15;
			const u32 dq = (*(const u32*)(w + (o & ~3u))) >> (8 * (o & 3));
			ent[c] = (u64)b | (u64)dq << 32;
		}
	}
	LQ_BLOCK_SYNC();
	if (threadIdx.x == 0) {
	// One step over bucket `bk` whose entry is (c, dq): the slot under the cursor is taken; store the entry for cursor c+1.
#define LQ_SOLO_ADVANCE(bk, c, dq) do { \
		const u32 nc_ = (c) + 1, o_ = (b15 + nc_) & 15; \
		if (o_ & 3) ent[bk] = (u64)nc_ | (u64)((dq) >> 8) << 32; \
		else if (o_) ent[bk] = (u64)nc_ | (u64)LQ_UNI(LQ_LDS_U32(&win[bk][o_])) << 32; \
		else { LQ_DMA_WIN16(D + base + nc_, &win[bk][0]); ent[bk] = (u64)(nc_ | LQ_SOLO_PEND); }   /* next window: fetch it asynchronously */ \
	} while (0)
	u32 k = 0;
	for (;;) {
		// START: next bucket with unread slots; the element under its cursor is picked up, leaving a hole there
		while (k < 256 && (LQ_UNI((u32)LQ_LDS_U64(&ent[k])) & ~LQ_SOLO_PEND) >= LQ_UNI(endb[k])) ++k;
		if (k >= 256) break;
		const u64 he = LQ_LDS_U64(&ent[k]);
		u32 hole = LQ_UNI((u32)he), hdq = LQ_UNI((u32)(he >> 32));
		if ((hole & ~LQ_SOLO_PEND) >= s_end) break;               // the next checkpoint's walker takes over from this slot
		if (hole & LQ_SOLO_PEND) { LQ_WAIT_VM0(); hole &= ~LQ_SOLO_PEND; hdq = LQ_UNI(LQ_LDS_U32(&win[k][0])); }
		u32 src = hole, l = hdq & 0xff;
		// CARRY: the carried element takes the slot under its bucket's cursor; that slot's occupant is carried on
		while (l != k) {
			const u64 e = LQ_LDS_U64(&ent[l]);
			u32 c = LQ_UNI((u32)e), dq = LQ_UNI((u32)(e >> 32));
			if (c & LQ_SOLO_PEND) { LQ_WAIT_VM0(); c &= ~LQ_SOLO_PEND; dq = LQ_UNI(LQ_LDS_U32(&win[l][0])); }   // l's window was in flight
			ds[src] = c;
			LQ_SOLO_ADVANCE(l, c, dq);
			src = c; l = dq & 0xff;
		}
		// CLOSE: the hole of bucket k is filled
		ds[src] = hole;
		LQ_SOLO_ADVANCE(k, hole, hdq);
	}
#undef LQ_SOLO_ADVANCE
	LQ_WAIT_VM0();
	}
	LQ_BLOCK_SYNC();
	}
}

