// longqc_amd/csrc/kernels_walk.hpp -- the token walk of one klib pass (ksort.h:99-129) with its state in registers.
//
// A general pass of klib's in-place radix sort over one sub-array is a serial walk (kernels_sort.hpp): the token sits
// on a bucket, the element it carries takes the slot under that bucket's cursor, the slot's occupant is carried on to
// its own bucket, until an element of the cycle's start bucket closes the cycle.  Nothing but the digit bytes drives
// it, and its speed is the latency of one trip times the length of the sub-array: the longest (query, strand)
// sub-arrays have millions of elements, so a trip has to be as short as the hardware allows.
//
// k_sort_walk_solo (kernels_sort.hpp) keeps {cursor, next digits} per bucket in LDS: a trip is an LDS round trip plus
// ~30 VALU instructions on one lane (~130 ns measured on MI355X).  Here the per-bucket state lives in VGPR *lanes* --
// bucket c = lane c & 63 of register group c >> 6 -- and the walker is the wave's scalar unit: v_readlane_b32 /
// v_writelane_b32 move a bucket's cursor and digit window between its lane and SGPRs, the walk's control flow is
// scalar branches, and LDS only holds the 16-byte digit windows that LDS-DMA (global_load_lds_dwordx4) refills in the
// background, touched once per four visits of a bucket.  One walk per wave.  The whole wave stays active (a partial
// EXEC mask would let compiler-made register copies drop the other lanes' buckets); EXEC is narrowed to lane 0 only
// inside the single asm statement of the 4-byte destination store, and for the DMA by a divergent block of its own.
#pragma once
#include "lq_common.hpp"
#include "kernels_sort.hpp"
#include "kernels_ckpt.hpp"

#ifndef LQ_EMU
#define LQ_RL(v, lane) ((u32)__builtin_amdgcn_readlane((int)(v), (int)(lane)))
// (clang has no builtin for it; the LLVM intrinsic is reached by its name, and the compiler routes the lane select through M0)
extern "C" __device__ int lq_writelane_i32(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
#define LQ_WL(v, val, lane) ((v) = (u32)lq_writelane_i32((int)(val), (int)(lane), (int)(v)))
// dst[idx] = val from lane 0 only (idx * 4 must fit 32 bits: sub-arrays below 2^30 elements)
__device__ __forceinline__ void lq_walk_store(u32 *ds, u32 idx, u32 val)
{
	const u32 off = idx << 2;
	asm volatile("s_mov_b64 exec, 1\n\tglobal_store_dword %0, %1, %2\n\ts_mov_b64 exec, -1" : : "v"(off), "v"(val), "s"(ds) : "memory");
}
// the 16-byte DMA on behalf of the walk: lane 0 only, as real control flow -- EXEC narrowed by asm statements around the
// builtin let the scheduler drop a register copy of the bucket state into the gap (seen on gfx950: v_mov of vcur[1] under
// EXEC = 1 lost 63 buckets' cursors); a divergent block of its own holds nothing but the load
__device__ __forceinline__ void lq_walk_dma16(const u8 *g, u8 *lds)
{
	if (threadIdx.x == 0) LQ_DMA_WIN16(g, lds);
}
#define LQ_WALK_REGS(name) u32 name[NG]
#define LQ_WALK_LANE_INIT(name, g, val) name[g] = (val)
#else
// emulator: a "register" is an array of 64 lanes shared by the wave's fibers; lane 0's fiber walks
#define LQ_RL(v, lane) ((v)[(lane)])
#define LQ_WL(v, val, lane) ((v)[(lane)] = (val))
static inline void lq_walk_store(u32 *ds, u32 idx, u32 val) { ds[idx] = val; }
static inline void lq_walk_dma16(const u8 *g, u8 *lds) { LQ_DMA_WIN16(g, lds); }
#define LQ_WALK_REGS(name) static u32 name[NG][64]
#define LQ_WALK_LANE_INIT(name, g, val) name[g][threadIdx.x] = (val)
#endif

// one trip's worth of state access on register group G (bk = uniform bucket number, its lane = bk & 63)
#define LQ_WR_ON_GROUP(bk, ...) do { \
		if (NG == 1) { constexpr int G = 0; __VA_ARGS__ } \
		else if (NG == 2) { if ((bk) < 64) { constexpr int G = 0; __VA_ARGS__ } else { constexpr int G = 1 % NG; __VA_ARGS__ } } \
		else { if ((bk) < 128) { if ((bk) < 64) { constexpr int G = 0; __VA_ARGS__ } else { constexpr int G = 1 % NG; __VA_ARGS__ } } \
		       else { if ((bk) < 192) { constexpr int G = 2 % NG; __VA_ARGS__ } else { constexpr int G = 3 % NG; __VA_ARGS__ } } } \
	} while (0)

// NG = register groups of 64 buckets: 1 (every digit of the pass below 64), 2, or 4.
// Two ways to name the work: a list of sub-arrays (cks == null: every block takes whole walks), or the checkpoints of
// kernels_ckpt.hpp (cks != null: work item = one checkpoint; the walker starts from the checkpoint's cursors ck_S and
// stops when the outer loop reaches the next checkpoint's slot ck_slot).
template <int NG>
__global__ void __launch_bounds__(64)
k_sort_walk_reg(const SortSeg *segs, const u32 *list, const u32 *n_list_p, const u8 *D, const u32 *hist, const u32 *begs, u32 *dst,
                const CkSeg *cks, const u32 *ckn, const u32 *ck_S, const u32 *ck_slot)
{
	LQ_SHARED __attribute__((aligned(16))) u8 win[NG * 64][16];   // DMA landing windows: 16 digits of each bucket's stream
	const u32 n_list = *n_list_p;
	const u32 lane = threadIdx.x;
	for (u32 li = blockIdx.x; li < n_list; li += gridDim.x) {
		u32 sgi, s_end = 0xffffffffu;
		const u32 *start = nullptr;
		if (cks) {
			u32 lo = 0, hi = ckn[1];
			while (hi - lo > 1) { const u32 mid = lo + ((hi - lo) >> 1); if (cks[mid].ck0 <= li) lo = mid; else hi = mid; }
			sgi = cks[lo].sgi;
			start = ck_S + (u64)li * LQ_CK_B;
			if (li + 1 < cks[lo].ck0 + cks[lo].n_ck) s_end = ck_slot[li + 1];
		} else sgi = list[li];
		const SortSeg sg = segs[sgi];
		const u32 *cnt = hist + (u64)sgi * 256, *bg = begs + (u64)sgi * 256;
		const u64 base = sg.off;                              // D is 16-byte aligned; this sub-array's digits start at D[base]
		const u32 b15 = (u32)(base & 15);
		u32 *ds = dst + sg.off;
#ifndef LQ_EMU
		{	// the store's base address goes into an SGPR pair
			const u64 a = (u64)ds;
			ds = (u32*)((u64)LQ_UNI((u32)a) | (u64)LQ_UNI((u32)(a >> 32)) << 32);
		}
#endif
		LQ_WALK_REGS(vcur); LQ_WALK_REGS(vend); LQ_WALK_REGS(vwin);   // per bucket: cursor | PEND, end, digits from the cursor to the next 4-byte boundary
		for (int g = 0; g < NG; ++g) {                        // every lane sets up its own buckets
			const u32 c = (u32)g * 64 + lane;
			const u32 b = (start && c < LQ_CK_B) ? start[c] : bg[c];
			const u8 *w = D + ((base + b) & ~(u64)15);
			*(uint4*)&win[c][0] = *(const uint4*)w;
			const u32 o = (b15 + b) & 15;
			const u32 dq = (*(const u32*)(w + (o & ~3u))) >> (8 * (o & 3));
			LQ_WALK_LANE_INIT(vcur, g, b); LQ_WALK_LANE_INIT(vend, g, bg[c] + cnt[c]); LQ_WALK_LANE_INIT(vwin, g, dq);
		}
		LQ_BLOCK_SYNC();
		{	// wave-uniform: every lane runs the walk, the state sits in the lanes of registers.  (The test emulator runs the same
			// code on every fiber of the wave: the first one scheduled does the work on the shared "registers", the others find
			// every bucket finished -- or the next checkpoint's slot reached -- and leave.)
			u32 k = 0;
			for (;;) {
				// START: next bucket with unread slots; the element under its cursor is picked up, leaving a hole there
				u32 hole = 0, hdq = 0;
				for (; k < (u32)NG * 64; ++k) {
					u32 hend = 0;
					const u32 kl = k & 63;
					LQ_WR_ON_GROUP(k, { hole = LQ_RL(vcur[G], kl); hend = LQ_RL(vend[G], kl); hdq = LQ_RL(vwin[G], kl); });
					if ((hole & ~LQ_SOLO_PEND) < hend) break;
				}
				if (k >= (u32)NG * 64) break;
				if ((hole & ~LQ_SOLO_PEND) >= s_end) break;          // the next checkpoint's walker takes over from this slot
				if (hole & LQ_SOLO_PEND) { LQ_WAIT_VM0(); hole &= ~LQ_SOLO_PEND; hdq = LQ_UNI(LQ_LDS_U32(&win[k][0])); }
				u32 src = hole, l = hdq & 0xff;
				// CARRY: the carried element takes the slot under its bucket's cursor; that slot's occupant is carried on
				while (l != k) {
					const u32 ll = l & 63;
					u32 nl = 0, nsrc = 0;
					LQ_WR_ON_GROUP(l, {
						u32 c = LQ_RL(vcur[G], ll), dq = LQ_RL(vwin[G], ll);
						if (c & LQ_SOLO_PEND) { LQ_WAIT_VM0(); c &= ~LQ_SOLO_PEND; dq = LQ_UNI(LQ_LDS_U32(&win[l][0])); }   // l's window was in flight
						lq_walk_store(ds, src, c);
						const u32 nc = c + 1, o = (b15 + nc) & 15;
						if (o & 3) { LQ_WL(vcur[G], nc, ll); LQ_WL(vwin[G], dq >> 8, ll); }
						else if (o) { LQ_WL(vcur[G], nc, ll); const u32 nw = LQ_UNI(LQ_LDS_U32(&win[l][o])); LQ_WL(vwin[G], nw, ll); }
						else { lq_walk_dma16(D + base + nc, &win[l][0]); LQ_WL(vcur[G], nc | LQ_SOLO_PEND, ll); }   // next window: fetched asynchronously
						nsrc = c; nl = dq & 0xff;
					});
					src = nsrc; l = nl;
				}
				// CLOSE: the hole of bucket k is filled
				lq_walk_store(ds, src, hole);
				{
					const u32 kl = k & 63;
					const u32 nc = hole + 1, o = (b15 + nc) & 15;
					LQ_WR_ON_GROUP(k, {
						if (o & 3) { LQ_WL(vcur[G], nc, kl); LQ_WL(vwin[G], hdq >> 8, kl); }
						else if (o) { LQ_WL(vcur[G], nc, kl); const u32 nw = LQ_UNI(LQ_LDS_U32(&win[k][o])); LQ_WL(vwin[G], nw, kl); }
						else { lq_walk_dma16(D + base + nc, &win[k][0]); LQ_WL(vcur[G], nc | LQ_SOLO_PEND, kl); }
					});
				}
			}
			LQ_WAIT_VM0();
		}
		LQ_BLOCK_SYNC();
	}
}
