// longqc_amd/csrc/kernels_sketch.hpp -- read packing and (w,k)-minimizer extraction on gfx950.
//
// What it computes: exactly the list mm_sketch emits (reference sketch.c:76-142: hash64 :27-37,
// canonical strand :105-108, palindrome skip :107, window ring :115, first-window ties :116-121,
// new-min / leave-window rules :122-137, final flush :140-141, HPC runs :93-104), in the same
// order, for every read of a read set.
//
// How (MI355X): reads sit in HBM 2-bit packed (+1 "ambiguous" bit/base), chunk-aligned.  One
// thread owns the loop iterations that *start* inside one 128-base chunk; the 64 lanes of a wave
// therefore stream 64 consecutive chunks = 3 KiB of packed bases with perfectly coalesced 16-B
// loads.  mm_sketch is a sequential state machine, but its state is a bounded function of recent
// history: the last k valid bases (k-mers), min(l, w+k) (emission thresholds) and the last w ring
// slots.  A thread re-creates it by running the machine silently over a short halo before its
// chunk and proving convergence (see sk_warm below); if the proof fails (palindrome / N-rich
// context) the halo is extended, down to the read start where the state is known exactly.
// Two passes (count, exclusive scan, emit) give every thread its output offset, so the minimizer
// array comes out in (read, position) order == the reference's emission order, deterministically.
#pragma once
#include "lq_common.hpp"
#include "kernels_isort.hpp"   // lq_wave_scan_add

__device__ __forceinline__ int lq_nt4(u8 c)
{	// seq_nt4_table (sketch.c:8-25): A/C/G/T/U in either case, raw 0..3; everything else ambiguous
	if (c < 4) return c;
	switch (c | 0x20) {
	case 'a': return 0;
	case 'c': return 1;
	case 'g': return 2;
	case 't': case 'u': return 3;
	}
	return 4;
}

// largest r with off[r] <= v  (off has n+1 non-decreasing entries, off[0] = 0, v < off[n])
__device__ __forceinline__ u32 lq_find_seg(const u64 *off, u32 n, u64 v)
{
	u32 lo = 0, hi = n;         // invariant: off[lo] <= v < off[hi]
	while (hi - lo > 1) {
		u32 mid = lo + ((hi - lo) >> 1);
		if (off[mid] <= v) lo = mid; else hi = mid;
	}
	return lo;
}

// ---- pack: ASCII -> 2-bit codes + ambiguity mask, one thread per 32-base word -------------------
__global__ void k_pack(const u8 *ascii, const u64 *seq_off, const u64 *coff, u32 n_reads, u64 n_words,
                       u64 *codes, u32 *amb)
{
	u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_words) return;
	u32 r = lq_find_seg(coff, n_reads, g >> 2);
	u64 p0 = (g - coff[r] * LQ_CHUNK_WORDS) * 32;
	u64 len = seq_off[r + 1] - seq_off[r];
	const u8 *s = ascii + seq_off[r];
	u64 w = 0; u32 m = 0;
	for (int j = 0; j < 32; ++j) {
		u64 pos = p0 + j;
		int c = pos < len ? lq_nt4(s[pos]) : 4;
		if (c < 4) w |= (u64)c << (2 * j); else m |= 1u << j;
	}
	codes[g] = w; amb[g] = m;
}

// ---- ambiguity bits of reads that hold no ambiguous base: only the positions past a read's end, inside its last chunk ---------
// (lqcov_part_add_packed with amb == NULL: a quarter of the packed reads' bytes stays off PCIe; the words are zero before this runs)
__global__ void k_amb_tails(const u64 *coff, const u32 *rlen, u32 r0, u32 r1, u32 *amb)
{
	const u32 r = r0 + blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= r1) return;
	const u32 len = rlen[r];
	const u64 w0 = coff[r] * LQ_CHUNK_WORDS, nw = (coff[r + 1] - coff[r]) * LQ_CHUNK_WORDS;
	for (u64 wi = len >> 5; wi < nw; ++wi) {
		const u64 p0 = wi * 32;
		amb[w0 + wi] = p0 >= len ? 0xffffffffu : ~0u << (len - p0);   // (as lq_pack_host / k_pack mark them)
	}
}

// ---- the state machine ------------------------------------------------------------------------
struct SkParams { i32 k, w, hpc; u64 mask; u32 shift1; };

// sketch.c:27-37 (Thomas Wang's invertible integer hash, masked to 2k bits).  T = u32 is exact for k <= 16: every step is
// masked to 2k <= 32 bits, the low 32 bits of the sums and left shifts do not depend on the register width, and the right
// shifts act on masked values.
template <class T>
__device__ __forceinline__ T lq_hash(T key, T mask)
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}
__device__ __forceinline__ u64 lq_hash64(u64 key, u64 mask) { return lq_hash<u64>(key, mask); }

struct ReadView {
	const u64 *codes; const u32 *amb; u32 len;
	u32 cw; u64 ccodes; u32 camb;                       // one cached word
	__device__ __forceinline__ void init(const u64 *c, const u32 *a, u32 l) { codes = c; amb = a; len = l; cw = 0xffffffffu; }
	__device__ __forceinline__ int at(u32 pos)
	{
		u32 wi = pos >> 5, b = pos & 31;
		if (wi != cw) { cw = wi; ccodes = codes[wi]; camb = amb[wi]; }
		return ((camb >> b) & 1) ? 4 : (int)((ccodes >> (2 * b)) & 3);
	}
	// is `pos` the start of a loop iteration of mm_sketch in HPC mode? (sketch.c:93-104)
	__device__ __forceinline__ bool run_start(u32 pos)
	{
		if (pos == 0) return true;
		int c = at(pos);
		if (c >= 4) return true;
		return at(pos - 1) != c;
	}
};

// Ring (last w slots) and HPC run queue live outside the struct: in LDS, one column per thread (STRIDE = block
// size, lane-minor, conflict-free while lanes are in step), or in a private array (STRIDE = 1) for w > 16.
// (With the ring as a struct member the dynamic slot index sent it to scratch memory: rocprofv3 counted ~190 GB
// of HBM traffic per launch for a kernel whose algorithmic traffic is 4.4 GB.)
template <int STRIDE>
struct SkState {
	u64 fw, rv, best_x;
	u32 best_y;
	i32 l, slot, best_slot, span;
	i32 rq_front, rq_count;
	u64 *rx;
	u32 *ry;
	i32 *rq;
	__device__ __forceinline__ u64 &RX(int j) { return rx[j * STRIDE]; }
	__device__ __forceinline__ u32 &RY(int j) { return ry[j * STRIDE]; }
	__device__ __forceinline__ i32 &RQ(int j) { return rq[j * STRIDE]; }
	__device__ __forceinline__ void reset(int w)
	{
		fw = rv = 0; best_x = LQ_U64MAX; best_y = 0xffffffffu;
		l = slot = best_slot = span = 0; rq_front = rq_count = 0;
		for (int j = 0; j < w; ++j) { RX(j) = LQ_U64MAX; RY(j) = 0xffffffffu; }
	}
};

struct SkOut {
	u64 n;            // emitted so far by this thread
	u64 *x, *y;       // destination (emit pass) or null
	u64 y_hi;         // rid << 32
	u32 *mask;        // mask mode: one bit per base of this read (bit p of the read's mask words = the minimizer ending at p is emitted)
	u32 *dup_flag;    // mask mode: set if a position is emitted twice (never: see k_sketch_emit_mask)
	u32 pos0;         // mask mode: first base of the chunk at hand; its 128 bits are collected in registers (lo, hi), stored once
	u64 lo, hi;
};

// MODE 0: count, 1: emit (x, y) at the thread's offset, 2: set the emitted position's bit in the read's mask
#define LQ_SK_COUNT 0
#define LQ_SK_EMIT  1
#define LQ_SK_MASK  2
template <int MODE>
__device__ __forceinline__ void sk_push(SkOut &o, bool live, u64 x, u32 y32)
{
	if (!live) return;
	if (MODE == LQ_SK_EMIT) { o.x[o.n] = x; o.y[o.n] = o.y_hi | y32; }
	if (MODE == LQ_SK_MASK) {
		const u32 pos = y32 >> 1, rel = pos - o.pos0;
		if (rel < 64) { const u64 b = 1ULL << rel; if (o.lo & b) atomicOr(o.dup_flag, 1u); o.lo |= b; }
		else if (rel < 128) { const u64 b = 1ULL << (rel - 64); if (o.hi & b) atomicOr(o.dup_flag, 1u); o.hi |= b; }
		else {                                                    // an older slot, in the chunk before: that chunk's owner may be at work on the word
			const u32 bit = 1u << (pos & 31);
			if (atomicOr(&o.mask[pos >> 5], bit) & bit) atomicOr(o.dup_flag, 1u);
		}
	}
	++o.n;
}

// One loop iteration of mm_sketch whose (last) base is at `pos` with code c (4 = ambiguous),
// run = homopolymer run length (1 unless HPC).  Returns true if a ring slot was produced.
template <int STRIDE, int EMIT>
__device__ __forceinline__ bool sk_step(SkState<STRIDE> &s, const SkParams &P, int c, u32 pos, int run, bool live, SkOut &o, bool &was_pal)
{
	const int w = P.w, k = P.k;
	u64 cx = LQ_U64MAX; u32 cy = 0xffffffffu;
	was_pal = false;
	if (c < 4) {
		if (P.hpc) {
			s.RQ((s.rq_count++ + s.rq_front) & 0x1f) = run;
			s.span += run;
			if (s.rq_count > k) { s.span -= s.RQ(s.rq_front++); s.rq_front &= 0x1f; --s.rq_count; }
		} else s.span = s.l + 1 < k ? s.l + 1 : k;
		s.fw = (s.fw << 2 | (u64)c) & P.mask;
		s.rv = (s.rv >> 2) | (3ULL ^ (u64)c) << P.shift1;
		if (s.fw == s.rv) { was_pal = true; return false; }       // sketch.c:107: no slot, l unchanged
		int z = s.fw < s.rv ? 0 : 1;
		++s.l;
		if (s.l >= k && s.span < 256) {
			cx = lq_hash64(z ? s.rv : s.fw, P.mask) << 8 | (u64)s.span;
			cy = pos << 1 | (u32)z;
		}
	} else { s.l = 0; s.rq_count = s.rq_front = 0; s.span = 0; }   // sketch.c:114
	const int slot = s.slot;
	s.RX(slot) = cx; s.RY(slot) = cy;
	if (s.l == w + k - 1 && s.best_x != LQ_U64MAX) {              // sketch.c:116-121
		for (int j = slot + 1; j < w; ++j) if (s.RX(j) == s.best_x && s.RY(j) != s.best_y) sk_push<EMIT>(o, live, s.RX(j), s.RY(j));
		for (int j = 0; j < slot; ++j)     if (s.RX(j) == s.best_x && s.RY(j) != s.best_y) sk_push<EMIT>(o, live, s.RX(j), s.RY(j));
	}
	if (cx <= s.best_x) {                                          // sketch.c:122-124
		if (s.l >= w + k && s.best_x != LQ_U64MAX) sk_push<EMIT>(o, live, s.best_x, s.best_y);
		s.best_x = cx; s.best_y = cy; s.best_slot = slot;
	} else if (slot == s.best_slot) {                              // sketch.c:125-137
		if (s.l >= w + k - 1 && s.best_x != LQ_U64MAX) sk_push<EMIT>(o, live, s.best_x, s.best_y);
		s.best_x = LQ_U64MAX;
		for (int j = slot + 1; j < w; ++j) if (s.best_x >= s.RX(j)) { s.best_x = s.RX(j); s.best_y = s.RY(j); s.best_slot = j; }
		for (int j = 0; j <= slot; ++j)    if (s.best_x >= s.RX(j)) { s.best_x = s.RX(j); s.best_y = s.RY(j); s.best_slot = j; }
		if (s.l >= w + k - 1 && s.best_x != LQ_U64MAX) {
			for (int j = slot + 1; j < w; ++j) if (s.RX(j) == s.best_x && s.RY(j) != s.best_y) sk_push<EMIT>(o, live, s.RX(j), s.RY(j));
			for (int j = 0; j <= slot; ++j)    if (s.RX(j) == s.best_x && s.RY(j) != s.best_y) sk_push<EMIT>(o, live, s.RX(j), s.RY(j));
		}
	}
	s.slot = slot + 1 == w ? 0 : slot + 1;
	return true;
}

// Fetch the loop iteration that starts at i: its code, run length and the index of its last base.
__device__ __forceinline__ void sk_fetch(ReadView &rv, const SkParams &P, u32 i, int &c, int &run, u32 &last)
{
	c = rv.at(i); run = 1; last = i;
	if (P.hpc && c < 4) {
		u32 j = i + 1;
		while (j < rv.len && rv.at(j) == c) ++j;
		run = (int)(j - i); last = j - 1;
	}
}

// Bring `s` to the state mm_sketch has just before the iteration starting at i0 (i0 is an
// iteration start).  Runs the machine silently from a halo start s0 < i0 and accepts the result
// only when it provably equals the true state:
//   * k k-mer updates have happened since s0                      -> fw/rv (and the HPC run queue) exact;
//   * since then, either an ambiguous base reset l (l exact from there on) or lsim >= w+k
//     non-palindromic steps were seen (every threshold test on l then agrees with the true l);
//   * the last w ring slots were all produced in that exact regime.
// Otherwise the halo is widened (x4) until it reaches the read start, where the state is exact.
template <int STRIDE>
__device__ __forceinline__ void sk_warm(SkState<STRIDE> &s, ReadView &rv, const SkParams &P, u32 i0)
{
	u32 halo = 64;
	SkOut none; none.n = 0; none.x = none.y = nullptr; none.y_hi = 0; none.mask = none.dup_flag = nullptr; none.pos0 = 0; none.lo = none.hi = 0;
	for (;;) {
		u32 s0 = i0 > halo ? i0 - halo : 0;
		if (P.hpc) while (s0 > 0 && !rv.run_start(s0)) --s0;
		s.reset(P.w);
		if (s0 == i0) return;
		int nk = 0, lsim = 0, exact_run = 0;
		bool lx = false;
		u32 i = s0;
		while (i < i0) {
			int c, run; u32 last; bool pal;
			sk_fetch(rv, P, i, c, run, last);
			bool slot = sk_step<STRIDE, LQ_SK_COUNT>(s, P, c, last, run, false, none, pal);
			if (c >= 4) { lx = nk >= P.k; lsim = 0; ++exact_run; }   // an N slot is (MAX,MAX) whatever the history
			else {
				++nk;
				if (nk < P.k) exact_run = 0;
				else if (slot) {
					++lsim;
					exact_run = (lx || lsim >= P.k) ? exact_run + 1 : 0;
				}
			}
			i = last + 1;
		}
		if (s0 == 0) return;
		if (nk >= P.k && exact_run >= P.w && (lx || lsim >= P.w + P.k)) return;
		halo = halo >= 4096 ? 0xffffffffu : halo * 4;
	}
}

// count pass: cnt[g] = minimizers decided by chunk g;  emit pass: written at off[g]...; mask pass (no -H): the emitted
// positions' bits set in `mask` (4 words per chunk, zeroed by the caller; k_sketch_emit_mask turns them into the list).
// RCAP <= 16: ring in LDS (block of LQ_SK_BLOCK threads); RCAP = 256: private ring, any w < 256.
#define LQ_SK_BLOCK 256
template <int RCAP, int EMIT, bool HPC>
__global__ void k_sketch(const u64 *codes, const u32 *amb, const u64 *coff, const u32 *rlen, u32 n_reads, u64 n_chunks, u32 kpt,
                         SkParams P, int rid_in_y, u32 *cnt, const u64 *off, u64 *out_x, u64 *out_y, const u8 *dp_owned, u32 *mask, u32 *dup_flag,
                         const u32 *list, const u32 *n_list)
{
	constexpr int STRIDE = RCAP <= 16 ? LQ_SK_BLOCK : 1;
	__shared__ u64 s_rx[RCAP <= 16 ? RCAP : 1][LQ_SK_BLOCK];
	__shared__ u32 s_ry[RCAP <= 16 ? RCAP : 1][LQ_SK_BLOCK];
	__shared__ i32 s_rq[RCAP <= 16 && HPC ? 32 : 1][LQ_SK_BLOCK];   // homopolymer run queue (sketch.c:39-58), -H only
	u64 p_rx[RCAP <= 16 ? 1 : RCAP];
	u32 p_ry[RCAP <= 16 ? 1 : RCAP];
	i32 p_rq[RCAP <= 16 ? 1 : 32];
	// A thread owns kpt consecutive chunks.  The machine's state carries over from one chunk to the next of the same read
	// (the next chunk's first iteration is the one the machine stands at), so the halo is walked once per kpt chunks.
	// With a list (mask mode beside the data-parallel kernel: the chunks that kernel left, k_sketch_unowned): one listed chunk per
	// turn of a thread -- every lane of a wave has a chunk to decide.  (Without it a wave of 64 consecutive chunks held one or two
	// of them, the first chunk of a read and little else: 6.9 ms per 4 Gbases for 1.2 % of the chunks.)
	const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x, n_thr = (u64)gridDim.x * blockDim.x;
	const u64 n_items = list ? (u64)*n_list : (n_chunks + kpt - 1) / kpt;
	SkState<STRIDE> s;
	if (RCAP <= 16) { s.rx = &s_rx[0][threadIdx.x]; s.ry = &s_ry[0][threadIdx.x]; s.rq = &s_rq[0][threadIdx.x]; }
	else { s.rx = p_rx; s.ry = p_ry; s.rq = p_rq; }
	ReadView rv;
	for (u64 item = tid; item < n_items; item += n_thr) {
	const u64 g0 = list ? (u64)list[item] : item * kpt;
	const u32 span = list ? 1u : kpt;
	bool have = false;                                         // s is the machine's state before the iteration that starts at i_next of read r_prev
	u32 r_prev = 0, i_next = 0;
	for (u32 cc = 0; cc < span && g0 + cc < n_chunks; ++cc) {
		const u64 g = g0 + cc;
		if (!list && dp_owned && dp_owned[g]) { have = false; continue; }   // k_sketch_dp decides this chunk
		u32 r;
		if (have && g < coff[r_prev + 1]) r = r_prev;
		else { r = lq_find_seg(coff, n_reads, g); have = false; rv.init(codes + coff[r] * LQ_CHUNK_WORDS, amb + coff[r] * LQ_CHUNK_WORDS, rlen[r]); }
		const u32 len = rv.len;
		const u32 pos0 = (u32)(g - coff[r]) * LQ_CHUNK;
		const u32 pos1 = pos0 + LQ_CHUNK < len ? pos0 + LQ_CHUNK : len;
		u32 i = pos0;
		if (P.hpc) while (i < len && !rv.run_start(i)) ++i;        // first iteration this chunk owns
		SkOut o; o.n = 0; o.y_hi = rid_in_y ? (u64)r << 32 : 0;
		o.x = o.y = nullptr; o.mask = nullptr; o.dup_flag = dup_flag;
		if (EMIT == LQ_SK_EMIT) { o.x = out_x + off[g]; o.y = out_y + off[g]; }
		o.pos0 = pos0; o.lo = o.hi = 0;
		if (EMIT == LQ_SK_MASK) o.mask = mask + coff[r] * LQ_CHUNK_WORDS;
		if (i < pos1) {
			if (!(have && i_next == i)) sk_warm<STRIDE>(s, rv, P, i);
			while (i < pos1) {
				int c, run; u32 last; bool pal;
				sk_fetch(rv, P, i, c, run, last);
				sk_step<STRIDE, EMIT>(s, P, c, last, run, true, o, pal);
				i = last + 1;
			}
			if (i >= len && s.best_x != LQ_U64MAX)                    // this thread ran the read's last iteration: sketch.c:140-141
				sk_push<EMIT>(o, true, s.best_x, s.best_y);
			have = true; r_prev = r; i_next = i;
		}
		if (EMIT == LQ_SK_COUNT) cnt[g] = (u32)o.n;
		if (EMIT == LQ_SK_MASK) {                                    // (atomics: the steps of the next chunk may emit slots of this one)
			u32 *mw = o.mask + (pos0 >> 5);
			const u32 w4[4] = { (u32)o.lo, (u32)(o.lo >> 32), (u32)o.hi, (u32)(o.hi >> 32) };
			for (int j = 0; j < 4; ++j) if (w4[j] && (atomicOr(&mw[j], w4[j]) & w4[j])) atomicOr(dup_flag, 1u);
		}
	}
	}
}

// the chunks k_sketch_dp_mask / k_sketch_dp_fast left to the machine, as a list (in no particular order: what the machine decides are bits of a mask)
__global__ void k_sketch_unowned(const u8 *owned, u64 n_chunks, u32 *list, u32 *n_list)
{
	const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const bool un = g < n_chunks && !owned[g];
	const u64 b = __ballot(un);
	if (!b) return;                                            // (uniform)
	const u32 lane = threadIdx.x & 63, lead = (u32)__ffsll((unsigned long long)b) - 1;
	u32 base = 0;
	if (lane == lead) base = atomicAdd(n_list, (u32)__popcll(b));
	base = __shfl(base, (int)lead);
	if (un) list[base + (u32)__popcll(b & ((1ULL << lane) - 1))] = (u32)g;
}

// ---- the same list, decided data-parallel where the machine is memoryless ------------------------------------------
// Inside an N-free stretch, LQ_DP_HALO bases or more away from the read start, with at least w + k - 1 non-palindromic
// positions in the halo, every threshold test on l holds (sketch.c:116-137 with l >= w + k) and every ring slot is a real
// k-mer: mm_sketch is then a sliding-window minimum over the non-palindromic positions ("slots") with its tie rules, and
// what step t emits depends on the values of slots t - w .. t only.  With m = the newest minimal slot of [t - w, t - 1]:
//   v_t <= v_m              -> emit m                                   (sketch.c:122-124)
//   else m == t - w         -> emit m, then with m' = the newest minimal slot of [t - w + 1, t] every other slot of that
//                              window with the value of m', oldest first  (sketch.c:125-137)
// and the read's last step is followed by the minimum of its window (sketch.c:140-141).
// One block per tile of LQ_DPT_CH chunks of one read (tile list: toff[r] = tiles of the reads before r) plus a 64-base halo:
// 5 consecutive positions per thread (their k-mers come out of one 32-base window of the packed codes, one bit reversal for
// all five), hashes compared as such (x = hash << 8 | k orders like the hash), compaction of the slots in LDS, and every
// step's emissions set as bits of the emitted positions (LDS, then one atomicOr per mask word).  What is decided here is
// *which positions are emitted*; k_sketch_emit_mask makes (x, y) of them.
// A tile does not qualify if an N is within reach or the halo holds too few slots (AT repeats): owned[g] = 0 for its chunks
// and k_sketch<mask> decides them, as it does every read's first chunk (l < w + k there).  Needs w <= 16, w + k - 1 <= 48, no -H.
#define LQ_DP_HALO 64
#define LQ_DPT_CH 12
#define LQ_DPT_PER 5
#define LQ_DPT_N (LQ_DPT_CH * LQ_CHUNK + LQ_DP_HALO)          // 1600 positions
#define LQ_DPT_THREADS (LQ_DPT_N / LQ_DPT_PER)                // 320
#define LQ_DPT_WAVES (LQ_DPT_THREADS / 64)
__device__ __forceinline__ u64 lq_rev2(u64 x)
{	// the 2-bit groups of x in reverse order
	x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
	x = ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((x & 0x0f0f0f0f0f0f0f0fULL) << 4);
	x = ((x >> 8) & 0x00ff00ff00ff00ffULL) | ((x & 0x00ff00ff00ff00ffULL) << 8);
	x = ((x >> 16) & 0x0000ffff0000ffffULL) | ((x & 0x0000ffff0000ffffULL) << 16);
	return x >> 32 | x << 32;
}

// 32 bases of a read starting at base `lo` (2-bit codes, first base lowest) and their ambiguity bits
__device__ __forceinline__ void lq_window32(const u64 *cw, const u32 *aw, u32 lo, u64 &raw, u32 &am)
{
	const u32 wi = lo >> 5, sh = lo & 31;
	raw = cw[wi] >> (2 * sh);
	am = aw[wi] >> sh;
	if (sh) { raw |= cw[wi + 1] << (64 - 2 * sh); am |= aw[wi + 1] << (32 - sh); }
}

// read of every tile of k_sketch_dp_mask and of the first chunk of every group of LQ_EM_CH chunks of k_sketch_emit_mask: one
// thread per read fills the entries of its tiles / of the groups that start inside it.  (A block that looks its read up by
// binary search spends ~20 dependent loads -- longer than the rest of its work -- before it can touch a base.)
#define LQ_EM_CH 32
__global__ void k_sketch_owners(const u64 *coff, const u64 *toff, u32 n_reads, u32 *tile_rid, u32 *group_rid)
{
	const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	if (tile_rid) for (u64 t = toff[r]; t < toff[r + 1]; ++t) tile_rid[t] = r;
	for (u64 g = (coff[r] + LQ_EM_CH - 1) / LQ_EM_CH; g * LQ_EM_CH < coff[r + 1]; ++g) group_rid[g] = r;
}

template <class HT, int W>   // HT: u32 for k <= 16 (hashes and k-mers in one register), u64 otherwise; W: the window as a constant
__global__ void __launch_bounds__(LQ_DPT_THREADS)   // (5, 10: the presets'; the window scans below unroll), 0 = P.w at run time
k_sketch_dp_mask(const u64 *codes, const u32 *amb, const u64 *coff, const u32 *rlen, const u64 *toff, const u32 *tile_rid, u32 n_reads, u64 tile0 /* tiles [tile0, n_tiles) */, u64 n_tiles, SkParams P,
                 u8 *owned, u32 *mask, u32 *dup_flag)
{
	__shared__ HT V[LQ_DPT_N];
	__shared__ u16 PS[LQ_DPT_N];
	__shared__ u32 lmask[LQ_DPT_N / 32];
	__shared__ u32 wsum[LQ_DPT_WAVES], whalo[LQ_DPT_WAVES], bad;
	const u32 t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const i32 w = W ? W : P.w, k = P.k;
	for (u64 T = tile0 + blockIdx.x; T < n_tiles; T += gridDim.x) {
		const u32 r = tile_rid[T];
		const u32 len = rlen[r];
		const u64 g0 = coff[r] + (T - toff[r]) * LQ_DPT_CH;        // first chunk of the tile
		const u32 n_ch = (u32)(coff[r + 1] - g0 < LQ_DPT_CH ? coff[r + 1] - g0 : LQ_DPT_CH);
		const u32 p0 = (u32)(T - toff[r]) * LQ_DPT_CH * LQ_CHUNK;
		const u32 a = p0 < LQ_CHUNK ? LQ_CHUNK : p0;              // the read's first chunk is the machine's
		const u32 b = p0 + n_ch * LQ_CHUNK < len ? p0 + n_ch * LQ_CHUNK : len;
		if (a >= b) { if (t < n_ch) owned[g0 + t] = 0; continue; }   // (uniform)
		const u64 *cw = codes + coff[r] * LQ_CHUNK_WORDS;
		const u32 *aw = amb + coff[r] * LQ_CHUNK_WORDS;
		const u32 base = a - LQ_DP_HALO;                          // position of index 0 (a multiple of 64)
		if (t == 0) bad = 0;
		if (t < LQ_DPT_N / 32) lmask[t] = 0;
		__syncthreads();
		// this thread's five positions base + 5 t + j; their k-mers are [p - k + 1, p]
		const u32 i0 = LQ_DPT_PER * t, q0 = base + i0;
		HT h[LQ_DPT_PER];
		const HT kmask = (HT)P.mask;
		u32 sl = 0;                                               // bit j: position j is a slot
		if (q0 < b) {
			u64 raw; u32 am;
			lq_window32(cw, aw, q0 - (u32)k + 1, raw, am);
			const u64 R = lq_rev2(raw);
			u32 n_amb = 0;
#pragma unroll
			for (int j = 0; j < LQ_DPT_PER; ++j) {
				h[j] = 0;
				if (q0 + j < b) {
					const HT rj = (HT)(raw >> (2 * j)) & kmask;
					n_amb |= (am >> j) & (u32)((1ULL << k) - 1);
					const HT rv = ~rj & kmask;                         // complement, oldest base lowest: the machine's rv
					const HT fw = (HT)(R >> (2 * (32 - j - k))) & kmask;   // newest base lowest: the machine's fw
					if (fw != rv) { h[j] = lq_hash<HT>(fw < rv ? fw : rv, kmask); sl |= 1u << j; }
				}
			}
			if (n_amb) atomicOr(&bad, 1u);                            // an ambiguous base within reach: the machine's memory matters
		}
		// slot index = number of slots before the position
		const u32 mine = (u32)__popc(sl);
		u32 inc = mine;
		for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
		if (lane == 63) wsum[wv] = inc;
		{	// slots of the halo (indices below 64 = the first 13 threads' positions, the 13th's in part)
			u32 hm = 0;
			for (int j = 0; j < LQ_DPT_PER; ++j) if (i0 + j < LQ_DP_HALO) hm |= 1u << j;
			u32 hs = (u32)__popc(sl & hm);
			for (int o = 32; o > 0; o >>= 1) hs += __shfl_xor(hs, o);
			if (lane == 0) whalo[wv] = hs;
		}
		__syncthreads();
		u32 ts = inc - mine, Tn = 0;
		for (u32 q = 0; q < LQ_DPT_WAVES; ++q) { if (q < wv) ts += wsum[q]; Tn += wsum[q]; }
		const u32 T0 = whalo[0];
		const bool ok = !bad && T0 >= (u32)(w + k - 1) && Tn > T0;    // (uniform; a stretch without a single slot is the machine's)
		if (t < n_ch) owned[g0 + t] = (ok && p0 + t * LQ_CHUNK >= a) ? 1 : 0;
		if (!ok) { __syncthreads(); continue; }
		{
			u32 s = ts;
			for (int j = 0; j < LQ_DPT_PER; ++j) if (sl >> j & 1) { V[s] = h[j]; PS[s] = (u16)(i0 + j); ++s; }
		}
		__syncthreads();
		// emissions of this thread's steps (the steps of the halo belong to whoever owns those positions)
		{
			u32 s = ts;
			for (int j = 0; j < LQ_DPT_PER; ++j) if (sl >> j & 1) {
				if (i0 + j >= LQ_DP_HALO) {
					const HT x = h[j];
					u32 m = s - 1;
					for (u32 u = 2; u <= (u32)w; ++u) if (V[s - u] < V[m]) m = s - u;          // the newest minimal slot of [s - w, s - 1]
					u32 after = m;                                                              // the window's minimum after the step
					if (x <= V[m]) { const u32 e = PS[m]; atomicOr(&lmask[e >> 5], 1u << (e & 31)); after = s; }
					else if (m == s - (u32)w) {
						{ const u32 e = PS[m]; atomicOr(&lmask[e >> 5], 1u << (e & 31)); }
						u32 m2 = s;
						for (u32 u = 1; u < (u32)w; ++u) if (V[s - u] < V[m2]) m2 = s - u;       // the newest minimal slot of [s - w + 1, s]
						for (u32 u = s - (u32)w + 1; u <= s; ++u) if (u != m2 && V[u] == V[m2]) { const u32 e = PS[u]; atomicOr(&lmask[e >> 5], 1u << (e & 31)); }
						after = m2;
					}
					if (s == Tn - 1 && b >= len) { const u32 e = PS[after]; atomicOr(&lmask[e >> 5], 1u << (e & 31)); }   // the read's last step: its window's minimum follows
				}
				++s;
			}
		}
		__syncthreads();
		if (t < LQ_DPT_N / 32) {
			const u32 v = lmask[t];
			if (v) { u32 *gw = mask + coff[r] * LQ_CHUNK_WORDS + (base >> 5) + t; if (atomicOr(gw, v) & v) atomicOr(dup_flag, 1u); }
		}
		__syncthreads();
	}
}

// ---- the same decision with the presets' k and w as constants (round 6) -------------------------------------------------------
// k_sketch_dp_mask spends ~170 lane-instructions per base (profiles/r06_a_sketch_counters.txt: 754 VALU + 304 SALU + 147 LDS
// instructions per thread of five positions) -- 64-bit window and bit reversal, a run-time hash, a rescan of the window in LDS for
// every step behind per-position branches.  With K + 4 <= 16 the five k-mers of a thread are one 32-bit window of a tile whose
// codes were staged in LDS once (and one v_bfrev for all five reverse strands); with 2 K + 4 <= 32 a slot is one word `hash << 4 |
// 15 - i`, i its place among the thread's W + 5 slots, so that "the newest minimal slot of a window" is a plain unsigned minimum
// (v_min3_u32) and the same word with the low bits flipped gives the oldest one: the two differ exactly when the minimum is tied.
// A thread whose five positions and the W slots before them are all slots, that holds no tie where sketch.c:125-137 would list
// one, and that does not run the read's last step goes through straight-line code and leaves with one 15-bit set of emitted
// positions; every other thread (a palindrome nearby: 1 in 4^(K/2) positions; tied minima: low-complexity sequence) walks the
// rules literally as k_sketch_dp_mask does.  Which tiles qualify, the halo, owned[] and the mask are k_sketch_dp_mask's.
__device__ __forceinline__ u32 lq_rev2_32(u32 x)
{	// the 2-bit groups of x in reverse order
	x = __brev(x);
	return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
__device__ __forceinline__ u32 lq_min3(u32 a, u32 b, u32 c) { const u32 m = a < b ? a : b; return m < c ? m : c; }

#define LQ_DPF_CW (LQ_DPT_N / 16 + 4)                          // 16-base words of a tile's codes, from base - 32
template <int K, int W>
__global__ void __launch_bounds__(LQ_DPT_THREADS)
k_sketch_dp_fast(const u64 *codes, const u32 *amb, const u64 *coff, const u32 *rlen, const u64 *toff, const u32 *tile_rid, u32 n_reads, u64 tile0 /* tiles [tile0, n_tiles) */, u64 n_tiles,
                 u8 *owned, u32 *mask, u32 *dup_flag)
{
	static_assert(K >= 2 && K + LQ_DPT_PER - 1 <= 16 && 2 * K + 4 <= 32 && W >= 2 && W + LQ_DPT_PER <= 16 && W + K - 1 <= LQ_DP_HALO - 16, "reach of the constant-k kernel");
	constexpr u32 KM = (1u << (2 * K)) - 1;
	constexpr int NW = W + LQ_DPT_PER;                          // slots a thread looks at: W before its own, its five
	constexpr int NPREV = (W + LQ_DPT_PER - 1) / LQ_DPT_PER;    // threads before this one that hold them when nothing is missing
	__shared__ u32 CW[LQ_DPF_CW];
	__shared__ uint2 VP[LQ_DPT_N];                               // slot -> (hash, index of its position in the tile)
	__shared__ u32 lmask[LQ_DPT_N / 32];
	__shared__ u8 full[LQ_DPT_THREADS + 4];                      // thread t's five positions are all slots (entry t + NPREV)
	__shared__ u32 wsum[LQ_DPT_WAVES], whalo, bad;
	const u32 t = threadIdx.x, lane = t & 63, wv = t >> 6;
	for (u64 T = tile0 + blockIdx.x; T < n_tiles; T += gridDim.x) {
		const u32 r = tile_rid[T];
		const u32 len = rlen[r];
		const u64 g0 = coff[r] + (T - toff[r]) * LQ_DPT_CH;        // first chunk of the tile
		const u32 n_ch = (u32)(coff[r + 1] - g0 < LQ_DPT_CH ? coff[r + 1] - g0 : LQ_DPT_CH);
		const u32 p0 = (u32)(T - toff[r]) * LQ_DPT_CH * LQ_CHUNK;
		const u32 a = p0 < LQ_CHUNK ? LQ_CHUNK : p0;              // the read's first chunk is the machine's
		const u32 b = p0 + n_ch * LQ_CHUNK < len ? p0 + n_ch * LQ_CHUNK : len;
		if (a >= b) { if (t < n_ch) owned[g0 + t] = 0; continue; }   // (uniform)
		const u64 *cw = codes + coff[r] * LQ_CHUNK_WORDS;
		const u32 *aw = amb + coff[r] * LQ_CHUNK_WORDS;
		const u32 base = a - LQ_DP_HALO;                          // position of index 0 (a multiple of 64)
		if (t < 64) {                                              // the first wave: the tile's codes into LDS, "an ambiguous base within reach"
			u32 am = 0;
			if (t < LQ_DPF_CW / 2) {
				const u32 pw = base - 32 + 32 * t;                    // first position of word t
				const u64 wi = pw >> 5, nw = (coff[r + 1] - coff[r]) * LQ_CHUNK_WORDS;
				u64 c = 0;
				if (wi < nw) { c = cw[wi]; am = aw[wi]; }
				CW[2 * t] = (u32)c; CW[2 * t + 1] = (u32)(c >> 32);
				const u32 lo = base - (u32)K + 1;                      // k-mers of [base, b) cover [lo, b)
				if (pw + 32 <= lo || pw >= b) am = 0;
				else { if (pw < lo) am &= ~0u << (lo - pw); if (pw + 32 > b) am &= ~0u >> (pw + 32 - b); }
			}
			const bool any = __ballot(am != 0) != 0;
			if (t == 0) bad = any ? 1u : 0u;
		}
		if (t < LQ_DPT_N / 32) lmask[t] = 0;
		if (t < NPREV) full[t] = 0;
		__syncthreads();
		// this thread's five positions base + 5 t + j; their k-mers are [p - K + 1, p]
		const u32 i0 = LQ_DPT_PER * t, q0 = base + i0;
		u32 h[LQ_DPT_PER];
		u32 sl = 0;                                               // bit j: position j is a slot
		if (q0 < b) {
			const u32 rel = i0 + 33 - (u32)K;                       // q0 - K + 1 - (base - 32)
			const u32 wi = rel >> 4, sh = 2 * (rel & 15);
			const u32 win = (u32)((((u64)CW[wi + 1] << 32) | CW[wi]) >> sh);   // 16 bases from q0 - K + 1 on, first base lowest
			const u32 R = lq_rev2_32(win);                           // the same, last base lowest
#pragma unroll
			for (int j = 0; j < LQ_DPT_PER; ++j) {
				const u32 rv = ((win >> (2 * j)) & KM) ^ KM;           // complement, oldest base lowest: the machine's rv
				const u32 fw = (R >> (2 * (16 - j - K))) & KM;         // newest base lowest: the machine's fw
				h[j] = lq_hash<u32>(fw < rv ? fw : rv, KM);
				if (fw != rv && q0 + j < b) sl |= 1u << j;
			}
		} else {
#pragma unroll
			for (int j = 0; j < LQ_DPT_PER; ++j) h[j] = 0;
		}
		// slot index = number of slots before the position
		const u32 mine = (u32)__popc(sl);
		const u32 inc = lq_wave_scan_add(mine);
		if (lane == 63) wsum[wv] = inc;
		if (t == LQ_DP_HALO / LQ_DPT_PER) whalo = inc - mine + (u32)__popc(sl & ((1u << (LQ_DP_HALO % LQ_DPT_PER)) - 1));   // slots of the halo (indices below 64)
		full[t + NPREV] = sl == (1u << LQ_DPT_PER) - 1 ? 1 : 0;
		__syncthreads();
		u32 ts = inc - mine, Tn = 0;
		for (u32 q = 0; q < LQ_DPT_WAVES; ++q) { if (q < wv) ts += wsum[q]; Tn += wsum[q]; }
		const u32 T0 = whalo;
		const bool ok = !bad && T0 >= (u32)(W + K - 1) && Tn > T0;    // (uniform; a stretch without a single slot is the machine's)
		if (t < n_ch) owned[g0 + t] = (ok && p0 + t * LQ_CHUNK >= a) ? 1 : 0;
		if (!ok) { __syncthreads(); continue; }
		const bool all5 = sl == (1u << LQ_DPT_PER) - 1;
		if (all5) {
#pragma unroll
			for (int j = 0; j < LQ_DPT_PER; ++j) VP[ts + j] = make_uint2(h[j], i0 + j);
		} else {
			u32 s = ts;
			for (int j = 0; j < LQ_DPT_PER; ++j) if (sl >> j & 1) { VP[s] = make_uint2(h[j], i0 + j); ++s; }
		}
		__syncthreads();
		// emissions of this thread's steps (the steps of the halo belong to whoever owns those positions)
		bool prev_full = true;
#pragma unroll
		for (int q = 0; q < NPREV; ++q) prev_full = prev_full && full[t + q];
		bool slow = mine != 0 && i0 + LQ_DPT_PER > LQ_DP_HALO;     // (a thread with steps to run)
		if (all5 && prev_full && i0 >= LQ_DP_HALO && !(ts + LQ_DPT_PER == Tn && b >= len)) {
			// the slots ts - W .. ts + 4 are the positions i0 - W .. i0 + 4: A[i] orders them by (hash, newest first), ^ 15: (hash, oldest first)
			u32 A[NW];
#pragma unroll
			for (int i = 0; i < W; ++i) A[i] = VP[ts - W + i].x << 4 | (u32)(15 - i);
#pragma unroll
			for (int j = 0; j < LQ_DPT_PER; ++j) A[W + j] = h[j] << 4 | (u32)(15 - W - j);
			u32 Mn[LQ_DPT_PER + 1], Mo[LQ_DPT_PER + 1];               // minimum of A[j .. j + W - 1], newest / oldest of the tied
#pragma unroll
			for (int j = 0; j <= LQ_DPT_PER; ++j) {
				u32 mn = A[j], mo = A[j] ^ 15u;
				int i = 1;
				for (; i + 1 < W; i += 2) { mn = lq_min3(mn, A[j + i], A[j + i + 1]); mo = lq_min3(mo, A[j + i] ^ 15u, A[j + i + 1] ^ 15u); }
				for (; i < W; ++i) { mn = mn < A[j + i] ? mn : A[j + i]; mo = mo < (A[j + i] ^ 15u) ? mo : (A[j + i] ^ 15u); }
				Mn[j] = mn; Mo[j] = mo;
			}
			u32 em = 0;                                              // bit i: the position of A[i] is emitted
			bool tie = false;
#pragma unroll
			for (int j = 0; j < LQ_DPT_PER; ++j) {
				const u32 m = Mn[j];                                    // the newest minimal slot of the W slots before step W + j
				const bool ca = h[j] <= (m >> 4);                       // sketch.c:122-124
				const bool cb = !ca && m == A[j];                       // sketch.c:125-137: the minimum leaves the window
				if (ca || cb) em |= 0x8000u >> (m & 15u);
				if (cb && (Mn[j + 1] ^ Mo[j + 1]) != 15u) tie = true;   // the new window's minimum is tied: its other holders are listed
			}
			slow = tie;
			if (em) {
				const u32 e0 = i0 - (u32)W;                             // index of A[0]'s position
				const u64 v = (u64)em << (e0 & 31);
				atomicOr(&lmask[e0 >> 5], (u32)v);
				if ((u32)(v >> 32)) atomicOr(&lmask[(e0 >> 5) + 1], (u32)(v >> 32));
			}
		}
		if (slow) {
			u32 s = ts;
			for (int j = 0; j < LQ_DPT_PER; ++j) if (sl >> j & 1) {
				if (i0 + j >= LQ_DP_HALO) {
					const u32 x = h[j];
					u32 m = s - 1;
					for (u32 u = 2; u <= (u32)W; ++u) if (VP[s - u].x < VP[m].x) m = s - u;          // the newest minimal slot of [s - W, s - 1]
					u32 after = m;                                                                  // the window's minimum after the step
					if (x <= VP[m].x) { const u32 e = VP[m].y; atomicOr(&lmask[e >> 5], 1u << (e & 31)); after = s; }
					else if (m == s - (u32)W) {
						{ const u32 e = VP[m].y; atomicOr(&lmask[e >> 5], 1u << (e & 31)); }
						u32 m2 = s;
						for (u32 u = 1; u < (u32)W; ++u) if (VP[s - u].x < VP[m2].x) m2 = s - u;     // the newest minimal slot of [s - W + 1, s]
						for (u32 u = s - (u32)W + 1; u <= s; ++u) if (u != m2 && VP[u].x == VP[m2].x) { const u32 e = VP[u].y; atomicOr(&lmask[e >> 5], 1u << (e & 31)); }
						after = m2;
					}
					if (s == Tn - 1 && b >= len) { const u32 e = VP[after].y; atomicOr(&lmask[e >> 5], 1u << (e & 31)); }   // the read's last step: its window's minimum follows
				}
				++s;
			}
		}
		__syncthreads();
		if (t < LQ_DPT_N / 32) {
			const u32 v = lmask[t];
			if (v) { u32 *gw = mask + coff[r] * LQ_CHUNK_WORDS + (base >> 5) + t; if (atomicOr(gw, v) & v) atomicOr(dup_flag, 1u); }
		}
		__syncthreads();
	}
}

// cnt[g] = emitted positions of chunk g
__global__ void k_mask_count(const u32 *mask, u64 n_chunks, u32 *cnt)
{
	const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_chunks) return;
	const uint4 m = *(const uint4*)(mask + g * LQ_CHUNK_WORDS);
	cnt[g] = (u32)(__popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w));
}

// The list from the mask: (x, y) of every emitted position, in (read, position) order = the order mm_sketch emits them in.
// (mm_sketch's emissions ascend in position and never repeat one: at any moment the slots of the window that are older
// than the running minimum either had its value -- and were emitted when it was established, sketch.c:116-121,125-137 --
// or a larger one and can never become the minimum; so whatever is emitted later is the minimum itself or newer.  The
// fixtures' lists are strictly ascending, and a bit set twice raises dup_flag.)  No -H here: span = k, the position is the
// k-mer's last base.
// One block per LQ_EM_CH chunks: the set bits become a dense list of positions in LDS (rank = bits set before), then one
// thread per list entry rebuilds the k-mer from the packed codes, hashes it and writes x and y at offset + rank: all lanes
// busy with a hash, the 16-byte outputs contiguous.
#define LQ_EM_THREADS 256
#define LQ_EM_WORDS (LQ_EM_CH * LQ_CHUNK_WORDS)                // mask words of a group: at most one per thread of the first half of the block
__global__ void __launch_bounds__(LQ_EM_THREADS)
k_sketch_emit_mask(const u64 *codes, const u32 *amb, const u64 *coff, const u32 *group_rid, u32 n_reads, u64 n_chunks, SkParams P, int rid_in_y,
                   const u32 *mask, const u64 *off, u64 *out_x, u64 *out_y, u32 *out_key /* not null: the hash alone as well (k <= 16: the index sort's key) */)
{
	static_assert(LQ_EM_WORDS <= LQ_EM_THREADS / 2 && LQ_EM_WORDS % 64 == 0 && LQ_EM_CH <= LQ_EM_THREADS / 2, "one mask word per thread of the block's first waves");
	__shared__ u16 lpos[LQ_EM_CH * LQ_CHUNK];
	__shared__ u32 woff[LQ_EM_WORDS], wbits[LQ_EM_WORDS], wtot[LQ_EM_WORDS / 64], rid[LQ_EM_CH];
	const u32 t = threadIdx.x;
	const i32 k = P.k;
	for (u64 g0 = (u64)blockIdx.x * LQ_EM_CH; g0 < n_chunks; g0 += (u64)gridDim.x * LQ_EM_CH) {
		const u32 n_ch = (u32)(n_chunks - g0 < LQ_EM_CH ? n_chunks - g0 : LQ_EM_CH);
		const u32 n_w = n_ch * LQ_CHUNK_WORDS;
		if (t < LQ_EM_WORDS) {                                    // the first waves: mask words and their exclusive bit counts (inside the wave)
			const u32 v = t < n_w ? mask[g0 * LQ_CHUNK_WORDS + t] : 0;
			const u32 c = (u32)__popc(v);
			const u32 inc = lq_wave_scan_add(c);
			wbits[t] = v; woff[t] = inc - c;
			if ((t & 63) == 63) wtot[t >> 6] = inc;
		} else if (t - LQ_EM_WORDS < n_ch) {                      // the group's first chunk is in read group_rid[..]; the others a few reads on at most
			const u32 c = t - LQ_EM_WORDS;
			u32 r = group_rid[g0 / LQ_EM_CH];
			while (g0 + c >= coff[r + 1]) ++r;
			rid[c] = r;
		}
		__syncthreads();
		u32 n = 0, wbase[LQ_EM_WORDS / 64];
#pragma unroll
		for (u32 q = 0; q < LQ_EM_WORDS / 64; ++q) { wbase[q] = n; n += wtot[q]; }
		for (u32 q = t; q < n_w * 8; q += LQ_EM_THREADS) {          // four positions (a nibble of a mask word) per turn
			const u32 wi = q >> 3, nb = (q & 7) * 4;
			const u32 v = wbits[wi];
			u32 nib = (v >> nb) & 15u;
			if (nib) {
				u32 rk = woff[wi] + (u32)__popc(v & ((1u << nb) - 1));
#pragma unroll
				for (u32 q2 = 0; q2 < LQ_EM_WORDS / 64; ++q2) if ((wi >> 6) == q2) rk += wbase[q2];
				for (; nib; nib &= nib - 1) lpos[rk++] = (u16)(wi * 32 + nb + (u32)__builtin_ctz(nib));
			}
		}
		__syncthreads();
		const u64 o0 = off[g0];
		for (u32 j = t; j < n; j += LQ_EM_THREADS) {
			const u32 pi = lpos[j], ch = pi >> 7;
			const u32 r = rid[ch];
			const u64 c0 = coff[r];
			const u32 pos = (u32)(g0 + ch - c0) * LQ_CHUNK + (pi & (LQ_CHUNK - 1));
			u32 z; u64 hv;
			if (k <= 16) {                                          // the k-mer is inside 16 bases = one 32-bit window of the codes
				const u32 *c32 = (const u32*)(codes + c0 * LQ_CHUNK_WORDS);
				const u32 p = pos - (u32)k + 1, wi = p >> 4, sh = 2 * (p & 15);
				const u32 w0 = c32[wi], w1 = (p & 15) + (u32)k > 16 ? c32[wi + 1] : 0u;   // (the word after only if the k-mer reaches into it: it may lie past the read's chunks)
				const u32 win = (u32)((((u64)w1 << 32) | w0) >> sh);
				const u32 km32 = (u32)P.mask;
				const u32 rv = (win & km32) ^ km32;
				const u32 fw = lq_rev2_32(win << (32 - 2 * k));       // the k groups of the window in reverse order: the machine's fw
				z = fw < rv ? 0u : 1u;
				hv = (u64)lq_hash<u32>(z ? rv : fw, km32);
			} else {
				u64 raw; u32 am;
				lq_window32(codes + c0 * LQ_CHUNK_WORDS, amb + c0 * LQ_CHUNK_WORDS, pos - (u32)k + 1, raw, am);
				raw &= P.mask;
				const u64 rv = ~raw & P.mask;
				const u64 fw = lq_rev2(raw) >> (64 - 2 * k);
				z = fw < rv ? 0u : 1u;
				hv = lq_hash<u64>(z ? rv : fw, P.mask);
			}
			out_x[o0 + j] = hv << 8 | (u64)k;
			out_y[o0 + j] = (rid_in_y ? (u64)r << 32 : 0) | (u64)(pos << 1 | z);
			if (out_key) out_key[o0 + j] = (u32)hv;
		}
		__syncthreads();
	}
}

// per-read minimizer offsets from per-chunk offsets
__global__ void k_read_moff(const u64 *coff, const u64 *chunk_off, u32 n_reads, u64 n_chunks, u64 total, u64 *moff)
{
	u32 r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r > n_reads) return;
	u64 c = coff[r];
	moff[r] = c < n_chunks ? chunk_off[c] : total;
}
