// longqc_amd/csrc/kernels_isort.hpp -- the engine's device-wide primitives, hand-written for gfx950: the index sort and the
// exclusive scan (rounds 1-5 called rocPRIM for both; since round 6 the engine launches no library kernel).
//
// What the sort replaces: worker_post's radix_sort_128x over a bucket's minimizers (index.c:150-201) -- semantically "every
// minimizer's occurrences grouped by hash, ascending y inside a hash".  The sketch emits a part's minimizers in ascending
// y, so a STABLE sort of (hash, y) pairs on the 2k bits of the hash gives exactly that.
//
// Shape: least-significant-digit radix sort, 8-bit digits, one sweep over the data per digit:
//   k_is_hist   the digit histogram(s) of a pass's input, one per range of tiles (LDS counters, one u64 atomic per non-empty
//               counter and block);
//   k_is_bases  where the output of every (range, digit) begins: the digits' totals scanned, then the ranges in order;
//   k_is_pass   one launch per digit.  A block takes the next tile of 5120 pairs (4-byte keys with 8-byte values), ranks its
//               pairs by digit *stably* -- per wave and 64 consecutive pairs the lanes with the same digit find each other
//               with 8 ballots, the first of them bumps the wave's counter of that digit in LDS, the others read the old
//               value from its lane --, publishes the tile's 256 digit counts as 8-byte {flag, count} granules (relaxed
//               agent-scope stores: written through, visible across XCDs, never torn: MI355X_MICROARCH.md, inter-workgroup
//               visibility), looks back over the predecessors' granules until it meets an inclusive prefix (decoupled
//               look-back: a tile waits for its predecessors' *counting* phases only), then moves first its keys and then its
//               values through one LDS buffer into digit order and writes every digit's run with consecutive lanes on
//               consecutive addresses.
// XCD-aware: a tile gives every digit a run of ~20 pairs (80 B of keys, 160 B of values); neighbouring tiles' runs are
// neighbours in memory, and on eight XCDs with an L2 each a line put together by tiles in start order leaves eight L2s in
// pieces.  So the tiles are dealt in eight contiguous RANGES, one per XCD (HW_REG_XCC_ID; a ticket per range; per-range
// histograms make every range's first tile independent of the ranges before it): the pieces of a line meet in one L2.
// Measured on 1.34 G pairs (configs[2]'s first part): tiles in start order 37.1 ms, ranges 31.9 ms although every pass then
// needs its own histogram read (rocPRIM's onesweep: 32 ms); keys and values taking turns in the staging buffer (tiles of
// 5120 instead of 3840 pairs at three blocks per CU) 27.4 ms.  Smaller tiles at four blocks per CU (36.5), larger ones at two
// (35.1), 5632 pairs with 5 spilled registers (30.6) and skipping a tile's own granule when the predecessor's inclusive one is
// already there (33.2 against 32.2), blocks of 512 threads with the same tiles (32.5 against 27.7) all lost.  So did dropping the sorted digit from the keys in every sweep (u32 -> u16 -> u8, the
// full key put together again in the last sweep from the pair's position: 53 instead of 72 bytes per pair; built, bit-exact, and
// measured at 7.8 + 15.0 + 11.2 ms for the three sweeps against 3 x 8.1: 2- and 1-byte stores cost an instruction each like 8-byte
// ones, the sweep is not byte-bound enough for 26 % fewer bytes to pay for them) -- taken out again.  Placement is only a matter of speed: a block whose range is used up takes a
// tile of the next range; the emulator build runs the same code with blockIdx & 7 as the XCD.
// Bytes per pair and pass: 12 read + 12 written for 4-byte keys (k <= 16) + 4 for the pass's histogram + 0.8 B of granules;
// LDS per block 47 KB (40 KB staging, 4 KB wave counters, 3 KB): three blocks per CU.
#pragma once
#include "lq_common.hpp"

#ifndef LQ_IS_THREADS
#define LQ_IS_THREADS 256
#endif
#define LQ_IS_WAVES   (LQ_IS_THREADS / 64)
#define LQ_IS_MAXPASS 8
#ifndef LQ_IS_FIT_KB
#define LQ_IS_FIT_KB 45
#endif
#ifndef LQ_IS_EMAX
#define LQ_IS_EMAX 20
#endif
#ifndef LQ_IS_FASTPATH
#define LQ_IS_FASTPATH 0
#endif

// {flag, count} granule of one (tile, digit): flag = 2 * pass + 1 (the tile's own count) or 2 * pass + 2 (count of the tile and
// every tile before it); granules are zeroed once per sort, the pass number keeps the passes apart
#define LQ_IS_SPIN_MAX (1u << 21)   // polls of one granule (each a load from L2 or further: seconds in all): a predecessor has always started, so this never comes close
#define LQ_IS_VAL(g)  ((g) & 0x00ffffffffffffffULL)
#define LQ_IS_FLAG(g) ((u32)((g) >> 56))

#ifndef LQ_EMU
__device__ __forceinline__ void lq_is_publish(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 lq_is_peek(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lq_is_nap() { __builtin_amdgcn_s_sleep(2); }
__device__ __forceinline__ void lq_is_stuck() { __builtin_trap(); }                    // a bounded spin that ran out: the launch fails instead of hanging the device
__device__ __forceinline__ u32 lq_xcc_id() { return (u32)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }   // HW_REG_XCC_ID, bits 3:0: the XCD this wave runs on
#else
inline void lq_is_publish(u64 *p, u64 v) { *p = v; }
inline u64 lq_is_peek(const u64 *p) { return *p; }
inline void lq_is_nap() {}
inline void lq_is_stuck() { fprintf(stderr, "hipemu: a look-back waited for a granule that never came\n"); abort(); }
inline u32 lq_xcc_id() { return blockIdx.x & 7u; }
#endif

template <class KT>
__global__ void __launch_bounds__(256)
k_is_hist(const KT *key, u64 n, u32 p0, u32 n_here, u32 n_pass, u32 last_mask, u64 per_block, u64 per_range, unsigned long long *ghist)
{
	// the digits of the passes p0 .. p0 + n_here - 1 of n_pass (all of them up front when the tiles are dealt in one range: the
	// totals do not depend on the order of the keys; pass by pass on each pass's input when there is a histogram per range)
	__shared__ u32 h[LQ_IS_MAXPASS * 256];
	for (u32 i = threadIdx.x; i < n_here * 256; i += 256) h[i] = 0;
	__syncthreads();
	// blockIdx.y: the range of tiles (one per XCD, k_is_pass) whose histogram this is
	const u64 r_lo = (u64)blockIdx.y * per_range, r_hi = r_lo + per_range < n ? r_lo + per_range : n;
	const u64 lo = r_lo + (u64)blockIdx.x * per_block, hi = lo + per_block < r_hi ? lo + per_block : r_hi;
	ghist += (size_t)blockIdx.y * LQ_IS_MAXPASS * 256;
	for (u64 base = lo; base < hi; base += 256 * 4) {
		KT k[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) { const u64 i = base + (u64)j * 256 + threadIdx.x; k[j] = i < hi ? key[i] : (KT)0; }
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const u64 i = base + (u64)j * 256 + threadIdx.x;
			if (i < hi) for (u32 p = p0; p < p0 + n_here; ++p) atomicAdd(&h[(p - p0) * 256 + ((u32)(k[j] >> (8 * p)) & (p + 1 == n_pass ? last_mask : 255u))], 1u);
		}
	}
	__syncthreads();
	for (u32 i = threadIdx.x; i < n_here * 256; i += 256) if (h[i]) atomicAdd(&ghist[p0 * 256 + i], (unsigned long long)h[i]);
}

// per pass (block p): where the output of every (range, digit) begins -- the digits' totals scanned, then the ranges in order
template <int PASSES_MAX>
__global__ void __launch_bounds__(256)
k_is_bases(unsigned long long *ghist, u32 n_ranges, u32 p0)
{
	__shared__ u64 s[256];
	unsigned long long *g = ghist + (size_t)(p0 + blockIdx.x) * 256;
	const u32 t = threadIdx.x;
	u64 mine = 0;
	for (u32 r = 0; r < n_ranges; ++r) mine += g[(size_t)r * PASSES_MAX * 256 + t];
	s[t] = mine;
	__syncthreads();
	for (u32 o = 1; o < 256; o <<= 1) {
		const u64 add = t >= o ? s[t - o] : 0;
		__syncthreads();
		s[t] += add;
		__syncthreads();
	}
	u64 at = s[t] - mine;
	for (u32 r = 0; r < n_ranges; ++r) { const u64 c = g[(size_t)r * PASSES_MAX * 256 + t]; g[(size_t)r * PASSES_MAX * 256 + t] = at; at += c; }
}

// pairs per thread: as many as keep the staged tile at 45 KB (three blocks per CU), 16 at most
template <class KT, class VT, bool PAIRS>
struct LqIsShape { static constexpr int BYTES = PAIRS && sizeof(VT) > sizeof(KT) ? (int)sizeof(VT) : (int)sizeof(KT);   // keys and values take turns in the staging buffer
                   static constexpr int FIT = LQ_IS_FIT_KB * 1024 / (LQ_IS_THREADS * BYTES);
                   static constexpr int E = FIT > LQ_IS_EMAX ? LQ_IS_EMAX : FIT; };

template <class KT, class VT, bool PAIRS>
__global__ void __launch_bounds__(LQ_IS_THREADS)
k_is_pass(const KT *kin, KT *kout, const VT *vin, VT *vout, u64 n, u32 shift, u32 mask, u32 pass, const unsigned long long *gbase, u64 *status, u32 *ticket,
          u32 n_ranges, u32 range_tiles, u32 n_tiles)
{
	constexpr int E = LqIsShape<KT, VT, PAIRS>::E;
	constexpr u32 TILE = LQ_IS_THREADS * E;
	__shared__ u32 wc[LQ_IS_WAVES][256];        // a wave's count of every digit, then its exclusive prefix over the waves before it
	__shared__ u32 toff[256];                   // where the digit's run begins inside the tile
	__shared__ u64 gdel[256];                   // global index of a staged pair = gdel[digit] + its slot in the tile
	__shared__ u32 wsum[LQ_IS_WAVES];
	__shared__ u32 s_tile, s_first;
	__shared__ __attribute__((aligned(16))) char stage[TILE * LqIsShape<KT, VT, PAIRS>::BYTES];   // the tile in digit order: first its keys, then its values
	KT *sk = reinterpret_cast<KT*>(stage);
	VT *sv = reinterpret_cast<VT*>(stage);
	const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	// The tiles are dealt in n_ranges contiguous ranges, one per XCD: a block takes the next tile of the range of the XCD it runs on
	// (a ticket per range: a tile's predecessors have always started), so that the runs neighbouring tiles write for a digit -- a
	// few dozen bytes each -- meet in ONE L2 and leave it as whole lines.  Placement is only a matter of speed: a block whose
	// range is used up takes a tile of the next one (as many blocks as tiles: every block finds one).
	if (tid == 0) {
		const u32 x = lq_xcc_id() % n_ranges;
		u32 r = x, j = 0;
		for (u32 a = 0; a < n_ranges; ++a) {
			r = x + a < n_ranges ? x + a : x + a - n_ranges;
			const u32 first = r * range_tiles, cap = first >= n_tiles ? 0u : (n_tiles - first < range_tiles ? n_tiles - first : range_tiles);
			j = cap ? atomicAdd(&ticket[r], 1u) : 0u;
			if (j < cap) break;
		}
		s_first = r * range_tiles; s_tile = r * range_tiles + j;
	}
	for (u32 i = tid; i < LQ_IS_WAVES * 256; i += LQ_IS_THREADS) (&wc[0][0])[i] = 0;
	__syncthreads();
	const u32 tile = s_tile, first = s_first;
	gbase += (size_t)(first / range_tiles) * LQ_IS_MAXPASS * 256;
	const u64 t0 = (u64)tile * TILE;
	const u32 tn = n - t0 < (u64)TILE ? (u32)(n - t0) : TILE;
	// the wave's E * 64 consecutive pairs, 64 at a time
	KT k[E]; VT v[PAIRS ? E : 1]; u32 r[E];
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const u32 idx = w * (E * 64) + e * 64 + lane;
		k[e] = 0;
		if (PAIRS) v[e] = 0;
		if (idx < tn) { k[e] = kin[t0 + idx]; if (PAIRS) v[e] = vin[t0 + idx]; }
	}
	const u64 below_me = lane ? (~0ULL >> (64 - lane)) : 0ULL;
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const u32 idx = w * (E * 64) + e * 64 + lane;
		const bool valid = idx < tn;
		const u32 d = ((u32)(k[e] >> shift) & mask);
		u64 peers = __ballot(valid);
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const bool bit = (d >> b) & 1;
			const u64 bal = __ballot(valid && bit);
			peers &= bit ? bal : ~bal;
		}
		const u32 below = (u32)__popcll(peers & below_me);
		u32 old = 0;
		if (valid && below == 0) { old = wc[w][d]; wc[w][d] = old + (u32)__popcll(peers); }
		old = __shfl(old, valid ? __ffsll((long long)peers) - 1 : (int)lane);
		r[e] = old + below;
	}
	__syncthreads();
	// per digit (thread d): the waves' counts -> exclusive over the waves, the tile's count, its granule, the look-back
	{
		const bool dthr = tid < 256;                              // (blocks of more than 256 threads: the first four waves own the digits)
		const u32 d = tid & 255;
		u32 run = 0;
		u64 *mine = status + (size_t)tile * 256 + d;
		const u64 f_own = (u64)(2 * pass + 1) << 56, f_all = (u64)(2 * pass + 2) << 56;
		u64 excl = 0;
		bool have_excl = tile == first;
		u32 inc = 0;
		if (dthr) {
#pragma unroll
			for (int i = 0; i < LQ_IS_WAVES; ++i) { const u32 c = wc[i][d]; wc[i][d] = run; run += c; }
			if (LQ_IS_FASTPATH && tile > first) {                    // the predecessor's inclusive prefix is already there: one granule instead of two
				const u64 s = lq_is_peek(mine - 256);
				if (LQ_IS_FLAG(s) == 2 * pass + 2) { excl = LQ_IS_VAL(s); have_excl = true; }
			}
			if (!have_excl) lq_is_publish(mine, f_own | run);
			// exclusive scan of the tile's counts over the digits
			inc = run;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const u32 up = __shfl_up(inc, o); if (lane >= (u32)o) inc += up; }
			if (lane == 63) wsum[w] = inc;
		}
		__syncthreads();
		if (dthr) {
			u32 before = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) if ((u32)i < w) before += wsum[i];
			const u32 off = before + inc - run;
			toff[d] = off;
			if (!have_excl) for (u32 p = tile; p-- > first; ) {
				const u64 *g = status + (size_t)p * 256 + d;
				u64 s = lq_is_peek(g);
				for (u32 spins = 0; LQ_IS_FLAG(s) != 2 * pass + 1 && LQ_IS_FLAG(s) != 2 * pass + 2; ) { if (++spins > LQ_IS_SPIN_MAX) lq_is_stuck(); lq_is_nap(); s = lq_is_peek(g); }
				excl += LQ_IS_VAL(s);
				if (LQ_IS_FLAG(s) == 2 * pass + 2) break;
			}
			lq_is_publish(mine, f_all | (excl + run));
			gdel[d] = (u64)gbase[pass * 256 + d] + excl - off;
		}
	}
	__syncthreads();
	// slot of every pair in digit order (kept in r[]), then the keys through the staging buffer, then the values
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const u32 idx = w * (E * 64) + e * 64 + lane;
		if (idx < tn) {
			const u32 d = ((u32)(k[e] >> shift) & mask);
			r[e] = toff[d] + wc[w][d] + r[e];
		}
	}
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const u32 idx = w * (E * 64) + e * 64 + lane;
		if (idx < tn) sk[r[e]] = k[e];
	}
	__syncthreads();
	u32 dd[(E + 3) / 4];                          // the digits of the slots this thread writes, a byte each: the values go where their keys went
#pragma unroll
	for (int e = 0; e < (E + 3) / 4; ++e) dd[e] = 0;
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const u32 s = e * LQ_IS_THREADS + tid;
		if (s < tn) {
			const KT kk = sk[s];
			const u32 d = (u32)(kk >> shift) & mask;
			dd[e >> 2] |= d << (8 * (e & 3));
			kout[gdel[d] + s] = kk;
		}
	}
	if (PAIRS) {
		__syncthreads();
#pragma unroll
		for (int e = 0; e < E; ++e) {
			const u32 idx = w * (E * 64) + e * 64 + lane;
			if (idx < tn) sv[r[e]] = v[e];
		}
		__syncthreads();
#pragma unroll
		for (int e = 0; e < E; ++e) {
			const u32 s = e * LQ_IS_THREADS + tid;
			if (s < tn) vout[gdel[(dd[e >> 2] >> (8 * (e & 3))) & 255u] + s] = sv[s];
		}
	}
}

// ---- exclusive scan of u32 counts (device-wide, one pass over the data) ----------------------------------------------------------
// The same decoupled look-back with one granule per tile: a tile of 256 x 16 counts is scanned in registers (wave scans over 64
// consecutive counts, carried along the wave's 16 rounds), its total published, the predecessors' granules summed 64 at a time by
// the first wave until one holds an inclusive prefix.  Sums are 64-bit whatever the output type; a granule carries 56 bits of them
// (the engine scans counts of hits, minimizers and records: below 2^40).
// inclusive prefix sum over the wave: DPP operands (row_shr inside a row of 16 lanes, row_bcast across rows) instead of six trips
// through the LDS crossbar (__shfl_up is a ds_bpermute)
#ifndef LQ_EMU
#define LQ_SK_DPP(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rows), 0xf, false)
__device__ __forceinline__ u32 lq_wave_scan_add(u32 v)
{
	v += (u32)LQ_SK_DPP(v, 0x111, 0xf);    // row_shr:1
	v += (u32)LQ_SK_DPP(v, 0x112, 0xf);    // row_shr:2
	v += (u32)LQ_SK_DPP(v, 0x114, 0xf);    // row_shr:4
	v += (u32)LQ_SK_DPP(v, 0x118, 0xf);    // row_shr:8
	v += (u32)LQ_SK_DPP(v, 0x142, 0xa);    // row_bcast:15 -> rows 1 and 3
	v += (u32)LQ_SK_DPP(v, 0x143, 0xc);    // row_bcast:31 -> rows 2 and 3
	return v;
}
#else
static inline u32 lq_wave_scan_add(u32 v) { for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(v, d); if ((int)(threadIdx.x & 63) >= d) v += o; } return v; }
#endif

// One granule per tile: the first wave of a block publishes its tile's total and returns the sum of the totals of every tile
// before it -- the predecessors' granules 64 at a time, nearest first, up to the first that holds an inclusive prefix.  The
// granules are zeroed before the launch; the tiles are taken in ticket order (a tile's predecessors have always started).
__device__ __forceinline__ u64 lq_tile_lookback(u64 *status, u32 tile, u64 total, u32 lane)
{
	u64 *mine = status + tile;
	if (tile > 0 && lane == 0) lq_is_publish(mine, (1ULL << 56) | total);
	u64 excl = 0;
	u32 done = tile == 0;
	for (u32 base = tile; !done; ) {                        // predecessors base - 1 - lane
		const bool have = lane < base;
		u64 s = 0;
		if (have) { const u64 *g = status + (base - 1 - lane); s = lq_is_peek(g); for (u32 spins = 0; LQ_IS_FLAG(s) == 0; ) { if (++spins > LQ_IS_SPIN_MAX) lq_is_stuck(); lq_is_nap(); s = lq_is_peek(g); } }
		const u64 full = __ballot(have && LQ_IS_FLAG(s) == 2);
		const u32 stop = full ? (u32)__ffsll((long long)full) - 1 : 63u;   // the nearest predecessor with an inclusive prefix
		u64 part = have && lane <= stop ? LQ_IS_VAL(s) : 0;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
		excl += part;
		if (full || base <= 64) done = 1; else base -= 64;
	}
	if (lane == 0) lq_is_publish(mine, (2ULL << 56) | (excl + total));
	return excl;
}

#define LQ_SC_E 16
#define LQ_SC_TILE (256 * LQ_SC_E)
template <class TO>
__global__ void __launch_bounds__(256)
k_scan_lookback(const u32 *in, TO *out, u64 n, u64 init, u64 *status, u32 *ticket)
{
	__shared__ u64 wtot[4];
	__shared__ u64 s_excl;
	__shared__ u32 s_tile;
	const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	if (tid == 0) s_tile = atomicAdd(ticket, 1u);
	__syncthreads();
	const u32 tile = s_tile;
	const u64 t0 = (u64)tile * LQ_SC_TILE;
	u32 x[LQ_SC_E]; u64 ex[LQ_SC_E];
#pragma unroll
	for (int e = 0; e < LQ_SC_E; ++e) { const u64 i = t0 + w * (LQ_SC_E * 64) + e * 64 + lane; x[e] = i < n ? in[i] : 0u; }
	u64 carry = 0;
#pragma unroll
	for (int e = 0; e < LQ_SC_E; ++e) {
		// 64 counts of up to 32 bits sum to 38: the two 16-bit halves are scanned apart (each sum fits 32 bits) -- unless the output
		// type says the grand total fits 32 bits
		u64 inc;
		if (sizeof(TO) == 4) inc = lq_wave_scan_add(x[e]);
		else inc = (u64)lq_wave_scan_add(x[e] & 0xffffu) + ((u64)lq_wave_scan_add(x[e] >> 16) << 16);
		ex[e] = carry + inc - x[e];
		carry += __shfl(inc, 63);
	}
	if (lane == 0) wtot[w] = carry;
	__syncthreads();
	u64 before = 0, total = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) { if ((u32)i < w) before += wtot[i]; total += wtot[i]; }
	if (w == 0) { const u64 excl = lq_tile_lookback(status, tile, total, lane); if (lane == 0) s_excl = excl; }
	__syncthreads();
	const u64 add = init + s_excl + before;
#pragma unroll
	for (int e = 0; e < LQ_SC_E; ++e) { const u64 i = t0 + w * (LQ_SC_E * 64) + e * 64 + lane; if (i < n) out[i] = (TO)(add + ex[e]); }
}
