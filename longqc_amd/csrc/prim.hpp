// longqc_amd/csrc/prim.hpp -- device buffers and the device-wide primitives (stable LSD radix sort, exclusive scan):
// the host side of kernels_isort.hpp.  Rounds 1-5 called rocPRIM here; since round 6 every kernel the engine launches is
// its own (the test-only emulator build runs the same kernels).
#pragma once
#include "lq_common.hpp"
#include "kernels_isort.hpp"
#include <stdexcept>
#include <cstdlib>
#include <utility>
#include <string>

#ifndef LQ_EMU
#include <cstring>
#define LQ_HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#else
#include <vector>
#include <algorithm>
#include <numeric>
#define LQ_HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	throw std::runtime_error(std::string(#expr) + ": emu error"); } while (0)
#endif

// grow-only device buffer.  Inside the mapping lanes growth must not stall the device: hipFree / hipMalloc wait for every
// stream (rocprofv3 showed 80-120 ms holes in a lane's kernel chain whenever one of its work buffers grew while the other
// lanes were inside long kernels).  A lane thread therefore names its stream in lq_alloc_stream and its buffers are taken
// from and returned to HIP's stream-ordered pool (hipMallocAsync / hipFreeAsync, release threshold raised so that
// returned blocks stay cached).
// A mapping lane's work space: one block per lane, taken when the lanes are made (the device idle: hipMalloc waits for every
// stream), from which the lane's buffers are cut one after the other while a batch is mapped -- no call into the runtime, nothing
// handed back, nothing that another stream could be given while it is still in use.  The lane starts every batch with an empty
// arena (MapLane::drop_arena_buffers): a buffer lives for one batch, like a fresh allocation.  What does not fit is allocated
// the ordinary way.
struct LqArena { char *base = nullptr; size_t size = 0, used = 0; };
inline thread_local LqArena *lq_arena = nullptr;
#ifndef LQ_EMU
inline thread_local hipStream_t lq_alloc_stream = nullptr;
inline void lq_pool_keep_memory(int device)
{
	hipMemPool_t pool = nullptr;
	if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
		uint64_t keep = ~0ULL;
		hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
	}
}
#else
inline thread_local hipStream_t lq_alloc_stream = nullptr;
inline void lq_pool_keep_memory(int) {}
#endif

// time and bytes of device allocations (run_files reports them: a fresh process allocates its whole work space once)
#include <atomic>
#include <chrono>
#include <algorithm>
inline std::atomic<uint64_t> lq_alloc_ns{0}, lq_alloc_bytes{0};
struct LqAllocTimer { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); size_t bytes; explicit LqAllocTimer(size_t b) : bytes(b) {}
	~LqAllocTimer() { lq_alloc_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); lq_alloc_bytes += bytes; } };

struct DBuf {
	void *p = nullptr; size_t cap = 0;
	hipStream_t pool_stream = nullptr;     // not null: the block came from the stream-ordered pool on this stream
	bool in_arena = false;                 // the block is a piece of the calling lane's arena (not freed, just forgotten)
	void ensure(size_t bytes)
	{
		if (bytes <= cap) return;
		release();
#ifndef LQ_EXACT_ALLOC
		if (lq_arena && lq_arena->base) {
			const size_t off = (lq_arena->used + 255) & ~(size_t)255, need = bytes + 256;   // (no head room here: the arena is the lane's whole share, 104 B per anchor of its largest batch, and a piece lives for one batch)
			if (off + need <= lq_arena->size) { p = lq_arena->base + off; lq_arena->used = off + need; cap = need; in_arena = true; return; }
		}
#endif
		LqAllocTimer alloc_timer(bytes);
		size_t want = bytes + bytes / 8 + 256;
#ifdef LQ_EXACT_ALLOC
		want = bytes;                                              // (tools/emu_asan.sh: no slack, so that AddressSanitizer sees the first byte past what was asked for)
#endif
#ifndef LQ_EMU
		if (lq_alloc_stream) {
			want = bytes + bytes / 8 + 4096;
			hipError_t e = hipMallocAsync(&p, want, lq_alloc_stream);
			if (e == hipErrorOutOfMemory) {
				(void)hipGetLastError();
				(void)hipDeviceSynchronize();
				int dev = 0; hipMemPool_t pool = nullptr;
				if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) (void)hipMemPoolTrimTo(pool, 0);
				e = hipMallocAsync(&p, want, lq_alloc_stream);
			}
			LQ_HIP_CHECK(e);
			pool_stream = lq_alloc_stream;
			cap = want;
			return;
		}
#endif
		hipError_t e = hipMalloc(&p, want);
#ifndef LQ_EMU
		if (e == hipErrorOutOfMemory) {                            // blocks cached by the stream-ordered pool count as used: hand them back and try again
			(void)hipGetLastError();
			(void)hipDeviceSynchronize();
			int dev = 0; hipMemPool_t pool = nullptr;
			if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) (void)hipMemPoolTrimTo(pool, 0);
			e = hipMalloc(&p, want);
		}
#endif
		LQ_HIP_CHECK(e);
		cap = want;
	}
	void release()
	{
		if (in_arena) { p = nullptr; cap = 0; in_arena = false; return; }
		if (p) {
#ifndef LQ_EMU
			if (pool_stream) (void)hipFreeAsync(p, lq_alloc_stream ? lq_alloc_stream : pool_stream);
			else
#endif
			(void)hipFree(p);
		}
		p = nullptr; cap = 0; pool_stream = nullptr;
	}
	template <class T> T *as() const { return (T*)p; }
	void swap(DBuf &o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(pool_stream, o.pool_stream); std::swap(in_arena, o.in_arena); }
	~DBuf() { release(); }
	DBuf() {}
	DBuf(const DBuf&) = delete; DBuf &operator=(const DBuf&) = delete;
};

// ranges of tiles of the radix sort's passes (kernels_isort.hpp): one per XCD; LQCOV_IS_RANGES=1: one range, tiles in start order
inline u32 lq_is_ranges()
{
	const char *e = std::getenv("LQCOV_IS_RANGES"); const long x = e ? std::atol(e) : 8;
	return (u32)(x < 1 ? 1 : x > 8 ? 8 : x);
}

inline size_t lq_is_ranges_min_tiles()                        // (tests: LQCOV_IS_RANGES_MIN_TILES=2 brings the ranges to small inputs)
{
	const char *e = std::getenv("LQCOV_IS_RANGES_MIN_TILES"); const long x = e ? std::atol(e) : 64;
	return (size_t)(x < 1 ? 1 : x);
}

struct Prim {
	DBuf tmp;
	hipStream_t stream = nullptr;

	// out[i] = init + sum_{j<i} in[j]; the total is out[n-1] + in[n-1] (callers scan n + 1 counts and read the last)
	template <class TO> void scan_u32(const u32 *in, TO *out, size_t n, u64 init)
	{
		if (n == 0) return;
		const size_t n_tiles = (n + LQ_SC_TILE - 1) / LQ_SC_TILE, ctl = n_tiles * 8 + 64;
		tmp.ensure(ctl);
		LQ_HIP_CHECK(hipMemsetAsync(tmp.p, 0, ctl, stream));
		LQ_LAUNCH(k_scan_lookback<TO>, (u32)n_tiles, 256, stream, in, out, (u64)n, init, tmp.as<u64>(), (u32*)(tmp.as<u64>() + n_tiles));
		LQ_HIP_CHECK(hipGetLastError());
	}
	void exclusive_scan_u32_u64(const u32 *in, u64 *out, size_t n, u64 init = 0) { scan_u32<u64>(in, out, n, init); }
	void exclusive_scan_u32_u32(const u32 *in, u32 *out, size_t n) { scan_u32<u32>(in, out, n, 0); }

	// stable sort of keys (and their values) on key bits [0, end_bit): kernels_isort.hpp.  The inputs are left as they are; one
	// more array of keys and of values is needed when there are two passes or more: ktmp / vtmp, or the scratch of this object
	// (ktmp may be kin itself when the caller has no use for the unsorted keys and the number of passes is odd).
	template <class KT, class VT, bool PAIRS>
	void radix(const KT *kin, KT *kout, const VT *vin, VT *vout, size_t n, unsigned end_bit, KT *ktmp = nullptr, VT *vtmp = nullptr)
	{
		if (n == 0) return;
		if (end_bit == 0 || end_bit > 8 * sizeof(KT)) end_bit = 8 * sizeof(KT);
		const u32 n_pass = (end_bit + 7) / 8, last_mask = (1u << (end_bit - 8 * (n_pass - 1))) - 1;
		constexpr size_t TILE = (size_t)LQ_IS_THREADS * LqIsShape<KT, VT, PAIRS>::E;
		const size_t n_tiles = (n + TILE - 1) / TILE;
		if (n_tiles > 0x7fffffffULL) throw std::domain_error("radix sort: too many tiles");
		if (ktmp == kin && n_pass % 2 == 0) ktmp = nullptr;
		const auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
		const u32 n_ranges = n_tiles >= lq_is_ranges_min_tiles() ? lq_is_ranges() : 1u, range_tiles = (u32)((n_tiles + n_ranges - 1) / n_ranges);
		const size_t b_status = up(n_tiles * 256 * 8), b_hist = (size_t)n_ranges * LQ_IS_MAXPASS * 256 * 8, b_ticket = LQ_IS_MAXPASS * 8 * 4, ctl = b_status + b_hist + b_ticket;
		const size_t b_k = n_pass > 1 && !ktmp ? up(n * sizeof(KT)) : 0, b_v = PAIRS && n_pass > 1 && !vtmp ? up(n * sizeof(VT)) : 0;
		tmp.ensure(ctl + b_k + b_v);
		char *base = tmp.as<char>();
		u64 *status = (u64*)base;
		unsigned long long *ghist = (unsigned long long*)(base + b_status);
		u32 *ticket = (u32*)(base + b_status + b_hist);
		if (b_k) ktmp = (KT*)(base + ctl);
		if (b_v) vtmp = (VT*)(base + ctl + b_k);
		LQ_HIP_CHECK(hipMemsetAsync(base, 0, ctl, stream));
		const size_t per_range = (size_t)range_tiles * TILE, hb = std::min<size_t>(4096 / n_ranges, (per_range + 1023) / 1024), per = ((per_range + hb - 1) / hb + 1023) / 1024 * 1024;
		const dim3 hgrid((u32)((per_range + per - 1) / per), n_ranges);
		if (n_ranges == 1) {
			LQ_LAUNCH(k_is_hist<KT>, hgrid, 256, stream, kin, (u64)n, 0u, n_pass, n_pass, last_mask, (u64)per, (u64)per_range, ghist);
			LQ_LAUNCH(k_is_bases<LQ_IS_MAXPASS>, n_pass, 256, stream, ghist, n_ranges, 0u);
		}
		const KT *ki = kin; const VT *vi = vin;
		for (u32 p = 0; p < n_pass; ++p) {
			const bool to_out = (n_pass - 1 - p) % 2 == 0;
			KT *ko = to_out ? kout : ktmp; VT *vo = to_out ? vout : vtmp;
			if (n_ranges > 1) {                                     // a histogram per range of tiles: of this pass's input order
				LQ_LAUNCH(k_is_hist<KT>, hgrid, 256, stream, ki, (u64)n, p, 1u, n_pass, last_mask, (u64)per, (u64)per_range, ghist);
				LQ_LAUNCH(k_is_bases<LQ_IS_MAXPASS>, 1, 256, stream, ghist, n_ranges, p);
			}
			LQ_LAUNCH((k_is_pass<KT, VT, PAIRS>), (u32)n_tiles, LQ_IS_THREADS, stream, ki, ko, vi, vo, (u64)n, 8 * p, p + 1 == n_pass ? last_mask : 255u, p, ghist, status, ticket + 8 * p, n_ranges, range_tiles, (u32)n_tiles);
			ki = ko; vi = vo;
		}
		LQ_HIP_CHECK(hipGetLastError());
	}

	void sort_pairs_u64(const u64 *kin, u64 *kout, const u64 *vin, u64 *vout, size_t n, unsigned end_bit, u64 *ktmp = nullptr) { radix<u64, u64, true>(kin, kout, vin, vout, n, end_bit, ktmp); }
	void sort_pairs_u32_u64(const u32 *kin, u32 *kout, const u64 *vin, u64 *vout, size_t n, unsigned end_bit, u32 *ktmp = nullptr) { radix<u32, u64, true>(kin, kout, vin, vout, n, end_bit, ktmp); }
	void sort_pairs_u32_u32(const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n) { radix<u32, u32, true>(kin, kout, vin, vout, n, 32); }
	void sort_keys_u32(const u32 *kin, u32 *kout, size_t n) { radix<u32, u32, false>(kin, kout, nullptr, nullptr, n, 32); }
};
