// longqc_amd/csrc/prim.hpp -- device-wide primitives (stable LSD radix sort, exclusive scan).
// On the GPU these are rocPRIM (ROCm's native primitive library); the test-only emulator build
// (tests/emu) substitutes std:: algorithms with the same contracts.
#pragma once
#include "lq_common.hpp"
#include <stdexcept>
#include <utility>
#include <string>

#ifndef LQ_EMU
#include <cstring>
#include <rocprim/rocprim.hpp>
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

struct Prim {
	DBuf tmp;
	hipStream_t stream = nullptr;

	// out[i] = sum_{j<i} in[j]  (u32 -> u64); returns nothing, total = out[n-1] + in[n-1] (caller reads)
	void exclusive_scan_u32_u64(const u32 *in, u64 *out, size_t n, u64 init = 0)
	{
		if (n == 0) return;
#ifndef LQ_EMU
		size_t bytes = 0;
		LQ_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, init, n, rocprim::plus<u64>(), stream));
		tmp.ensure(bytes);
		LQ_HIP_CHECK(rocprim::exclusive_scan(tmp.p, bytes, in, out, init, n, rocprim::plus<u64>(), stream));
#else
		u64 acc = init;
		for (size_t i = 0; i < n; ++i) { u64 v = in[i]; out[i] = acc; acc += v; }
#endif
	}

	void exclusive_scan_u32_u32(const u32 *in, u32 *out, size_t n)
	{
		if (n == 0) return;
#ifndef LQ_EMU
		size_t bytes = 0;
		LQ_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, (u32)0, n, rocprim::plus<u32>(), stream));
		tmp.ensure(bytes);
		LQ_HIP_CHECK(rocprim::exclusive_scan(tmp.p, bytes, in, out, (u32)0, n, rocprim::plus<u32>(), stream));
#else
		u32 acc = 0;
		for (size_t i = 0; i < n; ++i) { u32 v = in[i]; out[i] = acc; acc += v; }
#endif
	}

	// stable sort of (key,value) pairs on key bits [0, end_bit)
	void sort_pairs_u64(const u64 *kin, u64 *kout, const u64 *vin, u64 *vout, size_t n, unsigned end_bit)
	{
		if (n == 0) return;
#ifndef LQ_EMU
		size_t bytes = 0;
		LQ_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, end_bit, stream));
		tmp.ensure(bytes);
		LQ_HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0u, end_bit, stream));
#else
		std::vector<size_t> idx(n);
		std::iota(idx.begin(), idx.end(), (size_t)0);
		u64 mask = end_bit >= 64 ? ~0ULL : ((1ULL << end_bit) - 1);
		std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return (kin[a] & mask) < (kin[b] & mask); });
		for (size_t i = 0; i < n; ++i) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
#endif
	}

	void sort_pairs_u32_u64(const u32 *kin, u32 *kout, const u64 *vin, u64 *vout, size_t n, unsigned end_bit)
	{
		if (n == 0) return;
#ifndef LQ_EMU
		size_t bytes = 0;
		LQ_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, end_bit, stream));
		tmp.ensure(bytes);
		LQ_HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0u, end_bit, stream));
#else
		std::vector<size_t> idx(n);
		std::iota(idx.begin(), idx.end(), (size_t)0);
		std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return kin[a] < kin[b]; });
		for (size_t i = 0; i < n; ++i) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
#endif
	}

	void sort_pairs_u32_u32(const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n)
	{
		if (n == 0) return;
#ifndef LQ_EMU
		size_t bytes = 0;
		LQ_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, 32u, stream));
		tmp.ensure(bytes);
		LQ_HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0u, 32u, stream));
#else
		std::vector<size_t> idx(n);
		std::iota(idx.begin(), idx.end(), (size_t)0);
		std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return kin[a] < kin[b]; });
		for (size_t i = 0; i < n; ++i) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
#endif
	}

	void sort_keys_u32(const u32 *kin, u32 *kout, size_t n)
	{
		if (n == 0) return;
#ifndef LQ_EMU
		size_t bytes = 0;
		LQ_HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, kin, kout, n, 0u, 32u, stream));
		tmp.ensure(bytes);
		LQ_HIP_CHECK(rocprim::radix_sort_keys(tmp.p, bytes, kin, kout, n, 0u, 32u, stream));
#else
		std::copy(kin, kin + n, kout);
		std::sort(kout, kout + n);
#endif
	}
};
