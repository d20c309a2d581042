// longqc_amd/csrc/prim.hpp -- device-wide primitives (stable LSD radix sort, exclusive scan).
// On the GPU these are rocPRIM (ROCm's native primitive library); the test-only emulator build
// (tests/emu) substitutes std:: algorithms with the same contracts.
#pragma once
#include "lq_common.hpp"
#include <stdexcept>
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

// grow-only device buffer
struct DBuf {
	void *p = nullptr; size_t cap = 0;
	void ensure(size_t bytes)
	{
		if (bytes <= cap) return;
		if (p) LQ_HIP_CHECK(hipFree(p));
		p = nullptr; cap = 0;
		size_t want = bytes + bytes / 8 + 256;
		LQ_HIP_CHECK(hipMalloc(&p, want));
		cap = want;
	}
	void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
	template <class T> T *as() const { return (T*)p; }
	~DBuf() { release(); }
	DBuf() {}
	DBuf(const DBuf&) = delete; DBuf &operator=(const DBuf&) = delete;
};

struct Prim {
	DBuf tmp;
	hipStream_t stream = nullptr;

	// out[i] = sum_{j<i} in[j]  (u32 -> u64); returns nothing, total = out[n-1] + in[n-1] (caller reads)
	void exclusive_scan_u32_u64(const u32 *in, u64 *out, size_t n)
	{
		if (n == 0) return;
#ifndef LQ_EMU
		size_t bytes = 0;
		LQ_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, (u64)0, n, rocprim::plus<u64>(), stream));
		tmp.ensure(bytes);
		LQ_HIP_CHECK(rocprim::exclusive_scan(tmp.p, bytes, in, out, (u64)0, n, rocprim::plus<u64>(), stream));
#else
		u64 acc = 0;
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
