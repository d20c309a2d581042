// Test-only: the host-side pieces of longqc_amd/csrc/sat_replay.hpp behind a C ABI, compiled by tests/test_host.py with g++
// (LQ_EMU: lq_common.hpp without the HIP headers), so that they can be held against the oracle's restatement of klib's sort.
#define LQ_EMU 1
#include "../../longqc_amd/csrc/sat_replay.hpp"
#include <cstring>

extern "C" void shim_klib_sort_128x(uint64_t *xy, size_t n)
{
	std::vector<satreplay::Rec128> v(n);
	for (size_t i = 0; i < n; ++i) { v[i].x = xy[2 * i]; v[i].y = xy[2 * i + 1]; }
	satreplay::klib_sort_128x(v);
	for (size_t i = 0; i < n; ++i) { xy[2 * i] = v[i].x; xy[2 * i + 1] = v[i].y; }
}
extern "C" uint32_t shim_query_hash(const char *name, int32_t qlen, int32_t seed) { return satreplay::query_hash(name, qlen, seed); }
extern "C" uint64_t shim_mix64(uint64_t k) { return satreplay::mix64(k); }
