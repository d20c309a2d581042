// tests/emu/hipemu.hpp -- TEST INFRASTRUCTURE ONLY.
//
// A tiny serial stand-in for the slice of the HIP runtime that longqc_amd/csrc uses, so that the
// *logic* of every kernel (indexing, state machines, scans, table probes) can be exercised by the
// `-m "not gpu"` tests in a container without a GPU: the product's .hip sources are compiled as
// plain C++ with -DLQ_EMU -include this file into tests/emu/liblqcov_emu.so.  That library is
// loaded ONLY by tests (tests/conftest.py); longqc_amd/ never looks for it and the shipped
// liblqcov.so is always the hipcc/gfx950 build -- this is not a fallback path.
//
// Model: a kernel launch runs blocks and threads one after another on the calling thread, so
// kernels here must not rely on __syncthreads()/cross-lane exchange inside a launch (round-1
// kernels do not); atomics are plain read-modify-write.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <algorithm>
#include <chrono>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct emu_idx3 { unsigned x, y, z; };
inline emu_idx3 blockIdx, threadIdx, blockDim, gridDim;

typedef int hipError_t;
typedef void *hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{ *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = (size_t)64 << 30; return hipSuccess; }

template <class T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
template <class T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <class T> inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }

inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline float __fdiv_rn(float a, float b) { return a / b; }

template <class K, class... A>
inline void emu_launch(K kern, dim3 g, dim3 b, A... args)
{
	gridDim = { g.x, g.y, g.z }; blockDim = { b.x, b.y, b.z };
	for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by) for (unsigned bx = 0; bx < g.x; ++bx) {
		blockIdx = { bx, by, bz };
		for (unsigned tz = 0; tz < b.z; ++tz) for (unsigned ty = 0; ty < b.y; ++ty) for (unsigned tx = 0; tx < b.x; ++tx) {
			threadIdx = { tx, ty, tz };
			kern(args...);
		}
	}
}
#define LQ_LAUNCH(kern, grid, block, stream, ...) emu_launch(kern, dim3(grid), dim3(block), __VA_ARGS__)
