// tests/emu/hipemu.hpp -- TEST INFRASTRUCTURE ONLY.
//
// A small stand-in for the slice of the HIP runtime and of the gfx9 wave model that longqc_amd/csrc
// uses, so that the *logic* of every kernel -- including the wave-cooperative ones (ballots, lane
// reads, shuffles, LDS phases separated by __syncthreads) -- can be exercised by the
// `-m "not gpu"` tests in a container without a GPU: the product's sources are compiled as plain
// C++ with -DLQ_EMU -include this file into tests/emu/liblqcov_emu.so.  That library is loaded ONLY
// by tests (tests/conftest.py); longqc_amd/ never looks for it and the shipped liblqcov.so is
// always the hipcc/gfx950 build -- this is not a fallback path.
//
// Model: blocks of a launch run one after another on the calling thread; the threads of a block
// are fibers (own stacks, hand-written context switch).  A fiber runs until it reaches a wave
// collective (__ballot, __shfl*, readlane, readfirstlane, ...) or __syncthreads(), where it waits
// until every *live* lane of its wave (thread of its block) has arrived -- lanes that returned
// from the kernel no longer take part, exactly like an exec mask that lost them.  Collectives must
// therefore sit in wave-uniform control flow (the kernels are written that way for the GPU too);
// lanes meeting at different kinds of collectives abort the process with a message.  __shared__
// is `static` (one block at a time); atomics are plain read-modify-write.  Wave size 64.
//
// What the GPU does differently and the emulator imitates on request (environment, read at every block):
//   * LDS is filled with 0xA5 before every block: a block on the GPU finds what other kernels left in the CU's LDS.
//     (LQ_EMU_NOPOISON=1 turns that off.)
//   * LQ_EMU_ORDER=reverse: the highest ready thread runs first instead of the lowest; LQ_EMU_ORDER=random[:seed]: a random
//     ready one.  LQ_EMU_ORDER_KERNEL=<substring of the kernel's name> / LQ_EMU_ORDER_THREADS=<block size> restrict that
//     to some launches: the way to find which kernel a result depends on.
//   Round 3 found two defects this way that every test on the GPU had passed: a missing barrier after clearing an LDS
//   counter (k_run_list; a fault only beside other lanes' kernels), and tied anchors moved by a partition pass in the order
//   of its atomics (lq_ps_route).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <algorithm>
#include <chrono>
#include <vector>
#include <string>
#include <cctype>
#include <sys/mman.h>
#include <dlfcn.h>
#include <elf.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __restrict__
// LDS: statics collected in one section, so that the emulator can fill all of it with a poison pattern before every block
// (on the GPU a block finds what other kernels left in the CU's LDS, not what the previous block of this kernel wrote)
#define __shared__ static __attribute__((section("emu_lds")))
extern "C" char __start_emu_lds[] __attribute__((weak)), __stop_emu_lds[] __attribute__((weak));

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
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned, const unsigned *) { *s = nullptr; return hipSuccess; }
#define hipStreamDefault 0u
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, void *, unsigned = 0) { return hipSuccess; }
// device memory comes with whatever its last user left in it (large blocks from malloc are fresh zero pages: a kernel that
// counts on zeros would pass here and fail there): filled with 0xCD unless LQ_EMU_NOPOISON is set
inline hipError_t hipMalloc(void **p, size_t n)
{
	*p = std::malloc(n ? n : 1);
	if (!*p) return hipErrorOutOfMemory;
	static const bool off = std::getenv("LQ_EMU_NOPOISON") != nullptr;
	if (!off) std::memset(*p, 0xCD, n ? n : 1);
	return hipSuccess;
}
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{ *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = (size_t)64 << 30; return hipSuccess; }

inline void emu_atomic_hook();                     // LQ_EMU_ORDER=random: now and then another ready thread runs before an atomic
template <class T> inline T atomicAdd(T *p, T v) { emu_atomic_hook(); T o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { emu_atomic_hook(); unsigned long long o = *p; *p = o + v; return o; }
template <class T> inline T atomicSub(T *p, T v) { emu_atomic_hook(); T o = *p; *p = o - v; return o; }
template <class T> inline T atomicOr(T *p, T v) { emu_atomic_hook(); T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAnd(T *p, T v) { emu_atomic_hook(); T o = *p; *p = o & v; return o; }
template <class T> inline T atomicMax(T *p, T v) { emu_atomic_hook(); T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T *p, T v) { emu_atomic_hook(); T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicCAS(T *p, T cmp, T v) { emu_atomic_hook(); T o = *p; if (o == cmp) *p = v; return o; }
template <class T> inline T atomicExch(T *p, T v) { emu_atomic_hook(); T o = *p; *p = v; return o; }
inline void __threadfence() {}
inline void __threadfence_block() {}

struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { uint2 v; v.x = x; v.y = y; return v; }
inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; ++i) if (v >> i & 1) r |= 1u << (31 - i); return r; }

inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline float __fdiv_rn(float a, float b) { return a / b; }

// ---- fibers ---------------------------------------------------------------------------------
extern "C" void emu_ctx_switch(void **save_sp, void *next_sp);
// (one weak definition per translation unit; the linker keeps one)
__asm__(
	".text\n.weak emu_ctx_switch\n.type emu_ctx_switch,@function\n"
	"emu_ctx_switch:\n"
	"	pushq %rbp\n	pushq %rbx\n	pushq %r12\n	pushq %r13\n	pushq %r14\n	pushq %r15\n"
	"	movq %rsp, (%rdi)\n"
	"	movq %rsi, %rsp\n"
	"	popq %r15\n	popq %r14\n	popq %r13\n	popq %r12\n	popq %rbx\n	popq %rbp\n"
	"	ret\n"
	".size emu_ctx_switch,.-emu_ctx_switch\n");

enum { EMU_READY = 0, EMU_WAIT_WAVE = 1, EMU_WAIT_BLOCK = 2, EMU_DONE = 3 };
enum { EMU_MAX_THREADS = 1024, EMU_STACK = 128 << 10 };

struct EmuBlock {
	unsigned n = 0;                              // threads of the running block
	void *sched_sp = nullptr;
	void *sp[EMU_MAX_THREADS];
	unsigned char state[EMU_MAX_THREADS];
	int cur = -1;
	// wave collectives: values published by the lanes, double buffered by the wave's generation
	uint64_t slot[2][EMU_MAX_THREADS];
	unsigned char arrived_flag[EMU_MAX_THREADS];
	int wave_arrived[EMU_MAX_THREADS / 64], wave_live[EMU_MAX_THREADS / 64], wave_gen[EMU_MAX_THREADS / 64], wave_kind[EMU_MAX_THREADS / 64];
	uint64_t wave_mask[2][EMU_MAX_THREADS / 64];  // lanes that took part in the collective of that generation parity
	int block_arrived = 0, block_live = 0;
	char *stacks = nullptr;
	void (*body)(void *) = nullptr; void *body_arg = nullptr;
	emu_idx3 bidx;
	emu_idx3 tidx[EMU_MAX_THREADS];
};
inline EmuBlock g_emu;

inline void emu_die(const char *msg) { std::fprintf(stderr, "hipemu: %s (block %u thread %d)\n", msg, g_emu.bidx.x, g_emu.cur); std::abort(); }

// leave the running fiber: straight into the next ready fiber of the block (round robin), or back to the scheduler
// when nobody is ready (everything finished, or a deadlock for it to report)
// LQ_EMU_ORDER=reverse: the highest ready thread runs first (default: the lowest) -- "thread 0 writes, the others read"
// without a barrier in between goes unnoticed when thread 0 always runs first.  LQ_EMU_ORDER=random[:seed]: whenever a thread
// waits, a random ready one goes on, and before one atomic in eight as well (otherwise a thread runs from one barrier or
// collective to the next without interruption)
inline const char *g_emu_kernel = "";             // name of the kernel being launched (LQ_EMU_ORDER_KERNEL=<substring>: the order applies to it alone)
inline int g_emu_rev = 0;                        // (read from the environment at the start of every block: tests switch it)
inline int emu_reverse() { return g_emu_rev; }
inline uint64_t g_emu_rng = 0;                    // LQ_EMU_ORDER=random[:seed]: a random ready thread runs next
inline unsigned emu_rand(unsigned n) { g_emu_rng ^= g_emu_rng << 13; g_emu_rng ^= g_emu_rng >> 7; g_emu_rng ^= g_emu_rng << 17; return (unsigned)((g_emu_rng >> 11) % n); }
inline void emu_yield()
{
	const int me = g_emu.cur;
	const unsigned n = g_emu.n;
	const int rev = emu_reverse();
	unsigned t = rev == 2 ? emu_rand(n) : rev ? ((unsigned)me + n - 1) % n : ((unsigned)me + 1) % n;
	const bool down = rev == 1 || (rev == 2 && (g_emu_rng & 1));
	for (unsigned i = 0; i < n; ++i, t = down ? (t + n - 1) % n : (t + 1) % n) {
		if (g_emu.state[t] == EMU_READY) {
			if ((int)t == me) return;                    // (released by its own arrival)
			g_emu.cur = (int)t;
			threadIdx = g_emu.tidx[t];
			emu_ctx_switch(&g_emu.sp[me], g_emu.sp[t]);
			return;
		}
	}
	emu_ctx_switch(&g_emu.sp[me], g_emu.sched_sp);
}

inline void emu_atomic_hook() { if (g_emu_rev == 2 && g_emu.cur >= 0 && emu_rand(8) == 0) emu_yield(); }

inline void emu_wave_release(int w)
{
	const unsigned lo = (unsigned)w * 64, hi = std::min(lo + 64, g_emu.n);
	const int par = g_emu.wave_gen[w] & 1;
	uint64_t m = 0;
	for (unsigned t = lo; t < hi; ++t) if (g_emu.arrived_flag[t]) { m |= 1ULL << (t - lo); g_emu.arrived_flag[t] = 0; if (g_emu.state[t] == EMU_WAIT_WAVE) g_emu.state[t] = EMU_READY; }
	g_emu.wave_mask[par][w] = m;
	g_emu.wave_arrived[w] = 0;
	g_emu.wave_gen[w]++;
}

inline void emu_block_release()
{
	for (unsigned t = 0; t < g_emu.n; ++t) if (g_emu.state[t] == EMU_WAIT_BLOCK) g_emu.state[t] = EMU_READY;
	g_emu.block_arrived = 0;
}

// publish v, wait for the live lanes of the wave; afterwards emu_peer(par, lane) reads what a lane published and
// `mask` tells which lanes took part (the others read as 0)
struct EmuEx { int par; unsigned lo; uint64_t mask; };
inline EmuEx emu_wave_exchange(int kind, uint64_t v)
{
	const int me = g_emu.cur, w = me >> 6;
	const int par = g_emu.wave_gen[w] & 1;
	if (g_emu.wave_arrived[w] == 0) g_emu.wave_kind[w] = kind;
	else if (g_emu.wave_kind[w] != kind) emu_die("lanes of one wave met at different collectives (divergent control flow around a collective)");
	g_emu.slot[par][me] = v;
	g_emu.arrived_flag[me] = 1;
	if (++g_emu.wave_arrived[w] == g_emu.wave_live[w]) emu_wave_release(w);
	else { g_emu.state[me] = EMU_WAIT_WAVE; emu_yield(); }
	EmuEx e; e.par = par; e.lo = (unsigned)w * 64; e.mask = g_emu.wave_mask[par][w];
	return e;
}
inline uint64_t emu_peer(const EmuEx &e, unsigned lane) { return (e.mask >> (lane & 63) & 1) ? g_emu.slot[e.par][e.lo + (lane & 63)] : 0; }

inline void __syncthreads()
{
	const int me = g_emu.cur;
	if (g_emu.wave_arrived[me >> 6] != 0) emu_die("__syncthreads() while lanes of the wave wait at a collective");
	if (++g_emu.block_arrived == g_emu.block_live) emu_block_release();
	else { g_emu.state[me] = EMU_WAIT_BLOCK; emu_yield(); }
}
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }

extern "C" inline void emu_fiber_main()
{
	g_emu.body(g_emu.body_arg);
	const int me = g_emu.cur, w = me >> 6;
	g_emu.state[me] = EMU_DONE;
	--g_emu.wave_live[w]; --g_emu.block_live;
	if (g_emu.wave_arrived[w] > 0 && g_emu.wave_arrived[w] == g_emu.wave_live[w]) emu_wave_release(w);
	if (g_emu.block_arrived > 0 && g_emu.block_arrived == g_emu.block_live) emu_block_release();
	emu_yield();
	emu_die("finished fiber resumed");
}

// LDS of the kernel templates: GCC does not honour the section attribute for statics of templates (they are unique global
// objects, "_ZZ<n>k_...E<name>" in the library's dynamic symbol table); found once by reading the library's own ELF headers
struct EmuRegion { char *p; size_t n; std::string kernel; };
inline std::vector<EmuRegion> &emu_template_lds()
{
	static std::vector<EmuRegion> regs;
	static bool done = false;
	if (done) return regs;
	done = true;
	Dl_info di;
	if (!dladdr((void*)&g_emu, &di) || !di.dli_fname) return regs;
	FILE *f = std::fopen(di.dli_fname, "rb");
	if (!f) return regs;
	std::vector<char> img;
	std::fseek(f, 0, SEEK_END); const long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
	img.resize(sz > 0 ? (size_t)sz : 0);
	if (sz <= 0 || std::fread(img.data(), 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); return regs; }
	std::fclose(f);
	const Elf64_Ehdr *eh = (const Elf64_Ehdr*)img.data();
	if (std::memcmp(eh->e_ident, ELFMAG, SELFMAG) != 0 || eh->e_shoff == 0) return regs;
	const Elf64_Shdr *sh = (const Elf64_Shdr*)(img.data() + eh->e_shoff);
	for (unsigned i = 0; i < eh->e_shnum; ++i) {
		if (sh[i].sh_type != SHT_DYNSYM) continue;
		const Elf64_Sym *sym = (const Elf64_Sym*)(img.data() + sh[i].sh_offset);
		const char *str = img.data() + sh[sh[i].sh_link].sh_offset;
		const size_t n = sh[i].sh_size / sizeof(Elf64_Sym);
		for (size_t k = 0; k < n; ++k) {
			if (ELF64_ST_TYPE(sym[k].st_info) != STT_OBJECT || sym[k].st_size == 0 || sym[k].st_shndx == SHN_UNDEF || sym[k].st_shndx >= eh->e_shnum) continue;
			if (!(sh[sym[k].st_shndx].sh_flags & SHF_WRITE)) continue;
			const char *nm = str + sym[k].st_name;
			if (nm[0] != '_' || nm[1] != 'Z' || nm[2] != 'Z') continue;
			const char *q = nm + 3;
			while (*q >= '0' && *q <= '9') ++q;
			if (q == nm + 3 || q[0] != 'k' || q[1] != '_') continue;     // a static of a function k_...
			regs.push_back({(char*)di.dli_fbase + sym[k].st_value, (size_t)sym[k].st_size, std::string(q, (size_t)std::atoi(nm + 3))});
		}
	}
	if (std::getenv("LQ_EMU_VERBOSE")) { size_t tot = 0; for (const EmuRegion &r : regs) tot += r.n; std::fprintf(stderr, "hipemu: %zu LDS arrays of kernel templates, %zu bytes\n", regs.size(), tot); }
	return regs;
}
// on the GPU a block finds what other kernels left in the CU's LDS, not what the previous block of this kernel wrote
inline void emu_poison_lds()
{
	static const bool off = std::getenv("LQ_EMU_NOPOISON") != nullptr;
	if (off) return;
	const unsigned b = g_emu.bidx.x;                             // the first blocks of a launch and every fourth after them: a block
	if (b > 2 && (b & 3) && g_emu.bidx.y == 0) return;          // that counts on what LDS holds does so whatever its number
	if (__start_emu_lds && __stop_emu_lds > __start_emu_lds) std::memset(__start_emu_lds, 0xA5, (size_t)(__stop_emu_lds - __start_emu_lds));
	// of the templates' arrays only those of the kernel being launched (all of them: two thirds of a megabyte per block)
	static const char *last = nullptr;
	static std::vector<EmuRegion> mine;
	if (last != g_emu_kernel) {
		last = g_emu_kernel;
		const char *b = g_emu_kernel;
		while (*b == '(' || *b == ' ') ++b;
		size_t n = 0;
		while (std::isalnum((unsigned char)b[n]) || b[n] == '_') ++n;
		mine.clear();
		for (const EmuRegion &r : emu_template_lds()) if (r.kernel.size() == n && r.kernel.compare(0, n, b, n) == 0) mine.push_back(r);
	}
	for (const EmuRegion &r : mine) std::memset(r.p, 0xA5, r.n);
}

inline void emu_run_block(unsigned nthreads, void (*body)(void *), void *arg)
{
	if (nthreads > EMU_MAX_THREADS) emu_die("block larger than 1024 threads");
	if (!g_emu.stacks) {
		g_emu.stacks = (char*)mmap(nullptr, (size_t)EMU_STACK * EMU_MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (g_emu.stacks == (char*)MAP_FAILED) emu_die("cannot map fiber stacks");
	}
	g_emu.n = nthreads; g_emu.body = body; g_emu.body_arg = arg;
	{ const char *e = std::getenv("LQ_EMU_ORDER"); g_emu_rev = e && e[0] == 'r' ? (e[1] == 'a' ? 2 : 1) : 0;
	  if (g_emu_rev == 2 && g_emu_rng == 0) { const char *c = std::strchr(e, ':'); g_emu_rng = 0x9E3779B97F4A7C15ULL ^ (c ? std::strtoull(c + 1, nullptr, 10) * 0xD6E8FEB86659FD93ULL : 0); if (!g_emu_rng) g_emu_rng = 1; }
	  const char *o = std::getenv("LQ_EMU_ORDER_THREADS"); if (o && (unsigned)std::atoi(o) != nthreads) g_emu_rev = 0;
	  const char *k = std::getenv("LQ_EMU_ORDER_KERNEL"); if (k && !std::strstr(g_emu_kernel, k)) g_emu_rev = 0; }   // (only blocks of that many threads: narrows a finding down)
	emu_poison_lds();
	g_emu.block_arrived = 0; g_emu.block_live = (int)nthreads;
	for (unsigned w = 0; w < (nthreads + 63) / 64; ++w) {
		g_emu.wave_arrived[w] = 0; g_emu.wave_gen[w] = 0;
		g_emu.wave_live[w] = (int)std::min(64u, nthreads - w * 64);
	}
	for (unsigned t = 0; t < nthreads; ++t) {
		// initial frame: six callee-saved registers, then the entry point as the return address (16-byte aligned slot)
		uint64_t *top = (uint64_t*)(g_emu.stacks + (size_t)EMU_STACK * (t + 1));
		top -= 2;                                          // keep the slot 16-byte aligned: entry sees rsp = slot + 8
		*top = (uint64_t)(void*)&emu_fiber_main;
		for (int r = 0; r < 6; ++r) *--top = 0;
		g_emu.sp[t] = top;
		g_emu.state[t] = EMU_READY; g_emu.arrived_flag[t] = 0;
	}
	for (unsigned t = 0; t < nthreads; ++t) g_emu.tidx[t] = { t % blockDim.x, (t / blockDim.x) % blockDim.y, t / (blockDim.x * blockDim.y) };
	for (;;) {                                               // fibers hand over to each other; control returns here when none is ready
		unsigned t = 0;
		if (emu_reverse() == 2) { unsigned r = emu_rand(nthreads), i = 0; while (i < nthreads && g_emu.state[(r + i) % nthreads] != EMU_READY) ++i; if (i == nthreads) break; t = (r + i) % nthreads; }
		else if (emu_reverse()) { t = nthreads; while (t > 0 && g_emu.state[t - 1] != EMU_READY) --t; if (t == 0) break; --t; }
		else { while (t < nthreads && g_emu.state[t] != EMU_READY) ++t; if (t == nthreads) break; }
		g_emu.cur = (int)t;
		threadIdx = g_emu.tidx[t];
		emu_ctx_switch(&g_emu.sched_sp, g_emu.sp[t]);
	}
	for (unsigned t = 0; t < nthreads; ++t)
		if (g_emu.state[t] != EMU_DONE) { g_emu.cur = (int)t; emu_die("deadlock: threads wait at a barrier or collective that the others never reach"); }
	g_emu.cur = -1;
}

// ---- wave intrinsics on top of the exchange ---------------------------------------------------
enum { EMU_K_BALLOT = 1, EMU_K_SHFL, EMU_K_READLANE, EMU_K_FIRST, EMU_K_BPERM };
template <class T> inline uint64_t emu_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, "value too wide"); std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T emu_unbits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }
inline unsigned emu_lane() { return (unsigned)g_emu.cur & 63; }

inline unsigned long long __ballot(int pred)
{
	// the releasing lane could reduce once for all; reading 64 flags is cheap enough
	const EmuEx e = emu_wave_exchange(EMU_K_BALLOT, pred ? 1 : 0);
	unsigned long long r = 0;
	for (uint64_t m = e.mask; m; m &= m - 1) { const int l = __builtin_ctzll(m); if (g_emu.slot[e.par][e.lo + l]) r |= 1ULL << l; }
	return r;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { const EmuEx e = emu_wave_exchange(EMU_K_BALLOT, pred ? 1 : 0); for (uint64_t m = e.mask; m; m &= m - 1) if (!g_emu.slot[e.par][e.lo + __builtin_ctzll(m)]) return 0; return 1; }
template <class T> inline T __shfl(T v, int src, int width = 64)
{
	const EmuEx e = emu_wave_exchange(EMU_K_SHFL, emu_bits(v));
	const int lane = (int)emu_lane(), base = lane & ~(width - 1);
	return emu_unbits<T>(emu_peer(e, (unsigned)(base + (src & (width - 1)))));
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64)
{
	const EmuEx e = emu_wave_exchange(EMU_K_SHFL, emu_bits(v));
	const int lane = (int)emu_lane(), base = lane & ~(width - 1);
	return lane - (int)d >= base ? emu_unbits<T>(emu_peer(e, (unsigned)(lane - (int)d))) : v;
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64)
{
	const EmuEx e = emu_wave_exchange(EMU_K_SHFL, emu_bits(v));
	const int lane = (int)emu_lane(), base = lane & ~(width - 1);
	return lane + (int)d < base + width ? emu_unbits<T>(emu_peer(e, (unsigned)(lane + (int)d))) : v;
}
template <class T> inline T __shfl_xor(T v, int x, int width = 64)
{
	const EmuEx e = emu_wave_exchange(EMU_K_SHFL, emu_bits(v));
	return emu_unbits<T>(emu_peer(e, (unsigned)((int)emu_lane() ^ x)));
}
inline int __builtin_amdgcn_readlane(int v, int lane)
{
	const EmuEx e = emu_wave_exchange(EMU_K_READLANE, emu_bits(v));
	return emu_unbits<int>(emu_peer(e, (unsigned)lane));
}
inline int __builtin_amdgcn_readfirstlane(int v)
{
	const EmuEx e = emu_wave_exchange(EMU_K_FIRST, emu_bits(v));
	return emu_unbits<int>(emu_peer(e, (unsigned)__builtin_ctzll(e.mask)));
}
inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int v)
{
	const EmuEx e = emu_wave_exchange(EMU_K_BPERM, emu_bits(v));
	return emu_unbits<int>(emu_peer(e, (unsigned)(byte_addr >> 2)));
}
inline unsigned __lane_id() { return emu_lane(); }

template <class F> inline void emu_body_thunk(void *p) { (*(F*)p)(); }

template <class K, class... A>
inline void emu_launch(K kern, dim3 g, dim3 b, A... args)
{
	gridDim = { g.x, g.y, g.z }; blockDim = { b.x, b.y, b.z };
	auto body = [&]() { kern(args...); };
	for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by) for (unsigned bx = 0; bx < g.x; ++bx) {
		blockIdx = { bx, by, bz };
		g_emu.bidx = blockIdx;
		emu_run_block(b.x * b.y * b.z, &emu_body_thunk<decltype(body)>, &body);
	}
}
#define LQ_LAUNCH(kern, grid, block, stream, ...) do { g_emu_kernel = #kern; emu_launch(kern, dim3(grid), dim3(block), __VA_ARGS__); } while (0)
