// longqc_amd/csrc/lq_common.hpp -- shared types of the MI355X coverage engine.
#pragma once
#ifndef LQ_EMU
#include <hip/hip_runtime.h>
#include <cstdio>
// LQCOV_TRACE_LAUNCHES=1: every launch is named on stderr and waited for -- the last name before a GPU fault is the kernel at fault
extern int lq_trace_launches;
#define LQ_LAUNCH(kern, grid, block, stream, ...) do { \
		if (lq_trace_launches) { fprintf(stderr, "[lqcov] launch %s grid %u stream %p\n", #kern, (unsigned)dim3(grid).x, (void*)(stream)); fflush(stderr); } \
		hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, __VA_ARGS__); \
		if (lq_trace_launches) { hipError_t e_ = hipStreamSynchronize(stream); \
			if (e_ != hipSuccess) { fprintf(stderr, "[lqcov] FAILED (%s) at the end of %s stream %p\n", hipGetErrorString(e_), #kern, (void*)(stream)); fflush(stderr); } } } while (0)
#endif
#include <cstdint>
#include <cstddef>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#define LQ_U64MAX 0xffffffffffffffffULL

// One sketch thread owns the loop iterations that start inside one 128-base chunk of a read.
// Reads are laid out chunk-aligned in the packed arrays: 4 x u64 of 2-bit codes and 4 x u32 of
// "ambiguous" bits per chunk.
#define LQ_CHUNK       128
#define LQ_CHUNK_WORDS 4

#define LQ_COVT 150            // minimap2-coverage.h:20
#define LQ_SEED_TANDEM (1ULL << 42)   // mmpriv.h:18
#define LQ_TIE_MARK (1ULL << 63)      // y bit of anchors whose x may repeat inside their query (k_dup_mark); never read downstream
#define LQ_RS_MIN 64           // ksort.h:81

struct alignas(16) mm128 { u64 x, y; };        // minimap.h:42

// A sub-array the klib-order sort still has to partition by the byte at `shift` (ksort.h:99-129)
struct alignas(16) SortSeg { u64 off; u32 len; u32 shift; };

// interval tagged with its query: pos<<3 | flags (bit0 end, bit1 medium score, bit2 marker),
// minimap2-coverage.h:22-25, esterr.c:122-125, lqmap.c:69-71
struct Ivl { u32 q, start, end; };

struct ChainRec { i32 q, rid, rev, score, cnt, qs, qe, rs, re; };

// One kept chain of a (query, part) whose match counters are replayed on the host in the reference's order (sat_replay.hpp).
struct SatRec {
	u64 first_x, first_y;      // the chain's first anchor: chain.c:141 sorts the chains by its x, hit.c:62 hashes it
	u32 f_peak, run_hi, peak_j, seq;   // chain.c:99-107: the chains come out in descending (f[peak], peak index); run_hi: the run's strand | rid, seq: ordinal within the run
	u32 score, cnt;            // the chain's u[] entry (chain.c:117-121)
	i32 sti;                   // esterr.c:106
	u32 n_at;                  // counters it increments beyond sti (esterr.c:131-137), listed in the pool from at_off on
	u64 at_off;
	u32 good, span;            // good: reaches esterr.c:128; span: esterr.c:121's qe - qs + 1
};

struct MapParams {
	i32 k, w, hpc;
	i32 max_gap, bw, max_skip, min_cnt, min_sc;
	i32 min_sc_med, min_sc_good;
	i32 max_overhang, min_coverage;
	double min_ratio;
	i32 no_self, ava;
};
