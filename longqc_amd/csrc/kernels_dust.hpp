// longqc_amd/csrc/kernels_dust.hpp -- SURVEY 8(f)-4: the low-complexity scan of the reference's second binary,
// `sdust` (sdust.c:136-171 sdust_core, symmetric DUST), plus the two quality columns of its table (lqutils.c:51-69).
//
// The scan is a sequential state machine over a read: a window of the last <= W-2 triplet words with their counts, the
// "v" suffix of that window, the list P of perfect intervals of the window and the last masked interval, which may still
// grow.  An N ends the run of words and closes the pending intervals but leaves the window and its counts as they are
// (sdust.c:163-167), so a read cannot be cut into independent pieces: one thread walks one read.
//
// P is never materialised.  The reference keeps it sorted by descending start, then ascending finish (= insertion order),
// appends an interval (start, finish, r, l) after the last one with start >= its own, drops intervals by start, and reads
// it in exactly two ways: the maximum of r/l over the intervals with start >= some bound (sdust.c:119-123, compared by
// cross-multiplication, so only the value of the fraction matters) and the start and finish of the very last interval
// (sdust.c:97-104).  Live starts span fewer than 64 consecutive positions, so a ring of 64 buckets keyed by start with
// {largest r/l, finish of the newest interval} per bucket answers both; the list itself grows to ~1600 entries inside
// low-complexity sequence and scanning / shifting it would dominate (measured: 5 s for 5000 reads with the literal list
// in global scratch).  The window and the three count tables (64 one-byte entries each) live in LDS, lane-minor; the
// buckets (512 B per thread, touched only inside low-complexity sequence) in global scratch.
#pragma once
#include "lq_common.hpp"
#ifndef LQ_SHARED
#define LQ_SHARED __shared__
#endif

#define LQ_DUST_THREADS 64
#define LQ_DUST_PCAP 64         // buckets per thread
#define LQ_DUST_MAX_THREADS 65536  // threads per launch; reads are strided over them
struct DustPI { i32 finish; u32 rl; };                         // newest finish; largest r/l of the bucket as r << 8 | l

__global__ void __launch_bounds__(LQ_DUST_THREADS)
k_sdust(const u8 *seq, const u8 *qual, const u64 *seq_off, u32 n_reads, i32 W, i32 T, const double *q2p,
        DustPI *pi_scratch, u32 *masked_out, double *psum_out, u32 *qv_out)
{
	LQ_SHARED u8 s_q[64][LQ_DUST_THREADS], s_cw[64][LQ_DUST_THREADS], s_cv[64][LQ_DUST_THREADS], s_c[64][LQ_DUST_THREADS];
	const u32 tid = blockIdx.x * blockDim.x + threadIdx.x, n_threads = gridDim.x * blockDim.x;
	const u32 ln = threadIdx.x;
	DustPI *P = pi_scratch + (u64)tid * LQ_DUST_PCAP;
	for (u32 r = tid; r < n_reads; r += n_threads) {
	const u64 off = seq_off[r];
	const i32 len = (i32)(seq_off[r + 1] - off);
	const u8 *s = seq + off;
	for (int i = 0; i < 64; ++i) { s_cw[i][ln] = 0; s_cv[i][ln] = 0; }
	i32 qn = 0, qh = 0, rw = 0, rv = 0, L = 0;
	u64 occ = 0;                                               // non-empty buckets (bit = start & 63)
	i32 pmin = 0, pmax = 0;                                    // smallest / largest live start (occ != 0)
	i32 l = 0, have_last = 0, ls = 0, lf = 0;
	u32 t = 0;
	i64 masked = 0;
	// close the perfect intervals that start before `start` (sdust.c:93-108): the last of P -- the newest interval of the
	// smallest start -- extends or follows the last masked interval, then everything that starts before `start` goes
#define LQ_DUST_FLUSH(start_) do { \
		const i32 st_ = (start_); \
		if (occ != 0 && pmin < st_) { \
			const i32 p_start_ = pmin, p_fin_ = P[pmin & 63].finish; \
			if (have_last && p_start_ <= lf) { if (p_fin_ > lf) lf = p_fin_; } \
			else { if (have_last) masked += lf - ls; have_last = 1; ls = p_start_; lf = p_fin_; } \
			for (i32 s_ = pmin; s_ < st_ && s_ <= pmax; ++s_) occ &= ~(1ULL << (s_ & 63)); \
			if (occ != 0) { i32 s_ = st_; while (!(occ >> (s_ & 63) & 1)) ++s_; pmin = s_; } \
		} \
	} while (0)
	for (i32 i = 0; i <= len; ++i) {
		const u32 ch = i < len ? s[i] : 0u;
		const u32 cu = ch & 0xdfu;                             // upper case
		const i32 b = cu == 'A' ? 0 : cu == 'C' ? 1 : cu == 'G' ? 2 : cu == 'T' ? 3 : -1;   // seq_nt4_table of sdust.c:25-42
		if (b >= 0) {
			++l; t = (t << 2 | (u32)b) & 63u;
			if (l >= 3) {
				const i32 start = (l - W > 0 ? l - W : 0) + (i + 1 - l);
				LQ_DUST_FLUSH(start);
				// shift_window (sdust.c:70-91)
				if (qn >= W - 3 + 1) {
					const u32 so = s_q[qh][ln]; qh = (qh + 1) & 63; --qn;
					rw -= --s_cw[so][ln];
					if (L > qn) { --L; rv -= --s_cv[so][ln]; }
				}
				s_q[(qh + qn) & 63][ln] = (u8)t; ++qn;
				++L;
				rw += s_cw[t][ln]++;
				rv += s_cv[t][ln]++;
				if ((i32)s_cv[t][ln] * 10 > T << 1) {
					u32 so;
					do {
						so = s_q[(qh + qn - L) & 63][ln];
						rv -= --s_cv[so][ln];
						--L;
					} while (so != t);
				}
				if (rw * 10 > L * T) {
					// find_perfect (sdust.c:110-134)
					for (int z = 0; z < 64; ++z) s_c[z][ln] = s_cv[z][ln];
					i32 rr = rv, max_r = 0, max_l = 0;
					i32 folded = occ ? pmax + 1 : 0;               // buckets with start >= folded are already in (max_r, max_l)
					for (i32 wi = qn - L - 1; wi >= 0; --wi) {
						const u32 tw = s_q[(qh + wi) & 63][ln];
						rr += s_c[tw][ln]++;
						const i32 new_r = rr, new_l = qn - wi - 1;
						if (new_r * 10 > T * new_l) {
							const i32 thr = wi + start;
							if (occ) {
								for (i32 sv = (folded - 1 < pmax ? folded - 1 : pmax); sv >= thr && sv >= pmin; --sv)
									if (occ >> (sv & 63) & 1) {
										const u32 rl = P[sv & 63].rl;
										const i32 pr = (i32)(rl >> 8), pl = (i32)(rl & 0xff);
										if (max_r == 0 || pr * max_l > max_r * pl) { max_r = pr; max_l = pl; }
									}
								if (thr < folded) folded = thr;
							}
							if (max_r == 0 || new_r * max_l >= max_r * new_l) {
								max_r = new_r; max_l = new_l;
								const u32 idx = (u32)thr & 63u;
								DustPI e; e.finish = qn + 2 + start; e.rl = (u32)new_r << 8 | (u32)new_l;
								if (occ >> idx & 1) {                   // a newer interval of the same start: the bucket keeps its largest r/l
									const u32 rl = P[idx].rl;
									if ((i32)(rl >> 8) * new_l > new_r * (i32)(rl & 0xff)) e.rl = rl;
								} else {
									if (occ == 0) { pmin = thr; pmax = thr; folded = thr; }
									else { if (thr < pmin) pmin = thr; if (thr > pmax) pmax = thr; }
									occ |= 1ULL << idx;
								}
								P[idx] = e;
							}
						}
					}
				}
			}
		} else {
			i32 start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);
			while (occ) { LQ_DUST_FLUSH(start); ++start; }
			l = 0; t = 0;
		}
	}
#undef LQ_DUST_FLUSH
	if (have_last) masked += lf - ls;
	masked_out[r] = (u32)masked;
	// meanQ's sum (lqutils.c:51-56: sequential, in read order) and getQV(qual, 7) (lqutils.c:61-69); a record without
	// qualities arrives as zero bytes
	double ps = 0.0;
	u32 qv = 0;
	if (qual && len > 0 && qual[off] != 0) {
		const u8 *q = qual + off;
		for (i32 i = 0; i < len; ++i) {
			const i32 v = (i32)(signed char)q[i];                       // (char arithmetic, as lqutils.c:51-56 and k_qual_sum)
			const i32 t = v - 33;
			ps += q2p[t < 0 ? 0 : t > 126 ? 126 : t];                  // the reference indexes out of bounds outside Q0..Q126: clamped, like k_qual_sum
			if (v > 7 + 33) ++qv;
		}
	}
	psum_out[r] = ps; qv_out[r] = qv;
	}
}
