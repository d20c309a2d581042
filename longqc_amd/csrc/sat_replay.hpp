// Host side of the one order-dependent corner of the path: the reference's match counters are uint16, and the two tests that
// should saturate them both read a[st] -- the counter of the chain's FIRST minimizer -- instead of the counter they are about to
// increment (esterr.c:130,136).  So a[st] stops at 65535, a[j] (j beyond st) is incremented while a[st] is below 65535 and wraps
// at 65536, and the final counters of a query depend on the order in which lq_cnt_match walked its chains.  That order is the
// order of regs[] (esterr.c:99), i.e. of mm_gen_regs (hit.c:52-88):
//   1. mm_chain_dp emits the kept chains in descending (f[peak], peak index) (chain.c:99-125),
//   2. re-sorts them by the x of their first anchor with radix_sort_128x (chain.c:139-146; klib's unstable sort: chains that
//      start at equal x come out in an order that depends on the whole array),
//   3. mm_gen_regs sorts by (score << 32 | cnt) ^ h, h a 32-bit hash of the first anchor and of the query's name and length
//      (hit.c:60-66, lqmap.c:232-234), radix_sort_128x again, and reverses.
// The device counts in 32 bits (the low 16 bits are the reference's value as long as no counter of the query reaches 65535) and
// flags the queries where one does; for those -- piles of 65535 and more good overlaps on one minimizer inside one index part --
// the engine records every kept chain of the (query, part) (SatRec, kernels_chain.hpp) and this file replays them: the three
// orders above, then esterr.c:127-138 on 16-bit counters.  Nothing here runs for a query that never saturates.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>
#include "lq_common.hpp"

namespace satreplay {

struct Rec128 { u64 x, y; };

// klib's radix sort of 16-byte records by x (ksort.h KRADIX_SORT_INIT as instantiated by misc.c:147-150: bytes from the most
// significant one down, 256 buckets permuted in place by cycle leading, buckets of <= 64 records by insertion sort, larger ones
// by the next byte).  Only the resulting ORDER matters here, ties included, so the in-place permutation is followed literally:
// the cursor of the current bucket hands its record to the bucket it belongs to, takes that bucket's record in exchange, and so
// on until a record of the current bucket comes back.
static inline void klib_insertion(Rec128 *a, size_t n)
{
	for (size_t i = 1; i < n; ++i) {
		if (a[i].x < a[i - 1].x) {
			const Rec128 t = a[i];
			size_t j = i;
			while (j > 0 && t.x < a[j - 1].x) { a[j] = a[j - 1]; --j; }
			a[j] = t;
		}
	}
}

static inline void klib_pass(Rec128 *a, size_t n, int shift)
{
	size_t head[256], tail[256], cnt[256] = {0};
	for (size_t i = 0; i < n; ++i) ++cnt[a[i].x >> shift & 0xff];
	size_t at = 0;
	for (int c = 0; c < 256; ++c) { head[c] = at; at += cnt[c]; tail[c] = at; }
	for (int c = 0; c < 256; ) {
		if (head[c] == tail[c]) { ++c; continue; }
		int d = (int)(a[head[c]].x >> shift & 0xff);
		if (d == c) { ++head[c]; continue; }
		Rec128 carry = a[head[c]];
		do {
			const Rec128 out = a[head[d]];
			a[head[d]++] = carry;
			carry = out;
			d = (int)(carry.x >> shift & 0xff);
		} while (d != c);
		a[head[c]++] = carry;
	}
	if (shift == 0) return;
	at = 0;
	for (int c = 0; c < 256; ++c) {
		if (cnt[c] > 64) klib_pass(a + at, cnt[c], shift - 8);
		else if (cnt[c] > 1) klib_insertion(a + at, cnt[c]);
		at += cnt[c];
	}
}

static inline void klib_sort_128x(std::vector<Rec128> &a)
{
	if (a.size() <= 64) klib_insertion(a.data(), a.size());
	else klib_pass(a.data(), a.size(), 56);
}

static inline u64 mix64(u64 key)                               // hit.c:40-50 (Thomas Wang's 64-bit mix)
{
	key = ~key + (key << 21);
	key ^= key >> 24;
	key = key + (key << 3) + (key << 8);
	key ^= key >> 14;
	key = key + (key << 2) + (key << 4);
	key ^= key >> 28;
	key += key << 31;
	return key;
}
static inline u32 mix32(u32 key)                               // khash.h's __ac_Wang_hash
{
	key += ~(key << 15); key ^= key >> 10; key += key << 3;
	key ^= key >> 6; key += ~(key << 11); key ^= key >> 16;
	return key;
}
static inline u32 query_hash(const std::string &name, i32 qlen, i32 seed)   // lqmap.c:232-234
{
	u32 h = 0;
	if (!name.empty()) { h = (u32)(unsigned char)name[0]; for (size_t i = 1; i < name.size(); ++i) h = (h << 5) - h + (u32)(unsigned char)name[i]; }
	h ^= mix32((u32)qlen) + mix32((u32)seed);
	return mix32(h);
}

// the order in which lq_cnt_match meets the chains of one (query, part): indices into recs
static inline std::vector<u32> regs_order(const std::vector<SatRec> &recs, u32 qhash)
{
	const size_t n = recs.size();
	std::vector<u32> gen(n);
	for (size_t i = 0; i < n; ++i) gen[i] = (u32)i;
	// chain.c:99-107: u[] = f[peak] << 32 | peak index, ascending, walked from the top.  The peak index is a position in the whole
	// sorted anchor array: runs in the order of their strand | rid, positions inside a run as recorded (in a run whose equal-x
	// anchors could be told apart the recorded positions are klib's).  Two ends that share a peak are equal entries; the device
	// met them in the same order (seq).
	std::sort(gen.begin(), gen.end(), [&](u32 l, u32 r) {
		const SatRec &a = recs[l], &b = recs[r];
		if (a.f_peak != b.f_peak) return a.f_peak > b.f_peak;
		if (a.run_hi != b.run_hi) return a.run_hi > b.run_hi;
		if (a.peak_j != b.peak_j) return a.peak_j > b.peak_j;
		return a.seq < b.seq;
	});
	// chain.c:139-146: by the first anchor's x
	std::vector<Rec128> w(n);
	for (size_t i = 0; i < n; ++i) { w[i].x = recs[gen[i]].first_x; w[i].y = gen[i]; }
	klib_sort_128x(w);
	// hit.c:60-70: by (score << 32 | cnt) ^ h, then reversed
	std::vector<Rec128> z(n);
	for (size_t i = 0; i < n; ++i) {
		const SatRec &c = recs[(size_t)w[i].y];
		const u32 h = (u32)mix64((mix64(c.first_x) + mix64(c.first_y)) ^ qhash);
		z[i].x = ((u64)c.score << 32 | c.cnt) ^ h;
		z[i].y = w[i].y;
	}
	klib_sort_128x(z);
	std::vector<u32> order(n);
	for (size_t i = 0; i < n; ++i) order[i] = (u32)z[n - 1 - i].y;
	return order;
}

// esterr.c:127-138 over the good chains in that order, on counters of `cnt_max`'s width
static inline void replay(std::vector<u32> &cnt, const std::vector<SatRec> &recs, const std::vector<u32> &at, const std::vector<u32> &order, u32 cnt_max)
{
	for (u32 ri : order) {
		const SatRec &c = recs[ri];
		if (!c.good) continue;
		u32 &first = cnt[(size_t)c.sti];
		if (first < cnt_max) ++first;
		for (u32 k = 0; k < c.n_at; ++k)
			if (first < cnt_max) { u32 &o = cnt[at[c.at_off + k]]; o = (o + 1) & cnt_max; }
	}
}

} // namespace satreplay
