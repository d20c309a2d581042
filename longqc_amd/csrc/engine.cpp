// longqc_amd/csrc/engine.cpp -- see engine.hpp.  Compiled by hipcc for gfx950 (liblqcov.so).
#include "engine.hpp"
#include "kernels_sketch.hpp"
#include "kernels_index.hpp"
#include "kernels_seed.hpp"
#include "kernels_sort.hpp"
#include "kernels_walk.hpp"
#include "kernels_psort.hpp"
#include "kernels_chain.hpp"
#include "fastx.hpp"
#include "fastx_mem.hpp"
#include "sat_replay.hpp"
#ifndef LQ_EMU
extern "C" void lq_segv_altstack();                      // api.cpp (LQCOV_SEGV_TRACE)
#endif
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cinttypes>
#include <algorithm>
#include <unordered_map>
#include <deque>
#include <future>
#include <functional>

static inline u32 nblk(u64 n, u32 bs)
{
	u64 b = (n + bs - 1) / bs;
	if (b > 0x7fffffffULL) throw std::runtime_error("launch too large");
	return (u32)b;
}

template <class T> static void h2d(T *dst, const T *src, size_t n, hipStream_t s)
{
	if (n) LQ_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, s));
}
template <class T> static void d2h(T *dst, const T *src, size_t n, hipStream_t s)
{
	if (n) { LQ_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, s)); LQ_HIP_CHECK(hipStreamSynchronize(s)); }
}
static void dzero(void *p, size_t bytes, hipStream_t s) { if (bytes) LQ_HIP_CHECK(hipMemsetAsync(p, 0, bytes, s)); }
static void check_launch() { LQ_HIP_CHECK(hipGetLastError()); }
#include <chrono>
static double lq_now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// LQCOV_TIMELINE=1: host times (ms since the handle's last reset) at the points where a thread has just waited for its stream --
// what a step's critical path is made of, without a profiler in the way (rocprofv3 serialises dispatches and stretches the picture)
static const bool lq_timeline = getenv("LQCOV_TIMELINE") != nullptr;
static double lq_timeline_t0 = 0;
static void lq_tl(const char *who, int id, const char *what, double extra = -1)
{
	if (!lq_timeline) return;
	if (extra >= 0) fprintf(stderr, "[tl] %8.2f %s%d %s %.0f\n", (lq_now_s() - lq_timeline_t0) * 1e3, who, id, what, extra);
	else fprintf(stderr, "[tl] %8.2f %s%d %s\n", (lq_now_s() - lq_timeline_t0) * 1e3, who, id, what);
}
#ifndef LQ_EMU
int lq_trace_launches = 0;
#endif

// ---- stage timing ---------------------------------------------------------------------------
StageTimer::StageTimer(lqcov_handle *h_, hipStream_t s_, const char *name_, u64 bytes_) : h(h_), s(s_), name(name_), bytes(bytes_)
{
	// (profile_only: one stage name, or several separated by '|' -- bench.py times the group of kernels that bounds the step)
	on = h->profiling && (h->profile_only.empty() || h->profile_only == name_ || ("|" + h->profile_only + "|").find(std::string("|") + name_ + "|") != std::string::npos);
	if (!on) return;
	hipEventCreate(&a); hipEventCreate(&b);
	hipEventRecord(a, s);
}
StageTimer::StageTimer(lqcov_handle *h_, const char *name_, u64 bytes_) : StageTimer(h_, h_->stream, name_, bytes_) {}
StageTimer::~StageTimer()
{
	if (!on) return;
	hipEventRecord(b, s);
	if (h->profiling == 1) { hipEventSynchronize(b); std::lock_guard<std::mutex> g(h->stage_mu); h->account_stage(name, a, b, bytes); }
	else { std::lock_guard<std::mutex> g(h->stage_mu); h->stage_pending.push_back(lqcov_handle::StagePending{name, a, b, bytes}); }   // read later: nothing waits here
}
void lqcov_handle::account_stage(const char *name, hipEvent_t a, hipEvent_t b, u64 bytes)
{
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	hipEventDestroy(a); hipEventDestroy(b);
	auto it = stages.find(name);
	if (it == stages.end()) { stage_order.push_back(name); it = stages.emplace(name, StageAcc()).first; }
	it->second.ms += ms; it->second.launches += 1; it->second.bytes += bytes;
}
void lqcov_handle::drain_stages()
{
	std::lock_guard<std::mutex> g(stage_mu);
	for (lqcov_handle::StagePending &sp : stage_pending) { hipEventSynchronize(sp.b); account_stage(sp.name, sp.a, sp.b, sp.bytes); }
	stage_pending.clear();
	for (auto &kv : late_bytes) { auto it = stages.find(kv.first); if (it != stages.end()) it->second.bytes += kv.second; }
	late_bytes.clear();
}
void lqcov_handle::add_stage_bytes(const char *name, u64 bytes)
{
	if (!profiling || !bytes) return;
	std::lock_guard<std::mutex> g(stage_mu);
	late_bytes[name] += bytes;
}

// ---- handle ---------------------------------------------------------------------------------
void Knobs::read_env()
{
	auto num = [](const char *name, long dflt) { const char *e = getenv(name); return e && *e ? atol(e) : dflt; };
	auto is = [](const char *name, const char *val) { const char *e = getenv(name); return e && !strcmp(e, val); };
	lanes = (int)std::min<long>(8, std::max<long>(1, num("LQCOV_LANES", 3)));
	anchor_budget = getenv("LQCOV_ANCHOR_BUDGET") ? strtoull(getenv("LQCOV_ANCHOR_BUDGET"), 0, 10) : 0;
	query_order_file = is("LQCOV_QUERY_ORDER", "file");
	query_order_striped = is("LQCOV_QUERY_ORDER", "striped");
	all_klib = is("LQCOV_SORT", "klib");
	ps_shift = (u32)std::min<long>(12, std::max<long>(0, num("LQCOV_PS_SHIFT", 0)));
	reg_walker = !is("LQCOV_WALK", "solo");
	ckpt = !is("LQCOV_CKPT", "0");
	ckpt3 = num("LQCOV_CKPT3", 1) > 0;
	prune = num("LQCOV_PRUNE", 1) > 0;
	plan_lazy = num("LQCOV_PLAN_LAZY", 0) > 0;
	lazy_batches = (int)std::min<long>(16, std::max<long>(1, num("LQCOV_LAZY_BATCHES", 1)));
	sketch_grid = (u32)std::min<long long>(std::max<long long>(num("LQCOV_SKETCH_GRID", 1L << 22), 1), 1L << 22);
	build_prio = num("LQCOV_BUILD_PRIO", 1) > 0;
	ck_unit = (u32)std::max<long long>(num("LQCOV_CK_UNIT", 65536), 64); ck_unit_many = (u32)std::max<long long>(num("LQCOV_CK_UNIT_MANY", 8192), 64);
	sort_tile = (u32)std::max<long>(0, num("LQCOV_SORT_TILE", 0)); if (sort_tile && sort_tile < 64) sort_tile = 64;
	walk_shift = (u32)std::min<long>(16, std::max<long>(0, num("LQCOV_WALK_SHIFT", 0)));
	walk_grid = (u32)std::max<long>(64, num("LQCOV_WALK_GRID", 1L << 18));
	chain_wave_min = (int)std::max<long>(0, num("LQCOV_CHAIN_WAVE_MIN", 0));
	chain_cap = (int)num("LQCOV_CHAIN_CAP", 128);
	run_stage = (u32)std::max<long>(1, num("LQCOV_RUN_STAGE", LQ_RUN_STAGE));
	run_grid = (u32)std::max<long>(1, num("LQCOV_RUN_GRID", 2048));
	ps_key64 = num("LQCOV_PS_KEY64", 0) != 0;
	sketch_wgen = num("LQCOV_SKETCH_WGEN", 0) != 0;
	upload_amb = num("LQCOV_UPLOAD_AMB", 0) != 0;
	sketch_key = num("LQCOV_SKETCH_KEY", 1) != 0;
	emit_grid = (u32)std::min<long>(1L << 22, std::max<long>(1, num("LQCOV_EMIT_GRID", 1L << 22)));
	sketch_list = num("LQCOV_SKETCH_LIST", 1) != 0;
	sketch_fast = num("LQCOV_SKETCH_FAST", 1) != 0;
#ifndef LQ_EMU
	lq_trace_launches = (int)num("LQCOV_TRACE_LAUNCHES", 0);
#endif
	no_level_skip = getenv("LQCOV_NO_LEVEL_SKIP") != nullptr;
	debug_sort = getenv("LQCOV_DEBUG_SORT") != nullptr;
	sketch_kpt = (u32)std::min<long>(64, std::max<long>(1, num("LQCOV_SKETCH_KPT", 4)));
	sketch_machine_only = is("LQCOV_SKETCH", "machine");
	upload_slices = (u32)std::min<long>(8, std::max<long>(1, num("LQCOV_UPLOAD_SLICES", 4)));
	upload_min_chunks = (u64)std::max<long>(1, num("LQCOV_UPLOAD_MIN_CHUNKS", 1L << 16));
	ps_grid = (u32)std::max<long>(64, num("LQCOV_PS_GRID", 512));
	tile_grid = (u32)std::max<long>(64, num("LQCOV_TILE_GRID", 4096));
	ps_passes = (u32)std::min<long>(16, std::max<long>(0, num("LQCOV_PS_PASSES", 2))) & ~1u;
	ties_klib = is("LQCOV_TIES", "klib") || all_klib;
	walk_cu_mask = getenv("LQCOV_WALK_CU_MASK") ? (u32)strtoul(getenv("LQCOV_WALK_CU_MASK"), 0, 16) : 0x11111111u; if (!walk_cu_mask) walk_cu_mask = 0xffffffffu;
	filter = num("LQCOV_FILTER", 1) != 0;
	parse_threads = (int)std::min<long>(256, std::max<long>(0, num("LQCOV_PARSE_THREADS", 0)));
	parse_piece = (u64)std::max<long>(64, num("LQCOV_PARSE_PIECE", 32L << 20));
	parse_side = getenv("LQCOV_PARSE_SIDE") ? strtoull(getenv("LQCOV_PARSE_SIDE"), 0, 10) : 2ULL << 30;
	pipeline = num("LQCOV_PIPELINE", 1) != 0;
	plan_ahead = num("LQCOV_PLAN_AHEAD", 1) != 0;
	cnt_bits = (int)std::min<long>(16, std::max<long>(2, num("LQCOV_TEST_CNT_BITS", 16)));
	seed_bucket = (u32)std::min<long>(1L << 24, std::max<long>(16, num("LQCOV_SEED_BUCKET", 7600)));
	seed_chunk = getenv("LQCOV_SEED_CHUNK") ? std::max<u64>(1024, strtoull(getenv("LQCOV_SEED_CHUNK"), 0, 10)) : 1ULL << 30;
	if (seed_chunk > 0xfffffff0ULL) seed_chunk = 0xfffffff0ULL;
	seed_segl = (u32)std::min<long>(LQ_SD_SEGL, std::max<long>(1, num("LQCOV_SEED_SEGL", LQ_SD_SEGL)));
	seed_pair_bits = (u32)std::min<long>(LQ_SD_PAIR_BITS, std::max<long>(1, num("LQCOV_SEED_PAIR_BITS", LQ_SD_PAIR_BITS)));
	seed_hwords = (u32)std::min<long>(LQ_SD_HWORDS, std::max<long>(2, num("LQCOV_SEED_HWORDS", LQ_SD_HWORDS)));
	seed_surv_max = getenv("LQCOV_SEED_SURV_MAX") ? std::max<u64>(1, strtoull(getenv("LQCOV_SEED_SURV_MAX"), 0, 10)) : 2ULL << 30;
	seed_dcap = (u32)std::min<long>(LQ_SD_DCAP, std::max<long>(1, num("LQCOV_SEED_DCAP", LQ_SD_DCAP)));
	seed_bigcap = (u32)std::min<long>(LQ_SD_BIGCAP, std::max<long>(1, num("LQCOV_SEED_BIGCAP", LQ_SD_BIGCAP)));
}

lqcov_handle::lqcov_handle(const lqcov_params &p, int dev) : P(p), device(dev)
{
	if (P.k < 1 || P.k > 28 || P.w < 1 || P.w > 255) throw std::invalid_argument("k must be in [1,28] and w in [1,255] (sketch.c:83)");
	if (P.min_score_med >= 65536 || P.min_score_good >= 65536 || P.min_score_med < 0 || P.min_score_good < 0)
		throw std::invalid_argument("-p and -q must be below 65536 (lqmap.c:841 packs them into 16 bits each)");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available");
	if (dev < 0 || dev >= ndev) throw std::runtime_error("HIP device index out of range");
	LQ_HIP_CHECK(hipSetDevice(dev));
	LQ_HIP_CHECK(hipStreamCreate(&stream));
	// (rounds 2-3, steps of 1.5 s: the build side at the lanes' own priority stretched -- the sketch of part 2 560 ms under five lanes,
	// 21 ms alone -- but the part was ready in time, and a high-priority build stream was measured slower: 1544-1567 vs 1508-1519 ms)
	{	// The build side -- upload, sketch, index and seed plan of the NEXT part -- runs beside the mapping lanes, whose second passes
		// are thousands of single-lane waves that live for milliseconds and hold every wave slot they are given.  Round 5's trace:
		// the sketch of part 2 stretched from 33 to 164 ms under them and its plan ended after the lanes were done with part 1 -- the
		// build side had become the critical path.  Its streams get the queue priority that lets their blocks take the slots that
		// free up first (LQCOV_BUILD_PRIO=0: plain streams).
		int lo = 0, hi = 0;
		if (K.build_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) {
			LQ_HIP_CHECK(hipStreamCreateWithPriority(&bstream, hipStreamDefault, hi));
			LQ_HIP_CHECK(hipStreamCreateWithPriority(&cstream, hipStreamDefault, hi));
		} else {
			(void)hipGetLastError();
			LQ_HIP_CHECK(hipStreamCreate(&bstream));
			LQ_HIP_CHECK(hipStreamCreate(&cstream));
		}
	}
	for (hipEvent_t &e : ev_up) LQ_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	lq_pool_keep_memory(dev);
	prim.stream = stream; bprim.stream = bstream;
	mp.k = P.k; mp.w = P.w; mp.hpc = P.hpc;
	mp.max_gap = P.max_gap; mp.bw = P.bw; mp.max_skip = P.max_chain_skip; mp.min_cnt = P.min_cnt; mp.min_sc = P.min_chain_score;
	mp.min_sc_med = P.min_score_med; mp.min_sc_good = P.min_score_good;
	mp.max_overhang = P.max_overhang; mp.min_coverage = P.min_coverage; mp.min_ratio = P.min_ratio;
	mp.no_self = P.no_self; mp.ava = P.ava;
	K.read_env();
	cnt_max = (1u << K.cnt_bits) - 1;
	anchor_budget = K.anchor_budget;                        // 0: from the free HBM when the first part is mapped (map_part)
	n_lanes = K.lanes;
}

lqcov_handle::~lqcov_handle()
{
	drain_stages();
	for (auto &L : lanes) {
		hipStream_t s1 = L->stream, s2 = L->stream2, s3 = L->streamW, s4 = L->streamW2;
		hipEvent_t e1 = L->ev_fork, e2 = L->ev_join, e3 = L->ev_w0, e4 = L->ev_w1;
		if (s1) hipStreamSynchronize(s1);
		if (s2) hipStreamSynchronize(s2);
		if (s3) { hipStreamSynchronize(s3); hipStreamDestroy(s3); }
		if (s4) { hipStreamSynchronize(s4); hipStreamDestroy(s4); }
		if (L->ev_w2) hipEventDestroy(L->ev_w2);
		if (e3) hipEventDestroy(e3);
		if (e4) hipEventDestroy(e4);
		L.reset();                                              // the lane's buffers go back to the pool while its stream still exists
		if (s1) { hipStreamSynchronize(s1); hipStreamDestroy(s1); }
		if (s2) hipStreamDestroy(s2);
		if (e1) hipEventDestroy(e1);
		if (e2) hipEventDestroy(e2);
	}
	if (cstream) { hipStreamSynchronize(cstream); hipStreamDestroy(cstream); }
	for (hipEvent_t e : ev_up) if (e) hipEventDestroy(e);
	if (bstream) { hipStreamSynchronize(bstream); hipStreamDestroy(bstream); }
	if (stream) { hipStreamSynchronize(stream); hipStreamDestroy(stream); }
}

Part &lqcov_handle::part(int id)
{
	if (id < 0 || (size_t)id >= parts.size() || !parts[id] || !parts[id]->live) throw std::logic_error("no such index part");
	return *parts[id];
}

// ---- read sets ------------------------------------------------------------------------------
static void grow_keep(DBuf &b, size_t old_bytes, size_t new_bytes, hipStream_t s)
{
	if (new_bytes <= b.cap) return;
	void *np = nullptr;
	size_t want = new_bytes + new_bytes / 2 + 4096;
	LQ_HIP_CHECK(hipMalloc(&np, want));
	if (old_bytes) LQ_HIP_CHECK(hipMemcpyAsync(np, b.p, old_bytes, hipMemcpyDeviceToDevice, s));
	LQ_HIP_CHECK(hipStreamSynchronize(s));
	if (b.p) { void *old = b.p; b.p = nullptr; DBuf tmp; tmp.p = old; tmp.cap = 1; tmp.pool_stream = b.pool_stream; }   // (freed the way it was allocated)
	b.p = np; b.cap = want; b.pool_stream = nullptr;
}

// upload n reads (ASCII) and append them, 2-bit packed, to the set   (index.c:240-288 step 0)
void lqcov_handle::add_reads(ReadSetDev &rs, u32 n, const u8 *seq, const u64 *seq_off, const char *names, const u64 *name_off)
{
	hipStream_t stream = this->bstream; Prim &prim = this->bprim;   // the build side has a stream and scan / sort scratch of its own: a part can be built while another is mapped
	(void)prim;
	if (n == 0) return;
	if ((u64)rs.n + n > 0x7fffffffULL) throw std::domain_error("too many reads in one set");
	std::vector<u64> coff_local(n + 1, 0);
	for (u32 i = 0; i < n; ++i) {
		u64 len = seq_off[i + 1] - seq_off[i];
		if (len > 0x7fffffffULL) throw std::domain_error("read longer than 2^31-1 bases (bseq.c:80)");
		coff_local[i + 1] = coff_local[i] + (len + LQ_CHUNK - 1) / LQ_CHUNK;
		rs.h_len.push_back((u32)len);
		rs.h_coff.push_back(rs.n_chunks + coff_local[i + 1]);
		rs.names.emplace_back(names ? names + name_off[i] : "");
	}
	const u64 n_bases = seq_off[n] - seq_off[0], new_chunks = coff_local[n], n_words = new_chunks * LQ_CHUNK_WORDS;
	DBuf d_ascii, d_soff, d_coff;
	d_ascii.ensure(n_bases + 16); d_soff.ensure((n + 1) * 8); d_coff.ensure((n + 1) * 8);
	std::vector<u64> soff(n + 1);
	for (u32 i = 0; i <= n; ++i) soff[i] = seq_off[i] - seq_off[0];
	h2d(d_ascii.as<u8>(), seq + seq_off[0], n_bases, stream);
	h2d(d_soff.as<u64>(), soff.data(), n + 1, stream);
	h2d(d_coff.as<u64>(), coff_local.data(), n + 1, stream);
	grow_keep(rs.codes, rs.n_chunks * LQ_CHUNK_WORDS * 8, (rs.n_chunks + new_chunks) * LQ_CHUNK_WORDS * 8, stream);
	grow_keep(rs.amb, rs.n_chunks * LQ_CHUNK_WORDS * 4, (rs.n_chunks + new_chunks) * LQ_CHUNK_WORDS * 4, stream);
	if (n_words) {
		StageTimer t(this, stream, "k_pack", n_bases + n_words * 12);
		LQ_LAUNCH(k_pack, nblk(n_words, 256), 256, stream, d_ascii.as<u8>(), d_soff.as<u64>(), d_coff.as<u64>(), n, n_words,
		          rs.codes.as<u64>() + rs.n_chunks * LQ_CHUNK_WORDS, rs.amb.as<u32>() + rs.n_chunks * LQ_CHUNK_WORDS);
		check_launch();
	}
	LQ_HIP_CHECK(hipStreamSynchronize(stream));              // staging buffers die here
	rs.n += n; rs.n_chunks += new_chunks; rs.n_bases += n_bases;
	rs.sketched = false; rs.dp_n = 0; rs.dp_tiles = 0;
}

// ---- packed reads from the host -----------------------------------------------------------------
// The same layout k_pack produces (2-bit codes in u64 words + one "ambiguous" bit per base in u32 words, every read
// starting on a 128-base chunk), built on the host by the thread that parses the reads, so that 0.375 B per base cross
// PCIe instead of the ASCII byte (seq_nt4_table, sketch.c:8-25).
u64 lq_packed_chunks(u32 n, const u64 *seq_off)
{
	u64 c = 0;
	for (u32 i = 0; i < n; ++i) c += (seq_off[i + 1] - seq_off[i] + LQ_CHUNK - 1) / LQ_CHUNK;
	return c;
}

// does a packed read hold an ambiguous base (a bit of `aw` set at a position below len)?
bool lq_packed_read_ambiguous(const u32 *aw, u64 len)
{
	const u64 full = len >> 5;
	for (u64 wi = 0; wi < full; ++wi) if (aw[wi]) return true;
	return (len & 31) && (aw[full] & ((1u << (len & 31)) - 1u));
}

void lq_pack_host(u32 n, const u8 *seq, const u64 *seq_off, u64 *codes, u32 *amb, int n_threads)
{
	static u8 tab[256];
	static std::once_flag once;
	std::call_once(once, [] {
		for (int c = 0; c < 256; ++c) tab[c] = 4;
		tab[0] = 0; tab[1] = 1; tab[2] = 2; tab[3] = 3;
		tab['A'] = tab['a'] = 0; tab['C'] = tab['c'] = 1; tab['G'] = tab['g'] = 2; tab['T'] = tab['t'] = tab['U'] = tab['u'] = 3;
	});
	std::vector<u64> coff(n + 1, 0);
	for (u32 i = 0; i < n; ++i) coff[i + 1] = coff[i] + (seq_off[i + 1] - seq_off[i] + LQ_CHUNK - 1) / LQ_CHUNK;
	auto work = [&](u32 r0, u32 r1) {
		for (u32 r = r0; r < r1; ++r) {
			const u8 *s = seq + seq_off[r];
			const u64 len = seq_off[r + 1] - seq_off[r];
			u64 *cw = codes + coff[r] * LQ_CHUNK_WORDS;
			u32 *aw = amb + coff[r] * LQ_CHUNK_WORDS;
			const u64 nw = (coff[r + 1] - coff[r]) * LQ_CHUNK_WORDS;
			for (u64 wi = 0; wi < nw; ++wi) {
				const u64 p0 = wi * 32;
				u64 w = 0; u32 m = 0;
				const u64 lim = p0 >= len ? 0 : (len - p0 < 32 ? len - p0 : 32);
				for (u64 j = 0; j < lim; ++j) {
					const u32 c = tab[s[p0 + j]];
					if (c < 4) w |= (u64)c << (2 * j); else m |= 1u << j;
				}
				if (lim < 32) m |= lim == 0 ? 0xffffffffu : ~0u << lim;      // beyond the read: ambiguous, as k_pack marks it
				cw[wi] = w; aw[wi] = m;
			}
		}
	};
	if (n_threads <= 0) n_threads = (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));   // (a memory-bound loop into page-locked buffers: 16 threads pack 5.2 Gbases in 0.37 s on the 256-thread host, 64 threads in 0.53 s -- round 6)
	const u64 total = seq_off[n] - seq_off[0];
	if (n_threads == 1 || total < (1u << 22)) { work(0, n); return; }
	std::vector<std::thread> th;
	u32 r0 = 0;
	for (int t = 0; t < n_threads; ++t) {                          // equal shares of bases
		const u64 want = seq_off[0] + total * (u64)(t + 1) / (u64)n_threads;
		u32 r1 = t + 1 == n_threads ? n : (u32)(std::upper_bound(seq_off + r0, seq_off + n + 1, want) - seq_off);
		if (r1 > n) r1 = n;
		if (r1 < r0) r1 = r0;
		if (r1 > r0) th.emplace_back(work, r0, r1);
		r0 = r1;
	}
	for (auto &t : th) t.join();
}

// append n reads that are already packed (lq_pack_host layout for exactly these reads) to the set
void lqcov_handle::add_reads_packed(ReadSetDev &rs, u32 n, const u64 *codes, const u32 *amb, const u32 *lens, const char *names, const u64 *name_off,
                                    const u64 *codes_dev, const u32 *amb_dev, u64 stride_chunks, const std::vector<u64> *share_chunks)
{
	hipStream_t stream = this->bstream; Prim &prim = this->bprim;   // the build side has a stream and scan / sort scratch of its own: a part can be built while another is mapped
	(void)prim;
	if (n == 0) return;
	if ((u64)rs.n + n > 0x7fffffffULL) throw std::domain_error("too many reads in one set");
	u64 new_chunks = 0, n_bases = 0;
	rs.h_len.reserve(rs.h_len.size() + n); rs.h_coff.reserve(rs.h_coff.size() + n); rs.names.reserve(rs.names.size() + n);
	for (u32 i = 0; i < n; ++i) {
		if (lens[i] > 0x7fffffffu) throw std::domain_error("read longer than 2^31-1 bases (bseq.c:80)");
		new_chunks += ((u64)lens[i] + LQ_CHUNK - 1) / LQ_CHUNK;
		n_bases += lens[i];
		rs.h_len.push_back(lens[i]);
		rs.h_coff.push_back(rs.n_chunks + new_chunks);
		rs.names.emplace_back(names ? names + name_off[i] : "");
	}
	const u64 n_words = new_chunks * LQ_CHUNK_WORDS;
	grow_keep(rs.codes, rs.n_chunks * LQ_CHUNK_WORDS * 8, (rs.n_chunks + new_chunks) * LQ_CHUNK_WORDS * 8, stream);
	grow_keep(rs.amb, rs.n_chunks * LQ_CHUNK_WORDS * 4, (rs.n_chunks + new_chunks) * LQ_CHUNK_WORDS * 4, stream);
	const bool first = rs.n == 0;
	const u64 chunk0 = rs.n_chunks;
	rs.n += n; rs.n_chunks += new_chunks; rs.n_bases += n_bases;
	rs.sketched = false; rs.dp_n = 0; rs.dp_tiles = 0;
	if (share_chunks) {
		// the packed reads are on the device already, in shares of a common stride (lqcov_part_add_packed_shares_dev): back to back into the part
		u64 at = chunk0, tot = 0;
		for (u64 v : *share_chunks) tot += v;
		if (tot != new_chunks) throw std::invalid_argument("the shares' chunks do not add up to the reads' chunks");
		StageTimer t(this, stream, "d2d_packed_shares", n_words * 12);
		for (size_t i = 0; i < share_chunks->size(); ++i) {
			const u64 c = (*share_chunks)[i];
			if (!c) continue;
			LQ_HIP_CHECK(hipMemcpyAsync(rs.codes.as<u64>() + at * LQ_CHUNK_WORDS, codes_dev + (u64)i * stride_chunks * LQ_CHUNK_WORDS, c * LQ_CHUNK_WORDS * 8, hipMemcpyDeviceToDevice, stream));
			if (amb_dev) LQ_HIP_CHECK(hipMemcpyAsync(rs.amb.as<u32>() + at * LQ_CHUNK_WORDS, amb_dev + (u64)i * stride_chunks * LQ_CHUNK_WORDS, c * LQ_CHUNK_WORDS * 4, hipMemcpyDeviceToDevice, stream));
			at += c;
		}
		if (!amb_dev) {                                            // (no read of the part holds an ambiguous base: the ranks exchanged the codes alone)
			rs.d_coff.ensure((rs.n + 1) * 8); rs.d_len.ensure((rs.n + 1) * 4);
			h2d(rs.d_coff.as<u64>(), rs.h_coff.data(), rs.n + 1, stream);
			h2d(rs.d_len.as<u32>(), rs.h_len.data(), rs.n, stream);
			amb_tails(rs, rs.n - n, rs.n, chunk0, new_chunks);
		}
		LQ_HIP_CHECK(hipStreamSynchronize(stream));               // the caller's buffers are free again
		return;
	}
	u64 n_tiles = 0;
	if (first && K.upload_slices > 1 && new_chunks >= K.upload_min_chunks && sketch_dp_setup(rs, n_tiles)) {
		// The reads go up in slices on a stream of their own; the data-parallel sketch kernel takes the tiles of a slice on the build
		// stream as soon as the slice has arrived -- under the upload of the next one (a 4-Gbase part: 26 ms of upload and 34 ms of
		// kernel one after the other before).  The caller's buffers are free when this returns; the last slice may still be sketched.
		StageTimer t(this, stream, "h2d_packed_reads", n_words * (amb ? 12 : 8));
		if (!amb) amb_tails(rs, 0, n, chunk0, new_chunks);         // (no read holds an ambiguous base: the bits past the reads' ends are made here, on the build stream, before the first tile)
		const u32 ns = K.upload_slices;
		u32 r0 = 0;
		for (u32 sl = 0; sl < ns; ++sl) {
			u32 r1 = sl + 1 == ns ? n : r0;
			const u64 want = new_chunks * (u64)(sl + 1) / ns;
			while (r1 < n && rs.h_coff[r1] < want) ++r1;           // (h_coff[r] = chunks before read r: chunk0 == 0 here)
			if (r1 > r0) {
				const u64 w0 = rs.h_coff[r0] * LQ_CHUNK_WORDS, w1 = rs.h_coff[r1] * LQ_CHUNK_WORDS;
				LQ_HIP_CHECK(hipMemcpyAsync(rs.codes.as<u64>() + w0, codes + w0, (w1 - w0) * 8, hipMemcpyHostToDevice, cstream));
				if (amb) LQ_HIP_CHECK(hipMemcpyAsync(rs.amb.as<u32>() + w0, amb + w0, (w1 - w0) * 4, hipMemcpyHostToDevice, cstream));
				LQ_HIP_CHECK(hipEventRecord(ev_up[sl], cstream));
				LQ_HIP_CHECK(hipStreamWaitEvent(stream, ev_up[sl], 0));
				sketch_dp_launch(rs, sk_h_toff[r0], sk_h_toff[r1]);
			}
			r0 = r1;
		}
		LQ_HIP_CHECK(hipStreamSynchronize(cstream));          // the caller's buffers are free again
		rs.dp_n = rs.n; rs.dp_tiles = n_tiles; rs.dp_gen = sk_gen;
		return;
	}
	(void)first; (void)chunk0;
	{
		StageTimer t(this, stream, "h2d_packed_reads", n_words * (amb ? 12 : 8));
		h2d(rs.codes.as<u64>() + chunk0 * LQ_CHUNK_WORDS, codes, n_words, stream);
		if (amb) h2d(rs.amb.as<u32>() + chunk0 * LQ_CHUNK_WORDS, amb, n_words, stream);
		else {
			rs.d_coff.ensure((rs.n + 1) * 8); rs.d_len.ensure((rs.n + 1) * 4);
			h2d(rs.d_coff.as<u64>(), rs.h_coff.data(), rs.n + 1, stream);
			h2d(rs.d_len.as<u32>(), rs.h_len.data(), rs.n, stream);
			amb_tails(rs, rs.n - n, rs.n, chunk0, new_chunks);
		}
	}
	LQ_HIP_CHECK(hipStreamSynchronize(stream));               // the caller's buffers are free again
}

// the ambiguity words of the reads [r0, r1) -- chunks [chunk0, chunk0 + n_chunks) -- when none of them holds an ambiguous base: zero but for the
// positions past every read's end (rs.d_coff / rs.d_len describe the reads already)
void lqcov_handle::amb_tails(ReadSetDev &rs, u32 r0, u32 r1, u64 chunk0, u64 n_chunks)
{
	hipStream_t stream = this->bstream;
	if (r1 <= r0) return;
	dzero(rs.amb.as<u32>() + chunk0 * LQ_CHUNK_WORDS, n_chunks * LQ_CHUNK_WORDS * 4, stream);
	LQ_LAUNCH(k_amb_tails, nblk(r1 - r0, 256), 256, stream, rs.d_coff.as<u64>(), rs.d_len.as<u32>(), r0, r1, rs.amb.as<u32>()); check_launch();
}

// What k_sketch_dp_mask needs before its first tile: read offsets and lengths, the tiles of every read, an empty mask.  false: the
// data-parallel kernel does not apply (-H, a window or k-mer beyond its reach, LQCOV_SKETCH=machine).
bool lqcov_handle::sketch_dp_setup(ReadSetDev &rs, u64 &n_tiles)
{
	hipStream_t stream = this->bstream;
	const u64 nc = rs.n_chunks;
	if (P.hpc || !(P.w <= 16 && P.w + P.k - 1 <= 48 && P.k <= 28 && P.k >= 2) || K.sketch_machine_only || !nc) return false;
	rs.d_coff.ensure((rs.n + 1) * 8); rs.d_len.ensure((rs.n + 1) * 4);
	h2d(rs.d_coff.as<u64>(), rs.h_coff.data(), rs.n + 1, stream);
	h2d(rs.d_len.as<u32>(), rs.h_len.data(), rs.n, stream);
	sk_mask.ensure(nc * LQ_CHUNK_WORDS * 4 + 64); sk_flag.ensure(4);
	dzero(sk_mask.p, nc * LQ_CHUNK_WORDS * 4, stream); dzero(sk_flag.p, 4, stream);
	sk_grid.ensure((nc / LQ_EM_CH + 2) * 4);
	sk_owner = &rs; ++sk_gen;                                 // (the handle's mask, owner and tile buffers now describe these reads and nobody else's)
	sk_h_toff.assign(rs.n + 1, 0);
	for (u32 r = 0; r < rs.n; ++r) sk_h_toff[r + 1] = sk_h_toff[r] + (rs.h_coff[r + 1] - rs.h_coff[r] + LQ_DPT_CH - 1) / LQ_DPT_CH;
	n_tiles = sk_h_toff[rs.n];
	sk_toff.ensure((rs.n + 1) * 8); sk_owned.ensure(nc + 8); sk_trid.ensure(n_tiles * 4 + 4);
	h2d(sk_toff.as<u64>(), sk_h_toff.data(), rs.n + 1, stream);
	LQ_LAUNCH(k_sketch_owners, nblk(rs.n, 256), 256, stream, rs.d_coff.as<u64>(), sk_toff.as<u64>(), rs.n, sk_trid.as<u32>(), sk_grid.as<u32>()); check_launch();
	return true;
}

// k_sketch_dp_mask over the tiles [tile0, tile1) of the read set
void lqcov_handle::sketch_dp_launch(ReadSetDev &rs, u64 tile0, u64 tile1)
{
	hipStream_t stream = this->bstream;
	if (tile1 <= tile0) return;
	SkParams sp; sp.k = P.k; sp.w = P.w; sp.hpc = P.hpc; sp.mask = (1ULL << 2 * P.k) - 1; sp.shift1 = 2 * (P.k - 1);
	const u64 nt = tile1 - tile0;
	StageTimer t(this, stream, "k_sketch_dp_mask", nt * LQ_DPT_CH * (LQ_CHUNK_WORDS * 12 + 17));
#define LQ_DPM(HT, W) LQ_LAUNCH((k_sketch_dp_mask<HT, W>), (u32)std::min<u64>(nt, K.sketch_grid), LQ_DPT_THREADS, stream, rs.codes.as<u64>(), rs.amb.as<u32>(), rs.d_coff.as<u64>(), rs.d_len.as<u32>(), \
		sk_toff.as<u64>(), sk_trid.as<u32>(), rs.n, tile0, tile1, sp, sk_owned.as<u8>(), sk_mask.as<u32>(), sk_flag.as<u32>())
	const int wc = K.sketch_wgen ? 0 : P.w;                       // the presets' windows as compile-time constants
	if (K.sketch_fast && !K.sketch_wgen && P.k == 12 && (P.w == 5 || P.w == 10)) {   // LongQC's own -k 12 -w 5: k and w as constants (LQCOV_SKETCH_FAST=0: the general kernel)
#define LQ_DPF(KK, WW) LQ_LAUNCH((k_sketch_dp_fast<KK, WW>), (u32)std::min<u64>(nt, K.sketch_grid), LQ_DPT_THREADS, stream, rs.codes.as<u64>(), rs.amb.as<u32>(), rs.d_coff.as<u64>(), rs.d_len.as<u32>(), \
		sk_toff.as<u64>(), sk_trid.as<u32>(), rs.n, tile0, tile1, sk_owned.as<u8>(), sk_mask.as<u32>(), sk_flag.as<u32>())
		if (P.w == 5) LQ_DPF(12, 5); else LQ_DPF(12, 10);
#undef LQ_DPF
	}
	else if (P.k <= 16) { if (wc == 5) LQ_DPM(u32, 5); else if (wc == 10) LQ_DPM(u32, 10); else LQ_DPM(u32, 0); }
	else { if (wc == 5) LQ_DPM(u64, 5); else if (wc == 10) LQ_DPM(u64, 10); else LQ_DPM(u64, 0); }
#undef LQ_DPM
	check_launch();
}

// minimizers of every read of the set, in (read, position) order   (sketch.c:76-142)
void lqcov_handle::sketch(ReadSetDev &rs, bool rid_in_y)
{
	hipStream_t stream = this->bstream; Prim &prim = this->bprim;   // the build side has a stream and scan / sort scratch of its own: a part can be built while another is mapped
	(void)prim;
	static const bool timing = getenv("LQCOV_TIMING") != nullptr;
	double t_last = lq_now_s();
	auto lap = [&](const char *what) { if (!timing) return; hipStreamSynchronize(stream); const double t = lq_now_s(); fprintf(stderr, "[timing] sketch %-28s %.3f s (allocations so far %.3f s)\n", what, t - t_last, (double)lq_alloc_ns * 1e-9); t_last = t; };
	rs.d_coff.ensure((rs.n + 1) * 8); rs.d_len.ensure((rs.n + 1) * 4);
	h2d(rs.d_coff.as<u64>(), rs.h_coff.data(), rs.n + 1, stream);
	h2d(rs.d_len.as<u32>(), rs.h_len.data(), rs.n, stream);
	rs.moff.ensure((rs.n + 1) * 8);
	rs.n_mini = 0;
	rs.key_stamp = 0;
	const u64 nc = rs.n_chunks;
	if (nc) {
		SkParams sp; sp.k = P.k; sp.w = P.w; sp.hpc = P.hpc; sp.mask = (1ULL << 2 * P.k) - 1; sp.shift1 = 2 * (P.k - 1);
		lap("offsets up");
		DBuf &cnt = sk_cnt, &off = sk_off;
		cnt.ensure(nc * 4); off.ensure(nc * 8);
		lap("cnt/off ensure");
		const u64 in_bytes = nc * (LQ_CHUNK_WORDS * 12);
		// ring capacity 8 / 16 (LDS) or 256 (private), -H on/off: pick the instantiation
		// chunks per thread: the halo before a thread's first chunk is walked once per kpt chunks
		const u32 kpt = K.sketch_kpt;
		const u32 *sk_list = nullptr, *sk_nlist = nullptr;           // mask mode beside the data-parallel kernel: the chunks it left, listed
#define LQ_SK_LAUNCH(RC, EM, HP, BS, ...) LQ_LAUNCH((k_sketch<RC, EM, HP>), sk_list ? (u32)std::min<u64>(nblk(nc, BS), 2048) : nblk((nc + kpt - 1) / kpt, BS), BS, stream, rs.codes.as<u64>(), rs.amb.as<u32>(), rs.d_coff.as<u64>(), rs.d_len.as<u32>(), rs.n, nc, kpt, sp, (int)rid_in_y, __VA_ARGS__, sk_list, sk_nlist)
#define LQ_SK_DISPATCH(EM, HP, ...) do { \
		if (P.w <= 8)       LQ_SK_LAUNCH(8, EM, HP, LQ_SK_BLOCK, __VA_ARGS__); \
		else if (P.w <= 16) LQ_SK_LAUNCH(16, EM, HP, LQ_SK_BLOCK, __VA_ARGS__); \
		else                LQ_SK_LAUNCH(256, EM, HP, 64, __VA_ARGS__); \
	} while (0)
		if (!P.hpc) {
			// Which positions are emitted is decided once, as a bit per base (mask: 4 words per chunk): by the data-parallel kernel
			// over tiles of 12 chunks wherever the machine is memoryless, by the state machine for what that kernel leaves (the
			// first chunk of every read, tiles within reach of an N, AT-repeat halos).  The list is then made from the mask: no
			// second run of the machine, no halo for the output pass.
			const bool dp = P.w <= 16 && P.w + P.k - 1 <= 48 && P.k <= 28 && P.k >= 2 && !K.sketch_machine_only;
			const u8 *dp_owned = nullptr;
			if (!dp) {
				sk_owner = &rs; ++sk_gen;
				sk_mask.ensure(nc * LQ_CHUNK_WORDS * 4 + 64); sk_flag.ensure(4);
				dzero(sk_mask.p, nc * LQ_CHUNK_WORDS * 4, stream); dzero(sk_flag.p, 4, stream);
				sk_grid.ensure((nc / LQ_EM_CH + 2) * 4);
				LQ_LAUNCH(k_sketch_owners, nblk(rs.n, 256), 256, stream, rs.d_coff.as<u64>(), (const u64*)nullptr, rs.n, (u32*)nullptr, sk_grid.as<u32>()); check_launch();
			}
			if (dp) {
				// (the tiles may be done already: add_reads_packed runs the kernel slice by slice under the upload)
				u64 n_tiles = rs.dp_tiles;
				// (the tiles done ahead of time count only if the handle's mask buffers are still this read set's: a query set, or another
				// part added in packed form, sketched in between has taken them over)
				if (rs.dp_n != rs.n || sk_owner != &rs || rs.dp_gen != sk_gen) { if (!sketch_dp_setup(rs, n_tiles)) throw std::logic_error("sketch: set-up of the data-parallel kernel"); sketch_dp_launch(rs, 0, n_tiles); }
				rs.dp_n = 0; rs.dp_tiles = 0;                          // (the mask is used up below: a second sketch of the same reads starts from an empty one)
				lap("dp_mask");
				dp_owned = sk_owned.as<u8>();
				if (K.debug_sort) {
					std::vector<u8> ho(nc);
					d2h(ho.data(), sk_owned.as<u8>(), nc, stream);
					u64 n_own = 0;
					for (u8 v : ho) n_own += v;
					fprintf(stderr, "[sketch] %llu of %llu chunks decided data-parallel\n", (unsigned long long)n_own, (unsigned long long)nc);
				}
			}
			if (dp && K.sketch_list && nc < 0xffffffffULL) {
				sk_ulist.ensure(nc * 4 + 4);                             // (entry nc: the count)
				dzero(sk_ulist.as<u32>() + nc, 4, stream);
				LQ_LAUNCH(k_sketch_unowned, nblk(nc, 256), 256, stream, dp_owned, nc, sk_ulist.as<u32>(), sk_ulist.as<u32>() + nc); check_launch();
				sk_list = sk_ulist.as<u32>(); sk_nlist = sk_ulist.as<u32>() + nc;
			}
			{
				StageTimer t(this, stream, "k_sketch_mask", dp ? nc : in_bytes + nc * 16);
				LQ_SK_DISPATCH(LQ_SK_MASK, false, (u32*)nullptr, (const u64*)nullptr, (u64*)nullptr, (u64*)nullptr, dp_owned, sk_mask.as<u32>(), sk_flag.as<u32>());
				check_launch();
			}
			lap("state machine");
			sk_list = sk_nlist = nullptr;
			LQ_LAUNCH(k_mask_count, nblk(nc, 256), 256, stream, sk_mask.as<u32>(), nc, cnt.as<u32>()); check_launch();
		} else {
			StageTimer t(this, stream, "k_sketch_count", in_bytes + nc * 4);
			LQ_SK_DISPATCH(LQ_SK_COUNT, true, cnt.as<u32>(), (const u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (const u8*)nullptr, (u32*)nullptr, (u32*)nullptr);
			check_launch();
		}
		{ StageTimer t(this, stream, "scan"); prim.exclusive_scan_u32_u64(cnt.as<u32>(), off.as<u64>(), nc); }
		u64 last_off = 0; u32 last_cnt = 0;
		d2h(&last_off, off.as<u64>() + nc - 1, 1, stream);
		d2h(&last_cnt, cnt.as<u32>() + nc - 1, 1, stream);
		rs.n_mini = last_off + last_cnt;
		lap("count + scan");
		rs.mx.ensure(rs.n_mini * 8 + 8); rs.my.ensure(rs.n_mini * 8 + 8);
		lap("mx/my ensure");
		if (!P.hpc) {
			u32 dup = 0;
			d2h(&dup, sk_flag.as<u32>(), 1, stream);
			if (dup) throw std::logic_error("sketch: a position was emitted twice (mask form of the minimizer list does not hold)");
			StageTimer t(this, stream, "k_sketch_emit_mask", nc * 24 + rs.n_mini * (16 + 4));
			// (a part's minimizers: the hash alone as well, where it fits 32 bits -- the index sort's key, which k_sort_keys otherwise makes from x)
			u32 *okey = nullptr;
			if (rid_in_y && 2 * P.k <= 32 && K.sketch_key) { ix_key.ensure(rs.n_mini * 8); okey = ix_key.as<u32>(); rs.key_stamp = ix_key_stamp = ++ix_key_seq; }
			LQ_LAUNCH(k_sketch_emit_mask, (u32)std::min<u64>((nc + LQ_EM_CH - 1) / LQ_EM_CH, K.emit_grid), LQ_EM_THREADS, stream, rs.codes.as<u64>(), rs.amb.as<u32>(), rs.d_coff.as<u64>(), sk_grid.as<u32>(), rs.n, nc, sp, (int)rid_in_y,
			          sk_mask.as<u32>(), off.as<u64>(), rs.mx.as<u64>(), rs.my.as<u64>(), okey);
			check_launch();
		} else {
			StageTimer t(this, stream, "k_sketch_emit", in_bytes + nc * 8 + rs.n_mini * 16);
			LQ_SK_DISPATCH(LQ_SK_EMIT, true, (u32*)nullptr, off.as<u64>(), rs.mx.as<u64>(), rs.my.as<u64>(), (const u8*)nullptr, (u32*)nullptr, (u32*)nullptr);
			check_launch();
		}
#undef LQ_SK_DISPATCH
#undef LQ_SK_LAUNCH
		lap("emit");
		LQ_LAUNCH(k_read_moff, nblk(rs.n + 1, 256), 256, stream, rs.d_coff.as<u64>(), off.as<u64>(), rs.n, nc, rs.n_mini, rs.moff.as<u64>());
		check_launch();
		LQ_HIP_CHECK(hipStreamSynchronize(stream));
	} else {
		dzero(rs.moff.p, (rs.n + 1) * 8, stream);
		LQ_HIP_CHECK(hipStreamSynchronize(stream));
	}
	rs.sketched = true;
}

void lqcov_handle::export_minimizers(ReadSetDev &rs, u64 *x_dev, u64 *y_dev, u32 rid_base)
{
	hipStream_t stream = this->bstream; Prim &prim = this->bprim;   // the build side has a stream and scan / sort scratch of its own: a part can be built while another is mapped
	(void)prim;
	const u64 n = rs.n_mini;
	if (!n) return;
	LQ_HIP_CHECK(hipMemcpyAsync(x_dev, rs.mx.p, n * 8, hipMemcpyDeviceToDevice, stream));
	LQ_LAUNCH(k_rebase_y, nblk(n, 256), 256, stream, rs.my.as<u64>(), n, (u64)rid_base << 32, y_dev); check_launch();
	LQ_HIP_CHECK(hipStreamSynchronize(stream));
}

// meanQ's table (lqutils.c:26-49): 127 15-decimal literals for 10^(-q/10), Q0..Q126.  They are 10^(-q/10)
// rounded to 15 decimals, except that eight entries (Q34, 39, 58, 62, 67, 71, 72, 82) are one unit of the
// 15th decimal higher in the reference; rebuilt here from that description.
static void make_q2p(double *t)
{
	static const int up[8] = {34, 39, 58, 62, 67, 71, 72, 82};
	for (int q = 0; q < 127; ++q) {
		char buf[64];
		snprintf(buf, sizeof(buf), "%.15f", pow(10.0, -q / 10.0));
		long long units = (long long)(buf[0] - '0') * 1000000000000000LL + strtoll(buf + 2, nullptr, 10);
		for (int j = 0; j < 8; ++j) if (up[j] == q) ++units;
		snprintf(buf, sizeof(buf), "%lld.%015lld", units / 1000000000000000LL, units % 1000000000000000LL);
		t[q] = strtod(buf, nullptr);
	}
}

// == main pass 1 (minimap2-coverage.c:406-444)
void lqcov_handle::set_queries(u32 n, const u8 *seq_in, const u64 *seq_off_in, const u8 *qual_in, const char *names_in, const u64 *name_off_in)
{
	if (have_queries) throw std::logic_error("queries already set");
	// internal order: longest first (stable); LQCOV_QUERY_ORDER=file keeps the caller's order (A/B and test knob)
	q_perm.resize(n); q_inv.resize(n);
	for (u32 i = 0; i < n; ++i) q_perm[i] = i;
	if (!K.query_order_file) {
		std::stable_sort(q_perm.begin(), q_perm.end(), [&](u32 a, u32 b) { return seq_off_in[a + 1] - seq_off_in[a] > seq_off_in[b + 1] - seq_off_in[b]; });
		// LQCOV_QUERY_ORDER=striped (round 6, measured, not the default): dealt to as many stripes as there are mapping lanes, the stripes
		// one after the other, so that a part's batches -- contiguous ranges of this order, cut by the anchors the first pass writes --
		// hold like samples of the queries.  At configs[2] the lanes' second passes then carry 151 / 167 / 149 M anchors instead of
		// 178 / 165 / 124 M and the step takes as long (373 against 366-370 ms): a lane's second pass is a chain of a dozen kernels
		// whose length does not go by the batch (profiles/README.md, round 6).
		const u32 ns = (u32)std::max(1, K.lanes);
		if (K.query_order_striped && ns > 1 && n > ns) {
			std::vector<u32> dealt; dealt.reserve(n);
			for (u32 st = 0; st < ns; ++st) for (u32 i = st; i < n; i += ns) dealt.push_back(q_perm[i]);
			q_perm.swap(dealt);
		}
	}
	for (u32 i = 0; i < n; ++i) q_inv[q_perm[i]] = i;
	std::vector<u8> pseq, pqual;
	std::vector<u64> pseq_off(n + 1, 0), pname_off(n + 1, 0);
	std::vector<char> pnames;
	{
		const u64 nb = n ? seq_off_in[n] - seq_off_in[0] : 0;
		pseq.resize(nb + 1); if (qual_in) pqual.resize(nb + 1);
		for (u32 i = 0; i < n; ++i) {
			const u32 o = q_perm[i];
			const u64 len = seq_off_in[o + 1] - seq_off_in[o];
			memcpy(pseq.data() + pseq_off[i], seq_in + seq_off_in[o], len);
			if (qual_in) memcpy(pqual.data() + pseq_off[i], qual_in + seq_off_in[o], len);
			pseq_off[i + 1] = pseq_off[i] + len;
			const char *nm = names_in ? names_in + name_off_in[o] : "";
			pnames.insert(pnames.end(), nm, nm + strlen(nm) + 1);
			pname_off[i + 1] = pnames.size();
		}
	}
	const u8 *seq = pseq.data(), *qual = qual_in ? pqual.data() : nullptr;
	const u64 *seq_off = pseq_off.data(), *name_off = pname_off.data();
	const char *names = pnames.data();
	if (seq_off[n] - seq_off[0] >= 500000000ULL && n > 1) {
		// reference: a second 500-Mbase query mini-batch aliases the accumulator slots and crashes (lqmap.c:714,735)
		u64 but_last = seq_off[n - 1] - seq_off[0];
		if (but_last >= 500000000ULL) throw std::domain_error("query set spans more than one 500-Mbase mini-batch: outside the reference's domain (lqmap.c:714)");
	}
	add_reads(q, n, seq, seq_off, names, name_off);
	q.n = n;                                                // (add_reads returns early for n == 0)
	sketch(q, false);
	have_queries = true;
	q_has_qual = qual != nullptr;
	const u64 nm = q.n_mini;
	q_owner.ensure(nm * 4 + 4);
	if (nm) { LQ_LAUNCH(k_minimizer_owner, nblk(nm, 256), 256, stream, q.moff.as<u64>(), n, nm, q_owner.as<u32>()); check_launch(); }
	lambda.ensure((n + 1) * 8); lambda2.ensure((n + 1) * 8); avg_k.ensure((n + 1) * 4); qflags.ensure((n + 1) * 4);
	cnts.ensure(nm * 4 + 4); qual_psum.ensure((n + 1) * 8);
	n_pv.ensure(4);
	reset();
	dzero(qual_psum.p, (n + 1) * 8, stream);
	if (qual && n) {
		const u64 nb = seq_off[n] - seq_off[0];
		DBuf dq, dso, dtab;
		dq.ensure(nb + 16); dso.ensure((n + 1) * 8); dtab.ensure(127 * 8);
		std::vector<u64> soff(n + 1);
		for (u32 i = 0; i <= n; ++i) soff[i] = seq_off[i] - seq_off[0];
		double tab[127]; make_q2p(tab);
		h2d(dq.as<u8>(), qual + seq_off[0], nb, stream);
		h2d(dso.as<u64>(), soff.data(), n + 1, stream);
		h2d(dtab.as<double>(), tab, 127, stream);
		{
			StageTimer t(this, "k_qual_sum", nb);
			LQ_LAUNCH(k_qual_sum, nblk(n, 64), 64, stream, dq.as<u8>(), dso.as<u64>(), n, dtab.as<double>(), qual_psum.as<double>());
			check_launch();
		}
		LQ_HIP_CHECK(hipStreamSynchronize(stream));
	}
}

void lqcov_handle::reset()
{
	const u32 n = q.n;
	dzero(lambda.p, (n + 1) * 8, stream); dzero(lambda2.p, (n + 1) * 8, stream);
	dzero(avg_k.p, (n + 1) * 4, stream); dzero(qflags.p, (n + 1) * 4, stream);
	dzero(cnts.p, cnt_count() * 4 + 4, stream);
	dzero(n_pv.p, 4, stream);
	if (!distributed) mid_occ = -1;
	stat_sens_runs = 0; stat_p2_queries = 0; stat_p2_anchors = 0;
	for (auto &v : stat_tie_why) v = 0;
	sat_cnt.clear(); stat_sat_chains = 0;
	finished = false;
	LQ_HIP_CHECK(hipStreamSynchronize(stream));
	if (lq_timeline) { lq_timeline_t0 = lq_now_s(); lq_tl("main", 0, "reset"); }
}

// ---- index part -------------------------------------------------------------------------------
// hash -> occurrences sorted by y (index.c:150-201), mid_occ (index.c:123-144)
void lqcov_handle::build_index(Part &pt)
{
	hipStream_t stream = this->bstream; Prim &prim = this->bprim;   // the build side has a stream and scan / sort scratch of its own: a part can be built while another is mapped
	(void)prim;
	ReadSetDev &rs = pt.rs;
	const u64 M = rs.n_mini;
	ix_owner = &pt;
	pt.n_keys = 0; pt.cap_bits = 4;
	pt.pos.ensure(M * 8 + 8);
	// same-name targets per query (self diagonal, lqmap.c:180-186) and -X's name ranks: host work on the read names, done while
	// the device sorts (host_names is called between the last launch and the first wait)
	std::vector<u32> soff(q.n + 1, 0), srid;
	bool names_done = false;
	auto host_names = [&]() {
		names_done = true;
		if (P.no_self && q.n) {
			std::unordered_map<std::string, std::vector<u32>> byname;
			for (u32 i = 0; i < q.n; ++i) byname[q.names[i]].push_back(i);
			std::vector<std::vector<u32>> per(q.n);
			for (u32 r = 0; r < rs.n; ++r) {
				auto it = byname.find(rs.names[r]);
				if (it != byname.end()) for (u32 qi : it->second) per[qi].push_back(r);
			}
			for (u32 i = 0; i < q.n; ++i) { soff[i + 1] = soff[i] + (u32)per[i].size(); srid.insert(srid.end(), per[i].begin(), per[i].end()); }
		}
		if (P.ava) {	// -X: strcmp(qname, tname) > 0 drops the hit (lqmap.c:187) -> ranks among the part's distinct names
			std::vector<std::string> names(rs.names.begin(), rs.names.end());
			std::sort(names.begin(), names.end());
			names.erase(std::unique(names.begin(), names.end()), names.end());
			std::vector<u32> tr(rs.n + 1, 0), ql(q.n + 1, 0);
			for (u32 r = 0; r < rs.n; ++r) tr[r] = (u32)(std::lower_bound(names.begin(), names.end(), rs.names[r]) - names.begin());
			for (u32 i = 0; i < q.n; ++i) ql[i] = (u32)(std::lower_bound(names.begin(), names.end(), q.names[i]) - names.begin());
			pt.t_rank.ensure((rs.n + 1) * 4); pt.q_lo.ensure((q.n + 1) * 4);
			h2d(pt.t_rank.as<u32>(), tr.data(), rs.n + 1, stream);
			h2d(pt.q_lo.as<u32>(), ql.data(), q.n + 1, stream);
			LQ_HIP_CHECK(hipStreamSynchronize(stream));
		}
	};
	if (M) {
		DBuf &key = ix_key, &key2 = ix_key2, &head = ix_head, &uidx = ix_uidx, &ukey = ix_ukey, &ustart = ix_ustart, &ucnt = ix_ucnt;   // workspaces live with the handle: repeated builds do not re-allocate
		const u64 n_tiles = (M + LQ_HEAD_TILE - 1) / LQ_HEAD_TILE;
		key.ensure(M * 8); key2.ensure(M * 8); head.ensure((n_tiles + 1) * 4); uidx.ensure((n_tiles + 1) * 8);   // (head / uidx: run heads per tile of keys, scanned)
		const bool k32 = 2 * P.k <= 32;                         // the hash fits 32 bits: 4-byte sort keys
		const bool fused_heads = 2 * P.k <= 24;                   // at most 2^24 distinct keys: the run heads in one pass (k_head_lookback)
		if (k32) {
			if (!(rs.key_stamp && rs.key_stamp == ix_key_stamp)) { LQ_LAUNCH(k_sort_keys<u32>, nblk(M, 256), 256, stream, rs.mx.as<u64>(), M, key.as<u32>()); check_launch(); }   // (k_sketch_emit_mask has written them)
			ix_key_stamp = 0;
			{ StageTimer t(this, stream, "index_radix_sort", M * 24); prim.sort_pairs_u32_u64(key.as<u32>(), key2.as<u32>(), rs.my.as<u64>(), pt.pos.as<u64>(), M, (unsigned)(2 * P.k), key.as<u32>()); }   // (the unsorted keys are not needed again: their array is the sort's second key buffer)
			if (!fused_heads) { LQ_LAUNCH(k_head_count<u32>, (u32)n_tiles, LQ_HEAD_THREADS, stream, key2.as<u32>(), M, head.as<u32>()); check_launch(); }
		} else {
			ix_key_stamp = 0;
			LQ_LAUNCH(k_sort_keys<u64>, nblk(M, 256), 256, stream, rs.mx.as<u64>(), M, key.as<u64>()); check_launch();
			{ StageTimer t(this, stream, "index_radix_sort", M * 32); prim.sort_pairs_u64(key.as<u64>(), key2.as<u64>(), rs.my.as<u64>(), pt.pos.as<u64>(), M, (unsigned)(2 * P.k), key.as<u64>()); }
			LQ_LAUNCH(k_head_count<u64>, (u32)n_tiles, LQ_HEAD_THREADS, stream, key2.as<u64>(), M, head.as<u32>()); check_launch();
		}
		u64 K = 0;
		if (fused_heads) {
			const u64 k_max = std::min<u64>(M, (u64)1 << (2 * P.k));
			ukey.ensure(k_max * 8); ustart.ensure(k_max * 8);
			const u64 n_blocks = (n_tiles + LQ_HEADLB_SUB - 1) / LQ_HEADLB_SUB;
			uidx.ensure((n_blocks + 2) * 8 + 64);                   // (one look-back granule per block, the count, the ticket)
			dzero(uidx.p, (n_blocks + 2) * 8 + 64, stream);
			u64 *granules = uidx.as<u64>(), *count = granules + n_blocks; u32 *ticket = (u32*)(granules + n_blocks + 1);
			{
				StageTimer t(this, stream, "k_head_lookback", M * 4);
				LQ_LAUNCH(k_head_lookback<u32>, (u32)n_blocks, LQ_HEAD_THREADS, stream, key2.as<u32>(), M, granules, ticket, ukey.as<u64>(), ustart.as<u64>(), count);
				check_launch();
			}
			host_names();                                              // (host work while the sort runs: the first wait for the device is below)
			d2h(&K, count, 1, stream);
			pt.n_keys = K;
			ucnt.ensure(K * 4);
		} else {
			dzero(head.as<u32>() + n_tiles, 4, stream);
			prim.exclusive_scan_u32_u64(head.as<u32>(), uidx.as<u64>(), n_tiles + 1);
			host_names();                                              // (host work while the sort runs: the first wait for the device is below)
			d2h(&K, uidx.as<u64>() + n_tiles, 1, stream);
			pt.n_keys = K;
			ukey.ensure(K * 8); ustart.ensure(K * 8); ucnt.ensure(K * 4);
			if (k32) LQ_LAUNCH(k_head_fill<u32>, (u32)n_tiles, LQ_HEAD_THREADS, stream, key2.as<u32>(), M, uidx.as<u64>(), ukey.as<u64>(), ustart.as<u64>());
			else LQ_LAUNCH(k_head_fill<u64>, (u32)n_tiles, LQ_HEAD_THREADS, stream, key2.as<u64>(), M, uidx.as<u64>(), ukey.as<u64>(), ustart.as<u64>());
			check_launch();
		}
		LQ_LAUNCH(k_unique_counts, nblk(K, 256), 256, stream, ustart.as<u64>(), K, M, ucnt.as<u32>()); check_launch();
		u32 bits = 4;
		while (((u64)1 << bits) < 2 * K) ++bits;
		pt.cap_bits = bits;
		const u64 cap = (u64)1 << bits;
		pt.tkey.ensure(cap * 8); pt.tstart.ensure(cap * 8); pt.tcnt.ensure(cap * 4);
		LQ_HIP_CHECK(hipMemsetAsync(pt.tkey.p, 0xff, cap * 8, stream));
		{
			StageTimer t(this, stream, "k_table_insert", K * 20 + cap * 8);
			LQ_LAUNCH(k_table_insert, nblk(K, 256), 256, stream, ukey.as<u64>(), ustart.as<u64>(), ucnt.as<u32>(), K, pt.tkey.as<u64>(), pt.tstart.as<u64>(), pt.tcnt.as<u32>(), bits);
			check_launch();
		}
		if (mid_occ <= 0) {                                   // map.c:50: from the first part only
			if (P.mid_occ_frac <= 0.0f) mid_occ = INT32_MAX;
			else {
				DBuf &sorted = ix_sorted; sorted.ensure(K * 4);
				prim.sort_keys_u32(ucnt.as<u32>(), sorted.as<u32>(), K);
				const u32 kth = (u32)((1. - P.mid_occ_frac) * (double)K);   // index.c:141
				u32 v = 0;
				d2h(&v, sorted.as<u32>() + kth, 1, stream);
				mid_occ = (i32)(v + 1);
			}
		}
		LQ_HIP_CHECK(hipStreamSynchronize(stream));
	} else {
		const u64 cap = (u64)1 << pt.cap_bits;
		pt.tkey.ensure(cap * 8); pt.tstart.ensure(cap * 8); pt.tcnt.ensure(cap * 4);
		LQ_HIP_CHECK(hipMemsetAsync(pt.tkey.p, 0xff, cap * 8, stream));
		if (mid_occ <= 0) mid_occ = P.mid_occ_frac <= 0.0f ? INT32_MAX : 1;   // reference reads an empty array here; unobservable
	}
	if (!names_done) host_names();
	pt.self_off.ensure((q.n + 1) * 4); pt.self_rid.ensure(srid.size() * 4 + 4);
	h2d(pt.self_off.as<u32>(), soff.data(), q.n + 1, stream);
	h2d(pt.self_rid.as<u32>(), srid.data(), srid.size(), stream);
	LQ_HIP_CHECK(hipStreamSynchronize(stream));
	pt.built = true;
	pt.plan.valid = false;
	// the part's seed plan right away, on the build stream: under the mapping of the part before when parts are pipelined
	lq_tl("build", 0, "index done");
	// (with the lanes idle -- the first part of a job -- nothing hides the filter, the plan's longest stage: it is left to the lanes,
	// each of which decides its own batch of queries and starts mapping while the next lane decides its batch: map_part)
	if (K.plan_ahead && have_queries && mid_occ > 0 && !distributed) plan_part(pt, stream, prim, K.plan_lazy && active_maps.load() == 0);
	// (the build workspaces stay with the handle: repeated builds do not re-allocate, and the mapping lanes size their work
	// space from what is free once the first part stands -- map_part)
}

void lqcov_handle::build_part(Part &pt)
{
	if (!have_queries) throw std::logic_error("set the queries before building a part");
	lq_tl("build", 0, "build_part begins (reads uploaded)");
	sketch(pt.rs, true);
	lq_tl("build", 0, "sketched");
	build_index(pt);
	lq_tl("build", 0, "index + seed plan done");
}

// LQCOV_DEBUG_SORT: the sort is a permutation (a sum over the anchors as emitted = the same sum afterwards: the finishing
// kernels rebuild x from the key), every query is ascending in x, anchors with equal x both carry the tie mark
static u64 dbg_anchor_sum(const mm128 &a) { u64 h = (a.x * 0x9E3779B97F4A7C15ULL) ^ (a.y * 0xD6E8FEB86659FD93ULL); return h ^ (h >> 29); }

// sort_batch with the LQCOV_DEBUG_SORT checks around it; h_off: the batch's per-query anchor offsets (relative, nqb + 1), h_klib: which queries' anchors are in B
void lqcov_handle::sort_checked(MapLane &L, Part &pt, const u64 *aqb, const u32 *qkb, u32 nqb, u64 a_base, u64 nA, const std::vector<u64> &h_off, const std::vector<u32> &h_klib)
{
	mm128 *dA = L.A.as<mm128>(), *dB = L.B.as<mm128>();
	u64 sum_before = 0;
	if (K.debug_sort) {
		std::vector<mm128> ha(nA), hb(nA);
		d2h(ha.data(), dA, nA, L.stream); d2h(hb.data(), dB, nA, L.stream);
		for (u32 q = 0; q < nqb; ++q)
			for (u64 i = h_off[q]; i < h_off[q + 1]; ++i) sum_before += dbg_anchor_sum(h_klib[q] ? hb[i] : ha[i]);
	}
	sort_batch(L, pt, aqb, qkb, nqb, a_base, nA);           // lqmap.c:238
	if (K.debug_sort) {
		std::vector<mm128> ha(nA);
		d2h(ha.data(), dA, nA, L.stream);
		u64 sum_after = 0;
		for (u64 i = 0; i < nA; ++i) sum_after += dbg_anchor_sum(ha[i]);
		if (sum_after != sum_before) fprintf(stderr, "[sort] batch of %u queries: NOT A PERMUTATION of the emitted anchors (sums %016llx / %016llx)\n", nqb, (unsigned long long)sum_before, (unsigned long long)sum_after);
		u64 unsorted = 0, ties = 0, unmarked = 0;
		for (u32 q = 0; q < nqb; ++q)
			for (u64 i = h_off[q] + 1; i < h_off[q + 1]; ++i) {
				if (ha[i].x < ha[i - 1].x) ++unsorted;
				if (ha[i].x == ha[i - 1].x) { ++ties; if (!(ha[i].y & LQ_TIE_MARK) || !(ha[i - 1].y & LQ_TIE_MARK)) { if (unmarked++ < 4) fprintf(stderr, "[sort] query %u of the batch: equal x %016llx, y %016llx / %016llx\n", q, (unsigned long long)ha[i].x, (unsigned long long)ha[i - 1].y, (unsigned long long)ha[i].y); } }
			}
		fprintf(stderr, "[sort] batch of %u queries: %llu anchors, %llu out of order, %llu equal-x neighbours, %llu of them with an unmarked anchor\n", nqb, (unsigned long long)nA, (unsigned long long)unsorted, (unsigned long long)ties, (unsigned long long)unmarked);
		if (sum_after != sum_before || unsorted || unmarked) throw std::logic_error("LQCOV_DEBUG_SORT: the sorted anchors are not a permutation of the emitted ones, not ascending, or hold an unmarked tie (see stderr)");
	}
}

// work space of one batch of nA anchors on lane L
void lqcov_handle::batch_buffers(MapLane &L, u64 nA)
{
	if (nA > 0xfffffff0ULL) throw std::domain_error("a single query produces more than 2^32 anchors against this part");
	const u64 nA4 = ((nA + 1) * 4 + 15) & ~(u64)15;
	L.A.ensure((nA + 1) * 16); L.B.ensure(std::max<u64>((nA + 1) * 16, nA4 * 4));   // (B: the originals, then the chain's four 4-byte arrays of nA4 bytes each)
	// 20 B per anchor of scratch: the X / Y position lists of the two-bucket passes (8 B) and the second record array (8 B, in
	// the place of the chain's u[]) during the sort.  The chain's f/p/t/v (16 B per anchor) reuse B, which is dead once sorted.
	L.scr.ensure(nA4 * 5 + 64);
}

// The sorted anchors of a batch (in L.A; per-query offsets aqb, absolute, the batch starts at a_base): the (strand, rid) runs long
// enough to hold a chain, mm_chain_dp + mm_gen_regs + lq_cnt_match on them (chain.c, hit.c, esterr.c).  tie_mode: CovState.
void lqcov_handle::chain_stage(MapLane &L, Part &pt, const u64 *aqb, u64 a_base, u32 nqb, u32 q0, const u32 *qmap, u64 nA, int tie_mode, u32 n_want, u32 ivl_cap, bool dbg, const SatSink *sink)
{
	mm128 *dA = L.A.as<mm128>();
	const u64 nA4 = ((nA + 1) * 4 + 15) & ~(u64)15;
	u64 *d_cu = (u64*)((u8*)L.scr.p + 3 * nA4);
	i32 *d_cf = (i32*)L.B.p, *d_cp = (i32*)((u8*)L.B.p + nA4), *d_ct = (i32*)((u8*)L.B.p + 2 * nA4), *d_cv = (i32*)((u8*)L.B.p + 3 * nA4);
	// ---- (strand, rid) runs long enough to hold a chain ----
	u64 n_groups = 0;
	if (tie_mode == 2 && a_base == 0 && qmap) {                  // the second pass: the listed runs only, from their keys
		L.gstart.ensure(((u64)n_want + 1) * 8);
		if (n_want) {
			StageTimer t(this, L.stream, "k_run_list", (u64)n_want * 16);
			LQ_LAUNCH(k_want_runs, nblk(n_want, 256), 256, L.stream, dA, aqb, nqb, qmap, L.want.as<unsigned long long>(), n_want, L.gstart.as<u64>()); check_launch();
		}
		n_groups = n_want;
	} else {
		const u32 n_tiles = (u32)((nA + LQ_RUN_TILE - 1) / LQ_RUN_TILE);
		const u32 n_min = run_n_min();
		L.run_tiles.ensure(16);
		L.gstart.ensure((nA / n_min + 1) * 8);
		dzero(L.run_tiles.p, 4, L.stream);
		{
			StageTimer t(this, L.stream, "k_run_list", nA * 16);
			LQ_LAUNCH(k_run_list, std::min<u32>(n_tiles, K.run_grid), LQ_RUN_THREADS, L.stream, dA, nA, aqb, a_base, nqb, n_tiles, n_min, std::max<u32>(LQ_RUN_THREADS, std::min<u32>(K.run_stage, LQ_RUN_STAGE)), L.run_tiles.as<u32>(), L.gstart.as<u64>()); check_launch();
		}
		u32 ng = 0;
		d2h(&ng, L.run_tiles.as<u32>(), 1, L.stream);
		n_groups = ng;
	}
	// ---- chain + coverage ----
	ChainBufs cb; cb.f = d_cf; cb.p = d_cp; cb.t = d_ct; cb.v = d_cv; cb.u = d_cu;
	CovState cs;
	cs.lambda = lambda.as<unsigned long long>(); cs.lambda2 = lambda2.as<unsigned long long>();
	cs.cnts = cnts.as<u32>(); cs.qflags = qflags.as<u32>(); cs.skip = skip.as<u32>(); cs.qmoff = cnt_off_dev();
	cs.mini_pos = mini_pos.as<u64>(); cs.mpq_off = mpq_off.as<u64>(); cs.qlen = q.d_len.as<u32>(); cs.tlen = pt.rs.d_len.as<u32>();
	cs.ivl = L.ivl.as<Ivl>(); cs.n_ivl = L.n_ivl.as<u32>(); cs.ivl_cap = ivl_cap;
	cs.dbg = dbg ? dbg_chains.as<ChainRec>() : nullptr; cs.n_dbg = n_dbg.as<unsigned long long>(); cs.dbg_cap = dbg_cap;
	cs.tie_mode = tie_mode; cs.qmap = qmap;
	cs.cnt_max = cnt_max;
	cs.rec = sink ? sink->rec : nullptr; cs.n_rec = sink ? sink->n_rec : nullptr; cs.rec_cap = sink ? sink->rec_cap : 0;
	cs.rec_at = sink ? sink->at : nullptr; cs.n_at = sink ? sink->n_at : nullptr; cs.at_cap = sink ? sink->at_cap : 0;
	cs.sens = nullptr; cs.n_sens = L.n_sens.as<u32>(); cs.sens_cap = 0; cs.want = L.want.as<unsigned long long>(); cs.n_want = n_want;
	if (tie_mode == 1) {                                        // (every listed run can end up in the list)
		L.sens.ensure((n_groups + 1) * 8);
		cs.sens = L.sens.as<unsigned long long>(); cs.sens_cap = (u32)std::min<u64>(n_groups, 0xfffffff0ULL);
	}
	if (n_groups > 0xfffffff0ULL) throw std::domain_error("too many anchor runs in one batch");
	const int cap = K.chain_cap <= 64 ? 64 : K.chain_cap <= 128 ? 128 : 256;   // anchors of LDS per wave in k_chain
	// runs of >= wave_min anchors take the cooperative kernel; a run must fit k_chain's LDS budget on its own
	// (the second pass chains a few thousand runs, most of them true overlaps of 20-47 anchors: a wave each -- one thread per run
	// takes them in many rounds of the LDS budget with a handful of lanes busy, 28 ms of a lane's second pass at configs[2])
	const int wave_min = tie_mode == 2 && K.chain_wave_min <= 0 ? std::min<int>(LQ_CHAIN_WAVE_MIN, std::max<int>(P.min_cnt, 8)) : std::min(cap + 1, K.chain_wave_min > 0 ? K.chain_wave_min : LQ_CHAIN_WAVE_MIN);
	if (n_groups) {	// one thread per run, in array order, DP state of a wave's runs packed into LDS (in rounds if they exceed the budget).
		// Measured alternatives that were not faster on MI355X: a compacted longest-first work list for all runs (409 vs 292 ms at
		// configs[1]), a dense list in array order of the runs of min_cnt..47 anchors (configs[2]: within the run-to-run spread),
		// private-array DP for short runs, and a fixed [16][64] LDS column per lane (54 KiB per wave: 160 + 40 ms vs 113 ms).
		// wave_min - 1 <= 47 < the smallest budget, so every run fits.
		StageTimer t(this, L.stream, "k_chain", nA * 16);
#define LQ_CHAIN_LAUNCH(CAP) LQ_LAUNCH((k_chain<CAP>), nblk(n_groups, 64), 64, L.stream, dA, L.gstart.as<u64>(), (const u32*)nullptr, (u32)n_groups, aqb, a_base, nqb, q0, avg_qspan.as<float>(), mp, cs, (i32)P.min_cnt, (i32)(wave_min - 1))
		if (cap == 64) { LQ_CHAIN_LAUNCH(64); } else if (cap == 128) { LQ_CHAIN_LAUNCH(128); } else { LQ_CHAIN_LAUNCH(256); }
#undef LQ_CHAIN_LAUNCH
		check_launch();
	}
	{	// long runs: one wave per run, longest first
		u32 n_sel = 0;
		const u32 n_tiles = (u32)((n_groups + LQ_RUN_TILE - 1) / LQ_RUN_TILE);
		L.sel_tiles.ensure(((u64)n_tiles + 2) * 4);
		LQ_LAUNCH(k_sel_count, std::min<u32>(std::max<u32>(n_tiles, 1), 1u << 16), LQ_RUN_THREADS, L.stream, L.gstart.as<u64>(), n_groups, (i32)wave_min, (i32)0x7fffffff, n_tiles, L.sel_tiles.as<u32>()); check_launch();
		LQ_LAUNCH(k_tile_scan, 1, LQ_TSCAN_THREADS, L.stream, L.sel_tiles.as<u32>(), n_tiles); check_launch();
		d2h(&n_sel, L.sel_tiles.as<u32>() + n_tiles, 1, L.stream);
		if (n_sel) {
			L.gsel.ensure((u64)n_sel * 4); L.gkey.ensure((u64)n_sel * 4); L.gsel2.ensure((u64)n_sel * 4); L.gkey2.ensure((u64)n_sel * 4);
			LQ_LAUNCH(k_sel_write, std::min<u32>(n_tiles, 1u << 16), LQ_RUN_THREADS, L.stream, L.gstart.as<u64>(), n_groups, (i32)wave_min, (i32)0x7fffffff, n_tiles, L.sel_tiles.as<u32>(), L.gsel.as<u32>(), L.gkey.as<u32>()); check_launch();
			L.prim.sort_pairs_u32_u32(L.gkey.as<u32>(), L.gkey2.as<u32>(), L.gsel.as<u32>(), L.gsel2.as<u32>(), n_sel);
			StageTimer t(this, L.stream, "k_chain_wave");
			LQ_LAUNCH(k_chain_wave, n_sel, 64, L.stream, dA, L.gstart.as<u64>(), L.gsel2.as<u32>(), n_sel, aqb, a_base, nqb, q0, avg_qspan.as<float>(), mp, cb, cs);
			check_launch();
		}
	}
}

// one batch of queries [q0, q1) against one part, on lane L: seeds -> sort -> chains -> per-query intervals.
//
// Two ways (Knobs::ties_klib):
//  * klib's order everywhere (the reference's own arrangement, lqmap.c:238): every seed hit is written, queries with repeated
//    (hash, strand) minimizers go through klib's passes (sort_batch);
//  * the default: klib's order only where it can be observed.  First pass: only the hits whose (strand, rid) can reach a chain
//    at all are written (k_seed_count's survivors, h_aqf / aqf_off: their per-query offsets), every query is sorted by the
//    parallel sort (equal x in no particular order), and the chain kernels list the runs in which the order of equal-x anchors
//    matters instead of chaining them (kernels_chain.hpp: two of them inside one scan's band, or both peaks of one score;
//    oracle: sort modes 2 / 3, tests/test_tie_order.py).  Second pass, for the queries that own a listed run: all their seed
//    hits in the reference's emission order, klib's passes, and only the listed runs are chained.  Runs never interact
//    (chain.c:47) and everything they feed commutes, so the split is exact.
// ---- queries with a counter at its maximum (esterr.c:130,136; sat_replay.hpp) ------------------------------------------------------
// Called at the end of map_part with the part's plan still in place.  A flagged query is chained once more against this part --
// all its anchors in klib's order, nothing accumulated, every kept chain recorded -- and the host replays the chains in the
// order lq_cnt_match met them.  The first time a query is flagged its counters before this part are what the device holds minus
// this part's increments (no counter had reached the maximum until then, so until then the 32-bit counts are the reference's).
void lqcov_handle::sat_replay_part(Part &pt, const std::vector<u64> &h_aq, const std::vector<u64> &h_qmoff)
{
	const u32 n_q = q.n;
	if (!n_q || distributed) return;                        // (counters summed over ranks cannot be replayed: finish() refuses those)
	std::vector<u32> hf(n_q), hs(n_q);
	d2h(hf.data(), qflags.as<u32>(), n_q, stream);
	bool any = false;
	for (u32 i = 0; i < n_q; ++i) any |= (hf[i] & 1u) != 0;
	if (!any) return;
	d2h(hs.data(), skip.as<u32>(), n_q, stream);
	for (u32 qi = 0; qi < n_q; ++qi) {
		if (!(hf[qi] & 1u) || hs[qi]) continue;               // (skip: esterr.c:87 returned before anything was counted)
		std::vector<SatRec> recs; std::vector<u32> at;
		if (!sat_chains(pt, qi, h_aq, h_qmoff, recs, at)) continue;
		MapLane &L = *lanes[0];
		u64 off[2];
		d2h(off, cnt_off_dev() + qi, 2, L.stream);
		const size_t nc = (size_t)(off[1] - off[0]);
		sat_check(recs, at, nc);
		auto it = sat_cnt.find(qi);
		if (it == sat_cnt.end()) {
			std::vector<u32> c(nc);
			d2h(c.data(), cnts.as<u32>() + off[0], nc, L.stream);
			bool ok = true;
			for (const SatRec &r : recs) if (r.good) {
				ok &= c[(size_t)r.sti] > 0; --c[(size_t)r.sti];
				for (u32 k = 0; k < r.n_at; ++k) { u32 &o = c[at[r.at_off + k]]; ok &= o > 0; --o; }
			}
			for (u32 v : c) ok &= v < cnt_max;
			if (!ok) throw std::runtime_error("replay of saturated counters: the recorded chains of query " + q.names[qi] + " do not add up to its counters");
			it = sat_cnt.emplace(qi, std::move(c)).first;
		}
		const std::vector<u32> order = satreplay::regs_order(recs, satreplay::query_hash(q.names[qi], (i32)q.h_len[qi], 11 /* map.c:15 */));
		satreplay::replay(it->second, recs, at, order, cnt_max);
	}
}

// every kept chain of query qi against this part, recorded (the part's plan in place, the lanes made: after map_part); false: no anchors
bool lqcov_handle::sat_chains(Part &pt, u32 qi, const std::vector<u64> &h_aq, const std::vector<u64> &h_qmoff, std::vector<SatRec> &recs, std::vector<u32> &at)
{
	recs.clear(); at.clear();
	if (lanes.empty()) throw std::logic_error("the part has not been mapped yet");
	MapLane &L = *lanes[0];
	sat_n.ensure(16);
	const u64 len = h_aq[qi + 1] - h_aq[qi];
	if (!len) return false;
	if (len > (1ULL << 31) - 4096) throw std::domain_error("query " + q.names[qi] + ": too many anchors against one index part for the replay of its saturated counters");
	const u64 rec_cap = len / (u64)std::max<i32>(P.min_cnt, 1) + 1, at_cap = len;
	sat_rec.ensure(rec_cap * sizeof(SatRec)); sat_at.ensure(at_cap * 4 + 4);
	dzero(sat_n.p, 16, L.stream);
	const SatSink sink{sat_rec.as<SatRec>(), sat_n.as<unsigned long long>(), rec_cap, sat_at.as<u32>(), sat_n.as<unsigned long long>() + 1, at_cap};
	L.n_segs.ensure(64); L.n_ivl.ensure(4); L.n_sens.ensure(32); L.want.ensure(8); L.ivl.ensure(sizeof(Ivl));
	map_subset(L, pt, std::vector<u32>{qi}, std::vector<u32>{len > LQ_RS_MIN ? 1u : 0u}, std::vector<u64>{0, len}, h_qmoff[qi + 1] - h_qmoff[qi], 0, 0, 0, false, &sink);
	unsigned long long nn[2] = {0, 0};
	d2h(nn, sat_n.as<unsigned long long>(), 2, L.stream);
	if (nn[0] > rec_cap || nn[1] > at_cap) throw std::runtime_error("replay of saturated counters: record pool overflow");
	recs.resize(nn[0]); at.resize(nn[1]);
	d2h(recs.data(), sat_rec.as<SatRec>(), nn[0], L.stream); d2h(at.data(), sat_at.as<u32>(), nn[1], L.stream);
	stat_sat_chains += nn[0];
	return true;
}

void lqcov_handle::sat_check(const std::vector<SatRec> &recs, const std::vector<u32> &at, size_t nc)
{
	for (const SatRec &c : recs) if (c.good && (c.sti < 0 || (size_t)c.sti >= nc || c.at_off + c.n_at > at.size())) throw std::runtime_error("replay of saturated counters: inconsistent chain record");
	for (u32 v : at) if ((size_t)v >= nc) throw std::runtime_error("replay of saturated counters: inconsistent chain record");
}

// ---- the same for index parts spread over ranks (multigpu.PartRunner): the rank that mapped a part records a flagged query's chains
// against it (lqcov_part_sat_records), every rank replays the gathered records part by part on the counters the query had before the
// first part of the round (lqcov_sat_replay: host arithmetic only) and hands the result to finish() (lqcov_accum_set_replayed)
void lqcov_handle::part_sat_records(Part &pt, u32 qi, std::vector<SatRec> &recs, std::vector<u32> &at)
{
	if (qi >= q.n) throw std::invalid_argument("no such query");
	if (!pt.built) throw std::logic_error("part not built");
	if (!pt.plan.valid || pt.plan.mid_occ != mid_occ || pt.plan.n_q != q.n || pt.plan.n_qm != q.n_mini || (pt.plan.bucketed && pt.plan.q_begin != 0)) plan_part(pt, stream, prim);   // (as map_part does)
	swap_plan(pt.plan);
	struct PlanGuard { lqcov_handle *h; SeedPlan &S; ~PlanGuard() { h->swap_plan(S); } } plan_guard{this, pt.plan};
	sat_chains(pt, qi, pt.plan.h_aq, pt.plan.h_qmoff, recs, at);
	LQ_HIP_CHECK(hipStreamSynchronize(lanes[0]->stream));
}

void lqcov_handle::sat_replay_host(u32 qi, const SatRec *recs, u64 n_recs, const u32 *at, u64 n_at, u32 *counters, u64 n_counters)
{
	if (qi >= q.n) throw std::invalid_argument("no such query");
	std::vector<SatRec> r(recs, recs + n_recs); std::vector<u32> a(at, at + n_at), c(counters, counters + n_counters);
	sat_check(r, a, (size_t)n_counters);
	const std::vector<u32> order = satreplay::regs_order(r, satreplay::query_hash(q.names[qi], (i32)q.h_len[qi], 11 /* map.c:15 */));
	satreplay::replay(c, r, a, order, cnt_max);
	std::copy(c.begin(), c.end(), counters);
}

// A subset of the queries through klib's passes and the chain kernels: every anchor of the listed queries (nothing is filtered
// here: klib's order depends on the whole array), in the order the reference's sort leaves them.  tie_mode 2: only the runs in
// L.want are chained (the second pass of map_batch); 0 with a sink: every run, the chains recorded for the replay of a saturated
// query (sat_replay_part).  sq: the queries, sk: which of them go through klib's passes at all, so: their anchor offsets.
void lqcov_handle::map_subset(MapLane &L, Part &pt, const std::vector<u32> &sq, const std::vector<u32> &sk, const std::vector<u64> &so, u64 max_mini,
                              int tie_mode, u32 n_want, u32 ivl_cap, bool dbg, const SatSink *sink)
{
	const AvaView ava{P.ava ? pt.t_rank.as<u32>() : nullptr, P.ava ? pt.q_lo.as<u32>() : nullptr};
	const u32 ns = (u32)sq.size();
	const u64 nA2 = so.back();
	batch_buffers(L, nA2);
	L.sub_q.ensure(ns * 4 + 4); L.sub_off.ensure((ns + 1) * 8); L.sub_klib.ensure(ns * 4 + 4);
	h2d(L.sub_q.as<u32>(), sq.data(), ns, L.stream); h2d(L.sub_off.as<u64>(), so.data(), ns + 1, L.stream); h2d(L.sub_klib.as<u32>(), sk.data(), ns, L.stream);
	LQ_HIP_CHECK(hipStreamSynchronize(L.stream));         // (the host vectors may die with the caller's turn of its loop)
	{
		StageTimer t(this, L.stream, "k_seed_emit", nA2 * 24);
		LQ_LAUNCH(k_seed_emit, dim3(nblk(max_mini, LQ_EMIT_THREADS), ns), LQ_EMIT_THREADS, L.stream, q.mx.as<u64>(), q.my.as<u64>(), q_owner.as<u32>(), q.moff.as<u64>(), (u64)0, (u64)0,
		          pt.pos.as<u64>(), hit_start.as<u64>(), hit_n.as<u32>(), keep.as<u32>(), dup.as<u32>(),
		          a_off.as<u64>(), (u64)0, mp_off.as<u64>(), q.d_len.as<u32>(),
		          (int)P.no_self, pt.self_off.as<u32>(), pt.self_rid.as<u32>(), ava,
		          (const u32*)nullptr, L.A.as<mm128>(), L.B.as<mm128>(), mini_pos.as<u64>(), EmitSub{L.sub_q.as<u32>(), L.sub_off.as<u64>(), L.sub_klib.as<u32>()});
		check_launch();
	}
	if (nA2) {
		// (second pass: only the listed runs are chained -- klib's levels drop every bucket without one, k_rs_children)
		L.prune = tie_mode == 2 && !sink; L.prune_n_want = n_want; L.prune_n_sub = ns;
		struct PruneGuard { MapLane &L; ~PruneGuard() { L.prune = false; } } prune_guard{L};
		sort_checked(L, pt, L.sub_off.as<u64>(), L.sub_klib.as<u32>(), ns, 0, nA2, so, sk);
		if (lq_timeline) { LQ_HIP_CHECK(hipStreamSynchronize(L.stream)); int lane_id = 0; for (size_t i_ = 0; i_ < lanes.size(); ++i_) if (lanes[i_].get() == &L) lane_id = (int)i_; lq_tl("lane", lane_id, "  second pass sorted"); }
		chain_stage(L, pt, L.sub_off.as<u64>(), 0, ns, 0, L.sub_q.as<u32>(), nA2, tie_mode, n_want, ivl_cap, dbg, sink);
	}
}

void lqcov_handle::map_batch(MapLane &L, Part &pt, u32 q0, u32 q1, const std::vector<u64> &h_aq, const std::vector<u64> &h_aqf, const std::vector<u64> &h_qmoff, bool dbg)
{
	// (a batch starts with an empty arena: what the lane's last batch cut from it is dead -- its kernels are ahead of this batch's on
	// the lane's stream, its side streams were joined)
	L.drop_arena_buffers();
	lq_arena = L.arena.base ? &L.arena : nullptr;
	struct ArenaGuard { ~ArenaGuard() { lq_arena = nullptr; } } arena_guard;
	const u32 n_q = q.n;
	int lane_id = 0; for (size_t i_ = 0; i_ < lanes.size(); ++i_) if (lanes[i_].get() == &L) lane_id = (int)i_;
	lq_tl("lane", lane_id, "batch begins, queries", (double)(q1 - q0));
	L.gate_passed = false;
	struct GateGuard { lqcov_handle *h; MapLane &L; ~GateGuard() { if (!L.gate_passed) { L.gate_passed = true; h->open_gate(); } } } gate_guard{this, L};
	L.n_segs.ensure(64); L.n_ivl.ensure(4); L.n_sens.ensure(32); L.want.ensure(8);
	const bool opt = !K.ties_klib;
	const std::vector<u64> &h_off = opt ? h_aqf : h_aq;
	const u64 a_base = h_off[q0], nA = h_off[q1] - a_base;
	const u32 nqb = q1 - q0;
	const u64 j0 = h_qmoff[q0], nj = h_qmoff[q1] - j0;
	batch_buffers(L, nA);
	const AvaView ava{P.ava ? pt.t_rank.as<u32>() : nullptr, P.ava ? pt.q_lo.as<u32>() : nullptr};
	if (nj && !opt) {
		StageTimer t(this, L.stream, "k_seed_emit", nj * 32 + nA * 24);
		LQ_LAUNCH(k_seed_emit, nblk(nj, LQ_EMIT_THREADS), LQ_EMIT_THREADS, L.stream, q.mx.as<u64>(), q.my.as<u64>(), q_owner.as<u32>(), q.moff.as<u64>(), j0, nj,
		          pt.pos.as<u64>(), hit_start.as<u64>(), hit_n.as<u32>(), keep.as<u32>(), dup.as<u32>(),
		          a_off.as<u64>(), a_base, mp_off.as<u64>(), q.d_len.as<u32>(),
		          (int)P.no_self, pt.self_off.as<u32>(), pt.self_rid.as<u32>(), ava,
		          qklib.as<u32>(), L.A.as<mm128>(), L.B.as<mm128>(), mini_pos.as<u64>(), EmitSub{nullptr, nullptr, nullptr});
		check_launch();
	}
	if (nj && opt && pt.plan.bucketed) {
		if (nA) {                                                 // the survivors of the batch's queries (the part's seed plan holds them as records)
			StageTimer t(this, L.stream, "k_seed_emit_s", nA * 24);
			LQ_LAUNCH(k_seed_emit_s, nblk(nA, 256), 256, L.stream, L.use_surv ? L.use_surv : surv.as<u64>(), a_base, nA, L.use_aqf ? L.use_aqf : aqf_off.as<u64>(), q0, q1, SeedBits{pt.plan.rec_jb, pt.plan.rec_db},
			          q.mx.as<u64>(), q.my.as<u64>(), q.moff.as<u64>(), q.d_len.as<u32>(), dup.as<u32>(), L.A.as<mm128>());
			check_launch();
		}
	} else if (nj && opt) {                                       // no filter: every hit, nobody through klib's passes (h_aqf == h_aq)
		StageTimer t(this, L.stream, "k_seed_emit", nj * 32 + nA * 24);
		LQ_LAUNCH(k_seed_emit, nblk(nj, LQ_EMIT_THREADS), LQ_EMIT_THREADS, L.stream, q.mx.as<u64>(), q.my.as<u64>(), q_owner.as<u32>(), q.moff.as<u64>(), j0, nj,
		          pt.pos.as<u64>(), hit_start.as<u64>(), hit_n.as<u32>(), keep.as<u32>(), dup.as<u32>(),
		          a_off.as<u64>(), a_base, mp_off.as<u64>(), q.d_len.as<u32>(),
		          (int)P.no_self, pt.self_off.as<u32>(), pt.self_rid.as<u32>(), ava,
		          qzero.as<u32>(), L.A.as<mm128>(), L.B.as<mm128>(), mini_pos.as<u64>(), EmitSub{nullptr, nullptr, nullptr});
		check_launch();
	}
	const u32 ivl_cap = (u32)std::min<u64>(nA / (P.min_cnt > 0 ? P.min_cnt : 1) + 16, 0xfffffff0ULL);
	L.ivl.ensure((u64)ivl_cap * sizeof(Ivl));
	dzero(L.n_ivl.p, 4, L.stream); dzero(L.n_sens.p, 32, L.stream);
	if (nA) {
		const u64 *aqb = (opt ? (L.use_aqf ? L.use_aqf : aqf_off.as<u64>()) : aq_off.as<u64>()) + q0;          // batch view of the per-query anchor offsets
		const u32 *qkb = opt ? qzero.as<u32>() : qklib.as<u32>() + q0;               // (first pass: nobody goes through klib's passes)
		std::vector<u64> rel; std::vector<u32> hk;
		if (K.debug_sort) {
			rel.resize(nqb + 1); hk.assign(nqb, 0);
			for (u32 i = 0; i <= nqb; ++i) rel[i] = h_off[q0 + i] - a_base;
			if (!opt) d2h(hk.data(), qklib.as<u32>() + q0, nqb, L.stream);
		}
		sort_checked(L, pt, aqb, qkb, nqb, a_base, nA, rel, hk);
		if (opt && !L.gate_passed) { L.gate_passed = true; open_gate(); }          // (no walks in the first pass: the next lane may start under this batch's chains)
		chain_stage(L, pt, aqb, a_base, nqb, q0, nullptr, nA, opt ? 1 : 0, 0, ivl_cap, dbg);
	}
	u32 n_sens = 0;
	if (nA && opt) {
		u32 ns8[8] = {0};                                         // [0] listed runs, [1..6] by reason (lq_tie_list)
		d2h(ns8, L.n_sens.as<u32>(), 8, L.stream);
		n_sens = ns8[0];
		for (int i = 1; i <= 6; ++i) stat_tie_why[i - 1] += ns8[i];
	}
	lq_tl("lane", lane_id, "first pass done, anchors", (double)nA);
	if (n_sens) {
		// ---- second pass: the queries that own a run in which klib's order can be observed ----
		std::vector<u64> want(n_sens);
		d2h(want.data(), L.sens.as<u64>(), n_sens, L.stream);
		std::sort(want.begin(), want.end());
		want.erase(std::unique(want.begin(), want.end()), want.end());
		std::vector<u32> fq;
		for (u64 k : want) if (fq.empty() || fq.back() != (u32)(k >> 32)) fq.push_back((u32)(k >> 32));
		L.want.ensure(want.size() * 8);
		h2d(L.want.as<u64>(), want.data(), want.size(), L.stream);
		LQ_HIP_CHECK(hipStreamSynchronize(L.stream));
		stat_sens_runs += want.size(); stat_p2_queries += fq.size();
		u64 max_mini = 0;
		for (size_t i = 0; i < fq.size(); ) {
			std::vector<u32> sq, sk; std::vector<u64> so{0};
			size_t k = i;
			while (k < fq.size() && (k == i || so.back() + (h_aq[fq[k] + 1] - h_aq[fq[k]]) <= anchor_budget)) {
				const u64 len = h_aq[fq[k] + 1] - h_aq[fq[k]];
				sq.push_back(fq[k]); sk.push_back(len > LQ_RS_MIN ? 1u : 0u); so.push_back(so.back() + len);
				max_mini = std::max<u64>(max_mini, h_qmoff[fq[k] + 1] - h_qmoff[fq[k]]);
				++k;
			}
			stat_p2_anchors += so.back();
			lq_tl("lane", lane_id, "second pass begins, anchors", (double)so.back());
			map_subset(L, pt, sq, sk, so, max_mini, 2, (u32)want.size(), ivl_cap, dbg, nullptr);
			if (lq_timeline) { LQ_HIP_CHECK(hipStreamSynchronize(L.stream)); lq_tl("lane", lane_id, "second pass done"); }
			i = k;
		}
	}
	if (nA) {
		// ---- filter_redundant_coords per query (lqmap.c:287) ----
		u32 ni = 0;
		d2h(&ni, L.n_ivl.as<u32>(), 1, L.stream);
		if (ni > ivl_cap) throw std::runtime_error("interval pool overflow");
		if (ni) {
			L.iv_q.ensure((u64)ni * 4); L.iv_q2.ensure((u64)ni * 4); L.iv_se.ensure((u64)ni * 8); L.iv_se2.ensure((u64)ni * 8);
			L.ivq_off.ensure((n_q + 1) * 4); L.iv_scratch.ensure((u64)ni * 16);
			LQ_LAUNCH(k_split_ivl, nblk(ni, 256), 256, L.stream, L.ivl.as<Ivl>(), ni, L.iv_q.as<u32>(), L.iv_se.as<u64>()); check_launch();
			L.prim.sort_pairs_u32_u64(L.iv_q.as<u32>(), L.iv_q2.as<u32>(), L.iv_se.as<u64>(), L.iv_se2.as<u64>(), ni, 32);
			LQ_LAUNCH(k_ivl_offsets, nblk(n_q + 1, 256), 256, L.stream, L.iv_q2.as<u32>(), ni, n_q, L.ivq_off.as<u32>()); check_launch();
			// pv (the persisted intervals of all parts) is shared by the lanes: reserve the worst case under the lock, and grow
			// it only with every lane's kernels drained
			std::lock_guard<std::mutex> guard(pv_mu);
			const u64 need = pv_reserved + 2 * (u64)ni;
			if (need > 0xfffffff0ULL) throw std::domain_error("too many persisted intervals");
			if (need > pv_cap) {
				LQ_HIP_CHECK(hipDeviceSynchronize());
				u32 npv = 0;
				d2h(&npv, n_pv.as<u32>(), 1, L.stream);
				grow_keep(pv, (u64)npv * sizeof(Ivl), need * sizeof(Ivl) * 2, L.stream);
				pv_cap = (u32)std::min<u64>(pv.cap / sizeof(Ivl), 0xfffffff0ULL);
			}
			pv_reserved = need;
			StageTimer t(this, L.stream, "k_filter_redundant");
			LQ_LAUNCH(k_filter_redundant, nblk(n_q, 64), 64, L.stream, L.iv_se2.as<u64>(), L.ivq_off.as<u32>(), n_q, (u32)P.min_coverage,
			          L.iv_scratch.as<u32>(), pv.as<Ivl>(), n_pv.as<u32>(), pv_cap);
			check_launch();
		}
	}
}

static PsLists ps_lists(MapLane &L, int set, u32 sh)
{
	PsWork &W = L.ps[set];
	PsLists Ls;
	Ls.big[0] = W.big[0].as<PSeg>(); Ls.big[1] = W.big[1].as<PSeg>(); Ls.fin_s = W.fin_s.as<PSeg>(); Ls.fin_b = W.fin_b.as<PSeg>();
	Ls.cnt = L.sort_cnt.as<u32>() + (set ? LQ_C_PS1 : LQ_C_PS0);
	Ls.cap_big = (u32)std::min<u64>(W.big[0].cap / sizeof(PSeg), 0xfffffff0ULL); Ls.cap_fin = (u32)std::min<u64>(W.fin_s.cap / sizeof(PSeg), 0xfffffff0ULL);
	Ls.fin_s_max = std::max<u32>(LQ_PS_FIN_SMALL >> sh, 2); Ls.fin_b_max = std::max<u32>(LQ_PS_FIN_BIG >> sh, 4); Ls.child_target = std::max<u32>(LQ_PS_CHILD >> sh, 2);
	return Ls;
}

// one partition pass over the big list in slot `cur` of a set (kernels_psort.hpp); the children go to the other slot and the finishing lists
static void ps_pass(lqcov_handle *h, MapLane &L, int set, hipStream_t s, u64 nA, const KeyMap &km, const PsData &pd, u32 cur)
{
	PsWork &W = L.ps[set];
	u32 *cnt = L.sort_cnt.as<u32>() + (set ? LQ_C_PS1 : LQ_C_PS0);
	const PsLists Ls = ps_lists(L, set, h->K.ps_shift);
	const u32 cap_cnt = (u32)std::min<u64>(W.gcnt.cap / 4, 0xfffffff0ULL);
	// grids: a few blocks per CU striding over the device-side lists.  Most passes of a batch find their list short or empty,
	// and a launch sized for the worst case still has every one of its blocks placed (LDS and wave slots included) to find that out
	const u32 g_tiles = (u32)std::min<u64>((nA + LQ_PS_TILE - 1) / LQ_PS_TILE + 1, h->K.ps_grid);
	const u32 g_segs = (u32)std::min<u64>(Ls.cap_big, 256);
	const u32 nxt = cur ^ 1;
	const u32 cap_tiles = (u32)std::min<u64>(W.tmap.cap / 4, 0xfffffff0ULL);
	LQ_LAUNCH(k_ps_plan, 1, 256, s, Ls.big[cur], cnt + (cur ? LQ_P_BIG1 : LQ_P_BIG0), W.plan.as<PPlan>(), cnt, cap_cnt, cap_tiles, Ls.child_target, cnt + (nxt ? LQ_P_BIG1 : LQ_P_BIG0)); check_launch();
	LQ_LAUNCH(k_ps_tilemap, g_segs, 256, s, W.plan.as<PPlan>(), cnt + (cur ? LQ_P_BIG1 : LQ_P_BIG0), cnt, W.tmap.as<u32>(), cap_tiles, W.gcnt.as<u32>(), W.gdiff.as<unsigned long long>()); check_launch();
	{
		StageTimer t(h, s, "k_ps_hist");
		LQ_LAUNCH(k_ps_hist, g_tiles, LQ_PS_THREADS, s, Ls.big[cur], cnt + (cur ? LQ_P_BIG1 : LQ_P_BIG0), W.plan.as<PPlan>(), W.tmap.as<u32>(), cnt, pd, km, W.gcnt.as<u32>(), W.gdiff.as<unsigned long long>()); check_launch();
	}
	LQ_LAUNCH(k_ps_scan, g_segs, 256, s, Ls.big[cur], cnt + (cur ? LQ_P_BIG1 : LQ_P_BIG0), W.plan.as<PPlan>(), W.gcnt.as<u32>(), W.gcur.as<u32>(), W.gdiff.as<unsigned long long>(), Ls, (u32)(nxt ? LQ_P_BIG1 : LQ_P_BIG0),
	          (unsigned long long*)(L.sort_cnt.as<u32>() + (set ? LQ_C_PART1 : LQ_C_PART0))); check_launch();
	{
		StageTimer t(h, s, "k_ps_scatter");
		LQ_LAUNCH(k_ps_scatter, g_tiles, LQ_PS_THREADS, s, Ls.big[cur], cnt + (cur ? LQ_P_BIG1 : LQ_P_BIG0), W.plan.as<PPlan>(), W.tmap.as<u32>(), cnt, pd, km, W.gcur.as<u32>()); check_launch();
	}
}

// the two finishing kernels over a set's finishing lists
static void ps_finish(lqcov_handle *h, MapLane &L, int set, hipStream_t s, u64 nA, const KeyMap &km, const PsData &pd)
{
	u32 *cnt = L.sort_cnt.as<u32>() + (set ? LQ_C_PS1 : LQ_C_PS0);
	const PsLists Ls = ps_lists(L, set, h->K.ps_shift);
	const bool k32 = km.pbits + km.rbits + 1 <= 32 + 8 && !h->K.ps_key64;          // the key bits below the sub-bucket digit fit 32 bits (both kernels' digits are >= 8 bits)
	{
		StageTimer t(h, s, "k_ps_finish<8192>");
		unsigned long long *tl = (unsigned long long*)(L.sort_cnt.as<u32>() + (set ? LQ_C_FINB1 : LQ_C_FINB0));
		const u32 g = (u32)std::min<u64>(std::min<u64>(Ls.cap_fin, nA / LQ_PS_FIN_SMALL + 64), h->K.ps_grid / 4);
		// 1024 threads (measured at configs[2], 4 lanes, round 2: 256-thread blocks 2.40 s per step, 512: 2.25, 1024: 2.1-2.2; round 3: 512 = 1024).
		// Beside the other lanes' kernels a launch of this kernel takes ~3x its time alone, most of it waiting: with the class
		// emptied (everything partitioned down to 1024) the empty launches still took 367 ms per step and the step was the same.
		if (!k32) LQ_LAUNCH((k_ps_finish<LQ_PS_FIN_BIG, 1024, 10, u64>), g, 1024, s, Ls.fin_b, cnt + LQ_P_FIN_B, pd, km, tl);
		else LQ_LAUNCH((k_ps_finish<LQ_PS_FIN_BIG, 1024, 10, u32>), g, 1024, s, Ls.fin_b, cnt + LQ_P_FIN_B, pd, km, tl);
		check_launch();
	}
	{
		StageTimer t(h, s, "k_ps_finish<1024>");
		unsigned long long *tl = (unsigned long long*)(L.sort_cnt.as<u32>() + (set ? LQ_C_FINS1 : LQ_C_FINS0));
		const u32 g = (u32)std::min<u64>(std::min<u64>(Ls.cap_fin, nA / 16 + 256), h->K.ps_grid * 2);
		if (k32) LQ_LAUNCH((k_ps_finish<LQ_PS_FIN_SMALL, 256, 8, u32>), g, 256, s, Ls.fin_s, cnt + LQ_P_FIN_S, pd, km, tl);
		else LQ_LAUNCH((k_ps_finish<LQ_PS_FIN_SMALL, 256, 8, u64>), g, 256, s, Ls.fin_s, cnt + LQ_P_FIN_S, pd, km, tl);
		check_launch();
	}
}

// (K.ps_passes partition passes are issued without looking -- an even number: the big list ends in slot 0 --; psort_tail does the rest)

// The parallel sort of the segments of one list set (kernels_psort.hpp): partition passes while segments above the
// LDS capacity remain, then the two finishing kernels.  Everything is sized by upper bounds and strides over device-side
// counts: no host round trip here; psort_tail() looks at the counters once at the end of the batch's sort.
// Set 1 holds the buckets that left klib's passes: their anchors are still the originals in B, named by records.  Their
// finishing lists and the first pass of their big segments read B and write A; only then may a pass use B as its other
// buffer -- so set 1 finishes what is listed first, then runs the passes, then finishes the passes' children.
void lqcov_handle::psort_run(MapLane &L, int set, hipStream_t s, u64 nA, const KeyMap &km, const PsData &pd)
{
	u32 *cnt = L.sort_cnt.as<u32>() + (set ? LQ_C_PS1 : LQ_C_PS0);
	if (set == 1) {
		ps_finish(this, L, set, s, nA, km, pd);
		dzero(cnt + LQ_P_FIN_S, 8, s);                          // (LQ_P_FIN_S and LQ_P_FIN_B are neighbours)
	}
	for (u32 pass = 0; pass < K.ps_passes; ++pass) ps_pass(this, L, set, s, nA, km, pd, pass & 1);
	ps_finish(this, L, set, s, nA, km, pd);
}

// Segments that are still above the LDS capacity after the passes issued without looking (keys that agree in many leading bits take a
// pass per few bits): more passes, two at a time, with a look at the counter in between.  Rare; any input ends here sorted.
void lqcov_handle::psort_tail(MapLane &L, int set, hipStream_t s, u64 nA, const KeyMap &km, const PsData &pd)
{
	u32 *cnt = L.sort_cnt.as<u32>() + (set ? LQ_C_PS1 : LQ_C_PS0);
	for (int round = 0; round < 40; ++round) {                  // (a pass that moves anything uses up at least one key bit)
		u32 left = 0;
		d2h(&left, cnt + LQ_P_BIG0, 1, s);
		if (!left) return;
		dzero(cnt + LQ_P_FIN_S, 8, s);
		ps_pass(this, L, set, s, nA, km, pd, 0); ps_pass(this, L, set, s, nA, km, pd, 1);
		ps_finish(this, L, set, s, nA, km, pd);
		u32 ov = 0;
		d2h(&ov, cnt + LQ_P_OVERFLOW, 1, s);
		if (ov) throw std::runtime_error("parallel sort: list or counter space overflow");
	}
	throw std::logic_error("parallel sort: segments above the LDS capacity left after 84 passes");
}

// Sort every query's anchors by x, in klib's order wherever that order can be told apart (lqmap.c:238).
//   * queries without repeated (hash, strand) minimizers hold no equal x: the parallel sort, on the lane's second stream;
//   * the others (their anchors are in B, the originals) go through klib's passes byte by byte as 8-byte records
//     (kernels_sort.hpp, kernels_rsort.hpp); buckets of a pass that received fewer than two marked anchors leave for the
//     parallel sort as well (set 1, after the last pass), the others are written to A when they are finished.
void lqcov_handle::sort_batch(MapLane &L, Part &pt, const u64 *aqb, const u32 *qkb, u32 nqb, u64 a_base, u64 nA)
{
	mm128 *dA = L.A.as<mm128>(), *dB = L.B.as<mm128>();
	hipStream_t sD = L.stream, sC = L.stream2;
	if (nA >= 0x7ffffff0ULL) throw std::domain_error("more than 2^31 anchors in one query batch");
	// varying key bits of this part: x = strand:1 | rid:31 | position:32 (lqmap.c:190-196)
	u32 max_len = 0;
	for (u32 v : pt.rs.h_len) max_len = std::max(max_len, v);
	KeyMap km; km.pbits = 1; km.rbits = 0;
	while (km.pbits < 32 && ((u64)1 << km.pbits) < (u64)max_len) ++km.pbits;
	while (km.rbits < 31 && ((u64)1 << km.rbits) < (u64)pt.rs.n) ++km.rbits;
	u32 const_levels = 0;                                    // key bytes that are zero in every anchor of this part
	if (!K.no_level_skip) {
		if (pt.rs.n <= (1u << 16)) const_levels |= 1u << 6;
		if (pt.rs.n <= (1u << 8)) const_levels |= 1u << 5;
		if (max_len <= (1u << 24)) const_levels |= 1u << 3;
		if (max_len <= (1u << 16)) const_levels |= 1u << 2;
		if (max_len <= (1u << 8)) const_levels |= 1u << 1;
	}
	// list capacities: a segment in any list has more than 64 elements
	const u64 max_segs = nA / (LQ_RS_MIN + 1) + nqb + 1;
	const u64 cap_big = nA / ((LQ_PS_FIN_BIG >> K.ps_shift) + 1) + nqb + 16;
	L.sort_cnt.ensure(LQ_C_N * 4);
	for (int set = 0; set < 2; ++set) {
		PsWork &W = L.ps[set];
		W.big[0].ensure(cap_big * sizeof(PSeg)); W.big[1].ensure(cap_big * sizeof(PSeg)); W.plan.ensure((cap_big + 1) * sizeof(PPlan));
		// (a finishing list takes klib buckets and whole queries -- more than 64 elements each -- and the children of the partition
		// passes, at most 256 per segment and pass)
		const u64 cap_fin = max_segs + 512 * cap_big;
		W.fin_s.ensure(cap_fin * sizeof(PSeg)); W.fin_b.ensure(cap_fin * sizeof(PSeg));
		W.gcnt.ensure(cap_big * 256 * 4); W.gcur.ensure(cap_big * 256 * 4);
		W.gdiff.ensure(W.big[0].cap / sizeof(PSeg) * 8 + 8);   // (one word per entry the big list can hold)
		W.tmap.ensure((nA / LQ_PS_TILE + W.big[0].cap / sizeof(PSeg) + 2) * 4);   // (a segment's last tile may be partial)
	}
	L.segs0.ensure(max_segs * sizeof(SortSeg)); L.segs1.ensure(max_segs * sizeof(SortSeg));
	dzero(L.sort_cnt.p, LQ_C_N * 4, sD);
	auto lists = [&](int set) { return ps_lists(L, set, K.ps_shift); };
	u32 *cnt = L.sort_cnt.as<u32>();
	// records: R0 of its own, R1 in the part of the scratch area that is only used after the sort (map_batch)
	const u64 nA4 = ((nA + 1) * 4 + 15) & ~(u64)15;
	L.R0.ensure((nA + 1) * sizeof(RRec));
	RRec *R[2] = { L.R0.as<RRec>(), (RRec*)((u8*)L.scr.p + 3 * nA4) };
	PsData pd; pd.A = dA; pd.B = dB; pd.R[0] = R[0]; pd.R[1] = R[1];
	WalkCaps wcaps; wcaps.c[0] = 4096; wcaps.c[1] = 16384; wcaps.c[2] = 65536; wcaps.c[3] = 159744;
	for (int c = 0; c < 4; ++c) wcaps.c[c] >>= K.walk_shift;   // test knob
	{
		StageTimer t(this, sD, "k_sort_init");
		LQ_LAUNCH(k_sort_init, nblk(nqb, 64), 64, sD, aqb, a_base, nqb, qkb, dA, L.segs0.as<SortSeg>(), cnt, lists(0), km, wcaps);
		check_launch();
	}
	// the parallel sort of the clean queries runs beside klib's passes
	LQ_HIP_CHECK(hipEventRecord(L.ev_fork, sD));
	LQ_HIP_CHECK(hipStreamWaitEvent(sC, L.ev_fork, 0));
	psort_run(L, 0, sC, nA, km, pd);
	LQ_HIP_CHECK(hipEventRecord(L.ev_join, sC));
	// ---- klib's passes over the queries with repeated minimizers ----
	u32 hl[LQ_C_N];                                         // the level's counters as the host last saw them
	d2h(hl, cnt, LQ_C_N, sD);
	u32 ns = hl[LQ_C_KLIB0];
	if (ns) {
		u32 *hx = (u32*)L.scr.p, *py = (u32*)((u8*)L.scr.p + nA4);
		const u32 tile = K.sort_tile ? K.sort_tile : LQ_SORT_TILE;
		const u32 wgrid = K.walk_grid;
		L.sort_d.ensure(nA + 256); L.sort_dst.ensure((nA + 1) * 4);
		L.ck_n.ensure(16);
		SortSeg *cur = L.segs0.as<SortSeg>(), *nxt = L.segs1.as<SortSeg>();
		u32 cur_slot = LQ_C_KLIB0, nxt_slot = LQ_C_KLIB1;
		u32 shift = 56, rb = 0;                                // rb: the record array the level reads
		bool gated = false;
		for (int level = 0; level < 8 && ns > 0; ++level) {
			L.hist.ensure((u64)ns * 1024); L.begs.ensure((u64)ns * 1024); L.mhist.ensure((u64)ns * 1024);
			L.seg_info.ensure((u64)ns * sizeof(SegInfo)); L.walk_list.ensure((u64)ns * 4 * LQ_WALK_CLASSES); L.two_list.ensure((u64)ns * 4);
			const u32 g_seg = std::min<u32>(ns, 1u << 18);
			// tiles of the level's sub-arrays for the two streaming kernels
			const u64 max_tiles = nA / tile + ns + 1;
			const u32 g_tile = ((u32)std::min<u64>(max_tiles, K.tile_grid) + 7) & ~7u;   // a multiple of the XCD count (LQ_TILE_LOOP); blocks stride over the device-side tile list
			L.tile_list.ensure(max_tiles * sizeof(SortTile));
			// (the counters of the level -- next list, two-bucket and walk class lists, two-bucket tiles -- are zeroed by k_sort_tiles, the
			// tile counter by the k_rs_children of the level before or the batch's memset: no memset dispatches inside the level loop)
			LQ_LAUNCH(k_sort_tiles, std::min<u32>(ns / 256 + 1, 4096), 256, sD, cur, cnt + cur_slot, tile, L.tile_list.as<SortTile>(), cnt + LQ_C_TILES, L.hist.as<u32>(), L.mhist.as<u32>(),
			          cnt + nxt_slot, cnt + LQ_C_TWO, (u32)(1 + LQ_WALK_CLASSES), cnt + LQ_C_TWO_TILES, cnt + LQ_C_LEN0, (u32)LQ_WALK_CLASSES);
			check_launch();
			{
				StageTimer t(this, sD, level == 0 ? "k_rs_hist<first>" : "k_rs_hist");
				if (level == 0) LQ_LAUNCH((k_rs_hist<true>), g_tile, 256, sD, cur, L.tile_list.as<SortTile>(), cnt + LQ_C_TILES, tile, 1, dB, R[rb], L.sort_d.as<u8>(), L.hist.as<u32>(), L.mhist.as<u32>(), (unsigned long long*)(cnt + LQ_C_HIST0));
				else LQ_LAUNCH((k_rs_hist<false>), g_tile, 256, sD, cur, L.tile_list.as<SortTile>(), cnt + LQ_C_TILES, tile, 1, dB, R[rb], L.sort_d.as<u8>(), L.hist.as<u32>(), L.mhist.as<u32>(), (unsigned long long*)(cnt + LQ_C_HIST));
				check_launch();
			}
			LQ_LAUNCH(k_sort_classify, g_seg, LQ_CLASSIFY_THREADS, sD, cur, cnt + cur_slot, ns, L.hist.as<u32>(), L.begs.as<u32>(), L.seg_info.as<SegInfo>(),
			          L.walk_list.as<u32>(), L.two_list.as<u32>(), cnt + LQ_C_TWO, cnt + LQ_C_WALK0, wcaps);
			check_launch();
			{	// closed-form two-bucket passes (the strand bit at the top level), over tiles: count the X / Y elements of every tile, scan
				// per sub-array, position lists, then destinations + the move of the records
				StageTimer t(this, sD, "k_sort_two");
				L.two_tiles.ensure(max_tiles * sizeof(SortTile)); L.two_tile0.ensure((u64)ns * 4); L.two_tcnt.ensure(max_tiles * 8); L.two_m.ensure((u64)ns * 4);
				const SortTile *tt = L.two_tiles.as<SortTile>();
				const u32 *ntt = cnt + LQ_C_TWO_TILES;
				const u32 g_two = std::min<u32>(g_tile, 2048);         // grid-stride: a level without two-bucket sub-arrays costs near-empty launches
				const u64 *rc = (const u64*)R[rb]; u64 *rn = (u64*)R[rb ^ 1];
				LQ_LAUNCH(k_two_tiles, std::min<u32>(ns / 256 + 1, 4096), 256, sD, cur, L.two_list.as<u32>(), cnt + LQ_C_TWO, tile, L.two_tiles.as<SortTile>(), cnt + LQ_C_TWO_TILES, L.two_tile0.as<u32>()); check_launch();
				LQ_LAUNCH((k_sort_two_tiled<0>), g_two, 256, sD, cur, L.seg_info.as<SegInfo>(), tt, ntt, tile, L.sort_d.as<u8>(), L.two_tcnt.as<u32>(), L.two_m.as<u32>(), hx, py, rc, rn); check_launch();
				LQ_LAUNCH(k_sort_two_scan, std::min<u32>(ns, 16384), 64, sD, cur, L.two_list.as<u32>(), cnt + LQ_C_TWO, tile, L.two_tile0.as<u32>(), L.two_tcnt.as<u32>(), L.two_m.as<u32>()); check_launch();
				LQ_LAUNCH((k_sort_two_tiled<1>), g_two, 256, sD, cur, L.seg_info.as<SegInfo>(), tt, ntt, tile, L.sort_d.as<u8>(), L.two_tcnt.as<u32>(), L.two_m.as<u32>(), hx, py, rc, rn); check_launch();
				LQ_LAUNCH((k_sort_two_tiled<2>), g_two, 256, sD, cur, L.seg_info.as<SegInfo>(), tt, ntt, tile, L.sort_d.as<u8>(), L.two_tcnt.as<u32>(), L.two_m.as<u32>(), hx, py, rc, rn); check_launch();
			}
			{
				const u8 *dD = L.sort_d.as<u8>(); const u32 *dH = L.hist.as<u32>(), *dBg = L.begs.as<u32>(); u32 *dDst = L.sort_dst.as<u32>();
				const u32 *wl = L.walk_list.as<u32>();
				// two walker streams: the checkpointed walks with their solvers on one, the whole walks of the shorter size classes on the
				// other -- they concern different sub-arrays, and a level waits for the slower of the two, not for their sum
				hipStream_t sW = L.streamW, sW2 = L.streamW2;
				// the largest digit of this level decides how many register groups the long walker needs
				u32 max_digit = 255;
				if (shift == 48) max_digit = (pt.rs.n ? pt.rs.n - 1 : 0) >> 16;
				else if (shift == 40 && pt.rs.n <= (1u << 16)) max_digit = (pt.rs.n ? pt.rs.n - 1 : 0) >> 8;
				else if (shift == 32 && pt.rs.n <= (1u << 8)) max_digit = pt.rs.n ? pt.rs.n - 1 : 0;
				else if (shift == 24) max_digit = (max_len ? max_len - 1 : 0) >> 24;
				else if (shift == 16 && max_len <= (1u << 24)) max_digit = (max_len ? max_len - 1 : 0) >> 16;
				else if (shift == 8 && max_len <= (1u << 16)) max_digit = (max_len ? max_len - 1 : 0) >> 8;
				if (K.debug_sort) {
					u32 hc[LQ_C_N]; d2h(hc, cnt, LQ_C_N, sD);
					fprintf(stderr, "[sort] level %d shift %u max_digit %u ns %u two %u walk %u %u %u %u %u nA %llu\n", level, shift, max_digit, ns, hc[LQ_C_TWO],
					        hc[LQ_C_WALK0], hc[LQ_C_WALK1], hc[LQ_C_WALK2], hc[LQ_C_WALK3], hc[LQ_C_WALK4], (unsigned long long)nA);
					fflush(stderr);
				}
				// What the host knows of this level's sub-arrays: how many there are of every walk size class (counted by length when they
				// were made: an upper bound of the class lists, which hold the general passes only), and whether the level's byte can take
				// more than two values at all (byte 7 is strand << 7 | rid >> 24: two buckets below 2^24 targets -- no walk, ever).
				// Launches of empty classes are skipped, the others sized by the bound: a grid sized for the worst case has every block
				// placed, LDS included, only to find its list empty, and a level has up to ten such launches on its critical path.
				u32 lenc[LQ_WALK_CLASSES];
				for (int c = 0; c < LQ_WALK_CLASSES; ++c) lenc[c] = hl[LQ_C_LEN0 + c];
				const u32 max_buckets = shift == 56 ? 2 * (((pt.rs.n ? pt.rs.n - 1 : 0) >> 24) + 1) : max_digit + 1;
				const bool any_walk = max_buckets > 2 || K.no_level_skip;
				const bool ck_small = K.reg_walker && max_digit < LQ_CK_B;
				const bool ck3 = ck_small || K.ckpt3;                             // (round 6: on by default for passes with many buckets too -- their states are found by sub-chains side by side, a 65-160 k walk took 8-21 ms whole)
				const u64 n_ck_segs = K.ckpt ? (u64)(ck3 ? lenc[3] : 0) + lenc[4] : 0;
				int first_plain_class = K.ckpt ? (ck3 ? 2 : 3) : LQ_WALK_CLASSES - 1;
				bool w1 = false, w2 = false;                          // which walker streams got work
				// (the checkpoint buffers are sized before the fork: a block that the lane's stream allocates after the event the walker
				// streams wait for would not be ordered before their kernels)
				const u32 ck_unit = std::max<u32>((ck_small ? K.ck_unit : K.ck_unit_many) >> K.walk_shift, 8);
				const u32 ck_quantum = ck_small ? 1u : (u32)LQ_CKM_Q;   // many buckets: checkpoints come in sub-chains of LQ_CKM_Q (k_ck_chain256)
				const u32 ck_min_len = wcaps.c[ck3 ? 2 : 3] + 1;
				const u64 cks_max = std::min<u64>(std::min<u64>(nA / ck_min_len + 1, ns), n_ck_segs), ck_max = nA / ck_unit + (2 + ck_quantum) * cks_max, tiles_max = nA / LQ_CK_TILE + cks_max;
				const u32 per_ck = ck_small ? LQ_CK_B : 256;          // cursors per checkpoint
				if (any_walk && n_ck_segs) {
					L.ck_segs.ensure(cks_max * sizeof(CkSeg)); L.ck_S.ensure(ck_max * per_ck * 4); L.ck_slot.ensure(ck_max * 4 + 4);
					if (ck_small) { L.ck_T.ensure((tiles_max + 1) * LQ_CK_B * 4); L.ck_E.ensure(cks_max * LQ_CK_B * LQ_CK_B * 4); }
				}
				if (any_walk) { LQ_HIP_CHECK(hipEventRecord(L.ev_w0, sD)); }
				if (!gated && !L.gate_passed) { L.gate_passed = true; open_gate(); gated = true; }   // let the next lane start under these walks
				// Long sub-arrays: the walk's state at evenly spread checkpoints is computed without walking (kernels_ckpt.hpp) and
				// one walker per checkpoint runs a short piece.  Few buckets (the byte of rid above 65536 targets: the (query, strand)
				// arrays of the longest queries, millions of anchors each): states from prefix counts; up to 256 buckets: bulk-follow
				// solver, longest size class only (finding the states costs about as much as walking 50-100 k elements there).
				// The plan (which sub-arrays, their tiles and checkpoints) is laid out on the device from the class lists.
				if (any_walk && n_ck_segs) {
					LQ_HIP_CHECK(hipStreamWaitEvent(sW, L.ev_w0, 0)); w1 = true;
					const u32 unit = ck_unit;
					const CkSeg *dck = L.ck_segs.as<CkSeg>();
					const u32 *ckn = L.ck_n.as<u32>();
					LQ_LAUNCH(k_ck_plan, 1, 256, sW, cur, wl, ns, cnt + LQ_C_WALK0, (int)ck3, unit, ck_small ? 512u : 256u, ck_quantum, L.ck_segs.as<CkSeg>(), (u32)cks_max, L.ck_n.as<u32>()); check_launch();
					const u32 g_ck = (u32)std::min<u64>(ck_max, wgrid), g_cks = (u32)std::min<u64>(cks_max, wgrid);
					if (ck_small) {
						{
							StageTimer t(this, sW, "k_ck_prefix");
							LQ_LAUNCH(k_ck_tilehist, (u32)std::min<u64>(tiles_max, 1u << 16), 256, sW, dck, ckn, cur, dD, L.ck_T.as<u32>()); check_launch();
							LQ_LAUNCH(k_ck_tilescan, std::min<u32>(g_cks, 8192), 256, sW, dck, ckn, cur, L.ck_T.as<u32>()); check_launch();
						}
						{
							StageTimer t(this, sW, "k_ck_solve");
							LQ_LAUNCH(k_ck_phases, (u32)std::min<u64>(cks_max * LQ_CK_B, 1u << 16), 64, sW, dck, ckn, cur, dD, dH, dBg, L.ck_T.as<u32>(), L.ck_E.as<u32>()); check_launch();
							LQ_LAUNCH(k_ck_solve, g_ck, 64, sW, dck, ckn, cur, dD, dH, dBg, L.ck_T.as<u32>(), L.ck_E.as<u32>(), L.ck_S.as<u32>(), L.ck_slot.as<u32>()); check_launch();
						}
						{
							StageTimer t(this, sW, "k_sort_walk_reg<1>ck");
							LQ_LAUNCH((k_sort_walk_reg<1>), g_ck, 64, sW, cur, (const u32*)nullptr, ckn, dD, dH, dBg, dDst, dck, ckn, L.ck_S.as<u32>(), L.ck_slot.as<u32>()); check_launch();
						}
					} else {
						{
							StageTimer t(this, sW, "k_ck_chain256");
							LQ_LAUNCH(k_ck_chain256, (u32)std::min<u64>(ck_max / LQ_CKM_Q + 1, wgrid), LQ_CKM_THREADS, sW, dck, ckn, cur, dD, dH, dBg, L.ck_S.as<u32>(), L.ck_slot.as<u32>()); check_launch();
						}
						{
							StageTimer t(this, sW, "k_sort_walk_solo_ck");
							LQ_LAUNCH(k_sort_walk_solo, g_ck, 64, sW, cur, (const u32*)nullptr, ckn, dD, dH, dBg, dDst, dck, ckn, L.ck_S.as<u32>(), L.ck_slot.as<u32>()); check_launch();
						}
					}
				}
				// whole walks, longest class first: they outlast everything else of the level on a handful of CUs
				auto fork2 = [&]() { if (!w2) { LQ_HIP_CHECK(hipStreamWaitEvent(sW2, L.ev_w0, 0)); w2 = true; } };
				for (int c = first_plain_class; any_walk && c >= 2; --c) {
					if (!lenc[c]) continue;
					fork2();
					const u32 g = std::min<u32>(lenc[c], std::min<u32>(8192, wgrid));
					const CkSeg *nock = nullptr;
					if (K.reg_walker && max_digit < 64) { StageTimer t(this, sW2, "k_sort_walk_reg<1>"); LQ_LAUNCH((k_sort_walk_reg<1>), g, 64, sW2, cur, wl + (u64)c * ns, cnt + LQ_C_WALK0 + c, dD, dH, dBg, dDst, nock, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr); }
					else if (K.reg_walker && max_digit < 128) { StageTimer t(this, sW2, "k_sort_walk_reg<2>"); LQ_LAUNCH((k_sort_walk_reg<2>), g, 64, sW2, cur, wl + (u64)c * ns, cnt + LQ_C_WALK0 + c, dD, dH, dBg, dDst, nock, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr); }
					else { StageTimer t(this, sW2, "k_sort_walk_solo"); LQ_LAUNCH(k_sort_walk_solo, g, 64, sW2, cur, wl + (u64)c * ns, cnt + LQ_C_WALK0 + c, dD, dH, dBg, dDst, nock, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr); }
					check_launch();
				}
				if (any_walk && lenc[1]) { fork2(); StageTimer t(this, sW2, "k_sort_walk_lds<16384>"); LQ_LAUNCH((k_sort_walk_lds<16384>), std::min<u32>(lenc[1], std::min<u32>(8192, wgrid)), 64, sW2, cur, wl + (u64)1 * ns, cnt + LQ_C_WALK1, dD, dH, dBg, dDst); check_launch(); }
				if (any_walk && lenc[0]) { fork2(); StageTimer t(this, sW2, "k_sort_walk_lds<4096>"); LQ_LAUNCH((k_sort_walk_lds<4096>), std::min<u32>(lenc[0], std::min<u32>(1u << 16, wgrid * 4)), 64, sW2, cur, wl + (u64)0 * ns, cnt + LQ_C_WALK0, dD, dH, dBg, dDst); check_launch(); }
				if (w1) { LQ_HIP_CHECK(hipEventRecord(L.ev_w1, sW)); LQ_HIP_CHECK(hipStreamWaitEvent(sD, L.ev_w1, 0)); }
				if (w2) { LQ_HIP_CHECK(hipEventRecord(L.ev_w2, sW2)); LQ_HIP_CHECK(hipStreamWaitEvent(sD, L.ev_w2, 0)); }
			}
			{
				StageTimer t(this, sD, "k_rs_scatter");
				LQ_LAUNCH(k_rs_scatter, g_tile, 256, sD, cur, L.seg_info.as<SegInfo>(), L.tile_list.as<SortTile>(), cnt + LQ_C_TILES, tile, 1, R[rb], R[rb ^ 1], L.sort_dst.as<u32>(), (unsigned long long*)(cnt + LQ_C_SCATTERED));
				check_launch();
			}
			{
				StageTimer t(this, sD, "k_rs_children");
				LQ_LAUNCH(k_rs_children, (u32)std::min<u64>((u64)ns * 4, 1u << 20), LQ_CHILD_THREADS, sD, cur, cnt + cur_slot, R[rb ^ 1], rb ^ 1, dB, dA, L.hist.as<u32>(), L.mhist.as<u32>(), L.begs.as<u32>(),
				          nxt, cnt + nxt_slot, const_levels, lists(1), km, (int)K.all_klib, cnt + LQ_C_TILES, cnt + LQ_C_LEN0, wcaps,
			          L.prune && !K.debug_sort && K.prune ? PruneWant{L.want.as<unsigned long long>(), L.prune_n_want, L.sub_off.as<u64>(), L.sub_q.as<u32>(), L.prune_n_sub} : PruneWant{nullptr, 0, nullptr, nullptr, 0});
				check_launch();
			}
			d2h(hl, cnt, LQ_C_N, sD);
			if (lq_timeline) { int lane_id = 0; for (size_t i_ = 0; i_ < lanes.size(); ++i_) if (lanes[i_].get() == &L) lane_id = (int)i_; lq_tl("lane", lane_id, "  klib level done, shift", (double)shift); }
			ns = hl[nxt_slot];
			std::swap(cur, nxt); std::swap(cur_slot, nxt_slot);
			rb ^= 1;
			if (shift >= 8) { shift -= 8; while (shift > 0 && (const_levels >> (shift >> 3) & 1)) shift -= 8; }
		}
		psort_run(L, 1, sD, nA, km, pd);                      // the buckets that left klib's passes
	}
	LQ_HIP_CHECK(hipStreamWaitEvent(sD, L.ev_join, 0));
	{
		u32 hc[LQ_C_N];
		d2h(hc, cnt, LQ_C_N, sD);
		if (hc[LQ_C_PS0 + LQ_P_OVERFLOW] || hc[LQ_C_PS1 + LQ_P_OVERFLOW]) throw std::runtime_error("parallel sort: list or counter space overflow");
		if (profiling) {	// algorithmic bytes of the sort's stages from what the kernels really moved
			auto t64 = [&](int i) { return (u64)hc[i] | (u64)hc[i + 1] << 32; };
			add_stage_bytes("k_rs_hist<first>", t64(LQ_C_HIST0) * 25);          // anchor in, record + digit byte out
			add_stage_bytes("k_rs_hist", t64(LQ_C_HIST) * 9);                   // record in, digit byte out
			add_stage_bytes("k_rs_scatter", t64(LQ_C_SCATTERED) * 20);          // destination + record in, record out
			add_stage_bytes("k_ps_hist", (t64(LQ_C_PART0) + t64(LQ_C_PART1)) * 16);
			add_stage_bytes("k_ps_scatter", (t64(LQ_C_PART0) + t64(LQ_C_PART1)) * 32);
			add_stage_bytes("k_ps_finish<8192>", (t64(LQ_C_FINB0) + t64(LQ_C_FINB1)) * 32);
			add_stage_bytes("k_ps_finish<1024>", (t64(LQ_C_FINS0) + t64(LQ_C_FINS1)) * 32);
		}
		if (hc[LQ_C_PS0 + LQ_P_BIG0]) psort_tail(L, 0, sD, nA, km, pd);
		if (hc[LQ_C_PS1 + LQ_P_BIG0]) psort_tail(L, 1, sD, nA, km, pd);
	}
}

void lqcov_handle::open_gate()
{
	{ std::lock_guard<std::mutex> lk(gate_mu); ++gate_count; }
	gate_cv.notify_all();
}

// ---- the seed hits that can be part of a chain (kernels_seed.hpp) ------------------------------------------------------------
// Host side of the bucketed filter: the geometry (slices of targets per query, segments of minimizers, chunks of queries that
// fit the record buffer), then per chunk count -> scan -> scatter -> decide -> scan -> collect.  One host sync per chunk (the
// survivors' total sizes the plan's array).  The plan holds the survivors of a *group* of chunks -- the queries from q_begin on
// until LQCOV_SEED_SURV_MAX survivors are reached (ultra-long reads keep a third of tens of billions of hits): map_part maps a
// group's batches and asks for the next group.  h_aqf / aqf_off of the group's queries count from the group's first survivor.
bool lqcov_handle::seed_filter(Part &pt, hipStream_t s, Prim &pr, SeedWork &W, u32 n_min, u32 jb, u32 db, SeedJob &J)
{
	std::lock_guard<std::mutex> seed_lock(seed_mu);          // (the build thread plans the next part while this part's groups are made: one at a time)
	const u32 q_begin = J.q_begin;
	const std::vector<u64> &h_qmoff = *J.h_qmoff;
	const u32 n_q = q.n;
	const u64 n_qm = q.n_mini;
	const u32 n_targets = std::max<u32>(pt.rs.n, 1);
	const AvaView ava{P.ava ? pt.t_rank.as<u32>() : nullptr, P.ava ? pt.q_lo.as<u32>() : nullptr};
	const SeedBits bits{jb, db};
	// hits per kept minimizer, scanned (n_qm + 1 entries: the last one is the total); hits before every query
	W.hlen.ensure((n_qm + 1) * 4); W.h_off.ensure((n_qm + 1) * 8); W.hq_off.ensure((n_q + 1) * 8);
	LQ_LAUNCH(k_hit_len, nblk(n_qm, 256), 256, s, J.hit_n, J.keep, n_qm, W.hlen.as<u32>()); check_launch();
	dzero(W.hlen.as<u32>() + n_qm, 4, s);
	pr.exclusive_scan_u32_u64(W.hlen.as<u32>(), W.h_off.as<u64>(), n_qm + 1);
	LQ_LAUNCH(k_query_hoff, nblk(n_q + 1, 256), 256, s, q.moff.as<u64>(), W.h_off.as<u64>(), n_q, W.hq_off.as<u64>()); check_launch();
	std::vector<u64> h_hq(n_q + 1);
	d2h(h_hq.data(), W.hq_off.as<u64>(), n_q + 1, s);
	// geometry
	std::vector<SeedQ> qg(n_q);
	for (u32 i = 0; i < q_begin; ++i) memset(&qg[i], 0, sizeof(SeedQ));
	std::vector<SeedSeg> segs;
	std::vector<u32> has(n_q, 0);
	u64 max_hq = 0;
	for (u32 i = 0; i < n_q; ++i) max_hq = std::max(max_hq, h_hq[i + 1] - h_hq[i]);
	if (max_hq >= 0xfffffff0ULL) throw std::domain_error("more than 2^32 seed hits of one query against one index part");
	const u64 chunk_cap = std::max<u64>(K.seed_chunk, max_hq);
	struct Chunk { u32 q_lo, q_hi, g_lo, g_hi; u64 ne, nb, hits, h0; bool big; size_t bq_at; };
	std::vector<Chunk> chunks;
	std::vector<u32> bq_all;
	const u32 q_stop = std::min(J.q_stop, n_q);
	for (u32 i = q_begin; i < q_stop; ) {
		Chunk c; c.q_lo = i; c.g_lo = (u32)segs.size(); c.ne = 0; c.nb = 0; c.hits = 0; c.h0 = h_hq[i]; c.big = false; c.bq_at = bq_all.size();
		while (i < q_stop && (c.hits == 0 || c.hits + (h_hq[i + 1] - h_hq[i]) <= chunk_cap)) {
			const u64 hq = h_hq[i + 1] - h_hq[i], nm = h_qmoff[i + 1] - h_qmoff[i];
			SeedQ g; memset(&g, 0, sizeof(g));
			g.seg0 = (u32)segs.size(); g.cb = c.ne; g.bk = c.nb;
			if (hq) {
				has[i] = 1;
				u64 nsl = (hq + K.seed_bucket - 1) / K.seed_bucket;
				nsl = std::min<u64>(nsl, std::min<u64>(LQ_SD_SL_BIG, std::max<u32>(n_targets / 2, 1)));
				g.nsl = (u32)std::max<u64>(nsl, 1);
				g.mul = g.nsl == 1 ? 0u : (u32)((((u64)g.nsl) << 32) / n_targets);       // slice of rid = rid * mul >> 32 < nsl for every rid < n_targets
				g.nseg = (u32)((nm + K.seed_segl - 1) / K.seed_segl);
				for (u32 p = 0; p < g.nseg; ++p) {
					SeedSeg sg; memset(&sg, 0, sizeof(sg));
					sg.j0 = h_qmoff[i] + (u64)p * K.seed_segl; sg.j1 = std::min<u64>(sg.j0 + K.seed_segl, h_qmoff[i + 1]); sg.q = i; sg.ord = p;
					segs.push_back(sg);
				}
				if (g.nsl > LQ_SD_SL_SMALL) c.big = true;
				c.ne += (u64)g.nsl * g.nseg; c.nb += g.nsl; c.hits += hq;
			}
			bq_all.push_back((u32)g.bk);
			qg[i] = g;
			++i;
		}
		bq_all.push_back((u32)c.nb);
		c.q_hi = i; c.g_hi = (u32)segs.size();
		if (c.ne >= 0xfffffff0ULL || c.nb >= 0x7ffffff0ULL) throw std::domain_error("seed filter: too many (query, slice, segment) pieces in one chunk");
		chunks.push_back(c);
	}
	if (segs.size() >= 0x7ffffff0ULL) throw std::domain_error("seed filter: too many segments");
	u64 max_ne = 0, max_nb = 0, max_hits = 0;
	for (const Chunk &c : chunks) { max_ne = std::max(max_ne, c.ne); max_nb = std::max(max_nb, c.nb); max_hits = std::max(max_hits, c.hits); }
	W.qg.ensure(qg.size() * sizeof(SeedQ) + 16); W.segs.ensure(segs.size() * sizeof(SeedSeg) + 16); W.bq.ensure(bq_all.size() * 4 + 4); W.has.ensure((u64)n_q * 4 + 4);
	W.cnt.ensure((max_ne + 1) * 4); W.off.ensure((max_ne + 1) * 4); W.scnt.ensure((max_nb + 1) * 4); W.soff.ensure((max_nb + 1) * 4); W.bd.ensure((max_nb + 1) * sizeof(SeedBk)); W.big.ensure((max_nb + 2) * 4);
	W.rec.ensure(max_hits * 8 + 8);
	h2d(W.qg.as<SeedQ>(), qg.data(), qg.size(), s); h2d(W.segs.as<SeedSeg>(), segs.data(), segs.size(), s);
	h2d(W.bq.as<u32>(), bq_all.data(), bq_all.size(), s); h2d(W.has.as<u32>(), has.data(), has.size(), s);
	LQ_HIP_CHECK(hipStreamSynchronize(s));                    // (the host vectors above are pageable)
	SeedIn in; memset(&in, 0, sizeof(in));
	in.segs = W.segs.as<SeedSeg>(); in.qg = W.qg.as<SeedQ>(); in.h_off = W.h_off.as<u64>(); in.hit_start = J.hit_start; in.pos = pt.pos.as<u64>();
	in.qx = q.mx.as<u64>(); in.qy = q.my.as<u64>(); in.qmoff = q.moff.as<u64>(); in.qlen = q.d_len.as<u32>();
	SeedDecide dp; memset(&dp, 0, sizeof(dp));
	dp.n_min = n_min; dp.pair_bits = K.seed_pair_bits; dp.hwords = K.seed_hwords; dp.dcap = K.seed_dcap; dp.bigcap = K.seed_bigcap; dp.no_self = (int)P.no_self;
	for (u32 v : pt.rs.h_len) dp.max_tlen = std::max(dp.max_tlen, v);
	dp.dshift = 1; while (dp.dshift < 30 && (1u << dp.dshift) <= (u32)std::max<i32>(P.bw, 0)) ++dp.dshift;   // bins wider than the band (chain.c:55)
	const u32 span_const = (u32)(P.hpc ? 0 : P.k);
	DBuf d_stats;
	if (getenv("LQCOV_SEED_STATS")) { d_stats.ensure(64); dzero(d_stats.p, 64, s); dp.stats = d_stats.as<unsigned long long>(); }
	u64 n_surv = 0;
	DBuf &SV = *J.surv;
	SV.ensure(std::min<u64>(std::max<u64>(J.room_hint, 1024), K.seed_surv_max) * 8);   // (grows by chunk if the survivors outnumber the guess)
	u32 q_end = q_stop;
	for (const Chunk &c : chunks) {
		const u32 ns = c.g_hi - c.g_lo;
		if (!ns) continue;
		dzero(W.cnt.as<u32>() + c.ne, 4, s);
		{
			StageTimer t(this, s, "k_seed_count", c.hits * 8);
			LQ_LAUNCH(k_seed_count, ns, LQ_SD_THREADS, s, in, c.g_lo, W.cnt.as<u32>()); check_launch();
		}
		pr.exclusive_scan_u32_u32(W.cnt.as<u32>(), W.off.as<u32>(), c.ne + 1);
		{
			StageTimer t(this, s, "k_seed_scatter", c.hits * 16);
			if (c.big) LQ_LAUNCH((k_seed_scatter<LQ_SD_SL_BIG>), ns, LQ_SD_THREADS, s, in, c.g_lo, W.off.as<u32>(), bits, span_const, W.rec.as<u64>());
			else LQ_LAUNCH((k_seed_scatter<LQ_SD_SL_SMALL>), ns, LQ_SD_THREADS, s, in, c.g_lo, W.off.as<u32>(), bits, span_const, W.rec.as<u64>());
			check_launch();
		}
		SeedDecIn di; memset(&di, 0, sizeof(di));
		di.qg = W.qg.as<SeedQ>(); di.bq = W.bq.as<u32>() + c.bq_at; di.q_lo = c.q_lo; di.n_qc = c.q_hi - c.q_lo; di.off = W.off.as<u32>();
		di.qx = q.mx.as<u64>(); di.qy = q.my.as<u64>(); di.qmoff = q.moff.as<u64>(); di.qlen = q.d_len.as<u32>();
		di.self_off = pt.self_off.as<u32>(); di.self_rid = pt.self_rid.as<u32>(); di.ava = ava;
		di.tlen = pt.rs.d_len.as<u32>(); di.n_targets = pt.rs.n;
		dzero(W.big.p, 4, s);
		LQ_LAUNCH(k_seed_bdesc, nblk(c.nb, 256), 256, s, di, (u32)c.nb, dp.dcap, W.bd.as<SeedBk>(), W.big.as<u32>()); check_launch();
		{
			StageTimer t(this, s, "k_seed_decide", c.hits * 8);
			LQ_LAUNCH(k_seed_decide, (u32)c.nb, LQ_SD_DTHREADS, s, di, W.bd.as<SeedBk>(), dp, bits, span_const, W.rec.as<u64>(), W.scnt.as<u32>()); check_launch();
			LQ_LAUNCH(k_seed_decide_big, (u32)std::min<u64>(c.nb, 1024), LQ_SD_DTHREADS, s, di, W.bd.as<SeedBk>(), W.big.as<u32>(), dp, bits, span_const, W.rec.as<u64>(), W.scnt.as<u32>()); check_launch();
		}
		dzero(W.scnt.as<u32>() + c.nb, 4, s);
		pr.exclusive_scan_u32_u32(W.scnt.as<u32>(), W.soff.as<u32>(), c.nb + 1);
		u32 n_c = 0;
		d2h(&n_c, W.soff.as<u32>() + c.nb, 1, s);
		if (n_surv + n_c > K.seed_surv_max && n_surv) { q_end = c.q_lo; break; }   // the group is full: this chunk opens the next one
		if (dp.stats) fprintf(stderr, "[lqcov] seed filter: chunk of queries %u..%u: %llu hits, %llu buckets, %llu pieces, %u survivors (%llu before it; room for %zu)\n",
		                      c.q_lo, c.q_hi, (unsigned long long)c.hits, (unsigned long long)c.nb, (unsigned long long)c.ne, n_c, (unsigned long long)n_surv, SV.cap / 8);
		if ((n_surv + n_c) * 8 > SV.cap) {
			try { grow_keep(SV, n_surv * 8, (n_surv + n_c) * 8, s); }
			catch (const std::runtime_error &) { (void)hipGetLastError(); SV.release(); return false; }   // (no room for the survivors: the part is mapped without the filter)
		}
		LQ_LAUNCH(k_seed_collect, (u32)c.nb, 64, s, W.bd.as<SeedBk>(), W.rec.as<u64>(), W.scnt.as<u32>(), W.soff.as<u32>(), n_surv, J.base, SV.as<u64>(), J.aqf_off); check_launch();
		add_stage_bytes("k_seed_decide", (u64)n_c * 8);
		n_surv += n_c;
	}
	LQ_LAUNCH(k_seed_fill_off, 1, 64, s, W.has.as<u32>(), q_begin, q_end, J.base + n_surv, J.aqf_off); check_launch();
	if (J.h_aqf->size() != (size_t)n_q + 1) J.h_aqf->assign(n_q + 1, 0);
	d2h(J.h_aqf->data() + q_begin, J.aqf_off + q_begin, q_end - q_begin + 1, s);
	J.q_end = q_end; J.n_surv = n_surv;
	if (dp.stats) {
		unsigned long long st[8];
		d2h(st, dp.stats, 8, s);
		fprintf(stderr, "[lqcov] seed filter: %llu hits in %zu chunks, %zu segments; pairs with %u hits hold %llu (%.2f %%), survivors %llu (%.2f %%) of the queries %u..%u; %llu buckets beyond the block (%llu of them by pairs only), %llu pairs left without a histogram\n",
		        st[0], chunks.size(), segs.size(), n_min, st[1], st[0] ? 100.0 * st[1] / st[0] : 0.0, st[2], st[0] ? 100.0 * st[2] / st[0] : 0.0, q_begin, q_end, st[3], st[5], st[4]);
	}
	return true;
}

// The plan's next group of queries, from q_begin on (SeedPlan::q_begin / q_end, h_aqf, surv).  swapped: the part is being mapped and
// its plan's buffers stand in the handle's members of the same names (swap_plan).
bool lqcov_handle::seed_group(Part &pt, SeedPlan &S, bool swapped, hipStream_t s, Prim &pr, u32 q_begin)
{
	SeedJob J;
	J.hit_start = (swapped ? hit_start : S.hit_start).as<u64>(); J.hit_n = (swapped ? hit_n : S.hit_n).as<u32>(); J.keep = (swapped ? keep : S.keep).as<u32>();
	J.aqf_off = (swapped ? aqf_off : S.aqf_off).as<u64>();
	J.h_qmoff = &S.h_qmoff; J.h_aqf = &S.h_aqf;
	J.surv = swapped ? &surv : &S.surv; J.base = 0; J.room_hint = S.nA_total / 16;
	J.q_begin = q_begin; J.q_stop = q.n;
	if (!seed_filter(pt, s, pr, seed_ws, S.rec_nmin, S.rec_jb, S.rec_db, J)) return false;
	S.n_written = (q_begin ? S.n_written : 0) + J.n_surv; S.q_begin = q_begin; S.q_end = J.q_end;
	return true;
}

// ---- map every query against one part (lqmap.c:207-326) -----------------------------------------
// a part's seed plan (SeedPlan, engine.hpp) on stream s with scan scratch pr
void lqcov_handle::plan_part(Part &pt, hipStream_t s, Prim &pr, bool defer_filter)
{
	SeedPlan &S = pt.plan;
	S.valid = false; S.lazy = false;
	const u32 n_q = q.n;
	const u64 n_qm = q.n_mini;
	S.n_q = n_q; S.n_qm = n_qm; S.mid_occ = mid_occ; S.nA_total = 0; S.n_mp_total = 0; S.n_written = 0;
	S.h_aq.assign(n_q + 1, 0); S.h_qmoff.assign(n_q + 1, 0); S.h_aqf.clear();
	if (n_q == 0) { S.valid = true; return; }
	const AvaView ava{P.ava ? pt.t_rank.as<u32>() : nullptr, P.ava ? pt.q_lo.as<u32>() : nullptr};
	S.dup.ensure(n_qm * 4 + 4); S.qdirty.ensure((u64)n_q * 4 + 4);
	S.hit_start.ensure(n_qm * 8 + 8); S.hit_n.ensure(n_qm * 4 + 4); S.a_cnt.ensure(n_qm * 4 + 4); S.keep.ensure(n_qm * 4 + 4);
	S.a_off.ensure(n_qm * 8 + 8); S.mp_off.ensure(n_qm * 8 + 8);
	S.aq_off.ensure((n_q + 1) * 8); S.mpq_off.ensure((n_q + 1) * 8); S.avg_qspan.ensure((n_q + 1) * 4);
	u64 nA_total = 0, n_mp_total = 0;
	if (n_qm) {
		{
			StageTimer t(this, s, "k_seed_probe", n_qm * (16 + 16 + 20));
			LQ_LAUNCH(k_seed_probe, nblk(n_qm, 256), 256, s, q.mx.as<u64>(), q.my.as<u64>(), q_owner.as<u32>(), n_qm,
			          pt.tkey.as<u64>(), pt.tstart.as<u64>(), pt.tcnt.as<u32>(), pt.cap_bits, pt.pos.as<u64>(),
			          mid_occ, (int)P.no_self, pt.self_off.as<u32>(), pt.self_rid.as<u32>(), ava,
			          S.hit_start.as<u64>(), S.hit_n.as<u32>(), S.a_cnt.as<u32>(), S.keep.as<u32>());
			check_launch();
		}
		{	// minimizers / queries whose anchors can repeat an x (kernels_psort.hpp): everything else is sorted without klib's walks
			u32 tbits = 10;
			while (((u64)1 << tbits) < 2 * n_qm && tbits < 31) ++tbits;
			S.dup_table.ensure(((u64)1 << tbits) * 4);
			dzero(S.dup.p, n_qm * 4, s); dzero(S.qdirty.p, (u64)n_q * 4, s); dzero(S.dup_table.p, ((u64)1 << tbits) * 4, s);
			StageTimer t(this, s, "k_dup_mark", n_qm * 24);
			LQ_LAUNCH(k_dup_mark, nblk(n_qm, 256), 256, s, q.mx.as<u64>(), q.my.as<u64>(), q_owner.as<u32>(), S.a_cnt.as<u32>(), n_qm,
			          S.dup_table.as<u32>(), tbits, S.dup.as<u32>(), S.qdirty.as<u32>());
			check_launch();
		}
		pr.exclusive_scan_u32_u64(S.a_cnt.as<u32>(), S.a_off.as<u64>(), n_qm);
		pr.exclusive_scan_u32_u64(S.keep.as<u32>(), S.mp_off.as<u64>(), n_qm);
		u64 lo[2]; u32 lc[2];
		d2h(&lo[0], S.a_off.as<u64>() + n_qm - 1, 1, s); d2h(&lc[0], S.a_cnt.as<u32>() + n_qm - 1, 1, s);
		d2h(&lo[1], S.mp_off.as<u64>() + n_qm - 1, 1, s); d2h(&lc[1], S.keep.as<u32>() + n_qm - 1, 1, s);
		nA_total = lo[0] + lc[0]; n_mp_total = lo[1] + lc[1];
	}
	S.nA_total = nA_total; S.n_mp_total = n_mp_total; S.n_written = nA_total;
	S.mini_pos.ensure(n_mp_total * 8 + 8);
	// (per query: anchor and mini_pos ranges, avg_qspan from the unfiltered totals; the skip verdict of esterr.c:85-91 waits for the mapping)
	LQ_LAUNCH(k_query_prep, nblk(n_q + 1, 4), 256, s, q.moff.as<u64>(), S.a_off.as<u64>(), S.mp_off.as<u64>(), n_qm, nA_total, n_mp_total, n_q,
	          q.mx.as<u64>(), S.a_cnt.as<u32>(), S.keep.as<u32>(), q.d_len.as<u32>(),
	          S.aq_off.as<u64>(), S.mpq_off.as<u64>(), S.avg_qspan.as<float>(), (const u64*)nullptr, (float*)nullptr, (u32*)nullptr, 0, 1);
	check_launch();
	S.qklib.ensure((u64)n_q * 4 + 4);
	LQ_LAUNCH(k_query_klib, nblk(n_q, 256), 256, s, S.aq_off.as<u64>(), S.qdirty.as<u32>(), n_q, (int)K.all_klib, S.qklib.as<u32>()); check_launch();
	d2h(S.h_aq.data(), S.aq_off.as<u64>(), n_q + 1, s);
	d2h(S.h_qmoff.data(), q.moff.as<u64>(), n_q + 1, s);
	S.bucketed = false;
	if (!K.ties_klib) {
		// The seed hits that can be part of a chain at all (kernels_seed.hpp): records, dense per query.  Without the filter
		// (LQCOV_FILTER=0, a chain may be a single anchor, or a record would not fit 64 bits) the first pass writes every hit.
		S.qzero.ensure((u64)n_q * 4 + 4); dzero(S.qzero.p, (u64)n_q * 4 + 4, s);
		S.aqf_off.ensure((n_q + 1) * 8);
		if (n_qm) { LQ_LAUNCH(k_mini_pos, nblk(n_qm, 256), 256, s, q.mx.as<u64>(), q.my.as<u64>(), S.keep.as<u32>(), S.mp_off.as<u64>(), n_qm, S.mini_pos.as<u64>()); check_launch(); }
		const u32 n_min = K.filter ? run_n_min() : 0;
		u32 max_tlen = 0, max_qlen = 0; u64 max_nm = 1;
		for (u32 v : pt.rs.h_len) max_tlen = std::max(max_tlen, v);
		for (u32 v : q.h_len) max_qlen = std::max(max_qlen, v);
		for (u32 i = 0; i < n_q; ++i) max_nm = std::max<u64>(max_nm, S.h_qmoff[i + 1] - S.h_qmoff[i]);
		auto bits_for = [](u64 v) { u32 b = 1; while (b < 63 && (v >> b)) ++b; return b; };   // bits that hold 0 .. v
		const u32 jb = bits_for(max_nm - 1), db = bits_for((u64)max_tlen + max_qlen + 257), rb = bits_for(pt.rs.n ? pt.rs.n - 1 : 0);
		u64 max_hits = 0;
		for (u32 i = 0; i < n_q; ++i) max_hits = std::max<u64>(max_hits, S.h_aq[i + 1] - S.h_aq[i]);
		if (n_min >= 2 && n_min <= 15 && nA_total && jb + db + 1 + rb <= 64 && db <= 31 && jb <= 31 && max_hits < 0x7fff0000ULL) {
			S.rec_nmin = n_min; S.rec_jb = jb; S.rec_db = db;
			// (measured and dropped, round 5: the filter run batch by batch inside map_part, each lane starting as soon as its batch is
			// decided -- 586-588 ms per step at configs[2] against 576: beside the lanes' kernels the filter's take three times as long)
			// (an allocation that fails inside the filter -- its bucket buffer is up to 8 GB -- leaves the part without one: every hit is written)
			if (defer_filter) { S.lazy = true; S.bucketed = true; S.q_begin = 0; S.q_end = n_q; S.h_aqf = S.h_aq; S.n_written = 0; }
			else {
				try { S.bucketed = seed_group(pt, S, false, s, pr, 0); }
				catch (const std::runtime_error &) { (void)hipGetLastError(); S.surv.release(); for (DBuf *b : { &seed_ws.rec, &seed_ws.cnt, &seed_ws.off, &seed_ws.bd }) b->release(); S.bucketed = false; }
			}
		}
		if (!S.bucketed) {
			S.q_begin = 0; S.q_end = n_q;
			S.h_aqf = S.h_aq;
			LQ_HIP_CHECK(hipMemcpyAsync(S.aqf_off.p, S.aq_off.p, (n_q + 1) * 8, hipMemcpyDeviceToDevice, s));
			S.n_written = nA_total;
		}
	}
	LQ_HIP_CHECK(hipStreamSynchronize(s));
	S.valid = true;
}

// the plan's buffers <-> the handle's work buffers of the same names
void lqcov_handle::swap_plan(SeedPlan &S)
{
	hit_start.swap(S.hit_start); hit_n.swap(S.hit_n); a_cnt.swap(S.a_cnt); keep.swap(S.keep); dup.swap(S.dup); qdirty.swap(S.qdirty); dup_table.swap(S.dup_table);
	a_off.swap(S.a_off); mp_off.swap(S.mp_off); aq_off.swap(S.aq_off); mpq_off.swap(S.mpq_off); avg_qspan.swap(S.avg_qspan); qklib.swap(S.qklib); mini_pos.swap(S.mini_pos); qzero.swap(S.qzero);
	surv.swap(S.surv); aqf_off.swap(S.aqf_off);
}

void lqcov_handle::map_part(Part &pt)
{
	if (!pt.built) throw std::logic_error("part not built");
	finished = false;
	const u32 n_q = q.n;
	const u64 n_qm = q.n_mini;
	last_n_anchors = 0;
	n_dbg_host = 0;
	if (n_q == 0) return;
	// the part's seed plan: made with its index (build_index), or here if it was not (no queries then) or no longer fits (mid_occ
	// set afterwards: the parts of a round of PartRunner are built before part 0's mid_occ arrives)
	if (!pt.plan.valid || pt.plan.mid_occ != mid_occ || pt.plan.n_q != n_q || pt.plan.n_qm != n_qm || pt.plan.h_aqf.empty() != K.ties_klib || (pt.plan.bucketed && pt.plan.q_begin != 0)) plan_part(pt, stream, prim);
	lq_tl("main", 0, "map_part begins");
	swap_plan(pt.plan);
	struct PlanGuard { lqcov_handle *h; SeedPlan &S; ~PlanGuard() { h->swap_plan(S); } } plan_guard{this, pt.plan};
	++active_maps;
	struct ActiveGuard { lqcov_handle *h; ~ActiveGuard() { --h->active_maps; } } active_guard{this};
	const std::vector<u64> &h_aq = pt.plan.h_aq, &h_qmoff = pt.plan.h_qmoff, &h_aqf = pt.plan.h_aqf;
	const u64 nA_total = pt.plan.nA_total, n_mp_total = pt.plan.n_mp_total;
	last_n_anchors = nA_total;
	skip.ensure((n_q + 1) * 4);
	LQ_LAUNCH(k_query_prep, nblk(n_q + 1, 4), 256, stream, q.moff.as<u64>(), a_off.as<u64>(), mp_off.as<u64>(), n_qm, nA_total, n_mp_total, n_q,
	          q.mx.as<u64>(), a_cnt.as<u32>(), keep.as<u32>(), q.d_len.as<u32>(),
	          aq_off.as<u64>(), mpq_off.as<u64>(), avg_qspan.as<float>(), lambda.as<u64>(), avg_k.as<float>(), skip.as<u32>(), distributed ? 0 : 1, 2);
	check_launch();
	last_n_written = pt.plan.n_written;
	const bool opt = !K.ties_klib;
	const bool dbg = (debug_flags & 1) != 0;
	if (dbg) {
		dbg_cap = nA_total / (P.min_cnt > 0 ? P.min_cnt : 1) + 16;
		dbg_chains.ensure(dbg_cap * sizeof(ChainRec)); n_dbg.ensure(8);
		dzero(n_dbg.p, 8, stream);
	}
	if (anchor_budget == 0) {
		// The first part stands (reads, minimizers, index, the build's work space -- all kept for the parts to come, of which the
		// first is the largest: index.c:244) and so do the queries: the lanes share 85 % of what is free now.  Per anchor a lane
		// holds A and B (32 B), one record array (8), 20 B of scratch, digit + destination (5), ~12 B of lists and per-sub-array
		// rows, ~6 B of runs and intervals, and the buffers grow with an eighth of head room: ~95 B.
		size_t fr = 0, tot = 0;
		hipMemGetInfo(&fr, &tot);
		if (fr > hbm_reserve) fr -= hbm_reserve; else fr = 0;     // (lqcov_reserve_hbm: e.g. the part the caller builds while this one is mapped)
		{	// ... and the survivors of that part's seed plan: as many as this part's, with the head room they grow by
			const size_t sv = hbm_reserve ? surv.cap + surv.cap / 2 : 0;
			if (fr > sv) fr -= sv; else fr = 0;
		}
		// (0.75 since round 4: a lane's buffers now grow in more steps -- a small first pass, second passes of varying size -- and a
		// block the stream-ordered pool got back is not always the one the next, larger request can use; at 0.85 the ultra-long
		// slice of configs[4] ran the device out of memory inside the runtime (HSA_STATUS_ERROR_OUT_OF_RESOURCES))
		anchor_budget = (u64)((double)fr * 0.75 / 104.0 / n_lanes);
	}
	if (anchor_budget > (1ULL << 31) - 4096) anchor_budget = (1ULL << 31) - 4096;   // (a record names its anchor in 31 bits)
	if (anchor_budget < 1024) anchor_budget = 1024;
	while (lanes.size() < (size_t)n_lanes) {
		lanes.emplace_back(new MapLane());
		LQ_HIP_CHECK(hipStreamCreate(&lanes.back()->stream));
		LQ_HIP_CHECK(hipStreamCreate(&lanes.back()->stream2));
		{	// Token walks are latency-bound single-lane waves that live for milliseconds: left alone they fill the wave slots and
			// the LDS of every CU and the bandwidth kernels of the other streams crawl (rocprofv3, configs[2]: k_ps_scatter 66 ms
			// alone, 1100 ms beside the walkers).  Their stream may only use every fourth CU; 64 CUs x 32 waves are plenty for them.
			// (Keeping the other streams off those CUs as well was measured in round 3: slower, 2.15 vs 1.85-2.1 s per step.)
			// (Round 3: a different quarter of the CUs per lane, or 128 / 192 CUs instead of 64: no change, 1.69-1.71 s per step whatever
			// the mask; no mask at all: 2.03 s.)
			uint32_t mask[8];
			for (int i = 0; i < 8; ++i) mask[i] = K.walk_cu_mask;
			if (hipExtStreamCreateWithCUMask(&lanes.back()->streamW, 8, mask) != hipSuccess) { (void)hipGetLastError(); LQ_HIP_CHECK(hipStreamCreate(&lanes.back()->streamW)); }
			if (hipExtStreamCreateWithCUMask(&lanes.back()->streamW2, 8, mask) != hipSuccess) { (void)hipGetLastError(); LQ_HIP_CHECK(hipStreamCreate(&lanes.back()->streamW2)); }
		}
		LQ_HIP_CHECK(hipEventCreateWithFlags(&lanes.back()->ev_w0, hipEventDisableTiming));
		LQ_HIP_CHECK(hipEventCreateWithFlags(&lanes.back()->ev_w1, hipEventDisableTiming));
		LQ_HIP_CHECK(hipEventCreateWithFlags(&lanes.back()->ev_w2, hipEventDisableTiming));
		LQ_HIP_CHECK(hipEventCreateWithFlags(&lanes.back()->ev_fork, hipEventDisableTiming));
		LQ_HIP_CHECK(hipEventCreateWithFlags(&lanes.back()->ev_join, hipEventDisableTiming));
		lanes.back()->prim.stream = lanes.back()->stream;
	}
	{	// every lane's work space (prim.hpp, LqArena): as much as its largest batch of this part can need, taken while the device is
		// idle -- from the stream-ordered pool (one block, handed back only when the lanes go: plain hipMalloc of tens of gigabytes
		// takes seconds here, 30 ms per GB measured)
		size_t need = (size_t)std::min<u64>(anchor_budget, std::max<u64>(nA_total, 1)) * 104 / 9 * 8 + ((size_t)16 << 20);
#ifdef LQ_EMU
		need = std::min<size_t>(need, (size_t)64 << 20);            // (the test emulator fills fresh memory with a pattern)
#endif
		// (only where the first pass leaves a small part of the hits: a batch whose first pass fills the lane's share by itself --
		// ultra-long reads keep a third of their hits -- needs its buffers again in the second pass, and pieces of an arena are not
		// given back: there the lanes allocate buffer by buffer, grow-only, as before)
		const u64 group_hits = opt && pt.plan.bucketed ? h_aq[pt.plan.q_end] - h_aq[pt.plan.q_begin] : 0;
		const bool few = opt && pt.plan.bucketed && pt.plan.q_begin == 0 && pt.plan.n_written * 4 <= group_hits;
		if (!few) for (auto &Lp : lanes) if (Lp->arena.base) { Lp->drop_arena_buffers(); Lp->arena.base = nullptr; Lp->arena.size = 0; Lp->arena_buf.release(); }
		if (few) for (auto &Lp : lanes) if (Lp->arena.size < need) {
			Lp->drop_arena_buffers();
			Lp->arena.base = nullptr; Lp->arena.size = 0; Lp->arena.used = 0;
			lq_alloc_stream = stream;
			struct AllocGuard { ~AllocGuard() { lq_alloc_stream = nullptr; } } alloc_guard;
			try { Lp->arena_buf.ensure(need); Lp->arena.base = Lp->arena_buf.as<char>(); Lp->arena.size = Lp->arena_buf.cap; }
			catch (const std::runtime_error &) { (void)hipGetLastError(); }   // (no room for it in one piece: the lane allocates buffer by buffer)
		}
		LQ_HIP_CHECK(hipStreamSynchronize(stream));
	}
	{
		u32 npv = 0;
		d2h(&npv, n_pv.as<u32>(), 1, stream);
		pv_reserved = npv;
		// room for this part's intervals up front: growing the pool later means draining every lane (map_batch)
		const u64 want = std::min<u64>((u64)npv + nA_total / 16 + 4096, 0xfffffff0ULL);
		if (want > pv_cap) {
			grow_keep(pv, (u64)npv * sizeof(Ivl), want * sizeof(Ivl), stream);
			pv_cap = (u32)std::min<u64>(pv.cap / sizeof(Ivl), 0xfffffff0ULL);
		}
	}
	const bool can_thread =
#ifndef LQ_EMU
		profiling != 1 && !dbg;
#else
		false;
#endif
	// runs the batches on the lanes
	const bool lazy = opt && pt.plan.lazy && pt.plan.bucketed;
	std::atomic<u64> lazy_written{0};
	// one batch on one lane.  A lazy plan (the first part of a job): the lane first decides which of its queries' seed hits can reach
	// a chain -- the filter of kernels_seed.hpp on the lane's own stream, one lane at a time (they share its work space) -- into a
	// survivor list of its own, then maps them; the next lane decides its batch under this one's mapping.
	auto one_batch = [&](MapLane &L, u32 q0, u32 q1) {
		if (!lazy) { map_batch(L, pt, q0, q1, h_aq, h_aqf, h_qmoff, dbg); return; }
		int lane_id = 0; for (size_t i_ = 0; i_ < lanes.size(); ++i_) if (lanes[i_].get() == &L) lane_id = (int)i_;
		L.aqf_l.ensure(((u64)n_q + 1) * 8);
		for (u32 qb = q0; qb < q1; ) {
			SeedJob J;
			J.hit_start = hit_start.as<u64>(); J.hit_n = hit_n.as<u32>(); J.keep = keep.as<u32>();
			J.aqf_off = L.aqf_l.as<u64>(); J.h_qmoff = &h_qmoff; J.h_aqf = &L.h_aqf_l;
			J.surv = &L.surv_l; J.base = 0; J.room_hint = (h_aq[q1] - h_aq[qb]) / 16;
			J.q_begin = qb; J.q_stop = q1;
			bool ok = false;
			try { ok = seed_filter(pt, L.stream, L.prim, seed_ws, pt.plan.rec_nmin, pt.plan.rec_jb, pt.plan.rec_db, J); }
			catch (const std::runtime_error &) { (void)hipGetLastError(); ok = false; }
			if (!ok) throw std::runtime_error("seed filter: no room for a batch's survivors (LQCOV_PLAN_LAZY=0 maps the part without the filter instead)");
			lq_tl("lane", lane_id, "batch decided, survivors", (double)J.n_surv);
			lazy_written += J.n_surv;
			L.use_surv = L.surv_l.as<u64>(); L.use_aqf = L.aqf_l.as<u64>();
			struct UseGuard { MapLane &L; ~UseGuard() { L.use_surv = nullptr; L.use_aqf = nullptr; } } use_guard{L};
			// (what the filter left of [qb, q_end), in pieces the lane's work space holds)
			for (u32 a = qb; a < J.q_end; ) {
				u32 b = a + 1;
				while (b < J.q_end && L.h_aqf_l[b + 1] - L.h_aqf_l[a] <= anchor_budget) ++b;
				map_batch(L, pt, a, b, h_aq, L.h_aqf_l, h_qmoff, dbg);
				a = b;
			}
			qb = J.q_end;
		}
	};
	auto run_batches = [&](const std::vector<std::pair<u32, u32>> &batches) {
		const bool concurrent = can_thread && n_lanes > 1 && batches.size() > 1;
		if (!concurrent) {
			for (size_t i = 0; i < batches.size(); ++i) {
				one_batch(*lanes[i % n_lanes], batches[i].first, batches[i].second);
			}
			for (auto &L : lanes) LQ_HIP_CHECK(hipStreamSynchronize(L->stream));
			return;
		}
		std::atomic<size_t> next(0);
		{ std::lock_guard<std::mutex> lk(gate_mu); gate_count = 0; }
		std::vector<std::exception_ptr> errs(n_lanes);
		std::vector<std::thread> th;
		for (int li = 0; li < n_lanes; ++li)
			th.emplace_back([&, li]() {
				try {
					LQ_HIP_CHECK(hipSetDevice(device));
#ifndef LQ_EMU
					lq_segv_altstack();
#endif
					MapLane &L = *lanes[li];
					// (round 6: the switch that let a lane's buffers grow from HIP's stream-ordered pool -- rounds 2-4's way, under ROCm 7.2 a
					// GPU memory fault waiting to happen, see DESIGN.md section 2 -- is gone; only the lanes' arenas come from the pool, taken once)
					{	// staggered start: lane li begins when li batches have got past their sort (or ended), so that
						// one lane's serial tails run under another lane's wide kernels instead of side by side
						std::unique_lock<std::mutex> lk(gate_mu);
						gate_cv.wait(lk, [&] { return gate_count >= li || next.load() >= batches.size(); });
					}
					for (;;) {
						const size_t i = next.fetch_add(1);
						if (i >= batches.size()) break;
						one_batch(L, batches[i].first, batches[i].second);
					}
					LQ_HIP_CHECK(hipStreamSynchronize(L.stream));
				} catch (...) { errs[li] = std::current_exception(); next.store(batches.size()); gate_cv.notify_all(); }
			});
		for (auto &t : th) t.join();
		for (auto &e : errs) if (e) { hipDeviceSynchronize(); std::rethrow_exception(e); }
	};
	// as few batches as the work space allows, a multiple of the lane count, of about equal totals of `off` (the anchors the first
	// pass writes, or the seed hits): every batch has a serial critical path that does not shrink with the batch.  (Cutting the
	// last round finer was measured on MI355X at configs[2] in round 3: 1.75-1.78 s per step against 1.68-1.70 s; round 4: the
	// queries in six chunks with the survivors of a chunk decided under the mapping of the chunk before: 1160 ms per step against 888.)
	auto cut_batches = [&](const std::vector<u64> &off, u32 g_begin, u32 g_end) {
		std::vector<std::pair<u32, u32>> batches;
		const u64 total = off[g_end] - off[g_begin];
		u64 nb = (total + anchor_budget - 1) / anchor_budget;
		if (nb < (u64)n_lanes && h_aq[g_end] - h_aq[g_begin] >= ((u64)n_lanes << 24)) nb = n_lanes;   // (by the seed hits, not by what the filter left of them: the second pass works on the hits)
		if (nb > (u64)n_lanes) nb = (nb + n_lanes - 1) / n_lanes * n_lanes;
		if (nb == 0) nb = 1;
		u64 left = nb;
		for (u32 q0 = g_begin; q0 < g_end; ) {
			const u64 rem = off[g_end] - off[q0];
			const u64 lim = std::min(anchor_budget, left > 1 ? (rem + left - 1) / left : rem);
			u32 q1 = q0 + 1;
			while (q1 < g_end && off[q1 + 1] - off[q0] <= lim) ++q1;
			batches.emplace_back(q0, q1);
			q0 = q1;
			if (left > 1) --left;
		}
		return batches;
	};
	bool regrouped = false;
	if (lazy) {
		// batches of about equal seed hits, one per lane (the filter's cost and the second pass go by the hits)
		std::vector<std::pair<u32, u32>> batches;
		const u64 total = h_aq[n_q] - h_aq[0];
		const u32 nb = (u32)std::max<int>(1, n_lanes * K.lazy_batches);
		u32 q0 = 0;
		for (u32 b = 0; b < nb && q0 < n_q; ++b) {
			const u64 lim = h_aq[q0] + (total - (h_aq[q0] - h_aq[0])) / (nb - b);
			u32 q1 = q0 + 1;
			while (q1 < n_q && h_aq[q1 + 1] <= lim) ++q1;
			if (b + 1 == nb) q1 = n_q;
			batches.emplace_back(q0, q1);
			q0 = q1;
		}
		run_batches(batches);
		pt.plan.n_written = lazy_written.load(); last_n_written = pt.plan.n_written;
		pt.plan.valid = false;                                      // (no survivor list to map the part from again: the next map_part plans anew)
	} else
	// The plan holds the survivors of a group of queries (all of them, unless survivors abound: SeedPlan::q_end); the group's batches
	// are mapped, then the next group is planned -- with every lane drained, on the handle's own stream.
	for (u32 g_begin = 0, g_end = n_q; ; ) {
		if (opt && pt.plan.bucketed) { g_begin = pt.plan.q_begin; g_end = pt.plan.q_end; }
		run_batches(cut_batches(opt ? h_aqf : h_aq, g_begin, g_end));   // (cut by the anchors the first pass writes)
		if (g_end >= n_q) break;
		regrouped = true;
		bool ok = false;
		try { ok = seed_group(pt, pt.plan, true, stream, prim, g_end); }
		catch (const std::runtime_error &) { (void)hipGetLastError(); for (DBuf *b : { &seed_ws.rec, &seed_ws.cnt, &seed_ws.off, &seed_ws.bd }) b->release(); ok = false; }
		if (!ok) {                                                  // (no room: the rest of the part without the filter)
			pt.plan.bucketed = false;
			pt.plan.h_aqf = pt.plan.h_aq;
			LQ_HIP_CHECK(hipMemcpyAsync(aqf_off.p, aq_off.p, (n_q + 1) * 8, hipMemcpyDeviceToDevice, stream));
			LQ_HIP_CHECK(hipStreamSynchronize(stream));
			pt.plan.n_written += h_aq[n_q] - h_aq[g_end];
			g_begin = g_end; g_end = n_q;
		}
		last_n_written = pt.plan.n_written;
	}
	if (regrouped) pt.plan.valid = false;                         // (the plan no longer starts at the first query: made again if the part is mapped again)
	if (dbg) { unsigned long long nd = 0; d2h(&nd, n_dbg.as<unsigned long long>(), 1, stream); n_dbg_host = nd; }
	sat_replay_part(pt, h_aq, h_qmoff);
	LQ_HIP_CHECK(hipStreamSynchronize(stream));
	lq_tl("main", 0, "map_part ends");
}

// ---- pass 2 (minimap2-coverage.c:545-566) ---------------------------------------------------------
void lqcov_handle::finish()
{
	const u32 n_q = q.n;
	rows.assign(n_q, lqcov_row());
	regs.clear(); mregs.clear();
	if (n_q == 0) { finished = true; return; }
	u32 npv = 0;
	d2h(&npv, n_pv.as<u32>(), 1, stream);
	DBuf rowdev, pvq_off, k1, k2, s1, s2, scratch, dregs, dmregs, cnt2;
	rowdev.ensure((u64)n_q * sizeof(RowDev)); dzero(rowdev.p, (u64)n_q * sizeof(RowDev), stream);
	pvq_off.ensure((n_q + 1) * 4); cnt2.ensure(8); dzero(cnt2.p, 8, stream);
	k1.ensure((u64)npv * 4 + 4); k2.ensure((u64)npv * 4 + 4); s1.ensure((u64)npv * 8 + 8); s2.ensure((u64)npv * 8 + 8);
	scratch.ensure((u64)npv * 8 + 8);
	dregs.ensure(((u64)npv + 1) * sizeof(RegionT)); dmregs.ensure(((u64)npv + 1) * sizeof(RegionT));
	if (npv) {
		LQ_LAUNCH(k_split_ivl, nblk(npv, 256), 256, stream, pv.as<Ivl>(), npv, k1.as<u32>(), s1.as<u64>()); check_launch();
		prim.sort_pairs_u32_u64(k1.as<u32>(), k2.as<u32>(), s1.as<u64>(), s2.as<u64>(), npv, 32);
	}
	LQ_LAUNCH(k_ivl_offsets, nblk(n_q + 1, 256), 256, stream, k2.as<u32>(), npv, n_q, pvq_off.as<u32>()); check_launch();
	{
		StageTimer t(this, "k_reliable");
		LQ_LAUNCH(k_reliable, nblk(n_q, 64), 64, stream, s2.as<u64>(), pvq_off.as<u32>(), n_q, (u32)P.min_coverage, scratch.as<u32>(),
		          dregs.as<RegionT>(), cnt2.as<u32>(), dmregs.as<RegionT>(), cnt2.as<u32>() + 1, rowdev.as<RowDev>());
		check_launch();
	}
	for (auto &kv : sat_cnt) {                                   // replayed counters (sat_replay_part) take the place of the 32-bit counts
		u64 off[2];
		d2h(off, cnt_off_dev() + kv.first, 2, stream);
		if (kv.second.size() != (size_t)(off[1] - off[0])) throw std::logic_error("replayed counters: layout changed");
		h2d(cnts.as<u32>() + off[0], kv.second.data(), kv.second.size(), stream);
	}
	{
		StageTimer t(this, "k_cnt_stats", q.n_mini * 8);
		LQ_LAUNCH(k_cnt_stats, nblk(n_q, 64), 64, stream, cnts.as<u32>(), cnt_off_dev(), own_cnt_layout ? d_nsize.as<u32>() : (const u32*)nullptr, n_q, rowdev.as<RowDev>(), qflags.as<u32>(), cnt_max);
		check_launch();
	}
	std::vector<RowDev> hr(n_q);
	std::vector<u64> hl(n_q), hl2(n_q), hmoff(n_q + 1);
	std::vector<float> hk(n_q);
	std::vector<u32> hf(n_q);
	std::vector<double> hp(n_q);
	u32 nr[2] = {0, 0};
	d2h(hr.data(), rowdev.as<RowDev>(), n_q, stream);
	d2h(hl.data(), lambda.as<u64>(), n_q, stream); d2h(hl2.data(), lambda2.as<u64>(), n_q, stream);
	d2h(hk.data(), avg_k.as<float>(), n_q, stream); d2h(hf.data(), qflags.as<u32>(), n_q, stream);
	d2h(hp.data(), qual_psum.as<double>(), n_q, stream);
	d2h(hmoff.data(), q.moff.as<u64>(), n_q + 1, stream);
	d2h(nr, cnt2.as<u32>(), 2, stream);
	regs.resize(nr[0]); mregs.resize(nr[1]);
	static_assert(sizeof(RegionT) == sizeof(lqcov_region), "region layout");
	d2h((RegionT*)regs.data(), dregs.as<RegionT>(), nr[0], stream);
	d2h((RegionT*)mregs.data(), dmregs.as<RegionT>(), nr[1], stream);
	for (u32 i = 0; i < n_q; ++i)
		if (hf[i] & 2u)
			throw std::domain_error("query " + q.names[i] + ": a chain matched a minimizer beyond the counters sized from the command line's -k/-w/-H; "
			                        "the reference overruns its counter array there (minimap2-coverage.c:422, esterr.c:131) -- pass the prebuilt index's -k/-w/-H");
	for (u32 i = 0; i < n_q; ++i) {
		lqcov_row &r = rows[q_perm[i]];
		r.lambda = hl[i]; r.lambda2 = hl2[i]; r.qual_psum = hp[i]; r.qlen = q.h_len[i];
		r.n_mini = own_cnt_layout ? h_nsize[i] : (u32)(hmoff[i + 1] - hmoff[i]); r.n_match = hr[i].n_match; r.avg_k = hk[i];
		r.reg_off = hr[i].reg_off; r.n_reg = hr[i].n_reg; r.mreg_off = hr[i].mreg_off; r.n_mreg = hr[i].n_mreg;
		r.has_qual = q_has_qual ? 1u : 0u; r.flags = hf[i] | (sat_cnt.count(i) ? LQCOV_ROW_REPLAYED : 0u);
	}
	finished = true;
}

// rows as text (minimap2-coverage.c:567-605); name_of(i) = the name of the query of row i
void lq_format_rows(FILE *out, int filter_flag, const lqcov_row *rows, u32 n_rows, const lqcov_region *regs, const lqcov_region *mregs,
                    const std::function<const char *(u32)> &name_of)
{
	std::string line;
	char buf[128];
	for (u32 i = 0; i < n_rows; ++i) {
		const lqcov_row &r = rows[i];
		const double div = r.n_match > 0 ? logf((float)r.n_mini / (float)(int32_t)r.n_match) / r.avg_k : 1.0;   // :563
		double mq;
		if (r.has_qual) mq = -10 * log10(r.qual_psum / (int)r.qlen);                                          // lqutils.c:57
		else { volatile double z = 0.0; volatile int zl = 0; mq = -10 * log10(z / zl); }                                      // FASTA query: the reference's 0/0
		line.clear();
		line += name_of(i); line += '\t';
		snprintf(buf, sizeof(buf), "%d\t%" PRIu64 "\t", (int)r.qlen, r.lambda); line += buf;
		if (r.n_reg > 0) {
			u32 tot = 0;
			for (u32 k = 0; k < r.n_reg; ++k) {
				const lqcov_region &g = regs[r.reg_off + k];
				snprintf(buf, sizeof(buf), "%s%d-%d", k ? "," : "", (int)g.start, (int)g.end); line += buf;
				tot += g.end - g.start;
			}
			line += '\t';
			if (r.n_mreg > 0) {
				for (u32 k = 0; k < r.n_mreg; ++k) {
					const lqcov_region &g = mregs[r.mreg_off + k];
					snprintf(buf, sizeof(buf), "%s%d-%d", k ? "," : "", (int)g.start, (int)g.end); line += buf;
				}
			} else line += '0';
			if (filter_flag) snprintf(buf, sizeof(buf), "\t%.3f\t%.3f\t%.3f\t0.0\n", (double)tot / (int)r.qlen, mq, div);
			else snprintf(buf, sizeof(buf), "\t%.3f\t%.3f\t%.3f\t%.3f\n", (double)r.lambda / tot, mq, div, (double)r.lambda2 / tot);
			line += buf;
		} else {
			snprintf(buf, sizeof(buf), "0\t0\t0.0\t%.3f\t%.3f\t0.0\n", mq, div); line += buf;
		}
		fwrite(line.data(), 1, line.size(), out);
	}
}

void lqcov_handle::write_table(FILE *out)
{
	if (!finished) throw std::logic_error("finish() has not run");
	lq_format_rows(out, P.filter_flag, rows.data(), q.n, regs.data(), mregs.data(), [&](u32 i) { return q.names[q_inv[i]].c_str(); });
}

// ---- the whole run from files (minimap2-coverage.c:406-617) -----------------------------------------
// ---- prebuilt indexes: the reference's .mmi files (index.c:385-540) ----------------------------------
#define LQ_MMI_MAGIC "MMI\2"
#define LQ_MMI_BUCKET_BITS 14           // mm_idxopt_init (index.c:33); the option table has no switch for it
#define LQ_MMI_NO_SEQ 0x2

static void fwrite_or_throw(const void *p, size_t sz, size_t n, FILE *fp)
{
	if (n && fwrite(p, sz, n, fp) != n) throw std::runtime_error("write to the index dump failed");
}

// A prebuilt index brings its own k, w and -H (index.c:529-531), which the mapping then uses (lqmap.c:126-138), while the
// reference has already sized every query's counter array -- and with it the n of the count statistics -- from a sketch
// with the command-line values (minimap2-coverage.c:419-422, 552-563).  Keep those sizes, sketch the queries again with
// the index's values.  A query may then have more minimizers than counters; the reference writes past its array if a good
// chain matches one of the surplus minimizers (esterr.c:131-137; glibc aborts it on test data).  Here the array is long
// enough, the statistics look at the first n counters like the reference, and finish() refuses the run if a surplus
// counter was touched (outside the parity domain).
void lqcov_handle::adopt_index_params(i32 k, i32 w, i32 hpc)
{
	if (k == P.k && w == P.w && (hpc != 0) == (P.hpc != 0)) return;
	if (k < 1 || k > 28 || w < 1 || w > 255) throw std::domain_error("prebuilt index with k or w outside [1,28] / [1,255]");
	if (!parts.empty()) for (auto &p : parts) if (p && p->live) throw std::logic_error("index parameters change with live parts");
	std::vector<u64> old(q.n + 1, 0);
	if (have_queries && q.n) d2h(old.data(), q.moff.as<u64>(), q.n + 1, stream);
	P.k = k; P.w = w; P.hpc = hpc ? 1 : 0;
	mp.k = k; mp.w = w; mp.hpc = P.hpc;
	if (!have_queries) return;
	q.sketched = false;
	sketch(q, false);
	const u64 nm = q.n_mini;
	q_owner.ensure(nm * 4 + 4);
	if (nm) { LQ_LAUNCH(k_minimizer_owner, nblk(nm, 256), 256, stream, q.moff.as<u64>(), q.n, nm, q_owner.as<u32>()); check_launch(); }
	std::vector<u64> now(q.n + 1, 0), off(q.n + 1, 0);
	if (q.n) d2h(now.data(), q.moff.as<u64>(), q.n + 1, stream);
	h_nsize.resize(q.n);
	for (u32 i = 0; i < q.n; ++i) {
		h_nsize[i] = (u32)(old[i + 1] - old[i]);
		off[i + 1] = off[i] + std::max<u64>(h_nsize[i], now[i + 1] - now[i]);
	}
	cnt_total = off[q.n];
	cnt_off.ensure((q.n + 1) * 8); d_nsize.ensure((q.n + 1) * 4);
	h2d(cnt_off.as<u64>(), off.data(), q.n + 1, stream);
	h2d(d_nsize.as<u32>(), h_nsize.data(), q.n, stream);
	own_cnt_layout = true;
	cnts.ensure(cnt_total * 4 + 4);
	reset();
}

void lqcov_handle::build_part_from_host_minimizers(Part &pt, const std::vector<u64> &x, const std::vector<u64> &y,
                                                   std::vector<std::string> &&names, std::vector<u32> &&lens)
{
	ReadSetDev &rs = pt.rs;
	const u64 n = x.size();
	rs.key_stamp = 0;
	rs.mx.ensure(n * 8 + 8); rs.my.ensure(n * 8 + 8);
	h2d(rs.mx.as<u64>(), x.data(), n, stream);
	h2d(rs.my.as<u64>(), y.data(), n, stream);
	rs.n_mini = n; rs.n = (u32)lens.size();
	rs.h_len = std::move(lens);
	rs.names = std::move(names);
	rs.n_bases = 0;
	for (u32 v : rs.h_len) rs.n_bases += v;
	rs.d_len.ensure((rs.n + 1) * 4);
	h2d(rs.d_len.as<u32>(), rs.h_len.data(), rs.n, stream);
	LQ_HIP_CHECK(hipStreamSynchronize(stream));
	rs.sketched = true;
	build_index(pt);
}

// One part in the reference's on-disk layout.  What a reader needs is reproduced exactly (header, names with their
// one-byte length, per-bucket position arrays with every list ascending, key = minimizer>>b<<1 | singleton, value = the
// position or start<<32 | n, 4-bit sequences); the order of the (key, value) pairs inside a bucket is ascending key here,
// khash slot order in the reference -- mm_idx_load re-inserts them one by one, so the order is immaterial.
void lqcov_handle::dump_part(Part &pt, FILE *fp)
{
	if (!pt.built) throw std::logic_error("part not built");
	// (the distinct keys, their starts and counts are the build's work space, which lives with the handle: they describe the part
	// that was built last)
	if (ix_owner != &pt) throw std::logic_error("this part's index can no longer be dumped: another part was built after it (dump a part before the next lqcov_part_build)");
	ReadSetDev &rs = pt.rs;
	const u64 M = rs.n_mini, K = pt.n_keys;
	const u32 b = LQ_MMI_BUCKET_BITS, nb = 1u << b;
	std::vector<u64> hkey(K), hstart(K), hpos(M);
	std::vector<u32> hcnt(K);
	if (K) {
		d2h(hkey.data(), ix_ukey.as<u64>(), K, stream); d2h(hstart.data(), ix_ustart.as<u64>(), K, stream);
		d2h(hcnt.data(), ix_ucnt.as<u32>(), K, stream); d2h(hpos.data(), pt.pos.as<u64>(), M, stream);
	}
	u32 hdr[5] = {(u32)P.w, (u32)P.k, b, rs.n, (u32)(P.hpc ? 1 : 0)};
	fwrite_or_throw(LQ_MMI_MAGIC, 1, 4, fp);
	fwrite_or_throw(hdr, 4, 5, fp);
	u64 sum_len = 0;
	std::vector<u64> boff(rs.n + 1, 0);
	for (u32 i = 0; i < rs.n; ++i) {
		const u8 l = (u8)rs.names[i].size();                    // index.c:401: the length is stored in one byte
		fwrite_or_throw(&l, 1, 1, fp);
		fwrite_or_throw(rs.names[i].data(), 1, l, fp);
		fwrite_or_throw(&rs.h_len[i], 4, 1, fp);
		sum_len += rs.h_len[i]; boff[i + 1] = sum_len;
	}
	// keys by bucket (low b bits); hkey is ascending, so every bucket's keys stay ascending
	std::vector<u64> bstart(nb + 1, 0);
	for (u64 i = 0; i < K; ++i) ++bstart[(hkey[i] & (nb - 1)) + 1];
	for (u32 i = 0; i < nb; ++i) bstart[i + 1] += bstart[i];
	std::vector<u64> order(K), fill(bstart.begin(), bstart.end() - 1);
	for (u64 i = 0; i < K; ++i) order[fill[hkey[i] & (nb - 1)]++] = i;
	std::vector<u64> pbuf, kv;
	for (u32 bi = 0; bi < nb; ++bi) {
		pbuf.clear(); kv.clear();
		for (u64 t = bstart[bi]; t < bstart[bi + 1]; ++t) {
			const u64 i = order[t];
			const u64 key = hkey[i] >> b << 1;
			if (hcnt[i] == 1) { kv.push_back(key | 1); kv.push_back(hpos[hstart[i]]); }
			else {
				kv.push_back(key); kv.push_back((u64)pbuf.size() << 32 | hcnt[i]);
				pbuf.insert(pbuf.end(), hpos.begin() + hstart[i], hpos.begin() + hstart[i] + hcnt[i]);
			}
		}
		if (pbuf.size() > 0x7fffffffULL) throw std::domain_error("index bucket too large for the .mmi format");
		const i32 n_p = (i32)pbuf.size();
		const u32 size = (u32)(kv.size() / 2);
		fwrite_or_throw(&n_p, 4, 1, fp);
		fwrite_or_throw(pbuf.data(), 8, pbuf.size(), fp);
		fwrite_or_throw(&size, 4, 1, fp);
		fwrite_or_throw(kv.data(), 8, kv.size(), fp);
	}
	// mi->S
	const u64 n_words = (sum_len + 7) / 8;
	if (n_words) {
		if (rs.codes.p == nullptr) throw std::logic_error("this part has no sequences to dump (it was loaded from an index)");
		DBuf d_boff, d_out;
		d_boff.ensure((rs.n + 1) * 8); d_out.ensure(n_words * 4);
		h2d(d_boff.as<u64>(), boff.data(), rs.n + 1, stream);
		LQ_LAUNCH(k_seq4, nblk(n_words, 256), 256, stream, rs.codes.as<u64>(), rs.amb.as<u32>(), rs.d_coff.as<u64>(), d_boff.as<u64>(), rs.n, n_words, d_out.as<u32>());
		check_launch();
		std::vector<u32> hs(n_words);
		d2h(hs.data(), d_out.as<u32>(), n_words, stream);
		fwrite_or_throw(hs.data(), 4, n_words, fp);
	}
	fflush(fp);
}

static bool fread_exact(void *p, size_t sz, size_t n, FILE *fp) { return n == 0 || fread(p, sz, n, fp) == n; }

bool lqcov_handle::load_part(FILE *fp, Part &pt)
{
	char magic[4];
	if (fread(magic, 1, 4, fp) != 4) return false;              // end of file (index.c:436)
	if (memcmp(magic, LQ_MMI_MAGIC, 4) != 0) return false;
	u32 hdr[5];
	if (!fread_exact(hdr, 4, 5, fp)) throw std::runtime_error("truncated index file");
	const i32 w = (i32)hdr[0], k = (i32)hdr[1]; const u32 b = hdr[2], n_seq = hdr[3], flag = hdr[4];
	if (k != P.k || w != P.w || ((flag & 1) != 0) != (P.hpc != 0)) throw std::logic_error("index part with other k / w / -H than the handle's (adopt_index_params first)");
	if (b > 30) throw std::runtime_error("corrupt index file (bucket bits)");
	std::vector<std::string> names(n_seq);
	std::vector<u32> lens(n_seq);
	u64 sum_len = 0;
	for (u32 i = 0; i < n_seq; ++i) {
		u8 l;
		char buf[256];
		if (!fread_exact(&l, 1, 1, fp) || !fread_exact(buf, 1, l, fp) || !fread_exact(&lens[i], 4, 1, fp)) throw std::runtime_error("truncated index file");
		names[i].assign(buf, strnlen(buf, l));                     // the reference holds it as a C string (index.c:448-450)
		sum_len += lens[i];
	}
	std::vector<u64> x, y, pbuf;
	for (u32 bi = 0; bi < (1u << b); ++bi) {
		i32 n_p; u32 size;
		if (!fread_exact(&n_p, 4, 1, fp) || n_p < 0) throw std::runtime_error("truncated index file");
		pbuf.resize((size_t)n_p);
		if (!fread_exact(pbuf.data(), 8, (size_t)n_p, fp) || !fread_exact(&size, 4, 1, fp)) throw std::runtime_error("truncated index file");
		for (u32 j = 0; j < size; ++j) {
			u64 kv[2];
			if (!fread_exact(kv, 8, 2, fp)) throw std::runtime_error("truncated index file");
			const u64 minier = (kv[0] >> 1) << b | bi;             // index.c:69-86 read backwards
			if (kv[0] & 1) { x.push_back(minier << 8); y.push_back(kv[1]); }
			else {
				const u64 st = kv[1] >> 32, n = (u32)kv[1];
				if (st + n > (u64)n_p) throw std::runtime_error("corrupt index file (position list)");
				for (u64 t = 0; t < n; ++t) { x.push_back(minier << 8); y.push_back(pbuf[st + t]); }
			}
		}
	}
	if (!(flag & LQ_MMI_NO_SEQ)) {                                // mi->S: not used on this path; skipped
		const u64 bytes = (sum_len + 7) / 8 * 4;
		if (fseeko(fp, (off_t)bytes, SEEK_CUR) != 0) throw std::runtime_error("truncated index file");
	}
	build_part_from_host_minimizers(pt, x, y, std::move(names), std::move(lens));
	return true;
}

int lqcov_handle::run_files(const char *target, const char *query, FILE *out, FILE *log, const char *dump_path)
{
	bool is_idx = false;
	i32 ik = 0, iw = 0, ihpc = 0;
	{
		FILE *t = fopen(target, "rb");
		if (!t) throw std::runtime_error(std::string("failed to open file '") + target + "'");
		char magic[4]; u32 hdr[5];
		if (fread(magic, 1, 4, t) == 4 && memcmp(magic, LQ_MMI_MAGIC, 4) == 0) {       // mm_idx_is_idx (index.c:481-498)
			is_idx = true;
			if (fread(hdr, 4, 5, t) != 5) { fclose(t); throw std::runtime_error("truncated index file"); }
			iw = (i32)hdr[0]; ik = (i32)hdr[1]; ihpc = (i32)(hdr[4] & 1);
		}
		fclose(t);
	}
	if (is_idx && dump_path) throw std::domain_error("-d with a prebuilt index as the target is not supported");
	const double t_run0 = lq_now_s();
	// The queries (read, upload, sketch: minimap2-coverage.c:406-431) while the host makes the first part -- parse, page-locked
	// buffers, 2-bit packing, all host work; joined before anything of the targets touches the device.
	auto load_queries = [&]() {
		LQ_HIP_CHECK(hipSetDevice(device));
		FastxReader fq(query);
		ReadBatch qb;
		while (fq.read_minibatch(INT64_MAX, qb, true) > 0) {}
		const double tq = lq_now_s();
		set_queries(qb.size(), qb.seq.data(), qb.seq_off.data(), qb.any_qual ? qb.qual.data() : nullptr, qb.names.data(), qb.name_off.data());
		if (log) fprintf(log, "[lqcov] loaded %u query sequence(s), %" PRIu64 " bases, %" PRIu64 " minimizers (read %.3f s, upload + sketch %.3f s)\n", qb.size(), qb.bases(), q.n_mini, tq - t_run0, lq_now_s() - tq);
	};
	std::future<void> queries_loaded;
	if (query) {
#ifndef LQ_EMU
		if (!is_idx) queries_loaded = std::async(std::launch::async, load_queries);
		else load_queries();
#else
		load_queries();                                           // (the test emulator runs its kernels on the calling thread, one at a time)
#endif
	}
	struct QueriesJoin { std::future<void> &f; ~QueriesJoin() { if (f.valid()) { try { f.get(); } catch (...) {} } } } queries_join{queries_loaded};   // (never leave the task behind when something else throws)
	FILE *dump = nullptr;
	if (dump_path) { dump = fopen(dump_path, "wb"); if (!dump) throw std::runtime_error(std::string("failed to open file '") + dump_path + "'"); }
	struct Closer { FILE *f; ~Closer() { if (f) fclose(f); } } dump_closer{dump};
	int n_parts = 0;
	if (is_idx) {
		if (ik != P.k || iw != P.w || (ihpc != 0) != (P.hpc != 0)) {
			if (log) fprintf(log, "[WARNING] Indexing parameters (-k, -w or -H) overridden by parameters used in the prebuilt index.\n");
			adopt_index_params(ik, iw, ihpc);
		}
		FILE *fi = fopen(target, "rb");
		if (!fi) throw std::runtime_error(std::string("failed to open file '") + target + "'");
		Closer fi_closer{fi};
		for (;;) {
			parts.emplace_back(new Part());
			const int id = (int)parts.size() - 1;
			parts[id]->live = true;
			if (!load_part(fi, *parts[id])) { parts[id].reset(); break; }
			Part &pt = *parts[id];
			if (log) fprintf(log, "[lqcov] part %d (prebuilt): %u target sequence(s), %" PRIu64 " minimizers, %" PRIu64 " distinct, mid_occ = %d\n",
			                 n_parts, pt.rs.n, pt.rs.n_mini, pt.n_keys, mid_occ);
			if (query) map_part(pt);
			parts[id].reset();
			++n_parts;
		}
	} else {
		// One part = mini-batches of `chunk` bases while the running total is <= -I (index.c:244,311-316).  The parts go through a
		// pipeline of three stages that overlap: the host makes part k + 2 (a plain file: parsed by many threads from the mapping and
		// 2-bit packed into page-locked memory, fastx_mem.hpp; gzip or a pipe: the streaming reader, one thread like the
		// reference's kseq), the build stream uploads, sketches and indexes part k + 1, the lanes map part k.  (-d and the index-only
		// call keep one part at a time.)
		const int64_t chunk = (int)((u64)P.idx_mini_batch < P.batch_size ? (u64)P.idx_mini_batch : P.batch_size);   // index.c:316
		struct HostPart {
			bool packed = false, last = false;                         // last: the input ended inside this part
			std::vector<ReadBatch> bs;                                 // streaming form (ASCII)
			u64 *codes = nullptr; u32 *amb = nullptr; u64 cap_words = 0;   // packed form, page-locked
			std::vector<u32> lens; std::vector<char> names; std::vector<u64> name_off{0};
			u32 n = 0; u64 bases = 0;
			std::atomic<bool> any_amb{false};                           // packed form: a read of the part holds an ambiguous base
			bool empty() const { return packed ? n == 0 : bs.empty(); }
			~HostPart() { if (codes) hipHostFree(codes); if (amb) hipHostFree(amb); }
		};
		HostPart hp[2];
		MemFastx mf;
		MemRecords recs;
		struct FlatRec { const u8 *name; u32 name_len; const u8 *seq; u32 seq_len; };
		std::vector<FlatRec> flat;
		std::vector<std::pair<size_t, size_t>> ranges;                 // records of every part (memory path)
		size_t next_range = 0;
		std::unique_ptr<FastxReader> ft;
		bool mem = K.parse_threads != 1 && mf.open(target);
		const double t_parse0 = lq_now_s();
		if (mem) {
			// (records of several lines are copied out of the mapping: a wrapped FASTA of many gigabases would be held twice, so
			// past LQCOV_PARSE_SIDE bytes of copies the file is streamed like a gzip instead, one part in memory at a time)
			lq_parse_all(mf, K.parse_threads, K.parse_piece, recs, K.parse_side);
			if (recs.too_wrapped) { mem = false; mf.close(); if (log) fprintf(log, "[lqcov] target records span several lines: streaming reader\n"); }
		}
		if (mem) {
			const double t0 = t_parse0;
			flat.reserve(recs.n_recs);
			for (MemPiece &pc : recs.pieces) for (MemRec &r : pc.recs)
				flat.push_back(FlatRec{mf.data() + r.name_off, r.name_len, (r.own ? pc.side.data() : mf.data()) + r.seq_off, r.seq_len});
			size_t r0 = 0;
			while (r0 < flat.size()) {                                 // index.c:244,311-316 and bseq.c:86-98 on the record lengths
				u64 sum_len = 0; size_t r1 = r0;
				while (r1 < flat.size() && sum_len <= P.batch_size) {
					i64 size = 0;
					while (r1 < flat.size()) { size += flat[r1].seq_len; ++r1; if (size >= chunk) break; }
					sum_len += (u64)size;
				}
				ranges.emplace_back(r0, r1);
				r0 = r1;
			}
			if (log) fprintf(log, "[lqcov] parsed %zu target sequence(s) from the mapped file in %.3f s (%zu pieces, %u parsed again in order), %zu part(s)\n",
			                 flat.size(), lq_now_s() - t0, recs.pieces.size(), recs.reparsed, ranges.size());
		} else ft.reset(new FastxReader(target));
		double t_host[2] = {0, 0}, t_alloc[2] = {0, 0};
		auto produce = [&](int slot) {
			LQ_HIP_CHECK(hipSetDevice(device));                         // (may run on a thread of its own: page-locked allocations)
			const double tp0 = lq_now_s();
			struct Tm { double &t; double t0; ~Tm() { t = lq_now_s() - t0; } } tm{t_host[slot], tp0};
			HostPart &h = hp[slot];
			h.bs.clear(); h.n = 0; h.bases = 0; h.lens.clear(); h.names.clear(); h.name_off.assign(1, 0); h.last = false; h.any_amb.store(false);
			if (!mem) {
				h.packed = false;
				u64 sum_len = 0;
				for (;;) {
					if (sum_len > P.batch_size) break;
					h.bs.emplace_back();
					if (ft->read_minibatch(chunk, h.bs.back(), false) == 0) { h.bs.pop_back(); h.last = true; break; }
					sum_len += h.bs.back().bases();
				}
				for (ReadBatch &b : h.bs) h.bases += b.bases();
				return;
			}
			h.packed = true;
			if (next_range >= ranges.size()) { h.last = true; return; }
			const size_t r0 = ranges[next_range].first, r1 = ranges[next_range].second;
			++next_range;
			h.last = next_range >= ranges.size();
			h.n = (u32)(r1 - r0);
			h.lens.resize(h.n);
			std::vector<u64> coff(h.n + 1, 0);
			u64 name_bytes = 0;
			for (u32 i = 0; i < h.n; ++i) { const FlatRec &r = flat[r0 + i]; h.lens[i] = r.seq_len; coff[i + 1] = coff[i] + ((u64)r.seq_len + LQ_CHUNK - 1) / LQ_CHUNK; h.bases += r.seq_len; name_bytes += r.name_len + 1; }
			const u64 n_words = coff[h.n] * LQ_CHUNK_WORDS;
			const double ta0 = lq_now_s();
			if (n_words > h.cap_words) {
				if (h.codes) hipHostFree(h.codes);
				if (h.amb) hipHostFree(h.amb);
				h.codes = nullptr; h.amb = nullptr;
				h.cap_words = n_words + n_words / 16 + 64;
				LQ_HIP_CHECK(hipHostMalloc((void**)&h.codes, h.cap_words * 8, 0));
				LQ_HIP_CHECK(hipHostMalloc((void**)&h.amb, h.cap_words * 4, 0));
				t_alloc[slot] = lq_now_s() - ta0;
			}
			h.names.resize(name_bytes); h.name_off.resize(h.n + 1);
			{ u64 o = 0; for (u32 i = 0; i < h.n; ++i) { const FlatRec &r = flat[r0 + i]; h.name_off[i] = o; memcpy(h.names.data() + o, r.name, r.name_len); h.names[o + r.name_len] = 0; o += r.name_len + 1; } h.name_off[h.n] = o; }
			// pack, read by read, equal shares of bases per thread
			int nt = K.parse_threads > 0 ? K.parse_threads : (int)std::min<unsigned>(64, std::max(1u, std::thread::hardware_concurrency()));
			std::atomic<u32> next(0);
			auto work = [&]() {
				for (;;) {
					const u32 i0 = next.fetch_add(64);
					if (i0 >= h.n) break;
					for (u32 i = i0; i < std::min<u32>(i0 + 64, h.n); ++i) {
						const FlatRec &r = flat[r0 + i];
						const u64 off[2] = {0, r.seq_len};
						lq_pack_host(1, r.seq, off, h.codes + coff[i] * LQ_CHUNK_WORDS, h.amb + coff[i] * LQ_CHUNK_WORDS, 1);
						if (!h.any_amb.load(std::memory_order_relaxed) && lq_packed_read_ambiguous(h.amb + coff[i] * LQ_CHUNK_WORDS, r.seq_len)) h.any_amb.store(true);
					}
				}
			};
			if (nt <= 1 || h.n < 256) work();
			else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work); for (auto &t : th) t.join(); }
		};
		int n_built = 0;
		auto build = [&](int slot, Part &pt) {                      // upload + sketch + index of hp[slot] (the build stream)
			HostPart &h = hp[slot];
			{	// (a Part object is used again: what lqcov_part_clear does)
				ReadSetDev &rs = pt.rs;
				rs.n = 0; rs.n_chunks = 0; rs.n_bases = 0; rs.n_mini = 0; rs.sketched = false; rs.dp_n = 0; rs.dp_tiles = 0;
				rs.h_coff.assign(1, 0); rs.h_len.clear(); rs.names.clear();
				pt.built = false; pt.n_keys = 0; pt.live = true;
			}
			const double tb0 = lq_now_s();
			if (h.packed) add_reads_packed(pt.rs, h.n, h.codes, h.any_amb.load() || K.upload_amb ? h.amb : nullptr, h.lens.data(), h.names.data(), h.name_off.data());   // (a part without an N: the codes alone cross PCIe)
			else for (ReadBatch &tb : h.bs) { add_reads(pt.rs, tb.size(), tb.seq.data(), tb.seq_off.data(), tb.names.data(), tb.name_off.data()); tb = ReadBatch(); }
			const double tu = lq_now_s();
			sketch(pt.rs, true);
			const double ts = lq_now_s();
			build_index(pt);
			if (log) fprintf(log, "[lqcov] part %d: %u target sequence(s), %" PRIu64 " bases, %" PRIu64 " minimizers, %" PRIu64 " distinct, mid_occ = %d  (host %.3f s of which page-locked allocation %.3f, upload %.3f s, sketch %.3f s, index %.3f s)\n",
			                 n_built, pt.rs.n, pt.rs.n_bases, pt.rs.n_mini, pt.n_keys, mid_occ, t_host[slot], t_alloc[slot], tu - tb0, ts - tu, lq_now_s() - ts);
			++n_built;
			if (dump) dump_part(pt, dump);                          // mm_idx_reader_read (index.c:533)
		};
#ifndef LQ_EMU
		const bool pipeline = K.pipeline && query && !dump && profiling != 1;
#else
		const bool pipeline = false;                                // (the test emulator runs one kernel at a time, on the calling thread)
#endif
		parts.emplace_back(new Part()); parts.emplace_back(new Part());
		Part *dev[2] = { parts[parts.size() - 2].get(), parts[parts.size() - 1].get() };
		produce(0);
		if (queries_loaded.valid()) queries_loaded.get();          // (a query file that cannot be read: reported here)
		if (!hp[0].empty()) {
			std::future<void> fut_h;
			if (!hp[0].last) fut_h = std::async(std::launch::async, produce, 1);
			build(0, *dev[0]);
			// the lanes size their work space from what is free when the first part stands: leave room for the part that is built meanwhile
			if (pipeline && !hp[0].last && hbm_reserve == 0) hbm_reserve = (u64)((double)std::min<u64>(hp[0].bases, P.batch_size + (u64)chunk) * 9.5);
			for (int k = 0;; ++k) {
				const int cur = k & 1, nxt = cur ^ 1;
				std::future<bool> next_ready;
				auto advance = [&, k, nxt]() -> bool {                  // part k + 1 from the host, then the host starts on part k + 2
					LQ_HIP_CHECK(hipSetDevice(device));
					if (!fut_h.valid()) return false;
					fut_h.get();
					if (hp[nxt].empty()) return false;
					build(nxt, *dev[nxt]);
					if (!hp[nxt].last) fut_h = std::async(std::launch::async, produce, nxt ^ 1);   // (slot of part k: uploaded long ago)
					return true;
				};
				if (pipeline) next_ready = std::async(std::launch::async, advance);
				if (query) {
					const double tm0 = lq_now_s();
					map_part(*dev[cur]);
					if (log) fprintf(log, "[lqcov] part %d: mapped %u queries in %.3f s, %" PRIu64 " anchors (%" PRIu64 " written; so far %" PRIu64 " runs of %" PRIu64 " queries chained in klib's order, %" PRIu64 " anchors)\n",
					                 n_parts, q.n, lq_now_s() - tm0, last_n_anchors, last_n_written, (u64)stat_sens_runs, (u64)stat_p2_queries, (u64)stat_p2_anchors);
				}
				++n_parts;
				const bool more = pipeline ? next_ready.get() : advance();
				if (!more) break;
			}
			if (fut_h.valid()) fut_h.get();
		}
		for (Part *d : dev) for (auto &up : parts) if (up.get() == d) up.reset();
	}
	if (!query) return 0;                                         // index only (minimap2-coverage.c:460-468)
	const double tf0 = lq_now_s();
	finish();
	write_table(out);
	if (log) fprintf(log, "[lqcov] rows and table in %.3f s; the whole call %.3f s (device allocations of this process so far: %.3f s for %.1f GB)\n", lq_now_s() - tf0, lq_now_s() - t_run0,
	                 (double)lq_alloc_ns * 1e-9, (double)lq_alloc_bytes * 1e-9);
	// A uint16 match counter that reaches 65535 makes the reference's result depend on the order in which it processed the
	// chains (esterr.c:130,136 test a[st], not a[j]).  Such a query's counters were replayed in that order (sat_replay_part) and its
	// row is the reference's; only counters that reached the limit after being summed over ranks (accum_import) cannot be, and
	// then the run says so and does not report success.
	u32 n_sat = 0, n_rep = 0;
	for (u32 i = 0; i < q.n; ++i) if (rows[i].flags & LQCOV_ROW_SATURATED) {
		if (rows[i].flags & LQCOV_ROW_REPLAYED) { ++n_rep; continue; }
		if (log && n_sat < 10) fprintf(log, "[WARNING] query %s: a match counter reached 65535 (esterr.c:130,136) in counters merged from several ranks: its row is not guaranteed to equal the reference's\n", q.names[q_inv[i]].c_str());
		++n_sat;
	}
	if (log && n_rep) fprintf(log, "[lqcov] %u quer%s with a match counter at its 16-bit limit (esterr.c:130,136): %llu chains replayed in the reference's order\n", n_rep, n_rep == 1 ? "y" : "ies", (unsigned long long)stat_sat_chains);
	if (n_sat) throw std::domain_error(std::to_string(n_sat) + " quer" + (n_sat == 1 ? "y" : "ies") + " with a saturated uint16 match counter (esterr.c:130,136) in merged counters: table written, rows flagged LQCOV_ROW_SATURATED without LQCOV_ROW_REPLAYED are not guaranteed");
	return 0;
}

// ---- tests: the device-wide primitives on host arrays (lqcov_debug_sort_pairs / lqcov_debug_scan) ----------------------------------
void lqcov_handle::debug_sort_pairs(u64 *keys, u64 *vals, u64 n, unsigned bits, int key_bytes)
{
	if (key_bytes != 4 && key_bytes != 8) throw std::invalid_argument("key_bytes must be 4 or 8");
	if (!vals && key_bytes != 4) throw std::invalid_argument("keys-only sorts have 4-byte keys");
	DBuf ki, ko, vi, vo;
	ki.ensure(n * 8 + 8); ko.ensure(n * 8 + 8); vi.ensure(n * 8 + 8); vo.ensure(n * 8 + 8);
	if (key_bytes == 4) {
		std::vector<u32> k32(n);
		for (u64 i = 0; i < n; ++i) k32[i] = (u32)keys[i];
		h2d(ki.as<u32>(), k32.data(), n, stream);
		if (vals) { h2d(vi.as<u64>(), vals, n, stream); prim.sort_pairs_u32_u64(ki.as<u32>(), ko.as<u32>(), vi.as<u64>(), vo.as<u64>(), n, bits); }
		else prim.sort_keys_u32(ki.as<u32>(), ko.as<u32>(), n);
		d2h(k32.data(), ko.as<u32>(), n, stream);
		for (u64 i = 0; i < n; ++i) keys[i] = k32[i];
	} else {
		h2d(ki.as<u64>(), keys, n, stream); h2d(vi.as<u64>(), vals, n, stream);
		prim.sort_pairs_u64(ki.as<u64>(), ko.as<u64>(), vi.as<u64>(), vo.as<u64>(), n, bits);
		d2h(keys, ko.as<u64>(), n, stream);
	}
	if (vals) d2h(vals, vo.as<u64>(), n, stream);
	LQ_HIP_CHECK(hipStreamSynchronize(stream));
}

void lqcov_handle::debug_scan(const u32 *in, u64 *out, u64 n)
{
	DBuf di, dout;
	di.ensure(n * 4 + 4); dout.ensure(n * 8 + 8);
	h2d(di.as<u32>(), in, n, stream);
	prim.exclusive_scan_u32_u64(di.as<u32>(), dout.as<u64>(), n);
	d2h(out, dout.as<u64>(), n, stream);
	LQ_HIP_CHECK(hipStreamSynchronize(stream));
}
