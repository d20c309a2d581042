// longqc_amd/csrc/api.cpp -- the C ABI of liblqcov.so (include/lqcov.h): argument handling of the
// reference's option table (minimap2-coverage.c:63-197, defaults :229-388) and thin extern "C"
// wrappers that turn C++ exceptions into status codes.
#include "engine.hpp"
#include "fastx.hpp"
#include "fastx_mem.hpp"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <stdexcept>
#include <ios>
#include <algorithm>
#include <vector>

static thread_local std::string g_create_error;
// live handles per device: only the last one to go hands the pool's cache back (lqcov_destroy)
static std::mutex g_live_mu;
static std::map<int, int> g_live;

// (one thread may build a part while another maps: both may fail, the message is written under the handle's lock)
static void set_err(lqcov_handle *h, const char *what) { std::lock_guard<std::mutex> g(h->stage_mu); h->err = what; }
template <class F> static int guard(lqcov_handle *h, F &&f)
{
	if (!h) return LQCOV_E_ARG;
	try { f(); return 0; }
	catch (const std::invalid_argument &e) { set_err(h, e.what()); return LQCOV_E_ARG; }
	catch (const std::domain_error &e) { set_err(h, e.what()); return LQCOV_E_DOMAIN; }
	catch (const std::logic_error &e) { set_err(h, e.what()); return LQCOV_E_STATE; }
	catch (const std::ios_base::failure &e) { set_err(h, e.what()); return LQCOV_E_IO; }
	catch (const std::runtime_error &e) {
		set_err(h, e.what());
		return strstr(e.what(), "failed to open") ? LQCOV_E_IO : LQCOV_E_DEVICE;
	}
	catch (const std::exception &e) { set_err(h, e.what()); return LQCOV_E_DEVICE; }
}

extern "C" {

int lqcov_abi_version(void) { return LQCOV_ABI_VERSION; }

void lqcov_params_default(lqcov_params *p)
{
	memset(p, 0, sizeof(*p));
	p->k = 12; p->w = 5; p->hpc = 0;                         // minimap2-coverage.c:252-266
	p->batch_size = 4000000000ULL; p->idx_mini_batch = 50000000;   // index.c:35-36
	p->max_gap = 10000; p->min_cnt = 3; p->min_chain_score = 40;   // :302-321
	p->min_score_med = 40; p->min_score_good = 40;           // :324-332
	p->max_chain_skip = 25; p->bw = 500;                     // :362-367, map.c:20
	p->max_overhang = 2000; p->min_ovlp = 1000; p->min_coverage = 3; p->min_ratio = 0.4;   // :290-295, :369-388
	p->mid_occ_frac = 2e-4f;                                 // map.c:16
	p->no_self = 0; p->ava = 0; p->filter_flag = 0; p->n_threads = 1;
}

static int64_t parse_num(const char *str)                   // mm_parse_num (minimap2-coverage.c:22-31)
{
	char *p;
	double x = strtod(str, &p);
	if (*p == 'G' || *p == 'g') x *= 1e9;
	else if (*p == 'M' || *p == 'm') x *= 1e6;
	else if (*p == 'K' || *p == 'k') x *= 1e3;
	return (int64_t)(x + .499);
}

struct LongOpt { const char *name; char key; };
static const LongOpt kLong[] = {                            // minimap2-coverage.c:166-195
	{"homopolymer", 'H'}, {"k-mer", 'k'}, {"window", 'w'}, {"index-size", 'I'}, {"dump-index", 'd'},
	{"max-gap-length", 'g'}, {"min-cnt", 'n'}, {"min-score", 'm'}, {"min-score-t2", 'p'}, {"min-score-t3", 'q'},
	{"max-chain-skip", 's'}, {"skip-self-ava", 'X'}, {"skip-self", 'Y'}, {"max-overhang", 'a'},
	{"min-overlap-len", 'l'}, {"min-coverage", 'c'}, {"min-overlap-ratio", 'r'}, {"num-subset", 'u'},
	{"threads", 't'}, {"minimizer-cnt", 'z'}, {"filter", 'f'},
};

int lqcov_parse_args(int argc, const char *const *argv, lqcov_params *p, const char **target, const char **query,
                     const char **dump_path, char *errbuf, size_t errbuf_len)
{
	// "0 / -1 means default" exactly like the reference's ARGP_KEY_INIT block (minimap2-coverage.c:150-160)
	int k = 0, w = 0, max_gap = 0, min_cnt = 0, m = 0, pm = 0, qg = 0, skip = -1, ohang = -1, ovlp = -1, cov = -1;
	double ratio = 0.0;
	uint64_t I = 0;
	int X = 0, Y = 0, H = 0, f = 0, threads = 1;
	const char *pos[2] = {nullptr, nullptr}, *dump = nullptr;
	int npos = 0;
	auto fail = [&](const std::string &msg) { if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", msg.c_str()); return LQCOV_E_ARG; };
	auto apply = [&](char key, const char *v) -> bool {
		switch (key) {
		case 'H': H = 1; return true;
		case 'X': X = 1; return true;
		case 'Y': Y = 1; return true;
		case 'f': f = 1; return true;
		case 'z': return true;
		case 'k': k = atoi(v); return true;
		case 'w': w = atoi(v); return true;
		case 'I': I = (uint64_t)parse_num(v); return true;
		case 'd': dump = v; return true;
		case 'g': max_gap = atoi(v); return true;
		case 'n': min_cnt = atoi(v); return true;
		case 'm': m = atoi(v); return true;
		case 'p': pm = atoi(v); return true;
		case 'q': qg = atoi(v); return true;
		case 's': skip = atoi(v); return true;
		case 'a': ohang = atoi(v); return true;
		case 'l': ovlp = atoi(v); return true;
		case 'c': cov = atoi(v); return true;
		case 'r': ratio = atof(v); return true;
		case 'u': return true;
		case 't': threads = atoi(v); return true;
		}
		return false;
	};
	const std::string flags = "HXYfz";
	for (int i = 1; i < argc; ++i) {
		const char *a = argv[i];
		if (a[0] == '-' && a[1] == '-' && a[2]) {
			std::string name(a + 2), val;
			bool has_val = false;
			size_t eq = name.find('=');
			if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
			char key = 0;
			for (const LongOpt &lo : kLong) if (name == lo.name) key = lo.key;
			if (!key) return fail("unrecognized option '--" + name + "'");
			if (flags.find(key) != std::string::npos) { apply(key, nullptr); continue; }
			if (!has_val) { if (i + 1 >= argc) return fail("option '--" + name + "' requires an argument"); val = argv[++i]; }
			apply(key, val.c_str());
		} else if (a[0] == '-' && a[1]) {
			for (int j = 1; a[j]; ++j) {
				char key = a[j];
				if (flags.find(key) != std::string::npos) { apply(key, nullptr); continue; }
				const char *v = a[j + 1] ? &a[j + 1] : (i + 1 < argc ? argv[++i] : nullptr);
				if (!v) return fail(std::string("option requires an argument -- '") + key + "'");
				if (!apply(key, v)) return fail(std::string("invalid option -- '") + key + "'");
				break;
			}
		} else {
			if (npos >= 2) return fail("too many arguments");
			pos[npos++] = a;
		}
	}
	if (!dump && npos < 2) return fail("not enough arguments: <target.seqs> <query.seqs>");
	lqcov_params_default(p);
	if (X && Y) return fail("Error: -X and -Y are mutually exclusive");
	if (!X && !Y && !dump) return fail("Error: Choose either -X (all-vs-all) or -Y (all-vs-sub)");
	if (X) { p->no_self = 1; p->ava = 1; } else if (Y) { p->no_self = 1; }
	p->hpc = H;
	if (k) p->k = k;
	if (w) p->w = w;
	if (I) p->batch_size = I;
	if (cov != -1) p->min_coverage = cov;
	if (max_gap) p->max_gap = max_gap;
	if (min_cnt) p->min_cnt = min_cnt;
	if (m) p->min_chain_score = m;
	p->min_score_med = pm ? pm : p->min_chain_score;
	p->min_score_good = qg ? qg : p->min_chain_score;
	if (p->min_score_med < p->min_chain_score) return fail("Error: -p must be larger than or equal to -m.");
	if (p->min_score_good < p->min_chain_score || p->min_score_good < p->min_score_med) return fail("Error: -q must be larger than or equal to -m and -p.");
	if (skip != -1) p->max_chain_skip = skip;
	if (ohang != -1) p->max_overhang = ohang;
	if (ovlp != -1) p->min_ovlp = ovlp;
	if (ratio != 0.0) p->min_ratio = ratio;
	p->filter_flag = f;
	p->n_threads = threads;
	if (target) *target = pos[0];
	if (query) *query = pos[1];
	if (dump_path) *dump_path = dump;
	return 0;
}

#ifndef LQ_EMU
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>
// LQCOV_SEGV_TRACE=1 (diagnostics, like LQCOV_TRACE_LAUNCHES): the native stack of a thread that faults goes to stderr before
// the default action takes over -- a fault inside a lane's thread otherwise shows up as the caller waiting in lqcov_part_map
static int lq_segv_fd = 2;                                // LQCOV_SEGV_TRACE=<path>: the report goes there (a test runner may have captured fd 2); =1: stderr
static void lq_segv_trace(int sig, siginfo_t *si, void *)
{
	void *bt[64];
	char msg[160];
	const int m = snprintf(msg, sizeof(msg), "[lqcov] fatal signal %d at address %p, native stack of the faulting thread:\n", sig, si ? si->si_addr : nullptr);
	(void)!write(lq_segv_fd, msg, (size_t)m);
	const int n = backtrace(bt, 64);
	backtrace_symbols_fd(bt, n, lq_segv_fd);
	signal(sig, SIG_DFL);
	raise(sig);
}
void lq_segv_altstack()                                   // (per thread: the handler must run even when the thread's own stack is the problem)
{
	if (!getenv("LQCOV_SEGV_TRACE")) return;
	static thread_local char alt[1 << 16];
	stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
	sigaltstack(&ss, nullptr);
}
static void lq_segv_install()
{
	const char *e = getenv("LQCOV_SEGV_TRACE");
	if (e && e[0] == '/' && lq_segv_fd == 2) { const int fd = open(e, O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) lq_segv_fd = fd; }
	struct sigaction sa; memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = lq_segv_trace; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
	sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr); sigaction(SIGABRT, &sa, nullptr);
	lq_segv_altstack();
}
#endif

lqcov_handle *lqcov_create(const lqcov_params *p, int device)
{
#ifndef LQ_EMU
	if (getenv("LQCOV_SEGV_TRACE")) lq_segv_install();
#endif
	// HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share a queue run
	// one after the other: with the default never more than four of the lanes' kernels run at a time (rocprofv3 kernel trace,
	// configs[2]).  Eight queues: 1.71-1.73 s per step against 1.74-1.77 (16: the same).  Only effective when the HIP runtime
	// has not started yet in this process (the CLI, LongQC's exec); hosts that initialise HIP first set it themselves (bench.py).
	// LQCOV_HW_QUEUES=n asks for n instead; LQCOV_HW_QUEUES=0 leaves the host process's environment alone (a host that shares
	// the HIP runtime with other users -- torch in LqCovExec's in-process mode -- decides for itself).  A value already in the
	// environment is never overwritten.
	{
		const char *hq = getenv("LQCOV_HW_QUEUES");
		if (!hq || atoi(hq) > 0) setenv("GPU_MAX_HW_QUEUES", hq && atoi(hq) > 0 ? hq : "8", 0);
	}
	try { lqcov_handle *h = new lqcov_handle(*p, device); { std::lock_guard<std::mutex> lk(g_live_mu); ++g_live[device]; } return h; }
	catch (const std::exception &e) { g_create_error = e.what(); fprintf(stderr, "lqcov_create: %s\n", e.what()); return nullptr; }
}
void lqcov_destroy(lqcov_handle *h)
{
	if (!h) return;
#ifndef LQ_EMU
	const int dev = h->device;
#endif
	delete h;
#ifndef LQ_EMU
	bool last = false;
	{ std::lock_guard<std::mutex> lk(g_live_mu); auto it = g_live.find(dev); if (it != g_live.end() && it->second > 0 && --it->second == 0) last = true; }
	if (!last) return;                                          // (another handle lives on this device: its streams are not to be stalled, nor its cached blocks dropped)
	// the lanes' work space came from HIP's stream-ordered pool, whose release threshold the handle raised (blocks given back
	// stay cached): with the handle gone the cache goes back to the device -- the next handle (or another user of the GPU)
	// would otherwise find the memory taken and pay for the trim in the middle of its first part (5 s for 170 GB, measured)
	hipMemPool_t pool = nullptr;
	if (hipSetDevice(dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) { (void)hipDeviceSynchronize(); (void)hipMemPoolTrimTo(pool, 0); }
#endif
}
const char *lqcov_last_error(const lqcov_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }
int lqcov_set_profiling(lqcov_handle *h, int on)
{
	if (!h) return LQCOV_E_ARG;
	h->drain_stages();
	h->profiling = on < 0 ? 0 : on > 2 ? 2 : on;
	if (!on) { h->stages.clear(); h->stage_order.clear(); }
	return 0;
}
int lqcov_set_debug(lqcov_handle *h, unsigned flags) { if (!h) return LQCOV_E_ARG; h->debug_flags = flags; return 0; }

int lqcov_set_profiling_only(lqcov_handle *h, const char *stage)
{
	if (!h) return LQCOV_E_ARG;
	h->profile_only = stage ? stage : "";
	return 0;
}

int lqcov_get_stage_times(lqcov_handle *h, lqcov_stage_time *out, int max_out)
{
	if (!h) return LQCOV_E_ARG;
	h->drain_stages();
	int n = 0;
	for (const std::string &name : h->stage_order) {
		if (n >= max_out) break;
		const StageAcc &s = h->stages[name];
		memset(&out[n], 0, sizeof(out[n]));
		snprintf(out[n].name, sizeof(out[n].name), "%s", name.c_str());
		out[n].total_ms = s.ms; out[n].launches = s.launches; out[n].algo_bytes = s.bytes;
		++n;
	}
	return n;
}

int lqcov_set_queries(lqcov_handle *h, uint32_t n, const uint8_t *seq, const uint64_t *seq_off, const uint8_t *qual, const char *names, const uint64_t *name_off)
{
	return guard(h, [&] { if (!seq_off || (n && !seq)) throw std::invalid_argument("null read buffers"); h->set_queries(n, seq, seq_off, qual, names, name_off); });
}

int lqcov_part_begin(lqcov_handle *h)
{
	int id = -1;
	int rc = guard(h, [&] { h->parts.emplace_back(new Part()); id = (int)h->parts.size() - 1; h->parts[id]->live = true; });
	return rc ? rc : id;
}

int lqcov_part_add_targets(lqcov_handle *h, int part, uint32_t n, const uint8_t *seq, const uint64_t *seq_off, const char *names, const uint64_t *name_off)
{
	return guard(h, [&] {
		if (!seq_off || (n && !seq)) throw std::invalid_argument("null read buffers");
		Part &pt = h->part(part);
		if (pt.built) throw std::logic_error("part already built");
		h->add_reads(pt.rs, n, seq, seq_off, names, name_off);
	});
}

// ---- packed reads: the host side of the upload (the parser thread packs, 0.375 B per base cross PCIe) ----
uint64_t lqcov_packed_chunks(uint32_t n, const uint64_t *seq_off) { return seq_off ? lq_packed_chunks(n, seq_off) : 0; }

int lqcov_pack_reads(uint32_t n, const uint8_t *seq, const uint64_t *seq_off, uint64_t *codes, uint32_t *amb, int n_threads)
{
	if (!seq_off || (n && (!seq || !codes || !amb))) return LQCOV_E_ARG;
	try { lq_pack_host(n, seq, seq_off, codes, amb, n_threads); } catch (...) { return LQCOV_E_DEVICE; }
	return 0;
}

void *lqcov_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, 0) != hipSuccess) return nullptr;
	return p;
}
void lqcov_host_free(void *p) { if (p) hipHostFree(p); }

int lqcov_packed_ambiguous_reads(uint32_t n, const uint32_t *amb, const uint32_t *lens, uint8_t *flags)
{
	if (n && (!amb || !lens || !flags)) return LQCOV_E_ARG;
	uint64_t c = 0;
	for (uint32_t i = 0; i < n; ++i) { flags[i] = lq_packed_read_ambiguous(amb + c * LQ_CHUNK_WORDS, lens[i]) ? 1 : 0; c += ((uint64_t)lens[i] + LQ_CHUNK - 1) / LQ_CHUNK; }
	return 0;
}

int lqcov_part_add_packed(lqcov_handle *h, int part, uint32_t n, const uint64_t *codes, const uint32_t *amb, const uint32_t *lens,
                          const char *names, const uint64_t *name_off)
{
	return guard(h, [&] {
		if (n && (!codes || !lens)) throw std::invalid_argument("null read buffers");   // (amb == NULL: no read holds an ambiguous base)
		Part &pt = h->part(part);
		if (pt.built) throw std::logic_error("part already built");
		h->add_reads_packed(pt.rs, n, codes, amb, lens, names, name_off);
	});
}

int lqcov_part_add_packed_shares_dev(lqcov_handle *h, int part, const uint64_t *codes_dev, const uint32_t *amb_dev, uint64_t stride_chunks,
                                     uint32_t n_shares, const uint64_t *share_chunks, uint32_t n, const uint32_t *lens,
                                     const char *names, const uint64_t *name_off)
{
	return guard(h, [&] {
		if (n && (!codes_dev || !lens || !share_chunks)) throw std::invalid_argument("null read buffers");   // (amb_dev == NULL: no read holds an ambiguous base)
		Part &pt = h->part(part);
		if (pt.built) throw std::logic_error("part already built");
		if (pt.rs.n) throw std::logic_error("packed shares go into an empty part");
		std::vector<u64> sc(share_chunks, share_chunks + n_shares);
		for (u64 v : sc) if (v > stride_chunks) throw std::invalid_argument("a share is longer than the stride");
		h->add_reads_packed(pt.rs, n, nullptr, nullptr, lens, names, name_off, codes_dev, amb_dev, stride_chunks, &sc);
	});
}

int lqcov_part_clear(lqcov_handle *h, int part)
{
	return guard(h, [&] {
		Part &pt = h->part(part);
		ReadSetDev &rs = pt.rs;
		rs.n = 0; rs.n_chunks = 0; rs.n_bases = 0; rs.n_mini = 0; rs.sketched = false; rs.dp_n = 0; rs.dp_tiles = 0;
		rs.h_coff.assign(1, 0); rs.h_len.clear(); rs.names.clear();
		pt.built = false; pt.n_keys = 0;
	});
}

int lqcov_part_build(lqcov_handle *h, int part) { return guard(h, [&] { h->build_part(h->part(part)); }); }
int lqcov_part_map(lqcov_handle *h, int part) { return guard(h, [&] { h->map_part(h->part(part)); }); }
int lqcov_part_release(lqcov_handle *h, int part) { return guard(h, [&] { h->part(part); h->parts[part].reset(); }); }
int lqcov_reset(lqcov_handle *h) { return guard(h, [&] { if (!h->have_queries) throw std::logic_error("no queries"); h->reset(); }); }
int lqcov_sync(lqcov_handle *h) { return guard(h, [&] { LQ_HIP_CHECK(hipStreamSynchronize(h->bstream)); LQ_HIP_CHECK(hipStreamSynchronize(h->stream)); }); }
int lqcov_reserve_hbm(lqcov_handle *h, uint64_t bytes) { if (!h) return LQCOV_E_ARG; h->hbm_reserve = bytes; return 0; }
int lqcov_workspace_trim(lqcov_handle *h)
{
	return guard(h, [&] {
		LQ_HIP_CHECK(hipDeviceSynchronize());
		// the lanes' work space (sized for ~80 % of the free HBM at the first batch) is held in grow-only buffers: hand it back,
		// then the blocks the stream-ordered pool has cached.  Everything regrows on the next part_map.
		for (auto &L : h->lanes) L->release_buffers();
		for (DBuf *b : { &h->ix_key, &h->ix_key2, &h->ix_head, &h->ix_uidx, &h->ix_sorted }) b->release();
		h->ix_key_stamp = 0;
		{	// (the seed filter's bucket buffer and tables: as large as a gigabyte-scale chunk of records)
			std::lock_guard<std::mutex> lk(h->seed_mu);
			SeedWork &W = h->seed_ws;
			for (DBuf *b : { &W.hlen, &W.h_off, &W.hq_off, &W.qg, &W.segs, &W.bq, &W.cnt, &W.off, &W.scnt, &W.soff, &W.rec, &W.has, &W.bd, &W.big }) b->release();
		}
		LQ_HIP_CHECK(hipDeviceSynchronize());
#ifndef LQ_EMU
		hipMemPool_t pool = nullptr;
		if (hipDeviceGetDefaultMemPool(&pool, h->device) == hipSuccess && pool) (void)hipMemPoolTrimTo(pool, 0);
#endif
	});
}
int lqcov_finish(lqcov_handle *h) { return guard(h, [&] { if (!h->have_queries) throw std::logic_error("no queries"); h->finish(); }); }
int lqcov_n_queries(const lqcov_handle *h) { return h ? (int)h->q.n : LQCOV_E_ARG; }

int lqcov_query_order(lqcov_handle *h, uint32_t *perm, uint32_t n)
{
	return guard(h, [&] {
		if (!h->have_queries) throw std::logic_error("no queries");
		if (n < h->q.n || !perm) throw std::invalid_argument("buffer too small");
		for (u32 i = 0; i < h->q.n; ++i) perm[i] = h->q_perm[i];
	});
}

int lqcov_get_rows(lqcov_handle *h, lqcov_row *rows, uint32_t n_rows)
{
	return guard(h, [&] {
		if (!h->finished) throw std::logic_error("finish() has not run");
		if (n_rows < h->rows.size()) throw std::invalid_argument("row buffer too small");
		if (!h->rows.empty()) memcpy(rows, h->rows.data(), h->rows.size() * sizeof(lqcov_row));
	});
}

int lqcov_get_regions(lqcov_handle *h, const lqcov_region **regs, uint32_t *n_regs, const lqcov_region **mregs, uint32_t *n_mregs)
{
	return guard(h, [&] {
		if (!h->finished) throw std::logic_error("finish() has not run");
		*regs = h->regs.data(); *n_regs = (uint32_t)h->regs.size(); *mregs = h->mregs.data(); *n_mregs = (uint32_t)h->mregs.size();
	});
}

int lqcov_write_table(lqcov_handle *h, const char *out_path)
{
	return guard(h, [&] {
		FILE *o = out_path ? fopen(out_path, "w") : stdout;
		if (!o) throw std::runtime_error(std::string("failed to open file '") + out_path + "'");
		try { h->write_table(o); } catch (...) { if (out_path) fclose(o); throw; }
		if (out_path) fclose(o); else fflush(o);
	});
}

int lqcov_format_rows(int filter_flag, const lqcov_row *rows, uint32_t n_rows, const lqcov_region *regs, const lqcov_region *mregs,
                      const char *names, const uint64_t *name_off, const char *out_path)
{
	try {
		if ((n_rows && (!rows || !names || !name_off)) || !out_path) return LQCOV_E_ARG;
		FILE *o = fopen(out_path, "w");
		if (!o) return LQCOV_E_IO;
		lq_format_rows(o, filter_flag, rows, n_rows, regs, mregs, [&](u32 i) { return names + name_off[i]; });
		fclose(o);
		return 0;
	} catch (const std::exception &e) { g_create_error = e.what(); return LQCOV_E_STATE; }
}

int32_t lqcov_mid_occ(const lqcov_handle *h) { return h ? h->mid_occ : -1; }
uint64_t lqcov_part_n_minimizers(const lqcov_handle *h, int part) { return (h && part >= 0 && (size_t)part < h->parts.size() && h->parts[part]) ? h->parts[part]->rs.n_mini : 0; }
uint64_t lqcov_part_n_keys(const lqcov_handle *h, int part) { return (h && part >= 0 && (size_t)part < h->parts.size() && h->parts[part]) ? h->parts[part]->n_keys : 0; }
uint64_t lqcov_last_n_anchors(const lqcov_handle *h) { return h ? h->last_n_anchors : 0; }
int lqcov_fastx_digest(const char *path, int mode, int n_threads, uint64_t piece_bytes, uint64_t out[5])
{
	if (!path || !out) return LQCOV_E_ARG;
	try {
		auto mix = [](u64 h, const u8 *p, size_t n, bool u2t) { for (size_t i = 0; i < n; ++i) { u8 c = p[i]; if (u2t && (c == 'u' || c == 'U')) --c; h = (h ^ c) * 0x100000001b3ULL; } return (h ^ (u64)n) * 0x9E3779B97F4A7C15ULL; };
		u64 n = 0, b = 0, hn = 0xcbf29ce484222325ULL, hs = 0xcbf29ce484222325ULL, rep = 0;
		if (mode == 0) {
			FastxReader r(path);
			std::string nm, sq, ql;
			while (r.next(nm, sq, ql)) { ++n; b += sq.size(); hn = mix(hn, (const u8*)nm.data(), nm.size(), false); hs = mix(hs, (const u8*)sq.data(), sq.size(), true); }
		} else {
			MemFastx mf;
			if (!mf.open(path)) return LQCOV_E_ARG;
			MemRecords recs;
			lq_parse_all(mf, n_threads, piece_bytes, recs);
			for (MemPiece &pc : recs.pieces) for (MemRec &r : pc.recs) {
				++n; b += r.seq_len;
				hn = mix(hn, mf.data() + r.name_off, r.name_len, false);
				hs = mix(hs, (r.own ? pc.side.data() : mf.data()) + r.seq_off, r.seq_len, true);
			}
			rep = recs.reparsed;
		}
		out[0] = n; out[1] = b; out[2] = hn; out[3] = hs; out[4] = rep;
		return 0;
	} catch (const std::exception &e) { fprintf(stderr, "lqcov_fastx_digest: %s\n", e.what()); return LQCOV_E_IO; }
}

void lqcov_map_stats(const lqcov_handle *h, uint64_t out[4])
{
	if (!out) return;
	out[0] = h ? h->last_n_written : 0; out[1] = h ? (uint64_t)h->stat_sens_runs : 0; out[2] = h ? (uint64_t)h->stat_p2_queries : 0; out[3] = h ? (uint64_t)h->stat_p2_anchors : 0;
}

void lqcov_tie_reasons(const lqcov_handle *h, uint64_t out[6])
{
	if (!out) return;
	for (int i = 0; i < 6; ++i) out[i] = h ? (uint64_t)h->stat_tie_why[i] : 0;
}

static void copy_minimizers(lqcov_handle *h, ReadSetDev &rs, uint64_t *xy, uint64_t *off, uint64_t *n_total)
{
	if (!rs.sketched) throw std::logic_error("read set not sketched");
	if (n_total) *n_total = rs.n_mini;
	if (off) { LQ_HIP_CHECK(hipMemcpy(off, rs.moff.p, (rs.n + 1) * 8, hipMemcpyDeviceToHost)); }
	if (xy && rs.n_mini) {
		std::vector<u64> x(rs.n_mini), y(rs.n_mini);
		LQ_HIP_CHECK(hipMemcpy(x.data(), rs.mx.p, rs.n_mini * 8, hipMemcpyDeviceToHost));
		LQ_HIP_CHECK(hipMemcpy(y.data(), rs.my.p, rs.n_mini * 8, hipMemcpyDeviceToHost));
		for (u64 i = 0; i < rs.n_mini; ++i) { xy[2 * i] = x[i]; xy[2 * i + 1] = y[i]; }
	}
}

int lqcov_get_query_minimizers(lqcov_handle *h, uint64_t *xy, uint64_t *off, uint64_t *n_total)
{
	return guard(h, [&] {
		if (!h->have_queries) throw std::logic_error("no queries");
		// the engine keeps the queries in its own order (engine.hpp: q_perm); hand the lists out in the caller's
		const u32 n = h->q.n;
		std::vector<u64> ioff(n + 1, 0), ixy;
		u64 tot = 0;
		copy_minimizers(h, h->q, nullptr, ioff.data(), &tot);
		if (n_total) *n_total = tot;
		if (xy) { ixy.resize(2 * tot + 2); copy_minimizers(h, h->q, ixy.data(), nullptr, nullptr); }
		u64 o = 0;
		for (u32 c = 0; c < n; ++c) {                           // c: caller's index
			const u32 i = h->q_inv[c];
			const u64 cnt = ioff[i + 1] - ioff[i];
			if (off) off[c] = o;
			if (xy && cnt) memcpy(xy + 2 * o, ixy.data() + 2 * ioff[i], cnt * 16);
			o += cnt;
		}
		if (off) off[n] = o;
	});
}
int lqcov_get_part_minimizers(lqcov_handle *h, int part, uint64_t *xy, uint64_t *off, uint64_t *n_total)
{
	return guard(h, [&] { copy_minimizers(h, h->part(part).rs, xy, off, n_total); });
}

int lqcov_get_chains(lqcov_handle *h, int32_t *out, uint64_t cap, uint64_t *n_total)
{
	return guard(h, [&] {
		if (!(h->debug_flags & 1)) throw std::logic_error("chain recording is off (lqcov_set_debug(h, 1))");
		if (n_total) *n_total = h->n_dbg_host;
		u64 n = std::min<u64>(h->n_dbg_host, std::min<u64>(cap, h->dbg_cap));
		static_assert(sizeof(ChainRec) == 9 * sizeof(int32_t), "chain record layout");
		if (out && n) {
			LQ_HIP_CHECK(hipMemcpy(out, h->dbg_chains.p, n * sizeof(ChainRec), hipMemcpyDeviceToHost));
			ChainRec *r = (ChainRec*)out;
			for (u64 i = 0; i < n; ++i) r[i].q = (i32)h->q_perm[(u32)r[i].q];     // engine order -> caller's order
		}
	});
}

uint32_t lqcov_sat_record_bytes(void) { return (uint32_t)sizeof(SatRec); }

uint32_t lqcov_counter_max(const lqcov_handle *h) { return h ? h->cnt_max : 0u; }

int lqcov_counter_offsets(lqcov_handle *h, uint64_t *off)
{
	return guard(h, [&] {
		if (!h->have_queries || !off) throw std::invalid_argument("no queries / no buffer");
		LQ_HIP_CHECK(hipMemcpy(off, h->cnt_off_dev(), ((size_t)h->q.n + 1) * 8, hipMemcpyDeviceToHost));
	});
}

int lqcov_part_sat_records(lqcov_handle *h, int part, uint32_t query, void *recs, uint64_t rec_cap, uint32_t *at, uint64_t at_cap, uint64_t n_out[2])
{
	return guard(h, [&] {
		Part &pt = h->part(part);
		if (!(h->sat_last_part == part && h->sat_last_query == query && h->sat_last_valid)) {
			h->sat_last_valid = false;
			h->part_sat_records(pt, query, h->sat_last_recs, h->sat_last_at);
			h->sat_last_part = part; h->sat_last_query = query; h->sat_last_valid = true;
		}
		if (n_out) { n_out[0] = h->sat_last_recs.size(); n_out[1] = h->sat_last_at.size(); }
		if (recs) {
			if (rec_cap < h->sat_last_recs.size() || at_cap < h->sat_last_at.size() || (!at && !h->sat_last_at.empty())) throw std::invalid_argument("record buffers too small");
			if (!h->sat_last_recs.empty()) memcpy(recs, h->sat_last_recs.data(), h->sat_last_recs.size() * sizeof(SatRec));
			if (!h->sat_last_at.empty()) memcpy(at, h->sat_last_at.data(), h->sat_last_at.size() * 4);
			h->sat_last_valid = false; h->sat_last_recs.clear(); h->sat_last_at.clear();
		}
	});
}

int lqcov_sat_replay(lqcov_handle *h, uint32_t query, const void *recs, uint64_t n_recs, const uint32_t *at, uint64_t n_at, uint32_t *counters, uint64_t n_counters)
{
	return guard(h, [&] {
		if (!h->have_queries) throw std::logic_error("no queries");
		if ((n_recs && !recs) || (n_at && !at) || (n_counters && !counters)) throw std::invalid_argument("null buffers");
		h->sat_replay_host(query, (const SatRec*)recs, n_recs, at, n_at, counters, n_counters);
	});
}

int lqcov_accum_set_replayed(lqcov_handle *h, uint32_t query, const uint32_t *counters, uint64_t n_counters)
{
	return guard(h, [&] {
		if (!h->have_queries || query >= h->q.n) throw std::invalid_argument("no such query");
		h->sat_cnt[query] = std::vector<u32>(counters, counters + n_counters);
		h->finished = false;
	});
}

int lqcov_debug_sort_pairs(lqcov_handle *h, uint64_t *keys, uint64_t *vals, uint64_t n, unsigned bits, int key_bytes)
{
	return guard(h, [&] { h->debug_sort_pairs(keys, vals, n, bits, key_bytes); });
}
int lqcov_debug_scan(lqcov_handle *h, const uint32_t *in, uint64_t *out, uint64_t n)
{
	return guard(h, [&] { h->debug_scan(in, out, n); });
}

int lqcov_part_minimizers_dev(lqcov_handle *h, int part, const uint64_t **x_dev, const uint64_t **y_dev, uint64_t *n)
{
	return guard(h, [&] {
		Part &pt = h->part(part);
		if (!pt.rs.sketched) throw std::logic_error("part not sketched");
		*x_dev = pt.rs.mx.as<u64>(); *y_dev = pt.rs.my.as<u64>(); *n = pt.rs.n_mini;
	});
}

// this rank's minimizers into caller-owned device buffers, rid moved up by rid_base (the part-global index of the rank's first read)
int lqcov_part_minimizers_export_dev(lqcov_handle *h, int part, uint64_t *x_dev, uint64_t *y_dev, uint64_t cap, uint32_t rid_base)
{
	return guard(h, [&] {
		Part &pt = h->part(part);
		if (!pt.rs.sketched) throw std::logic_error("part not sketched");
		if (cap < pt.rs.n_mini) throw std::invalid_argument("minimizer buffers too small");
		h->export_minimizers(pt.rs, x_dev, y_dev, rid_base);
	});
}

int lqcov_part_sketch(lqcov_handle *h, int part)
{
	return guard(h, [&] { if (!h->have_queries) throw std::logic_error("set the queries first"); h->sketch(h->part(part).rs, true); });
}

// The minimizers of a part as `n_shares` shares laid out `stride` words apart in x_dev / y_dev (what an all-gather of equally
// sized send buffers leaves behind: share i holds share_n[i] <= stride entries, in read order): copied back to back into the
// part, then the index is built.  One share of n entries = a plain array.
int lqcov_part_build_from_minimizer_shares_dev(lqcov_handle *h, int part, const uint64_t *x_dev, const uint64_t *y_dev, uint64_t stride,
                                               uint32_t n_shares, const uint64_t *share_n,
                                               uint32_t n_targets, const uint32_t *target_len, const char *names, const uint64_t *name_off)
{
	return guard(h, [&] {
		Part &pt = h->part(part);
		ReadSetDev &rs = pt.rs;
		uint64_t n = 0;
		for (uint32_t i = 0; i < n_shares; ++i) { if (share_n[i] > stride) throw std::invalid_argument("share longer than the stride"); n += share_n[i]; }
		rs.key_stamp = 0;
		rs.mx.ensure(n * 8 + 8); rs.my.ensure(n * 8 + 8);
		uint64_t at = 0;
		for (uint32_t i = 0; i < n_shares; ++i) {
			if (!share_n[i]) continue;
			LQ_HIP_CHECK(hipMemcpyAsync(rs.mx.as<u64>() + at, x_dev + (uint64_t)i * stride, share_n[i] * 8, hipMemcpyDeviceToDevice, h->stream));
			LQ_HIP_CHECK(hipMemcpyAsync(rs.my.as<u64>() + at, y_dev + (uint64_t)i * stride, share_n[i] * 8, hipMemcpyDeviceToDevice, h->stream));
			at += share_n[i];
		}
		rs.n_mini = n; rs.n = n_targets;
		rs.h_len.assign(target_len, target_len + n_targets);
		rs.names.clear();
		for (uint32_t i = 0; i < n_targets; ++i) rs.names.emplace_back(names ? names + name_off[i] : "");
		rs.d_len.ensure((n_targets + 1) * 4);
		if (n_targets) LQ_HIP_CHECK(hipMemcpyAsync(rs.d_len.p, target_len, n_targets * 4, hipMemcpyHostToDevice, h->stream));
		LQ_HIP_CHECK(hipStreamSynchronize(h->stream));
		rs.sketched = true;
		h->build_index(pt);
	});
}

int lqcov_part_build_from_minimizers_dev(lqcov_handle *h, int part, const uint64_t *x_dev, const uint64_t *y_dev, uint64_t n,
                                         uint32_t n_targets, const uint32_t *target_len, const char *names, const uint64_t *name_off)
{
	return lqcov_part_build_from_minimizer_shares_dev(h, part, x_dev, y_dev, n, 1, &n, n_targets, target_len, names, name_off);
}

// ---- multi-GPU plumbing: per-part accumulators in / out (device pointers) ------------------------
int lqcov_set_distributed(lqcov_handle *h, int on) { if (!h) return LQCOV_E_ARG; h->distributed = on != 0; return 0; }
int lqcov_set_mid_occ(lqcov_handle *h, int32_t v) { if (!h) return LQCOV_E_ARG; h->mid_occ = v; return 0; }

int lqcov_accum_sizes(lqcov_handle *h, uint32_t *n_queries, uint64_t *n_counters, uint32_t *n_intervals)
{
	return guard(h, [&] {
		if (!h->have_queries) throw std::logic_error("no queries");
		u32 npv = 0;
		LQ_HIP_CHECK(hipMemcpy(&npv, h->n_pv.p, 4, hipMemcpyDeviceToHost));
		if (n_queries) *n_queries = h->q.n;
		if (n_counters) *n_counters = h->q.n_mini;
		if (n_intervals) *n_intervals = npv;
	});
}

static void dcopy(void *dst, const void *src, size_t bytes, hipStream_t s)
{
	if (bytes && dst && src) LQ_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
}

int lqcov_accum_export_dev(lqcov_handle *h, uint64_t *lambda_dev, uint64_t *lambda2_dev, float *avg_k_dev, uint32_t *flags_dev,
                           uint32_t *counters_dev, uint32_t *counter_owner_dev, uint32_t *intervals_dev)
{
	return guard(h, [&] {
		if (!h->have_queries) throw std::logic_error("no queries");
		const u32 n = h->q.n;
		u32 npv = 0;
		LQ_HIP_CHECK(hipMemcpy(&npv, h->n_pv.p, 4, hipMemcpyDeviceToHost));
		dcopy(lambda_dev, h->lambda.p, (size_t)n * 8, h->stream); dcopy(lambda2_dev, h->lambda2.p, (size_t)n * 8, h->stream);
		dcopy(avg_k_dev, h->avg_k.p, (size_t)n * 4, h->stream); dcopy(flags_dev, h->qflags.p, (size_t)n * 4, h->stream);
		dcopy(counters_dev, h->cnts.p, (size_t)h->q.n_mini * 4, h->stream);
		dcopy(counter_owner_dev, h->q_owner.p, (size_t)h->q.n_mini * 4, h->stream);
		dcopy(intervals_dev, h->pv.p, (size_t)npv * sizeof(Ivl), h->stream);
		LQ_HIP_CHECK(hipStreamSynchronize(h->stream));
	});
}

int lqcov_accum_import_dev(lqcov_handle *h, const uint64_t *lambda_dev, const uint64_t *lambda2_dev, const float *avg_k_dev, const uint32_t *flags_dev,
                           const uint32_t *counters_dev, const uint32_t *intervals_dev, uint32_t n_intervals)
{
	return guard(h, [&] {
		if (!h->have_queries) throw std::logic_error("no queries");
		const u32 n = h->q.n;
		static_assert(sizeof(Ivl) == 12, "interval layout");
		h->pv.ensure((size_t)n_intervals * sizeof(Ivl) + 16);
		h->pv_cap = (u32)std::min<u64>(h->pv.cap / sizeof(Ivl), 0xfffffff0ULL);
		dcopy(h->lambda.p, lambda_dev, (size_t)n * 8, h->stream); dcopy(h->lambda2.p, lambda2_dev, (size_t)n * 8, h->stream);
		dcopy(h->avg_k.p, avg_k_dev, (size_t)n * 4, h->stream); dcopy(h->qflags.p, flags_dev, (size_t)n * 4, h->stream);
		dcopy(h->cnts.p, counters_dev, (size_t)h->q.n_mini * 4, h->stream);
		dcopy(h->pv.p, intervals_dev, (size_t)n_intervals * sizeof(Ivl), h->stream);
		LQ_HIP_CHECK(hipMemcpyAsync(h->n_pv.p, &n_intervals, 4, hipMemcpyHostToDevice, h->stream));
		LQ_HIP_CHECK(hipStreamSynchronize(h->stream));
		h->sat_cnt.clear();                                       // (merged counters: nothing this rank replayed describes them)
		h->finished = false;
	});
}

int lqcov_run_files_ex(lqcov_handle *h, const char *target_path, const char *query_path, const char *dump_path, const char *out_path, const char *err_path)
{
	return guard(h, [&] {
		if (!target_path) throw std::invalid_argument("no target");
		if (!query_path && !dump_path) throw std::invalid_argument("neither a query file nor an index dump was asked for");
		FILE *o = out_path ? fopen(out_path, "w") : stdout;
		if (!o) throw std::runtime_error(std::string("failed to open file '") + out_path + "'");
		FILE *e = err_path ? fopen(err_path, "a") : stderr;
		try { h->run_files(target_path, query_path, o, e, dump_path); }
		catch (...) { if (out_path) fclose(o); if (err_path && e) fclose(e); throw; }
		if (out_path) fclose(o); else fflush(o);
		if (err_path && e) fclose(e);
	});
}

int lqcov_run_files(lqcov_handle *h, const char *target, const char *query, const char *out_path, const char *err_path)
{
	return lqcov_run_files_ex(h, target, query, nullptr, out_path, err_path);
}

int lqcov_part_dump(lqcov_handle *h, int part, const char *path, int append)
{
	return guard(h, [&] {
		FILE *fp = fopen(path, append ? "ab" : "wb");
		if (!fp) throw std::runtime_error(std::string("failed to open file '") + path + "'");
		try { h->dump_part(h->part(part), fp); } catch (...) { fclose(fp); throw; }
		fclose(fp);
	});
}

int lqcov_part_load(lqcov_handle *h, const char *path, uint64_t *offset)
{
	int id = -1;
	int rc = guard(h, [&] {
		if (!offset) throw std::invalid_argument("null offset");
		FILE *fp = fopen(path, "rb");
		if (!fp) throw std::runtime_error(std::string("failed to open file '") + path + "'");
		try {
			if (fseeko(fp, (off_t)*offset, SEEK_SET) != 0) throw std::runtime_error("seek failed");
			h->parts.emplace_back(new Part());
			const int pid = (int)h->parts.size() - 1;
			h->parts[pid]->live = true;
			if (h->load_part(fp, *h->parts[pid])) { id = pid; *offset = (uint64_t)ftello(fp); }
			else h->parts[pid].reset();
		} catch (...) { fclose(fp); throw; }
		fclose(fp);
	});
	return rc ? rc : (id < 0 ? LQCOV_EOF : id);
}

// == the subprocess (minimap2-coverage.c:206-734)
int lqcov_main(int argc, const char *const *argv, const char *out_path, const char *err_path, int device)
{
	FILE *e = err_path ? fopen(err_path, "w") : stderr;
	if (!e) return 1;
	lqcov_params p;
	const char *target = nullptr, *query = nullptr, *dump = nullptr;
	char errbuf[256] = {0};
	int rc = lqcov_parse_args(argc, argv, &p, &target, &query, &dump, errbuf, sizeof(errbuf));
	if (rc) { fprintf(e, "%s\n", errbuf); if (err_path) fclose(e); return 1; }
	// effective parameters, as the reference echoes them (minimap2-coverage.c:392-404)
	if (query) fprintf(e, "=== Parameters are listed below === \nInputs are target: %s, query: %s\n", target, query);
	else fprintf(e, "=== Parameters are listed below === \nInputs is target: %s\n", target);
	fprintf(e, "kmer %d, window %d, index loading size %llu\n", p.k, p.w, (unsigned long long)p.batch_size);
	fprintf(e, "min-score %d, min-score-med %d, min-score-good %d, max-gap %d, min-cnt %d\n", p.min_chain_score, p.min_score_med, p.min_score_good, p.max_gap, p.min_cnt);
	fprintf(e, "Homo-polymer compression: %d, Filtering: %d\n", p.hpc, p.filter_flag);
	fprintf(e, "max-overhang %d, min-overlaplen %d, min-overapratio %.2f\n===\n", p.max_overhang, p.min_ovlp, p.min_ratio);
	fflush(e);
	{	// unopenable target: exit 1 with the reference's message (minimap2-coverage.c:276-279)
		FILE *t = fopen(target, "rb");
		if (!t) { fprintf(e, "ERROR: failed to open file '%s'\n", target); if (err_path) fclose(e); return 1; }
		fclose(t);
	}
	lqcov_handle *h = lqcov_create(&p, device);
	if (!h) { fprintf(e, "ERROR: %s\n", g_create_error.c_str()); if (err_path) fclose(e); return LQCOV_E_DEVICE; }
	if (err_path) { fclose(e); e = nullptr; }
	rc = lqcov_run_files_ex(h, target, query, dump, out_path, err_path);
	if (rc) {
		FILE *e2 = err_path ? fopen(err_path, "a") : stderr;
		if (e2) { fprintf(e2, "ERROR: %s\n", h->err.c_str()); if (err_path) fclose(e2); }
	}
	// LQCOV_NO_TEARDOWN=1 (set by the executable's own main, which leaves with _exit): the process is about to end, and handing
	// a few hundred GB back to the device block by block takes seconds that the caller would wait for
	if (!getenv("LQCOV_NO_TEARDOWN")) lqcov_destroy(h);
	return rc == 0 ? 0 : (rc == LQCOV_E_IO ? 1 : rc);
}

} // extern "C"
