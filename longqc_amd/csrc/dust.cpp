// longqc_amd/csrc/dust.cpp -- host side of the low-complexity table: the reference's `sdust` binary
// (sdust.c:181-222) behind the C ABI of include/lqcov.h (lqsdust_*).  Reads go to the device as ASCII, one thread
// walks one read (kernels_dust.hpp); rows are formatted on the host with libc/libm like the reference.
#include "engine.hpp"
#include "kernels_dust.hpp"
#include "fastx.hpp"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <exception>

static double lq_now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline dim3 nblk_d(u64 n, u32 b) { return dim3((unsigned)((n + b - 1) / b)); }

namespace {
struct DustDev {
	hipStream_t stream = nullptr;
	DBuf seq, qual, off, pi, masked, psum, qv, q2p;
	bool tab_ready = false;
	~DustDev() { if (stream) { hipStreamSynchronize(stream); hipStreamDestroy(stream); } }
};

void make_q2p_table(double *t)
{	// lqutils.c:26-49, rebuilt as in engine.cpp (10^(-q/10) to 15 decimals, eight entries one unit higher)
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

void dust_batch(DustDev &D, u32 n, const u8 *seq, const u64 *seq_off, const u8 *qual, int W, int T,
                u32 *masked, double *psum, u32 *qv)
{
	if (W < 3 || W > 66) throw std::domain_error("sdust window outside [3, 66] (the window ring holds 64 words; the reference's default is 64)");
	if (n == 0) return;
	for (u32 i = 0; i < n; ++i) if (seq_off[i + 1] - seq_off[i] > 0x7fffffffULL) throw std::domain_error("read longer than 2^31-1 bases");
	if (!D.stream) LQ_HIP_CHECK(hipStreamCreate(&D.stream));
	const u64 nb = seq_off[n] - seq_off[0];
	std::vector<u64> off(n + 1);
	for (u32 i = 0; i <= n; ++i) off[i] = seq_off[i] - seq_off[0];
	D.seq.ensure(nb + 16); D.off.ensure((n + 1) * 8); const u32 n_thr = (u32)std::min<u64>(((u64)n + LQ_DUST_THREADS - 1) / LQ_DUST_THREADS * LQ_DUST_THREADS, LQ_DUST_MAX_THREADS);
	D.pi.ensure((u64)n_thr * LQ_DUST_PCAP * sizeof(DustPI));
	D.masked.ensure(n * 4 + 4); D.psum.ensure(n * 8 + 8); D.qv.ensure(n * 4 + 4);
	if (!D.tab_ready) {
		double tab[127]; make_q2p_table(tab);
		D.q2p.ensure(127 * 8);
		LQ_HIP_CHECK(hipMemcpyAsync(D.q2p.p, tab, sizeof(tab), hipMemcpyHostToDevice, D.stream));
		LQ_HIP_CHECK(hipStreamSynchronize(D.stream));
		D.tab_ready = true;
	}
	LQ_HIP_CHECK(hipMemcpyAsync(D.seq.p, seq + seq_off[0], nb, hipMemcpyHostToDevice, D.stream));
	LQ_HIP_CHECK(hipMemcpyAsync(D.off.p, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, D.stream));
	if (qual) {
		D.qual.ensure(nb + 16);
		LQ_HIP_CHECK(hipMemcpyAsync(D.qual.p, qual + seq_off[0], nb, hipMemcpyHostToDevice, D.stream));
	}
	LQ_LAUNCH(k_sdust, nblk_d(n_thr, LQ_DUST_THREADS), LQ_DUST_THREADS, D.stream, D.seq.as<u8>(), qual ? D.qual.as<u8>() : (const u8*)nullptr,
	          D.off.as<u64>(), n, (i32)W, (i32)T, D.q2p.as<double>(), D.pi.as<DustPI>(), D.masked.as<u32>(), D.psum.as<double>(), D.qv.as<u32>());
	LQ_HIP_CHECK(hipGetLastError());
	LQ_HIP_CHECK(hipMemcpyAsync(masked, D.masked.p, n * 4, hipMemcpyDeviceToHost, D.stream));
	LQ_HIP_CHECK(hipMemcpyAsync(psum, D.psum.p, n * 8, hipMemcpyDeviceToHost, D.stream));
	LQ_HIP_CHECK(hipMemcpyAsync(qv, D.qv.p, n * 4, hipMemcpyDeviceToHost, D.stream));
	LQ_HIP_CHECK(hipStreamSynchronize(D.stream));
}

void set_err(char *err, size_t n, const char *msg) { if (err && n) snprintf(err, n, "%s", msg); }

int select_device(int device)
{
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) throw std::runtime_error("no HIP device available");
	if (device < 0 || device >= nd) throw std::runtime_error("HIP device index out of range");
	LQ_HIP_CHECK(hipSetDevice(device));
	return device;
}

template <class F> int guarded(char *err, size_t errlen, F &&f)
{
	try { f(); return 0; }
	catch (const std::domain_error &e) { set_err(err, errlen, e.what()); return LQCOV_E_DOMAIN; }
	catch (const std::invalid_argument &e) { set_err(err, errlen, e.what()); return LQCOV_E_ARG; }
	catch (const std::runtime_error &e) {
		set_err(err, errlen, e.what());
		return strstr(e.what(), "failed to open") ? LQCOV_E_IO : LQCOV_E_DEVICE;
	}
	catch (const std::exception &e) { set_err(err, errlen, e.what()); return LQCOV_E_STATE; }
}
} // namespace

extern "C" {

int lqsdust_reads(int device, uint32_t n, const uint8_t *seq, const uint64_t *seq_off, const uint8_t *qual, int W, int T,
                  uint32_t *masked, double *qual_psum, uint32_t *n_above_q7, char *errbuf, size_t errbuf_len)
{
	return guarded(errbuf, errbuf_len, [&] {
		if (!seq_off || (n && (!seq || !masked || !qual_psum || !n_above_q7))) throw std::invalid_argument("null buffers");
		select_device(device);
		DustDev D;
		dust_batch(D, n, seq, seq_off, qual, W, T, masked, qual_psum, n_above_q7);
	});
}

// == `sdust [-w W] [-t T] <in.fa>` with stdout -> out_path (sdust.c:181-222)
int lqsdust_main(int argc, const char *const *argv, const char *out_path, const char *err_path, int device)
{
	FILE *e = err_path ? fopen(err_path, "w") : stderr;
	if (!e) return 1;
	int W = 64, T = 20;
	const char *in = nullptr;
	for (int i = 1; i < argc; ++i) {                         // getopt "w:t:" (sdust.c:188-191): -w 64, -w64; first non-option = input
		const char *a = argv[i];
		if (a[0] == '-' && (a[1] == 'w' || a[1] == 't') ) {
			const char *v = a[2] ? a + 2 : (i + 1 < argc ? argv[++i] : nullptr);
			if (!v) { fprintf(e, "sdust: option requires an argument -- '%c'\n", a[1]); if (err_path) fclose(e); return 1; }
			if (a[1] == 'w') W = atoi(v); else T = atoi(v);
		} else if (a[0] == '-' && a[1]) {                      // getopt rejects what "w:t:" does not name
			fprintf(e, "sdust: invalid option -- '%c'\nUsage: sdust [-w %d] [-t %d] <in.fa>\n", a[1], W, T); if (err_path) fclose(e); return 1;
		} else if (!in) in = a;
	}
	if (in && !strcmp(in, "-")) in = "/dev/stdin";            // sdust.c:197: "-" reads standard input
	if (!in) { fprintf(e, "Usage: sdust [-w %d] [-t %d] <in.fa>\n", W, T); if (err_path) fclose(e); return 1; }
	char err[512] = {0};
	int rc = guarded(err, sizeof(err), [&] {
		FILE *t = fopen(in, "rb");
		if (!t) throw std::runtime_error(std::string("failed to open file '") + in + "'");
		fclose(t);
		select_device(device);
		FILE *o = out_path ? fopen(out_path, "w") : stdout;
		if (!o) throw std::runtime_error(std::string("failed to open file '") + out_path + "'");
		struct Closer { FILE *f; bool own; ~Closer() { if (own && f) fclose(f); else if (f) fflush(f); } } oc{o, out_path != nullptr};
		DustDev D;
		FastxReader fr(in);
		std::vector<u32> masked, qv;
		std::vector<double> psum;
		const bool timing = getenv("LQCOV_TIMING") != nullptr;
		double t_wait = 0, t_dev = 0, t_rows = 0, t_parse = 0;
		// The reader runs ahead on a thread of its own (round 6): mini-batch i + 1 is parsed while mini-batch i is on the device --
		// the two took 0.63 s and 0.64 s one after the other for configs[1]'s 744 Mbases.  Two batches in flight, handed over in order.
		ReadBatch rbs[2];
		std::mutex mu; std::condition_variable cv;
		int filled[2] = {0, 0};                                  // 0: the reader may fill it, 1: ready, 2: the stream is over
		std::exception_ptr rerr;
		std::thread reader([&] {
			try {
				for (int k = 0;; k ^= 1) {
					{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled[k] == 0; }); }
					const double ta = lq_now_s();
					rbs[k].clear();
					const bool more = fr.read_minibatch(200000000, rbs[k], true, true) != 0;
					t_parse += lq_now_s() - ta;
					{ std::lock_guard<std::mutex> lk(mu); filled[k] = more ? 1 : 2; }
					cv.notify_all();
					if (!more) break;
				}
			} catch (...) { std::lock_guard<std::mutex> lk(mu); rerr = std::current_exception(); filled[0] = filled[1] = 2; cv.notify_all(); }
		});
		struct Joiner { std::thread &t; std::mutex &mu; std::condition_variable &cv; int *filled; ~Joiner() { { std::lock_guard<std::mutex> lk(mu); if (filled[0] == 1) filled[0] = 0; if (filled[1] == 1) filled[1] = 0; } cv.notify_all(); if (t.joinable()) t.join(); } } joiner{reader, mu, cv, filled};
		double t0 = lq_now_s();
		for (int k = 0;; k ^= 1) {
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled[k] != 0; }); }
			if (rerr) std::rethrow_exception(rerr);
			if (filled[k] == 2) break;
			ReadBatch &rb = rbs[k];
			const u32 n = rb.size();
			masked.resize(n); qv.resize(n); psum.resize(n);
			double t1 = lq_now_s(); t_wait += t1 - t0;
			dust_batch(D, n, rb.seq.data(), rb.seq_off.data(), rb.any_qual ? rb.qual.data() : nullptr, W, T, masked.data(), psum.data(), qv.data());
			t0 = lq_now_s(); t_dev += t0 - t1;
			for (u32 i = 0; i < n; ++i) {
				const int len = (int)(rb.seq_off[i + 1] - rb.seq_off[i]);
				const bool has_q = rb.any_qual && len > 0 && rb.qual[rb.seq_off[i]] != 0;
				volatile double num = has_q ? psum[i] : 0.0; volatile int ql = has_q ? len : 0;     // meanQ(qual.s, qual.l): 0/0 without qualities
				const double mq = -10 * log10(num / ql);
				volatile double m = (double)masked[i]; volatile int sl = len;
				fprintf(o, "%s\t%d\t%d\t%.3f\t%.3f\t%d\n", rb.name(i), (int)masked[i], len, m / sl, mq, (int)qv[i]);
			}
			t1 = lq_now_s(); t_rows += t1 - t0; t0 = t1;
			{ std::lock_guard<std::mutex> lk(mu); filled[k] = 0; }
			cv.notify_all();
		}
		if (timing) fprintf(e, "[timing] sdust: parse %.3f s on the reader's thread (waited for: %.3f s), upload + kernel + download %.3f s, rows %.3f s\n", t_parse, t_wait, t_dev, t_rows);
	});
	if (rc) fprintf(e, "ERROR: %s\n", err);
	if (err_path) fclose(e);
	return rc == 0 ? 0 : (rc == LQCOV_E_IO ? 1 : rc);
}

} // extern "C"
