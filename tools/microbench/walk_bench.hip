// tools/microbench/walk_bench.hip -- development aid: the klib token walkers and the checkpoint solvers on MI355X, kernel by kernel.
//   hipcc --offload-arch=gfx950 -O3 -I longqc_amd/csrc tools/microbench/walk_bench.hip -o tools/microbench/walk_bench
//   walk_bench N copies B shape [unit]      shape: r = random digits, s = sawtooth (ascending lists of random length: what a query's hits
//                                           look like below the strand byte -- minimizer-major, ascending rid inside a minimizer), rNN = runs of NN
// Digit streams of B buckets, N elements, `copies` sub-arrays; the destinations of every walk are compared with a host walk.
#define LQ_CKM_STATS
#include "kernels_sort.hpp"
#include "kernels_walk.hpp"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void host_walk(const std::vector<u8> &d, const u32 *cnt, const u32 *bg, std::vector<u32> &dst)
{
	const u32 n = (u32)d.size();
	std::vector<u32> cur(bg, bg + 256), end(256);
	for (int c = 0; c < 256; ++c) end[c] = bg[c] + cnt[c];
	dst.assign(n, 0xffffffffu);
	for (u32 k = 0; k < 256; ++k) {
		while (cur[k] < end[k]) {
			u32 hole = cur[k], src = hole, l = d[hole];
			while (l != k) { u32 c = cur[l]++; dst[src] = c; src = c; l = d[c]; }
			dst[src] = hole; cur[k]++;
		}
	}
}

int main(int argc, char **argv)
{
	const u32 N = argc > 1 ? (u32)atol(argv[1]) : 1000000;
	const int copies = argc > 2 ? atoi(argv[2]) : 64;
	const int B = argc > 3 ? atoi(argv[3]) : 256;
	const std::string shape = argc > 4 ? argv[4] : "r";
	const u32 unit = argc > 5 ? (u32)atol(argv[5]) : (B <= LQ_CK_B ? 16384u : 4096u);
	const int variants = 4;                                 // different streams among the copies (copy c uses stream c % variants)
	std::vector<std::vector<u8>> dv(variants);
	std::vector<std::vector<u32>> refs(variants), cnts(variants, std::vector<u32>(256)), bgs(variants, std::vector<u32>(256));
	for (int v = 0; v < variants; ++v) {
		std::mt19937_64 rng(12345 + B * 7 + v);
		std::vector<u8> &d = dv[v]; d.resize(N);
		if (shape[0] == 's') { for (u32 i = 0; i < N; ) { u32 len = 1 + (u32)(rng() % 80); std::vector<u8> l(len); for (u8 &x : l) x = (u8)(rng() % B); std::sort(l.begin(), l.end()); for (u32 j = 0; j < len && i < N; ++j) d[i++] = l[j]; } }
		else { const int runs = shape.size() > 1 ? atoi(shape.c_str() + 1) : 1; for (u32 i = 0; i < N; ) { u8 x = (u8)(rng() % B); for (int r = 0; r < runs && i < N; ++r) d[i++] = x; } }
		for (u8 x : d) ++cnts[v][x];
		u32 acc = 0; for (int c = 0; c < 256; ++c) { bgs[v][c] = acc; acc += cnts[v][c]; }
		host_walk(d, cnts[v].data(), bgs[v].data(), refs[v]);
	}
	const u32 pad = 7;                                      // misaligned start
	const u64 stride = ((u64)N + pad + 63) & ~(u64)15;
	u8 *dD; u32 *dH, *dB, *dDst, *dList, *dN; SortSeg *dS;
	CK(hipMalloc(&dD, stride * copies + 256)); CK(hipMalloc(&dH, 1024 * (size_t)copies)); CK(hipMalloc(&dB, 1024 * (size_t)copies));
	CK(hipMalloc(&dDst, (stride * copies + 64) * 4)); CK(hipMalloc(&dList, 4 * (size_t)copies * LQ_WALK_CLASSES)); CK(hipMalloc(&dN, 64)); CK(hipMalloc(&dS, sizeof(SortSeg) * copies));
	std::vector<SortSeg> segs(copies); std::vector<u32> list((size_t)copies * LQ_WALK_CLASSES, 0);
	for (int c = 0; c < copies; ++c) {
		const int v = c % variants;
		segs[c].off = stride * c + pad; segs[c].len = N; segs[c].shift = 40; list[(size_t)4 * copies + c] = c; list[c] = c;
		CK(hipMemcpy(dD + segs[c].off, dv[v].data(), N, hipMemcpyHostToDevice));
		CK(hipMemcpy(dH + 256 * c, cnts[v].data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB + 256 * c, bgs[v].data(), 1024, hipMemcpyHostToDevice));
	}
	CK(hipMemcpy(dS, segs.data(), sizeof(SortSeg) * copies, hipMemcpyHostToDevice)); CK(hipMemcpy(dList, list.data(), 4 * list.size(), hipMemcpyHostToDevice));
	u32 nw[8] = {0, 0, 0, 0, (u32)copies, 0, 0, 0};     // n_walk[c]: every copy in class 4 (the longest)
	u32 nl = copies;
	CK(hipMemcpy(dN, nw, 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dN + 8, &nl, 4, hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto timed = [&](const char *name, auto launch) {
		CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
		CK(hipGetLastError());
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		printf("    %-28s %9.3f ms\n", name, ms); fflush(stdout);
		return ms;
	};
	auto check = [&](const char *what) {
		std::vector<u32> got(N);
		int bad = 0;
		for (int c = 0; c < copies; ++c) {
			if (c >= 2 * variants && c < copies - variants) continue;
			CK(hipMemcpy(got.data(), dDst + segs[c].off, (u64)N * 4, hipMemcpyDeviceToHost));
			if (got != refs[c % variants]) { ++bad; if (bad == 1) { u32 i = 0; while (got[i] == refs[c % variants][i]) ++i; printf("    first difference: copy %d element %u got %u want %u\n", c, i, got[i], refs[c % variants][i]); } }
		}
		printf("  %-30s destinations %s\n", what, bad ? "MISMATCH" : "ok"); fflush(stdout);
	};
	printf("N %u copies %d (%.1f M elements) B %d shape %s unit %u\n", N, copies, (double)N * copies / 1e6, B, shape.c_str(), unit);
	const u32 cap_cks = copies;
	CkSeg *dck; u32 *dckn;
	CK(hipMalloc(&dck, sizeof(CkSeg) * copies)); CK(hipMalloc(&dckn, 64));
	const u64 ck_max = (u64)N / unit * copies + (u64)(2 + LQ_CKM_Q) * copies, tiles_max = ((u64)N / LQ_CK_TILE + 1) * copies;
	u32 *dSt, *dSl, *dT = nullptr, *dE = nullptr;
	const bool small = B <= LQ_CK_B;
	CK(hipMalloc(&dSt, ck_max * (small ? LQ_CK_B : 256) * 4)); CK(hipMalloc(&dSl, ck_max * 4 + 4));
	if (small) { CK(hipMalloc(&dT, (tiles_max + 1) * LQ_CK_B * 4)); CK(hipMalloc(&dE, (u64)copies * LQ_CK_B * LQ_CK_B * 4)); }
	for (int rep = 0; rep < 2; ++rep) {
		printf(" pass %d\n", rep);
		CK(hipMemset(dDst, 0xff, (stride * copies + 64) * 4));
		timed("k_ck_plan", [&] { hipLaunchKernelGGL(k_ck_plan, dim3(1), dim3(256), 0, 0, dS, dList, (u32)copies, dN, 0, unit, small ? 512u : 256u, small ? 1u : (u32)LQ_CKM_Q, dck, cap_cks, dckn); });
		u32 hn[3]; CK(hipMemcpy(hn, dckn, 12, hipMemcpyDeviceToHost));
		if (rep == 0) printf("    checkpoints %u sub-arrays %u tiles %u\n", hn[0], hn[1], hn[2]);
		float tot = 0;
		if (small) {
			tot += timed("k_ck_tilehist", [&] { hipLaunchKernelGGL(k_ck_tilehist, dim3((u32)std::min<u64>(tiles_max, 1u << 16)), dim3(256), 0, 0, dck, dckn, dS, dD, dT); });
			tot += timed("k_ck_tilescan", [&] { hipLaunchKernelGGL(k_ck_tilescan, dim3(std::min<u32>(copies, 8192)), dim3(256), 0, 0, dck, dckn, dS, dT); });
			tot += timed("k_ck_phases", [&] { hipLaunchKernelGGL(k_ck_phases, dim3((u32)std::min<u64>((u64)copies * LQ_CK_B, 1u << 16)), dim3(64), 0, 0, dck, dckn, dS, dD, dH, dB, dT, dE); });
			tot += timed("k_ck_solve", [&] { hipLaunchKernelGGL(k_ck_solve, dim3((u32)std::min<u64>(ck_max, 1u << 18)), dim3(64), 0, 0, dck, dckn, dS, dD, dH, dB, dT, dE, dSt, dSl); });
			tot += timed("k_sort_walk_reg<1> pieces", [&] { hipLaunchKernelGGL((k_sort_walk_reg<1>), dim3((u32)std::min<u64>(ck_max, 1u << 18)), dim3(64), 0, 0, dS, (const u32*)nullptr, dckn, dD, dH, dB, dDst, dck, dckn, dSt, dSl); });
		} else {
			tot += timed("k_ck_chain256", [&] { hipLaunchKernelGGL(k_ck_chain256, dim3((u32)std::min<u64>(ck_max / LQ_CKM_Q + 1, 1u << 18)), dim3(LQ_CKM_THREADS), 0, 0, dck, dckn, dS, dD, dH, dB, dSt, dSl); });
			{ unsigned long long st[8]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(lq_ckm_stats), 64)); const double nb = (double)std::max<unsigned long long>(st[2], 1);
			  printf("    blocks %llu: rounds per block %.0f (from scratch %.0f, most %llu), solves per block %.1f, cycles per block %.0f (from scratch %.0f) -> %.0f cycles per round\n", st[2], st[0] / nb, st[1] / nb, st[5], st[6] / nb, st[3] / nb, st[4] / nb, (double)st[3] / std::max<double>((double)st[0], 1));
			  unsigned long long z[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(lq_ckm_stats), z, 64)); }
			tot += timed("k_sort_walk_solo pieces", [&] { hipLaunchKernelGGL(k_sort_walk_solo, dim3((u32)std::min<u64>(ck_max, 1u << 18)), dim3(64), 0, 0, dS, (const u32*)nullptr, dckn, dD, dH, dB, dDst, dck, dckn, dSt, dSl); });
		}
		printf("    %-28s %9.3f ms\n", "checkpointed, in all", tot);
		check("checkpointed");
	}
	if (argc > 6) {                                         // whole walks for comparison (slow for long sub-arrays)
		CK(hipMemset(dDst, 0xff, (stride * copies + 64) * 4));
		timed("k_sort_walk_solo whole", [&] { hipLaunchKernelGGL(k_sort_walk_solo, dim3(copies), dim3(64), 0, 0, dS, dList, dN + 8, dD, dH, dB, dDst, (const CkSeg*)nullptr, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr); });
		check("whole walks");
	}
	return 0;
}
