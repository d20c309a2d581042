// tools/microbench/walk_bench.hip -- development aid: ns per trip of the klib token walkers on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -I longqc_amd/csrc tools/microbench/walk_bench.hip -o tools/microbench/walk_bench
// Random digit streams (B buckets, N elements), the walk's destinations checked against a host walk.
#include "kernels_sort.hpp"
#include "kernels_walk.hpp"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

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
	const int copies_hi = argc > 2 ? atoi(argv[2]) : 1024;
	const int only_b = argc > 3 ? atoi(argv[3]) : 0;
	struct Case { const char *name; int B; int runs; };
	const Case cases[] = { {"B=6 random", 6, 1}, {"B=6 runs of 37", 6, 37}, {"B=16 random", 16, 1}, {"B=3 runs of 5", 3, 5}, {"B=50 random", 50, 1}, {"B=79 random", 79, 1}, {"B=100 random", 100, 1}, {"B=196 random", 196, 1}, {"B=200 runs of 3", 200, 3}, {"B=256 random", 256, 1} };
	for (const Case &cs : cases) {
		if (only_b && cs.B != only_b) continue;
		std::mt19937_64 rng(12345 + cs.B);
		const u32 pad = 7;                                     // misaligned start
		std::vector<u8> d(N);
		for (u32 i = 0; i < N; ) { u8 v = (u8)(rng() % cs.B); for (int r = 0; r < cs.runs && i < N; ++r) d[i++] = v; }
		u32 cnt[256] = {0}, bg[256];
		for (u8 v : d) ++cnt[v];
		u32 acc = 0; for (int c = 0; c < 256; ++c) { bg[c] = acc; acc += cnt[c]; }
		std::vector<u32> ref; host_walk(d, cnt, bg, ref);
		for (int copies : {1, copies_hi}) {
			const u64 stride = ((u64)N + pad + 63) & ~(u64)15;
			u8 *dD; u32 *dH, *dB, *dDst, *dList, *dN; SortSeg *dS;
			CK(hipMalloc(&dD, stride * copies + 64)); CK(hipMalloc(&dH, 1024 * copies)); CK(hipMalloc(&dB, 1024 * copies));
			CK(hipMalloc(&dDst, (stride * copies + 64) * 4)); CK(hipMalloc(&dList, 4 * copies)); CK(hipMalloc(&dN, 4)); CK(hipMalloc(&dS, sizeof(SortSeg) * copies));
			std::vector<SortSeg> segs(copies); std::vector<u32> list(copies);
			for (int c = 0; c < copies; ++c) {
				segs[c].off = stride * c + pad; segs[c].len = N; segs[c].shift = 40; list[c] = c;
				CK(hipMemcpy(dD + segs[c].off, d.data(), N, hipMemcpyHostToDevice));
				CK(hipMemcpy(dH + 256 * c, cnt, 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB + 256 * c, bg, 1024, hipMemcpyHostToDevice));
			}
			CK(hipMemcpy(dS, segs.data(), sizeof(SortSeg) * copies, hipMemcpyHostToDevice)); CK(hipMemcpy(dList, list.data(), 4 * copies, hipMemcpyHostToDevice));
			u32 nl = copies; CK(hipMemcpy(dN, &nl, 4, hipMemcpyHostToDevice));
			hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
			auto run = [&](const char *name, auto launch) {
				CK(hipMemset(dDst, 0xff, (stride * copies + 64) * 4));
				launch(); CK(hipDeviceSynchronize());       // warm
				CK(hipMemset(dDst, 0xff, (stride * copies + 64) * 4));
				CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				std::vector<u32> got(N);
				bool ok = true;
				for (int c : {0, copies - 1}) {
					CK(hipMemcpy(got.data(), dDst + segs[c].off, (u64)N * 4, hipMemcpyDeviceToHost));
					if (got != ref) ok = false;
				}
				printf("%-16s copies %5d  %-22s %9.3f ms  %7.1f ns/trip  %s\n", cs.name, copies, name, ms, ms * 1e6 / N, ok ? "ok" : "MISMATCH");
				fflush(stdout);
			};
			run("solo (LDS state)", [&] { hipLaunchKernelGGL(k_sort_walk_solo, dim3(copies), dim3(64), 0, 0, dS, dList, dN, dD, dH, dB, dDst, (const CkSeg*)nullptr, 0u, (const u32*)nullptr, (const u32*)nullptr); });
			const CkSeg *nock = nullptr; const u32 *nou = nullptr;
			if (cs.B <= 64) run("reg<1>", [&] { hipLaunchKernelGGL((k_sort_walk_reg<1>), dim3(copies), dim3(64), 0, 0, dS, dList, dN, dD, dH, dB, dDst, nock, 0u, nou, nou); });
			if (cs.B <= 128) run("reg<2>", [&] { hipLaunchKernelGGL((k_sort_walk_reg<2>), dim3(copies), dim3(64), 0, 0, dS, dList, dN, dD, dH, dB, dDst, nock, 0u, nou, nou); });
			run("reg<4>", [&] { hipLaunchKernelGGL((k_sort_walk_reg<4>), dim3(copies), dim3(64), 0, 0, dS, dList, dN, dD, dH, dB, dDst, nock, 0u, nou, nou); });
			if (cs.B > LQ_CK_B) {
				// many buckets: states by following the elements in bulk (k_ck_chain256), pieces by the solo walker
				const u32 n_ck1 = std::min<u32>(64, std::max<u32>(2, N / 16384));
				std::vector<CkSeg> hck(copies);
				u32 ckt = 0;
				for (int c = 0; c < copies; ++c) { hck[c].sgi = c; hck[c].tile0 = 0; hck[c].ck0 = ckt; hck[c].n_ck = n_ck1; ckt += n_ck1; }
				CkSeg *dck; u32 *dSt, *dSl, *dNck;
				CK(hipMalloc(&dck, sizeof(CkSeg) * copies)); CK(hipMalloc(&dSt, (u64)ckt * 256 * 4)); CK(hipMalloc(&dSl, (u64)ckt * 4 + 4)); CK(hipMalloc(&dNck, 4));
				CK(hipMemcpy(dck, hck.data(), sizeof(CkSeg) * copies, hipMemcpyHostToDevice)); CK(hipMemcpy(dNck, &ckt, 4, hipMemcpyHostToDevice));
				char nm[64]; snprintf(nm, sizeof(nm), "ckpt256 x%u (all)", n_ck1);
				const u32 *nou2 = nullptr;
				run(nm, [&] {
					hipLaunchKernelGGL(k_ck_chain256, dim3(copies), dim3(64), 0, 0, dck, (u32)copies, dS, dD, dH, dB, dSt, dSl);
					hipLaunchKernelGGL(k_sort_walk_solo, dim3(std::min<u32>(ckt, 1u << 18)), dim3(64), 0, 0, dS, nou2, dNck, dD, dH, dB, dDst, dck, (u32)copies, dSt, dSl);
				});
				run("  of which walk pieces", [&] {
					hipLaunchKernelGGL(k_sort_walk_solo, dim3(std::min<u32>(ckt, 1u << 18)), dim3(64), 0, 0, dS, nou2, dNck, dD, dH, dB, dDst, dck, (u32)copies, dSt, dSl);
				});
				hipFree(dck); hipFree(dSt); hipFree(dSl); hipFree(dNck);
			}
			if (cs.B <= LQ_CK_B) {
				// checkpointed: the walk of every copy cut into n_ck pieces from computed states (kernels_ckpt.hpp)
				const u32 n_ck1 = std::min<u32>(512, std::max<u32>(2, N / 16384));
				std::vector<CkSeg> hck(copies);
				u32 tiles = 0, ckt = 0;
				for (int c = 0; c < copies; ++c) { hck[c].sgi = c; hck[c].tile0 = tiles; hck[c].ck0 = ckt; hck[c].n_ck = n_ck1; tiles += N / LQ_CK_TILE + 1; ckt += n_ck1; }
				CkSeg *dck; u32 *dT, *dE, *dSt, *dSl, *dNck;
				CK(hipMalloc(&dck, sizeof(CkSeg) * copies)); CK(hipMalloc(&dT, (u64)tiles * LQ_CK_B * 4)); CK(hipMalloc(&dE, (u64)copies * LQ_CK_B * LQ_CK_B * 4));
				CK(hipMalloc(&dSt, (u64)ckt * LQ_CK_B * 4)); CK(hipMalloc(&dSl, (u64)ckt * 4 + 4)); CK(hipMalloc(&dNck, 4));
				CK(hipMemcpy(dck, hck.data(), sizeof(CkSeg) * copies, hipMemcpyHostToDevice)); CK(hipMemcpy(dNck, &ckt, 4, hipMemcpyHostToDevice));
				char nm[64]; snprintf(nm, sizeof(nm), "ckpt x%u (all kernels)", n_ck1);
				run(nm, [&] {
					hipLaunchKernelGGL(k_ck_tilehist, dim3(std::min<u32>(tiles, 65536)), dim3(256), 0, 0, dck, (u32)copies, tiles, dS, dD, dT);
					hipLaunchKernelGGL(k_ck_tilescan, dim3(copies), dim3(256), 0, 0, dck, (u32)copies, dS, dT);
					hipLaunchKernelGGL(k_ck_phases, dim3(copies), dim3(64), 0, 0, dck, (u32)copies, dS, dD, dH, dB, dT, dE);
					hipLaunchKernelGGL(k_ck_solve, dim3(std::min<u32>(ckt, 1u << 18)), dim3(64), 0, 0, dck, (u32)copies, ckt, dS, dD, dH, dB, dT, dE, dSt, dSl);
					hipLaunchKernelGGL((k_sort_walk_reg<1>), dim3(std::min<u32>(ckt, 1u << 18)), dim3(64), 0, 0, dS, nou, dNck, dD, dH, dB, dDst, dck, (u32)copies, dSt, dSl);
				});
				run("  of which walk pieces", [&] {
					hipLaunchKernelGGL((k_sort_walk_reg<1>), dim3(std::min<u32>(ckt, 1u << 18)), dim3(64), 0, 0, dS, nou, dNck, dD, dH, dB, dDst, dck, (u32)copies, dSt, dSl);
				});
				hipFree(dck); hipFree(dT); hipFree(dE); hipFree(dSt); hipFree(dSl); hipFree(dNck);
			}
			hipFree(dD); hipFree(dH); hipFree(dB); hipFree(dDst); hipFree(dList); hipFree(dN); hipFree(dS);
		}
	}
	return 0;
}
