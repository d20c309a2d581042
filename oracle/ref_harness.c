/* oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A small driver, written for this repo, that links the reference's own objects
 * (oracle/_ref/libminimap2_ref.a, compiled from /root/reference/minimap2-coverage/ *.c by
 * oracle/Makefile `make ref`) and dumps function-level results so that the C restatement in
 * lqcov_oracle.c -- and through it the HIP kernels -- can be pinned stage by stage:
 *
 *   ref_harness sketch <k> <w> <hpc> <reads.fx>              mm_sketch  (sketch.c:76)
 *   ref_harness index  <k> <w> <hpc> <I> <targets.fx>        mm_idx_gen (index.c:311) + mm_idx_cal_max_occ (index.c:123)
 *   ref_harness chains <k> <w> <hpc> <I> <m> <p> <q> <targets.fx> <queries.fx>
 *                                                             lq_map_frag_mod (lqmap.c:207) on the first index part only
 *
 * Output is plain text on stdout, one record per line, in a canonical order.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <inttypes.h>
#include "minimap.h"
#include "mmpriv.h"
#include "kalloc.h"
#include "minimap2-coverage.h"
#include "kseq.h"
KSEQ_INIT(gzFile, gzread)

int lq_map_frag_mod(const mm_idx_t *mi, int n_segs, const int *qlens, const char **seqs, const char **quals, int *n_regs, mm_reg1_t **regs,
                    mm_tbuf_t *b, const mm_mapopt_t *opt, const lq_fltopt_t *fopt, int min_score_med_good, const char *qname,
                    uint64_t *lambdas, uint64_t *lambdas2, lq_minimizer_cnt_v *m_cnts, lq_subcoords_v **ovlp_cors, float *ks, int off_q);

static int64_t parse_num(const char *str)
{
	double x; char *p;
	x = strtod(str, &p);
	if (*p == 'G' || *p == 'g') x *= 1e9;
	else if (*p == 'M' || *p == 'm') x *= 1e6;
	else if (*p == 'K' || *p == 'k') x *= 1e3;
	return (int64_t)(x + .499);
}

static int cmd_sketch(int argc, char **argv)
{
	int k = atoi(argv[2]), w = atoi(argv[3]), hpc = atoi(argv[4]);
	gzFile f = gzopen(argv[5], "r");
	kseq_t *ks;
	uint32_t rid = 0;
	if (!f) { fprintf(stderr, "cannot open %s\n", argv[5]); return 1; }
	ks = kseq_init(f);
	while (kseq_read(ks) >= 0) {
		mm128_v mv = {0,0,0};
		size_t j;
		if (ks->seq.l > 0) mm_sketch(0, ks->seq.s, ks->seq.l, w, k, rid, hpc, &mv);
		printf("R\t%s\t%d\t%zu\n", ks->name.s, (int)ks->seq.l, mv.n);
		for (j = 0; j < mv.n; ++j)
			printf("M\t%016" PRIx64 "\t%016" PRIx64 "\n", mv.a[j].x, mv.a[j].y);
		free(mv.a);
		++rid;
	}
	kseq_destroy(ks); gzclose(f);
	return 0;
}

static int cmd_index(int argc, char **argv)
{
	mm_idxopt_t iopt;
	mm_idx_reader_t *r;
	mm_idx_t *mi;
	int part = 0;
	mm_idxopt_init(&iopt);
	iopt.k = atoi(argv[2]); iopt.w = atoi(argv[3]);
	if (atoi(argv[4])) iopt.flag |= MM_I_HPC;
	iopt.batch_size = parse_num(argv[5]);
	mm_verbose = 1;
	r = mm_idx_reader_open(argv[6], &iopt, 0);
	if (!r) return 1;
	while ((mi = mm_idx_reader_read(r, 3)) != 0) {
		int32_t mo = mm_idx_cal_max_occ(mi, 2e-4f);
		uint64_t tot_len = 0;
		uint32_t i;
		for (i = 0; i < mi->n_seq; ++i) tot_len += mi->seq[i].len;
		printf("P\t%d\t%u\t%" PRIu64 "\t%d\n", part, mi->n_seq, tot_len, mo);
		++part;
		mm_idx_destroy(mi);
	}
	mm_idx_reader_close(r);
	return 0;
}

static int reg_cmp(const void *a_, const void *b_)
{
	const mm_reg1_t *a = (const mm_reg1_t*)a_, *b = (const mm_reg1_t*)b_;
	if (a->rid != b->rid) return a->rid < b->rid ? -1 : 1;
	if (a->rev != b->rev) return a->rev < b->rev ? -1 : 1;
	if (a->rs != b->rs) return a->rs < b->rs ? -1 : 1;
	if (a->qs != b->qs) return a->qs < b->qs ? -1 : 1;
	if (a->re != b->re) return a->re < b->re ? -1 : 1;
	if (a->qe != b->qe) return a->qe < b->qe ? -1 : 1;
	if (a->score0 != b->score0) return a->score0 < b->score0 ? -1 : 1;
	return a->cnt < b->cnt ? -1 : a->cnt > b->cnt;
}

static int cmd_chains(int argc, char **argv)
{
	mm_idxopt_t iopt;
	mm_mapopt_t mopt;
	lq_fltopt_t fopt;
	mm_idx_reader_t *r;
	mm_idx_t *mi;
	gzFile f;
	kseq_t *ks;
	int m, p, q, qi = 0;
	mm_tbuf_t *tb;
	mm_idxopt_init(&iopt); mm_mapopt_init(&mopt);
	iopt.k = atoi(argv[2]); iopt.w = atoi(argv[3]);
	if (atoi(argv[4])) iopt.flag |= MM_I_HPC;
	iopt.batch_size = parse_num(argv[5]);
	m = atoi(argv[6]); p = atoi(argv[7]); q = atoi(argv[8]);
	mopt.flag |= MM_F_NO_SELF | LQ_F_AVA;
	mopt.max_gap = 10000; mopt.min_cnt = 3; mopt.min_chain_score = m; mopt.max_chain_skip = 25;
	fopt.min_coverage = 3; fopt.max_overhang = 2000; fopt.min_ovlp = 0; fopt.min_ratio = 0.4;
	mm_verbose = 1;
	r = mm_idx_reader_open(argv[9], &iopt, 0);
	if (!r) return 1;
	mi = mm_idx_reader_read(r, 3);
	if (!mi) return 1;
	mm_mapopt_update(&mopt, mi);
	printf("I\t%u\t%d\n", mi->n_seq, mopt.mid_occ);
	f = gzopen(argv[10], "r");
	if (!f) return 1;
	ks = kseq_init(f);
	tb = mm_tbuf_init();
	while (kseq_read(ks) >= 0) {
		mm128_v mv = {0,0,0};
		lq_minimizer_cnt_v m_cnts = {0,0,0};
		m_array ma;
		uint64_t lambda = 0, lambda2 = 0;
		float avgk = 0.0f;
		lq_subcoords_v *ov = (lq_subcoords_v*)calloc(1, sizeof(lq_subcoords_v)), **ovp = &ov;
		int n_regs = 0, qlen = ks->seq.l, j;
		const char *seq = ks->seq.s;
		mm_reg1_t *regs = 0;
		size_t z;
		if (qlen == 0) { printf("Q\t%d\t%s\t0\t0\t0\t0\n", qi++, ks->name.s); free(ov); continue; }
		mm_sketch(0, ks->seq.s, ks->seq.l, iopt.w, iopt.k, qi, !!(iopt.flag&MM_I_HPC), &mv);
		ma.n = mv.n; ma.a = (uint16_t*)calloc(mv.n ? mv.n : 1, 2);
		kv_push(m_array, 0, m_cnts, ma);
		free(mv.a);
		lq_map_frag_mod(mi, 1, &qlen, &seq, 0, &n_regs, &regs, tb, &mopt, &fopt, (p << 16) | q, ks->name.s,
		                &lambda, &lambda2, &m_cnts, ovp, &avgk, 0);
		printf("Q\t%d\t%s\t%d\t%d\t%" PRIu64 "\t%" PRIu64 "\n", qi, ks->name.s, qlen, n_regs, lambda, lambda2);
		qsort(regs, n_regs, sizeof(mm_reg1_t), reg_cmp);
		for (j = 0; j < n_regs; ++j)
			printf("C\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", regs[j].rid, regs[j].rev, regs[j].score0, regs[j].cnt,
			       regs[j].qs, regs[j].qe, regs[j].rs, regs[j].re);
		for (z = 0; z < ov->n; ++z) printf("V\t%u\t%u\n", ov->a[z].start, ov->a[z].end);
		for (z = 0; z < ma.n; ++z) if (ma.a[z]) printf("N\t%zu\t%u\n", z, ma.a[z]);
		for (j = 0; j < n_regs; ++j) free(regs[j].p);
		free(regs); free(ov->a); free(ov); free(ma.a); free(m_cnts.a);
		++qi;
	}
	mm_tbuf_destroy(tb);
	kseq_destroy(ks); gzclose(f);
	mm_idx_destroy(mi);
	mm_idx_reader_close(r);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc >= 6 && strcmp(argv[1], "sketch") == 0) return cmd_sketch(argc, argv);
	if (argc >= 7 && strcmp(argv[1], "index") == 0) return cmd_index(argc, argv);
	if (argc >= 11 && strcmp(argv[1], "chains") == 0) return cmd_chains(argc, argv);
	fprintf(stderr, "usage: ref_harness sketch|index|chains ... (see header)\n");
	return 2;
}
