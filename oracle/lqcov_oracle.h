/* oracle/lqcov_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's `minimap2-coverage` hot path
 * (/root/reference/minimap2-coverage, LongQC v1.2.1).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product (longqc_amd/, include/lqcov.h) never does.
 * Parity status: PINNED -- checked byte-for-byte against the reference binary and against
 * function-level dumps of the reference's own objects (oracle/_ref, built by oracle/Makefile) in
 * tests/test_oracle_vs_ref.py, and against the golden fixtures in tests/golden/.
 */
#ifndef LQCOV_ORACLE_H
#define LQCOV_ORACLE_H
#include <stdint.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } lqo_mm128;

typedef struct {
	int k, w, hpc;              /* -k -w -H            (minimap2-coverage.c:252-266) */
	uint64_t batch_size;        /* -I, index part size (minimap2-coverage.c:268-272, index.c:36) */
	int idx_mini_batch;         /* 50,000,000          (index.c:35)  */
	int64_t qry_mini_batch;     /* 500,000,000         (map.c:40)    */
	int max_gap;                /* -g default 10000    (minimap2-coverage.c:302-307) */
	int min_cnt;                /* -n default 3 */
	int min_chain_score;        /* -m default 40 */
	int min_score_med;          /* -p default m */
	int min_score_good;         /* -q default m */
	int max_chain_skip;         /* -s default 25 */
	int bw;                     /* 500 (map.c:20) */
	int max_overhang;           /* -a default 2000 */
	int min_ovlp;               /* -l default 1000 (unused by lq_cnt_match) */
	int min_coverage;           /* -c default 3 */
	double min_ratio;           /* -r default 0.4 */
	float mid_occ_frac;         /* 2e-4f (map.c:16) */
	int seed;                   /* 11 (map.c:15) */
	int no_self;                /* -Y / -X set MM_F_NO_SELF */
	int ava;                    /* -X sets MM_F_AVA (cmp>0 skip) */
	int filter_flag;            /* --filter row format */
	/* ---- knobs of the restatement, not of the reference ---- */
	int sort_mode;              /* 0 = klib in-place radix order (reference), 1 = stable (x, emission),
	                               2 / 3 = the engine's scheme: an adversarial order of equal-x anchors (reverse emission /
	                               hashed), klib's order only in the (strand, rid) runs where the order is observable */
	int chain_mode;             /* 0 = whole-array DP (reference), 1 = per (strand,rid) group DP with
	                               the <min_cnt group pre-filter (the GPU formulation) */
} lqo_params;

void lqo_params_default(lqo_params *p);

/* (w,k)-minimizers of one read; appends to *out (realloc'd). sketch.c:76-142 */
void lqo_sketch(const char *seq, int len, int w, int k, uint32_t rid, int hpc,
                lqo_mm128 **out, size_t *n, size_t *cap);

/* Sketch every read of a FASTA/Q file; writes the same text as `ref_harness sketch`. */
int lqo_dump_sketch(const lqo_params *p, const char *fn, FILE *out);
/* Per index part: "P part n_seq tot_len mid_occ_of_this_part" (as `ref_harness index`). */
int lqo_dump_index(const lqo_params *p, const char *fn, FILE *out);
/* First index part only: per query Q/C/V/N records (as `ref_harness chains`). */
int lqo_dump_chains(const lqo_params *p, const char *target_fn, const char *query_fn, FILE *out);
/* The whole path: the 9-column table (minimap2-coverage.c:545-617). */
int lqo_run_files(const lqo_params *p, const char *target_fn, const char *query_fn, FILE *out, FILE *log);

/* path-level convenience for ctypes (returns 0 on success) */
int lqo_run_paths(const lqo_params *p, const char *target_fn, const char *query_fn, const char *out_fn);
int lqo_dump_paths(const lqo_params *p, const char *what, const char *target_fn, const char *query_fn, const char *out_fn);

/* klib sorts restated (ksort.h:84-134), exported for unit tests */
void lqo_sort_128x(lqo_mm128 *a, size_t n);
void lqo_sort_64(uint64_t *a, size_t n);
void lqo_sort_32(uint32_t *a, size_t n);

/* the reference's `sdust` binary (sdust.c): masked bases of one read, and its whole table */
uint32_t lqo_sdust_masked(const char *seq, int l_seq, int T, int W);
int lqo_sdust_file(const char *fn, int W, int T, FILE *out);
int lqo_sdust_path(const char *fn, int W, int T, const char *out_fn);

/* sort_mode 2 / 3 bookkeeping: [0] queries [1] anchors [2] runs that can hold a chain [3] their anchors [4] such runs with
 * equal-x anchors [5] of those, runs where the order is observable [6] their anchors [7] MISMATCHES (runs declared
 * order-free whose chains differ between klib's order and the adversarial one: must stay 0) [8] queries with any equal x
 * [9] queries with an observable run [10] anchors of those queries */
void lqo_tie_stats(uint64_t out[12]);
void lqo_tie_stats_reset(void);

/* timing breakdown of the last lqo_run_files call, seconds: [0]=parse [1]=sketch+index [2]=map [3]=format */
void lqo_last_timing(double t[4]);

#ifdef __cplusplus
}
#endif
#endif
