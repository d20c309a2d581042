/* include/lqcov.h -- C ABI of liblqcov.so: the MI355X-native stand-in for LongQC's
 * `minimap2-coverage` subprocess (the hot path behind `longQC.py sampleqc`).
 *
 * The reference has no FFI for this path: the boundary is a process boundary --
 * lq_exec.py:13-38 (`LqExec.exec(*argv, out=, err=)` -> Popen) driven by longQC.py:438-446 and
 * polled at longQC.py:520-526; the C side of it is `main` (minimap2-coverage.c:206) and, one level
 * down, the in-process seam `lq_map_file` (minimap2-coverage.h:41-42; lqmap.c:852).  This header
 * therefore offers two levels, each citing what it replaces:
 *
 *   1. lqcov_main / lqcov_run_files     == the subprocess: same argv in, same 9-column table out
 *                                          (minimap2-coverage.c:166-195 options, :545-617 rows).
 *   2. handle + read-set + part calls    == main's body (minimap2-coverage.c:406-458) and
 *                                          lq_map_file (lqmap.c:852): caller owns the reads, the
 *                                          library owns the device state and the accumulators.
 *
 * All pointers are plain host pointers unless a name ends in _dev (HIP device pointers, used by
 * bench.py and the multi-GPU driver to keep data resident / exchange it over RCCL).  No torch or
 * C++ types cross this boundary.  Every int-returning call yields 0 on success and a negative
 * LQCOV_E_* code on failure; the message is available from lqcov_last_error().  Handles are not
 * thread-safe; one handle drives one HIP device and one stream.
 */
#ifndef LQCOV_H
#define LQCOV_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LQCOV_ABI_VERSION 1

#define LQCOV_E_ARG     (-1)   /* bad argument / bad option value                     */
#define LQCOV_E_IO      (-2)   /* cannot open / read / write a file                   */
#define LQCOV_E_DEVICE  (-3)   /* HIP error (no device, out of memory, launch failed) */
#define LQCOV_E_STATE   (-4)   /* call sequence violated                              */
#define LQCOV_E_DOMAIN  (-5)   /* input outside the supported domain  */
#define LQCOV_EOF       (-100) /* lqcov_part_load: no further part in the file        */

typedef struct lqcov_handle lqcov_handle;

/* Effective parameters == what minimap2-coverage echoes on stderr (minimap2-coverage.c:392-404).
 * Defaults are the binary's own (minimap2-coverage.c:252-388, map.c:12-44, index.c:31-37). */
typedef struct lqcov_params {
	int32_t  k;                /* -k  (default 12)                                   */
	int32_t  w;                /* -w  (default 5)                                    */
	int32_t  hpc;              /* -H  homopolymer-compressed k-mers                  */
	uint64_t batch_size;       /* -I  bases per index part (default 4G)              */
	int32_t  idx_mini_batch;   /* 50,000,000: part boundaries fall on these (index.c:35,244) */
	int32_t  max_gap;          /* -g  (default 10000), both axes                     */
	int32_t  min_cnt;          /* -n  (default 3)                                    */
	int32_t  min_chain_score;  /* -m  (default 40)                                   */
	int32_t  min_score_med;    /* -p  (default m)                                    */
	int32_t  min_score_good;   /* -q  (default m)                                    */
	int32_t  max_chain_skip;   /* -s  (default 25)                                   */
	int32_t  bw;               /* 500 (map.c:20)                                     */
	int32_t  max_overhang;     /* -a  (default 2000)                                 */
	int32_t  min_ovlp;         /* -l  (default 1000; parsed, unused by lq_cnt_match) */
	int32_t  min_coverage;     /* -c  (default 3)                                    */
	double   min_ratio;        /* -r  (default 0.4)                                  */
	float    mid_occ_frac;     /* 2e-4f (map.c:16)                                   */
	int32_t  no_self;          /* -Y or -X: skip the self diagonal (lqmap.c:185)     */
	int32_t  ava;              /* -X: also skip cmp>0 pairs (lqmap.c:187)            */
	int32_t  filter_flag;      /* --filter row format (minimap2-coverage.c:586-588)  */
	int32_t  n_threads;        /* -t: accepted, ignored (the device is the pool)     */
} lqcov_params;

/* One output row before text formatting (minimap2-coverage.c:545-605). Regions live in the pools
 * returned by lqcov_get_regions(). */
typedef struct lqcov_row {
	uint64_t lambda;           /* column 3  */
	uint64_t lambda2;          /* numerator of column 9 */
	double   qual_psum;        /* sum of q2p[q-33] over the read (lqutils.c:51-56); NaN-free, host takes log10 */
	uint32_t qlen;             /* column 2  */
	uint32_t n_mini;           /* mv.n      (minimap2-coverage.c:422) */
	uint32_t n_match;          /* #counters above the integer mean (minimap2-coverage.c:552-561) */
	float    avg_k;            /* esterr.c:93-97 */
	uint32_t reg_off, n_reg;   /* regs  (column 4) */
	uint32_t mreg_off, n_mreg; /* mregs (column 5) */
	uint32_t has_qual;         /* 0 for FASTA queries: meanQ prints as the reference's NaN */
	uint32_t flags;            /* LQCOV_ROW_* */
} lqcov_row;
#define LQCOV_ROW_SATURATED 1u /* a uint16 match counter reached 65535: esterr.c:130,136 make the result depend on the order of the chains */
#define LQCOV_ROW_REPLAYED  4u /* ... and the query's counters were replayed in the reference's chain order (hit.c:52-88): the row is exact.
                                  SATURATED without REPLAYED (counters merged from several ranks): row not guaranteed */

typedef struct lqcov_region { uint32_t start, end; } lqcov_region;

/* Per-stage device time of the calls made so far on the handle (ms, HIP events on the handle's
 * stream); filled when profiling was switched on with lqcov_set_profiling(). */
typedef struct lqcov_stage_time {
	char     name[48];
	double   total_ms;
	uint64_t launches;
	uint64_t algo_bytes;       /* algorithmic (compulsory) bytes moved, SURVEY.md section 8(d) */
} lqcov_stage_time;

/* ---- level 1: the subprocess ------------------------------------------------------------- */
/* == `minimap2-coverage argv...` with stdout -> out_path (NULL: stdout), stderr -> err_path
 * (NULL: stderr).  Returns the process exit status the reference would give (0 / 1), or a
 * negative LQCOV_E_* for device errors.                         minimap2-coverage.c:206-734 */
int lqcov_main(int argc, const char *const *argv, const char *out_path, const char *err_path, int device);

/* same, on an existing handle and with explicit paths (NULL out: stdout, NULL err: stderr) */
int lqcov_run_files(lqcov_handle *h, const char *target_path, const char *query_path, const char *out_path, const char *err_path);
/* same with the reference's -d: every index part is also appended to dump_path in the reference's .mmi layout
 * (mm_idx_dump, index.c:390-426); query_path may be NULL (index only, minimap2-coverage.c:460-468).  target_path may
 * itself be such a file -- from the reference or from here -- (mm_idx_load, index.c:428-479): its k, w and -H then
 * override the handle's for the mapping, as in the reference (index.c:529-531).                                        */
int lqcov_run_files_ex(lqcov_handle *h, const char *target_path, const char *query_path, const char *dump_path,
                       const char *out_path, const char *err_path);

/* ---- level 2: handle ---------------------------------------------------------------------- */
void lqcov_params_default(lqcov_params *p);                                /* minimap2-coverage.c:229-388 */
/* parse the reference's option table; fills target/query with pointers into argv.   :166-197 */
int  lqcov_parse_args(int argc, const char *const *argv, lqcov_params *p, const char **target, const char **query,
                      const char **dump_path, char *errbuf, size_t errbuf_len);
lqcov_handle *lqcov_create(const lqcov_params *p, int device);            /* NULL if no HIP device */
void lqcov_destroy(lqcov_handle *h);
const char *lqcov_last_error(const lqcov_handle *h);
int  lqcov_abi_version(void);
int  lqcov_set_profiling(lqcov_handle *h, int on);                         /* 0 off; 1 wait for every kernel (exclusive per-kernel times); 2 record events only, no waits */
int  lqcov_set_profiling_only(lqcov_handle *h, const char *stage);         /* time only the named stage (NULL / "": all) */
int  lqcov_set_debug(lqcov_handle *h, unsigned flags);                     /* bit0: record chains for lqcov_get_chains */
int  lqcov_get_stage_times(lqcov_handle *h, lqcov_stage_time *out, int max_out);   /* returns count */

/* Query reads (the subsample): n reads, bases seq[seq_off[i] .. seq_off[i+1]) as ASCII, optional
 * qualities with the same offsets, names as n NUL-terminated strings name[name_off[i]...].
 * Uploads, 2-bit packs and sketches them; sizes the per-query accumulators.
 * == main pass 1 (minimap2-coverage.c:406-444) + mm_bseq_read2 (bseq.c:68-102).              */
int lqcov_set_queries(lqcov_handle *h, uint32_t n, const uint8_t *seq, const uint64_t *seq_off,
                      const uint8_t *qual, const char *names, const uint64_t *name_off);

/* Index parts == iterations of the loop at minimap2-coverage.c:449-458.  The caller decides the
 * part boundaries (lqcov_run_files applies the reference's rule, index.c:244,311-316). */
int lqcov_part_begin(lqcov_handle *h);                                     /* returns part id >= 0 */
int lqcov_part_add_targets(lqcov_handle *h, int part, uint32_t n, const uint8_t *seq, const uint64_t *seq_off,
                           const char *names, const uint64_t *name_off);   /* == mm_idx_gen step 0 (index.c:240-288) */
/* The same reads handed over 2-bit packed, as the parser thread of lqcov_run_files does it (0.375 B per base cross
 * PCIe instead of the ASCII byte; seq_nt4_table, sketch.c:8-25).  Layout: every read starts on a 128-base chunk; a chunk
 * is 4 x uint64 of codes (base j of a word at bits 2j..2j+1) in `codes` and 4 x uint32 of "not A/C/G/T/U" bits in `amb`
 * (bits beyond the read's end set).  lqcov_packed_chunks gives the chunk count of n reads, lqcov_pack_reads fills
 * caller-owned buffers of 32 / 16 bytes per chunk on the host (n_threads <= 0: up to 16), lqcov_host_alloc / _free hand
 * out page-locked host memory so that the upload runs at PCIe speed.                                               */
uint64_t lqcov_packed_chunks(uint32_t n, const uint64_t *seq_off);
int   lqcov_pack_reads(uint32_t n, const uint8_t *seq, const uint64_t *seq_off, uint64_t *codes, uint32_t *amb, int n_threads);
void *lqcov_host_alloc(size_t bytes);
void  lqcov_host_free(void *p);
int lqcov_part_add_packed(lqcov_handle *h, int part, uint32_t n, const uint64_t *codes, const uint32_t *amb, const uint32_t *lens,
                          const char *names, const uint64_t *name_off);
/* amb == NULL in lqcov_part_add_packed: none of the n reads holds an ambiguous base -- only `codes` crosses PCIe (0.25 B per base) and the
 * device makes the bits beyond the reads' ends itself.  lqcov_packed_ambiguous_reads says which reads of a packed set do hold one
 * (flags[i] = 1: a bit of `amb` below read i's length is set), so that a caller can tell for any range of reads.  (seq_nt4_table,
 * sketch.c:8-25: an ambiguous base resets mm_sketch's k-mer, sketch.c:114.) */
int lqcov_packed_ambiguous_reads(uint32_t n, const uint32_t *amb, const uint32_t *lens, uint8_t *flags);
/* The same reads from DEVICE memory, in shares: share i (share_chunks[i] 128-base chunks, a whole number of reads) starts at chunk
 * i * stride_chunks of codes_dev (4 x u64 per chunk) / amb_dev (4 x u32 per chunk); the shares are copied back to back in
 * share order = read order.  For a host that received the packed reads of a part from its peers (the query-sharded multi-GPU
 * split all-gathers 0.375 B per base over RCCL instead of 16 B per minimizer): lens / names describe every read of the part.
 * amb_dev == NULL: as amb == NULL above.  No counterpart in the reference (its parts come from one file, bseq.c:68-102). */
int lqcov_part_add_packed_shares_dev(lqcov_handle *h, int part, const uint64_t *codes_dev, const uint32_t *amb_dev, uint64_t stride_chunks,
                                     uint32_t n_shares, const uint64_t *share_chunks, uint32_t n, const uint32_t *lens,
                                     const char *names, const uint64_t *name_off);
int lqcov_part_clear(lqcov_handle *h, int part);    /* forget the part's reads and index, keep its device buffers (bench: the same part object every step) */
int lqcov_part_build(lqcov_handle *h, int part);    /* sketch + index (+ mid_occ once): index.c:291-330, map.c:46-54 */
int lqcov_part_map(lqcov_handle *h, int part);      /* == lq_map_file (lqmap.c:852): accumulates into the handle */
int lqcov_part_release(lqcov_handle *h, int part);  /* == mm_idx_destroy (minimap2-coverage.c:457) */
/* .mmi parts.  lqcov_part_dump writes a built part in the reference's layout (append != 0: after the parts already in
 * the file).  lqcov_part_load reads the part that starts at *offset of an .mmi file into a new, built part: returns its
 * id and advances *offset; LQCOV_EOF when no part starts there; the file's k / w / -H must be the handle's.            */
int lqcov_part_dump(lqcov_handle *h, int part, const char *path, int append);               /* index.c:390-426 */
int lqcov_part_load(lqcov_handle *h, const char *path, uint64_t *offset);                   /* index.c:428-479 */
int lqcov_reset(lqcov_handle *h);                   /* zero the accumulators, keep resident reads (bench) */
int lqcov_sync(lqcov_handle *h);                    /* wait for the handle's stream */
/* One part can be built while another is mapped: lqcov_part_add_* / lqcov_part_sketch / lqcov_part_build on one part object
 * from one host thread, lqcov_part_map on another part object from another thread (upload, sketch and index run on a stream
 * and scratch of their own).  mm_idx_reader_read + the mapping of the previous part in a pipeline (minimap2-coverage.c:449-458
 * runs them one after the other).  The mapping lanes size their work space from the HBM that is free when the first part is
 * mapped: tell the engine how much to leave for the part that will be built meanwhile. */
int lqcov_reserve_hbm(lqcov_handle *h, uint64_t bytes);

/* Hand the work space the mapping lanes keep between calls (HIP's stream-ordered pool, kept so that the next part does not
 * pay for it again) back to the device: for a host that needs the HBM for buffers of its own between two parts, e.g. the
 * all-gather buffers of the query-sharded multi-GPU split.  Waits for the device.  No counterpart in the reference. */
int lqcov_workspace_trim(lqcov_handle *h);

/* == main pass 2 up to, not including, printf (minimap2-coverage.c:545-566). */
int lqcov_finish(lqcov_handle *h);
int lqcov_n_queries(const lqcov_handle *h);
/* The engine holds the queries longest first; perm[i] = the caller's index of the i-th query in that order.  Only the
 * per-query arrays of lqcov_accum_export_dev / lqcov_accum_import_dev are in the engine's order; rows, regions, minimizer
 * and chain dumps are in the caller's. */
int lqcov_query_order(lqcov_handle *h, uint32_t *perm, uint32_t n);
int lqcov_get_rows(lqcov_handle *h, lqcov_row *rows, uint32_t n_rows);
int lqcov_get_regions(lqcov_handle *h, const lqcov_region **regs, uint32_t *n_regs, const lqcov_region **mregs, uint32_t *n_mregs);
/* Text of the table, rows in query order (minimap2-coverage.c:567-605). names as in lqcov_set_queries. */
int lqcov_write_table(lqcov_handle *h, const char *out_path);

/* ---- parity / inspection ------------------------------------------------------------------ */
int32_t  lqcov_mid_occ(const lqcov_handle *h);                             /* map.c:50 */
uint64_t lqcov_part_n_minimizers(const lqcov_handle *h, int part);
uint64_t lqcov_part_n_keys(const lqcov_handle *h, int part);
uint64_t lqcov_last_n_anchors(const lqcov_handle *h);
/* What the mapping did with the seed hits (collect_seed_hits, lqmap.c:140-205 / radix_sort_128x, lqmap.c:238), for logs and
 * benchmarks: out[0] = anchors written against the last part (the hits whose (strand, target) can reach a chain), out[1..3] =
 * since lqcov_reset: runs chained in klib's own order of equal-x anchors, the queries that own them, the anchors those
 * queries were sorted by klib's passes for.  No counterpart in the reference (it writes and sorts every hit). */
void lqcov_map_stats(const lqcov_handle *h, uint64_t out[4]);
/* Why runs were left to klib's own order (since lqcov_reset; a run is counted once, by the first reason found -- the rule is in
 * kernels_chain.hpp, TieGroup, and restates where mm_chain_dp, chain.c:41-108, can see the order radix_sort_128x, lqmap.c:238, leaves
 * equal-x anchors in): out[0] a skip was pending when a group of tied candidates began (chain.c:72-74), out[1] a member of the
 * group counts as a skip, out[2] the group's top score is reached by two members (chain.c:69-71: max_j), out[3] the scan broke off
 * (chain.c:73) before a tie partner that would have raised the best score, out[4] two equal-x peaks of one score in the backtrack
 * order (chain.c:102-108), out[5] other.  No counterpart in the reference. */
void lqcov_tie_reasons(const lqcov_handle *h, uint64_t out[6]);
/* The records of a FASTA/FASTQ file as the target reader sees them (kseq_read + the U -> T of kseq2bseq, kseq.h:179-224,
 * bseq.c:56-66): out[0] = records, out[1] = bases, out[2] = a hash over the names, out[3] = a hash over the sequences, out[4] =
 * pieces of the file that had to be parsed again in order (parallel reader only).  mode 0: the streaming reader (one thread,
 * also gzip); mode 1: the reader over the mapped file with n_threads threads and pieces of piece_bytes (plain files only:
 * LQCOV_E_ARG otherwise).  Both must agree on every input -- that is what this call is for (tests); no device needed. */
int lqcov_fastx_digest(const char *path, int mode, int n_threads, uint64_t piece_bytes, uint64_t out[5]);
/* minimizers of the query set / of a part, reference encoding (sketch.c:70-72): xy[2*i], xy[2*i+1];
 * off[n+1] per-read offsets.  Pass NULL buffers to get the total in *n_total. */
int lqcov_get_query_minimizers(lqcov_handle *h, uint64_t *xy, uint64_t *off, uint64_t *n_total);
int lqcov_get_part_minimizers(lqcov_handle *h, int part, uint64_t *xy, uint64_t *off, uint64_t *n_total);
/* chains of the last lqcov_part_map call: 9 int32 per chain
 * (query, rid, rev, score, cnt, qs, qe, rs, re), unordered. */
int lqcov_get_chains(lqcov_handle *h, int32_t *out, uint64_t cap, uint64_t *n_total);

/* Saturated uint16 match counters with the index parts spread over ranks (esterr.c:127-138: once a counter is at 65535 the
 * others depend on the order in which lq_cnt_match met the chains, hit.c:52-88 -- on one handle the engine replays that by
 * itself).  The ranks see a counter reach the limit only in the merged sums; then, for that query (engine order:
 * lqcov_query_order) and every part of the round in part order:
 *   lqcov_part_sat_records   on the rank that mapped the part: the query is chained once more against it, every kept chain
 *                            recorded (records of lqcov_sat_record_bytes() bytes + a pool of counter indices); call with
 *                            recs == NULL to get the two counts in n_out first (the chaining is done once and kept);
 *   lqcov_sat_replay         on every rank, host arithmetic only: the records replayed in the reference's order on the query's
 *                            counters (lqcov_counter_offsets: counters[off[q]] .. counters[off[q + 1] - 1] of the exported
 *                            array) as they stood before the part; lqcov_counter_max: 65535 (the uint16 limit);
 *   lqcov_accum_set_replayed after lqcov_accum_import_dev: these counters take the place of the merged sums, the row carries
 *                            LQCOV_ROW_REPLAYED. */
uint32_t lqcov_sat_record_bytes(void);
uint32_t lqcov_counter_max(const lqcov_handle *h);
int lqcov_counter_offsets(lqcov_handle *h, uint64_t *off /* n_queries + 1 */);
int lqcov_part_sat_records(lqcov_handle *h, int part, uint32_t query, void *recs, uint64_t rec_cap, uint32_t *at, uint64_t at_cap, uint64_t n_out[2]);
int lqcov_sat_replay(lqcov_handle *h, uint32_t query, const void *recs, uint64_t n_recs, const uint32_t *at, uint64_t n_at,
                     uint32_t *counters, uint64_t n_counters);
int lqcov_accum_set_replayed(lqcov_handle *h, uint32_t query, const uint32_t *counters, uint64_t n_counters);

/* Test access to the engine's device-wide primitives (kernels_isort.hpp; they stand where the reference calls klib's
 * radix_sort_128x on a bucket of minimizers, index.c:150-201): a stable sort of n (key, value) pairs by the low `bits` bits of
 * the keys (key_bytes 4: the keys travel as 32-bit words, as for k <= 16; 8: as 64-bit words; vals == NULL: keys only, 4-byte
 * keys), and out[i] = sum of in[0..i-1].  Host arrays in and out. */
int lqcov_debug_sort_pairs(lqcov_handle *h, uint64_t *keys, uint64_t *vals, uint64_t n, unsigned bits, int key_bytes);
int lqcov_debug_scan(lqcov_handle *h, const uint32_t *in, uint64_t *out, uint64_t n);

/* The table text (minimap2-coverage.c:567-605) of rows computed elsewhere: the ranks of a multi-GPU run gather their rows and
 * region pools as they are (lqcov_get_rows / lqcov_get_regions; reg_off / mreg_off rebased onto the concatenated pools) and
 * one rank prints them.  No handle: formatting needs nothing but the rows.  names: n_rows NUL-terminated strings. */
int lqcov_format_rows(int filter_flag, const lqcov_row *rows, uint32_t n_rows, const lqcov_region *regs, const lqcov_region *mregs,
                      const char *names, const uint64_t *name_off, const char *out_path);

/* ---- multi-GPU plumbing (device pointers; torch.distributed/RCCL moves the bytes) ----------- */
/* minimizers of a built part as two device arrays (x = hash<<8|span, y = rid<<32|pos<<1|strand) */
int lqcov_part_minimizers_dev(lqcov_handle *h, int part, const uint64_t **x_dev, const uint64_t **y_dev, uint64_t *n);
/* the same into caller-owned device buffers of `cap` entries each (e.g. the send buffers of an all-gather), with rid_base
 * added to every rid: the part-global index of this rank's first read */
int lqcov_part_minimizers_export_dev(lqcov_handle *h, int part, uint64_t *x_dev, uint64_t *y_dev, uint64_t cap, uint32_t rid_base);
/* sketch only (no index): step 1 of mm_idx_gen (index.c:291-302) on this rank's share of the part */
int lqcov_part_sketch(lqcov_handle *h, int part);
/* replace the part's minimizer set by caller-provided device arrays (rank-concatenated, y-sorted),
 * with the part-global target lengths and names, then (re)build the index from them */
int lqcov_part_build_from_minimizers_dev(lqcov_handle *h, int part, const uint64_t *x_dev, const uint64_t *y_dev, uint64_t n,
                                         uint32_t n_targets, const uint32_t *target_len, const char *names, const uint64_t *name_off);
/* the same from the receive buffers of an all-gather with equally sized send buffers: share i (the minimizers of rank i's reads,
 * share_n[i] <= stride entries, host array) starts at word i * stride of x_dev / y_dev; the shares are copied back to back
 * (read order: mm_idx_gen's order, index.c:291-302) -- no concatenated copy on the caller's side */
int lqcov_part_build_from_minimizer_shares_dev(lqcov_handle *h, int part, const uint64_t *x_dev, const uint64_t *y_dev, uint64_t stride,
                                               uint32_t n_shares, const uint64_t *share_n,
                                               uint32_t n_targets, const uint32_t *target_len, const char *names, const uint64_t *name_off);

/* Index parts on different GPUs == the reference's own -I partitioning (minimap2-coverage.c:449-458) run in
 * parallel.  Parts only interact through (i) mid_occ, frozen from part 0 (map.c:50), (ii) the COVT cap, which
 * drops part p for a query whose lambda/qlen already exceeds 150 (esterr.c:87), and (iii) avg_k, set by the first
 * part that sees the query (esterr.c:93-97).  In distributed mode a handle therefore maps its part with fresh
 * accumulators and no cap; the driver (longqc_amd/multigpu.py) exchanges the per-part accumulators over RCCL,
 * replays (ii) and (iii) in part order, and imports the sums for lqcov_finish().                            */
int lqcov_set_distributed(lqcov_handle *h, int on);
int lqcov_set_mid_occ(lqcov_handle *h, int32_t mid_occ);
int lqcov_accum_sizes(lqcov_handle *h, uint32_t *n_queries, uint64_t *n_counters, uint32_t *n_intervals);
/* copy the accumulators to caller-owned device buffers: lambda/lambda2 [n_queries] u64, avg_k [n_queries] f32,
 * flags [n_queries] u32, counters [n_counters] u32 with their query index counter_owner [n_counters] u32,
 * intervals [n_intervals][3] u32 = (query, start, end) encoded as lqmap.c:69-71.  NULL pointers are skipped. */
int lqcov_accum_export_dev(lqcov_handle *h, uint64_t *lambda_dev, uint64_t *lambda2_dev, float *avg_k_dev, uint32_t *flags_dev,
                           uint32_t *counters_dev, uint32_t *counter_owner_dev, uint32_t *intervals_dev);
int lqcov_accum_import_dev(lqcov_handle *h, const uint64_t *lambda_dev, const uint64_t *lambda2_dev, const float *avg_k_dev, const uint32_t *flags_dev,
                           const uint32_t *counters_dev, const uint32_t *intervals_dev, uint32_t n_intervals);

/* ---- the reference's second binary, `sdust` (low-complexity table of every read) ---------------- */
/* == `sdust [-w W] [-t T] <in.fa|fq[.gz]>` with stdout -> out_path (NULL: stdout), stderr -> err_path: one row per read,
 * name, masked bases, length, masked/length %.3f, meanQ %.3f, #qualities above Q7.            sdust.c:181-222 */
int lqsdust_main(int argc, const char *const *argv, const char *out_path, const char *err_path, int device);
/* buffer level: reads as ASCII (seq_off has n+1 entries; qual NULL or parallel to seq, zero bytes = no qualities);
 * per read the masked bases (sdust_core, sdust.c:136-171), the sum of 10^(-q/10) over its qualities in read order
 * (meanQ = -10 log10(sum / length), lqutils.c:51-58) and getQV(qual, 7) (lqutils.c:61-69).  W in [3, 66].           */
int lqsdust_reads(int device, uint32_t n, const uint8_t *seq, const uint64_t *seq_off, const uint8_t *qual, int W, int T,
                  uint32_t *masked, double *qual_psum, uint32_t *n_above_q7, char *errbuf, size_t errbuf_len);

#ifdef __cplusplus
}
#endif
#endif
