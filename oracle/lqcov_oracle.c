/* oracle/lqcov_oracle.c -- TEST INFRASTRUCTURE ONLY (see lqcov_oracle.h).
 *
 * CPU restatement of LongQC's minimap2-coverage path.  Each function cites the reference
 * file:line (under /root/reference/minimap2-coverage/) whose behaviour it restates.  It is written
 * batch-wise (arrays per index part / per query) rather than as the reference's streaming pipeline,
 * but every observable quantity -- minimizer lists, hit lists, mid_occ, chains, the 9-column table
 * -- is bit-identical to the reference (tests/test_oracle_vs_ref.py).
 *
 * Parity status: PINNED against oracle/_ref (the reference's own sources compiled here).
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <ctype.h>
#include <inttypes.h>
#include <time.h>
#include <zlib.h>
#include "lqcov_oracle.h"

#define U64MAX 0xffffffffffffffffULL
#define COVT 150                         /* minimap2-coverage.h:20 */
#define SEED_TANDEM (1ULL<<42)           /* mmpriv.h:18 */

#define VEC(T) struct { T *a; size_t n, m; }
#define vpush(v, val) do { \
		if ((v).n == (v).m) { (v).m = (v).m ? (v).m * 2 : 16; (v).a = realloc((v).a, (v).m * sizeof(*(v).a)); } \
		(v).a[(v).n++] = (val); \
	} while (0)

static double now_s(void)
{
	struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static double g_timing[4];
void lqo_last_timing(double t[4]) { memcpy(t, g_timing, sizeof(g_timing)); }

void lqo_params_default(lqo_params *p)
{
	memset(p, 0, sizeof(*p));
	p->k = 12; p->w = 5; p->hpc = 0;                /* minimap2-coverage.c:252-266 */
	p->batch_size = 4000000000ULL;                  /* index.c:36 */
	p->idx_mini_batch = 50000000;                   /* index.c:35 */
	p->qry_mini_batch = 500000000;                  /* map.c:40 */
	p->max_gap = 10000; p->min_cnt = 3; p->min_chain_score = 40;   /* minimap2-coverage.c:302-321 */
	p->min_score_med = 40; p->min_score_good = 40;  /* :324-332 (default = m) */
	p->max_chain_skip = 25; p->bw = 500;            /* :362-367, map.c:20 */
	p->max_overhang = 2000; p->min_ovlp = 1000; p->min_coverage = 3; p->min_ratio = 0.4; /* :290-295,:369-388 */
	p->mid_occ_frac = 2e-4f; p->seed = 11;          /* map.c:15-16 */
	p->no_self = 1; p->ava = 0; p->filter_flag = 0;
}

/* ------------------------------------------------------------------------------------------------
 * klib's in-place MSD radix sort (ksort.h:84-134), restated out-of-place.
 *
 * One pass of rs_sort over a sub-array is a token walk: the token sits on a bucket, consumes the
 * next not-yet-consumed element of that bucket's region (in original slot order) and jumps to the
 * bucket that element belongs to; an element is written to the next free slot of its own bucket at
 * the moment it is consumed.  The outer bucket k only advances when its region is full.  The order
 * inside each bucket is therefore the order of consumption -- not the input order: the sort is
 * unstable but deterministic, and callers observe that order for equal keys (lqmap.c:238).
 * Buckets of <= 64 elements are finished by a (stable) insertion sort (ksort.h:87-97,121-127).
 * ---------------------------------------------------------------------------------------------- */
#define RS_MIN 64

#define DEFINE_KLIB_SORT(NAME, T, KEY, KEYBYTES) \
static void ins_##NAME(T *a, size_t n) \
{ \
	size_t i, j; \
	for (i = 1; i < n; ++i) { \
		if (KEY(a[i]) < KEY(a[i-1])) { \
			T t = a[i]; \
			for (j = i; j > 0 && KEY(t) < KEY(a[j-1]); --j) a[j] = a[j-1]; \
			a[j] = t; \
		} \
	} \
} \
static void walk_##NAME(T *a, size_t n, int shift, T *src) \
{ \
	size_t cnt[256], beg[256], rd[256], wr[256], i; \
	int k, c; \
	memset(cnt, 0, sizeof(cnt)); memset(rd, 0, sizeof(rd)); memset(wr, 0, sizeof(wr)); \
	for (i = 0; i < n; ++i) ++cnt[(KEY(a[i]) >> shift) & 0xff]; \
	for (beg[0] = 0, c = 1; c < 256; ++c) beg[c] = beg[c-1] + cnt[c-1]; \
	memcpy(src, a, n * sizeof(T)); \
	for (k = 0; k < 256; ++k) { \
		while (wr[k] < cnt[k]) { \
			T e = src[beg[k] + rd[k]++]; \
			int l = (int)((KEY(e) >> shift) & 0xff); \
			while (l != k) { \
				a[beg[l] + wr[l]++] = e; \
				e = src[beg[l] + rd[l]++]; \
				l = (int)((KEY(e) >> shift) & 0xff); \
			} \
			a[beg[k] + wr[k]++] = e; \
		} \
	} \
	if (shift) { \
		int ns = shift > 8 ? shift - 8 : 0; \
		for (c = 0; c < 256; ++c) { \
			if (cnt[c] > RS_MIN) walk_##NAME(a + beg[c], cnt[c], ns, src + beg[c]); \
			else if (cnt[c] > 1) ins_##NAME(a + beg[c], cnt[c]); \
		} \
	} \
} \
void lqo_sort_##NAME(T *a, size_t n) \
{ \
	if (n <= RS_MIN) ins_##NAME(a, n); \
	else { \
		T *src = (T*)malloc(n * sizeof(T)); \
		walk_##NAME(a, n, ((KEYBYTES) - 1) * 8, src); \
		free(src); \
	} \
}
#define KEY128(e) ((e).x)
#define KEYSELF(e) (e)
DEFINE_KLIB_SORT(128x, lqo_mm128, KEY128, 8)
DEFINE_KLIB_SORT(64, uint64_t, KEYSELF, 8)
DEFINE_KLIB_SORT(32, uint32_t, KEYSELF, 4)

/* ------------------------------------------------------------------------------------------------
 * FASTA/Q records (kseq.h:184-224 semantics) from a whole file held in memory (zlib inflates gz).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { char *buf; size_t len, pos; int last_char; } fx_t;
typedef VEC(char) cstr;
typedef struct { char *name, *seq, *qual; int l_seq, l_qual; } fx_rec;

static int fx_open(fx_t *f, const char *fn)
{
	gzFile g = gzopen(fn, "r");
	size_t cap = 1 << 20;
	int r;
	memset(f, 0, sizeof(*f));
	if (!g) return -1;
	f->buf = (char*)malloc(cap);
	while ((r = gzread(g, f->buf + f->len, (unsigned)(cap - f->len))) > 0) {
		f->len += r;
		if (f->len == cap) { cap *= 2; f->buf = (char*)realloc(f->buf, cap); }
	}
	gzclose(g);
	return 0;
}
static void fx_close(fx_t *f) { free(f->buf); memset(f, 0, sizeof(*f)); }
static inline int fx_getc(fx_t *f) { return f->pos < f->len ? (unsigned char)f->buf[f->pos++] : -1; }

/* ks_getuntil2 with KS_SEP_LINE (kseq.h:92-141): append up to '\n'; strip one trailing '\r' */
static int fx_line(fx_t *f, cstr *s, int append)
{
	size_t i;
	if (!append) s->n = 0;
	if (f->pos >= f->len) return -1;
	for (i = f->pos; i < f->len && f->buf[i] != '\n'; ++i) vpush(*s, f->buf[i]);
	f->pos = i < f->len ? i + 1 : f->len;
	if (s->n > 1 && s->a[s->n - 1] == '\r') --s->n;
	return (int)s->n;
}

/* returns l_seq >= 0, -1 at EOF, -2 on truncated quality (kseq.h:179-224) */
static int fx_next(fx_t *f, cstr *name, cstr *seq, cstr *qual)
{
	int c;
	size_t i;
	if (f->last_char == 0) {
		while ((c = fx_getc(f)) != -1 && c != '>' && c != '@') {}
		if (c == -1) return -1;
		f->last_char = c;
	}
	seq->n = qual->n = name->n = 0;
	if (f->pos >= f->len) return -1;
	for (i = f->pos; i < f->len && !isspace((unsigned char)f->buf[i]); ++i) vpush(*name, f->buf[i]);
	c = i < f->len ? (unsigned char)f->buf[i] : 0;
	f->pos = i < f->len ? i + 1 : f->len;
	if (c != '\n') { cstr cm = {0,0,0}; fx_line(f, &cm, 0); free(cm.a); }
	while ((c = fx_getc(f)) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		vpush(*seq, (char)c);
		fx_line(f, seq, 1);
	}
	if (c == '>' || c == '@') f->last_char = c;
	if (c != '+') return (int)seq->n;
	while ((c = fx_getc(f)) != -1 && c != '\n') {}
	if (c == -1) return -2;
	while (fx_line(f, qual, 1) >= 0 && qual->n < seq->n) {}
	f->last_char = 0;
	if (seq->n != qual->n) return -2;
	return (int)seq->n;
}

typedef VEC(fx_rec) rec_v;

static char *dupn(const char *s, size_t n)
{
	char *r = (char*)malloc(n + 1);
	memcpy(r, s, n); r[n] = 0;
	return r;
}

/* One mini-batch (bseq.c:68-102, frag_mode=0): records until cumulative bases >= chunk. U->T as
 * kseq2bseq (bseq.c:56-66).  Returns number of records appended (0 at EOF). */
static int read_minibatch(fx_t *f, int64_t chunk, rec_v *out, int64_t *bases)
{
	cstr nm = {0,0,0}, sq = {0,0,0}, ql = {0,0,0};
	int64_t size = 0;
	int n = 0, l;
	while ((l = fx_next(f, &nm, &sq, &ql)) >= 0) {
		fx_rec r;
		int i;
		r.name = dupn(nm.a, nm.n); r.seq = dupn(sq.a, sq.n); r.qual = ql.n ? dupn(ql.a, ql.n) : 0;
		r.l_seq = (int)sq.n; r.l_qual = (int)ql.n;
		for (i = 0; i < r.l_seq; ++i) if (r.seq[i] == 'u' || r.seq[i] == 'U') --r.seq[i];
		vpush(*out, r);
		++n; size += r.l_seq;
		if (size >= chunk) break;
	}
	free(nm.a); free(sq.a); free(ql.a);
	if (bases) *bases = size;
	return n;
}

static void free_recs(rec_v *v)
{
	size_t i;
	for (i = 0; i < v->n; ++i) { free(v->a[i].name); free(v->a[i].seq); free(v->a[i].qual); }
	free(v->a); v->a = 0; v->n = v->m = 0;
}

/* ------------------------------------------------------------------------------------------------
 * mm_sketch (sketch.c:76-142), hash64 (sketch.c:27-37), seq_nt4_table (sketch.c:8-25)
 * ---------------------------------------------------------------------------------------------- */
static unsigned char nt4[256];
static int nt4_ready = 0;
static void nt4_init(void)
{
	if (nt4_ready) return;
	memset(nt4, 4, 256);
	nt4[0] = 0; nt4[1] = 1; nt4[2] = 2; nt4[3] = 3;   /* sketch.c:9 -- raw codes 0..3 map to themselves */
	nt4['A'] = nt4['a'] = 0; nt4['C'] = nt4['c'] = 1; nt4['G'] = nt4['g'] = 2;
	nt4['T'] = nt4['t'] = 3; nt4['U'] = nt4['u'] = 3;
	nt4_ready = 1;
}

static inline uint64_t mix64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

typedef struct {
	lqo_mm128 **out; size_t *n, *cap;
} sk_sink;
static inline void sk_emit(sk_sink *s, lqo_mm128 v)
{
	if (*s->n == *s->cap) { *s->cap = *s->cap ? *s->cap * 2 : 64; *s->out = (lqo_mm128*)realloc(*s->out, *s->cap * sizeof(lqo_mm128)); }
	(*s->out)[(*s->n)++] = v;
}

/* every ring entry other than `best` that carries best's key, oldest -> newest; the slot `skip`
 * (the one just written) is left out when skip >= 0 (sketch.c:116-121 vs :131-136) */
static void sk_emit_ties(sk_sink *s, const lqo_mm128 *ring, int w, int slot, int skip_cur, lqo_mm128 best)
{
	int j;
	for (j = slot + 1; j < w; ++j)
		if (ring[j].x == best.x && ring[j].y != best.y) sk_emit(s, ring[j]);
	for (j = 0; j < (skip_cur ? slot : slot + 1); ++j)
		if (ring[j].x == best.x && ring[j].y != best.y) sk_emit(s, ring[j]);
}

void lqo_sketch(const char *seq, int len, int w, int k, uint32_t rid, int hpc,
                lqo_mm128 **out, size_t *n, size_t *cap)
{
	const uint64_t mask = (1ULL << 2 * k) - 1, shift1 = 2 * (k - 1);
	uint64_t fw = 0, rv = 0;
	lqo_mm128 ring[256], best = { U64MAX, U64MAX };
	int i, j, l = 0, slot = 0, best_slot = 0, span = 0;
	int runq[32], rq_front = 0, rq_count = 0;
	sk_sink sink = { out, n, cap };
	nt4_init();
	if (len <= 0) return;
	memset(ring, 0xff, sizeof(lqo_mm128) * w);
	for (i = 0; i < len; ++i) {
		int c = nt4[(unsigned char)seq[i]];
		lqo_mm128 cur = { U64MAX, U64MAX };
		if (c < 4) {
			int z;
			if (hpc) {                                   /* sketch.c:93-104 */
				int run = 1;
				if (i + 1 < len && nt4[(unsigned char)seq[i + 1]] == c) {
					for (run = 2; i + run < len; ++run)
						if (nt4[(unsigned char)seq[i + run]] != c) break;
					i += run - 1;
				}
				runq[(rq_count++ + rq_front) & 0x1f] = run;
				span += run;
				if (rq_count > k) { span -= runq[rq_front++]; rq_front &= 0x1f; --rq_count; }
			} else span = l + 1 < k ? l + 1 : k;
			fw = (fw << 2 | (uint64_t)c) & mask;
			rv = (rv >> 2) | (3ULL ^ (uint64_t)c) << shift1;
			if (fw == rv) continue;                      /* palindromic k-mer: no ring slot at all (sketch.c:107) */
			z = fw < rv ? 0 : 1;
			++l;
			if (l >= k && span < 256) {
				cur.x = mix64(z ? rv : fw, mask) << 8 | (uint64_t)span;
				cur.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint32_t)z;
			}
		} else { l = 0; rq_count = rq_front = 0; span = 0; }   /* sketch.c:114 */
		ring[slot] = cur;
		if (l == w + k - 1 && best.x != U64MAX)         /* first full window: older ties of the running min (sketch.c:116-121) */
			sk_emit_ties(&sink, ring, w, slot, 1, best);
		if (cur.x <= best.x) {                           /* sketch.c:122-124 */
			if (l >= w + k && best.x != U64MAX) sk_emit(&sink, best);
			best = cur; best_slot = slot;
		} else if (slot == best_slot) {                  /* running min leaves the window (sketch.c:125-137) */
			if (l >= w + k - 1 && best.x != U64MAX) sk_emit(&sink, best);
			best.x = U64MAX;
			for (j = slot + 1; j < w; ++j) if (best.x >= ring[j].x) { best = ring[j]; best_slot = j; }
			for (j = 0; j <= slot; ++j)    if (best.x >= ring[j].x) { best = ring[j]; best_slot = j; }
			if (l >= w + k - 1 && best.x != U64MAX)
				sk_emit_ties(&sink, ring, w, slot, 0, best);
		}
		if (++slot == w) slot = 0;
	}
	if (best.x != U64MAX) sk_emit(&sink, best);          /* sketch.c:140-141 */
}

/* ------------------------------------------------------------------------------------------------
 * One index part (index.c:229-330 semantics): hash -> occurrences sorted by y.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
	uint32_t n_seq;
	char **name;            /* borrowed from recs */
	uint32_t *len;
	uint64_t tot_len;
	size_t n_mini;          /* M_t */
	uint64_t *pos;          /* y values grouped by key, ascending inside a group (index.c:188) */
	size_t n_keys;          /* K_t */
	uint64_t *key;          /* distinct x>>8, ascending */
	uint64_t *start;        /* offset into pos[] */
	uint32_t *cnt;
	rec_v recs;             /* owns the strings */
} part_t;

static int cmp_hash_y(const void *a_, const void *b_)
{
	const lqo_mm128 *a = (const lqo_mm128*)a_, *b = (const lqo_mm128*)b_;
	uint64_t ha = a->x >> 8, hb = b->x >> 8;
	if (ha != hb) return ha < hb ? -1 : 1;
	return a->y < b->y ? -1 : a->y > b->y;
}

static void part_free(part_t *pt)
{
	free(pt->name); free(pt->len); free(pt->pos); free(pt->key); free(pt->start); free(pt->cnt);
	free_recs(&pt->recs);
	memset(pt, 0, sizeof(*pt));
}

/* Reads the next part: mini-batches of min(50M, I) bases while the running total <= I
 * (index.c:244,311-316; checked before each mini-batch).  Returns 0 when the file is exhausted. */
static int part_read(fx_t *f, const lqo_params *p, part_t *pt)
{
	uint64_t sum_len = 0;
	int64_t chunk = (uint64_t)p->idx_mini_batch < p->batch_size ? p->idx_mini_batch : (int64_t)p->batch_size;
	chunk = (int)chunk;
	memset(pt, 0, sizeof(*pt));
	for (;;) {
		int64_t b;
		if (sum_len > p->batch_size) break;
		if (read_minibatch(f, chunk, &pt->recs, &b) == 0) break;
		sum_len += b;
	}
	if (pt->recs.n == 0) return 0;
	return 1;
}

static void part_build(const lqo_params *p, part_t *pt)
{
	lqo_mm128 *mv = 0;
	size_t n = 0, cap = 0, i, j;
	pt->n_seq = (uint32_t)pt->recs.n;
	pt->name = (char**)malloc(sizeof(char*) * pt->n_seq);
	pt->len = (uint32_t*)malloc(sizeof(uint32_t) * pt->n_seq);
	for (i = 0; i < pt->n_seq; ++i) {
		fx_rec *r = &pt->recs.a[i];
		pt->name[i] = r->name; pt->len[i] = (uint32_t)r->l_seq; pt->tot_len += r->l_seq;
		if (r->l_seq > 0) lqo_sketch(r->seq, r->l_seq, p->w, p->k, (uint32_t)i, p->hpc, &mv, &n, &cap);  /* index.c:293-300 */
	}
	qsort(mv, n, sizeof(lqo_mm128), cmp_hash_y);        /* net effect of worker_post (index.c:150-201) */
	pt->n_mini = n;
	pt->pos = (uint64_t*)malloc(8 * (n ? n : 1));
	for (i = 0, pt->n_keys = 0; i < n; ++i) if (i == 0 || mv[i].x >> 8 != mv[i-1].x >> 8) ++pt->n_keys;
	pt->key = (uint64_t*)malloc(8 * (pt->n_keys ? pt->n_keys : 1));
	pt->start = (uint64_t*)malloc(8 * (pt->n_keys ? pt->n_keys : 1));
	pt->cnt = (uint32_t*)malloc(4 * (pt->n_keys ? pt->n_keys : 1));
	for (i = 0, j = 0; i < n; ++i) {
		pt->pos[i] = mv[i].y;
		if (i == 0 || mv[i].x >> 8 != mv[i-1].x >> 8) { pt->key[j] = mv[i].x >> 8; pt->start[j] = i; pt->cnt[j] = 0; ++j; }
		++pt->cnt[j-1];
	}
	free(mv);
}

/* mm_idx_get (index.c:69-86) */
static const uint64_t *part_get(const part_t *pt, uint64_t minier, int *n)
{
	size_t lo = 0, hi = pt->n_keys;
	*n = 0;
	while (lo < hi) {
		size_t mid = (lo + hi) >> 1;
		if (pt->key[mid] < minier) lo = mid + 1; else hi = mid;
	}
	if (lo < pt->n_keys && pt->key[lo] == minier) { *n = (int)pt->cnt[lo]; return pt->pos + pt->start[lo]; }
	return 0;
}

static int cmp_u32(const void *a, const void *b)
{
	uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
	return x < y ? -1 : x > y;
}

/* mm_idx_cal_max_occ (index.c:123-144): (kth smallest occurrence count) + 1 */
static int32_t part_mid_occ(const part_t *pt, float f)
{
	uint32_t *a, thres;
	size_t n = pt->n_keys;
	if (f <= 0.) return INT32_MAX;
	if (n == 0) return 1;   /* reference reads out of bounds here (ks_ksmall on an empty array); any value is unobservable */
	a = (uint32_t*)malloc(4 * n);
	memcpy(a, pt->cnt, 4 * n);
	qsort(a, n, 4, cmp_u32);
	thres = a[(uint32_t)((1. - f) * n)] + 1;
	free(a);
	return (int32_t)thres;
}

/* ------------------------------------------------------------------------------------------------
 * Per-query persistent state (minimap2-coverage.c:406-444)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t start, end; } subc_t;
typedef VEC(subc_t) subc_v;

typedef struct {
	uint64_t lambda, lambda2;
	float avg_k;
	uint32_t n_cnt;       /* unfiltered minimizer count (sizes m_cnts, minimap2-coverage.c:422) */
	uint16_t *cnt;
	subc_v ovlp;
} qstate_t;

/* chain record after mm_gen_regs (hit.c:52-88) */
typedef struct {
	int32_t cnt, rid, qs, qe, rs, re, as, score0;
	uint32_t rev, hash;
} reg_t;

static inline uint32_t x31_hash(const char *s)      /* khash.h:383-388 */
{
	uint32_t h = (uint32_t)*s;
	if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
	return h;
}
static inline uint32_t wang_hash(uint32_t key)      /* khash.h:400-409 */
{
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3);
	key ^= (key >> 6);   key += ~(key << 11); key ^= (key >> 16);
	return key;
}
static inline uint64_t mix64_full(uint64_t key)     /* hit.c:40-50 */
{
	key = (~key + (key << 21)); key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)); key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)); key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}

static inline int ilog2_32(uint32_t v)              /* chain.c:8-20 */
{
	int r = -1;
	while (v) { v >>= 1; ++r; }
	return r;
}

/* ---- mm_chain_dp (chain.c:22-157) with n_segs=1, is_cdna=0, over a[lo..hi) of one array.
 * Appends kept chains to (u, b): u = score<<32|cnt, b = the chain's anchors in ascending order.
 * avg_qspan is supplied by the caller (global over the query's full anchor list, chain.c:37-38). */
typedef VEC(uint64_t) u64_v;
typedef VEC(lqo_mm128) mm_v;

unsigned long long lqo_tie_reason[8];   /* debug statistics (LQO_TIE_DEBUG): why a run was called observable */
static void chain_dp_range_s(const lqo_params *P, float avg_qspan, const lqo_mm128 *a, int64_t n, u64_v *u_out, mm_v *b_out, int *peak_tie, int *band_tie)
{
	int32_t *f, *p, *t, *v, n_u, n_v, k;
	int64_t i, j, st = 0;
	uint64_t *u;
	const int max_dist_x = P->max_gap, max_dist_y = P->max_gap, bw = P->bw, max_skip = P->max_chain_skip;
	const int min_cnt = P->min_cnt, min_sc = P->min_chain_score;
	if (n == 0) return;
	f = (int32_t*)malloc(n * 4); p = (int32_t*)malloc(n * 4); t = (int32_t*)calloc(n, 4); v = (int32_t*)malloc(n * 4);
	for (i = 0; i < n; ++i) {                           /* chain.c:41-81 */
		uint64_t ri = a[i].x;
		int64_t max_j = -1;
		int32_t qi = (int32_t)a[i].y, q_span = a[i].y >> 32 & 0xff;
		int32_t max_f = q_span, n_skip = 0;
		int have_act = 0, grp_loud = 0, grp_s0 = 0, grp_tm = 0, grp_dup = 0, grp_R = 0; uint64_t act_x = 0; int32_t grp_m = 0, grp_top = 0;
		while (st < i && ri - a[st].x > (uint64_t)max_dist_x) ++st;
		for (j = i - 1; j >= st; --j) {
			int64_t dr = ri - a[j].x;
			int32_t dq = qi - (int32_t)a[j].y, dd, sc, log_dd, min_d;
			if (dr == 0 || dq <= 0) continue;
			if (dq > max_dist_y || dq > max_dist_x) continue;
			dd = dr > dq ? dr - dq : dq - dr;
			if (dd > bw) continue;
			min_d = dq < dr ? dq : dr;
			sc = min_d > q_span ? q_span : dq < dr ? dq : dr;
			log_dd = dd ? ilog2_32(dd) : 0;
			sc -= (int)(dd * .01 * avg_qspan) + (log_dd >> 1);
			sc += f[j];
			if (band_tie) {
				/* Candidates of equal x inside the band of one scan (a "group"; they are neighbours in the array).  What a candidate
				 * does: raise the best score (sc > max_f: max_f, max_j, one skip forgiven), or count as a skip (t[j] == i), or
				 * neither ("quiet") -- and leave its mark (below), which every scanned candidate does whatever the order.  Quiet
				 * members (sc <= the best score before the group, which only grows; t[j] != i) commute with everything.  A group
				 * in which no member raises the best score (sc <= the best before the group for all of them) commutes as a whole:
				 * its marked members count as skips one by one in any order, the scan ends -- if it does -- at the same count, and
				 * nothing that could have raised the score is left unscanned.  With a member that raises it, two or more loud members
				 * still commute when no skip is pending before the group and none of them counts as one (then n_skip stays 0 in
				 * any order) and the highest score among them is reached by one member only (then max_f, max_j end the same). */
				if (!(have_act && a[j].x == act_x)) {
					if (have_act && grp_R >= 1 && grp_loud >= 2 && (!grp_s0 || grp_tm || grp_dup)) { *band_tie = 1; ++lqo_tie_reason[grp_dup ? 0 : grp_tm ? 1 : 2]; }
					have_act = 1; act_x = a[j].x; grp_m = max_f; grp_s0 = n_skip == 0; grp_loud = 0; grp_tm = 0; grp_dup = 0; grp_R = 0; grp_top = INT32_MIN;
				}
				if (!(sc <= grp_m && t[j] != i)) {
					++grp_loud;
					if (sc > grp_m) ++grp_R;
					if (t[j] == i) grp_tm = 1;
					if (sc > grp_top) grp_top = sc, grp_dup = 0; else if (sc == grp_top) grp_dup = 1;
				}
			}
			if (sc > max_f) {
				max_f = sc, max_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == i) {
				if (++n_skip > max_skip) {
					if (band_tie) {                         /* the scan ends at j (a loud member that counts as a skip): in another order a tie partner of j would have been scanned before it */
						int64_t jj;
						for (jj = j - 1; jj >= st && a[jj].x == a[j].x; --jj) {
							int64_t dr2 = ri - a[jj].x; int32_t dq2 = qi - (int32_t)a[jj].y, dd2, sc2, md2;
							if (dr2 == 0 || dq2 <= 0 || dq2 > max_dist_y) continue;
							dd2 = dr2 > dq2 ? dr2 - dq2 : dq2 - dr2;
							if (dd2 > bw) continue;
							md2 = dq2 < dr2 ? dq2 : dr2;
							sc2 = (md2 > q_span ? q_span : md2) - ((int)(dd2 * .01 * avg_qspan) + ((dd2 ? ilog2_32(dd2) : 0) >> 1)) + f[jj];
							if (sc2 > grp_m) { *band_tie = 1; ++lqo_tie_reason[3]; }      /* a member that would have raised the best score, had it come before the one that ended the scan */
						}
					}
					break;
				}
			}
			if (p[j] >= 0) t[p[j]] = i;
		}
		if (band_tie && have_act && grp_R >= 1 && grp_loud >= 2 && (!grp_s0 || grp_tm || grp_dup)) { *band_tie = 1; ++lqo_tie_reason[grp_dup ? 0 : grp_tm ? 1 : 2]; }   /* the scan's last group */
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
	}
	memset(t, 0, n * 4);                                /* chain.c:84-101 */
	for (i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
	for (i = n_u = 0; i < n; ++i) if (t[i] == 0 && v[i] >= min_sc) ++n_u;
	if (n_u == 0) { free(f); free(p); free(t); free(v); return; }
	u = (uint64_t*)malloc(n_u * 8);
	for (i = n_u = 0; i < n; ++i) {
		if (t[i] == 0 && v[i] >= min_sc) {
			j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];
			if (j < 0) j = i;
			u[n_u++] = (uint64_t)f[j] << 32 | j;
		}
	}
	lqo_sort_64(u, n_u);                                /* chain.c:102-106 */
	if (peak_tie)                                       /* two anchors of equal x end chains with the same peak score: the order of the backtracks follows their array order */
		for (i = 1; i < n_u; ++i)
			if (u[i] >> 32 == u[i-1] >> 32 && (int32_t)u[i] != (int32_t)u[i-1] && a[(int32_t)u[i]].x == a[(int32_t)u[i-1]].x) { *peak_tie = 1; ++lqo_tie_reason[4]; }
	for (i = 0; i < n_u >> 1; ++i) { uint64_t tt = u[i]; u[i] = u[n_u - i - 1], u[n_u - i - 1] = tt; }
	memset(t, 0, n * 4);                                /* chain.c:108-125 backtrack */
	for (i = n_v = k = 0; i < n_u; ++i) {
		int32_t n_v0 = n_v, k0 = k;
		j = (int32_t)u[i];
		do { v[n_v++] = j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
		if (j < 0) {
			if (n_v - n_v0 >= min_cnt) u[k++] = u[i] >> 32 << 32 | (n_v - n_v0);
		} else if ((int32_t)(u[i] >> 32) - f[j] >= min_sc) {
			if (n_v - n_v0 >= min_cnt) u[k++] = ((u[i] >> 32) - f[j]) << 32 | (n_v - n_v0);
		}
		if (k0 == k) n_v = n_v0;
	}
	n_u = k;
	for (i = 0, k = 0; i < n_u; ++i) {                  /* chain.c:131-137 */
		int32_t k0 = k, ni = (int32_t)u[i];
		vpush(*u_out, u[i]);
		for (j = 0; j < ni; ++j) { vpush(*b_out, a[v[k0 + (ni - j - 1)]]); ++k; }
	}
	free(f); free(p); free(t); free(v); free(u);
}

static void chain_dp_range(const lqo_params *P, float avg_qspan, const lqo_mm128 *a, int64_t n, u64_v *u_out, mm_v *b_out)
{
	chain_dp_range_s(P, avg_qspan, a, n, u_out, b_out, 0, 0);
}

/* ---- "does the order of equal-x anchors matter in this (strand, rid) run?"  (the GPU engine's question: it sorts with any correct
 * sort and takes klib's order only where it can be observed).  Anchors of equal x never chain to each other (dr == 0, chain.c:52).
 * For a later anchor i two candidates A, B of equal x are both inside the band only if |dq_A - dq_B| <= 2 bw (chain.c:55), i.e.
 * |y_A - y_B| <= 2 bw; a candidate outside the band is stepped over before any state changes (the `continue`s of chain.c:52-56
 * precede chain.c:69-76).  So if all members of every tie group are more than 2 bw apart in y, every scan sees the same sequence
 * of effective candidates whatever the order inside the groups, and f, p, v are the same per anchor.  The order of the backtracks
 * (chain.c:102-125) follows (score, array index): it can differ only if two members of a group are peaks of equal score.  Returns
 * 1 if the first condition fails (static part). */
static int run_ties_close(const lqo_params *P, const lqo_mm128 *a, int64_t n)
{
	int64_t i, j, k;
	for (i = 0; i < n; i = j) {
		for (j = i + 1; j < n && a[j].x == a[i].x; ++j) {}
		for (k = i; k < j; ++k) {
			int64_t l;
			for (l = k + 1; l < j; ++l) {
				int64_t d = (int64_t)(int32_t)a[k].y - (int64_t)(int32_t)a[l].y;
				if (d < 0) d = -d;
				if (d <= 2 * (int64_t)P->bw) return 1;
			}
		}
	}
	return 0;
}
/* An upper bound, whatever the order of equal-x anchors, of the score of any chain inside a[0..n): the heaviest path (by
 * spans) through the pairs that can be chained at all (chain.c:52-56).  A chain's score is at most the sum of its anchors'
 * spans (chain.c:57-67), so a run whose bound is below min_sc yields nothing in any order. */
static int run_score_bound(const lqo_params *P, const lqo_mm128 *a, int64_t n)
{
	int32_t *w = (int32_t*)malloc(n * 4), best = 0;
	int64_t i, j, st = 0;
	for (i = 0; i < n; ++i) {
		int32_t qi = (int32_t)a[i].y, m = 0;
		while (st < i && a[i].x - a[st].x > (uint64_t)P->max_gap) ++st;
		for (j = i - 1; j >= st; --j) {
			int64_t dr = a[i].x - a[j].x;
			int32_t dq = qi - (int32_t)a[j].y, dd;
			if (dr == 0 || dq <= 0 || dq > P->max_gap) continue;
			dd = dr > dq ? dr - dq : dq - dr;
			if (dd > P->bw) continue;
			if (w[j] > m) m = w[j];
		}
		w[i] = m + (int32_t)(a[i].y >> 32 & 0xff);
		if (w[i] > best) best = w[i];
	}
	free(w);
	return best;
}
static uint64_t g_tie_stats[12];
void lqo_tie_stats(uint64_t out[12]) { memcpy(out, g_tie_stats, sizeof(g_tie_stats)); }
void lqo_tie_stats_reset(void) { memset(g_tie_stats, 0, sizeof(g_tie_stats)); }

/* chain.c:139-155: order chains by the x of their first anchor (klib 128x sort on (x, k<<32|i)) */
static void chains_order_by_x(u64_v *u, mm_v *b)
{
	size_t n_u = u->n, i, k;
	lqo_mm128 *w, *nb;
	uint64_t *u2;
	if (n_u == 0) return;
	w = (lqo_mm128*)malloc(n_u * sizeof(lqo_mm128));
	for (i = k = 0; i < n_u; ++i) { w[i].x = b->a[k].x; w[i].y = (uint64_t)k << 32 | i; k += (int32_t)u->a[i]; }
	lqo_sort_128x(w, n_u);
	u2 = (uint64_t*)malloc(n_u * 8);
	nb = (lqo_mm128*)malloc((b->n ? b->n : 1) * sizeof(lqo_mm128));
	for (i = k = 0; i < n_u; ++i) {
		int32_t j = (int32_t)w[i].y, n = (int32_t)u->a[j];
		u2[i] = u->a[j];
		memcpy(&nb[k], &b->a[w[i].y >> 32], n * sizeof(lqo_mm128));
		k += n;
	}
	memcpy(u->a, u2, n_u * 8);
	memcpy(b->a, nb, b->n * sizeof(lqo_mm128));
	free(w); free(u2); free(nb);
}

/* mm_gen_regs + mm_reg_set_coor (hit.c:23-38,52-88) */
static reg_t *gen_regs(uint32_t hash, int qlen, int n_u, const uint64_t *u, const lqo_mm128 *a)
{
	lqo_mm128 *z, tmp;
	reg_t *r;
	int i, k;
	if (n_u == 0) return 0;
	z = (lqo_mm128*)malloc(n_u * 16);
	for (i = k = 0; i < n_u; ++i) {
		uint32_t h = (uint32_t)mix64_full((mix64_full(a[k].x) + mix64_full(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (int32_t)u[i];
		k += (int32_t)u[i];
	}
	lqo_sort_128x(z, n_u);
	/* LQO_REGS_ASCENDING: a test switch that leaves out the reversal -- the chains then reach lq_cnt_match in the opposite order,
	 * which only saturated counters can tell (tests use it to show that an input does exercise the order) */
	if (!getenv("LQO_REGS_ASCENDING"))
		for (i = 0; i < n_u >> 1; ++i) tmp = z[i], z[i] = z[n_u-1-i], z[n_u-1-i] = tmp;
	r = (reg_t*)calloc(n_u, sizeof(reg_t));
	for (i = 0; i < n_u; ++i) {
		reg_t *ri = &r[i];
		int32_t kk, q_span;
		ri->score0 = z[i].x >> 32;
		ri->hash = (uint32_t)z[i].x;
		ri->cnt = (int32_t)z[i].y;
		ri->as = z[i].y >> 32;
		kk = ri->as; q_span = (int32_t)(a[kk].y >> 32 & 0xff);
		ri->rev = a[kk].x >> 63;
		ri->rid = a[kk].x << 1 >> 33;
		ri->rs = (int32_t)a[kk].x + 1 > q_span ? (int32_t)a[kk].x + 1 - q_span : 0;
		ri->re = (int32_t)a[kk + ri->cnt - 1].x + 1;
		if (!ri->rev) {
			ri->qs = (int32_t)a[kk].y + 1 - q_span;
			ri->qe = (int32_t)a[kk + ri->cnt - 1].y + 1;
		} else {
			ri->qs = qlen - ((int32_t)a[kk + ri->cnt - 1].y + 1);
			ri->qe = qlen - ((int32_t)a[kk].y + 1 - q_span);
		}
	}
	free(z);
	return r;
}

/* esterr.c:17-38 */
static inline int32_t fwd_qpos(int32_t qlen, const lqo_mm128 *a)
{
	int32_t x = (int32_t)a->y, q_span = a->y >> 32 & 0xff;
	if (a->x >> 63) x = qlen - 1 - (x + 1 - q_span);
	return x;
}
static int mini_idx(int qlen, const lqo_mm128 *a, int32_t n, const uint64_t *mini_pos)
{
	int32_t x = fwd_qpos(qlen, a), L = 0, R = n - 1;
	while (L <= R) {
		int32_t m = ((uint64_t)L + R) >> 1, y = (int32_t)mini_pos[m];
		if (y < x) L = m + 1; else if (y > x) R = m - 1; else return m;
	}
	return -1;
}

/* Width of the match counters: uint16 in the reference (UINT16_MAX, esterr.c:130,136).  LQO_CNT_BITS=b (2..16) narrows them to
 * b bits -- a test hook (the product has the same one, LQCOV_TEST_CNT_BITS) that brings the saturated regime, where the result
 * depends on the order of the chains (hit.c:52-88), within reach of small inputs. */
static uint16_t lqo_cnt_max(void)
{
	const char *e = getenv("LQO_CNT_BITS");
	int b = e ? atoi(e) : 16;
	if (b < 2 || b > 16) b = 16;
	return (uint16_t)((1u << b) - 1);
}

/* lq_cnt_match (esterr.c:72-140) */
static void cnt_match(const lqo_params *P, const part_t *pt, int qlen, int n_regs, const reg_t *regs, const lqo_mm128 *a,
                      int32_t n, const uint64_t *mini_pos, qstate_t *qs_, subc_v *cv)
{
	int i;
	uint64_t sum_k = 0;
	uint32_t qs, qe, rs, re, rl, hang5, hang3;
	uint16_t min_sc_m = (uint16_t)P->min_score_med, min_sc_g = (uint16_t)P->min_score_good;  /* packed p<<16|q, lqmap.c:841 */
	const uint16_t cmax = lqo_cnt_max();
	if (n == 0) return;
	if (qs_->lambda / qlen > COVT && qs_->avg_k != 0.0) return;   /* esterr.c:87-91; the frac branch cannot trigger */
	if (qs_->avg_k == 0.0) {
		for (i = 0; i < n; ++i) sum_k += mini_pos[i] >> 32 & 0xff;
		qs_->avg_k = (float)sum_k / n;
	}
	for (i = 0; i < n_regs; ++i) {
		const reg_t *r = &regs[i];
		int32_t st, j, k;
		int flag = 0;
		subc_t s;
		if (r->cnt == 0) continue;
		st = mini_idx(qlen, r->rev ? &a[r->as + r->cnt - 1] : &a[r->as], n, mini_pos);
		if (st < 0) continue;
		rl = pt->len[r->rid];
		qs = r->qs; qe = r->qe; rs = r->rs; re = r->re;
		hang5 = qs < rs ? qs : rs;
		hang3 = qlen - qe < rl - re ? qlen - qe : rl - re;
		if ((qe - qs) < (qe - qs + hang5 + hang3) * P->min_ratio || hang5 > (uint32_t)P->max_overhang || hang3 > (uint32_t)P->max_overhang)
			continue;
		qs_->lambda += (qe - qs + 1);
		if (r->score0 >= min_sc_m) flag |= 0x2;
		s.start = qs << 3 | flag;
		flag |= 0x1;
		s.end = qe << 3 | flag;
		vpush(*cv, s);
		if (r->score0 < min_sc_g) continue;
		qs_->lambda2 += (qe - qs + 1);
		if (qs_->cnt[st] < cmax) qs_->cnt[st]++;
		for (k = 1, j = st + 1; j < n && k < r->cnt; ++j) {
			int32_t x = fwd_qpos(qlen, r->rev ? &a[r->as + r->cnt - 1 - k] : &a[r->as + k]);
			if (x == (int32_t)mini_pos[j]) {
				++k;
				if (qs_->cnt[st] < cmax) qs_->cnt[j] = (uint16_t)((qs_->cnt[j] + 1) & cmax);   /* sic: guard reads [st], esterr.c:136 -- cnt[j] wraps */
			}
		}
	}
}

/* filter_redundant_coords (lqmap.c:25-100) */
static void filter_redundant(subc_v *v, subc_v *cv, uint32_t min_cov)
{
	size_t i, j;
	subc_v mc = {0,0,0};
	uint32_t med_start = 0, med_cov = 0, *vc;
	size_t nvc = 0;
	if (cv->n == 0) return;
	vc = (uint32_t*)malloc(8 * cv->n);
	for (i = 0; i < cv->n; ++i) { vc[nvc++] = cv->a[i].start; vc[nvc++] = cv->a[i].end; }
	lqo_sort_32(vc, nvc);
	for (j = 0; j < nvc; ++j) {
		uint32_t old = med_cov;
		if (vc[j] & 2) {
			if (vc[j] & 1) { if (vc[j] & 4) med_cov -= min_cov; else --med_cov; }
			else           { if (vc[j] & 4) med_cov += min_cov; else ++med_cov; }
		}
		if (old < min_cov && med_cov >= min_cov) med_start = vc[j];
		else if (old >= min_cov && med_cov < min_cov) {
			uint32_t mlen = (vc[j] >> 3) - med_start;        /* sic: mixes decoded and encoded units (lqmap.c:63) */
			if (mlen > 0) {
				subc_t m, marker;
				m.start = med_start; m.end = vc[j];
				vpush(mc, m);
				marker.start = med_start | 0x4; marker.end = vc[j] | 0x4;
				vpush(*v, marker);
			}
		}
	}
	free(vc);
	for (i = 0; i < cv->n; ++i) {
		int flag = 0;
		if (!(cv->a[i].start & 4))
			for (j = 0; j < mc.n; ++j)
				if (cv->a[i].start >= mc.a[j].start && cv->a[i].end <= mc.a[j].end) flag |= 1;
		if (!flag) vpush(*v, cv->a[i]);
	}
	free(mc.a);
}

/* compute_reliable_region (lqutils.c:83-155) */
static void reliable_region(const subc_v *v, uint32_t min_cov, subc_v *coords, subc_v *mcoords)
{
	size_t j, nvc = 0;
	uint32_t start = 0, cov = 0, med_start = 0, med_cov = 0, *vc;
	vc = (uint32_t*)malloc(8 * (v->n ? v->n : 1));
	for (j = 0; j < v->n; ++j) { vc[nvc++] = v->a[j].start; vc[nvc++] = v->a[j].end; }
	lqo_sort_32(vc, nvc);
	for (j = 0; j < nvc; ++j) {
		uint32_t old_cov = cov, old_med = med_cov, e = vc[j];
		if (e & 1) {
			--cov;
			if (e & 2) { if (e & 4) { med_cov -= min_cov; cov -= (min_cov - 1); } else --med_cov; }
		} else {
			++cov;
			if (e & 2) { if (e & 4) { med_cov += min_cov; cov += (min_cov - 1); } else ++med_cov; }
		}
		if (old_cov < min_cov && cov >= min_cov) {
			start = e >> 3;
			if (old_med < min_cov && med_cov >= min_cov) med_start = e >> 3;
		} else if (old_cov >= min_cov && cov < min_cov) {
			if ((e >> 3) - start > 0) { subc_t c; c.start = start; c.end = e >> 3; vpush(*coords, c); }
			if (old_med >= min_cov && med_cov < min_cov)
				if ((e >> 3) - med_start > 0) { subc_t c; c.start = med_start; c.end = e >> 3; vpush(*mcoords, c); }
		} else if (old_med < min_cov && med_cov >= min_cov) {
			med_start = e >> 3;
		} else if (old_med >= min_cov && med_cov < min_cov) {
			if ((e >> 3) - med_start > 0) { subc_t c; c.start = med_start; c.end = e >> 3; vpush(*mcoords, c); }
		}
	}
	free(vc);
}

/* meanQ (lqutils.c:26-58).  q2p[] holds 127 15-decimal literals for 10^(-q/10), q = 0..126.  They equal
 * 10^(-q/10) rounded to 15 decimals except for eight entries (q = 34, 39, 58, 62, 67, 71, 72, 82) that are one
 * unit of the 15th decimal higher; the table is rebuilt here from that description and compared with the
 * reference's literals in tests/test_oracle_vs_ref.py. */
static double q2p[127];
static int q2p_ready = 0;
static void q2p_init(void)
{
	static const int up[8] = { 34, 39, 58, 62, 67, 71, 72, 82 };
	int q, j;
	if (q2p_ready) return;
	for (q = 0; q < 127; ++q) {
		char buf[64];
		long long units;
		snprintf(buf, sizeof(buf), "%.15f", pow(10.0, -q / 10.0));      /* "d.ddddddddddddddd" */
		units = (long long)(buf[0] - '0') * 1000000000000000LL + strtoll(buf + 2, 0, 10);
		for (j = 0; j < 8; ++j) if (up[j] == q) ++units;
		snprintf(buf, sizeof(buf), "%lld.%015lld", units / 1000000000000000LL, units % 1000000000000000LL);
		q2p[q] = strtod(buf, 0);
	}
	q2p_ready = 1;
}
static double mean_q(const char *qual, int length)
{
	int i; double sum = 0.0;
	q2p_init();
	for (i = 0; i < length; ++i) sum += q2p[(int)qual[i] - 33];
	return -10 * log10(sum / length);
}

/* ------------------------------------------------------------------------------------------------
 * One query against one part: lq_map_frag_mod (lqmap.c:207-326)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
	int n_regs;
	reg_t *regs;
	lqo_mm128 *a;          /* chained anchors */
	size_t n_a;
	subc_v cv;
} qmap_out;

static int cmp_anchor_group(const void *a_, const void *b_)    /* stable (x, emission index) order, used by sort_mode=1 */
{
	const lqo_mm128 *a = *(const lqo_mm128* const*)a_, *b = *(const lqo_mm128* const*)b_;
	if (a->x != b->x) return a->x < b->x ? -1 : 1;
	return a < b ? -1 : a > b;
}

static int cmp_anchor_rev(const void *a_, const void *b_)      /* x ascending, ties in reverse emission order (sort_mode=2) */
{
	const lqo_mm128 *a = *(const lqo_mm128* const*)a_, *b = *(const lqo_mm128* const*)b_;
	if (a->x != b->x) return a->x < b->x ? -1 : 1;
	return a > b ? -1 : a < b;
}
static int cmp_anchor_hash(const void *a_, const void *b_)     /* x ascending, ties by a hash of y (sort_mode=3) */
{
	const lqo_mm128 *a = *(const lqo_mm128* const*)a_, *b = *(const lqo_mm128* const*)b_;
	uint64_t ha, hb;
	if (a->x != b->x) return a->x < b->x ? -1 : 1;
	ha = mix64_full(a->y ^ 0x1234567ULL); hb = mix64_full(b->y ^ 0x1234567ULL);
	if (ha != hb) return ha < hb ? -1 : 1;
	return a < b ? -1 : a > b;
}

static void map_query(const lqo_params *P, const part_t *pt, int32_t mid_occ, const fx_rec *q, qstate_t *qst, qmap_out *dbg)
{
	lqo_mm128 *mv = 0, *a;
	size_t n_mv = 0, cap = 0, i;
	int64_t n_a = 0;
	uint64_t *mini_pos, sum_qspan = 0;
	int n_mini_pos = 0, qlen = q->l_seq, n_regs0 = 0;
	uint32_t hash;
	u64_v u = {0,0,0};
	mm_v b = {0,0,0};
	reg_t *regs;
	subc_v cv = {0,0,0};
	float avg_qspan;
	if (qlen == 0) return;                              /* lqmap.c:233 */
	hash = x31_hash(q->name);                           /* lqmap.c:235-237 */
	hash ^= wang_hash((uint32_t)qlen) + wang_hash((uint32_t)P->seed);
	hash = wang_hash(hash);
	lqo_sketch(q->seq, qlen, P->w, P->k, 0, P->hpc, &mv, &n_mv, &cap);   /* collect_minimizers (lqmap.c:126-138) */
	/* collect_seed_hits (lqmap.c:140-205) */
	mini_pos = (uint64_t*)malloc(8 * (n_mv ? n_mv : 1));
	for (i = 0; i < n_mv; ++i) { int t; part_get(pt, mv[i].x >> 8, &t); if (t < mid_occ) n_a += t; }
	a = (lqo_mm128*)malloc(sizeof(lqo_mm128) * (n_a ? n_a : 1));
	for (i = 0, n_a = 0; i < n_mv; ++i) {
		int t, k, q_span = mv[i].x & 0xff, is_tandem = 0;
		uint32_t qpos = (uint32_t)mv[i].y;
		const uint64_t *r = part_get(pt, mv[i].x >> 8, &t);
		if (t >= mid_occ) continue;
		mini_pos[n_mini_pos++] = (uint64_t)q_span << 32 | qpos >> 1;
		if (i > 0 && mv[i].x >> 8 == mv[i-1].x >> 8) is_tandem = 1;
		if (i + 1 < n_mv && mv[i].x >> 8 == mv[i+1].x >> 8) is_tandem = 1;
		for (k = 0; k < t; ++k) {
			int32_t rpos = (uint32_t)r[k] >> 1;
			lqo_mm128 *p;
			if (P->no_self || P->ava) {
				int cmp = strcmp(q->name, pt->name[r[k] >> 32]);
				if (P->no_self && cmp == 0 && (uint32_t)rpos == (qpos >> 1)) continue;
				if (P->ava && cmp > 0) continue;
			}
			p = &a[n_a++];
			if ((r[k] & 1) == (qpos & 1)) {
				p->x = (r[k] & 0xffffffff00000000ULL) | (uint32_t)rpos;
				p->y = (uint64_t)q_span << 32 | qpos >> 1;
			} else {
				p->x = 1ULL << 63 | (r[k] & 0xffffffff00000000ULL) | (uint32_t)rpos;
				p->y = (uint64_t)q_span << 32 | (uint32_t)(qlen - ((qpos >> 1) + 1 - q_span) - 1);
			}
			if (is_tandem) p->y |= SEED_TANDEM;
		}
	}
	/* sort anchors by x (lqmap.c:238) */
	if (P->sort_mode == 0) lqo_sort_128x(a, n_a);
	else {
		const lqo_mm128 **ptr = (const lqo_mm128**)malloc(sizeof(void*) * (n_a ? n_a : 1));
		lqo_mm128 *a2 = (lqo_mm128*)malloc(sizeof(lqo_mm128) * (n_a ? n_a : 1));
		for (i = 0; i < (size_t)n_a; ++i) ptr[i] = &a[i];
		qsort(ptr, n_a, sizeof(void*), P->sort_mode == 1 ? cmp_anchor_group : P->sort_mode == 2 ? cmp_anchor_rev : cmp_anchor_hash);
		for (i = 0; i < (size_t)n_a; ++i) a2[i] = *ptr[i];
		free(ptr);
		if (P->sort_mode >= 2) {
			/* The engine's scheme: any correct sort (here: an adversarial one), and klib's order only for the (strand, rid) runs
			 * where the order of equal-x anchors can be observed.  Every other run is chained in both orders and compared. */
			int64_t lo = 0, hi;
			int q_any_tie = 0, q_sens = 0;
			const int span_max = P->hpc ? 255 : P->k;
			const int64_t n_min = P->min_cnt > (P->min_chain_score + span_max - 1) / span_max ? P->min_cnt : (P->min_chain_score + span_max - 1) / span_max;
			float avg = 0.0f;
			uint64_t ss = 0;
			lqo_sort_128x(a, n_a);                          /* a: klib's order, a2: the adversarial order */
			for (i = 0; i < (size_t)n_a; ++i) ss += a[i].y >> 32 & 0xff;
			avg = n_a ? (float)ss / n_a : 0.0f;
			g_tie_stats[0] += 1; g_tie_stats[1] += n_a;
			while (lo < n_a) {
				int has_tie = 0, sens = 0;
				int64_t z;
				for (hi = lo + 1; hi < n_a && (a2[hi].x >> 32) == (a2[lo].x >> 32); ++hi) {}
				for (z = lo + 1; z < hi; ++z) if (a2[z].x == a2[z-1].x) has_tie = 1;
				if (has_tie) q_any_tie = 1;
				if (hi - lo >= n_min) {
					g_tie_stats[2] += 1; g_tie_stats[3] += hi - lo;
					if (has_tie) {
						u64_v u1 = {0,0,0}, u2 = {0,0,0}; mm_v b1 = {0,0,0}, b2 = {0,0,0};
						int peak = 0, band = 0;
						g_tie_stats[4] += 1;
						chain_dp_range_s(P, avg, a2 + lo, hi - lo, &u2, &b2, &peak, &band);
						if (getenv("LQO_TIE_STATIC")) sens = run_ties_close(P, a2 + lo, hi - lo);
						else if (getenv("LQO_TIE_NONE")) sens = 0;
						else sens = band;
						if (peak && !getenv("LQO_TIE_NONE")) sens = 1;
						if (sens && !getenv("LQO_TIE_NOBOUND") && run_score_bound(P, a2 + lo, hi - lo) < P->min_chain_score) sens = 0;   /* nothing to chain in any order */
						if (peak) g_tie_stats[11] += 1;
						if (sens) { g_tie_stats[5] += 1; g_tie_stats[6] += hi - lo; q_sens = 1; memcpy(a2 + lo, a + lo, (hi - lo) * sizeof(lqo_mm128)); }
						else {
							chain_dp_range(P, avg, a + lo, hi - lo, &u1, &b1);
							if (u1.n != u2.n || b1.n != b2.n || (u1.n && memcmp(u1.a, u2.a, u1.n * 8)) || (b1.n && memcmp(b1.a, b2.a, b1.n * sizeof(lqo_mm128)))) {
								g_tie_stats[7] += 1;
								if (getenv("LQO_TIE_VERBOSE")) fprintf(stderr, "[tie] query %s run at %lld len %lld: chains differ between klib's order and another although no tie is observable by the rule\n", q->name, (long long)lo, (long long)(hi - lo));
							}
						}
						free(u1.a); free(u2.a); free(b1.a); free(b2.a);
					}
				}
				lo = hi;
			}
			g_tie_stats[8] += q_any_tie; g_tie_stats[9] += q_sens;
			if (q_sens) g_tie_stats[10] += n_a;
		}
		free(a); a = a2;
	}
	/* mm_chain_dp (lqmap.c:252) */
	for (i = 0; i < (size_t)n_a; ++i) sum_qspan += a[i].y >> 32 & 0xff;
	avg_qspan = n_a ? (float)sum_qspan / n_a : 0.0f;    /* chain.c:37-38; n_a==0 gives NaN in the reference, never used */
	if (P->chain_mode == 0) chain_dp_range(P, avg_qspan, a, n_a, &u, &b);
	else {                                              /* per-(strand,rid) groups, groups below min_cnt dropped */
		int64_t lo = 0, hi;
		while (lo < n_a) {
			for (hi = lo + 1; hi < n_a && (a[hi].x >> 32) == (a[lo].x >> 32); ++hi) {}
			if (hi - lo >= P->min_cnt) chain_dp_range(P, avg_qspan, a + lo, hi - lo, &u, &b);
			lo = hi;
		}
	}
	chains_order_by_x(&u, &b);
	n_regs0 = (int)u.n;
	regs = gen_regs(hash, qlen, n_regs0, u.a, b.a);    /* lqmap.c:279 */
	cnt_match(P, pt, qlen, n_regs0, regs, b.a, n_mini_pos, mini_pos, qst, &cv);   /* lqmap.c:282 */
	if (dbg) {
		dbg->n_regs = n_regs0; dbg->regs = regs; dbg->a = b.a; dbg->n_a = b.n;
		dbg->cv.a = (subc_t*)malloc(sizeof(subc_t) * (cv.n ? cv.n : 1)); dbg->cv.n = dbg->cv.m = cv.n;
		memcpy(dbg->cv.a, cv.a, sizeof(subc_t) * cv.n);
	}
	filter_redundant(&qst->ovlp, &cv, (uint32_t)P->min_coverage);   /* lqmap.c:287 */
	free(cv.a); free(mv); free(a); free(mini_pos); free(u.a);
	if (!dbg) { free(regs); free(b.a); }
}

/* ------------------------------------------------------------------------------------------------
 * Drivers
 * ---------------------------------------------------------------------------------------------- */
static int read_all(fx_t *f, rec_v *out)
{
	while (read_minibatch(f, INT64_MAX, out, 0) > 0) {}
	return (int)out->n;
}

int lqo_dump_sketch(const lqo_params *p, const char *fn, FILE *out)
{
	fx_t f; rec_v recs = {0,0,0};
	size_t i, j;
	if (fx_open(&f, fn) < 0) return -1;
	read_all(&f, &recs);
	for (i = 0; i < recs.n; ++i) {
		lqo_mm128 *mv = 0; size_t n = 0, cap = 0;
		if (recs.a[i].l_seq > 0) lqo_sketch(recs.a[i].seq, recs.a[i].l_seq, p->w, p->k, (uint32_t)i, p->hpc, &mv, &n, &cap);
		fprintf(out, "R\t%s\t%d\t%zu\n", recs.a[i].name, recs.a[i].l_seq, n);
		for (j = 0; j < n; ++j) fprintf(out, "M\t%016" PRIx64 "\t%016" PRIx64 "\n", mv[j].x, mv[j].y);
		free(mv);
	}
	free_recs(&recs); fx_close(&f);
	return 0;
}

int lqo_dump_index(const lqo_params *p, const char *fn, FILE *out)
{
	fx_t f; part_t pt; int part = 0;
	if (fx_open(&f, fn) < 0) return -1;
	while (part_read(&f, p, &pt)) {
		part_build(p, &pt);
		fprintf(out, "P\t%d\t%u\t%" PRIu64 "\t%d\n", part, pt.n_seq, pt.tot_len, part_mid_occ(&pt, p->mid_occ_frac));
		++part;
		part_free(&pt);
	}
	fx_close(&f);
	return 0;
}

static int cmp_reg(const void *a_, const void *b_)
{
	const reg_t *a = (const reg_t*)a_, *b = (const reg_t*)b_;
	if (a->rid != b->rid) return a->rid < b->rid ? -1 : 1;
	if (a->rev != b->rev) return a->rev < b->rev ? -1 : 1;
	if (a->rs != b->rs) return a->rs < b->rs ? -1 : 1;
	if (a->qs != b->qs) return a->qs < b->qs ? -1 : 1;
	if (a->re != b->re) return a->re < b->re ? -1 : 1;
	if (a->qe != b->qe) return a->qe < b->qe ? -1 : 1;
	if (a->score0 != b->score0) return a->score0 < b->score0 ? -1 : 1;
	return a->cnt < b->cnt ? -1 : a->cnt > b->cnt;
}

static void qstate_init(qstate_t *s, const lqo_params *p, const fx_rec *q, uint32_t idx)
{
	lqo_mm128 *mv = 0; size_t n = 0, cap = 0;
	memset(s, 0, sizeof(*s));
	if (q->l_seq > 0) lqo_sketch(q->seq, q->l_seq, p->w, p->k, idx, p->hpc, &mv, &n, &cap);  /* minimap2-coverage.c:418-426 */
	free(mv);
	s->n_cnt = (uint32_t)n;
	s->cnt = (uint16_t*)calloc(n ? n : 1, 2);
}

int lqo_dump_chains(const lqo_params *p, const char *target_fn, const char *query_fn, FILE *out)
{
	fx_t ft, fq; part_t pt; rec_v qs = {0,0,0};
	size_t i, z;
	int32_t mid_occ;
	if (fx_open(&ft, target_fn) < 0) return -1;
	if (fx_open(&fq, query_fn) < 0) { fx_close(&ft); return -1; }
	if (!part_read(&ft, p, &pt)) return -1;
	part_build(p, &pt);
	mid_occ = part_mid_occ(&pt, p->mid_occ_frac);
	fprintf(out, "I\t%u\t%d\n", pt.n_seq, mid_occ);
	read_all(&fq, &qs);
	for (i = 0; i < qs.n; ++i) {
		qstate_t st; qmap_out d;
		int j;
		memset(&d, 0, sizeof(d));
		if (qs.a[i].l_seq == 0) { fprintf(out, "Q\t%zu\t%s\t0\t0\t0\t0\n", i, qs.a[i].name); continue; }
		qstate_init(&st, p, &qs.a[i], (uint32_t)i);
		map_query(p, &pt, mid_occ, &qs.a[i], &st, &d);
		fprintf(out, "Q\t%zu\t%s\t%d\t%d\t%" PRIu64 "\t%" PRIu64 "\n", i, qs.a[i].name, qs.a[i].l_seq, d.n_regs, st.lambda, st.lambda2);
		if (d.n_regs) qsort(d.regs, d.n_regs, sizeof(reg_t), cmp_reg);
		for (j = 0; j < d.n_regs; ++j)
			fprintf(out, "C\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", d.regs[j].rid, d.regs[j].rev, d.regs[j].score0, d.regs[j].cnt,
			        d.regs[j].qs, d.regs[j].qe, d.regs[j].rs, d.regs[j].re);
		for (z = 0; z < st.ovlp.n; ++z) fprintf(out, "V\t%u\t%u\n", st.ovlp.a[z].start, st.ovlp.a[z].end);
		for (z = 0; z < st.n_cnt; ++z) if (st.cnt[z]) fprintf(out, "N\t%zu\t%u\n", z, st.cnt[z]);
		free(d.regs); free(d.a); free(d.cv.a); free(st.cnt); free(st.ovlp.a);
	}
	free_recs(&qs); part_free(&pt); fx_close(&ft); fx_close(&fq);
	return 0;
}

/* pass-2 formatter (minimap2-coverage.c:545-617) */
static void format_row(const lqo_params *p, const fx_rec *q, qstate_t *s, FILE *out)
{
	subc_v regs = {0,0,0}, mregs = {0,0,0};
	int32_t n_match = 0;
	double div;
	uint32_t sum = 0, j;
	for (j = 0; j < s->n_cnt; ++j) sum += s->cnt[j];
	if (s->n_cnt) sum /= s->n_cnt;                     /* reference divides by zero here (UB, elided by gcc -O2) */
	for (j = 0; j < s->n_cnt; ++j) if (s->cnt[j] > sum) n_match++;
	div = n_match > 0 ? logf((float)s->n_cnt / n_match) / s->avg_k : 1.0;
	reliable_region(&s->ovlp, (uint32_t)p->min_coverage, &regs, &mregs);
	if (regs.n > 0) {
		uint32_t tot = 0;
		size_t k;
		fprintf(out, "%s\t%d\t%" PRIu64 "\t", q->name, q->l_seq, s->lambda);
		for (k = 0; k < regs.n; ++k) {
			fprintf(out, "%s%d-%d", k ? "," : "", regs.a[k].start, regs.a[k].end);
			tot += regs.a[k].end - regs.a[k].start;
		}
		fputc('\t', out);
		if (mregs.n > 0) for (k = 0; k < mregs.n; ++k) fprintf(out, "%s%d-%d", k ? "," : "", mregs.a[k].start, mregs.a[k].end);
		else fputc('0', out);
		if (p->filter_flag)
			fprintf(out, "\t%.3f\t%.3f\t%.3f\t0.0\n", (double)tot / q->l_seq, mean_q(q->qual, q->l_qual), div);
		else
			fprintf(out, "\t%.3f\t%.3f\t%.3f\t%.3f\n", (double)s->lambda / tot, mean_q(q->qual, q->l_qual), div, (double)s->lambda2 / tot);
	} else
		fprintf(out, "%s\t%d\t%" PRIu64 "\t0\t0\t0.0\t%.3f\t%.3f\t0.0\n", q->name, q->l_seq, s->lambda, mean_q(q->qual, q->l_qual), div);
	free(regs.a); free(mregs.a);
}

int lqo_run_files(const lqo_params *p, const char *target_fn, const char *query_fn, FILE *out, FILE *log)
{
	fx_t ft, fq; part_t pt; rec_v qs = {0,0,0};
	qstate_t *st;
	size_t i;
	int32_t mid_occ = 0;
	int part = 0;
	double t0 = now_s(), t1;
	memset(g_timing, 0, sizeof(g_timing));
	if (fx_open(&ft, target_fn) < 0) { if (log) fprintf(log, "ERROR: failed to open file '%s'\n", target_fn); return -1; }
	if (fx_open(&fq, query_fn) < 0) { fx_close(&ft); if (log) fprintf(log, "ERROR: failed to open file '%s'\n", query_fn); return -1; }
	/* query records: the reference maps them in 500-Mbase mini-batches (lqmap.c:762) but slots alias
	 * across mini-batches; parity domain = one mini-batch.  We read them all. */
	read_all(&fq, &qs);
	st = (qstate_t*)calloc(qs.n ? qs.n : 1, sizeof(qstate_t));
	g_timing[0] += now_s() - t0;
	t0 = now_s();
	for (i = 0; i < qs.n; ++i) qstate_init(&st[i], p, &qs.a[i], (uint32_t)i);
	g_timing[1] += now_s() - t0;
	for (;;) {
		t0 = now_s();
		if (!part_read(&ft, p, &pt)) break;
		t1 = now_s(); g_timing[0] += t1 - t0;
		part_build(p, &pt);
		if (mid_occ <= 0) mid_occ = part_mid_occ(&pt, p->mid_occ_frac);   /* map.c:50: first part only */
		t0 = now_s(); g_timing[1] += t0 - t1;
		if (log) fprintf(log, "[oracle] part %d: %u seqs, %" PRIu64 " bases, %zu minimizers, mid_occ=%d\n", part, pt.n_seq, pt.tot_len, pt.n_mini, mid_occ);
		for (i = 0; i < qs.n; ++i) map_query(p, &pt, mid_occ, &qs.a[i], &st[i], 0);
		g_timing[2] += now_s() - t0;
		part_free(&pt);
		++part;
	}
	t0 = now_s();
	for (i = 0; i < qs.n; ++i) {
		format_row(p, &qs.a[i], &st[i], out);
		free(st[i].cnt); free(st[i].ovlp.a);
	}
	g_timing[3] += now_s() - t0;
	free(st); free_recs(&qs); fx_close(&ft); fx_close(&fq);
	return 0;
}

int lqo_run_paths(const lqo_params *p, const char *target_fn, const char *query_fn, const char *out_fn)
{
	FILE *o = fopen(out_fn, "w");
	int r;
	if (!o) return -1;
	r = lqo_run_files(p, target_fn, query_fn, o, 0);
	fclose(o);
	return r;
}

int lqo_dump_paths(const lqo_params *p, const char *what, const char *target_fn, const char *query_fn, const char *out_fn)
{
	FILE *o = fopen(out_fn, "w");
	int r = -2;
	if (!o) return -1;
	if (strcmp(what, "sketch") == 0) r = lqo_dump_sketch(p, target_fn, o);
	else if (strcmp(what, "index") == 0) r = lqo_dump_index(p, target_fn, o);
	else if (strcmp(what, "chains") == 0) r = lqo_dump_chains(p, target_fn, query_fn, o);
	fclose(o);
	return r;
}

/* ------------------------------------------------------------------------------------------------
 * the low-complexity table of the reference's second binary, `sdust` (sdust.c:136-217):
 * symmetric DUST (window W of 3-mers, threshold T) over every read, one row per read:
 * name, masked bases, length, masked/length, meanQ, #qualities > Q7 (lqutils.c:51-69).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int start, finish, r, l; } dust_pi;                 /* a perfect interval of the current window */
typedef struct {
	int q[256], qn, qh;        /* the last <= W-2 words, oldest first (ring) */
	int cw[64], cv[64];        /* word counts over the window / over its last L words */
	int rw, rv, L;
	dust_pi *P; size_t np, mp; /* descending start, then ascending finish */
	int have_last, ls, lf;     /* the last masked interval (may still grow) */
	int64_t masked;            /* total length of the closed masked intervals */
} dust_t;

static inline int dust_at(const dust_t *d, int i) { return d->q[(d->qh + i) & 255]; }

/* sdust.c:93-108 */
static void dust_flush(dust_t *d, int start)
{
	long i;
	const dust_pi *p;
	if (d->np == 0 || d->P[d->np - 1].start >= start) return;
	p = &d->P[d->np - 1];
	if (d->have_last && p->start <= d->lf) { if (p->finish > d->lf) d->lf = p->finish; }   /* overlaps / touches the last one */
	else {
		if (d->have_last) d->masked += d->lf - d->ls;
		d->have_last = 1; d->ls = p->start; d->lf = p->finish;
	}
	for (i = (long)d->np - 1; i >= 0 && d->P[i].start < start; --i) {}
	d->np = (size_t)(i + 1);
}

/* sdust.c:70-91 */
static void dust_shift(dust_t *d, int t, int T, int W)
{
	int s;
	if (d->qn >= W - 3 + 1) {
		s = d->q[d->qh]; d->qh = (d->qh + 1) & 255; --d->qn;
		d->rw -= --d->cw[s];
		if (d->L > d->qn) { --d->L; d->rv -= --d->cv[s]; }
	}
	d->q[(d->qh + d->qn) & 255] = t; ++d->qn;
	++d->L;
	d->rw += d->cw[t]++;
	d->rv += d->cv[t]++;
	if (d->cv[t] * 10 > T << 1) {
		do {
			s = dust_at(d, d->qn - d->L);
			d->rv -= --d->cv[s];
			--d->L;
		} while (s != t);
	}
}

/* sdust.c:110-134 */
static void dust_find(dust_t *d, int T, int start)
{
	int c[64], r = d->rv, i, max_r = 0, max_l = 0;
	memcpy(c, d->cv, sizeof(c));
	for (i = d->qn - d->L - 1; i >= 0; --i) {
		const int t = dust_at(d, i);
		int new_r, new_l;
		size_t j;
		r += c[t]++;
		new_r = r; new_l = d->qn - i - 1;
		if (new_r * 10 > T * new_l) {
			for (j = 0; j < d->np && d->P[j].start >= i + start; ++j) {
				const dust_pi *p = &d->P[j];
				if (max_r == 0 || p->r * max_l > max_r * p->l) { max_r = p->r; max_l = p->l; }
			}
			if (max_r == 0 || new_r * max_l >= max_r * new_l) {
				max_r = new_r; max_l = new_l;
				if (d->np == d->mp) { d->mp = d->mp ? d->mp * 2 : 16; d->P = (dust_pi*)realloc(d->P, d->mp * sizeof(dust_pi)); }
				memmove(&d->P[j + 1], &d->P[j], (d->np - j) * sizeof(dust_pi));
				++d->np;
				d->P[j].start = i + start; d->P[j].finish = d->qn + 2 + start; d->P[j].r = new_r; d->P[j].l = new_l;
			}
		}
	}
}

/* sdust_core (sdust.c:136-171): the number of masked bases of one read.  Note that an N ends the run of words (l, t)
 * and closes the pending intervals but leaves the window and its counts as they are. */
uint32_t lqo_sdust_masked(const char *seq, int l_seq, int T, int W)
{
	static const signed char nt4[128] = { ['A'] = 1, ['C'] = 2, ['G'] = 3, ['T'] = 4, ['a'] = 1, ['c'] = 2, ['g'] = 3, ['t'] = 4 };
	dust_t d;
	int i, l = 0, start;
	unsigned t = 0;
	uint32_t total;
	memset(&d, 0, sizeof(d));
	for (i = 0; i <= l_seq; ++i) {
		const int ch = i < l_seq ? (unsigned char)seq[i] : 0;
		const int b = ch < 128 ? nt4[ch] - 1 : -1;
		if (b >= 0) {
			++l; t = (t << 2 | (unsigned)b) & 63;
			if (l >= 3) {
				start = (l - W > 0 ? l - W : 0) + (i + 1 - l);
				dust_flush(&d, start);
				dust_shift(&d, (int)t, T, W);
				if (d.rw * 10 > d.L * T) dust_find(&d, T, start);
			}
		} else {
			start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);
			while (d.np) dust_flush(&d, start++);
			l = 0; t = 0;
		}
	}
	if (d.have_last) d.masked += d.lf - d.ls;
	total = (uint32_t)d.masked;
	free(d.P);
	return total;
}

/* sdust's main (sdust.c:181-222) */
int lqo_sdust_file(const char *fn, int W, int T, FILE *out)
{
	fx_t f;
	cstr name = {0,0,0}, seq = {0,0,0}, qual = {0,0,0};
	int l;
	if (W > 254 || W < 3) return -1;
	if (fx_open(&f, fn) != 0) return -1;
	while ((l = fx_next(&f, &name, &seq, &qual)) >= 0) {
		uint32_t masked, qv = 0;
		size_t i;
		vpush(name, 0);
		masked = lqo_sdust_masked(seq.a, (int)seq.n, T, W);
		for (i = 0; i < qual.n; ++i) if ((int)qual.a[i] > 7 + 33) ++qv;            /* getQV (lqutils.c:61-69) */
		fprintf(out, "%s\t%d\t%d\t%.3f\t%.3f\t%d\n", name.a, (int)masked, (int)seq.n, (double)masked / seq.n,
		        mean_q(qual.a, (int)qual.n), (int)qv);
	}
	free(name.a); free(seq.a); free(qual.a);
	fx_close(&f);
	return 0;
}

int lqo_sdust_path(const char *fn, int W, int T, const char *out_fn)
{
	FILE *o = fopen(out_fn, "w");
	int rc;
	if (!o) return -1;
	rc = lqo_sdust_file(fn, W, T, o);
	fclose(o);
	return rc;
}
