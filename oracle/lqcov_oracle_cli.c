/* oracle/lqcov_oracle_cli.c -- TEST INFRASTRUCTURE ONLY.
 * Command-line front end of the CPU restatement; accepts the option letters of the reference's
 * argp table (minimap2-coverage.c:166-195) plus dump sub-commands that print the same text as
 * oracle/ref_harness.c, so the two can be diffed.
 *
 *   lqcov_oracle table  [opts] <targets> <queries>     (stdout = the 9-column table)
 *   lqcov_oracle sketch [opts] <reads>
 *   lqcov_oracle index  [opts] <targets>
 *   lqcov_oracle chains [opts] <targets> <queries>
 *   opts: -k -w -H -I -g -n -m -p -q -s -a -l -c -r -Y -X -f  --stable-sort --grouped --ties-reversed --ties-hashed
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lqcov_oracle.h"

static int64_t parse_num(const char *str)
{
	double x; char *p;
	x = strtod(str, &p);
	if (*p == 'G' || *p == 'g') x *= 1e9;
	else if (*p == 'M' || *p == 'm') x *= 1e6;
	else if (*p == 'K' || *p == 'k') x *= 1e3;
	return (int64_t)(x + .499);
}

int main(int argc, char **argv)
{
	lqo_params p;
	const char *pos[4];
	int npos = 0, i, p_set = 0, q_set = 0;
	if (argc < 3) { fprintf(stderr, "usage: lqcov_oracle table|sketch|index|chains [opts] files... | sdust [-w W] [-t T] <reads>\n"); return 2; }
	if (!strcmp(argv[1], "sdust")) {                        /* == `sdust [-w 64] [-t 20] <in.fa>` (sdust.c:181-196) */
		int W = 64, T = 20, i2;
		const char *fn = 0;
		for (i2 = 2; i2 < argc; ++i2) {
			const char *a2 = argv[i2];                       /* getopt "w:t:": -w 16 and -w16 */
			if (a2[0] == '-' && (a2[1] == 'w' || a2[1] == 't')) {
				const char *v = a2[2] ? a2 + 2 : (i2 + 1 < argc ? argv[++i2] : "0");
				if (a2[1] == 'w') W = atoi(v); else T = atoi(v);
			} else if (!fn) fn = a2;
		}
		if (!fn) return 2;
		return lqo_sdust_file(fn, W, T, stdout) ? 1 : 0;
	}
	lqo_params_default(&p);
	p.min_ovlp = 1000;
	for (i = 2; i < argc; ++i) {
		const char *a = argv[i];
		if (a[0] == '-' && a[1] == '-') {
			if (!strcmp(a, "--stable-sort")) p.sort_mode = 1;
			else if (!strcmp(a, "--grouped")) p.chain_mode = 1;
			else if (!strcmp(a, "--ties-reversed")) p.sort_mode = 2, p.chain_mode = 1;
			else if (!strcmp(a, "--ties-hashed")) p.sort_mode = 3, p.chain_mode = 1;
			else if (!strcmp(a, "--filter")) p.filter_flag = 1;
			else { fprintf(stderr, "unknown option %s\n", a); return 2; }
		} else if (a[0] == '-' && a[1]) {
			int j;
			for (j = 1; a[j]; ++j) {
				char c = a[j];
				const char *v = 0;
				if (strchr("HYXf", c)) {
					if (c == 'H') p.hpc = 1;
					else if (c == 'Y') { p.no_self = 1; p.ava = 0; }
					else if (c == 'X') { p.no_self = 1; p.ava = 1; }
					else p.filter_flag = 1;
					continue;
				}
				v = a[j+1] ? &a[j+1] : (i + 1 < argc ? argv[++i] : 0);
				if (!v) { fprintf(stderr, "option -%c needs a value\n", c); return 2; }
				switch (c) {
				case 'k': p.k = atoi(v); break;
				case 'w': p.w = atoi(v); break;
				case 'I': p.batch_size = (uint64_t)parse_num(v); break;
				case 'g': p.max_gap = atoi(v); break;
				case 'n': p.min_cnt = atoi(v); break;
				case 'm': p.min_chain_score = atoi(v); break;
				case 'p': p.min_score_med = atoi(v); p_set = 1; break;
				case 'q': p.min_score_good = atoi(v); q_set = 1; break;
				case 's': p.max_chain_skip = atoi(v); break;
				case 'a': p.max_overhang = atoi(v); break;
				case 'l': p.min_ovlp = atoi(v); break;
				case 'c': p.min_coverage = atoi(v); break;
				case 'r': p.min_ratio = atof(v); break;
				case 't': break;
				default: fprintf(stderr, "unknown option -%c\n", c); return 2;
				}
				break;
			}
		} else if (npos < 4) pos[npos++] = a;
	}
	if (!p_set || p.min_score_med == 0) p.min_score_med = p.min_chain_score;   /* minimap2-coverage.c:324-332 */
	if (!q_set || p.min_score_good == 0) p.min_score_good = p.min_chain_score;
	if (!strcmp(argv[1], "table") && npos == 2) {
		int rc = lqo_run_files(&p, pos[0], pos[1], stdout, stderr) ? 1 : 0;
		if (p.sort_mode >= 2) {
			uint64_t s[12]; lqo_tie_stats(s);
			fprintf(stderr, "[ties] queries %llu anchors %llu | runs>=n_min %llu (%llu anchors) with ties %llu observable %llu (%llu anchors) MISMATCH %llu | queries with ties %llu with observable runs %llu (%llu anchors)\n",
			        (unsigned long long)s[0], (unsigned long long)s[1], (unsigned long long)s[2], (unsigned long long)s[3], (unsigned long long)s[4], (unsigned long long)s[5],
			        (unsigned long long)s[6], (unsigned long long)s[7], (unsigned long long)s[8], (unsigned long long)s[9], (unsigned long long)s[10]);
			if (getenv("LQO_TIE_DEBUG")) { extern unsigned long long lqo_tie_reason[8]; fprintf(stderr, "[ties] events: top score twice %llu, a member counts as a skip %llu, a skip pending %llu, break inside a group %llu, equal peaks %llu, groups without a raiser %llu\n", lqo_tie_reason[0], lqo_tie_reason[1], lqo_tie_reason[2], lqo_tie_reason[3], lqo_tie_reason[4], lqo_tie_reason[5]); }
			if (s[7]) rc = 3;
		}
		return rc;
	}
	if (!strcmp(argv[1], "sketch") && npos == 1) return lqo_dump_sketch(&p, pos[0], stdout) ? 1 : 0;
	if (!strcmp(argv[1], "index") && npos == 1) return lqo_dump_index(&p, pos[0], stdout) ? 1 : 0;
	if (!strcmp(argv[1], "chains") && npos == 2) return lqo_dump_chains(&p, pos[0], pos[1], stdout) ? 1 : 0;
	fprintf(stderr, "bad arguments\n");
	return 2;
}
