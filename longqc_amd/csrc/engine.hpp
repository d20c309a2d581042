// longqc_amd/csrc/engine.hpp -- host-side orchestration of the MI355X coverage engine.
//
// Mirrors the structure of the reference's main (minimap2-coverage.c:206-734): query set once
// (pass 1, :406-444), then per index part: build (index.c:311-330) + mid_occ on the first part
// (map.c:46-54) + map every query (lqmap.c:852 -> :207-326), then pass 2 (:545-617).  All state
// that the reference keeps per query on the host (lambdas, lambdas2, avg_ks, m_cnts, ovlp_coords)
// lives in HBM here; the host sees it only in finish().
#pragma once
#include "lq_common.hpp"
#include "prim.hpp"
struct KeyMap;
struct PsData;
#include "../../include/lqcov.h"
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <exception>

struct StageAcc { double ms = 0; u64 launches = 0; u64 bytes = 0; };

// Environment switches, read once when the handle is made (test and A/B aids; the defaults are the measured choice).
struct Knobs {
	int lanes = 3;                        // LQCOV_LANES: concurrent mapping lanes (round 4, configs[2], ms per step: 2 lanes 908, 3: 888-922, 4: 964, 5: 995; round 3, every hit sorted: 1 lane 1882, 3: 1582, 5: 1510, 8: 1610)
	u64 anchor_budget = 0;                // LQCOV_ANCHOR_BUDGET: anchors per query batch (0 = from free HBM)
	bool query_order_striped = false;     // LQCOV_QUERY_ORDER=striped: longest first, then dealt to the lanes' stripes (round 6: measured, no gain)
	bool query_order_file = false;        // LQCOV_QUERY_ORDER=file: keep the caller's query order inside
	bool all_klib = false;                // LQCOV_SORT=klib: every query through klib's passes, no bucket leaves them early
	u32 ps_shift = 0;                     // LQCOV_PS_SHIFT: shrinks the size classes of the parallel sort (tests)
	bool reg_walker = true;               // LQCOV_WALK=solo: no register-lane walker
	bool ckpt = true, ckpt3 = true;       // LQCOV_CKPT=0: no checkpointed walks; LQCOV_CKPT3=0: the 65-160 k class is walked whole
	int lazy_batches = 1;                 // LQCOV_LAZY_BATCHES: batches per lane of a lazy plan
	bool plan_lazy = false;               // LQCOV_PLAN_LAZY=1: the first part's seed filter is left to the mapping lanes, each deciding its own batch of queries before it maps them (the next lane decides under this one's mapping).  Measured in round 6 at configs[2]: 425 ms per step against 399 -- beside a mapping lane and the next part's sketch the filter's kernels take twice as long (60 + 55 + 56 ms for the three batches, 85 ms for all of them alone) and the last lane starts later than it does after a whole plan; two / three batches per lane: 449 / 483 ms.  Round 5 had found the same with the filter inside map_part (586 vs 576 ms)
	bool prune = true;                    // LQCOV_PRUNE=0: the second pass sorts every bucket of its queries (rounds 4-5), not only those that hold a listed run
	u32 sketch_grid = 1u << 22;           // LQCOV_SKETCH_GRID: blocks of k_sketch_dp_mask (a block strides over the tiles)
	bool build_prio = true;               // LQCOV_BUILD_PRIO=0: the build side's streams without the higher queue priority
	u32 ck_unit = 65536, ck_unit_many = 8192;   // LQCOV_CK_UNIT / LQCOV_CK_UNIT_MANY: elements per checkpoint, passes of up to 16 / up to 256 buckets (configs[2], ms per step: 16384 / 4096: 502, 65536 / 4096: 502, 65536 / 8192: 491, 65536 / 16384: 491, 131072 / 8192: 494)
	u32 sort_tile = 0;                    // LQCOV_SORT_TILE: anchors per tile of the sort's streaming kernels (0 = LQ_SORT_TILE)
	u32 walk_shift = 0;                   // LQCOV_WALK_SHIFT: shrinks the walker size classes and the checkpoint spacing (tests)
	u32 walk_grid = 1u << 18;             // LQCOV_WALK_GRID: cap on resident walker waves
	u32 walk_cu_mask = 0x11111111u;       // LQCOV_WALK_CU_MASK (hex, repeated over the 256 CUs): the CUs the walkers' streams may use
	int chain_wave_min = 0, chain_cap = 128;   // LQCOV_CHAIN_WAVE_MIN (0 = LQ_CHAIN_WAVE_MIN), LQCOV_CHAIN_CAP
	bool sketch_fast = true;              // LQCOV_SKETCH_FAST=0: k_sketch_dp_mask although k_sketch_dp_fast applies (-k 12 with -w 5 or 10; tests, A/B)
	bool sketch_list = true;              // LQCOV_SKETCH_LIST=0: the state machine looks for the chunks the data-parallel kernel left in every wave of 64 consecutive chunks (rounds 3-5) instead of taking them from a list
	bool sketch_key = true;               // LQCOV_SKETCH_KEY=0: the index sort's keys by k_sort_keys from x instead of by k_sketch_emit_mask
	u32 emit_grid = 1u << 22;             // LQCOV_EMIT_GRID: blocks of k_sketch_emit_mask (a block strides over the groups of 32 chunks)
	bool upload_amb = false;              // LQCOV_UPLOAD_AMB=1: lqcov_run_files uploads the ambiguity words of a part although none of its reads holds an ambiguous base (rounds 2-5)
	bool sketch_wgen = false;             // LQCOV_SKETCH_WGEN=1: k_sketch_dp_mask with the window read at run time although it is 5 or 10 (tests, A/B)
	bool ps_key64 = false;                // LQCOV_PS_KEY64=1: the finishing kernels' 64-bit key shape although 32 bits would do (tests: parts with more than 2^40 (target, position) pairs are out of their reach)
	u32 run_grid = 2048;                  // LQCOV_RUN_GRID: blocks of k_run_list, each with a contiguous stretch of tiles (tests: 1 or 2, so that a block walks many)
	u32 run_stage = 2048;                 // LQCOV_RUN_STAGE: entries of the run list a block collects before it reserves their place (tests shrink it)
	bool no_level_skip = false;           // LQCOV_NO_LEVEL_SKIP: constant key bytes are walked, not stepped over
	bool debug_sort = false;              // LQCOV_DEBUG_SORT
	u32 sketch_kpt = 4;                   // LQCOV_SKETCH_KPT: chunks per thread of the sketch state machine
	u32 ps_passes = 2;                    // LQCOV_PS_PASSES: partition passes issued without looking (even; the tail looks at the counter and does the rest).  Two cover queries of up to ~500 M anchors against 65 536 targets; measured 4 vs 2 at configs[2]: 1.71-1.77 vs 1.69-1.71 s per step
	u32 tile_grid = 4096;                 // LQCOV_TILE_GRID: blocks of klib's tile kernels (histogram, scatter)
	u32 ps_grid = 512;                   // LQCOV_PS_GRID: blocks of the parallel sort's tile kernels (the finishing kernels: a quarter / twice that; twice as many blocks for them: no change, measured with the 64-register kernels)
	u64 upload_min_chunks = 1u << 16;     // LQCOV_UPLOAD_MIN_CHUNKS: only read sets of that many 128-base chunks go up in slices (tests lower it)
	u32 upload_slices = 4;                // LQCOV_UPLOAD_SLICES (1..8): packed reads go up in slices, the data-parallel sketch kernel takes a slice while the next one is on its way (1: one copy, then the sketch)
	bool sketch_machine_only = false;     // LQCOV_SKETCH=machine: the state machine decides every chunk (no data-parallel kernel)
	bool ties_klib = false;               // LQCOV_TIES=klib (or LQCOV_SORT=klib): klib's order of equal-x anchors everywhere, every seed hit written and sorted (rounds 1-3); default: only where it can be observed (map_batch)
	bool filter = true;                   // LQCOV_FILTER=0: the first pass writes every seed hit (no counting filter)
	bool plan_ahead = true;               // LQCOV_PLAN_AHEAD=0: a part's seed plan (probe, survivors) is made when the part is mapped, not right after its index
	int parse_threads = 0;                // LQCOV_PARSE_THREADS: threads that parse and pack a plain target file (0: one per core, at most 64; 1: the streaming reader as for gzip)
	u64 parse_piece = 32u << 20;          // LQCOV_PARSE_PIECE: bytes of the file a thread parses at a time (tests shrink it: many guessed record starts)
	u64 parse_side = 2ULL << 30;          // LQCOV_PARSE_SIDE: bytes of multi-line records the mapped reader may copy before run_files falls back to the streaming reader
	bool pipeline = true;                 // LQCOV_PIPELINE=0: run_files builds a part only after the one before is mapped
	int cnt_bits = 16;                    // LQCOV_TEST_CNT_BITS (2..16): width of the match counters.  A test hook: narrower counters bring the saturated regime (sat_replay.hpp) within reach of small inputs; the oracle has the same one (LQO_CNT_BITS)
	u32 seed_bucket = 7600;               // LQCOV_SEED_BUCKET: hits per (query, slice of targets) bucket aimed at; k_seed_decide holds a bucket of up to 8192 records in registers (tests shrink it: many slices on small inputs)
	u64 seed_chunk = 1ULL << 30;          // LQCOV_SEED_CHUNK: records (8 B) of the bucket buffer; the queries of a part are bucketed in chunks of that many hits
	u32 seed_segl = 256;                  // LQCOV_SEED_SEGL: minimizers per segment (one block of the count / scatter kernels), at most LQ_SD_SEGL (tests shrink it)
	u32 seed_pair_bits = 13;              // LQCOV_SEED_PAIR_BITS: pair counters of k_seed_decide in use (tests shrink it: pairs alias on small inputs)
	u32 seed_dcap = 8192, seed_bigcap = 65536;   // LQCOV_SEED_DCAP / LQCOV_SEED_BIGCAP: records of a bucket k_seed_decide takes from registers / in passes over stretches of its targets (tests shrink them)
	u64 seed_surv_max = 2ULL << 30;       // LQCOV_SEED_SURV_MAX: survivors (8 B each) a part's plan may hold; beyond that the part is mapped without the filter (tests shrink it)
	u32 seed_hwords = 12288;              // LQCOV_SEED_HWORDS: histogram space of a bucket in words of eight bins (tests shrink it: pairs that find no room are kept as they are)
	void read_env();
};

struct SatSink {                          // where the chain kernels record a replayed query's chains (CovState::rec)
	SatRec *rec; unsigned long long *n_rec; u64 rec_cap;
	u32 *at; unsigned long long *n_at; u64 at_cap;
};

struct ReadSetDev {                       // a read set 2-bit packed in HBM, chunk aligned
	u32 n = 0;
	u64 n_chunks = 0, n_bases = 0;
	std::vector<u64> h_coff{0};           // chunk offset of every read (+ total)
	std::vector<u32> h_len;
	std::vector<std::string> names;
	DBuf codes, amb;                      // 4 x u64 / 4 x u32 per chunk
	u64 cap_chunks = 0;
	DBuf d_coff, d_len;                   // device copies of h_coff / h_len
	DBuf mx, my, moff;                    // minimizers (x, y) in emission order + per-read offsets
	u64 n_mini = 0;
	bool sketched = false;
	u64 key_stamp = 0;                    // not 0: the handle's ix_key held this set's sort keys when ix_key_stamp had the same value
	u32 dp_n = 0; u64 dp_tiles = 0;       // add_reads_packed has already run k_sketch_dp_mask over the tiles of these reads (slice by slice, under the upload of the next slice)
	u64 dp_gen = 0;                       // ... into the handle's mask buffers of that generation (lqcov_handle::sk_gen): another read set sketched since, and dp_n is void
};

// What the mapping of a part needs before its first batch, and what depends only on the part, the query set and mid_occ (not
// on what earlier parts accumulated): the probe of every query minimizer, which minimizers can repeat an x, the surviving
// seed hits (k_seed_count) and every offset that follows.  Made right after the part's index on the build stream -- under the
// mapping of the part before -- and swapped into the handle's work buffers of the same names when the part is mapped.
struct lqcov_handle;
struct SeedPlan {
	DBuf hit_start, hit_n, a_cnt, keep, dup, qdirty, dup_table, a_off, mp_off, aq_off, mpq_off, avg_qspan, qklib, mini_pos, qzero;
	DBuf surv, aqf_off;                   // the seed hits that can be part of a chain (kernels_seed.hpp), dense, as records; where every query's start
	std::vector<u64> h_aq, h_qmoff, h_aqf;
	u64 nA_total = 0, n_mp_total = 0, n_written = 0;
	i32 mid_occ = -2; u32 n_q = 0; u64 n_qm = 0;
	u32 rec_jb = 0, rec_db = 0, rec_nmin = 0;   // the records' bit layout (SeedBits); the filter's n_min
	u32 q_begin = 0, q_end = 0;           // the queries whose survivors the plan holds right now (a group of chunks; all of them unless survivors abound)
	bool lazy = false;                    // the survivors are not made yet: every mapping lane runs the filter for its own batch of queries before it maps
	                                      // them (map_part) -- the first part of a job, whose plan nothing else could hide (round 6)
	bool bucketed = false;                // false: the first pass writes every hit (no filter asked for, or the records do not fit 64 bits): h_aqf == h_aq
	bool valid = false;
};

// One run of the seed filter (lqcov_handle::seed_filter) over the queries [q_begin, q_stop) of a part: where the part's probe
// results live right now (in the plan, or swapped into the handle while the part is mapped), where the survivors go.  A run
// ends at q_stop or when `surv` holds LQCOV_SEED_SURV_MAX survivors, whichever comes first (q_end).  Survivors are numbered
// from `base` on (aqf_off / h_aqf hold those numbers): surv[0] is survivor `base`.
struct SeedJob {
	const u64 *hit_start = nullptr; const u32 *hit_n = nullptr, *keep = nullptr; u64 *aqf_off = nullptr;
	const std::vector<u64> *h_qmoff = nullptr; std::vector<u64> *h_aqf = nullptr;
	DBuf *surv = nullptr; u64 base = 0, room_hint = 0;
	u32 q_begin = 0, q_stop = 0;
	u32 q_end = 0; u64 n_surv = 0;        // out
};

// work space of the seed filter: one plan is made at a time
struct SeedWork { DBuf hlen, h_off, hq_off, qg, segs, bq, cnt, off, scnt, soff, rec, has, bd, big; };

struct Part {
	bool live = false, built = false;
	SeedPlan plan;
	ReadSetDev rs;
	DBuf pos;                             // y of every minimizer, grouped by hash, ascending (index.c:188)
	DBuf tkey, tstart, tcnt;              // open-addressed table
	u32 cap_bits = 0;
	u64 n_keys = 0;
	DBuf self_off, self_rid;              // per query: same-name targets (lqmap.c:180-186)
	DBuf t_rank, q_lo;                    // -X only: name ranks for strcmp(qname, tname) > 0 (lqmap.c:187)
};

// One mapping lane: a stream with its own scan/sort scratch and per-batch work space.  Query batches of a part are
// independent (lqmap.c:170-330 runs one query at a time), so lanes run them concurrently.
struct PsWork {                           // one set of psort lists + the scratch of its partition passes (kernels_psort.hpp)
	DBuf big[2], fin_s, fin_b, plan, gcnt, gcur, gdiff, tmap;
};

struct MapLane {
	hipStream_t stream = nullptr;         // klib's passes (queries with repeated minimizers), then runs and chains
	hipStream_t stream2 = nullptr;        // the parallel sort of every other query, meanwhile
	hipStream_t streamW = nullptr;        // the serial token walks: streams confined to a quarter of the CUs (see map_part):
	hipStream_t streamW2 = nullptr;       //   checkpointed walks + their solvers | whole walks of the shorter size classes
	hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_w0 = nullptr, ev_w1 = nullptr, ev_w2 = nullptr;
	DBuf sort_cnt, mhist;
	DBuf ck_segs, ck_T, ck_E, ck_S, ck_slot, ck_n;   // checkpointed walks (kernels_ckpt.hpp)
	PsWork ps[2];
	Prim prim;
	bool gate_passed = false;             // this batch has reached its long walks (see map_part)
	DBuf surv_l, aqf_l; std::vector<u64> h_aqf_l;   // a lazy plan (SeedPlan::lazy): the survivors of the lane's own batch and their per-query offsets
	const u64 *use_surv = nullptr, *use_aqf = nullptr;   // ... which map_batch then reads instead of the plan's
	bool prune = false; u32 prune_n_want = 0, prune_n_sub = 0;   // second pass: klib's levels drop the buckets without a listed run (k_rs_children; L.want, L.sub_off, L.sub_q)
	DBuf sens, n_sens, want, sub_q, sub_off, sub_klib;   // runs left to the second pass (map_batch), its queries
	DBuf A, B, R0, segs0, segs1, n_segs, hist, begs;     // A: anchors (final home), B: originals of the klib queries / other buffer of the parallel sort, R0: records (R1 lives in scr)
	DBuf tile_list, two_tiles, two_tile0, two_tcnt, two_m;
	DBuf sort_d, sort_dst, seg_info, walk_list, two_list, scr;
	DBuf gsel, gkey, gsel2, gkey2, gstart, run_tiles, sel_tiles;
	DBuf ivl, n_ivl, iv_q, iv_q2, iv_se, iv_se2, ivq_off, iv_scratch;
	DBuf arena_buf; LqArena arena;        // the lane's work space (prim.hpp): its buffers are pieces of it, for the length of a batch
	// every buffer of the lane, for the two functions below
	template <class F> void each_buffer(F f)
	{
		for (DBuf *b : { &sens, &n_sens, &want, &sub_q, &sub_off, &sub_klib, &sort_cnt, &mhist, &ck_segs, &ck_T, &ck_E, &ck_S, &ck_slot, &ck_n, &prim.tmp, &A, &B, &R0, &segs0, &segs1, &n_segs, &hist, &begs, &tile_list, &two_tiles, &two_tile0, &two_tcnt, &two_m,
		                 &sort_d, &sort_dst, &seg_info, &walk_list, &two_list, &scr, &gsel, &gkey, &gsel2, &gkey2, &gstart, &run_tiles, &sel_tiles,
		                 &ivl, &n_ivl, &iv_q, &iv_q2, &iv_se, &iv_se2, &ivq_off, &iv_scratch }) f(*b);
		for (PsWork &W : ps) for (DBuf *b : { &W.big[0], &W.big[1], &W.fin_s, &W.fin_b, &W.plan, &W.gcnt, &W.gcur, &W.gdiff, &W.tmap }) f(*b);
	}
	// hand every buffer back (they regrow on the next batch); the caller has drained the lane's streams
	void release_buffers() { each_buffer([](DBuf &b) { b.release(); }); arena.used = 0; arena.base = nullptr; arena.size = 0; arena_buf.release(); }
	// a new batch starts with an empty arena: the pieces the last batch cut from it are forgotten
	void drop_arena_buffers() { each_buffer([](DBuf &b) { if (b.in_arena) b.release(); }); arena.used = 0; }
};

struct lqcov_handle {
	lqcov_params P;
	Knobs K;
	MapParams mp;
	int device = 0;
	hipStream_t stream = nullptr;         // queries, the head of map_part, finish
	hipStream_t bstream = nullptr;        // upload, sketch and index of a part (with bprim): another part may be mapped meanwhile
	hipStream_t cstream = nullptr;        // the upload of packed reads in slices: a slice is sketched on bstream while the next one arrives
	hipEvent_t ev_up[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	Prim prim, bprim;
	u64 hbm_reserve = 0;                  // bytes the caller wants left free when the lanes size their work space (a second part being built)
	std::string err;
	int profiling = 0;                    // 0 off, 1 per-kernel (waits for every kernel, lanes run in turn), 2 events only (read when asked)
	std::string profile_only;             // if not empty: only this stage is timed
	struct StagePending { const char *name; hipEvent_t a, b; u64 bytes; };
	std::vector<StagePending> stage_pending; std::mutex stage_mu;
	void account_stage(const char *name, hipEvent_t a, hipEvent_t b, u64 bytes);
	void add_stage_bytes(const char *name, u64 bytes);       // algorithmic bytes known only after the fact (device-side tallies)
	std::map<std::string, u64> late_bytes;
	void drain_stages();
	u32 debug_flags = 0;
	bool distributed = false;             // per-part accumulators, COVT replayed by the caller (multi-GPU)
	std::map<std::string, StageAcc> stages;
	std::vector<std::string> stage_order;

	// query set and per-query accumulators
	ReadSetDev q;
	bool have_queries = false, q_has_qual = false;
	// queries are held longest first (anchors per query grow with its length: the batch with the longest serial walks then
	// starts first and the last batch has the shortest); q_perm[internal] = position in the caller's order, q_inv the inverse.
	// Everything per query inside the engine (and in the accumulator exchange of the multi-GPU path) uses the internal order;
	// rows, regions, minimizer and chain dumps are handed out in the caller's order.
	std::vector<u32> q_perm, q_inv;
	DBuf q_owner;                         // query of every query minimizer
	DBuf lambda, lambda2, avg_k, cnts, qflags, qual_psum;
	DBuf dup, qdirty, dup_table;          // k_dup_mark: minimizers / queries whose anchors can repeat an x (per part)
	DBuf qklib;                           // queries that go through klib's passes (per part): marked and more than 64 anchors
	DBuf qzero;                           // zeros: nobody goes through klib's passes (first pass of map_batch)
	DBuf surv, aqf_off;                   // the part's surviving seed hits as records and their per-query offsets (SeedPlan, swapped in by map_part)
	SeedWork seed_ws; std::mutex seed_mu;  // work space of the seed filter: one plan (or group of a plan) is made at a time, whichever thread asks
	// Queries one of whose match counters reached cnt_max (uint16 in the reference: 65535; esterr.c:130,136): from the part in
	// which that happens on, their counters live here, replayed part by part in the reference's chain order (sat_replay.hpp,
	// sat_replay_part), and go back to the device before the rows are made.  Key: the query in the engine's order.
	std::map<u32, std::vector<u32>> sat_cnt;
	u32 cnt_max = 65535;                  // (LQCOV_TEST_CNT_BITS)
	DBuf sat_rec, sat_at, sat_n;
	u64 stat_sat_chains = 0;
	std::vector<SatRec> sat_last_recs; std::vector<u32> sat_last_at;   // lqcov_part_sat_records: the records between the sizing call and the copying one
	int sat_last_part = -1; u32 sat_last_query = 0; bool sat_last_valid = false;
	u64 last_n_written = 0;               // anchors the first pass wrote against the last part
	std::atomic<u64> stat_tie_why[6] = {};   // listed runs by the first reason that listed them (lq_tie_list: skip pending, member counts as a skip, top score twice, scan broke off, peak tie, other), since reset()
	std::atomic<u64> stat_sens_runs{0}, stat_p2_queries{0}, stat_p2_anchors{0};   // second pass, since reset(): runs, queries, anchors
	u32 run_n_min() const { const i32 span_max = P.hpc ? 255 : P.k; return (u32)std::max<i32>(std::max<i32>(P.min_cnt, 1), (mp.min_sc + span_max - 1) / span_max); }   // anchors a run needs to hold a chain (k_run_list)
	// counter layout: normally the query minimizer offsets; after adopt_index_params() (prebuilt index with other -k/-w/-H)
	// the reference's sizes (from the command-line sketch, minimap2-coverage.c:419-422) and the mapping's differ
	bool own_cnt_layout = false; DBuf cnt_off, d_nsize; std::vector<u32> h_nsize; u64 cnt_total = 0;
	const u64 *cnt_off_dev() const { return own_cnt_layout ? cnt_off.as<u64>() : q.moff.as<u64>(); }
	u64 cnt_count() const { return own_cnt_layout ? cnt_total : q.n_mini; }
	DBuf pv; DBuf n_pv; u32 pv_cap = 0;   // persisted intervals + markers (ovlp_coords)
	i32 mid_occ = -1;

	std::vector<std::unique_ptr<Part>> parts;

	// work buffers of part_map
	DBuf hit_start, hit_n, a_cnt, keep, a_off, mp_off, mini_pos, aq_off, mpq_off, avg_qspan, skip;
	std::vector<std::unique_ptr<MapLane>> lanes;
	int n_lanes = 3;
	std::mutex pv_mu; u64 pv_reserved = 0;
	std::mutex gate_mu; std::condition_variable gate_cv; int gate_count = 0;   // staggered lane start
	DBuf dbg_chains, n_dbg; u64 dbg_cap = 0; u64 n_dbg_host = 0;
	DBuf misc;
	DBuf ix_key, ix_key2, ix_head, ix_uidx, ix_ukey, ix_ustart, ix_ucnt, ix_sorted;   // build_index workspaces
	const Part *ix_owner = nullptr;       // the part ix_ukey / ix_ustart / ix_ucnt describe (dump_part reads them)
	u64 ix_key_stamp = 0, ix_key_seq = 0; // ix_key holds the sort keys of the read set with this stamp (written by k_sketch_emit_mask with its x and y)
	DBuf sk_ulist;                        // sketch: the chunks the data-parallel kernel left to the machine + their count
	DBuf sk_cnt, sk_off, sk_owned, sk_mask, sk_flag, sk_toff, sk_trid, sk_grid;   // sketch: per-chunk minimizer counts / offsets, which kernel decides a chunk, emitted positions (a bit per base), tile offsets
	std::vector<u64> sk_h_toff;           // tiles of k_sketch_dp_mask before every read (host copy of sk_toff)
	const ReadSetDev *sk_owner = nullptr; u64 sk_gen = 0;   // whose reads the sk_* buffers describe right now (sketch_dp_setup): a read set's dp_n counts only while they are its own
	u64 last_n_anchors = 0;
	u64 anchor_budget = 0;

	// finish()
	bool finished = false;
	std::vector<lqcov_row> rows;
	std::vector<lqcov_region> regs, mregs;

	lqcov_handle(const lqcov_params &p, int dev);
	~lqcov_handle();

	void add_reads(ReadSetDev &rs, u32 n, const u8 *seq, const u64 *seq_off, const char *names, const u64 *name_off);
	void amb_tails(ReadSetDev &rs, u32 r0, u32 r1, u64 chunk0, u64 n_chunks);
	void add_reads_packed(ReadSetDev &rs, u32 n, const u64 *codes, const u32 *amb, const u32 *lens, const char *names, const u64 *name_off,
	                      const u64 *codes_dev = nullptr, const u32 *amb_dev = nullptr, u64 stride_chunks = 0, const std::vector<u64> *share_chunks = nullptr);
	void sketch(ReadSetDev &rs, bool rid_in_y);
	bool sketch_dp_setup(ReadSetDev &rs, u64 &n_tiles);
	void sketch_dp_launch(ReadSetDev &rs, u64 tile0, u64 tile1);
	void export_minimizers(ReadSetDev &rs, u64 *x_dev, u64 *y_dev, u32 rid_base);
	void set_queries(u32 n, const u8 *seq, const u64 *seq_off, const u8 *qual, const char *names, const u64 *name_off);
	void build_index(Part &pt);
	void build_part(Part &pt);
	void open_gate();
	void plan_part(Part &pt, hipStream_t s, Prim &pr, bool defer_filter = false);
	std::atomic<int> active_maps{0};      // map_part calls in progress: a part built meanwhile gets its whole plan ahead of time, a part built with the lanes idle a lazy one
	bool seed_filter(Part &pt, hipStream_t s, Prim &pr, SeedWork &W, u32 n_min, u32 jb, u32 db, SeedJob &J);
	bool seed_group(Part &pt, SeedPlan &S, bool swapped, hipStream_t s, Prim &pr, u32 q_begin);
	void swap_plan(SeedPlan &S);
	void map_part(Part &pt);
	void map_batch(MapLane &L, Part &pt, u32 q0, u32 q1, const std::vector<u64> &h_aq, const std::vector<u64> &h_aqf, const std::vector<u64> &h_qmoff, bool dbg);
	void batch_buffers(MapLane &L, u64 nA);
	void chain_stage(MapLane &L, Part &pt, const u64 *aqb, u64 a_base, u32 nqb, u32 q0, const u32 *qmap, u64 nA, int tie_mode, u32 n_want, u32 ivl_cap, bool dbg, const SatSink *sink = nullptr);
	void map_subset(MapLane &L, Part &pt, const std::vector<u32> &sq, const std::vector<u32> &sk, const std::vector<u64> &so, u64 max_mini, int tie_mode, u32 n_want, u32 ivl_cap, bool dbg, const SatSink *sink);
	void debug_sort_pairs(u64 *keys, u64 *vals, u64 n, unsigned bits, int key_bytes);   // tests: the primitives of kernels_isort.hpp on host arrays
	void debug_scan(const u32 *in, u64 *out, u64 n);
	bool sat_chains(Part &pt, u32 qi, const std::vector<u64> &h_aq, const std::vector<u64> &h_qmoff, std::vector<SatRec> &recs, std::vector<u32> &at);
	void sat_check(const std::vector<SatRec> &recs, const std::vector<u32> &at, size_t nc);
	void part_sat_records(Part &pt, u32 qi, std::vector<SatRec> &recs, std::vector<u32> &at);
	void sat_replay_host(u32 qi, const SatRec *recs, u64 n_recs, const u32 *at, u64 n_at, u32 *counters, u64 n_counters);
	void sat_replay_part(Part &pt, const std::vector<u64> &h_aq, const std::vector<u64> &h_qmoff);
	void sort_checked(MapLane &L, Part &pt, const u64 *aqb, const u32 *qkb, u32 nqb, u64 a_base, u64 nA, const std::vector<u64> &h_off, const std::vector<u32> &h_klib);
	void sort_batch(MapLane &L, Part &pt, const u64 *aqb, const u32 *qkb, u32 nqb, u64 a_base, u64 nA);
	void psort_run(MapLane &L, int set, hipStream_t s, u64 nA, const KeyMap &km, const struct PsData &pd);
	void psort_tail(MapLane &L, int set, hipStream_t s, u64 nA, const KeyMap &km, const struct PsData &pd);
	void reset();
	void finish();
	void write_table(FILE *out);
	int run_files(const char *target, const char *query, FILE *out, FILE *log, const char *dump_path = nullptr);
	void adopt_index_params(i32 k, i32 w, i32 hpc);
	void dump_part(Part &pt, FILE *fp);                      // mm_idx_dump (index.c:390-426)
	bool load_part(FILE *fp, Part &pt);                      // mm_idx_load (index.c:428-479); false at end of file
	void build_part_from_host_minimizers(Part &pt, const std::vector<u64> &x, const std::vector<u64> &y,
	                                     std::vector<std::string> &&names, std::vector<u32> &&lens);
	Part &part(int id);
};

u64 lq_packed_chunks(u32 n, const u64 *seq_off);
#include <functional>
void lq_format_rows(FILE *out, int filter_flag, const lqcov_row *rows, u32 n_rows, const lqcov_region *regs, const lqcov_region *mregs,
                    const std::function<const char *(u32)> &name_of);
void lq_pack_host(u32 n, const u8 *seq, const u64 *seq_off, u64 *codes, u32 *amb, int n_threads);
bool lq_packed_read_ambiguous(const u32 *aw, u64 len);

struct StageTimer {
	lqcov_handle *h; hipStream_t s; const char *name; u64 bytes;
	hipEvent_t a = nullptr, b = nullptr; bool on = false;
	StageTimer(lqcov_handle *h_, const char *name_, u64 bytes_ = 0);
	StageTimer(lqcov_handle *h_, hipStream_t s_, const char *name_, u64 bytes_ = 0);
	~StageTimer();
};
