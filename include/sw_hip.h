/*
 * include/sw_hip.h -- ADDITIVE entry points of libsw_hip.so.
 *
 * The reference's C ABI is one pair per call (rust/src/bindings.rs:71-157,187-229) and cannot express a
 * batch, a resident database, top-k or a score above 65535.  These symbols have no
 * counterpart in rust/src/bindings.rs; each one cites the reference code whose job it takes over.  The
 * five drop-in symbols of include/sw.h stay and give reference-identical answers for one pair.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * SW_E_* code and records a message readable with sw_hip_last_error(); nothing aborts or throws.
 */
#ifndef DLSA_AMD_SW_HIP_H
#define DLSA_AMD_SW_HIP_H

#include "sw.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SW_OK 0
#define SW_E_NOGPU -1    /* no usable HIP device / kernels missing: there is NO CPU fallback */
#define SW_E_ARG -2      /* bad argument (NULL pointer, invalid UTF-8, out-of-range size) */
#define SW_E_HIP -3      /* a HIP runtime call failed */
#define SW_E_NOMEM -4
#define SW_E_DOMAIN -5   /* scoring outside the supported domain for this entry point */
#define SW_E_LIMIT -6    /* the input exceeds a stated size limit of this entry point */

/* which reference entry point's semantics to reproduce */
#define SW_PATH_LOW_MEMORY 0 /* rust/src/algorithm/mod.rs:204-445 incl. the T < 2Q dispatch at :213 */
#define SW_PATH_SIMD 1       /* rust/src/lib.rs:105-136 */
#define SW_PATH_STRAIGHT 2   /* rust/src/lib.rs:45-71 */

/* ---- lifecycle ------------------------------------------------------------------------------------ */
/* Selects the HIP device for the calling process (one process per GPU) and creates the library's
 * stream.  Optional: every entry point initialises device 0 lazily.  Plays the role of "the Rust crate
 * is stateless" (bindings.rs has no init) for a library that owns device state. */
int sw_hip_init(int device);
/* Thread-local message of the last failure on this thread ("" if none). */
const char *sw_hip_last_error(void);
/* Library/version string, the analogue of rust/src/lib.rs:214-216 get_version(). */
const char *sw_hip_version(void);

/* ---- wide score ----------------------------------------------------------------------------------- */
/* The un-truncated int32 score of a Result returned by this library (struct Result carries 16 bits,
 * rust/header/sw.h:8).  Equal to (int16_t)result->score on the reference's i16-safe domain.  Every entry point
 * computes in 32 bits and refuses (NULL / SW_E_ARG, message in sw_hip_last_error) a pair whose score could reach 2^29:
 * (strlen(query) + strlen(target)) * max(match, mismatch, gap, 1) + 65536 must stay below 2^29. */
int32_t sw_result_score32(const struct Result *result);

/* ---- batch of independent pairs: what internal/worker/worker_impl.go:108-168 (ExecuteWork) does with
 * one cgo call per pair.  out[i] receives what find_alignment_low_memory / _simd / _sequential_straight
 * (per `path`) would return for (queries[i], targets[i]); free each with free_alignment_result().  All
 * pairs run on the GPU in one submission. */
int sw_align_batch(const char *const *queries, const char *const *targets, size_t n_pairs,
                   struct AlignmentScores scores, int path, struct Result **out);

/* free_alignment_result() for every non-NULL entry of results[0..n) (entries are set to NULL): the results of one
 * sw_align_batch call share one block of memory, and this releases it with one reference-count update per run of
 * results instead of one per result.  Mixing it with free_alignment_result / free_c_string on other results of the
 * same batch is fine. */
void sw_free_alignment_results(struct Result **results, size_t n);

/* ---- resident target database + top-k: what cli/__main__.py:153-173 computes client-side after every
 * pair's result has travelled worker -> master -> CLI. */
typedef struct sw_db sw_db;

/* Builds a database from n sequences stored back to back in `flat`; sequence i is
 * flat[offsets[i] .. offsets[i+1]) (offsets has n+1 entries).  Sequences are length-sorted, 2-bit
 * packed in groups of 128 (DESIGN.md "Data layout") and uploaded; sequences with symbols outside
 * ACGT are kept as bytes and scored by the exact kernel.  Returns NULL on failure. */
sw_db *sw_db_create(const char *flat, const uint64_t *offsets, uint64_t n);
/* Parses a FASTA file the way cli/__main__.py:19-34 does (id = first token after '>', multi-line
 * bodies concatenated). */
sw_db *sw_db_load_fasta(const char *path);
/* The packed database on disk: everything sw_db_create computes from raw sequences (validation, length sort,
 * lane groups, the 2-bit strip stream, the escape list of sequences outside ACGT), stored as it goes to the
 * device, about a quarter of the FASTA size.  sw_db_open is a read, a consistency check of every table against
 * the others, and an upload; searches on a re-opened database give identical results.  The input side of the
 * reference is FASTA parsed on every submission (cli/__main__.py:19-34) and one HTTP GET per sequence per work
 * package (internal/worker/master_rest_client.go:185-204). */
int sw_db_save(sw_db *db, const char *path);
sw_db *sw_db_open(const char *path);
void sw_db_free(sw_db *db);
uint64_t sw_db_size(const sw_db *db);
/* total residues */
uint64_t sw_db_residues(const sw_db *db);

/* One reported hit.  Ordering of hits: score descending, then target index ascending (the reference's
 * order among equal scores is the master's arrival order, cli/__main__.py:171 + SURVEY.md A10, i.e. not
 * deterministic; this is the declared replacement). */
struct sw_hit {
  uint64_t target_index;  /* index into the database as passed to sw_db_create */
  int32_t score;          /* un-truncated */
  uint32_t length;        /* strlen(query_aln) -- the worker's AlignmentDetail.Length, worker_impl.go:147 */
  uint64_t max_x, max_y;  /* as struct Result */
  char *query_aln;        /* free with free_c_string */
  char *target_aln;
};

/* Scores `query` against every database sequence with the semantics the worker sees for a pair -- the fallback
 * chain of worker_impl.go:170-202: find_alignment_low_memory, or find_alignment_simd where that one fails (gap == 0
 * on the ring path) -- and returns the k best hits with full alignments.  *n_hits <= k.  Exact for every
 * reported field. */
int sw_db_search_topk(sw_db *db, const char *query, struct AlignmentScores scores, int k,
                      struct sw_hit *hits, int *n_hits);
/* The same for n_queries queries in one call: what a work package of the reference holds (many queries against
 * the same targets, internal/worker/worker_impl.go:108-168 loops over its pairs).  hits has room for
 * n_queries * k records, query q's hits start at hits[q * k]; n_hits has n_queries entries.  Consecutive queries
 * of at most 1024 residues share a launch of the score kernel (up to 16 queries / 4096 rows, one score array per
 * query; SW_HIP_QUERY_BATCH=0 switches that off, =n caps a batch at n queries); the on-device top-k selection runs
 * per query; end cells and tracebacks of all reported hits go to the GPU in one submission.  Results do not depend
 * on how the queries were grouped. */
int sw_db_search_topk_batch(sw_db *db, const char *const *queries, size_t n_queries, struct AlignmentScores scores,
                            int k, struct sw_hit *hits, int *n_hits);
/* All scores (int32, database order), for tests and for schedulers that want every pair. */
int sw_db_scores(sw_db *db, const char *query, struct AlignmentScores scores, int32_t *scores_out);
void sw_hits_free(struct sw_hit *hits, int n_hits);

/* ---- the one entry point that allocates nothing --------------------------------------------------------------
 * No counterpart in the reference.  For callers that cannot call hipMalloc: tight serving loops, harnesses that
 * forbid allocation, code that runs while another stream of the process is under HIP graph capture (an allocation
 * there would invalidate the capture).  The call itself is NOT capturable: it synchronises its own stream, reads
 * event timers and finishes on the host.  sw_db_scores for the same arguments, with the same result,
 * but every byte its kernels need beyond the inputs (the resident database, the query) and the output lives in a
 * file-scope device array of sw_hip_noalloc_scratch_bytes() bytes (96 MiB; every GPU the library runs on holds
 * its own copy until the process ends) and in a page-locked host arena reserved when the device's context is set
 * up (sw_hip_init / first use of the device), so its copies are asynchronous and the runtime stages nothing.  The
 * call performs no hipMalloc, hipFree or hipHostMalloc; only an error path builds a message string on the host
 * heap.  Limits, each an error (SW_E_LIMIT / SW_E_DOMAIN with a
 * message), never a fallback to allocation:
 *   - scoring with match > 0, mismatch < 0, gap < 0; query and every non-empty database sequence over ACGT only
 *     (the packed kernel's domain; anything else: use sw_db_scores);
 *   - query <= 65536 residues, and short enough that the fixed boundary ring (72 MiB) holds a few columns of it:
 *     about 24 500 residues; fewer workgroups run the longer the query is (256 up to ~200 residues, ~110 at
 *     1 kb, 9 at 10 kb), so long queries are slower here than through sw_db_scores;
 *   - database <= 2 097 152 lane slots (targets rounded up to groups of 128) and <= 131 072 work units;
 *   - targets that need the int32 re-run (int16 saturation, padded-length quirks) are re-scored one pair per
 *     submission inside the same scratch: exact, but slow when there are many.
 * All calls on one GPU share that GPU's array; the library serialises them with the device's lock, callers that
 * bypass the library's lock (there are none in this ABI) would race.  The array keeps the last call's contents;
 * what must start at zero is cleared by every call. */
int sw_db_scores_noalloc(sw_db *db, const char *query, struct AlignmentScores scores, int32_t *scores_out);
uint64_t sw_hip_noalloc_scratch_bytes(void);

/* ---- one worker process, several GPUs -----------------------------------------------------
 * The reference spreads a job over worker PROCESSES by (query, target) pair
 * (master/work_package/_scheduler/proportional_work_scheduler.py:30-55) and sorts every pair's score client-side
 * (cli/__main__.py:153-173).  A node handle lets ONE worker process drive all GPUs of its machine: the target
 * database is sharded by target, balanced by residues within length buckets; every GPU searches its shard on its
 * own host thread, stream and scratch and keeps its own top-k; the only cross-GPU traffic is a host merge of
 * n_queries x k records by (score descending, database index ascending); end cells and tracebacks are computed
 * for the merged hits only, on the GPU that owns the target.  Results are identical to sw_db_search_topk_batch on
 * the unsharded database. */
typedef struct sw_node sw_node;
typedef struct sw_node_db sw_node_db;
/* devices[i] = HIP device of shard i; a device may appear more than once (logical shards: tests, or a rehearsal
 * of the N-GPU path on one GPU).  devices == NULL: devices 0 .. n-1 of the machine (n_devices <= 0: all). */
sw_node *sw_node_create(const int *devices, int n_devices);
void sw_node_free(sw_node *node);
int sw_node_shards(const sw_node *node);
/* The sharder alone (host arithmetic, no GPU): shard_of[i] = shard of sequence i for a database given by n + 1
 * offsets.  Sequences are length-sorted, cut into tiles of up to 128 consecutive ones (one wave's worth of lanes)
 * and the tiles dealt longest-first to the shard with the fewest residues so far. */
int sw_shard_plan(const uint64_t *offsets, uint64_t n, int n_shards, uint32_t *shard_of);
sw_node_db *sw_node_db_create(sw_node *node, const char *flat, const uint64_t *offsets, uint64_t n);
/* Per-GPU shard files: <dir>/shard-000.swdb ... (sw_db_save of every shard) plus <dir>/node.swnode (which
 * database index each shard-local index stands for).  sw_node_db_open needs a node with the same number of shards. */
int sw_node_db_save(sw_node_db *db, const char *dir);
sw_node_db *sw_node_db_open(sw_node *node, const char *dir);
void sw_node_db_free(sw_node_db *db);
uint64_t sw_node_db_size(const sw_node_db *db);
int sw_node_db_shards(const sw_node_db *db);
int sw_node_db_shard_info(const sw_node_db *db, int shard, int *device, uint64_t *n_targets, uint64_t *residues);
/* sw_db_search_topk_batch over all shards; target_index of a hit is the index as passed to sw_node_db_create. */
int sw_node_search_topk_batch(sw_node_db *db, const char *const *queries, size_t n_queries,
                              struct AlignmentScores scores, int k, struct sw_hit *hits, int *n_hits);

/* ---- measurement hooks (bench.py) -------------------------------------------------------------------- */
struct sw_kernel_stats {
  double score_kernel_ms;   /* HIP-event time of the last inter-task score-kernel launch(es), on the
                               library's stream */
  double exact_kernel_ms;   /* same for the exact (intra-task) kernels + traceback walker */
  uint64_t score_cells;     /* sum Q*T (unpadded) the score kernel covered */
  uint64_t exact_cells;
  uint32_t score_launches;
  uint32_t rerun_pairs;     /* pairs re-run by the exact kernel (saturation / quirk / non-ACGT) */
  /* what the score-kernel launches executed, counted by the library (no pasted constants in bench.py): */
  uint64_t score_strip_rows;   /* sum over launches of (16-column strips of the database) x (query rows): one
                                  iteration of the generated row loop each, score_packed_per_row packed VALU
                                  wave-instructions + score_aux_per_2rows / 2 address increments.  The last strip of
                                  a group whose targets all end within its first 8 columns runs the 8-column form of
                                  the loop (half the instructions) and counts as half a strip */
  uint64_t score_chain_rows;   /* of those, rows also stored to the HBM boundary ring (256 B written once and
                                  read back once each): units that hand their last column to the next unit */
  uint32_t score_packed_per_row, score_aux_per_2rows; /* read off the generated loop (sw_score_asm.inc) */
  uint32_t score_workgroups;   /* persistent workgroups of the last launch */
  uint32_t pad_;
  /* the one-pair-per-lane exact kernel (batches of short pairs, sw_lanes.hip); its time is ALSO part of
     exact_kernel_ms and its cells of exact_cells */
  double lanes_kernel_ms;      /* fill + traceback walker */
  uint64_t lanes_cells;        /* sum Q*T (unpadded) of the pairs it answered */
  uint64_t lanes_padded_cells; /* cells its waves computed: 64 x rows x 16 x strips per group */
  uint64_t lanes_pairs;
};
int sw_hip_last_stats(struct sw_kernel_stats *out);

/* The worker's capacity benchmark (internal/worker/worker_rust_benchmark.go:38-68, consumed by
 * master/api_models/worker.py:7-16): cells per second of the batch path on synthetic DNA, as a
 * float32-representable INTEGER-valued number (the master's field is a pydantic int). */
float sw_hip_benchmark_cups(void);
/* sw_align_batch over the GPUs of a node: the pairs are cut into one part per shard of the handle (pairs of one
 * query kept together where the balance allows, parts balanced by cells, deterministic), every part runs as an
 * ordinary batch on its device at the same time, and out[i] is what sw_align_batch would have written -- pairs are
 * independent, so the split cannot change a result.  The reference distributes by pair over worker processes
 * (master/work_package/_scheduler/proportional_work_scheduler.py:30-55); this is the same split inside one
 * process.  On failure nothing is handed over. */
int sw_node_align_batch(const sw_node *node, const char *const *queries, const char *const *targets, size_t n_pairs,
                        struct AlignmentScores scores, int path, struct Result **out);
/* The same for a whole node: every distinct device of the handle runs the benchmark at the same time and the
 * figures are added -- the number a one-process-per-node GPU worker registers with the master. */
float sw_node_benchmark_cups(sw_node *node);
/* sw_hip_last_stats for one shard of a node database: the counters of THAT shard's last search (logical shards
 * that share a device do not overwrite each other); a caller adds the shards up. */
int sw_node_last_stats(const sw_node_db *db, int shard, struct sw_kernel_stats *out);

/* Scratch budget of one device (-1: the default device) in MiB: what one exact submission may take for direction
 * bits and boundary rows (larger batches are cut), and half of what the score kernel's boundary rings may take
 * (a work package beyond that is processed in slices of its targets).  Default 8192, or SW_HIP_SCRATCH_MB at the
 * time the device is first used.  A deployment limit, and the way the tests reach the slicing code with small
 * inputs. */
int sw_hip_set_scratch_mb(int device, uint64_t mb);

/* ---- test hook (no GPU needed) ------------------------------------------------------------------------ */
/* The four integers by which one plain Smith-Waterman reproduces a reference entry point for a pair of
 * lengths (Q, T) (DESIGN.md "Effective semantics"): out[0] = last column computed for rows x >= 2,
 * out[1] = last column computed for row x == 1, out[2] = Qp if the ring path's two-phase discovery order
 * applies (else 0), out[3] = 1 for the straight entry's tie order.  Pure host arithmetic; lets the CPU test
 * suite check the library's dispatch (rust/src/algorithm/mod.rs:213 and the padded-length quirks) against
 * the oracle's clean model without a GPU. */
int sw_hip_path_params(int path, uint64_t q_len, uint64_t t_len, uint32_t out[4]);
/* Counters of the CALLING THREAD's last sw_align_batch / sw_execute_work / sw_node_align_batch /
 * sw_node_execute_work call (a node call adds up its devices) that took the work-package fast path (long ACGT
 * targets: score pass by the packed inter-task kernel over overlapping chunks of the targets, end cell and traceback
 * inside windows of at most 256 target columns): out[0] = pairs it answered, out[1] = score passes (queries),
 * out[2] = windows filled by the exact kernel, out[3] = rounds (1 + continuations to the left), out[4] = cells of
 * those windows, out[5] = chunks the targets were cut into, out[6] = tracebacks that tried to leave their chunk
 * (must be 0: the overlap bound; such a pair is redone by the whole-matrix path).  All zero if the last call did
 * not take it. */
void sw_hip_package_stats(uint64_t out[7]);
/* sizeof() of the additive ABI structs as the library was compiled: 0 = sw_kernel_stats, 1 = sw_hit,
 * 2 = sw_alignment_detail, 3 = sw_work_package, 4 = sw_work_pair (0 for an unknown index).  Lets a binding written
 * in another language (ctypes, cgo) assert that its mirror of a struct has not fallen behind the header. */
uint32_t sw_hip_struct_size(int which);
/* Builds the score kernel's ticket-ordered unit table for a database whose groups have the given numbers of strips
 * (W strips per unit, S interleaved lanes, R ring slots, max_wg persistent workgroups) and checks the invariants
 * the kernel relies on (coverage, backward dependencies, ring-slot reuse).  Returns the number of units or a
 * negative code; out[0] = short units that cut the partial last round, out[1] = their size.  Host arithmetic. */
int64_t sw_hip_test_units(const uint32_t *group_strips, uint32_t n_groups, uint32_t W, uint32_t S, uint32_t R,
                          uint32_t max_wg, uint32_t out[2]);

#ifdef __cplusplus
}
#endif
#endif /* DLSA_AMD_SW_HIP_H */
