/*
 * include/sw_worker.h -- the worker-side callers of the hot path, as C entry points of libsw_hip.so.
 *
 * The reference's callers are Go (internal/worker/worker_impl.go, worker_rust_benchmark.go) and Python
 * (cli/__main__.py); Go is not available where this library is built, so the per-work-package logic that sits
 * directly on top of the cgo boundary is provided here behind a C ABI a Go worker can bind with the same few
 * cgo lines as include/sw.h.  Everything here is host code over include/sw.h and
 * include/sw_hip.h; no alignment is computed on the CPU.
 *
 * Out of scope on purpose: the REST client, heartbeat, registration (internal/worker/master_rest_client.go) --
 * the wire structs are only mirrored as far as ExecuteWork fills them.
 */
#ifndef DLSA_AMD_SW_WORKER_H
#define DLSA_AMD_SW_WORKER_H

#include "sw_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ExecuteWork ----------------------------------------------------------------------------------------
 * A CompleteWorkPackage (internal/worker/master_rest_client.go:30-42): the distinct sequences of the package
 * and the (query, target) combinations to align, as indices into `sequences`.  The three scores are what the
 * master sends (WorkPackage.MatchScore / MismatchPenalty / GapPenalty): penalties are POSITIVE numbers here;
 * the sign flip of worker_impl.go:126-127 happens inside sw_execute_work. */
struct sw_work_pair {
  uint32_t query;
  uint32_t target;
};
struct sw_work_package {
  const char *const *sequences; /* NUL-terminated UTF-8 */
  uint64_t n_sequences;
  const struct sw_work_pair *pairs;
  uint64_t n_pairs;
  int32_t match_score;
  int32_t mismatch_penalty;
  int32_t gap_penalty;
};
/* One AlignmentDetail (master_rest_client.go:21-28,65-68) as ExecuteWork fills it (worker_impl.go:136-155). */
struct sw_alignment_detail {
  uint32_t query;          /* combination.query  (index into sequences) */
  uint32_t target;         /* combination.target */
  char *query_alignment;   /* free with sw_alignment_details_free */
  char *target_alignment;
  int64_t length;          /* len(qRes): BYTES of query_alignment (Go len() of a string) */
  int64_t score;           /* int(rustRes.Score): the uint16 field, 0..65535 */
  int64_t max_x;           /* int(rustRes.MaxX): 2^64-1 becomes -1 */
  int64_t max_y;
  int32_t score32;         /* additive: the un-truncated score (sw_result_score32) */
  int32_t entry;           /* additive: which entry point answered -- SW_PATH_LOW_MEMORY / _SIMD / _STRAIGHT
                              (the fallback chain of worker_impl.go:170-202), -1 if all three failed */
  void *owner_;            /* internal: the library's Result the two strings belong to (they are not copied);
                              released by sw_alignment_details_free */
};
/* Return value of sw_execute_work when every pair was computed but `n` of them failed on all three entry points
 * (out[i].entry == -1, empty strings): SW_W_FAILED_PAIRS + n > 0.  The reference dereferences nil at that point
 * (worker_impl.go:136); a caller should log or skip those pairs rather than post a score-0 alignment. */
#define SW_W_FAILED_PAIRS 0
/* What Worker.ExecuteWork (worker_impl.go:108-168) computes for its pair list, in ONE GPU submission per
 * fallback level instead of one cgo call per pair: out[i] is the AlignmentDetail of pairs[i], bit-identical
 * to the reference's (same entry-point order: find_alignment_low_memory, then find_alignment_simd for the
 * pairs where that returned NULL, then find_alignment_sequential_straight).  Returns SW_OK (even if some pair
 * failed on all three entries: out[i].entry == -1, empty strings -- the reference dereferences nil there,
 * worker_impl.go:136) or a negative SW_E_* code when nothing could be computed (no GPU, bad index). */
int sw_execute_work(const struct sw_work_package *work, struct sw_alignment_detail *out);
/* The same over the GPUs of a node handle (sw_node_align_batch per fallback level): what a one-process-per-node
 * GPU worker, registered with sw_node_benchmark_cups(), calls per work package.  Same results, same return codes. */
int sw_node_execute_work(const sw_node *node, const struct sw_work_package *work, struct sw_alignment_detail *out);
void sw_alignment_details_free(struct sw_alignment_detail *details, uint64_t n);

/* ---- capacity benchmark -------------------------------------------------------------------------------------
 * BenchmarkRust (internal/worker/worker_rust_benchmark.go:38-68) with the reference's protocol, unchanged: one
 * synthetic pair ("TTTTCA" + 'A' * nQ vs "TTTTCA" + 'T' * nT, match 1 / mismatch 1 / gap 2) through the
 * fallback chain, target doubled from 2^18 until one call exceeds `threshold_seconds`, then q_steps x t_steps
 * timed runs; returns the mean cells per second as float32.  cmd/worker/main.go:43 calls it with
 * (10 ms, 1, 2).  This is the number a drop-in (one pair per call) worker would register; a worker that uses
 * sw_execute_work / the database API registers sw_hip_benchmark_cups() instead. */
float sw_benchmark_rust(double threshold_seconds, int q_steps, int t_steps);

/* ---- FASTA in, result files out (the formats either side of the path) --------------------------------------------
 * parse_fasta of cli/__main__.py:19-34, exactly: universal newlines, split on every '>', the record is
 * stripped, id = first whitespace-delimited token of its first line, sequence = the remaining lines
 * concatenated.  Returns a handle or NULL (sw_hip_last_error()). */
typedef struct sw_fasta sw_fasta;
sw_fasta *sw_fasta_parse(const char *path);
uint64_t sw_fasta_count(const sw_fasta *f);
const char *sw_fasta_id(const sw_fasta *f, uint64_t i);
const char *sw_fasta_sequence(const sw_fasta *f, uint64_t i);
void sw_fasta_free(sw_fasta *f);
/* Uploads the records as a resident database (sw_db_create on the parsed sequences). */
sw_db *sw_db_from_fasta(const sw_fasta *f);

/* ---- synthetic sequences ------------------------------------------------------------------------------------------
 * utils/generate_synthetic_data.py:7-12 restated with a fixed PRNG (the script uses Python's unseeded `random`):
 * sequence i (0-based; its FASTA id is "seq{i+1}") has a length uniform in [min_len, max_len] and i.i.d. uniform
 * residues over "ATCG" (the script's alphabet order).  Length and residues derive from (seed, i) alone
 * (splitmix64 -> xoshiro256**), so any subset -- one GPU's shard -- can be generated on its own and two machines
 * produce the same bytes. */
/* lengths[k] = length of sequence first + k, for k < n */
int sw_synth_lengths(uint64_t seed, uint64_t first, uint64_t n, uint32_t min_len, uint32_t max_len, uint32_t *lengths);
/* Writes the residues of sequences indices[0..n) (indices == NULL: 0 .. n-1) back to back into flat;
 * offsets (n + 1 entries, offsets[k+1] - offsets[k] = the sequence's length) says where each one goes. */
int sw_synth_fill(uint64_t seed, const uint64_t *indices, uint64_t n, uint32_t min_len, uint32_t max_len, char *flat,
                  const uint64_t *offsets);
/* The script's file format: ">seq{i}\n<sequence>\n" for i = 1 .. n. */
int sw_synth_write_fasta(const char *path, uint64_t seed, uint64_t n, uint32_t min_len, uint32_t max_len);

/* Appends the hits of one query to <results_dir>/<query_id>.txt in the CLI's format
 * (cli/__main__.py:175-194): per hit ">target\nAlignment: <query alignment>\nLength: <n>\nScore: <s>\n\n";
 * the directory is created if missing, an existing file is appended to.  target_ids[i] names hits[i]. */
int sw_write_query_results(const char *results_dir, const char *query_id, const char *const *target_ids,
                           const struct sw_hit *hits, int n_hits);

#ifdef __cplusplus
}
#endif
#endif /* DLSA_AMD_SW_WORKER_H */
