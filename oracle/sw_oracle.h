/*
 * oracle/sw_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the Smith-Waterman hot path of Noorts/DLSA, written from a reading
 * of the reference's Rust sources.  It exists to check the HIP library; it is never linked into,
 * called from or shipped with the product (dlsa_amd/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.
 *
 * Parity pin: the reference's Rust/Go toolchains are absent from the build image, so the reference
 * cannot be run.  The restatement is pinned by the reference's own known-answer tests
 * (rust/src/lib.rs:271-410, 21 vectors x 4 functions; tests/golden/kat_reference.json).  Those
 * vectors assert aligned strings only; numeric score, max_x/max_y and the no-hit convention are
 * "parity unpinned" by the reference and are pinned here by the line-by-line restatement below.
 *
 * Sequences are arrays of Unicode scalar values (Rust `char`), exactly what the reference's
 * algorithms receive after bindings.rs decodes the C strings (rust/src/bindings.rs:125-129).
 */
#ifndef SW_ORACLE_H
#define SW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rust/src/algorithm/mod.rs:18-24 (field order gap, match, miss; all i16) */
typedef struct {
  int16_t gap;
  int16_t match;
  int16_t miss;
} oracle_scores;

/* rust/src/lib.rs:15  type AlignResult = (Vec<char>, Vec<char>, i16, usize, usize) */
typedef struct {
  uint32_t *q_aln; /* malloc'ed, aln_len entries, '-' = 45 */
  uint32_t *t_aln;
  size_t aln_len;
  int16_t score;
  uint64_t max_x; /* usize, wrapping arithmetic as in a release build */
  uint64_t max_y;
} oracle_result;

/* return codes: 0 = Ok, ORACLE_PANIC = the Rust code would have panicked at this point (index out of
 * bounds, division by zero, assert!, "Mismatch in data") which bindings.rs turns into NULL
 * (rust/src/bindings.rs:109-112,153-156). */
#define ORACLE_OK 0
#define ORACLE_PANIC 1
#define ORACLE_NOMEM 2

/* rust/src/lib.rs:105-136 find_alignment_simd::<64> (full skewed matrix) */
int oracle_find_alignment_simd(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                               oracle_scores s, oracle_result *out);
/* rust/src/algorithm/mod.rs:204-445 find_alignment_simd_lowmem::<64> (dispatch + ring buffer) */
int oracle_find_alignment_simd_lowmem(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                                      oracle_scores s, oracle_result *out);
/* rust/src/lib.rs:45-71 find_alignment_sequential_straight */
int oracle_find_alignment_sequential_straight(const uint32_t *q, size_t Q, const uint32_t *t,
                                              size_t T, oracle_scores s, oracle_result *out);
/* rust/src/lib.rs:17-43 find_alignment_sequential (not exported by bindings.rs, but run against the
 * same known-answer vectors at rust/src/lib.rs:423-426) */
int oracle_find_alignment_sequential(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                                     oracle_scores s, oracle_result *out);

void oracle_result_free(oracle_result *r);

/* The Go worker's fallback chain (internal/worker/worker_impl.go:170-202): low_memory, then simd,
 * then sequential_straight.  *which = 0/1/2 for the entry that produced the result, -1 if all three
 * "panicked" (the Go code then dereferences nil). */
int oracle_find_alignment_with_fallback(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                                        oracle_scores s, oracle_result *out, int *which);

/* ---- "clean" model: the semantics the HIP library implements ("Effective
 * semantics").  Plain int32 Smith-Waterman over the reference's *effective* cell domain, the
 * reference's tie order and traceback priority.  tests/ proves clean == restatement on the i16-safe
 * domain, including the quirk inputs. */
#define ORACLE_PATH_LOWMEM 0 /* dispatch on T < 2Q like find_alignment_low_memory */
#define ORACLE_PATH_SIMD 1   /* always the full-matrix semantics */
#define ORACLE_PATH_STRAIGHT 2
typedef struct {
  uint32_t *q_aln;
  uint32_t *t_aln;
  size_t aln_len;
  int32_t score32; /* un-truncated */
  uint64_t max_x;
  uint64_t max_y;
} oracle_clean_result;
int oracle_clean_align(const uint32_t *q, size_t Q, const uint32_t *t, size_t T, oracle_scores s,
                       int path, oracle_clean_result *out);
void oracle_clean_result_free(oracle_clean_result *r);
/* score only, int32, standard SW over the full Q x T domain (no quirks) */
int32_t oracle_clean_score_plain(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                                 oracle_scores s);

/* ---- CPU baseline driver (bench.py cpu_baseline leg): mimics the Go worker
 * (internal/worker/worker_impl.go:204-238): `threads` workers, static contiguous chunks of the pair
 * list, one find_alignment_with_fallback call per pair.  Sequences are bytes here (ASCII DNA).
 * Returns wall seconds; *cells = sum Q*T. */
double oracle_bench_pairs(const uint8_t *const *queries, const size_t *qlens,
                          const uint8_t *const *targets, const size_t *tlens, size_t n_pairs,
                          oracle_scores s, int threads, double *cells, int32_t *scores_out);

#ifdef __cplusplus
}
#endif
#endif
