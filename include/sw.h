/*
 * include/sw.h -- the drop-in C ABI of libsw_hip.so (MI355X / gfx950 Smith-Waterman engine).
 *
 * This header declares exactly the symbols the DLSA Go worker binds through cgo.  The reference
 * declares them in a hand-written header, rust/header/sw.h:1-28, included from
 * internal/worker/worker_rust_bridge.go:4 and internal/smithwaterman/smithwaterman.go:5, and defines
 * them in Rust (rust/src/bindings.rs).  A maintainer switches the worker to this library by changing
 * the two `#cgo LDFLAGS` lines (worker_rust_bridge.go:3, smithwaterman.go:4) -- see INTEGRATION.md.
 *
 * Layout contract (x86-64 SysV, identical to rust/src/bindings.rs:9-16 and algorithm/mod.rs:18-24):
 *   struct Result           size 40: query_ptr@0 target_ptr@8 score@16 max_x@24 max_y@32
 *   struct AlignmentScores  size 6, passed BY VALUE, field order gap, match, miss.
 * The C header types the 16-bit fields unsigned while Rust uses i16; the bits are what matter.  The Go
 * worker stores C.ushort(-penalty) (worker_rust_bridge.go:108-112), i.e. gap and miss arrive as
 * two's-complement NEGATIVE numbers and match as a positive one (worker_impl.go:126-127).
 *
 * Semantics of every entry point: identical results to the reference on the same input -- aligned
 * strings, score, max_x/max_y (0-based index of the last aligned query/target character), the
 * reference's tie-breaking, its traceback priority, its no-hit convention (empty strings, score 0,
 * max_x = max_y = 2^64-1) and its padded-length quirks (DESIGN.md "Effective semantics").  Inputs are
 * NUL-terminated UTF-8, borrowed for the call; invalid UTF-8 or any internal failure returns NULL
 * (bindings.rs:109-112), never aborts.  All entry points are thread-safe (the worker enters from
 * NumCPU-1 OS threads, worker_impl.go:204-238).  There is no CPU fallback: without a usable GPU every
 * find_* call returns NULL and sw_hip_last_error() (include/sw_hip.h) says why.
 */
#ifndef DLSA_AMD_SW_H
#define DLSA_AMD_SW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rust/header/sw.h:4-11 == rust/src/bindings.rs:9-16 (AlignmentResult) */
struct Result {
  char *query_ptr;  /* aligned query, '-' for a gap; owned by the library until freed */
  char *target_ptr; /* aligned target */
  uint16_t score;   /* low 16 bits of the score (i16 in Rust); see sw_result_score32() in sw_hip.h */
  uint64_t max_x;   /* index of the last aligned query char, 2^64-1 when nothing aligns */
  uint64_t max_y;   /* same for the target */
};

/* rust/header/sw.h:13-18 == rust/src/algorithm/mod.rs:18-24 */
struct AlignmentScores {
  uint16_t gap;   /* e.g. (uint16_t)-1 */
  uint16_t match; /* e.g. 2 */
  uint16_t miss;  /* e.g. (uint16_t)-1 */
};

/* replaces rust/src/bindings.rs:116-157 (called at worker_rust_bridge.go:114): the worker's first
 * choice.  Dispatches like rust/src/algorithm/mod.rs:213: full-matrix semantics when
 * strlen(target) < 2*strlen(query) (in chars), ring-buffer semantics otherwise. */
struct Result *find_alignment_low_memory(char *query, char *target, struct AlignmentScores alignmentScores);

/* replaces rust/src/bindings.rs:72-113 (called at worker_rust_bridge.go:93): always the full-matrix
 * semantics of rust/src/lib.rs:105-136. */
struct Result *find_alignment_simd(char *query, char *target, struct AlignmentScores alignmentScores);

/* replaces rust/src/bindings.rs:188-229 (called at worker_rust_bridge.go:73): un-skewed matrix, ties to
 * the smallest target index then the smallest query index, score field 0 and 1-based un-shifted
 * coordinates exactly as rust/src/lib.rs:45-71 returns them. */
struct Result *find_alignment_sequential_straight(char *query, char *target, struct AlignmentScores alignmentScores);

/* Declared by rust/header/sw.h:20-21 but NOT defined by the reference (commented out at
 * rust/src/bindings.rs:30-69,159-185; the Go calls are commented out at worker_rust_bridge.go:50-61).
 * Exported for header completeness; both return NULL. */
struct Result *find_alignment_parallel(char *query, char *target, size_t threads, struct AlignmentScores alignmentScore);
struct Result *find_alignment_sequential(char *query, char *target, struct AlignmentScores alignmentScore);

/* replaces rust/src/bindings.rs:232-240 (called at worker_rust_bridge.go:143).  NULL is accepted.  The
 * reference frees the struct only and leaks both strings (TODO at worker_rust_bridge.go:141); this
 * library frees the strings too unless they were already handed to free_c_string(). */
void free_alignment_result(struct Result *result);

/* replaces rust/src/bindings.rs:242-249 (wrapped at worker_rust_bridge.go:147-149, never invoked).  NULL
 * is accepted.  May be called on either string of a Result before free_alignment_result().  Only pointers this
 * library returned may be passed to the two free functions (strings of a Result or of a sw_hit); a pointer that
 * does not carry the library's marker is left untouched. */
void free_c_string(char *str);

#ifdef __cplusplus
}
#endif
#endif /* DLSA_AMD_SW_H */
