/*
 * oracle/sw_clean.c -- TEST INFRASTRUCTURE ONLY (see sw_oracle.h).
 *
 * (1) The "clean" model: the semantics the HIP library is built to (DESIGN.md "Effective
 *     semantics"), written as a plain int32 Smith-Waterman so that tests can prove, on the CPU and at
 *     scale, that it equals the line-by-line restatement in sw_oracle.c on the i16-safe domain --
 *     including the reference's quirks.  Derivation (reference lines):
 *       - matrix: linear-gap SW, H(x,j)=max(0, H(x,j-1)+gap, H(x-1,j)+gap, H(x-1,j-1)+s)
 *         (rust/src/algorithm/mod.rs:108-117 in the skewed layout y = x + j).
 *       - best cell: maximal H, ties -> smallest y = x+j, then smallest x (first flat index of argmax,
 *         rust/src/lib.rs:122-123; ring path: strict '>' across rows + first-index row argmax,
 *         algorithm/mod.rs:358-362).
 *       - full-matrix path (T < 2Q, or the find_alignment_simd entry): column j = Tp is skipped
 *         (algorithm/mod.rs:98,178 compare with the padded length).  When T % 64 == 0 that is the last
 *         real target character; only H(1,T) is still computed, by the lane-parallel phase, and only
 *         if Tp >= Qp (algorithm/mod.rs:134).
 *       - ring path with T < Qp (Q <= 31, T <= 63): the first phase skips j >= T
 *         (algorithm/mod.rs:251), the lane phase is empty, so H(1,T) stays 0; cells with j < T and
 *         x+j <= Qp are "discovered" by phase 1 before every other cell (algorithm/mod.rs:249-298 then
 *         :385-442), which changes the tie order.
 *       - traceback priority diag, then (x-1,j), then (x,j-1) (algorithm/mod.rs:665-706).
 *       - no hit: empty strings, score 0, max_x = max_y = 2^64-1 (usize wrap of x-1 with x = 0).
 *       - straight entry: plain matrix, first flat index y*width+x -> smallest j then smallest x,
 *         score field 0, un-shifted 1-based coordinates (rust/src/lib.rs:55-70).
 * (2) The CPU baseline driver used by bench.py (worker-like threading over the restatement).
 */
#include "sw_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static size_t roundup64(size_t n) { return (n + 63) / 64 * 64; }

void oracle_clean_result_free(oracle_clean_result *r) {
  if (!r) return;
  free(r->q_aln);
  free(r->t_aln);
  memset(r, 0, sizeof(*r));
}

int32_t oracle_clean_score_plain(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                                 oracle_scores s) {
  int32_t *prev = (int32_t *)calloc(T + 1, sizeof(int32_t));
  int32_t *cur = (int32_t *)calloc(T + 1, sizeof(int32_t));
  int32_t best = 0;
  for (size_t x = 1; x <= Q; x++) {
    cur[0] = 0;
    for (size_t j = 1; j <= T; j++) {
      int32_t sub = q[x - 1] == t[j - 1] ? s.match : s.miss;
      int32_t h = prev[j - 1] + sub;
      int32_t a = cur[j - 1] + s.gap, b = prev[j] + s.gap;
      if (a > h) h = a;
      if (b > h) h = b;
      if (h < 0) h = 0;
      cur[j] = h;
      if (h > best) best = h;
    }
    int32_t *tmp = prev;
    prev = cur;
    cur = tmp;
  }
  free(prev);
  free(cur);
  return best;
}

int oracle_clean_align(const uint32_t *q, size_t Q, const uint32_t *t, size_t T, oracle_scores s,
                       int path, oracle_clean_result *out) {
  memset(out, 0, sizeof(*out));
  const size_t W = T + 1;
  int32_t *H = (int32_t *)calloc((Q + 1) * W, sizeof(int32_t));
  if (!H) return ORACLE_NOMEM;
#define HH(x, j) H[(size_t)(x) * W + (size_t)(j)]
  const size_t Qp = roundup64(Q), Tp = roundup64(T);
  int straight = path == ORACLE_PATH_STRAIGHT;
  int full = !straight && (path == ORACLE_PATH_SIMD || T < 2 * Q);
  int ring = !straight && !full;
  size_t Teff = T;
  int special = 0;      /* full path: H(1,T) is the only computed cell of column T */
  int zero_1T = 0;      /* ring path, T < Qp: H(1,T) stays 0 */
  int phase_order = 0;  /* ring path, T < Qp: discovery order has two phases */
  if (full && T > 0 && T % 64 == 0) {
    Teff = T - 1;
    special = (Tp >= Qp) && Q >= 1;
  }
  if (ring && T < Qp) {
    zero_1T = 1;
    phase_order = 1;
  }
  for (size_t x = 1; x <= Q; x++)
    for (size_t j = 1; j <= Teff; j++) {
      if (zero_1T && x == 1 && j == T) continue;
      int32_t sub = q[x - 1] == t[j - 1] ? s.match : s.miss;
      int32_t h = HH(x - 1, j - 1) + sub;
      int32_t a = HH(x, j - 1) + s.gap, b = HH(x - 1, j) + s.gap;
      if (a > h) h = a;
      if (b > h) h = b;
      if (h < 0) h = 0;
      HH(x, j) = h;
    }
  if (special) {
    int32_t sub = q[0] == t[T - 1] ? s.match : s.miss;
    int32_t h = sub, a = HH(1, T - 1) + s.gap, b = s.gap;
    if (a > h) h = a;
    if (b > h) h = b;
    if (h < 0) h = 0;
    HH(1, T) = h;
  }
  /* best cell */
  int32_t best = 0;
  size_t bx = 0, bj = 0;
  int bp = 0;
  for (size_t x = 1; x <= Q; x++)
    for (size_t j = 1; j <= T; j++) {
      int32_t h = HH(x, j);
      if (h <= 0) continue;
      int p = phase_order ? !((j < T) && (x + j <= Qp)) : 0;
      int better;
      if (h != best)
        better = h > best;
      else if (straight)
        better = (j < bj) || (j == bj && x < bx);
      else if (p != bp)
        better = p < bp;
      else
        better = (x + j < bx + bj) || (x + j == bx + bj && x < bx);
      if (better) {
        best = h;
        bx = x;
        bj = j;
        bp = p;
      }
    }
  /* traceback */
  size_t cap = Q + T + 1, n = 0;
  uint32_t *qa = (uint32_t *)malloc(cap * sizeof(uint32_t));
  uint32_t *ta = (uint32_t *)malloc(cap * sizeof(uint32_t));
  if (!qa || !ta) {
    free(qa);
    free(ta);
    free(H);
    return ORACLE_NOMEM;
  }
  size_t x = bx, j = bj;
  int rc = ORACLE_OK;
  while (x > 0 && j > 0) {
    int32_t h = HH(x, j);
    if (h == 0) break;
    int32_t sub = q[x - 1] == t[j - 1] ? s.match : s.miss;
    if (h == HH(x - 1, j - 1) + sub) {
      qa[n] = q[x - 1];
      ta[n] = t[j - 1];
      x--;
      j--;
    } else if (h == HH(x - 1, j) + s.gap) {
      qa[n] = q[x - 1];
      ta[n] = '-';
      x--;
    } else if (h == HH(x, j - 1) + s.gap) {
      qa[n] = '-';
      ta[n] = t[j - 1];
      j--;
    } else {
      rc = ORACLE_PANIC;
      break;
    }
    n++;
  }
  for (size_t a = 0, b = n; a + 1 < b; a++) {
    b--;
    uint32_t v = qa[a];
    qa[a] = qa[b];
    qa[b] = v;
    v = ta[a];
    ta[a] = ta[b];
    ta[b] = v;
  }
  out->q_aln = qa;
  out->t_aln = ta;
  out->aln_len = n;
  out->score32 = straight ? 0 : best;
  if (straight) {
    out->max_x = bx;
    out->max_y = bj;
  } else {
    out->max_x = (uint64_t)bx - 1; /* wraps to 2^64-1 when there is no hit */
    out->max_y = (uint64_t)bj - 1;
  }
  free(H);
#undef HH
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * CPU baseline driver -- internal/worker/worker_impl.go:204-238 (ExecuteWorkInParallel: static
 * contiguous chunks, one goroutine each) over :108-168 (one fallback-chain call per pair).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t *const *queries;
  const size_t *qlens;
  const uint8_t *const *targets;
  const size_t *tlens;
  size_t begin, end;
  oracle_scores s;
  int32_t *scores_out;
} bench_job;

static void *bench_worker(void *arg) {
  bench_job *jb = (bench_job *)arg;
  for (size_t i = jb->begin; i < jb->end; i++) {
    size_t Q = jb->qlens[i], T = jb->tlens[i];
    /* rust/src/bindings.rs:125-129: bytes -> String -> Vec<char> (4 bytes per char) on every call */
    uint32_t *q = (uint32_t *)malloc((Q + 1) * sizeof(uint32_t));
    uint32_t *t = (uint32_t *)malloc((T + 1) * sizeof(uint32_t));
    for (size_t k = 0; k < Q; k++) q[k] = jb->queries[i][k];
    for (size_t k = 0; k < T; k++) t[k] = jb->targets[i][k];
    oracle_result r;
    int which;
    int rc = oracle_find_alignment_with_fallback(q, Q, t, T, jb->s, &r, &which);
    if (jb->scores_out) jb->scores_out[i] = rc == ORACLE_OK ? r.score : -1;
    if (rc == ORACLE_OK) oracle_result_free(&r);
    free(q);
    free(t);
  }
  return NULL;
}

double oracle_bench_pairs(const uint8_t *const *queries, const size_t *qlens,
                          const uint8_t *const *targets, const size_t *tlens, size_t n_pairs,
                          oracle_scores s, int threads, double *cells, int32_t *scores_out) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n_pairs && n_pairs > 0) threads = (int)n_pairs;
  bench_job *jobs = (bench_job *)calloc((size_t)threads, sizeof(bench_job));
  pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
  double c = 0;
  for (size_t i = 0; i < n_pairs; i++) c += (double)qlens[i] * (double)tlens[i];
  if (cells) *cells = c;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < threads; i++) {
    jobs[i].queries = queries;
    jobs[i].qlens = qlens;
    jobs[i].targets = targets;
    jobs[i].tlens = tlens;
    jobs[i].begin = n_pairs * (size_t)i / (size_t)threads; /* worker_impl.go:218-219 */
    jobs[i].end = n_pairs * (size_t)(i + 1) / (size_t)threads;
    jobs[i].s = s;
    jobs[i].scores_out = scores_out;
    pthread_create(&th[i], NULL, bench_worker, &jobs[i]);
  }
  for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(jobs);
  free(th);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
