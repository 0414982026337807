/*
 * tests/c_harness/worker_harness.c -- a plain-C stand-in for the Go worker's use of the drop-in ABI.
 *
 * Go is not available where this repository is built, so this program exercises the cgo boundary the way
 * internal/worker does, from C: it includes the three public headers as C (not C++), links libsw_hip.so with the
 * link line INTEGRATION.md gives for cgo, and
 *   selftest          -- no GPU needed: struct layouts (rust/header/sw.h:4-18), symbol presence, host-only calls;
 *   pairs FILE N      -- ExecuteWorkInParallel (worker_impl.go:204-238): N threads, static contiguous chunks of the
 *                        pair list, one findAlignmentWithFallback (worker_impl.go:170-202) per pair, C.GoString-style
 *                        copy-out (worker_rust_bridge.go:131-139), free_alignment_result once per result (:121);
 *                        prints one line per pair, in input order:
 *                        entry <TAB> score <TAB> int(max_x) <TAB> int(max_y) <TAB> len(qRes) <TAB> qRes <TAB> tRes
 * FILE holds one pair per line: gap match miss query target (whitespace separated; "-" stands for an empty string).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sw.h"
#include "sw_hip.h"
#include "sw_worker.h"

struct pair {
  int gap, match, miss;
  char *q, *t;
  /* outputs */
  int entry;
  unsigned score;
  long long max_x, max_y;
  char *qa, *ta;
};

static struct pair *g_pairs;
static size_t g_n;
static int g_threads;

/* findAlignmentWithFallback, worker_impl.go:170-202 */
static struct Result *align_with_fallback(struct pair *p, int *entry) {
  struct AlignmentScores s;
  struct Result *r;
  s.gap = (uint16_t)p->gap; /* C.ushort(alignmentScore.GapPenalty), worker_rust_bridge.go:108-112 */
  s.match = (uint16_t)p->match;
  s.miss = (uint16_t)p->miss;
  *entry = 0;
  if ((r = find_alignment_low_memory(p->q, p->t, s))) return r;
  *entry = 1;
  if ((r = find_alignment_simd(p->q, p->t, s))) return r;
  *entry = 2;
  if ((r = find_alignment_sequential_straight(p->q, p->t, s))) return r;
  *entry = -1;
  return NULL;
}

static char *go_string(const char *s) { /* C.GoString: a copy the caller owns */
  size_t n = strlen(s);
  char *c = (char *)malloc(n + 1);
  memcpy(c, s, n + 1);
  return c;
}

static void *chunk_main(void *arg) {
  size_t i = (size_t)arg;
  size_t start = g_n * i / (size_t)g_threads, end = g_n * (i + 1) / (size_t)g_threads; /* worker_impl.go:217-221 */
  for (size_t k = start; k < end; k++) {
    struct pair *p = &g_pairs[k];
    struct Result *r = align_with_fallback(p, &p->entry);
    if (!r) {
      p->qa = go_string("");
      p->ta = go_string("");
      continue;
    }
    p->qa = go_string(r->query_ptr);
    p->ta = go_string(r->target_ptr);
    p->score = r->score;
    p->max_x = (long long)r->max_x; /* int(rustRes.MaxX), worker_impl.go:139 */
    p->max_y = (long long)r->max_y;
    free_alignment_result(r);
  }
  return NULL;
}

static int selftest(void) {
  uint32_t pp[4];
  int bad = 0;
  bad |= sizeof(struct Result) != 40 || offsetof(struct Result, query_ptr) != 0 || offsetof(struct Result, target_ptr) != 8 ||
         offsetof(struct Result, score) != 16 || offsetof(struct Result, max_x) != 24 || offsetof(struct Result, max_y) != 32;
  bad |= sizeof(struct AlignmentScores) != 6 || offsetof(struct AlignmentScores, gap) != 0 ||
         offsetof(struct AlignmentScores, match) != 2 || offsetof(struct AlignmentScores, miss) != 4;
  if (bad) {
    printf("layout mismatch\n");
    return 1;
  }
  if (!sw_hip_version() || !*sw_hip_version()) return 2;
  free_alignment_result(NULL); /* must be tolerated */
  free_c_string(NULL);
  if (sw_hip_path_params(SW_PATH_LOW_MEMORY, 100, 128, pp) != SW_OK || pp[0] != 127 || pp[1] != 128) return 3;
  if (sw_hip_path_params(SW_PATH_LOW_MEMORY, 20, 50, pp) != SW_OK || pp[1] != 49 || pp[2] != 64) return 4;
  {
    void (*syms[])(void) = {(void (*)(void))find_alignment_low_memory, (void (*)(void))find_alignment_simd,
                            (void (*)(void))find_alignment_sequential_straight, (void (*)(void))find_alignment_parallel,
                            (void (*)(void))find_alignment_sequential, (void (*)(void))sw_align_batch,
                            (void (*)(void))sw_db_search_topk_batch, (void (*)(void))sw_execute_work,
                            (void (*)(void))sw_benchmark_rust, (void (*)(void))sw_fasta_parse};
    for (size_t i = 0; i < sizeof syms / sizeof syms[0]; i++)
      if (!syms[i]) return 5;
  }
  printf("selftest ok: %s\n", sw_hip_version());
  return 0;
}

static char *dup_field(const char *s) { return go_string(strcmp(s, "-") == 0 ? "" : s); }

int main(int argc, char **argv) {
  if (argc >= 2 && strcmp(argv[1], "selftest") == 0) return selftest();
  if (argc < 4 || strcmp(argv[1], "pairs") != 0) {
    fprintf(stderr, "usage: %s selftest | pairs FILE THREADS\n", argv[0]);
    return 64;
  }
  FILE *fp = fopen(argv[2], "r");
  if (!fp) return 65;
  g_threads = atoi(argv[3]);
  if (g_threads < 1) g_threads = 1;
  size_t cap = 0;
  static char qbuf[1 << 20], tbuf[1 << 20];
  int gap, match, miss;
  while (fscanf(fp, "%d %d %d %1048575s %1048575s", &gap, &match, &miss, qbuf, tbuf) == 5) {
    if (g_n == cap) {
      cap = cap ? cap * 2 : 64;
      g_pairs = (struct pair *)realloc(g_pairs, cap * sizeof *g_pairs);
    }
    memset(&g_pairs[g_n], 0, sizeof *g_pairs);
    g_pairs[g_n].gap = gap;
    g_pairs[g_n].match = match;
    g_pairs[g_n].miss = miss;
    g_pairs[g_n].q = dup_field(qbuf);
    g_pairs[g_n].t = dup_field(tbuf);
    g_n++;
  }
  fclose(fp);
  sw_hip_init(0); /* device and code objects up before the clock starts, like the worker's benchmark run */
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_t *th = (pthread_t *)malloc((size_t)g_threads * sizeof *th);
  for (int i = 0; i < g_threads; i++) pthread_create(&th[i], NULL, chunk_main, (void *)(size_t)i);
  for (int i = 0; i < g_threads; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  {
    double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    fprintf(stderr, "%zu pairs, %d threads: %.3f s, %.0f pairs/s\n", g_n, g_threads, secs, (double)g_n / secs);
  }
  for (size_t k = 0; k < g_n; k++) {
    struct pair *p = &g_pairs[k];
    printf("%d\t%u\t%lld\t%lld\t%zu\t%s\t%s\n", p->entry, p->score, p->max_x, p->max_y, strlen(p->qa), p->qa, p->ta);
  }
  return 0;
}
