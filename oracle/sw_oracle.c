/*
 * oracle/sw_oracle.c -- TEST INFRASTRUCTURE ONLY (see sw_oracle.h).
 *
 * Line-by-line CPU restatement of the reference's Smith-Waterman path.  Every function cites the
 * reference lines it follows.  All score arithmetic is i16 with two's-complement wrap-around, as
 * in the reference's release build (no overflow checks; std::simd adds always wrap).  usize
 * arithmetic wraps too.  A Rust panic (slice index out of bounds, division by zero, assert!,
 * explicit panic!) is restated as `return ORACLE_PANIC`.
 *
 * The row loops of the "SIMD" phases are written as plain loops over the 64-lane chunks; each output
 * row depends only on the two previous rows, so the result is identical to the lane-parallel code
 * and gcc -O3 vectorises them, which gives the CPU baseline the reference's cost profile.
 */
#include "sw_oracle.h"

#include <stdlib.h>
#include <string.h>

#define LANES 64 /* rust/src/bindings.rs:7 */

typedef int16_t i16;
typedef uint16_t u16;

static inline i16 wadd(i16 a, i16 b) { return (i16)(u16)((u16)a + (u16)b); }
static inline i16 max16(i16 a, i16 b) { return a > b ? a : b; }

/* rust/src/lib.rs:149-162 */
static size_t roundup(size_t number, size_t multiple) {
  if (multiple == 0) return number;
  size_t rem = number % multiple;
  if (rem == 0) return number;
  return number + (multiple - rem);
}
/* rust/src/lib.rs:139-147 */
#define INDEX(x, y, width) ((size_t)(y) * (size_t)(width) + (size_t)(x))

/* the recurrence, rust/src/algorithm/mod.rs:108-117 (identical at :50-59, :188-197, :261-270 ...) */
static inline i16 cell(i16 left_up_same_x /* (x, y-1) */, i16 prev_x /* (x-1, y-1) */,
                       i16 diag /* (x-1, y-2) */, i16 sub, i16 gap) {
  return max16(max16(wadd(left_up_same_x, gap), wadd(prev_x, gap)), max16(wadd(diag, sub), 0));
}

/* argminmax 0.6.1 `ArgMinMax::argmax` on an i16 slice: index of the FIRST maximal element
 * (rust/Cargo.lock:27-33; call sites rust/src/lib.rs:28,55,88,122, algorithm/mod.rs:273,361,414).
 * The crate's source is not under /root/reference; "first" is its documented behaviour and is what
 * known-answer vector 9 ("AA" vs "ATTA" -> "A"/"A") requires on the ring path. */
static size_t argmax16(const i16 *d, size_t n) {
  size_t best = 0;
  i16 bv = d[0];
  for (size_t i = 1; i < n; i++)
    if (d[i] > bv) {
      bv = d[i];
      best = i;
    }
  return best;
}

/* growable (q,t) op list written backwards: the reference's recursion pushes after the recursive
 * call returns (rust/src/algorithm/mod.rs:666-679), so the first cell visited is the last element. */
typedef struct {
  uint32_t *q, *t;
  size_t n, cap;
} revbuf;
static int rb_push(revbuf *b, uint32_t qc, uint32_t tc) {
  if (b->n == b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 64;
    uint32_t *nq = (uint32_t *)realloc(b->q, nc * sizeof(uint32_t));
    if (!nq) return ORACLE_NOMEM;
    b->q = nq;
    uint32_t *nt = (uint32_t *)realloc(b->t, nc * sizeof(uint32_t));
    if (!nt) return ORACLE_NOMEM;
    b->t = nt;
    b->cap = nc;
  }
  b->q[b->n] = qc;
  b->t[b->n] = tc;
  b->n++;
  return ORACLE_OK;
}
static void rb_reverse(revbuf *b) {
  for (size_t i = 0, j = b->n; i + 1 < j; i++) {
    j--;
    uint32_t a = b->q[i];
    b->q[i] = b->q[j];
    b->q[j] = a;
    a = b->t[i];
    b->t[i] = b->t[j];
    b->t[j] = a;
  }
}
static void rb_free(revbuf *b) {
  free(b->q);
  free(b->t);
  b->q = b->t = NULL;
  b->n = b->cap = 0;
}

/* rust/src/algorithm/mod.rs:637-714 `traceback` and :556-635 `traceback_wrapping` (store_height != 0:
 * rows are addressed modulo data.len()/width).  The recursion is unrolled into a loop; pushes happen
 * in reverse and the buffer is reversed by the caller. */
static int traceback_skewed(const i16 *data, size_t data_len, const uint32_t *query, size_t Q,
                            const uint32_t *target, size_t T, size_t x, size_t y, size_t width,
                            int wrapping, oracle_scores sc, revbuf *out) {
  size_t h = wrapping ? data_len / width : 0;
  for (;;) {
    if (x == 0 || y == 0 || x == y) return ORACLE_OK;
    size_t ti = y - x - 1; /* wraps when y < x + 1, then the index below is out of bounds */
    if (x - 1 >= Q || ti >= T) return ORACLE_PANIC;
    i16 match_score = (query[x - 1] == target[ti]) ? sc.match : sc.miss;
    size_t r0 = wrapping ? y % h : y;
    size_t i0 = INDEX(x, r0, width);
    if (i0 >= data_len) return ORACLE_PANIC;
    i16 score = data[i0];
    if (score == 0) return ORACLE_OK;
    /* y >= 2 here: x >= 1, x != y and ti did not wrap imply y > x >= 1 */
    size_t r1 = wrapping ? (y - 1) % h : (y - 1);
    size_t r2 = wrapping ? (y - 2) % h : (y - 2);
    size_t id = INDEX(x - 1, r2, width), iu = INDEX(x - 1, r1, width), il = INDEX(x, r1, width);
    if (id >= data_len) return ORACLE_PANIC;
    if (score == wadd(data[id], match_score)) {
      int rc = rb_push(out, query[x - 1], target[ti]);
      if (rc) return rc;
      x -= 1;
      y -= 2;
      continue;
    }
    if (iu >= data_len) return ORACLE_PANIC;
    if (score == wadd(data[iu], sc.gap)) {
      int rc = rb_push(out, query[x - 1], '-');
      if (rc) return rc;
      x -= 1;
      y -= 1;
      continue;
    }
    if (il >= data_len) return ORACLE_PANIC;
    if (score == wadd(data[il], sc.gap)) {
      int rc = rb_push(out, '-', target[ti]);
      if (rc) return rc;
      y -= 1;
      continue;
    }
    return ORACLE_PANIC; /* "Mismatch in data" */
  }
}

static void result_from(revbuf *b, oracle_result *out, i16 score, uint64_t mx, uint64_t my) {
  rb_reverse(b);
  out->q_aln = b->q;
  out->t_aln = b->t;
  out->aln_len = b->n;
  out->score = score;
  out->max_x = mx;
  out->max_y = my;
  b->q = b->t = NULL;
}

void oracle_result_free(oracle_result *r) {
  if (!r) return;
  free(r->q_aln);
  free(r->t_aln);
  memset(r, 0, sizeof(*r));
}

/* ------------------------------------------------------------------------------------------------
 * rust/src/algorithm/mod.rs:66-202 string_scores_simd::<64>
 * ---------------------------------------------------------------------------------------------- */
static int string_scores_simd(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                              oracle_scores sc, i16 **data_out, size_t *len_out, size_t *width_out) {
  size_t Qp = roundup(Q, LANES), Tp = roundup(T, LANES);
  /* :75-90  query -> char as u16 + 2, padded with 0; target -> char as u16 + 2, padded with 1 */
  u16 *query = (u16 *)malloc((Qp + 1) * sizeof(u16));
  u16 *target = (u16 *)malloc((Tp + 1) * sizeof(u16));
  u16 *target_rev = (u16 *)malloc((Tp + 1) * sizeof(u16));
  if (!query || !target || !target_rev) {
    free(query);
    free(target);
    free(target_rev);
    return ORACLE_NOMEM;
  }
  for (size_t i = 0; i < Qp; i++) query[i] = i < Q ? (u16)((u16)q[i] + 2) : 0;
  for (size_t i = 0; i < Tp; i++) target[i] = i < T ? (u16)((u16)t[i] + 2) : 1;
  size_t width = Qp + 1;            /* :92 */
  size_t data_height = Qp + Tp + 1; /* :93 */
  size_t len = width * (width + data_height); /* :94 */
  i16 *data = (i16 *)calloc(len, sizeof(i16));
  if (!data) {
    free(query);
    free(target);
    free(target_rev);
    return ORACLE_NOMEM;
  }
  const i16 gap = sc.gap, mat = sc.match, mis = sc.miss;

  /* :96-119 scalar upper triangle */
  for (size_t y = 2; y < width; y++) {
    for (size_t x = 1; x < y; x++) {
      if (y - x >= Tp) continue; /* :98, compares with the PADDED target length */
      i16 sub = query[x - 1] == target[y - x - 1] ? mat : mis;
      data[INDEX(x, y, width)] = cell(data[INDEX(x, y - 1, width)], data[INDEX(x - 1, y - 1, width)],
                                      data[INDEX(x - 1, y - 2, width)], sub, gap);
    }
  }
  /* :126-127 reversed target copy */
  for (size_t i = 0; i < Tp; i++) target_rev[i] = target[Tp - 1 - i];
  /* :134-171 lane-parallel rows y in width..=Tp+1, every x in 1..=Qp */
  for (size_t y = width; y <= Tp + 1; y++) {
    i16 *row0 = data + INDEX(1, y, width);
    const i16 *row1 = row0 - width;             /* (x, y-1)   */
    const i16 *row1m = row0 - width - 1;        /* (x-1, y-1) */
    const i16 *row2m = row0 - 2 * width - 1;    /* (x-1, y-2) */
    const u16 *tr = target_rev + (Tp - (y - 1)); /* :137 */
    for (size_t k = 0; k < Qp; k++) {
      i16 sub = query[k] == tr[k] ? mat : mis;
      i16 a = wadd(row1[k], gap), b = wadd(row1m[k], gap), c = wadd(row2m[k], sub);
      i16 m = a > b ? a : b;
      m = m > c ? m : c;
      row0[k] = m > 0 ? m : 0;
    }
  }
  /* :173-199 scalar lower triangle */
  for (size_t y = data_height - width; y < data_height; y++) {
    for (size_t x = 1; x < width; x++) {
      if (x + 1 > y) continue;
      if (y - x >= Tp) continue; /* :178 */
      i16 sub = query[x - 1] == target[y - x - 1] ? mat : mis;
      data[INDEX(x, y, width)] = cell(data[INDEX(x, y - 1, width)], data[INDEX(x - 1, y - 1, width)],
                                      data[INDEX(x - 1, y - 2, width)], sub, gap);
    }
  }
  free(query);
  free(target);
  free(target_rev);
  *data_out = data;
  *len_out = len;
  *width_out = width;
  return ORACLE_OK;
}

/* rust/src/lib.rs:105-136 */
int oracle_find_alignment_simd(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                               oracle_scores sc, oracle_result *out) {
  memset(out, 0, sizeof(*out));
  i16 *data;
  size_t len, width;
  int rc = string_scores_simd(q, Q, t, T, sc, &data, &len, &width);
  if (rc) return rc;
  size_t max_index = argmax16(data, len); /* :122 */
  size_t x = max_index % width, y = max_index / width; /* :123 coord */
  revbuf rb = {0};
  rc = traceback_skewed(data, len, q, Q, t, T, x, y, width, 0, sc, &rb);
  if (rc) {
    rb_free(&rb);
    free(data);
    return rc;
  }
  /* :135  (.., data[max_index], x - 1, y - x - 1) with wrapping usize */
  result_from(&rb, out, data[max_index], (uint64_t)x - 1, (uint64_t)y - (uint64_t)x - 1);
  free(data);
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------------
 * rust/src/algorithm/mod.rs:204-445 find_alignment_simd_lowmem::<64>
 * ---------------------------------------------------------------------------------------------- */
int oracle_find_alignment_simd_lowmem(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                                      oracle_scores sc, oracle_result *out) {
  /* :213-215 */
  if (T < 2 * Q) return oracle_find_alignment_simd(q, Q, t, T, sc, out);
  memset(out, 0, sizeof(*out));

  size_t Qp = roundup(Q, LANES); /* :221 */
  size_t width = Qp + 1;         /* :232 */
  size_t data_height = T + Q + 1; /* :234 */
  /* :236-238 unsigned_abs of i16, truncating division; gap == 0 panics (division by zero) */
  size_t abs_match = (size_t)(u16)(sc.match < 0 ? -(int)sc.match : sc.match);
  size_t abs_gap = (size_t)(u16)(sc.gap < 0 ? -(int)sc.gap : sc.gap);
  if (abs_gap == 0) return ORACLE_PANIC;
  size_t wrapping_height = Qp + (Qp * abs_match) / abs_gap;
  size_t h = wrapping_height + width; /* :240 data_store_height */
  size_t len = width * h;             /* :242 */

  u16 *query = (u16 *)malloc((Qp + 1) * sizeof(u16));
  u16 *target = (u16 *)malloc((T + 1) * sizeof(u16));
  u16 *target_rev = (u16 *)malloc((T + 1) * sizeof(u16));
  i16 *data = (i16 *)calloc(len ? len : 1, sizeof(i16));
  revbuf rb = {0};
  int rc = ORACLE_OK;
  if (!query || !target || !target_rev || !data) {
    rc = ORACLE_NOMEM;
    goto done;
  }
  /* :222-230 query -> char as u16 + 1 padded with 0, target -> char as u16 + 1, NOT padded */
  for (size_t i = 0; i < Qp; i++) query[i] = i < Q ? (u16)((u16)q[i] + 1) : 0;
  for (size_t i = 0; i < T; i++) target[i] = (u16)((u16)t[i] + 1);
  const i16 gap = sc.gap, mat = sc.match, mis = sc.miss;
  i16 current_max = 0;
  size_t current_max_x = 0, current_max_y = 0;

#define RUN_TRACEBACK()                                                                       \
  do {                                                                                        \
    rb.n = 0; /* :284-285 fresh result vectors */                                             \
    rc = traceback_skewed(data, len, q, Q, t, T, current_max_x, current_max_y, width, 1, sc,  \
                          &rb);                                                               \
    if (rc) goto done;                                                                        \
  } while (0)

  /* :249-298 scalar upper triangle, rows not wrapped */
  for (size_t y = 2; y < width; y++) {
    for (size_t x = 1; x < y; x++) {
      if (y - x >= T) continue; /* :251, UNPADDED target length */
      i16 sub = query[x - 1] == target[y - x - 1] ? mat : mis;
      size_t i0 = INDEX(x, y, width);
      if (i0 >= len) {
        rc = ORACLE_PANIC;
        goto done;
      }
      data[i0] = cell(data[INDEX(x, y - 1, width)], data[INDEX(x - 1, y - 1, width)],
                      data[INDEX(x - 1, y - 2, width)], sub, gap);
    }
    size_t left = INDEX(1, y, width); /* :272 */
    if (left + width - 1 > len) {
      rc = ORACLE_PANIC;
      goto done;
    }
    size_t row_max_index = argmax16(data + left, width - 1);
    i16 row_max = data[left + row_max_index];
    if (row_max > current_max) { /* :277 strict */
      current_max = row_max;
      current_max_x = (left + row_max_index) % width;
      current_max_y = y;
      RUN_TRACEBACK();
    }
  }

  for (size_t i = 0; i < T; i++) target_rev[i] = target[T - 1 - i]; /* :305-306 */

  /* :313-381 lane-parallel rows y in width..T+2, every x in 1..=Qp, rows modulo h */
  for (size_t y = width; y < T + 2; y++) {
    i16 *row0 = data + INDEX(1, y % h, width);
    const i16 *row1 = data + INDEX(1, (y - 1) % h, width);
    const i16 *row2 = data + INDEX(1, (y - 2) % h, width);
    const u16 *tr = target_rev + (T - (y - 1)); /* :318; slice [..+LANES] stays inside target_rev */
    i16 row_max = 0;                            /* :319 */
    for (size_t k = 0; k < Qp; k++) {
      i16 sub = query[k] == tr[k] ? mat : mis;
      i16 a = wadd(row1[k], gap), b = wadd(row1[(ptrdiff_t)k - 1], gap),
          c = wadd(row2[(ptrdiff_t)k - 1], sub);
      i16 m = a > b ? a : b;
      m = m > c ? m : c;
      m = m > 0 ? m : 0;
      row0[k] = m;
      row_max = m > row_max ? m : row_max; /* :355 */
    }
    if (row_max > current_max) { /* :358 */
      current_max = row_max;
      size_t left = INDEX(1, y % h, width);
      size_t am = argmax16(data + left, width - 1); /* :361 */
      current_max_x = (left + am) % width;
      current_max_y = y;
      RUN_TRACEBACK();
    }
  }

  /* :384-442 scalar lower triangle, x in start_x..=Q, rows modulo h */
  {
    size_t start_x = 2;
    for (size_t y = T + 2; y < data_height; y++) {
      for (size_t x = start_x; x < Q + 1; x++) {
        if (!(y - x - 1 <= T)) { /* :387 assert! */
          rc = ORACLE_PANIC;
          goto done;
        }
        if (x + 1 > y) continue;
        if (y - x - 1 >= T) { /* slice index target_u16[y-x-1] */
          rc = ORACLE_PANIC;
          goto done;
        }
        i16 sub = query[x - 1] == target[y - x - 1] ? mat : mis;
        data[INDEX(x, y % h, width)] =
            cell(data[INDEX(x, (y - 1) % h, width)], data[INDEX(x - 1, (y - 1) % h, width)],
                 data[INDEX(x - 1, (y - 2) % h, width)], sub, gap);
      }
      size_t left = INDEX(start_x, y % h, width);
      size_t right = INDEX(Q + 1, y % h, width);
      if (left == right) { /* :413 assert_ne! */
        rc = ORACLE_PANIC;
        goto done;
      }
      if (left > right) { /* slice with start > end panics */
        rc = ORACLE_PANIC;
        goto done;
      }
      size_t am = argmax16(data + left, right - left);
      i16 row_max = data[am + left];
      if (row_max > current_max) { /* :416 */
        current_max = row_max;
        current_max_x = (left + am) % width;
        current_max_y = y;
        if (!(current_max_x >= start_x)) { /* :422 assert! */
          rc = ORACLE_PANIC;
          goto done;
        }
        RUN_TRACEBACK();
      }
      start_x += 1;
    }
  }
#undef RUN_TRACEBACK

  /* :444 */
  result_from(&rb, out, current_max, (uint64_t)current_max_x - 1,
              (uint64_t)current_max_y - (uint64_t)current_max_x - 1);
done:
  rb_free(&rb);
  free(query);
  free(target);
  free(target_rev);
  free(data);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * rust/src/algorithm/mod.rs:518-549 string_scores_straight, :716-785 traceback_straight,
 * rust/src/lib.rs:45-71 find_alignment_sequential_straight
 * ---------------------------------------------------------------------------------------------- */
int oracle_find_alignment_sequential_straight(const uint32_t *q, size_t Q, const uint32_t *t,
                                              size_t T, oracle_scores sc, oracle_result *out) {
  memset(out, 0, sizeof(*out));
  size_t width = Q + 1, height = T + 1;
  size_t len = width * height;
  i16 *data = (i16 *)calloc(len, sizeof(i16));
  if (!data) return ORACLE_NOMEM;
  for (size_t y = 1; y < height; y++)
    for (size_t x = 1; x < width; x++) {
      i16 sub = q[x - 1] == t[y - 1] ? sc.match : sc.miss;
      data[INDEX(x, y, width)] =
          max16(max16(wadd(data[INDEX(x, y - 1, width)], sc.gap), wadd(data[INDEX(x - 1, y, width)], sc.gap)),
                max16(wadd(data[INDEX(x - 1, y - 1, width)], sub), 0));
    }
  size_t max_index = argmax16(data, len); /* lib.rs:55 */
  size_t x = max_index % width, y = max_index / width;
  size_t mx = x, my = y;
  revbuf rb = {0};
  int rc = ORACLE_OK;
  for (;;) { /* traceback_straight */
    if (x == 0 || y == 0) break;
    i16 ms = q[x - 1] == t[y - 1] ? sc.match : sc.miss;
    i16 score = data[INDEX(x, y, width)];
    if (score == 0) break;
    if (score == wadd(data[INDEX(x - 1, y - 1, width)], ms)) {
      rc = rb_push(&rb, q[x - 1], t[y - 1]);
      x--;
      y--;
    } else if (score == wadd(data[INDEX(x - 1, y, width)], sc.gap)) {
      rc = rb_push(&rb, q[x - 1], '-');
      x--;
    } else if (score == wadd(data[INDEX(x, y - 1, width)], sc.gap)) {
      rc = rb_push(&rb, '-', t[y - 1]);
      y--;
    } else {
      rc = ORACLE_PANIC; /* "Mismatch in data" */
    }
    if (rc) break;
  }
  free(data);
  if (rc) {
    rb_free(&rb);
    return rc;
  }
  result_from(&rb, out, 0, (uint64_t)mx, (uint64_t)my); /* lib.rs:70: score 0, un-shifted x, y */
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------------
 * rust/src/algorithm/mod.rs:27-64 string_scores_sequential, rust/src/lib.rs:17-43
 * ---------------------------------------------------------------------------------------------- */
int oracle_find_alignment_sequential(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                                     oracle_scores sc, oracle_result *out) {
  memset(out, 0, sizeof(*out));
  size_t width = Q + 1, height = Q + T + 1;
  size_t len = width * (width + height);
  i16 *data = (i16 *)calloc(len, sizeof(i16));
  if (!data) return ORACLE_NOMEM;
  for (size_t y = 1; y < height; y++)
    for (size_t x = 1; x < width; x++) {
      if (x >= y) continue;
      if (y > T + x) continue;
      i16 sub = q[x - 1] == t[y - x - 1] ? sc.match : sc.miss;
      data[INDEX(x, y, width)] = cell(data[INDEX(x, y - 1, width)], data[INDEX(x - 1, y - 1, width)],
                                      data[INDEX(x - 1, y - 2, width)], sub, sc.gap);
    }
  size_t max_index = argmax16(data, len);
  size_t x = max_index % width, y = max_index / width;
  revbuf rb = {0};
  int rc = traceback_skewed(data, len, q, Q, t, T, x, y, width, 0, sc, &rb);
  free(data);
  if (rc) {
    rb_free(&rb);
    return rc;
  }
  result_from(&rb, out, 0, (uint64_t)x - 1, (uint64_t)y - (uint64_t)x - 1); /* lib.rs:42 */
  return ORACLE_OK;
}

/* internal/worker/worker_impl.go:170-202 */
int oracle_find_alignment_with_fallback(const uint32_t *q, size_t Q, const uint32_t *t, size_t T,
                                        oracle_scores s, oracle_result *out, int *which) {
  int rc = oracle_find_alignment_simd_lowmem(q, Q, t, T, s, out);
  if (rc == ORACLE_OK) {
    if (which) *which = 0;
    return rc;
  }
  rc = oracle_find_alignment_simd(q, Q, t, T, s, out);
  if (rc == ORACLE_OK) {
    if (which) *which = 1;
    return rc;
  }
  rc = oracle_find_alignment_sequential_straight(q, Q, t, T, s, out);
  if (rc == ORACLE_OK) {
    if (which) *which = 2;
    return rc;
  }
  if (which) *which = -1;
  return rc;
}
