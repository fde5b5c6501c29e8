/* l2_lru_sim.c -- LRU model of the eight per-XCD L2s under a task schedule (helper of tools/l2_hit_simulation.py).
 *
 * One cache line per gathered X row (a 32-column fp32 panel is 128 bytes).  Every XCD works through its own list of
 * workgroups in order, `conc` of them resident at a time; each resident workgroup's tasks advance round-robin,
 * `batch` entries per turn (the kernel's 8 row loads in flight per lane group).  Build: gcc -O2 -shared -fPIC.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int32_t *prev, *next, *in;  /* intrusive LRU list over line ids; in[id] = 1 when cached */
  int32_t head, tail, count, cap;
} lru_t;

static void lru_init(lru_t* c, int64_t ids, int cap) {
  c->prev = (int32_t*)malloc(sizeof(int32_t) * ids);
  c->next = (int32_t*)malloc(sizeof(int32_t) * ids);
  c->in = (int32_t*)calloc(ids, sizeof(int32_t));
  c->head = c->tail = -1;
  c->count = 0;
  c->cap = cap;
}
static void lru_free(lru_t* c) { free(c->prev); free(c->next); free(c->in); }
static inline void unlink_(lru_t* c, int32_t x) {
  const int32_t p = c->prev[x], n = c->next[x];
  if (p >= 0) c->next[p] = n; else c->head = n;
  if (n >= 0) c->prev[n] = p; else c->tail = p;
}
static inline void push_front(lru_t* c, int32_t x) {
  c->prev[x] = -1;
  c->next[x] = c->head;
  if (c->head >= 0) c->prev[c->head] = x; else c->tail = x;
  c->head = x;
}
static inline int lru_access(lru_t* c, int32_t x) {
  if (c->in[x]) {
    if (c->head != x) { unlink_(c, x); push_front(c, x); }
    return 1;
  }
  if (c->count == c->cap) {
    const int32_t t = c->tail;
    unlink_(c, t);
    c->in[t] = 0;
    c->count--;
  }
  push_front(c, x);
  c->in[x] = 1;
  c->count++;
  return 0;
}

/* tasks of one XCD: e0[i], len[i], kind[i] (0/1: statistics bucket), i in [0, n); workgroup w owns tasks
 * [wg_start[w], wg_start[w+1]).  hits[2], total[2] are accumulated per kind. */
int simulate_xcd(const int32_t* col, int64_t n_ids, const int32_t* e0, const int32_t* len, const int8_t* kind,
                 const int64_t* wg_start, int64_t n_wg, int cap_lines, int conc, int batch, int64_t* hits, int64_t* total) {
  lru_t c;
  lru_init(&c, n_ids, cap_lines);
  int64_t* act = (int64_t*)malloc(sizeof(int64_t) * conc);
  int n_act = 0;
  int64_t next_wg = 0, max_tasks = 0;
  for (int64_t w = 0; w < n_wg; ++w)
    if (wg_start[w + 1] - wg_start[w] > max_tasks) max_tasks = wg_start[w + 1] - wg_start[w];
  int64_t n_tasks = wg_start[n_wg];
  int32_t* done = (int32_t*)calloc(n_tasks > 0 ? n_tasks : 1, sizeof(int32_t));
  while (n_act > 0 || next_wg < n_wg) {
    while (n_act < conc && next_wg < n_wg) act[n_act++] = next_wg++;
    int k = 0;
    for (int a = 0; a < n_act; ++a) {
      const int64_t w = act[a];
      int live = 0;
      for (int64_t t = wg_start[w]; t < wg_start[w + 1]; ++t) {
        int32_t d = done[t];
        if (d >= len[t]) continue;
        int32_t stop = d + batch < len[t] ? d + batch : len[t];
        for (; d < stop; ++d) {
          const int h = lru_access(&c, col[(int64_t)e0[t] + d]);
          hits[kind[t]] += h;
          total[kind[t]]++;
        }
        done[t] = d;
        if (d < len[t]) live = 1;
      }
      if (live) act[k++] = w;
    }
    n_act = k;
  }
  free(done);
  free(act);
  lru_free(&c);
  return 0;
}
