/*
 * rx_oracle.c — functional CPU restatement of the reference hot path.  TEST INFRASTRUCTURE.
 * See rx_oracle.h for the "parity unpinned" statement and who may call this.
 *
 * Follows, per pass k (one input byte per stream):
 *   Design/FPGA.v:158,744-765   outer scan over states i = 0..size-1, inactive states skipped
 *   Design/FPGA.v:166-207       row bounds: base = row_ptr[i], deg = row_ptr[i+1]-row_ptr[i]
 *   Design/FPGA.v:210-226       deg == 0  =>  accept pulse for state i (state is a sink)
 *   Design/FPGA.v:227-714       every edge word w = W[size+1+base+j], j < deg, is compared:
 *                               (w>>24) == input_char  =>  next[w & 0xFFFFFF] = 1
 *   Design/FPGA.v:733-741       end of pass: current <- next, next <- 0, next byte requested
 *   Design/FPGA.v:134-147       reset: current = {state 0}
 *   Simulation/testbench_BLK_Mem.sv:53-71   byte m is fed for pass m; the run stops when m
 *                               reaches the trace length, so TB_COMPAT sees passes 0..N-2
 * Word layout: Design/FPGA.v:773,793 (offset = size+1), :881-898 (symbol = w[31:24],
 * target = w[23:0]); .coe token = four u32, leftmost 8 hex digits first (cache[0]=rd_bus[127:96]).
 */
#define _GNU_SOURCE
#include "rx_oracle.h"

#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

void orx_free(void* p) { free(p); }

static char* slurp(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) { fclose(f); return NULL; }
  char* buf = (char*)malloc((size_t)n + 1);
  if (!buf) { fclose(f); return NULL; }
  size_t got = fread(buf, 1, (size_t)n, f);
  fclose(f);
  buf[got] = 0;
  *len = got;
  return buf;
}

static int hexval(int c) {
  if (c >= '0' && c <= '9') return c - '0';
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  if (c >= 'A' && c <= 'F') return c - 'A' + 10;
  return -1;
}

/* radix-16 header, then 32-hex-digit tokens split on [\s,;]+ */
int orx_load_coe(const char* path, uint32_t** words, size_t* nwords) {
  size_t len;
  char* txt = slurp(path, &len);
  if (!txt) return -2;
  const char* key = "memory_initialization_vector";
  char* p = strstr(txt, key);
  if (!p || !strstr(txt, "memory_initialization_radix=16")) { free(txt); return -3; }
  p += strlen(key);
  while (*p && *p != '=') p++;
  if (*p != '=') { free(txt); return -3; }
  p++;
  size_t cap = len / 33 * 4 + 16, n = 0;
  uint32_t* w = (uint32_t*)malloc(cap * sizeof(uint32_t));
  if (!w) { free(txt); return -5; }
  while (*p) {
    while (*p && (isspace((unsigned char)*p) || *p == ',' || *p == ';')) p++;
    if (!*p) break;
    int ndig = 0;
    uint32_t lane[4] = {0, 0, 0, 0};
    while (hexval((unsigned char)*p) >= 0) {
      if (ndig >= 32) { free(w); free(txt); return -3; }
      lane[ndig >> 3] = (lane[ndig >> 3] << 4) | (uint32_t)hexval((unsigned char)*p);
      ndig++;
      p++;
    }
    if (ndig != 32) { free(w); free(txt); return -3; }
    if (n + 4 > cap) { free(w); free(txt); return -3; }
    for (int l = 0; l < 4; l++) w[n++] = lane[l];
  }
  free(txt);
  *words = w;
  *nwords = n;
  return 0;
}

/* SURVEY App. A.3 inference rule. */
int orx_infer_size(const uint32_t* W, size_t nwords, uint32_t* size) {
  if (nwords < 2 || W[0] != 0) return -4;
  uint32_t found = 0;
  int nfound = 0;
  for (size_t s = 1; s < nwords; s++) {
    if (W[s] < W[s - 1]) break; /* row_ptr must be non-decreasing up to s */
    uint64_t used = (uint64_t)W[s] + s + 1;
    if (used > nwords || nwords - used > 3) continue;
    int ok = 1;
    for (size_t j = used; j < nwords; j++) if (W[j] != 0) ok = 0;
    for (size_t j = s + 1; ok && j < used; j++) if ((W[j] & 0xFFFFFFu) >= s) ok = 0;
    if (ok) { found = (uint32_t)s; nfound++; }
  }
  if (nfound != 1) return -4;
  *size = found;
  return 0;
}

/* one 1-2 digit hex value per line. */
int orx_load_mem(const char* path, uint8_t** bytes, size_t* n) {
  size_t len;
  char* txt = slurp(path, &len);
  if (!txt) return -2;
  uint8_t* b = (uint8_t*)malloc(len / 2 + 1);
  if (!b) { free(txt); return -5; }
  size_t cnt = 0;
  char* p = txt;
  while (*p) {
    while (*p && isspace((unsigned char)*p)) p++;
    if (!*p) break;
    unsigned v = 0;
    int nd = 0;
    while (hexval((unsigned char)*p) >= 0) { v = v * 16 + (unsigned)hexval((unsigned char)*p); nd++; p++; }
    if (nd < 1 || nd > 2 || (*p && !isspace((unsigned char)*p))) { free(b); free(txt); return -3; }
    b[cnt++] = (uint8_t)v;
  }
  free(txt);
  *bytes = b;
  *n = cnt;
  return 0;
}

uint64_t orx_passes(size_t n, int mode) {
  if (mode == ORX_MODE_TB_COMPAT) return n >= 1 ? (uint64_t)n - 1 : 0;
  return (uint64_t)n + 1;
}

/* per consumed input byte 1 + sum_{i in S_k}(8 + 4 deg(i)); output 1 bit per pass
 * (rounded up to bytes per stream) + 12 B per accept event. */
uint64_t orx_alg_bytes(uint64_t bytes_consumed, uint64_t sum_active, uint64_t sum_edges,
                       uint64_t n_streams, uint64_t passes_per_stream, uint64_t n_events) {
  return bytes_consumed + 8 * sum_active + 4 * sum_edges +
         n_streams * ((passes_per_stream + 7) / 8) + 12 * n_events;
}

int orx_match_stream(const uint32_t* W, uint32_t size, const uint8_t* bytes, size_t n, int mode,
                     uint32_t stream_id, const uint64_t* init_active, uint32_t* match_count,
                     orx_event* events, size_t events_cap, uint64_t* n_events_io,
                     uint32_t* anymatch, uint64_t* final_active, orx_stats* stats) {
  if (!W || size == 0 || (!bytes && n)) return -1;
  const uint32_t* row_ptr = W;
  const uint32_t* col = W + size + 1;
  const size_t nw = ((size_t)size + 63) / 64;
  uint64_t* cur = (uint64_t*)calloc(nw * 2, sizeof(uint64_t));
  if (!cur) return -5;
  uint64_t* nxt = cur + nw;
  if (init_active) memcpy(cur, init_active, nw * sizeof(uint64_t));
  else cur[0] = 1; /* FPGA.v:146 current[0] <= 1 */

  const uint64_t n_passes = orx_passes(n, mode);
  uint64_t nev = n_events_io ? *n_events_io : 0;
  uint64_t sum_active = 0, sum_edges = 0, ev_local = 0, max_active = 0;
  if (anymatch) memset(anymatch, 0, ((n_passes + 31) / 32) * sizeof(uint32_t));
  int rc = 0;

  for (uint64_t k = 0; k < n_passes; k++) {
    const int has_byte = k < n; /* pass N of FULL mode only checks accepts */
    const uint32_t c = has_byte ? bytes[k] : 0;
    const uint64_t active_before = sum_active;
    for (size_t wi = 0; wi < nw; wi++) {       /* FPGA.v:158/744: scan i upward            */
      uint64_t x = cur[wi];
      while (x) {
        const uint32_t i = (uint32_t)(wi * 64 + (size_t)__builtin_ctzll(x));
        x &= x - 1;
        const uint32_t base = row_ptr[i], deg = row_ptr[i + 1] - base; /* FPGA.v:182-183 */
        if (deg == 0) {                          /* FPGA.v:210-226 accept pulse            */
          if (match_count) match_count[i]++;
          if (events && nev < events_cap) { events[nev].stream = stream_id; events[nev].k = (uint32_t)k; events[nev].state = i; }
          nev++; ev_local++;
          if (anymatch) anymatch[k >> 5] |= 1u << (k & 31);
        }
        if (has_byte) {
          sum_active++;
          sum_edges += deg;
          for (uint32_t j = 0; j < deg; j++) {   /* FPGA.v:264-305 compare each edge        */
            const uint32_t w = col[base + j];
            if ((w >> 24) == c) {
              const uint32_t t = w & 0xFFFFFFu;
              if (t >= size) { rc = -1; goto done; }
              nxt[t >> 6] |= 1ull << (t & 63);
            }
          }
        }
      }
    }
    if (sum_active - active_before > max_active) max_active = sum_active - active_before;
    if (has_byte) {                              /* FPGA.v:733-737 current<=next; next<=0   */
      uint64_t* t = cur; cur = nxt; nxt = t;
      memset(nxt, 0, nw * sizeof(uint64_t));
    }
  }
done:
  if (final_active) memcpy(final_active, cur, nw * sizeof(uint64_t));
  if (n_events_io) *n_events_io = nev;
  if (stats) {
    stats->n_passes = n_passes;
    stats->n_events += ev_local;
    stats->sum_active += sum_active;
    stats->sum_edges += sum_edges;
    if (max_active > stats->max_active) stats->max_active = max_active;
  }
  free(cur < nxt ? cur : nxt);
  return rc;
}

/* ---- threaded batch ---------------------------------------------------------------------- */
typedef struct {
  const uint32_t* W; uint32_t size; const uint8_t* bytes; size_t s0, s1, stream_len, stride; int mode;
  const uint64_t* init_active; uint32_t* match_count; uint32_t* anymatch; size_t anymatch_stride;
  uint64_t* final_active; orx_event* ev; size_t ev_cap; uint64_t ev_n; orx_stats st; uint64_t* mc_total; int rc;
} batch_job;

static void* batch_worker(void* arg) {
  batch_job* j = (batch_job*)arg;
  const size_t nw = ((size_t)j->size + 63) / 64;
  for (size_t s = j->s0; s < j->s1; s++) {
    /* grow the private event buffer so one stream's events always fit */
    uint64_t before = j->ev_n;
    uint32_t* mc = j->match_count ? j->match_count + s * j->size : NULL;
    if (mc) memset(mc, 0, j->size * sizeof(uint32_t));
    for (;;) {
      uint64_t nev = before;
      orx_stats st = {0, 0, 0, 0, 0, 0};
      int rc = orx_match_stream(j->W, j->size, j->bytes + s * j->stride, j->stream_len, j->mode, (uint32_t)s,
                                j->init_active ? j->init_active + s * nw : NULL,
                                NULL, j->ev, j->ev_cap, &nev,
                                j->anymatch ? j->anymatch + s * j->anymatch_stride : NULL,
                                j->final_active ? j->final_active + s * nw : NULL, &st);
      if (rc) { j->rc = rc; return NULL; }
      if (nev > j->ev_cap) { /* retry this stream with a bigger buffer */
        size_t ncap = (size_t)nev * 2 + 64;
        orx_event* ne = (orx_event*)realloc(j->ev, ncap * sizeof(orx_event));
        if (!ne) { j->rc = -5; return NULL; }
        j->ev = ne; j->ev_cap = ncap;
        continue;
      }
      j->ev_n = nev;
      j->st.n_passes = st.n_passes;
      j->st.n_events += st.n_events; j->st.sum_active += st.sum_active; j->st.sum_edges += st.sum_edges;
      if (st.max_active > j->st.max_active) j->st.max_active = st.max_active;
      break;
    }
    /* Blk_Mem_tb's counters (testbench_BLK_Mem.sv:61-69): one increment per pulse */
    for (uint64_t e = before; e < j->ev_n; e++) {
      if (mc) mc[j->ev[e].state]++;
      if (j->mc_total) j->mc_total[j->ev[e].state]++;
    }
  }
  return NULL;
}

int orx_match_batch(const uint32_t* W, uint32_t size, const uint8_t* bytes, size_t n_streams,
                    size_t stream_len, size_t stride, int mode, int nthreads,
                    const uint64_t* init_active, orx_event* events, size_t events_cap,
                    uint64_t* n_events, uint32_t* match_count, uint64_t* match_count_total,
                    uint32_t* anymatch, size_t anymatch_stride, uint64_t* final_active,
                    orx_stats* stats, int* threads_used) {
  if (!W || size == 0 || stride < stream_len) return -1;
  if (nthreads <= 0) { long nc = sysconf(_SC_NPROCESSORS_ONLN); nthreads = nc > 0 ? (int)nc : 1; }
  if ((size_t)nthreads > n_streams) nthreads = n_streams ? (int)n_streams : 1;
  if (threads_used) *threads_used = nthreads;
  batch_job* jobs = (batch_job*)calloc((size_t)nthreads, sizeof(batch_job));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  uint64_t** mct = (uint64_t**)calloc((size_t)nthreads, sizeof(uint64_t*));
  if (!jobs || !th || !mct) return -5;
  size_t per = n_streams / (size_t)nthreads, rem = n_streams % (size_t)nthreads, s = 0;
  for (int t = 0; t < nthreads; t++) {
    batch_job* j = &jobs[t];
    j->W = W; j->size = size; j->bytes = bytes; j->stream_len = stream_len; j->stride = stride; j->mode = mode;
    j->s0 = s; s += per + ((size_t)t < rem ? 1 : 0); j->s1 = s;   /* contiguous blocks, remainder to low ranks */
    j->init_active = init_active; j->match_count = match_count; j->anymatch = anymatch; j->anymatch_stride = anymatch_stride;
    j->final_active = final_active;
    j->ev_cap = 1024; j->ev = (orx_event*)malloc(j->ev_cap * sizeof(orx_event));
    if (match_count_total) { mct[t] = (uint64_t*)calloc(size, sizeof(uint64_t)); j->mc_total = mct[t]; }
    if (nthreads == 1) batch_worker(j);
    else pthread_create(&th[t], NULL, batch_worker, j);
  }
  int rc = 0;
  uint64_t total = 0;
  orx_stats st = {0, 0, 0, 0, 0, 0};
  if (match_count_total) memset(match_count_total, 0, size * sizeof(uint64_t));
  for (int t = 0; t < nthreads; t++) {
    if (nthreads > 1) pthread_join(th[t], NULL);
    batch_job* j = &jobs[t];
    if (j->rc) rc = j->rc;
    for (uint64_t e = 0; e < j->ev_n; e++) { /* jobs are in stream order; each stream's events in (k,state) order */
      if (events && total < events_cap) events[total] = j->ev[e];
      total++;
    }
    st.n_passes = j->st.n_passes ? j->st.n_passes : st.n_passes;
    st.n_events += j->st.n_events; st.sum_active += j->st.sum_active; st.sum_edges += j->st.sum_edges;
    if (j->st.max_active > st.max_active) st.max_active = j->st.max_active;
    if (match_count_total) for (uint32_t i = 0; i < size; i++) match_count_total[i] += mct[t][i];
    free(j->ev); free(mct[t]);
  }
  if (n_streams == 0) st.n_passes = orx_passes(stream_len, mode);
  if (n_events) *n_events = total;
  if (stats) {
    *stats = st;
    uint64_t consumed = (mode == ORX_MODE_TB_COMPAT ? st.n_passes : (uint64_t)stream_len) * n_streams;
    stats->alg_bytes = orx_alg_bytes(consumed, st.sum_active, st.sum_edges, n_streams, st.n_passes, st.n_events);
  }
  free(jobs); free(th); free(mct);
  return rc;
}
