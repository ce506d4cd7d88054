/*
 * rx_cycle.c — clock-by-clock CPU restatement of the reference design.  TEST INFRASTRUCTURE.
 * (See rx_oracle.h: parity is unpinned by the reference; this model is the second, independent
 *  restatement that the functional one must agree with.)
 *
 * Registers and their update rules restate module CSR_traversal:
 *   Design/FPGA.v:119-153   synchronous reset
 *   Design/FPGA.v:158-165   micro-state 0, state i active   -> request row_ptr line
 *   Design/FPGA.v:166-175   micro-state 1                   -> ROM wait (+ next line if lane 3)
 *   Design/FPGA.v:176-207   micro-state 2                   -> range / up_counter
 *   Design/FPGA.v:208-407   micro-state 3, accept / 3-deep line pipeline while range > 0
 *   Design/FPGA.v:408-714   micro-state 3, drain when range == 0
 *   Design/FPGA.v:717-743   micro-state 4                   -> next i / end-of-pass swap
 *   Design/FPGA.v:744-765   micro-state 0, state i inactive
 *   Design/FPGA.v:771-874   combinational address / useful-entries block
 *   Design/FPGA.v:876-900   128-bit line -> 4 x {symbol[31:24], target[23:0]}, lane 0 = [127:96]
 * The ROM `design_1_wrapper` is absent from the reference; it is modelled as SURVEY App. A.4:
 * dout <= mem[addr sampled before the edge], no output register (latency 1).
 * The caller restates Blk_Mem_tb (Simulation/testbench_BLK_Mem.sv:26-87): reset edge, byte
 * feeder on input_char_flag, pulse counters qualified by i, stop when m == m_stop.
 */
#include "rx_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
  /* FPGA.v:39-43 outputs */
  uint32_t i;            /* [19:0] */
  uint16_t rd_address;   /* [15:0] */
  uint8_t input_char_flag, amf, amf2;
  /* FPGA.v:50-111 internals that are actually read */
  uint8_t state;         /* [2:0] */
  uint32_t flag;         /* [9:0], takes 0/1/2 */
  uint32_t range;        /* [23:0] */
  uint32_t up_counter;   /* [23:0] */
  uint8_t range_2_state, range_1_state, range_next;
  uint32_t cache_temp;
  uint8_t block_offset_reg, block_offset_plus_one_reg, block_offset_flag_0; /* [3:0] */
  uint8_t flag_1_or_2, flag_2;                                              /* [1:0] */
  uint8_t ncb_flag_0, ncb_flag_1, ncb_flag_2, ncb_flag_2_prev;              /* [4:0] */
} regs;

typedef struct {
  const uint32_t* W;
  size_t nwords, nlines;
  uint32_t size;
  size_t nw;
  uint64_t *cur1, *cur2, *nxt1, *nxt2;
  regs r;
  /* ROM model: pipe[0] is dout; latency L keeps L-1 further stages */
  int latency;
  uint32_t pipe_addr[4];
  uint8_t input_char, input_char_2;
  /* optional log of compared word addresses (row-coverage probe) */
  uint32_t* log; size_t log_cap, log_n;
  uint64_t bram_reads;
} sim;

static inline int bit(const uint64_t* b, uint32_t i) { return (int)((b[i >> 6] >> (i & 63)) & 1); }
static inline void setbit(uint64_t* b, uint32_t i) { b[i >> 6] |= 1ull << (i & 63); }

static inline uint32_t rom_word(const sim* s, uint32_t line, int lane) {
  size_t a = (size_t)line * 4 + (size_t)lane;
  return a < s->nwords ? s->W[a] : 0u; /* BRAM beyond the .coe initialises to 0 */
}

/* FPGA.v:264-305 style compare of one lane of the line currently on rd_bus */
static inline void cmp_lane(sim* s, int lane) {
  const uint32_t line = s->pipe_addr[0];
  const uint32_t w = rom_word(s, line, lane);
  const uint32_t sym = w >> 24, tgt = w & 0xFFFFFFu;
  if (s->log && s->log_n < s->log_cap) s->log[s->log_n] = line * 4 + (uint32_t)lane;
  if (s->log) s->log_n++;
  if (tgt >= s->size) return; /* next[] is size_range bits wide: out-of-range select writes nothing */
  if (bit(s->cur1, s->r.i) && sym == s->input_char) setbit(s->nxt1, tgt);
  if (bit(s->cur2, s->r.i) && sym == s->input_char_2) setbit(s->nxt2, tgt);
}
/* lanes [bo, bo+n) — the block_offset_flag_0 / no_cached_blocks_flag_0 form (FPGA.v:262-305) */
static inline void cmp_offset_form(sim* s, unsigned bo, unsigned n) {
  for (unsigned l = 0; l < 4; l++) if (l >= bo && n >= l - bo + 1) cmp_lane(s, (int)l);
}
/* lanes [0, n) — the no_cached_blocks_flag_1/_2 form (FPGA.v:315-349) */
static inline void cmp_count_form(sim* s, unsigned n) {
  for (unsigned l = 0; l < 4; l++) if (n > l) cmp_lane(s, (int)l);
}

static void end_of_pass(sim* s, regs* n) { /* FPGA.v:733-741 / 756-763 */
  uint64_t* t;
  t = s->cur1; s->cur1 = s->nxt1; s->nxt1 = t;
  t = s->cur2; s->cur2 = s->nxt2; s->nxt2 = t;
  memset(s->nxt1, 0, s->nw * sizeof(uint64_t));
  memset(s->nxt2, 0, s->nw * sizeof(uint64_t));
  n->i = 0;
  n->input_char_flag = 1;
}

/* One posedge with reset == 0.  Returns 1 if this edge was an end-of-pass edge. */
static int posedge(sim* s) {
  const regs r = s->r; /* values before the edge */
  regs n = r;          /* non-blocking targets   */
  int eop = 0;
  const int active = bit(s->cur1, r.i) || bit(s->cur2, r.i);
  uint32_t cache[4];
  for (int l = 0; l < 4; l++) cache[l] = rom_word(s, s->pipe_addr[0], l); /* FPGA.v:881-884 */

  /* ---- combinational block, FPGA.v:771-874 (only the driven cases are evaluated) ---- */
  const uint32_t offset = (s->size + 1) & 0x1FFFFFFu;
  uint32_t cache_line_no = 0, block_offset = 0, block_offset_plus_one = 0;
  uint32_t no_cached_blocks = 0, up_counter_int = 0, range_int = 0;
  if (active && r.state == 0) { /* :780-786 */
    const uint32_t rai = r.i;
    block_offset = rai & 3;
    block_offset_plus_one = block_offset + 1;
    cache_line_no = (rai >> 2) & 0xFFFFu;
  }
  if (r.flag == 0 && r.state == 3 && r.range > 0) { /* :788-817 */
    const uint32_t rai = (offset + r.up_counter) & 0x1FFFFFFu;
    block_offset = rai & 3;
    cache_line_no = (rai >> 2) & 0xFFFFu;
    const uint32_t ncbi = 4 - block_offset;
    no_cached_blocks = r.range > ncbi ? ncbi : r.range;
    up_counter_int = (r.up_counter + no_cached_blocks) & 0xFFFFFFu;
    range_int = (r.range - no_cached_blocks) & 0xFFFFFFu;
  } else if ((r.flag == 1 || r.flag == 2) && r.state == 3 && r.range > 0) { /* :818-867 */
    cache_line_no = (uint32_t)(r.rd_address + 1) & 0xFFFFu;
    no_cached_blocks = r.range > 4 ? 4 : r.range;
    up_counter_int = (r.up_counter + no_cached_blocks) & 0xFFFFFFu;
    range_int = (r.range - no_cached_blocks) & 0xFFFFFFu;
  }

  /* ---- sequential block, FPGA.v:154-767 ---- */
  if (active && r.state == 0) { /* :158-165 */
    n.input_char_flag = 0;
    n.rd_address = (uint16_t)cache_line_no;
    n.block_offset_reg = (uint8_t)block_offset;
    n.block_offset_plus_one_reg = (uint8_t)block_offset_plus_one;
    n.state = 1;
  } else if (r.state == 1) { /* :166-175 */
    if (r.block_offset_reg == 3) n.rd_address = (uint16_t)(r.rd_address + 1);
    n.state = 2;
  } else if (r.state == 2) { /* :176-207 */
    if (r.block_offset_reg != 3) {
      n.range = (cache[r.block_offset_plus_one_reg & 3] - cache[r.block_offset_reg]) & 0xFFFFFFu;
      n.up_counter = cache[r.block_offset_reg] & 0xFFFFFFu;
      n.flag = 0;
      n.state = 3;
    } else if (r.range_next == 0) {
      n.range_next = 1;
      n.cache_temp = cache[3];
    } else {
      n.range_next = 0;
      n.range = (cache[0] - r.cache_temp) & 0xFFFFFFu;
      n.up_counter = r.cache_temp & 0xFFFFFFu;
      n.flag = 0;
      n.state = 3;
    }
  } else if (r.state == 3) {
    if (r.range == 0 && r.flag == 0) { /* :210-226 accept */
      if (bit(s->cur1, r.i)) n.amf = 1;
      if (bit(s->cur2, r.i)) n.amf2 = 1;
      n.state = 4;
    } else if (r.range > 0) {
      if (r.flag == 0) { /* :229-242 */
        n.rd_address = (uint16_t)cache_line_no;
        n.flag_1_or_2 = 0;
        n.block_offset_flag_0 = (uint8_t)block_offset;
        n.ncb_flag_0 = (uint8_t)no_cached_blocks;
        n.range = range_int;
        n.up_counter = up_counter_int;
        n.flag = 1;
      } else if (r.flag == 1) { /* :243-254 */
        n.flag = 2;
        n.rd_address = (uint16_t)cache_line_no;
        n.flag_1_or_2 = 1;
        n.ncb_flag_1 = (uint8_t)no_cached_blocks;
        n.range = range_int;
        n.up_counter = up_counter_int;
        n.flag_2 = 0;
      } else if (r.flag == 2) { /* :255-406 */
        if (r.flag_2 == 0) { cmp_offset_form(s, r.block_offset_flag_0, r.ncb_flag_0); n.flag_2 = 1; }
        else if (r.flag_2 <= 1) { cmp_count_form(s, r.ncb_flag_1); n.flag_2 = 2; }
        else if (r.flag_2 <= 2) { cmp_count_form(s, r.ncb_flag_2); }
        n.flag_1_or_2 = 2;
        n.ncb_flag_2_prev = r.ncb_flag_2;
        n.ncb_flag_2 = (uint8_t)no_cached_blocks;
        n.range = range_int;
        n.up_counter = up_counter_int;
        n.rd_address = (uint16_t)cache_line_no;
      }
    } else { /* range == 0, :408-714 */
      if (r.flag == 1 && r.range_1_state == 0) {
        n.range_1_state = 1;
      } else if (r.flag == 1 && r.range_1_state == 1) {
        if (r.flag_1_or_2 == 0) cmp_offset_form(s, r.block_offset_flag_0, r.ncb_flag_0);
        n.range_1_state = 0; n.state = 4; n.flag = 0;
      } else if (r.flag == 2 && r.range_2_state == 0) {
        if (r.flag_2 == 2) cmp_count_form(s, r.ncb_flag_2_prev);
        if (r.flag_2 == 1) cmp_count_form(s, r.ncb_flag_1);
        if (r.flag_1_or_2 == 1) cmp_offset_form(s, r.block_offset_flag_0, r.ncb_flag_0);
        n.range_2_state = 1;
      } else if (r.flag == 2 && r.range_2_state == 1) {
        if (r.flag_1_or_2 == 1) cmp_count_form(s, r.ncb_flag_1);
        if (r.flag_1_or_2 == 2) cmp_count_form(s, r.ncb_flag_2);
        n.range_2_state = 0; n.state = 4; n.flag = 0;
      } else {
        n.state = 4; n.flag = 0;
      }
    }
  } else if (r.state == 4) { /* :717-743 */
    n.amf = 0; n.amf2 = 0; n.flag_2 = 0; n.state = 0;
    if (r.i + 1 < s->size) n.i = (r.i + 1) & 0xFFFFFu;
    else { end_of_pass(s, &n); eop = 1; }
  } else if (!active && r.state == 0) { /* :744-765 */
    if (r.i + 1 < s->size) { n.input_char_flag = 0; n.i = (r.i + 1) & 0xFFFFFu; }
    else { end_of_pass(s, &n); eop = 1; }
  }

  /* ---- ROM: dout <= mem[address as it was before this edge] after `latency` edges ---- */
  for (int p = 0; p + 1 < s->latency; p++) s->pipe_addr[p] = s->pipe_addr[p + 1];
  if (s->pipe_addr[s->latency - 1] != r.rd_address) s->bram_reads++;
  s->pipe_addr[s->latency - 1] = r.rd_address;

  s->r = n;
  return eop;
}

static int sim_init(sim* s, const uint32_t* W, size_t nwords, uint32_t size, int latency) {
  memset(s, 0, sizeof(*s));
  if (!W || size == 0 || latency < 1 || latency > 4) return -1;
  s->W = W; s->nwords = nwords; s->size = size; s->latency = latency;
  s->nw = ((size_t)size + 63) / 64;
  s->cur1 = (uint64_t*)calloc(s->nw * 4, sizeof(uint64_t));
  if (!s->cur1) return -5;
  s->cur2 = s->cur1 + s->nw; s->nxt1 = s->cur2 + s->nw; s->nxt2 = s->nxt1 + s->nw;
  /* reset edge, FPGA.v:119-153 */
  s->r.input_char_flag = 1;
  s->cur1[0] = 1; s->cur2[0] = 1;
  return 0;
}
static void sim_free(sim* s) {
  uint64_t* base = s->cur1;
  if (s->cur2 < base) base = s->cur2;
  if (s->nxt1 < base) base = s->nxt1;
  if (s->nxt2 < base) base = s->nxt2;
  free(base);
}

int orx_tb_cycle(const uint32_t* W, size_t nwords, uint32_t size, const uint8_t* lo,
                 const uint8_t* hi, size_t n_mem, uint64_t m_stop, int bram_latency, int skip_idle,
                 uint64_t max_cycles, uint32_t* match_count, uint32_t* match_count_2,
                 orx_event* events, size_t events_cap, uint64_t* cyc_of_event, orx_tb_result* out) {
  sim s;
  if (bram_latency != 1) skip_idle = 0; /* the fast-forward is only exact for the latency-1 ROM */
  int rc = sim_init(&s, W, nwords, size, bram_latency);
  if (rc) return rc;
  if (!lo || !hi || !out || m_stop == 0 || m_stop > n_mem) { sim_free(&s); return -1; }
  memset(out, 0, sizeof(*out));
  if (match_count) memset(match_count, 0, size * sizeof(uint32_t));
  if (match_count_2) memset(match_count_2, 0, size * sizeof(uint32_t));

  uint64_t cycles = 0, m = 0, pass = 0, nev = 0;
  /* t = 10 ns: the one posedge with reset == 1 (testbench_BLK_Mem.sv:28-38); the always block
   * counts it and, seeing input_char_flag == 1, loads byte 0 (:52-59). */
  cycles = 1;
  s.input_char = lo[0]; s.input_char_2 = hi[0]; m = 1;
  if (m == m_stop) goto finish;

  for (;;) {
    if (max_cycles && cycles >= max_cycles) { out->hung = 1; break; }
    /* fast-forward over inactive states: each costs exactly one clock (FPGA.v:744-752) and
     * leaves every register but i and input_char_flag untouched */
    if (skip_idle && s.r.state == 0 && size > 1) {
      uint32_t i = s.r.i;
      if (!bit(s.cur1, i) && !bit(s.cur2, i) && i + 1 < size) {
        uint32_t j = i;
        size_t wi = j >> 6;
        uint64_t x = (s.cur1[wi] | s.cur2[wi]) & (~0ull << (j & 63));
        while (!x && ++wi < s.nw) x = s.cur1[wi] | s.cur2[wi];
        j = x ? (uint32_t)(wi * 64 + (size_t)__builtin_ctzll(x)) : size - 1;
        if (j > size - 1) j = size - 1;
        if (j > i) {
          /* the ROM keeps latching the unchanged rd_address */
          for (int p = 0; p < s.latency; p++) s.pipe_addr[p] = s.r.rd_address;
          cycles += j - i;
          s.r.i = j;
          s.r.input_char_flag = 0;
          continue;
        }
      }
    }
    const int eop = posedge(&s);
    cycles++;                                   /* testbench_BLK_Mem.sv:52 */
    if (eop) pass++;
    if (s.r.input_char_flag) {                  /* :53-59 */
      s.input_char = lo[m]; s.input_char_2 = hi[m]; m++;
    }
    if (s.r.amf) {                              /* :61-64 */
      if (match_count) match_count[s.r.i]++;
      if (events && nev < events_cap) { events[nev].stream = 0; events[nev].k = (uint32_t)pass; events[nev].state = s.r.i; if (cyc_of_event) cyc_of_event[nev] = cycles; }
      nev++; out->n_events[0]++;
    }
    if (s.r.amf2) {                             /* :66-69 */
      if (match_count_2) match_count_2[s.r.i]++;
      if (events && nev < events_cap) { events[nev].stream = 1; events[nev].k = (uint32_t)pass; events[nev].state = s.r.i; if (cyc_of_event) cyc_of_event[nev] = cycles; }
      nev++; out->n_events[1]++;
    }
    if (m == m_stop) break;                     /* :71 */
  }
finish:
  out->total_cycles = cycles;
  out->passes = pass;
  out->bram_reads = s.bram_reads;
  sim_free(&s);
  return 0;
}

/* Row-coverage probe: with only state i active in stream 1, run one pass and log every word
 * address the design compares.  Tests check it equals exactly size+1+row_ptr[i] .. +deg-1. */
int orx_cycle_probe_row(const uint32_t* W, size_t nwords, uint32_t size, uint32_t state_i,
                        int bram_latency, uint8_t c, uint32_t* addrs, size_t cap, size_t* n_addrs,
                        uint64_t* clocks, uint64_t* next_bits /*[ceil(size/64)]*/, int* accepted) {
  sim s;
  int rc = sim_init(&s, W, nwords, size, bram_latency);
  if (rc) return rc;
  if (state_i >= size) { sim_free(&s); return -1; }
  s.cur1[0] = 0; s.cur2[0] = 0;
  setbit(s.cur1, state_i);
  s.input_char = c; s.input_char_2 = c;
  s.log = addrs; s.log_cap = cap; s.log_n = 0;
  s.r.input_char_flag = 0;
  uint64_t clk = 0;
  int acc = 0;
  for (;;) {
    /* stop at the end-of-pass edge; nxt1 has been swapped into cur1 by then */
    int eop = posedge(&s);
    clk++;
    if (s.r.amf) acc = 1;
    if (eop) break;
    if (clk > 16ull * size + 4096) { sim_free(&s); return -9; } /* hung (e.g. latency 2) */
  }
  if (n_addrs) *n_addrs = s.log_n;
  if (clocks) *clocks = clk;
  if (accepted) *accepted = acc;
  if (next_bits) memcpy(next_bits, s.cur1, s.nw * sizeof(uint64_t));
  sim_free(&s);
  return 0;
}

/* SURVEY §3.2 closed form, evaluated on the functional model of both streams. */
int orx_predict_cycles(const uint32_t* W, uint32_t size, const uint8_t* lo, const uint8_t* hi,
                       uint64_t n_passes, uint64_t* total_cycles) {
  if (!W || !size || !lo || !hi || !total_cycles) return -1;
  const uint32_t* row_ptr = W;
  const uint32_t* col = W + size + 1;
  const size_t nw = ((size_t)size + 63) / 64;
  uint64_t* b = (uint64_t*)calloc(nw * 4, sizeof(uint64_t));
  if (!b) return -5;
  uint64_t *c1 = b, *c2 = b + nw, *n1 = b + 2 * nw, *n2 = b + 3 * nw;
  c1[0] = 1; c2[0] = 1;
  uint64_t cycles = 1; /* the reset edge */
  for (uint64_t k = 0; k < n_passes; k++) {
    cycles += size;
    for (size_t wi = 0; wi < nw; wi++) {
      uint64_t x = c1[wi] | c2[wi];
      while (x) {
        const uint32_t i = (uint32_t)(wi * 64 + (size_t)__builtin_ctzll(x));
        x &= x - 1;
        const uint32_t base = row_ptr[i], deg = row_ptr[i + 1] - base;
        uint64_t cost = 3 + ((i & 3) == 3 ? 1 : 0) + 1;
        if (deg == 0) cost += 1;
        else {
          const uint64_t a = (uint64_t)size + 1 + base;
          cost += ((a + deg - 1) >> 2) - (a >> 2) + 1 + 2;
        }
        cycles += cost - 1;
        const int a1 = bit(c1, i), a2 = bit(c2, i);
        for (uint32_t j = 0; j < deg; j++) {
          const uint32_t w = col[base + j], t = w & 0xFFFFFFu;
          if (t >= size) continue;
          if (a1 && (w >> 24) == lo[k]) setbit(n1, t);
          if (a2 && (w >> 24) == hi[k]) setbit(n2, t);
        }
      }
    }
    uint64_t* t;
    t = c1; c1 = n1; n1 = t; t = c2; c2 = n2; n2 = t;
    memset(n1, 0, nw * sizeof(uint64_t)); memset(n2, 0, nw * sizeof(uint64_t));
  }
  *total_cycles = cycles;
  free(b);
  return 0;
}
