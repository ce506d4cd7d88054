/*
 * rx_oracle.h — CPU ORACLE for the CSR-NFA hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / reported CPU baseline.  librxmatch.so never links or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference (Verilog RTL + one SystemVerilog testbench)
 * ships no expected outputs, no assertions and no golden vectors, and no HDL simulator exists in
 * the build image, so the RTL cannot be executed here (see DESIGN.md "Oracle").  What pins this
 * oracle instead: two independent restatements of the reference that must agree with each other —
 *   (1) rx_oracle.c  : functional restatement of the per-pass semantics of Design/FPGA.v
 *   (2) rx_cycle.c   : clock-by-clock restatement of the CSR_traversal FSM (FPGA.v:115-900) driven
 *                      by the Blk_Mem_tb protocol (Simulation/testbench_BLK_Mem.sv:26-87) with a
 *                      latency-1 ROM, which also yields the testbench's "Total no. cycles"
 * — plus the sha256-pinned reference inputs (data/) and the digests in tests/golden/.
 */
#ifndef RX_ORACLE_H
#define RX_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORX_MODE_FULL = 0, ORX_MODE_TB_COMPAT = 1 };

typedef struct orx_event { uint32_t stream, k, state; } orx_event;

typedef struct orx_stats {
  uint64_t n_passes;   /* per stream */
  uint64_t n_events;   /* all streams */
  uint64_t sum_active; /* sum_k |S_k| over all streams */
  uint64_t sum_edges;  /* sum_k sum_{i in S_k} deg(i) */
  uint64_t alg_bytes;  /* SURVEY §8(d) accounting, see orx_alg_bytes() */
  uint64_t max_active; /* largest |S_k| seen in any pass of any stream */
} orx_stats;

/* ---- file formats (Block_Mem .coe, Simulation .mem) ---- */
int orx_load_coe(const char* path, uint32_t** words, size_t* nwords);
int orx_infer_size(const uint32_t* words, size_t nwords, uint32_t* size);
int orx_load_mem(const char* path, uint8_t** bytes, size_t* n);
void orx_free(void* p);

/* ---- functional restatement (rx_oracle.c) ---- */
/* One stream.  All outputs optional.  events receive .stream = stream_id.  init_active NULL = {0}.
 * Returns 0, or -1 on bad arguments / target out of range. */
int orx_match_stream(const uint32_t* words, uint32_t size, const uint8_t* bytes, size_t n,
                     int mode, uint32_t stream_id, const uint64_t* init_active,
                     uint32_t* match_count /*[size], +=*/, orx_event* events, size_t events_cap,
                     uint64_t* n_events, uint32_t* anymatch /*[ceil(n_passes/32)]*/,
                     uint64_t* final_active /*[ceil(size/64)]*/, orx_stats* stats);

/* Batch of independent streams, statically sharded over nthreads POSIX threads (0 = all cores).
 * events come back sorted by (stream,k,state); *n_events is the total even beyond the cap. */
int orx_match_batch(const uint32_t* words, uint32_t size, const uint8_t* bytes, size_t n_streams,
                    size_t stream_len, size_t stride, int mode, int nthreads,
                    const uint64_t* init_active, orx_event* events, size_t events_cap,
                    uint64_t* n_events, uint32_t* match_count /*[n_streams][size]*/,
                    uint64_t* match_count_total /*[size]*/, uint32_t* anymatch,
                    size_t anymatch_stride, uint64_t* final_active, orx_stats* stats,
                    int* threads_used);

uint64_t orx_passes(size_t n, int mode);
uint64_t orx_alg_bytes(uint64_t passes_total, uint64_t sum_active, uint64_t sum_edges,
                       uint64_t n_streams, uint64_t passes_per_stream, uint64_t n_events);

/* ---- clock-accurate restatement of FPGA.v + testbench (rx_cycle.c) ---- */
typedef struct orx_tb_result {
  uint64_t total_cycles;   /* what `$display("Total no. cycles: %d", cycles)` prints */
  uint64_t passes;         /* completed passes (end-of-pass edges seen)               */
  uint64_t n_events[2];    /* accept pulses per stream (beyond cap still counted)     */
  uint64_t bram_reads;     /* distinct line addresses latched by the ROM model        */
  uint64_t hung;           /* 1 if max_cycles was hit before termination              */
} orx_tb_result;

/* lo/hi: the two byte arrays as $readmemh fills them (index 0 = first line); n_mem entries each
 * (entries past the file are X in the simulator; here they must not be reached).  m_stop is the
 * testbench's hard-coded 200000 (testbench_BLK_Mem.sv:71).  bram_latency 1 is the reference's ROM;
 * other values exist only to show the design breaks with them.  skip_idle!=0 fast-forwards runs of
 * inactive states (provably equal clock count; tests compare both).  match_count*: [size] raw
 * 32-bit pulse counts (the testbench's 10-bit wrap is applied by the caller).  events: (k,state)
 * per stream with .stream = 0/1, in pulse order; cyc_of_event optional [cap] clock of each pulse. */
int orx_tb_cycle(const uint32_t* words, size_t nwords, uint32_t size, const uint8_t* lo,
                 const uint8_t* hi, size_t n_mem, uint64_t m_stop, int bram_latency, int skip_idle,
                 uint64_t max_cycles, uint32_t* match_count, uint32_t* match_count_2,
                 orx_event* events, size_t events_cap, uint64_t* cyc_of_event,
                 orx_tb_result* out);

/* Row-coverage probe: only state_i active (stream 1), one pass on byte c.  Logs every table word
 * address the design compares; tests check the log is exactly that state's CSR row. */
int orx_cycle_probe_row(const uint32_t* words, size_t nwords, uint32_t size, uint32_t state_i,
                        int bram_latency, uint8_t c, uint32_t* addrs, size_t cap, size_t* n_addrs,
                        uint64_t* clocks, uint64_t* next_bits, int* accepted);

/* Closed-form clock count of the same run from the functional model:
 * 1 reset clock + sum over passes of [ size + sum_{i active in either stream} (cost(i)-1) ]. */
int orx_predict_cycles(const uint32_t* words, uint32_t size, const uint8_t* lo, const uint8_t* hi,
                       uint64_t n_passes, uint64_t* total_cycles);

#ifdef __cplusplus
}
#endif
#endif
