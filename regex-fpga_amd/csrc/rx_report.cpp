// rx_report — command-line stand-in for the reference's simulation run.
//
// The reference is used by copying a trace pair to input_trace_lo.mem / input_trace_hi.mem, running
// Blk_Mem_tb, and reading its $display output (Simulation/testbench_BLK_Mem.sv:34-35, :75-85).  This
// tool takes the same .coe and .mem files, runs both traces as two streams of one GPU batch in
// RX_MODE_TB_COMPAT through the C-ABI (include/rxmatch.h), and prints the same report lines:
//     match_count[<p, %11d>] = <count mod 1024, %4d>      highest state first, stream 1 then 2
// Only the C-ABI is used: this file is also the example of a C/C++ caller of librxmatch.so.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rxmatch.h"

static int die(const char* what, int rc) {
  fprintf(stderr, "rx_report: %s: %s [%d] %s\n", what, rx_strerror(rc), rc, rx_last_hip_error());
  return 1;
}

int main(int argc, char** argv) {
  const char *coe = nullptr, *lo = nullptr, *hi = nullptr;
  uint32_t size = 0, kernel = RX_KERNEL_AUTO;
  size_t m_stop = 200000;  // testbench_BLK_Mem.sv:71
  int device = 0;
  bool full = false, events = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--size") && i + 1 < argc) size = (uint32_t)strtoul(argv[++i], nullptr, 0);
    else if (!strcmp(argv[i], "--m-stop") && i + 1 < argc) m_stop = strtoull(argv[++i], nullptr, 0);
    else if (!strcmp(argv[i], "--kernel") && i + 1 < argc) kernel = (uint32_t)strtoul(argv[++i], nullptr, 0);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--full")) full = true;
    else if (!strcmp(argv[i], "--events")) events = true;
    else if (!coe) coe = argv[i];
    else if (!lo) lo = argv[i];
    else if (!hi) hi = argv[i];
    else { fprintf(stderr, "unexpected argument %s\n", argv[i]); return 2; }
  }
  if (!coe || !lo || !hi) {
    fprintf(stderr,
            "usage: rx_report <table.coe> <input_trace_lo.mem> <input_trace_hi.mem>\n"
            "                 [--size N] [--m-stop 200000] [--kernel 0..5] [--device D] [--full] [--events]\n");
    return 2;
  }
  rx_nfa* nfa = nullptr;
  int rc = rx_nfa_load_coe(coe, size, &nfa);
  if (rc) return die(coe, rc);
  rx_nfa_info info;
  rx_nfa_get_info(nfa, &info);
  uint8_t *blo = nullptr, *bhi = nullptr;
  size_t nlo = 0, nhi = 0;
  if ((rc = rx_trace_load_mem(lo, &blo, &nlo))) return die(lo, rc);
  if ((rc = rx_trace_load_mem(hi, &bhi, &nhi))) return die(hi, rc);
  if (nlo < m_stop || nhi < m_stop) {
    fprintf(stderr, "rx_report: traces hold %zu / %zu bytes, need m_stop = %zu\n", nlo, nhi, m_stop);
    return 1;
  }
  std::vector<uint8_t> rows(2 * m_stop);
  memcpy(rows.data(), blo, m_stop);
  memcpy(rows.data() + m_stop, bhi, m_stop);
  std::vector<uint32_t> mc(2 * (size_t)info.size, 0);
  std::vector<rx_event> ev(1 << 20);
  rx_opts o;
  memset(&o, 0, sizeof(o));
  o.struct_size = sizeof(o);
  o.device = device;
  o.mode = full ? RX_MODE_FULL : RX_MODE_TB_COMPAT;
  o.kernel = kernel;
  o.collect_stats = full ? 1 : 2;  // tb-compat: also predict the testbench's clock count
  rx_result r;
  memset(&r, 0, sizeof(r));
  r.struct_size = sizeof(r);
  r.match_count = mc.data();
  r.events = ev.data();
  r.events_cap = ev.size();
  if ((rc = rx_match(nfa, rows.data(), 2, m_stop, m_stop, nullptr, &o, &r))) return die("rx_match", rc);
  for (int s = 0; s < 2; s++) {
    const uint32_t* c = mc.data() + (size_t)s * info.size;
    for (uint32_t p = info.size; p-- > 0;)  // foreach over [size_range-1:0] iterates downwards
      if ((c[p] & 1023u) != 0)              // logic [9:0] counters
        printf("%s[%11d] = %4u\n", s == 0 ? "match_count" : "match_count_2", (int)p, c[p] & 1023u);
  }
  if (r.stats.tb_cycles) {  // $display($time,"\nTotal no. cycles: %d", cycles)  (testbench_BLK_Mem.sv:84)
    const unsigned long long cyc = r.stats.tb_cycles;
    // `int cycles` is a 32-bit signed int in the testbench (:19), so large counts wrap exactly like this cast
    printf("%20llu\nTotal no. cycles: %11d\n", 10ull * cyc + 22ull, (int)(uint32_t)cyc);
    fprintf(stderr, "rx_report: predicted FPGA clocks (unwrapped) %llu\n", cyc);
  }
  if (events)
    for (size_t e = 0; e < r.n_events; e++)
      printf("event stream=%u k=%u state=%u\n", r.events[e].stream, r.events[e].k, r.events[e].state);
  fprintf(stderr, "rx_report: %u states, %u edges; %" PRIu64 " passes/stream; %" PRIu64
                  " accept pulses; kernel %u %.3f ms\n",
          info.size, info.nnz, r.stats.n_passes, r.stats.n_events, r.stats.kernel_used, r.stats.kernel_ms);
  rx_free(blo);
  rx_free(bhi);
  rx_nfa_free(nfa);
  return 0;
}
