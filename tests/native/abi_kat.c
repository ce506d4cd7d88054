/* Plain-C caller of the C-ABI (include/rxmatch.h only): builds the "ab" known-answer automaton of SURVEY App. B.4
 * as a word array, matches "xabab" on the GPU in both modes and checks the pulses.  Compiled with gcc by
 * tests/test_gpu_parity.py::test_plain_c_caller. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rxmatch.h"

int main(void) {
  static uint32_t W[520];
  /* row_ptr = [0,257,514,515,515]; rows 0 and 1: every byte -> 1, 'a' -> 2; row 2: 'b' -> 3; row 3 empty */
  W[0] = 0; W[1] = 257; W[2] = 514; W[3] = 515; W[4] = 515;
  uint32_t* col = W + 5;
  for (int r = 0; r < 2; r++) {
    for (int c = 0; c < 256; c++) col[r * 257 + c] = ((uint32_t)c << 24) | 1u;
    col[r * 257 + 256] = ((uint32_t)'a' << 24) | 2u;
  }
  col[514] = ((uint32_t)'b' << 24) | 3u;
  rx_nfa* nfa = NULL;
  int rc = rx_nfa_from_words(W, 520, 0, &nfa);
  if (rc) { fprintf(stderr, "from_words: %s\n", rx_strerror(rc)); return 1; }
  rx_nfa_info info;
  rx_nfa_get_info(nfa, &info);
  if (info.size != 4 || info.nnz != 515 || info.n_accept != 1) { fprintf(stderr, "info mismatch\n"); return 1; }
  const uint8_t text[] = "xabab";
  for (int mode = 0; mode < 2; mode++) {
    rx_opts o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.device = 0;
    o.mode = mode ? RX_MODE_TB_COMPAT : RX_MODE_FULL;
    rx_event ev[8];
    uint32_t mc[4] = {0, 0, 0, 0};
    uint64_t fin[1] = {0};
    rx_result r;
    memset(&r, 0, sizeof r);
    r.struct_size = sizeof r;
    r.events = ev; r.events_cap = 8; r.match_count = mc; r.final_active = fin;
    rc = rx_match(nfa, text, 1, 5, 5, NULL, &o, &r);
    if (rc) { fprintf(stderr, "rx_match: %s %s\n", rx_strerror(rc), rx_last_hip_error()); return 1; }
    const size_t want = mode ? 1 : 2;          /* the testbench never sees M_5 */
    if (r.n_events != want || ev[0].k != 3 || ev[0].state != 3 || mc[3] != want) { fprintf(stderr, "mode %d: wrong pulses\n", mode); return 1; }
    if (!mode && (ev[1].k != 5 || fin[0] != ((1ull << 1) | (1ull << 3)))) { fprintf(stderr, "full mode: wrong tail\n"); return 1; }
  }
  rx_nfa_free(nfa);
  puts("abi_kat ok");
  return 0;
}
