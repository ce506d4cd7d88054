// rx_kernels.hip — gfx950 (CDNA4) kernels for the per-byte active-state update of a CSR NFA.
//
// What they replace: the per-clock FSM of module CSR_traversal (Design/FPGA.v:115-768).  The
// FPGA walks i = 0..size-1 every byte and spends >= size clocks per byte on inactive states
// (FPGA.v:744-765); here the active set is a compacted list per stream, so work is proportional
// to |S_k|, and thousands of independent streams are resident at once.
//
// Six kernels, one result (tests/test_gpu_parity.py compares every one of them with the oracle):
//   rx_csr_wave_kernel   ONE WAVEFRONT OWNS ONE INPUT STREAM and reads the state-major CSR exactly as the .coe holds
//                        it (row_ptr pair, then the whole row, as FPGA.v:166-207 / :227-714 do): long rows are swept
//                        by all 64 lanes (256 B coalesced per load), short rows one lane per row.  North-star form.
//   rx_sym_wave_kernel   one wavefront per stream over the load-time slice index: one u32 per (state, byte) that
//                        holds "the current byte's slice" of that row (rx_internal.hpp).  Also the kernel that
//                        finishes streams the kernels below hand off (resume mode).
//   rx_sym_group_kernel  G lanes per stream, 64/G streams per wavefront (static lane groups).
//   rx_sym_pack_kernel   S streams per wavefront, the 64 lanes assigned dynamically to one wave-wide list of
//                        (stream, state) entries — the throughput kernel RX_KERNEL_AUTO normally picks.
//                        FOLD builds keep the always-on `.*` state out of the lists and step over the passes in which
//                        nothing happens to any of the wave's streams; PRUNE builds insert only what survives the next byte.
//   rx_sym_reg_kernel    one wavefront per stream, the active set in a VGPR (one state per lane, updated in place from a
//                        precomputed index): the shortest pass — few long streams (the reference's own run) and small
//                        batches.  A second build steps over groups of passes in which no state is active.
//   rx_dfa_kernel        one LANE per stream over a lazily built subset-construction cache (opt-in).
// Common to the two wave-per-stream kernels:
//   * per-stream state lives in that wave's private LDS slice: two size-bit bitmasks (dedup
//     filter for `next`, and the dense spill form of `current`) and two active-state lists;
//   * the stream's bytes are fetched 256 B per wave-load (one dword per lane, coalesced) one
//     chunk ahead, and the current byte is broadcast with v_readlane (wave-uniform, so the
//     symbol ends up in an SGPR);
//   * next-state insertion = ds_or_rtn_b32 on the bitmask (dedup) + __ballot/mbcnt/__popcll to
//     allocate list slots and to flag accept states — no workgroup barrier anywhere: the four
//     waves of a block never communicate;
//   * active set larger than RX_LIST_CAP: the list stops growing but the bitmask keeps every bit, and
//     the next pass walks the bitmask instead ("dense" form).  Results are identical either way.
// No MFMA anywhere: this is integer gather / bit-scatter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "rx_internal.hpp"

// Diagnostic A/B build (make -C csrc ab -> ../librxmatch_ab.so, never shipped): the pack kernel's lanes WITHOUT a list
// entry sit out of the filter-clear store and of the two filter atomics instead of running them as no-ops on spread-out
// words.  Used once per round to re-measure what those extra LDS lanes cost (profiles/r02_lds_ab/).
#ifndef RX_AB_PREDICATE_IDLE
#define RX_AB_PREDICATE_IDLE 0
#endif

namespace {

// 64-lane ballot straight from the predicate (the generic __ballot goes through an int and costs two
// extra VALU instructions per call)
template <typename T>
__device__ __forceinline__ uint64_t wballot(T pred) {
  return __builtin_amdgcn_ballot_w64(pred != 0);
}
__device__ __forceinline__ uint32_t rank_below(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// base + number of set bits of m below this lane (the addend rides along in v_mbcnt_lo for free)
__device__ __forceinline__ uint32_t rank_below_plus(uint64_t m, uint32_t base) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
}
// The waves of a block are independent; ordering is only needed between the lanes of one wave,
// which execute LDS instructions in program order.  This keeps the compiler from reordering.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src_lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_lane);
}

// Kernel arguments for COLD code (hand-offs, the final sets at the end of a kernel): read from the kernel-argument segment at
// the place of use instead of being kept in SGPRs across the pass loop, where they would push hot values into lane spills
// (the pack kernel sits at 100 of 102 SGPRs).  The address goes through an empty asm so that the loads are not merged with
// the hoisted ones.  The kernels take their RxParams by value as the first argument: offset 0 of the segment.
typedef const RxParams __attribute__((address_space(4))) * RxColdParams;
__device__ __forceinline__ RxColdParams cold_params() {
  unsigned long long a = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return (RxColdParams)a;
}

// ---- launch prologue: what used to be host-side resets between two launches ------------------------
// Block 0 zeroes the counter set of the NEXT launch (RxParams::zero_next); nothing in this launch touches that set.
__device__ __forceinline__ void zero_next_counters(const RxParams& p) {
  if (blockIdx.x == 0 && p.zero_next)
    for (uint32_t w = threadIdx.x; w < p.zero_words; w += blockDim.x) p.zero_next[w] = 0ull;
}
// Kernels that set bits in the final-set rows (instead of storing whole rows) first clear the rows of their own streams
// [first, first + n); the same wavefront ORs the bits in much later, through the same L2.
__device__ __forceinline__ void zero_final_rows(const RxParams& p, uint32_t first, uint32_t n, uint32_t lane) {
  if (!p.final_active) return;
  uint32_t* rows = p.final_active + (size_t)first * p.nw64x2;
  const uint32_t words = n * p.nw64x2;
  for (uint32_t w = lane; w < words; w += 64u) rows[w] = 0u;
}

// ---- input bytes: 256-byte chunks, one dword per lane ------------------------------------------
struct ByteFeed {
  const uint8_t* base;
  uint32_t len;
  bool aligned;
  __device__ __forceinline__ uint32_t load_chunk(uint32_t chunk, uint32_t lane) const {
    const uint32_t off = chunk * 256u + lane * 4u;
    uint32_t w = 0;
    if (aligned && off + 4u <= len) {
      w = *reinterpret_cast<const uint32_t*>(base + off);
    } else {
#pragma unroll
      for (uint32_t b = 0; b < 4; b++)
        if (off + b < len) w |= (uint32_t)base[off + b] << (8u * b);
    }
    return w;
  }
};

// ---- accept pulses (FPGA.v:210-226 -> testbench_BLK_Mem.sv:61-69) -----------------------------
__device__ __forceinline__ void emit_events(const RxParams& p, bool acc, uint32_t state, uint32_t stream,
                                            uint32_t k, uint32_t lane, uint32_t& am_word) {
  const uint64_t ma = wballot(acc);
  if (ma == 0) return;
  const uint32_t cnt = (uint32_t)__popcll(ma);
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(p.ev_count, (unsigned long long)cnt);
  const uint32_t blo = bcast((uint32_t)base, 0), bhi = bcast((uint32_t)(base >> 32), 0);
  base = ((unsigned long long)bhi << 32) | blo;
  if (acc) {
    const unsigned long long idx = base + rank_below(ma);
    if (p.events && idx < p.events_cap) {
      rx_event e;
      e.stream = p.stream_base + stream;
      e.k = p.k_base + k;
      e.state = state;
      p.events[idx] = e;
    }
    if (p.match_count) atomicAdd(&p.match_count[(size_t)stream * p.size + state], 1u);
    if (p.match_count_total) atomicAdd(&p.match_count_total[state], 1ull);
  }
  am_word |= 1u << (k & 31u);
}

// ---- per-stream LDS state -----------------------------------------------------------------------
struct StreamState {
  uint32_t* cb;     // bitmask of the current set (valid only when dense)
  uint32_t* nb;     // dedup filter / bitmask of the next set (all zero at pass start)
  uint32_t* clist;  // current active list (valid when !dense)
  uint32_t* nlist;
  uint32_t n_cur, n_next;
  uint32_t cap;     // entries a list holds (RxParams::lds_words_per_stream minus the two bitmasks, halved)
  bool dense;
};

// insert target entry `t` (state id + flag bits) into the next set; wave-uniform call
__device__ __forceinline__ void emit_target(StreamState& st, bool pred, uint32_t t, uint32_t lane) {
  (void)lane;
  bool fresh = false;
  if (pred) {
    const uint32_t s = t & RXE_TGT_MASK;
    const uint32_t bit = 1u << (s & 31u);
    const uint32_t old = atomicOr(&st.nb[s >> 5], bit);  // ds_or_rtn_b32: next[t] <= 1 with dedup
    fresh = (old & bit) == 0;
  }
  const uint64_t m = wballot(fresh);
  if (m) {
    const uint32_t slot = st.n_next + rank_below(m);
    if (fresh && slot < st.cap) st.nlist[slot] = t;
    st.n_next += (uint32_t)__popcll(m);
  }
}

__device__ __forceinline__ void stream_reset(const RxParams& p, StreamState& st, uint32_t* my,
                                             const uint32_t* init_row, uint32_t lane) {
  st.cb = my;
  st.nb = my + p.nw32;
  st.clist = st.nb + p.nw32;
  st.cap = (p.lds_words_per_stream - 2u * p.nw32) >> 1;
  st.nlist = st.clist + st.cap;
  for (uint32_t w = lane; w < 2u * p.nw32; w += 64u) my[w] = 0u;
  st.n_next = 0;
  if (init_row) {  // chunked streaming / spill hand-off: resume from a given active set
    const uint32_t* row = init_row;
    for (uint32_t w = lane; w < p.nw32; w += 64u) st.cb[w] = row[w];
    st.dense = true;
    st.n_cur = 0;
  } else {  // FPGA.v:134-147: current = {state 0}
    if (lane == 0) st.clist[0] = p.state0_entry;
    st.dense = false;
    st.n_cur = 1;
  }
  wave_sync();
}

// end of a byte-consuming pass: current <- next, next <- 0 (FPGA.v:733-737)
__device__ __forceinline__ void stream_swap(const RxParams& p, StreamState& st, uint32_t lane) {
  wave_sync();
  if (st.dense)
    for (uint32_t w = lane; w < p.nw32; w += 64u) st.cb[w] = 0u;
  if (st.n_next > st.cap) {  // list overflowed: the bitmask is the set
    uint32_t* t = st.cb; st.cb = st.nb; st.nb = t;
    st.dense = true;
    st.n_cur = 0;
  } else {  // the list is the set: wipe the filter words it touched
    for (uint32_t i = lane; i < st.n_next; i += 64u) st.nb[(st.nlist[i] & RXE_TGT_MASK) >> 5] = 0u;
    uint32_t* t = st.clist; st.clist = st.nlist; st.nlist = t;
    st.dense = false;
    st.n_cur = st.n_next;
  }
  st.n_next = 0;
  wave_sync();
}

// The final set of one stream as a compact list (RxParams::fin_states): `src` = its bitmask (LDS, nw32 words), one
// wavefront.  One atomic for the space, then the states in ascending order (word order, lane order within a sweep).
__device__ __forceinline__ void emit_compact_from_words(const RxParams& p, const uint32_t* src, uint32_t stream, uint32_t lane) {
  uint32_t mine = 0;
  for (uint32_t w = lane; w < p.nw32; w += 64u) mine += (uint32_t)__popc(src[w]);
  uint32_t total = mine;
  for (int d = 32; d >= 1; d >>= 1) total += (uint32_t)__shfl_xor((int)total, d);
  unsigned long long base = 0;
  if (lane == 0 && total) base = atomicAdd(p.fin_count, (unsigned long long)total);
  base = ((unsigned long long)bcast((uint32_t)(base >> 32), 0) << 32) | bcast((uint32_t)base, 0);
  if (lane == 0) {
    p.fin_off[stream] = (uint32_t)(base < p.fin_cap ? base : p.fin_cap);
    p.fin_cnt[stream] = total;
  }
  unsigned long long at = base;
  for (uint32_t w0 = 0; w0 < p.nw32 && total; w0 += 64u) {
    const uint32_t w = w0 + lane;
    uint32_t bits = w < p.nw32 ? src[w] : 0u;
    const uint32_t n = (uint32_t)__popc(bits);
    uint32_t incl = n;  // inclusive prefix sum over the lanes
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= (uint32_t)d) incl += v;
    }
    unsigned long long o = at + (incl - n);
    while (bits) {
      const uint32_t bpos = (uint32_t)__builtin_ctz(bits);
      bits &= bits - 1u;
      if (o < p.fin_cap) p.fin_states[o] = w * 32u + bpos;
      o++;
    }
    at += bcast(incl, 63);
  }
}

__device__ __forceinline__ void stream_store_final(const RxParams& p, StreamState& st, uint32_t stream,
                                                   uint32_t lane) {
  if (!p.final_active && !p.fin_states) return;
  uint32_t* row = p.final_active + (size_t)stream * p.nw64x2;
  uint32_t* src = st.cb;
  if (!st.dense) {  // rebuild the bitmask from the list in the (all-zero) filter
    for (uint32_t i = lane; i < st.n_cur; i += 64u) {
      const uint32_t s = st.clist[i] & RXE_TGT_MASK;
      atomicOr(&st.nb[s >> 5], 1u << (s & 31u));
    }
    src = st.nb;
    wave_sync();
  }
  if (p.fin_states) emit_compact_from_words(p, src, stream, lane);
  else
    for (uint32_t w = lane; w < p.nw64x2; w += 64u) row[w] = w < p.nw32 ? src[w] : 0u;
  wave_sync();
  if (!st.dense)
    for (uint32_t i = lane; i < st.n_cur; i += 64u) st.nb[(st.clist[i] & RXE_TGT_MASK) >> 5] = 0u;
  wave_sync();
}

// Walk the current set in groups of <= 64 states (one per lane) and hand each group to `body`.
// body(valid, entry) is called wave-uniformly.
template <bool WITH_FLAGS, typename Body>
__device__ __forceinline__ void for_each_active(const RxParams& p, const StreamState& st, uint32_t lane,
                                                Body&& body) {
  if (!st.dense) {
    for (uint32_t b = 0; b < st.n_cur; b += 64u) {
      const uint32_t idx = b + lane;
      const bool valid = idx < st.n_cur;
      const uint32_t e = valid ? st.clist[idx] : 0u;
      body(valid, e);
    }
  } else {
    for (uint32_t w0 = 0; w0 < p.nw32; w0 += 64u) {
      const uint32_t wi = w0 + lane;
      const uint32_t word = wi < p.nw32 ? st.cb[wi] : 0u;
      uint64_t m = wballot(word != 0u);
      while (m) {
        const uint32_t src = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        const uint32_t wv = bcast(word, src);
        const bool valid = lane < 32u && ((wv >> lane) & 1u);
        uint32_t e = (w0 + src) * 32u + lane;
        if (WITH_FLAGS && valid && ((p.accept_bits[e >> 5] >> (e & 31u)) & 1u)) e |= RXE_ACCEPT;
        body(valid, e);
      }
    }
  }
}

// =================================================================================================
// Kernel 1: wavefront-per-stream over the state-major CSR exactly as loaded
// =================================================================================================
template <bool STATS>
__global__ void __launch_bounds__(256) rx_csr_wave_kernel(const RxParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  uint32_t* my = lds + (size_t)wib * p.lds_words_per_stream;
  const uint32_t* __restrict__ rp = p.words;                // row_ptr  (FPGA.v:780-786 addresses i>>2)
  const uint32_t* __restrict__ col = p.words + p.size + 1;  // offset = size+1 (FPGA.v:773,793)
  unsigned long long st_active = 0, st_edges = 0;
  zero_next_counters(p);

  for (uint32_t stream = blockIdx.x * wpb + wib; stream < p.n_streams; stream += gridDim.x * wpb) {
    StreamState st;
    stream_reset(p, st, my, p.init_active ? p.init_active + (size_t)stream * p.nw64x2 : nullptr, lane);
    ByteFeed feed;
    feed.base = p.bytes + (size_t)stream * p.stride;
    feed.len = p.stream_len;
    feed.aligned = ((reinterpret_cast<uintptr_t>(feed.base)) & 3u) == 0;
    uint32_t cur_word = 0, nxt_word = feed.load_chunk(0, lane);
    uint32_t am_word = 0;

    for (uint32_t k = 0; k < p.n_passes; k++) {
      const bool consume = k < p.n_consume;
      uint32_t c = 0;
      if (consume) {
        if ((k & 255u) == 0) {
          cur_word = nxt_word;
          nxt_word = feed.load_chunk((k >> 8) + 1u, lane);  // one chunk ahead
        }
        c = (bcast(cur_word, (k >> 2) & 63u) >> ((k & 3u) * 8u)) & 0xFFu;  // input_char
      }
      for_each_active<false>(p, st, lane, [&](bool valid, uint32_t s) {
        uint32_t base = 0, deg = 0;
        if (valid) {  // FPGA.v:182-183: range = row_ptr[i+1]-row_ptr[i], up_counter = row_ptr[i]
          base = rp[s];
          deg = rp[s + 1] - base;
        }
        emit_events(p, valid && deg == 0, s, stream, k, lane, am_word);
        if (!consume) return;
        if (STATS && valid) { st_active += 1; st_edges += deg; }
        // short rows: one lane per row; all of the row's edge words are requested before any is examined
        const bool small = valid && deg > 0 && deg <= RX_SMALL_DEG;
        if (wballot(small)) {
          uint32_t w[RX_SMALL_DEG];
#pragma unroll
          for (uint32_t j = 0; j < RX_SMALL_DEG; j++) w[j] = (small && j < deg) ? col[base + j] : 0u;
#pragma unroll
          for (uint32_t j = 0; j < RX_SMALL_DEG; j++) {
            const bool hit = small && j < deg && (w[j] >> 24) == c;  // FPGA.v:264: transition == input_char
            if (wballot(hit)) emit_target(st, hit, w[j] & RXE_TGT_MASK, lane);
          }
        }
        // long rows: all 64 lanes sweep one row, 256 B per load, up to five loads (320 edges) in flight
        uint64_t mb = wballot(valid && deg > RX_SMALL_DEG);
        while (mb) {
          const uint32_t src = (uint32_t)__builtin_ctzll(mb);
          mb &= mb - 1;
          const uint32_t b = bcast(base, src), d = bcast(deg, src);
          for (uint32_t j0 = 0; j0 < d; j0 += 320u) {
            uint32_t w[5];
#pragma unroll
            for (uint32_t u = 0; u < 5; u++) {
              const uint32_t j = j0 + u * 64u + lane;
              w[u] = j < d ? col[b + j] : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < 5; u++) {
              const uint32_t j = j0 + u * 64u + lane;
              const bool hit = j < d && (w[u] >> 24) == c;
              if (wballot(hit)) emit_target(st, hit, w[u] & RXE_TGT_MASK, lane);
            }
          }
        }
      });
      if (consume) stream_swap(p, st, lane);
      if (p.anymatch && ((k & 31u) == 31u || k + 1 == p.n_passes)) {
        if (lane == 0) p.anymatch[(size_t)stream * p.anymatch_stride + (k >> 5)] = am_word;
        am_word = 0;
      }
    }
    stream_store_final(p, st, stream, lane);
  }
  if (STATS) {
    if (st_active) atomicAdd(&p.counters[1], st_active);
    if (st_edges) atomicAdd(&p.counters[2], st_edges);
  }
}

// =================================================================================================
// Kernel 2: wavefront-per-stream over the per-(state, byte) slice index
// =================================================================================================
// ---- finishing hand-offs with the whole WORKGROUP on one stream (SURVEY §8 f2's variant: one stream's active states split
// over the wavefronts of a workgroup) ---------------------------------------------------------------------------------
// A stream that the pack / group / register kernels hand off holds more states than their lists take (the trap of
// bench.py's handoff_mix_T: 222 per pass).  One wavefront alone walks such a set in four dependent sweeps per pass; here the
// wavefronts of the block take every fourth sweep (list form) or every fourth group of 64 bitmask words (dense form) of the
// SAME stream: the next set's dedup bitmask and list are shared (LDS atomics work across the wavefronts of a block), list
// slots come from one LDS counter (one atomic per wavefront and insertion step), and two block barriers per pass separate
// "everybody inserts" from "current <- next".  Used when few streams were handed off (RX_BLOCK_RESUME_MAX): with many, one
// wavefront per stream fills the chip better.  Wave-uniform / block-uniform control flow throughout.
static constexpr uint32_t RX_BLOCK_RESUME_MAX = 4096;

__device__ __forceinline__ void resume_streams_by_block(const RxParams& p, uint32_t* lds, uint32_t total) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wib = tid >> 6, nthr = blockDim.x, nwav = blockDim.x >> 6;
  const uint32_t cap = (p.lds_words_per_stream - 2u * p.nw32) >> 1;
  uint32_t* cb = lds;                 // region of wavefront 0: the stream's two bitmasks and two lists
  uint32_t* nb = cb + p.nw32;
  uint32_t* clist = nb + p.nw32;
  uint32_t* nlist = clist + cap;
  uint32_t* ctrl = lds + p.lds_words_per_stream;  // (region of wavefront 1, unused here) [0],[1]: list counters of even / odd passes, [2]: any-match bits
  const uint32_t* __restrict__ symidx = p.symidx;
  const uint32_t* __restrict__ ovf = p.ovf;
  for (uint32_t idx = blockIdx.x; idx < total; idx += gridDim.x) {  // (block-uniform)
    const uint32_t stream = p.spill_streams[idx], k0 = p.spill_k[idx];
    const uint32_t* row = p.spill_rows + (size_t)idx * p.nw64x2;
    __syncthreads();  // the previous stream's final set has been stored
    for (uint32_t w = tid; w < p.nw32; w += nthr) { cb[w] = row[w]; nb[w] = 0u; }
    if (tid == 0) {
      ctrl[0] = 0u;
      ctrl[1] = 0u;
      // the hand-off pass already emitted its accept pulses; keep the bits of its partial bitmap word
      ctrl[2] = p.anymatch ? p.anymatch[(size_t)stream * p.anymatch_stride + (k0 >> 5)] : 0u;
    }
    bool dense = true;  // S_k arrives as a bitmask row
    uint32_t n_cur = 0;
    ByteFeed feed;
    feed.base = p.bytes + (size_t)stream * p.stride;
    feed.len = p.stream_len;
    feed.aligned = ((reinterpret_cast<uintptr_t>(feed.base)) & 3u) == 0;
    uint32_t cur_word = 0, nxt_word = feed.load_chunk(k0 >> 8, lane);  // (every wavefront keeps its own copy of the bytes)
    __syncthreads();

    for (uint32_t k = k0; k < p.n_passes; k++) {
      const bool consume = k < p.n_consume;
      const bool pulses = k != k0;
      uint32_t* cnt = &ctrl[k & 1u];
      uint32_t c = 0;
      if (consume) {
        if ((k & 255u) == 0 || k == k0) {
          cur_word = nxt_word;
          nxt_word = feed.load_chunk((k >> 8) + 1u, lane);
        }
        c = (bcast(cur_word, (k >> 2) & 63u) >> ((k & 3u) * 8u)) & 0xFFu;
      }
      // insert one target per lane into the stream's next set; wave-uniform call
      auto emit = [&](bool pred, uint32_t t) {
        bool fresh = false;
        if (pred) {
          const uint32_t s = t & RXE_TGT_MASK;
          const uint32_t bit = 1u << (s & 31u);
          fresh = (atomicOr(&nb[s >> 5], bit) & bit) == 0u;
        }
        const uint64_t m = wballot(fresh);
        if (m) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(cnt, (uint32_t)__popcll(m));
          const uint32_t slot = bcast(base, 0) + rank_below(m);
          if (fresh && slot < cap) nlist[slot] = t;
        }
      };
      // up to one entry of S_k per lane: pulse, slice, insertions; wave-uniform call
      auto step = [&](bool valid, uint32_t e) {
        const uint32_t s = e & RXE_TGT_MASK;
        const bool acc = valid && (e & RXE_ACCEPT);
        if (pulses) {
          uint32_t am = 0;
          emit_events(p, acc, s, stream, k, lane, am);
          if (am != 0u && lane == 0) atomicOr(&ctrl[2], am);
        }
        if (!consume) return;
        const uint32_t ent = (valid && !acc) ? symidx[(size_t)s * 256u + c] : 0u;
        if (wballot(ent & RXE_SELF)) emit((ent & RXE_SELF) != 0, s);
        if (wballot(ent & RXE_INLINE)) emit((ent & RXE_INLINE) != 0, ent & (RXE_TGT_MASK | RXE_ACCEPT));
        uint64_t mo = wballot(ent & RXE_OVF);
        if (mo == 0ull) return;
        const uint32_t mycnt = (ent & RXE_OVF) ? ovf[ent & RXE_TGT_MASK] : 0u;
        while (mo) {
          const uint32_t src = (uint32_t)__builtin_ctzll(mo);
          mo &= mo - 1;
          const uint32_t off = bcast(ent & RXE_TGT_MASK, src);
          const uint32_t n = bcast(mycnt, src);
          for (uint32_t j0 = 0; j0 < n; j0 += 64u) {
            const bool act = j0 + lane < n;
            emit(act, act ? ovf[off + 1u + j0 + lane] : 0u);
          }
        }
      };
      const uint32_t w_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)wib) * 64u;  // (scalar for the compiler too)
      if (!dense) {
        for (uint32_t b = w_first; b < n_cur; b += nwav * 64u) {
          const uint32_t i = b + lane;
          step(i < n_cur, i < n_cur ? clist[i] : 0u);
        }
      } else {
        for (uint32_t w0 = w_first; w0 < p.nw32; w0 += nwav * 64u) {
          const uint32_t wi = w0 + lane;
          const uint32_t word = wi < p.nw32 ? cb[wi] : 0u;
          uint64_t m = wballot(word != 0u);
          while (m) {
            const uint32_t src = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            const uint32_t wv = bcast(word, src);
            const bool valid = lane < 32u && ((wv >> lane) & 1u);
            uint32_t e = (w0 + src) * 32u + lane;
            if (valid && ((p.accept_bits[e >> 5] >> (e & 31u)) & 1u)) e |= RXE_ACCEPT;
            step(valid, e);
          }
        }
      }
      __syncthreads();  // every wavefront has inserted (and pulsed)
      if (consume) {  // current <- next (FPGA.v:733-737)
        const uint32_t n_next = *cnt;
        if (dense)
          for (uint32_t w = tid; w < p.nw32; w += nthr) cb[w] = 0u;
        if (n_next > cap) {  // the bitmask is the set
          uint32_t* t = cb; cb = nb; nb = t;
          dense = true;
          n_cur = 0;
        } else {  // the list is the set: wipe the filter words it touched
          for (uint32_t i = tid; i < n_next; i += nthr) nb[(nlist[i] & RXE_TGT_MASK) >> 5] = 0u;
          uint32_t* t = clist; clist = nlist; nlist = t;
          dense = false;
          n_cur = n_next;
        }
      }
      if (tid == 0) {
        ctrl[(k + 1u) & 1u] = 0u;  // (nobody touches the other counter before the barrier below)
        if (p.anymatch && ((k & 31u) == 31u || k + 1 == p.n_passes)) {
          p.anymatch[(size_t)stream * p.anymatch_stride + (k >> 5)] = ctrl[2];
          ctrl[2] = 0u;
        }
      }
      __syncthreads();
    }
    if (wib == 0) {  // the final set, by one wavefront (the others wait at the top of the loop / leave)
      StreamState st;
      st.cb = cb; st.nb = nb; st.clist = clist; st.nlist = nlist;
      st.n_cur = n_cur; st.n_next = 0; st.cap = cap; st.dense = dense;
      stream_store_final(p, st, stream, lane);
    }
  }
}

template <bool STATS>
__global__ void __launch_bounds__(256) rx_sym_wave_kernel(const RxParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  uint32_t* my = lds + (size_t)wib * p.lds_words_per_stream;
  const uint32_t* __restrict__ rp = p.words;
  const uint32_t* __restrict__ symidx = p.symidx;
  const uint32_t* __restrict__ ovf = p.ovf;
  unsigned long long st_active = 0, st_edges = 0;
  zero_next_counters(p);
  // resume mode: finish the streams the group kernel handed off (their active set outgrew its list)
  uint32_t total = p.n_streams;
  if (p.resume) {
    const unsigned long long n = *p.spill_count;
    total = n < p.n_streams ? (uint32_t)n : p.n_streams;
    if (!STATS && wpb >= 2u && total <= RX_BLOCK_RESUME_MAX) {  // few hand-offs: the whole block on one stream at a time
      resume_streams_by_block(p, lds, total);
      return;
    }
  }

  for (uint32_t idx = blockIdx.x * wpb + wib; idx < total; idx += gridDim.x * wpb) {
    uint32_t stream = idx, k0 = 0;
    const uint32_t* init_row = p.init_active ? p.init_active + (size_t)idx * p.nw64x2 : nullptr;
    if (p.resume) {
      stream = p.spill_streams[idx];
      k0 = p.spill_k[idx];
      init_row = p.spill_rows + (size_t)idx * p.nw64x2;
    }
    StreamState st;
    stream_reset(p, st, my, init_row, lane);
    ByteFeed feed;
    feed.base = p.bytes + (size_t)stream * p.stride;
    feed.len = p.stream_len;
    feed.aligned = ((reinterpret_cast<uintptr_t>(feed.base)) & 3u) == 0;
    uint32_t cur_word = 0, nxt_word = feed.load_chunk(k0 >> 8, lane);
    // the hand-off pass already emitted its accept pulses; keep the bits of its partial bitmap word
    uint32_t am_word = (p.resume && p.anymatch) ? p.anymatch[(size_t)stream * p.anymatch_stride + (k0 >> 5)] : 0u;

    for (uint32_t k = k0; k < p.n_passes; k++) {
      const bool consume = k < p.n_consume;
      const bool pulses = !(p.resume && k == k0);
      uint32_t c = 0;
      if (consume) {
        if ((k & 255u) == 0 || k == k0) {
          cur_word = nxt_word;
          nxt_word = feed.load_chunk((k >> 8) + 1u, lane);
        }
        c = (bcast(cur_word, (k >> 2) & 63u) >> ((k & 3u) * 8u)) & 0xFFu;
      }
      for_each_active<true>(p, st, lane, [&](bool valid, uint32_t e) {
        const uint32_t s = e & RXE_TGT_MASK;
        const bool acc = valid && (e & RXE_ACCEPT);
        if (pulses) emit_events(p, acc, s, stream, k, lane, am_word);
        if (!consume) return;
        if (STATS && valid && pulses) { st_active += 1; st_edges += rp[s + 1] - rp[s]; }  // hand-off pass already counted
        // the current byte's slice of row s: one dword
        const uint32_t ent = (valid && !acc) ? symidx[(size_t)s * 256u + c] : 0u;
        if (wballot(ent & RXE_SELF)) emit_target(st, (ent & RXE_SELF) != 0, s, lane);
        if (wballot(ent & RXE_INLINE))
          emit_target(st, (ent & RXE_INLINE) != 0, ent & (RXE_TGT_MASK | RXE_ACCEPT), lane);
        // rows with several targets on this byte: the whole wave expands one list at a time, 64 targets
        // per step (coalesced), instead of one lane walking it alone
        uint64_t mo = wballot(ent & RXE_OVF);
        const uint32_t mycnt = (ent & RXE_OVF) ? ovf[ent & RXE_TGT_MASK] : 0u;  // all list lengths in one gather
        while (mo) {
          const uint32_t src = (uint32_t)__builtin_ctzll(mo);
          mo &= mo - 1;
          const uint32_t off = bcast(ent & RXE_TGT_MASK, src);
          const uint32_t cnt = bcast(mycnt, src);
          for (uint32_t j0 = 0; j0 < cnt; j0 += 64u) {
            const bool act = j0 + lane < cnt;
            const uint32_t t = act ? ovf[off + 1u + j0 + lane] : 0u;
            emit_target(st, act, t, lane);
          }
        }
      });
      if (consume) stream_swap(p, st, lane);
      if (p.anymatch && ((k & 31u) == 31u || k + 1 == p.n_passes)) {
        if (lane == 0) p.anymatch[(size_t)stream * p.anymatch_stride + (k >> 5)] = am_word;
        am_word = 0;
      }
    }
    stream_store_final(p, st, stream, lane);
  }
  if (STATS) {
    if (st_active) atomicAdd(&p.counters[1], st_active);
    if (st_edges) atomicAdd(&p.counters[2], st_edges);
  }
}

// =================================================================================================
// Kernel 3: G lanes per stream, 64/G streams per wavefront, slice index
// =================================================================================================
// The snort_16 / l7 active sets are tiny (1-4 states per byte), so a whole wavefront per stream
// leaves ~60 lanes idle and the kernel is bound by instruction issue, not by memory.  Here a
// wavefront carries 64/G streams in lock-step: the G lanes of a group take the group's active-list
// entries G at a time, every lane gathers "its" (state, byte) slice dword, and insertions use
//   * a 1024-bit hashed filter per stream in LDS (ds_or_rtn_b32): bit clear => certainly new;
//   * __ballot + popcount of the group's nibble for list slots;
//   * bit already set (true duplicate or hash collision) => that lane scans the group's next list
//     (rare; one lane per group at a time so two equal targets cannot both be appended).
// Two filters alternate by pass: the entries of the current list were inserted through filter P, so
// the lane that processes an entry zeroes its word in P (no separate clearing sweep) while new
// targets go through filter Q.
// The "pinned" state (a state with a self-loop on all 256 bytes that state 0 reaches on every byte —
// the `.*` state 1 of snort_16) never leaves a stream once active, so it is kept as one flag per
// stream instead of a list entry; its slice row sits in LDS (1 KB per block) and its out-edges are
// taken by the otherwise idle lane 0 of the group.
// A stream whose next set would exceed RX_GROUP_CAP is handed to the wavefront-per-stream kernel in
// resume mode (S_k as a bitmask row + k), so results stay exact for any automaton / input.
template <int G>
struct GroupLayout {
  // wider groups are for automata/inputs with larger active sets: longer lists, larger filters
  static constexpr uint32_t FW = G >= 16 ? 4u * RX_GROUP_FILTER_WORDS : (G == 8 ? 2u * RX_GROUP_FILTER_WORDS : RX_GROUP_FILTER_WORDS);
  static constexpr uint32_t CAP = G >= 16 ? 128u : (G == 8 ? 64u : RX_GROUP_CAP);
  static constexpr uint32_t BUFW = 4u * G;  // byte window: 16 B per lane
  static constexpr uint32_t RAW = 2u * FW + 2u * CAP + BUFW;
  // region stride == G (mod 2G): within a 32-lane half the groups' list slots (g*REGION + j) fall on
  // distinct LDS banks, and equal filter words of different streams are at most 2-way conflicted
  static constexpr uint32_t PAD = ((G + 2u * G * 64u) - RAW) % (2u * G);
  static constexpr uint32_t REGION = RAW + PAD;
  static constexpr uint32_t SPW = 64u / G;  // streams per wavefront
  static constexpr uint32_t PINW = 256u;    // pinned state's slice row, shared by the block
};

template <int G, bool STATS>
__global__ void __launch_bounds__(256) rx_sym_group_kernel(const RxParams p) {
  using L = GroupLayout<G>;
  constexpr uint32_t GMASK = (1u << G) - 1u;
  constexpr uint32_t CH = 16u * G;  // bytes of each stream held in LDS at a time
  constexpr uint32_t HMASK = 32u * L::FW - 1u;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const uint32_t g = lane / G, j = lane % G;
  const uint32_t gshift = lane & ~(uint32_t)(G - 1);
  const uint32_t below = (1u << j) - 1u;
  uint32_t* pinrow = lds;
  uint32_t* reg = lds + L::PINW + ((size_t)wib * L::SPW + g) * L::REGION;
  uint32_t* filts = reg;                         // [2][FW]
  uint32_t* lists = reg + 2u * L::FW;            // [2][CAP]
  uint32_t* bufw = lists + 2u * L::CAP;          // [BUFW] byte window
  const uint32_t* __restrict__ rp = p.words;
  const uint32_t* __restrict__ symidx = p.symidx;
  const uint32_t* __restrict__ ovf = p.ovf;
  unsigned long long st_active = 0, st_edges = 0;

  // pinned state's row -> LDS (self-loop flag stripped: survival is the per-stream flag)
  const bool have_pin = p.pin_state != 0xFFFFFFFFu;
  if (have_pin)
    for (uint32_t w = threadIdx.x; w < 256u; w += blockDim.x)
      pinrow[w] = symidx[(size_t)p.pin_state * 256u + w] & ~RXE_SELF;
  __syncthreads();  // the only block-wide barrier; after this the waves never meet again

  zero_next_counters(p);
  const uint32_t wave = blockIdx.x * wpb + wib;
  const uint32_t stream = wave * L::SPW + g;
  bool alive = stream < p.n_streams;
  if (wave * L::SPW < p.n_streams)
    zero_final_rows(p, wave * L::SPW, p.n_streams - wave * L::SPW < L::SPW ? p.n_streams - wave * L::SPW : L::SPW, lane);
  const uint8_t* base = p.bytes + (size_t)(alive ? stream : 0) * p.stride;
  const bool aligned = (reinterpret_cast<uintptr_t>(base) & 3u) == 0;

  // 16 bytes of this lane's share of chunk `chunk`
  auto load16 = [&](uint32_t chunk, uint32_t (&w)[4]) {
    const uint32_t off = chunk * CH + j * 16u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t o = off + 4u * q;
      uint32_t v = 0;
      if (alive) {
        if (aligned && o + 4u <= p.stream_len) v = *reinterpret_cast<const uint32_t*>(base + o);
        else
          for (uint32_t b = 0; b < 4; b++)
            if (o + b < p.stream_len) v |= (uint32_t)base[o + b] << (8u * b);
      }
      w[q] = v;
    }
  };

  for (uint32_t w = j; w < 2u * L::FW; w += G) filts[w] = 0u;
  if (j == 0) lists[0] = p.state0_entry;  // FPGA.v:134-147: current = {state 0}
  uint32_t n_cur = 1, n_next = 0, tog = 0, am_word = 0, cw = 0;
  uint32_t pinned = 0;                    // 1 once the pinned state is active in this group's stream
  uint32_t nxt[4];
  load16(0, nxt);
  wave_sync();

  for (uint32_t k = 0; k < p.n_passes; k++) {
    const bool consume = k < p.n_consume;
    uint32_t c = 0;
    if (consume) {
      if ((k & 3u) == 0) {
        if ((k % CH) == 0) {  // refill the LDS byte window, fetch the next one into registers
          wave_sync();
#pragma unroll
          for (int q = 0; q < 4; q++) bufw[j * 4u + q] = nxt[q];
          load16(k / CH + 1u, nxt);
          wave_sync();
        }
        cw = bufw[(k % CH) >> 2];  // 4 input bytes of this group's stream (same address in all G lanes)
      }
      c = (cw >> ((k & 3u) * 8u)) & 0xFFu;  // input_char
    }
    uint32_t* clist = lists + tog * L::CAP;
    uint32_t* nlist = lists + (tog ^ 1u) * L::CAP;
    uint32_t* fcur = filts + tog * L::FW;          // filter the current entries went through
    uint32_t* fnext = filts + (tog ^ 1u) * L::FW;  // filter for this pass's insertions (all zero)
    n_next = 0;
    uint32_t pin_next = pinned;

    const uint32_t gbelow = below;
    auto grp_bits = [&](uint64_t m) { return (uint32_t)(m >> gshift) & GMASK; };
    // accept pulses of up to one entry per lane; wave-uniform call
    auto pulses = [&](bool acc, uint32_t s) {
      const uint64_t ma = wballot(acc);
      if (ma) {
        uint32_t dummy = 0;
        emit_events(p, acc, s, stream, k, lane, dummy);
        if (grp_bits(ma)) am_word |= 1u << (k & 31u);
      }
    };
    // exact membership check for candidates whose filter bit was already set (true duplicate or hash
    // collision): one lane per group at a time scans the group's next list; wave-uniform call
    auto resolve = [&](bool maybe, uint32_t t) {
      uint32_t gm = grp_bits(wballot(maybe));
      while (wballot(gm != 0)) {
        const bool mine = maybe && gm != 0 && (gm & (0u - gm)) == (1u << j);
        bool found = false;
        if (mine) {
          const uint32_t lim = n_next < L::CAP ? n_next : L::CAP;
          for (uint32_t q = 0; q < lim; q++) found |= ((nlist[q] ^ t) & RXE_TGT_MASK) == 0;
        }
        const bool app = mine && !found;
        const uint32_t ga = grp_bits(wballot(app));
        if (app && n_next < L::CAP) nlist[n_next] = t & (RXE_TGT_MASK | RXE_ACCEPT);
        n_next += ga ? 1u : 0u;
        gm &= gm - 1u;
        wave_sync();
      }
    };
    // insert one candidate per lane into the group's next set; wave-uniform call (slow paths)
    auto insert = [&](bool pred, uint32_t t) {
      const bool topin = pred && (t & RXE_PIN) != 0;  // the pinned state is a flag, not a list entry
      if (grp_bits(wballot(topin))) pin_next = 1;
      pred = pred && !topin;
      const uint32_t h = t & HMASK;
      const uint32_t bit = 1u << (h & 31u);
      uint32_t old = 0;
      if (pred) old = atomicOr(&fnext[h >> 5], bit);
      const bool fresh = pred && (old & bit) == 0;
      const bool maybe = pred && (old & bit) != 0;
      const uint32_t gb = grp_bits(wballot(fresh));
      const uint32_t slot = n_next + (uint32_t)__popc(gb & gbelow);
      if (fresh && slot < L::CAP) nlist[slot] = t & (RXE_TGT_MASK | RXE_ACCEPT);
      n_next += (uint32_t)__popc(gb);
      if (wballot(maybe)) {
        wave_sync();
        resolve(maybe, t);
      }
    };
    // rows with several targets on this byte: the G lanes of the group expand one such list at a time,
    // G targets per step, instead of the owning lane walking it alone; wave-uniform call
    auto insert_ovf = [&](uint32_t ent) {
      uint32_t gm = grp_bits(wballot(ent & RXE_OVF));  // lanes of MY group that hold an overflow slice
      while (wballot(gm != 0)) {
        const uint32_t srcj = gm ? (uint32_t)__builtin_ctz(gm) : 0u;
        const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((gshift + srcj) << 2), (int)(ent & RXE_TGT_MASK));
        const uint32_t cnt = gm ? ovf[off] : 0u;
        for (uint32_t q0 = 0; wballot(q0 < cnt) != 0; q0 += G) {
          const bool act = q0 + j < cnt;
          insert(act, act ? ovf[off + 1u + q0 + j] : 0u);
        }
        gm &= gm - 1u;
      }
    };

    if (!consume) {  // last pass of full mode: accept check only
      for (uint32_t it = 0;; it++) {
        const uint32_t idx = it * G + j;
        const bool valid = alive && idx < n_cur;
        if (wballot(valid) == 0) break;
        const uint32_t e = valid ? clist[idx] : 0u;
        pulses(valid && (e & RXE_ACCEPT), e & RXE_TGT_MASK);
      }
    } else {
      // ---- batched fast path: virtual entries 0 .. 2G-1 of every stream, two per lane ----------
      // virtual entry 0 of a pinned stream is the pinned state (lane 0); list entries follow.
      // Phase A: both list reads in flight.
      const bool is_pin = alive && pinned && j == 0;
      const uint32_t i0 = j - pinned, i1 = G + j - pinned;
      const bool v0 = alive && j >= pinned && i0 < n_cur;
      const bool v1 = alive && i1 < n_cur;
      uint32_t e0 = 0, e1 = 0;
      if (v0) e0 = clist[i0];
      if (v1) e1 = clist[i1];
      const uint32_t s0 = e0 & RXE_TGT_MASK, s1 = e1 & RXE_TGT_MASK;
      const bool a0 = v0 && (e0 & RXE_ACCEPT), a1 = v1 && (e1 & RXE_ACCEPT);
      if (wballot(a0 || a1)) { pulses(a0, s0); pulses(a1, s1); }
      // these entries went through filter fcur: zero their words for the pass after next
      if (v0) fcur[(s0 & HMASK) >> 5] = 0u;
      if (v1) fcur[(s1 & HMASK) >> 5] = 0u;
      // Phase B: both slice dwords in flight (the current byte's slice of rows s0, s1)
      uint32_t x0 = 0, x1 = 0;
      if (v0 && !a0) x0 = symidx[(size_t)s0 * 256u + c];
      if (v1 && !a1) x1 = symidx[(size_t)s1 * 256u + c];
      if (is_pin) x0 = pinrow[c];
      if (STATS) {
        if (v0) { st_active += 1; st_edges += rp[s0 + 1] - rp[s0]; }
        if (v1) { st_active += 1; st_edges += rp[s1 + 1] - rp[s1]; }
        if (is_pin) { st_active += 1; st_edges += p.pin_degree; }
      }
      // Phase C: four candidates per lane: self0, inline0, self1, inline1
      bool p0 = (x0 & RXE_SELF) != 0, p1 = (x0 & RXE_INLINE) != 0;
      bool p2 = (x1 & RXE_SELF) != 0, p3 = (x1 & RXE_INLINE) != 0;
      {
        const bool tp1 = p1 && (x0 & RXE_PIN), tp3 = p3 && (x1 & RXE_PIN);
        if (grp_bits(wballot(tp1 || tp3))) pin_next = 1;
        p1 = p1 && !tp1;
        p3 = p3 && !tp3;
      }
      const uint32_t h0 = e0 & HMASK, h1 = x0 & HMASK, h2 = e1 & HMASK, h3 = x1 & HMASK;
      const uint32_t b0 = 1u << (h0 & 31u), b1 = 1u << (h1 & 31u), b2 = 1u << (h2 & 31u), b3 = 1u << (h3 & 31u);
      uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
      if (p0) o0 = atomicOr(&fnext[h0 >> 5], b0);  // four ds_or_rtn_b32 back to back, one wait
      if (p1) o1 = atomicOr(&fnext[h1 >> 5], b1);
      if (p2) o2 = atomicOr(&fnext[h2 >> 5], b2);
      if (p3) o3 = atomicOr(&fnext[h3 >> 5], b3);
      // Phase D: slots by ballot + popcount of the group's bits
      const bool f0 = p0 && !(o0 & b0), f1 = p1 && !(o1 & b1), f2 = p2 && !(o2 & b2), f3 = p3 && !(o3 & b3);
      const bool m0 = p0 && (o0 & b0), m1 = p1 && (o1 & b1), m2 = p2 && (o2 & b2), m3 = p3 && (o3 & b3);
      const uint32_t g0 = grp_bits(wballot(f0)), g1 = grp_bits(wballot(f1));
      const uint32_t g2 = grp_bits(wballot(f2)), g3 = grp_bits(wballot(f3));
      uint32_t slot = n_next + (uint32_t)__popc(g0 & gbelow);
      if (f0 && slot < L::CAP) nlist[slot] = e0 & (RXE_TGT_MASK | RXE_ACCEPT);
      n_next += (uint32_t)__popc(g0);
      slot = n_next + (uint32_t)__popc(g1 & gbelow);
      if (f1 && slot < L::CAP) nlist[slot] = x0 & (RXE_TGT_MASK | RXE_ACCEPT);
      n_next += (uint32_t)__popc(g1);
      slot = n_next + (uint32_t)__popc(g2 & gbelow);
      if (f2 && slot < L::CAP) nlist[slot] = e1 & (RXE_TGT_MASK | RXE_ACCEPT);
      n_next += (uint32_t)__popc(g2);
      slot = n_next + (uint32_t)__popc(g3 & gbelow);
      if (f3 && slot < L::CAP) nlist[slot] = x1 & (RXE_TGT_MASK | RXE_ACCEPT);
      n_next += (uint32_t)__popc(g3);
      if (wballot(m0 || m1 || m2 || m3)) {  // rare
        wave_sync();
        resolve(m0, e0);
        resolve(m1, x0);
        resolve(m2, e1);
        resolve(m3, x1);
      }
      if (wballot((x0 | x1) & RXE_OVF)) {  // rows with several targets on this byte
        insert_ovf(x0);
        insert_ovf(x1);
      }
      // ---- streams with more than 2G virtual entries: one entry per lane per iteration ----------
      if (wballot(alive && n_cur + pinned > 2u * G)) {
        for (uint32_t it = 2;; it++) {
          const uint32_t idx = it * G + j - pinned;
          const bool valid = alive && idx < n_cur;
          if (wballot(valid) == 0) break;
          const uint32_t e = valid ? clist[idx] : 0u;
          const uint32_t s = e & RXE_TGT_MASK;
          const bool acc = valid && (e & RXE_ACCEPT);
          pulses(acc, s);
          if (valid) fcur[(s & HMASK) >> 5] = 0u;
          if (STATS && valid) { st_active += 1; st_edges += rp[s + 1] - rp[s]; }
          const uint32_t ent = (valid && !acc) ? symidx[(size_t)s * 256u + c] : 0u;
          insert((ent & RXE_SELF) != 0, e);
          insert((ent & RXE_INLINE) != 0, ent);
          if (wballot(ent & RXE_OVF)) insert_ovf(ent);
        }
      }
    }

    if (consume) {
      // next set too large for the group's list: hand the stream (S_k, k) to the wave kernel
      const bool spill = alive && n_next > L::CAP;
      if (wballot(spill)) {
        uint32_t slot = 0;
        if (spill && j == 0) slot = (uint32_t)atomicAdd(p.spill_count, 1ull);
        slot = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(gshift << 2), (int)slot);  // group leader's slot
        if (spill) {
          if (j == 0) {
            p.spill_streams[slot] = stream;
            p.spill_k[slot] = k;
            if (p.anymatch) p.anymatch[(size_t)stream * p.anymatch_stride + (k >> 5)] = am_word;
          }
          uint32_t* row = p.spill_rows + (size_t)slot * p.nw64x2;
          for (uint32_t w = j; w < p.nw64x2; w += G) {  // S_k as a bitmask row, word by word
            uint32_t vv = 0;
            for (uint32_t q = 0; q < n_cur; q++) {
              const uint32_t sq = clist[q] & RXE_TGT_MASK;
              if ((sq >> 5) == w) vv |= 1u << (sq & 31u);
            }
            if (pinned && (p.pin_state >> 5) == w) vv |= 1u << (p.pin_state & 31u);
            row[w] = vv;
          }
          alive = false;
        }
      }
      // current <- next (FPGA.v:733-737)
      tog ^= 1u;
      n_cur = n_next;
      pinned = pin_next;
      wave_sync();
    }
    if (p.anymatch && ((k & 31u) == 31u || k + 1 == p.n_passes)) {
      if (alive && j == 0) p.anymatch[(size_t)stream * p.anymatch_stride + (k >> 5)] = am_word;
      am_word = 0;
    }
  }
  // final active set: the row was zeroed at the start of this kernel; set the listed bits
  if (p.final_active && alive) {
    const uint32_t* clist = lists + tog * L::CAP;
    uint32_t* row = p.final_active + (size_t)stream * p.nw64x2;
    for (uint32_t idx = j; idx < n_cur; idx += G) {
      const uint32_t sq = clist[idx] & RXE_TGT_MASK;
      atomicOr(&row[sq >> 5], 1u << (sq & 31u));
    }
    if (pinned && j == 0) atomicOr(&row[p.pin_state >> 5], 1u << (p.pin_state & 31u));
  }
  if (STATS) {
    if (st_active) atomicAdd(&p.counters[1], st_active);
    if (st_edges) atomicAdd(&p.counters[2], st_edges);
  }
}

// =================================================================================================
// Kernel 4: S streams per wavefront, lanes assigned DYNAMICALLY to (stream, state) entries
// =================================================================================================
// Static G-lane groups still idle most lanes (|S_k| is 1-3 but the wave pays for the largest group).
// Here a wavefront owns S streams and ONE wave-wide active list whose entries carry the stream slot
// (bits 28:24) next to the state id: lane L simply takes entry L, whatever stream it belongs to, so a
// pass is one sweep of ceil(N/64) iterations with N = sum of the S active-set sizes (about 2.3 S).
//   * per stream in LDS: two alternating 1024-bit hashed filters (insert through one, the entries
//     that went through the other zero their word when processed), a 64-byte window of the input's
//     byte classes (one cooperative 16 B/lane wave-load per 64 passes), one word of any-match bits;
//   * the slice index is stored per byte CLASS (bytes whose edges are the same in every state);
//   * slots of the wave-wide next list come from __ballot + mbcnt (wave-level, scalar count);
//   * filter bit already set => wave-parallel exact scan of the next list for that (stream,state);
//   * lanes 0..S-1 additionally own one stream each for bitmap stores and hand-off rows;
//   * rows with several targets on the byte: lists laid side by side over the lanes, one sweep; PRUNE:
//     only the targets that survive the stream's next byte;
//   * next list would exceed the layout's CAPW => all S streams are handed to the wave kernel (resume);
//   * the passes run in chunked loops (refill per 64, bitmap store per 32, mode as a compile-time tag).
template <int S, bool PRUNE, bool FOLD>
struct PackLayout {
  // few streams per wavefront = automata/inputs with many active states per stream: longer list, wider filters
  // (FOLD builds keep the always-on state out of the lists: about one entry per stream is left, half the filter does)
  static constexpr uint32_t FW = S <= 4 ? 2u * RX_GROUP_FILTER_WORDS : (FOLD ? RX_GROUP_FILTER_WORDS / 2u : RX_GROUP_FILTER_WORDS);
  // (the PRUNE build serves automata with bursts of active states: twice the list for up to 13 streams per wavefront)
  static constexpr uint32_t CAPW = S <= 4 ? 512u : (PRUNE && S <= 13 ? 2u * RX_PACK_CAP : (S >= 48 ? 256u : RX_PACK_CAP));
  // 64 input bytes (as byte classes) per stream; look-ahead builds (PRUNE, FOLD): + byte 64 = first class of the next
  // chunk (look-ahead at the window's last byte) + a pad word that keeps the stride odd
  static constexpr uint32_t WINW = (PRUNE || FOLD) ? 18 : 16;
  // any-match bits of 256 passes per stream: eight words, stored as ONE aligned 32-byte group by the stream's owner lane (a
  // lone dword store costs 32 bytes of HBM write traffic on gfx950: tools/write_calib.hip; round 2 stored 33 of them per
  // stream, 69 MB for 8.7 MB of bits)
  static constexpr uint32_t AMW = 8;
  static constexpr uint32_t STRIDE = (2u * FW + WINW + AMW) | 1u;   // odd => banks spread
  static constexpr uint32_t LISTW = CAPW + 128u;            // a sweep appends at most 128 entries past CAPW: no bounds check
  // lists, stream regions, spill slots; FOLD: + which of the window's 64 passes have an emission of the folded state
  static constexpr uint32_t WAVE_WORDS = 2u * LISTW + S * STRIDE + S + (FOLD ? 2u : 0u);
  static constexpr uint32_t CMAPW = 64;                     // byte -> class map (256 bytes), shared by the block
  // FOLD: the block also keeps the pinned state's folding table (RxParams::pin_tab) behind the class map
};

// PROF: diagnostic build only (RX_PROFILE_PACK=1): s_memtime stamps around the phases of a pass; the sums go to
// counters[8..15], which nothing else reads.  Its run time is not representative — read the SHARES.
// PRUNE: look-ahead pruning (rx_host.cpp) — multi-target rows insert only the targets that survive the stream's NEXT
// byte.  The statistics build counts every active state of the reference's sets, so it always runs unpruned; it marks
// the entries that came out of multi-target rows (bit 29) and counts those that die at once, which is what AUTO
// needs to know to pick PRUNE and the streams per wavefront that go with it.
// FOLD: always-on-state folding (rx_host.cpp) — the pinned `.*` state is no list entry: every stream from reset holds it
// from pass 1 on, and what its row emits on the current byte comes from a (class x next class) table in LDS, looked
// up by the stream's OWNER lane while the list entries' slice gather is in flight; of its targets only those that are
// accept states or survive the next byte are inserted.  Six-bit stream slots (up to 64 streams per wavefront).
template <int S, bool STATS, bool PROF, bool PRUNE, bool FOLD>
__global__ void __launch_bounds__(FOLD ? 512 : 256) rx_sym_pack_kernel(const RxParams p) {
  static_assert(!((PRUNE || FOLD) && (STATS || PROF)), "statistics / stamped builds run unpruned and unfolded");
  constexpr bool LOOK = PRUNE || FOLD;  // the window carries one byte of look-ahead
  constexpr uint32_t MARK = 1u << 29;  // STATS only: entry was inserted from a multi-target row
  unsigned long long t_prev = 0, t_sum[7] = {0, 0, 0, 0, 0, 0, 0};
  // PROF: shader clock this wave ran at = (s_memtime delta) / (s_memrealtime delta) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
  const unsigned long long t0c = PROF ? __builtin_amdgcn_s_memtime() : 0ull, t0r = PROF ? __builtin_amdgcn_s_memrealtime() : 0ull;
  auto stamp = [&](int phase) {
    if (PROF) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      if (phase >= 0) t_sum[phase] += t - t_prev;
      t_prev = t;
    }
  };
  using L = PackLayout<S, PRUNE, FOLD>;
  constexpr uint32_t HMASK = 32u * L::FW - 1u;
  constexpr uint32_t SID_SHIFT = 24, SID_BITS = FOLD ? 63u : 31u, SID_MASK = SID_BITS << SID_SHIFT;
  constexpr uint32_t KEY_MASK = RXE_TGT_MASK | SID_MASK;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  uint32_t* cmapw = lds;                              // [64] byte -> class
  const uint8_t* cmap = reinterpret_cast<const uint8_t*>(cmapw);
  const uint32_t pin_words = FOLD ? p.n_classes * p.pin_cols : 0u;
  const uint32_t* pintab = lds + L::CMAPW;           // FOLD: [n_classes][n_classes + 1], shared by the block
  uint32_t* wl = lds + L::CMAPW + pin_words + (size_t)wib * L::WAVE_WORDS;  // [2][CAPW] wave-wide lists
  uint32_t* sreg0 = wl + 2u * L::LISTW;              // [S][STRIDE]: filters[2][FW], window[WINW], am word
  uint32_t* slotw = sreg0 + S * L::STRIDE;           // [S] spill slots
  uint32_t* busyw = slotw + S;                       // FOLD: [2] = 64 bits, one per pass of the current window
  const uint32_t* __restrict__ rp = p.words;
  constexpr bool prune = PRUNE;
  const uint32_t* __restrict__ symidx = PRUNE ? p.symidx_p : p.symidx_c;
  // PRUNE, narrow index: target / list number in bits 15:0 of a slice word, the inline target's next-class bits above
  const bool narrow = PRUNE && p.prune_narrow != 0u;
  const uint32_t xtmask = narrow ? 0xFFFFu : RXE_TGT_MASK;
  const uint32_t ncls = p.n_classes;
  uint32_t ncls_v = ncls;  // the same in a VGPR: a VALU instruction with an SGPR operand issues at half rate
  asm volatile("" : "+v"(ncls_v));
  const uint32_t* __restrict__ ovf = p.ovf;
  unsigned long long st_active = 0, st_edges = 0, st_cost = 0, st_ovf = 0, st_dead = 0, fold_entries = 0;

  zero_next_counters(p);
  for (uint32_t w = threadIdx.x; w < L::CMAPW; w += blockDim.x) cmapw[w] = p.byte_class[w];
  if (FOLD)
    for (uint32_t w = threadIdx.x; w < pin_words; w += blockDim.x) lds[L::CMAPW + w] = p.pin_tab[w];
  __syncthreads();  // the only block-wide barrier; the waves never meet again

  const uint32_t wave = blockIdx.x * wpb + wib;
  const uint32_t stream0 = wave * S;
  if (stream0 >= p.n_streams) return;
  const uint32_t n_mine = p.n_streams - stream0 < (uint32_t)S ? p.n_streams - stream0 : (uint32_t)S;
  // Stream slots this wavefront still handles (wave-uniform): a stream whose active set makes the wave-wide list overflow
  // is handed to the wave kernel ALONE (evict, below) and its slot goes idle; the others stay.
  // (n_mine comes from threadIdx.x >> 6, which the compiler cannot know to be the same in all 64 lanes: one readfirstlane
  // here keeps `alive`, and with it the loop conditions below, in SGPRs)
  const uint32_t n_mine_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_mine);
  unsigned long long alive = n_mine_s >= 64u ? ~0ull : (1ull << n_mine_s) - 1ull;
  bool owner = lane < n_mine;  // lane == stream slot it owns (and still handles)
  bool overflow = false;       // the pass that just ran could not hold its next sets: a stream has to leave (scalar; see `evict`)
  // input windows: lane = 4*slot + part fetches bytes [64*chunk + 16*part, +16) of stream `slot`
  static_assert(S >= 1 && S <= (FOLD ? 64 : 32), "five-bit (FOLD: six-bit) stream slot; the window loader covers 16 streams per wave-load");
  constexpr uint32_t NLOAD = (S + 15) / 16;  // wave-loads per refill
  auto load_win = [&](uint32_t chunk, uint32_t (&o)[NLOAD][4]) {
#pragma unroll
    for (uint32_t g = 0; g < NLOAD; g++) {
      const uint32_t slot = g * 16u + (lane >> 2), part = lane & 3u;
      const bool have = slot < n_mine;
      const uint8_t* bp = p.bytes + (size_t)(stream0 + (have ? slot : 0)) * p.stride;
      const uint32_t off = chunk * 64u + part * 16u;
      uint32_t v[4] = {0, 0, 0, 0};
      if (have) {
        if ((reinterpret_cast<uintptr_t>(bp + off) & 15u) == 0 && off + 16u <= p.stream_len) {
          const uint4 q = *reinterpret_cast<const uint4*>(bp + off);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int w4 = 0; w4 < 4; w4++)
            for (uint32_t b = 0; b < 4; b++)
              if (off + 4u * w4 + b < p.stream_len) v[w4] |= (uint32_t)bp[off + 4u * w4 + b] << (8u * b);
        }
      }
#pragma unroll
      for (int w4 = 0; w4 < 4; w4++) o[g][w4] = v[w4];
    }
  };

  for (uint32_t w = lane; w < S * L::STRIDE; w += 64u) sreg0[w] = 0u;
  if (owner) wl[lane] = p.state0_entry | (lane << SID_SHIFT);  // FPGA.v:134-147: current = {state 0}, per stream
  uint32_t N = n_mine_s, Nn = 0;  // (scalar: see n_mine_s)
  uint32_t* clist = wl;              // the two wave-wide lists and the two filter halves swap roles every pass
  uint32_t* nlist = wl + L::LISTW;
  // the two filter halves of a stream region, as BYTE offsets (carried in VGPRs and added with plain v_add: a word offset
  // would be scaled by a three-operand v_lshl_add in every sweep, which issues at half rate)
  uint32_t fcur_b = 0, fnext_b = L::FW * 4u;
  auto fword = [](uint32_t* sreg, uint32_t half_b, uint32_t h) -> uint32_t* {
    return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sreg) + half_b + ((h >> 5) << 2));
  };
  uint32_t nxt[NLOAD][4];
  load_win(0, nxt);
  wave_sync();
  bool spilled = false;
  // A lane without an entry still runs the unpredicated LDS operations of a pass (zero stores into the filter being
  // wiped, OR 0 into the filter being filled): its pseudo entry points every lane at a different (slot, filter word)
  // so that those no-ops do not pile up on one LDS address.
  // (Its state id is `size`: the row behind the last state's in the slice index, all zero — idle lanes gather like
  // everybody else, no EXEC masking, and find nothing; so do accept states, whose rows are empty by definition.)
  const uint32_t e_none = 0x80000000u | ((lane % (uint32_t)S) << SID_SHIFT) | p.size;

  unsigned long long busy = ~0ull;  // FOLD: bit j = pass j of the current window has an emission of the folded state (wave-uniform)
  // window refill at a pass k that is a multiple of 64: bytes -> byte classes on the way into LDS, next window requested
  auto refill = [&](uint32_t k) {
    wave_sync();
#pragma unroll
    for (uint32_t g = 0; g < NLOAD; g++) {
      const uint32_t slot = g * 16u + (lane >> 2), part = lane & 3u;
      if (slot < n_mine) {
        uint32_t* win = sreg0 + slot * L::STRIDE + 2u * L::FW + part * 4u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t v = nxt[g][q];
          win[q] = (uint32_t)cmap[v & 0xFFu] | ((uint32_t)cmap[(v >> 8) & 0xFFu] << 8) |
                   ((uint32_t)cmap[(v >> 16) & 0xFFu] << 16) | ((uint32_t)cmap[v >> 24] << 24);
        }
      }
    }
    if (FOLD) {
      // Which passes of this window can be skipped when the wave's list is empty: those in which no stream's folded state
      // emits anything (on inputs that rarely touch a pattern nearly all of them).  The lane that converted 16 bytes of
      // a stream looks the 16 emissions up; pass 63 needs the next window's first byte and is added at mid-window
      // (stash_next_first).
      if (lane < 2u) busyw[lane] = 0u;
      wave_sync();
#pragma unroll
      for (uint32_t g = 0; g < NLOAD; g++) {
        const uint32_t slot = g * 16u + (lane >> 2), part = lane & 3u;
        const uint32_t* win = sreg0 + (slot < n_mine ? slot : 0u) * L::STRIDE + 2u * L::FW + part * 4u;
        const uint32_t w0 = win[0], w1 = win[1], w2 = win[2], w3 = win[3];
        const uint32_t nf = (uint32_t)__shfl_down((int)(w0 & 0xFFu), 1);  // first class of the next 16 bytes (part < 3)
        const unsigned long long lo = ((unsigned long long)w1 << 32) | w0, hi = ((unsigned long long)w3 << 32) | w2;
        uint32_t bits = 0u;
#pragma unroll
        for (uint32_t i = 0; i < 16u; i++) {
          const uint32_t c0 = (uint32_t)((i < 8u ? lo : hi) >> (8u * (i & 7u))) & 0xFFu;
          const uint32_t c1 = i == 15u ? nf : (uint32_t)((i + 1u < 8u ? lo : hi) >> (8u * ((i + 1u) & 7u))) & 0xFFu;
          const uint32_t kg = k + part * 16u + i;
          const bool last = kg + 1u >= p.n_consume;
          const bool known = kg >= 1u && kg < p.n_consume && (last || !(part == 3u && i == 15u));
          const uint32_t em = known ? pintab[c0 * p.pin_cols + (last ? ncls : c1)] : 0u;
          bits |= (em != 0u ? 1u : 0u) << i;
        }
        if (slot < n_mine && bits != 0u) atomicOr(&busyw[part >> 1], bits << ((part & 1u) * 16u));
      }
    }
    load_win((k >> 6) + 1u, nxt);
    wave_sync();
    if (FOLD) busy = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)busyw[1]) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)busyw[0]);
  };
  // per-pass any-match bits of 256 passes: words [8 * group, 8 * group + 8) of every stream, stored by its owner lane.  The
  // plan pads the rows of its bitmap to a multiple of eight words, so a group is one aligned 32-byte sector.
  auto store_anymatch = [&](uint32_t group) {
    wave_sync();
    if (owner) {
      uint32_t* am = sreg0 + lane * L::STRIDE + 2u * L::FW + L::WINW;
      uint32_t* dst = p.anymatch + (size_t)(stream0 + lane) * p.anymatch_stride + 8u * group;
      if ((p.anymatch_stride & 7u) == 0u) {
        reinterpret_cast<uint4*>(dst)[0] = make_uint4(am[0], am[1], am[2], am[3]);
        reinterpret_cast<uint4*>(dst)[1] = make_uint4(am[4], am[5], am[6], am[7]);
      } else {  // (a bitmap with another pitch: word by word, as far as the row goes)
        for (uint32_t w = 0; w < L::AMW && 8u * group + w < p.anymatch_stride; w++) dst[w] = am[w];
      }
#pragma unroll
      for (uint32_t w = 0; w < L::AMW; w++) am[w] = 0u;
    }
    wave_sync();
  };

  // PRUNE: class of the first byte of the NEXT chunk (already requested, in `nxt`) behind the window, so that the
  // look-ahead also works at the window's last byte; called in the middle of a chunk, when that load has landed
  auto stash_next_first = [&]() {
#pragma unroll
    for (uint32_t g = 0; g < NLOAD; g++) {
      const uint32_t slot = g * 16u + (lane >> 2);
      if ((lane & 3u) == 0u && slot < n_mine) sreg0[slot * L::STRIDE + 2u * L::FW + 16u] = cmap[nxt[g][0] & 0xFFu];
    }
    if (FOLD) {  // pass 63 of the window: its look-ahead class has just arrived
      wave_sync();
      uint32_t em = 0u;
      if (owner) {
        const uint8_t* wb = reinterpret_cast<const uint8_t*>(sreg0 + lane * L::STRIDE + 2u * L::FW);
        em = pintab[(uint32_t)wb[63] * p.pin_cols + (uint32_t)wb[64]];
      }
      if (wballot(em != 0u) != 0ull) busy |= 1ull << 63;
    }
  };

  // One pass (FPGA.v:158-741 for S streams).  `consume` is a compile-time tag: the passes that take an input byte are
  // driven by the chunked loops below (no per-pass refill / bitmap / mode tests); FULL mode's pass N only looks for
  // accept states.
  // `replay_tag`: the pass is being run again after an eviction — its accept pulses (and statistics) are out already.
  auto pass = [&](const uint32_t k, auto consume_tag, auto replay_tag) {
    constexpr bool consume = decltype(consume_tag)::value;
    constexpr bool replay = decltype(replay_tag)::value;
    const uint32_t kk = k & 63u;
    stamp(-1);
    Nn = 0;

    // exact check for a candidate whose filter bit was already set; wave-uniform call
    auto resolve = [&](bool maybe, uint32_t t) {
      uint64_t mm = wballot(maybe);
      while (mm) {
        const uint32_t src = (uint32_t)__builtin_ctzll(mm);
        mm &= mm - 1;
        const uint32_t key = bcast(t, src) & KEY_MASK;
        const uint32_t lim = Nn < L::CAPW ? Nn : L::CAPW;
        bool found = false;
        for (uint32_t q = lane; q < lim; q += 64u) found |= (nlist[q] & KEY_MASK) == key;
        if (wballot(found) == 0) {
          if (lane == src && Nn < L::CAPW) nlist[Nn] = t;
          Nn += 1;
          wave_sync();
        }
      }
    };
    // one candidate per lane into the wave-wide next list; wave-uniform call
    auto insert = [&](bool pred, uint32_t t, uint32_t* sreg) {
      const uint32_t h = t & HMASK;
      const uint32_t bit = 1u << (h & 31u);
      uint32_t old = 0;
      if (pred) old = atomicOr(fword(sreg, fnext_b, h), bit);
      const bool fresh = pred && (old & bit) == 0;
      const bool maybe = pred && (old & bit) != 0;
      const uint64_t mf = wballot(fresh);
      const uint32_t slot = Nn + rank_below(mf);
      if (fresh && slot < L::CAPW) nlist[slot] = t;
      Nn += (uint32_t)__popcll(mf);
      if (wballot(maybe)) {
        wave_sync();
        resolve(maybe, t);
      }
    };

    // FOLD, part 1: every stream from reset holds the pinned state from pass 1 on.  Its owner lane looks up what that
    // state's row emits on this byte, already reduced to the targets that survive the NEXT byte (full slice at the
    // stream's last byte, whose sets are reported).  Two dependent LDS reads that do not depend on the list: they are
    // in flight while the sweep below reads its entries, and the insertion (part 2) runs in the shadow of the gather.
    // PRUNE: bits 23:16 forced to ones where inline targets are not to be pruned (wide index; the stream's last byte)
    const uint32_t keep_all = (PRUNE && narrow && k + 1u < p.n_consume) ? 0u : 0x00FF0000u;
    uint32_t vA = 0u;
    const bool pin_now = FOLD && consume && k >= 1u;  // wave-uniform
    if (pin_now && owner) {
      const uint8_t* wb = reinterpret_cast<const uint8_t*>(sreg0 + lane * L::STRIDE + 2u * L::FW);
      const uint32_t c0 = wb[kk];
      const uint32_t sel = (k + 1u < p.n_consume) ? (uint32_t)wb[kk + 1u] : ncls_v;  // byte 64 of the window: stash
      vA = pintab[c0 * p.pin_cols + sel];
    }
    bool pin_done = !pin_now;
    auto pin_stage = [&]() {  // FOLD, part 2; wave-uniform call
      pin_done = true;
      uint32_t* oreg = sreg0 + lane * L::STRIDE;  // (lanes that own no stream pass pred = false)
      if (wballot(vA & RXE_INLINE))
        insert((vA & RXE_INLINE) != 0u, (vA & (RXE_TGT_MASK | RXE_ACCEPT)) | (lane << SID_SHIFT), oreg);
      uint64_t mo = wballot(vA & RXE_OVF);  // several pattern heads on this byte survive the next one: rare
      while (mo) {
        const uint32_t src = (uint32_t)__builtin_ctzll(mo);
        mo &= mo - 1;
        const uint32_t off = bcast(vA & RXE_TGT_MASK, src);
        const uint32_t cnt = ovf[off];
        uint32_t* sr = sreg0 + src * L::STRIDE;
        for (uint32_t q0 = 0; q0 < cnt; q0 += 64u) {
          const bool act = q0 + lane < cnt;
          const uint32_t w = act ? ovf[off + 1u + q0 + lane] : 0u;
          insert(act && !(w & RXE_PIN), (w & (RXE_TGT_MASK | RXE_ACCEPT)) | (src << SID_SHIFT), sr);
        }
      }
    };

    // Predicates are kept as integer tests taken right at the ballot (one v_and + v_cmp each): a bool assembled from
    // several flags reaches __ballot through a VGPR (v_cndmask + v_cmp), and this loop is bound by instruction issue.
    constexpr uint32_t E_NONE = 0x80000000u;  // list-entry flag of a lane without an entry (bit 31 is otherwise unused)
    const uint32_t Ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)N);  // wave-uniform: keep the loop scalar
    if ((FOLD || PRUNE) && consume) fold_entries += Ns;  // (scalar) what AUTO's probe reads: list entries left per stream-byte
    const uint32_t* cp = clist + lane;  // this lane's entry of the sweep (a carried address: no index arithmetic per sweep)
    for (uint32_t b0 = 0; b0 < Ns; b0 += 64u, cp += 64) {
      const bool have = lane < Ns - b0;  // (scalar subtraction, one v_cmp)
      uint32_t e = *cp;  // lanes past N read harmless LDS words of this wave and are overwritten below
      if (!have) e = e_none;
      if (PROF) { asm volatile("" ::"v"(e)); stamp(0); }  // phase 0: refill check + list read
      const uint32_t sid = (e >> SID_SHIFT) & SID_BITS;
      const uint32_t s = e & RXE_TGT_MASK;
      uint32_t* sreg = sreg0 + sid * L::STRIDE;
      // accept pulses of the entries that are accept states (FPGA.v:210-226); a lane without an entry never has the flag
      auto accept_pulses = [&]() {
        if (wballot(e & RXE_ACCEPT) != 0ull && !replay) {
          const bool acc = (e & RXE_ACCEPT) != 0u;
          uint32_t dummy = 0;
          emit_events(p, acc, s, stream0 + sid, k, lane, dummy);
          if (acc) atomicOr(&sreg[2u * L::FW + L::WINW + ((k >> 5) & (L::AMW - 1u))], 1u << (k & 31u));
        }
      };
      if (!consume) {  // RX_MODE_FULL's last pass: nothing but the pulses
        accept_pulses();
        continue;
      }
      const bool live = (e & (E_NONE | RXE_ACCEPT)) == 0u;  // a real entry that is not an accept state: it has a row (statistics only)
      const uint32_t c = reinterpret_cast<const uint8_t*>(sreg + 2u * L::FW)[kk];  // class of that stream's input_char
      uint32_t cnx = 0u;  // PRUNE: class of its NEXT byte (byte 64 of the window: the stash)
      if (PRUNE) cnx = reinterpret_cast<const uint8_t*>(sreg + 2u * L::FW)[kk + 1u];
      // zero the filter word this entry went through (lanes without an entry hit some word of the CURRENT filter of
      // a valid slot; that filter is being wiped this pass anyway and is not read before the next swap)
      if (!RX_AB_PREDICATE_IDLE || have) *fword(sreg, fcur_b, s & HMASK) = 0u;
      if (STATS && !replay && (e & E_NONE) == 0u) {
        const uint32_t deg = rp[s + 1] - rp[s];
        st_active += 1;
        st_edges += deg;
        if (p.pair_cycles) {
          // Blk_Mem_tb scans both streams of a pair in lock-step (FPGA.v:158): a state active in either
          // stream costs its clocks once.  Streams 2q / 2q+1 of the batch are such a pair; the odd
          // stream skips states the even stream also holds.  cost(i) per SURVEY.md §3.2.
          bool dup = false;
          if (sid & 1u) {
            const uint32_t key = (e ^ (1u << SID_SHIFT)) & KEY_MASK;
            for (uint32_t q = 0; q < N; q++) dup |= (clist[q] & KEY_MASK) == key;
          }
          if (!dup) {
            const uint32_t a = p.size + 1u + rp[s];
            const uint32_t nlines = ((a + deg - 1u) >> 2) - (a >> 2) + 1u;
            st_cost += 3u + ((s & 3u) == 3u ? 1u : 0u) + (deg == 0 ? 1u : nlines + 2u) + 1u - 1u;
          }
        }
      }
      if (PROF) { asm volatile("" ::"v"(c)); stamp(1); }  // phase 1: accept check, window byte, filter clear
      // current byte's slice of row s; 32-bit byte offset from a scalar base (table < 4 GiB) keeps the address
      // arithmetic out of the 64-bit VALU path
      // s < 2^24, ncls <= 256: 24-bit multiply; every lane gathers (accept states and idle lanes read empty rows)
      const uint32_t x = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(symidx) + ((__umul24(s, ncls_v) + c) << 2));
      if (STATS && !replay) {
        if (x & RXE_OVF) st_ovf += 1;
        if ((e & MARK) && live && x == 0u) st_dead += 1;  // came out of a multi-target row and dies at once
      }
      const uint32_t e_keep = STATS ? e & ~MARK : e;
      if (FOLD && !pin_done) {  // the gather is in flight: insert what the folded state emits meanwhile
        __builtin_amdgcn_sched_barrier(0);
        pin_stage();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (PROF) { asm volatile("" ::"v"(x)); stamp(2); }  // phase 2: slice gather
      // two candidates per lane: the state itself (self-loop) and the inline target.  Both filter atomics are issued
      // by every lane, back to back, with one wait: a lane without a candidate ORs 0 (a no-op) into the word its
      // hash names anyway instead of sitting out in a branch or selecting another address.
      constexpr uint32_t T1_MASK = RXE_TGT_MASK | RXE_ACCEPT;
      const uint32_t t1 = PRUNE ? (x & (xtmask | RXE_ACCEPT)) | (e_keep & ~T1_MASK)
                                : (x & T1_MASK) | (e_keep & ~T1_MASK);  // v_bfi: a live e has only its slot bits outside the mask
      const uint32_t h0 = e & HMASK, h1 = x & HMASK;
      // bit to set, 0 = no candidate: flag bit shifted down to 0/1, then up by the hash (two plain shifts; a select on the
      // flag would go through v_bfe_i32 + v_and, and three-operand VALU forms issue at half rate — DESIGN.md 3.4)
      const uint32_t v0 = ((x >> 29) & 1u) << (h0 & 31u);
      static_assert(RXE_SELF == (1u << 29) && RXE_INLINE == (1u << 31), "flag positions used as shift counts");
      // (FOLD: a target that IS the folded state is dropped — the stream holds it anyway)
      // (PRUNE, narrow index: an inline target that is no accept state and has no edge on the stream's next byte can neither
      // pulse nor produce a successor: it is not inserted — except at the stream's last byte, whose sets are reported)
      bool inl = FOLD ? (x & (RXE_INLINE | RXE_PIN)) == RXE_INLINE : (x & RXE_INLINE) != 0u;
      if (PRUNE) inl = inl && (((x | keep_all) >> (16u + (cnx & 7u))) & 1u) != 0u;
      const uint32_t v1 = (!PRUNE && !FOLD) ? (x >> 31) << (h1 & 31u) : (inl ? 1u << (h1 & 31u) : 0u);
      uint32_t o0 = 0u, o1 = 0u;
      if (!RX_AB_PREDICATE_IDLE || v0) o0 = atomicOr(fword(sreg, fnext_b, h0), v0);
      if (!RX_AB_PREDICATE_IDLE || v1) o1 = atomicOr(fword(sreg, fnext_b, h1), v1);
      __builtin_amdgcn_sched_barrier(0);  // keep the first result's consumers behind the second atomic's issue
      if (PROF) { asm volatile("" ::"v"(o0), "v"(o1)); stamp(3); }  // phase 3: the two filter atomics
      // fresh: candidate whose bit was clear; maybe: candidate whose bit was already set
      // (v is one bit or nothing, d = the part of it that was already set: fresh <=> v != d, and d is needed below anyway)
      const uint32_t d0 = v0 & o0, d1 = v1 & o1;
      const uint64_t mf0 = wballot(v0 != d0), mf1 = wballot(v1 != d1);
      if (__builtin_expect(Nn <= L::CAPW, 1)) {  // (wave-uniform) past that the pass ends in a hand-off anyway; keeps writes inside LISTW
        if (v0 != d0) nlist[rank_below_plus(mf0, Nn)] = e_keep;
        if (v1 != d1) nlist[rank_below_plus(mf1, Nn + (uint32_t)__popcll(mf0))] = t1;
      }
      Nn += (uint32_t)__popcll(mf0) + (uint32_t)__popcll(mf1);
      // Everything that is rare — an accept state among the entries, a candidate whose filter bit was already set, a row
      // with several targets on the byte — hides behind ONE test (a v_cmp that writes a lane mask and the branch on it cost
      // as much as four plain VALU instructions each: tools/issue_bench).
      // (PRUNE builds serve automata that meet multi-target rows in nearly every sweep: there the pre-test is one ballot too many)
      const uint32_t dup_bits = d0 | d1;
      if (!PRUNE && __builtin_expect(wballot(((e & RXE_ACCEPT) | (x & RXE_OVF) | dup_bits) != 0u) == 0ull, 1)) {
        stamp(4);
        continue;
      }
      accept_pulses();
      if (wballot(dup_bits != 0u) != 0ull) {
        wave_sync();
        resolve(d0 != 0u, e_keep);
        resolve(d1 != 0u, t1);
      }
      stamp(4);  // phase 4: ballots, slots, list writes, rare duplicate resolution
      // rows with several targets on this byte (rare on snort_16, every pass on l7 and on compiled rule sets):
      // the target lists of as many such entries as fit are laid side by side over the 64 lanes (a scalar walk
      // hands every entry its lane range), then ONE sweep loads and inserts them all
      uint64_t mo = wballot(x & RXE_OVF);
      if (__builtin_expect(mo != 0, 0)) {
        uint32_t myoff = x & RXE_TGT_MASK, mycnt = 0u;  // this lane's list: ovf[myoff] = count, targets behind it
        if (prune) {
          // directory entry for the class of the stream's next byte; the full list when the sets built now are the
          // ones reported (the stream's last byte)
          // (ncls_v, the copy in a VGPR: the scalar one does not survive this loop's register pressure — the compiler re-reads
          // it from the kernel-argument segment, an s_load and its wait in every pass of every workload with multi-target rows:
          // SQ_INSTS_SMEM 1e5 -> 5.9e6 per launch on the rule-set stand-in; taking it out did not move that launch's time, though)
          uint32_t sel = ncls_v;
          if (k + 1u < p.n_consume) sel = reinterpret_cast<const uint8_t*>(sreg + 2u * L::FW)[kk + 1u];  // byte 64: stash
          const uint32_t d = (x & RXE_OVF) ? p.ovf_dir[(x & xtmask) * (ncls_v + 1u) + sel] : 0u;
          myoff = d >> 8;
          mycnt = d & 255u;
          if (mycnt == 255u) mycnt = ovf[myoff];
        } else if (x & RXE_OVF) {
          mycnt = ovf[myoff];
        }
        // lists of one (what pruning leaves of most multi-target rows): every such lane inserts its single target
        // itself, all of them in one step; only longer lists go through the lane-range walk below
        if (wballot(mycnt == 1u)) {
          const bool one = mycnt == 1u;
          const uint32_t w1 = one ? ovf[myoff + 1u] : 0u;
          insert(one && !(FOLD && (w1 & RXE_PIN)), (w1 & (RXE_TGT_MASK | RXE_ACCEPT)) | (sid << SID_SHIFT) | (STATS ? MARK : 0u), sreg);
        }
        mo = wballot(mycnt >= 2u);
        uint32_t total = 0, my_at = 0, my_sid = 0;
        auto flush = [&]() {
          const bool act = lane < total;
          const uint32_t w = act ? ovf[my_at] : 0u;
          insert(act && !(FOLD && (w & RXE_PIN)), (w & (RXE_TGT_MASK | RXE_ACCEPT)) | (my_sid << SID_SHIFT) | (STATS ? MARK : 0u), sreg0 + my_sid * L::STRIDE);
          total = 0;
        };
        while (mo) {
          const uint32_t src = (uint32_t)__builtin_ctzll(mo);
          mo &= mo - 1;
          const uint32_t off = bcast(myoff, src);
          const uint32_t osid = bcast(sid, src);
          const uint32_t cnt = bcast(mycnt, src);
          if (total != 0 && total + cnt > 64u) flush();
          if (cnt > 64u) {  // a list longer than the wave: swept on its own
            uint32_t* oreg = sreg0 + osid * L::STRIDE;
            for (uint32_t q0 = 0; q0 < cnt; q0 += 64u) {
              const bool act = q0 + lane < cnt;
              const uint32_t w = act ? ovf[off + 1u + q0 + lane] : 0u;
              insert(act && !(FOLD && (w & RXE_PIN)), (w & (RXE_TGT_MASK | RXE_ACCEPT)) | (osid << SID_SHIFT) | (STATS ? MARK : 0u), oreg);
            }
          } else {
            const uint32_t d = lane - total;  // lanes [total, total+cnt) take this entry's targets
            if (d < cnt) { my_at = off + 1u + d; my_sid = osid; }
            total += cnt;
          }
        }
        if (total != 0) flush();
      }
    }

    if (FOLD && !pin_done) pin_stage();  // no list entry this pass: the folded state is all there is
    stamp(5);  // phase 5: overflow lists + loop control
    if (consume) {
      if (__builtin_expect(Nn > L::CAPW, 0)) {
        overflow = true;  // (handled by the caller, outside the loop of passes: `evict`, then this pass again)
      } else {
        {  // current <- next (FPGA.v:733-737)
          uint32_t* t = clist; clist = nlist; nlist = t;
          const uint32_t f = fcur_b; fcur_b = fnext_b; fnext_b = f;
        }
        N = Nn;
        wave_sync();
      }
    }
    stamp(6);  // phase 6: end of pass (swap, wave sync)
  };

  // Called when a pass has left `overflow` set (wave-uniform; kept out of the loop of passes — inside the pass it cost every
  // pass of every workload 1.8 %: same-box build without it, tools/r3_ab8.sh).
  auto evict = [&](const uint32_t k) {
    // The wave-wide list cannot hold the next sets.  The streams with the most entries in the part of the next list that
    // was written leave — one if that brings the written part under 5/8 of the capacity (a single stream that explodes),
    // more if the load is spread (then one at a time would overflow again a few passes on) — each with its S_k (its
    // entries of the current list, still intact) and k, to be finished by the wave kernel; their slots go idle.  The other
    // streams run this pass again (replay: their accept pulses of pass k are out already, as are the leaving streams',
    // which the wave kernel therefore skips at k).
    const RxColdParams cq = cold_params();
    wave_sync();
    uint32_t c_next = 0;
    const uint32_t lim = Nn < L::CAPW ? Nn : L::CAPW;
    for (uint32_t q = 0; q < lim; q++) c_next += ((nlist[q] >> SID_SHIFT) & SID_BITS) == lane ? 1u : 0u;
    unsigned long long victims = 0ull;  // (scalar)
    uint32_t left = lim;
    do {
      // most entries, lowest slot on ties; > 0 for every live slot that has not been picked yet
      uint32_t key = (owner && ((victims >> lane) & 1ull) == 0ull) ? ((c_next + 1u) << 6) | (63u - lane) : 0u;
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)key, d);
        key = o > key ? o : key;
      }
      // (every lane holds the maximum; through an SGPR so that everything derived from it — the set of live slots, the
      // loop conditions — stays scalar for the compiler too)
      const uint32_t best = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
      if (best == 0u) break;
      victims |= 1ull << (63u - (best & 63u));
      left -= (best >> 6) - 1u;
    } while (left > (L::CAPW * 5u) / 8u);
    const uint32_t n_vict = (uint32_t)__popcll(victims);
    unsigned long long b = 0;
    if (lane == 0) b = atomicAdd(cq->spill_count, (unsigned long long)n_vict);
    const uint32_t slot_base = bcast((uint32_t)b, 0);
    const bool i_leave = ((victims >> lane) & 1ull) != 0ull;  // (lane == stream slot)
    const uint32_t my_slot = slot_base + (uint32_t)__popcll(victims & ((1ull << lane) - 1ull));
    if (lane < (uint32_t)S) slotw[lane] = my_slot;
    {
      uint32_t* rows = cq->spill_rows + (size_t)slot_base * cq->nw64x2;
      for (uint32_t w = lane; w < n_vict * cq->nw64x2; w += 64u) rows[w] = 0u;
    }
    if (i_leave) {
      cq->spill_streams[my_slot] = stream0 + lane;
      cq->spill_k[my_slot] = k;
      if (cq->anymatch) {  // the words of the current 256-pass group up to the one of pass k (which the wave kernel reads back)
        const uint32_t* am = sreg0 + lane * L::STRIDE + 2u * L::FW + L::WINW;
        uint32_t* dst = cq->anymatch + (size_t)(stream0 + lane) * cq->anymatch_stride + ((k >> 8) << 3);
        for (uint32_t w = 0; w <= ((k >> 5) & (L::AMW - 1u)); w++) dst[w] = am[w];
      }
    }
    __threadfence();
    wave_sync();
    // S_k of the leaving streams into their hand-off rows; the list without them into the other buffer
    uint32_t M = 0;
    for (uint32_t b0 = 0; b0 < N; b0 += 64u) {
      const uint32_t li = b0 + lane;
      const uint32_t e = li < N ? clist[li] : 0u;
      const uint32_t es = (e >> SID_SHIFT) & SID_BITS;
      const bool goes = li < N && ((victims >> es) & 1ull) != 0ull;
      if (goes) {
        const uint32_t sq = e & RXE_TGT_MASK;
        atomicOr(&cq->spill_rows[(size_t)slotw[es] * cq->nw64x2 + (sq >> 5)], 1u << (sq & 31u));
      }
      const bool keep = li < N && !goes;
      const uint64_t mk = wballot(keep);
      if (keep) nlist[rank_below_plus(mk, M)] = e;
      M += (uint32_t)__popcll(mk);
    }
    if (FOLD && k >= 1u && i_leave)  // S_k holds the folded state
      atomicOr(&cq->spill_rows[(size_t)my_slot * cq->nw64x2 + (cq->pin_state >> 5)], 1u << (cq->pin_state & 31u));
    {
      uint32_t* t = clist; clist = nlist; nlist = t;
    }
    N = M;
    // both filters of every slot start clean (bits of entries that were counted but not written would otherwise stay)
    for (uint32_t w = lane; w < (uint32_t)S * 2u * L::FW; w += 64u) sreg0[(w / (2u * L::FW)) * L::STRIDE + (w % (2u * L::FW))] = 0u;
    alive &= ~victims;
    owner = lane < n_mine && ((alive >> lane) & 1ull) != 0ull;
    if (alive == 0ull) spilled = true;  // nothing left here
    wave_sync();
  };

  uint32_t k = 0;
  const uint32_t n_consume = p.n_consume < p.n_passes ? p.n_consume : p.n_passes;
  while (k < n_consume && !spilled) {  // k is a multiple of 64 here
    refill(k);
    if (FOLD && k == 0u) {
      // FOLD builds serve inputs on which most streams end without a list entry: their final rows are all zero but for the
      // folded state's bit.  The rows are cleared HERE — behind the wait for the first window (on gfx9 stores count in
      // `vmcnt`: issued before it they would sit in front of the first bytes), in the shadow of the first passes — and at the
      // end such a stream stores one 8-byte granule; written behind the last pass of every wavefront at once, the zero rows
      // were 16 us of pure HBM time, 7 % of a launch on uniform bytes.  (A stream that does end with entries stores its whole
      // row again, as in the other builds.)
      const RxColdParams cz = cold_params();
      if (cz->final_active && !cz->fin_states) {
        uint32_t* rows = cz->final_active + (size_t)stream0 * cz->nw64x2;
        const uint32_t words = n_mine * cz->nw64x2;
        for (uint32_t w = lane; w < words; w += 64u) rows[w] = 0u;
      }
    }
    const uint32_t kend = n_consume - k < 64u ? n_consume : k + 64u;
    while (k < kend && !spilled) {
      const uint32_t k32 = kend - k < 32u ? kend : k + 32u;
      do {
        if (FOLD && __builtin_amdgcn_readfirstlane((int)N) == 0) {
          // nothing in the wave's list: the passes up to the next emission of a folded state change nothing at all
          // (both lists empty, both filters clean, no pulse) — skip them in one step
          const unsigned long long rest = busy >> (k & 63u);
          if ((rest & 1ull) == 0ull) {
            const uint32_t skip = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
            k += skip < k32 - k ? skip : k32 - k;
            continue;
          }
        }
        pass(k, std::true_type{}, std::false_type{});
        while (__builtin_expect(overflow, 0)) {  // streams leave until the pass fits; it is run again for those that stay
          overflow = false;
          evict(k);
          if (spilled) break;
          // (FOLD: nothing left in the list and no folded state emits at k — the pass is a no-op for the streams that stayed)
          if (FOLD && __builtin_amdgcn_readfirstlane((int)N) == 0 && ((busy >> (k & 63u)) & 1ull) == 0ull) break;
          pass(k, std::true_type{}, std::true_type{});
        }
        k++;
      } while (k < k32 && !spilled);
      if (!spilled && p.anymatch && (k & 255u) == 0u) store_anymatch((k >> 8) - 1u);
      if (LOOK && (k & 63u) == 32u) stash_next_first();
    }
  }
  while (k < p.n_passes && !spilled) {  // RX_MODE_FULL: pass N
    pass(k, std::false_type{}, std::false_type{});
    k++;
    if (p.anymatch && (k & 255u) == 0u) store_anymatch((k >> 8) - 1u);
  }
  if (!spilled && p.anymatch && (k & 255u) != 0u) store_anymatch(k >> 8);
  if (PROF && lane == 0) {
    for (int q = 0; q < 7; q++) atomicAdd(&p.counters[8 + q], t_sum[q]);
    // wave 0 only: [63:32] shader cycles / 64, [31:0] 100 MHz ticks, both over the whole wave
    if (wave == 0) p.counters[15] = (((__builtin_amdgcn_s_memtime() - t0c) >> 6) << 32) | ((__builtin_amdgcn_s_memrealtime() - t0r) & 0xFFFFFFFFull);
  }
  if ((FOLD || PRUNE) && lane == 0 && fold_entries) atomicAdd(&p.counters[7], fold_entries);
  // Final active sets of the streams that stayed (FPGA.v:733-737: the set that survives the last byte).  The next-list
  // buffer is dead now and serves as scratch.
  const bool pin_in = FOLD && n_consume >= 1u;  // the folded state is in every set after the first byte
  const RxColdParams cq = cold_params();
  if (cq->fin_states && !spilled) {
    // As compact lists, straight from the list entries — no bitmask row is ever built.  Per entry its rank among its
    // stream's entries (ascending state), per stream (owner lane) the count and, FOLD, where the folded state goes; one
    // atomic per wavefront for the space.
    wave_sync();
    uint32_t my_cnt = 0, pin_rank = 0;
    for (uint32_t q = 0; q < N; q++) {
      const uint32_t eq = clist[q];
      const uint32_t sq = eq & RXE_TGT_MASK;
      if (((eq >> SID_SHIFT) & SID_BITS) == lane && !(pin_in && sq == cq->pin_state)) {
        my_cnt++;
        if (pin_in && sq < cq->pin_state) pin_rank++;
      }
    }
    if (!owner) my_cnt = 0;
    else if (pin_in) my_cnt++;
    uint32_t incl = my_cnt;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= (uint32_t)d) incl += o;
    }
    const uint32_t total = bcast(incl, 63);
    unsigned long long base = 0;
    if (lane == 0 && total) base = atomicAdd(cq->fin_count, (unsigned long long)total);
    base = ((unsigned long long)bcast((uint32_t)(base >> 32), 0) << 32) | bcast((uint32_t)base, 0);
    const unsigned long long my_off64 = base + (incl - my_cnt);
    const uint32_t my_off = (uint32_t)(my_off64 < cq->fin_cap ? my_off64 : cq->fin_cap);
    if (owner) {
      cq->fin_off[stream0 + lane] = my_off;
      cq->fin_cnt[stream0 + lane] = my_cnt;
      if (pin_in && my_off + pin_rank < cq->fin_cap) cq->fin_states[my_off + pin_rank] = cq->pin_state;
    }
    if (lane < (uint32_t)S) slotw[lane] = my_off;
    wave_sync();
    for (uint32_t b0 = 0; b0 < N; b0 += 64u) {
      const uint32_t li = b0 + lane;
      const uint32_t e = li < N ? clist[li] : 0u;
      const uint32_t sid = (e >> SID_SHIFT) & SID_BITS, sq = e & RXE_TGT_MASK;
      const bool valid = li < N && !(pin_in && sq == cq->pin_state);
      uint32_t rank = (pin_in && sq > cq->pin_state) ? 1u : 0u;
      for (uint32_t q = 0; q < N; q++) {
        const uint32_t eq = clist[q];
        rank += (((eq >> SID_SHIFT) & SID_BITS) == sid && (eq & RXE_TGT_MASK) < sq) ? 1u : 0u;
      }
      if (valid) {
        const uint32_t o = slotw[sid] + rank;
        if (o < cq->fin_cap) cq->fin_states[o] = sq;
      }
    }
  } else if (cq->final_active && !spilled) {
    // As bitmask rows: each row is built in LDS (in slices of the scratch buffer's size for automata whose row is longer)
    // and stored ONCE with 8-byte stores — no zeroing in the prologue, no global atomics.
    constexpr uint32_t SLICE = L::LISTW & ~1u;
    // which streams have list entries at all (wave-uniform): the rows of the others are zero but for the folded state's bit and
    // are stored without the detour through LDS (inputs that rarely touch a pattern: nearly all of them)
    if (lane < 2u) slotw[lane] = 0u;
    wave_sync();
    for (uint32_t li = lane; li < N; li += 64u) {
      const uint32_t es = (clist[li] >> SID_SHIFT) & SID_BITS;
      atomicOr(&slotw[es >> 5], 1u << (es & 31u));
    }
    wave_sync();
    const unsigned long long has = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)slotw[S > 32 ? 1 : 0]) << 32) * (S > 32 ? 1ull : 0ull) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)slotw[0]);
    const uint32_t pin_w = pin_in ? cq->pin_state >> 5 : 0xFFFFFFFFu, pin_b = pin_in ? 1u << (cq->pin_state & 31u) : 0u;
    for (uint32_t sl = 0; sl < n_mine; sl++) {
      if (((alive >> sl) & 1ull) == 0ull) continue;  // (wave-uniform) finished by the wave kernel
      if (((has >> sl) & 1ull) == 0ull) {
        uint2* row8 = reinterpret_cast<uint2*>(cq->final_active + (size_t)(stream0 + sl) * cq->nw64x2);
        if (FOLD && n_consume >= 1u) {  // cleared behind the first window (above): only the folded state's granule is left
          if (pin_in && lane == 0) row8[pin_w >> 1] = make_uint2((pin_w & 1u) ? 0u : pin_b, (pin_w & 1u) ? pin_b : 0u);
          continue;
        }
        for (uint32_t w = lane; w < cq->nw64x2 / 2u; w += 64u)
          row8[w] = make_uint2(2u * w == pin_w ? pin_b : 0u, 2u * w + 1u == pin_w ? pin_b : 0u);
        continue;
      }
      for (uint32_t w0 = 0; w0 < cq->nw64x2; w0 += SLICE) {
        const uint32_t nwords = cq->nw64x2 - w0 < SLICE ? cq->nw64x2 - w0 : SLICE;
        wave_sync();
        for (uint32_t w = lane; w < nwords; w += 64u) nlist[w] = 0u;
        wave_sync();
        for (uint32_t li = lane; li < N; li += 64u) {
          const uint32_t e = clist[li];
          const uint32_t wd = (e & RXE_TGT_MASK) >> 5;
          if (((e >> SID_SHIFT) & SID_BITS) == sl && wd >= w0 && wd - w0 < nwords) atomicOr(&nlist[wd - w0], 1u << (e & 31u));
        }
        if (pin_in && lane == 0) {
          const uint32_t wd = cq->pin_state >> 5;
          if (wd >= w0 && wd - w0 < nwords) atomicOr(&nlist[wd - w0], 1u << (cq->pin_state & 31u));
        }
        wave_sync();
        uint2* row8 = reinterpret_cast<uint2*>(cq->final_active + (size_t)(stream0 + sl) * cq->nw64x2 + w0);
        for (uint32_t w = lane; w < nwords / 2u; w += 64u) row8[w] = make_uint2(nlist[2u * w], nlist[2u * w + 1u]);
      }
    }
  }
  if (STATS) {
    if (st_active) atomicAdd(&p.counters[1], st_active);
    if (st_edges) atomicAdd(&p.counters[2], st_edges);
    if (st_cost) atomicAdd(&p.counters[4], st_cost);
    if (st_ovf) atomicAdd(&p.counters[5], st_ovf);    // entries that met a multi-target row
    if (st_dead) atomicAdd(&p.counters[6], st_dead);  // entries out of such rows that died on the next byte
    if (st_active) atomicAdd(&p.counters[7], st_active);  // this kernel's share of counters[1]
  }
}

// =================================================================================================
// Kernel 5: lazy DFA — one LANE per stream, one table lookup per input byte
// =================================================================================================
// Every NFA kernel above pays, per stream and byte, for each active state separately.  Here a stream carries ONE
// id that names its whole active SET (subset construction done lazily, as RE2/Hyperscan do on CPUs): a pass is
//     id <- dfa_trans[id][class(byte)]
// for 64 streams per wavefront.  The table lives in HBM per automaton and device and persists across launches.
// A zero entry means "not built yet": the wavefront then builds that transition cooperatively — the members of the
// set (one per lane) gather their slice dwords, the targets are OR-ed into an LDS bitmask (dedup), the bitmask is
// enumerated in ascending order (canonical form), the set is looked up / inserted in a global hash table, and the
// transition is published.  Publication is monotonic (0 -> value), every state's member chunk is written and
// released (agent scope) before its id appears anywhere, chunks are whole 128-byte lines that nobody can have read
// before, and member reads bypass L1 — so a stale view can only cause duplicate work, never a wrong set.
// Sets with more than DFA_MAXM members, or a full table, yield DFA_EXIT: that stream is handed to the wave kernel
// (resume mode) like in the group / pack kernels.  Accept pulses: the transition value carries a flag when the
// target set contains accept states; such streams read the member chunk in the next pass and emit the pulses.
__device__ __forceinline__ uint32_t aload(const uint32_t* ptr) {  // agent-scope load: bypasses this CU's L1
  return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct DfaWave {
  uint32_t* bits;   // [nw32] LDS bitmask, all zero between uses
  uint32_t* mlist;  // [64] LDS: sorted members of the set under construction
};

// Build dfa_trans[id][cc]; whole wave, wave-uniform arguments and result.
__device__ uint32_t dfa_build(const RxParams& p, const DfaWave& w, uint32_t id, uint32_t cc, uint32_t lane) {
  {  // another wave may have built it since this wave's (possibly stale) look-up: re-read past the L1
    const uint32_t now = aload(p.dfa_trans + (size_t)id * p.n_classes + cc);
    if (now != 0u) return now;
  }
  const uint32_t ncls = p.n_classes;
  const uint32_t* chunk = p.dfa_pool + (size_t)id * 32u;
  const uint32_t n = aload(chunk);
  // 1. targets of every member on this byte class -> bitmask
  for (uint32_t q0 = 0; q0 < n; q0 += 64u) {
    const uint32_t q = q0 + lane;
    const bool valid = q < n;
    const uint32_t e = valid ? aload(chunk + DFA_HDR_WORDS + q) : 0u;
    const uint32_t s = e & RXE_TGT_MASK;
    const uint32_t x = (valid && !(e & RXE_ACCEPT)) ? p.symidx_c[s * ncls + cc] : 0u;
    if (x & RXE_SELF) atomicOr(&w.bits[s >> 5], 1u << (s & 31u));
    if (x & RXE_INLINE) { const uint32_t t = x & RXE_TGT_MASK; atomicOr(&w.bits[t >> 5], 1u << (t & 31u)); }
    uint64_t mo = wballot(x & RXE_OVF);
    while (mo) {
      const uint32_t src = (uint32_t)__builtin_ctzll(mo);
      mo &= mo - 1;
      const uint32_t off = bcast(x & RXE_TGT_MASK, src);
      const uint32_t cnt = p.ovf[off];
      for (uint32_t j0 = 0; j0 < cnt; j0 += 64u)
        if (j0 + lane < cnt) { const uint32_t t = p.ovf[off + 1u + j0 + lane] & RXE_TGT_MASK; atomicOr(&w.bits[t >> 5], 1u << (t & 31u)); }
    }
  }
  wave_sync();
  // 2. enumerate the bitmask in ascending order (canonical member list), wiping it on the way
  uint32_t cnt = 0, sumdeg = 0, hasacc = 0;
  for (uint32_t w0 = 0; w0 < p.nw32; w0 += 64u) {
    const uint32_t wi = w0 + lane;
    uint32_t word = 0;
    if (wi < p.nw32) { word = w.bits[wi]; w.bits[wi] = 0u; }
    uint32_t pc = (uint32_t)__popc(word), incl = pc;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {  // inclusive prefix sum over the lanes
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= d) incl += up;
    }
    uint32_t at = cnt + incl - pc;
    const uint32_t accw = word ? p.accept_bits[wi] : 0u;
    while (word) {
      const uint32_t b = (uint32_t)__builtin_ctz(word);
      word &= word - 1u;
      const uint32_t st = wi * 32u + b;
      const bool isacc = (accw >> b) & 1u;
      if (at < 64u) w.mlist[at] = st | (isacc ? RXE_ACCEPT : 0u);
      at++;
      sumdeg += p.words[st + 1] - p.words[st];
      hasacc |= isacc ? 1u : 0u;
    }
    cnt += bcast(incl, 63);
  }
  wave_sync();
#pragma unroll
  for (uint32_t d = 32; d >= 1; d >>= 1) {  // wave totals
    sumdeg += (uint32_t)__shfl_xor((int)sumdeg, d);
    hasacc |= (uint32_t)__shfl_xor((int)hasacc, d);
  }
  uint32_t nv;
  if (cnt > DFA_MAXM) {
    nv = DFA_EXIT;
  } else {
    // 3. hash of the member list
    uint32_t hsh = lane < cnt ? (w.mlist[lane] & RXE_TGT_MASK) * 0x9E3779B1u + lane * 0x85EBCA6Bu : 0u;
    hsh ^= hsh >> 15;
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) hsh += (uint32_t)__shfl_xor((int)hsh, d);
    hsh = (hsh ^ (hsh >> 13)) * 0xC2B2AE35u + cnt;
    // 4. find or insert
    const uint32_t nch = cnt + DFA_HDR_WORDS <= 32u ? 1u : 2u;
    uint32_t mine = 0;  // chunk index allocated by this wave (0 = none yet)
    uint32_t found = 0;
    uint32_t slot = hsh & p.dfa_hash_mask;
    for (uint32_t probe = 0; probe <= p.dfa_hash_mask; probe++, slot = (slot + 1u) & p.dfa_hash_mask) {
      uint32_t cand = aload(p.dfa_hash + slot);
      if (cand == 0) {
        if (!mine) {  // allocate + write + release, then publish
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(p.dfa_hdr + 1, nch);
          base = bcast(base, 0);
          if (base + nch > p.dfa_pool_chunks) { found = 0; mine = 0; break; }  // pool exhausted -> EXIT
          uint32_t* nc = p.dfa_pool + (size_t)base * 32u;
          if (lane == 0) { nc[0] = cnt; nc[1] = sumdeg; nc[2] = hasacc; }
          if (lane < cnt) nc[DFA_HDR_WORDS + lane] = w.mlist[lane];
          __threadfence();
          mine = base;
          if (lane == 0) atomicAdd(p.dfa_hdr + 2, 1u);
        }
        uint32_t old = 0;
        if (lane == 0) old = atomicCAS(p.dfa_hash + slot, 0u, mine);
        old = bcast(old, 0);
        if (old == 0) { found = mine; break; }
        cand = old;  // somebody else took the slot meanwhile: is it the same set?
      }
      const uint32_t* cc2 = p.dfa_pool + (size_t)cand * 32u;
      bool diff = aload(cc2) != cnt;
      if (!diff && lane < cnt) diff = (aload(cc2 + DFA_HDR_WORDS + lane) ^ w.mlist[lane]) != 0;
      if (wballot(diff) == 0) { found = cand; break; }
    }
    nv = found ? (found | (hasacc ? DFA_ACC : 0u)) : DFA_EXIT;
  }
  // 5. publish the transition (monotonic 0 -> value)
  if (lane == 0) {
    __hip_atomic_store(p.dfa_trans + (size_t)id * ncls + cc, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicAdd(p.dfa_hdr + 3, 1u);
  }
  return nv;
}

template <bool STATS>
__global__ void __launch_bounds__(256) rx_dfa_kernel(const RxParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  uint32_t* cmapw = lds;
  const uint8_t* cmap = reinterpret_cast<const uint8_t*>(cmapw);
  DfaWave w;
  w.bits = lds + 64u + (size_t)wib * (p.nw32 + 64u);
  w.mlist = w.bits + p.nw32;
  for (uint32_t i = threadIdx.x; i < 64u; i += blockDim.x) cmapw[i] = p.byte_class[i];
  for (uint32_t i = lane; i < p.nw32; i += 64u) w.bits[i] = 0u;
  __syncthreads();
  const uint32_t ncls = p.n_classes;
  unsigned long long st_active = 0, st_edges = 0;

  zero_next_counters(p);
  const uint32_t stream = (blockIdx.x * wpb + wib) * 64u + lane;
  bool alive = stream < p.n_streams;
  {
    const uint32_t first = (blockIdx.x * wpb + wib) * 64u;
    if (first < p.n_streams) zero_final_rows(p, first, p.n_streams - first < 64u ? p.n_streams - first : 64u, lane);
  }
  const uint8_t* base = p.bytes + (size_t)(alive ? stream : 0) * p.stride;
  const bool aligned = (reinterpret_cast<uintptr_t>(base) & 3u) == 0;
  auto load16 = [&](uint32_t chunk, uint32_t (&o)[4]) {  // 16 bytes of this lane's own stream
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t off = chunk * 16u + 4u * q;
      uint32_t v = 0;
      if (alive) {
        if (aligned && off + 4u <= p.stream_len) v = *reinterpret_cast<const uint32_t*>(base + off);
        else
          for (uint32_t b = 0; b < 4; b++)
            if (off + b < p.stream_len) v |= (uint32_t)base[off + b] << (8u * b);
      }
      o[q] = v;
    }
  };
  uint32_t cur = 1u | ((p.state0_entry & RXE_ACCEPT) ? DFA_ACC : 0u);  // id 1 = {state 0} (FPGA.v:134-147)
  uint32_t win[4] = {0, 0, 0, 0}, nxt[4];
  load16(0, nxt);
  uint32_t am_word = 0;

  for (uint32_t k = 0; k < p.n_passes; k++) {
    const bool consume = k < p.n_consume;
    const uint32_t id = cur & ~DFA_ACC;
    // accept pulses of the current sets (rare): the wave reads one flagged stream's members at a time
    {
      uint64_t ma = wballot(alive && (cur & DFA_ACC));
      while (ma) {
        const uint32_t src = (uint32_t)__builtin_ctzll(ma);
        ma &= ma - 1;
        const uint32_t sid = bcast(id, src), sstream = bcast(stream, src);
        const uint32_t* chunk = p.dfa_pool + (size_t)sid * 32u;
        const uint32_t n = aload(chunk);
        for (uint32_t q0 = 0; q0 < n; q0 += 64u) {
          const uint32_t e = q0 + lane < n ? aload(chunk + DFA_HDR_WORDS + q0 + lane) : 0u;
          uint32_t dummy = 0;
          emit_events(p, (e & RXE_ACCEPT) != 0, e & RXE_TGT_MASK, sstream, k, lane, dummy);
        }
        if (lane == src) am_word |= 1u << (k & 31u);
      }
    }
    if (consume) {
      if ((k & 15u) == 0) {  // next 16 bytes of the own stream, as byte classes
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t v = nxt[q];
          win[q] = (uint32_t)cmap[v & 0xFFu] | ((uint32_t)cmap[(v >> 8) & 0xFFu] << 8) |
                   ((uint32_t)cmap[(v >> 16) & 0xFFu] << 16) | ((uint32_t)cmap[v >> 24] << 24);
        }
        load16((k >> 4) + 1u, nxt);
      }
      const uint32_t kq = (k >> 2) & 3u;
      const uint32_t wsel = kq == 0 ? win[0] : (kq == 1 ? win[1] : (kq == 2 ? win[2] : win[3]));
      const uint32_t c = (wsel >> ((k & 3u) * 8u)) & 0xFFu;
      if (STATS && alive) {
        const uint32_t* chunk = p.dfa_pool + (size_t)id * 32u;
        st_active += aload(chunk);
        st_edges += aload(chunk + 1);
      }
      uint32_t v = 1u;
      if (alive) {  // THE pass: one look-up per stream
        v = p.dfa_trans[(size_t)id * ncls + c];
      }
      uint64_t mm = wballot(alive && v == 0u);
      while (mm) {  // transitions not built yet: the whole wave builds one at a time
        const uint32_t src = (uint32_t)__builtin_ctzll(mm);
        const uint32_t bid = bcast(id, src), bc = bcast(c, src);
        const uint32_t nv = dfa_build(p, w, bid, bc, lane);
        const bool same = alive && v == 0u && id == bid && c == bc;
        if (same) v = nv;
        mm &= ~wballot(same);
      }
      // sets the table cannot hold: hand the stream (S_k as a bitmask row, k) to the wave kernel
      uint64_t mx = wballot(alive && v == DFA_EXIT);
      while (mx) {
        const uint32_t src = (uint32_t)__builtin_ctzll(mx);
        mx &= mx - 1;
        const uint32_t sid = bcast(id, src), sstream = bcast(stream, src), sam = bcast(am_word, src);
        uint32_t slot = 0;
        if (lane == 0) slot = (uint32_t)atomicAdd(p.spill_count, 1ull);
        slot = bcast(slot, 0);
        const uint32_t* chunk = p.dfa_pool + (size_t)sid * 32u;
        const uint32_t n = aload(chunk);
        if (lane < n) { const uint32_t s = aload(chunk + DFA_HDR_WORDS + lane) & RXE_TGT_MASK; atomicOr(&w.bits[s >> 5], 1u << (s & 31u)); }
        wave_sync();
        uint32_t* row = p.spill_rows + (size_t)slot * p.nw64x2;
        for (uint32_t i = lane; i < p.nw64x2; i += 64u) row[i] = i < p.nw32 ? w.bits[i] : 0u;
        wave_sync();
        for (uint32_t i = lane; i < p.nw32; i += 64u) w.bits[i] = 0u;
        wave_sync();
        if (lane == 0) {
          p.spill_streams[slot] = sstream;
          p.spill_k[slot] = k;
          if (p.anymatch) p.anymatch[(size_t)sstream * p.anymatch_stride + (k >> 5)] = sam;
        }
        if (lane == src) alive = false;
      }
      if (alive) cur = v;
    }
    if (p.anymatch && ((k & 31u) == 31u || k + 1 == p.n_passes)) {
      if (alive) p.anymatch[(size_t)stream * p.anymatch_stride + (k >> 5)] = am_word;
      am_word = 0;
    }
  }
  // final active sets: the rows were zeroed at the start of this kernel
  if (p.final_active) {
    uint64_t mf = wballot(alive);
    while (mf) {
      const uint32_t src = (uint32_t)__builtin_ctzll(mf);
      mf &= mf - 1;
      const uint32_t sid = bcast(cur & ~DFA_ACC, src), sstream = bcast(stream, src);
      const uint32_t* chunk = p.dfa_pool + (size_t)sid * 32u;
      const uint32_t n = aload(chunk);
      if (lane < n) {
        const uint32_t s = aload(chunk + DFA_HDR_WORDS + lane) & RXE_TGT_MASK;
        atomicOr(&p.final_active[(size_t)sstream * p.nw64x2 + (s >> 5)], 1u << (s & 31u));
      }
    }
  }
  if (STATS) {
    if (st_active) atomicAdd(&p.counters[1], st_active);
    if (st_edges) atomicAdd(&p.counters[2], st_edges);
  }
}

// =================================================================================================
// Kernel 6: register-resident active set, one wavefront per stream (few long streams)
// =================================================================================================
// The reference's own run is ONE lock-step pair of streams (testbench_BLK_Mem.sv:49-87): a single dependency chain
// per stream, so what counts is the latency of a pass — a lone wavefront issues in order at ~5 cycles per instruction,
// an L1 hit costs it ~200 cycles, a taken branch more than a straight line (DESIGN.md 3.3 has the measurements).
// So the active set never leaves the registers and the common pass is a dozen straight-line instructions:
//   * lane L holds at most one state id in a VGPR; a free lane holds the id `size`, whose row in the index is empty
//     (accept states have empty rows anyway), so every lane gathers unconditionally — no EXEC masking, no branch;
//   * the in-place update of a lane is PRECOMPUTED per (state, class) at load time (RxParams::regidx, rx_host.cpp):
//     the state itself if it loops on the byte, its one target if nothing else can reach that target, else "free" —
//     the pass is   gather 8 bytes  ->  v_and  ->  address  ->  next gather;
//   * the byte classes of four passes come out of a register window with one v_readlane per four passes; what the
//     folded `.*` state emits on each byte (already reduced to the targets that survive the NEXT byte) is looked up for
//     256 passes at a time, vectorised, when the window is refilled, and costs one v_readlane per pass;
//   * which lanes hold an accept state is a flag bit of the fast word (pulses are rare);
//   * no LDS list, no filter, no atomics.  What needs a lane of its own — a target next to a surviving state, a target
//     that a second active state could also reach (RXE_MAYDUP: compared against all lanes at once), the targets of
//     multi-target rows, the folded state's targets — is flagged in the fast word and placed by scalar code (one single
//     target: a straight-line path; else loops over v_readlane / first free lane).  Single targets get one byte of
//     look-ahead (RxParams::reg_tmask): one that has no edge on the next byte's class (mod 8) is not placed at all —
//     about every third pass of the busier shipped trace places something, every second would without it.
// More than 64 active states: the stream is handed to the wave kernel (resume mode) like in the group / pack kernels.
// SKIP: groups of four passes in which no lane holds a state and the folded state emits nothing are stepped over (the
// quieter shipped trace: 63 % of all passes; uniform bytes: nearly all).  A build of its own because the test, three
// instructions per group, costs the other build's code 4-14 % through register allocation (measured: hi trace 42.4 ->
// 44.7 ms, l7 small batches 0.23 -> 0.26 ms): AUTO picks per batch (rx_api.cpp).
template <bool FOLD, bool SKIP>
__global__ void __launch_bounds__(64) rx_sym_reg_kernel(const RxParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t* cmapw = lds;
  const uint8_t* cmap = reinterpret_cast<const uint8_t*>(cmapw);
  cmapw[lane] = p.byte_class[lane];
  wave_sync();
  zero_next_counters(p);
  const uint32_t stream = blockIdx.x;
  if (stream >= p.n_streams) return;
  zero_final_rows(p, stream, 1u, lane);
  const uint32_t ncls = p.n_classes, ncls8 = ncls * 8u;
  const uint32_t FREE = p.size;  // the empty row behind the index
  const char* __restrict__ regidx = reinterpret_cast<const char*>(p.regidx);
  const uint32_t* __restrict__ ovf = p.ovf;
  ByteFeed feed;
  feed.base = p.bytes + (size_t)stream * p.stride;
  feed.len = p.stream_len;
  feed.aligned = ((reinterpret_cast<uintptr_t>(feed.base)) & 3u) == 0;
  auto classes = [&](uint32_t v) {
    return (uint32_t)cmap[v & 0xFFu] | ((uint32_t)cmap[(v >> 8) & 0xFFu] << 8) | ((uint32_t)cmap[(v >> 16) & 0xFFu] << 16) |
           ((uint32_t)cmap[v >> 24] << 24);
  };
  const uint32_t n_consume = p.n_consume < p.n_passes ? p.n_consume : p.n_passes;
  // byte classes: lane j holds the classes of bytes 4j..4j+3 of the current 256-byte chunk (cw) and of the next (cwn);
  // the raw chunk after that is in flight.  va[q]: what the folded state emits in pass (chunk base + 4j + q).
  uint32_t cw = 0, cwn = classes(feed.load_chunk(0, lane)), raw = feed.load_chunk(1, lane);
  uint32_t va[4] = {0u, 0u, 0u, 0u};
  uint32_t e = lane == 0 ? 0u : FREE;             // FPGA.v:134-147: current = {state 0}
  uint64_t macc = (p.state0_entry & RXE_ACCEPT) ? 1ull : 0ull;  // lanes that hold an accept state
  uint32_t am_word = 0;
  bool handed_off = false;
  uint32_t n_skipped = 0;  // SKIP: groups stepped over in the second half of the stream (what AUTO's trial run reads)
  // shader clock the stream ran at (diagnostic; read by RX_OPT_VERBOSE): cycles and 100 MHz ticks of stream 0
  const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();

  auto pulses = [&](uint32_t k) {  // accept pulses of S_k (FPGA.v:210-226); rare
    const bool acc = (macc >> lane) & 1ull;
    emit_events(p, acc, e, stream, k, lane, am_word);
  };
  auto store_anymatch = [&](uint32_t k) {
    if (p.anymatch && lane == 0) p.anymatch[(size_t)stream * p.anymatch_stride + (k >> 5)] = am_word;
    am_word = 0;
  };
  // {fast word, slice word} of every lane's state for a byte of class c
  auto gather = [&](uint32_t c) {
    return *reinterpret_cast<const uint2*>(regidx + (__umul24(e, ncls8) + (c << 3)));
  };
  // the rare part of a pass: everything that needs a lane of its own.  xs = slice words, e_in = S_k (for the hand-off).
  auto slow = [&](uint32_t k, uint32_t xs, bool kept, uint32_t e_in, uint32_t vA) {
    const bool surv = (xs & RXE_SELF) != 0u;
    const bool inl = FOLD ? (xs & (RXE_INLINE | RXE_PIN)) == RXE_INLINE : (xs & RXE_INLINE) != 0u;
    uint64_t mx = wballot(inl && kept && (surv || (xs & RXE_MAYDUP)));  // (an in-place target is neither)
    uint64_t mo = wballot(xs & RXE_OVF);
    uint64_t mfree = wballot(e == FREE);
    bool full = false;
    auto place = [&](uint32_t tw) {  // tw wave-uniform: target | RXE_ACCEPT | RXE_MAYDUP | RXE_PIN
      if (FOLD && (tw & RXE_PIN)) return;
      const uint32_t t = tw & RXE_TGT_MASK;
      if ((tw & RXE_MAYDUP) && wballot(e == t)) return;  // already in the next set
      if (mfree == 0ull) { full = true; return; }
      const uint32_t dst = (uint32_t)__builtin_ctzll(mfree);
      mfree &= mfree - 1ull;
      if (lane == dst) e = t;
      if (tw & RXE_ACCEPT) macc |= 1ull << dst;
    };
    auto place_list = [&](uint32_t off) {
      const uint32_t cnt = ovf[off];
      for (uint32_t j0 = 0; j0 < cnt; j0 += 64u) {
        const uint32_t mine = j0 + lane < cnt ? ovf[off + 1u + j0 + lane] : 0u;  // one coalesced load per 64 targets
        const uint32_t lim = cnt - j0 < 64u ? cnt - j0 : 64u;
        for (uint32_t j = 0; j < lim; j++) place(bcast(mine, j));
      }
    };
    while (mx) {
      const uint32_t src = (uint32_t)__builtin_ctzll(mx);
      mx &= mx - 1ull;
      place(bcast(xs, src));
    }
    while (mo) {
      const uint32_t src = (uint32_t)__builtin_ctzll(mo);
      mo &= mo - 1ull;
      place_list(bcast(xs, src) & RXE_TGT_MASK);
    }
    if (FOLD) {
      if (vA & RXE_INLINE) place(vA);
      else if (vA & RXE_OVF) place_list(vA & RXE_TGT_MASK);
    }
    if (__builtin_expect(full, 0)) {
      // more than 64 active states: hand the stream (S_k, k) to the wave kernel
      unsigned long long b = 0;
      if (lane == 0) b = atomicAdd(p.spill_count, 1ull);
      const uint32_t slot = bcast((uint32_t)b, 0);
      uint32_t* row = p.spill_rows + (size_t)slot * p.nw64x2;
      for (uint32_t w = lane; w < p.nw64x2; w += 64u) row[w] = 0u;
      if (lane == 0) {
        p.spill_streams[slot] = stream;
        p.spill_k[slot] = k;
        if (p.anymatch) p.anymatch[(size_t)stream * p.anymatch_stride + (k >> 5)] = am_word;
      }
      __threadfence();
      wave_sync();
      if (e_in != FREE) atomicOr(&row[e_in >> 5], 1u << (e_in & 31u));
      if (FOLD && k >= 1u && lane == 0) atomicOr(&row[p.pin_state >> 5], 1u << (p.pin_state & 31u));
      handed_off = true;
    }
  };

  uint2 x = n_consume ? gather(bcast(cwn, 0) & 0xFFu) : make_uint2(0u, 0u);  // the words for pass 0 are in flight
  // One byte-consuming pass.  cn = class of the NEXT byte (scalar), vA = the folded state's emission in this pass,
  // look = the next byte will be consumed too (a single target with no edge on it need not be placed; the set after
  // the stream's last byte is reported and gets everything).
  const uint32_t tmask = p.reg_tmask;
  const uint32_t nm_base = tmask == 0xFFFFu ? 0x10000u : RXR_NEED, nm_and = tmask == 0xFFFFu ? 7u : 0u;
  auto pass = [&](uint32_t k, uint32_t cn, uint32_t vA, bool look) {
    if (__builtin_expect(macc != 0ull, 0)) pulses(k);
    const uint32_t xf = x.x, xs = x.y, e_in = e;
    e = xf & tmask;                                 // every lane's in-place update, precomputed
    const uint32_t nm = look ? nm_base << (cn & nm_and) : RXR_NEED;
    const uint64_t need = wballot(xf & nm);         // RXR_NEED, or its look-ahead bit for the next byte's class
    if (__builtin_expect(need == 0ull && vA == 0u, 1)) {
      x = gather(cn);                                // the dependency chain of the stream ends here: gather k+1 is out
      __builtin_amdgcn_sched_barrier(0);
      macc = wballot(xf & RXR_ACC);                  // built while the gather is in flight
    } else {
      macc = wballot(xf & RXR_ACC);
      // Most such passes have exactly ONE single target to place — the folded state's emission, or the target of one
      // lane: straight-line code for that (the general path below is three loops and ~100 instructions).
      uint32_t tw = 0u;
      if (need == 0ull) tw = vA;
      else if (vA == 0u && (need & (need - 1ull)) == 0ull) tw = bcast(xs, (uint32_t)__builtin_ctzll(need));
      bool done = false;
      if ((tw & RXE_INLINE) && !(FOLD && (tw & RXE_PIN))) {
        const uint32_t t = tw & RXE_TGT_MASK;
        const uint64_t mfree = wballot(e == FREE);
        if ((tw & RXE_MAYDUP) && wballot(e == t) != 0ull) {
          done = true;                                // already in the next set
        } else if (mfree != 0ull) {
          const uint32_t dst = (uint32_t)__builtin_ctzll(mfree);
          if (lane == dst) e = t;
          if (tw & RXE_ACCEPT) macc |= 1ull << dst;
          done = true;
        }
      }
      if (!done) slow(k, xs, (xf & nm) != 0u, e_in, vA);
      x = gather(cn);
    }
  };

  uint32_t k = 0;
  while (k < n_consume && !handed_off) {  // k is a multiple of 256 here: next chunk of the stream
    cw = cwn;
    cwn = classes(raw);
    raw = feed.load_chunk((k >> 8) + 2u, lane);  // two chunks ahead as bytes, one ahead as classes
    if (FOLD) {
      // the folded `.*` state's emissions for the 256 passes of this chunk, four per lane: table column = class of the
      // NEXT byte (look-ahead pruning), except at the stream's last byte, whose sets are reported; none in pass 0
      const uint32_t nxt_first = (uint32_t)__shfl_down((int)cw, 1);
      const uint32_t next_chunk_first = bcast(cwn, 0);
      const unsigned long long c5 = ((unsigned long long)(lane == 63u ? next_chunk_first : nxt_first) << 32) | cw;
#pragma unroll
      for (uint32_t q = 0; q < 4u; q++) {
        const uint32_t kk = k + 4u * lane + q;
        const uint32_t c0 = (uint32_t)(c5 >> (8u * q)) & 0xFFu, c1 = (uint32_t)(c5 >> (8u * q + 8u)) & 0xFFu;
        va[q] = (kk >= 1u && kk < n_consume) ? p.pin_tab[c0 * p.pin_cols + (kk + 1u < n_consume ? c1 : ncls)] : 0u;
      }
    }
    const uint32_t kchunk = n_consume - k < 256u ? n_consume : k + 256u;
    // SKIP: groups of four passes of this chunk in which the folded state emits something (lane = group index; kept per
    // lane and balloted only when needed: the loop below has no scalar register to spare)
    const uint32_t gbusy_v = (SKIP && FOLD) ? (va[0] | va[1] | va[2] | va[3]) : 0u;
    // four passes on the classes of one SGPR pair, no per-pass loop tests; never the stream's last pass
    while (k + 4u <= kchunk && k + 4u < n_consume && !handed_off) {
      const uint32_t gi = (k >> 2) & 63u;
      if (SKIP && __builtin_expect(macc == 0ull && wballot(e != FREE) == 0ull, 0)) {
        // No active state but the folded one: until that state emits something again nothing happens at all — skip those
        // groups of passes in one step (x holds the empty row's words, which are the same for every byte).
        const unsigned long long rest = wballot(gbusy_v != 0u && lane >= gi) >> gi;  // bit 0 = this group
        if ((rest & 1ull) == 0ull) {
          uint32_t skip = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;   // groups
          const uint32_t to_word = 8u - ((k >> 2) & 7u), to_chunk = (kchunk - k) >> 2, to_end = (n_consume - k - 1u) >> 2;
          skip = skip < to_word ? skip : to_word;
          skip = skip < to_chunk ? skip : to_chunk;
          skip = skip < to_end ? skip : to_end;
          if (k >= (n_consume >> 1)) n_skipped += skip;
          k += 4u * skip;
          if (p.anymatch && (k & 31u) == 0u) store_anymatch(k - 1u);
          continue;
        }
      }
      const uint32_t c4 = bcast(cw, gi);
      const uint32_t c4n = gi == 63u ? bcast(cwn, 0) : bcast(cw, gi + 1u);
      pass(k, (c4 >> 8) & 0xFFu, FOLD ? bcast(va[0], gi) : 0u, true);
      if (__builtin_expect(handed_off, 0)) break;
      pass(k + 1u, (c4 >> 16) & 0xFFu, FOLD ? bcast(va[1], gi) : 0u, true);
      if (__builtin_expect(handed_off, 0)) break;
      pass(k + 2u, c4 >> 24, FOLD ? bcast(va[2], gi) : 0u, true);
      if (__builtin_expect(handed_off, 0)) break;
      pass(k + 3u, c4n & 0xFFu, FOLD ? bcast(va[3], gi) : 0u, true);
      if (__builtin_expect(handed_off, 0)) break;
      k += 4u;
      if (p.anymatch && (k & 31u) == 0u) store_anymatch(k - 1u);
    }
    while (k < kchunk && !handed_off) {  // the last passes of the stream (or of a chunk that ends with it)
      const uint32_t gi = (k >> 2) & 63u, q = k & 3u;
      const uint32_t c4 = bcast(cw, gi);
      const uint32_t c4n = gi == 63u ? bcast(cwn, 0) : bcast(cw, gi + 1u);
      const uint32_t vq = q == 0u ? va[0] : (q == 1u ? va[1] : (q == 2u ? va[2] : va[3]));
      const uint32_t cn = q == 3u ? c4n & 0xFFu : (c4 >> (8u * q + 8u)) & 0xFFu;
      pass(k, cn, FOLD ? bcast(vq, gi) : 0u, k + 1u < n_consume);
      if (handed_off) break;
      k++;
      if (p.anymatch && (k & 31u) == 0u) store_anymatch(k - 1u);
    }
  }
  if (!handed_off) {
    bool unsaved = (k & 31u) != 0u;  // passes whose any-match bits are still in am_word
    if (k < p.n_passes) {  // RX_MODE_FULL: pass N only looks for accept states
      if (macc != 0ull) pulses(k);
      k++;
      unsaved = true;
    }
    if (p.anymatch && unsaved) store_anymatch(k - 1u);
  }
  if (stream == 0 && lane == 0) {
    p.counters[8] = __builtin_amdgcn_s_memtime() - t0c;
    p.counters[9] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
  if (SKIP && lane == 0 && n_skipped) atomicAdd(&p.counters[10], (unsigned long long)n_skipped);
  // final active set: the row was zeroed at the start of this kernel
  if (p.final_active && !handed_off) {
    uint32_t* row = p.final_active + (size_t)stream * p.nw64x2;
    if (e != FREE) atomicOr(&row[e >> 5], 1u << (e & 31u));
    if (FOLD && n_consume >= 1u && lane == 0) atomicOr(&row[p.pin_state >> 5], 1u << (p.pin_state & 31u));
  }
}

// =================================================================================================
// Final sets as compact lists (rx_plan_run on request): one wavefront per stream over its bitmask row
// =================================================================================================
// A row is 2*ceil(size/64) words whatever it holds (1.2 KB for snort_16, ~3 bits set): the lists are what goes over PCIe.
__global__ void __launch_bounds__(1024) rx_final_compact_kernel(const uint32_t* __restrict__ rows, uint32_t n_streams, uint32_t row_words,
                                                                 uint32_t* __restrict__ states, uint32_t cap, uint32_t* __restrict__ off,
                                                                 uint32_t* __restrict__ cnt, unsigned long long* counter) {
  __shared__ uint32_t wtot[16];
  __shared__ unsigned long long wbase[16];
  const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
  const uint32_t stream = blockIdx.x * 16u + wib;
  const bool have = stream < n_streams;
  const uint32_t* row = rows + (size_t)(have ? stream : 0u) * row_words;
  // pass 1: how many states
  uint32_t mine = 0;
  if (have)
    for (uint32_t w = lane; w < row_words; w += 64u) mine += (uint32_t)__popc(row[w]);
  uint32_t total = mine;
  for (int d = 32; d >= 1; d >>= 1) total += (uint32_t)__shfl_xor((int)total, d);
  // one atomic per block of 16 streams (one per stream serialises on the counter: 0.39 ms for 32 768 streams)
  if (lane == 0) wtot[wib] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sum = 0;
    for (uint32_t i = 0; i < 16u; i++) sum += wtot[i];
    unsigned long long b = sum ? atomicAdd(counter, (unsigned long long)sum) : 0ull;
    for (uint32_t i = 0; i < 16u; i++) { wbase[i] = b; b += wtot[i]; }
  }
  __syncthreads();
  if (!have) return;
  const unsigned long long base = wbase[wib];
  if (lane == 0) {
    off[stream] = (uint32_t)(base < cap ? base : cap);
    cnt[stream] = total;
  }
  if (total == 0) return;
  // pass 2: ascending order = word order, and within a sweep of 64 words lane order
  unsigned long long at = base;
  for (uint32_t w0 = 0; w0 < row_words; w0 += 64u) {
    const uint32_t w = w0 + lane;
    uint32_t bits = w < row_words ? row[w] : 0u;
    const uint32_t n = (uint32_t)__popc(bits);
    uint32_t incl = n;  // inclusive prefix sum over the lanes
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= (uint32_t)d) incl += v;
    }
    unsigned long long o = at + (incl - n);
    while (bits) {
      const uint32_t bpos = (uint32_t)__builtin_ctz(bits);
      bits &= bits - 1u;
      if (o < cap) states[o] = w * 32u + bpos;
      o++;
    }
    at += bcast(incl, 63);
  }
}

}  // namespace

int rx_launch_final_compact(const uint32_t* rows, uint32_t n_streams, uint32_t row_words, uint32_t* states, uint32_t cap,
                            uint32_t* off, uint32_t* cnt, unsigned long long* counter, void* hip_stream) {
  if (n_streams == 0) return 0;
  hipLaunchKernelGGL(rx_final_compact_kernel, dim3((n_streams + 15u) / 16u), dim3(1024), 0, reinterpret_cast<hipStream_t>(hip_stream), rows,
                     n_streams, row_words, states, cap, off, cnt, counter);
  return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// launch configuration
// -------------------------------------------------------------------------------------------------
int rx_pick_launch(uint32_t kernel, uint32_t size, uint32_t n_streams, int cu_count, size_t lds_per_cu,
                   RxParams* p, RxLaunchCfg* cfg) {
  cfg->cu_count = cu_count;
  cfg->lds_per_cu = lds_per_cu;
  if (kernel == RX_KERNEL_AUTO) kernel = RX_KERNEL_SYM_PACK;  // fastest parity-checked kernel
  if (kernel != RX_KERNEL_CSR_WAVE && kernel != RX_KERNEL_SYM_WAVE && kernel != RX_KERNEL_SYM_GROUP &&
      kernel != RX_KERNEL_SYM_PACK && kernel != RX_KERNEL_DFA && kernel != RX_KERNEL_SYM_REG)
    return RX_EINVAL;
  const uint32_t nw32 = (size + 31u) / 32u;
  p->nw32 = nw32;
  // wave-per-stream carve (also used by the resume launch that follows a group launch)
  // Lists as long as the LDS allows, up to RX_LIST_CAP_MAX entries: a set that outgrows its list is walked as a bitmask, one
  // 32-state word at a time — 222 active states of snort_16 are ~150 nearly empty words per pass against four full sweeps of
  // a list (the streams the pack kernel hands off are exactly such streams: bench.py handoff_mix_T)
  // (a batch that runs on the wave kernels as a whole keeps eight wavefronts per SIMD resident: 256 entries; the launch that
  // only finishes hand-offs has few streams and takes the long lists)
  uint32_t cap = (kernel == RX_KERNEL_CSR_WAVE || kernel == RX_KERNEL_SYM_WAVE) ? 256u : RX_LIST_CAP_MAX;
  while (cap > RX_LIST_CAP && (size_t)(2u * nw32 + 2u * cap) * 4u * 4u > lds_per_cu / 2) cap >>= 1;
  p->lds_words_per_stream = 2u * nw32 + 2u * cap;
  const size_t per_wave = (size_t)p->lds_words_per_stream * 4u;
  if (per_wave > lds_per_cu) return RX_ECAPACITY;  // automaton too large for an LDS-resident bitmask
  uint32_t wpb = 4;
  while (wpb > 1 && per_wave * wpb > lds_per_cu / 2) wpb >>= 1;
  cfg->kernel = kernel;
  cfg->block_threads = wpb * 64u;
  cfg->lds_bytes = (uint32_t)(per_wave * wpb);
  uint32_t blocks = (n_streams + wpb - 1) / wpb;
  cfg->grid_blocks = blocks ? blocks : 1;
  if (kernel == RX_KERNEL_SYM_PACK) {
    const uint32_t gl = cfg->group_lanes;  // here: streams per wavefront
    if (gl != 2 && gl != 4 && gl != 8 && gl != 11 && gl != 12 && gl != 13 && gl != 16 && gl != 20 && gl != 22 && gl != 24 &&
        gl != 32 && gl != 48 && gl != 64)  // 48 and 64 exist as FOLD builds only (rx_launch maps them down otherwise)
      cfg->group_lanes = 16;
  }
  if (kernel == RX_KERNEL_SYM_GROUP) {
    const uint32_t gl = cfg->group_lanes;
    if (gl != 1 && gl != 2 && gl != 4 && gl != 8 && gl != 16) cfg->group_lanes = 4;
  }
  return RX_OK;
}

static thread_local bool g_verbose = false;  // set per rx_launch call from RxLaunchCfg::verbose (rx_opts.flags)

template <typename K>
static int launch_one(K kern, const RxParams& p, uint32_t grid, uint32_t block, uint32_t lds, hipStream_t s) {
  if (lds > 64u * 1024u) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (g_verbose) {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, (int)block, lds);
    fprintf(stderr, "[rxmatch] grid %u x %u threads, %u B LDS/block -> %d blocks/CU resident\n", grid, block, lds, nb);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, s, p);
  return (int)hipGetLastError();
}

template <int G>
static int launch_group(const RxParams& p, const RxLaunchCfg& cfg, hipStream_t s) {
  using L = GroupLayout<G>;
  const uint32_t wpb = (G == 1) ? 2 : 4;
  const uint32_t waves = (p.n_streams + L::SPW - 1) / L::SPW;
  const uint32_t grid = (waves + wpb - 1) / wpb;
  const uint32_t lds = (L::PINW + wpb * L::SPW * L::REGION) * 4u;
  return cfg.stats ? launch_one(rx_sym_group_kernel<G, true>, p, grid ? grid : 1, wpb * 64u, lds, s)
                   : launch_one(rx_sym_group_kernel<G, false>, p, grid ? grid : 1, wpb * 64u, lds, s);
}

template <int S, bool PRUNE>
static int launch_pack_as(const RxParams& p, const RxLaunchCfg& cfg, hipStream_t s) {
  using L = PackLayout<S, PRUNE, false>;
  const uint32_t wpb = 4;
  const uint32_t waves = (p.n_streams + S - 1) / S;
  const uint32_t grid = (waves + wpb - 1) / wpb;
  const uint32_t g = grid ? grid : 1;
  const uint32_t lds = (L::CMAPW + wpb * L::WAVE_WORDS) * 4u;
  if (PRUNE) return launch_one(rx_sym_pack_kernel<S, false, false, true, false>, p, g, wpb * 64u, lds, s);
  if (cfg.stats) return launch_one(rx_sym_pack_kernel<S, true, false, false, false>, p, g, wpb * 64u, lds, s);
  if (S == 16 && cfg.profile_pack)  // stamped diagnostic build, see the kernel's PROF note
    return launch_one(rx_sym_pack_kernel<16, false, true, false, false>, p, g, wpb * 64u, lds, s);
  return launch_one(rx_sym_pack_kernel<S, false, false, false, false>, p, g, wpb * 64u, lds, s);
}

template <int S>
static int launch_pack(const RxParams& p, const RxLaunchCfg& cfg, hipStream_t s) {
  if (cfg.prune && !cfg.stats && p.symidx_p) return launch_pack_as<S, true>(p, cfg, s);
  return launch_pack_as<S, false>(p, cfg, s);
}

// FOLD builds: the block shares one copy of the folding table, so blocks are as large as the LDS allows (up to 8
// wavefronts); never with statistics.
template <int S>
static int launch_fold(const RxParams& p, const RxLaunchCfg& cfg, hipStream_t s, size_t lds_per_cu) {
  const bool prune = cfg.prune && p.symidx_p;
  const uint32_t ww = prune ? PackLayout<S, true, true>::WAVE_WORDS : PackLayout<S, false, true>::WAVE_WORDS;
  const uint32_t fixed = PackLayout<S, false, true>::CMAPW + p.n_classes * p.pin_cols;
  const uint32_t waves = (p.n_streams + S - 1) / S;
  // wavefronts per block (every block carries its own copy of the folding table): fewest rounds of resident blocks
  // per CU first, then the fewest wavefronts on the busiest CU, then the larger block
  uint32_t wpb = 0, best_rounds = ~0u, best_load = ~0u;
  const uint32_t cus = cfg.cu_count > 0 ? (uint32_t)cfg.cu_count : 256u;
  for (uint32_t w = 1; w <= 8; w++) {
    const size_t bytes = (size_t)(fixed + w * ww) * 4u;
    if (bytes > lds_per_cu) break;
    const uint32_t resident = (uint32_t)std::min<size_t>(lds_per_cu / bytes, 32u / w);  // blocks per CU at a time
    const uint32_t blocks = (waves + w - 1) / w;
    const uint32_t per_cu = (blocks + cus - 1) / cus;  // blocks the busiest CU gets
    const uint32_t rounds = (per_cu + resident - 1) / resident, load = per_cu * w;
    if (rounds < best_rounds || (rounds == best_rounds && load <= best_load)) { best_rounds = rounds; best_load = load; wpb = w; }
  }
  if (wpb == 0) return (int)hipErrorInvalidValue;
  const uint32_t grid = (waves + wpb - 1) / wpb;
  const uint32_t g = grid ? grid : 1;
  const uint32_t lds = (fixed + wpb * ww) * 4u;
  if (prune) return launch_one(rx_sym_pack_kernel<S, false, false, true, true>, p, g, wpb * 64u, lds, s);
  return launch_one(rx_sym_pack_kernel<S, false, false, false, true>, p, g, wpb * 64u, lds, s);
}

// returns a hipError_t value (0 = hipSuccess)
int rx_launch(const RxParams& p, const RxLaunchCfg& cfg, void* hip_stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  g_verbose = cfg.verbose;
  switch (cfg.kernel) {
    case RX_KERNEL_CSR_WAVE:
      return cfg.stats ? launch_one(rx_csr_wave_kernel<true>, p, cfg.grid_blocks, cfg.block_threads, cfg.lds_bytes, s)
                       : launch_one(rx_csr_wave_kernel<false>, p, cfg.grid_blocks, cfg.block_threads, cfg.lds_bytes, s);
    case RX_KERNEL_SYM_WAVE:
      return cfg.stats ? launch_one(rx_sym_wave_kernel<true>, p, cfg.grid_blocks, cfg.block_threads, cfg.lds_bytes, s)
                       : launch_one(rx_sym_wave_kernel<false>, p, cfg.grid_blocks, cfg.block_threads, cfg.lds_bytes, s);
    case RX_KERNEL_DFA:
    case RX_KERNEL_SYM_REG:
    case RX_KERNEL_SYM_PACK:
    case RX_KERNEL_SYM_GROUP: {
      int e;
      if (cfg.kernel == RX_KERNEL_SYM_REG) {  // one wavefront (= one block) per stream
        const bool fold = cfg.fold && p.pin_tab;
        const uint32_t lds = 64u * 4u;  // the byte -> class map; the folding table is read with scalar loads
        if (cfg.reg_skip)
          e = fold ? launch_one(rx_sym_reg_kernel<true, true>, p, p.n_streams, 64u, lds, s)
                   : launch_one(rx_sym_reg_kernel<false, true>, p, p.n_streams, 64u, lds, s);
        else
          e = fold ? launch_one(rx_sym_reg_kernel<true, false>, p, p.n_streams, 64u, lds, s)
                   : launch_one(rx_sym_reg_kernel<false, false>, p, p.n_streams, 64u, lds, s);
      } else if (cfg.kernel == RX_KERNEL_DFA) {
        const uint32_t wpb = 4;
        const uint32_t grid = (p.n_streams + wpb * 64u - 1) / (wpb * 64u);
        const uint32_t lds = (64u + wpb * (p.nw32 + 64u)) * 4u;
        e = cfg.stats ? launch_one(rx_dfa_kernel<true>, p, grid ? grid : 1, wpb * 64u, lds, s)
                      : launch_one(rx_dfa_kernel<false>, p, grid ? grid : 1, wpb * 64u, lds, s);
      } else if (cfg.kernel == RX_KERNEL_SYM_PACK && cfg.fold && !cfg.stats && p.pin_tab) {
        const size_t lds_cu = cfg.lds_per_cu ? cfg.lds_per_cu : 160u * 1024u;
        const uint32_t gl = cfg.group_lanes;  // nearest instantiated number of streams per wavefront
        if (gl <= 8) e = launch_fold<8>(p, cfg, s, lds_cu);
        else if (gl <= 13) e = launch_fold<13>(p, cfg, s, lds_cu);
        else if (gl <= 16) e = launch_fold<16>(p, cfg, s, lds_cu);
        else if (gl <= 24) e = launch_fold<24>(p, cfg, s, lds_cu);
        else if (gl <= 32) e = launch_fold<32>(p, cfg, s, lds_cu);
        else if (gl <= 48) e = launch_fold<48>(p, cfg, s, lds_cu);
        else e = launch_fold<64>(p, cfg, s, lds_cu);
      } else if (cfg.kernel == RX_KERNEL_SYM_PACK) {
        if (cfg.group_lanes == 2) e = launch_pack<2>(p, cfg, s);
        else if (cfg.group_lanes == 4) e = launch_pack<4>(p, cfg, s);
        else if (cfg.group_lanes == 8) e = launch_pack<8>(p, cfg, s);
        else if (cfg.group_lanes == 11) e = launch_pack<11>(p, cfg, s);
        else if (cfg.group_lanes == 12) e = launch_pack<12>(p, cfg, s);
        else if (cfg.group_lanes == 13) e = launch_pack<13>(p, cfg, s);
        else if (cfg.group_lanes == 22) e = launch_pack<22>(p, cfg, s);
        else if (cfg.group_lanes == 20) e = launch_pack<20>(p, cfg, s);
        else if (cfg.group_lanes == 24) e = launch_pack<24>(p, cfg, s);
        else if (cfg.group_lanes >= 32) e = launch_pack<32>(p, cfg, s);
        else e = launch_pack<16>(p, cfg, s);
      } else if (cfg.group_lanes == 1) e = launch_group<1>(p, cfg, s);
      else if (cfg.group_lanes == 2) e = launch_group<2>(p, cfg, s);
      else if (cfg.group_lanes == 8) e = launch_group<8>(p, cfg, s);
      else if (cfg.group_lanes == 16) e = launch_group<16>(p, cfg, s);
      else e = launch_group<4>(p, cfg, s);
      if (e) return e;
      // second launch: the wave kernel finishes whatever the group kernel handed off (usually nothing;
      // the count is read on the device, so no host round-trip)
      RxParams r = p;
      r.resume = 1;
      r.zero_next = nullptr;  // the first launch of the pair has done it
      r.zero_words = 0;
      const uint32_t wpb = cfg.block_threads / 64u;
      uint32_t grid = (p.n_streams + wpb - 1) / wpb;
      const uint32_t fill = (cfg.cu_count > 0 ? (uint32_t)cfg.cu_count : 256u) * 8u;  // enough blocks to fill the chip
      if (grid > fill) grid = fill;
      return cfg.stats ? launch_one(rx_sym_wave_kernel<true>, r, grid ? grid : 1, cfg.block_threads, cfg.lds_bytes, s)
                       : launch_one(rx_sym_wave_kernel<false>, r, grid ? grid : 1, cfg.block_threads, cfg.lds_bytes, s);
    }
    default:
      return (int)hipErrorInvalidValue;
  }
}
