// rx_kernels.hip — gfx950 (CDNA4) kernels for the per-byte active-state update of a CSR NFA.
//
// What they replace: the per-clock FSM of module CSR_traversal (Design/FPGA.v:115-768).  The
// FPGA walks i = 0..size-1 every byte and spends >= size clocks per byte on inactive states
// (FPGA.v:744-765); here the active set is a compacted list per stream, so work is proportional
// to |S_k|, and thousands of independent streams are resident at once.
//
// Execution model (both kernels): ONE WAVEFRONT (64 lanes) OWNS ONE INPUT STREAM.
//   * per-stream state lives in that wave's private LDS slice: two size-bit bitmasks (dedup
//     filter for `next`, and the dense spill form of `current`) and two active-state lists;
//   * the stream's bytes are fetched 256 B per wave-load (one dword per lane, coalesced) one
//     chunk ahead, and the current byte is broadcast with v_readlane (wave-uniform, so the
//     symbol ends up in an SGPR);
//   * next-state insertion = ds_or_rtn_b32 on the bitmask (dedup) + __ballot/mbcnt/__popcll to
//     allocate list slots and to flag accept states — no workgroup barrier anywhere: the four
//     waves of a block never communicate;
//   * no MFMA: this is integer gather / bit-scatter.
//
// rx_csr_wave_kernel  reads the state-major CSR exactly as the .coe holds it (row_ptr pair, then
//                     the whole row, as FPGA.v:166-207 / :227-714 do): long rows are swept by all
//                     64 lanes (256 B coalesced per load), short rows one lane per row.
// rx_sym_wave_kernel  reads the load-time slice index instead: one u32 per (state, byte) that
//                     holds "the current byte's slice" of that row (rx_internal.hpp).
//
// Active set larger than RX_LIST_CAP: the list stops growing but the bitmask keeps every bit, and
// the next pass walks the bitmask instead ("dense" form).  Results are identical either way.
#include <hip/hip_runtime.h>

#include "rx_internal.hpp"

namespace {

__device__ __forceinline__ uint32_t rank_below(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
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
  const uint64_t ma = __ballot(acc);
  if (ma == 0) return;
  const uint32_t cnt = (uint32_t)__popcll(ma);
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(&p.counters[0], (unsigned long long)cnt);
  const uint32_t blo = bcast((uint32_t)base, 0), bhi = bcast((uint32_t)(base >> 32), 0);
  base = ((unsigned long long)bhi << 32) | blo;
  if (acc) {
    const unsigned long long idx = base + rank_below(ma);
    if (p.events && idx < p.events_cap) {
      rx_event e;
      e.stream = stream;
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
  const uint64_t m = __ballot(fresh);
  if (m) {
    const uint32_t slot = st.n_next + rank_below(m);
    if (fresh && slot < RX_LIST_CAP) st.nlist[slot] = t;
    st.n_next += (uint32_t)__popcll(m);
  }
}

__device__ __forceinline__ void stream_reset(const RxParams& p, StreamState& st, uint32_t* my, uint32_t stream,
                                             uint32_t lane) {
  st.cb = my;
  st.nb = my + p.nw32;
  st.clist = st.nb + p.nw32;
  st.nlist = st.clist + RX_LIST_CAP;
  for (uint32_t w = lane; w < 2u * p.nw32; w += 64u) my[w] = 0u;
  st.n_next = 0;
  if (p.init_active) {  // chunked streaming: resume from a caller-supplied active set
    const uint32_t* row = p.init_active + (size_t)stream * p.nw64x2;
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
  if (st.n_next > RX_LIST_CAP) {  // list overflowed: the bitmask is the set
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

__device__ __forceinline__ void stream_store_final(const RxParams& p, StreamState& st, uint32_t stream,
                                                   uint32_t lane) {
  if (!p.final_active) return;
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
      uint64_t m = __ballot(word != 0u);
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

  for (uint32_t stream = blockIdx.x * wpb + wib; stream < p.n_streams; stream += gridDim.x * wpb) {
    StreamState st;
    stream_reset(p, st, my, stream, lane);
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
        // short rows: one lane per row, edges in sequence
        const bool small = valid && deg > 0 && deg <= RX_SMALL_DEG;
        for (uint32_t j = 0; __ballot(small && j < deg) != 0; j++) {
          const bool act = small && j < deg;
          const uint32_t w = act ? col[base + j] : 0u;
          const bool hit = act && (w >> 24) == c;  // FPGA.v:264: transition == input_char
          if (__ballot(hit)) emit_target(st, hit, w & RXE_TGT_MASK, lane);
        }
        // long rows: all 64 lanes sweep one row, 256 B per load
        uint64_t mb = __ballot(valid && deg > RX_SMALL_DEG);
        while (mb) {
          const uint32_t src = (uint32_t)__builtin_ctzll(mb);
          mb &= mb - 1;
          const uint32_t b = bcast(base, src), d = bcast(deg, src);
          for (uint32_t j0 = 0; j0 < d; j0 += 64u) {
            const uint32_t j = j0 + lane;
            const bool act = j < d;
            const uint32_t w = act ? col[b + j] : 0u;
            const bool hit = act && (w >> 24) == c;
            if (__ballot(hit)) emit_target(st, hit, w & RXE_TGT_MASK, lane);
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

  for (uint32_t stream = blockIdx.x * wpb + wib; stream < p.n_streams; stream += gridDim.x * wpb) {
    StreamState st;
    stream_reset(p, st, my, stream, lane);
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
          nxt_word = feed.load_chunk((k >> 8) + 1u, lane);
        }
        c = (bcast(cur_word, (k >> 2) & 63u) >> ((k & 3u) * 8u)) & 0xFFu;
      }
      for_each_active<true>(p, st, lane, [&](bool valid, uint32_t e) {
        const uint32_t s = e & RXE_TGT_MASK;
        const bool acc = valid && (e & RXE_ACCEPT);
        emit_events(p, acc, s, stream, k, lane, am_word);
        if (!consume) return;
        if (STATS && valid) { st_active += 1; st_edges += rp[s + 1] - rp[s]; }
        // the current byte's slice of row s: one dword
        const uint32_t ent = (valid && !acc) ? symidx[(size_t)s * 256u + c] : 0u;
        if (__ballot(ent & RXE_SELF)) emit_target(st, (ent & RXE_SELF) != 0, s, lane);
        if (__ballot(ent & RXE_INLINE))
          emit_target(st, (ent & RXE_INLINE) != 0, ent & (RXE_TGT_MASK | RXE_ACCEPT), lane);
        if (__ballot(ent & RXE_OVF)) {
          const bool has = (ent & RXE_OVF) != 0;
          const uint32_t off = ent & RXE_TGT_MASK;
          const uint32_t cnt = has ? ovf[off] : 0u;
          for (uint32_t j = 0; __ballot(j < cnt) != 0; j++) {
            const bool act = j < cnt;
            const uint32_t t = act ? ovf[off + 1u + j] : 0u;
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

}  // namespace

// -------------------------------------------------------------------------------------------------
// launch configuration
// -------------------------------------------------------------------------------------------------
int rx_pick_launch(uint32_t kernel, uint32_t size, uint32_t n_streams, int cu_count, size_t lds_per_cu,
                   RxParams* p, RxLaunchCfg* cfg) {
  (void)cu_count;
  if (kernel == RX_KERNEL_AUTO) kernel = RX_KERNEL_SYM_WAVE;
  if (kernel != RX_KERNEL_CSR_WAVE && kernel != RX_KERNEL_SYM_WAVE) return RX_EINVAL;
  const uint32_t nw32 = (size + 31u) / 32u;
  p->nw32 = nw32;
  p->lds_words_per_stream = 2u * nw32 + 2u * RX_LIST_CAP;
  const size_t per_wave = (size_t)p->lds_words_per_stream * 4u;
  if (per_wave > lds_per_cu) return RX_ECAPACITY;  // automaton too large for an LDS-resident bitmask
  uint32_t wpb = 4;
  while (wpb > 1 && per_wave * wpb > lds_per_cu / 2) wpb >>= 1;
  cfg->kernel = kernel;
  cfg->block_threads = wpb * 64u;
  cfg->lds_bytes = (uint32_t)(per_wave * wpb);
  uint32_t blocks = (n_streams + wpb - 1) / wpb;
  cfg->grid_blocks = blocks ? blocks : 1;
  return RX_OK;
}

template <typename K>
static int launch_one(K kern, const RxParams& p, const RxLaunchCfg& cfg, hipStream_t s) {
  if (cfg.lds_bytes > 64u * 1024u) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(cfg.grid_blocks), dim3(cfg.block_threads), cfg.lds_bytes, s, p);
  return (int)hipGetLastError();
}

// returns a hipError_t value (0 = hipSuccess)
int rx_launch(const RxParams& p, const RxLaunchCfg& cfg, void* hip_stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
  switch (cfg.kernel) {
    case RX_KERNEL_CSR_WAVE:
      return cfg.stats ? launch_one(rx_csr_wave_kernel<true>, p, cfg, s)
                       : launch_one(rx_csr_wave_kernel<false>, p, cfg, s);
    case RX_KERNEL_SYM_WAVE:
      return cfg.stats ? launch_one(rx_sym_wave_kernel<true>, p, cfg, s)
                       : launch_one(rx_sym_wave_kernel<false>, p, cfg, s);
    default:
      return (int)hipErrorInvalidValue;
  }
}
