// rx_internal.hpp — shared between the host side (rx_host.cpp, rx_api.hip) and the gfx950 kernels
// (rx_kernels.hip).  Not part of the C-ABI; the boundary is include/rxmatch.h.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rxmatch.h"

// ---- slice-index entry encoding (derived at load time from the unchanged CSR table) ---------
// One u32 per (state, symbol): what row(state) — Design/FPGA.v:227-714 — yields for that byte.
//   0                       no edge of this state carries the symbol
//   RXE_INLINE | tgt        exactly one non-self target, held inline (bits 23:0)
//   RXE_ACCEPT              the inline target is an accept state (empty row, FPGA.v:210-226)
//   RXE_SELF                the state has an edge to ITSELF on this symbol (e.g. the `.*` state)
//   RXE_OVF | off           >= 2 non-self targets: ovf[off] = count, ovf[off+1..] = targets
//                           (each target word may carry RXE_ACCEPT)
static constexpr uint32_t RXE_INLINE = 0x80000000u;
static constexpr uint32_t RXE_ACCEPT = 0x40000000u;
static constexpr uint32_t RXE_SELF = 0x20000000u;
static constexpr uint32_t RXE_OVF = 0x10000000u;
static constexpr uint32_t RXE_PIN = 0x08000000u;   // target is the pinned state (see RxHostNfa::pin_state)
// The target can be reached on this byte class from a SECOND state that may be active at the same time (another
// predecessor on the class, or itself through a self-loop): only such a target can already be in the next set when it
// is inserted.  Per-class index, overflow lists and the folding table carry the flag; kernels that deduplicate every
// insertion through a filter ignore it, the register-resident single-stream kernel checks only flagged targets.
static constexpr uint32_t RXE_MAYDUP = 0x04000000u;
static constexpr uint32_t RXR_NEED = 0x80000000u;  // register kernel's fast word (RxParams::regidx): any of the three below
static constexpr uint32_t RXR_ACC = 0x40000000u;   //   bits 23:0 name an accept state
static constexpr uint32_t RXR_EXTRA = 0x20000000u; //   the state stays AND has one target nothing else can reach: the target needs a free lane
static constexpr uint32_t RXR_DUPC = 0x10000000u;  //   its one target may already be in the next set (RXE_MAYDUP): check, then a free lane
static constexpr uint32_t RXR_OVFL = 0x08000000u;  //   several targets on the byte (overflow list in the slice word)
static constexpr uint32_t RXE_TGT_MASK = 0x00FFFFFFu;

// Active-list entry (LDS): state id in bits 23:0, RXE_ACCEPT if the state is an accept state.

// Every kernel keeps (or can fall back to) two size-bit bitmasks + two RX_LIST_CAP lists per stream in one CU's 160 KB
// of LDS: 2*(size/8) + 1 KB <= 160 KB.  Larger automata are refused at load time, before any index is built.
static constexpr uint32_t RX_MAX_STATES = 650000;
static constexpr uint32_t RX_LIST_CAP = 128;  // sparse active-list capacity per stream (entries), at least
static constexpr uint32_t RX_LIST_CAP_MAX = 512;  // ... and at most (rx_pick_launch: what four wavefronts per block leave room for)
static constexpr uint32_t RX_SMALL_DEG = 8;   // CSR kernel: rows up to this length are scanned per lane

// Kernel argument block (passed by value).
struct RxParams {
  // automaton in HBM — `words` is the .coe content unchanged: row_ptr = words, col = words+size+1
  const uint32_t* words;
  const uint32_t* symidx;       // [size][256] slice index (null for the CSR kernel)
  const uint32_t* symidx_c;     // [size][n_classes] the same index per byte class (pack kernel)
  // look-ahead pruning of multi-target rows (pack kernel, see rx_host.cpp): index whose RXE_OVF payload is a LIST
  // NUMBER, and per list (n_classes + 1) directory words (ovf offset << 8 | min(count, 255)): entry c = the targets
  // that survive a next byte of class c, entry n_classes = the full list.  Both null when there is nothing to prune.
  // prune_narrow = 1 (automata with <= 65 536 states and lists): symidx_p keeps target / list number in bits 15:0 and, for an
  // inline target, bits 23:16 = which byte classes (mod 8) the TARGET has an edge on (all ones for an accept state): an
  // inline target whose bit for the stream's next byte is clear dies at once and is not inserted either.
  const uint32_t* symidx_p;
  const uint32_t* ovf_dir;
  uint32_t prune_narrow;
  // always-on-state folding (pack kernel FOLD builds, see rx_host.cpp): what the pinned `.*` state's row emits on a byte
  // of class c when the stream's next byte has class n — pin_tab[c * pin_cols + n], column n_classes = no look-ahead (the
  // full slice).  Entry: 0 | RXE_INLINE|target[|RXE_ACCEPT] | RXE_OVF|offset into `ovf`.  Null when the automaton has no
  // foldable state.
  const uint32_t* pin_tab;
  uint32_t pin_cols;            // n_classes + 1
  // register kernel's index: [(size + 1)][n_classes] pairs {fast word, slice word}.  Fast word: bits 23:0 = what the lane
  // holds after the byte (the state itself if it loops, its one target if nothing else can reach that, else the id
  // `size` = free), RXR_NEED = something needs a lane of its own (see rx_sym_reg_kernel), RXR_ACC = bits 23:0 name an
  // accept state.  Row `size` is all {size, 0}.  Built with the folded state's targets dropped iff pin_tab exists.
  // reg_tmask = 0xFFFF (automata with < 65 536 states): the value is in bits 15:0 and bits 23:16 say for which NEXT byte
  // the need is real — bit (n & 7) set iff the single target that wants a lane has an edge on some class n' with
  // n' & 7 == n & 7 (all ones: accept state, or several targets) — one byte of look-ahead for single targets, as the
  // folding table has it for the folded state's.  reg_tmask = 0xFFFFFF otherwise (bits 23:0 value, no look-ahead).
  const uint32_t* regidx;
  uint32_t reg_tmask;
  const uint32_t* byte_class;   // [64] words = 256 bytes: class id of every input byte
  uint32_t n_classes;
  const uint32_t* ovf;          // overflow target lists of the slice index
  const uint32_t* accept_bits;  // [nw32] bit i set iff deg(i) == 0
  uint32_t size;
  uint32_t nw32;                // ceil(size/32)
  // input streams
  const uint8_t* bytes;
  uint64_t stride;
  uint32_t n_streams;
  uint32_t stream_len;
  uint32_t n_passes;            // accept checks per stream
  uint32_t n_consume;           // bytes consumed per stream (= n_passes in tb-compat, N in full mode)
  uint32_t k_base;
  uint32_t stream_base;         // added to every reported stream id (rx_plan_run launches a batch in chunks of streams)
  uint32_t state0_entry;        // list entry for reset state 0 (accept flag folded in)
  const uint32_t* init_active;  // optional [n_streams][2*nw64] start bitmasks (u64 rows viewed as u32)
  uint32_t nw64x2;              // u32 words per init/final row = 2*ceil(size/64)
  // outputs
  rx_event* events;
  uint32_t events_cap;
  // where accept events take their slots in `events` (capacity events_cap): &counters[0] for a plain launch; ONE word shared
  // by every block of an rx_plan_run call, so that the blocks fill one caller-sized buffer in launch order
  unsigned long long* ev_count;
  unsigned long long* counters; // [0] n_events [1] sum_active [2] sum_edges [3] spilled streams [4] pair clock cost
                                // pack statistics build: [5] entries on multi-target rows [6] of their targets, dead at once [7] its own active
  // The plan keeps TWO sets of {counters[16], match_count_total[size]} and alternates between them: the kernel of launch
  // n accumulates into one set and its first block zeroes the other for launch n+1, so that no reset sits between two
  // launches on the stream.  Null / 0 for launches that must not do that (the resume launch, AUTO's probe).
  unsigned long long* zero_next;
  uint32_t zero_words;
  uint32_t* match_count;        // [n_streams][size] or null
  unsigned long long* match_count_total; // [size] or null
  uint32_t* anymatch;           // [n_streams][anymatch_stride] or null
  uint32_t anymatch_stride;
  uint32_t* final_active;       // [n_streams][nw64x2] or null
  // The final sets as compact lists, written by the match kernel itself (pack kernel and the wave kernel that finishes its
  // hand-offs; rx_plan_run on request): states of stream s ascending at fin_states[fin_off[s] .. + fin_cnt[s]), space taken
  // from *fin_count (one atomic per wavefront), entries at or beyond fin_cap not written.  When set, no rows are written.
  uint32_t* fin_states;
  uint32_t* fin_off;
  uint32_t* fin_cnt;
  unsigned long long* fin_count;
  uint32_t fin_cap;
  // LDS carve
  uint32_t lds_words_per_stream;
  // spill hand-off: group kernel -> wave kernel (streams whose active set outgrew the group's list)
  unsigned long long* spill_count;  // == &counters[3]
  uint32_t* spill_streams;          // [n_streams] stream id per spill slot
  uint32_t* spill_k;                // [n_streams] pass at which the stream must be resumed
  uint32_t* spill_rows;             // [n_streams][nw64x2] S_k of the spilled stream as a bitmask row
  uint32_t resume;                  // wave kernel: 1 = walk the spill list instead of all streams
  uint32_t pin_state;               // group kernel: pinned state id, 0xFFFFFFFF = none
  uint32_t pin_degree;              // its row length (for the algorithmic-byte statistics)
  uint32_t pair_cycles;             // stats build: also sum the FPGA clock cost of stream pairs (2q, 2q+1)
  // lazy DFA cache (rx_dfa_kernel): persistent per automaton and device, grown on the device
  uint32_t* dfa_trans;              // [pool_chunks][n_classes]: 0 unknown, DFA_EXIT, else next id | DFA_ACC
  uint32_t* dfa_pool;               // [pool_chunks][32]: state id = index of its first 32-word chunk
  uint32_t* dfa_hash;               // [hash_mask+1] open-addressing table of ids
  uint32_t* dfa_hdr;                // [0] unused [1] next free chunk [2] states created [3] transitions built
  uint32_t dfa_pool_chunks;
  uint32_t dfa_hash_mask;
};

static constexpr uint32_t RX_GROUP_CAP = 24;     // group kernel: active-list capacity per stream
static constexpr uint32_t DFA_ACC = 0x80000000u;   // transition value: the target set contains an accept state
static constexpr uint32_t DFA_EXIT = 0xFFFFFFFFu;  // transition value: target set not representable -> NFA kernel
static constexpr uint32_t DFA_MAXM = 61;           // members per DFA state (two 32-word chunks minus header)
static constexpr uint32_t DFA_HDR_WORDS = 3;       // chunk header: count, sum of degrees, has-accept
static constexpr uint32_t RX_PACK_CAP = 192;      // pack kernel: wave-wide active-list capacity (entries)
static constexpr uint32_t RX_GROUP_FILTER_WORDS = 32;  // 1024-bit hashed dedup filter per stream

struct RxLaunchCfg {
  uint32_t kernel;         // RX_KERNEL_* (resolved, never AUTO)
  uint32_t group_lanes;    // SYM_GROUP: lanes per stream (4 or 8)
  uint32_t block_threads;
  uint32_t grid_blocks;
  uint32_t lds_bytes;      // dynamic LDS per block
  int cu_count;
  size_t lds_per_cu;
  bool stats;
  bool prune;              // SYM_PACK: look-ahead pruning of multi-target rows (needs RxParams::ovf_dir)
  bool fold;               // SYM_PACK: the pinned `.*` state is folded out of the lists (needs RxParams::pin_tab)
  bool verbose;            // rx_opts.flags & RX_OPT_VERBOSE: print the launch geometry
  bool reg_skip;           // SYM_REG: the build that steps over groups of passes in which nothing is active
  bool profile_pack;       // rx_opts.flags & RX_OPT_PROFILE_PACK: stamped diagnostic build of the pack kernel (S=16)
};

// rx_kernels.hip
int rx_pick_launch(uint32_t kernel, uint32_t size, uint32_t n_streams, int cu_count, size_t lds_per_cu,
                   RxParams* p, RxLaunchCfg* cfg);
int rx_launch(const RxParams& p, const RxLaunchCfg& cfg, void* hip_stream);
// Final sets as compact lists: rows[n_streams][row_words] (bitmask rows as the kernels leave them) -> per stream its states
// in ascending order at states[off[s] .. off[s] + cnt[s]); off is relative to `states`; *counter (zero before the launch)
// ends as the number of entries the sets need, entries beyond `cap` are not written.
int rx_launch_final_compact(const uint32_t* rows, uint32_t n_streams, uint32_t row_words, uint32_t* states, uint32_t cap,
                            uint32_t* off, uint32_t* cnt, unsigned long long* counter, void* hip_stream);

// ---- host-side automaton (rx_host.cpp; no HIP in here) ---------------------------------------
struct RxHostNfa {
  std::vector<uint32_t> words;  // exactly the .coe words incl. pad
  uint32_t size = 0, nnz = 0, n_accept = 0, max_degree = 0;
  // derived
  std::vector<uint32_t> symidx;       // size*256
  std::vector<uint32_t> ovf;          // ovf[0] unused so that offset 0 never occurs
  std::vector<uint32_t> accept_bits;  // ceil(size/32)
  // bytes with identical slice-index columns form one class; symidx_c is the index stored per class
  uint8_t byte_class[256] = {0};
  uint32_t n_classes = 0;
  std::vector<uint32_t> symidx_c;     // size*n_classes
  // A state with a self-loop on all 256 bytes stays active forever once entered.  The one state 0 feeds on
  // the most bytes (snort_16: state 1, the `.*` state) is "pinned": targets equal to it carry RXE_PIN.
  uint32_t pin_state = 0xFFFFFFFFu;
  // look-ahead pruning tables (RxParams::symidx_p / ovf_dir); empty when the automaton has no multi-target rows
  std::vector<uint32_t> symidx_p, ovf_dir;
  bool prune_narrow = false;  // symidx_p carries the inline targets' next-class bits (RxParams::prune_narrow)
  // folding table of the pinned state (RxParams::pin_tab), n_classes * (n_classes + 1) words; empty unless state 0
  // enters the pinned state on every byte (then every stream that starts from reset holds it from pass 1 on)
  std::vector<uint32_t> pin_tab;
  // RxParams::regidx; empty for automata whose table would exceed 256 MB (the register kernel is then not offered)
  std::vector<uint32_t> regidx;
  uint32_t reg_tmask = RXE_TGT_MASK;  // RxParams::reg_tmask
  const uint32_t* row_ptr() const { return words.data(); }
  const uint32_t* col() const { return words.data() + size + 1; }
};

int rxh_parse_coe_text(const char* text, size_t len, std::vector<uint32_t>* words);
int rxh_read_file(const char* path, std::string* out);
int rxh_infer_size(const uint32_t* W, size_t nwords, uint32_t* size);
int rxh_validate(const uint32_t* W, size_t nwords, uint32_t size);
int rxh_build(const uint32_t* W, size_t nwords, uint32_t size_or_0, RxHostNfa* out);
// rx_compile.cpp
int rxc_compile(const char* const* patterns, size_t n, uint32_t flags, std::vector<uint32_t>* words,
                std::vector<int32_t>* accept_pattern, std::string* err);
int rxc_write_coe(const char* path, const std::vector<uint32_t>& words);
int rxh_parse_mem_text(const char* text, size_t len, std::vector<uint8_t>* bytes);
