/*
 * rxmatch.h — C-ABI drop-in boundary of the MI355X-native CSR-NFA matcher (librxmatch.so).
 *
 * Plain C: extern "C", plain pointers and sizes, no torch / C++ types.  Every entry point returns
 * 0 (RX_OK) or a negative RX_E* code; rx_strerror() names it.  Nothing throws across the boundary.
 *
 * What each entry point replaces in the reference (paths under /root/reference):
 *
 *   rx_nfa_load_coe / rx_nfa_from_words
 *        the Block-RAM initialisation of `design_1_wrapper` from the Block_Mem .coe files
 *        (instantiated Simulation/testbench_BLK_Mem.sv:89-92, Design/top.v:10-13) plus the
 *        `size` port / `size_range` parameter (Design/FPGA.v:26,46; testbench_BLK_Mem.sv:20,39).
 *        The 128-bit lines are kept UNCHANGED as one u32 word array: row_ptr = W[0..size],
 *        packed (symbol<<24|target) edges = W[size+1 ..] (FPGA.v:773,793,881-898).
 *   rx_trace_load_mem
 *        `$readmemh("input_trace_lo.mem", data_read_lo)` (testbench_BLK_Mem.sv:34-35).
 *   rx_match / rx_plan_*
 *        module CSR_traversal (Design/FPGA.v:23-43 ports, :115-768 per-clock active-state
 *        update) driven by the byte feeder + match counters of Blk_Mem_tb
 *        (testbench_BLK_Mem.sv:49-87): `input_char/input_char_2` become rows of `bytes`,
 *        the `accepting_match_flag` pulses qualified by `i` become rx_event records /
 *        match_count / the per-pass any-match bitmap.
 *   rx_match_sharded
 *        nothing in the reference (it has one device); contiguous stream blocks per GPU, no
 *        collective.
 *
 * Pass indexing follows the reference: pass k examines the active set S_k (S_0 = {0}) while
 * input byte c[k] is on `input_char`; an accept event (k, state) means `state` (a row with no
 * out-edges, FPGA.v:210-226) was active in pass k, i.e. a match ENDED on byte c[k-1].
 *   RX_MODE_FULL      passes k = 0 .. N      (N+1 passes; pass N only checks accepts)
 *   RX_MODE_TB_COMPAT passes k = 0 .. N-2    (what Blk_Mem_tb observes before $finish,
 *                                             testbench_BLK_Mem.sv:71-86)
 *
 * The product path is HIP only.  There is no CPU fallback: without a usable HIP device every
 * compute entry point fails with RX_ENODEVICE.
 */
#ifndef RXMATCH_H
#define RXMATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RX_ABI_VERSION 3

/* ---- error codes ------------------------------------------------------------------------ */
enum {
  RX_OK = 0,
  RX_EINVAL = -1,    /* bad argument (NULL, zero size, stride < stream_len, ...) */
  RX_EIO = -2,       /* file could not be opened / read */
  RX_EFORMAT = -3,   /* .coe / .mem text is malformed */
  RX_ENFA = -4,      /* word array is not a valid CSR automaton (row_ptr / targets / size) — or size 0 was passed and
                        more than one size fits the words (e.g. a last edge word of 0 reads like padding): pass the size,
                        as the reference does (size_range, testbench_BLK_Mem.sv:20) */
  RX_ENOMEM = -5,    /* host or device allocation failed */
  RX_ENODEVICE = -6, /* no usable HIP device (there is NO CPU fallback) */
  RX_EHIP = -7,      /* a HIP runtime call failed; rx_last_hip_error() has the text */
  RX_ECAPACITY = -8, /* automaton or batch exceeds a kernel limit */
  RX_ESTATE = -9     /* plan used out of order (launch before input, download before launch) */
};
const char* rx_strerror(int code);
/* Text of the last HIP failure on the calling thread ("" if none). */
const char* rx_last_hip_error(void);
int rx_abi_version(void);

/* ---- automaton -------------------------------------------------------------------------- */
typedef struct rx_nfa rx_nfa; /* immutable after load => rx_match is re-entrant per rx_nfa */

typedef struct rx_nfa_info {
  uint32_t size;        /* number of NFA states                                   */
  uint32_t nnz;         /* number of edges = row_ptr[size]                        */
  uint32_t n_accept;    /* states with an empty row (accept states)               */
  uint32_t n_words;     /* u32 words in the table incl. 0-3 pad words             */
  uint32_t max_degree;  /* longest row                                            */
  uint32_t n_bitmask_words64; /* ceil(size/64): u64 words of one active-state bitmask */
} rx_nfa_info;

/* size_or_0 = 0 infers the state count from the table (unique s with W[0]=0, W[0..s]
 * non-decreasing, 0 <= nwords-(W[s]+s+1) <= 3 zero pad words, all targets < s). */
int rx_nfa_load_coe(const char* path, uint32_t size_or_0, rx_nfa** out);
int rx_nfa_from_words(const uint32_t* words, size_t nwords, uint32_t size_or_0, rx_nfa** out);
int rx_nfa_get_info(const rx_nfa* nfa, rx_nfa_info* info);
/* The table exactly as it sits in HBM: row_ptr = words, edges = words + size + 1. */
const uint32_t* rx_nfa_words(const rx_nfa* nfa, size_t* nwords);
void rx_nfa_free(rx_nfa* nfa);

/* ---- regex list -> CSR automaton (the step BEFORE the path; the reference ships tables only) ---- */
enum { RX_RE_ICASE = 1, RX_RE_DOTALL = 2 };
/* Compiles n patterns (PCRE subset, optionally written /regex/flags with flags i,s) into ONE automaton in
 * the reference's table conventions (state 0 -> `.*` state 1 on every byte, accept = empty row, edge word
 * = symbol<<24|target), ready for rx_match or rx_nfa_save_coe.  On RX_EFORMAT errbuf names the pattern. */
int rx_compile_patterns(const char* const* patterns, size_t n, uint32_t flags, rx_nfa** out, char* errbuf,
                        size_t errbuf_len);
/* Pattern index an accept state reports (-1: not an accept state / table was not compiled here). */
int rx_nfa_accept_pattern(const rx_nfa* nfa, uint32_t state, int32_t* pattern_index);
/* Writes the table as a Xilinx .coe in the layout of Block_Mem/CSR_BlockMem_snort_16.coe
 * (radix 16, one 128-bit line per row of text) so it can initialise the reference's ROM. */
int rx_nfa_save_coe(const rx_nfa* nfa, const char* path);

/* ---- lazy-DFA cache of RX_KERNEL_DFA ------------------------------------------------------- */
/* Number of DFA states / transitions built so far on `device` (0/0 before the first DFA launch). */
int rx_nfa_dfa_info(const rx_nfa* nfa, int device, uint64_t* n_states, uint64_t* n_transitions);
/* Forget everything built so far on `device` (the next DFA launch starts cold). */
int rx_nfa_dfa_reset(const rx_nfa* nfa, int device);

/* ---- traces ----------------------------------------------------------------------------- */
/* $readmemh text (one 1-2 digit hex byte per line) -> malloc'ed byte array; release with rx_free. */
int rx_trace_load_mem(const char* path, uint8_t** bytes, size_t* n);
void rx_free(void* p);

/* ---- options / results ------------------------------------------------------------------ */
enum { RX_MODE_FULL = 0, RX_MODE_TB_COMPAT = 1 };

enum {
  RX_KERNEL_AUTO = 0,     /* probe, then choose: on the first launch for a batch (and again on every 32nd batch
                             of the same shape) the plan runs the pack kernel's statistics build over a corner of
                             it (512 K stream-bytes: up to 4 KB of <= 512 streams; this synchronises the stream) and picks
                             RX_KERNEL_SYM_PACK — streams per wavefront ~ 33 / (list entries per stream), look-ahead
                             pruning of multi-target rows when it removes >= 10 % of the entries, the FOLD build
                             (always-on `.*` state out of the lists, idle passes stepped over) when that leaves the
                             lists nearly empty — or, when even the long-list form of the pack kernel hands streams
                             off, RX_KERNEL_SYM_WAVE.  Batches of at most 16 wavefronts per SIMD (one stream each):
                             the pack kernel's choice and the two builds of RX_KERNEL_SYM_REG run the batch once,
                             timed, and the fastest is taken; up to 4 streams: always RX_KERNEL_SYM_REG            */
  RX_KERNEL_CSR_WAVE = 1, /* wavefront-per-stream over the state-major CSR exactly as loaded         */
  RX_KERNEL_SYM_WAVE = 2, /* wavefront-per-stream over the per-(state,symbol) slice index; also the kernel that finishes
                             the streams the kernels below hand off (few of them: one workgroup per stream)   */
  RX_KERNEL_SYM_GROUP = 3, /* G lanes per stream (64/G streams per wavefront), slice index;
                             rx_opts.group_lanes = G (1/2/4/8/16, default 4); streams whose active set
                             outgrows the group's list are finished by RX_KERNEL_SYM_WAVE in the same call */
  RX_KERNEL_SYM_PACK = 4  /* S streams per wavefront, the 64 lanes assigned dynamically to one wave-wide
                             list of (stream,state) entries; rx_opts.group_lanes = S (2/4/8/11/12/13/16/20/22/24/32,
                             default 16; 2 and 4 use 512-entry lists); same hand-off to RX_KERNEL_SYM_WAVE */
  ,
  RX_KERNEL_DFA = 5       /* opt-in: lazy DFA, one LANE per stream and one table lookup per byte; the subset-
                             construction cache lives in HBM per automaton and device, is grown on the device
                             and persists across launches (rx_nfa_dfa_reset clears it); sets it cannot hold
                             are handed to RX_KERNEL_SYM_WAVE.  Never chosen by RX_KERNEL_AUTO.             */
  ,
  RX_KERNEL_SYM_REG = 6   /* one wavefront per stream, the active set register-resident (one state per lane, updated
                             in place: no LDS list, no filter), the always-on `.*` state folded out when the automaton
                             has one: the kernel for FEW LONG streams — the reference's own run is one lock-step pair
                             (testbench_BLK_Mem.sv:49-87) — and for small batches, where the latency of a pass is
                             what counts (see RX_KERNEL_AUTO).  A second build steps over groups of passes in which no
                             state is active (RX_OPT_REG_NO_SKIP selects the plain one).  More than 64 active states:
                             hand-off to RX_KERNEL_SYM_WAVE.  With collect_stats or a caller-supplied start set
                             RX_KERNEL_SYM_WAVE runs instead.                                                  */
};

typedef struct rx_opts {
  uint32_t struct_size; /* = sizeof(rx_opts).  0 = the ABI-1 layout (everything before `flags`):
                           what "this version" meant when 0 was introduced                */
  int32_t device;       /* HIP device ordinal; -1 = the calling thread's current device  */
  uint32_t mode;        /* RX_MODE_*                                                     */
  uint32_t kernel;      /* RX_KERNEL_*                                                   */
  void* stream;         /* hipStream_t to launch on; NULL = the default stream           */
  uint64_t k_base;      /* added to every reported pass index (chunked streaming).  rx_event.k is
                           32 bits: k_base + passes of the batch must stay <= 2^32, else RX_EINVAL */
  uint32_t collect_stats; /* 1: also accumulate rx_stats.sum_active/sum_edges on the device;
                             2: additionally treat streams (2q, 2q+1) as Blk_Mem_tb's lock-step pair
                                and predict its clock count -> rx_stats.tb_cycles (n_streams even)  */
  uint32_t group_lanes;   /* SYM_GROUP: lanes per stream; SYM_PACK: streams per wavefront; 0 = default */
  uint32_t flags;         /* RX_OPT_* bits (ABI >= 2; a caller whose struct_size ends before this field gets 0) */
} rx_opts;

/* rx_opts.flags — A/B and diagnostic switches.  They are read when the plan is created, never from the
 * environment and never on the launch path. */
enum {
  RX_OPT_NO_PRUNE = 1u,     /* SYM_PACK: never use look-ahead pruning of multi-target rows                       */
  RX_OPT_FORCE_PRUNE = 2u,  /* SYM_PACK: always use it when the automaton has such rows (batches too small to probe) */
  RX_OPT_VERBOSE = 4u,      /* print AUTO's probe figures / choice and the launch geometry to stderr               */
  RX_OPT_PROFILE_PACK = 8u, /* SYM_PACK S=16: the s_memtime-stamped diagnostic build (phase shares on stderr)     */
  RX_OPT_NO_FOLD = 16u,     /* SYM_PACK: never fold the always-on `.*` state out of the lists                     */
  RX_OPT_FORCE_FOLD = 32u,  /* SYM_PACK: fold it whenever the automaton has such a state                          */
  RX_OPT_REG_NO_SKIP = 64u, /* SYM_REG: the build that does not step over passes in which no state is active (A/B runs) */
  RX_OPT_INJECT_RUN_FAULT = 128u, /* test hook: rx_plan_run fails with RX_EHIP once block 0's kernels and copies are in
                                     flight — exercises the drain-before-error-return path                           */
  RX_OPT_NO_PROBE = 256u    /* RX_KERNEL_AUTO never probes inside rx_plan_launch / rx_plan_run: no sample launches, no
                               timed candidates, no stream synchronisation.  The decision is the one rx_plan_tune made
                               for the shape, or a default (SYM_PACK, 16 streams per wavefront; SYM_REG up to 4 streams) */
};

/* One accept pulse: `state` was active and accepting in pass `k` of stream `stream`.
 * 12 bytes — the B_out event term of SURVEY.md §8(d). */
typedef struct rx_event {
  uint32_t stream;
  uint32_t k;
  uint32_t state;
} rx_event;

typedef struct rx_stats {
  uint64_t n_passes;     /* passes executed per stream                                    */
  uint64_t n_events;     /* accept events over all streams (may exceed events_cap)        */
  uint64_t sum_active;   /* sum over passes of |S_k|        (collect_stats only, else 0)  */
  uint64_t sum_edges;    /* sum over passes of sum deg(i)   (collect_stats only, else 0)  */
  uint64_t alg_bytes;    /* passes*1 + 8*sum_active + 4*sum_edges +
                            ceil(passes/8) + 12*n_events   (collect_stats only, else 0)   */
  double kernel_ms;      /* hipEvent time of the match kernel(s) of the last launch       */
  double h2d_ms, d2h_ms; /* hipEvent time of the copies rx_match() issued (0 for plans)   */
  uint32_t kernel_used;  /* RX_KERNEL_* actually launched                                 */
  uint32_t n_launches;
  uint64_t tb_cycles;    /* collect_stats == 2: what `$display("Total no. cycles: %d", cycles)`
                            (testbench_BLK_Mem.sv:84) prints for the pair(s), summed over pairs;
                            0 if unavailable                                               */
  /* ABI >= 2 (written only when the caller's rx_result.struct_size covers them) */
  uint32_t lanes_used;   /* SYM_GROUP: lanes per stream; SYM_PACK: streams per wavefront; else 0 */
  uint32_t variant;      /* RX_VARIANT_* bits of the build that ran                              */
} rx_stats;
enum {
  RX_VARIANT_STATS = 1u,  /* the statistics build (collect_stats)                          */
  RX_VARIANT_PRUNE = 2u,  /* look-ahead pruning of multi-target rows                       */
  RX_VARIANT_FOLD = 4u    /* the always-on `.*` state folded out of the lists              */
};

/* All output arrays are caller-allocated and optional (NULL = not wanted). */
typedef struct rx_result {
  uint32_t struct_size;      /* = sizeof(rx_result); 0 = the ABI-1 layout (up to stats.tb_cycles:
                                no lanes_used / variant, no compact final sets)           */
  uint32_t events_overflow;  /* out: 1 if n_events > events_cap (events[] holds the first
                                events_cap in (stream,k,state) order of those captured)   */
  rx_event* events;          /* [events_cap], returned sorted by (stream, k, state)       */
  size_t events_cap;
  size_t n_events;           /* out: events written (<= events_cap)                       */
  uint32_t* match_count;     /* [n_streams][size]: pulses per state, full 32-bit; the
                                testbench's 10-bit wrap (testbench_BLK_Mem.sv:21-22) is
                                applied by the report tool, not here                      */
  uint64_t* match_count_total; /* [size]: match_count summed over streams                 */
  uint32_t* anymatch;        /* [n_streams][anymatch_stride]: bit k (word k>>5, bit k&31)
                                set iff some accept state was active in pass k            */
  size_t anymatch_stride;    /* in u32 words, >= ceil(n_passes/32); a multiple of 8 that equals
                                ceil(ceil(max passes / 32) / 8) * 8 of the plan is copied flat,
                                any other pitch row by row                                 */
  uint64_t* final_active;    /* [n_streams][ceil(size/64)]: S after the last pass's byte  */
  rx_stats stats;            /* out */
  /* The same final sets as compact lists (the plan must have been created with want_final): the states of
   * stream s are final_states[final_off[s] .. final_off[s] + final_cnt[s]), ascending.  The bitmask rows are 1.2 KB per
   * stream for snort_16 whatever they hold — 90 % of what a call downloads; the lists are ~12 bytes + 4 per active state.
   * All three arrays or none; final_active may be NULL then.  rx_plan_run and rx_match from reset; rx_match with a start
   * set, rx_match_sharded and rx_plan_download return RX_EINVAL / ignore them.  A caller whose struct_size ends before
   * these fields gets rows. */
  uint32_t* final_states;    /* [final_states_cap] */
  uint32_t* final_off;       /* [n_streams] */
  uint32_t* final_cnt;       /* [n_streams] */
  size_t final_states_cap;
  size_t n_final_states;     /* out: entries written to final_states */
  uint32_t final_states_overflow; /* out: 1 if the sets did not fit final_states_cap (final_cnt is exact, final_off then
                                     names only the part that was written) */
  uint32_t reserved0;
} rx_result;

/* ---- one-shot match over host buffers ---------------------------------------------------- */
/* bytes: n_streams rows of stream_len bytes, row s at bytes + s*stride.  init_active (optional,
 * [n_streams][ceil(size/64)]) replaces the reset state S_0={0} (FPGA.v:134-147) per stream. */
int rx_match(const rx_nfa* nfa, const uint8_t* bytes, size_t n_streams, size_t stream_len,
             size_t stride, const uint64_t* init_active, const rx_opts* opts, rx_result* res);

/* Same, streams split into contiguous blocks over n_devices GPUs (one host thread each, the
 * table replicated per device, no collective).  devices == NULL => ordinals 0..n_devices-1. */
int rx_match_sharded(const rx_nfa* nfa, const uint8_t* bytes, size_t n_streams, size_t stream_len,
                     size_t stride, const int* devices, int n_devices, const rx_opts* opts,
                     rx_result* res);

/* ---- resident plan: inputs stay in HBM between launches (serving / benchmarking) --------- */
typedef struct rx_plan rx_plan;

int rx_plan_create(const rx_nfa* nfa, const rx_opts* opts, size_t max_streams,
                   size_t max_stream_len, size_t events_cap, uint32_t want_match_count,
                   uint32_t want_anymatch, uint32_t want_final, rx_plan** out);
/* Copy host bytes into the plan's own HBM input buffer (row stride = stride).  The copy is enqueued on the plan's
 * stream: from pageable memory it has completed when the call returns, from page-locked memory (rx_host_register,
 * hipHostMalloc) it is asynchronous — leave the buffer untouched until rx_plan_sync / rx_plan_download returns. */
int rx_plan_upload(rx_plan* plan, const uint8_t* bytes, size_t n_streams, size_t stream_len,
                   size_t stride);
/* Use a caller-owned DEVICE buffer as the input (e.g. a torch tensor's data_ptr). */
int rx_plan_set_device_input(rx_plan* plan, const void* device_bytes, size_t n_streams,
                             size_t stream_len, size_t stride);
/* Optional per-stream start state (host array; copied — bits beyond `size` cleared — before the call returns) for the
 * batch given last; every new input (rx_plan_upload / rx_plan_set_device_input / rx_plan_run) restores reset, and so
 * does NULL. */
int rx_plan_set_init_active(rx_plan* plan, const uint64_t* init_active);
/* Enqueue result-reset + the match kernel on the plan's stream, bracketed by hipEvents.  With RX_KERNEL_AUTO the first
 * launch for a batch shape (and every 32nd of the same shape) first probes — sample launches and a stream synchronisation
 * — unless the shape was tuned (rx_plan_tune) or the plan was created with RX_OPT_NO_PROBE: then this call only enqueues. */
int rx_plan_launch(rx_plan* plan);
/* Run RX_KERNEL_AUTO's probes NOW for the batch the plan holds (input set, nothing else needed) and pin the decision to
 * its shape bucket (ceil log2 of stream count and length): later launches of that shape never probe and never
 * synchronise the stream.  Serving recipe: tune once per shape at start-up, create the plan with RX_OPT_NO_PROBE so that
 * an untuned shape falls back to the default instead of probing.  Blocks until the probes have finished. */
int rx_plan_tune(rx_plan* plan);
/* Non-blocking: which of the plan's streams still have work queued — bit 0 uploads, bit 1 kernels, bit 2 downloads of
 * rx_plan_run, bit 3 the launch stream (rx_opts.stream).  0 after rx_plan_run has returned, also with an error. */
int rx_plan_busy(rx_plan* plan, uint32_t* busy);
/* Wait for the last launch; kernel_ms (optional) = its hipEvent duration. */
int rx_plan_sync(rx_plan* plan, double* kernel_ms);
/* hipEvent durations of every launch since the previous call (or plan creation): their number,
 * sum, minimum and maximum in ms.  Waits for the last launch.  Any pointer may be NULL. */
int rx_plan_kernel_times(rx_plan* plan, uint32_t* n_launches, double* sum_ms, double* min_ms,
                         double* max_ms);
/* Copy the last launch's results to the caller's arrays (sorted events, counts, ...). */
int rx_plan_download(rx_plan* plan, rx_result* res);
/* Host buffers in, host results out, in ONE call — the same results as rx_plan_upload + rx_plan_launch +
 * rx_plan_download, but pipelined: the batch is cut into up to 8 blocks of >= 32 768 streams that share three HIP
 * streams (uploads, kernels, downloads), so that the upload of block i+1, the kernel of block i and the download of
 * block i-1 overlap.  Host memory that is page-locked — registered with rx_host_register, or allocated pinned by the
 * caller — is read and written by DMA directly; pageable memory works too, at the speed of the runtime's staging copies.
 * The plan must have been created for at least n_streams / stream_len; the caller's start sets are not supported here
 * (streams start from reset).  Capacities hold for the WHOLE call: the blocks fill the plan's event buffer (and the
 * caller's final_states) one behind the other, so events_overflow / final_states_overflow are set only when the batch's
 * total exceeds the capacity, wherever in the batch the matches lie.  On ANY error return no copy is in flight any more:
 * the caller may release `bytes` and the result arrays at once. */
int rx_plan_run(rx_plan* plan, const uint8_t* bytes, size_t n_streams, size_t stream_len, size_t stride, rx_result* res);
void rx_plan_free(rx_plan* plan);

/* Page-lock / release a host buffer (hipHostRegister) so that rx_plan_run / rx_plan_upload / rx_plan_download move it
 * by DMA without a staging copy.  Register long-lived buffers once; registration itself costs about as much as
 * copying the buffer. */
int rx_host_register(void* ptr, size_t bytes);
int rx_host_unregister(void* ptr);

/* ---- device helpers (so callers need no HIP binding of their own) ------------------------ */
int rx_device_count(int* n);
int rx_device_name(int device, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* RXMATCH_H */
