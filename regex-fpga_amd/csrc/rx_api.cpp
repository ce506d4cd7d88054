// rx_api.cpp — the C-ABI of librxmatch.so (include/rxmatch.h): automaton handles, resident plans,
// one-shot and multi-GPU sharded matching.  Host C++ over the HIP runtime; the only compute path is
// the gfx950 kernels in rx_kernels.hip — there is no CPU fallback and no oracle code in here.
//
// Replaces Blk_Mem_tb's role (Simulation/testbench_BLK_Mem.sv:26-106): it owns the ROM image
// (now an HBM buffer), feeds bytes (now whole resident batches), and collects the
// accepting_match_flag pulses (now rx_event records, counters and bitmaps).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rx_internal.hpp"

// ---- errors ---------------------------------------------------------------------------------------
static thread_local std::string g_last_hip;

static int hip_fail(hipError_t e, const char* what) {
  g_last_hip = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
  if (e == hipErrorNoDevice || e == hipErrorInsufficientDriver || e == hipErrorInvalidDevice ||
      e == hipErrorNotInitialized || e == hipErrorInitializationError)
    return RX_ENODEVICE;
  if (e == hipErrorOutOfMemory) return RX_ENOMEM;
  return RX_EHIP;
}
#define HIPCHK(call)                                           \
  do {                                                         \
    hipError_t e_ = (call);                                    \
    if (e_ != hipSuccess) return hip_fail(e_, #call);          \
  } while (0)

// Nothing throws across the C boundary: allocation failures inside an entry point become RX_ENOMEM.
#define RX_TRY try {
#define RX_CATCH \
  } catch (const std::bad_alloc&) { return RX_ENOMEM; } catch (...) { return RX_ENOMEM; }

extern "C" const char* rx_strerror(int code) {
  switch (code) {
    case RX_OK: return "ok";
    case RX_EINVAL: return "invalid argument";
    case RX_EIO: return "file could not be read";
    case RX_EFORMAT: return "malformed input text (.coe / .mem / regex)";
    case RX_ENFA: return "word array is not a valid CSR automaton, or (size 0) its size cannot be inferred unambiguously: pass size";
    case RX_ENOMEM: return "out of memory";
    case RX_ENODEVICE: return "no usable HIP device (librxmatch has no CPU fallback)";
    case RX_EHIP: return "HIP runtime error (see rx_last_hip_error)";
    case RX_ECAPACITY: return "automaton or batch exceeds a kernel limit";
    case RX_ESTATE: return "plan used out of order";
    default: return "unknown error";
  }
}
extern "C" const char* rx_last_hip_error(void) { return g_last_hip.c_str(); }
extern "C" int rx_abi_version(void) { return RX_ABI_VERSION; }

// ---- automaton ------------------------------------------------------------------------------------
struct DevTables {
  uint32_t* words = nullptr;
  uint32_t* symidx = nullptr;
  uint32_t* ovf = nullptr;
  uint32_t* accept_bits = nullptr;
  uint32_t* symidx_c = nullptr;
  uint32_t *symidx_p = nullptr, *ovf_dir = nullptr;  // look-ahead pruning tables (pack kernel), may stay null
  uint32_t* pin_tab = nullptr;                       // folding table of the pinned state (pack kernel), may stay null
  uint32_t* regidx = nullptr;                        // register kernel's index, may stay null (huge automata)
  uint32_t* byte_class = nullptr;
  // lazy-DFA cache (allocated by the first RX_KERNEL_DFA launch)
  uint32_t *dfa_trans = nullptr, *dfa_pool = nullptr, *dfa_hash = nullptr, *dfa_hdr = nullptr;
  uint32_t dfa_pool_chunks = 0, dfa_hash_mask = 0;
  int cu_count = 0;
  size_t lds_per_cu = 0;
};

struct rx_nfa {
  RxHostNfa h;
  std::mutex mu;
  std::map<int, DevTables> dev;  // HBM copies, one per device, uploaded on first use
  std::vector<int32_t> accept_pattern;  // filled by rx_compile_patterns
  // AUTO, batches too small for a probe: do few of the cells the register kernel would place from hold lists?  (-1: not
  // looked at yet; the index is immutable, so the answer is computed once, under mu)
  int few_lists = -1;
};

extern "C" int rx_nfa_from_words(const uint32_t* words, size_t nwords, uint32_t size_or_0, rx_nfa** out) {
  RX_TRY
  if (!words || !out || nwords == 0) return RX_EINVAL;
  std::unique_ptr<rx_nfa> n(new rx_nfa());  // released on every error path, also when the index build throws
  int rc = rxh_build(words, nwords, size_or_0, &n->h);
  if (rc) return rc;
  *out = n.release();
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_nfa_load_coe(const char* path, uint32_t size_or_0, rx_nfa** out) {
  RX_TRY
  if (!path || !out) return RX_EINVAL;
  std::string txt;
  int rc = rxh_read_file(path, &txt);
  if (rc) return rc;
  std::vector<uint32_t> w;
  rc = rxh_parse_coe_text(txt.data(), txt.size(), &w);
  if (rc) return rc;
  return rx_nfa_from_words(w.data(), w.size(), size_or_0, out);
  RX_CATCH
}

extern "C" int rx_compile_patterns(const char* const* patterns, size_t n, uint32_t flags, rx_nfa** out, char* errbuf,
                                   size_t errbuf_len) {
  RX_TRY
  if (!patterns || !out || n == 0) return RX_EINVAL;
  std::vector<uint32_t> words;
  std::vector<int32_t> acc;
  std::string err;
  int rc = rxc_compile(patterns, n, flags, &words, &acc, &err);
  if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", err.c_str());
  if (rc) return rc;
  rc = rx_nfa_from_words(words.data(), words.size(), (uint32_t)acc.size(), out);
  if (rc) return rc;
  (*out)->accept_pattern = std::move(acc);
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_nfa_accept_pattern(const rx_nfa* nfa, uint32_t state, int32_t* pattern_index) {
  RX_TRY
  if (!nfa || !pattern_index || state >= nfa->h.size) return RX_EINVAL;
  *pattern_index = state < nfa->accept_pattern.size() ? nfa->accept_pattern[state] : -1;
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_nfa_save_coe(const rx_nfa* nfa, const char* path) {
  RX_TRY
  if (!nfa || !path) return RX_EINVAL;
  return rxc_write_coe(path, nfa->h.words);
  RX_CATCH
}

extern "C" int rx_nfa_get_info(const rx_nfa* nfa, rx_nfa_info* info) {
  RX_TRY
  if (!nfa || !info) return RX_EINVAL;
  info->size = nfa->h.size;
  info->nnz = nfa->h.nnz;
  info->n_accept = nfa->h.n_accept;
  info->n_words = (uint32_t)nfa->h.words.size();
  info->max_degree = nfa->h.max_degree;
  info->n_bitmask_words64 = (nfa->h.size + 63u) / 64u;
  return RX_OK;
  RX_CATCH
}

extern "C" const uint32_t* rx_nfa_words(const rx_nfa* nfa, size_t* nwords) {
  if (!nfa) return nullptr;
  if (nwords) *nwords = nfa->h.words.size();
  return nfa->h.words.data();
}

extern "C" void rx_nfa_free(rx_nfa* nfa) {
  if (!nfa) return;
  int prev = -1;
  bool have_prev = hipGetDevice(&prev) == hipSuccess;
  for (auto& kv : nfa->dev) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    (void)hipFree(kv.second.words);
    (void)hipFree(kv.second.symidx);
    (void)hipFree(kv.second.ovf);
    (void)hipFree(kv.second.accept_bits);
    (void)hipFree(kv.second.symidx_c);
    (void)hipFree(kv.second.symidx_p);
    (void)hipFree(kv.second.ovf_dir);
    (void)hipFree(kv.second.pin_tab);
    (void)hipFree(kv.second.regidx);
    (void)hipFree(kv.second.byte_class);
    (void)hipFree(kv.second.dfa_trans);
    (void)hipFree(kv.second.dfa_pool);
    (void)hipFree(kv.second.dfa_hash);
    (void)hipFree(kv.second.dfa_hdr);
  }
  if (have_prev) (void)hipSetDevice(prev);
  delete nfa;
}

extern "C" int rx_trace_load_mem(const char* path, uint8_t** bytes, size_t* n) {
  RX_TRY
  if (!path || !bytes || !n) return RX_EINVAL;
  std::string txt;
  int rc = rxh_read_file(path, &txt);
  if (rc) return rc;
  std::vector<uint8_t> b;
  rc = rxh_parse_mem_text(txt.data(), txt.size(), &b);
  if (rc) return rc;
  uint8_t* o = (uint8_t*)malloc(b.size() ? b.size() : 1);
  if (!o) return RX_ENOMEM;
  memcpy(o, b.data(), b.size());
  *bytes = o;
  *n = b.size();
  return RX_OK;
  RX_CATCH
}
extern "C" void rx_free(void* p) { free(p); }

extern "C" int rx_device_count(int* n) {
  RX_TRY
  if (!n) return RX_EINVAL;
  *n = 0;
  HIPCHK(hipGetDeviceCount(n));
  return RX_OK;
  RX_CATCH
}
extern "C" int rx_device_name(int device, char* buf, size_t buflen) {
  RX_TRY
  if (!buf || buflen == 0) return RX_EINVAL;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return RX_OK;
  RX_CATCH
}

template <typename T>
static int upload_vec(const std::vector<T>& v, T** d) {
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIPCHK(hipMalloc((void**)d, bytes));
  if (!v.empty()) HIPCHK(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return RX_OK;
}

// HBM copy of the automaton on `device` (current device must already be `device`).
static int get_dev_tables(const rx_nfa* cnfa, int device, DevTables* out) {
  rx_nfa* nfa = const_cast<rx_nfa*>(cnfa);
  std::lock_guard<std::mutex> lk(nfa->mu);
  auto it = nfa->dev.find(device);
  if (it != nfa->dev.end()) { *out = it->second; return RX_OK; }
  DevTables t;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  t.cu_count = prop.multiProcessorCount;
  t.lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor
                                                       : prop.sharedMemPerBlock;
  int rc;
  if ((rc = upload_vec(nfa->h.words, &t.words))) return rc;  // the .coe words, unchanged
  if ((rc = upload_vec(nfa->h.symidx, &t.symidx))) return rc;
  if ((rc = upload_vec(nfa->h.ovf, &t.ovf))) return rc;
  if ((rc = upload_vec(nfa->h.accept_bits, &t.accept_bits))) return rc;
  if ((rc = upload_vec(nfa->h.symidx_c, &t.symidx_c))) return rc;
  if (!nfa->h.symidx_p.empty() && (rc = upload_vec(nfa->h.symidx_p, &t.symidx_p))) return rc;
  if (!nfa->h.ovf_dir.empty() && (rc = upload_vec(nfa->h.ovf_dir, &t.ovf_dir))) return rc;
  if (!nfa->h.pin_tab.empty() && (rc = upload_vec(nfa->h.pin_tab, &t.pin_tab))) return rc;
  if (!nfa->h.regidx.empty() && (rc = upload_vec(nfa->h.regidx, &t.regidx))) return rc;
  {
    std::vector<uint32_t> bc(64);
    memcpy(bc.data(), nfa->h.byte_class, 256);
    if ((rc = upload_vec(bc, &t.byte_class))) return rc;
  }
  nfa->dev[device] = t;
  *out = t;
  return RX_OK;
}

// ---- lazy-DFA cache -----------------------------------------------------------------------------
static int dfa_init_tables(const rx_nfa* nfa, DevTables& t) {
  const RxHostNfa& h = nfa->h;
  HIPCHK(hipMemset(t.dfa_trans, 0, (size_t)t.dfa_pool_chunks * h.n_classes * sizeof(uint32_t)));
  HIPCHK(hipMemset(t.dfa_pool, 0, (size_t)t.dfa_pool_chunks * 32 * sizeof(uint32_t)));
  HIPCHK(hipMemset(t.dfa_hash, 0, ((size_t)t.dfa_hash_mask + 1) * sizeof(uint32_t)));
  // chunk 0 is never used (0 = "unknown"); chunk 1 = the reset set {state 0} (Design/FPGA.v:134-147)
  const bool acc0 = (h.accept_bits[0] & 1u) != 0;
  uint32_t first[32] = {0};
  first[0] = 1;
  first[1] = h.row_ptr()[1] - h.row_ptr()[0];
  first[2] = acc0 ? 1u : 0u;
  first[DFA_HDR_WORDS] = 0u | (acc0 ? RXE_ACCEPT : 0u);
  HIPCHK(hipMemcpy(t.dfa_pool + 32, first, sizeof(first), hipMemcpyHostToDevice));
  const uint32_t hdr[4] = {0, 2, 1, 0};  // next free chunk = 2, one state so far
  HIPCHK(hipMemcpy(t.dfa_hdr, hdr, sizeof(hdr), hipMemcpyHostToDevice));
  return RX_OK;
}

static int ensure_dfa_tables(const rx_nfa* cnfa, int device, DevTables* out) {
  rx_nfa* nfa = const_cast<rx_nfa*>(cnfa);
  std::lock_guard<std::mutex> lk(nfa->mu);
  DevTables& t = nfa->dev[device];
  if (!t.dfa_trans) {
    t.dfa_pool_chunks = 1u << 18;  // 262 144 chunks of 128 B: up to that many DFA states (32 MB)
    t.dfa_hash_mask = (1u << 19) - 1;
    HIPCHK(hipMalloc((void**)&t.dfa_trans, (size_t)t.dfa_pool_chunks * nfa->h.n_classes * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void**)&t.dfa_pool, (size_t)t.dfa_pool_chunks * 32 * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void**)&t.dfa_hash, ((size_t)t.dfa_hash_mask + 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void**)&t.dfa_hdr, 4 * sizeof(uint32_t)));
    int rc = dfa_init_tables(nfa, t);
    if (rc) return rc;
  }
  *out = t;
  return RX_OK;
}

extern "C" int rx_nfa_dfa_info(const rx_nfa* cnfa, int device, uint64_t* n_states, uint64_t* n_transitions) {
  RX_TRY
  if (!cnfa) return RX_EINVAL;
  rx_nfa* nfa = const_cast<rx_nfa*>(cnfa);
  std::lock_guard<std::mutex> lk(nfa->mu);
  if (n_states) *n_states = 0;
  if (n_transitions) *n_transitions = 0;
  auto it = nfa->dev.find(device);
  if (it == nfa->dev.end() || !it->second.dfa_hdr) return RX_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  HIPCHK(hipSetDevice(device));
  uint32_t hdr[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpy(hdr, it->second.dfa_hdr, sizeof(hdr), hipMemcpyDeviceToHost));
  if (prev >= 0) (void)hipSetDevice(prev);
  if (n_states) *n_states = hdr[2];
  if (n_transitions) *n_transitions = hdr[3];
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_nfa_dfa_reset(const rx_nfa* cnfa, int device) {
  RX_TRY
  if (!cnfa) return RX_EINVAL;
  rx_nfa* nfa = const_cast<rx_nfa*>(cnfa);
  std::lock_guard<std::mutex> lk(nfa->mu);
  auto it = nfa->dev.find(device);
  if (it == nfa->dev.end() || !it->second.dfa_trans) return RX_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipDeviceSynchronize());
  int rc = dfa_init_tables(nfa, it->second);
  if (prev >= 0) (void)hipSetDevice(prev);
  return rc;
  RX_CATCH
}

// ---- plan -----------------------------------------------------------------------------------------
struct rx_plan {
  const rx_nfa* nfa = nullptr;
  rx_opts opts{};
  int device = 0;
  hipStream_t stream = nullptr;
  DevTables tab;
  size_t max_streams = 0, max_len = 0, events_cap = 0;
  bool want_mc = false, want_am = false, want_final = false;
  // device buffers
  uint8_t* d_in_own = nullptr;
  size_t d_in_own_bytes = 0;
  const uint8_t* d_in = nullptr;
  rx_event* d_events = nullptr;
  // two sets of {counters[16], match_count_total[size]} that alternate between launches: the kernel zeroes the set of
  // the NEXT launch (RxParams::zero_next), so no reset has to be enqueued between two launches
  unsigned long long* d_cset[2] = {nullptr, nullptr};
  int cur_set = 0;             // the set the LAST launch used (what download reads)
  bool sets_clean = false;     // both sets are zero except for what the last launch accumulated in d_cset[cur_set]
  unsigned long long* d_counters = nullptr;  // == d_cset[cur_set]
  uint32_t* d_mc = nullptr;
  unsigned long long* d_mct = nullptr;
  uint32_t* d_am = nullptr;
  uint32_t* d_final = nullptr;
  // rx_plan_run, compact final sets (on request): states per block of streams, offset / count per stream
  uint32_t* d_fstates = nullptr;
  // rx_plan_run: page-locked staging for the two downloads whose sizes are only known once the counters are on the host
  // (accept events, compact final lists); grown on demand
  void* h_stage_ev = nullptr;
  size_t h_stage_ev_bytes = 0;
  void* h_stage_fs = nullptr;
  size_t h_stage_fs_bytes = 0;
  uint32_t *d_foff = nullptr, *d_fcnt = nullptr;
  size_t fstates_cap = 0;
  uint32_t* d_init = nullptr;
  bool have_init = false;             // start sets belong to ONE batch: every new input clears the flag
  std::vector<uint64_t> init_stage;   // host staging of the caller's start sets (tail bits masked)
  uint32_t *d_spill_streams = nullptr, *d_spill_k = nullptr, *d_spill_rows = nullptr;
  size_t am_stride = 0;
  // current batch
  size_t n_streams = 0, stream_len = 0, stride = 0;
  bool have_input = false, launched = false;
  bool auto_decided = false;   // RX_KERNEL_AUTO: the probe's decision is valid for the current batch
  uint32_t batches_since_probe = 0;
  // AUTO's decisions by batch shape (ceil log2 of the stream count and of the stream length): a plan that is fed
  // alternating shapes probes each of them once, not on every change.  `pinned`: made by rx_plan_tune — never probed again.
  struct AutoChoice {
    uint32_t kernel = RX_KERNEL_SYM_PACK, lanes = 16;
    bool prune = false, fold = false, reg_skip = true, probe_prune = false, pinned = false;
    double probe_active = 0;
  };
  std::map<uint32_t, AutoChoice> choices;
  uint32_t shape_key = 0;
  bool choice_pinned = false;  // the current decision came from rx_plan_tune
  bool tuning = false;         // inside rx_plan_tune: probe even under RX_OPT_NO_PROBE
  uint32_t auto_kernel = RX_KERNEL_SYM_PACK;
  uint32_t auto_lanes = 16;    // streams per wavefront chosen for the pack kernel
  bool auto_prune = false;     // look-ahead pruning chosen (and verified at auto_lanes) by the probe
  bool auto_fold = false;      // always-on-state folding chosen (and verified at auto_lanes) by the probe
  bool auto_reg_skip = true;   // register kernel: the build that steps over idle passes (AUTO's trial / timed choice)
  bool probe_prune = false;    // the probe's statistics say pruning pays (used when the caller fixes the kernel)
  double probe_active = 0;     // active states per stream-byte seen by the probe
  RxParams params{};
  RxLaunchCfg cfg{};
  // rx_plan_run: blocks of streams in flight on their own HIP streams
  struct Pipe {
    unsigned long long* d_set = nullptr;   // {counters[16], match_count_total[size]} of the block (device)
    unsigned long long* h_set = nullptr;   // the same, page-locked host memory
    hipEvent_t up = nullptr;               // the block's input is in HBM
    hipEvent_t k0 = nullptr, k1 = nullptr; // bracket the block's kernels
  };
  std::vector<Pipe> pipes;
  hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;  // uploads / kernels / downloads of rx_plan_run
  // rx_plan_run: one capacity for the whole call.  d_run_ctr[0] = accept events of all blocks so far (their slot counter),
  // [1] = entries of the compact final sets so far, [2 + b] / [10 + b] = those two after block b (snapshots taken on the
  // kernel stream, so that the host can cut the shared buffers back into blocks); h_run_ctr: page-locked copy
  unsigned long long* d_run_ctr = nullptr;
  unsigned long long* h_run_ctr = nullptr;
  // one hipEvent pair per launch since the last rx_plan_kernel_times() call
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
  size_t n_timed = 0;
  double last_ms = 0;
};

// struct_size == 0 was documented by ABI 1 as "this version": its meaning is frozen at the ABI-1 layouts (rx_opts up to
// `flags`, rx_result up to the end of stats.tb_cycles), so that a caller built against ABI 1 that left the field at 0
// is neither read nor written past the end of its structs.  Callers that want the newer fields state their size.
static constexpr size_t RX_OPTS_ABI1_BYTES = offsetof(rx_opts, flags);
static constexpr size_t RX_RESULT_ABI1_BYTES = offsetof(rx_result, stats) + offsetof(rx_stats, lanes_used);

// rx_opts as the caller's version of the header laid it out: fields beyond its struct_size read as 0
static rx_opts read_opts(const rx_opts* opts) {
  rx_opts o{};
  o.device = -1;
  if (opts) {
    const size_t have = opts->struct_size ? std::min<size_t>(opts->struct_size, sizeof(rx_opts)) : RX_OPTS_ABI1_BYTES;
    memcpy(&o, opts, have);
  }
  return o;
}

static uint64_t passes_for(size_t n, uint32_t mode) {
  if (mode == RX_MODE_TB_COMPAT) return n >= 1 ? n - 1 : 0;  // testbench_BLK_Mem.sv:71: $finish at m == N
  return (uint64_t)n + 1;
}

static int bind_device(int device, int* resolved) {
  if (device < 0) {
    HIPCHK(hipGetDevice(resolved));
  } else {
    HIPCHK(hipSetDevice(device));
    *resolved = device;
  }
  return RX_OK;
}

extern "C" int rx_plan_create(const rx_nfa* nfa, const rx_opts* opts, size_t max_streams, size_t max_stream_len,
                              size_t events_cap, uint32_t want_match_count, uint32_t want_anymatch,
                              uint32_t want_final, rx_plan** out) {
  RX_TRY
  if (!nfa || !out || max_streams == 0) return RX_EINVAL;
  if (max_streams > 0xFFFFFFFFull || max_stream_len > 0xFFFFFFF0ull || events_cap > 0xFFFFFFFFull)
    return RX_ECAPACITY;
  const rx_opts o = read_opts(opts);
  if (o.mode > RX_MODE_TB_COMPAT) return RX_EINVAL;
  // rx_event.k and the kernels' pass counters are 32 bits wide: a chained stream may not run past 2^32 passes
  if (o.k_base + passes_for(max_stream_len, RX_MODE_FULL) > (1ull << 32)) return RX_EINVAL;
  int ndev = 0;
  {
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
    if (ndev <= 0) return RX_ENODEVICE;
  }
  rx_plan* p = new (std::nothrow) rx_plan();
  if (!p) return RX_ENOMEM;
  p->nfa = nfa;
  p->opts = o;
  p->stream = (hipStream_t)o.stream;
  p->max_streams = max_streams;
  p->max_len = max_stream_len;
  p->events_cap = events_cap;
  p->want_mc = want_match_count != 0;
  p->want_am = want_anymatch != 0;
  p->want_final = want_final != 0;
  int rc = bind_device(o.device, &p->device);
  if (rc) { delete p; return rc; }
  rc = get_dev_tables(nfa, p->device, &p->tab);
  if (rc) { delete p; return rc; }
  const uint32_t size = nfa->h.size;
  const size_t nw64x2 = 2 * (((size_t)size + 63) / 64);
  auto fail = [&](int code) { rx_plan_free(p); return code; };
#define PLCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(hip_fail(e_, #call)); } while (0)
  for (int q = 0; q < 2; q++) {
    PLCHK(hipMalloc((void**)&p->d_cset[q], (16 + (size_t)size) * sizeof(unsigned long long)));
    PLCHK(hipMemset(p->d_cset[q], 0, (16 + (size_t)size) * sizeof(unsigned long long)));
  }
  p->d_counters = p->d_cset[0];
  p->d_mct = p->d_cset[0] + 16;
  p->sets_clean = true;
  PLCHK(hipMalloc((void**)&p->d_events, std::max<size_t>(events_cap, 1) * sizeof(rx_event)));
  if (p->want_mc) PLCHK(hipMalloc((void**)&p->d_mc, max_streams * size * sizeof(uint32_t)));
  // rows of the plan's any-match bitmap: padded to a multiple of eight words, so that the pack kernel's 256-pass groups are
  // aligned 32-byte sectors (a caller whose anymatch_stride is the same gets flat copies, any other stride row-by-row ones)
  p->am_stride = ((size_t)((passes_for(max_stream_len, RX_MODE_FULL) + 31) / 32) + 7) & ~(size_t)7;
  if (p->want_am) PLCHK(hipMalloc((void**)&p->d_am, max_streams * p->am_stride * sizeof(uint32_t)));
  if (p->want_final) PLCHK(hipMalloc((void**)&p->d_final, max_streams * nw64x2 * sizeof(uint32_t)));
#undef PLCHK
  *out = p;
  return RX_OK;
  RX_CATCH
}

extern "C" void rx_plan_free(rx_plan* p) {
  if (!p) return;
  int prev = -1;
  bool have_prev = hipGetDevice(&prev) == hipSuccess;
  (void)hipSetDevice(p->device);
  (void)hipFree(p->d_in_own);
  (void)hipFree(p->d_events);
  (void)hipFree(p->d_cset[0]);
  (void)hipFree(p->d_cset[1]);
  (void)hipFree(p->d_mc);
  (void)hipFree(p->d_am);
  (void)hipFree(p->d_final);
  (void)hipFree(p->d_fstates);
  if (p->h_stage_ev) (void)hipHostFree(p->h_stage_ev);
  if (p->h_stage_fs) (void)hipHostFree(p->h_stage_fs);
  (void)hipFree(p->d_foff);
  (void)hipFree(p->d_fcnt);
  (void)hipFree(p->d_init);
  (void)hipFree(p->d_spill_streams);
  (void)hipFree(p->d_spill_k);
  (void)hipFree(p->d_spill_rows);
  (void)hipFree(p->d_run_ctr);
  if (p->h_run_ctr) (void)hipHostFree(p->h_run_ctr);
  for (auto& e : p->evs) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& q : p->pipes) {
    (void)hipFree(q.d_set);
    if (q.h_set) (void)hipHostFree(q.h_set);
    if (q.up) (void)hipEventDestroy(q.up);
    if (q.k0) (void)hipEventDestroy(q.k0);
    if (q.k1) (void)hipEventDestroy(q.k1);
  }
  if (p->s_in) (void)hipStreamDestroy(p->s_in);
  if (p->s_k) (void)hipStreamDestroy(p->s_k);
  if (p->s_out) (void)hipStreamDestroy(p->s_out);
  if (have_prev) (void)hipSetDevice(prev);
  delete p;
}

static uint32_t ceil_log2(size_t v) {
  uint32_t b = 0;
  while (b < 63 && ((size_t)1 << b) < v) b++;
  return b;
}

static void store_choice(rx_plan* p, bool pinned) {
  rx_plan::AutoChoice c;
  c.kernel = p->auto_kernel;
  c.lanes = p->auto_lanes;
  c.prune = p->auto_prune;
  c.fold = p->auto_fold;
  c.reg_skip = p->auto_reg_skip;
  c.probe_prune = p->probe_prune;
  c.probe_active = p->probe_active;
  c.pinned = pinned;
  p->choices[p->shape_key] = c;
  p->choice_pinned = pinned;
}

static int set_batch(rx_plan* p, size_t n_streams, size_t stream_len, size_t stride) {
  if (n_streams == 0 || n_streams > p->max_streams || stream_len > p->max_len || stride < stream_len)
    return RX_EINVAL;
  // AUTO's probe costs about as much as a launch.  Its decision is kept per shape bucket; a plan that is fed batch after
  // batch of one shape (serving) looks again every 32nd batch — unless the decision was made by rx_plan_tune or the plan
  // was created with RX_OPT_NO_PROBE.  A wrong guess only costs speed (hand-offs keep every kernel exact).
  const bool same_shape = p->have_input && p->n_streams == n_streams && p->stream_len == stream_len;
  if (!same_shape) {
    p->shape_key = (ceil_log2(n_streams) << 8) | ceil_log2(stream_len + 1);
    auto it = p->choices.find(p->shape_key);
    p->batches_since_probe = 0;
    if (it != p->choices.end()) {
      const rx_plan::AutoChoice& c = it->second;
      p->auto_kernel = c.kernel;
      p->auto_lanes = c.lanes;
      p->auto_prune = c.prune;
      p->auto_fold = c.fold;
      p->auto_reg_skip = c.reg_skip;
      p->probe_prune = c.probe_prune;
      p->probe_active = c.probe_active;
      p->choice_pinned = c.pinned;
      p->auto_decided = true;
    } else {
      p->auto_decided = false;
      p->choice_pinned = false;
    }
  } else if (!p->choice_pinned && !(p->opts.flags & RX_OPT_NO_PROBE) && ++p->batches_since_probe >= 32) {
    p->auto_decided = false;
    p->batches_since_probe = 0;
  }
  p->n_streams = n_streams;
  p->stream_len = stream_len;
  p->stride = stride;
  p->have_input = true;
  p->launched = false;
  p->have_init = false;  // a start set describes the batch it was given for (rx_plan_set_init_active comes AFTER the input)
  return RX_OK;
}

extern "C" int rx_plan_upload(rx_plan* p, const uint8_t* bytes, size_t n_streams, size_t stream_len,
                              size_t stride) {
  RX_TRY
  if (!p || (!bytes && stream_len)) return RX_EINVAL;
  int dev;
  int rc = bind_device(p->device, &dev);
  if (rc) return rc;
  rc = set_batch(p, n_streams, stream_len, stride);
  if (rc) return rc;
  // rows are packed to a 4-byte-aligned pitch in HBM so every lane can take a whole dword
  const size_t pitch = (stream_len + 3) & ~(size_t)3;
  const size_t need = std::max<size_t>(n_streams * pitch, 4);
  if (need > p->d_in_own_bytes) {
    (void)hipFree(p->d_in_own);
    p->d_in_own = nullptr;
    p->d_in_own_bytes = 0;
    HIPCHK(hipMalloc((void**)&p->d_in_own, need));
    p->d_in_own_bytes = need;
  }
  if (stream_len) {
    if (stride == pitch && stream_len == pitch)  // rows packed without padding: one flat copy (the 2-D path is slower from pageable memory)
      HIPCHK(hipMemcpyAsync(p->d_in_own, bytes, n_streams * pitch, hipMemcpyHostToDevice, p->stream));
    else
      HIPCHK(hipMemcpy2DAsync(p->d_in_own, pitch, bytes, stride, stream_len, n_streams, hipMemcpyHostToDevice,
                              p->stream));
  }
  p->d_in = p->d_in_own;
  p->stride = pitch;
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_plan_set_device_input(rx_plan* p, const void* device_bytes, size_t n_streams,
                                        size_t stream_len, size_t stride) {
  RX_TRY
  if (!p || (!device_bytes && stream_len)) return RX_EINVAL;
  int rc = set_batch(p, n_streams, stream_len, stride);
  if (rc) return rc;
  p->d_in = (const uint8_t*)device_bytes;
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_plan_set_init_active(rx_plan* p, const uint64_t* init_active) {
  RX_TRY
  if (!p) return RX_EINVAL;
  if (!init_active) { p->have_init = false; return RX_OK; }
  if (!p->have_input) return RX_ESTATE;
  int dev;
  int rc = bind_device(p->device, &dev);
  if (rc) return rc;
  const uint32_t size = p->nfa->h.size;
  const size_t nw64 = ((size_t)size + 63) / 64;
  if (!p->d_init) {
    HIPCHK(hipMalloc((void**)&p->d_init, p->max_streams * nw64 * sizeof(uint64_t)));
    HIPCHK(hipMemsetAsync(p->d_init, 0, p->max_streams * nw64 * sizeof(uint64_t), p->stream));
  }
  // Bits at or above `size` in a row's last word name states that do not exist (the kernels would index the
  // tables with them): they are cleared in a staging copy, which also makes the call safe to return from — the
  // caller's array is not read after this function returns.
  try {
    p->init_stage.assign(init_active, init_active + p->n_streams * nw64);
  } catch (...) {
    return RX_ENOMEM;
  }
  if (size & 63u) {
    const uint64_t keep = (1ull << (size & 63u)) - 1ull;
    for (size_t s = 0; s < p->n_streams; s++) p->init_stage[s * nw64 + nw64 - 1] &= keep;
  }
  HIPCHK(hipMemcpyAsync(p->d_init, p->init_stage.data(), p->n_streams * nw64 * sizeof(uint64_t), hipMemcpyHostToDevice,
                        p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));  // pageable staging memory: the copy has left it when this returns
  p->have_init = true;
  return RX_OK;
  RX_CATCH
}

static void fill_common(rx_plan* p, RxParams& a) {
  const RxHostNfa& h = p->nfa->h;
  a = RxParams{};
  a.words = p->tab.words;
  a.symidx = p->tab.symidx;
  a.ovf = p->tab.ovf;
  a.accept_bits = p->tab.accept_bits;
  a.symidx_c = p->tab.symidx_c;
  a.symidx_p = p->tab.symidx_p;
  a.prune_narrow = h.prune_narrow ? 1u : 0u;
  a.ovf_dir = p->tab.ovf_dir;
  a.byte_class = p->tab.byte_class;
  a.pin_tab = p->tab.pin_tab;
  a.regidx = p->tab.regidx;
  a.reg_tmask = h.reg_tmask;
  a.pin_cols = h.n_classes + 1u;
  a.n_classes = h.n_classes;
  a.size = h.size;
  a.bytes = p->d_in;
  a.stride = p->stride;
  a.state0_entry = (h.accept_bits[0] & 1u) ? RXE_ACCEPT : 0u;
  a.nw64x2 = 2u * ((h.size + 63u) / 64u);
  a.counters = p->d_counters;
  a.ev_count = p->d_counters;
  a.pin_state = h.pin_state;
  a.pin_degree = h.pin_state != 0xFFFFFFFFu ? h.row_ptr()[h.pin_state + 1] - h.row_ptr()[h.pin_state] : 0;
}

static int ensure_spill_area(rx_plan* p, RxParams& a) {
  if (!p->d_spill_rows) {  // hand-off area group/pack kernel -> wave kernel, sized so that it cannot overflow
    HIPCHK(hipMalloc((void**)&p->d_spill_streams, p->max_streams * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void**)&p->d_spill_k, p->max_streams * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void**)&p->d_spill_rows, p->max_streams * (size_t)a.nw64x2 * sizeof(uint32_t)));
  }
  a.spill_count = p->d_counters + 3;
  a.spill_streams = p->d_spill_streams;
  a.spill_k = p->d_spill_k;
  a.spill_rows = p->d_spill_rows;
  return RX_OK;
}

// RX_KERNEL_AUTO: the fastest kernel depends on how many states are active per stream, which depends on
// the input.  Probe: the pack kernel's statistics build over a corner of the batch (512 K stream-bytes: the first <= 4096
// bytes of <= 512 streams, no outputs), then: small active sets -> pack kernel, larger ones -> wavefront-per-stream slice kernel.
static int auto_probe_pack(rx_plan* p) {
  p->auto_kernel = RX_KERNEL_SYM_PACK;
  p->auto_lanes = 16;
  p->auto_prune = false;
  p->auto_fold = false;
  p->probe_prune = false;
  p->probe_active = 0;
  if (p->n_streams * p->stream_len < (256u << 10)) return RX_OK;  // tiny batch: not worth a probe
  // one run of the pack kernel with `lanes` streams per wavefront over the corner of the batch: the statistics
  // build (counters) or, with stats = false, the build that would really run (only the hand-off count is read)
  // the sample: up to 4 KB of each stream (how many states are active grows along a stream: the first KB of 4 KB windows
  // shows 2.5 list entries per stream-byte, the whole window 3.8), as many streams as make 512 K stream-bytes
  const size_t sample_len = std::min<size_t>(p->stream_len, 4096);
  const size_t sample_streams = std::min<size_t>(p->n_streams, std::max<size_t>(128, (512u << 10) / std::max<size_t>(sample_len, 1)));
  unsigned long long cnt[16];
  bool run_fold = false;  // the next run() uses the FOLD build
  auto run = [&](uint32_t lanes, bool stats, bool prune, double* spilled) -> int {
    RxParams a;
    fill_common(p, a);
    a.n_streams = (uint32_t)sample_streams;
    a.stream_len = (uint32_t)sample_len;
    a.n_passes = a.stream_len + 1;
    a.n_consume = a.stream_len;
    RxLaunchCfg cfg{};
    cfg.group_lanes = lanes;
    int rc = rx_pick_launch(RX_KERNEL_SYM_PACK, a.size, a.n_streams, p->tab.cu_count, p->tab.lds_per_cu, &a, &cfg);
    if (rc) return rc;
    cfg.stats = stats;
    cfg.prune = prune;
    cfg.fold = run_fold && !stats;
    if ((rc = ensure_spill_area(p, a))) return rc;
    HIPCHK(hipMemsetAsync(p->d_counters, 0, 16 * sizeof(unsigned long long), p->stream));
    p->sets_clean = false;  // (the probe accumulates into the current set; both are reset before the real launch)
    hipError_t e = (hipError_t)rx_launch(a, cfg, p->stream);
    if (e != hipSuccess) return hip_fail(e, "probe launch");
    HIPCHK(hipMemcpyAsync(cnt, p->d_counters, sizeof(cnt), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    *spilled = (double)cnt[3] / a.n_streams;
    return RX_OK;
  };
  const double units = (double)sample_streams * (double)std::max<size_t>(sample_len, 1);
  double spilled = 0;
  int rc = run(16, true, false, &spilled);
  if (rc) return rc;
  // (When only a few streams left the sample run, the pack kernel's own share of the active states is the better measure of
  // how full its lists are: what a stream that was handed off does afterwards, on the wave kernel, says nothing about them —
  // since round 3 only the stream that overflows leaves.  When most of the sample left, the own share is what was counted
  // BEFORE they left and says nothing either: then the total stands.)
  const double spilled16 = spilled;
  const double active = (double)((spilled16 <= 0.05 && cnt[7]) ? cnt[7] : cnt[1]) / units;
  p->probe_active = active;
  // the pack kernel is fastest when one pass of a wavefront is ONE sweep with 30-37 of the 64 lanes busy:
  // streams per wavefront ~ 33 / (list entries per stream)   (snort_16: T 2.3 -> 13, U 1.15 -> 32)
  // A batch that cannot give every SIMD at least two wavefronts at that size is latency-bound (a wavefront alone on
  // its SIMD finishes a pass in ~1 100 cycles whatever it holds): then fewer streams per wavefront win
  // (4 096 streams: S=4 0.52 ms, S=13 0.60 ms; 16 384 streams: S=8 0.61 ms, S=16 0.69 ms).
  const double per_simd = (double)p->n_streams / (4.0 * std::max(p->tab.cu_count, 1));
  auto lanes_for = [per_simd](double entries) {
    // 13 and 22 = ceil(64 / 5) and ceil(64 / 3): at 65 536 streams on 1 024 SIMDs they fill every SIMD with the same
    // number of wavefronts, like 16 (4) and 32 (2); measured optimum 13-16 for 2.3 entries per stream
    static const uint32_t choices[] = {4, 8, 11, 13, 16, 22, 24, 32};
    const double want = std::min(33.0 / std::max(entries, 0.5), std::max(per_simd / 2.0, 4.0));
    uint32_t best = 16;
    double bd = 1e9;
    for (uint32_t c : choices) {
      const double d = std::abs((double)c - want);
      if (d < bd) { bd = d; best = c; }
    }
    return best;
  };
  // multi-target rows met by at least 2 % of the list entries: look-ahead pruning pays (rule sets, l7-filter); the
  // entries it keeps out of the lists are the ones the statistics build saw die at once
  const double own = (double)std::max<unsigned long long>(cnt[7], 1);
  const bool dbg = (p->opts.flags & RX_OPT_VERBOSE) != 0;
  if (dbg)
    fprintf(stderr, "[rxmatch] probe: %.2f active states per stream-byte, hand-offs %.1f %%, %.1f %% of the entries on "
                    "multi-target rows, %.1f %% of the entries dead on arrival from such rows\n",
            active, 100.0 * spilled16, 100.0 * (double)cnt[5] / own, 100.0 * (double)cnt[6] / own);
  // (pruning must remove at least a tenth of the entries to pay for its directory look-ups: l7-filter meets
  // multi-target rows in every pass but nearly all of their targets live on)
  p->probe_prune = p->tab.ovf_dir && (double)cnt[5] / own >= 0.02 && (double)cnt[6] / own >= 0.10;
  // (With the narrow pruned index the PRUNE build also drops INLINE targets that die on the next byte — -21 % list entries
  // on the snort_16 trace windows.  Measured this round it does not pay on its own: the PRUNE pass reads a second class
  // byte and carries the look-ahead layout, and costs more than the entries it saves at every batch size — 65 536
  // streams 1.09 vs 1.05 ms, 262 144 streams 4.09 vs 3.90 ms.  So the trigger stays the multi-target statistic above;
  // where PRUNE runs for that reason, the inline targets are pruned along.)
  const double dead_frac = (double)cnt[6] / own;  // (the runs below overwrite cnt[])
  const double left_pruned = active;
  // Always-on-state folding (automata whose state 0 enters a `.*` state on every byte): that state leaves the lists, and
  // of its targets only those that survive the next byte enter them.  The FOLD build pays a fixed price per pass for
  // the folded state's table look-ups and wins when the lists are nearly empty afterwards (measured, snort_16, one
  // MI355X: uniform bytes 0.006 entries left per stream-byte: 65 536 streams 1 050 -> 2 600 Gbit/s, 131 072 streams
  // 1 120 -> 2 770 Gbit/s; trace windows 1.1 left: 505 -> 476 Gbit/s, no gain) — so the probe runs it on the sample,
  // reads how many entries were left, and takes it below 0.3 per stream-byte.  Streams per wavefront: as many as still
  // give every SIMD two wavefronts (16 ... 64).
  if (p->tab.pin_tab && !(p->opts.flags & RX_OPT_NO_FOLD) && active <= 3.0) {
    const bool prune = p->probe_prune && !(p->opts.flags & RX_OPT_NO_PRUNE);
    run_fold = true;
    rc = run(32, false, prune, &spilled);
    run_fold = false;
    if (rc) return rc;
    const double left = (double)cnt[7] / units;
    static const uint32_t fold_s[] = {16, 24, 32, 48, 64};
    uint32_t lanes = 16;
    for (uint32_t c : fold_s) if ((double)c <= per_simd / 2.0) lanes = c;
    // Lists that are empty nearly all the time: the FOLD build skips the passes in which nothing happens to ANY of a
    // wavefront's streams, so fewer streams per wavefront mean more passes skipped, and 16 streams are exactly one
    // wave-load of the window refill (uniform bytes, ms: 65 536 streams S=16 0.206 / S=32 0.274 / S=64 0.426; 131 072:
    // 0.387 / 0.537 / 0.477; 262 144: 0.714 / 0.849 / 0.924; 32 768: S=8 0.147, S=16 0.160)
    if (left <= 0.03) lanes = per_simd >= 48.0 ? 16u : 8u;
    if (dbg) fprintf(stderr, "[rxmatch] probe: folded build leaves %.3f list entries per stream-byte, hand-offs %.1f %% -> %s\n", left,
                     100.0 * spilled, (left <= 0.3 && spilled <= 0.02) ? "fold" : "no fold");
    if (left <= 0.3 && spilled <= 0.02) { p->auto_lanes = lanes; p->auto_fold = true; p->auto_prune = prune; return RX_OK; }
  }
  if (p->probe_prune && !(p->opts.flags & RX_OPT_NO_PRUNE)) {
    const double entries = std::min(active * (1.0 - dead_frac), left_pruned);
    if (entries <= 6.0) {
      const uint32_t lanes = lanes_for(entries);
      if ((rc = run(lanes, false, true, &spilled))) return rc;
      if (spilled <= 0.02) { p->auto_lanes = lanes; p->auto_prune = true; return RX_OK; }
    }
    if ((rc = run(4, false, true, &spilled))) return rc;
    if (spilled <= 0.02) { p->auto_lanes = 4; p->auto_prune = true; return RX_OK; }
  }
  if (active <= 6.0 && spilled16 <= 0.02) {
    p->auto_lanes = lanes_for(active);
    return RX_OK;
  }
  // many active states per stream: four streams per wavefront with the long list (512 entries) and the wider
  // filters, if that form keeps (nearly) all of the sample; otherwise one wavefront per stream
  if ((rc = run(4, true, false, &spilled))) return rc;
  if (spilled <= 0.02) p->auto_lanes = 4;
  else p->auto_kernel = RX_KERNEL_SYM_WAVE;
  return RX_OK;
}

// Small batches: with few wavefronts per SIMD the pack kernel is bound by the latency of its pass (a batch of 64 streams
// takes as long as one of 4 096), and one wavefront per stream on the register kernel is usually the shorter chain
// (snort_16, 1 KB streams: 0.29 against 0.61 ms at 16 streams, 0.35 / 0.52 at 1 024, 0.45 / 0.53 at 4 096; uniform bytes,
// where it skips the passes in which nothing is active: 0.02 / 0.42 at 16, 0.04 / 0.28 at 4 096; l7: 0.23 / 0.48) — unless
// the automaton places many targets per pass (the rule-set stand-in: 1.45 against 0.79 ms).  So for batches of up to 16
// wavefronts per SIMD AUTO asks the hardware: both candidates run the batch once, timed, without outputs.  Batches too
// small for a probe go to the register kernel when the folded state's emissions are single targets (no folding table:
// when few cells hold lists).
static int auto_probe(rx_plan* p, bool reg_eligible) {
  int rc = auto_probe_pack(p);
  if (rc || !reg_eligible) return rc;
  const RxHostNfa& h = p->nfa->h;
  if (p->n_streams * p->stream_len < (256u << 10)) {
    rx_nfa* n = const_cast<rx_nfa*>(p->nfa);
    {
      std::lock_guard<std::mutex> lk(n->mu);
      if (n->few_lists < 0) {
        size_t nz = 0, ov = 0;
        for (uint32_t w : (!h.pin_tab.empty() ? h.pin_tab : h.symidx_c)) { nz += w != 0u; ov += (w & RXE_OVF) != 0u; }
        n->few_lists = ov * 4u <= nz ? 1 : 0;
      }
    }
    if (n->few_lists == 1) p->auto_kernel = RX_KERNEL_SYM_REG;
    p->auto_reg_skip = !h.pin_tab.empty();
    return RX_OK;
  }
  if (p->auto_kernel != RX_KERNEL_SYM_PACK) return RX_OK;  // (many active states per stream: neither of the two)
  struct EventPair {  // (destroyed on every path out of the probe)
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } ev;
  HIPCHK(hipEventCreate(&ev.a));
  HIPCHK(hipEventCreate(&ev.b));
  const hipEvent_t e0 = ev.a, e1 = ev.b;
  // one warm-up launch on the probe's corner of the batch (the first launch of a kernel pays for its code object), then
  // the WHOLE batch, timed, without outputs: between 4 and 16 wavefronts per SIMD neither kernel's time can be read off
  // a smaller sample (the register kernel grows with the batch, the pack kernel does not), and the batch is small
  auto timed = [&](uint32_t kernel, bool reg_skip, float* ms) -> int {
    for (int it = 0; it < 2; it++) {
      RxParams a;
      fill_common(p, a);
      a.n_streams = it == 0 ? (uint32_t)std::min<size_t>(p->n_streams, 512) : (uint32_t)p->n_streams;
      a.stream_len = it == 0 ? (uint32_t)std::min<size_t>(p->stream_len, 1024) : (uint32_t)p->stream_len;
      a.n_passes = a.stream_len + 1;
      a.n_consume = a.stream_len;
      RxLaunchCfg cfg{};
      cfg.group_lanes = p->auto_lanes;
      int r = rx_pick_launch(kernel, a.size, a.n_streams, p->tab.cu_count, p->tab.lds_per_cu, &a, &cfg);
      if (r) return r;
      cfg.prune = kernel == RX_KERNEL_SYM_PACK && p->auto_prune;
      cfg.fold = kernel == RX_KERNEL_SYM_REG ? p->tab.pin_tab != nullptr : p->auto_fold;
      cfg.reg_skip = reg_skip;
      if ((r = ensure_spill_area(p, a))) return r;
      p->sets_clean = false;
      HIPCHK(hipMemsetAsync(p->d_counters, 0, 16 * sizeof(unsigned long long), p->stream));
      HIPCHK(hipEventRecord(e0, p->stream));
      hipError_t e = (hipError_t)rx_launch(a, cfg, p->stream);
      if (e != hipSuccess) return hip_fail(e, "probe launch");
      HIPCHK(hipEventRecord(e1, p->stream));
    }
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipEventElapsedTime(ms, e0, e1));
    return RX_OK;
  };
  float t_pack = 0.f, t_reg = 0.f, t_skip = 0.f;
  rc = timed(RX_KERNEL_SYM_PACK, false, &t_pack);
  if (!rc) rc = timed(RX_KERNEL_SYM_REG, false, &t_reg);
  if (!rc) rc = timed(RX_KERNEL_SYM_REG, true, &t_skip);
  if (rc) return rc;
  if (p->opts.flags & RX_OPT_VERBOSE)
    fprintf(stderr, "[rxmatch] probe: the batch on the pack kernel %.3f ms, one wavefront per stream %.3f ms, stepping over idle "
                    "passes %.3f ms\n", t_pack, t_reg, t_skip);
  p->auto_reg_skip = t_skip < t_reg;
  if (std::min(t_reg, t_skip) < 0.95f * t_pack) p->auto_kernel = RX_KERNEL_SYM_REG;
  return RX_OK;
}

// Everything a launch decides before anything is enqueued for it: kernel arguments for the whole batch (p->params),
// kernel choice (AUTO's probe runs here when its decision is not valid for the batch) and launch geometry (p->cfg).
static int prepare_launch(rx_plan* p) {
  int rc;
  const RxHostNfa& h = p->nfa->h;
  RxParams& a = p->params;
  fill_common(p, a);
  a.n_streams = (uint32_t)p->n_streams;
  a.stream_len = (uint32_t)p->stream_len;
  a.n_passes = (uint32_t)passes_for(p->stream_len, p->opts.mode);
  a.n_consume = p->opts.mode == RX_MODE_TB_COMPAT ? a.n_passes : (uint32_t)p->stream_len;
  if (p->opts.k_base + a.n_passes > (1ull << 32)) return RX_EINVAL;  // rx_event.k would wrap
  a.k_base = (uint32_t)p->opts.k_base;
  a.init_active = p->have_init ? p->d_init : nullptr;
  a.events = p->events_cap ? p->d_events : nullptr;
  a.events_cap = (uint32_t)p->events_cap;
  a.match_count = p->want_mc ? p->d_mc : nullptr;
  a.match_count_total = p->d_mct;
  a.anymatch = p->want_am ? p->d_am : nullptr;
  a.anymatch_stride = (uint32_t)p->am_stride;
  a.final_active = p->want_final ? p->d_final : nullptr;
  uint32_t kernel = p->opts.kernel;
  uint32_t auto_lanes = 0;
  const bool pair = p->opts.collect_stats == 2;
  if (pair) {  // the testbench's clock count needs both streams of a pair in one wavefront: pack kernel only
    if ((kernel != RX_KERNEL_AUTO && kernel != RX_KERNEL_SYM_PACK) || (p->n_streams & 1) || p->have_init) return RX_EINVAL;
    kernel = RX_KERNEL_SYM_PACK;
  }
  a.pair_cycles = pair ? 1u : 0u;
  // few long streams from reset (the reference's own run is one lock-step pair): latency per pass is what counts, and the
  // register-resident kernel has the shortest pass; it has no statistics build
  const bool reg_ok = p->opts.collect_stats == 0 && !p->have_init && p->tab.regidx;
  // RX_OPT_NO_PROBE: nothing below may run a kernel or wait for the stream; a shape rx_plan_tune has not seen gets the
  // defaults (the pack kernel at 16 streams per wavefront; up to 4 streams the register kernel's skipping build)
  const bool may_probe = p->tuning || !(p->opts.flags & RX_OPT_NO_PROBE);
  if (kernel == RX_KERNEL_AUTO && p->n_streams <= 4 && reg_ok) {
    kernel = RX_KERNEL_SYM_REG;
    if (!p->auto_decided && !may_probe) {
      p->auto_reg_skip = true;
      p->auto_decided = true;
    }
    if (!p->auto_decided) {
      // Which build: a trial run over the first 8 192 bytes with the one that steps over idle passes, which counts the
      // groups of passes it skipped in the second half (the busier shipped trace: none after pass 680 — the `.*` states
      // inside its patterns never leave once entered; the quieter one: 63 %).
      p->auto_reg_skip = true;
      if (p->stream_len >= 16384) {
        RxParams t;
        fill_common(p, t);
        t.n_streams = (uint32_t)p->n_streams;
        t.stream_len = 8192u;
        t.n_passes = t.n_consume = 8192u;
        RxLaunchCfg cfg{};
        if ((rc = rx_pick_launch(RX_KERNEL_SYM_REG, t.size, t.n_streams, p->tab.cu_count, p->tab.lds_per_cu, &t, &cfg))) return rc;
        cfg.fold = p->tab.pin_tab != nullptr;
        cfg.reg_skip = true;
        if ((rc = ensure_spill_area(p, t))) return rc;
        p->sets_clean = false;
        unsigned long long cnt[16];
        HIPCHK(hipMemsetAsync(p->d_counters, 0, sizeof(cnt), p->stream));
        hipError_t e = (hipError_t)rx_launch(t, cfg, p->stream);
        if (e != hipSuccess) return hip_fail(e, "trial launch");
        HIPCHK(hipMemcpyAsync(cnt, p->d_counters, sizeof(cnt), hipMemcpyDeviceToHost, p->stream));
        HIPCHK(hipStreamSynchronize(p->stream));
        p->auto_reg_skip = cnt[10] * 10u >= (unsigned long long)t.n_streams * 1024u;  // >= 10 % of the 1 024 groups per stream
        if (p->opts.flags & RX_OPT_VERBOSE)
          fprintf(stderr, "[rxmatch] register kernel trial: %llu of %u groups of passes idle -> %s\n", cnt[10], t.n_streams * 1024u,
                  p->auto_reg_skip ? "step over them" : "plain build");
      }
      p->auto_decided = true;
      store_choice(p, p->tuning);
    }
  }
  // (more streams, but at most 16 wavefronts of them per SIMD: the probe times both kernels on the batch)
  const bool reg_eligible = kernel == RX_KERNEL_AUTO && reg_ok && p->n_streams <= 64u * (size_t)std::max(p->tab.cu_count, 1);
  // the probe also serves an explicit RX_KERNEL_SYM_PACK: whether look-ahead pruning pays depends on the input
  const bool probe_for_pack = kernel == RX_KERNEL_SYM_PACK && p->tab.symidx_p && p->opts.collect_stats == 0;
  if ((kernel == RX_KERNEL_AUTO || probe_for_pack) && !pair && !p->have_init) {
    if (!p->auto_decided && !may_probe) {
      p->auto_kernel = RX_KERNEL_SYM_PACK;
      p->auto_lanes = 16;
      p->auto_prune = p->auto_fold = p->probe_prune = false;
      p->auto_reg_skip = p->tab.pin_tab != nullptr;
      p->auto_decided = true;
    }
    if (!p->auto_decided) {
      if ((rc = auto_probe(p, reg_eligible))) return rc;
      p->auto_decided = true;
      store_choice(p, p->tuning);
      if (p->opts.flags & RX_OPT_VERBOSE)
        fprintf(stderr, "[rxmatch] AUTO -> kernel %u, %u streams per wavefront, look-ahead pruning %s, folding %s\n",
                p->auto_kernel, p->auto_lanes, p->auto_prune ? "on" : "off", p->auto_fold ? "on" : "off");
    }
    if (kernel == RX_KERNEL_AUTO) {
      kernel = p->auto_kernel;
      if (p->opts.group_lanes == 0 && kernel == RX_KERNEL_SYM_PACK) auto_lanes = p->auto_lanes;
    }
  }
  // a caller-supplied start set is a bitmask row: that is the wave kernel's dense form
  if (p->have_init && (kernel == RX_KERNEL_AUTO || kernel == RX_KERNEL_SYM_GROUP || kernel == RX_KERNEL_SYM_PACK ||
                       kernel == RX_KERNEL_DFA || kernel == RX_KERNEL_SYM_REG))
    kernel = RX_KERNEL_SYM_WAVE;
  if (kernel == RX_KERNEL_SYM_REG && (p->opts.collect_stats != 0 || !p->tab.regidx)) kernel = RX_KERNEL_SYM_WAVE;
  p->cfg.group_lanes = auto_lanes ? auto_lanes : p->opts.group_lanes;
  rc = rx_pick_launch(kernel, h.size, a.n_streams, p->tab.cu_count, p->tab.lds_per_cu, &a, &p->cfg);
  if (rc) return rc;
  p->cfg.stats = p->opts.collect_stats != 0;
  // always-on-state folding: AUTO's verified choice; an explicit RX_KERNEL_SYM_PACK folds only on RX_OPT_FORCE_FOLD (its
  // group_lanes then names one of the FOLD builds: 8/13/16/24/32/48/64 streams per wavefront)
  p->cfg.fold = p->tab.pin_tab != nullptr && !p->cfg.stats && !p->have_init && !(p->opts.flags & RX_OPT_NO_FOLD) &&
                ((p->cfg.kernel == RX_KERNEL_SYM_PACK &&
                  ((p->opts.flags & RX_OPT_FORCE_FOLD) != 0 || (p->opts.kernel == RX_KERNEL_AUTO && p->auto_fold))) ||
                 p->cfg.kernel == RX_KERNEL_SYM_REG);  // the register kernel folds whenever the automaton allows
  if (p->cfg.kernel == RX_KERNEL_SYM_REG)
    p->cfg.fold = p->tab.pin_tab != nullptr;  // (their index is built for exactly that)
  if (p->cfg.fold && p->cfg.kernel == RX_KERNEL_SYM_PACK) {
    static const uint32_t fold_s[] = {8, 13, 16, 24, 32, 48, 64};
    uint32_t pick = 64;
    for (uint32_t c : fold_s) if (p->cfg.group_lanes <= c) { pick = c; break; }
    p->cfg.group_lanes = pick;
  } else if (p->cfg.kernel == RX_KERNEL_SYM_PACK && p->cfg.group_lanes > 32) {
    p->cfg.group_lanes = 32;
  }
  p->cfg.verbose = (p->opts.flags & RX_OPT_VERBOSE) != 0;
  p->cfg.profile_pack = (p->opts.flags & RX_OPT_PROFILE_PACK) != 0;
  p->cfg.reg_skip = !(p->opts.flags & RX_OPT_REG_NO_SKIP) && (p->opts.kernel == RX_KERNEL_AUTO ? p->auto_reg_skip : true);
  // look-ahead pruning of multi-target rows follows the probe: AUTO's verified choice, or for an explicit
  // RX_KERNEL_SYM_PACK what the probe's statistics say.  rx_opts.flags RX_OPT_NO_PRUNE / RX_OPT_FORCE_PRUNE override it
  // (A/B measurements; tests, whose batches are too small for a probe).
  p->cfg.prune = p->tab.symidx_p != nullptr && !(p->opts.flags & RX_OPT_NO_PRUNE) &&
                 ((p->opts.flags & RX_OPT_FORCE_PRUNE) != 0 ||
                  (p->opts.kernel == RX_KERNEL_SYM_PACK ? p->probe_prune : p->opts.kernel == RX_KERNEL_AUTO && p->auto_prune));
  const bool two_tier = p->cfg.kernel == RX_KERNEL_SYM_GROUP || p->cfg.kernel == RX_KERNEL_SYM_PACK ||
                        p->cfg.kernel == RX_KERNEL_DFA || p->cfg.kernel == RX_KERNEL_SYM_REG;
  if (two_tier && (rc = ensure_spill_area(p, a))) return rc;
  if (p->cfg.kernel == RX_KERNEL_DFA) {
    if (pair) return RX_EINVAL;
    DevTables t;
    if ((rc = ensure_dfa_tables(p->nfa, p->device, &t))) return rc;
    a.dfa_trans = t.dfa_trans;
    a.dfa_pool = t.dfa_pool;
    a.dfa_hash = t.dfa_hash;
    a.dfa_hdr = t.dfa_hdr;
    a.dfa_pool_chunks = t.dfa_pool_chunks;
    a.dfa_hash_mask = t.dfa_hash_mask;
  }
  return RX_OK;
}

extern "C" int rx_plan_launch(rx_plan* p) {
  RX_TRY
  if (!p) return RX_EINVAL;
  if (!p->have_input) return RX_ESTATE;
  int dev;
  int rc = bind_device(p->device, &dev);
  if (rc) return rc;
  if ((rc = prepare_launch(p))) return rc;
  const RxHostNfa& h = p->nfa->h;
  RxParams& a = p->params;

  // counters + match_count_total: the other set, which the previous launch's kernel has zeroed (or, after a probe or
  // the plan's creation, a reset enqueued here); this launch's kernel zeroes the one after
  const size_t set_words = 16 + (size_t)h.size;
  if (!p->sets_clean) {
    for (int q = 0; q < 2; q++) HIPCHK(hipMemsetAsync(p->d_cset[q], 0, set_words * sizeof(unsigned long long), p->stream));
    p->sets_clean = true;
  }
  p->cur_set ^= 1;
  p->d_counters = p->d_cset[p->cur_set];
  p->d_mct = p->d_counters + 16;
  a.counters = p->d_counters;
  a.ev_count = p->d_counters;
  a.match_count_total = p->d_mct;
  if (a.spill_count) a.spill_count = p->d_counters + 3;
  a.zero_next = p->d_cset[p->cur_set ^ 1];
  a.zero_words = (uint32_t)set_words;
  if (p->want_mc) HIPCHK(hipMemsetAsync(p->d_mc, 0, p->n_streams * h.size * sizeof(uint32_t), p->stream));
  if (p->n_timed >= 4096) p->n_timed = 0;  // nobody is reading the times: recycle the pool
  if (p->n_timed == p->evs.size()) {
    hipEvent_t a0 = nullptr, a1 = nullptr;
    HIPCHK(hipEventCreate(&a0));
    HIPCHK(hipEventCreate(&a1));
    p->evs.emplace_back(a0, a1);
  }
  auto& ev = p->evs[p->n_timed];
  HIPCHK(hipEventRecord(ev.first, p->stream));  // brackets the match kernel(s) only, on their own stream
  hipError_t e = (hipError_t)rx_launch(a, p->cfg, p->stream);
  if (e != hipSuccess) return hip_fail(e, "kernel launch");
  HIPCHK(hipEventRecord(ev.second, p->stream));
  p->n_timed++;
  p->launched = true;
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_plan_tune(rx_plan* p) {
  RX_TRY
  if (!p) return RX_EINVAL;
  if (!p->have_input) return RX_ESTATE;
  int dev;
  int rc = bind_device(p->device, &dev);
  if (rc) return rc;
  // AUTO's probes for the batch the plan holds, now: sample runs, timed candidates, stream synchronisation — everything
  // rx_plan_launch would otherwise do on the first batch of a shape and again on every 32nd.  The decision is pinned to
  // the shape bucket: later launches of that shape (also after other shapes in between) enqueue and return.
  p->auto_decided = false;
  p->choices.erase(p->shape_key);
  p->tuning = true;
  rc = prepare_launch(p);
  p->tuning = false;
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(p->stream));
  if (p->opts.kernel != RX_KERNEL_AUTO && !p->choices.count(p->shape_key)) store_choice(p, true);  // (nothing to decide: pin the no-op)
  auto it = p->choices.find(p->shape_key);
  if (it != p->choices.end()) it->second.pinned = true;
  p->choice_pinned = true;
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_plan_busy(rx_plan* p, uint32_t* busy) {
  RX_TRY
  if (!p || !busy) return RX_EINVAL;
  int dev;
  int rc = bind_device(p->device, &dev);
  if (rc) return rc;
  *busy = 0;
  const hipStream_t ss[4] = {p->s_in, p->s_k, p->s_out, p->stream};
  for (int i = 0; i < 4; i++) {
    if (i < 3 && !ss[i]) continue;  // (rx_plan_run's streams exist from its first call on; the launch stream may be the null stream)
    const hipError_t e = hipStreamQuery(ss[i]);
    if (e == hipErrorNotReady) { *busy |= 1u << i; (void)hipGetLastError(); }
    else if (e != hipSuccess) return hip_fail(e, "hipStreamQuery");
  }
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_plan_kernel_times(rx_plan* p, uint32_t* n_launches, double* sum_ms, double* min_ms,
                                    double* max_ms) {
  RX_TRY
  if (!p) return RX_EINVAL;
  double sum = 0, mn = 0, mx = 0;
  for (size_t i = 0; i < p->n_timed; i++) {
    HIPCHK(hipEventSynchronize(p->evs[i].second));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, p->evs[i].first, p->evs[i].second));
    sum += ms;
    mn = i == 0 ? ms : std::min<double>(mn, ms);
    mx = std::max<double>(mx, ms);
  }
  if (n_launches) *n_launches = (uint32_t)p->n_timed;
  if (sum_ms) *sum_ms = sum;
  if (min_ms) *min_ms = mn;
  if (max_ms) *max_ms = mx;
  p->n_timed = 0;
  return RX_OK;
  RX_CATCH
}

extern "C" int rx_plan_sync(rx_plan* p, double* kernel_ms) {
  RX_TRY
  if (!p) return RX_EINVAL;
  if (!p->launched) return RX_ESTATE;
  if (p->n_timed == 0) { if (kernel_ms) *kernel_ms = p->last_ms; return RX_OK; }
  auto& ev = p->evs[p->n_timed - 1];
  HIPCHK(hipEventSynchronize(ev.second));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
  p->last_ms = ms;
  if (kernel_ms) *kernel_ms = ms;
  return RX_OK;
  RX_CATCH
}

static bool ev_less(const rx_event& a, const rx_event& b) {
  if (a.stream != b.stream) return a.stream < b.stream;
  if (a.k != b.k) return a.k < b.k;
  return a.state < b.state;
}

// Canonical order (stream, k, state) of the events of streams [lo, lo + n_streams).  The device hands them over in
// arrival order — per stream already ascending in k (a wavefront's passes allocate their slots one after the other) —
// so a stable counting sort by stream does nearly everything in O(n); an insertion sort per stream finishes equal-k
// runs and anything an unusual kernel left out of order.  (std::sort on 75 000 events took 5 ms of a 8 ms call.)
// `src` (device order) -> `dst` in (stream, k, state) order; the two may not overlap
static void sort_events_into(const rx_event* src, rx_event* dst, size_t n, uint32_t lo, size_t n_streams) {
  if (n == 0) return;
  bool ok = n >= 64 && n_streams <= 8 * n + 1024;
  std::vector<uint32_t> at;
  if (ok) {
    at.assign(n_streams + 1, 0u);
    for (size_t i = 0; i < n && ok; i++) {
      const rx_event& e = src[i];
      if (e.stream < lo || e.stream - lo >= n_streams) ok = false;  // not ours: be safe
      else at[e.stream - lo + 1]++;
    }
  }
  if (!ok) {
    memcpy(dst, src, n * sizeof(rx_event));
    std::sort(dst, dst + n, ev_less);
    return;
  }
  for (size_t i = 0; i < n_streams; i++) at[i + 1] += at[i];
  {
    std::vector<uint32_t> pos(at.begin(), at.end() - 1);
    for (size_t i = 0; i < n; i++) dst[pos[src[i].stream - lo]++] = src[i];
  }
  for (size_t st = 0; st < n_streams; st++) {
    const uint32_t b = at[st], e = at[st + 1];
    for (uint32_t i = b + 1; i < e; i++) {
      const rx_event x = dst[i];
      uint32_t j = i;
      while (j > b && ev_less(x, dst[j - 1])) { dst[j] = dst[j - 1]; j--; }
      dst[j] = x;
    }
  }
}

static void sort_events(rx_event* ev, size_t n, uint32_t lo, size_t n_streams, std::vector<rx_event>& scratch) {
  if (n < 2) return;
  if (n < 64 || n_streams > 8 * n + 1024) { std::sort(ev, ev + n, ev_less); return; }
  std::vector<uint32_t> at(n_streams + 1, 0u);
  for (size_t i = 0; i < n; i++) {
    const rx_event& e = ev[i];
    if (e.stream < lo || e.stream - lo >= n_streams) { std::sort(ev, ev + n, ev_less); return; }  // not ours: be safe
    at[e.stream - lo + 1]++;
  }
  for (size_t i = 0; i < n_streams; i++) at[i + 1] += at[i];
  scratch.resize(n);
  {
    std::vector<uint32_t> pos(at.begin(), at.end() - 1);
    for (size_t i = 0; i < n; i++) scratch[pos[ev[i].stream - lo]++] = ev[i];
  }
  for (size_t st = 0; st < n_streams; st++) {
    const uint32_t b = at[st], e = at[st + 1];
    for (uint32_t i = b + 1; i < e; i++) {
      const rx_event x = scratch[i];
      uint32_t j = i;
      while (j > b && ev_less(x, scratch[j - 1])) { scratch[j] = scratch[j - 1]; j--; }
      scratch[j] = x;
    }
  }
  memcpy(ev, scratch.data(), n * sizeof(rx_event));
}

// rx_result as the caller's version of the header laid it out (see read_opts)
static size_t result_bytes(const rx_result* res) {
  return res->struct_size ? std::min<size_t>(res->struct_size, sizeof(rx_result)) : RX_RESULT_ABI1_BYTES;
}

static int plan_download(rx_plan* p, rx_result* res);

extern "C" int rx_plan_download(rx_plan* p, rx_result* caller) {
  RX_TRY
  if (!p || !caller) return RX_EINVAL;
  rx_result full{};  // work on a full-size copy: fields behind the caller's struct_size are never written to it
  const size_t have = result_bytes(caller);
  memcpy(&full, caller, have);
  const int rc = plan_download(p, &full);
  memcpy(caller, &full, have);
  return rc;
  RX_CATCH
}

static int plan_download(rx_plan* p, rx_result* res) {
  if (!p->launched) return RX_ESTATE;
  int dev;
  int rc = bind_device(p->device, &dev);
  if (rc) return rc;
  rc = rx_plan_sync(p, nullptr);
  if (rc) return rc;
  const RxHostNfa& h = p->nfa->h;
  unsigned long long cnt[16] = {0};
  HIPCHK(hipMemcpy(cnt, p->d_counters, sizeof(cnt), hipMemcpyDeviceToHost));
  rx_stats& st = res->stats;
  st = rx_stats{};
  if ((p->opts.flags & RX_OPT_PROFILE_PACK) && p->cfg.kernel == RX_KERNEL_SYM_PACK) {
    static const char* names[7] = {"list read", "accept check + window byte + filter clear", "slice gather", "filter atomics",
                                   "ballots + slots + list writes", "overflow lists", "end of pass"};
    unsigned long long tot = 0;
    for (int q = 0; q < 7; q++) tot += cnt[8 + q];
    for (int q = 0; q < 7 && tot; q++)
      fprintf(stderr, "[rxmatch] pack pass phase %d %-44s %5.1f %%  (%.0f cycles per wave-pass)\n", q, names[q],
              100.0 * cnt[8 + q] / tot, (double)cnt[8 + q] / ((double)((p->n_streams + 15) / 16) * p->params.n_passes));
    if (cnt[15] & 0xFFFFFFFFull)
      fprintf(stderr, "[rxmatch] pack kernel (stamped build), wave 0: %llu shader cycles in %.3f ms = %.0f MHz under load\n", (cnt[15] >> 32) << 6,
              (double)(cnt[15] & 0xFFFFFFFFull) * 1e-5, (double)((cnt[15] >> 32) << 6) / (double)(cnt[15] & 0xFFFFFFFFull) * 100.0);
  }
  if ((p->opts.flags & RX_OPT_VERBOSE) && cnt[3])
    fprintf(stderr, "[rxmatch] %llu of %u streams were handed to the wave kernel\n", cnt[3], (unsigned)p->n_streams);
  if ((p->opts.flags & RX_OPT_VERBOSE) && p->cfg.kernel == RX_KERNEL_SYM_REG && cnt[9])
    fprintf(stderr, "[rxmatch] register kernel, stream 0: %llu shader cycles in %.3f ms = %.0f MHz, %.0f cycles per pass\n", cnt[8],
            cnt[9] * 1e-5, (double)cnt[8] / cnt[9] * 100.0, (double)cnt[8] / std::max<uint32_t>(p->params.n_passes, 1));
  st.n_passes = p->params.n_passes;
  st.n_events = cnt[0];
  st.kernel_ms = p->last_ms;
  st.kernel_used = p->cfg.kernel;
  st.lanes_used = (p->cfg.kernel == RX_KERNEL_SYM_GROUP || p->cfg.kernel == RX_KERNEL_SYM_PACK) ? p->cfg.group_lanes : 0u;
  st.variant = (p->cfg.stats ? RX_VARIANT_STATS : 0u) |
               (p->cfg.kernel == RX_KERNEL_SYM_PACK && p->cfg.prune && !p->cfg.stats && p->tab.symidx_p ? RX_VARIANT_PRUNE : 0u) |
               (p->cfg.fold ? RX_VARIANT_FOLD : 0u);
  st.n_launches = (p->cfg.kernel == RX_KERNEL_SYM_GROUP || p->cfg.kernel == RX_KERNEL_SYM_PACK ||
                   p->cfg.kernel == RX_KERNEL_DFA || p->cfg.kernel == RX_KERNEL_SYM_REG) ? 2 : 1;
  if (p->cfg.stats) {
    st.sum_active = cnt[1];
    st.sum_edges = cnt[2];
    // 1 B per consumed byte + 8 B per active state + 4 B per edge of its row
    // + 1 bit per pass (per stream, rounded up to bytes) + 12 B per accept event
    st.alg_bytes = (uint64_t)p->params.n_consume * p->n_streams + 8 * st.sum_active + 4 * st.sum_edges +
                   (uint64_t)p->n_streams * ((st.n_passes + 7) / 8) + 12 * st.n_events;
    // per pair 1 reset clock + per pass [size + sum over states active in either stream of
    // (cost - 1)].  Only defined if no stream left the pack kernel (cnt[3] = handed-off streams).
    if (p->params.pair_cycles && cnt[3] == 0)
      st.tb_cycles = (p->n_streams / 2) * (1 + (uint64_t)p->params.n_consume * h.size) + cnt[4];
  }
  const size_t captured = (size_t)std::min<unsigned long long>(cnt[0], p->events_cap);
  res->events_overflow = cnt[0] > p->events_cap ? 1u : 0u;
  res->n_events = 0;
  if (res->events && res->events_cap && captured) {
    std::vector<rx_event> tmp(captured), scratch;
    HIPCHK(hipMemcpy(tmp.data(), p->d_events, captured * sizeof(rx_event), hipMemcpyDeviceToHost));
    sort_events(tmp.data(), tmp.size(), p->params.stream_base, p->n_streams, scratch);  // device order is arrival order; canonical = (stream,k,state)
    const size_t n = std::min(captured, res->events_cap);
    memcpy(res->events, tmp.data(), n * sizeof(rx_event));
    res->n_events = n;
    if (captured > res->events_cap) res->events_overflow = 1u;
  } else if (cnt[0] && (!res->events || !res->events_cap)) {
    res->events_overflow = res->events ? 1u : 0u;
  }
  if (res->match_count) {
    if (!p->want_mc) return RX_ESTATE;
    HIPCHK(hipMemcpy(res->match_count, p->d_mc, p->n_streams * h.size * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  if (res->match_count_total)
    HIPCHK(hipMemcpy(res->match_count_total, p->d_mct, (size_t)h.size * sizeof(uint64_t), hipMemcpyDeviceToHost));
  if (res->anymatch) {
    if (!p->want_am) return RX_ESTATE;
    const size_t need = (size_t)((st.n_passes + 31) / 32);
    if (res->anymatch_stride < need) return RX_EINVAL;
    if (need && res->anymatch_stride == p->am_stride)  // same pitch on both sides: one flat copy (2-D copies go row by row)
      HIPCHK(hipMemcpy(res->anymatch, p->d_am, p->n_streams * p->am_stride * 4, hipMemcpyDeviceToHost));
    else if (need)
      HIPCHK(hipMemcpy2D(res->anymatch, res->anymatch_stride * 4, p->d_am, p->am_stride * 4, need * 4, p->n_streams,
                         hipMemcpyDeviceToHost));
  }
  if (res->final_active) {
    if (!p->want_final) return RX_ESTATE;
    HIPCHK(hipMemcpy(res->final_active, p->d_final, p->n_streams * p->params.nw64x2 * sizeof(uint32_t),
                     hipMemcpyDeviceToHost));
  }
  return RX_OK;
}

// ---- pipelined host-to-host run ---------------------------------------------------------------------
extern "C" int rx_host_register(void* ptr, size_t bytes) {
  if (!ptr || !bytes) return RX_EINVAL;
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return RX_OK; }
  if (e != hipSuccess) return hip_fail(e, "hipHostRegister");
  return RX_OK;
}
extern "C" int rx_host_unregister(void* ptr) {
  if (!ptr) return RX_EINVAL;
  hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) return hip_fail(e, "hipHostUnregister");
  return RX_OK;
}

static int plan_run(rx_plan* p, const uint8_t* bytes, size_t n_streams, size_t stream_len, size_t stride, rx_result* res);

extern "C" int rx_plan_run(rx_plan* p, const uint8_t* bytes, size_t n_streams, size_t stream_len, size_t stride,
                           rx_result* caller) {
  RX_TRY
  if (!p || !caller || (!bytes && stream_len)) return RX_EINVAL;
  rx_result full{};
  const size_t have = result_bytes(caller);
  memcpy(&full, caller, have);
  const int rc = plan_run(p, bytes, n_streams, stream_len, stride, &full);
  memcpy(caller, &full, have);
  return rc;
  RX_CATCH
}

static int plan_run_body(rx_plan* p, const uint8_t* bytes, size_t n_streams, size_t stream_len, size_t stride, rx_result* res,
                         bool* enqueued);

// Error discipline of the pipelined call: everything that can be checked without the device is checked before the first
// enqueue; once copies are in flight from / into the caller's arrays, EVERY exit that reports a failure first waits for the
// three streams — the caller is free to release its buffers as soon as it sees the error code.
static int plan_run(rx_plan* p, const uint8_t* bytes, size_t n_streams, size_t stream_len, size_t stride, rx_result* res) {
  bool enqueued = false;
  const int rc = plan_run_body(p, bytes, n_streams, stream_len, stride, res, &enqueued);
  if (rc != RX_OK && enqueued) {
    const std::string keep = g_last_hip;  // (the drain must not replace the text of the failure that is being reported)
    if (p->s_in) (void)hipStreamSynchronize(p->s_in);
    if (p->s_k) (void)hipStreamSynchronize(p->s_k);
    if (p->s_out) (void)hipStreamSynchronize(p->s_out);
    (void)hipStreamSynchronize(p->stream);
    (void)hipGetLastError();
    g_last_hip = keep;
    p->launched = false;
    p->sets_clean = false;
  }
  return rc;
}

static int plan_run_body(rx_plan* p, const uint8_t* bytes, size_t n_streams, size_t stream_len, size_t stride, rx_result* res,
                         bool* enqueued) {
  int dev;
  int rc = bind_device(p->device, &dev);
  if (rc) return rc;
  const RxHostNfa& h = p->nfa->h;
  const uint32_t size = h.size;
  const size_t nw64 = ((size_t)size + 63) / 64, set_words = 16 + (size_t)size;
  // ---- checks that need no device state (nothing has been enqueued yet) ----
  if (n_streams == 0 || n_streams > p->max_streams || stream_len > p->max_len || stride < stream_len) return RX_EINVAL;
  if ((res->match_count && !p->want_mc) || (res->anymatch && !p->want_am) || (res->final_active && !p->want_final)) return RX_ESTATE;
  const bool compact = res->final_states || res->final_off || res->final_cnt;
  if (compact) {
    if (!res->final_states || !res->final_off || !res->final_cnt || res->final_states_cap == 0) return RX_EINVAL;
    if (!p->want_final) return RX_ESTATE;
    if (res->final_states_cap > 0xFFFFFFFFull) return RX_EINVAL;
  }
  {
    const uint64_t passes = passes_for(stream_len, p->opts.mode);
    if (p->opts.k_base + passes > (1ull << 32)) return RX_EINVAL;
    if (res->anymatch && res->anymatch_stride < (size_t)((passes + 31) / 32)) return RX_EINVAL;
    if (p->opts.collect_stats == 2 &&
        ((n_streams & 1) || (p->opts.kernel != RX_KERNEL_AUTO && p->opts.kernel != RX_KERNEL_SYM_PACK)))
      return RX_EINVAL;
  }
  // a preceding rx_plan_launch may still be reading the plan's input and output buffers on the plan's own stream
  HIPCHK(hipStreamSynchronize(p->stream));
  if ((rc = set_batch(p, n_streams, stream_len, stride))) return rc;
  if (compact) {
    if (res->final_states_cap > p->fstates_cap) {
      (void)hipFree(p->d_fstates);
      p->d_fstates = nullptr;
      p->fstates_cap = 0;
      HIPCHK(hipMalloc((void**)&p->d_fstates, res->final_states_cap * sizeof(uint32_t)));
      p->fstates_cap = res->final_states_cap;
    }
    if (!p->d_foff) {
      HIPCHK(hipMalloc((void**)&p->d_foff, p->max_streams * sizeof(uint32_t)));
      HIPCHK(hipMalloc((void**)&p->d_fcnt, p->max_streams * sizeof(uint32_t)));
    }
  }
  // input buffer of the plan, rows at a 4-byte-aligned pitch
  const size_t pitch = (stream_len + 3) & ~(size_t)3;
  const size_t need = std::max<size_t>(n_streams * pitch, 4);
  if (need > p->d_in_own_bytes) {
    (void)hipFree(p->d_in_own);
    p->d_in_own = nullptr;
    p->d_in_own_bytes = 0;
    HIPCHK(hipMalloc((void**)&p->d_in_own, need));
    p->d_in_own_bytes = need;
  }
  p->d_in = p->d_in_own;
  p->stride = pitch;
  // blocks of streams: 32 768 or more each (smaller launches leave SIMDs idle), at most EIGHT, sizes a multiple of
  // 1 024 (lock-step pairs stay together).  THREE HIP streams shared by all blocks: uploads, kernels, downloads — a copy
  // runs beside a kernel, the two copy directions share the link (measured on the MI355X box: 56 GB/s in either direction
  // or in both together), so the pipeline's floor is (input + output bytes) / 56 GB/s.
  constexpr size_t MAX_BLOCKS = 8;
  size_t n_blocks = std::min<size_t>(MAX_BLOCKS, std::max<size_t>(1, n_streams / 32768));
  size_t per = (n_streams + n_blocks - 1) / n_blocks;
  per = (per + 1023) & ~(size_t)1023;
  n_blocks = (n_streams + per - 1) / per;
  if (!p->s_in) {
    HIPCHK(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&p->s_k, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
  }
  if (!p->d_run_ctr) {
    HIPCHK(hipMalloc((void**)&p->d_run_ctr, (2 + 2 * MAX_BLOCKS) * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc((void**)&p->h_run_ctr, (2 + 2 * MAX_BLOCKS) * sizeof(unsigned long long), hipHostMallocDefault));
  }
  while (p->pipes.size() < n_blocks) {
    rx_plan::Pipe q;
    HIPCHK(hipMalloc((void**)&q.d_set, set_words * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc((void**)&q.h_set, set_words * sizeof(unsigned long long), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&q.up, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&q.k0));
    HIPCHK(hipEventCreate(&q.k1));
    p->pipes.push_back(q);
  }
  auto upload = [&](size_t b) -> int {
    const size_t s0 = b * per, cnt = std::min(per, n_streams - s0);
    *enqueued = true;
    if (stream_len) {
      if (stride == pitch && stream_len == pitch)
        HIPCHK(hipMemcpyAsync(p->d_in_own + s0 * pitch, bytes + s0 * stride, cnt * pitch, hipMemcpyHostToDevice, p->s_in));
      else
        HIPCHK(hipMemcpy2DAsync(p->d_in_own + s0 * pitch, pitch, bytes + s0 * stride, stride, stream_len, cnt, hipMemcpyHostToDevice,
                                p->s_in));
    }
    HIPCHK(hipEventRecord(p->pipes[b].up, p->s_in));
    return RX_OK;
  };
  // block 0 goes up first: AUTO's probe (when its decision is not valid for this batch) reads a corner of it
  if ((rc = upload(0))) return rc;
  if (!p->auto_decided) HIPCHK(hipStreamSynchronize(p->s_in));
  if ((rc = prepare_launch(p))) return rc;
  HIPCHK(hipStreamSynchronize(p->stream));  // (the probe ran on the plan's own stream)
  const size_t am_need = (size_t)((p->params.n_passes + 31) / 32);
  const bool two_tier = p->cfg.kernel == RX_KERNEL_SYM_GROUP || p->cfg.kernel == RX_KERNEL_SYM_PACK ||
                        p->cfg.kernel == RX_KERNEL_DFA || p->cfg.kernel == RX_KERNEL_SYM_REG;
  for (size_t b = 1; b < n_blocks; b++)  // all uploads are queued before any download (both use the same link)
    if ((rc = upload(b))) return rc;
  // ONE capacity for the whole call: the blocks' kernels run one after the other on the kernel stream and take their
  // event slots from one counter over the plan's whole event buffer (and the compaction kernels theirs from one counter
  // over the caller-sized list buffer); a snapshot of both counters behind every block tells the host where the block's
  // part ends.  A call whose events all lie in one block loses none as long as the total fits.
  HIPCHK(hipMemsetAsync(p->d_run_ctr, 0, (2 + 2 * MAX_BLOCKS) * sizeof(unsigned long long), p->s_k));
  for (size_t b = 0; b < n_blocks; b++) {
    rx_plan::Pipe& q = p->pipes[b];
    const size_t s0 = b * per, cnt = std::min(per, n_streams - s0);
    RxParams a = p->params;  // the block's view of the batch
    RxLaunchCfg cfg = p->cfg;
    a.bytes = p->d_in_own + s0 * pitch;
    a.n_streams = (uint32_t)cnt;
    a.stream_base = (uint32_t)s0;
    a.events = p->events_cap ? p->d_events : nullptr;
    a.events_cap = (uint32_t)p->events_cap;
    a.ev_count = p->d_run_ctr;
    a.counters = q.d_set;
    a.match_count_total = q.d_set + 16;
    a.zero_next = nullptr;
    a.zero_words = 0;
    if (a.match_count) a.match_count += s0 * size;
    if (a.anymatch) a.anymatch += s0 * p->am_stride;
    if (a.final_active) a.final_active += s0 * (size_t)a.nw64x2;
    if (two_tier) {
      a.spill_count = q.d_set + 3;
      a.spill_streams += s0;
      a.spill_k += s0;
      a.spill_rows += s0 * (size_t)a.nw64x2;
    }
    const uint32_t lanes = cfg.group_lanes;
    if ((rc = rx_pick_launch(cfg.kernel, size, a.n_streams, p->tab.cu_count, p->tab.lds_per_cu, &a, &cfg))) return rc;
    cfg.group_lanes = lanes;
    HIPCHK(hipMemsetAsync(q.d_set, 0, set_words * sizeof(unsigned long long), p->s_k));
    if (p->want_mc) HIPCHK(hipMemsetAsync(p->d_mc + s0 * size, 0, cnt * size * sizeof(uint32_t), p->s_k));
    HIPCHK(hipStreamWaitEvent(p->s_k, q.up, 0));
    HIPCHK(hipEventRecord(q.k0, p->s_k));
    // compact final sets: the pack kernel (and the wave kernel behind it) writes the lists itself and builds no rows; the
    // other kernels leave rows, which a small kernel behind them turns into lists
    const bool direct = compact && cfg.kernel == RX_KERNEL_SYM_PACK;
    if (direct) {
      a.fin_states = p->d_fstates;
      a.fin_cap = (uint32_t)res->final_states_cap;
      a.fin_off = p->d_foff + s0;
      a.fin_cnt = p->d_fcnt + s0;
      a.fin_count = p->d_run_ctr + 1;
    }
    hipError_t e = (hipError_t)rx_launch(a, cfg, p->s_k);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    if (compact && !direct) {  // the lists of all blocks share the caller's capacity; offsets are positions in the whole buffer
      e = (hipError_t)rx_launch_final_compact(a.final_active, a.n_streams, a.nw64x2, p->d_fstates, (uint32_t)res->final_states_cap,
                                              p->d_foff + s0, p->d_fcnt + s0, p->d_run_ctr + 1, p->s_k);
      if (e != hipSuccess) return hip_fail(e, "final-set compaction launch");
    }
    HIPCHK(hipMemcpyAsync(p->d_run_ctr + 2 + b, p->d_run_ctr, sizeof(unsigned long long), hipMemcpyDeviceToDevice, p->s_k));
    HIPCHK(hipMemcpyAsync(p->d_run_ctr + 2 + MAX_BLOCKS + b, p->d_run_ctr + 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, p->s_k));
    HIPCHK(hipEventRecord(q.k1, p->s_k));
    // results of the block straight into the caller's arrays
    HIPCHK(hipStreamWaitEvent(p->s_out, q.k1, 0));
    HIPCHK(hipMemcpyAsync(q.h_set, q.d_set, set_words * sizeof(unsigned long long), hipMemcpyDeviceToHost, p->s_out));
    if (res->final_active)
      HIPCHK(hipMemcpyAsync(res->final_active + s0 * nw64, p->d_final + s0 * (size_t)a.nw64x2, cnt * nw64 * sizeof(uint64_t),
                            hipMemcpyDeviceToHost, p->s_out));
    if (compact) {
      HIPCHK(hipMemcpyAsync(res->final_off + s0, p->d_foff + s0, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, p->s_out));
      HIPCHK(hipMemcpyAsync(res->final_cnt + s0, p->d_fcnt + s0, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, p->s_out));
    }
    if (res->anymatch && am_need) {
      if (res->anymatch_stride == p->am_stride)  // same pitch on both sides: one flat copy (2-D copies go row by row)
        HIPCHK(hipMemcpyAsync(res->anymatch + s0 * p->am_stride, p->d_am + s0 * p->am_stride, cnt * p->am_stride * 4,
                              hipMemcpyDeviceToHost, p->s_out));
      else
        HIPCHK(hipMemcpy2DAsync(res->anymatch + s0 * res->anymatch_stride, res->anymatch_stride * 4, p->d_am + s0 * p->am_stride,
                                p->am_stride * 4, am_need * 4, cnt, hipMemcpyDeviceToHost, p->s_out));
    }
    if (res->match_count)
      HIPCHK(hipMemcpyAsync(res->match_count + s0 * size, p->d_mc + s0 * size, cnt * size * sizeof(uint32_t), hipMemcpyDeviceToHost,
                            p->s_out));
    // (test hook, RX_OPT_INJECT_RUN_FAULT: fail here, with block 0's kernels and copies in flight)
    if (b == 0 && (p->opts.flags & RX_OPT_INJECT_RUN_FAULT)) {
      g_last_hip = "injected fault (RX_OPT_INJECT_RUN_FAULT)";
      return RX_EHIP;
    }
  }
  // the counters and their per-block snapshots (the last block's k1 orders this copy behind every kernel)
  HIPCHK(hipMemcpyAsync(p->h_run_ctr, p->d_run_ctr, (2 + 2 * MAX_BLOCKS) * sizeof(unsigned long long), hipMemcpyDeviceToHost, p->s_out));
  const bool verbose = (p->opts.flags & RX_OPT_VERBOSE) != 0;
  const auto w_issued = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(p->s_out));
  if (verbose)
    fprintf(stderr, "[rxmatch] run: %zu block(s) of <= %zu streams, all results down %.3f ms after the last enqueue\n", n_blocks, per,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w_issued).count());
  // collect: counters, events (sorted per block; blocks are in stream order, so the concatenation is sorted)
  rx_stats& st = res->stats;
  st = rx_stats{};
  st.n_passes = p->params.n_passes;
  st.kernel_used = p->cfg.kernel;
  st.lanes_used = (p->cfg.kernel == RX_KERNEL_SYM_GROUP || p->cfg.kernel == RX_KERNEL_SYM_PACK) ? p->cfg.group_lanes : 0u;
  st.variant = (p->cfg.stats ? RX_VARIANT_STATS : 0u) |
               (p->cfg.kernel == RX_KERNEL_SYM_PACK && p->cfg.prune && !p->cfg.stats && p->tab.symidx_p ? RX_VARIANT_PRUNE : 0u) |
               (p->cfg.fold ? RX_VARIANT_FOLD : 0u);
  res->n_events = 0;
  res->events_overflow = 0;
  if (res->match_count_total) memset(res->match_count_total, 0, (size_t)size * sizeof(uint64_t));
  unsigned long long spilled = 0, pair_cost = 0;
  for (size_t b = 0; b < n_blocks; b++) {
    rx_plan::Pipe& q = p->pipes[b];
    float ms = 0;
    (void)hipEventElapsedTime(&ms, q.k0, q.k1);
    st.kernel_ms += ms;
    st.n_launches += two_tier ? 2 : 1;
    const unsigned long long* cnt = q.h_set;
    st.sum_active += cnt[1];
    st.sum_edges += cnt[2];
    spilled += cnt[3];
    pair_cost += cnt[4];
    if (res->match_count_total)
      for (uint32_t i = 0; i < size; i++) res->match_count_total[i] += cnt[16 + i];
  }
  const unsigned long long* ev_after = p->h_run_ctr + 2;  // accept events of blocks 0..b
  const unsigned long long ev_total = p->h_run_ctr[0];
  st.n_events = ev_total;
  if (ev_total > p->events_cap && res->events) res->events_overflow = 1u;
  const size_t captured = (size_t)std::min<unsigned long long>(ev_total, p->events_cap);
  // The two downloads whose sizes the host has only now: both go out together, into page-locked staging (a blocking copy
  // into the caller's pageable arrays, one after the other, was 0.3 of the 2.5 ms of a configs[2] call); the events are put
  // into order on their way from the staging buffer to the caller's array.
  auto stage = [](void** buf, size_t* have, size_t need) -> int {
    if (need <= *have) return RX_OK;
    if (*buf) (void)hipHostFree(*buf);
    *buf = nullptr;
    *have = 0;
    const size_t want = need + need / 2;
    HIPCHK(hipHostMalloc(buf, want, hipHostMallocDefault));
    *have = want;
    return RX_OK;
  };
  const bool ev_down = res->events && res->events_cap && captured;
  size_t fs_n = 0;
  if (compact) {
    // (final_off is a position in final_states; a set that did not fit wholly is cut at the capacity, final_cnt stays exact)
    const unsigned long long needed = p->h_run_ctr[1];
    fs_n = (size_t)std::min<unsigned long long>(needed, res->final_states_cap);
    res->final_states_overflow = needed > res->final_states_cap ? 1u : 0u;
    res->n_final_states = fs_n;
  }
  if (ev_down) {
    if ((rc = stage(&p->h_stage_ev, &p->h_stage_ev_bytes, captured * sizeof(rx_event)))) return rc;
    HIPCHK(hipMemcpyAsync(p->h_stage_ev, p->d_events, captured * sizeof(rx_event), hipMemcpyDeviceToHost, p->s_out));
  }
  if (fs_n) {
    if ((rc = stage(&p->h_stage_fs, &p->h_stage_fs_bytes, fs_n * sizeof(uint32_t)))) return rc;
    HIPCHK(hipMemcpyAsync(p->h_stage_fs, p->d_fstates, fs_n * sizeof(uint32_t), hipMemcpyDeviceToHost, p->s_in));
  }
  if (ev_down) {
    HIPCHK(hipStreamSynchronize(p->s_out));
    // the device buffer holds the blocks' events back to back in launch order (the first `captured` slots); each block's
    // part is brought into (stream, k, state) order on its own (blocks are in stream order: the concatenation is sorted)
    const rx_event* src = static_cast<const rx_event*>(p->h_stage_ev);
    const size_t n = std::min(captured, res->events_cap);
    if (n == captured) {
      for (size_t b = 0; b < n_blocks; b++) {
        const size_t lo = (size_t)std::min<unsigned long long>(b ? ev_after[b - 1] : 0ull, captured);
        const size_t hi = (size_t)std::min<unsigned long long>(ev_after[b], captured);
        if (hi > lo) sort_events_into(src + lo, res->events + lo, hi - lo, (uint32_t)(b * per), std::min(per, n_streams - b * per));
      }
    } else {  // the caller's array is shorter than what was captured: order everything, hand over the first n
      std::vector<rx_event> tmp(src, src + captured), scratch;
      for (size_t b = 0; b < n_blocks; b++) {
        const size_t lo = (size_t)std::min<unsigned long long>(b ? ev_after[b - 1] : 0ull, captured);
        const size_t hi = (size_t)std::min<unsigned long long>(ev_after[b], captured);
        if (hi > lo) sort_events(tmp.data() + lo, hi - lo, (uint32_t)(b * per), std::min(per, n_streams - b * per), scratch);
      }
      memcpy(res->events, tmp.data(), n * sizeof(rx_event));
      res->events_overflow = 1u;
    }
    res->n_events = n;
  }
  if (fs_n) {
    HIPCHK(hipStreamSynchronize(p->s_in));
    memcpy(res->final_states, p->h_stage_fs, fs_n * sizeof(uint32_t));
  }
  if (verbose)
    fprintf(stderr, "[rxmatch] run: events and final lists on the host %.3f ms after the last enqueue\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w_issued).count());
  if (p->cfg.stats) {
    st.alg_bytes = (uint64_t)p->params.n_consume * n_streams + 8 * st.sum_active + 4 * st.sum_edges +
                   (uint64_t)n_streams * ((st.n_passes + 7) / 8) + 12 * st.n_events;
    if (p->params.pair_cycles && spilled == 0)
      st.tb_cycles = (n_streams / 2) * (1 + (uint64_t)p->params.n_consume * size) + pair_cost;
  }
  p->launched = false;  // (nothing is left on the device for rx_plan_download)
  p->sets_clean = false;
  return RX_OK;
}

// ---- one-shot -------------------------------------------------------------------------------------
extern "C" int rx_match(const rx_nfa* nfa, const uint8_t* bytes, size_t n_streams, size_t stream_len, size_t stride,
                        const uint64_t* init_active, const rx_opts* opts, rx_result* res) {
  RX_TRY
  if (!nfa || !res || (!bytes && stream_len) || n_streams == 0 || stride < stream_len) return RX_EINVAL;
  if (read_opts(opts).k_base + passes_for(stream_len, RX_MODE_FULL) > (1ull << 32)) return RX_EINVAL;
  rx_plan* p = nullptr;
  // (compact final sets: only when the caller's struct has the fields, and only on the pipelined path below)
  const bool has_compact = result_bytes(res) >= offsetof(rx_result, final_states_overflow) + sizeof(uint32_t);
  if (has_compact && res->final_states && init_active) return RX_EINVAL;
  int rc = rx_plan_create(nfa, opts, n_streams, stream_len, res->events ? res->events_cap : 0,
                          res->match_count != nullptr, res->anymatch != nullptr,
                          res->final_active != nullptr || (has_compact && res->final_states != nullptr), &p);
  if (rc) return rc;
  auto done = [&](int code) {
    rx_plan_free(p);
    return code;
  };
  if (!init_active) {
    // streams from reset: the pipelined path (blocks of streams, upload / kernel / download overlapped)
    const auto w0 = std::chrono::steady_clock::now();
    rc = rx_plan_run(p, bytes, n_streams, stream_len, stride, res);
    if (rc == RX_OK && result_bytes(res) >= offsetof(rx_result, stats) + offsetof(rx_stats, d2h_ms) + sizeof(double)) {
      res->stats.h2d_ms = 0;  // (the copies overlap the kernels: there is no separate figure)
      res->stats.d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();  // whole call
    }
    return done(rc);
  }
  hipEvent_t t0 = nullptr, t1 = nullptr;
  auto done2 = [&](int code) {
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    return done(code);
  };
  if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) return done2(RX_EHIP);
  (void)hipEventRecord(t0, p->stream);
  if ((rc = rx_plan_upload(p, bytes, n_streams, stream_len, stride))) return done2(rc);
  if ((rc = rx_plan_set_init_active(p, init_active))) return done2(rc);
  (void)hipEventRecord(t1, p->stream);
  if ((rc = rx_plan_launch(p))) return done2(rc);
  if ((rc = rx_plan_sync(p, nullptr))) return done2(rc);
  const auto w0 = std::chrono::steady_clock::now();
  if ((rc = rx_plan_download(p, res))) return done2(rc);  // honours res->struct_size
  const auto w1 = std::chrono::steady_clock::now();
  float h2d = 0;
  (void)hipEventElapsedTime(&h2d, t0, t1);
  res->stats.h2d_ms = h2d;
  res->stats.d2h_ms = std::chrono::duration<double, std::milli>(w1 - w0).count();
  return done2(RX_OK);
  RX_CATCH
}

// ---- multi-GPU: contiguous stream blocks, one host thread per device, no collective --------------
extern "C" int rx_match_sharded(const rx_nfa* nfa, const uint8_t* bytes, size_t n_streams, size_t stream_len,
                                size_t stride, const int* devices, int n_devices, const rx_opts* opts,
                                rx_result* res) {
  RX_TRY
  if (!nfa || !res || n_devices <= 0 || n_streams == 0 || stride < stream_len) return RX_EINVAL;
  // (the list form of the final sets is per device; the sharded call returns rows)
  if (result_bytes(res) >= offsetof(rx_result, final_states_overflow) + sizeof(uint32_t) && (res->final_states || res->final_off || res->final_cnt))
    return RX_EINVAL;
  const int nd = (int)std::min<size_t>((size_t)n_devices, n_streams);
  const uint32_t size = nfa->h.size;
  const size_t nw64 = ((size_t)size + 63) / 64;
  struct Shard {
    size_t s0 = 0, n = 0;
    rx_result r{};
    std::vector<rx_event> ev;
    std::vector<uint64_t> mct;
    int rc = RX_OK;
    std::string err;
  };
  std::vector<Shard> sh((size_t)nd);
  const size_t per = n_streams / (size_t)nd, rem = n_streams % (size_t)nd;
  size_t s = 0;
  for (int d = 0; d < nd; d++) {  // remainder to the low ranks
    sh[d].s0 = s;
    sh[d].n = per + ((size_t)d < rem ? 1 : 0);
    s += sh[d].n;
  }
  std::vector<std::thread> th;
  for (int d = 0; d < nd; d++) {
    th.emplace_back([&, d]() {
      Shard& x = sh[d];
      rx_opts o = read_opts(opts);
      o.struct_size = sizeof(rx_opts);
      o.device = devices ? devices[d] : d;
      o.stream = nullptr;  // a stream handle belongs to one device
      x.r = rx_result{};
      x.r.struct_size = sizeof(rx_result);
      if (res->events && res->events_cap) {
        x.ev.resize(res->events_cap);
        x.r.events = x.ev.data();
        x.r.events_cap = res->events_cap;
      }
      if (res->match_count) x.r.match_count = res->match_count + x.s0 * size;
      if (res->match_count_total) { x.mct.assign(size, 0); x.r.match_count_total = x.mct.data(); }
      if (res->anymatch) { x.r.anymatch = res->anymatch + x.s0 * res->anymatch_stride; x.r.anymatch_stride = res->anymatch_stride; }
      if (res->final_active) x.r.final_active = res->final_active + x.s0 * nw64;
      x.rc = rx_match(nfa, bytes + x.s0 * stride, x.n, stream_len, stride, nullptr, &o, &x.r);
      if (x.rc) x.err = rx_last_hip_error();
    });
  }
  for (auto& t : th) t.join();
  {  // zero the statistics the caller's struct has room for
    rx_stats z{};
    const size_t off = offsetof(rx_result, stats), have = result_bytes(res);
    if (have > off) memcpy(&res->stats, &z, std::min(sizeof(rx_stats), have - off));
  }
  res->n_events = 0;
  res->events_overflow = 0;
  if (res->match_count_total) memset(res->match_count_total, 0, (size_t)size * sizeof(uint64_t));
  for (int d = 0; d < nd; d++) {
    Shard& x = sh[d];
    if (x.rc) { g_last_hip = x.err; return x.rc; }
    for (size_t e = 0; e < x.r.n_events; e++) {  // shards are in stream order => output stays sorted
      if (res->n_events < res->events_cap) {
        rx_event ev = x.r.events[e];
        ev.stream += (uint32_t)x.s0;
        res->events[res->n_events++] = ev;
      } else {
        res->events_overflow = 1;
      }
    }
    if (x.r.events_overflow) res->events_overflow = 1;
    if (res->match_count_total)
      for (uint32_t i = 0; i < size; i++) res->match_count_total[i] += x.mct[i];
    res->stats.n_passes = x.r.stats.n_passes;
    res->stats.n_events += x.r.stats.n_events;
    res->stats.sum_active += x.r.stats.sum_active;
    res->stats.sum_edges += x.r.stats.sum_edges;
    res->stats.alg_bytes += x.r.stats.alg_bytes;
    res->stats.kernel_ms = std::max(res->stats.kernel_ms, x.r.stats.kernel_ms);  // slowest device
    res->stats.h2d_ms = std::max(res->stats.h2d_ms, x.r.stats.h2d_ms);
    res->stats.d2h_ms = std::max(res->stats.d2h_ms, x.r.stats.d2h_ms);
    res->stats.kernel_used = x.r.stats.kernel_used;
    if (result_bytes(res) >= offsetof(rx_result, stats) + sizeof(rx_stats)) { res->stats.lanes_used = x.r.stats.lanes_used; res->stats.variant = x.r.stats.variant; }
    res->stats.n_launches += x.r.stats.n_launches;
    res->stats.tb_cycles += x.r.stats.tb_cycles;  // pairs never straddle shards when every shard is even-sized
  }
  return RX_OK;
  RX_CATCH
}
