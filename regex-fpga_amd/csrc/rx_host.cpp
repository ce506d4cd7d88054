// rx_host.cpp — host-side loaders for the reference's two file formats and the load-time
// derivation of the per-(state,symbol) slice index.  Plain C++17, no HIP, no oracle code.
//
//   .coe  : Block_Mem/CSR_BlockMem*.coe — Xilinx COE, radix 16, one 128-bit token per BRAM line.
//           Token -> four u32, leftmost 8 hex digits = word 0 (Design/FPGA.v:881-884:
//           cache[0] = rd_bus[127:96]).  The word array is kept UNCHANGED; it is what goes to HBM.
//   .mem  : Simulation/input_trace_*.mem — $readmemh text read by testbench_BLK_Mem.sv:34-35.
//   table : W[0..size] = row_ptr, W[size+1+row_ptr[i]+j] = (symbol<<24 | target)
//           (Design/FPGA.v:773,793 offset = size+1; :888-898 field split).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "rx_internal.hpp"

static inline int hexv(unsigned char c) {
  if (c >= '0' && c <= '9') return c - '0';
  c |= 0x20;
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  return -1;
}
static inline bool is_sep(unsigned char c) {
  return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == ',' || c == ';' || c == '\f' || c == '\v';
}

int rxh_read_file(const char* path, std::string* out) {
  if (!path) return RX_EINVAL;
  std::ifstream f(path, std::ios::binary);
  if (!f) return RX_EIO;
  std::ostringstream ss;
  ss << f.rdbuf();
  if (f.bad()) return RX_EIO;
  *out = ss.str();
  return RX_OK;
}

int rxh_parse_coe_text(const char* text, size_t len, std::vector<uint32_t>* words) {
  const std::string s(text, len);
  // header: memory_initialization_radix=16; memory_initialization_vector=
  size_t r = s.find("memory_initialization_radix");
  if (r == std::string::npos) return RX_EFORMAT;
  size_t eq = s.find('=', r);
  if (eq == std::string::npos) return RX_EFORMAT;
  size_t q = eq + 1;
  while (q < s.size() && (s[q] == ' ' || s[q] == '\t')) q++;
  if (s.compare(q, 2, "16") != 0) return RX_EFORMAT;  // only radix 16 is produced by the reference
  size_t v = s.find("memory_initialization_vector");
  if (v == std::string::npos) return RX_EFORMAT;
  eq = s.find('=', v);
  if (eq == std::string::npos) return RX_EFORMAT;
  words->clear();
  words->reserve((s.size() - eq) / 33 * 4 + 8);
  size_t i = eq + 1;
  while (i < s.size()) {
    while (i < s.size() && is_sep((unsigned char)s[i])) i++;
    if (i >= s.size()) break;
    size_t j = i;
    while (j < s.size() && hexv((unsigned char)s[j]) >= 0) j++;
    if (j - i != 32) return RX_EFORMAT;  // every token is one 128-bit line
    if (j < s.size() && !is_sep((unsigned char)s[j])) return RX_EFORMAT;
    for (int lane = 0; lane < 4; lane++) {
      uint32_t w = 0;
      for (int d = 0; d < 8; d++) w = (w << 4) | (uint32_t)hexv((unsigned char)s[i + (size_t)lane * 8 + d]);
      words->push_back(w);
    }
    i = j;
  }
  return words->empty() ? RX_EFORMAT : RX_OK;
}

int rxh_parse_mem_text(const char* text, size_t len, std::vector<uint8_t>* bytes) {
  bytes->clear();
  bytes->reserve(len / 3 + 1);
  size_t i = 0;
  while (i < len) {
    while (i < len && (text[i] == ' ' || text[i] == '\t' || text[i] == '\n' || text[i] == '\r')) i++;
    if (i >= len) break;
    unsigned v = 0;
    size_t nd = 0;
    while (i < len && hexv((unsigned char)text[i]) >= 0) { v = v * 16 + (unsigned)hexv((unsigned char)text[i]); i++; nd++; }
    if (nd == 0 || nd > 2) return RX_EFORMAT;  // 8-bit memory: at most two hex digits per entry
    if (i < len && !(text[i] == ' ' || text[i] == '\t' || text[i] == '\n' || text[i] == '\r')) return RX_EFORMAT;
    bytes->push_back((uint8_t)v);
  }
  return RX_OK;
}

// size is not stored in the file (the reference passes it as a parameter,
// testbench_BLK_Mem.sv:20).  Unique s with: W[0]=0, W[0..s] non-decreasing, W[s]+s+1 words used,
// 0-3 zero pad words, every target < s.
int rxh_infer_size(const uint32_t* W, size_t nwords, uint32_t* size) {
  if (!W || nwords < 2 || W[0] != 0) return RX_ENFA;
  uint32_t hit = 0;
  int nhit = 0;
  for (size_t s = 1; s < nwords && s <= 0xFFFFFFu; s++) {
    if (W[s] < W[s - 1]) break;
    const uint64_t used = (uint64_t)W[s] + s + 1;
    if (used > nwords || nwords - used > 3) continue;
    if (rxh_validate(W, nwords, (uint32_t)s) == RX_OK) { hit = (uint32_t)s; nhit++; }
  }
  if (nhit != 1) return RX_ENFA;
  *size = hit;
  return RX_OK;
}

int rxh_validate(const uint32_t* W, size_t nwords, uint32_t size) {
  if (!W || size == 0 || size > 0xFFFFFFu || (size_t)size + 1 > nwords) return RX_ENFA;
  if (W[0] != 0) return RX_ENFA;
  for (uint32_t i = 0; i < size; i++)
    if (W[i + 1] < W[i]) return RX_ENFA;
  const uint64_t nnz = W[size];
  const uint64_t used = nnz + size + 1;
  if (used > nwords || nwords - used > 3) return RX_ENFA;
  for (uint64_t j = used; j < nwords; j++)
    if (W[j] != 0) return RX_ENFA;
  const uint32_t* col = W + size + 1;
  for (uint64_t e = 0; e < nnz; e++)
    if ((col[e] & 0xFFFFFFu) >= size) return RX_ENFA;  // next[] is size bits wide (FPGA.v:54)
  return RX_OK;
}

int rxh_build(const uint32_t* W, size_t nwords, uint32_t size_or_0, RxHostNfa* out) {
  uint32_t size = size_or_0;
  if (size == 0) {
    int rc = rxh_infer_size(W, nwords, &size);
    if (rc) return rc;
  } else {
    int rc = rxh_validate(W, nwords, size);
    if (rc) return rc;
  }
  if (size > RX_MAX_STATES) return RX_ECAPACITY;  // before the size*256-word index is allocated
  out->words.assign(W, W + nwords);
  out->size = size;
  out->nnz = W[size];
  const uint32_t* rp = out->row_ptr();
  const uint32_t* col = out->col();
  out->n_accept = 0;
  out->max_degree = 0;
  out->accept_bits.assign(((size_t)size + 31) / 32, 0u);
  for (uint32_t i = 0; i < size; i++) {
    const uint32_t deg = rp[i + 1] - rp[i];
    out->max_degree = std::max(out->max_degree, deg);
    if (deg == 0) {  // accept <=> empty row (FPGA.v:210)
      out->n_accept++;
      out->accept_bits[i >> 5] |= 1u << (i & 31);
    }
  }
  auto is_acc = [&](uint32_t t) { return (out->accept_bits[t >> 5] >> (t & 31)) & 1u; };

  // ---- pinned state: self-loop on every byte, fed by state 0 on the most bytes -----------------
  out->pin_state = 0xFFFFFFFFu;
  {
    std::vector<uint32_t> fed(size, 0);
    for (uint32_t j = rp[0]; j < rp[1]; j++) fed[col[j] & 0xFFFFFFu]++;
    uint32_t best = 0;
    for (uint32_t i = 1; i < size; i++) {
      if (fed[i] <= best) continue;
      bool seen[256] = {false};
      uint32_t nself = 0;
      for (uint32_t j = rp[i]; j < rp[i + 1]; j++)
        if ((col[j] & 0xFFFFFFu) == i && !seen[col[j] >> 24]) { seen[col[j] >> 24] = true; nself++; }
      if (nself == 256) { best = fed[i]; out->pin_state = i; }
    }
  }
  const uint32_t pin = out->pin_state;
  auto pin_flag = [&](uint32_t t) { return t == pin ? RXE_PIN : 0u; };

  // ---- slice index: for every (state, symbol) the SET of targets its row yields -------------
  out->symidx.assign((size_t)size * 256, 0u);
  out->ovf.assign(1, 0u);
  std::vector<uint32_t> bucket[256];
  std::map<std::vector<uint32_t>, uint32_t> ovf_at;
  for (uint32_t i = 0; i < size; i++) {
    const uint32_t base = rp[i], deg = rp[i + 1] - base;
    if (deg == 0) continue;
    for (auto& b : bucket) b.clear();
    for (uint32_t j = 0; j < deg; j++) {
      const uint32_t w = col[base + j];
      bucket[w >> 24].push_back(w & 0xFFFFFFu);
    }
    for (int c = 0; c < 256; c++) {
      auto& b = bucket[c];
      if (b.empty()) continue;
      std::sort(b.begin(), b.end());
      b.erase(std::unique(b.begin(), b.end()), b.end());  // next[t] <= 1 is idempotent
      uint32_t ent = 0;
      auto self = std::find(b.begin(), b.end(), i);
      if (self != b.end()) { ent |= RXE_SELF; b.erase(self); }
      if (b.size() == 1) {
        ent |= RXE_INLINE | b[0] | (is_acc(b[0]) ? RXE_ACCEPT : 0u) | pin_flag(b[0]);
      } else if (b.size() >= 2) {
        auto it = ovf_at.find(b);  // identical target sets share one list, so equal slices are equal words
        if (it == ovf_at.end()) {
          const size_t off = out->ovf.size();
          if (off + b.size() + 1 > RXE_TGT_MASK) return RX_ECAPACITY;
          out->ovf.push_back((uint32_t)b.size());
          for (uint32_t t : b) out->ovf.push_back(t | (is_acc(t) ? RXE_ACCEPT : 0u) | pin_flag(t));
          it = ovf_at.emplace(b, (uint32_t)off).first;
        }
        ent |= RXE_OVF | it->second;
      }
      out->symidx[(size_t)i * 256 + c] = ent;
    }
  }
  // ---- byte classes: bytes whose column of the slice index is identical behave identically in every
  // state, so the index can be stored per class (snort_16: 74 classes, l7: 164) ------------------------
  {
    std::map<std::vector<uint32_t>, uint32_t> class_of_col;
    std::vector<uint32_t> colv(size);
    std::vector<int> rep;  // representative byte of each class
    for (int c = 0; c < 256; c++) {
      for (uint32_t i = 0; i < size; i++) colv[i] = out->symidx[(size_t)i * 256 + c];
      auto it = class_of_col.find(colv);
      if (it == class_of_col.end()) {
        it = class_of_col.emplace(colv, (uint32_t)rep.size()).first;
        rep.push_back(c);
      }
      out->byte_class[c] = (uint8_t)it->second;
    }
    out->n_classes = (uint32_t)rep.size();
    out->symidx_c.assign((size_t)size * out->n_classes, 0u);
    for (uint32_t i = 0; i < size; i++)
      for (uint32_t k = 0; k < out->n_classes; k++)
        out->symidx_c[(size_t)i * out->n_classes + k] = out->symidx[(size_t)i * 256 + rep[k]];
  }
  // ---- look-ahead pruning of multi-target rows (pack kernel) -------------------------------------------------
  // A row with several targets on one byte (rule sets: every pattern that starts with that byte) activates states
  // most of which die on the very next byte.  A state that is not an accept state and has no edge on the next byte
  // cannot produce a pulse or a successor, so the pack kernel may skip inserting it — provided the sets it reports
  // (final sets, hand-off rows after the stream's last byte) are built from the full lists.  Per primary list and
  // per class of the NEXT byte the surviving targets are precomputed here as ordinary (interned) overflow lists.
  out->symidx_p.clear();
  out->ovf_dir.clear();
  {
    const uint32_t ncls = out->n_classes;
    std::map<uint32_t, uint32_t> list_no;  // overflow offset -> list number
    for (uint32_t w : out->symidx_c)
      if (w & RXE_OVF) list_no.emplace(w & RXE_TGT_MASK, 0u);
    uint32_t n = 0;
    for (auto& kv : list_no) kv.second = n++;
    bool ok = n != 0 && (uint64_t)n * (ncls + 1u) <= (16u << 20);
    if (ok) {
      auto dir_word = [](uint32_t off, uint32_t cnt) { return (off << 8) | (cnt < 255u ? cnt : 255u); };
      out->ovf_dir.assign((size_t)n * (ncls + 1u), 0u);
      std::vector<uint32_t> sub;
      for (auto& kv : list_no) {
        const uint32_t off = kv.first, cnt = out->ovf[off];
        uint32_t* dir = &out->ovf_dir[(size_t)kv.second * (ncls + 1u)];
        dir[ncls] = dir_word(off, cnt);
        for (uint32_t k = 0; k < ncls && ok; k++) {
          sub.clear();
          for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t t = out->ovf[off + 1u + j];
            if ((t & RXE_ACCEPT) || out->symidx_c[(size_t)(t & RXE_TGT_MASK) * ncls + k] != 0u) sub.push_back(t & RXE_TGT_MASK);
          }
          if (sub.size() == cnt) { dir[k] = dir[ncls]; continue; }
          if (sub.empty()) { dir[k] = 0u; continue; }
          auto it = ovf_at.find(sub);  // sub-lists are ordinary lists: [count][targets with their flags]
          if (it == ovf_at.end()) {
            const size_t o2 = out->ovf.size();
            if (o2 + sub.size() + 1 > RXE_TGT_MASK) { ok = false; break; }
            out->ovf.push_back((uint32_t)sub.size());
            for (uint32_t t : sub) out->ovf.push_back(t | (is_acc(t) ? RXE_ACCEPT : 0u) | pin_flag(t));
            it = ovf_at.emplace(sub, (uint32_t)o2).first;
          }
          dir[k] = dir_word(it->second, (uint32_t)sub.size());
        }
        if (!ok) break;
      }
    }
    if (!ok) out->ovf_dir.clear();
    // The pruned index exists when the directory does, or when there is nothing to put in a directory.  "Narrow" automata
    // (state ids and list numbers fit 16 bits) also get the INLINE targets' next-class bits: bit (n & 7) of bits 23:16 is
    // set iff the target has an edge on some class n' with n' & 7 == n & 7 (conservative: a set bit only means "may
    // live"), all ones for an accept state — what look-ahead pruning does for multi-target rows, for single targets.
    out->prune_narrow = false;
    if (ok || n == 0) {
      out->symidx_p = out->symidx_c;
      for (uint32_t& w : out->symidx_p)
        if (w & RXE_OVF) w = (w & ~RXE_TGT_MASK) | list_no[w & RXE_TGT_MASK];
      if (size <= 65536u && n <= 65536u) {
        out->prune_narrow = true;
        std::vector<uint8_t> live8(size, 0);
        for (uint32_t t = 0; t < size; t++) {
          if (is_acc(t)) { live8[t] = 0xFF; continue; }
          for (uint32_t k = 0; k < ncls; k++)
            if (out->symidx_c[(size_t)t * ncls + k] != 0u) live8[t] |= (uint8_t)(1u << (k & 7u));
        }
        for (uint32_t& w : out->symidx_p)
          if (w & RXE_INLINE) w |= (uint32_t)live8[w & 0xFFFFu] << 16;
      }
    }
  }
  // ---- always-on-state folding (pack kernel FOLD builds) ----------------------------------------------------------
  // The pinned state loops on every byte, so once a stream holds it it holds it forever; when state 0 enters it on
  // EVERY byte, every stream that starts from reset (FPGA.v:134-147) holds it from pass 1 on.  Such a state need not be
  // a list entry: what its row emits on a byte is a function of the byte alone — and, with one byte of look-ahead, only
  // the targets that are accept states or have an edge on the NEXT byte need to be inserted (the others can neither
  // pulse nor produce a successor; the sets that are REPORTED are built from the unpruned column).  snort_16: state 1
  // starts 34 patterns; on the shipped traces 2.4 list entries per stream-byte become 1.1, on uniform bytes 1.1 -> 0.01.
  out->pin_tab.clear();
  if (pin != 0xFFFFFFFFu) {
    const uint32_t ncls = out->n_classes;
    bool all = true;  // state 0 --every byte--> pin
    for (uint32_t k = 0; k < ncls && all; k++) {
      const uint32_t w = out->symidx_c[k];  // row of state 0
      bool has = (w & RXE_INLINE) && (w & RXE_PIN);
      if (!has && (w & RXE_OVF)) {
        const uint32_t off = w & RXE_TGT_MASK;
        for (uint32_t j = 0; j < out->ovf[off] && !has; j++) has = (out->ovf[off + 1u + j] & RXE_PIN) != 0;
      }
      all = has;
    }
    bool ok = all && pin != 0 && (uint64_t)ncls * (ncls + 1u) * 4u <= 48u * 1024u;  // the table lives in LDS
    std::vector<uint32_t> tab((size_t)ncls * (ncls + 1u), 0u), tg, sub;
    for (uint32_t k = 0; k < ncls && ok; k++) {
      const uint32_t w = out->symidx_c[(size_t)pin * ncls + k];  // pin's slice on class k (RXE_SELF set by definition)
      tg.clear();
      if (w & RXE_INLINE) tg.push_back(w & RXE_TGT_MASK);
      if (w & RXE_OVF) {
        const uint32_t off = w & RXE_TGT_MASK;
        for (uint32_t j = 0; j < out->ovf[off]; j++) tg.push_back(out->ovf[off + 1u + j] & RXE_TGT_MASK);
      }
      for (uint32_t n = 0; n <= ncls && ok; n++) {
        sub.clear();
        for (uint32_t t : tg)
          if (n == ncls || is_acc(t) || out->symidx_c[(size_t)t * ncls + n] != 0u) sub.push_back(t);
        uint32_t ent = 0;
        if (sub.size() == 1) {
          ent = RXE_INLINE | sub[0] | (is_acc(sub[0]) ? RXE_ACCEPT : 0u);
        } else if (sub.size() >= 2) {
          auto it = ovf_at.find(sub);
          if (it == ovf_at.end()) {
            const size_t o2 = out->ovf.size();
            if (o2 + sub.size() + 1 > RXE_TGT_MASK) { ok = false; break; }
            out->ovf.push_back((uint32_t)sub.size());
            for (uint32_t t : sub) out->ovf.push_back(t | (is_acc(t) ? RXE_ACCEPT : 0u) | pin_flag(t));
            it = ovf_at.emplace(sub, (uint32_t)o2).first;
          }
          ent = RXE_OVF | it->second;
        }
        tab[(size_t)k * (ncls + 1u) + n] = ent;
      }
    }
    if (ok) out->pin_tab.swap(tab);
  }
  // ---- RXE_MAYDUP: which insertions can meet a duplicate -----------------------------------------------------------
  // For every (target, class): the number of distinct states with an edge to the target on that class (the target
  // itself counts when it loops on the class).  States nothing leads to (state 0) are active in pass 0 only, alone, and
  // do not count.  Fewer than two => a duplicate-free set S_k cannot produce the target twice.
  {
    const uint32_t ncls = out->n_classes;
    const size_t cells = (size_t)size * ncls;
    std::vector<uint8_t> np;
    const bool exact = cells <= ((size_t)256 << 20);
    if (exact) {
      np.assign(cells, 0);
      std::vector<uint8_t> entered(size, 0);
      for (uint32_t e = 0; e < out->nnz; e++) entered[col[e] & 0xFFFFFFu] = 1;
      auto bump = [&](uint32_t t, uint32_t k) { uint8_t& c = np[(size_t)t * ncls + k]; if (c < 2) c++; };
      for (uint32_t i = 0; i < size; i++) {
        if (!entered[i]) continue;
        for (uint32_t k = 0; k < ncls; k++) {
          const uint32_t w = out->symidx_c[(size_t)i * ncls + k];
          if (w & RXE_SELF) bump(i, k);
          if (w & RXE_INLINE) bump(w & RXE_TGT_MASK, k);
          if (w & RXE_OVF) {
            const uint32_t off = w & RXE_TGT_MASK;
            for (uint32_t j = 0; j < out->ovf[off]; j++) bump(out->ovf[off + 1u + j] & RXE_TGT_MASK, k);
          }
        }
      }
    }
    auto dup = [&](uint32_t t, uint32_t k) { return !exact || np[(size_t)t * ncls + k] >= 2; };
    auto flag_list = [&](uint32_t off, uint32_t k) {  // lists are shared between slices: flags accumulate (conservative)
      for (uint32_t j = 0; j < out->ovf[off]; j++)
        if (dup(out->ovf[off + 1u + j] & RXE_TGT_MASK, k)) out->ovf[off + 1u + j] |= RXE_MAYDUP;
    };
    const bool have_dir = !out->ovf_dir.empty(), have_p = !out->symidx_p.empty();
    for (uint32_t i = 0; i < size; i++)
      for (uint32_t k = 0; k < ncls; k++) {
        const size_t at = (size_t)i * ncls + k;
        const uint32_t w = out->symidx_c[at];
        if ((w & RXE_INLINE) && dup(w & RXE_TGT_MASK, k)) {
          out->symidx_c[at] |= RXE_MAYDUP;
          if (have_p) out->symidx_p[at] |= RXE_MAYDUP;
        }
        if (w & RXE_OVF) {
          flag_list(w & RXE_TGT_MASK, k);
          if (have_dir) {
            const uint32_t* dir = &out->ovf_dir[(size_t)(out->symidx_p[at] & RXE_TGT_MASK) * (ncls + 1u)];
            for (uint32_t n = 0; n <= ncls; n++)
              if (dir[n]) flag_list(dir[n] >> 8, k);
          }
        }
      }
    // one all-zero row behind the per-class index: the id `size` is the register kernel's "free lane" (rx_kernels.hip)
    out->symidx_c.resize((size_t)(size + 1u) * ncls, 0u);
    if (have_p) out->symidx_p.resize((size_t)(size + 1u) * ncls, 0u);
    if (!out->pin_tab.empty())
      for (uint32_t k = 0; k < ncls; k++)
        for (uint32_t n = 0; n <= ncls; n++) {
          uint32_t& w = out->pin_tab[(size_t)k * (ncls + 1u) + n];
          if ((w & RXE_INLINE) && dup(w & RXE_TGT_MASK, k)) w |= RXE_MAYDUP;
          if (w & RXE_OVF) flag_list(w & RXE_TGT_MASK, k);
        }
    // ---- the register kernel's index: the in-place update of a lane precomputed per (state, class) ------------------
    out->regidx.clear();
    out->reg_tmask = RXE_TGT_MASK;
    if ((uint64_t)(size + 1u) * ncls * 8u <= ((uint64_t)256 << 20)) {
      const bool fold = !out->pin_tab.empty();
      const bool narrow = size < 65536u;
      std::vector<uint8_t> live8;  // which byte classes (mod 8) a state has an edge on; accept states: all
      if (narrow) {
        out->reg_tmask = 0xFFFFu;
        live8.assign(size, 0);
        for (uint32_t t = 0; t < size; t++) {
          if (is_acc(t)) { live8[t] = 0xFF; continue; }
          for (uint32_t k = 0; k < ncls; k++)
            if (out->symidx_c[(size_t)t * ncls + k] != 0u) live8[t] |= (uint8_t)(1u << (k & 7u));
        }
      }
      out->regidx.assign((size_t)(size + 1u) * ncls * 2u, 0u);
      for (uint32_t i = 0; i <= size; i++)
        for (uint32_t k = 0; k < ncls; k++) {
          const uint32_t w = out->symidx_c[(size_t)i * ncls + k];  // (row `size` is all zero)
          const bool surv = (w & RXE_SELF) != 0u;
          const bool inl = (w & RXE_INLINE) != 0u && !(fold && (w & RXE_PIN));
          const bool own = inl && !surv && !(w & RXE_MAYDUP);  // moves on to a target nothing else can reach: in place
          uint32_t fast = surv ? i : (own ? (w & RXE_TGT_MASK) : size);
          if (own && (w & RXE_ACCEPT)) fast |= RXR_ACC;
          if ((inl && !own) || (w & RXE_OVF)) fast |= RXR_NEED;
          if (inl && !own) fast |= (w & RXE_MAYDUP) ? RXR_DUPC : RXR_EXTRA;  // (!own and not MAYDUP means the state survives)
          if (w & RXE_OVF) fast |= RXR_OVFL;
          if (narrow && (fast & RXR_NEED)) fast |= (uint32_t)((w & RXE_OVF) ? 0xFFu : live8[w & RXE_TGT_MASK]) << 16;
          out->regidx[((size_t)i * ncls + k) * 2u] = fast;
          out->regidx[((size_t)i * ncls + k) * 2u + 1u] = w;
        }
    }
  }
  return RX_OK;
}
