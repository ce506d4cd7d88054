// rx_compile.cpp — regex list -> epsilon-free NFA -> CSR word table in the reference's conventions.
//
// The reference ships two pre-compiled tables and no compiler, so nothing here can be
// checked against the reference: parity for this step is UNPINNED; tests compare against Python's `re`.
// The table conventions are the ones the shipped tables follow so the result runs on
// the FPGA design and on every kernel here unchanged:
//   * state 0 = reset state (Design/FPGA.v:146), never re-entered;
//   * unanchored matching is encoded in the automaton: state 0 --every byte--> state 1, state 1 loops on
//     every byte (the `.*` state), and both feed the first positions of every pattern;
//   * accept <=> empty row (FPGA.v:210-226): every pattern ends in sink states with no out-edges;
//   * edge word = symbol<<24 | target, rows state-major, row_ptr first (FPGA.v:773,793,888-898).
// Construction: Glushkov position automaton (no epsilon moves, one state per symbol occurrence).
//
// Supported syntax: literals, escapes (\n \r \t \f \v \0 \xHH \d \D \w \W \s \S and escaped punctuation),
// `.`, classes [a-z] [^...], groups (...) and (?:...), alternation |, quantifiers * + ? {m} {m,} {m,n},
// a leading ^ (anchor to stream start).  A pattern may be written /regex/flags with flags i (ignore
// case) and s (dot matches \n).  Unsupported: $ and other look-around/back-references (RX_EFORMAT).
#include <algorithm>
#include <array>
#include <bitset>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "rx_internal.hpp"

namespace {

using ByteSet = std::bitset<256>;

struct Node {
  enum Kind { CHAR, CAT, ALT, STAR, PLUS, OPT, EMPTY } kind = EMPTY;
  ByteSet cls;
  uint32_t depth = 1;  // height of the subtree: parse, clone, walk and the destructor recurse this deep
  std::unique_ptr<Node> a, b;
};
// Recursion bound for everything that walks a tree (a few hundred bytes of stack per level).  Sequences and
// alternations are built as balanced trees, so only real nesting — parentheses, {m,n} tails — counts.
constexpr uint32_t RX_MAX_TREE_DEPTH = 2500;
constexpr uint32_t RX_MAX_GROUP_NESTING = 250;
using NodeP = std::unique_ptr<Node>;

NodeP mk(Node::Kind k, NodeP a = nullptr, NodeP b = nullptr) {
  NodeP n(new Node());
  n->kind = k;
  n->a = std::move(a);
  n->b = std::move(b);
  n->depth = 1u + std::max(n->a ? n->a->depth : 0u, n->b ? n->b->depth : 0u);
  return n;
}
NodeP clone(const Node* n) {
  if (!n) return nullptr;
  NodeP c(new Node());
  c->kind = n->kind;
  c->cls = n->cls;
  c->depth = n->depth;
  c->a = clone(n->a.get());
  c->b = clone(n->b.get());
  return c;
}

struct Parser {
  const std::string& s;
  size_t i = 0;
  bool icase, dotall;
  std::string err;
  size_t budget = 200000;  // leaf budget (bounded repetition expands by copying)
  uint32_t nesting = 0;    // open groups at the cursor
  Parser(const std::string& src, bool ic, bool da) : s(src), icase(ic), dotall(da) {}

  bool fail(const std::string& m) {
    if (err.empty()) err = m + " at offset " + std::to_string(i);
    return false;
  }
  bool eof() const { return i >= s.size(); }

  ByteSet fold(ByteSet c) const {
    if (!icase) return c;
    for (int ch = 'a'; ch <= 'z'; ch++) {
      if (c[ch]) c[ch - 32] = true;
      if (c[ch - 32]) c[ch] = true;
    }
    return c;
  }
  static ByteSet range(int lo, int hi) {
    ByteSet b;
    for (int c = lo; c <= hi; c++) b[c] = true;
    return b;
  }
  static ByteSet named(char k) {
    ByteSet b;
    switch (k) {
      case 'd': b = range('0', '9'); break;
      case 'w': b = range('0', '9') | range('a', 'z') | range('A', 'Z'); b['_'] = true; break;
      case 's': for (char c : {' ', '\t', '\n', '\r', '\f', '\v'}) b[(unsigned char)c] = true; break;
    }
    return b;
  }
  // escape after the backslash; returns false on error.  is_set: a class such as \d was read
  bool escape(ByteSet* out, int* single) {
    if (eof()) return fail("dangling backslash");
    const char c = s[i++];
    *single = -1;
    switch (c) {
      case 'n': *single = '\n'; break;
      case 'r': *single = '\r'; break;
      case 't': *single = '\t'; break;
      case 'f': *single = '\f'; break;
      case 'v': *single = '\v'; break;
      case '0': *single = 0; break;
      case 'x': {
        int v = 0, nd = 0;
        while (nd < 2 && !eof() && isxdigit((unsigned char)s[i])) {
          const char h = s[i++];
          v = v * 16 + (h <= '9' ? h - '0' : (h | 32) - 'a' + 10);
          nd++;
        }
        if (nd == 0) return fail("\\x needs hex digits");
        *single = v;
        break;
      }
      case 'd': case 'w': case 's': *out = named(c); return true;
      case 'D': case 'W': case 'S': *out = ~named((char)(c | 32)); return true;
      default:
        if (isalnum((unsigned char)c)) return fail(std::string("unsupported escape \\") + c);
        *single = (unsigned char)c;
    }
    out->reset();
    (*out)[*single] = true;
    return true;
  }
  bool char_class(ByteSet* out) {  // after '['
    ByteSet b;
    bool neg = false;
    if (!eof() && s[i] == '^') { neg = true; i++; }
    bool first = true;
    for (;;) {
      if (eof()) return fail("unterminated [");
      char c = s[i++];
      if (c == ']' && !first) break;
      first = false;
      ByteSet item;
      int lo = -1;
      if (c == '\\') {
        if (!escape(&item, &lo)) return false;
      } else {
        lo = (unsigned char)c;
        item[lo] = true;
      }
      if (lo >= 0 && i + 1 < s.size() && s[i] == '-' && s[i + 1] != ']') {
        i++;
        int hi;
        char d = s[i++];
        if (d == '\\') {
          ByteSet tmp;
          if (!escape(&tmp, &hi)) return false;
          if (hi < 0) return fail("bad range end");
        } else {
          hi = (unsigned char)d;
        }
        if (hi < lo) return fail("reversed range");
        item = range(lo, hi);
      }
      b |= item;
    }
    b = fold(b);
    *out = neg ? ~b : b;
    return true;
  }

  // kind-tree over parts[lo, hi) of logarithmic height (a left-deep chain of an 80 000-byte literal would recurse
  // 80 000 levels in clone / walk / ~Node)
  NodeP balanced(Node::Kind kind, std::vector<NodeP>& parts, size_t lo, size_t hi) {
    if (hi - lo == 1) return std::move(parts[lo]);
    const size_t mid = lo + (hi - lo) / 2;
    NodeP l = balanced(kind, parts, lo, mid), r = balanced(kind, parts, mid, hi);
    return mk(kind, std::move(l), std::move(r));
  }
  NodeP checked(NodeP n) {
    if (n && n->depth > RX_MAX_TREE_DEPTH) { fail("pattern nests too deeply"); return nullptr; }
    return n;
  }

  NodeP leaf(const ByteSet& c) {
    if (budget == 0) { fail("pattern expands to too many positions"); return nullptr; }
    budget--;
    NodeP n = mk(Node::CHAR);
    n->cls = c;
    return n;
  }
  NodeP atom() {
    if (eof()) { fail("unexpected end"); return nullptr; }
    const char c = s[i];
    if (c == '(') {
      i++;
      if (i + 1 < s.size() && s[i] == '?') {
        if (s[i + 1] == ':') i += 2;
        else { fail("unsupported group (?"); return nullptr; }
      }
      if (++nesting > RX_MAX_GROUP_NESTING) { fail("groups nest too deeply"); return nullptr; }
      NodeP e = alt();
      nesting--;
      if (!e) return nullptr;
      if (eof() || s[i] != ')') { fail("missing )"); return nullptr; }
      i++;
      return e;
    }
    if (c == '[') {
      i++;
      ByteSet b;
      if (!char_class(&b)) return nullptr;
      return leaf(b);
    }
    if (c == '.') {
      i++;
      ByteSet b;
      b.set();
      if (!dotall) b['\n'] = false;
      return leaf(b);
    }
    if (c == '\\') {
      i++;
      ByteSet b;
      int single;
      if (!escape(&b, &single)) return nullptr;
      return leaf(single >= 0 ? fold(b) : b);
    }
    if (c == '$' || c == '^') { fail("anchors are only supported as a leading ^"); return nullptr; }
    if (c == '*' || c == '+' || c == '?' || c == '{' || c == ')' || c == '|') { fail("nothing to repeat"); return nullptr; }
    i++;
    ByteSet b;
    b[(unsigned char)c] = true;
    return leaf(fold(b));
  }
  NodeP repeat(NodeP a, int lo, int hi) {  // hi < 0: unbounded
    std::vector<NodeP> parts;
    auto cat = [&](NodeP x) { parts.push_back(std::move(x)); };
    for (int k = 0; k < lo; k++) {
      NodeP c = clone(a.get());
      cat(std::move(c));
    }
    if (hi < 0) {
      cat(mk(Node::STAR, clone(a.get())));
    } else {
      NodeP tail;  // (a (a (a)?)?)?
      for (int k = lo; k < hi; k++) {
        NodeP inner = clone(a.get());
        if (tail) inner = mk(Node::CAT, std::move(inner), std::move(tail));
        tail = mk(Node::OPT, std::move(inner));
      }
      if (tail) cat(std::move(tail));
    }
    if (parts.empty()) return mk(Node::EMPTY);
    return checked(balanced(Node::CAT, parts, 0, parts.size()));
  }
  size_t count_leaves(const Node* n) {
    if (!n) return 0;
    return (n->kind == Node::CHAR ? 1 : 0) + count_leaves(n->a.get()) + count_leaves(n->b.get());
  }
  NodeP piece() {
    NodeP a = atom();
    if (!a) return nullptr;
    while (!eof()) {
      const char c = s[i];
      if (c == '*') { i++; a = checked(mk(Node::STAR, std::move(a))); }
      else if (c == '+') { i++; a = checked(mk(Node::PLUS, std::move(a))); }
      else if (c == '?') { i++; a = checked(mk(Node::OPT, std::move(a))); }
      else if (c == '{') {
        size_t j = i + 1;
        int lo = 0, hi = -2, nd = 0;
        while (j < s.size() && isdigit((unsigned char)s[j])) { lo = lo * 10 + (s[j++] - '0'); nd++; if (lo > 1000) break; }
        if (nd == 0 || j >= s.size()) { fail("bad {m,n}"); return nullptr; }
        if (s[j] == '}') hi = lo;
        else if (s[j] == ',') {
          j++;
          if (j < s.size() && s[j] == '}') hi = -1;
          else {
            hi = 0; nd = 0;
            while (j < s.size() && isdigit((unsigned char)s[j])) { hi = hi * 10 + (s[j++] - '0'); nd++; if (hi > 1000) break; }
            if (nd == 0) { fail("bad {m,n}"); return nullptr; }
          }
        }
        if (j >= s.size() || s[j] != '}' || hi == -2 || lo > 1000 || hi > 1000 || (hi >= 0 && hi < lo)) { fail("bad {m,n}"); return nullptr; }
        i = j + 1;
        const size_t copies = (size_t)(hi < 0 ? lo + 1 : hi);
        const size_t leaves = count_leaves(a.get());
        if (leaves * copies > budget) { fail("pattern expands to too many positions"); return nullptr; }
        budget -= leaves * (copies ? copies - 1 : 0) > budget ? budget : leaves * (copies ? copies - 1 : 0);
        a = repeat(std::move(a), lo, hi);
      } else break;
      if (!a) return nullptr;
      if (!eof() && (s[i] == '?' || s[i] == '+') && (c == '*' || c == '+' || c == '?' || c == '{')) i++;  // lazy/possessive: same language
    }
    return a;
  }
  NodeP seq() {
    std::vector<NodeP> parts;
    while (!eof() && s[i] != '|' && s[i] != ')') {
      NodeP p = piece();
      if (!p) return nullptr;
      parts.push_back(std::move(p));
    }
    if (parts.empty()) return mk(Node::EMPTY);
    return checked(balanced(Node::CAT, parts, 0, parts.size()));
  }
  NodeP alt() {
    std::vector<NodeP> parts;
    NodeP a = seq();
    if (!a) return nullptr;
    parts.push_back(std::move(a));
    while (!eof() && s[i] == '|') {
      i++;
      NodeP b = seq();
      if (!b) return nullptr;
      parts.push_back(std::move(b));
    }
    return checked(balanced(Node::ALT, parts, 0, parts.size()));
  }
};

// Glushkov sets over positions numbered in `cls`
struct Glushkov {
  std::vector<ByteSet> cls;
  std::vector<std::vector<uint32_t>> follow;
  struct Info { bool nullable = false; std::vector<uint32_t> first, last; };

  static void uni(std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    a.insert(a.end(), b.begin(), b.end());
    std::sort(a.begin(), a.end());
    a.erase(std::unique(a.begin(), a.end()), a.end());
  }
  void link(const std::vector<uint32_t>& from, const std::vector<uint32_t>& to) {
    for (uint32_t p : from) uni(follow[p], to);
  }
  Info walk(const Node* n) {
    Info r;
    switch (n->kind) {
      case Node::EMPTY: r.nullable = true; break;
      case Node::CHAR: {
        const uint32_t p = (uint32_t)cls.size();
        cls.push_back(n->cls);
        follow.emplace_back();
        r.first = {p};
        r.last = {p};
        break;
      }
      case Node::CAT: {
        Info a = walk(n->a.get()), b = walk(n->b.get());
        link(a.last, b.first);
        r.nullable = a.nullable && b.nullable;
        r.first = a.first;
        if (a.nullable) uni(r.first, b.first);
        r.last = b.last;
        if (b.nullable) uni(r.last, a.last);
        break;
      }
      case Node::ALT: {
        Info a = walk(n->a.get()), b = walk(n->b.get());
        r.nullable = a.nullable || b.nullable;
        r.first = a.first; uni(r.first, b.first);
        r.last = a.last; uni(r.last, b.last);
        break;
      }
      case Node::STAR: case Node::PLUS: case Node::OPT: {
        Info a = walk(n->a.get());
        if (n->kind != Node::OPT) link(a.last, a.first);
        r.nullable = n->kind == Node::PLUS ? a.nullable : true;
        r.first = a.first;
        r.last = a.last;
        break;
      }
    }
    return r;
  }
};

}  // namespace

// words: CSR table; accept_pattern[state] = pattern index for accept states, -1 otherwise
int rxc_compile(const char* const* patterns, size_t n, uint32_t flags, std::vector<uint32_t>* words,
                std::vector<int32_t>* accept_pattern, std::string* err) {
  struct Edge { uint32_t src; uint8_t sym; uint32_t dst; };
  std::vector<Edge> edges;
  std::vector<int32_t> acc;  // per state
  auto new_state = [&](int32_t pat) { acc.push_back(pat); return (uint32_t)acc.size() - 1; };
  const uint32_t S0 = new_state(-1);
  const uint32_t S1 = new_state(-1);  // the `.*` state
  for (int c = 0; c < 256; c++) {
    edges.push_back({S0, (uint8_t)c, S1});
    edges.push_back({S1, (uint8_t)c, S1});
  }
  for (size_t pi = 0; pi < n; pi++) {
    if (!patterns[pi]) { *err = "null pattern"; return RX_EINVAL; }
    std::string src = patterns[pi];
    bool icase = (flags & RX_RE_ICASE) != 0, dotall = (flags & RX_RE_DOTALL) != 0;
    if (src.size() >= 2 && src[0] == '/') {  // /regex/flags — only if what follows the last '/' is all flag letters
      const size_t close = src.rfind('/');
      bool flags_ok = close > 0;
      for (size_t k = close + 1; flags_ok && k < src.size(); k++) flags_ok = src[k] == 'i' || src[k] == 's';
      if (flags_ok) {
        for (size_t k = close + 1; k < src.size(); k++) {
          if (src[k] == 'i') icase = true;
          if (src[k] == 's') dotall = true;
        }
        src = src.substr(1, close - 1);
      }
    }
    bool anchored = false;
    if (!src.empty() && src[0] == '^') { anchored = true; src.erase(0, 1); }
    Parser ps(src, icase, dotall);
    NodeP ast = ps.alt();
    if (ast && !ps.eof()) ps.fail("unbalanced )");
    if (!ast || !ps.err.empty()) { *err = "pattern " + std::to_string(pi) + ": " + ps.err; return RX_EFORMAT; }
    Glushkov g;
    Glushkov::Info top = g.walk(ast.get());
    if (top.nullable) { *err = "pattern " + std::to_string(pi) + " matches the empty string"; return RX_EFORMAT; }
    const size_t np = g.cls.size();
    if (acc.size() + np + 1 > 0xFFFFFEu) { *err = "too many states"; return RX_ECAPACITY; }
    std::vector<bool> is_last(np, false);
    for (uint32_t p : top.last) is_last[p] = true;
    // a last position without successors IS an accept sink (empty row); the others share one extra sink
    std::vector<uint32_t> st(np);
    bool need_sink = false;
    for (size_t p = 0; p < np; p++) {
      const bool sink = is_last[p] && g.follow[p].empty();
      st[p] = new_state(sink ? (int32_t)pi : -1);
      if (is_last[p] && !sink) need_sink = true;
    }
    const uint32_t sinkA = need_sink ? new_state((int32_t)pi) : 0;
    auto enter = [&](uint32_t from, uint32_t p) {  // every byte of class(p) moves `from` into position p
      for (int c = 0; c < 256; c++) {
        if (!g.cls[p][c]) continue;
        edges.push_back({from, (uint8_t)c, st[p]});
        if (is_last[p] && !g.follow[p].empty()) edges.push_back({from, (uint8_t)c, sinkA});
      }
    };
    for (uint32_t p : top.first) {
      enter(S0, p);
      if (!anchored) enter(S1, p);
    }
    for (size_t p = 0; p < np; p++)
      for (uint32_t q : g.follow[p]) enter(st[p], q);
  }
  // ---- rows: state-major, duplicates removed, row_ptr first, zero pad to a 128-bit line -------
  const uint32_t size = (uint32_t)acc.size();
  std::sort(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) {
    if (a.src != b.src) return a.src < b.src;
    if (a.sym != b.sym) return a.sym < b.sym;
    return a.dst < b.dst;
  });
  edges.erase(std::unique(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) {
                return a.src == b.src && a.sym == b.sym && a.dst == b.dst;
              }), edges.end());
  words->assign((size_t)size + 1, 0u);
  for (const Edge& e : edges) (*words)[e.src + 1]++;
  for (uint32_t i = 0; i < size; i++) (*words)[i + 1] += (*words)[i];
  for (const Edge& e : edges) words->push_back(((uint32_t)e.sym << 24) | e.dst);
  while (words->size() % 4) words->push_back(0u);
  *accept_pattern = acc;
  // sanity: accept <=> empty row
  for (uint32_t i = 0; i < size; i++) {
    const bool empty = (*words)[i + 1] == (*words)[i];
    if (empty != (acc[i] >= 0)) { *err = "internal: accept/empty-row mismatch at state " + std::to_string(i); return RX_ENFA; }
  }
  return RX_OK;
}

// Xilinx COE exactly as the reference's files are laid out: radix line, vector line(s), one 128-bit
// token (4 words, word 0 leftmost) per line.
int rxc_write_coe(const char* path, const std::vector<uint32_t>& words) {
  if (!path || words.empty() || words.size() % 4) return RX_EINVAL;
  FILE* f = fopen(path, "w");
  if (!f) return RX_EIO;
  fprintf(f, "memory_initialization_radix=16;\nmemory_initialization_vector=");
  for (size_t i = 0; i < words.size(); i += 4)
    fprintf(f, "%08x%08x%08x%08x%s", words[i], words[i + 1], words[i + 2], words[i + 3], i + 4 < words.size() ? "\n" : ";\n");
  const bool ok = !ferror(f);
  return (fclose(f) == 0 && ok) ? RX_OK : RX_EIO;
}
