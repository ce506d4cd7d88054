// CPU model of rx_sym_res_kernel (same tables, same order of steps, 64 "lanes" per wavefront) against a plain set-based walk of
// the slice index: finds algorithmic holes without a GPU.
// build: g++ -O2 -std=c++17 -Iregex-fpga_amd/csrc -Iinclude tools/res_model.cpp regex-fpga_amd/csrc/rx_host.cpp -o gpurun_out/tmp/res_model
// usage: res_model words.bin rows.bin n_streams stream_len S [full_mode]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>
#include "rx_internal.hpp"
typedef std::tuple<uint32_t, uint32_t, uint32_t> Ev;
static std::vector<uint8_t> slurp(const char* p) { std::string s; rxh_read_file(p, &s); return std::vector<uint8_t>(s.begin(), s.end()); }
int main(int argc, char** argv) {
  if (argc < 6) return 2;
  auto wb = slurp(argv[1]);
  std::vector<uint32_t> W((const uint32_t*)wb.data(), (const uint32_t*)wb.data() + wb.size() / 4);
  RxHostNfa h;
  if (rxh_build(W.data(), W.size(), 0, &h)) { printf("build failed\n"); return 1; }
  auto rows = slurp(argv[2]);
  const uint32_t ns = atoi(argv[3]), sl = atoi(argv[4]), S = atoi(argv[5]);
  const bool full = argc > 6 && atoi(argv[6]);
  const uint32_t n_consume = full ? sl : (sl ? sl - 1 : 0), n_passes = full ? sl + 1 : n_consume;  // tb-compat: passes 0..n-2
  const uint32_t ncls = h.n_classes, size = h.size, FREE = size, pin_cols = ncls + 1;
  printf("size %u classes %u pin %u res_dwords %u\n", size, ncls, h.pin_state, h.res_dwords);
  if (!h.res_dwords) return 1;
  auto is_acc = [&](uint32_t s) { return (h.accept_bits[s >> 5] >> (s & 31)) & 1u; };
  // consistency of the tables: every list offset is a list head, every target word carries its state's number
  {
    std::set<uint32_t> heads; size_t at = 1;
    while (at < h.ovf.size()) { heads.insert((uint32_t)at); at += (size_t)h.ovf[at] + 1; }
    if (at != h.ovf.size()) printf("ovf walk ends at %zu of %zu\n", at, h.ovf.size());
    std::map<uint32_t, uint32_t> num;
    auto chk = [&](uint32_t w, const char* what) { uint32_t t = w & 0xFFFF, d = (w >> 16) & 0x3FF; if (num.count(t) && num[t] != d) printf("number mismatch %s state %u: %u vs %u\n", what, t, d, num[t]); num[t] = d; };
    for (uint32_t i = 0; i < size; i++) for (uint32_t k = 0; k < ncls; k++) { uint32_t w = h.res_idx[((size_t)i * ncls + k) * 2 + 1]; if (w & RXE_INLINE) chk(w, "idx"); if ((w & RXE_OVF) && !heads.count(w & RXE_TGT_MASK)) printf("bad list offset\n"); }
    for (uint32_t w : h.res_pin) { if (w & RXE_INLINE) chk(w, "pin"); if ((w & RXE_OVF) && !heads.count(w & RXE_TGT_MASK)) printf("bad pin list offset\n"); }
    for (uint32_t hd : heads) for (uint32_t j = 0; j < h.res_ovf[hd]; j++) chk(h.res_ovf[hd + 1 + j], "ovf");
  }
  // exact walk
  std::multiset<Ev> ref, got;
  auto exact_from = [&](uint32_t s, std::set<uint32_t> cur, uint32_t k0, bool skip_first, std::multiset<Ev>& out, std::set<uint32_t>* fin) {
    for (uint32_t k = k0; k < n_passes; k++) {
      if (!(skip_first && k == k0)) for (uint32_t st : cur) if (is_acc(st)) out.insert(Ev(s, k, st));
      if (k >= n_consume) break;
      const uint32_t c = h.byte_class[rows[(size_t)s * sl + k]];
      if (getenv("RES_MODEL_TRACE") && (int)s == atoi(getenv("RES_MODEL_TRACE")) && k <= 4 && k0 == 0) {
        const uint32_t cn = k + 1 < sl ? h.byte_class[rows[(size_t)s * sl + k + 1]] : 0;
        printf("[exact] stream %u pass %u class %u next %u:", s, k, c, cn);
        for (uint32_t st : cur) printf("  %u{sym %08x fast %08x tgt %08x}", st, h.symidx_c[(size_t)st * ncls + c], h.res_idx[((size_t)st * ncls + c) * 2], h.res_idx[((size_t)st * ncls + c) * 2 + 1]);
        printf("\n");
      }
      std::set<uint32_t> nx;
      for (uint32_t st : cur) {
        const uint32_t w = h.symidx_c[(size_t)st * ncls + c];
        if (w & RXE_SELF) nx.insert(st);
        if (w & RXE_INLINE) nx.insert(w & RXE_TGT_MASK);
        if (w & RXE_OVF) { uint32_t off = w & RXE_TGT_MASK; for (uint32_t j = 0; j < h.ovf[off]; j++) nx.insert(h.ovf[off + 1 + j] & RXE_TGT_MASK); }
      }
      cur.swap(nx);
    }
    if (fin) *fin = cur;
  };
  std::vector<std::set<uint32_t>> fin_ref(ns), fin_got(ns);
  for (uint32_t s = 0; s < ns; s++) exact_from(s, {0u}, 0, false, ref, &fin_ref[s]);
  // model
  unsigned evictions = 0;
  for (uint32_t stream0 = 0; stream0 < ns; stream0 += S) {
    const uint32_t n_mine = std::min(S, ns - stream0);
    std::vector<uint32_t> e(64, FREE), sid(64, 0), dnum(64, 0);
    std::vector<char> acc(64, 0), alive(S, 0);
    std::vector<int> how(64, 0);  // debug: 0 initial/in place, 1 slow path, 2 vector placement
    std::vector<std::set<uint32_t>> D(S);
    for (uint32_t l = 0; l < 64; l++) { sid[l] = l < S ? l : 0; if (l < n_mine) { e[l] = 0; alive[l] = 1; } }
    bool spilled = false;
    uint32_t k = 0;
    for (; k < n_passes && !spilled; k++) {
      for (uint32_t l = 0; l < 64; l++) if (acc[l]) got.insert(Ev(stream0 + sid[l], k, e[l]));
      if (k >= n_consume) continue;
      const bool look = k + 1 < n_consume;
      std::vector<uint32_t> e_in = e, sid_in = sid, xs(64), vA(64, 0), xf(64);
      std::vector<char> candA(64, 0), listA(64, 0), candB(64, 0), listB(64, 0);
      auto cls = [&](uint32_t slot, uint32_t kk) { const uint32_t st = stream0 + slot; return (st < ns && kk < sl) ? (uint32_t)h.byte_class[rows[(size_t)st * sl + kk]] : (uint32_t)h.byte_class[0]; };
      for (uint32_t l = 0; l < 64; l++) {
        const uint32_t c = cls(sid[l], k), cn = cls(sid[l], k + 1);
        xf[l] = h.res_idx[((size_t)e[l] * ncls + c) * 2]; xs[l] = h.res_idx[((size_t)e[l] * ncls + c) * 2 + 1];
        if (k >= 1 && l < S && alive[l]) vA[l] = h.res_pin[(size_t)cls(l, k) * pin_cols + (look ? cls(l, k + 1) : ncls)];
        e[l] = xf[l] & 0xFFFF; acc[l] = (xf[l] & RXR_ACC) != 0;
        const bool needA = look ? ((xf[l] >> (16 + (cn & 7))) & 1) : (xf[l] & RXR_NEED) != 0;
        if (dnum[l] && e[l] != e_in[l]) { D[sid[l]].erase(dnum[l]); dnum[l] = 0; }
        candA[l] = needA && !(xf[l] & RXR_OVFL); listA[l] = needA && (xf[l] & RXR_OVFL);
        candB[l] = (vA[l] & RXE_INLINE) && !(vA[l] & RXE_PIN); listB[l] = (vA[l] & RXE_OVF) != 0;
      }
      for (uint32_t l = 0; l < 64; l++) {  // (the GPU does all A then all B atomics; any order is equivalent up to which duplicate wins)
        const uint32_t dA = (xs[l] >> 16) & 0x3FF;
        if (candA[l] && dA) { if (D[sid[l]].count(dA)) candA[l] = 0; else D[sid[l]].insert(dA); }
      }
      for (uint32_t l = 0; l < 64; l++) {
        const uint32_t dB = (vA[l] >> 16) & 0x3FF;
        if (candB[l] && dB) { if (D[l].count(dB)) candB[l] = 0; else D[l].insert(dB); }
      }
      auto evict_one = [&]() {
        std::vector<uint32_t> cnt(S, 0);
        for (uint32_t l = 0; l < 64; l++) { if (e[l] != FREE) cnt[sid[l]]++; if (candA[l]) cnt[sid[l]]++; if (listA[l]) cnt[sid[l]] += 4; if (candB[l]) cnt[l]++; if (listB[l]) cnt[l] += 4; }
        int v = -1; uint32_t best = 0;
        for (uint32_t s2 = 0; s2 < S; s2++) if (alive[s2] && cnt[s2] + 1 > best) { best = cnt[s2] + 1; v = (int)s2; }
        if (v < 0) { spilled = true; return; }
        evictions++;
        if (getenv("RES_MODEL_VERBOSE")) printf("[model] wave %u pass %u: stream slot %d leaves (%u lanes + wishes)\n", stream0 / S, k, v, best - 1);
        std::set<uint32_t> Sk;
        for (uint32_t l = 0; l < 64; l++) if (e_in[l] != FREE && sid_in[l] == (uint32_t)v) Sk.insert(e_in[l]);
        if (getenv("RES_MODEL_VERBOSE")) for (uint32_t l = 0; l < 64; l++) if (e_in[l] != FREE && sid_in[l] == (uint32_t)v) printf("[model]   S_k member: lane %u state %u\n", l, e_in[l]);
        if (k >= 1) Sk.insert(h.pin_state);
        exact_from(stream0 + v, Sk, k, true, got, &fin_got[stream0 + v]);
        for (uint32_t l = 0; l < 64; l++) { if (sid[l] == (uint32_t)v) { if (e[l] != FREE) { e[l] = FREE; acc[l] = 0; } dnum[l] = 0; candA[l] = 0; listA[l] = 0; } }
        candB[v] = 0; listB[v] = 0; D[v].clear(); alive[v] = 0;
        bool any = false; for (uint32_t s2 = 0; s2 < S; s2++) any |= alive[s2];
        if (!any) spilled = true;
      };
      for (int kind = 0; kind < 2 && !spilled; kind++) {
        for (;;) {  // (the pending flags are read afresh for every list: an eviction may have cancelled some)
          int srci = -1;
          for (uint32_t l = 0; l < 64 && srci < 0; l++) if (kind == 0 ? listA[l] : listB[l]) srci = (int)l;
          if (srci < 0) break;
          const uint32_t src = (uint32_t)srci;
          const uint32_t tsid = kind == 0 ? sid[src] : src, off = (kind == 0 ? xs[src] : vA[src]) & RXE_TGT_MASK;
          for (uint32_t j = 0; j < h.res_ovf[off] && alive[tsid]; j++) {
            const uint32_t tw = h.res_ovf[off + 1 + j];
            if (tw & RXE_PIN) continue;
            const uint32_t d = (tw >> 16) & 0x3FF;
            if (d) { if (D[tsid].count(d)) continue; D[tsid].insert(d); }
            auto first_free = [&]() { for (uint32_t l = 0; l < 64; l++) if (e[l] == FREE && !candA[l] && !listA[l]) return (int)l; return -1; };
            int dst = first_free();
            while (dst < 0 && !spilled && alive[tsid]) { evict_one(); dst = first_free(); }
            if (spilled || !alive[tsid]) break;
            e[dst] = tw & 0xFFFF; acc[dst] = (tw & RXE_ACCEPT) != 0; sid[dst] = tsid; dnum[dst] = d; how[dst] = 1 + 10 * (int)src + 1000 * kind;
          }
          if (kind == 0) listA[src] = 0; else listB[src] = 0;
          if (spilled) break;
        }
      }
      if (spilled) break;
      for (;;) {
        uint32_t nA = 0, nB = 0, nf = 0;
        for (uint32_t l = 0; l < 64; l++) { nA += candA[l]; nB += candB[l]; nf += e[l] == FREE; }
        if (nA + nB <= nf) break;
        evict_one();
        if (spilled) break;
      }
      if (spilled) break;
      std::vector<std::tuple<uint32_t, uint32_t>> scr;
      for (uint32_t l = 0; l < 64; l++) if (candA[l]) scr.push_back({xs[l], sid[l]});
      for (uint32_t l = 0; l < 64; l++) if (candB[l]) scr.push_back({vA[l], l});
      uint32_t r = 0;
      for (uint32_t l = 0; l < 64 && r < scr.size(); l++) if (e[l] == FREE) {
        const uint32_t tw = std::get<0>(scr[r]);
        e[l] = tw & 0xFFFF; acc[l] = (tw & RXE_ACCEPT) != 0; sid[l] = std::get<1>(scr[r]); dnum[l] = (tw >> 16) & 0x3FF; how[l] = 2; r++;
      }
      // invariant: no (stream, state) twice
      std::set<std::pair<uint32_t, uint32_t>> seen;
      for (uint32_t l = 0; l < 64; l++) if (e[l] != FREE && !seen.insert({sid[l], e[l]}).second) {
        printf("DUPLICATE entry stream %u state %u after pass %u:", stream0 + sid[l], e[l], k);
        for (uint32_t m = 0; m < 64; m++) if (e[m] == e[l] && sid[m] == sid[l]) printf(" lane %u how %d dnum %u e_in %u sid_in %u xf %08x;", m, how[m], dnum[m], e_in[m], sid_in[m], xf[m]);
        printf("\n");
        for (uint32_t m = 0; m < 64; m++) if (e[m] == e[l] && sid[m] == sid[l] && how[m] % 10 == 1) {
          const uint32_t src = (how[m] % 1000) / 10;
          printf("   source lane %u: e_in %u sid_in %u xf %08x xs %08x class %u; list:", src, e_in[src], sid_in[src], xf[src], xs[src], cls(sid_in[src], k));
          const uint32_t off = xs[src] & RXE_TGT_MASK;
          for (uint32_t j = 0; j < h.res_ovf[off]; j++) printf(" %08x", h.res_ovf[off + 1 + j]);
          printf("\n");
        }
      }
    }
    if (!spilled) for (uint32_t l = 0; l < 64; l++) if (e[l] != FREE) fin_got[stream0 + sid[l]].insert(e[l]);
    if (!spilled && n_consume >= 1) for (uint32_t s2 = 0; s2 < n_mine; s2++) if (alive[s2]) fin_got[stream0 + s2].insert(h.pin_state);
  }
  std::vector<Ev> extra, missing;
  std::set_difference(got.begin(), got.end(), ref.begin(), ref.end(), std::back_inserter(extra));
  std::set_difference(ref.begin(), ref.end(), got.begin(), got.end(), std::back_inserter(missing));
  printf("events ref %zu model %zu, evictions %u, extra %zu missing %zu\n", ref.size(), got.size(), evictions, extra.size(), missing.size());
  unsigned fin_bad = 0;
  for (uint32_t s = 0; s < ns; s++) fin_bad += fin_ref[s] != fin_got[s];
  printf("final sets wrong: %u\n", fin_bad);
  for (size_t i = 0; i < extra.size() && i < 8; i++) printf("  extra (%u,%u,%u)\n", std::get<0>(extra[i]), std::get<1>(extra[i]), std::get<2>(extra[i]));
  for (size_t i = 0; i < missing.size() && i < 8; i++) printf("  missing (%u,%u,%u)\n", std::get<0>(missing[i]), std::get<1>(missing[i]), std::get<2>(missing[i]));
  return (extra.size() || missing.size() || fin_bad) ? 3 : 0;
}
