// CPU walk of the register kernel's tables over one trace: how often a pass needs the placement path, and for what.
// build: g++ -O2 -std=c++17 -Iregex-fpga_amd/csrc -Iinclude tools/reg_sim.cpp regex-fpga_amd/csrc/rx_host.cpp -o /tmp/reg_sim
// usage: /tmp/reg_sim table.coe trace.mem [n [window]]   (window: the set is reset every `window` bytes, like the T workload's streams)
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "rx_internal.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::string txt;
  std::vector<uint32_t> W;
  if (rxh_read_file(argv[1], &txt) || rxh_parse_coe_text(txt.data(), txt.size(), &W)) return 1;
  RxHostNfa h;
  if (rxh_build(W.data(), W.size(), 0, &h)) return 1;
  std::string mt;
  std::vector<uint8_t> tr;
  if (rxh_read_file(argv[2], &mt) || rxh_parse_mem_text(mt.data(), mt.size(), &tr)) return 1;
  size_t n = argc > 3 ? atol(argv[3]) : 200000;
  const size_t win = argc > 4 ? atol(argv[4]) : 0;
  if (n > tr.size()) n = tr.size();
  const uint32_t ncls = h.n_classes, size = h.size, FREE = size;
  printf("size %u classes %u pin %u fold %d\n", size, ncls, h.pin_state, (int)!h.pin_tab.empty());
  // static: per state, how many classes carry something
  {
    std::map<int, int> hist;
    int simple = 0, selfy = 0;
    for (uint32_t s = 0; s < size; s++) {
      int nz = 0, need = 0, self = 0;
      std::set<uint32_t> vals;
      for (uint32_t k = 0; k < ncls; k++) {
        uint32_t f = h.regidx[((size_t)s * ncls + k) * 2];
        if ((f & h.reg_tmask) != FREE || (f & RXR_NEED)) nz++, vals.insert(f);
        if (f & RXR_NEED) need++;
        if ((f & h.reg_tmask) == s) self++;
      }
      hist[nz > 8 ? 9 : nz]++;
      if (vals.size() <= 1 && need == 0) simple++;
      if (self) selfy++;
    }
    for (auto& kv : hist) printf("states with %d%s live classes: %d\n", kv.first, kv.first == 9 ? "+" : "", kv.second);
    printf("states whose live classes share one fast word and never need placement: %d; states with a self loop: %d\n", simple, selfy);
  }
  // descriptor-representable: every class but at most one has the default word (self or free, nothing to place)
  std::vector<uint8_t> cplx(size + 1, 0);
  {
    int nc = 0;
    for (uint32_t s = 0; s < size; s++) {
      int self = 0, fre = 0, other = 0;
      std::set<std::pair<uint32_t,uint32_t>> ov;
      for (uint32_t k = 0; k < ncls; k++) {
        uint32_t f = h.regidx[((size_t)s * ncls + k) * 2];
        if (f == s) self++; else if (f == FREE) fre++; else other++, ov.insert({f, h.regidx[((size_t)s * ncls + k) * 2 + 1]});
      }
      const bool ok = other <= 1 && (self == 0 || fre == 0);
      const bool ok2 = ov.size() <= 1 && (self == 0 || fre == 0);
      cplx[s] = ok ? 0 : (ok2 ? 1 : 2);
      nc += !ok;
    }
    printf("states a one-special-class descriptor cannot hold: %d of %u\n", nc, size);
  }
  unsigned long long hist_items[6] = {0, 0, 0, 0, 0, 0}, n_empty = 0, last_empty = 0, cl1 = 0, cl2 = 0, cpass = 0, slow_la = 0, slow_la1 = 0, slow_la8 = 0, kept = 0, kept1 = 0;
  std::vector<uint32_t> e(1, 0u);  // lanes in use (no FREE kept)
  unsigned long long slow = 0, c_extra = 0, c_dupc = 0, c_ovfl = 0, c_va_inl = 0, c_va_ovf = 0, places = 0, dupchecks = 0, act = 0, maxact = 0;
  unsigned long long p_extra = 0, p_dupc = 0, p_ovfl = 0, p_va = 0, only_va = 0, moved = 0, complexlanes = 0;
  std::map<uint32_t, unsigned long long> hot;
  for (size_t k = 0; k + 1 < n; k++) {  // tb-compat: passes 0..n-2
    const uint32_t c = h.byte_class[tr[k]], cn = h.byte_class[tr[k + 1]];
    if (win && k % win == 0) e.assign(1, 0u);
    uint32_t vA = 0;
    if (!h.pin_tab.empty() && (win ? k % win : k) >= 1) vA = h.pin_tab[(size_t)c * (ncls + 1) + (k + 2 < n ? cn : ncls)];
    std::vector<uint32_t> nx, cand_nodup, cand_dup, lists;
    int ne = 0, nd = 0, no = 0;
    { int any = 0; for (uint32_t s : e) { if (cplx[s] == 1) cl1++, any = 1; if (cplx[s] == 2) cl2++, any = 1; } cpass += any; }
    for (uint32_t s : e) {
      const uint32_t f = h.regidx[((size_t)s * ncls + c) * 2], w = h.regidx[((size_t)s * ncls + c) * 2 + 1];
      hot[s]++;
      const uint32_t v = f & h.reg_tmask;
      if (v != FREE) { nx.push_back(v); if (v != s) moved++; }
      if (f & RXR_NEED) {
        if (f & RXR_EXTRA) ne++, cand_nodup.push_back(w);
        if (f & RXR_DUPC) nd++, cand_dup.push_back(w);
        if (f & RXR_OVFL) no++, lists.push_back(w & RXE_TGT_MASK);
      }
    }
    {  // with one byte of look-ahead on the single targets too (exact, and the one-live-class approximation)
      int keep = 0, keep1 = 0, keep8 = 0;
      auto live8 = [&](uint32_t w) {  // what the kernel does: the target's classes mod 8 (RxParams::reg_tmask)
        const uint32_t t = w & RXE_TGT_MASK;
        if ((w & RXE_ACCEPT) || k + 2 >= n) return true;
        for (uint32_t q = cn & 7u; q < ncls; q += 8u) if (h.symidx_c[(size_t)t * ncls + q] != 0u) return true;
        return false;
      };
      auto live = [&](uint32_t w, bool one) {
        const uint32_t t = w & RXE_TGT_MASK;
        if (w & RXE_ACCEPT) return true;
        if (k + 2 >= n) return true;
        if (one) { int nl = 0; for (uint32_t q = 0; q < ncls; q++) nl += h.symidx_c[(size_t)t * ncls + q] != 0; if (nl != 1) return true; }
        return h.symidx_c[(size_t)t * ncls + cn] != 0u;
      };
      for (uint32_t w : cand_nodup) keep += live(w, false), keep1 += live(w, true), keep8 += live8(w);
      for (uint32_t w : cand_dup) keep += live(w, false), keep1 += live(w, true), keep8 += live8(w);
      if (keep8 || no || vA) slow_la8++;
      { int items = keep8 + (vA ? 1 : 0) + no; hist_items[items > 5 ? 5 : items]++; }
      if (keep || no || vA) slow_la++;
      if (keep1 || no || vA) slow_la1++;
      kept += keep; kept1 += keep1;
    }
    const bool is_slow = ne || nd || no || vA;
    if (is_slow) slow++;
    c_extra += ne; c_dupc += nd; c_ovfl += no;
    p_extra += ne != 0; p_dupc += nd != 0; p_ovfl += no != 0; p_va += vA != 0;
    if (vA && !(ne || nd || no)) only_va++;
    auto place = [&](uint32_t tw) {
      if (!h.pin_tab.empty() && (tw & RXE_PIN)) return;
      const uint32_t t = tw & RXE_TGT_MASK;
      places++;
      if (tw & RXE_MAYDUP) { dupchecks++; for (uint32_t x : nx) if (x == t) return; }
      nx.push_back(t);
    };
    for (uint32_t w : cand_nodup) place(w);
    for (uint32_t w : cand_dup) place(w);
    for (uint32_t off : lists) for (uint32_t j = 0; j < h.ovf[off]; j++) place(h.ovf[off + 1 + j]);
    if (vA & RXE_INLINE) { c_va_inl++; place(vA); }
    else if (vA & RXE_OVF) { c_va_ovf++; const uint32_t off = vA & RXE_TGT_MASK; for (uint32_t j = 0; j < h.ovf[off]; j++) place(h.ovf[off + 1 + j]); }
    e.swap(nx);
    if (e.empty()) { n_empty++; last_empty = k; }
    act += e.size();
    if (e.size() > maxact) maxact = e.size();
  }
  const double P = (double)(n - 1);
  printf("passes %zu mean lanes in use %.2f max %llu; lanes that moved to another state per pass %.2f\n", n - 1, act / P, maxact, moved / P);
  printf("passes needing placement %.3f  (EXTRA %.3f DUPC %.3f OVFL %.3f folded-state emission %.3f, emission only %.3f)\n", slow / P, p_extra / P, p_dupc / P,
         p_ovfl / P, p_va / P, only_va / P);
  printf("per pass: EXTRA lanes %.3f DUPC lanes %.3f OVFL lanes %.3f; emissions inline %.3f list %.3f; place() calls %.3f, of them with duplicate check %.3f\n",
         c_extra / P, c_dupc / P, c_ovfl / P, c_va_inl / P, c_va_ovf / P, places / P, dupchecks / P);
  printf("things to place per pass (single targets that survive the mod-8 look-ahead + emission + lists): 0: %.3f  1: %.3f  2: %.3f  3: %.3f  4: %.3f  5+: %.3f\n",
         hist_items[0] / P, hist_items[1] / P, hist_items[2] / P, hist_items[3] / P, hist_items[4] / P, hist_items[5] / P);
  printf("passes after which no lane is in use: %.3f of all, the last one is pass %llu\n", n_empty / P, last_empty);
  printf("with look-ahead on single targets: passes needing placement %.3f (exact), %.3f (classes mod 8, as built), %.3f (targets with one live class only); single targets placed per pass %.3f / %.3f\n", slow_la / P, slow_la8 / P, slow_la1 / P, kept / P, kept1 / P);
  printf("lane-passes on states without a descriptor: same-action multi-class %.4f, other %.4f per pass; passes with any %.4f\n", cl1 / P, cl2 / P, cpass / P);
  // which states hold lanes
  std::vector<std::pair<unsigned long long, uint32_t>> hv;
  for (auto& kv : hot) hv.push_back({kv.second, kv.first});
  std::sort(hv.rbegin(), hv.rend());
  unsigned long long tot = 0, cum = 0;
  for (auto& x : hv) tot += x.first;
  printf("distinct states seen %zu; lane-passes %llu\n", hv.size(), tot);
  for (size_t i = 0; i < hv.size() && i < 40; i++) {
    cum += hv[i].first;
    int nz = 0, self = 0, need = 0;
    for (uint32_t k = 0; k < ncls; k++) {
      uint32_t f = h.regidx[((size_t)hv[i].second * ncls + k) * 2];
      if ((f & h.reg_tmask) != FREE || (f & RXR_NEED)) nz++;
      if ((f & h.reg_tmask) == hv[i].second) self++;
      if (f & RXR_NEED) need++;
    }
    printf("  state %5u  share %.3f cum %.3f  live classes %d self %d need %d\n", hv[i].second, (double)hv[i].first / tot, (double)cum / tot, nz, self, need);
  }
  return 0;
}
