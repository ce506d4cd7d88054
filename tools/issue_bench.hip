// issue_bench.hip — what a gfx950 SIMD can ISSUE, per instruction mix and waves per SIMD.
// Diagnostic only (never linked into librxmatch.so).  Build: hipcc -O2 --offload-arch=gfx950 tools/issue_bench.hip -o tools/issue_bench
// The pack kernel's pass loop (rx_kernels.hip) runs at ~0.25 wave-instructions per cycle per SIMD while no single pipe is
// above a third busy; this prints the attainable wave-instructions/cycle/SIMD for streams of INDEPENDENT instructions of
// one kind, for mixes of kinds, and for a stream shaped like that loop (52 % VALU, 30 % SALU, 7 % branches, 6 % LDS, waits),
// at 1..8 waves per SIMD, so that the kernel can be quoted against the real ceiling for its mix.
// Every body is written in inline asm (volatile: the compiler neither reorders nor removes it); registers are named by
// the compiler through constraints.  One block = 256 threads = one wave per SIMD; dynamic LDS is sized so that exactly
// `w` blocks fit a CU, and the grid is 256 * w blocks: every SIMD of the chip holds w waves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// --- instruction atoms: strings that are concatenated into ONE asm statement per loop body (the compiler pads
// back-to-back asm statements that write SGPRs with s_nop; inside one statement nothing is inserted) ------------------
// operands: %0-%7 VGPRs v0..v7, %8-%15 SGPRs s0..s7, %16 LDS address, %17 %18 LDS results, %19 scalar count, %20 vector rank
#define V(n) "v_add_u32 %" #n ", %" #n ", %" #n "\n\t"
#define S(n) "s_add_u32 %" #n ", %" #n ", %" #n "\n\t"
#define VX(a, b) "v_xor_b32 %" #a ", %" #a ", %" #b "\n\t"
// never-taken branch: s_cmp + s_cbranch (as the kernel's rare-path tests)
#define BR(n) "s_cmp_eq_u32 %" #n ", 0x7fffffff\n\ts_cbranch_scc1 99f\n\t"
// taken branch to the next instruction (the wave's instruction buffer is refilled)
#define BT(l) "s_branch " #l "f\n" #l ":\n\t"
#define LR(d) "ds_read_b32 %" #d ", %16\n\t"
#define LW(v) "ds_write_b32 %16, %" #v "\n\t"
#define LA(d, v) "ds_or_rtn_b32 %" #d ", %16, %" #v "\n\t"
#define WL "s_waitcnt lgkmcnt(0)\n\t"
// ballot -> scalar count -> vector rank: the kernel's slot allocation (VALU -> SGPR pair -> SALU, VALU)
#define BAL(v) "v_cmp_ne_u32 vcc, 0, %" #v "\n\ts_bcnt1_i32_b64 %19, vcc\n\tv_mbcnt_lo_u32_b32 %20, vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %20, vcc_hi, %20\n\t"
#define SC "s_add_u32 %19, %19, %19\n\t"
#define VR "v_add_u32 %20, %20, %20\n\t"
#define BODY(str)                                                                                                              \
  asm volatile(str "99:\n\t"                                                                                                   \
               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), \
                 "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7), "+v"(a0), "+v"(l0), "+v"(l1), "+s"(c0), "+v"(r0)            \
               :                                                                                                               \
               : "scc", "vcc", "memory")
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R12(x) R8(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R24(x) R16(x) R8(x)
#define R96(x) R16(x) R16(x) R16(x) R16(x) R16(x) R16(x)

enum { MIX_V = 0, MIX_S, MIX_VS, MIX_VS31, MIX_BR, MIX_BT, MIX_LDS, MIX_BAL, MIX_VDEP, MIX_SDEP, MIX_LOOP, MIX_LOOP_DEP,
       MIX_CMPV, MIX_CMPS, MIX_RFL, MIX_CBRV, MIX_VOP3, MIX_VLIT, MIX_WAIT, MIX_NOP, MIX_LATOM, MIX_LWR,
       MIX_VSGPR, MIX_CMPV_S, MIX_VOP3_S, MIX_VOP3_2SRC, MIX_MBCNT, MIX_V_CMPV, MIX_VSGPR_S, MIX_N };
static const char* mix_name[MIX_N] = {"VALU x96 independent",           "SALU x96 independent",           "VALU/SALU alternating 48+48",
                                      "VALU x3 : SALU x1, 72+24",       "s_cmp+s_cbranch (not taken) x48", "s_branch taken x96",
                                      "ds_read_b32 x16 + wait, x6",     "ballot->bcnt->mbcnt x24",         "VALU x96 one dependent chain",
                                      "SALU x96 one dependent chain",   "pack-loop mix (independent)",     "pack-loop mix (dependent chain)",
                                      "v_cmp -> vcc x96",               "v_cmp -> vcc, s_bcnt1 vcc x48",   "v_readfirstlane, s_add x48",
                                      "v_cmp, s_cbranch_vccz (not taken) x48", "VOP3 v_lshl_add_u32 x96",    "VOP2 + 32-bit literal x96",
                                      "s_waitcnt (nothing pending) x96", "s_nop 0 x96",                    "ds_or_rtn_b32 x16 + wait, x6",
                                      "ds_write_b32 x16 + wait, x6",
                                      "VOP2 with an SGPR source x96",   "v_cmp -> vcc / s_add alternating 48+48", "VOP3 3-src / s_add alternating 48+48",
                                      "VOP3 encoding, 2 VGPR sources x96", "v_mbcnt_lo (SGPR source) x96",   "VALU / v_cmp -> vcc alternating 48+48",
                                      "VOP2 with SGPR source / s_add alternating 48+48"};
// instructions per loop body, counted in the .s by tools/count_loop_insts.py (loop control adds 3)
static const int mix_insts[MIX_N] = {96, 96, 96, 96, 96, 96, 102, 120, 96, 96, 96, 102, 96, 96, 96, 96, 96, 96, 96, 96, 102, 102, 96, 96, 96, 96, 96, 96, 96};

template <int MIX>
__global__ void __launch_bounds__(256) issue_kernel(uint32_t iters, unsigned long long* cyc, unsigned long long* rt, uint32_t* sink) {
  extern __shared__ uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7;
  uint32_t s0 = iters, s1 = iters + 1, s2 = iters + 2, s3 = iters + 3, s4 = iters + 4, s5 = iters + 5, s6 = iters + 6, s7 = iters + 7;
  uint32_t a0 = (threadIdx.x * 4u) & 4095u, l0 = 0, l1 = 0;
  uint32_t c0 = 0, r0 = 0;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t it = 0; it < iters; it++) {
    if (MIX == MIX_V) {
      BODY(R12(V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7)));
    } else if (MIX == MIX_S) {
      BODY(R12(S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)));
    } else if (MIX == MIX_VS) {
      BODY(R12(V(0) S(8) V(1) S(9) V(2) S(10) V(3) S(11)));
    } else if (MIX == MIX_VS31) {
      BODY(R12(V(0) V(1) V(2) S(8) V(3) V(4) V(5) S(9)));
    } else if (MIX == MIX_BR) {
      BODY(R12(BR(8) BR(9) BR(10) BR(11)));
    } else if (MIX == MIX_BT) {
      BODY(R12(BT(1) BT(2) BT(3) BT(4) BT(5) BT(6) BT(7) BT(8)));
    } else if (MIX == MIX_LDS) {
      BODY(R2(R2(LR(17) LR(18) LR(17) LR(18)) R2(LR(17) LR(18) LR(17) LR(18)) WL) R4(R2(LR(17) LR(18) LR(17) LR(18)) R2(LR(17) LR(18) LR(17) LR(18)) WL));
    } else if (MIX == MIX_BAL) {
      BODY(R24(BAL(0) "v_add_u32 %0, %20, %19\n\t"));
    } else if (MIX == MIX_VDEP) {
      BODY(R96(V(0)));
    } else if (MIX == MIX_SDEP) {
      BODY(R96(S(8)));
    } else if (MIX == MIX_CMPV) {
      BODY(R12("v_cmp_ne_u32 vcc, 0, %0\n\t" "v_cmp_ne_u32 vcc, 0, %1\n\t" "v_cmp_ne_u32 vcc, 0, %2\n\t" "v_cmp_ne_u32 vcc, 0, %3\n\t"
               "v_cmp_ne_u32 vcc, 0, %4\n\t" "v_cmp_ne_u32 vcc, 0, %5\n\t" "v_cmp_ne_u32 vcc, 0, %6\n\t" "v_cmp_ne_u32 vcc, 0, %7\n\t"));
    } else if (MIX == MIX_CMPS) {
      BODY(R12(R4("v_cmp_ne_u32 vcc, 0, %0\n\ts_bcnt1_i32_b64 %19, vcc\n\t")));
    } else if (MIX == MIX_RFL) {
      BODY(R12("v_readfirstlane_b32 %8, %0\n\t" S(8) "v_readfirstlane_b32 %9, %1\n\t" S(9) "v_readfirstlane_b32 %10, %2\n\t" S(10)
               "v_readfirstlane_b32 %11, %3\n\t" S(11)));
    } else if (MIX == MIX_CBRV) {
      BODY(R12(R4("v_cmp_ne_u32 vcc, 0x7fffffff, %0\n\ts_cbranch_vccz 99f\n\t")));
    } else if (MIX == MIX_VOP3) {
      BODY(R12("v_lshl_add_u32 %0, %0, 1, %0\n\t" "v_lshl_add_u32 %1, %1, 1, %1\n\t" "v_lshl_add_u32 %2, %2, 1, %2\n\t" "v_lshl_add_u32 %3, %3, 1, %3\n\t"
               "v_lshl_add_u32 %4, %4, 1, %4\n\t" "v_lshl_add_u32 %5, %5, 1, %5\n\t" "v_lshl_add_u32 %6, %6, 1, %6\n\t" "v_lshl_add_u32 %7, %7, 1, %7\n\t"));
    } else if (MIX == MIX_VLIT) {
      BODY(R12("v_and_b32 %0, 0x12345678, %0\n\t" "v_and_b32 %1, 0x12345678, %1\n\t" "v_and_b32 %2, 0x12345678, %2\n\t" "v_and_b32 %3, 0x12345678, %3\n\t"
               "v_and_b32 %4, 0x12345678, %4\n\t" "v_and_b32 %5, 0x12345678, %5\n\t" "v_and_b32 %6, 0x12345678, %6\n\t" "v_and_b32 %7, 0x12345678, %7\n\t"));
    } else if (MIX == MIX_VSGPR) {
      BODY(R12("v_add_u32 %0, %8, %0\n\t" "v_add_u32 %1, %9, %1\n\t" "v_add_u32 %2, %10, %2\n\t" "v_add_u32 %3, %11, %3\n\t"
               "v_add_u32 %4, %12, %4\n\t" "v_add_u32 %5, %13, %5\n\t" "v_add_u32 %6, %14, %6\n\t" "v_add_u32 %7, %15, %7\n\t"));
    } else if (MIX == MIX_CMPV_S) {
      BODY(R12("v_cmp_ne_u32 vcc, 0, %0\n\t" S(8) "v_cmp_ne_u32 vcc, 0, %1\n\t" S(9) "v_cmp_ne_u32 vcc, 0, %2\n\t" S(10) "v_cmp_ne_u32 vcc, 0, %3\n\t" S(11)));
    } else if (MIX == MIX_VOP3_S) {
      BODY(R12("v_lshl_add_u32 %0, %0, 1, %0\n\t" S(8) "v_lshl_add_u32 %1, %1, 1, %1\n\t" S(9) "v_lshl_add_u32 %2, %2, 1, %2\n\t" S(10)
               "v_lshl_add_u32 %3, %3, 1, %3\n\t" S(11)));
    } else if (MIX == MIX_VOP3_2SRC) {
      BODY(R12("v_add_u32_e64 %0, %0, %0\n\t" "v_add_u32_e64 %1, %1, %1\n\t" "v_add_u32_e64 %2, %2, %2\n\t" "v_add_u32_e64 %3, %3, %3\n\t"
               "v_add_u32_e64 %4, %4, %4\n\t" "v_add_u32_e64 %5, %5, %5\n\t" "v_add_u32_e64 %6, %6, %6\n\t" "v_add_u32_e64 %7, %7, %7\n\t"));
    } else if (MIX == MIX_MBCNT) {
      BODY(R12("v_mbcnt_lo_u32_b32 %0, %8, %0\n\t" "v_mbcnt_lo_u32_b32 %1, %9, %1\n\t" "v_mbcnt_lo_u32_b32 %2, %10, %2\n\t" "v_mbcnt_lo_u32_b32 %3, %11, %3\n\t"
               "v_mbcnt_lo_u32_b32 %4, %12, %4\n\t" "v_mbcnt_lo_u32_b32 %5, %13, %5\n\t" "v_mbcnt_lo_u32_b32 %6, %14, %6\n\t" "v_mbcnt_lo_u32_b32 %7, %15, %7\n\t"));
    } else if (MIX == MIX_V_CMPV) {
      BODY(R12(V(0) "v_cmp_ne_u32 vcc, 0, %4\n\t" V(1) "v_cmp_ne_u32 vcc, 0, %5\n\t" V(2) "v_cmp_ne_u32 vcc, 0, %6\n\t" V(3) "v_cmp_ne_u32 vcc, 0, %7\n\t"));
    } else if (MIX == MIX_VSGPR_S) {
      BODY(R12("v_add_u32 %0, %12, %0\n\t" S(8) "v_add_u32 %1, %13, %1\n\t" S(9) "v_add_u32 %2, %14, %2\n\t" S(10) "v_add_u32 %3, %15, %3\n\t" S(11)));
    } else if (MIX == MIX_WAIT) {
      BODY(R96(WL));
    } else if (MIX == MIX_NOP) {
      BODY(R96("s_nop 0\n\t"));
    } else if (MIX == MIX_LATOM) {
      BODY(R2(R8(LA(17, 0) LA(18, 1)) WL) R4(R8(LA(17, 0) LA(18, 1)) WL));
    } else if (MIX == MIX_LWR) {
      BODY(R2(R8(LW(0) LW(1)) WL) R4(R8(LW(0) LW(1)) WL));
    } else if (MIX == MIX_LOOP) {
      // 100 instructions: 52 VALU, 30 SALU, 3.5 x (s_cmp + s_cbranch) = 7, 6 LDS (3 reads, 1 write, 2 atomics with
      // return), 5 waits.  Independent registers: what the issue logic can do with this mix.
      BODY(V(0) V(1) S(8) V(2) LR(17) S(9) V(3) V(4) S(10) V(5)
           BR(11) V(6) V(7) S(12) WL V(0) S(13) V(1) LW(2) V(3)
           S(14) V(4) V(5) S(15) V(6) BR(8) V(7) S(9) V(0) V(1)
           LA(17, 7) LA(18, 6) S(10) V(2) S(11) V(3) WL V(4) S(12) V(5)
           V(6) S(13) V(7) S(14) V(0) V(1) S(15) LR(18) V(2) S(8)
           V(3) V(4) S(9) WL V(5) S(10) V(6) V(7) S(11) V(0)
           BR(12) V(1) S(13) V(2) V(3) S(14) V(4) V(5) S(15) V(6)
           LR(17) V(7) S(8) V(0) S(9) V(1) WL V(2) S(10) V(3)
           S(11) V(4) S(12) V(5) WL S(13) V(6) S(14) V(7) V(0)
           V(1) V(2) "s_cmp_eq_u32 %15, 0x7fffffff\n\t");
    } else if (MIX == MIX_LOOP_DEP) {
      // the same counts, shaped like the pass: list read -> wait -> class byte read -> wait -> address VALU -> [global
      // gather replaced by an LDS read] -> wait -> candidate VALU -> two LDS atomics -> wait -> ballots -> list write
      BODY(S(8) S(9) LR(17) S(10) V(1) WL VX(0, 17) V(0) V(0) V(0)
           BR(11) V(0) LR(18) S(12) V(1) WL VX(0, 18) V(0) LW(0) V(0)
           S(13) V(0) V(0) S(14) V(0) BR(8) V(0) S(15) V(0) V(0)
           LR(17) S(9) V(1) S(10) WL VX(0, 17) V(0) V(0) V(0) V(0)
           V(0) S(11) V(0) S(12) V(0) V(0) S(13) V(0) V(0) V(0)
           LA(17, 0) LA(18, 0) S(14) WL VX(0, 17) VX(0, 18) V(0) V(0) S(15) V(0)
           BAL(0) SC VR VR SC SC VR
           BAL(20) SC VR LW(20) SC VR
           BR(12) S(8) V(0) S(9) V(0) S(10) V(0) S(11) V(0) S(12)
           V(0) S(13) V(0) S(14) S(15) WL V(0) V(0) V(0) V(0));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (lane == 0) { cyc[wave] = t1 - t0; rt[wave] = q1 - q0; }
  uint32_t acc = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7 ^ l0 ^ l1 ^ c0 ^ r0;
  if (acc == 0x12345678u) sink[0] = acc;  // keeps every register alive
}

// Scattered dword gathers, the pack kernel's slice read: per gather two VALU (new pseudo-random offset inside a region
// of `mask + 1` bytes) + global_load_dword with a scalar base; 8 in flight, then s_waitcnt vmcnt(0).  `active` lanes.
__global__ void __launch_bounds__(256) gather_kernel(uint32_t iters, const uint32_t* __restrict__ table, uint32_t mask, uint32_t active,
                                                     unsigned long long* cyc, unsigned long long* rt, uint32_t* sink) {
  extern __shared__ uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  lds[threadIdx.x] = 0;
  uint32_t o[8], d[8];
  uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
#pragma unroll
  for (int q = 0; q < 8; q++) { h = h * 1664525u + 1013904223u; o[q] = (h >> 4) & mask & ~3u; d[q] = 0; }
  uint32_t stride = 0x9E3779B1u & mask & ~3u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  if (lane < active) {
    for (uint32_t it = 0; it < iters; it++) {
#define G(n, m) "v_add_u32 %" #n ", %" #n ", %17\n\tv_and_b32 %" #n ", %18, %" #n "\n\tglobal_load_dword %" #m ", %" #n ", %16\n\t"
      asm volatile(G(0, 8) G(1, 9) G(2, 10) G(3, 11) G(4, 12) G(5, 13) G(6, 14) G(7, 15) "s_waitcnt vmcnt(0)\n\t"
                   : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "=&v"(d[0]), "=&v"(d[1]),
                     "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
                   : "s"(table), "s"(stride), "s"(mask & ~3u)
                   : "memory");
#undef G
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (lane == 0) { cyc[wave] = t1 - t0; rt[wave] = q1 - q0; }
  uint32_t acc = d[0] ^ d[1] ^ d[2] ^ d[3] ^ d[4] ^ d[5] ^ d[6] ^ d[7];
  if (acc == 0x12345678u) sink[0] = acc;
}

static void run_gather(int w, uint32_t iters, uint32_t region, uint32_t active, const uint32_t* d_table, unsigned long long* d_cyc,
                       unsigned long long* d_rt, uint32_t* d_sink, int cus) {
  const uint32_t lds = (uint32_t)((160 * 1024) / w) & ~1023u;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gather_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const uint32_t grid = (uint32_t)cus * (uint32_t)w;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), lds, 0, iters, d_table, region - 1u, active, d_cyc, d_rt, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const size_t nw = (size_t)grid * 4;
  std::vector<unsigned long long> cyc(nw), rt(nw);
  CHECK(hipMemcpy(cyc.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(rt.data(), d_rt, nw * 8, hipMemcpyDeviceToHost));
  std::sort(cyc.begin(), cyc.end());
  double clk = 0;
  for (size_t i = 0; i < nw; i++) clk += rt[i] ? (double)cyc[i] / (double)rt[i] * 0.1 : 0.0;
  clk /= (double)nw;
  const double gathers = 8.0 * iters * w * 4.0;  // per CU
  printf("gather: region %8u B, %2u lanes, w=%d  %7.3f ms  clock %.2f GHz  %.1f cycles per gather per CU (wall), %.0f cycles per gather for its wave\n",
         region, active, w, ms, clk, ms * 1e-3 * clk * 1e9 / gathers, (double)cyc[nw / 2] / (8.0 * iters));
  fflush(stdout);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

template <int MIX>
static void run_mix(int w, uint32_t iters, unsigned long long* d_cyc, unsigned long long* d_rt, uint32_t* d_sink, int cus) {
  const uint32_t lds = (uint32_t)((160 * 1024) / w) & ~1023u;
  const uint32_t lds_use = lds > 65536u ? 65536u : lds;  // above 64 KB needs an attribute; 64 KB already caps a CU at 2 blocks
  // for w = 1 and 2 the LDS cap alone cannot pin the count: use max dynamic LDS
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(issue_kernel<MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  (void)lds_use;
  const uint32_t grid = (uint32_t)cus * (uint32_t)w;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {  // first run warms the instruction cache and the clock
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(issue_kernel<MIX>, dim3(grid), dim3(256), lds, 0, iters, d_cyc, d_rt, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const size_t nw = (size_t)grid * 4;
  std::vector<unsigned long long> cyc(nw), rt(nw);
  CHECK(hipMemcpy(cyc.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(rt.data(), d_rt, nw * 8, hipMemcpyDeviceToHost));
  std::vector<double> clk(nw);
  for (size_t i = 0; i < nw; i++) clk[i] = rt[i] ? (double)cyc[i] / (double)rt[i] * 0.1 : 0.0;  // GHz (memrealtime = 100 MHz)
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  const double med = (double)cyc[nw / 2];
  const double insts = (double)mix_insts[MIX] + 3.0;  // + loop counter, compare, branch
  const double per_wave = med / ((double)iters * insts);
  printf("%-34s w=%d  %7.3f ms  clock %.2f GHz  cycles/inst/wave %6.2f  wave-inst/cycle/SIMD %.3f  (wall-based %.3f)\n", mix_name[MIX], w, ms,
         clk[nw / 2], per_wave, (double)w / per_wave, insts * iters * w / (ms * 1e-3 * clk[nw / 2] * 1e9));
  fflush(stdout);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 20000u;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, iters %u\n", prop.gcnArchName, cus, iters);
  unsigned long long *d_cyc, *d_rt;
  uint32_t* d_sink;
  CHECK(hipMalloc(&d_cyc, (size_t)cus * 8 * 4 * 8));
  CHECK(hipMalloc(&d_rt, (size_t)cus * 8 * 4 * 8));
  CHECK(hipMalloc(&d_sink, 64));
  const bool quick = argc > 2;  // second argument: only the mixes added in the second session + the gathers
  const bool third = argc > 2 && argv[2][0] == '3';  // "3": only the scalar-path mixes of the third session
  uint32_t* d_table;
  CHECK(hipMalloc(&d_table, 8u << 20));
  CHECK(hipMemset(d_table, 1, 8u << 20));
  if (third) {
    for (int w : {1, 2, 4, 8}) {
      run_mix<MIX_CMPV>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_VOP3>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_VOP3_2SRC>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_VLIT>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_VSGPR>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_MBCNT>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_CMPV_S>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_VOP3_S>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_VSGPR_S>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_V_CMPV>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_CMPS>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_RFL>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_CBRV>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_LATOM>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_LWR>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_NOP>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_V>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_S>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_BAL>(w, iters, d_cyc, d_rt, d_sink, cus);
      run_mix<MIX_LOOP_DEP>(w, iters, d_cyc, d_rt, d_sink, cus);
    }
    return 0;
  }
  for (int w : {1, 2, 4, 5, 8})
    for (uint32_t region : {4096u, 32768u, 262144u, 4u << 20})
      for (uint32_t active : {64u, 32u, 8u}) run_gather(w, iters / 10, region, active, d_table, d_cyc, d_rt, d_sink, cus);
  const int ws[] = {1, 2, 4, 8};
  for (int w : ws) {
    run_mix<MIX_CMPV>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_CMPS>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_RFL>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_CBRV>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_VOP3>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_VLIT>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_WAIT>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_NOP>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_LATOM>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_LWR>(w, iters, d_cyc, d_rt, d_sink, cus);
    if (quick) continue;
    run_mix<MIX_V>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_S>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_VS>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_VS31>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_BR>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_BT>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_LDS>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_BAL>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_VDEP>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_SDEP>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_LOOP>(w, iters, d_cyc, d_rt, d_sink, cus);
    run_mix<MIX_LOOP_DEP>(w, iters, d_cyc, d_rt, d_sink, cus);
  }
  return 0;
}
