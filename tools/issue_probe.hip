// issue_probe.hip -- register-only instruction-issue microbenchmark for gfx950 (VERDICT r5 "next" #2).
//
// What it answers: how many shader clocks does one SIMD need per wave64 instruction of the kinds the greedy step is made
// of (integer VALU, DPP row ops, v_readlane / v_writelane, SALU), at 1 .. 8 resident waves per SIMD -- with NO memory
// instruction in the timed loop, so that the answer is the issue rate alone.  DESIGN.md 5 prices the step's 161 VALU +
// 118 SALU against it (bench.py: roofline.issue).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/issue_probe tools/issue_probe.hip && tools/issue_probe
//
// Method.  One workgroup per CU for <= 4 waves per SIMD (256 * w threads; the dynamic LDS request keeps a second workgroup
// off the CU), two per CU above that (128 * w threads each).  Every wave runs `iters` trips of a loop whose body is
// kUnroll copies of one instruction (or 152-176 instructions of the step's mix) over eight independent registers, brackets it with s_memtime
// (the shader clock) and s_memrealtime (the constant 100 MHz clock), and writes both.  Reported per probe:
//   clk/instr/SIMD = median over waves of (s_memtime delta) / (instructions per wave x waves per SIMD)
// i.e. the clocks of ONE SIMD's issue that one wave64 instruction takes when w waves compete for it.  The shader clock in
// MHz (s_memtime ticks per s_memrealtime tick x 100) is printed so that the figure can be turned into time.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

enum Kind {
    kVAdd, kVAddDep, kVMad24, kVMin, kVXad, kVDppRow, kVDppDep, kVReadlane, kVWritelane, kVReadWrite, kVCndmask, kVPerm, kVBfe,
    kVCndmaskSgpr, kVCmp, kVCmpCnd, kVCmpCndSgpr, kVMinMaxPair, kVLshlOr, kVMbcnt, kVAdd3, kSAdd, kSAddDep, kSBcnt, kSCselect, kSAnd64, kMix, kMixValuOnly, kMixSaluOnly, kNKinds
};
static const char *kNames[kNKinds] = {
    "v_add_u32 (8 independent)", "v_add_u32 (dependent chain)", "v_mad_i32_i24", "v_min_i32", "v_xad_u32", "v_add_u32 dpp row_shr:1 (8 independent)",
    "v_min_i32 dpp row_shr (dependent chain)", "v_readlane_b32 -> SGPR", "v_writelane_b32", "v_readlane + v_writelane pairs", "v_cndmask_b32 (vcc)", "v_perm_b32", "v_bfe_i32",
    "v_cndmask_b32 (mask in an SGPR pair)", "v_cmp_lt_u32 -> vcc", "v_cmp_lt_u32 vcc + v_cndmask_b32 vcc pairs", "v_cmp_lt_u32 s[..] + v_cndmask_b32 s[..] pairs",
    "v_min_i32 + v_max_i32 pairs (a select-free compare-exchange)", "v_lshl_or_b32", "v_mbcnt_lo_u32_b32", "v_add3_u32",
    "s_add_u32 (8 independent)", "s_add_u32 (dependent chain)", "s_bcnt1_i32_b64", "s_cmp + s_cselect pairs", "s_and_b64 (vcc, exec)",
    "greedy-step mix 161 VALU : 118 SALU (interleaved)", "  its 161 VALU alone", "  its 118 SALU alone"};

constexpr int kUnroll = 256;

// eight vector and eight scalar registers carried through the loop; "+v"/"+s" keeps every instruction alive
#define V8 "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
#define S8 "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int KIND> __global__ void probe_kernel(unsigned iters, unsigned long long *__restrict__ out, unsigned *__restrict__ sink)
{
    extern __shared__ unsigned dyn[];
    unsigned v0 = threadIdx.x, v1 = v0 * 3u + 1u, v2 = v0 ^ 0x55u, v3 = v0 + 7u, v4 = v0 * 5u, v5 = v0 + 11u, v6 = v0 ^ 0x33u, v7 = v0 * 9u;
    unsigned s0 = blockIdx.x + 1u, s1 = s0 * 3u, s2 = s0 ^ 5u, s3 = s0 + 7u, s4 = s0 * 5u, s5 = s0 + 11u, s6 = s0 ^ 3u, s7 = s0 * 9u;
    // (force them into SGPRs)
    s0 = __builtin_amdgcn_readfirstlane(s0); s1 = __builtin_amdgcn_readfirstlane(s1); s2 = __builtin_amdgcn_readfirstlane(s2); s3 = __builtin_amdgcn_readfirstlane(s3);
    s4 = __builtin_amdgcn_readfirstlane(s4); s5 = __builtin_amdgcn_readfirstlane(s5); s6 = __builtin_amdgcn_readfirstlane(s6); s7 = __builtin_amdgcn_readfirstlane(s7);
    __syncthreads();   // the workgroup's waves start together
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (unsigned i = 0; i < iters; i++) {
        if constexpr (KIND == kVAdd) {
            asm volatile(REP32("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n"
                              "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0\n") : V8);
        } else if constexpr (KIND == kVAddDep) {
            asm volatile(REP32(REP8("v_add_u32 %0, %0, %1\n")) : V8);
        } else if constexpr (KIND == kVMad24) {
            asm volatile(REP32("v_mad_i32_i24 %0, %1, %2, %0\n v_mad_i32_i24 %1, %2, %3, %1\n v_mad_i32_i24 %2, %3, %4, %2\n v_mad_i32_i24 %3, %4, %5, %3\n"
                              "v_mad_i32_i24 %4, %5, %6, %4\n v_mad_i32_i24 %5, %6, %7, %5\n v_mad_i32_i24 %6, %7, %0, %6\n v_mad_i32_i24 %7, %0, %1, %7\n") : V8);
        } else if constexpr (KIND == kVMin) {
            asm volatile(REP32("v_min_i32 %0, %0, %1\n v_min_i32 %1, %1, %2\n v_min_i32 %2, %2, %3\n v_min_i32 %3, %3, %4\n"
                              "v_min_i32 %4, %4, %5\n v_min_i32 %5, %5, %6\n v_min_i32 %6, %6, %7\n v_min_i32 %7, %7, %0\n") : V8);
        } else if constexpr (KIND == kVXad) {
            asm volatile(REP32("v_xad_u32 %0, %0, %1, -1\n v_xad_u32 %1, %1, %2, -1\n v_xad_u32 %2, %2, %3, -1\n v_xad_u32 %3, %3, %4, -1\n"
                              "v_xad_u32 %4, %4, %5, -1\n v_xad_u32 %5, %5, %6, -1\n v_xad_u32 %6, %6, %7, -1\n v_xad_u32 %7, %7, %0, -1\n") : V8);
        } else if constexpr (KIND == kVDppRow) {
            // (each reads through DPP a register written eight instructions earlier: no s_nop needed, no dependent stall)
            asm volatile(REP32("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_u32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %6, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_u32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %0, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n") : V8);
        } else if constexpr (KIND == kVDppDep) {
            // the shape of a wave minimum: each DPP reads what the previous instruction wrote (two wait states: s_nop 1)
            asm volatile(REP32(REP4("v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n")) : V8);
        } else if constexpr (KIND == kVReadlane) {
            asm volatile(REP32("v_readlane_b32 %8, %0, 1\n v_readlane_b32 %9, %1, 2\n v_readlane_b32 %10, %2, 3\n v_readlane_b32 %11, %3, 4\n"
                              "v_readlane_b32 %12, %4, 5\n v_readlane_b32 %13, %5, 6\n v_readlane_b32 %14, %6, 7\n v_readlane_b32 %15, %7, 8\n") : V8, S8);
        } else if constexpr (KIND == kVWritelane) {
            asm volatile(REP32("v_writelane_b32 %0, %8, 1\n v_writelane_b32 %1, %9, 2\n v_writelane_b32 %2, %10, 3\n v_writelane_b32 %3, %11, 4\n"
                              "v_writelane_b32 %4, %12, 5\n v_writelane_b32 %5, %13, 6\n v_writelane_b32 %6, %14, 7\n v_writelane_b32 %7, %15, 8\n") : V8, S8);
        } else if constexpr (KIND == kVReadWrite) {
            // SGPR written by a VALU, read by the next VALU as a lane-select-free operand: the apply's round trips
            asm volatile(REP32("v_readlane_b32 %8, %0, 1\n v_readlane_b32 %9, %1, 2\n v_readlane_b32 %10, %2, 3\n v_readlane_b32 %11, %3, 4\n"
                              "v_writelane_b32 %4, %8, 5\n v_writelane_b32 %5, %9, 6\n v_writelane_b32 %6, %10, 7\n v_writelane_b32 %7, %11, 8\n") : V8, S8);
        } else if constexpr (KIND == kVCndmask) {
            asm volatile(REP32("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n") : V8 : : "vcc");
        } else if constexpr (KIND == kVPerm) {
            asm volatile(REP32("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %4, %5\n"
                              "v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %5, %5, %6, %7\n v_perm_b32 %6, %6, %7, %0\n v_perm_b32 %7, %7, %0, %1\n") : V8);
        } else if constexpr (KIND == kVBfe) {
            asm volatile(REP32("v_bfe_i32 %0, %0, %1, 1\n v_bfe_i32 %1, %1, %2, 1\n v_bfe_i32 %2, %2, %3, 1\n v_bfe_i32 %3, %3, %4, 1\n"
                              "v_bfe_i32 %4, %4, %5, 1\n v_bfe_i32 %5, %5, %6, 1\n v_bfe_i32 %6, %6, %7, 1\n v_bfe_i32 %7, %7, %0, 1\n") : V8);
        } else if constexpr (KIND == kVCndmaskSgpr) {
            asm volatile(REP32("v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %1, %1, %2, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]\n"
                               "v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %7, %7, %0, s[20:21]\n") : V8 : : "s20", "s21");
        } else if constexpr (KIND == kVCmp) {
            asm volatile(REP32("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %4\n"
                               "v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %7, %0\n") : V8 : : "vcc");
        } else if constexpr (KIND == kVCmpCnd) {
            asm volatile(REP32("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n"
                               "v_cmp_lt_u32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_u32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc\n") : V8 : : "vcc");
        } else if constexpr (KIND == kVCmpCndSgpr) {
            asm volatile(REP32("v_cmp_lt_u32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cmp_lt_u32 s[22:23], %2, %3\n v_cndmask_b32 %2, %2, %3, s[22:23]\n"
                               "v_cmp_lt_u32 s[20:21], %4, %5\n v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cmp_lt_u32 s[22:23], %6, %7\n v_cndmask_b32 %6, %6, %7, s[22:23]\n") : V8 : : "s20", "s21", "s22", "s23");
        } else if constexpr (KIND == kVMinMaxPair) {
            asm volatile(REP32("v_min_i32 %0, %1, %2\n v_max_i32 %1, %1, %2\n v_min_i32 %2, %3, %4\n v_max_i32 %3, %3, %4\n"
                               "v_min_i32 %4, %5, %6\n v_max_i32 %5, %5, %6\n v_min_i32 %6, %7, %0\n v_max_i32 %7, %7, %0\n") : V8);
        } else if constexpr (KIND == kVLshlOr) {
            asm volatile(REP32("v_lshl_or_b32 %0, %0, 8, %1\n v_lshl_or_b32 %1, %1, 8, %2\n v_lshl_or_b32 %2, %2, 8, %3\n v_lshl_or_b32 %3, %3, 8, %4\n"
                               "v_lshl_or_b32 %4, %4, 8, %5\n v_lshl_or_b32 %5, %5, 8, %6\n v_lshl_or_b32 %6, %6, 8, %7\n v_lshl_or_b32 %7, %7, 8, %0\n") : V8);
        } else if constexpr (KIND == kVMbcnt) {
            asm volatile(REP32("v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_lo_u32_b32 %1, %9, %1\n v_mbcnt_lo_u32_b32 %2, %10, %2\n v_mbcnt_lo_u32_b32 %3, %11, %3\n"
                               "v_mbcnt_lo_u32_b32 %4, %12, %4\n v_mbcnt_lo_u32_b32 %5, %13, %5\n v_mbcnt_lo_u32_b32 %6, %14, %6\n v_mbcnt_lo_u32_b32 %7, %15, %7\n") : V8, S8);
        } else if constexpr (KIND == kVAdd3) {
            asm volatile(REP32("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %3, %3, %4, %5\n"
                               "v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %0\n v_add3_u32 %7, %7, %0, %1\n") : V8);
        } else if constexpr (KIND == kSAnd64) {
            asm volatile(REP32("s_and_b64 s[20:21], vcc, exec\n s_and_b64 s[22:23], vcc, exec\n s_and_b64 s[24:25], vcc, exec\n s_and_b64 s[26:27], vcc, exec\n"
                               "s_and_b64 s[20:21], vcc, exec\n s_and_b64 s[22:23], vcc, exec\n s_and_b64 s[24:25], vcc, exec\n s_and_b64 s[26:27], vcc, exec\n") : S8 : : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if constexpr (KIND == kSAdd) {
            asm volatile(REP32("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %4\n"
                              "s_add_u32 %4, %4, %5\n s_add_u32 %5, %5, %6\n s_add_u32 %6, %6, %7\n s_add_u32 %7, %7, %0\n") : S8 : : "scc");
        } else if constexpr (KIND == kSAddDep) {
            asm volatile(REP32(REP8("s_add_u32 %0, %0, %1\n")) : S8 : : "scc");
        } else if constexpr (KIND == kSBcnt) {
            asm volatile(REP32("s_bcnt1_i32_b64 %0, vcc\n s_bcnt1_i32_b64 %1, vcc\n s_bcnt1_i32_b64 %2, vcc\n s_bcnt1_i32_b64 %3, vcc\n"
                              "s_bcnt1_i32_b64 %4, vcc\n s_bcnt1_i32_b64 %5, vcc\n s_bcnt1_i32_b64 %6, vcc\n s_bcnt1_i32_b64 %7, vcc\n") : S8 : : "scc", "vcc");
        } else if constexpr (KIND == kSCselect) {
            asm volatile(REP32("s_cmp_lt_u32 %0, %1\n s_cselect_b32 %2, %3, %4\n s_cmp_lt_u32 %4, %5\n s_cselect_b32 %6, %7, %0\n"
                              "s_cmp_lt_u32 %1, %2\n s_cselect_b32 %3, %4, %5\n s_cmp_lt_u32 %5, %6\n s_cselect_b32 %7, %0, %1\n") : S8 : : "scc");
        } else if constexpr (KIND == kMix) {
            // the step's instruction mix, 161 VALU : 118 SALU ~ 11 : 8, interleaved the way a compiled step is (runs of two or
            // three vector instructions with scalar ones between): 22 VALU + 16 SALU per copy, two copies = 44 + 32 (kMixN)
#define IIV_MIX_BODY                                                                                                          \
    "v_add_u32 %0, %0, %1\n v_mad_i32_i24 %1, %2, %3, %1\n s_add_u32 %8, %8, %9\n v_min_i32 %2, %2, %3\n v_xad_u32 %3, %3, %4, -1\n"       \
    "s_lshl_b32 %9, %10, 1\n s_and_b32 %10, %11, %12\n v_add_u32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_bfe_i32 %5, %5, %6, 1\n"  \
    "v_min_i32 %6, %6, %7\n s_cmp_lt_u32 %11, %12\n s_cselect_b32 %12, %13, %14\n v_add_u32 %7, %7, %0\n v_cndmask_b32 %0, %0, %1, vcc\n"  \
    "s_add_u32 %13, %13, %14\n v_readlane_b32 %14, %2, 3\n s_bfe_u32 %15, %8, 0x80008\n v_mad_i32_i24 %1, %2, %3, %1\n v_min_i32 %2, %2, %3\n" \
    "s_lshr_b32 %8, %9, 5\n v_writelane_b32 %3, %15, 2\n v_add_u32 %4, %4, %5\n s_or_b32 %9, %9, %10\n s_bcnt1_i32_b64 %10, vcc\n"        \
    "v_perm_b32 %5, %5, %6, %7\n v_add_u32 %6, %6, %7\n s_add_u32 %11, %11, %12\n v_min_i32_dpp %7, %0, %7 row_shr:2 row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32 %0, %0, %1\n s_and_b32 %12, %13, %14\n s_sub_u32 %13, %13, %15\n v_xad_u32 %1, %1, %2, -1\n v_lshlrev_b32 %2, 1, %3\n"      \
    "s_lshl_b32 %14, %8, 2\n v_and_b32 %3, %3, %4\n v_add_u32 %4, %4, %5\n s_add_u32 %15, %15, %8\n v_min_i32 %5, %5, %6\n"
            asm volatile(REP4(IIV_MIX_BODY) : V8, S8 : : "scc", "vcc");
        } else if constexpr (KIND == kMixValuOnly) {
#define IIV_MIXV_BODY                                                                                                          \
    "v_add_u32 %0, %0, %1\n v_mad_i32_i24 %1, %2, %3, %1\n v_min_i32 %2, %2, %3\n v_xad_u32 %3, %3, %4, -1\n"                              \
    "v_add_u32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_bfe_i32 %5, %5, %6, 1\n"                                        \
    "v_min_i32 %6, %6, %7\n v_add_u32 %7, %7, %0\n v_cndmask_b32 %0, %0, %1, vcc\n"                                                    \
    "v_readlane_b32 %14, %2, 3\n v_mad_i32_i24 %1, %2, %3, %1\n v_min_i32 %2, %2, %3\n"                                                \
    "v_writelane_b32 %3, %15, 2\n v_add_u32 %4, %4, %5\n"                                                                              \
    "v_perm_b32 %5, %5, %6, %7\n v_add_u32 %6, %6, %7\n v_min_i32_dpp %7, %0, %7 row_shr:2 row_mask:0xf bank_mask:0xf\n"               \
    "v_add_u32 %0, %0, %1\n v_xad_u32 %1, %1, %2, -1\n v_lshlrev_b32 %2, 1, %3\n"                                                      \
    "v_and_b32 %3, %3, %4\n v_add_u32 %4, %4, %5\n"
            asm volatile(REP4(IIV_MIXV_BODY) REP4(IIV_MIXV_BODY) : V8, S8 : : "scc", "vcc");
        } else if constexpr (KIND == kMixSaluOnly) {
#define IIV_MIXS_BODY                                                                                                          \
    "s_add_u32 %8, %8, %9\n s_lshl_b32 %9, %10, 1\n s_and_b32 %10, %11, %12\n s_cmp_lt_u32 %11, %12\n s_cselect_b32 %12, %13, %14\n"      \
    "s_add_u32 %13, %13, %14\n s_bfe_u32 %15, %8, 0x80008\n s_lshr_b32 %8, %9, 5\n s_or_b32 %9, %9, %10\n s_bcnt1_i32_b64 %10, vcc\n"    \
    "s_add_u32 %11, %11, %12\n s_and_b32 %12, %13, %14\n s_sub_u32 %13, %13, %15\n s_lshl_b32 %14, %8, 2\n s_add_u32 %15, %15, %8\n s_xor_b32 %14, %14, %9\n"
            asm volatile(REP8(IIV_MIXS_BODY) : V8, S8 : : "scc", "vcc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned x = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
    if (x == 0x12345678u) sink[0] = x + dyn[0];   // (never true in practice: keeps the registers live)
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}

// instructions per loop trip (what the clocks are divided by); the loop's own s_add / s_cmp / s_cbranch are 3 scalar
// instructions and one taken branch per trip and are NOT counted (~36 clocks of one wave against >= 128 x 4: the first
// version of this probe, 32 instructions per trip, read 5.13 clocks per v_add_u32 for one wave alone)
static int instrs_per_trip(int kind)
{
    switch (kind) {
    case kVDppDep: return 256;         // 256 DPP instructions (+ 256 s_nop 1, which cost issue slots too: see the name)
    case kMix: return 4 * 38;
    case kMixValuOnly: return 8 * 22;
    case kMixSaluOnly: return 8 * 16;
    case kVCmpCnd: return 256;
    case kVCmpCndSgpr: return 256;
    default: return kUnroll;
    }
}

template <int KIND> static void run_kind(int waves_per_simd, unsigned iters, int n_cu, unsigned long long *d_out, unsigned *d_sink, double &clk_per_instr, double &mhz, double &ms)
{
    const int wg_per_cu = waves_per_simd <= 4 ? 1 : 2;
    const int threads = 256 * waves_per_simd / wg_per_cu;
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : 70 * 1024;
    const int grid = n_cu * wg_per_cu;
    CHECK(hipFuncSetAttribute((const void *)probe_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe_kernel<KIND>), dim3(grid), dim3(threads), lds, 0, iters / 8, d_out, d_sink);   // warm-up (clocks ramp)
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_kernel<KIND>), dim3(grid), dim3(threads), lds, 0, iters, d_out, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float t = 0;
    CHECK(hipEventElapsedTime(&t, e0, e1));
    ms = t;
    const size_t n_waves = (size_t)grid * (threads / 64);
    std::vector<unsigned long long> h(2 * n_waves);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> clk(n_waves), rt(n_waves);
    for (size_t i = 0; i < n_waves; i++) {
        clk[i] = (double)h[2 * i];
        rt[i] = (double)h[2 * i + 1];
    }
    std::sort(clk.begin(), clk.end());
    std::sort(rt.begin(), rt.end());
    const double c = clk[n_waves / 2], r = rt[n_waves / 2];
    clk_per_instr = c / ((double)iters * instrs_per_trip(KIND) * waves_per_simd);
    mhz = r > 0 ? c / r * 100.0 : 0.0;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
    unsigned iters = argc > 1 ? (unsigned)atoi(argv[1]) : 4000u;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long *d_out;
    unsigned *d_sink;
    CHECK(hipMalloc(&d_out, sizeof(unsigned long long) * 2 * 32 * (size_t)n_cu * 2));
    CHECK(hipMalloc(&d_sink, 64));
    printf("# issue_probe: %s, %d CUs, %u loop trips per wave; clocks of one SIMD per wave64 instruction (s_memtime, median over waves)\n", prop.gcnArchName, n_cu, iters);
    printf("# waves per SIMD:                                      ");
    const int ws[] = {1, 2, 3, 4, 6, 7, 8};
    for (int w : ws) printf("%8d", w);
    printf("   shader MHz\n");
    auto row = [&](int kind, auto tag) {
        printf("%-56s", kNames[kind]);
        double mhz = 0, ms = 0;
        for (int w : ws) {
            double c;
            run_kind<decltype(tag)::value>(w, iters, n_cu, d_out, d_sink, c, mhz, ms);
            printf("%8.2f", c);
        }
        printf("   %8.0f\n", mhz);
        fflush(stdout);
    };
#define ROW(K) row(K, std::integral_constant<int, K>{})
    ROW(kVAdd); ROW(kVAdd); ROW(kVAddDep); ROW(kVMad24); ROW(kVMin); ROW(kVXad); ROW(kVDppRow); ROW(kVDppDep); ROW(kVReadlane); ROW(kVWritelane);
    ROW(kVReadWrite); ROW(kVCndmask); ROW(kVCndmaskSgpr); ROW(kVCmp); ROW(kVCmpCnd); ROW(kVCmpCndSgpr); ROW(kVMinMaxPair); ROW(kVLshlOr); ROW(kVMbcnt); ROW(kVAdd3);
    ROW(kVPerm); ROW(kVBfe); ROW(kSAdd); ROW(kSAddDep); ROW(kSBcnt); ROW(kSCselect); ROW(kSAnd64);
    ROW(kMix); ROW(kMixValuOnly); ROW(kMixSaluOnly);
    printf("# (mix rows: clocks per instruction of the row's own count -- 152 / 176 / 128 per trip; a step of 161 VALU + 118 SALU +\n"
           "#  28 other at w waves per SIMD needs >= 279 x w x [mix row] clocks of its SIMD; the first row is a warm-up: clocks ramp)\n");
    return 0;
}
