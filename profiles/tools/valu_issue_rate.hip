// Microbenchmark (gfx950): issue cost of the VALU instruction classes the blend kernels are made of, in SIMD cycles per wave64
// instruction, measured with the shader clock itself (s_memtime) and with the 100 MHz wall clock (s_memrealtime), so that the
// result does not depend on an assumed frequency.  Each wave runs ITERS x 64 independent instructions of one class (16 register
// chains: no dependency stall at >= 2 waves per SIMD); W = 1, 2, 4, 8 waves per SIMD on all 256 CUs.
//   cycles per SIMD-instruction = (t1 - t0 of the slowest wave of a SIMD) / (ITERS * 64 * W)
// Output: one line per (class, W) + a JSON summary at W = 8 (profiles/r5_valu_issue_rate.json is a copy of it).
// Build: hipcc --offload-arch=gfx950 -O3 -o profiles/tools/_bin/valu_issue_rate profiles/tools/valu_issue_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))

enum Cls { FMA, ADD, MUL, MAXF, PKFMA, PKMUL, PKADD, EXP, RCP, SQRT, ADDU, ANDB, LSHL, MADU24, CVTFU, MOV, CNDMASK, CMP, MOVDPP, READLANE, MBCNT,
           BFE, PERM, PSWAP32, PSWAP16, ADDDPP, FMA16, CNDMASK64, NCLS };
static const char* kNames[NCLS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_max_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                   "v_exp_f32", "v_rcp_f32", "v_sqrt_f32", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_mad_u32_u24",
                                   "v_cvt_f32_u32", "v_mov_b32", "v_cndmask_b32", "v_cmp_gt_f32", "v_mov_b32_dpp", "v_readlane_b32",
                                   "v_mbcnt_lo_u32_b32", "v_bfe_u32", "v_perm_b32", "v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_add_f32_dpp", "v_fma_f32 (16 chains)", "v_cndmask_b32_e64 (SGPR mask)"};

template <int C>
__global__ __launch_bounds__(256) void k(unsigned long long* stamps, float* sink, int iters, float seed)
{
    float v0 = seed + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, x = 1.0001f, y = 0.5f;
    f32x2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v1, v2}, p3 = {v3, v0}, px = {1.0001f, 1.0002f}, py = {0.5f, 0.25f};
    unsigned s0 = 0;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        // 16 x 4 = 64 instructions per iteration over four independent chains
#define ONE(INS) asm volatile(INS : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(x), "v"(y));
        if (C == FMA) { REP16(ONE("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")) }
        if (C == ADD) { REP16(ONE("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")) }
        if (C == MUL) { REP16(ONE("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")) }
        if (C == MAXF) { REP16(ONE("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n")) }
        if (C == EXP) { REP16(ONE("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")) }
        if (C == RCP) { REP16(ONE("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")) }
        if (C == SQRT) { REP16(ONE("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")) }
        if (C == ADDU) { REP16(ONE("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")) }
        if (C == ANDB) { REP16(ONE("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n")) }
        if (C == LSHL) { REP16(ONE("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n")) }
        if (C == MADU24) { REP16(ONE("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5\n")) }
        if (C == CVTFU) { REP16(ONE("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n")) }
        if (C == MOV) { REP16(ONE("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n")) }
        if (C == CNDMASK) { REP16(ONE("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")) }
        if (C == CMP) { REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %4\n v_cmp_gt_f32 vcc, %1, %4\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_gt_f32 vcc, %3, %4\n"
                                            : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(x), "v"(y) : "vcc");) }
        if (C == MOVDPP) { REP16(ONE("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                      " v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n")) }
        if (C == READLANE) { REP16(asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %2, 5\n v_readlane_b32 %0, %3, 7\n v_readlane_b32 %0, %4, 9\n"
                                                 : "+s"(s0) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
        if (C == MBCNT) { REP16(ONE("v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_lo_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_lo_u32_b32 %3, -1, %3\n")) }
        if (C == BFE) { REP16(ONE("v_bfe_u32 %0, %0, 1, 8\n v_bfe_u32 %1, %1, 1, 8\n v_bfe_u32 %2, %2, 1, 8\n v_bfe_u32 %3, %3, 1, 8\n")) }
        if (C == PERM) { REP16(ONE("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n")) }
        if (C == PSWAP32) { REP16(ONE("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n")) }
        if (C == PSWAP16) { REP16(ONE("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n")) }
        if (C == ADDDPP) { REP16(ONE("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                                      " v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n")) }
        if (C == FMA16) {      // 16 independent chains per wave: what ONE wave can issue when nothing depends on anything recent
            float u0 = v0, u1 = v1, u2 = v2, u3 = v3, u4 = v0 + 4, u5 = v1 + 4, u6 = v2 + 4, u7 = v3 + 4, u8 = v0 + 8, u9 = v1 + 8, ua = v2 + 8, ub = v3 + 8;
            REP4(asm volatile("v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                              "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                              "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                              "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                              : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7),
                                "+v"(u8), "+v"(u9), "+v"(ua), "+v"(ub) : "v"(x), "v"(y));)
            v0 += u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + u8 + u9 + ua + ub;
        }
        if (C == CNDMASK64) {
            const unsigned long long msk = 0x5555555555555555ull + (unsigned long long)iters;
            REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %6\n v_cndmask_b32_e64 %1, %1, %4, %6\n v_cndmask_b32_e64 %2, %2, %4, %6\n v_cndmask_b32_e64 %3, %3, %4, %6\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(x), "v"(y), "s"(msk));)
        }
#define ONEP(INS) asm volatile(INS : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(px), "v"(py));
        if (C == PKFMA) { REP16(ONEP("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")) }
        if (C == PKMUL) { REP16(ONEP("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")) }
        if (C == PKADD) { REP16(ONEP("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[4 * wave] = t0; stamps[4 * wave + 1] = t1; stamps[4 * wave + 2] = w0; stamps[4 * wave + 3] = w1;
    }
    const float s = v0 + v1 + v2 + v3 + p0.x + p1.y + p2.x + p3.y + (float)s0;
    if (s == 12345.678f) sink[0] = s;
}

struct Res { double cyc_clock, cyc_wall_2p4, ghz; };

template <int C>
Res run(int w, unsigned long long* d_st, float* d_sink)
{
    const int blocks = 256 * w, iters = 2000, nw = blocks * 4;       // 4 waves per workgroup: one per SIMD
    k<C><<<blocks, 256>>>(d_st, d_sink, 10, 1.0f);
    hipDeviceSynchronize();
    k<C><<<blocks, 256>>>(d_st, d_sink, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(4 * (size_t)nw);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    // every wave ran beside W - 1 others on its SIMD for (almost) its whole life when all start together: per-wave span
    std::vector<double> span(nw), wall(nw);
    for (int i = 0; i < nw; ++i) { span[i] = (double)(st[4 * i + 1] - st[4 * i]); wall[i] = (double)(st[4 * i + 3] - st[4 * i + 2]); }
    std::sort(span.begin(), span.end()); std::sort(wall.begin(), wall.end());
    const double med = span[nw / 2], medw = wall[nw / 2];
    Res r;
    r.cyc_clock = med / ((double)iters * 64.0 * w);
    r.ghz = med / (medw * 10.0);                                      // wall clock: 100 MHz -> 10 ns per tick
    r.cyc_wall_2p4 = medw * 10e-9 * 2.4e9 / ((double)iters * 64.0 * w);
    return r;
}

template <int C>
void all(unsigned long long* d_st, float* d_sink, double* out8, double* ghz8)
{
    for (int w = 1; w <= 8; w *= 2) {
        Res r = run<C>(w, d_st, d_sink);
        printf("%-20s waves/SIMD=%d: %6.2f cycles per SIMD-instruction by s_memtime (counter at %.3f GHz), %6.2f at a nominal 2.4 GHz by the wall clock\n",
               kNames[C], w, r.cyc_clock, r.ghz, r.cyc_wall_2p4);
        if (w == 8) { out8[C] = r.cyc_wall_2p4; ghz8[C] = r.ghz; }
    }
}

int main()
{
    unsigned long long* d_st; float* d_sink;
    hipMalloc(&d_st, 256 * 8 * 4 * 4 * 8); hipMalloc(&d_sink, 64);
    double c8[NCLS], g8[NCLS];
    all<FMA>(d_st, d_sink, c8, g8); all<ADD>(d_st, d_sink, c8, g8); all<MUL>(d_st, d_sink, c8, g8); all<MAXF>(d_st, d_sink, c8, g8);
    all<PKFMA>(d_st, d_sink, c8, g8); all<PKMUL>(d_st, d_sink, c8, g8); all<PKADD>(d_st, d_sink, c8, g8);
    all<EXP>(d_st, d_sink, c8, g8); all<RCP>(d_st, d_sink, c8, g8); all<SQRT>(d_st, d_sink, c8, g8);
    all<ADDU>(d_st, d_sink, c8, g8); all<ANDB>(d_st, d_sink, c8, g8); all<LSHL>(d_st, d_sink, c8, g8); all<MADU24>(d_st, d_sink, c8, g8);
    all<CVTFU>(d_st, d_sink, c8, g8); all<MOV>(d_st, d_sink, c8, g8); all<CNDMASK>(d_st, d_sink, c8, g8); all<CMP>(d_st, d_sink, c8, g8);
    all<MOVDPP>(d_st, d_sink, c8, g8); all<READLANE>(d_st, d_sink, c8, g8); all<MBCNT>(d_st, d_sink, c8, g8); all<BFE>(d_st, d_sink, c8, g8);
    all<PERM>(d_st, d_sink, c8, g8); all<PSWAP32>(d_st, d_sink, c8, g8); all<PSWAP16>(d_st, d_sink, c8, g8); all<ADDDPP>(d_st, d_sink, c8, g8);
    all<FMA16>(d_st, d_sink, c8, g8); all<CNDMASK64>(d_st, d_sink, c8, g8);
    printf("JSON {\"what\": \"SIMD cycles per wave64 instruction at 8 waves per SIMD, wall time x 2.4 GHz (profiles/tools/valu_issue_rate.hip)\", \"cycles_at_2.4GHz\": {");
    for (int c = 0; c < NCLS; ++c) printf("%s\"%s\": %.3f", c ? ", " : "", kNames[c], c8[c]);
    printf("}, \"counter_ghz\": {");
    for (int c = 0; c < NCLS; ++c) printf("%s\"%s\": %.3f", c ? ", " : "", kNames[c], g8[c]);
    printf("}}\n");
    return 0;
}
