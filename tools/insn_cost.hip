// Microbenchmark: issue cost of single VALU instructions (four independent copies of each, unrolled, explicit
// registers), at 1 / 3 waves per SIMD.  Prints ns per wave-instruction per SIMD.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REGS "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27", \
             "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","vcc"
#define M(N, A, B, C, D) if (MODE == N) { if (PAD) asm volatile(".p2align 3\n s_nop 0\n" REP64(A "\n" B "\n" C "\n" D "\n") ::: REGS); else asm volatile(".p2align 3\n" REP64(A "\n" B "\n" C "\n" D "\n") ::: REGS); }

template<int MODE, int PAD>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, unsigned long long* clk)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    asm volatile("s_movk_i32 s50, 0x300\n s_movk_i32 s51, 0xc00\n v_mov_b32 v24, 3\n v_mov_b32 v25, 0x300" ::: "s50", "s51", "v24", "v25");
    for (int it = 0; it < iters; it++) {
        M(0, "v_and_b32 v4, v5, v6", "v_and_b32 v8, v9, v10", "v_and_b32 v12, v13, v14", "v_and_b32 v16, v17, v18")
        M(1, "v_bitop3_b32 v4, v5, v6, v7 bitop3:0x96", "v_bitop3_b32 v8, v9, v10, v11 bitop3:0x96", "v_bitop3_b32 v12, v13, v14, v15 bitop3:0x96", "v_bitop3_b32 v16, v17, v18, v19 bitop3:0x96")
        M(2, "v_bfe_u32 v4, v5, 6, 2", "v_bfe_u32 v8, v9, 8, 2", "v_bfe_u32 v12, v13, 10, 2", "v_bfe_u32 v16, v17, 12, 2")
        M(3, "v_lshl_or_b32 v4, v5, 8, v6", "v_lshl_or_b32 v8, v9, 8, v10", "v_lshl_or_b32 v12, v13, 8, v14", "v_lshl_or_b32 v16, v17, 8, v18")
        M(4, "v_and_or_b32 v4, v5, s50, v6", "v_and_or_b32 v8, v9, s51, v10", "v_and_or_b32 v12, v13, s50, v14", "v_and_or_b32 v16, v17, s51, v18")
        M(5, "v_and_or_b32 v4, v5, v25, v6", "v_and_or_b32 v8, v9, v25, v10", "v_and_or_b32 v12, v13, v25, v14", "v_and_or_b32 v16, v17, v25, v18")
        M(6, "v_and_b32_sdwa v4, v5, v24 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD", "v_and_b32_sdwa v8, v9, v24 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD",
             "v_and_b32_sdwa v12, v13, v24 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:DWORD", "v_and_b32_sdwa v16, v17, v24 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD")
        M(7, "v_and_b32_sdwa v4, v5, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD", "v_and_b32_sdwa v8, v9, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD",
             "v_and_b32_sdwa v12, v13, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD", "v_and_b32_sdwa v16, v17, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD")
        M(8, "v_lshrrev_b32 v4, 2, v5", "v_lshrrev_b32 v8, 4, v9", "v_lshrrev_b32 v12, 6, v13", "v_lshrrev_b32 v16, 2, v17")
        M(9, "v_add_u32 v4, v5, v6", "v_add_u32 v8, v9, v10", "v_add_u32 v12, v13, v14", "v_add_u32 v16, v17, v18")
        M(10, "v_add_co_u32_e64 v4, s[40:41], v5, v6", "v_add_co_u32_e64 v8, s[42:43], v9, v10", "v_add_co_u32_e64 v12, s[44:45], v13, v14", "v_add_co_u32_e64 v16, s[46:47], v17, v18")
        M(11, "v_addc_co_u32_e64 v4, s[40:41], v5, v6, s[48:49]", "v_addc_co_u32_e64 v8, s[42:43], v9, v10, s[48:49]", "v_addc_co_u32_e64 v12, s[44:45], v13, v14, s[48:49]", "v_addc_co_u32_e64 v16, s[46:47], v17, v18, s[48:49]")
        M(12, "v_cmp_gt_u32 vcc, v5, v6", "v_cmp_gt_u32 vcc, v9, v10", "v_cmp_gt_u32 vcc, v13, v14", "v_cmp_gt_u32 vcc, v17, v18")
        M(13, "v_cndmask_b32 v4, v5, v6, vcc", "v_cndmask_b32 v8, v9, v10, vcc", "v_cndmask_b32 v12, v13, v14, vcc", "v_cndmask_b32 v16, v17, v18, vcc")
        M(14, "v_pk_add_u16 v4, v5, v6", "v_pk_add_u16 v8, v9, v10", "v_pk_add_u16 v12, v13, v14", "v_pk_add_u16 v16, v17, v18")
        M(15, "v_perm_b32 v4, v5, v6, v7", "v_perm_b32 v8, v9, v10, v11", "v_perm_b32 v12, v13, v14, v15", "v_perm_b32 v16, v17, v18, v19")
        M(16, "v_alignbit_b32 v4, v5, v6, 16", "v_alignbit_b32 v8, v9, v10, 16", "v_alignbit_b32 v12, v13, v14, 16", "v_alignbit_b32 v16, v17, v18, 16")
        M(17, "v_bcnt_u32_b32 v4, v5, v4", "v_bcnt_u32_b32 v8, v9, v8", "v_bcnt_u32_b32 v12, v13, v12", "v_bcnt_u32_b32 v16, v17, v16")
        M(18, "v_and_b32 v4, 0xffff, v6", "v_and_b32 v8, 0xffff, v10", "v_and_b32 v12, 0xffff, v14", "v_and_b32 v16, 0xffff, v18")
        M(19, "v_add_u32_sdwa v4, v5, v6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3", "v_add_u32_sdwa v8, v9, v10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3",
              "v_add_u32_sdwa v12, v13, v14 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3", "v_add_u32_sdwa v16, v17, v18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3")
        M(20, "v_sub_u32 v4, v5, v6", "v_xor_b32 v8, v9, v10", "v_or_b32 v12, v13, v14", "v_lshlrev_b32 v16, 3, v18")
        M(21, "v_and_b32 v4, v5, v6", "v_add_co_u32_e64 v8, s[42:43], v9, v10", "v_bitop3_b32 v12, v13, v14, v15 bitop3:0x96", "v_and_b32_sdwa v16, v17, v24 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD")
        M(22, "v_xad_u32 v4, v5, v6, v7", "v_xad_u32 v8, v9, v10, v11", "v_xad_u32 v12, v13, v14, v15", "v_xad_u32 v16, v17, v18, v19")
        M(23, "v_and_b32 v4, s50, v6", "v_and_b32 v8, s51, v10", "v_and_b32 v12, s50, v14", "v_and_b32 v16, s51, v18")
        M(24, "v_mad_u32_u24 v4, v5, v6, v7", "v_mad_u32_u24 v8, v9, v10, v11", "v_mad_u32_u24 v12, v13, v14, v15", "v_mad_u32_u24 v16, v17, v18, v19")
        M(25, "v_or3_b32 v4, v5, v6, v7", "v_or3_b32 v8, v9, v10, v11", "v_or3_b32 v12, v13, v14, v15", "v_or3_b32 v16, v17, v18, v19")
        M(26, "v_bitop3_b32 v4, v5, v6, s50 bitop3:0x96", "v_bitop3_b32 v8, v9, v10, s51 bitop3:0x96", "v_bitop3_b32 v12, v13, v14, s50 bitop3:0x96", "v_bitop3_b32 v16, v17, v18, s51 bitop3:0x96")
        M(27, "v_bitop3_b32 v4, v5, v6, v6 bitop3:0x96", "v_bitop3_b32 v8, v9, v10, v10 bitop3:0x96", "v_bitop3_b32 v12, v13, v14, v14 bitop3:0x96", "v_bitop3_b32 v16, v17, v18, v18 bitop3:0x96")
        M(28, "v_xor_b32 v4, v5, v6", "v_xor_b32 v8, v9, v10", "v_xor_b32 v12, v13, v14", "v_xor_b32 v16, v17, v18")
        M(29, "v_or_b32 v4, v5, v6", "v_or_b32 v8, v9, v10", "v_or_b32 v12, v13, v14", "v_or_b32 v16, v17, v18")
        M(30, "v_sub_u32 v4, v5, v6", "v_sub_u32 v8, v9, v10", "v_sub_u32 v12, v13, v14", "v_sub_u32 v16, v17, v18")
        M(31, "v_lshlrev_b32 v4, 3, v6", "v_lshlrev_b32 v8, 3, v10", "v_lshlrev_b32 v12, 3, v14", "v_lshlrev_b32 v16, 3, v18")
        M(32, "v_and_b32 v4, v5, v6", "v_add_u32 v8, v9, v10", "v_and_b32 v12, v13, v14", "v_add_u32 v16, v17, v18")
        M(33, "v_and_b32 v4, v5, v6", "v_and_b32 v8, v9, v10", "v_add_u32 v12, v13, v14", "v_add_u32 v16, v17, v18")
        M(34, "v_and_b32 v4, v5, v6", "v_bitop3_b32 v8, v9, v10, v11 bitop3:0x96", "v_and_b32 v12, v13, v14", "v_bitop3_b32 v16, v17, v18, v19 bitop3:0x96")
        M(35, "v_and_b32 v4, v5, v6", "v_and_b32 v8, v9, v10", "v_bitop3_b32 v12, v13, v14, v15 bitop3:0x96", "v_bitop3_b32 v16, v17, v18, v19 bitop3:0x96")
        M(36, "v_and_b32 v4, v5, v6", "v_lshrrev_b32 v8, 4, v9", "v_add_u32 v12, v13, v14", "v_bitop3_b32 v16, v17, v18, v19 bitop3:0x96")
        M(37, "v_and_b32 v4, v5, v6\n v_and_b32 v8, v9, v10", "v_add_co_u32_e64 v12, s[42:43], v13, v14\n v_addc_co_u32_e64 v16, s[44:45], v17, v18, s[48:49]", "v_bitop3_b32 v20, v21, v22, v23 bitop3:0x96\n v_bitop3_b32 v4, v5, v6, v7 bitop3:0x96", "v_and_b32_sdwa v16, v17, v24 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD")
        M(38, "v_and_b32 v4, v5, v6\n v_and_b32 v8, v9, v10", "v_add_u32 v12, v13, v14\n v_add_u32 v16, v17, v18", "v_bitop3_b32 v20, v21, v22, v23 bitop3:0x96\n v_bitop3_b32 v4, v5, v6, v7 bitop3:0x96", "v_and_b32 v16, v17, v24")
    }
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = threadIdx.x;
}

template<int MODE> int run(const char* name)
{
    for (int pad : {0, 1}) {
        const int wps = 3;
        const int blocks = 256 * wps;
        uint32_t* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
        unsigned long long* clk; CK(hipMalloc(&clk, 16)); unsigned long long hclk[2];
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = 400;
        if (pad) k<MODE, 1><<<blocks, 256>>>(d, 50, nullptr); else k<MODE, 0><<<blocks, 256>>>(d, 50, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (pad) k<MODE, 1><<<blocks, 256>>>(d, iters, clk); else k<MODE, 0><<<blocks, 256>>>(d, iters, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)hclk[0] / ((double)hclk[1] * 10.0);
        const double per_simd = (double)iters * 64 * 4 * wps;
        printf("%-44s start=%d mod 8 waves/SIMD=%d  %.3f ms  %.3f ns per instruction per SIMD  clock %.2f GHz -> %.2f clk\n", name, pad * 4, wps, ms, ms * 1e6 / per_simd, ghz, (double)hclk[0] / per_simd);
        CK(hipFree(d));
    }
    return 0;
}

int main()
{
    if (run<0>("v_and_b32 (VOP2)")) return 1;
    if (run<1>("v_bitop3_b32")) return 1;
    if (run<2>("v_bfe_u32 v, imm, imm")) return 1;
    if (run<3>("v_lshl_or_b32 v, imm, v")) return 1;
    if (run<4>("v_and_or_b32 v, s, v")) return 1;
    if (run<5>("v_and_or_b32 v, v, v")) return 1;
    if (run<6>("v_and_b32_sdwa BYTE_k -> BYTE_1 preserve")) return 1;
    if (run<7>("v_and_b32_sdwa BYTE_k -> DWORD pad")) return 1;
    if (run<8>("v_lshrrev_b32 (VOP2)")) return 1;
    if (run<9>("v_add_u32 (VOP2)")) return 1;
    if (run<10>("v_add_co_u32_e64")) return 1;
    if (run<11>("v_addc_co_u32_e64")) return 1;
    if (run<12>("v_cmp_gt_u32 vcc (VOPC)")) return 1;
    if (run<13>("v_cndmask_b32 vcc (VOP2)")) return 1;
    if (run<14>("v_pk_add_u16")) return 1;
    if (run<15>("v_perm_b32")) return 1;
    if (run<16>("v_alignbit_b32 v, v, imm")) return 1;
    if (run<17>("v_bcnt_u32_b32")) return 1;
    if (run<18>("v_and_b32 literal")) return 1;
    if (run<19>("v_add_u32_sdwa src1 BYTE_3")) return 1;
    if (run<20>("sub/xor/or/lshl VOP2 mix")) return 1;
    if (run<21>("and + add_co + bitop3 + sdwa mix")) return 1;
    if (run<22>("v_xad_u32")) return 1;
    if (run<23>("v_and_b32 s, v (VOP2)")) return 1;
    if (run<24>("v_mad_u32_u24")) return 1;
    if (run<25>("v_or3_b32")) return 1;
    if (run<26>("v_bitop3_b32 v, v, s")) return 1;
    if (run<27>("v_bitop3_b32 v, v, same v")) return 1;
    if (run<28>("v_xor_b32")) return 1;
    if (run<29>("v_or_b32")) return 1;
    if (run<30>("v_sub_u32")) return 1;
    if (run<31>("v_lshlrev_b32")) return 1;
    if (run<32>("and add and add")) return 1;
    if (run<33>("and and add add")) return 1;
    if (run<34>("and bitop3 and bitop3")) return 1;
    if (run<35>("and and bitop3 bitop3")) return 1;
    if (run<36>("and lshr add bitop3 (all fast alone)")) return 1;
    if (run<37>("column: 2 and, add_co, addc_co, 2 bitop3, sdwa (7 instructions: x 4/7)")) return 1;
    if (run<38>("column without carries or sdwa (7 instructions: x 4/7)")) return 1;
    return 0;
}
