// Microbenchmark: what does an addition with an SGPR-pair carry cost next to the other VALU instructions of the
// window kernel?  Independent instructions, explicit registers, 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template<int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters)
{
    uint32_t r = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            asm volatile(REP64("v_bitop3_b32 v4, v5, v6, v7 bitop3:0x96\n v_bitop3_b32 v8, v9, v10, v11 bitop3:0x96\n"
                               "v_bitop3_b32 v12, v13, v14, v15 bitop3:0x96\n v_bitop3_b32 v16, v17, v18, v19 bitop3:0x96\n")
                         ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
        } else if (MODE == 1) {   // carry in and out through four different SGPR pairs, no dependence between instructions
            asm volatile(REP64("v_addc_co_u32_e64 v4, s[40:41], v5, v6, s[48:49]\n v_addc_co_u32_e64 v8, s[42:43], v9, v10, s[50:51]\n"
                               "v_addc_co_u32_e64 v12, s[44:45], v13, v14, s[52:53]\n v_addc_co_u32_e64 v16, s[46:47], v17, v18, s[54:55]\n")
                         ::: "v4","v5","v6","v8","v9","v10","v12","v13","v14","v16","v17","v18","s40","s41","s42","s43","s44","s45","s46","s47");
        } else if (MODE == 2) {   // carry out only
            asm volatile(REP64("v_add_co_u32_e64 v4, s[40:41], v5, v6\n v_add_co_u32_e64 v8, s[42:43], v9, v10\n"
                               "v_add_co_u32_e64 v12, s[44:45], v13, v14\n v_add_co_u32_e64 v16, s[46:47], v17, v18\n")
                         ::: "v4","v5","v6","v8","v9","v10","v12","v13","v14","v16","v17","v18","s40","s41","s42","s43","s44","s45","s46","s47");
        } else if (MODE == 3) {   // three-operand add, no carry
            asm volatile(REP64("v_add3_u32 v4, v5, v6, v7\n v_add3_u32 v8, v9, v10, v11\n v_add3_u32 v12, v13, v14, v15\n v_add3_u32 v16, v17, v18, v19\n")
                         ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
        } else if (MODE == 4) {   // 64-bit add
            asm volatile(REP64("v_lshl_add_u64 v[4:5], v[6:7], 0, v[8:9]\n v_lshl_add_u64 v[10:11], v[12:13], 0, v[14:15]\n"
                               "v_lshl_add_u64 v[16:17], v[18:19], 0, v[20:21]\n v_lshl_add_u64 v[22:23], v[24:25], 0, v[26:27]\n")
                         ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27");
        } else if (MODE == 5) {   // plain VOP2 add
            asm volatile(REP64("v_add_u32 v4, v5, v6\n v_add_u32 v8, v9, v10\n v_add_u32 v12, v13, v14\n v_add_u32 v16, v17, v18\n")
                         ::: "v4","v5","v6","v8","v9","v10","v12","v13","v14","v16","v17","v18");
        } else if (MODE == 6) {   // VOP2 carry through vcc, independent sums (vcc is rewritten by every instruction)
            asm volatile(REP64("v_addc_co_u32 v4, vcc, v5, v6, vcc\n v_addc_co_u32 v8, vcc, v9, v10, vcc\n v_addc_co_u32 v12, vcc, v13, v14, vcc\n v_addc_co_u32 v16, vcc, v17, v18, vcc\n")
                         ::: "v4","v5","v6","v8","v9","v10","v12","v13","v14","v16","v17","v18","vcc");
        } else if (MODE == 7) {   // the window kernel's mix: 11 bitop3/bfe-like + 3 additions with SGPR carries (independent)
            asm volatile(REP64("v_bitop3_b32 v4, v5, v6, v7 bitop3:0x96\n v_bitop3_b32 v8, v9, v10, v11 bitop3:0x96\n v_bitop3_b32 v12, v13, v14, v15 bitop3:0x96\n"
                               "v_addc_co_u32_e64 v16, s[40:41], v17, v18, s[48:49]\n")
                         ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","s40","s41");
        } else if (MODE == 8) {   // the same with the carry replaced by add3 + shift
            asm volatile(REP64("v_bitop3_b32 v4, v5, v6, v7 bitop3:0x96\n v_bitop3_b32 v8, v9, v10, v11 bitop3:0x96\n v_bitop3_b32 v12, v13, v14, v15 bitop3:0x96\n"
                               "v_add3_u32 v16, v17, v18, v19\n v_lshrrev_b32 v20, 31, v21\n")
                         ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template<int MODE> int run(const char* name, int blocks, int per_rep)
{
    uint32_t* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 200;
    k<MODE><<<blocks, 256>>>(d, 10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_wave = (double)iters * 64 * per_rep;
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    // cycles per instruction per SIMD at 2.4 GHz nominal
    printf("%-52s waves/SIMD=%.0f  %.3f ms  %.2f clk per instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd));
    CK(hipFree(d));
    return 0;
}

int main()
{
    for (int wps : {1, 2, 4}) {
        int blocks = 256 * wps;
        if (run<0>("v_bitop3_b32", blocks, 4)) return 1;
        if (run<1>("v_addc_co_u32_e64 sgpr in/out", blocks, 4)) return 1;
        if (run<2>("v_add_co_u32_e64 sgpr out", blocks, 4)) return 1;
        if (run<3>("v_add3_u32", blocks, 4)) return 1;
        if (run<4>("v_lshl_add_u64", blocks, 4)) return 1;
        if (run<5>("v_add_u32 (VOP2)", blocks, 4)) return 1;
        if (run<6>("v_addc_co_u32 vcc (VOP2)", blocks, 4)) return 1;
        if (run<7>("3 bitop3 + addc sgpr", blocks, 4)) return 1;
        if (run<8>("3 bitop3 + add3 + lshr", blocks, 5)) return 1;
    }
    return 0;
}
