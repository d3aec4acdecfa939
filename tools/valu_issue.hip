// Microbenchmark: issue rate of the VALU instructions family 1 is made of, with explicit registers
// (bank layout matters: VGPR bank = register number mod 4).  Evidence for DESIGN.md.
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
        if (MODE == 0) {        // bitop3, 3 sources in 3 different banks, 4 independent chains
            asm volatile(REP64("v_bitop3_b32 v4, v5, v6, v7 bitop3:0x96\n v_bitop3_b32 v8, v9, v10, v11 bitop3:0x96\n"
                               "v_bitop3_b32 v12, v13, v14, v15 bitop3:0x96\n v_bitop3_b32 v16, v17, v18, v19 bitop3:0x96\n")
                         ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
        } else if (MODE == 1) { // bitop3, 3 sources in the same bank
            asm volatile(REP64("v_bitop3_b32 v4, v8, v12, v16 bitop3:0x96\n v_bitop3_b32 v5, v9, v13, v17 bitop3:0x96\n"
                               "v_bitop3_b32 v6, v10, v14, v18 bitop3:0x96\n v_bitop3_b32 v7, v11, v15, v19 bitop3:0x96\n")
                         ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
        } else if (MODE == 2) { // VOP2 xor, 2 sources
            asm volatile(REP64("v_xor_b32 v4, v5, v6\n v_xor_b32 v8, v9, v10\n v_xor_b32 v12, v13, v14\n v_xor_b32 v16, v17, v18\n")
                         ::: "v4","v5","v6","v8","v9","v10","v12","v13","v14","v16","v17","v18");
        } else if (MODE == 3) { // dependent carry chain only
            asm volatile(REP64("v_add_co_u32 v4, vcc, v5, v6\n v_addc_co_u32 v8, vcc, v9, v10, vcc\n v_addc_co_u32 v12, vcc, v13, v14, vcc\n v_addc_co_u32 v16, vcc, v17, v18, vcc\n")
                         ::: "v4","v5","v6","v8","v9","v10","v12","v13","v14","v16","v17","v18","vcc");
        } else if (MODE == 4) { // the family-1 limb pattern: 2 bitop3, addc, bitop3 (addc chain dependent)
            asm volatile(REP64("v_bitop3_b32 v4, v5, v6, v7 bitop3:0x90\n v_bitop3_b32 v4, v4, v9, v10 bitop3:0x90\n v_addc_co_u32 v12, vcc, v5, v4, vcc\n v_bitop3_b32 v16, v12, v5, v4 bitop3:0xf6\n")
                         ::: "v4","v5","v6","v7","v9","v10","v12","v16","vcc");
        } else if (MODE == 5) { // bitop3 with one SGPR-like constant source (2 VGPR sources)
            asm volatile(REP64("v_bitop3_b32 v4, v5, v6, 0 bitop3:0x96\n v_bitop3_b32 v8, v9, v10, 0 bitop3:0x96\n"
                               "v_bitop3_b32 v12, v13, v14, 0 bitop3:0x96\n v_bitop3_b32 v16, v17, v18, 0 bitop3:0x96\n")
                         ::: "v4","v5","v6","v8","v9","v10","v12","v13","v14","v16","v17","v18");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template<int MODE> int run(const char* name, int blocks)
{
    uint32_t* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 200;
    k<MODE><<<blocks, 256>>>(d, 10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 256 * iters * 256.0;
    printf("%-44s blocks=%5d  %.3f ms  %.2f T lane-ops/s\n", name, blocks, ms, ops / (ms * 1e-3) / 1e12);
    CK(hipFree(d));
    return 0;
}

int main()
{
    for (int wpc : {4, 8, 16, 32}) {
        int blocks = 256 * wpc / 4;
        printf("-- %d waves/CU\n", wpc);
        if (run<0>("bitop3 3 VGPR src, distinct banks", blocks)) return 1;
        if (run<1>("bitop3 3 VGPR src, same bank", blocks)) return 1;
        if (run<5>("bitop3 2 VGPR src + const", blocks)) return 1;
        if (run<2>("v_xor (VOP2)", blocks)) return 1;
        if (run<3>("v_addc chain (dependent)", blocks)) return 1;
        if (run<4>("limb pattern 3 bitop3 + addc", blocks)) return 1;
    }
    return 0;
}
