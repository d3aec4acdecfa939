// Issue rate of the window kernel's instruction stream with two limbs per lane (10 instructions per column, two
// dependency chains) against three limbs (14 instructions, three chains), at 1 / 2 / 3 waves per SIMD.  Evidence for
// the "two limbs, half-limb steps" item under DESIGN.md "next".
//   hipcc --offload-arch=gfx950 -O2 tools/two_limb_rate.hip -o tools/two_limb_rate && tools/two_limb_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define BFE(s, A0, A1) "v_bfe_i32 " A0 ", v32, " #s ", 1\n\tv_bfe_i32 " A1 ", v36, " #s ", 1\n\t"
#define UA(U, V, P0, X0) "v_bitop3_b32 " U ", " V ", " P0 ", " X0 " bitop3:0x90\n\t"
#define UB(U, P1, X1) "v_bitop3_b32 " U ", " U ", " P1 ", " X1 " bitop3:0x90\n\t"
#define AD0(S, O, V, U) "v_add_co_u32_e64 " S ", " O ", " V ", " U "\n\t"
#define ADC(S, O, V, U, I) "v_addc_co_u32_e64 " S ", " O ", " V ", " U ", " I "\n\t"
#define NX(V, U, S) "v_bitop3_b32 " V ", " U ", " S ", " V " bitop3:0xde\n\t"
// step s: limb 0 (v20, planes v21 v33) on column s with text bits A, limb 1 (v24, v25 v37) on column s - 1 with B
#define STEP2(s, A0, A1, B0, B1, I1, O1) BFE(s, A0, A1) UA("v26", "v24", "v25", B0) UA("v22", "v20", "v21", A0) \
    UB("v26", "v37", B1) UB("v22", "v33", A1) ADC("v27", "s[48:49]", "v24", "v26", I1) AD0("v23", O1, "v20", "v22") \
    NX("v24", "v26", "v27") NX("v20", "v22", "v23")
#define STEP3(s, A0, A1, B0, B1, C0, C1, I1, I2, O1, O2) BFE(s, A0, A1) UA("v26", "v24", "v25", B0) UA("v30", "v28", "v29", C0) \
    UA("v22", "v20", "v21", A0) UB("v26", "v37", B1) UB("v30", "v41", C1) UB("v22", "v33", A1) \
    ADC("v31", "s[50:51]", "v28", "v30", I2) ADC("v27", O2, "v24", "v26", I1) AD0("v23", O1, "v20", "v22") \
    NX("v28", "v30", "v31") NX("v24", "v26", "v27") NX("v20", "v22", "v23")
#define TWO2(s0, s1) STEP2(s0, "v35", "v47", "v39", "v51", "s[44:45]", "s[40:41]") STEP2(s1, "v39", "v51", "v35", "v47", "s[40:41]", "s[44:45]")
#define TWO3(s0, s1) STEP3(s0, "v35", "v47", "v39", "v51", "v43", "v55", "s[44:45]", "s[46:47]", "s[40:41]", "s[42:43]") \
    STEP3(s1, "v39", "v51", "v43", "v55", "v35", "v47", "s[40:41]", "s[42:43]", "s[44:45]", "s[46:47]")
#define X16(M) M(0, 1) M(2, 3) M(4, 5) M(6, 7) M(8, 9) M(10, 11) M(12, 13) M(14, 15) M(16, 17) M(18, 19) M(20, 21) M(22, 23) M(24, 25) M(26, 27) M(28, 29) M(30, 31)
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v35","v36","v37","v39","v41","v43","v47","v51","v55", \
    "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51"

template <int LIMBS>
__global__ void __launch_bounds__(64) k(uint32_t *out, int blocks)
{
    asm volatile("v_mov_b32 v20, -1\n\tv_mov_b32 v24, -1\n\tv_mov_b32 v28, -1\n\tv_mov_b32 v21, 0x12345678\n\tv_mov_b32 v25, 0x9abcdef0\n\t"
                 "v_mov_b32 v29, 0x0f1e2d3c\n\tv_mov_b32 v33, 0x55aa55aa\n\tv_mov_b32 v37, 0x33cc33cc\n\tv_mov_b32 v41, 0x0ff00ff0\n\t"
                 "v_mov_b32 v32, 0x6b8b4567\n\tv_mov_b32 v36, 0x327b23c6\n\ts_mov_b64 s[40:41], 0\n\ts_mov_b64 s[42:43], 0\n\t"
                 "s_mov_b64 s[44:45], 0\n\ts_mov_b64 s[46:47], 0" ::: CLOB);
    for (int b = 0; b < blocks; b++) {
        if (LIMBS == 2) asm volatile(".p2align 3\n\ts_nop 0\n\t" X16(TWO2) ::: CLOB);
        else            asm volatile(".p2align 3\n\ts_nop 0\n\t" X16(TWO3) ::: CLOB);
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v20, v24" : "=v"(r) :: CLOB);
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

int main()
{
    uint32_t *d;
    CK(hipMalloc(&d, 4096 * 64 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 64;
    for (int limbs = 2; limbs <= 3; limbs++)
        for (int waves : {1024, 2048, 3072, 4096}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                if (limbs == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, d, blocks);
                else            hipLaunchKernelGGL(k<3>, dim3(waves), dim3(64), 0, 0, d, blocks);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double instr = (double)blocks * 32 * (limbs == 2 ? 10 : 14);
            const double per_simd = waves / 1024.0;
            printf("limbs %d, %d waves/SIMD: %.1f us, %.2f clk (2.4 GHz) per instruction and SIMD, %.2f ns per column and wave\n", limbs, waves / 1024,
                   best * 1e3, best * 1e-3 * 2.4e9 / (instr * per_simd), best * 1e6 / (blocks * 32.0 * per_simd));
        }
    return 0;
}
