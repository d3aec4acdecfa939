// Issue rate of a two-limb window step built on one 64-bit addition (v_lshl_add_u64: the window of 64 rows needs no
// carry in or out) against the two-limb step with two carry-chained 32-bit additions (tools/two_limb_rate.hip), at
// 1 ... 4 waves per SIMD.  Variants of the 9-instruction step differ in order (plain / text bits of the next column
// extracted early) only; register banks are chosen so that no instruction but the last two reads two sources from one
// bank (V, U and the sum are even-aligned pairs, so V' = f(U, S, V) always has two sources in one of banks 0 / 2).
//   hipcc --offload-arch=gfx950 -O2 tools/add64_rate.hip -o tools/add64_rate && tools/add64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define BFE(s, A0, A1) "v_bfe_i32 " A0 ", v32, " #s ", 1\n\tv_bfe_i32 " A1 ", v37, " #s ", 1\n\t"
#define UA(U, V, P0, X0) "v_bitop3_b32 " U ", " V ", " P0 ", " X0 " bitop3:0x90\n\t"
#define UB(U, P1, X1) "v_bitop3_b32 " U ", " U ", " P1 ", " X1 " bitop3:0x90\n\t"
#define NX(V, U, S) "v_bitop3_b32 " V ", " U ", " S ", " V " bitop3:0xde\n\t"
#define ADD64 "v_lshl_add_u64 v[24:25], v[20:21], 0, v[22:23]\n\t"
// V v[20:21], U v[22:23], S v[24:25]; planes P00 v29 P10 v30 P01 v33 P11 v34; text words v32 v37
#define CORE(X0, X1) UA("v22", "v20", "v29", X0) UA("v23", "v21", "v30", X0) UB("v22", "v33", X1) UB("v23", "v34", X1) ADD64
#define STEP_A(s, X0, X1) BFE(s, X0, X1) CORE(X0, X1) NX("v20", "v22", "v24") NX("v21", "v23", "v25")
// B: the text bits of column s + 1 are extracted between the addition and the last two instructions of column s
#define STEP_B(s1, X0, X1, Y0, Y1) CORE(X0, X1) BFE(s1, Y0, Y1) NX("v20", "v22", "v24") NX("v21", "v23", "v25")
#define TWO_A(s0, s1) STEP_A(s0, "v35", "v36") STEP_A(s1, "v35", "v36")
#define TWO_B(s0, s1) STEP_B(s1, "v35", "v36", "v39", "v40") STEP_B(s0, "v39", "v40", "v35", "v36")
// C: as A but V' written with conflicting banks avoided by keeping the sum in v[26:27] (bank 2/3): same pigeonhole, other bank
#define ADD64C "v_lshl_add_u64 v[26:27], v[20:21], 0, v[22:23]\n\t"
#define STEP_C(s, X0, X1) BFE(s, X0, X1) UA("v22", "v20", "v29", X0) UA("v23", "v21", "v30", X0) UB("v22", "v33", X1) UB("v23", "v34", X1) ADD64C \
    NX("v20", "v22", "v26") NX("v21", "v23", "v27")
#define TWO_C(s0, s1) STEP_C(s0, "v35", "v36") STEP_C(s1, "v35", "v36")
// reference: the stream of two_limb_rate.hip (10 instructions per column)
#define AD0(S, O, V, U) "v_add_co_u32_e64 " S ", " O ", " V ", " U "\n\t"
#define ADC(S, O, V, U, I) "v_addc_co_u32_e64 " S ", " O ", " V ", " U ", " I "\n\t"
#define BFE2(s, A0, A1) "v_bfe_i32 " A0 ", v32, " #s ", 1\n\tv_bfe_i32 " A1 ", v36, " #s ", 1\n\t"
#define STEP2(s, A0, A1, B0, B1, I1, O1) BFE2(s, A0, A1) UA("v26", "v24", "v25", B0) UA("v22", "v20", "v21", A0) \
    UB("v26", "v37", B1) UB("v22", "v33", A1) ADC("v27", "s[48:49]", "v24", "v26", I1) AD0("v23", O1, "v20", "v22") \
    NX("v24", "v26", "v27") NX("v20", "v22", "v23")
#define TWO_R(s0, s1) STEP2(s0, "v35", "v47", "v39", "v51", "s[44:45]", "s[40:41]") STEP2(s1, "v39", "v51", "v35", "v47", "s[40:41]", "s[44:45]")
#define X16(M) M(0, 1) M(2, 3) M(4, 5) M(6, 7) M(8, 9) M(10, 11) M(12, 13) M(14, 15) M(16, 17) M(18, 19) M(20, 21) M(22, 23) M(24, 25) M(26, 27) M(28, 29) M(30, 31)
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v39","v40","v41","v43","v47","v51","v55", \
    "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51"

template <int MODE>
__global__ void __launch_bounds__(64) k(uint32_t *out, int blocks)
{
    asm volatile("v_mov_b32 v20, -1\n\tv_mov_b32 v21, -1\n\tv_mov_b32 v24, -1\n\tv_mov_b32 v29, 0x12345678\n\tv_mov_b32 v30, 0x9abcdef0\n\t"
                 "v_mov_b32 v25, 0x0f1e2d3c\n\tv_mov_b32 v33, 0x55aa55aa\n\tv_mov_b32 v34, 0x33cc33cc\n\tv_mov_b32 v37, 0x0ff00ff0\n\t"
                 "v_mov_b32 v32, 0x6b8b4567\n\tv_mov_b32 v36, 0x327b23c6\n\ts_mov_b64 s[40:41], 0\n\ts_mov_b64 s[42:43], 0\n\t"
                 "v_mov_b32 v35, 0\n\tv_mov_b32 v39, 0\n\tv_mov_b32 v40, 0\n\ts_mov_b64 s[44:45], 0\n\ts_mov_b64 s[46:47], 0" ::: CLOB);
    for (int b = 0; b < blocks; b++) {
        if (MODE == 0) asm volatile(".p2align 3\n\ts_nop 0\n\t" X16(TWO_R) ::: CLOB);
        if (MODE == 1) asm volatile(".p2align 3\n\ts_nop 0\n\t" X16(TWO_A) ::: CLOB);
        if (MODE == 2) asm volatile(".p2align 3\n\ts_nop 0\n\t" X16(TWO_B) ::: CLOB);
        if (MODE == 3) asm volatile(".p2align 3\n\ts_nop 0\n\t" X16(TWO_C) ::: CLOB);
        if (MODE == 4) asm volatile(".p2align 3\n\t" X16(TWO_A) ::: CLOB);
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v20, v21" : "=v"(r) :: CLOB);
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int MODE> int run(const char *name, int per_col, uint32_t *d)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 64;
    for (int waves : {1024, 2048, 3072, 4096}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, d, blocks);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double instr = (double)blocks * 32 * per_col;
        const double per_simd = waves / 1024.0;
        printf("%-34s %d waves/SIMD: %6.1f us, %.2f clk (2.4 GHz) per instruction and SIMD, %.2f ns per column and wave\n", name, waves / 1024,
               best * 1e3, best * 1e-3 * 2.4e9 / (instr * per_simd), best * 1e6 / (blocks * 32.0 * per_simd));
    }
    return 0;
}

int main()
{
    uint32_t *d;
    CK(hipMalloc(&d, 4096 * 64 * 4));
    if (run<0>("two addc (10 instr)", 10, d)) return 1;
    if (run<1>("add64 plain (9 instr)", 9, d)) return 1;
    if (run<2>("add64, next bfe early", 9, d)) return 1;
    if (run<3>("add64, sum in v[26:27]", 9, d)) return 1;
    if (run<4>("add64 plain, 8-byte aligned", 9, d)) return 1;
    return 0;
}
