// Would match masks looked up in LDS beat match masks folded from the bit planes?  The window kernel's two-limb step
// spends 4 of its 10 instructions on U = V & ~(P0 ^ x0) & ~(P1 ^ x1) (two v_bitop3 per limb) and 2 on the text bits.
// With the four match masks of the window (A, C, G, T; 64 rows = one ds_read_b64) in LDS, a step is: the column's
// symbol -> LDS address (v_bfe_u32 + v_lshl_or_b32), ds_read_b64 (LDS pipe, issued 8 columns ahead), 2 v_and_b32,
// the two additions, 2 v_bitop3 (V') = 8 VALU instructions, two of them VOP2.  This measures that stream against
// the current one at 1 ... 4 waves per SIMD.  (The masks would have to be rebuilt and written to LDS every 16
// columns: about 12 more instructions per 16 columns, not in this stream.)
//   hipcc --offload-arch=gfx950 -O2 tools/lds_match_rate.hip -o tools/lds_match_rate && tools/lds_match_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define UA(U, V, P0, X0) "v_bitop3_b32 " U ", " V ", " P0 ", " X0 " bitop3:0x90\n\t"
#define UB(U, P1, X1) "v_bitop3_b32 " U ", " U ", " P1 ", " X1 " bitop3:0x90\n\t"
#define NX_(V, U, S, W) "v_bitop3_b32 " V ", " U ", " S ", " W " bitop3:0xde\n\t"
#define NX3(V, U, S) NX_(V, U, S, V)
#define GET_NX(_1, _2, _3, _4, NAME, ...) NAME
#define NX(...) GET_NX(__VA_ARGS__, NX_, NX3)(__VA_ARGS__)
#define AD0(S, O, V, U) "v_add_co_u32_e64 " S ", " O ", " V ", " U "\n\t"
#define ADC(S, O, V, U, I) "v_addc_co_u32_e64 " S ", " O ", " V ", " U ", " I "\n\t"
#define BFE2(s, A0, A1) "v_bfe_i32 " A0 ", v32, " #s ", 1\n\tv_bfe_i32 " A1 ", v36, " #s ", 1\n\t"
#define STEP2(s, A0, A1, B0, B1, I1, O1) BFE2(s, A0, A1) UA("v26", "v24", "v25", B0) UA("v22", "v20", "v21", A0) \
    UB("v26", "v37", B1) UB("v22", "v33", A1) ADC("v27", "s[48:49]", "v24", "v26", I1) AD0("v23", O1, "v20", "v22") \
    NX("v24", "v26", "v27") NX("v20", "v22", "v23")
#define TWO_R(s0, s1) STEP2(s0, "v35", "v47", "v39", "v51", "s[44:45]", "s[40:41]") STEP2(s1, "v39", "v51", "v35", "v47", "s[40:41]", "s[44:45]")
// LDS variant: v60 = lane's table base (lane * 8), text v32 holds 2-bit symbols (16 per dword); masks of column s in M
// step: address of column s + 8 and its read; wait until column s has arrived (8 reads in flight); the recurrence
#define LADDR(s2) "v_bfe_u32 v61, v32, " #s2 ", 2\n\tv_lshl_or_b32 v61, v61, 9, v60\n\t"
#define LREAD(M) "ds_read_b64 " M ", v61\n\t"
#define LSTEP(s2, MN, M0, M1, I1, O1) LADDR(s2) LREAD(MN) "s_waitcnt lgkmcnt(7)\n\t" \
    "v_and_b32 v26, v24, " M1 "\n\tv_and_b32 v22, v20, " M0 "\n\t" \
    ADC("v27", "s[48:49]", "v24", "v26", I1) AD0("v23", O1, "v20", "v22") NX("v24", "v26", "v27") NX("v20", "v22", "v23")
// variants of the LDS step, to price its parts in context: no LDS read; plain VOP2 additions; no address arithmetic
#define LSTEP_NOREAD(s2, MN, M0, M1, I1, O1) LADDR(s2) \
    "v_and_b32 v26, v24, " M1 "\n\tv_and_b32 v22, v20, " M0 "\n\t" \
    ADC("v27", "s[48:49]", "v24", "v26", I1) AD0("v23", O1, "v20", "v22") NX("v24", "v26", "v27") NX("v20", "v22", "v23")
#define LSTEP_ADD2(s2, MN, M0, M1, I1, O1) LADDR(s2) LREAD(MN) "s_waitcnt lgkmcnt(7)\n\t" \
    "v_and_b32 v26, v24, " M1 "\n\tv_and_b32 v22, v20, " M0 "\n\t" \
    "v_add_u32 v27, v24, v26\n\tv_add_u32 v23, v20, v22\n\t" NX("v24", "v26", "v27") NX("v20", "v22", "v23")
#define LSTEP_NOADDR(s2, MN, M0, M1, I1, O1) LREAD(MN) "s_waitcnt lgkmcnt(7)\n\t" \
    "v_and_b32 v26, v24, " M1 "\n\tv_and_b32 v22, v20, " M0 "\n\t" \
    ADC("v27", "s[48:49]", "v24", "v26", I1) AD0("v23", O1, "v20", "v22") NX("v24", "v26", "v27") NX("v20", "v22", "v23")
// the same eight VALU instructions and the read with no dependence between neighbours: results go to registers nothing reads
#define LSTEP_INDEP(s2, MN, M0, M1, I1, O1) "v_bfe_u32 v62, v32, " #s2 ", 2\n\tv_lshl_or_b32 v62, v63, 9, v60\n\t" LREAD(MN) "s_waitcnt lgkmcnt(7)\n\t" \
    "v_and_b32 v26, v24, " M1 "\n\tv_and_b32 v22, v20, " M0 "\n\t" \
    ADC("v27", "s[48:49]", "v25", "v28", I1) AD0("v23", O1, "v21", "v29") NX("v34", "v38", "v39", "v33") NX("v40", "v41", "v42", "v43")
#define L8X(S, a) \
    S(a, "v[64:65]", "v64", "v65", "s[44:45]", "s[40:41]") S(a, "v[66:67]", "v66", "v67", "s[40:41]", "s[44:45]") \
    S(a, "v[68:69]", "v68", "v69", "s[44:45]", "s[40:41]") S(a, "v[70:71]", "v70", "v71", "s[40:41]", "s[44:45]") \
    S(a, "v[72:73]", "v72", "v73", "s[44:45]", "s[40:41]") S(a, "v[74:75]", "v74", "v75", "s[40:41]", "s[44:45]") \
    S(a, "v[76:77]", "v76", "v77", "s[44:45]", "s[40:41]") S(a, "v[78:79]", "v78", "v79", "s[40:41]", "s[44:45]")
#define L8(a) \
    LSTEP(a, "v[64:65]", "v64", "v65", "s[44:45]", "s[40:41]") LSTEP(a, "v[66:67]", "v66", "v67", "s[40:41]", "s[44:45]") \
    LSTEP(a, "v[68:69]", "v68", "v69", "s[44:45]", "s[40:41]") LSTEP(a, "v[70:71]", "v70", "v71", "s[40:41]", "s[44:45]") \
    LSTEP(a, "v[72:73]", "v72", "v73", "s[44:45]", "s[40:41]") LSTEP(a, "v[74:75]", "v74", "v75", "s[40:41]", "s[44:45]") \
    LSTEP(a, "v[76:77]", "v76", "v77", "s[44:45]", "s[40:41]") LSTEP(a, "v[78:79]", "v78", "v79", "s[40:41]", "s[44:45]")
#define X16(M) M(0, 1) M(2, 3) M(4, 5) M(6, 7) M(8, 9) M(10, 11) M(12, 13) M(14, 15) M(16, 17) M(18, 19) M(20, 21) M(22, 23) M(24, 25) M(26, 27) M(28, 29) M(30, 31)
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v32","v33","v35","v36","v37","v39","v47","v51","v60","v61", \
    "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
    "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49"

template <int MODE>
__global__ void __launch_bounds__(64) k(uint32_t *out, int blocks)
{
    __shared__ uint64_t table[4 * 64];
    for (int i = threadIdx.x; i < 256; i += 64) table[i] = 0x0123456789abcdefull * (i + 1);
    __syncthreads();
    asm volatile("v_mov_b32 v20, -1\n\tv_mov_b32 v24, -1\n\tv_mov_b32 v21, 0x12345678\n\tv_mov_b32 v25, 0x9abcdef0\n\t"
                 "v_mov_b32 v33, 0x55aa55aa\n\tv_mov_b32 v37, 0x33cc33cc\n\tv_mov_b32 v32, 0x6b8b4567\n\tv_mov_b32 v36, 0x327b23c6\n\t"
                 "v_lshlrev_b32 v60, 3, %0\n\ts_mov_b64 s[40:41], 0\n\ts_mov_b64 s[44:45], 0\n\t"
                 "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\tv_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\t"
                 "v_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\tv_mov_b32 v72, 0\n\tv_mov_b32 v73, 0\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\t"
                 "v_mov_b32 v76, 0\n\tv_mov_b32 v77, 0\n\tv_mov_b32 v78, 0\n\tv_mov_b32 v79, 0" :: "v"(threadIdx.x) : CLOB);
    for (int b = 0; b < blocks; b++) {
        if (MODE == 0) asm volatile(".p2align 3\n\ts_nop 0\n\t" X16(TWO_R) ::: CLOB);
        if (MODE == 1) asm volatile(".p2align 3\n\ts_nop 0\n\t" L8(0) L8(16) L8(4) L8(20) "s_waitcnt lgkmcnt(0)" ::: CLOB, "memory");
        if (MODE == 2) asm volatile(".p2align 3\n\ts_nop 0\n\t" L8X(LSTEP_NOREAD, 0) L8X(LSTEP_NOREAD, 16) L8X(LSTEP_NOREAD, 4) L8X(LSTEP_NOREAD, 20) ::: CLOB, "memory");
        if (MODE == 3) asm volatile(".p2align 3\n\ts_nop 0\n\t" L8X(LSTEP_ADD2, 0) L8X(LSTEP_ADD2, 16) L8X(LSTEP_ADD2, 4) L8X(LSTEP_ADD2, 20) "s_waitcnt lgkmcnt(0)" ::: CLOB, "memory");
        if (MODE == 5) asm volatile(".p2align 3\n\ts_nop 0\n\t" "v_mov_b32 v61, v60\n\tv_mov_b32 v63, 1\n\t" L8X(LSTEP_INDEP, 0) L8X(LSTEP_INDEP, 16) L8X(LSTEP_INDEP, 4) L8X(LSTEP_INDEP, 20) "s_waitcnt lgkmcnt(0)" ::: CLOB, "v28", "v29", "v34", "v38", "v40", "v41", "v42", "v43", "v62", "v63", "memory");
        if (MODE == 4) asm volatile(".p2align 3\n\ts_nop 0\n\t" "v_mov_b32 v61, v60\n\t" L8X(LSTEP_NOADDR, 0) L8X(LSTEP_NOADDR, 16) L8X(LSTEP_NOADDR, 4) L8X(LSTEP_NOADDR, 20) "s_waitcnt lgkmcnt(0)" ::: CLOB, "memory");
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v20, v24" : "=v"(r) :: CLOB);
    out[blockIdx.x * 64 + threadIdx.x] = r + (uint32_t)table[threadIdx.x];
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
        const double per_simd = waves / 1024.0;
        printf("%-36s %d waves/SIMD: %6.1f us, %.2f clk (2.4 GHz) per VALU instruction and SIMD, %.2f ns per column and wave\n", name, waves / 1024,
               best * 1e3, best * 1e-3 * 2.4e9 / ((double)blocks * 32 * per_col * per_simd), best * 1e6 / (blocks * 32.0 * per_simd));
    }
    return 0;
}

int main()
{
    uint32_t *d;
    CK(hipMalloc(&d, 4096 * 64 * 4));
    if (run<0>("planes folded (10 VALU per column)", 10, d)) return 1;
    if (run<1>("masks from LDS (8 VALU + ds_read_b64)", 8, d)) return 1;
    if (run<2>("  the same without the LDS read", 8, d)) return 1;
    if (run<3>("  the same with plain VOP2 additions", 8, d)) return 1;
    if (run<4>("  the same without address arithmetic", 6, d)) return 1;
    if (run<5>("  the same, no dependent neighbours", 8, d)) return 1;
    return 0;
}
