// Microbenchmark for the window kernel's step (three skewed limbs, DNA): what limits its issue rate?
// Modes differ only in register banks (VGPR bank = register number mod 4) and padding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// one step; register names as macro arguments
#define STEP(V0,V1,V2,P00,P01,P10,P11,P20,P21,A0,A1,B0,B1,C0,C1,U0,U1,U2,S0,S1,S2,W0,W1,NOP) \
 "v_bfe_i32 " A0 ", " W0 ", 5, 1\n v_bfe_i32 " A1 ", " W1 ", 5, 1\n" \
 "v_bitop3_b32 " U1 ", " V1 ", " P10 ", " B0 " bitop3:0x90\n" \
 "v_bitop3_b32 " U2 ", " V2 ", " P20 ", " C0 " bitop3:0x90\n" \
 "v_bitop3_b32 " U0 ", " V0 ", " P00 ", " A0 " bitop3:0x90\n" NOP \
 "v_bitop3_b32 " U1 ", " U1 ", " P11 ", " B1 " bitop3:0x90\n" \
 "v_bitop3_b32 " U2 ", " U2 ", " P21 ", " C1 " bitop3:0x90\n" \
 "v_bitop3_b32 " U0 ", " U0 ", " P01 ", " A1 " bitop3:0x90\n" NOP \
 "v_addc_co_u32_e64 " S2 ", s[48:49], " V2 ", " U2 ", s[42:43]\n" \
 "v_addc_co_u32_e64 " S1 ", s[42:43], " V1 ", " U1 ", s[40:41]\n" \
 "v_add_co_u32_e64 " S0 ", s[40:41], " V0 ", " U0 "\n" NOP \
 "v_bitop3_b32 " V2 ", " U2 ", " S2 ", " V2 " bitop3:0xde\n" \
 "v_bitop3_b32 " V1 ", " U1 ", " S1 ", " V1 " bitop3:0xde\n" \
 "v_bitop3_b32 " V0 ", " U0 ", " S0 ", " V0 " bitop3:0xde\n" NOP

// two pairs per lane: six independent chains per step, bank-aware (second pair: registers + 44, carries s[50:55])
#define STEP2(NOP) \
 "v_bfe_i32 v35, v32, 5, 1\n v_bfe_i32 v47, v36, 5, 1\n v_bfe_i32 v79, v76, 5, 1\n v_bfe_i32 v91, v80, 5, 1\n" \
 "v_bitop3_b32 v26, v24, v25, v39 bitop3:0x90\n v_bitop3_b32 v70, v68, v69, v83 bitop3:0x90\n" \
 "v_bitop3_b32 v30, v28, v29, v43 bitop3:0x90\n v_bitop3_b32 v74, v72, v73, v87 bitop3:0x90\n" \
 "v_bitop3_b32 v22, v20, v21, v35 bitop3:0x90\n v_bitop3_b32 v66, v64, v65, v79 bitop3:0x90\n" NOP \
 "v_bitop3_b32 v26, v26, v37, v51 bitop3:0x90\n v_bitop3_b32 v70, v70, v81, v95 bitop3:0x90\n" \
 "v_bitop3_b32 v30, v30, v41, v55 bitop3:0x90\n v_bitop3_b32 v74, v74, v85, v99 bitop3:0x90\n" \
 "v_bitop3_b32 v22, v22, v33, v47 bitop3:0x90\n v_bitop3_b32 v66, v66, v77, v91 bitop3:0x90\n" NOP \
 "v_addc_co_u32_e64 v31, s[48:49], v28, v30, s[42:43]\n v_addc_co_u32_e64 v75, s[54:55], v72, v74, s[52:53]\n" \
 "v_addc_co_u32_e64 v27, s[42:43], v24, v26, s[40:41]\n v_addc_co_u32_e64 v71, s[52:53], v68, v70, s[50:51]\n" \
 "v_add_co_u32_e64 v23, s[40:41], v20, v22\n v_add_co_u32_e64 v67, s[50:51], v64, v66\n" NOP \
 "v_bitop3_b32 v28, v30, v31, v28 bitop3:0xde\n v_bitop3_b32 v72, v74, v75, v72 bitop3:0xde\n" \
 "v_bitop3_b32 v24, v26, v27, v24 bitop3:0xde\n v_bitop3_b32 v68, v70, v71, v68 bitop3:0xde\n" \
 "v_bitop3_b32 v20, v22, v23, v20 bitop3:0xde\n v_bitop3_b32 v64, v66, v67, v64 bitop3:0xde\n" NOP
#define CLOB2 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","s50","s51","s52","s53","s54","s55"

#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","s40","s41","s42","s43","s48","s49"

#define BA(NOP) STEP("v20","v24","v28","v21","v33","v25","v37","v29","v41","v35","v47","v39","v51","v43","v55","v22","v26","v30","v23","v27","v31","v32","v36",NOP)
template<int MODE>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, const uint32_t* rows = nullptr)
{
    uint32_t r = threadIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();
    const uint32_t* p = rows + ((size_t)blockIdx.x * 36 * 4 * 64 + threadIdx.x);
    for (int it = 0; it < iters; it++) {
        if (MODE >= 100 && MODE < 132) {    // alignment sweep: the stream starts (MODE - 100) * 4 bytes after a 128-byte boundary
            asm volatile(".p2align 7\n .rept %c0\n s_nop 0\n .endr\n" REP32(BA("")) :: "n"(MODE - 100) : CLOB);
        }
        if (MODE == 7) {      // loads, no wait (the data is never used)
            uint32_t n0, n1, n2, n3;
            asm volatile("global_load_dword %0, %4, off\n global_load_dword %1, %4, off offset:256\n"
                         "global_load_dword %2, %4, off offset:512\n global_load_dword %3, %4, off offset:768\n"
                         REP32(BA(""))
                         : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3) : "v"(p + (size_t)(it % 34) * 256) : CLOB);
        }
        if (MODE == 8) {      // wait only
            asm volatile(REP32(BA("")) "s_waitcnt vmcnt(0)" ::: CLOB);
        }
        if (MODE == 9) {      // loads of one shared row (cache resident) + wait
            uint32_t n0, n1, n2, n3;
            asm volatile("global_load_dword %0, %4, off\n global_load_dword %1, %4, off offset:256\n"
                         "global_load_dword %2, %4, off offset:512\n global_load_dword %3, %4, off offset:768\n"
                         REP32(BA("")) "s_waitcnt vmcnt(0)"
                         : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3) : "v"(rows + threadIdx.x) : CLOB);
            r += n0 + n1 + n2 + n3;
        }
        if (MODE == 10) {     // wait in the middle of the next block instead of the end of this one
            uint32_t n0, n1, n2, n3;
            asm volatile("global_load_dword %0, %4, off\n global_load_dword %1, %4, off offset:256\n"
                         "global_load_dword %2, %4, off offset:512\n global_load_dword %3, %4, off offset:768\n"
                         REP8(BA("")) REP8(BA("")) REP8(BA("")) "s_waitcnt vmcnt(0)\n" REP8(BA(""))
                         : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3) : "v"(p + (size_t)(it % 34) * 256) : CLOB);
            r += n0 + n1 + n2 + n3;
        }
        if (MODE == 6) {      // bank-aware + the kernel's memory traffic: 4 coalesced rows per 32 steps, waited for at the end
            uint32_t n0, n1, n2, n3;
            asm volatile("global_load_dword %0, %4, off\n global_load_dword %1, %4, off offset:256\n"
                         "global_load_dword %2, %4, off offset:512\n global_load_dword %3, %4, off offset:768\n"
                         REP32(BA("")) "s_waitcnt vmcnt(0)"
                         : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3) : "v"(p + (size_t)(it % 34) * 256) : CLOB);
            r += n0 + n1 + n2 + n3;
        }
        if (MODE == 0)        // naive names as in the first fused kernel: V v20-22, P v24-29, X v52-57, U v58-60, S v61-63
            asm volatile(REP32(STEP("v20","v21","v22","v24","v25","v26","v27","v28","v29","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v30","v31","")) ::: CLOB);
        else if (MODE == 1)   // same with s_nop 0 between the groups
            asm volatile(REP32(STEP("v20","v21","v22","v24","v25","v26","v27","v28","v29","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v30","v31","s_nop 0\n")) ::: CLOB);
        else if (MODE == 2)   // bank-aware: V bank 0, P bank 1, U bank 2, S and X bank 3
            asm volatile(REP32(STEP("v20","v24","v28","v21","v33","v25","v37","v29","v41","v35","v47","v39","v51","v43","v55","v22","v26","v30","v23","v27","v31","v32","v36","")) ::: CLOB);
        else if (MODE == 3)   // all sources of every bitop in the same bank (worst case): everything = 0 mod 4
            asm volatile(REP32(STEP("v20","v24","v28","v32","v36","v40","v44","v48","v52","v56","v60","v56","v60","v56","v60","v20","v24","v28","v32","v36","v40","v44","v48","")) ::: CLOB);
        else if (MODE == 4)   // two pairs per lane, bank-aware: 28 instructions per step, dependent ones 6 apart
            asm volatile(REP8(STEP2("")) REP8(STEP2("")) ::: CLOB, CLOB2);
        else if (MODE == 5)   // bank-aware single pair with s_nop 0 between groups
            asm volatile(REP32(STEP("v20","v24","v28","v21","v33","v25","v37","v29","v41","v35","v47","v39","v51","v43","v55","v22","v26","v30","v23","v27","v31","v32","v36","s_nop 0\n")) ::: CLOB);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0 && iters > 10) {
        const long long c1 = clock64(), w1 = wall_clock64();
        out[0] = (uint32_t)(c1 - c0); out[1] = (uint32_t)(w1 - w0);
    }
}

template<int MODE> int run(const char* name, int blocks)
{
    uint32_t* d; CK(hipMalloc(&d, (size_t)blocks * 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 300;
    static uint32_t* rows = nullptr;
    if (!rows) { CK(hipMalloc(&rows, (size_t)16384 * 36 * 4 * 64 * 4)); CK(hipMemset(rows, 0, (size_t)16384 * 36 * 4 * 64 * 4)); }
    k<MODE><<<blocks, 64>>>(d, 10, rows); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 64>>>(d, iters, rows);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double instr = (double)iters * 32 * 14;                       // VALU instructions per wave
    double waves_per_simd = blocks / 1024.0;
    double cyc = ms * 1e-3 * 2.4e9 / (instr * (waves_per_simd < 1 ? 1 : waves_per_simd));
    printf("%-40s waves=%5d  %.3f ms  %.2f T lane-ops/s  %.2f clk/instr/SIMD (at 2.4 GHz)\n", name, blocks, ms,
           instr * 64 * blocks / (ms * 1e-3) / 1e12, cyc);
    uint32_t h[2]; CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    printf("      wave 0: %u shader clocks in %u ticks of 100 MHz -> %.0f MHz\n", h[0], h[1], h[0] / (h[1] / 100.0));
    CK(hipFree(d));
    return 0;
}

int main()
{
    {
        const int waves = 8192;
        if (run<100>("align 0", waves)) return 1;
        if (run<101>("align 4", waves)) return 1;
        if (run<102>("align 8", waves)) return 1;
        if (run<103>("align 12", waves)) return 1;
        if (run<104>("align 16", waves)) return 1;
        if (run<106>("align 24", waves)) return 1;
        if (run<108>("align 32", waves)) return 1;
        if (run<109>("align 36", waves)) return 1;
        if (run<112>("align 48", waves)) return 1;
        if (run<116>("align 64", waves)) return 1;
        if (run<117>("align 68", waves)) return 1;
        if (run<124>("align 96", waves)) return 1;
    }
    for (int waves : {8192}) {
        if (run<2>("bank-aware", waves)) return 1;
        if (run<6>("bank-aware + row loads + wait", waves)) return 1;
        if (run<7>("bank-aware + row loads, no wait", waves)) return 1;
        if (run<8>("bank-aware + wait only", waves)) return 1;
        if (run<9>("bank-aware + shared-row loads + wait", waves)) return 1;
        if (run<10>("bank-aware + row loads + wait at 3/4", waves)) return 1;
    }
    return 0;
    for (int waves : {512, 1024, 2048, 4096, 8192}) {
        if (run<6>("bank-aware + row loads", waves)) return 1;
        if (run<4>("two pairs per lane, bank-aware", waves)) return 1;
        if (run<5>("bank-aware + s_nop", waves)) return 1;
        if (run<0>("naive banks", waves)) return 1;
        if (run<1>("naive banks + s_nop between groups", waves)) return 1;
        if (run<2>("bank-aware", waves)) return 1;
        if (run<3>("all same bank", waves)) return 1;
    }
    return 0;
}
