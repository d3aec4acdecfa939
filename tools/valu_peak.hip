// Microbenchmark: measured integer-VALU peak on gfx950 (lane-ops/s) for the op mix the
// bit-parallel LCS kernels use (v_xor, v_or, v_bfi, v_and_or, v_add_co/v_addc_co chains).
// Evidence for the roofline peak quoted in DESIGN.md.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template<int MODE>
__global__ void __launch_bounds__(256) k_ops(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[8], b[8];
    #pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1) + i; b[i] = seed ^ (i * 0x9e3779b9u + threadIdx.x); }
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int r = 0; r < 8; r++) {
            if (MODE == 0) {            // independent v_xor / v_add
                #pragma unroll
                for (int i = 0; i < 8; i++) { a[i] ^= b[i]; b[i] += a[i]; }
            } else if (MODE == 1) {     // bfi + and_or
                #pragma unroll
                for (int i = 0; i < 8; i++) {
                    a[i] = (a[i] & b[i]) | (~a[i] & b[(i + 1) & 7]);
                    b[i] = (b[i] & a[i]) | a[(i + 3) & 7];
                }
            } else {                    // 256-bit add with carry chain (v_add_co + 7 x v_addc_co)
                uint64_t c = 0;
                #pragma unroll
                for (int i = 0; i < 8; i++) { uint64_t s = (uint64_t)a[i] + b[i] + c; a[i] = (uint32_t)s; c = s >> 32; }
                #pragma unroll
                for (int i = 0; i < 8; i++) b[i] ^= a[i];
            }
        }
    }
    uint32_t x = 0;
    #pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template<int MODE> int run(const char* name, int blocks, double ops_per_iter)
{
    uint32_t* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 4000;
    k_ops<MODE><<<blocks, 256>>>(d, 100, 12345u); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_ops<MODE><<<blocks, 256>>>(d, iters, 12345u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double lane_ops = (double)blocks * 256 * iters * ops_per_iter;
    printf("%-28s blocks=%5d  %.3f ms  %.2f T lane-ops/s\n", name, blocks, ms, lane_ops / (ms * 1e-3) / 1e12);
    CK(hipFree(d));
    return 0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s  CUs=%d  clock=%d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    for (int wpc : {4, 8, 16, 32}) {   // waves per CU
        int blocks = p.multiProcessorCount * wpc / 4;
        printf("-- %d waves/CU\n", wpc);
        if (run<0>("xor+add (indep)", blocks, 8 * 16.0)) return 1;
        if (run<1>("bfi+and_or", blocks, 8 * 16.0)) return 1;
        if (run<2>("addc chain x8 + 8 xor", blocks, 8 * 16.0)) return 1;
    }
    return 0;
}
