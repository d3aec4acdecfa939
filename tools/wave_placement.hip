// Where do the workgroups of a one-wave-per-workgroup launch land?  Records the hardware ids (XCC, SE, CU, SIMD) of
// every workgroup of a launch shaped like the window kernel's (64 threads, all workgroups resident at once) and
// prints how many SIMDs hold 0 / 1 / 2 / 3+ waves and which workgroup indices share a SIMD.  Evidence for DESIGN.md
// (the order of tiles in a launch decides which tiles share a SIMD).
//   hipcc --offload-arch=gfx950 -O2 tools/wave_placement.hip -o tools/wave_placement && tools/wave_placement [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

__global__ void __launch_bounds__(64) k(uint32_t *out, int spin)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t r = threadIdx.x;
    for (int i = 0; i < spin; i++) asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1" : "+v"(r));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc | (r & 0x80000000u); }
}

int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 1563;
    const int spin = argc > 2 ? atoi(argv[2]) : 4000;      // ~ 16 k instructions, like a tile of the window kernel
    uint32_t *d;
    CK(hipMalloc(&d, waves * 8));
    std::vector<uint32_t> h(2 * waves);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, spin);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost));
        std::map<uint32_t, std::vector<int>> simd;
        for (int i = 0; i < waves; i++) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const uint32_t simd_id = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            simd[(xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd_id].push_back(i);
        }
        int hist[8] = {0};
        for (auto &kv : simd) hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
        printf("launch %d: %d waves on %zu SIMDs; SIMDs with 1 / 2 / 3 / 4+ waves: %d / %d / %d / %d\n", rep, waves, simd.size(),
               hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
        {
            // does workgroup w share its SIMD exactly with the workgroups w' = w (mod #SIMDs)?
            int same = 0, total = 0;
            const int S = (int)simd.size();
            for (auto &kv : simd) { total++; bool ok = true; for (int i : kv.second) ok = ok && (i % S == kv.second[0] % S); same += ok; }
            printf("  SIMDs whose workgroups all agree mod %d: %d of %d\n", S, same, total);
        }
        if (rep == 2) {
            int shown = 0;
            std::map<int, int> gap;
            for (auto &kv : simd) if (kv.second.size() >= 2) {
                gap[kv.second[1] - kv.second[0]]++;
                if (shown++ < 12) { printf("  simd %05x:", kv.first); for (int i : kv.second) printf(" %d", i); printf("\n"); }
            }
            printf("  index distance between the first two workgroups of a SIMD (distance: count), most frequent:\n   ");
            std::vector<std::pair<int, int>> g(gap.begin(), gap.end());
            for (int n = 0; n < 12 && !g.empty(); n++) {
                size_t best = 0;
                for (size_t j = 1; j < g.size(); j++) if (g[j].second > g[best].second) best = j;
                printf(" %d: %d,", g[best].first, g[best].second);
                g.erase(g.begin() + best);
            }
            printf("\n  first 32 workgroups (xcc.se.cu.simd):");
            for (int i = 0; i < 32 && i < waves; i++) {
                const uint32_t hw = h[2 * i];
                printf(" %u.%u.%u.%u", h[2 * i + 1] & 0xf, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3);
            }
            printf("\n");
        }
    }
    hipFree(d);
    return 0;
}
