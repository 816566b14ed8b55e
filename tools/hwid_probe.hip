// Probe: which HW_ID fields tell two co-resident workgroups of one CU apart (512 WGs with 72 KiB LDS each => 2 per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = hw;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}
int main() {
    const int grid = 512;
    unsigned* d;
    (void)hipMalloc(&d, grid * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 73728, 0, d, 2000);   // 20 us spin: all WGs resident together
    std::vector<unsigned> h(grid * 2);
    (void)hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    for (int i = 0; i < grid; ++i) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        unsigned wave = hw & 0xF, simd = (hw >> 4) & 3, pipe = (hw >> 6) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 0xF;
        if (i < 24) printf("wg %3d hw %08x xcc %u se %u sh %u cu %2u tg %2u wave %2u simd %u pipe %u\n", i, hw, xcc, se, sh, cu, tg, wave, simd, pipe);
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back((tg << 8) | (wave << 4) | i % 16);
    }
    printf("distinct CUs: %zu\n", by_cu.size());
    int shown = 0;
    for (auto& kv : by_cu) {
        if (shown++ < 12) {
            printf("cu key %05x:", kv.first);
            for (int v : kv.second) printf(" tg %d wave %d |", v >> 8, (v >> 4) & 15);
            printf("\n");
        }
    }
    return 0;
}
