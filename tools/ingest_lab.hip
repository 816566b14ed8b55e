// Per-CU ingest microbenchmark (gfx950): how many bytes per cycle a CU can bring from L2 into LDS
//   mode 0: LDS-DMA only (global_load_lds_dwordx4, 1 KiB per wave instruction)
//   mode 1: vector loads to registers + ds_write_b128 only
//   mode 2: half the bytes on each path, interleaved
// 256-thread workgroups, `wgs_per_cu` per CU, each iteration moves 24 KB per workgroup (the P16 GEMM's 64 x 128 k-step) from a
// 4 MB L2-resident buffer.  Prints B/cycle/CU from s_memtime.   hipcc --offload-arch=gfx950 -O3 tools/ingest_lab.hip -o ingest_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define GLDS16(gp, lp) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int MODE>
__global__ __launch_bounds__(256, 2) void ingest(const char* __restrict__ src, size_t src_bytes, int iters, unsigned long long* stamps,
                                                  unsigned int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];     // 2 stages x 24 KB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int PIECES = 24;                                       // 24 KB per iteration per workgroup = 6 per wave
    const size_t wg_off = ((size_t)blockIdx.x * 24576 * 7) % (src_bytes - 24576 * 64);
    const char* p = src + wg_off + (size_t)wave * 6 * 1024 + lane * 16;
    unsigned long long t0 = 0;
    if (tid == 0) t0 = __builtin_amdgcn_s_memtime();
    unsigned int acc = 0;
    for (int it = 0; it < iters; ++it) {
        char* st = lds + (it & 1) * 24576 + wave * 6 * 1024;
        const char* q = p + (size_t)(it & 63) * 24576;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j) GLDS16(q + j * 1024, st + j * 1024);
        } else if constexpr (MODE == 1) {
            u32x4 r[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) r[j] = *reinterpret_cast<const u32x4*>(q + j * 1024);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4*>(st + j * 1024 + lane * 16) = r[j];
        } else {
            u32x4 r[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) r[j] = *reinterpret_cast<const u32x4*>(q + (3 + j) * 1024);
#pragma unroll
            for (int j = 0; j < 3; ++j) GLDS16(q + j * 1024, st + j * 1024);
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4*>(st + (3 + j) * 1024 + lane * 16) = r[j];
        }
        __syncthreads();                                             // (vmcnt(0) + barrier: the stage has landed)
        acc += *reinterpret_cast<const unsigned int*>(lds + (it & 1) * 24576 + tid * 16);
    }
    if (tid == 0) {
        stamps[blockIdx.x * 2] = t0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(const char* d_src, size_t bytes, int grid, int iters, unsigned long long* d_st, unsigned int* d_sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(ingest<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(ingest<MODE>, dim3(grid), dim3(256), 49152, 0, d_src, bytes, iters, d_st, d_sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(grid * 2);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (int i = 0; i < grid; ++i) cyc.push_back((double)(st[2 * i + 1] - st[2 * i]));
    std::sort(cyc.begin(), cyc.end());
    const double med = cyc[cyc.size() / 2];
    const int per_cu = grid / 256 > 0 ? grid / 256 : 1;
    printf("mode %d  grid %4d (%d WG/CU)  iters %d: median %.0f cycles per workgroup = %.0f cycles per 24 KB step, %.1f B/cycle/CU\n", MODE, grid,
           per_cu, iters, med, med / iters, 24576.0 * iters * per_cu / med);
}

int main() {
    const size_t bytes = 8u << 20;
    char* d_src;
    unsigned long long* d_st;
    unsigned int* d_sink;
    hipMalloc(&d_src, bytes);
    hipMemset(d_src, 1, bytes);
    hipMalloc(&d_st, 4096 * 16);
    hipMalloc(&d_sink, 64);
    for (int grid : {256, 512}) {
        run<0>(d_src, bytes, grid, 200, d_st, d_sink);
        run<1>(d_src, bytes, grid, 200, d_st, d_sink);
        run<2>(d_src, bytes, grid, 200, d_st, d_sink);
    }
    return 0;
}
