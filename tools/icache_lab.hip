// Instruction-fetch cost of straight-line code on gfx950 (diagnostic; tools/, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/icache_lab.hip -o tools/icache_lab.bin && tools/icache_lab.bin
// Kernel `line<N>`: N x 8-byte VALU instructions executed once per wave, straight-line.  Kernel `loop<N>`: the same N
// instructions as 64-instruction loop body.  Both launched alternately with a third kernel in between (so that no launch
// follows itself), 512 workgroups x 256 threads, time per launch by events and cycles per wave by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define I1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define I4 I1 I1 I1 I1
#define I16 I4 I4 I4 I4
#define I64 I16 I16 I16 I16
#define I256 I64 I64 I64 I64
#define I1024 I256 I256 I256 I256
template <int N> __global__ void line(float* out, unsigned long long* cyc, float a, float b) {
    float x = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (N >= 1024) { I1024 }
    if constexpr (N >= 2048) { I1024 }
    if constexpr (N >= 4096) { I1024 I1024 }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int N> __global__ void loop(float* out, unsigned long long* cyc, float a, float b) {
    float x = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N / 64; ++i) { I64 }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void other(float* out, float a, float b) {       // a different 8 KB of code between the measured launches
    float x = threadIdx.x;
    I1024
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + 1.f;
}
template <typename K> static void run(const char* name, K kern, int n, float* out, unsigned long long* cyc, bool between) {
    const int G = 512;
    std::vector<unsigned long long> h(G);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f; double med = 0;
    for (int rep = 0; rep < 20; ++rep) {
        if (between) hipLaunchKernelGGL(other, dim3(G), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(G), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end()); med = (double)h[G / 2];
    }
    printf("%-10s N=%5d %s: %7.1f us/launch  median %8.0f cycles per wave = %5.2f cycles/instr (last rep)\n", name, n, between ? "cold" : "warm", best * 1e3, med, med / n);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8);
    for (int between = 0; between < 2; ++between) {
        run("line", line<1024>, 1024, out, cyc, between); run("loop", loop<1024>, 1024, out, cyc, between);
        run("line", line<2048>, 2048, out, cyc, between); run("loop", loop<2048>, 2048, out, cyc, between);
        run("line", line<4096>, 4096, out, cyc, between); run("loop", loop<4096>, 4096, out, cyc, between);
    }
    return 0;
}
