// Calibration tool (not part of the library): sustained v_mfma_f32_32x32x2_f32 rate on this device, i.e. the
// ceiling any fp32 GEMM here can reach at the clock the chip holds under MFMA load.  Operand data matters (DVFS):
// mode 0 = constant operands, mode 1 = 16 different random operand registers per lane, mode 2 = zeros.
//   hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak.hip -o tools/mfma_peak.bin && ./tools/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ inline float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) * (2.0f / 65536.0f) - 1.0f; }

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void peak_kernel(float* out, int iters, int mode, unsigned long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    unsigned s = threadIdx.x * 7919u + blockIdx.x * 104729u + 1u;
    for (int i = 0; i < 8; ++i) {
        a[i] = mode == 1 ? rnd(s) : mode == 0 ? 0.37f : 0.f;
        b[i] = mode == 1 ? rnd(s) : mode == 0 ? -0.21f : 0.f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 1) & 7], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[u], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[(u + 1) & 7], acc[3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int WAVES>
void run(const char* name, int blocks_per_cu, int mode) {
    const int iters = 20000, grid = 256 * blocks_per_cu;
    float* out;
    unsigned long long* clk;
    (void)hipMalloc(&out, sizeof(float) * grid * 64 * WAVES);
    (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(peak_kernel<WAVES>, dim3(grid), dim3(64 * WAVES), 0, 0, out, iters, mode, clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flops = 2.0 * 32 * 32 * 2 * 32.0 * iters * WAVES * grid;
        if (rep == 2)
            printf("%-32s mode %d %8.3f ms  %7.1f TFLOP/s  in-kernel clock %.0f MHz\n", name, mode, ms, flops / ms / 1e9,
                   (double)h[0] / (double)h[1] * 100.0);
    }
    (void)hipFree(out);
    (void)hipFree(clk);
}

int main() {
    for (int mode = 0; mode < 3; ++mode) {
        run<4>("4 waves/CU (1 per SIMD)", 1, mode);
        run<4>("8 waves/CU (2 blocks x 4 waves)", 2, mode);
    }
    return 0;
}
