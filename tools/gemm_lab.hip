// GEMM laboratory (not part of the library): ablations of the fp32-MFMA GEMM main loop to locate what keeps the
// product kernel below the MFMA rate.  C[M,N] = A[M,K] . W[N,K]^T, all dims multiples of the tile.
//   hipcc --offload-arch=gfx950 -O3 -DVARIANT=0 tools/gemm_lab.hip -o /tmp/lab0
// VARIANT 0: full pipeline (global -> regs -> LDS double buffer -> MFMA)      1: no global loads in the loop
//         2: no LDS writes / barrier in the loop (MFMA + ds_read only)        3: MFMA only (operands in registers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef BK
#define BK 32
#endif
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BM = 128, BN = 128, LS = BK + 4;
constexpr int TILE = (BM + BN) * LS;
constexpr int NL = BK / 8;   // float4 loads per row-slot: threads cover 8 float4 per 32 floats

__global__ __launch_bounds__(256, 2) void lab_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M,
                                                     int N, int K, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int n_tiles = N / BN;
#ifdef XCD_SWZ
    const int nwg = gridDim.x, id = blockIdx.x, xcd = id & 7, qq = nwg >> 3, rr = nwg & 7;
    const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (id >> 3);
#else
    const int bid = blockIdx.x;
#endif
    const int m0 = (bid / n_tiles) * BM, n0 = (bid % n_tiles) * BN;
    constexpr int TPR = BK / 4;            // threads per row
    constexpr int RPP = 256 / TPR;         // rows per pass
    constexpr int NP = BM / RPP;           // passes
    const int lrow = tid / TPR, lq = (tid % TPR) * 4;
    const float* ap = A + (size_t)(m0 + lrow) * K + lq;
    const float* wp = W + (size_t)(n0 + lrow) * K + lq;
    f32x4 ra[NP], rb[NP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)(RPP * i) * K + k0);
            rb[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)(RPP * i) * K + k0);
        }
    };
    auto stage = [&](int buf) {
        float* As = lds + buf * TILE;
        float* Bs = As + BM * LS;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            *reinterpret_cast<f32x4*>(As + (lrow + RPP * i) * LS + lq) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + (lrow + RPP * i) * LS + lq) = rb[i];
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;
    fetch(0);
    stage(0);
    __syncthreads();
    const int frag_off = (lane & 31) * LS + 4 * (lane >> 5);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
#if VARIANT == 0
        if (more) fetch((kt + 1) * BK);
#endif
        const int buf = (VARIANT >= 2) ? 0 : (kt & 1);
        const float* Aw = lds + buf * TILE + (wm * 64) * LS + frag_off;
        const float* Bw = lds + buf * TILE + (BM + wn * 64) * LS + frag_off;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 a[2], b[2];
#if VARIANT == 3
            a[0] = ra[0]; a[1] = ra[1]; b[0] = rb[0]; b[1] = rb[1];
#else
            a[0] = *reinterpret_cast<const f32x4*>(Aw + 8 * g);
            a[1] = *reinterpret_cast<const f32x4*>(Aw + 32 * LS + 8 * g);
            b[0] = *reinterpret_cast<const f32x4*>(Bw + 8 * g);
            b[1] = *reinterpret_cast<const f32x4*>(Bw + 32 * LS + 8 * g);
#endif
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
        }
#if VARIANT <= 1
        if (more) stage((kt + 1) & 1);
        __syncthreads();
#endif
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 77 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    // epilogue as in the product kernel: park the wave tile in LDS, store float4 rows
    constexpr int CS = 68;
    __syncthreads();
    float* Cw = lds + wave * (64 * CS);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Cw[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + j * 32 + (lane & 31)] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int it = 0; it < 16; ++it) {
        const int rl = it * 4 + (lane >> 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(Cw + rl * CS + (lane & 15) * 4);
        *reinterpret_cast<f32x4*>(C + (size_t)(m0 + wm * 64 + rl) * N + n0 + wn * 64 + (lane & 15) * 4) = v;
    }
}

int main(int argc, char** argv) {
    struct Sh { int M, N, K; } shapes[] = {{20480, 384, 1536}, {20480, 1536, 384}, {20480, 384, 384}, {10240, 384, 1152}, {32768, 128, 1536}, {65536, 128, 1536}};
    const int lds_bytes = 2 * TILE * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(lab_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    printf("VARIANT %d BK %d lds %d\n", VARIANT, BK, lds_bytes);
    for (auto sh : shapes) {
        float *A, *W, *C;
        unsigned long long* clk;
        hipMalloc(&clk, 16);
        hipMalloc(&A, sizeof(float) * (size_t)sh.M * sh.K);
        hipMalloc(&W, sizeof(float) * (size_t)sh.N * sh.K);
        hipMalloc(&C, sizeof(float) * (size_t)sh.M * sh.N);
        std::vector<float> h((size_t)sh.M * sh.K);
        for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
        hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)sh.N * sh.K * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int grid = (sh.M / BM) * (sh.N / BN), reps = 20;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(lab_kernel, dim3(grid), dim3(256), lds_bytes, 0, A, W, C, sh.M, sh.N, sh.K, clk);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(lab_kernel, dim3(grid), dim3(256), lds_bytes, 0, A, W, C, sh.M, sh.N, sh.K, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        unsigned long long hc[2];
        hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
        printf("M %6d N %5d K %5d grid %5d  %8.1f us  %6.1f TFLOP/s   loop: %.0f cycles/k-step (ideal %d), clock %.0f MHz, loop %.1f us\n", sh.M, sh.N, sh.K, grid, us,
               2.0 * sh.M * sh.N * sh.K / us / 1e6, (double)hc[0] / (sh.K / BK), 64 * 64 * BK / 32, (double)hc[0] / (double)hc[1] * 100.0, (double)hc[1] / 100.0);
        hipFree(A); hipFree(W); hipFree(C);
    }
    return 0;
}
