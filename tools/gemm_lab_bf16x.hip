// GEMM laboratory 2 (not part of the library): fp32-accurate GEMM on the bf16 matrix cores by operand splitting.
// Every fp32 operand is written as the exact sum of three bf16 terms (8+8+8 significand bits); the six products whose
// weight is above 2^-24 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (TERMS=6), or the three above 2^-16 (TERMS=3).
//   hipcc --offload-arch=gfx950 -O3 -w -DTERMS=6 tools/gemm_lab_bf16x.hip -o tools/lab_x6.bin
// C[M,N] = A[M,K] . W[N,K]^T; A is split on the fly while staging, W is pre-split (3 planes) by a prep kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#ifndef TERMS
#define TERMS 6
#endif
#ifndef VARIANT
#define VARIANT 0   // 0 full, 1 no global loads in the loop, 2 no staging/barriers (LDS reads + MFMA), 3 MFMA only
#endif
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int RS = 40;                       // row stride in bf16 elements: 64 B data + 16 B pad = 80 B (5 x 16-B slots)
constexpr int PLANE = 128 * RS;              // one plane of one operand (bf16 elements)
constexpr int NPL = (TERMS == 6) ? 3 : 2;    // planes per operand
#if TERMS == 2
#define F16S 1
using h16 = _Float16;
#else
#define F16S 0
using h16 = h16;
#endif
using h16x8 = __attribute__((ext_vector_type(8))) h16;
using h16x4 = __attribute__((ext_vector_type(4))) h16;
constexpr int STAGE_BYTES = 2 * NPL * PLANE * 2;
constexpr int EPI_BYTES = 4 * 64 * 68 * 4;
#if VARIANT == 4
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
#else
constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
#endif

__device__ __forceinline__ void split3(float x, h16& h, h16& m, h16& l) {
#if F16S
    const float xc = fminf(fmaxf(x, -65504.f), 65504.f);
    h = (h16)xc;
    m = (h16)fminf(fmaxf((x - (float)h) * 2048.0f, -65504.f), 65504.f);    // scaled residual
    l = (h16)0.f;
#else
    h = (h16)x;
    const float r1 = x - (float)h;
    m = (h16)r1;
    const float r2 = r1 - (float)m;
    l = (h16)r2;
#endif
}

__global__ void split_w_kernel(const float* __restrict__ w, h16* __restrict__ planes, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        h16 h, m, l;
        split3(w[i], h, m, l);
        planes[i] = h; planes[n + i] = m; planes[2 * n + i] = l;
    }
}

#if VARIANT == 4
#define LB 1
#else
#define LB 2
#endif
__global__ __launch_bounds__(256, LB) void lab_kernel(const float* __restrict__ A, const h16* __restrict__ Wp, float* __restrict__ C, int M,
                                                     int N, int K, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) h16 lds[];
    h16* As = lds;                       // [NPL][128][RS]
    h16* Bs = lds + NPL * PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int n_tiles = N / BN;
    const int nwg = gridDim.x, id = blockIdx.x, xcd = id & 7, qq = nwg >> 3, rr = nwg & 7;
    const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (id >> 3);
    const int m0 = (bid / n_tiles) * BM, n0 = (bid % n_tiles) * BN;
    // A staging: thread -> rows lrow + 32 i, float4 at k = lq
    const int lrow = tid >> 3, lq = (tid & 7) * 4;
    const float* ap = A + (size_t)(m0 + lrow) * K + lq;
    // W staging: per plane 128 rows x 32 bf16 = 512 chunks of 16 B: thread -> row (tid>>2) + 64 j, chunk (tid&3)*8
    const int wr = tid >> 2, wc = (tid & 3) * 8;
    const size_t wplane = (size_t)N * K;
    const h16* wp = Wp + (size_t)(n0 + wr) * K + wc;
    f32x4 ra[4];
    h16x8 rw[NPL][2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)(32 * i) * K + k0);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) rw[pl][j] = *reinterpret_cast<const h16x8*>(wp + pl * wplane + (size_t)(64 * j) * K + k0);
    };
    auto stage = [&](int buf) {
        h16* As = lds + buf * (STAGE_BYTES / 2);
        h16* Bs = As + NPL * PLANE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h16 a, b, c; split3(ra[i][e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
            h16* d = As + (lrow + 32 * i) * RS + lq;
            *reinterpret_cast<h16x4*>(d) = h;
            *reinterpret_cast<h16x4*>(d + PLANE) = m;
            if (NPL == 3) *reinterpret_cast<h16x4*>(d + 2 * PLANE) = l;
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<h16x8*>(Bs + pl * PLANE + (wr + 64 * j) * RS + wc) = rw[pl][j];
    };
    f32x16 acc[2][2], accx[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
    const int nk = K / BK;
    fetch(0);
    const int frag = (lane & 31) * RS + 8 * (lane >> 5);
    unsigned long long t0 = 0, t1 = 0;
#if VARIANT == 4
    stage(0);
    __syncthreads();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        if (kt == 1) t0 = __builtin_amdgcn_s_memtime();
#if VARIANT == 4
        const int cur = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * BK);
        const h16* As = lds + cur * (STAGE_BYTES / 2);
        const h16* Bs = As + NPL * PLANE;
#elif VARIANT <= 1
        if (kt) __syncthreads();          // everyone finished reading the previous tile
        stage(0);
        __syncthreads();
#else
        if (kt == 0) { stage(0); __syncthreads(); }
#endif
#if VARIANT == 0
        if (kt + 1 < nk) fetch((kt + 1) * BK);
#endif
        const h16* Aw = As + (wm * 64) * RS + frag;
        const h16* Bw = Bs + (wn * 64) * RS + frag;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            h16x8 a[NPL][2], b[NPL][2];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#if VARIANT == 3
                    a[pl][i] = rw[pl][i]; b[pl][i] = rw[pl][1 - i];
#else
                    a[pl][i] = *reinterpret_cast<const h16x8*>(Aw + pl * PLANE + i * 32 * RS + kb * 16);
                    b[pl][i] = *reinterpret_cast<const h16x8*>(Bw + pl * PLANE + i * 32 * RS + kb * 16);
#endif
                }
#if F16S
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], accx[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
                }
#else
            // smallest terms first
#pragma unroll
            for (int s = NPL - 1; s >= 0; --s)          // s = pa + pb
#pragma unroll
                for (int pa = 0; pa <= s; ++pa) {
                    const int pb = s - pa;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[pb][j], acc[i][j], 0, 0, 0);
                }
#endif
        }
#if VARIANT == 4
        if (kt + 1 < nk) stage((kt + 1) & 1);
        __syncthreads();
#endif
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0 && blockIdx.x < 512) clk[blockIdx.x] = t1 - t0;
    // epilogue as in the product kernel: park the wave tile in LDS, store float4 rows
    constexpr int CS = 68;
    __syncthreads();
    float* Cw = reinterpret_cast<float*>(lds) + wave * (64 * CS);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Cw[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + j * 32 + (lane & 31)] = F16S ? acc[i][j][r] + accx[i][j][r] * (1.0f / 2048.0f) : acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int it = 0; it < 16; ++it) {
        const int rl = it * 4 + (lane >> 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(Cw + rl * CS + (lane & 15) * 4);
        *reinterpret_cast<f32x4*>(C + (size_t)(m0 + wm * 64 + rl) * N + n0 + wn * 64 + (lane & 15) * 4) = v;
    }
}

int main() {
    struct Sh { int M, N, K; } shapes[] = {{20480, 384, 1536}, {20480, 1536, 384}, {20480, 384, 384}, {32768, 128, 1536}, {65536, 128, 1536}, {20480, 1152, 384}};
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lab_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    printf("bf16-split GEMM, TERMS %d, VARIANT %d, planes %d, LDS %d B\n", TERMS, VARIANT, NPL, LDS_BYTES);
    for (auto sh : shapes) {
        float *A, *W, *C;
        h16* Wp;
        unsigned long long* clk;
        (void)hipMalloc(&clk, 512 * 8);
        (void)hipMalloc(&A, sizeof(float) * (size_t)sh.M * sh.K);
        (void)hipMalloc(&W, sizeof(float) * (size_t)sh.N * sh.K);
        (void)hipMalloc(&Wp, 2 * 3 * (size_t)sh.N * sh.K);
        (void)hipMalloc(&C, sizeof(float) * (size_t)sh.M * sh.N);
        std::vector<float> ha((size_t)sh.M * sh.K), hw((size_t)sh.N * sh.K);
        for (auto& v : ha) v = (rand() % 20001 - 10000) * 1e-4f * (1.0f + (rand() % 7) * 0.37f);
        for (auto& v : hw) v = (rand() % 20001 - 10000) * 1e-4f / sqrtf((float)sh.K);
        (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(split_w_kernel, dim3(1024), dim3(256), 0, 0, W, Wp, (size_t)sh.N * sh.K);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        const int grid = (sh.M / BM) * (sh.N / BN), reps = 20;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(lab_kernel, dim3(grid), dim3(256), LDS_BYTES, 0, A, Wp, C, sh.M, sh.N, sh.K, clk);
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(lab_kernel, dim3(grid), dim3(256), LDS_BYTES, 0, A, Wp, C, sh.M, sh.N, sh.K, clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        // accuracy on a sample of outputs against fp64
        std::vector<float> hc((size_t)sh.M * sh.N);
        (void)hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int s = 0; s < 4000; ++s) {
            const int m = rand() % sh.M, n = rand() % sh.N;
            double ref = 0;
            for (int k = 0; k < sh.K; ++k) ref += (double)ha[(size_t)m * sh.K + k] * (double)hw[(size_t)n * sh.K + k];
            maxerr = fmax(maxerr, fabs(ref - hc[(size_t)m * sh.N + n]));
            maxref = fmax(maxref, fabs(ref));
        }
        std::vector<unsigned long long> hcv(512);
        (void)hipMemcpy(hcv.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
        const int nb = grid < 512 ? grid : 512;
        std::sort(hcv.begin(), hcv.begin() + nb);
        unsigned long long hc2[2] = {hcv[nb / 2], 0};
        printf("   per-block loop cycles/k-step: min %.0f median %.0f max %.0f\n", (double)hcv[0] / (sh.K / BK - 1), (double)hcv[nb / 2] / (sh.K / BK - 1), (double)hcv[nb - 1] / (sh.K / BK - 1));
        printf("M %6d N %5d K %5d grid %5d  %8.1f us  %6.1f TFLOP/s (fp32-equivalent)   %6.0f cycles/k-step (MFMA %d)   max err %.3e\n", sh.M, sh.N, sh.K, grid, us,
               2.0 * sh.M * sh.N * sh.K / us / 1e6, (double)hc2[0] / (sh.K / BK - 1), 8 * (TERMS == 2 ? 3 : TERMS) * 32, maxerr);
        (void)hipFree(A); (void)hipFree(W); (void)hipFree(Wp); (void)hipFree(C);
    }
    return 0;
}
