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
#ifndef TERMS
#define TERMS 6
#endif
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int RS = 40;                       // row stride in bf16 elements: 64 B data + 16 B pad = 80 B (5 x 16-B slots)
constexpr int PLANE = 128 * RS;              // one plane of one operand (bf16 elements)
constexpr int NPL = (TERMS == 6) ? 3 : 2;    // planes per operand
constexpr int STAGE_BYTES = 2 * NPL * PLANE * 2;
constexpr int EPI_BYTES = 4 * 64 * 68 * 4;
constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

__global__ void split_w_kernel(const float* __restrict__ w, __bf16* __restrict__ planes, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        __bf16 h, m, l;
        split3(w[i], h, m, l);
        planes[i] = h; planes[n + i] = m; planes[2 * n + i] = l;
    }
}

__global__ __launch_bounds__(256, 2) void lab_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, float* __restrict__ C, int M,
                                                     int N, int K) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16* As = lds;                       // [NPL][128][RS]
    __bf16* Bs = lds + NPL * PLANE;
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
    const __bf16* wp = Wp + (size_t)(n0 + wr) * K + wc;
    f32x4 ra[4];
    bf16x8 rw[NPL][2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)(32 * i) * K + k0);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) rw[pl][j] = *reinterpret_cast<const bf16x8*>(wp + pl * wplane + (size_t)(64 * j) * K + k0);
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(ra[i][e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
            __bf16* d = As + (lrow + 32 * i) * RS + lq;
            *reinterpret_cast<bf16x4*>(d) = h;
            *reinterpret_cast<bf16x4*>(d + PLANE) = m;
            if (NPL == 3) *reinterpret_cast<bf16x4*>(d + 2 * PLANE) = l;
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<bf16x8*>(Bs + pl * PLANE + (wr + 64 * j) * RS + wc) = rw[pl][j];
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;
    fetch(0);
    const int frag = (lane & 31) * RS + 8 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt) __syncthreads();          // everyone finished reading the previous tile
        stage();
        __syncthreads();
        if (kt + 1 < nk) fetch((kt + 1) * BK);
        const __bf16* Aw = As + (wm * 64) * RS + frag;
        const __bf16* Bw = Bs + (wn * 64) * RS + frag;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 a[NPL][2], b[NPL][2];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[pl][i] = *reinterpret_cast<const bf16x8*>(Aw + pl * PLANE + i * 32 * RS + kb * 16);
                    b[pl][i] = *reinterpret_cast<const bf16x8*>(Bw + pl * PLANE + i * 32 * RS + kb * 16);
                }
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
        }
    }
    // epilogue as in the product kernel: park the wave tile in LDS, store float4 rows
    constexpr int CS = 68;
    __syncthreads();
    float* Cw = reinterpret_cast<float*>(lds) + wave * (64 * CS);
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

int main() {
    struct Sh { int M, N, K; } shapes[] = {{20480, 384, 1536}, {20480, 1536, 384}, {20480, 384, 384}, {20480, 384, 1152}, {10240, 384, 1152}, {20480, 1152, 384}};
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lab_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    printf("bf16-split GEMM, TERMS %d, planes %d, LDS %d B\n", TERMS, NPL, LDS_BYTES);
    for (auto sh : shapes) {
        float *A, *W, *C;
        __bf16* Wp;
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
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(lab_kernel, dim3(grid), dim3(256), LDS_BYTES, 0, A, Wp, C, sh.M, sh.N, sh.K);
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(lab_kernel, dim3(grid), dim3(256), LDS_BYTES, 0, A, Wp, C, sh.M, sh.N, sh.K);
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
        printf("M %6d N %5d K %5d grid %5d  %8.1f us  %6.1f TFLOP/s (fp32-equivalent)   max err %.3e (|ref| max %.2f)\n", sh.M, sh.N, sh.K, grid, us,
               2.0 * sh.M * sh.N * sh.K / us / 1e6, maxerr, maxref);
        (void)hipFree(A); (void)hipFree(W); (void)hipFree(Wp); (void)hipFree(C);
    }
    return 0;
}
