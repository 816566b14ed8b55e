// GEMM laboratory 3 (not part of the library): the fp16 two-term split GEMM with operands that are ALREADY split in memory
// (two fp16 planes per operand, the low plane scaled by 2^11), so both tiles reach LDS by LDS-DMA (global_load_lds_dwordx4)
// with no register staging and no split arithmetic in the loop.
//   hipcc --offload-arch=gfx950 -O3 -w [-DNBUF=2] tools/gemm_lab_planes.hip -o tools/lab_planes.bin
// C[M,N] = A[M,K] . W[N,K]^T
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#ifndef NBUF
#define NBUF 2
#endif
#ifndef ILV
#define ILV 0   // 1: operands stored [rows][K/32][h 32 | l 32] so every row piece is a full 128-B line
#endif
#ifndef BIG
#define BIG 0
#endif
#ifndef BUF
#define BUF 0   // 1 (with ILV): tiles by buffer_load ... lds (SGPR base + fixed per-lane offsets) instead of global_load_lds
#endif
#ifndef VAR
#define VAR 0   // ablations of the 2-stage loop: 1 = no DMA after the first tile, 2 = DMA + barriers but no fragment reads / MFMAs
#endif
#ifndef M16
#define M16 0   // 1 (ILV, 2-stage kernel): v_mfma_f32_16x16x32_f16 tiles (4 x 4 per wave) instead of 32x32x16 (2 x 2)
#endif
#ifndef MINB
#define MINB 2
#endif
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h16 = _Float16;
using h16x8 = __attribute__((ext_vector_type(8))) h16;
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PLANE_B = 128 * BK * 2;            // bytes of one plane of one operand tile (8 KiB), rows of 64 B, linear
constexpr int STAGE_B = 4 * PLANE_B;             // Ah | Al | Wh | Wl
constexpr int EPI_BYTES = 4 * 64 * 68 * 4;
constexpr int LDS_BYTES = (NBUF * STAGE_B > EPI_BYTES) ? NBUF * STAGE_B : EPI_BYTES;

__global__ void split_kernel(const float* __restrict__ x, h16* __restrict__ h, h16* __restrict__ l, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const h16 hh = (h16)fminf(fmaxf(v, -65504.f), 65504.f);
        h[i] = hh;
        l[i] = (h16)fminf(fmaxf((v - (float)hh) * 2048.0f, -65504.f), 65504.f);
    }
}
__global__ void split_ilv_kernel(const float* __restrict__ x, h16* __restrict__ o, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const h16 hh = (h16)fminf(fmaxf(v, -65504.f), 65504.f);
        const size_t g = i >> 5, e = i & 31;          // K % 32 == 0: groups never straddle rows
        o[g * 64 + e] = hh;
        o[g * 64 + 32 + e] = (h16)fminf(fmaxf((v - (float)hh) * 2048.0f, -65504.f), 65504.f);
    }
}

#define GLDS16(gp, lp) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

__global__ __launch_bounds__(256, MINB) void gemm_planes(const h16* __restrict__ Ah, const h16* __restrict__ Al,
                                                         const h16* __restrict__ Wh, const h16* __restrict__ Wl,
                                                         float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive ids of one XCD share the column panel
    const int ntm = (M + BM - 1) / BM, ntn = N / BN, nwg = ntm * ntn;
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tm = wg / ntn, tn = wg - tm * ntn;   // N-tiles of one M-tile run consecutively on one XCD (A panel stays in its L2)
    const int row0 = tm * BM, col0 = tn * BN;

    // DMA sources: a wave moves pieces 2*wave, 2*wave+1 (16 rows x 64 B each) of every plane.  Lane i of a piece fills LDS
    // bytes [16 i, 16 i + 16): row i>>2, slot i&3, which holds k-chunk (i&3) ^ ((row>>2)&3) (the read side applies the same
    // involution: rule "swizzle both sides").
#if ILV && BUF
    // buffer form: one resource per operand, per-lane byte offsets fixed for the whole kernel, the k-step enters as soffset
    auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ah, 0, (int)((size_t)M * K * 4), 0x00020000);
    auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wh, 0, (int)((size_t)N * K * 4), 0x00020000);
    int offA[4], offW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j, rr = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        offA[j] = (min(row0 + rr, M - 1) * K * 2 + chunk * 8) * 2;
        offW[j] = ((col0 + rr) * K * 2 + chunk * 8) * 2;
    }
    auto issue = [&](int kt, int buf) {
        char* base = lds + buf * STAGE_B + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(base + j * 1024), 16, offA[j], kt * 128, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(base + 2 * PLANE_B + j * 1024), 16, offW[j], kt * 128, 0, 0);
        }
    };
#elif ILV
    // piece = 8 rows x 128 B (h|l of one 32-k group); a wave moves pieces 4*wave .. 4*wave+3 of A and of W
    const h16* srcA[4]; const h16* srcW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j, rr = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        srcA[j] = Ah + (size_t)min(row0 + rr, M - 1) * K * 2 + chunk * 8;
        srcW[j] = Wh + (size_t)(col0 + rr) * K * 2 + chunk * 8;
    }
    auto issue = [&](int kt, int buf) {
        char* base = lds + buf * STAGE_B + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            GLDS16(srcA[j] + kt * 64, base + j * 1024);
            GLDS16(srcW[j] + kt * 64, base + 2 * PLANE_B + j * 1024);
        }
    };
#else
    const int prow = lane >> 2, chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const h16* srcA[2]; const h16* srcW[2];
    size_t dA, dW;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rr = (wave * 2 + j) * 16 + prow;
        srcA[j] = Ah + (size_t)min(row0 + rr, M - 1) * K + chunk * 8;
        srcW[j] = Wh + (size_t)(col0 + rr) * K + chunk * 8;
    }
    dA = Al - Ah; dW = Wl - Wh;
    auto issue = [&](int kt, int buf) {
        char* base = lds + buf * STAGE_B + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            GLDS16(srcA[j] + kt * BK, base + 0 * PLANE_B + j * 1024);
            GLDS16(srcA[j] + dA + kt * BK, base + 1 * PLANE_B + j * 1024);
            GLDS16(srcW[j] + kt * BK, base + 2 * PLANE_B + j * 1024);
            GLDS16(srcW[j] + dW + kt * BK, base + 3 * PLANE_B + j * 1024);
        }
    };
#endif
    f32x16 acc[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accx[i][j][e] = 0.f; }

#if M16
    f32x4 c16[4][4], cx[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { c16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; cx[i][j] = c16[i][j]; }
#endif
    const int fr = lane & 31, fh = lane >> 5, fswz = (fr >> 2) & 3;
    const int nk = K / BK;
    issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = (NBUF == 2) ? (kt & 1) : 0;
#if M16
        {   // one MFMA spans the whole 32-deep k-step: lane (r = lane&15, q = lane>>4) holds k = 8q .. 8q+7
            const char* sb = lds + buf * STAGE_B;
            if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
            const int r16 = lane & 15, q16 = lane >> 4;
            h16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + r16, f = (row >> 1) & 7;
                ah[i] = *reinterpret_cast<const h16x8*>(sb + row * 128 + ((q16) ^ f) * 16);
                al[i] = *reinterpret_cast<const h16x8*>(sb + row * 128 + ((4 + q16) ^ f) * 16);
                const int col = wn * 64 + i * 16 + r16, g = (col >> 1) & 7;
                bh[i] = *reinterpret_cast<const h16x8*>(sb + 2 * PLANE_B + col * 128 + ((q16) ^ g) * 16);
                bl[i] = *reinterpret_cast<const h16x8*>(sb + 2 * PLANE_B + col * 128 + ((4 + q16) ^ g) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], cx[i][j], 0, 0, 0);
                    cx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], cx[i][j], 0, 0, 0);
                    c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], c16[i][j], 0, 0, 0);
                }
        }
#elif VAR == 3
        {   // kb = 0 fragments first, THEN the next tile's requests, so the matrix pipe has work while the DMAs issue
            const char* sb = lds + buf * STAGE_B;
            const int f8 = (fr >> 1) & 7;
            h16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
            auto rd = [&](int kb) {
                const int sh = ((2 * kb + fh) ^ f8) * 16, sl = ((4 + 2 * kb + fh) ^ f8) * 16;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ro = (wm * 64 + i * 32 + fr) * 128;
                    ah[kb][i] = *reinterpret_cast<const h16x8*>(sb + ro + sh);
                    al[kb][i] = *reinterpret_cast<const h16x8*>(sb + ro + sl);
                    const int co = 2 * PLANE_B + (wn * 64 + i * 32 + fr) * 128;
                    bh[kb][i] = *reinterpret_cast<const h16x8*>(sb + co + sh);
                    bl[kb][i] = *reinterpret_cast<const h16x8*>(sb + co + sl);
                }
            };
            auto mm = [&](int kb) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb][i], bl[kb][j], accx[i][j], 0, 0, 0);
                        accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb][i], bh[kb][j], accx[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb][i], bh[kb][j], acc[i][j], 0, 0, 0);
                    }
            };
            rd(0);
            rd(1);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(0);
            mm(1);
        }
#else
        if (NBUF == 2 && kt + 1 < nk && VAR != 1) issue(kt + 1, buf ^ 1);
        const char* sb = lds + (VAR == 1 ? 0 : buf) * STAGE_B;
#if VAR != 2
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            h16x8 ah[2], al[2], bh[2], bl[2];
#if ILV
            const int f8 = (fr >> 1) & 7;
            const int sh = ((2 * kb + fh) ^ f8) * 16, sl = ((4 + 2 * kb + fh) ^ f8) * 16;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ro = (wm * 64 + i * 32 + fr) * 128;
                ah[i] = *reinterpret_cast<const h16x8*>(sb + ro + sh);
                al[i] = *reinterpret_cast<const h16x8*>(sb + ro + sl);
                const int co = 2 * PLANE_B + (wn * 64 + i * 32 + fr) * 128;
                bh[i] = *reinterpret_cast<const h16x8*>(sb + co + sh);
                bl[i] = *reinterpret_cast<const h16x8*>(sb + co + sl);
            }
#else
            const int slot = ((2 * kb + fh) ^ fswz) * 16;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ro = (wm * 64 + i * 32 + fr) * 64 + slot;
                ah[i] = *reinterpret_cast<const h16x8*>(sb + 0 * PLANE_B + ro);
                al[i] = *reinterpret_cast<const h16x8*>(sb + 1 * PLANE_B + ro);
                const int co = (wn * 64 + i * 32 + fr) * 64 + slot;
                bh[i] = *reinterpret_cast<const h16x8*>(sb + 2 * PLANE_B + co);
                bl[i] = *reinterpret_cast<const h16x8*>(sb + 3 * PLANE_B + co);
            }
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
#endif
#endif
        __syncthreads();
        if (NBUF == 1 && kt + 1 < nk) { issue(kt + 1, 0); __syncthreads(); }
    }
    // epilogue: park the wave tile in LDS, write float4 rows
    float* et = reinterpret_cast<float*>(lds) + wave * 64 * 68;
#if M16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)      // D of 16x16x32: lane (col = lane&15, row block lane>>4), register e = row 4 (lane>>4) + e
                et[(i * 16 + 4 * (lane >> 4) + e) * 68 + j * 16 + (lane & 15)] = c16[i][j][e] + cx[i][j][e] * (1.0f / 2048.0f);
#else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh, cc = j * 32 + fr;
                et[rr * 68 + cc] = acc[i][j][e] + accx[i][j][e] * (1.0f / 2048.0f);
            }
#endif
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = it * 4 + (lane >> 4), cc = (lane & 15) * 4;
        const int grow = row0 + wm * 64 + rr;
        if (grow < M) *reinterpret_cast<f32x4*>(C + (size_t)grow * N + col0 + wn * 64 + cc) = *reinterpret_cast<const f32x4*>(et + rr * 68 + cc);
    }
}


// ---- BIG: 256 x 128 tile, 8 waves (4 x 2, wave tile 64 x 64), 3-stage LDS ring with TWO tiles in flight across the
// barrier (counted vmcnt, raw s_barrier), one workgroup per CU.  ILV layout only.
constexpr int BIG_STAGE = (256 + 128) * 128;        // 48 KiB
constexpr int BIG_LDS = 3 * BIG_STAGE;              // 144 KiB (the epilogue's 8 x 64 x 68 floats = 136 KiB fit inside)
__global__ __launch_bounds__(512, 1) void gemm_planes_big(const h16* __restrict__ A, const h16* __restrict__ W,
                                                          float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntm = (M + 255) / 256, ntn = N / 128, nwg = ntm * ntn;
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tm = wg / ntn, tn = wg - tm * ntn;
    const int row0 = tm * 256, col0 = tn * 128;
    const h16* srcA[4]; const h16* srcW[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j, rr = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        srcA[j] = A + (size_t)min(row0 + rr, M - 1) * K * 2 + chunk * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wave * 2 + j, rr = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        srcW[j] = W + (size_t)(col0 + rr) * K * 2 + chunk * 8;
    }
    auto issue = [&](int st) {
        char* base = lds + st * BIG_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) { GLDS16(srcA[j], base + (wave * 4 + j) * 1024); srcA[j] += 64; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { GLDS16(srcW[j], base + 256 * 128 + (wave * 2 + j) * 1024); srcW[j] += 64; }
    };
    f32x16 acc[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accx[i][j][e] = 0.f; }
    const int fr = lane & 31, fh = lane >> 5, f8 = (fr >> 1) & 7;
    const int nk = K / BK;
    issue(0);
    if (nk > 1) issue(1);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed for this wave once at most the 6 DMAs of tile kt+1 are outstanding; lgkmcnt(0): this wave's
        // reads of tile kt-1 are complete, so after the barrier its stage may be refilled
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nk) issue(st >= 1 ? st - 1 : 2);      // stage (kt+2) % 3 == (st + 2) % 3
        const char* sb = lds + st * BIG_STAGE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            h16x8 ah[2], al[2], bh[2], bl[2];
            const int sh = ((2 * kb + fh) ^ f8) * 16, sl = ((4 + 2 * kb + fh) ^ f8) * 16;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ro = (wm * 64 + i * 32 + fr) * 128;
                ah[i] = *reinterpret_cast<const h16x8*>(sb + ro + sh);
                al[i] = *reinterpret_cast<const h16x8*>(sb + ro + sl);
                const int co = 256 * 128 + (wn * 64 + i * 32 + fr) * 128;
                bh[i] = *reinterpret_cast<const h16x8*>(sb + co + sh);
                bl[i] = *reinterpret_cast<const h16x8*>(sb + co + sl);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        st = st == 2 ? 0 : st + 1;
    }
    __syncthreads();
    float* et = reinterpret_cast<float*>(lds) + wave * 64 * 68;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh, cc = j * 32 + fr;
                et[rr * 68 + cc] = acc[i][j][e] + accx[i][j][e] * (1.0f / 2048.0f);
            }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = it * 4 + (lane >> 4), cc = (lane & 15) * 4;
        const int grow = row0 + wm * 64 + rr;
        if (grow < M) *reinterpret_cast<f32x4*>(C + (size_t)grow * N + col0 + wn * 64 + cc) = *reinterpret_cast<const f32x4*>(et + rr * 68 + cc);
    }
}

// ---- BIG=2: the 256 x 128 / 8-wave tile with the two wave groups (waves 0-3, 4-7: one of each per SIMD) running half a
// k-step apart: while one group reads its fragments from LDS the other issues its MFMAs, a barrier between phases.
//   phase 2k  : A loads fragments of tile k   | B multiplies tile k-1 | everyone requests tile k+2
//   phase 2k+1: A multiplies tile k           | B loads fragments of tile k
// 3-stage ring: tile k's stage is last read in phase 2k+1 and refilled (tile k+3) from phase 2k+2 on.
__global__ __launch_bounds__(512, 1) void gemm_planes_big8(const h16* __restrict__ A, const h16* __restrict__ W,
                                                           float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const bool grpB = wave >= 4;
    const int ntm = (M + 255) / 256, ntn = N / 128, nwg = ntm * ntn;
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int tm = wg / ntn, tn = wg - tm * ntn;
    const int row0 = tm * 256, col0 = tn * 128;
    const h16* srcA[4]; const h16* srcW[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j, rr = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        srcA[j] = A + (size_t)min(row0 + rr, M - 1) * K * 2 + chunk * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wave * 2 + j, rr = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((piece & 1) * 4 + (lane >> 4)) & 7);
        srcW[j] = W + (size_t)(col0 + rr) * K * 2 + chunk * 8;
    }
    auto issue = [&](int st) {
        char* base = lds + st * BIG_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) { GLDS16(srcA[j], base + (wave * 4 + j) * 1024); srcA[j] += 64; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { GLDS16(srcW[j], base + 256 * 128 + (wave * 2 + j) * 1024); srcW[j] += 64; }
    };
    f32x16 acc[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accx[i][j][e] = 0.f; }
    const int fr = lane & 31, fh = lane >> 5, f8 = (fr >> 1) & 7;
    const int nk = K / BK;
    h16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];      // [kb][tile]: one whole k-step of fragments
    auto load = [&](int st) {
        const char* sb = lds + st * BIG_STAGE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int sh = ((2 * kb + fh) ^ f8) * 16, sl = ((4 + 2 * kb + fh) ^ f8) * 16;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ro = (wm * 64 + i * 32 + fr) * 128;
                ah[kb][i] = *reinterpret_cast<const h16x8*>(sb + ro + sh);
                al[kb][i] = *reinterpret_cast<const h16x8*>(sb + ro + sl);
                const int co = 256 * 128 + (wn * 64 + i * 32 + fr) * 128;
                bh[kb][i] = *reinterpret_cast<const h16x8*>(sb + co + sh);
                bl[kb][i] = *reinterpret_cast<const h16x8*>(sb + co + sl);
            }
        }
    };
    auto mma = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb][i], bl[kb][j], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb][i], bh[kb][j], accx[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb][i], bh[kb][j], acc[i][j], 0, 0, 0);
                }
    };
    // prologue: tiles 0 and 1 requested.  E_k = barrier opening phase 2k (tile k landed for everyone: each wave waits for its
    // own pieces first), O_k = barrier opening phase 2k+1.  Both groups execute 2 nk + 1 barriers.
#define WAIT_TILE(k) do { if ((k) + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define PHASE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    issue(0);
    if (nk > 1) issue(1);
    if (!grpB) {
        int st = 0;
        for (int k = 0; k < nk; ++k) {
            WAIT_TILE(k);
            PHASE_BARRIER();                               // E_k
            if (k + 2 < nk) issue(st >= 1 ? st - 1 : 2);   // stage (k+2) % 3
            load(st);
            PHASE_BARRIER();                               // O_k
            mma();
            st = st == 2 ? 0 : st + 1;
        }
        PHASE_BARRIER();                                   // E_nk (group B's last multiply)
    } else {
        int st = 0;
        for (int k = 0; k < nk; ++k) {
            WAIT_TILE(k);
            PHASE_BARRIER();                               // E_k
            if (k + 2 < nk) issue(st >= 1 ? st - 1 : 2);
            if (k >= 1) mma();                             // tile k-1
            PHASE_BARRIER();                               // O_k
            load(st);                                      // tile k
            st = st == 2 ? 0 : st + 1;
        }
        PHASE_BARRIER();                                   // E_nk
        mma();                                             // tile nk-1
    }
    __syncthreads();
    float* et = reinterpret_cast<float*>(lds) + wave * 64 * 68;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh, cc = j * 32 + fr;
                et[rr * 68 + cc] = acc[i][j][e] + accx[i][j][e] * (1.0f / 2048.0f);
            }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rr = it * 4 + (lane >> 4), cc = (lane & 15) * 4;
        const int grow = row0 + wm * 64 + rr;
        if (grow < M) *reinterpret_cast<f32x4*>(C + (size_t)grow * N + col0 + wn * 64 + cc) = *reinterpret_cast<const f32x4*>(et + rr * 68 + cc);
    }
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 20480, N = argc > 2 ? atoi(argv[2]) : 1152, K = argc > 3 ? atoi(argv[3]) : 384;
    int iters = argc > 4 ? atoi(argv[4]) : 50;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((int64_t)(s >> 11) % 200001 - 100000) * 1e-5f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = rnd() * 0.1f;
    float *dA, *dW, *dC; h16 *pA, *pW;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dW, hW.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMalloc(&pA, hA.size() * 4); hipMalloc(&pW, hW.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
#if ILV
    split_ilv_kernel<<<1024, 256>>>(dA, pA, hA.size());
    split_ilv_kernel<<<1024, 256>>>(dW, pW, hW.size());
#else
    split_kernel<<<1024, 256>>>(dA, pA, pA + hA.size(), hA.size());
    split_kernel<<<1024, 256>>>(dW, pW, pW + hW.size(), hW.size());
#endif
#if BIG == 2
    const int grid = ((M + 255) / 256) * (N / BN);
    hipFuncSetAttribute((const void*)gemm_planes_big8, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
    auto run = [&]() { gemm_planes_big8<<<grid, 512, BIG_LDS>>>(pA, pW, dC, M, N, K); };
#elif BIG
    const int grid = ((M + 255) / 256) * (N / BN);
    hipFuncSetAttribute((const void*)gemm_planes_big, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
    auto run = [&]() { gemm_planes_big<<<grid, 512, BIG_LDS>>>(pA, pW, dC, M, N, K); };
#else
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    hipFuncSetAttribute((const void*)gemm_planes, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    auto run = [&]() { gemm_planes<<<grid, 256, LDS_BYTES>>>(pA, pA + hA.size(), pW, pW + hW.size(), dC, M, N, K); };
#endif
    run();
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int t = 0; t < 4000; ++t) {
        const int i = (int)((t * 7919ull) % M), j = (int)((t * 104729ull) % N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hW[(size_t)j * K + k];
        maxerr = std::max(maxerr, std::fabs(ref - hC[(size_t)i * N + j]));
        maxref = std::max(maxref, std::fabs(ref));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) run();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("planes M16=%d VAR=%d BUF=%d BIG=%d ILV=%d NBUF=%d MINB=%d M=%d N=%d K=%d grid=%d lds=%d: %.1f us  %.1f TF-eq  maxerr %.3g (max|ref| %.3g)\n", M16, VAR, BUF, BIG, ILV, NBUF, MINB, M, N, K, grid,
           LDS_BYTES, us, tf, maxerr, maxref);
    return 0;
}
