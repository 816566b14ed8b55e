// fp32 MFMA GEMM / implicit 1-D convolution for gfx950 (CDNA4).
//
//   C[M,N] = epilogue( prologue(A)[M,K] . W[N,K]^T )        M = B*T_out rows of channels-last activations
//
// Every Linear and Conv1d/ConvTranspose1d of the reference path runs on this one kernel:
//   Linear                  reference transformer.py:104-120,249-261 (to_q/k/v, to_out, FeedForward)
//   Conv1d k3 / k5 / 1x1    reference decoder.py:32-72,252-254,301-310; text_encoder.py:30-62,101-112,210-258
//   ConvTranspose1d k4 s2   reference decoder.py:146 (two phase-GEMMs of two taps each)
// A convolution is a GEMM whose K axis is (tap, channel): the A tile of tap j is the same activation matrix read at
// rows shifted by tap_off[j] (zero outside [0,T_in)), so no im2col buffer exists.  Channel concatenation
// (pack([x, mu]) decoder.py:371, skip connections decoder.py:410) is two K segments read from two tensors.
//
// Tiling (one 256-thread workgroup = 4 waves, 2x2):
//   block tile 128x128x32 (or 64x128x32 for small grids), wave tile 64x64 = 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 VGPRs)
//   global -> registers (prefetch of tile k+1 issued before the MFMAs of tile k) -> LDS, two LDS buffers, one barrier per k-step
//   LDS rows are K-contiguous, stride 36 floats: the 16-byte fragment reads (ds_read_b128) and tile writes are conflict free
//   inside a group of 8 k the MFMA kk consumes k = {kk, 4+kk}: lane half h owns k = 4h..4h+3, i.e. one ds_read_b128
//   feeds four MFMAs per operand (the k permutation is the same for A and W, so the sum is unchanged)
// Arithmetic, template parameter TERMS:
//   0  v_mfma_f32_32x32x2_f32: fp32 inputs, an exact fp32 FMA chain (157 TFLOP/s dense peak).
//   6  fp32-equivalent on the bf16 matrix cores: every operand is the exact sum of three bf16 terms (8+8+8 significand
//      bits, A split while staging, W pre-split in the weight image) and the six cross products whose weight is above
//      2^-24 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (products of bf16 values are exact in fp32).  The
//      dropped products are below fp32 rounding, so the result differs from the fp32 chain by summation order only
//      (measured: same error vs fp64), at 6/16 of the fp32-MFMA cycles.
//   3  two-term bf16 split, three products (weights above 2^-16): ~2^-17 relative per product; opt-in.
//   2  two-term fp16 split with a scaled residual: x ~ h + l/2^11, h = fp16(x), l = fp16((x - h) * 2^11) (22 significand
//      bits; the scaling keeps the residual out of the fp16 subnormal range), products h.h into one accumulator and
//      h.l + l.h into a second one that is folded in with 2^-11 in the epilogue: three v_mfma_f32_32x32x16_f16 per 16-k
//      block (3/16 of the fp32-MFMA cycles).  Measured error vs fp64 equals the fp32 chain's.  Inputs beyond the fp16
//      range (|x| > 65504) saturate instead of overflowing.
#include "kernels.h"
#include <string>
#include "device_utils.h"
#include "gemm_epilogue.h"

#include <cstdlib>
#include <cstring>

namespace mtts {

thread_local const char* g_kernel_tag = nullptr;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int LDS_STRIDE = GEMM_BK + 4;                              // 36 floats = 144 B (9 x 16 B)
// one (A,B) stage of a BM x 128 block tile; two stages: BM=128 -> 73,728 B, BM=64 -> 55,296 B (2 workgroups per CU)
constexpr int tile_floats(int BM) { return (BM + GEMM_BN) * LDS_STRIDE; }
// Split mode (TERMS = 3 or 6, see below): one stage of NPL bf16 planes per operand, rows of 32 bf16 + 16 B pad = 80 B
// (5 x 16-byte slots: the 16-byte fragment reads of 32 rows hit 16 different slots), single-buffered; the epilogue tile
// (4 waves x BM/2 rows x 68 floats) reuses the same LDS and is the larger of the two at BM = 128.
constexpr int SPLIT_RS = 40;                                         // row stride in bf16 elements
constexpr int split_planes(int terms) { return terms == 6 ? 3 : 2; }
constexpr int epi_bytes(int BM) { return 4 * (BM / 2) * 68 * 4; }
constexpr int gemm_lds_bytes(int BM, int terms) {
    const int stage = terms == 0 ? 2 * tile_floats(BM) * 4 : split_planes(terms) * (BM + GEMM_BN) * SPLIT_RS * 2;
    return stage > epi_bytes(BM) ? stage : epi_bytes(BM);
}

// x = h + m + l exactly (three round-to-nearest bf16 terms of 8 significand bits each; the subtractions are exact)
__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// BM = 128: wave tile 64x64 (2x2 MFMA tiles).  BM = 64: wave tile 32x64 (1x2), used when the 128-row grid would leave
// CUs with a single resident workgroup (nothing to overlap its staging with).
__device__ __forceinline__ int round_up_dev(int n) { return (n + GEMM_BN - 1) / GEMM_BN * GEMM_BN; }

// Registers: the 64-row split-mode tile is held to 168 VGPRs (3 waves per SIMD = 3 workgroups per CU; its LDS footprint
// of 35 KB allows 4) -- with the short 16-bit MFMA phases more resident workgroups hide the staging better.
template <int BM, bool A_MASK, bool A_NORM, int TERMS>
__global__ __launch_bounds__(256, (BM == 64 && TERMS != 0) ? 3 : 2) void gemm_f32_kernel(const GemmArgs p) {
    constexpr int MI = BM / 64;                 // 32-row MFMA tiles per wave along M
    constexpr int AR = BM / 32;                 // A rows staged per thread
    constexpr int TILE_FLOATS = tile_floats(BM);
    constexpr int NPL = split_planes(TERMS);    // bf16 planes per operand (split mode)
    constexpr int APLANE = BM * SPLIT_RS, BPLANE = GEMM_BN * SPLIT_RS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = p.B * p.T_out;
    const int Kp = p.ntaps * p.ktap;
    const int n_tiles = (p.N + GEMM_BN - 1) / GEMM_BN;

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run of
    // tiles (the N-tiles of one M-tile share their A rows in that XCD's L2).  Bijective for any grid size.
    int swz;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int m0 = (swz / n_tiles) * BM;
    const int n0 = (swz % n_tiles) * GEMM_BN;

    // ---- per-thread staging coordinates: 4 rows x one float4 of K for A and for W
    const int lrow = tid >> 3;
    const int lq = (tid & 7) * 4;
    int arow_base[AR], at[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < M) {
            const int b = m / p.T_out;
            arow_base[i] = b * p.T_in;
            at[i] = (m - b * p.T_out) * p.in_stride;
        } else {
            arow_base[i] = 0;
            at[i] = -(1 << 28);
        }
    }
    const float* wrow = p.w + (size_t)(n0 + lrow) * Kp + lq;
    // split mode: W planes [NPL of 3][Np][Kp] bf16; a thread stages rows (tid>>2) + 64 j, 8 bf16 (16 B) at k = (tid&3)*8
    const int wr16 = tid >> 2, wc16 = (tid & 3) * 8;
    const size_t wplane = (size_t)round_up_dev(p.N) * Kp;
    const __bf16* wrow16 = reinterpret_cast<const __bf16*>(p.w16) + (size_t)(n0 + wr16) * Kp + wc16;
    // TERMS = 2: interleaved image [Np][Kp/32][h 32 | l 32]; a thread stages rows (tid>>3) + 32 j, 16-B chunk tid&7 of the line
    const int wr8 = tid >> 3, wc8 = tid & 7;
    const __bf16* wrow8 = reinterpret_cast<const __bf16*>(p.w16) + (size_t)(n0 + wr8) * Kp * 2 + wc8 * 8;

    // Tile fetch: every load is issued unconditionally (out-of-range rows read row 0 and are zeroed later), no load
    // depends on another load's result, and the mask / LayerNorm transform is deferred to the LDS write -- so the 8-20
    // loads of tile k+1 are all in flight under the MFMAs of tile k.
    f32x4 ra[AR], rb[TERMS == 0 ? 4 : 1];
    bf16x8 rw[TERMS == 0 ? 1 : NPL][2];
    float r_mask[AR], r_mean[AR], r_rstd[AR];
    bool r_ok[AR];
    int ld_tap = 0, ld_c = 0;   // (tap, channel) of the next tile to fetch; a 32-wide K chunk never straddles segments
    auto fetch = [&]() {
        const bool seg1 = p.a1 != nullptr && ld_c >= p.c0;          // wave-uniform
        const float* src = seg1 ? p.a1 : p.a0;
        const int ld = seg1 ? p.lda1 : p.lda0;
        const int cc = (seg1 ? ld_c - p.c0 : ld_c) + lq;
        const bool cvalid = cc < (seg1 ? p.c1 : p.c0);
        const int off = p.tap_off[ld_tap];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int tin = at[i] + off;
            const bool ok = cvalid && (unsigned)tin < (unsigned)p.T_in;
            const int row = ok ? arow_base[i] + tin : 0;
            ra[i] = *reinterpret_cast<const f32x4*>(src + (size_t)row * ld + (ok ? cc : 0));
            r_ok[i] = ok;
            if (A_MASK) { r_mask[i] = p.a_mask[row]; }
        }
        if constexpr (TERMS == 0) {
            const float* wp = wrow + ld_tap * p.ktap + ld_c;
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)(32 * i) * Kp);
        } else if constexpr (TERMS == 2) {
            const __bf16* wp = wrow8 + (ld_tap * p.ktap + ld_c) * 2;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) rw[jj >> 1][jj & 1] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(32 * jj) * Kp * 2);
        } else {
            const __bf16* wp = wrow16 + ld_tap * p.ktap + ld_c;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j) rw[pl][j] = *reinterpret_cast<const bf16x8*>(wp + pl * wplane + (size_t)(64 * j) * Kp);
        }
        ld_c += GEMM_BK;
        if (ld_c >= p.ktap) { ld_c = 0; ++ld_tap; }
    };
    bool range_bad = false;      // terms 2: an A element beyond the fp16 range saturates in split_f16 (GemmArgs::range_flag)
    auto stage = [&](int buf) {
        if constexpr (TERMS == 0) {
            float* As = lds + buf * TILE_FLOATS;
            float* Bs = As + BM * LDS_STRIDE;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                f32x4 v = ra[i];
                if (A_NORM) v = (v - r_mean[i]) * r_rstd[i];
                if (A_MASK) v *= r_mask[i];
                if (!r_ok[i]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(As + (lrow + 32 * i) * LDS_STRIDE + lq) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * i) * LDS_STRIDE + lq) = rb[i];
        } else {
            __bf16* As = reinterpret_cast<__bf16*>(lds);
            __bf16* Bs = As + NPL * APLANE;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                f32x4 v = ra[i];
                if (A_NORM) v = (v - r_mean[i]) * r_rstd[i];
                if (A_MASK) v *= r_mask[i];
                if (!r_ok[i]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                __bf16* d = As + (lrow + 32 * i) * SPLIT_RS + lq;
                if constexpr (TERMS == 2) {
                    f16x4 h, l;
                    range_bad |= out_of_f16_range(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { _Float16 a, b; split_f16(v[e], a, b); h[e] = a; l[e] = b; }
                    *reinterpret_cast<f16x4*>(d) = h;
                    *reinterpret_cast<f16x4*>(d + APLANE) = l;
                } else {
                    bf16x4 h, m, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(v[e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
                    *reinterpret_cast<bf16x4*>(d) = h;
                    *reinterpret_cast<bf16x4*>(d + APLANE) = m;
                    if (NPL == 3) *reinterpret_cast<bf16x4*>(d + 2 * APLANE) = l;
                }
            }
            if constexpr (TERMS == 2) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    *reinterpret_cast<bf16x8*>(Bs + (wc8 >> 2) * BPLANE + (wr8 + 32 * jj) * SPLIT_RS + (wc8 & 3) * 8) = rw[jj >> 1][jj & 1];
            } else {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                    for (int j = 0; j < 2; ++j) *reinterpret_cast<bf16x8*>(Bs + pl * BPLANE + (wr16 + 64 * j) * SPLIT_RS + wc16) = rw[pl][j];
            }
        }
    };

    f32x16 accx[TERMS == 2 ? MI : 1][2];      // TERMS = 2: cross products (scaled by 2^11)
    if constexpr (TERMS == 2) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accx[i][j][r] = 0.f;
    }
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = Kp / GEMM_BK;
    // LayerNorm prologue (ntaps == 1: a thread's rows are the same for every k-step).  Statistics come either from arrays
    // or from the 64-column partial moments the producing GEMM left behind: the 8 threads tid&7 that stage one row each load
    // one partial (issued BEFORE the first tile fetch so they return first) and merge them with an equal-count Chan merge.
    const int pj = tid & 7;
    float2 pstat[AR];
    int prow[AR];
    if (A_NORM) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int tin = at[i];
            prow[i] = (unsigned)tin < (unsigned)p.T_in ? arow_base[i] + tin : 0;
            if (p.a_part) {
                const float* q = p.a_part + (size_t)prow[i] * p.a_nparts * 2;
                float2 acc2 = {0.f, 0.f};
                for (int k = pj; k < p.a_nparts; k += 8) {
                    const float2 v = *reinterpret_cast<const float2*>(q + 2 * k);
                    acc2.x += v.x;
                    acc2.y += v.y;
                }
                pstat[i] = acc2;
            } else {
                pstat[i] = float2{p.a_mean[prow[i]], p.a_rstd[prow[i]]};
            }
        }
    }
    fetch();
    if (A_NORM) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            if (p.a_part) {
                float m2 = pstat[i].y;
                const float mean = allreduce8(pstat[i].x) / (float)p.a_nparts;
                if (p.a_nparts <= 8) {
                    if (pj < p.a_nparts) { const float d = pstat[i].x - mean; m2 += 64.0f * (d * d); }
                } else {
                    const float* q = p.a_part + (size_t)prow[i] * p.a_nparts * 2;
                    for (int k = pj; k < p.a_nparts; k += 8) { const float d = q[2 * k] - mean; m2 += 64.0f * (d * d); }
                }
                r_mean[i] = mean;
                r_rstd[i] = 1.0f / sqrtf(allreduce8(m2) / (64.0f * (float)p.a_nparts) + p.a_eps);
            } else {
                r_mean[i] = pstat[i].x;
                r_rstd[i] = pstat[i].y;
            }
        }
    }

    if constexpr (TERMS == 0) {
        stage(0);
        __syncthreads();
        const int frag_off = (lane & 31) * LDS_STRIDE + 4 * (lane >> 5);
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) fetch();
            const float* Aw = lds + (kt & 1) * TILE_FLOATS + (wm * (BM / 2)) * LDS_STRIDE + frag_off;
            const float* Bw = lds + (kt & 1) * TILE_FLOATS + (BM + wn * 64) * LDS_STRIDE + frag_off;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a[MI], b[2];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(Aw + i * 32 * LDS_STRIDE + 8 * g);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bw + j * 32 * LDS_STRIDE + 8 * g);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
            }
            if (more) stage((kt + 1) & 1);
            __syncthreads();
        }
    } else {
        // split mode: one LDS stage; the other resident workgroup of the CU covers the two barriers per k-step.
        // Fragment of v_mfma_f32_32x32x16_bf16: lane (r = lane&31, h = lane>>5) holds k = 8h .. 8h+7 of a 16-wide k block.
        const __bf16* As = reinterpret_cast<const __bf16*>(lds);
        const int frag = (lane & 31) * SPLIT_RS + 8 * (lane >> 5);
        const __bf16* Aw = As + (wm * (BM / 2)) * SPLIT_RS + frag;
        const __bf16* Bw = As + NPL * APLANE + (wn * 64) * SPLIT_RS + frag;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt) __syncthreads();          // everyone finished reading the previous tile
            stage(0);
            __syncthreads();
            if (kt + 1 < nk) fetch();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 a[NPL][MI], b[NPL][2];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(Aw + pl * APLANE + i * 32 * SPLIT_RS + kb * 16);
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(Bw + pl * BPLANE + j * 32 * SPLIT_RS + kb * 16);
                }
                if constexpr (TERMS == 2) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f16x8 ah = __builtin_bit_cast(f16x8, a[0][i]), al = __builtin_bit_cast(f16x8, a[1][i]);
                            const f16x8 bh = __builtin_bit_cast(f16x8, b[0][j]), bl = __builtin_bit_cast(f16x8, b[1][j]);
                            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[i][j], 0, 0, 0);
                            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int sum = NPL - 1; sum >= 0; --sum)      // plane-index sum: smallest products first
#pragma unroll
                        for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                            for (int i = 0; i < MI; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[sum - pa][j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();                      // the epilogue tile overwrites the stage
    }

    // ---- epilogue.  Accumulator map (32x32 tile): column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Each wave parks its 64x64 tile in LDS (free after the last barrier) and re-reads it as float4 rows, so that the
    // epilogue math runs on 4 consecutive columns per lane and every store instruction writes whole 256-byte row pieces.
    constexpr int CS = GEMM_CS;   // row stride of the parked tile (floats)
    float* Cw = lds + wave * ((BM / 2) * CS);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cw[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + j * 32 + (lane & 31)] =
                    (TERMS == 2) ? acc[i][j][r] + accx[i][j][r] * (1.0f / F16_RES_SCALE) : acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the tile is private to this wave, no barrier needed
    __builtin_amdgcn_wave_barrier();
    if constexpr (TERMS == 2) raise_range_flag(p.range_flag, range_bad);

    const bool plain_rows = (p.out_stride == 1 && p.out_off == 0 && p.out_T == p.T_out);
    const int nc = n0 + wn * 64 + (lane & 15) * 4;          // first of this lane's 4 columns
    const bool vec = ((p.N & 3) == 0) && (!p.out || (p.ldc & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
    if (vec) {     // the common case: unrolled, residuals prefetched, activation chosen once (gemm_epilogue.h)
        gemm_epilogue_rows<BM, false>(p, Cw, nullptr, M, m0, n0, wm, wn, lane);
        return;
    }
    // odd widths (N or a leading dimension not a multiple of 4: the 1-channel duration head, 100-bin mel rows): scalar stores
    float bias4[4] = {0.f, 0.f, 0.f, 0.f}, s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (nc + e < p.N) {
            if (p.bias) bias4[e] = p.bias[nc + e];
            if (p.act == ACT_SNAKE) { s0[e] = p.p0[nc + e]; s1[e] = p.p1[nc + e]; }
        }
    }
    for (int it = 0; it < BM / 8; ++it) {
        const int rl = it * 4 + (lane >> 4);
        const int m = m0 + wm * (BM / 2) + rl;
        if (m >= M || nc >= p.N) continue;
        int orow = m;
        if (!plain_rows) {
            const int b = m / p.T_out;
            orow = b * p.out_T + (m - b * p.T_out) * p.out_stride + p.out_off;
        }
        const f32x4 a = *reinterpret_cast<const f32x4*>(Cw + rl * CS + (lane & 15) * 4);
        const float om = p.out_mask ? p.out_mask[orow] : 1.0f;
        float c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = act_apply(a[e] + bias4[e], p.act, s0[e], s1[e]);
            if (p.out_mask) v *= om;
            if (p.out_scale != 1.0f) v *= p.out_scale;
            c[e] = v;
        }
        float* op = p.out + (size_t)orow * p.ldc + nc;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (nc + e < p.N) op[e] = p.res ? c[e] + p.res[(size_t)orow * p.ldr + nc + e] : c[e];
    }
}

template <int BM, bool A_MASK, bool A_NORM, int TERMS>
static hipError_t launch_variant(const GemmArgs& a, hipStream_t s) {
    static bool configured = false;   // per instantiation
    auto kern = gemm_f32_kernel<BM, A_MASK, A_NORM, TERMS>;
    constexpr int lds_bytes = gemm_lds_bytes(BM, TERMS);
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int M = a.B * a.T_out;
    const int grid = ((M + BM - 1) / BM) * ((a.N + GEMM_BN - 1) / GEMM_BN);
    static const std::string tag = "gemm_f32_kernel<" + std::to_string(BM) + ", " + tf(A_MASK) + ", " + tf(A_NORM) + ", " + std::to_string(TERMS) + ">";
    g_kernel_tag = tag.c_str();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

template <int BM, int TERMS>
static hipError_t launch_bm(const GemmArgs& a, hipStream_t s) {
    const bool mk = a.a_mask != nullptr, nm = a.a_mean != nullptr || a.a_part != nullptr;
    if (mk && nm) return launch_variant<BM, true, true, TERMS>(a, s);
    if (mk) return launch_variant<BM, true, false, TERMS>(a, s);
    if (nm) return launch_variant<BM, false, true, TERMS>(a, s);
    return launch_variant<BM, false, false, TERMS>(a, s);
}

template <int BM>
static hipError_t launch_terms(const GemmArgs& a, hipStream_t s) {
    if (a.terms == 6) return launch_bm<BM, 6>(a, s);
    if (a.terms == 3) return launch_bm<BM, 3>(a, s);
    if (a.terms == 2) return launch_bm<BM, 2>(a, s);
    return launch_bm<BM, 0>(a, s);
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.a16_0) return launch_gemm_p16(a, s);     // operands already split in memory: gemm_p16.hip
    if (a.res16) return hipErrorInvalidValue;      // P16 residuals: gemm_p16.hip only
    // shape contract (the kernel indexes without further checks)
    if (!a.a0 || (!a.w && !a.w16) || (!a.out && !a.out16) || a.N <= 0 || a.B <= 0 || a.T_out <= 0 || a.T_in <= 0) return hipErrorInvalidValue;
    // a P16 copy of the result (the consumer is a P16 GEMM) needs the vectorised epilogue; out may then be null
    if (a.out16 && ((a.N % 32) || a.ld16 < 2 * a.N || (a.ld16 & 3) || (a.out && (a.ldc & 3)) || (a.res && (a.ldr & 3)))) return hipErrorInvalidValue;
    if (a.ntaps < 1 || a.ntaps > MAX_TAPS) return hipErrorInvalidValue;
    if (a.ktap % GEMM_BK != 0 || a.ktap < a.c0 + a.c1) return hipErrorInvalidValue;
    if ((a.c0 & 3) || (a.c1 & 3) || (a.lda0 & 3) || (a.lda1 & 3)) return hipErrorInvalidValue;
    if (a.a1 && (a.c0 % GEMM_BK)) return hipErrorInvalidValue;
    if (!a.a1 && a.c1) return hipErrorInvalidValue;
    if (a.lda0 < a.c0 || (a.a1 && a.lda1 < a.c1)) return hipErrorInvalidValue;
    if ((a.a_mean == nullptr) != (a.a_rstd == nullptr)) return hipErrorInvalidValue;
    if (a.a_part && (a.a_mean || a.a_nparts <= 0)) return hipErrorInvalidValue;
    if ((a.a_mean || a.a_part) && (a.ntaps != 1 || a.in_stride != 1 || a.tap_off[0] != 0)) return hipErrorInvalidValue;
    if (a.stats_out && ((a.N & 63) || (a.ldc & 3) || (a.res && (a.ldr & 3)))) return hipErrorInvalidValue;
    if (a.act == ACT_SNAKE && (!a.p0 || !a.p1)) return hipErrorInvalidValue;
    if (a.terms != 0 && a.terms != 2 && a.terms != 3 && a.terms != 6) return hipErrorInvalidValue;
    if (a.terms != 0 && !a.w16) return hipErrorInvalidValue;
    // Block-tile height: 256 CUs x 2 resident workgroups = 512 slots per round; pick the height whose grid wastes the
    // least of its last round (e.g. M=10240, N=1152: 720 tiles of 128 rows fill 70 % of two rounds, 1440 tiles of 64
    // rows fill 94 % of three).  The 64-row tile has half the A-fragment reuse, hence the small handicap.
    const int M = a.B * a.T_out;
    const int nt = (a.N + GEMM_BN - 1) / GEMM_BN;
    auto fill = [&](int bm) {
        const int tiles = ((M + bm - 1) / bm) * nt;
        const int rounds = (tiles + 511) / 512;
        return (double)tiles / (rounds * 512.0) * ((double)M / (((M + bm - 1) / bm) * bm));
    };
    static const int env_bm = [] { const char* e = getenv("MTTS_GEMM_BM"); return e ? atoi(e) : 0; }();   // A/B runs only
    const int force = a.force_bm ? a.force_bm : env_bm;
    if (force == 64 || (force == 0 && 0.97 * fill(64) > fill(128))) return launch_terms<64>(a, s);
    return launch_terms<128>(a, s);
}

// ------------------------------------------------------------------------------------------------ weight packing
void pack_weight_host(const float* w, int kind, int N, int C, int ntaps, int kT, const int* tsel, const float* col_scale,
                      float* dst, int ktap) {
    if (ktap <= 0) ktap = round_up(C, GEMM_BK);
    const int Kp = ntaps * ktap;
    const int Np = round_up(N, GEMM_BN);
    for (size_t i = 0; i < (size_t)Np * Kp; ++i) dst[i] = 0.f;
    for (int n = 0; n < N; ++n)
        for (int j = 0; j < ntaps; ++j)
            for (int c = 0; c < C; ++c) {
                float v;
                if (kind == 0) v = w[(size_t)n * C + c];
                else if (kind == 1) v = w[((size_t)n * C + c) * ntaps + j];
                else v = w[((size_t)c * N + n) * kT + tsel[j]];
                if (col_scale) v *= col_scale[c];
                dst[(size_t)n * Kp + j * ktap + c] = v;
            }
}

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);        // inf / nan unchanged (top bits)
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// fp32 panel [Np][Kp] -> three bf16 planes [3][Np][Kp] with w = h + m + l exactly (host side, same arithmetic as split3)
void split_panel_host(const float* panel, size_t n, uint16_t* planes) {
    for (size_t i = 0; i < n; ++i) {
        const float x = panel[i];
        const uint16_t h = f32_to_bf16_rne(x);
        const float r1 = x - bf16_to_f32(h);
        const uint16_t m = f32_to_bf16_rne(r1);
        const float r2 = r1 - bf16_to_f32(m);
        planes[i] = h;
        planes[n + i] = m;
        planes[2 * n + i] = f32_to_bf16_rne(r2);
    }
}

// TERMS = 2 image: per row and 32-wide k group, 32 fp16 heads then the 32 scaled fp16 residuals (same arithmetic as
// split_f16): a k-step's operand row is one whole 128-B line (half-line requests cost ~10 % in the lab, gemm_lab_planes.hip)
__global__ void split_panel_f16_kernel(const float* __restrict__ panel, size_t n, _Float16* __restrict__ planes) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        _Float16 h, l;
        split_f16(panel[i], h, l);
        const size_t o = (i >> 5) * 64 + (i & 31);      // Kp % 32 == 0: a group never straddles rows
        planes[o] = h;
        planes[o + 32] = l;
    }
}
hipError_t launch_split_panel_f16(const float* panel, size_t n, void* planes, hipStream_t s) {
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(split_panel_f16_kernel, dim3(grid), dim3(256), 0, s, panel, n, static_cast<_Float16*>(planes));
    return hipGetLastError();
}
void panel_h16_host(const float* panel, size_t n, uint16_t* plane) {
    for (size_t i = 0; i < n; ++i) {
        const float x = panel[i];
        const _Float16 h = (_Float16)(x < -65504.f ? -65504.f : (x > 65504.f ? 65504.f : x));
        memcpy(&plane[i], &h, 2);
    }
}
void panel_bf16_host(const float* panel, size_t n, uint16_t* plane) {
    for (size_t i = 0; i < n; ++i) {
        const __bf16 h = (__bf16)panel[i];
        memcpy(&plane[i], &h, 2);
    }
}
void split_panel_f16_host(const float* panel, size_t n, uint16_t* planes) {
    for (size_t i = 0; i < n; ++i) {
        const float x = panel[i];
        const float xc = x < -65504.f ? -65504.f : (x > 65504.f ? 65504.f : x);
        const _Float16 h = (_Float16)xc;
        float r = (x - (float)h) * F16_RES_SCALE;
        r = r < -65504.f ? -65504.f : (r > 65504.f ? 65504.f : r);
        const _Float16 l = (_Float16)r;
        const size_t o = (i >> 5) * 64 + (i & 31);
        memcpy(&planes[o], &h, 2);
        memcpy(&planes[o + 32], &l, 2);
    }
}

__global__ void split_panel_kernel(const float* __restrict__ panel, size_t n, __bf16* __restrict__ planes) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        __bf16 h, m, l;
        split3(panel[i], h, m, l);
        planes[i] = h;
        planes[n + i] = m;
        planes[2 * n + i] = l;
    }
}
hipError_t launch_split_panel(const float* panel, size_t n, void* planes, hipStream_t s) {
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(split_panel_kernel, dim3(grid), dim3(256), 0, s, panel, n, static_cast<__bf16*>(planes));
    return hipGetLastError();
}

__global__ void pack_weight_kernel(const float* __restrict__ w, int N, int C, int ntaps, int ktap, int Np, float* __restrict__ dst) {
    const int Kp = ntaps * ktap;
    const size_t total = (size_t)Np * Kp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kp);
        const int k = (int)(i - (size_t)n * Kp);
        const int j = k / ktap, c = k - j * ktap;
        float v = 0.f;
        if (n < N && c < C) v = w[((size_t)n * C + c) * ntaps + j];
        dst[i] = v;
    }
}

hipError_t launch_pack_weight(const float* w, int N, int C, int ntaps, float* dst, hipStream_t s) {
    const int ktap = round_up(C, GEMM_BK), Np = round_up(N, GEMM_BN);
    const size_t total = (size_t)Np * ntaps * ktap;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, s, w, N, C, ntaps, ktap, Np, dst);
    return hipGetLastError();
}

}  // namespace mtts
