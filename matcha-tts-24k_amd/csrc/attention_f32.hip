// fp32 flash-style self-attention for gfx950 (CDNA4): one pass over the keys, online softmax, no T x T buffer.
//
// Replaces, on packed q|k|v rows [B*T, 3*H*D]:
//   decoder: diffusers Attention via BasicTransformerBlock (reference transformer.py:249-261; decoder.py:379-385):
//            softmax(q k^T / sqrt(D) + bias) v with an ADDITIVE float key bias (mask value 1.0 valid / 0.0 padded)
//   encoder: F.scaled_dot_product_attention with a boolean query*key mask (reference text_encoder.py:228-235,306)
//
// Work split: one workgroup per (batch, head, block of 128 or 64 queries), each of its 4 or 2 waves 32 of them.  Keys/values stream through LDS in tiles of 64 (global -> registers prefetch -> LDS).
// Both products run in the "transposed" orientation so that the query sits on the lane:
//   S^T[key][q] = K[key][:] . Q[q][:]     A = K fragment (LDS, ds_read_b128), B = Q fragment (registers, loaded once)
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]   A = V column (LDS, transposed by ds_read_b64_tr_b16), B = the S^T accumulator itself
// Both run on the f16 matrix pipe with split operands (x = h + l, h = fp16(x), l = fp16(x - h); products h.h + h.l + l.h,
// fp32 accumulate: 3 v_mfma_f32_32x32x16_f16 per 16-deep block instead of 8 v_mfma_f32_32x32x2_f32).  q, k, v are O(1)
// and p <= 1, so the residuals stay representable (absolute error ~3e-8 per operand); the softmax scale multiplies the
// fp32 scores.  The S^T accumulator feeds P.V without leaving registers: registers 8s..8s+7 of a 32-key sub-tile are the
// B fragment of key block s in the permuted key order 16s + 8(j>>2) + 4h + (j&3); V is staged row-major and the matching A
// fragment is two transposing LDS reads (ds_read_b64_tr_b16: 4 keys x 16 d per 16-lane group).
// so the softmax row (max, sum) is a reduction over the lane's own registers plus one cross-half shuffle, the
// probabilities never leave registers, and the per-query rescale is a lane-uniform multiply of the O^T accumulators.
// Head dims below 64 (encoder: 48) are zero-padded to 64 in the staged tiles.
#include "kernels.h"
#include "device_utils.h"
#include <cstdlib>
#include <string>

namespace mtts {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int AT_QW = 32;     // queries per wave (one 32-column MFMA tile)
constexpr int AT_K = 64;      // keys per tile
constexpr int AT_D = 64;      // padded head dim
constexpr int AT_KS = 72;     // LDS row stride of a K plane (halves): 144 B = 9 x 16 B, conflict-free 16-byte fragment reads
constexpr float NEG_BIG = -1e30f;
constexpr float LOG2E = 1.44269504088896340736f;
// exp(x) = 2^(x log2 e) as one v_exp_f32 (~1 ulp); the log2(e) factor is folded into the score scale and the key bias, so
// the softmax weights carry ~1e-6 relative error at |x| ~ 10 (far inside the path's 1e-3 budget) for 2 VALU ops per score
// (subtract the running maximum, exponentiate) instead of ~20 for expf.

// NW waves per workgroup = NW*32 queries.  NW = 4 for long sequences; NW = 2 when T is short enough that 128-query
// blocks would leave the last block mostly empty or the grid under one round (e.g. T = 320: 3 blocks of 128 waste 17 %).
// P16: q|k|v rows arrive as a P16 image with UNSCALED residuals (written by the q|k|v projection's epilogue, gemm_p16.hip)
// and the output leaves as a P16 image with the 2^11-scaled residual for the out-projection: no split arithmetic on the
// way in, one per output element on the way out.  Head dim 64 only (a head = 256 contiguous bytes of the row).
// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of halves leaves LDS column-major (EXEC must be full)
__device__ __forceinline__ f16x4 tr_read4(const _Float16* lds_ptr) {
    typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    const h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(lds_ptr));
    return __builtin_bit_cast(f16x4, v);
}

// ONE (with P16): the opt-in fp16 mode -- heads x heads products only (see gemm_p16.hip).
// HALF (with P16 and ONE): q|k|v and the output are H16 images (AttnArgs::half16) -- a head is 64 contiguous halves.
// BF (with HALF): the H16 images hold bfloat16 (AttnArgs::bf16): bf16 MFMAs, probabilities and outputs rounded to bfloat16.
using bf16x8_a = __attribute__((ext_vector_type(8))) __bf16;
template <bool BF>
__device__ __forceinline__ f32x16 att_mfma(f16x8 a, f16x8 b, f32x16 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_a, a), __builtin_bit_cast(bf16x8_a, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// KR: key rows held in LDS.  AT_K (the default): one 64-key tile at a time, the next one prefetched into registers, two
// barriers per tile.  KR = 192 ("whole"): short sequences (the half-length level of the estimator: 161 keys) -- one workgroup
// per (utterance, head) with a wave per 32 queries stages ALL keys and values once, in one batch of loads and behind one
// barrier, and the tile loop then runs without fetches or barriers.  At 161 keys the tiled form is mostly prologue and exposed
// round trips (16 us per launch for 1.3 GFLOP: round-2 verdict, "attention_f32_kernel<2, ...> MFMA busy 0.09"), and its three
// query blocks per head each re-stage the same keys.
template <int NW, bool P16, bool ONE = false, bool HALF = false, bool BF = false, int KR = AT_K>
__global__ __launch_bounds__(64 * NW, KR > AT_K ? 1 : 2) void attention_f32_kernel(const AttnArgs p) {
    static_assert(!HALF || (P16 && ONE), "H16 I/O runs the single-product loop");
    static_assert(!BF || HALF, "bfloat16 planes exist in the 16-bit storage mode only");
    constexpr bool WHOLE = KR > AT_K;
    constexpr int NT = 64 * NW;                   // threads
    constexpr int SROWS = NT / 4;                 // key rows staged per pass (4 threads x float4 x 4 = one 64-float row)
    constexpr int SP = KR / SROWS;                // passes over the rows held
    static_assert(SP >= 1 && SP * SROWS == KR && (!WHOLE || NW * AT_QW >= KR), "staging passes cover the rows held");
    __shared__ __attribute__((aligned(16))) _Float16 Ks[2 * KR * AT_KS];   // planes h | l
    // V planes h | l, row-major and split by d half: [plane][d >> 5][key][d & 31] (64-byte rows, no padding).  P.V needs V^T
    // fragments; ds_read_b64_tr_b16 transposes 4 keys x 16 d blocks on the way out of LDS, so staging is plain row stores.
    __shared__ __attribute__((aligned(16))) _Float16 Vs[2 * 2 * KR * 32];
    __shared__ __attribute__((aligned(16))) float Bs[KR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, h = lane >> 5;
    // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of
    // (batch, head, q-block) so the q-blocks that re-read one head's K/V hit the same L2 (measured before this remap:
    // 331 MB fetched per launch at B=32, T=640 against 94 MB of q|k|v).
    const int qblocks = (p.T + NW * AT_QW - 1) / (NW * AT_QW);
    int swz;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int qb = swz % qblocks, head = (swz / qblocks) % p.H, b = swz / (qblocks * p.H);
    const int q0 = qb * (NW * AT_QW) + wave * AT_QW;
    const int ld = 3 * p.H * p.D;
    const size_t rowbase = (size_t)b * p.T;
    const int Tb = p.klen ? min(p.T, p.klen[b]) : p.T;      // keys of this utterance: [0, Tb) (per-request / folded padding)
    const float* qptr = p.qkv + head * p.D;
    const float* kptr = p.qkv + p.H * p.D + head * p.D;
    const float* vptr = p.qkv + 2 * p.H * p.D + head * p.D;
    const float ninf = -__builtin_huge_valf();

    // ---- Q fragments for v_mfma_f32_32x32x16_f16 (B operand): lane (q = lq, half h) holds Q[q][16kb + 8h + j], j < 8,
    // kb < 4, as two fp16 terms (unscaled; the softmax scale multiplies the fp32 scores)
    const int qi = q0 + lq;
    const bool q_in = qi < p.T;            // masked QUERY rows (boolean mode) are computed like any other: the caller multiplies
                                           // them by the mask afterwards (text_encoder.py:236-237), so their values are don't-care
    const float scale2 = p.scale * LOG2E;  // softmax in the log2 domain: exp(x) = exp2(x * log2 e), folded into the score FMA
    f16x8 qh[4], ql[4];
    const _Float16* q16 = p.qkv16 + head * ((HALF ? 1 : 2) * AT_D);          // head h = groups 2h, 2h+1 of the q section
    const _Float16* k16 = q16 + (HALF ? 1 : 2) * p.H * AT_D;
    const _Float16* v16 = q16 + (HALF ? 2 : 4) * p.H * AT_D;
    if constexpr (P16) {
        const _Float16* qrow = q16 + (rowbase + (q_in ? qi : 0)) * (size_t)p.ld16;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            const _Float16* src = HALF ? qrow + 16 * kb + 8 * h : qrow + (kb >> 1) * 64 + (kb & 1) * 16 + 8 * h;
            qh[kb] = q_in ? *reinterpret_cast<const f16x8*>(src) : z;
            ql[kb] = (q_in && !HALF) ? *reinterpret_cast<const f16x8*>(src + 32) : z;
        }
    } else
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int half4 = 0; half4 < 2; ++half4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int d = 16 * kb + 8 * h + 4 * half4;
            if (q_in && d < p.D) v = *reinterpret_cast<const f32x4*>(qptr + (rowbase + qi) * ld + d);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 a = (_Float16)v[e];
                qh[kb][4 * half4 + e] = a;
                ql[kb][4 * half4 + e] = (_Float16)(v[e] - (float)a);
            }
        }
    }

    f32x16 o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    // ---- staging: thread -> key row (tid>>2) + SROWS*pass, float4 columns (tid&3)*4 + 16c: 4 lanes read 64 contiguous bytes
    const int srow = tid >> 2;
    const int sd = (tid & 3) * 4;
    f32x4 rk[SP][4], rv[SP][4];       // P16: the same 16 bytes hold 8 halves (chunk (tid&3) + 4c of the head's 256-B slice)
    float rbias[SP];
    bool r_in[SP];
    // all loads unconditional (clamped addresses), zeroing deferred to the LDS write: nothing waits inside the fetch
    auto fetch = [&](int k0) {
#pragma unroll
        for (int sp = 0; sp < SP; ++sp) {
            const int key = k0 + srow + SROWS * sp;
            r_in[sp] = key < Tb;
            const size_t row = rowbase + (r_in[sp] ? key : 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (HALF) {          // 8 chunks of 8 halves per head: this thread's are (tid&3) and (tid&3) + 4
                    if (c < 2) {
                        rk[sp][c] = *reinterpret_cast<const f32x4*>(k16 + row * p.ld16 + ((tid & 3) + 4 * c) * 8);
                        rv[sp][c] = *reinterpret_cast<const f32x4*>(v16 + row * p.ld16 + ((tid & 3) + 4 * c) * 8);
                    }
                } else if constexpr (P16) {
                    rk[sp][c] = *reinterpret_cast<const f32x4*>(k16 + row * p.ld16 + ((tid & 3) + 4 * c) * 8);
                    rv[sp][c] = *reinterpret_cast<const f32x4*>(v16 + row * p.ld16 + ((tid & 3) + 4 * c) * 8);
                } else {
                    const int d = sd + 16 * c;
                    const int dd = d < p.D ? d : 0;
                    rk[sp][c] = *reinterpret_cast<const f32x4*>(kptr + row * ld + dd);
                    rv[sp][c] = *reinterpret_cast<const f32x4*>(vptr + row * ld + dd);
                }
            }
            rbias[sp] = p.mask ? p.mask[row] : 1.0f;
        }
    };
    auto stage = [&]() {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp = 0; sp < SP; ++sp) {
            const int r = srow + SROWS * sp;
            if constexpr (HALF) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {      // chunk (tid&3) + 4c = dims 8 (tid&3) + 32 c ..+7 of the one plane
                    const int d = 32 * c + 8 * (tid & 3);
                    *reinterpret_cast<f32x4*>(Ks + r * AT_KS + d) = r_in[sp] ? rk[sp][c] : zero;
                    *reinterpret_cast<f32x4*>(Vs + (c * KR + r) * 32 + 8 * (tid & 3)) = r_in[sp] ? rv[sp][c] : zero;
                }
            } else if constexpr (P16) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {      // chunk (tid&3) + 4c: plane c&1 (head / residual), dims 32 (c>>1) + 8 (tid&3) ..+7
                    const int plane = c & 1, d = 32 * (c >> 1) + 8 * (tid & 3);
                    const f32x4 kraw = r_in[sp] ? rk[sp][c] : zero;
                    *reinterpret_cast<f32x4*>(Ks + plane * KR * AT_KS + r * AT_KS + d) = kraw;
                    const f32x4 vraw = r_in[sp] ? rv[sp][c] : zero;
                    *reinterpret_cast<f32x4*>(Vs + ((plane * 2 + (c >> 1)) * KR + r) * 32 + 8 * (tid & 3)) = vraw;
                }
            } else
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = r_in[sp] && (sd + 16 * c) < p.D;
                const f32x4 kv = ok ? rk[sp][c] : zero;
                f16x4 kh, kl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 a = (_Float16)kv[e];
                    kh[e] = a;
                    kl[e] = (_Float16)(kv[e] - (float)a);
                }
                *reinterpret_cast<f16x4*>(Ks + r * AT_KS + sd + 16 * c) = kh;
                *reinterpret_cast<f16x4*>(Ks + KR * AT_KS + r * AT_KS + sd + 16 * c) = kl;
                const f32x4 vv = ok ? rv[sp][c] : zero;
                f16x4 vh4, vl4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 a = (_Float16)vv[e];
                    vh4[e] = a;
                    vl4[e] = (_Float16)(vv[e] - (float)a);
                }
                const int dcol = sd + 16 * c;              // 4 consecutive d of key r
                *reinterpret_cast<f16x4*>(Vs + ((dcol >> 5) * KR + r) * 32 + (dcol & 31)) = vh4;
                *reinterpret_cast<f16x4*>(Vs + ((2 + (dcol >> 5)) * KR + r) * 32 + (dcol & 31)) = vl4;
            }
            if ((tid & 3) == 0) {
                float bv = ninf;
                if (r_in[sp]) bv = (p.mask_mode == 0) ? rbias[sp] * LOG2E : (rbias[sp] != 0.f ? 0.f : ninf);
                Bs[r] = bv;                 // key bias in the log2 domain
            }
        }
    };

    const int ntiles = (Tb + AT_K - 1) / AT_K;
    fetch(0);
    if constexpr (WHOLE) {                // every key row of the utterance, once
        stage();
        __syncthreads();
    }
    for (int kt = 0; kt < ntiles; ++kt) {
        const int kbase = WHOLE ? kt * AT_K : 0;      // first row of this tile in the LDS arrays
        if constexpr (!WHOLE) {
            if (kt) __syncthreads();      // everyone finished reading the previous tile
            stage();
            __syncthreads();
            if (kt + 1 < ntiles) fetch((kt + 1) * AT_K);
        }

        // ---- S^T = K . Q^T for 2 sub-tiles of 32 keys (A = K fragment: lane (key, h) holds K[key][16kb + 8h + j]).
        // (Skipping the second sub-tile of a last tile with <= 32 keys -- T = 322 leaves two -- measured SLOWER, 3.20 vs 3.02 ms of
        // attention per step: the uniform branches break the MFMA / LDS interleaving of the unrolled body.)
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
            const _Float16* kp = Ks + (kbase + 32 * t + lq) * AT_KS + 8 * h;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(kp + 16 * kb);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(kp + KR * AT_KS + 16 * kb);
                if constexpr (!ONE) {
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[kb], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[kb], s[t], 0, 0, 0);
                }
                s[t] = att_mfma<BF>(kh, qh[kb], s[t]);
            }
        }
        // ---- bias + online softmax.  Register r of sub-tile t is key 32t + (r&3) + 8(r>>2) + 4h.
        float mx = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(Bs + kbase + 32 * t + 8 * g4 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(s[t][4 * g4 + e], scale2, bb[e]);     // log2 domain: one FMA per score
                    s[t][4 * g4 + e] = v;
                    mx = fmaxf(mx, v);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[t][r] - m_new);
                s[t][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
        if (__any(m_new != m_run)) {       // the running maximum moved for some query of this wave: rescale (else alpha == 1)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        m_run = m_new;
        // ---- O^T += V^T . P^T on split f16 operands
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ph, pl;
                if constexpr (BF) {
                    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
                    u32x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = pack_bf16(s[t][8 * ks + 2 * j], s[t][8 * ks + 2 * j + 1]);
                    ph = __builtin_bit_cast(f16x8, w);
                    pl = ph;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pv = s[t][8 * ks + j];
                        const _Float16 a = (_Float16)pv;
                        ph[j] = a;
                        pl[j] = (_Float16)(pv - (float)a);
                    }
                }
                const int k0 = 32 * t + 16 * ks + 4 * h;       // this lane half's keys: k0..k0+3 and k0+8..k0+11
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    // transposed reads: the 16-lane group g = lane>>4 takes the block keys k0..k0+3 x d 32 dt + 16 (g&1) ..+15;
                    // lane 4q+p of the group addresses key k0+q, columns 4p..4p+3, and receives column lane&15 of the 4 keys
                    const _Float16* vp = Vs + ((dt * KR) + kbase + k0 + ((lane & 15) >> 2)) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                    f16x8 vh, vl;
                    const f16x4 h0 = tr_read4(vp), h1 = tr_read4(vp + 8 * 32);
                    const f16x4 l0 = tr_read4(vp + 2 * KR * 32), l1 = tr_read4(vp + 2 * KR * 32 + 8 * 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vh[e] = h0[e]; vh[4 + e] = h1[e]; vl[e] = l0[e]; vl[4 + e] = l1[e]; }
                    if constexpr (!ONE) {
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[dt], 0, 0, 0);
                    }
                    o[dt] = att_mfma<BF>(vh, ph, o[dt]);
                }
            }
    }

    // ---- normalise and store: lane holds query qi; register r of d-tile t is d = 32t + (r&3) + 8(r>>2) + 4h
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    bool range_bad = false;
    if (P16) {
        if (q_in) {
            _Float16* op = p.out16 + (rowbase + qi) * (size_t)p.ldo16 + head * ((HALF ? 1 : 2) * AT_D);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f16x4 hh, ll;
                    if constexpr (BF) {
                        using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
                        *reinterpret_cast<u32x2*>(op + t * 32 + 8 * g4 + 4 * h) =
                            u32x2{pack_bf16(o[t][4 * g4] * inv, o[t][4 * g4 + 1] * inv), pack_bf16(o[t][4 * g4 + 2] * inv, o[t][4 * g4 + 3] * inv)};
                        continue;
                    }
                    range_bad |= out_of_f16_range(o[t][4 * g4] * inv, o[t][4 * g4 + 1] * inv, o[t][4 * g4 + 2] * inv, o[t][4 * g4 + 3] * inv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 a, b;
                        split_f16(o[t][4 * g4 + e] * inv, p.out_lscale, a, b);
                        hh[e] = a;
                        ll[e] = b;
                    }
                    if constexpr (HALF) {
                        *reinterpret_cast<f16x4*>(op + t * 32 + 8 * g4 + 4 * h) = hh;
                    } else {
                        *reinterpret_cast<f16x4*>(op + t * 64 + 8 * g4 + 4 * h) = hh;
                        *reinterpret_cast<f16x4*>(op + t * 64 + 32 + 8 * g4 + 4 * h) = ll;
                    }
                }
        }
        raise_range_flag(p.range_flag, range_bad);
    } else if (q_in) {
        float* op = p.out + (rowbase + qi) * (size_t)(p.H * p.D) + head * p.D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * t + 8 * g4 + 4 * h;
                if (d < p.D) {
                    f32x4 v = {o[t][4 * g4] * inv, o[t][4 * g4 + 1] * inv, o[t][4 * g4 + 2] * inv, o[t][4 * g4 + 3] * inv};
                    *reinterpret_cast<f32x4*>(op + d) = v;
                }
            }
    }
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    const bool p16 = a.qkv16 != nullptr;
    const int ew = a.half16 ? 1 : 2;
    if ((a.half16 && !p16) || (a.bf16 && !a.half16)) return hipErrorInvalidValue;
    if (p16 ? (!a.out16 || a.D != AT_D || a.ld16 < 3 * ew * a.H * AT_D || (a.ld16 & 7) || a.ldo16 < ew * a.H * AT_D || (a.ldo16 & 3))
            : (!a.qkv || !a.out))
        return hipErrorInvalidValue;
    if (a.B <= 0 || a.T <= 0 || a.H <= 0) return hipErrorInvalidValue;
    if (a.D <= 0 || a.D > AT_D || (a.D & 3)) return hipErrorInvalidValue;
    if (a.mask_mode == 1 && !a.mask) return hipErrorInvalidValue;
    // 128-query blocks when they fill at least ~1.5 rounds of the chip without much tail waste, else 64-query blocks
    const int b128 = (a.T + 127) / 128, b64 = (a.T + 63) / 64;
    const long blocks128 = (long)b128 * a.H * a.B;
    const double waste128 = 1.0 - (double)a.T / (b128 * 128.0);
    static const int env_nw = [] { const char* e = getenv("MTTS_ATTN_NW"); return e ? atoi(e) : 0; }();     // A/B runs only
    // (or when 64-query blocks would pad to the same row count anyway -- T = 322: 3 x 128 = 6 x 64 -- and the grid still has
    // two workgroups per CU: half as many workgroups re-stage each head's keys and values)
    const bool same_rows = b128 * 128 == b64 * 64 && blocks128 >= 512;
    const bool use128 = env_nw == 4 || (env_nw == 0 && ((blocks128 >= 768 && waste128 < 0.1) || same_rows));
    {
        static thread_local std::string tag;
        const bool one = a.half16 || (p16 && a.fast16);
        tag = std::string("attention_f32_kernel<") + (use128 ? "4" : "2") + ", " + tf(p16) + ", " + tf(one) + ", " + tf(a.half16) + ", " + tf(a.half16 && a.bf16) + ", 64>";
        g_kernel_tag = tag.c_str();
    }
    // short sequences (65..192 keys: the estimator's half-length level) on P16 / H16 images: one workgroup per (utterance, head),
    // all keys staged once (KR = 192); MTTS_ATTN_WHOLE=0 keeps the tiled kernel (A/B runs)
    static const bool whole_on = [] { const char* e = getenv("MTTS_ATTN_WHOLE"); return !(e && e[0] == '0'); }();
    if (p16 && whole_on && env_nw == 0 && a.T > 64 && a.T <= 192) {
        static thread_local std::string wtag;
        const bool one = a.half16 || a.fast16;
        wtag = std::string("attention_f32_kernel<6, true, ") + tf(one) + ", " + tf(a.half16) + ", " + tf(a.half16 && a.bf16) + ", 192>";
        g_kernel_tag = wtag.c_str();
        const dim3 grid(a.H * a.B), block(384);
        if (a.half16 && a.bf16) hipLaunchKernelGGL((attention_f32_kernel<6, true, true, true, true, 192>), grid, block, 0, s, a);
        else if (a.half16) hipLaunchKernelGGL((attention_f32_kernel<6, true, true, true, false, 192>), grid, block, 0, s, a);
        else if (a.fast16) hipLaunchKernelGGL((attention_f32_kernel<6, true, true, false, false, 192>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attention_f32_kernel<6, true, false, false, false, 192>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (use128) {
        if (a.half16 && a.bf16) hipLaunchKernelGGL((attention_f32_kernel<4, true, true, true, true>), dim3(b128 * a.H * a.B), dim3(256), 0, s, a);
        else if (a.half16) hipLaunchKernelGGL((attention_f32_kernel<4, true, true, true>), dim3(b128 * a.H * a.B), dim3(256), 0, s, a);
        else if (p16 && a.fast16) hipLaunchKernelGGL((attention_f32_kernel<4, true, true>), dim3(b128 * a.H * a.B), dim3(256), 0, s, a);
        else if (p16) hipLaunchKernelGGL((attention_f32_kernel<4, true>), dim3(b128 * a.H * a.B), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attention_f32_kernel<4, false>), dim3(b128 * a.H * a.B), dim3(256), 0, s, a);
    } else {
        if (a.half16 && a.bf16) hipLaunchKernelGGL((attention_f32_kernel<2, true, true, true, true>), dim3(b64 * a.H * a.B), dim3(128), 0, s, a);
        else if (a.half16) hipLaunchKernelGGL((attention_f32_kernel<2, true, true, true>), dim3(b64 * a.H * a.B), dim3(128), 0, s, a);
        else if (p16 && a.fast16) hipLaunchKernelGGL((attention_f32_kernel<2, true, true>), dim3(b64 * a.H * a.B), dim3(128), 0, s, a);
        else if (p16) hipLaunchKernelGGL((attention_f32_kernel<2, true>), dim3(b64 * a.H * a.B), dim3(128), 0, s, a);
        else hipLaunchKernelGGL((attention_f32_kernel<2, false>), dim3(b64 * a.H * a.B), dim3(128), 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace mtts
