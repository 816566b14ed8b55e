// Shared GEMM epilogue (gemm_f32.hip, gemm_p16.hip): the wave's BM/2 x 64 tile is parked in LDS (row stride GEMM_CS
// floats) and re-read as rows of float4, 16 lanes per row, so every store instruction writes whole 256-byte row pieces.
//
//   c = act(LN'(acc) + bias);  c *= out_mask[row];  c *= out_scale;  c += res[row][n]   (res: fp32 rows or a P16 image)
//   -> fp32 rows (out) and/or a P16 image (out16, optionally times out16_mask[row]); optional 64-column partial moments
//      (stats_out)
//
// Built for memory-level parallelism: the tile's rows go in chunks of 16 with no early exits, a chunk's residual rows and
// mask values are all requested BEFORE the first one is consumed, and the activation / residual choice is made once per
// kernel (wave-uniform dispatch into compile-time variants) instead of per element.  The previous form (a rolled loop: LDS read -> residual load -> wait -> store, an activation switch per
// element) cost 20-50 us per launch on the decoder's shapes (tools/p16_ablate.py).
#pragma once
#include "kernels.h"
#include "device_utils.h"

#ifndef MTTS_STAMP
#define MTTS_STAMP(i) do { } while (0)
#endif

namespace mtts {

constexpr int GEMM_CS = 68;   // row stride of the parked tile (floats)

template <int V> struct IntC { static constexpr int value = V; };

__device__ __forceinline__ float allreduce64(float v) {     // every lane gets the wave's total
    v = allreduce16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// LN: LayerNorm-in-the-epilogue (P16 kernel): srow = [BM means | BM rstds] in LDS, p.wsum = panel row sums.
// GN: the GroupNorm-statistics part is compiled in (conv GEMMs feeding a Block1D); every other instantiation stays lean.
template <int BM, bool LN, bool GN = false>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmArgs& p, const float* __restrict__ Cw, const float* __restrict__ srow,
                                                   int M, int m0, int n0, int wm, int wn, int lane, const float* __restrict__ gstat = nullptr) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
    constexpr int NIT = BM / 8;                              // rows per lane: 4 rows per pass x NIT passes
    const bool plain_rows = (p.out_stride == 1 && p.out_off == 0 && p.out_T == p.T_out);
    const int nc = n0 + wn * 64 + (lane & 15) * 4;          // first of this lane's 4 columns
    const bool col_ok = nc < p.N;                            // N % 4 == 0: a lane's 4 columns are in or out together
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, s0 = bias4, s1 = bias4, ws4 = bias4;
    if (col_ok) {
        if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nc);
        if (p.act == ACT_SNAKE) { s0 = *reinterpret_cast<const f32x4*>(p.p0 + nc); s1 = *reinterpret_cast<const f32x4*>(p.p1 + nc); }
        if (LN) ws4 = *reinterpret_cast<const f32x4*>(p.wsum + nc);
    }
    // ---- GroupNorm statistics of the output (conv feeding a Block1D): two passes over the parked tile -- the group means of
    // this wave tile first, then squared deviations inside the main loop -- so no second trip over the tensor is needed.
    // Granule: the wave tile (BM/2 rows x 64 columns), split where it crosses into the next utterance (T_out >= BM/2, so at most
    // two parts): per part and group slice (<= 2) an entry (n, mean, M2).
    const bool gn = GN && p.gn_stats != nullptr;
    int gn_gi = 0, gn_cols0 = 64, gn_bnd = BM, gn_cnt0 = 0, gn_cnt1 = 0;   // part 0 = rows [0, cnt0), part 1 = rows [bnd, bnd + cnt1)
    float gn_mean[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gn_mu0 = 0.f, gn_mu1 = 0.f, gn_q0 = 0.f, gn_q1 = 0.f;
    if constexpr (GN) if (gn) {
        constexpr int rows_w = BM / 2;
        const int row_w0 = m0 + wm * rows_w;
        const int cpg = p.N / p.gn_groups, n0w = n0 + wn * 64, g0 = n0w / cpg;
        gn_cols0 = min(64, (g0 + 1) * cpg - n0w);
        gn_gi = (nc / cpg) - g0;                                          // 0 or 1: this lane's group slice (cpg >= 32)
        if (row_w0 < M) {
            const int b0 = row_w0 / p.T_out, t_w0 = row_w0 - b0 * p.T_out;
            const int nr0 = p.gn_nrows ? min(p.T_out, p.gn_nrows[b0]) : p.T_out;
            const int nr1 = (b0 + 1 < p.B) ? (p.gn_nrows ? min(p.T_out, p.gn_nrows[b0 + 1]) : p.T_out) : 0;
            gn_bnd = min(rows_w, p.T_out - t_w0);
            gn_cnt0 = max(0, min(gn_bnd, nr0 - t_w0));
            gn_cnt1 = max(0, min(rows_w - gn_bnd, nr1));
        }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rl = it * 4 + (lane >> 4);
            const bool in0 = rl < gn_cnt0, in1 = rl >= gn_bnd && rl < gn_bnd + gn_cnt1;
            if (in0 || in1) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Cw + rl * GEMM_CS + (lane & 15) * 4) + bias4;
                const float t = (a[0] + a[1]) + (a[2] + a[3]);
                if (in0) s0 += t; else s1 += t;
            }
        }
        const float c0f = (float)gn_cols0, c1f = (float)(64 - gn_cols0);
        const float t00 = allreduce64(gn_gi == 0 ? s0 : 0.f), t01 = allreduce64(gn_gi == 1 ? s0 : 0.f);
        gn_mean[0][0] = gn_cnt0 > 0 ? t00 / ((float)gn_cnt0 * c0f) : 0.f;
        gn_mean[0][1] = (gn_cnt0 > 0 && gn_cols0 < 64) ? t01 / ((float)gn_cnt0 * c1f) : 0.f;
        if (gn_cnt1 > 0) {                                                // wave-uniform
            const float t10 = allreduce64(gn_gi == 0 ? s1 : 0.f), t11 = allreduce64(gn_gi == 1 ? s1 : 0.f);
            gn_mean[1][0] = t10 / ((float)gn_cnt1 * c0f);
            gn_mean[1][1] = gn_cols0 < 64 ? t11 / ((float)gn_cnt1 * c1f) : 0.f;
        }
        gn_mu0 = gn_gi == 0 ? gn_mean[0][0] : gn_mean[0][1];
        gn_mu1 = gn_gi == 0 ? gn_mean[1][0] : gn_mean[1][1];
    }
    bool range_bad = false;
    // Block1D tail: this lane's group statistics (merged in the prologue into gstat = [mean x 4 | rstd x 4]) and affine
    // (a workgroup's rows may straddle two utterances: gstat = [utterance u: mean x 4 | rstd x 4] for u = 0, 1; rows from
    // gnr_bnd on belong to the second)
    float gnr_mu = 0.f, gnr_rs = 1.f, gnr_mu1 = 0.f, gnr_rs1 = 1.f;
    int gnr_bnd = 0x7fffffff;
    f32x4 gnr_gm = {0.f, 0.f, 0.f, 0.f}, gnr_bt = gnr_gm;
    if (p.gnr_y && col_ok) {
        const int cpg = p.N / p.gnr_groups, gl = nc / cpg - n0 / cpg;
        gnr_mu = gstat[gl];
        gnr_rs = gstat[4 + gl];
        gnr_mu1 = gstat[8 + gl];
        gnr_rs1 = gstat[12 + gl];
        gnr_bnd = (m0 / p.T_out + 1) * p.T_out;
        gnr_gm = *reinterpret_cast<const f32x4*>(p.gnr_gamma + nc);
        gnr_bt = *reinterpret_cast<const f32x4*>(p.gnr_beta + nc);
    }
    auto run = [&](auto act_c, auto res_c) {
        constexpr int ACT = decltype(act_c)::value;          // 0 none, 1 snake, 2 anything else (runtime switch)
        constexpr int RESK = decltype(res_c)::value;         // 0 none, 1 fp32 rows (p.res), 2 a P16 image (p.res16), 3 Block1D tail (gnr_*)
        constexpr bool RES = RESK == 1 || RESK == 3;          // both prefetch fp32 rows
        // chunks of U passes (4 U rows of the tile): all of a chunk's residual / mask loads are in flight together, and the
        // next chunk's are requested before this one is computed and stored (two register sets; the accumulators are dead)
        constexpr int U = 4;
        struct ChunkLoads {
            int orow[U];
            bool ok[U];
            float om[U], om16[U], gmk[U];
            f32x4 rres[U];
            f16x4 r16h[U], r16l[U];
        };
        auto load_chunk = [&](int c0, ChunkLoads& L) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + wm * (BM / 2) + (c0 + u) * 4 + (lane >> 4);
                L.ok[u] = m < M && col_ok;
                const int mc = m < M ? m : M - 1;
                int r = mc;
                if (!plain_rows) {
                    const int b = mc / p.T_out;
                    r = b * p.out_T + (mc - b * p.T_out) * p.out_stride + p.out_off;
                }
                L.orow[u] = r;
                L.om[u] = p.out_mask ? p.out_mask[r] : 1.0f;
                L.om16[u] = p.out16_mask ? p.out16_mask[r] : 1.0f;
                if constexpr (RESK == 1) L.rres[u] = *reinterpret_cast<const f32x4*>(p.res + (size_t)r * p.ldr + (col_ok ? nc : 0));
                if constexpr (RESK == 3) {
                    L.rres[u] = *reinterpret_cast<const f32x4*>(p.gnr_y + (size_t)r * p.N + (col_ok ? nc : 0));
                    L.gmk[u] = p.gnr_mask[r];
                }
                if constexpr (RESK == 2) {
                    const int ncl = col_ok ? nc : 0;
                    if (p.half16) {
                        L.r16h[u] = *reinterpret_cast<const f16x4*>(p.res16 + (size_t)r * p.ldr16 + ncl);
                        L.r16l[u] = f16x4{0, 0, 0, 0};
                    } else {
                        const _Float16* q = p.res16 + (size_t)r * p.ldr16 + (ncl >> 5) * 64 + (ncl & 31);
                        L.r16h[u] = *reinterpret_cast<const f16x4*>(q);
                        L.r16l[u] = *reinterpret_cast<const f16x4*>(q + 32);
                    }
                }
            }
        };
        auto process_chunk = [&](int c0, const ChunkLoads& L) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rl = (c0 + u) * 4 + (lane >> 4);
                const f32x4 a = *reinterpret_cast<const f32x4*>(Cw + rl * GEMM_CS + (lane & 15) * 4);
                f32x4 o;
                if constexpr (LN) {
                    const float mean = srow[wm * (BM / 2) + rl], rstd = srow[BM + wm * (BM / 2) + rl];
                    o = (a - mean * ws4) * rstd + bias4;
                } else {
                    o = a + bias4;
                }
                if (GN && gn) {
                    const bool in0 = rl < gn_cnt0, in1 = rl >= gn_bnd && rl < gn_bnd + gn_cnt1;
                    if (in0 || in1) {
                        const f32x4 d = o - (in0 ? gn_mu0 : gn_mu1);
                        const float t = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                        if (in0) gn_q0 += t; else gn_q1 += t;
                    }
                }
                if constexpr (ACT == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] + s1[e] * sin_sq(o[e] * s0[e]);     // SnakeBeta, reference transformer.py:75
                } else if constexpr (ACT == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = act_apply(o[e], p.act, s0[e], s1[e]);
                }
                o *= L.om[u];
                if (p.out_scale != 1.0f) o *= p.out_scale;
                if constexpr (RESK == 1) o += L.rres[u];
                if constexpr (RESK == 3) {           // Mish(GroupNorm(y)) * mask, same operation order as gn_apply_kernel
                    const bool second = L.orow[u] >= gnr_bnd;
                    f32x4 v = ((L.rres[u] - (second ? gnr_mu1 : gnr_mu)) * (second ? gnr_rs1 : gnr_rs)) * gnr_gm + gnr_bt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mish_f(v[e]) * L.gmk[u];
                    o += v;
                }
                if constexpr (RESK == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += (float)L.r16h[u][e] + (float)L.r16l[u][e] * (1.0f / F16_RES_SCALE);
                }
                if (L.ok[u]) {
                    if (p.out) *reinterpret_cast<f32x4*>(p.out + (size_t)L.orow[u] * p.ldc + nc) = o;
                    if (p.out16) {                           // P16 copy: 8 lanes write one whole 128-B line
                        f16x4 h, l;
                        range_bad |= out_of_f16_range(o[0], o[1], o[2], o[3]) && L.om16[u] != 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            _Float16 hh, ll;
                            split_f16(o[e] * L.om16[u], p.out_lscale, hh, ll);
                            h[e] = hh;
                            l[e] = ll;
                        }
                        if (p.half16) {
                            *reinterpret_cast<f16x4*>(p.out16 + (size_t)L.orow[u] * p.ld16 + nc) = h;
                        } else {
                            _Float16* o16 = p.out16 + (size_t)L.orow[u] * p.ld16 + (nc >> 5) * 64 + (nc & 31);
                            *reinterpret_cast<f16x4*>(o16) = h;
                            *reinterpret_cast<f16x4*>(o16 + 32) = l;
                        }
                    }
                }
                if (p.stats_out) {   // (mean, M2) of this wave's 64 columns of the row (N % 64 == 0): the 16 lanes lane&15 hold them
                    const float mu = allreduce16((o[0] + o[1]) + (o[2] + o[3])) * (1.0f / 64.0f);
                    const f32x4 d = o - mu;
                    const float m2 = allreduce16((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
                    if ((lane & 15) == 0 && L.ok[u]) {
                        float* so = p.stats_out + ((size_t)L.orow[u] * (p.N >> 6) + ((n0 + wn * 64) >> 6)) * 2;
                        so[0] = mu;
                        so[1] = m2;
                    }
                }
            }
        };
        ChunkLoads LA, LB;
        load_chunk(0, LA);
#pragma unroll
        for (int c0 = 0; c0 < NIT; c0 += 2 * U) {
            if (c0 + U < NIT) load_chunk(c0 + U, LB);
            process_chunk(c0, LA);
            if (c0 + 2 * U < NIT) load_chunk(c0 + 2 * U, LA);
            if (c0 + U < NIT) process_chunk(c0 + U, LB);
        }
    };
    const int actk = p.act == ACT_NONE ? 0 : (p.act == ACT_SNAKE ? 1 : 2);     // wave-uniform dispatch
    if (p.gnr_y) {             // ResNet output = residual conv + Block1D tail
        run(IntC<0>{}, IntC<3>{});
    } else if (p.res16) {      // the residual stream kept only as a P16 image (decoder transformer blocks)
        if (actk == 0) run(IntC<0>{}, IntC<2>{});
        else if (actk == 1) run(IntC<1>{}, IntC<2>{});
        else run(IntC<2>{}, IntC<2>{});
    } else if (p.res) {
        if (actk == 0) run(IntC<0>{}, IntC<1>{});
        else if (actk == 1) run(IntC<1>{}, IntC<1>{});
        else run(IntC<2>{}, IntC<1>{});
    } else {
        if (actk == 0) run(IntC<0>{}, IntC<0>{});
        else if (actk == 1) run(IntC<1>{}, IntC<0>{});
        else run(IntC<2>{}, IntC<0>{});
    }
    raise_range_flag(p.range_flag, range_bad);
    if constexpr (GN) if (gn) {
        const int row_w0 = m0 + wm * (BM / 2), col_wave = (n0 + wn * 64) >> 6;
        const float q00 = allreduce64(gn_gi == 0 ? gn_q0 : 0.f), q01 = allreduce64(gn_gi == 1 ? gn_q0 : 0.f);
        float q10 = 0.f, q11 = 0.f;
        if (gn_cnt1 > 0) { q10 = allreduce64(gn_gi == 0 ? gn_q1 : 0.f); q11 = allreduce64(gn_gi == 1 ? gn_q1 : 0.f); }
        if (lane == 0 && row_w0 < M) {
            const int tile = row_w0 / (BM / 2);
            float* e = p.gn_stats + ((size_t)((tile * 2) * (p.N >> 6) + col_wave) * 2) * 4;       // part 0
            *reinterpret_cast<f32x4*>(e) = f32x4{(float)(gn_cnt0 * gn_cols0), gn_mean[0][0], q00, 0.f};
            *reinterpret_cast<f32x4*>(e + 4) = f32x4{(float)(gn_cnt0 * (64 - gn_cols0)), gn_mean[0][1], q01, 0.f};
            float* e1 = e + (size_t)(p.N >> 6) * 8;                                                // part 1
            *reinterpret_cast<f32x4*>(e1) = f32x4{(float)(gn_cnt1 * gn_cols0), gn_mean[1][0], q10, 0.f};
            *reinterpret_cast<f32x4*>(e1 + 4) = f32x4{(float)(gn_cnt1 * (64 - gn_cols0)), gn_mean[1][1], q11, 0.f};
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same epilogue with EIGHT columns per lane (8 lanes per row, 8 rows per pass): every P16 / H16 image store and residual
// load is 16 bytes per lane instead of 8, fp32 rows go as two 16-byte stores.  An epilogue of 8-byte stores is store-ISSUE
// bound (MI355X_MICROARCH.md, epilogue store tail; profiles/r02_kstamp.log: 11.8k cycles of a 29k-cycle workgroup for the
// attention out-projection's 64 x 128 tile): half the passes, half the store and load instructions.  N % 4 == 0 as before; the
// second half of a lane's columns has its own validity (N = 100).  Used by gemm_p16.hip; gemm_f32.hip keeps the 4-column form.
//
// What is left after that is vector-instruction issue (profiles/r02_kstamp_sub.log: a 16-row chunk costs 3-4k cycles with two
// waves per SIMD, ~33 instructions per element in the ISA), so the element path is kept short:
//   * the residual image is widened by ONE v_fma_mix_f32 per element (h + l / 2^11 is exact in the fma: bit-identical to
//     cvt, cvt, mul, add);
//   * the fp16 split packs h with v_cvt_pk_f16_f32 and takes the residual c - h as a v_fma_mix_f32 on the packed register (no
//     second conversion, no unpack); l = (r * lscale) rounds once more to fp16 as before (v_fma_mixlo/hi_f16);
//   * the range guard keeps a running v_max3_f32 of |value after the image mask| (4 instructions per 8 elements, one compare
//     per workgroup at the end; a masked row cannot raise it: inf * 0 = NaN, which max3 drops, as `om16 != 0` did);
//   * out_mask and out_scale are one factor (the mask is 0 or 1: (c * m) * s == c * (m * s) bit for bit);
//   * the kernel arguments the passes need sit in SGPRs (pinned once; the compiler otherwise re-loads them from the kernarg
//     segment inside each pass, an s_load + s_waitcnt per use).
// The column constants (bias, SnakeBeta, panel sums) and -- for 64-row tiles -- the whole residual image tile are requested
// BEFORE the k-loop (EpiPre / epi_prefetch), so their round trip to L2 / HBM hides under it instead of opening the epilogue.
using f32x4_e = __attribute__((ext_vector_type(4))) float;
using f16x8_e = __attribute__((ext_vector_type(8))) _Float16;
using f16x2_e = __attribute__((ext_vector_type(2))) _Float16;
using u32x4_e = __attribute__((ext_vector_type(4))) unsigned int;

struct EpiCols {                                             // a lane's 8 columns: [nc, nc+4) and [nc+4, nc+8)
    f32x4_e bias[2], s0[2], s1[2], ws[2];
};
template <int BM>
struct EpiPre {                                              // residual image rows of a 64-row tile (<= 4 passes), else unused
    static constexpr int N = BM == 64 ? 4 : 1;
    f16x8_e h[N], l[N];
    float om[N], om16[N];                                    // out_mask / out16_mask of the pass's row (1 where there is no mask)
    bool valid;                                              // h / l hold the residual image
};

template <bool LN>
__device__ __forceinline__ EpiCols epi_load_cols(const GemmArgs& p, int n0, int wn, int lane) {
    const f32x4_e zero4 = {0.f, 0.f, 0.f, 0.f};
    EpiCols c;
    const int nc = n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        c.bias[h] = zero4; c.s0[h] = zero4; c.s1[h] = zero4; c.ws[h] = zero4;
        if (nc + 4 * h < p.N) {
            if (p.bias) c.bias[h] = *reinterpret_cast<const f32x4_e*>(p.bias + nc + 4 * h);
            if (p.act == ACT_SNAKE) { c.s0[h] = *reinterpret_cast<const f32x4_e*>(p.p0 + nc + 4 * h); c.s1[h] = *reinterpret_cast<const f32x4_e*>(p.p1 + nc + 4 * h); }
            if (LN) c.ws[h] = *reinterpret_cast<const f32x4_e*>(p.wsum + nc + 4 * h);
        }
    }
    return c;
}

// output row of tile row m (the epilogue's mapping: plain rows, or strided / offset rows of a longer output)
__device__ __forceinline__ int epi_out_row(const GemmArgs& p, int m, int M, bool plain_rows) {
    const int mc = m < M ? m : M - 1;
    if (plain_rows) return mc;
    const int b = fdiv(mc, p.T_out, p.rcp_T_out);
    return b * p.out_T + (mc - b * p.T_out) * p.out_stride + p.out_off;
}

// the row masks of a 64-row tile's passes: small L2-resident loads, requested BEFORE the first tiles (oldest in the counter)
template <int BM, int NPASS = 4>
__device__ __forceinline__ void epi_prefetch_masks(const GemmArgs& p, EpiPre<BM>& pre, int M, int m0, int wm, int lane, int row0 = 0) {
    if constexpr (BM == 64) {
        const bool plain_rows = (p.out_stride == 1 && p.out_off == 0 && p.out_T == p.T_out);
#pragma unroll
        for (int it = 0; it < NPASS; ++it) {
            const int r = epi_out_row(p, m0 + wm * 32 + row0 + it * 8 + (lane >> 3), M, plain_rows);
            pre.om[it] = p.out_mask ? p.out_mask[r] : 1.0f;
            pre.om16[it] = p.out16_mask ? p.out16_mask[r] : 1.0f;
        }
    }
}

template <int BM, int NPASS = 4>
__device__ __forceinline__ void epi_prefetch(const GemmArgs& p, EpiPre<BM>& pre, int M, int m0, int n0, int wm, int wn, int lane, int row0 = 0) {
    pre.valid = false;
    if constexpr (BM == 64) {
        if (p.res16 && !p.gnr_y) {                           // (an image implies N % 32 == 0: a lane's 8 columns are valid together)
            pre.valid = true;
            const bool plain_rows = (p.out_stride == 1 && p.out_off == 0 && p.out_T == p.T_out);
            const int nc = n0 + wn * 64 + (lane & 7) * 8, rg = lane >> 3;
            const int ncl = nc < p.N ? nc : 0;
#pragma unroll
            for (int it = 0; it < NPASS; ++it) {
                const int r = epi_out_row(p, m0 + wm * 32 + row0 + it * 8 + rg, M, plain_rows);
                if (p.half16) {
                    pre.h[it] = *reinterpret_cast<const f16x8_e*>(p.res16 + (size_t)r * p.ldr16 + ncl);
                    pre.l[it] = f16x8_e{0, 0, 0, 0, 0, 0, 0, 0};
                } else {
                    const _Float16* q = p.res16 + (size_t)r * p.ldr16 + (ncl >> 5) * 64 + (ncl & 31);
                    pre.h[it] = *reinterpret_cast<const f16x8_e*>(q);
                    pre.l[it] = *reinterpret_cast<const f16x8_e*>(q + 32);
                }
            }
        }
    }
}

// h + l / 2^11 of a packed pair of image halves (element 2k and 2k+1 of the 8): one v_fma_mix_f32 each
__device__ __forceinline__ void widen_pair(unsigned int hpk, unsigned int lpk, float& a, float& b) {
    const float k = 1.0f / F16_RES_SCALE;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(a) : "v"(lpk), "v"(k), "v"(hpk));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(b) : "v"(lpk), "v"(k), "v"(hpk));
}
// the fp16 split of two values (already multiplied by the image mask): packed h, packed l, running |max| for the range guard
template <bool WANT_L>
__device__ __forceinline__ void split_pair(float x0, float x1, float lscale, unsigned int& hpk, unsigned int& lpk, float& rmax) {
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(rmax) : "v"(x0), "v"(x1));
    const float c0 = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f), c1 = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
    const f16x2_e hp = {(_Float16)c0, (_Float16)c1};         // v_cvt_pk_f16_f32 (round to nearest even, as the scalar casts)
    hpk = __builtin_bit_cast(unsigned int, hp);
    lpk = 0u;
    if constexpr (WANT_L) {
        const float neg1 = -1.0f;
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hpk), "v"(neg1), "v"(c0));                      // c0 - h0 (exact)
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hpk), "v"(neg1), "v"(c1));
        const f16x2_e lp = {(_Float16)(r0 * lscale), (_Float16)(r1 * lscale)};
        lpk = __builtin_bit_cast(unsigned int, lp);
    }
}

// GroupNorm-statistics bookkeeping of a wave tile (wave-uniform; loaded before the k-loop so that the per-utterance row counts'
// round trip is off the epilogue): part 0 = tile rows [0, cnt0), part 1 = rows [bnd, bnd + cnt1)
struct EpiGnRows { int bnd, cnt0, cnt1; };
// row_w0: first row of the wave's (sub-)tile, rows_w: its height (= the statistics' tile height the consumers are told)
__device__ __forceinline__ EpiGnRows epi_gn_rows(const GemmArgs& p, int M, int row_w0, int rows_w) {
    EpiGnRows g = {rows_w, 0, 0};
    if (p.gn_stats && row_w0 < M) {
        const int b0 = fdiv(row_w0, p.T_out, p.rcp_T_out), t_w0 = row_w0 - b0 * p.T_out;
        const int nr0 = p.gn_nrows ? min(p.T_out, p.gn_nrows[b0]) : p.T_out;
        const int nr1 = (b0 + 1 < p.B) ? (p.gn_nrows ? min(p.T_out, p.gn_nrows[b0 + 1]) : p.T_out) : 0;
        g.bnd = min(rows_w, p.T_out - t_w0);
        g.cnt0 = max(0, min(g.bnd, nr0 - t_w0));
        g.cnt1 = max(0, min(rows_w - g.bnd, nr1));
    }
    return g;
}

// NPASS / row0: the passes this wave runs and the first wave-tile row they cover (split-K: the two wave sets share a tile's rows).
template <int BM, bool LN, bool GN = false, int NPASS = BM / 16>
__device__ __forceinline__ void gemm_epilogue_rows8(const GemmArgs& p, const EpiCols& cols, const EpiPre<BM>& pre, const EpiGnRows& gnr0, const float* __restrict__ Cw,
                                                    const float* __restrict__ Cw2, const float* __restrict__ srow, int M, int m0, int n0, int wm, int wn, int lane,
                                                    const float* __restrict__ gstat = nullptr, int row0 = 0) {
    using f32x4 = f32x4_e;
    using f16x8 = f16x8_e;
    // (GroupNorm statistics are per NPASS * 8 rows: the whole wave tile, or the 32-row half of a 128-row kernel's wave tile that a
    // split-K wave set finishes -- the height gemm_p16_wave_rows() reports either way)
    constexpr int NIT = NPASS;                               // passes: 8 rows per pass
    const bool plain_rows = (p.out_stride == 1 && p.out_off == 0 && p.out_T == p.T_out);
    const int co = lane & 7, rg = lane >> 3;
    const int nc = n0 + wn * 64 + co * 8;                   // first of this lane's 8 columns
    const bool ok_a = nc < p.N, ok_b = nc + 4 < p.N;        // halves [nc, nc+4) and [nc+4, nc+8)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias[2] = {cols.bias[0], cols.bias[1]}, s0[2] = {cols.s0[0], cols.s0[1]}, s1[2] = {cols.s1[0], cols.s1[1]},
                ws[2] = {cols.ws[0], cols.ws[1]};
    [[maybe_unused]] const f32x4 s1h[2] = {cols.s1[0] * 0.5f, cols.s1[1] * 0.5f};
    // the arguments the passes use, pinned in SGPRs
    // (the empty asm makes each an SGPR value the compiler cannot re-load; a laundered pointer is generic, so it is cast back
    // to the global address space: flat stores would also tick lgkmcnt)
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) const float gcfloat;
    typedef __attribute__((address_space(1))) _Float16 ghalf;
    unsigned long long q_out = (unsigned long long)p.out, q_out16 = (unsigned long long)p.out16, q_stats = (unsigned long long)p.stats_out,
                       q_om = (unsigned long long)p.out_mask, q_om16 = (unsigned long long)p.out16_mask;
    int a_ldc = p.ldc, a_ld16 = p.ld16, a_half = p.half16 ? 1 : 0, a_nw = p.N >> 6, a_bf = p.bf16 ? 1 : 0;
    float a_scale = p.out_scale, a_lscale = p.out_lscale;
    asm volatile("" : "+s"(q_out), "+s"(q_out16), "+s"(q_stats), "+s"(q_om), "+s"(q_om16));
    asm volatile("" : "+s"(a_ldc), "+s"(a_ld16), "+s"(a_half), "+s"(a_nw), "+s"(a_scale), "+s"(a_lscale), "+s"(a_bf));
    gfloat* const a_out = (gfloat*)q_out;
    ghalf* const a_out16 = (ghalf*)q_out16;
    gfloat* const a_stats = (gfloat*)q_stats;
    gcfloat* const a_om = (gcfloat*)q_om;
    gcfloat* const a_om16 = (gcfloat*)q_om16;
    auto sum4 = [](const f32x4& a) { return (a[0] + a[1]) + (a[2] + a[3]); };
    auto sq4 = [](const f32x4& d) { return (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]); };
    // ---- GroupNorm statistics (see gemm_epilogue_rows): per wave tile, utterance part and group slice; a lane's 8 columns lie in
    // one group (channels per group % 8 == 0)
    const bool gn = GN && p.gn_stats != nullptr;
    int gn_gi = 0, gn_cols0 = 64;
    const int gn_bnd = gnr0.bnd, gn_cnt0 = gnr0.cnt0, gn_cnt1 = gnr0.cnt1;
    float gn_mean[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gn_mu0 = 0.f, gn_mu1 = 0.f, gn_q0 = 0.f, gn_q1 = 0.f;
    if constexpr (GN) if (gn) {
        const int cpg = p.gn_cpg, n0w = n0 + wn * 64, g0 = fdiv(n0w, cpg, p.rcp_gn_cpg);
        gn_cols0 = min(64, (g0 + 1) * cpg - n0w);
        gn_gi = nc - n0w >= gn_cols0 ? 1 : 0;                // (channels per group >= 32: a wave's 64 columns lie in at most two groups)
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rl = it * 8 + rg;                      // row inside this wave's (sub-)tile
            const bool in0 = rl < gn_cnt0, in1 = rl >= gn_bnd && rl < gn_bnd + gn_cnt1;
            if (in0 || in1) {
                const float* q = Cw + (row0 + rl) * GEMM_CS + co * 8;
                f32x4 qa = *reinterpret_cast<const f32x4*>(q), qb = *reinterpret_cast<const f32x4*>(q + 4);
                if (Cw2) {
                    const float* q2 = Cw2 + (row0 + rl) * GEMM_CS + co * 8;
                    qa += *reinterpret_cast<const f32x4*>(q2);
                    qb += *reinterpret_cast<const f32x4*>(q2 + 4);
                }
                const float t = sum4(qa + bias[0]) + sum4(qb + bias[1]);
                if (in0) t0 += t; else t1 += t;
            }
        }
        const float c0f = (float)gn_cols0, c1f = (float)(64 - gn_cols0);
        const float t00 = allreduce64(gn_gi == 0 ? t0 : 0.f), t01 = allreduce64(gn_gi == 1 ? t0 : 0.f);
        gn_mean[0][0] = gn_cnt0 > 0 ? t00 / ((float)gn_cnt0 * c0f) : 0.f;
        gn_mean[0][1] = (gn_cnt0 > 0 && gn_cols0 < 64) ? t01 / ((float)gn_cnt0 * c1f) : 0.f;
        if (gn_cnt1 > 0) {
            const float t10 = allreduce64(gn_gi == 0 ? t1 : 0.f), t11 = allreduce64(gn_gi == 1 ? t1 : 0.f);
            gn_mean[1][0] = t10 / ((float)gn_cnt1 * c0f);
            gn_mean[1][1] = gn_cols0 < 64 ? t11 / ((float)gn_cnt1 * c1f) : 0.f;
        }
        gn_mu0 = gn_gi == 0 ? gn_mean[0][0] : gn_mean[0][1];
        gn_mu1 = gn_gi == 0 ? gn_mean[1][0] : gn_mean[1][1];
    }
    float rmax = 0.f;                                        // range guard: running max of |value written to an image|
    float gnr_mu = 0.f, gnr_rs = 1.f, gnr_mu1 = 0.f, gnr_rs1 = 1.f;
    int gnr_bnd = 0x7fffffff;
    f32x4 gnr_gm[2] = {zero4, zero4}, gnr_bt[2] = {zero4, zero4};
    if (p.gnr_y && ok_a) {
        const int cpg = p.gnr_cpg, gl = fdiv(nc, cpg, p.rcp_gnr_cpg) - fdiv(n0, cpg, p.rcp_gnr_cpg);
        gnr_mu = gstat[gl];
        gnr_rs = gstat[4 + gl];
        gnr_mu1 = gstat[8 + gl];
        gnr_rs1 = gstat[12 + gl];
        gnr_bnd = (fdiv(m0, p.T_out, p.rcp_T_out) + 1) * p.T_out;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            gnr_gm[h] = *reinterpret_cast<const f32x4*>(p.gnr_gamma + nc + 4 * h);
            gnr_bt[h] = *reinterpret_cast<const f32x4*>(p.gnr_beta + nc + 4 * h);
        }
    }
    auto run = [&](auto act_c, auto res_c) {
        constexpr int ACT = decltype(act_c)::value;          // 0 none, 1 snake, 2 anything else (runtime switch)
        constexpr int RESK = decltype(res_c)::value;         // 0 none, 1 fp32 rows, 2 a P16 / H16 image, 3 Block1D tail
        constexpr int U = 2;                                 // passes per chunk: 16 rows, all their loads in flight together
        struct ChunkLoads {
            int orow[U];
            bool ok[U];
            float om[U], om16[U], gmk[U];
            f32x4 rres[U][2];
            f16x8 r16h[U], r16l[U];
        };
        auto load_chunk = [&](auto c0_c, ChunkLoads& L) {
            constexpr int c0 = decltype(c0_c)::value;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + wm * (BM / 2) + row0 + (c0 + u) * 8 + rg;
                L.ok[u] = m < M;
                const int r = epi_out_row(p, m, M, plain_rows);
                L.orow[u] = r;
                if constexpr (BM == 64) {                    // requested before the k-loop (epi_prefetch_masks)
                    L.om[u] = pre.om[c0 + u] * a_scale;
                    L.om16[u] = pre.om16[c0 + u];
                } else {
                    L.om[u] = (a_om ? a_om[r] : 1.0f) * a_scale;    // the mask is 0 or 1: one factor serves both
                    L.om16[u] = a_om16 ? a_om16[r] : 1.0f;
                }
                const int ncl = ok_a ? nc : 0, ncb = ok_b ? nc + 4 : 0;
                if constexpr (RESK == 1) {
                    L.rres[u][0] = *reinterpret_cast<const f32x4*>(p.res + (size_t)r * p.ldr + ncl);
                    L.rres[u][1] = *reinterpret_cast<const f32x4*>(p.res + (size_t)r * p.ldr + ncb);
                }
                if constexpr (RESK == 3) {
                    L.rres[u][0] = *reinterpret_cast<const f32x4*>(p.gnr_y + (size_t)r * p.N + ncl);
                    L.rres[u][1] = *reinterpret_cast<const f32x4*>(p.gnr_y + (size_t)r * p.N + ncb);
                    L.gmk[u] = p.gnr_mask[r];
                }
                if constexpr (RESK == 2) {                   // N % 32 == 0 here: both halves valid together
                    if (BM == 64 && pre.valid) {             // requested beside the first tiles (epi_prefetch)
                        L.r16h[u] = pre.h[BM == 64 ? c0 + u : 0];
                        L.r16l[u] = pre.l[BM == 64 ? c0 + u : 0];
                    } else if (a_half) {
                        L.r16h[u] = *reinterpret_cast<const f16x8*>(p.res16 + (size_t)r * p.ldr16 + ncl);
                        L.r16l[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    } else {
                        const _Float16* q = p.res16 + (size_t)r * p.ldr16 + (ncl >> 5) * 64 + (ncl & 31);
                        L.r16h[u] = *reinterpret_cast<const f16x8*>(q);
                        L.r16l[u] = *reinterpret_cast<const f16x8*>(q + 32);
                    }
                }
            }
        };
        auto process_chunk = [&](int c0, const ChunkLoads& L) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rl = row0 + (c0 + u) * 8 + rg;
                const float* q = Cw + rl * GEMM_CS + co * 8;
                f32x4 o[2] = {*reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4)};
                if (Cw2) {                                   // split-K: the second wave set's partial sums
                    const float* q2 = Cw2 + rl * GEMM_CS + co * 8;
                    o[0] += *reinterpret_cast<const f32x4*>(q2);
                    o[1] += *reinterpret_cast<const f32x4*>(q2 + 4);
                }
                if constexpr (LN) {      // rstd (x.W' - mean rowsum(W')) + bias as two FMAs per element: x.W' rstd + (bias - mean rstd rowsum)
                    const float mean = srow[wm * (BM / 2) + rl], rstd = srow[BM + wm * (BM / 2) + rl];
                    const float nmr = -(mean * rstd);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[h][e] = __builtin_fmaf(o[h][e], rstd, __builtin_fmaf(nmr, ws[h][e], bias[h][e]));
                } else {
                    o[0] += bias[0];
                    o[1] += bias[1];
                }
                if (GN && gn) {
                    const int rs = rl - row0;
                    const bool in0 = rs < gn_cnt0, in1 = rs >= gn_bnd && rs < gn_bnd + gn_cnt1;
                    if (in0 || in1) {
                        const float mu = in0 ? gn_mu0 : gn_mu1;
                        const float t = sq4(o[0] - mu) + sq4(o[1] - mu);
                        if (in0) gn_q0 += t; else gn_q1 += t;
                    }
                }
                [[maybe_unused]] u32x4_e rh, rlw;
                if constexpr (RESK == 2) {
                    rh = __builtin_bit_cast(u32x4_e, L.r16h[u]);
                    rlw = __builtin_bit_cast(u32x4_e, L.r16l[u]);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if constexpr (ACT == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {                                                          // SnakeBeta
#ifdef MTTS_SNAKE_POLY
                            o[h][e] = o[h][e] + s1[h][e] * sin_sq(o[h][e] * s0[h][e]);
#else
                            o[h][e] = snake_hw(o[h][e], s0[h][e], s1h[h][e]);
#endif
                        }
                    } else if constexpr (ACT == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[h][e] = act_apply(o[h][e], p.act, s0[h][e], s1[h][e]);
                    }
                    o[h] *= L.om[u];
                    if constexpr (RESK == 1) o[h] += L.rres[u][h];
                    if constexpr (RESK == 3) {       // Mish(GroupNorm(y)) * mask, same operation order as gn_apply_kernel
                        const bool second = L.orow[u] >= gnr_bnd;
                        f32x4 v = ((L.rres[u][h] - (second ? gnr_mu1 : gnr_mu)) * (second ? gnr_rs1 : gnr_rs)) * gnr_gm[h] + gnr_bt[h];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = mish_f(v[e]) * L.gmk[u];
                        o[h] += v;
                    }
                    if constexpr (RESK == 2) {
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            float ra, rb;
                            if (a_bf) unpack_bf16(rh[2 * h + e2], ra, rb);          // (bfloat16 planes: one term)
                            else widen_pair(rh[2 * h + e2], rlw[2 * h + e2], ra, rb);
                            o[h][2 * e2] += ra;
                            o[h][2 * e2 + 1] += rb;
                        }
                    }
                }
                if (L.ok[u]) {
                    if (a_out) {
                        gfloat* dst = a_out + (size_t)L.orow[u] * a_ldc + nc;
                        if (ok_a) *(__attribute__((address_space(1))) f32x4*)dst = o[0];
                        if (ok_b) *(__attribute__((address_space(1))) f32x4*)(dst + 4) = o[1];
                    }
                    if (a_out16 && ok_a) {                   // (N % 32 == 0 with an image: both halves valid) 8 lanes = one 128-B line
                        const float m16 = L.om16[u];
                        if (a_half) {                        // H16: one plane
                            u32x4_e hv;
                            unsigned int lp;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                unsigned int hp;
                                if (a_bf) hp = pack_bf16(o[k >> 1][2 * (k & 1)] * m16, o[k >> 1][2 * (k & 1) + 1] * m16);
                                else split_pair<false>(o[k >> 1][2 * (k & 1)] * m16, o[k >> 1][2 * (k & 1) + 1] * m16, a_lscale, hp, lp, rmax);
                                hv[k] = hp;
                            }
                            *(__attribute__((address_space(1))) u32x4_e*)(a_out16 + (size_t)L.orow[u] * a_ld16 + nc) = hv;
                        } else {
                            u32x4_e hv, lv;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                unsigned int hp, lp;
                                split_pair<true>(o[k >> 1][2 * (k & 1)] * m16, o[k >> 1][2 * (k & 1) + 1] * m16, a_lscale, hp, lp, rmax);
                                hv[k] = hp;
                                lv[k] = lp;
                            }
                            ghalf* o16 = a_out16 + (size_t)L.orow[u] * a_ld16 + (nc >> 5) * 64 + (nc & 31);
                            *(__attribute__((address_space(1))) u32x4_e*)o16 = hv;
                            *(__attribute__((address_space(1))) u32x4_e*)(o16 + 32) = lv;
                        }
                    }
                }
                if (a_stats) {   // (mean, M2) of this wave's 64 columns of the row: the 8 lanes of the row hold them
                    const float mu = allreduce8(sum4(o[0]) + sum4(o[1])) * (1.0f / 64.0f);
                    const float m2 = allreduce8(sq4(o[0] - mu) + sq4(o[1] - mu));
                    if (co == 0 && L.ok[u]) {
                        gfloat* so = a_stats + ((size_t)L.orow[u] * a_nw + ((n0 + wn * 64) >> 6)) * 2;
                        so[0] = mu;
                        so[1] = m2;
                    }
                }
            }
        };
        ChunkLoads LA, LB;
        load_chunk(IntC<0>{}, LA);
        if constexpr (NIT > U) load_chunk(IntC<U>{}, LB);
        process_chunk(0, LA);
#ifndef MTTS_KSTAMP_SETUP
        MTTS_STAMP(7);
#endif
        if constexpr (NIT > 2 * U) load_chunk(IntC<2 * U>{}, LA);
        if constexpr (NIT > U) process_chunk(U, LB);
        if constexpr (NIT > 2 * U) {
            load_chunk(IntC<3 * U>{}, LB);
            process_chunk(2 * U, LA);
            process_chunk(3 * U, LB);
        }
    };
    const int actk = p.act == ACT_NONE ? 0 : (p.act == ACT_SNAKE ? 1 : 2);     // wave-uniform dispatch
    if (p.gnr_y) {
        run(IntC<0>{}, IntC<3>{});
    } else if (p.res16) {
        if (actk == 0) run(IntC<0>{}, IntC<2>{});
        else if (actk == 1) run(IntC<1>{}, IntC<2>{});
        else run(IntC<2>{}, IntC<2>{});
    } else if (p.res) {
        if (actk == 0) run(IntC<0>{}, IntC<1>{});
        else if (actk == 1) run(IntC<1>{}, IntC<1>{});
        else run(IntC<2>{}, IntC<1>{});
    } else {
        if (actk == 0) run(IntC<0>{}, IntC<0>{});
        else if (actk == 1) run(IntC<1>{}, IntC<0>{});
        else run(IntC<2>{}, IntC<0>{});
    }
    raise_range_flag(p.range_flag, rmax > 65504.f);
    if constexpr (GN) if (gn) {
        const int row_w0 = m0 + wm * (BM / 2) + row0, col_wave = (n0 + wn * 64) >> 6;
        const float q00 = allreduce64(gn_gi == 0 ? gn_q0 : 0.f), q01 = allreduce64(gn_gi == 1 ? gn_q0 : 0.f);
        float q10 = 0.f, q11 = 0.f;
        if (gn_cnt1 > 0) { q10 = allreduce64(gn_gi == 0 ? gn_q1 : 0.f); q11 = allreduce64(gn_gi == 1 ? gn_q1 : 0.f); }
        if (lane == 0 && row_w0 < M) {
            const int tile = row_w0 / (NPASS * 8);
            float* e = p.gn_stats + ((size_t)((tile * 2) * (p.N >> 6) + col_wave) * 2) * 4;
            *reinterpret_cast<f32x4*>(e) = f32x4{(float)(gn_cnt0 * gn_cols0), gn_mean[0][0], q00, 0.f};
            *reinterpret_cast<f32x4*>(e + 4) = f32x4{(float)(gn_cnt0 * (64 - gn_cols0)), gn_mean[0][1], q01, 0.f};
            float* e1 = e + (size_t)(p.N >> 6) * 8;
            *reinterpret_cast<f32x4*>(e1) = f32x4{(float)(gn_cnt1 * gn_cols0), gn_mean[1][0], q10, 0.f};
            *reinterpret_cast<f32x4*>(e1 + 4) = f32x4{(float)(gn_cnt1 * (64 - gn_cols0)), gn_mean[1][1], q11, 0.f};
        }
    }
}


}  // namespace mtts
