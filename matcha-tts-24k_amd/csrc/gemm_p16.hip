// GEMM / implicit conv1d on operands that are ALREADY fp16-split in memory ("P16" images, kernels.h) for gfx950.
//
// Same arithmetic as gemm_f32.hip's TERMS = 2 mode (x = h + l / 2^11, products h.h + (h.l + l.h) / 2^11 accumulated in fp32
// by three v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block, two accumulators), but the producer of the activations has done the
// split once, in its epilogue, so this kernel's loop has no split arithmetic, no staging registers and no LDS stores: both
// tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4), double-buffered, one barrier per k-step.
//
//   LDS image of a tile: rows of 128 B = [32 heads | 32 residuals] of one 32-wide k group, lane-linear per DMA piece
//   (a piece = one wave instruction = 8 rows x 128 B).  16-B chunk c of row r lives in slot c ^ ((r >> 1) & 7): the
//   involution is applied to the per-lane SOURCE address of the DMA and again by the fragment reads, which makes the
//   32-row x 16-B fragment reads conflict-free.
//   Every operand row of a k-step is one whole 128-B line: half-line requests measured ~10 % slower (tools/gemm_lab_planes.hip).
//
// What the loop cannot do any more moves to the epilogue: LayerNorm of the A rows is applied as
//   LN(x) . W'^T = rstd * (x . W'^T - mean * wsum),   wsum[n] = sum_k W'[n][k]
// (exact algebra; the row moments come from the producer's 64-column partials or from arrays), and row masks are the
// producer's job (P16 images are written already masked).
// Replaces, like gemm_f32.hip: nn.Linear / nn.Conv1d of the reference decoder (decoder.py, transformer.py) on the hot path.
#include "kernels.h"
#include "device_utils.h"
#ifdef MTTS_KSTAMP
#define MTTS_STAMP(i) do { if (p.kstamp && threadIdx.x == 0) p.kstamp[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MTTS_STAMP_RT(i) do { if (p.kstamp && threadIdx.x == 0) p.kstamp[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MTTS_STAMP(i) do { } while (0)
#define MTTS_STAMP_RT(i) do { } while (0)
#endif
#include "gemm_epilogue.h"
#include <cstdlib>
#include <string>


namespace mtts {

#ifdef MTTS_KSTAMP
static unsigned long long* g_kstamp = nullptr;            // diagnostic build: where the next launches put their stamps
static int g_kstamp_skip = 0;                              // P16 GEMM launches to let pass first (in-situ: the n-th launch of a step)
static int g_kstamp_info[8];                               // the stamped launch: M, N, K, tile rows, stages, wave sets, taps, flags
extern "C" void mtts_debug_set_kstamp(unsigned long long* p) { g_kstamp = p; g_kstamp_skip = 0; }
extern "C" void mtts_debug_set_kstamp_nth(unsigned long long* p, int n) { g_kstamp = p; g_kstamp_skip = n; }
extern "C" void mtts_debug_kstamp_info(int* out) { for (int i = 0; i < 8; ++i) out[i] = g_kstamp_info[i]; }
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// one 16-bit MFMA on raw 16-byte operand fragments: fp16 or (BF) bfloat16 planes
template <bool BF>
__device__ __forceinline__ f32x4 mfma16_raw(f16x8 a, f16x8 b, f32x4 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ f32x16 mfma32_raw(f16x8 a, f16x8 b, f32x16 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __attribute__((aligned(128))) _Float16 g_p16_zero_line[64];     // source of out-of-range conv taps (zero-initialised)

constexpr int P16_CS = GEMM_CS;                                             // epilogue tile row stride (floats)
constexpr int p16_stage_bytes(int BM) { return (BM + GEMM_BN) * 128; }
constexpr int p16_epi_bytes(int BM) { return 4 * (BM / 2) * P16_CS * 4; }
constexpr int p16_main_bytes(int BM, int NST, int KS = 1) {
    return KS * NST * p16_stage_bytes(BM) > KS * p16_epi_bytes(BM) ? KS * NST * p16_stage_bytes(BM) : KS * p16_epi_bytes(BM);
}
constexpr int p16_lds_bytes(int BM, int NST, int KS = 1) { return p16_main_bytes(BM, NST, KS) + 2 * BM * 4; }   // + per-row (mean, rstd)

#define MTTS_GLDS16(gp, lp) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

// NST = 2: two LDS stages, the next tile requested at the top of a k-step and waited for at its end (2-3 workgroups per
// CU hide each other's waits).  NST = 3 / 4 (small grids, <= 2 / 1 workgroups per CU, where nothing else hides the DMA
// round trip): a ring with NST-1 tiles in flight across the barrier -- counted s_waitcnt vmcnt, raw s_barrier -- so a
// k-step costs its MFMAs instead of a full global->LDS latency (B <= 8 serving shapes: 0.75 us -> ~0.3 us per k-step).
// ONE: the opt-in fp16 mode (MTTS_GEMM_TERMS=1): only the head planes are multiplied (one MFMA per 16-deep block instead of
// three, half the fragment reads) -- fp16 operand precision with fp32 accumulation, what torch.autocast gives the reference.
// M16: v_mfma_f32_16x16x32_f16 tiles (MT x 4 per wave) instead of 32x32x16 (MI x 2).
// MODE: 0 = P16 operands, three products per MAC (the default, fp32-equivalent); 1 = P16 operands, heads x heads only (the
// opt-in fp16 mode, ONE); 2 = H16 operands (GemmArgs::half16): a 128-byte line holds 64 k of one fp16 plane, a k-step is 64
// deep and its two 32-k halves are what the head / residual chunks of a P16 line are to the DMA and the fragment reads.
// KS: intra-workgroup split-K.  A grid of at most one workgroup per CU (B = 32 at the half-length level, every serving shape)
// leaves one wave per SIMD, and a k-step then costs the wave's own serial chain (fragment reads -> MFMAs -> DMA issue, ~1.05k
// cycles for 384 cycles of MFMA: profiles/r02_kstamp.log).  With KS = 2 the workgroup has 8 waves: waves 0-3 run the first half
// of the K axis, waves 4-7 the second half, each set on its own ring of LDS stages -- two waves per SIMD hide each other's
// chains exactly as two co-resident workgroups do, without needing a second tile.  Both sets park their partial tiles; waves
// 0-3 add them in the epilogue (fp32; the order of the two partial sums is fixed, so results are run-to-run identical).
template <int BM, bool LN, int NST, int MODE, bool M16, bool GN = false, int KS = 1>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void gemm_p16_kernel(const GemmArgs p) {
    static_assert(!(GN && LN), "GroupNorm statistics come from conv GEMMs, which have no LayerNorm prologue");
    constexpr bool ONE = MODE == 1;
    constexpr bool HALF = MODE == 2 || MODE == 3;          // MODE 3: the H16 planes hold bfloat16 (GemmArgs::bf16)
    constexpr bool BF = MODE == 3;
    constexpr int KSTEP = HALF ? 64 : 32;                  // k elements per 128-byte line
    constexpr int APW = BM / 32;           // A pieces (8 rows x 128 B) a wave moves per k-step; W: 4 per wave
    constexpr int STAGE = p16_stage_bytes(BM);
    extern __shared__ __attribute__((aligned(16))) char lds[];   // ONE array: stages | epilogue tile | row statistics
    float* srow = reinterpret_cast<float*>(lds + p16_main_bytes(BM, NST, KS));

    const int tid_all = threadIdx.x;
    const int ks = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid_all >> 8);      // which part of the K axis (wave-uniform)
    const int tid = tid_all & 255, lane = tid & 63;                                 // position inside the 4-wave set
    // the wave index as a SCALAR: every LDS destination of a DMA piece is then an SGPR expression (s_add + s_mov m0) instead of
    // a v_readfirstlane per piece, and the fragment / tile bases derived from it stay off the vector ALU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const lds_k = lds + ks * (NST * STAGE);                                   // this set's ring
    const int wm = wave >> 1, wn = wave & 1;
    const int M = p.B * p.T_out;
    const int Kp = p.ntaps * p.ktap;
    const int n_tiles = (p.N + GEMM_BN - 1) / GEMM_BN;
    int swz;   // XCD-aware order: the N-tiles of one M-tile run consecutively on one XCD (same remap as gemm_f32.hip)
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int mt = fdiv(swz, n_tiles, p.rcp_ntiles);
    const int m0 = mt * BM;
    const int n0 = (swz - mt * n_tiles) * GEMM_BN;

    // ---- DMA coordinates.  Piece pa of the A tile = rows 8 pa .. 8 pa + 7; lane i fills LDS bytes [16 i, 16 i + 16) of the
    // piece = row i>>3, slot i&7, which must hold chunk (i&7) ^ ((row>>1)&7), row>>1 = 4 pa + (i>>4).
    int a_base[APW], a_t[APW], a_chunk[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int pa = wave * APW + j;
        const int m = m0 + pa * 8 + (lane >> 3);
        a_chunk[j] = ((lane & 7) ^ (((pa & 1) * 4 + (lane >> 4)) & 7)) * 8;
        if (m < M) {
            const int b = fdiv(m, p.T_out, p.rcp_T_out);
            a_base[j] = b * p.T_in;
            a_t[j] = (m - b * p.T_out) * p.in_stride;
        } else {
            a_base[j] = 0;
            a_t[j] = -(1 << 28);
        }
    }
    const _Float16* wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pw = wave * 4 + j;
        const int chunk = (lane & 7) ^ (((pw & 1) * 4 + (lane >> 4)) & 7);
        wsrc[j] = HALF ? reinterpret_cast<const _Float16*>(p.w16h) + (size_t)(n0 + pw * 8 + (lane >> 3)) * Kp + chunk * 8
                       : reinterpret_cast<const _Float16*>(p.w16) + (size_t)(n0 + pw * 8 + (lane >> 3)) * Kp * 2 + chunk * 8;
    }
    // The K axis is a sequence of runs, one per (tap, channel segment); inside a run every k-step only advances the source
    // pointers by one 128-B group, so the per-lane address arithmetic (tap shift, sequence bounds, segment base) is done once
    // per run and the loop body is pointer bumps + DMA issue.
    const _Float16* asrc[APW];
    int astep[APW];
    int run_tap = 0, run_seg = 0, run_left = 0;
    auto setup_run = [&]() {
        const bool seg1 = run_seg == 1;                                  // wave-uniform
        const _Float16* src = seg1 ? p.a16_1 : p.a16_0;
        const int ld = seg1 ? p.lda16_1 : p.lda16_0;
        const int off = p.tap_off[run_tap];
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int tin = a_t[j] + off;
            const bool ok = (unsigned)tin < (unsigned)p.T_in;
            asrc[j] = ok ? src + (size_t)(a_base[j] + tin) * ld + a_chunk[j] : g_p16_zero_line + a_chunk[j];
            astep[j] = ok ? 64 : 0;                                      // the zero line is re-read, not walked
        }
        run_left = (seg1 ? p.c1 : p.c0) / KSTEP;
    };
    auto issue = [&](int buf) {
        char* st = lds_k + buf * STAGE;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            MTTS_GLDS16(asrc[j], st + (wave * APW + j) * 1024);
            asrc[j] += astep[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            MTTS_GLDS16(wsrc[j], st + BM * 128 + (wave * 4 + j) * 1024);
            wsrc[j] += 64;
        }
        if (--run_left == 0) {                                           // next run: other segment, then next tap
            if (run_seg == 0 && p.a16_1 != nullptr) run_seg = 1;
            else { run_seg = 0; ++run_tap; }
            if (run_tap < p.ntaps) setup_run();
        }
    };

    // v_mfma_f32_16x16x32_f16 tiles (MT x 4 per wave; one instruction spans the whole 32-deep k-step): on random operands the
    // chip holds a higher clock with this shape than with 32x32x16 at equal cycles per FLOP -- 5-11 % faster launches on the
    // decoder's shapes (tools/gemm_lab_planes.hip -DM16=1; MI355X_MICROARCH.md, clock notes).
    constexpr int MT = BM / 32;            // 16-row tiles per wave along M
    constexpr int MI = BM / 64;            // 32-row tiles per wave along M (32x32x16 path)
    f32x16 acc32[MI][2], accx32[MI][2];      // 32x32x16 path (!M16); the unused set is dead code
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc32[i][j][r] = 0.f; accx32[i][j][r] = 0.f; }

    const int fr32 = lane & 31, fh32 = lane >> 5, f832 = (fr32 >> 1) & 7;
    f32x4 acc[MT][4], accx[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][j] = acc[i][j]; }

    // operand fragment: lane (r = lane&15, q = lane>>4) holds k = 8q .. 8q+7 of row r -- head chunk q, residual chunk 4 + q
    const int fr = lane & 15, fq = lane >> 4, f8 = (fr >> 1) & 7;
    const int nk = Kp / KSTEP / KS;                        // k-steps of this set (the host checks divisibility)
    auto compute16 = [&](const char* stage) {
        const char* sa = stage + (wm * (BM / 2) + fr) * 128;
        const char* sw = stage + BM * 128 + (wn * 64 + fr) * 128;
        const int sh = (fq ^ f8) * 16, sl = ((4 + fq) ^ f8) * 16;      // (row >> 1) & 7 == (fr >> 1) & 7: tile bases are multiples of 16
        constexpr int JB = (BM == 128 && KS > 1) ? 2 : 4;       // weight fragments held at a time (the 8-wave 128-row kernel has 256 registers per wave)
        f16x8 ah[MT], al[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(sa + i * 16 * 128 + sh);
            if constexpr (!ONE) al[i] = *reinterpret_cast<const f16x8*>(sa + i * 16 * 128 + sl);
        }
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += JB) {
            if constexpr (JB < 4) __builtin_amdgcn_sched_barrier(0);      // (keep the next pair's fragment reads behind this pair's MFMAs: registers)
            f16x8 bh[JB], bl[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(sw + (j0 + j) * 16 * 128 + sh);
                if constexpr (!ONE) bl[j] = *reinterpret_cast<const f16x8*>(sw + (j0 + j) * 16 * 128 + sl);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    if constexpr (HALF) {          // second 32-k half of the line
                        acc[i][j0 + j] = mfma16_raw<BF>(al[i], bl[j], acc[i][j0 + j]);
                    } else if constexpr (!ONE) {
                        accx[i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], accx[i][j0 + j], 0, 0, 0);
                        accx[i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], accx[i][j0 + j], 0, 0, 0);
                    }
                    acc[i][j0 + j] = mfma16_raw<BF>(ah[i], bh[j], acc[i][j0 + j]);
                }
        }
    };
    auto compute32 = [&](const char* stage) {
        const char* sa = stage + (wm * (BM / 2) + fr32) * 128;
        const char* sw = stage + BM * 128 + (wn * 64 + fr32) * 128;
#pragma unroll
        for (int kb = 0; kb < (HALF ? 4 : 2); ++kb) {      // H16: the line's eight 16-byte chunks are four 16-k blocks of one plane
            const int sh = ((2 * kb + fh32) ^ f832) * 16, sl = ((4 + 2 * kb + fh32) ^ f832) * 16;
            f16x8 ah[MI], al[MI], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(sa + i * 32 * 128 + sh);
                if constexpr (MODE == 0) al[i] = *reinterpret_cast<const f16x8*>(sa + i * 32 * 128 + sl);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(sw + j * 32 * 128 + sh);
                if constexpr (MODE == 0) bl[j] = *reinterpret_cast<const f16x8*>(sw + j * 32 * 128 + sl);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (MODE == 0) {
                        accx32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx32[i][j], 0, 0, 0);
                        accx32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx32[i][j], 0, 0, 0);
                    }
                    acc32[i][j] = mfma32_raw<BF>(ah[i], bh[j], acc32[i][j]);
                }
        }
    };
    auto compute = [&](const char* stage) { if constexpr (M16) compute16(stage); else compute32(stage); };
    // ---- LayerNorm row statistics (threads 0..BM-1, one row each), parked in LDS for the epilogue; called right after the
    // first tile requests so that their loads overlap the tiles' flight
    auto ln_stats = [&]() {
        if (LN) {
            if (tid_all < BM) {
                const int row = min(m0 + tid_all, M - 1);
                float mean, rstd;
                if (p.a_part) {
                    const float* q = p.a_part + (size_t)row * p.a_nparts * 2;
                    float sm = 0.f, m2 = 0.f;
                    if (p.a_nparts == 6) {         // width 384: the row's 12 floats as three 16-byte loads, one round trip
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(q), v1 = *reinterpret_cast<const f32x4*>(q + 4),
                                    v2 = *reinterpret_cast<const f32x4*>(q + 8);
                        const float mk[6] = {v0[0], v0[2], v1[0], v1[2], v2[0], v2[2]};
                        sm = ((mk[0] + mk[1]) + (mk[2] + mk[3])) + (mk[4] + mk[5]);
                        m2 = ((v0[1] + v0[3]) + (v1[1] + v1[3])) + (v2[1] + v2[3]);
                        mean = sm / 6.0f;
#pragma unroll
                        for (int k = 0; k < 6; ++k) { const float d = mk[k] - mean; m2 += 64.0f * (d * d); }
                    } else {
                        for (int k = 0; k < p.a_nparts; ++k) { sm += q[2 * k]; m2 += q[2 * k + 1]; }
                        mean = sm / (float)p.a_nparts;
                        for (int k = 0; k < p.a_nparts; ++k) { const float d = q[2 * k] - mean; m2 += 64.0f * (d * d); }
                    }
                    rstd = 1.0f / sqrtf(m2 / (64.0f * (float)p.a_nparts) + p.a_eps);
                } else {
                    mean = p.a_mean[row];
                    rstd = p.a_rstd[row];
                }
                srow[tid_all] = mean;
                srow[BM + tid_all] = rstd;
            }
        }
    };
    // ---- Block1D tail (gnr_*): wave 0 merges the producing conv's tile entries for the <= 4 groups under this workgroup's 128
    // columns (16 lanes per group, Chan merge, fixed order) into gstat = [mean x 4 | rstd x 4], parked where srow would be.
    // A workgroup's rows lie in at most two utterances (T_out >= BM): wave 0 merges for the first, wave 1 for the second.
    auto gnr_prologue = [&]() {
        if (p.gnr_y && tid_all < 128) {
            const int cpg = p.gnr_cpg, gl = (tid & 63) >> 4, j = tid & 15, g = fdiv(n0, cpg, p.rcp_gnr_cpg) + gl;
            const int b = fdiv(m0, p.T_out, p.rcp_T_out) + (tid >> 6);
            const bool live = g < p.gnr_groups && g * cpg < min(p.N, n0 + GEMM_BN) && b < p.B;
            const int ncw = p.N >> 6, R = p.gnr_tile_rows;
            const int t_first = fdiv(b * p.T_out, R, p.rcp_gnr_R), nrw = fdiv((b + 1) * p.T_out - 1, R, p.rcp_gnr_R) - t_first + 1;     // wave tiles touching utterance b
            float n = 0.f, mean = 0.f, m2 = 0.f;
            // (the closed-form bias rows' operands are requested with the entries: one round trip, not two)
            float x_ne = 0.f, x_bm = 0.f, x_bq = 0.f;
            if (live && p.gnr_nextra && j == 0) {
                x_ne = (float)p.gnr_nextra[b];
                x_bm = p.gnr_bias_stats[2 * g];
                x_bq = p.gnr_bias_stats[2 * g + 1];
            }
            if (live) {
                const int w_lo = (g * cpg) >> 6, w_hi = ((g + 1) * cpg - 1) >> 6, nw = w_hi - w_lo + 1;
                // entries in batches of 8 per lane: the loads of a batch are independent (one round trip), the merge order is fixed
                const int total = nrw * nw;
                for (int k0 = j; k0 < total; k0 += 16 * 8) {
                    f32x4 q[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = k0 + 16 * e;
                        const int kc = k < total ? k : j;
                        const int rw = nw == 1 ? kc : (nw == 2 ? kc >> 1 : kc / nw), w = w_lo + (kc - rw * nw), tile = t_first + rw;
                        const int part = tile * R >= b * p.T_out ? 0 : 1;       // (a tile that starts in the previous utterance holds b's rows as its part 1)
                        q[e] = *reinterpret_cast<const f32x4*>(p.gnr_stats + ((size_t)(((size_t)tile * 2 + part) * ncw + w) * 2 + (g - fdiv(w * 64, cpg, p.rcp_gnr_cpg))) * 4);
                        if (k >= total) q[e][0] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float nb = q[e][0];
                        if (nb <= 0.f) continue;
                        const float delta = q[e][1] - mean, nt = n + nb, r = nb / nt;      // one division per entry
                        mean += delta * r;
                        m2 += q[e][2] + delta * delta * (n * r);
                        n = nt;
                    }
                }
            }
            for (int off = 1; off < 16; off <<= 1) {
                const float n2 = __shfl_xor(n, off), mean2 = __shfl_xor(mean, off), m22 = __shfl_xor(m2, off);
                const float nt = n + n2;
                if (nt > 0.f) {       // both lanes of a pair form the same mean (the lower lane's expression); one division each
                    const bool up = (j & off) != 0;
                    const float delta = mean2 - mean, wsel = (up ? n : n2) / nt;
                    const float merged = up ? mean2 + (mean - mean2) * wsel : mean + delta * wsel;
                    m2 = m2 + m22 + delta * delta * ((up ? n2 : n) * wsel);
                    mean = merged;
                    n = nt;
                }
            }
            float* gs = srow + (tid >> 6) * 8;
            if (j == 0) {
                if (live && p.gnr_nextra) {            // folded padding: nextra copies of the conv's bias row (closed form)
                    const float ne = x_ne;
                    if (ne > 0.f) {
                        const float nb = ne * (float)cpg, delta = x_bm - mean, nt = n + nb, r = nb / nt;
                        mean += delta * r;
                        m2 += ne * x_bq + delta * delta * (n * r);
                        n = nt;
                    }
                }
                gs[gl] = mean;
                gs[4 + gl] = n > 0.f ? 1.0f / sqrtf(m2 / n + p.gnr_eps) : 0.f;
            }
        }
    };
    MTTS_STAMP(0);
    MTTS_STAMP_RT(4);
    // epilogue operands whose round trip should hide under the k-loop: the residual image tile (64-row tiles) and, where the
    // register budget allows (64-row tiles again), the column constants; requested BEFORE the first tiles so that they are the
    // oldest entries of the vector-memory counter and the ring's counted waits stay exact
    // split-K epilogue: without GroupNorm statistics both wave sets finish the tile, half of every wave tile's rows each (two
    // waves per SIMD keep the vector ALU issuing every cycle pair); with them (per whole wave tile) the first set does it alone
    constexpr bool EPI_SPLIT = KS > 1 && (!GN || BM == 128);     // (128-row kernels: a half wave tile is the 32 rows the statistics' consumers expect)
    constexpr int EPI_PASS = EPI_SPLIT ? BM / 16 / KS : BM / 16;
    const int epi_row0 = EPI_SPLIT ? ks * (BM / 2 / KS) : 0;
    const bool epi_wave = EPI_SPLIT || ks == 0;               // does this wave run an epilogue
    EpiPre<BM> pre;
    pre.valid = false;
#ifndef MTTS_EPI_PRE
#define MTTS_EPI_PRE 2
#endif
#if MTTS_EPI_PRE == 1
    if (epi_wave) epi_prefetch<BM, EPI_PASS>(p, pre, M, m0, n0, wm, wn, lane, epi_row0);
#endif
    EpiCols cols;
    if constexpr (BM == 64) {
        if (epi_wave) {
            cols = epi_load_cols<LN>(p, n0, wn, lane);
            epi_prefetch_masks<BM, EPI_PASS>(p, pre, M, m0, wm, lane, epi_row0);
        }
    }
    EpiGnRows gn_rows = {BM, 0, 0};
    if constexpr (GN) { if (epi_wave) gn_rows = epi_gn_rows(p, M, m0 + wm * (BM / 2) + epi_row0, EPI_PASS * 8); }
    if constexpr (KS > 1) {                     // this set's first k-step: walk the runs (tap, segment) up to step ks * nk
        int skip = ks * nk;
        for (;;) {
            const int len = (run_seg == 1 ? p.c1 : p.c0) / KSTEP;
            if (skip < len) break;
            skip -= len;
            if (run_seg == 0 && p.a16_1 != nullptr) run_seg = 1;
            else { run_seg = 0; ++run_tap; }
        }
        setup_run();
#pragma unroll
        for (int j = 0; j < APW; ++j) asrc[j] += skip * astep[j];
        run_left -= skip;
#pragma unroll
        for (int j = 0; j < 4; ++j) wsrc[j] += (size_t)(ks * nk) * 64;
    } else {
        setup_run();
    }
    if constexpr (NST == 2) {
        issue(0);
#if MTTS_EPI_PRE == 2
        if (epi_wave) epi_prefetch<BM, EPI_PASS>(p, pre, M, m0, n0, wm, wn, lane, epi_row0);
#endif
        ln_stats();
        gnr_prologue();
        __syncthreads();                       // (emits vmcnt(0): the first tile has landed)
        MTTS_STAMP(1);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) issue(buf ^ 1);   // the other buffer was last read before the barrier that ended step kt-1
            compute(lds_k + buf * STAGE);
            __syncthreads();                   // tile kt+1 landed (vmcnt(0)) and everyone is done reading tile kt
        }
    } else {
        constexpr int D = NST - 1;             // tiles in flight: the one about to be computed + D-1 behind it
        constexpr int PER_TILE = APW + 4;      // DMA instructions per tile and wave (the only VMEM ops in the loop)
#ifdef MTTS_KSTAMP_SETUP
        MTTS_STAMP(7);                         // (diagnostic: setup done, first tile about to be requested)
#endif
        for (int t = 0; t < D && t < nk; ++t) issue(t);
#if MTTS_EPI_PRE == 2
        // the residual image tile of a 64-row tile: requested right behind the first D tiles; its R loads sit in the
        // vector-memory counter between tile D-1 and tile D, so the first D counted waits allow R more (loads retire in order).
        // The counts depend on that ORDER: the empty asm statements keep the compiler from moving these plain loads across the
        // tile requests on either side (they do not alias the LDS writes, so it otherwise may).
        asm volatile("" ::: "memory");
        if (epi_wave) epi_prefetch<BM, EPI_PASS>(p, pre, M, m0, n0, wm, wn, lane, epi_row0);
        asm volatile("" ::: "memory");
#endif
        const int pre_r = (BM == 64 && MTTS_EPI_PRE == 2 && pre.valid) ? (p.half16 ? EPI_PASS : 2 * EPI_PASS) : 0;
        ln_stats();
        gnr_prologue();
        int st = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // Tile kt has landed for this wave once at most the D-1 younger tiles are outstanding (the ring's tail just
            // drains); lgkmcnt(0): this wave's fragment reads of tile kt-1 are complete, so after the barrier that tile's
            // stage may be refilled.  Raw s_barrier: __syncthreads() would wait for every DMA in flight.
            if (kt + D - 1 < nk) {
                if (kt < D && pre_r == 8) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((D - 1) * PER_TILE + 8) : "memory");
                else if (kt < D && pre_r == 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((D - 1) * PER_TILE + 4) : "memory");
                else if (kt < D && pre_r == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((D - 1) * PER_TILE + 2) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((D - 1) * PER_TILE) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt == 0) MTTS_STAMP(1);
            // (split-K: letting the second wave set compute first and request afterwards, so that one set's MFMAs run under the
            // other's DMA issue, measured SLOWER than the two sets in lockstep: 26.58 vs 26.16 ms per step, r02)
            if (kt + D < nk) issue(st + D >= NST ? st + D - NST : st + D);
            compute(lds_k + st * STAGE);
            st = st + 1 == NST ? 0 : st + 1;
        }
        __syncthreads();                       // the epilogue tile overlays the stages: everyone is done reading
    }

    MTTS_STAMP(2);
    if constexpr (BM != 64) { if (epi_wave) cols = epi_load_cols<LN>(p, n0, wn, lane); }      // (their flight overlaps the parking below)
    // ---- epilogue: park the wave's tile in LDS, re-read it as rows of 2 x float4 (8 lanes per row)
    float* Cw = reinterpret_cast<float*>(lds) + (ks * 4 + wave) * ((BM / 2) * P16_CS);
    if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)    // D of 16x16x32: lane (col = lane&15, row block lane>>4), register r = row 4 (lane>>4) + r
                Cw[(i * 16 + 4 * fq + r) * P16_CS + j * 16 + fr] = acc[i][j][r] + accx[i][j][r] * (1.0f / F16_RES_SCALE);
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cw[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh32) * P16_CS + j * 32 + fr32] = acc32[i][j][r] + accx32[i][j][r] * (1.0f / F16_RES_SCALE);
    }
    const float* Cw2 = nullptr;            // the other set's partial tile (KS = 2)
    if constexpr (KS > 1) {
        __syncthreads();                   // every set's partial tile is parked
        if (!epi_wave) return;
        Cw = reinterpret_cast<float*>(lds) + wave * ((BM / 2) * P16_CS);            // partial sums in set order: run-to-run identical
        Cw2 = Cw + 4 * ((BM / 2) * P16_CS);
    } else {
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the tile is private to this wave
        __builtin_amdgcn_wave_barrier();
    }
    MTTS_STAMP(6);

    gemm_epilogue_rows8<BM, LN, GN, EPI_PASS>(p, cols, pre, gn_rows, Cw, Cw2, srow, M, m0, n0, wm, wn, lane, srow, epi_row0);    // (gstat shares srow's slot: never both)
    MTTS_STAMP(3);
    MTTS_STAMP_RT(5);
}

// MFMA shape: 16x16x32 wherever a CU holds more than one workgroup (+10 % at B = 32), 32x32x16 on the 4-stage ring (grids of
// at most one workgroup per CU are latency-bound and lose 4 % with the longer 16x16 issue sequence; B <= 8 serving shapes).
template <int BM, bool LN, int NST, int MODE, bool M16, bool GN = false, int KS = 1>
static hipError_t launch_p16_shape(const GemmArgs& a, hipStream_t s) {
    static bool configured = false;   // per instantiation
    auto kern = gemm_p16_kernel<BM, LN, NST, MODE, M16, GN, KS>;
    constexpr int lds_bytes = p16_lds_bytes(BM, NST, KS);
    static_assert(lds_bytes <= 160 * 1024, "LDS per workgroup");
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int M = a.B * a.T_out;
    const int grid = ((M + BM - 1) / BM) * ((a.N + GEMM_BN - 1) / GEMM_BN);
    static const std::string tag = "gemm_p16_kernel<" + std::to_string(BM) + ", " + tf(LN) + ", " + std::to_string(NST) + ", " + std::to_string(MODE) +
                                   ", " + tf(M16) + ", " + tf(GN) + ", " + std::to_string(KS) + ">";
    g_kernel_tag = tag.c_str();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * KS), lds_bytes, s, a);
    return hipGetLastError();
}

// the split-K form of a one-workgroup-per-CU grid: 64-row tiles, two 4-wave sets on 3-stage rings, 16x16x32 MFMAs (two waves
// per SIMD, as with two co-resident workgroups)
template <bool LN, int MODE>
static hipError_t launch_p16_splitk(const GemmArgs& a, hipStream_t s) {
    if constexpr (!LN && MODE != 1) {
        if (a.gn_stats) return launch_p16_shape<64, LN, 3, MODE, true, true, 2>(a, s);
    }
    return launch_p16_shape<64, LN, 3, MODE, true, false, 2>(a, s);
}

template <int BM, bool LN, int NST, int MODE>
static hipError_t launch_p16_one(const GemmArgs& a, hipStream_t s) {
    // MFMA shape by pipeline depth: 16x16x32 wherever a CU holds more than one workgroup, 32x32x16 on the 4-stage ring (see above)
    if constexpr (!LN && MODE != 1) {
        if (a.gn_stats) return launch_p16_shape<BM, LN, NST, MODE, !(BM == 64 && NST == 4), true>(a, s);      // conv feeding a Block1D
    }
    return launch_p16_shape<BM, LN, NST, MODE, !(BM == 64 && NST == 4)>(a, s);
}

template <int BM, bool LN, int NST = 2>
static hipError_t launch_p16_variant(const GemmArgs& a, hipStream_t s) {
    if (a.half16 && a.bf16) return launch_p16_one<BM, LN, NST, 3>(a, s);
    if (a.half16) return launch_p16_one<BM, LN, NST, 2>(a, s);
    return a.fast16 ? launch_p16_one<BM, LN, NST, 1>(a, s) : launch_p16_one<BM, LN, NST, 0>(a, s);
}

// Block-tile height and pipeline depth for a shape (also what gemm_p16_wave_rows reports to callers that must match it).
// Height by grid fill, as launch_gemm: 2 resident workgroups per CU for either height (LDS 70 / 49 KB).  Small grids (serving
// shapes, B <= 8): 64-row tiles on the prefetch ring -- 4 stages at <= 1 workgroup per CU (96 KB of LDS), 3 stages at <= 2
// (72 KB).  MTTS_P16_RING=0 keeps the two-stage kernel, 2 puts every 64-row grid on the ring; MTTS_GEMM_BM forces a height.
static int p16_choose(const GemmArgs& a, int& nst) {
    const int M = a.B * a.T_out;
    const int nt = (a.N + GEMM_BN - 1) / GEMM_BN;
    auto fill = [&](int bm) {
        const int tiles = ((M + bm - 1) / bm) * nt;
        const int rounds = (tiles + 511) / 512;
        return (double)tiles / (rounds * 512.0) * ((double)M / (((M + bm - 1) / bm) * bm));
    };
    static const int ring_mode = [] { const char* e = getenv("MTTS_P16_RING"); return e ? atoi(e) : 1; }();
    static const int env_bm = [] { const char* e = getenv("MTTS_GEMM_BM"); return e ? atoi(e) : 0; }();
    const int tiles64 = ((M + 63) / 64) * nt;
    nst = 2;
    if (ring_mode != 0 && a.force_bm == 0 && tiles64 <= 512) {
        nst = tiles64 <= 256 ? 4 : 3;      // (a 6-stage ring for the one-round grids measured the same: r02)
        return 64;
    }
    const int force = a.force_bm ? a.force_bm : env_bm;
    const bool bm64 = force == 64 || (force == 0 && 0.97 * fill(64) > fill(128));
    if (bm64 && ring_mode == 2) nst = 3;
    return bm64 ? 64 : 128;
}
int gemm_p16_wave_rows(const GemmArgs& a) {
    int nst;
    return p16_choose(a, nst) / 2;
}

static unsigned int rcp32(int d) { return d <= 1 ? 0u : (unsigned int)((1ull << 32) / (unsigned long long)d + 1ull); }

hipError_t launch_gemm_p16(const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    {   // reciprocals for the kernel's index arithmetic (kernels.h); exactness needs n * d < 2^32 for every quotient taken:
        // numerators are rows (< M + 256), columns (< N + 128) or tile counts; a launch beyond that divides (rcp = 0)
        if (a.B <= 0 || a.T_out <= 0 || a.N <= 0) return hipErrorInvalidValue;
        const long long Mp = (long long)a.B * a.T_out + 256, Np = (long long)a.N + 128;
        a.gn_cpg = a.gn_groups > 0 ? a.N / a.gn_groups : 0;
        a.gnr_cpg = a.gnr_groups > 0 ? a.N / a.gnr_groups : 0;
        a.rcp_T_out = Mp * a.T_out < (1ll << 32) ? rcp32(a.T_out) : 0u;
        a.rcp_ntiles = Mp * Np < (1ll << 32) ? rcp32((a.N + GEMM_BN - 1) / GEMM_BN) : 0u;
        a.rcp_gn_cpg = Np * Np < (1ll << 32) ? rcp32(a.gn_cpg) : 0u;
        a.rcp_gnr_cpg = Np * Np < (1ll << 32) ? rcp32(a.gnr_cpg) : 0u;
        a.rcp_gnr_R = Mp * 64 < (1ll << 32) && a.gnr_tile_rows <= 64 ? rcp32(a.gnr_tile_rows) : 0u;
    }
    // shape contract (the kernel indexes without further checks)
    const int kq = a.half16 ? 64 : GEMM_BK, ew = a.half16 ? 1 : 2;      // k elements per 128-byte line; halves per element
    if (!a.a16_0 || (a.half16 ? !a.w16h : !a.w16) || a.terms != 2 || (!a.out && !a.out16) || a.N <= 0 || (a.N & 3) || a.B <= 0 || a.T_out <= 0 || a.T_in <= 0)
        return hipErrorInvalidValue;
    if (a.ntaps < 1 || a.ntaps > MAX_TAPS || a.ktap <= 0 || a.ktap % kq != 0) return hipErrorInvalidValue;
    if ((a.c0 % kq) || (a.c1 % kq) || a.c0 + a.c1 != a.ktap) return hipErrorInvalidValue;   // images are physically padded
    if ((a.a16_1 == nullptr) != (a.c1 == 0)) return hipErrorInvalidValue;
    if (a.lda16_0 < ew * a.c0 || (a.lda16_0 & 7) || (a.a16_1 && (a.lda16_1 < ew * a.c1 || (a.lda16_1 & 7)))) return hipErrorInvalidValue;
    if (a.a_mask) return hipErrorInvalidValue;                       // P16 images are written already masked
    if ((a.a_mean == nullptr) != (a.a_rstd == nullptr)) return hipErrorInvalidValue;
    if (a.a_part && (a.a_mean || a.a_nparts <= 0)) return hipErrorInvalidValue;
    const bool ln = a.a_mean || a.a_part;
    if (ln && (!a.wsum || a.ntaps != 1 || a.in_stride != 1 || a.tap_off[0] != 0 || a.T_in != a.T_out)) return hipErrorInvalidValue;
    if (a.out && (a.ldc & 3)) return hipErrorInvalidValue;
    if (a.res && (a.ldr & 3)) return hipErrorInvalidValue;
    if (a.res16 && (a.res || (a.N % 32) || a.ldr16 < ew * a.N || (a.ldr16 & 3))) return hipErrorInvalidValue;
    if (a.out16 && ((a.N % 32) || a.ld16 < ew * a.N || (a.ld16 & 3))) return hipErrorInvalidValue;
    if (a.stats_out && (a.N & 63)) return hipErrorInvalidValue;
    if (a.act == ACT_SNAKE && (!a.p0 || !a.p1)) return hipErrorInvalidValue;
    if (a.bf16 && !a.half16) return hipErrorInvalidValue;
    if (a.gn_stats && a.fast16) return hipErrorInvalidValue;          // (the opt-in fp16 mode keeps the separate statistics pass)
    if (a.gn_stats) {
        const bool plain = a.out_stride == 1 && a.out_off == 0 && a.out_T == a.T_out;
        if (a.gn_groups <= 0 || (a.N % a.gn_groups) || (a.N / a.gn_groups) < 32 || ((a.N / a.gn_groups) & 7) || (a.N & 63) || !plain ||
            a.T_out < gemm_p16_wave_rows(a) || a.act != ACT_NONE || a.res || a.res16 || a.out_mask || a.out_scale != 1.0f || ln)
            return hipErrorInvalidValue;
    }
    int nst = 2;
    const int bm = p16_choose(a, nst);
#ifdef MTTS_KSTAMP
    if (g_kstamp && g_kstamp_skip > 0) {
        --g_kstamp_skip;
    } else if (g_kstamp) {                                   // one launch per set call
        GemmArgs b = a;
        b.kstamp = g_kstamp;
        g_kstamp = nullptr;
        const int nk_all = a.ntaps * a.ktap / kq;
        const bool sk = bm == 64 && nst == 4 && nk_all >= 4 && (nk_all % 2) == 0 && !(getenv("MTTS_P16_SPLITK") && atoi(getenv("MTTS_P16_SPLITK")) == 0);
        const int info[8] = {a.B * a.T_out, a.N, a.ntaps * a.ktap, bm, sk ? 3 : nst, sk ? 2 : 1, a.ntaps,
                             (ln ? 1 : 0) | (a.res16 ? 2 : 0) | (a.out16 ? 4 : 0) | (a.out ? 8 : 0) | (a.gn_stats ? 16 : 0) | (a.gnr_y ? 32 : 0) |
                                 (a.act == ACT_SNAKE ? 64 : 0) | (a.stats_out ? 128 : 0)};
        for (int i = 0; i < 8; ++i) g_kstamp_info[i] = info[i];
        return launch_gemm_p16(b, s);
    }
#endif
    if (a.gnr_y) {
        const bool plain = a.out_stride == 1 && a.out_off == 0 && a.out_T == a.T_out;
        if (!a.gnr_stats || !a.gnr_gamma || !a.gnr_beta || !a.gnr_mask || a.gnr_groups <= 0 || a.gnr_tile_rows <= 0 || (a.N % a.gnr_groups) ||
            (a.N / a.gnr_groups) < 32 || ((a.N / a.gnr_groups) & 7) || (a.N & 63) || !plain || a.T_out < bm || a.T_out < a.gnr_tile_rows || ln ||
            ((a.gnr_nextra != nullptr) != (a.gnr_bias_stats != nullptr)) ||
            a.act != ACT_NONE || a.res || a.res16 || a.out_mask || a.out_scale != 1.0f)
            return hipErrorInvalidValue;
    }
    if (bm == 64) {
        // one workgroup per CU or fewer: split the K axis between two wave sets when it divides (MTTS_P16_SPLITK=0: the 4-stage ring)
        static const int splitk = [] { const char* e = getenv("MTTS_P16_SPLITK"); return e ? atoi(e) : 1; }();
        const int nk_all = a.ntaps * a.ktap / kq;
        if (nst == 4 && splitk && nk_all >= 4 && (nk_all % 2) == 0) {
            if (a.half16 && a.bf16) return ln ? launch_p16_splitk<true, 3>(a, s) : launch_p16_splitk<false, 3>(a, s);
            if (a.half16) return ln ? launch_p16_splitk<true, 2>(a, s) : launch_p16_splitk<false, 2>(a, s);
            if (a.fast16) return ln ? launch_p16_splitk<true, 1>(a, s) : launch_p16_splitk<false, 1>(a, s);
            return ln ? launch_p16_splitk<true, 0>(a, s) : launch_p16_splitk<false, 0>(a, s);
        }
        // (the same split on 128 x 128 tiles -- 64 x 64 per wave, one 8-wave workgroup per CU -- for the 257..512-tile grids measured
        // slower than two co-resident 64-row workgroups: 27.26 vs 26.55 ms per step, r02)
        if (nst == 4) return ln ? launch_p16_variant<64, true, 4>(a, s) : launch_p16_variant<64, false, 4>(a, s);
        if (nst == 3) return ln ? launch_p16_variant<64, true, 3>(a, s) : launch_p16_variant<64, false, 3>(a, s);
        return ln ? launch_p16_variant<64, true>(a, s) : launch_p16_variant<64, false>(a, s);
    }
    return ln ? launch_p16_variant<128, true>(a, s) : launch_p16_variant<128, false>(a, s);
}

// ------------------------------------------------------------------------------------------------ fp32 <-> P16
__global__ void to_p16_kernel(const float* __restrict__ x, int ld, const float* __restrict__ mask, int M, int C, int C_valid,
                              _Float16* __restrict__ out, int ld16, float lscale, unsigned int* range_flag, bool half16, bool bf16) {
    const int c4n = C >> 2;
    const size_t n = (size_t)M * c4n;
    bool range_bad = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / c4n), c = (int)(i - (size_t)row * c4n) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < C_valid) v = *reinterpret_cast<const f32x4*>(x + (size_t)row * ld + c);     // C_valid % 4 == 0
        if (mask) v *= mask[row];
        range_bad |= out_of_f16_range(v[0], v[1], v[2], v[3]);
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 a, b;
            split_f16(v[e], lscale, a, b);
            h[e] = a;
            l[e] = b;
        }
        if (half16 && bf16) {
            using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
            *reinterpret_cast<u32x2*>(out + (size_t)row * ld16 + c) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        } else if (half16) {
            *reinterpret_cast<f16x4*>(out + (size_t)row * ld16 + c) = h;
        } else {
            _Float16* o = out + (size_t)row * ld16 + (c >> 5) * 64 + (c & 31);
            *reinterpret_cast<f16x4*>(o) = h;
            *reinterpret_cast<f16x4*>(o + 32) = l;
        }
    }
    raise_range_flag(range_flag, range_bad && !bf16);
}
hipError_t launch_to_p16(const float* x, int ld, const float* mask, int M, int C, int C_valid, _Float16* out, int ld16, float lscale,
                         hipStream_t s, unsigned int* range_flag, bool half16, bool bf16) {
    if (!x || !out || M <= 0 || C <= 0 || (C % 32) || (ld & 3) || C_valid > C || (C_valid & 3) || ld < C_valid || ld16 < (half16 ? 1 : 2) * C || (ld16 & 3))
        return hipErrorInvalidValue;
    const size_t n = (size_t)M * (C >> 2);
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(to_p16_kernel, dim3(grid), dim3(256), 0, s, x, ld, mask, M, C, C_valid, out, ld16, lscale, range_flag, half16, bf16);
    return hipGetLastError();
}
__global__ void from_p16_kernel(const _Float16* __restrict__ x, int ld16, int M, int C, float inv_lscale, float* __restrict__ out, int ld) {
    const size_t n = (size_t)M * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / C), c = (int)(i - (size_t)row * C);
        const _Float16* q = x + (size_t)row * ld16 + (c >> 5) * 64 + (c & 31);
        out[(size_t)row * ld + c] = (float)q[0] + (float)q[32] * inv_lscale;
    }
}
hipError_t launch_from_p16(const _Float16* x, int ld16, int M, int C, float lscale, float* out, int ld, hipStream_t s) {
    if (!x || !out || M <= 0 || C <= 0 || (C % 32) || ld < C || ld16 < 2 * C || lscale == 0.f) return hipErrorInvalidValue;
    const size_t n = (size_t)M * C;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(from_p16_kernel, dim3(grid), dim3(256), 0, s, x, ld16, M, C, 1.0f / lscale, out, ld);
    return hipGetLastError();
}

}  // namespace mtts
