// Transformer-block chain for gfx950: out-projection + residual -> LayerNorm -> FeedForward (Linear, SnakeBeta, Linear) + residual
// -> LayerNorm -> the NEXT block's q|k|v projection, as ONE launch (kernels.h ChainArgs).
//
// Replaces, per BasicTransformerBlock of the reference decoder (transformer.py:261 to_out + residual, :278-301 norm3 / ff /
// residual, :249-258 norm1 + to_q/k/v of the following block; FeedForward and SnakeBeta transformer.py:104-120,61-77), the four
// launches gemm_p16 (out-projection), gemm_p16 (FF1 + SnakeBeta), gemm_p16 (FF2) and gemm_p16 (q|k|v) of the unfused path.
//
// Why this shape.  Every k-step of a 2-D tiled GEMM fetches an activation tile AND a weight tile through the CU's vector
// memory path (~40 B/clk/CU in practice), each launch pays ~3 us until its first tile has landed and ~2.5 us of epilogue, and
// every activation makes a round trip through HBM as a 4-byte image.  Behind the attention the block is
// row-local, so a workgroup can own QB rows for the whole chain:
//   * the residual-stream tile x [QB x C] lives in LDS as a P16 image (row = 128-byte lines of 32 heads | 32 residuals per
//     32-channel group, XOR-swizzled like gemm_p16.hip's stages) and is the stationary MFMA operand of FF1 and q|k|v;
//   * the hidden layer exists only as one [QB x CH] chunk in LDS: FF1 of a chunk, SnakeBeta, split, then straight into FF2's
//     accumulators (QB x C, in registers for the whole FeedForward);
//   * the weights never touch LDS.  Each wave owns C/8 output channels (CH/8 hidden channels in FF1) and reads "its" panel rows as
//     a fragment stream: 1 KiB fragments (16 rows x 32 k of one fp16 plane, lane-major) in consumption order, one
//     global_load_dwordx4 per lane each, through a register ring of R fragments (~12 KiB per wave in flight: what a ~2k-cycle
//     L2 round trip needs at the CU's ingest rate).  The loads are inline asm with hand-written waits (see CH_LOAD below: hipcc
//     sinks plain loads next to their use), and every k-loop ENDS with a drain of the ring: no asm load is in flight while
//     compiler-scheduled code (an epilogue, LayerNorm moments) runs, because the compiler may copy such a register before the
//     load has landed;
//   * products are computed TRANSPOSED, D^T[channel][row] = W . X^T (A = weight fragment, B = activation fragment): a lane then
//     holds 4 consecutive channels of one row, so every LDS / global store of an epilogue is 8 bytes of one image line;
//   * LayerNorm statistics are taken from the LDS tile (one wave per row, two passes), applied after the product as
//     rstd (x.W'^T - mean rowsum(W')) + b' exactly as gemm_p16.hip does.
//   * what bounds it is the CU's line-fill rate (~37 B/clk): every workgroup pulls the block's whole fragment stream (7 MB) through
//     its CU.  Two answers: prefetch workgroups -- one per XCD does nothing but touch the stream just ahead of
//     the others, so their loads hit the L2 -- and, where the rows of a level do not fill the chip with one workgroup per tile, the
//     PAIR form: two workgroups of one XCD share a tile, each streams half of the FeedForward and of the q|k|v passes, and the two
//     FF2 partial sums meet through global memory behind an agent-scope release / acquire.
// Arithmetic = gemm_p16.hip MODE 0: x = h + l / 2^11, products h.h + (h.l + l.h) / 2^11, fp32 accumulation.
#include "kernels.h"
#include "device_utils.h"
#include <cstring>
#include <string>
#include <cmath>

namespace mtts {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

constexpr int CHAIN_NW = CHAIN_WAVES;          // waves per workgroup

template <int C, int QB, int CH>
struct ChainCfg {
    static constexpr int NT = C / (16 * CHAIN_NW);       // 16-channel tiles of the C-wide outputs per wave
    static constexpr int NT1 = CH / (16 * CHAIN_NW);     // 16-channel tiles of a hidden chunk per wave
    static constexpr int MT = QB / 16;                   // 16-row tiles
    static constexpr int KG = C / 32, KG2 = CH / 32;     // k-steps over the stream width / over a hidden chunk
    static constexpr int NCH = 4 * C / CH;               // hidden chunks
    // register ring (fragments): ~12 KiB per wave in flight.  (A 24-deep ring for 32-row workgroups spilled ring registers in the
    // out-projection loop -- and a spill of an asm-loaded register stores it BEFORE its wait, i.e. garbage: tests/test_isa_guard.py
    // checks that no loop of this kernel touches scratch.)
    static constexpr int R = C == 384 ? 12 : 8;
    static constexpr int XT_BYTES = QB * C * 4, HT_BYTES = QB * CH * 4;
    static constexpr int CT_FLOATS = 18 * C;            // column constants kept in LDS: wsum1 | b1 | p0 | p1/2 (4C each) | b_out | b2 (C each)
    static constexpr int LDS_BYTES = XT_BYTES + HT_BYTES + 2 * QB * 4 + CT_FLOATS * 4;
    static_assert(C % 128 == 0 && CH % 128 == 0 && QB % 16 == 0 && QB <= 64, "shape");
    static_assert(R % (2 * NT) == 0 && R % (2 * NT1) == 0, "a ring period is a whole number of steps");
    static_assert((KG * 2 * NT1 + KG2 * 2 * NT) % R == 0 && (KG * 2 * NT) % R == 0, "phases start on ring slot 0");
    static_assert(((R / (2 * NT)) & 1) == 0, "out-projection: even number of steps per ring period");
};

// ------------------------------------------------------------------------------------------------ host: fragment streams
// Per wave: [out-projection: inner/32 steps x NT tiles][per hidden chunk: C/32 steps x NT1 tiles (FF1), CH/32 steps x NT tiles
// (FF2)][q|k|v: passes x C/32 steps x NT tiles][R padding fragments]; a tile = fragment of the head plane, then of the residual plane.
static int chain_ring(int C) { return C == 384 ? 24 : 8; }          // tail padding: the deepest ring any kernel shape uses
static int chain_qkv_passes(int C, int n_qkv) {
    const int per_pass = CHAIN_NW * (C / 128);          // 16-channel tiles per pass
    return n_qkv > 0 ? ((n_qkv / 16) + per_pass - 1) / per_pass : 0;
}
long chain_stream_frags(int C, int inner, int ch, int n_qkv) {
    const int NT = C / 128, NT1 = ch / 128;
    long f = (long)(inner / 32) * 2 * NT;
    f += (long)(4 * C / ch) * ((C / 32) * 2 * NT1 + (ch / 32) * 2 * NT);
    f += (long)chain_qkv_passes(C, n_qkv) * (C / 32) * 2 * NT;
    return f + chain_ring(C);
}
bool chain_supported(int C, int inner, int n_qkv) {
    if (C != 128 && C != 256 && C != 384) return false;
    if (inner < 0 || (inner % 32) || inner > C) return false;
    const int NT = C / 128, R = chain_ring(C);
    if (inner && ((inner / 32) * 2 * NT) % R) return false;          // the out-projection ends on ring slot 0 (of every ring depth)
    if (n_qkv && ((n_qkv % 32) || !inner)) return false;
    return true;
}
static void put_frag(uint16_t* dst, const float* w, int ldw, int n0, int n_valid, int k0, bool* sat) {
    // dst: [plane][64 lanes][8]; lane (r = lane & 15, q = lane >> 4) holds row n0 + r, k = k0 + 8 q .. + 7
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + (lane & 15), k = k0 + 8 * (lane >> 4) + j;
            const float x = n < n_valid ? w[(size_t)n * ldw + k] : 0.f;
            if (std::fabs(x) > 65504.f && sat) *sat = true;
            const float xc = x < -65504.f ? -65504.f : (x > 65504.f ? 65504.f : x);
            const _Float16 h = (_Float16)xc;
            float r = (x - (float)h) * F16_RES_SCALE;
            r = r < -65504.f ? -65504.f : (r > 65504.f ? 65504.f : r);
            const _Float16 l = (_Float16)r;
            std::memcpy(&dst[lane * 8 + j], &h, 2);
            std::memcpy(&dst[512 + lane * 8 + j], &l, 2);
        }
}
void chain_stream_pack(int C, int inner, int ch, int n_qkv, const float* w_out, const float* w1, const float* w2, const float* w_qkv,
                       uint16_t* dst, bool* saturates) {
    const int NT = C / 128, NT1 = ch / 128, KG = C / 32, KG2 = ch / 32, NCH = 4 * C / ch;
    const long per_wave = chain_stream_frags(C, inner, ch, n_qkv);
    const int passes = chain_qkv_passes(C, n_qkv);
    for (int w = 0; w < CHAIN_NW; ++w) {
        uint16_t* o = dst + (size_t)w * per_wave * 512;
        if (inner && w_out)
            for (int s = 0; s < inner / 32; ++s)
                for (int t = 0; t < NT; ++t, o += 1024) put_frag(o, w_out, inner, 16 * (w * NT + t), C, 32 * s, saturates);
        for (int j = 0; j < NCH; ++j) {
            for (int s = 0; s < KG; ++s)
                for (int t = 0; t < NT1; ++t, o += 1024) put_frag(o, w1, C, j * ch + 16 * (w * NT1 + t), 4 * C, 32 * s, saturates);
            for (int s = 0; s < KG2; ++s)
                for (int t = 0; t < NT; ++t, o += 1024) put_frag(o, w2, 4 * C, 16 * (w * NT + t), C, j * ch + 32 * s, saturates);
        }
        for (int ps = 0; ps < passes; ++ps)
            for (int s = 0; s < KG; ++s)
                for (int t = 0; t < NT; ++t, o += 1024)
                    put_frag(o, w_qkv, C, 16 * (ps * CHAIN_NW * NT + w * NT + t), n_qkv, 32 * s, saturates);
        std::memset(o, 0, (size_t)chain_ring(C) * 512 * sizeof(uint16_t));      // (a fragment = 512 halves)
    }
}

// Pair form (ChainArgs::pair): TWO workgroups on one XCD share a row tile.  Both run the out-projection; half h takes the hidden
// chunks [h NCH/2, (h+1) NCH/2) of the FeedForward and -- after the two FF2 partial sums have been exchanged -- the q|k|v passes
// [0, ceil(P/2)) or [ceil(P/2), P).  Streams: [half][wave][out-projection | this half's chunks | this half's passes | R padding],
// every (half, wave) stream padded to the same length (chain_stream_frags_pair).
long chain_stream_frags_pair(int C, int inner, int ch, int n_qkv) {
    const int NT = C / 128, NT1 = ch / 128, nch = 4 * C / ch, P = chain_qkv_passes(C, n_qkv);
    long f = (long)(inner / 32) * 2 * NT;
    f += (long)(nch / 2) * ((C / 32) * 2 * NT1 + (ch / 32) * 2 * NT);
    f += (long)((P + 1) / 2) * (C / 32) * 2 * NT;
    return f + chain_ring(C);
}
bool chain_supported_pair(int C, int inner, int ch, int n_qkv) {
    return chain_supported(C, inner, n_qkv) && inner > 0 && ((4 * C / ch) % 2) == 0;
}
void chain_stream_pack_pair(int C, int inner, int ch, int n_qkv, const float* w_out, const float* w1, const float* w2, const float* w_qkv,
                            uint16_t* dst, bool* saturates) {
    const int NT = C / 128, NT1 = ch / 128, KG = C / 32, KG2 = ch / 32, NCH = 4 * C / ch;
    const long per_wave = chain_stream_frags_pair(C, inner, ch, n_qkv);
    const int passes = chain_qkv_passes(C, n_qkv), p_half = (passes + 1) / 2;
    std::memset(dst, 0, (size_t)2 * CHAIN_NW * per_wave * 512 * sizeof(uint16_t));
    for (int h = 0; h < 2; ++h)
        for (int w = 0; w < CHAIN_NW; ++w) {
            uint16_t* o = dst + (size_t)(h * CHAIN_NW + w) * per_wave * 512;
            for (int s = 0; s < inner / 32; ++s)
                for (int t = 0; t < NT; ++t, o += 1024) put_frag(o, w_out, inner, 16 * (w * NT + t), C, 32 * s, saturates);
            for (int j = h * (NCH / 2); j < (h + 1) * (NCH / 2); ++j) {
                for (int s = 0; s < KG; ++s)
                    for (int t = 0; t < NT1; ++t, o += 1024) put_frag(o, w1, C, j * ch + 16 * (w * NT1 + t), 4 * C, 32 * s, saturates);
                for (int s = 0; s < KG2; ++s)
                    for (int t = 0; t < NT; ++t, o += 1024) put_frag(o, w2, 4 * C, 16 * (w * NT + t), C, j * ch + 32 * s, saturates);
            }
            for (int ps = h ? p_half : 0; ps < (h ? passes : p_half); ++ps)
                for (int s = 0; s < KG; ++s)
                    for (int t = 0; t < NT; ++t, o += 1024)
                        put_frag(o, w_qkv, C, 16 * (ps * CHAIN_NW * NT + w * NT + t), n_qkv, 32 * s, saturates);
        }
}

// ------------------------------------------------------------------------------------------------ device
// diagnostic builds (-DMTTS_CHAIN_STAMP, tools/chain_sweep.py --stamps): s_memtime of wave 0 of workgroup 0 at the phase boundaries
#ifdef MTTS_CHAIN_STAMP
#define CH_STAMP(i) do { if (p.kstamp && tid == 0 && wg >= 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (wg == 0) p.kstamp[i] = t_; \
        if ((i) == 0) p.kstamp[16 + 2 * wg] = t_; if ((i) == 12) p.kstamp[17 + 2 * wg] = t_; \
        if (wg == 0 && ((i) == 0 || (i) == 12)) p.kstamp[16 + 2 * ((M + QB - 1) / QB + p.pf_wgs) + ((i) == 12)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define CH_STAMP(i) do { } while (0)
#endif
#ifdef MTTS_CHAIN_STAMP
#define CH_XSTAMP(k) do { if (p.kstamp && tid == 0 && wg == 0) p.kstamp[4000 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)   /* inside the pair exchange */
#else
#define CH_XSTAMP(k) do { } while (0)
#endif
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// Vector-memory loads whose PLACE in the instruction stream matters (the weight ring, the attention rows, the column constants)
// are inline asm: hipcc otherwise sinks a load towards its use when registers are tight, which turns the ring into one exposed
// L2 round trip per k-step (seen in the first build of this kernel: s_waitcnt vmcnt(0) in front of every MFMA group).  The
// compiler does not count asm loads, so their waits are written here, by the rule
//     a load is complete once at most N vector-memory operations YOUNGER than it are outstanding  (they retire in order),
// with N = the number of younger loads ISSUED BY THIS FILE's asm.  Anything else in the queue (the compiler's own loads and
// stores) only makes the true count larger, i.e. the wait stricter than needed -- never too weak.  The registers a wait covers
// are passed through an empty asm right behind it (CH_TIE): their consumers then depend on the wait and cannot be scheduled
// above it (the MFMA-hoisting hazard of cdna_hip_programming.md rule 18).
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
#define CH_LOAD(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase))
#define CH_LOAD2(d0, d1, voff, sbase) \
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024" : "=&v"(d0), "=&v"(d1) : "v"(voff), "s"(sbase))
#define CH_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n))
#define CH_TIE(x) asm volatile("" : "+v"(x))

// the fp16 split of 4 consecutive channels -> packed head / residual words (v_cvt_pk_f16_f32; the residual of the clamped value)
__device__ __forceinline__ void split4(const f32x4 v, float lscale, u32x2& hw, u32x2& lw, float& rmax) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float x0 = v[2 * e], x1 = v[2 * e + 1];
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(rmax) : "v"(x0), "v"(x1));
        const float c0 = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f), c1 = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
        const f16x2 hp = {(_Float16)c0, (_Float16)c1};
        const f16x2 lp = {(_Float16)((c0 - (float)hp[0]) * lscale), (_Float16)((c1 - (float)hp[1]) * lscale)};
        hw[e] = __builtin_bit_cast(unsigned int, hp);
        lw[e] = __builtin_bit_cast(unsigned int, lp);
    }
}

template <int C, int QB, int CH>
__global__ __launch_bounds__(64 * CHAIN_NW, 1) void tblock_chain_kernel(const ChainArgs p) {
    using K = ChainCfg<C, QB, CH>;
    constexpr int NT = K::NT, NT1 = K::NT1, MT = K::MT, KG = K::KG, KG2 = K::KG2, R = K::R;
    constexpr int FW = 2 * NT, F1S = 2 * NT1;           // fragments per k-step of a C-wide product / of FF1
    extern __shared__ __attribute__((aligned(16))) char lds[];      // ONE array: x tile | hidden chunk / attention stages | row statistics
    char* const XT = lds;
    char* const HT = lds + K::XT_BYTES;
    float* const srow = reinterpret_cast<float*>(lds + K::XT_BYTES + K::HT_BYTES);
    float* const CT = srow + 2 * QB;                      // column constants (see ChainCfg::CT_FLOATS)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4, swz = (c >> 1) & 7;     // fragment coordinates: row / channel c, k block q
    const int wg = (int)blockIdx.x - p.pf_wgs;                      // the first pf_wgs workgroups only prefetch (below)
    // pair form: workgroups wg and wg + 8 (the same XCD) share row tile rtile; half = which hidden chunks / q|k|v passes it takes
    const int pair = p.pair;
    const int half = pair ? ((wg >> 3) & 1) : 0;
    const int rtile = pair ? (((wg >> 4) << 3) | (wg & 7)) : wg;
    const int M = p.M, m0 = rtile * QB;
    const bool has_out = p.inner > 0, has_qkv = p.b_qkv != nullptr;
    const unsigned int lane16 = lane * 16;                          // per-lane byte offset of the stream loads
    CH_STAMP(0);
    // ---- prefetch workgroups (ChainArgs::pf_wgs, the lowest workgroup ids: dispatched first, round-robin over the XCDs).  Every
    // computing workgroup of an XCD reads the SAME stream addresses at about the same time, so inside the model (other kernels have
    // emptied the L2s) each line is ONE L2 miss that all of them wait for.  A workgroup per XCD that does nothing but touch the
    // stream -- a 128-byte line per lane, 8 KiB per instruction and wave -- runs at the CU's line-fill rate (~37 B/clk: it takes
    // ~195k cycles for the 7 MB, about as long as a computing workgroup, so it stays just ahead of them without pacing; a wider
    // window, sleeps or more prefetchers per XCD measured the same) and turns those misses into hits: a computing workgroup's
    // lifetime 117 -> 99 us (32 rows) / 134 -> 116 us (48 rows) on the 100 MHz real-time counter, -0.5..0.7 ms per bench step.
    if (wg < 0) {
        const long bytes = (long)p.stream_frags * 1024;
        unsigned int sink = 0;
        // (pair form: 16 prefetchers, two per XCD, one per half -- a single one would need longer for both halves' streams than the
        // computing workgroups live and be the launch's tail)
        const int pfid = p.pf_wgs + wg, hh = pair ? (pfid >> 3) & 1 : 0;
        {
        const char* base = reinterpret_cast<const char*>(p.wstream) + (size_t)(hh * CHAIN_NW + wave) * (size_t)p.stream_frags * 1024;
#ifdef MTTS_CHAIN_STAMP
        if (p.kstamp && tid == 0) p.kstamp[16 + 2 * ((M + QB - 1) / QB + p.pf_wgs + wg)] = __builtin_amdgcn_s_memtime();
#endif
        const int parts = pair ? 1 : (p.pf_wgs + 7) / 8;
        for (long off = pair ? 0 : (long)(pfid >> 3) * 8192; off < bytes; off += 8192L * parts) {
            const long o = off + lane * 128;                        // (several prefetch workgroups per XCD interleave their 8 KiB pieces)
            const unsigned int ob = (unsigned int)(o < bytes ? o : bytes - 128);
            asm volatile("global_load_dword %0, %1, %2" : "+v"(sink) : "v"(ob), "s"(base));
            asm volatile("s_waitcnt vmcnt(8)");
        }
        }
        asm volatile("s_waitcnt vmcnt(0)");
        asm volatile("" : "+v"(sink));
#ifdef MTTS_CHAIN_STAMP
        __syncthreads();
        if (p.kstamp && tid == 0) p.kstamp[17 + 2 * ((M + QB - 1) / QB + p.pf_wgs + wg)] = __builtin_amdgcn_s_memtime();
#endif
        return;
    }
#ifdef MTTS_CHAIN_DUMP
    // diagnostic: copies of the LDS regions at the phase boundaries, per workgroup [x0 | ct | x1 | srow | h0 | x2] (p.kstamp = base)
    constexpr int DUMP_WG = 3 * K::XT_BYTES + K::CT_FLOATS * 4 + 2 * QB * 4 + K::HT_BYTES;
    auto dump = [&](int sect_off, const char* src, int bytes) {
        if (!p.kstamp) return;
        char* dst = reinterpret_cast<char*>(p.kstamp) + (size_t)wg * DUMP_WG + sect_off;
        for (int o = tid * 16; o < bytes; o += 64 * CHAIN_NW * 16) *reinterpret_cast<u32x4*>(dst + o) = *reinterpret_cast<const u32x4*>(src + o);
    };
#define CH_DUMP(off, src, bytes) dump(off, src, bytes)
#else
#define CH_DUMP(off, src, bytes) do { } while (0)
#endif

    // ---- the wave's weight stream through a register ring: fragment f of the current position sits in ring[f % R].
    // wpos: (uniform) address of the fragment that is the current position.
    if (wg >= 0 && m0 >= M) return;                                 // (pair form: the grid is rounded up to whole groups of 16)
    const char* wpos = reinterpret_cast<const char*>(p.wstream) + (size_t)(half * CHAIN_NW + wave) * (size_t)p.stream_frags * 1024;
    u32x4 ring[R];
#pragma unroll
    for (int i = 0; i < R; i += 2) CH_LOAD2(ring[i], ring[i + 1], lane16, wpos + i * 1024);
    // refill the two slots of tile t of a step whose first fragment is `frag` (position-relative) with the fragments R further on
    auto refill = [&](int slot, int frag) __attribute__((always_inline)) {
        CH_LOAD2(ring[slot % R], ring[(slot + 1) % R], lane16, wpos + (frag + R) * 1024);
    };

    // ---- staging coordinates (8 lanes per 128-byte line): thread -> row tid >> 3, 16-byte chunk tid & 7
    const int st_row = tid >> 3, st_chunk = tid & 7;
    const bool st_on = st_row < QB;
    const size_t st_grow = (size_t)min(m0 + st_row, M - 1);
    const int st_lds = st_row * 128 + ((st_chunk ^ ((st_row >> 1) & 7)) * 16);
    const unsigned int att_off = (unsigned int)((st_grow * p.ld_att + st_chunk * 8) * 2);
    u32x4 areg[2];
    if (has_out) {
        CH_LOAD(areg[0], att_off, reinterpret_cast<const char*>(p.att16));
        CH_LOAD(areg[1], att_off, reinterpret_cast<const char*>(p.att16) + (p.inner > 32 ? 128 : 0));
    }
    // residual stream tile and column constants -> LDS.  Every load is requested before the first LDS write: one round trip for
    // the whole prologue (plain loads; the compiler's waits for them also cover every asm load above).  The constants are read
    // per hidden chunk / per phase end; keeping them out of registers and out of the vector-memory queue keeps the k-loops free
    // of spills and of foreign waits.
    {
        const _Float16* src = p.x16 + st_grow * p.ld_x + st_chunk * 8;
        f16x8 xv[KG];
#pragma unroll
        for (int g0 = 0; g0 < KG; ++g0) xv[g0] = *reinterpret_cast<const f16x8*>(src + g0 * 64);
        constexpr int NCT = (K::CT_FLOATS / 4 + 64 * CHAIN_NW - 1) / (64 * CHAIN_NW);      // f32x4 per thread
        f32x4 cv[NCT];
        // one contiguous block (ChainArgs::consts), unconditional loads: a branch around a load makes hipcc wait for each one
        // separately -- the six-pointer version of this prologue spent three dependent round trips here
#pragma unroll
        for (int n = 0; n < NCT; ++n) {
            const int idx = (tid + n * 64 * CHAIN_NW) * 4;
            cv[n] = *reinterpret_cast<const f32x4*>(p.consts + min(idx, K::CT_FLOATS - 4));
            cv[n] *= (idx >= 12 * C && idx < 16 * C) ? 0.5f : 1.0f;            // SnakeBeta: 1 / (2 (exp(beta) + 1e-9))
        }
        if (st_on) {
#pragma unroll
            for (int g0 = 0; g0 < KG; ++g0) *reinterpret_cast<f16x8*>(XT + g0 * (QB * 128) + st_lds) = xv[g0];
        }
#pragma unroll
        for (int n = 0; n < NCT; ++n) {
            const int idx = (tid + n * 64 * CHAIN_NW) * 4;
            if (idx < K::CT_FLOATS) *reinterpret_cast<f32x4*>(CT + idx) = cv[n];
        }
    }
    CH_WAIT(0);
#pragma unroll
    for (int i = 0; i < R; ++i) CH_TIE(ring[i]);
    if (has_out) { CH_TIE(areg[0]); CH_TIE(areg[1]); }

    // activation fragment (B operand): rows 16 i + c of k-group kg of an image at `base`
    auto bfrag = [&](const char* base, int kg, int i, int plane) -> f16x8 {
        return *reinterpret_cast<const f16x8*>(base + kg * (QB * 128) + (16 * i + c) * 128 + (((plane * 4 + q) ^ swz) * 16));
    };
    // position of this lane's 4 consecutive channels ch..ch+3 of row 16 i + c inside an image: head word pair; the residual pair
    // sits at the same place of chunk + 4
    auto img_off = [&](int ch, int i, int plane) -> int {
        const int off = ch & 31;
        return (ch >> 5) * (QB * 128) + (16 * i + c) * 128 + ((((plane * 4) + (off >> 3)) ^ swz) * 16) + (off & 7) * 2;
    };

    f32x4 acc[NT][MT], accx[NT][MT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < MT; ++i) { acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[t][i] = acc[t][i]; }
    };
    // one k-step of a C-wide product: wait for its NT weight tiles (ring slots fb ..), multiply with the activation fragments of
    // k-group kg of `base`, then request the fragments R further on into the same slots.  The wait, by what else this file has
    // requested since (mode): 0 nothing -- the R - FW younger fragments of the ring; 1 the out-projection's attention rows (see
    // phase 0: FW + 1).  (fb, frag, mode: constants once the caller's loop is unrolled.)
    auto step_wide = [&](int fb, const char* base, int kg, int frag, int mode) __attribute__((always_inline)) {
        if (mode == 1) CH_WAIT(FW + 1);
        else CH_WAIT(R - FW);
#pragma unroll
        for (int f = 0; f < FW; ++f) CH_TIE(ring[(fb + f) % R]);
        // row tiles outermost: two activation fragments live at a time, each accumulator touched again only NT MFMAs later
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f16x8 bh = bfrag(base, kg, i, 0), bl = bfrag(base, kg, i, 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) accx[t][i] = mfma16(__builtin_bit_cast(f16x8, ring[(fb + 2 * t) % R]), bl, accx[t][i]);
#pragma unroll
            for (int t = 0; t < NT; ++t) accx[t][i] = mfma16(__builtin_bit_cast(f16x8, ring[(fb + 2 * t + 1) % R]), bh, accx[t][i]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][i] = mfma16(__builtin_bit_cast(f16x8, ring[(fb + 2 * t) % R]), bh, acc[t][i]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) refill(fb + 2 * t, frag + 2 * t);
    };
    // End of every k-loop: nothing this file requested may still be in flight when compiler-scheduled code follows.  The compiler
    // treats an asm output as available at once, so wherever it splits the live range of a ring register (it does around the
    // register-hungry epilogues) it copies the register BEFORE the load has landed -- the copy then holds the previous fragment.
    // That was the round-3 determinism failure: right results whenever the weights came out of the L2 fast enough, wrong ones for
    // a whole launch when they did not (first launch after other kernels had emptied the L2s).  tests/test_isa_guard.py walks the
    // control-flow graph of the built kernel and fails on any instruction that touches a register with its load outstanding.
    auto ring_drain = [&]() __attribute__((always_inline)) {
        CH_WAIT(0);
#pragma unroll
        for (int i = 0; i < R; ++i) CH_TIE(ring[i]);
    };
    float rmax = 0.f;
    // acc (+ bias + the residual rows in XT) -> XT, in place: this lane's channels 16 (wave NT + t) + 4 q .. + 3 of rows 16 i + c
    auto rows_to_xt = [&](const float* bias) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ch = 16 * (wave * NT + t) + 4 * q;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                char* ph = XT + img_off(ch, i, 0);
                char* pl = XT + img_off(ch, i, 1);
                const f16x4 rh = *reinterpret_cast<const f16x4*>(ph), rl = *reinterpret_cast<const f16x4*>(pl);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = (acc[t][i][e] + accx[t][i][e] * (1.0f / F16_RES_SCALE) + b4[e]) + ((float)rh[e] + (float)rl[e] * (1.0f / F16_RES_SCALE));
                u32x2 hw, lw;
                split4(v, F16_RES_SCALE, hw, lw, rmax);
                *reinterpret_cast<u32x2*>(ph) = hw;
                *reinterpret_cast<u32x2*>(pl) = lw;
            }
        }
    };
    // LayerNorm moments of the rows in XT -> srow = [mean x QB | rstd x QB].  A wave takes QB/8 rows, 8 lanes per row (each C/64
    // 8-channel chunks of it), all rows at once: two passes over values held in registers, reductions over 8 lanes by DPP.
    auto ln_stats = [&]() {
        constexpr int RPW = QB / CHAIN_NW, CPL = C / 64;  // rows per wave; chunks per lane
        const int rl = lane >> 3, part = lane & 7;
        const bool on = rl < RPW;
        const int row = wave * RPW + (on ? rl : 0), rs = (row >> 1) & 7;
        float x[CPL][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int ck = part * CPL + k;                // 8-channel chunk of the row: k-group ck >> 2, chunk ck & 3
            const char* b = XT + (ck >> 2) * (QB * 128) + row * 128;
            const f16x8 h = *reinterpret_cast<const f16x8*>(b + (((ck & 3) ^ rs) * 16));
            const f16x8 l = *reinterpret_cast<const f16x8*>(b + (((4 + (ck & 3)) ^ rs) * 16));
#pragma unroll
            for (int e = 0; e < 8; ++e) x[k][e] = (float)h[e] + (float)l[e] * (1.0f / F16_RES_SCALE);
            s += ((x[k][0] + x[k][1]) + (x[k][2] + x[k][3])) + ((x[k][4] + x[k][5]) + (x[k][6] + x[k][7]));
        }
        const float mean = allreduce8(s) * (1.0f / C);
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = x[k][e] - mean; m2 += d * d; }
        m2 = allreduce8(m2);
        if (on && part == 0) {
            srow[row] = mean;
            srow[QB + row] = 1.0f / sqrtf(m2 * (1.0f / C) + p.eps);
        }
    };

    // ================================================================ phase 0: out-projection + residual (reference transformer.py:261)
    if (has_out) {
        constexpr int PER0 = R / FW;                     // k-steps per ring period (even)
        constexpr int NS = KG2;                          // attention k-step stages in HT
        const int nk0 = p.inner >> 5;
        if (st_on) *reinterpret_cast<u32x4*>(HT + st_lds) = areg[0];
        zero_acc();
        __syncthreads();
        CH_STAMP(1);
        CH_DUMP(0, XT, K::XT_BYTES);
        CH_DUMP(K::XT_BYTES, reinterpret_cast<const char*>(CT), K::CT_FLOATS * 4);
        for (int s0 = 0; s0 < nk0; s0 += PER0) {
#pragma unroll
            for (int u = 0; u < PER0; ++u) {
                const int s = s0 + u;
                // rows of step s+2 into the register that held step s (in LDS since the previous step).  Queue, oldest first:
                // ... ATT(s+1) | refills of step s-1 (FW) | ATT(s+2): step s's weights are older than ATT(s+1), so one wait
                // for ATT(s+1) -- FW + 1 younger loads -- covers both.
                CH_LOAD(areg[u & 1], att_off, reinterpret_cast<const char*>(p.att16) + min(s + 2, nk0 - 1) * 128);
                const char* stage = HT + (s % NS) * (QB * 128);
                step_wide((u * FW) % R, stage, 0, u * FW, 1);        // (a stage holds one k-group)
                CH_TIE(areg[(u + 1) & 1]);
                if (st_on && s + 1 < nk0) *reinterpret_cast<u32x4*>(HT + ((s + 1) % NS) * (QB * 128) + st_lds) = areg[(u + 1) & 1];
                __syncthreads();
            }
            wpos += R * 1024;
        }
        ring_drain();
        CH_TIE(areg[0]); CH_TIE(areg[1]);
        CH_STAMP(2);
        rows_to_xt(CT + 16 * C);
        __syncthreads();
        CH_DUMP(K::XT_BYTES + K::CT_FLOATS * 4, XT, K::XT_BYTES);
    } else {
        __syncthreads();                                  // the x tile is in LDS
    }
    ln_stats();
    __syncthreads();
    CH_STAMP(3);
    CH_DUMP(2 * K::XT_BYTES + K::CT_FLOATS * 4, reinterpret_cast<const char*>(srow), 2 * QB * 4);

    // ================================================================ phase 1: FeedForward (reference transformer.py:278-301,104-120)
    zero_acc();
    {
        constexpr int F1 = KG * F1S;                      // fragments of a chunk's FF1 part
        const int nch = pair ? K::NCH / 2 : K::NCH, j0 = half * (K::NCH / 2);
        for (int jj = 0; jj < nch; ++jj) {
            const int j = j0 + jj;
            f32x4 a1[NT1][MT], a1x[NT1][MT];
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i) { a1[t][i] = f32x4{0.f, 0.f, 0.f, 0.f}; a1x[t][i] = a1[t][i]; }
            // ---- FF1: hidden chunk^T = W1'[chunk] . x^T
#pragma unroll
            for (int s = 0; s < KG; ++s) {
                const int fb = (s * F1S) % R;
                CH_WAIT(R - F1S);
#pragma unroll
                for (int f = 0; f < F1S; ++f) CH_TIE(ring[(fb + f) % R]);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const f16x8 bh = bfrag(XT, s, i, 0), bl = bfrag(XT, s, i, 1);
#pragma unroll
                    for (int t = 0; t < NT1; ++t) a1x[t][i] = mfma16(__builtin_bit_cast(f16x8, ring[(fb + 2 * t) % R]), bl, a1x[t][i]);
#pragma unroll
                    for (int t = 0; t < NT1; ++t) a1[t][i] = mfma16(__builtin_bit_cast(f16x8, ring[(fb + 2 * t) % R]), bh, a1[t][i]);
#pragma unroll
                    for (int t = 0; t < NT1; ++t) a1x[t][i] = mfma16(__builtin_bit_cast(f16x8, ring[(fb + 2 * t + 1) % R]), bh, a1x[t][i]);
                }
#pragma unroll
                for (int t = 0; t < NT1; ++t) refill(fb + 2 * t, s * F1S + 2 * t);
            }
            ring_drain();
            if (jj < 2) CH_STAMP(4 + 3 * jj);
            // ---- LayerNorm after the product, SnakeBeta, split -> hidden chunk image in HT
            float nmr[MT], rstd[MT];                      // this lane's rows: -mean rstd, rstd
#pragma unroll
            for (int i = 0; i < MT; ++i) { rstd[i] = srow[QB + 16 * i + c]; nmr[i] = -(srow[16 * i + c] * rstd[i]); }
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                const int hl = 16 * (wave * NT1 + t) + 4 * q;          // channel inside the chunk
                const float* cc = CT + j * CH + hl;
                const f32x4 cw = *reinterpret_cast<const f32x4*>(cc), cb = *reinterpret_cast<const f32x4*>(cc + 4 * C),
                            cs0 = *reinterpret_cast<const f32x4*>(cc + 8 * C), cs1h = *reinterpret_cast<const f32x4*>(cc + 12 * C);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = a1[t][i][e] + a1x[t][i][e] * (1.0f / F16_RES_SCALE);
                        const float z = __builtin_fmaf(y, rstd[i], __builtin_fmaf(nmr[i], cw[e], cb[e]));
                        v[e] = snake_hw(z, cs0[e], cs1h[e]);
                    }
                    u32x2 hw, lw;
                    split4(v, F16_RES_SCALE, hw, lw, rmax);
                    *reinterpret_cast<u32x2*>(HT + img_off(hl, i, 0)) = hw;
                    *reinterpret_cast<u32x2*>(HT + img_off(hl, i, 1)) = lw;
                }
            }
            __syncthreads();
            if (jj < 2) CH_STAMP(5 + 3 * jj);
            if (jj == 0) CH_DUMP(2 * K::XT_BYTES + K::CT_FLOATS * 4 + 2 * QB * 4, HT, K::HT_BYTES);
            // ---- FF2: out^T += W2[:, chunk] . hidden chunk^T
#pragma unroll
            for (int s = 0; s < KG2; ++s) step_wide((F1 + s * FW) % R, HT, s, F1 + s * FW, 0);
            ring_drain();
            wpos += (F1 + KG2 * FW) * 1024;
            __syncthreads();                              // the hidden chunk may be overwritten
            if (jj < 2) CH_STAMP(6 + 3 * jj);
        }
    }
    CH_STAMP(10);
    if (pair) {
        // ---- the two halves' FF2 partial sums meet: each workgroup publishes its own (fp32, global scratch), waits for the other's
        // flag and forms x2 = ((x1 + b2) + P0) + P1 -- the same expression in both, so both hold the same bits.  Publication:
        // every wave's stores drained, barrier, then ONE lane releases at agent scope and stores the flag; the reader polls relaxed,
        // acquires once, and the barrier behind it orders every lane's loads (MI355X_MICROARCH.md, handoff recipe).  Both
        // workgroups are resident (the launcher refuses pair grids beyond one round of the chip); the poll is bounded all the same.
        // (lane-native layout [tile][half][t][i][thread] of 16-byte pieces: a wave instruction writes / reads 1 KiB of whole lines;
        // the partner's thread of the same index owns the same element)
        float* const mine = p.pair_part + (size_t)(rtile * 2 + half) * QB * C;
        const float* const other = p.pair_part + (size_t)(rtile * 2 + (half ^ 1)) * QB * C;
        auto slot = [&](int t, int i) { return (size_t)((t * MT + i) * (64 * CHAIN_NW) + tid) * 4; };
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][i][e] = acc[t][i][e] + accx[t][i][e] * (1.0f / F16_RES_SCALE);
                // write-through stores (sc1): the partial leaves the L2 at once and the release below finds nothing dirty to write back
                // (plain stores measured slower: 13k instead of 3.6k cycles until they had landed, 27k instead of 23k for the exchange)
                float* dst = mine + slot(t, i);
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(acc[t][i]) : "memory");
            }
        }
        CH_XSTAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CH_XSTAMP(1);
        __syncthreads();
        CH_XSTAMP(2);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CH_XSTAMP(3);
            __hip_atomic_store(p.pair_flag + rtile * 2 + half, p.pair_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int budget = 1 << 21;
            while (__hip_atomic_load(p.pair_flag + rtile * 2 + (half ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.pair_epoch && --budget > 0)
                __builtin_amdgcn_s_sleep(8);
            CH_XSTAMP(4);
            if (budget <= 0 && p.range_flag) atomicOr(p.range_flag + 1, 1u);      // the partner never showed up: the call's results are void
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CH_XSTAMP(5);
        }
        __syncthreads();
        CH_XSTAMP(6);
        const float* bias = CT + 17 * C;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ch = 16 * (wave * NT + t) + 4 * q;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const f32x4 o4 = *reinterpret_cast<const f32x4*>(other + slot(t, i));
                const f32x4 p0 = half ? o4 : acc[t][i], p1 = half ? acc[t][i] : o4;
                char* ph = XT + img_off(ch, i, 0);
                char* pl = XT + img_off(ch, i, 1);
                const f16x4 rh = *reinterpret_cast<const f16x4*>(ph), rl = *reinterpret_cast<const f16x4*>(pl);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = ((((float)rh[e] + (float)rl[e] * (1.0f / F16_RES_SCALE)) + b4[e]) + p0[e]) + p1[e];
                u32x2 hw, lw;
                split4(v, F16_RES_SCALE, hw, lw, rmax);
                *reinterpret_cast<u32x2*>(ph) = hw;
                *reinterpret_cast<u32x2*>(pl) = lw;
            }
        }
        CH_XSTAMP(7);
    } else {
        rows_to_xt(CT + 17 * C);
    }
    __syncthreads();
    CH_DUMP(2 * K::XT_BYTES + K::CT_FLOATS * 4 + 2 * QB * 4 + K::HT_BYTES, XT, K::XT_BYTES);

    // ---- the block's output rows: LDS image -> global image, whole 16-byte chunks, coalesced
    {
        constexpr int CPR = C / 4;                        // 16-byte chunks per row
        const int r_lo = pair ? half * (QB / 2) : 0, r_hi = pair ? (half + 1) * (QB / 2) : QB;      // (pair form: each half stores half the rows)
        for (int idx = tid + r_lo * CPR; idx < r_hi * CPR; idx += 64 * CHAIN_NW) {
            const int row = idx / CPR, cc = idx - row * CPR;
            if (m0 + row < M) {
                f16x8 v = *reinterpret_cast<const f16x8*>(XT + (cc >> 3) * (QB * 128) + row * 128 + (((cc & 7) ^ ((row >> 1) & 7)) * 16));
                if (p.x_out_mask && p.x_out_mask[m0 + row] == 0.f) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<f16x8*>(p.x_out + (size_t)(m0 + row) * p.ld_out + cc * 8) = v;
            }
        }
    }

    CH_STAMP(11);
    // ================================================================ phase 2: the next block's q|k|v (reference transformer.py:249-258)
    if (has_qkv) {
        ln_stats();
        float* const QC = reinterpret_cast<float*>(HT);   // the hidden-chunk area is free: panel row sums | bias of the q|k|v columns
        for (int idx = tid * 4; idx < 2 * p.n_qkv; idx += 4 * 64 * CHAIN_NW)
            *reinterpret_cast<f32x4*>(QC + idx) = *reinterpret_cast<const f32x4*>(idx < p.n_qkv ? p.wsum_qkv + idx : p.b_qkv + (idx - p.n_qkv));
        __syncthreads();
        const int ntiles = p.n_qkv >> 4;
        const int passes = (ntiles + CHAIN_NW * NT - 1) / (CHAIN_NW * NT);
        const bool all_stores = m0 + QB <= M;             // (uniform) no row of the tile is past the end
        const int p_half = (passes + 1) / 2;
        const int ps0 = pair ? (half ? p_half : 0) : 0, ps1 = pair ? (half ? passes : p_half) : passes;
        for (int ps = ps0; ps < ps1; ++ps) {
            zero_acc();
#pragma unroll
            for (int s = 0; s < KG; ++s) step_wide((s * FW) % R, XT, s, s * FW, 0);
            ring_drain();
            wpos += KG * FW * 1024;
            CH_STAMP(13 + (ps < 2 ? ps : 2));
            float nmr[MT], rstd[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) { rstd[i] = srow[QB + 16 * i + c]; nmr[i] = -(srow[16 * i + c] * rstd[i]); }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tile = ps * CHAIN_NW * NT + wave * NT + t;
                if (tile < ntiles) {
                    const int col = 16 * tile + 4 * q;
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(QC + col), b4 = *reinterpret_cast<const f32x4*>(QC + p.n_qkv + col);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int row = m0 + 16 * i + c;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = acc[t][i][e] + accx[t][i][e] * (1.0f / F16_RES_SCALE);
                            v[e] = __builtin_fmaf(y, rstd[i], __builtin_fmaf(nmr[i], w4[e], b4[e]));
                        }
                        u32x2 hw, lw;
                        split4(v, 1.0f, hw, lw, rmax);      // unscaled residuals: the attention kernel's operands
                        if (all_stores || row < M) {
                            _Float16* dst = p.qkv16 + (size_t)row * p.ld_qkv + (col >> 5) * 64 + (col & 31);
                            *reinterpret_cast<u32x2*>(dst) = hw;
                            *reinterpret_cast<u32x2*>(dst + 32) = lw;
                        }
                    }
                }
            }
        }
    }
    raise_range_flag(p.range_flag, rmax > 65504.f);
    CH_STAMP(12);
#ifdef MTTS_CHAIN_PROBE
    // diagnostic: per workgroup [HW_ID, XCC_ID, hash of the constants in LDS, hash of the kernel arguments as this wave holds them,
    // hash of the row statistics, -, -, -] (p.kstamp = base, zeroed by the host)
    if (p.kstamp) {
        unsigned int* rec = reinterpret_cast<unsigned int*>(p.kstamp) + (size_t)wg * 8;
        unsigned int hc = 0, hs = 0;
        for (int i = tid; i < K::CT_FLOATS; i += 64 * CHAIN_NW) hc ^= (__float_as_uint(CT[i]) + 0x9e3779b9u * (unsigned)i) * 2654435761u;
        for (int i = tid; i < 2 * QB; i += 64 * CHAIN_NW) hs ^= (__float_as_uint(srow[i]) + 0x9e3779b9u * (unsigned)i) * 2654435761u;
        atomicXor(rec + 2, hc);
        atomicXor(rec + 4, hs);
        if (lane == 0) {
            const unsigned long long ptrs[10] = {(unsigned long long)p.att16, (unsigned long long)p.x16, (unsigned long long)p.wstream, (unsigned long long)p.b_out,
                (unsigned long long)p.b1, (unsigned long long)p.wsum1, (unsigned long long)p.p0, (unsigned long long)p.p1, (unsigned long long)p.b2,
                (unsigned long long)p.x_out_mask};
            unsigned long long h = (unsigned long long)p.M * 1315423911ull + (unsigned long long)p.stream_frags;
            for (int i = 0; i < 10; ++i) h = (h ^ ptrs[i]) * 1099511628211ull;
            atomicXor(rec + 3, (unsigned int)(h ^ (h >> 32)) * (1u + 0u * wave));      // all 8 waves: equal hashes cancel pairwise -> 0 when they agree
            atomicAdd(rec + 5, (unsigned int)(h ^ (h >> 32)) == 0u ? 0u : 1u);
            if (wave == 0) {
                rec[0] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_ID
                rec[1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // XCC_ID
                rec[6] = (unsigned int)(h ^ (h >> 32));
            }
        }
    }
#endif
}

template <int C, int QB, int CH>
static hipError_t launch_chain_shape(const ChainArgs& a, hipStream_t s) {
    using K = ChainCfg<C, QB, CH>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "LDS per workgroup");
    static bool configured = false;
    auto kern = tblock_chain_kernel<C, QB, CH>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
        if (e != hipSuccess) return e;
        configured = true;
    }
    static const std::string tag = "tblock_chain_kernel<" + std::to_string(C) + ", " + std::to_string(QB) + ", " + std::to_string(CH) + ">";
    g_kernel_tag = tag.c_str();
    const int tiles = (a.M + QB - 1) / QB;
    const int wgs = a.pair ? 16 * ((tiles + 7) / 8) : tiles;
    if (a.pair && (wgs + a.pf_wgs > 256 || (a.pf_wgs != 0 && a.pf_wgs != 16))) return hipErrorInvalidValue;      // both halves of every pair must be resident at once
    hipLaunchKernelGGL(kern, dim3(wgs + (a.pf_wgs > 0 ? a.pf_wgs : 0)), dim3(64 * CHAIN_NW), K::LDS_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_tblock_chain(const ChainArgs& a, hipStream_t s) {
    if (a.pf_wgs < 0 || a.pf_wgs > 64) return hipErrorInvalidValue;
    if (a.M <= 0 || !a.x16 || !a.wstream || !a.x_out || !a.consts) return hipErrorInvalidValue;
    if (!chain_supported(a.C, a.inner, a.n_qkv)) return hipErrorInvalidValue;
    if (a.inner && (!a.att16 || a.ld_att < 2 * a.inner || (a.ld_att & 7))) return hipErrorInvalidValue;
    if (a.b_qkv && (!a.wsum_qkv || !a.qkv16 || a.n_qkv <= 0 || a.ld_qkv < 2 * a.n_qkv || (a.ld_qkv & 3))) return hipErrorInvalidValue;
    if (a.ld_x < 2 * a.C || (a.ld_x & 7) || a.ld_out < 2 * a.C || (a.ld_out & 7)) return hipErrorInvalidValue;
    if (a.pair) {
        if (!chain_supported_pair(a.C, a.inner, a.ch, a.n_qkv) || !a.pair_part || !a.pair_flag || a.pair_epoch == 0 || (a.pf_wgs & 7)) return hipErrorInvalidValue;
        if (a.stream_frags != chain_stream_frags_pair(a.C, a.inner, a.ch, a.b_qkv ? a.n_qkv : 0)) return hipErrorInvalidValue;
    } else if (a.stream_frags != chain_stream_frags(a.C, a.inner, a.ch, a.b_qkv ? a.n_qkv : 0)) return hipErrorInvalidValue;
    if (a.C == 384) {
        if (a.ch == 128 && a.qb == 64) return launch_chain_shape<384, 64, 128>(a, s);
        if (a.ch == 128 && a.qb == 32) return launch_chain_shape<384, 32, 128>(a, s);
        if (a.ch == 256 && a.qb == 48) return launch_chain_shape<384, 48, 256>(a, s);
        if (a.ch == 256 && a.qb == 32) return launch_chain_shape<384, 32, 256>(a, s);
    } else if (a.C == 256) {
        if (a.ch == 128 && a.qb == 64) return launch_chain_shape<256, 64, 128>(a, s);
        if (a.ch == 128 && a.qb == 32) return launch_chain_shape<256, 32, 128>(a, s);
    } else if (a.C == 128) {
        if (a.ch == 128 && a.qb == 64) return launch_chain_shape<128, 64, 128>(a, s);
        if (a.ch == 128 && a.qb == 32) return launch_chain_shape<128, 32, 128>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace mtts
