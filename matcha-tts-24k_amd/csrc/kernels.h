// Internal kernel-launch interface of libmtts_hip.so (gfx950 only).
// Every launcher enqueues on the given stream and returns hipError_t; none allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace mtts {

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 32;
constexpr int MAX_TAPS = 7;

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Name of the kernel instantiation the last launcher call on this thread chose, spelled as rocprofv3 prints it (e.g.
// "gemm_p16_kernel<64, false, 3, 0, true, false, 2>"): the per-launch event pass (mtts_prof_*) stores it with each record, so a
// per-instantiation table can join measured FLOP/s with the profiler's counters (tools/kernel_table.py).  Null = untagged.
extern thread_local const char* g_kernel_tag;
static inline const char* tf(bool b) { return b ? "true" : "false"; }

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_SNAKE = 3, ACT_GELU = 4 };

// C[M,N] = epi( pro(A)[M,K] . W[N,K]^T )   -- see gemm_f32.hip for the tiling.
struct GemmArgs {
    // ---- A operand: rows of channels-last activations; up to two channel segments (concat along K per tap)
    const float* a0 = nullptr;
    const float* a1 = nullptr;
    int lda0 = 0, lda1 = 0;
    int c0 = 0, c1 = 0;          // channels read from each segment (c0 % 32 == 0 whenever a1 != nullptr)
    int ktap = 0;                // padded K per tap = round_up(c0 + c1, 32)
    int ntaps = 1;
    int tap_off[MAX_TAPS] = {0, 0, 0, 0, 0, 0, 0};
    int in_stride = 1;           // t_in = t_out * in_stride + tap_off[tap]
    int B = 1, T_in = 1, T_out = 1;   // M = B * T_out
    const float* a_mask = nullptr;    // [B*T_in]  multiply A rows
    const float* a_mean = nullptr;    // [B*T_in]  (x - mean) * rstd   (LayerNorm with the affine folded into W); ntaps == 1
    const float* a_rstd = nullptr;
    const float* a_part = nullptr;    // alternative to a_mean/a_rstd: per-row partial moments [B*T_in][a_nparts][2] = (mean, M2) of
    int a_nparts = 0;                 //   64-column slices, written by the producing GEMM's epilogue (stats_out); merged in the prologue
    float a_eps = 1e-5f;
    // ---- B operand: packed [Np][Kp], Np = round_up(N,128), Kp = ntaps * ktap
    const float* w = nullptr;
    const void* w16 = nullptr;        // split modes: bf16 planes [3][Np][Kp] of the same panel (w = h + m + l)
    int terms = 0;                    // 0: fp32 MFMA; 6 / 3: bf16 MFMA, 2: fp16 MFMA on split operands (see gemm_f32.hip)
    const float* bias = nullptr;      // [Np] or null
    int N = 0;
    // ---- epilogue: c = act(acc + bias); c *= out_mask[row]; c = c * out_scale + res[row][n]
    int act = ACT_NONE;
    const float* p0 = nullptr;        // snake: exp(alpha)[N]
    const float* p1 = nullptr;        // snake: 1 / (exp(beta) + 1e-9)[N]
    const float* out_mask = nullptr;  // [B*out_T] indexed by the output row
    float out_scale = 1.0f;
    const float* res = nullptr;
    int ldr = 0;
    float* out = nullptr;
    int ldc = 0;
    float* stats_out = nullptr;       // [M][N/64][2]: (mean, M2) of every 64-column slice of the output rows (N % 64 == 0)
    int out_T = 1, out_stride = 1, out_off = 0;   // output row = b*out_T + t*out_stride + out_off
    int force_bm = 0;                 // 0 = choose the block tile height from the grid size; 64 / 128 = force (tests)
    // ---- P16 operands (gemm_p16.hip; terms == 2 only).  A "P16" tensor is the fp16 two-term split of an fp32 tensor kept
    // in memory: row-major, per row and 32-channel group 32 heads then 32 residuals (64 halves = one 128-B line), so a
    // tensor of C channels has rows of 2*C halves.  When a16_0 is set the A tiles come from these images by LDS-DMA and
    // a0/a1/a_mask are ignored; LayerNorm (a_part or a_mean/a_rstd) is then applied in the epilogue as
    // rstd * (x.W' - mean * wsum) with wsum[n] = sum_k W'[n][k].
    const _Float16* a16_0 = nullptr;  // segment 0 image, c0 % 32 == 0
    const _Float16* a16_1 = nullptr;  // segment 1 image or null, c1 % 32 == 0
    int lda16_0 = 0, lda16_1 = 0;     // row strides in halves
    const float* wsum = nullptr;      // [Np] row sums of the panel (LayerNorm-in-the-epilogue)
    _Float16* out16 = nullptr;        // optional P16 copy of the output (N % 32 == 0)
    int ld16 = 0;                     // its row stride in halves
    float out_lscale = 2048.0f;       // residual scale of out16: 2048 for GEMM operands, 1 for the attention kernel's q|k|v
    const float* out16_mask = nullptr;// [rows] multiplies the out16 copy only (conv consumers read masked rows; out stays as is)
    const _Float16* res16 = nullptr;  // residual as a P16 image (2^11-scaled residual plane) instead of fp32 rows; N % 32 == 0
    int ldr16 = 0;                    // its row stride in halves; may alias out16 (in-place update of the residual stream)
    // GroupNorm statistics of the output from the epilogue (conv -> Block1D, P16 kernel): per wave tile (R = BM/2 rows x 64
    // columns), per PART of it (part 0 = the rows of the tile's first utterance, part 1 = those of the next one: T_out >= R, so a
    // tile touches at most two) and per group slice (<= 2: needs (N / gn_groups) >= 32) an entry (n, mean, M2, -) at
    // gn_stats[(((tile * 2 + part) * (N/64) + col_wave) * 2 + slice) * 4]; launch_gn_apply merges them (tile_stats, tile_rows = R
    // = gemm_p16_wave_rows).  Needs plain rows, N % 64 == 0 and an epilogue of bias only; any T_out >= R.
    float* gn_stats = nullptr;
    int gn_groups = 0;
    const int* gn_nrows = nullptr;    // [B] rows of each utterance that enter the statistics (the GroupNorm kernels' nrows); null = T_out
    // Block1D tail in the epilogue (ResNet output, P16 kernel): c += Mish(GroupNorm(y)[row][n]) * gnr_mask[row], where y is the
    // second conv's fp32 output and its statistics are the tile entries that conv's epilogue left (gn_stats there): this GEMM
    // is the ResNet's 1x1 residual conv, so the sum is the ResNet output (reference decoder.py:58-63) and no gn_apply pass or
    // residual round trip remains.  Needs T_out >= BM (a workgroup's rows in at most two utterances) and N / gnr_groups >= 32.
    const float* gnr_y = nullptr;     // [M][N] fp32 rows (ld = N)
    const float* gnr_stats = nullptr; // entries as GemmArgs::gn_stats of the producing conv
    int gnr_tile_rows = 0, gnr_groups = 0;
    // filled by launch_gemm_p16 (host), not by callers: reciprocals for the P16 kernel's index arithmetic -- q = umulhi(n, rcp) is
    // n / d exactly while n * d < 2^32 (device_utils.h fdiv; an integer division costs ~25 vector instructions, and a
    // workgroup's prologue had ~20 of them in front of its first tile request)
    unsigned int rcp_T_out = 0, rcp_ntiles = 0, rcp_gn_cpg = 0, rcp_gnr_cpg = 0, rcp_gnr_R = 0;
    int gn_cpg = 0, gnr_cpg = 0;
    const float* gnr_gamma = nullptr; const float* gnr_beta = nullptr; const float* gnr_mask = nullptr;
    float gnr_eps = 1e-5f;
    const int* gnr_nextra = nullptr;      // folded padding (GnApplyArgs::nextra / bias_stats): copies of the producing conv's bias
    const float* gnr_bias_stats = nullptr;//   row that belong to the statistics without existing as rows
    bool fast16 = false;              // P16 kernel only: heads x heads product alone (fp16 operands, fp32 accumulate), see MTTS_GEMM_TERMS=1
    // 16-bit storage mode (BASELINE config #3; mtts_set_arithmetic(ctx, 16)): every image named "16" above is an "H16" image --
    // ONE fp16 plane, rows of C halves (C % 64 == 0: a 128-byte line = 64 channels), 2 bytes per element -- the weight plane is
    // w16h [Np][Kp] halves, and a MAC is one v_mfma_f32_16x16x32_f16 with fp32 accumulation.  Row strides stay in halves.
    bool half16 = false;
    const void* w16h = nullptr;
    // the same mode with BFLOAT16 planes (mtts_set_arithmetic(ctx, 17): what BASELINE configs[2] names): identical layout and data
    // movement, v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16 on the planes, v_cvt_pk_bf16_f32 in the producers; the fp32 exponent
    // range, so no range guard, 8 significand bits instead of 11
    bool bf16 = false;
    // fp16-split arithmetic only: set to 1 (sticky, atomicOr) when an operand this launch splits -- an A element while staging
    // (gemm_f32.hip, terms 2) or an out16 element (epilogue) -- lies beyond +-65504 and saturates.  Null = no check.
    unsigned int* range_flag = nullptr;
    // diagnostic builds only (-DMTTS_KSTAMP, tools/kstamp.py): per workgroup 8 words = s_memtime at kernel start, first tile
    // landed, loop end, epilogue end, then s_memrealtime at start and end.  No output value depends on them.
    unsigned long long* kstamp = nullptr;
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
hipError_t launch_gemm_p16(const GemmArgs& a, hipStream_t s);      // called by launch_gemm when a.a16_0 is set
int gemm_p16_wave_rows(const GemmArgs& a);                          // rows of a wave tile (BM/2) launch_gemm_p16 will use for these shapes
// fp32 rows [M][ld] -> P16 image [M][ld16 halves] of channels [0, C) (C % 32 == 0), optionally times mask[row]
hipError_t launch_to_p16(const float* x, int ld, const float* mask, int M, int C, int C_valid, _Float16* out, int ld16, float lscale,
                         hipStream_t s, unsigned int* range_flag = nullptr, bool half16 = false, bool bf16 = false);   // half16: H16 image, lscale unused   // channels [C_valid, C) are written as zeros
hipError_t launch_from_p16(const _Float16* x, int ld16, int M, int C, float lscale, float* out, int ld, hipStream_t s);
static inline double gemm_flops(const GemmArgs& a) {
    return 2.0 * double(a.B) * a.T_out * a.N * double(a.ntaps) * (a.c0 + a.c1);
}
// compulsory HBM bytes of one launch: every input row, weight, residual and output element once (a P16 image has the bytes of
// the fp32 tensor it stands for; an fp32 and a P16 copy of the output count separately, as does the Block1D-tail input)
static inline double gemm_bytes(const GemmArgs& a) {
    const double M = double(a.B) * a.T_out, Min = double(a.B) * a.T_in;
    const double outs = (a.out ? 1.0 : 0.0) + (a.out16 ? 1.0 : 0.0);
    const double ins = ((a.res || a.res16) ? 1.0 : 0.0) + (a.gnr_y ? 1.0 : 0.0);
    return 4.0 * (Min * (a.c0 + a.c1) + double(a.N) * a.ntaps * (a.c0 + a.c1) + M * a.N * (outs + ins));
}

// Pack a torch weight into the GEMM panel layout [Np][ntaps*ktap] (host side).
//   kind 0: Linear [N, C]   kind 1: Conv1d [N, C, ntaps]   kind 2: ConvTranspose1d [C, N, kT] taking taps `tsel[0..ntaps)`
void pack_weight_host(const float* w, int kind, int N, int C, int ntaps, int kT, const int* tsel, const float* col_scale,
                      float* dst, int ktap = 0);      // ktap: padded K per tap (0 = round_up(C, 32))
// Device-side packing of an unpacked Linear/Conv1d weight (used by the single-kernel test entry point).
hipError_t launch_pack_weight(const float* w, int N, int C, int ntaps, float* dst, hipStream_t s);
void split_panel_host(const float* panel, size_t n, uint16_t* planes);
hipError_t launch_split_panel(const float* panel, size_t n, void* planes, hipStream_t s);
void split_panel_f16_host(const float* panel, size_t n, uint16_t* planes);
void panel_h16_host(const float* panel, size_t n, uint16_t* plane);      // the fp16 head plane alone, same element order as the panel
void panel_bf16_host(const float* panel, size_t n, uint16_t* plane);     // the same as bfloat16 (round to nearest even)
hipError_t launch_split_panel_f16(const float* panel, size_t n, void* planes, hipStream_t s);

struct AttnArgs {
    const float* qkv = nullptr;   // [B*T, 3*H*D]
    const float* mask = nullptr;  // [B*T] float 0/1
    float* out = nullptr;         // [B*T, H*D]
    int B = 0, T = 0, H = 0, D = 0;
    float scale = 1.0f;
    int mask_mode = 0;            // 0 additive key bias, 1 boolean query*key
    // P16 I/O (D == 64): q|k|v as a P16 image with unscaled residuals, output as a P16 image (residual times out_lscale)
    const _Float16* qkv16 = nullptr; int ld16 = 0;     // row stride in halves (>= 6*H*64)
    _Float16* out16 = nullptr; int ldo16 = 0;          // row stride in halves (>= 2*H*64)
    float out_lscale = 2048.0f;
    const int* klen = nullptr;    // [B] keys of utterance b are rows [0, klen[b]); null = T.  With folded padding `mask` holds the
                                  // additive key bias itself: 1 for valid frames, ln(n_pad) for the one row that stands for n_pad
                                  // identical padded frames (reference bias +0 each)
    bool fast16 = false;          // P16 I/O only: single fp16 product per MAC (no residual terms)
    unsigned int* range_flag = nullptr;   // P16 output: sticky flag for values beyond +-65504 (GemmArgs::range_flag)
    bool half16 = false;          // q|k|v and the output are H16 images (GemmArgs::half16): a head = 64 contiguous halves
    bool bf16 = false;            // ... holding bfloat16 instead of fp16 values (GemmArgs::bf16)
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
static inline double attn_flops(const AttnArgs& a) { return 4.0 * double(a.B) * a.H * double(a.T) * a.T * a.D; }
static inline double attn_bytes(const AttnArgs& a) { return 4.0 * double(a.B) * a.T * 4.0 * a.H * a.D; }

// ---- transformer-block chain (tblock_chain.hip): the row-local part of a BasicTransformerBlock as ONE launch.
// Behind the attention everything up to the next attention is row-local (reference transformer.py:261-301): out-projection +
// residual, LayerNorm, FeedForward (Linear - SnakeBeta - Linear) + residual and, when another block follows, its LayerNorm'd
// q|k|v projection.  A workgroup (8 waves) keeps QB rows of the residual stream in LDS for the whole chain and streams the
// weights from L2 straight into registers: they are stored as "fragment streams" (chain_stream_*), per wave the 1 KiB MFMA
// operand fragments in exactly the order the wave consumes them, so every load is one coalesced global_load_dwordx4 per lane
// and a register ring keeps ~12 KiB per wave in flight.  Neither the attention output tile, the 4C-wide hidden layer nor the
// LayerNorm moments touch HBM between the phases.  Arithmetic: the fp16 two-term split of gemm_p16.hip (three MFMAs per MAC).
constexpr int CHAIN_WAVES = 8;           // waves per workgroup of the chain kernel = per-wave streams in a fragment stream
struct ChainArgs {
    int M = 0;                           // rows (B * T): rows are independent, a workgroup takes QB consecutive ones
    int C = 0;                           // width of the residual stream (128, 256 or 384)
    int inner = 0;                       // attention width (heads * 64) = K of the out-projection; 0: no out-projection phase
    const _Float16* att16 = nullptr;     // P16 image of the attention output [M][ld_att halves]
    int ld_att = 0;
    const _Float16* x16 = nullptr;       // P16 image of the residual stream x [M][ld_x halves] (the residual; FF input when inner == 0)
    int ld_x = 0;
    const _Float16* wstream = nullptr;   // fragment stream of the whole chain (chain_stream_pack), [8 waves][frags][64 lanes][8 halves]
    long stream_frags = 0;               // fragments per wave incl. the tail padding
    // column constants, ONE block of 18 C floats: rowsum(W1') | b1' (LayerNorm shift folded in) | SnakeBeta exp(alpha) |
    // SnakeBeta 1 / (exp(beta) + 1e-9) (4 C each) | out-projection bias (zeros without that phase) | b2 (C each)
    const float* consts = nullptr;
    const float* b_qkv = nullptr;        // [3 * inner] or null: no q|k|v phase
    const float* wsum_qkv = nullptr;
    int n_qkv = 0;                       // 3 * inner
    _Float16* x_out = nullptr;           // P16 image of the block's output rows [M][ld_out halves] (may alias x16)
    int ld_out = 0;
    const float* x_out_mask = nullptr;   // [M] or null: rows of x_out are multiplied by it (0 / 1): the masked copy convs read
    _Float16* qkv16 = nullptr;           // P16 image (unscaled residuals) of the next block's q|k|v [M][ld_qkv halves]
    int ld_qkv = 0;
    float eps = 1e-5f;
    unsigned int* range_flag = nullptr;
    int qb = 0, ch = 0;                  // rows per workgroup (64 / 48 / 32) and hidden chunk (128 / 256) the stream was packed for
    // pair form (tblock_chain.hip): two workgroups of one XCD per row tile, each streams half of the FeedForward's chunks and of
    // the q|k|v passes; FF2 partial sums meet in `pair_part` [2 tiles][QB][C] fp32 behind the flags `pair_flag` [2 tiles] == pair_epoch
    int pair = 0;
    float* pair_part = nullptr;
    unsigned int* pair_flag = nullptr;
    unsigned int pair_epoch = 0;         // a value no earlier launch on these flags used
    int pf_wgs = 0;                      // extra workgroups (lowest ids) that only touch the weight stream ahead of the others; 8 = one per XCD, 16 = two (the model's default; the pair form takes 16: one per XCD and half)
    unsigned long long* kstamp = nullptr;// diagnostic builds only (-DMTTS_CHAIN_STAMP): 16 phase stamps of workgroup 0
};
// fragments per wave of the stream for (C, inner, hidden chunk, q|k|v width), incl. the padding the register ring may run into
long chain_stream_frags(int C, int inner, int ch, int n_qkv);
// Host packing from the fp32 panels ([N][K] row-major, K contiguous): w_out [C][inner], w1 [4C][C], w2 [C][4C], w_qkv [n_qkv][C];
// null panels leave their phase out.  dst: chain_stream_frags(...) * 8 * 512 halves.
void chain_stream_pack(int C, int inner, int ch, int n_qkv, const float* w_out, const float* w1, const float* w2, const float* w_qkv,
                       uint16_t* dst, bool* saturates);
bool chain_supported(int C, int inner, int n_qkv);
long chain_stream_frags_pair(int C, int inner, int ch, int n_qkv);       // pair form: fragments per (half, wave)
void chain_stream_pack_pair(int C, int inner, int ch, int n_qkv, const float* w_out, const float* w1, const float* w2, const float* w_qkv,
                            uint16_t* dst, bool* saturates);             // dst: 2 * 8 * chain_stream_frags_pair(...) * 512 halves
bool chain_supported_pair(int C, int inner, int ch, int n_qkv);
hipError_t launch_tblock_chain(const ChainArgs& a, hipStream_t s);
static inline double chain_flops(const ChainArgs& a) {
    return 2.0 * double(a.M) * (double(a.C) * a.inner + 8.0 * double(a.C) * a.C + double(a.C) * a.n_qkv);
}
static inline double chain_bytes(const ChainArgs& a) {
    return 4.0 * (double(a.M) * (a.inner + 2.0 * a.C + a.n_qkv) + double(a.C) * a.inner + 8.0 * double(a.C) * a.C + double(a.C) * a.n_qkv);
}

// ---- normalisation / activation / glue (norm_glue.hip)
hipError_t launch_row_stats(const float* x, int M, int C, int ld, float eps, float* mean, float* rstd, hipStream_t s);

// y = act(LN_C(x) * gamma + beta) [* film_g[b] + film_b[b]] [* mask[row]]   (channel LayerNorm of the text encoder)
struct LayerNormArgs {
    const float* x = nullptr; int ldx = 0;
    float* y = nullptr; int ldy = 0;
    int M = 0, C = 0, T = 1;           // T rows per batch element (for the FiLM lookup)
    const float* gamma = nullptr; const float* beta = nullptr;
    float eps = 1e-5f;
    int act = ACT_NONE;                // ACT_SILU for the prenet
    const float* film = nullptr;       // [B, 2C]: gamma | beta (DurationPredictor)
    const float* mask = nullptr;       // [M]
};
hipError_t launch_layernorm(const LayerNormArgs& a, hipStream_t s);

// GroupNorm statistics over (C/G channels x T frames) of y [B,T,C]: partial (mean, M2) per chunk of GN_CHUNK rows.
// The chunk height adapts to the batch: 32 rows when that already gives >= 2048 workgroups (192-thread workgroups of a pure
// stream need ~10 per CU to keep HBM busy: 4.3 -> 4.1 ms of streaming kernels per step at B = 32), down to 8 rows otherwise
// (B = 1, T = 640: 80 workgroups of 4 row passes instead of 20 of 16 -- these kernels are latency-bound there).
constexpr int GN_CHUNK = 32;       // largest chunk
constexpr int GN_CHUNK_MIN = 8;
static inline int gn_min_blocks() {
    static const int v = [] { const char* e = getenv("MTTS_GN_BLOCKS"); return e ? atoi(e) : 2048; }();     // env: A/B runs only
    return v;
}
static inline int gn_chunk_rows(int B, int T) {
    int rows = GN_CHUNK;
    while (rows > GN_CHUNK_MIN && (long)B * ((T + rows - 1) / rows) < gn_min_blocks()) rows >>= 1;
    return rows;
}
static inline int gn_chunks(int B, int T) { const int r = gn_chunk_rows(B, T); return (T + r - 1) / r; }
static inline int gn_chunks_max(int T) { return (T + GN_CHUNK_MIN - 1) / GN_CHUNK_MIN; }      // scratch sizing
// nrows (optional, [B] int32): only frames [0, nrows[b]) of utterance b enter the statistics
hipError_t launch_gn_partial(const float* y, int B, int T, int C, int G, float* partial, hipStream_t s, const int* nrows = nullptr);
// out = Mish(GN(y)) ; out = (out [+ chbias[c]]) * mask[row] ; out += res[row][c]
struct GnApplyArgs {
    const float* y = nullptr; const float* partial = nullptr;
    const float* gamma = nullptr; const float* beta = nullptr;
    const float* mask = nullptr;      // [B*T]
    const float* chbias = nullptr;    // [C] (time-embedding bias of the ResNet block) or null
    const float* res = nullptr; int ldr = 0;
    float* out = nullptr;
    float* stats_out = nullptr;       // [B*T][C/64][2] LayerNorm partial moments of the output rows (C % 64 == 0)
    _Float16* out16 = nullptr;        // optional P16 copy of the output rows (C % 32 == 0), row stride ld16 halves; out may then be null
    int ld16 = 0;
    const float* out16_mask = nullptr;// [B*T] multiplies the out16 copy only
    const int* nrows = nullptr;       // [B] rows of utterance b that enter the statistics; null = T
    // Folded padding: the reference pads every utterance to the batch-wide length and GroupNorm counts those frames; beyond the
    // first padded frame the conv output is exactly its bias row, so nextra[b] copies of that row enter the statistics in closed
    // form: per group (mean of the bias, sum of squared deviations of ONE copy) = bias_stats[g][2], packed with the weights.
    const int* nextra = nullptr;
    const float* bias_stats = nullptr;
    int chunk_rows = 0;               // set by launch_gn_apply (gn_chunk_rows(B, T)), must match the partial pass
    const float* tile_stats = nullptr;// alternative to `partial`: the entries a P16 GEMM's epilogue left (GemmArgs::gn_stats)
    int tile_rows = 0;                // rows per wave tile of that GEMM (gemm_p16_wave_rows); T % tile_rows == 0
    int B = 0, T = 0, C = 0, G = 8; float eps = 1e-5f;
    unsigned int* range_flag = nullptr;   // out16: sticky flag for values beyond +-65504 (GemmArgs::range_flag)
    bool half16 = false;                  // out16 is an H16 image (GemmArgs::half16)
    bool bf16 = false;                    // ... of bfloat16 values (GemmArgs::bf16)
};
hipError_t launch_gn_apply(const GnApplyArgs& a, hipStream_t s);

// channels-first [B,C,T] <-> channels-last [B,T,ld] moves
// dst[b*T + t, col_off + c] = src[b,c,t] (+ add[b,c,t]) for t < T; the source rows are T_src long (T_src = 0: T)
hipError_t launch_cf_to_cl(const float* src, const float* add, int B, int C, int T, float* dst, int ld, int col_off, hipStream_t s,
                           int T_src = 0);
// dst[b,c,t] = src[b,t,c] * scale + shift for t < T_out
hipError_t launch_cl_to_cf(const float* src, int ld, int B, int C, int T, float* dst, int T_out, float scale, float shift, hipStream_t s);
hipError_t launch_fill_cols(float* dst, int M, int ld, int col0, int ncols, float v, hipStream_t s);

// elementwise helpers
constexpr int MAX_EVALS = 256;
struct TimeVals { float t[MAX_EVALS]; };
hipError_t launch_time_sinusoid(const float* freqs, const TimeVals& tv, int nt, int half, float scale, float* out, hipStream_t s);
hipError_t launch_unary(const float* x, float* y, int64_t n, int act_mish, hipStream_t s);
// ODE state updates on the x columns of the channels-last state; stage semantics in norm_glue.hip
hipError_t launch_ode_combine(int stage, float dt, const float* y, int ldy, const float* k1, const float* k2, const float* k3,
                              const float* k4, int ldk, float* out, int ldo, int M, int C, hipStream_t s);

// text-encoder glue
hipError_t launch_embedding(const int64_t* ids, const float* table, int rows, int C, float scale, const float* mask, float* out, int ld, hipStream_t s);
hipError_t launch_seq_mask(const int64_t* lengths, int B, int T, float* mask, hipStream_t s);
hipError_t launch_bcast_rows(const float* src, int B, int T, int C, const float* mask, float* dst, int ld, int col_off, hipStream_t s);
hipError_t launch_rope(float* qkv, int B, int T, int H, int D, int d_rope, const float* cos_t, const float* sin_t, hipStream_t s);
hipError_t launch_durations(const float* logw, const float* mask, float scale_correction, float length_scale, int B, int Tx, float* dur,
                            int32_t* cum, int64_t* yfl, hipStream_t s, const float* sc_b = nullptr, const float* ls_b = nullptr);
// level mask of the U-Net: dst[b, t] = src[b, t * stride], t < T_dst (reference decoder.py:390 mask[:, :, ::2])
hipError_t launch_mask_down(const float* src, int B, int T_src, int stride, float* dst, int T_dst, hipStream_t s);
// Per-level frame tables of one estimator call (norm_glue.hip).  Level l has T[l] rows per utterance.
//  folded (y_len != null): the reference runs the estimator on T_true frames per utterance of which only the first y_len[b] are
//    valid; all padded frames of a level are identical rows, so the folded layout keeps L_l = ceil(y_len / 2^l) valid
//    rows + ONE row for the n_pad = (T_true >> l) - L_l padded ones:  mask = [t < L_l], kbias = 1 | ln(n_pad) at t = L_l,
//    nrows = L_l + (n_pad > 0) (attention keys, GroupNorm rows), nextra = max(n_pad - 1, 0) (GroupNorm bias-row copies).
//  unfolded (y_len == null, tlen != null; per-request padding): nrows = min(T[l], tlen[b] >> l), nextra = 0, masks untouched.
//  tlen (optional, [B]): utterance b's own padded length T_true (per-request padding); else T_true for all.
struct FrameTableArgs {
    const int64_t* y_len = nullptr;
    const int* tlen = nullptr;
    int B = 0, T_true = 0, nl = 0;
    int T[4] = {0, 0, 0, 0};
    float* mask[4] = {nullptr, nullptr, nullptr, nullptr};
    float* kbias[4] = {nullptr, nullptr, nullptr, nullptr};
    int* nrows[4] = {nullptr, nullptr, nullptr, nullptr};
    int* nextra[4] = {nullptr, nullptr, nullptr, nullptr};
};
hipError_t launch_frame_tables(const FrameTableArgs& a, hipStream_t s);
hipError_t launch_align_pool(const float* mu_x, const int32_t* cum, const int64_t* yfl, int B, int nf, int Tx, int T_pad,
                             float* mu_y, float* y_mask, int64_t* y_len, hipStream_t s);

// ---- Vocos head (vocos.hip)
// y = LayerNorm_C(depthwise_conv_k7(x) + bias) * gamma + beta on channels-last rows [B*T, C]; w7 is [7][C]
hipError_t launch_dwconv7_ln(const float* x, const float* w7, const float* bias, const float* gamma, const float* beta, float eps,
                             int B, int T, int C, float* y, hipStream_t s);
// in place on rows [M, ld]: (log-magnitude m_k, phase p_k) at columns (k, off + k) -> (Re, Im) = min(exp(m),clip) * (cos p, sin p)
hipError_t launch_spec_polar(float* x, int M, int ld, int nbins, int off, float clip, hipStream_t s);
// overlap-add of windowed frames [B*T, n_fft] (hop) with the squared-window envelope, centre trim: audio [B, hop*(T-1)]
hipError_t launch_istft_ola(const float* frames, const float* window, int B, int T, int n_fft, int hop, float* audio, hipStream_t s);

}  // namespace mtts
