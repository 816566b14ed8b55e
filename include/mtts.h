/*
 * mtts.h -- C ABI of libmtts_hip.so: the MI355X (gfx950) implementation of the mel-synthesis
 * hot path of faltiska/Matcha-TTS-24k.
 *
 * The reference has no native code and no FFI on this path: its boundary is the Python module
 * matcha/inference.py.  This header is what a Python (ctypes) or C++ host
 * binds instead of calling PyTorch ops; each entry point cites the reference function it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (fp32 unless stated); h_* is a HOST pointer
 *   - `stream` is a hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream)
 *   - functions return 0 on success, <0 on error; mtts_last_error() gives a thread-local message
 *   - launch functions never allocate, never synchronise and never touch the default stream:
 *     scratch memory is a caller-provided workspace sized by the matching *_workspace_bytes()
 *   - tensors use the reference's layouts at the boundary: activations [B, C, T] ("channels first"),
 *     ids/lengths int64
 */
#ifndef MTTS_H
#define MTTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTTS_ABI_VERSION 2
/* bumped whenever the packed weight image changes layout (invalidates mtts_export_weights caches) */
#define MTTS_IMAGE_REVISION 5

typedef struct mtts_ctx mtts_ctx;

/* Architecture of the path; mirrors the checkpoint's hyper_parameters
 * (reference matcha/inference.py:45-55, configs/experiment/v20.yaml:17-63). */
typedef struct mtts_config {
    int32_t n_feats;        /* mel bins (100) */
    int32_t n_spks;
    int32_t spk_emb_dim;    /* 96 */
    int32_t n_vocab;        /* 600 */
    /* text encoder, reference text_encoder.py:319-373 */
    int32_t enc_channels;   /* 192; hidden = enc_channels + spk_emb_dim */
    int32_t enc_filter;     /* 1152 */
    int32_t enc_heads;      /* 6 */
    int32_t enc_layers;     /* 4 */
    int32_t enc_kernel;     /* 5 */
    int32_t prenet_layers;  /* 6 */
    int32_t prenet_kernel;  /* 3 */
    int32_t dp_filter;      /* 96 */
    int32_t dp_kernel;      /* 5 */
    int32_t dp_layers;      /* 4 */
    /* decoder, reference decoder.py:202-310 */
    int32_t dec_levels;     /* len(channels) == 2 */
    int32_t dec_channels[4];
    int32_t dec_head_dim;   /* 64 */
    int32_t dec_heads;      /* 6 */
    int32_t dec_n_blocks;   /* 2 */
    int32_t dec_mid_blocks; /* 2 */
} mtts_config;

enum { MTTS_SOLVER_EULER = 0, MTTS_SOLVER_MIDPOINT = 1, MTTS_SOLVER_RK4 = 2 };

/* ---------------------------------------------------------------- library / context */
int mtts_abi_version(void);
const char* mtts_last_error(void);

/* Create a context for one architecture.  Host-side only (no device work). */
mtts_ctx* mtts_create(const mtts_config* cfg);
void mtts_destroy(mtts_ctx* ctx);

/* Register one tensor of the reference state dict by its key, e.g.
 * "decoder.estimator.down_blocks.0.0.block1.block.0.weight".  h_data: host fp32, copied.
 * Replaces nn.Module.load_state_dict (reference inference.py:186-197). */
int mtts_set_tensor(mtts_ctx* ctx, const char* key, const float* h_data, int64_t numel);

/* Threading: a context carries per-call state (the call's range-flag pointer, frame limits, the profiler's records): one thread
 * drives a context at a time -- use one context per serving worker / stream.  The path's entry points (mtts_text_encoder_forward,
 * mtts_decoder_forward, mtts_cfm_solve*) enforce it: a call that finds the context held by another thread returns -1 ("in use
 * by another thread") instead of interleaving.  Only mtts_last_error is thread-local. */

/* (test hook: holds the context as a path entry point does for `ms` milliseconds; a concurrent entry-point call fails) */
int mtts_debug_hold(mtts_ctx* ctx, int ms);

/* Packed-image cache ("pre-packed MFMA weight layouts cached beside the converted checkpoint").
 * mtts_weights_signature: a string naming everything the image layout depends on (ABI and image revision, architecture,
 * arithmetic, layout switches); mtts_export_weights copies the packed image (mtts_weights_bytes) to host memory;
 * mtts_import_weights adopts such an image in a context with the same signature whose tensors have been registered
 * (mtts_set_tensor) -- it runs the layout pass only, not the splitting / fragment packing (~7 s at production size).
 * `saturates`: the range-guard finding of the packing pass (mtts_weights_saturate), stored with the image. */
int mtts_weights_signature(mtts_ctx* ctx, char* buf, int64_t n);
int mtts_export_weights(mtts_ctx* ctx, void* h_dst, int64_t bytes, int* saturates);
int mtts_import_weights(mtts_ctx* ctx, const void* h_src, int64_t bytes, int saturates);

/* After all tensors are registered: size of the packed device image, then pack + upload it
 * (GEMM-ready [N][K] panels, conv taps unrolled along K, LayerNorm affine folded into the
 * following projection, exp() of the SnakeBeta parameters).  Synchronous; load time only. */
int64_t mtts_weights_bytes(mtts_ctx* ctx);
int mtts_upload_weights(mtts_ctx* ctx, void* d_weights, int64_t bytes);

/* ---------------------------------------------------------------- the path */

/* TextEncoder.forward -- reference matcha/models/components/text_encoder.py:375-406.
 * d_x [B,Tx] int64, d_x_lengths [B] int64, d_e_enc/d_e_dur [B,spk_emb_dim].
 * Outputs: d_mu_x [B,n_feats,Tx], d_logw [B,1,Tx], d_x_mask [B,1,Tx] (float 0/1). */
int64_t mtts_encoder_workspace_bytes(mtts_ctx* ctx, int B, int Tx);
int mtts_text_encoder_forward(mtts_ctx* ctx, const int64_t* d_x, const int64_t* d_x_lengths, const float* d_e_enc,
                              const float* d_e_dur, int B, int Tx, float* d_mu_x, float* d_logw, float* d_x_mask,
                              void* d_ws, int64_t ws_bytes, void* stream);

/* Speaker table lookup -- reference inference.py:115-121 (table: 0 = enc, 1 = dur); d_ids [B] int64. */
int mtts_speaker_embedding(mtts_ctx* ctx, int table, const int64_t* d_ids, int B, float* d_out, void* stream);

/* Durations -- reference inference.py:127-146:
 * round((exp(logw)-2)*mask*scale_correction*length_scale).clamp(min=1)*mask, their inclusive cumulative sum and the
 * per-utterance fine length clamp_min(sum,1).
 * d_durations [B,Tx] f32, d_cum [B,Tx] int32, d_y_fine_lengths [B] int64. */
int mtts_durations(const float* d_logw, const float* d_x_mask, float scale_correction, float length_scale, int B, int Tx,
                   float* d_durations, int32_t* d_cum, int64_t* d_y_fine_lengths, void* stream);

/* The same with one (scale_correction, length_scale) pair per utterance, device float [B] each: a serving batch mixes voices
 * (reference inference.py:16-32 VOICES[..]["scale_correction"], server.py:111-115) and client speeds. */
int mtts_durations_per_utterance(const float* d_logw, const float* d_x_mask, const float* d_scale_correction,
                                 const float* d_length_scale, int B, int Tx, float* d_durations, int32_t* d_cum,
                                 int64_t* d_y_fine_lengths, void* stream);

/* generate_path + matmul + downsample + sequence_mask -- reference inference.py:146-167,
 * utils/model.py:7-9,24-40,57-68.  T_pad = fix_len_compatibility(max fine length) (host decides it).
 * Outputs: d_mu_y [B,n_feats,T_pad], d_y_mask [B,1,T_pad], d_y_lengths [B] int64. */
int mtts_align_pool(const float* d_mu_x, const int32_t* d_cum, const int64_t* d_y_fine_lengths, int B, int n_feats,
                    int Tx, int T_pad, float* d_mu_y, float* d_y_mask, int64_t* d_y_lengths, void* stream);

/* Per-request padding inside a batch.  The reference computes T_pad, the GroupNorm statistics, the attention key set and
 * the noise shape from the longest utterance of the call (inference.py:147-148, decoder.py:32-45, transformer.py:249-261),
 * so an utterance's mel depends on what it was batched with.  With d_t_len[b] (device int32, even, <= T) set, the next
 * mtts_decoder_forward / mtts_cfm_solve calls treat utterance b as if only frames [0, d_t_len[b]) existed: GroupNorm
 * statistics and attention keys stop there (convolutions already read masked zeros beyond it), so every utterance of a
 * ragged batch gets the values of a batch-of-one call (up to the summation order of differently shaped tiles).  NULL restores whole-batch padding.  The pointer is kept,
 * not copied: it must stay valid until replaced. */
int mtts_set_frame_limits(mtts_ctx* ctx, const int32_t* d_t_len);

/* Decoder.forward -- reference matcha/models/components/decoder.py:359-426 (one evaluation of the velocity field).
 * d_x, d_mu, d_out [B,n_feats,T]; d_mask [B,1,T]; t scalar. */
int64_t mtts_decoder_workspace_bytes(mtts_ctx* ctx, int B, int T);
int mtts_decoder_forward(mtts_ctx* ctx, const float* d_x, const float* d_mask, const float* d_mu, float t, int B, int T,
                         float* d_out, void* d_ws, int64_t ws_bytes, void* stream);

/* BASECFM.solve -- reference matcha/models/components/flow_matching.py:60-63 + torchdiffeq fixed-grid odeint
 * (euler / midpoint / rk4 = 3/8 rule) over the grid h_t_span[0..n_steps].
 * d_x0 [B,n_feats,T] initial state; if add_mu != 0 the state starts at d_x0 + d_mu (use_mu_prior,
 * flow_matching.py:52-55).  d_out [B,n_feats,T_out] receives state[:, :, :T_out]*out_scale + out_shift
 * (the slice and denormalize of reference inference.py:170-172; pass T_out=T, 1, 0 for the raw state). */
int mtts_cfm_solve(mtts_ctx* ctx, const float* d_x0, const float* d_mu, const float* d_mask, int add_mu,
                   const float* h_t_span, int n_steps, int solver, int B, int T, float* d_out, int T_out,
                   float out_scale, float out_shift, void* d_ws, int64_t ws_bytes, void* stream);

/* The same solve on FOLDED padding -- what MatchaTTSInfer.synthesise runs (reference matcha/inference.py:146-170 pads the
 * decoder to T = roundup_even(longest fine length), i.e. twice the valid mel length, and GroupNorm / attention see those
 * frames: decoder.py:32-45, transformer.py:249-261).  With prefix masks (frame t of utterance b valid iff t < d_y_lengths[b])
 * every padded frame of a U-Net level is the same row: convolutions read `x * mask` (decoder.py:43,62), so beyond the first
 * padded frame a conv output is its bias, a ResNet output is the residual conv's bias, and transformer blocks act row-wise with
 * keys that carry no position.  The estimator therefore holds, per utterance and level l, the ceil(y_len / 2^l) valid rows plus
 * ONE row standing for the n_pad padded ones: attention gives that key the bias ln(n_pad) (n_pad reference keys of bias +0),
 * GroupNorm merges the remaining n_pad - 1 bias rows in closed form.  Same results as mtts_cfm_solve on the full [B, n_feats, T]
 * problem up to summation order; T_fold / T of the arithmetic.
 *   d_x0, d_mu [B, n_feats, T] as for mtts_cfm_solve (only frames < T_fold are read; frames of the state at or beyond an
 *   utterance's own length pass through unchanged, as in the reference); d_y_lengths device int64 [B], all <= y_max < T.
 *   T_fold: rows held per utterance, a multiple of 2^(levels-1) with mtts_fold_rows(ctx, y_max, 1) <= T_fold <= T;
 *   mtts_fold_rows(ctx, y_max, align) = roundup(ceil(y_max / 2^(levels-1)) + 1, align) * 2^(levels-1) (align 32 keeps whole
 *   wave tiles per utterance for the fused GroupNorm statistics).  Workspace: mtts_decoder_workspace_bytes(ctx, B, T_fold).
 *   mtts_set_frame_limits composes: utterance b's reference length is then d_t_len[b] instead of T. */
int mtts_fold_rows(mtts_ctx* ctx, int y_max, int align);
int mtts_cfm_solve_folded(mtts_ctx* ctx, const float* d_x0, const float* d_mu, const int64_t* d_y_lengths, int y_max, int add_mu,
                          const float* h_t_span, int n_steps, int solver, int B, int T, int T_fold, float* d_out, int T_out,
                          float out_scale, float out_shift, void* d_ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- single kernels (parity tests, building blocks) */

/* C[M,N] = epilogue(prologue(A)[M,K] * W[N,K]^T): the fp32-MFMA GEMM every Linear/Conv1d of the path runs on.
 * A is [B*T_in, lda] row major (channels last).  ntaps>1 makes it an implicit 1-D convolution:
 * K = ntaps*C, output row (b,t) reads input rows t*in_stride + tap_off[tap].  W is the *unpacked* torch weight:
 * Linear [N,C] or Conv1d [N,C,ntaps]; it is packed into d_wpacked (mtts_gemm_packed_bytes) on the stream first
 * (d_w = NULL: d_wpacked already holds the packed panel from an earlier call).
 * act: 0 none, 1 relu, 2 silu, 3 SnakeBeta with d_p0 = exp(alpha)[N], d_p1 = 1/(exp(beta)+1e-9)[N]
 * (reference transformer.py:61-77).  Epilogue: c = act(acc + bias); c *= out_mask[row]; c = c*out_scale + res[row][n].
 * terms: arithmetic of the products, all with fp32 accumulation: 0 = v_mfma_f32_32x32x2_f32 (native fp32);
 * 2 = operands split into two fp16 terms with a 2^11-scaled residual (22 significand bits), three f16 MFMA products --
 * measured error equals the fp32 chain's, inputs beyond +-65504 saturate; 6 = three exact bf16 terms, six bf16 MFMA
 * products (full fp32 range); 3 = two bf16 terms (looser, opt-in); -1 = the library default (2, or MTTS_GEMM_TERMS).
 * LayerNorm prologue: either d_a_mean/d_a_rstd [rows], or d_a_part [rows][a_nparts][2] = (mean, M2) of 64-column slices
 * as written by a previous call's d_stats_out [M][N/64][2] (N % 64 == 0).  All optional pointers may be NULL. */
int64_t mtts_gemm_packed_bytes(int N, int C, int ntaps);
int mtts_gemm_f32(const float* d_a, int lda, int B, int T_in, int C, int ntaps, const int* h_tap_off, int in_stride,
                  int T_out, const float* d_a_mask, const float* d_a_mean, const float* d_a_rstd, const float* d_a_part,
                  int a_nparts, const float* d_w,
                  void* d_wpacked, const float* d_bias, int N, int act, const float* d_p0, const float* d_p1,
                  const float* d_res, int ldr, const float* d_out_mask, float out_scale, float* d_out, int ldc,
                  float* d_stats_out, int terms, void* stream);

/* P16 GEMM (csrc/gemm_p16.hip): same contract as mtts_gemm_f32 in its fp16-split mode, but the A operand is first
 * written as a "P16" image (fp16 head + scaled fp16 residual, 128-B lines per 32 channels; here by a conversion pass into
 * d_scratch, in the model by the producing kernel's epilogue) and both tiles reach LDS by LDS-DMA.  C % 32 == 0, N % 4 == 0.
 * LayerNorm statistics (arrays or partial moments) are applied in the epilogue: rstd * (x.W - mean * rowsum(W)).
 * d_out (fp32) and/or d_out16_f32 (the P16 output image decoded back to fp32 [M][N], N % 32 == 0, residual scale
 * out_lscale) receive the result.  Replaces F.linear / F.conv1d like mtts_gemm_f32 (reference decoder.py, transformer.py). */
int64_t mtts_gemm_p16_scratch_bytes(int B, int T_in, int C, int T_out, int N);
int mtts_gemm_p16(const float* d_a, int lda, int B, int T_in, int C, int ntaps, const int* h_tap_off, int in_stride, int T_out,
                  const float* d_a_mask, const float* d_a_mean, const float* d_a_rstd, const float* d_a_part, int a_nparts,
                  const float* d_w, void* d_wpacked, const float* d_bias, int N, int act, const float* d_p0, const float* d_p1,
                  const float* d_res, int ldr, const float* d_out_mask, float out_scale, float* d_out, int ldc,
                  float* d_out16_f32, float out_lscale, float* d_stats_out, int force_bm, void* d_scratch, void* stream);

/* Self-attention over packed [B*T, 3*H*D] q|k|v rows -> [B*T, H*D].  mask_mode 0: additive float key bias
 * (diffusers semantics, reference transformer.py:253-258); 1: boolean query*key mask (reference
 * text_encoder.py:228-235,306).  d_mask [B,T] float 0/1. */
int mtts_attention_f32(const float* d_qkv, const float* d_mask, int B, int T, int H, int D, float scale, int mask_mode,
                       float* d_out, void* stream);

/* mtts_attention_f32 with P16 I/O (D == 64): q|k|v read as a P16 image (unscaled residuals), output written as a P16
 * image; here both conversions happen around the kernel, in d_scratch (>= 16*B*T*H*64 bytes). */
int mtts_attention_p16(const float* d_qkv, const float* d_mask, int B, int T, int H, int D, float scale, int mask_mode,
                       float* d_out, void* d_scratch, void* stream);

/* Transformer-block chain (csrc/tblock_chain.hip): the row-local part of a BasicTransformerBlock behind the attention as ONE
 * launch -- x1 = x + att . W_out^T + b_out (reference transformer.py:261); x2 = x1 + W2 . SnakeBeta(W1' . LN(x1) + b1') + b2
 * (transformer.py:278-301, FeedForward :104-120, SnakeBeta :61-77); qkv = W_qkv' . LN(x2) + b_qkv' (the following block's
 * norm1 + to_q/k/v, transformer.py:249-258).  LN = LayerNorm without affine (eps 1e-5): the caller folds gamma / beta into the
 * primed panels, as the model's packer does.  Rows are independent; att [M][inner], x [M][C] fp32 on the device; the panels
 * ([N][K] row major), biases and SnakeBeta constants (p0 = exp(alpha), p1 = 1 / (exp(beta) + 1e-9), [4C]) on the HOST -- this
 * test entry packs the fragment stream itself.  C in {128, 256, 384}, inner % 32 == 0 (0: FeedForward only), h_w_qkv NULL: no
 * q|k|v phase.  d_out_mask [M] (0/1) or NULL multiplies the rows of x_out (the masked image convs read).  qb: rows per
 * workgroup (64 / 48 / 32), ch: hidden chunk (128; 256 with C = 384, qb 48 / 32).  Outputs: x_out [M][C], qkv_out [M][n_qkv]. */
/* Host-only: the fragment stream of a chain (no device needed).  Per wave (8): [out-projection: inner/32 k-steps x C/128 tiles]
 * [per hidden chunk of ch: C/32 k-steps x ch/128 tiles of w1, then ch/32 k-steps x C/128 tiles of w2][q|k|v passes: C/32 k-steps x
 * C/128 tiles] + ring padding; a tile = the fp16 head fragment then the 2^11-scaled residual fragment, a fragment = 64 lanes x 8
 * halves with lane (r = lane & 15, q = lane >> 4) holding panel row n0 + r, columns k0 + 8 q .. + 7 (the A operand of
 * v_mfma_f32_16x16x32_f16).  h_dst: mtts_chain_stream_frags(...) * 8 * 512 halves. */
int64_t mtts_chain_stream_frags(int C, int inner, int ch, int n_qkv);
/* Threading / sharing note for the estimator entry points: between 3000 and 6000 estimator rows the transformer blocks run the PAIR
 * form of the chain launch (below), in which two workgroups wait for each other inside the kernel.  It assumes the launch has the GPU
 * to itself (one stream per device at a time); a process that shares a device between several contexts sets MTTS_CHAIN_PAIR=0.  A
 * workgroup whose partner does not arrive within ~0.5 s gives up and sets the second word of the workspace header. */
/* The model's launch plan of a chain launch over M rows (hidden chunk ch = 128 / 256): rows per workgroup and the number of
 * prefetch workgroups (MTTS_CHAIN_PF, default 16).  Host arithmetic only. */
int mtts_chain_plan(int M, int ch, int* qb, int* prefetch_wgs);
int mtts_chain_stream_pack(int C, int inner, int ch, int n_qkv, const float* h_w_out, const float* h_w1, const float* h_w2,
                           const float* h_w_qkv, uint16_t* h_dst);
/* pair form (below): fragments per (half, wave) and the packing of the 2 x 8 streams, [half][wave][fragment][64 lanes][8 halves];
 * h_dst: 2 * 8 * mtts_chain_stream_frags_pair(...) * 512 halves.  Host only. */
int64_t mtts_chain_stream_frags_pair(int C, int inner, int ch, int n_qkv);
int mtts_chain_stream_pack_pair(int C, int inner, int ch, int n_qkv, const float* h_w_out, const float* h_w1, const float* h_w2,
                                const float* h_w_qkv, uint16_t* h_dst);
int64_t mtts_tblock_chain_scratch_bytes(int M, int C, int inner, int n_qkv, int ch);
int mtts_tblock_chain(const float* d_att, const float* d_x, int M, int C, int inner, const float* h_w_out, const float* h_b_out,
                      const float* h_w1, const float* h_b1, const float* h_p0, const float* h_p1, const float* h_w2,
                      const float* h_b2, const float* h_w_qkv, const float* h_b_qkv, int n_qkv, const float* d_out_mask, int qb,
                      int ch, float* d_x_out, float* d_qkv_out, void* d_scratch, void* stream);
/* the same, followed by `repeat` further launches of the kernel alone between two events: *h_ms = their mean duration */
int mtts_tblock_chain_timed(const float* d_att, const float* d_x, int M, int C, int inner, const float* h_w_out, const float* h_b_out,
                            const float* h_w1, const float* h_b1, const float* h_p0, const float* h_p1, const float* h_w2,
                            const float* h_b2, const float* h_w_qkv, const float* h_b_qkv, int n_qkv, const float* d_out_mask,
                            int qb, int ch, float* d_x_out, float* d_qkv_out, void* d_scratch, void* stream, int repeat, float* h_ms);
/* the PAIR form of the same launch: two workgroups of one XCD share a row tile, each streams half of the FeedForward's hidden chunks
 * and of the q|k|v passes and they exchange their FF2 partial sums through the L2 (csrc/tblock_chain.hip).  Needs an out-projection
 * (inner > 0), an even number of hidden chunks and a grid that is resident at once: 16 * ceil(ceil(M / qb) / 8) + 16 <= 256. */
int mtts_tblock_chain_pair_timed(const float* d_att, const float* d_x, int M, int C, int inner, const float* h_w_out, const float* h_b_out,
                                 const float* h_w1, const float* h_b1, const float* h_p0, const float* h_p1, const float* h_w2,
                                 const float* h_b2, const float* h_w_qkv, const float* h_b_qkv, int n_qkv, const float* d_out_mask,
                                 int qb, int ch, float* d_x_out, float* d_qkv_out, void* d_scratch, void* stream, int repeat, float* h_ms);

/* Row statistics for LayerNorm over C (biased variance, eps inside rsqrt): mean[M], rstd[M]. */
int mtts_row_stats(const float* d_x, int M, int C, int ld, float eps, float* d_mean, float* d_rstd, void* stream);

/* Channel LayerNorm of the text encoder -- reference text_encoder.py:19-27 -- over x [B*T, C] rows, optionally followed by
 * SiLU (act = 2: ConvSiluNorm, text_encoder.py:58-60), the DurationPredictor's speaker FiLM `* gamma_b + beta_b`
 * (d_film [B, 2C] = gamma | beta, text_encoder.py:102-109) and the row mask (d_mask [B*T]).  Null pointers skip a stage. */
int mtts_channel_layernorm(const float* d_x, int B, int T, int C, const float* d_gamma, const float* d_beta, float eps, int act,
                           const float* d_film, const float* d_mask, float* d_y, void* stream);

/* Block1D tail -- reference decoder.py:38-45: Mish(GroupNorm_G(y)) * mask over y [B,T,C] (channels last);
 * statistics over (C/G channels x all T frames).  d_scratch: mtts_groupnorm_scratch_bytes. */
int64_t mtts_groupnorm_scratch_bytes(int B, int T, int G);
int mtts_groupnorm_mish(const float* d_y, const float* d_gamma, const float* d_beta, const float* d_mask, int B, int T,
                        int C, int G, float eps, float* d_out, void* d_scratch, void* stream);

/* ---------------------------------------------------------------- Vocos-24k head */

/* Vocos.decode -- reference matcha/vocos24k/vocos_wrapper.py:8-9 (third-party vocos package; architecture sizes from
 * reference matcha/vocos24k/config.yaml:10-24).  mel [B, n_mels, T] -> audio [B, hop*(T-1)] (torch.istft, center=True).
 * Tensors are registered under the vocos state-dict keys ("backbone.embed.weight", "backbone.convnext.0.dwconv.weight",
 * "head.out.weight", ...) plus "aux.window" = the periodic hann window [n_fft]. */
typedef struct mtts_vocos mtts_vocos;
mtts_vocos* mtts_vocos_create(int n_mels, int dim, int intermediate_dim, int num_layers, int n_fft, int hop_length);
void mtts_vocos_destroy(mtts_vocos* v);
int mtts_vocos_set_tensor(mtts_vocos* v, const char* key, const float* h_data, int64_t numel);
int64_t mtts_vocos_weights_bytes(mtts_vocos* v);
int mtts_vocos_upload_weights(mtts_vocos* v, void* d_weights, int64_t bytes);
int64_t mtts_vocos_workspace_bytes(mtts_vocos* v, int B, int T);
int mtts_vocos_decode(mtts_vocos* v, const float* d_mel, int B, int T, float* d_audio, void* d_ws, int64_t ws_bytes,
                      void* stream);

/* ---------------------------------------------------------------- arithmetic and its range guard */

/* Range guard of the default arithmetic.  The fp16 two-term split represents an operand x as h + l / 2^11 with h = fp16(x):
 * beyond +-65504 h saturates and the product is wrong.  Every kernel that splits an operand (the P16 image writers: GEMM
 * epilogues, attention, GroupNorm-apply, the state conversion; the fp32-operand GEMM while staging) ORs 1 into the FIRST
 * 32-bit WORD OF THE WORKSPACE of the call it belongs to (mtts_text_encoder_forward / mtts_decoder_forward / mtts_cfm_solve*
 * clear it on entry): read it back after the call; non-zero = rerun on a context whose arithmetic has the fp32 range
 * (mtts_set_arithmetic(ctx, 6): three bf16 terms).  The Python mirror does this in synthesise().
 * mtts_set_arithmetic: products per fp32 multiply-accumulate as for mtts_gemm_f32's `terms` (0, 2, 3, 6; 1 = the opt-in fp16
 * mode on P16 images), overriding MTTS_GEMM_TERMS; call before mtts_weights_bytes / mtts_upload_weights (it invalidates the
 * packed image).  16 = the 16-BIT STORAGE MODE of BASELINE config #3 (the arithmetic torch.autocast gives the reference on its
 * GPU, reference matcha/inference.py:238): every activation image of the estimator and its weight planes are single fp16
 * planes (2 bytes per element, half the operand traffic), one MFMA per multiply-accumulate, fp32 accumulation, fp32
 * GroupNorm / LayerNorm statistics and ODE state; the text encoder and duration predictor keep the fp32-equivalent split
 * (durations must not move).  Not inside the 1e-3 bar: its measured mel error is reported beside its throughput.
 * 17 = the same mode with BFLOAT16 planes (the dtype BASELINE config #3 names): same layout and traffic, bf16 MFMAs, the fp32
 * exponent range (no saturation, the range guard stays silent), 8 significand bits per operand instead of 11 -- the reference's
 * own bf16 autocast deviates ~8x more from its fp32 mel than its fp16 autocast (tests/golden/prod_autocast.npz).
 * mtts_weights_saturate: 1 if a WEIGHT exceeds the fp16 range in the fp16-split mode (decided while packing). */
int mtts_set_arithmetic(mtts_ctx* ctx, int terms);
int mtts_weights_saturate(mtts_ctx* ctx);

/* ---------------------------------------------------------------- measurement */

/* GEMM arithmetic of a context (NULL: the library default): 0 native fp32 MFMA, 2 fp16 two-term split (default),
 * 6 bf16 three-term split, 3 bf16 two-term split -- see mtts_gemm_f32; 1 / 16 / 17 = the 16-bit modes of mtts_set_arithmetic. */
int mtts_gemm_terms(mtts_ctx* ctx);

/* Per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline line).
 * Classes: 0 gemm, 1 attention, 2 norm/activation/elementwise. */
int mtts_prof_enable(mtts_ctx* ctx, int on);
int mtts_prof_reset(mtts_ctx* ctx);
/* Synchronises the recorded events; returns launches, summed milliseconds, algorithmic FLOPs and compulsory HBM
 * bytes (every operand and result element once) of a class. */
int mtts_prof_read(mtts_ctx* ctx, int klass, int64_t* launches, double* ms, double* flops, double* bytes);
/* The same records one by one, in launch order: h_out[4 i ..] = (class, ms, flops, bytes); returns the count (<= max_records). */
int64_t mtts_prof_records(mtts_ctx* ctx, double* h_out, int64_t max_records);
/* the kernel instantiation of each of those records, in the same order, '\n'-separated ("-" = untagged): the names rocprofv3
 * prints, e.g. "gemm_p16_kernel<64, false, 3, 0, true, false, 2>"; returns the record count */
int64_t mtts_prof_tags(mtts_ctx* ctx, char* out, int64_t max_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MTTS_H */
