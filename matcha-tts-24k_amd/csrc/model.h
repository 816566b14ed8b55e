// Host-side model of the path: packed weights, workspace planning and the launch sequences of
// TextEncoder.forward, Decoder.forward and BASECFM.solve (reference files cited at each function in model.hip).
#pragma once
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "../../include/mtts.h"
#include "kernels.h"

namespace mtts {

void set_error(const std::string& msg);
const char* get_error();

// One GEMM-ready weight panel inside the device image (offsets in floats).
struct Panel {
    size_t w = 0, b = 0;
    size_t w16 = 0;            // bf16 split planes [3][Np][Kp] (offset in floats), present when the context uses a split mode
    bool has_bias = false;
    size_t wsum = 0;           // [Np] row sums of the panel (fp16-split mode): LayerNorm in the P16 GEMM's epilogue
    size_t wh16 = 0;           // 16-bit storage mode: the fp16 head plane alone [Np][Kp] halves (offset in floats)
    int N = 0, C = 0, ntaps = 1, ktap = 0;
};
struct Vec { size_t off = 0; int n = 0; };

struct ResnetW {
    Panel conv1, conv2, res;
    Vec gn1_g, gn1_b, gn2_g, gn2_b;
    Vec gn1_bs, gn2_bs;       // per GroupNorm group (mean, sum of squared deviations) of conv1's / conv2's bias row (folded padding)
    int cin = 0, cout = 0;
    int tb_off = 0;           // column offset of this block's time bias inside the per-evaluation bias row
};
struct TBlockW {
    Panel qkv, out, ff1, ff2;  // LayerNorm affines folded into qkv / ff1
    Vec alpha_exp, inv_beta;
    // fragment stream of the block's row-local chain (kernels.h ChainArgs; tblock_chain.hip): out-projection, FeedForward and --
    // when another block of the same run follows -- that block's q|k|v projection.  0 frags = chain not packed for this block.
    size_t chain = 0;          // offset in the image (floats)
    long chain_frags = 0;
    size_t chain_consts = 0;   // the chain kernel's column constants as one block of 18 C floats (kernels.h ChainArgs::consts)
    int chain_ch = 0;          // hidden chunk the stream was packed for
    int chain_nqkv = 0;        // width of the q|k|v part (0: none)
    int next = -1;             // index of the block whose q|k|v the chain computes
    size_t chain_pair = 0;     // the same chain as TWO half streams per wave (kernels.h ChainArgs::pair), offset in floats
    long chain_pair_frags = 0; // fragments per (half, wave); 0 = not packed
};
struct DecW {
    Vec freqs;
    Panel t1, t2, tmlp;        // time MLP and the concatenated per-ResNet Linear(Mish(t))
    int tb_total = 0;
    std::vector<ResnetW> res;            // down..., mid..., up... in execution order
    std::vector<TBlockW> tb;             // n_blocks per resnet, execution order
    std::vector<Panel> down;             // per level: stride-2 conv (or k3 conv at the last level)
    std::vector<Panel> up_even, up_odd;  // ConvTranspose phases (levels-1 entries)
    Panel up_last;                       // k3 conv of the last up block
    Panel final_conv, final_proj;
    Vec fgn_g, fgn_b, fgn_bs;
};
struct EncW {
    Vec emb, spk_enc, spk_dur, rope_cos, rope_sin;
    std::vector<Panel> pre_conv;
    std::vector<Vec> pre_g, pre_b;
    Panel pre_proj;
    std::vector<Panel> qkv, o, ffn1, ffn2;
    std::vector<Vec> n1_g, n1_b, n2_g, n2_b;
    Panel pm0, pm2;
    Panel film;
    std::vector<Panel> dp_conv;
    std::vector<Vec> dp_g, dp_b;
    Panel dp_proj;
};

struct VocosW {
    Panel embed, head, basis;
    Vec norm_g, norm_b, fin_g, fin_b, window;
    std::vector<Vec> dw_w, dw_b, ln_g, ln_b;
    std::vector<Panel> pw1, pw2;
};

struct ProfRec { hipEvent_t e0, e1; int klass; double flops, bytes; std::string tag; };      // tag: a copy (launchers reuse their buffers)

}  // namespace mtts

struct mtts_ctx {
    mtts_config cfg;
    std::map<std::string, std::vector<float>> raw;
    std::vector<float> image;     // host staging of the packed device image
    float* d_image = nullptr;     // caller-owned device buffer
    bool packed = false, uploaded = false;
    int gemm_terms = 6;           // 0: fp32 MFMA, 6 / 3: split-bf16 MFMA, 2: split-fp16 (MTTS_GEMM_TERMS; see gemm_f32.hip)
    unsigned int* cur_flag = nullptr;   // range flag of the call being enqueued: first word of its workspace (include/mtts.h)
    bool weights_saturate = false;      // fp16-split mode: a weight beyond +-65504 was met while packing
    const int* d_tlen = nullptr;  // per-utterance frame limits of the next estimator calls (mtts_set_frame_limits), device [B]
    bool half16 = false;          // 16-bit storage mode (mtts_set_arithmetic(ctx, 16) / MTTS_GEMM_TERMS=16): the estimator's images are
                                  // single fp16 planes, one MFMA per MAC (BASELINE config #3); everything else as for terms 2
    bool bf16 = false;            // ... with bfloat16 planes (mtts_set_arithmetic(ctx, 17) / MTTS_GEMM_TERMS=17; half16 is set as well)
    bool half_now = false;        // set while the estimator's launches are being enqueued in that mode
    bool fast16 = false;          // MTTS_GEMM_TERMS=1 at mtts_create: the estimator's P16 kernels multiply the fp16 heads only
    bool p16_on = true;           // fp16-split mode: activations as P16 images between kernels (MTTS_P16=0 at mtts_create disables)
    bool chain_on = true;         // transformer blocks' row-local part as one launch (tblock_chain.hip; MTTS_CHAIN=0 at mtts_create disables)
    int chain_ch = 256;           // hidden chunk of the chain's FeedForward at width 384 (MTTS_CHAIN_CH at mtts_create: 128 / 256)
    int chain_qb = 0;             // rows per workgroup (MTTS_CHAIN_QB at mtts_create; 0 = by shape)
    bool pair_on = true;          // pair form of the chain launch for levels below chain_min_rows (MTTS_CHAIN_PAIR=0 at mtts_create disables)
    unsigned int pair_epoch = 0;  // flag value of the latest pair launch (unique per launch)
    int chain_min_rows = 5761;    // (= where the pair form's residency bound, 120 tiles of 48 rows, ends) estimator rows (B * T of a level) from which the chain replaces the four GEMM launches (MTTS_CHAIN_MIN_ROWS):
                                  // measured at width 384 -- 10304 rows: 138 vs ~160 us per block; 5152 rows: 100 vs ~92 us (profiles/r03_chain_*)
    mtts::DecW dec;
    mtts::EncW enc;
    // one thread at a time: the path's entry points hold this while they enqueue (per-call state above: cur_flag, half_now,
    // d_tlen, prof); a second thread's call fails instead of interleaving its launches with another call's flag pointer
    std::atomic<bool> in_use{false};
    // profiling
    bool prof_on = false;
    std::vector<mtts::ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
};

// Vocos-24k head: its own weight image and context (the reference loads it as a separate object,
// reference matcha/inference.py:223-231).  `base` carries the tensor registry, the packed image and the profiler.
struct mtts_vocos {
    mtts_ctx base;
    int n_mels = 100, dim = 512, inter = 1536, layers = 8, n_fft = 1024, hop = 256;
    int ld_spec = 0, im_off = 0;     // head output row: [Re/logmag 0..n_fft/2 | pad | Im/phase at im_off.. | pad]
    mtts::VocosW w;
};
