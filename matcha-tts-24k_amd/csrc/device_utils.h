// Device helpers shared by the GEMM kernels (gemm_f32.hip, gemm_p16.hip), attention and the normalisation kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace mtts {

constexpr float F16_RES_SCALE = 2048.0f;   // fp16 two-term split: the residual is stored times 2^11

// x ~ h + l / 2^11 with h = fp16(x) (saturating) and l the scaled fp16 residual: 22 significand bits
// (the residual is taken from the CLAMPED value: |xc - h| <= ulp(h)/2, so it needs no clamp of its own; a saturated operand is
// reported through the range flag, device_utils.h out_of_f16_range)
__device__ __forceinline__ void split_f16(float x, _Float16& h, _Float16& l) {
    const float xc = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    h = (_Float16)xc;
    l = (_Float16)((xc - (float)h) * F16_RES_SCALE);
}
__device__ __forceinline__ void split_f16(float x, float lscale, _Float16& h, _Float16& l) {
    const float xc = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    h = (_Float16)xc;
    l = (_Float16)((xc - (float)h) * lscale);
}

// bfloat16 planes (GemmArgs::bf16): two values -> one packed word (v_cvt_pk_bf16_f32, round to nearest even), and back
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {
    const bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ void unpack_bf16(unsigned int w, float& a, float& b) {
    a = __builtin_bit_cast(float, w << 16);
    b = __builtin_bit_cast(float, w & 0xffff0000u);
}

// n / d by a host-made reciprocal (rcp = floor(2^32 / d) + 1; exact while n * d < 2^32).  rcp == 0 (wave-uniform) means "divide":
// d <= 1, or a launch whose quotients could leave that range (the launcher decides, gemm_p16.hip rcp32)
__device__ __forceinline__ int fdiv(int n, int d, unsigned int rcp) { return rcp ? (int)__umulhi((unsigned int)n, rcp) : (d <= 1 ? n : n / d); }

// Range guard of the fp16 split: an operand beyond +-65504 saturates (h clamps), which the caller must learn about.  Producers
// OR their lanes' findings into a register and raise the sticky flag once per thread (atomics only on the rare bad path).
__device__ __forceinline__ bool out_of_f16_range(float a, float b, float c, float d) {
    return fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))) > 65504.f;
}
__device__ __forceinline__ void raise_range_flag(unsigned int* flag, bool bad) {
    if (bad && flag) atomicOr(flag, 1u);
}

// sin(y)^2 for the SnakeBeta epilogue: 3-constant Cody-Waite reduction by pi/2 to r in [-pi/4, pi/4] and the minimax sine
// kernel; in odd quadrants sin(y)^2 = cos(r)^2 = 1 - sin(r)^2, so one polynomial serves both (sin(r)^2 <= 1/2: the
// subtraction is benign).  ~1 ulp of sinf(y)^2 for |y| < 1e4 (arguments here are O(10)), a quarter of the library sinf.
// sin(y)^2 = (1 - cos(2y)) / 2 with the hardware cosine (v_cos_f32 takes revolutions; v_fract_f32 reduces the argument first):
// 4 VALU instructions instead of the ~16 of the polynomial below.  The FeedForward's first projection spends half of its
// workgroup lifetime in this epilogue (profiles/r02_kstamp.log).  Absolute error of v_cos_f32 ~1e-6, i.e. ~5e-7 on sin^2 (the
// polynomial: ~1e-7): the mel error against the goldens is unchanged (5.6e-5 / 3.4e-5 / 4.1e-5) and the
// kernel test against fp64 holds its 1e-5.  -DMTTS_SNAKE_POLY builds the polynomial (3-constant Cody-Waite + minimax sine).
__device__ __forceinline__ float sin_sq_hw(float y) {
    const float t = __builtin_amdgcn_fractf(y * 0.31830988618379067154f);     // 2y / (2 pi), reduced to [0, 1)
    return fmaf(-0.5f, __builtin_amdgcn_cosf(t), 0.5f);
}
// x + s1 sin(s0 x)^2 with the identity folded in: (x + s1/2) - (s1/2) cos(2 s0 x) -- an add and an FMA behind the cosine instead of
// FMA, multiply, add (the epilogue of the FeedForward's first projection is vector-issue bound: DESIGN.md section 5)
__device__ __forceinline__ float snake_hw(float x, float s0, float s1_half) {
    const float t = __builtin_amdgcn_fractf((x * s0) * 0.31830988618379067154f);
    return __builtin_fmaf(-s1_half, __builtin_amdgcn_cosf(t), x + s1_half);
}
__device__ __forceinline__ float sin_sq_poly(float y);
__device__ __forceinline__ float sin_sq(float y) {
#ifdef MTTS_SNAKE_POLY
    return sin_sq_poly(y);
#else
    return sin_sq_hw(y);
#endif
}
__device__ __forceinline__ float sin_sq_poly(float y) {
    const float n = rintf(y * 0.63661977236758134308f);
    float r = fmaf(n, -1.5707962513e+00f, y);       // pi/2 split: hi, mid, lo
    r = fmaf(n, -7.5497894159e-08f, r);
    r = fmaf(n, -5.3903029534e-15f, r);
    const float z = r * r;
    // sin(r) = r + r*z*(S1 + z*(S2 + z*(S3 + z*S4)))
    const float sp = fmaf(z, fmaf(z, fmaf(z, 2.7183114939e-06f, -1.9839334836e-04f), 8.3333298564e-03f), -1.6666665459e-01f);
    const float sn = fmaf(r * z, sp, r);
    const float s2 = sn * sn;
    return (((int)n) & 1) ? 1.0f - s2 : s2;
}

// Mish(x) = x tanh(softplus(x)) = x w / (w + 2), w = e^x (e^x + 2)   (reference decoder.py:32-45, nn.Mish)
// Hardware exponential and reciprocal (v_exp_f32 on x log2 e, v_rcp_f32: ~1 ulp each, relative error of Mish ~3e-7) instead of
// expf and an IEEE division: ~8 instead of ~30 vector instructions per element.  The fused Block1D tail of the ResNet GEMM
// runs this on 16-32 elements per lane with one wave per SIMD (profiles/r02_kstamp_insitu.log: its first 16-row chunk cost
// 8-10k cycles); mel error against the goldens unchanged.  -DMTTS_MISH_LIBM builds the libm form.
__device__ __forceinline__ float mish_f(float x) {
    if (x > 20.f) return x;
#ifdef MTTS_MISH_LIBM
    const float n = expf(x);
    const float w = n * (n + 2.f);
    return x * (w / (w + 2.f));
#else
    const float n = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
    const float w = n * (n + 2.f);
    return x * (w * __builtin_amdgcn_rcpf(w + 2.f));
#endif
}

__device__ __forceinline__ float act_apply(float c, int act, float p0, float p1) {
    switch (act) {
        case ACT_RELU: return c > 0.f ? c : 0.f;
        case ACT_SILU: return c / (1.0f + expf(-c));
        case ACT_SNAKE: {   // reference transformer.py:75: x + 1/(beta+1e-9) * sin(x*alpha)^2
            return c + p1 * sin_sq(c * p0);
        }
        case ACT_GELU: return 0.5f * c * (1.0f + erff(c * 0.70710678118654752440f));   // exact GELU (vocos ConvNeXtBlock)
        default: return c;
    }
}


// DPP all-reduce sums (every lane gets the total): 16-lane rows via row_mirror, row_half_mirror, quad reverse, quad swap;
// 8-lane groups skip the first step.  One VALU instruction per step instead of a ds_bpermute round trip.
#define MTTS_DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true)))
__device__ __forceinline__ float allreduce16(float v) {
    v = MTTS_DPP_ADD(v, 0x140);   // row_mirror
    v = MTTS_DPP_ADD(v, 0x141);   // row_half_mirror
    v = MTTS_DPP_ADD(v, 0x1B);    // quad_perm [3,2,1,0]
    v = MTTS_DPP_ADD(v, 0xB1);    // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ float allreduce8(float v) {
    v = MTTS_DPP_ADD(v, 0x141);
    v = MTTS_DPP_ADD(v, 0x1B);
    v = MTTS_DPP_ADD(v, 0xB1);
    return v;
}


}  // namespace mtts
