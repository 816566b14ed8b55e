// Normalisation, activation and data-movement kernels of the path (gfx950).  All are HBM/L2-bound streaming
// kernels over channels-last activations [rows, C]: float4 per lane, rows assigned to waves, no atomics, and every
// reduction has a fixed order (results are bitwise reproducible run to run).
#include "kernels.h"
#include "device_utils.h"

namespace mtts {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Mish(x) = x * tanh(softplus(x)), softplus threshold 20 (torch.nn.Mish; reference decoder.py:40,51).
// tanh(log(1+e^x)) = (n^2 + 2n) / (n^2 + 2n + 2) with n = e^x: no cancellation for negative x.
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per row; the row lives in registers (C <= 2048), two-pass mean / centred variance.
constexpr int LN_MAXV = 8;   // float4 per lane

__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int M, int C, int ld, float eps,
                                                        float* __restrict__ mean, float* __restrict__ rstd) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * ld;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < C) { v[i] = *reinterpret_cast<const f32x4*>(xr + c); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < C) {
            const f32x4 d = v[i] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float var = wave_sum(q) / (float)C;
    if (lane == 0) { mean[row] = mu; rstd[row] = 1.0f / sqrtf(var + eps); }
}

hipError_t launch_row_stats(const float* x, int M, int C, int ld, float eps, float* mean, float* rstd, hipStream_t s) {
    if (!x || !mean || !rstd || M <= 0 || C <= 0 || (C & 3) || (ld & 3) || C > 64 * 4 * LN_MAXV) return hipErrorInvalidValue;
    hipLaunchKernelGGL(row_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, M, C, ld, eps, mean, rstd);
    return hipGetLastError();
}

// Channel LayerNorm of the text encoder (reference text_encoder.py:19-27) + what follows it at each call site:
// SiLU (prenet :58-60), FiLM (duration predictor :107-109), sequence mask.
__global__ __launch_bounds__(256) void layernorm_kernel(const LayerNormArgs p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= p.M) return;
    const float* xr = p.x + (size_t)row * p.ldx;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < p.C) { v[i] = *reinterpret_cast<const f32x4*>(xr + c); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
    const float mu = wave_sum(s) / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < p.C) {
            const f32x4 d = v[i] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)p.C + p.eps);
    const float mk = p.mask ? p.mask[row] : 1.0f;
    const float* film = p.film ? p.film + (size_t)(row / p.T) * 2 * p.C : nullptr;
    float* yr = p.y + (size_t)row * p.ldy;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < p.C) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + c);
            f32x4 o = ((v[i] - mu) * rs) * g + bt;
            if (p.act == ACT_SILU) { for (int e = 0; e < 4; ++e) o[e] = silu_f(o[e]); }
            if (film) {
                const f32x4 fg = *reinterpret_cast<const f32x4*>(film + c);
                const f32x4 fb = *reinterpret_cast<const f32x4*>(film + p.C + c);
                o = o * fg + fb;
            }
            if (p.mask) o *= mk;
            *reinterpret_cast<f32x4*>(yr + c) = o;
        }
    }
}

hipError_t launch_layernorm(const LayerNormArgs& a, hipStream_t s) {
    if (!a.x || !a.y || !a.gamma || !a.beta || a.M <= 0 || a.C <= 0 || (a.C & 3) || (a.ldx & 3) || (a.ldy & 3) ||
        a.C > 64 * 4 * LN_MAXV || a.T <= 0)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ GroupNorm + Mish
// Block1D (reference decoder.py:32-45): statistics per (batch, group) over C/G channels x ALL T frames (padding included).
// Pass 1: per chunk of GN_CHUNK rows, (mean, M2) about the chunk mean.  Pass 2 merges the chunk moments in chunk order
// (Chan et al.) and applies GN affine -> Mish -> mask [-> + time bias -> mask] [-> + residual].
// Block shape: x = C/4 float4 columns, y = RT row lanes.
__global__ void gn_partial_kernel(const float* __restrict__ y, int T, int C, int G, float* __restrict__ partial,
                                  const int* __restrict__ nrows, int chunk_rows) {
    extern __shared__ float red[];                 // [blockDim.x*blockDim.y] + [G]
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int c4 = threadIdx.x, ry = threadIdx.y, RT = blockDim.y, nth = blockDim.x * blockDim.y;
    const int tid = ry * blockDim.x + c4;
    const int t0 = chunk * chunk_rows;
    const int Tb = nrows ? min(T, nrows[b]) : T;               // this utterance's own rows (per-request / folded padding)
    const int rows = max(0, min(chunk_rows, Tb - t0));
    const int cpg4 = (C / G) / 4;                  // float4 columns per group
    const float* base = y + ((size_t)b * T + t0) * C + c4 * 4;
    float* gmean = red + nth;
    if (rows == 0) {                               // chunk beyond the utterance: an empty partial, skipped by the merge
        if (tid < G) {
            float* o = partial + (((size_t)b * nchunks + chunk) * G + tid) * 2;
            o[0] = 0.f;
            o[1] = 0.f;
        }
        return;
    }

    // A thread's share of the chunk (<= KEEP rows of 4 channels) stays in registers between the two passes: one trip to
    // memory, every load in flight at once, and still the exact two-pass (mean, then squared deviations) arithmetic.
    constexpr int KEEP = 16;
    const bool in_regs = (chunk_rows + RT - 1) / RT <= KEEP;       // block-uniform
    f32x4 keep[KEEP];
    float s = 0.f;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int r = ry + i * RT;
            keep[i] = r < rows ? *reinterpret_cast<const f32x4*>(base + (size_t)r * C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < KEEP; ++i) s += (keep[i][0] + keep[i][1]) + (keep[i][2] + keep[i][3]);
    } else {
#pragma unroll 4
        for (int r = ry; r < rows; r += RT) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)r * C);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    red[tid] = s;
    __syncthreads();
    if (tid < G) {
        float a = 0.f;
        for (int r = 0; r < RT; ++r)
            for (int k = 0; k < cpg4; ++k) a += red[r * blockDim.x + tid * cpg4 + k];
        gmean[tid] = a / (float)(rows * (C / G));
    }
    __syncthreads();
    const float mu = gmean[c4 / cpg4];
    float q = 0.f;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            if (ry + i * RT < rows) {
                const f32x4 d = keep[i] - mu;
                q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        }
    } else {
#pragma unroll 4
        for (int r = ry; r < rows; r += RT) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(base + (size_t)r * C) - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    red[tid] = q;
    __syncthreads();
    if (tid < G) {
        float a = 0.f;
        for (int r = 0; r < RT; ++r)
            for (int k = 0; k < cpg4; ++k) a += red[r * blockDim.x + tid * cpg4 + k];
        float* o = partial + (((size_t)b * nchunks + chunk) * G + tid) * 2;
        o[0] = gmean[tid];
        o[1] = a;
    }
}

static inline bool gn_shape_ok(int C, int G) { return C > 0 && G > 0 && C % G == 0 && ((C / G) & 3) == 0 && C / 4 <= 1024 && G <= 64; }
static inline dim3 gn_block(int C) {
    const int c4 = C / 4;
    int rt = 256 / c4;
    if (rt < 1) rt = 1;
    if (rt > GN_CHUNK) rt = GN_CHUNK;
    return dim3(c4, rt);
}

hipError_t launch_gn_partial(const float* y, int B, int T, int C, int G, float* partial, hipStream_t s, const int* nrows) {
    if (!y || !partial || B <= 0 || T <= 0 || !gn_shape_ok(C, G)) return hipErrorInvalidValue;
    const dim3 blk = gn_block(C);
    const size_t lds = (size_t)(blk.x * blk.y + G) * sizeof(float);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(gn_chunks(B, T), B), blk, lds, s, y, T, C, G, partial, nrows, gn_chunk_rows(B, T));
    return hipGetLastError();
}

#define GN_DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true)))
__device__ __forceinline__ float gn_allreduce16(float v) {   // all-reduce over a 16-lane DPP row
    v = GN_DPP_ADD(v, 0x140);
    v = GN_DPP_ADD(v, 0x141);
    v = GN_DPP_ADD(v, 0x1B);
    v = GN_DPP_ADD(v, 0xB1);
    return v;
}

__global__ void gn_apply_kernel(const GnApplyArgs p) {
    __shared__ float smean[64], srstd[64];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int c4 = threadIdx.x, ry = threadIdx.y, RT = blockDim.y;
    const int tid = ry * blockDim.x + c4;
    const int cpg = p.C / p.G;
    // Merge the chunk moments (Chan et al.): L lanes per group, each folds its strided share of the chunks in chunk order,
    // then a fixed butterfly over the L lanes -- deterministic, and a handful of dependent loads instead of nchunks of them
    // (every workgroup repeats this merge; at B = 1 there are 80 chunks).
    {
        const int nth = blockDim.x * blockDim.y;
        int L = 32;
        while (L > 1 && L * p.G > nth) L >>= 1;
        const int g = tid / L, j = tid - g * L;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        if (g < p.G && p.tile_stats) {
            // entries left by the producing GEMM's epilogue: per wave tile (tile_rows rows x 64 columns) and group slice
            const int ncw = p.C >> 6, R = p.tile_rows;
            const int t_first = (b * p.T) / R, nrw = ((b + 1) * p.T - 1) / R - t_first + 1;           // wave tiles touching utterance b
            const int w_lo = (g * cpg) >> 6, w_hi = ((g + 1) * cpg - 1) >> 6, nw = w_hi - w_lo + 1;
            // entries in batches of 4 per lane: a batch's loads are independent (one round trip), the merge order is fixed
            const int total = nrw * nw;
            for (int k0 = j; k0 < total; k0 += L * 4) {
                f32x4 q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + L * e;
                    const int kc = k < total ? k : j;
                    const int rw = kc / nw, w = w_lo + (kc - rw * nw), tile = t_first + rw;
                    const int part = (tile * R) / p.T == b ? 0 : 1;
                    const int slice = g - (w * 64) / cpg;
                    q[e] = *reinterpret_cast<const f32x4*>(p.tile_stats + ((size_t)(((size_t)tile * 2 + part) * ncw + w) * 2 + slice) * 4);
                    if (k >= total) q[e][0] = 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float nb = q[e][0];
                    if (nb <= 0.f) continue;
                    const float delta = q[e][1] - mean;
                    const float nt = n + nb;
                    mean += delta * (nb / nt);
                    m2 += q[e][2] + delta * delta * (n * nb / nt);
                    n = nt;
                }
            }
        } else if (g < p.G) {
            const int Tb = p.nrows ? min(p.T, p.nrows[b]) : p.T;
            for (int k = j; k < nchunks; k += L) {
                const int rows_k = min(p.chunk_rows, Tb - k * p.chunk_rows);
                if (rows_k <= 0) break;
                const float* q = p.partial + (((size_t)b * nchunks + k) * p.G + g) * 2;
                const float nb = (float)(rows_k * cpg);
                const float delta = q[0] - mean;
                const float nt = n + nb;
                mean += delta * (nb / nt);
                m2 += q[1] + delta * delta * (n * nb / nt);
                n = nt;
            }
        }
        for (int off = 1; off < L; off <<= 1) {        // lanes of a group are contiguous and L divides 64
            const float n2 = __shfl_xor(n, off), mean2 = __shfl_xor(mean, off), m22 = __shfl_xor(m2, off);
            const float nt = n + n2;
            if (nt > 0.f) {
                const float delta = mean2 - mean;
                // symmetric form: both partners compute the same merged triple
                const float w2 = n2 / nt;
                const float merged_mean = (j & off) ? mean2 + (mean - mean2) * (n / nt) : mean + delta * w2;
                m2 = m2 + m22 + delta * delta * (n * w2);
                mean = merged_mean;
                n = nt;
            }
        }
        if (g < p.G && j == 0) {
            if (p.nextra) {                            // folded padding: nextra copies of the conv's bias row, in closed form
                const float ne = (float)p.nextra[b];
                if (ne > 0.f) {
                    const float nb = ne * (float)cpg, delta = p.bias_stats[2 * g] - mean, nt = n + nb;
                    mean += delta * (nb / nt);
                    m2 += ne * p.bias_stats[2 * g + 1] + delta * delta * (n * nb / nt);
                    n = nt;
                }
            }
            smean[g] = mean;
            srstd[g] = 1.0f / sqrtf(m2 / n + p.eps);
        }
    }
    __syncthreads();
    const int g = (c4 * 4) / cpg;
    const float mu = smean[g], rs = srstd[g];
    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c4 * 4);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + c4 * 4);
    f32x4 cb = {0.f, 0.f, 0.f, 0.f};
    if (p.chbias) cb = *reinterpret_cast<const f32x4*>(p.chbias + c4 * 4);
    const int t0 = chunk * p.chunk_rows;
    const int rows = min(p.chunk_rows, p.T - t0);
    // U rows per pass, every load of the pass issued before the first use (the loop is a pure stream: without this each
    // thread had one 16-byte load in flight and the kernel ran at half the HBM rate)
    constexpr int U = 4;
    using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
    bool range_bad = false;
    for (int r0 = ry; r0 < rows; r0 += RT * U) {
        size_t row[U];
        bool ok[U];
        f32x4 v[U], rs4[U];
        float mk[U], m16[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * RT;
            ok[u] = r < rows;
            row[u] = (size_t)b * p.T + t0 + (ok[u] ? r : rows - 1);
            v[u] = *reinterpret_cast<const f32x4*>(p.y + row[u] * p.C + c4 * 4);
            mk[u] = p.mask[row[u]];
            m16[u] = p.out16_mask ? p.out16_mask[row[u]] : 1.0f;
            if (p.res) rs4[u] = *reinterpret_cast<const f32x4*>(p.res + row[u] * p.ldr + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 o = ((v[u] - mu) * rs) * gm + bt;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = mish_f(o[e]) * mk[u];
            if (p.chbias) o = (o + cb) * mk[u];
            if (p.res) o += rs4[u];
            if (!ok[u]) continue;                  // a whole 16-lane DPP row shares ry, so the reductions below stay uniform
            if (p.out) *reinterpret_cast<f32x4*>(p.out + row[u] * p.C + c4 * 4) = o;
            if (p.out16) {       // P16 image for the next GEMM's LDS-DMA (gemm_p16.hip)
                f16x4 hh, ll;
                range_bad |= out_of_f16_range(o[0], o[1], o[2], o[3]) && m16[u] != 0.f && !p.bf16;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a, b;
                    split_f16(o[e] * m16[u], a, b);
                    hh[e] = a;
                    ll[e] = b;
                }
                if (p.half16 && p.bf16) {
                    using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
                    *reinterpret_cast<u32x2*>(p.out16 + row[u] * (size_t)p.ld16 + c4 * 4) =
                        u32x2{pack_bf16(o[0] * m16[u], o[1] * m16[u]), pack_bf16(o[2] * m16[u], o[3] * m16[u])};
                } else if (p.half16) {
                    *reinterpret_cast<f16x4*>(p.out16 + row[u] * (size_t)p.ld16 + c4 * 4) = hh;
                } else {
                    _Float16* o16 = p.out16 + row[u] * (size_t)p.ld16 + (c4 >> 3) * 64 + (c4 & 7) * 4;
                    *reinterpret_cast<f16x4*>(o16) = hh;
                    *reinterpret_cast<f16x4*>(o16 + 32) = ll;
                }
            }
            if (p.stats_out) {   // LayerNorm partial moments of the row's 64-column slices (16 threads = one DPP row each), as
                                 // the GEMM epilogue leaves them: the first transformer block needs no row_stats pass
                const float mean = gn_allreduce16((o[0] + o[1]) + (o[2] + o[3])) * (1.0f / 64.0f);
                const f32x4 d = o - mean;
                const float m2 = gn_allreduce16((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
                if ((c4 & 15) == 0) {
                    float* so = p.stats_out + (row[u] * (size_t)(p.C >> 6) + (c4 >> 4)) * 2;
                    so[0] = mean;
                    so[1] = m2;
                }
            }
        }
    }
    raise_range_flag(p.range_flag, range_bad);
}

hipError_t launch_gn_apply(const GnApplyArgs& a, hipStream_t s) {
    if (!a.y || (!a.partial && !a.tile_stats) || !a.gamma || !a.beta || !a.mask || (!a.out && !a.out16) || a.B <= 0 || a.T <= 0 || !gn_shape_ok(a.C, a.G) ||
        (a.res && (a.ldr & 3)))
        return hipErrorInvalidValue;
    // stats_out needs whole 16-thread DPP rows per 64-column slice and one thread row per wave-aligned offset
    if (a.stats_out && ((a.C & 63) || ((a.C / 4) & 15))) return hipErrorInvalidValue;
    if (a.out16 && ((a.C & 31) || a.ld16 < (a.half16 ? 1 : 2) * a.C || (a.ld16 & 3))) return hipErrorInvalidValue;
    if (a.tile_stats && (a.tile_rows <= 0 || a.T < a.tile_rows || (a.C & 63) || (a.C / a.G) < 32)) return hipErrorInvalidValue;
    if ((a.nextra != nullptr) != (a.bias_stats != nullptr)) return hipErrorInvalidValue;
    GnApplyArgs b = a;
    b.chunk_rows = gn_chunk_rows(a.B, a.T);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gn_chunks(a.B, a.T), a.B), gn_block(a.C), 0, s, b);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ layout moves
// dst[b, t, col_off + c] = src[b, c, t] (+ add[b, c, t])        [B,C,T] -> rows of [B*T, ld]
__global__ void cf_to_cl_kernel(const float* __restrict__ src, const float* __restrict__ add, int C, int T, int T_src,
                                float* __restrict__ dst, int ld, int col_off) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, t = t0 + tx;
        float v = 0.f;
        if (c < C && t < T) {
            const size_t i = ((size_t)b * C + c) * T_src + t;
            v = src[i];
            if (add) v += add[i];
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + ty + 8 * k, c = c0 + tx;
        if (c < C && t < T) dst[((size_t)b * T + t) * ld + col_off + c] = tile[tx][ty + 8 * k];
    }
}
hipError_t launch_cf_to_cl(const float* src, const float* add, int B, int C, int T, float* dst, int ld, int col_off, hipStream_t s,
                           int T_src) {
    if (T_src == 0) T_src = T;
    if (!src || !dst || B <= 0 || C <= 0 || T <= 0 || T_src < T) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cf_to_cl_kernel, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(32, 8), 0, s, src, add, C, T, T_src, dst, ld, col_off);
    return hipGetLastError();
}

// dst[b, c, t] = src[b, t, c] * scale + shift, t < T_out        rows of [B*T, ld] -> [B,C,T_out]
__global__ void cl_to_cf_kernel(const float* __restrict__ src, int ld, int C, int T, float* __restrict__ dst, int T_out,
                                float scale, float shift) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (c < C && t < T_out) ? src[((size_t)b * T + t) * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, t = t0 + tx;
        if (c < C && t < T_out) dst[((size_t)b * C + c) * T_out + t] = tile[tx][ty + 8 * k] * scale + shift;
    }
}
hipError_t launch_cl_to_cf(const float* src, int ld, int B, int C, int T, float* dst, int T_out, float scale, float shift, hipStream_t s) {
    if (!src || !dst || B <= 0 || C <= 0 || T <= 0 || T_out <= 0 || T_out > T) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cl_to_cf_kernel, dim3((T_out + 31) / 32, (C + 31) / 32, B), dim3(32, 8), 0, s, src, ld, C, T, dst, T_out, scale, shift);
    return hipGetLastError();
}

__global__ void fill_cols_kernel(float* dst, int M, int ld, int col0, int ncols, float v) {
    const size_t n = (size_t)M * ncols;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[(i / ncols) * ld + col0 + (i % ncols)] = v;
}
hipError_t launch_fill_cols(float* dst, int M, int ld, int col0, int ncols, float v, hipStream_t s) {
    if (ncols <= 0 || M <= 0) return hipSuccess;
    const size_t n = (size_t)M * ncols;
    hipLaunchKernelGGL(fill_cols_kernel, dim3((unsigned)min((n + 255) / 256, (size_t)2048)), dim3(256), 0, s, dst, M, ld, col0, ncols, v);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ time embedding helpers
// SinusoidalPosEmb (reference decoder.py:20-29): out[i] = [sin((scale*t_i)*f_j) | cos((scale*t_i)*f_j)], f from the host table.
__global__ void time_sinusoid_kernel(const float* __restrict__ freqs, const TimeVals tv, int nt, int half, float scale,
                                     float* __restrict__ out) {
    const int i = blockIdx.x;
    const float st = scale * tv.t[i];
    for (int j = threadIdx.x; j < half; j += blockDim.x) {
        const float arg = st * freqs[j];
        out[(size_t)i * 2 * half + j] = sinf(arg);
        out[(size_t)i * 2 * half + half + j] = cosf(arg);
    }
}
hipError_t launch_time_sinusoid(const float* freqs, const TimeVals& tv, int nt, int half, float scale, float* out, hipStream_t s) {
    if (!freqs || !out || nt <= 0 || nt > MAX_EVALS || half <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(time_sinusoid_kernel, dim3(nt), dim3(128), 0, s, freqs, tv, nt, half, scale, out);
    return hipGetLastError();
}

__global__ void unary_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act_mish) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = act_mish ? mish_f(x[i]) : silu_f(x[i]);
}
hipError_t launch_unary(const float* x, float* y, int64_t n, int act_mish, hipStream_t s) {
    if (!x || !y || n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(unary_kernel, dim3((unsigned)min((size_t)(n + 255) / 256, (size_t)2048)), dim3(256), 0, s, x, y, (size_t)n, act_mish);
    return hipGetLastError();
}

// ODE stage combinations of torchdiffeq's fixed-grid rk4 (3/8 rule), written in its operation order:
//   stage 1: y + dt*k1*(1/3)            stage 2: y + dt*(k2 - k1*(1/3))
//   stage 3: y + dt*(k1 - k2 + k3)      stage 4: y + (k1 + 3*(k2+k3) + k4)*dt*0.125
//   stage 0: y + dt*k1  (plain axpy)
__global__ void ode_combine_kernel(int stage, float dt, const float* __restrict__ y, int ldy, const float* __restrict__ k1,
                                   const float* __restrict__ k2, const float* __restrict__ k3, const float* __restrict__ k4,
                                   int ldk, float* __restrict__ out, int ldo, int M, int C) {
    const float third = 1.0f / 3.0f;   // torchdiffeq _one_third, rounded to fp32 when it meets an fp32 tensor
    const size_t n = (size_t)M * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C, c = i % C;
        const float yv = y[r * ldy + c];
        const size_t ik = r * ldk + c;
        float v;
        switch (stage) {
            case 1: v = yv + (dt * k1[ik]) * third; break;
            case 2: v = yv + dt * (k2[ik] - k1[ik] * third); break;
            case 3: v = yv + dt * ((k1[ik] - k2[ik]) + k3[ik]); break;
            case 4: v = yv + (((k1[ik] + 3.0f * (k2[ik] + k3[ik])) + k4[ik]) * dt) * 0.125f; break;
            default: v = yv + dt * k1[ik]; break;
        }
        out[r * ldo + c] = v;
    }
}
hipError_t launch_ode_combine(int stage, float dt, const float* y, int ldy, const float* k1, const float* k2, const float* k3,
                              const float* k4, int ldk, float* out, int ldo, int M, int C, hipStream_t s) {
    if (!y || !k1 || !out || M <= 0 || C <= 0) return hipErrorInvalidValue;
    const size_t n = (size_t)M * C;
    hipLaunchKernelGGL(ode_combine_kernel, dim3((unsigned)min((n + 255) / 256, (size_t)2048)), dim3(256), 0, s, stage, dt, y, ldy, k1,
                       k2, k3, k4, ldk, out, ldo, M, C);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ text-encoder glue
// out[row, :C] = table[ids[row], :] * scale (* mask[row])       nn.Embedding * sqrt(C), reference text_encoder.py:395
__global__ void embedding_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table, int rows, int C, float scale,
                                 const float* __restrict__ mask, float* __restrict__ out, int ld) {
    const int row = blockIdx.x;
    const int64_t id = ids[row];
    const float mk = mask ? mask[row] : 1.0f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = table[(size_t)id * C + c] * scale;
        if (mask) v *= mk;
        out[(size_t)row * ld + c] = v;
    }
}
hipError_t launch_embedding(const int64_t* ids, const float* table, int rows, int C, float scale, const float* mask, float* out, int ld, hipStream_t s) {
    if (!ids || !table || !out || rows <= 0 || C <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embedding_kernel, dim3(rows), dim3(64), 0, s, ids, table, rows, C, scale, mask, out, ld);
    return hipGetLastError();
}

// sequence_mask (reference utils/model.py:7-9) as float 0/1
__global__ void seq_mask_kernel(const int64_t* __restrict__ lengths, int T, float* __restrict__ mask) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) mask[(size_t)b * T + t] = (int64_t)t < lengths[b] ? 1.0f : 0.0f;
}
hipError_t launch_seq_mask(const int64_t* lengths, int B, int T, float* mask, hipStream_t s) {
    if (!lengths || !mask || B <= 0 || T <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(seq_mask_kernel, dim3((T + 255) / 256, B), dim3(256), 0, s, lengths, T, mask);
    return hipGetLastError();
}

__global__ void mask_down_kernel(const float* __restrict__ src, int T_src, int stride, float* __restrict__ dst, int T_dst) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T_dst) dst[(size_t)b * T_dst + t] = src[(size_t)b * T_src + (size_t)t * stride];
}
hipError_t launch_mask_down(const float* src, int B, int T_src, int stride, float* dst, int T_dst, hipStream_t s) {
    if (!src || !dst || B <= 0 || T_dst <= 0 || (size_t)(T_dst - 1) * stride >= (size_t)T_src) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mask_down_kernel, dim3((T_dst + 255) / 256, B), dim3(256), 0, s, src, T_src, stride, dst, T_dst);
    return hipGetLastError();
}

// Per-level frame tables (kernels.h FrameTableArgs): one thread per (utterance, frame of level 0); levels walked in the thread.
__global__ void frame_tables_kernel(const FrameTableArgs p) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int t_true = p.tlen ? p.tlen[b] : p.T_true;
    if (!p.y_len) {                                   // unfolded, per-request padding: only the row counts
        if (t == 0)
            for (int l = 0; l < p.nl; ++l) {
                p.nrows[l][b] = min(p.T[l], t_true >> l);
                p.nextra[l][b] = 0;
            }
        return;
    }
    int L = (int)min((int64_t)p.y_len[b], (int64_t)(1 << 30));
    L = max(L, 0);
    for (int l = 0; l < p.nl; ++l) {
        if (l) L = (L + 1) >> 1;                      // mask[:, :, ::2] of a prefix mask (reference decoder.py:390)
        const int Tl = p.T[l];
        const int Lc = min(L, Tl);                    // (the host sized T[l] >= L + 1 whenever padded frames exist)
        const int npad = max((t_true >> l) - Lc, 0);
        const int rows = min(Lc + (npad > 0 ? 1 : 0), Tl);
        if (t == 0) {
            p.nrows[l][b] = rows;
            p.nextra[l][b] = max(npad - 1, 0);
        }
        if (t < Tl) {
            p.mask[l][(size_t)b * Tl + t] = t < Lc ? 1.0f : 0.0f;
            p.kbias[l][(size_t)b * Tl + t] = t < Lc ? 1.0f : ((t == Lc && npad > 0) ? logf((float)npad) : 0.0f);
        }
    }
}
hipError_t launch_frame_tables(const FrameTableArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.nl < 1 || a.nl > 4 || a.T_true <= 0) return hipErrorInvalidValue;
    for (int l = 0; l < a.nl; ++l) {
        if (a.T[l] <= 0 || !a.nrows[l] || !a.nextra[l]) return hipErrorInvalidValue;
        if (a.y_len && (!a.mask[l] || !a.kbias[l])) return hipErrorInvalidValue;
    }
    if (!a.y_len && !a.tlen) return hipErrorInvalidValue;
    const int T0 = a.y_len ? a.T[0] : 1;
    hipLaunchKernelGGL(frame_tables_kernel, dim3((T0 + 255) / 256, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// dst[b*T + t, col_off + c] = src[b, c] (* mask[b*T+t])     speaker embedding concat, reference text_encoder.py:400
__global__ void bcast_rows_kernel(const float* __restrict__ src, int T, int C, const float* __restrict__ mask, float* __restrict__ dst,
                                  int ld, int col_off) {
    const int row = blockIdx.x;
    const int b = row / T;
    const float mk = mask ? mask[row] : 1.0f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[(size_t)row * ld + col_off + c] = src[(size_t)b * C + c] * mk;
}
hipError_t launch_bcast_rows(const float* src, int B, int T, int C, const float* mask, float* dst, int ld, int col_off, hipStream_t s) {
    if (!src || !dst || B <= 0 || T <= 0 || C <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(B * T), dim3(64), 0, s, src, T, C, mask, dst, ld, col_off);
    return hipGetLastError();
}

// RoPE on the q and k sections of packed [B*T, 3*H*D] rows, first d_rope dims of each head, half-rotation pairs
// (i, i + d_rope/2), absolute positions (reference text_encoder.py:151-173).  cos/sin tables [>=T, d_rope] from the host.
__global__ void rope_kernel(float* __restrict__ qkv, int T, int H, int D, int d_rope, const float* __restrict__ cos_t,
                            const float* __restrict__ sin_t) {
    const int row = blockIdx.x;
    const int t = row % T;
    const int half = d_rope / 2;
    const int per_sec = H * half;
    float* base = qkv + (size_t)row * 3 * H * D;
    for (int i = threadIdx.x; i < 2 * per_sec; i += blockDim.x) {
        const int sec = i / per_sec, j = i % per_sec;
        const int hd = j / half, k = j % half;
        float* x = base + sec * H * D + hd * D;
        const float a = x[k], b2 = x[k + half];
        const float c0 = cos_t[(size_t)t * d_rope + k], s0 = sin_t[(size_t)t * d_rope + k];
        const float c1 = cos_t[(size_t)t * d_rope + k + half], s1 = sin_t[(size_t)t * d_rope + k + half];
        x[k] = a * c0 + (-b2) * s0;
        x[k + half] = b2 * c1 + a * s1;
    }
}
hipError_t launch_rope(float* qkv, int B, int T, int H, int D, int d_rope, const float* cos_t, const float* sin_t, hipStream_t s) {
    if (!qkv || !cos_t || !sin_t || B <= 0 || T <= 0 || d_rope <= 0 || (d_rope & 1) || d_rope > D) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rope_kernel, dim3(B * T), dim3(64), 0, s, qkv, T, H, D, d_rope, cos_t, sin_t);
    return hipGetLastError();
}

// Durations (reference inference.py:127-146): d = round(((exp(logw) - 2) * mask) * sc * ls).clamp(min=1) * mask,
// inclusive cumsum as int32, fine length = max(sum, 1).  One workgroup per utterance, serial-chunk + block scan.
__global__ __launch_bounds__(256) void durations_kernel(const float* __restrict__ logw, const float* __restrict__ mask, float sc, float ls,
                                                        const float* __restrict__ sc_b, const float* __restrict__ ls_b,
                                                        int Tx, float* __restrict__ dur, int32_t* __restrict__ cum,
                                                        int64_t* __restrict__ yfl) {
    __shared__ int part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (sc_b) sc = sc_b[b];          // per-utterance factors (a batch of requests with different voices / speeds)
    if (ls_b) ls = ls_b[b];
    const int per = (Tx + 255) / 256;
    const int i0 = tid * per, i1 = min(Tx, i0 + per);
    int s = 0;
    for (int i = i0; i < i1; ++i) {
        const float m = mask[(size_t)b * Tx + i];
        float d = (expf(logw[(size_t)b * Tx + i]) - 2.0f) * m;
        d = d * sc;
        d = d * ls;
        d = fmaxf(rintf(d), 1.0f) * m;
        dur[(size_t)b * Tx + i] = d;
        s += (int)d;
    }
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {      // Hillis-Steele inclusive scan
        int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int i = i0; i < i1; ++i) {
        run += (int)dur[(size_t)b * Tx + i];
        cum[(size_t)b * Tx + i] = run;
    }
    if (tid == 255) yfl[b] = max(part[255], 1);
}
hipError_t launch_durations(const float* logw, const float* mask, float sc, float ls, int B, int Tx, float* dur, int32_t* cum,
                            int64_t* yfl, hipStream_t s, const float* sc_b, const float* ls_b) {
    if (!logw || !mask || !dur || !cum || !yfl || B <= 0 || Tx <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(256), 0, s, logw, mask, sc, ls, sc_b, ls_b, Tx, dur, cum, yfl);
    return hipGetLastError();
}

// generate_path + (mu_x @ path) + avg_pool1d(k3,s2,p1) + sequence_mask (reference inference.py:146-167,
// utils/model.py:24-40,57-68) without the one-hot [Tx, 2*T_pad] matrix: fine frame f belongs to the first token whose
// cumulative duration exceeds f (frames >= fine length are zero); mu_y[t] = (fine[2t-1] + fine[2t] + fine[2t+1]) / 3.
__global__ __launch_bounds__(256) void align_pool_kernel(const float* __restrict__ mu_x, const int32_t* __restrict__ cum,
                                                         const int64_t* __restrict__ yfl, int nf, int Tx, int T_pad,
                                                         float* __restrict__ mu_y, float* __restrict__ y_mask,
                                                         int64_t* __restrict__ y_len) {
    __shared__ int tok[3][32];
    const int b = blockIdx.y, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int32_t* cb = cum + (size_t)b * Tx;
    const int total = cb[Tx - 1];
    const int64_t ylen = max((yfl[b] + 1) / 2, (int64_t)1);
    if (ty < 3) {
        const int f = 2 * (t0 + tx) - 1 + ty;
        int r = -1;
        if (f >= 0 && f < total) {
            int lo = 0, hi = Tx - 1;            // first i with cum[i] > f
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cb[mid] > f) hi = mid; else lo = mid + 1;
            }
            r = lo;
        }
        tok[ty][tx] = r;
    }
    __syncthreads();
    const int t = t0 + tx;
    if (t >= T_pad) return;
    const int k0 = tok[0][tx], k1 = tok[1][tx], k2 = tok[2][tx];
    for (int c = ty; c < nf; c += 8) {
        const float* mx = mu_x + ((size_t)b * nf + c) * Tx;
        float sum = 0.f;
        if (k0 >= 0) sum += mx[k0];
        if (k1 >= 0) sum += mx[k1];
        if (k2 >= 0) sum += mx[k2];
        mu_y[((size_t)b * nf + c) * T_pad + t] = sum / 3.0f;
    }
    if (ty == 0) {
        y_mask[(size_t)b * T_pad + t] = (int64_t)t < ylen ? 1.0f : 0.0f;
        if (t == 0) y_len[b] = ylen;
    }
}
hipError_t launch_align_pool(const float* mu_x, const int32_t* cum, const int64_t* yfl, int B, int nf, int Tx, int T_pad,
                             float* mu_y, float* y_mask, int64_t* y_len, hipStream_t s) {
    if (!mu_x || !cum || !yfl || !mu_y || !y_mask || !y_len || B <= 0 || nf <= 0 || Tx <= 0 || T_pad <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(align_pool_kernel, dim3((T_pad + 31) / 32, B), dim3(256), 0, s, mu_x, cum, yfl, nf, Tx, T_pad, mu_y, y_mask, y_len);
    return hipGetLastError();
}

}  // namespace mtts
