// Kernels of the Vocos-24k head that are not GEMMs (gfx950): depthwise k7 conv + LayerNorm, polar spectrum, iSTFT
// overlap-add.  The dense parts (embed conv k7, pwconv1/2, head projection, inverse real DFT as a [n_fft x (n_fft+2)]
// matrix) run on gemm_f32_kernel.  Architecture: reference matcha/vocos24k/config.yaml:10-24 + the vocos package
// (VocosBackbone / ConvNeXtBlock / ISTFTHead); call site reference matcha/vocos24k/vocos_wrapper.py:8-9.
#include "kernels.h"

namespace mtts {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int DW_MAXV = 8;   // float4 per lane => C <= 2048

__device__ __forceinline__ float wave_sum_v(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ConvNeXtBlock front: dwconv(k7, pad 3, groups=C) -> LayerNorm(eps).  One wave per output row; the 7 input rows are
// neighbours' rows too (L1/L2 hits).  HBM-bound: reads and writes the tensor once.
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, const float* __restrict__ w7,
                                                         const float* __restrict__ bias, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, int T, int C, int M,
                                                         float* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const int t = row % T;
    f32x4 v[DW_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < DW_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int tt = t + j - 3;
                if (tt >= 0 && tt < T) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)(row + j - 3) * C + c);
                    acc += xv * *reinterpret_cast<const f32x4*>(w7 + (size_t)j * C + c);
                }
            }
            v[i] = acc;
            s += (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
    }
    const float mu = wave_sum_v(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < DW_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < C) {
            const f32x4 d = v[i] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum_v(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < DW_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < C) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
            *reinterpret_cast<f32x4*>(y + (size_t)row * C + c) = ((v[i] - mu) * rs) * g + b;
        }
    }
}

hipError_t launch_dwconv7_ln(const float* x, const float* w7, const float* bias, const float* gamma, const float* beta, float eps,
                             int B, int T, int C, float* y, hipStream_t s) {
    if (!x || !w7 || !bias || !gamma || !beta || !y || B <= 0 || T <= 0 || C <= 0 || (C & 3) || C > 64 * 4 * DW_MAXV) return hipErrorInvalidValue;
    const int M = B * T;
    hipLaunchKernelGGL(dwconv7_ln_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, w7, bias, gamma, beta, eps, T, C, M, y);
    return hipGetLastError();
}

// ISTFTHead: mag = clip(exp(m), max), S = mag * (cos p + i sin p)
__global__ void spec_polar_kernel(float* __restrict__ x, int M, int ld, int nbins, int off, float clip) {
    const size_t n = (size_t)M * nbins;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / nbins, k = i % nbins;
        float* row = x + r * ld;
        const float mag = fminf(expf(row[k]), clip);
        const float p = row[off + k];
        row[k] = mag * cosf(p);
        row[off + k] = mag * sinf(p);
    }
}
hipError_t launch_spec_polar(float* x, int M, int ld, int nbins, int off, float clip, hipStream_t s) {
    if (!x || M <= 0 || nbins <= 0 || off < nbins || off + nbins > ld) return hipErrorInvalidValue;
    const size_t n = (size_t)M * nbins;
    hipLaunchKernelGGL(spec_polar_kernel, dim3((unsigned)min((n + 255) / 256, (size_t)4096)), dim3(256), 0, s, x, M, ld, nbins, off, clip);
    return hipGetLastError();
}

// torch.istft(center=True) tail: y[pos] = sum_f frame_f[pos - f*hop] / sum_f window^2[pos - f*hop], pos = s + n_fft/2,
// output sample s in [0, hop*(T-1)).  (The frames already carry one window factor from the DFT matrix.)
__global__ void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window, int T, int n_fft, int hop,
                                 float* __restrict__ audio) {
    const int b = blockIdx.y;
    const int L = hop * (T - 1);
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    const int pos = s + n_fft / 2;
    int f0 = (pos - n_fft + hop) / hop;     // ceil((pos - n_fft + 1) / hop) for pos - n_fft + 1 > 0
    if (pos - n_fft + 1 <= 0) f0 = 0;
    int f1 = pos / hop;
    if (f1 > T - 1) f1 = T - 1;
    float acc = 0.f, env = 0.f;
    for (int f = f0; f <= f1; ++f) {
        const int j = pos - f * hop;
        const float w = window[j];
        acc += frames[((size_t)b * T + f) * n_fft + j];
        env += w * w;
    }
    audio[(size_t)b * L + s] = env > 1e-11f ? acc / env : acc;
}
hipError_t launch_istft_ola(const float* frames, const float* window, int B, int T, int n_fft, int hop, float* audio, hipStream_t s) {
    if (!frames || !window || !audio || B <= 0 || T < 2 || n_fft <= 0 || hop <= 0 || n_fft % hop) return hipErrorInvalidValue;
    const int L = hop * (T - 1);
    hipLaunchKernelGGL(istft_ola_kernel, dim3((L + 255) / 256, B), dim3(256), 0, s, frames, window, T, n_fft, hop, audio);
    return hipGetLastError();
}

}  // namespace mtts
