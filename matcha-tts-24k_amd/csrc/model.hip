// Host orchestration of the path on one HIP stream: weight packing, workspace planning and the launch sequences.
// No torch, no allocation or synchronisation inside the launch functions (graph-capturable).
#include "model.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <time.h>

namespace mtts {

static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
const char* get_error() { return g_err.c_str(); }

#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)
#define RET_IF(expr)          \
    do {                      \
        int _r = (expr);      \
        if (_r) return _r;    \
    } while (0)

// ------------------------------------------------------------------------------------------------ profiling wrappers
static int prof_begin(mtts_ctx* c, int klass, double flops, double bytes, hipStream_t s) {
    if (!c || !c->prof_on) return 0;
    while (c->ev_pool.size() < c->ev_used + 2) {
        hipEvent_t e;
        HIP_OK(hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    ProfRec r{c->ev_pool[c->ev_used], c->ev_pool[c->ev_used + 1], klass, flops, bytes, std::string()};
    g_kernel_tag = nullptr;
    c->ev_used += 2;
    HIP_OK(hipEventRecord(r.e0, s));
    c->prof.push_back(r);
    return 0;
}
static int prof_end(mtts_ctx* c, hipStream_t s) {
    if (!c || !c->prof_on) return 0;
    HIP_OK(hipEventRecord(c->prof.back().e1, s));
    c->prof.back().tag = g_kernel_tag ? g_kernel_tag : "";        // (set by the launcher that ran in between, or null)
    return 0;
}
#define LAUNCH(ctx, klass, flops, stream, call)  \
    LAUNCHB(ctx, klass, flops, 0.0, stream, call)
#define LAUNCHB(ctx, klass, flops, bytes, stream, call)  \
    do {                                         \
        RET_IF(prof_begin(ctx, klass, flops, bytes, stream)); \
        HIP_OK(call);                            \
        RET_IF(prof_end(ctx, stream));           \
    } while (0)

static int run_gemm(mtts_ctx* c, const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.range_flag = c->cur_flag;
    a.half16 = c->half_now && a.a16_0 != nullptr;
    a.bf16 = a.half16 && c->bf16;
    LAUNCHB(c, 0, gemm_flops(a), gemm_bytes(a), s, launch_gemm(a, s));
    return 0;
}
static int run_attn(mtts_ctx* c, const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
    a.range_flag = c->cur_flag;
    a.half16 = c->half_now && a.qkv16 != nullptr;
    a.bf16 = a.half16 && c->bf16;
    LAUNCHB(c, 1, attn_flops(a), attn_bytes(a), s, launch_attention(a, s));
    return 0;
}
// prefetch workgroups of the chain launch (tblock_chain.hip): two per XCD (one alone takes ~93 us for the 7 MB stream and is the tail
// of the launches without a q|k|v phase: 16 instead of 8 = -0.15 ms of GEMM time per step); MTTS_CHAIN_PF=<n> (0: none) for A/B runs
static int chain_prefetch_wgs() {
    static const int n = [] { const char* e = getenv("MTTS_CHAIN_PF"); const int v = e ? atoi(e) : 16; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
    return n;
}
// Launch plan of a chain launch over M rows (hidden chunk ch; qb_forced = MTTS_CHAIN_QB or 0): rows per workgroup and prefetch
// workgroups.  32-row workgroups while they -- with the prefetchers -- are one round of the 256 CUs (a 32-row workgroup lives 99 us, a
// 48-row one 116 us: both are bound by the 7 MB they stream, profiles/r03_chain_prefetch_stamps.log), the largest shape beyond; no
// prefetchers when they would push a one-round grid into a second round.
static void chain_plan(int M, int ch, int qb_forced, int* qb, int* pf) {
    const int qb_big = ch == 256 ? 48 : 64, want = chain_prefetch_wgs();
    const bool fits32 = (M + 31) / 32 + want <= 256;
    *qb = (qb_forced == 32 || qb_forced == qb_big) ? qb_forced : (fits32 ? 32 : qb_big);
    const int nwg = (M + *qb - 1) / *qb;
    *pf = want;
    if (nwg <= 256 && nwg + *pf > 256) *pf = (want >= 8 && nwg + 8 <= 256) ? 8 : 0;
}
static int run_chain(mtts_ctx* c, const ChainArgs& a0, hipStream_t s) {
    ChainArgs a = a0;
    a.range_flag = c->cur_flag;
    { int qb_unused = 0; chain_plan(a.M, a.ch, a.qb, &qb_unused, &a.pf_wgs); }
    if (a.pair) a.pf_wgs = chain_prefetch_wgs() ? 16 : 0;      // two per XCD, one per half (the model admits pair grids up to 240 workgroups)
    LAUNCHB(c, 0, chain_flops(a), chain_bytes(a), s, launch_tblock_chain(a, s));
    return 0;
}
#ifdef MTTS_CHAIN_VERIFY
// Diagnostic builds (tools/build_variant.sh ... -DMTTS_CHAIN_VERIFY): a chain launch that does not update its inputs in place is
// run a second time into a scratch image and the two results are compared on the device -- per launch slot
// [differing 16-byte chunks, first row, last row, first chunk, last chunk, rows, 0, 0] (mtts_debug_verify_read).
constexpr int VERIFY_SLOTS = 8192;
static int* g_verify = nullptr;
static int g_verify_n = 0;
__global__ void chain_verify_kernel(const uint4* a, const uint4* b, int rows, int cpr, int* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cpr) return;
    const uint4 x = a[i], y = b[i];
    if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) {
        const int row = (int)(i / cpr), ch = (int)(i % cpr);
        atomicAdd(out, 1); atomicMin(out + 1, row); atomicMax(out + 2, row); atomicMin(out + 3, ch); atomicMax(out + 4, ch);
    }
}
// the first launch whose executions 1 and 2 differ: both images kept for inspection (mtts_debug_verify_snapshot)
static int* g_snap_flag = nullptr;
static uint4 *g_snap_a = nullptr, *g_snap_b = nullptr;
static size_t g_snap_bytes = 0;
__global__ void verify_claim_kernel(const int* slot, int* flag, int id) { if (slot[0] > 0 && flag[0] == 0) flag[0] = id; }
__global__ void verify_copy_kernel(const int* flag, int id, const uint4* a, const uint4* b, uint4* sa, uint4* sb, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (flag[0] != id || i >= n) return;
    sa[i] = a[i]; sb[i] = b[i];
}
extern "C" int mtts_debug_verify_snapshot(void* host_a, void* host_b, size_t bytes) {
    if (!g_snap_flag) return 0;
    (void)hipDeviceSynchronize();
    int id = 0;
    (void)hipMemcpy(&id, g_snap_flag, sizeof(int), hipMemcpyDeviceToHost);
    if (id == 0 || bytes < g_snap_bytes) return 0;
    (void)hipMemcpy(host_a, g_snap_a, g_snap_bytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(host_b, g_snap_b, g_snap_bytes, hipMemcpyDeviceToHost);
    (void)hipMemset(g_snap_flag, 0, sizeof(int));
    return id;
}
#ifdef MTTS_CHAIN_DUMP
// LDS dumps of executions 1 and 2 (tblock_chain.hip CH_DUMP), compared per section: [x0, ct, x1, srow, h0, x2, 0, 0] per launch
static char *g_dump_a = nullptr, *g_dump_b = nullptr;
static int* g_sect = nullptr;
__global__ void dump_compare_kernel(const uint4* a, const uint4* b, long n, int wg_chunks, int e0, int e1, int e2, int e3, int e4, int* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 x = a[i], y = b[i];
    if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) {
        const int o = (int)(i % wg_chunks);
        const int sec = o < e0 ? 0 : o < e1 ? 1 : o < e2 ? 2 : o < e3 ? 3 : o < e4 ? 4 : 5;
        atomicAdd(out + sec, 1);
        if (sec == 1) atomicMin(out + 6, o - e0);       // first differing chunk of the constants
        if (sec == 3) atomicMin(out + 7, o - e2);
    }
}
extern "C" int mtts_debug_verify_sections(int* out, int max_launches) {
    if (!g_sect) return 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out, g_sect, (size_t)max_launches * 8 * sizeof(int), hipMemcpyDeviceToHost);
    std::vector<int> h((size_t)VERIFY_SLOTS * 4, 0);
    for (int k = 0; k < VERIFY_SLOTS / 2; ++k) { h[8 * k + 6] = 1 << 30; h[8 * k + 7] = 1 << 30; }
    (void)hipMemcpy(g_sect, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    return 1;
}
#endif
#ifdef MTTS_CHAIN_PROBE
static unsigned int *g_probe_a = nullptr, *g_probe_b = nullptr, *g_probe_sa = nullptr, *g_probe_sb = nullptr;
static int g_probe_wgs = 0;
__global__ void probe_keep_kernel(const int* flag, int id, const unsigned int* a, const unsigned int* b, unsigned int* sa, unsigned int* sb, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (flag[0] != id || i >= n) return;
    sa[i] = a[i]; sb[i] = b[i];
}
extern "C" int mtts_debug_probe_read(unsigned int* host_a, unsigned int* host_b, int max_wgs) {
    if (!g_probe_sa) return 0;
    (void)hipDeviceSynchronize();
    const int n = g_probe_wgs < max_wgs ? g_probe_wgs : max_wgs;
    (void)hipMemcpy(host_a, g_probe_sa, (size_t)n * 8 * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(host_b, g_probe_sb, (size_t)n * 8 * 4, hipMemcpyDeviceToHost);
    return n;
}
#endif
static void verify_reset() {
    std::vector<int> h((size_t)VERIFY_SLOTS * 8, 0);
    for (int k = 0; k < VERIFY_SLOTS; ++k) { h[8 * k + 1] = 1 << 30; h[8 * k + 3] = 1 << 30; h[8 * k + 2] = -1; h[8 * k + 4] = -1; }
    if (!g_verify) (void)hipMalloc(&g_verify, h.size() * sizeof(int));
    (void)hipMemcpy(g_verify, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    g_verify_n = 0;
}
static int run_chain_verified(mtts_ctx* c, const ChainArgs& a0, _Float16* scratch, hipStream_t s) {
    ChainArgs a = a0;
    if (!g_verify) verify_reset();
#ifdef MTTS_CHAIN_DUMP
    {
        const size_t wgb = 3 * (size_t)(a.qb * a.C * 4) + 18 * a.C * 4 + 2 * a.qb * 4 + a.qb * a.ch * 4, nwgs = (a.M + a.qb - 1) / a.qb;
        if (!g_dump_a) {
            (void)hipMalloc(&g_dump_a, wgb * nwgs); (void)hipMalloc(&g_dump_b, wgb * nwgs);
            (void)hipMalloc(&g_sect, (size_t)VERIFY_SLOTS * 4 * sizeof(int));
            std::vector<int> h((size_t)VERIFY_SLOTS * 4, 0);
            for (int k = 0; k < VERIFY_SLOTS / 2; ++k) { h[8 * k + 6] = 1 << 30; h[8 * k + 7] = 1 << 30; }
            (void)hipMemcpy(g_sect, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
        }
        if (a.x_out != a.x16) a.kstamp = reinterpret_cast<unsigned long long*>(g_dump_a);
    }
#endif
#ifdef MTTS_CHAIN_PROBE
    {
        const int nwgs = (a.M + a.qb - 1) / a.qb;
        if (!g_probe_a) {
            g_probe_wgs = nwgs;
            (void)hipMalloc(&g_probe_a, (size_t)nwgs * 32); (void)hipMalloc(&g_probe_b, (size_t)nwgs * 32);
            (void)hipMalloc(&g_probe_sa, (size_t)nwgs * 32); (void)hipMalloc(&g_probe_sb, (size_t)nwgs * 32);
        }
        if (a.x_out != a.x16 && nwgs == g_probe_wgs) {
            (void)hipMemsetAsync(g_probe_a, 0, (size_t)nwgs * 32, s); (void)hipMemsetAsync(g_probe_b, 0, (size_t)nwgs * 32, s);
            a.kstamp = reinterpret_cast<unsigned long long*>(g_probe_a);
        }
    }
#endif
    RET_IF(run_chain(c, a, s));
    if (a.pair) return 0;            // (a pair launch's flags carry one epoch: it cannot simply be executed again)
    if (a.x_out == a.x16 || g_verify_n + 2 > VERIFY_SLOTS) return 0;
    ChainArgs b = a, b3 = a;
    b.x_out = scratch;
    b3.x_out = scratch + (size_t)a.M * 2 * a.C;
    b3.kstamp = nullptr;
#ifdef MTTS_CHAIN_PROBE
    if (a.kstamp) b.kstamp = reinterpret_cast<unsigned long long*>(g_probe_b);
#endif
#ifdef MTTS_CHAIN_DUMP
    const int xt = a.qb * a.C * 4, ctb = 18 * a.C * 4, sr = 2 * a.qb * 4, ht = a.qb * a.ch * 4;
    const size_t wg_bytes = 3 * (size_t)xt + ctb + sr + ht, nwg = (a.M + a.qb - 1) / a.qb;
    if (!g_dump_a) {
        (void)hipMalloc(&g_dump_a, wg_bytes * nwg); (void)hipMalloc(&g_dump_b, wg_bytes * nwg);
        (void)hipMalloc(&g_sect, (size_t)VERIFY_SLOTS * 4 * sizeof(int));
        std::vector<int> h((size_t)VERIFY_SLOTS * 4, 0);
        for (int k = 0; k < VERIFY_SLOTS / 2; ++k) { h[8 * k + 6] = 1 << 30; h[8 * k + 7] = 1 << 30; }
        (void)hipMemcpy(g_sect, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    b.kstamp = reinterpret_cast<unsigned long long*>(g_dump_b);
#endif
    RET_IF(run_chain(c, b, s));
    RET_IF(run_chain(c, b3, s));
    const int cpr = a.C / 4;         // 16-byte chunks per image row
    const long n = (long)a.M * cpr;
    // two slots per launch: execution 1 vs 2, execution 2 vs 3
    int* slot = g_verify + 8 * (size_t)g_verify_n++;
    hipLaunchKernelGGL(chain_verify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(a.x_out), reinterpret_cast<const uint4*>(b.x_out), a.M, cpr, slot);
    if (!g_snap_flag) {
        g_snap_bytes = (size_t)n * 16;
        (void)hipMalloc(&g_snap_flag, sizeof(int)); (void)hipMemset(g_snap_flag, 0, sizeof(int));
        (void)hipMalloc(&g_snap_a, g_snap_bytes); (void)hipMalloc(&g_snap_b, g_snap_bytes);
    }
    if ((size_t)n * 16 == g_snap_bytes) {
        hipLaunchKernelGGL(verify_claim_kernel, dim3(1), dim3(1), 0, s, slot, g_snap_flag, g_verify_n);
        hipLaunchKernelGGL(verify_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g_snap_flag, g_verify_n,
                           reinterpret_cast<const uint4*>(a.x_out), reinterpret_cast<const uint4*>(b.x_out), g_snap_a, g_snap_b, n);
#ifdef MTTS_CHAIN_PROBE
        if (a.kstamp) hipLaunchKernelGGL(probe_keep_kernel, dim3((unsigned)((g_probe_wgs * 8 + 255) / 256)), dim3(256), 0, s, g_snap_flag, g_verify_n,
                                         g_probe_a, g_probe_b, g_probe_sa, g_probe_sb, g_probe_wgs * 8);
#endif
    }
#ifdef MTTS_CHAIN_DUMP
    {
        const int wgc = (int)(wg_bytes / 16), e0 = xt / 16, e1 = e0 + ctb / 16, e2 = e1 + xt / 16, e3 = e2 + sr / 16, e4 = e3 + ht / 16;
        const long nd = (long)wgc * (long)nwg;
        hipLaunchKernelGGL(dump_compare_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint4*>(g_dump_a),
                           reinterpret_cast<const uint4*>(g_dump_b), nd, wgc, e0, e1, e2, e3, e4, g_sect + 8 * (size_t)(g_verify_n / 2));
    }
#endif
    slot = g_verify + 8 * (size_t)g_verify_n++;
    hipLaunchKernelGGL(chain_verify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(b.x_out), reinterpret_cast<const uint4*>(b3.x_out), a.M, cpr, slot);
    return 0;
}
extern "C" int mtts_debug_verify_read(int* out, int max_slots) {
    if (!g_verify) return 0;
    (void)hipDeviceSynchronize();
    const int n = g_verify_n < max_slots ? g_verify_n : max_slots;
    (void)hipMemcpy(out, g_verify, (size_t)n * 8 * sizeof(int), hipMemcpyDeviceToHost);
    verify_reset();
    return n;
}
#endif
static int run_gn_apply(mtts_ctx* c, const GnApplyArgs& a0, hipStream_t s) {
    GnApplyArgs a = a0;
    a.range_flag = c->cur_flag;
    a.half16 = c->half_now && a.out16 != nullptr;
    a.bf16 = a.half16 && c->bf16;
    LAUNCH(c, 2, 0, s, launch_gn_apply(a, s));
    return 0;
}
// The sticky range flag of a call = the first word of its workspace, cleared here (include/mtts.h "range guard").
static int begin_call(mtts_ctx* c, void* d_ws, hipStream_t s) {
    c->cur_flag = static_cast<unsigned int*>(d_ws);
    // a kernel, not hipMemsetAsync: a captured memset node of one HIP graph was seen to write another instantiated graph's
    // bytes (pointer-like words in this header) after a second context captured its own graph (ROCm 7.2); kernel nodes are safe
    HIP_OK(launch_fill_cols(static_cast<float*>(d_ws), 1, 64, 0, 64, 0.f, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------ workspace
struct WS {
    char* base;
    size_t off = 0, cap;
    bool overflow = false;
    WS(void* p, size_t c) : base(static_cast<char*>(p)), cap(c) {}
    void* bytes(size_t n) {
        off = (off + 255) & ~size_t(255);
        void* r = base ? base + off : nullptr;
        off += n;
        if (base && off > cap) overflow = true;
        return r;
    }
    float* f(size_t n) { return static_cast<float*>(bytes(n * sizeof(float))); }
};

// ------------------------------------------------------------------------------------------------ weight packing
struct Packer {
    mtts_ctx* c;
    bool ok = true;
    std::string why;
    int kq = GEMM_BK;          // K padding per tap of the panels being packed: 64 for the estimator in the 16-bit storage mode
    bool h16 = false;          // ... which also get the single fp16 plane (Panel::wh16)
    bool dry = false;          // layout only: offsets and sizes are computed (the registered tensors are still checked for presence and
                               // shape), nothing but zeros is written: mtts_import_weights takes the image itself from a cache
    explicit Packer(mtts_ctx* ctx) : c(ctx) {}
    const std::vector<float>* get(const std::string& key, size_t numel) {
        auto it = c->raw.find(key);
        if (it == c->raw.end()) { fail("missing tensor " + key); return nullptr; }
        if (it->second.size() != numel) {
            fail("tensor " + key + " has " + std::to_string(it->second.size()) + " elements, expected " + std::to_string(numel));
            return nullptr;
        }
        return &it->second;
    }
    void fail(const std::string& m) { if (ok) { ok = false; why = m; } }
    size_t alloc(size_t n) {
        size_t off = (c->image.size() + 63) & ~size_t(63);
        c->image.resize(off + n, 0.f);
        return off;
    }
    Vec vec(const std::string& key, int n) {
        Vec v;
        const auto* t = get(key, n);
        if (!t) return v;
        v.off = alloc(n);
        v.n = n;
        if (!dry) std::memcpy(&c->image[v.off], t->data(), n * sizeof(float));
        return v;
    }
    // Folded padding (kernels.h GnApplyArgs::bias_stats): where every input tap of a conv is a masked (zero) frame its output
    // row is exactly the bias, so any number of such rows enters the following GroupNorm in closed form from, per group,
    // (mean of the bias, sum of squared deviations from that mean), computed here in double.
    Vec bias_group_stats(const Panel& p, int G) {
        Vec v;
        v.off = alloc(2 * G);
        v.n = 2 * G;
        if (dry) return v;
        const int cpg = p.N / G;
        for (int g = 0; g < G; ++g) {
            double m = 0.0, q = 0.0;
            for (int k = 0; k < cpg; ++k) m += p.has_bias ? (double)c->image[p.b + g * cpg + k] : 0.0;
            m /= cpg;
            for (int k = 0; k < cpg; ++k) { const double d = (p.has_bias ? (double)c->image[p.b + g * cpg + k] : 0.0) - m; q += d * d; }
            c->image[v.off + 2 * g] = (float)m;
            c->image[v.off + 2 * g + 1] = (float)q;
        }
        return v;
    }
    // kind 0 Linear [N,C]; 1 Conv1d [N,C,ntaps]; 2 ConvTranspose1d [C,N,kT] with taps tsel
    Panel panel(const std::string& wkey, const std::string& bkey, int kind, int N, int C, int ntaps, int kT = 0,
                const int* tsel = nullptr, const std::vector<float>* col_scale = nullptr,
                const std::vector<float>* col_shift = nullptr) {
        return panel_multi({wkey}, {bkey}, kind, N, C, ntaps, kT, tsel, col_scale, col_shift);
    }
    // bf16 split planes of a finished fp32 panel (split modes only)
    void add_planes(Panel& p) {
        if (c->gemm_terms == 0) return;
        const size_t n = (size_t)round_up(p.N, GEMM_BN) * p.ntaps * p.ktap;
        p.w16 = alloc((3 * n + 1) / 2);
        if (c->gemm_terms == 2) {
            for (size_t i = 0; i < n && !dry; ++i)
                if (std::fabs(c->image[p.w + i]) > 65504.f) { c->weights_saturate = true; break; }
            if (!dry) split_panel_f16_host(&c->image[p.w], n, reinterpret_cast<uint16_t*>(&c->image[p.w16]));
            const int Np = round_up(p.N, GEMM_BN);
            const size_t Kp = (size_t)p.ntaps * p.ktap;
            p.wsum = alloc(Np);
            if (h16) {            // 16-bit storage mode: the fp16 head plane alone + row sums of the ROUNDED weights (LN epilogue)
                p.wh16 = alloc((n + 1) / 2);
                if (!dry && c->bf16) panel_bf16_host(&c->image[p.w], n, reinterpret_cast<uint16_t*>(&c->image[p.wh16]));
                else if (!dry) panel_h16_host(&c->image[p.w], n, reinterpret_cast<uint16_t*>(&c->image[p.wh16]));
            }
            for (int r = 0; r < Np && !dry; ++r) {
                double acc = 0.0;
                for (size_t k = 0; k < Kp; ++k) {
                    const float w = c->image[p.w + (size_t)r * Kp + k];
                    acc += !h16 ? (double)w : c->bf16 ? (double)(float)(__bf16)w : (double)(float)(_Float16)fminf(fmaxf(w, -65504.f), 65504.f);
                }
                c->image[p.wsum + r] = (float)acc;
            }
        } else if (!dry) split_panel_host(&c->image[p.w], n, reinterpret_cast<uint16_t*>(&c->image[p.w16]));
    }
    // a panel from explicit host data (rearranged / synthesised weights)
    Panel panel_from(const float* w, const float* bias, int kind, int N, int C, int ntaps) {
        Panel p;
        p.N = N; p.C = C; p.ntaps = ntaps; p.ktap = round_up(C, kq);
        const int Np = round_up(N, GEMM_BN);
        const size_t Kp = (size_t)ntaps * p.ktap;
        p.w = alloc((size_t)Np * Kp);
        p.b = alloc(Np);
        if (!dry) pack_weight_host(w, kind, N, C, ntaps, 0, nullptr, nullptr, &c->image[p.w], p.ktap);
        if (bias) { p.has_bias = true; if (!dry) std::memcpy(&c->image[p.b], bias, N * sizeof(float)); }
        add_planes(p);
        return p;
    }
    // several [N_i, C(,k)] tensors stacked along N into one panel (q|k|v, concatenated time MLPs)
    Panel panel_multi(const std::vector<std::string>& wkeys, const std::vector<std::string>& bkeys, int kind, int N_each, int C,
                      int ntaps, int kT = 0, const int* tsel = nullptr, const std::vector<float>* col_scale = nullptr,
                      const std::vector<float>* col_shift = nullptr) {
        Panel p;
        const int parts = (int)wkeys.size();
        p.N = N_each * parts;
        p.C = C;
        p.ntaps = ntaps;
        p.ktap = round_up(C, kq);
        const int Np = round_up(p.N, GEMM_BN);
        const size_t Kp = (size_t)ntaps * p.ktap;
        p.w = alloc((size_t)Np * Kp);
        p.b = alloc(Np);
        const size_t per = (kind == 2) ? (size_t)C * N_each * kT : (size_t)N_each * C * ntaps;
        std::vector<float> tmp((size_t)round_up(N_each, GEMM_BN) * Kp);
        for (int part = 0; part < parts; ++part) {
            const auto* w = get(wkeys[part], per);
            if (!w) return p;
            const bool hb = part < (int)bkeys.size() && !bkeys[part].empty();
            const std::vector<float>* b = hb ? get(bkeys[part], N_each) : nullptr;
            if (hb && !b) return p;
            if (hb || col_shift) p.has_bias = true;
            if (dry) continue;
            pack_weight_host(w->data(), kind, N_each, C, ntaps, kT, tsel, col_scale ? col_scale->data() : nullptr, tmp.data(), p.ktap);
            std::memcpy(&c->image[p.w + (size_t)part * N_each * Kp], tmp.data(), (size_t)N_each * Kp * sizeof(float));
            for (int n = 0; n < N_each; ++n) {
                double acc = b ? (double)(*b)[n] : 0.0;
                if (col_shift) {   // LayerNorm beta folded through the projection: b' = b + W . beta
                    for (int cc = 0; cc < C; ++cc) acc += (double)(*w)[(size_t)n * C + cc] * (double)(*col_shift)[cc];
                }
                c->image[p.b + (size_t)part * N_each + n] = (float)acc;
            }
        }
        add_planes(p);
        return p;
    }
};

static int pack_all(mtts_ctx* c, bool dry = false) {
    const mtts_config& g = c->cfg;
    c->image.clear();
    c->weights_saturate = false;
    Packer P(c);
    P.dry = dry;
    auto S = [](const std::string& a, int i, const std::string& b) { return a + std::to_string(i) + b; };
    const int taps3[3] = {-1, 0, 1};
    (void)taps3;

    // ---------------- text encoder (reference text_encoder.py:319-373)
    EncW& E = c->enc;
    E = EncW();
    const int nch = g.enc_channels, Sd = g.spk_emb_dim, Hd = nch + Sd, F = g.dp_filter;
    const int dh = Hd / g.enc_heads, d_rope = dh / 2;
    E.emb = P.vec("encoder.emb.weight", g.n_vocab * nch);
    E.spk_enc = P.vec("speaker_embeddings_enc.weight", g.n_spks * Sd);
    E.spk_dur = P.vec("speaker_embeddings_dur.weight", g.n_spks * Sd);
    {
        auto it = c->raw.find("aux.rope_cos");
        if (it == c->raw.end() || it->second.size() % d_rope) P.fail("aux.rope_cos missing or misshaped");
        else {
            E.rope_cos = P.vec("aux.rope_cos", (int)it->second.size());
            E.rope_sin = P.vec("aux.rope_sin", (int)it->second.size());
        }
    }
    for (int i = 0; i < g.prenet_layers; ++i) {
        E.pre_conv.push_back(P.panel(S("encoder.prenet.conv_layers.", i, ".weight"), S("encoder.prenet.conv_layers.", i, ".bias"), 1, nch, nch, g.prenet_kernel));
        E.pre_g.push_back(P.vec(S("encoder.prenet.norm_layers.", i, ".gamma"), nch));
        E.pre_b.push_back(P.vec(S("encoder.prenet.norm_layers.", i, ".beta"), nch));
    }
    E.pre_proj = P.panel("encoder.prenet.proj.weight", "encoder.prenet.proj.bias", 1, nch, nch, 1);
    for (int i = 0; i < g.enc_layers; ++i) {
        const std::string a = S("encoder.encoder.attn_layers.", i, ".");
        E.qkv.push_back(P.panel_multi({a + "conv_q.weight", a + "conv_k.weight", a + "conv_v.weight"},
                                      {a + "conv_q.bias", a + "conv_k.bias", a + "conv_v.bias"}, 1, Hd, Hd, 1));
        E.o.push_back(P.panel(a + "conv_o.weight", a + "conv_o.bias", 1, Hd, Hd, 1));
        E.n1_g.push_back(P.vec(S("encoder.encoder.norm_layers_1.", i, ".gamma"), Hd));
        E.n1_b.push_back(P.vec(S("encoder.encoder.norm_layers_1.", i, ".beta"), Hd));
        const std::string f = S("encoder.encoder.ffn_layers.", i, ".");
        E.ffn1.push_back(P.panel(f + "conv_1.weight", f + "conv_1.bias", 1, g.enc_filter, Hd, g.enc_kernel));
        E.ffn2.push_back(P.panel(f + "conv_2.weight", f + "conv_2.bias", 1, Hd, g.enc_filter, g.enc_kernel));
        E.n2_g.push_back(P.vec(S("encoder.encoder.norm_layers_2.", i, ".gamma"), Hd));
        E.n2_b.push_back(P.vec(S("encoder.encoder.norm_layers_2.", i, ".beta"), Hd));
    }
    E.pm0 = P.panel("encoder.proj_m.0.weight", "encoder.proj_m.0.bias", 1, nch, Hd, 1);
    E.pm2 = P.panel("encoder.proj_m.2.weight", "encoder.proj_m.2.bias", 1, g.n_feats, nch, 1);
    E.film = P.panel("encoder.proj_w.spk_proj.weight", "encoder.proj_w.spk_proj.bias", 0, 2 * F, Sd, 1);
    for (int i = 0; i < g.dp_layers; ++i) {
        E.dp_conv.push_back(P.panel(S("encoder.proj_w.conv_layers.", i, ".weight"), S("encoder.proj_w.conv_layers.", i, ".bias"), 1, F,
                                    i == 0 ? Hd : F, g.dp_kernel));
        E.dp_g.push_back(P.vec(S("encoder.proj_w.norm_layers.", i, ".gamma"), F));
        E.dp_b.push_back(P.vec(S("encoder.proj_w.norm_layers.", i, ".beta"), F));
    }
    E.dp_proj = P.panel("encoder.proj_w.proj.weight", "encoder.proj_w.proj.bias", 1, 1, F, 1);

    // ---------------- decoder (reference decoder.py:202-310)
    P.kq = c->half16 ? 64 : GEMM_BK;
    P.h16 = c->half16;
    DecW& D = c->dec;
    D = DecW();
    const std::string R = "decoder.estimator.";
    const int cin0 = 2 * g.n_feats, nl = g.dec_levels, temb = g.dec_channels[0] * 4;
    const int inner = g.dec_heads * g.dec_head_dim;
    D.freqs = P.vec("aux.time_freqs", cin0 / 2);
    D.t1 = P.panel(R + "time_mlp.linear_1.weight", R + "time_mlp.linear_1.bias", 0, temb, cin0, 1);
    D.t2 = P.panel(R + "time_mlp.linear_2.weight", R + "time_mlp.linear_2.bias", 0, temb, temb, 1);

    std::vector<std::string> mlp_w, mlp_b;
    std::vector<int> mlp_n;
    auto resnet = [&](const std::string& p, int ci, int co) {
        ResnetW r;
        r.cin = ci;
        r.cout = co;
        r.conv1 = P.panel(p + "block1.block.0.weight", p + "block1.block.0.bias", 1, co, ci, 3);
        r.gn1_g = P.vec(p + "block1.block.1.weight", co);
        r.gn1_b = P.vec(p + "block1.block.1.bias", co);
        r.conv2 = P.panel(p + "block2.block.0.weight", p + "block2.block.0.bias", 1, co, co, 3);
        r.gn2_g = P.vec(p + "block2.block.1.weight", co);
        r.gn2_b = P.vec(p + "block2.block.1.bias", co);
        r.res = P.panel(p + "res_conv.weight", p + "res_conv.bias", 1, co, ci, 1);
        if (P.ok) { r.gn1_bs = P.bias_group_stats(r.conv1, 8); r.gn2_bs = P.bias_group_stats(r.conv2, 8); }
        mlp_w.push_back(p + "mlp.1.weight");
        mlp_b.push_back(p + "mlp.1.bias");
        mlp_n.push_back(co);
        D.res.push_back(r);
    };
    auto tblock = [&](const std::string& p, int ch) {
        TBlockW t;
        const auto* g1 = P.get(p + "norm1.weight", ch);
        const auto* b1 = P.get(p + "norm1.bias", ch);
        const auto* g3 = P.get(p + "norm3.weight", ch);
        const auto* b3 = P.get(p + "norm3.bias", ch);
        if (!g1 || !b1 || !g3 || !b3) return;
        // nn.LayerNorm affine folded into the projection that consumes it: W' = W * gamma (per column), b' = b + W . beta
        t.qkv = P.panel_multi({p + "attn1.to_q.weight", p + "attn1.to_k.weight", p + "attn1.to_v.weight"}, {}, 0, inner, ch, 1, 0,
                              nullptr, g1, b1);
        t.out = P.panel(p + "attn1.to_out.0.weight", p + "attn1.to_out.0.bias", 0, ch, inner, 1);
        t.ff1 = P.panel(p + "ff.net.0.proj.weight", p + "ff.net.0.proj.bias", 0, 4 * ch, ch, 1, 0, nullptr, g3, b3);
        t.alpha_exp = P.vec(p + "ff.net.0.alpha_exp", 4 * ch);
        t.inv_beta = P.vec(p + "ff.net.0.inv_beta", 4 * ch);
        t.ff2 = P.panel(p + "ff.net.2.weight", p + "ff.net.2.bias", 0, ch, 4 * ch, 1);
        D.tb.push_back(t);
    };
    int co = cin0;
    for (int i = 0; i < nl; ++i) {
        const int ci = co;
        co = g.dec_channels[i];
        resnet(R + S("down_blocks.", i, ".0."), ci, co);
        for (int j = 0; j < g.dec_n_blocks; ++j) tblock(R + S("down_blocks.", i, ".1.") + std::to_string(j) + ".", co);
        if (i < nl - 1) D.down.push_back(P.panel(R + S("down_blocks.", i, ".2.conv.weight"), R + S("down_blocks.", i, ".2.conv.bias"), 1, co, co, 3));
        else D.down.push_back(P.panel(R + S("down_blocks.", i, ".2.weight"), R + S("down_blocks.", i, ".2.bias"), 1, co, co, 3));
    }
    const int cmid = g.dec_channels[nl - 1];
    for (int i = 0; i < g.dec_mid_blocks; ++i) {
        resnet(R + S("mid_blocks.", i, ".0."), cmid, cmid);
        for (int j = 0; j < g.dec_n_blocks; ++j) tblock(R + S("mid_blocks.", i, ".1.") + std::to_string(j) + ".", cmid);
    }
    for (int i = 0; i < nl; ++i) {       // up path: channels reversed + channels[0]
        const int ci = g.dec_channels[nl - 1 - i];
        const int cu = (i + 1 < nl) ? g.dec_channels[nl - 2 - i] : g.dec_channels[0];
        resnet(R + S("up_blocks.", i, ".0."), 2 * ci, cu);
        for (int j = 0; j < g.dec_n_blocks; ++j) tblock(R + S("up_blocks.", i, ".1.") + std::to_string(j) + ".", cu);
        if (i < nl - 1) {
            // ConvTranspose1d(k4, s2, p1): out[2j] = W1.x[j] + W3.x[j-1];  out[2j+1] = W0.x[j+1] + W2.x[j]
            const int even[2] = {1, 3}, odd[2] = {0, 2};
            D.up_even.push_back(P.panel(R + S("up_blocks.", i, ".2.conv.weight"), R + S("up_blocks.", i, ".2.conv.bias"), 2, cu, cu, 2, 4, even));
            D.up_odd.push_back(P.panel(R + S("up_blocks.", i, ".2.conv.weight"), R + S("up_blocks.", i, ".2.conv.bias"), 2, cu, cu, 2, 4, odd));
        } else {
            D.up_last = P.panel(R + S("up_blocks.", i, ".2.weight"), R + S("up_blocks.", i, ".2.bias"), 1, cu, cu, 3);
        }
    }
    const int cfin = g.dec_channels[0];
    D.final_conv = P.panel(R + "final_block.block.0.weight", R + "final_block.block.0.bias", 1, cfin, cfin, 3);
    D.fgn_g = P.vec(R + "final_block.block.1.weight", cfin);
    D.fgn_b = P.vec(R + "final_block.block.1.bias", cfin);
    D.final_proj = P.panel(R + "final_proj.weight", R + "final_proj.bias", 1, g.n_feats, cfin, 1);
    if (P.ok) D.fgn_bs = P.bias_group_stats(D.final_conv, 8);
    // per-ResNet Linear(Mish(t)) stacked into one [sum(cout), temb] panel (rows of different blocks may differ in count)
    {
        int total = 0;
        for (size_t i = 0; i < D.res.size(); ++i) { D.res[i].tb_off = total; total += mlp_n[i]; }
        D.tb_total = total;
        Panel p;
        p.N = total; p.C = temb; p.ntaps = 1; p.ktap = round_up(temb, P.kq); p.has_bias = true;
        p.w = P.alloc((size_t)round_up(total, GEMM_BN) * p.ktap);
        p.b = P.alloc(round_up(total, GEMM_BN));
        for (size_t i = 0; i < D.res.size() && P.ok; ++i) {
            const auto* w = P.get(mlp_w[i], (size_t)mlp_n[i] * temb);
            const auto* b = P.get(mlp_b[i], mlp_n[i]);
            if (!w || !b) break;
            for (int n = 0; n < mlp_n[i] && !dry; ++n) {
                std::memcpy(&c->image[p.w + (size_t)(D.res[i].tb_off + n) * p.ktap], &(*w)[(size_t)n * temb], temb * sizeof(float));
                c->image[p.b + D.res[i].tb_off + n] = (*b)[n];
            }
        }
        if (P.ok) P.add_planes(p);
        D.tmlp = p;
    }
    // fragment streams of the transformer blocks' row-local chains (tblock_chain.hip): fp16-split arithmetic, P16 flow only
    if (P.ok && c->chain_on && c->gemm_terms == 2 && !c->half16 && !c->fast16 && c->p16_on && g.dec_head_dim == 64) {
        const int nb = g.dec_n_blocks;
        for (size_t k = 0; k < D.tb.size(); ++k) {
            TBlockW& t = D.tb[k];
            const int C = t.out.N, nq = ((int)(k % nb) + 1 < nb) ? D.tb[k + 1].qkv.N : 0;
            const int ch = (C == 384 && c->chain_ch == 256) ? 256 : 128;
            if (!chain_supported(C, inner, nq) || t.ff1.N != 4 * C || t.ff1.ktap != C || t.ff2.ktap != 4 * C || t.out.ktap != inner) continue;
            if (nq && D.tb[k + 1].qkv.ktap != C) continue;
            t.chain_frags = chain_stream_frags(C, inner, ch, nq);
            t.chain_ch = ch;
            t.chain_nqkv = nq;
            t.next = nq ? (int)k + 1 : -1;
            t.chain = P.alloc((size_t)t.chain_frags * CHAIN_WAVES * 256);
            if (c->pair_on && chain_supported_pair(C, inner, ch, nq)) {
                t.chain_pair_frags = chain_stream_frags_pair(C, inner, ch, nq);
                t.chain_pair = P.alloc((size_t)t.chain_pair_frags * 2 * CHAIN_WAVES * 256);
                if (!dry) chain_stream_pack_pair(C, inner, ch, nq, &c->image[t.out.w], &c->image[t.ff1.w], &c->image[t.ff2.w],
                                                 nq ? &c->image[D.tb[k + 1].qkv.w] : nullptr, reinterpret_cast<uint16_t*>(&c->image[t.chain_pair]),
                                                 &c->weights_saturate);
            }
            t.chain_consts = P.alloc((size_t)18 * C);
            if (!dry) {
                chain_stream_pack(C, inner, ch, nq, &c->image[t.out.w], &c->image[t.ff1.w], &c->image[t.ff2.w],
                                  nq ? &c->image[D.tb[k + 1].qkv.w] : nullptr, reinterpret_cast<uint16_t*>(&c->image[t.chain]),
                                  &c->weights_saturate);
                float* cc = &c->image[t.chain_consts];          // the chain kernel's column constants as one block (kernels.h)
                std::memcpy(cc, &c->image[t.ff1.wsum], (size_t)4 * C * sizeof(float));
                std::memcpy(cc + 4 * C, &c->image[t.ff1.b], (size_t)4 * C * sizeof(float));
                std::memcpy(cc + 8 * C, &c->image[t.alpha_exp.off], (size_t)4 * C * sizeof(float));
                std::memcpy(cc + 12 * C, &c->image[t.inv_beta.off], (size_t)4 * C * sizeof(float));
                std::memcpy(cc + 16 * C, &c->image[t.out.b], (size_t)C * sizeof(float));
                std::memcpy(cc + 17 * C, &c->image[t.ff2.b], (size_t)C * sizeof(float));
            }
        }
    }
    if (!P.ok) { set_error(P.why); return -1; }
    c->packed = true;
    return 0;
}

static inline const float* W(const mtts_ctx* c, size_t off) { return c->d_image + off; }

// GEMM arithmetic of new contexts (gemm_f32.hip): 2 (default) fp16 two-term split with scaled residual, 6 bf16 three-term
// split (both fp32-equivalent), 0 native fp32 MFMA, 3 bf16 two-term split (looser, opt-in).  MTTS_GEMM_TERMS overrides.
static int default_gemm_terms() {
    const char* e = getenv("MTTS_GEMM_TERMS");
    if (!e) return 2;
    const int t = atoi(e);
    return (t == 0 || t == 2 || t == 3 || t == 6) ? t : 2;
}

static void panel_args(const mtts_ctx* c, const Panel& p, GemmArgs& a) {
    a.w = W(c, p.w);
    a.terms = c->gemm_terms;
    a.w16 = c->gemm_terms ? static_cast<const void*>(W(c, p.w16)) : nullptr;
    a.bias = p.has_bias ? W(c, p.b) : nullptr;
    a.wsum = c->gemm_terms == 2 ? W(c, p.wsum) : nullptr;
    a.fast16 = c->fast16;
    a.w16h = (c->half16 && p.wh16) ? static_cast<const void*>(W(c, p.wh16)) : nullptr;
    a.N = p.N;
    a.ntaps = p.ntaps;
    a.ktap = p.ktap;
}
static void rows_plain(GemmArgs& a, int B, int T) {
    a.B = B; a.T_in = T; a.T_out = T; a.in_stride = 1;
    a.out_T = T; a.out_stride = 1; a.out_off = 0;
}
static void taps_centered(GemmArgs& a, int k) {
    for (int j = 0; j < k; ++j) a.tap_off[j] = j - k / 2;
}

// ================================================================================================ decoder
struct DecBufs {
    int B = 0, T = 0, nl = 0;
    std::vector<int> Tl;                 // frames per level
    std::vector<float*> mask;            // [B*T_l]
    std::vector<float*> bufA, bufB, skip;
    float *Y = nullptr, *Hh = nullptr, *Rr = nullptr, *QKV = nullptr, *ATT = nullptr, *FF = nullptr;
    float *mean = nullptr, *rstd = nullptr, *gnp = nullptr, *lnp = nullptr;
    float* gns = nullptr;                // GroupNorm tile statistics left by the conv GEMMs' epilogues (P16 decoder)
    _Float16* X16 = nullptr;             // P16 image of the residual stream x (the LayerNorm'd projections' LDS-DMA source)
    // P16 decoder (decoder_eval_p16): every conv / projection input exists as a P16 image, written by its producer
    bool p16 = false;
    std::vector<_Float16*> A16, B16, S16;   // twins of bufA / bufB / skip per level (masked rows)
    _Float16 *H16 = nullptr, *XM16 = nullptr;   // Block1D output (conv2 / final projection input); masked x|mu|0 state
    float *xmu = nullptr, *xmu2 = nullptr, *vel[4] = {nullptr, nullptr, nullptr, nullptr};
    float *TS = nullptr, *T1 = nullptr, *T2 = nullptr, *T3 = nullptr, *TB = nullptr;
    int ldx = 0, ldv = 0;
    int ew = 2;                          // halves per image element: 2 = P16 (head + residual), 1 = H16 (16-bit storage mode)
    // frame tables (kernels.h FrameTableArgs), per level: null when every utterance owns all T rows
    int T_true = 0;                      // the reference's padded length; T above is the rows per utterance actually held
    bool folded = false;
    std::vector<int*> nrows, nextra;     // rows in the statistics / attention keys; closed-form bias-row copies
    std::vector<float*> kbias;           // additive attention key bias (= mask when not folded)
    const int* nr(int l) const { return tables ? nrows[l] : nullptr; }
    const int* ne(int l) const { return folded ? nextra[l] : nullptr; }
    const float* kb(int l) const { return folded ? kbias[l] : mask[l]; }
    bool tables = false;
    unsigned int* pair_flag = nullptr;   // pair form of the chain launch: one flag per (row tile, half), zeroed per call
    bool qkv_ready = false;              // the previous block's chain launch already left this block's q|k|v image in QKV
};

// Transformer blocks of width C run on P16 images (gemm_p16.hip, attention P16 I/O) when the context computes in the
// fp16-split mode and the shapes allow whole 32-channel groups and 64-wide heads; MTTS_P16=0 (read at mtts_create) keeps
// the fp32-operand path.
static bool p16_blocks(const mtts_ctx* c, int C) {
    const mtts_config& g = c->cfg;
    return c->p16_on && c->gemm_terms == 2 && (C % 64) == 0 && g.dec_head_dim == 64;
}

// The whole estimator runs on P16 images (decoder_eval_p16) when every level qualifies and there is at least one transformer
// block per ResNet (the ResNet output then always feeds a LayerNorm'd projection first).
static bool p16_decoder(const mtts_ctx* c) {
    const mtts_config& g = c->cfg;
    if (g.dec_n_blocks < 1 || (g.n_feats & 1)) return false;
    for (int l = 0; l < g.dec_levels; ++l)
        if (!p16_blocks(c, g.dec_channels[l])) return false;
    return true;
}

static int plan_decoder(const mtts_ctx* c, int B, int T, int max_evals, int n_state, int n_vel, WS& ws, DecBufs& d) {
    const mtts_config& g = c->cfg;
    d.B = B; d.T = T; d.nl = g.dec_levels;
    if (T % (1 << (d.nl - 1))) { set_error("T must be a multiple of 2^(levels-1) (reference utils/model.py:15-21)"); return -1; }
    int cmax = 0;
    for (int i = 0; i < d.nl; ++i) cmax = std::max(cmax, g.dec_channels[i]);
    const int inner = g.dec_heads * g.dec_head_dim;
    const size_t M0 = (size_t)B * T;
    d.Tl.resize(d.nl);
    d.mask.resize(d.nl); d.bufA.resize(d.nl); d.bufB.resize(d.nl); d.skip.resize(d.nl);
    (void)ws.bytes(256);                 // header: the call's range flag (begin_call)
    d.pair_flag = static_cast<unsigned int*>(ws.bytes(2048));
    d.nrows.resize(d.nl); d.nextra.resize(d.nl); d.kbias.resize(d.nl);
    for (int l = 0; l < d.nl; ++l) {
        d.Tl[l] = T >> l;
        const size_t Ml = (size_t)B * d.Tl[l];
        d.mask[l] = ws.f(Ml);
        d.kbias[l] = ws.f(Ml);
        d.nrows[l] = reinterpret_cast<int*>(ws.f(B));
        d.nextra[l] = reinterpret_cast<int*>(ws.f(B));
        d.bufA[l] = ws.f(Ml * cmax);
        d.bufB[l] = ws.f(Ml * cmax);
        d.skip[l] = ws.f(Ml * cmax);
    }
    d.Y = ws.f(M0 * cmax); d.Hh = ws.f(M0 * cmax); d.Rr = ws.f(M0 * cmax);
    d.QKV = ws.f(M0 * 3 * inner); d.ATT = ws.f(M0 * inner); d.FF = ws.f(M0 * 4 * cmax);
    d.mean = ws.f(M0); d.rstd = ws.f(M0);
    d.lnp = ws.f(M0 * (size_t)((cmax + 63) / 64) * 2);
    d.X16 = reinterpret_cast<_Float16*>(ws.f(M0 * round_up(cmax, 32)));
    d.p16 = p16_decoder(c);
    if (d.p16) {
        d.A16.resize(d.nl); d.B16.resize(d.nl); d.S16.resize(d.nl);
        for (int l = 0; l < d.nl; ++l) {
            const size_t Ml = (size_t)B * d.Tl[l];
            d.A16[l] = reinterpret_cast<_Float16*>(ws.f(Ml * cmax));
            d.B16[l] = reinterpret_cast<_Float16*>(ws.f(Ml * cmax));
            d.S16[l] = reinterpret_cast<_Float16*>(ws.f(Ml * cmax));
        }
        d.H16 = reinterpret_cast<_Float16*>(ws.f(M0 * cmax));
        d.XM16 = reinterpret_cast<_Float16*>(ws.f(M0 * round_up(2 * g.n_feats, 64)));
    }
    d.gnp = ws.f((size_t)B * gn_chunks_max(T) * 8 * 2);
    d.gns = ws.f((M0 / 32 + 2) * 2 * (size_t)((cmax + 63) / 64) * 8);
    d.ew = (d.p16 && c->half16) ? 1 : 2;
    d.ldx = round_up(2 * g.n_feats, c->half16 ? 64 : GEMM_BK);
    d.ldv = round_up(g.n_feats, 4);
    d.xmu = ws.f(M0 * d.ldx);
    if (n_state > 1) d.xmu2 = ws.f(M0 * d.ldx);
    for (int i = 0; i < n_vel; ++i) d.vel[i] = ws.f(M0 * d.ldv);
    const int temb = g.dec_channels[0] * 4;
    d.TS = ws.f((size_t)max_evals * 2 * g.n_feats);
    d.T1 = ws.f((size_t)max_evals * temb); d.T2 = ws.f((size_t)max_evals * temb); d.T3 = ws.f((size_t)max_evals * temb);
    d.TB = ws.f((size_t)max_evals * c->dec.tb_total);
    return 0;
}

// SinusoidalPosEmb + TimestepEmbedding + every ResNet's Linear(Mish(t)) for all evaluation times at once
// (reference decoder.py:14-29,107-119,51,60): they depend on t only, so the whole ODE grid is done before the loop.
static int time_embed(mtts_ctx* c, DecBufs& d, const TimeVals& tv, int nt, hipStream_t s) {
    const mtts_config& g = c->cfg;
    const DecW& D = c->dec;
    const int cin0 = 2 * g.n_feats, temb = g.dec_channels[0] * 4;
    LAUNCH(c, 2, 0, s, launch_time_sinusoid(W(c, D.freqs.off), tv, nt, cin0 / 2, 1000.0f, d.TS, s));
    GemmArgs a;
    panel_args(c, D.t1, a); rows_plain(a, nt, 1);
    a.a0 = d.TS; a.lda0 = cin0; a.c0 = cin0; a.act = ACT_SILU; a.out = d.T1; a.ldc = temb;
    RET_IF(run_gemm(c, a, s));
    GemmArgs b;
    panel_args(c, D.t2, b); rows_plain(b, nt, 1);
    b.a0 = d.T1; b.lda0 = temb; b.c0 = temb; b.out = d.T2; b.ldc = temb;
    RET_IF(run_gemm(c, b, s));
    LAUNCH(c, 2, 0, s, launch_unary(d.T2, d.T3, (int64_t)nt * temb, 1, s));
    GemmArgs m;
    panel_args(c, D.tmlp, m); rows_plain(m, nt, 1);
    m.a0 = d.T3; m.lda0 = temb; m.c0 = temb; m.out = d.TB; m.ldc = D.tb_total;
    RET_IF(run_gemm(c, m, s));
    return 0;
}

// ResnetBlock1D.forward (reference decoder.py:58-63) on channels-last rows; input = up to two channel segments.
static int resnet_block(mtts_ctx* c, DecBufs& d, const ResnetW& r, const float* in0, int ld0, int c0, const float* in1, int ld1,
                        int c1, int lvl, const float* tbias, float* out, bool emit_stats, hipStream_t s) {
    const int B = d.B, T = d.Tl[lvl], C = r.cout;
    const float* mask = d.mask[lvl];
    GemmArgs a;
    panel_args(c, r.conv1, a); rows_plain(a, B, T); taps_centered(a, 3);
    a.a0 = in0; a.lda0 = ld0; a.c0 = c0; a.a1 = in1; a.lda1 = ld1; a.c1 = c1; a.a_mask = mask;
    a.out = d.Y; a.ldc = C;
    RET_IF(run_gemm(c, a, s));
    LAUNCH(c, 2, 0, s, launch_gn_partial(d.Y, B, T, C, 8, d.gnp, s, d.nr(lvl)));
    GnApplyArgs g1;
    g1.y = d.Y; g1.partial = d.gnp; g1.gamma = W(c, r.gn1_g.off); g1.beta = W(c, r.gn1_b.off); g1.mask = mask; g1.nrows = d.nr(lvl);
    if (d.folded) { g1.nextra = d.ne(lvl); g1.bias_stats = W(c, r.gn1_bs.off); }
    g1.chbias = tbias; g1.out = d.Hh; g1.B = B; g1.T = T; g1.C = C;
    RET_IF(run_gn_apply(c, g1, s));
    GemmArgs b;
    panel_args(c, r.conv2, b); rows_plain(b, B, T); taps_centered(b, 3);
    b.a0 = d.Hh; b.lda0 = C; b.c0 = C; b.out = d.Y; b.ldc = C;      // Hh is already masked
    RET_IF(run_gemm(c, b, s));
    LAUNCH(c, 2, 0, s, launch_gn_partial(d.Y, B, T, C, 8, d.gnp, s, d.nr(lvl)));
    GemmArgs rc;
    panel_args(c, r.res, rc); rows_plain(rc, B, T);
    rc.a0 = in0; rc.lda0 = ld0; rc.c0 = c0; rc.a1 = in1; rc.lda1 = ld1; rc.c1 = c1; rc.a_mask = mask;
    rc.out = d.Rr; rc.ldc = C;
    RET_IF(run_gemm(c, rc, s));
    GnApplyArgs g2;
    g2.y = d.Y; g2.partial = d.gnp; g2.gamma = W(c, r.gn2_g.off); g2.beta = W(c, r.gn2_b.off); g2.mask = mask; g2.nrows = d.nr(lvl);
    if (d.folded) { g2.nextra = d.ne(lvl); g2.bias_stats = W(c, r.gn2_bs.off); }
    g2.res = d.Rr; g2.ldr = C; g2.out = out; g2.B = B; g2.T = T; g2.C = C;
    if (emit_stats && (C % 64) == 0) {       // for the first transformer block: LayerNorm moments and, in P16 mode, x's image
        g2.stats_out = d.lnp;
        if (p16_blocks(c, C)) { g2.out16 = d.X16; g2.ld16 = d.ew * C; }
    }
    RET_IF(run_gn_apply(c, g2, s));
    return 0;
}

// BasicTransformerBlock.forward (reference transformer.py:230-303, self-attention only), in place on x [B*T, C].
// LayerNorm statistics travel with the data: the GEMM that writes x (attention out-projection, second FF projection)
// leaves per-row partial moments of its 64-column slices behind (stats_out) and the next projection merges them in its
// prologue; the ResNet block's last kernel (gn_apply) does the same for the first LayerNorm after it.  The row_stats
// kernel only runs for widths that are not a multiple of 64 (the tiny test model).
// last16 / last16_mask (P16 decoder): where the LAST block of a run leaves the masked P16 image of x for the convs.
static int transformer_block(mtts_ctx* c, DecBufs& d, const TBlockW& t, float* x, int C, int lvl, bool have_stats, bool emit_stats,
                             hipStream_t s, _Float16* last16 = nullptr) {
    const mtts_config& g = c->cfg;
    const int B = d.B, T = d.Tl[lvl], M = B * T, inner = g.dec_heads * g.dec_head_dim;
    const bool fuse = (C % 64) == 0;
    if (p16_blocks(c, C) && have_stats) {
        // P16 flow: q|k|v, the attention output and the FF hidden layer exist only as P16 images (same bytes as fp32, in the
        // same buffers); x stays fp32 (the residual stream) with a P16 copy for the two LayerNorm'd projections.
        _Float16* QKV16 = reinterpret_cast<_Float16*>(d.QKV);
        _Float16* ATT16 = reinterpret_cast<_Float16*>(d.ATT);
        _Float16* FF16 = reinterpret_cast<_Float16*>(d.FF);
        // the row-local part as one launch (tblock_chain.hip) when the stream was packed and the batch is large enough that a
        // workgroup per QB rows fills the chip: every workgroup streams ALL of the chain's weights (~7 MB at width 384), which
        // only pays when their cost is shared by many rows per CU
        static const int chain_only = [] { const char* e = getenv("MTTS_CHAIN_ONLY"); return !e ? 0 : (e[0] == 'f' ? 1 : 2); }();   // diagnostic
        const bool chain = t.chain_frags > 0 && d.p16 && !c->half_now && M >= c->chain_min_rows && (emit_stats ? t.chain_nqkv > 0 : true) &&
                           (chain_only == 0 || (chain_only == 1) == emit_stats);
        if (!d.qkv_ready) {
            GemmArgs q;
            panel_args(c, t.qkv, q); rows_plain(q, B, T);
            q.a16_0 = d.X16; q.lda16_0 = d.ew * C; q.c0 = C; q.a_part = d.lnp; q.a_nparts = C / 64;
            q.out16 = QKV16; q.ld16 = 3 * d.ew * inner; q.out_lscale = 1.0f;
            RET_IF(run_gemm(c, q, s));
        }
        d.qkv_ready = false;
        AttnArgs at;
        at.qkv16 = QKV16; at.ld16 = 3 * d.ew * inner; at.out16 = ATT16; at.ldo16 = d.ew * inner; at.mask = d.kb(lvl);
        at.B = B; at.T = T; at.H = g.dec_heads; at.D = g.dec_head_dim;
        at.scale = 1.0f / sqrtf((float)g.dec_head_dim); at.mask_mode = 0; at.klen = d.nr(lvl); at.fast16 = c->fast16;
        RET_IF(run_attn(c, at, s));
        // below that row count: the pair form -- two workgroups of one XCD per 48-row tile, each streaming half of the FeedForward
        // and of the q|k|v passes -- while all of them (and the prefetchers) are resident at once
        const int tiles48 = (M + 47) / 48;
        static const int pair_min = [] { const char* e = getenv("MTTS_CHAIN_PAIR_MIN_ROWS"); return e ? atoi(e) : 3000; }();      // (3864 rows: -0.3..0.5 ms per step, 2576 rows: +0.3; profiles/r03_pair_ab.log)
        const bool pair = !chain && c->pair_on && t.chain_pair_frags > 0 && d.p16 && !c->half_now && d.pair_flag && M >= pair_min &&
                          16 * ((tiles48 + 7) / 8) + 16 <= 256 && (emit_stats ? t.chain_nqkv > 0 : true);
        if (chain || pair) {
            ChainArgs a;
            a.M = M; a.C = C; a.inner = inner;
            a.att16 = ATT16; a.ld_att = 2 * inner;
            a.x16 = d.X16; a.ld_x = 2 * C;
            a.wstream = reinterpret_cast<const _Float16*>(W(c, t.chain)); a.stream_frags = t.chain_frags;
            a.consts = W(c, t.chain_consts);
            if (emit_stats) {                 // another block follows: its q|k|v leaves this launch, x stays unmasked
                const TBlockW& nx = c->dec.tb[t.next];
                a.b_qkv = W(c, nx.qkv.b); a.wsum_qkv = W(c, nx.qkv.wsum); a.n_qkv = nx.qkv.N;
                a.qkv16 = QKV16; a.ld_qkv = 2 * nx.qkv.N;
                a.x_out = d.X16; a.ld_out = 2 * C;
                d.qkv_ready = true;
            } else if (last16) { a.x_out = last16; a.ld_out = 2 * C; a.x_out_mask = d.mask[lvl]; }
            else { a.x_out = d.X16; a.ld_out = 2 * C; }
            a.ch = t.chain_ch;
            { int pf_unused = 0; chain_plan(M, a.ch, c->chain_qb, &a.qb, &pf_unused); }
            if (pair) {
                a.pair = 1; a.qb = 48;
                a.wstream = reinterpret_cast<const _Float16*>(W(c, t.chain_pair)); a.stream_frags = t.chain_pair_frags;
                a.pair_part = d.FF;              // (the tiled path's hidden image: unused by a chain launch)
                a.pair_flag = d.pair_flag;
                a.pair_epoch = ++c->pair_epoch;
                if (c->pair_epoch == 0) a.pair_epoch = ++c->pair_epoch;
            }
#ifdef MTTS_CHAIN_VERIFY
            RET_IF(run_chain_verified(c, a, FF16, s));
#else
            RET_IF(run_chain(c, a, s));
#endif
            return 0;
        }
        GemmArgs o;
        panel_args(c, t.out, o); rows_plain(o, B, T);
        o.a16_0 = ATT16; o.lda16_0 = d.ew * inner; o.c0 = inner;
        o.out16 = d.X16; o.ld16 = d.ew * C; o.stats_out = d.lnp;
        if (d.p16) { o.res16 = d.X16; o.ldr16 = d.ew * C; }      // P16 decoder: the residual stream exists only as its image
        else { o.res = x; o.ldr = C; o.out = x; o.ldc = C; }
        RET_IF(run_gemm(c, o, s));
        GemmArgs f1;
        panel_args(c, t.ff1, f1); rows_plain(f1, B, T);
        f1.a16_0 = d.X16; f1.lda16_0 = d.ew * C; f1.c0 = C; f1.a_part = d.lnp; f1.a_nparts = C / 64; f1.act = ACT_SNAKE;
        f1.p0 = W(c, t.alpha_exp.off); f1.p1 = W(c, t.inv_beta.off); f1.out16 = FF16; f1.ld16 = 4 * d.ew * C;
        RET_IF(run_gemm(c, f1, s));
        GemmArgs f2;
        panel_args(c, t.ff2, f2); rows_plain(f2, B, T);
        f2.a16_0 = FF16; f2.lda16_0 = 4 * d.ew * C; f2.c0 = 4 * C;
        if (d.p16) { f2.res16 = d.X16; f2.ldr16 = d.ew * C; }
        else { f2.res = x; f2.ldr = C; f2.out = x; f2.ldc = C; }
        if (emit_stats) { f2.stats_out = d.lnp; f2.out16 = d.X16; f2.ld16 = d.ew * C; }
        else if (last16) { f2.out16 = last16; f2.ld16 = d.ew * C; f2.out16_mask = d.mask[lvl]; }
        RET_IF(run_gemm(c, f2, s));
        return 0;
    }
    GemmArgs q;
    panel_args(c, t.qkv, q); rows_plain(q, B, T);
    q.a0 = x; q.lda0 = C; q.c0 = C; q.out = d.QKV; q.ldc = 3 * inner;
    if (fuse && have_stats) { q.a_part = d.lnp; q.a_nparts = C / 64; }
    else {
        LAUNCH(c, 2, 0, s, launch_row_stats(x, M, C, C, 1e-5f, d.mean, d.rstd, s));
        q.a_mean = d.mean; q.a_rstd = d.rstd;
    }
    RET_IF(run_gemm(c, q, s));
    AttnArgs at;
    at.qkv = d.QKV; at.mask = d.kb(lvl); at.out = d.ATT; at.B = B; at.T = T; at.H = g.dec_heads; at.D = g.dec_head_dim;
    at.scale = 1.0f / sqrtf((float)g.dec_head_dim); at.mask_mode = 0; at.klen = d.nr(lvl); at.fast16 = c->fast16;
    RET_IF(run_attn(c, at, s));
    GemmArgs o;
    panel_args(c, t.out, o); rows_plain(o, B, T);
    o.a0 = d.ATT; o.lda0 = inner; o.c0 = inner; o.res = x; o.ldr = C; o.out = x; o.ldc = C;
    if (fuse) o.stats_out = d.lnp;
    RET_IF(run_gemm(c, o, s));
    GemmArgs f1;
    panel_args(c, t.ff1, f1); rows_plain(f1, B, T);
    f1.a0 = x; f1.lda0 = C; f1.c0 = C; f1.act = ACT_SNAKE;
    f1.p0 = W(c, t.alpha_exp.off); f1.p1 = W(c, t.inv_beta.off); f1.out = d.FF; f1.ldc = 4 * C;
    if (fuse) { f1.a_part = d.lnp; f1.a_nparts = C / 64; }
    else {
        LAUNCH(c, 2, 0, s, launch_row_stats(x, M, C, C, 1e-5f, d.mean, d.rstd, s));
        f1.a_mean = d.mean; f1.a_rstd = d.rstd;
    }
    RET_IF(run_gemm(c, f1, s));
    GemmArgs f2;
    panel_args(c, t.ff2, f2); rows_plain(f2, B, T);
    f2.a0 = d.FF; f2.lda0 = 4 * C; f2.c0 = 4 * C; f2.res = x; f2.ldr = C; f2.out = x; f2.ldc = C;
    if (fuse && emit_stats) f2.stats_out = d.lnp;
    RET_IF(run_gemm(c, f2, s));
    return 0;
}

struct FinalOut {   // where the masked velocity goes: out = v * scale (+ res)
    float* out; int ldc; const float* res; int ldr; float scale;
};

// Decoder.forward (reference decoder.py:359-426) for evaluation `ev` (row of the precomputed time biases).
// xin: channels-last state [B*T, ldx] holding x | mu.
static int decoder_eval_p16(mtts_ctx* c, DecBufs& d, const float* xin, int ev, const FinalOut& fo, hipStream_t s);
static int decoder_eval(mtts_ctx* c, DecBufs& d, const float* xin, int ev, const FinalOut& fo, hipStream_t s) {
    if (d.p16) {
        c->half_now = d.ew == 1;          // 16-bit storage mode: the estimator's images are H16 (kernels.h GemmArgs::half16)
        const int r = decoder_eval_p16(c, d, xin, ev, fo, s);
        c->half_now = false;
        return r;
    }
    const mtts_config& g = c->cfg;
    const DecW& D = c->dec;
    const int nl = d.nl, nb = g.dec_n_blocks, B = d.B;
    const float* tb = d.TB + (size_t)ev * D.tb_total;
    size_t ri = 0, ti = 0;
    const float* cur = xin;
    int cur_ld = d.ldx, cur_c = 2 * g.n_feats;
    // ---- down path
    for (int l = 0; l < nl; ++l) {
        const ResnetW& r = D.res[ri++];
        RET_IF(resnet_block(c, d, r, cur, cur_ld, cur_c, nullptr, 0, 0, l, tb + r.tb_off, d.skip[l], nb > 0, s));
        for (int j = 0; j < nb; ++j) RET_IF(transformer_block(c, d, D.tb[ti++], d.skip[l], r.cout, l, true, j + 1 < nb, s));
        GemmArgs a;
        panel_args(c, D.down[l], a);
        taps_centered(a, 3);
        a.a0 = d.skip[l]; a.lda0 = r.cout; a.c0 = r.cout; a.a_mask = d.mask[l];
        a.B = B; a.T_in = d.Tl[l];
        if (l < nl - 1) {   // Downsample1D: Conv1d(k3, s2, p1) (reference decoder.py:66-72)
            a.T_out = d.Tl[l + 1]; a.in_stride = 2; a.out_T = d.Tl[l + 1];
            a.out = d.bufA[l + 1];
        } else {            // last level: Conv1d(k3, p1) (reference decoder.py:252-254)
            a.T_out = d.Tl[l]; a.out_T = d.Tl[l];
            a.out = d.bufA[l];
        }
        a.ldc = r.cout;
        RET_IF(run_gemm(c, a, s));
        cur = a.out; cur_ld = r.cout; cur_c = r.cout;
    }
    // ---- mid blocks at the coarsest level
    const int lm = nl - 1;
    for (int i = 0; i < g.dec_mid_blocks; ++i) {
        const ResnetW& r = D.res[ri++];
        float* dst = (cur == d.bufA[lm]) ? d.bufB[lm] : d.bufA[lm];
        RET_IF(resnet_block(c, d, r, cur, cur_ld, cur_c, nullptr, 0, 0, lm, tb + r.tb_off, dst, nb > 0, s));
        for (int j = 0; j < nb; ++j) RET_IF(transformer_block(c, d, D.tb[ti++], dst, r.cout, lm, true, j + 1 < nb, s));
        cur = dst; cur_ld = r.cout; cur_c = r.cout;
    }
    // ---- up path
    for (int i = 0; i < nl; ++i) {
        const int l = nl - 1 - i;
        const ResnetW& r = D.res[ri++];
        const int cskip = g.dec_channels[l];
        float* dst = (cur == d.bufA[l]) ? d.bufB[l] : d.bufA[l];
        RET_IF(resnet_block(c, d, r, cur, cur_ld, cur_c, d.skip[l], cskip, cskip, l, tb + r.tb_off, dst, nb > 0, s));
        for (int j = 0; j < nb; ++j) RET_IF(transformer_block(c, d, D.tb[ti++], dst, r.cout, l, true, j + 1 < nb, s));
        if (i < nl - 1) {   // Upsample1D: ConvTranspose1d(k4, s2, p1) as two phase GEMMs (reference decoder.py:146)
            float* up = d.bufA[l - 1];
            for (int ph = 0; ph < 2; ++ph) {
                GemmArgs a;
                panel_args(c, ph == 0 ? D.up_even[i] : D.up_odd[i], a);
                a.a0 = dst; a.lda0 = r.cout; a.c0 = r.cout; a.a_mask = d.mask[l];
                a.B = B; a.T_in = d.Tl[l]; a.T_out = d.Tl[l]; a.in_stride = 1;
                a.tap_off[0] = ph == 0 ? 0 : 1;
                a.tap_off[1] = ph == 0 ? -1 : 0;
                a.out = up; a.ldc = r.cout; a.out_T = d.Tl[l - 1]; a.out_stride = 2; a.out_off = ph;
                RET_IF(run_gemm(c, a, s));
            }
            cur = up;
        } else {
            GemmArgs a;
            panel_args(c, D.up_last, a); rows_plain(a, B, d.Tl[l]); taps_centered(a, 3);
            a.a0 = dst; a.lda0 = r.cout; a.c0 = r.cout; a.a_mask = d.mask[l];
            float* o2 = (dst == d.bufA[l]) ? d.bufB[l] : d.bufA[l];
            a.out = o2; a.ldc = r.cout;
            RET_IF(run_gemm(c, a, s));
            cur = o2;
        }
        cur_ld = r.cout; cur_c = r.cout;
    }
    // ---- final Block1D + 1x1 projection + mask (reference decoder.py:423-426)
    const int C0 = g.dec_channels[0], T = d.T;
    GemmArgs a;
    panel_args(c, D.final_conv, a); rows_plain(a, B, T); taps_centered(a, 3);
    a.a0 = cur; a.lda0 = cur_ld; a.c0 = C0; a.a_mask = d.mask[0]; a.out = d.Y; a.ldc = C0;
    RET_IF(run_gemm(c, a, s));
    LAUNCH(c, 2, 0, s, launch_gn_partial(d.Y, B, T, C0, 8, d.gnp, s, d.nr(0)));
    GnApplyArgs ga;
    ga.y = d.Y; ga.partial = d.gnp; ga.gamma = W(c, D.fgn_g.off); ga.beta = W(c, D.fgn_b.off); ga.mask = d.mask[0]; ga.nrows = d.nr(0);
    if (d.folded) { ga.nextra = d.ne(0); ga.bias_stats = W(c, D.fgn_bs.off); }
    ga.out = d.Hh; ga.B = B; ga.T = T; ga.C = C0;
    RET_IF(run_gn_apply(c, ga, s));
    GemmArgs p;
    panel_args(c, D.final_proj, p); rows_plain(p, B, T);
    p.a0 = d.Hh; p.lda0 = C0; p.c0 = C0; p.out_mask = d.mask[0];
    p.out = fo.out; p.ldc = fo.ldc; p.res = fo.res; p.ldr = fo.ldr; p.out_scale = fo.scale;
    RET_IF(run_gemm(c, p, s));
    return 0;
}

// GroupNorm statistics from the conv GEMM's epilogue instead of a gn_partial pass over its output (gemm_epilogue.h): entries per
// wave tile and utterance part, so an utterance must be at least one wave tile long, and groups of >= 32 channels;
// MTTS_GN_FUSE=0 keeps the separate pass (A/B runs).  Returns the wave-tile height (the consumers' tile_rows) or 0.
static int gn_fuse_rows(const GemmArgs& a, int C, int G, int T) {
    static const bool on = [] { const char* e = getenv("MTTS_GN_FUSE"); return !(e && e[0] == '0'); }();
    if (!on || !a.a16_0 || a.fast16 || (C % 64) || (C % G) || (C / G) < 32 || ((C / G) & 7)) return 0;
    const int rows = gemm_p16_wave_rows(a);
    return T >= rows ? rows : 0;
}

// ---- P16 decoder: the same network with every GEMM on pre-split operands (gemm_p16.hip).  Each producer writes the P16
// image its consumers read (already multiplied by the frame mask where the reference masks the input): GroupNorm-apply,
// the GEMM epilogues, and one conversion pass for the ODE state.  fp32 copies exist only where an fp32 consumer remains
// (GroupNorm statistics of the conv outputs, the residual stream x, the ResNet skip sum).
static int resnet_block_p16(mtts_ctx* c, DecBufs& d, const ResnetW& r, const _Float16* in0, int c0, const _Float16* in1, int c1,
                            int lvl, const float* tbias, float* out, hipStream_t s) {
    const int B = d.B, T = d.Tl[lvl], C = r.cout;
    const float* mask = d.mask[lvl];
    GemmArgs a;
    panel_args(c, r.conv1, a); rows_plain(a, B, T); taps_centered(a, 3);
    a.a16_0 = in0; a.lda16_0 = d.ew * c0; a.c0 = c0; a.a16_1 = in1; a.lda16_1 = d.ew * c1; a.c1 = c1;
    a.out = d.Y; a.ldc = C;
    const int fr1 = gn_fuse_rows(a, C, 8, T);
    if (fr1) { a.gn_stats = d.gns; a.gn_groups = 8; a.gn_nrows = d.nr(lvl); }
    RET_IF(run_gemm(c, a, s));
    if (!fr1) LAUNCH(c, 2, 0, s, launch_gn_partial(d.Y, B, T, C, 8, d.gnp, s, d.nr(lvl)));
    GnApplyArgs g1;
    if (fr1) { g1.tile_stats = d.gns; g1.tile_rows = fr1; }
    g1.y = d.Y; g1.partial = d.gnp; g1.gamma = W(c, r.gn1_g.off); g1.beta = W(c, r.gn1_b.off); g1.mask = mask; g1.nrows = d.nr(lvl);
    if (d.folded) { g1.nextra = d.ne(lvl); g1.bias_stats = W(c, r.gn1_bs.off); }
    g1.chbias = tbias; g1.out16 = d.H16; g1.ld16 = d.ew * C; g1.B = B; g1.T = T; g1.C = C;      // already masked
    RET_IF(run_gn_apply(c, g1, s));
    GemmArgs b;
    panel_args(c, r.conv2, b); rows_plain(b, B, T); taps_centered(b, 3);
    b.a16_0 = d.H16; b.lda16_0 = d.ew * C; b.c0 = C; b.out = d.Y; b.ldc = C;
    const int fr2 = gn_fuse_rows(b, C, 8, T);
    if (fr2) { b.gn_stats = d.gns; b.gn_groups = 8; b.gn_nrows = d.nr(lvl); }
    RET_IF(run_gemm(c, b, s));
    if (!fr2) LAUNCH(c, 2, 0, s, launch_gn_partial(d.Y, B, T, C, 8, d.gnp, s, d.nr(lvl)));
    GemmArgs rc;
    panel_args(c, r.res, rc); rows_plain(rc, B, T);
    rc.a16_0 = in0; rc.lda16_0 = d.ew * c0; rc.c0 = c0; rc.a16_1 = in1; rc.lda16_1 = d.ew * c1; rc.c1 = c1;
    (void)out;                                                        // no fp32 copy: x lives on as its image only
    static const bool tail_on = [] { const char* e = getenv("MTTS_GN_TAIL"); return !(e && e[0] == '0'); }();   // A/B runs
    if (tail_on && fr2 && T >= 2 * gemm_p16_wave_rows(rc)) {      // a workgroup's rows in at most two utterances
        // The 1x1 residual conv finishes the block: its epilogue adds Mish(GroupNorm(conv2 output)) * mask from the tile
        // statistics conv2 left, and writes x's image + LayerNorm moments -- no gn_apply pass, no residual round trip.
        rc.gnr_y = d.Y; rc.gnr_stats = d.gns; rc.gnr_tile_rows = fr2; rc.gnr_groups = 8;
        rc.gnr_gamma = W(c, r.gn2_g.off); rc.gnr_beta = W(c, r.gn2_b.off); rc.gnr_mask = mask;
        if (d.folded) { rc.gnr_nextra = d.ne(lvl); rc.gnr_bias_stats = W(c, r.gn2_bs.off); }
        rc.out16 = d.X16; rc.ld16 = d.ew * C; rc.stats_out = d.lnp;
        RET_IF(run_gemm(c, rc, s));
        return 0;
    }
    rc.out = d.Rr; rc.ldc = C;
    RET_IF(run_gemm(c, rc, s));
    GnApplyArgs g2;
    if (fr2) { g2.tile_stats = d.gns; g2.tile_rows = fr2; }
    g2.y = d.Y; g2.partial = d.gnp; g2.gamma = W(c, r.gn2_g.off); g2.beta = W(c, r.gn2_b.off); g2.mask = mask; g2.nrows = d.nr(lvl);
    if (d.folded) { g2.nextra = d.ne(lvl); g2.bias_stats = W(c, r.gn2_bs.off); }
    g2.res = d.Rr; g2.ldr = C; g2.B = B; g2.T = T; g2.C = C;
    g2.stats_out = d.lnp; g2.out16 = d.X16; g2.ld16 = d.ew * C;          // unmasked: the first transformer block's LayerNorm input
    RET_IF(run_gn_apply(c, g2, s));
    return 0;
}

static int decoder_eval_p16(mtts_ctx* c, DecBufs& d, const float* xin, int ev, const FinalOut& fo, hipStream_t s) {
    const mtts_config& g = c->cfg;
    const DecW& D = c->dec;
    const int nl = d.nl, nb = g.dec_n_blocks, B = d.B;
    const float* tb = d.TB + (size_t)ev * D.tb_total;
    size_t ri = 0, ti = 0;
    // masked x | mu | zero padding as a P16 image (reference decoder.py:379: the first ResNet sees x * mask)
    LAUNCH(c, 2, 0, s, launch_to_p16(xin, d.ldx, d.mask[0], B * d.T, d.ldx, 2 * g.n_feats, d.XM16, d.ew * d.ldx, 2048.0f, s, c->cur_flag, d.ew == 1, d.ew == 1 && c->bf16));
    const _Float16* cur = d.XM16;
    int cur_c = d.ldx;
    // ---- down path
    for (int l = 0; l < nl; ++l) {
        const ResnetW& r = D.res[ri++];
        if (r.conv1.ktap != cur_c) { set_error("P16 decoder: unexpected ResNet input width"); return -1; }
        RET_IF(resnet_block_p16(c, d, r, cur, cur_c, nullptr, 0, l, tb + r.tb_off, d.skip[l], s));
        for (int j = 0; j < nb; ++j)
            RET_IF(transformer_block(c, d, D.tb[ti++], d.skip[l], r.cout, l, true, j + 1 < nb, s, d.S16[l]));
        GemmArgs a;
        panel_args(c, D.down[l], a);
        taps_centered(a, 3);
        a.a16_0 = d.S16[l]; a.lda16_0 = d.ew * r.cout; a.c0 = r.cout;
        a.B = B; a.T_in = d.Tl[l];
        const int lo = l < nl - 1 ? l + 1 : l;
        if (l < nl - 1) { a.T_out = d.Tl[l + 1]; a.in_stride = 2; a.out_T = d.Tl[l + 1]; }   // Downsample1D (reference decoder.py:66-72)
        else { a.T_out = d.Tl[l]; a.out_T = d.Tl[l]; }                                          // last level: Conv1d(k3, p1)
        a.out16 = d.A16[lo]; a.ld16 = d.ew * r.cout; a.out16_mask = d.mask[lo];
        RET_IF(run_gemm(c, a, s));
        cur = d.A16[lo]; cur_c = r.cout;
    }
    // ---- mid blocks at the coarsest level
    const int lm = nl - 1;
    float* xbuf = d.bufA[lm];
    _Float16* x16 = d.B16[lm];
    for (int i = 0; i < g.dec_mid_blocks; ++i) {
        const ResnetW& r = D.res[ri++];
        RET_IF(resnet_block_p16(c, d, r, cur, cur_c, nullptr, 0, lm, tb + r.tb_off, xbuf, s));
        for (int j = 0; j < nb; ++j) RET_IF(transformer_block(c, d, D.tb[ti++], xbuf, r.cout, lm, true, j + 1 < nb, s, x16));
        cur = x16; cur_c = r.cout;
        x16 = (x16 == d.B16[lm]) ? d.A16[lm] : d.B16[lm];
    }
    // ---- up path
    for (int i = 0; i < nl; ++i) {
        const int l = nl - 1 - i;
        const ResnetW& r = D.res[ri++];
        const int cskip = g.dec_channels[l];
        float* xb = d.bufA[l];
        _Float16* dst16 = (cur == d.B16[l]) ? d.A16[l] : d.B16[l];
        RET_IF(resnet_block_p16(c, d, r, cur, cur_c, d.S16[l], cskip, l, tb + r.tb_off, xb, s));
        for (int j = 0; j < nb; ++j) RET_IF(transformer_block(c, d, D.tb[ti++], xb, r.cout, l, true, j + 1 < nb, s, dst16));
        if (i < nl - 1) {   // Upsample1D: ConvTranspose1d(k4, s2, p1) as two phase GEMMs (reference decoder.py:146)
            _Float16* up16 = d.A16[l - 1];
            for (int ph = 0; ph < 2; ++ph) {
                GemmArgs a;
                panel_args(c, ph == 0 ? D.up_even[i] : D.up_odd[i], a);
                a.a16_0 = dst16; a.lda16_0 = d.ew * r.cout; a.c0 = r.cout;
                a.B = B; a.T_in = d.Tl[l]; a.T_out = d.Tl[l]; a.in_stride = 1;
                a.tap_off[0] = ph == 0 ? 0 : 1;
                a.tap_off[1] = ph == 0 ? -1 : 0;
                a.out16 = up16; a.ld16 = d.ew * r.cout; a.out16_mask = d.mask[l - 1];
                a.out_T = d.Tl[l - 1]; a.out_stride = 2; a.out_off = ph;
                RET_IF(run_gemm(c, a, s));
            }
            cur = up16;
        } else {
            GemmArgs a;
            panel_args(c, D.up_last, a); rows_plain(a, B, d.Tl[l]); taps_centered(a, 3);
            a.a16_0 = dst16; a.lda16_0 = d.ew * r.cout; a.c0 = r.cout;
            _Float16* o16 = (dst16 == d.A16[l]) ? d.B16[l] : d.A16[l];
            a.out16 = o16; a.ld16 = d.ew * r.cout; a.out16_mask = d.mask[l];
            RET_IF(run_gemm(c, a, s));
            cur = o16;
        }
        cur_c = r.cout;
    }
    // ---- final Block1D + 1x1 projection + mask (reference decoder.py:423-426)
    const int C0 = g.dec_channels[0], T = d.T;
    GemmArgs a;
    panel_args(c, D.final_conv, a); rows_plain(a, B, T); taps_centered(a, 3);
    a.a16_0 = cur; a.lda16_0 = d.ew * C0; a.c0 = C0; a.out = d.Y; a.ldc = C0;
    const int frf = gn_fuse_rows(a, C0, 8, T);
    if (frf) { a.gn_stats = d.gns; a.gn_groups = 8; a.gn_nrows = d.nr(0); }
    RET_IF(run_gemm(c, a, s));
    if (!frf) LAUNCH(c, 2, 0, s, launch_gn_partial(d.Y, B, T, C0, 8, d.gnp, s, d.nr(0)));
    GnApplyArgs ga;
    if (frf) { ga.tile_stats = d.gns; ga.tile_rows = frf; }
    ga.y = d.Y; ga.partial = d.gnp; ga.gamma = W(c, D.fgn_g.off); ga.beta = W(c, D.fgn_b.off); ga.mask = d.mask[0]; ga.nrows = d.nr(0);
    if (d.folded) { ga.nextra = d.ne(0); ga.bias_stats = W(c, D.fgn_bs.off); }
    ga.out16 = d.H16; ga.ld16 = d.ew * C0; ga.B = B; ga.T = T; ga.C = C0;
    RET_IF(run_gn_apply(c, ga, s));
    GemmArgs p;
    panel_args(c, D.final_proj, p); rows_plain(p, B, T);
    p.a16_0 = d.H16; p.lda16_0 = d.ew * C0; p.c0 = C0; p.out_mask = d.mask[0];
    p.out = fo.out; p.ldc = fo.ldc; p.res = fo.res; p.ldr = fo.ldr; p.out_scale = fo.scale;
    RET_IF(run_gemm(c, p, s));
    return 0;
}

// Level masks and frame tables of one call.  y_len == null: any float mask [B, T] (reference decoder.py:390 mask[:, :, ::2]),
// every utterance owns its T rows (or tlen[b] of them: per-request padding).  y_len != null: prefix masks of y_len[b] frames in
// the folded layout -- d.T rows per utterance stand for T_true reference frames (FrameTableArgs).
static int build_frames(mtts_ctx* c, DecBufs& d, const float* mask, const int64_t* y_len, int T_true, hipStream_t s) {
    d.T_true = T_true;
    d.folded = y_len != nullptr;
    d.tables = d.folded || c->d_tlen != nullptr;
    if (!d.folded)
        for (int l = 0; l < d.nl; ++l)
            LAUNCH(c, 2, 0, s, launch_mask_down(mask, d.B, d.T, 1 << l, d.mask[l], d.Tl[l], s));
    if (d.tables) {
        FrameTableArgs f;
        f.y_len = y_len; f.tlen = c->d_tlen; f.B = d.B; f.T_true = T_true; f.nl = d.nl;
        for (int l = 0; l < d.nl; ++l) {
            f.T[l] = d.Tl[l]; f.mask[l] = d.mask[l]; f.kbias[l] = d.kbias[l]; f.nrows[l] = d.nrows[l]; f.nextra[l] = d.nextra[l];
        }
        LAUNCH(c, 2, 0, s, launch_frame_tables(f, s));
    }
    return 0;
}

// Entry-point guard (round-2 verdict item 8 / advisor): a context is single-threaded by design; concurrent use is an error, not a race.
struct CtxGuard {
    mtts_ctx* c;
    bool ok;
    explicit CtxGuard(mtts_ctx* ctx) : c(ctx), ok(false) {
        if (!c) return;
        bool expect = false;
        ok = c->in_use.compare_exchange_strong(expect, true);
        if (!ok) set_error("this mtts_ctx is in use by another thread: a context is single-threaded (one context per worker / stream, include/mtts.h)");
    }
    ~CtxGuard() { if (ok) c->in_use.store(false); }
};
#define CTX_GUARD(ctx)            \
    CtxGuard _guard(ctx);         \
    if ((ctx) && !_guard.ok) return -1

static int check_ready(const mtts_ctx* c) {
    if (!c) { set_error("null context"); return -1; }
    if (!c->uploaded || !c->d_image) { set_error("weights not uploaded (mtts_upload_weights)"); return -1; }
    return 0;
}

}  // namespace mtts

using namespace mtts;

// ================================================================================================ C ABI
extern "C" {

int mtts_abi_version(void) { return MTTS_ABI_VERSION; }
const char* mtts_last_error(void) { return get_error(); }

mtts_ctx* mtts_create(const mtts_config* cfg) {
    if (!cfg) { set_error("null config"); return nullptr; }
    const mtts_config& g = *cfg;
    std::string why;
    if (g.dec_levels < 1 || g.dec_levels > 4) why = "dec_levels must be 1..4";
    else if (g.n_feats <= 0 || (2 * g.n_feats) % 4) why = "n_feats must be even";
    else if ((g.enc_channels + g.spk_emb_dim) % g.enc_heads) why = "encoder hidden size not divisible by heads";
    else if (((g.enc_channels + g.spk_emb_dim) / g.enc_heads) % 4 || (g.enc_channels + g.spk_emb_dim) / g.enc_heads > 64) why = "encoder head dim must be a multiple of 4, <= 64";
    else if (g.dec_head_dim % 4 || g.dec_head_dim > 64) why = "decoder head dim must be a multiple of 4, <= 64";
    else if (g.enc_channels % 4 || g.spk_emb_dim % 4 || g.enc_filter % 4 || g.dp_filter % 4) why = "channel counts must be multiples of 4";
    else if (g.enc_kernel > MAX_TAPS || g.prenet_kernel > MAX_TAPS || g.dp_kernel > MAX_TAPS) why = "kernel sizes above 5 unsupported";
    for (int i = 0; why.empty() && i < g.dec_levels; ++i)
        if (g.dec_channels[i] % 32) why = "decoder channels must be multiples of 32 (8 GroupNorm groups of 4k channels, K segments of 32)";
    if (!why.empty()) { set_error(why); return nullptr; }
    mtts_ctx* c = new mtts_ctx();
    c->cfg = g;
    c->gemm_terms = default_gemm_terms();
    { const char* e = getenv("MTTS_GEMM_TERMS"); c->fast16 = e && atoi(e) == 1; c->half16 = e && (atoi(e) == 16 || atoi(e) == 17); c->bf16 = e && atoi(e) == 17; }   // 1 / 16 / 17: 16-bit modes (include/mtts.h)
    { const char* e = getenv("MTTS_P16"); c->p16_on = !(e && e[0] == '0'); }
    { const char* e = getenv("MTTS_CHAIN"); c->chain_on = !(e && e[0] == '0'); }
    { const char* e = getenv("MTTS_CHAIN_CH"); c->chain_ch = (e && atoi(e) == 128) ? 128 : 256; }
    { const char* e = getenv("MTTS_CHAIN_QB"); c->chain_qb = e ? atoi(e) : 0; }
    { const char* e = getenv("MTTS_CHAIN_MIN_ROWS"); if (e) c->chain_min_rows = atoi(e); }
    { const char* e = getenv("MTTS_CHAIN_PAIR"); c->pair_on = !(e && e[0] == '0'); }
    if (c->pair_on) {          // the pair form's residency bound assumes the whole 256-CU chip: a partitioned or smaller device runs without it
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount < 256) c->pair_on = false;
        (void)hipGetLastError();             // (no device at all -- the CPU-side packing tests -- leaves the setting as it is)
    }
    return c;
}

int mtts_set_arithmetic(mtts_ctx* c, int terms) {
    if (!c) { set_error("null context"); return -1; }
    if (terms != 0 && terms != 1 && terms != 2 && terms != 3 && terms != 6 && terms != 16 && terms != 17) {
        set_error("mtts_set_arithmetic: terms must be 0, 1, 2, 3, 6, 16 or 17");
        return -1;
    }
    c->fast16 = terms == 1;
    c->half16 = terms == 16 || terms == 17;
    c->bf16 = terms == 17;
    c->gemm_terms = (terms == 1 || terms == 16 || terms == 17) ? 2 : terms;
    c->packed = false;
    c->uploaded = false;
    return 0;
}

int mtts_weights_saturate(mtts_ctx* c) {
    if (!c) { set_error("null context"); return -1; }
    if (!c->packed && pack_all(c)) return -1;
    return c->weights_saturate ? 1 : 0;
}

void mtts_destroy(mtts_ctx* c) {
    if (!c) return;
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    delete c;
}

int mtts_set_tensor(mtts_ctx* c, const char* key, const float* h, int64_t numel) {
    if (!c || !key || !h || numel < 0) { set_error("mtts_set_tensor: bad argument"); return -1; }
    c->raw[key].assign(h, h + numel);
    c->packed = false;
    c->uploaded = false;
    return 0;
}

int64_t mtts_weights_bytes(mtts_ctx* c) {
    if (!c) { set_error("null context"); return -1; }
    if (!c->packed && pack_all(c)) return -1;
    return (int64_t)(c->image.size() * sizeof(float));
}

// The packed image depends on the architecture, the arithmetic and the layout switches -- everything below, as one string: a
// cache file written by mtts_export_weights is valid for a context with the same signature and the same checkpoint tensors.
int mtts_weights_signature(mtts_ctx* c, char* buf, int64_t n) {
    if (!c || !buf || n < 64) { set_error("mtts_weights_signature: bad argument"); return -1; }
    const mtts_config& g = c->cfg;
    const int v[] = {MTTS_ABI_VERSION, MTTS_IMAGE_REVISION, g.n_feats, g.n_spks, g.spk_emb_dim, g.n_vocab, g.enc_channels, g.enc_filter,
                     g.enc_heads, g.enc_layers, g.enc_kernel, g.prenet_layers, g.prenet_kernel, g.dp_filter, g.dp_kernel, g.dp_layers,
                     g.dec_levels, g.dec_channels[0], g.dec_channels[1], g.dec_channels[2], g.dec_channels[3], g.dec_head_dim, g.dec_heads,
                     g.dec_n_blocks, g.dec_mid_blocks, c->gemm_terms, c->half16, c->bf16, c->fast16, c->p16_on, c->chain_on, c->chain_ch, c->pair_on};
    std::string sig = "mtts";
    for (int x : v) sig += "-" + std::to_string(x);
    if ((int64_t)sig.size() + 1 > n) { set_error("mtts_weights_signature: buffer too small"); return -1; }
    std::memcpy(buf, sig.c_str(), sig.size() + 1);
    return (int)sig.size();
}
int mtts_export_weights(mtts_ctx* c, void* h_dst, int64_t bytes, int* saturates) {
    if (!c || !h_dst) { set_error("mtts_export_weights: bad argument"); return -1; }
    if (!c->packed && pack_all(c)) return -1;
    if ((size_t)bytes < c->image.size() * sizeof(float)) { set_error("mtts_export_weights: buffer too small"); return -1; }
    std::memcpy(h_dst, c->image.data(), c->image.size() * sizeof(float));
    if (saturates) *saturates = c->weights_saturate ? 1 : 0;
    return 0;
}
// Adopt an image exported earlier by a context of the same signature over the same tensors: the layout pass runs (offsets, sizes,
// presence and shape of every registered tensor), the ~7 s of splitting and fragment packing do not.
int mtts_import_weights(mtts_ctx* c, const void* h_src, int64_t bytes, int saturates) {
    if (!c || !h_src) { set_error("mtts_import_weights: bad argument"); return -1; }
    if (pack_all(c, true)) return -1;
    if ((size_t)bytes != c->image.size() * sizeof(float)) {
        c->packed = false;
        set_error("mtts_import_weights: the image does not have this context's size (other architecture / arithmetic / library?)");
        return -1;
    }
    std::memcpy(c->image.data(), h_src, (size_t)bytes);
    c->weights_saturate = saturates != 0;
    c->uploaded = false;
    return 0;
}

int mtts_upload_weights(mtts_ctx* c, void* d_weights, int64_t bytes) {
    if (!c || !d_weights) { set_error("mtts_upload_weights: bad argument"); return -1; }
    if (!c->packed && pack_all(c)) return -1;
    if ((size_t)bytes < c->image.size() * sizeof(float)) { set_error("weight buffer too small"); return -1; }
    HIP_OK(hipMemcpy(d_weights, c->image.data(), c->image.size() * sizeof(float), hipMemcpyHostToDevice));
    c->d_image = static_cast<float*>(d_weights);
    c->uploaded = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------ decoder entry points
int64_t mtts_decoder_workspace_bytes(mtts_ctx* c, int B, int T) {
    if (!c || (!c->packed && pack_all(c))) return -1;
    WS ws(nullptr, 0);
    DecBufs d;
    if (plan_decoder(c, B, T, MAX_EVALS, 2, 4, ws, d)) return -1;
    return (int64_t)ws.off + 256;
}

// Test hook of the entry-point guard: holds the context as an entry point does, for `ms` milliseconds.
int mtts_debug_hold(mtts_ctx* c, int ms) {
    if (!c) { set_error("null context"); return -1; }
    CTX_GUARD(c);
    struct timespec ts = {ms / 1000, (long)(ms % 1000) * 1000000L};
    nanosleep(&ts, nullptr);
    return 0;
}

int mtts_set_frame_limits(mtts_ctx* c, const int32_t* d_t_len) {
    if (!c) { set_error("null context"); return -1; }
    c->d_tlen = d_t_len;
    return 0;
}

int mtts_decoder_forward(mtts_ctx* c, const float* d_x, const float* d_mask, const float* d_mu, float t, int B, int T,
                         float* d_out, void* d_ws, int64_t ws_bytes, void* stream) {
    CTX_GUARD(c);
    RET_IF(check_ready(c));
    hipStream_t s = static_cast<hipStream_t>(stream);
    WS ws(d_ws, (size_t)ws_bytes);
    DecBufs d;
    RET_IF(plan_decoder(c, B, T, MAX_EVALS, 2, 4, ws, d));
    if (ws.overflow) { set_error("decoder workspace too small"); return -1; }
    RET_IF(begin_call(c, d_ws, s));
    if (c->pair_on) HIP_OK(launch_fill_cols(reinterpret_cast<float*>(d.pair_flag), 1, 512, 0, 512, 0.f, s));
    const int nf = c->cfg.n_feats;
    RET_IF(build_frames(c, d, d_mask, nullptr, T, s));
    LAUNCH(c, 2, 0, s, launch_fill_cols(d.xmu, B * T, d.ldx, 2 * nf, d.ldx - 2 * nf, 0.f, s));
    LAUNCH(c, 2, 0, s, launch_cf_to_cl(d_x, nullptr, B, nf, T, d.xmu, d.ldx, 0, s));
    LAUNCH(c, 2, 0, s, launch_cf_to_cl(d_mu, nullptr, B, nf, T, d.xmu, d.ldx, nf, s));
    TimeVals tv;
    tv.t[0] = t;
    RET_IF(time_embed(c, d, tv, 1, s));
    FinalOut fo{d.vel[0], d.ldv, nullptr, 0, 1.0f};
    RET_IF(decoder_eval(c, d, d.xmu, 0, fo, s));
    LAUNCH(c, 2, 0, s, launch_cl_to_cf(d.vel[0], d.ldv, B, nf, T, d_out, T, 1.0f, 0.0f, s));
    return 0;
}

// BASECFM.solve (reference flow_matching.py:60-63) + torchdiffeq's fixed-grid loop.  The inputs are [B, n_feats, T_src]; the
// estimator holds T <= T_src rows per utterance (T < T_src: folded padding, y_len gives the prefix masks; else d_mask).
static int solve_core(mtts_ctx* c, const float* d_x0, const float* d_mu, const float* d_mask, const int64_t* d_y_len, int add_mu,
                      const float* h_t_span, int n_steps, int solver, int B, int T_src, int T, float* d_out, int T_out, float out_scale,
                      float out_shift, void* d_ws, int64_t ws_bytes, void* stream) {
    CTX_GUARD(c);
    RET_IF(check_ready(c));
    if (!h_t_span || n_steps < 1) { set_error("mtts_cfm_solve: bad time grid"); return -1; }
    const int stages = solver == MTTS_SOLVER_EULER ? 1 : solver == MTTS_SOLVER_MIDPOINT ? 2 : solver == MTTS_SOLVER_RK4 ? 4 : 0;
    if (!stages) { set_error("unsupported solver"); return -1; }
    if (n_steps * stages > MAX_EVALS) { set_error("too many function evaluations in one solve (max 256)"); return -1; }
    if (T_out > T) { set_error("mtts_cfm_solve: T_out exceeds the rows held per utterance"); return -1; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    WS ws(d_ws, (size_t)ws_bytes);
    DecBufs d;
    RET_IF(plan_decoder(c, B, T, MAX_EVALS, 2, 4, ws, d));
    if (ws.overflow) { set_error("decoder workspace too small"); return -1; }
    RET_IF(begin_call(c, d_ws, s));
    if (c->pair_on) HIP_OK(launch_fill_cols(reinterpret_cast<float*>(d.pair_flag), 1, 512, 0, 512, 0.f, s));
    const int nf = c->cfg.n_feats, M = B * T;
    RET_IF(build_frames(c, d, d_mask, d_y_len, T_src, s));
    // state rows: x | mu | zero pad.  z = mu + noise when use_mu_prior (reference flow_matching.py:52-55)
    float* states[2] = {d.xmu, d.xmu2};
    for (int k = 0; k < (stages > 1 ? 2 : 1); ++k) {
        LAUNCH(c, 2, 0, s, launch_fill_cols(states[k], M, d.ldx, 2 * nf, d.ldx - 2 * nf, 0.f, s));
        LAUNCH(c, 2, 0, s, launch_cf_to_cl(d_mu, nullptr, B, nf, T, states[k], d.ldx, nf, s, T_src));
    }
    LAUNCH(c, 2, 0, s, launch_cf_to_cl(d_x0, add_mu ? d_mu : nullptr, B, nf, T, d.xmu, d.ldx, 0, s, T_src));

    // evaluation times in torchdiffeq's fp32 arithmetic (fixed grid = t_span)
    TimeVals tv;
    int ne = 0;
    for (int i = 0; i < n_steps; ++i) {
        const float t0 = h_t_span[i], t1 = h_t_span[i + 1], dt = t1 - t0;
        if (solver == MTTS_SOLVER_EULER) tv.t[ne++] = t0;
        else if (solver == MTTS_SOLVER_MIDPOINT) { tv.t[ne++] = t0; tv.t[ne++] = t0 + 0.5f * dt; }
        else {
            const float third = 1.0f / 3.0f, two_thirds = 2.0f / 3.0f;
            tv.t[ne++] = t0; tv.t[ne++] = t0 + dt * third; tv.t[ne++] = t0 + dt * two_thirds; tv.t[ne++] = t1;
        }
    }
    RET_IF(time_embed(c, d, tv, ne, s));

    int ev = 0;
    for (int i = 0; i < n_steps; ++i) {
        const float dt = h_t_span[i + 1] - h_t_span[i];
        if (solver == MTTS_SOLVER_EULER) {             // y += dt * f(t0, y), fused into the last GEMM's epilogue
            FinalOut fo{d.xmu, d.ldx, d.xmu, d.ldx, dt};
            RET_IF(decoder_eval(c, d, d.xmu, ev++, fo, s));
        } else if (solver == MTTS_SOLVER_MIDPOINT) {   // y_mid = y + f(t0,y)*dt/2 ; y += dt * f(t0+dt/2, y_mid)
            FinalOut f1{d.xmu2, d.ldx, d.xmu, d.ldx, 0.5f * dt};
            RET_IF(decoder_eval(c, d, d.xmu, ev++, f1, s));
            FinalOut f2{d.xmu, d.ldx, d.xmu, d.ldx, dt};
            RET_IF(decoder_eval(c, d, d.xmu2, ev++, f2, s));
        } else {                                       // rk4, 3/8 rule
            for (int k = 0; k < 4; ++k) {
                FinalOut fk{d.vel[k], d.ldv, nullptr, 0, 1.0f};
                RET_IF(decoder_eval(c, d, k == 0 ? d.xmu : d.xmu2, ev++, fk, s));
                float* dst = k < 3 ? d.xmu2 : d.xmu;
                LAUNCH(c, 2, 0, s, launch_ode_combine(k + 1, dt, d.xmu, d.ldx, d.vel[0], d.vel[1], d.vel[2], d.vel[3], d.ldv, dst, d.ldx, M, nf, s));
            }
        }
    }
    LAUNCH(c, 2, 0, s, launch_cl_to_cf(d.xmu, d.ldx, B, nf, T, d_out, T_out, out_scale, out_shift, s));
    return 0;
}

int mtts_cfm_solve(mtts_ctx* c, const float* d_x0, const float* d_mu, const float* d_mask, int add_mu, const float* h_t_span,
                   int n_steps, int solver, int B, int T, float* d_out, int T_out, float out_scale, float out_shift, void* d_ws,
                   int64_t ws_bytes, void* stream) {
    if (!d_mask) { set_error("mtts_cfm_solve: null mask"); return -1; }
    return solve_core(c, d_x0, d_mu, d_mask, nullptr, add_mu, h_t_span, n_steps, solver, B, T, T, d_out, T_out, out_scale, out_shift,
                      d_ws, ws_bytes, stream);
}

int mtts_fold_rows(mtts_ctx* c, int y_max, int align) {
    if (!c || y_max < 1 || align < 1) { set_error("mtts_fold_rows: bad argument"); return -1; }
    const int f = 1 << (c->cfg.dec_levels - 1);
    return round_up((y_max + f - 1) / f + 1, align) * f;
}

int mtts_cfm_solve_folded(mtts_ctx* c, const float* d_x0, const float* d_mu, const int64_t* d_y_lengths, int y_max, int add_mu,
                          const float* h_t_span, int n_steps, int solver, int B, int T, int T_fold, float* d_out, int T_out,
                          float out_scale, float out_shift, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!c || !d_y_lengths) { set_error("mtts_cfm_solve_folded: bad argument"); return -1; }
    if (T_fold > T || T_fold < mtts_fold_rows(c, y_max, 1)) {
        set_error("mtts_cfm_solve_folded: T_fold must hold ceil(y_max / 2^l) + 1 rows at every level and not exceed T (mtts_fold_rows)");
        return -1;
    }
    if (y_max >= T) { set_error("mtts_cfm_solve_folded: no padded frame to fold (y_max >= T)"); return -1; }
    if (T_fold % (1 << (c->cfg.dec_levels - 1))) {      // (plan_decoder halves the row count per level: a remainder would truncate)
        set_error("mtts_cfm_solve_folded: T_fold must be a multiple of 2^(levels-1) (mtts_fold_rows returns such counts)");
        return -1;
    }
    return solve_core(c, d_x0, d_mu, nullptr, d_y_lengths, add_mu, h_t_span, n_steps, solver, B, T, T_fold, d_out, T_out, out_scale,
                      out_shift, d_ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ text encoder
struct EncBufs {
    float *X0, *P1, *P2, *Y, *H, *H2, *QKV, *ATT, *F1, *PM, *MU, *FILM, *D1, *D2;
};
static void plan_encoder(const mtts_ctx* c, int B, int Tx, WS& ws, EncBufs& e) {
    const mtts_config& g = c->cfg;
    const size_t M = (size_t)B * Tx;
    const int nch = g.enc_channels, Hd = nch + g.spk_emb_dim, F = g.dp_filter;
    (void)ws.bytes(256);                 // header: the call's range flag (begin_call)
    e.X0 = ws.f(M * nch); e.P1 = ws.f(M * nch); e.P2 = ws.f(M * nch); e.Y = ws.f(M * std::max(nch, F));
    e.H = ws.f(M * Hd); e.H2 = ws.f(M * Hd); e.QKV = ws.f(M * 3 * Hd); e.ATT = ws.f(M * Hd);
    e.F1 = ws.f(M * g.enc_filter); e.PM = ws.f(M * nch); e.MU = ws.f(M * round_up(g.n_feats, 4));
    e.FILM = ws.f((size_t)B * 2 * F); e.D1 = ws.f(M * F); e.D2 = ws.f(M * F);
}

int64_t mtts_encoder_workspace_bytes(mtts_ctx* c, int B, int Tx) {
    if (!c || (!c->packed && pack_all(c))) return -1;
    WS ws(nullptr, 0);
    EncBufs e;
    plan_encoder(c, B, Tx, ws, e);
    return (int64_t)ws.off + 256;
}

// TextEncoder.forward (reference text_encoder.py:375-406)
int mtts_text_encoder_forward(mtts_ctx* c, const int64_t* d_x, const int64_t* d_x_lengths, const float* d_e_enc, const float* d_e_dur,
                              int B, int Tx, float* d_mu_x, float* d_logw, float* d_x_mask, void* d_ws, int64_t ws_bytes, void* stream) {
    CTX_GUARD(c);
    RET_IF(check_ready(c));
    const mtts_config& g = c->cfg;
    const EncW& E = c->enc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nch = g.enc_channels, Sd = g.spk_emb_dim, Hd = nch + Sd, F = g.dp_filter, M = B * Tx;
    const int dh = Hd / g.enc_heads, d_rope = dh / 2;
    if ((size_t)Tx * d_rope > (size_t)E.rope_cos.n) { set_error("Phonetic representation too long, exceeds RoPE cache size"); return -1; }
    WS ws(d_ws, (size_t)ws_bytes);
    EncBufs e;
    plan_encoder(c, B, Tx, ws, e);
    if (ws.overflow) { set_error("encoder workspace too small"); return -1; }
    RET_IF(begin_call(c, d_ws, s));
    float* xm = d_x_mask;   // [B,1,Tx] == rows [B*Tx]
    LAUNCH(c, 2, 0, s, launch_seq_mask(d_x_lengths, B, Tx, xm, s));
    LAUNCH(c, 2, 0, s, launch_embedding(d_x, W(c, E.emb.off), M, nch, sqrtf((float)nch), xm, e.X0, nch, s));
    // ---- prenet: ConvSiluNorm (reference text_encoder.py:55-62)
    const float* cur = e.X0;
    for (int i = 0; i < g.prenet_layers; ++i) {
        GemmArgs a;
        panel_args(c, E.pre_conv[i], a); rows_plain(a, B, Tx); taps_centered(a, g.prenet_kernel);
        a.a0 = cur; a.lda0 = nch; a.c0 = nch; a.a_mask = xm; a.out = e.Y; a.ldc = nch;
        RET_IF(run_gemm(c, a, s));
        float* dst = (i & 1) ? e.P2 : e.P1;
        LayerNormArgs ln;
        ln.x = e.Y; ln.ldx = nch; ln.y = dst; ln.ldy = nch; ln.M = M; ln.C = nch; ln.T = Tx;
        ln.gamma = W(c, E.pre_g[i].off); ln.beta = W(c, E.pre_b[i].off); ln.act = ACT_SILU;
        LAUNCH(c, 2, 0, s, launch_layernorm(ln, s));
        cur = dst;
    }
    {
        GemmArgs a;   // (x_org + proj(x)) * mask, written into the first n_channels columns of the hidden rows
        panel_args(c, E.pre_proj, a); rows_plain(a, B, Tx);
        a.a0 = cur; a.lda0 = nch; a.c0 = nch; a.out_mask = xm; a.res = e.X0; a.ldr = nch; a.out = e.H; a.ldc = Hd;
        RET_IF(run_gemm(c, a, s));
    }
    LAUNCH(c, 2, 0, s, launch_bcast_rows(d_e_enc, B, Tx, Sd, xm, e.H, Hd, nch, s));
    // ---- Encoder: post-LN transformer with RoPE attention and conv FFN (reference text_encoder.py:299-316)
    for (int l = 0; l < g.enc_layers; ++l) {
        GemmArgs q;
        panel_args(c, E.qkv[l], q); rows_plain(q, B, Tx);
        q.a0 = e.H; q.lda0 = Hd; q.c0 = Hd; q.out = e.QKV; q.ldc = 3 * Hd;
        RET_IF(run_gemm(c, q, s));
        LAUNCH(c, 2, 0, s, launch_rope(e.QKV, B, Tx, g.enc_heads, dh, d_rope, W(c, E.rope_cos.off), W(c, E.rope_sin.off), s));
        AttnArgs at;
        at.qkv = e.QKV; at.mask = xm; at.out = e.ATT; at.B = B; at.T = Tx; at.H = g.enc_heads; at.D = dh;
        at.scale = 1.0f / sqrtf((float)dh); at.mask_mode = 1;
        RET_IF(run_attn(c, at, s));
        GemmArgs o;
        panel_args(c, E.o[l], o); rows_plain(o, B, Tx);
        o.a0 = e.ATT; o.lda0 = Hd; o.c0 = Hd; o.res = e.H; o.ldr = Hd; o.out = e.H2; o.ldc = Hd;
        RET_IF(run_gemm(c, o, s));
        LayerNormArgs n1;
        n1.x = e.H2; n1.ldx = Hd; n1.y = e.H; n1.ldy = Hd; n1.M = M; n1.C = Hd; n1.T = Tx;
        n1.gamma = W(c, E.n1_g[l].off); n1.beta = W(c, E.n1_b[l].off); n1.mask = xm;
        LAUNCH(c, 2, 0, s, launch_layernorm(n1, s));
        // FFN: conv k5 -> ReLU -> mask -> conv k5.  The second conv is the encoder's long-K GEMM (K = 5 x filter) on a grid
        // far under one round of workgroups: in the fp16-split mode the hidden layer is handed over as a masked P16 image
        // (written by the first conv's epilogue) so that it runs on gemm_p16.hip's prefetch ring (198 -> ~80 us at B = 32).
        const bool ffn_p16 = c->p16_on && c->gemm_terms == 2 && (g.enc_filter % 32) == 0;
        _Float16* F16 = reinterpret_cast<_Float16*>(e.F1);        // same bytes as the fp32 hidden layer
        GemmArgs f1;
        panel_args(c, E.ffn1[l], f1); rows_plain(f1, B, Tx); taps_centered(f1, g.enc_kernel);
        f1.a0 = e.H; f1.lda0 = Hd; f1.c0 = Hd; f1.act = ACT_RELU;
        if (ffn_p16) { f1.out16 = F16; f1.ld16 = 2 * g.enc_filter; f1.out16_mask = xm; }
        else { f1.out = e.F1; f1.ldc = g.enc_filter; }
        RET_IF(run_gemm(c, f1, s));
        GemmArgs f2;
        panel_args(c, E.ffn2[l], f2); rows_plain(f2, B, Tx); taps_centered(f2, g.enc_kernel);
        if (ffn_p16) { f2.a16_0 = F16; f2.lda16_0 = 2 * g.enc_filter; f2.c0 = g.enc_filter; f2.fast16 = false; }   // durations: full precision
        else { f2.a0 = e.F1; f2.lda0 = g.enc_filter; f2.c0 = g.enc_filter; f2.a_mask = xm; }
        f2.out_mask = xm; f2.res = e.H; f2.ldr = Hd; f2.out = e.H2; f2.ldc = Hd;
        RET_IF(run_gemm(c, f2, s));
        LayerNormArgs n2 = n1;
        n2.gamma = W(c, E.n2_g[l].off); n2.beta = W(c, E.n2_b[l].off);
        LAUNCH(c, 2, 0, s, launch_layernorm(n2, s));
    }
    // ---- proj_m: 1x1 -> SiLU -> 1x1, masked (reference text_encoder.py:359-363,402)
    {
        GemmArgs a;
        panel_args(c, E.pm0, a); rows_plain(a, B, Tx);
        a.a0 = e.H; a.lda0 = Hd; a.c0 = Hd; a.act = ACT_SILU; a.out = e.PM; a.ldc = nch;
        RET_IF(run_gemm(c, a, s));
        GemmArgs b;
        const int ldm = round_up(g.n_feats, 4);
        panel_args(c, E.pm2, b); rows_plain(b, B, Tx);
        b.a0 = e.PM; b.lda0 = nch; b.c0 = nch; b.out_mask = xm; b.out = e.MU; b.ldc = ldm;
        RET_IF(run_gemm(c, b, s));
        LAUNCH(c, 2, 0, s, launch_cl_to_cf(e.MU, ldm, B, g.n_feats, Tx, d_mu_x, Tx, 1.0f, 0.0f, s));
    }
    // ---- DurationPredictor with FiLM (reference text_encoder.py:101-112)
    {
        GemmArgs fm;
        panel_args(c, E.film, fm); rows_plain(fm, B, 1);
        fm.a0 = d_e_dur; fm.lda0 = Sd; fm.c0 = Sd; fm.out = e.FILM; fm.ldc = 2 * F;
        RET_IF(run_gemm(c, fm, s));
        const float* dcur = e.H;
        int dc = Hd;
        for (int i = 0; i < g.dp_layers; ++i) {
            GemmArgs a;
            panel_args(c, E.dp_conv[i], a); rows_plain(a, B, Tx); taps_centered(a, g.dp_kernel);
            a.a0 = dcur; a.lda0 = dc; a.c0 = dc; a.a_mask = xm; a.act = ACT_RELU; a.out = e.Y; a.ldc = F;
            RET_IF(run_gemm(c, a, s));
            float* dst = (i & 1) ? e.D2 : e.D1;
            LayerNormArgs ln;
            ln.x = e.Y; ln.ldx = F; ln.y = dst; ln.ldy = F; ln.M = M; ln.C = F; ln.T = Tx;
            ln.gamma = W(c, E.dp_g[i].off); ln.beta = W(c, E.dp_b[i].off); ln.film = e.FILM;
            LAUNCH(c, 2, 0, s, launch_layernorm(ln, s));
            dcur = dst;
            dc = F;
        }
        GemmArgs p;
        panel_args(c, E.dp_proj, p); rows_plain(p, B, Tx);
        p.a0 = dcur; p.lda0 = dc; p.c0 = dc; p.a_mask = xm; p.out_mask = xm; p.out = d_logw; p.ldc = 1;
        RET_IF(run_gemm(c, p, s));
    }
    return 0;
}

int mtts_speaker_embedding(mtts_ctx* c, int table, const int64_t* d_ids, int B, float* d_out, void* stream) {
    RET_IF(check_ready(c));
    const Vec& v = table == 0 ? c->enc.spk_enc : c->enc.spk_dur;
    HIP_OK(launch_embedding(d_ids, W(c, v.off), B, c->cfg.spk_emb_dim, 1.0f, nullptr, d_out, c->cfg.spk_emb_dim, static_cast<hipStream_t>(stream)));
    return 0;
}

int mtts_durations(const float* d_logw, const float* d_x_mask, float scale_correction, float length_scale, int B, int Tx,
                   float* d_durations, int32_t* d_cum, int64_t* d_y_fine_lengths, void* stream) {
    HIP_OK(launch_durations(d_logw, d_x_mask, scale_correction, length_scale, B, Tx, d_durations, d_cum, d_y_fine_lengths,
                            static_cast<hipStream_t>(stream)));
    return 0;
}

int mtts_durations_per_utterance(const float* d_logw, const float* d_x_mask, const float* d_scale_correction, const float* d_length_scale,
                                 int B, int Tx, float* d_durations, int32_t* d_cum, int64_t* d_y_fine_lengths, void* stream) {
    if (!d_scale_correction || !d_length_scale) { set_error("mtts_durations_per_utterance: null factor array"); return -1; }
    HIP_OK(launch_durations(d_logw, d_x_mask, 1.0f, 1.0f, B, Tx, d_durations, d_cum, d_y_fine_lengths, static_cast<hipStream_t>(stream),
                            d_scale_correction, d_length_scale));
    return 0;
}

int mtts_align_pool(const float* d_mu_x, const int32_t* d_cum, const int64_t* d_y_fine_lengths, int B, int n_feats, int Tx, int T_pad,
                    float* d_mu_y, float* d_y_mask, int64_t* d_y_lengths, void* stream) {
    HIP_OK(launch_align_pool(d_mu_x, d_cum, d_y_fine_lengths, B, n_feats, Tx, T_pad, d_mu_y, d_y_mask, d_y_lengths,
                             static_cast<hipStream_t>(stream)));
    return 0;
}

// ------------------------------------------------------------------------------------------------ single kernels
int64_t mtts_gemm_packed_bytes(int N, int C, int ntaps) {   // fp32 panel + three bf16 planes
    const int64_t n = (int64_t)round_up(N, GEMM_BN) * ntaps * round_up(C, GEMM_BK);
    return n * 4 + ((3 * n + 1) / 2) * 4 + 256;
}

int mtts_gemm_f32(const float* d_a, int lda, int B, int T_in, int C, int ntaps, const int* h_tap_off, int in_stride, int T_out,
                  const float* d_a_mask, const float* d_a_mean, const float* d_a_rstd, const float* d_a_part, int a_nparts,
                  const float* d_w, void* d_wpacked, const float* d_bias, int N, int act, const float* d_p0, const float* d_p1,
                  const float* d_res, int ldr, const float* d_out_mask, float out_scale, float* d_out, int ldc, float* d_stats_out,
                  int terms, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ntaps < 1 || ntaps > MAX_TAPS) { set_error("ntaps out of range"); return -1; }
    if (terms < 0) terms = default_gemm_terms();
    if (terms != 0 && terms != 2 && terms != 3 && terms != 6) { set_error("terms must be 0, 2, 3 or 6"); return -1; }
    const size_t npanel = (size_t)round_up(N, GEMM_BN) * ntaps * round_up(C, GEMM_BK);
    float* planes = static_cast<float*>(d_wpacked) + ((npanel + 63) & ~size_t(63));
    if (d_w) {   // NULL: d_wpacked already packed by an earlier call
        HIP_OK(launch_pack_weight(d_w, N, C, ntaps, static_cast<float*>(d_wpacked), s));
        if (terms == 2) HIP_OK(launch_split_panel_f16(static_cast<const float*>(d_wpacked), npanel, planes, s));
        else HIP_OK(launch_split_panel(static_cast<const float*>(d_wpacked), npanel, planes, s));
    }
    GemmArgs a;
    a.a0 = d_a; a.lda0 = lda; a.c0 = C; a.ktap = round_up(C, GEMM_BK); a.ntaps = ntaps;
    for (int j = 0; j < ntaps; ++j) a.tap_off[j] = h_tap_off ? h_tap_off[j] : 0;
    a.in_stride = in_stride; a.B = B; a.T_in = T_in; a.T_out = T_out;
    a.a_mask = d_a_mask; a.a_mean = d_a_mean; a.a_rstd = d_a_rstd; a.a_part = d_a_part; a.a_nparts = a_nparts;
    a.stats_out = d_stats_out;
    a.w = static_cast<const float*>(d_wpacked); a.w16 = planes; a.terms = terms; a.bias = d_bias; a.N = N; a.act = act; a.p0 = d_p0; a.p1 = d_p1;
    a.res = d_res; a.ldr = ldr; a.out_mask = d_out_mask; a.out_scale = out_scale; a.out = d_out; a.ldc = ldc;
    a.out_T = T_out; a.out_stride = 1; a.out_off = 0;
    HIP_OK(launch_gemm(a, s));
    return 0;
}

// Test entry for the P16 GEMM (gemm_p16.hip): the fp32 operand is converted to its P16 image (optionally masked) in
// d_scratch, the panel is packed as for mtts_gemm_f32 (terms = 2) and its row sums are computed for the LayerNorm algebra;
// the optional P16 output is decoded back to fp32 into d_out16_f32.
int64_t mtts_gemm_p16_scratch_bytes(int B, int T_in, int C, int T_out, int N) {
    return (int64_t)B * T_in * C * 4 + (int64_t)B * T_out * round_up(N, 32) * 4 + (int64_t)round_up(N, GEMM_BN) * 4 + 1024;
}
__global__ void panel_rowsum_kernel(const float* __restrict__ panel, int Np, int Kp, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Np) return;
    double acc = 0.0;
    for (int k = 0; k < Kp; ++k) acc += (double)panel[(size_t)n * Kp + k];
    out[n] = (float)acc;
}
int mtts_gemm_p16(const float* d_a, int lda, int B, int T_in, int C, int ntaps, const int* h_tap_off, int in_stride, int T_out,
                  const float* d_a_mask, const float* d_a_mean, const float* d_a_rstd, const float* d_a_part, int a_nparts,
                  const float* d_w, void* d_wpacked, const float* d_bias, int N, int act, const float* d_p0, const float* d_p1,
                  const float* d_res, int ldr, const float* d_out_mask, float out_scale, float* d_out, int ldc,
                  float* d_out16_f32, float out_lscale, float* d_stats_out, int force_bm, void* d_scratch, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ntaps < 1 || ntaps > MAX_TAPS) { set_error("ntaps out of range"); return -1; }
    if (C % GEMM_BK) { set_error("P16 operands need C % 32 == 0"); return -1; }
    if (!d_w || !d_wpacked || !d_scratch) { set_error("null buffer"); return -1; }
    const int Np = round_up(N, GEMM_BN), Kp = ntaps * C;
    const size_t npanel = (size_t)Np * Kp;
    float* planes = static_cast<float*>(d_wpacked) + ((npanel + 63) & ~size_t(63));
    HIP_OK(launch_pack_weight(d_w, N, C, ntaps, static_cast<float*>(d_wpacked), s));
    HIP_OK(launch_split_panel_f16(static_cast<const float*>(d_wpacked), npanel, planes, s));
    char* sc = static_cast<char*>(d_scratch);
    _Float16* a16 = reinterpret_cast<_Float16*>(sc);
    sc += (size_t)B * T_in * C * 4;
    _Float16* o16 = reinterpret_cast<_Float16*>(sc);
    sc += (size_t)B * T_out * round_up(N, 32) * 4;
    float* wsum = reinterpret_cast<float*>(sc);
    HIP_OK(launch_to_p16(d_a, lda, d_a_mask, B * T_in, C, C, a16, 2 * C, 2048.0f, s));
    hipLaunchKernelGGL(panel_rowsum_kernel, dim3((Np + 127) / 128), dim3(128), 0, s, static_cast<const float*>(d_wpacked), Np, Kp, wsum);
    HIP_OK(hipGetLastError());
    GemmArgs a;
    a.a16_0 = a16; a.lda16_0 = 2 * C; a.c0 = C; a.ktap = C; a.ntaps = ntaps;
    for (int j = 0; j < ntaps; ++j) a.tap_off[j] = h_tap_off ? h_tap_off[j] : 0;
    a.in_stride = in_stride; a.B = B; a.T_in = T_in; a.T_out = T_out;
    a.a_mean = d_a_mean; a.a_rstd = d_a_rstd; a.a_part = d_a_part; a.a_nparts = a_nparts; a.wsum = wsum;
    a.stats_out = d_stats_out;
    a.w16 = planes; a.terms = 2; a.bias = d_bias; a.N = N; a.act = act; a.p0 = d_p0; a.p1 = d_p1;
    a.res = d_res; a.ldr = ldr; a.out_mask = d_out_mask; a.out_scale = out_scale; a.out = d_out; a.ldc = ldc;
    if (d_out16_f32) { a.out16 = o16; a.ld16 = 2 * N; a.out_lscale = out_lscale; }
    a.out_T = T_out; a.out_stride = 1; a.out_off = 0; a.force_bm = force_bm;
    HIP_OK(launch_gemm(a, s));
    if (d_out16_f32) HIP_OK(launch_from_p16(o16, 2 * N, B * T_out, N, out_lscale, d_out16_f32, N, s));
    return 0;
}

int mtts_attention_f32(const float* d_qkv, const float* d_mask, int B, int T, int H, int D, float scale, int mask_mode, float* d_out,
                       void* stream) {
    AttnArgs a;
    a.qkv = d_qkv; a.mask = d_mask; a.out = d_out; a.B = B; a.T = T; a.H = H; a.D = D; a.scale = scale; a.mask_mode = mask_mode;
    HIP_OK(launch_attention(a, static_cast<hipStream_t>(stream)));
    return 0;
}

// Test entry for the attention kernel's P16 I/O: q|k|v converted to a P16 image with unscaled residuals in d_scratch
// (>= 16*B*T*H*64 bytes), the P16 output decoded back to fp32.  D must be 64.
int mtts_attention_p16(const float* d_qkv, const float* d_mask, int B, int T, int H, int D, float scale, int mask_mode, float* d_out,
                       void* d_scratch, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (D != 64 || !d_scratch) { set_error("P16 attention needs D == 64 and a scratch buffer"); return -1; }
    const int M = B * T, C3 = 3 * H * D;
    _Float16* q16 = static_cast<_Float16*>(d_scratch);
    _Float16* o16 = q16 + (size_t)M * 2 * C3;
    HIP_OK(launch_to_p16(d_qkv, C3, nullptr, M, C3, C3, q16, 2 * C3, 1.0f, s));
    AttnArgs a;
    a.qkv16 = q16; a.ld16 = 2 * C3; a.out16 = o16; a.ldo16 = 2 * H * D; a.mask = d_mask;
    a.B = B; a.T = T; a.H = H; a.D = D; a.scale = scale; a.mask_mode = mask_mode;
    HIP_OK(launch_attention(a, s));
    HIP_OK(launch_from_p16(o16, 2 * H * D, M, H * D, a.out_lscale, d_out, H * D, s));
    return 0;
}

int mtts_row_stats(const float* d_x, int M, int C, int ld, float eps, float* d_mean, float* d_rstd, void* stream) {
    HIP_OK(launch_row_stats(d_x, M, C, ld, eps, d_mean, d_rstd, static_cast<hipStream_t>(stream)));
    return 0;
}

int mtts_channel_layernorm(const float* d_x, int B, int T, int C, const float* d_gamma, const float* d_beta, float eps, int act,
                           const float* d_film, const float* d_mask, float* d_y, void* stream) {
    if (B <= 0 || T <= 0) { set_error("mtts_channel_layernorm: empty batch"); return -1; }
    if (act != ACT_NONE && act != ACT_SILU) { set_error("mtts_channel_layernorm: act must be 0 (none) or 2 (SiLU)"); return -1; }
    LayerNormArgs a;
    a.x = d_x; a.ldx = C; a.y = d_y; a.ldy = C; a.M = B * T; a.C = C; a.T = T; a.gamma = d_gamma; a.beta = d_beta; a.eps = eps;
    a.act = act; a.film = d_film; a.mask = d_mask;
    HIP_OK(launch_layernorm(a, static_cast<hipStream_t>(stream)));
    return 0;
}

int64_t mtts_groupnorm_scratch_bytes(int B, int T, int G) { return (int64_t)B * gn_chunks_max(T) * G * 2 * (int64_t)sizeof(float); }

int mtts_groupnorm_mish(const float* d_y, const float* d_gamma, const float* d_beta, const float* d_mask, int B, int T, int C, int G,
                        float eps, float* d_out, void* d_scratch, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_OK(launch_gn_partial(d_y, B, T, C, G, static_cast<float*>(d_scratch), s));
    GnApplyArgs a;
    a.y = d_y; a.partial = static_cast<const float*>(d_scratch); a.gamma = d_gamma; a.beta = d_beta; a.mask = d_mask;
    a.out = d_out; a.B = B; a.T = T; a.C = C; a.G = G; a.eps = eps;
    HIP_OK(launch_gn_apply(a, s));
    return 0;
}

// Test entry for the transformer-block chain (tblock_chain.hip).  fp32 operands are converted to P16 images in d_scratch, the fp32
// panels (LayerNorm affines already folded: w1 / b1 for the FeedForward, w_qkv / b_qkv for the following block) are packed into a
// fragment stream on the host, and the P16 outputs are decoded back to fp32.  w_qkv == NULL: no q|k|v phase; inner == 0: no
// out-projection (the FeedForward alone on d_x).  h_* pointers are HOST memory, d_* device memory.
// the model's launch plan for M rows (test entry; no GPU): rows per workgroup and prefetch workgroups
int mtts_chain_plan(int M, int ch, int* qb, int* prefetch_wgs) {
    if (M <= 0 || (ch != 128 && ch != 256) || !qb || !prefetch_wgs) { set_error("mtts_chain_plan: bad argument"); return -1; }
    chain_plan(M, ch, 0, qb, prefetch_wgs);
    return 0;
}
int64_t mtts_chain_stream_frags(int C, int inner, int ch, int n_qkv) {
    if (!chain_supported(C, inner, n_qkv) || (ch != 128 && ch != 256)) { set_error("mtts_chain_stream_frags: unsupported shape"); return -1; }
    return chain_stream_frags(C, inner, ch, n_qkv);
}
int mtts_chain_stream_pack(int C, int inner, int ch, int n_qkv, const float* h_w_out, const float* h_w1, const float* h_w2,
                           const float* h_w_qkv, uint16_t* h_dst) {
    if (!chain_supported(C, inner, n_qkv) || (ch != 128 && ch != 256) || !h_w1 || !h_w2 || !h_dst || (inner && !h_w_out) || (n_qkv && !h_w_qkv)) {
        set_error("mtts_chain_stream_pack: unsupported shape or null panel");
        return -1;
    }
    chain_stream_pack(C, inner, ch, n_qkv, h_w_out, h_w1, h_w2, h_w_qkv, h_dst, nullptr);
    return 0;
}
// pair form: fragments per (half, wave), and the packing of the 2 x 8 streams (host only)
int64_t mtts_chain_stream_frags_pair(int C, int inner, int ch, int n_qkv) {
    if (!chain_supported_pair(C, inner, ch, n_qkv) || (ch != 128 && ch != 256)) { set_error("mtts_chain_stream_frags_pair: unsupported shape"); return -1; }
    return chain_stream_frags_pair(C, inner, ch, n_qkv);
}
int mtts_chain_stream_pack_pair(int C, int inner, int ch, int n_qkv, const float* h_w_out, const float* h_w1, const float* h_w2,
                                const float* h_w_qkv, uint16_t* h_dst) {
    if (!chain_supported_pair(C, inner, ch, n_qkv) || (ch != 128 && ch != 256) || !h_w_out || !h_w1 || !h_w2 || !h_dst || (n_qkv && !h_w_qkv)) {
        set_error("mtts_chain_stream_pack_pair: unsupported shape or null panel");
        return -1;
    }
    chain_stream_pack_pair(C, inner, ch, n_qkv, h_w_out, h_w1, h_w2, h_w_qkv, h_dst, nullptr);
    return 0;
}
int64_t mtts_tblock_chain_scratch_bytes(int M, int C, int inner, int n_qkv, int ch) {
    if (!chain_supported(C, inner, n_qkv)) return -1;
    int64_t stream = (int64_t)chain_stream_frags(C, inner, ch, n_qkv) * CHAIN_WAVES * 1024;
    if (chain_supported_pair(C, inner, ch, n_qkv)) stream = std::max<int64_t>(stream, (int64_t)chain_stream_frags_pair(C, inner, ch, n_qkv) * 2 * CHAIN_WAVES * 1024);
    const int64_t pair_scratch = 2 * ((int64_t)M + 64) * C * 4 + 2 * ((int64_t)M / 32 + 2) * 4 + 512;       // partial sums + flags of the pair form
    return stream + (int64_t)M * 4 * (inner + 2 * C + n_qkv) + 4 * (int64_t)(2 * n_qkv + 18 * C) + 4096 + pair_scratch;
}
int mtts_tblock_chain(const float* d_att, const float* d_x, int M, int C, int inner, const float* h_w_out, const float* h_b_out,
                      const float* h_w1, const float* h_b1, const float* h_p0, const float* h_p1, const float* h_w2, const float* h_b2,
                      const float* h_w_qkv, const float* h_b_qkv, int n_qkv, const float* d_out_mask, int qb, int ch, float* d_x_out,
                      float* d_qkv_out, void* d_scratch, void* stream) {
    return mtts_tblock_chain_timed(d_att, d_x, M, C, inner, h_w_out, h_b_out, h_w1, h_b1, h_p0, h_p1, h_w2, h_b2, h_w_qkv, h_b_qkv, n_qkv,
                                   d_out_mask, qb, ch, d_x_out, d_qkv_out, d_scratch, stream, 0, nullptr);
}
static int tblock_chain_entry(bool pair, const float* d_att, const float* d_x, int M, int C, int inner, const float* h_w_out, const float* h_b_out,
                              const float* h_w1, const float* h_b1, const float* h_p0, const float* h_p1, const float* h_w2, const float* h_b2,
                              const float* h_w_qkv, const float* h_b_qkv, int n_qkv, const float* d_out_mask, int qb, int ch, float* d_x_out,
                              float* d_qkv_out, void* d_scratch, void* stream, int repeat, float* h_ms) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!h_w_qkv) n_qkv = 0;
    if (!chain_supported(C, inner, n_qkv) || !d_x || !h_w1 || !h_w2 || !d_scratch || !d_x_out) { set_error("mtts_tblock_chain: unsupported shape or null buffer"); return -1; }
    if (pair && !chain_supported_pair(C, inner, ch, n_qkv)) { set_error("mtts_tblock_chain_pair: unsupported shape"); return -1; }
    const long frags = pair ? chain_stream_frags_pair(C, inner, ch, n_qkv) : chain_stream_frags(C, inner, ch, n_qkv);
    std::vector<uint16_t> hs((size_t)frags * (pair ? 2 : 1) * CHAIN_WAVES * 512);
    if (pair) chain_stream_pack_pair(C, inner, ch, n_qkv, h_w_out, h_w1, h_w2, h_w_qkv, hs.data(), nullptr);
    else chain_stream_pack(C, inner, ch, n_qkv, h_w_out, h_w1, h_w2, h_w_qkv, hs.data(), nullptr);
    std::vector<float> hc((size_t)18 * C + 2 * (size_t)n_qkv, 0.f);         // wsum1 | b1 | p0 | p1 | b_out | b2 | wsum_qkv | b_qkv
    for (int n = 0; n < 4 * C; ++n) {
        double a = 0.0;
        for (int k = 0; k < C; ++k) a += (double)h_w1[(size_t)n * C + k];
        hc[n] = (float)a;
        hc[4 * C + n] = h_b1 ? h_b1[n] : 0.f;
        hc[8 * C + n] = h_p0[n];
        hc[12 * C + n] = h_p1[n];
    }
    for (int n = 0; n < C; ++n) { hc[16 * C + n] = (inner && h_b_out) ? h_b_out[n] : 0.f; hc[17 * C + n] = h_b2 ? h_b2[n] : 0.f; }
    for (int n = 0; n < n_qkv; ++n) {
        double a = 0.0;
        for (int k = 0; k < C; ++k) a += (double)h_w_qkv[(size_t)n * C + k];
        hc[18 * C + n] = (float)a;
        hc[18 * C + n_qkv + n] = h_b_qkv ? h_b_qkv[n] : 0.f;
    }
    char* sc = static_cast<char*>(d_scratch);
    _Float16* d_stream = reinterpret_cast<_Float16*>(sc); sc += hs.size() * 2;
    float* d_c = reinterpret_cast<float*>(sc); sc += hc.size() * 4;
    sc = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(sc) + 255) & ~uintptr_t(255));
    _Float16* att16 = reinterpret_cast<_Float16*>(sc); sc += (size_t)M * inner * 4;
    _Float16* x16 = reinterpret_cast<_Float16*>(sc); sc += (size_t)M * C * 4;
    _Float16* xo16 = reinterpret_cast<_Float16*>(sc); sc += (size_t)M * C * 4;
    _Float16* q16 = reinterpret_cast<_Float16*>(sc); sc += (size_t)M * n_qkv * 4;
    sc = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(sc) + 255) & ~uintptr_t(255));
    float* d_part = reinterpret_cast<float*>(sc); sc += 2 * ((size_t)M + 64) * C * 4;
    unsigned int* d_flag = reinterpret_cast<unsigned int*>(sc);
    if (pair) HIP_OK(hipMemsetAsync(d_flag, 0, 2 * ((size_t)M / 32 + 2) * 4, s));
    HIP_OK(hipMemcpyAsync(d_stream, hs.data(), hs.size() * 2, hipMemcpyHostToDevice, s));
    HIP_OK(hipMemcpyAsync(d_c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s));                      // (the host vectors go out of scope)
    if (inner) HIP_OK(launch_to_p16(d_att, inner, nullptr, M, inner, inner, att16, 2 * inner, 2048.0f, s));
    HIP_OK(launch_to_p16(d_x, C, nullptr, M, C, C, x16, 2 * C, 2048.0f, s));
    ChainArgs a;
    a.M = M; a.C = C; a.inner = inner; a.att16 = att16; a.ld_att = 2 * inner; a.x16 = x16; a.ld_x = 2 * C;
    a.wstream = d_stream; a.stream_frags = frags;
    a.consts = d_c;
    if (n_qkv) { a.wsum_qkv = d_c + 18 * C; a.b_qkv = d_c + 18 * C + n_qkv; a.n_qkv = n_qkv; a.qkv16 = q16; a.ld_qkv = 2 * n_qkv; }
    a.x_out = xo16; a.ld_out = 2 * C; a.x_out_mask = d_out_mask;
    a.qb = qb; a.ch = ch;
    a.pf_wgs = chain_prefetch_wgs();
    if (pair) { a.pair = 1; a.pair_part = d_part; a.pair_flag = d_flag; a.pair_epoch = 1; a.pf_wgs = a.pf_wgs ? 16 : 0; }
#ifdef MTTS_CHAIN_STAMP
    a.kstamp = reinterpret_cast<unsigned long long*>(d_qkv_out);      // (diagnostic build: the stamps land in the q|k|v output buffer)
#endif
    HIP_OK(launch_tblock_chain(a, s));
#ifdef MTTS_CHAIN_STAMP
    HIP_OK(hipStreamSynchronize(s));
    return 0;
#endif
    if (repeat > 0 && h_ms) {                             // measurement: `repeat` further launches between two events
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipEventRecord(e0, s));
        for (int i = 0; i < repeat; ++i) { if (pair) a.pair_epoch = 2 + i; HIP_OK(launch_tblock_chain(a, s)); }
        HIP_OK(hipEventRecord(e1, s));
        HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipEventElapsedTime(h_ms, e0, e1));
        *h_ms /= (float)repeat;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    HIP_OK(launch_from_p16(xo16, 2 * C, M, C, 2048.0f, d_x_out, C, s));
    if (n_qkv && d_qkv_out) HIP_OK(launch_from_p16(q16, 2 * n_qkv, M, n_qkv, 1.0f, d_qkv_out, n_qkv, s));
    return 0;
}

int mtts_tblock_chain_timed(const float* d_att, const float* d_x, int M, int C, int inner, const float* h_w_out, const float* h_b_out,
                            const float* h_w1, const float* h_b1, const float* h_p0, const float* h_p1, const float* h_w2, const float* h_b2,
                            const float* h_w_qkv, const float* h_b_qkv, int n_qkv, const float* d_out_mask, int qb, int ch, float* d_x_out,
                            float* d_qkv_out, void* d_scratch, void* stream, int repeat, float* h_ms) {
    return tblock_chain_entry(false, d_att, d_x, M, C, inner, h_w_out, h_b_out, h_w1, h_b1, h_p0, h_p1, h_w2, h_b2, h_w_qkv, h_b_qkv, n_qkv,
                              d_out_mask, qb, ch, d_x_out, d_qkv_out, d_scratch, stream, repeat, h_ms);
}
// the pair form of the same launch (two workgroups per row tile; ChainArgs::pair): qb = 48 or 32, at most 120 row tiles
int mtts_tblock_chain_pair_timed(const float* d_att, const float* d_x, int M, int C, int inner, const float* h_w_out, const float* h_b_out,
                                 const float* h_w1, const float* h_b1, const float* h_p0, const float* h_p1, const float* h_w2, const float* h_b2,
                                 const float* h_w_qkv, const float* h_b_qkv, int n_qkv, const float* d_out_mask, int qb, int ch, float* d_x_out,
                                 float* d_qkv_out, void* d_scratch, void* stream, int repeat, float* h_ms) {
    return tblock_chain_entry(true, d_att, d_x, M, C, inner, h_w_out, h_b_out, h_w1, h_b1, h_p0, h_p1, h_w2, h_b2, h_w_qkv, h_b_qkv, n_qkv,
                              d_out_mask, qb, ch, d_x_out, d_qkv_out, d_scratch, stream, repeat, h_ms);
}

// ------------------------------------------------------------------------------------------------ measurement
int mtts_gemm_terms(mtts_ctx* c) { return c ? (c->bf16 ? 17 : c->half16 ? 16 : c->fast16 ? 1 : c->gemm_terms) : default_gemm_terms(); }

int mtts_prof_enable(mtts_ctx* c, int on) {
    if (!c) { set_error("null context"); return -1; }
    c->prof_on = on != 0;
    return 0;
}
int mtts_prof_reset(mtts_ctx* c) {
    if (!c) { set_error("null context"); return -1; }
    c->prof.clear();
    c->ev_used = 0;
    return 0;
}
int mtts_prof_read(mtts_ctx* c, int klass, int64_t* launches, double* ms, double* flops, double* bytes) {
    if (!c) { set_error("null context"); return -1; }
    int64_t n = 0;
    double t = 0, f = 0, by = 0;
    if (!c->prof.empty()) HIP_OK(hipEventSynchronize(c->prof.back().e1));
    for (const ProfRec& r : c->prof) {
        if (r.klass != klass) continue;
        float el = 0.f;
        HIP_OK(hipEventElapsedTime(&el, r.e0, r.e1));
        t += el;
        f += r.flops;
        by += r.bytes;
        ++n;
    }
    if (launches) *launches = n;
    if (ms) *ms = t;
    if (flops) *flops = f;
    if (bytes) *bytes = by;
    return 0;
}

// Per-launch records of the event pass, in launch order: out[i] = (class, ms, flops, bytes); returns the number written.
int64_t mtts_prof_records(mtts_ctx* c, double* out, int64_t max_records) {
    if (!c || !out) { set_error("mtts_prof_records: bad argument"); return -1; }
    if (!c->prof.empty()) HIP_OK(hipEventSynchronize(c->prof.back().e1));
    int64_t n = 0;
    for (const ProfRec& r : c->prof) {
        if (n >= max_records) break;
        float el = 0.f;
        HIP_OK(hipEventElapsedTime(&el, r.e0, r.e1));
        out[4 * n] = r.klass; out[4 * n + 1] = el; out[4 * n + 2] = r.flops; out[4 * n + 3] = r.bytes;
        ++n;
    }
    return n;
}

// the instantiation names of the same records (kernels.h g_kernel_tag), '\n'-separated, "-" for untagged launches
int64_t mtts_prof_tags(mtts_ctx* c, char* out, int64_t max_bytes) {
    if (!c || !out || max_bytes < 2) { set_error("mtts_prof_tags: bad argument"); return -1; }
    std::string all;
    for (const ProfRec& r : c->prof) { all += r.tag.empty() ? "-" : r.tag.c_str(); all += '\n'; }
    if ((int64_t)all.size() + 1 > max_bytes) { set_error("mtts_prof_tags: buffer too small"); return -1; }
    std::memcpy(out, all.c_str(), all.size() + 1);
    return (int64_t)c->prof.size();
}

// ================================================================================================ Vocos head
static int vocos_pack(mtts_vocos* v) {
    mtts_ctx* c = &v->base;
    c->image.clear();
    Packer P(c);
    VocosW& W = v->w;
    W = VocosW();
    const int C = v->dim, nb = v->n_fft / 2 + 1;
    auto S = [](const std::string& a, int i, const std::string& b) { return a + std::to_string(i) + b; };
    W.embed = P.panel("backbone.embed.weight", "backbone.embed.bias", 1, C, v->n_mels, 7);
    W.norm_g = P.vec("backbone.norm.weight", C);
    W.norm_b = P.vec("backbone.norm.bias", C);
    for (int i = 0; i < v->layers; ++i) {
        const std::string p = S("backbone.convnext.", i, ".");
        // depthwise weight [C,1,7] -> [7][C] so that a lane's 4 channels are one float4 per tap
        const auto* dw = P.get(p + "dwconv.weight", (size_t)C * 7);
        if (!dw) break;
        Vec wv;
        wv.off = P.alloc((size_t)7 * C);
        wv.n = 7 * C;
        for (int ch = 0; ch < C; ++ch)
            for (int j = 0; j < 7; ++j) c->image[wv.off + (size_t)j * C + ch] = (*dw)[(size_t)ch * 7 + j];
        W.dw_w.push_back(wv);
        W.dw_b.push_back(P.vec(p + "dwconv.bias", C));
        W.ln_g.push_back(P.vec(p + "norm.weight", C));
        W.ln_b.push_back(P.vec(p + "norm.bias", C));
        W.pw1.push_back(P.panel(p + "pwconv1.weight", p + "pwconv1.bias", 0, v->inter, C, 1));
        // layer scale folded into pwconv2: gamma * (W x + b) = (gamma W) x + gamma b
        const auto* w2 = P.get(p + "pwconv2.weight", (size_t)C * v->inter);
        const auto* b2 = P.get(p + "pwconv2.bias", C);
        const auto* gm = P.get(p + "gamma", C);
        if (!w2 || !b2 || !gm) break;
        std::vector<float> ws(w2->size()), bs(C);
        for (int n = 0; n < C; ++n) {
            for (int k = 0; k < v->inter; ++k) ws[(size_t)n * v->inter + k] = (*gm)[n] * (*w2)[(size_t)n * v->inter + k];
            bs[n] = (*gm)[n] * (*b2)[n];
        }
        W.pw2.push_back(P.panel_from(ws.data(), bs.data(), 0, C, v->inter, 1));
    }
    W.fin_g = P.vec("backbone.final_layer_norm.weight", C);
    W.fin_b = P.vec("backbone.final_layer_norm.bias", C);
    // head: rows [log-magnitude 0..nb) | phase nb..2nb) re-spaced so that both halves start on a multiple of 4 columns
    v->im_off = round_up(nb, 4);
    v->ld_spec = round_up(v->im_off + nb, 4);
    {
        const auto* hw = P.get("head.out.weight", (size_t)2 * nb * C);
        const auto* hb = P.get("head.out.bias", (size_t)2 * nb);
        if (hw && hb) {
            std::vector<float> ws((size_t)v->ld_spec * C, 0.f), bs(v->ld_spec, 0.f);
            for (int r = 0; r < 2 * nb; ++r) {
                const int dst = r < nb ? r : v->im_off + (r - nb);
                std::memcpy(&ws[(size_t)dst * C], &(*hw)[(size_t)r * C], C * sizeof(float));
                bs[dst] = (*hb)[r];
            }
            W.head = P.panel_from(ws.data(), bs.data(), 0, v->ld_spec, C, 1);
        }
    }
    // inverse real DFT (torch.fft.irfft, norm "backward") times the synthesis window, as a [n_fft][ld_spec] matrix:
    // frame[n] = w[n]/N * sum_k c_k (Re_k cos(2 pi k n / N) - Im_k sin(2 pi k n / N)), c_0 = c_{N/2} = 1, else 2
    W.window = P.vec("aux.window", v->n_fft);
    if (P.ok) {
        const int N = v->n_fft;
        std::vector<float> bm((size_t)N * v->ld_spec, 0.f);
        const float* win = &c->image[W.window.off];
        const double two_pi = 6.283185307179586476925286766559;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < nb; ++k) {
                const double ck = (k == 0 || k == N / 2) ? 1.0 : 2.0;
                const double ang = two_pi * (double)(((long long)k * n) % N) / (double)N;
                bm[(size_t)n * v->ld_spec + k] = (float)((double)win[n] * ck * std::cos(ang) / N);
                bm[(size_t)n * v->ld_spec + v->im_off + k] = (float)(-(double)win[n] * ck * std::sin(ang) / N);
            }
        W.basis = P.panel_from(bm.data(), nullptr, 0, N, v->ld_spec, 1);
    }
    if (!P.ok) { set_error(P.why); return -1; }
    c->packed = true;
    return 0;
}

struct VocosBufs { float *MEL, *X, *Y, *H, *SPEC, *FR; };
static void vocos_plan(const mtts_vocos* v, int B, int T, WS& ws, VocosBufs& b) {
    const size_t M = (size_t)B * T;
    b.MEL = ws.f(M * round_up(v->n_mels, 4));
    b.X = ws.f(M * v->dim); b.Y = ws.f(M * v->dim); b.H = ws.f(M * v->inter);
    b.SPEC = ws.f(M * v->ld_spec); b.FR = ws.f(M * v->n_fft);
}

mtts_vocos* mtts_vocos_create(int n_mels, int dim, int inter, int layers, int n_fft, int hop) {
    if (n_mels <= 0 || (n_mels & 3) || dim <= 0 || (dim & 3) || dim > 2048 || inter <= 0 || (inter & 3) || layers < 0 || n_fft <= 0 ||
        (n_fft & 3) || hop <= 0 || n_fft % hop) {
        set_error("mtts_vocos_create: unsupported shape (channels multiples of 4, dim <= 2048, hop divides n_fft)");
        return nullptr;
    }
    mtts_vocos* v = new mtts_vocos();
    v->base.gemm_terms = default_gemm_terms();
    v->n_mels = n_mels; v->dim = dim; v->inter = inter; v->layers = layers; v->n_fft = n_fft; v->hop = hop;
    return v;
}
void mtts_vocos_destroy(mtts_vocos* v) {
    if (!v) return;
    for (hipEvent_t e : v->base.ev_pool) (void)hipEventDestroy(e);
    delete v;
}
int mtts_vocos_set_tensor(mtts_vocos* v, const char* key, const float* h, int64_t numel) {
    if (!v) { set_error("null context"); return -1; }
    return mtts_set_tensor(&v->base, key, h, numel);
}
int64_t mtts_vocos_weights_bytes(mtts_vocos* v) {
    if (!v) { set_error("null context"); return -1; }
    if (!v->base.packed && vocos_pack(v)) return -1;
    return (int64_t)(v->base.image.size() * sizeof(float));
}
int mtts_vocos_upload_weights(mtts_vocos* v, void* d_weights, int64_t bytes) {
    if (!v || !d_weights) { set_error("mtts_vocos_upload_weights: bad argument"); return -1; }
    if (!v->base.packed && vocos_pack(v)) return -1;
    mtts_ctx* c = &v->base;
    if ((size_t)bytes < c->image.size() * sizeof(float)) { set_error("weight buffer too small"); return -1; }
    HIP_OK(hipMemcpy(d_weights, c->image.data(), c->image.size() * sizeof(float), hipMemcpyHostToDevice));
    c->d_image = static_cast<float*>(d_weights);
    c->uploaded = true;
    return 0;
}
int64_t mtts_vocos_workspace_bytes(mtts_vocos* v, int B, int T) {
    if (!v) { set_error("null context"); return -1; }
    WS ws(nullptr, 0);
    VocosBufs b;
    vocos_plan(v, B, T, ws, b);
    return (int64_t)ws.off + 256;
}

// Vocos.decode (reference matcha/vocos24k/vocos_wrapper.py:8-9): mel [B, n_mels, T] -> audio [B, hop*(T-1)]
int mtts_vocos_decode(mtts_vocos* v, const float* d_mel, int B, int T, float* d_audio, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!v) { set_error("null context"); return -1; }
    mtts_ctx* c = &v->base;
    RET_IF(check_ready(c));
    if (T < 2) { set_error("mtts_vocos_decode: need at least 2 frames"); return -1; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    WS ws(d_ws, (size_t)ws_bytes);
    VocosBufs b;
    vocos_plan(v, B, T, ws, b);
    if (ws.overflow) { set_error("vocos workspace too small"); return -1; }
    const VocosW& Wt = v->w;
    const int C = v->dim, M = B * T, ldm = round_up(v->n_mels, 4), nb = v->n_fft / 2 + 1;
    LAUNCH(c, 2, 0, s, launch_cf_to_cl(d_mel, nullptr, B, v->n_mels, T, b.MEL, ldm, 0, s));
    {   // embed: Conv1d(n_mels -> dim, k7, pad 3), then LayerNorm(eps 1e-6)
        GemmArgs a;
        panel_args(c, Wt.embed, a); rows_plain(a, B, T); taps_centered(a, 7);
        a.a0 = b.MEL; a.lda0 = ldm; a.c0 = v->n_mels; a.out = b.Y; a.ldc = C;
        RET_IF(run_gemm(c, a, s));
        LayerNormArgs ln;
        ln.x = b.Y; ln.ldx = C; ln.y = b.X; ln.ldy = C; ln.M = M; ln.C = C; ln.T = T; ln.eps = 1e-6f;
        ln.gamma = W(c, Wt.norm_g.off); ln.beta = W(c, Wt.norm_b.off);
        LAUNCH(c, 2, 0, s, launch_layernorm(ln, s));
    }
    for (int i = 0; i < v->layers; ++i) {   // ConvNeXtBlock: x += gamma * pwconv2(GELU(pwconv1(LN(dwconv(x)))))
        LAUNCH(c, 2, 0, s, launch_dwconv7_ln(b.X, W(c, Wt.dw_w[i].off), W(c, Wt.dw_b[i].off), W(c, Wt.ln_g[i].off), W(c, Wt.ln_b[i].off),
                                             1e-6f, B, T, C, b.Y, s));
        GemmArgs p1;
        panel_args(c, Wt.pw1[i], p1); rows_plain(p1, B, T);
        p1.a0 = b.Y; p1.lda0 = C; p1.c0 = C; p1.act = ACT_GELU; p1.out = b.H; p1.ldc = v->inter;
        RET_IF(run_gemm(c, p1, s));
        GemmArgs p2;
        panel_args(c, Wt.pw2[i], p2); rows_plain(p2, B, T);
        p2.a0 = b.H; p2.lda0 = v->inter; p2.c0 = v->inter; p2.res = b.X; p2.ldr = C; p2.out = b.X; p2.ldc = C;
        RET_IF(run_gemm(c, p2, s));
    }
    {
        LayerNormArgs ln;
        ln.x = b.X; ln.ldx = C; ln.y = b.Y; ln.ldy = C; ln.M = M; ln.C = C; ln.T = T; ln.eps = 1e-6f;
        ln.gamma = W(c, Wt.fin_g.off); ln.beta = W(c, Wt.fin_b.off);
        LAUNCH(c, 2, 0, s, launch_layernorm(ln, s));
        GemmArgs h;   // ISTFTHead.out
        panel_args(c, Wt.head, h); rows_plain(h, B, T);
        h.a0 = b.Y; h.lda0 = C; h.c0 = C; h.out = b.SPEC; h.ldc = v->ld_spec;
        RET_IF(run_gemm(c, h, s));
        LAUNCH(c, 2, 0, s, launch_spec_polar(b.SPEC, M, v->ld_spec, nb, v->im_off, 1e2f, s));
        GemmArgs d;   // irfft * window as a GEMM
        panel_args(c, Wt.basis, d); rows_plain(d, B, T);
        d.a0 = b.SPEC; d.lda0 = v->ld_spec; d.c0 = v->ld_spec; d.out = b.FR; d.ldc = v->n_fft;
        RET_IF(run_gemm(c, d, s));
        LAUNCH(c, 2, 0, s, launch_istft_ola(b.FR, W(c, Wt.window.off), B, T, v->n_fft, v->hop, d_audio, s));
    }
    return 0;
}

}  // extern "C"
