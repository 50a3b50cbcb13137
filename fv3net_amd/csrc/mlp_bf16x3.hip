// EXPERIMENTAL second arithmetic for the fused column MLP: the contraction on the bf16 matrix cores with every fp32
// operand split into three bf16 pieces (x = hi + mid + lo, round-to-nearest; the residuals are exact in fp32) and the six
// significant cross products accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest first:
//     w x  ~=  w_hi x_hi + (w_hi x_mid + w_mid x_hi) + (w_hi x_lo + w_lo x_hi + w_mid x_mid)        (~24 mantissa bits)
// The fp32 MFMA of the product kernel (mlp.hip) peaks at 157 TFLOP/s, 1/16 of the bf16 rate; DESIGN.md section 10 holds
// the single-layer measurement that motivated this (same accuracy as fp32 arithmetic, above the fp32 matrix peak).
// The fp32 kernel stays the product path and the headline; this one is reached only through fv3hip_mlp3_* and is
// validated against the same float64 oracle (tests/test_gpu_mlp.py::test_split_bf16_kernel_against_oracle).
//
// Same graph as mlp.hip (reference lines there): log / centre inputs (1/std folded into the layer-1 weights), Dense + ReLU
// stack, output heads with scale / centre folded in, optional residual outputs `after = before + difference`.
// Restrictions of this first version: hidden width 256, float32 sources and outputs that are sample-contiguous, every
// log epsilon >= FLT_MIN (the fast log), no output limits / masks.  Anything else: FV3HIP_EUNSUPPORTED.
//
// Structure: a workgroup is 4 waves x 32 samples and walks 128-sample tiles persistently.  A layer is a sequence of k-steps
// of 16 contraction indices; per k-step the wave holds its B operand (8 values per lane -> three bf16x8 vectors, split in
// registers) and runs 6 MFMAs per 32-feature output tile; the A operands (the three pre-split weight pieces of the k-step,
// [piece][tile][lane][8 bf16]) come from an LDS buffer the four waves share, double-buffered through registers from one
// packed stream (L2-resident).  The accumulator layout of a layer (lane = sample, registers = features) is, by choice of
// the k-slot -> feature map the host packs the weights with, exactly the B operand layout of the next layer's k-steps:
// activations never leave registers.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"

namespace fv3hip {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHT = 8;          // hidden feature tiles (width 256)
constexpr int kMaxSrc = 16;
constexpr int kMaxOut = 32;

__host__ __device__ constexpr int rho3(int r) { return (r & 3) + 8 * (r >> 2); }

struct Mlp3Launch {
    const f32x4 *w;        // packed stream: per k-step chunk [piece 3][tile][lane 64] float4 (= 8 bf16)
    const float *bias;     // [layer][tile][reg 16][half 2]
    const float *center;   // [n_ks1 * 16] layer-1 centre per k-slot ([ks][half][j])
    const float *eps;      // [n_ks1 * 16] log epsilon per k-slot (log k-steps)
    const int *xsrc;       // [n_ks1 * 16] source index per k-slot, -1 = padding
    const int *xfeat;      // [n_ks1 * 16] feature inside the source
    const int *ofeat;      // [n_ot * 32] (output slot << 20 | feature), -1 = padding
    const int *ores;       // [n_ot * 32] (residual slot << 8 | residual source), -1 = none
    int n_ks1, n_log_ks, n_hidden, n_ot, n_residual;
    int64_t n_samples, n_tiles;
    const float *src[kMaxSrc];
    int64_t src_fs[kMaxSrc];
    float *out[kMaxOut];
    int64_t out_fs[kMaxOut];
};

// three bf16x8 pieces of 8 fp32 values
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)x[j];
        const float r1 = x[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        hi[j] = a;
        mid[j] = b;
        lo[j] = (__bf16)r2;
    }
}

__device__ __forceinline__ void mfma6(f32x16 &acc, const bf16x8 (&a)[3], const bf16x8 &bh, const bf16x8 &bm, const bf16x8 &bl)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bh, acc, 0, 0, 0);
}

template <int OT>  // output tiles held in registers at once (13 for the Zhao-Carr emulator)
__global__ __launch_bounds__(256, 1) void mlp3_kernel(const Mlp3Launch p)
{
    constexpr int CH_H = 3 * kHT * 64;                 // float4 per hidden-type chunk (24 KB)
    constexpr int CH_O = ((3 * OT * 64 + 255) / 256) * 256;  // per output-type chunk, padded to whole rounds of 256 threads
    constexpr int CH_MAX = (CH_H > CH_O) ? CH_H : CH_O;
    constexpr int PER_H = CH_H / 256, PER_O = CH_O / 256, PER_MAX = (PER_H > PER_O) ? PER_H : PER_O;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *wbuf = reinterpret_cast<f32x4 *>(smem);                 // [2][CH_MAX]
    float *cen = reinterpret_cast<float *>(wbuf + 2 * CH_MAX);     // [n_ks1 * 16]
    float *epsl = cen + p.n_ks1 * 16;                              // [n_ks1 * 16]
    int64_t *xrow = reinterpret_cast<int64_t *>(epsl + p.n_ks1 * 16);   // [n_ks1 * 16] byte address of (feature row, sample 0); 0 = padding
    int64_t *orow = xrow + p.n_ks1 * 16;                           // [n_ot * 32] output row address, 0 = none
    int64_t *rsrc = orow + p.n_ot * 32;                            // [n_ot * 32] residual `before` row, 0 = none
    int64_t *rout = rsrc + p.n_ot * 32;                            // [n_ot * 32] residual `after` row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    for (int i = tid; i < p.n_ks1 * 16; i += 256) {
        cen[i] = p.center[i];
        epsl[i] = p.eps[i];
        const int s = p.xsrc[i];
        xrow[i] = s < 0 ? 0 : reinterpret_cast<int64_t>(p.src[s]) + (int64_t)p.xfeat[i] * p.src_fs[s] * 4;
    }
    for (int i = tid; i < p.n_ot * 32; i += 256) {
        const int of = p.ofeat[i], rs = p.ores[i];
        orow[i] = of < 0 ? 0 : reinterpret_cast<int64_t>(p.out[of >> 20]) + (int64_t)(of & 0xFFFFF) * p.out_fs[of >> 20] * 4;
        if (of >= 0 && rs >= 0) {
            const int feat = of & 0xFFFFF;
            rsrc[i] = reinterpret_cast<int64_t>(p.src[rs & 0xFF]) + (int64_t)feat * p.src_fs[rs & 0xFF] * 4;
            rout[i] = reinterpret_cast<int64_t>(p.out[rs >> 8]) + (int64_t)feat * p.out_fs[rs >> 8] * 4;
        } else {
            rsrc[i] = 0;
            rout[i] = 0;
        }
    }
    __syncthreads();

    // ---- the weight stream: chunk g of a tile's G = n_ks1 + 16 (n_hidden - 1) + 16 chunks ----
    const int n_hid_chunks = p.n_ks1 + 16 * (p.n_hidden - 1);
    const int G = n_hid_chunks + 16;
    f32x4 stage[PER_MAX];
    auto chunk_off = [&](int g) -> size_t { return g < n_hid_chunks ? (size_t)g * CH_H : (size_t)n_hid_chunks * CH_H + (size_t)(g - n_hid_chunks) * CH_O; };
    auto issue_w = [&](int g) {
        const f32x4 *src = p.w + chunk_off(g) + tid;
        if (g < n_hid_chunks) {
#pragma unroll
            for (int i = 0; i < PER_H; ++i) stage[i] = src[i * 256];
        } else {
#pragma unroll
            for (int i = 0; i < PER_O; ++i) stage[i] = src[i * 256];
        }
    };
    auto commit_w = [&](int g, int buf) {
        f32x4 *dst = wbuf + (size_t)buf * CH_MAX + tid;
        if (g < n_hid_chunks) {
#pragma unroll
            for (int i = 0; i < PER_H; ++i) dst[i * 256] = stage[i];
        } else {
#pragma unroll
            for (int i = 0; i < PER_O; ++i) dst[i * 256] = stage[i];
        }
    };
    int par = 0;
    issue_w(0);
    commit_w(0, 0);
    __syncthreads();
    // one k-step of NT output tiles: 6 MFMAs per tile on the chunk in wbuf[par]; the next chunk is requested before and
    // committed after them; one barrier per k-step
    auto kstep = [&](auto &acc, auto nt_c, int g, const bf16x8 &bh, const bf16x8 &bm, const bf16x8 &bl) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_c)::value;
        const int gn = (g + 1 < G) ? g + 1 : 0;
        issue_w(gn);
        const f32x4 *lw = wbuf + (size_t)par * CH_MAX + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bf16x8 a[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) a[q] = __builtin_bit_cast(bf16x8, lw[(q * NT + t) * 64]);
            mfma6(acc[t], a, bh, bm, bl);
        }
        commit_w(gn, par ^ 1);
        __syncthreads();
        par ^= 1;
    };

    for (int64_t tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int64_t n = tile * 128 + wave * 32 + col;
        const bool valid = n < p.n_samples;
        const int64_t nb = (valid ? n : p.n_samples - 1) * 4;  // byte offset of this lane's sample inside a row
        int g = 0;
        // ================= layer 1 =================
        f32x16 h[kHT];
#pragma unroll
        for (int t = 0; t < kHT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[t][r] = p.bias[(t * 16 + r) * 2 + half];
        float xn[8];
        auto load_x = [&](int ks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t row = xrow[ks * 16 + half * 8 + j];
                xn[j] = row ? *reinterpret_cast<const float *>(row + nb) : 0.f;
            }
        };
        load_x(0);
        for (int ks = 0; ks < p.n_ks1; ++ks) {
            float x[8];
            const float *c = cen + ks * 16 + half * 8;
            if (ks < p.n_log_ks) {
                const float *e = epsl + ks * 16 + half * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = xn[j] < e[j] ? e[j] : xn[j];
                    x[j] = __builtin_amdgcn_logf(v) * 0.693147180559945f - c[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = xn[j] - c[j];
            }
            if (ks + 1 < p.n_ks1) load_x(ks + 1);
            bf16x8 bh, bm, bl;
            split3(x, bh, bm, bl);
            kstep(h, std::integral_constant<int, kHT>{}, g, bh, bm, bl);
            ++g;
        }
#pragma unroll
        for (int t = 0; t < kHT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[t][r] = h[t][r] < 0.f ? 0.f : h[t][r];
        // ================= hidden -> hidden =================
        for (int l = 1; l < p.n_hidden; ++l) {
            f32x16 h2[kHT];
            const float *bl_ = p.bias + l * kHT * 32;
#pragma unroll
            for (int t = 0; t < kHT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h2[t][r] = bl_[(t * 16 + r) * 2 + half];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = h[ks / 2][(ks % 2) * 8 + j];
                bf16x8 bh, bm, blo;
                split3(x, bh, bm, blo);
                kstep(h2, std::integral_constant<int, kHT>{}, g, bh, bm, blo);
                ++g;
            }
#pragma unroll
            for (int t = 0; t < kHT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = h2[t][r] < 0.f ? 0.f : h2[t][r];
        }
        // ================= hidden -> outputs =================
        f32x16 y[OT];
        {
            const float *bo = p.bias + p.n_hidden * kHT * 32;
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[t][r] = bo[(t * 16 + r) * 2 + half];
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = h[ks / 2][(ks % 2) * 8 + j];
            bf16x8 bh, bm, blo;
            split3(x, bh, bm, blo);
            kstep(y, std::integral_constant<int, OT>{}, g, bh, bm, blo);
            ++g;
        }
        // ================= epilogue: direct stores (a row of a wave = 32 samples = 128 bytes) =================
        if (valid) {
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                float before[16];
                if (p.n_residual) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t rs = rsrc[t * 32 + rho3(r) + 4 * half];
                        before[r] = rs ? *reinterpret_cast<const float *>(rs + nb) : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = t * 32 + rho3(r) + 4 * half;
                    const int64_t o = orow[idx];
                    if (o) *reinterpret_cast<float *>(o + nb) = y[t][r];
                    if (p.n_residual) {
                        const int64_t ro = rout[idx];
                        if (ro) *reinterpret_cast<float *>(ro + nb) = before[r] + y[t][r];
                    }
                }
            }
        }
    }
}

inline unsigned short bf16_rne(float x)
{
    uint32_t b;
    memcpy(&b, &x, 4);
    if ((b & 0x7F800000u) == 0x7F800000u) return (unsigned short)(b >> 16);  // inf / nan
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (unsigned short)(b >> 16);
}
inline float bf16_val(unsigned short h)
{
    const uint32_t b = (uint32_t)h << 16;
    float x;
    memcpy(&x, &b, 4);
    return x;
}

}  // namespace
}  // namespace fv3hip

using namespace fv3hip;

struct fv3hip_mlp3 {
    int device = 0, n_cu = 256;
    int n_sources = 0, n_outputs = 0, n_residual = 0, n_hidden = 0, n_ks1 = 0, n_log_ks = 0, n_ot = 0;
    int64_t flops = 0;
    void *d_w = nullptr, *d_bias = nullptr, *d_center = nullptr, *d_eps = nullptr, *d_xsrc = nullptr, *d_xfeat = nullptr,
         *d_ofeat = nullptr, *d_ores = nullptr;
    size_t lds_bytes = 0;
};

namespace {
template <typename T>
int upload3(const std::vector<T> &v, void **dptr)
{
    *dptr = nullptr;
    if (v.empty()) return FV3HIP_OK;
    FV3HIP_CHECK_HIP(hipMalloc(dptr, v.size() * sizeof(T)));
    FV3HIP_CHECK_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return FV3HIP_OK;
}
}  // namespace

extern "C" int fv3hip_mlp3_destroy(fv3hip_mlp3_t m)
{
    if (!m) return FV3HIP_OK;
    for (void *q : {m->d_w, m->d_bias, m->d_center, m->d_eps, m->d_xsrc, m->d_xfeat, m->d_ofeat, m->d_ores})
        if (q) hipFree(q);
    delete m;
    return FV3HIP_OK;
}

extern "C" int64_t fv3hip_mlp3_flops_per_sample(fv3hip_mlp3_t m) { return m ? m->flops : 0; }

extern "C" int fv3hip_mlp3_create(const fv3hip_mlp_desc_t *d, fv3hip_mlp3_t *out)
{
    FV3HIP_REQUIRE(d && out, "null pointer");
    *out = nullptr;
    FV3HIP_REQUIRE(d->n_sources >= 1 && d->n_sources <= kMaxSrc && d->n_inputs >= 1 && d->n_outputs >= 1, "bad counts");
    FV3HIP_REQUIRE(d->n_outputs + d->n_residual <= kMaxOut, "too many outputs");
    if (d->width != 256 || d->n_hidden < 1 || d->hidden_activation != FV3HIP_ACT_RELU || d->hidden_output)
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel takes ReLU networks of hidden width 256 without a hidden output");
    if (d->out_min || d->out_max || d->out_mask)
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel does not implement output limits / masks");
    const int W = 256;
    int K = 0, F = 0;
    for (int i = 0; i < d->n_inputs; ++i) K += d->in_nfeat[i];
    for (int j = 0; j < d->n_outputs; ++j) F += d->out_nfeat[j];
    const int n_ot = (F + 31) / 32;
    if (n_ot != 13 && n_ot != 3 && n_ot != 5)
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel is compiled for 3, 5 or 13 output tiles of 32 (got %d outputs)", F);
    // k-slots of layer 1: log-transformed features first, padded to whole k-steps of 16, then the others
    struct Slot { int src, feat; float center, rscale, eps; int orig; };
    std::vector<Slot> slots;
    {
        int k = 0;
        std::vector<Slot> logs, plain;
        for (int i = 0; i < d->n_inputs; ++i)
            for (int f = 0; f < d->in_nfeat[i]; ++f, ++k) {
                Slot s{d->in_source[i], d->in_feat_start[i] + f, d->in_center ? d->in_center[k] : 0.f,
                       d->in_scale ? (float)(1.0 / (double)d->in_scale[k]) : 1.f, d->in_eps ? d->in_eps[i] : 0.f, k};
                const bool is_log = d->in_transform && d->in_transform[i] == FV3HIP_TRANSFORM_LOG;
                if (is_log && !(s.eps >= FLT_MIN))
                    return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel needs log epsilons >= FLT_MIN (the fast logarithm)");
                (is_log ? logs : plain).push_back(s);
            }
        slots = logs;
        while (slots.size() % 16) slots.push_back(Slot{-1, 0, 0.f, 0.f, 1.f, -1});
        const int n_log_slots = (int)slots.size();
        slots.insert(slots.end(), plain.begin(), plain.end());
        while (slots.size() % 16) slots.push_back(Slot{-1, 0, 0.f, 0.f, 1.f, -1});
        fv3hip_mlp3 *m = new fv3hip_mlp3();
        hipGetDevice(&m->device);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, m->device) == hipSuccess) m->n_cu = prop.multiProcessorCount;
        m->n_sources = d->n_sources;
        m->n_outputs = d->n_outputs;
        m->n_residual = d->n_residual;
        m->n_hidden = d->n_hidden;
        m->n_ks1 = (int)slots.size() / 16;
        m->n_log_ks = n_log_slots / 16;
        m->n_ot = n_ot;
        m->flops = 2 * ((int64_t)K * W + (int64_t)(d->n_hidden - 1) * W * W + (int64_t)W * F);
        *out = m;
    }
    fv3hip_mlp3 *m = *out;
    const int n_ks1 = m->n_ks1;
    const int CH_H = 3 * kHT * 64, CH_O = ((3 * n_ot * 64 + 255) / 256) * 256;
    const int n_hid_chunks = n_ks1 + 16 * (d->n_hidden - 1);
    std::vector<unsigned short> w(((size_t)n_hid_chunks * CH_H + (size_t)16 * CH_O) * 8, 0);
    auto put = [&](size_t chunk_base_f4, int nt, int t, int lane, int j, float value) {
        float r = value;
        for (int q = 0; q < 3; ++q) {
            const unsigned short b = bf16_rne(r);
            w[(chunk_base_f4 + (size_t)(q * nt + t) * 64 + lane) * 8 + j] = b;
            r -= bf16_val(b);
        }
    };
    // layer 1: k-slot (ks, half, j) = slots[16 ks + 8 half + j]
    for (int ks = 0; ks < n_ks1; ++ks)
        for (int t = 0; t < kHT; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const Slot &s = slots[ks * 16 + (lane >> 5) * 8 + j];
                    const int f = 32 * t + (lane & 31);
                    put((size_t)ks * CH_H, kHT, t, lane, j, s.orig < 0 ? 0.f : d->hidden_kernels[0][(size_t)s.orig * W + f] * s.rscale);
                }
    // hidden layers and the output layer: k-slot (ks, half, j) = feature 32 (ks / 2) + rho(8 (ks % 2) + j) + 4 half of the
    // previous layer (its accumulator layout)
    auto kfeat = [](int ks, int hf, int j) { return 32 * (ks / 2) + rho3(8 * (ks % 2) + j) + 4 * hf; };
    for (int l = 1; l < d->n_hidden; ++l)
        for (int ks = 0; ks < 16; ++ks)
            for (int t = 0; t < kHT; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        put((size_t)(n_ks1 + 16 * (l - 1) + ks) * CH_H, kHT, t, lane, j,
                            d->hidden_kernels[l][(size_t)kfeat(ks, lane >> 5, j) * W + 32 * t + (lane & 31)]);
    for (int ks = 0; ks < 16; ++ks)
        for (int t = 0; t < n_ot; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int f = 32 * t + (lane & 31);
                    put((size_t)n_hid_chunks * CH_H + (size_t)ks * CH_O, n_ot, t, lane, j,
                        f < F ? d->out_kernel[(size_t)kfeat(ks, lane >> 5, j) * F + f] * (d->out_scale ? d->out_scale[f] : 1.f) : 0.f);
                }
    std::vector<float> bias((size_t)d->n_hidden * kHT * 32 + (size_t)n_ot * 32, 0.f);
    for (int l = 0; l < d->n_hidden; ++l)
        for (int t = 0; t < kHT; ++t)
            for (int r = 0; r < 16; ++r)
                for (int hf = 0; hf < 2; ++hf) bias[(size_t)l * kHT * 32 + (t * 16 + r) * 2 + hf] = d->hidden_biases[l][32 * t + rho3(r) + 4 * hf];
    for (int t = 0; t < n_ot; ++t)
        for (int r = 0; r < 16; ++r)
            for (int hf = 0; hf < 2; ++hf) {
                const int f = 32 * t + rho3(r) + 4 * hf;
                if (f < F)
                    bias[(size_t)d->n_hidden * kHT * 32 + (t * 16 + r) * 2 + hf] =
                        (float)((double)d->out_bias[f] * (d->out_scale ? d->out_scale[f] : 1.f) + (d->out_center ? d->out_center[f] : 0.f));
            }
    std::vector<float> center(slots.size()), eps(slots.size());
    std::vector<int> xsrc(slots.size()), xfeat(slots.size());
    for (size_t i = 0; i < slots.size(); ++i) {
        center[i] = slots[i].orig < 0 ? 0.f : slots[i].center;
        eps[i] = slots[i].eps;
        xsrc[i] = slots[i].orig < 0 ? -1 : slots[i].src;
        xfeat[i] = slots[i].feat;
    }
    // (a padding slot of a log k-step: x = 0 -> max(0, eps = 1) = 1 -> log 1 = 0, centre 0: exactly 0 times a zero weight)
    std::vector<int> ofeat((size_t)n_ot * 32, -1), ores((size_t)n_ot * 32, -1);
    {
        int f = 0;
        for (int j = 0; j < d->n_outputs; ++j) {
            int res = -1;
            for (int r = 0; r < d->n_residual; ++r)
                if (d->res_output[r] == j) res = ((d->n_outputs + r) << 8) | d->res_source[r];
            for (int q = 0; q < d->out_nfeat[j]; ++q, ++f) {
                ofeat[f] = (j << 20) | q;
                ores[f] = res;
            }
        }
    }
    int rc;
    if ((rc = upload3(w, &m->d_w)) || (rc = upload3(bias, &m->d_bias)) || (rc = upload3(center, &m->d_center)) ||
        (rc = upload3(eps, &m->d_eps)) || (rc = upload3(xsrc, &m->d_xsrc)) || (rc = upload3(xfeat, &m->d_xfeat)) ||
        (rc = upload3(ofeat, &m->d_ofeat)) || (rc = upload3(ores, &m->d_ores))) {
        fv3hip_mlp3_destroy(m);
        *out = nullptr;
        return rc;
    }
    const size_t ch_max = (size_t)((CH_H > CH_O) ? CH_H : CH_O);
    m->lds_bytes = 2 * ch_max * 16 + (size_t)slots.size() * (4 + 4 + 8) + (size_t)n_ot * 32 * 24;
    if (m->lds_bytes > 160 * 1024) {
        fv3hip_mlp3_destroy(m);
        *out = nullptr;
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel's tables need %zu bytes of LDS (> 160 KiB)", m->lds_bytes);
    }
    return FV3HIP_OK;
}

extern "C" int fv3hip_mlp3_predict(fv3hip_mlp3_t m, const void *const *sources, const int64_t *src_feat_stride, int64_t n_samples,
                                   void *const *outputs, const int64_t *out_feat_stride, void *stream)
{
    FV3HIP_REQUIRE(m, "null model handle");
    FV3HIP_REQUIRE(n_samples >= 0, "negative n_samples");
    if (n_samples == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(sources && src_feat_stride && outputs && out_feat_stride, "null pointer");
    {
        int cur = -1;
        FV3HIP_CHECK_HIP(hipGetDevice(&cur));
        FV3HIP_REQUIRE(cur == m->device, "the model lives on device %d but the current device is %d", m->device, cur);
    }
    Mlp3Launch lp;
    memset(&lp, 0, sizeof(lp));
    for (int i = 0; i < m->n_sources; ++i) {
        FV3HIP_REQUIRE(sources[i], "source %d is null", i);
        lp.src[i] = static_cast<const float *>(sources[i]);
        lp.src_fs[i] = src_feat_stride[i];
    }
    for (int j = 0; j < m->n_outputs + m->n_residual; ++j) {
        FV3HIP_REQUIRE(outputs[j], "output %d is null", j);
        lp.out[j] = static_cast<float *>(outputs[j]);
        lp.out_fs[j] = out_feat_stride[j];
    }
    lp.w = static_cast<const f32x4 *>(m->d_w);
    lp.bias = static_cast<const float *>(m->d_bias);
    lp.center = static_cast<const float *>(m->d_center);
    lp.eps = static_cast<const float *>(m->d_eps);
    lp.xsrc = static_cast<const int *>(m->d_xsrc);
    lp.xfeat = static_cast<const int *>(m->d_xfeat);
    lp.ofeat = static_cast<const int *>(m->d_ofeat);
    lp.ores = static_cast<const int *>(m->d_ores);
    lp.n_ks1 = m->n_ks1;
    lp.n_log_ks = m->n_log_ks;
    lp.n_hidden = m->n_hidden;
    lp.n_ot = m->n_ot;
    lp.n_residual = m->n_residual;
    lp.n_samples = n_samples;
    lp.n_tiles = ceil_div(n_samples, 128);
    const int grid = (int)(lp.n_tiles < m->n_cu ? lp.n_tiles : m->n_cu);
    hipStream_t st = as_stream(stream);
#define LAUNCH3_(OT)                                                                                                     \
    {                                                                                                                    \
        auto kern = mlp3_kernel<OT>;                                                                                     \
        FV3HIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_bytes)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), m->lds_bytes, st, lp);                                           \
    }
    if (m->n_ot == 13) LAUNCH3_(13) else if (m->n_ot == 5) LAUNCH3_(5) else LAUNCH3_(3)
#undef LAUNCH3_
    return check_launch("mlp3_kernel");
}
