// Fused column-MLP forward for gfx950 (MI355X) on the fp32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// Reference graph restated (paths relative to the reference checkout):
//   external/fv3fit/fv3fit/keras/_models/dense.py:239-310         build_model (predict_model)
//   external/fv3fit/fv3fit/keras/_models/shared/utils.py:34-105   standard (de)normalize
//   external/fv3fit/fv3fit/emulation/layers/normalization.py:45-49  NormLayer.forward/backward
//   external/fv3fit/fv3fit/keras/_models/shared/dense_network.py:59-81  Dense + ReLU stack
//   external/fv3fit/fv3fit/keras/_models/shared/output_limit.py:29-48, clip.py:33-46
//   external/fv3fit/fv3fit/emulation/models/microphysics.py:123-139, layers/architecture.py:27-50,
//     228-282,304-343, layers/fields.py:33-66, transforms/transforms.py:17-58,111-129
//
// Design (MI355X-first, not a GEMM-library call chain):
//   * the whole network runs in ONE launch; a workgroup is 4 waves, one per SIMD, each wave owns
//     32 samples (columns of the atmosphere) and the full 512-register file;
//   * the contraction is computed TRANSPOSED, D[feature][sample] = W^T[feature][k] * x[k][sample]:
//     the MFMA result then has the sample on the lane and the features in the 16 accumulator
//     registers, which is exactly the B-operand layout of the next layer's MFMA -- activations
//     never leave registers between layers (no LDS round trip, no HBM);
//   * inputs are consumed in their native [feature][sample] layout ([z,y,x] model arrays,
//     call_py_fort's [feature, sample]); any stride pair is accepted, so there is no stack/transpose
//     copy; float64 inputs are rounded to float32 on load; the log transform, the clip slice and
//     the (x - mean)/(std + eps) normalisation are applied to the B operand on the fly;
//   * the weights of all layers are one pre-packed stream of LDS-image chunks (conflict-free
//     ds_read_b128, 16 k-pairs per chunk) that every workgroup walks with a register-staged
//     double buffer: the global loads of chunk g+1 are issued before the MFMAs of chunk g and
//     written to the other LDS buffer after them; one barrier per chunk; the stream wraps around
//     from the last chunk of a sample tile to the first chunk of the next one;
//   * bias enters as the initial accumulator; ReLU, the output (de)normalisation, the range
//     limiter, the zero mask of clipped levels and the residual "after = before + difference"
//     outputs are applied in registers in the epilogue.
#include <cmath>
#include <cstddef>
#include <type_traits>
#include <vector>

#include "common.h"

namespace fv3hip {
namespace {

// Scheduling of one k-pair slot: the matrix pipe takes a 32x32x2 f32 MFMA every 64 cycles and the
// wave issues in order, so the slot's other work (LDS reads of the next A fragments, the input
// prefetch / normalisation, the LDS commits) only overlaps if it sits BETWEEN the MFMAs.  Ask the
// scheduler for (1 MFMA, up to Q others) x NT, then fence the slot.
#define MLP_SLOT_SCHED(NT, Q)                                                          \
    _Pragma("unroll") for (int _i = 0; _i < (NT); ++_i)                                \
    {                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x010 | 0x080 | 0x400, (Q), 0); \
    }                                                                                  \
    __builtin_amdgcn_sched_barrier(0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxSources = 16;
constexpr int kMaxOutputs = 32;
constexpr int kMaxK = 2048;       // network inputs whose per-feature table still fits LDS
constexpr int kThreads = 256;     // 4 waves, one per SIMD
constexpr int kTileSamples = 128; // 32 per wave

struct KEntry {  // one network input feature (32 bytes)
    int src;     // source array, -1 for padding
    int feat;    // feature (level) index inside the source
    float center;
    float scale;
    int transform;
    float eps;
    int pad0, pad1;
};

struct XAddr {  // per call: where network input k is read from (16 bytes)
    int64_t row;  // byte address of (feature, sample 0)
    int64_t ss;   // byte stride between samples
};

struct XNorm {  // per model: what is done to it (16 bytes)
    float center, scale, eps;
    int flags;  // bit 0: log transform, bit 1: a real input (0 = padding -> 0)
};

struct OEntry {  // one network output feature (32 bytes)
    float scale, center, lo, hi;
    float mask;
    int out_feat;  // (output slot << 20) | feature inside the slot; -1 for padding
    int res;       // (residual slot << 8) | residual source; -1 for none
    int pad0;
};

struct MlpLaunch {
    const f32x4 *w;     // packed weight stream
    const KEntry *ktab; // [2 * 16 * n_chunks1]
    const OEntry *otab; // [32 * OC * n_pass]
    const float *bias;  // packed biases
    int n_chunks1;      // layer-1 chunks of 16 k-pairs
    int n_hidden;
    int n_pass;
    int n_ktab;
    int n_otab;
    int n_bias;
    int out64;
    int64_t n_samples;
    int64_t n_tiles;
    const void *src[kMaxSources];
    int64_t src_fs[kMaxSources];
    int64_t src_ss[kMaxSources];
    void *out[kMaxOutputs];
    int64_t out_fs[kMaxOutputs];
    int64_t out_ss[kMaxOutputs];
};

// row of a 32x32 accumulator held by register r of a lane in half h is rho(r) + 4*h
__host__ __device__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

template <int HT, int OC, bool SRC64>
__global__ __launch_bounds__(kThreads, 1) void mlp_fused_kernel(const MlpLaunch p)
{
    constexpr int HG = (HT + 3) / 4;          // float4 groups of hidden-feature tiles
    constexpr int OG = (OC + 3) / 4;          // float4 groups of output-feature tiles
    constexpr int KC_H = 16;                  // k-pairs per hidden-type chunk
    constexpr int KC_O = (OG <= 2) ? 16 : 8;  // k-pairs per output-type chunk
    constexpr int CH_H = KC_H * HG * 64;      // float4 per hidden-type chunk
    constexpr int CH_O = KC_O * OG * 64;      // float4 per output-type chunk
    constexpr int NV_H = CH_H / kThreads;
    constexpr int NV_O = CH_O / kThreads;
    constexpr int NV_MAX = (NV_H > NV_O) ? NV_H : NV_O;
    constexpr int CH_MAX = (CH_H > CH_O) ? CH_H : CH_O;
    constexpr int OHALVES = 16 / KC_O;        // output-type chunks per 32-feature k tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *wbuf = reinterpret_cast<f32x4 *>(smem);                       // [2][CH_MAX]
    XAddr *xa_tab = reinterpret_cast<XAddr *>(wbuf + 2 * CH_MAX);        // [n_ktab] where input k lives
    XNorm *xn_tab = reinterpret_cast<XNorm *>(xa_tab + p.n_ktab);        // [n_ktab] how it is normalised
    OEntry *otab = reinterpret_cast<OEntry *>(xn_tab + p.n_ktab);        // [n_otab]
    float *biasl = reinterpret_cast<float *>(otab + p.n_otab);           // [n_bias]
    int64_t *src_base = reinterpret_cast<int64_t *>(biasl + ((p.n_bias + 3) & ~3));  // [16]
    int64_t *src_fs = src_base + kMaxSources;
    int64_t *src_ss = src_fs + kMaxSources;
    int64_t *out_base = src_ss + kMaxSources;                            // [32]
    int64_t *out_fs = out_base + kMaxOutputs;
    int64_t *out_ss = out_fs + kMaxOutputs;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;

    // ---- one-time prologue: tables to LDS ----
    {
        // Per-feature input tables: the byte address of every network input's row for sample 0
        // (per call) and its normalisation constants, so that a k-pair slot needs one 16-byte LDS
        // read to issue an input load and one to finish it -- no dependent LDS chain in a slot.
        typedef const int64_t __attribute__((address_space(4))) *KargPtr64;
        typedef const char __attribute__((address_space(4))) *KargBytes64;
        KargPtr64 ksrc = (KargPtr64)((KargBytes64)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MlpLaunch, src));
        for (int i = tid; i < p.n_ktab; i += kThreads) {
            const KEntry e = p.ktab[i];
            const int sidx = e.src < 0 ? 0 : e.src;
            const int64_t esz = SRC64 ? 8 : 4;
            XAddr a;
            a.row = ksrc[sidx] + (e.src < 0 ? 0 : (int64_t)e.feat * ksrc[kMaxSources + sidx] * esz);
            a.ss = e.src < 0 ? 0 : ksrc[2 * kMaxSources + sidx] * esz;
            xa_tab[i] = a;
            XNorm nrm;
            nrm.center = e.center;
            nrm.scale = e.scale;
            nrm.eps = e.eps;
            nrm.flags = (e.transform == FV3HIP_TRANSFORM_LOG ? 1 : 0) | (e.src < 0 ? 0 : 2);
            xn_tab[i] = nrm;
        }
        const f32x4 *go = reinterpret_cast<const f32x4 *>(p.otab);
        f32x4 *lo = reinterpret_cast<f32x4 *>(otab);
        for (int i = tid; i < p.n_otab * 2; i += kThreads) lo[i] = go[i];
        for (int i = tid; i < p.n_bias; i += kThreads) biasl[i] = p.bias[i];
        // The pointer/stride tables are indexed per lane later on, so they go to LDS too.  They
        // are read straight from the kernarg segment with vector loads (p is the only kernel
        // argument, at offset 0): indexing p.src[] by thread would pull the whole struct into
        // SGPRs.
        typedef const int64_t __attribute__((address_space(4))) *KargPtr;
        typedef const char __attribute__((address_space(4))) *KargBytes;
        KargPtr ka = (KargPtr)((KargBytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MlpLaunch, src));
        constexpr int kTableWords = 3 * kMaxSources + 3 * kMaxOutputs;  // src, src_fs, src_ss, out, out_fs, out_ss
        if (tid < kTableWords) src_base[tid] = ka[tid];
    }
    __syncthreads();

    // ---- weight-stream helpers ----
    const int n_hid_chunks = p.n_chunks1 + (p.n_hidden - 1) * HT;  // hidden-type chunks per tile
    const int n_out_chunks = p.n_pass * HT * OHALVES;              // output-type chunks per tile
    const int G = n_hid_chunks + n_out_chunks;
    f32x4 stage[NV_MAX];
    auto issue_w = [&](int g) {  // global -> registers for chunk g of the stream
        if (g < n_hid_chunks) {
            const f32x4 *gp = p.w + (int64_t)g * CH_H + tid;
#pragma unroll
            for (int i = 0; i < NV_MAX; ++i)
                if (i < NV_H) stage[i] = gp[i * kThreads];
        } else {
            const f32x4 *gp = p.w + (int64_t)n_hid_chunks * CH_H + (int64_t)(g - n_hid_chunks) * CH_O + tid;
#pragma unroll
            for (int i = 0; i < NV_MAX; ++i)
                if (i < NV_O) stage[i] = gp[i * kThreads];
        }
    };
    auto commit_w = [&](int g, int buf) {  // registers -> LDS buffer
        f32x4 *lp = wbuf + buf * CH_MAX + tid;
        const int nv = (g < n_hid_chunks) ? NV_H : NV_O;
#pragma unroll
        for (int i = 0; i < NV_MAX; ++i)
            if (i < nv) lp[i * kThreads] = stage[i];
    };

    // ---- layer-1 B operand helpers ----
    using Raw = typename std::conditional<SRC64, double, float>::type;
    typedef const Raw __attribute__((address_space(1))) *GRawPtr;
    typedef float __attribute__((address_space(1))) *GF32Ptr;
    typedef double __attribute__((address_space(1))) *GF64Ptr;
    Raw xraw[KC_H];
    float xcur[KC_H];
    // one element (k-pair s of a layer-1 chunk): raw load / transform + normalise, both branch-free
    // so that they can be scheduled between MFMAs
    auto issue_x1 = [&](const XAddr a, int s, int64_t nc) { xraw[s] = *(GRawPtr)(a.row + nc * a.ss); };
    auto finish_x1 = [&](const XNorm e, int s) {
        const float raw = (float)xraw[s];
        const float lg = logf(raw < e.eps ? e.eps : raw);
        float v = (e.flags & 1) ? lg : raw;
        v = (v - e.center) / e.scale;
        xcur[s] = (e.flags & 2) ? v : 0.f;
    };
    auto issue_x = [&](int c, int64_t nc) {  // a whole chunk at once (tile boundaries only)
#pragma unroll
        for (int s = 0; s < KC_H; ++s) issue_x1(xa_tab[2 * (c * KC_H + s) + half], s, nc);
    };
    auto finish_x = [&](int c) {
#pragma unroll
        for (int s = 0; s < KC_H; ++s) finish_x1(xn_tab[2 * (c * KC_H + s) + half], s);
    };
    // one staged float4 of the next chunk -> the other LDS buffer
    // (unconditional: a chunk type with fewer float4s just leaves the tail of the buffer unused)
    auto commit_w1 = [&](int buf, int i) {
        if (i < NV_MAX) wbuf[buf * CH_MAX + tid + i * kThreads] = stage[i];
    };

    int par = 0;  // LDS buffer holding the chunk about to be consumed
    int64_t tile = blockIdx.x;
    if (tile >= p.n_tiles) return;

    // prime the pipeline: chunk 0 of the stream and the first tile's first inputs
    {
        issue_w(0);
        int64_t n = tile * kTileSamples + wave * 32 + (lane & 31);
        issue_x(0, n < p.n_samples ? n : p.n_samples - 1);
        commit_w(0, 0);
        finish_x(0);
        __syncthreads();
    }

    for (; tile < p.n_tiles; tile += gridDim.x) {
        const int64_t n = tile * kTileSamples + wave * 32 + (lane & 31);
        const bool valid = n < p.n_samples;
        const int64_t nc = valid ? n : p.n_samples - 1;
        const int64_t next_tile = tile + gridDim.x;
        int64_t nn = next_tile * kTileSamples + wave * 32 + (lane & 31);
        if (nn >= p.n_samples) nn = p.n_samples - 1;
        int g = 0;

        f32x16 h[HT];
        // ================= layer 1: inputs -> hidden =================
        {
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = biasl[(t * 16 + r) * 2 + half];
            // Software pipeline, one k-pair (HT MFMAs = HT x 64 cycles of matrix pipe) per slot:
            //   * the A fragments of k-pair s+1 are read from LDS before the MFMAs of k-pair s;
            //   * slot s issues the raw input load of element s of the NEXT chunk and finishes
            //     (transform, normalise) the element issued 8 slots earlier, so every global load
            //     has 8 slots (>= 4096 cycles) to land and the VALU work rides under the MFMAs;
            //   * the staged weights of the next chunk go to the other LDS buffer one float4 per
            //     slot in the second half of the chunk.
            for (int c = 0; c < p.n_chunks1; ++c) {
                const int gnext = (g + 1 < G) ? g + 1 : 0;
                issue_w(gnext);
                const int cn = (c + 1 < p.n_chunks1) ? c + 1 : c;
                const f32x4 *lw = wbuf + par * CH_MAX + lane;
                f32x4 a_cur[HG], a_nxt[HG];
#pragma unroll
                for (int j = 0; j < HG; ++j) a_cur[j] = lw[j * 64];
                // table entries are read one slot ahead of their use
                const XAddr *xa_n = xa_tab + 2 * cn * KC_H + half;    // issue: element s of chunk cn
                const XNorm *xn_c = xn_tab + 2 * c * KC_H + half;     // finish, first half: chunk c, s+8
                const XNorm *xn_n = xn_tab + 2 * cn * KC_H + half;    // finish, second half: chunk cn, s-8
                XAddr xa_cur = xa_n[0], xa_nxt = xa_cur;
                XNorm xn_cur = xn_c[2 * (KC_H / 2)], xn_nxt = xn_cur;
#pragma unroll
                for (int s = 0; s < KC_H; ++s) {
                    if (s + 1 < KC_H) {
#pragma unroll
                        for (int j = 0; j < HG; ++j) a_nxt[j] = lw[((s + 1) * HG + j) * 64];
                        xa_nxt = xa_n[2 * (s + 1)];
                        xn_nxt = (s + 1 < KC_H / 2) ? xn_c[2 * (s + 1 + KC_H / 2)] : xn_n[2 * (s + 1 - KC_H / 2)];
                    }
                    const float b = xcur[s];
#pragma unroll
                    for (int t = 0; t < HT; ++t)
                        h[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t / 4][t % 4], b, h[t], 0, 0, 0);
                    // No branches in a slot (the scheduler only interleaves inside a basic block): on
                    // the last chunk the "next chunk" is the chunk itself, which re-derives values
                    // that are already there; re-finishing elements 8..15 of chunk 0 is idempotent.
                    issue_x1(xa_cur, s, nc);
                    if (s < KC_H / 2) {
                        finish_x1(xn_cur, s + KC_H / 2);
                    } else {
                        finish_x1(xn_cur, s - KC_H / 2);
                        commit_w1(par ^ 1, s - KC_H / 2);
                    }
#pragma unroll
                    for (int j = 0; j < HG; ++j) a_cur[j] = a_nxt[j];
                    xa_cur = xa_nxt;
                    xn_cur = xn_nxt;
                    MLP_SLOT_SCHED(HT, 10);
                }
                __syncthreads();
                par ^= 1;
                ++g;
            }
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = (h[t][r] < 0.f) ? 0.f : h[t][r];
        }
        // ================= hidden -> hidden =================
        for (int l = 1; l < p.n_hidden; ++l) {
            f32x16 h2[HT];
            const float *bl = biasl + l * HT * 32;
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h2[t][r] = bl[(t * 16 + r) * 2 + half];
#pragma unroll
            for (int kt = 0; kt < HT; ++kt) {
                const int gnext = (g + 1 < G) ? g + 1 : 0;
                issue_w(gnext);
                const f32x4 *lw = wbuf + par * CH_MAX + lane;
                f32x4 a_cur[HG], a_nxt[HG];
#pragma unroll
                for (int j = 0; j < HG; ++j) a_cur[j] = lw[j * 64];
#pragma unroll
                for (int s = 0; s < KC_H; ++s) {
                    if (s + 1 < KC_H) {
#pragma unroll
                        for (int j = 0; j < HG; ++j) a_nxt[j] = lw[((s + 1) * HG + j) * 64];
                    }
                    const float b = h[kt][s];
#pragma unroll
                    for (int t = 0; t < HT; ++t)
                        h2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t / 4][t % 4], b, h2[t], 0, 0, 0);
                    if (s >= KC_H / 2) commit_w1(par ^ 1, s - KC_H / 2);
#pragma unroll
                    for (int j = 0; j < HG; ++j) a_cur[j] = a_nxt[j];
                    MLP_SLOT_SCHED(HT, 3);
                }
                __syncthreads();
                par ^= 1;
                ++g;
            }
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = (h2[t][r] < 0.f) ? 0.f : h2[t][r];
        }
        // ================= hidden -> outputs, OC feature tiles per pass =================
        for (int pass = 0; pass < p.n_pass; ++pass) {
            f32x16 y[OC];
            const float *bl = biasl + p.n_hidden * HT * 32 + pass * OC * 32;
#pragma unroll
            for (int t = 0; t < OC; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[t][r] = bl[(t * 16 + r) * 2 + half];
            // the next tile's first inputs ride under the last pass of this one
            const bool prefetch_next = (pass + 1 == p.n_pass) && next_tile < p.n_tiles;
            if (prefetch_next) issue_x(0, nn);
#pragma unroll
            for (int kt = 0; kt < HT; ++kt) {
#pragma unroll
                for (int hf = 0; hf < OHALVES; ++hf) {
                    const int gnext = (g + 1 == G) ? 0 : g + 1;
                    issue_w(gnext);
                    const f32x4 *lw = wbuf + par * CH_MAX + lane;
                    f32x4 a_cur[OG], a_nxt[OG];
#pragma unroll
                    for (int j = 0; j < OG; ++j) a_cur[j] = lw[j * 64];
#pragma unroll
                    for (int s = 0; s < KC_O; ++s) {
                        if (s + 1 < KC_O) {
#pragma unroll
                            for (int j = 0; j < OG; ++j) a_nxt[j] = lw[((s + 1) * OG + j) * 64];
                        }
                        const float b = h[kt][hf * KC_O + s];
#pragma unroll
                        for (int t = 0; t < OC; ++t)
                            y[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t / 4][t % 4], b, y[t], 0, 0, 0);
                        // NV_MAX float4 per thread to commit, over the second half of the chunk's slots
                        constexpr int per_slot = (NV_MAX + KC_O / 2 - 1) / (KC_O / 2);
                        if (s >= KC_O / 2) {
#pragma unroll
                            for (int i = 0; i < per_slot; ++i) commit_w1(par ^ 1, (s - KC_O / 2) * per_slot + i);
                        }
#pragma unroll
                        for (int j = 0; j < OG; ++j) a_cur[j] = a_nxt[j];
                        MLP_SLOT_SCHED(OC, 3);
                    }
                    __syncthreads();
                    par ^= 1;
                    ++g;
                }
            }
            if (prefetch_next) finish_x(0);
            // ---- epilogue: denormalise, limit, mask, store (+ residual outputs) ----
#pragma unroll
            for (int t = 0; t < OC; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const OEntry e = otab[(pass * OC + t) * 32 + rho(r) + 4 * half];
                    if (e.out_feat < 0 || !valid) continue;
                    float v = y[t][r] * e.scale + e.center;
                    if (v < e.lo) v = e.lo;
                    if (v >= e.hi) v = e.hi;
                    v = v * e.mask;
                    const int slot = e.out_feat >> 20, feat = e.out_feat & 0xFFFFF;
                    const int64_t off = (int64_t)feat * out_fs[slot] + n * out_ss[slot];
                    if (p.out64)
                        ((GF64Ptr)out_base[slot])[off] = (double)v;
                    else
                        ((GF32Ptr)out_base[slot])[off] = v;
                    if (e.res >= 0) {
                        const int rslot = e.res >> 8, rs = e.res & 0xFF;
                        GRawPtr sb = (GRawPtr)src_base[rs];
                        const float before = (float)sb[(int64_t)feat * src_fs[rs] + n * src_ss[rs]];
                        const float after = before + v;
                        const int64_t roff = (int64_t)feat * out_fs[rslot] + n * out_ss[rslot];
                        if (p.out64)
                            ((GF64Ptr)out_base[rslot])[roff] = (double)after;
                        else
                            ((GF32Ptr)out_base[rslot])[roff] = after;
                    }
                }
            }
        }
    }
}

template <int HT, int OC>
constexpr size_t wbuf_bytes()
{
    constexpr int HG = (HT + 3) / 4, OG = (OC + 3) / 4;
    constexpr int KC_O = (OG <= 2) ? 16 : 8;
    constexpr int CH_H = 16 * HG * 64, CH_O = KC_O * OG * 64;
    return 2 * (size_t)((CH_H > CH_O) ? CH_H : CH_O) * 16;
}

}  // namespace
}  // namespace fv3hip

using namespace fv3hip;

// ---------------------------------------------------------------------------------------------
// Model object
// ---------------------------------------------------------------------------------------------
struct fv3hip_mlp {
    int device = 0;
    int HT = 0, OC = 0;
    int n_sources = 0, n_inputs = 0, K = 0, width = 0, n_hidden = 0, n_outputs = 0, F = 0, n_residual = 0;
    int n_chunks1 = 0, n_pass = 0, n_ktab = 0, n_otab = 0, n_bias = 0;
    int64_t flops = 0;
    void *d_w = nullptr, *d_ktab = nullptr, *d_otab = nullptr, *d_bias = nullptr;
    int n_cu = 256;
    size_t lds_bytes = 0;
};

namespace {

struct Variant {
    int HT, OC;
};
const Variant kVariants[] = {{1, 4}, {2, 4}, {4, 4}, {8, 4}, {8, 13}};

template <int HT, int OC>
int launch_variant(const fv3hip_mlp *m, const MlpLaunch &lp, bool src64, int grid, size_t lds, hipStream_t st)
{
    if (src64) {
        auto kern = mlp_fused_kernel<HT, OC, true>;
        FV3HIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, lp);
    } else {
        auto kern = mlp_fused_kernel<HT, OC, false>;
        FV3HIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, lp);
    }
    (void)m;
    return check_launch("mlp_fused_kernel");
}

template <typename T>
int upload(const std::vector<T> &v, void **dptr)
{
    *dptr = nullptr;
    if (v.empty()) return FV3HIP_OK;
    FV3HIP_CHECK_HIP(hipMalloc(dptr, v.size() * sizeof(T)));
    FV3HIP_CHECK_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return FV3HIP_OK;
}

}  // namespace

extern "C" int fv3hip_mlp_create(const fv3hip_mlp_desc_t *d, fv3hip_mlp_t *out)
{
    FV3HIP_REQUIRE(d && out, "null pointer");
    *out = nullptr;
    FV3HIP_REQUIRE(d->n_sources >= 1 && d->n_sources <= kMaxSources, "n_sources must be in [1, %d], got %d", kMaxSources, d->n_sources);
    FV3HIP_REQUIRE(d->n_inputs >= 1, "n_inputs must be >= 1");
    FV3HIP_REQUIRE(d->n_outputs >= 1, "n_outputs must be >= 1");
    FV3HIP_REQUIRE(d->n_residual >= 0 && d->n_outputs + d->n_residual <= kMaxOutputs,
                   "n_outputs + n_residual must be <= %d", kMaxOutputs);
    FV3HIP_REQUIRE(d->width >= 1, "width must be >= 1");
    if (d->n_hidden < 1)
        return fail(FV3HIP_EUNSUPPORTED, "networks without a hidden layer (n_hidden=%d) are not implemented", d->n_hidden);
    if (d->width > 256)
        return fail(FV3HIP_EUNSUPPORTED, "hidden width %d > 256 is not implemented by the fused kernel", d->width);
    FV3HIP_REQUIRE(d->hidden_activation == FV3HIP_ACT_RELU || d->hidden_activation == FV3HIP_ACT_LINEAR, "unknown activation %d", d->hidden_activation);
    if (d->hidden_activation != FV3HIP_ACT_RELU)
        return fail(FV3HIP_EUNSUPPORTED, "only ReLU hidden activations are implemented");

    int K = 0;
    for (int i = 0; i < d->n_inputs; ++i) {
        FV3HIP_REQUIRE(d->in_source[i] >= 0 && d->in_source[i] < d->n_sources, "in_source[%d] out of range", i);
        FV3HIP_REQUIRE(d->in_nfeat[i] >= 1 && d->in_feat_start[i] >= 0, "bad feature range for input %d", i);
        K += d->in_nfeat[i];
    }
    if (K > kMaxK) return fail(FV3HIP_EUNSUPPORTED, "%d network inputs > %d is not implemented", K, kMaxK);
    int F = 0;
    for (int j = 0; j < d->n_outputs; ++j) {
        FV3HIP_REQUIRE(d->out_nfeat[j] >= 1 && d->out_nfeat[j] < (1 << 20), "bad out_nfeat[%d]", j);
        F += d->out_nfeat[j];
    }
    for (int r = 0; r < d->n_residual; ++r) {
        FV3HIP_REQUIRE(d->res_source[r] >= 0 && d->res_source[r] < d->n_sources, "res_source[%d] out of range", r);
        FV3HIP_REQUIRE(d->res_output[r] >= 0 && d->res_output[r] < d->n_outputs, "res_output[%d] out of range", r);
    }

    // pick the kernel variant: smallest hidden tiling that holds `width`, then the output
    // tiling that wastes the fewest padded feature tiles
    const int width = d->width;
    const int nt_out = (F + 31) / 32;
    int HT = 0, OC = 0, best_cost = 1 << 30;
    for (const Variant &v : kVariants) {
        if (v.HT * 32 < width) continue;
        if (HT && v.HT != HT) continue;
        if (!HT) HT = v.HT;
        const int cost = ((nt_out + v.OC - 1) / v.OC) * v.OC;
        if (cost < best_cost || (cost == best_cost && v.OC > OC)) {
            best_cost = cost;
            OC = v.OC;
        }
    }
    FV3HIP_REQUIRE(HT > 0 && OC > 0, "no kernel variant for width %d", width);

    fv3hip_mlp *m = new fv3hip_mlp();
    hipGetDevice(&m->device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m->device) == hipSuccess) m->n_cu = prop.multiProcessorCount;
    m->HT = HT;
    m->OC = OC;
    m->n_sources = d->n_sources;
    m->n_inputs = d->n_inputs;
    m->K = K;
    m->width = width;
    m->n_hidden = d->n_hidden;
    m->n_outputs = d->n_outputs;
    m->F = F;
    m->n_residual = d->n_residual;
    m->n_chunks1 = ((K + 1) / 2 + 15) / 16;
    m->n_pass = (nt_out + OC - 1) / OC;
    m->n_ktab = 2 * 16 * m->n_chunks1;
    m->n_otab = 32 * OC * m->n_pass;
    m->n_bias = d->n_hidden * HT * 32 + m->n_pass * OC * 32;
    m->flops = 2 * ((int64_t)K * width + (int64_t)(d->n_hidden - 1) * width * width + (int64_t)width * F);

    const int HG = (HT + 3) / 4, OG = (OC + 3) / 4;
    const int KC_O = (OG <= 2) ? 16 : 8;
    const int OHALVES = 16 / KC_O;
    const int64_t CH_H = 16 * HG * 64, CH_O = (int64_t)KC_O * OG * 64;  // float4 per chunk
    const int n_hid_chunks = m->n_chunks1 + (d->n_hidden - 1) * HT;
    const int n_out_chunks = m->n_pass * HT * OHALVES;

    // ---- packed weight stream ----
    std::vector<float> w((size_t)(n_hid_chunks * CH_H + n_out_chunks * CH_O) * 4, 0.f);
    auto hid_slot = [&](int g, int s, int j, int lane, int e) -> float & {
        return w[(size_t)((g * CH_H + ((int64_t)(s * HG + j) * 64 + lane)) * 4 + e)];
    };
    // layer 1: k = 2 * kpair + half
    {
        const float *W = d->hidden_kernels[0];
        for (int g = 0; g < m->n_chunks1; ++g)
            for (int s = 0; s < 16; ++s)
                for (int j = 0; j < HG; ++j)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int k = 2 * (g * 16 + s) + (lane >> 5);
                            const int f = 32 * (4 * j + e) + (lane & 31);
                            if (k < K && f < width) hid_slot(g, s, j, lane, e) = W[(size_t)k * width + f];
                        }
    }
    // hidden layers l >= 1: k = 32*kt + rho(s) + 4*half (the accumulator layout of the layer before)
    for (int l = 1; l < d->n_hidden; ++l) {
        const float *W = d->hidden_kernels[l];
        for (int kt = 0; kt < HT; ++kt) {
            const int g = m->n_chunks1 + (l - 1) * HT + kt;
            for (int s = 0; s < 16; ++s)
                for (int j = 0; j < HG; ++j)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int k = 32 * kt + rho(s) + 4 * (lane >> 5);
                            const int f = 32 * (4 * j + e) + (lane & 31);
                            if (k < width && f < width) hid_slot(g, s, j, lane, e) = W[(size_t)k * width + f];
                        }
        }
    }
    // output layer
    {
        const float *W = d->out_kernel;
        const size_t base = (size_t)n_hid_chunks * CH_H * 4;
        for (int pass = 0; pass < m->n_pass; ++pass)
            for (int kt = 0; kt < HT; ++kt)
                for (int hf = 0; hf < OHALVES; ++hf) {
                    const int go = (pass * HT + kt) * OHALVES + hf;
                    for (int s = 0; s < KC_O; ++s)
                        for (int j = 0; j < OG; ++j)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 4; ++e) {
                                    const int t = 4 * j + e;
                                    if (t >= OC) continue;
                                    const int k = 32 * kt + rho(hf * KC_O + s) + 4 * (lane >> 5);
                                    const int f = 32 * (pass * OC + t) + (lane & 31);
                                    if (k < width && f < F)
                                        w[base + (size_t)((go * CH_O + ((int64_t)(s * OG + j) * 64 + lane)) * 4 + e)] =
                                            W[(size_t)k * F + f];
                                }
                }
    }
    // ---- input table ----
    std::vector<KEntry> ktab(m->n_ktab);
    for (auto &e : ktab) e = KEntry{-1, 0, 0.f, 1.f, 0, 0.f, 0, 0};
    {
        int k = 0;
        for (int i = 0; i < d->n_inputs; ++i)
            for (int f = 0; f < d->in_nfeat[i]; ++f, ++k) {
                KEntry &e = ktab[k];
                e.src = d->in_source[i];
                e.feat = d->in_feat_start[i] + f;
                e.center = d->in_center ? d->in_center[k] : 0.f;
                e.scale = d->in_scale ? d->in_scale[k] : 1.f;
                e.transform = d->in_transform ? d->in_transform[i] : 0;
                e.eps = d->in_eps ? d->in_eps[i] : 0.f;
            }
    }
    // ---- output table ----
    std::vector<OEntry> otab(m->n_otab);
    for (auto &e : otab) e = OEntry{1.f, 0.f, -INFINITY, INFINITY, 1.f, -1, -1, 0};
    {
        int f = 0;
        for (int j = 0; j < d->n_outputs; ++j) {
            int res = -1;
            for (int r = 0; r < d->n_residual; ++r)
                if (d->res_output[r] == j) res = ((d->n_outputs + r) << 8) | d->res_source[r];
            for (int q = 0; q < d->out_nfeat[j]; ++q, ++f) {
                OEntry &e = otab[f];
                e.scale = d->out_scale ? d->out_scale[f] : 1.f;
                e.center = d->out_center ? d->out_center[f] : 0.f;
                e.lo = d->out_min ? d->out_min[f] : -INFINITY;
                e.hi = d->out_max ? d->out_max[f] : INFINITY;
                e.mask = d->out_mask ? d->out_mask[f] : 1.f;
                e.out_feat = (j << 20) | q;
                e.res = res;
            }
        }
    }
    // ---- biases: [layer][tile][reg][half] ----
    std::vector<float> bias(m->n_bias, 0.f);
    for (int l = 0; l < d->n_hidden; ++l)
        for (int t = 0; t < HT; ++t)
            for (int r = 0; r < 16; ++r)
                for (int hf = 0; hf < 2; ++hf) {
                    const int f = 32 * t + rho(r) + 4 * hf;
                    if (f < width) bias[(size_t)l * HT * 32 + (t * 16 + r) * 2 + hf] = d->hidden_biases[l][f];
                }
    for (int pass = 0; pass < m->n_pass; ++pass)
        for (int t = 0; t < OC; ++t)
            for (int r = 0; r < 16; ++r)
                for (int hf = 0; hf < 2; ++hf) {
                    const int f = 32 * (pass * OC + t) + rho(r) + 4 * hf;
                    if (f < F) bias[(size_t)d->n_hidden * HT * 32 + (size_t)pass * OC * 32 + (t * 16 + r) * 2 + hf] = d->out_bias[f];
                }

    int rc;
    if ((rc = upload(w, &m->d_w)) || (rc = upload(ktab, &m->d_ktab)) || (rc = upload(otab, &m->d_otab)) ||
        (rc = upload(bias, &m->d_bias))) {
        fv3hip_mlp_destroy(m);
        return rc;
    }
    const size_t wb = 2 * (size_t)((CH_H > CH_O) ? CH_H : CH_O) * 16;
    m->lds_bytes = wb + (size_t)m->n_ktab * sizeof(KEntry) + (size_t)m->n_otab * sizeof(OEntry) +
                   (size_t)((m->n_bias + 3) & ~3) * sizeof(float) + (size_t)(3 * kMaxSources + 3 * kMaxOutputs) * 8;
    if (m->lds_bytes > 160 * 1024) {
        fv3hip_mlp_destroy(m);
        return fail(FV3HIP_EUNSUPPORTED, "model tables need %zu bytes of LDS (> 160 KiB)", m->lds_bytes);
    }
    *out = m;
    return FV3HIP_OK;
}

extern "C" int fv3hip_mlp_destroy(fv3hip_mlp_t m)
{
    if (!m) return FV3HIP_OK;
    if (m->d_w) hipFree(m->d_w);
    if (m->d_ktab) hipFree(m->d_ktab);
    if (m->d_otab) hipFree(m->d_otab);
    if (m->d_bias) hipFree(m->d_bias);
    delete m;
    return FV3HIP_OK;
}

extern "C" int64_t fv3hip_mlp_flops_per_sample(fv3hip_mlp_t m) { return m ? m->flops : 0; }

extern "C" int fv3hip_mlp_predict(fv3hip_mlp_t m, const void *const *sources, const int *src_dtype,
                                  const int64_t *src_feat_stride, const int64_t *src_sample_stride,
                                  int64_t n_samples, void *const *outputs, int out_dtype,
                                  const int64_t *out_feat_stride, const int64_t *out_sample_stride,
                                  void *stream)
{
    FV3HIP_REQUIRE(m, "null model handle");
    FV3HIP_REQUIRE(n_samples >= 0, "negative n_samples");
    if (n_samples == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(sources && src_dtype && src_feat_stride && src_sample_stride && outputs &&
                       out_feat_stride && out_sample_stride, "null pointer");
    FV3HIP_REQUIRE(out_dtype == FV3HIP_F32 || out_dtype == FV3HIP_F64, "out_dtype must be F32 or F64");
    MlpLaunch lp;
    memset(&lp, 0, sizeof(lp));
    const int dt0 = src_dtype[0];
    FV3HIP_REQUIRE(dt0 == FV3HIP_F32 || dt0 == FV3HIP_F64, "source dtype must be F32 or F64");
    for (int i = 0; i < m->n_sources; ++i) {
        FV3HIP_REQUIRE(sources[i], "source %d is null", i);
        if (src_dtype[i] != dt0)
            return fail(FV3HIP_EUNSUPPORTED, "all sources must share one dtype (source 0 is %d, source %d is %d)", dt0, i, src_dtype[i]);
        lp.src[i] = sources[i];
        lp.src_fs[i] = src_feat_stride[i];
        lp.src_ss[i] = src_sample_stride[i];
    }
    for (int i = m->n_sources; i < kMaxSources; ++i) lp.src[i] = sources[0];
    for (int j = 0; j < m->n_outputs + m->n_residual; ++j) {
        FV3HIP_REQUIRE(outputs[j], "output %d is null", j);
        lp.out[j] = outputs[j];
        lp.out_fs[j] = out_feat_stride[j];
        lp.out_ss[j] = out_sample_stride[j];
    }
    lp.w = static_cast<const f32x4 *>(m->d_w);
    lp.ktab = static_cast<const KEntry *>(m->d_ktab);
    lp.otab = static_cast<const OEntry *>(m->d_otab);
    lp.bias = static_cast<const float *>(m->d_bias);
    lp.n_chunks1 = m->n_chunks1;
    lp.n_hidden = m->n_hidden;
    lp.n_pass = m->n_pass;
    lp.n_ktab = m->n_ktab;
    lp.n_otab = m->n_otab;
    lp.n_bias = m->n_bias;
    lp.out64 = (out_dtype == FV3HIP_F64);
    lp.n_samples = n_samples;
    lp.n_tiles = ceil_div(n_samples, kTileSamples);
    const int grid = (int)(lp.n_tiles < m->n_cu ? lp.n_tiles : m->n_cu);
    const bool src64 = (dt0 == FV3HIP_F64);
    hipStream_t st = as_stream(stream);
#define VARIANT_(H, O) \
    if (m->HT == H && m->OC == O) return launch_variant<H, O>(m, lp, src64, grid, m->lds_bytes, st)
    VARIANT_(1, 4);
    VARIANT_(2, 4);
    VARIANT_(4, 4);
    VARIANT_(8, 4);
    VARIANT_(8, 13);
#undef VARIANT_
    return fail(FV3HIP_EUNSUPPORTED, "no compiled kernel variant for HT=%d OC=%d", m->HT, m->OC);
}
