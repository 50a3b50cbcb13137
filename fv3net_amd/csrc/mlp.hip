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
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <cfloat>
#include <vector>

#include "common.h"

namespace fv3hip {
namespace {

// Scheduling of one k-pair slot: the matrix pipe takes a 32x32x2 f32 MFMA every 64 cycles and the
// wave issues in order, so the slot's other work (LDS reads of the next A fragments, the input
// prefetch / normalisation, the LDS commits) only overlaps if it sits BETWEEN the MFMAs.  Ask the
// scheduler for (1 MFMA, up to Q others) x NT, then fence the slot.
// End-of-chunk barrier: the other waves only need this wave's LDS writes (the committed weights),
// so wait for LDS traffic only; __syncthreads() would also drain vmcnt, i.e. the loads just issued
// for the NEXT chunk.
#define MLP_CHUNK_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Measured on gfx950 (scratch/ubench): beside fp32 32x32x2 MFMAs a VALU instruction is never free --
// a run of n VALU instructions between two MFMAs costs about 8 + 4n cycles of matrix-pipe time --
// while LDS reads/writes, SALU and s_nop are free and a global load costs ~6.  So a slot's VALU
// work goes into ONE run behind its first MFMA, and the memory instructions are spread over the
// other gaps.
#define MLP_SLOT_SCHED(NT, Q)                                                          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
    __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, 64, 0);                        \
    _Pragma("unroll") for (int _i = 1; _i < (NT); ++_i)                                \
    {                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x004 | 0x010 | 0x080, (Q), 0);           \
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
    int64_t row;       // byte address of (feature, sample 0)
    unsigned int ss;   // byte stride between samples (< 4 GiB)
    unsigned int pad;
};

struct XNorm {  // per model: what is done to it (16 bytes)
    float center, rscale, eps;  // rscale = 1 / (std + epsilon), rounded once
    int flags;  // bit 0: log transform, bit 1: a real input (0 = padding -> 0)
};

// per-call output tables (LDS), one entry per network output feature
struct OFast {      // 16 bytes
    int64_t row;    // byte address of (feature, sample 0) in its output array; 0 = padding, no store
    float scale, center;
};
struct OSlow {      // 16 bytes
    float lo, hi, mask;
    unsigned int ss;  // byte stride between samples
};
struct ORes {       // 32 bytes; residual output  after = before + value
    int64_t src_row;  // 0 = none
    int64_t out_row;
    unsigned int src_ss, out_ss, pad0, pad1;
};

struct OEntry {  // one network output feature as the host describes it (32 bytes, global memory)
    float scale, center, lo, hi;
    float mask;
    int out_feat;  // (output slot << 20) | feature inside the slot; -1 for padding
    int res;       // (residual slot << 8) | residual source; -1 for none
    int pad0;
};

struct MlpLaunch {
    unsigned int w_bytes;  // size of the packed weight stream
    const f32x4 *w;     // packed weight stream
    const KEntry *ktab; // [2 * 16 * n_chunks1]
    const OEntry *otab; // [32 * n_otiles]
    const float *bias;  // packed biases
    int n_chunks1;      // layer-1 chunks of 16 k-pairs
    int n_hidden;
    int n_otiles;       // 32-feature output tiles (= output-type chunks per sample tile)
    int n_hout_tiles;   // leading tiles of the output table that are the last hidden layer's activations themselves (0 or HT)
    int smallf_off;     // small-output launches: offset (floats) in the bias block of the [HT*32][4] output weights
    int smallf_n;       // small-output launches: number of network outputs (1..4)
    int64_t sink;       // fast-I/O launches: device row of n_samples values that padded output rows are stored to
    int n_ktab;
    int n_otab;
    int n_bias;
    int out64;
    int n_log_chunks;   // layer-1 chunks [0, n_log_chunks) hold log-transformed inputs (the host orders them first)
    int n_logfast_chunks;  // the leading ones of them that are all-log with every eps >= FLT_MIN
    int epi_fast;      // sources and outputs are sample-contiguous and 16-byte aligned, n_samples % 32 == 0: fast-I/O kernel
    int has_limits;    // any output limit or zero mask
    int n_residual;
    int64_t n_samples;
    int64_t n_tiles;
    unsigned long long *stamps;  // diagnostic builds only (-DMLP_STAMPS): [wave][8] cycle sums
    const void *src[kMaxSources];
    int64_t src_fs[kMaxSources];
    int64_t src_ss[kMaxSources];
    void *out[kMaxOutputs];
    int64_t out_fs[kMaxOutputs];
    int64_t out_ss[kMaxOutputs];
};

// row of a 32x32 accumulator held by register r of a lane in half h is rho(r) + 4*h
__host__ __device__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

#ifndef MLP_STAMP_PHASE
#define MLP_STAMP_PHASE 0  // per-slot stamps: 0/1/2 layer-1 XBULK chunks by log mode, 10 hidden, 20+NT output
#endif
#ifdef MLP_STAMPS
#define SLOT_STAMP(phase, s)                                          \
    if ((phase) == MLP_STAMP_PHASE) {                                 \
        const unsigned long long t_ = __builtin_readcyclecounter();   \
        if ((s) > 0) sl_acc[(s) - 1] += t_ - sl_t;                    \
        sl_t = t_;                                                    \
    }
#define CHUNK_STAMP_END(phase, KC)                                    \
    if ((phase) == MLP_STAMP_PHASE) {                                 \
        const unsigned long long t_ = __builtin_readcyclecounter();   \
        sl_acc[(KC) - 1] += t_ - sl_t;                                \
        sl_acc[16] += 1;                                              \
    }
#else
#define SLOT_STAMP(phase, s) ((void)0)
#define CHUNK_STAMP_END(phase, KC) ((void)0)
#endif
// XBULK ("fast I/O" kernels): sources AND outputs are sample-contiguous and 16-byte aligned and
// n_samples is a multiple of 32.  Outputs then leave through the row-wise LDS-transposed epilogue
// (the general per-value epilogue is not compiled into these kernels), and the inputs of a layer-1 chunk
// (32 features x 128 samples) are brought in by the whole workgroup with 16-byte loads, normalised
// four at a time and parked in LDS; a k-pair slot then needs one ds_read for its B operand
// instead of a table lookup, an address computation, a 4-byte load and the normalisation.
template <int HT, bool SRC64, bool XBULK, bool HOUT, bool SMALLF, bool L1SHORT>
__global__ __launch_bounds__(kThreads, 1) void mlp_fused_kernel(const MlpLaunch p)
{
    constexpr int HG = (HT + 3) / 4;          // float4 groups of hidden-feature tiles
    constexpr int KC_H = 16;                  // slots (k-pairs) per hidden-type chunk
    // The output layer runs TILE-MAJOR: one 32-feature output tile at a time over the whole
    // contraction, a single 16-register accumulator (a dependent chain of fp32 32x32x2 MFMAs issues
    // at the full rate -- measured).  An output-type chunk is the weights of one output tile; a slot
    // is GS consecutive k-pairs of it.  The tile's epilogue then runs as side work of the NEXT tile's
    // MFMAs, and the output layer needs 32 accumulator registers instead of 16 per feature tile.
    constexpr int GS = (HT == 1) ? 4 : 8;     // k-pairs (MFMAs) per output slot
    constexpr int NGO = GS / 4;               // float4 A fragments per output slot
    constexpr int KC_O = HT * 16 / GS;        // slots per output-type chunk
    constexpr int CH_H = KC_H * HG * 64;      // float4 per hidden-type chunk
    constexpr int CH_O = KC_O * NGO * 64;     // float4 per output-type chunk
    constexpr int NV_H = CH_H / kThreads;
    constexpr int NV_O = CH_O / kThreads;
    constexpr int NV_MAX = (NV_H > NV_O) ? NV_H : NV_O;
    constexpr int CH_MAX = (CH_H > CH_O) ? CH_H : CH_O;
    // the epilogue of output tile t rides under the MFMAs of tile t+1 (needs enough slots per chunk)
    constexpr bool EPI_SIDE = XBULK && KC_O >= 12;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *wbuf = reinterpret_cast<f32x4 *>(smem);                       // [2][CH_MAX]
    float *xs = reinterpret_cast<float *>(wbuf + 2 * CH_MAX);            // [2][32][128] normalised inputs (XBULK)
    XAddr *xa_tab = reinterpret_cast<XAddr *>(xs + (XBULK ? 2 * 32 * kTileSamples : 0));  // [n_ktab] where input k lives
    XNorm *xn_tab = reinterpret_cast<XNorm *>(xa_tab + p.n_ktab);        // [n_ktab] how it is normalised
    OFast *ofast = reinterpret_cast<OFast *>(xn_tab + p.n_ktab);         // [n_otab]
    OSlow *oslow = reinterpret_cast<OSlow *>(ofast + p.n_otab);          // [n_otab]
    ORes *ores = reinterpret_cast<ORes *>(oslow + p.n_otab);             // [n_otab] if the model has residual outputs
    float *biasl = reinterpret_cast<float *>(ores + (p.n_residual ? p.n_otab : 0));  // [n_bias]

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
            // padding reads a finite constant (the first bias) and is multiplied by rscale = 0
            a.row = e.src < 0 ? reinterpret_cast<int64_t>(p.bias)
                              : ksrc[sidx] + (int64_t)e.feat * ksrc[kMaxSources + sidx] * esz;
            a.ss = e.src < 0 ? 0u : (unsigned int)(ksrc[2 * kMaxSources + sidx] * esz);
            a.pad = 0;
            xa_tab[i] = a;
            XNorm nrm;
            nrm.center = e.src < 0 ? 0.f : e.center;
            nrm.rscale = e.src < 0 ? 0.f : e.scale;  // the host stores the reciprocal in KEntry.scale
            nrm.eps = e.eps;
            nrm.flags = (e.transform == FV3HIP_TRANSFORM_LOG ? 1 : 0) | (e.src < 0 ? 0 : 2);
            xn_tab[i] = nrm;
        }
        // Per-feature output tables: absolute row addresses for this call, so that the epilogue is
        // one LDS read, a multiply-add, one address computation and a store per value.
        KargPtr64 kout = (KargPtr64)((KargBytes64)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(MlpLaunch, out));
        const int64_t osz = p.out64 ? 8 : 4;
        for (int i = tid; i < p.n_otab; i += kThreads) {
            const OEntry e = p.otab[i];
            OFast f;
            OSlow sl;
            f.scale = e.scale;
            f.center = e.center;
            sl.lo = e.lo;
            sl.hi = e.hi;
            sl.mask = e.mask;
            sl.ss = 0;
            f.row = p.sink;  // (fast-I/O launches: rows without an output are stored to a scratch row; else 0)
            int slot = 0, feat = 0;
            if (e.out_feat >= 0) {
                slot = e.out_feat >> 20;
                feat = e.out_feat & 0xFFFFF;
                f.row = kout[slot] + (int64_t)feat * kout[kMaxOutputs + slot] * osz;
                sl.ss = (unsigned int)(kout[2 * kMaxOutputs + slot] * osz);
            }
            ofast[i] = f;
            oslow[i] = sl;
            if (p.n_residual) {
                ORes r;
                // (fast-I/O launches: an output row without a residual reads and writes the scratch row, so that
                // the branch-free epilogue treats every row alike; general launches: 0 = none)
                r.src_row = p.sink;
                r.out_row = p.sink;
                r.src_ss = r.out_ss = r.pad0 = r.pad1 = 0;
                if (e.out_feat >= 0 && e.res >= 0) {
                    const int rslot = e.res >> 8, rs = e.res & 0xFF;
                    const int64_t esz = SRC64 ? 8 : 4;
                    r.src_row = ksrc[rs] + (int64_t)feat * ksrc[kMaxSources + rs] * esz;
                    r.src_ss = (unsigned int)(ksrc[2 * kMaxSources + rs] * esz);
                    r.out_row = kout[rslot] + (int64_t)feat * kout[kMaxOutputs + rslot] * osz;
                    r.out_ss = (unsigned int)(kout[2 * kMaxOutputs + rslot] * osz);
                }
                ores[i] = r;
            }
        }
        for (int i = tid; i < p.n_bias; i += kThreads) biasl[i] = p.bias[i];
    }
    __syncthreads();

    int par = 0;  // LDS buffer holding the chunk about to be consumed
    // ---- weight-stream helpers ----
    const int n_hid_chunks = p.n_chunks1 + (p.n_hidden - 1) * HT;  // hidden-type chunks per tile
    const int n_out_chunks = p.n_otiles;                           // output-type chunks per tile (one per 32 outputs)
    const int G = n_hid_chunks + n_out_chunks;
    // The next chunk is staged through registers in two halves (first half requested at the start
    // of a chunk and committed to the other LDS buffer in its second quarter, second half requested
    // at mid-chunk and committed in the last quarter): half the staging registers.
    constexpr int NVH = (NV_MAX + 1) / 2;
    f32x4 stage[NVH];
    // Buffer loads: the stream's base sits in an SGPR resource descriptor, the chunk offset in an
    // SGPR and the per-thread offset in one loop-invariant VGPR -- no per-load VALU address math.
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(p.w), 0, p.w_bytes, 0x00020000);
    const unsigned int w_voff = (unsigned int)tid * 16u;
    auto chunk_off = [&](int g) -> int {  // float4 index of chunk g in the stream
        return (g < n_hid_chunks) ? g * CH_H : n_hid_chunks * CH_H + (g - n_hid_chunks) * CH_O;
    };
    // (for a chunk type with fewer than 2*NVH float4s per thread the second half reads on into the
    // stream -- the host pads it by one maximal chunk -- and lands in LDS words nobody reads)
    auto issue_w = [&](int g, int part) {  // global -> registers
        const int base = (chunk_off(g) + part * NVH * kThreads) * 16;
#pragma unroll
        for (int i = 0; i < NVH; ++i)
            stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff, base + i * kThreads * 16, 0));
    };
    auto commit_w1 = [&](int buf, int part, int i) {  // one staged float4 -> the other LDS buffer
        if (i < NVH) wbuf[buf * CH_MAX + tid + (part * NVH + i) * kThreads] = stage[i];
    };
    // slot s of a chunk of KC slots: its share of staging chunk `gnext` into buffer `buf`.  The last
    // slot stages nothing: it opens with the chunk barrier (see run_slot).
    auto stage_step = [&](int s, int KC, int gnext, int buf) {
        const int Q = KC / 4, per = (NVH + Q - 1) / Q;
        if (s == 0) issue_w(gnext, 0);
        if (s >= Q && s < 2 * Q) {
#pragma unroll
            for (int i = 0; i < per; ++i) commit_w1(buf, 0, (s - Q) * per + i);
        }
        if (s == 2 * Q - 1) issue_w(gnext, 1);  // (after this slot's commit, which frees the registers)
        if (s >= 3 * Q - 1 && s < 4 * Q - 1) {
#pragma unroll
            for (int i = 0; i < per; ++i) commit_w1(buf, 1, (s - (3 * Q - 1)) * per + i);
        }
    };
    // One k-pair slot: NT MFMAs on the A fragments in a_cur, with the A fragments of the next slot
    // read from LDS meanwhile.  The chunk barrier sits at the START of the chunk's last slot, not
    // after it: by then every wave has committed its share of the next chunk (stage_step) and issued
    // its last read of this one, so after the barrier the first fragments of the NEXT chunk are
    // read from the other buffer under the last slot's MFMAs -- no LDS latency is exposed at a
    // chunk boundary.  (__syncthreads() would also drain vmcnt; only LDS traffic matters here.)
    constexpr int AG = (HG > NGO) ? HG : NGO;
    f32x4 a_cur[AG];
    auto run_slot = [&](auto &acc, auto nt_c, auto ng_c, auto q_c, int s, int KC, float b, auto &&side) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_c)::value, NG = decltype(ng_c)::value, Q = decltype(q_c)::value;
        f32x4 a_nxt[AG];
        if (s + 1 < KC) {
            const f32x4 *lw = wbuf + par * CH_MAX + lane;
#pragma unroll
            for (int j = 0; j < NG; ++j) a_nxt[j] = lw[((s + 1) * NG + j) * 64];
#pragma unroll
            for (int j = NG; j < AG; ++j) a_nxt[j] = a_cur[j];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t / 4][t % 4], b, acc[t], 0, 0, 0);
            side(s);
#pragma unroll
            for (int j = 0; j < AG; ++j) a_cur[j] = a_nxt[j];
            MLP_SLOT_SCHED(NT, Q);
        } else {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0][0], b, acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            MLP_CHUNK_BARRIER();
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 *lwn = wbuf + (par ^ 1) * CH_MAX + lane;
#pragma unroll
            for (int j = 0; j < AG; ++j) a_nxt[j] = lwn[j * 64];
#pragma unroll
            for (int t = 1; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t / 4][t % 4], b, acc[t], 0, 0, 0);
            side(s);
#pragma unroll
            for (int j = 0; j < AG; ++j) a_cur[j] = a_nxt[j];
            MLP_SLOT_SCHED(NT - 1, Q);
        }
    };

    // ---- layer-1 B operand helpers ----
    using Raw = typename std::conditional<SRC64, double, float>::type;
    typedef const Raw __attribute__((address_space(1))) *GRawPtr;
    Raw xraw[KC_H];
    float xcur[KC_H];
    // one element (k-pair s of a layer-1 chunk): raw load / transform + normalise, both branch-free
    // so that they can be scheduled between MFMAs
    auto issue_x1 = [&](const XAddr a, int s, int64_t nc) {
        // one v_mad_u64_u32: samples and strides are below 2^32
        xraw[s] = *(GRawPtr)(a.row + (int64_t)((uint64_t)(unsigned int)nc * (uint64_t)a.ss));
    };
    // padding entries need no special case: their row points at a finite constant, center = rscale = 0
    auto finish_x1 = [&](const XNorm e, int s) {  // generic: log transform where flagged
        const float raw = (float)xraw[s];
        const float lg = logf(raw < e.eps ? e.eps : raw);
        const float v = (e.flags & 1) ? lg : raw;
        // (x - mean) * (1 / (std + eps)) instead of the reference's division: <= 1 ulp apart
        xcur[s] = v - e.center;  // (1 / std lives in the layer-1 weights)
    };
    auto finish_x1_plain = [&](const XNorm e, int s) { xcur[s] = (float)xraw[s] - e.center; };
    auto issue_x = [&](int c, int64_t nc) {  // a whole chunk at once (tile boundaries only)
#pragma unroll
        for (int s = 0; s < KC_H; ++s) issue_x1(xa_tab[2 * (c * KC_H + s) + half], s, nc);
    };
    auto finish_x = [&](int c) {
#pragma unroll
        for (int s = 0; s < KC_H; ++s) finish_x1(xn_tab[2 * (c * KC_H + s) + half], s);
    };
#ifdef MLP_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t0 = 0;
    unsigned long long sl_acc[17] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, sl_t = 0;
#define STAMP_BEGIN() st_t0 = __builtin_readcyclecounter()
#define STAMP_END(i)                                         \
    {                                                        \
        const unsigned long long _t = __builtin_readcyclecounter(); \
        st_acc[i] += _t - st_t0;                             \
        st_t0 = _t;                                          \
    }
#else
#define STAMP_BEGIN() ((void)0)
#define STAMP_END(i) ((void)0)
#endif
    // ---- XBULK: cooperative staging of a layer-1 chunk's inputs ----
    typedef double d64x2 __attribute__((ext_vector_type(2)));
    const int sg = tid & 31;  // sample group: samples 4*sg .. 4*sg+3 of the workgroup's 128
    const int fr = tid >> 5;  // feature rows fr, fr+8, fr+16, fr+24 of the chunk's 32
    f32x4 braw[XBULK && !SRC64 ? 4 : 1];
    d64x2 brawd[XBULK && SRC64 ? 4 : 1][2];
    auto bulk_sample = [&](int64_t tile_n0) -> unsigned int {
        int64_t nb = tile_n0 + 4 * sg;
        if (nb > p.n_samples - 4) nb = p.n_samples - 4;  // tail: those samples are never stored
        return (unsigned int)nb;
    };
    auto bulk_issue1 = [&](const XAddr a, int i, unsigned int nb) {
        // one v_mad_u64_u32 (a.ss is the sample stride in bytes, 0 for the padding rows)
        const int64_t addr = a.row + (int64_t)((uint64_t)nb * (uint64_t)a.ss);
        if (SRC64) {
            typedef const d64x2 __attribute__((address_space(1))) *GD2;
            brawd[SRC64 ? i : 0][0] = *(GD2)addr;
            brawd[SRC64 ? i : 0][1] = *(GD2)(addr + (a.ss ? 16 : 0));
        } else {
            typedef const f32x4 __attribute__((address_space(1))) *GF4;
#ifdef MLP_NT_X
            braw[SRC64 ? 0 : i] = __builtin_nontemporal_load((GF4)addr);
#else
            braw[SRC64 ? 0 : i] = *(GF4)addr;
#endif
        }
    };
    auto bulk_issue = [&](int c, int64_t tile_n0) {
        const unsigned int nb = bulk_sample(tile_n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) bulk_issue1(xa_tab[c * 32 + fr + 8 * i], i, nb);
    };
    auto bulk_finish_e = [&](const XNorm e, int i, int buf, auto with_log) {
        const int kk = fr + 8 * i;
        f32x4 v;
        if (SRC64) {
            const d64x2 lo = brawd[SRC64 ? i : 0][0], hi = brawd[SRC64 ? i : 0][1];
            v[0] = (float)lo[0]; v[1] = (float)lo[1]; v[2] = (float)hi[0]; v[3] = (float)hi[1];
        } else {
            v = braw[SRC64 ? 0 : i];
        }
        // with_log: 0 = no transform in this chunk; 1 = log where flagged, libm-accurate; 2 = every
        // feature of the chunk is log-transformed with eps >= FLT_MIN, so the argument is a normal
        // number and v_log_f32 (1 ulp in log2) times ln 2 needs no denormal rescaling; the extra
        // rounding of the product keeps it within 2 ulp -- the reference's own float32 log
        // (numpy/TF SIMD kernels) is not correctly rounded either.  Non-finite values pass as in libm.
        constexpr int LOGM = (int)decltype(with_log)::value;
        if (LOGM == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lg = logf(v[q] < e.eps ? e.eps : v[q]);
                v[q] = (e.flags & 1) ? lg : v[q];
            }
        }
        if (LOGM == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = __builtin_amdgcn_logf(v[q] < e.eps ? e.eps : v[q]) * 0.693147180559945f;
        }
        v = v - e.center;  // (1 / std lives in the layer-1 weights)
        *reinterpret_cast<f32x4 *>(xs + (buf * 32 + kk) * kTileSamples + 4 * sg) = v;
    };
    auto bulk_finish1 = [&](int c, int i, int buf, auto with_log) {
        bulk_finish_e(xn_tab[c * 32 + fr + 8 * i], i, buf, with_log);
    };
    int xb = 0;  // xs buffer the current layer-1 chunk reads

    int64_t tile = blockIdx.x;
    if (tile >= p.n_tiles) return;

    // prime the pipeline: chunk 0 of the stream and the first tile's first inputs
    {
        int64_t n = tile * kTileSamples + wave * 32 + (lane & 31);
        if (XBULK) bulk_issue(0, tile * kTileSamples); else issue_x(0, n < p.n_samples ? n : p.n_samples - 1);
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            issue_w(0, part);
#pragma unroll
            for (int i = 0; i < NVH; ++i) commit_w1(0, part, i);
        }
        if (XBULK) {
            // (the same flavour the steady state gives chunk 0 -- see the end of the layer-1 chunk loop -- so that a
            // column's result does not depend on whether its tile is a workgroup's first)
            if (p.n_logfast_chunks > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bulk_finish1(0, i, 0, std::integral_constant<int, 2>{});
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) bulk_finish1(0, i, 0, std::integral_constant<int, 1>{});
            }
        } else {
            finish_x(0);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < AG; ++j) a_cur[j] = wbuf[lane + j * 64];
    }
    float b_cur = XBULK ? xs[half * kTileSamples + wave * 32 + (lane & 31)] : 0.f;  // XBULK: B operand of the next slot

    for (; tile < p.n_tiles; tile += gridDim.x) {
        const int64_t n = tile * kTileSamples + wave * 32 + (lane & 31);
        const bool valid = n < p.n_samples;
        const int64_t nc = valid ? n : p.n_samples - 1;
        const int64_t next_tile = tile + gridDim.x;
        int64_t nn = next_tile * kTileSamples + wave * 32 + (lane & 31);
        if (nn >= p.n_samples) nn = p.n_samples - 1;
        int g = 0;

        f32x16 h[HT];
        STAMP_BEGIN();
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
            // The host orders the input features with the log-transformed ones first, so chunks
            // [0, n_log_chunks) need the transform and the rest finish an element with a subtract
            // and a multiply.  Each kind has its own chunk loop: a flavour branch INSIDE one loop
            // makes the register allocator shuttle all 128 accumulator registers between AGPRs and
            // VGPRs on every chunk (measured: +2000 cycles per chunk).
            using T_ = std::true_type;
            using F_ = std::false_type;
            const int L = p.n_log_chunks, NC = p.n_chunks1;
            // per-sample path: chunk c consumes xcur, finishes its elements 8..15 in the first half of
            // the slots and requests/finishes elements 0..7 of chunk cn in the second half
            using HT_c = std::integral_constant<int, HT>;
            using HG_c = std::integral_constant<int, HG>;
            using Q5_c = std::integral_constant<int, 5>;
            auto l1_chunk = [&](int c, int cn, auto with_log) __attribute__((always_inline)) {
                const int gnext = (g + 1 < G) ? g + 1 : 0;
                // table entries are read one slot ahead of their use
                const XAddr *xa_n = xa_tab + 2 * cn * KC_H + half;    // issue: element s of chunk cn
                const XNorm *xn_c = xn_tab + 2 * c * KC_H + half;     // finish, first half: chunk c, s+8
                const XNorm *xn_n = xn_tab + 2 * cn * KC_H + half;    // finish, second half: chunk cn, s-8
                XAddr xa_cur = xa_n[0], xa_nxt = xa_cur;
                XNorm xn_cur = xn_c[2 * (KC_H / 2)], xn_nxt = xn_cur;
#pragma unroll
                for (int s = 0; s < KC_H; ++s) {
                    // No branches in a slot (the scheduler only interleaves inside a basic block): on
                    // the last chunk the "next chunk" is the chunk itself, which re-derives values
                    // that are already there; re-finishing elements 8..15 of chunk 0 is idempotent.
                    run_slot(h, HT_c{}, HG_c{}, Q5_c{}, s, KC_H, xcur[s], [&](int s_) {
                        if (s_ + 1 < KC_H) {
                            xa_nxt = xa_n[2 * (s_ + 1)];
                            xn_nxt = (s_ + 1 < KC_H / 2) ? xn_c[2 * (s_ + 1 + KC_H / 2)] : xn_n[2 * (s_ + 1 - KC_H / 2)];
                        }
                        issue_x1(xa_cur, s_, nc);
                        const int el = (s_ < KC_H / 2) ? s_ + KC_H / 2 : s_ - KC_H / 2;
                        if (decltype(with_log)::value) finish_x1(xn_cur, el); else finish_x1_plain(xn_cur, el);
                        stage_step(s_, KC_H, gnext, par ^ 1);
                        xa_cur = xa_nxt;
                        xn_cur = xn_nxt;
                    });
                }
                par ^= 1;
                ++g;
            };
            // XBULK path: the B operand comes from the LDS input tile xs[xb]; the tile of chunk cn of
            // the workgroup tile starting at sample n0 is requested in slot 1 (4 x 16-byte loads per
            // thread) and normalised and written to xs[xb ^ 1] in slots 9..12.
            // (kc1_c: slots of the chunk -- 16, or 8 in the L1SHORT kernels, whose single layer-1 chunk holds at most 16
            // inputs: the k-pairs beyond them would multiply zeros; the input tile is then finished in slots 4..7)
            auto l1_chunk_bulk = [&](int cn, int64_t n0, auto with_log, auto kc1_c) __attribute__((always_inline)) {
                constexpr int KC1 = decltype(kc1_c)::value, FIN0 = (KC1 == 16) ? 8 : 3;
                const int gnext = (g + 1 < G) ? g + 1 : 0;
                const float *xsb = xs + xb * 32 * kTileSamples + half * kTileSamples + wave * 32 + (lane & 31);
                const float *xsn = xs + (xb ^ 1) * 32 * kTileSamples + half * kTileSamples + wave * 32 + (lane & 31);
                const unsigned int nb = bulk_sample(n0);
                XAddr xa4[4];
                XNorm xn4[4];
                // (opaque bases: the four entries are then read at immediate offsets from one address)
                typedef const XAddr __attribute__((address_space(3))) *LXAddr;
                typedef const XNorm __attribute__((address_space(3))) *LXNorm;
                LXAddr xa_c = (LXAddr)(xa_tab + cn * 32 + fr);
                LXNorm xn_c = (LXNorm)(xn_tab + cn * 32 + fr);
                asm volatile("" : "+v"(xa_c), "+v"(xn_c));
                const XAddr *xa_g = (const XAddr *)xa_c;
                const XNorm *xn_g = (const XNorm *)xn_c;
#pragma unroll
                for (int s = 0; s < KC1; ++s) {
                    SLOT_STAMP((int)decltype(with_log)::value, s);
                    run_slot(h, HT_c{}, HG_c{}, Q5_c{}, s, KC1, b_cur, [&](int s_) {
                        // (the last slot runs this after the chunk barrier: xs[xb ^ 1] is complete)
                        b_cur = (s_ + 1 < KC1) ? xsb[2 * (s_ + 1) * kTileSamples] : xsn[0];
                        // table entries are read a slot before they are used
                        if (s_ == 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) xa4[i] = xa_g[8 * i];
                        }
                        if (s_ == 1) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) bulk_issue1(xa4[i], i, nb);
                        }
                        if (s_ == FIN0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) xn4[i] = xn_g[8 * i];
                        }
                        if (s_ > FIN0 && s_ <= FIN0 + 4) bulk_finish_e(xn4[s_ - FIN0 - 1], s_ - FIN0 - 1, xb ^ 1, with_log);
                        stage_step(s_, KC1, gnext, par ^ 1);
                    });
                }
                CHUNK_STAMP_END((int)decltype(with_log)::value, KC1);
                par ^= 1;
                xb ^= 1;
                ++g;
            };
            if (XBULK) {
                const int64_t n0 = tile * kTileSamples;
                using M0_ = std::integral_constant<int, 0>;
                using M1_ = std::integral_constant<int, 1>;
                using M2_ = std::integral_constant<int, 2>;
                const int NF = p.n_logfast_chunks;  // chunks [0, NF) take the fast log (NF <= L)
                using KC16_ = std::integral_constant<int, 16>;
                // the last chunk brings in chunk 0 of the workgroup's next tile
                const int64_t n1 = (next_tile < p.n_tiles ? next_tile : tile) * kTileSamples;
                if constexpr (L1SHORT) {  // one chunk of at most 16 plain inputs (the host checks)
                    l1_chunk_bulk(0, n1, M0_{}, std::integral_constant<int, 8>{});
                } else {
                    int c = 0;  // chunk c prepares the inputs of chunk c + 1
                    for (; c + 1 < NF; ++c) l1_chunk_bulk(c + 1, n0, M2_{}, KC16_{});
                    for (; c + 1 < L; ++c) l1_chunk_bulk(c + 1, n0, M1_{}, KC16_{});
                    for (; c < NC - 1; ++c) l1_chunk_bulk(c + 1, n0, M0_{}, KC16_{});
                    if (NF > 0) l1_chunk_bulk(0, n1, M2_{}, KC16_{});
                    else if (L > 0) l1_chunk_bulk(0, n1, M1_{}, KC16_{});
                    else l1_chunk_bulk(0, n1, M0_{}, KC16_{});
                }
            } else {
                // a chunk that mixes both kinds (the boundary chunk, or the re-derivation on the last
                // chunk) takes the with-log flavour, which selects per feature
                int c = 0;
                for (; c < L && c < NC - 1; ++c) l1_chunk(c, c + 1, T_{});
                for (; c < NC - 1; ++c) l1_chunk(c, c + 1, F_{});
                if (L > 0) l1_chunk(NC - 1, NC - 1, T_{}); else l1_chunk(NC - 1, NC - 1, F_{});
            }
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = (h[t][r] < 0.f) ? 0.f : h[t][r];
        }
        STAMP_END(0);
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
#pragma unroll
                for (int s = 0; s < KC_H; ++s) {
                    SLOT_STAMP(10, s);
                    run_slot(h2, std::integral_constant<int, HT>{}, std::integral_constant<int, HG>{},
                             std::integral_constant<int, 2>{}, s, KC_H, h[kt][s],
                             [&](int s_) { stage_step(s_, KC_H, gnext, par ^ 1); });
                }
                CHUNK_STAMP_END(10, KC_H);
                par ^= 1;
                ++g;
            }
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = (h2[t][r] < 0.f) ? 0.f : h2[t][r];
        }
        STAMP_END(1);
        // ================= hidden -> outputs, one 32-feature tile per chunk =================
        {
            typedef f32x4 __attribute__((address_space(1))) *GF32x4;
            typedef const f32x4 __attribute__((address_space(1))) *GCF32x4;
            typedef d64x2 __attribute__((address_space(1))) *GF64x2;
            typedef const d64x2 __attribute__((address_space(1))) *GCF64x2;
            const int64_t n0t = tile * kTileSamples + wave * 32;
            const int e_wrow = lane >> 3, e_wcol = (lane & 7) * 4;
            constexpr int NHO = HOUT ? HT : 0;  // leading table tiles that are the hidden activations themselves
            // ---- fast-I/O epilogue pieces.  (n_samples is a multiple of 32: a wave's 32 samples are all
            // there or all beyond the end.)  An accumulator tile goes through the wave's 4 KB slice of
            // the xs half that is idle during the output layer, is read back row-wise and leaves as
            // 4 x dwordx4 stores of 8 full 128-byte rows each, with one table read per row.
            float *scr = xs + (xb ^ 1) * 32 * kTileSamples + wave * 1024;
            auto epi_put_tile = [&](const f32x16 &y) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[(rho(r) + 4 * half) * 32 + (lane & 31)] = y[r];
            };
            auto epi_put4 = [&](int64_t row_addr, const f32x4 v) {
                if (p.out64) {
                    const d64x2 lo = {(double)v[0], (double)v[1]}, hi = {(double)v[2], (double)v[3]};
                    *(GF64x2)(row_addr + (n0t + e_wcol) * 8) = lo;
                    *(GF64x2)(row_addr + (n0t + e_wcol) * 8 + 16) = hi;
                } else {
                    *(GF32x4)(row_addr + (n0t + e_wcol) * 4) = v;
                }
            };
            // (mode_c: 1 = "plain": float32 outputs, no limits / masks, no residual outputs; 2 = plain with
            // residual outputs (float32 sources; the residual rows are handled by epi_side); 0 = everything
            // else -- decided once per tile loop, so that the side work inside the MFMA slots of modes 1 and 2
            // is branch-free: the MFMAs of a slot queue only one deep, every branch in its side work is
            // matrix-pipe idle time)
            auto epi_row = [&](const OFast of, f32x4 v, int idx, auto mode_c) {
                // (v is the physical value: scale and center live in the output weights and bias)
                if (decltype(mode_c)::value != 0) {
                    *(GF32x4)(of.row + (n0t + e_wcol) * 4) = v;  // (padded rows point at the sink row)
                    return;
                }
                if (p.has_limits) {
                    const OSlow os = oslow[idx];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float x = v[q];
                        if (x < os.lo) x = os.lo;
                        if (x >= os.hi) x = os.hi;
                        v[q] = x * os.mask;
                    }
                }
                if (of.row != 0) epi_put4(of.row, v);
                if (p.n_residual) {
                    const ORes e = ores[idx];
                    if (e.src_row != 0) {
                        f32x4 before;
                        if (SRC64) {
                            const d64x2 lo = *(GCF64x2)(e.src_row + (n0t + e_wcol) * 8);
                            const d64x2 hi = *(GCF64x2)(e.src_row + (n0t + e_wcol) * 8 + 16);
                            before[0] = (float)lo[0]; before[1] = (float)lo[1];
                            before[2] = (float)hi[0]; before[3] = (float)hi[1];
                        } else {
                            before = *(GCF32x4)(e.src_row + (n0t + e_wcol) * 4);
                        }
                        epi_put4(e.out_row, before + v);
                    }
                }
            };
            // the whole epilogue of output tile t at once (matrix pipe idle)
            auto epi_fast_now = [&](const f32x16 &y, int t) {
                if (n0t >= p.n_samples) return;
                epi_put_tile(y);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = e_wrow + 8 * j, idx = t * 32 + row;
                    epi_row(ofast[idx], *reinterpret_cast<const f32x4 *>(scr + row * 32 + e_wcol), idx, std::integral_constant<int, 0>{});
                }
            };
            // ... or spread over the slots of the next tile's chunk (EPI_SIDE, KC_O >= 12):
            //   slot 1: accumulator tile -> scratch;  slot 8: rows 0,1 (table entry + transposed row) read;
            //   slot 9: they are finished and stored, rows 2,3 read;  slot 10: those finished and stored.
            // Why slots 8..10: vmcnt retires loads AND stores in issue order, so a store must not be
            // older than a staging load that is waited for soon -- the second half of the weight
            // staging is requested in slot KC/2 - 1 and committed from slot 3KC/4 - 1 on; stores issued
            // after the request are younger than it, and the next chunk's first commit is >= 10 slots
            // (~2 us) away.  All of it ends before the chunk's last slot, whose barrier orders the
            // scratch against whatever the other waves do next.
            OFast e_of[2];
            f32x4 e_v[2];
            // mode 2 (the production Zhao-Carr graph: every difference output also leaves as after = before + difference,
            // transforms.py:54-58): the four rows' `before` row addresses are read from the table in slot 0, the four
            // 16-byte `before` loads issued in slot 1 -- older than the second staging request of slot KC/2 - 1, eight slots
            // (~1.7 us: they come from HBM, like the layer-1 inputs) before their first use -- and the sum leaves with a
            // second store next to the row's own in slots 9, 10.  (Tried and dropped: requesting the last tile's rows in
            // slots 11, 12 of its own chunk -- branch-free, every other chunk reading an all-scratch table tile -- cost
            // 900 cycles per chunk in slot 12.)
            // Rows without a residual (total_precipitation, padding) read and write the scratch row.
            constexpr bool RES_SIDE = EPI_SIDE && !SRC64 && !HOUT && !SMALLF;
            int64_t e_src[RES_SIDE ? 4 : 1], e_out[2];
            f32x4 e_before[RES_SIDE ? 4 : 1];
            const int64_t e_loff = (n0t + e_wcol) * 4;
            // (mode 2) `before` row addresses of tile t in slot s0, the four loads in slot s0 + 1
            auto epi_res_fetch = [&](int t, int s_, int s0) __attribute__((always_inline)) {
                if (s_ == s0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) e_src[RES_SIDE ? j : 0] = ores[t * 32 + e_wrow + 8 * j].src_row;
                }
                if (s_ == s0 + 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) e_before[RES_SIDE ? j : 0] = *(GCF32x4)(e_src[RES_SIDE ? j : 0] + e_loff);
                }
            };
            auto epi_side = [&](const f32x16 &y, int t, int s_, auto mode_c) __attribute__((always_inline)) {
                constexpr int MODE = decltype(mode_c)::value;
                if (MODE == 0 && n0t >= p.n_samples) return;  // (modes 1, 2: only for full tiles)
                if (s_ == 1) epi_put_tile(y);
                if constexpr (MODE == 2 && RES_SIDE) epi_res_fetch(t, s_, 0);
                if (s_ == 9 || s_ == 10) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        epi_row(e_of[i], e_v[i], t * 32 + e_wrow + 8 * (2 * (s_ - 9) + i), mode_c);
                        if constexpr (MODE == 2 && RES_SIDE)
                            *(GF32x4)(e_out[i] + e_loff) = e_before[2 * (s_ - 9) + i] + e_v[i];
                    }
                }
                if (s_ == 8 || s_ == 9) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = e_wrow + 8 * (2 * (s_ - 8) + i);
                        e_of[i] = ofast[t * 32 + row];
                        e_v[i] = *reinterpret_cast<const f32x4 *>(scr + row * 32 + e_wcol);
                        if constexpr (MODE == 2 && RES_SIDE) e_out[i] = ores[t * 32 + row].out_row;
                    }
                }
            };
            // (mode 2) the last tile's epilogue, matrix pipe idle: the `before` loads first, the rows' own stores while they fly
            auto epi_res_last = [&](const f32x16 &y, int t) {
                epi_res_fetch(t, 0, 0);
                epi_res_fetch(t, 1, 0);
                epi_put_tile(y);
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = e_wrow + 8 * j;
                    v[j] = *reinterpret_cast<const f32x4 *>(scr + row * 32 + e_wcol);
                    *(GF32x4)(ofast[t * 32 + row].row + e_loff) = v[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(GF32x4)(ores[t * 32 + e_wrow + 8 * j].out_row + e_loff) = e_before[RES_SIDE ? j : 0] + v[j];
            };
            // (mode 1) the last tile's epilogue without a branch
            auto epi_plain_last = [&](const f32x16 &y, int t) {
                epi_put_tile(y);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = e_wrow + 8 * j;
                    *(GF32x4)(ofast[t * 32 + row].row + e_loff) = *reinterpret_cast<const f32x4 *>(scr + row * 32 + e_wcol);
                }
            };
            // ---- general epilogue (any strides / dtypes): per-value stores straight from the accumulator
            auto epi_general = [&](const f32x16 &y, int t) {
                typedef float __attribute__((address_space(1))) *GF32;
                typedef double __attribute__((address_space(1))) *GF64;
                const unsigned int n32 = (unsigned int)n;
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    OFast of[8];
                    OSlow os[8];
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int idx = t * 32 + rho(gq * 8 + i) + 4 * half;
                        of[i] = ofast[idx];
                        os[i] = oslow[idx];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float x = y[gq * 8 + i];  // (physical value: scale and center live in the weights)
                        if (p.has_limits) {
                            if (x < os[i].lo) x = os[i].lo;
                            if (x >= os[i].hi) x = os[i].hi;
                            x = x * os[i].mask;
                        }
                        v[i] = x;
                        if (valid && of[i].row != 0) {
                            const int64_t addr = of[i].row + (int64_t)((uint64_t)n32 * (uint64_t)os[i].ss);
                            if (p.out64)
                                *(GF64)addr = (double)x;
                            else
                                *(GF32)addr = x;
                        }
                    }
                    if (p.n_residual) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int idx = t * 32 + rho(gq * 8 + i) + 4 * half;
                            const ORes e = ores[idx];
                            if (valid && e.src_row != 0) {
                                const float before = (float)*(GRawPtr)(e.src_row + (int64_t)((uint64_t)n32 * (uint64_t)e.src_ss));
                                const float after = before + v[i];
                                const int64_t addr = e.out_row + (int64_t)((uint64_t)n32 * (uint64_t)e.out_ss);
                                if (p.out64)
                                    *(GF64)addr = (double)after;
                                else
                                    *(GF32)addr = after;
                            }
                        }
                    }
                }
            };

            // one slot of an output chunk: GS MFMAs of the single accumulator (A fragment i of the
            // slot, B = hidden activation of k-pair s*GS+i), next slot's fragments read meanwhile,
            // chunk barrier + next chunk's first fragments in the last slot (as run_slot)
            auto run_slot_out = [&](f32x16 &acc, int s, auto &&side) __attribute__((always_inline)) {
                f32x4 a_nxt[AG];
                if (s + 1 < KC_O) {
                    const f32x4 *lw = wbuf + par * CH_MAX + lane;
#pragma unroll
                    for (int j = 0; j < NGO; ++j) a_nxt[j] = lw[((s + 1) * NGO + j) * 64];
#pragma unroll
                    for (int j = NGO; j < AG; ++j) a_nxt[j] = a_cur[j];
#pragma unroll
                    for (int i = 0; i < GS; ++i)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i / 4][i % 4], h[(s * GS + i) / 16][(s * GS + i) % 16], acc, 0, 0, 0);
                    side(s);
#pragma unroll
                    for (int j = 0; j < AG; ++j) a_cur[j] = a_nxt[j];
                    MLP_SLOT_SCHED(GS, 2);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0][0], h[(s * GS) / 16][(s * GS) % 16], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    MLP_CHUNK_BARRIER();
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 *lwn = wbuf + (par ^ 1) * CH_MAX + lane;
#pragma unroll
                    for (int j = 0; j < AG; ++j) a_nxt[j] = lwn[j * 64];
#pragma unroll
                    for (int i = 1; i < GS; ++i)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i / 4][i % 4], h[(s * GS + i) / 16][(s * GS + i) % 16], acc, 0, 0, 0);
                    side(s);
#pragma unroll
                    for (int j = 0; j < AG; ++j) a_cur[j] = a_nxt[j];
                    MLP_SLOT_SCHED(GS - 1, 2);
                }
            };

            // accumulator init = bias of output tile t (clamped: the look-ahead may run one past the end)
            typedef const float __attribute__((address_space(3))) *LBias;
            auto load_bias = [&](f32x16 &y, int t) {
                const int tc = t < p.n_otiles ? t : p.n_otiles - 1;
                LBias bl = (LBias)(biasl + p.n_hidden * HT * 32 + tc * 32 + half);
                asm volatile("" : "+v"(bl));  // (opaque base: the 16 reads use immediate offsets)
                const float *blg = (const float *)bl;
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = blg[r * 2];
            };
            // one output tile = one chunk.  EPI_SIDE: two accumulators used alternately -- while `y`
            // accumulates tile t, `yo` still holds tile t-1: its epilogue runs in slots 1..10, then
            // (slot 12) it is re-initialised with the bias of tile t+1.
            auto out_chunk = [&](f32x16 &y, f32x16 &yo, int t, auto has_prev_c, auto mode_c) __attribute__((always_inline)) {
                const int gnext = (g + 1 == G) ? 0 : g + 1;
#pragma unroll
                for (int s = 0; s < KC_O; ++s) {
                    SLOT_STAMP(20, s);
                    run_slot_out(y, s, [&](int s_) {
                        stage_step(s_, KC_O, gnext, par ^ 1);
                        if (EPI_SIDE) {
                            if (decltype(has_prev_c)::value) epi_side(yo, NHO + t - 1, s_, mode_c);
                            if (s_ == 12) load_bias(yo, t + 1);
                        }
                    });
                }
                CHUNK_STAMP_END(20, KC_O);
                par ^= 1;
                ++g;
            };
            f32x16 yA, yB;
            const int NT = p.n_otiles;
            // Hidden-output models (the cells of the RNN emulators): the first NHO tiles of the output table are
            // the last hidden layer's activations themselves -- no weights, no MFMAs, the accumulators go
            // through the same epilogue as an output tile (relu is already applied).
            if constexpr (HOUT) {
                const bool prefetch_next = !XBULK && NT == 0 && next_tile < p.n_tiles;
                if (prefetch_next) issue_x(0, nn);
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    if (XBULK) epi_fast_now(h[t], t); else epi_general(h[t], t);
                }
                if (prefetch_next) finish_x(0);
            }
            // Small-output models (at most 4 network outputs: the "dense-local" regressors and classifier, the RNN
            // emulators' output convolutions): a 32-row output tile would spend 128 fp32 32x32x2 MFMAs (8 192 cycles) on
            // 2-4 useful rows.  Instead the contraction runs on v_mfma_f32_4x4x1 (16 blocks of 4 outputs x 4 samples, 8
            // cycles): a lane's B operand is its own hidden activation h[t][r] (feature 32t + rho(r) + 4*half of its
            // sample), its A operand the weight of output (lane & 3) for that feature, and D[v] accumulates output v of
            // the lane's sample over the lane's half of the features; the two halves are added with one cross-lane move.
            // The launch carries no output chunk in its weight stream (NT = 0); float32 outputs without limits only.
            if constexpr (SMALLF) {
                // the lane's 16 weights of a hidden tile are contiguous ([half][output][tile][reg], packed by the host): four
                // 16-byte LDS reads per tile, those of tile t+1 in flight under the 16 MFMAs of tile t
                typedef const f32x4 __attribute__((address_space(3))) *LW4;
                LW4 wl = (LW4)(biasl + p.smallf_off + (half * 4 + (lane & 3)) * HT * 16);
                asm volatile("" : "+v"(wl));  // (opaque base: the reads use immediate offsets)
                const f32x4 *wq = (const f32x4 *)wl;
                f32x4 acc4[4], wcur[4], wnxt[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    wcur[i] = wq[i];
                }
#pragma unroll
                for (int t = 0; t < HT; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) wnxt[i] = wq[((t + 1 < HT) ? t + 1 : t) * 4 + i];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc4[r & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wcur[r >> 2][r & 3], h[t][r], acc4[r & 3], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) wcur[i] = wnxt[i];
                }
                f32x4 y4 = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
#pragma unroll
                for (int v = 0; v < 4; ++v) y4[v] += __shfl_xor(y4[v], 32, 64);
                if (half == 0 && n0t < p.n_samples) {
                    typedef float __attribute__((address_space(1))) *GF32;
                    const float *bo = biasl + p.n_hidden * HT * 32;  // [reg][half] of output tile 0: output f sits at 2 f
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (v < p.smallf_n) *(GF32)(ofast[NHO * 32 + v].row + (n0t + lane) * 4) = y4[v] + bo[2 * v];
                }
            }
            auto tile_loop = [&](auto mode_c) __attribute__((always_inline)) {
                if constexpr (EPI_SIDE) {
                    load_bias(yA, 0);
                    out_chunk(yA, yB, 0, std::false_type{}, mode_c);
                    for (int t = 1; t < NT; t += 2) {
                        out_chunk(yB, yA, t, std::true_type{}, mode_c);
                        if (t + 1 < NT) out_chunk(yA, yB, t + 1, std::true_type{}, mode_c);
                    }
                } else {
                    for (int t = 0; t < NT; ++t) {
                        load_bias(yA, t);
                        out_chunk(yA, yB, t, std::false_type{}, mode_c);
                        // (general kernels: the next tile's first inputs are requested before the last
                        // epilogue and finished after it)
                        const bool prefetch_next = (t + 1 == NT) && next_tile < p.n_tiles;
                        if (!XBULK && prefetch_next) issue_x(0, nn);
                        if (XBULK) epi_fast_now(yA, NHO + t); else epi_general(yA, NHO + t);
                        if (!XBULK && prefetch_next) finish_x(0);
                    }
                }
            };
            bool done = false, res_last = false, plain_last = false;
            if constexpr (!SMALLF) {  // (small-output launches carry no output chunk: NT = 0)
            if (!HOUT || NT > 0) {
                const bool side_plain = EPI_SIDE && !p.has_limits && !p.out64 && (tile + 1) * kTileSamples <= p.n_samples;
                if (side_plain && !p.n_residual) {
                    tile_loop(std::integral_constant<int, 1>{});
                    done = plain_last = true;
                }
                if constexpr (RES_SIDE) {
                    if (!done && side_plain) {
                        tile_loop(std::integral_constant<int, 2>{});
                        done = res_last = true;
                    }
                }
                if (!done) tile_loop(std::integral_constant<int, 0>{});
            }
            STAMP_END(2);
            if (EPI_SIDE && (!HOUT || NT > 0)) {  // the last tile's epilogue (tile NT-1 sits in yA if NT is odd)
                if (RES_SIDE && res_last) {
                    if (NT & 1) epi_res_last(yA, NHO + NT - 1); else epi_res_last(yB, NHO + NT - 1);
                } else if (plain_last) {
                    if (NT & 1) epi_plain_last(yA, NHO + NT - 1); else epi_plain_last(yB, NHO + NT - 1);
                } else {
                    if (NT & 1) epi_fast_now(yA, NHO + NT - 1); else epi_fast_now(yB, NHO + NT - 1);
                }
            }
            }
            // the scratch half of xs is rewritten by the next tile's first layer-1 chunk
            if (XBULK) MLP_CHUNK_BARRIER();
            STAMP_END(3);
        }
    }
#ifdef MLP_STAMPS
    if (p.stamps && lane == 0) {
        unsigned long long *o = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 32;
        for (int i = 0; i < 6; ++i) o[i] = st_acc[i];
        for (int i = 0; i < 17; ++i) o[8 + i] = sl_acc[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Small sample counts: one model rank's columns (2 304 at C48 on 6 ranks) are 18 tiles of the kernel above -- 18 of 256 CUs,
// and a wave spends a whole tile's matrix time (~390 K cycles for the Zhao-Carr graph) on its 32 columns whatever the
// count.  mlp_small_kernel gives 32 samples to a WORKGROUP instead and splits the layers' output features over its
// waves (32-feature tiles dealt round-robin): four times as many CUs work on a call, and a call lasts a quarter of a tile.
// The price is an exchange of every layer's activations through LDS (one barrier per layer) and weights read straight from
// L2 in operand layout (nothing to share between waves, so no LDS staging) -- throughput per CU is no better than the big
// kernel's, so the host picks this kernel only while the big one would leave CUs idle (fv3hip_mlp_predict).
// Same graph, same float32 MFMA arithmetic (v_mfma_f32_32x32x2_f32, bias as the initial accumulator, one rounding per
// folded weight), the contraction in plain feature order; the general per-value epilogue (any strides and dtypes, limits,
// masks, residual and hidden outputs).  Checked against the same float64 oracle at the same per-level tolerance
// (tests/test_gpu_mlp.py::test_small_sample_kernel_*).
// ---------------------------------------------------------------------------------------------
struct SmallLaunch {
    const float *w1;   // [n_ktab][Wp]   layer 1, rows in table order (log inputs first), 1 / std folded in
    const float *wh;   // [n_hidden - 1][Wp][Wp]
    const float *wo;   // [Wp][Fp]       output scale folded in
    const float *bh;   // [n_hidden][Wp]
    const float *bo;   // [Fp]           output scale and centre folded in
    const KEntry *ktab;
    const OEntry *otab;
    int n_ktab, Wp, HT, n_hidden, Fp, n_otiles, n_hout_tiles, out64, has_limits, hout_slot;
    int64_t n_samples;
    unsigned long long *stamps;  // diagnostic builds only (-DSMALL_STAMPS): [workgroup][wave][8] cycle stamps
    const void *src[kMaxSources];
    int64_t src_fs[kMaxSources];
    int64_t src_ss[kMaxSources];
    void *out[kMaxOutputs];
    int64_t out_fs[kMaxOutputs];
    int64_t out_ss[kMaxOutputs];
};

constexpr int kSmallWaves = 8;     // 512 threads: two waves per SIMD hide each other's operand loads
constexpr int kSmallChunk = 128;   // input features staged per step (2 x 16 KB of LDS)

// (uniform index: a scalar read of the kernel arguments) what: 0 base address, 1 feature stride, 2 sample stride, in bytes
__device__ __forceinline__ int64_t otab_base_early(const SmallLaunch &p, int slot, int what)
{
    const int64_t osz = p.out64 ? 8 : 4;
    return what == 0 ? reinterpret_cast<int64_t>(p.out[slot]) : what == 1 ? p.out_fs[slot] * osz : p.out_ss[slot] * osz;
}

#ifdef SMALL_STAMPS
#define SM_STAMP(i) if (lane == 0 && p.stamps) p.stamps[((int64_t)blockIdx.x * kSmallWaves + wave) * 8 + (i)] = __builtin_readcyclecounter()
#else
#define SM_STAMP(i) ((void)0)
#endif

template <bool SRC64>
__global__ __launch_bounds__(kSmallWaves * 64) void mlp_small_kernel(const SmallLaunch p)
{
    using Raw = typename std::conditional<SRC64, double, float>::type;
    constexpr int NW = kSmallWaves, NT = NW * 64, KCH = kSmallChunk, PER = KCH * 32 / NT;
    constexpr int MAXT = (8 + NW - 1) / NW;  // hidden tiles a wave may own (HT <= 8)
    constexpr int U = 8;                     // k-pairs per operand batch
    struct XRow {  // one network input of this call: where its row starts and what is done to it (32 bytes, LDS)
        int64_t row;      // byte address of (feature, sample 0); 0 = padding
        unsigned int ss;  // bytes between samples
        float center, eps;
        int is_log;
        int pad0, pad1;
    };
    extern __shared__ float small_lds[];
    float *xs = small_lds;               // [2][KCH][32]
    float *hA = xs + 2 * KCH * 32;       // [Wp][32]
    float *hB = hA + p.Wp * 32;          // [Wp][32]
    XRow *kt = reinterpret_cast<XRow *>(hB + p.Wp * 32);  // [n_ktab]
    // the call's source / output arrays, so that a per-feature slot number is an LDS lookup (indexing the kernel arguments
    // with a per-lane index costs a dependent memory round trip per value: the first version spent 4 K cycles per chunk
    // just ISSUING its input loads, and 13 K storing the hidden outputs)
    int64_t *otab_base = reinterpret_cast<int64_t *>(kt + p.n_ktab);  // [3][kMaxOutputs]: base, feature stride, sample stride (bytes)
    int64_t *stab_base = otab_base + 3 * kMaxOutputs;                  // [3][kMaxSources]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    const int Wp = p.Wp, HT = p.HT;
    const int n_chunks = (p.n_ktab + KCH - 1) / KCH;
    constexpr int64_t ESZ = SRC64 ? 8 : 4;
    const int64_t osz = p.out64 ? 8 : 4;

    // ---- the weight stream ----
    // A call of this kernel is a chain of dependent phases (table, inputs, layer by layer, epilogue); at 32 samples per
    // workgroup each phase is a few microseconds, so every memory round trip that sits BETWEEN phases shows (the first
    // version paid ~10 of them, 3x its matrix time).  The weights therefore form ONE stream over layer 1 and the hidden
    // layers -- segment 0 = w1 [n_ktab][Wp], segment l = wh[l-1] [Wp][Wp], all read as rows of k-pairs for this wave's
    // tiles -- requested one batch (U k-pairs) ahead of the MFMAs that consume it, ACROSS chunk, barrier and layer
    // boundaries: while a phase ends, the next phase's first operands are already on their way.  Every request is an
    // unconditional raw buffer load (descriptor + loop-invariant lane offset + scalar row offset): no vector address
    // arithmetic, exact counted waits.
    int tcl[MAXT];  // this wave's tiles, clamped (uniform)
#pragma unroll
    for (int ti = 0; ti < MAXT; ++ti) tcl[ti] = (wave_u + ti * NW < HT) ? wave_u + ti * NW : HT - 1;
    const int lane_w = (half * Wp + col) * 4, lane_b = half * 32 + col;
    int a_seg = 0, a_pair = 0;  // the next batch to request (uniform)
    const int last_seg = p.n_hidden - 1;
    auto request_a = [&](float (&a)[MAXT][U]) {
        const float *base = (a_seg == 0) ? p.w1 : p.wh + (int64_t)(a_seg - 1) * Wp * Wp;
        const int seg_pairs = (a_seg == 0) ? p.n_ktab / 2 : Wp / 2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, seg_pairs * 2 * Wp * 4, 0x00020000);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ti = 0; ti < MAXT; ++ti)
                a[ti][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_w, (2 * (a_pair + u) * Wp + 32 * tcl[ti]) * 4, 0));
        a_pair += U;  // (segments are whole numbers of batches: n_ktab / 2 and Wp / 2 are multiples of 2 U)
        if (a_pair >= seg_pairs && a_seg < last_seg) {
            a_pair = 0;
            ++a_seg;
        } else if (a_pair >= seg_pairs) {
            a_pair = seg_pairs - U;  // past the end of the stream: the last batch again (never consumed)
        }
    };
    auto load_b = [&](float (&b)[U], const float *bsrc, int p0, int n_pairs) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pr = (p0 + u < n_pairs) ? p0 + u : n_pairs - 1;
            b[u] = (bsrc + 64 * pr)[lane_b];
        }
    };
    f32x16 acc[MAXT];
    auto mfma_ops = [&](const float (&a)[MAXT][U], const float (&b)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ti = 0; ti < MAXT; ++ti)
                if (wave_u + ti * NW < HT) acc[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][u], b[u], acc[ti], 0, 0, 0);
    };
    float a0[MAXT][U], a1[MAXT][U], b0[U], b1[U];
    // n_pairs (a multiple of 2 U) k-pairs of the stream against B[k][sample] in LDS; on entry a0 holds the first batch
    auto contract = [&](const float *bsrc, int n_pairs) {
        load_b(b0, bsrc, 0, n_pairs);
        for (int p0 = 0; p0 < n_pairs; p0 += 2 * U) {
            request_a(a1);
            load_b(b1, bsrc, p0 + U, n_pairs);
            __builtin_amdgcn_sched_barrier(0);  // (the requests stay in front of this batch's MFMAs)
            mfma_ops(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            request_a(a0);
            load_b(b0, bsrc, p0 + 2 * U, n_pairs);
            __builtin_amdgcn_sched_barrier(0);
            mfma_ops(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    SM_STAMP(0);
    request_a(a0);  // the first weights are on their way before anything else

    for (int i = tid; i < p.n_ktab; i += NT) {
        const KEntry e = p.ktab[i];
        XRow x;
        x.row = e.src < 0 ? 0 : reinterpret_cast<int64_t>(p.src[e.src]) + (int64_t)e.feat * p.src_fs[e.src] * ESZ;
        x.ss = e.src < 0 ? 0u : (unsigned int)(p.src_ss[e.src] * ESZ);
        x.center = e.center;
        x.eps = e.eps;
        x.is_log = (e.transform == FV3HIP_TRANSFORM_LOG) ? 1 : 0;
        x.pad0 = x.pad1 = 0;
        kt[i] = x;
    }
    if (tid < kMaxOutputs) {
        otab_base[tid] = reinterpret_cast<int64_t>(p.out[tid]);
        otab_base[kMaxOutputs + tid] = p.out_fs[tid] * osz;
        otab_base[2 * kMaxOutputs + tid] = p.out_ss[tid] * osz;
    }
    if (tid < kMaxSources) {
        stab_base[tid] = reinterpret_cast<int64_t>(p.src[tid]);
        stab_base[kMaxSources + tid] = p.src_fs[tid] * ESZ;
        stab_base[2 * kMaxSources + tid] = p.src_ss[tid] * ESZ;
    }
    // the output slots of the hidden activations (hidden-output models): read now, used after the last hidden layer
    int hout_feat[MAXT][16];
    if (p.n_hout_tiles) {
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) hout_feat[ti][r] = p.otab[32 * tcl[ti] + rho(r) + 4 * half].out_feat;
    }
    __syncthreads();
    SM_STAMP(1);
    // ---- input staging: raw loads now, transform + centre when the rows are parked in LDS ----
    Raw xr[PER];
    auto issue = [&](int c) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = tid + j * NT, kk = idx >> 5, n = idx & 31, k = c * KCH + kk;
            xr[j] = (Raw)1;
            if (k < p.n_ktab) {
                const XRow e = kt[k];
                int64_t ns = n0 + n;
                if (ns >= p.n_samples) ns = p.n_samples - 1;  // (a ragged last tile reads its last sample again)
                if (e.row != 0) xr[j] = *reinterpret_cast<const Raw *>(e.row + ns * e.ss);
            }
        }
    };
    auto commit = [&](int c, int buf) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = tid + j * NT, kk = idx >> 5, n = idx & 31, k = c * KCH + kk;
            float v = 0.f;
            if (k < p.n_ktab) {
                const XRow e = kt[k];
                if (e.row != 0) {
                    v = (float)xr[j];
                    if (e.is_log) v = logf(v < e.eps ? e.eps : v);
                    v = v - e.center;  // (1 / std lives in the layer-1 weights)
                }
            }
            xs[(buf * KCH + kk) * 32 + n] = v;
        }
    };
    auto init_bias = [&](const float *b) {
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][r] = b[32 * tcl[ti] + rho(r) + 4 * half];
    };
    // ReLU, park the activations for the next layer.  When the model returns its last hidden layer (recurrent cells) the
    // rows leave AFTER the barrier, from LDS, 16 bytes per lane (hidden_rows_out) where the output is float32, sample-contiguous
    // and 16-byte aligned and the tile is whole -- four row pieces per thread instead of sixteen 4-byte stores per lane.
    const int hslot = p.n_hout_tiles ? (p.otab[0].out_feat >> 20) : 0;  // (uniform: every hidden feature goes to one slot)
    bool hout_rows = false;
    if (p.n_hout_tiles) {
        const int64_t hb = otab_base_early(p, hslot, 0), hfs = otab_base_early(p, hslot, 1), hss = otab_base_early(p, hslot, 2);
        hout_rows = !p.out64 && hss == 4 && (hb % 16 == 0) && (hfs % 16 == 0) && (n0 % 4 == 0) && n0 + 32 <= p.n_samples;
    }
    auto finish_hidden = [&](float *dst, bool last) {
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti) {
            const int t = wave_u + ti * NW;
            if (t >= HT) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * t + rho(r) + 4 * half;
                const float h = acc[ti][r] < 0.f ? 0.f : acc[ti][r];  // (a NaN stays a NaN, as in the big kernel and in Keras' relu)
                dst[f * 32 + col] = h;
                if (last && p.n_hout_tiles && !hout_rows) {
                    const int of = hout_feat[ti][r];
                    if (of >= 0 && n0 + col < p.n_samples) {
                        const int slot = of >> 20, q = of & 0xFFFFF;
                        const int64_t addr = otab_base[slot] + q * otab_base[kMaxOutputs + slot] + (n0 + col) * otab_base[2 * kMaxOutputs + slot];
                        if (p.out64) *reinterpret_cast<double *>(addr) = (double)h;
                        else *reinterpret_cast<float *>(addr) = h;
                    }
                }
            }
        }
    };
    auto hidden_rows_out = [&](const float *srcl) {  // after the barrier that follows finish_hidden(last = true)
        if (!(p.n_hout_tiles && hout_rows)) return;
        const int width = 32 * p.n_hout_tiles;
        for (int i = tid; i < width * 8; i += NT) {
            const int f = i >> 3, quad = i & 7;
            if (p.otab[f].out_feat < 0) continue;  // (padding rows of the last tile)
            const f32x4 v = *reinterpret_cast<const f32x4 *>(srcl + f * 32 + 4 * quad);
            *reinterpret_cast<f32x4 *>(otab_base[hslot] + f * otab_base[kMaxOutputs + hslot] + (n0 + 4 * quad) * 4) = v;
        }
    };

    // ---- layer 1 ----
    issue(0);
    init_bias(p.bh);
    commit(0, 0);
    __syncthreads();
    SM_STAMP(2);
    for (int c = 0; c < n_chunks; ++c) {
        if (c + 1 < n_chunks) issue(c + 1);
        const int kn = (p.n_ktab - c * KCH < KCH) ? p.n_ktab - c * KCH : KCH;  // (n_ktab is a multiple of 32)
        contract(xs + (c & 1) * KCH * 32, kn / 2);
        if (c + 1 < n_chunks) commit(c + 1, (c + 1) & 1);
        __syncthreads();
    }
    SM_STAMP(3);
    float *hin = hA, *hout = hB;
    finish_hidden(hin, p.n_hidden == 1);
    __syncthreads();
    if (p.n_hidden == 1) hidden_rows_out(hin);
    SM_STAMP(4);
    // ---- hidden layers ----
    for (int l = 1; l < p.n_hidden; ++l) {
        init_bias(p.bh + (int64_t)l * Wp);
        contract(hin, Wp / 2);
        finish_hidden(hout, l == p.n_hidden - 1);
        __syncthreads();
        if (l == p.n_hidden - 1) hidden_rows_out(hout);
        float *tmp = hin; hin = hout; hout = tmp;
    }
    // ---- output layer: 32-feature tiles dealt over the waves; per-value epilogue ----
    const __amdgpu_buffer_rsrc_t wo_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wo), 0, Wp * p.Fp * 4, 0x00020000);
    const int lane_wo = (half * p.Fp + col) * 4;
    const int n_pairs_o = Wp / 2;
    // One output tile only (recurrent cells' 4 outputs, the dense-local models' 2): the contraction index is split over
    // the waves instead -- every wave a slice of the k-pairs, the partial tiles summed through LDS -- or seven waves would
    // watch one work through the whole contraction (a fifth of a cell launch).
    const bool ksplit = p.n_otiles == 1 && n_pairs_o % (NW * U) == 0;
    const int pairs_lo = ksplit ? wave_u * (n_pairs_o / NW) : 0, pairs_hi = ksplit ? pairs_lo + n_pairs_o / NW : n_pairs_o;
    for (int t = ksplit ? 0 : wave_u; t < p.n_otiles; t += NW) {
        f32x16 y;
        float c0[U], c1[U];
        auto load_o = [&](float (&a)[U], float (&b)[U], int p0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pr = (p0 + u < pairs_hi) ? p0 + u : pairs_hi - 1;
                b[u] = (hin + 64 * pr)[lane_b];
                a[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wo_rsrc, lane_wo, (2 * pr * p.Fp + 32 * t) * 4, 0));
            }
        };
        auto mfma_o = [&](const float (&a)[U], const float (&b)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) y = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], y, 0, 0, 0);
        };
        load_o(c0, b0, pairs_lo);
        // this tile's table entries and biases travel under its MFMAs
        int of_[16], res_[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * t + rho(r) + 4 * half;
            y[r] = (ksplit && wave_u != 0) ? 0.f : p.bo[f];
            of_[r] = p.otab[32 * p.n_hout_tiles + f].out_feat;
            res_[r] = p.otab[32 * p.n_hout_tiles + f].res;
        }
        for (int p0 = pairs_lo; p0 < pairs_hi; p0 += 2 * U) {
            load_o(c1, b1, p0 + U);
            __builtin_amdgcn_sched_barrier(0);
            mfma_o(c0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_o(c0, b0, p0 + 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            if (p0 + U < pairs_hi) mfma_o(c1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ksplit) {  // partial tiles -> LDS (the input staging buffers are free by now), summed by wave 0 in wave order
            float *part = xs + wave_u * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r * 64 + lane] = y[r];
            __syncthreads();
            if (wave_u != 0) break;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                for (int w = 1; w < NW; ++w) y[r] += xs[w * 1024 + r * 64 + lane];
        }
        if (n0 + col >= p.n_samples) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (of_[r] < 0) continue;
            float x = y[r];  // (the physical value: scale and centre live in the output weights and bias)
            if (p.has_limits) {
                const OEntry e = p.otab[32 * p.n_hout_tiles + 32 * t + rho(r) + 4 * half];
                if (x < e.lo) x = e.lo;
                if (x >= e.hi) x = e.hi;
                x = x * e.mask;
            }
            const int slot = of_[r] >> 20, q = of_[r] & 0xFFFFF;
            const int64_t addr = otab_base[slot] + q * otab_base[kMaxOutputs + slot] + (n0 + col) * otab_base[2 * kMaxOutputs + slot];
            if (p.out64) *reinterpret_cast<double *>(addr) = (double)x;
            else *reinterpret_cast<float *>(addr) = x;
            if (res_[r] >= 0) {  // residual output: after = before + value (transforms.py:54-58)
                const int rslot = res_[r] >> 8, rs = res_[r] & 0xFF;
                const float before = (float)*reinterpret_cast<const Raw *>(stab_base[rs] + q * stab_base[kMaxSources + rs] + (n0 + col) * stab_base[2 * kMaxSources + rs]);
                const int64_t raddr = otab_base[rslot] + q * otab_base[kMaxOutputs + rslot] + (n0 + col) * otab_base[2 * kMaxOutputs + rslot];
                if (p.out64) *reinterpret_cast<double *>(raddr) = (double)(before + x);
                else *reinterpret_cast<float *>(raddr) = before + x;
            }
        }
    }
    SM_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// Layered path: networks the two fused kernels do not hold -- no hidden layer at all (the reference's "linear"
// architecture, fv3fit/emulation/layers/architecture.py:285-302 with MLPBlock(depth=0)), hidden layers wider than 256,
// more inputs than the per-feature LDS table takes.  One launch per step, activations [feature][sample] float32 in an
// HBM scratch the model keeps: gather (the same input table: source arrays of any stride and dtype, logarithm, centre),
// one dense kernel per layer (the same float32 MFMA, bias as the initial accumulator, folded scales), scatter (the same
// output table: limits, masks, residual outputs).  Not a fast path -- each layer's activations make a round trip through
// HBM -- it exists so that every network the reference's configs can describe runs; the column counts and widths the
// benchmarks quote stay on the fused kernels.
// ---------------------------------------------------------------------------------------------
struct LayeredIo {
    const KEntry *ktab;
    const OEntry *otab;
    float *x;           // scratch [rows][np]
    int64_t np, n0, n_samples;  // padded slab width; first sample of the slab; samples of the call
    int n_rows, out64, has_limits;
    const void *src[kMaxSources];
    int64_t src_fs[kMaxSources];
    int64_t src_ss[kMaxSources];
    void *out[kMaxOutputs];
    int64_t out_fs[kMaxOutputs];
    int64_t out_ss[kMaxOutputs];
};

// grid (np / 256, n_ktab): row k of the normalised input matrix for the slab's samples (a ragged end repeats the last sample)
template <bool SRC64>
__global__ __launch_bounds__(256) void layered_gather_kernel(const LayeredIo p)
{
    using Raw = typename std::conditional<SRC64, double, float>::type;
    const int k = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const KEntry e = p.ktab[k];
    float v = 0.f;
    if (e.src >= 0) {
        int64_t ns = p.n0 + j;
        if (ns >= p.n_samples) ns = p.n_samples - 1;
        const Raw *row = static_cast<const Raw *>(p.src[e.src]) + (int64_t)e.feat * p.src_fs[e.src];
        v = (float)row[ns * p.src_ss[e.src]];
        if (e.transform == FV3HIP_TRANSFORM_LOG) v = logf(v < e.eps ? e.eps : v);
        v = v - e.center;  // (1 / std lives in the first layer's weights)
    }
    p.x[(int64_t)k * p.np + j] = v;
}

struct LayeredDense {
    const float *w;  // [kp][fp]
    const float *b;  // [fp]
    const float *x;  // [kp][np]
    float *y;        // [fp][np]
    int kp, fp, relu;
    int64_t np;
};

// grid (np / 256, fp / 64), 4 waves: a wave owns 64 features x 64 samples (2 x 2 MFMA tiles); operands straight from
// global memory (rows of 128 B per half wave), requested one batch of k-pairs ahead of the MFMAs that consume them
__global__ __launch_bounds__(256) void layered_dense_kernel(const LayeredDense p)
{
    constexpr int U = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int64_t s0 = (int64_t)blockIdx.x * 256 + wave * 64;
    const int f0 = blockIdx.y * 64;
    const int n_pairs = p.kp / 2;  // a multiple of 2 U (kp is a multiple of 32)
    const float *wl = p.w + (int64_t)half * p.fp + f0 + col;
    const float *xl = p.x + (int64_t)half * p.np + s0 + col;
    f32x16 acc[2][2];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ft][0][r] = acc[ft][1][r] = p.b[f0 + 32 * ft + rho(r) + 4 * half];
    float a0[2][U], b0[2][U], a1[2][U], b1[2][U];
    auto load = [&](float (&a)[2][U], float (&b)[2][U], int p0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pr = (p0 + u < n_pairs) ? p0 + u : n_pairs - 1;
            a[0][u] = wl[(int64_t)2 * pr * p.fp];
            a[1][u] = wl[(int64_t)2 * pr * p.fp + 32];
            b[0][u] = xl[(int64_t)2 * pr * p.np];
            b[1][u] = xl[(int64_t)2 * pr * p.np + 32];
        }
    };
    auto mfma = [&](const float (&a)[2][U], const float (&b)[2][U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int st = 0; st < 2; ++st) acc[ft][st] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ft][u], b[st][u], acc[ft][st], 0, 0, 0);
    };
    load(a0, b0, 0);
    for (int p0 = 0; p0 < n_pairs; p0 += 2 * U) {
        load(a1, b1, p0 + U);
        mfma(a0, b0);
        load(a0, b0, p0 + 2 * U);
        mfma(a1, b1);
    }
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[ft][st][r];
                if (p.relu) v = v < 0.f ? 0.f : v;  // (a NaN stays a NaN)
                p.y[(int64_t)(f0 + 32 * ft + rho(r) + 4 * half) * p.np + s0 + 32 * st + col] = v;
            }
}

// grid (np / 256, F): output feature f of the slab's samples through the output table (mlp_small_kernel's epilogue)
template <bool SRC64>
__global__ __launch_bounds__(256) void layered_scatter_kernel(const LayeredIo p)
{
    using Raw = typename std::conditional<SRC64, double, float>::type;
    const int f = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, ns = p.n0 + j;
    const OEntry e = p.otab[f];
    if (e.out_feat < 0 || ns >= p.n_samples) return;
    float x = p.x[(int64_t)f * p.np + j];  // (the physical value: scale and centre live in the output weights and bias)
    if (p.has_limits) {
        if (x < e.lo) x = e.lo;
        if (x >= e.hi) x = e.hi;
        x = x * e.mask;
    }
    const int slot = e.out_feat >> 20, q = e.out_feat & 0xFFFFF;
    const int64_t at = (int64_t)q * p.out_fs[slot] + ns * p.out_ss[slot];
    if (p.out64) static_cast<double *>(p.out[slot])[at] = (double)x;
    else static_cast<float *>(p.out[slot])[at] = x;
    if (e.res >= 0) {  // residual output: after = before + value
        const int rslot = e.res >> 8, rs = e.res & 0xFF;
        const float before = (float)static_cast<const Raw *>(p.src[rs])[(int64_t)q * p.src_fs[rs] + ns * p.src_ss[rs]];
        const int64_t rat = (int64_t)q * p.out_fs[rslot] + ns * p.out_ss[rslot];
        if (p.out64) static_cast<double *>(p.out[rslot])[rat] = (double)(before + x);
        else static_cast<float *>(p.out[rslot])[rat] = before + x;
    }
}

}  // namespace
}  // namespace fv3hip

using namespace fv3hip;

// ---------------------------------------------------------------------------------------------
// Model object
// ---------------------------------------------------------------------------------------------
struct fv3hip_mlp {
    int device = 0;
    int HT = 0;
    int n_sources = 0, n_inputs = 0, K = 0, width = 0, n_hidden = 0, n_outputs = 0, F = 0, n_residual = 0;
    int n_chunks1 = 0, n_otiles = 0, n_hout_tiles = 0, n_ktab = 0, n_otab = 0, n_bias = 0;
    int smallf_off = 0, smallf_n = 0;  // small-output path: where its [HT*32][4] weights sit in the bias block; 0 = not eligible
    int64_t flops = 0;
    int has_limits = 0;
    int n_log_chunks = 0, n_logfast_chunks = 0;
    unsigned int w_bytes = 0;
    void *d_sink = nullptr;
    size_t sink_bytes = 0;
    void *d_w = nullptr, *d_ktab = nullptr, *d_otab = nullptr, *d_bias = nullptr;
    // plain (operand-layout) copies of the weights for mlp_small_kernel
    void *d_w1 = nullptr, *d_wh = nullptr, *d_wo = nullptr, *d_bh = nullptr, *d_bo = nullptr;
    int Wp = 0, Fp = 0;
    int64_t small_limit = -1;  // fv3hip_mlp_set_small_limit: -1 = default rule, 0 = never, n = calls of at most n samples
    // layered path (networks the fused kernels do not hold): d_w1 / d_wh / d_wo / d_bh / d_bo are its matrices
    // ([n_ktab][Wp], [Wp][Wp] each, [Wp or n_ktab][Fp]; Wp and Fp multiples of 64), d_scr its two activation buffers
    int layered = 0, relu = 1;
    void *d_scr[2] = {nullptr, nullptr};
    size_t scr_bytes = 0;
    int n_cu = 256;
    size_t lds_bytes = 0;
    char last_variant[160] = {0};  // what the last fv3hip_mlp_predict launched (fv3hip_mlp_last_variant)
};

namespace {

const int kHiddenTilings[] = {1, 2, 4, 8};  // compiled kernel variants: hidden width <= 32 * HT

template <int HT, bool SRC64, bool XBULK, bool HOUT = false, bool SMALLF = false, bool L1SHORT = false>
int launch_one(const MlpLaunch &lp, int grid, size_t lds, hipStream_t st)
{
    auto kern = mlp_fused_kernel<HT, SRC64, XBULK, HOUT, SMALLF, L1SHORT>;
    FV3HIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, lp);
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

// ---- the tables both kernel families and the layered path read ----
struct InputTable {
    std::vector<KEntry> ktab;
    std::vector<int> perm;     // table row -> original input feature, -1 = padding
    int n_log_padded = 0;      // rows at the head of the table that take the logarithm (padded to whole 32-row chunks if room)
    bool eps_normal = true;    // every logarithm's floor is a normal number
};

// Network input k' is original input feature perm[k']: the log-transformed features come first (any order of the
// contraction index is the same dense layer), so that whole 32-row chunks are either with or without the transform.
void build_input_table(const fv3hip_mlp_desc_t *d, int K, int n_ktab, InputTable &t)
{
    t.ktab.assign(n_ktab, KEntry{-1, 0, 0.f, 1.f, 0, 0.f, 0, 0});
    t.perm.clear();
    t.perm.reserve(n_ktab);
    std::vector<KEntry> orig(K);
    int k = 0;
    for (int i = 0; i < d->n_inputs; ++i)
        for (int f = 0; f < d->in_nfeat[i]; ++f, ++k) {
            KEntry &e = orig[k];
            e = KEntry{-1, 0, 0.f, 1.f, 0, 0.f, 0, 0};
            e.src = d->in_source[i];
            e.feat = d->in_feat_start[i] + f;
            e.center = d->in_center ? d->in_center[k] : 0.f;
            e.scale = d->in_scale ? (float)(1.0 / (double)d->in_scale[k]) : 1.f;  // reciprocal
            e.transform = d->in_transform ? d->in_transform[i] : 0;
            e.eps = d->in_eps ? d->in_eps[i] : 0.f;
        }
    for (int k2 = 0; k2 < K; ++k2)
        if (orig[k2].transform == FV3HIP_TRANSFORM_LOG) t.perm.push_back(k2);
    const int n_log = (int)t.perm.size();
    // if the chunk count allows, pad the log block to whole chunks (entries -1: zero weight rows reading a constant,
    // eps = 1 so that the logarithm is of a normal number) -- then no chunk mixes both kinds and every log chunk takes
    // the fast path
    const int n_pad = (32 - n_log % 32) % 32;
    if (n_log > 0 && n_log + n_pad + (K - n_log) <= n_ktab)
        for (int i = 0; i < n_pad; ++i) t.perm.push_back(-1);
    t.n_log_padded = (int)t.perm.size();
    for (int k2 = 0; k2 < K; ++k2)
        if (orig[k2].transform != FV3HIP_TRANSFORM_LOG) t.perm.push_back(k2);
    for (size_t k2 = 0; k2 < t.perm.size(); ++k2) {
        if (t.perm[k2] >= 0) {
            t.ktab[k2] = orig[t.perm[k2]];
        } else {
            t.ktab[k2].transform = FV3HIP_TRANSFORM_LOG;
            t.ktab[k2].eps = 1.f;
        }
    }
    t.eps_normal = true;
    for (int k2 = 0; k2 < t.n_log_padded; ++k2) t.eps_normal = t.eps_normal && t.ktab[k2].eps >= FLT_MIN;
}

// Rows [0, n_hidden_rows): the last hidden layer's features (hidden-output models), stored to the slot after the outputs
// and the residual outputs; rows first_out + f: output feature f.
void build_output_table(const fv3hip_mlp_desc_t *d, int n_otab, int first_out, int n_hidden_rows, std::vector<OEntry> &otab)
{
    otab.assign(n_otab, OEntry{1.f, 0.f, -INFINITY, INFINITY, 1.f, -1, -1, 0});
    for (int q = 0; q < n_hidden_rows; ++q) otab[q].out_feat = ((d->n_outputs + d->n_residual) << 20) | q;
    int f = 0;
    for (int j = 0; j < d->n_outputs; ++j) {
        int res = -1;
        for (int r = 0; r < d->n_residual; ++r)
            if (d->res_output[r] == j) res = ((d->n_outputs + r) << 8) | d->res_source[r];
        for (int q = 0; q < d->out_nfeat[j]; ++q, ++f) {
            OEntry &e = otab[first_out + f];
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

}  // namespace

namespace {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

int create_layered(const fv3hip_mlp_desc_t *d, int K, fv3hip_mlp_t *out)
{
    int F = 0;
    for (int j = 0; j < d->n_outputs; ++j) {
        FV3HIP_REQUIRE(d->out_nfeat[j] >= 1 && d->out_nfeat[j] < (1 << 20), "bad out_nfeat[%d]", j);
        F += d->out_nfeat[j];
    }
    for (int r = 0; r < d->n_residual; ++r) {
        FV3HIP_REQUIRE(d->res_source[r] >= 0 && d->res_source[r] < d->n_sources, "res_source[%d] out of range", r);
        FV3HIP_REQUIRE(d->res_output[r] >= 0 && d->res_output[r] < d->n_outputs, "res_output[%d] out of range", r);
    }
    const int nh = d->n_hidden, width = nh ? d->width : K;
    FV3HIP_REQUIRE(K < 65536 && F < 65536 && width < 65536, "more than 65 535 features in a layer");
    fv3hip_mlp *m = new fv3hip_mlp();
    hipGetDevice(&m->device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m->device) == hipSuccess) m->n_cu = prop.multiProcessorCount;
    m->layered = 1;
    m->relu = d->hidden_activation == FV3HIP_ACT_RELU;
    m->n_sources = d->n_sources;
    m->n_inputs = d->n_inputs;
    m->K = K;
    m->width = width;
    m->n_hidden = nh;
    m->n_outputs = d->n_outputs;
    m->F = F;
    m->n_residual = d->n_residual;
    m->has_limits = (d->out_min || d->out_max || d->out_mask) ? 1 : 0;
    m->n_ktab = round_up(K, 32);
    m->Wp = nh ? round_up(width, 64) : 0;
    m->Fp = round_up(F, 64);
    m->n_otab = m->Fp;
    m->flops = nh ? 2 * ((int64_t)K * width + (int64_t)(nh - 1) * width * width + (int64_t)width * F) : 2 * (int64_t)K * F;
    InputTable it;
    build_input_table(d, K, m->n_ktab, it);
    std::vector<OEntry> otab;
    build_output_table(d, m->n_otab, 0, 0, otab);
    const int Wp = m->Wp, Fp = m->Fp, kin_o = nh ? Wp : m->n_ktab;
    auto oscale = [&](int f) { return d->out_scale ? d->out_scale[f] : 1.f; };
    std::vector<float> w1((size_t)(nh ? m->n_ktab * (size_t)Wp : 0), 0.f), wh((size_t)(nh > 1 ? nh - 1 : 0) * Wp * Wp, 0.f),
        wo((size_t)kin_o * Fp, 0.f), bh((size_t)nh * Wp, 0.f), bo((size_t)Fp, 0.f);
    for (size_t k = 0; k < it.perm.size(); ++k) {
        if (it.perm[k] < 0) continue;
        if (nh)
            for (int f = 0; f < width; ++f) w1[k * Wp + f] = d->hidden_kernels[0][(size_t)it.perm[k] * width + f] * it.ktab[k].scale;
        else  // the only layer carries both foldings: 1 / std of its input row, the scale of its output column
            for (int f = 0; f < F; ++f)
                wo[k * Fp + f] = (float)((double)d->out_kernel[(size_t)it.perm[k] * F + f] * (double)it.ktab[k].scale * (double)oscale(f));
    }
    for (int l = 1; l < nh; ++l)
        for (int k = 0; k < width; ++k)
            for (int f = 0; f < width; ++f) wh[((size_t)(l - 1) * Wp + k) * Wp + f] = d->hidden_kernels[l][(size_t)k * width + f];
    for (int l = 0; l < nh; ++l)
        for (int f = 0; f < width; ++f) bh[(size_t)l * Wp + f] = d->hidden_biases[l][f];
    if (nh)
        for (int k = 0; k < width; ++k)
            for (int f = 0; f < F; ++f) wo[(size_t)k * Fp + f] = d->out_kernel[(size_t)k * F + f] * oscale(f);
    for (int f = 0; f < F; ++f)
        bo[f] = (float)((double)d->out_bias[f] * oscale(f) + (d->out_center ? d->out_center[f] : 0.f));
    int rc;
    if ((rc = upload(w1, &m->d_w1)) || (rc = upload(wh, &m->d_wh)) || (rc = upload(wo, &m->d_wo)) || (rc = upload(bh, &m->d_bh)) ||
        (rc = upload(bo, &m->d_bo)) || (rc = upload(it.ktab, &m->d_ktab)) || (rc = upload(otab, &m->d_otab))) {
        fv3hip_mlp_destroy(m);
        return rc;
    }
    *out = m;
    return FV3HIP_OK;
}

// One slab of at most kLayeredSlab samples at a time through gather -> layers -> scatter (the scratch stays bounded:
// 2 x rows x slab x 4 bytes, whatever the call's size); all on the caller's stream, so slabs follow each other.
constexpr int64_t kLayeredSlab = 65536;

int predict_layered(fv3hip_mlp *m, const MlpLaunch &lp, bool src64, int out_dtype, int64_t n_samples, hipStream_t st)
{
    const int rows = std::max(m->n_ktab, std::max(m->Wp, m->Fp));
    const int64_t np_max = (std::min(n_samples, kLayeredSlab) + 255) / 256 * 256;
    const size_t need = (size_t)rows * np_max * sizeof(float);
    if (m->scr_bytes < need) {  // (grown on demand, like the fused kernels' scratch row: warm up before capturing a graph)
        for (void *&q : m->d_scr) {
            if (q) FV3HIP_CHECK_HIP(hipFree(q));
            q = nullptr;
        }
        m->scr_bytes = 0;
        FV3HIP_CHECK_HIP(hipMalloc(&m->d_scr[0], need));
        FV3HIP_CHECK_HIP(hipMalloc(&m->d_scr[1], need));
        m->scr_bytes = need;
    }
    LayeredIo io;
    memset(&io, 0, sizeof(io));
    io.ktab = static_cast<const KEntry *>(m->d_ktab);
    io.otab = static_cast<const OEntry *>(m->d_otab);
    io.n_samples = n_samples;
    io.out64 = (out_dtype == FV3HIP_F64);
    io.has_limits = m->has_limits;
    memcpy(io.src, lp.src, sizeof(io.src));
    memcpy(io.src_fs, lp.src_fs, sizeof(io.src_fs));
    memcpy(io.src_ss, lp.src_ss, sizeof(io.src_ss));
    memcpy(io.out, lp.out, sizeof(io.out));
    memcpy(io.out_fs, lp.out_fs, sizeof(io.out_fs));
    memcpy(io.out_ss, lp.out_ss, sizeof(io.out_ss));
    snprintf(m->last_variant, sizeof(m->last_variant), "layered (gather, %d x layered_dense_kernel, scatter; slabs of %lld samples)",
             m->n_hidden + 1, (long long)kLayeredSlab);
    for (int64_t n0 = 0; n0 < n_samples; n0 += kLayeredSlab) {
        const int64_t np = (std::min(n_samples - n0, kLayeredSlab) + 255) / 256 * 256;
        float *cur = static_cast<float *>(m->d_scr[0]), *nxt = static_cast<float *>(m->d_scr[1]);
        io.n0 = n0;
        io.np = np;
        io.x = cur;
        const dim3 gg((unsigned)(np / 256), (unsigned)m->n_ktab);
        if (src64) hipLaunchKernelGGL(layered_gather_kernel<true>, gg, dim3(256), 0, st, io);
        else hipLaunchKernelGGL(layered_gather_kernel<false>, gg, dim3(256), 0, st, io);
        int rc = check_launch("layered_gather_kernel");
        if (rc) return rc;
        auto dense = [&](const void *w, const void *b, int kp, int fp, int relu) {
            LayeredDense dp;
            dp.w = static_cast<const float *>(w);
            dp.b = static_cast<const float *>(b);
            dp.x = cur;
            dp.y = nxt;
            dp.kp = kp;
            dp.fp = fp;
            dp.relu = relu;
            dp.np = np;
            hipLaunchKernelGGL(layered_dense_kernel, dim3((unsigned)(np / 256), (unsigned)(fp / 64)), dim3(256), 0, st, dp);
            std::swap(cur, nxt);
            return check_launch("layered_dense_kernel");
        };
        for (int l = 0; l < m->n_hidden; ++l) {
            const float *w = l == 0 ? static_cast<const float *>(m->d_w1) : static_cast<const float *>(m->d_wh) + (size_t)(l - 1) * m->Wp * m->Wp;
            if ((rc = dense(w, static_cast<const float *>(m->d_bh) + (size_t)l * m->Wp, l == 0 ? m->n_ktab : m->Wp, m->Wp, m->relu))) return rc;
        }
        if ((rc = dense(m->d_wo, m->d_bo, m->n_hidden ? m->Wp : m->n_ktab, m->Fp, 0))) return rc;
        io.x = cur;
        const dim3 gs((unsigned)(np / 256), (unsigned)m->F);
        if (src64) hipLaunchKernelGGL(layered_scatter_kernel<true>, gs, dim3(256), 0, st, io);
        else hipLaunchKernelGGL(layered_scatter_kernel<false>, gs, dim3(256), 0, st, io);
        if ((rc = check_launch("layered_scatter_kernel"))) return rc;
    }
    return FV3HIP_OK;
}

}  // namespace

extern "C" int fv3hip_mlp_create(const fv3hip_mlp_desc_t *d, fv3hip_mlp_t *out)
{
    FV3HIP_REQUIRE(d && out, "null pointer");
    *out = nullptr;
    FV3HIP_REQUIRE(d->n_sources >= 1 && d->n_sources <= kMaxSources, "n_sources must be in [1, %d], got %d", kMaxSources, d->n_sources);
    FV3HIP_REQUIRE(d->n_inputs >= 1, "n_inputs must be >= 1");
    const int hout = d->hidden_output ? 1 : 0;
    FV3HIP_REQUIRE(d->n_outputs >= 1 || (d->n_outputs == 0 && hout), "n_outputs must be >= 1 (or 0 with hidden_output)");
    FV3HIP_REQUIRE(d->n_residual >= 0 && d->n_outputs + d->n_residual + hout <= kMaxOutputs,
                   "n_outputs + n_residual (+ the hidden output) must be <= %d", kMaxOutputs);
    FV3HIP_REQUIRE(d->width >= 1, "width must be >= 1");
    FV3HIP_REQUIRE(d->n_hidden >= 0, "negative n_hidden");
    FV3HIP_REQUIRE(d->hidden_activation == FV3HIP_ACT_RELU || d->hidden_activation == FV3HIP_ACT_LINEAR, "unknown activation %d", d->hidden_activation);

    int K = 0;
    for (int i = 0; i < d->n_inputs; ++i) {
        FV3HIP_REQUIRE(d->in_source[i] >= 0 && d->in_source[i] < d->n_sources, "in_source[%d] out of range", i);
        FV3HIP_REQUIRE(d->in_nfeat[i] >= 1 && d->in_feat_start[i] >= 0, "bad feature range for input %d", i);
        K += d->in_nfeat[i];
    }
    // what the fused kernels do not hold goes layer by layer (see layered_dense_kernel)
    const bool layered = d->n_hidden < 1 || d->width > 256 || K > kMaxK || d->hidden_activation != FV3HIP_ACT_RELU;
    if (layered) {
        if (hout) return fail(FV3HIP_EUNSUPPORTED, "hidden-output models need 1+ ReLU hidden layers of width <= 256 and <= %d inputs", kMaxK);
        FV3HIP_REQUIRE(d->n_hidden == 0 || d->width >= 1, "width must be >= 1");
        return create_layered(d, K, out);
    }
    int F = 0;
    for (int j = 0; j < d->n_outputs; ++j) {
        FV3HIP_REQUIRE(d->out_nfeat[j] >= 1 && d->out_nfeat[j] < (1 << 20), "bad out_nfeat[%d]", j);
        F += d->out_nfeat[j];
    }
    for (int r = 0; r < d->n_residual; ++r) {
        FV3HIP_REQUIRE(d->res_source[r] >= 0 && d->res_source[r] < d->n_sources, "res_source[%d] out of range", r);
        FV3HIP_REQUIRE(d->res_output[r] >= 0 && d->res_output[r] < d->n_outputs, "res_output[%d] out of range", r);
    }

    // pick the kernel variant: the smallest hidden tiling that holds `width`
    const int width = d->width;
    const int nt_out = (F + 31) / 32;
    int HT = 0;
    for (int v : kHiddenTilings)
        if (!HT && v * 32 >= width) HT = v;
    FV3HIP_REQUIRE(HT > 0, "no kernel variant for width %d", width);

    fv3hip_mlp *m = new fv3hip_mlp();
    hipGetDevice(&m->device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m->device) == hipSuccess) m->n_cu = prop.multiProcessorCount;
    m->HT = HT;
    m->n_sources = d->n_sources;
    m->n_inputs = d->n_inputs;
    m->K = K;
    m->width = width;
    m->n_hidden = d->n_hidden;
    m->n_outputs = d->n_outputs;
    m->F = F;
    m->n_residual = d->n_residual;
    m->has_limits = (d->out_min || d->out_max || d->out_mask) ? 1 : 0;
    m->n_chunks1 = ((K + 1) / 2 + 15) / 16;
    m->n_otiles = nt_out;
    m->n_ktab = 2 * 16 * m->n_chunks1;
    m->n_hout_tiles = hout ? HT : 0;
    m->n_otab = 32 * (m->n_hout_tiles + nt_out);
    m->n_bias = d->n_hidden * HT * 32 + nt_out * 32;
    if (F >= 1 && F <= 4 && !m->has_limits && d->n_residual == 0) {  // eligible for the small-output path (see the kernel)
        m->smallf_n = F;
        m->smallf_off = m->n_bias;
        m->n_bias += HT * 32 * 4;
    }
    m->flops = 2 * ((int64_t)K * width + (int64_t)(d->n_hidden - 1) * width * width + (int64_t)width * F);

    const int HG = (HT + 3) / 4;
    const int GS = (HT == 1) ? 4 : 8, NGO = GS / 4, KC_O = HT * 16 / GS;  // as in the kernel
    const int64_t CH_H = 16 * HG * 64, CH_O = (int64_t)KC_O * NGO * 64;  // float4 per chunk
    const int n_hid_chunks = m->n_chunks1 + (d->n_hidden - 1) * HT;
    const int n_out_chunks = nt_out;

    // ---- input table ----
    InputTable it;
    build_input_table(d, K, m->n_ktab, it);
    std::vector<KEntry> &ktab = it.ktab;
    std::vector<int> &perm = it.perm;
    m->n_log_chunks = (it.n_log_padded + 31) / 32;
    m->n_logfast_chunks = it.eps_normal ? it.n_log_padded / 32 : 0;
    // ---- packed weight stream ----
    // (+ one maximal chunk of zero padding: the two-half staging may read past a short last chunk)
    std::vector<float> w((size_t)(n_hid_chunks * CH_H + n_out_chunks * CH_O + (CH_H > CH_O ? CH_H : CH_O)) * 4, 0.f);
    auto hid_slot = [&](int g, int s, int j, int lane, int e) -> float & {
        return w[(size_t)((g * CH_H + ((int64_t)(s * HG + j) * 64 + lane)) * 4 + e)];
    };
    // layer 1: k = 2 * kpair + half.  The inputs' 1 / (std + eps) is folded into the row of the kernel that
    // multiplies them (one rounding per weight, once): the kernel only subtracts the mean.
    {
        const float *W = d->hidden_kernels[0];
        for (int g = 0; g < m->n_chunks1; ++g)
            for (int s = 0; s < 16; ++s)
                for (int j = 0; j < HG; ++j)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int k = 2 * (g * 16 + s) + (lane >> 5);
                            const int f = 32 * (4 * j + e) + (lane & 31);
                            if (k < (int)perm.size() && perm[k] >= 0 && f < width)
                                hid_slot(g, s, j, lane, e) = W[(size_t)perm[k] * width + f] * ktab[k].scale;
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
    // output layer (the outputs' denormalisation y * scale + center is folded into kernel and bias, so
    // the accumulator already holds the physical value), tile-major: chunk t = output features 32t..32t+31 over the whole contraction;
    // slot s holds k-pairs GS*s .. GS*s+GS-1, four per float4: k-pair m pairs hidden activations
    // k = 32*(m/16) + rho(m%16) + 4*half (the accumulator layout of the last hidden layer)
    {
        const float *W = d->out_kernel;
        const size_t base = (size_t)n_hid_chunks * CH_H * 4;
        for (int t = 0; t < nt_out; ++t)
            for (int s = 0; s < KC_O; ++s)
                for (int j = 0; j < NGO; ++j)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int mm = GS * s + 4 * j + e;
                            const int k = 32 * (mm / 16) + rho(mm % 16) + 4 * (lane >> 5);
                            const int f = 32 * t + (lane & 31);
                            if (k < width && f < F)
                                w[base + (size_t)((t * CH_O + ((int64_t)(s * NGO + j) * 64 + lane)) * 4 + e)] =
                                    W[(size_t)k * F + f] * (d->out_scale ? d->out_scale[f] : 1.f);
                        }
    }
    // ---- output table ----
    // (hidden-output models: the table starts with the last hidden layer's features, stored to the output slot
    // after the outputs and the residual outputs)
    std::vector<OEntry> otab;
    build_output_table(d, m->n_otab, 32 * m->n_hout_tiles, hout ? width : 0, otab);
    // ---- biases: [layer][tile][reg][half] ----
    std::vector<float> bias(m->n_bias, 0.f);
    for (int l = 0; l < d->n_hidden; ++l)
        for (int t = 0; t < HT; ++t)
            for (int r = 0; r < 16; ++r)
                for (int hf = 0; hf < 2; ++hf) {
                    const int f = 32 * t + rho(r) + 4 * hf;
                    if (f < width) bias[(size_t)l * HT * 32 + (t * 16 + r) * 2 + hf] = d->hidden_biases[l][f];
                }
    for (int t = 0; t < nt_out; ++t)
        for (int r = 0; r < 16; ++r)
            for (int hf = 0; hf < 2; ++hf) {
                const int f = 32 * t + rho(r) + 4 * hf;
                if (f < F)
                    bias[(size_t)d->n_hidden * HT * 32 + (size_t)t * 32 + r * 2 + hf] =
                        (float)((double)d->out_bias[f] * (d->out_scale ? d->out_scale[f] : 1.f) + (d->out_center ? d->out_center[f] : 0.f));
            }

    if (m->smallf_n)  // [half][output f][hidden tile t][accumulator register r] -> feature k = 32 t + rho(r) + 4 half; scale folded in
        for (int hf = 0; hf < 2; ++hf)
            for (int f = 0; f < F; ++f)
                for (int t = 0; t < HT; ++t)
                    for (int r = 0; r < 16; ++r) {
                        const int k = 32 * t + rho(r) + 4 * hf;
                        if (k < width)
                            bias[(size_t)m->smallf_off + (size_t)(((hf * 4 + f) * HT + t) * 16 + r)] =
                                d->out_kernel[(size_t)k * F + f] * (d->out_scale ? d->out_scale[f] : 1.f);
                    }

    // ---- plain copies for mlp_small_kernel: the same folded values, rows in table order ----
    const int Wp = HT * 32, Fp = (nt_out > 0 ? nt_out : 1) * 32;
    m->Wp = Wp;
    m->Fp = Fp;
    std::vector<float> w1((size_t)m->n_ktab * Wp, 0.f), wh((size_t)(d->n_hidden > 1 ? d->n_hidden - 1 : 1) * Wp * Wp, 0.f),
        wo((size_t)Wp * Fp, 0.f), bh((size_t)d->n_hidden * Wp, 0.f), bo((size_t)Fp, 0.f);
    for (size_t k = 0; k < perm.size(); ++k)
        if (perm[k] >= 0)
            for (int f = 0; f < width; ++f) w1[k * Wp + f] = d->hidden_kernels[0][(size_t)perm[k] * width + f] * ktab[k].scale;
    for (int l = 1; l < d->n_hidden; ++l)
        for (int k = 0; k < width; ++k)
            for (int f = 0; f < width; ++f) wh[((size_t)(l - 1) * Wp + k) * Wp + f] = d->hidden_kernels[l][(size_t)k * width + f];
    for (int l = 0; l < d->n_hidden; ++l)
        for (int f = 0; f < width; ++f) bh[(size_t)l * Wp + f] = d->hidden_biases[l][f];
    for (int k = 0; k < width; ++k)
        for (int f = 0; f < F; ++f) wo[(size_t)k * Fp + f] = d->out_kernel[(size_t)k * F + f] * (d->out_scale ? d->out_scale[f] : 1.f);
    for (int f = 0; f < F; ++f)
        bo[f] = (float)((double)d->out_bias[f] * (d->out_scale ? d->out_scale[f] : 1.f) + (d->out_center ? d->out_center[f] : 0.f));

    int rc;
    if ((rc = upload(w1, &m->d_w1)) || (rc = upload(wh, &m->d_wh)) || (rc = upload(wo, &m->d_wo)) || (rc = upload(bh, &m->d_bh)) ||
        (rc = upload(bo, &m->d_bo))) {
        fv3hip_mlp_destroy(m);
        return rc;
    }
    FV3HIP_REQUIRE(w.size() * sizeof(float) < (1ull << 31), "model too large: the packed weight stream exceeds 2 GiB");
    m->w_bytes = (unsigned int)(w.size() * sizeof(float));
    if ((rc = upload(w, &m->d_w)) || (rc = upload(ktab, &m->d_ktab)) || (rc = upload(otab, &m->d_otab)) ||
        (rc = upload(bias, &m->d_bias))) {
        fv3hip_mlp_destroy(m);
        return rc;
    }
    const size_t wb = 2 * (size_t)((CH_H > CH_O) ? CH_H : CH_O) * 16;
    // (the fast-I/O kernels add the 32 KB input tiles at launch)
    m->lds_bytes = wb + (size_t)m->n_ktab * sizeof(KEntry) +
                   (size_t)m->n_otab * (sizeof(OFast) + sizeof(OSlow) + (d->n_residual ? sizeof(ORes) : 0)) +
                   (size_t)((m->n_bias + 3) & ~3) * sizeof(float) + (size_t)(3 * kMaxSources + 3 * kMaxOutputs) * 8;
    if (m->lds_bytes > 160 * 1024) {
        fv3hip_mlp_destroy(m);
        return fail(FV3HIP_EUNSUPPORTED, "model tables need %zu bytes of LDS (> 160 KiB)", m->lds_bytes);
    }
    *out = m;
    return FV3HIP_OK;
}

#ifdef SMALL_STAMPS
static unsigned long long *g_small_stamps = nullptr;
extern "C" void fv3hip_diag_set_small_stamps(void *p) { g_small_stamps = static_cast<unsigned long long *>(p); }
#endif
#ifdef MLP_STAMPS
static unsigned long long *g_mlp_stamps = nullptr;
extern "C" void fv3hip_diag_set_mlp_stamps(void *p) { g_mlp_stamps = static_cast<unsigned long long *>(p); }
#endif

extern "C" int fv3hip_mlp_destroy(fv3hip_mlp_t m)
{
    if (!m) return FV3HIP_OK;
    if (m->d_sink) hipFree(m->d_sink);
    if (m->d_w) hipFree(m->d_w);
    if (m->d_ktab) hipFree(m->d_ktab);
    if (m->d_otab) hipFree(m->d_otab);
    if (m->d_bias) hipFree(m->d_bias);
    for (void *q : {m->d_w1, m->d_wh, m->d_wo, m->d_bh, m->d_bo, m->d_scr[0], m->d_scr[1]})
        if (q) hipFree(q);
    delete m;
    return FV3HIP_OK;
}

extern "C" int64_t fv3hip_mlp_flops_per_sample(fv3hip_mlp_t m) { return m ? m->flops : 0; }

extern "C" int fv3hip_mlp_set_small_limit(fv3hip_mlp_t m, int64_t max_samples)
{
    FV3HIP_REQUIRE(m, "null model handle");
    m->small_limit = max_samples < 0 ? -1 : max_samples;
    return FV3HIP_OK;
}

extern "C" const char *fv3hip_mlp_last_variant(fv3hip_mlp_t m) { return m ? m->last_variant : ""; }

extern "C" int fv3hip_mlp_predict(fv3hip_mlp_t m, const void *const *sources, const int *src_dtype,
                                  const int64_t *src_feat_stride, const int64_t *src_sample_stride,
                                  int64_t n_samples, void *const *outputs, int out_dtype,
                                  const int64_t *out_feat_stride, const int64_t *out_sample_stride,
                                  void *stream)
{
    FV3HIP_REQUIRE(m, "null model handle");
    FV3HIP_REQUIRE(n_samples >= 0, "negative n_samples");
    if (n_samples == 0) return FV3HIP_OK;
    {   // the model's tables (and the scratch row this call may grow) live on the device it was created on
        int cur = -1;
        FV3HIP_CHECK_HIP(hipGetDevice(&cur));
        FV3HIP_REQUIRE(cur == m->device, "the model lives on device %d but the current device is %d: make the model's "
                       "device current around fv3hip_mlp_predict", m->device, cur);
    }
    FV3HIP_REQUIRE(sources && src_dtype && src_feat_stride && src_sample_stride && outputs &&
                       out_feat_stride && out_sample_stride, "null pointer");
    FV3HIP_REQUIRE(out_dtype == FV3HIP_F32 || out_dtype == FV3HIP_F64, "out_dtype must be F32 or F64");
    MlpLaunch lp;
    memset(&lp, 0, sizeof(lp));
    const int dt0 = src_dtype[0];
    const bool src64 = (dt0 == FV3HIP_F64);
    FV3HIP_REQUIRE(dt0 == FV3HIP_F32 || dt0 == FV3HIP_F64, "source dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_samples < ((int64_t)1 << 32), "n_samples must be below 2^32");
    for (int i = 0; i < m->n_sources; ++i) {
        FV3HIP_REQUIRE(sources[i], "source %d is null", i);
        FV3HIP_REQUIRE(src_sample_stride[i] >= 0 && src_sample_stride[i] * (dt0 == FV3HIP_F64 ? 8 : 4) < ((int64_t)1 << 32),
                       "sample stride of source %d must be in [0, 4 GiB)", i);
        if (src_dtype[i] != dt0)
            return fail(FV3HIP_EUNSUPPORTED, "all sources must share one dtype (source 0 is %d, source %d is %d)", dt0, i, src_dtype[i]);
        lp.src[i] = sources[i];
        lp.src_fs[i] = src_feat_stride[i];
        lp.src_ss[i] = src_sample_stride[i];
    }
    for (int i = m->n_sources; i < kMaxSources; ++i) lp.src[i] = sources[0];
    const int n_slots = m->n_outputs + m->n_residual + (m->n_hout_tiles ? 1 : 0);
    for (int j = 0; j < n_slots; ++j) {
        FV3HIP_REQUIRE(outputs[j], "output %d is null", j);
        FV3HIP_REQUIRE(out_sample_stride[j] >= 0 && out_sample_stride[j] * (out_dtype == FV3HIP_F64 ? 8 : 4) < ((int64_t)1 << 32),
                       "sample stride of output %d must be in [0, 4 GiB)", j);
        lp.out[j] = outputs[j];
        lp.out_fs[j] = out_feat_stride[j];
        lp.out_ss[j] = out_sample_stride[j];
    }
    if (m->layered) return predict_layered(m, lp, src64, out_dtype, n_samples, as_stream(stream));
    // ---- few samples: the feature-split kernel while the big one would leave CUs without a tile (see mlp_small_kernel) ----
    {
        static const int64_t small_max = [] {
            const char *e = getenv("FV3HIP_MLP_SMALL_MAX_SAMPLES");  // 0 disables; default: one round of 32-sample workgroups over the CUs
            return e ? (int64_t)atoll(e) : (int64_t)-1;
        }();
        // (measured, profiles/r03_bench.json: a 32-sample workgroup takes 0.5-0.7 of the time of a 128-sample tile of the
        // big kernel, so a second round of workgroups -- 9 216 columns on 256 CUs -- already loses to the big kernel's one)
        const int64_t limit = m->small_limit >= 0 ? m->small_limit : small_max >= 0 ? small_max : (int64_t)32 * m->n_cu;
        const size_t lds_small = (size_t)(2 * kSmallChunk * 32 + 2 * m->Wp * 32) * sizeof(float) + (size_t)m->n_ktab * 32 +
                                 (size_t)3 * (kMaxOutputs + kMaxSources) * sizeof(int64_t);
        if (n_samples <= limit && m->n_otiles + m->n_hout_tiles > 0 && lds_small <= 160 * 1024) {
            SmallLaunch sp;
            memset(&sp, 0, sizeof(sp));
            sp.w1 = static_cast<const float *>(m->d_w1);
            sp.wh = static_cast<const float *>(m->d_wh);
            sp.wo = static_cast<const float *>(m->d_wo);
            sp.bh = static_cast<const float *>(m->d_bh);
            sp.bo = static_cast<const float *>(m->d_bo);
            sp.ktab = static_cast<const KEntry *>(m->d_ktab);
            sp.otab = static_cast<const OEntry *>(m->d_otab);
            sp.n_ktab = m->n_ktab;
            sp.Wp = m->Wp;
            sp.HT = m->HT;
            sp.n_hidden = m->n_hidden;
            sp.Fp = m->Fp;
            sp.n_otiles = m->n_otiles;
            sp.n_hout_tiles = m->n_hout_tiles;
            sp.out64 = (out_dtype == FV3HIP_F64);
            sp.has_limits = m->has_limits;
            sp.n_samples = n_samples;
#ifdef SMALL_STAMPS
            sp.stamps = g_small_stamps;
#endif
            memcpy(sp.src, lp.src, sizeof(sp.src));
            memcpy(sp.src_fs, lp.src_fs, sizeof(sp.src_fs));
            memcpy(sp.src_ss, lp.src_ss, sizeof(sp.src_ss));
            memcpy(sp.out, lp.out, sizeof(sp.out));
            memcpy(sp.out_fs, lp.out_fs, sizeof(sp.out_fs));
            memcpy(sp.out_ss, lp.out_ss, sizeof(sp.out_ss));
            const int grid_s = (int)ceil_div(n_samples, (int64_t)32);
            hipStream_t st_s = as_stream(stream);
            snprintf(m->last_variant, sizeof(m->last_variant), "mlp_small_kernel<%s> (32-sample workgroups, features split over %d waves)",
                     src64 ? "true" : "false", kSmallWaves);
            if (src64) {
                auto kern = mlp_small_kernel<true>;
                FV3HIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small));
                hipLaunchKernelGGL(kern, dim3(grid_s), dim3(kSmallWaves * 64), lds_small, st_s, sp);
            } else {
                auto kern = mlp_small_kernel<false>;
                FV3HIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small));
                hipLaunchKernelGGL(kern, dim3(grid_s), dim3(kSmallWaves * 64), lds_small, st_s, sp);
            }
            return check_launch("mlp_small_kernel");
        }
    }
    lp.w = static_cast<const f32x4 *>(m->d_w);
    lp.w_bytes = m->w_bytes;
    lp.ktab = static_cast<const KEntry *>(m->d_ktab);
    lp.otab = static_cast<const OEntry *>(m->d_otab);
    lp.bias = static_cast<const float *>(m->d_bias);
    lp.n_chunks1 = m->n_chunks1;
    lp.n_hidden = m->n_hidden;
    lp.n_otiles = m->n_otiles;
    lp.n_hout_tiles = m->n_hout_tiles;
    lp.n_ktab = m->n_ktab;
    lp.n_otab = m->n_otab;
    lp.n_bias = m->n_bias;
    lp.out64 = (out_dtype == FV3HIP_F64);
    lp.has_limits = m->has_limits;
    {
        const int oal = (out_dtype == FV3HIP_F64) ? 2 : 4, sal = src64 ? 2 : 4;  // elements per 16 bytes
        bool fast = (n_samples % 32 == 0);
        for (int j = 0; j < n_slots && fast; ++j)
            fast = out_sample_stride[j] == 1 && (reinterpret_cast<uintptr_t>(outputs[j]) % 16 == 0) &&
                   (out_feat_stride[j] % oal == 0);
        for (int i = 0; i < m->n_sources && fast; ++i)
            fast = src_sample_stride[i] == 1 && (reinterpret_cast<uintptr_t>(sources[i]) % 16 == 0) &&
                   (src_feat_stride[i] % sal == 0);
        lp.epi_fast = fast ? 1 : 0;
    }
    lp.n_log_chunks = m->n_log_chunks;
    lp.n_logfast_chunks = m->n_logfast_chunks;
    lp.n_residual = m->n_residual;
    lp.n_samples = n_samples;
    lp.n_tiles = ceil_div(n_samples, kTileSamples);
#ifdef MLP_STAMPS
    lp.stamps = g_mlp_stamps;
#endif
    const int grid = (int)(lp.n_tiles < m->n_cu ? lp.n_tiles : m->n_cu);
    // fast-I/O kernels when every source and output is sample-contiguous and aligned (see the kernel)
    bool xbulk = lp.epi_fast != 0;
    hipStream_t st = as_stream(stream);
    constexpr size_t kLdsMax = 160 * 1024, kXsBytes = 2 * 32 * kTileSamples * sizeof(float);
    if (m->lds_bytes + kXsBytes > kLdsMax) xbulk = false;  // (large tables: the general kernels still fit)
    const size_t lds = m->lds_bytes + (xbulk ? kXsBytes : 0);
    lp.sink = 0;
    if (xbulk) {
        // scratch row for the padded output rows (kept with the model; grown on demand -- a first
        // call or a larger n_samples allocates, so warm the model up before capturing a graph)
        const size_t need = (size_t)n_samples * 8;
        if (m->sink_bytes < need) {
            if (m->d_sink) FV3HIP_CHECK_HIP(hipFree(m->d_sink));
            m->d_sink = nullptr;
            m->sink_bytes = 0;
            FV3HIP_CHECK_HIP(hipMalloc(&m->d_sink, need));
            m->sink_bytes = need;
        }
        lp.sink = reinterpret_cast<int64_t>(m->d_sink);
    }
    // hidden-output models (recurrent cells) have their own instantiations, so that the plain ones keep their
    // register allocation and schedule; they take float32 sources only (packed inputs and states)
    if (m->n_hout_tiles && src64)
        return fail(FV3HIP_EUNSUPPORTED, "hidden-output models take float32 sources only");
    // small-output launches (<= 4 outputs, fast I/O, float32 in and out): no output chunk in the stream
    const bool smallf = m->smallf_n && xbulk && !src64 && out_dtype == FV3HIP_F32;
    if (smallf) {
        lp.n_otiles = 0;
        lp.smallf_off = m->smallf_off;
        lp.smallf_n = m->smallf_n;
    }
    // ... and, with one layer-1 chunk of at most 16 untransformed inputs, an 8-slot first chunk (the dense-local models)
    const bool l1short = smallf && !m->n_hout_tiles && m->K <= 16 && m->n_log_chunks == 0;
    // the instantiation and the epilogue flavour of this launch, for fv3hip_mlp_last_variant: full sample tiles of a
    // fast-I/O float32 launch of the 8-tile kernel take the branch-free side epilogue ("plain", or "residual" when the
    // model has residual outputs), everything else the general one
    {
        const bool hout_v = m->n_hout_tiles != 0;
        const bool src64_v = src64 && !smallf && !hout_v;
        const bool side = xbulk && !smallf && m->HT == 8 && !m->has_limits && out_dtype == FV3HIP_F32 && n_samples >= kTileSamples;
        const char *epi = smallf ? "small-output" : (side && !m->n_residual) ? "plain"
                          : (side && !src64_v && !hout_v) ? "residual" : "general";
        snprintf(m->last_variant, sizeof(m->last_variant), "mlp_fused_kernel<%d,%s,%s,%s,%s,%s> epilogue=%s", m->HT,
                 src64_v ? "true" : "false", xbulk ? "true" : "false", hout_v ? "true" : "false",
                 smallf ? "true" : "false", l1short ? "true" : "false", epi);
    }
#define VARIANT_(H)                                                                                \
    if (m->HT == H && l1short) return launch_one<H, false, true, false, true, true>(lp, grid, lds, st); \
    if (m->HT == H && smallf)                                                                      \
        return m->n_hout_tiles ? launch_one<H, false, true, true, true>(lp, grid, lds, st)         \
                               : launch_one<H, false, true, false, true>(lp, grid, lds, st);       \
    if (m->HT == H && m->n_hout_tiles)                                                             \
        return xbulk ? launch_one<H, false, true, true>(lp, grid, lds, st) : launch_one<H, false, false, true>(lp, grid, lds, st); \
    if (m->HT == H) {                                                                              \
        if (xbulk) return src64 ? launch_one<H, true, true>(lp, grid, lds, st) : launch_one<H, false, true>(lp, grid, lds, st); \
        return src64 ? launch_one<H, true, false>(lp, grid, lds, st) : launch_one<H, false, false>(lp, grid, lds, st);          \
    }
#ifndef MLP_FAST_BUILD  // (experiments compile the flagship variant only)
    VARIANT_(1)
    VARIANT_(2)
    VARIANT_(4)
#endif
    VARIANT_(8)
#undef VARIANT_
    return fail(FV3HIP_EUNSUPPORTED, "no compiled kernel variant for HT=%d", m->HT);
}
