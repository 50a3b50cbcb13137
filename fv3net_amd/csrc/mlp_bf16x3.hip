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
// stack, output heads with scale / centre folded in, optional residual outputs `after = before + difference`, optionally the
// last hidden layer's activations as an output (a recurrent cell's state) and no output layer at all.
// Restrictions: hidden width 256, float32 sources and outputs that are sample-contiguous, every log epsilon >= FLT_MIN (the
// fast log), no output limits / masks, 1 / 3 / 5 / 13 output tiles of 32 features.  Anything else: FV3HIP_EUNSUPPORTED.
//
// Structure: a workgroup is 4 waves x 32 samples and walks 128-sample tiles persistently.  A layer is a sequence of k-steps
// of 16 contraction indices; per k-step the wave holds its B operand (8 activations per lane as three bf16x8 pieces) and runs
// 6 MFMAs per 32-feature output tile; the A operands (the three pre-split weight pieces of the k-step, [piece][tile][lane]
// [8 bf16], 24 KB for 8 tiles) are shared by the four waves through LDS.  The accumulator layout of a layer (lane = sample,
// registers = features) is, by choice of the k-slot -> feature map the host packs the weights with, exactly the B operand
// layout of the next layer's k-steps: activations never leave registers.
//   * weight stream: one packed stream per tile (L2-resident, 2.1 MB for the Zhao-Carr network), chunk = one k-step, brought
//     in by `buffer_load_dwordx4 ... lds` (no registers, no ds_write) into three LDS buffers, two k-steps ahead; one barrier
//     per k-step.  The wait is an exact `s_waitcnt vmcnt(N)`: N = the loads issued after the chunk that must have landed.
//   * a k-step is ONE inline-assembly block (generated: gen/mlp3_kstep.py -> mlp3_kstep.inc): LDS reads of the A operands one
//     tile pair ahead of the MFMAs, and -- in the shadow of the MFMAs, measured free up to ~4 instructions per MFMA pair --
//     the chunk request and the split of the NEXT k-step's activations into their pieces.
//   * layer 1: a k-step = 16 consecutive feature rows of one input (inputs padded to 16), 8 bounds-checked buffer loads per
//     lane two k-steps ahead; centre / epsilon rows ride along in the block's LDS reads.
//   * epilogue: through a per-wave LDS patch to 16 bytes per lane where alignment allows (at most 63 memory operations of a
//     wave can be in flight); the residual outputs' `before` rows in a rolling window of 5 tiles, the first 5 requested
//     during the last k-steps of the output layer.
// Diagnostics: -DMLP3_STAMPS accumulates cycle counts per phase (benchmarks/mlp3_stamps prints them).
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
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) float global_float;   // (row addresses come out of tables: say that they are global, or
                                                                // the accesses become FLAT ones, which also count as LDS traffic)

constexpr int kHT = 8;          // hidden feature tiles (width 256)
constexpr int kMaxSrc = 16;
constexpr int kMaxOut = 32;
constexpr int kMaxKs1 = 64;     // layer-1 k-steps (<= 1024 input features after padding every input to a multiple of 16)

__host__ __device__ constexpr int rho3(int r) { return (r & 3) + 8 * (r >> 2); }

// one layer-1 k-step: its 16 contraction indices are 16 consecutive features of ONE input (inputs are padded to whole
// k-steps), so its rows are `base + (8 half + j) fs4`, fetched by bounds-checked buffer loads (a padding row reads 0)
struct XStep {
    uint64_t base;   // address of (first feature of the k-step, sample 0)
    uint32_t rows;   // real feature rows of the k-step (1..16; the others read 0)
    uint32_t fs4;    // bytes per feature row
};

struct Mlp3Launch {
    const f32x4 *w;        // packed stream: per k-step chunk [piece 3][tile][lane 64] float4 (= 8 bf16)
    uint32_t w_bytes;
    const float *bias;     // [(n_hidden 8 + n_ot) tiles][half 2][reg 16]
    const float *center;   // [n_ks1][half][8] layer-1 centre per k-slot
    const float *eps;      // [n_ks1][half][8] log epsilon per k-slot (log k-steps)
    const int *ofeat;      // [n_ot * 32] (output slot << 20 | feature), -1 = padding
    const int *ores;       // [n_ot * 32] (residual slot << 8 | residual source), -1 = none
    float *sink;           // a row of n_samples floats nobody reads: where padding features / absent residuals are stored
    unsigned long long *stamps;  // diagnostic builds only (-DMLP3_STAMPS): [workgroup][wave][8] cycle sums per phase
    int n_ks1, n_log_ks, n_hidden, n_ot, n_residual;
    int x_wrap;            // 16 feature rows span 4 GiB or more: a padding row's 32-bit offset could wrap into range
    int has_out;           // 0: no output layer on the matrix cores (none at all, or the small one below)
    int small_out;         // 1..4 output features without residuals: the output layer is 128 x F fp32 FMAs per lane
    const float *wsmall;   // its weights [8 tiles][4 outputs][2 halves][16 registers] (scale folded in) + [4] biases (centre folded in)
    float *hout;           // the last hidden layer's activations [256][hout_fs] (`hidden_output` of the descriptor), or null
    int64_t hout_fs;
    int64_t n_samples, n_tiles;
    XStep xk[kMaxKs1];
    const float *src[kMaxSrc];
    int64_t src_fs[kMaxSrc];
    float *out[kMaxOut];
    int64_t out_fs[kMaxOut];
};

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS reads the compiler does not see: the weight chunks arrive by `buffer_load ... lds`, and the compiler orders every LDS
// read it knows of behind ALL such loads in flight (s_waitcnt vmcnt(0)) -- which would serialise the two-chunk-deep prefetch.
// So the LDS reads between the first and the last k-step of a tile are inline assembly.  A read and its s_waitcnt always sit
// in ONE asm statement: the compiler takes an asm output as valid when the statement ends (it may copy the register at once).
__device__ __forceinline__ void lds_read64_sync(uint32_t addr, f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(addr)
                 : "memory");
}
// two rows of 64 bytes, 128 bytes apart (the two register blocks of a tile in a row table)
__device__ __forceinline__ void lds_read2x64_sync(uint32_t addr, f32x4 (&r)[8])
{
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:144\n\tds_read_b128 %6, %8 offset:160\n\tds_read_b128 %7, %8 offset:176\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(addr)
                 : "memory");
}
// a tile's bias row (64 bytes) and its two rows of one / two row tables, one LDS round trip
__device__ __forceinline__ void lds_tile_tables_sync(uint32_t bias, uint32_t rows0, f32x4 (&bq)[4], f32x4 (&r0)[8])
{
    asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:16\n\tds_read_b128 %2, %12 offset:32\n\tds_read_b128 %3, %12 offset:48\n\t"
                 "ds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:16\n\tds_read_b128 %6, %13 offset:32\n\tds_read_b128 %7, %13 offset:48\n\t"
                 "ds_read_b128 %8, %13 offset:128\n\tds_read_b128 %9, %13 offset:144\n\tds_read_b128 %10, %13 offset:160\n\tds_read_b128 %11, %13 offset:176\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(r0[0]), "=&v"(r0[1]), "=&v"(r0[2]), "=&v"(r0[3]), "=&v"(r0[4]),
                   "=&v"(r0[5]), "=&v"(r0[6]), "=&v"(r0[7])
                 : "v"(bias), "v"(rows0)
                 : "memory");
}
// The 16-byte epilogue: a wave's 32 x 32 output tile goes through its own LDS patch ([feature][sample], rows of 36 floats) so
// that a lane ends up with four consecutive samples of one feature.  Write: the lane's 16 values (features rho3(r) + 4 half
// of its sample).  Read: features l / 8 + 8 i (i = 0..3), samples 4 (l % 8)..+3 -- together with the row addresses of those
// features from a feature-ordered table (8 bytes per feature).
constexpr int kPatchRow = 36 * 4;   // bytes
#define PATCH_WRITES_                                                                                                          \
    "ds_write_b32 %[pw], %[v0]\n\tds_write_b32 %[pw], %[v1] offset:144\n\tds_write_b32 %[pw], %[v2] offset:288\n\t"                \
    "ds_write_b32 %[pw], %[v3] offset:432\n\tds_write_b32 %[pw], %[v4] offset:1152\n\tds_write_b32 %[pw], %[v5] offset:1296\n\t"   \
    "ds_write_b32 %[pw], %[v6] offset:1440\n\tds_write_b32 %[pw], %[v7] offset:1584\n\tds_write_b32 %[pw], %[v8] offset:2304\n\t"  \
    "ds_write_b32 %[pw], %[v9] offset:2448\n\tds_write_b32 %[pw], %[v10] offset:2592\n\tds_write_b32 %[pw], %[v11] offset:2736\n\t" \
    "ds_write_b32 %[pw], %[v12] offset:3456\n\tds_write_b32 %[pw], %[v13] offset:3600\n\tds_write_b32 %[pw], %[v14] offset:3744\n\t" \
    "ds_write_b32 %[pw], %[v15] offset:3888\n\t"                                                                                   \
    "ds_read_b128 %[t0], %[pr]\n\tds_read_b128 %[t1], %[pr] offset:1152\n\tds_read_b128 %[t2], %[pr] offset:2304\n\t"              \
    "ds_read_b128 %[t3], %[pr] offset:3456\n\t"                                                                                    \
    "ds_read_b64 %[r0], %[ro]\n\tds_read_b64 %[r1], %[ro] offset:64\n\tds_read_b64 %[r2], %[ro] offset:128\n\tds_read_b64 %[r3], %[ro] offset:192\n\t" \
    "ds_read_b32 %[b0], %[bo]\n\tds_read_b32 %[b1], %[bo] offset:32\n\tds_read_b32 %[b2], %[bo] offset:64\n\tds_read_b32 %[b3], %[bo] offset:96\n\t"
#define PATCH_V_(v) "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), \
                    "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15])
// one LDS round trip per tile: write the lane's 16 values, read back the transposed tile (t), the output rows (r) and biases (b)
// of its four features ...
__device__ __forceinline__ void patch_exchange(uint32_t pw, uint32_t pr, uint32_t ro, uint32_t bo, const float (&v)[16], f32x4 (&t)[4], f32x2 (&r)[4],
                                               float (&b)[4])
{
    asm volatile(PATCH_WRITES_ "s_waitcnt lgkmcnt(0)"
                 : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]),
                   [r3] "=&v"(r[3]), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [b2] "=&v"(b[2]), [b3] "=&v"(b[3])
                 : [pw] "v"(pw), [pr] "v"(pr), [ro] "v"(ro), [bo] "v"(bo), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]),
                   [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]),
                   [v11] "v"(v[11]), [v12] "v"(v[12]), [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15])
                 : "memory");
}
// ... and, with residual outputs, the `after` rows of this tile (a) and the `before` rows of a later tile (n)
__device__ __forceinline__ void patch_exchange_res(uint32_t pw, uint32_t pr, uint32_t ro, uint32_t bo, uint32_t ao, uint32_t no, const float (&v)[16],
                                                   f32x4 (&t)[4], f32x2 (&r)[4], float (&b)[4], f32x2 (&a)[4], f32x2 (&n)[4])
{
    asm volatile(PATCH_WRITES_
                 "ds_read_b64 %[a0], %[ao]\n\tds_read_b64 %[a1], %[ao] offset:64\n\tds_read_b64 %[a2], %[ao] offset:128\n\tds_read_b64 %[a3], %[ao] offset:192\n\t"
                 "ds_read_b64 %[n0], %[no]\n\tds_read_b64 %[n1], %[no] offset:64\n\tds_read_b64 %[n2], %[no] offset:128\n\tds_read_b64 %[n3], %[no] offset:192\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]),
                   [r3] "=&v"(r[3]), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [b2] "=&v"(b[2]), [b3] "=&v"(b[3]), [a0] "=&v"(a[0]), [a1] "=&v"(a[1]),
                   [a2] "=&v"(a[2]), [a3] "=&v"(a[3]), [n0] "=&v"(n[0]), [n1] "=&v"(n[1]), [n2] "=&v"(n[2]), [n3] "=&v"(n[3])
                 : [pw] "v"(pw), [pr] "v"(pr), [ro] "v"(ro), [bo] "v"(bo), [ao] "v"(ao), [no] "v"(no), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]),
                   [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]),
                   [v11] "v"(v[11]), [v12] "v"(v[12]), [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15])
                 : "memory");
}
#undef PATCH_V_
// the exchange alone (hidden activations: no bias, rows by arithmetic)
__device__ __forceinline__ void patch_exchange_plain(uint32_t pw, uint32_t pr, const float (&v)[16], f32x4 (&t)[4])
{
    asm volatile("ds_write_b32 %[pw], %[v0]\n\tds_write_b32 %[pw], %[v1] offset:144\n\tds_write_b32 %[pw], %[v2] offset:288\n\t"
                 "ds_write_b32 %[pw], %[v3] offset:432\n\tds_write_b32 %[pw], %[v4] offset:1152\n\tds_write_b32 %[pw], %[v5] offset:1296\n\t"
                 "ds_write_b32 %[pw], %[v6] offset:1440\n\tds_write_b32 %[pw], %[v7] offset:1584\n\tds_write_b32 %[pw], %[v8] offset:2304\n\t"
                 "ds_write_b32 %[pw], %[v9] offset:2448\n\tds_write_b32 %[pw], %[v10] offset:2592\n\tds_write_b32 %[pw], %[v11] offset:2736\n\t"
                 "ds_write_b32 %[pw], %[v12] offset:3456\n\tds_write_b32 %[pw], %[v13] offset:3600\n\tds_write_b32 %[pw], %[v14] offset:3744\n\t"
                 "ds_write_b32 %[pw], %[v15] offset:3888\n\t"
                 "ds_read_b128 %[t0], %[pr]\n\tds_read_b128 %[t1], %[pr] offset:1152\n\tds_read_b128 %[t2], %[pr] offset:2304\n\t"
                 "ds_read_b128 %[t3], %[pr] offset:3456\n\ts_waitcnt lgkmcnt(0)"
                 : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3])
                 : [pw] "v"(pw), [pr] "v"(pr), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]),
                   [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]), [v11] "v"(v[11]), [v12] "v"(v[12]),
                   [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15])
                 : "memory");
}
__device__ __forceinline__ void rows_read(uint32_t rows, f32x2 (&r)[4])
{
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:64\n\tds_read_b64 %2, %4 offset:128\n\tds_read_b64 %3, %4 offset:192\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                 : "v"(rows)
                 : "memory");
}
typedef __attribute__((address_space(1))) f32x4 global_f32x4;
// (Non-temporal hints on the streaming accesses -- `nt` on the x loads, the `before` loads or the output stores, to keep them from
// displacing the weight stream in L2 -- were measured: each of the three made the launch 3-6 % slower.  Plain accesses.)
#define MLP3_NT_AUX 0
#define MLP3_NT_LOAD(p) (*(p))
#define MLP3_NT_STORE(v, p) (*(p) = (v))
// 4 rows of 64 bytes, 128 bytes apart (the bias rows of four feature tiles)
__device__ __forceinline__ void lds_read4x64_sync(uint32_t addr, f32x4 (&r)[4][4])
{
    asm volatile("ds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:16\n\tds_read_b128 %2, %16 offset:32\n\tds_read_b128 %3, %16 offset:48\n\t"
                 "ds_read_b128 %4, %16 offset:128\n\tds_read_b128 %5, %16 offset:144\n\tds_read_b128 %6, %16 offset:160\n\tds_read_b128 %7, %16 offset:176\n\t"
                 "ds_read_b128 %8, %16 offset:256\n\tds_read_b128 %9, %16 offset:272\n\tds_read_b128 %10, %16 offset:288\n\tds_read_b128 %11, %16 offset:304\n\t"
                 "ds_read_b128 %12, %16 offset:384\n\tds_read_b128 %13, %16 offset:400\n\tds_read_b128 %14, %16 offset:416\n\tds_read_b128 %15, %16 offset:432\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0][0]), "=&v"(r[0][1]), "=&v"(r[0][2]), "=&v"(r[0][3]), "=&v"(r[1][0]), "=&v"(r[1][1]), "=&v"(r[1][2]), "=&v"(r[1][3]),
                   "=&v"(r[2][0]), "=&v"(r[2][1]), "=&v"(r[2][2]), "=&v"(r[2][3]), "=&v"(r[3][0]), "=&v"(r[3][1]), "=&v"(r[3][2]), "=&v"(r[3][3])
                 : "v"(addr)
                 : "memory");
}

// three bf16x8 pieces of 8 fp32 values
struct B3 {
    bf16x8 hi, mid, lo;
};
__device__ __forceinline__ B3 split3(const float (&x)[8])
{
    B3 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)x[j];
        const float r1 = x[j] - (float)a;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        b.hi[j] = a;
        b.mid[j] = m;
        b.lo[j] = (__bf16)r2;
    }
    return b;
}

// The MFMAs of one k-step over NT output tiles, two tiles at a time, each shape one asm block (generated: gen/mlp3_kstep.py).
// FIRST: the layer's first k-step (the accumulators start at 0, whatever they held).
#include "mlp3_kstep.inc"

// The chunk request of a k-step when it is made inside the block (`kstep_asm_*_dma*`): buffer resource of the weight
// stream, this wave's byte offset in it, this wave's LDS byte address, this lane's 16 bytes.
struct Dma {
    i32x4 rsrc;
    uint32_t goff, lds, voff;
};

// One k-step on NT tiles.  FIRST: the layer's first (accumulators start at 0).  SPLIT: also split `xn`, the next k-step's
// activations, into `bn`.  INSIDE: the chunk two k-steps ahead is requested inside the block (`d`), else the caller did.
template <int NT, bool FIRST, bool SPLIT, bool INSIDE>
__device__ __forceinline__ void kstep_mfma(f32x16 (&acc)[NT], uint32_t ab, const B3 &b, const float (&xn)[8], B3 &bn, const Dma &d)
{
    static_assert(NT == 8 || NT == 13 || NT == 5 || NT == 3 || NT == 1, "k-step shapes generated: 8, 13, 5, 3, 1 tiles");
#define KSTEP3_(NTV, PERV)                                                                                                  \
    if constexpr (NT == NTV) {                                                                                              \
        if constexpr (SPLIT && INSIDE) kstep_asm_##NTV##_split_dma##PERV<FIRST>(acc, ab, b, xn, bn, d.rsrc, d.goff, d.lds, d.voff); \
        if constexpr (SPLIT && !INSIDE) kstep_asm_##NTV##_split<FIRST>(acc, ab, b, xn, bn);                                 \
        if constexpr (!SPLIT && INSIDE) kstep_asm_##NTV##_dma##PERV<FIRST>(acc, ab, b, d.rsrc, d.goff, d.lds, d.voff);      \
        if constexpr (!SPLIT && !INSIDE) kstep_asm_##NTV<FIRST>(acc, ab, b);                                                \
    }
    KSTEP3_(8, 6)
    KSTEP3_(13, 10)
    KSTEP3_(5, 4)
    KSTEP3_(3, 3)
    KSTEP3_(1, 1)
#undef KSTEP3_
}

// OT: output tiles held in registers at once (13 for the Zhao-Carr emulator); RES: residual outputs; FAST: every output and
// residual row is 16-byte aligned with a stride that is a multiple of 4 and n_samples % 4 == 0 (the 16-byte epilogue)
template <int OT, bool RES, bool FAST>
__global__ __launch_bounds__(256, 1) void mlp3_kernel(const Mlp3Launch p)
{
    constexpr int CH_H = 3 * kHT * 64;                        // float4 per hidden-type chunk (24 KB)
    constexpr int CH_O = ((3 * OT * 64 + 255) / 256) * 256;   // per output-type chunk, padded to whole rounds of 256 lanes
    constexpr int CH_MAX = (CH_H > CH_O) ? CH_H : CH_O;
    constexpr int PER_H = CH_H / 256, PER_O = CH_O / 256;     // buffer_load ... lds instructions per wave and chunk
    constexpr uint32_t CHB = CH_MAX * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [3 chunk buffers][bias table][centre][eps][row tables of the epilogue]
    float *bias_t = reinterpret_cast<float *>(smem + 3 * (size_t)CHB);     // [(n_hidden 8 + OT)][2][16]
    float *ce = bias_t + (p.n_hidden * kHT + OT) * 32;                     // [n_ks1][2 halves]{centre[8], epsilon[8]}
    // row tables: [OT][2 register blocks][2 halves][8] (the order a lane holds them in) or, FAST, [OT * 32] by feature
    int64_t *orow = reinterpret_cast<int64_t *>(ce + p.n_ks1 * 32);        // output row (the sink where none)
    int64_t *rsrc = orow + OT * 32;                                        // residual `before` row (a readable dummy where none)
    int64_t *rout = rsrc + OT * 32;                                        // residual `after` row (the sink where none)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char *)smem;

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (p.n_hidden * kHT + OT) * 32; i += 256) bias_t[i] = p.bias[i];
    for (int i = tid; i < p.n_ks1 * 16; i += 256) {   // i = (k-step, half, j)
        ce[(i >> 3) * 16 + (i & 7)] = p.center[i];
        ce[(i >> 3) * 16 + 8 + (i & 7)] = p.eps[i];
    }
    for (int i = tid; i < OT * 32; i += 256) {
        // table slot (t, rb, hf, jj) holds feature 32 t + rho3(8 rb + jj) + 4 hf
        const int t = i >> 5, rb = (i >> 4) & 1, hf = (i >> 3) & 1, jj = i & 7;
        const int f = FAST ? i : 32 * t + rho3(8 * rb + jj) + 4 * hf;
        const int of = p.ofeat[f], rs = p.ores[f];
        const int64_t sink = reinterpret_cast<int64_t>(p.sink);
        orow[i] = of < 0 ? sink : reinterpret_cast<int64_t>(p.out[of >> 20]) + (int64_t)(of & 0xFFFFF) * p.out_fs[of >> 20] * 4;
        if (of >= 0 && rs >= 0) {
            const int feat = of & 0xFFFFF;
            rsrc[i] = reinterpret_cast<int64_t>(p.src[rs & 0xFF]) + (int64_t)feat * p.src_fs[rs & 0xFF] * 4;
            rout[i] = reinterpret_cast<int64_t>(p.out[rs >> 8]) + (int64_t)feat * p.out_fs[rs >> 8] * 4;
        } else {
            rsrc[i] = reinterpret_cast<int64_t>(p.src[0]);
            rout[i] = sink;
        }
    }
    if constexpr (FAST) {   // the output layer's biases once more, by feature (the transposed tile adds them after the exchange)
        float *bias_f = reinterpret_cast<float *>(smem + 3 * (size_t)CHB + (p.n_hidden * kHT + OT) * 128 + p.n_ks1 * 128 + 3 * OT * 256 +
                                                  4 * (32 * kPatchRow));
        for (int i = tid; i < OT * 32; i += 256) {
            const int t = i >> 5, fl = i & 31, hf = (fl >> 2) & 1, r = (fl & 3) + 4 * (fl >> 3);   // fl = rho3(r) + 4 hf
            bias_f[i] = p.bias[((p.n_hidden * kHT + t) * 2 + hf) * 16 + r];
        }
    }
    // the small output layer's weights behind everything else: [8 tiles][4 outputs][2 halves][16] floats + 4 biases
    const uint32_t tail_off = 3 * CHB + (p.n_hidden * kHT + OT) * 128 + p.n_ks1 * 128 + 3 * OT * 256 + (FAST ? 4 * (32 * kPatchRow) + OT * 128 : 0);
    if (p.small_out) {
        float *ws = reinterpret_cast<float *>(smem + tail_off);
        for (int i = tid; i < 8 * 4 * 2 * 16 + 4; i += 256) ws[i] = p.wsmall[i];
    }
    __syncthreads();

    // ---- the weight stream: chunk g of a tile's G = n_ks1 + 16 (n_hidden - 1) + 16 chunks; three LDS buffers, the chunk of
    // k-step g + 2 is requested at the start of k-step g and awaited (this wave's share, then the barrier) at the end of g + 1
    const int n_hid_chunks = p.n_ks1 + 16 * (p.n_hidden - 1);
    const int G = n_hid_chunks + (p.has_out ? 16 : 0);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(p.w), 0, p.w_bytes, 0x00020000);
    uint32_t b0 = 0, b1 = CHB, b2 = 2 * CHB;   // LDS byte offsets of the buffers holding chunks g, g + 1, g + 2
    auto chunk_bytes = [&](int g) -> uint32_t {
        return g < n_hid_chunks ? (uint32_t)g * (CH_H * 16) : (uint32_t)n_hid_chunks * (CH_H * 16) + (uint32_t)(g - n_hid_chunks) * (CH_O * 16);
    };
    auto dma = [&](auto per_c, int g, uint32_t buf) {
        constexpr int PER = decltype(per_c)::value;
        const uint32_t goff = chunk_bytes(g) + wave * 1024;
        lds_char *dst = (lds_char *)smem + buf + wave * 1024;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void *)(dst + i * 4096), 16, lane * 16,
                                                     goff + i * 4096, 0, 0);
    };
    // request the chunk two k-steps ahead of k-step g; returns whether it is a hidden-type chunk (its load count)
    // KIND 0 / 1: the caller knows it is a hidden-type / an output-type chunk of this tile; 2: decided here.  (Where the
    // compiler has loads of its own in flight -- layer 1, the residual rows -- the chunk loads must not sit in a branch: its
    // wait counts drop them at the join, and every wait it then places also waits for the loads just issued.)
    const uint32_t w_lo = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(p.w)),
                   w_hi = __builtin_amdgcn_readfirstlane((uint32_t)(reinterpret_cast<uintptr_t>(p.w) >> 32));
    Dma dsc;
    dsc.rsrc = i32x4{(int)w_lo, (int)(w_hi & 0xFFFFu), (int)p.w_bytes, 0x00020000};
    dsc.voff = lane * 16;
    dsc.goff = dsc.lds = 0;
    auto request_ahead = [&](int g, auto kind_c) -> bool {
        constexpr int KIND = decltype(kind_c)::value;
        if constexpr (KIND == 0 || KIND == 1) {   // (made inside the k-step's block: only say where from and where to)
            int g2 = g + 2;   // (past the tile's last chunk: the first chunks of the next tile)
            if (g2 >= G) g2 -= G;
            dsc.goff = __builtin_amdgcn_readfirstlane(chunk_bytes(g2) + wave * 1024);
            dsc.lds = __builtin_amdgcn_readfirstlane(lds0 + b2 + wave * 1024);
            return KIND == 0;
        } else {
            int g2 = g + 2;
            if (g2 >= G) g2 -= G;
            const bool is_h = g2 < n_hid_chunks;
            if (is_h)
                dma(std::integral_constant<int, PER_H>{}, g2, b2);
            else
                dma(std::integral_constant<int, PER_O>{}, g2, b2);
            return is_h;
        }
    };
    // end of a k-step: this wave's share of chunk g + 1 has landed when at most the loads issued after it are in flight
    // (EXTRA other loads + the chunk requested in this k-step); then the barrier, and the buffers rotate
    auto fence = [&](auto extra_c, bool ahead_is_h) {
        constexpr int EXTRA = decltype(extra_c)::value;
        if (ahead_is_h)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(EXTRA + PER_H) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(EXTRA + PER_O) : "memory");
        const uint32_t t_ = b0;
        b0 = b1;
        b1 = b2;
        b2 = t_;
    };
    const uint32_t a_lane = lds0 + lane * 16;
    const uint32_t bias_lane = lds0 + 3 * CHB + half * 64;                         // + tile * 128
    const uint32_t ce_lane = bias_lane + (p.n_hidden * kHT + OT) * 128;            // + ks * 128: {centre[8], epsilon[8]} of this half
    const uint32_t row_addr = ce_lane + p.n_ks1 * 128;                             // orow: + (t * 2 + rb) * 128
    // FAST: this wave's transposition patch behind the tables, this lane's write / read positions in it, its table position
    const uint32_t patch0 = lds0 + 3 * CHB + (p.n_hidden * kHT + OT) * 128 + p.n_ks1 * 128 + 3 * OT * 256 + wave * (32 * kPatchRow);
    const uint32_t patch_w = patch0 + (4 * half) * kPatchRow + col * 4, patch_r = patch0 + (lane >> 3) * kPatchRow + (lane & 7) * 16;
    const uint32_t rowsF = lds0 + 3 * CHB + (p.n_hidden * kHT + OT) * 128 + p.n_ks1 * 128 + (lane >> 3) * 8;   // + table * OT * 256 + t * 256
    const uint32_t biasF = patch0 - wave * (32 * kPatchRow) + 4 * (32 * kPatchRow) + (lane >> 3) * 4;           // + t * 128: output biases by feature

    // the first two chunks of the first tile
    dma(std::integral_constant<int, PER_H>{}, 0, b0);   // (n_hid_chunks >= 1: chunk 0 is always hidden-type)
    if (1 < n_hid_chunks)
        dma(std::integral_constant<int, PER_H>{}, 1, b1);
    else
        dma(std::integral_constant<int, PER_O>{}, 1, b1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

#ifdef MLP3_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0;
#define STAMP3_BEGIN() st_t0 = __builtin_readcyclecounter()
#define STAMP3_END(i)                                               \
    {                                                               \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        st_acc[i] += t_ - st_t0;                                    \
        st_t0 = t_;                                                 \
    }
#else
#define STAMP3_BEGIN() ((void)0)
#define STAMP3_END(i) ((void)0)
#endif
    for (int64_t tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        STAMP3_BEGIN();
        const int64_t n = tile * 128 + wave * 32 + col;
        const bool valid = n < p.n_samples;
        const uint32_t nb = (uint32_t)((valid ? n : p.n_samples - 1) * 4);  // byte offset of this lane's sample inside a row
        int g = 0;
        // ================= layer 1 =================
        // x of k-step ks: 8 buffer loads (rows 8 half + j of the k-step's 16), requested two k-steps ahead
        auto load_x = [&](const XStep &s, float (&x)[8]) {
            // one bounds-checked resource over the k-step's real rows; the row goes into the VECTOR offset (a scalar offset
            // would not be bounds-checked): row 8 half + j of a k-step with fewer rows is out of range and reads 0
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(s.base), 0, s.rows * s.fs4, 0x00020000);
            const uint32_t voff = nb + (half ? 8u * s.fs4 : 0u);
            if (p.x_wrap) {   // (rows of 256 MiB and more: the offsets of the rows past the last real one are forced out of range)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t o = (uint32_t)(8 * half + j) < s.rows ? voff + j * s.fs4 : 0xFFFFFFFFu;
                    x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, MLP3_NT_AUX));
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff + j * s.fs4, 0, MLP3_NT_AUX));
        };
        auto xstep = [&](int ks) { return p.xk[ks < p.n_ks1 ? ks : p.n_ks1 - 1]; };   // (past the end: the last k-step's rows again)
        // normalise (log / centre); `t` = the k-step's table row {centre[8], epsilon[8]} of this half
        auto normalise = [&](int ks, const float (&xr)[8], const f32x4 (&t)[4], float (&x)[8]) {
            if (ks < p.n_log_ks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float e = t[2 + (j >> 2)][j & 3], c = t[j >> 2][j & 3];
                    const float v = xr[j] < e ? e : xr[j];
                    x[j] = __builtin_amdgcn_logf(v) * 0.693147180559945f - c;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = xr[j] - t[j >> 2][j & 3];
            }
        };
        f32x16 h[kHT];
        float xA[8], xB[8], xn[8];
        f32x4 tA[4], tB[4];
        load_x(xstep(0), xA);
        load_x(xstep(1), xB);
        lds_read64_sync(ce_lane, tA[0], tA[1], tA[2], tA[3]);
        normalise(0, xA, tA, xn);
        B3 b = split3(xn);
        load_x(xstep(2), xA);
        XStep xs = xstep(3);   // (fetched one k-step before its loads are issued)
        lds_read64_sync(ce_lane + (p.n_ks1 > 1 ? 128 : 0), tB[0], tB[1], tB[2], tB[3]);
        // k-step ks.  On entry: `b` = its operand; x_in / t_in = the loaded rows and the table row of ks + 1 (the other x
        // buffer holds the rows of ks + 2); t_out is free.  It normalises x of ks + 1 (requested two k-steps ago) -- the split
        // into pieces happens inside the MFMA block, which also requests the chunk of g + 2 and reads the table row of ks + 2
        // into t_out --, then requests x of ks + 3 into x_in.  The order matters: the compiler does not see the chunk loads of
        // the block, so where it waits for "all but the last 8" of its own loads, the chunk loads behind them wait too; with
        // the x loads AFTER the block, what it forces at the next normalise is the chunk requested a whole k-step earlier.
        auto layer1_step = [&](auto first_c, auto kind_c, int ks, float (&x_in)[8], f32x4 (&t_in)[4], f32x4 (&t_out)[4])
                               __attribute__((always_inline)) {
            const int ks2 = ks + 2 < p.n_ks1 ? ks + 2 : p.n_ks1 - 1;   // (past the end: the last row again, unused)
            normalise(ks + 1 < p.n_ks1 ? ks + 1 : ks, x_in, t_in, xn);
            const bool ah = request_ahead(g, kind_c);
            B3 bn;
            if constexpr (decltype(kind_c)::value == 0)
                kstep_asm_8_tab_split_dma6<decltype(first_c)::value>(h, a_lane + b0, b, ce_lane + ks2 * 128, t_out[0], t_out[1], t_out[2], t_out[3], xn,
                                                                     bn, dsc.rsrc, dsc.goff, dsc.lds, dsc.voff);
            else
                kstep_asm_8_tab_split<decltype(first_c)::value>(h, a_lane + b0, b, ce_lane + ks2 * 128, t_out[0], t_out[1], t_out[2], t_out[3], xn, bn);
            load_x(xs, x_in);   // (past the end: the last k-step's rows again -- the load count stays static)
            xs = xstep(ks + 4);
            fence(std::integral_constant<int, 16>{}, ah);   // (behind the chunk of g + 1: the x loads of the last k-step and of this one)
            b = bn;
            ++g;
        };
        STAMP3_END(0);
        // The chunk two k-steps ahead of a layer-1 k-step is hidden-type -- and requested inside the block -- when there are more
        // hidden layers (a later hidden chunk) or no output layer (a first chunk of the next tile).  A single hidden layer in
        // front of an output layer asks for output-type chunks in its last two k-steps: that layer 1 requests through the
        // compiler's builtin throughout (two instantiations of the loop, not two of every step: the registers are full).
        auto layer1 = [&](auto kind_c) __attribute__((always_inline)) {
            layer1_step(std::true_type{}, kind_c, 0, xB, tB, tA);
            int ks = 1;
            for (; ks + 1 < p.n_ks1; ks += 2) {
                layer1_step(std::false_type{}, kind_c, ks, xA, tA, tB);
                layer1_step(std::false_type{}, kind_c, ks + 1, xB, tB, tA);
            }
            if (ks < p.n_ks1) layer1_step(std::false_type{}, kind_c, ks, xA, tA, tB);
        };
        if (p.n_hidden >= 2 || !p.has_out)
            layer1(std::integral_constant<int, 0>{});
        else
            layer1(std::integral_constant<int, 2>{});
        STAMP3_END(1);
        asm volatile("s_nop 15\n\ts_nop 7");   // (the last MFMAs' results, before anything the compiler schedules reads them)
        // bias + ReLU (bias rows: four tiles per LDS round trip)
        auto bias_relu = [&](f32x16 (&dst)[kHT], const f32x16 (&raw)[kHT], int layer) {
            const uint32_t ba = bias_lane + layer * (kHT * 128);
            static_for<kHT / 4>([&](auto gc) {
                constexpr int T4 = decltype(gc)::value * 4;
                f32x4 bq[4][4];
                lds_read4x64_sync(ba + T4 * 128, bq);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = raw[T4 + tt][r] + bq[tt][r >> 2][r & 3];
                        dst[T4 + tt][r] = v < 0.f ? 0.f : v;
                    }
            });
        };
        bias_relu(h, h, 0);
        STAMP3_END(2);
        auto x_of = [&](const f32x16 (&hh)[kHT], auto ks_c, float (&x)[8]) {
            constexpr int KS = decltype(ks_c)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = hh[KS / 2][(KS % 2) * 8 + j];
        };
        auto b_of = [&](const f32x16 (&hh)[kHT], auto ks_c) -> B3 {
            float x[8];
            x_of(hh, ks_c, x);
            return split3(x);
        };
        // ================= hidden -> hidden =================
        for (int l = 1; l < p.n_hidden; ++l) {
            f32x16 h2[kHT];
            b = b_of(h, std::integral_constant<int, 0>{});
            static_for<16>([&](auto ks_c) {
                constexpr int KS = decltype(ks_c)::value;
                const bool ah = request_ahead(g, std::integral_constant<int, (KS < 14 ? 0 : 2)>{});
                B3 bn;
                if constexpr (KS + 1 < 16) x_of(h, std::integral_constant<int, (KS + 1 < 16 ? KS + 1 : 0)>{}, xn);
                kstep_mfma<kHT, KS == 0, (KS + 1 < 16), (KS < 14)>(h2, a_lane + b0, b, xn, bn, dsc);
                STAMP3_END(3);
                fence(std::integral_constant<int, 0>{}, ah);
                STAMP3_END(6);   // (the hidden layers' waits + barriers alone)
                if constexpr (KS + 1 < 16) b = bn;
                ++g;
            });
            asm volatile("s_nop 15\n\ts_nop 7");
            STAMP3_END(3);
            bias_relu(h, h2, l);
            STAMP3_END(2);
        }
        // ================= the hidden output (`hidden_output` of the descriptor: a recurrent cell's new state) =================
        if (p.hout) {
            if constexpr (FAST) {   // through the wave's LDS patch, 16 bytes per lane: 4 stores per 32-feature tile
                const int64_t qh = tile * 128 + wave * 32 + (lane & 7) * 4;
                const bool qv = qh < p.n_samples;
                char *hrow = reinterpret_cast<char *>(p.hout) + ((int64_t)(lane >> 3) * p.hout_fs + (qv ? qh : 0)) * 4;
                static_for<kHT>([&](auto t_c) {
                    constexpr int T = decltype(t_c)::value;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = h[T][r];
                    f32x4 tq[4];
                    patch_exchange_plain(patch_w, patch_r, v, tq);
                    if (qv) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)   // features 32 T + lane / 8 + 8 i
                            *reinterpret_cast<global_f32x4 *>(reinterpret_cast<int64_t>(hrow) + (int64_t)(32 * T + 8 * i) * p.hout_fs * 4) = tq[i];
                    }
                });
            } else if (valid) {
                char *hrow = reinterpret_cast<char *>(p.hout) + (int64_t)nb + (int64_t)(4 * half) * p.hout_fs * 4;
                static_for<kHT>([&](auto t_c) {
                    constexpr int T = decltype(t_c)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        *reinterpret_cast<global_float *>(reinterpret_cast<int64_t>(hrow) + (int64_t)(32 * T + rho3(r)) * p.hout_fs * 4) = h[T][r];
                });
            }
        }
        // ================= a small output layer (<= 4 features, no residuals) on the vector ALU =================
        // 16 k-steps of 6 MFMAs each would cost ~10 000 cycles of barriers and LDS round trips for 1.5 % of the tile's
        // arithmetic; here: 128 x 4 fp32 FMAs per lane on its half of the features, one cross-half exchange, one store per output.
        if (p.small_out) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const uint32_t wl = lds0 + tail_off + half * 64;
            auto two_outputs = [&](auto cp_c) {   // outputs CP and CP + 1 (two at a time: 32 registers of weights beside the 128 of h)
                constexpr int CP = decltype(cp_c)::value;
                static_for<kHT>([&](auto t_c) {
                    constexpr int T = decltype(t_c)::value;
                    f32x4 wq[8];
                    lds_read2x64_sync(wl + T * 512 + CP * 128, wq);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[CP + c] = __builtin_fmaf(h[T][r], wq[c * 4 + (r >> 2)][r & 3], acc[CP + c]);
                });
            };
            two_outputs(std::integral_constant<int, 0>{});
            if (p.small_out > 2) two_outputs(std::integral_constant<int, 2>{});
            const float *bs = reinterpret_cast<const float *>(smem + tail_off) + 8 * 4 * 2 * 16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float other = __shfl_xor(acc[c], 32);   // the other half's 128 features of the same sample
                const float yv = acc[c] + other + bs[c];
                if (c < p.small_out) {
                    const int of = p.ofeat[c];
                    const int64_t row = reinterpret_cast<int64_t>(p.out[of >> 20]) + (int64_t)(of & 0xFFFFF) * p.out_fs[of >> 20] * 4;
                    if (valid && half == 0) *reinterpret_cast<global_float *>(row + (int64_t)nb) = yv;
                }
            }
        }
        // ================= hidden -> outputs =================
        if (p.has_out) {
        f32x16 y[OT];
        b = b_of(h, std::integral_constant<int, 0>{});
        const int64_t nb64 = nb;
        auto row_of = [](const f32x4 (&rr)[8], int r) -> int64_t {   // row r (0..15) of a tile out of its two table rows
            const f32x2 pr = {rr[r >> 1][(r & 1) * 2], rr[r >> 1][(r & 1) * 2 + 1]};
            return __builtin_bit_cast(int64_t, pr);
        };
        // the row tables: 0 / 1 / 2 = output rows, `before` rows, `after` rows; a tile's two register blocks are 128 bytes apart
        auto table_addr = [&](int table, int t) -> uint32_t { return row_addr + table * (OT * 256) + t * 256; };
        // FAST: this lane's quad of samples (4 consecutive, all inside or all outside the batch) of the wave's 32
        const int64_t q0 = tile * 128 + wave * 32 + (lane & 7) * 4;
        const bool qvalid = q0 < p.n_samples;
        const int64_t qb64 = (qvalid ? q0 : 0) * 4;
        auto load_before = [&](int t, float (&dst)[16]) {
            if constexpr (FAST) {
                f32x2 rq[4];
                rows_read(rowsF + OT * 256 + t * 256, rq);   // (the prefetch of the first window; later tiles get their rows from the exchange)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = MLP3_NT_LOAD(reinterpret_cast<const global_f32x4 *>(__builtin_bit_cast(int64_t, rq[i]) + qb64));
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[i * 4 + e] = v[e];
                }
            } else {
                f32x4 rr[8];
                lds_read2x64_sync(table_addr(1, t), rr);
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[r] = MLP3_NT_LOAD(reinterpret_cast<const global_float *>(row_of(rr, r) + nb64));
            }
        };
        // `before` rows of the residual outputs: a rolling window of WIN tiles; the first WIN are requested during the last
        // WIN k-steps of this layer (most hidden activations are dead by then), the others as the epilogue frees the window
        constexpr int WIN = RES ? (OT < 5 ? OT : 5) : 1;
        float before[WIN][16];
        static_for<16>([&](auto ks_c) {
            constexpr int KS = decltype(ks_c)::value;
            constexpr bool PRE = RES && KS >= 16 - WIN;
            if constexpr (PRE) load_before(KS - (16 - WIN), before[PRE ? KS - (16 - WIN) : 0]);
            const bool ah = request_ahead(g, std::integral_constant<int, (KS < 14 ? 1 : 2)>{});
            B3 bn;
            if constexpr (KS + 1 < 16) x_of(h, std::integral_constant<int, (KS + 1 < 16 ? KS + 1 : 0)>{}, xn);
            kstep_mfma<OT, KS == 0, (KS + 1 < 16), (KS < 14)>(y, a_lane + b0, b, xn, bn, dsc);
            if (KS == 0 && FAST && p.hout)   // (the 32 stores of the hidden output sit between the last chunk request and this one)
                fence(std::integral_constant<int, (PRE ? 4 : 0) + (KS == 0 ? 32 : 0)>{}, ah);
            else
                fence(std::integral_constant<int, PRE ? (FAST ? 4 : 16) : 0>{}, ah);
            if constexpr (KS + 1 < 16) b = bn;
            ++g;
        });
        asm volatile("s_nop 15\n\ts_nop 7");
        STAMP3_END(4);
        // ================= epilogue: bias, stores =================
        // (the next tile's first chunks are in flight: row tables and biases by hand-placed LDS reads here too)
        if constexpr (FAST) {
            // 16 bytes per lane: at most 63 memory operations of a wave are in flight, so the 4-byte epilogue below is bound by
            // their latency; here a tile is 4 stores (+ 4 loads and 4 stores for a residual output) instead of 16 (+ 32)
            static_for<OT>([&](auto t_c) {
                constexpr int T = decltype(t_c)::value;
                constexpr int TN = T + WIN < OT ? T + WIN : T;   // the tile whose `before` rows this step requests
                float v[16], bf[4];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = y[T][r];
                f32x4 tq[4];
                f32x2 rq[4], ra[4], rn[4];
                if constexpr (RES)
                    patch_exchange_res(patch_w, patch_r, rowsF + T * 256, biasF + T * 128, rowsF + 2 * OT * 256 + T * 256, rowsF + OT * 256 + TN * 256, v, tq,
                                       rq, bf, ra, rn);
                else
                    patch_exchange(patch_w, patch_r, rowsF + T * 256, biasF + T * 128, v, tq, rq, bf);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) tq[i][e] += bf[i];
                if (qvalid) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) MLP3_NT_STORE(tq[i], reinterpret_cast<global_f32x4 *>(__builtin_bit_cast(int64_t, rq[i]) + qb64));
                    if constexpr (RES) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = before[T % WIN][i * 4 + e] + tq[i][e];
                            MLP3_NT_STORE(o, reinterpret_cast<global_f32x4 *>(__builtin_bit_cast(int64_t, ra[i]) + qb64));
                        }
                    }
                }
                if constexpr (RES && T + WIN < OT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 nv = MLP3_NT_LOAD(reinterpret_cast<const global_f32x4 *>(__builtin_bit_cast(int64_t, rn[i]) + qb64));
#pragma unroll
                        for (int e = 0; e < 4; ++e) before[T % WIN][i * 4 + e] = nv[e];
                    }
                }
            });
        } else if (valid) {   // (lanes past the last sample store nothing; they took part in everything above with the last sample's inputs)
            // the direct outputs: y + bias (kept in y for the residual outputs)
            static_for<OT>([&](auto t_c) {
                constexpr int T = decltype(t_c)::value;
                f32x4 bq[4], rr[8];
                lds_tile_tables_sync(bias_lane + (p.n_hidden * kHT + T) * 128, table_addr(0, T), bq, rr);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    y[T][r] += bq[r >> 2][r & 3];
                    MLP3_NT_STORE(y[T][r], reinterpret_cast<global_float *>(row_of(rr, r) + nb64));
                }
            });
            if constexpr (RES)
                static_for<OT>([&](auto t_c) {
                    constexpr int T = decltype(t_c)::value;
                    f32x4 rr[8];
                    lds_read2x64_sync(table_addr(2, T), rr);
#pragma unroll
                    for (int r = 0; r < 16; ++r) MLP3_NT_STORE(before[T % WIN][r] + y[T][r], reinterpret_cast<global_float *>(row_of(rr, r) + nb64));
                    if constexpr (T + WIN < OT) load_before(T + WIN, before[T % WIN]);
                });
        }
        }   // (has_out)
        STAMP3_END(5);
    }
    // (chunks requested for a tile this workgroup does not have are still in flight towards its LDS)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MLP3_STAMPS
    if (p.stamps && lane == 0) {
        unsigned long long *o = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = st_acc[i];
    }
#endif
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
    int n_sources = 0, n_outputs = 0, n_residual = 0, n_hidden = 0, n_ks1 = 0, n_log_ks = 0, n_ot = 0, has_out = 1, hout = 0, small_out = 0;
    void *d_wsmall = nullptr;
    int64_t flops = 0;
    void *d_w = nullptr, *d_bias = nullptr, *d_center = nullptr, *d_eps = nullptr, *d_ofeat = nullptr, *d_ores = nullptr;
    size_t lds_bytes = 0, lds_fast = 0, w_bytes = 0;
    void *d_sink = nullptr;
    int64_t sink_samples = 0;
    // per layer-1 k-step: the source it reads, its first feature row there and how many real rows follow (<= 16)
    std::vector<int> ks_src, ks_feat0, ks_rows, res_source;
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

#ifdef MLP3_STAMPS
static unsigned long long *g_mlp3_stamps = nullptr;
extern "C" void fv3hip_diag_set_mlp3_stamps(void *p) { g_mlp3_stamps = static_cast<unsigned long long *>(p); }
#endif

extern "C" int fv3hip_mlp3_destroy(fv3hip_mlp3_t m)
{
    if (!m) return FV3HIP_OK;
    for (void *q : {m->d_w, m->d_bias, m->d_center, m->d_eps, m->d_ofeat, m->d_ores, m->d_wsmall})
        if (q) hipFree(q);
    if (m->d_sink) hipFree(m->d_sink);
    delete m;
    return FV3HIP_OK;
}

extern "C" int64_t fv3hip_mlp3_flops_per_sample(fv3hip_mlp3_t m) { return m ? m->flops : 0; }

extern "C" int fv3hip_mlp3_create(const fv3hip_mlp_desc_t *d, fv3hip_mlp3_t *out)
{
    FV3HIP_REQUIRE(d && out, "null pointer");
    *out = nullptr;
    const int hout = d->hidden_output ? 1 : 0;
    FV3HIP_REQUIRE(d->n_sources >= 1 && d->n_sources <= kMaxSrc && d->n_inputs >= 1, "bad counts");
    FV3HIP_REQUIRE(d->n_outputs >= 1 || (d->n_outputs == 0 && hout), "n_outputs must be >= 1 (or 0 with hidden_output)");
    FV3HIP_REQUIRE(d->n_outputs + d->n_residual + hout <= kMaxOut, "too many outputs");
    if (d->width != 256 || d->n_hidden < 1 || d->hidden_activation != FV3HIP_ACT_RELU)
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel takes ReLU networks of hidden width 256");
    if (d->out_min || d->out_max || d->out_mask)
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel does not implement output limits / masks");
    const int W = 256;
    int K = 0, F = 0;
    for (int i = 0; i < d->n_inputs; ++i) K += d->in_nfeat[i];
    for (int j = 0; j < d->n_outputs; ++j) F += d->out_nfeat[j];
    const int small_out = (F >= 1 && F <= 4 && d->n_residual == 0) ? F : 0;   // (output layer on the vector ALU, see the kernel)
    const int has_out = (F > 0 && !small_out) ? 1 : 0;
    const int n_ot = has_out ? (F + 31) / 32 : 1;   // (no output layer on the matrix cores: the one-tile instantiation, that phase skipped)
    if (n_ot != 13 && n_ot != 3 && n_ot != 5 && n_ot != 1)
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel is compiled for 1, 3, 5 or 13 output tiles of 32 (got %d outputs)", F);
    // k-slots of layer 1: every input padded to whole k-steps of 16 (a k-step reads 16 consecutive rows of one source), the
    // log-transformed inputs first
    struct Slot { int src, feat; float center, rscale, eps; int orig; };
    std::vector<Slot> slots;
    std::vector<int> ks_src, ks_feat0, ks_rows;
    int n_log_slots = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int k = 0;
        for (int i = 0; i < d->n_inputs; ++i) {
            const bool is_log = d->in_transform && d->in_transform[i] == FV3HIP_TRANSFORM_LOG;
            const float eps = d->in_eps ? d->in_eps[i] : 0.f;
            if (is_log && !(eps >= FLT_MIN))
                return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel needs log epsilons >= FLT_MIN (the fast logarithm)");
            if (is_log == (pass == 0)) {
                for (int f = 0; f < d->in_nfeat[i]; ++f)
                    slots.push_back(Slot{d->in_source[i], d->in_feat_start[i] + f, d->in_center ? d->in_center[k + f] : 0.f,
                                         d->in_scale ? (float)(1.0 / (double)d->in_scale[k + f]) : 1.f, eps, k + f});
                while (slots.size() % 16) slots.push_back(Slot{-1, 0, 0.f, 0.f, 1.f, -1});
                for (int f0 = 0; f0 < d->in_nfeat[i]; f0 += 16) {
                    ks_src.push_back(d->in_source[i]);
                    ks_feat0.push_back(d->in_feat_start[i] + f0);
                    ks_rows.push_back(d->in_nfeat[i] - f0 < 16 ? d->in_nfeat[i] - f0 : 16);
                }
            }
            k += d->in_nfeat[i];
        }
        if (pass == 0) n_log_slots = (int)slots.size();
    }
    if (slots.size() / 16 > (size_t)kMaxKs1)
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel takes at most %d layer-1 k-steps of 16 input features (got %zu)", kMaxKs1,
                    slots.size() / 16);
    {
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
        m->has_out = has_out;
        m->small_out = small_out;
        m->hout = hout;
        m->flops = 2 * ((int64_t)K * W + (int64_t)(d->n_hidden - 1) * W * W + (int64_t)W * F);
        m->ks_src = ks_src;
        m->ks_feat0 = ks_feat0;
        m->ks_rows = ks_rows;
        for (int r = 0; r < d->n_residual; ++r) m->res_source.push_back(d->res_source[r]);
        *out = m;
    }
    fv3hip_mlp3 *m = *out;
    const int n_ks1 = m->n_ks1;
    const int CH_H = 3 * kHT * 64, CH_O = ((3 * n_ot * 64 + 255) / 256) * 256;
    const int n_hid_chunks = n_ks1 + 16 * (d->n_hidden - 1);
    if (n_hid_chunks + (has_out ? 16 : 0) < 2)
        return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel needs at least two k-steps per tile");
    std::vector<unsigned short> w(((size_t)n_hid_chunks * CH_H + (size_t)(has_out ? 16 : 0) * CH_O) * 8, 0);
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
    for (int ks = 0; ks < (has_out ? 16 : 0); ++ks)
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
                for (int hf = 0; hf < 2; ++hf) bias[(((size_t)l * kHT + t) * 2 + hf) * 16 + r] = d->hidden_biases[l][32 * t + rho3(r) + 4 * hf];
    for (int t = 0; t < n_ot; ++t)
        for (int r = 0; r < 16; ++r)
            for (int hf = 0; hf < 2; ++hf) {
                const int f = 32 * t + rho3(r) + 4 * hf;
                if (f < F)
                    bias[(((size_t)d->n_hidden * kHT + t) * 2 + hf) * 16 + r] =
                        (float)((double)d->out_bias[f] * (d->out_scale ? d->out_scale[f] : 1.f) + (d->out_center ? d->out_center[f] : 0.f));
            }
    std::vector<float> center(slots.size()), eps(slots.size());
    for (size_t i = 0; i < slots.size(); ++i) {
        center[i] = slots[i].orig < 0 ? 0.f : slots[i].center;
        eps[i] = slots[i].eps;
    }
    // (a padding slot reads 0 (out of the bounds of its buffer resource); in a log k-step: max(0, eps = 1) = 1 -> log 1 = 0,
    // centre 0: exactly 0 times a zero weight)
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
    std::vector<float> wsmall;
    if (small_out) {   // [tile][output c][half][register r] <- out_kernel[feature 32 t + rho3(r) + 4 half][c] * scale_c; then the biases
        wsmall.assign(8 * 4 * 2 * 16 + 4, 0.f);
        for (int t = 0; t < kHT; ++t)
            for (int c = 0; c < F; ++c)
                for (int hf = 0; hf < 2; ++hf)
                    for (int r = 0; r < 16; ++r)
                        wsmall[((t * 4 + c) * 2 + hf) * 16 + r] =
                            d->out_kernel[(size_t)(32 * t + rho3(r) + 4 * hf) * F + c] * (d->out_scale ? d->out_scale[c] : 1.f);
        for (int c = 0; c < F; ++c)
            wsmall[8 * 4 * 2 * 16 + c] = (float)((double)d->out_bias[c] * (d->out_scale ? d->out_scale[c] : 1.f) + (d->out_center ? d->out_center[c] : 0.f));
    }
    int rc;
    if ((rc = upload3(wsmall, &m->d_wsmall)) || (rc = upload3(w, &m->d_w)) || (rc = upload3(bias, &m->d_bias)) || (rc = upload3(center, &m->d_center)) ||
        (rc = upload3(eps, &m->d_eps)) || (rc = upload3(ofeat, &m->d_ofeat)) || (rc = upload3(ores, &m->d_ores))) {
        fv3hip_mlp3_destroy(m);
        *out = nullptr;
        return rc;
    }
    const size_t ch_max = (size_t)((CH_H > CH_O) ? CH_H : CH_O);
    m->w_bytes = w.size() * sizeof(unsigned short);
    m->lds_bytes = 3 * ch_max * 16 + (size_t)(d->n_hidden * kHT + n_ot) * 128 + (size_t)slots.size() * 8 + (size_t)n_ot * 32 * 24;
    m->lds_fast = m->lds_bytes + 4 * 32 * 36 * 4 + (size_t)n_ot * 128;   // + a transposition patch per wave and the output biases by feature
    if (small_out) {   // + the small output layer's table (behind everything else in both layouts)
        m->lds_bytes += 8 * 4 * 2 * 16 * 4 + 16;
        m->lds_fast += 8 * 4 * 2 * 16 * 4 + 16;
    }
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
    lp.has_out = m->has_out;
    lp.small_out = m->small_out;
    lp.wsmall = static_cast<const float *>(m->d_wsmall);
    if (m->hout) {   // (the hidden output is the last entry of `outputs`, as for fv3hip_mlp_predict)
        const int j = m->n_outputs + m->n_residual;
        FV3HIP_REQUIRE(outputs[j], "the hidden output (output %d) is null", j);
        FV3HIP_REQUIRE(out_feat_stride[j] >= n_samples, "hidden output: feature stride %lld < n_samples", (long long)out_feat_stride[j]);
        lp.hout = static_cast<float *>(outputs[j]);
        lp.hout_fs = out_feat_stride[j];
    }
    lp.w = static_cast<const f32x4 *>(m->d_w);
    lp.bias = static_cast<const float *>(m->d_bias);
    lp.center = static_cast<const float *>(m->d_center);
    lp.eps = static_cast<const float *>(m->d_eps);
    if (n_samples > m->sink_samples) {   // (grow-only; a first call or a larger batch than any before)
        if (m->d_sink) FV3HIP_CHECK_HIP(hipFree(m->d_sink));
        m->d_sink = nullptr;
        m->sink_samples = 0;
        FV3HIP_CHECK_HIP(hipMalloc(&m->d_sink, (size_t)n_samples * 4));
        m->sink_samples = n_samples;
    }
    lp.sink = static_cast<float *>(m->d_sink);
    lp.w_bytes = (uint32_t)m->w_bytes;
#ifdef MLP3_STAMPS
    lp.stamps = g_mlp3_stamps;
#endif
    for (int ks = 0; ks < m->n_ks1; ++ks) {
        const int sidx = m->ks_src[ks];
        const int64_t fs4 = src_feat_stride[sidx] * 4;
        FV3HIP_REQUIRE(fs4 >= n_samples * 4, "source %d: feature stride %lld < n_samples", sidx, (long long)src_feat_stride[sidx]);
        if (fs4 * m->ks_rows[ks] >= ((int64_t)1 << 32))
            return fail(FV3HIP_EUNSUPPORTED, "the split-bf16 kernel addresses the %d feature rows of a k-step with 32-bit offsets (feature stride %lld too large)",
                        m->ks_rows[ks], (long long)src_feat_stride[sidx]);
        if (fs4 * 16 >= ((int64_t)1 << 32)) lp.x_wrap = 1;
        lp.xk[ks].base = reinterpret_cast<uint64_t>(sources[sidx]) + (uint64_t)m->ks_feat0[ks] * (uint64_t)fs4;
        lp.xk[ks].rows = (uint32_t)m->ks_rows[ks];
        lp.xk[ks].fs4 = (uint32_t)fs4;
    }
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
    // the 16-byte epilogue: every output row and every residual `before` row 16-byte aligned at every feature, whole quads
    bool fast = n_samples % 4 == 0 && m->lds_fast <= 160 * 1024;
    for (int j = 0; j < m->n_outputs + m->n_residual && fast; ++j)
        fast = reinterpret_cast<uintptr_t>(outputs[j]) % 16 == 0 && out_feat_stride[j] % 4 == 0;
    for (int r = 0; r < m->n_residual && fast; ++r)
        fast = reinterpret_cast<uintptr_t>(sources[m->res_source[r]]) % 16 == 0 && src_feat_stride[m->res_source[r]] % 4 == 0;
    if (fast && lp.hout) fast = reinterpret_cast<uintptr_t>(lp.hout) % 16 == 0 && lp.hout_fs % 4 == 0;
    const size_t lds = fast ? m->lds_fast : m->lds_bytes;
#define LAUNCH3_(OT)                                                                                                     \
    if (m->n_residual) {                                                                                                 \
        if (fast) LAUNCH3R_(OT, true, true) else LAUNCH3R_(OT, true, false)                                              \
    } else {                                                                                                             \
        if (fast) LAUNCH3R_(OT, false, true) else LAUNCH3R_(OT, false, false)                                            \
    }
    // The kernel's chunk waits count memory operations (s_waitcnt vmcnt(N)); a register spill is a memory operation the
    // count does not know -- an instantiation that needs scratch memory must not run.
#define LAUNCH3R_(OT, RES, FAST)                                                                                         \
    {                                                                                                                    \
        auto kern = mlp3_kernel<OT, RES, FAST>;                                                                          \
        hipFuncAttributes attr;                                                                                          \
        FV3HIP_CHECK_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kern)));                              \
        if (attr.localSizeBytes != 0)                                                                                    \
            return fail(FV3HIP_EUNSUPPORTED, "mlp3_kernel<%d,%d,%d> was built with %zu bytes of scratch per lane (register spills)", OT, \
                        (int)RES, (int)FAST, (size_t)attr.localSizeBytes);                                               \
        FV3HIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, lp);                                                    \
    }
    if (m->n_ot == 13) LAUNCH3_(13) else if (m->n_ot == 5) LAUNCH3_(5) else if (m->n_ot == 3) LAUNCH3_(3) else LAUNCH3_(1)
#undef LAUNCH3_
#undef LAUNCH3R_
    return check_launch("mlp3_kernel");
}
