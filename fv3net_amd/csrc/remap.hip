// mappm for the configuration every caller of the reference passes (kord <= 3: the standard PPM monotonicity
// constraint on all levels, external/vcm/vcm/cubedsphere/regridz.py:227-228; any iv) on the model's native
// [batch][level][column] layout -- the "sweep" kernel.
//
// Reference restated: external/mappm/mappm/mappm.f90:10-126 (mappm), 614-851 (ppm_profile), 854-931 (ppm_limiters).
//
// Same algorithm as mappm_merge_multi_kernel (vertical.hip): one lane per column, the wave marches over the SOURCE
// layers together with the PPM reconstruction in a sliding register window, each lane emits the target layers that end
// inside the current source layer.  What is different -- the r01 counters showed the merge kernels to be bound by VALU
// issue (30 K instructions per column, two thirds of them not division):
//   * one wavefront per workgroup and n_inner % 64 == 0, so a wave's 64 columns share their batch: every row address is
//     an SGPR base (bumped on the scalar unit per level) plus one loop-invariant VGPR lane offset -- no per-access 64-bit
//     VALU address arithmetic;
//   * the LDS ring of target interfaces is filled uniformly (scalar bookkeeping, rows follow lane 0's cursor) instead of
//     per lane; a lane outside the ring's window reads memory;
//   * kord, the limiter and the a6 recomputation are compile-time facts (lmt = 0 everywhere), the limiter is select-only;
//   * every pressure-only subexpression (d4, the five denominators of a level) is formed once per level and shared by
//     dc(k), al(k) and all NF fields; d4(k+1) and its reciprocal are carried to the next level;
//   * with 2 or 4 fields per launch the per-field arithmetic runs two fields to an instruction (v_pk_add_f32 / v_pk_mul_f32
//     on float2 register pairs);
//   * ARITHMETIC MODES.  EXACT: IEEE division and the Fortran's association order -> bit-identical to the compiled
//     reference (tests/test_gpu_vertical.py).  FAST: a / b = a * v_rcp_f32(b) (1 ulp) with shared reciprocals; the result
//     differs from EXACT by a few ulp of the layer's values (<= 1e-5 relative to the column's range is asserted in the
//     tests; BASELINE.json north_star asks for 1e-5, and the reference itself is not bit-reproducible across platforms,
//     external/vcm/tests/test_coarsen_restarts.py:119-123).
// Columns with NaN or non-monotone pressures go on the worklist and are redone by mappm_fallback_kernel, as before.
#include <type_traits>

#include "common.h"
#include "remap.h"

namespace fv3hip {
namespace {

// Per-field quantities are held W fields to a register tuple: W = 1 (float) or W = 2 (two fields in an ext_vector float2, whose
// adds and multiplies the compiler emits as v_pk_add_f32 / v_pk_mul_f32 -- two fields per VALU instruction; compares and
// selects stay per component).  Every operation is component-wise IEEE, so a packed field is bit-identical to a lone one.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
template <int W> struct FieldVec;
template <> struct FieldVec<1> { using type = float; };
template <> struct FieldVec<2> { using type = f32x2; };

__device__ __forceinline__ float v_sel(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ f32x2 v_sel(i32x2 c, f32x2 a, f32x2 b) { return c ? a : b; }
__device__ __forceinline__ float v_min2(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ f32x2 v_min2(f32x2 a, f32x2 b) { return (a < b) ? a : b; }
__device__ __forceinline__ float v_max2(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ f32x2 v_max2(f32x2 a, f32x2 b) { return (a > b) ? a : b; }
template <typename V> __device__ __forceinline__ V v_min3(V a, V b, V c) { return v_min2(v_min2(a, b), c); }
template <typename V> __device__ __forceinline__ V v_max3(V a, V b, V c) { return v_max2(v_max2(a, b), c); }
__device__ __forceinline__ float v_abs(float a) { return fabsf(a); }
__device__ __forceinline__ f32x2 v_abs(f32x2 a) { return f32x2{fabsf(a[0]), fabsf(a[1])}; }
__device__ __forceinline__ float v_sign(float a, float b) { return copysignf(fabsf(a), b); }
__device__ __forceinline__ f32x2 v_sign(f32x2 a, f32x2 b) { return f32x2{copysignf(fabsf(a[0]), b[0]), copysignf(fabsf(a[1]), b[1])}; }
__device__ __forceinline__ float v_splat(float, float x) { return x; }
__device__ __forceinline__ f32x2 v_splat(f32x2, float x) { return f32x2{x, x}; }
__device__ __forceinline__ float v_get(float a, int) { return a; }
__device__ __forceinline__ float v_get(f32x2 a, int i) { return a[i]; }
__device__ __forceinline__ void v_set(float &a, int, float x) { a = x; }
__device__ __forceinline__ void v_set(f32x2 &a, int i, float x) { a[i] = x; }

// FAST-mode primitives.  The exact mode keeps the reference's compare-and-select minima (a NaN operand picks the second
// argument, as the compiled Fortran does) and never contracts; the fast mode takes the hardware's v_min3 / v_max3 (one
// 4-cycle instruction instead of two compares and two selects, 12 cycles -- measured issue costs in
// benchmarks/valu_ubench) and fused multiply-adds (2 cycles for a multiply and an add).
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_min3_abs(float a, float b, float c) { float r; asm("v_min3_f32 %0, |%1|, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x2 hw_min3(f32x2 a, f32x2 b, f32x2 c) { return f32x2{hw_min3(a[0], b[0], c[0]), hw_min3(a[1], b[1], c[1])}; }
__device__ __forceinline__ f32x2 hw_max3(f32x2 a, f32x2 b, f32x2 c) { return f32x2{hw_max3(a[0], b[0], c[0]), hw_max3(a[1], b[1], c[1])}; }
__device__ __forceinline__ f32x2 hw_min3_abs(f32x2 a, f32x2 b, f32x2 c) { return f32x2{hw_min3_abs(a[0], b[0], c[0]), hw_min3_abs(a[1], b[1], c[1])}; }
__device__ __forceinline__ float v_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f32x2 v_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 v_fma(float a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(f32x2{a, a}, b, c); }
__device__ __forceinline__ f32x2 v_fma(f32x2 a, float b, f32x2 c) { return __builtin_elementwise_fma(a, f32x2{b, b}, c); }
__device__ __forceinline__ auto v_or(bool a, bool b) { return a | b; }
__device__ __forceinline__ i32x2 v_or(i32x2 a, i32x2 b) { return a | b; }

// A denominator (always a pressure-only scalar): EXACT keeps the value and divides (IEEE) at every use, FAST takes the
// reciprocal once.
template <bool FAST>
struct Den {
    float v;
    __device__ __forceinline__ explicit Den(float d) : v(FAST ? __builtin_amdgcn_rcpf(d) : d) {}
    __device__ __forceinline__ float under(float num) const { return FAST ? num * v : num / v; }
    __device__ __forceinline__ f32x2 under(f32x2 num) const { return FAST ? num * v : num / v; }
};

// mappm.f90:875-894 (lmt = 0), select-only
template <typename V>
__device__ __forceinline__ void limit0(V dm, V q, V &al, V &ar, V &a6)
{
    const V da1 = ar - al;
    const V da2 = da1 * da1;
    const V a6da = a6 * da1;
    const auto lo = a6da < -da2, hi = a6da > da2, flat = (dm == v_splat(dm, 0.f));
    const V a6_lo = 3.f * (al - q), a6_hi = 3.f * (ar - q);
    const V ar_lo = al - a6_lo, al_hi = ar - a6_hi;
    const V nal = v_sel(lo, al, v_sel(hi, al_hi, al));
    const V nar = v_sel(lo, ar_lo, ar);
    const V na6 = v_sel(lo, a6_lo, v_sel(hi, a6_hi, a6));
    al = v_sel(flat, q, nal);
    ar = v_sel(flat, q, nar);
    a6 = v_sel(flat, v_splat(q, 0.f), na6);
}

// The same constraint for the fast mode: flatten first (al = ar = q makes a6 = 0 and both tests false), then ONE kept edge
// e -- the left one where the parabola undershoots, the right one where it overshoots -- gives a6 = 3 (e - q) and the other
// edge e - a6: 3 compares, 6 selects, 11 arithmetic instructions per field.  Also returns da1 = ar - al for the emit code.
template <typename V>
__device__ __forceinline__ void limit0_fast(V dm, V q, V &al, V &ar, V &a6, V &da1)
{
    const auto flat = (dm == v_splat(dm, 0.f));
    al = v_sel(flat, q, al);
    ar = v_sel(flat, q, ar);
    da1 = ar - al;
    a6 = 3.f * v_fma(v_splat(q, 2.f), q, -(al + ar));
    const V da2 = da1 * da1, a6da = a6 * da1;
    const auto lo = a6da < -da2, hi = a6da > da2;
    const V e = v_sel(lo, al, ar);
    const V a6n = 3.f * (e - q);
    const V oth = e - a6n;
    a6 = v_sel(v_or(lo, hi), a6n, a6);
    ar = v_sel(lo, oth, ar);
    al = v_sel(hi, oth, al);
    da1 = ar - al;
}

template <typename Tin>
__device__ __forceinline__ float ld(const char *base, unsigned int boff)
{
    return (float)*reinterpret_cast<const Tin *>(base + boff);
}

// Dynamic LDS of a wave: [target interfaces: kRing rows x 64 lanes, or (TGT) the whole (kn + 1) x 8 table][result ring]
extern __shared__ float sweep_lds[];
constexpr int kSweepRing = 16;
// Result rows held per field (see the result ring below).  8 everywhere but in the instantiation the float64 restart pipelines
// launch in the fast arithmetic: on BASELINE configs[2]'s iid data (lanes up to 16 rows apart) 12 rows cut its partial
// stores enough to win 11 % (1.16 -> 1.04 ms) although a wave then holds 14.5 KB of LDS; every other instantiation lost
// (occupancy, or the modulo by 12), 16 rows lost everywhere.
template <typename Tin, int NF, bool FAST> constexpr int sweep_out_rows() { return (sizeof(Tin) == 8 && NF == 4 && FAST) ? 12 : 8; }

// ---- MEAN: the fused remap + masked 8 x 8 block mean (fv3hip_mappm_block_mean) ----
// What the restart pipelines do with a remapped field is average it at once over the 8 x 8 blocks of the coarse grid, with the
// area masked where the coarse layer lies below the fine surface (regridz.py:149-220, coarsen_restarts.py:483-495, 940-961).
// With a wave's 64 lanes on ONE block the target grid is the wave's own coarse column, and a finished target row need not
// leave the chip: lane l parks p = q2 * w (w = area_l or 0) in its column of an LDS ring, and once every lane is past four
// rows the wave sums them -- 32 lanes, each over one half-row of one (row, field) in the order of wavg_block_kernel /
// mass_wavg_block_kernel (coarsen.hip: dy = 0..7, x = 0..3 | 4..7, then the two halves), so the result is bit-identical to
// mask_weights + weighted_block_average of the unfused route -- divides by the row's sum of weights and writes NF x 4 floats.
// No 4-byte scatter of result rows (what the sweep kernel loses a fifth of its time to on BASELINE configs[2]'s data, see
// DESIGN 4.3b), no fine-size q2 written and read back, no masked-weights array.
// A ring line is one (row, field): 64 values in half-major order (pos = (x >> 2) * 32 + y * 4 + (x & 3)) so that a summing lane
// reads its 32 values as eight ds_read_b128; lines are 68 floats apart, which spreads the 16 lines of a group over the banks.
// A lane that runs kOut rows ahead of the slowest evicts its own oldest value to the fine-size scratch q2[] and the flush
// brings it back -- same value, same sum.  A block with an ill-formed column is listed and redone (mean_redo_kernel).
constexpr int kMeanStride = 68;
constexpr int kMeanGroup = 4;   // rows per flush
#ifndef MEAN_KOUT
#define MEAN_KOUT 8
#endif
template <typename Tin, int NF, bool FAST> constexpr int mean_out_rows() { return MEAN_KOUT; }   // a multiple of kMeanGroup
__host__ __device__ constexpr int mean_n1(int kn) { return (kn + 2 + 3) & ~3; }   // kn + 1 interfaces and one spare word (the spill flag)
__host__ __device__ constexpr int mean_nl(int kn, int esz) { return ((kn * esz + 15) & ~15) / 4; }
// LDS of a MEAN wave, in floats: [target interfaces: n1][compared levels: kn of Tin][row weight sums: n1][ring]
__host__ __device__ constexpr int mean_tab_floats(int kn, int esz) { return 2 * mean_n1(kn) + mean_nl(kn, esz); }

__device__ __forceinline__ int mean_pos(int lane) { return ((lane & 4) << 3) | ((lane >> 3) << 2) | (lane & 3); }
__device__ __forceinline__ bool f_isnan(float x) { return x != x; }

// The tables of one block: compared levels -> lvl[0..kn), the block's row weight sums -> den[0..kn).  `tmp` (64 floats + 64 Tin,
// the ring's first bytes) holds the lanes' areas and surface pressures in half-major order while the sums are formed.
template <typename Tin>
__device__ __forceinline__ void mean_tables(const char *lvl_col, unsigned int row_p2, int cmp_offset, int kn, int lane, float area_l,
                                            Tin ps_raw, Tin *lvl, float *den, float *tmp)
{
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = lane + 64 * j;
        const Tin x = *reinterpret_cast<const Tin *>(lvl_col + (size_t)((i < kn ? i : kn - 1) + cmp_offset) * row_p2);
        if (i < kn) lvl[i] = x;
    }
    float *tA = tmp;
    Tin *tP = reinterpret_cast<Tin *>(tmp + 64);
    const int pos = mean_pos(lane);
    tA[pos] = area_l;
    tP[pos] = ps_raw;
    const int half = lane & 1;
    for (int r0 = 0; r0 < kn; r0 += 32) {
        const int r = r0 + (lane >> 1);
        const Tin lv = lvl[r < kn ? r : kn - 1];
        float s = 0.f;
#pragma unroll 8
        for (int t = 0; t < 32; ++t) {
            const float w0 = (lv < tP[half * 32 + t]) ? tA[half * 32 + t] : 0.f;   // regridz.py:209-220
            s += f_isnan(w0) ? 0.f : w0;
        }
        s += __shfl_xor(s, 1);
        if (r < kn && half == 0) den[r] = s;
    }
}

// Rows [row0, row0 + nrows) (nrows <= kMeanGroup) of NF fields, parked in the lines (slot0 + i) * NF + f of `ring`: their masked
// block means to mean[f][row * plane2].
template <int NF>
__device__ __forceinline__ void mean_reduce(const float *ring, int slot0, int row0, int nrows, const float *den, float *const (&mean)[4],
                                            int64_t plane2, int lane)
{
    const int pair = lane >> 1, half = lane & 1, ri = pair / NF, f = pair - ri * NF;
    const bool act = ri < nrows;   // (ri < kMeanGroup follows: nrows <= kMeanGroup)
    const float *src = ring + ((slot0 + (act ? ri : 0)) * NF + f) * kMeanStride + half * 32;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef MEAN_RBATCH
#define MEAN_RBATCH 4
#endif
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < 8 / MEAN_RBATCH; ++h) {  // (batches of reads: 16 registers in flight, not 32 -- the call sits inside the sweep's loop)
        f32x4 v[MEAN_RBATCH];
#pragma unroll
        for (int t = 0; t < MEAN_RBATCH; ++t) v[t] = *reinterpret_cast<const f32x4 *>(src + 4 * MEAN_RBATCH * h + 4 * t);
#pragma unroll
        for (int t = 0; t < MEAN_RBATCH; ++t) {
            s += v[t][0];
            s += v[t][1];
            s += v[t][2];
            s += v[t][3];
        }
    }
    s += __shfl_xor(s, 1);
    if (act && half == 0) {
        const int row = row0 + ri;
        float *m = (f == 0) ? mean[0] : (f == 1) ? mean[1] : (f == 2) ? mean[2] : mean[3];
        m[(int64_t)row * plane2] = s / den[row];
    }
}

#ifndef SWEEP_KERNEL_ATTR
#define SWEEP_KERNEL_ATTR
#endif
template <typename Tin, int NV, int W, bool FAST, bool TGT, bool MEAN = false>
__global__ __launch_bounds__(64) SWEEP_KERNEL_ATTR void mappm_sweep_kernel(const SweepArgs a)
{
    static_assert(!MEAN || TGT, "the fused block mean keeps its (single) target column in LDS");
    using V = typename FieldVec<W>::type;
    constexpr int NF = NV * W;
    constexpr unsigned int ESZ = sizeof(Tin);
    const int lane = threadIdx.x;
    // Workgroups go round-robin over the 8 XCDs (each with its own L2): give every XCD a contiguous eighth of the waves, so
    // the waves resident on one XCD at a time read neighbouring 256 / 512-byte pieces of a row.
    const unsigned int nwg = gridDim.x, wid = (nwg % 8u == 0u) ? (blockIdx.x % 8u) * (nwg / 8u) + blockIdx.x / 8u : blockIdx.x;
    int64_t b, c0;          // batch and first column (inside the batch plane) of the wave, uniform
    unsigned int lcol;      // the lane's column, counted from c0
    int64_t blk = 0;        // (MEAN) the wave's block inside the coarse plane
    if constexpr (MEAN) {
        const int64_t gw = a.col0 / 64 + wid;
        b = gw / a.pe2_plane;
        blk = gw - b * a.pe2_plane;
        const int64_t Y = blk / a.pe2_nx, X = blk - Y * a.pe2_nx;
        c0 = Y * 8 * a.nx + X * 8;
        lcol = (unsigned int)(lane >> 3) * (unsigned int)a.nx + (unsigned int)(lane & 7);
    } else {
        const int64_t wcol = a.col0 + (int64_t)wid * 64;
        b = wcol / a.n_inner;
        c0 = wcol - b * a.n_inner;
        lcol = (unsigned int)lane;
    }
    const int km = a.km, kn = a.kn, iv = a.iv;
    const int64_t plane = a.n_inner;
    // uniform row bases of the wave's batch, first column of the wave
    const char *pe1_row = static_cast<const char *>(a.pe1) + (b * (km + 1) * plane + c0) * ESZ;  // walks down the levels
    // target interfaces: same grid, or a grid coarser by pe2_f in y and x (the lane's column offset is then its coarse column's)
    const bool coarse2 = a.pe2_f > 1;
    const int64_t plane2 = coarse2 ? a.pe2_plane : plane;
    const char *pe2_b = static_cast<const char *>(a.pe2) + (b * (kn + 1) * plane2 + (coarse2 ? 0 : c0)) * ESZ;
    const unsigned int row_p2 = (unsigned int)plane2 * ESZ;  // bytes between target interfaces
    const char *q1_row[NF];
    char *q2_b[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        q1_row[f] = static_cast<const char *>(a.q1[f]) + (b * km * plane + c0) * ESZ;
        q2_b[f] = reinterpret_cast<char *>(a.q2[f]) + (b * kn * plane + c0) * 4;
    }
    const unsigned int lin = lcol * ESZ;                    // lane offset inside a source row
    const unsigned int row_in = (unsigned int)plane * ESZ;  // bytes between levels of the inputs (host: < 2^32 / levels)
    const unsigned int row_out = (unsigned int)plane * 4u;
    const float r3 = 1.f / 3.f, r23 = 2.f / 3.f;
    const int km1 = km - 1;
    auto ldq = [&](int v, size_t row_off) {  // the W fields of slot v at a row of q1
        V x;
#pragma unroll
        for (int w = 0; w < W; ++w) v_set(x, w, ld<Tin>(q1_row[v * W + w] + row_off, lin));
        return x;
    };

    // ---- head of the column: pe1(1..5), q1(1..4) ----
    float pe_a = ld<Tin>(pe1_row, lin), pe_b = ld<Tin>(pe1_row + row_in, lin), pe_c = ld<Tin>(pe1_row + 2 * (size_t)row_in, lin),
          pe_d = ld<Tin>(pe1_row + 3 * (size_t)row_in, lin), pe_e = ld<Tin>(pe1_row + 4 * (size_t)row_in, lin);
    const Tin ps_raw = *reinterpret_cast<const Tin *>(pe1_row + (size_t)km * row_in + lin);  // (MEAN compares it in the input type)
    const float pe1_top = pe_a, pe1_bot = (float)ps_raw;
    pe1_row += 5 * (size_t)row_in;  // -> pe1(6)
    V q0[NV], qp1[NV], qp2[NV], qp3[NV], q_top[NV], q_bot[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        q0[v] = ldq(v, 0);
        qp1[v] = ldq(v, row_in);
        qp2[v] = ldq(v, 2 * (size_t)row_in);
        qp3[v] = ldq(v, 3 * (size_t)row_in);
        q_top[v] = q0[v];
        q_bot[v] = ldq(v, (size_t)km1 * row_in);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) q1_row[f] += 4 * (size_t)row_in;  // -> q1(5)
    bool bad = !(pe_b >= pe_a) | !(pe_c >= pe_b) | !(pe_d >= pe_c) | !(pe_e >= pe_d);
    float d0 = pe_b - pe_a, dp1 = pe_c - pe_b, dp2 = pe_d - pe_c, dp3 = pe_e - pe_d;  // dp(L..L+3)

    // One level kk of the reconstruction: dc(kk) and al(kk) = a4(2,kk) (mappm.f90:658-683) from dp(kk-2..kk+1) =
    // (z, x, y, w), q(kk-1..kk+1) and dc(kk-1).  The pressure-only terms are formed once for all fields; d4c and its
    // reciprocal are next level's d4b.
    float d4b = d0 + dp1;  // d4(2) = dp(1) + dp(2): becomes d4(kk) of the first reconstructed level below
    // (`fast_c`: the arithmetic of this call -- the loop's own mode, or exact for the prologue, see there)
    auto level = [&](auto fast_c, float z, float x, float y, float w, float d4b_, const auto r4b, const V *qa, const V *qb,
                     const V *qc, const V *dca, V *dcb, V *alb, float &d4c_out, auto &r4c_out) {
        constexpr bool FASTL = decltype(fast_c)::value;
        const float d4a = z + x, d4c = y + w;
        const Den<FASTL> r4c(d4c);
        if constexpr (FASTL) {
            // everything that multiplies a field difference is folded into pressure-only factors first:
            //   df2 = k1 (qc - qb) + k2 (qb - qa),   al = qa + c1f h + m2 dc(k-1) - m1 dc(k),   c1f = (qb - qa) xr
            const float yr = y * __builtin_amdgcn_rcpf(d4b_ + w);
            const float k1 = __builtin_fmaf(0.5f, y, x) * r4c.v * yr, k2 = __builtin_fmaf(0.5f, y, w) * r4b.v * yr;
            const float a1 = d4a * __builtin_amdgcn_rcpf(d4b_ + x), a2 = d4c * __builtin_amdgcn_rcpf(d4b_ + y);
            const float g = 2.f * __builtin_amdgcn_rcpf(d4a + d4c), gy = g * y;
            const float h = __builtin_fmaf(gy, a1 - a2, 1.f), m2 = gy * a2, m1 = g * x * a1;
            const float xr = x * r4b.v;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const V dqa = qb[v] - qa[v], dqc = qc[v] - qb[v];
                const V df2 = v_fma(v_splat(dqc, k1), dqc, k2 * dqa);
                // (the hardware minima drop a NaN operand where the reference's compare-and-select chain keeps the one of
                // q(k) or q(k+1): 0 * (qc - qb) puts exactly that NaN back, so a NaN in a field poisons the same levels)
                const V dc = v_fma(v_splat(dqc, 0.f), dqc,
                                   v_sign(hw_min3_abs(df2, hw_max3(qa[v], qb[v], qc[v]) - qb[v], qb[v] - hw_min3(qa[v], qb[v], qc[v])), df2));
                V t = v_fma(dqa * xr, v_splat(dqa, h), qa[v]);
                t = v_fma(v_splat(dqa, m2), dca[v], t);
                alb[v] = v_fma(v_splat(dqa, -m1), dc, t);
                dcb[v] = dc;
            }
        } else {
        const float c1 = r4c.under(x + 0.5f * y);          // (dp(k-1) + 0.5 dp(k)) / d4(k+1)
        const float c2 = r4b.under(w + 0.5f * y);          // (dp(k+1) + 0.5 dp(k)) / d4(k)
        const Den<FASTL> r3p(d4b_ + w);                     // d4(k) + dp(k+1)
        const float a1 = Den<FASTL>(d4b_ + x).under(d4a);   // d4(k-1) / (d4(k) + dp(k-1))
        const float a2 = Den<FASTL>(d4b_ + y).under(d4c);   // d4(k+1) / (d4(k) + dp(k))
        const float g = Den<FASTL>(d4a + d4c).under(2.f);   // 2 / (d4(k-1) + d4(k+1))
        const float a12 = a1 - a2, xa1 = x * a1;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const V df2 = r3p.under(y * (c1 * (qc[v] - qb[v]) + c2 * (qb[v] - qa[v])));
            const V dc = v_sign(v_min3(v_abs(df2), v_max3(qa[v], qb[v], qc[v]) - qb[v], qb[v] - v_min3(qa[v], qb[v], qc[v])), df2);
            const V c1f = r4b.under((qb[v] - qa[v]) * x);
            alb[v] = qa[v] + c1f + g * (y * (c1f * a12 + a2 * dca[v]) - xa1 * dc);
            dcb[v] = dc;
        }
        }
        d4c_out = d4c;
        r4c_out = r4c;
    };

    // ---- prologue: dc(2), dc(3), al(3), then the top boundary (mappm.f90:689-725) ----
    // Once per column, and in EXACT arithmetic in both modes: the top boundary clamps al(2) between q(1) and q(2), and a
    // clamp that lands exactly on q(1) makes dm(1) = 0 -- the limiter's discontinuous branch.  Left to the fast arithmetic,
    // an ulp decided it differently from the reference on one level in 3e8 (found at C384 with a third of configs[2]'s
    // spread: 24.7 on values of +-1000, inside the source layers' bounds, but avoidable here for a few divisions per column).
    V al0[NV], al1[NV], al2[NV], dc0[NV], dc1[NV], dc2[NV], ar_km[NV];
    float d4c = dp1 + dp2;  // d4(3)
    Den<false> r4c(d4c);
    {
        // dc(2): the dc part of a level with (x, y, w) = (dp(1), dp(2), dp(3)); its al is not used
        const Den<false> r4b(d4b);
        const float c1 = r4c.under(d0 + 0.5f * dp1);
        const float c2 = r4b.under(dp2 + 0.5f * dp1);
        const Den<false> r3p(d4b + dp2);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const V df2 = r3p.under(dp1 * (c1 * (qp2[v] - qp1[v]) + c2 * (qp1[v] - q0[v])));
            dc1[v] = v_sign(v_min3(v_abs(df2), v_max3(q0[v], qp1[v], qp2[v]) - qp1[v], qp1[v] - v_min3(q0[v], qp1[v], qp2[v])), df2);
        }
    }
    {
        V dc_3[NV], al_3[NV];
        float d4n;
        Den<false> r4n(1.f);
        level(std::false_type{}, d0, dp1, dp2, dp3, d4c, r4c, qp1, qp2, qp3, dc1, dc_3, al_3, d4n, r4n);  // kk = 3 (recomputed by the loop's L = 1)
        const float d1 = d0, d2 = dp1;
        const Den<false> r12(d1 + d2);
        const Den<false> rcub(d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
        const float poly = d2 * (5.f * d1 + d2) - 3.f * d1 * d1, d2p = d2 + 3.f * d1, d1sq = d1 * d1;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const V zero = v_splat(q0[v], 0.f);
            const V qm = r12.under(d2 * q0[v] + d1 * qp1[v]);
            const V dq = r12.under(2.f * (qp1[v] - q0[v]));
            const V c1 = rcub.under(4.f * (al_3[v] - qm - d2 * dq));
            const V c3 = dq - 0.5f * c1 * poly;
            V a2 = qm - 0.25f * c1 * d1 * d2 * d2p;
            V a1 = d1 * (2.f * c1 * d1sq - c3) + a2;
            a2 = v_max2(a2, v_min2(q0[v], qp1[v]));
            a2 = v_min2(a2, v_max2(q0[v], qp1[v]));
            dc0[v] = 0.5f * (a2 - q0[v]);
            if (iv == 0) {
                a1 = v_max2(zero, a1);
                a2 = v_max2(zero, a2);
            } else if (iv == -1) {
                a1 = v_sel(a1 * q0[v] <= zero, zero, a1);
            } else if (iv == 2 || iv == -2) {
                a1 = q0[v];
            }
            al0[v] = a1;
            al1[v] = a2;
            al2[v] = zero;
            dc2[v] = zero;
            ar_km[v] = zero;
        }
    }
    // the loop's first level is kk = 3: its d4(kk) = d4(3)
    d4b = d4c;
    Den<FAST> r4b(d4b);

    // ---- per-lane target cursor ----
    // A lane consumes the target interfaces at its own pace.  Loading pe2(k+1) where it is needed would put a memory
    // round trip (and, vmcnt being one in-order counter, a wait for everything else in flight) into the emit code, so
    // the wave keeps a ring of kRing interface rows in LDS, filled UNIFORMLY: row `jr` is requested while lane 0's
    // cursor is within kAhead rows of it (at most two rows per source layer, scalar bookkeeping) and written to the ring
    // at the top of the next iteration.  A lane whose cursor has left the window [jl - kRing, jl) -- grids whose lanes
    // drift apart by more than the ring holds -- reads its interface from memory instead: slower, same value.
    //
    // TGT (round 3) -- what the restart pipelines launch: the target grid is coarser by a power of two f >= 8 and a wave's
    // 64 columns lie in one fine row (nx % 64 == 0), so the wave has only 64 / f <= 8 DISTINCT target columns.  All their
    // kn + 1 interfaces fit in (kn + 1) x 8 floats of LDS (2.5 KB at kn = 79), read once up front from the (L2-resident)
    // coarse array: no ring, no window bookkeeping, no interface requests inside the loop and -- the point -- no lane ever
    // leaves a window.  On BASELINE configs[2]'s synthetic grid (delp ~ U(300, 1500) iid per cell) the lanes of a wave are
    // spread over ~16 target rows at the bottom of the column and most waves took the ring's memory path (a full round trip
    // behind `s_waitcnt vmcnt(0)`) on every level of the lower two thirds.
    constexpr int kRing = kSweepRing, kAhead = 9;
    float *ring = sweep_lds + lane;
    unsigned int lane2 = (unsigned int)lane * ESZ;
    if (coarse2 && !TGT) {
        const unsigned int c = (unsigned int)c0 + lane, y = c / (unsigned int)a.nx, x = c - y * (unsigned int)a.nx;
        lane2 = ((y / (unsigned int)a.pe2_f) * (unsigned int)a.pe2_nx + x / (unsigned int)a.pe2_f) * ESZ;
    }
    int jl = 0, jr = 0;  // interface rows (0-based) [0, jl) have landed in the ring, [jl, jr) are in flight (pv0, pv1)
    const float *tgt = sweep_lds;
    if constexpr (MEAN) {   // one target column: the block's own coarse column
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = lane + 64 * j;   // (kn + 1 <= 128, host)
            const Tin x = *reinterpret_cast<const Tin *>(pe2_b + (size_t)(i <= kn ? i : kn) * row_p2 + (size_t)blk * ESZ);
            if (i <= kn) sweep_lds[i] = (float)x;
        }
    } else if constexpr (TGT) {
        const unsigned int y = (unsigned int)c0 / (unsigned int)a.nx, x0 = (unsigned int)c0 - y * (unsigned int)a.nx;
        const unsigned int sh = 31u - (unsigned int)__builtin_clz((unsigned int)a.pe2_f);  // log2 f
        const unsigned int cell0 = (y >> sh) * (unsigned int)a.pe2_nx + (x0 >> sh), cells = 64u >> sh;
        tgt = sweep_lds + (lane >> sh);
        const int n = (kn + 1) * 8;
        constexpr int kTrips = 16;  // (kn + 1) * 8 <= 1024 (host)
        Tin tmp[kTrips];
#pragma unroll
        for (int j = 0; j < kTrips; ++j) {
            const int i = lane + 64 * j, row = i >> 3;
            const unsigned int cell = (unsigned int)(i & 7);
            // (unconditional, clamped: sixteen requests in flight, one wait)
            tmp[j] = *reinterpret_cast<const Tin *>(pe2_b + (size_t)(row <= kn ? row : kn) * row_p2 + (size_t)(cell0 + (cell < cells ? cell : cells - 1u)) * ESZ);
        }
#pragma unroll
        for (int j = 0; j < kTrips; ++j)
            if (lane + 64 * j < n) sweep_lds[lane + 64 * j] = (float)tmp[j];
    } else {
        float tmp[kRing];
#pragma unroll
        for (int i = 0; i < kRing; ++i) tmp[i] = ld<Tin>(pe2_b, (unsigned int)(i <= kn ? i : kn) * row_p2 + lane2);
#pragma unroll
        for (int i = 0; i < kRing; ++i) ring[i * 64] = tmp[i];
        jl = jr = (kRing < kn + 1) ? kRing : kn + 1;
    }
    Tin pv0 = (Tin)0, pv1 = (Tin)0;  // raw, converted when they land (see q_raw)
    auto PE2 = [&](int i) -> float {  // interface row i (0-based, <= kn)
        if constexpr (MEAN) {
            return tgt[i];
        } else if constexpr (TGT) {
            return tgt[i * 8];
        } else {
            // (the LDS read is unconditional and the rare memory read sits in a branch of its own that waits for it right
            // there: a select between the two addresses becomes a flat load, and a wait placed after the branches merge
            // would stall every lane on everything the wave has in flight)
            float v = ring[(i & (kRing - 1)) * 64];
            if (!((unsigned int)(jl - 1 - i) < (unsigned int)kRing)) {
                v = ld<Tin>(pe2_b, (unsigned int)i * row_p2 + lane2);
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
            return v;
        }
    };
    unsigned int offq = lcol * 4u;  // offset of q2(k)
    int k = 1;
    float p2k = PE2(0), p2k1 = PE2(1);  // pe2(k), pe2(k+1)   (kn >= 1)
    bool accum = false;
    V qsum[NV];
    float dpsum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) qsum[v] = v_splat(q0[v], 0.f);
    constexpr int kOut = MEAN ? mean_out_rows<Tin, NF, FAST>() : sweep_out_rows<Tin, NF, FAST>();  // rows of the result ring (below)
    int oslot = 0, fslot = 0;  // ring slot of this lane's row k - 1 / of the wave's row rf, counted along (kOut need not be a power of two)
    auto advance = [&]() {
        ++k;
        oslot = (oslot + 1 == kOut) ? 0 : oslot + 1;
        if (!(p2k1 >= p2k)) bad = true;  // also catches NaN
        p2k = p2k1;
        p2k1 = PE2(k <= kn ? k : kn);
        offq += row_out;
    };
    // ---- results leave through a second LDS ring ----
    // A lane finishes its targets at its own pace: stored directly, every q2 store is a scatter of 4-byte pieces over
    // as many rows as the wave's lanes are spread over, and (vmcnt being one in-order counter for loads and stores) it is
    // waited for by the next wait that guards the row loads.  Measured at C384, 4 fields: direct stores 2.18 ms, the
    // same stores deferred to the top of the next iteration 1.97 ms, no stores at all 0.80 ms -- the scatter itself is
    // what costs.  So a lane parks target row r = k - 1 in slot r % kOut of its column of the ring, and the WAVE writes
    // row `rf` out -- one coalesced row per field, scalar address -- once every lane is past it, at the top of an
    // iteration, right after the wait.  `rl` (per lane) = rows [0, rl) of this lane no longer live in the ring: a lane
    // that runs kOut rows ahead of the slowest one writes its own oldest row out first.  Ill-formed lanes count as past
    // every row; whatever the flush writes for them is overwritten by the fallback pass.  (kOut = 16 and 32 were slower:
    // the ring's LDS footprint costs occupancy.)
    float *oring = sweep_lds + (TGT ? (kn + 1) * 8 : kRing * 64) + lane;
    int rf = 0;  // uniform: rows [0, rf) are in memory for every lane
    int rl = 0;  // per lane (>= rf where it matters)
    V out_v[NV];
    auto OUT = [&](int v, V x) { out_v[v] = x; };
    // (MEAN) tables and ring of the block, the lane's weight
    Tin *m_lvl = nullptr;
    float *m_den = nullptr, *m_ring = nullptr;
    float area_l = 0.f;
    float *m_mean[4] = {nullptr, nullptr, nullptr, nullptr};
    const int mpos = mean_pos(lane);
    if constexpr (MEAN) {
        m_lvl = reinterpret_cast<Tin *>(sweep_lds + mean_n1(kn));
        m_den = sweep_lds + mean_n1(kn) + mean_nl(kn, ESZ);
        m_ring = sweep_lds + mean_tab_floats(kn, ESZ);
        area_l = a.area[(b / a.area_repeat) * plane + c0 + lcol];
        const char *lvl_col = static_cast<const char *>(a.lvl) + (b * a.cmp_levels * plane2 + blk) * ESZ;
        mean_tables<Tin>(lvl_col, row_p2, a.cmp_offset, kn, lane, area_l, ps_raw, m_lvl, m_den, m_ring);
#pragma unroll
        for (int f = 0; f < NF; ++f) m_mean[f] = a.mean[f] + b * kn * plane2 + blk;
        if (lane == 0) sweep_lds[mean_n1(kn) - 1] = 0.f;   // the wave's spill flag (see below)
    }
    // A ring of kOut rows holds the block while its lanes stay within kOut target rows of each other -- smooth thicknesses, i.e.
    // most of a real restart file.  Where they do not (steep terrain inside the block; BASELINE configs[2]'s iid thicknesses
    // everywhere) the first lane that finds its slot still occupied raises the wave's `spill` flag, and from the next iteration
    // on the wave behaves like the plain sweep for the rest of its column: finished rows leave as coalesced stores of p to
    // the scratch rows, a lane kOut rows ahead writes its own oldest row first, nothing waits for memory.  The block is
    // listed with the first row it did not sum, and mean_rest_kernel sums those rows from the scratch afterwards -- same
    // values, same order, same means.  (Bringing evicted rows back into the ring instead -- at the flush, or asynchronously a
    // row per iteration -- was measured: every variant ends with most of the wave's rows making a round trip through memory
    // on which the flush then waits; 1.2 - 2.7 ms against 0.95 for the plain sweep on the iid data.)
    float *m_flag = sweep_lds + mean_n1(kn) - 1;   // (the spare word behind the kn + 1 interfaces)
    bool spill = false;   // uniform
    int spill_row = 0;    // uniform: first row that was not summed here
    // (spilled rows lie BLOCK-major in the scratch -- [block][row][lane], a row of the wave = 256 contiguous bytes -- so they
    // leave and come back as whole rows; the fine layout would cut them into the block's eight 32-byte pieces)
    char *s_b[NF];   // (uniform: the lane's 4 bytes are added at the store)
#pragma unroll
    for (int f = 0; f < NF; ++f) s_b[f] = nullptr;
    if constexpr (MEAN) {
#pragma unroll
        for (int f = 0; f < NF; ++f) s_b[f] = reinterpret_cast<char *>(a.q2[f] + (a.col0 / 64 + wid) * (int64_t)kn * 64);
    }
    const unsigned int lane4 = (unsigned int)lane * 4u;
    auto out_end_mean = [&]() {
        const int r = k - 1, slot = oslot;
        if (r - kOut >= (rl > rf ? rl : rf)) {  // the slot still holds this lane's row r - kOut: that one goes to the scratch now
#pragma unroll
            for (int f = 0; f < NF; ++f)
                *reinterpret_cast<float *>(s_b[f] + ((unsigned int)(r - kOut) * 256u + lane4)) = m_ring[(slot * NF + f) * kMeanStride + mpos];
            rl = r - kOut + 1;
            *m_flag = 1.f;
        }
        const float w = (m_lvl[r] < ps_raw) ? area_l : 0.f;   // regridz.py:209-220, compared in the pressures' own type
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float p = v_get(out_v[f / W], f % W) * w;
            m_ring[(slot * NF + f) * kMeanStride + mpos] = f_isnan(p) ? 0.f : p;
        }
    };
    auto flush_groups = [&](int max_groups) {  // uniform: groups of rows every lane has parked become block means
#pragma unroll 1
        for (int n = 0; n < max_groups && rf < kn; ++n) {
            const int nrows = (kn - rf < kMeanGroup) ? kn - rf : kMeanGroup;
            const bool past = bad | (k > rf + nrows);
            if (__builtin_amdgcn_ballot_w64(past) != __builtin_amdgcn_ballot_w64(true)) break;
#ifndef MEAN_NO_REDUCE  // (timing experiment: wrong results)
            mean_reduce<NF>(m_ring, fslot, rf, nrows, m_den, m_mean, plane2, lane);
#endif
            rf += nrows;
            fslot = (fslot + nrows >= kOut) ? fslot + nrows - kOut : fslot + nrows;
        }
    };
    auto flush_spill = [&](int max_rows) {  // uniform, spill mode: rows every lane has parked go to the scratch as they are
#pragma unroll 1
        for (int n = 0; n < max_rows && rf < kn; ++n) {
            const bool past = bad | (k - 1 > rf);
            if (__builtin_amdgcn_ballot_w64(past) != __builtin_amdgcn_ballot_w64(true)) break;
            if (rl <= rf) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    *reinterpret_cast<float *>(s_b[f] + ((unsigned int)rf * 256u + lane4)) = m_ring[(fslot * NF + f) * kMeanStride + mpos];
            }
            ++rf;
            fslot = (fslot + 1 == kOut) ? 0 : fslot + 1;
        }
    };
    auto out_end = [&]() {  // after the OUTs of target k (before advance())
        if constexpr (MEAN) {
            out_end_mean();
            return;
        }
        const int r = k - 1, slot = oslot;
        if (r - kOut >= (rl > rf ? rl : rf)) {  // the slot still holds this lane's row r - kOut: write it out now
#pragma unroll
            for (int f = 0; f < NF; ++f)
                *reinterpret_cast<float *>(q2_b[f] + (offq - (unsigned int)kOut * row_out)) = oring[(f * kOut + slot) * 64];
            rl = r - kOut + 1;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) oring[(f * kOut + slot) * 64] = v_get(out_v[f / W], f % W);
    };
    auto flush_rows = [&](int max_rows) {  // uniform: rows every lane has written go to memory
        if constexpr (MEAN) {
            if (!spill && *m_flag != 0.f) {   // (raised inside the emit code of the previous iteration; uniform: one LDS word)
                spill = true;
                spill_row = rf;
            }
            if (spill)
                flush_spill(max_rows);
            else
                flush_groups(max_rows >= kn ? kn : 1);
            return;
        }
#pragma unroll 1
        for (int n = 0; n < max_rows && rf < kn; ++n) {
            const bool past = bad | (k - 1 > rf);
            if (__builtin_amdgcn_ballot_w64(past) != __builtin_amdgcn_ballot_w64(true)) break;
            if (rl <= rf) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    *reinterpret_cast<float *>(q2_b[f] + (size_t)rf * row_out + (unsigned int)lane * 4u) =
                        oring[(f * kOut + fslot) * 64];
            }
            ++rf;
            fslot = (fslot + 1 == kOut) ? 0 : fslot + 1;
        }
    };
    if (!(p2k1 >= p2k)) bad = true;
    while (k <= kn && !bad && p2k <= pe1_top) {  // targets that start at or above the old top (mappm.f90:62-64)
#pragma unroll
        for (int v = 0; v < NV; ++v) OUT(v, q_top[v]);
        out_end();
        advance();
    }
    bool live = (k <= kn) && !bad && !(p2k >= pe1_bot);

    // ---- row prefetch, two levels deep ----
    // The rows of level L + 4 (q) / L + 5 (pe1) are consumed at the BOTTOM of iteration L.  Requested at the top of the same
    // iteration (round 2) they had ~0.8 of an iteration in flight, and a wave holds few enough bytes in flight (5 rows) that
    // the launch ran at the latency-bandwidth product of its occupancy, not at the VALU or HBM limit (r03 counters: 35 % of
    // the wave cycles parked at s_waitcnt).  Now iteration L requests the rows of iteration L + 1 into the OTHER of two
    // staging sets, so a request has a whole iteration more; the loop body is instantiated twice with the sets swapped
    // (a rotation by register moves would wait for the rows just requested).  The requests are unconditional -- the row
    // pointers stop at the last row -- and issued after the iteration's other memory operations, so that the counted wait at
    // the bottom (`vmcnt(NF + 1)`: everything but the newest set) is exact on every path.
    // The rows stay in the INPUT type until they are consumed: a conversion written next to the load makes the compiler
    // wait for the load right there (r02's float64 instantiations had `s_waitcnt vmcnt(0)` directly behind the loads).
    Tin q_setA[NF], q_setB[NF], pe_setA, pe_setB = (Tin)0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        q_setA[f] = *reinterpret_cast<const Tin *>(q1_row[f] + lin);  // q(5)  (km >= 8)
        q1_row[f] += row_in;
        q_setB[f] = (Tin)0;
    }
    pe_setA = *reinterpret_cast<const Tin *>(pe1_row + lin);  // pe1(6)
    pe1_row += row_in;
    auto sweep_level = [&](const int L, Tin (&q_ld)[NF], Tin &pe_ld, Tin (&q_use)[NF], Tin &pe_use) {
        // ---- the interface rows requested during the previous iteration land in the ring ----
        if constexpr (!TGT) {
            if (jr > jl) {
                asm volatile("" : "+v"(pv0));
                ring[(jl & (kRing - 1)) * 64] = (float)pv0;
            }
            if (jr > jl + 1) {
                asm volatile("" : "+v"(pv1));
                ring[((jl + 1) & (kRing - 1)) * 64] = (float)pv1;
            }
            jl = jr;
        }
        flush_rows(2);
        // Order inside an iteration: the memory requests first (target-interface rows, then the rows of iteration L + 1),
        // then layer L's finalisation and its emits, then the reconstruction of level L + 2 -- the wait at the top of the
        // next iteration also waits for whatever was stored here, so the reconstruction sits behind the stores.
        if constexpr (!TGT) {  // up to two target-interface rows, kept kAhead ahead of lane 0's cursor
            const int want = __builtin_amdgcn_readfirstlane(k) + kAhead;
            if (jr <= kn && jr < want) {
                pv0 = *reinterpret_cast<const Tin *>(pe2_b + (size_t)jr * row_p2 + lane2);
                ++jr;
                if (jr <= kn && jr < want) {
                    pv1 = *reinterpret_cast<const Tin *>(pe2_b + (size_t)jr * row_p2 + lane2);
                    ++jr;
                }
            }
        }
        // ---- requests for iteration L + 1: q(L+5), pe1(L+6) (the last row again once the column is exhausted) ----
#pragma unroll
        for (int f = 0; f < NF; ++f) q_ld[f] = *reinterpret_cast<const Tin *>(q1_row[f] + lin);
        pe_ld = *reinterpret_cast<const Tin *>(pe1_row + lin);
        if (L + 6 <= km) {
#pragma unroll
            for (int f = 0; f < NF; ++f) q1_row[f] += row_in;
            pe1_row += row_in;
        }
        // ---- finalise layer L: A6 and the monotonicity constraint (mappm.f90:773-849 with lmt = 0) ----
        V al[NV], ar[NV], a6[NV], da1[NV], a6pd[NV];  // (da1 = ar - al and a6pd = a6 + da1: fast mode only)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            al[v] = al0[v];
            ar[v] = (L == km) ? ar_km[v] : al1[v];
            if constexpr (FAST) {
                limit0_fast(dc0[v], q0[v], al[v], ar[v], a6[v], da1[v]);
                a6pd[v] = a6[v] + da1[v];
            } else {
                a6[v] = 3.f * (2.f * q0[v] - (al[v] + ar[v]));
                limit0(dc0[v], q0[v], al[v], ar[v], a6[v]);
            }
        }
        const float pL = pe_a, pL1 = pe_b;
        const Den<FAST> rd0(d0);

        // ---- emit the target layers that end inside layer L (see mappm_merge_kernel for the event order) ----
        if (live && accum && !(p2k1 > pL1)) {
            const float delp = p2k1 - pL;
            const float PR = rd0.under(delp);
            dpsum = dpsum + delp;
            const Den<FAST> rs(dpsum);
            const float hp = 0.5f * PR, tp = FAST ? __builtin_fmaf(-r23, PR, 1.f) : 1.f - r23 * PR;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if constexpr (FAST)
                    qsum[v] = v_fma(v_splat(qsum[v], delp), v_fma(v_splat(qsum[v], hp), v_fma(a6[v], v_splat(qsum[v], tp), da1[v]), al[v]), qsum[v]);
                else
                    qsum[v] = qsum[v] + delp * (al[v] + hp * (ar[v] - al[v] + a6[v] * tp));
                OUT(v, rs.under(qsum[v]));
            }
            out_end();
            accum = false;
            advance();
            live = (k <= kn) && !bad && !(p2k >= pe1_bot);
        }
        while (live && !accum && (p2k >= pL && p2k <= pL1) && (p2k1 <= pL1)) {
            const float PR = rd0.under(p2k1 - pL);
            const float PL = rd0.under(p2k - pL);
            const float sp = PR + PL, TT = FAST ? r3 * __builtin_fmaf(PR, sp, PL * PL) : r3 * (PR * (PR + PL) + PL * PL);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if constexpr (FAST)
                    OUT(v, v_fma(-a6[v], v_splat(a6[v], TT), v_fma(a6pd[v], v_splat(a6[v], 0.5f * sp), al[v])));
                else
                    OUT(v, al[v] + 0.5f * (a6[v] + ar[v] - al[v]) * sp - a6[v] * TT);
            }
            out_end();
            advance();
            live = (k <= kn) && !bad && !(p2k >= pe1_bot);
        }
        if (live) {
            if (accum) {  // whole layer (mappm.f90:99-104)
#pragma unroll
                for (int v = 0; v < NV; ++v) qsum[v] = FAST ? v_fma(v_splat(q0[v], d0), q0[v], qsum[v]) : qsum[v] + d0 * q0[v];
                dpsum = dpsum + d0;
            } else if (p2k >= pL && p2k <= pL1) {  // fractional area (mappm.f90:85-92)
                const float PL = rd0.under(p2k - pL);
                const float delp = pL1 - p2k;
                const float sp = 1.f + PL, TT = FAST ? r3 * __builtin_fmaf(PL, sp, 1.f) : r3 * (1.f + PL * (1.f + PL));
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if constexpr (FAST)
                        qsum[v] = delp * v_fma(-a6[v], v_splat(a6[v], TT), v_fma(a6pd[v], v_splat(a6[v], 0.5f * sp), al[v]));
                    else
                        qsum[v] = delp * (al[v] + 0.5f * (a6[v] + ar[v] - al[v]) * sp - a6[v] * TT);
                }
                dpsum = delp;
                accum = true;
            }
        }
        // ---- reconstruction of level kk = L + 2 from dp(L..L+3), q(L+1..L+3) ----
        const int kk = L + 2;
        if (kk <= km1) {
            float d4n;
            Den<FAST> r4n(1.f);
            level(std::integral_constant<bool, FAST>{}, d0, dp1, dp2, dp3, d4b, r4b, qp1, qp2, qp3, dc1, dc2, al2, d4n, r4n);
            d4b = d4n;
            r4b = r4n;
        } else if (kk == km) {
            // bottom boundary (mappm.f90:729-761): al(km), ar(km), dc(km) from al(km-1) = al1
            const float d1 = dp2, d2 = dp1;  // dp(km), dp(km-1)
            const Den<FAST> r12(d1 + d2);
            const Den<FAST> rcub(d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
            const float poly = d2 * (5.f * d1 + d2) - 3.f * d1 * d1, d2p = d2 + 3.f * d1, d1sq = d1 * d1;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const V zero = v_splat(q0[v], 0.f);
                const V qk = qp2[v], qk1 = qp1[v];  // q(km), q(km-1)
                const V qm = r12.under(d2 * qk + d1 * qk1);
                const V dq = r12.under(2.f * (qk1 - qk));
                const V c1 = rcub.under(al1[v] - qm - d2 * dq);
                const V c3 = dq - 2.0f * c1 * poly;
                V alk = qm - c1 * d1 * d2 * d2p;
                V ark = d1 * (8.f * c1 * d1sq - c3) + alk;
                alk = v_max2(alk, v_min2(qk, qk1));
                alk = v_min2(alk, v_max2(qk, qk1));
                dc2[v] = 0.5f * (qk - alk);
                if (iv == 0) {
                    alk = v_max2(zero, alk);
                    ark = v_max2(zero, ark);
                } else if (iv < 0) {
                    ark = v_sel(qk * ark <= zero, zero, ark);
                }
                al2[v] = alk;
                ar_km[v] = ark;
            }
        }
        // ---- level L + 1 becomes the current one (at the bottom of the iteration, unconditionally: with the rotation under
        // `if (L > 1)` at the top the compiler shuffled the window forth at the top and back at the bottom, 18 moves a level)
        V q_in[NV];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            asm volatile("" : "+v"(q_use[f]));  // the conversion stays here, behind the wait for the row
            v_set(q_in[f / W], f % W, (float)q_use[f]);
        }
        asm volatile("" : "+v"(pe_use));
        const float pe_in = (float)pe_use;
        if (!(pe_in >= pe_e)) bad = bad | (L < km);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            q0[v] = qp1[v]; qp1[v] = qp2[v]; qp2[v] = qp3[v]; qp3[v] = q_in[v];
            al0[v] = al1[v]; al1[v] = al2[v];
            dc0[v] = dc1[v]; dc1[v] = dc2[v];
        }
        d0 = dp1; dp1 = dp2; dp2 = dp3; dp3 = pe_in - pe_e;
        pe_a = pe_b; pe_b = pe_c; pe_c = pe_d; pe_d = pe_e; pe_e = pe_in;
    };
    int L = 1;
    for (; L < km; L += 2) {
        sweep_level(L, q_setB, pe_setB, q_setA, pe_setA);
        sweep_level(L + 1, q_setA, pe_setA, q_setB, pe_setB);
    }
    if (L == km) sweep_level(L, q_setB, pe_setB, q_setA, pe_setA);

    // ---- past the old surface (mappm.f90:115-121), then the run that copies q1(km) ----
    if (k <= kn && !bad && accum) {
        const float delp = p2k1 - pe1_bot;
        if (delp > 0.f) {
#pragma unroll
            for (int v = 0; v < NV; ++v) qsum[v] = qsum[v] + delp * q_bot[v];
            dpsum = dpsum + delp;
        }
        const Den<FAST> rs(dpsum);
#pragma unroll
        for (int v = 0; v < NV; ++v) OUT(v, rs.under(qsum[v]));
        out_end();
        advance();
    }
    while (k <= kn && !bad) {
        if (p2k >= pe1_bot) {
#pragma unroll
            for (int v = 0; v < NV; ++v) OUT(v, q_bot[v]);
            out_end();
            advance();
        } else {
            bad = true;  // a top-edge search that no source layer satisfied
        }
    }
    flush_rows(kn);  // every lane is done (or ill-formed): the rows still in the ring
    if constexpr (MEAN) {
        if (spill && __builtin_amdgcn_ballot_w64(bad) == 0 && lane == 0) {  // the rows from spill_row on are summed by mean_rest_kernel
            const unsigned int i = atomicAdd(a.n_bad + 2, 1u);
            a.rest_blocks[2 * i] = (unsigned int)(a.col0 / 64 + wid);
            a.rest_blocks[2 * i + 1] = (unsigned int)spill_row;
        }
        if (__builtin_amdgcn_ballot_w64(bad) != 0) {  // the whole block is redone: its columns by mappm_fallback_kernel, its means by mean_redo_kernel
            unsigned int base = 0;
            if (lane == 0) {
                base = atomicAdd(a.n_bad, 64u);
                a.bad_blocks[atomicAdd(a.n_bad + 1, 1u)] = (unsigned int)(a.col0 / 64 + wid);
            }
            base = __builtin_amdgcn_readfirstlane(base);
            a.bad_cols[base + lane] = (unsigned int)(b * plane + c0 + lcol);
        }
    } else {
        if (bad) a.bad_cols[atomicAdd(a.n_bad, 1u)] = (unsigned int)(wid * 64 + lane);  // redone by mappm_fallback_kernel
    }
}

// The blocks the fused kernel listed, after mappm_fallback_kernel has rewritten all their columns in the scratch rows: the same
// masked means from there.  One wave per block, one field at a time; the arithmetic is mean_tables / mean_reduce, as above.
template <typename Tin>
__global__ __launch_bounds__(64) void mean_redo_kernel(const SweepArgs a, int nf)
{
    constexpr unsigned int ESZ = sizeof(Tin);
    const int lane = threadIdx.x, kn = a.kn, km = a.km;
    const unsigned int count = a.n_bad[1];
    const int64_t plane = a.n_inner, plane2 = a.pe2_plane;
    float *tgt_unused = sweep_lds;
    (void)tgt_unused;
    Tin *m_lvl = reinterpret_cast<Tin *>(sweep_lds + mean_n1(kn));
    float *m_den = sweep_lds + mean_n1(kn) + mean_nl(kn, ESZ);
    float *m_ring = sweep_lds + mean_tab_floats(kn, ESZ);
    const int mpos = mean_pos(lane);
    for (unsigned int i = blockIdx.x; i < count; i += gridDim.x) {
        const int64_t gw = a.bad_blocks[i];
        const int64_t b = gw / plane2, blk = gw - b * plane2, Y = blk / a.pe2_nx, X = blk - Y * a.pe2_nx;
        const int64_t c0 = Y * 8 * a.nx + X * 8;
        const unsigned int lcol = (unsigned int)(lane >> 3) * (unsigned int)a.nx + (unsigned int)(lane & 7);
        const float area_l = a.area[(b / a.area_repeat) * plane + c0 + lcol];
        const Tin ps_raw = static_cast<const Tin *>(a.pe1)[(b * (km + 1) + km) * plane + c0 + lcol];
        const char *lvl_col = static_cast<const char *>(a.lvl) + (b * a.cmp_levels * plane2 + blk) * ESZ;
        mean_tables<Tin>(lvl_col, (unsigned int)plane2 * ESZ, a.cmp_offset, kn, lane, area_l, ps_raw, m_lvl, m_den, m_ring);
        for (int f = 0; f < nf; ++f) {
            const float *q = a.q2[f] + b * kn * plane + c0 + lcol;
            float *mean[4] = {a.mean[f] + b * kn * plane2 + blk, nullptr, nullptr, nullptr};
            for (int row0 = 0; row0 < kn; row0 += kMeanGroup) {
                const int nrows = (kn - row0 < kMeanGroup) ? kn - row0 : kMeanGroup;
                for (int j = 0; j < nrows; ++j) {
                    const int r = row0 + j;
                    const float w = (m_lvl[r] < ps_raw) ? area_l : 0.f;
                    const float p = q[(int64_t)r * plane] * w;
                    m_ring[j * kMeanStride + mpos] = f_isnan(p) ? 0.f : p;
                }
                mean_reduce<1>(m_ring, 0, row0, nrows, m_den, mean, plane2, lane);
            }
        }
    }
}

// (the file is compiled twice -- remap.o: the sweep launches, remap_mean.o: the fused block-mean launches -- so that the two
// sets of instantiations build side by side; see the Makefile)
#ifndef FV3HIP_REMAP_PART_MEAN
template <typename Tin, int NV, int W>
void launch_sweep2(const SweepArgs &a, int64_t n_waves, bool fast, bool tgt, hipStream_t st)
{
#define SWEEP_(F, T)                                                                                                                \
    hipLaunchKernelGGL((mappm_sweep_kernel<Tin, NV, W, F, T>), dim3((unsigned)n_waves), dim3(64),                                   \
                       (size_t)((tgt ? (a.kn + 1) * 8 : kSweepRing * 64) + NV * W * sweep_out_rows<Tin, NV * W, F>() * 64) * sizeof(float), st, a)
    if (fast) {
        if (tgt) SWEEP_(true, true); else SWEEP_(true, false);
    } else {
        if (tgt) SWEEP_(false, true); else SWEEP_(false, false);
    }
#undef SWEEP_
}

// fields per launch -> (register slots, fields per slot): pairs of fields share packed-math instructions
template <typename Tin>
void launch_sweep1(const SweepArgs &a, int nf, int64_t n_waves, bool fast, bool tgt, hipStream_t st)
{
#ifdef FV3HIP_REMAP_SUBSET  // (experiment builds: the float64 four-field instantiations only)
    if constexpr (sizeof(Tin) == 8) launch_sweep2<Tin, 2, 2>(a, n_waves, fast, tgt, st);
#else
    switch (nf) {
        case 1: launch_sweep2<Tin, 1, 1>(a, n_waves, fast, tgt, st); break;
        case 2: launch_sweep2<Tin, 1, 2>(a, n_waves, fast, tgt, st); break;
        case 3: launch_sweep2<Tin, 3, 1>(a, n_waves, fast, tgt, st); break;
        default: launch_sweep2<Tin, 2, 2>(a, n_waves, fast, tgt, st); break;
    }
#endif
}

#endif  // sweep part
#ifndef FV3HIP_REMAP_PART_SWEEP
// The blocks whose waves went into spill mode: rows [row0, kn) of their NF fields lie in the scratch, block-major, as parked
// (p = q2 * w, NaN -> 0); their means from there -- mean_reduce again, so the same sums.  One wave per listed block, the loads of a group of
// rows in flight while the previous group is summed.
template <typename Tin, int NF>
__global__ __launch_bounds__(64) void mean_rest_kernel(const SweepArgs a)
{
    constexpr unsigned int ESZ = sizeof(Tin);
    const int lane = threadIdx.x, kn = a.kn, km = a.km;
    const unsigned int count = a.n_bad[2];
    const int64_t plane = a.n_inner, plane2 = a.pe2_plane;
    Tin *m_lvl = reinterpret_cast<Tin *>(sweep_lds + mean_n1(kn));
    float *m_den = sweep_lds + mean_n1(kn) + mean_nl(kn, ESZ);
    float *m_ring = sweep_lds + mean_tab_floats(kn, ESZ);
    const int mpos = mean_pos(lane);
    const unsigned int lcol = (unsigned int)(lane >> 3) * (unsigned int)a.nx + (unsigned int)(lane & 7);
    for (unsigned int i = blockIdx.x; i < count; i += gridDim.x) {
        const int64_t gw = a.rest_blocks[2 * i];
        const int row0 = (int)a.rest_blocks[2 * i + 1];
        const int64_t b = gw / plane2, blk = gw - b * plane2, Y = blk / a.pe2_nx, X = blk - Y * a.pe2_nx;
        const int64_t c0 = Y * 8 * a.nx + X * 8;
        const float area_l = a.area[(b / a.area_repeat) * plane + c0 + lcol];
        const Tin ps_raw = static_cast<const Tin *>(a.pe1)[(b * (km + 1) + km) * plane + c0 + lcol];
        const char *lvl_col = static_cast<const char *>(a.lvl) + (b * a.cmp_levels * plane2 + blk) * ESZ;
        const float *q[NF];
        float *mean[4] = {nullptr, nullptr, nullptr, nullptr};
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            q[f] = a.q2[f] + gw * (int64_t)kn * 64 + lane;
            mean[f] = a.mean[f] + b * kn * plane2 + blk;
        }
        float cur[kMeanGroup][NF], nxt[kMeanGroup][NF];
        auto request = [&](float (&v)[kMeanGroup][NF], int r0) {
#pragma unroll
            for (int j = 0; j < kMeanGroup; ++j) {
                const int r = (r0 + j < kn) ? r0 + j : kn - 1;
#pragma unroll
                for (int f = 0; f < NF; ++f) v[j][f] = q[f][r * 64];
            }
        };
        request(cur, row0 < kn ? row0 : kn - 1);
        mean_tables<Tin>(lvl_col, (unsigned int)plane2 * ESZ, a.cmp_offset, kn, lane, area_l, ps_raw, m_lvl, m_den, m_ring);
        for (int r0 = row0; r0 < kn; r0 += kMeanGroup) {
            const int nrows = (kn - r0 < kMeanGroup) ? kn - r0 : kMeanGroup;
            if (r0 + kMeanGroup < kn) request(nxt, r0 + kMeanGroup);
#pragma unroll
            for (int j = 0; j < kMeanGroup; ++j)
#pragma unroll
                for (int f = 0; f < NF; ++f) m_ring[(j * NF + f) * kMeanStride + mpos] = cur[j][f];
            mean_reduce<NF>(m_ring, 0, r0, nrows, m_den, mean, plane2, lane);
#pragma unroll
            for (int j = 0; j < kMeanGroup; ++j)
#pragma unroll
                for (int f = 0; f < NF; ++f) cur[j][f] = nxt[j][f];
        }
    }
}

template <typename Tin, int NV, int W>
void launch_mean2(const SweepArgs &a, int64_t n_waves, bool fast, hipStream_t st)
{
#define MEAN_(F)                                                                                                                    \
    hipLaunchKernelGGL((mappm_sweep_kernel<Tin, NV, W, F, true, true>), dim3((unsigned)n_waves), dim3(64),                          \
                       (size_t)(mean_tab_floats(a.kn, (int)sizeof(Tin)) + NV * W * mean_out_rows<Tin, NV * W, F>() * kMeanStride) * sizeof(float), st, a)
    if (fast) MEAN_(true); else MEAN_(false);
#undef MEAN_
}

template <typename Tin>
void launch_mean1(const SweepArgs &a, int nf, int64_t n_waves, bool fast, hipStream_t st)
{
#ifdef FV3HIP_REMAP_SUBSET
    if constexpr (sizeof(Tin) == 8) launch_mean2<Tin, 2, 2>(a, n_waves, fast, st);
#else
    switch (nf) {
        case 1: launch_mean2<Tin, 1, 1>(a, n_waves, fast, st); break;
        case 2: launch_mean2<Tin, 1, 2>(a, n_waves, fast, st); break;
        case 3: launch_mean2<Tin, 3, 1>(a, n_waves, fast, st); break;
        default: launch_mean2<Tin, 2, 2>(a, n_waves, fast, st); break;
    }
#endif
}

#endif  // mean part
}  // namespace

#ifndef FV3HIP_REMAP_PART_SWEEP
bool mappm_mean_eligible(int ny, int nx, int factor, int km, int kn, int kord, int in_dtype)
{
    // (32-bit in the kernel: the byte distance between two levels of an input; everything else it addresses -- the coarse
    // tables, the block-major scratch rows, the means -- goes through 64-bit pointers per wave)
    const int64_t esz = (in_dtype == FV3HIP_F64) ? 8 : 4;
    return factor == 8 && ny > 0 && nx > 0 && ny % 8 == 0 && nx % 8 == 0 && kord <= 3 && km >= 8 && kn >= 1 && kn + 1 <= 128 &&
           (int64_t)ny * nx * esz < ((int64_t)1 << 32);
}

void mappm_mean_launch(const SweepArgs &a, int nf, int in_dtype, int64_t col_end, bool fast, hipStream_t st)
{
    const int64_t n_waves = (col_end - a.col0) / 64;
    if (in_dtype == FV3HIP_F32)
        launch_mean1<float>(a, nf, n_waves, fast, st);
    else
        launch_mean1<double>(a, nf, n_waves, fast, st);
}

void mappm_mean_rest_launch(const SweepArgs &a, int nf, int in_dtype, int64_t n_blocks, hipStream_t st)
{
    const int esz = (in_dtype == FV3HIP_F64) ? 8 : 4;
    const size_t lds = (size_t)(mean_tab_floats(a.kn, esz) + kMeanGroup * 4 * kMeanStride) * sizeof(float);
    const unsigned grid = (unsigned)(n_blocks < 8192 ? (n_blocks < 1 ? 1 : n_blocks) : 8192);
#define REST_(T, N) hipLaunchKernelGGL((mean_rest_kernel<T, N>), dim3(grid), dim3(64), lds, st, a)
#define REST_T(T)                                                                                                                   \
    switch (nf) {                                                                                                                   \
        case 1: REST_(T, 1); break;                                                                                                 \
        case 2: REST_(T, 2); break;                                                                                                 \
        case 3: REST_(T, 3); break;                                                                                                 \
        default: REST_(T, 4); break;                                                                                                \
    }
    if (in_dtype == FV3HIP_F32) { REST_T(float) } else { REST_T(double) }
#undef REST_T
#undef REST_
}

void mappm_mean_redo_launch(const SweepArgs &a, int nf, int in_dtype, hipStream_t st)
{
    const int esz = (in_dtype == FV3HIP_F64) ? 8 : 4;
    const size_t lds = (size_t)(mean_tab_floats(a.kn, esz) + kMeanGroup * kMeanStride + 64 * 3) * sizeof(float);
    if (in_dtype == FV3HIP_F32)
        hipLaunchKernelGGL((mean_redo_kernel<float>), dim3(256), dim3(64), lds, st, a, nf);
    else
        hipLaunchKernelGGL((mean_redo_kernel<double>), dim3(256), dim3(64), lds, st, a, nf);
}

#endif  // mean part
#ifndef FV3HIP_REMAP_PART_MEAN
bool mappm_sweep_eligible(int64_t n_inner, int km, int kn, int kord, int layout, int in_dtype, int64_t pe2_plane)
{
    // What the kernel holds in 32 bits: the byte distance between two levels of an input (row_in), a lane's offset into its
    // result column (offq <= kn result rows of 4-byte values), and -- on the paths that read target interfaces from memory
    // per lane -- the offset of an interface inside the target array of the batch ((kn + 1) rows of pe2_plane values; the
    // target's own plane when it lives on a coarser grid).  Row and batch bases are 64-bit pointers per wave.  A C3072 tile
    // of float64 restarts (9.4 M columns: 75 MB per level) remapped to its coarse grid's levels fits; the same tile remapped
    // to a target on its own grid does not ((kn + 1) x 75 MB) and takes the merge kernels.
    const int64_t esz = (in_dtype == FV3HIP_F64) ? 8 : 4, lim = (int64_t)1 << 32;
    if (pe2_plane <= 0) pe2_plane = n_inner;
    return layout == FV3HIP_LAYOUT_LEVEL_COL && kord <= 3 && km >= 8 && kn >= 1 && n_inner > 0 && n_inner % 64 == 0 &&
           n_inner * esz < lim && (int64_t)(kn + 1) * n_inner * 4 < lim && (int64_t)(kn + 2) * pe2_plane * esz < lim;
}

void mappm_sweep_launch(const SweepArgs &a, int nf, int in_dtype, int64_t col_end, bool fast, hipStream_t st)
{
    const int64_t n_waves = (col_end - a.col0) / 64;
    // the whole target table in LDS: coarser by a power of two >= 8, waves inside one fine row, (kn + 1) * 8 <= 1024 floats
    const bool tgt = a.pe2_f >= 8 && a.pe2_f <= 64 && (a.pe2_f & (a.pe2_f - 1)) == 0 && a.nx % 64 == 0 && a.kn + 1 <= 128;
    if (in_dtype == FV3HIP_F32)
        launch_sweep1<float>(a, nf, n_waves, fast, tgt, st);
    else
        launch_sweep1<double>(a, nf, n_waves, fast, tgt, st);
}

#endif  // sweep part
}  // namespace fv3hip
