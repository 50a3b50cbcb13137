// Horizontal block coarse-graining kernels for gfx950 (MI355X).
//
// Reference behaviour restated (paths relative to the reference checkout):
//   external/vcm/vcm/cubedsphere/coarsen.py:183-218  weighted_block_average
//   external/vcm/vcm/cubedsphere/coarsen.py:221-273  edge_weighted_block_average
//   external/vcm/vcm/cubedsphere/coarsen.py:795-840  block_coarsen (sum/min/max/mean)
//   external/vcm/vcm/cubedsphere/coarsen.py:557-588, 750-786  block_median, _block_mode
//   external/vcm/vcm/cubedsphere/coarsen.py:869-938  block_upsample(_like)
//
// All of these are HBM-bound: the fast path streams each fine-grid row with one 16-byte load
// per lane (a wave covers 1 KiB of a row), keeps the 2-D weights of its f x f blocks in
// registers while it walks the vertical levels that share them, and finishes a block with a
// couple of cross-lane adds.  There is no data reuse to stage through LDS.
#include "common.h"

namespace fv3hip {
namespace {

template <typename T>
__device__ __forceinline__ bool is_nan(T x)
{
    return x != x;
}

// N elements of T as one register tuple; 16-byte ones become global_load_dwordx4.
template <typename T, int N>
struct VecOf {
    typedef T type __attribute__((ext_vector_type(N)));
};
template <typename T, int N>
using Vec = typename VecOf<T, N>::type;

template <typename A, typename B>
struct Promote {
    using type = float;
};
template <>
struct Promote<double, double> {
    using type = double;
};
template <>
struct Promote<double, float> {
    using type = double;
};
template <>
struct Promote<float, double> {
    using type = double;
};

// ---------------------------------------------------------------------------------------
// Generic windowed weighted mean: one thread per output element, any window / stride.
// Used for edge-weighted averages and for shapes the fast path does not cover.
// ---------------------------------------------------------------------------------------
template <typename To, typename Tw>
__global__ void wavg_generic_kernel(const To *__restrict__ obj, const Tw *__restrict__ w,
                                    typename Promote<To, Tw>::type *__restrict__ out,
                                    int64_t n_outer, int ny, int nx, int64_t w_repeat, int by,
                                    int bx, int sy, int sx, int nyo, int nxo)
{
    using P = typename Promote<To, Tw>::type;
    const int64_t total = n_outer * nyo * nxo;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(idx % nxo);
        const int64_t t = idx / nxo;
        const int Y = (int)(t % nyo);
        const int64_t o = t / nyo;
        const int64_t sp = (int64_t)Y * sy * nx + (int64_t)X * sx;
        const To *po = obj + o * ny * (int64_t)nx + sp;
        const Tw *pw = w + (o / w_repeat) * ny * (int64_t)nx + sp;
        // the denominator is the block sum of the weights alone, accumulated in the weights' own
        // dtype as `weights.coarsen(...).sum()` does (coarsen.py:212), then promoted
        P num = 0;
        Tw den = 0;
        for (int dy = 0; dy < by; ++dy) {
            for (int dx = 0; dx < bx; ++dx) {
                const Tw w0 = pw[(int64_t)dy * nx + dx];
                const P p = (P)po[(int64_t)dy * nx + dx] * (P)w0;
                if (!is_nan(p)) num += p;
                if (!is_nan(w0)) den += w0;
            }
        }
        out[idx] = num / (P)den;
    }
}

// ---------------------------------------------------------------------------------------
// Fast path: full F x F blocks, rows read with 16-byte loads.
//   VEC  = elements of the promoted type in 16 bytes (4 for f32, 2 for f64)
//   LPB  = lanes that share one block row  (F / VEC, when F >= VEC)
//   BPL  = blocks held by one lane         (VEC / F, when F <  VEC)
// Thread s of a spatial slab handles vector column xv = s % (nx/VEC) of block row
// Y = s / (nx/VEC); gridDim.y walks (weight slice, z-chunk) pairs.
// ---------------------------------------------------------------------------------------
template <typename To, typename Tw, int F>
__global__ __launch_bounds__(256) void wavg_block_kernel(
    const To *__restrict__ obj, const Tw *__restrict__ w,
    typename Promote<To, Tw>::type *__restrict__ out, int64_t n_outer, int ny, int nx,
    int64_t w_repeat, int zsplit, int64_t n_gy)
{
    using P = typename Promote<To, Tw>::type;
    constexpr int VEC = 16 / sizeof(P);
    constexpr int LPB = (F >= VEC) ? F / VEC : 1;
    constexpr int BPL = (F >= VEC) ? 1 : VEC / F;
    constexpr int EPB = VEC / BPL;  // elements of one lane's vector that fall in one block
    constexpr bool kCacheW = (F * VEC <= 64);

    const int XV = nx / VEC;
    const int nyo = ny / F, nxo = nx / F;
    const int64_t S = (int64_t)nyo * XV;
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool active = s < S;
    const int Y = active ? (int)(s / XV) : 0;
    const int xv = active ? (int)(s % XV) : 0;
    const int64_t sp = (int64_t)Y * F * nx + (int64_t)xv * VEC;  // offset inside one slice
    const int64_t slice = (int64_t)ny * nx;
    const int64_t zper = (w_repeat + zsplit - 1) / zsplit;

    for (int64_t gy = blockIdx.y; gy < n_gy; gy += gridDim.y) {
        const int64_t g = gy / zsplit;
        const int part = (int)(gy % zsplit);
        const int64_t o_begin = g * w_repeat + part * zper;
        int64_t o_end = o_begin + zper;
        if (o_end > (g + 1) * w_repeat) o_end = (g + 1) * w_repeat;
        if (o_end > n_outer) o_end = n_outer;

        const Tw *pw = w + g * slice + sp;
        P wreg[kCacheW ? F : 1][VEC];
        Tw den[BPL];  // accumulated in the weights' dtype, like weights.coarsen(...).sum()
#pragma unroll
        for (int b = 0; b < BPL; ++b) den[b] = 0;
#pragma unroll
        for (int dy = 0; dy < F; ++dy) {
            Vec<Tw, VEC> wv;
            if (active) {
                wv = *reinterpret_cast<const Vec<Tw, VEC> *>(pw + (int64_t)dy * nx);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) wv[e] = 0;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const Tw w0 = wv[e];
                if (kCacheW) wreg[kCacheW ? dy : 0][e] = (P)w0;
                den[e / EPB] += is_nan(w0) ? (Tw)0 : w0;
            }
        }
#pragma unroll
        for (int m = 1; m < LPB; m <<= 1) den[0] += __shfl_xor(den[0], m);

        for (int64_t o = o_begin; o < o_end; ++o) {
            const To *po = obj + o * slice + sp;
            Vec<To, VEC> ov[F];
#pragma unroll
            for (int dy = 0; dy < F; ++dy) {
                if (active) {
                    ov[dy] = __builtin_nontemporal_load(
                        reinterpret_cast<const Vec<To, VEC> *>(po + (int64_t)dy * nx));
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ov[dy][e] = 0;
                }
            }
            P num[BPL];
#pragma unroll
            for (int b = 0; b < BPL; ++b) num[b] = 0;
#pragma unroll
            for (int dy = 0; dy < F; ++dy) {
                Vec<Tw, VEC> wv;
                if (!kCacheW) {
                    if (active) {
                        wv = *reinterpret_cast<const Vec<Tw, VEC> *>(pw + (int64_t)dy * nx);
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) wv[e] = 0;
                    }
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const P ww = kCacheW ? wreg[kCacheW ? dy : 0][e] : (P)wv[e];
                    const P p = (P)ov[dy][e] * ww;
                    num[e / EPB] += is_nan(p) ? (P)0 : p;
                }
            }
#pragma unroll
            for (int m = 1; m < LPB; m <<= 1) num[0] += __shfl_xor(num[0], m);
            if (active && (xv % LPB) == 0) {
                P *dst = out + (o * nyo + Y) * (int64_t)nxo + (int64_t)(xv / LPB) * BPL;
                if (BPL == 1) {
                    dst[0] = num[0] / (P)den[0];
                } else {
                    Vec<P, BPL> r;
#pragma unroll
                    for (int b = 0; b < BPL; ++b) r[b] = num[b] / (P)den[b];
                    *reinterpret_cast<Vec<P, BPL> *>(dst) = r;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Mass-weighted block average of NF fields that share their weights delp * area
// (coarsen_restarts.py:335-427, 856-900: W, T, ua, va and the eight non-fraction tracers are averaged with the
// weights `delp * area`).  The product is formed in registers -- never materialised -- and read once for up to
// four fields: per launch the traffic is delp + NF fields (+ the 2-D area once per z-chunk) instead of
// NF x (field + a materialised 3-D weight).  Arithmetic as the reference's: w = delp * area in the promoted type,
// numerator nansum(obj * w), denominator nansum(w) in the weights' (promoted) type.
// ---------------------------------------------------------------------------------------
constexpr int kMassFields = 4;
struct MassFieldPtrs {
    const void *obj[kMassFields];
    void *out[kMassFields];
};

template <typename Tf, typename Ta, int F, int NF>
__global__ __launch_bounds__(256) void mass_wavg_block_kernel(
    const MassFieldPtrs fp, const Tf *__restrict__ delp, const Ta *__restrict__ area, int64_t n_outer, int ny, int nx,
    int64_t a_repeat, int zsplit, int64_t n_gy)
{
    using P = typename Promote<Tf, Ta>::type;
    constexpr int VEC = 16 / sizeof(P);
    constexpr int LPB = (F >= VEC) ? F / VEC : 1;
    constexpr int BPL = (F >= VEC) ? 1 : VEC / F;
    constexpr int EPB = VEC / BPL;

    const int XV = nx / VEC;
    const int nyo = ny / F, nxo = nx / F;
    const int64_t S = (int64_t)nyo * XV;
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool active = s < S;
    const int Y = active ? (int)(s / XV) : 0;
    const int xv = active ? (int)(s % XV) : 0;
    const int64_t sp = (int64_t)Y * F * nx + (int64_t)xv * VEC;
    const int64_t slice = (int64_t)ny * nx;
    const int64_t zper = (a_repeat + zsplit - 1) / zsplit;

    for (int64_t gy = blockIdx.y; gy < n_gy; gy += gridDim.y) {
        const int64_t g = gy / zsplit;
        const int part = (int)(gy % zsplit);
        const int64_t o_begin = g * a_repeat + part * zper;
        int64_t o_end = o_begin + zper;
        if (o_end > (g + 1) * a_repeat) o_end = (g + 1) * a_repeat;
        if (o_end > n_outer) o_end = n_outer;
        // the block's area stays in registers over the levels that share it
        P areg[F][VEC];
#pragma unroll
        for (int dy = 0; dy < F; ++dy) {
            Vec<Ta, VEC> av;
            if (active) {
                av = *reinterpret_cast<const Vec<Ta, VEC> *>(area + g * slice + sp + (int64_t)dy * nx);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) av[e] = 0;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) areg[dy][e] = (P)av[e];
        }
        for (int64_t o = o_begin; o < o_end; ++o) {
            P wreg[F][VEC];
            P den[BPL];
#pragma unroll
            for (int b = 0; b < BPL; ++b) den[b] = 0;
#pragma unroll
            for (int dy = 0; dy < F; ++dy) {
                Vec<Tf, VEC> dv;
                if (active && delp) {
                    dv = __builtin_nontemporal_load(reinterpret_cast<const Vec<Tf, VEC> *>(delp + o * slice + sp + (int64_t)dy * nx));
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) dv[e] = 0;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    // (delp == nullptr: plain 2-D weights shared by the NF fields -- the surface-data means)
                    const P w0 = delp ? (P)dv[e] * areg[dy][e] : (active ? areg[dy][e] : (P)0);
                    wreg[dy][e] = w0;
                    den[e / EPB] += is_nan(w0) ? (P)0 : w0;
                }
            }
#pragma unroll
            for (int m = 1; m < LPB; m <<= 1) den[0] += __shfl_xor(den[0], m);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const Tf *po = static_cast<const Tf *>(fp.obj[f]) + o * slice + sp;
                Vec<Tf, VEC> ov[F];
#pragma unroll
                for (int dy = 0; dy < F; ++dy) {
                    if (active) {
                        ov[dy] = __builtin_nontemporal_load(reinterpret_cast<const Vec<Tf, VEC> *>(po + (int64_t)dy * nx));
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) ov[dy][e] = 0;
                    }
                }
                P num[BPL];
#pragma unroll
                for (int b = 0; b < BPL; ++b) num[b] = 0;
#pragma unroll
                for (int dy = 0; dy < F; ++dy) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const P p = (P)ov[dy][e] * wreg[dy][e];
                        num[e / EPB] += is_nan(p) ? (P)0 : p;
                    }
                }
#pragma unroll
                for (int m = 1; m < LPB; m <<= 1) num[0] += __shfl_xor(num[0], m);
                if (active && (xv % LPB) == 0) {
                    P *dst = static_cast<P *>(fp.out[f]) + (o * nyo + Y) * (int64_t)nxo + (int64_t)(xv / LPB) * BPL;
                    if (BPL == 1) {
                        dst[0] = num[0] / den[0];
                    } else {
                        Vec<P, BPL> r;
#pragma unroll
                        for (int b = 0; b < BPL; ++b) r[b] = num[b] / den[b];
                        *reinterpret_cast<Vec<P, BPL> *>(dst) = r;
                    }
                }
            }
        }
    }
}

template <typename Tf, typename Ta, int F>
int launch_mass_wavg(const MassFieldPtrs &fp, int nf, const Tf *delp, const Ta *area, int64_t n_outer, int ny, int nx,
                     int64_t a_repeat, hipStream_t stream)
{
    using P = typename Promote<Tf, Ta>::type;
    constexpr int VEC = 16 / sizeof(P);
    const int64_t S = (int64_t)(ny / F) * (nx / VEC);
    const int64_t n_groups = n_outer / a_repeat;
    const int64_t gx = ceil_div(S, 256);
    int zsplit = 1;
    const int64_t want_blocks = 256 * 16;
    while (gx * n_groups * zsplit < want_blocks && (a_repeat / (zsplit * 2)) >= 4) zsplit *= 2;
    const int64_t n_gy = n_groups * zsplit;
    dim3 grid((unsigned)gx, (unsigned)(n_gy < 65535 ? n_gy : 65535));
    switch (nf) {
        case 1: hipLaunchKernelGGL((mass_wavg_block_kernel<Tf, Ta, F, 1>), grid, dim3(256), 0, stream, fp, delp, area, n_outer, ny, nx, a_repeat, zsplit, n_gy); break;
        case 2: hipLaunchKernelGGL((mass_wavg_block_kernel<Tf, Ta, F, 2>), grid, dim3(256), 0, stream, fp, delp, area, n_outer, ny, nx, a_repeat, zsplit, n_gy); break;
        case 3: hipLaunchKernelGGL((mass_wavg_block_kernel<Tf, Ta, F, 3>), grid, dim3(256), 0, stream, fp, delp, area, n_outer, ny, nx, a_repeat, zsplit, n_gy); break;
        default: hipLaunchKernelGGL((mass_wavg_block_kernel<Tf, Ta, F, 4>), grid, dim3(256), 0, stream, fp, delp, area, n_outer, ny, nx, a_repeat, zsplit, n_gy); break;
    }
    return check_launch("mass_wavg_block_kernel");
}

template <typename Tf, typename Ta>
int dispatch_mass_wavg(const void *const *fields, int n_fields, const void *delp_, const void *area_, int64_t n_outer, int ny,
                       int nx, int64_t a_repeat, int factor, void *const *outs, hipStream_t stream)
{
    using P = typename Promote<Tf, Ta>::type;
    constexpr int VEC = 16 / sizeof(P);
    const Tf *delp = static_cast<const Tf *>(delp_);
    const Ta *area = static_cast<const Ta *>(area_);
    bool ok = (nx % VEC == 0) && (!delp || reinterpret_cast<uintptr_t>(delp) % (sizeof(Tf) * VEC) == 0) &&
              (reinterpret_cast<uintptr_t>(area) % (sizeof(Ta) * VEC) == 0);
    for (int f = 0; f < n_fields && ok; ++f)
        ok = (reinterpret_cast<uintptr_t>(fields[f]) % (sizeof(Tf) * VEC) == 0) && (reinterpret_cast<uintptr_t>(outs[f]) % 16 == 0);
    if (!ok || !(factor == 2 || factor == 4 || factor == 8 || factor == 16))
        return fail(FV3HIP_EUNSUPPORTED, "fused mass-weighted average needs factor in {2, 4, 8, 16}, nx %% %d == 0 and 16-byte aligned arrays",
                    VEC);
    for (int f0 = 0; f0 < n_fields; f0 += kMassFields) {
        const int nf = (n_fields - f0 < kMassFields) ? n_fields - f0 : kMassFields;
        MassFieldPtrs fp;
        for (int f = 0; f < kMassFields; ++f) {
            fp.obj[f] = f < nf ? fields[f0 + f] : nullptr;
            fp.out[f] = f < nf ? outs[f0 + f] : nullptr;
        }
        int rc;
        switch (factor) {
            case 2: rc = launch_mass_wavg<Tf, Ta, 2>(fp, nf, delp, area, n_outer, ny, nx, a_repeat, stream); break;
            case 4: rc = launch_mass_wavg<Tf, Ta, 4>(fp, nf, delp, area, n_outer, ny, nx, a_repeat, stream); break;
            case 8: rc = launch_mass_wavg<Tf, Ta, 8>(fp, nf, delp, area, n_outer, ny, nx, a_repeat, stream); break;
            default: rc = launch_mass_wavg<Tf, Ta, 16>(fp, nf, delp, area, n_outer, ny, nx, a_repeat, stream); break;
        }
        if (rc) return rc;
    }
    return FV3HIP_OK;
}

template <typename To, typename Tw, int F>
int launch_wavg_block(const To *obj, const Tw *w, void *out, int64_t n_outer, int ny, int nx,
                      int64_t w_repeat, hipStream_t stream)
{
    using P = typename Promote<To, Tw>::type;
    constexpr int VEC = 16 / sizeof(P);
    const int64_t S = (int64_t)(ny / F) * (nx / VEC);
    const int64_t n_groups = n_outer / w_repeat;
    const int64_t gx = ceil_div(S, 256);
    // Split the levels that share a weight slice until the grid fills the chip several times
    // over (256 CUs x 8 blocks of 256 threads), but keep chunks long enough to amortise the
    // weight loads.
    int zsplit = 1;
    const int64_t want_blocks = 256 * 16;
    while (gx * n_groups * zsplit < want_blocks && (w_repeat / (zsplit * 2)) >= 4) zsplit *= 2;
    const int64_t n_gy = n_groups * zsplit;
    dim3 grid((unsigned)gx, (unsigned)(n_gy < 65535 ? n_gy : 65535));
    hipLaunchKernelGGL((wavg_block_kernel<To, Tw, F>), grid, dim3(256), 0, stream, obj, w,
                       reinterpret_cast<P *>(out), n_outer, ny, nx, w_repeat, zsplit, n_gy);
    return check_launch("wavg_block_kernel");
}

template <typename To, typename Tw>
int dispatch_wavg(const void *obj_, const void *w_, void *out, int64_t n_outer, int ny, int nx,
                  int64_t w_repeat, int by, int bx, int sy, int sx, hipStream_t stream)
{
    using P = typename Promote<To, Tw>::type;
    constexpr int VEC = 16 / sizeof(P);
    const To *obj = static_cast<const To *>(obj_);
    const Tw *w = static_cast<const Tw *>(w_);
    const bool full_blocks = (by == bx && sy == by && sx == bx);
    const bool vec_ok = (nx % VEC == 0) && (nx % bx == 0) && (ny % by == 0) &&
                        ((reinterpret_cast<uintptr_t>(obj) % (sizeof(To) * VEC)) == 0) &&
                        ((reinterpret_cast<uintptr_t>(w) % (sizeof(Tw) * VEC)) == 0) &&
                        ((reinterpret_cast<uintptr_t>(out) % 16) == 0);
    if (full_blocks && vec_ok) {
        switch (by) {
            case 2: return launch_wavg_block<To, Tw, 2>(obj, w, out, n_outer, ny, nx, w_repeat, stream);
            case 4: return launch_wavg_block<To, Tw, 4>(obj, w, out, n_outer, ny, nx, w_repeat, stream);
            case 8: return launch_wavg_block<To, Tw, 8>(obj, w, out, n_outer, ny, nx, w_repeat, stream);
            case 16: return launch_wavg_block<To, Tw, 16>(obj, w, out, n_outer, ny, nx, w_repeat, stream);
            case 32: return launch_wavg_block<To, Tw, 32>(obj, w, out, n_outer, ny, nx, w_repeat, stream);
            default: break;
        }
    }
    const int nyo = (ny - by) / sy + 1, nxo = (nx - bx) / sx + 1;
    const int64_t total = n_outer * nyo * nxo;
    if (total == 0) return FV3HIP_OK;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL((wavg_generic_kernel<To, Tw>), dim3((unsigned)blocks), dim3(256), 0, stream,
                       obj, w, reinterpret_cast<P *>(out), n_outer, ny, nx, w_repeat, by, bx, sy,
                       sx, nyo, nxo);
    return check_launch("wavg_generic_kernel");
}

int wavg_entry(const void *obj, int obj_dtype, const void *w, int w_dtype, int64_t n_outer, int ny,
               int nx, int64_t w_repeat, int by, int bx, int sy, int sx, void *out, void *stream)
{
    FV3HIP_REQUIRE(obj_dtype == FV3HIP_F32 || obj_dtype == FV3HIP_F64,
                   "obj dtype must be F32 or F64, got %d", obj_dtype);
    FV3HIP_REQUIRE(w_dtype == FV3HIP_F32 || w_dtype == FV3HIP_F64,
                   "weights dtype must be F32 or F64, got %d", w_dtype);
    FV3HIP_REQUIRE(n_outer >= 0 && ny >= 0 && nx >= 0, "negative extent");
    FV3HIP_REQUIRE(w_repeat >= 1, "w_repeat must be >= 1, got %lld", (long long)w_repeat);
    FV3HIP_REQUIRE(n_outer % w_repeat == 0, "n_outer (%lld) is not a multiple of w_repeat (%lld)",
                   (long long)n_outer, (long long)w_repeat);
    if (n_outer == 0 || ny == 0 || nx == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(obj && w && out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (obj_dtype == FV3HIP_F32 && w_dtype == FV3HIP_F32)
        return dispatch_wavg<float, float>(obj, w, out, n_outer, ny, nx, w_repeat, by, bx, sy, sx, st);
    if (obj_dtype == FV3HIP_F64 && w_dtype == FV3HIP_F64)
        return dispatch_wavg<double, double>(obj, w, out, n_outer, ny, nx, w_repeat, by, bx, sy, sx, st);
    if (obj_dtype == FV3HIP_F64 && w_dtype == FV3HIP_F32)
        return dispatch_wavg<double, float>(obj, w, out, n_outer, ny, nx, w_repeat, by, bx, sy, sx, st);
    return dispatch_wavg<float, double>(obj, w, out, n_outer, ny, nx, w_repeat, by, bx, sy, sx, st);
}

// ---------------------------------------------------------------------------------------
// Generic block reductions (sum / mean / min / max / median / mode).
// ---------------------------------------------------------------------------------------
template <typename T>
struct IsFloat {
    static constexpr bool value = false;
};
template <>
struct IsFloat<float> {
    static constexpr bool value = true;
};
template <>
struct IsFloat<double> {
    static constexpr bool value = true;
};

template <typename T>
__device__ __forceinline__ bool nan_of(T x)
{
    if constexpr (IsFloat<T>::value) return x != x;
    return false;
}

template <typename T>
__device__ __forceinline__ T quiet_nan()
{
    if constexpr (sizeof(T) == 4) return (T)__builtin_nanf("");
    return (T)__builtin_nan("");
}

template <typename T>
__global__ void block_reduce_kernel(const T *__restrict__ in, T *__restrict__ out, int64_t n_outer,
                                    int ny, int nx, int by, int bx, int sy, int sx, int nyo, int nxo,
                                    int op, int nan_policy)
{
    const int64_t total = n_outer * nyo * nxo;
    const int n = by * bx;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(idx % nxo);
        const int64_t t = idx / nxo;
        const int Y = (int)(t % nyo);
        const int64_t o = t / nyo;
        const T *p = in + o * ny * (int64_t)nx + (int64_t)Y * sy * nx + (int64_t)X * sx;
        auto at = [&](int i) { return p[(int64_t)(i / bx) * nx + (i % bx)]; };
        T result;
        bool poisoned = false;  // NAN_PROPAGATE for sum/mean/min/max: numpy's plain reductions
        if (nan_policy == FV3HIP_NAN_PROPAGATE && op <= FV3HIP_OP_MAX) {
            for (int i = 0; i < n; ++i) poisoned |= nan_of(at(i));
        }
        if (poisoned) {
            result = quiet_nan<T>();
        } else if (op == FV3HIP_OP_SUM || op == FV3HIP_OP_MEAN) {
            T acc = 0;
            int cnt = 0;
            for (int i = 0; i < n; ++i) {
                const T v = at(i);
                if (!nan_of(v)) {
                    acc += v;
                    ++cnt;
                }
            }
            if (op == FV3HIP_OP_MEAN) {
                if constexpr (IsFloat<T>::value)
                    result = acc / (T)cnt;  // 0/0 = NaN for an all-NaN block (numpy.nanmean)
                else
                    result = acc;
            } else {
                result = acc;
            }
        } else if (op == FV3HIP_OP_MIN || op == FV3HIP_OP_MAX) {
            bool have = false;
            T best = 0;
            for (int i = 0; i < n; ++i) {
                const T v = at(i);
                if (nan_of(v)) continue;
                if (!have || (op == FV3HIP_OP_MIN ? v < best : v > best)) best = v;
                have = true;
            }
            if constexpr (IsFloat<T>::value)
                result = have ? best : quiet_nan<T>();
            else
                result = best;
        } else if (op == FV3HIP_OP_MEDIAN) {
            // numpy.median: NaN if any NaN, else the mean of the two middle order statistics.
            bool any_nan = false;
            for (int i = 0; i < n; ++i) any_nan |= nan_of(at(i));
            if (any_nan) {
                result = quiet_nan<T>();
            } else {
                const int r1 = (n - 1) / 2, r2 = n / 2;
                T m1 = 0, m2 = 0;
                for (int i = 0; i < n; ++i) {
                    const T v = at(i);
                    int less = 0, eq = 0;
                    for (int j = 0; j < n; ++j) {
                        const T u = at(j);
                        less += (u < v);
                        eq += (u == v);
                    }
                    if (less <= r1 && r1 < less + eq) m1 = v;
                    if (less <= r2 && r2 < less + eq) m2 = v;
                }
                if constexpr (IsFloat<T>::value)
                    result = (r1 == r2) ? m1 : (m1 + m2) / (T)2;
                else
                    result = m1;
            }
        } else {
            // scipy.stats.mode (1.7.3): the most frequent non-NaN value, smallest value on ties.
            int best_cnt = 0;
            T best = 0;
            for (int i = 0; i < n; ++i) {
                const T v = at(i);
                if (nan_of(v)) continue;
                int eq = 0;
                for (int j = 0; j < n; ++j) eq += (at(j) == v);
                if (eq > best_cnt || (eq == best_cnt && v < best)) {
                    best_cnt = eq;
                    best = v;
                }
            }
            if constexpr (IsFloat<T>::value) {
                // all-NaN block: "propagate" gives 0.0 in scipy 1.7.3 (count 0); for "omit"
                // the masked result is reported as NaN here.
                result = (best_cnt > 0) ? best
                                        : (nan_policy == FV3HIP_NAN_OMIT ? quiet_nan<T>() : (T)0);
            } else {
                result = best;
            }
        }
        out[idx] = result;
    }
}

// median / mode of blocks of at most 64 values (f <= 8): one wavefront per output block, lane i holds
// element i, and the order statistics come from 64 rounds of a wave-wide broadcast instead of the
// n^2 strided global loads of the one-thread-per-block kernel.  Same results to the bit, including the
// tie rules: mode = the most frequent non-NaN value, the smallest on ties, the first in block order among
// values that compare equal (-0.0 / 0.0); median = NaN if any NaN, else the mean of the two middle order
// statistics, each taken from the last element in block order that holds that rank.
template <typename T>
__global__ __launch_bounds__(256) void block_rank_wave_kernel(const T *__restrict__ in, T *__restrict__ out, int64_t n_outer,
                                                              int ny, int nx, int by, int bx, int sy, int sx, int nyo,
                                                              int nxo, int op, int nan_policy)
{
    const int64_t total = n_outer * nyo * nxo;
    const int n = by * bx;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t idx = wave0; idx < total; idx += n_waves) {
        const int X = (int)(idx % nxo);
        const int64_t t = idx / nxo;
        const int Y = (int)(t % nyo);
        const int64_t o = t / nyo;
        const T *p = in + o * ny * (int64_t)nx + (int64_t)Y * sy * nx + (int64_t)X * sx;
        const bool present = lane < n;
        const T v = present ? p[(int64_t)(lane / bx) * nx + (lane % bx)] : (T)0;
        const bool is_nan = present && nan_of(v);
        int less = 0, eq = 0;
        for (int j = 0; j < n; ++j) {
            const T u = __shfl(v, j, 64);
            less += (u < v);
            eq += (u == v);
        }
        T result;
        if (op == FV3HIP_OP_MEDIAN) {
            const bool any_nan = __ballot(is_nan) != 0ull;
            const int r1 = (n - 1) / 2, r2 = n / 2;
            const unsigned long long h1 = __ballot(present && less <= r1 && r1 < less + eq);
            const unsigned long long h2 = __ballot(present && less <= r2 && r2 < less + eq);
            const T m1 = __shfl(v, h1 ? 63 - __clzll(h1) : 0, 64), m2 = __shfl(v, h2 ? 63 - __clzll(h2) : 0, 64);
            if (any_nan) {
                result = quiet_nan<T>();
            } else {
                if constexpr (IsFloat<T>::value)
                    result = (r1 == r2) ? m1 : (m1 + m2) / (T)2;
                else
                    result = m1;
            }
        } else {
            int cnt = (present && !is_nan) ? eq : 0;
            T best = v;
            int who = lane;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const int c2 = __shfl_xor(cnt, d, 64);
                const T v2 = __shfl_xor(best, d, 64);
                const int w2 = __shfl_xor(who, d, 64);
                // (cnt desc, value asc, block order asc); a lane with cnt == 0 never wins against cnt > 0
                const bool take = (c2 > cnt) || (c2 == cnt && c2 > 0 && (v2 < best || (!(best < v2) && w2 < who)));
                if (take) {
                    cnt = c2;
                    best = v2;
                    who = w2;
                }
            }
            if constexpr (IsFloat<T>::value)
                result = (cnt > 0) ? best : (nan_policy == FV3HIP_NAN_OMIT ? quiet_nan<T>() : (T)0);
            else
                result = best;
        }
        if (lane == 0) out[idx] = result;
    }
}

template <typename T>
int launch_block_reduce(const void *in, void *out, int64_t n_outer, int ny, int nx, int by, int bx,
                        int sy, int sx, int op, int nan_policy, hipStream_t stream)
{
    const int nyo = (ny - by) / sy + 1, nxo = (nx - bx) / sx + 1;
    const int64_t total = n_outer * nyo * nxo;
    if (total <= 0) return FV3HIP_OK;
    if ((op == FV3HIP_OP_MEDIAN || op == FV3HIP_OP_MODE) && by * bx <= 64) {
        int64_t wblocks = ceil_div(total, 4);  // one wavefront per output block
        if (wblocks > 256 * 64) wblocks = 256 * 64;
        hipLaunchKernelGGL((block_rank_wave_kernel<T>), dim3((unsigned)wblocks), dim3(256), 0, stream,
                           static_cast<const T *>(in), static_cast<T *>(out), n_outer, ny, nx, by, bx, sy, sx, nyo, nxo, op,
                           nan_policy);
        return check_launch("block_rank_wave_kernel");
    }
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL((block_reduce_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream,
                       static_cast<const T *>(in), static_cast<T *>(out), n_outer, ny, nx, by, bx,
                       sy, sx, nyo, nxo, op, nan_policy);
    return check_launch("block_reduce_kernel");
}

// ---------------------------------------------------------------------------------------
// block_upsample: out[o][y][x] = in[o][y / f][x / f]
// ---------------------------------------------------------------------------------------
template <typename U>
__global__ void upsample_kernel(const U *__restrict__ in, U *__restrict__ out, int64_t n_outer,
                                int ny_in, int nx_in, int ny_out, int nx_out, int fy, int fx)
{
    const int64_t total = n_outer * ny_out * nx_out;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % nx_out);
        const int64_t t = idx / nx_out;
        const int y = (int)(t % ny_out);
        const int64_t o = t / ny_out;
        out[idx] = in[(o * ny_in + y / fy) * nx_in + x / fx];
    }
}


// The same by rows: a (column chunk, output row) grid, VEC consecutive outputs per thread stored as one 16-byte piece -- no
// 64-bit division per element (the kernel above spends its time on two of them per value: 3.0 TB/s of output on a C3072
// tile), one 32-bit division per thread.  Rows must be whole vectors (nx_out % VEC == 0, 16-byte aligned arrays).
template <typename U, int VEC>
__global__ __launch_bounds__(256) void upsample_rows_kernel(const U *__restrict__ in, U *__restrict__ out, int64_t n_rows,
                                                            int ny_in, int nx_in, int ny_out, int nx_out, int fy, int fx)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (x0 >= nx_out) return;
    const int xi0 = x0 / fx;
    int rem = x0 - xi0 * fx;   // position of x0 inside its input cell
    for (int64_t row = blockIdx.y; row < n_rows; row += gridDim.y) {
        const int64_t o = row / ny_out;
        const int y = (int)(row - o * ny_out);
        const U *src = in + (o * ny_in + y / fy) * (int64_t)nx_in;
        Vec<U, VEC> v;
        int xi = xi0, r = rem;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[e] = src[xi < nx_in ? xi : nx_in - 1];
            if (++r == fx) {
                r = 0;
                ++xi;
            }
        }
        *reinterpret_cast<Vec<U, VEC> *>(out + row * (int64_t)nx_out + x0) = v;
    }
}

template <typename U>
static void launch_upsample(const U *in, U *out, int64_t n_outer, int ny_in, int nx_in, int ny_out, int nx_out, int fy, int fx,
                            hipStream_t st)
{
    constexpr int VEC = 16 / sizeof(U);
    const int64_t n_rows = n_outer * ny_out;
    if (nx_out % VEC == 0 && nx_out >= 256 && reinterpret_cast<uintptr_t>(out) % 16 == 0) {
        const dim3 grid((unsigned)ceil_div(nx_out, 256 * VEC), (unsigned)(n_rows < 65535 ? n_rows : 65535));
        hipLaunchKernelGGL((upsample_rows_kernel<U, VEC>), grid, dim3(256), 0, st, in, out, n_rows, ny_in, nx_in, ny_out, nx_out, fy, fx);
        return;
    }
    const int64_t total = n_rows * nx_out;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL((upsample_kernel<U>), dim3((unsigned)blocks), dim3(256), 0, st, in, out, n_outer, ny_in, nx_in, ny_out, nx_out, fy, fx);
}

// ---------------------------------------------------------------------------------------
// Cell centres -> cell edges across the cube (external/vcm/vcm/cubedsphere/xgcm.py:7-34,
// regridz.py:123-135: xgcm.Grid.interp(delp, axis) with FV3_FACE_CONNECTIONS)
// ---------------------------------------------------------------------------------------
// rows[t][e][o][j]: the four boundary vectors of tile t, indexed along the edge:
//   e = 0: x = 0 (j = y), 1: x = n-1 (j = y), 2: y = 0 (j = x), 3: y = n-1 (j = x)
template <typename U>
__global__ void cube_edge_rows_kernel(const U *__restrict__ in, U *__restrict__ rows, int n_tiles, int64_t n_mid,
                                      int n)
{
    const int64_t total = (int64_t)n_tiles * 4 * n_mid * n;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % n);
        int64_t r = idx / n;
        const int64_t o = r % n_mid;
        r /= n_mid;
        const int e = (int)(r % 4);
        const int64_t t = r / 4;
        const int y = (e == 0 || e == 1) ? j : (e == 2 ? 0 : n - 1);
        const int x = (e == 2 || e == 3) ? j : (e == 0 ? 0 : n - 1);
        rows[idx] = in[((t * n_mid + o) * n + y) * n + x];
    }
}

// lo / hi halo vectors of the listed tiles from the table of ALL tiles' boundary vectors (cube_edge_rows_kernel):
// out[side][i][o][j] = rows[nbr(i, side)][row(i, side)][o][flip(i, side) ? n - 1 - j : j] -- the pick-and-orient step of
// xgcm's face connections (xgcm.py:7-34) without torch flip / stack kernels.
struct HaloPick {
    int nbr[2][6], row[2][6], flip[2][6];
};

template <typename U>
__global__ void halo_pick_kernel(const U *__restrict__ rows, U *__restrict__ out, const HaloPick hp, int n_local, int64_t n_mid, int n)
{
    const int64_t total = 2 * (int64_t)n_local * n_mid * n;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % n);
        int64_t r = idx / n;
        const int64_t o = r % n_mid;
        r /= n_mid;
        const int i = (int)(r % n_local), side = (int)(r / n_local);
        const int jj = hp.flip[side][i] ? n - 1 - j : j;
        out[idx] = rows[(((int64_t)hp.nbr[side][i] * 4 + hp.row[side][i]) * n_mid + o) * n + jj];
    }
}

// element-wise dtype conversion (the surface-data arithmetic runs in one dtype; restart files mix float32 and float64)
template <typename Tin, typename Tout>
__global__ void cast_kernel(const Tin *__restrict__ in, Tout *__restrict__ out, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (Tout)in[i];
}

// out[o][y][x'] = 0.5 * (left + right) along `axis` (0 = x: nx+1 points, 1 = y: ny+1 points); beyond the
// tile the neighbours come from lo / hi [o][along-edge index].  step > 1: only every step-th edge along `axis` (the lines an
// edge-weighted block average keeps, coarsen.py:221-273): n / step + 1 points, point j' is edge j' * step.
template <typename T>
__global__ void interp_to_outer_kernel(const T *__restrict__ in, const T *__restrict__ lo, const T *__restrict__ hi,
                                       T *__restrict__ out, int64_t n_outer, int ny, int nx, int axis, int step)
{
    const int nyo = (axis == 1) ? ny / step + 1 : ny, nxo = (axis == 0) ? nx / step + 1 : nx;
    const int64_t total = n_outer * nyo * nxo;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int x = (int)(idx % nxo);
        const int64_t r = idx / nxo;
        int y = (int)(r % nyo);
        const int64_t o = r / nyo;
        const T *f = in + o * (int64_t)ny * nx;
        if (axis == 0) x *= step; else y *= step;
        T a, b;
        if (axis == 0) {
            a = (x == 0) ? lo[o * ny + y] : f[(int64_t)y * nx + x - 1];
            b = (x == nx) ? hi[o * ny + y] : f[(int64_t)y * nx + x];
        } else {
            a = (y == 0) ? lo[o * nx + x] : f[(int64_t)(y - 1) * nx + x];
            b = (y == ny) ? hi[o * nx + x] : f[(int64_t)y * nx + x];
        }
        out[idx] = (T)0.5 * (a + b);
    }
}


// ---------------------------------------------------------------------------------------
// Small elementwise vocabulary for the mask arithmetic of the surface-data coarse-graining
// (external/vcm/vcm/cubedsphere/coarsen_restarts.py:1140-1470: isclose / where / & / products /
// xr.where / fillna on [tile, (level,) y, x] fields).  Masks are 0 / 1 in the fields' dtype.
// Operands b and c may be 2-D fields shared by `rep` consecutive outer slices of a.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void ew_kernel(int op, const T *__restrict__ a, const T *__restrict__ b, const T *__restrict__ c, T s,
                          int64_t n, int64_t inner, int64_t b_rep, int64_t c_rep, T *__restrict__ out)
{
    const T nan = (T)__builtin_nanf("");
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = i / inner, r = i - o * inner;
        const T x = a[i];
        const T y = b ? b[(o / b_rep) * inner + r] : (T)0;
        const T z = c ? c[(o / c_rep) * inner + r] : (T)0;
        T v;
        switch (op) {
            case FV3HIP_EW_MUL: v = x * y; break;
            case FV3HIP_EW_ISCLOSE: {  // np.isclose(a, b): |a - b| <= atol + rtol |b|, equal infinities close, NaN never
                const T d = x - y;
                v = ((x == y) || ((d < 0 ? -d : d) <= (T)1e-8 + (T)1e-5 * (y < 0 ? -y : y))) ? (T)1 : (T)0;
                if (x != x || y != y) v = (T)0;
                break;
            }
            case FV3HIP_EW_ISCLOSE_S: {
                const T d = x - s;
                v = ((x == s) || ((d < 0 ? -d : d) <= (T)1e-8 + (T)1e-5 * (s < 0 ? -s : s))) ? (T)1 : (T)0;
                if (x != x) v = (T)0;
                break;
            }
            case FV3HIP_EW_WHERE_NAN: v = (y != (T)0) ? x : nan; break;        // a.where(mask b)
            case FV3HIP_EW_SELECT: v = (z != (T)0) ? x : y; break;             // xr.where(mask c, a, b)
            case FV3HIP_EW_SELECT_S: v = (y != (T)0) ? s : x; break;           // xr.where(mask b, s, a)
            case FV3HIP_EW_GT_S: v = (x > s) ? (T)1 : (T)0; break;
            case FV3HIP_EW_LT_S: v = (x < s) ? (T)1 : (T)0; break;
            case FV3HIP_EW_FILLNA_S: v = (x != x) ? s : x; break;
            case FV3HIP_EW_AND: v = (x != (T)0 && y != (T)0) ? (T)1 : (T)0; break;
            case FV3HIP_EW_MIN_S: v = (x < s) ? x : s; break;                  // a.where(a < s, other=s)
            case FV3HIP_EW_BLEND: v = x * y + ((T)1 - x) * z; break;
            case FV3HIP_EW_WHERE_S: v = (y != (T)0) ? x : s; break;            // a.where(mask b, other=s)
            case FV3HIP_EW_ADD: v = x + y; break;
            case FV3HIP_EW_ADD_S: v = x + s; break;
            case FV3HIP_EW_SUB: v = x - y; break;
            case FV3HIP_EW_LOG_FLOOR_S: v = log(x < s ? s : x); break;         // tf.math.log(tf.maximum(a, s)); a NaN stays a NaN 
            case FV3HIP_EW_EXP: v = exp(x); break;
            case FV3HIP_EW_RELU_THRESHOLD_S: v = (x > s) ? x : (T)0; break;    // tf.keras.activations.relu(a, threshold=s)
            case FV3HIP_EW_BELOW_S: v = (x < s) ? x : (T)0; break;             // tf.cast(a < s, a.dtype) * a
            case FV3HIP_EW_DIV_S: v = x / s; break;
            case FV3HIP_EW_CLIP01: v = (x != x) ? x : (x < (T)0 ? (T)0 : (x > (T)1 ? (T)1 : x)); break;  // np.clip(a, 0, 1)
            case FV3HIP_EW_POW_BASE_S: v = (T)pow((double)s, (double)x); break;                            // scalar ** a
            case FV3HIP_EW_MINIMUM_S: v = (x != x) ? x : (x < s ? x : s); break;                           // np.minimum(a, scalar)
            case FV3HIP_EW_DIV: v = x / y; break;
            case FV3HIP_EW_WHERE_POS_S: v = (y > (T)0) ? x : s; break;                                     // xr.where(b > 0, a, scalar)
            case FV3HIP_EW_SIGN: v = (x != x) ? x : (x > (T)0 ? (T)1 : (x < (T)0 ? (T)-1 : (T)0)); break;  // np.sign: NaN stays, +-0 -> 0
            case FV3HIP_EW_ABS: v = (x < 0) ? -x : ((x == 0) ? (T)0 : x); break;                           // np.abs (-0 -> +0)
            case FV3HIP_EW_RSUB_S: v = s - x; break;
            case FV3HIP_EW_RDIV_S: v = s / x; break;
            case FV3HIP_EW_WHERE_GT_S: v = (x > s) ? x : s; break;                                         // a.where(a > scalar, scalar): a NaN becomes scalar
            case FV3HIP_EW_LE_S: v = (x <= s) ? (T)1 : (T)0; break;
            case FV3HIP_EW_SIN: v = sin(x); break;
            case FV3HIP_EW_COS: v = cos(x); break;
            case FV3HIP_EW_INCLOUD_TO_GRIDCELL: {  // vcm/calc/clouds.py:40-66 (a = cloud fraction, b = in-cloud condensate; CLIMIT1 = 1e-3, CLIMIT2 = 5e-2)
                const T rectified = (x > (T)5.0e-2) ? x : (T)5.0e-2;
                v = (x <= (T)1.0e-3) ? y : y * rectified;
                break;
            }
            case FV3HIP_EW_MUL_S: v = s * x; break;                            // scalar * a           // blend(weights a, pressure-level b, model-level c)
            default: v = x;
        }
        out[i] = v;
    }
}

}  // namespace
}  // namespace fv3hip

using namespace fv3hip;

extern "C" int fv3hip_weighted_block_average(const void *obj, int obj_dtype, const void *weights,
                                             int w_dtype, int64_t n_outer, int ny, int nx,
                                             int64_t w_repeat, int factor, void *out, void *stream)
{
    FV3HIP_REQUIRE(factor >= 1, "coarsening factor must be >= 1, got %d", factor);
    FV3HIP_REQUIRE(ny % factor == 0 && nx % factor == 0,
                   "horizontal extents (%d, %d) are not multiples of the coarsening factor %d", ny,
                   nx, factor);
    return wavg_entry(obj, obj_dtype, weights, w_dtype, n_outer, ny, nx, w_repeat, factor, factor,
                      factor, factor, out, stream);
}

extern "C" int fv3hip_mass_weighted_block_average(const void *const *fields, int n_fields, int dtype, const void *delp,
                                                  const void *area, int area_dtype, int64_t n_outer, int ny, int nx,
                                                  int64_t a_repeat, int factor, void *const *outs, void *stream)
{
    FV3HIP_REQUIRE(n_fields >= 0, "negative field count");
    if (n_fields == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64, got %d", dtype);
    FV3HIP_REQUIRE(area_dtype == FV3HIP_F32 || area_dtype == FV3HIP_F64, "area dtype must be F32 or F64, got %d", area_dtype);
    FV3HIP_REQUIRE(factor >= 1 && n_outer >= 0 && ny >= 0 && nx >= 0, "bad extents");
    FV3HIP_REQUIRE(ny % factor == 0 && nx % factor == 0, "horizontal extents (%d, %d) are not multiples of the coarsening factor %d",
                   ny, nx, factor);
    FV3HIP_REQUIRE(a_repeat >= 1 && n_outer % a_repeat == 0, "n_outer (%lld) is not a multiple of a_repeat (%lld)",
                   (long long)n_outer, (long long)a_repeat);
    if (n_outer == 0 || ny == 0 || nx == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(fields && outs && area, "null pointer");  // (delp may be null: the weights are `area` alone)
    for (int f = 0; f < n_fields; ++f) FV3HIP_REQUIRE(fields[f] && outs[f], "null field pointer");
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F32 && area_dtype == FV3HIP_F32)
        return dispatch_mass_wavg<float, float>(fields, n_fields, delp, area, n_outer, ny, nx, a_repeat, factor, outs, st);
    if (dtype == FV3HIP_F64 && area_dtype == FV3HIP_F64)
        return dispatch_mass_wavg<double, double>(fields, n_fields, delp, area, n_outer, ny, nx, a_repeat, factor, outs, st);
    if (dtype == FV3HIP_F64 && area_dtype == FV3HIP_F32)
        return dispatch_mass_wavg<double, float>(fields, n_fields, delp, area, n_outer, ny, nx, a_repeat, factor, outs, st);
    return dispatch_mass_wavg<float, double>(fields, n_fields, delp, area, n_outer, ny, nx, a_repeat, factor, outs, st);
}

extern "C" int fv3hip_edge_weighted_block_average(const void *obj, int obj_dtype,
                                                  const void *spacing, int w_dtype, int64_t n_outer,
                                                  int ny, int nx, int64_t w_repeat, int factor,
                                                  int edge, void *out, void *stream)
{
    FV3HIP_REQUIRE(factor >= 1, "coarsening factor must be >= 1, got %d", factor);
    FV3HIP_REQUIRE(edge == 0 || edge == 1, "edge must be 0 ('x') or 1 ('y'), got %d", edge);
    if (edge == 0) {
        FV3HIP_REQUIRE(nx % factor == 0, "x extent %d is not a multiple of the factor %d", nx, factor);
        return wavg_entry(obj, obj_dtype, spacing, w_dtype, n_outer, ny, nx, w_repeat, 1, factor,
                          factor, factor, out, stream);
    }
    FV3HIP_REQUIRE(ny % factor == 0, "y extent %d is not a multiple of the factor %d", ny, factor);
    return wavg_entry(obj, obj_dtype, spacing, w_dtype, n_outer, ny, nx, w_repeat, factor, 1, factor,
                      factor, out, stream);
}

extern "C" int fv3hip_weighted_window_average(const void *obj, int obj_dtype, const void *weights, int w_dtype, int64_t n_outer,
                                              int ny, int nx, int64_t w_repeat, int by, int bx, int sy, int sx, void *out,
                                              void *stream)
{
    FV3HIP_REQUIRE(by >= 1 && bx >= 1 && sy >= 1 && sx >= 1, "window and stride must be >= 1");
    FV3HIP_REQUIRE(ny >= by && nx >= bx, "the window (%d x %d) exceeds the field (%d x %d)", by, bx, ny, nx);
    return wavg_entry(obj, obj_dtype, weights, w_dtype, n_outer, ny, nx, w_repeat, by, bx, sy, sx, out, stream);
}

extern "C" int fv3hip_block_reduce(const void *in, int dtype, int64_t n_outer, int ny, int nx, int by,
                                   int bx, int sy, int sx, int op, int nan_policy, void *out,
                                   void *stream)
{
    FV3HIP_REQUIRE(by >= 1 && bx >= 1 && sy >= 1 && sx >= 1, "window and stride must be >= 1");
    FV3HIP_REQUIRE(op >= FV3HIP_OP_SUM && op <= FV3HIP_OP_MODE, "unknown reduction op %d", op);
    FV3HIP_REQUIRE(n_outer >= 0 && ny >= 0 && nx >= 0, "negative extent");
    if (n_outer == 0 || ny < by || nx < bx) return FV3HIP_OK;
    FV3HIP_REQUIRE(in && out, "null pointer");
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case FV3HIP_F32: return launch_block_reduce<float>(in, out, n_outer, ny, nx, by, bx, sy, sx, op, nan_policy, st);
        case FV3HIP_F64: return launch_block_reduce<double>(in, out, n_outer, ny, nx, by, bx, sy, sx, op, nan_policy, st);
        case FV3HIP_I32:
        case FV3HIP_I64:
            if (op == FV3HIP_OP_MEAN || op == FV3HIP_OP_MEDIAN)
                return fail(FV3HIP_EUNSUPPORTED, "mean/median of integer fields is not supported");
            return dtype == FV3HIP_I32
                       ? launch_block_reduce<int32_t>(in, out, n_outer, ny, nx, by, bx, sy, sx, op, nan_policy, st)
                       : launch_block_reduce<int64_t>(in, out, n_outer, ny, nx, by, bx, sy, sx, op, nan_policy, st);
        default: return fail(FV3HIP_EINVAL, "unknown dtype %d", dtype);
    }
}

extern "C" int fv3hip_block_upsample(const void *in, int elem_size, int64_t n_outer, int ny_in,
                                     int nx_in, int factor, void *out, void *stream)
{
    FV3HIP_REQUIRE(factor >= 1, "upsampling factor must be >= 1, got %d", factor);
    FV3HIP_REQUIRE(elem_size == 4 || elem_size == 8, "elem_size must be 4 or 8, got %d", elem_size);
    FV3HIP_REQUIRE(n_outer >= 0 && ny_in >= 0 && nx_in >= 0, "negative extent");
    if (n_outer == 0 || ny_in == 0 || nx_in == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(in && out, "null pointer");
    // odd size = staggered (interface) dimension: the last point is not repeated
    const int ny_out = (ny_in % 2 == 1) ? (ny_in - 1) * factor + 1 : ny_in * factor;
    const int nx_out = (nx_in % 2 == 1) ? (nx_in - 1) * factor + 1 : nx_in * factor;
    hipStream_t st = as_stream(stream);
    if (elem_size == 4)
        launch_upsample(static_cast<const uint32_t *>(in), static_cast<uint32_t *>(out), n_outer, ny_in, nx_in, ny_out, nx_out, factor, factor, st);
    else
        launch_upsample(static_cast<const uint64_t *>(in), static_cast<uint64_t *>(out), n_outer, ny_in, nx_in, ny_out, nx_out, factor, factor, st);
    return check_launch("upsample_kernel");
}

extern "C" int fv3hip_repeat(const void *in, int elem_size, int64_t n_outer, int ny_in, int nx_in, int fy, int fx, void *out,
                             void *stream)
{
    FV3HIP_REQUIRE(fy >= 1 && fx >= 1, "repeat counts must be >= 1, got %d, %d", fy, fx);
    FV3HIP_REQUIRE(elem_size == 4 || elem_size == 8, "elem_size must be 4 or 8, got %d", elem_size);
    FV3HIP_REQUIRE(n_outer >= 0 && ny_in >= 0 && nx_in >= 0, "negative extent");
    if (n_outer == 0 || ny_in == 0 || nx_in == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(in && out, "null pointer");
    const int ny_out = ny_in * fy, nx_out = nx_in * fx;
    hipStream_t st = as_stream(stream);
    if (elem_size == 4)
        launch_upsample(static_cast<const uint32_t *>(in), static_cast<uint32_t *>(out), n_outer, ny_in, nx_in, ny_out, nx_out, fy, fx, st);
    else
        launch_upsample(static_cast<const uint64_t *>(in), static_cast<uint64_t *>(out), n_outer, ny_in, nx_in, ny_out, nx_out, fy, fx, st);
    return check_launch("upsample_kernel");
}

extern "C" int fv3hip_cube_edge_rows(const void *in, int elem_size, int n_tiles, int64_t n_mid, int n, void *rows,
                                     void *stream)
{
    FV3HIP_REQUIRE(elem_size == 4 || elem_size == 8, "elem_size must be 4 or 8, got %d", elem_size);
    FV3HIP_REQUIRE(n_tiles >= 0 && n_mid >= 0 && n >= 0, "negative extent");
    if (n_tiles == 0 || n_mid == 0 || n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(in && rows, "null pointer");
    const int64_t total = (int64_t)n_tiles * 4 * n_mid * n;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t st = as_stream(stream);
    if (elem_size == 4)
        hipLaunchKernelGGL((cube_edge_rows_kernel<uint32_t>), dim3((unsigned)blocks), dim3(256), 0, st,
                           static_cast<const uint32_t *>(in), static_cast<uint32_t *>(rows), n_tiles, n_mid, n);
    else
        hipLaunchKernelGGL((cube_edge_rows_kernel<uint64_t>), dim3((unsigned)blocks), dim3(256), 0, st,
                           static_cast<const uint64_t *>(in), static_cast<uint64_t *>(rows), n_tiles, n_mid, n);
    return check_launch("cube_edge_rows_kernel");
}

extern "C" int fv3hip_halo_pick(const void *rows, int elem_size, int n_local, int64_t n_mid, int n, const int *nbr, const int *row,
                                const int *flip, void *out, void *stream)
{
    FV3HIP_REQUIRE(elem_size == 4 || elem_size == 8, "elem_size must be 4 or 8, got %d", elem_size);
    FV3HIP_REQUIRE(n_local >= 0 && n_local <= 6 && n_mid >= 0 && n >= 0, "bad extents");
    if (n_local == 0 || n_mid == 0 || n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(rows && out && nbr && row && flip, "null pointer");
    HaloPick hp;
    memset(&hp, 0, sizeof(hp));
    for (int side = 0; side < 2; ++side)
        for (int i = 0; i < n_local; ++i) {
            const int k = side * n_local + i;
            FV3HIP_REQUIRE(nbr[k] >= 0 && nbr[k] < 6 && row[k] >= 0 && row[k] < 4, "bad neighbour entry %d", k);
            hp.nbr[side][i] = nbr[k];
            hp.row[side][i] = row[k];
            hp.flip[side][i] = flip[k] ? 1 : 0;
        }
    const int64_t total = 2 * (int64_t)n_local * n_mid * n;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t st = as_stream(stream);
    if (elem_size == 4)
        hipLaunchKernelGGL((halo_pick_kernel<uint32_t>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const uint32_t *>(rows),
                           static_cast<uint32_t *>(out), hp, n_local, n_mid, n);
    else
        hipLaunchKernelGGL((halo_pick_kernel<uint64_t>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const uint64_t *>(rows),
                           static_cast<uint64_t *>(out), hp, n_local, n_mid, n);
    return check_launch("halo_pick_kernel");
}

namespace {
constexpr int kCastMany = 48;
struct CastTable {
    const void *in[kCastMany];
    void *out[kCastMany];
    int64_t n[kCastMany];
    signed char in_dtype[kCastMany];
};

template <typename Tout>
__global__ void cast_many_kernel(const CastTable t)
{
    const int k = blockIdx.y;
    const int64_t n = t.n[k];
    Tout *out = static_cast<Tout *>(t.out[k]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        switch (t.in_dtype[k]) {
            case FV3HIP_F32: out[i] = (Tout) static_cast<const float *>(t.in[k])[i]; break;
            case FV3HIP_F64: out[i] = (Tout) static_cast<const double *>(t.in[k])[i]; break;
            case FV3HIP_I32: out[i] = (Tout) static_cast<const int32_t *>(t.in[k])[i]; break;
            default: out[i] = (Tout) static_cast<const int64_t *>(t.in[k])[i]; break;
        }
    }
}
}  // namespace

extern "C" int fv3hip_cast_many(const void *const *in, const int *in_dtype, void *const *out, int out_dtype, const int64_t *n,
                                int count, void *stream)
{
    FV3HIP_REQUIRE(out_dtype == FV3HIP_F32 || out_dtype == FV3HIP_F64, "out_dtype must be F32 or F64, got %d", out_dtype);
    FV3HIP_REQUIRE(count >= 0, "negative count");
    if (count == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(in && in_dtype && out && n, "null pointer");
    hipStream_t st = as_stream(stream);
    for (int k0 = 0; k0 < count; k0 += kCastMany) {
        const int m = (count - k0 < kCastMany) ? count - k0 : kCastMany;
        CastTable t;
        memset(&t, 0, sizeof(t));
        int64_t nmax = 0;
        for (int k = 0; k < m; ++k) {
            FV3HIP_REQUIRE(n[k0 + k] >= 0 && (n[k0 + k] == 0 || (in[k0 + k] && out[k0 + k])), "bad entry %d", k0 + k);
            FV3HIP_REQUIRE(in_dtype[k0 + k] >= FV3HIP_F32 && in_dtype[k0 + k] <= FV3HIP_I64, "unknown in_dtype %d", in_dtype[k0 + k]);
            t.in[k] = in[k0 + k];
            t.out[k] = out[k0 + k];
            t.n[k] = n[k0 + k];
            t.in_dtype[k] = (signed char)in_dtype[k0 + k];
            if (n[k0 + k] > nmax) nmax = n[k0 + k];
        }
        if (nmax == 0) continue;
        int64_t bx = ceil_div(nmax, 256 * 4);
        if (bx > 4096) bx = 4096;
        if (bx < 1) bx = 1;
        dim3 grid((unsigned)bx, (unsigned)m);
        if (out_dtype == FV3HIP_F32)
            hipLaunchKernelGGL((cast_many_kernel<float>), grid, dim3(256), 0, st, t);
        else
            hipLaunchKernelGGL((cast_many_kernel<double>), grid, dim3(256), 0, st, t);
        const int rc = check_launch("cast_many_kernel");
        if (rc) return rc;
    }
    return FV3HIP_OK;
}

namespace {
template <typename Tin>
int launch_cast(const void *in, int out_dtype, void *out, int64_t n, hipStream_t st)
{
    int64_t blocks = ceil_div(n, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (out_dtype == FV3HIP_F32)
        hipLaunchKernelGGL((cast_kernel<Tin, float>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const Tin *>(in), static_cast<float *>(out), n);
    else
        hipLaunchKernelGGL((cast_kernel<Tin, double>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const Tin *>(in), static_cast<double *>(out), n);
    return check_launch("cast_kernel");
}
}  // namespace

extern "C" int fv3hip_cast(const void *in, int in_dtype, void *out, int out_dtype, int64_t n, void *stream)
{
    FV3HIP_REQUIRE(out_dtype == FV3HIP_F32 || out_dtype == FV3HIP_F64, "out_dtype must be F32 or F64, got %d", out_dtype);
    FV3HIP_REQUIRE(n >= 0, "negative length");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(in && out, "null pointer");
    hipStream_t st = as_stream(stream);
    switch (in_dtype) {
        case FV3HIP_F32: return launch_cast<float>(in, out_dtype, out, n, st);
        case FV3HIP_F64: return launch_cast<double>(in, out_dtype, out, n, st);
        case FV3HIP_I32: return launch_cast<int32_t>(in, out_dtype, out, n, st);
        case FV3HIP_I64: return launch_cast<int64_t>(in, out_dtype, out, n, st);
        default: return fail(FV3HIP_EINVAL, "unknown in_dtype %d", in_dtype);
    }
}

extern "C" int fv3hip_interp_center_to_outer_lines(const void *in, int dtype, int64_t n_outer, int ny, int nx, int axis, int step,
                                                   const void *lo, const void *hi, void *out, void *stream)
{
    FV3HIP_REQUIRE(axis == 0 || axis == 1, "axis must be 0 ('x') or 1 ('y'), got %d", axis);
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_outer >= 0 && ny >= 0 && nx >= 0, "negative extent");
    FV3HIP_REQUIRE(step >= 1 && (axis == 0 ? nx : ny) % step == 0, "the extent along the axis must be a multiple of step (%d)", step);
    if (n_outer == 0 || ny == 0 || nx == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(in && lo && hi && out, "null pointer");
    const int64_t total = n_outer * ((axis == 1) ? ny / step + 1 : ny) * ((axis == 0) ? nx / step + 1 : nx);
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F32)
        hipLaunchKernelGGL((interp_to_outer_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st,
                           static_cast<const float *>(in), static_cast<const float *>(lo), static_cast<const float *>(hi),
                           static_cast<float *>(out), n_outer, ny, nx, axis, step);
    else
        hipLaunchKernelGGL((interp_to_outer_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st,
                           static_cast<const double *>(in), static_cast<const double *>(lo), static_cast<const double *>(hi),
                           static_cast<double *>(out), n_outer, ny, nx, axis, step);
    return check_launch("interp_to_outer_kernel");
}

extern "C" int fv3hip_interp_center_to_outer(const void *in, int dtype, int64_t n_outer, int ny, int nx, int axis,
                                             const void *lo, const void *hi, void *out, void *stream)
{
    return fv3hip_interp_center_to_outer_lines(in, dtype, n_outer, ny, nx, axis, 1, lo, hi, out, stream);
}

extern "C" int fv3hip_ew(int op, const void *a, const void *b, const void *c, double scalar, int dtype, int64_t n,
                         int64_t inner, int64_t b_rep, int64_t c_rep, void *out, void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(op >= FV3HIP_EW_MUL && op <= FV3HIP_EW_COS, "unknown elementwise op %d", op);
    FV3HIP_REQUIRE(n >= 0 && inner >= 1 && b_rep >= 1 && c_rep >= 1, "bad extents");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(a && out, "null pointer");
    const bool needs_b = op == FV3HIP_EW_MUL || op == FV3HIP_EW_ISCLOSE || op == FV3HIP_EW_WHERE_NAN ||
                         op == FV3HIP_EW_SELECT || op == FV3HIP_EW_SELECT_S || op == FV3HIP_EW_AND || op == FV3HIP_EW_BLEND ||
                         op == FV3HIP_EW_WHERE_S || op == FV3HIP_EW_ADD || op == FV3HIP_EW_SUB;
    FV3HIP_REQUIRE(!needs_b || b, "this op needs operand b");
    FV3HIP_REQUIRE((op != FV3HIP_EW_SELECT && op != FV3HIP_EW_BLEND) || c, "this op needs operand c");
    FV3HIP_REQUIRE(n % inner == 0, "n must be a multiple of inner");
    int64_t blocks = ceil_div(n, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((ew_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st, op, static_cast<const double *>(a),
                           static_cast<const double *>(b), static_cast<const double *>(c), scalar, n, inner, b_rep, c_rep,
                           static_cast<double *>(out));
    else
        hipLaunchKernelGGL((ew_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, op, static_cast<const float *>(a),
                           static_cast<const float *>(b), static_cast<const float *>(c), (float)scalar, n, inner, b_rep, c_rep,
                           static_cast<float *>(out));
    return check_launch("ew_kernel");
}
