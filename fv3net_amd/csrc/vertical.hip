// Vertical kernels for gfx950: interface pressures, weight masking, and the PPM remap (mappm).
//
// Reference behaviour restated (paths relative to the reference checkout):
//   external/vcm/vcm/calc/thermo/vertically_dependent.py:41-66   pressure_at_interface
//   external/vcm/vcm/cubedsphere/regridz.py:200-220               _mask_weights
//   external/mappm/mappm/mappm.f90:10-126, 614-851, 854-931       mappm, ppm_profile, ppm_limiters
//
// This file is compiled with -ffp-contract=off: the remap is single precision with the exact
// association order of the Fortran, so that its results are bit-identical to the reference
// built without FMA contraction (and to oracle/mappm_oracle.c).
#include "common.h"
#include "remap.h"

namespace fv3hip {
namespace {

// ---------------------------------------------------------------------------------------
// pressure_at_interface: sequential cumulative sum down each column, in the array's dtype.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void pressure_at_interface_kernel(const T *__restrict__ delp, T *__restrict__ out,
                                             int64_t n_batch, int nz, int64_t n_inner, T toa)
{
    const int64_t ncol = n_batch * n_inner;
    for (int64_t col = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; col < ncol;
         col += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = col / n_inner, c = col % n_inner;
        const T *pd = delp + b * nz * n_inner + c;
        T *po = out + b * (nz + 1) * n_inner + c;
        T p = toa;
        po[0] = p;
        for (int k = 0; k < nz; ++k) {
            p = p + pd[(int64_t)k * n_inner];
            po[(int64_t)(k + 1) * n_inner] = p;
        }
    }
}

// ---------------------------------------------------------------------------------------
// _mask_weights (extrapolate=False): keep the weight where the bottom interface of coarse
// layer k lies above the fine-grid surface pressure.
// ---------------------------------------------------------------------------------------
template <typename Tw, typename Tp>
__global__ void mask_weights_kernel(const Tw *__restrict__ w, const Tp *__restrict__ pc,
                                    const Tp *__restrict__ pf, Tw *__restrict__ out,
                                    int64_t n_batch, int nz, int64_t n_inner, int64_t w_repeat,
                                    int cmp_levels, int cmp_offset)
{
    const int64_t total = n_batch * nz * n_inner;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = idx % n_inner;
        const int64_t t = idx / n_inner;
        const int k = (int)(t % nz);
        const int64_t b = t / nz;
        const Tp level = pc[(b * cmp_levels + (k + cmp_offset)) * n_inner + c];
        const Tp ps = pf[(b * (nz + 1) + nz) * n_inner + c];
        out[idx] = (level < ps) ? w[(b / w_repeat) * n_inner + c] : (Tw)0;
    }
}

// The same on a (column chunk, batch x level) grid: no 64-bit division per element, 4 columns per thread (n_inner % 4 == 0).
template <typename Tw, typename Tp>
__global__ __launch_bounds__(256) void mask_weights_rows_kernel(const Tw *__restrict__ w, const Tp *__restrict__ pc, const Tp *__restrict__ pf,
                                                                Tw *__restrict__ out, int nz, int64_t n_inner, int64_t w_repeat, int cmp_levels,
                                                                int cmp_offset)
{
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= n_inner) return;
    const int64_t b = blockIdx.y / nz;
    const int k = (int)(blockIdx.y - b * nz);
    const Tp *lv = pc + (b * cmp_levels + (k + cmp_offset)) * n_inner + c;
    const Tp *ps = pf + (b * (nz + 1) + nz) * n_inner + c;
    const Tw *wr = w + (b / w_repeat) * n_inner + c;
    Tw *o = out + (b * nz + k) * n_inner + c;
    Tp l4[4], p4[4];
    Tw w4[4], r4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        l4[i] = lv[i];
        p4[i] = ps[i];
        w4[i] = wr[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = (l4[i] < p4[i]) ? w4[i] : (Tw)0;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = r4[i];
}

// ... with the compared pressures on a grid coarser by f in y and x (never upsampled): column (y, x) compares (y / f, x / f)
template <typename Tw, typename Tp>
__global__ __launch_bounds__(256) void mask_weights_coarse_kernel(const Tw *__restrict__ w, const Tp *__restrict__ pc, const Tp *__restrict__ pf,
                                                                  Tw *__restrict__ out, int nz, int64_t n_inner, int nx, int f, int nxc,
                                                                  int64_t plane2, int64_t w_repeat, int cmp_levels, int cmp_offset)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_inner) return;
    const int64_t b = blockIdx.y / nz;
    const int k = (int)(blockIdx.y - b * nz);
    const unsigned int y = (unsigned int)c / (unsigned int)nx, x = (unsigned int)c - y * (unsigned int)nx;
    const Tp level = pc[(b * cmp_levels + (k + cmp_offset)) * plane2 + (int64_t)(y / f) * nxc + x / f];
    const Tp ps = pf[(b * (nz + 1) + nz) * n_inner + c];
    out[(b * nz + k) * n_inner + c] = (level < ps) ? w[(b / w_repeat) * n_inner + c] : (Tw)0;
}

// ... four columns of one row per thread (nx % 4 == 0, f >= 4: the four share their row and at most two coarse columns), 16-byte
// accesses, one division per four elements
template <typename Tw, typename Tp>
__global__ __launch_bounds__(256) void mask_weights_coarse4_kernel(const Tw *__restrict__ w, const Tp *__restrict__ pc, const Tp *__restrict__ pf,
                                                                   Tw *__restrict__ out, int nz, int64_t n_inner, int nx, int f, int nxc,
                                                                   int64_t plane2, int64_t w_repeat, int cmp_levels, int cmp_offset)
{
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= n_inner) return;
    const int64_t b = blockIdx.y / nz;
    const int k = (int)(blockIdx.y - b * nz);
    const unsigned int y = (unsigned int)c / (unsigned int)nx, x = (unsigned int)c - y * (unsigned int)nx;
    const unsigned int xc = x / (unsigned int)f, rem = x - xc * (unsigned int)f;
    const Tp *lv = pc + (b * cmp_levels + (k + cmp_offset)) * plane2 + (int64_t)(y / f) * nxc + xc;
    const Tp l0 = lv[0], l1 = (rem + 3 >= (unsigned int)f) ? lv[1] : l0;   // (x + 3 stays inside the row: its coarse column exists)
    const Tp *ps = pf + (b * (nz + 1) + nz) * n_inner + c;
    const Tw *wr = w + (b / w_repeat) * n_inner + c;
    Tw *o = out + (b * nz + k) * n_inner + c;
    Tp p4[4];
    Tw w4[4], r4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p4[i] = ps[i];
        w4[i] = wr[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = (((rem + i >= (unsigned int)f) ? l1 : l0) < p4[i]) ? w4[i] : (Tw)0;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = r4[i];
}

// pressure_at_midpoint_log: delp / diff(log(p_interface)), sequential down the column
template <typename T>
__global__ void pressure_at_midpoint_log_kernel(const T *__restrict__ delp, T *__restrict__ out,
                                                int64_t n_batch, int nz, int64_t n_inner, T toa)
{
    const int64_t ncol = n_batch * n_inner;
    for (int64_t col = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; col < ncol;
         col += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = col / n_inner, c = col % n_inner;
        const T *pd = delp + b * nz * n_inner + c;
        T *po = out + b * nz * n_inner + c;
        T p = toa;
        T lp_prev = log(p);
        for (int k = 0; k < nz; ++k) {
            const T d = pd[(int64_t)k * n_inner];
            p = p + d;
            const T lp = log(p);
            po[(int64_t)k * n_inner] = d / (lp - lp_prev);
            lp_prev = lp;
        }
    }
}

// ---------------------------------------------------------------------------------------
// mappm, version 1: one thread per column, the Fortran control flow verbatim, the
// reconstruction (AL, AR, A6) and the slopes DC / H2 kept in a global workspace laid out
// [array][level][column-in-chunk] so that every access of a wave is one coalesced row.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float f_sign(float a, float b) { return copysignf(fabsf(a), b); }
__device__ __forceinline__ float f_min2(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float f_max2(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float f_min3(float a, float b, float c) { return f_min2(f_min2(a, b), c); }
__device__ __forceinline__ float f_max3(float a, float b, float c) { return f_max2(f_max2(a, b), c); }

// mappm.f90:854-931 for one (column, level)
__device__ __forceinline__ void ppm_limiters1(float dm, float a1, float &a2, float &a3, float &a4,
                                              int lmt)
{
    const float r12 = 1.f / 12.f;
    if (lmt == 3) return;
    if (lmt == 0) {
        if (dm == 0.f) {
            a2 = a1;
            a3 = a1;
            a4 = 0.f;
        } else {
            const float da1 = a3 - a2;
            const float da2 = da1 * da1;
            const float a6da = a4 * da1;
            if (a6da < -da2) {
                a4 = 3.f * (a2 - a1);
                a3 = a2 - a4;
            } else if (a6da > da2) {
                a4 = 3.f * (a3 - a1);
                a2 = a3 - a4;
            }
        }
    } else if (lmt == 1) {
        const float qmp = 2.f * dm;
        a2 = a1 - f_sign(f_min2(fabsf(qmp), fabsf(a2 - a1)), qmp);
        a3 = a1 + f_sign(f_min2(fabsf(qmp), fabsf(a3 - a1)), qmp);
        a4 = 3.f * (2.f * a1 - (a2 + a3));
    } else if (lmt == 2) {
        if (fabsf(a3 - a2) < -a4) {
            const float d = a3 - a2;
            const float fmin = a1 + 0.25f * (d * d) / a4 + a4 * r12;
            if (fmin < 0.f) {
                if (a1 < a3 && a1 < a2) {
                    a3 = a1;
                    a2 = a1;
                    a4 = 0.f;
                } else if (a3 > a2) {
                    a4 = 3.f * (a2 - a1);
                    a3 = a2 - a4;
                } else {
                    a4 = 3.f * (a3 - a1);
                    a2 = a3 - a4;
                }
            }
        }
    }
}

// element (column, level k) of an array with `nlev` levels lives at base + (k-1)*ks
struct ColumnAddr {
    int64_t ks, o_pe1, o_q1, o_pe2, o_q2;
    int64_t ks2;  // level stride of pe2 (= ks unless the target interfaces live on a coarser grid)
};

__device__ __forceinline__ ColumnAddr column_addr(int64_t col, int64_t n_inner, int km, int kn, int layout)
{
    ColumnAddr a;
    if (layout == FV3HIP_LAYOUT_COL_LEVEL) {
        a.ks = a.ks2 = 1;
        a.o_pe1 = col * (km + 1);
        a.o_q1 = col * km;
        a.o_pe2 = col * (kn + 1);
        a.o_q2 = col * kn;
    } else {
        a.ks = a.ks2 = n_inner;
        const int64_t b = col / n_inner, c = col % n_inner;
        a.o_pe1 = b * (km + 1) * n_inner + c;
        a.o_q1 = b * km * n_inner + c;
        a.o_pe2 = b * (kn + 1) * n_inner + c;
        a.o_q2 = b * kn * n_inner + c;
    }
    return a;
}

// LEVEL_COL layout with the target interfaces on a grid coarser by f in y and x (rows of nx columns)
__device__ __forceinline__ ColumnAddr column_addr_coarse_target(int64_t col, int64_t n_inner, int km, int kn, int f, int nx, int nxc,
                                                                int64_t plane2)
{
    ColumnAddr a = column_addr(col, n_inner, km, kn, FV3HIP_LAYOUT_LEVEL_COL);
    const int64_t b = col / n_inner, c = col % n_inner, y = c / nx, x = c - y * nx;
    a.ks2 = plane2;
    a.o_pe2 = b * (kn + 1) * plane2 + (y / f) * nxc + x / f;
    return a;
}

// cs_limiters (mappm.f90:532-611) for one level; mode = the routine's `iv` argument
__device__ __forceinline__ void cs_limiters1(bool extm, float a1, float &a2, float &a3, float &a4, int mode)
{
    const float r12 = 1.f / 12.f;
    if (mode == 0) {  // positive definite constraint
        if (a1 <= 0.f) {
            a2 = a1;
            a3 = a1;
            a4 = 0.f;
        } else if (fabsf(a3 - a2) < -a4) {
            if ((a1 + 0.25f * ((a3 - a2) * (a3 - a2)) / a4 + a4 * r12) < 0.f) {  // the local minimum is negative
                if (a1 < a3 && a1 < a2) {
                    a3 = a1;
                    a2 = a1;
                    a4 = 0.f;
                } else if (a3 > a2) {
                    a4 = 3.f * (a2 - a1);
                    a3 = a2 - a4;
                } else {
                    a4 = 3.f * (a3 - a1);
                    a2 = a3 - a4;
                }
            }
        }
        return;
    }
    if (mode == 1 ? ((a1 - a2) * (a1 - a3) >= 0.f) : extm) {
        a2 = a1;
        a3 = a1;
        a4 = 0.f;
        return;
    }
    const float da1 = a3 - a2, da2 = da1 * da1, a6da = a4 * da1;
    if (a6da < -da2) {
        a4 = 3.f * (a2 - a1);
        a3 = a2 - a4;
    } else if (a6da > da2) {
        a4 = 3.f * (a3 - a1);
        a2 = a3 - a4;
    }
}

// cs_profile (mappm.f90:132-529, kord > 7, iv != -2) for one column, the Fortran's operations in its order.  The edge
// values q(1:km) live in the QE plane of the workspace (q(km+1) in a register), the tridiagonal's gam in the GAM plane; the
// differences of the cell means the constraints use and the extremum flags ext5 / ext6 are recomputed where they are
// read (the flags from the edge values as they were BEFORE the subgrid constraints, which is what QE keeps).
template <typename FQ, typename FDP>
__device__ __forceinline__ void cs_profile_column(FQ Q, FDP DP, float *AL, float *AR, float *A6, float *GAM, float *QE,
                                                  int64_t ws_cols, int km, int iv, int kord)
{
#define W_(arr, k) arr[(int64_t)((k)-1) * ws_cols]
    float d4 = 0.f, qe_last;
    {
        const float grat = DP(2) / DP(1);
        const float bet = grat * (grat + 0.5f);
        W_(QE, 1) = ((grat + grat) * (grat + 1.f) * Q(1) + Q(2)) / bet;
        W_(GAM, 1) = (1.f + grat * (grat + 1.5f)) / bet;
    }
    for (int k = 2; k <= km; ++k) {
        d4 = DP(k - 1) / DP(k);
        const float bet = 2.f + d4 + d4 - W_(GAM, k - 1);
        W_(QE, k) = (3.f * (Q(k - 1) + d4 * Q(k)) - W_(QE, k - 1)) / bet;
        W_(GAM, k) = d4 / bet;
    }
    {
        const float a_bot = 1.f + d4 * (d4 + 1.5f);
        qe_last = (2.f * d4 * (d4 + 1.f) * Q(km) + Q(km - 1) - a_bot * W_(QE, km)) / (d4 * (d4 + 0.5f) - a_bot * W_(GAM, km));
    }
    {
        float below = qe_last;
        for (int k = km; k >= 1; --k) {
            below = W_(QE, k) - W_(GAM, k) * below;
            W_(QE, k) = below;
        }
    }
    auto QEK = [&](int k) { return k > km ? qe_last : W_(QE, k); };
    if (kord > 16) {  // perfectly linear scheme
        for (int k = 1; k <= km; ++k) {
            const float al = QEK(k), ar = QEK(k + 1);
            W_(AL, k) = al;
            W_(AR, k) = ar;
            W_(A6, k) = 3.f * (2.f * Q(k) - (al + ar));
        }
        return;
    }
    auto G = [&](int k) { return Q(k) - Q(k - 1); };  // gam(i,k) of the constraints, k = 2 .. km
    auto bound = [&](float q, int k) {  // min(q, max(a1(k-1), a1(k))) then max(.., min(a1(k-1), a1(k)))
        q = f_min2(q, f_max2(Q(k - 1), Q(k)));
        return f_max2(q, f_min2(Q(k - 1), Q(k)));
    };
    W_(QE, 2) = bound(W_(QE, 2), 2);
    for (int k = 3; k <= km - 1; ++k) {
        float q = W_(QE, k);
        const float gm = G(k - 1);
        if (gm * G(k + 1) > 0.f) {
            q = bound(q, k);
        } else if (gm > 0.f) {  // a local maximum
            q = f_max2(q, f_min2(Q(k - 1), Q(k)));
        } else {                // a local minimum
            q = f_min2(q, f_max2(Q(k - 1), Q(k)));
            if (iv == 0) q = f_max2(0.f, q);
        }
        W_(QE, k) = q;
    }
    W_(QE, km) = bound(W_(QE, km), km);
    for (int k = 1; k <= km; ++k) {
        W_(AL, k) = QEK(k);
        W_(AR, k) = QEK(k + 1);
    }
    const bool extm_top = (QEK(1) - Q(1)) * (QEK(2) - Q(1)) > 0.f, extm_bot = (QEK(km) - Q(km)) * (qe_last - Q(km)) > 0.f;
    auto EXTM = [&](int k) { return k == 1 ? extm_top : k == km ? extm_bot : (G(k) * G(k + 1) < 0.f); };
    auto EXT5 = [&](int k) {
        const float l = QEK(k), r = QEK(k + 1);
        return fabsf(2.f * Q(k) - (l + r)) > fabsf(l - r);
    };
    auto EXT6 = [&](int k) {
        const float l = QEK(k), r = QEK(k + 1);
        return fabsf(3.f * (2.f * Q(k) - (l + r))) > fabsf(l - r);
    };
    auto edges_a6 = [](float a1, float al, float ar) { return 3.f * (2.f * a1 - (al + ar)); };
    auto huynh = [&](int k, float a1, float &al, float &ar) {
        const float pmp_1 = a1 - 2.f * G(k + 1), lac_1 = pmp_1 + 1.5f * G(k + 2);
        al = f_min2(f_max2(al, f_min3(a1, pmp_1, lac_1)), f_max3(a1, pmp_1, lac_1));
        const float pmp_2 = a1 + 2.f * G(k), lac_2 = pmp_2 - 1.5f * G(k - 1);
        ar = f_min2(f_max2(ar, f_min3(a1, pmp_2, lac_2)), f_max3(a1, pmp_2, lac_2));
    };
    auto store = [&](int k, float al, float ar, float a6) {
        W_(AL, k) = al;
        W_(AR, k) = ar;
        W_(A6, k) = a6;
    };
    {   // top layer
        const float a1 = Q(1);
        float al = W_(AL, 1), ar = W_(AR, 1), a6 = 0.f;
        if (iv == 0) {
            al = f_max2(0.f, al);
        } else if (iv == -1) {
            if (al * a1 <= 0.f) al = 0.f;
        } else if (iv == 2) {
            al = a1;
            ar = a1;
        }
        if (iv != 2) {
            a6 = edges_a6(a1, al, ar);
            cs_limiters1(extm_top, a1, al, ar, a6, 1);
        }
        store(1, al, ar, a6);
    }
    {   // k = 2
        const float a1 = Q(2);
        float al = W_(AL, 2), ar = W_(AR, 2), a6 = edges_a6(a1, al, ar);
        cs_limiters1(EXTM(2), a1, al, ar, a6, 2);
        store(2, al, ar, a6);
    }
    for (int k = 3; k <= km - 2; ++k) {
        const float a1 = Q(k);
        float al = W_(AL, k), ar = W_(AR, k), a6;
        if (kord < 9) {
            huynh(k, a1, al, ar);
            a6 = edges_a6(a1, al, ar);
        } else if (kord == 9 || kord == 12) {
            if (kord == 9 ? (EXTM(k) && (EXTM(k - 1) || EXTM(k + 1))) : EXTM(k)) {  // a 2-delta-z wave
                al = a1;
                ar = a1;
                a6 = 0.f;
            } else {
                a6 = 6.f * a1 - 3.f * (al + ar);
                if (fabsf(a6) > fabsf(al - ar)) {  // not monotonic inside the smooth region
                    huynh(k, a1, al, ar);
                    a6 = 6.f * a1 - 3.f * (al + ar);
                }
            }
        } else if (kord == 10 || kord == 16) {
            if (EXT5(k)) {
                if (EXT5(k - 1) || EXT5(k + 1)) {
                    al = a1;
                    ar = a1;
                } else if (EXT6(k - 1) || EXT6(k + 1)) {
                    huynh(k, a1, al, ar);
                }
            } else if (kord == 10 && EXT6(k)) {
                if (EXT5(k - 1) || EXT5(k + 1)) huynh(k, a1, al, ar);
            }
            a6 = edges_a6(a1, al, ar);
        } else if (kord == 13) {
            if (EXT6(k) && EXT6(k - 1) && EXT6(k + 1)) {
                al = a1;
                ar = a1;
            }
            a6 = edges_a6(a1, al, ar);
        } else if (kord == 14) {
            a6 = edges_a6(a1, al, ar);
        } else if (kord == 15) {
            if (EXT5(k)) {
                if (EXT5(k - 1) || EXT5(k + 1)) {
                    al = a1;
                    ar = a1;
                }
            } else if (EXT6(k)) {
                huynh(k, a1, al, ar);
            }
            a6 = edges_a6(a1, al, ar);
        } else {  // kord == 11
            if (EXT5(k) && (EXT5(k - 1) || EXT5(k + 1))) {  // a noisy region
                al = a1;
                ar = a1;
                a6 = 0.f;
            } else {
                a6 = edges_a6(a1, al, ar);
            }
        }
        if (iv == 0) cs_limiters1(EXTM(k), a1, al, ar, a6, 0);
        store(k, al, ar, a6);
    }
    {   // bottom two layers
        float ar_km = W_(AR, km);
        if (iv == 0) {
            ar_km = f_max2(0.f, ar_km);
        } else if (iv == -1) {
            if (ar_km * Q(km) <= 0.f) ar_km = 0.f;
        }
        {
            const float a1 = Q(km - 1);
            float al = W_(AL, km - 1), ar = W_(AR, km - 1), a6 = edges_a6(a1, al, ar);
            cs_limiters1(EXTM(km - 1), a1, al, ar, a6, 2);
            store(km - 1, al, ar, a6);
        }
        const float a1 = Q(km);
        float al = W_(AL, km), a6 = edges_a6(a1, al, ar_km);
        cs_limiters1(extm_bot, a1, al, ar_km, a6, 1);
        store(km, al, ar_km, a6);
    }
#undef W_
}

// One column, the Fortran control flow verbatim (all iv and kord, any input whatsoever).
template <typename Tin>
__device__ __noinline__ void mappm_column_exact(const Tin *__restrict__ pe1_, const Tin *__restrict__ q1_,
                                                const Tin *__restrict__ pe2_, float *__restrict__ q2_,
                                                const ColumnAddr addr, int km, int kn, int iv, int kord,
                                                float *__restrict__ ws, int64_t lc, int64_t ws_cols)
{
    const int64_t ks = addr.ks, o_pe1 = addr.o_pe1, o_q1 = addr.o_q1, o_pe2 = addr.o_pe2, o_q2 = addr.o_q2;
    auto PE1 = [&](int k) { return (float)pe1_[o_pe1 + (int64_t)(k - 1) * ks]; };
    auto Q = [&](int k) { return (float)q1_[o_q1 + (int64_t)(k - 1) * ks]; };
    auto PE2 = [&](int k) { return (float)pe2_[o_pe2 + (int64_t)(k - 1) * addr.ks2]; };
    auto DP = [&](int k) { return PE1(k + 1) - PE1(k); };          // dp1(i,k)
    auto DELQ = [&](int k) { return Q(k + 1) - Q(k); };            // delq(i,k)
    auto D4 = [&](int k) { return DP(k - 1) + DP(k); };            // d4(i,k)

    const int64_t plane = (int64_t)km * ws_cols;
    float *AL = ws + 0 * plane + lc, *AR = ws + 1 * plane + lc, *A6 = ws + 2 * plane + lc,
          *DC = ws + 3 * plane + lc, *H2 = ws + 4 * plane + lc;
#define W_(arr, k) arr[(int64_t)((k)-1) * ws_cols]

    const int km1 = km - 1;
    if (kord > 7) {
        cs_profile_column(Q, DP, AL, AR, A6, DC, H2, ws_cols, km, iv, kord);
    } else {
    // ---- ppm_profile (mappm.f90:651-683) ----
    for (int k = 2; k <= km1; ++k) {
        const float dpk = DP(k), d4k = D4(k), d4k1 = D4(k + 1);
        const float c1 = (DP(k - 1) + 0.5f * dpk) / d4k1;
        const float c2 = (DP(k + 1) + 0.5f * dpk) / d4k;
        const float df2 = dpk * (c1 * DELQ(k) + c2 * DELQ(k - 1)) / (d4k + DP(k + 1));
        const float qm1 = Q(k - 1), q0 = Q(k), qp1 = Q(k + 1);
        W_(DC, k) = f_sign(f_min3(fabsf(df2), f_max3(qm1, q0, qp1) - q0, q0 - f_min3(qm1, q0, qp1)), df2);
    }
    for (int k = 3; k <= km1; ++k) {
        const float d4k = D4(k);
        const float c1 = DELQ(k - 1) * DP(k - 1) / d4k;
        const float a1 = D4(k - 1) / (d4k + DP(k - 1));
        const float a2 = D4(k + 1) / (d4k + DP(k));
        W_(AL, k) = Q(k - 1) + c1 +
                    2.f / (D4(k - 1) + D4(k + 1)) *
                        (DP(k) * (c1 * (a1 - a2) + a2 * W_(DC, k - 1)) - DP(k - 1) * a1 * W_(DC, k));
    }
    {
        // Top (mappm.f90:689-725)
        const float d1 = DP(1), d2 = DP(2);
        const float qm = (d2 * Q(1) + d1 * Q(2)) / (d1 + d2);
        const float dq = 2.f * (Q(2) - Q(1)) / (d1 + d2);
        const float c1 = 4.f * (W_(AL, 3) - qm - d2 * dq) / (d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
        const float c3 = dq - 0.5f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
        float al2 = qm - 0.25f * c1 * d1 * d2 * (d2 + 3.f * d1);
        float al1 = d1 * (2.f * c1 * (d1 * d1) - c3) + al2;
        al2 = f_max2(al2, f_min2(Q(1), Q(2)));
        al2 = f_min2(al2, f_max2(Q(1), Q(2)));
        W_(DC, 1) = 0.5f * (al2 - Q(1));
        if (iv == 0) {
            al1 = f_max2(0.f, al1);
            al2 = f_max2(0.f, al2);
        } else if (iv == -1) {
            if (al1 * Q(1) <= 0.f) al1 = 0.f;
        } else if (iv == 2 || iv == -2) {
            al1 = Q(1);
            // a4(3,i,1) = a4(1,i,1) is overwritten below by a4(3,i,1) = a4(2,i,2)
        }
        W_(AL, 1) = al1;
        W_(AL, 2) = al2;
    }
    {
        // Bottom (mappm.f90:729-761)
        const float d1 = DP(km), d2 = DP(km1);
        const float qm = (d2 * Q(km) + d1 * Q(km1)) / (d1 + d2);
        const float dq = 2.f * (Q(km1) - Q(km)) / (d1 + d2);
        const float c1 = (W_(AL, km1) - qm - d2 * dq) / (d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
        const float c3 = dq - 2.0f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
        float alk = qm - c1 * d1 * d2 * (d2 + 3.f * d1);
        float ark = d1 * (8.f * c1 * (d1 * d1) - c3) + alk;
        alk = f_max2(alk, f_min2(Q(km), Q(km1)));
        alk = f_min2(alk, f_max2(Q(km), Q(km1)));
        W_(DC, km) = 0.5f * (Q(km) - alk);
        if (iv == 0) {
            alk = f_max2(0.f, alk);
            ark = f_max2(0.f, ark);
        } else if (iv < 0) {
            if (Q(km) * ark <= 0.f) ark = 0.f;
        }
        W_(AL, km) = alk;
        W_(AR, km) = ark;
    }
    for (int k = 1; k <= km1; ++k) W_(AR, k) = W_(AL, k + 1);

    auto finish_level = [&](int k, int lmt, bool recompute_a6, bool limit) {
        const float q0 = Q(k);
        float al = W_(AL, k), ar = W_(AR, k), a6 = W_(A6, k);
        if (recompute_a6) a6 = 3.f * (2.f * q0 - (al + ar));
        if (limit) ppm_limiters1(W_(DC, k), q0, al, ar, a6, lmt);
        W_(AL, k) = al;
        W_(AR, k) = ar;
        W_(A6, k) = a6;
    };
    // Top 2 layers always use monotonic mapping (mappm.f90:773-778)
    for (int k = 1; k <= 2; ++k) finish_level(k, 0, true, true);

    if (kord >= 7) {
        // Huynh's 2nd constraint (mappm.f90:780-826)
        for (int k = 2; k <= km1; ++k) {
            const float dpk = DP(k);
            W_(H2, k) = 2.f * (W_(DC, k + 1) / DP(k + 1) - W_(DC, k - 1) / DP(k - 1)) /
                        (dpk + 0.5f * (DP(k - 1) + DP(k + 1))) * (dpk * dpk);
        }
        const float fac = 1.5f;
        for (int k = 3; k <= km - 2; ++k) {
            const float q0 = Q(k), dck = W_(DC, k);
            float al = W_(AL, k), ar = W_(AR, k), a6;
            const float pmp = 2.f * dck;
            float qmp = q0 + pmp;
            float lac = q0 + fac * W_(H2, k - 1) + dck;
            ar = f_min2(f_max2(ar, f_min3(q0, qmp, lac)), f_max3(q0, qmp, lac));
            qmp = q0 - pmp;
            lac = q0 + fac * W_(H2, k + 1) - dck;
            al = f_min2(f_max2(al, f_min3(q0, qmp, lac)), f_max3(q0, qmp, lac));
            a6 = 3.f * (2.f * q0 - (al + ar));
            if (iv == 0 && kord >= 6) ppm_limiters1(dck, q0, al, ar, a6, 2);
            W_(AL, k) = al;
            W_(AR, k) = ar;
            W_(A6, k) = a6;
        }
    } else {
        int lmt = kord - 3;
        lmt = (lmt > 0) ? lmt : 0;
        if (iv == 0) lmt = (lmt < 2) ? lmt : 2;
        for (int k = 3; k <= km - 2; ++k) finish_level(k, lmt, kord != 4, kord != 6);
    }
    for (int k = km1; k <= km; ++k) finish_level(k, 0, true, true);
    }  // kord <= 7

    // ---- remap (mappm.f90:58-124) ----
    const float r3 = 1.f / 3.f, r23 = 2.f / 3.f;
    int k0 = 1, k1 = 1;
    float qsum = 0.f, dpsum = 0.f;
    const float pe1_top = PE1(1), pe1_bot = PE1(km + 1);
    for (int k = 1; k <= kn; ++k) {
        const float p2k = PE2(k);
        float result;
        if (p2k <= pe1_top) {
            result = Q(1);
        } else if (p2k >= pe1_bot) {
            result = Q(km);
        } else {
            const float p2k1 = PE2(k + 1);
            bool done = false;
            for (int L = k0; L <= km; ++L) {
                const float pL = PE1(L), pL1 = PE1(L + 1);
                if (p2k >= pL && p2k <= pL1) {
                    k0 = L;
                    const float dpL = pL1 - pL;
                    const float al = W_(AL, L), ar = W_(AR, L), a6 = W_(A6, L);
                    const float PL = (p2k - pL) / dpL;
                    if (p2k1 <= pL1) {
                        const float PR = (p2k1 - pL) / dpL;
                        const float TT = r3 * (PR * (PR + PL) + PL * PL);
                        result = al + 0.5f * (a6 + ar - al) * (PR + PL) - a6 * TT;
                        done = true;
                    } else {
                        const float delp = pL1 - p2k;
                        const float TT = r3 * (1.f + PL * (1.f + PL));
                        qsum = delp * (al + 0.5f * (a6 + ar - al) * (1.f + PL) - a6 * TT);
                        dpsum = delp;
                        k1 = L + 1;
                    }
                    break;
                }
            }
            if (!done) {
                bool finished = false;
                for (int L = k1; L <= km; ++L) {
                    const float pL = PE1(L), pL1 = PE1(L + 1);
                    const float dpL = pL1 - pL;
                    if (p2k1 > pL1) {
                        qsum = qsum + dpL * Q(L);
                        dpsum = dpsum + dpL;
                    } else {
                        const float delp = p2k1 - pL;
                        const float esl = delp / dpL;
                        const float al = W_(AL, L), ar = W_(AR, L), a6 = W_(A6, L);
                        qsum = qsum + delp * (al + 0.5f * esl * (ar - al + a6 * (1.f - r23 * esl)));
                        dpsum = dpsum + delp;
                        k0 = L;
                        finished = true;
                        break;
                    }
                }
                if (!finished) {
                    const float delp = p2k1 - pe1_bot;
                    if (delp > 0.f) {
                        qsum = qsum + delp * Q(km);
                        dpsum = dpsum + delp;
                    }
                }
                result = qsum / dpsum;
            }
        }
        q2_[o_q2 + (int64_t)(k - 1) * ks] = result;
    }
#undef W_
}

template <typename Tin>
__global__ __launch_bounds__(256) void mappm_simple_kernel(
    const Tin *__restrict__ pe1_, const Tin *__restrict__ q1_, const Tin *__restrict__ pe2_,
    float *__restrict__ q2_, int64_t col0, int64_t col_end, int64_t n_inner, int km, int kn, int iv,
    int kord, int layout, float *__restrict__ ws, int64_t ws_cols)
{
    const int64_t lc = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // column inside the chunk
    const int64_t col = col0 + lc;
    if (col >= col_end) return;
    mappm_column_exact<Tin>(pe1_, q1_, pe2_, q2_, column_addr(col, n_inner, km, kn, layout), km, kn, iv, kord,
                            ws, lc, ws_cols);
}

// ---------------------------------------------------------------------------------------
// mappm, fast path (kord <= 6, km >= 8): one thread per column, no workspace traffic.
//
// The Fortran walks TARGET layers and, for each, searches and integrates over source layers,
// reading a4(2:4, L) at a data-dependent L.  Here the sweep is SOURCE-layer major: the wave
// marches L = 1..km together, the PPM reconstruction of layer L lives in a register sliding
// window (every pe1/q1 load is a coalesced row, each value loaded once), and each lane emits
// the target layers that end inside layer L (a two-pointer merge; only the pe2 loads and q2
// stores are at a per-lane level).  For every (target k, source L) pair the same tests are made
// in the same order with the same single-precision expressions as mappm.f90:58-124, so for
// columns whose pe1 and pe2 are finite and non-decreasing the result is bit-identical to the
// sequential routine.  A lane that meets anything else (NaN or non-monotone pressures, a
// top-edge search that finds no layer) reruns its column through mappm_column_exact.
// ---------------------------------------------------------------------------------------
template <typename Tin>
__global__ __launch_bounds__(256) void mappm_merge_kernel(
    const Tin *__restrict__ pe1_, const Tin *__restrict__ q1_, const Tin *__restrict__ pe2_,
    float *__restrict__ q2_, int64_t col0, int64_t col_end, int64_t n_inner, int km, int kn, int iv,
    int kord, int layout, unsigned int *__restrict__ n_bad, unsigned int *__restrict__ bad_cols)
{
    const int64_t lc = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t col = col0 + lc;
    if (col >= col_end) return;
    const ColumnAddr addr = column_addr(col, n_inner, km, kn, layout);
    const int64_t ks = addr.ks;
    const Tin *pp1 = pe1_ + addr.o_pe1, *pq1 = q1_ + addr.o_q1, *pp2 = pe2_ + addr.o_pe2;
    float *pq2 = q2_ + addr.o_q2;
    auto PE1 = [&](int k) { return (float)pp1[(int64_t)(k - 1) * ks]; };
    auto Q = [&](int k) { return (float)pq1[(int64_t)(k - 1) * ks]; };
    auto PE2g = [&](int k) { return (float)pp2[(int64_t)(k - 1) * ks]; };
    auto OUT = [&](int k, float v) { pq2[(int64_t)(k - 1) * ks] = v; };
    __shared__ float ring_lds[16 * 256];

    const int km1 = km - 1;
    int lmt_int = kord - 3;
    lmt_int = (lmt_int > 0) ? lmt_int : 0;
    if (iv == 0) lmt_int = (lmt_int < 2) ? lmt_int : 2;
    const bool int_recompute_a6 = (kord != 4), int_limit = (kord != 6);
    const float r3 = 1.f / 3.f, r23 = 2.f / 3.f;

    // ---- head of the column: pe1(1..5), q1(1..4), and the two values the pre-checks need ----
    float pe_a = PE1(1), pe_b = PE1(2), pe_c = PE1(3), pe_d = PE1(4), pe_e = PE1(5);
    float q0 = Q(1), qp1 = Q(2), qp2 = Q(3), qp3 = Q(4);
    const float pe1_top = pe_a, pe1_bot = PE1(km + 1), q_top = q0, q_bot = Q(km);
    bool bad = !(pe_b >= pe_a) | !(pe_c >= pe_b) | !(pe_d >= pe_c) | !(pe_e >= pe_d);
    float d0 = pe_b - pe_a, dp1 = pe_c - pe_b, dp2 = pe_d - pe_c, dp3 = pe_e - pe_d;  // dp(L..L+3)
    float dm1 = 0.f, qm1 = 0.f;                                                       // dp(L-1), q(L-1)

    // dc(k) for 2 <= k <= km-1 (mappm.f90:658-668) from the values around level k
    auto DCI = [&](float dpa, float dpb, float dpc, float qa, float qb, float qc) {
        const float d4b = dpa + dpb, d4c = dpb + dpc;  // d4(k), d4(k+1)
        const float c1 = (dpa + 0.5f * dpb) / d4c;
        const float c2 = (dpc + 0.5f * dpb) / d4b;
        const float df2 = dpb * (c1 * (qc - qb) + c2 * (qb - qa)) / (d4b + dpc);
        return f_sign(f_min3(fabsf(df2), f_max3(qa, qb, qc) - qb, qb - f_min3(qa, qb, qc)), df2);
    };
    // a4(2,k) for 3 <= k <= km-1 (mappm.f90:674-683): dpz..dpc = dp(k-2..k+1), qa = q(k-1), qb = q(k)
    auto INT = [&](float dpz, float dpa, float dpb, float dpc, float qa, float qb, float dca, float dcb) {
        const float d4a = dpz + dpa, d4b = dpa + dpb, d4c = dpb + dpc;  // d4(k-1), d4(k), d4(k+1)
        const float c1 = (qb - qa) * dpa / d4b;
        const float a1 = d4a / (d4b + dpa);
        const float a2 = d4c / (d4b + dpb);
        return qa + c1 + 2.f / (d4a + d4c) * (dpb * (c1 * (a1 - a2) + a2 * dca) - dpa * a1 * dcb);
    };

    // ---- prologue: dc(2), dc(3), al(3), then the top boundary (mappm.f90:689-725) ----
    float dc1 = DCI(d0, dp1, dp2, q0, qp1, qp2);                 // dc(2)
    const float dc_3 = DCI(dp1, dp2, dp3, qp1, qp2, qp3);        // dc(3)
    const float al_3 = INT(d0, dp1, dp2, dp3, qp1, qp2, dc1, dc_3);
    float al0, al1, dc0, ar_km = 0.f;
    {
        const float d1 = d0, d2 = dp1;
        const float qm = (d2 * q0 + d1 * qp1) / (d1 + d2);
        const float dq = 2.f * (qp1 - q0) / (d1 + d2);
        const float c1 = 4.f * (al_3 - qm - d2 * dq) / (d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
        const float c3 = dq - 0.5f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
        float a2 = qm - 0.25f * c1 * d1 * d2 * (d2 + 3.f * d1);
        float a1 = d1 * (2.f * c1 * (d1 * d1) - c3) + a2;
        a2 = f_max2(a2, f_min2(q0, qp1));
        a2 = f_min2(a2, f_max2(q0, qp1));
        dc0 = 0.5f * (a2 - q0);
        if (iv == 0) {
            a1 = f_max2(0.f, a1);
            a2 = f_max2(0.f, a2);
        } else if (iv == -1) {
            if (a1 * q0 <= 0.f) a1 = 0.f;
        } else if (iv == 2 || iv == -2) {
            a1 = q0;
        }
        al0 = a1;
        al1 = a2;
    }

    // ---- per-lane target cursor ----
    // For finite, non-decreasing pressures the targets of a column come in three runs: those
    // whose top edge is at or above the old top (copy q1(1), mappm.f90:62-64), those that
    // start inside the column (the merge below), those that start at or below the old surface
    // (copy q1(km), mappm.f90:65-67).  `live` = this lane still has a target of the middle run.
    //
    // The target interfaces are consumed at a per-lane level.  Reading them from HBM where they
    // are needed would cost a memory round trip per emitted target, so each lane keeps a private
    // ring of the next kRing interfaces of its column in LDS ([slot][thread], conflict-free):
    // rows are requested at the top of a source-layer iteration and written to the ring at the
    // top of the next one, so the emit loop touches LDS only and every global load of the kernel
    // has a whole iteration to land.
    constexpr int kRing = 16;
    float *ring = ring_lds + threadIdx.x;
    auto RING = [&](int j) -> float & { return ring[(j & (kRing - 1)) * 256]; };
    int jw = 1;  // interfaces < jw are in the ring (those >= jw - kRing are still there)
    {
        float tmp[kRing];
#pragma unroll
        for (int i = 0; i < kRing; ++i) tmp[i] = PE2g((i + 1 <= kn + 1) ? i + 1 : kn + 1);
#pragma unroll
        for (int i = 0; i < kRing; ++i) RING(i + 1) = tmp[i];
        jw = (kRing < kn + 1 ? kRing : kn + 1) + 1;
    }
    int jp = jw;            // interfaces in [jw, jp) have been requested (at most two)
    float pv0 = 0.f, pv1 = 0.f;
    auto PE2 = [&](int j) { return (j < jw) ? RING(j) : PE2g(j); };  // (j >= jw - kRing by construction)

    int k = 1;
    float p2k = PE2(1), p2k1 = PE2(2);  // pe2(k), pe2(k+1)
    bool accum = false;
    float qsum = 0.f, dpsum = 0.f;
    auto advance = [&]() {
        ++k;
        if (!(p2k1 >= p2k)) bad = true;  // also catches NaN
        p2k = p2k1;
        p2k1 = PE2(k + 1 <= kn + 1 ? k + 1 : kn + 1);
    };
    if (!(p2k1 >= p2k)) bad = true;
    while (k <= kn && !bad && p2k <= pe1_top) {
        OUT(k, q_top);
        advance();
    }
    bool live = (k <= kn) && !bad && !(p2k >= pe1_bot);

    float q_in = 0.f, pe_in = pe_e;  // q(L+3), pe1(L+4) for the NEXT iteration's window, in flight
    float al2 = 0.f, dc2 = 0.f;
    for (int L = 1; L <= km; ++L) {
        // ---- (1) everything requested during the previous iteration lands here ----
        if (jp > jw) RING(jw) = pv0;
        if (jp > jw + 1) RING(jw + 1) = pv1;
        jw = jp;
        if (L > 1) {  // level L becomes the current one
            if (!(pe_in >= pe_e)) bad = true;
            qm1 = q0; q0 = qp1; qp1 = qp2; qp2 = qp3; qp3 = q_in;
            dm1 = d0; d0 = dp1; dp1 = dp2; dp2 = dp3; dp3 = pe_in - pe_e;
            pe_a = pe_b; pe_b = pe_c; pe_c = pe_d; pe_d = pe_e; pe_e = pe_in;
            al0 = al1; al1 = al2;
            dc0 = dc1; dc1 = dc2;
        }
        // ---- (2) reconstruction of level L+2 from the window dp(L-1..L+3), q(L..L+3) ----
        const int kk = L + 2;
        al2 = 0.f;
        dc2 = 0.f;
        if (kk <= km1) {
            dc2 = DCI(dp1, dp2, dp3, qp1, qp2, qp3);           // dc(L+2)
            al2 = INT(d0, dp1, dp2, dp3, qp1, qp2, dc1, dc2);  // al(L+2)
        } else if (kk == km) {
            // bottom boundary (mappm.f90:729-761): al(km), ar(km), dc(km) from al(km-1) = al1
            const float d1 = dp2, d2 = dp1;   // dp(km), dp(km-1)
            const float qk = qp2, qk1 = qp1;  // q(km), q(km-1)
            const float qm = (d2 * qk + d1 * qk1) / (d1 + d2);
            const float dq = 2.f * (qk1 - qk) / (d1 + d2);
            const float c1 = (al1 - qm - d2 * dq) / (d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
            const float c3 = dq - 2.0f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
            float alk = qm - c1 * d1 * d2 * (d2 + 3.f * d1);
            float ark = d1 * (8.f * c1 * (d1 * d1) - c3) + alk;
            alk = f_max2(alk, f_min2(qk, qk1));
            alk = f_min2(alk, f_max2(qk, qk1));
            dc2 = 0.5f * (qk - alk);
            if (iv == 0) {
                alk = f_max2(0.f, alk);
                ark = f_max2(0.f, ark);
            } else if (iv < 0) {
                if (qk * ark <= 0.f) ark = 0.f;
            }
            al2 = alk;
            ar_km = ark;
        }
        // ---- (3) requests for the next iteration: q(L+4), pe1(L+5), up to two target interfaces ----
        if (L + 4 <= km) {
            q_in = Q(L + 4);
            pe_in = PE1(L + 5);
        }
        if (jp <= kn + 1 && jp < k + kRing) {
            pv0 = PE2g(jp);
            ++jp;
            if (jp <= kn + 1 && jp < k + kRing) {
                pv1 = PE2g(jp);
                ++jp;
            }
        }

        const bool edge = (L <= 2) | (L >= km1);
        // ---- (4) finalise layer L: A6 and the limiter (mappm.f90:773-849) ----
        float al = al0, ar = (L == km) ? ar_km : al1, a6 = 0.f;
        if (edge | int_recompute_a6) a6 = 3.f * (2.f * q0 - (al + ar));
        if (edge | int_limit) ppm_limiters1(dc0, q0, al, ar, a6, edge ? 0 : lmt_int);
        const float pL = pe_a, pL1 = pe_b;

        // ---- (5) emit the target layers that end inside layer L ----
        // Per layer a lane goes through zero or more emitting events (the bottom part of an
        // accumulating target; targets lying entirely inside the layer) and then exactly one
        // non-emitting one (start a target that leaves the layer / add the whole layer / nothing).
        // an accumulating target can end in this layer only once, before any target that lies inside it:
        // that step is taken out of the loop, so that lanes closing a target and lanes emitting inside
        // ones do not serialise each other's branch
        if (live && accum && !(p2k1 > pL1)) {
            const float delp = p2k1 - pL;
            const float PR = delp / d0;
            qsum = qsum + delp * (al + 0.5f * PR * (ar - al + a6 * (1.f - r23 * PR)));
            dpsum = dpsum + delp;
            OUT(k, qsum / dpsum);
            accum = false;
            advance();
            live = (k <= kn) && !bad && !(p2k >= pe1_bot);
        }
        while (live && !accum && (p2k >= pL && p2k <= pL1) && (p2k1 <= pL1)) {
            const float PR = (p2k1 - pL) / d0;
            const float PL = (p2k - pL) / d0;
            const float TT = r3 * (PR * (PR + PL) + PL * PL);
            OUT(k, al + 0.5f * (a6 + ar - al) * (PR + PL) - a6 * TT);
            advance();
            live = (k <= kn) && !bad && !(p2k >= pe1_bot);
        }
        if (live) {
            if (accum) {  // whole layer (mappm.f90:99-104)
                qsum = qsum + d0 * q0;
                dpsum = dpsum + d0;
            } else if (p2k >= pL && p2k <= pL1) {  // fractional area (mappm.f90:85-92)
                const float PL = (p2k - pL) / d0;
                const float delp = pL1 - p2k;
                const float TT = r3 * (1.f + PL * (1.f + PL));
                qsum = delp * (al + 0.5f * (a6 + ar - al) * (1.f + PL) - a6 * TT);
                dpsum = delp;
                accum = true;
            }
        }
    }
    (void)qm1;
    (void)dm1;

    // ---- past the old surface (mappm.f90:115-121), then the run that copies q1(km) ----
    if (k <= kn && !bad && accum) {
        const float delp = p2k1 - pe1_bot;
        if (delp > 0.f) {
            qsum = qsum + delp * q_bot;
            dpsum = dpsum + delp;
        }
        OUT(k, qsum / dpsum);
        advance();
    }
    while (k <= kn && !bad) {
        if (p2k >= pe1_bot) {
            OUT(k, q_bot);
            advance();
        } else {
            bad = true;  // a top-edge search that no source layer satisfied
        }
    }
    if (bad) bad_cols[atomicAdd(n_bad, 1u)] = (unsigned int)lc;  // redone by mappm_fallback_kernel
}

// ---------------------------------------------------------------------------------------
// The merge sweep for NF fields that share their source and target pressures (the restart pipelines
// remap 4 fv_core fields and 9 tracers between the same two pressure grids, regridz.py:163-185): the
// control flow, the target-interface ring, the pressure loads and every pressure-only term of the
// reconstruction (the c1/c2/a1/a2 ratios, 2/(d4a+d4c), PR, PL, TT) are computed once per column
// instead of once per field -- about 7 of the ~10 IEEE divisions per layer.  Per field the operations
// and their order are those of mappm_merge_kernel (the field loops are unrolled and the compiler
// merges the identical pressure-only subexpressions), so each field's result is bit-identical to a
// single-field call.  `bad` depends on the pressures only: one worklist for all NF fields.
// ---------------------------------------------------------------------------------------
constexpr int kMaxMultiFields = 4;
struct MultiFieldPtrs {
    const void *q1[kMaxMultiFields];
    float *q2[kMaxMultiFields];
};

template <typename Tin, int NF>
__global__ __launch_bounds__(256) void mappm_merge_multi_kernel(
    const Tin *__restrict__ pe1_, const MultiFieldPtrs fp, const Tin *__restrict__ pe2_, int64_t col0, int64_t col_end,
    int64_t n_inner, int km, int kn, int iv, int kord, int layout, unsigned int *__restrict__ n_bad,
    unsigned int *__restrict__ bad_cols)
{
    const int64_t lc = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t col = col0 + lc;
    if (col >= col_end) return;
    const ColumnAddr addr = column_addr(col, n_inner, km, kn, layout);
    const int64_t ks = addr.ks;
    const Tin *pp1 = pe1_ + addr.o_pe1, *pp2 = pe2_ + addr.o_pe2;
    const Tin *pq1[NF];
    float *pq2[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        pq1[f] = static_cast<const Tin *>(fp.q1[f]) + addr.o_q1;
        pq2[f] = fp.q2[f] + addr.o_q2;
    }
    auto PE1 = [&](int k) { return (float)pp1[(int64_t)(k - 1) * ks]; };
    auto Q = [&](int f, int k) { return (float)pq1[f][(int64_t)(k - 1) * ks]; };
    auto PE2g = [&](int k) { return (float)pp2[(int64_t)(k - 1) * ks]; };
    auto OUT = [&](int f, int k, float v) { pq2[f][(int64_t)(k - 1) * ks] = v; };
    __shared__ float ring_lds[16 * 256];

    const int km1 = km - 1;
    int lmt_int = kord - 3;
    lmt_int = (lmt_int > 0) ? lmt_int : 0;
    if (iv == 0) lmt_int = (lmt_int < 2) ? lmt_int : 2;
    const bool int_recompute_a6 = (kord != 4), int_limit = (kord != 6);
    const float r3 = 1.f / 3.f, r23 = 2.f / 3.f;

    float pe_a = PE1(1), pe_b = PE1(2), pe_c = PE1(3), pe_d = PE1(4), pe_e = PE1(5);
    const float pe1_top = pe_a, pe1_bot = PE1(km + 1);
    float q0[NF], qp1[NF], qp2[NF], qp3[NF], q_top[NF], q_bot[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        q0[f] = Q(f, 1);
        qp1[f] = Q(f, 2);
        qp2[f] = Q(f, 3);
        qp3[f] = Q(f, 4);
        q_top[f] = q0[f];
        q_bot[f] = Q(f, km);
    }
    bool bad = !(pe_b >= pe_a) | !(pe_c >= pe_b) | !(pe_d >= pe_c) | !(pe_e >= pe_d);
    float d0 = pe_b - pe_a, dp1 = pe_c - pe_b, dp2 = pe_d - pe_c, dp3 = pe_e - pe_d;

    auto DCI = [&](float dpa, float dpb, float dpc, float qa, float qb, float qc) {
        const float d4b = dpa + dpb, d4c = dpb + dpc;
        const float c1 = (dpa + 0.5f * dpb) / d4c;
        const float c2 = (dpc + 0.5f * dpb) / d4b;
        const float df2 = dpb * (c1 * (qc - qb) + c2 * (qb - qa)) / (d4b + dpc);
        return f_sign(f_min3(fabsf(df2), f_max3(qa, qb, qc) - qb, qb - f_min3(qa, qb, qc)), df2);
    };
    auto INT = [&](float dpz, float dpa, float dpb, float dpc, float qa, float qb, float dca, float dcb) {
        const float d4a = dpz + dpa, d4b = dpa + dpb, d4c = dpb + dpc;
        const float c1 = (qb - qa) * dpa / d4b;
        const float a1 = d4a / (d4b + dpa);
        const float a2 = d4c / (d4b + dpb);
        return qa + c1 + 2.f / (d4a + d4c) * (dpb * (c1 * (a1 - a2) + a2 * dca) - dpa * a1 * dcb);
    };

    float al0[NF], al1[NF], al2[NF], dc0[NF], dc1[NF], dc2[NF], ar_km[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        dc1[f] = DCI(d0, dp1, dp2, q0[f], qp1[f], qp2[f]);                   // dc(2)
        const float dc_3 = DCI(dp1, dp2, dp3, qp1[f], qp2[f], qp3[f]);       // dc(3)
        const float al_3 = INT(d0, dp1, dp2, dp3, qp1[f], qp2[f], dc1[f], dc_3);
        const float d1 = d0, d2 = dp1;
        const float qm = (d2 * q0[f] + d1 * qp1[f]) / (d1 + d2);
        const float dq = 2.f * (qp1[f] - q0[f]) / (d1 + d2);
        const float c1 = 4.f * (al_3 - qm - d2 * dq) / (d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
        const float c3 = dq - 0.5f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
        float a2 = qm - 0.25f * c1 * d1 * d2 * (d2 + 3.f * d1);
        float a1 = d1 * (2.f * c1 * (d1 * d1) - c3) + a2;
        a2 = f_max2(a2, f_min2(q0[f], qp1[f]));
        a2 = f_min2(a2, f_max2(q0[f], qp1[f]));
        dc0[f] = 0.5f * (a2 - q0[f]);
        if (iv == 0) {
            a1 = f_max2(0.f, a1);
            a2 = f_max2(0.f, a2);
        } else if (iv == -1) {
            if (a1 * q0[f] <= 0.f) a1 = 0.f;
        } else if (iv == 2 || iv == -2) {
            a1 = q0[f];
        }
        al0[f] = a1;
        al1[f] = a2;
        al2[f] = 0.f;
        dc2[f] = 0.f;
        ar_km[f] = 0.f;
    }

    constexpr int kRing = 16;
    float *ring = ring_lds + threadIdx.x;
    auto RING = [&](int j) -> float & { return ring[(j & (kRing - 1)) * 256]; };
    int jw = 1;
    {
        float tmp[kRing];
#pragma unroll
        for (int i = 0; i < kRing; ++i) tmp[i] = PE2g((i + 1 <= kn + 1) ? i + 1 : kn + 1);
#pragma unroll
        for (int i = 0; i < kRing; ++i) RING(i + 1) = tmp[i];
        jw = (kRing < kn + 1 ? kRing : kn + 1) + 1;
    }
    int jp = jw;
    float pv0 = 0.f, pv1 = 0.f;
    auto PE2 = [&](int j) { return (j < jw) ? RING(j) : PE2g(j); };

    int k = 1;
    float p2k = PE2(1), p2k1 = PE2(2);
    bool accum = false;
    float qsum[NF], dpsum = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) qsum[f] = 0.f;
    auto advance = [&]() {
        ++k;
        if (!(p2k1 >= p2k)) bad = true;
        p2k = p2k1;
        p2k1 = PE2(k + 1 <= kn + 1 ? k + 1 : kn + 1);
    };
    if (!(p2k1 >= p2k)) bad = true;
    while (k <= kn && !bad && p2k <= pe1_top) {
#pragma unroll
        for (int f = 0; f < NF; ++f) OUT(f, k, q_top[f]);
        advance();
    }
    bool live = (k <= kn) && !bad && !(p2k >= pe1_bot);

    float q_in[NF], pe_in = pe_e;
#pragma unroll
    for (int f = 0; f < NF; ++f) q_in[f] = 0.f;
    for (int L = 1; L <= km; ++L) {
        if (jp > jw) RING(jw) = pv0;
        if (jp > jw + 1) RING(jw + 1) = pv1;
        jw = jp;
        if (L > 1) {
            if (!(pe_in >= pe_e)) bad = true;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                q0[f] = qp1[f]; qp1[f] = qp2[f]; qp2[f] = qp3[f]; qp3[f] = q_in[f];
                al0[f] = al1[f]; al1[f] = al2[f];
                dc0[f] = dc1[f]; dc1[f] = dc2[f];
            }
            d0 = dp1; dp1 = dp2; dp2 = dp3; dp3 = pe_in - pe_e;
            pe_a = pe_b; pe_b = pe_c; pe_c = pe_d; pe_d = pe_e; pe_e = pe_in;
        }
        const int kk = L + 2;
        if (kk <= km1) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                dc2[f] = DCI(dp1, dp2, dp3, qp1[f], qp2[f], qp3[f]);
                al2[f] = INT(d0, dp1, dp2, dp3, qp1[f], qp2[f], dc1[f], dc2[f]);
            }
        } else if (kk == km) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float d1 = dp2, d2 = dp1;
                const float qk = qp2[f], qk1 = qp1[f];
                const float qm = (d2 * qk + d1 * qk1) / (d1 + d2);
                const float dq = 2.f * (qk1 - qk) / (d1 + d2);
                const float c1 = (al1[f] - qm - d2 * dq) / (d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
                const float c3 = dq - 2.0f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
                float alk = qm - c1 * d1 * d2 * (d2 + 3.f * d1);
                float ark = d1 * (8.f * c1 * (d1 * d1) - c3) + alk;
                alk = f_max2(alk, f_min2(qk, qk1));
                alk = f_min2(alk, f_max2(qk, qk1));
                dc2[f] = 0.5f * (qk - alk);
                if (iv == 0) {
                    alk = f_max2(0.f, alk);
                    ark = f_max2(0.f, ark);
                } else if (iv < 0) {
                    if (qk * ark <= 0.f) ark = 0.f;
                }
                al2[f] = alk;
                ar_km[f] = ark;
            }
        } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                al2[f] = 0.f;
                dc2[f] = 0.f;
            }
        }
        if (L + 4 <= km) {
#pragma unroll
            for (int f = 0; f < NF; ++f) q_in[f] = Q(f, L + 4);
            pe_in = PE1(L + 5);
        }
        if (jp <= kn + 1 && jp < k + kRing) {
            pv0 = PE2g(jp);
            ++jp;
            if (jp <= kn + 1 && jp < k + kRing) {
                pv1 = PE2g(jp);
                ++jp;
            }
        }

        const bool edge = (L <= 2) | (L >= km1);
        float al[NF], ar[NF], a6[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            al[f] = al0[f];
            ar[f] = (L == km) ? ar_km[f] : al1[f];
            a6[f] = 0.f;
            if (edge | int_recompute_a6) a6[f] = 3.f * (2.f * q0[f] - (al[f] + ar[f]));
            if (edge | int_limit) ppm_limiters1(dc0[f], q0[f], al[f], ar[f], a6[f], edge ? 0 : lmt_int);
        }
        const float pL = pe_a, pL1 = pe_b;

        if (live && accum && !(p2k1 > pL1)) {  // (see mappm_merge_kernel)
            const float delp = p2k1 - pL;
            const float PR = delp / d0;
            dpsum = dpsum + delp;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                qsum[f] = qsum[f] + delp * (al[f] + 0.5f * PR * (ar[f] - al[f] + a6[f] * (1.f - r23 * PR)));
                OUT(f, k, qsum[f] / dpsum);
            }
            accum = false;
            advance();
            live = (k <= kn) && !bad && !(p2k >= pe1_bot);
        }
        while (live && !accum && (p2k >= pL && p2k <= pL1) && (p2k1 <= pL1)) {
            const float PR = (p2k1 - pL) / d0;
            const float PL = (p2k - pL) / d0;
            const float TT = r3 * (PR * (PR + PL) + PL * PL);
#pragma unroll
            for (int f = 0; f < NF; ++f) OUT(f, k, al[f] + 0.5f * (a6[f] + ar[f] - al[f]) * (PR + PL) - a6[f] * TT);
            advance();
            live = (k <= kn) && !bad && !(p2k >= pe1_bot);
        }
        if (live) {
            if (accum) {
#pragma unroll
                for (int f = 0; f < NF; ++f) qsum[f] = qsum[f] + d0 * q0[f];
                dpsum = dpsum + d0;
            } else if (p2k >= pL && p2k <= pL1) {
                const float PL = (p2k - pL) / d0;
                const float delp = pL1 - p2k;
                const float TT = r3 * (1.f + PL * (1.f + PL));
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    qsum[f] = delp * (al[f] + 0.5f * (a6[f] + ar[f] - al[f]) * (1.f + PL) - a6[f] * TT);
                dpsum = delp;
                accum = true;
            }
        }
    }

    if (k <= kn && !bad && accum) {
        const float delp = p2k1 - pe1_bot;
        if (delp > 0.f) {
#pragma unroll
            for (int f = 0; f < NF; ++f) qsum[f] = qsum[f] + delp * q_bot[f];
            dpsum = dpsum + delp;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) OUT(f, k, qsum[f] / dpsum);
        advance();
    }
    while (k <= kn && !bad) {
        if (p2k >= pe1_bot) {
#pragma unroll
            for (int f = 0; f < NF; ++f) OUT(f, k, q_bot[f]);
            advance();
        } else {
            bad = true;
        }
    }
    if (bad) bad_cols[atomicAdd(n_bad, 1u)] = (unsigned int)lc;
}

// The columns the merge sweep gave up on (listed by their index inside the chunk), through the
// sequential routine.  Launched after every merge launch; exits at once when the list is empty.
template <typename Tin>
__global__ __launch_bounds__(256) void mappm_fallback_kernel(
    const Tin *__restrict__ pe1_, const Tin *__restrict__ q1_, const Tin *__restrict__ pe2_,
    float *__restrict__ q2_, int64_t col0, int64_t n_inner, int km, int kn, int iv, int kord, int layout,
    const unsigned int *__restrict__ n_bad, const unsigned int *__restrict__ bad_cols,
    float *__restrict__ ws, int64_t ws_cols, int pe2_f = 0, int nx = 0, int nxc = 0, int64_t plane2 = 0)
{
    const unsigned int count = *n_bad;
    const int64_t slot = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t i = slot; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = col0 + bad_cols[i];
        const ColumnAddr addr =
            pe2_f > 1 ? column_addr_coarse_target(col, n_inner, km, kn, pe2_f, nx, nxc, plane2) : column_addr(col, n_inner, km, kn, layout);
        mappm_column_exact<Tin>(pe1_, q1_, pe2_, q2_, addr, km, kn, iv, kord, ws, slot, ws_cols);
    }
}

constexpr int64_t kMappmChunk = 1 << 20;  // columns per launch; bounds the workspace
constexpr int kMappmPlanes = 5;           // AL, AR, A6, DC, H2
constexpr int64_t kFallbackSlots = 256 * 256;  // threads (= reconstruction slots) of the fallback pass
constexpr size_t kCounterBytes = 256;

// workspace: [counter][bad-column list: chunk x u32][5 planes x km x slots floats]
inline int64_t ws_slots(int64_t ncol) { return ncol < kMappmChunk ? ncol : kMappmChunk; }
inline size_t ws_list_bytes(int64_t ncol) { return ((size_t)ws_slots(ncol) * 4 + 255) & ~(size_t)255; }


// ---------------------------------------------------------------------------------------
// interpolate_2d (external/mappm/mappm/interpolate_2d.f90:1-28), the other routine of the reference's
// native module: per column, linear interpolation of y(x) onto the points xp.  The search runs over
// every interval (a later match overwrites an earlier one, as in the Fortran loop); points outside
// the column's range keep fill_value.  One thread per output point; float64 throughout.
// ---------------------------------------------------------------------------------------
__global__ void interpolate_2d_kernel(const double *__restrict__ xp, const double *__restrict__ x,
                                      const double *__restrict__ y, double *__restrict__ out, double fill_value,
                                      int64_t n_batch, int64_t n_inner, int n_in, int n_out, int layout)
{
    const int64_t ncol = n_batch * n_inner;
    const int64_t total = ncol * n_out;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t col, base_in, base_out, ks;
        int j;
        if (layout == FV3HIP_LAYOUT_LEVEL_COL) {  // [batch][level][inner]: adjacent threads = adjacent columns
            const int64_t inner = idx % n_inner;
            const int64_t r = idx / n_inner;
            j = (int)(r % n_out);
            const int64_t bt = r / n_out;
            col = bt * n_inner + inner;
            ks = n_inner;
            base_in = bt * (int64_t)n_in * n_inner + inner;
            base_out = bt * (int64_t)n_out * n_inner + inner;
        } else {                                   // [column][level]
            j = (int)(idx % n_out);
            col = idx / n_out;
            ks = 1;
            base_in = col * (int64_t)n_in;
            base_out = col * (int64_t)n_out;
        }
        (void)col;
        const double p = xp[base_out + (int64_t)j * ks];
        double r = fill_value;
        double x0 = x[base_in], y0 = y[base_in];
        for (int k = 0; k < n_in - 1; ++k) {
            const double x1 = x[base_in + (int64_t)(k + 1) * ks], y1 = y[base_in + (int64_t)(k + 1) * ks];
            if (x0 <= p && p < x1) {
                const double w = (p - x0) / (x1 - x0);
                r = y0 * (1 - w) + y1 * w;
            } else if (x0 == p) {
                r = y0;
            } else if (x1 == p) {
                r = y1;
            }
            x0 = x1;
            y0 = y1;
        }
        out[base_out + (int64_t)j * ks] = r;
    }
}


// ---------------------------------------------------------------------------------------
// Column helpers of the restart pipelines (external/vcm/vcm/cubedsphere/coarsen_restarts.py:559-676,
// 990-1017; external/vcm/vcm/calc/thermo/vertically_dependent.py:69-99,182-235).  Arrays are
// [n_batch][nz][n_inner] (level-major columns, adjacent threads = adjacent columns).
// ---------------------------------------------------------------------------------------
// surface_pressure_from_delp: sum over the levels + addend
template <typename T>
__global__ void column_sum_kernel(const T *__restrict__ x, T *__restrict__ out, int64_t n_batch, int nz, int64_t n_inner,
                                  T addend)
{
    const int64_t ncol = n_batch * n_inner;
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < ncol; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = c / n_inner, i = c - b * n_inner;
        const T *p = x + b * (int64_t)nz * n_inner + i;
        T acc = 0;
        for (int k = 0; k < nz; ++k) acc += p[(int64_t)k * n_inner];
        out[c] = acc + addend;
    }
}

// compute_blending_weights: (ps - p) / (ps - pb) where p > pb, else 1
template <typename T>
__global__ void blend_weights_kernel(const T *__restrict__ pb, const T *__restrict__ ps, const T *__restrict__ pfull,
                                     T *__restrict__ out, int64_t n_batch, int nz, int64_t n_inner)
{
    const int64_t total = n_batch * nz * n_inner;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx % n_inner, b = idx / (n_inner * nz);
        const T p = pfull[idx], s = ps[b * n_inner + i], q = pb[b * n_inner + i];
        out[idx] = (p > q) ? (s - p) / (s - q) : (T)1;
    }
}

// _impose_hydrostatic_balance: DZ from the hypsometric equation with the virtual temperature, phis such
// that the model-top height is unchanged
template <typename T>
__global__ void hydrostatic_kernel(const T *__restrict__ dz, const T *__restrict__ phis, const T *__restrict__ t,
                                   const T *__restrict__ q, const T *__restrict__ delp, T *__restrict__ dz_out,
                                   T *__restrict__ phis_out, int64_t n_batch, int nz, int64_t n_inner, T toa)
{
    const T g = (T)9.80665, rd = (T)287.05, rv = (T)461.5;
    const int64_t ncol = n_batch * n_inner;
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < ncol; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = c / n_inner, i = c - b * n_inner;
        const int64_t base = b * (int64_t)nz * n_inner + i;
        // height of the model top: cumulative sum from the surface upwards of [-dz..., phis / g]
        T top = phis[c] / g;
        for (int k = nz - 1; k >= 0; --k) top = top + (-dz[base + (int64_t)k * n_inner]);
        // new thicknesses
        T p_hi = toa, lp_hi = log(p_hi), sum = 0;
        for (int k = 0; k < nz; ++k) {
            const int64_t o = base + (int64_t)k * n_inner;
            const T p_lo = p_hi + delp[o];
            const T lp_lo = log(p_lo);
            const T tv = t[o] * ((T)1 + (rv / rd - (T)1) * q[o]);
            const T d = -(lp_lo - lp_hi) * rd * tv / g;
            dz_out[o] = d;
            sum += d;
            p_hi = p_lo;
            lp_hi = lp_lo;
        }
        phis_out[c] = g * (top + sum);
    }
}

}  // namespace
}  // namespace fv3hip

using namespace fv3hip;

extern "C" int fv3hip_pressure_at_interface(const void *delp, int dtype, int64_t n_batch, int nz,
                                            int64_t n_inner, double toa_pressure, void *out,
                                            void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64, got %d", dtype);
    FV3HIP_REQUIRE(n_batch >= 0 && nz >= 0 && n_inner >= 0, "negative extent");
    const int64_t ncol = n_batch * n_inner;
    if (ncol == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(delp && out, "null pointer");
    int64_t blocks = ceil_div(ncol, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F32)
        hipLaunchKernelGGL((pressure_at_interface_kernel<float>), dim3((unsigned)blocks), dim3(256), 0,
                           st, static_cast<const float *>(delp), static_cast<float *>(out), n_batch,
                           nz, n_inner, (float)toa_pressure);
    else
        hipLaunchKernelGGL((pressure_at_interface_kernel<double>), dim3((unsigned)blocks), dim3(256),
                           0, st, static_cast<const double *>(delp), static_cast<double *>(out),
                           n_batch, nz, n_inner, toa_pressure);
    return check_launch("pressure_at_interface_kernel");
}

extern "C" int fv3hip_pressure_at_midpoint_log(const void *delp, int dtype, int64_t n_batch, int nz,
                                               int64_t n_inner, double toa_pressure, void *out,
                                               void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64, got %d", dtype);
    FV3HIP_REQUIRE(n_batch >= 0 && nz >= 0 && n_inner >= 0, "negative extent");
    const int64_t ncol = n_batch * n_inner;
    if (ncol == 0 || nz == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(delp && out, "null pointer");
    int64_t blocks = ceil_div(ncol, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F32)
        hipLaunchKernelGGL((pressure_at_midpoint_log_kernel<float>), dim3((unsigned)blocks), dim3(256),
                           0, st, static_cast<const float *>(delp), static_cast<float *>(out), n_batch,
                           nz, n_inner, (float)toa_pressure);
    else
        hipLaunchKernelGGL((pressure_at_midpoint_log_kernel<double>), dim3((unsigned)blocks), dim3(256),
                           0, st, static_cast<const double *>(delp), static_cast<double *>(out),
                           n_batch, nz, n_inner, toa_pressure);
    return check_launch("pressure_at_midpoint_log_kernel");
}

extern "C" int fv3hip_mask_weights(const void *weights, int w_dtype, const void *p_cmp, int cmp_levels,
                                   int cmp_offset, const void *p_fine, int p_dtype, int64_t n_batch,
                                   int nz, int64_t n_inner, int64_t w_repeat, void *out, void *stream)
{
    FV3HIP_REQUIRE(w_dtype == FV3HIP_F32 || w_dtype == FV3HIP_F64, "weights dtype must be F32 or F64");
    FV3HIP_REQUIRE(p_dtype == FV3HIP_F32 || p_dtype == FV3HIP_F64, "pressure dtype must be F32 or F64");
    FV3HIP_REQUIRE(w_repeat >= 1 && n_batch % w_repeat == 0, "bad w_repeat %lld", (long long)w_repeat);
    FV3HIP_REQUIRE(cmp_offset >= 0 && cmp_levels >= nz + cmp_offset,
                   "p_cmp has %d levels, need at least nz + cmp_offset = %d", cmp_levels, nz + cmp_offset);
    const int64_t total = n_batch * nz * n_inner;
    if (total <= 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(weights && p_cmp && p_fine && out, "null pointer");
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t st = as_stream(stream);
    // rows grid (no per-element division) when a row is whole quads of 16-byte aligned columns and the grid fits
    const bool rows = n_inner % 4 == 0 && n_batch * nz <= 65535 && reinterpret_cast<uintptr_t>(weights) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(p_cmp) % 16 == 0 && reinterpret_cast<uintptr_t>(p_fine) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(out) % 16 == 0;
#define LAUNCH_(TW, TP)                                                                              \
    if (rows)                                                                                        \
        hipLaunchKernelGGL((mask_weights_rows_kernel<TW, TP>), dim3((unsigned)ceil_div(n_inner, 1024), (unsigned)(n_batch * nz)), dim3(256), 0, st, \
                           static_cast<const TW *>(weights), static_cast<const TP *>(p_cmp), static_cast<const TP *>(p_fine),                      \
                           static_cast<TW *>(out), nz, n_inner, w_repeat, cmp_levels, cmp_offset);                                                 \
    else                                                                                             \
        hipLaunchKernelGGL((mask_weights_kernel<TW, TP>), dim3((unsigned)blocks), dim3(256), 0, st,  \
                           static_cast<const TW *>(weights), static_cast<const TP *>(p_cmp),         \
                           static_cast<const TP *>(p_fine), static_cast<TW *>(out), n_batch, nz, n_inner, \
                           w_repeat, cmp_levels, cmp_offset)
    if (w_dtype == FV3HIP_F32 && p_dtype == FV3HIP_F32) LAUNCH_(float, float);
    else if (w_dtype == FV3HIP_F32) LAUNCH_(float, double);
    else if (p_dtype == FV3HIP_F32) LAUNCH_(double, float);
    else LAUNCH_(double, double);
#undef LAUNCH_
    return check_launch("mask_weights_kernel");
}

// the coarse extent of a fine extent under block_upsample's rule (coarsen.py:843-897): an odd extent is a staggered dim whose last
// point is not repeated
inline int coarse_extent(int n, int factor) { return (n % 2 == 1) ? (n - 1) / factor + 1 : n / factor; }
inline bool coarse_divides(int n, int factor) { return ((n % 2 == 1) ? (n - 1) : n) % factor == 0; }

// fv3hip_mask_weights with p_cmp on a grid coarser by `factor` in y and x: [n_batch][cmp_levels][ny / factor][nx / factor]
extern "C" int fv3hip_mask_weights_coarse(const void *weights, int w_dtype, const void *p_cmp_coarse, int cmp_levels, int cmp_offset,
                                          const void *p_fine, int p_dtype, int64_t n_batch, int nz, int ny, int nx, int factor,
                                          int64_t w_repeat, void *out, void *stream)
{
    FV3HIP_REQUIRE(w_dtype == FV3HIP_F32 || w_dtype == FV3HIP_F64, "weights dtype must be F32 or F64");
    FV3HIP_REQUIRE(p_dtype == FV3HIP_F32 || p_dtype == FV3HIP_F64, "pressure dtype must be F32 or F64");
    FV3HIP_REQUIRE(w_repeat >= 1 && n_batch % w_repeat == 0, "bad w_repeat %lld", (long long)w_repeat);
    FV3HIP_REQUIRE(cmp_offset >= 0 && cmp_levels >= nz + cmp_offset, "p_cmp has %d levels, need at least nz + cmp_offset = %d", cmp_levels,
                   nz + cmp_offset);
    FV3HIP_REQUIRE(factor >= 1 && ny >= 0 && nx >= 0 && coarse_divides(ny, factor) && coarse_divides(nx, factor),
                   "extents (%d, %d) are not multiples of the factor %d (an odd extent is staggered: n - 1 must be)", ny, nx, factor);
    const int64_t n_inner = (int64_t)ny * nx;
    if (n_batch * nz * n_inner <= 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(weights && p_cmp_coarse && p_fine && out, "null pointer");
    if (n_batch * nz > 65535 || n_inner >= ((int64_t)1 << 31))
        return fail(FV3HIP_EUNSUPPORTED, "coarse-grid mask_weights needs n_batch * nz <= 65535 and fewer than 2^31 columns per plane");
    hipStream_t st = as_stream(stream);
    const int nyc = coarse_extent(ny, factor), nxc = coarse_extent(nx, factor);
    const int64_t plane2 = (int64_t)nyc * nxc;
    const dim3 grid((unsigned)ceil_div(n_inner, 256), (unsigned)(n_batch * nz));
    // four columns per thread where a row is whole quads of 16-byte aligned columns (a staggered x dim is not: 385 columns)
    const bool quads = nx % 4 == 0 && factor >= 4 && reinterpret_cast<uintptr_t>(weights) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(p_fine) % 32 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
    const dim3 grid4((unsigned)ceil_div(n_inner, 1024), (unsigned)(n_batch * nz));
#define LAUNCH_(TW, TP)                                                                                                            \
    if (quads)                                                                                                                     \
        hipLaunchKernelGGL((mask_weights_coarse4_kernel<TW, TP>), grid4, dim3(256), 0, st, static_cast<const TW *>(weights),      \
                           static_cast<const TP *>(p_cmp_coarse), static_cast<const TP *>(p_fine), static_cast<TW *>(out), nz, n_inner, nx, \
                           factor, nxc, plane2, w_repeat, cmp_levels, cmp_offset);                                                 \
    else                                                                                                                           \
        hipLaunchKernelGGL((mask_weights_coarse_kernel<TW, TP>), grid, dim3(256), 0, st, static_cast<const TW *>(weights),        \
                           static_cast<const TP *>(p_cmp_coarse), static_cast<const TP *>(p_fine), static_cast<TW *>(out), nz, n_inner, nx, factor, \
                           nxc, plane2, w_repeat, cmp_levels, cmp_offset)
    if (w_dtype == FV3HIP_F32 && p_dtype == FV3HIP_F32) LAUNCH_(float, float);
    else if (w_dtype == FV3HIP_F32) LAUNCH_(float, double);
    else if (p_dtype == FV3HIP_F32) LAUNCH_(double, float);
    else LAUNCH_(double, double);
#undef LAUNCH_
    return check_launch("mask_weights_coarse_kernel");
}

extern "C" size_t fv3hip_mappm_workspace_bytes(int64_t ncol, int km)
{
    if (ncol <= 0 || km <= 0) return 0;
    return kCounterBytes + ws_list_bytes(ncol) + (size_t)kMappmPlanes * (size_t)km * (size_t)ws_slots(ncol) * sizeof(float);
}

extern "C" int fv3hip_mappm(const void *pe1, const void *q1, const void *pe2, int in_dtype, float *q2,
                            int64_t n_batch, int64_t n_inner, int km, int kn, int iv, int kord,
                            int layout, int arith, void *workspace, size_t workspace_bytes, void *stream)
{
    FV3HIP_REQUIRE(in_dtype == FV3HIP_F32 || in_dtype == FV3HIP_F64, "in_dtype must be F32 or F64, got %d", in_dtype);
    FV3HIP_REQUIRE(arith == FV3HIP_ARITH_EXACT || arith == FV3HIP_ARITH_FAST, "unknown arithmetic mode %d", arith);
    FV3HIP_REQUIRE(layout == FV3HIP_LAYOUT_COL_LEVEL || layout == FV3HIP_LAYOUT_LEVEL_COL, "unknown layout %d", layout);
    FV3HIP_REQUIRE(n_batch >= 0 && n_inner >= 0 && kn >= 0, "negative extent");
    FV3HIP_REQUIRE(iv >= -2 && iv <= 2, "iv must be in [-2, 2], got %d", iv);
    if (kord > 7 && iv == -2)
        return fail(FV3HIP_EUNSUPPORTED, "kord=%d with iv=-2: cs_profile would read the array qs that mappm never sets (mappm.f90:34,51,152-176)", kord);
    FV3HIP_REQUIRE(km >= 4, "km must be >= 4 (ppm_profile reads a4(2,i,3)), got %d", km);
    if (layout == FV3HIP_LAYOUT_COL_LEVEL)
        FV3HIP_REQUIRE(n_inner == 1, "COL_LEVEL layout requires n_inner == 1");
    const int64_t ncol = n_batch * n_inner;
    if (ncol == 0 || kn == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(pe1 && q1 && pe2 && q2, "null pointer");
    FV3HIP_REQUIRE(workspace && workspace_bytes >= fv3hip_mappm_workspace_bytes(ncol, km),
                   "workspace too small: need %zu bytes, got %zu", fv3hip_mappm_workspace_bytes(ncol, km), workspace_bytes);
    hipStream_t st = as_stream(stream);
    const int64_t ws_cols = ws_slots(ncol);
    unsigned int *n_bad = static_cast<unsigned int *>(workspace);
    unsigned int *bad_cols = reinterpret_cast<unsigned int *>(static_cast<char *>(workspace) + kCounterBytes);
    float *planes = reinterpret_cast<float *>(static_cast<char *>(workspace) + kCounterBytes + ws_list_bytes(ncol));
    // kord <= 6: the register-window merge sweep, then the (normally empty) list of columns it gave
    // up on through the sequential routine; kord == 7 (Huynh's constraint needs a wider stencil)
    // goes through the sequential routine directly
    const bool fast = (kord <= 6);
    // the sweep kernel (remap.hip) where it applies: native layout, kord <= 3, whole waves per batch plane
    const bool sweep = mappm_sweep_eligible(n_inner, km, kn, kord, layout, in_dtype);
    for (int64_t col0 = 0; col0 < ncol; col0 += kMappmChunk) {
        const int64_t col_end = (col0 + kMappmChunk < ncol) ? col0 + kMappmChunk : ncol;
        const int64_t blocks = ceil_div(col_end - col0, 256);
        if (fast) {
            FV3HIP_CHECK_HIP(hipMemsetAsync(n_bad, 0, 16, st));
            const int64_t fb_threads = (col_end - col0 < kFallbackSlots) ? (col_end - col0) : kFallbackSlots;
            const int64_t fb_blocks = ceil_div(fb_threads, 256);
            if (sweep) {
                SweepArgs sa;
                memset(&sa, 0, sizeof(sa));
                sa.pe1 = pe1;
                sa.pe2 = pe2;
                sa.q1[0] = q1;
                sa.q2[0] = q2;
                sa.col0 = col0;
                sa.n_inner = n_inner;
                sa.km = km;
                sa.kn = kn;
                sa.iv = iv;
                sa.n_bad = n_bad;
                sa.bad_cols = bad_cols;
                mappm_sweep_launch(sa, 1, in_dtype, col_end, arith == FV3HIP_ARITH_FAST, st);
            }
#define LAUNCH_(T)                                                                                       \
    if (!sweep)                                                                                          \
        hipLaunchKernelGGL((mappm_merge_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st,            \
                           static_cast<const T *>(pe1), static_cast<const T *>(q1), static_cast<const T *>(pe2), \
                           q2, col0, col_end, n_inner, km, kn, iv, kord, layout, n_bad, bad_cols);       \
    hipLaunchKernelGGL((mappm_fallback_kernel<T>), dim3((unsigned)fb_blocks), dim3(256), 0, st,          \
                       static_cast<const T *>(pe1), static_cast<const T *>(q1), static_cast<const T *>(pe2), \
                       q2, col0, n_inner, km, kn, iv, kord, layout, n_bad, bad_cols, planes, ws_cols)
            if (in_dtype == FV3HIP_F32) { LAUNCH_(float); } else { LAUNCH_(double); }
#undef LAUNCH_
        } else {
#define LAUNCH_(T)                                                                                       \
    hipLaunchKernelGGL((mappm_simple_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st,               \
                       static_cast<const T *>(pe1), static_cast<const T *>(q1), static_cast<const T *>(pe2), \
                       q2, col0, col_end, n_inner, km, kn, iv, kord, layout, planes, ws_cols)
            if (in_dtype == FV3HIP_F32) { LAUNCH_(float); } else { LAUNCH_(double); }
#undef LAUNCH_
        }
        int rc = check_launch("mappm kernel");
        if (rc) return rc;
    }
    return FV3HIP_OK;
}

namespace {
template <typename T, int NF>
void launch_merge_multi(const void *pe1, const MultiFieldPtrs &fp, const void *pe2, int64_t col0, int64_t col_end, int64_t n_inner,
                        int km, int kn, int iv, int kord, int layout, unsigned int *n_bad, unsigned int *bad_cols, hipStream_t st)
{
    const int64_t blocks = ceil_div(col_end - col0, 256);
    hipLaunchKernelGGL((mappm_merge_multi_kernel<T, NF>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const T *>(pe1), fp,
                       static_cast<const T *>(pe2), col0, col_end, n_inner, km, kn, iv, kord, layout, n_bad, bad_cols);
}
}  // namespace

extern "C" int fv3hip_mappm_multi(const void *pe1, const void *const *q1, const void *pe2, int in_dtype, float *const *q2,
                                  int n_fields, int64_t n_batch, int64_t n_inner, int km, int kn, int iv, int kord, int layout,
                                  int arith, void *workspace, size_t workspace_bytes, void *stream)
{
    FV3HIP_REQUIRE(arith == FV3HIP_ARITH_EXACT || arith == FV3HIP_ARITH_FAST, "unknown arithmetic mode %d", arith);
    FV3HIP_REQUIRE(n_fields >= 0, "negative field count");
    if (n_fields == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(q1 && q2, "null pointer");
    if (kord > 6 || n_fields == 1) {  // kord >= 7 goes through the sequential routine
        for (int f = 0; f < n_fields; ++f) {
            const int rc = fv3hip_mappm(pe1, q1[f], pe2, in_dtype, q2[f], n_batch, n_inner, km, kn, iv, kord, layout, arith, workspace,
                                        workspace_bytes, stream);
            if (rc) return rc;
        }
        return FV3HIP_OK;
    }
    FV3HIP_REQUIRE(in_dtype == FV3HIP_F32 || in_dtype == FV3HIP_F64, "in_dtype must be F32 or F64, got %d", in_dtype);
    FV3HIP_REQUIRE(layout == FV3HIP_LAYOUT_COL_LEVEL || layout == FV3HIP_LAYOUT_LEVEL_COL, "unknown layout %d", layout);
    FV3HIP_REQUIRE(n_batch >= 0 && n_inner >= 0 && kn >= 0, "negative extent");
    FV3HIP_REQUIRE(iv >= -2 && iv <= 2, "iv must be in [-2, 2], got %d", iv);
    FV3HIP_REQUIRE(km >= 4, "km must be >= 4 (ppm_profile reads a4(2,i,3)), got %d", km);
    if (layout == FV3HIP_LAYOUT_COL_LEVEL) FV3HIP_REQUIRE(n_inner == 1, "COL_LEVEL layout requires n_inner == 1");
    const int64_t ncol = n_batch * n_inner;
    if (ncol == 0 || kn == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(pe1 && pe2, "null pointer");
    for (int f = 0; f < n_fields; ++f) FV3HIP_REQUIRE(q1[f] && q2[f], "null field pointer");
    FV3HIP_REQUIRE(workspace && workspace_bytes >= fv3hip_mappm_workspace_bytes(ncol, km),
                   "workspace too small: need %zu bytes, got %zu", fv3hip_mappm_workspace_bytes(ncol, km), workspace_bytes);
    hipStream_t st = as_stream(stream);
    const int64_t ws_cols = ws_slots(ncol);
    unsigned int *n_bad = static_cast<unsigned int *>(workspace);
    unsigned int *bad_cols = reinterpret_cast<unsigned int *>(static_cast<char *>(workspace) + kCounterBytes);
    float *planes = reinterpret_cast<float *>(static_cast<char *>(workspace) + kCounterBytes + ws_list_bytes(ncol));
    for (int64_t col0 = 0; col0 < ncol; col0 += kMappmChunk) {
        const int64_t col_end = (col0 + kMappmChunk < ncol) ? col0 + kMappmChunk : ncol;
        const int64_t fb_threads = (col_end - col0 < kFallbackSlots) ? (col_end - col0) : kFallbackSlots;
        const int64_t fb_blocks = ceil_div(fb_threads, 256);
        for (int f0 = 0; f0 < n_fields; f0 += kMaxMultiFields) {
            const int nf = (n_fields - f0 < kMaxMultiFields) ? n_fields - f0 : kMaxMultiFields;
            MultiFieldPtrs fp;
            for (int f = 0; f < kMaxMultiFields; ++f) {
                fp.q1[f] = f < nf ? q1[f0 + f] : nullptr;
                fp.q2[f] = f < nf ? q2[f0 + f] : nullptr;
            }
            FV3HIP_CHECK_HIP(hipMemsetAsync(n_bad, 0, 16, st));
            const bool sweep = mappm_sweep_eligible(n_inner, km, kn, kord, layout, in_dtype);
            if (sweep) {
                SweepArgs sa;
                memset(&sa, 0, sizeof(sa));
                sa.pe1 = pe1;
                sa.pe2 = pe2;
                for (int f = 0; f < nf; ++f) {
                    sa.q1[f] = fp.q1[f];
                    sa.q2[f] = fp.q2[f];
                }
                sa.col0 = col0;
                sa.n_inner = n_inner;
                sa.km = km;
                sa.kn = kn;
                sa.iv = iv;
                sa.n_bad = n_bad;
                sa.bad_cols = bad_cols;
                mappm_sweep_launch(sa, nf, in_dtype, col_end, arith == FV3HIP_ARITH_FAST, st);
            }
#define LAUNCH_(T)                                                                                                          \
    if (!sweep) switch (nf) {                                                                                               \
        case 1: launch_merge_multi<T, 1>(pe1, fp, pe2, col0, col_end, n_inner, km, kn, iv, kord, layout, n_bad, bad_cols, st); break; \
        case 2: launch_merge_multi<T, 2>(pe1, fp, pe2, col0, col_end, n_inner, km, kn, iv, kord, layout, n_bad, bad_cols, st); break; \
        case 3: launch_merge_multi<T, 3>(pe1, fp, pe2, col0, col_end, n_inner, km, kn, iv, kord, layout, n_bad, bad_cols, st); break; \
        default: launch_merge_multi<T, 4>(pe1, fp, pe2, col0, col_end, n_inner, km, kn, iv, kord, layout, n_bad, bad_cols, st); break; \
    }                                                                                                                       \
    for (int f = 0; f < nf; ++f)                                                                                            \
        hipLaunchKernelGGL((mappm_fallback_kernel<T>), dim3((unsigned)fb_blocks), dim3(256), 0, st, static_cast<const T *>(pe1), \
                           static_cast<const T *>(fp.q1[f]), static_cast<const T *>(pe2), fp.q2[f], col0, n_inner, km, kn, iv,    \
                           kord, layout, n_bad, bad_cols, planes, ws_cols)
            if (in_dtype == FV3HIP_F32) { LAUNCH_(float); } else { LAUNCH_(double); }
#undef LAUNCH_
            const int rc = check_launch("mappm multi-field kernel");
            if (rc) return rc;
        }
    }
    return FV3HIP_OK;
}

// mappm_multi with the target interfaces on a horizontally coarser grid: pe2 is [n_batch][kn + 1][nyc][nxc] and every fine column
// (y, x) is remapped to the interfaces of coarse column (y / factor, x / factor) -- what regridz.py:119-185 does through an
// upsampled copy of the coarse pressures.  LEVEL_COL layout ([n_batch][level][ny][nx]); the sweep kernel only:
// FV3HIP_EUNSUPPORTED where it does not apply (the caller then upsamples and calls fv3hip_mappm_multi).
extern "C" int fv3hip_mappm_multi_coarse_target(const void *pe1, const void *const *q1, const void *pe2_coarse, int in_dtype, float *const *q2,
                                                int n_fields, int64_t n_batch, int ny, int nx, int factor, int km, int kn, int iv, int kord,
                                                int arith, void *workspace, size_t workspace_bytes, void *stream)
{
    FV3HIP_REQUIRE(arith == FV3HIP_ARITH_EXACT || arith == FV3HIP_ARITH_FAST, "unknown arithmetic mode %d", arith);
    FV3HIP_REQUIRE(n_fields >= 0, "negative field count");
    if (n_fields == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(q1 && q2, "null pointer");
    FV3HIP_REQUIRE(in_dtype == FV3HIP_F32 || in_dtype == FV3HIP_F64, "in_dtype must be F32 or F64, got %d", in_dtype);
    FV3HIP_REQUIRE(n_batch >= 0 && ny >= 0 && nx >= 0 && kn >= 0, "negative extent");
    FV3HIP_REQUIRE(factor >= 1 && coarse_divides(ny, factor) && coarse_divides(nx, factor),
                   "extents (%d, %d) are not multiples of the factor %d (an odd extent is staggered: n - 1 must be)", ny, nx, factor);
    FV3HIP_REQUIRE(iv >= -2 && iv <= 2, "iv must be in [-2, 2], got %d", iv);
    FV3HIP_REQUIRE(km >= 4, "km must be >= 4 (ppm_profile reads a4(2,i,3)), got %d", km);
    const int64_t n_inner = (int64_t)ny * nx, ncol = n_batch * n_inner;
    if (ncol == 0 || kn == 0) return FV3HIP_OK;
    const int nyc = coarse_extent(ny, factor), nxc = coarse_extent(nx, factor);
    const int64_t plane2 = (int64_t)nyc * nxc;
    if (factor < 2 || !mappm_sweep_eligible(n_inner, km, kn, kord, FV3HIP_LAYOUT_LEVEL_COL, in_dtype, plane2))
        return fail(FV3HIP_EUNSUPPORTED, "coarse-target remap needs factor >= 2 and a shape the sweep kernel takes");
    FV3HIP_REQUIRE(pe1 && pe2_coarse, "null pointer");
    for (int f = 0; f < n_fields; ++f) FV3HIP_REQUIRE(q1[f] && q2[f], "null field pointer");
    FV3HIP_REQUIRE(workspace && workspace_bytes >= fv3hip_mappm_workspace_bytes(ncol, km),
                   "workspace too small: need %zu bytes, got %zu", fv3hip_mappm_workspace_bytes(ncol, km), workspace_bytes);
    hipStream_t st = as_stream(stream);
    const int64_t ws_cols = ws_slots(ncol);
    unsigned int *n_bad = static_cast<unsigned int *>(workspace);
    unsigned int *bad_cols = reinterpret_cast<unsigned int *>(static_cast<char *>(workspace) + kCounterBytes);
    float *planes = reinterpret_cast<float *>(static_cast<char *>(workspace) + kCounterBytes + ws_list_bytes(ncol));
    for (int64_t col0 = 0; col0 < ncol; col0 += kMappmChunk) {
        const int64_t col_end = (col0 + kMappmChunk < ncol) ? col0 + kMappmChunk : ncol;
        const int64_t fb_threads = (col_end - col0 < kFallbackSlots) ? (col_end - col0) : kFallbackSlots;
        const int64_t fb_blocks = ceil_div(fb_threads, 256);
        for (int f0 = 0; f0 < n_fields; f0 += kMaxMultiFields) {
            const int nf = (n_fields - f0 < kMaxMultiFields) ? n_fields - f0 : kMaxMultiFields;
            FV3HIP_CHECK_HIP(hipMemsetAsync(n_bad, 0, 16, st));
            SweepArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.pe1 = pe1;
            sa.pe2 = pe2_coarse;
            for (int f = 0; f < nf; ++f) {
                sa.q1[f] = q1[f0 + f];
                sa.q2[f] = q2[f0 + f];
            }
            sa.col0 = col0;
            sa.n_inner = n_inner;
            sa.km = km;
            sa.kn = kn;
            sa.iv = iv;
            sa.n_bad = n_bad;
            sa.bad_cols = bad_cols;
            sa.pe2_f = factor;
            sa.nx = nx;
            sa.pe2_nx = nxc;
            sa.pe2_plane = plane2;
            mappm_sweep_launch(sa, nf, in_dtype, col_end, arith == FV3HIP_ARITH_FAST, st);
#define LAUNCH_(T)                                                                                                                  \
    for (int f = 0; f < nf; ++f)                                                                                                    \
        hipLaunchKernelGGL((mappm_fallback_kernel<T>), dim3((unsigned)fb_blocks), dim3(256), 0, st, static_cast<const T *>(pe1),    \
                           static_cast<const T *>(q1[f0 + f]), static_cast<const T *>(pe2_coarse), q2[f0 + f], col0, n_inner, km, kn, iv, \
                           kord, (int)FV3HIP_LAYOUT_LEVEL_COL, n_bad, bad_cols, planes, ws_cols, factor, nx, nxc, plane2)
            if (in_dtype == FV3HIP_F32) { LAUNCH_(float); } else { LAUNCH_(double); }
#undef LAUNCH_
            const int rc = check_launch("mappm coarse-target kernels");
            if (rc) return rc;
        }
    }
    return FV3HIP_OK;
}

extern "C" size_t fv3hip_mappm_block_mean_workspace_bytes(int64_t ncol, int km)
{
    if (ncol <= 0 || km <= 0) return 0;
    return fv3hip_mappm_workspace_bytes(ncol, km) + (size_t)(kMappmChunk / 64) * 3 * sizeof(unsigned int);  // block lists: redo, (rest, row)
}

extern "C" int fv3hip_mappm_block_mean(const void *pe1, const void *const *q1, const void *pe2_coarse, const void *level_coarse,
                                       int cmp_levels, int cmp_offset, int in_dtype, const float *area, int64_t area_repeat,
                                       float *const *scratch, float *const *mean, int n_fields, int64_t n_batch, int ny, int nx, int factor,
                                       int km, int kn, int iv, int kord, int arith, void *workspace, size_t workspace_bytes, void *stream)
{
    FV3HIP_REQUIRE(arith == FV3HIP_ARITH_EXACT || arith == FV3HIP_ARITH_FAST, "unknown arithmetic mode %d", arith);
    FV3HIP_REQUIRE(n_fields >= 0, "negative field count");
    if (n_fields == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(q1 && scratch && mean, "null pointer");
    FV3HIP_REQUIRE(in_dtype == FV3HIP_F32 || in_dtype == FV3HIP_F64, "in_dtype must be F32 or F64, got %d", in_dtype);
    FV3HIP_REQUIRE(n_batch >= 0 && ny >= 0 && nx >= 0 && kn >= 0, "negative extent");
    FV3HIP_REQUIRE(iv >= -2 && iv <= 2, "iv must be in [-2, 2], got %d", iv);
    FV3HIP_REQUIRE(km >= 4, "km must be >= 4 (ppm_profile reads a4(2,i,3)), got %d", km);
    FV3HIP_REQUIRE(area_repeat >= 1, "area_repeat must be >= 1");
    FV3HIP_REQUIRE(cmp_offset >= 0 && cmp_levels >= kn + cmp_offset, "compared levels: %d rows cannot serve %d layers at offset %d",
                   cmp_levels, kn, cmp_offset);
    const int64_t n_inner = (int64_t)ny * nx, ncol = n_batch * n_inner;
    if (ncol == 0 || kn == 0) return FV3HIP_OK;
    if (!mappm_mean_eligible(ny, nx, factor, km, kn, kord, in_dtype) || ncol >= ((int64_t)1 << 32))
        return fail(FV3HIP_EUNSUPPORTED, "the fused remap + block mean needs factor 8, extents that are multiples of 8, kord <= 3, km >= 8, kn < 128");
    const int nyc = ny / factor, nxc = nx / factor;
    const int64_t plane2 = (int64_t)nyc * nxc;
    FV3HIP_REQUIRE(pe1 && pe2_coarse && level_coarse && area, "null pointer");
    for (int f = 0; f < n_fields; ++f) FV3HIP_REQUIRE(q1[f] && scratch[f] && mean[f], "null field pointer");
    FV3HIP_REQUIRE(workspace && workspace_bytes >= fv3hip_mappm_block_mean_workspace_bytes(ncol, km),
                   "workspace too small: need %zu bytes, got %zu", fv3hip_mappm_block_mean_workspace_bytes(ncol, km), workspace_bytes);
    hipStream_t st = as_stream(stream);
    const int64_t ws_cols = ws_slots(ncol);
    unsigned int *n_bad = static_cast<unsigned int *>(workspace);
    unsigned int *bad_cols = reinterpret_cast<unsigned int *>(static_cast<char *>(workspace) + kCounterBytes);
    float *planes = reinterpret_cast<float *>(static_cast<char *>(workspace) + kCounterBytes + ws_list_bytes(ncol));
    unsigned int *bad_blocks = reinterpret_cast<unsigned int *>(static_cast<char *>(workspace) + fv3hip_mappm_workspace_bytes(ncol, km));
    unsigned int *rest_blocks = bad_blocks + kMappmChunk / 64;
    // a launch = kMappmChunk columns' worth of blocks (the lists are sized for that); `col0` counts 64 columns per block
    for (int64_t col0 = 0; col0 < ncol; col0 += kMappmChunk) {
        const int64_t col_end = (col0 + kMappmChunk < ncol) ? col0 + kMappmChunk : ncol;
        const int64_t fb_threads = (col_end - col0 < kFallbackSlots) ? (col_end - col0) : kFallbackSlots;
        const int64_t fb_blocks = ceil_div(fb_threads, 256);
        for (int f0 = 0; f0 < n_fields; f0 += kMaxMultiFields) {
            const int nf = (n_fields - f0 < kMaxMultiFields) ? n_fields - f0 : kMaxMultiFields;
            FV3HIP_CHECK_HIP(hipMemsetAsync(n_bad, 0, 16, st));
            SweepArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.pe1 = pe1;
            sa.pe2 = pe2_coarse;
            for (int f = 0; f < nf; ++f) {
                sa.q1[f] = q1[f0 + f];
                sa.q2[f] = scratch[f0 + f];
                sa.mean[f] = mean[f0 + f];
            }
            sa.col0 = col0;
            sa.n_inner = n_inner;
            sa.km = km;
            sa.kn = kn;
            sa.iv = iv;
            sa.n_bad = n_bad;
            sa.bad_cols = bad_cols;
            sa.bad_blocks = bad_blocks;
            sa.rest_blocks = rest_blocks;
            sa.pe2_f = factor;
            sa.nx = nx;
            sa.pe2_nx = nxc;
            sa.pe2_plane = plane2;
            sa.area = area;
            sa.area_repeat = area_repeat;
            sa.lvl = level_coarse;
            sa.cmp_levels = cmp_levels;
            sa.cmp_offset = cmp_offset;
            mappm_mean_launch(sa, nf, in_dtype, col_end, arith == FV3HIP_ARITH_FAST, st);
            mappm_mean_rest_launch(sa, nf, in_dtype, (col_end - col0) / 64, st);
            // blocks with an ill-formed column (normally none): all their columns through the sequential routine into the
            // scratch rows (absolute column indices in the list: col0 = 0), then their means from there
#define LAUNCH_(T)                                                                                                                  \
    for (int f = 0; f < nf; ++f)                                                                                                    \
        hipLaunchKernelGGL((mappm_fallback_kernel<T>), dim3((unsigned)fb_blocks), dim3(256), 0, st, static_cast<const T *>(pe1),    \
                           static_cast<const T *>(q1[f0 + f]), static_cast<const T *>(pe2_coarse), scratch[f0 + f], (int64_t)0, n_inner, km, \
                           kn, iv, kord, (int)FV3HIP_LAYOUT_LEVEL_COL, n_bad, bad_cols, planes, ws_cols, factor, nx, nxc, plane2)
            if (in_dtype == FV3HIP_F32) { LAUNCH_(float); } else { LAUNCH_(double); }
#undef LAUNCH_
            mappm_mean_redo_launch(sa, nf, in_dtype, st);
            const int rc = check_launch("mappm block-mean kernels");
            if (rc) return rc;
        }
    }
    return FV3HIP_OK;
}

extern "C" int fv3hip_interpolate_2d(const void *xp, const void *x, const void *y, int64_t n_batch, int64_t n_inner, int n_in,
                                     int n_out, double fill_value, int layout, void *out, void *stream)
{
    FV3HIP_REQUIRE(layout == FV3HIP_LAYOUT_COL_LEVEL || layout == FV3HIP_LAYOUT_LEVEL_COL, "unknown layout %d", layout);
    FV3HIP_REQUIRE(n_batch >= 0 && n_inner >= 0 && n_in >= 1 && n_out >= 0, "bad extents");
    const int64_t total = n_batch * n_inner * n_out;
    if (total == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(xp && x && y && out, "null pointer");
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(interpolate_2d_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       static_cast<const double *>(xp), static_cast<const double *>(x), static_cast<const double *>(y),
                       static_cast<double *>(out), fill_value, n_batch, n_inner, n_in, n_out, layout);
    return check_launch("interpolate_2d_kernel");
}

namespace {
inline unsigned col_grid(int64_t n)
{
    int64_t b = ceil_div(n, 256);
    return (unsigned)(b > 256 * 64 ? 256 * 64 : (b < 1 ? 1 : b));
}
}  // namespace

extern "C" int fv3hip_column_sum(const void *x, int dtype, int64_t n_batch, int nz, int64_t n_inner, double addend, void *out,
                                 void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_batch >= 0 && nz >= 0 && n_inner >= 0, "negative extent");
    if (n_batch * n_inner == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(x && out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((column_sum_kernel<double>), dim3(col_grid(n_batch * n_inner)), dim3(256), 0, st,
                           static_cast<const double *>(x), static_cast<double *>(out), n_batch, nz, n_inner, addend);
    else
        hipLaunchKernelGGL((column_sum_kernel<float>), dim3(col_grid(n_batch * n_inner)), dim3(256), 0, st,
                           static_cast<const float *>(x), static_cast<float *>(out), n_batch, nz, n_inner, (float)addend);
    return check_launch("column_sum_kernel");
}

extern "C" int fv3hip_blend_weights(const void *blending_pressure, const void *ps_coarse, const void *pfull_coarse, int dtype,
                                    int64_t n_batch, int nz, int64_t n_inner, void *out, void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_batch >= 0 && nz >= 0 && n_inner >= 0, "negative extent");
    const int64_t total = n_batch * nz * n_inner;
    if (total == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(blending_pressure && ps_coarse && pfull_coarse && out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((blend_weights_kernel<double>), dim3(col_grid(total)), dim3(256), 0, st,
                           static_cast<const double *>(blending_pressure), static_cast<const double *>(ps_coarse),
                           static_cast<const double *>(pfull_coarse), static_cast<double *>(out), n_batch, nz, n_inner);
    else
        hipLaunchKernelGGL((blend_weights_kernel<float>), dim3(col_grid(total)), dim3(256), 0, st,
                           static_cast<const float *>(blending_pressure), static_cast<const float *>(ps_coarse),
                           static_cast<const float *>(pfull_coarse), static_cast<float *>(out), n_batch, nz, n_inner);
    return check_launch("blend_weights_kernel");
}

extern "C" int fv3hip_hydrostatic_balance(const void *dz, const void *phis, const void *t, const void *q, const void *delp,
                                          int dtype, int64_t n_batch, int nz, int64_t n_inner, double toa_pressure,
                                          void *dz_out, void *phis_out, void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_batch >= 0 && nz >= 0 && n_inner >= 0, "negative extent");
    if (n_batch * n_inner == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(dz && phis && t && q && delp && dz_out && phis_out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((hydrostatic_kernel<double>), dim3(col_grid(n_batch * n_inner)), dim3(256), 0, st,
                           static_cast<const double *>(dz), static_cast<const double *>(phis), static_cast<const double *>(t),
                           static_cast<const double *>(q), static_cast<const double *>(delp), static_cast<double *>(dz_out),
                           static_cast<double *>(phis_out), n_batch, nz, n_inner, toa_pressure);
    else
        hipLaunchKernelGGL((hydrostatic_kernel<float>), dim3(col_grid(n_batch * n_inner)), dim3(256), 0, st,
                           static_cast<const float *>(dz), static_cast<const float *>(phis), static_cast<const float *>(t),
                           static_cast<const float *>(q), static_cast<const float *>(delp), static_cast<float *>(dz_out),
                           static_cast<float *>(phis_out), n_batch, nz, n_inner, (float)toa_pressure);
    return check_launch("hydrostatic_kernel");
}
