// Zhao-Carr emulator post-processing on the device: masks and conservation fixes applied to the
// emulator's outputs right after the network, on the same [feature(z), sample] arrays.
//
// Replaces (numpy / numba on the host in the reference):
//   external/emulation/emulation/masks.py:23-76          RangeMask, LevelMask
//   external/emulation/emulation/zhao_carr.py:60-86      squash_*, infer_gscond_cloud_from_conservation
//   external/emulation/emulation/zhao_carr.py:97-246     net-condensation limit, apply_condensation
//                                                        (liquid / phase dependent via ice_water_flag),
//                                                        the gscond masks that choose the cloud first
//   external/emulation/emulation/zhao_carr.py:249-344    strict TOA-to-surface precipitation scan,
//                                                        enforce_conservative_precpd, conservative_precip_simple
//
// All arrays are contiguous [n0][n1] (n0 = levels, n1 = samples for the Fortran hook's state).  Every
// array comes with its dtype code; arithmetic is done in T = float64 if any operand is float64 (the
// numpy promotion of the reference: Fortran state is float64), else float32.  HBM-bound, one pass.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/fv3hip.h"
#include "common.h"

namespace fv3hip {
namespace {

template <typename T>
__device__ __forceinline__ T ld(const void *p, int dt, int64_t i)
{
    return dt == FV3HIP_F64 ? (T) static_cast<const double *>(p)[i] : (T) static_cast<const float *>(p)[i];
}
template <typename T>
__device__ __forceinline__ void st(void *p, int64_t i, T v)
{
    static_cast<T *>(p)[i] = v;
}

constexpr double kCp = 1.0046e3, kLv = 2.5e6, kHfus = 3.3358e5, kGravity = 9.80665, kRhoWater = 1000.0;

template <typename T>
__global__ void squash_kernel(const void *cloud, int cdt, const void *hum, int hdt, int64_t n, T bound, void *cloud_out,
                              int cloud_out_dt, T *qv_out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T c = ld<T>(cloud, cdt, i);
        const T co = (c < bound) ? (T)0 : c;
        // (the squashed cloud keeps the cloud's own dtype, as np.where(cloud < bound, 0, cloud) does)
        if (cloud_out_dt == FV3HIP_F64) static_cast<double *>(cloud_out)[i] = (double)co;
        else static_cast<float *>(cloud_out)[i] = (float)co;
        qv_out[i] = ld<T>(hum, hdt, i) + (c - co);
    }
}

template <typename T>
__global__ void infer_cloud_kernel(const void *cloud_in, const void *qv_in, int sdt, const void *qv_emul, int edt, int64_t n,
                                   T *cloud_out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        cloud_out[i] = ld<T>(cloud_in, sdt, i) - (ld<T>(qv_emul, edt, i) - ld<T>(qv_in, sdt, i));
}

// which cloud goes into the conservation step (zhao_carr.py:180-246)
template <typename T>
__device__ __forceinline__ T choose_cloud(int mode, T c_emul, T c_in, const void *aux, int adt, int64_t i, int64_t n,
                                          int n_class, int cls)
{
    switch (mode) {
        case FV3HIP_ZC_FORTRAN_VANISHES: return ld<T>(aux, adt, i) < (T)1e-15 ? (T)0 : c_emul;
        case FV3HIP_ZC_FORTRAN_IDENTICAL: return ld<T>(aux, adt, i) == c_in ? c_in : c_emul;
        case FV3HIP_ZC_CLASS_ZERO_CLOUD:
        case FV3HIP_ZC_CLASS_ZERO_TEND: {
            // one-hot by arg-max with ties all hot: logit[cls] == max over classes (zhao_carr.py:193-198)
            T mx = ld<T>(aux, adt, i);
            for (int c = 1; c < n_class; ++c) {
                const T v = ld<T>(aux, adt, (int64_t)c * n + i);
                mx = v > mx ? v : mx;
            }
            const bool hot = ld<T>(aux, adt, (int64_t)cls * n + i) == mx;
            return hot ? (mode == FV3HIP_ZC_CLASS_ZERO_CLOUD ? (T)0 : c_in) : c_emul;
        }
        default: return c_emul;
    }
}

template <typename T>
__device__ __forceinline__ void conserve_one(T c_in, T qv_in, T t_in, T cloud_choice, T lv, int64_t i, T *cloud_out,
                                             T *qv_out, T *t_out)
{
    T net = cloud_choice - c_in;
    const T cond = net > (T)0 ? net : (T)0, evap = net < (T)0 ? net : (T)0;
    const T le = evap > -c_in ? evap : -c_in;   // np.maximum(evaporation, -available_liquid)
    const T lc = cond < qv_in ? cond : qv_in;   // np.minimum(condensation, available_vapor)
    net = le + lc;
    cloud_out[i] = c_in + net;
    qv_out[i] = qv_in - net;
    t_out[i] = t_in + lv * net / (T)kCp;
}

// liquid-phase latent heat: purely elementwise
template <typename T>
__global__ void gscond_conserve_kernel(const void *cloud_in, const void *qv_in, const void *t_in, int sdt,
                                       const void *cloud_emul, int edt, int mode, const void *aux, int adt, int n_class,
                                       int cls, int64_t n, T *cloud_out, T *qv_out, T *t_out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T c_in = ld<T>(cloud_in, sdt, i);
        const T choice = choose_cloud<T>(mode, ld<T>(cloud_emul, edt, i), c_in, aux, adt, i, n, n_class, cls);
        conserve_one<T>(c_in, ld<T>(qv_in, sdt, i), ld<T>(t_in, sdt, i), choice, (T)kLv, i, cloud_out, qv_out, t_out);
    }
}

// Phase-dependent latent heat: the ice/water flag is a scan along the LAST axis from its end
// (zhao_carr.py:114-138; see the quirk noted in oracle/emulation_np.py):
//   t < -15 -> 1;  t > 0 -> 0;  otherwise 1 iff the previous (higher index) flag is 1 and cloud > 1e-20.
// Each element is a map {0,1} -> {0,1} (constant 1, constant 0, or identity); maps compose
// associatively, so a row is cut into one contiguous segment per thread: pass 1 composes each
// segment's map, thread 0 chains the 256 segment maps, pass 2 replays each segment with its
// incoming flag and applies the conservation fix.  One workgroup per row.
template <typename T>
__global__ __launch_bounds__(256) void gscond_conserve_phase_kernel(const void *cloud_in, const void *qv_in, const void *t_in,
                                                                    int sdt, const void *cloud_emul, int edt, int mode,
                                                                    const void *aux, int adt, int n_class, int cls,
                                                                    int64_t n0, int64_t n1, T *cloud_out, T *qv_out,
                                                                    T *t_out)
{
    __shared__ unsigned char seg_map[256];   // bit 0: f(0), bit 1: f(1)
    __shared__ unsigned char seg_in[256];    // incoming flag of each segment
    const int64_t row = blockIdx.x;
    const int64_t seg = (n1 + 255) / 256;
    // thread 0 owns the END of the row (the scan starts there)
    const int64_t hi = n1 - (int64_t)threadIdx.x * seg;            // exclusive
    const int64_t lo = hi - seg > 0 ? hi - seg : 0;
    const int64_t n = n0 * n1;
    auto elem_map = [&](int64_t k) -> unsigned {
        const int64_t i = row * n1 + k;
        const T tc = ld<T>(t_in, sdt, i) - (T)273.16;
        if (tc < (T)-15) return 3u;               // constant 1
        if (tc > (T)0) return 0u;                 // constant 0
        return ld<T>(cloud_in, sdt, i) > (T)1e-20 ? 2u : 0u;  // identity : constant 0
    };
    unsigned m = 2u;  // identity
    for (int64_t k = hi - 1; k >= lo && hi > 0; --k) {
        const unsigned e = elem_map(k);
        // new map = e o m : x -> e(m(x))
        const unsigned m0 = m & 1u, m1 = (m >> 1) & 1u;
        m = ((e >> m0) & 1u) | (((e >> m1) & 1u) << 1);
    }
    seg_map[threadIdx.x] = (unsigned char)m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned state = 0;  // beyond the end of the row there is no ice
        for (int j = 0; j < 256; ++j) {
            seg_in[j] = (unsigned char)state;
            state = (seg_map[j] >> state) & 1u;
        }
    }
    __syncthreads();
    unsigned state = seg_in[threadIdx.x];
    for (int64_t k = hi - 1; k >= lo && hi > 0; --k) {
        state = (elem_map(k) >> state) & 1u;
        const int64_t i = row * n1 + k;
        const T c_in = ld<T>(cloud_in, sdt, i);
        const T choice = choose_cloud<T>(mode, ld<T>(cloud_emul, edt, i), c_in, aux, adt, i, n, n_class, cls);
        const T lv = (T)kLv + (T)state * (T)kHfus;
        conserve_one<T>(c_in, ld<T>(qv_in, sdt, i), ld<T>(t_in, sdt, i), choice, lv, i, cloud_out, qv_out, t_out);
    }
}

// strict precipitation budget, one thread per sample, levels from the last (TOA) to the first
template <typename T>
__global__ void precpd_conserve_kernel(const void *cloud_g, const void *qv_g, const void *t_g, const void *delp, int sdt,
                                       const void *cloud_p, const void *qv_p, int edt, int64_t n0, int64_t n1, T *cloud_out,
                                       T *qv_out, T *t_out, T *precip_out)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n1; s += (int64_t)gridDim.x * blockDim.x) {
        T total = 0;
        for (int64_t k = n0 - 1; k >= 0; --k) {
            const int64_t i = k * n1 + s;
            const T dp = ld<T>(delp, sdt, i), cg = ld<T>(cloud_g, sdt, i), qg = ld<T>(qv_g, sdt, i);
            T src = (T)-1 * (ld<T>(cloud_p, edt, i) - cg) * dp / (T)kGravity;
            T sink = (ld<T>(qv_p, edt, i) - qg) * dp / (T)kGravity;
            src = src > (T)0 ? src : (T)0;
            sink = sink > (T)0 ? sink : (T)0;
            total = total + src;
            const T ev = total < sink ? total : sink;
            total = total - ev;
            const T evap = ev / dp * (T)kGravity;
            cloud_out[i] = cg + ((T)-1 * src) / dp * (T)kGravity;
            qv_out[i] = qg + evap;
            t_out[i] = ld<T>(t_g, sdt, i) + (T)(kLv / kCp) * (T)-1 * evap;
        }
        precip_out[s] = total / (T)kRhoWater;
    }
}

template <typename T>
__global__ void precip_simple_kernel(const void *cloud_g, const void *qv_g, const void *delp, int sdt, const void *cloud_p,
                                     const void *qv_p, int edt, int64_t n0, int64_t n1, T *precip_out)
{
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n1; s += (int64_t)gridDim.x * blockDim.x) {
        T before = 0, after = 0;
        for (int64_t k = 0; k < n0; ++k) {  // np.sum over axis 0: sequential in k
            const int64_t i = k * n1 + s;
            const T dp = ld<T>(delp, sdt, i);
            // (qv + qc is formed in the arrays' own dtype before it meets delp, as numpy does:
            // float32 emulator outputs are added in float32)
            const T wb = sdt == FV3HIP_F64 ? (T)(ld<double>(qv_g, sdt, i) + ld<double>(cloud_g, sdt, i))
                                           : (T)(ld<float>(qv_g, sdt, i) + ld<float>(cloud_g, sdt, i));
            const T wa = edt == FV3HIP_F64 ? (T)(ld<double>(qv_p, edt, i) + ld<double>(cloud_p, edt, i))
                                           : (T)(ld<float>(qv_p, edt, i) + ld<float>(cloud_p, edt, i));
            before += wb * dp / (T)kGravity;
            after += wa * dp / (T)kGravity;
        }
        precip_out[s] = (before - after) / (T)kRhoWater;
    }
}

template <typename T>
__global__ void clamp_kernel(const T *x, int64_t n, T lo, T hi, int has_lo, int has_hi, T *out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T v = x[i];
        // np.maximum / np.minimum propagate NaN
        if (has_lo) v = (v != v) ? v : (v > lo ? v : lo);
        if (has_hi) v = (v != v) ? v : (v < hi ? v : hi);
        out[i] = v;
    }
}

__global__ void level_fill_kernel(const void *emul, int edt, const void *src, int sdt, double fill, int64_t n0, int64_t n1,
                                  int64_t start, int64_t stop, double *out)
{
    const int64_t n = n0 * n1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i / n1;
        if (k >= start && k < stop)
            out[i] = src ? ld<double>(src, sdt, i) : fill;
        else
            out[i] = ld<double>(emul, edt, i);
    }
}

template <typename T>
__global__ void class_zero_kernel(const T *x, const void *logits, int ldt, int n_class, int cls, int64_t n, T *out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double mx = ld<double>(logits, ldt, i);
        for (int c = 1; c < n_class; ++c) {
            const double v = ld<double>(logits, ldt, (int64_t)c * n + i);
            mx = v > mx ? v : mx;
        }
        out[i] = (ld<double>(logits, ldt, (int64_t)cls * n + i) == mx) ? (T)0 : x[i];
    }
}

// ---------------------------------------------------------------------------------------
// Humidity limiters applied to the ML tendencies each timestep
// (external/vcm/vcm/calc/thermo/non_negative_sphum.py:6-45, local.py:25-28,317-360)
// ---------------------------------------------------------------------------------------
constexpr double kHeatCapacity = 1004.0 - 287.05;  // _SPECIFIC_HEAT_CONST_PRESSURE - _RDGAS
constexpr double kLvFreezing = 2.5e6;              // latent_heat_vaporization(_FREEZING_TEMPERATURE)

template <typename T>
__global__ void non_negative_sphum_kernel(const T *sphum, const T *q1, const T *q2, int64_t n, T dt, int mse_conserving,
                                          T *q1_out, T *q2_out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T s = sphum[i], b = q2[i];
        if (!mse_conserving) {
            // non_negative_sphum: scale both tendencies by -sphum / (dt * dQ2) where the humidity would go negative
            const T ratio = (-s) / (dt * b);
            const bool ok = s + b * dt >= (T)0;
            q2_out[i] = ok ? b : ratio * b;
            if (q1) q1_out[i] = ok ? q1[i] : ratio * q1[i];
        } else {
            // limit the moistening tendency, then re-derive the heating that keeps the MSE tendency
            const T b_new = (s + b * dt >= (T)0) ? b : -s / dt;
            q2_out[i] = b_new;
            if (q1) {
                const T mse = (T)kHeatCapacity * q1[i] + (T)kLvFreezing * b;
                q1_out[i] = (mse - (T)kLvFreezing * b_new) / (T)kHeatCapacity;
            }
        }
    }
}

inline unsigned grid_for(int64_t n)
{
    int64_t b = ceil_div(n, 256);
    return (unsigned)(b > 256 * 64 ? 256 * 64 : (b < 1 ? 1 : b));
}
inline bool float_code(int dt) { return dt == FV3HIP_F32 || dt == FV3HIP_F64; }

}  // namespace
}  // namespace fv3hip

using namespace fv3hip;

#define ZC_COMMON_CHECKS(n)                                                                        \
    FV3HIP_REQUIRE(out_dtype == FV3HIP_F32 || out_dtype == FV3HIP_F64, "out_dtype must be F32 or F64"); \
    FV3HIP_REQUIRE((n) >= 0, "negative extent");                                                   \
    if ((n) == 0) return FV3HIP_OK

extern "C" int fv3hip_zc_squash(const void *cloud, int cloud_dtype, const void *humidity, int hum_dtype, int64_t n,
                                double bound, int out_dtype, void *cloud_out, void *qv_out, void *stream)
{
    ZC_COMMON_CHECKS(n);
    FV3HIP_REQUIRE(float_code(cloud_dtype) && float_code(hum_dtype), "arrays must be F32 or F64");
    FV3HIP_REQUIRE(cloud && humidity && cloud_out && qv_out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (out_dtype == FV3HIP_F64)
        hipLaunchKernelGGL((squash_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, cloud, cloud_dtype, humidity,
                           hum_dtype, n, bound, cloud_out, cloud_dtype, static_cast<double *>(qv_out));
    else
        hipLaunchKernelGGL((squash_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, cloud, cloud_dtype, humidity,
                           hum_dtype, n, (float)bound, cloud_out, cloud_dtype, static_cast<float *>(qv_out));
    return check_launch("squash_kernel");
}

extern "C" int fv3hip_zc_infer_cloud(const void *cloud_in, const void *qv_in, int state_dtype, const void *qv_emul,
                                     int emul_dtype, int64_t n, int out_dtype, void *cloud_out, void *stream)
{
    ZC_COMMON_CHECKS(n);
    FV3HIP_REQUIRE(float_code(state_dtype) && float_code(emul_dtype), "arrays must be F32 or F64");
    FV3HIP_REQUIRE(cloud_in && qv_in && qv_emul && cloud_out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (out_dtype == FV3HIP_F64)
        hipLaunchKernelGGL((infer_cloud_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, cloud_in, qv_in, state_dtype,
                           qv_emul, emul_dtype, n, static_cast<double *>(cloud_out));
    else
        hipLaunchKernelGGL((infer_cloud_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, cloud_in, qv_in, state_dtype,
                           qv_emul, emul_dtype, n, static_cast<float *>(cloud_out));
    return check_launch("infer_cloud_kernel");
}

extern "C" int fv3hip_zc_gscond_conserve(const void *cloud_in, const void *qv_in, const void *t_in, int state_dtype,
                                         const void *cloud_emul, int emul_dtype, int mode, const void *aux, int aux_dtype,
                                         int n_class, int cls, int64_t n0, int64_t n1, int phase_dependent, int out_dtype,
                                         void *cloud_out, void *qv_out, void *t_out, void *stream)
{
    const int64_t n = n0 * n1;
    FV3HIP_REQUIRE(n0 >= 0 && n1 >= 0, "negative extent");
    ZC_COMMON_CHECKS(n);
    FV3HIP_REQUIRE(float_code(state_dtype) && float_code(emul_dtype), "arrays must be F32 or F64");
    FV3HIP_REQUIRE(mode >= FV3HIP_ZC_NO_MASK && mode <= FV3HIP_ZC_CLASS_ZERO_TEND, "unknown gscond mask mode %d", mode);
    FV3HIP_REQUIRE(cloud_in && qv_in && t_in && cloud_emul && cloud_out && qv_out && t_out, "null pointer");
    if (mode != FV3HIP_ZC_NO_MASK) {
        FV3HIP_REQUIRE(aux && float_code(aux_dtype), "this mask mode needs its auxiliary array (F32 or F64)");
        if (mode >= FV3HIP_ZC_CLASS_ZERO_CLOUD)
            FV3HIP_REQUIRE(n_class >= 1 && cls >= 0 && cls < n_class, "class index %d out of range [0, %d)", cls, n_class);
    }
    hipStream_t st = as_stream(stream);
#define ZC_LAUNCH_(T)                                                                                              \
    if (phase_dependent)                                                                                           \
        hipLaunchKernelGGL((gscond_conserve_phase_kernel<T>), dim3((unsigned)n0), dim3(256), 0, st, cloud_in, qv_in, t_in, \
                           state_dtype, cloud_emul, emul_dtype, mode, aux, aux_dtype, n_class, cls, n0, n1,        \
                           static_cast<T *>(cloud_out), static_cast<T *>(qv_out), static_cast<T *>(t_out));        \
    else                                                                                                           \
        hipLaunchKernelGGL((gscond_conserve_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, cloud_in, qv_in, t_in, \
                           state_dtype, cloud_emul, emul_dtype, mode, aux, aux_dtype, n_class, cls, n,             \
                           static_cast<T *>(cloud_out), static_cast<T *>(qv_out), static_cast<T *>(t_out))
    if (out_dtype == FV3HIP_F64) {
        ZC_LAUNCH_(double);
    } else {
        ZC_LAUNCH_(float);
    }
#undef ZC_LAUNCH_
    return check_launch("gscond_conserve_kernel");
}

extern "C" int fv3hip_zc_precpd_conserve(const void *cloud_g, const void *qv_g, const void *t_g, const void *delp,
                                         int state_dtype, const void *cloud_p, const void *qv_p, int emul_dtype, int64_t n0,
                                         int64_t n1, int out_dtype, void *cloud_out, void *qv_out, void *t_out,
                                         void *precip_out, void *stream)
{
    FV3HIP_REQUIRE(n0 >= 0 && n1 >= 0, "negative extent");
    ZC_COMMON_CHECKS(n0 * n1);
    FV3HIP_REQUIRE(float_code(state_dtype) && float_code(emul_dtype), "arrays must be F32 or F64");
    FV3HIP_REQUIRE(cloud_g && qv_g && t_g && delp && cloud_p && qv_p && cloud_out && qv_out && t_out && precip_out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (out_dtype == FV3HIP_F64)
        hipLaunchKernelGGL((precpd_conserve_kernel<double>), dim3(grid_for(n1)), dim3(256), 0, st, cloud_g, qv_g, t_g, delp,
                           state_dtype, cloud_p, qv_p, emul_dtype, n0, n1, static_cast<double *>(cloud_out),
                           static_cast<double *>(qv_out), static_cast<double *>(t_out), static_cast<double *>(precip_out));
    else
        hipLaunchKernelGGL((precpd_conserve_kernel<float>), dim3(grid_for(n1)), dim3(256), 0, st, cloud_g, qv_g, t_g, delp,
                           state_dtype, cloud_p, qv_p, emul_dtype, n0, n1, static_cast<float *>(cloud_out),
                           static_cast<float *>(qv_out), static_cast<float *>(t_out), static_cast<float *>(precip_out));
    return check_launch("precpd_conserve_kernel");
}

extern "C" int fv3hip_zc_precip_simple(const void *cloud_g, const void *qv_g, const void *delp, int state_dtype,
                                       const void *cloud_p, const void *qv_p, int emul_dtype, int64_t n0, int64_t n1,
                                       int out_dtype, void *precip_out, void *stream)
{
    FV3HIP_REQUIRE(n0 >= 0 && n1 >= 0, "negative extent");
    ZC_COMMON_CHECKS(n1);
    FV3HIP_REQUIRE(float_code(state_dtype) && float_code(emul_dtype), "arrays must be F32 or F64");
    FV3HIP_REQUIRE(cloud_g && qv_g && delp && cloud_p && qv_p && precip_out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (out_dtype == FV3HIP_F64)
        hipLaunchKernelGGL((precip_simple_kernel<double>), dim3(grid_for(n1)), dim3(256), 0, st, cloud_g, qv_g, delp,
                           state_dtype, cloud_p, qv_p, emul_dtype, n0, n1, static_cast<double *>(precip_out));
    else
        hipLaunchKernelGGL((precip_simple_kernel<float>), dim3(grid_for(n1)), dim3(256), 0, st, cloud_g, qv_g, delp,
                           state_dtype, cloud_p, qv_p, emul_dtype, n0, n1, static_cast<float *>(precip_out));
    return check_launch("precip_simple_kernel");
}

extern "C" int fv3hip_clamp(const void *x, int dtype, int64_t n, double lo, double hi, int has_lo, int has_hi, void *out,
                            void *stream)
{
    FV3HIP_REQUIRE(float_code(dtype), "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n >= 0, "negative extent");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(x && out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((clamp_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, static_cast<const double *>(x), n, lo,
                           hi, has_lo, has_hi, static_cast<double *>(out));
    else
        hipLaunchKernelGGL((clamp_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, static_cast<const float *>(x), n,
                           (float)lo, (float)hi, has_lo, has_hi, static_cast<float *>(out));
    return check_launch("clamp_kernel");
}

extern "C" int fv3hip_level_fill(const void *emul, int emul_dtype, const void *src, int src_dtype, double fill_value,
                                 int64_t n0, int64_t n1, int64_t start, int64_t stop, void *out, void *stream)
{
    FV3HIP_REQUIRE(float_code(emul_dtype), "emul_dtype must be F32 or F64");
    FV3HIP_REQUIRE(!src || float_code(src_dtype), "src_dtype must be F32 or F64");
    FV3HIP_REQUIRE(n0 >= 0 && n1 >= 0, "negative extent");
    if (n0 * n1 == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(emul && out, "null pointer");
    hipLaunchKernelGGL(level_fill_kernel, dim3(grid_for(n0 * n1)), dim3(256), 0, as_stream(stream), emul, emul_dtype, src,
                       src_dtype, fill_value, n0, n1, start, stop, static_cast<double *>(out));
    return check_launch("level_fill_kernel");
}

extern "C" int fv3hip_zc_class_zero(const void *x, int dtype, const void *logits, int logits_dtype, int n_class, int cls,
                                    int64_t n, void *out, void *stream)
{
    FV3HIP_REQUIRE(float_code(dtype) && float_code(logits_dtype), "arrays must be F32 or F64");
    FV3HIP_REQUIRE(n_class >= 1 && cls >= 0 && cls < n_class, "class index %d out of range [0, %d)", cls, n_class);
    FV3HIP_REQUIRE(n >= 0, "negative extent");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(x && logits && out, "null pointer");
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((class_zero_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, static_cast<const double *>(x),
                           logits, logits_dtype, n_class, cls, n, static_cast<double *>(out));
    else
        hipLaunchKernelGGL((class_zero_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, static_cast<const float *>(x),
                           logits, logits_dtype, n_class, cls, n, static_cast<float *>(out));
    return check_launch("class_zero_kernel");
}

extern "C" int fv3hip_non_negative_sphum(const void *sphum, const void *q1, const void *q2, int dtype, int64_t n, double dt,
                                         int mse_conserving, void *q1_out, void *q2_out, void *stream)
{
    FV3HIP_REQUIRE(float_code(dtype), "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n >= 0, "negative extent");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(sphum && q2 && q2_out && (!q1 || q1_out), "null pointer");
    hipStream_t st = as_stream(stream);
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((non_negative_sphum_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st,
                           static_cast<const double *>(sphum), static_cast<const double *>(q1), static_cast<const double *>(q2),
                           n, dt, mse_conserving, static_cast<double *>(q1_out), static_cast<double *>(q2_out));
    else
        hipLaunchKernelGGL((non_negative_sphum_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st,
                           static_cast<const float *>(sphum), static_cast<const float *>(q1), static_cast<const float *>(q2), n,
                           (float)dt, mse_conserving, static_cast<float *>(q1_out), static_cast<float *>(q2_out));
    return check_launch("non_negative_sphum_kernel");
}
