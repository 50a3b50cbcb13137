// Predict-side glue of the fv3fit composite models on the device, applied to the [.., z, ..]
// prediction arrays right after the network.
//
// Replaces (xarray on the host in the reference):
//   external/fv3fit/fv3fit/_shared/config.py:11-24 + external/vcm/vcm/calc/calc.py:52-56
//       TaperConfig.apply: prediction * vertical_tapering_scale_factors along the taper dim
//   external/fv3fit/fv3fit/_shared/models.py:253-260
//       EnsembleModel.predict: xr.concat(member predictions, "member").mean / .median (NaN-skipping)
// (SquashedOutputConfig.squash, config.py:135-142, is two fv3hip_ew steps.)  HBM-bound, one pass.
//   external/vcm/vcm/calc/flux_form.py:7-104 (+ thermo/vertically_dependent.py:18-38)
//       the flux-form output transforms of TransformedPredictor (vcm/data_transform.py:140-300): tendencies <-> net
//       fluxes at the cell interfaces, the surface flux that closes the column budget
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/fv3hip.h"
#include "common.h"

namespace fv3hip {
namespace {

constexpr int kMaxMembers = 32;

struct MemberPtrs {
    const void *p[kMaxMembers];
};

// out[o][z][i] = scale[z] * x[o][z][i] in float64: the scale factors are numpy float64, and float64 * float32 promotes
template <typename T>
__global__ void level_scale_kernel(const T *__restrict__ x, const double *__restrict__ scale, int64_t n_outer, int nz,
                                   int64_t n_inner, double *__restrict__ out)
{
    const int64_t total = n_outer * nz * n_inner;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)((idx / n_inner) % nz);
        out[idx] = scale[z] * (double)x[idx];
    }
}

// mean / median over the members, skipping NaNs (all-NaN -> NaN), as xarray's reductions do for floats
template <typename T>
__global__ void member_reduce_kernel(MemberPtrs m, int n_members, int op, int64_t n, T *__restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T v[kMaxMembers];
        int cnt = 0;
#pragma unroll 1
        for (int k = 0; k < n_members; ++k) {
            const T x = static_cast<const T *>(m.p[k])[i];
            if (x == x) v[cnt++] = x;
        }
        T r;
        if (cnt == 0) {
            r = (T)NAN;
        } else if (op == FV3HIP_OP_MEAN) {
            T s = 0;
            for (int k = 0; k < cnt; ++k) s += v[k];
            r = s / (T)cnt;
        } else {  // median: insertion sort of at most 32 values, mean of the two middle ones
            for (int a = 1; a < cnt; ++a) {
                const T key = v[a];
                int b = a - 1;
                while (b >= 0 && v[b] > key) {
                    v[b + 1] = v[b];
                    --b;
                }
                v[b + 1] = key;
            }
            r = (cnt & 1) ? v[cnt >> 1] : (T)0.5 * (v[(cnt >> 1) - 1] + v[cnt >> 1]);
        }
        out[i] = r;
    }
}

constexpr double kGravity = 9.80665;  // vcm/calc/thermo/constants.py:2

// One thread per column of [outer][nz][inner] arrays; toa / surface fluxes are [outer][inner] (toa may be null = 0).
// flux_form.py:34-45: flux = -cumsum(tendency * delp / g) shifted down one interface, + toa; the surface downward flux is
// what is left at the lowest interface plus the upward flux, not below zero when rectified.  Same operations in the same
// order as numpy's (cumsum is a running sum), in the arrays' own precision.
// closure = 1 (flux_form.py:69-75): no interface fluxes, downward = toa + upward - sum(tendency * delp / g).
template <typename T>
__global__ void tendency_to_flux_kernel(const T *__restrict__ tend, const T *__restrict__ delp, const T *__restrict__ toa,
                                        const T *__restrict__ up, int64_t n_outer, int nz, int64_t inner, int rectify, int closure,
                                        T *__restrict__ flux, T *__restrict__ down)
{
    const int64_t col = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (col >= n_outer * inner) return;
    const int64_t o = col / inner, i = col - o * inner;
    const T g = (T)kGravity, t0 = toa ? toa[col] : (T)0;
    T run = 0, d;
    if (closure) {
        for (int k = 0; k < nz; ++k) {
            const int64_t at = (o * nz + k) * inner + i;
            run = run + tend[at] * delp[at] / g;
        }
        d = t0 + up[col] - run;
    } else {
        T at_interface = (T)0 + t0;  // (the padded zero at the model top)
        for (int k = 0; k < nz; ++k) {
            const int64_t at = (o * nz + k) * inner + i;
            flux[at] = at_interface;
            const T term = tend[at] * delp[at] / g;
            run = (k == 0) ? term : run + term;
            at_interface = -run + t0;
        }
        d = at_interface + up[col];
    }
    if (rectify) d = (d >= (T)0) ? d : (T)0;  // x.where(x >= 0, 0): a NaN becomes 0
    down[col] = d;
}

// flux_form.py:95-104: tendency = -(g * diff(concat(net_flux, down - up)) / delp)
template <typename T>
__global__ void flux_to_tendency_kernel(const T *__restrict__ flux, const T *__restrict__ down, const T *__restrict__ up,
                                        const T *__restrict__ delp, int64_t n_outer, int nz, int64_t inner, T *__restrict__ tend)
{
    const int64_t col = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (col >= n_outer * inner) return;
    const int64_t o = col / inner, i = col - o * inner;
    const T g = (T)kGravity, surface_net = down[col] - up[col];
    T above = flux[(o * nz) * inner + i];
    for (int k = 0; k < nz; ++k) {
        const int64_t at = (o * nz + k) * inner + i;
        const T below = (k + 1 < nz) ? flux[at + inner] : surface_net;
        tend[at] = -(g * (below - above) / delp[at]);
        above = below;
    }
}

// MinMaxNoveltyDetector.predict (fv3fit/sklearn/_min_max_novelty_detector.py:94-121): one launch per packed variable folds
// its features into the running maximum / minimum of MinMaxScaler.transform(X) = X * scale_ + min_ (float64, as sklearn
// computes it); the last launch turns them into the score max(max - 1, 0) + max(-min, 0).
template <typename T>
__global__ void minmax_score_kernel(const T *__restrict__ x, int64_t feat_stride, int64_t sample_stride, int n_feat,
                                    const double *__restrict__ scale, const double *__restrict__ offset, int64_t n, int first,
                                    int finish, double *__restrict__ run_max, double *__restrict__ run_min, double *__restrict__ score)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double hi = first ? -INFINITY : run_max[i], lo = first ? INFINITY : run_min[i];
    bool bad = false;
    for (int f = 0; f < n_feat; ++f) {
        const double v = (double)x[f * feat_stride + i * sample_stride] * scale[f] + offset[f];
        bad = bad || (v != v);
        hi = v > hi ? v : hi;
        lo = v < lo ? v : lo;
    }
    if (bad) hi = lo = NAN;  // (numpy's max / min propagate a NaN)
    run_max[i] = hi;
    run_min[i] = lo;
    if (finish) {
        const double a = hi - 1.0, b = -1.0 * lo;
        score[i] = (hi != hi) ? NAN : ((a > 0.0 ? a : 0.0) + (b > 0.0 ? b : 0.0));
    }
}

// OCSVMNoveltyDetector.predict (fv3fit/sklearn/_ocsvm_novelty_detector.py:124-160): the negated
// Pipeline(StandardScaler, OneClassSVM(kernel="rbf")).score_samples, i.e. -sum_i dual_coef_i exp(-gamma |z - sv_i|^2) with
// z = (x - mean) / scale, float64.  A workgroup holds 64 samples' standardised features in LDS ([feature][sample]: lanes
// read consecutive words) and every lane walks the support vectors, whose rows are wave-uniform (scalar) reads.
constexpr int kSvmSamples = 64;

__global__ __launch_bounds__(kSvmSamples) void ocsvm_score_kernel(const double *__restrict__ x, int n_feat, int64_t n,
                                                                   const double *__restrict__ mean, const double *__restrict__ scale,
                                                                   const double *__restrict__ sv, const double *__restrict__ coef,
                                                                   int n_sv, double gamma, double *__restrict__ score)
{
    extern __shared__ double svm_z[];  // [n_feat][64]
    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x * (int64_t)kSvmSamples + lane, ic = i < n ? i : n - 1;
    for (int f = 0; f < n_feat; ++f) svm_z[f * kSvmSamples + lane] = (x[(int64_t)f * n + ic] - mean[f]) / scale[f];
    double total = 0.0;
    for (int v = 0; v < n_sv; ++v) {
        const double *row = sv + (int64_t)v * n_feat;
        double d2 = 0.0;
        for (int f = 0; f < n_feat; ++f) {
            const double d = row[f] - svm_z[f * kSvmSamples + lane];
            d2 += d * d;
        }
        total += coef[v] * exp(-gamma * d2);
    }
    if (i < n) score[i] = -1.0 * total;
}

inline unsigned flat_grid(int64_t n)
{
    int64_t b = ceil_div(n, 256);
    return (unsigned)(b > 256 * 64 ? 256 * 64 : (b < 1 ? 1 : b));
}

}  // namespace
}  // namespace fv3hip

using namespace fv3hip;

extern "C" int fv3hip_level_scale(const void *x, int dtype, const double *scale, int64_t n_outer, int nz, int64_t n_inner,
                                  double *out, void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_outer >= 0 && nz >= 0 && n_inner >= 0, "negative extent");
    const int64_t total = n_outer * nz * n_inner;
    if (total == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(x && scale && out, "null pointer");
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((level_scale_kernel<double>), dim3(flat_grid(total)), dim3(256), 0, as_stream(stream),
                           static_cast<const double *>(x), scale, n_outer, nz, n_inner, out);
    else
        hipLaunchKernelGGL((level_scale_kernel<float>), dim3(flat_grid(total)), dim3(256), 0, as_stream(stream),
                           static_cast<const float *>(x), scale, n_outer, nz, n_inner, out);
    return check_launch("level_scale_kernel");
}

extern "C" int fv3hip_member_reduce(const void *const *members, int n_members, int dtype, int op, int64_t n, void *out,
                                    void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(op == FV3HIP_OP_MEAN || op == FV3HIP_OP_MEDIAN, "op must be FV3HIP_OP_MEAN or FV3HIP_OP_MEDIAN");
    FV3HIP_REQUIRE(n_members >= 1 && n_members <= kMaxMembers, "between 1 and %d members", kMaxMembers);
    FV3HIP_REQUIRE(n >= 0, "negative extent");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(members && out, "null pointer");
    MemberPtrs m;
    for (int k = 0; k < kMaxMembers; ++k) m.p[k] = nullptr;
    for (int k = 0; k < n_members; ++k) {
        FV3HIP_REQUIRE(members[k], "null member pointer");
        m.p[k] = members[k];
    }
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((member_reduce_kernel<double>), dim3(flat_grid(n)), dim3(256), 0, as_stream(stream), m, n_members, op, n,
                           static_cast<double *>(out));
    else
        hipLaunchKernelGGL((member_reduce_kernel<float>), dim3(flat_grid(n)), dim3(256), 0, as_stream(stream), m, n_members, op, n,
                           static_cast<float *>(out));
    return check_launch("member_reduce_kernel");
}

extern "C" int fv3hip_tendency_to_flux(const void *tendency, const void *delp, const void *toa_net_flux,
                                       const void *surface_upward_flux, int dtype, int64_t n_outer, int nz, int64_t n_inner,
                                       int rectify, int closure, void *net_flux, void *surface_downward_flux, void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_outer >= 0 && nz >= 1 && n_inner >= 0, "bad extents");
    const int64_t cols = n_outer * n_inner;
    if (cols == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(tendency && delp && surface_upward_flux && surface_downward_flux && (closure || net_flux), "null pointer");
    const dim3 grid((unsigned)ceil_div(cols, (int64_t)256));
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((tendency_to_flux_kernel<double>), grid, dim3(256), 0, as_stream(stream), static_cast<const double *>(tendency),
                           static_cast<const double *>(delp), static_cast<const double *>(toa_net_flux),
                           static_cast<const double *>(surface_upward_flux), n_outer, nz, n_inner, rectify, closure,
                           static_cast<double *>(net_flux), static_cast<double *>(surface_downward_flux));
    else
        hipLaunchKernelGGL((tendency_to_flux_kernel<float>), grid, dim3(256), 0, as_stream(stream), static_cast<const float *>(tendency),
                           static_cast<const float *>(delp), static_cast<const float *>(toa_net_flux),
                           static_cast<const float *>(surface_upward_flux), n_outer, nz, n_inner, rectify, closure,
                           static_cast<float *>(net_flux), static_cast<float *>(surface_downward_flux));
    return check_launch("tendency_to_flux_kernel");
}

extern "C" int fv3hip_flux_to_tendency(const void *net_flux, const void *surface_downward_flux, const void *surface_upward_flux,
                                       const void *delp, int dtype, int64_t n_outer, int nz, int64_t n_inner, void *tendency,
                                       void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_outer >= 0 && nz >= 1 && n_inner >= 0, "bad extents");
    const int64_t cols = n_outer * n_inner;
    if (cols == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(net_flux && surface_downward_flux && surface_upward_flux && delp && tendency, "null pointer");
    const dim3 grid((unsigned)ceil_div(cols, (int64_t)256));
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((flux_to_tendency_kernel<double>), grid, dim3(256), 0, as_stream(stream), static_cast<const double *>(net_flux),
                           static_cast<const double *>(surface_downward_flux), static_cast<const double *>(surface_upward_flux),
                           static_cast<const double *>(delp), n_outer, nz, n_inner, static_cast<double *>(tendency));
    else
        hipLaunchKernelGGL((flux_to_tendency_kernel<float>), grid, dim3(256), 0, as_stream(stream), static_cast<const float *>(net_flux),
                           static_cast<const float *>(surface_downward_flux), static_cast<const float *>(surface_upward_flux),
                           static_cast<const float *>(delp), n_outer, nz, n_inner, static_cast<float *>(tendency));
    return check_launch("flux_to_tendency_kernel");
}

extern "C" int fv3hip_minmax_score(const void *x, int dtype, int64_t feat_stride, int64_t sample_stride, int n_feat,
                                   const double *scale, const double *offset, int64_t n, int first, int finish, double *run_max,
                                   double *run_min, double *score, void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n >= 0 && n_feat >= 1, "bad extents");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(x && scale && offset && run_max && run_min && (score || !finish), "null pointer");
    const dim3 grid((unsigned)ceil_div(n, (int64_t)256));
    if (dtype == FV3HIP_F64)
        hipLaunchKernelGGL((minmax_score_kernel<double>), grid, dim3(256), 0, as_stream(stream), static_cast<const double *>(x),
                           feat_stride, sample_stride, n_feat, scale, offset, n, first, finish, run_max, run_min, score);
    else
        hipLaunchKernelGGL((minmax_score_kernel<float>), grid, dim3(256), 0, as_stream(stream), static_cast<const float *>(x),
                           feat_stride, sample_stride, n_feat, scale, offset, n, first, finish, run_max, run_min, score);
    return check_launch("minmax_score_kernel");
}

extern "C" int fv3hip_ocsvm_score(const double *x, int n_feat, int64_t n, const double *mean, const double *scale,
                                  const double *support_vectors, const double *dual_coef, int n_sv, double gamma, double *score,
                                  void *stream)
{
    FV3HIP_REQUIRE(n >= 0 && n_feat >= 1 && n_sv >= 0, "bad extents");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(x && mean && scale && score && (n_sv == 0 || (support_vectors && dual_coef)), "null pointer");
    const size_t lds = (size_t)n_feat * kSvmSamples * sizeof(double);
    if (lds > 160 * 1024) return fail(FV3HIP_EUNSUPPORTED, "%d packed features need %zu bytes of LDS (> 160 KiB)", n_feat, lds);
    FV3HIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ocsvm_score_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ocsvm_score_kernel, dim3((unsigned)ceil_div(n, (int64_t)kSvmSamples)), dim3(kSvmSamples), lds, as_stream(stream), x,
                       n_feat, n, mean, scale, support_vectors, dual_coef, n_sv, gamma, score);
    return check_launch("ocsvm_score_kernel");
}
