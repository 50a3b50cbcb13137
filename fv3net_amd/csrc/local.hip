// Pre- and post-passes of the "dense-local" microphysics emulators (one MLP shared by all levels,
// applied to every (level, column) point) and the classifier decode of ModelWithClassifier.
//
// Replaces (TensorFlow graph pieces on the host in the reference):
//   external/fv3fit/fv3fit/emulation/layers/architecture.py:53-75   combine_sequence_inputs
//   external/fv3fit/fv3fit/emulation/layers/fields.py:6-66          FieldInput / FieldOutput (per-level norm)
//   external/fv3fit/fv3fit/emulation/transforms/transforms.py:111-129  LogTransform.forward
//   external/fv3fit/fv3fit/emulation/transforms/transforms.py:192-224  ConditionallyScaledTransform.backward
//   external/fv3fit/fv3fit/keras/math.py:5-23                       piecewise (0th-order interpolation)
//   external/fv3fit/fv3fit/emulation/transforms/transforms.py:55-58 Difference.backward
//   external/fv3fit/fv3fit/emulation/transforms/transforms.py:131-158 LimitValueTransform.backward
//   external/emulation/emulation/zhao_carr.py:193-198               _get_classify_output
//
// The network between them is mlp_fused_kernel on the packed [n_inputs][nz * ncol] array: a
// (level, column) point is one sample.  All state arrays are [nz][ncol] (call_py_fort's
// [feature, sample]) or [ncol]; HBM-bound, one pass each.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/fv3hip.h"
#include "common.h"

namespace fv3hip {
namespace {

__device__ __forceinline__ float ldf(const void *p, int dt, int64_t i)
{
    // Keras casts float64 inputs to the layer dtype first
    return dt == FV3HIP_F64 ? (float)static_cast<const double *>(p)[i] : static_cast<const float *>(p)[i];
}

// one network input: out[z][c] = (t(x[z or 0][c]) - center[z]) / scale[z]
__global__ void local_pack_kernel(const void *__restrict__ x, int dtype, int has_levels, int transform, float eps,
                                  const float *__restrict__ center, const float *__restrict__ scale, int nz, int64_t ncol,
                                  float *__restrict__ out)
{
    const int z = blockIdx.y;
    const float c = center[z], s = scale[z];
    const int64_t row_in = has_levels ? (int64_t)z * ncol : 0, row_out = (int64_t)z * ncol;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncol; i += (int64_t)gridDim.x * blockDim.x) {
        float v = ldf(x, dtype, row_in + i);
        if (transform == FV3HIP_TRANSFORM_LOG) v = logf(v > eps ? v : eps);  // log(max(x, epsilon))
        out[row_out + i] = (v - c) / s;
    }
}

// piecewise(edges[:-1], values, x): values[max(upper_bound(edges, x) - 1, 0)]
__device__ __forceinline__ int bin_of(const float *__restrict__ edges, int n, float x)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (x < edges[mid]) hi = mid;
        else lo = mid + 1;
    }
    return lo > 0 ? lo - 1 : 0;
}

// LimitValueTransform.backward (transforms.py:148-158): keras relu(x, threshold=lower), then (x < upper) * x;
// NaN stays NaN
__device__ __forceinline__ float limit_value(float v, int flags, float lo, float hi)
{
    if ((flags & 1) && v < lo) v = 0.f;
    if ((flags & 2) && !(v < hi) && v == v) v = 0.f;
    return v;
}

// one network output channel: direct = yhat * scale[z] + center[z];
// unscaled = direct * max(cs_scale[bin(on)], min_scale) + cs_center[bin(on)]   (when conditional scaling is configured);
// the value limits apply to the last of these;
// after = limit(before + value)                                                 (when a Difference is configured)
__global__ void local_unpack_kernel(const float *__restrict__ yhat, int64_t yhat_level_stride, const float *__restrict__ scale,
                                    const float *__restrict__ center, const void *__restrict__ cond_on, int cond_dtype,
                                    const float *__restrict__ edges, const float *__restrict__ cs_scale,
                                    const float *__restrict__ cs_center, int n_bins, float min_scale,
                                    const void *__restrict__ before, int before_dtype, int limit_flags, float value_lo,
                                    float value_hi, float after_lo, float after_hi, int nz, int64_t ncol,
                                    float *__restrict__ out_direct, float *__restrict__ out_unscaled,
                                    float *__restrict__ out_after)
{
    const int z = blockIdx.y;
    const float s = scale ? scale[z] : 1.f, c = center ? center[z] : 0.f;
    const int64_t row = (int64_t)z * ncol, yrow = (int64_t)z * yhat_level_stride;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncol; i += (int64_t)gridDim.x * blockDim.x) {
        float v = yhat[yrow + i] * s + c;
        if (cond_on) {
            if (out_direct) out_direct[row + i] = v;
            const int b = bin_of(edges, n_bins, ldf(cond_on, cond_dtype, row + i));
            const float sc = cs_scale[b];
            v = v * (sc > min_scale ? sc : min_scale) + cs_center[b];
            v = limit_value(v, limit_flags, value_lo, value_hi);
            if (out_unscaled) out_unscaled[row + i] = v;
        } else {
            v = limit_value(v, limit_flags, value_lo, value_hi);
            if (out_direct) out_direct[row + i] = v;
        }
        if (before) out_after[row + i] = limit_value(ldf(before, before_dtype, row + i) + v, limit_flags >> 2, after_lo, after_hi);
    }
}

// one-hot by arg-max with every tied maximum hot (logits == max over classes), plus the union of two classes
__global__ void classify_onehot_kernel(const void *__restrict__ logits, int dtype, int n_class, int64_t n,
                                       uint8_t *__restrict__ onehot, uint8_t *__restrict__ any_of, int cls_a, int cls_b)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double mx = dtype == FV3HIP_F64 ? static_cast<const double *>(logits)[i] : (double)static_cast<const float *>(logits)[i];
        for (int c = 1; c < n_class; ++c) {
            const double v = dtype == FV3HIP_F64 ? static_cast<const double *>(logits)[(int64_t)c * n + i]
                                                 : (double)static_cast<const float *>(logits)[(int64_t)c * n + i];
            mx = v > mx ? v : mx;
        }
        uint8_t both = 0;
        for (int c = 0; c < n_class; ++c) {
            const double v = dtype == FV3HIP_F64 ? static_cast<const double *>(logits)[(int64_t)c * n + i]
                                                 : (double)static_cast<const float *>(logits)[(int64_t)c * n + i];
            const uint8_t hot = (v == mx) ? 1 : 0;
            onehot[(int64_t)c * n + i] = hot;
            if (c == cls_a || c == cls_b) both |= hot;
        }
        if (any_of) any_of[i] = both;
    }
}

inline dim3 level_grid(int nz, int64_t ncol)
{
    int64_t bx = ceil_div(ncol, 256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 4096) bx = 4096;
    return dim3((unsigned)bx, (unsigned)nz);
}

}  // namespace
}  // namespace fv3hip

using namespace fv3hip;

extern "C" int fv3hip_local_pack(const void *x, int dtype, int has_levels, int transform, double eps, const float *center,
                                 const float *scale, int nz, int64_t ncol, float *out, void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(transform == FV3HIP_TRANSFORM_NONE || transform == FV3HIP_TRANSFORM_LOG, "unknown transform");
    FV3HIP_REQUIRE(nz >= 0 && nz <= 65535 && ncol >= 0, "bad extent");
    if (nz == 0 || ncol == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(x && center && scale && out, "null pointer");
    hipLaunchKernelGGL(local_pack_kernel, level_grid(nz, ncol), dim3(256), 0, as_stream(stream), x, dtype, has_levels ? 1 : 0,
                       transform, (float)eps, center, scale, nz, ncol, out);
    return check_launch("local_pack_kernel");
}

extern "C" int fv3hip_local_unpack(const float *yhat, int64_t yhat_level_stride, const float *scale, const float *center,
                                   const void *cond_on, int cond_dtype, const float *edges, const float *cs_scale,
                                   const float *cs_center, int n_bins, double min_scale, const void *before, int before_dtype,
                                   int limit_flags, double value_lower, double value_upper, double after_lower,
                                   double after_upper, int nz, int64_t ncol, float *out_direct, float *out_unscaled,
                                   float *out_after, void *stream)
{
    FV3HIP_REQUIRE(limit_flags >= 0 && limit_flags < 16, "limit_flags is a 4-bit mask");
    FV3HIP_REQUIRE(yhat_level_stride >= ncol || nz <= 1, "yhat_level_stride must be at least ncol");
    FV3HIP_REQUIRE(nz >= 0 && nz <= 65535 && ncol >= 0, "bad extent");
    if (nz == 0 || ncol == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(yhat, "null pointer");
    if (cond_on) {
        FV3HIP_REQUIRE(cond_dtype == FV3HIP_F32 || cond_dtype == FV3HIP_F64, "cond_dtype must be F32 or F64");
        FV3HIP_REQUIRE(edges && cs_scale && cs_center && n_bins >= 1, "conditional scaling needs edges, scale and center tables");
    }
    if (before) {
        FV3HIP_REQUIRE(before_dtype == FV3HIP_F32 || before_dtype == FV3HIP_F64, "before_dtype must be F32 or F64");
        FV3HIP_REQUIRE(out_after, "a Difference needs its output array");
    }
    FV3HIP_REQUIRE(out_direct || out_unscaled || out_after, "no output requested");
    hipLaunchKernelGGL(local_unpack_kernel, level_grid(nz, ncol), dim3(256), 0, as_stream(stream), yhat, yhat_level_stride, scale,
                       center, cond_on, cond_dtype, edges, cs_scale, cs_center, n_bins, (float)min_scale, before, before_dtype,
                       limit_flags, (float)value_lower, (float)value_upper, (float)after_lower, (float)after_upper, nz, ncol,
                       out_direct, out_unscaled, out_after);
    return check_launch("local_unpack_kernel");
}

extern "C" int fv3hip_classify_onehot(const void *logits, int dtype, int n_class, int64_t n, uint8_t *onehot, uint8_t *any_of,
                                      int cls_a, int cls_b, void *stream)
{
    FV3HIP_REQUIRE(dtype == FV3HIP_F32 || dtype == FV3HIP_F64, "dtype must be F32 or F64");
    FV3HIP_REQUIRE(n_class >= 1 && n >= 0, "bad extent");
    if (n == 0) return FV3HIP_OK;
    FV3HIP_REQUIRE(logits && onehot, "null pointer");
    int64_t blocks = ceil_div(n, 256);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(classify_onehot_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), logits, dtype, n_class, n,
                       onehot, any_of, cls_a, cls_b);
    return check_launch("classify_onehot_kernel");
}
