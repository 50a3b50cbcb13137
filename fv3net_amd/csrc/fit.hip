// Predict-side glue of the fv3fit composite models on the device, applied to the [.., z, ..]
// prediction arrays right after the network.
//
// Replaces (xarray on the host in the reference):
//   external/fv3fit/fv3fit/_shared/config.py:11-24 + external/vcm/vcm/calc/calc.py:52-56
//       TaperConfig.apply: prediction * vertical_tapering_scale_factors along the taper dim
//   external/fv3fit/fv3fit/_shared/models.py:253-260
//       EnsembleModel.predict: xr.concat(member predictions, "member").mean / .median (NaN-skipping)
// (SquashedOutputConfig.squash, config.py:135-142, is two fv3hip_ew steps.)  HBM-bound, one pass.
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
