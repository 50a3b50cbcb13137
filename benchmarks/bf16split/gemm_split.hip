// EXPLORATORY: a dense layer  C[f][n] = sum_k W[k][f] * X[k][n]  (K = F = 256, the hidden layer of the
// Zhao-Carr emulator) with the fp32 operands split into NS bf16 pieces and contracted on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation):
//   NS = 2:  x = hi + lo          -> products hi*hi, hi*lo, lo*hi                      (3 MFMAs, ~16 mantissa bits)
//   NS = 3:  x = hi + mid + lo    -> hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid      (6 MFMAs, ~24 mantissa bits)
// against the fp32 MFMA (v_mfma_f32_32x32x2_f32) of the product kernel.  Not part of libfv3hip; the headline stays fp32.
// Layouts: W pieces are packed on the host in A-operand order [piece][kstep 16][ftile 8][lane 64][8 bf16]; X is [K][N] fp32
// (the product kernel's [feature][sample]); a workgroup is 4 waves x 64 samples; weights go global -> registers -> LDS
// (double buffered per k-step, shared by the 4 waves); the X values of a k-step are split in registers.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

template <int NS>
__global__ __launch_bounds__(256, 1) void gemm_split_kernel(const f32x4 *__restrict__ Wp, const float *__restrict__ X,
                                                            float *__restrict__ C, int64_t N, int reps)
{
    constexpr int K = 256, KS = 16, FT = 8, SETS = 2;           // k-steps of 16, feature tiles of 32, sample sets of 32 per wave
    constexpr int CHUNK = NS * FT * 64;                         // float4 (= 8 bf16) per k-step chunk
    __shared__ f32x4 wbuf[2][CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * SETS);
    f32x16 acc[SETS][FT];
#pragma unroll
    for (int s = 0; s < SETS; ++s)
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;
    constexpr int PER = CHUNK / 256;  // float4 per thread per chunk
    f32x4 stage[PER];
    auto issue = [&](int ks) {
#pragma unroll
        for (int i = 0; i < PER; ++i) stage[i] = Wp[(size_t)ks * CHUNK + tid + i * 256];
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) wbuf[buf][tid + i * 256] = stage[i];
    };
    float xn[SETS][8];  // the X values of the NEXT k-step, requested one k-step ahead
    auto load_x = [&](int ks) {
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
            const int64_t n = n0 + s * 32 + col;
#pragma unroll
            for (int j = 0; j < 8; ++j) xn[s][j] = (n < N) ? X[(size_t)(ks * 16 + half * 8 + j) * N + n] : 0.f;
        }
    };
    for (int rep = 0; rep < reps; ++rep) {
        issue(0);
        load_x(0);
        commit(0);
        __syncthreads();
        for (int ks = 0; ks < KS; ++ks) {
            const int buf = ks & 1;
            // this lane's 8 X values of the k-step for each sample set, split into NS bf16 vectors
            bf16x8 xb[SETS][NS];
#pragma unroll
            for (int s = 0; s < SETS; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float r = xn[s][j];
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
                        const __bf16 b = (__bf16)r;
                        xb[s][p][j] = b;
                        r = r - (float)b;
                    }
                }
            }
            if (ks + 1 < KS) {
                issue(ks + 1);
                load_x(ks + 1);
            }
#pragma unroll
            for (int t = 0; t < FT; ++t) {
                bf16x8 a[NS];
#pragma unroll
                for (int p = 0; p < NS; ++p) a[p] = __builtin_bit_cast(bf16x8, wbuf[buf][(p * FT + t) * 64 + lane]);
#pragma unroll
                for (int s = 0; s < SETS; ++s) {
                    // smallest products first
                    if (NS == 3) {
                        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], xb[s][1], acc[s][t], 0, 0, 0);
                        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], xb[s][2], acc[s][t], 0, 0, 0);
                        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], xb[s][0], acc[s][t], 0, 0, 0);
                    }
                    if (NS >= 2) {
                        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], xb[s][1], acc[s][t], 0, 0, 0);
                        acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], xb[s][0], acc[s][t], 0, 0, 0);
                    }
                    acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], xb[s][0], acc[s][t], 0, 0, 0);
                }
            }
            if (ks + 1 < KS) commit(buf ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
        const int64_t n = n0 + s * 32 + col;
        if (n < N) {
#pragma unroll
            for (int t = 0; t < FT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) C[(size_t)(t * 32 + rho(r) + 4 * half) * N + n] = acc[s][t][r];
        }
    }
}

extern "C" int bf16split_gemm(int ns, const void *Wp, const float *X, float *C, int64_t N, int reps, void *stream)
{
    const int64_t blocks = (N + 255) / 256;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ns == 1)
        hipLaunchKernelGGL((gemm_split_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const f32x4 *>(Wp), X, C, N, reps);
    else if (ns == 2)
        hipLaunchKernelGGL((gemm_split_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const f32x4 *>(Wp), X, C, N, reps);
    else
        hipLaunchKernelGGL((gemm_split_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const f32x4 *>(Wp), X, C, N, reps);
    return (int)hipGetLastError();
}
