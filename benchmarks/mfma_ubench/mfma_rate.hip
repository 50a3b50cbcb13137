#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: 8 independent accumulators round robin; 1: 2 accumulators alternating (dependent at distance 2); 2: one accumulator (back to back)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters)
{
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 a = {1.f, 2.f, 3.f, (float)threadIdx.x}, b = {0.5f, 0.25f, 1.f, 2.f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %2, %8, %9, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %8, %9, %3\n\t"
                "v_mfma_f32_32x32x16_bf16 %4, %8, %9, %4\n\tv_mfma_f32_32x32x16_bf16 %5, %8, %9, %5\n\t"
                "v_mfma_f32_32x32x16_bf16 %6, %8, %9, %6\n\tv_mfma_f32_32x32x16_bf16 %7, %8, %9, %7\n\t"
                : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
                : "+a"(acc[0]), "+a"(acc[1]) : "v"(a), "v"(b));
        } else {
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                : "+a"(acc[0]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, int grid, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double mf = (double)grid * 4 * iters * 8;
    printf("%s grid %d: %.3f ms  %.1f TFLOP/s bf16  %.2f cycles/MFMA/SIMD at 2.4 GHz\n", name, grid, ms, mf * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0) / ((grid + 255) / 256));
}
int main()
{
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    for (int grid : {256, 64}) {
        run<0>("8 independent acc   ", d, grid, 20000);
        run<1>("2 alternating acc   ", d, grid, 20000);
        run<2>("1 acc back to back  ", d, grid, 20000);
    }
    return 0;
}
