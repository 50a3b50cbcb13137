#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
#define VA "v_add_f32 %4, %4, %5\n\t"
#define VP "v_cvt_pk_bf16_f32 %4, %5, %5\n\t"
#define DS "ds_read_b128 %6, %7\n\t"
template <int K>  // K VALU after each MFMA pair
__global__ __launch_bounds__(256, 1) void k(float* out, int iters)
{
    __shared__ f32x4 sm[1024];
    sm[threadIdx.x] = f32x4{1, 2, 3, 4};
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    f32x4 a = {1.f, 2.f, 3.f, (float)threadIdx.x}, b = {0.5f, 0.25f, 1.f, 2.f}, d;
    float v = 1.f, w = 0.5f;
    uint32_t addr = threadIdx.x * 16;
    for (int it = 0; it < iters; ++it) {
        if (K == 0) asm volatile(MF MF MF MF : "+a"(a0), "+a"(a1) : "v"(a), "v"(b));
        if (K == 1) asm volatile(MF VA MF VA MF VA MF VA : "+a"(a0), "+a"(a1) : "v"(a), "v"(b), "v"(v), "v"(w));
        if (K == 2) asm volatile(MF VA VA MF VA VA MF VA VA MF VA VA : "+a"(a0), "+a"(a1) : "v"(a), "v"(b), "v"(v), "v"(w));
        if (K == 4) asm volatile(MF VA VA VA VA MF VA VA VA VA MF VA VA VA VA MF VA VA VA VA : "+a"(a0), "+a"(a1) : "v"(a), "v"(b), "v"(v), "v"(w));
        if (K == 8) asm volatile(MF VA VA VA VA VA VA VA VA MF VA VA VA VA VA VA VA VA MF VA VA VA VA VA VA VA VA MF VA VA VA VA VA VA VA VA : "+a"(a0), "+a"(a1) : "v"(a), "v"(b), "v"(v), "v"(w));
        if (K == 16) asm volatile(MF VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA MF VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA MF VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA MF VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA VA : "+a"(a0), "+a"(a1) : "v"(a), "v"(b), "v"(v), "v"(w));
        if (K == 104) asm volatile(MF VP VP VP VP MF VP VP VP VP MF VP VP VP VP MF VP VP VP VP : "+a"(a0), "+a"(a1) : "v"(a), "v"(b), "v"(v), "v"(w));
        if (K == 200) asm volatile(MF DS DS MF DS DS MF DS DS MF DS DS "s_waitcnt lgkmcnt(0)\n\t" : "+a"(a0), "+a"(a1) : "v"(a), "v"(b), "v"(v), "v"(w), "v"(d), "v"(addr));
        if (K == 300) asm volatile(MF MF MF MF "s_barrier\n\t" : "+a"(a0), "+a"(a1) : "v"(a), "v"(b));
    }
    float s = v;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int K> void run(const char* name, float* d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<K>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<K>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%s: %.3f ms  %.1f ns per group of 2 MFMAs (+extras)  = %.1f cycles at 2.4 GHz\n", name, ms, ms * 1e6 / (iters * 4.0), ms * 1e-3 * 2.4e9 / (iters * 4.0));
}
int main()
{
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    run<0>("2 MFMA                    ", d, 20000);
    run<1>("2 MFMA + 1 v_add          ", d, 20000);
    run<2>("2 MFMA + 2 v_add          ", d, 20000);
    run<4>("2 MFMA + 4 v_add          ", d, 20000);
    run<8>("2 MFMA + 8 v_add          ", d, 20000);
    run<16>("2 MFMA + 16 v_add         ", d, 20000);
    run<104>("2 MFMA + 4 v_cvt_pk_bf16  ", d, 20000);
    run<200>("2 MFMA + 2 ds_read_b128   ", d, 20000);
    run<300>("8 MFMA + barrier (per 2)  ", d, 20000);
    return 0;
}
