// Issue rate of the vector instructions the remap kernel is made of, per SIMD, at 1 / 2 / 4 / 8 waves per SIMD.
//   hipcc -O2 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
// Every wave runs ITERS trips of 32 independent instructions of one kind (16 registers, each written twice per trip) and
// stamps s_memtime around the loop; cycles per instruction per SIMD = wave cycles / (ITERS * 32 * waves per SIMD).
// The answer decides what "VALU-issue bound" means for mappm_sweep_kernel: 4 cycles per wave64 instruction
// (a 16-lane SIMD) or 2 (a 32-lane one), and whether a packed v_pk_* costs one slot or two.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t *cycles, float *sink, int iters, float seed)
{
    float a[16];
    double d[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = seed + i + threadIdx.x;
    float b = seed * 0.5f, c = seed * 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{seed + i, seed - i};
    f2 pb = f2{b, c};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R16(X)
#undef X
            } else if (OP == 1) {  // 16 packed = 8 register pairs twice
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(pb));
                R16(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                R16(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R16(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
                R16(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                R16(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                R16(X)
#undef X
            } else if (OP == 7) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                R16(X)
#undef X
            } else if (OP == 8) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(pb));
                R16(X)
#undef X
            } else if (OP == 9) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[i]) : "v"(d[i & 7]));
                R16(X)
#undef X
            } else if (OP == 10) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
                R16(X)
#undef X
            } else if (OP == 11) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "+v"(p[i & 7]) : "v"(pb));
                R16(X)
#undef X
            } else if (OP == 12) {  // alternate fma / cndmask (an arithmetic and a select pipe?)
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(a[i & 7]), "+v"(a[8 + (i & 7)]) : "v"(b), "v"(c));
                R16(X)
#undef X
            } else if (OP == 13) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                R16(X)
#undef X
            } else if (OP == 14) {  // a dependent chain on one register
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                R16(X)
#undef X
            } else if (OP == 15) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R16(X)
#undef X
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)d[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char *name, int per_instr)
{
    const int iters = 2000, cus = 256;
    uint64_t *cyc;
    float *sink;
    hipMalloc(&cyc, sizeof(uint64_t) * cus * 8 * 4 * 2);
    hipMalloc(&sink, 4);
    printf("%-34s", name);
    for (int w : {1, 2, 4, 8}) {  // waves per SIMD: w blocks of 256 threads per CU
        const int blocks = cus * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, cyc, sink, 10, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(blocks * 4);
        hipMemcpy(h.data(), cyc, sizeof(uint64_t) * blocks * 4, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        // s_memtime ticks at 100 MHz on this part?  report both the tick-based and the wall-time-based figure
        const double n = (double)iters * 32 * per_instr;
        printf("  w=%d: %6.2f tick/instr/wave  %7.3f ns/instr/SIMD", w, med / n, ms * 1e6 / (n * w));
    }
    printf("\n");
    hipFree(cyc);
    hipFree(sink);
}

int main()
{
    run<0>("v_fma_f32", 1);
    run<13>("v_mul_f32", 1);
    run<14>("v_fma_f32 dependent chain", 1);
    run<1>("v_pk_fma_f32", 1);
    run<8>("v_pk_mul_f32", 1);
    run<2>("v_cndmask_b32", 1);
    run<3>("v_max3_f32", 1);
    run<15>("v_bfi_b32", 1);
    run<4>("v_mov_b32", 1);
    run<11>("v_mov_b64", 1);
    run<5>("v_rcp_f32", 1);
    run<6>("v_cmp_lt_f32", 1);
    run<7>("v_add_u32", 1);
    run<9>("v_cvt_f32_f64", 1);
    run<10>("v_add_f64", 1);
    run<12>("v_fma_f32 + v_cndmask_b32 pairs", 2);
    return 0;
}
