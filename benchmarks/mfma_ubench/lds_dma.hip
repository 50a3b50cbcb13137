#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
__global__ __launch_bounds__(256) void k(const f32x4* w, f32x4* out, int nbytes, uint32_t ldsoff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(w), 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4* all = reinterpret_cast<f32x4*>(smem);
    for (int i = threadIdx.x; i < 160*1024/16 - 64; i += 256) all[i] = f32x4{-1.f,-1.f,-1.f,-1.f};
    __syncthreads();
    lds_char* dst = (lds_char*)smem + ldsoff + wave * 1024;
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, lane * 16, wave * 1024 + i * 4096, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // read back the 8 KB at ldsoff
    const uint32_t base = (uint32_t)(uintptr_t)(lds_char*)smem + ldsoff;
    for (int i = 0; i < 2; ++i) {
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + i * 4096 + threadIdx.x * 16) : "memory");
        out[i * 256 + threadIdx.x] = v;
    }
}
int main() {
    const int n = 512; // float4
    std::vector<float> h(n * 4);
    for (int i = 0; i < n * 4; ++i) h[i] = (float)i;
    f32x4 *dw, *dout;
    hipMalloc(&dw, n * 16); hipMalloc(&dout, n * 16);
    hipMemcpy(dw, h.data(), n * 16, hipMemcpyHostToDevice);
    const int lds = 160 * 1024 - 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (uint32_t off : {0u, 40960u, 81920u, 122880u, 147456u}) {
        hipMemset(dout, 0, n * 16);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), lds, 0, dw, dout, n * 16, off);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(n * 4);
        hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int i = 0; i < n * 4; ++i) if (o[i] != h[i]) { if (first < 0) first = i; ++bad; }
        printf("ldsoff %u: err=%d bad=%d first=%d got=%g\n", off, (int)e, bad, first, first >= 0 ? o[first] : 0.f);
    }
    return 0;
}
