// Library-level entry points of libfv3hip.so: error reporting, device selection, HIP-event timers.
#include "common.h"

namespace fv3hip {

char *last_error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace fv3hip

using namespace fv3hip;

extern "C" const char *fv3hip_last_error(void) { return last_error_buffer(); }

extern "C" int fv3hip_abi_version(void) { return FV3HIP_ABI_VERSION; }

extern "C" int fv3hip_init(int device)
{
    int n = 0;
    FV3HIP_CHECK_HIP(hipGetDeviceCount(&n));
    FV3HIP_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    // (a query only: the caller's current device is not changed.  Every other entry point works on the CURRENT device --
    // launches, scratch allocations -- so callers make the device of their pointers current around a call, as
    // fv3net_amd/_lib.py:call_on does.)
    hipDeviceProp_t prop;
    FV3HIP_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FV3HIP_EUNSUPPORTED, "libfv3hip is built for gfx950 (MI355X) only; device %d is %s",
                    device, prop.gcnArchName);
    return FV3HIP_OK;
}

extern "C" int fv3hip_device_info(fv3hip_device_info_t *out)
{
    FV3HIP_REQUIRE(out, "null pointer");
    int dev = 0;
    FV3HIP_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FV3HIP_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    memset(out, 0, sizeof(*out));
    strncpy(out->name, prop.name, sizeof(out->name) - 1);
    strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
    out->compute_units = prop.multiProcessorCount;
    out->wavefront_size = prop.warpSize;
    out->lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    out->clock_mhz = prop.clockRate / 1000;
    out->hbm_bytes = prop.totalGlobalMem;
    return FV3HIP_OK;
}

struct fv3hip_timer {
    hipEvent_t start, stop;
};

// Wavefronts that do nothing for `microseconds` each (bounded: the loop ends when the 100 MHz wall clock has advanced that far).
// For the host layer's stream calibration: how soon does a small kernel on one stream start beside a grid on another that takes
// several rounds to dispatch?  At once where the two streams sit on different dispatch pipes, after the grid where they share
// one (the runtime multiplexes streams onto four hardware queues) -- see fv3net_amd/cubedsphere/_device.py.
__global__ void spin_kernel(long long ticks)
{
    // (two clocks: the constant 100 MHz one decides; the shader clock -- at most 2.5 GHz -- bounds the loop should the first not run)
    const long long t0 = wall_clock64(), c0 = clock64();
    while (wall_clock64() - t0 < ticks && clock64() - c0 < ticks * 40) {
    }
}

extern "C" int fv3hip_spin(int64_t microseconds, int n_workgroups, void *stream)
{
    FV3HIP_REQUIRE(microseconds >= 0 && microseconds <= 100000, "spin time must be within [0, 100000] microseconds");
    FV3HIP_REQUIRE(n_workgroups >= 1 && n_workgroups <= (1 << 20), "n_workgroups must be within [1, 2^20]");
    hipLaunchKernelGGL(spin_kernel, dim3((unsigned)n_workgroups), dim3(64), 0, static_cast<hipStream_t>(stream), (long long)microseconds * 100);
    if (hipGetLastError() != hipSuccess) return fail(FV3HIP_EHIP, "spin_kernel launch failed");
    return FV3HIP_OK;
}

extern "C" int fv3hip_timer_create(fv3hip_timer_t *out)
{
    FV3HIP_REQUIRE(out, "null pointer");
    fv3hip_timer *t = new fv3hip_timer();
    if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess) {
        delete t;
        return fail(FV3HIP_EHIP, "hipEventCreate failed");
    }
    *out = t;
    return FV3HIP_OK;
}

extern "C" int fv3hip_timer_start(fv3hip_timer_t t, void *stream)
{
    FV3HIP_REQUIRE(t, "null timer");
    FV3HIP_CHECK_HIP(hipEventRecord(t->start, as_stream(stream)));
    return FV3HIP_OK;
}

extern "C" int fv3hip_timer_stop(fv3hip_timer_t t, void *stream)
{
    FV3HIP_REQUIRE(t, "null timer");
    FV3HIP_CHECK_HIP(hipEventRecord(t->stop, as_stream(stream)));
    return FV3HIP_OK;
}

extern "C" int fv3hip_timer_elapsed_ms(fv3hip_timer_t t, float *ms)
{
    FV3HIP_REQUIRE(t && ms, "null pointer");
    FV3HIP_CHECK_HIP(hipEventSynchronize(t->stop));
    FV3HIP_CHECK_HIP(hipEventElapsedTime(ms, t->start, t->stop));
    return FV3HIP_OK;
}

extern "C" int fv3hip_timer_destroy(fv3hip_timer_t t)
{
    if (!t) return FV3HIP_OK;
    hipEventDestroy(t->start);
    hipEventDestroy(t->stop);
    delete t;
    return FV3HIP_OK;
}
