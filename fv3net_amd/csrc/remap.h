// Interface between vertical.hip (the C ABI of the remap) and remap.hip (the sweep kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fv3hip {

constexpr int kSweepMaxFields = 4;

struct SweepArgs {
    const void *pe1, *pe2;
    const void *q1[kSweepMaxFields];
    float *q2[kSweepMaxFields];
    int64_t col0;     // first column of this launch (a multiple of 64)
    int64_t n_inner;  // columns per batch plane (a multiple of 64)
    int km, kn, iv;
    unsigned int *n_bad, *bad_cols;
    // target interfaces on a horizontally coarser grid (regridz.py:119-121 upsamples them; here the upsampled copy is never
    // made): pe2 is [batch][kn + 1][pe2_plane = pe2_ny * pe2_nx] and column (y, x) of rows of nx columns reads coarse column
    // (y / pe2_f, x / pe2_f) -- also right for a staggered dim, whose last point maps to the last coarse point.  pe2_f 0 / 1:
    // same grid.
    int pe2_f, nx, pe2_nx;
    int64_t pe2_plane;
};

// LEVEL_COL layout, kord <= 3, km >= 8, n_inner % 64 == 0, every row offset of a batch below 4 GiB
bool mappm_sweep_eligible(int64_t n_inner, int km, int kn, int kord, int layout, int in_dtype);
// columns [a.col0, col_end) -- whole waves -- of up to 4 fields; fast = reciprocal arithmetic (remap.hip)
void mappm_sweep_launch(const SweepArgs &a, int nf, int in_dtype, int64_t col_end, bool fast, hipStream_t st);

}  // namespace fv3hip
