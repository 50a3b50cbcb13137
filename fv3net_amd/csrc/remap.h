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
};

// LEVEL_COL layout, kord <= 3, km >= 8, n_inner % 64 == 0, every row offset of a batch below 4 GiB
bool mappm_sweep_eligible(int64_t n_inner, int km, int kn, int kord, int layout, int in_dtype);
// columns [a.col0, col_end) -- whole waves -- of up to 4 fields; fast = reciprocal arithmetic (remap.hip)
void mappm_sweep_launch(const SweepArgs &a, int nf, int in_dtype, int64_t col_end, bool fast, hipStream_t st);

}  // namespace fv3hip
