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
    // MEAN (the fused remap + masked block mean, fv3hip_mappm_block_mean): a wave's 64 columns are one 8 x 8 block -- one
    // coarse column, so one target grid for the whole wave -- and `col0 / 64` is the first BLOCK of the launch.  The remapped
    // values never reach memory: target layer k of lane l is parked in LDS as area_l * q2 where lvl(k + cmp_offset) < pe1(km + 1)
    // (regridz.py:200-220), else 0, and the wave writes sum / sum-of-weights to mean[f][batch][k][block].  q2[] is then a
    // fine-size scratch that only the rows a lane has to evict from the LDS ring (and the redone blocks) pass through.
    const float *area;       // [n_batch / area_repeat][ny][nx]
    int64_t area_repeat;
    const void *lvl;         // compared pressures on the coarse grid, input dtype: [batch][cmp_levels][pe2_plane]
    int cmp_levels, cmp_offset;
    float *mean[kSweepMaxFields];
    unsigned int *bad_blocks;  // blocks with an ill-formed column (counter: n_bad[1]); bad_cols then lists all their columns
    unsigned int *rest_blocks; // (block, first row not summed) of the waves that went into spill mode (counter: n_bad[2])
};

// LEVEL_COL layout, kord <= 3, km >= 8, n_inner % 64 == 0, and the three offsets the kernel keeps in 32 bits below 4 GiB (a level
// of the inputs, a result column, the target array of a batch: `pe2_plane` = the target's plane where it lives on a coarser grid)
bool mappm_sweep_eligible(int64_t n_inner, int km, int kn, int kord, int layout, int in_dtype, int64_t pe2_plane = 0);
// columns [a.col0, col_end) -- whole waves -- of up to 4 fields; fast = reciprocal arithmetic (remap.hip)
void mappm_sweep_launch(const SweepArgs &a, int nf, int in_dtype, int64_t col_end, bool fast, hipStream_t st);
// the fused remap + masked 8 x 8 block mean: shapes it takes, the launch of blocks [a.col0 / 64, col_end / 64), and the pass that
// redoes the listed blocks from the scratch rows the sequential routine has rewritten
bool mappm_mean_eligible(int ny, int nx, int factor, int km, int kn, int kord, int in_dtype);
void mappm_mean_launch(const SweepArgs &a, int nf, int in_dtype, int64_t col_end, bool fast, hipStream_t st);
void mappm_mean_redo_launch(const SweepArgs &a, int nf, int in_dtype, hipStream_t st);
// ... and the pass that sums, from the scratch rows, what the waves that ran out of LDS ring left unsummed
void mappm_mean_rest_launch(const SweepArgs &a, int nf, int in_dtype, int64_t n_blocks, hipStream_t st);

}  // namespace fv3hip
