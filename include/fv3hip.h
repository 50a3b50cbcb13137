/*
 * fv3hip.h -- C ABI of libfv3hip.so, the MI355X (gfx950) implementation of fv3net's
 * column-wise ML tendency inference and cubed-sphere coarse-graining hot path.
 *
 * The reference has no native ABI for this path except the f2py-generated `mappm.mappm`
 * module; everything else is Python on top of TensorFlow / xarray / numpy.  Each entry point
 * below therefore cites the reference *Python or Fortran interface* it replaces (paths are
 * relative to the reference checkout).  INTEGRATION.md shows the ctypes stub a maintainer
 * of the reference would add for each one.
 *
 * Conventions
 *   - every function returns 0 on success or a negative FV3HIP_E* code; the message for the
 *     calling thread's last failure is returned by fv3hip_last_error();
 *   - all data pointers are DEVICE pointers (hipMalloc / torch.cuda tensors), caller-owned,
 *     never freed or retained by the library beyond the call (model handles copy their
 *     weights at create time);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only
 *     enqueues work on that stream and never synchronises, so calls can be captured in a
 *     hipGraph;
 *   - arrays are dense, row-major, "x fastest": horizontal fields are [n_outer][ny][nx].
 */
#ifndef FV3HIP_H
#define FV3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FV3HIP_ABI_VERSION 3

/* status codes */
#define FV3HIP_OK 0
#define FV3HIP_EINVAL -1      /* bad argument (shape, dtype, factor ...)            */
#define FV3HIP_EUNSUPPORTED -2 /* valid in the reference but not implemented here    */
#define FV3HIP_EHIP -3        /* a HIP runtime call failed                           */
#define FV3HIP_ENOMEM -4

/* element types */
#define FV3HIP_F32 0
#define FV3HIP_F64 1
#define FV3HIP_I32 2
#define FV3HIP_I64 3

/* block reductions (fv3hip_block_reduce) */
#define FV3HIP_OP_SUM 0
#define FV3HIP_OP_MEAN 1
#define FV3HIP_OP_MIN 2
#define FV3HIP_OP_MAX 3
#define FV3HIP_OP_MEDIAN 4
#define FV3HIP_OP_MODE 5

/* NaN handling for fv3hip_block_reduce */
#define FV3HIP_NAN_SKIP 0      /* xarray coarsen().sum()/min()/max()/mean() default (skipna) */
#define FV3HIP_NAN_PROPAGATE 1 /* numpy.sum/mean/min/max/median (NaN if any NaN in the window);
                                  scipy.stats.mode(nan_policy="propagate")                   */
#define FV3HIP_NAN_OMIT 2      /* scipy.stats.mode(nan_policy="omit")                        */

/* arithmetic of the remap (fv3hip_mappm, fv3hip_mappm_multi) */
#define FV3HIP_ARITH_EXACT 0 /* IEEE division, the Fortran's association order: bit-identical to the compiled reference */
#define FV3HIP_ARITH_FAST 1  /* a / b = a * v_rcp_f32(b) with shared reciprocals where the sweep kernel applies (LEVEL_COL,
                                kord <= 3, km >= 8, n_inner % 64 == 0); a few ulp from EXACT; elsewhere the same as EXACT */

/* column layouts (fv3hip_mappm, fv3hip_pressure_at_interface) */
#define FV3HIP_LAYOUT_COL_LEVEL 0 /* [column][level]: level fastest (what f2py callers pass)  */
#define FV3HIP_LAYOUT_LEVEL_COL 1 /* [batch][level][inner]: column fastest (native [z,y,x])   */

const char *fv3hip_last_error(void);
int fv3hip_abi_version(void);

/* Check that `device` exists and is a gfx950 part.  The calling thread's current device is NOT changed.
 * DEVICE RULE for every other entry point: kernels are launched on, and scratch memory is allocated on, the CURRENT
 * HIP device, which must be the device of the pointers passed (callers holding several GPUs make it current around
 * the call; fv3hip_mlp_predict returns FV3HIP_EINVAL when the model was created on another device). */
int fv3hip_init(int device);

typedef struct {
    char name[128];
    char arch[64];
    int compute_units;
    int wavefront_size;
    int lds_bytes_per_cu;
    int clock_mhz;
    size_t hbm_bytes;
} fv3hip_device_info_t;
int fv3hip_device_info(fv3hip_device_info_t *out);

/* ------------------------------------------------------------------------------------------
 * Horizontal coarse-graining
 * ------------------------------------------------------------------------------------------ */

/*
 * Replaces vcm.cubedsphere.weighted_block_average
 * (external/vcm/vcm/cubedsphere/coarsen.py:183-218):
 *     out = nansum_block(obj * w) / nansum_block(w)      over factor x factor blocks
 * obj: [n_outer][ny][nx]; weights: [n_outer / w_repeat][ny][nx], each weight slice shared by
 * w_repeat consecutive outer slices (w_repeat = 1: same shape as obj; w_repeat = nz: 2-D
 * area weights of a [tile][z][y][x] field).  out: [n_outer][ny/factor][nx/factor] in the
 * promoted type (F64 if either input is F64, else F32).  NaN products/weights are skipped as
 * xarray's skipna sums do; an all-zero denominator gives NaN (0/0) as in the reference.
 */
int fv3hip_weighted_block_average(const void *obj, int obj_dtype, const void *weights,
                                  int w_dtype, int64_t n_outer, int ny, int nx,
                                  int64_t w_repeat, int factor, void *out, void *stream);

/*
 * The mass-weighted means of the restart pipelines (external/vcm/vcm/cubedsphere/coarsen_restarts.py:335-427, 856-900:
 * weighted_block_average(ds[mass_weighted_vars], delp * area, ...)) for n_fields fields that share their weights:
 *     out_f[o][Y][X] = nansum_block(field_f * (delp * area)) / nansum_block(delp * area)
 * with the product delp * area formed in registers (never written) and read once per four fields.  `fields` / `outs` are
 * HOST arrays of device pointers; fields and delp [n_outer][ny][nx] in `dtype`, area [n_outer / a_repeat][ny][nx] in
 * `area_dtype`, outputs [n_outer][ny / f][nx / f] in the promoted type (F64 unless both are F32).  `delp` may be NULL: the
 * weights are then `area` alone, shared by the fields (the area-weighted means of the surface data,
 * coarsen_restarts.py:1163-1230, four fields per launch).  factor in
 * {2, 4, 8, 16} with 16-byte aligned rows; anything else returns FV3HIP_EUNSUPPORTED (callers then form the product with
 * fv3hip_ew and use fv3hip_weighted_block_average).
 */
int fv3hip_mass_weighted_block_average(const void *const *fields, int n_fields, int dtype, const void *delp,
                                       const void *area, int area_dtype, int64_t n_outer, int ny, int nx,
                                       int64_t a_repeat, int factor, void *const *outs, void *stream);

/*
 * Replaces vcm.cubedsphere.edge_weighted_block_average
 * (external/vcm/vcm/cubedsphere/coarsen.py:221-273).  edge = 0 ('x'): weighted mean over
 * `factor` cells along x, every factor-th row kept along y (out [n_outer][ceil(ny/f)][nx/f]);
 * edge = 1 ('y'): the transpose of that (out [n_outer][ny/f][ceil(nx/f)]).
 */
/* The general form of the two weighted means above: windows of by x bx cells every (sy, sx) cells,
 * out [n_outer][(ny - by) / sy + 1][(nx - bx) / sx + 1] (weights as in fv3hip_weighted_block_average; ABI v3).
 * weighted_block_average = (f, f, f, f); edge_weighted 'x' = (1, f, f, f); on fields already reduced to the kept lines: (1, f, 1, f). */
int fv3hip_weighted_window_average(const void *obj, int obj_dtype, const void *weights, int w_dtype,
                                   int64_t n_outer, int ny, int nx, int64_t w_repeat, int by, int bx,
                                   int sy, int sx, void *out, void *stream);
int fv3hip_edge_weighted_block_average(const void *obj, int obj_dtype, const void *spacing,
                                       int w_dtype, int64_t n_outer, int ny, int nx,
                                       int64_t w_repeat, int factor, int edge, void *out,
                                       void *stream);

/*
 * Replaces vcm.cubedsphere.block_coarsen / block_median / _block_mode / block_edge_coarsen
 * (external/vcm/vcm/cubedsphere/coarsen.py:795-840, 557-588, 750-786, 629-683) and the
 * vendored block_reduce they sit on (external/vcm/vcm/cubedsphere/_skimage.py:125-202):
 *     out[o][Y][X] = op over in[o][Y*sy .. Y*sy+by-1][X*sx .. X*sx+bx-1]
 * with ny_out = (ny - by)/sy + 1, nx_out = (nx - bx)/sx + 1.  Full blocks: by=bx=sy=sx=f;
 * edge 'x' coarsening: by=1, bx=f, sy=sx=f.  The output has the input's dtype except MEAN /
 * MEDIAN of integers, which is not supported (the reference only applies them to floats).
 */
int fv3hip_block_reduce(const void *in, int dtype, int64_t n_outer, int ny, int nx, int by,
                        int bx, int sy, int sx, int op, int nan_policy, void *out,
                        void *stream);

/*
 * Replaces vcm.cubedsphere.block_upsample / block_upsample_like
 * (external/vcm/vcm/cubedsphere/coarsen.py:869-938): out[o][y][x] = in[o][y/f][x/f], where a
 * coarse dimension of odd size is a staggered one whose last point is not repeated
 * (ny_out = (ny_in-1)*f+1) and an even one is repeated uniformly (ny_out = ny_in*f).
 * elem_size is 4 or 8 bytes.
 */
/* out[o][y][x] = in[o][y / fy][x / fx], out is [n_outer][ny_in * fy][nx_in * fx] (xarray_utils.repeat, vcm/xarray_utils.py:37-82,
 * with a count per horizontal dim; ABI v3) */
int fv3hip_repeat(const void *in, int elem_size, int64_t n_outer, int ny_in, int nx_in, int fy,
                  int fx, void *out, void *stream);
int fv3hip_block_upsample(const void *in, int elem_size, int64_t n_outer, int ny_in, int nx_in,
                          int factor, void *out, void *stream);

/*
 * The mask arithmetic of the 'complex' surface-data coarse-graining
 * (external/vcm/vcm/cubedsphere/coarsen_restarts.py:1140-1470), one elementwise launch per step.
 * a, out: n values; b, c: n values, or 2-D fields of `inner` values shared by b_rep / c_rep consecutive
 * outer slices of a (a = [.., level, y, x], b = [.., y, x]).  Masks are 0 / 1 in the fields' dtype.
 */
#define FV3HIP_EW_MUL 0        /* a * b                                              */
#define FV3HIP_EW_ISCLOSE 1    /* np.isclose(a, b) (rtol 1e-5, atol 1e-8)            */
#define FV3HIP_EW_ISCLOSE_S 2  /* np.isclose(a, scalar)                              */
#define FV3HIP_EW_WHERE_NAN 3  /* a.where(b): a where the mask b holds, else NaN     */
#define FV3HIP_EW_SELECT 4     /* xr.where(c, a, b)                                  */
#define FV3HIP_EW_SELECT_S 5   /* xr.where(b, scalar, a)                             */
#define FV3HIP_EW_GT_S 6       /* a > scalar                                         */
#define FV3HIP_EW_LT_S 7       /* a < scalar                                         */
#define FV3HIP_EW_FILLNA_S 8   /* a.fillna(scalar)                                   */
#define FV3HIP_EW_AND 9        /* a & b                                              */
#define FV3HIP_EW_MIN_S 10     /* a.where(a < scalar, other=scalar)                  */
#define FV3HIP_EW_BLEND 11     /* coarsen_restarts.blend: a * b + (1 - a) * c         */
#define FV3HIP_EW_MUL_S 12     /* scalar * a                                          */
#define FV3HIP_EW_WHERE_S 13   /* a.where(mask b, other=scalar)                       */
#define FV3HIP_EW_ADD 14       /* a + b                                               */
#define FV3HIP_EW_ADD_S 15     /* a + scalar                                          */
/* the emulators' tensor transforms (fv3fit/emulation/transforms/transforms.py:17-158; ABI v3) */
#define FV3HIP_EW_SUB 16               /* a - b                                (Difference.forward)               */
#define FV3HIP_EW_LOG_FLOOR_S 17       /* log(max(a, scalar))                  (LogTransform.forward)             */
#define FV3HIP_EW_EXP 18               /* exp(a)                               (LogTransform.backward)            */
#define FV3HIP_EW_RELU_THRESHOLD_S 19  /* a where a > scalar else 0            (LimitValueTransform, lower bound) */
#define FV3HIP_EW_BELOW_S 20           /* a where a < scalar else 0            (LimitValueTransform, upper bound) */
#define FV3HIP_EW_DIV_S 21             /* a / scalar                           (vcm temperature_tendency, thermo/local.py:340-358) */
#define FV3HIP_EW_INCLOUD_TO_GRIDCELL 22 /* b where a <= 1e-3 else b * max-like(a, 5e-2): in-cloud -> gridcell condensate by cloud
                                          fraction a (vcm/calc/clouds.py:40-66) */
#define FV3HIP_EW_CLIP01 23            /* np.clip(a, 0, 1), a NaN stays        (taper_ramp, fv3fit/_shared/taper_function.py:41-51) */
#define FV3HIP_EW_POW_BASE_S 24        /* scalar ** a                          (taper_decay, taper_function.py:54-64) */
#define FV3HIP_EW_MINIMUM_S 25         /* np.minimum(a, scalar), a NaN stays   (taper_decay) */
/* ... and of the derived variables of vcm.DerivedMapping (external/vcm/vcm/derived_mapping.py) */
#define FV3HIP_EW_DIV 26               /* a / b                                (flux fractions, transmissivity :247-262) */
#define FV3HIP_EW_WHERE_POS_S 27       /* a where b > 0 else scalar            (_limit_sw_positive :243-244) */
#define FV3HIP_EW_SIGN 28              /* np.sign(a)                           (tendencies parallel to the wind :167-176) */
#define FV3HIP_EW_ABS 29               /* |a| */
#define FV3HIP_EW_RSUB_S 30            /* scalar - a                           (1 - fraction :293-295) */
#define FV3HIP_EW_RDIV_S 31            /* scalar / a                           (gridcell_to_incloud_condensate, calc/clouds.py:33) */
#define FV3HIP_EW_WHERE_GT_S 32        /* a.where(a > scalar, other=scalar)    (calc/clouds.py:33) */
#define FV3HIP_EW_LE_S 33              /* a <= scalar                          (calc/clouds.py:34-36) */
#define FV3HIP_EW_SIN 34               /* sin(a)                               (cos_zenith_angle, calc/_zenith_angle.py:226-242) */
#define FV3HIP_EW_COS 35               /* cos(a) */
int fv3hip_ew(int op, const void *a, const void *b, const void *c, double scalar, int dtype,
              int64_t n, int64_t inner, int64_t b_rep, int64_t c_rep, void *out, void *stream);

/*
 * Pick and orient the halo vectors of `n_local` tiles from the boundary-vector table of all six tiles (fv3hip_cube_edge_rows,
 * [6][4][n_mid][n]): out[side][i][o][j] = rows[nbr[side * n_local + i]][row[...]][o][flip[...] ? n - 1 - j : j], side 0 = the
 * neighbour below the first cell, 1 = beyond the last.  The connectivity of external/vcm/vcm/cubedsphere/xgcm.py:7-34
 * is the caller's (fv3net_amd/cubedsphere/grid.py); nbr / row / flip are HOST arrays of 2 * n_local ints.
 */
int fv3hip_halo_pick(const void *rows, int elem_size, int n_local, int64_t n_mid, int n, const int *nbr, const int *row,
                     const int *flip, void *out, void *stream);

/* out[i] = (out_dtype) in[i]: F32 / F64 / I32 / I64 -> F32 / F64 (the surface-data arithmetic of
 * coarsen_restarts.py:1140-1470 runs in one dtype; restart files mix them). */
int fv3hip_cast(const void *in, int in_dtype, void *out, int out_dtype, int64_t n, void *stream);
/* The same for `count` arrays in one launch (the 35 surface fields of a restart set): HOST arrays of device pointers,
 * input dtypes and lengths. */
int fv3hip_cast_many(const void *const *in, const int *in_dtype, void *const *out, int out_dtype, const int64_t *n, int count,
                     void *stream);

/*
 * Cell centres -> cell edges across the faces of the cube: the device half of what
 * xgcm.Grid.interp(delp, axis) does for vcm.cubedsphere.regridz.regrid_to_edge_weighted_pressure and
 * coarsen_restarts.compute_edge_delp (external/vcm/vcm/cubedsphere/regridz.py:123-135,
 * coarsen_restarts.py:825-853, connectivity table xgcm.py:7-34).
 *
 * fv3hip_cube_edge_rows: in [n_tiles][n_mid][n][n] -> rows [n_tiles][4][n_mid][n], the four
 * boundary vectors of every (square) tile indexed along the edge: 0: x = 0, 1: x = n-1 (index = y),
 * 2: y = 0, 3: y = n-1 (index = x).  These are what neighbouring tiles -- on this or another
 * GPU -- need as their one-cell halo (the only exchange step of the coarse-graining path).
 * elem_size is 4 or 8 bytes.
 *
 * fv3hip_interp_center_to_outer: out = 0.5 * (left + right) along axis (0 = x: [n_outer][ny][nx+1],
 * 1 = y: [n_outer][ny+1][nx]); at the two ends the outside neighbour comes from lo / hi
 * [n_outer][ny] (axis 0) or [n_outer][nx] (axis 1), already oriented along this tile's edge.
 */
int fv3hip_cube_edge_rows(const void *in, int elem_size, int n_tiles, int64_t n_mid, int n,
                          void *rows, void *stream);
int fv3hip_interp_center_to_outer(const void *in, int dtype, int64_t n_outer, int ny, int nx,
                                  int axis, const void *lo, const void *hi, void *out,
                                  void *stream);
/* ... only every step-th edge along the axis (n / step + 1 points; point j' is edge j' * step): the lines
 * vcm.cubedsphere.edge_weighted_block_average (coarsen.py:221-273) keeps -- the pressure-level D-grid wind path
 * (regridz.py:81-146, coarsen_restarts.py:497-556) needs the edge pressures on those lines only (ABI v3). */
int fv3hip_interp_center_to_outer_lines(const void *in, int dtype, int64_t n_outer, int ny, int nx,
                                        int axis, int step, const void *lo, const void *hi,
                                        void *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Vertical: interface pressures and the PPM remap
 * ------------------------------------------------------------------------------------------ */

/*
 * Replaces vcm.pressure_at_interface
 * (external/vcm/vcm/calc/thermo/vertically_dependent.py:41-66): p[0] = toa,
 * p[k+1] = p[k] + delp[k], accumulated sequentially in the array's own dtype (as
 * numpy.cumsum does).  delp: [n_batch][nz][n_inner] -> out: [n_batch][nz+1][n_inner]
 * (LEVEL_COL) or [ncol][nz] -> [ncol][nz+1] (COL_LEVEL, n_batch = ncol, n_inner = 1).
 */
int fv3hip_pressure_at_interface(const void *delp, int dtype, int64_t n_batch, int nz,
                                 int64_t n_inner, double toa_pressure, void *out, void *stream);

/*
 * Replaces vcm.pressure_at_midpoint_log
 * (external/vcm/vcm/calc/thermo/vertically_dependent.py:153-179): delp / diff(log(p_interface)).
 * Same layout and dtype rules as fv3hip_pressure_at_interface; out has nz levels.
 */
int fv3hip_pressure_at_midpoint_log(const void *delp, int dtype, int64_t n_batch, int nz,
                                    int64_t n_inner, double toa_pressure, void *out, void *stream);

/*
 * Replaces vcm.cubedsphere.regridz._mask_weights
 * (external/vcm/vcm/cubedsphere/regridz.py:200-220):
 *     out[b][k][c] = weights[b / w_repeat][c]  if p_cmp[b][k + cmp_offset][c] < p_fine[b][nz][c] else 0
 * extrapolate=False: p_cmp = coarse interface pressures (nz+1 levels), cmp_offset = 1 (bottom
 * interface of layer k); extrapolate=True: p_cmp = coarse midpoint pressures (nz levels),
 * cmp_offset = 0.  p_cmp is [n_batch][cmp_levels][n_inner], p_fine [n_batch][nz+1][n_inner],
 * both in p_dtype; weights [n_batch / w_repeat][n_inner] and out [n_batch][nz][n_inner] in w_dtype.
 */
int fv3hip_mask_weights(const void *weights, int w_dtype, const void *p_cmp, int cmp_levels,
                        int cmp_offset, const void *p_fine, int p_dtype, int64_t n_batch, int nz,
                        int64_t n_inner, int64_t w_repeat, void *out, void *stream);

/* fv3hip_mask_weights with p_cmp on a horizontally coarser grid, [n_batch][cmp_levels][ny / factor][nx / factor]: column (y, x)
 * compares the level of coarse column (y / factor, x / factor) -- regridz.py:119-121 upsamples it first; this does not.
 * FV3HIP_EUNSUPPORTED unless nx % 4 == 0 and n_batch * nz <= 65535 (upsample and call fv3hip_mask_weights then). */
int fv3hip_mask_weights_coarse(const void *weights, int w_dtype, const void *p_cmp_coarse, int cmp_levels, int cmp_offset,
                               const void *p_fine, int p_dtype, int64_t n_batch, int nz, int ny, int nx, int factor,
                               int64_t w_repeat, void *out, void *stream);

/*
 * Replaces the f2py module function mappm.mappm(p_in, f_in, p_out, i1, i2, iv, kord, ptop)
 * (external/mappm/mappm/mappm.f90:10-126 with ppm_profile :614-851 and ppm_limiters
 * :854-931; caller external/vcm/vcm/cubedsphere/regridz.py:304-338).  Single precision
 * inside, as in the Fortran (default REAL); inputs may be F32 or F64 (F64 is rounded to F32
 * on load, which is what f2py's argument conversion does).  q2 is always F32.
 *   COL_LEVEL: pe1 [ncol][km+1], q1 [ncol][km], pe2 [ncol][kn+1], q2 [ncol][kn]
 *              (n_batch = ncol, n_inner = 1)
 *   LEVEL_COL: pe1 [n_batch][km+1][n_inner], ... ; ncol = n_batch * n_inner
 * Every kord: ppm_profile (kord <= 7) and cs_profile / cs_limiters (kord > 7, mappm.f90:132-611: schemes 8 .. 16 and the linear
 * one above); kord > 7 with iv = -2 returns FV3HIP_EUNSUPPORTED (cs_profile would read the array qs that mappm never sets).
 * `arith`: FV3HIP_ARITH_EXACT or FV3HIP_ARITH_FAST (see above).
 * `workspace` must hold fv3hip_mappm_workspace_bytes(...) bytes of device memory; it carries the call's
 * list of columns to redo, so concurrent calls (other streams) need workspaces of their own.
 */
size_t fv3hip_mappm_workspace_bytes(int64_t ncol, int km);
int fv3hip_mappm(const void *pe1, const void *q1, const void *pe2, int in_dtype, float *q2,
                 int64_t n_batch, int64_t n_inner, int km, int kn, int iv, int kord, int layout, int arith,
                 void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same remap for n_fields fields that share pe1 and pe2 (regridz.py:163-185 remaps every variable of a
 * dataset between the same two pressure grids): q1 / q2 are HOST arrays of n_fields device pointers.  Fields are
 * processed four at a time in one sweep that computes the control flow and every pressure-only term once;
 * each field's result is bit-identical to fv3hip_mappm on that field.  Same workspace as fv3hip_mappm.
 */
int fv3hip_mappm_multi(const void *pe1, const void *const *q1, const void *pe2, int in_dtype,
                       float *const *q2, int n_fields, int64_t n_batch, int64_t n_inner, int km,
                       int kn, int iv, int kord, int layout, int arith, void *workspace,
                       size_t workspace_bytes, void *stream);

/*
 * fv3hip_mappm_multi with the target interfaces on a horizontally coarser grid: pe2_coarse is [n_batch][kn + 1][ny / factor]
 * [nx / factor], the fields and pe1 [n_batch][level][ny][nx]; fine column (y, x) is remapped to the interfaces of coarse column
 * (y / factor, x / factor).  regridz.py:119-185 reaches the same result through an upsampled copy of the coarse pressures
 * (block_upsample_like); here that copy is never made and the 64 columns of a wave share 64 / factor target columns.  Results are
 * identical to upsampling and calling fv3hip_mappm_multi.  FV3HIP_EUNSUPPORTED unless factor >= 2, nx % 64 == 0 and the shape is
 * one the sweep kernel takes (kord <= 3, km >= 8, ny * nx % 64 == 0); same workspace as fv3hip_mappm for n_batch * ny * nx columns.
 */
int fv3hip_mappm_multi_coarse_target(const void *pe1, const void *const *q1, const void *pe2_coarse, int in_dtype, float *const *q2,
                                     int n_fields, int64_t n_batch, int ny, int nx, int factor, int km, int kn, int iv, int kord,
                                     int arith, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Vertical remap to the coarse grid's pressure levels and the masked 8 x 8 block mean of the result in one pass -- what the
 * pressure-level restart pipelines do with every cell-centred field (coarsen_restarts.py:483-495, 940-961:
 * weighted_block_average(*regrid_to_area_weighted_pressure(ds, delp, area, ...)); regridz.py:149-220):
 *   mean[f][b][k][Y][X] = sum_block(q2 * w) / sum_block(w),   q2 = mappm of q1[f] onto the interfaces pe2_coarse[b][.][Y][X],
 *   w[y][x] = area[b / area_repeat][y][x]  where  level_coarse[b][k + cmp_offset][Y][X] < pe1[b][km][y][x] (the fine surface), else 0
 * (cmp_offset 1 with level_coarse = pe2_coarse, cmp_levels = kn + 1: the layer's bottom interface, extrapolate=False;
 * cmp_offset 0 with the coarse midpoint pressures, cmp_levels = kn: extrapolate=True).  pe1 / q1[f] are [n_batch][km(+1)][ny][nx],
 * pe2_coarse / level_coarse [n_batch][levels][ny / 8][nx / 8] in the input dtype, area float32.  One wavefront owns one block;
 * the remapped values are summed in LDS in the order of fv3hip_weighted_block_average and never written: the means are
 * bit-identical to fv3hip_mappm_multi_coarse_target + fv3hip_mask_weights_coarse + fv3hip_weighted_block_average in the same
 * `arith` (blocks with an ill-formed column -- NaN or non-monotone pressures -- are redone through the sequential routine, i.e.
 * in EXACT arithmetic).  scratch[f]: [n_batch][kn][ny][nx] float32 each, touched only by values a lane has to park outside
 * LDS and by redone blocks.  q1 / scratch / mean are HOST arrays of n_fields device pointers.
 * FV3HIP_EUNSUPPORTED unless factor == 8, ny % 8 == 0, nx % 8 == 0, kord <= 3, km >= 8, kn + 1 <= 128.
 */
size_t fv3hip_mappm_block_mean_workspace_bytes(int64_t ncol, int km);
int fv3hip_mappm_block_mean(const void *pe1, const void *const *q1, const void *pe2_coarse, const void *level_coarse,
                            int cmp_levels, int cmp_offset, int in_dtype, const float *area, int64_t area_repeat,
                            float *const *scratch, float *const *mean, int n_fields, int64_t n_batch, int ny, int nx, int factor,
                            int km, int kn, int iv, int kord, int arith, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Column MLP (fv3fit dense model / Zhao-Carr microphysics emulator)
 * ------------------------------------------------------------------------------------------ */

#define FV3HIP_TRANSFORM_NONE 0
#define FV3HIP_TRANSFORM_LOG 1 /* log(max(x, eps)) : fv3fit/emulation/transforms/transforms.py:111-129 */

#define FV3HIP_ACT_LINEAR 0
#define FV3HIP_ACT_RELU 1

/*
 * Description of the fused predict graph of
 *   fv3fit.keras._models.dense.build_model (external/fv3fit/fv3fit/keras/_models/dense.py:239-310)
 *   fv3fit.emulation MicrophysicsConfig.build with the "dense" architecture
 *     (external/fv3fit/fv3fit/emulation/models/microphysics.py:123-139,
 *      layers/architecture.py:27-50,228-282,304-343, layers/fields.py:33-66)
 * i.e.  x_i -> (optional log) -> clip slice -> (x - center) / scale -> concat ->
 *       [Dense + activation] * n_hidden -> Dense per output -> y * out_scale + out_center ->
 *       limit to [out_min, out_max] -> multiply by 0/1 level mask
 *       -> optional residual outputs  after = before + difference.
 * All arrays are HOST pointers, copied (and re-laid-out for the MFMA kernel) at create time.
 */
typedef struct {
    /* inputs */
    int n_sources;           /* number of distinct arrays the caller passes to predict()     */
    int n_inputs;            /* network input variables, concatenated in this order          */
    const int *in_source;    /* [n_inputs] which source array feeds input i                  */
    const int *in_feat_start;/* [n_inputs] first feature (level) of the source that is used  */
    const int *in_nfeat;     /* [n_inputs] number of features used (K = sum)                 */
    const int *in_transform; /* [n_inputs] FV3HIP_TRANSFORM_*                                */
    const float *in_eps;     /* [n_inputs] epsilon of the log transform                      */
    const float *in_center;  /* [K] subtracted                                               */
    const float *in_scale;   /* [K] divided by (the reference's scale + epsilon, in f32)     */
    /* hidden layers: n_hidden dense layers of `width` units, Keras kernels [in][out]        */
    int n_hidden;
    int width;
    int hidden_activation;   /* FV3HIP_ACT_* */
    const float *const *hidden_kernels; /* [n_hidden] -> [in][width] row-major               */
    const float *const *hidden_biases;  /* [n_hidden] -> [width]                             */
    /* output heads, concatenated: F = sum(out_nfeat) */
    int n_outputs;
    const int *out_nfeat;    /* [n_outputs] */
    const float *out_kernel; /* [width (or K if n_hidden == 0)][F] row-major                 */
    const float *out_bias;   /* [F] */
    const float *out_scale;  /* [F] y = yhat * scale + center                                */
    const float *out_center; /* [F] */
    const float *out_min;    /* [F] or NULL; -inf = no lower limit                           */
    const float *out_max;    /* [F] or NULL; +inf = no upper limit                           */
    const float *out_mask;   /* [F] of 0/1 or NULL (ClipConfig.zero_mask_clipped_layer)      */
    /* residual outputs: derived d = source[res_source[d]] + output[res_output[d]]           */
    int n_residual;
    const int *res_source;   /* [n_residual] */
    const int *res_output;   /* [n_residual] */
    /* 1: the activations of the last hidden layer ([width] features) are an output of their own, stored
     * to the array passed after the outputs and the residual outputs (n_outputs may then be 0).  The
     * cell of a recurrent layer -- tf.keras.layers.SimpleRNN, relu([x_t, h_{t-1}] [W; U] + b)
     * (external/fv3fit/fv3fit/emulation/layers/architecture.py:186-191) -- is such a model. */
    int hidden_output;
} fv3hip_mlp_desc_t;

typedef struct fv3hip_mlp *fv3hip_mlp_t;

int fv3hip_mlp_create(const fv3hip_mlp_desc_t *desc, fv3hip_mlp_t *out);
int fv3hip_mlp_destroy(fv3hip_mlp_t model);

/*
 * Replaces the Keras call inside fv3fit._shared.xr_prediction._predict
 * (external/fv3fit/fv3fit/_shared/xr_prediction.py:75-108: `model(inputs)`) and inside
 * emulation.models.ModelWithClassifier.__call__ (external/emulation/emulation/models.py:37-53:
 * `model.predict(inputs, batch_size=...)`).
 *   sources[s]: device pointer to source array s, dtype src_dtype[s] (F32/F64), element
 *               (feature f, sample n) at  f * src_feat_stride[s] + n * src_sample_stride[s]
 *               (so both [feature][sample] and [sample][feature] are accepted without a copy);
 *   outputs[j]: n_outputs + n_residual device pointers, dtype out_dtype (F32 or F64),
 *               element (f, n) at f * out_feat_stride[j] + n * out_sample_stride[j].
 */
int fv3hip_mlp_predict(fv3hip_mlp_t model, const void *const *sources, const int *src_dtype,
                       const int64_t *src_feat_stride, const int64_t *src_sample_stride,
                       int64_t n_samples, void *const *outputs, int out_dtype,
                       const int64_t *out_feat_stride, const int64_t *out_sample_stride,
                       void *stream);

/* FLOPs of the dense contraction per sample (2 * sum(in * out)), for roofline accounting. */
int64_t fv3hip_mlp_flops_per_sample(fv3hip_mlp_t model);
/* Calls of at most max_samples samples go to the feature-split kernel for small sample counts (mlp_small_kernel: 32 samples
 * per workgroup, a layer's output features split over its waves -- four times the CUs of the 128-sample tiles, a quarter
 * of their latency; same graph and arithmetic class, contraction in plain feature order).  -1 (default): while one round of
 * such workgroups covers the call (32 x CUs samples; FV3HIP_MLP_SMALL_MAX_SAMPLES overrides per process); 0: never.
 * Results of the two kernels agree to float32 rounding, not bit for bit: pin the limit where runs on different domain
 * decompositions must reproduce each other exactly (ABI v3). */
int fv3hip_mlp_set_small_limit(fv3hip_mlp_t model, int64_t max_samples);

/* Diagnostic: the kernel instantiation the model's last fv3hip_mlp_predict launched and the epilogue its full
 * sample tiles took, e.g. "mlp_fused_kernel<8,false,true,false,false,false> epilogue=residual" (the name rocprofv3
 * reports, so that tests and bench.py can tell which code path a result came from).  "" before the first call.
 * The string lives in the model handle. */
const char *fv3hip_mlp_last_variant(fv3hip_mlp_t model);

/*
 * EXPERIMENTAL: the same predict graph with the contraction on the bf16 matrix cores, every fp32 operand split into three
 * bf16 pieces and six cross products accumulated in fp32 (fv3net_amd/csrc/mlp_bf16x3.hip).  Same
 * descriptor as fv3hip_mlp_create; restrictions: hidden width 256, ReLU, no hidden output, no limits / masks, log epsilons
 * >= FLT_MIN, 3 / 5 / 13 output tiles of 32.  Sources and outputs are float32 [feature][sample] with unit sample stride
 * (element (f, n) at f * feat_stride + n).  Not a replacement of fv3hip_mlp_predict: the fp32 kernel is the product path.
 */
typedef struct fv3hip_mlp3 *fv3hip_mlp3_t;
int fv3hip_mlp3_create(const fv3hip_mlp_desc_t *desc, fv3hip_mlp3_t *out);
int fv3hip_mlp3_destroy(fv3hip_mlp3_t model);
int64_t fv3hip_mlp3_flops_per_sample(fv3hip_mlp3_t model);
int fv3hip_mlp3_predict(fv3hip_mlp3_t model, const void *const *sources, const int64_t *src_feat_stride, int64_t n_samples,
                        void *const *outputs, const int64_t *out_feat_stride, void *stream);

/* `n_workgroups` idle wavefronts of `microseconds` (<= 100000) each on `stream`: the host layer times a small one beside a large one
 * on another stream to learn which streams the runtime lets run side by side (cubedsphere/_device.py: the pipelines' side streams). */
int fv3hip_spin(int64_t microseconds, int n_workgroups, void *stream);

/* ------------------------------------------------------------------------------------------
 * Timing helper: HIP events on the caller's stream (bench.py measures kernels with these
 * because torch.cuda.Event only sees torch's current stream).
 * ------------------------------------------------------------------------------------------ */
typedef struct fv3hip_timer *fv3hip_timer_t;
int fv3hip_timer_create(fv3hip_timer_t *out);
int fv3hip_timer_start(fv3hip_timer_t t, void *stream);
int fv3hip_timer_stop(fv3hip_timer_t t, void *stream);
int fv3hip_timer_elapsed_ms(fv3hip_timer_t t, float *ms); /* synchronises on the stop event */
int fv3hip_timer_destroy(fv3hip_timer_t t);

/*
 * Column helpers of the restart pipelines coarsen_restarts_on_pressure / _via_blended_method
 * (external/vcm/vcm/cubedsphere/coarsen_restarts.py:559-676, 990-1017).  Arrays are
 * [n_batch][nz][n_inner]; per-column results [n_batch][n_inner].
 *   fv3hip_column_sum          surface_pressure_from_delp (vertically_dependent.py:189-208): sum_k x + addend
 *   fv3hip_blend_weights       compute_blending_weights (coarsen_restarts.py:559-576)
 *   fv3hip_hydrostatic_balance _impose_hydrostatic_balance (coarsen_restarts.py:990-1017 with
 *                              height_at_interface, hydrostatic_dz, dz_and_top_to_phis,
 *                              vertically_dependent.py:69-99, 182-186, 211-235)
 */
int fv3hip_column_sum(const void *x, int dtype, int64_t n_batch, int nz, int64_t n_inner,
                      double addend, void *out, void *stream);
int fv3hip_blend_weights(const void *blending_pressure, const void *ps_coarse,
                         const void *pfull_coarse, int dtype, int64_t n_batch, int nz,
                         int64_t n_inner, void *out, void *stream);
int fv3hip_hydrostatic_balance(const void *dz, const void *phis, const void *t, const void *q,
                               const void *delp, int dtype, int64_t n_batch, int nz,
                               int64_t n_inner, double toa_pressure, void *dz_out,
                               void *phis_out, void *stream);

/*
 * Pre- and post-passes of the "dense-local" emulators (one MLP shared by all levels, a (level, column)
 * point is one sample of fv3hip_mlp_predict on the packed [n_inputs][nz * ncol] array) and the
 * classifier decode of emulation.models.ModelWithClassifier.  State arrays are [nz][ncol]
 * (call_py_fort's [feature, sample]) or, with has_levels = 0, [ncol]; tables are DEVICE float arrays.
 *   fv3hip_local_pack      one network input row: (t(x) - center[z]) / scale[z], t = identity or
 *                          log(max(x, eps)); float64 sources are cast to float32 first, as Keras does
 *                          (fv3fit/emulation/layers/architecture.py:53-75 combine_sequence_inputs,
 *                          layers/fields.py:6-41 FieldInput, transforms/transforms.py:111-129)
 *   fv3hip_local_unpack    one network output channel (layers/fields.py:44-66 FieldOutput, then the
 *                          backward transforms in the reference's order, transforms.py:219-224, 55-58):
 *                            direct   = yhat * scale[z] + center[z]            (scale/center NULL = 1/0)
 *                            unscaled = direct * max(cs_scale[b], min_scale) + cs_center[b],
 *                                       b = max(upper_bound(edges[0..n_bins), cond_on) - 1, 0)
 *                                       (keras/math.py:5-23 piecewise)          (cond_on NULL = skipped)
 *                            LimitValueTransform.backward (transforms.py:131-158) on the last of these:
 *                                       x < value_lower -> 0 (limit_flags bit 0), x >= value_upper -> 0 (bit 1)
 *                            after    = before + that value, limited likewise with after_lower / after_upper
 *                                       (bits 2, 3)                             (before NULL = skipped)
 *                          level z of yhat starts at yhat + z * yhat_level_stride; any of out_direct /
 *                          out_unscaled / out_after may be NULL
 *   fv3hip_classify_onehot logits [n_class][n] -> onehot [n_class][n] (logits == max over classes,
 *                          ties all hot) and any_of [n] = onehot[cls_a] | onehot[cls_b]
 *                          (emulation/zhao_carr.py:193-198 _get_classify_output); any_of may be NULL
 */
int fv3hip_local_pack(const void *x, int dtype, int has_levels, int transform, double eps,
                      const float *center, const float *scale, int nz, int64_t ncol, float *out,
                      void *stream);
int fv3hip_local_unpack(const float *yhat, int64_t yhat_level_stride, const float *scale,
                        const float *center, const void *cond_on, int cond_dtype,
                        const float *edges, const float *cs_scale, const float *cs_center,
                        int n_bins, double min_scale, const void *before, int before_dtype,
                        int limit_flags, double value_lower, double value_upper,
                        double after_lower, double after_upper, int nz, int64_t ncol,
                        float *out_direct, float *out_unscaled, float *out_after, void *stream);
int fv3hip_classify_onehot(const void *logits, int dtype, int n_class, int64_t n,
                           uint8_t *onehot, uint8_t *any_of, int cls_a, int cls_b, void *stream);

/*
 * Predict-side glue of the fv3fit composite models (external/fv3fit/fv3fit/_shared/models.py:65-107, 223-276,
 * 442-483), on the prediction arrays right after the network.
 *   fv3hip_level_scale    TaperConfig.apply (_shared/config.py:11-24 with vcm/calc/calc.py:52-56): x [n_outer][nz][n_inner]
 *                         of `dtype` times the DEVICE float64 factors scale[nz]; the product is float64, as
 *                         numpy's float64 * float32 is
 *   fv3hip_member_reduce  EnsembleModel.predict (models.py:253-260): NaN-skipping mean / median (FV3HIP_OP_MEAN /
 *                         FV3HIP_OP_MEDIAN) over up to 32 member arrays of n values; `members` is a HOST array of
 *                         device pointers
 * SquashedOutputConfig.squash (_shared/config.py:135-142) is fv3hip_ew GT_S followed by WHERE_S.
 */
int fv3hip_level_scale(const void *x, int dtype, const double *scale, int64_t n_outer, int nz,
                       int64_t n_inner, double *out, void *stream);
int fv3hip_member_reduce(const void *const *members, int n_members, int dtype, int op, int64_t n,
                         void *out, void *stream);

/*
 * The flux-form output transforms of TransformedPredictor (external/fv3fit/fv3fit/_shared/models.py:279-337 applying
 * external/vcm/vcm/data_transform.py:140-300), replacing external/vcm/vcm/calc/flux_form.py on xarray:
 *   fv3hip_tendency_to_flux  _tendency_to_flux (flux_form.py:7-46): net_flux [n_outer][nz][n_inner] = flux at the interface
 *                            ABOVE each cell = toa_net_flux - cumsum(tendency * delp / g) of the cells above, and
 *                            surface_downward_flux [n_outer][n_inner] = the flux below the last cell + surface_upward_flux,
 *                            zero where negative if `rectify`.  closure = 1: _tendency_to_implied_surface_downward_flux
 *                            (:49-75), toa + upward - sum(tendency * delp / g); net_flux is not written (may be null).
 *                            toa_net_flux may be null (zero).  All arrays share `dtype` and are computed in it, operation by
 *                            operation as numpy does (a running sum: bit-identical for closure = 0).
 *   fv3hip_flux_to_tendency  _flux_to_tendency (:78-104): -(g * diff(concat(net_flux, down - up)) / delp).
 */
int fv3hip_tendency_to_flux(const void *tendency, const void *delp, const void *toa_net_flux,
                            const void *surface_upward_flux, int dtype, int64_t n_outer, int nz, int64_t n_inner,
                            int rectify, int closure, void *net_flux, void *surface_downward_flux, void *stream);
int fv3hip_flux_to_tendency(const void *net_flux, const void *surface_downward_flux, const void *surface_upward_flux,
                            const void *delp, int dtype, int64_t n_outer, int nz, int64_t n_inner, void *tendency,
                            void *stream);

/*
 * The novelty detectors OutOfSampleModel consults every timestep (external/fv3fit/fv3fit/_shared/models.py:340-440),
 * replacing sklearn on the host:
 *   fv3hip_minmax_score  MinMaxNoveltyDetector.predict (fv3fit/sklearn/_min_max_novelty_detector.py:94-121): one call per
 *                        packed variable x (feature f of sample i at x[f * feat_stride + i * sample_stride], `dtype`) folds
 *                        MinMaxScaler.transform's x * scale[f] + offset[f] (float64) into run_max / run_min [n]
 *                        (`first` != 0 starts them); `finish` != 0 writes score = max(max - 1, 0) + max(-min, 0).
 *   fv3hip_ocsvm_score   OCSVMNoveltyDetector.predict (_ocsvm_novelty_detector.py:124-160): score[i] = -sum_v dual_coef[v]
 *                        exp(-gamma |z_i - support_vectors[v]|^2), z = (x - mean) / scale; x [n_feat][n] float64 packed,
 *                        support_vectors [n_sv][n_feat] (already in the scaler's space, as sklearn stores them).
 */
int fv3hip_minmax_score(const void *x, int dtype, int64_t feat_stride, int64_t sample_stride, int n_feat,
                        const double *scale, const double *offset, int64_t n, int first, int finish, double *run_max,
                        double *run_min, double *score, void *stream);
int fv3hip_ocsvm_score(const double *x, int n_feat, int64_t n, const double *mean, const double *scale,
                       const double *support_vectors, const double *dual_coef, int n_sv, double gamma, double *score,
                       void *stream);

/*
 * Replaces mappm.interpolate_2d (external/mappm/mappm/interpolate_2d.f90:1-28; called from
 * external/vcm/vcm/interpolate.py:165-169): per column, linear interpolation of y(x) (n_in points,
 * x increasing) onto xp (n_out points); outside the column's range the result is fill_value.
 * float64.  Layouts as for fv3hip_mappm: COL_LEVEL [ncol][n] (n_batch = ncol, n_inner = 1) or
 * LEVEL_COL [n_batch][n][n_inner].  Bit-identical to the compiled Fortran.
 */
int fv3hip_interpolate_2d(const void *xp, const void *x, const void *y, int64_t n_batch,
                          int64_t n_inner, int n_in, int n_out, double fill_value, int layout,
                          void *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Zhao-Carr emulator post-processing (masks and conservation fixes on the emulator outputs)
 * ------------------------------------------------------------------------------------------
 * Replace the numpy / numba functions of external/emulation/emulation/masks.py:23-76 and
 * external/emulation/emulation/zhao_carr.py:60-344 that ModelConfig._build_masks composes
 * (external/emulation/emulation/config.py:175-221).  Arrays are contiguous [n0][n1] = [level][sample]
 * (the Fortran hook's layout) or flat (n); every array carries a dtype code (FV3HIP_F32/F64);
 * arithmetic and outputs use out_dtype = F64 if any operand is float64 (numpy's promotion), else F32.
 */
#define FV3HIP_ZC_NO_MASK 0            /* enforce_conservative_*: the emulator's cloud as is          */
#define FV3HIP_ZC_FORTRAN_VANISHES 1   /* mask_where_fortran_cloud_vanishes_gscond: aux = state cloud after gscond */
#define FV3HIP_ZC_FORTRAN_IDENTICAL 2  /* mask_where_fortran_cloud_identical: aux = state cloud after gscond       */
#define FV3HIP_ZC_CLASS_ZERO_CLOUD 3   /* mask_zero_cloud_classifier: aux = logits [n_class][n0][n1]               */
#define FV3HIP_ZC_CLASS_ZERO_TEND 4    /* mask_zero_tend_classifier:  aux = logits [n_class][n0][n1]               */

/* squash_gscond / squash_precpd (zhao_carr.py:60-86): cloud < bound -> 0, the removed water goes to qv.
 * cloud_out keeps cloud's dtype; qv_out has out_dtype. */
int fv3hip_zc_squash(const void *cloud, int cloud_dtype, const void *humidity, int hum_dtype,
                     int64_t n, double bound, int out_dtype, void *cloud_out, void *qv_out,
                     void *stream);
/* infer_gscond_cloud_from_conservation (zhao_carr.py:80-84). */
int fv3hip_zc_infer_cloud(const void *cloud_in, const void *qv_in, int state_dtype,
                          const void *qv_emul, int emul_dtype, int64_t n, int out_dtype,
                          void *cloud_out, void *stream);
/* The gscond masks + _update_with_net_condensation (zhao_carr.py:97-246): choose the cloud (mode),
 * limit the net condensation by the available vapour / liquid, apply it with the liquid latent heat
 * or (phase_dependent) the ice/water-flag scan along the last axis (zhao_carr.py:114-151). */
int fv3hip_zc_gscond_conserve(const void *cloud_in, const void *qv_in, const void *t_in,
                              int state_dtype, const void *cloud_emul, int emul_dtype, int mode,
                              const void *aux, int aux_dtype, int n_class, int cls, int64_t n0,
                              int64_t n1, int phase_dependent, int out_dtype, void *cloud_out,
                              void *qv_out, void *t_out, void *stream);
/* enforce_conservative_precpd (zhao_carr.py:249-323): strict TOA (last level) to surface budget. */
int fv3hip_zc_precpd_conserve(const void *cloud_g, const void *qv_g, const void *t_g,
                              const void *delp, int state_dtype, const void *cloud_p,
                              const void *qv_p, int emul_dtype, int64_t n0, int64_t n1,
                              int out_dtype, void *cloud_out, void *qv_out, void *t_out,
                              void *precip_out, void *stream);
/* conservative_precip_simple (zhao_carr.py:326-344): precip[n1] from the column water budget. */
int fv3hip_zc_precip_simple(const void *cloud_g, const void *qv_g, const void *delp,
                            int state_dtype, const void *cloud_p, const void *qv_p,
                            int emul_dtype, int64_t n0, int64_t n1, int out_dtype,
                            void *precip_out, void *stream);
/* mask_zero_cloud_classifier_precpd (zhao_carr.py:230-237): x -> 0 where class `cls` is hot. */
int fv3hip_zc_class_zero(const void *x, int dtype, const void *logits, int logits_dtype,
                         int n_class, int cls, int64_t n, void *out, void *stream);
/*
 * Replaces vcm.non_negative_sphum and vcm.non_negative_sphum_mse_conserving
 * (external/vcm/vcm/calc/thermo/non_negative_sphum.py:6-45), applied to the ML tendencies every
 * timestep (workflows/prognostic_c48_run/runtime/steppers/machine_learning.py:226-237).
 * mse_conserving = 0: where sphum + dQ2 dt < 0 both tendencies are scaled by -sphum / (dt dQ2);
 * mse_conserving = 1: dQ2 -> -sphum / dt there, and dQ1 is re-derived so that the moist static
 * energy tendency (cp - Rd) dQ1 + Lv dQ2 is unchanged.  q1 / q1_out may be NULL.  All arrays: n values
 * of `dtype` (FV3HIP_F32 / FV3HIP_F64).
 */
int fv3hip_non_negative_sphum(const void *sphum, const void *q1, const void *q2, int dtype,
                              int64_t n, double dt, int mse_conserving, void *q1_out,
                              void *q2_out, void *stream);
/* RangeMask (masks.py:23-41): np.maximum(x, lo) / np.minimum(x, hi), NaN-propagating. */
int fv3hip_clamp(const void *x, int dtype, int64_t n, double lo, double hi, int has_lo,
                 int has_hi, void *out, void *stream);
/* LevelMask (masks.py:44-76): float64 copy of the emulator field with levels [start, stop) taken from
 * src (or fill_value when src is NULL). */
int fv3hip_level_fill(const void *emul, int emul_dtype, const void *src, int src_dtype,
                      double fill_value, int64_t n0, int64_t n1, int64_t start, int64_t stop,
                      void *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FV3HIP_H */
