/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, single precision, one column at a time) of the
 * reference's vertical remapping routine.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's shared library.
 *
 * Follows (reference paths relative to /root/reference):
 *   external/mappm/mappm/mappm.f90:10-126    subroutine mappm
 *   external/mappm/mappm/mappm.f90:614-851   subroutine ppm_profile  (kord <= 7)
 *   external/mappm/mappm/mappm.f90:854-931   subroutine ppm_limiters
 *   external/mappm/mappm/mappm.f90:132-529   subroutine cs_profile   (kord > 7)
 *   external/mappm/mappm/mappm.f90:532-611   subroutine cs_limiters
 * (callers only ever pass iv=1, kord=1, external/vcm/vcm/cubedsphere/regridz.py:227-228,296;
 * kord > 7 with iv = -2 reads the array `qs` that mappm never sets, mappm.f90:34,51 and
 * :152-176, and is refused here.)
 *
 * Arithmetic is written in the same association order as the Fortran source
 * (default REAL = real*4, `x**2` = x*x, left-to-right `*` and `/`) and this
 * file is compiled with -ffp-contract=off, so that on hardware with IEEE
 * single precision the results are bit-identical to the Fortran compiled
 * without FMA contraction (oracle/_ref/libmappm_ref.so, built by
 * oracle/Makefile from the reference's own source).  That pinning is checked
 * by tests/test_oracle_mappm.py.
 *
 * Semantics kept from the Fortran, including the odd ones:
 *   - `k0` carries from one target layer to the next within a column
 *     (mappm.f90:59,74,111);
 *   - if the top-edge search (loop 45) falls through, `qsum`, `dpsum`, `k1`
 *     keep whatever they held (mappm.f90:95-98).  In the Fortran these are
 *     subroutine-scope scalars that even carry from the previous column (and
 *     are uninitialised for the first one); here they are reset to
 *     (0, 0, 1) at the start of every column, which is the one deliberate
 *     difference: it only matters for columns whose pressures are NaN or not
 *     monotone, for which the reference result is undefined.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FV3_ORACLE_OK 0
#define FV3_ORACLE_EKORD -1
#define FV3_ORACLE_EKM -2
#define FV3_ORACLE_EIV -4

static inline float f_sign(float a, float b) { return copysignf(fabsf(a), b); }
/* Fortran MIN/MAX as flang lowers them: a compare-and-select chain. */
static inline float f_min2(float a, float b) { return (a < b) ? a : b; }
static inline float f_max2(float a, float b) { return (a > b) ? a : b; }
static inline float f_min3(float a, float b, float c) { return f_min2(f_min2(a, b), c); }
static inline float f_max3(float a, float b, float c) { return f_max2(f_max2(a, b), c); }

/* mappm.f90:854-931; one column, one level.  a[0..3] = a4(1:4). */
static void ppm_limiters1(float dm, float *a1, float *a2, float *a3, float *a4, int lmt)
{
    const float r12 = 1.f / 12.f;
    if (lmt == 3) return;
    if (lmt == 0) {
        /* Standard PPM constraint */
        if (dm == 0.f) {
            *a2 = *a1;
            *a3 = *a1;
            *a4 = 0.f;
        } else {
            float da1 = *a3 - *a2;
            float da2 = da1 * da1;
            float a6da = *a4 * da1;
            if (a6da < -da2) {
                *a4 = 3.f * (*a2 - *a1);
                *a3 = *a2 - *a4;
            } else if (a6da > da2) {
                *a4 = 3.f * (*a3 - *a1);
                *a2 = *a3 - *a4;
            }
        }
    } else if (lmt == 1) {
        /* Improved full monotonicity constraint (Lin 2004) */
        float qmp = 2.f * dm;
        *a2 = *a1 - f_sign(f_min2(fabsf(qmp), fabsf(*a2 - *a1)), qmp);
        *a3 = *a1 + f_sign(f_min2(fabsf(qmp), fabsf(*a3 - *a1)), qmp);
        *a4 = 3.f * (2.f * *a1 - (*a2 + *a3));
    } else if (lmt == 2) {
        /* Positive definite constraint */
        if (fabsf(*a3 - *a2) < -*a4) {
            float d = *a3 - *a2;
            float fmin = *a1 + 0.25f * (d * d) / *a4 + *a4 * r12;
            if (fmin < 0.f) {
                if (*a1 < *a3 && *a1 < *a2) {
                    *a3 = *a1;
                    *a2 = *a1;
                    *a4 = 0.f;
                } else if (*a3 > *a2) {
                    *a4 = 3.f * (*a2 - *a1);
                    *a3 = *a2 - *a4;
                } else {
                    *a4 = 3.f * (*a3 - *a1);
                    *a2 = *a3 - *a4;
                }
            }
        }
    }
}

/* mappm.f90:532-611; one column, one level.  mode = the routine's `iv` argument. */
static void cs_limiters1(int extm, float a1, float *a2, float *a3, float *a4, int mode)
{
    const float r12 = 1.f / 12.f;
    float da1, da2, a6da;
    if (mode == 0) {
        /* Positive definite constraint */
        if (a1 <= 0.f) {
            *a2 = a1;
            *a3 = a1;
            *a4 = 0.f;
        } else if (fabsf(*a3 - *a2) < -*a4) {
            if ((a1 + 0.25f * ((*a3 - *a2) * (*a3 - *a2)) / *a4 + *a4 * r12) < 0.f) {
                /* local minimum is negative */
                if (a1 < *a3 && a1 < *a2) {
                    *a3 = a1;
                    *a2 = a1;
                    *a4 = 0.f;
                } else if (*a3 > *a2) {
                    *a4 = 3.f * (*a2 - a1);
                    *a3 = *a2 - *a4;
                } else {
                    *a4 = 3.f * (*a3 - a1);
                    *a2 = *a3 - *a4;
                }
            }
        }
        return;
    }
    if (mode == 1 ? ((a1 - *a2) * (a1 - *a3) >= 0.f) : extm) {
        *a2 = a1;
        *a3 = a1;
        *a4 = 0.f;
        return;
    }
    /* (mode 1 and the standard PPM constraint share the rest) */
    da1 = *a3 - *a2;
    da2 = da1 * da1;
    a6da = *a4 * da1;
    if (a6da < -da2) {
        *a4 = 3.f * (*a2 - a1);
        *a3 = *a2 - *a4;
    } else if (a6da > da2) {
        *a4 = 3.f * (*a3 - a1);
        *a2 = *a3 - *a4;
    }
}

/*
 * mappm.f90:132-529 for one column (iv != -2).  1-based arrays of at least km+2 entries:
 * a1 = a4(1,:) (the cell means, untouched), al / ar / a6 = a4(2:4,:), qe = the edge values
 * q(1:km+1), gam = the tridiagonal's work array and then the differences of the means.
 * ext holds three flags per level: bit 0 extm, bit 1 ext5, bit 2 ext6.
 */
static void cs_profile1(const float *a1, float *al, float *ar, float *a6, const float *delp, int km, int iv,
                        int kord, float *qe, float *gam, int *ext)
{
    int k;
    float grat, bet, d4 = 0.f, a_bot, pmp_1, lac_1, pmp_2, lac_2, x0, x1;

    grat = delp[2] / delp[1];
    bet = grat * (grat + 0.5f);
    qe[1] = ((grat + grat) * (grat + 1.f) * a1[1] + a1[2]) / bet;
    gam[1] = (1.f + grat * (grat + 1.5f)) / bet;
    for (k = 2; k <= km; ++k) {
        d4 = delp[k - 1] / delp[k];
        bet = 2.f + d4 + d4 - gam[k - 1];
        qe[k] = (3.f * (a1[k - 1] + d4 * a1[k]) - qe[k - 1]) / bet;
        gam[k] = d4 / bet;
    }
    a_bot = 1.f + d4 * (d4 + 1.5f);
    qe[km + 1] = (2.f * d4 * (d4 + 1.f) * a1[km] + a1[km - 1] - a_bot * qe[km]) / (d4 * (d4 + 0.5f) - a_bot * gam[km]);
    for (k = km; k >= 1; --k) qe[k] = qe[k] - gam[k] * qe[k + 1];

    if (kord > 16) { /* perfectly linear scheme */
        for (k = 1; k <= km; ++k) {
            al[k] = qe[k];
            ar[k] = qe[k + 1];
            a6[k] = 3.f * (2.f * a1[k] - (al[k] + ar[k]));
        }
        return;
    }

    /* large-scale constraints */
    qe[2] = f_min2(qe[2], f_max2(a1[1], a1[2]));
    qe[2] = f_max2(qe[2], f_min2(a1[1], a1[2]));
    for (k = 2; k <= km; ++k) gam[k] = a1[k] - a1[k - 1];
    for (k = 3; k <= km - 1; ++k) {
        if (gam[k - 1] * gam[k + 1] > 0.f) {
            qe[k] = f_min2(qe[k], f_max2(a1[k - 1], a1[k]));
            qe[k] = f_max2(qe[k], f_min2(a1[k - 1], a1[k]));
        } else if (gam[k - 1] > 0.f) { /* a local maximum */
            qe[k] = f_max2(qe[k], f_min2(a1[k - 1], a1[k]));
        } else { /* a local minimum */
            qe[k] = f_min2(qe[k], f_max2(a1[k - 1], a1[k]));
            if (iv == 0) qe[k] = f_max2(0.f, qe[k]);
        }
    }
    qe[km] = f_min2(qe[km], f_max2(a1[km - 1], a1[km]));
    qe[km] = f_max2(qe[km], f_min2(a1[km - 1], a1[km]));

    for (k = 1; k <= km; ++k) {
        al[k] = qe[k];
        ar[k] = qe[k + 1];
    }
    for (k = 1; k <= km; ++k) {
        ext[k] = (k == 1 || k == km) ? ((al[k] - a1[k]) * (ar[k] - a1[k]) > 0.f) : (gam[k] * gam[k + 1] < 0.f);
        if (kord > 9) {
            x0 = 2.f * a1[k] - (al[k] + ar[k]);
            x1 = fabsf(al[k] - ar[k]);
            a6[k] = 3.f * x0;
            if (fabsf(x0) > x1) ext[k] |= 2;
            if (fabsf(a6[k]) > x1) ext[k] |= 4;
        }
    }
#define EXTM(k) (ext[k] & 1)
#define EXT5(k) (ext[k] & 2)
#define EXT6(k) (ext[k] & 4)
#define A6_FROM_EDGES(k) a6[k] = 3.f * (2.f * a1[k] - (al[k] + ar[k]))
#define HUYNH(k)                                                                              \
    do {                                                                                      \
        pmp_1 = a1[k] - 2.f * gam[k + 1];                                                     \
        lac_1 = pmp_1 + 1.5f * gam[k + 2];                                                    \
        al[k] = f_min2(f_max2(al[k], f_min3(a1[k], pmp_1, lac_1)), f_max3(a1[k], pmp_1, lac_1)); \
        pmp_2 = a1[k] + 2.f * gam[k];                                                         \
        lac_2 = pmp_2 - 1.5f * gam[k - 1];                                                    \
        ar[k] = f_min2(f_max2(ar[k], f_min3(a1[k], pmp_2, lac_2)), f_max3(a1[k], pmp_2, lac_2)); \
    } while (0)
#define FLAT(k)        \
    do {               \
        al[k] = a1[k]; \
        ar[k] = a1[k]; \
    } while (0)

    /* subgrid constraints: the top two and the bottom two layers always use the monotonic mapping */
    if (iv == 0) {
        al[1] = f_max2(0.f, al[1]);
    } else if (iv == -1) {
        if (al[1] * a1[1] <= 0.f) al[1] = 0.f;
    } else if (iv == 2) {
        FLAT(1);
        a6[1] = 0.f;
    }
    if (iv != 2) {
        A6_FROM_EDGES(1);
        cs_limiters1(EXTM(1), a1[1], &al[1], &ar[1], &a6[1], 1);
    }
    A6_FROM_EDGES(2);
    cs_limiters1(EXTM(2), a1[2], &al[2], &ar[2], &a6[2], 2);

    for (k = 3; k <= km - 2; ++k) {
        if (kord < 9) {
            HUYNH(k);
            A6_FROM_EDGES(k);
        } else if (kord == 9 || kord == 12) {
            if (kord == 9 ? (EXTM(k) && (EXTM(k - 1) || EXTM(k + 1))) : EXTM(k)) { /* a 2-delta-z wave */
                FLAT(k);
                a6[k] = 0.f;
            } else {
                a6[k] = 6.f * a1[k] - 3.f * (al[k] + ar[k]);
                if (fabsf(a6[k]) > fabsf(al[k] - ar[k])) { /* not monotonic inside the smooth region */
                    HUYNH(k);
                    a6[k] = 6.f * a1[k] - 3.f * (al[k] + ar[k]);
                }
            }
        } else if (kord == 10 || kord == 16) {
            if (EXT5(k)) {
                if (EXT5(k - 1) || EXT5(k + 1)) {
                    FLAT(k);
                } else if (EXT6(k - 1) || EXT6(k + 1)) {
                    HUYNH(k);
                }
            } else if (kord == 10 && EXT6(k)) {
                if (EXT5(k - 1) || EXT5(k + 1)) HUYNH(k);
            }
            A6_FROM_EDGES(k);
        } else if (kord == 13) {
            if (EXT6(k) && EXT6(k - 1) && EXT6(k + 1)) FLAT(k);
            A6_FROM_EDGES(k);
        } else if (kord == 14) {
            A6_FROM_EDGES(k);
        } else if (kord == 15) {
            if (EXT5(k)) {
                if (EXT5(k - 1) || EXT5(k + 1)) FLAT(k);
            } else if (EXT6(k)) {
                HUYNH(k);
            }
            A6_FROM_EDGES(k);
        } else { /* kord == 11 */
            if (EXT5(k) && (EXT5(k - 1) || EXT5(k + 1))) { /* a noisy region */
                FLAT(k);
                a6[k] = 0.f;
            } else {
                A6_FROM_EDGES(k);
            }
        }
        if (iv == 0) cs_limiters1(EXTM(k), a1[k], &al[k], &ar[k], &a6[k], 0);
    }

    if (iv == 0) {
        ar[km] = f_max2(0.f, ar[km]);
    } else if (iv == -1) {
        if (ar[km] * a1[km] <= 0.f) ar[km] = 0.f;
    }
    A6_FROM_EDGES(km - 1);
    cs_limiters1(EXTM(km - 1), a1[km - 1], &al[km - 1], &ar[km - 1], &a6[km - 1], 2);
    A6_FROM_EDGES(km);
    cs_limiters1(EXTM(km), a1[km], &al[km], &ar[km], &a6[km], 1);
#undef EXTM
#undef EXT5
#undef EXT6
#undef A6_FROM_EDGES
#undef HUYNH
#undef FLAT
}

/*
 * mappm.f90:614-851 for one column.  All arrays are 1-based (index 0 unused)
 * and have at least km+2 entries.  q = a4(1,:), al = a4(2,:), ar = a4(3,:),
 * a6 = a4(4,:).
 */
static void ppm_profile1(const float *q, float *al, float *ar, float *a6, const float *delp,
                         int km, int iv, int kord, float *dc, float *h2, float *delq, float *df2,
                         float *d4)
{
    int k;
    const int km1 = km - 1;
    float c1, c2, c3, a1, a2, d1, d2, qm, dq, qmp, pmp, lac, fac;

    for (k = 2; k <= km; ++k) {
        delq[k - 1] = q[k] - q[k - 1];
        d4[k] = delp[k - 1] + delp[k];
    }
    for (k = 2; k <= km1; ++k) {
        c1 = (delp[k - 1] + 0.5f * delp[k]) / d4[k + 1];
        c2 = (delp[k + 1] + 0.5f * delp[k]) / d4[k];
        df2[k] = delp[k] * (c1 * delq[k] + c2 * delq[k - 1]) / (d4[k] + delp[k + 1]);
        dc[k] = f_sign(f_min3(fabsf(df2[k]), f_max3(q[k - 1], q[k], q[k + 1]) - q[k],
                              q[k] - f_min3(q[k - 1], q[k], q[k + 1])),
                       df2[k]);
    }
    /* 4th order interpolation of the provisional cell edge value */
    for (k = 3; k <= km1; ++k) {
        c1 = delq[k - 1] * delp[k - 1] / d4[k];
        a1 = d4[k - 1] / (d4[k] + delp[k - 1]);
        a2 = d4[k + 1] / (d4[k] + delp[k]);
        al[k] = q[k - 1] + c1 +
                2.f / (d4[k - 1] + d4[k + 1]) *
                    (delp[k] * (c1 * (a1 - a2) + a2 * dc[k - 1]) - delp[k - 1] * a1 * dc[k]);
    }
    /* Area preserving cubic with 2nd deriv. = 0 at the boundaries.  Top: */
    d1 = delp[1];
    d2 = delp[2];
    qm = (d2 * q[1] + d1 * q[2]) / (d1 + d2);
    dq = 2.f * (q[2] - q[1]) / (d1 + d2);
    c1 = 4.f * (al[3] - qm - d2 * dq) / (d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
    c3 = dq - 0.5f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
    al[2] = qm - 0.25f * c1 * d1 * d2 * (d2 + 3.f * d1);
    al[1] = d1 * (2.f * c1 * (d1 * d1) - c3) + al[2];
    al[2] = f_max2(al[2], f_min2(q[1], q[2]));
    al[2] = f_min2(al[2], f_max2(q[1], q[2]));
    dc[1] = 0.5f * (al[2] - q[1]);

    if (iv == 0) {
        al[1] = f_max2(0.f, al[1]);
        al[2] = f_max2(0.f, al[2]);
    } else if (iv == -1) {
        if (al[1] * q[1] <= 0.f) al[1] = 0.f;
    } else if (abs(iv) == 2) {
        al[1] = q[1];
        ar[1] = q[1];
    }

    /* Bottom */
    d1 = delp[km];
    d2 = delp[km1];
    qm = (d2 * q[km] + d1 * q[km1]) / (d1 + d2);
    dq = 2.f * (q[km1] - q[km]) / (d1 + d2);
    c1 = (al[km1] - qm - d2 * dq) / (d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
    c3 = dq - 2.0f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
    al[km] = qm - c1 * d1 * d2 * (d2 + 3.f * d1);
    ar[km] = d1 * (8.f * c1 * (d1 * d1) - c3) + al[km];
    al[km] = f_max2(al[km], f_min2(q[km], q[km1]));
    al[km] = f_min2(al[km], f_max2(q[km], q[km1]));
    dc[km] = 0.5f * (q[km] - al[km]);

    if (iv == 0) {
        al[km] = f_max2(0.f, al[km]);
        ar[km] = f_max2(0.f, ar[km]);
    } else if (iv < 0) {
        if (q[km] * ar[km] <= 0.f) ar[km] = 0.f;
    }

    for (k = 1; k <= km1; ++k) ar[k] = al[k + 1];

    /* Top 2 and bottom 2 layers always use monotonic mapping */
    for (k = 1; k <= 2; ++k) {
        a6[k] = 3.f * (2.f * q[k] - (al[k] + ar[k]));
        /* ppm_limiters receives a4(1:4,i,k) by reference: q is intent(in) to it */
        float qq = q[k];
        ppm_limiters1(dc[k], &qq, &al[k], &ar[k], &a6[k], 0);
    }

    if (kord >= 7) {
        /* Huynh's 2nd constraint */
        for (k = 2; k <= km1; ++k) {
            h2[k] = 2.f * (dc[k + 1] / delp[k + 1] - dc[k - 1] / delp[k - 1]) /
                    (delp[k] + 0.5f * (delp[k - 1] + delp[k + 1])) * (delp[k] * delp[k]);
        }
        fac = 1.5f;
        for (k = 3; k <= km - 2; ++k) {
            /* Right edges */
            pmp = 2.f * dc[k];
            qmp = q[k] + pmp;
            lac = q[k] + fac * h2[k - 1] + dc[k];
            ar[k] = f_min2(f_max2(ar[k], f_min3(q[k], qmp, lac)), f_max3(q[k], qmp, lac));
            /* Left edges */
            qmp = q[k] - pmp;
            lac = q[k] + fac * h2[k + 1] - dc[k];
            al[k] = f_min2(f_max2(al[k], f_min3(q[k], qmp, lac)), f_max3(q[k], qmp, lac));
            /* Recompute A6 */
            a6[k] = 3.f * (2.f * q[k] - (al[k] + ar[k]));
            /* Additional constraint to ensure positivity when kord=7 */
            if (iv == 0 && kord >= 6) {
                float qq = q[k];
                ppm_limiters1(dc[k], &qq, &al[k], &ar[k], &a6[k], 2);
            }
        }
    } else {
        int lmt = kord - 3;
        lmt = (lmt > 0) ? lmt : 0;
        if (iv == 0) lmt = (lmt < 2) ? lmt : 2;
        for (k = 3; k <= km - 2; ++k) {
            if (kord != 4) a6[k] = 3.f * (2.f * q[k] - (al[k] + ar[k]));
            if (kord != 6) {
                float qq = q[k];
                ppm_limiters1(dc[k], &qq, &al[k], &ar[k], &a6[k], lmt);
            }
        }
    }

    for (k = km1; k <= km; ++k) {
        a6[k] = 3.f * (2.f * q[k] - (al[k] + ar[k]));
        float qq = q[k];
        ppm_limiters1(dc[k], &qq, &al[k], &ar[k], &a6[k], 0);
    }
}

/*
 * mappm.f90:58-124 for one column.  pe1[1..km+1], q1[1..km], pe2[1..kn+1],
 * q2[1..kn]; dp1/al/ar/a6 from the reconstruction above (1-based).
 */
static void remap_column(int km, const float *pe1, const float *q1, int kn, const float *pe2,
                         float *q2, const float *dp1, const float *al, const float *ar,
                         const float *a6)
{
    const float r3 = 1.f / 3.f, r23 = 2.f / 3.f;
    int k, L, k0 = 1, k1 = 1;
    float PL, PR, TT, delp, esl, qsum = 0.f, dpsum = 0.f;

    for (k = 1; k <= kn; ++k) {
        if (pe2[k] <= pe1[1]) {
            /* above old ptop */
            q2[k] = q1[1];
            continue;
        } else if (pe2[k] >= pe1[km + 1]) {
            /* Entire grid below old ps */
            q2[k] = q1[km];
            continue;
        }
        int done = 0;
        for (L = k0; L <= km; ++L) {
            /* locate the top edge at pe2(k) */
            if (pe2[k] >= pe1[L] && pe2[k] <= pe1[L + 1]) {
                k0 = L;
                PL = (pe2[k] - pe1[L]) / dp1[L];
                if (pe2[k + 1] <= pe1[L + 1]) {
                    /* entire new grid is within the original grid */
                    PR = (pe2[k + 1] - pe1[L]) / dp1[L];
                    TT = r3 * (PR * (PR + PL) + PL * PL);
                    q2[k] = al[L] + 0.5f * (a6[L] + ar[L] - al[L]) * (PR + PL) - a6[L] * TT;
                    done = 1;
                } else {
                    /* Fractional area... */
                    delp = pe1[L + 1] - pe2[k];
                    TT = r3 * (1.f + PL * (1.f + PL));
                    qsum = delp * (al[L] + 0.5f * (a6[L] + ar[L] - al[L]) * (1.f + PL) - a6[L] * TT);
                    dpsum = delp;
                    k1 = L + 1;
                }
                break;
            }
        }
        if (done) continue;
        /* label 111 (also reached, with stale qsum/dpsum/k1, if loop 45 found nothing) */
        int finished = 0;
        for (L = k1; L <= km; ++L) {
            if (pe2[k + 1] > pe1[L + 1]) {
                /* Whole layer.. */
                qsum = qsum + dp1[L] * q1[L];
                dpsum = dpsum + dp1[L];
            } else {
                delp = pe2[k + 1] - pe1[L];
                esl = delp / dp1[L];
                qsum = qsum +
                       delp * (al[L] + 0.5f * esl * (ar[L] - al[L] + a6[L] * (1.f - r23 * esl)));
                dpsum = dpsum + delp;
                k0 = L;
                finished = 1;
                break;
            }
        }
        if (!finished) {
            delp = pe2[k + 1] - pe1[km + 1];
            if (delp > 0.f) {
                /* Extended below old ps */
                qsum = qsum + delp * q1[km];
                dpsum = dpsum + delp;
            }
        }
        q2[k] = qsum / dpsum;
    }
}

/*
 * Column-major-in-memory driver.  Layout of every array is [column][level]
 * (level fastest), i.e. the C-order arrays the reference's Python caller
 * passes to f2py (external/vcm/vcm/cubedsphere/regridz.py:326-334).
 *   pe1: [ncol][km+1], q1: [ncol][km], pe2: [ncol][kn+1], q2: [ncol][kn]
 * Returns 0, or a negative error code (km < 4; kord > 7 with iv = -2).
 */
int fv3_oracle_mappm(const float *pe1, const float *q1, const float *pe2, float *q2, long ncol,
                     int km, int kn, int iv, int kord)
{
    if (kord > 7 && iv == -2) return FV3_ORACLE_EIV;
    if (km < 4) return FV3_ORACLE_EKM;
    const int n = km + 3;
    float *buf = (float *)malloc(sizeof(float) * (size_t)n * 12 + sizeof(float) * (size_t)(kn + 3) * 2 + sizeof(int) * (size_t)n);
    if (!buf) return -3;
    int *ext = (int *)(buf + (size_t)n * 12 + (size_t)(kn + 3) * 2);
    float *p1 = buf, *q = p1 + n, *dp = q + n, *al = dp + n, *ar = al + n, *a6 = ar + n,
          *dc = a6 + n, *h2 = dc + n, *delq = h2 + n, *df2 = delq + n, *d4 = df2 + n;
    float *p2 = d4 + n, *o = p2 + (kn + 3);
    for (long i = 0; i < ncol; ++i) {
        memset(buf, 0, sizeof(float) * (size_t)n * 12);
        for (int k = 1; k <= km + 1; ++k) p1[k] = pe1[i * (km + 1) + (k - 1)];
        for (int k = 1; k <= km; ++k) q[k] = q1[i * km + (k - 1)];
        for (int k = 1; k <= kn + 1; ++k) p2[k] = pe2[i * (kn + 1) + (k - 1)];
        for (int k = 1; k <= km; ++k) dp[k] = p1[k + 1] - p1[k];
        if (kord > 7)
            cs_profile1(q, al, ar, a6, dp, km, iv, kord, h2, dc, ext);
        else
            ppm_profile1(q, al, ar, a6, dp, km, iv, kord, dc, h2, delq, df2, d4);
        remap_column(km, p1, q, kn, p2, o, dp, al, ar, a6);
        for (int k = 1; k <= kn; ++k) q2[i * kn + (k - 1)] = o[k];
    }
    free(buf);
    return FV3_ORACLE_OK;
}

/* Exposes the reconstruction alone (AL, AR, A6 per level) for kernel debugging. */
int fv3_oracle_ppm_profile(const float *pe1, const float *q1, float *al_out, float *ar_out,
                           float *a6_out, long ncol, int km, int iv, int kord)
{
    if (kord > 7 && iv == -2) return FV3_ORACLE_EIV;
    if (km < 4) return FV3_ORACLE_EKM;
    const int n = km + 3;
    float *buf = (float *)malloc(sizeof(float) * (size_t)n * 12 + sizeof(int) * (size_t)n);
    if (!buf) return -3;
    int *ext = (int *)(buf + (size_t)n * 12);
    float *p1 = buf, *q = p1 + n, *dp = q + n, *al = dp + n, *ar = al + n, *a6 = ar + n,
          *dc = a6 + n, *h2 = dc + n, *delq = h2 + n, *df2 = delq + n, *d4 = df2 + n;
    for (long i = 0; i < ncol; ++i) {
        memset(buf, 0, sizeof(float) * (size_t)n * 12);
        for (int k = 1; k <= km + 1; ++k) p1[k] = pe1[i * (km + 1) + (k - 1)];
        for (int k = 1; k <= km; ++k) q[k] = q1[i * km + (k - 1)];
        for (int k = 1; k <= km; ++k) dp[k] = p1[k + 1] - p1[k];
        if (kord > 7)
            cs_profile1(q, al, ar, a6, dp, km, iv, kord, h2, dc, ext);
        else
            ppm_profile1(q, al, ar, a6, dp, km, iv, kord, dc, h2, delq, df2, d4);
        for (int k = 1; k <= km; ++k) {
            al_out[i * km + (k - 1)] = al[k];
            ar_out[i * km + (k - 1)] = ar[k];
            a6_out[i * km + (k - 1)] = a6[k];
        }
    }
    free(buf);
    return FV3_ORACLE_OK;
}


/* interpolate_2d (external/mappm/mappm/interpolate_2d.f90:1-28): per row i, linear interpolation of
 * y(x) onto the points xp; the search runs over ALL intervals k (a later match overwrites an earlier
 * one), points outside [x(1), x(n_in)] keep fill_value.  Row-major [m][n] arrays, double precision. */
int fv3_oracle_interpolate_2d(const double *xp, const double *x, const double *y, double *y_out, double fill_value,
                              long m, int n_in, int n_out)
{
    for (long i = 0; i < m; ++i) {
        const double *xi = x + i * n_in, *yi = y + i * n_in;
        for (int j = 0; j < n_out; ++j) {
            const double p = xp[i * n_out + j];
            double r = fill_value;
            for (int k = 0; k < n_in - 1; ++k) {
                if (xi[k] <= p && p < xi[k + 1]) {
                    const double w = (p - xi[k]) / (xi[k + 1] - xi[k]);
                    r = yi[k] * (1 - w) + yi[k + 1] * w;
                } else if (xi[k] == p) {
                    r = yi[k];
                } else if (xi[k + 1] == p) {
                    r = yi[k + 1];
                }
            }
            y_out[i * n_out + j] = r;
        }
    }
    return 0;
}
