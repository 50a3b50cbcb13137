// mappm for the configuration every caller of the reference passes (kord <= 3: the standard PPM monotonicity
// constraint on all levels, external/vcm/vcm/cubedsphere/regridz.py:227-228; any iv) on the model's native
// [batch][level][column] layout -- the "sweep" kernel.
//
// Reference restated: external/mappm/mappm/mappm.f90:10-126 (mappm), 614-851 (ppm_profile), 854-931 (ppm_limiters).
//
// Same algorithm as mappm_merge_multi_kernel (vertical.hip): one lane per column, the wave marches over the SOURCE
// layers together with the PPM reconstruction in a sliding register window, each lane emits the target layers that end
// inside the current source layer.  What is different -- the r01 counters showed the merge kernels to be bound by VALU
// issue (30 K instructions per column, two thirds of them not division):
//   * one wavefront per workgroup and n_inner % 64 == 0, so a wave's 64 columns share their batch: every row address is
//     an SGPR base (bumped on the scalar unit per level) plus one loop-invariant VGPR lane offset -- no per-access 64-bit
//     VALU address arithmetic;
//   * the LDS ring of target interfaces is filled uniformly (scalar bookkeeping, rows follow lane 0's cursor) instead of
//     per lane; a lane outside the ring's window reads memory;
//   * kord, the limiter and the a6 recomputation are compile-time facts (lmt = 0 everywhere), the limiter is select-only;
//   * every pressure-only subexpression (d4, the five denominators of a level) is formed once per level and shared by
//     dc(k), al(k) and all NF fields; d4(k+1) and its reciprocal are carried to the next level;
//   * ARITHMETIC MODES.  EXACT: IEEE division and the Fortran's association order -> bit-identical to the compiled
//     reference (tests/test_gpu_vertical.py).  FAST: a / b = a * v_rcp_f32(b) (1 ulp) with shared reciprocals; the result
//     differs from EXACT by a few ulp of the layer's values (<= 1e-5 relative to the column's range is asserted in the
//     tests; BASELINE.json north_star asks for 1e-5, and the reference itself is not bit-reproducible across platforms,
//     external/vcm/tests/test_coarsen_restarts.py:119-123).
// Columns with NaN or non-monotone pressures go on the worklist and are redone by mappm_fallback_kernel, as before.
#include "common.h"
#include "remap.h"

namespace fv3hip {
namespace {

__device__ __forceinline__ float s_sign(float a, float b) { return copysignf(fabsf(a), b); }
__device__ __forceinline__ float s_min2(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float s_max2(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float s_min3(float a, float b, float c) { return s_min2(s_min2(a, b), c); }
__device__ __forceinline__ float s_max3(float a, float b, float c) { return s_max2(s_max2(a, b), c); }

// A denominator: EXACT keeps the value and divides (IEEE) at every use, FAST takes the reciprocal once.
template <bool FAST>
struct Den {
    float v;
    __device__ __forceinline__ explicit Den(float d) : v(FAST ? __builtin_amdgcn_rcpf(d) : d) {}
    __device__ __forceinline__ float under(float num) const { return FAST ? num * v : num / v; }
};

// mappm.f90:875-894 (lmt = 0), select-only
__device__ __forceinline__ void limit0(float dm, float q, float &al, float &ar, float &a6)
{
    const float da1 = ar - al;
    const float da2 = da1 * da1;
    const float a6da = a6 * da1;
    const bool lo = a6da < -da2, hi = a6da > da2, flat = (dm == 0.f);
    const float a6_lo = 3.f * (al - q), a6_hi = 3.f * (ar - q);
    const float ar_lo = al - a6_lo, al_hi = ar - a6_hi;
    float nal = (!lo && hi) ? al_hi : al;
    float nar = lo ? ar_lo : ar;
    float na6 = lo ? a6_lo : (hi ? a6_hi : a6);
    al = flat ? q : nal;
    ar = flat ? q : nar;
    a6 = flat ? 0.f : na6;
}

template <typename Tin>
__device__ __forceinline__ float ld(const char *base, unsigned int boff)
{
    return (float)*reinterpret_cast<const Tin *>(base + boff);
}

template <typename Tin, int NF, bool FAST>
__global__ __launch_bounds__(64) void mappm_sweep_kernel(const SweepArgs a)
{
    constexpr unsigned int ESZ = sizeof(Tin);
    const int lane = threadIdx.x;
    const int64_t wcol = a.col0 + (int64_t)blockIdx.x * 64;  // first column of the wave (uniform)
    const int64_t b = wcol / a.n_inner, c0 = wcol - b * a.n_inner;
    const int km = a.km, kn = a.kn, iv = a.iv;
    const int64_t plane = a.n_inner;
    // uniform row bases of the wave's batch, first column of the wave
    const char *pe1_row = static_cast<const char *>(a.pe1) + (b * (km + 1) * plane + c0) * ESZ;  // row L+4 (0-based), walks down
    const char *pe2_b = static_cast<const char *>(a.pe2) + (b * (kn + 1) * plane + c0) * ESZ;
    const char *q1_row[NF];
    char *q2_b[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        q1_row[f] = static_cast<const char *>(a.q1[f]) + (b * km * plane + c0) * ESZ;
        q2_b[f] = reinterpret_cast<char *>(a.q2[f]) + (b * kn * plane + c0) * 4;
    }
    const unsigned int lin = (unsigned int)lane * ESZ;     // lane offset inside a source row
    const unsigned int row_in = (unsigned int)plane * ESZ;  // bytes between levels of the inputs (host: < 2^32 / levels)
    const unsigned int row_out = (unsigned int)plane * 4u;
    const float r3 = 1.f / 3.f, r23 = 2.f / 3.f;
    const int km1 = km - 1;

    // ---- head of the column: pe1(1..5), q1(1..4) ----
    float pe_a = ld<Tin>(pe1_row, lin), pe_b = ld<Tin>(pe1_row + row_in, lin), pe_c = ld<Tin>(pe1_row + 2 * (size_t)row_in, lin),
          pe_d = ld<Tin>(pe1_row + 3 * (size_t)row_in, lin), pe_e = ld<Tin>(pe1_row + 4 * (size_t)row_in, lin);
    const float pe1_top = pe_a, pe1_bot = ld<Tin>(pe1_row + (size_t)km * row_in, lin);
    pe1_row += 5 * (size_t)row_in;  // -> pe1(6)
    float q0[NF], qp1[NF], qp2[NF], qp3[NF], q_top[NF], q_bot[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        q0[f] = ld<Tin>(q1_row[f], lin);
        qp1[f] = ld<Tin>(q1_row[f] + row_in, lin);
        qp2[f] = ld<Tin>(q1_row[f] + 2 * (size_t)row_in, lin);
        qp3[f] = ld<Tin>(q1_row[f] + 3 * (size_t)row_in, lin);
        q_top[f] = q0[f];
        q_bot[f] = ld<Tin>(q1_row[f] + (size_t)km1 * row_in, lin);
        q1_row[f] += 4 * (size_t)row_in;  // -> q1(5)
    }
    bool bad = !(pe_b >= pe_a) | !(pe_c >= pe_b) | !(pe_d >= pe_c) | !(pe_e >= pe_d);
    float d0 = pe_b - pe_a, dp1 = pe_c - pe_b, dp2 = pe_d - pe_c, dp3 = pe_e - pe_d;  // dp(L..L+3)

    // One level kk of the reconstruction: dc(kk) and al(kk) = a4(2,kk) (mappm.f90:658-683) from dp(kk-2..kk+1) =
    // (z, x, y, w), q(kk-1..kk+1) and dc(kk-1).  The pressure-only terms are formed once for all fields; d4c and its
    // reciprocal are next level's d4b.
    float d4b = d0 + dp1;  // d4(2) = dp(1) + dp(2): becomes d4(kk) of the first reconstructed level below
    auto level = [&](float z, float x, float y, float w, float d4b_, const Den<FAST> r4b, const float *qa, const float *qb,
                     const float *qc, const float *dca, float *dcb, float *alb, float &d4c_out, Den<FAST> &r4c_out) {
        const float d4a = z + x, d4c = y + w;
        const Den<FAST> r4c(d4c);
        const float c1 = r4c.under(x + 0.5f * y);          // (dp(k-1) + 0.5 dp(k)) / d4(k+1)
        const float c2 = r4b.under(w + 0.5f * y);          // (dp(k+1) + 0.5 dp(k)) / d4(k)
        const Den<FAST> r3p(d4b_ + w);                     // d4(k) + dp(k+1)
        const float a1 = Den<FAST>(d4b_ + x).under(d4a);   // d4(k-1) / (d4(k) + dp(k-1))
        const float a2 = Den<FAST>(d4b_ + y).under(d4c);   // d4(k+1) / (d4(k) + dp(k))
        const float g = Den<FAST>(d4a + d4c).under(2.f);   // 2 / (d4(k-1) + d4(k+1))
        const float a12 = a1 - a2;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float df2 = r3p.under(y * (c1 * (qc[f] - qb[f]) + c2 * (qb[f] - qa[f])));
            const float dc = s_sign(s_min3(fabsf(df2), s_max3(qa[f], qb[f], qc[f]) - qb[f], qb[f] - s_min3(qa[f], qb[f], qc[f])), df2);
            const float c1f = r4b.under((qb[f] - qa[f]) * x);
            alb[f] = qa[f] + c1f + g * (y * (c1f * a12 + a2 * dca[f]) - x * a1 * dc);
            dcb[f] = dc;
        }
        d4c_out = d4c;
        r4c_out = r4c;
    };

    // ---- prologue: dc(2), dc(3), al(3), then the top boundary (mappm.f90:689-725) ----
    float al0[NF], al1[NF], al2[NF], dc0[NF], dc1[NF], dc2[NF], ar_km[NF];
    float d4c = dp1 + dp2;  // d4(3)
    Den<FAST> r4c(d4c);
    {
        // dc(2): the dc part of a level with (x, y, w) = (dp(1), dp(2), dp(3)); its al is not used
        const Den<FAST> r4b(d4b);
        const float c1 = r4c.under(d0 + 0.5f * dp1);
        const float c2 = r4b.under(dp2 + 0.5f * dp1);
        const Den<FAST> r3p(d4b + dp2);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float df2 = r3p.under(dp1 * (c1 * (qp2[f] - qp1[f]) + c2 * (qp1[f] - q0[f])));
            dc1[f] = s_sign(s_min3(fabsf(df2), s_max3(q0[f], qp1[f], qp2[f]) - qp1[f], qp1[f] - s_min3(q0[f], qp1[f], qp2[f])), df2);
        }
    }
    {
        float dc_3[NF], al_3[NF], d4n;
        Den<FAST> r4n(1.f);
        level(d0, dp1, dp2, dp3, d4c, r4c, qp1, qp2, qp3, dc1, dc_3, al_3, d4n, r4n);  // kk = 3 (recomputed by the loop's L = 1)
        const float d1 = d0, d2 = dp1;
        const Den<FAST> r12(d1 + d2);
        const Den<FAST> rcub(d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float qm = r12.under(d2 * q0[f] + d1 * qp1[f]);
            const float dq = r12.under(2.f * (qp1[f] - q0[f]));
            const float c1 = rcub.under(4.f * (al_3[f] - qm - d2 * dq));
            const float c3 = dq - 0.5f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
            float a2 = qm - 0.25f * c1 * d1 * d2 * (d2 + 3.f * d1);
            float a1 = d1 * (2.f * c1 * (d1 * d1) - c3) + a2;
            a2 = s_max2(a2, s_min2(q0[f], qp1[f]));
            a2 = s_min2(a2, s_max2(q0[f], qp1[f]));
            dc0[f] = 0.5f * (a2 - q0[f]);
            if (iv == 0) {
                a1 = s_max2(0.f, a1);
                a2 = s_max2(0.f, a2);
            } else if (iv == -1) {
                if (a1 * q0[f] <= 0.f) a1 = 0.f;
            } else if (iv == 2 || iv == -2) {
                a1 = q0[f];
            }
            al0[f] = a1;
            al1[f] = a2;
            al2[f] = 0.f;
            dc2[f] = 0.f;
            ar_km[f] = 0.f;
        }
    }
    // the loop's first level is kk = 3: its d4(kk) = d4(3), already in (d4c, r4c)
    d4b = d4c;
    Den<FAST> r4b = r4c;

    // ---- per-lane target cursor ----
    // A lane consumes the target interfaces at its own pace.  Loading pe2(k+1) where it is needed would put a memory
    // round trip (and, vmcnt being one in-order counter, a wait for everything else in flight) into the emit code, so
    // the wave keeps a ring of kRing interface rows in LDS, filled UNIFORMLY: row `jr` is requested while lane 0's
    // cursor is within kAhead rows of it (at most two rows per source layer, scalar bookkeeping) and written to the ring
    // at the top of the next iteration.  A lane whose cursor has left the window [jl - kRing, jl) -- grids whose lanes
    // drift apart by more than the ring holds -- reads its interface from memory instead: slower, same value.
    constexpr int kRing = 16, kAhead = 9;
    __shared__ float ring_lds[kRing * 64];
    float *ring = ring_lds + lane;
    const unsigned int lane2 = (unsigned int)lane * ESZ;
    int jl, jr;  // interface rows (0-based) [0, jl) have landed in the ring, [jl, jr) are in flight (pv0, pv1)
    {
        float tmp[kRing];
#pragma unroll
        for (int i = 0; i < kRing; ++i) tmp[i] = ld<Tin>(pe2_b, (unsigned int)(i <= kn ? i : kn) * row_in + lane2);
#pragma unroll
        for (int i = 0; i < kRing; ++i) ring[i * 64] = tmp[i];
        jl = jr = (kRing < kn + 1) ? kRing : kn + 1;
    }
    float pv0 = 0.f, pv1 = 0.f;
    auto PE2 = [&](int i) -> float {  // interface row i (0-based, <= kn)
        // (the LDS read is unconditional and the rare memory read sits in a branch of its own that waits for it right
        // there: a select between the two addresses becomes a flat load, and a wait placed after the branches merge
        // would stall every lane on everything the wave has in flight)
        float v = ring[(i & (kRing - 1)) * 64];
        if (!((unsigned int)(jl - 1 - i) < (unsigned int)kRing)) {
            v = ld<Tin>(pe2_b, (unsigned int)i * row_in + lane2);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }
        return v;
    };
    unsigned int offq = (unsigned int)lane * 4u;  // offset of q2(k)
    int k = 1;
    float p2k = PE2(0), p2k1 = PE2(1);  // pe2(k), pe2(k+1)   (kn >= 1)
    bool accum = false;
    float qsum[NF], dpsum = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) qsum[f] = 0.f;
    auto advance = [&]() {
        ++k;
        if (!(p2k1 >= p2k)) bad = true;  // also catches NaN
        p2k = p2k1;
        p2k1 = PE2(k <= kn ? k : kn);
        offq += row_out;
    };
    // ---- results leave through a second LDS ring ----
    // A lane finishes its targets at its own pace: stored directly, every q2 store is a scatter of 4-byte pieces over
    // as many rows as the wave's lanes are spread over, and (vmcnt being one in-order counter for loads and stores) it is
    // waited for by the next wait that guards the row loads.  Measured at C384, 4 fields: direct stores 2.18 ms, the
    // same stores deferred to the top of the next iteration 1.97 ms, no stores at all 0.80 ms -- the scatter itself is
    // what costs.  So a lane parks target row r = k - 1 in slot r % kOut of its column of the ring, and the WAVE writes
    // row `rf` out -- one coalesced row per field, scalar address -- once every lane is past it, at the top of an
    // iteration, right after the wait.  `rl` (per lane) = rows [0, rl) of this lane no longer live in the ring: a lane
    // that runs kOut rows ahead of the slowest one writes its own oldest row out first.  Ill-formed lanes count as past
    // every row; whatever the flush writes for them is overwritten by the fallback pass.
#ifndef SWEEP_KOUT
#define SWEEP_KOUT 8
#endif
    constexpr int kOut = SWEEP_KOUT;
    __shared__ float oring_lds[NF * kOut * 64];
    float *oring = oring_lds + lane;
    int rf = 0;  // uniform: rows [0, rf) are in memory for every lane
    int rl = 0;  // per lane (>= rf where it matters)
    float out_v[NF];
    auto OUT = [&](int f, float v) { out_v[f] = v; };
    auto out_end = [&]() {  // after the OUTs of target k (before advance())
        const int r = k - 1, slot = r & (kOut - 1);
        if (r - kOut >= (rl > rf ? rl : rf)) {  // the slot still holds this lane's row r - kOut: write it out now
#pragma unroll
            for (int f = 0; f < NF; ++f)
                *reinterpret_cast<float *>(q2_b[f] + (offq - (unsigned int)kOut * row_out)) = oring[(f * kOut + slot) * 64];
            rl = r - kOut + 1;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) oring[(f * kOut + slot) * 64] = out_v[f];
    };
    auto flush_rows = [&](int max_rows) {  // uniform: rows every lane has written go to memory
#pragma unroll 1
        for (int n = 0; n < max_rows && rf < kn; ++n) {
            const bool past = bad | (k - 1 > rf);
            if (__builtin_amdgcn_ballot_w64(past) != __builtin_amdgcn_ballot_w64(true)) break;
            if (rl <= rf) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    *reinterpret_cast<float *>(q2_b[f] + (size_t)rf * row_out + (unsigned int)lane * 4u) =
                        oring[(f * kOut + (rf & (kOut - 1))) * 64];
            }
            ++rf;
        }
    };
    if (!(p2k1 >= p2k)) bad = true;
    while (k <= kn && !bad && p2k <= pe1_top) {  // targets that start at or above the old top (mappm.f90:62-64)
#pragma unroll
        for (int f = 0; f < NF; ++f) OUT(f, q_top[f]);
        out_end();
        advance();
    }
    bool live = (k <= kn) && !bad && !(p2k >= pe1_bot);

    float q_in[NF], pe_in = pe_e;
#pragma unroll
    for (int f = 0; f < NF; ++f) q_in[f] = 0.f;
    for (int L = 1; L <= km; ++L) {
        // ---- the interface rows requested during the previous iteration land in the ring ----
        if (jr > jl) ring[(jl & (kRing - 1)) * 64] = pv0;
        if (jr > jl + 1) ring[((jl + 1) & (kRing - 1)) * 64] = pv1;
        jl = jr;
        flush_rows(2);
        if (L > 1) {  // level L becomes the current one
            if (!(pe_in >= pe_e)) bad = true;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                q0[f] = qp1[f]; qp1[f] = qp2[f]; qp2[f] = qp3[f]; qp3[f] = q_in[f];
                al0[f] = al1[f]; al1[f] = al2[f];
                dc0[f] = dc1[f]; dc1[f] = dc2[f];
            }
            d0 = dp1; dp1 = dp2; dp2 = dp3; dp3 = pe_in - pe_e;
            pe_a = pe_b; pe_b = pe_c; pe_c = pe_d; pe_d = pe_e; pe_e = pe_in;
        }
        // Order inside an iteration: the requests for the next iteration first, then layer L's finalisation and its
        // emits (the q2 stores), then the reconstruction of level L + 2 -- vmcnt is one in-order counter for loads and
        // stores, so the wait at the top of the next iteration also waits for these stores: with the reconstruction
        // behind them they have that long to complete instead of no time at all.
        // ---- requests for the next iteration: q(L+4), pe1(L+5) ----
        if (L + 4 <= km) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                q_in[f] = ld<Tin>(q1_row[f], lin);
                q1_row[f] += row_in;
            }
            pe_in = ld<Tin>(pe1_row, lin);
            pe1_row += row_in;
        }
        {   // ... and up to two target-interface rows, kept kAhead ahead of lane 0's cursor
            const int want = __builtin_amdgcn_readfirstlane(k) + kAhead;
            if (jr <= kn && jr < want) {
                pv0 = ld<Tin>(pe2_b + (size_t)jr * row_in, lane2);
                ++jr;
                if (jr <= kn && jr < want) {
                    pv1 = ld<Tin>(pe2_b + (size_t)jr * row_in, lane2);
                    ++jr;
                }
            }
        }
        // ---- finalise layer L: A6 and the monotonicity constraint (mappm.f90:773-849 with lmt = 0) ----
        float al[NF], ar[NF], a6[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            al[f] = al0[f];
            ar[f] = (L == km) ? ar_km[f] : al1[f];
            a6[f] = 3.f * (2.f * q0[f] - (al[f] + ar[f]));
            limit0(dc0[f], q0[f], al[f], ar[f], a6[f]);
        }
        const float pL = pe_a, pL1 = pe_b;
        const Den<FAST> rd0(d0);

        // ---- emit the target layers that end inside layer L (see mappm_merge_kernel for the event order) ----
        if (live && accum && !(p2k1 > pL1)) {
            const float delp = p2k1 - pL;
            const float PR = rd0.under(delp);
            dpsum = dpsum + delp;
            const Den<FAST> rs(dpsum);
    #pragma unroll
            for (int f = 0; f < NF; ++f) {
                qsum[f] = qsum[f] + delp * (al[f] + 0.5f * PR * (ar[f] - al[f] + a6[f] * (1.f - r23 * PR)));
                OUT(f, rs.under(qsum[f]));
            }
            out_end();
            accum = false;
            advance();
            live = (k <= kn) && !bad && !(p2k >= pe1_bot);
        }
        while (live && !accum && (p2k >= pL && p2k <= pL1) && (p2k1 <= pL1)) {
            const float PR = rd0.under(p2k1 - pL);
            const float PL = rd0.under(p2k - pL);
            const float TT = r3 * (PR * (PR + PL) + PL * PL);
    #pragma unroll
            for (int f = 0; f < NF; ++f) OUT(f, al[f] + 0.5f * (a6[f] + ar[f] - al[f]) * (PR + PL) - a6[f] * TT);
            out_end();
            advance();
            live = (k <= kn) && !bad && !(p2k >= pe1_bot);
        }
        if (live) {
            if (accum) {  // whole layer (mappm.f90:99-104)
#pragma unroll
                for (int f = 0; f < NF; ++f) qsum[f] = qsum[f] + d0 * q0[f];
                dpsum = dpsum + d0;
            } else if (p2k >= pL && p2k <= pL1) {  // fractional area (mappm.f90:85-92)
                const float PL = rd0.under(p2k - pL);
                const float delp = pL1 - p2k;
                const float TT = r3 * (1.f + PL * (1.f + PL));
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    qsum[f] = delp * (al[f] + 0.5f * (a6[f] + ar[f] - al[f]) * (1.f + PL) - a6[f] * TT);
                dpsum = delp;
                accum = true;
            }
        }
        // ---- reconstruction of level kk = L + 2 from dp(L..L+3), q(L+1..L+3) ----
        const int kk = L + 2;
        if (kk <= km1) {
            float d4n;
            Den<FAST> r4n(1.f);
            level(d0, dp1, dp2, dp3, d4b, r4b, qp1, qp2, qp3, dc1, dc2, al2, d4n, r4n);
            d4b = d4n;
            r4b = r4n;
        } else if (kk == km) {
            // bottom boundary (mappm.f90:729-761): al(km), ar(km), dc(km) from al(km-1) = al1
            const float d1 = dp2, d2 = dp1;  // dp(km), dp(km-1)
            const Den<FAST> r12(d1 + d2);
            const Den<FAST> rcub(d2 * (2.f * d2 * d2 + d1 * (d2 + 3.f * d1)));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float qk = qp2[f], qk1 = qp1[f];  // q(km), q(km-1)
                const float qm = r12.under(d2 * qk + d1 * qk1);
                const float dq = r12.under(2.f * (qk1 - qk));
                const float c1 = rcub.under(al1[f] - qm - d2 * dq);
                const float c3 = dq - 2.0f * c1 * (d2 * (5.f * d1 + d2) - 3.f * d1 * d1);
                float alk = qm - c1 * d1 * d2 * (d2 + 3.f * d1);
                float ark = d1 * (8.f * c1 * (d1 * d1) - c3) + alk;
                alk = s_max2(alk, s_min2(qk, qk1));
                alk = s_min2(alk, s_max2(qk, qk1));
                dc2[f] = 0.5f * (qk - alk);
                if (iv == 0) {
                    alk = s_max2(0.f, alk);
                    ark = s_max2(0.f, ark);
                } else if (iv < 0) {
                    if (qk * ark <= 0.f) ark = 0.f;
                }
                al2[f] = alk;
                ar_km[f] = ark;
            }
        }
    }

    // ---- past the old surface (mappm.f90:115-121), then the run that copies q1(km) ----
    if (k <= kn && !bad && accum) {
        const float delp = p2k1 - pe1_bot;
        if (delp > 0.f) {
#pragma unroll
            for (int f = 0; f < NF; ++f) qsum[f] = qsum[f] + delp * q_bot[f];
            dpsum = dpsum + delp;
        }
        const Den<FAST> rs(dpsum);
#pragma unroll
        for (int f = 0; f < NF; ++f) OUT(f, rs.under(qsum[f]));
        out_end();
        advance();
    }
    while (k <= kn && !bad) {
        if (p2k >= pe1_bot) {
    #pragma unroll
            for (int f = 0; f < NF; ++f) OUT(f, q_bot[f]);
            out_end();
            advance();
        } else {
            bad = true;  // a top-edge search that no source layer satisfied
        }
    }
    flush_rows(kn);  // every lane is done (or ill-formed): the rows still in the ring
    if (bad) a.bad_cols[atomicAdd(a.n_bad, 1u)] = (unsigned int)(blockIdx.x * 64 + lane);  // redone by mappm_fallback_kernel
}

template <typename Tin, int NF>
void launch_sweep2(const SweepArgs &a, int64_t n_waves, bool fast, hipStream_t st)
{
    if (fast)
        hipLaunchKernelGGL((mappm_sweep_kernel<Tin, NF, true>), dim3((unsigned)n_waves), dim3(64), 0, st, a);
    else
        hipLaunchKernelGGL((mappm_sweep_kernel<Tin, NF, false>), dim3((unsigned)n_waves), dim3(64), 0, st, a);
}

template <typename Tin>
void launch_sweep1(const SweepArgs &a, int nf, int64_t n_waves, bool fast, hipStream_t st)
{
    switch (nf) {
        case 1: launch_sweep2<Tin, 1>(a, n_waves, fast, st); break;
        case 2: launch_sweep2<Tin, 2>(a, n_waves, fast, st); break;
        case 3: launch_sweep2<Tin, 3>(a, n_waves, fast, st); break;
        default: launch_sweep2<Tin, 4>(a, n_waves, fast, st); break;
    }
}

}  // namespace

bool mappm_sweep_eligible(int64_t n_inner, int km, int kn, int kord, int layout, int in_dtype)
{
    const int64_t esz = (in_dtype == FV3HIP_F64) ? 8 : 4;
    const int64_t levels = (km + 1 > kn + 1 ? km + 1 : kn + 1) + 5;
    return layout == FV3HIP_LAYOUT_LEVEL_COL && kord <= 3 && km >= 8 && kn >= 1 && n_inner > 0 && n_inner % 64 == 0 &&
           n_inner * esz * levels < ((int64_t)1 << 32);
}

void mappm_sweep_launch(const SweepArgs &a, int nf, int in_dtype, int64_t col_end, bool fast, hipStream_t st)
{
    const int64_t n_waves = (col_end - a.col0) / 64;
    if (in_dtype == FV3HIP_F32)
        launch_sweep1<float>(a, nf, n_waves, fast, st);
    else
        launch_sweep1<double>(a, nf, n_waves, fast, st);
}

}  // namespace fv3hip
