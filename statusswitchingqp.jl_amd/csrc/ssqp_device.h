// ssqp_device.h -- wavefront-level device helpers shared by the two solve kernels
// (ssqp_kernels.hip: one workgroup per QP, any shape; ssqp_wave.hip: one wavefront per QP, small free sets).
// gfx950 only: 64-wide wavefronts, DPP lane permutes, v_readlane broadcasts.
#ifndef SSQP_DEVICE_H
#define SSQP_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ssqp {

// orders the LDS/global accesses of the lanes of ONE wavefront (the wave runs in
// lockstep; this only stops the compiler from moving accesses across it and
// waits for outstanding ones)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// 1/d by v_rcp_f64 and two Newton steps (about 1 ulp; the IEEE division sequence is ~3x longer and sits on
// the critical path of every elimination step)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
// x - d*y and x/d with the reference's two roundings (no FMA contraction)
__device__ __forceinline__ double sub_mul_nc(double x, double d, double y) {
#pragma clang fp contract(off)
    const double t = d * y;
    return x - t;
}

// ---------------------------------------------------------------- reductions
// Wavefront reductions on DPP (data-parallel primitives: lane permutes inside the VALU, no LDS round
// trip): xor-1 and xor-2 by quad_perm, then row_half_mirror and row_mirror complete a 16-lane row; the two
// cross-row steps use the 64-lane shuffle.  Every lane ends with the result.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    // (every lane is written by these in-row permutes: no `old` operand, so no copy in front of the v_mov_dpp)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, false);
}
// v_max_f64 / v_min_f64 as single instructions: fmax()/fmin() put a canonicalising v_max(x, x) in front of every
// operand (signalling-NaN semantics), which doubles the length of the reduction chains.  Operands here are
// never NaN by construction (|x|, ratios already filtered, +-inf sentinels).
__device__ __forceinline__ double max_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double min_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8 lanes
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i inside each 16 lanes

// cross-row steps of a 64-lane reduction without the LDS: row_bcast15 (into rows 1 and 3) and row_bcast31 (into
// rows 2 and 3) leave the result in lane 63; v_readlane hands it to every lane as a wavefront-uniform value
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64_rows(double v, double oldv) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(oldv), lo, CTRL, ROWMASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(oldv), hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast63_f64(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<DPP_XOR1>(v);
    v += dpp_f64<DPP_XOR2>(v);
    v += dpp_f64<DPP_HALF_MIRROR>(v);
    v += dpp_f64<DPP_MIRROR>(v);
    v += dpp_f64_rows<0x142, 0xA>(v, 0.0);   // rows 1,3 += row 0,2 totals (other rows add 0)
    v += dpp_f64_rows<0x143, 0xC>(v, 0.0);   // rows 2,3 += total of rows 0-1
    return bcast63_f64(v);
}
__device__ __forceinline__ double wave_max(double v) {
    v = max_raw(v, dpp_f64<DPP_XOR1>(v));
    v = max_raw(v, dpp_f64<DPP_XOR2>(v));
    v = max_raw(v, dpp_f64<DPP_HALF_MIRROR>(v));
    v = max_raw(v, dpp_f64<DPP_MIRROR>(v));
    v = max_raw(v, dpp_f64_rows<0x142, 0xA>(v, v));
    v = max_raw(v, dpp_f64_rows<0x143, 0xC>(v, v));
    return bcast63_f64(v);
}

// maximum over lanes 0..15 only (a 16-lane DPP row: four steps, no cross-row traffic), as a uniform value
__device__ __forceinline__ double row0_max(double v) {
    v = max_raw(v, dpp_f64<DPP_XOR1>(v));
    v = max_raw(v, dpp_f64<DPP_XOR2>(v));
    v = max_raw(v, dpp_f64<DPP_HALF_MIRROR>(v));
    v = max_raw(v, dpp_f64<DPP_MIRROR>(v));
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int srcLane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max_uniform(double v) { return wave_max(v); }
// A value every lane holds identically (read from one LDS address, say) but the compiler cannot prove uniform:
// pin it to scalar registers, so that everything derived from it (loop bounds, branches, addresses) runs on the
// scalar unit instead of as per-lane arithmetic under exec masks.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}


struct KeyMin {  // minimum value, ties -> smallest order
    double v;
    int ord;
};
__device__ __forceinline__ KeyMin keymin(KeyMin a, KeyMin b) {
    const bool take = (b.v < a.v) | ((b.v == a.v) & (b.ord < a.ord));  // (no short circuit: straight-line code)
    return KeyMin{take ? b.v : a.v, take ? b.ord : a.ord};
}
template <int CTRL>
__device__ __forceinline__ KeyMin keymin_dpp(KeyMin a) {
    KeyMin b;
    b.v = dpp_f64<CTRL>(a.v);
    b.ord = dpp_i32<CTRL>(a.ord);
    return keymin(a, b);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ KeyMin keymin_rows(KeyMin a) {
    KeyMin b;
    b.v = dpp_f64_rows<CTRL, ROWMASK>(a.v, a.v);
    b.ord = __builtin_amdgcn_update_dpp(a.ord, a.ord, CTRL, ROWMASK, 0xF, false);
    return keymin(a, b);
}
__device__ __forceinline__ double wave_min(double v) {
    v = min_raw(v, dpp_f64<DPP_XOR1>(v));
    v = min_raw(v, dpp_f64<DPP_XOR2>(v));
    v = min_raw(v, dpp_f64<DPP_HALF_MIRROR>(v));
    v = min_raw(v, dpp_f64<DPP_MIRROR>(v));
    v = min_raw(v, dpp_f64_rows<0x142, 0xA>(v, v));
    v = min_raw(v, dpp_f64_rows<0x143, 0xC>(v, v));
    return bcast63_f64(v);
}
// Two steps instead of a (value, order) pair through every reduction stage: the minimum value first, then the
// smallest order among the lanes that hold it (one lane unless values tie exactly).  Values must not be NaN.
__device__ __forceinline__ KeyMin wave_keymin(KeyMin a) {
    KeyMin r;
    r.v = wave_min(a.v);
    const bool mine = (a.v == r.v);
    unsigned long long tie = __ballot(mine);
    int o = __builtin_amdgcn_readlane(a.ord, __ffsll((long long)tie) - 1);
    tie &= tie - 1;
    if (tie) {  // exact ties (or a wavefront without any candidate, all at +inf): integer minimum over the tied lanes
        int q = mine ? a.ord : 0x7fffffff;
        q = min(q, dpp_i32<DPP_XOR1>(q));
        q = min(q, dpp_i32<DPP_XOR2>(q));
        q = min(q, dpp_i32<DPP_HALF_MIRROR>(q));
        q = min(q, dpp_i32<DPP_MIRROR>(q));
        q = min(q, __builtin_amdgcn_update_dpp(q, q, 0x142, 0xA, 0xF, false));
        q = min(q, __builtin_amdgcn_update_dpp(q, q, 0x143, 0xC, 0xF, false));
        o = __builtin_amdgcn_readlane(q, 63);
    }
    r.ord = o;
    return r;
}

// lambda of the Schur system H lam = s (W <= WM <= 11), H symmetric (lower part given), one wavefront:
// lane i holds row i of H and s_i; eliminations broadcast the pivot row with v_readlane.  The unit-lower factor
// is also written to `tr` (WM*WM doubles of LDS scratch) column by column, so that the back substitution reads
// column c of L into lane c and needs one broadcast per step instead of one per entry.  Returns false when a
// pivot is not > 0 (the reference's cholesky(C) throws).  On return lane w < W holds lam_w.
template <int WM>
__device__ __forceinline__ bool small_spd_solve(const double *H, const double *rhs_, int W, double &lam, double *tr) {
    const int lane = threadIdx.x & 63;
    double a[WM];
#pragma unroll
    for (int c = 0; c < WM; ++c) {
        const int r = lane < W ? lane : 0, cc = c < W ? c : 0;
        const double v = (r >= cc) ? H[r + W * cc] : H[cc + W * r];  // symmetric read from the lower part
        a[c] = (lane < W && c < W) ? v : 0.0;
    }
    double y = (lane < W) ? rhs_[lane] : 0.0;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < WM; ++c) {
        if (c < W) {  // uniform
            const double d = readlane_f64(a[c], c);
            if (!(d > 0.0)) ok = false;
            const double r = fast_rcp(d);
            const double yc = readlane_f64(y, c);
            const double lic = a[c] * r;  // L(i,c) for this lane's row i
            const bool below = lane > c;
            y = below ? fma(-lic, yc, y) : ((lane == c) ? y * r : y);  // forward substitution rides along; D^-1 on row c
#pragma unroll
            for (int c2 = 0; c2 < WM; ++c2) {  // (columns beyond W hold zeros: no guard on W)
                if (c2 > c) {  // (rows <= c are finished and never read again: no predicate)
                    const double bq = readlane_f64(a[c], c2);
                    a[c2] = fma(-lic, bq, a[c2]);
                }
            }
            if (below && lane < WM) tr[c * WM + lane] = lic;
        }
    }
    wave_sync();
    // y = D^-1 L^-1 s ; x = L'^-1 y with u[i] = L(i, lane)
    double u[WM];
#pragma unroll
    for (int i = 1; i < WM; ++i) u[i] = tr[(lane < WM ? lane : 0) * WM + i];
#pragma unroll
    for (int i = WM - 1; i >= 1; --i) {
        if (i < W) {  // uniform
            const double xi = readlane_f64(y, i);
            y = (lane < i) ? fma(-u[i], xi, y) : y;
        }
    }
    lam = y;
    return ok;
}

}  // namespace ssqp
#endif
