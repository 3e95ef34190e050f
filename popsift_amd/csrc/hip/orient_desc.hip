// orient_desc.hip -- orientation assignment, orientation scan + feature preparation, and the
// 4x4x8 "loop" descriptor with fused normalisation, for gfx950.
//
// Replaces (behaviour):
//   ori_par            s_orientation.cu:75-259   -> k_orientation   (one wave64 per extremum)
//   ori_prefix_sum     s_orientation.cu:320-362  -> k_scan          (+ prep_features, sift_pyramid.cu:250-280)
//   ext_desc_loop      s_desc_loop.cu:19-158     -> k_descriptors   (one wave64 per descriptor)
//   normalize_histogram s_desc_normalize.h:14-33 -> fused into k_descriptors
//
// wave64 re-design notes:
//   * one launch covers all octaves: each wave looks its extremum up through the per-octave
//     counters, so the host never reads a counter between stages (the reference does four blocking
//     symbol copies and three device-wide syncs per image);
//   * histograms are 18.14 unsigned fixed point accumulated with ds_add_u32 (ds_add_f32 is ~30x slower
//     on gfx950, tools/ubench/lds_atomic.hip; integer sums do not depend on arrival order): 36 bins x 8
//     copies per wave for the orientation, a padded 4x4x8 tile grid x 4 copies for the descriptor;
//     smoothing and the parabola fit run on lanes 0..35 with __shfl;
//   * the 64-candidate bitonic sort becomes four rounds of wave-max selection;
//   * descriptor: every pixel of the rotated 5x5-SBP window is visited ONCE (gradient, magnitude,
//     angle once) and scattered trilinearly into 2x2 tiles x 2 bins; the reference visits it from each
//     of its 16 tile scans;
//   * both gather kernels are VALU-issue bound: per-wave state lives in SGPRs (readfirstlane), loads use
//     a scalar plane base + one 32-bit offset, magnitude / angle / weights use v_sqrt_f32, v_rcp_f32,
//     v_exp_f32 and a degree-13 atan polynomial (the reference: hypotf/atan2f for the orientation,
//     __fdividef/__expf for the descriptor).
#include "psx_internal.h"

#include <cstdlib>

namespace {

constexpr int NT = 256;
constexpr int WPB = NT / PSX_WAVE;       // waves per block
constexpr int ORI_NBINS = 36;
constexpr int HCOPIES = 8;
// sift_constants.h:21-33: float constants
constexpr float PI_F  = 3.14159265358979323846f;
constexpr float PI2_F = 2.0f * 3.14159265358979323846f;
constexpr float M_4RPI_F = 4.0f / PI_F;
constexpr float ORI_WINFACTOR = 1.5f;
constexpr float DESC_MAGNIFY = 3.0f;

__device__ __forceinline__ void wave_fence()
{
    // LDS traffic of one wave is executed in order; this only stops compiler reordering
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Fixed-point accumulation.  Measured on gfx950 (tools/ubench/lds_atomic.hip): ds_add_f32 is
// serialised per lane (~190 clk per wave-instruction), ds_add_u64 runs ~29x faster.  Histogram
// weights are non-negative, so they are accumulated as unsigned 32.32 fixed point with
// ds_add_u64: integer addition is associative, which also makes the histograms independent of
// the order in which lanes and waves arrive (the reference's shared-memory float atomicAdd,
// s_orientation.cu:159, is order dependent).
typedef unsigned long long fix64;
// A single contribution is grad * weight <= |gradient| <= 255*sqrt(2) < 512, so 9.23 fixed point fits
// one v_cvt_u32_f32; the 64-bit accumulator leaves 2^32 of headroom for the sum.
constexpr float FIX_SCALE = 8388608.0f;           // 2^23
__device__ __forceinline__ fix64 to_fix(float w) { return (fix64)(unsigned)(w * FIX_SCALE); }
__device__ __forceinline__ float from_fix(fix64 v) { return __ull2float_rn(v) * (1.0f / FIX_SCALE); }

#define LDS_AS __attribute__((address_space(3)))
// ds_add_u64 without return value on an LDS-address-space pointer (immediate offsets fold into the instruction)
__device__ __forceinline__ void lds_add(fix64 LDS_AS* p, fix64 v)
{
#ifdef PSX_MODEL_NOATOMIC
    // measurement build only (tools/polar_patch_model.py): the operands are formed, the ds_add_u64 is not issued
    asm volatile("" :: "v"(p), "v"(v));
#else
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
__device__ __forceinline__ void lds_add(unsigned LDS_AS* p, unsigned v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Butterfly over the 64 lanes, partner lane ^ K for K = 32, 16, 8, 4, 2, 1 -- the order and the operand pairs of
// "v = op(v, __shfl_xor(v, K))", hence the same bits, without the six VALU of address arithmetic that every __shfl_xor
// (ds_bpermute) carries: v_permlane32_swap (gfx950) for the half-wave exchange, ds_swizzle's lane-swap patterns for
// 16 / 8 / 4 (no address operand), DPP quad_perm for 2 / 1.
template <int K>
__device__ __forceinline__ float lane_xor(float v)
{
    static_assert(K == 16 || K == 8 || K == 4 || K == 2 || K == 1, "lane_xor");
    if constexpr (K == 2) return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    else if constexpr (K == 1) return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    else return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (K << 10) | 0x1f));                      // bit-mask mode: lane ^ K
}
__device__ __forceinline__ float wave_max(float v)
{
    // lanes 0..31 of the pair hold (v[i], v[i + 32]), lanes 32..63 (v[i - 32], v[i]): op of the two is op(v[i], v[i ^ 32]) everywhere
    const auto h = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = fmaxf(__int_as_float(h[0]), __int_as_float(h[1]));
    v = fmaxf(v, lane_xor<16>(v)); v = fmaxf(v, lane_xor<8>(v)); v = fmaxf(v, lane_xor<4>(v));
    v = fmaxf(v, lane_xor<2>(v));  v = fmaxf(v, lane_xor<1>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
    const auto h = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = __int_as_float(h[0]) + __int_as_float(h[1]);
    v += lane_xor<16>(v); v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
    return v;
}

// float -> 64-bit fixed point for k_orientation's out-of-range case (a function so that it stays out of the common path)
__device__ __noinline__ fix64 wide_fix(float v) { return (fix64)v; }

// atan2 on the gradient: degree-13 odd minimax polynomial for atan (max error 3.3e-7 rad), v_rcp_f32
// atan2 evaluated in double and rounded once (what oracle/sift_oracle.c::atan2f_1r does): only on the rare exact path
__device__ __noinline__ float atan2_1r(float y, float x) { return (float)atan2((double)y, (double)x); }

__device__ __forceinline__ float fast_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1e-30f));     // v_rcp_f32, 1 ulp
    const float s = a * a;
    float r = 0.006811792496591806f;
    r = fmaf(r, s, -0.0336042195558548f);
    r = fmaf(r, s, 0.07962366938591003f);
    r = fmaf(r, s, -0.1323334127664566f);
    r = fmaf(r, s, 0.19807815551757812f);
    r = fmaf(r, s, -0.3331736922264099f);
    r = fmaf(r, s, 0.9999961256980896f);
    r = r * a;
    if (ay > ax) r = 1.57079632679489662f - r;
    if (x < 0.0f) r = PI_F - r;
    return (y < 0.0f) ? -r : r;
}

// x / 3.0f, correctly rounded, in three instructions: q = RN(x * RN(1/3)), r = x - 3q (exact in an fma), q + r * RN(1/3).
// (Markstein: with a correctly rounded reciprocal one such correction step gives the correctly rounded quotient;
// checked against x / 3.0f on 2e7 random floats over nine decades.)  The IEEE division sequence costs ~10.
__device__ __forceinline__ float div3(float x)
{
    const float y = 0.3333333432674407958984375f;
    const float q = x * y;
    const float r = fmaf(-3.0f, q, x);
    return fmaf(r, y, q);
}

typedef const __attribute__((address_space(1))) float* gfloat_p;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) v2f* gv2f_p;
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { return (v2f){a, a}; }

// clamped per-octave extrema count (find_extrema_in_dog's atomicMin, s_extrema.cu:553)
__device__ __forceinline__ int ext_count(const PsxParams* P, const PsxCounters* cnt, int o)
{
    return min(cnt->ext_ct[o], P->max_extrema);
}

// ---------------------------------------------------------------------------------------------
// Orientation
// ---------------------------------------------------------------------------------------------
template <int WB>
__global__ __launch_bounds__(PSX_WAVE * WB) void k_orientation(const PsxParams* __restrict__ P, const PsxCounters* cnt)
{
    // 41.23 fixed-point bins (ds_add_u64): the parabola fit through the histogram peak is ill conditioned
    // for low-contrast keypoints, so the bins keep (almost) the full float precision of every weight
    // (18.14 bins in 32 bits moved 2 of 74 000 orientations by up to 1e-3 rad in tools/fuzz_sweep.py)
    constexpr float OFIX = 8388608.0f;
    __shared__ fix64 s_hist[WB][HCOPIES * ORI_NBINS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    fix64* hist = s_hist[wave];

    int total = 0;
    for (int o = 0; o < P->num_octaves; o++) total += ext_count(P, cnt, o);
    if (total > P->ext_capacity) total = P->ext_capacity;

    const int nwaves = gridDim.x * WB;
    for (int ev = blockIdx.x * WB + wave; ev < total; ev += nwaves) {
        const int e = __builtin_amdgcn_readfirstlane(ev);          // wave uniform: scalar loads below
        int o = 0, base = 0;
        for (;;) {
            const int c = ext_count(P, cnt, o);
            if (e < base + c) break;
            base += c; o++;
        }
        const PsxOctave oc = P->oct[o];
        const int w = oc.w, h = oc.h;
        const psx_iext ie = P->iext[o][P->iext_off[o][e - base]];

        for (int i = lane; i < HCOPIES * ORI_NBINS; i += PSX_WAVE) hist[i] = 0ull;
        wave_fence();

        const float x = ie.xpos, y = ie.ypos;
        const int   level = psx_clampi(ie.lpos, 0, P->L - 1);
        const float sig = ie.sigma;
        // the keypoint record arrives through vector loads: put the (wave-uniform) plane address back into scalar registers, so
        // that the gradient loads are "scalar base + 32-bit lane offset"
        const unsigned long long pb = (unsigned long long)(uintptr_t)(oc.data + (size_t)level * oc.plane);
        const unsigned pb_lo = __builtin_amdgcn_readfirstlane((unsigned)pb);
        const unsigned pb_hi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
        const char* plane = reinterpret_cast<const char*>((uintptr_t)(((unsigned long long)pb_hi << 32) | pb_lo));
        const unsigned pitch4 = (unsigned)oc.pitch * 4u;

        const float sigw = ORI_WINFACTOR * sig;
        const int   rad  = (int)roundf((3.0f * sigw));
        const float factor = -0.5f / (sigw * sigw);
        const float factor2 = factor * 1.4426950408889634f;        // exp(s*factor) = exp2(s*factor2)
        const int   sq_thres = rad * rad;

        const int xmin = max(1,     (int)roundf(x) - rad);
        const int xmax = min(w - 2, (int)roundf(x) + rad);
        const int ymin = max(1,     (int)roundf(y) - rad);
        const int ymax = min(h - 2, (int)roundf(y) + rad);
        const int wx = xmax - xmin + 1;
        const int hy = ymax - ymin + 1;
        const int loops = (wx > 0 && hy > 0) ? wx * hy : 0;

        fix64* myhist = hist + (lane & (HCOPIES - 1)) * ORI_NBINS;
        // a lane owns the pixel pair (xx, xx+1), xx even: packed f32 math for everything that is the same
        // operation on both pixels (gradient, squared distance, the atan polynomial), one 8-byte load per row.
        // Per-sample values are those of the one-pixel-per-lane loop, and the histogram adds are integer, so
        // the result is bit-identical.
        const int xs = xmin & ~1;
        const int pw = (wx > 0) ? ((xmax - xs) >> 1) + 1 : 0;              // pairs per row
        const int loops2 = (pw > 0 && hy > 0) ? pw * hy : 0;
        const float rcp_pw = __builtin_amdgcn_rcpf((float)max(pw, 1));       // 1 ulp: (i + 0.5) * rcp_pw stays >= 0.5 / pw - 4e-6 away from an integer
        for (int i = lane; i < loops2; i += PSX_WAVE) {
            // i / pw without integer division: (i+0.5)/pw is >= 0.5/pw away from an integer
            const int q = (int)(((float)i + 0.5f) * rcp_pw);
            const int yy = q + ymin;
            const int xx = 2 * (i - __mul24(q, pw)) + xs;
            // uniform plane base + 32-bit byte offset: global_load with scalar base
            const unsigned off = __umul24((unsigned)yy, pitch4) + (unsigned)xx * 4u;   // rows and pitch bytes < 2^24: v_mul_u32_u24 (v_mul_lo_u32 is quarter rate)
            const v2f  ctr = *(gv2f_p)(plane + off);                  // p[xx], p[xx+1]
            const float lft = *(gfloat_p)(plane + off - 4u);          // p[xx-1]
            const float rgt = *(gfloat_p)(plane + off + 8u);          // p[xx+2]
            const v2f  dwn = *(gv2f_p)(plane + (off + pitch4));
            const v2f  upp = *(gv2f_p)(plane + (off - pitch4));
#ifdef PSX_MODEL_NOGRAD
            // measurement build only (tools/polar_patch_model.py): what the kernel costs when magnitude and angle of a pixel
            // pair come out of ONE 8-byte read (a per-keypoint polar-gradient patch would deliver them that way)
            const v2f gdx = ctr, gdy = (v2f){ctr.y, ctr.x};
            (void)lft; (void)rgt; (void)dwn; (void)upp;
#else
            const v2f gdx = (v2f){ctr.y - lft, rgt - ctr.x};
            const v2f gdy = dwn - upp;
#endif
            // the reference uses hypotf / atan2f here (s_gradiant.h:56-69).  The magnitude only scales a weight
            // (v_sqrt_f32 is enough); the angle picks the histogram bin and must round exactly like the CPU restatement's
            // (oracle/sift_oracle.c) roundf(36 (atan2f + pi) / 2pi) -- see the bin computation below.
#ifdef PSX_MODEL_NOGRAD
            const v2f m2 = gdx;
#else
            const v2f m2 = pk_fma(gdx, gdx, gdy * gdy);
#endif
            const float dy = yy - y;
            const v2f dxv = (v2f){(float)xx, (float)(xx + 1)} - splat(x);      // each column converted, then - x: as the CPU restatement
            const v2f d2 = dxv * dxv + splat(dy * dy);
            const int sq0 = (int)d2.x, sq1 = (int)d2.y;
            const bool on0 = xx >= xmin && sq0 <= sq_thres;
            const bool on1 = xx + 1 <= xmax && sq1 <= sq_thres;
            if (on0 || on1) {
                // fast_atan2 on both pixels (same operations per component)
                const v2f ax = (v2f){fabsf(gdx.x), fabsf(gdx.y)}, ay = (v2f){fabsf(gdy.x), fabsf(gdy.y)};
#ifdef PSX_MODEL_NOGRAD
                const v2f r = gdy * splat(0.03f);
#else
                const v2f mx = (v2f){fmaxf(ax.x, ay.x), fmaxf(ax.y, ay.y)};
                const v2f mn = (v2f){fminf(ax.x, ay.x), fminf(ax.y, ay.y)};
                const v2f rc = (v2f){__builtin_amdgcn_rcpf(fmaxf(mx.x, 1e-30f)), __builtin_amdgcn_rcpf(fmaxf(mx.y, 1e-30f))};
                const v2f a = mn * rc;
                const v2f s2 = a * a;
                // the polynomial in BIN units (36 bins per turn: coefficients x 18 / pi), so that the octant fix-ups work on
                // 9 / 18 bins and "+ 18" gives the bin value directly (the fast value only has to be within 1e-5 bins of the
                // exact expression, see below)
                constexpr float KB = 5.729577951308232f;
                v2f r = splat(0.006811792496591806f * KB);
                r = pk_fma(r, s2, splat(-0.0336042195558548f * KB));
                r = pk_fma(r, s2, splat(0.07962366938591003f * KB));
                r = pk_fma(r, s2, splat(-0.1323334127664566f * KB));
                r = pk_fma(r, s2, splat(0.19807815551757812f * KB));
                r = pk_fma(r, s2, splat(-0.3331736922264099f * KB));
                r = pk_fma(r, s2, splat(0.9999961256980896f * KB));
                r = r * a;
#endif
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (!(e == 0 ? on0 : on1)) continue;
                    const float gx = e == 0 ? gdx.x : gdx.y, gy = e == 0 ? gdy.x : gdy.y;
                    float at = e == 0 ? r.x : r.y;
                    if ((e == 0 ? ay.x : ay.y) > (e == 0 ? ax.x : ax.y)) at = 9.0f - at;
                    if (gx < 0.0f) at = 18.0f - at;
                    at = (gy < 0.0f) ? -at : at;
#ifdef PSX_MODEL_NOGRAD
                    const float grad = e == 0 ? m2.x : m2.y;
#else
                    const float grad = __builtin_amdgcn_sqrtf(e == 0 ? m2.x : m2.y);
#endif
                    const float weight = grad * __builtin_amdgcn_exp2f((float)(e == 0 ? sq0 : sq1) * factor2);
                    // Bin = roundf(36 (atan2f(gdy,gdx) + pi) / 2pi) with the correctly rounded atan2 (evaluated in double,
                    // rounded once: the CPU restatement and the reference shim do the same) and IEEE division: gradients
                    // along exact bin boundaries (gdx == gdy gives 45 deg = bin 22.5) are common in smooth images and
                    // the rule in binary / symmetric ones, where the bin hangs on the last ulp of the angle; one heavy
                    // sample in the wrong bin moves the interpolated peak by ~6e-3 rad.
                    // The 3.3e-7 rad polynomial decides every sample that is not within 2e-4 bins of a boundary.  Error budget
                    // of bfast against the exact expression, in bins (1 rad = 5.73 bins): polynomial + v_rcp 4e-7 rad and the
                    // rounding of the exact angle 1.2e-7 rad -> 3e-6; the fix-ups 9 - r, 18 - r and + 18 on this side < 4e-6;
                    // RN(at + pi), RN(36 s) and the division on the exact side 1.4e-6 + 1.2e-6 + 1.9e-6: < 1.2e-5
                    // in total, 16 times below the threshold.  About 0.04 % of the samples take the exact form (two calls
                    // of ~130 VALU instructions, f64): with 128 samples per step that is 5 % of the steps (round 3's 1e-3
                    // threshold put 23 % of the steps through it: a quarter of the kernel's instructions).
                    static_assert(ORI_NBINS == 36, "bin units of the fast angle");
                    const float bfast = at + 18.0f;
                    const float bfl = floorf(bfast);
                    int bidx = (int)bfl + ((bfast - bfl) >= 0.5f ? 1 : 0);
                    if (fabsf((bfast - bfl) - 0.5f) < 2e-4f)
                        bidx = (int)roundf((float)ORI_NBINS * (atan2_1r(gy, gx) + PI_F) / PI2_F);
                    bidx = (bidx == ORI_NBINS) ? 0 : bidx;
                    // weight <= |gradient| <= 255 sqrt 2 for pixel values in 0..255 (float images: 0..1 scaled by 255 at level
                    // 0), so weight * 2^23 < 2^32: one v_cvt_u32_f32 (truncating like the 64-bit conversion) instead of the seven
                    // instructions of float -> u64; a float image outside [0, 1] can exceed it and takes the long form
                    const float wf = weight * OFIX;
                    fix64 wq = (fix64)(unsigned)fminf(wf, 4294967040.0f);
                    if (__builtin_amdgcn_ballot_w64(wf >= 4294967296.0f) != 0ull) wq = wide_fix(wf);   // a call: not speculated into the common path
                    atomicAdd(&myhist[bidx], wq);
                }
            }
        }
        wave_fence();

        // bins live on lanes 0..35; lanes >= 36 mirror the reference's scratch bins 36..63
        const bool isbin = lane < ORI_NBINS;
        float hval = 0.0f;
        if (isbin) {
            fix64 hsum = 0ull;
#pragma unroll
            for (int c = 0; c < HCOPIES; c++) hsum += hist[c * ORI_NBINS + lane];
            hval = __ull2float_rn(hsum) * (1.0f / OFIX);
        }
        const int prev_l = isbin ? (lane == 0 ? ORI_NBINS - 1 : lane - 1) : lane;
        const int next_l = isbin ? (lane == ORI_NBINS - 1 ? 0 : lane + 1) : lane;
        // ds_bpermute with the two byte addresses formed once (__shfl recomputes them, ~4 VALU, at each of its 14 uses)
        auto from = [](int byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); };
        const int prev_a = prev_l * 4, next_a = next_l * 4;
#pragma unroll
        for (int it = 0; it < 6; it++) {   // 3 x (hist->sm_hist->hist), s_orientation.cu:166-174
            const float pv = from(prev_a, hval);
            const float nv = from(next_a, hval);
            hval = div3(pv + hval + nv);
        }
        const float hp = from(prev_a, hval);
        const float hn = from(next_a, hval);
        // A peak is a bin above both neighbours (s_orientation.cu:199).  The right-hand test is >= here: the histogram is an
        // exact integer sum, so a gradient field that is mirror symmetric about a bin boundary (blobs, corners) gives two
        // EQUAL top bins, which the strict test would drop altogether -- the keypoint would get the orientation of some
        // minor peak.  In the reference's float accumulation (and the oracle's) rounding noise breaks such ties, one of the
        // two bins wins and the parabola puts the angle at their common boundary; the left bin of the pair with hn == hval
        // gives exactly that (newbin = 1.5).  Bins that differ are treated as before.  The tie counts only when the PAIR is a
        // local maximum (the bin behind it is lower): an equal pair on an ascending shoulder (1, 2, 2, 3) is no peak.
        const float hnn = from(next_a, hn);
        bool predicate = isbin && hval > hp && (hval > hn || (hval == hn && hn > hnn));
        const float num  = predicate ? 3.0f * hp - 4.0f * hval + 1.0f * hn : 0.0f;
        const float denB = predicate ? 2.0f * (hp - 2.0f * hval + hn) : 1.0f;
        const float newbin = num / denB;
        predicate = predicate && newbin >= 0.0f && newbin <= 2.0f;
        const float refined = predicate ? (float)prev_l + newbin : -1.0f;
        const float yval    = predicate ? -(num * num) / (4.0f * denB) + hp : -INFINITY;

        // top-4 by value (BitonicSort::Warp32::sort64 + lanes 0..3, s_orientation.cu:224-247): equal values keep
        // their lane order, missing entries are (-inf, -1).  The peaks are few (<= 18, typically 3..5): a wave-uniform
        // insertion over the set bits of the peak mask instead of four 64-lane max reductions.
        float sel_val[PSX_ORI_MAX], sel_bin[PSX_ORI_MAX];
#pragma unroll
        for (int k = 0; k < PSX_ORI_MAX; k++) { sel_val[k] = -INFINITY; sel_bin[k] = -1.0f; }
        for (unsigned long long pm = __ballot(predicate); pm != 0ull; pm &= pm - 1ull) {
            const int pl = __ffsll((long long)pm) - 1;
            float cv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yval), pl));
            float cb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(refined), pl));
#pragma unroll
            for (int k = 0; k < PSX_ORI_MAX; k++) {
                if (cv > sel_val[k]) {                   // strict: an earlier lane stays in front of an equal later one
                    const float tv = sel_val[k], tb = sel_bin[k];
                    sel_val[k] = cv; sel_bin[k] = cb;
                    cv = tv; cb = tb;
                }
            }
        }
        if (lane == 0) {
            psx_extremum ex;
            const float yval_ref = 0.8f * sel_val[0];
            int angles = 0;
#pragma unroll
            for (int k = 0; k < PSX_ORI_MAX; k++) {
                ex.orientation[k] = 0.0f;
                if (sel_val[k] >= yval_ref) {
                    float chosen_bin = sel_bin[k];
                    if (chosen_bin >= ORI_NBINS) chosen_bin -= ORI_NBINS;
                    ex.orientation[k] = fmaf(PI2_F * chosen_bin, 1.0f / ORI_NBINS, -PI_F);
                    angles++;
                }
            }
            ex.xpos = ie.xpos; ex.ypos = ie.ypos; ex.lpos = ie.lpos; ex.sigma = ie.sigma;
            ex.octave = o; ex.num_ori = angles; ex.idx_ori = 0;
            P->extrema[e] = ex;
            P->ext_nori[e] = angles;
        }
        wave_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// Exclusive scan of num_ori over all extrema (octave-major) -> idx_ori, feat_to_ext map, counters.
// SCAN_WGS workgroups of 1024 threads; workgroup b owns the b-th contiguous segment of the extrema.  It first
// sums num_ori of everything in front of its segment (a few int4 loads per thread, L2 resident), then scans its
// own segment (4 extrema per thread and pass: wave __shfl_up + 16 wave totals) and writes the results.  No
// workgroup waits for another one; the last workgroup also knows the grand total and writes the frame counters.
// (One workgroup for the whole list, round 1, was a 22 us chain of dependent passes and ~1 MB of writes from one CU.)
// Replaces the reference's 32x32-thread ExclusivePrefixSum::Block (excl_blk_prefix_sum.h:34-145).
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_NT = 1024, SCAN_WGS = 16;

__device__ __forceinline__ void write_feature(const PsxParams* P, const PsxExport& X, int i, const psx_extremum& ex, int excl, int limit)
{
    // prep_features, sift_pyramid.cu:250-280 (descriptor pointers become indices, -1 == nullptr)
    psx_feature f;
    const float s = ldexpf(1.0f, ex.octave - P->up_fac);
    f.debug_octave = ex.octave;
    f.xpos = ex.xpos * s;
    f.ypos = ex.ypos * s;
    f.sigma = ex.sigma * s;
    f.num_ori = ex.num_ori;
#pragma unroll
    for (int k = 0; k < PSX_ORI_MAX; k++) {
        const bool on = k < ex.num_ori;
        f.orientation[k] = on ? ex.orientation[k] : 0.0f;
        f.desc_idx[k] = (on && excl + k < limit) ? excl + k : -1;
    }
    P->features[i] = f;
    if (X.features != nullptr && i < X.feat_capacity) X.features[i] = f;
}

__global__ __launch_bounds__(SCAN_NT) void k_scan(const PsxParams* __restrict__ P, PsxCounters* cnt, const PsxExport X)
{
    __shared__ int s_wsum[SCAN_NT / PSX_WAVE];
    __shared__ int s_total, s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x;
    const bool last_wg = b == SCAN_WGS - 1;

    // per-octave prefix sums of the extrema counts: lane o of the first wave owns octave o
    if (wave == 0) {
        const int c = (lane < P->num_octaves) ? ext_count(P, cnt, lane) : 0;
        int v = c;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int u = __shfl_up(v, off);
            if (lane >= off) v += u;
        }
        const int ps = __shfl(v, PSX_MAX_OCTAVES - 1);
        if (b == 0) {
            if (lane < PSX_MAX_OCTAVES) cnt->ext_ps[lane] = v - c;
            if (lane == 0) cnt->ext_ps[PSX_MAX_OCTAVES] = ps;
        }
        if (lane == 0) s_total = min(ps, P->ext_capacity);
    }
    __syncthreads();
    const int total = s_total;
    // Descriptor capacity the reference would have for this frame: max(2 max_extrema, 1.25 max_extrema)
    // (sift_pyramid.cu:154-159), grown to 2 x the extrema count rounded up to 1024 when there are more than
    // max_extrema extrema (Pyramid::reallocExtrema, sift_pyramid.cu:179-209).  Orientations beyond it are
    // dropped (the reference would write past its buffer).  The buffers of this context may be smaller: then
    // the host grows them to `rule` and reruns this kernel (api.hip regrow_descriptors).
    int rule = max(2 * P->max_extrema, P->max_extrema + P->max_extrema / 4);
    if (total > P->max_extrema) rule = max(rule, 2 * ((total + 1024) & ~1023));
    const int cap = min(P->ori_capacity, rule);
    const int* nori = P->ext_nori;                 // padded to a multiple of 16 entries

    // segment of this workgroup: multiples of 16 extrema, so that int4 loads never straddle a boundary
    const int seg_len = (((total + SCAN_WGS - 1) / SCAN_WGS) + 15) & ~15;
    const int seg_start = b * seg_len;
    const int seg_end = min(seg_start + seg_len, total);
    if (seg_start >= total && !last_wg) return;

    // ---- everything in front of the segment ----
    {
        const int lim = min(seg_start, total);
        int acc = 0;
        for (int i = t * 4; i < lim; i += SCAN_NT * 4) {
            const int4 v4 = *reinterpret_cast<const int4*>(nori + i);
            acc += v4.x + ((i + 1 < lim) ? v4.y : 0) + ((i + 2 < lim) ? v4.z : 0) + ((i + 3 < lim) ? v4.w : 0);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) s_wsum[wave] = acc;
        __syncthreads();
        if (t == 0) {
            int sum = 0;
            for (int w = 0; w < SCAN_NT / PSX_WAVE; w++) sum += s_wsum[w];
            s_base = sum;
        }
        __syncthreads();
    }

    // ---- the segment: SCAN_K consecutive extrema per thread and pass ----
    constexpr int SCAN_K = 4;
    int carry = s_base;
    for (int pbase = seg_start; pbase < seg_end; pbase += SCAN_NT * SCAN_K) {
        const int i0 = pbase + t * SCAN_K;
        int4 v4 = make_int4(0, 0, 0, 0);
        if (i0 < seg_end) v4 = *reinterpret_cast<const int4*>(nori + i0);
        int nv[SCAN_K] = {v4.x, v4.y, v4.z, v4.w};
        int local = 0;
#pragma unroll
        for (int k = 0; k < SCAN_K; k++) { if (i0 + k >= seg_end) nv[k] = 0; local += nv[k]; }

        int v = local;
#pragma unroll
        for (int off = 1; off < PSX_WAVE; off <<= 1) {
            const int u = __shfl_up(v, off);
            if (lane >= off) v += u;
        }
        __syncthreads();                      // s_wsum free (previous pass / the prologue finished reading)
        if (lane == PSX_WAVE - 1) s_wsum[wave] = v;
        __syncthreads();
        if (wave == 0) {
            int ws = (lane < SCAN_NT / PSX_WAVE) ? s_wsum[lane] : 0;
#pragma unroll
            for (int off = 1; off < SCAN_NT / PSX_WAVE; off <<= 1) {
                const int u = __shfl_up(ws, off);
                if (lane >= off) ws += u;
            }
            if (lane < SCAN_NT / PSX_WAVE) s_wsum[lane] = ws;   // inclusive over waves
        }
        __syncthreads();
        int excl = carry + v - local + (wave > 0 ? s_wsum[wave - 1] : 0);
        carry += s_wsum[SCAN_NT / PSX_WAVE - 1];
#pragma unroll
        for (int k = 0; k < SCAN_K; k++) {
            const int i = i0 + k;
            if (i < seg_end) {
                const int n = nv[k];
                P->extrema[i].idx_ori = excl;
                for (int q = 0; q < n; q++)
                    if (excl + q < cap) P->feat_to_ext[excl + q] = i;
                if (excl >= cap) {                 // no descriptor wave will visit this extremum
                    psx_extremum ex = P->extrema[i];
                    ex.idx_ori = excl;
                    write_feature(P, X, i, ex, excl, cap);
                }
                excl += n;
            }
        }
    }
    if (last_wg && t == 0) {
        const int grand = carry;                   // everything in front + the last segment
        const int ori_total = min(grand, cap);
        cnt->ext_total = total;
        cnt->ori_total = ori_total;
        cnt->ori_raw = min(grand, rule);           // what the frame needs: the host grows the descriptor buffers up to it
        if (X.counts != nullptr) { X.counts[0] = total; X.counts[1] = ori_total; X.counts[2] = min(grand, rule); X.counts[3] = cnt->flow_error; }
    }
}

// normalize_histogram (s_desc_norm_rs.h:42-77 / s_desc_norm_l2.h:86-135); the lane owns bins 2*lane, 2*lane+1 of
// descriptor j and stores them (device array and, when attached, the zero-copy export)
__device__ __forceinline__ void normalize_store(const PsxParams* P, const PsxExport& X, int j, int lane, float a, float b)
{
    if (P->norm_mode == PSX_NORM_ROOTSIFT) {
        const float sum = wave_sum(a + b);
        a = ldexpf(sqrtf(a / sum), P->norm_multi);
        b = ldexpf(sqrtf(b / sum), P->norm_multi);
    } else {
        float norm = sqrtf(wave_sum(a * a + b * b));
        a = fminf(a, 0.2f * norm);
        b = fminf(b, 0.2f * norm);
        norm = wave_sum(a * a + b * b);
        norm = 1.0f / sqrtf(norm);
        norm = ldexpf(norm, P->norm_multi);
        a = a * norm;
        b = b * norm;
    }
    reinterpret_cast<float2*>(P->desc + (size_t)j * 128)[lane] = make_float2(a, b);
    if (X.desc != nullptr && j < X.desc_capacity)
        reinterpret_cast<float2*>(X.desc + (size_t)j * 128)[lane] = make_float2(a, b);
}

// ---------------------------------------------------------------------------------------------
// Descriptor ("loop" mode) + normalisation + Feature record
//
// One wave64 per descriptor.  The wave walks the bounding box of the rotated 5x5-SBP window in
// 16x4-pixel tiles (lane = (x&15, y&3)); tiles with no pixel inside the window are skipped with
// one ballot.  Every pixel inside is visited once: gradient magnitude / angle once, Gaussian
// weight once, then the classic trilinear scatter into <= 2x2 spatial tiles x 2 orientation bins
// with integer LDS atomics (fixed-point bins, below).  This is the same sum the reference forms by scanning the window from each of
// its 16 tile warps (s_desc_loop.cu:60-124): a pixel contributes to tile (ix,iy) with weight
// (1-|u-ix|)(1-|v-iy|) iff |u-ix|<1 and |v-iy|<1, where (u,v) are the pixel's coordinates in
// tile units; only rounding differs (<= 1e-6 relative per sample; the test tolerance on the
// normalised descriptor is 1e-3).  The reference's fast intrinsics (__expf, __sincosf,
// __fdividef) are matched with gfx950 fast paths here: v_exp_f32, v_rcp_f32, v_sqrt_f32 and a
// degree-13 odd minimax polynomial for atan (max error 3.3e-7 rad).
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// k_descriptors.  Round 1's kernel (one pixel per lane, eight ds_add_u32 per pixel) was measured with PMC and with
// its atomics / its loads compiled out (profiles/r02_descriptor_experiments.txt): 85 M wave-level VALU instructions
// and an LDS atomic pipe with 70 % conflict cycles and 30 % of all wave cycles stalled on LDS issue; removing the
// atomics alone or the loads alone did not shorten it.  Both instruction streams are cut here:
//   * a lane owns the pixel PAIR (2c, 2c+1): gfx950 runs v_pk_{add,mul,fma}_f32 on two floats per lane at the
//     single-float rate, the pair shares its row terms and its gradient loads (one 16-byte load per row, rows are
//     256-byte aligned and 2c is even); a step covers 16 x 8 pixels (lane = (c 0..7, row 0..7));
//   * the two orientation bins of a contribution (fo, fo+1) go out as ONE ds_add_u64: bins are 32-bit fixed point
//     with head-room (below), so the low word never carries into the high word.  A tile keeps its 8
//     bins twice: as pairs (0,1)(2,3)(4,5)(6,7) for even fo and (1,2)(3,4)(5,6)(7,0) for odd fo; the epilogue adds
//     the two views;
//   * the angle lives in bin units (0..8 = 0..2 pi) from the atan polynomial on and is not wrapped into [0, 8):
//     floor() and "& 7" wrap the bin index, the fractional part is the same.
// ---------------------------------------------------------------------------------------------

// DENORM (round 4): the fixed-point scale is folded into the Gaussian weight as 2^14 * 2^-149, so that every final
// contribution (bin weight x tile weight x magnitude) is a DENORMAL float -- and the bit pattern of the denormal k * 2^-149 is
// the integer k: the multiply itself rounds to the 18.14 grid (round to nearest even), the pair (this bin, next bin) comes
// out of one v_pk_mul_f32 as the 64-bit word ds_add_u64 adds.  16 v_cvt_u32_f32 per pixel pair and the + 0.5 go away
// (134 -> ~118 VALU per step).  f32 denormals are not flushed on gfx950 (HIP default) and run at the normal rate; the
// intermediate weights are denormal too, i.e. rounded to the same grid before the last product: <= 2 units of 2^-14 per
// contribution instead of 1/2, against descriptor sums of 10^2..10^3.
// byte offset of the u64 that holds bins (fo, fo + 1) in a tile's two views: even fo -> word fo, odd fo -> word 8 + fo - 1
__device__ __forceinline__ unsigned bin_slot(unsigned fo)
{
    const unsigned h = fo & 7u;
    return ((((h << 3) | h) >> 1) & 7u) << 3;            // v_and, v_lshl_or, v_bfe; the shift rides on the add
}

// WB = waves per workgroup.  A workgroup's LDS and wave slots are released when its LAST wave ends; descriptors cost 1 : 4 by their
// sigma, so in a 4-wave workgroup three waves wait for the one that drew the largest window (WGPC counts 4-wave workgroups per CU
// whatever WB is: the resident waves are the same).
template <bool DENORM, int WGPC, int WB>
__global__ __launch_bounds__(PSX_WAVE * WB, WGPC * 4 / WB) void k_descriptors(const PsxParams* __restrict__ P, const PsxCounters* cnt, const PsxExport X)
{
    // Histogram layout per copy: tiles (iy, ix), iy, ix in -1..4, at index (iy+1)*5 + (ix+1) (31 slots), 16 words
    // each (two views of 8 bins).  Column 0 and rows 0 / 5 are never read: the trilinear scatter of a pixel near
    // the window border lands there instead of being range-checked; ix = 4 of row iy aliases ix = -1 of row iy+1,
    // both are dump slots.  4 private copies per wave keyed by the pixel pair's column and row parity:
    // neighbouring pixels fall into the same tile and orientation bin, and same-address atomics serialise.
    // Bins are 18.14 unsigned fixed point (round to nearest per contribution): a bin collects at most (2*SBP)^2
    // pixels x |gradient| <= 360.7.  SBP = 3*sigma, and sigma <= 2 * 2^((levels+1.5)/levels) <= 6.73 for any legal
    // Config (sigma0 <= 2, gauss_filter.cu:131; levels >= 2, popsift.cpp:86) gives SBP <= 20.2, < 5.9e5 in total
    // and < 1.5e5 per copy (each copy takes a quarter of the pixels) against 2^18 = 2.6e5
    // (tests/test_gpu_configs.py::test_descriptor_bins_do_not_overflow_at_large_sigma).
    // Round 6: the copies OVERLAP.  Slots 25..30 of a copy (row iy = 4, written, never read) lie on slots 0..5 of the next copy
    // (row iy = -1 and tile (0, -1): written, never read), and the last copy of a wave on the first of the next wave (the last wave
    // has a tail of its own): a copy advances by 25 slots + 2 words (the bank offset between copies that 31 slots + 2 had) instead of
    // 31 slots + 2.  31.9 KB -> 26.1 KB per workgroup: six workgroups per CU fit the 160 KB instead of five.
    constexpr int DCOPIES = 4, DTILES = 31, DSTRIDE = 25 * 16 + 2, WSTRIDE = DCOPIES * DSTRIDE;         // even: 8-byte aligned copies
    static_assert(DSTRIDE % 2 == 0 && WSTRIDE % 4 == 0, "8-byte aligned copies, 16-byte aligned waves");
    static_assert(DTILES * 16 - DSTRIDE <= 6 * 16, "a copy's tail must end inside the next copy's dump slots 0..5");
    constexpr float DFIX = 16384.0f;
    // (the WGPC = 5 instantiation pads the array to round 5's 31.9 KB: the A/B partner, POPSIFT_DESC_OCC=5)
    __shared__ __attribute__((aligned(16))) unsigned s_desc[WB * WSTRIDE + (DTILES * 16 - DSTRIDE) + 2 + (WGPC == 5 ? 1440 : 0)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned* acc = s_desc + wave * WSTRIDE;
    const int lx = lane & 7, ly = lane >> 3;
    const int copy = (lane & 1) | (((lane >> 3) & 1) << 1);                    // neighbours in x and in y use different copies
    const unsigned myacc = (unsigned)(uintptr_t)(acc + copy * DSTRIDE);

    const int total = cnt->ori_total;
    const int nwaves = gridDim.x * WB;
    for (int jv = blockIdx.x * WB + wave; jv < total; jv += nwaves) {
        const int j = __builtin_amdgcn_readfirstlane(jv);
        const int ext_idx = P->feat_to_ext[j];
        const psx_extremum ex = P->extrema[ext_idx];
        const int ori_num = psx_clampi(j - ex.idx_ori, 0, PSX_ORI_MAX - 1);
        // a select chain, not ex.orientation[ori_num]: the dynamic index sent the whole record through scratch memory
        static_assert(PSX_ORI_MAX == 4, "orientation select chain");
        const float ang = ori_num == 0 ? ex.orientation[0] : ori_num == 1 ? ex.orientation[1] : ori_num == 2 ? ex.orientation[2] : ex.orientation[3];
        const PsxOctave oc = P->oct[ex.octave];
        const int width = oc.w, height = oc.h;

        if (ori_num == 0 && lane == 0) write_feature(P, X, ext_idx, ex, ex.idx_ori, total);

        // (the wave clears ITS words; the tail it shares with the next wave holds dump slots only)
        for (int i = lane; i < WSTRIDE / 4; i += PSX_WAVE) reinterpret_cast<uint4*>(acc)[i] = make_uint4(0u, 0u, 0u, 0u);
        wave_fence();

        const float x = ex.xpos, y = ex.ypos;
        const int   level = psx_clampi(ex.lpos, 0, P->L - 1);
        const float SBP = fabsf(DESC_MAGNIFY * ex.sigma);
        const unsigned long long pb = (unsigned long long)(uintptr_t)(oc.data + (size_t)level * oc.plane);
        const unsigned pb_lo = __builtin_amdgcn_readfirstlane((unsigned)pb);
        const unsigned pb_hi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
        const char* plane = reinterpret_cast<const char*>((uintptr_t)(((unsigned long long)pb_hi << 32) | pb_lo));
        const unsigned pitch4 = __builtin_amdgcn_readfirstlane((unsigned)oc.pitch * 4u);     // the octave record came through vector loads
        const char* plane_dn = plane + pitch4;
        const char* plane_up = plane - pitch4;

        if (SBP != 0.0f) {
            // the reference takes __sincosf here (s_desc_loop.cu:38); v_sin_f32 / v_cos_f32 work in revolutions
            const float rev   = ang * 0.15915494309189535f;
            const float cos_t = __builtin_amdgcn_cosf(rev);
            const float sin_t = __builtin_amdgcn_sinf(rev);
            const float csbp  = cos_t * SBP;
            const float ssbp  = sin_t * SBP;
            const float crsbp = cos_t / SBP;
            const float srsbp = sin_t / SBP;
            const float bsz   = fabsf(csbp) + fabsf(ssbp);
            const float ang_bins = ang * M_4RPI_F;

            // bounding box of the four corners (+-1.5, +-1.5) widened by bsz (s_desc_loop.cu:46-58): floor(p - bsz) and
            // floor(p + bsz) are monotone in p, so the minimum / maximum over the corners is taken BEFORE the floor -- the
            // same integers as four floors each, a third of the instructions
            const float ix0 = fmaf(-ssbp, -1.5f, x), ix1 = fmaf(-ssbp, 1.5f, x);     // corner x = csbp*ox - ssbp*oy + x
            const float iy0 = fmaf( ssbp, -1.5f, y), iy1 = fmaf( ssbp, 1.5f, y);     // corner y = csbp*oy + ssbp*ox + y
            const float px0 = fmaf(csbp, -1.5f, ix0), px1 = fmaf(csbp, 1.5f, ix0), px2 = fmaf(csbp, -1.5f, ix1), px3 = fmaf(csbp, 1.5f, ix1);
            const float py0 = fmaf(csbp, -1.5f, iy0), py1 = fmaf(csbp, -1.5f, iy1), py2 = fmaf(csbp, 1.5f, iy0), py3 = fmaf(csbp, 1.5f, iy1);
            int xmin = (int)floorf(fminf(fminf(px0, px1), fminf(px2, px3)) - bsz);
            int xmax = (int)floorf(fmaxf(fmaxf(px0, px1), fmaxf(px2, px3)) + bsz);
            int ymin = (int)floorf(fminf(fminf(py0, py1), fminf(py2, py3)) - bsz);
            int ymax = (int)floorf(fmaxf(fmaxf(py0, py1), fmaxf(py2, py3)) + bsz);
            xmin = max(1, xmin); ymin = max(1, ymin);
            xmax = min(width - 2, xmax); ymax = min(height - 2, ymax);
            // Row spans.  The window is the rotated square -1 < u, v < 4; in a row (dy fixed) that is an interval of
            // dx from each of the two constraints.  A step covers 8 rows x 8 pixel pairs and every row starts at ITS
            // span (the bounding box of the rotated square is up to twice its area; walking the box in common
            // columns left about half of the lanes outside the window).  The spans are conservative by a pixel,
            // the exact predicate below decides (round 4: the slack was 2-3 pixels per side, 4.6 % more steps).
            const bool use_c = fabsf(crsbp) > 1e-6f, use_s = fabsf(srsbp) > 1e-6f;
            // v_rcp_f32 (1 ulp) instead of two IEEE divisions: these only place the conservative spans
            const float rcc = use_c ? __builtin_amdgcn_rcpf(crsbp) : 0.0f, rcs = use_s ? __builtin_amdgcn_rcpf(srsbp) : 0.0f;
            const float fxmin = (float)xmin - x, fxmax = (float)xmax - x;

            // The spans of 64 rows at a time, one row per lane (the same expressions as before, evaluated once per row instead
            // of once per lane and 8-row step: ~35 VALU per step-group became one ds_bpermute and ~10); a step-group fetches
            // its rows' entries from the lanes that hold them.  Entry = first column | end column << 16, relative to the box.
            const int xmin0 = xmin & ~1;
            // one pixel pair (jj, jj + 1) of row `rowoff`: ok0 / ok1 = the pixel lies in the row's span; accb = the lane's histogram copy
            auto pixel_pair = [&](int jj, float fj, float ub, float vb, unsigned rowoff, bool ok0, bool ok1, unsigned accb) __attribute__((always_inline)) {
                    const float dx0 = fj - x;
                    const v2f dxk = (v2f){dx0, dx0 + 1.0f};
                    const v2f u = pk_fma(splat(crsbp), dxk, splat(ub));
                    const v2f v = pk_fma(splat(-srsbp), dxk, splat(vb));
                    // the window -1 < u, v < 4 as |u - 1.5|, |v - 1.5| < 2.5: four compares with free abs modifiers instead of
                    // eight (a pixel within an ulp of the border has a weight of that order: it does not matter which side it falls)
                    const v2f un = u - splat(1.5f), vn = v - splat(1.5f);
                    const bool in0 = ok0 && (fabsf(un.x) < 2.5f) && (fabsf(vn.x) < 2.5f);     // xbe <= xmax
                    const bool in1 = ok1 && (fabsf(un.y) < 2.5f) && (fabsf(vn.y) < 2.5f);
                    // (no separate "any lane?" test: the branch around the block below is taken when exec comes out empty)
                    // (a software pipeline that issues the next step's loads before this step's arithmetic was
                    // measured: the compiler's conservative s_waitcnt across the loop edge undoes it, +12 %)
                    if (in0 || in1) {
                        const unsigned off = rowoff + (unsigned)jj * 4u;
                        const v2f  ctr = *(gv2f_p)(plane + off);                  // p[jj], p[jj+1]
                        const float lft = *(gfloat_p)(plane + off - 4u);          // p[jj-1]
                        const float rgt = *(gfloat_p)(plane + off + 8u);          // p[jj+2]
                        // the rows above and below through their own (uniform) bases: one lane offset serves all three loads
                        const v2f  dwn = *(gv2f_p)(plane_dn + off);
                        const v2f  upp = *(gv2f_p)(plane_up + off);
#ifdef PSX_MODEL_NOGRAD
                        // measurement build only: magnitude and angle of the pixel pair from one 8-byte read (see k_orientation)
                        const v2f gdx = ctr, gdy = (v2f){ctr.y, ctr.x};
                        (void)lft; (void)rgt; (void)dwn; (void)upp;
                        const v2f mod = ctr;
#else
                        const v2f gdx = (v2f){ctr.y - lft, rgt - ctr.x};
                        const v2f gdy = dwn - upp;
                        const v2f m2 = pk_fma(gdx, gdx, gdy * gdy);
                        const v2f mod = (v2f){__builtin_amdgcn_sqrtf(m2.x), __builtin_amdgcn_sqrtf(m2.y)};
#endif
                        // atan2 in bin units: (4/pi) atan(min/max) by the degree-13 odd polynomial, then the octant
                        const v2f ax = (v2f){fabsf(gdx.x), fabsf(gdx.y)}, ay = (v2f){fabsf(gdy.x), fabsf(gdy.y)};
#ifdef PSX_MODEL_NOGRAD
                        v2f r = gdy * splat(0.01f);
#else
                        const v2f mx = (v2f){fmaxf(ax.x, ay.x), fmaxf(ax.y, ay.y)};
                        const v2f mn = (v2f){fminf(ax.x, ay.x), fminf(ax.y, ay.y)};
                        const v2f rc = (v2f){__builtin_amdgcn_rcpf(fmaxf(mx.x, 1e-30f)), __builtin_amdgcn_rcpf(fmaxf(mx.y, 1e-30f))};
                        const v2f a = mn * rc;
                        const v2f s2 = a * a;
                        v2f r = splat(0.006811792496591806f * M_4RPI_F);
                        r = pk_fma(r, s2, splat(-0.0336042195558548f * M_4RPI_F));
                        r = pk_fma(r, s2, splat(0.07962366938591003f * M_4RPI_F));
                        r = pk_fma(r, s2, splat(-0.1323334127664566f * M_4RPI_F));
                        r = pk_fma(r, s2, splat(0.19807815551757812f * M_4RPI_F));
                        r = pk_fma(r, s2, splat(-0.3331736922264099f * M_4RPI_F));
                        r = pk_fma(r, s2, splat(0.9999961256980896f * M_4RPI_F));
                        r = r * a;
#endif
                        float r0 = r.x, r1 = r.y;
                        r0 = (ay.x > ax.x) ? 2.0f - r0 : r0;   r1 = (ay.y > ax.y) ? 2.0f - r1 : r1;
                        r0 = (gdx.x < 0.0f) ? 4.0f - r0 : r0;  r1 = (gdx.y < 0.0f) ? 4.0f - r1 : r1;
                        // r >= 0 here: "negative for gdy < 0" is the sign bit of gdy (one v_bfi); gdy == -0.0 turns angle 0 / 4 into
                        // -0 / -4, the same bin and fraction after the wrap below
                        r0 = __builtin_copysignf(r0, gdy.x);   r1 = __builtin_copysignf(r1, gdy.y);
                        const v2f tth = (v2f){r0, r1} - splat(ang_bins);          // in (-12, 12): not wrapped
                        const v2f ffo = (v2f){floorf(tth.x), floorf(tth.y)};
                        const v2f wgt2 = tth - ffo;
                        const v2f wgt1 = splat(1.0f) - wgt2;

                        const v2f d2 = pk_fma(un, un, vn * vn) * splat(-0.125f * 1.4426950408889634f);
                        const v2f ww = (v2f){__builtin_amdgcn_exp2f(d2.x), __builtin_amdgcn_exp2f(d2.y)} * (mod * splat(DENORM ? 0x1p-135f : DFIX));
                        const v2f fu = (v2f){floorf(u.x), floorf(u.y)}, fv = (v2f){floorf(v.x), floorf(v.y)};
                        const v2f ax1 = u - fu, ay1 = v - fv;
                        const v2f ax0 = splat(1.0f) - ax1, ay0 = splat(1.0f) - ay1;
                        const v2f wy0 = ay0 * ww, wy1 = ay1 * ww;
                        const v2f w00 = wy0 * ax0, w01 = wy0 * ax1, w10 = wy1 * ax0, w11 = wy1 * ax1;
                        if constexpr (DENORM) {
                            // per pixel the pair (bin fo, bin fo + 1) = (wgt1, wgt2) x tile weight: the product's BITS are the word
                            auto bits = [](v2f p) { fix64 b; __builtin_memcpy(&b, &p, 8); return b; };
                            // tile index (iy0 + 1) * 5 + (ix0 + 1) in float (small exact integers), one conversion per pixel
                            const v2f tidx = pk_fma(fv, splat(5.0f), fu + splat(6.0f));
                            if (in0) {
                                const v2f pw = pk_fma(splat(wgt2.x), (v2f){-1.0f, 1.0f}, (v2f){1.0f, 0.0f});     // (1 - w, w) = (wgt1.x, wgt2.x) without moving components
                                // tile (iy0, ix0) at 64-byte granules; pair view: even fo -> word fo, odd fo -> word 8 + fo - 1, i.e. the
                                // u64 slot (b0 b2 b1) for fo = (b2 b1 b0): the three bits rotated right by one
                                const unsigned tb = bin_slot((unsigned)(int)ffo.x) + (((unsigned)(int)tidx.x << 6) + accb);
                                fix64 LDS_AS* t = (fix64 LDS_AS*)tb;
                                lds_add(t, bits(pw * splat(w00.x)));      lds_add(t + 8, bits(pw * splat(w01.x)));       // +1 tile = 16 words = 8 u64
                                lds_add(t + 40, bits(pw * splat(w10.x))); lds_add(t + 48, bits(pw * splat(w11.x)));      // +5 / +6 tiles
                            }
                            if (in1) {
                                const v2f pw = pk_fma(splat(wgt2.y), (v2f){-1.0f, 1.0f}, (v2f){1.0f, 0.0f});
                                const unsigned tb = bin_slot((unsigned)(int)ffo.y) + (((unsigned)(int)tidx.y << 6) + accb);
                                fix64 LDS_AS* t = (fix64 LDS_AS*)tb;
                                lds_add(t, bits(pw * splat(w00.y)));      lds_add(t + 8, bits(pw * splat(w01.y)));
                                lds_add(t + 40, bits(pw * splat(w10.y))); lds_add(t + 48, bits(pw * splat(w11.y)));
                            }
                        } else {
                        const v2f h = splat(0.5f);
                        const v2f a00 = pk_fma(wgt1, w00, h), b00 = pk_fma(wgt2, w00, h);
                        const v2f a01 = pk_fma(wgt1, w01, h), b01 = pk_fma(wgt2, w01, h);
                        const v2f a10 = pk_fma(wgt1, w10, h), b10 = pk_fma(wgt2, w10, h);
                        const v2f a11 = pk_fma(wgt1, w11, h), b11 = pk_fma(wgt2, w11, h);
                        auto pack = [](float lo, float hi) { return (fix64)(unsigned)lo | ((fix64)(unsigned)hi << 32); };
                        if (in0) {
                            const unsigned fo = (unsigned)((int)ffo.x & 7);
                            // tile (iy0, ix0) at 64-byte granules; pair view: even fo -> word fo, odd fo -> word 8 + fo - 1
                            const unsigned tb = accb + (unsigned)(((int)fv.x + 1) * 5 + ((int)fu.x + 1)) * 64u + (fo + 7u * (fo & 1u)) * 4u;
                            fix64 LDS_AS* t = (fix64 LDS_AS*)tb;
                            lds_add(t, pack(a00.x, b00.x));      lds_add(t + 8, pack(a01.x, b01.x));       // +1 tile = 16 words = 8 u64
                            lds_add(t + 40, pack(a10.x, b10.x)); lds_add(t + 48, pack(a11.x, b11.x));      // +5 / +6 tiles
                        }
                        if (in1) {
                            const unsigned fo = (unsigned)((int)ffo.y & 7);
                            const unsigned tb = accb + (unsigned)(((int)fv.y + 1) * 5 + ((int)fu.y + 1)) * 64u + (fo + 7u * (fo & 1u)) * 4u;
                            fix64 LDS_AS* t = (fix64 LDS_AS*)tb;
                            lds_add(t, pack(a00.y, b00.y));      lds_add(t + 8, pack(a01.y, b01.y));
                            lds_add(t + 40, pack(a10.y, b10.y)); lds_add(t + 48, pack(a11.y, b11.y));
                        }
                        }
                    }
            };
            // (Round 6 also built the walk as ONE flat list of 4-pair groups per 64-row block, 16 groups per step, each slot's row found
            // by a binary search over the rows' prefix sums: bit-identical descriptors, 16 % fewer executed step bodies -- and the
            // same 41.7 M VALU instructions per frame, because scan + search + per-step row terms cost what the saved bodies did; 4 %
            // slower.  Removed again: profiles/r06_desc_flat_walk.txt.)
            for (int cy = ymin; cy <= ymax; cy += PSX_WAVE) {
                int sp;
                {
                    const int ii = cy + lane;
                    const float dyk = ii - y;
                    const float ub = fmaf(srsbp, dyk, 1.5f);      // u = crsbp*dx + srsbp*dy + 1.5
                    const float vb = fmaf(crsbp, dyk, 1.5f);      // v = crsbp*dy - srsbp*dx + 1.5
                    // -1 < u < 4  <=>  crsbp*dx in (-1 - ub, 4 - ub);   -1 < v < 4  <=>  srsbp*dx in (vb - 4, vb + 1)
                    float lo = fxmin, hi = fxmax;
                    if (use_c) { const float t1 = (-1.0f - ub) * rcc, t2 = (4.0f - ub) * rcc; lo = fmaxf(lo, fminf(t1, t2)); hi = fminf(hi, fmaxf(t1, t2)); }
                    if (use_s) { const float t1 = (vb - 4.0f) * rcs, t2 = (vb + 1.0f) * rcs; lo = fmaxf(lo, fminf(t1, t2)); hi = fminf(hi, fmaxf(t1, t2)); }
                    // first / last pixel with lo < dx < hi is floor(x + lo) + 1 / ceil(x + hi) - 1: one pixel of slack on each side
                    // covers the rounding of lo / hi (~1e-5 pixel) many times over
                    const int xa = max(xmin, (int)floorf(x + lo)) & ~1;               // even: aligned pixel pairs
                    const int xb = min(xmax, (int)floorf(x + hi) + 1);
                    sp = (ii <= ymax && lo <= hi && xa <= xb) ? ((xa - xmin0) | ((xb - xmin0) << 16)) : 1;      // 1: first column 1 > end 0
                }
            const int cy_end = min(cy + PSX_WAVE - 1, ymax);
            for (int ty = cy; ty <= cy_end; ty += 8) {
                const int ii = ty + ly;
                const int ent = __builtin_amdgcn_ds_bpermute((ty - cy) * 4 + ly * 4, sp);
                const int xa = xmin0 + (ent & 0xffff), xb = xmin0 + (int)((unsigned)ent >> 16);
                // an empty row ends before every column: "jj <= xbe" is then the whole row test (one v_cmp feeds the loop
                // condition and, as a mask, the window test; all window pixels of a row lie in [xa, xb])
                const int xbe = xa <= xb ? xb : -0x7fffffff;
                const float dyk = ii - y;
                const float ub = fmaf(srsbp, dyk, 1.5f);
                const float vb = fmaf(crsbp, dyk, 1.5f);
                const unsigned rowoff = (unsigned)ii * pitch4;
                float fj = (float)(xa + 2 * lx);          // the column as a float beside the integer: + 16 is exact, no conversion per step
                for (int jj = xa + 2 * lx; __builtin_amdgcn_sicmp(jj, xbe, 41 /* ICMP_SLE */) != 0ull; jj += 16, fj += 16.0f)
                    pixel_pair(jj, fj, ub, vb, rowoff, (jj <= xbe) && (jj >= xmin), jj < xbe, myacc);
            }
            }
        }
        wave_fence();

        // the lane owns bins b0 = 2 (lane & 3) and b0 + 1 of tile lane >> 2 (iy = lane >> 4, ix = (lane >> 2) & 3):
        //   bin b0   = evenview[b0] + oddview[(b0 - 1) & 7 pair].hi,   bin b0+1 = evenview[b0 + 1] + oddview[b0 pair].lo
        unsigned sa = 0u, sb = 0u;
        const int p2 = (lane & 3) * 2;
        const int tbase = (((lane >> 4) + 1) * 5 + ((lane >> 2) & 3) + 1) * 16;
#pragma unroll
        for (int c = 0; c < DCOPIES; c++) {
            const unsigned* tp = acc + c * DSTRIDE + tbase;
            sa += tp[p2] + tp[8 + ((p2 + 6) & 7) + 1];
            sb += tp[p2 + 1] + tp[8 + p2];
        }
        normalize_store(P, X, j, lane, (float)sa * (1.0f / DFIX), (float)sb * (1.0f / DFIX));
        wave_fence();
    }
}

// ConstInfo::desc_gauss[40][40] / desc_tile[16] (sift_constants.cu:34-47), evaluated on demand with the same
// float operations instead of a __constant__ table
__device__ __forceinline__ float desc_gauss_entry(int yy, int xx)
{
    const float dn_step = 1.0f / 8.0f;
    const float dn_base = 0.5f * dn_step - 20.0f * dn_step;
    const float dnx = dn_base + xx * dn_step;
    const float dny = dn_base + yy * dn_step;
    return __builtin_amdgcn_exp2f((dnx * dnx + dny * dny) * (-0.125f * 1.4426950408889634f));      // v_exp_f32, as the default kernel's window weight
}
__device__ __forceinline__ float desc_tile_entry(int i)
{
    const float nx = -1.0f + 1.0f / 16.0f + i * 1.0f / 8.0f;
    return 1.0f - fabsf(nx);
}

// ---------------------------------------------------------------------------------------------
// Alternative descriptor modes: iloop, grid, igrid, notile (s_desc_iloop.cu, s_desc_grid.cu, s_desc_igrid.cu,
// s_desc_notile.cu).  Each samples the window differently and therefore yields a DIFFERENT descriptor than
// "loop"; they exist for API completeness of Config::setDescMode and follow the CPU restatement
// (oracle/sift_oracle.c descriptor_iloop / _grid / _igrid / _notile) sample by sample.  One workgroup per
// descriptor; the reference's 32- / 16- / 8-lane groups (one tile each) sit side by side in a wave: 2, 4 or
// 8 tiles per pass, the passes spread over the four waves.  Every lane accumulates into its own 8 (+1 wrap)
// bins -- a private LDS column, no atomics -- and the groups are then reduced with the reference's shuffle
// trees.  Gradients come from a software model of the linear-filtered layered texture (1.8 fixed-point
// weights), as in pyramid_alt.hip, read from a window of the plane staged in LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_lerp(float p, float q, float a) { return fmaf(a, q, (1.0f - a) * p); }
constexpr int ALT_BINS = 9;
// one contribution into the lane's private bins (column `lane` of a [9][64] block)
__device__ __forceinline__ void alt_add(float* bins, int lane, int b, float v) { bins[b * PSX_WAVE + lane] += v; }

// The plane as these modes read it.  AltPlane: the clamped plane in HBM.  AltWindow: the part of it the workgroup staged in
// LDS, addressed with UNCLAMPED coordinates -- entry (cx, cy) of the window holds plane[clamp(cy)][clamp(cx)], so the texel
// pairs (i, i + 1) / (j, j + 1) of a bilinear fetch are neighbours in the window whatever the clamp did to them.
struct AltPlane {
    const float* p; int W, H, pitch;
    __device__ __forceinline__ float at(int x, int y) const { return p[(size_t)psx_clampi(y, 0, H - 1) * pitch + psx_clampi(x, 0, W - 1)]; }
    // the 2 x 2 texels at (i, j) = (fx, fy), integers held as floats: top = row j, columns (i, i + 1); bot = row j + 1
    __device__ __forceinline__ void quad(float fx, float fy, v2f& top, v2f& bot) const
    {
        const int i = (int)fx, j = (int)fy;
        const int i0 = psx_clampi(i, 0, W - 1), i1 = psx_clampi(i + 1, 0, W - 1);
        const int j0 = psx_clampi(j, 0, H - 1), j1 = psx_clampi(j + 1, 0, H - 1);
        top = (v2f){p[(size_t)j0 * pitch + i0], p[(size_t)j0 * pitch + i1]};
        bot = (v2f){p[(size_t)j1 * pitch + i0], p[(size_t)j1 * pitch + i1]};
    }
};
// Rows of the window are ALT_WIN_MAX texels apart whatever its size: the texel below is an immediate offset of the same
// ds_read2_b32.  The byte address of texel (i, j) is formed in FLOAT, 4 i + 336 j + c with c = the window's LDS address
// - 4 (bx0 + 84 by0): all terms are integers below 2^24 (the kernel checks the plane's height), so two fma and one conversion
// replace two conversions, two subtractions, a quarter-rate integer multiply, two shifts and an add.
constexpr int ALT_WIN_MAX = 84;
constexpr int ALT_WIN_CAP = ALT_WIN_MAX * ALT_WIN_MAX;
struct AltWindow {
    const float LDS_AS* w; int bx0, by0; float c4;
    __device__ __forceinline__ float at(int x, int y) const { return w[(y - by0) * ALT_WIN_MAX + (x - bx0)]; }
    __device__ __forceinline__ void quad(float fx, float fy, v2f& top, v2f& bot) const
    {
        const unsigned addr = (unsigned)(int)fmaf(fy, 4.0f * ALT_WIN_MAX, fmaf(fx, 4.0f, c4));
        const float LDS_AS* q = (const float LDS_AS*)addr;
        top = (v2f){q[0], q[1]}; bot = (v2f){q[ALT_WIN_MAX], q[ALT_WIN_MAX + 1]};        // two ds_read2_b32
    }
};
// The linear-filtered layered texture at p = (x, y): texel centres at integer + 0.5, 1.8 fixed-point weights (the software
// model of pyramid_alt.hip), the two axes side by side in packed f32 instructions.  In two halves, so that a caller can put
// the texel reads of several fetches in flight before the first interpolation.
struct AltTap { v2f f, ab; };
__device__ __forceinline__ AltTap alt_tap(v2f p)
{
    const v2f ps = p + splat(0.5f);
    const v2f pb = ps - splat(0.5f);
    AltTap t;
    t.f = (v2f){floorf(pb.x), floorf(pb.y)};
    const v2f fr = (pb - t.f) * splat(256.0f);
    t.ab = (v2f){rintf(fr.x), rintf(fr.y)} * splat(1.0f / 256.0f);
    return t;
}
// d_lerp(p, q, w) = fma(w, q, (1 - w) * p): between the two rows first, both columns at once -- top and bot are
// the register pairs the two ds_read2_b32 deliver, so the packed operands need no moves -- then between
// the columns.  (The oracle's model blends the columns first; the two orders differ by an ulp of the texel difference.)
__device__ __forceinline__ float alt_blend(const AltTap& t, v2f top, v2f bot)
{
    const float nb = 1.0f - t.ab.y;
    const v2f r = pk_fma(splat(t.ab.y), bot, splat(nb) * top);
    return fmaf(t.ab.x, r.y, (1.0f - t.ab.x) * r.x);
}
// get_gradiant with the rotated stencil on the linear texture (s_gradiant.h:72-88).  Magnitude and angle only SCALE /
// interpolate a contribution (bin weights are continuous in the angle): v_sqrt_f32 and the degree-13 atan polynomial of
// the default kernel instead of hypotf and a double-precision atan2.
template <class V>
__device__ __forceinline__ void alt_gradiant_rot(const V& v, float& grad, float& theta, float x, float y, float cos_t, float sin_t)
{
    const v2f p = (v2f){x, y}, cs = (v2f){cos_t, sin_t}, sc = (v2f){-sin_t, cos_t};
    // (x + cos, y + sin), (x - cos, y - sin), (x - sin, y + cos), (x + sin, y - cos)
    const AltTap t0 = alt_tap(p + cs), t1 = alt_tap(p - cs), t2 = alt_tap(p + sc), t3 = alt_tap(p - sc);
    v2f a0, b0, a1, b1, a2, b2, a3, b3;
    v.quad(t0.f.x, t0.f.y, a0, b0);
    v.quad(t1.f.x, t1.f.y, a1, b1);
    v.quad(t2.f.x, t2.f.y, a2, b2);
    v.quad(t3.f.x, t3.f.y, a3, b3);
    // all sixteen texels before the first use (otherwise each fetch is waited for on its own: four LDS round trips per sample)
    asm volatile("" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));
    const float dx = alt_blend(t0, a0, b0) - alt_blend(t1, a1, b1);
    const float dy = alt_blend(t2, a2, b2) - alt_blend(t3, a3, b3);
    grad = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy));
    theta = fast_atan2(dy, dx);
}
// get_gradiant on the point texture at integer coordinates (s_gradiant.h:56-69)
template <class V>
__device__ __forceinline__ void alt_gradiant_pt(const V& v, float& grad, float& theta, int x, int y)
{
    const float dx = v.at(x + 1, y) - v.at(x - 1, y);
    const float dy = v.at(x, y + 1) - v.at(x, y - 1);
    grad = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy));
    theta = fast_atan2(dy, dx);
}

// Window staged per descriptor: every coordinate these modes read lies within E = 2.5 sqrt(2) SBP + 2.51 of the keypoint
// (samples: the rotated square |step| <= 2.5 in units of SBP; + 1 for the gradient stencil; `grid` snaps a sample to a pixel,
// + 0.5, through an (int) conversion, + 1; the bilinear fetch takes floor and floor + 1), i.e. integer coordinates
// floor(x) - ceil(E) .. floor(x) + ceil(E) + 1.  84 x 84 texels hold SBP <= 10.8 = sigma <= 3.6, the largest a keypoint of a
// three-level octave gets (sigma0 2^(3.5/3)); larger windows (more levels, larger sigma0) are read from the plane in HBM.

// The tiles of one wave.  The reference's 32- / 16- / 8-lane groups (one tile each) sit side by side in the wave: 2, 4 or 8
// tiles per pass; the four waves of the workgroup split the passes (iloop: two each; grid / igrid: one each; notile: its two
// passes x the two halves of the per-lane sample loop, the halves summed in the epilogue).
template <int MODE, class V>
__device__ __forceinline__ void alt_tiles(const V& v, float* bins, float* out, int lane, int wave,
                                          float x, float y, float ang, float SBP, float cos_t, float sin_t)
{
    const float csbp = cos_t * SBP, ssbp = sin_t * SBP;
    if (MODE == PSX_DESC_ILOOP) {
        // 32 lanes per tile, lane = j of the 32 x 32 sample grid, 2 tiles per pass
        const int sub = lane & 31, half = lane >> 5;
        for (int pass = wave * 2; pass < wave * 2 + 2; pass++) {
            const int tz = pass * 2 + half;
            const int ix = tz & 3, iy = tz >> 2;
            const float offx = ix - 1.5f, offy = iy - 1.5f;
            const float ptx = fmaf(csbp, offx, -ssbp * offy);
            const float pty = fmaf(csbp, offy,  ssbp * offx);
            const float bsz = fabsf(cos_t) + fabsf(sin_t);
#pragma unroll
            for (int b = 0; b < ALT_BINS; b++) bins[b * PSX_WAVE + lane] = 0.0f;
            for (int i = 0; i < 32; i++) {
                const float dx = (-bsz + sub * bsz / 16.0f);
                const float dy = (-bsz + i * bsz / 16.0f);
                const float nx = fmaf(cos_t, dx,  sin_t * dy);
                const float ny = fmaf(cos_t, dy, -sin_t * dx);
                const float nnx = fabsf(nx), nny = fabsf(ny);
                if (nnx < 1.0f && nny < 1.0f) {
                    const float jj = x + ptx + dx * SBP;
                    const float ii = y + pty + dy * SBP;
                    float mod, th;
                    alt_gradiant_rot(v, mod, th, jj, ii, cos_t, sin_t);
                    const float dnx = nx + offx, dny = ny + offy;
                    const float ww = __builtin_amdgcn_exp2f((dnx * dnx + dny * dny) * (-0.125f * 1.4426950408889634f));
                    const float wgt = ww * (1.0f - nnx) * (1.0f - nny) * mod;
                    th += (th <  0.0f  ? PI2_F : 0.0f);
                    th -= (th >= PI2_F ? PI2_F : 0.0f);
                    const float tth = th * M_4RPI_F;
                    const int   fo0 = (int)floorf(tth);
                    const float do0 = tth - fo0;
                    const int   fo  = fo0 & 7;                  // th in [0, 2 pi): fo0 in 0..8
                    alt_add(bins, lane, fo, (1.0f - do0) * wgt);
                    alt_add(bins, lane, fo + 1, do0 * wgt);
                }
            }
            bins[lane] += bins[8 * PSX_WAVE + lane];                 // dpt[0] += dpt[8]
#pragma unroll
            for (int b = 0; b < 8; b++) {
                float s = bins[b * PSX_WAVE + lane];
#pragma unroll
                for (int d = 16; d >= 1; d >>= 1) s += __shfl_down(s, d, 32);
                if (sub == 0) out[tz * 8 + b] = s;
            }
        }
    } else if (MODE == PSX_DESC_GRID || MODE == PSX_DESC_IGRID) {
        // 16 lanes per tile (lane = xd), 4 tiles per pass, 16 samples (yd) per lane
        const int xd = lane & 15, q = lane >> 4;
        const int tz = wave * 4 + q;
        const int ix = tz & 3, iy = tz >> 2;
        const float offx = ix - 1.5f, offy = iy - 1.5f;
#pragma unroll
        for (int b = 0; b < ALT_BINS; b++) bins[b * PSX_WAVE + lane] = 0.0f;
        if (MODE == PSX_DESC_GRID) {
            const float ptx = fmaf(csbp, offx, fmaf(-ssbp, offy, x));
            const float pty = fmaf(csbp, offy, fmaf( ssbp, offx, y));
            const float ldx = -cos_t + sin_t, ldy = -cos_t - sin_t;
            const float rsx = cos_t / 8.0f, rsy = sin_t / 8.0f;
            const float usx = -sin_t / 8.0f, usy = cos_t / 8.0f;
            const float rsbp = __builtin_amdgcn_rcpf(SBP);
            for (int yd = 0; yd < 16; yd++) {
                float pox = fmaf(yd + 0.5f, usx, fmaf(xd + 0.5f, rsx, ldx));
                float poy = fmaf(yd + 0.5f, usy, fmaf(xd + 0.5f, rsy, ldy));
                const float pix_x = roundf(fmaf(pox, SBP, ptx)) - ptx;
                const float pix_y = roundf(fmaf(poy, SBP, pty)) - pty;
                pox = pix_x * rsbp; poy = pix_y * rsbp;          // the reference divides; these only weigh the sample (continuous)
                float mod, th;
                alt_gradiant_pt(v, mod, th, (int)(ptx + pix_x), (int)(pty + pix_y));
                const float npx = fmaf(cos_t, pox,  sin_t * poy);
                const float npy = fmaf(cos_t, poy, -sin_t * pox);
                const float dnx = npx + offx, dny = npy + offy;
                const float ww = __builtin_amdgcn_exp2f((dnx * dnx + dny * dny) * (-0.125f * 1.4426950408889634f));
                const float wx = 1.0f - fabsf(npx), wy = 1.0f - fabsf(npy);
                if (wx < 0.0f || wy < 0.0f) continue;
                const float wgt = ww * wx * wy * mod;
                th -= ang;
                th += (th <  0.0f  ? PI2_F : 0.0f);
                th -= (th >= PI2_F ? PI2_F : 0.0f);
                const float tth = th * M_4RPI_F;
                const int   fo0 = (int)floorf(tth);
                const float do0 = tth - fo0;
                const int   fo  = fo0 & 7;                  // th in [0, 2 pi): fo0 in 0..8
                alt_add(bins, lane, fo, (1.0f - do0) * wgt);
                alt_add(bins, lane, fo + 1, do0 * wgt);
            }
            bins[lane] += bins[8 * PSX_WAVE + lane];
        } else {
            for (int yd = 0; yd < 16; yd++) {
                const float stepx = ix - 2.5f + 1.0f / 16.0f + xd / 8.0f;
                const float stepy = iy - 2.5f + 1.0f / 16.0f + yd / 8.0f;
                const float ptx = fmaf(cos_t, stepx, -sin_t * stepy);
                const float pty = fmaf(cos_t, stepy,  sin_t * stepx);
                float mod, th;
                alt_gradiant_rot(v, mod, th, fmaf(ptx, SBP, x), fmaf(pty, SBP, y), cos_t, sin_t);
                th += (th <  0.0f  ? PI2_F : 0.0f);
                th -= (th >= PI2_F ? PI2_F : 0.0f);
                const float ww = desc_gauss_entry(iy * 8 + yd, ix * 8 + xd);
                const float wgt = ww * desc_tile_entry(xd) * desc_tile_entry(yd) * mod;
                const float tth = th * M_4RPI_F;
                const int   fo  = (int)floorf(tth);
                const float do0 = tth - fo;
                alt_add(bins, lane, (fo + 1) & 7, wgt * do0);
                alt_add(bins, lane, fo & 7, wgt * (1.0f - do0));
            }
        }
#pragma unroll
        for (int b = 0; b < 8; b++) {
            float s = bins[b * PSX_WAVE + lane];
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) s += __shfl_down(s, d, 16);
            if (xd == 0) out[tz * 8 + b] = s;
        }
    } else {
        // notile: threads (tx 0..31, ty 0..3) of the reference = 2 passes of a wave64; wave = (pass, half of the xoff loop)
        const int tx = lane & 31, in_x = tx & 7;
        const float stepbase = -2.5f + 1.0f / 16.0f;
        const int pass = wave >> 1, xoff = wave & 1;
        const int out_y = pass * 2 + (lane >> 5);
#pragma unroll
        for (int b = 0; b < 8; b++) bins[b * PSX_WAVE + lane] = 0.0f;
        const int xd = (xoff << 3) + in_x;
        const int newx = (xoff << 3) + tx;
        for (int yd = 0; yd < 16; yd++) {
            const int newy = (out_y << 3) + yd;
            const float wgt = desc_tile_entry(xd) * desc_tile_entry(yd);
            const float stepx = stepbase + ldexpf((float)newx, -3);
            const float stepy = stepbase + ldexpf((float)newy, -3);
            const float ptx = fmaf(cos_t, stepx, -sin_t * stepy);
            const float pty = fmaf(cos_t, stepy,  sin_t * stepx);
            float mod, th;
            alt_gradiant_rot(v, mod, th, fmaf(ptx, SBP, x), fmaf(pty, SBP, y), cos_t, sin_t);
            th += (th < 0.0f ? PI2_F : 0.0f);
            const float tth = th * M_4RPI_F;
            const int   fo  = (int)floorf(tth);
            const float do0 = tth - fo;
            const int fo0 = fo & 7, fo1 = (fo0 + 1) & 7;
            const float ww = desc_gauss_entry(newy, newx) * mod;
            alt_add(bins, lane, fo0, wgt * ((1.0f - do0) * ww));
            alt_add(bins, lane, fo1, wgt * (do0 * ww));
        }
#pragma unroll
        for (int b = 0; b < 8; b++) {
            float s = bins[b * PSX_WAVE + lane];
#pragma unroll
            for (int d = 4; d >= 1; d >>= 1) s += __shfl_down(s, d, 8);
            if (in_x == 0) out[xoff * 128 + out_y * 32 + (tx >> 3) * 8 + b] = s;
        }
    }
}

// One WORKGROUP per descriptor (round 5; before: one wave per descriptor reading the plane in HBM).  These modes take 4096
// (igrid, notile), ~8000 (iloop) or 4096 (grid) samples per descriptor and every sample is four bilinear fetches = 16 texels
// (grid: 4): as 64-address gathers on the plane they kept the texture-address path busy for 1.4 ms per 1080p frame.  The
// workgroup stages the window once (coalesced rows, clamped as the texture clamps), the four waves split the tiles and fetch
// from LDS (two ds_read2_b32 per bilinear fetch).  Same samples, same order within a wave, same arithmetic.
template <int MODE>
__global__ __launch_bounds__(NT) void k_descriptors_alt(const PsxParams* __restrict__ P, const PsxCounters* cnt, const PsxExport X, const int use_window)
{
    __shared__ __attribute__((aligned(16))) float s_win[ALT_WIN_CAP];
    __shared__ float s_bins[WPB][ALT_BINS * PSX_WAVE];
    __shared__ float s_out[256];                 // [2][128]: notile sums two halves, the others use the first
    __shared__ float s_cs[2];                    // cos, sin of the descriptor's orientation
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    float* bins = s_bins[wave];

    const int total = cnt->ori_total;
    for (int jv = blockIdx.x; jv < total; jv += gridDim.x) {
        const int j = __builtin_amdgcn_readfirstlane(jv);
        const int ext_idx = P->feat_to_ext[j];
        const psx_extremum ex = P->extrema[ext_idx];
        const int ori_num = psx_clampi(j - ex.idx_ori, 0, PSX_ORI_MAX - 1);
        const float ang = ori_num == 0 ? ex.orientation[0] : ori_num == 1 ? ex.orientation[1] : ori_num == 2 ? ex.orientation[2] : ex.orientation[3];   // no scratch copy of ex
        const PsxOctave oc = P->oct[ex.octave];
        const int W = oc.w, H = oc.h, pitch = oc.pitch;
        if (ori_num == 0 && t == 0) write_feature(P, X, ext_idx, ex, ex.idx_ori, total);

        const float x = ex.xpos, y = ex.ypos;
        const float* plane = oc.data + (size_t)psx_clampi(ex.lpos, 0, P->L - 1) * oc.plane;
        const float SBP = fabsf(DESC_MAGNIFY * ex.sigma);
        s_out[t] = 0.0f;

        const int Ei = (int)ceilf(fmaf(3.5356f, SBP, 2.51f));
        const int bw = 2 * Ei + 2;
        const int bx0 = (int)floorf(x) - Ei, by0 = (int)floorf(y) - Ei;
        if (wave == WPB - 1) {
            // __sincosf in the reference.  GRID snaps its sample points through (int)(pt + (round(pt + pix) - pt)), which
            // flips on the last bit of sin / cos: that mode evaluates them in double and rounds once, as the oracle does.
            // One wave does it for the workgroup (the double-precision pair is ~600 instructions).
            const float c = MODE == PSX_DESC_GRID ? (float)cos((double)ang) : cosf(ang);
            const float sn = MODE == PSX_DESC_GRID ? (float)sin((double)ang) : sinf(ang);
            if (lane == 0) { s_cs[0] = c; s_cs[1] = sn; }
        }
        // the window's texel addresses are formed in float: 4 (bx0 + 84 by0) and every 4 i + 336 j must stay below 2^24
        const bool windowed = use_window != 0 && SBP < 64.0f && bw <= ALT_WIN_MAX && H <= 32768 && 4 * (long long)W + 336ll * (H + 84) < (1ll << 24);
        if (windowed && SBP != 0.0f) {
            for (int r = wave; r < bw; r += WPB) {
                const float* row = plane + (size_t)psx_clampi(by0 + r, 0, H - 1) * pitch;
                for (int c = lane; c < bw; c += PSX_WAVE) s_win[r * ALT_WIN_MAX + c] = row[psx_clampi(bx0 + c, 0, W - 1)];
            }
        }
        __syncthreads();

        if (SBP != 0.0f) {
            const float cos_t = s_cs[0], sin_t = s_cs[1];
            const float c4 = (float)((int)(unsigned)(uintptr_t)(const float LDS_AS*)s_win - 4 * (bx0 + ALT_WIN_MAX * by0));
            if (windowed) alt_tiles<MODE>(AltWindow{(const float LDS_AS*)s_win, bx0, by0, c4}, bins, s_out, lane, wave, x, y, ang, SBP, cos_t, sin_t);
            else          alt_tiles<MODE>(AltPlane{plane, W, H, pitch}, bins, s_out, lane, wave, x, y, ang, SBP, cos_t, sin_t);
        }
        __syncthreads();
        if (wave == 0) {
            float a = s_out[2 * lane], b = s_out[2 * lane + 1];
            if (MODE == PSX_DESC_NOTILE) { a += s_out[128 + 2 * lane]; b += s_out[128 + 2 * lane + 1]; }
            normalize_store(P, X, j, lane, a, b);
        }
        __syncthreads();                                   // s_out and the window are rewritten for the next descriptor
    }
}

} // namespace

hipError_t psx_launch_orientation(const PsxParams* d_params, PsxCounters* d_cnt, hipStream_t s)
{
    // POPSIFT_ORI_WPB=1 / 4: waves per workgroup (as k_descriptors: a keypoint's window grows with sigma^2); measured: no difference
    // (profiles/r06_desc_waves_per_workgroup.txt), four stays
    static const int wb = [] { const char* e = getenv("POPSIFT_ORI_WPB"); const int v = e ? atoi(e) : 0; return v == 1 || v == 4 ? v : 4; }();
    if (wb == 1) hipLaunchKernelGGL(k_orientation<1>, dim3(2048 * 4), dim3(PSX_WAVE), 0, s, d_params, d_cnt);
    else         hipLaunchKernelGGL(k_orientation<4>, dim3(2048), dim3(NT), 0, s, d_params, d_cnt);
    return hipGetLastError();
}

hipError_t psx_launch_scan(const PsxParams* d_params, PsxCounters* d_cnt, const PsxExport& x, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan, dim3(SCAN_WGS), dim3(SCAN_NT), 0, s, d_params, d_cnt, x);
    return hipGetLastError();
}

hipError_t psx_launch_descriptors_alt(const PsxParams* d_params, const PsxCounters* d_cnt, int desc_mode, const PsxExport& x, int cus, hipStream_t s)
{
    // 4 workgroups (38 KB of LDS each) are resident per CU; the workgroups loop over the descriptors
    static const int per_cu = [] { const char* e = getenv("POPSIFT_ALT_WGS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 64 ? v : 8; }();
    // POPSIFT_ALT_WINDOW=0: measurement / test switch, every texel from the plane in HBM (what large-sigma keypoints do anyway)
    static const int win = [] { const char* e = getenv("POPSIFT_ALT_WINDOW"); return (e != nullptr && e[0] == '0') ? 0 : 1; }();
    const dim3 grid((cus > 0 ? cus : 256) * per_cu), block(NT);
    switch (desc_mode) {
    case PSX_DESC_ILOOP:  hipLaunchKernelGGL(k_descriptors_alt<PSX_DESC_ILOOP>, grid, block, 0, s, d_params, d_cnt, x, win); break;
    case PSX_DESC_GRID:   hipLaunchKernelGGL(k_descriptors_alt<PSX_DESC_GRID>, grid, block, 0, s, d_params, d_cnt, x, win); break;
    case PSX_DESC_IGRID:  hipLaunchKernelGGL(k_descriptors_alt<PSX_DESC_IGRID>, grid, block, 0, s, d_params, d_cnt, x, win); break;
    case PSX_DESC_NOTILE: hipLaunchKernelGGL(k_descriptors_alt<PSX_DESC_NOTILE>, grid, block, 0, s, d_params, d_cnt, x, win); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t psx_launch_descriptors(const PsxParams* d_params, const PsxCounters* d_cnt, const PsxExport& x, int cus, hipStream_t s)
{
    const bool exporting = x.desc != nullptr;
    // 6 workgroups (24 waves, 26 KB of LDS each; 5 of 32 KB until round 6) are resident per CU; two full rounds measured best for
    // the descriptors of a 1080p frame (stage time 0.159 ms at 8 per CU, 0.143 at 10, 0.142 at 12, 0.147 at 15, 0.148
    // at 20).  With the zero-copy export attached every wave ends in stores that cross PCIe; fewer resident waves
    // leave room for the other streams' kernels meanwhile (3 per CU measured +11 % on the export leg of bench.py; with six resident
    // workgroups per CU, round 6: 2 per CU 5534 Mpix/s, 3: 5204, 4 and more: 5050).
    // cus = compute units of the CONTEXT's device (one PopSift per GPU may sit on unequal devices).
    if (cus <= 0) cus = 256;
    // POPSIFT_DESC_OCC=5: the instantiation padded to round 5's LDS footprint (five workgroups per CU), the A/B partner of the
    // overlapped histogram copies (six per CU: descriptor stage 0.111-0.114 -> 0.107-0.108 ms with two full rounds = 12 per CU)
    static const bool occ5 = [] { const char* e = getenv("POPSIFT_DESC_OCC"); return e != nullptr && e[0] == '5'; }();
    // POPSIFT_DESC_WGS=<workgroups per CU>: measurement switch for the grid (the waves loop over the descriptors)
    static const int per_cu = [] { const char* e = getenv("POPSIFT_DESC_WGS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 64 ? v : 0; }();
    const int grid = per_cu ? per_cu * cus : exporting ? (occ5 ? 3 : 2) * cus : (occ5 ? 10 : 12) * cus;
    // POPSIFT_DESC_DENORM=0: round 2's conversion path (v_cvt_u32_f32 of every contribution) instead of the denormal products
    static const bool denorm = [] { const char* e = getenv("POPSIFT_DESC_DENORM"); return !(e != nullptr && e[0] == '0'); }();
    // POPSIFT_DESC_WPB=1 / 2 / 4: waves per workgroup (the grid keeps its number of waves).  Default: ONE wave per workgroup (a wave's
    // LDS and slot are free the moment IT is done, not when the slowest of four is: end to end 6806-6847 -> 6872-6879 Mpix/s, device
    // resident 7005-7011 -> 7047-7058 in three A/B pairs, profiles/r06_desc_waves_per_workgroup.txt); four with the zero-copy export
    // attached (5451 against 5386 Mpix/s on that leg)
    static const int wb = [] { const char* e = getenv("POPSIFT_DESC_WPB"); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 || v == 4 ? v : 0; }();
    if (!denorm)      hipLaunchKernelGGL((k_descriptors<false, 5, 4>), dim3(grid), dim3(NT), 0, s, d_params, d_cnt, x);
    else if (occ5)    hipLaunchKernelGGL((k_descriptors<true, 5, 4>), dim3(grid), dim3(NT), 0, s, d_params, d_cnt, x);
    else if (wb == 1 || (wb == 0 && !exporting)) hipLaunchKernelGGL((k_descriptors<true, 6, 1>), dim3(grid * 4), dim3(PSX_WAVE), 0, s, d_params, d_cnt, x);
    else if (wb == 2) hipLaunchKernelGGL((k_descriptors<true, 6, 2>), dim3(grid * 2), dim3(2 * PSX_WAVE), 0, s, d_params, d_cnt, x);
    else              hipLaunchKernelGGL((k_descriptors<true, 6, 4>), dim3(grid), dim3(NT), 0, s, d_params, d_cnt, x);
    return hipGetLastError();
}
