// blur_interp.h -- the arithmetic of GaussMode VLFeat_Relative (absoluteSourceInterpolated::horiz / vert,
// s_pyramid_build_ai.cu:17-69): every pair of taps is ONE linearly filtered fetch at distance off = offset + (1 - u) on either
// side.  Shared by pyramid_interp.hip (levels >= 1: H and V pass) and pyramid.hip (level 0 of octave 0: V pass behind the
// "dd" H pass).  pyramid_interp.hip's header states why the two texels of a fetch are adjacent LDS words at a fixed distance
// and why the weight is nearly always uniform over a workgroup.
//
// Both passes walk OUTWARDS from the centre, pair by pair, and load the LDS words of a pair just before it is used (a
// sliding window of 6 columns / 5 rows per side): holding the whole window (4 + 4 NP columns x 2 rows, or rows x 2 columns)
// in registers costs up to 72 of them and the kernels spilled.  Operation order: pairs in ascending offset,
// out += (L + R) * (a + b), the centre last -- k_alt_interp's (pyramid_alt.hip), with explicit fma (-ffp-contract=off).
#pragma once

#include "blur_arith.h"

#define PSX_LDS __attribute__((address_space(3)))

// the linear filter's weight on the texel pair (kstat, kstat + 1), kstat = c - offset - 1 (left) / c + offset (right), exactly as
// readTex + the texture unit arrive at it (plane_linear_1d of pyramid_alt.hip); floor(t) = kstat + 1 happens only with
// weight 0, which is the value of weight 1 on the fixed pair
__device__ __forceinline__ float psx_lit_weight(int c, float off, int offset, bool right)
{
    const float t = right ? (float)c + off : (float)c - off;
    const float ts = t + 0.5f, tb = ts - 0.5f;
    const float ft = floorf(tb);
    const float w = rintf((tb - ft) * 256.0f) * (1.0f / 256.0f);
    const int kstat = right ? c + offset : c - offset - 1;
    return (int)ft == kstat ? w : 1.0f;
}

__device__ __forceinline__ v2f psx_splat(float x) { return (v2f){x, x}; }
// a_lerp(p, q, w) = fma(w, q, (1 - w) * p) on two lanes
__device__ __forceinline__ v2f psx_lerp2(v2f p, v2f q, float w) { return __builtin_elementwise_fma(psx_splat(w), q, psx_splat(1.0f - w) * p); }

// The table of one pass in LDS: tab[p] = (left weight, right weight, a + b, off) of pair p -- read (one broadcast
// ds_read_b128 per pair) right where the pair is evaluated, so that no weight occupies a register across the kernel.  Bit
// (2 p + side) of mask (workgroup uniform) says that the weight of (pair p, side) varies over the workgroup's columns / rows and
// is evaluated per element from its coordinate instead.

// H pass: 4 adjacent columns of 2 rows.  win: the thread's window in a row-pair interleaved staged row ([column][2]),
// chunk q = columns 2q, 2q + 1; output column e is window column HALO + e; c0 = plane column of output 0.
template <int NP, int HALO>
__device__ __forceinline__ void psx_hinterp2x4(const PSX_LDS float* win, const PSX_LDS v4f* tab, unsigned mask, float g0, int c0, v2f* out)
{
    constexpr int H2 = HALO / 2;
    static_assert(HALO >= 2 * NP && HALO % 2 == 0, "window too narrow");
    auto ld = [&](int q) __attribute__((always_inline)) { return ((const volatile PSX_LDS v4f*)win)[q]; };
    v2f lc[6], rc[6];
    {
        const v4f a = ld(H2 - 1), b = ld(H2), c = ld(H2 + 1), d = ld(H2 + 2);
        lc[0] = (v2f){a.x, a.y}; lc[1] = (v2f){a.z, a.w}; lc[2] = (v2f){b.x, b.y}; lc[3] = (v2f){b.z, b.w}; lc[4] = (v2f){c.x, c.y}; lc[5] = (v2f){c.z, c.w};
        rc[0] = lc[2]; rc[1] = lc[3]; rc[2] = lc[4]; rc[3] = lc[5]; rc[4] = (v2f){d.x, d.y}; rc[5] = (v2f){d.z, d.w};
    }
#pragma unroll
    for (int e = 0; e < 4; e++) out[e] = psx_splat(0.0f);
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int offset = 2 * p + 1;
        v4f nl, nr;
        if (p + 1 < NP) { nl = ld(H2 - p - 2); nr = ld(H2 + p + 3); }      // the next pair's new columns
        else            { nl = ld(H2); nr = ld(H2 + 1); }                  // ... the centre columns, which come last
        const v4f tp = ((const volatile PSX_LDS v4f*)tab)[p];
        const unsigned m2 = (mask >> (2 * p)) & 3u;                        // workgroup uniform
        if (m2 == 0u) {
#pragma unroll
            for (int e = 0; e < 4; e++) out[e] = pk_fma(psx_lerp2(lc[e], lc[e + 1], tp.x) + psx_lerp2(rc[e + 1], rc[e + 2], tp.y), tp.z, out[e]);
        } else {
            // a weight of this pair varies over the strip's columns (rare): per column, from the coordinate
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float wl = (m2 & 1u) ? psx_lit_weight(c0 + e, tp.w, offset, false) : tp.x;
                const float wr = (m2 & 2u) ? psx_lit_weight(c0 + e, tp.w, offset, true) : tp.y;
                out[e] = pk_fma(psx_lerp2(lc[e], lc[e + 1], wl) + psx_lerp2(rc[e + 1], rc[e + 2], wr), tp.z, out[e]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // the next pairs' loads stay behind this pair's arithmetic (register pressure)
        if (p + 1 < NP) {
            lc[5] = lc[3]; lc[4] = lc[2]; lc[3] = lc[1]; lc[2] = lc[0]; lc[0] = (v2f){nl.x, nl.y}; lc[1] = (v2f){nl.z, nl.w};
            rc[0] = rc[2]; rc[1] = rc[3]; rc[2] = rc[4]; rc[3] = rc[5]; rc[4] = (v2f){nr.x, nr.y}; rc[5] = (v2f){nr.z, nr.w};
        } else {
            out[0] = pk_fma((v2f){nl.x, nl.y}, g0, out[0]); out[1] = pk_fma((v2f){nl.z, nl.w}, g0, out[1]);
            out[2] = pk_fma((v2f){nr.x, nr.y}, g0, out[2]); out[3] = pk_fma((v2f){nr.z, nr.w}, g0, out[3]);
        }
    }
}

// V pass: 2 adjacent columns of 4 rows.  col: the thread's column pair in ring row (first output row - 2 NP); rows are
// RS floats apart; r0 = plane row of output 0.
template <int NP, int RS>
__device__ __forceinline__ void psx_vinterp2x4(const PSX_LDS float* col, const PSX_LDS v4f* tab, unsigned mask, float g0, int r0, v2f* o)
{
    constexpr int RI = 2 * NP;
    auto ld = [&](int j) __attribute__((always_inline)) { return *(const volatile PSX_LDS v2f*)(col + j * RS); };
    v2f lr[5], rr[5];
#pragma unroll
    for (int j = 0; j < 5; j++) lr[j] = ld(RI - 2 + j);
    rr[0] = lr[3]; rr[1] = lr[4];
#pragma unroll
    for (int j = 2; j < 5; j++) rr[j] = ld(RI + 1 + j);
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = psx_splat(0.0f);
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int offset = 2 * p + 1;
        v2f n0, n1, n2, n3;
        if (p + 1 < NP) { n0 = ld(RI - 2 * p - 4); n1 = ld(RI - 2 * p - 3); n2 = ld(RI + 2 * p + 6); n3 = ld(RI + 2 * p + 7); }   // the next pair's new rows
        else            { n0 = ld(RI); n1 = ld(RI + 1); n2 = ld(RI + 2); n3 = ld(RI + 3); }                                       // ... the centre rows, which come last
        const v4f tp = ((const volatile PSX_LDS v4f*)tab)[p];
        const unsigned m2 = (mask >> (2 * p)) & 3u;                        // workgroup uniform
        if (m2 == 0u) {
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = pk_fma(psx_lerp2(lr[i], lr[i + 1], tp.x) + psx_lerp2(rr[i], rr[i + 1], tp.y), tp.z, o[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float wl = (m2 & 1u) ? psx_lit_weight(r0 + i, tp.w, offset, false) : tp.x;
                const float wr = (m2 & 2u) ? psx_lit_weight(r0 + i, tp.w, offset, true) : tp.y;
                o[i] = pk_fma(psx_lerp2(lr[i], lr[i + 1], wl) + psx_lerp2(rr[i], rr[i + 1], wr), tp.z, o[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (p + 1 < NP) {
            lr[4] = lr[2]; lr[3] = lr[1]; lr[2] = lr[0]; lr[0] = n0; lr[1] = n1;
            rr[0] = rr[2]; rr[1] = rr[3]; rr[2] = rr[4]; rr[3] = n2; rr[4] = n3;
        } else {
            o[0] = pk_fma(n0, g0, o[0]); o[1] = pk_fma(n1, g0, o[1]); o[2] = pk_fma(n2, g0, o[2]); o[3] = pk_fma(n3, g0, o[3]);
        }
    }
}

// Fills the tables of both passes -- tab_h for the columns x0 .. x0 + ncol - 1 of a strip, tab_v for the rows y0 .. y0 + nrow - 1 of
// a chunk -- and returns, in s_mask[0] / s_mask[1], which (pair, side) weights vary over them.  Before the call: tab[p].z / .w
// (a + b, off) set, s_mask zeroed, a barrier.  Contains one barrier; a barrier must follow it.  nt threads, index t.
__device__ __forceinline__ void psx_interp_survey(int np, int t, int nt, int x0, int ncol, int y0, int nrow,
                                                  PSX_LDS v4f* tab_h, PSX_LDS v4f* tab_v, unsigned* s_mask)
{
    // the weight at the first coordinate: one thread per (pass, pair, side)
    if (t < 4 * np) {
        const int pass = t >= 2 * np, ts = t - pass * 2 * np, p = ts >> 1;
        PSX_LDS float* e = (PSX_LDS float*)&(pass ? tab_v : tab_h)[p];
        e[ts & 1] = psx_lit_weight(pass ? y0 : x0, e[3], 2 * p + 1, ts & 1);
    }
    __syncthreads();
    // every other coordinate against it
    const int nh = 2 * np * ncol, ntot = nh + 2 * np * nrow;
    for (int idx = t; idx < ntot; idx += nt) {
        const int pass = idx >= nh, j = idx - pass * nh, n = pass ? nrow : ncol;
        const int ts = j / n, i = j - ts * n, p = ts >> 1;
        const PSX_LDS float* e = (const PSX_LDS float*)&(pass ? tab_v : tab_h)[p];
        if (psx_lit_weight((pass ? y0 : x0) + i, e[3], 2 * p + 1, ts & 1) != e[ts & 1]) atomicOr(&s_mask[pass], 1u << ts);
    }
}
