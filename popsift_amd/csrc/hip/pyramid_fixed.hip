// pyramid_fixed.hip -- GaussMode Fixed9 / Fixed15: a whole octave in ONE kernel.
//
// Replaces (behaviour, not code) fixedSpan::absoluteTexAddress::octave_fixed and
// fixedSpan::relativeTexAddress::octave_fixed (s_pyramid_fixed.cu:69-120, 148-202; launches :217-266): every level
// of an octave is filtered straight from the octave's level 0 (octave 0: from the input image) with a 9- / 15-tap
// kernel, VERTICAL pass first, horizontal pass second.  Algorithmic traffic: level 0 read once (4 B/px) + five
// planes written (20 B/px) = 24 B per pixel and octave; octave 0 reads the input image (1/16 of a plane for u8)
// and writes six planes.  The per-level kernels of pyramid_alt.hip (k_fixed_v_* + k_fixed_h through an intermediate
// plane: 10 launches and >= 80 B/px per octave) remain the path of the configurations this kernel does not cover
// (an image / octave-0 ratio other than x2).
//
// Shape: the marching strip of pyramid.hip.  One 256-thread workgroup owns a 64-column strip and marches down a chunk of
// rows in steps of BR rows (28 at 9 taps, 24 at 15).  Per step:
//   * stage BR source rows (64 + 2 HALO columns, row-coalesced 16-byte loads issued a step ahead) into an LDS ring of 64
//     rows -- level 0 of the octave is read from HBM exactly once and serves every derived level;
//   * V pass: a thread owns (2 adjacent columns, VR rows): it reads its VR + 2 SHIFT ring rows ONCE (ds_read_b64),
//     forms the tap-pair sums once (they do not depend on the level) and runs the NLEV weight sets over them in packed
//     f32 -- the V results of all levels of the step live in registers;
//   * per level: the V results go to one of two LDS buffers (ds_write_b64), ONE barrier, then the H pass: a thread owns
//     (row, 4 adjacent columns), reads its 4 + 2 HALO window with ds_read_b128 and stores 16 bytes -- 16 lanes write 256
//     contiguous bytes of a plane row.  The level that feeds the next octave (L - 3) also stores its even rows / columns
//     into level 0 of the next octave (get_by_2_pick_every_second, s_pyramid_build.cu:50-71).
// Barriers per step: one per level.  LDS: 40.7 KB (9 taps) / 46.8 KB (15 taps).
//
// Arithmetic order is the reference's, with explicit fma (this file is compiled -ffp-contract=off), identical to
// pyramid_alt.hip and oracle/sift_oracle.c: V: val(y) * g0, then += (val(y-i) + val(y+i)) * g[i], i ascending; H the same
// over the V results; octave 0: x 255 last.  Planes are bit-identical (tests/test_gpu_modes.py, the ref_mode_fixed*
// and ref_fixed15_* fixtures).
//
// Octave 0 at the x2 upsampling (W = 2w, H = 2h): the reference fetches tex2D(((cx + 1) / W, (y + 1 -+ i) / H) of the
// normalised, clamped, linearly filtered input.  With W = 2w these coordinates sit on the half-texel grid, so the 1.8
// fixed-point weights are the constants {0, 1/2} (k_level0_x2 in pyramid.hip states the argument, including the
// knife edge where the general formula lands on (i0 - 1, weight 1): the value is T[i0] either way), rows and columns
// outside the plane are the clamped TEXELS (U(-1) = lerp(T[0], T[0], 1/2) = T[0] exactly), and the fetch of tap i at row
// y is U(cx, y + i) of ONE virtual plane U -- which the staging phase computes from the texels (3 x 3 texels per 4 x 4
// block of U) straight into the ring.
#include "psx_internal.h"
#include "blur_arith.h"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int TW = 64;    // strip width (columns per workgroup)
constexpr int NT = 256;   // threads per workgroup

#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))

enum { SRC_PLANE = 0, SRC_U8X2 = 1, SRC_F32X2 = 2 };

template <int S>
struct GeomF {
    static constexpr int HALO = (S + 3) & ~3;             // 4 / 8
    static constexpr int SW   = TW + 2 * HALO;            // staged columns: 72 / 80
    static constexpr int SW4  = SW / 4;
    static constexpr int NP   = SW / 2;                   // column pairs of the V pass: 36 / 40
    static constexpr int VR   = 4;                        // rows per V thread (two row pairs)
    static constexpr int NRG  = NT / NP;                  // row groups: 7 / 6
    static constexpr int BR   = VR * NRG;                 // rows per step: 28 / 24
    static constexpr int RING = 64;
    static constexpr int VWIN = VR + 2 * S;               // ring rows a V thread reads
    static constexpr int MIRROR = VWIN - 1;               // slots < MIRROR are duplicated at slot + RING
    static constexpr int RSS  = SW + 4;                   // ring row stride (floats), 16-byte aligned rows
    // V results in LDS: [row pair][column][row of the pair] -- the H pass is packed over two ROWS, so that every tap pair
    // (column c - i, column c + i) of both rows is an aligned register pair whatever the parity of i (packing adjacent
    // columns needs a register move for every odd tap).  A row pair is VS2 floats long; the H pass reads 16-byte chunks
    // (2 columns x 2 rows) with lanes = (row pair, column quad 0..15): a lane's chunks are 32 bytes apart, and a
    // ds_read_b128 lane group ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) takes quads {0-3, 12-15} of one row pair and
    // quads {4-11} of the next: conflict free when consecutive row pairs are an odd number of 16-byte chunks apart
    static constexpr int VS2  = 2 * SW + 4;               // 148 / 164 floats
    static constexpr int NRP  = BR / 2;                   // row pairs per step: 14 / 12
    static constexpr int NWIN = (4 + 2 * HALO) / 2;       // 16-byte chunks of an H window: 6 / 10
    static constexpr int NLD  = (BR * SW4 + NT - 1) / NT; // staging slots per thread (plane source)
    static constexpr int NGRP = (BR + 3) / 4 + 1;         // aligned 4-row groups of U a step can touch (image source)
    static constexpr int LDS_BYTES = ((RING + MIRROR) * RSS + 2 * NRP * VS2) * 4;
    static constexpr int WGPC = (160 * 1024) / LDS_BYTES > 4 ? 4 : (160 * 1024) / LDS_BYTES;
    static_assert(BR + 2 * S <= RING, "ring too small");
    static_assert(NGRP * SW4 <= NT, "one 4 x 4 block of U per thread");
    static_assert((VS2 / 4) % 2 == 1 && NRP * 16 <= NT, "H pass layout");
};

template <int S, int NLEV>
struct FixedArgs {
    const void* src;            // SRC_PLANE: level 0 of the octave (pitch floats per row); otherwise the input image
    int src_w, src_h;           // image size (SRC_U8X2 / SRC_F32X2)
    float* dst;                 // plane of the first level written
    size_t plane;               // floats between two levels
    float* half_dst;            // level 0 of the next octave, or nullptr
    int half_pitch, half_level; // index (0 .. NLEV-1) of the level that is decimated into half_dst
    int W, H, pitch;
    int nstrips, chunk_rows;
    float scale;                // 255 for octave 0 (the image is normalised to [0, 1]), 1 otherwise
    float g[NLEV][S + 1];
};

// compile-time loop over the levels (the V results are indexed by the level: they must stay in registers)
template <int L, int N, class F> __device__ __forceinline__ void level_loop_n(F&& f)
{
    if constexpr (L < N) { f(std::integral_constant<int, L>{}); level_loop_n<L + 1, N>(f); }
}

template <int S, int NLEV, int SRC>
__global__ __launch_bounds__(NT, GeomF<S>::WGPC) void k_fixed_octave(FixedArgs<S, NLEV> a)
{
    using G = GeomF<S>;
    constexpr int HALO = G::HALO, SW4 = G::SW4, NP = G::NP, VR = G::VR, NRG = G::NRG, BR = G::BR, RING = G::RING;
    constexpr int VWIN = G::VWIN, MIRROR = G::MIRROR, RSS = G::RSS, VS2 = G::VS2, NRP = G::NRP, NWIN = G::NWIN, NLD = G::NLD, NGRP = G::NGRP;
    constexpr bool ISFLOAT = SRC == SRC_F32X2;
    __shared__ __attribute__((aligned(16))) float s_ring[(RING + MIRROR) * RSS];
    __shared__ __attribute__((aligned(16))) float s_v[2 * NRP * VS2];

    const int t     = threadIdx.x;
    const int lid   = psx_xcd_remap(blockIdx.x, gridDim.x);
    const int strip = lid % a.nstrips;
    const int chunk = lid / a.nstrips;
    const int x0    = strip * TW;
    const int Y0    = chunk * a.chunk_rows;
    const int Y1    = min(Y0 + a.chunk_rows, a.H);
    const int nsteps = (Y1 - Y0 + 2 * S + BR - 1) / BR;

    // one staged row (4 floats at column quad c4) into the ring, mirrored behind its end when it is one of the first rows
    auto ring_store = [&](int ridx, int c4, const v4f& q) __attribute__((always_inline)) {
        const int slot = ridx & (RING - 1);
        float* rp = &s_ring[slot * RSS + c4 * 4];
        *reinterpret_cast<v4f*>(rp) = q;
        if (slot < MIRROR) *reinterpret_cast<v4f*>(rp + RING * RSS) = q;
    };

    // ---- V pass geometry: thread = (column pair, VR rows) ----
    const int v_p = t % NP, v_rg = t / NP;
    const bool v_on = v_rg < NRG;
    // ---- H pass geometry: thread = (row pair, column quad): 4 columns of 2 rows ----
    const int h_quad = t & 15, h_rp = t >> 4;
    const bool h_on = h_rp < NRP;
    const int h_x = x0 + 4 * h_quad;
    GLOBAL_AS float* const gdst = (GLOBAL_AS float*)a.dst;
    GLOBAL_AS float* const ghalf = (GLOBAL_AS float*)a.half_dst;

    // ---- the arithmetic of one step, shared by both sources ----
    v2f vo[NLEV][VR];
    auto vpass = [&](const int k) __attribute__((always_inline)) {
        if (!v_on) return;
        const int rel0 = k * BR - 2 * S + v_rg * VR;          // ring index of the first row of the window
        const LDS_AS float* vp = (const LDS_AS float*)&s_ring[(rel0 & (RING - 1)) * RSS + 2 * v_p];
        v2f v[VWIN];
#pragma unroll
        for (int j = 0; j < VWIN; j++) v[j] = *(const volatile LDS_AS v2f*)(vp + j * RSS);
        // the tap-pair sums val(y - q) + val(y + q) are the same for every level
        v2f sum[VR][S];
#pragma unroll
        for (int i = 0; i < VR; i++)
#pragma unroll
            for (int q = 1; q <= S; q++) sum[i][q - 1] = v[S + i - q] + v[S + i + q];
        // level by level (VR independent chains each): only one level's weights are live at a time
#pragma unroll
        for (int l = 0; l < NLEV; l++) {
#pragma unroll
            for (int i = 0; i < VR; i++) vo[l][i] = v[S + i] * (v2f){a.g[l][0], a.g[l][0]};
#pragma unroll
            for (int q = 1; q <= S; q++)
#pragma unroll
                for (int i = 0; i < VR; i++) vo[l][i] = pk_fma(sum[i][q - 1], a.g[l][q], vo[l][i]);
        }
    };
    auto vwrite = [&](const int l, const int buf) __attribute__((always_inline)) {
        if (!v_on) return;
        float* bp = &s_v[buf * NRP * VS2 + (v_rg * (VR / 2)) * VS2 + 4 * v_p];
#pragma unroll
        for (int i = 0; i < VR; i += 2)     // (row i, row i+1) of column 2 v_p, then of column 2 v_p + 1: a 2 x 2 transpose in registers
            *reinterpret_cast<v4f*>(bp + (i / 2) * VS2) = (v4f){vo[l][i].x, vo[l][i + 1].x, vo[l][i].y, vo[l][i + 1].y};
    };
    // FAST (workgroup uniform): every row of the step lies inside the chunk and the strip is a full one -- all but the first
    // step of a chunk, its last one and the plane's last strip
    auto hpass = [&](const int k, const int l, const int buf, auto fast_c) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_c)::value;
        if (!h_on) return;
        v2f win[4 + 2 * HALO];                                           // (row 0, row 1) of the window's columns
        {
            const LDS_AS float* hp = (const LDS_AS float*)&s_v[buf * NRP * VS2 + h_rp * VS2 + 8 * h_quad];
#pragma unroll
            for (int q = 0; q < NWIN; q++) {
                const v4f w4 = ((const volatile LDS_AS v4f*)hp)[q];
                win[2 * q] = (v2f){w4.x, w4.y}; win[2 * q + 1] = (v2f){w4.z, w4.w};
            }
        }
        v2f out[4];
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = win[HALO + e] * (v2f){a.g[l][0], a.g[l][0]};
#pragma unroll
        for (int i = 1; i <= S; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) out[e] = pk_fma(win[HALO + e - i] + win[HALO + e + i], a.g[l][i], out[e]);
        if (SRC != SRC_PLANE) {
#pragma unroll
            for (int e = 0; e < 4; e++) out[e] = out[e] * (v2f){a.scale, a.scale};
        }
        const int r = Y0 + k * BR - 2 * S + 2 * h_rp;                    // rows r, r + 1
        GLOBAL_AS float* dp = gdst + (size_t)l * a.plane + (size_t)r * a.pitch + h_x;
        const bool halve = l == a.half_level && ghalf != nullptr;         // uniform
        // get_by_2_pick_every_second: rows and columns 0, 2, 4, .. of level L - 3 are level 0 of the next octave; which of
        // the pair's rows is the even one is workgroup uniform (Y0 + k BR - 2 S is)
        const int re = (r & 1) ? r + 1 : r;
        const v2f hv = (r & 1) ? (v2f){out[0].y, out[2].y} : (v2f){out[0].x, out[2].x};
        GLOBAL_AS float* hd = ghalf + (size_t)(re >> 1) * a.half_pitch + (h_x >> 1);
        if constexpr (FAST) {
            *reinterpret_cast<GLOBAL_AS v4f*>(dp) = (v4f){out[0].x, out[1].x, out[2].x, out[3].x};
            *reinterpret_cast<GLOBAL_AS v4f*>(dp + a.pitch) = (v4f){out[0].y, out[1].y, out[2].y, out[3].y};
            if (halve) *reinterpret_cast<GLOBAL_AS v2f*>(hd) = hv;
        } else {
            if (h_x < a.W) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (r + j < Y0 || r + j >= Y1) continue;
                    GLOBAL_AS float* dj = dp + (size_t)j * a.pitch;
                    const v4f o = j ? (v4f){out[0].y, out[1].y, out[2].y, out[3].y} : (v4f){out[0].x, out[1].x, out[2].x, out[3].x};
                    if (h_x + 3 < a.W) *reinterpret_cast<GLOBAL_AS v4f*>(dj) = o;
                    else { dj[0] = o.x; if (h_x + 1 < a.W) dj[1] = o.y; if (h_x + 2 < a.W) dj[2] = o.z; }
                }
                if (halve && re >= Y0 && re < Y1) {
                    if (h_x + 2 < a.W) *reinterpret_cast<GLOBAL_AS v2f*>(hd) = hv;
                    else hd[0] = hv.x;
                }
            }
        }
    };
    // steps: V of step k, then per level: V results -> LDS buffer, barrier, (first level: the next step's rows -> ring,
    // whose previous tenants every thread has finished reading once it is past this barrier), H pass + stores
    auto march = [&](auto&& issue, auto&& commit) __attribute__((always_inline)) {
        issue(0);
        commit(0);
        __syncthreads();
        int buf = 0;
        for (int k = 0; k < nsteps; k++) {
            if (k + 1 < nsteps) issue(k + 1);
            const bool fast = x0 + TW <= a.W && k * BR >= 2 * S && Y0 + (k + 1) * BR - 2 * S <= Y1;
            vpass(k);
            level_loop_n<0, NLEV>([&](auto lc) __attribute__((always_inline)) {
                constexpr int l = decltype(lc)::value;
                vwrite(l, buf);
                __syncthreads();
                if (l == 0 && k + 1 < nsteps) commit(k + 1);
                if (fast) hpass(k, l, buf, std::true_type{}); else hpass(k, l, buf, std::false_type{});
                buf ^= 1;
            });
        }
    };

    if constexpr (SRC == SRC_PLANE) {
        const GLOBAL_AS float* const gsrc = (const GLOBAL_AS float*)a.src;
        // every staged column exists in the source row (no clamping needed): workgroup uniform
        const bool interior = (x0 - HALO >= 0) && (x0 + TW + HALO <= a.W);
        constexpr bool LAST_PARTIAL = (BR * SW4) % NT != 0;
        int st_row[NLD], st_x[NLD], st_c4[NLD];
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int idx = t + j * NT;
            st_row[j] = idx / SW4; st_c4[j] = idx - st_row[j] * SW4;
            st_x[j] = x0 - HALO + st_c4[j] * 4;
        }
        const bool last_on = !LAST_PARTIAL || (t + (NLD - 1) * NT < BR * SW4);
        v4f pre[NLD];
        float pre_l[NLD], pre_r[NLD];
        auto issue = [&](const int k) __attribute__((always_inline)) {
            const int ybase = Y0 - S + k * BR;
#pragma unroll
            for (int j = 0; j < NLD; j++) {
                if (j < NLD - 1 || last_on) {
                    const int y = psx_clampi(ybase + st_row[j], 0, a.H - 1);
                    const GLOBAL_AS float* rp = gsrc + (size_t)y * a.pitch;
                    if (interior) pre[j] = *reinterpret_cast<const GLOBAL_AS v4f*>(rp + st_x[j]);
                    else {
                        // edge strip: the 16-byte slot at the clamped in-row position plus the row's first / last pixel;
                        // the columns outside the plane are patched in commit (as k_blur does it)
                        const int xc = psx_clampi(st_x[j], 0, a.pitch - 4);
                        pre[j] = *reinterpret_cast<const GLOBAL_AS v4f*>(rp + xc);
                        pre_l[j] = rp[0]; pre_r[j] = rp[a.W - 1];
                    }
                }
            }
        };
        auto commit = [&](const int k) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NLD; j++) {
                if (j < NLD - 1 || last_on) {
                    v4f q = pre[j];
                    if (!interior) {
                        const int x = st_x[j];
                        q.x = x + 0 < 0 ? pre_l[j] : (x + 0 > a.W - 1 ? pre_r[j] : q.x);
                        q.y = x + 1 < 0 ? pre_l[j] : (x + 1 > a.W - 1 ? pre_r[j] : q.y);
                        q.z = x + 2 < 0 ? pre_l[j] : (x + 2 > a.W - 1 ? pre_r[j] : q.z);
                        q.w = x + 3 < 0 ? pre_l[j] : (x + 3 > a.W - 1 ? pre_r[j] : q.w);
                    }
                    ring_store(k * BR + st_row[j], st_c4[j], q);
                }
            }
        };
        march(issue, commit);
    } else {
        // ---- octave 0 at x2: the thread's 4 x 4 block of U (aligned group of 4 rows x column quad) from 3 x 3 texels ----
        const int gs = t / SW4, cq = t - gs * SW4;
        const bool active = gs < NGRP;
        const int X0c = x0 - HALO + 4 * cq;                       // first column of the quad (a multiple of 4)
        const int cfirst = X0c >> 1;                              // first texel column the quad reads
        const int cbase = psx_clampi(cfirst, 0, a.src_w - 4);     // position of the 4-texel load (host: w >= 4)
        int csel[3];
#pragma unroll
        for (int i = 0; i < 3; i++) csel[i] = psx_clampi(cfirst + i, 0, a.src_w - 1) - cbase;
        typedef typename std::conditional<ISFLOAT, v4f, unsigned>::type texel4;
        texel4 pre[3];
        auto issue = [&](const int k) __attribute__((always_inline)) {
            if (!active) return;
            const int ybase = Y0 - S + k * BR;
            const int g = (ybase >> 2) + gs;                      // U rows 4g .. 4g+3 <- texel rows 2g .. 2g+2 (clamped)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int jr = psx_clampi(2 * g + i, 0, a.src_h - 1);
                if (ISFLOAT) __builtin_memcpy(&pre[i], static_cast<const float*>(a.src) + (size_t)jr * a.src_w + cbase, 16);
                else         __builtin_memcpy(&pre[i], static_cast<const uint8_t*>(a.src) + (size_t)jr * a.src_w + cbase, 4);
            }
        };
        // the linear filter with the constant weights of the x2 grid (a_lerp of pyramid_alt.hip with weight 1/2 / 0)
        auto half_ = [](float p, float q) { return fmaf(0.5f, q, 0.5f * p); };
        auto same_ = [](float p) { return ISFLOAT ? p + 0.0f : p; };
        // u8 texel -> v / 255 correctly rounded (a_texel of pyramid_alt.hip, l0_unorm8 of pyramid.hip)
        auto unorm8 = [](unsigned q) { const float c = 1.0f / 255.0f, f = (float)q, r = f * c; return fmaf(fmaf(-255.0f, r, f), c, r); };
        auto commit = [&](const int k) __attribute__((always_inline)) {
            if (!active) return;
            const int ybase = Y0 - S + k * BR;
            const int g = (ybase >> 2) + gs;
            float ux[3][4];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                float c[3];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    if constexpr (ISFLOAT) {
                        const float lo = csel[q] & 1 ? pre[i].y : pre[i].x, hi = csel[q] & 1 ? pre[i].w : pre[i].z;
                        c[q] = csel[q] & 2 ? hi : lo;
                    } else c[q] = unorm8((pre[i] >> (8 * csel[q])) & 0xffu);
                }
                ux[i][0] = same_(c[0]); ux[i][1] = half_(c[0], c[1]); ux[i][2] = same_(c[1]); ux[i][3] = half_(c[1], c[2]);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = 4 * g + q - ybase;                  // row of the step
                if (r < 0 || r >= BR) continue;
                v4f o;
                if (q == 0)      o = (v4f){same_(ux[0][0]), same_(ux[0][1]), same_(ux[0][2]), same_(ux[0][3])};
                else if (q == 1) o = (v4f){half_(ux[0][0], ux[1][0]), half_(ux[0][1], ux[1][1]), half_(ux[0][2], ux[1][2]), half_(ux[0][3], ux[1][3])};
                else if (q == 2) o = (v4f){same_(ux[1][0]), same_(ux[1][1]), same_(ux[1][2]), same_(ux[1][3])};
                else             o = (v4f){half_(ux[1][0], ux[2][0]), half_(ux[1][1], ux[2][1]), half_(ux[1][2], ux[2][2]), half_(ux[1][3], ux[2][3])};
                ring_store(k * BR + r, cq, o);
            }
        };
        march(issue, commit);
    }
}

inline int device_cus()
{
    static const int n = [] { int d = 0, c = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256; return c; }();
    return n;
}

// Rows per chunk.  A large plane is ONE round of resident workgroups (a second, nearly empty round would double the
// launch), its chunks as long as that allows (the 2 SHIFT warm-up rows of a chunk are recomputed work); a small plane
// is cut into chunks of two steps so that it still spreads over the chip.  POPSIFT_FIXED_WGS overrides the number of
// workgroups aimed at (measurement switch).
template <int S>
void fixed_chunking(int W, int H, int& chunk_rows, int& nchunks)
{
    using G = GeomF<S>;
    const int nstrips = (W + TW - 1) / TW;
    static const int want = [] { const char* e = getenv("POPSIFT_FIXED_WGS"); return e ? atoi(e) : 0; }();
    const int slots = want > 0 ? want : G::WGPC * device_cus();
    int maxc = slots / nstrips; if (maxc < 1) maxc = 1;
    // POPSIFT_FIXED_MINSTEPS: steps per chunk of a plane that does not fill the chip (measurement switch)
    static const int minsteps = [] { const char* e = getenv("POPSIFT_FIXED_MINSTEPS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 8 ? v : 1; }();
    const int cr_min = minsteps * G::BR - 2 * S;
    int nc = (H + cr_min - 1) / cr_min; if (nc > maxc) nc = maxc; if (nc < 1) nc = 1;
    int cr = (H + nc - 1) / nc;
    // full steps: a chunk of n steps yields n BR - 2 SHIFT rows
    const int ns = (cr + 2 * S + G::BR - 1) / G::BR;
    cr = ns * G::BR - 2 * S;
    if (cr > H) cr = H;
    chunk_rows = cr;
    nchunks = (H + cr - 1) / cr;
}

template <int S, int NLEV, int SRC>
hipError_t launch_fixed(const PsxFixedOctaveArgs& h, hipStream_t s)
{
    FixedArgs<S, NLEV> a;
    a.src = h.src; a.src_w = h.src_w; a.src_h = h.src_h;
    a.dst = h.dst; a.plane = h.plane;
    a.half_dst = h.half_dst; a.half_pitch = h.half_pitch; a.half_level = h.half_level;
    a.W = h.W; a.H = h.H; a.pitch = h.pitch;
    a.nstrips = (h.W + TW - 1) / TW;
    int nchunks;
    fixed_chunking<S>(h.W, h.H, a.chunk_rows, nchunks);
    a.scale = h.scale;
    for (int l = 0; l < NLEV; l++)
        for (int i = 0; i <= S; i++) a.g[l][i] = h.taps[l * PSX_GAUSS_ALIGN + i];
    const dim3 grid(a.nstrips * nchunks), block(NT);
    if (h.ev0 != nullptr || h.ev1 != nullptr) hipExtLaunchKernelGGL((k_fixed_octave<S, NLEV, SRC>), grid, block, 0, s, h.ev0, h.ev1, 0, a);
    else                                      hipLaunchKernelGGL((k_fixed_octave<S, NLEV, SRC>), grid, block, 0, s, a);
    return hipGetLastError();
}

} // namespace

// true when psx_launch_fixed_octave covers octave 0 of this configuration (otherwise: the per-level kernels of pyramid_alt.hip)
bool psx_fixed_octave0_ok(int w, int h, int W, int H)
{
    static const bool off = [] { const char* e = getenv("POPSIFT_FIXED_FUSED"); return e != nullptr && e[0] == '0'; }();
    // up to 4096 texels per side the float coordinates stay within 1e-3 texel of the half-texel grid, far from the 1/512
    // rounding boundaries of the 1.8 weight (tests/test_numeric_tricks_cpu.py checks every column); larger images: the literal kernels
    return !off && W == 2 * w && H == 2 * h && w >= 4 && w <= 4096 && h <= 4096;
}
bool psx_fixed_octave_enabled()
{
    static const bool off = [] { const char* e = getenv("POPSIFT_FIXED_FUSED"); return e != nullptr && e[0] == '0'; }();
    return !off;
}

hipError_t psx_launch_fixed_octave(const PsxFixedOctaveArgs& h, hipStream_t s)
{
    if (h.shift != 4 && h.shift != 7) return hipErrorInvalidValue;
    if (h.from_input) {
        if (h.nlev != 6) return hipErrorInvalidValue;
        if (h.shift == 4) return h.is_float ? launch_fixed<4, 6, SRC_F32X2>(h, s) : launch_fixed<4, 6, SRC_U8X2>(h, s);
        return h.is_float ? launch_fixed<7, 6, SRC_F32X2>(h, s) : launch_fixed<7, 6, SRC_U8X2>(h, s);
    }
    if (h.nlev != 5) return hipErrorInvalidValue;
    return h.shift == 4 ? launch_fixed<4, 5, SRC_PLANE>(h, s) : launch_fixed<7, 5, SRC_PLANE>(h, s);
}
