// pyramid_interp.hip -- GaussMode VLFeat_Relative: the separable blur whose tap PAIRS are single linearly filtered
// fetches (absoluteSourceInterpolated::horiz / vert, s_pyramid_build_ai.cu:17-69), fused H + V in one marching-strip
// kernel per level.  The reference reads, for every pair of taps (offset, offset + 1), ONE texel of a linear-filtering
// texture at distance off = offset + (1 - u) on either side, u = a / (a + b) for the pair's weights a, b, and multiplies the sum by
// (a + b) (GaussTable::transformBlurTable).  The texture unit's linear filter has 1.8 fixed-point weights, so the result is
// not the plain blur: this mode is its own numerical result (its own branch in oracle/sift_oracle.c).  pyramid_alt.hip holds the
// one-thread-per-pixel kernels (k_alt_interp, two launches per level through an intermediate plane) that this file replaces
// for pair counts up to 8 (sigma up to ~4); they remain the path beyond.
//
// The fetch as arithmetic.  Output coordinate c (a column in the H pass, a row in the V pass), pair p, offset = 2p + 1:
//     left :  t = fl(c - off), texels k = floor(t), k + 1, weight w = rint(frac(t) * 256) / 256
//     right:  t = fl(c + off)                                    (readTex adds 0.5, the unit takes it off again: both kept)
// off lies in [offset, offset + 1], so k is c - offset - 1 (or c - offset with w = 0, which is the same value as weight 1 on
// the fixed texel pair (c - offset - 1, c - offset)); likewise the right pair is (c + offset, c + offset + 1).  The two texels of
// a fetch are therefore ADJACENT LDS WORDS AT A FIXED DISTANCE; only the weight comes out of the float arithmetic, and it
// depends on c through the rounding of c -+ off alone: it is constant wherever c -+ off stays inside one binade, i.e. almost
// everywhere.  Each workgroup evaluates the literal weight of every (pair, side) for all of its 64 columns and all of
// its chunk's rows once, at its start; where they agree (nearly always) the weight is a workgroup-uniform scalar, and a
// (pair, side) that does vary across the strip / the chunk is evaluated per element, literally, for that pair only.
// Planes are bit-identical to k_alt_interp's (tests/test_gpu_modes.py, ref_mode_relative* fixtures).
//
// Shape: the marching strip of pyramid.hip (64 columns x steps of 32 rows, 256 threads).  The staged rows are stored
// ROW-PAIR INTERLEAVED ([row pair][column][2]) and the H pass is packed over the two rows of a pair, so that the texel pairs
// of every tap -- at any distance, odd or even -- are aligned register pairs (v_pk_fma_f32 without moves); the V pass is
// packed over two adjacent columns as in k_blur.  Operation order: pairs in ascending offset, out += (L + R) * (a + b), the
// centre last -- k_alt_interp's, with explicit fma (-ffp-contract=off).
#include "psx_internal.h"
#include "blur_arith.h"
#include "blur_interp.h"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int TW = 64, BR = 32, NT = 256;
#ifndef PSX_INTERP_WGPC_BIG
#define PSX_INTERP_WGPC_BIG 4
#endif

#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))

template <int NP>
struct GeomI {
    static constexpr int RI   = 2 * NP;                   // largest texel distance of a fetch
    static constexpr int HALO = (RI + 3) & ~3;
    static constexpr int SW   = TW + 2 * HALO;
    static constexpr int SW4  = SW / 4;
    static constexpr int SS   = 2 * SW + 4;               // staged row pair (floats): consecutive pairs an odd number of 16-byte chunks apart
    static constexpr int NTASK = (BR / 2) * SW4;          // staging tasks (row pair, column quad), two 16-byte loads each
    static constexpr int NLD  = (NTASK + NT - 1) / NT;
    static constexpr int NWIN = (4 + 2 * HALO) / 2;       // 16-byte chunks of an H window (4 + 2 HALO columns x 2 rows)
    static constexpr int RING = 64;
    static constexpr int VWIN = 4 + 2 * RI;               // ring rows a V thread reads
    static constexpr int MIRROR = VWIN - 1;
    static constexpr int RS   = TW + 4;                   // ring row stride (floats)
    static constexpr int LDS_BYTES = ((BR / 2) * SS + (RING + MIRROR) * RS) * 4;
    // 4 workgroups per CU (128 registers) up to 4 pairs; beyond, the two windows (H: 4 + 2 RI columns x 2 rows, V: 4 + 2 RI rows x
    // 2 columns) want up to 168
    static constexpr int WGPC = PSX_INTERP_WGPC_BIG > 0 && NP >= 5 ? PSX_INTERP_WGPC_BIG : 4;
    static_assert(BR + 2 * RI <= RING, "ring too small");
    static_assert(WGPC * LDS_BYTES <= 160 * 1024, "LDS");
    static_assert((SS / 4) % 2 == 1, "row pairs must be an odd number of chunks apart");
};

template <int NP>
struct InterpArgs {
    const float* src;
    float*       dst;
    float*       half_dst;          // next octave level 0 (pick every second), or nullptr
    int W, H, pitch, half_pitch;
    int nstrips, chunk_rows;
    int force_literal;              // test switch (POPSIFT_INTERP_LITERAL=1): every weight from its coordinate, as if none were uniform
    float g0;                       // centre weight
    float mul[NP];                  // a + b of pair p
    float off[NP];                  // offset + (1 - u) of pair p, offset = 2p + 1
};

template <int NP>
__device__ __forceinline__ void interp_body(const InterpArgs<NP>& a, const int lid, float* const s_stage, float* const s_ring,
                                            v4f (*s_tab)[NP], unsigned* const s_mask)
{
    using G = GeomI<NP>;
    constexpr int RI = G::RI, HALO = G::HALO, SW = G::SW, SW4 = G::SW4, SS = G::SS, NTASK = G::NTASK, NLD = G::NLD;
    constexpr int RING = G::RING, MIRROR = G::MIRROR, RS = G::RS;

    const int t     = threadIdx.x;
    const int strip = lid % a.nstrips;
    const int chunk = lid / a.nstrips;
    const int x0    = strip * TW;
    const int Y0    = chunk * a.chunk_rows;
    const int Y1    = min(Y0 + a.chunk_rows, a.H);
    const int nsteps = (Y1 - Y0 + 2 * RI + BR - 1) / BR;
    const GLOBAL_AS float* const gsrc = (const GLOBAL_AS float*)a.src;
    GLOBAL_AS float* const gdst = (GLOBAL_AS float*)a.dst;
    GLOBAL_AS float* const ghalf = (GLOBAL_AS float*)a.half_dst;

    // ---- the weight tables of this workgroup's columns / rows ----
    if (t < 2) s_mask[t] = 0u;
    if (t == 0) {
#pragma unroll
        for (int p = 0; p < NP; p++) { s_tab[0][p] = (v4f){0.0f, 0.0f, a.mul[p], a.off[p]}; s_tab[1][p] = (v4f){0.0f, 0.0f, a.mul[p], a.off[p]}; }
    }
    __syncthreads();

    // every staged column exists in the source row (no clamping needed): workgroup uniform
    const bool interior = (x0 - HALO >= 0) && (x0 + TW + HALO <= a.W);
    // ---- staging geometry: task = (row pair, column quad): rows 2 rp, 2 rp + 1 of the step, 4 columns ----
    int st_rp[NLD], st_c4[NLD], st_x[NLD];
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int idx = t + j * NT;
        st_rp[j] = idx / SW4; st_c4[j] = idx - st_rp[j] * SW4;
        st_x[j] = x0 - HALO + st_c4[j] * 4;
    }
    constexpr bool LAST_PARTIAL = NTASK % NT != 0;
    const bool last_on = !LAST_PARTIAL || (t + (NLD - 1) * NT < NTASK);
    v4f pre[NLD][2];
    auto issue = [&](const int k) __attribute__((always_inline)) {
        const int ybase = Y0 - RI + k * BR;
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            if (j < NLD - 1 || last_on) {
                // edge strips load the 16-byte slot at the clamped in-row position; their columns outside the plane are patched in LDS
                const int xc = interior ? st_x[j] : psx_clampi(st_x[j], 0, a.pitch - 4);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int y = psx_clampi(ybase + 2 * st_rp[j] + h, 0, a.H - 1);
                    pre[j][h] = *reinterpret_cast<const GLOBAL_AS v4f*>(gsrc + (size_t)y * a.pitch + xc);
                }
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            if (j < NLD - 1 || last_on) {
                float* sp = &s_stage[st_rp[j] * SS + st_c4[j] * 8];
                *reinterpret_cast<v4f*>(sp)     = (v4f){pre[j][0].x, pre[j][1].x, pre[j][0].y, pre[j][1].y};
                *reinterpret_cast<v4f*>(sp + 4) = (v4f){pre[j][0].z, pre[j][1].z, pre[j][0].w, pre[j][1].w};
            }
        }
    };
    // edge strips: staged columns [0, e_l) take the value of column e_l, columns [e_r, SW) that of column e_r - 1 (the clamp of
    // the texture addressing): patched in LDS once the step's rows have landed -- one more barrier, on the plane's first and
    // last strip only
    const int e_l = min(max(HALO - x0, 0), SW - 1), e_r = max(min(a.W - (x0 - HALO), SW), 1);
    auto patch = [&]() __attribute__((always_inline)) {
        for (int idx = t; idx < (BR / 2) * SW; idx += NT) {
            const int rp = idx / SW, c = idx - rp * SW;
            v2f* row = reinterpret_cast<v2f*>(&s_stage[rp * SS]);
            if (c < e_l) row[c] = row[e_l];
            else if (c >= e_r) row[c] = row[e_r - 1];
        }
    };

    // ---- H pass geometry: thread = (row pair, column quad) ----
    const int h_quad = t & 15, h_rp = t >> 4;
    const int h_c0 = x0 + 4 * h_quad;                         // first output column of the thread
    const LDS_AS float* const h_src = (const LDS_AS float*)&s_stage[h_rp * SS + 8 * h_quad];
    // ---- V pass geometry: thread = (2 adjacent columns, 4 output rows) ----
    const int v_pp = t & 31, v_rg = t >> 5;
    const int v_x  = x0 + 2 * v_pp;
    const bool v_xok = v_x < a.W, v_pair = v_x + 1 < a.W;

    v2f pend[4];
    auto flush = [&](const int kk) __attribute__((always_inline)) {
        const int r_out0 = Y0 + kk * BR - 2 * RI + v_rg * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r_out = r_out0 + i;
            if (r_out >= Y0 && r_out < Y1 && v_xok) {
                GLOBAL_AS float* di = gdst + (size_t)r_out * a.pitch + v_x;
                if (v_pair) {
                    // system-scope store: written through the XCD's L2 while the kernel runs (k_blur measured the same choice)
                    unsigned long long bits; __builtin_memcpy(&bits, &pend[i], 8);
                    __hip_atomic_store(reinterpret_cast<GLOBAL_AS unsigned long long*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else *di = pend[i].x;
                // get_by_2_pick_every_second: rows and columns 0, 2, 4, .. (v_x is even)
                if (ghalf != nullptr && (r_out & 1) == 0) ghalf[(size_t)(r_out >> 1) * a.half_pitch + (v_x >> 1)] = pend[i].x;
            }
        }
    };

    issue(0);                                  // the first rows are on their way while the weights are surveyed
    psx_interp_survey(NP, t, NT, x0, min(TW, a.W - x0), Y0, Y1 - Y0, (LDS_AS v4f*)s_tab[0], (LDS_AS v4f*)s_tab[1], s_mask);
    __syncthreads();
    const unsigned hmask = a.force_literal ? ~0u : __builtin_amdgcn_readfirstlane(s_mask[0]);
    const unsigned vmask = a.force_literal ? ~0u : __builtin_amdgcn_readfirstlane(s_mask[1]);
    for (int k = 0; k < nsteps; k++) {
        commit();
        flush(k - 1);
        if (!interior) { __syncthreads(); patch(); }
        __syncthreads();
        if (k + 1 < nsteps) issue(k + 1);

        // ---- horizontal: 4 columns of 2 rows ----
        {
            // the column index, opaque per step: the literal weights of the rare varying pairs are step invariant, and hoisted
            // out of the step loop they would sit in registers (4 per pair and side) for the whole kernel
            int h_cc = h_c0;
            asm volatile("" : "+v"(h_cc));
            v2f out[4];
            psx_hinterp2x4<NP, HALO>(h_src, (const LDS_AS v4f*)s_tab[0], hmask, a.g0, h_cc, out);
            const int slot = (k * BR + 2 * h_rp) & (RING - 1);       // even: slot + 1 is the pair's second row
            float* rp = &s_ring[slot * RS + 4 * h_quad];
            const v4f o0 = (v4f){out[0].x, out[1].x, out[2].x, out[3].x}, o1 = (v4f){out[0].y, out[1].y, out[2].y, out[3].y};
            *reinterpret_cast<v4f*>(rp) = o0;
            *reinterpret_cast<v4f*>(rp + RS) = o1;
            if (slot < MIRROR) *reinterpret_cast<v4f*>(rp + RING * RS) = o0;
            if (slot + 1 < MIRROR) *reinterpret_cast<v4f*>(rp + (RING + 1) * RS) = o1;
        }
        __syncthreads();

        // ---- vertical: 2 columns of 4 rows ----
        {
            const int rel0 = k * BR - 2 * RI + v_rg * 4;      // ring index of the first row of the window
            const int r_out0 = Y0 + rel0;
            if (r_out0 + 3 >= Y0 && r_out0 < Y1) {
                const LDS_AS float* vp = (const LDS_AS float*)&s_ring[(rel0 & (RING - 1)) * RS + 2 * v_pp];
                v2f o[4];
                psx_vinterp2x4<NP, RS>(vp, (const LDS_AS v4f*)s_tab[1], vmask, a.g0, r_out0, o);
                asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
                for (int i = 0; i < 4; i++) pend[i] = o[i];
            }
        }
    }
    flush(nsteps - 1);
}

template <int NP>
__global__ __launch_bounds__(NT, GeomI<NP>::WGPC) void k_blur_interp(InterpArgs<NP> a)
{
    using G = GeomI<NP>;
    __shared__ __attribute__((aligned(16))) float s_stage[(BR / 2) * G::SS];
    __shared__ __attribute__((aligned(16))) float s_ring[(G::RING + G::MIRROR) * G::RS];
    __shared__ __attribute__((aligned(16))) v4f s_tab[2][NP];   // [H / V][pair]: left weight, right weight, a + b, off (blur_interp.h)
    __shared__ unsigned s_mask[2];            // bit (pair * 2 + side): the weight varies over the strip's columns / the chunk's rows
    interp_body<NP>(a, psx_xcd_remap(blockIdx.x, gridDim.x), s_stage, s_ring, s_tab, s_mask);
}

// Two independent planes in one launch (the diagonal schedule: level l of octave o together with level l - (L - 3) of octave
// o + 1, as k_blur2 in pyramid.hip): the first na logical blocks belong to job a, the rest to job b.  The small octaves are
// chains of ~8 us launches on their own; riding along with the octave above hides them.  Two inlined copies, so that both
// argument blocks stay in scalar registers.
template <int NP>
__global__ __launch_bounds__(NT, GeomI<NP>::WGPC) void k_blur_interp2(InterpArgs<NP> a, InterpArgs<NP> b, int na)
{
    using G = GeomI<NP>;
    __shared__ __attribute__((aligned(16))) float s_stage[(BR / 2) * G::SS];
    __shared__ __attribute__((aligned(16))) float s_ring[(G::RING + G::MIRROR) * G::RS];
    __shared__ __attribute__((aligned(16))) v4f s_tab[2][NP];
    __shared__ unsigned s_mask[2];
    const int lid = psx_xcd_remap(blockIdx.x, gridDim.x);
    if (lid < na) interp_body<NP>(a, lid, s_stage, s_ring, s_tab, s_mask);
    else          interp_body<NP>(b, lid - na, s_stage, s_ring, s_tab, s_mask);
}

inline int device_cus()
{
    static const int n = [] { int d = 0, c = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256; return c; }();
    return n;
}

// steps per chunk as k_blur chooses them: 5 on planes that fill the chip, fewer on the small octaves (latency chains)
inline void interp_chunking(int W, int H, int RI, int& chunk_rows, int& nchunks)
{
    const int nstrips = (W + TW - 1) / TW;
    static const int steps = [] { const char* e = getenv("POPSIFT_INTERP_STEPS"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 64 ? v : 5; }();
    static const int minwg = [] { const char* e = getenv("POPSIFT_INTERP_MINWG"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 384; }();
    int S = steps;
    for (; S > 2; S--) {
        const int cr = S * BR - 2 * RI;
        if (cr >= BR && nstrips * ((H + cr - 1) / cr) >= minwg) break;
    }
    // the small octaves are latency chains: where the filter is narrow enough that ONE step still yields >= 12 rows, a plane
    // that does not fill a quarter of the chip with two-step chunks is cut into one-step chunks (POPSIFT_INTERP_ONESTEP=0: off)
    static const bool onestep = [] { const char* e = getenv("POPSIFT_INTERP_ONESTEP"); return !(e != nullptr && e[0] == '0'); }();
    if (onestep && S == 2 && BR - 2 * RI >= 12 && nstrips * ((H + (2 * BR - 2 * RI) - 1) / (2 * BR - 2 * RI)) < 256) S = 1;
    int cr = S * BR - 2 * RI;
    if (cr < 4) cr = 4;
    if (cr > H) cr = H;
    chunk_rows = cr;
    nchunks = (H + cr - 1) / cr;
}

template <int NP>
int fill_interp(InterpArgs<NP>& a, const PsxInterpJob& j)
{
    a.src = j.src; a.dst = j.dst; a.half_dst = j.half_dst;
    a.W = j.W; a.H = j.H; a.pitch = j.pitch; a.half_pitch = j.half_pitch;
    a.nstrips = (j.W + TW - 1) / TW;
    int nchunks;
    interp_chunking(j.W, j.H, GeomI<NP>::RI, a.chunk_rows, nchunks);
    const int npairs = (j.ispan - 1) / 2;
    static const int force = [] { const char* e = getenv("POPSIFT_INTERP_LITERAL"); return e != nullptr && e[0] == '1' ? 1 : 0; }();
    a.force_literal = force;
    a.g0 = j.fi[0];
    for (int p = 0; p < NP; p++) {
        const int offset = 2 * p + 1;
        // pairs beyond the table's span: weight 0 (fma(val, 0, out) == out for the finite values of a plane)
        const float u = p < npairs ? j.fi[offset] : 0.0f;
        a.mul[p] = p < npairs ? j.fi[offset + 1] : 0.0f;
        a.off[p] = offset + (1.0f - u);
    }
    return a.nstrips * nchunks;
}

template <int NP>
hipError_t launch_interp(const PsxInterpJob& j, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    InterpArgs<NP> a;
    const dim3 grid(fill_interp<NP>(a, j)), block(NT);
    if (ev0 != nullptr || ev1 != nullptr) hipExtLaunchKernelGGL((k_blur_interp<NP>), grid, block, 0, s, ev0, ev1, 0, a);
    else                                  hipLaunchKernelGGL((k_blur_interp<NP>), grid, block, 0, s, a);
    return hipGetLastError();
}

template <int NP>
hipError_t launch_interp2(const PsxInterpJob& ja, const PsxInterpJob& jb, hipStream_t s)
{
    InterpArgs<NP> a, b;
    const int na = fill_interp<NP>(a, ja), nb = fill_interp<NP>(b, jb);
    hipLaunchKernelGGL((k_blur_interp2<NP>), dim3(na + nb), dim3(NT), 0, s, a, b, na);
    return hipGetLastError();
}

} // namespace

// fi: the level's row of the interpolated table (i_filter: [0] centre, [2p+1] = u, [2p+2] = a + b), ispan: its odd span.
// The loop of the reference runs offset = 1, 3, .. <= ispan; the table is zero from index ispan on, so the pairs that count
// are offset <= ispan - 2.  Returns hipErrorNotSupported beyond 8 pairs (the caller keeps the per-level kernels).
bool psx_blur_interp_ok(int ispan)
{
    static const bool off = [] { const char* e = getenv("POPSIFT_INTERP_FUSED"); return e != nullptr && e[0] == '0'; }();
    return !off && (ispan - 1) / 2 <= 8;
}

hipError_t psx_launch_blur_interp(const PsxInterpJob& j, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    const int np = (j.ispan - 1) / 2;
    if (np <= 3) return launch_interp<3>(j, s, ev0, ev1);
    if (np <= 4) return launch_interp<4>(j, s, ev0, ev1);
    if (np <= 5) return launch_interp<5>(j, s, ev0, ev1);
    if (np <= 6) return launch_interp<6>(j, s, ev0, ev1);
    if (np <= 7) return launch_interp<7>(j, s, ev0, ev1);
    if (np <= 8) return launch_interp<8>(j, s, ev0, ev1);
    return hipErrorNotSupported;
}

// workgroups a level of a W x H plane is launched with
int psx_blur_interp_grid(int W, int H, int ispan)
{
    const int np = (ispan - 1) / 2, npt = np <= 3 ? 3 : np;      // the instantiation's pair count sets the chunk length
    int cr, nchunks;
    interp_chunking(W, H, 2 * npt, cr, nchunks);
    return ((W + TW - 1) / TW) * nchunks;
}

// pairing rule of the diagonal schedule: two levels share ONE launch when both grids fit one round of resident workgroups
bool psx_blur_interp_pair_ok(int W1, int H1, int ispan1, int W2, int H2, int ispan2)
{
    const int isp = ispan1 > ispan2 ? ispan1 : ispan2;
    const int wgpc = (isp - 1) / 2 >= 5 && PSX_INTERP_WGPC_BIG > 0 ? PSX_INTERP_WGPC_BIG : 4;
    // POPSIFT_INTERP_PAIR_ROUNDS (measurement switch): percent of one round the two grids together may take (150 / 200, i.e. octave 0's
    // levels 4 and 5 sharing a launch with octave 1's levels 1 and 2: pyramid 0.315 -> 0.319 / 0.322 ms, profiles/r06_interp_pair_rounds.txt)
    static const int pct = [] { const char* e = getenv("POPSIFT_INTERP_PAIR_ROUNDS"); const int v = e ? atoi(e) : 0; return v >= 50 && v <= 1000 ? v : 100; }();
    return (psx_blur_interp_grid(W1, H1, isp) + psx_blur_interp_grid(W2, H2, isp)) * 100 <= wgpc * device_cus() * pct;
}

// two independent levels in one launch; the kernel is instantiated for the larger pair count (zero-weight pairs for the other)
hipError_t psx_launch_blur_interp2(const PsxInterpJob& a, const PsxInterpJob& b, hipStream_t s)
{
    const int np = ((a.ispan > b.ispan ? a.ispan : b.ispan) - 1) / 2;
    if (np <= 3) return launch_interp2<3>(a, b, s);
    if (np <= 4) return launch_interp2<4>(a, b, s);
    if (np <= 5) return launch_interp2<5>(a, b, s);
    if (np <= 6) return launch_interp2<6>(a, b, s);
    if (np <= 7) return launch_interp2<7>(a, b, s);
    if (np <= 8) return launch_interp2<8>(a, b, s);
    return hipErrorNotSupported;
}
