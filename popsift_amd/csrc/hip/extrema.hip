// extrema.hip -- 3x3x3 DoG extrema scan with sub-pixel refinement for gfx950.
//
// Replaces (behaviour) find_extrema_in_dog<HEIGHT,mode> and its helpers
// (s_extrema.cu:22-558, s_solve.h:25-86) plus make_dog (s_pyramid_build.cu:74-92):
// the DoG planes are never written to HBM.  A 256-thread workgroup stages a 64x16 tile (+1 px
// halo) of all L-1 DoG levels in LDS, computed as G[l+1]-G[l] while loading the L Gaussian
// planes once (24 B/pixel for L=6).  Candidates that pass the contrast pre-test and the strict
// 26-neighbour test are appended to a per-octave candidate list; k_refine (one launch for all octaves)
// runs the refinement (<= 5 Newton steps, closed-form 3x3 solve) one candidate per lane, reading DoG
// values on the fly from the Gaussian planes (a single subtraction, bit-identical to a materialised
// DoG).  Survivors are compacted with a 64-bit wave ballot and one atomicAdd per wave (wave64
// re-design of extrema_count, s_extrema.cu:22-44).
#include "psx_internal.h"

namespace {

constexpr int ETW = 64, ETH = 16;          // tile
constexpr int TWP = ETW + 2, THP = ETH + 2;
constexpr int QCAP = 2 * (ETW / 2) * (ETH / 2);   // most strict 3x3 extrema (maxima + minima) one DoG level of a tile can hold
constexpr int NT = 256;

__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// DoG "texture" read: clamp in x,y (sift_octave.cu:233-236), layer index clamped
__device__ __forceinline__ float rdog(const PsxOctave& oc, int NL, int x, int y, int z)
{
    x = psx_clampi(x, 0, oc.w - 1);
    y = psx_clampi(y, 0, oc.h - 1);
    z = psx_clampi(z, 0, NL - 1);
    const float* p = oc.data + (size_t)z * oc.plane + (size_t)y * oc.pitch + x;
    return p[oc.plane] - p[0];
}

// Second-order Taylor model of the DoG around a sample: gradient g and the symmetric Hessian H in (x, y, s).
struct Taylor
{
    float gx, gy, gs;             // central differences
    float hxx, hyy, hss;          // second differences
    float hxy, hxs, hys;          // mixed differences
};

// Newton offset  off = -H^-1 g  through the adjugate of the symmetric H (what solve() of s_solve.h:25-86 computes).
// The parenthesisation is part of the parity contract (every product rounded, sums left to right; the file is built
// with -ffp-contract=off): cofactor = (-(p*p')) + (q*q'), det = ((hxx*Cxx) + (hxy*Cxy)) + (hxs*Cxs).
// Returns false for a singular H.
__device__ __forceinline__ bool newton_offset(const Taylor& t, float off[3])
{
    const float Cxx = (-(t.hys * t.hys)) + (t.hyy * t.hss);
    const float Cxy = (-(t.hxy * t.hss)) + (t.hys * t.hxs);
    const float Cxs = (-(t.hyy * t.hxs)) + (t.hxy * t.hys);
    const float Cyy = (-(t.hxs * t.hxs)) + (t.hxx * t.hss);
    const float Cys = (-(t.hxx * t.hys)) + (t.hxy * t.hxs);
    const float Css = (-(t.hxy * t.hxy)) + (t.hxx * t.hyy);

    const float det = ((t.hxx * Cxx) + (t.hxy * Cxy)) + (t.hxs * Cxs);
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    const float Ixx = Cxx * inv, Ixy = Cxy * inv, Ixs = Cxs * inv;
    const float Iyy = Cyy * inv, Iys = Cys * inv, Iss = Css * inv;

    const float rx = -t.gx, ry = -t.gy, rs = -t.gs;
    off[0] = ((0.0f + Ixx * rx) + Ixy * ry) + Ixs * rs;
    off[1] = ((0.0f + Ixy * rx) + Iyy * ry) + Iys * rs;
    off[2] = ((0.0f + Ixs * rx) + Iys * ry) + Iss * rs;
    return true;
}

// ModeFunctions<mode>::refine, s_extrema.cu:155-284
template <int MODE>
__device__ __forceinline__ int refine_step(const float d[3], int n[3], int width, int height, int maxlevel, bool last_it)
{
    if (MODE == PSX_MODE_OPENCV) {
        const float tx = fabsf(d[0]), ty = fabsf(d[1]), tz = fabsf(d[2]);
        if (tx < 0.5f && ty < 0.5f && tz < 0.5f) return 1;
        n[0] += (int)roundf(d[0]);
        n[1] += (int)roundf(d[1]);
        n[2] += (int)roundf(d[2]);
        return (n[0] < 5 || n[0] >= width - 5 || n[1] < 5 || n[1] >= height - 5 ||
                n[2] < 1 || n[2] > maxlevel - 2) ? -1 : 0;
    }
    if (last_it) return 0;
    const int tx = ((d[0] >= 0.6f && n[0] < width - 2) ? 1 : 0) + ((d[0] <= -0.6f && n[0] > 1) ? -1 : 0);
    const int ty = ((d[1] >= 0.6f && n[1] < height - 2) ? 1 : 0) + ((d[1] <= -0.6f && n[1] > 1) ? -1 : 0);
    int tz = 0;
    if (MODE == PSX_MODE_POPSIFT)
        tz = ((d[2] >= 0.6f && n[2] < maxlevel - 1) ? 1 : 0) + ((d[2] <= -0.6f && n[2] > 1) ? -1 : 0);
    if (tx == 0 && ty == 0 && tz == 0) return 1;
    n[0] += tx; n[1] += ty; n[2] += tz;
    return 0;
}

// DoG access for the refinement: the staged LDS tile when (x,y) +- 1 lies inside it (always true
// for the first Newton step), the Gaussian planes in HBM otherwise.  Both give the same bits:
// the tile holds G[z+1]-G[z] of the clamped coordinates.
struct DogView {
    const PsxOctave& oc;
    const float* sD;     // [NL][THP][TWP]
    int NL, tx0, ty0;
    __device__ __forceinline__ float at(int x, int y, int z, bool in_tile) const
    {
        if (in_tile) {
            z = psx_clampi(z, 0, NL - 1);
            return sD[(z * THP + (y - ty0 + 1)) * TWP + (x - tx0 + 1)];
        }
        return rdog(oc, NL, x, y, z);
    }
};

// find_extrema_in_dog_sub after the extremum test, s_extrema.cu:341-503
template <int MODE>
__device__ bool refine(const PsxParams* P, const DogView& dv, int octave, int x, int y, int level,
                       float val, psx_iext& ec)
{
    const PsxOctave& oc = dv.oc;
    const int width = oc.w, height = oc.h;
    const int NL = P->L - 1;
    const int maxlevel = P->L - 1;
    const float thr = P->threshold;

    Taylor t;
    float d[3] = {0.0f, 0.0f, 0.0f};
    const float v = val;
    int n[3] = {x, y, level};
    int iter = 0;
    constexpr int MAX_ITERATIONS = 5;
    do {
        iter++;
        // the 19 DoG samples of the 3x3x3 cross / edge stencil around n (corners are not needed); x,y +- 1 lie
        // inside the haloed tile [tx0-1, tx0+ETW] x [ty0-1, ty0+ETH] when n is inside the tile
        const bool in_tile = (n[0] >= dv.tx0 && n[0] < dv.tx0 + ETW && n[1] >= dv.ty0 && n[1] < dv.ty0 + ETH);
        auto S = [&](int dx, int dy, int ds) { return dv.at(n[0] + dx, n[1] + dy, n[2] + ds, in_tile); };
        const float c  = S(0, 0, 0);
        const float xp = S(+1, 0, 0), xm = S(-1, 0, 0);
        const float yp = S(0, +1, 0), ym = S(0, -1, 0);
        const float sp = S(0, 0, +1), sm = S(0, 0, -1);
        t.gx = 0.5f * (xp - xm);
        t.gy = 0.5f * (yp - ym);
        t.gs = 0.5f * (sp - sm);
        t.hxx = xp + xm - 2.0f * c;
        t.hyy = yp + ym - 2.0f * c;
        t.hss = sp + sm - 2.0f * c;
        // mixed terms: (++) + (--) - (-+) - (+-), summed in that order
        t.hxy = 0.25f * (S(+1, +1, 0) + S(-1, -1, 0) - S(-1, +1, 0) - S(+1, -1, 0));
        t.hxs = 0.25f * (S(+1, 0, +1) + S(-1, 0, -1) - S(-1, 0, +1) - S(+1, 0, -1));
        t.hys = 0.25f * (S(0, +1, +1) + S(0, -1, -1) - S(0, +1, -1) - S(0, -1, +1));

        if (!newton_offset(t, d)) { d[0] = d[1] = d[2] = 0.0f; break; }

        const int retval = refine_step<MODE>(d, n, width, height, maxlevel, iter == MAX_ITERATIONS);
        if (retval == -1) return false;
        else if (retval == 1) break;
    } while (iter < MAX_ITERATIONS);

    if (MODE == PSX_MODE_OPENCV && iter >= MAX_ITERATIONS) return false;
    if (MODE == PSX_MODE_POPSIFT || MODE == PSX_MODE_VLFEAT)
        if (d[0] >= 1.5f || d[1] >= 1.5f || d[2] >= 1.5f) return false;

    const float xn = n[0] + d[0];
    const float yn = n[1] + d[1];
    const float sn = n[2] + d[2];

    if (MODE != PSX_MODE_OPENCV) {
        if (xn < 0.0f || xn > width - 1.0f || yn < 0.0f || yn > height - 1.0f ||
            sn < 0.0f || sn > maxlevel) return false;
    }

    const float contr   = v + 0.5f * (t.gx * d[0] + t.gy * d[1] + t.gs * d[2]);
    const float tr      = t.hxx + t.hyy;
    const float det     = t.hxx * t.hyy - t.hxy * t.hxy;
    const float edgeval = tr * tr / det;

    if (det <= 0.0f) return false;
    if (fabsf(contr) < 2.0f * thr) return false;
    if (edgeval >= (P->edge_limit + 1.0f) * (P->edge_limit + 1.0f) / P->edge_limit) return false;

    ec.xpos  = xn;
    ec.ypos  = yn;
    ec.lpos  = (int)roundf(sn);
    ec.sigma = P->sigma0 * powf(P->sigma_k, sn);
    ec.cell  = (int)(floorf(yn / P->h_grid_div[octave]) * P->grid_size + floorf(xn / P->w_grid_div[octave]));
    ec.ignore = 0;
    return true;
}

#ifdef PSX_PHASE_TIMING
__device__ long long* g_dbg = nullptr;
#define STAMP(i) if (threadIdx.x == 0 && g_dbg) g_dbg[blockIdx.x * 8 + (i)] = clock64()
#else
#define STAMP(i)
#endif

template <int MODE>
__global__ __launch_bounds__(NT) void k_extrema(const PsxParams* __restrict__ P, PsxCounters* cnt, const PsxExtBatch b)
{
    STAMP(0);
    // the tiles of up to PSX_EXT_BATCH octaves share the launch (the small octaves: 36 .. 510 tiles each would be
    // three latency-bound launches); logical tile ids are octave-major
    const int glid = xcd_remap(blockIdx.x, gridDim.x);
    int bk = 0;
    while (bk + 1 < b.n && glid >= b.tile_end[bk]) bk++;
    const int octave = b.octave[bk], tiles_x = b.tiles_x[bk];
    const int lid = glid - (bk > 0 ? b.tile_end[bk - 1] : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PsxOctave oc = P->oct[octave];
    const int L = P->L, NL = L - 1, NZ = L - 3;
    float* sD = smem;                                    // [NL][THP][TWP]
    // two queues of packed (kind, z, y, x) codes.  A strict extremum of its own 3x3 neighbourhood cannot be
    // 8-adjacent to another one of the same kind, so a level of the 64x16 tile holds at most QCAP = 2 * 32 * 8.
    unsigned short* sQ1 = reinterpret_cast<unsigned short*>(smem + NL * THP * TWP);   // in-plane extrema, [NZ*QCAP]
    // The 26-neighbour extrema reuse the space of the queue they are filtered from (round 4): pass 2 works in rounds of NT
    // entries -- read, barrier, test, write -- and a survivor's slot is below the number of entries processed so far,
    // i.e. never an entry that is still to be read.  26.8 instead of 29.9 KB of LDS: six workgroups per CU instead of five.
    unsigned short* sQ  = sQ1;
    __shared__ int sCount, sCount1;

    const int t = threadIdx.x;
    const int tx0 = (lid % tiles_x) * ETW;
    const int ty0 = (lid / tiles_x) * ETH;
    if (t == 0) { sCount = 0; sCount1 = 0; }
    STAMP(1);

    // ---- stage DoG tile: every thread owns <= NE elements; all their loads are issued before the
    // first use so that one HBM round trip covers the whole tile ----
    constexpr int NE = (THP * TWP + NT - 1) / NT;
    // addresses as "uniform level base (scalar registers) + 32-bit byte offset of the pixel within its plane": the generic
    // form (64-bit pointer arithmetic per load through v_mad_u64_u32 / v_mul_lo_u32, flat loads) was a third of the
    // kernel's vector instructions
    typedef const __attribute__((address_space(1))) char* gchar_p;
    typedef const __attribute__((address_space(1))) float* gfloat_p;
    const gchar_p gbase = (gchar_p)oc.data;
    const size_t plane_b = (size_t)oc.plane * sizeof(float);
    const unsigned pitch_b = (unsigned)oc.pitch * 4u;                 // rows and pitch bytes < 2^24: v_mul_u32_u24
    if (L == 6) {
        // one scalar base per level (readfirstlane keeps "base + level * plane" from being re-associated into per-lane 64-bit
        // additions): every load is "scalar base + the pixel's 32-bit offset".  (A staging map in which a wave takes whole
        // rows -- scalar row clamp and offset, the two halo columns in a spare slot -- cut another 2 M vector instructions per
        // frame and measured 1.3 % SLOWER end to end, 3 % on sparse frames: not kept; the kernel is not bound by its
        // instruction count.)
        gchar_p lb[6];
#pragma unroll
        for (int l = 0; l < 6; l++) {
            const unsigned long long a = (unsigned long long)(uintptr_t)(gbase + l * plane_b);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            lb[l] = (gchar_p)(uintptr_t)(((unsigned long long)hi << 32) | lo);
        }
        float g[NE][6];
#pragma unroll
        for (int k = 0; k < NE; k++) {
            const int e = t + k * NT;
            const int ec = min(e, THP * TWP - 1);
            const int ry = ec / TWP, rx = ec - ry * TWP;
            const int gx = psx_clampi(tx0 - 1 + rx, 0, oc.w - 1);
            const int gy = psx_clampi(ty0 - 1 + ry, 0, oc.h - 1);
            const unsigned off = __umul24((unsigned)gy, pitch_b) + (unsigned)gx * 4u;
#pragma unroll
            for (int l = 0; l < 6; l++) g[k][l] = *(gfloat_p)(lb[l] + off);
        }
#pragma unroll
        for (int k = 0; k < NE; k++) {
            const int e = t + k * NT;
            if (e < THP * TWP) {
                const int ry = e / TWP, rx = e - ry * TWP;
#pragma unroll
                for (int l = 0; l < 5; l++) sD[(l * THP + ry) * TWP + rx] = g[k][l + 1] - g[k][l];
            }
        }
    } else {
        for (int e = t; e < THP * TWP; e += NT) {
            const int ry = e / TWP, rx = e - ry * TWP;
            const int gx = psx_clampi(tx0 - 1 + rx, 0, oc.w - 1);
            const int gy = psx_clampi(ty0 - 1 + ry, 0, oc.h - 1);
            const unsigned off = __umul24((unsigned)gy, pitch_b) + (unsigned)gx * 4u;
            float prev = *(gfloat_p)(gbase + off);
            for (int l = 0; l < NL; l++) {
                const float cur = *(gfloat_p)(gbase + (size_t)(l + 1) * plane_b + off);
                sD[(l * THP + ry) * TWP + rx] = cur - prev;
                prev = cur;
            }
        }
    }
    __syncthreads();
    STAMP(2);

    // ---- scan: contrast pre-test + strict 26-neighbour extremum (s_extrema.cu:56-120, 341-349) ----
    const float thr = P->threshold;
    const float thr1 = (MODE == PSX_MODE_OPENCV) ? floorf(thr)
                     : (MODE == PSX_MODE_VLFEAT) ? 0.8f * 2.0f * thr : 1.6f * thr;
    // Two passes.  Pass 1, register-tiled: a thread owns 4 vertically adjacent pixels of one column and per DoG
    // level pulls the 6x3 window of THAT level only (18 LDS reads in flight), reduces rows with max3/min3 and
    // queues the pixels that pass the contrast pre-test and are a strict maximum / minimum of their own 3x3
    // neighbourhood (a few per cent of the tile).  Pass 2, one queued pixel per lane: the 2 x 9 neighbours in the
    // levels below and above.  Together: the predicate of is_extremum (s_extrema.cu:56-120), strict against all
    // 26 neighbours.  (Testing all three levels for every pixel, as round 1 did, cost 150 VALU instructions and
    // 54 LDS reads per thread and level; the kernel's 14 M wave instructions were a tenth of a frame's total.)
    const int lx = t & (ETW - 1);
    const int ly0 = (t >> 6) * 4;
    const int x = tx0 + lx;
    for (int z = 1; z <= NZ; z++) {
        bool pre[4];
        bool anyp = false;
        // the four centre values in one go (pinned: otherwise each read is sunk behind its row's validity branch and
        // waited for on its own -- four LDS round trips in a row per level)
        float cv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) cv[r] = sD[(z * THP + ly0 + r + 1) * TWP + lx + 1];
        asm volatile("" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]));
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int y = ty0 + ly0 + r;
            bool valid = (x >= 1 && y >= 1 && x <= oc.w - 2 && y <= oc.h - 2);
            if (MODE == PSX_MODE_OPENCV) valid = valid && (x >= 5 && y >= 5 && x < oc.w - 5 && y < oc.h - 5);
            pre[r] = valid && fabsf(cv[r]) >= thr1;
            anyp = anyp || pre[r];
        }
        if (__ballot(anyp) == 0ull) continue;

        float mx[6], mn[6], lft[4], rgt[4], ctr[4];
#pragma unroll
        for (int rr = 0; rr < 6; rr++) {
            const float* q = &sD[(z * THP + ly0 + rr) * TWP + lx];
            const float a = q[0], b = q[1], c = q[2];
            if (rr >= 1 && rr <= 4) { lft[rr - 1] = a; rgt[rr - 1] = c; ctr[rr - 1] = b; }
            mx[rr] = fmaxf(fmaxf(a, b), c);
            mn[rr] = fminf(fminf(a, b), c);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = ctr[r];
            const float M = fmaxf(fmaxf(mx[r], mx[r + 2]), fmaxf(lft[r], rgt[r]));     // the centre is excluded from its own row
            const float N = fminf(fminf(mn[r], mn[r + 2]), fminf(lft[r], rgt[r]));
            const bool is_max = v > M, is_min = v < N;
            if (pre[r] && (is_max || is_min)) {
                const int slot = atomicAdd(&sCount1, 1);
                sQ1[slot] = (unsigned short)((is_max ? 0x8000 : 0) | (z << 10) | ((ly0 + r) << 6) | lx);
            }
        }
    }
    __syncthreads();
    {
        const int n1 = sCount1;
        for (int base = 0; base < n1; base += NT) {
            const int q = base + t;
            const int code = q < n1 ? (int)sQ1[q] : -1;
            __syncthreads();                             // the round's entries are in registers: their space may be overwritten
            if (code < 0) continue;
            const bool is_max = (code & 0x8000) != 0;
            const int z = (code >> 10) & 31, ly = (code >> 6) & 15, cx = code & 63;
            const float v = sD[(z * THP + ly + 1) * TWP + cx + 1];
            float M = -INFINITY, N = INFINITY;
#pragma unroll
            for (int dz = -1; dz <= 1; dz += 2) {
#pragma unroll
                for (int rr = 0; rr < 3; rr++) {
                    const float* p = &sD[((z + dz) * THP + ly + rr) * TWP + cx];
                    const float a = p[0], b = p[1], c = p[2];
                    M = fmaxf(M, fmaxf(fmaxf(a, b), c));
                    N = fminf(N, fminf(fminf(a, b), c));
                }
            }
            if (is_max ? (v > M) : (v < N)) {
                const int slot = atomicAdd(&sCount, 1);
                sQ[slot] = (unsigned short)(code & 0x7fff);
            }
        }
    }
    __syncthreads();
    STAMP(3);

    // ---- hand the candidates to k_refine: the refinement is a serial chain of up to 5 Newton steps per
    // candidate and a tile has ~1 candidate, so refining here kept the whole workgroup (and its LDS)
    // resident for one lane's latency.  Wave-aggregated append to the octave's candidate list; if the
    // list is full the candidate is refined in place (same result, just slower). ----
    const int nq = sCount;
    const int lane = t & (PSX_WAVE - 1);
    for (int base = 0; base < nq; base += NT) {
        const int q = base + t;
        const bool have = q < nq;
        int cx = 0, ly = 0, z = 0;
        if (have) {
            const int code = sQ[q];
            z = (code >> 10) & 31; ly = (code >> 6) & 15; cx = code & 63;
        }
        const unsigned long long mask = __ballot(have);
        if (mask == 0ull) continue;
        const int leader = __ffsll((long long)mask) - 1;
        int wbase = 0;
        const int sub = blockIdx.x & (PSX_CAND_SUB - 1);
        if (lane == leader) wbase = atomicAdd(&P->cand_ct[(octave * PSX_CAND_SUB + sub) * 32], __popcll(mask));
        wbase = __shfl(wbase, leader);
        if (have) {
            const int idx = wbase + __popcll(mask & ((1ull << lane) - 1ull));
            if (idx < P->cand_capacity) {
                P->cand[octave][(size_t)sub * P->cand_capacity + idx] =
                    ((unsigned long long)(unsigned)(ty0 + ly) << 32) | ((unsigned long long)(unsigned)z << 24) |
                    (unsigned)(tx0 + cx);
            } else {
                const DogView dv{oc, sD, NL, tx0, ty0};
                const float v = sD[(z * THP + ly + 1) * TWP + cx + 1];
                psx_iext ec;
                if (refine<MODE>(P, dv, octave, tx0 + cx, ty0 + ly, z, v, ec)) {
                    const int o = atomicAdd(&cnt->ext_ct[octave], 1);
                    if (o < P->max_extrema) { P->iext[octave][o] = ec; P->iext_off[octave][o] = o; }
                }
            }
        }
    }
    STAMP(4);
#ifdef PSX_PHASE_TIMING
    if (threadIdx.x == 0 && g_dbg) g_dbg[blockIdx.x * 8 + 5] = nq;
#endif
}

// Refinement of the queued candidates of ALL octaves in one launch: one workgroup per (octave, sub-list), one
// candidate per lane, DoG values read from the Gaussian planes.  Survivors are compacted with 64-bit wave ballots,
// a prefix over the workgroup's waves and one atomicAdd per workgroup and round (wave64 re-design of extrema_count,
// s_extrema.cu:22-44).
// REFINE_NT threads per (octave, sub-list).  What this kernel costs is a chain -- list entry -> DoG values -> up to five Newton
// steps of 19 dependent gathers each -> the slot in the octave's extremum array -- and the slot comes from ONE counter per
// octave: same-address atomics serialise in the L2 (~100 ns each), so one atomic per wave and round (190 for the bench
// frame's octave 0) set a floor however the candidates were spread: 1 / 2 / 4 / 8 / 16 one-wave blocks per list measured
// 0.094 / 0.090 / 0.085 / 0.108 / 0.117 ms for the extrema stage.  Eight waves per list cover a list of the bench frame in
// one round and take ONE slot range per workgroup and round: 64 / 128 / 256 / 512 / 1024 threads measured 0.087 / 0.088 /
// 0.081 / 0.082 / 0.085 ms, the kernel 21.6 -> 17.2 us (profiles/r05_refine_ab.txt).  What is left is the chain itself.
constexpr int REFINE_NT = 512;

template <int MODE>
__global__ __launch_bounds__(REFINE_NT) void k_refine(const PsxParams* __restrict__ P, PsxCounters* cnt)
{
    constexpr int NW = REFINE_NT / PSX_WAVE;
    __shared__ int s_cnt[NW], s_base;
    const int t = threadIdx.x, lane = t & (PSX_WAVE - 1), wave = t >> 6;
    const int list_id = blockIdx.x;
    const int o = list_id / PSX_CAND_SUB, sub = list_id % PSX_CAND_SUB;
    const int NL = P->L - 1;
    const int n = min(P->cand_ct[list_id * 32], P->cand_capacity);
    if (n <= 0) return;
    const PsxOctave oc = P->oct[o];
    const unsigned long long* list = P->cand[o] + (size_t)sub * P->cand_capacity;
    const DogView dv{oc, nullptr, NL, -(1 << 30), -(1 << 30)};      // never "in tile": global reads
    for (int e0 = 0; e0 < n; e0 += REFINE_NT) {
        const int e = e0 + t;
        bool ok = false;
        psx_iext ec;
        if (e < n) {
            const unsigned long long code = list[e];
            const int x = (int)(code & 0xffffffu), z = (int)((code >> 24) & 0xffu), y = (int)(code >> 32);
            const float v = rdog(oc, NL, x, y, z);
            ok = refine<MODE>(P, dv, o, x, y, z, v, ec);
        }
        const unsigned long long mask = __ballot(ok);
        if (lane == 0) s_cnt[wave] = __popcll(mask);
        __syncthreads();
        if (t == 0) {
            int tot = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) { const int c = s_cnt[w]; s_cnt[w] = tot; tot += c; }
            s_base = tot > 0 ? atomicAdd(&cnt->ext_ct[o], tot) : 0;
        }
        __syncthreads();
        if (ok) {
            const int idx = s_base + s_cnt[wave] + __popcll(mask & ((1ull << lane) - 1ull));
            if (idx < P->max_extrema) {
                P->iext[o][idx] = ec;
                P->iext_off[o][idx] = idx;
            }
        }
        __syncthreads();                                   // the next round rewrites s_cnt / s_base
    }
}

} // namespace

#ifdef PSX_PHASE_TIMING
extern "C" void psx_debug_set_buffer(long long* d) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &d, sizeof(d)); }
#endif

hipError_t psx_launch_extrema_batch(const PsxParams* d_params, const PsxParams& hp, PsxCounters* d_cnt,
                                    const int* octaves, int n, hipStream_t s)
{
    const int NL = hp.L - 1, NZ = hp.L - 3;
    if (NZ < 1 || n < 1) return hipSuccess;
    if (n > PSX_EXT_BATCH) return hipErrorInvalidValue;
    PsxExtBatch b;
    b.n = n;
    int tiles = 0;
    for (int k = 0; k < PSX_EXT_BATCH; k++) {
        const int o = octaves[k < n ? k : n - 1];
        const PsxOctave& oc = hp.oct[o];
        b.octave[k] = o;
        b.tiles_x[k] = (oc.w + ETW - 1) / ETW;
        if (k < n) tiles += b.tiles_x[k] * ((oc.h + ETH - 1) / ETH);
        b.tile_end[k] = tiles;
    }
    const size_t smem = sizeof(float) * (size_t)NL * THP * TWP + sizeof(unsigned short) * (size_t)NZ * QCAP;
    const dim3 grid(tiles), block(NT);
    // levels >= 7 need more than the 64 KiB of dynamic LDS a kernel gets by default (72..107 KB of the 160 KB per CU)
    if (smem > 64 * 1024) {
        hipError_t e = hipSuccess;
        switch (hp.sift_mode) {
        case PSX_MODE_VLFEAT: e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extrema<PSX_MODE_VLFEAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); break;
        case PSX_MODE_OPENCV: e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extrema<PSX_MODE_OPENCV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); break;
        default:              e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_extrema<PSX_MODE_POPSIFT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); break;
        }
        if (e != hipSuccess) return e;
    }
    switch (hp.sift_mode) {
    case PSX_MODE_VLFEAT: hipLaunchKernelGGL(k_extrema<PSX_MODE_VLFEAT>, grid, block, smem, s, d_params, d_cnt, b); break;
    case PSX_MODE_OPENCV: hipLaunchKernelGGL(k_extrema<PSX_MODE_OPENCV>, grid, block, smem, s, d_params, d_cnt, b); break;
    default:              hipLaunchKernelGGL(k_extrema<PSX_MODE_POPSIFT>, grid, block, smem, s, d_params, d_cnt, b); break;
    }
    return hipGetLastError();
}

hipError_t psx_launch_extrema(const PsxParams* d_params, const PsxParams& hp, PsxCounters* d_cnt,
                              int octave, hipStream_t s)
{
    return psx_launch_extrema_batch(d_params, hp, d_cnt, &octave, 1, s);
}

// tiles of one octave (what decides whether its scan is worth a launch of its own)
int psx_extrema_tiles(const PsxParams& hp, int octave)
{
    const PsxOctave& oc = hp.oct[octave];
    return ((oc.w + ETW - 1) / ETW) * ((oc.h + ETH - 1) / ETH);
}

hipError_t psx_launch_refine(const PsxParams* d_params, const PsxParams& hp, PsxCounters* d_cnt, hipStream_t s)
{
    if (hp.L - 3 < 1) return hipSuccess;
    const dim3 grid(hp.num_octaves * PSX_CAND_SUB), block(REFINE_NT);
    switch (hp.sift_mode) {
    case PSX_MODE_VLFEAT: hipLaunchKernelGGL(k_refine<PSX_MODE_VLFEAT>, grid, block, 0, s, d_params, d_cnt); break;
    case PSX_MODE_OPENCV: hipLaunchKernelGGL(k_refine<PSX_MODE_OPENCV>, grid, block, 0, s, d_params, d_cnt); break;
    default:              hipLaunchKernelGGL(k_refine<PSX_MODE_POPSIFT>, grid, block, 0, s, d_params, d_cnt); break;
    }
    return hipGetLastError();
}
