// blur_tile_core.h -- several consecutive blur levels of one octave per launch, on LDS-resident tiles.
//
// Why: octaves >= 1 of a 1080p frame hold a third of a frame's pixels but were a chain of 13
// dependent ~5-10 us launches (one per level, the reference's own decomposition: s_pyramid_build.cu:547-575 runs
// horiz + vert per level and per octave).  A launch costs ~5 us whatever its workgroups do, so the chain gets shorter
// only with FEWER launches: here one workgroup owns a TX x TY tile of the octave, loads the previous level's plane
// around it once (halo = the sum of the radii of the fused levels), and runs level after level out of LDS --
// H pass (P -> Q), barrier, V pass (Q -> P, and the tile's core rows to HBM), barrier -- so levels 1..L-3 of an
// octave (and the decimation into the next octave) are ONE launch, levels L-2..L-1 another.
//
// Bit-exactness: every pixel is produced by hfilter8_km / vfilter2x4_km (blur_arith.h), the functions of the marching
// kernel, on the same operands in the same order; texture clamping is "the cell outside the plane holds a copy of the
// clamped cell", established by the loader and re-established after every level (tile_fixup, edge tiles only).  Cells
// outside the region a level needs are computed from whatever lies there (always finite plane values: the loader fills
// the whole of P) and never reach a needed cell.
//
// The phase functions take the thread index as an argument and touch nothing but their arguments, so the SAME code
// runs on the CPU, one "thread" after the other and one phase after the other (tests/cpp/tile_emu.cpp, built with
// PSX_TILE_EMU; tests/test_tile_emu_cpu.py compares its planes with the CPU checker's bit for bit, without a GPU).
#pragma once

#include "blur_arith.h"

#include <stddef.h>

#define PSX_TILE_MAXLEV 6
#define PSX_TILE_NRAD   5
#define PSX_TILE_SQ     100          // row stride of Q (floats): <= 96 columns, 16-byte aligned rows, == 4 (mod 8)
#define PSX_TILE_LDS_MAX (160 * 1024)

#ifdef PSX_TILE_EMU
#define PSX_HOSTDEV static inline
#else
#define PSX_HOSTDEV __host__ __device__ inline
#endif

PSX_HOSTDEV constexpr int psx_tile_radius(int i) { return i == 0 ? 5 : i == 1 ? 7 : i == 2 ? 8 : i == 3 ? 10 : 13; }

struct PsxTileLevel {                // one fused level, LDS coordinates of P
    int rsel;                        // psx_tile_radius(rsel) >= the level's radius (zero taps beyond it)
    int vc0, vc1;                    // columns the level produces (multiples of 8)
    int vr0, vr1;                    // rows the level produces (multiples of 4)
    int pad[3];
};

struct PsxTileHdr {                  // what every phase reads: copied into registers once per workgroup
    const float* src;                // plane of the level in front of the first fused one
    float* half_dst;                 // level 0 of the next octave (get_by_2_pick_every_second), or nullptr
    int W, H, pitch, half_pitch;
    int nlev, half_lev;              // half_lev: index (within the job) of the level that also feeds half_dst; -1: none
    int TX, TY;                      // tile core (columns a multiple of 8, rows a multiple of 4)
    int OX, OY;                      // LDS coordinates of the core's first pixel (multiples of 4)
    int NC, NR;                      // P: NR rows of NC columns ...
    int SP;                          // ... SP floats apart (an odd number of 16-byte chunks)
    int QOFF;                        // P column of Q's column 0
    int lpr_shift;                   // loader: 1 << lpr_shift lanes per row of P (NC / 4 chunks)
    int tiles_x, tiles_y;
    int block0;                      // first logical workgroup of this job in its launch
};

struct PsxTileJob {                  // device resident, read with scalar loads
    PsxTileHdr h;
    float* dst[PSX_TILE_MAXLEV];     // planes of the fused levels
    PsxTileLevel lev[PSX_TILE_MAXLEV];
    PsxTaps taps[PSX_TILE_MAXLEV];
};

// ---- host: the plan of one job ------------------------------------------------------------------------------------------
// radii[l] = span - 1 of fused level l.  Returns the LDS bytes the job needs, or 0 when the job is outside what the kernel
// is built for (a radius above 13, more than 96 Q columns, more than 160 KB of LDS): the caller keeps the launch-per-level
// schedule then.
static inline size_t psx_tile_plan_job(PsxTileJob& job, int W, int H, int pitch, int nlev, const int* radii, int TX, int TY)
{
    PsxTileHdr& j = job.h;
    if (nlev < 1 || nlev > PSX_TILE_MAXLEV || (TX & 7) || (TY & 3) || TX < 8 || TY < 4) return 0;
    int rt[PSX_TILE_MAXLEV], halo[PSX_TILE_MAXLEV], hafter[PSX_TILE_MAXLEV];
    for (int l = 0; l < nlev; l++) {
        int i = 0;
        while (i < PSX_TILE_NRAD && psx_tile_radius(i) < radii[l]) i++;
        if (i == PSX_TILE_NRAD) return 0;
        job.lev[l].rsel = i;
        rt[l] = psx_tile_radius(i);
        halo[l] = (rt[l] + 3) & ~3;
    }
    hafter[nlev - 1] = 0;
    for (int l = nlev - 2; l >= 0; l--) hafter[l] = hafter[l + 1] + rt[l + 1];
    int ox = 0, oy = 0, qw = 0;
    for (int l = 0; l < nlev; l++) {
        const int hq = (hafter[l] + 3) & ~3;             // region = core +- hq: whole 8-column segments, whole 4-row groups
        if (hq + halo[l] > ox) ox = hq + halo[l];
        if (hq + rt[l] > oy) oy = hq + rt[l];
        if (hq > qw) qw = hq;
    }
    oy = (oy + 3) & ~3;
    j.W = W; j.H = H; j.pitch = pitch;
    j.nlev = nlev; j.TX = TX; j.TY = TY; j.OX = ox; j.OY = oy;
    j.NC = TX + 2 * ox; j.NR = TY + 2 * oy;
    j.SP = 4 * ((j.NC / 4) | 1);
    j.QOFF = ox - qw;
    if (TX + 2 * qw > PSX_TILE_SQ - 4) return 0;
    int sh = 3;
    while ((4 << sh) < j.NC) sh++;
    if (sh > 6) return 0;
    j.lpr_shift = sh;
    for (int l = 0; l < nlev; l++) {
        const int hq = (hafter[l] + 3) & ~3;
        job.lev[l].vc0 = ox - hq; job.lev[l].vc1 = ox + TX + hq;
        job.lev[l].vr0 = oy - hq; job.lev[l].vr1 = oy + TY + hq;
        job.lev[l].pad[0] = job.lev[l].pad[1] = job.lev[l].pad[2] = 0;
    }
    j.tiles_x = (W + TX - 1) / TX; j.tiles_y = (H + TY - 1) / TY;
    const size_t bytes = sizeof(float) * ((size_t)j.NR * j.SP + (size_t)j.NR * PSX_TILE_SQ);
    return bytes <= PSX_TILE_LDS_MAX ? bytes : 0;
}

// ---- memory access through macros: LDS address space + volatile on the device (one ds_read_b128 / ds_read_b64 per
// access, as in blur_body), plain pointers on the host ----
#ifdef PSX_TILE_EMU
#define TL_RD128(p) (*reinterpret_cast<const v4f*>(p))
#define TL_RD64(p)  (*reinterpret_cast<const v2f*>(p))
#define TL_UNIFORM(x) (x)
#define TL_GLOBAL
static inline void tl_store64(float* p, v2f v) { p[0] = v.x; p[1] = v.y; }
static inline void tl_store32(float* p, float v) { p[0] = v; }
#else
#define TL_LDS __attribute__((address_space(3)))
#define TL_RD128(p) (*reinterpret_cast<const volatile TL_LDS v4f*>((const TL_LDS float*)(p)))
#define TL_RD64(p)  (*reinterpret_cast<const volatile TL_LDS v2f*>((const TL_LDS float*)(p)))
#define TL_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// plane pointers come out of a job record in memory and would be generic (flat_load / flat_store) to the compiler
#define TL_GLOBAL __attribute__((address_space(1)))
// system-scope (write-through) stores, as the marching kernel's: nothing is left dirty in the XCD's L2 at kernel end
PSX_DEV void tl_store64(TL_GLOBAL float* p, v2f v)
{
    unsigned long long bits; __builtin_memcpy(&bits, &v, 8);
    __hip_atomic_store(reinterpret_cast<TL_GLOBAL unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
PSX_DEV void tl_store32(TL_GLOBAL float* p, float v) { *p = v; }
#endif

PSX_DEV int tl_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- loader: ALL of P (every cell is read by some level, needed or not: it must hold a finite value), coordinates
// clamped to the plane.  Thread -> (row = tid >> lpr_shift (+ k * rows per pass), 16-byte chunk); every load of the
// thread is issued before the first LDS store (the stores wait for the data, the loads do not wait for each other).
template <int NT>
PSX_DEV void tile_load(const PsxTileHdr& j, const int X0, const int Y0, float* P, const int tid)
{
    constexpr int MAXIT = NT >= 1024 ? 8 : 16;           // NR <= 8 * 32 / 16 * 8 rows even at 64 lanes per row
    const int lpr = 1 << j.lpr_shift;
    const int c4 = tid & (lpr - 1);
    const int rows_per_pass = NT >> j.lpr_shift;
    const int row0 = tid >> j.lpr_shift;
    const int x = X0 + 4 * c4 - j.OX;                    // a multiple of 4
    const bool col_on = 4 * c4 < j.NC;
    const bool vec = x >= 0 && x + 3 < j.W;
    const int xa = tl_clamp(x, 0, j.W - 1), xb = tl_clamp(x + 1, 0, j.W - 1), xc = tl_clamp(x + 2, 0, j.W - 1), xd = tl_clamp(x + 3, 0, j.W - 1);
    const TL_GLOBAL float* const src = (const TL_GLOBAL float*)j.src;
    v4f pre[MAXIT];
#pragma unroll
    for (int k = 0; k < MAXIT; k++) {
        const int row = row0 + k * rows_per_pass;
        if (col_on && row < j.NR) {
            const int y = tl_clamp(Y0 + row - j.OY, 0, j.H - 1);
            // uniform plane base + one 32-bit offset per load: an address costs one register, not two
            const unsigned ro = (unsigned)(y * j.pitch);
            if (vec) pre[k] = *reinterpret_cast<const TL_GLOBAL v4f*>(src + (ro + (unsigned)x));
            else     pre[k] = (v4f){src[ro + (unsigned)xa], src[ro + (unsigned)xb], src[ro + (unsigned)xc], src[ro + (unsigned)xd]};
        }
    }
#pragma unroll
    for (int k = 0; k < MAXIT; k++) {
        const int row = row0 + k * rows_per_pass;
        if (col_on && row < j.NR) *reinterpret_cast<v4f*>(&P[row * j.SP + 4 * c4]) = pre[k];
    }
}

// ---- horizontal pass of one level: Q[row][c] = H(P[row][c - R .. c + R]) for the rows the vertical pass reads and the
// columns it produces.  A wave owns blocks of 8 rows x 64 columns; lane -> (row, 8-column segment) as in blur_body: the
// ds_read_b128 lane groups hold {2 adjacent rows} x {8 segments} (P's row stride is an odd number of chunks) and the
// ds_write_b128 groups pair rows an odd distance apart (Q's row stride == 4 mod 8 dwords): conflict free. ----
template <int RT, int NT>
PSX_DEV void tile_hpass(const PsxTileHdr& j, const PsxTileLevel& lv, const PsxTaps& tp, const float* P, float* Q, const int tid)
{
    constexpr int HALO = (RT + 3) & ~3;
    constexpr int NW = NT / 64;
    const int lane = tid & 63;
    const int wave = TL_UNIFORM(tid >> 6);
    const int blk = (lane & 31) >> 2;
    const int rq = (0x21120330 >> (4 * blk)) & 3;        // rows          0 3 3 0 2 1 1 2
    const int sh = (0xCC >> blk) & 1;                    // segments 4-7? 0 0 1 1 0 0 1 1
    const int l_row = ((lane >> 5) & 1) * 4 + rq;
    const int l_seg = sh * 4 + (lane & 3);
    const int hr0 = lv.vr0 - RT, hr1 = lv.vr1 + RT;
    const int nrb = (hr1 - hr0 + 7) >> 3;
    for (int cb = lv.vc0; cb < lv.vc1; cb += 64) {
        const int col0 = cb + l_seg * 8;
        for (int rb = wave; rb < nrb; rb += NW) {
            const int row = hr0 + rb * 8 + l_row;
            if (row < hr1 && col0 < lv.vc1) {
                const float* src = P + row * j.SP + col0 - HALO;
                float win[8 + 2 * HALO];
#pragma unroll
                for (int q = 0; q < (8 + 2 * HALO) / 4; q++) {
                    const v4f v = TL_RD128(src + 4 * q);
                    win[4 * q + 0] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
                }
                float out[8];
                hfilter8_km<RT, HALO, false>(win, tp, out);
                float* dp = Q + row * PSX_TILE_SQ + (col0 - j.QOFF);
                *reinterpret_cast<v4f*>(dp)     = (v4f){out[0], out[1], out[2], out[3]};
                *reinterpret_cast<v4f*>(dp + 4) = (v4f){out[4], out[5], out[6], out[7]};
            }
        }
    }
}

// ---- vertical pass of one level: P[r][c] = V(Q[r - R .. r + R][c]); the cells of the tile's core that lie inside the
// plane also go to HBM (and, for level L-3, every second one to the next octave).  Thread = (2 adjacent columns, 4 rows);
// a half wave reads 64 contiguous dwords per ds_read_b64. ----
template <int RT, int NT>
PSX_DEV void tile_vpass(const PsxTileHdr& j, const PsxTileLevel& lv, const PsxTaps& tp, const float* Q, float* P,
                        float* gdst_, float* ghalf_, const int X0, const int Y0, const bool keep, const int tid)
{
    constexpr int VWIN = 4 + 2 * RT;
    constexpr int NW = NT / 64;
    TL_GLOBAL float* const gdst = (TL_GLOBAL float*)gdst_;
    TL_GLOBAL float* const ghalf = (TL_GLOBAL float*)ghalf_;
    const int lane = tid & 63;
    const int wave = TL_UNIFORM(tid >> 6);
    const int pp = lane & 31, rg = lane >> 5;
    const int nrb = (lv.vr1 - lv.vr0 + 7) >> 3;
    for (int cb = lv.vc0; cb < lv.vc1; cb += 64) {
        const int c = cb + 2 * pp;
        const int x = X0 + c - j.OX;
        const bool core_c = c >= j.OX && c < j.OX + j.TX && x < j.W;
        const bool pair = x + 1 < j.W;
        for (int rb = wave; rb < nrb; rb += NW) {
            const int r0 = lv.vr0 + rb * 8 + rg * 4;
            if (r0 < lv.vr1 && c < lv.vc1) {
                const float* vp = Q + (r0 - RT) * PSX_TILE_SQ + (c - j.QOFF);
                v2f v[VWIN];
#pragma unroll
                for (int q = 0; q < VWIN; q++) v[q] = TL_RD64(vp + q * PSX_TILE_SQ);
                v2f o[4];
                vfilter2x4_km<RT>(v, tp, o);
#ifndef PSX_TILE_EMU
                asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#endif
                if (keep) {
#pragma unroll
                    for (int i = 0; i < 4; i++) *reinterpret_cast<v2f*>(&P[(r0 + i) * j.SP + c]) = o[i];
                }
                if (core_c) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int r = r0 + i;
                        const int y = Y0 + r - j.OY;
                        if (r >= j.OY && r < j.OY + j.TY && y < j.H) {
                            TL_GLOBAL float* di = gdst + (size_t)y * j.pitch + x;
                            if (pair) tl_store64(di, o[i]); else tl_store32(di, o[i].x);
                            // get_by_2_pick_every_second (s_pyramid_build.cu:50-71): rows and columns 0, 2, 4, .. (x is even)
                            if (ghalf != nullptr && (y & 1) == 0) tl_store32(ghalf + (size_t)(y >> 1) * j.half_pitch + (x >> 1), o[i].x);
                        }
                    }
                }
            }
        }
    }
}

// ---- edge tiles, between two levels: the cells of the level's region that lie outside the plane take the value of the
// clamped cell (texture clamp addressing of the next level's reads) ----
template <int NT>
PSX_DEV void tile_fixup(const PsxTileHdr& j, const PsxTileLevel& lv, float* P, const int X0, const int Y0, const int tid)
{
    const int ncol = lv.vc1 - lv.vc0;                    // a multiple of 8; <= 96
    // thread -> (column, first row): 128 column slots per row pass keeps the index arithmetic to shifts
    const int c = lv.vc0 + (tid & 127);
    const int x = X0 + c - j.OX;
    const int xcl = tl_clamp(x, 0, j.W - 1);
    const bool c_on = (tid & 127) < ncol;
    for (int r = lv.vr0 + (tid >> 7); r < lv.vr1; r += NT >> 7) {
        const int y = Y0 + r - j.OY;
        const int ycl = tl_clamp(y, 0, j.H - 1);
        if (c_on && (x != xcl || y != ycl)) P[r * j.SP + c] = P[(j.OY + ycl - Y0) * j.SP + (j.OX + xcl - X0)];
    }
}
