// pyramid.hip -- Gaussian scale-space construction for gfx950 (MI355X).
//
// Replaces (behaviour, not code) the reference's default build_pyramid branch
// (s_pyramid_build.cu:547-575):
//   normalizedSource::horiz + absoluteSource::vert (level 0 of octave 0)   -> k_upscale + k_blur<R,true>
//   absoluteSource::horiz + absoluteSource::vert   (levels 1..L-1)         -> k_blur
//   get_by_2_pick_every_second                                             -> fused into k_blur
//   make_dog                                                               -> not materialised;
//       the extrema kernel forms G[l+1]-G[l] on the fly (bit-identical single subtraction).
//
// Kernel shape (DESIGN.md "separable Gaussian"): one 256-thread workgroup owns a 64-column
// strip and marches down a chunk of rows in steps of 32 rows.  Per step: 32 input rows (+halo
// columns) are staged in LDS with row-coalesced float4 loads (the loads for step k+1 are issued
// before the arithmetic of step k), the horizontal filter writes into an LDS ring of 64
// H-filtered rows, the vertical filter reads 4+2R ring rows per thread to produce 4 output
// rows of two adjacent columns (packed f32 math).  Each plane is read once and written once (8 B/pixel algorithmic traffic);
// the intermediate plane of the reference ("intm", 16 B/pixel more) never exists.
//
// Arithmetic order is the reference's, written with explicit fmaf (the file is compiled with
// -ffp-contract=off), so planes are bit-identical to oracle/sift_oracle.c:
//   H (s_pyramid_build_aa.cu:17-50): centre, then pairs (x-k)+(x+k) from k=span-1 down to 1
//   V (s_pyramid_build_aa.cu:52-86): k=span-1..1: acc+=T[y-k]*g; acc+=T[y+k]*g; then centre
//   level 0 H (s_pyramid_build_ra.cu:17-55): pairs outermost-in, then centre, then *255
#include "psx_internal.h"
#include "blur_arith.h"
#include "blur_interp.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <type_traits>
#include <vector>

namespace {

constexpr int TW = 64;    // strip width (columns per workgroup)
constexpr int BR = 32;    // rows per marching step
constexpr int NT = 256;   // threads per workgroup

__device__ __forceinline__ int xcd_remap(int b, int n) { return psx_xcd_remap(b, n); }

struct BlurArgs {
    const float* src;
    float*       dst;
    float*       half_dst;      // next octave level 0 (pick every second), or nullptr
    int W, H, pitch, half_pitch;
    int src_pitch;              // floats per source row
    int src_xoff, src_width;    // source column of output column 0; number of valid source columns
    int nstrips, chunk_rows;
    PsxTaps taps;               // horizontal taps (and vertical, unless LEVEL0)
    PsxTaps taps_v;             // LEVEL0 only: vertical taps
#ifdef PSX_PHASE_TIMING
    int dbg;                    // measurement build only: 1 = all loads from 64 cache-resident rows, 2 = no stores
#endif
};

#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))
#ifdef PSX_PHASE_TIMING
__device__ long long* g_blur_dbg = nullptr;
extern "C" void psx_debug_set_blur_buffer(long long* d) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_blur_dbg), &d, sizeof(d)); }
__device__ long long* g_l0_dbg = nullptr;         // the same stamps in k_level0_x2 (tools/level0_phase.py)
extern "C" void psx_debug_set_level0_buffer(long long* d) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_l0_dbg), &d, sizeof(d)); }
#define BSTAMP(i) do { long long c_ = clock64(); tacc[i] += c_ - tprev; tprev = c_; } while (0)
#else
#define BSTAMP(i)
#endif
// ---------------------------------------------------------------------------------------------
// k_blur: marching strips, laid out so that the loop body is almost only the arithmetic (an earlier
// version spent ~2/3 of its VALU instructions on addresses, swizzles and ring wrap-around):
//   * staged rows have an ODD stride in 16-byte chunks and the horizontal pass maps its lanes to
//     (row, segment) so that every ds_read_b128 lane group is {2 adjacent rows} x {8 segments}:
//     conflict free with plain immediate offsets, no XOR swizzle;
//   * the ring of H-filtered rows is stored unpermuted with a 68-float row stride (aligned, conflict-free
//     b128 writes and b64 reads) and its first rows are mirrored behind its end, so a
//     vertical window never wraps: every ds_read_b64 is base + immediate offset;
//   * all per-thread staging geometry is computed once, outside the step loop;
//   * vertical pass on two adjacent columns per lane (v_pk_fma_f32), horizontal pass packed by
//     the compiler over adjacent outputs.
// ---------------------------------------------------------------------------------------------
template <int R>
struct Geom2 {
    static constexpr int HALO = (R + 3) & ~3;
    static constexpr int SW   = TW + 2 * HALO;          // staged row width (floats)
    static constexpr int SW4  = SW / 4;
    static constexpr int NLD  = (BR * SW4 + NT - 1) / NT;
    static constexpr int SWA  = 4 * (SW4 | 1);          // odd number of chunks per row
    static constexpr int RING = (BR + 2 * R <= 64) ? 64 : 128;
    static constexpr int VWIN = 4 + 2 * R;              // ring rows read by one vertical thread
    static constexpr int MIRROR = VWIN - 1;             // slots < MIRROR are duplicated at slot + RING
    static constexpr int RS   = TW + 4;                 // ring row stride (floats): 16-byte aligned rows, == 4 (mod 8)
};

// hfilter8_km / vfilter2x4_km: blur_arith.h (shared with the tile kernel and its host emulation)

// FLOW (k_pyramid_flow, the whole-pyramid kernel with device-side dependencies): 0 = an ordinary launch; 1 = every store is a
// system-scope (write-through) store, so that a dependent workgroup of the SAME launch may read the rows once this
// workgroup has drained its stores and bumped the chunk counter, plain loads; 2 = as 1 and the staging loads bypass the
// vector L1 (buffer_load ... sc1: MI355X_MICROARCH.md "Valid forms": sc0 sc1 stores + sc1 loads on both sides).
template <int R, bool LEVEL0, bool DEFER, int FLOW = 0>
__device__ __forceinline__ void blur_body(const BlurArgs& a, const int lid, float* const s_stage, float* const s_ring, const int t)
{
    using G = Geom2<R>;
    constexpr int HALO = G::HALO, SW4 = G::SW4, NLD = G::NLD, SWA = G::SWA, RING = G::RING;
    constexpr int VWIN = G::VWIN, MIRROR = G::MIRROR, RS = G::RS;
    constexpr bool LAST_PARTIAL = (BR * SW4) % NT != 0;     // only the last staging slot can be empty
    constexpr bool FAST_ROWS = R < 13;

    // explicitly global: in k_pyramid_flow the pointers come out of a job record in memory and would otherwise be
    // generic (flat_load / flat_store); for kernel arguments the casts change nothing
    const GLOBAL_AS float* const gsrc = (const GLOBAL_AS float*)a.src;
    GLOBAL_AS float* const gdst = (GLOBAL_AS float*)a.dst;
    GLOBAL_AS float* const ghalf = (GLOBAL_AS float*)a.half_dst;
    const int strip = lid % a.nstrips;
    const int chunk = lid / a.nstrips;
    const int x0    = strip * TW;
    const int Y0    = chunk * a.chunk_rows;
    const int Y1    = min(Y0 + a.chunk_rows, a.H);
    const int nsteps = (Y1 - Y0 + 2 * R + BR - 1) / BR;
    // workgroup uniform: every staged column exists in the source row (no clamping needed)
    const bool interior = (x0 - HALO + a.src_xoff >= 0) && (x0 + TW + HALO + a.src_xoff <= a.src_width);

    // ---- staging geometry of this thread (step invariant) ----
    int st_row[NLD], st_x[NLD], st_lds[NLD];
    unsigned st_off[NLD];                               // byte offset from the first staged row of a step
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int idx = t + j * NT;
        const int row = idx / SW4, c4 = idx - row * SW4;
        st_row[j] = row;
        st_x[j]   = x0 - HALO + c4 * 4 + a.src_xoff;
        st_lds[j] = row * SWA + c4 * 4;
        st_off[j] = FAST_ROWS ? (unsigned)(row * a.src_pitch + st_x[j]) * 4u : 0u;
    }
    const bool last_on = !LAST_PARTIAL || (t + (NLD - 1) * NT < BR * SW4);

    // ---- horizontal pass geometry: lane -> (row, 8-column segment).  Two constraints (MI355X_MICROARCH.md,
    // LDS): the ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} must each hold two rows
    // an odd distance apart (rows {0,1} / {2,3}; the staged row stride is an odd number of chunks), and
    // the ds_write_b128 groups (8 contiguous lanes) of the ring store must pair rows an odd distance
    // apart as well (ring row stride == 4 mod 8 dwords): (0,3) (3,0) (2,1) (1,2). ----
    int h_row, h_seg;
    {
        const int blk = (t & 31) >> 2;                    // 4-lane block within the 32-lane half
        const int rq = (0x21120330 >> (4 * blk)) & 3;     // rows          0 3 3 0 2 1 1 2
        const int sh = (0xCC >> blk) & 1;                 // segments 4-7? 0 0 1 1 0 0 1 1
        h_row = (t >> 6) * 8 + ((t >> 5) & 1) * 4 + rq;
        h_seg = sh * 4 + (t & 3);
    }
    const LDS_AS float* h_src = (const LDS_AS float*)&s_stage[h_row * SWA + h_seg * 8];
    // ---- vertical pass geometry: thread = (2 adjacent columns, 4 output rows) ----
    const int v_pp = t & 31, v_rg = t >> 5;
    const int v_x  = x0 + 2 * v_pp;
    // byte offsets relative to row (Y0 - 2R + k*BR) of the destination planes; the row part is uniform
    const unsigned v_doff = (unsigned)((v_rg * 4) * a.pitch + v_x) * 4u;
    const bool v_xok = v_x < a.W, v_pair = v_x + 1 < a.W;

#ifdef PSX_PHASE_TIMING
    long long tacc[5] = {0, 0, 0, 0, 0};
    long long tprev = clock64();
    const long long tstart = tprev;
#endif
    auto run = [&](auto interior_c) {
        constexpr bool INTERIOR = decltype(interior_c)::value;
        v4f pre[NLD];
        float pre_l[INTERIOR ? 1 : NLD], pre_r[INTERIOR ? 1 : NLD];     // edge strips: the staged rows' first / last valid pixel
        // FLOW == 2: the source plane as a raw buffer (base = plane, 32-bit byte offsets), loads with cache policy sc1
        auto src_rsrc = [&]() { return __builtin_amdgcn_make_buffer_rsrc((float*)gsrc, 0, 0x7fffffff, 0x00020000); };
        auto issue = [&](int k) {
            const int ybase = Y0 - R + k * BR;
            // workgroup uniform: every staged row of this step exists (all but the first / last chunk of a plane).
            // Not at R >= 13: the extra per-thread offsets do not fit into 128 VGPRs there.
            const bool rows_exist = FAST_ROWS && ybase >= 0 && ybase + BR <= a.H;
            const GLOBAL_AS char* step_base = reinterpret_cast<const GLOBAL_AS char*>(gsrc + (ptrdiff_t)ybase * a.src_pitch);
#pragma unroll
            for (int j = 0; j < NLD; j++) {
                if (j < NLD - 1 || last_on) {
                    int y = psx_clampi(ybase + st_row[j], 0, a.H - 1);
#ifdef PSX_PHASE_TIMING
                    if (a.dbg & 1) y &= 63;
#endif
                    if (INTERIOR) {
                        if (rows_exist) {
                            // no clamp: uniform row base (scalar) + the thread's step-invariant byte offset
                            if constexpr (FLOW == 2) {
                                const v4u q = __builtin_amdgcn_raw_buffer_load_b128(src_rsrc(), st_off[j], ybase * a.src_pitch * 4, 16);
                                __builtin_memcpy(&pre[j], &q, 16);
                            } else pre[j] = *reinterpret_cast<const GLOBAL_AS v4f*>(step_base + st_off[j]);
                        } else {
                            const unsigned off = (unsigned)(y * a.src_pitch + st_x[j]) * 4u;
                            if constexpr (FLOW == 2) {
                                const v4u q = __builtin_amdgcn_raw_buffer_load_b128(src_rsrc(), off, 0, 16);
                                __builtin_memcpy(&pre[j], &q, 16);
                            } else pre[j] = *reinterpret_cast<const GLOBAL_AS v4f*>(reinterpret_cast<const GLOBAL_AS char*>(gsrc) + off);
                        }
                    } else {
                        // Edge strip (the first / last of a plane): the 16-byte slot at the clamped in-row position, plus
                        // the row's first / last valid pixel (one address per row: broadcast loads); the columns outside the
                        // plane are patched in commit(), when the loads have landed -- nothing here waits.  (Four clamped
                        // dword loads per slot made the two edge strips, for which every dependent workgroup of the
                        // whole-pyramid kernel waits and with which a level ends, ~1.5-2x slower than the other 58.)
                        const int xc = psx_clampi(st_x[j], 0, a.src_pitch - 4);
                        const unsigned rb = (unsigned)(y * a.src_pitch) * 4u;
                        if constexpr (FLOW == 2) {
                            const v4u u = __builtin_amdgcn_raw_buffer_load_b128(src_rsrc(), rb + (unsigned)xc * 4u, 0, 16);
                            __builtin_memcpy(&pre[j], &u, 16);
                            const unsigned ul = __builtin_amdgcn_raw_buffer_load_b32(src_rsrc(), rb, 0, 16);
                            const unsigned ur = __builtin_amdgcn_raw_buffer_load_b32(src_rsrc(), rb + (unsigned)(a.src_width - 1) * 4u, 0, 16);
                            __builtin_memcpy(&pre_l[j], &ul, 4); __builtin_memcpy(&pre_r[j], &ur, 4);
                        } else {
                            pre[j] = *reinterpret_cast<const GLOBAL_AS v4f*>(reinterpret_cast<const GLOBAL_AS char*>(gsrc) + rb + (unsigned)xc * 4u);
                            const GLOBAL_AS float* rp = gsrc + (size_t)y * a.src_pitch;
                            pre_l[j] = rp[0]; pre_r[j] = rp[a.src_width - 1];
                        }
                    }
                }
            }
        };
        auto commit = [&]() {
#pragma unroll
            for (int j = 0; j < NLD; j++)
                if (j < NLD - 1 || last_on) {
                    v4f q = pre[j];
                    if constexpr (!INTERIOR) {
                        // xc == x whenever any of the slot's four columns is inside the plane (x is a multiple of 4)
                        const int x = st_x[j];
                        q.x = x + 0 < 0 ? pre_l[j] : (x + 0 > a.src_width - 1 ? pre_r[j] : q.x);
                        q.y = x + 1 < 0 ? pre_l[j] : (x + 1 > a.src_width - 1 ? pre_r[j] : q.y);
                        q.z = x + 2 < 0 ? pre_l[j] : (x + 2 > a.src_width - 1 ? pre_r[j] : q.z);
                        q.w = x + 3 < 0 ? pre_l[j] : (x + 3 > a.src_width - 1 ? pre_r[j] : q.w);
                    }
                    *reinterpret_cast<v4f*>(&s_stage[st_lds[j]]) = q;
                }
        };

        // Results of the vertical pass are not stored at once: on gfx950 loads and stores share one counter
        // (vmcnt), so "wait for the prefetched rows" at the top of the next step would also wait for stores
        // issued a moment ago -- an exposed HBM write latency per step.  The four row pairs of step k stay in
        // registers and are stored at the top of step k+1, behind the commit; by the next wait they are a
        // whole step old.
        v2f pend[4];
        // kk = the step whose results are pending: workgroup uniform, so the row base stays in scalar registers;
        // threads that skipped the vertical pass of step kk hold rows outside [Y0, Y1) and store nothing, and for
        // kk = -1 (before the first step) every row lies above Y0
        auto flush = [&](const int kk) {
#ifdef PSX_PHASE_TIMING
            if (a.dbg & 2) return;
#endif
            const int r_out0 = Y0 + kk * BR - 2 * R + v_rg * 4;
            GLOBAL_AS char* drow = reinterpret_cast<GLOBAL_AS char*>(gdst + (ptrdiff_t)(Y0 - 2 * R + kk * BR) * a.pitch);
#ifdef PSX_PHASE_TIMING
            if (a.dbg & 4) drow = reinterpret_cast<GLOBAL_AS char*>(gdst + (ptrdiff_t)(64 + (blockIdx.x & 7) * 40) * a.pitch);   // stores stay in L2
#endif
            // Fast path (workgroup uniform): every row of step kk lies inside the chunk and the strip is a full one with both
            // columns of every pair inside the plane -- all but the first flush of a chunk and the plane's last strip.  No
            // per-row range test, no exec juggling around the stores: 4 x (base of the row in scalar registers + the thread's
            // 32-bit offset).
            if constexpr (FLOW == 0) {
                if (kk * BR >= 2 * R && Y0 + (kk + 1) * BR - 2 * R <= Y1 && x0 + TW <= a.W) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        GLOBAL_AS char* di = (drow + (size_t)i * a.pitch * 4) + v_doff;
                        unsigned long long bits; __builtin_memcpy(&bits, &pend[i], 8);
                        __hip_atomic_store(reinterpret_cast<GLOBAL_AS unsigned long long*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    if (a.half_dst != nullptr) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int r_out = r_out0 + i;
                            if ((r_out & 1) == 0) ghalf[(size_t)(r_out >> 1) * a.half_pitch + (v_x >> 1)] = pend[i].x;
                        }
                    }
                    return;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r_out = r_out0 + i;
                if (r_out >= Y0 && r_out < Y1 && v_xok) {
                    GLOBAL_AS char* di = (drow + (size_t)i * a.pitch * 4) + v_doff;
                    if (v_pair) {
                        // system-scope store (sc0 sc1): written through the XCD's L2 while the kernel runs.  Plain
                        // stores left ~33 MB of dirty lines to be written back after the last wave, inside the
                        // kernel's duration (17.6 -> 16.9 us per octave-0 launch); agent scope measured the same,
                        // non-temporal stores 4 % slower (the next level reads these rows).
                        unsigned long long bits; __builtin_memcpy(&bits, &pend[i], 8);
                        __hip_atomic_store(reinterpret_cast<GLOBAL_AS unsigned long long*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    } else if constexpr (FLOW != 0) {
                        const float px_ = pend[i].x; unsigned bits; __builtin_memcpy(&bits, &px_, 4);
                        __hip_atomic_store(reinterpret_cast<GLOBAL_AS unsigned*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    } else *reinterpret_cast<GLOBAL_AS float*>(di) = pend[i].x;
                    // get_by_2_pick_every_second: rows and columns 0,2,4,.. (v_x is even)
                    if (a.half_dst != nullptr && (r_out & 1) == 0) {
                        GLOBAL_AS float* hd = ghalf + (size_t)(r_out >> 1) * a.half_pitch + (v_x >> 1);
                        if constexpr (FLOW != 0) {
                            const float px_ = pend[i].x; unsigned bits; __builtin_memcpy(&bits, &px_, 4);
                            __hip_atomic_store(reinterpret_cast<GLOBAL_AS unsigned*>(hd), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        } else *hd = pend[i].x;
                    }
                }
            }
        };

        issue(0);
        for (int k = 0; k < nsteps; k++) {
            commit();
            if (DEFER) flush(k - 1);
            BSTAMP(0);
            __syncthreads();
            BSTAMP(1);
            if (k + 1 < nsteps) issue(k + 1);

            // ---- horizontal ----
            {
                float win[8 + 2 * HALO];
#pragma unroll
                for (int q = 0; q < (8 + 2 * HALO) / 4; q++) {
                    // volatile: keep one ds_read_b128 per chunk (otherwise the vectoriser re-reads every
                    // odd-aligned pair with ds_read2_b32, at a quarter of the b128 rate and with conflicts);
                    // issuing the chunks in the filter's use order instead (centre, outermost inwards) measured 3 % slower
                    const v4f v = ((const volatile LDS_AS v4f*)h_src)[q];
                    win[4 * q + 0] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
                }
                float out[8];
                hfilter8_km<R, HALO, LEVEL0>(win, a.taps, out);
                const int slot = (k * BR + h_row) & (RING - 1);
                float* rp = &s_ring[slot * RS + h_seg * 8];
                reinterpret_cast<float4*>(rp)[0] = make_float4(out[0], out[1], out[2], out[3]);
                reinterpret_cast<float4*>(rp)[1] = make_float4(out[4], out[5], out[6], out[7]);
                if (slot < MIRROR) {
                    reinterpret_cast<float4*>(rp + RING * RS)[0] = make_float4(out[0], out[1], out[2], out[3]);
                    reinterpret_cast<float4*>(rp + RING * RS)[1] = make_float4(out[4], out[5], out[6], out[7]);
                }
            }
            BSTAMP(2);
            __syncthreads();
            BSTAMP(3);

            // ---- vertical ----
            {
                const int rel0 = k * BR - 2 * R + v_rg * 4;       // ring-relative index of T[r_out0 - R]
                const int r_out0 = Y0 + rel0;
                if (r_out0 + 3 >= Y0 && r_out0 < Y1) {
                    const LDS_AS float* vp = (const LDS_AS float*)&s_ring[(rel0 & (RING - 1)) * RS + 2 * v_pp];
                    v2f v[VWIN];
#pragma unroll
                    // volatile: plain ds_read_b64 (256 B/clk); merged ds_read2_b64 runs at half that rate
                    for (int j = 0; j < VWIN; j++) v[j] = *(const volatile LDS_AS v2f*)(vp + j * RS);
                    v2f o[4];
                    vfilter2x4_km<R>(v, LEVEL0 ? a.taps_v : a.taps, o);
                    // pin the four results here: otherwise each chain is sunk into its own predicated
                    // store block and runs alone, dependent v_pk_fma_f32 back to back
                    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
                    for (int i = 0; i < 4; i++) pend[i] = o[i];
                }
            }
            if (!DEFER) flush(k);
            BSTAMP(4);
        }
        if (DEFER) flush(nsteps - 1);
    };
    if (interior) run(std::true_type{}); else run(std::false_type{});
#ifdef PSX_PHASE_TIMING
    if (threadIdx.x == 0 && g_blur_dbg) {
        for (int q = 0; q < 5; q++) g_blur_dbg[blockIdx.x * 8 + q] = tacc[q];
        g_blur_dbg[blockIdx.x * 8 + 5] = clock64() - tstart; g_blur_dbg[blockIdx.x * 8 + 6] = nsteps;
    }
#endif
}


// ---------------------------------------------------------------------------------------------
// blur_body_dma: the same marching strip with the staged rows brought in by LDS-DMA
// (global_load_lds_dwordx4: HBM/L2 -> LDS without passing through VGPRs) into a ring of NBUF stage
// buffers, NBUF-1 steps ahead.  Why: a 3840x2160 plane is ONE round of ~900 workgroups that all
// start together and run in lockstep, so with a one-step register prefetch the memory system idles
// while every workgroup computes and the VALUs idle while every workgroup waits (kernel time ~ sum
// of the two phases).  Two steps of loads in flight per workgroup keep HBM busy through the
// arithmetic; the commit phase (3-4 ds_write_b128 per thread and step) and its 12-16 VGPRs go away.
//
// vmcnt accounting (the compiler does not count an asm load, cdna_hip_programming.md 5.7): per step a wave issues
//   s_waitcnt vmcnt(pending DMA batches newer than this step's) -> flush(k-1) stores -> barrier A -> DMA(k+NBUF-1)
// so at the wait the NEWEST operations are exactly the DMA batches that may stay in flight (NLD wave
// instructions each, every wave issues all of them: partial lane masks only); everything older -- this
// step's batch and the stores of step k-2 -- must have completed.  Loads return in order among loads, so
// the count is right whether or not stores may overtake loads.
//
// Staged layout = the old one (odd number of 16-byte chunks per row, pad chunk never written): chunk c of a
// buffer is row c / CH, column chunk c % CH; DMA instruction i (NLDT per step, NLD per wave) covers chunks
// [i*P, (i+1)*P) with lanes 0..P-1, its LDS base (M0) = buffer + i*P*16.  Columns outside the source row
// (first / last strip) cannot be clamped per element by a 16-byte DMA: those strips load in-row chunks and
// patch the out-of-range columns in LDS after the batch has landed (one extra barrier, edge strips only).
// ---------------------------------------------------------------------------------------------
template <int R, int NBUF, int RINGROWS>
struct GeomD {
    static constexpr int HALO = (R + 3) & ~3;
    static constexpr int SW   = TW + 2 * HALO;
    static constexpr int SW4  = SW / 4;
    static constexpr int CH   = SW4 | 1;                // chunks per staged row (odd)
    static constexpr int SWA  = 4 * CH;
    static constexpr int NCH  = BR * CH;                // chunks per stage buffer
    static constexpr int nldt() { for (int n = 4; n <= 64; n += 4) if (NCH % n == 0 && NCH / n <= 64) return n; return 0; }
    static constexpr int NLDT = nldt();
    static constexpr int NLD  = NLDT / 4;               // DMA wave instructions per wave and step
    static constexpr int P    = NCH / NLDT;             // chunks (= active lanes) per DMA instruction
    static constexpr int RING = RINGROWS;
    static constexpr int VWIN = 4 + 2 * R;
    static constexpr int MIRROR = VWIN - 1;
    static constexpr int RS   = TW + 4;
    static constexpr int STAGE_FLOATS = BR * SWA;
    static constexpr int LDS_FLOATS = NBUF * STAGE_FLOATS + (RING + MIRROR) * RS;
    static_assert(NLDT > 0 && P <= 64, "no DMA tiling for this radius");
    static_assert(RING >= BR + 2 * R && RING % 16 == 0, "ring too small");
    static_assert((SW4 & 1) == 0, "the pad chunk is assumed to exist");
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// one LDS-DMA wave instruction: lanes 0..P-1 each move 16 bytes from (base + voff[lane]) to LDS byte address
// lds_dst + 16*lane.  EXEC is narrowed to the low P lanes inside the statement (the callers run in wave-uniform
// control flow with all 64 lanes on) and M0, which is compiler-reserved, is saved and restored; s_nop 4 covers the
// SALU-written base / M0 / EXEC -> VMEM hazards, which hipcc does not pad inside an asm statement.
template <int P>
__device__ __forceinline__ void dma16(const char* base, unsigned voff, unsigned lds_dst)
{
    unsigned keep_m0;
    unsigned long long keep_exec;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b64 %1, exec\n\t"
                 "s_lshr_b64 exec, -1, %5\n\t"
                 "s_mov_b32 m0, %4\n\t"
                 "s_nop 4\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep_m0), "=&s"(keep_exec) : "v"(voff), "s"(base), "s"(lds_dst), "n"(64 - P) : "memory");
}

template <int R, bool LEVEL0, int NBUF, int RINGROWS>
__device__ __forceinline__ void blur_body_dma(const BlurArgs& a, const int lid)
{
    using G = GeomD<R, NBUF, RINGROWS>;
    constexpr int HALO = G::HALO, SW = G::SW, SW4 = G::SW4, CH = G::CH, SWA = G::SWA, NLD = G::NLD, P = G::P;
    constexpr int RING = G::RING, VWIN = G::VWIN, MIRROR = G::MIRROR, RS = G::RS, STAGE = G::STAGE_FLOATS;
    constexpr int D = NBUF - 1;                          // DMA batches in flight ahead of the step being filtered
    __shared__ __attribute__((aligned(16))) float s_all[G::LDS_FLOATS];
    float* const s_ring = s_all + NBUF * STAGE;

    const int t     = threadIdx.x;
    const int lane  = t & 63;
    const int wv    = __builtin_amdgcn_readfirstlane(t >> 6);
    const int strip = lid % a.nstrips;
    const int chunk = lid / a.nstrips;
    const int x0    = strip * TW;
    const int Y0    = chunk * a.chunk_rows;
    const int Y1    = min(Y0 + a.chunk_rows, a.H);
    const int nsteps = (Y1 - Y0 + 2 * R + BR - 1) / BR;
    const int xs0   = x0 - HALO + a.src_xoff;            // source column of staged column 0
    const bool interior = (xs0 >= 0) && (xs0 + SW <= a.src_width);
    const unsigned lds0 = (unsigned)(unsigned long)(LDS_AS float*)s_all;

    // ---- DMA geometry of this lane (step invariant) ----
    int st_row[NLD]; unsigned st_xb[NLD], st_off[NLD];
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int c = (wv * NLD + j) * P + lane;
        const int row = c / CH, cc = c - row * CH;
        st_row[j] = row;
        // in-row chunk (the source rows are pitch floats long, pitch % 4 == 0): columns outside [0, src_width) are patched
        // later; the lane of the pad chunk (cc == SW4) loads 16 in-row bytes nobody reads, lanes >= P are masked in dma16
        const int xc = psx_clampi(xs0 + cc * 4, 0, a.src_pitch - 4);
        st_xb[j]  = (unsigned)xc * 4u;
        st_off[j] = (unsigned)(row * a.src_pitch) * 4u + st_xb[j];
    }
    auto issue = [&](const int k) {
        const int ybase = Y0 - R + k * BR;
        const unsigned buf = lds0 + (unsigned)((k % NBUF) * STAGE * 4);
        if (ybase >= 0 && ybase + BR <= a.H) {           // workgroup uniform: scalar row base + step-invariant offsets
            const char* step_base = reinterpret_cast<const char*>(a.src + (ptrdiff_t)ybase * a.src_pitch);
#pragma unroll
            for (int j = 0; j < NLD; j++)
                dma16<P>(step_base, st_off[j], buf + (unsigned)((wv * NLD + j) * P * 16));
        } else {
#pragma unroll
            for (int j = 0; j < NLD; j++) {
                const int y = psx_clampi(ybase + st_row[j], 0, a.H - 1);
                const unsigned off = (unsigned)(y * a.src_pitch) * 4u + st_xb[j];
                dma16<P>(reinterpret_cast<const char*>(a.src), off, buf + (unsigned)((wv * NLD + j) * P * 16));
            }
        }
    };

    // ---- horizontal pass geometry (as blur_body) ----
    int h_row, h_seg;
    {
        const int blk = (t & 31) >> 2;
        const int rq = (0x21120330 >> (4 * blk)) & 3;
        const int sh = (0xCC >> blk) & 1;
        h_row = (t >> 6) * 8 + ((t >> 5) & 1) * 4 + rq;
        h_seg = sh * 4 + (t & 3);
    }
    const int h_off = h_row * SWA + h_seg * 8;           // floats into a stage buffer
    const int v_pp = t & 31, v_rg = t >> 5;
    const int v_x  = x0 + 2 * v_pp;
    const unsigned v_doff = (unsigned)((v_rg * 4) * a.pitch + v_x) * 4u;
    const bool v_xok = v_x < a.W, v_pair = v_x + 1 < a.W;
    // edge strips: staged columns [0, e_l) take the value of column e_l, columns [e_r, SW) that of column e_r - 1
    const int e_l = min(max(-xs0, 0), SW - 1), e_r = max(min(a.src_width - xs0, SW), 1);

    v2f pend[4];
    auto flush = [&](const int kk) {
        const int r_out0 = Y0 + kk * BR - 2 * R + v_rg * 4;
        char* drow = reinterpret_cast<char*>(a.dst + (ptrdiff_t)(Y0 - 2 * R + kk * BR) * a.pitch);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r_out = r_out0 + i;
            if (r_out >= Y0 && r_out < Y1 && v_xok) {
                char* di = (drow + (size_t)i * a.pitch * 4) + v_doff;
                if (v_pair) {
                    unsigned long long bits; __builtin_memcpy(&bits, &pend[i], 8);
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else *reinterpret_cast<float*>(di) = pend[i].x;
                if (a.half_dst != nullptr && (r_out & 1) == 0)
                    a.half_dst[(size_t)(r_out >> 1) * a.half_pitch + (v_x >> 1)] = pend[i].x;
            }
        }
    };

    if (NBUF == 1) issue(0);                             // single stage buffer: batch k+1 goes out after H(k) has read batch k
#pragma unroll
    for (int q = 0; q < D; q++) if (q < nsteps) issue(q);
    int sbase = 0;                                       // ring slot of the first row filtered in step k = (k * BR) mod RING
    for (int k = 0; k < nsteps; k++) {
        // this step's batch has landed (newer batches stay in flight)
        if (D >= 2 && k + 1 < nsteps) {
            if (D >= 3 && k + 2 < nsteps) wait_vmcnt<(D >= 3 ? 2 : 0) * NLD>(); else wait_vmcnt<(D >= 2 ? 1 : 0) * NLD>();
        } else wait_vmcnt<0>();
        flush(k - 1);
        __syncthreads();                                 // A: everyone's part of batch k is in LDS; V(k-1) and H(k-1) are done
        if (NBUF > 1 && k + D < nsteps) issue(k + D);    // into the buffer H(k-1) read
        float* const stage = s_all + (k % NBUF) * STAGE;
        if (!interior) {
            const int row = t >> 3, sub = t & 7;
            float* rp = stage + row * SWA;
            const float vl = rp[e_l], vr = rp[e_r - 1];
            for (int c = sub; c < e_l; c += 8) rp[c] = vl;
            for (int c = e_r + sub; c < SW; c += 8) rp[c] = vr;
            __syncthreads();
        }

        // ---- horizontal ----
        {
            const LDS_AS float* h_src = (const LDS_AS float*)(stage + h_off);
            float win[8 + 2 * HALO];
#pragma unroll
            for (int q = 0; q < (8 + 2 * HALO) / 4; q++) {
                const v4f v = ((const volatile LDS_AS v4f*)h_src)[q];
                win[4 * q + 0] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
            }
            float out[8];
            hfilter8_km<R, HALO, LEVEL0>(win, a.taps, out);
            int slot = sbase + h_row; if (slot >= RING) slot -= RING;
            float* rp = &s_ring[slot * RS + h_seg * 8];
            reinterpret_cast<float4*>(rp)[0] = make_float4(out[0], out[1], out[2], out[3]);
            reinterpret_cast<float4*>(rp)[1] = make_float4(out[4], out[5], out[6], out[7]);
            if (slot < MIRROR) {
                reinterpret_cast<float4*>(rp + RING * RS)[0] = make_float4(out[0], out[1], out[2], out[3]);
                reinterpret_cast<float4*>(rp + RING * RS)[1] = make_float4(out[4], out[5], out[6], out[7]);
            }
        }
        __syncthreads();                                 // B
        if (NBUF == 1 && k + 1 < nsteps) issue(k + 1);   // the one stage buffer is free again

        // ---- vertical ----
        {
            const int rel0 = k * BR - 2 * R + v_rg * 4;
            const int r_out0 = Y0 + rel0;
            if (r_out0 + 3 >= Y0 && r_out0 < Y1) {
                int vs = sbase - 2 * R + v_rg * 4;       // ring slot of T[r_out0 - R]
                if (vs < 0) vs += RING;
                if (vs >= RING) vs -= RING;
                const LDS_AS float* vp = (const LDS_AS float*)&s_ring[vs * RS + 2 * v_pp];
                v2f v[VWIN];
#pragma unroll
                for (int j = 0; j < VWIN; j++) v[j] = *(const volatile LDS_AS v2f*)(vp + j * RS);
                v2f o[4];
                vfilter2x4_km<R>(v, LEVEL0 ? a.taps_v : a.taps, o);
                asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
                for (int i = 0; i < 4; i++) pend[i] = o[i];
            }
        }
        sbase += BR; if (sbase >= RING) sbase -= RING;
    }
    flush(nsteps - 1);
}

// workgroups of 256 threads a CU can hold for a variant (LDS bound; 4 waves per SIMD at <= 128 VGPRs)
template <int R, int NBUF, int RINGROWS>
constexpr int dma_wg_per_cu()
{
    const int lds = GeomD<R, NBUF, RINGROWS>::LDS_FLOATS * 4;
    const int n = (160 * 1024) / lds;
    const int cap = NBUF == 1 ? 5 : 4;               // the single-buffer variant is lean enough for 5 (<= 96 VGPRs asked for)
    return n > cap ? cap : (n < 1 ? 1 : n);
}

template <int R, bool LEVEL0, int NBUF, int RINGROWS>
__global__ __launch_bounds__(NT, (dma_wg_per_cu<R, NBUF, RINGROWS>())) void k_blur_dma(BlurArgs a)
{
    blur_body_dma<R, LEVEL0, NBUF, RINGROWS>(a, xcd_remap(blockIdx.x, gridDim.x));
}

// (a variant with the horizontal pass packed over row pairs, the form of pyramid_fixed.hip / blur_interp.h, was built here in
// round 6, bit-exact, and measured 12-17 % SLOWER although its H pass has a third fewer instructions:
// profiles/r06_blur_rowpair_h_ab.txt; the code is in the history)
template <int R, bool LEVEL0, bool DEFER = true>
__global__ __launch_bounds__(NT, (R <= 13) ? 4 : ((R <= 22) ? 2 : 1)) void k_blur(BlurArgs a)
{
    using G = Geom2<R>;
    __shared__ __attribute__((aligned(16))) float s_stage[BR * G::SWA];
    __shared__ __attribute__((aligned(16))) float s_ring[(G::RING + G::MIRROR) * G::RS];
    blur_body<R, LEVEL0, DEFER>(a, xcd_remap(blockIdx.x, gridDim.x), s_stage, s_ring, threadIdx.x);
}

// Two independent planes in one launch (the diagonal schedule of psx_build_pyramid: level l of octave o
// together with level l-3 of octave o+1): the first na logical blocks belong to job a, the rest to job b.
// Small octaves are latency chains of ~5 us launches; riding along with a larger octave's launch hides them.
template <int R>
__global__ __launch_bounds__(NT, (R <= 13) ? 4 : ((R <= 22) ? 2 : 1)) void k_blur2(BlurArgs a, BlurArgs b, int na)
{
    using G = Geom2<R>;
    __shared__ __attribute__((aligned(16))) float s_stage[BR * G::SWA];
    __shared__ __attribute__((aligned(16))) float s_ring[(G::RING + G::MIRROR) * G::RS];
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    // two inlined copies: selecting the arguments through a pointer moves the taps out of the preloaded
    // kernel-argument SGPRs (120 bytes of VGPR spills at R = 13)
    if (lid < na) blur_body<R, false, true>(a, lid, s_stage, s_ring, threadIdx.x);
    else          blur_body<R, false, true>(b, lid - na, s_stage, s_ring, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_pyramid_flow: every plane-to-plane blur of a frame (levels 1..L-1 of every octave, the decimations riding on the
// level L-3 stores as before) in ONE launch.  The launch-per-level schedule pays a kernel boundary per level -- ~19 of
// them per 1080p frame, and octaves 1-4 (a third of the pixels) are a chain of ~14 launches of ~5 us each that cannot fill
// the chip.  Here a persistent grid (4 workgroups per CU) pulls work items
//     (job = (octave, level), 64-column strip, chunk of rows)            -- exactly the (strip, chunk) of one k_blur workgroup
// from a list the host has sorted topologically (psx_flow_plan); an item starts when the chunks of its SOURCE plane that
// hold its input rows (+- R) are complete.  Mechanics (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup
// visibility" and the price list):
//   * tickets: PSX_FLOW_SHARDS counters, one per class blockIdx & 7, each on its own 128-byte line (a single word
//     saturates at ~88 dequeues / us; 1024 workgroups starting together would queue for 12 us); class c owns the items
//     c, c + 8, c + 16, .. of the ONE global order and takes them in order.  Deadlock freedom needs no co-residency of
//     the whole grid: the smallest unfinished item is either running (its dependencies are smaller, hence done) or the
//     next ticket of its class, which the class's oldest workgroup takes as soon as it finishes a smaller item.  The
//     next ticket is requested before the current item is worked on (its latency hides behind the item).
//   * publish: the body's stores are system-scope (write-through, FLOW != 0); every thread drains its stores
//     (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane bumps the chunk counter (agent-scope atomic).
//   * wait: lanes of wave 0 poll the <= 64 chunk counters of the item with agent-scope relaxed loads (+ s_sleep), the
//     workgroup meets at a barrier, then reads the plane: sc1 loads (LD = 2) or plain loads (LD = 1: no line of the
//     source rows can be resident in this CU's L1 or this XCD's L2 before they were complete -- a frame writes every
//     plane address once, lines are never shared between producers, and caches are invalidated at kernel start).
//   * every wait is bounded; a workgroup that runs into the bound raises PsxCounters::flow_error and carries on, so a
//     protocol bug ends in an error code from psx_counts, never in a hung GPU.
// The arithmetic is blur_body's, instruction for instruction: planes stay bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr int FLOW_NR = 5;
constexpr int flow_radius(int i) { return i == 0 ? 5 : i == 1 ? 7 : i == 2 ? 8 : i == 3 ? 10 : 13; }
constexpr int FLOW_LDS_FLOATS = BR * Geom2<13>::SWA + (Geom2<13>::RING + Geom2<13>::MIRROR) * Geom2<13>::RS;
constexpr int FLOW_SPIN_LIMIT = 1 << 19;           // polls of ~2 us each: about a second

template <int R, int LD>
__device__ __forceinline__ void flow_run(const PsxFlowJob* __restrict__ jb, const int lid, float* s_lds)
{
    using G = Geom2<R>;
    static_assert(BR * G::SWA + (G::RING + G::MIRROR) * G::RS <= FLOW_LDS_FLOATS, "flow LDS too small");
    BlurArgs a;
    // pointers loaded from memory are generic to the compiler (flat_load / flat_store, which also count on lgkmcnt);
    // the round trip through address space 1 tells it they are global, like kernel arguments
    a.src = (const float*)(const GLOBAL_AS float*)jb->src;
    a.dst = (float*)(GLOBAL_AS float*)jb->dst;
    a.half_dst = (float*)(GLOBAL_AS float*)jb->half_dst;
    a.W = jb->W; a.H = jb->H; a.pitch = jb->pitch; a.half_pitch = jb->half_pitch;
    a.src_pitch = jb->pitch; a.src_xoff = 0; a.src_width = jb->W;
    a.nstrips = jb->nstrips; a.chunk_rows = jb->chunk_rows;
#pragma unroll
    for (int i = 0; i <= R; i++) a.taps.g[i] = jb->taps.g[i];
#ifdef PSX_PHASE_TIMING
    a.dbg = 0;
#endif
    // the thread index is laundered per item: otherwise the compiler hoists the thread-invariant geometry of ALL five
    // bodies out of the persistent loop and spills it (200 bytes of scratch per thread)
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    blur_body<R, false, true, LD>(a, lid, s_lds, s_lds + BR * G::SWA, t);
}

template <int LD>
__global__ __launch_bounds__(NT, 4) void k_pyramid_flow(const PsxFlowJob* __restrict__ jobs, const PsxFlowItem* __restrict__ items,
                                                        int nitems, int* __restrict__ state, int* __restrict__ err, int dbg,
                                                        long long* __restrict__ trace)
{
    __shared__ __attribute__((aligned(16))) float s_lds[FLOW_LDS_FLOATS];
    __shared__ int s_ticket;
    const int cls = blockIdx.x & (PSX_FLOW_SHARDS - 1);
    int* const head = state + cls * 32;
    int* const cnt = state + PSX_FLOW_HEAD_INTS;
    int next = 0;
    if (threadIdx.x == 0) next = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (threadIdx.x == 0) {
            s_ticket = next * PSX_FLOW_SHARDS + cls;
            if (s_ticket < nitems) next = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next round
        }
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
        if (ticket >= nitems) break;
        long long tr0 = 0, tr1 = 0, tr2 = 0;
        if (trace != nullptr) tr0 = wall_clock64();
        const PsxFlowItem it = items[ticket];
        const PsxFlowJob* jb = jobs + it.job;
        const int pub = __builtin_amdgcn_readfirstlane(jb->cnt_off + (int)it.chunk);   // kept across the body: no reload in front of the publish
        // ---- wait for the source rows ----
        const int dep0 = jb->dep_cnt_off;
        if (dep0 >= 0 && !(dbg & 1)) {
            if (threadIdx.x < 64) {
                const int nd = (int)it.dep_c1 - (int)it.dep_c0 + 1;
                const int need = jb->dep_need;
                const bool mine = (int)threadIdx.x < nd;
                // every counter has a 128-byte line to itself: a thousand workgroups polling (and sixty publishing into)
                // counters that share a line serialise on it at ~11 ns per access -- tens of microseconds per level
                const int* c = cnt + (size_t)(dep0 + it.dep_c0 + (mine ? (int)threadIdx.x : 0)) * PSX_FLOW_CNT_STRIDE;
                int spins = 0;
                for (;;) {
                    const int v = mine ? __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                    if (__ballot(v < need) == 0ull) break;
                    // back off: the first polls come quickly (a producer that is about to finish), later ones every ~2 us
                    if (spins < 4) __builtin_amdgcn_s_sleep(8); else if (spins < 16) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(80);
                    ++spins;
                    // bounded: ~1 s on this wait, or another workgroup has already given up (then everybody does at once)
                    if ((spins & 255) == 0 &&
                        (spins > FLOW_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        if (threadIdx.x == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
        }
        if (trace != nullptr) tr1 = wall_clock64();
        // ---- the item: one (strip, chunk) of the job's plane, as one k_blur workgroup would do it ----
        const int lid = (int)it.chunk * jb->nstrips + (int)it.strip;
        if (!(dbg & 2))
        switch (jb->rsel) {
            case 0:  flow_run<flow_radius(0), LD>(jb, lid, s_lds); break;
            case 1:  flow_run<flow_radius(1), LD>(jb, lid, s_lds); break;
            case 2:  flow_run<flow_radius(2), LD>(jb, lid, s_lds); break;
            case 3:  flow_run<flow_radius(3), LD>(jb, lid, s_lds); break;
            default: flow_run<flow_radius(4), LD>(jb, lid, s_lds); break;
        }
        if (trace != nullptr) tr2 = wall_clock64();
        // ---- publish: my stores have reached memory; everybody's have; one lane tells the world ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + (size_t)pub * PSX_FLOW_CNT_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (trace != nullptr && threadIdx.x == 0) {
            // measurement only (psx_flow_trace): 100 MHz wall clock at dequeue / dependencies met / arithmetic done / published
            long long* r = trace + (size_t)ticket * 6;
            r[0] = tr0; r[1] = tr1; r[2] = tr2; r[3] = wall_clock64();
            r[4] = ((long long)jb->octave << 40) | ((long long)jb->level << 32) | ((long long)it.chunk << 16) | it.strip;
            r[5] = (long long)blockIdx.x;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Octave 0, level 0: the reference filters a normalised, clamped, bilinear texture of the input image
// (s_image.cu:138-167, s_pyramid_build_ra.cu).  Tap k of output (x, y) reads U(x-k, y) and U(x+k, y)
// with U(X, y) = tex2D at the coordinates of output column X, row y (DESIGN.md "octave 0").  k_upscale
// materialises U once (software model of the texture unit: unnormalise, -0.5, 1.8 fixed-point weight,
// clamped texels, u8 -> v/255), with `pad` extra columns on both sides because U(X) for X outside
// [0, W) is defined by clamping TEXELS, not columns; k_blur<R, true> then runs the "dd" horizontal /
// "inc" vertical filters over it like over any other level.
// ---------------------------------------------------------------------------------------------
struct UpArgs {
    const void* img; int w, h, is_float;
    float* dst; int W, H, pitch, pad, ncol4;
    float shift;
};

__device__ __forceinline__ void l0_axis(float cn, int size, int& i0, float& al)
{
    const float tcoord = cn * (float)size;
    const float tb = tcoord - 0.5f;
    const float fl = floorf(tb);
    float f = tb - fl;
    f = rintf(f * 256.0f) * (1.0f / 256.0f);     // 1.8 fixed-point filter weight
    i0 = (int)fl;
    al = f;
}
__device__ __forceinline__ float l0_lerp(float p, float q, float a) { return fmaf(a, q, (1.0f - a) * p); }

// u8 texel -> v/255 (cudaReadModeNormalizedFloat), correctly rounded without a division: one
// Newton correction of q * fl(1/255) reproduces fl(q / 255) for all 256 inputs (checked exhaustively
// by tests/test_gpu_parity.py::test_u8_normalisation_exact through a ramp image).
__device__ __forceinline__ float l0_unorm8(unsigned q)
{
    const float c = 1.0f / 255.0f;
    const float f = (float)q;
    const float r = f * c;
    const float e = fmaf(-255.0f, r, f);
    return fmaf(e, c, r);
}

// One thread = 4 adjacent columns x UP_ROWS consecutive rows.  The column side (texel indices, 1.8 weights)
// is computed once; the horizontal lerp of a texel row, r(j, X) = lerp(T[j][i0], T[j][i0+1], alpha_X), is
// computed once per texel row and reused by the output rows that share it (at x2 upsampling two output
// rows share each texel row); all lanes of a wave work on the same rows, so the reuse test is a scalar branch.
constexpr int UP_ROWS = 4;

__global__ __launch_bounds__(256) void k_upscale(UpArgs a)
{
    const int t = threadIdx.x;
    const int cq = blockIdx.x * 64 + (t & 63);
    const int y0 = (blockIdx.y * 4 + (t >> 6)) * UP_ROWS;
    if (cq >= a.ncol4 || y0 >= a.H) return;
    int ia[4], ib[4]; float al[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        int i0;
        l0_axis(((float)(cq * 4 + e - a.pad) + a.shift) / (float)a.W, a.w, i0, al[e]);
        ia[e] = psx_clampi(i0, 0, a.w - 1); ib[e] = psx_clampi(i0 + 1, 0, a.w - 1);
    }
    // the 8 texel columns of 4 adjacent outputs normally lie within 4 consecutive texels (any upscale
    // >= 1): they are fetched as one unaligned dword per texel row instead of 8 byte loads
    const int base = min(ia[0], max(a.w - 4, 0));
    const bool packed = !a.is_float && a.w >= 4 && ib[3] - base <= 3;
    int sa[4], sb[4];
#pragma unroll
    for (int e = 0; e < 4; e++) { sa[e] = (ia[e] - base) * 8; sb[e] = (ib[e] - base) * 8; }

    auto lerp_row = [&](int j, float* r) {          // r[e] = lerp_x of texel row j at the 4 columns
        float p[4], q[4];
        if (a.is_float) {
            const float* row = static_cast<const float*>(a.img) + (size_t)j * a.w;
#pragma unroll
            for (int e = 0; e < 4; e++) { p[e] = row[ia[e]]; q[e] = row[ib[e]]; }
        } else {
            const uint8_t* row = static_cast<const uint8_t*>(a.img) + (size_t)j * a.w;
            if (packed) {
                unsigned wv;
                __builtin_memcpy(&wv, row + base, 4);
#pragma unroll
                for (int e = 0; e < 4; e++) { p[e] = l0_unorm8((wv >> sa[e]) & 0xffu); q[e] = l0_unorm8((wv >> sb[e]) & 0xffu); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) { p[e] = l0_unorm8(row[ia[e]]); q[e] = l0_unorm8(row[ib[e]]); }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) r[e] = l0_lerp(p[e], q[e], al[e]);
    };

    int cj0 = -1, cj1 = -1;                          // texel rows whose lerp_x is cached in r0 / r1
    float r0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, r1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int rr = 0; rr < UP_ROWS; rr++) {
        const int y = y0 + rr;
        if (y >= a.H) break;
        int j0; float be;
        l0_axis(((float)y + a.shift) / (float)a.H, a.h, j0, be);
        // wave uniform (every lane of a wave has the same y): scalar compares and branches
        const int ja = __builtin_amdgcn_readfirstlane(psx_clampi(j0, 0, a.h - 1));
        const int jb = __builtin_amdgcn_readfirstlane(psx_clampi(j0 + 1, 0, a.h - 1));
        float ra[4], rb[4];
        if (ja == cj0)      { for (int e = 0; e < 4; e++) ra[e] = r0[e]; }
        else if (ja == cj1) { for (int e = 0; e < 4; e++) ra[e] = r1[e]; }
        else                lerp_row(ja, ra);
        if (jb == ja)       { for (int e = 0; e < 4; e++) rb[e] = ra[e]; }
        else if (jb == cj1) { for (int e = 0; e < 4; e++) rb[e] = r1[e]; }
        else if (jb == cj0) { for (int e = 0; e < 4; e++) rb[e] = r0[e]; }
        else                lerp_row(jb, rb);
        cj0 = ja; cj1 = jb;
#pragma unroll
        for (int e = 0; e < 4; e++) { r0[e] = ra[e]; r1[e] = rb[e]; }
        // lerp_y( lerp_x(T[j0]), lerp_x(T[j0+1]) ): same operations, same order as the oracle's texture model
        *reinterpret_cast<float4*>(a.dst + (size_t)y * a.pitch + cq * 4) =
            make_float4(l0_lerp(ra[0], rb[0], be), l0_lerp(ra[1], rb[1], be),
                        l0_lerp(ra[2], rb[2], be), l0_lerp(ra[3], rb[3], be));
    }
}

// ---------------------------------------------------------------------------------------------
// Octave 0, level 0 at the default x2 upsampling, FUSED (round 3): k_upscale materialised U (33 MB written,
// 33 MB read back) only so that k_blur<R, true> could stage it; here the marching strip's staging phase reads the
// input texels themselves (1/16 of the bytes for a u8 image) and the texture fetch is split along its two lerps:
//   staging : Tx[j][X] = lerp_x(T[j][i0(X)], T[j][i0(X)+1], alpha_X) for the ~18 texel rows j a step touches,
//             once per texel row (two output rows share each at x2), into an LDS buffer of NTX rows;
//   H pass  : U(X, y) = lerp_y(Tx[j0(y)][X], Tx[j0(y)+1][X], beta_y) on the thread's two windows, then the
//             "dd" taps exactly as k_blur<R, true> (pairs outermost-in, centre, x255).
// Same operations in the same order as k_upscale + k_blur<R, true> (l0_axis, l0_unorm8, l0_lerp are shared), so the
// plane is bit-identical; U for columns outside [0, W) is defined by clamped TEXELS, which the staging computes
// directly -- no padded plane, no edge-strip special case.  Vertical pass, ring, deferred stores: as k_blur.
// Used when upscale_factor == 1 and the image is at least 4 texels wide; every other configuration keeps
// k_upscale + k_blur<R, true>.
// ---------------------------------------------------------------------------------------------
struct L0Args {
    const void* img; int w, h;
    float* dst; int W, H, pitch;
    float shift;
    int nstrips, chunk_rows;
    PsxTaps taps;      // dd horizontal
    PsxTaps taps_v;    // inc[0] vertical
    // GaussMode VLFeat_Relative (k_level0_x2<.., VNP > 0>): the vertical pass is the interpolated one (blur_interp.h)
    float vg0, vmul[4], voff[4];
    int vforce;        // test switch (POPSIFT_INTERP_LITERAL=1): every weight from its coordinate
};

template <int R>
struct GeomL0 {
    static constexpr int HALO = (R + 3) & ~3;
    static constexpr int SW   = TW + 2 * HALO;
    static constexpr int SW4  = SW / 4;
    static constexpr int SWA  = 4 * (SW4 | 1);
    static constexpr int NTX  = BR / 2 + 3;                // texel rows a step of BR output rows touches at x2 (+ rounding slack)
    static constexpr int NIT  = (NTX * SW4 + NT - 1) / NT; // staging items (texel row, column quad) per thread
    static constexpr int RING = (BR + 2 * R <= 64) ? 64 : 128;
    static constexpr int VWIN = 4 + 2 * R;
    static constexpr int MIRROR = VWIN - 1;
    static constexpr int RS   = TW + 4;
};

__device__ __forceinline__ float sel4(float a, float b, float c, float d, int i)
{
    const float lo = i & 1 ? b : a, hi = i & 1 ? d : c;
    return i & 2 ? hi : lo;
}

template <int R, bool ISFLOAT>
__device__ __forceinline__ void level0_body(const L0Args& a, const int lid)
{
    using G = GeomL0<R>;
    constexpr int HALO = G::HALO, SW4 = G::SW4, SWA = G::SWA, NTX = G::NTX, NIT = G::NIT, RING = G::RING;
    constexpr int VWIN = G::VWIN, MIRROR = G::MIRROR, RS = G::RS;
    __shared__ __attribute__((aligned(16))) float s_tx[NTX * SWA];
    __shared__ __attribute__((aligned(16))) float s_ring[(RING + MIRROR) * RS];

    const int t     = threadIdx.x;
    const int strip = lid % a.nstrips;
    const int chunk = lid / a.nstrips;
    const int x0    = strip * TW;
    const int Y0    = chunk * a.chunk_rows;
    const int Y1    = min(Y0 + a.chunk_rows, a.H);
    const int nsteps = (Y1 - Y0 + 2 * R + BR - 1) / BR;
    const float fW = (float)a.W, fH = (float)a.H;

    // ---- staging items of this thread: (texel row slot, column quad); the column side is step invariant ----
    int it_row[NIT], it_lds[NIT], it_base[NIT];
    int it_sa[NIT][4], it_sb[NIT][4];                   // texel of output e within the 4-texel window: index (float) / bit shift (u8)
    float it_al[NIT][4];
    bool it_on[NIT];
#pragma unroll
    for (int j = 0; j < NIT; j++) {
        const int idx = t + j * NT;
        const int row = idx / SW4, cq = idx - row * SW4;
        it_on[j]  = row < NTX;
        it_row[j] = row;
        it_lds[j] = row * SWA + cq * 4;
        int ia[4], ib[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            int i0;
            l0_axis(((float)(x0 - HALO + cq * 4 + e) + a.shift) / fW, a.w, i0, it_al[j][e]);
            ia[e] = psx_clampi(i0, 0, a.w - 1); ib[e] = psx_clampi(i0 + 1, 0, a.w - 1);
        }
        // the 8 texel columns of 4 adjacent outputs lie within 4 consecutive texels at x2 (host checks w >= 4)
        const int base = min(ia[0], a.w - 4);
        it_base[j] = base;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int da = psx_clampi(ia[e] - base, 0, 3), db = psx_clampi(ib[e] - base, 0, 3);
            it_sa[j][e] = ISFLOAT ? da : da * 8; it_sb[j][e] = ISFLOAT ? db : db * 8;
        }
    }
    // texel row of output row y (clamped like the plane rows of k_blur): j0, j0 + 1 clamped, and the 1.8 weight
    auto row_axis = [&](int y, int& ja, int& jb, float& be) {
        int j0;
        l0_axis(((float)psx_clampi(y, 0, a.H - 1) + a.shift) / fH, a.h, j0, be);
        ja = psx_clampi(j0, 0, a.h - 1); jb = psx_clampi(j0 + 1, 0, a.h - 1);
    };
    auto step_jlo = [&](int k) { int ja, jb; float be; row_axis(Y0 - R + k * BR, ja, jb, be); return __builtin_amdgcn_readfirstlane(ja); };

    // ---- horizontal / vertical geometry as blur_body ----
    int h_row, h_seg;
    {
        const int blk = (t & 31) >> 2;
        const int rq = (0x21120330 >> (4 * blk)) & 3;
        const int sh = (0xCC >> blk) & 1;
        h_row = (t >> 6) * 8 + ((t >> 5) & 1) * 4 + rq;
        h_seg = sh * 4 + (t & 3);
    }
    const int v_pp = t & 31, v_rg = t >> 5;
    const int v_x  = x0 + 2 * v_pp;
    const unsigned v_doff = (unsigned)((v_rg * 4) * a.pitch + v_x) * 4u;
    const bool v_xok = v_x < a.W, v_pair = v_x + 1 < a.W;

    typedef typename std::conditional<ISFLOAT, v4f, unsigned>::type texel4;
    texel4 pre[NIT];
    auto issue = [&](int k) {
        const int jlo = step_jlo(k);
#pragma unroll
        for (int j = 0; j < NIT; j++) {
            if (it_on[j]) {
                const int jr = psx_clampi(jlo + it_row[j], 0, a.h - 1);
                if (ISFLOAT) __builtin_memcpy(&pre[j], static_cast<const float*>(a.img) + (size_t)jr * a.w + it_base[j], 16);
                else         __builtin_memcpy(&pre[j], static_cast<const uint8_t*>(a.img) + (size_t)jr * a.w + it_base[j], 4);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NIT; j++) {
            if (it_on[j]) {
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float p, q;
                    if constexpr (ISFLOAT) {
                        p = sel4(pre[j].x, pre[j].y, pre[j].z, pre[j].w, it_sa[j][e]);
                        q = sel4(pre[j].x, pre[j].y, pre[j].z, pre[j].w, it_sb[j][e]);
                    } else {
                        p = l0_unorm8((pre[j] >> it_sa[j][e]) & 0xffu);
                        q = l0_unorm8((pre[j] >> it_sb[j][e]) & 0xffu);
                    }
                    r[e] = l0_lerp(p, q, it_al[j][e]);
                }
                *reinterpret_cast<float4*>(&s_tx[it_lds[j]]) = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
    };

    v2f pend[4];
    auto flush = [&](const int kk) {
        const int r_out0 = Y0 + kk * BR - 2 * R + v_rg * 4;
        char* drow = reinterpret_cast<char*>(a.dst + (ptrdiff_t)(Y0 - 2 * R + kk * BR) * a.pitch);
        // workgroup uniform fast path, as in blur_body: all 32 rows of the step inside the chunk, a full strip
        if (kk * BR >= 2 * R && Y0 + (kk + 1) * BR - 2 * R <= Y1 && x0 + TW <= a.W) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                char* di = (drow + (size_t)i * a.pitch * 4) + v_doff;
                unsigned long long bits; __builtin_memcpy(&bits, &pend[i], 8);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r_out = r_out0 + i;
            if (r_out >= Y0 && r_out < Y1 && v_xok) {
                char* di = (drow + (size_t)i * a.pitch * 4) + v_doff;
                if (v_pair) {
                    unsigned long long bits; __builtin_memcpy(&bits, &pend[i], 8);
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else *reinterpret_cast<float*>(di) = pend[i].x;
            }
        }
    };

    issue(0);
    for (int k = 0; k < nsteps; k++) {
        commit();
        flush(k - 1);
        const int jlo = step_jlo(k);
        __syncthreads();
        if (k + 1 < nsteps) issue(k + 1);

        // ---- horizontal: lerp_y of the two texel-row windows, then the dd taps ----
        {
            int ja, jb; float be;
            row_axis(Y0 - R + k * BR + h_row, ja, jb, be);
            const int oa = psx_clampi(ja - jlo, 0, NTX - 1), ob = psx_clampi(jb - jlo, 0, NTX - 1);
            const LDS_AS float* pa = (const LDS_AS float*)&s_tx[oa * SWA + h_seg * 8];
            const LDS_AS float* pb = (const LDS_AS float*)&s_tx[ob * SWA + h_seg * 8];
            float win[8 + 2 * HALO];
#pragma unroll
            for (int q = 0; q < (8 + 2 * HALO) / 4; q++) {
                const v4f va = ((const volatile LDS_AS v4f*)pa)[q];
                const v4f vb = ((const volatile LDS_AS v4f*)pb)[q];
                win[4 * q + 0] = l0_lerp(va.x, vb.x, be); win[4 * q + 1] = l0_lerp(va.y, vb.y, be);
                win[4 * q + 2] = l0_lerp(va.z, vb.z, be); win[4 * q + 3] = l0_lerp(va.w, vb.w, be);
            }
            float out[8];
            hfilter8_km<R, HALO, true>(win, a.taps, out);
            const int slot = (k * BR + h_row) & (RING - 1);
            float* rp = &s_ring[slot * RS + h_seg * 8];
            reinterpret_cast<float4*>(rp)[0] = make_float4(out[0], out[1], out[2], out[3]);
            reinterpret_cast<float4*>(rp)[1] = make_float4(out[4], out[5], out[6], out[7]);
            if (slot < MIRROR) {
                reinterpret_cast<float4*>(rp + RING * RS)[0] = make_float4(out[0], out[1], out[2], out[3]);
                reinterpret_cast<float4*>(rp + RING * RS)[1] = make_float4(out[4], out[5], out[6], out[7]);
            }
        }
        __syncthreads();

        // ---- vertical ----
        {
            const int rel0 = k * BR - 2 * R + v_rg * 4;
            const int r_out0 = Y0 + rel0;
            if (r_out0 + 3 >= Y0 && r_out0 < Y1) {
                const LDS_AS float* vp = (const LDS_AS float*)&s_ring[(rel0 & (RING - 1)) * RS + 2 * v_pp];
                v2f v[VWIN];
#pragma unroll
                for (int j = 0; j < VWIN; j++) v[j] = *(const volatile LDS_AS v2f*)(vp + j * RS);
                v2f o[4];
                vfilter2x4_km<R>(v, a.taps_v, o);
                asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
                for (int i = 0; i < 4; i++) pend[i] = o[i];
            }
        }
    }
    flush(nsteps - 1);
}

template <int R, bool ISFLOAT>
__global__ __launch_bounds__(NT, 4) void k_level0_fused(L0Args a)
{
    level0_body<R, ISFLOAT>(a, xcd_remap(blockIdx.x, gridDim.x));
}

// ---------------------------------------------------------------------------------------------
// k_level0_x2 (round 4): level 0 at the x2 upsampling, specialised on what x2 means.  With W = 2w the texture
// coordinate of output column X is X/2 (sampling shift 1.0: PopSift, VLFeat) or X/2 - 1/4 (shift 0.5: OpenCV), so the
// 1.8 fixed-point filter weights are CONSTANTS: {0, 1/2} or {3/4, 1/4} alternating with the column's parity, and the
// same for rows.  k_level0_fused computed index and weight per thread with l0_axis, converted 8 texels for 4 outputs,
// ran 12 VALU for their 4 lerp_x and a generic lerp_y on all 24 values of every horizontal window -- 6.9 M wave
// instructions per 1080p frame against 2.7 M for k_blur<5> with the same taps.  Here a thread owns a 4 x 4 block of
// outputs (an aligned group of 4 rows x a quad of columns): 3 x 3 (shift 1) or 4 x 4 (shift 1/2) texels, each converted
// once, lerp(p, q, 0) is p (+ 0 for float images, which keeps fma's -0 -> +0), lerp(p, q, 1/2) is one multiply and one
// fma, and the block goes to LDS as finished rows of U: the H pass reads ONE window and runs the "dd" taps exactly as
// k_blur<R, true>.  Same operations on the same operands as the texture model (lerp_x, then lerp_y; weights that the
// general formula also arrives at, including its X-even knife edge where it lands on (i0 - 1, alpha = 1): the value is
// T[i0] either way), so the plane is bit-identical (the GPU suite runs through it in every SiftMode).
// Rows outside the plane (first / last chunk) replicate U(0) / U(H-1): patched in LDS on those steps only.
// ---------------------------------------------------------------------------------------------
// VNP > 0 (GaussMode VLFeat_Relative, level 0 of octave 0: normalizedSource::horiz + absoluteSourceInterpolated::vert,
// s_pyramid_build.cu:505-512): the same staging and "dd" H pass, the vertical pass with VNP interpolated tap pairs
// (psx_vinterp2x4, blur_interp.h); R >= 2 VNP, the H taps beyond their span are zero.
template <int R, bool ISFLOAT, bool SHIFT1, int VNP = 0>
__global__ __launch_bounds__(NT, 4) void k_level0_x2(L0Args a)
{
    using G = Geom2<R>;
    constexpr int HALO = G::HALO, SW4 = G::SW4, SWA = G::SWA, RING = G::RING, VWIN = G::VWIN, MIRROR = G::MIRROR, RS = G::RS;
    constexpr int NRW = SHIFT1 ? 3 : 4;                  // texel rows / columns a 4 x 4 block of outputs needs
    constexpr int NGRP = BR / 4 + 1;                     // aligned row groups that intersect a step of BR rows
    static_assert(NGRP * SW4 <= NT, "one block per thread");
    static_assert(2 * VNP <= R, "the interpolated vertical pass reaches 2 VNP rows");
    __shared__ __attribute__((aligned(16))) float s_u[BR * SWA];
    __shared__ __attribute__((aligned(16))) float s_ring[(RING + MIRROR) * RS];
    __shared__ __attribute__((aligned(16))) v4f s_vtab[VNP > 0 ? VNP : 1];   // VNP > 0: the V pass's weight table (blur_interp.h)
    __shared__ __attribute__((aligned(16))) v4f s_htab[VNP > 0 ? VNP : 1];   //          (the survey's unused column half)
    __shared__ unsigned s_vmask[2];

    const int lid   = xcd_remap(blockIdx.x, gridDim.x);
    const int t     = threadIdx.x;
    const int strip = lid % a.nstrips;
    const int chunk = lid / a.nstrips;
    const int x0    = strip * TW;
    const int Y0    = chunk * a.chunk_rows;
    const int Y1    = min(Y0 + a.chunk_rows, a.H);
    const int nsteps = (Y1 - Y0 + 2 * R + BR - 1) / BR;

    // ---- this thread's block: row group slot gs, column quad cq; the column side is step invariant ----
    const int gs = t / SW4, cq = t - gs * SW4;
    const bool active = gs < NGRP;
    const int X0c = x0 - HALO + 4 * cq;                                  // first output column of the quad (a multiple of 4)
    const int cfirst = (X0c >> 1) - (SHIFT1 ? 0 : 1);                    // first texel column the quad reads
    const int cbase = psx_clampi(cfirst, 0, a.w - 4);                    // position of the 4-texel load (host: w >= 4)
    int csel[NRW];                                                        // texel i of the quad within the loaded four (clamped texels)
#pragma unroll
    for (int i = 0; i < NRW; i++) csel[i] = psx_clampi(cfirst + i, 0, a.w - 1) - cbase;

    // ---- horizontal / vertical geometry as blur_body ----
    int h_row, h_seg;
    {
        const int blk = (t & 31) >> 2;
        const int rq = (0x21120330 >> (4 * blk)) & 3;
        const int sh = (0xCC >> blk) & 1;
        h_row = (t >> 6) * 8 + ((t >> 5) & 1) * 4 + rq;
        h_seg = sh * 4 + (t & 3);
    }
    const LDS_AS float* h_src = (const LDS_AS float*)&s_u[h_row * SWA + h_seg * 8];
    const int v_pp = t & 31, v_rg = t >> 5;
    const int v_x  = x0 + 2 * v_pp;
    const unsigned v_doff = (unsigned)((v_rg * 4) * a.pitch + v_x) * 4u;
    const bool v_xok = v_x < a.W, v_pair = v_x + 1 < a.W;

    typedef typename std::conditional<ISFLOAT, v4f, unsigned>::type texel4;
    texel4 pre[NRW];
    auto issue = [&](int k) {
        if (!active) return;
        const int ybase = Y0 - R + k * BR;
        const int g = (ybase >> 2) + gs;                                 // rows 4g .. 4g+3
        const int rfirst = 2 * g - (SHIFT1 ? 0 : 1);
#pragma unroll
        for (int i = 0; i < NRW; i++) {
            const int jr = psx_clampi(rfirst + i, 0, a.h - 1);
            if (ISFLOAT) __builtin_memcpy(&pre[i], static_cast<const float*>(a.img) + (size_t)jr * a.w + cbase, 16);
            else         __builtin_memcpy(&pre[i], static_cast<const uint8_t*>(a.img) + (size_t)jr * a.w + cbase, 4);
        }
    };
    // lerp with the constant weights of the x2 grid; w0 = 0 / 1/2 (SHIFT1) or 3/4 / 1/4
    auto half_ = [](float p, float q) { return fmaf(0.5f, q, 0.5f * p); };                      // l0_lerp(p, q, 1/2)
    auto q34_  = [](float p, float q) { return fmaf(0.75f, q, 0.25f * p); };                    // l0_lerp(p, q, 3/4)
    auto q14_  = [](float p, float q) { return fmaf(0.25f, q, 0.75f * p); };                    // l0_lerp(p, q, 1/4)
    auto same_ = [](float p) { return ISFLOAT ? p + 0.0f : p; };                                 // l0_lerp(p, q, 0) for finite q
    auto commit = [&](int k) {
        if (!active) return;
        const int ybase = Y0 - R + k * BR;
        const int g = (ybase >> 2) + gs;
        float ux[NRW][4];                                                // lerp_x of the texel rows at the quad's 4 columns
#pragma unroll
        for (int i = 0; i < NRW; i++) {
            float c[NRW];
#pragma unroll
            for (int q = 0; q < NRW; q++) {
                if constexpr (ISFLOAT) c[q] = sel4(pre[i].x, pre[i].y, pre[i].z, pre[i].w, csel[q]);
                else                   c[q] = l0_unorm8((pre[i] >> (8 * csel[q])) & 0xffu);
            }
            if constexpr (SHIFT1) { ux[i][0] = same_(c[0]); ux[i][1] = half_(c[0], c[1]); ux[i][2] = same_(c[1]); ux[i][3] = half_(c[1], c[2]); }
            else                  { ux[i][0] = q34_(c[0], c[1]); ux[i][1] = q14_(c[1], c[2]); ux[i][2] = q34_(c[1], c[2]); ux[i][3] = q14_(c[2], c[3]); }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = 4 * g + q - ybase;                             // row of the step
            if (r < 0 || r >= BR) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if constexpr (SHIFT1) o[e] = q == 0 ? same_(ux[0][e]) : q == 1 ? half_(ux[0][e], ux[1][e]) : q == 2 ? same_(ux[1][e]) : half_(ux[1][e], ux[2][e]);
                else                  o[e] = q == 0 ? q34_(ux[0][e], ux[1][e]) : q == 1 ? q14_(ux[1][e], ux[2][e]) : q == 2 ? q34_(ux[1][e], ux[2][e]) : q14_(ux[2][e], ux[3][e]);
            }
            *reinterpret_cast<float4*>(&s_u[r * SWA + cq * 4]) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };

    v2f pend[4];
    auto flush = [&](const int kk) {
        const int r_out0 = Y0 + kk * BR - 2 * R + v_rg * 4;
        char* drow = reinterpret_cast<char*>(a.dst + (ptrdiff_t)(Y0 - 2 * R + kk * BR) * a.pitch);
        // workgroup uniform fast path, as in blur_body: all 32 rows of the step inside the chunk, a full strip
        if (kk * BR >= 2 * R && Y0 + (kk + 1) * BR - 2 * R <= Y1 && x0 + TW <= a.W) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                char* di = (drow + (size_t)i * a.pitch * 4) + v_doff;
                unsigned long long bits; __builtin_memcpy(&bits, &pend[i], 8);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r_out = r_out0 + i;
            if (r_out >= Y0 && r_out < Y1 && v_xok) {
                char* di = (drow + (size_t)i * a.pitch * 4) + v_doff;
                if (v_pair) {
                    unsigned long long bits; __builtin_memcpy(&bits, &pend[i], 8);
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(di), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else *reinterpret_cast<float*>(di) = pend[i].x;
            }
        }
    };

#ifdef PSX_PHASE_TIMING
    long long tacc[5] = {0, 0, 0, 0, 0};
    long long tprev = clock64();
    const long long tstart = tprev;
#endif
    issue(0);
    unsigned vmask = 0u;
    if constexpr (VNP > 0) {
        if (t < 2) s_vmask[t] = 0u;
        if (t == 0) {
#pragma unroll
            for (int p = 0; p < VNP; p++) { s_vtab[p] = (v4f){0.0f, 0.0f, a.vmul[p], a.voff[p]}; s_htab[p] = (v4f){0.0f, 0.0f, a.vmul[p], a.voff[p]}; }
        }
        __syncthreads();
        psx_interp_survey(VNP, t, NT, x0, 0, Y0, Y1 - Y0, (LDS_AS v4f*)s_htab, (LDS_AS v4f*)s_vtab, s_vmask);
        __syncthreads();
        vmask = a.vforce ? ~0u : __builtin_amdgcn_readfirstlane(s_vmask[1]);
    }
    for (int k = 0; k < nsteps; k++) {
        commit(k);
        flush(k - 1);
        const int ybase = Y0 - R + k * BR;
        if (ybase < 0 || ybase + BR > a.H) {
            // first / last chunk: the rows of the step outside the plane are copies of U(0) / U(H - 1) (the vertical
            // filter clamps its row index), which the blocks above have just produced inside this step
            __syncthreads();
            for (int idx = t; idx < BR * SW4; idx += NT) {
                const int r = idx / SW4, c4 = idx - r * SW4;
                const int y = ybase + r;
                if (y < 0 || y > a.H - 1) {
                    const int rc = psx_clampi(y, 0, a.H - 1) - ybase;
                    if (rc >= 0 && rc < BR)
                        *reinterpret_cast<float4*>(&s_u[r * SWA + c4 * 4]) = *reinterpret_cast<const float4*>(&s_u[rc * SWA + c4 * 4]);
                }
            }
        }
        BSTAMP(0);
        __syncthreads();
        BSTAMP(1);
        if (k + 1 < nsteps) issue(k + 1);

        // ---- horizontal: the dd taps over one window of U ----
        {
            float win[8 + 2 * HALO];
#pragma unroll
            for (int q = 0; q < (8 + 2 * HALO) / 4; q++) {
                const v4f v = ((const volatile LDS_AS v4f*)h_src)[q];
                win[4 * q + 0] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
            }
            float out[8];
            hfilter8_km<R, HALO, true>(win, a.taps, out);
            const int slot = (k * BR + h_row) & (RING - 1);
            float* rp = &s_ring[slot * RS + h_seg * 8];
            reinterpret_cast<float4*>(rp)[0] = make_float4(out[0], out[1], out[2], out[3]);
            reinterpret_cast<float4*>(rp)[1] = make_float4(out[4], out[5], out[6], out[7]);
            if (slot < MIRROR) {
                reinterpret_cast<float4*>(rp + RING * RS)[0] = make_float4(out[0], out[1], out[2], out[3]);
                reinterpret_cast<float4*>(rp + RING * RS)[1] = make_float4(out[4], out[5], out[6], out[7]);
            }
        }
        BSTAMP(2);
        __syncthreads();
        BSTAMP(3);

        // ---- vertical ----
        {
            const int rel0 = k * BR - 2 * R + v_rg * 4;
            const int r_out0 = Y0 + rel0;
            if (r_out0 + 3 >= Y0 && r_out0 < Y1) {
                const LDS_AS float* vp = (const LDS_AS float*)&s_ring[(rel0 & (RING - 1)) * RS + 2 * v_pp];
                v2f o[4];
                if constexpr (VNP > 0) psx_vinterp2x4<VNP, RS>(vp + (R - 2 * VNP) * RS, (const LDS_AS v4f*)s_vtab, vmask, a.vg0, r_out0, o);
                else {
                    v2f v[VWIN];
#pragma unroll
                    for (int j = 0; j < VWIN; j++) v[j] = *(const volatile LDS_AS v2f*)(vp + j * RS);
                    vfilter2x4_km<R>(v, a.taps_v, o);
                }
                asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
                for (int i = 0; i < 4; i++) pend[i] = o[i];
            }
        }
        BSTAMP(4);
    }
    flush(nsteps - 1);
#ifdef PSX_PHASE_TIMING
    if (threadIdx.x == 0 && g_l0_dbg) {
        for (int q = 0; q < 5; q++) g_l0_dbg[blockIdx.x * 8 + q] = tacc[q];
        g_l0_dbg[blockIdx.x * 8 + 5] = clock64() - tstart; g_l0_dbg[blockIdx.x * 8 + 6] = nsteps;
    }
#endif
}

// make_dog (s_pyramid_build.cu:74-92) for one level pair; debug/dump use only
__global__ void k_dog(const float* a, const float* b, float* d, int W, int H, int pitch)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * pitch + x;
    d[i] = b[i] - a[i];
}

// Tuning switches for A/B measurements on the GPU (read once): POPSIFT_BLUR_STEPS = marching steps per chunk
// on large planes (default 5), POPSIFT_BLUR_DEFER=0 stores the vertical results at once (the round-1 kernel).
struct BlurTuning { int steps; bool defer; int dma; int dma_steps; };
inline const BlurTuning& blur_tuning()
{
    static const BlurTuning t = [] {
        BlurTuning v{5, true, 0, 0};
        // POPSIFT_BLUR_DMA: 0 = register-staged k_blur, 2 / 3 = LDS-DMA staging with 2 / 3 stage buffers (k_blur_dma)
        // (a dedicated loader wave, H and V on different waves and a one-shot tile kernel for the small planes were built
        // on top of it in round 3, measured slower and removed again: profiles/r03_blur_staging_experiments.txt, git history)
        // 1 = one stage buffer (the next batch goes out after the H pass): the least LDS, up to 5 workgroups per CU
        if (const char* e = getenv("POPSIFT_BLUR_DMA")) { const int n = atoi(e); if (n >= 0 && n <= 3) v.dma = n; }
        if (const char* e = getenv("POPSIFT_BLUR_DMA_STEPS")) { const int n = atoi(e); if (n >= 2 && n <= 64) v.dma_steps = n; }
        if (const char* e = getenv("POPSIFT_BLUR_STEPS")) { const int n = atoi(e); if (n >= 2 && n <= 64) v.steps = n; }
        if (const char* e = getenv("POPSIFT_BLUR_DEFER")) v.defer = e[0] != '0';
        return v;
    }();
    return t;
}

inline void chunking(int W, int H, int R, int& chunk_rows, int& nchunks)
{
    // S marching steps per chunk: the 2R warm-up rows cost ~2R/(S*BR) extra horizontal work, but a
    // chunk is a serial chain of S steps.  Large planes take S=5; small octaves trade efficiency
    // for more, shorter workgroups (they are latency bound, not bandwidth bound).
    const int nstrips = (W + TW - 1) / TW;
    int S = blur_tuning().steps;
    // planes that fill the chip four times over even with longer chunks take 7 steps: less warm-up work per output
    // row (8192 x 8192 planes: 0.574 -> 0.593 of 8 TB/s; at 3840 x 2160 7 steps would leave 660 workgroups: slower)
    if (S == 5 && nstrips * ((H + (7 * BR - 2 * R) - 1) / (7 * BR - 2 * R)) >= 4096) S = 7;
    for (; S > 2; S--) {
        const int cr = S * BR - 2 * R;
        if (cr >= BR && nstrips * ((H + cr - 1) / cr) >= 384) break;       // 256, 512, 768, 1024 measured: no better
    }
    // POPSIFT_BLUR_ONESTEP=1 (measurement switch, round 6): the small octaves as one-step chunks where one step still yields
    // >= 12 rows (what took the fixed-span and interpolated kernels' small launches from ~9 to ~5.5 us)
    static const bool onestep = [] { const char* e = getenv("POPSIFT_BLUR_ONESTEP"); return e != nullptr && e[0] == '1'; }();
    if (onestep && S == 2 && BR - 2 * R >= 12 && nstrips * ((H + (2 * BR - 2 * R) - 1) / (2 * BR - 2 * R)) < 256) S = 1;
    int cr = S * BR - 2 * R;
    if (cr < BR / 2 && S > 1) cr = BR / 2;
    if (cr > H) cr = H;
    chunk_rows = cr;
    nchunks = (H + cr - 1) / cr;
}

// fills the arguments of one plane-to-plane blur; returns its number of workgroups
template <int R>
int fill_job(BlurArgs& a, const PsxBlurJob& j)
{
    a.src = j.src; a.dst = j.dst; a.half_dst = j.half_dst;
    a.W = j.W; a.H = j.H; a.pitch = j.pitch; a.half_pitch = j.half_pitch;
    a.src_pitch = j.pitch; a.src_xoff = 0; a.src_width = j.W;
    a.nstrips = (j.W + TW - 1) / TW;
    int nchunks;
    chunking(j.W, j.H, R, a.chunk_rows, nchunks);
    a.taps = j.taps; a.taps_v = j.taps;
#ifdef PSX_PHASE_TIMING
    { const char* e = getenv("POPSIFT_BLUR_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
    return a.nstrips * nchunks;
}

inline int device_cus()
{
    static const int n = [] { int d = 0, c = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256; return c; }();
    return n;
}

template <int R>
hipError_t launch_blur2_r(const PsxBlurJob& ja, const PsxBlurJob& jb, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    BlurArgs a, b;
    const int na = fill_job<R>(a, ja), nb = fill_job<R>(b, jb);
    const dim3 grid(na + nb), block(NT);
    if (ev0 != nullptr || ev1 != nullptr) hipExtLaunchKernelGGL((k_blur2<R>), grid, block, 0, s, ev0, ev1, 0, a, b, na);
    else                                  hipLaunchKernelGGL((k_blur2<R>), grid, block, 0, s, a, b, na);
    return hipGetLastError();
}

// ring rows of the LDS-DMA variant: the smallest multiple of 16 that holds a step's BR + 2R rows
constexpr int dma_ring(int R) { return (BR + 2 * R + 15) & ~15; }

// Chunking for k_blur_dma.  A plane that fits one round of resident workgroups should be exactly one round (a second,
// nearly empty round doubles the launch): take the smallest S >= 5 steps per chunk that fits; planes of several rounds
// take 7 steps (less warm-up work per output row), small planes fewer (they are latency chains).
inline void chunking_dma(int W, int H, int R, int slots, int& chunk_rows, int& nchunks)
{
    const int nstrips = (W + TW - 1) / TW;
    auto nwg = [&](int S) { const int cr = S * BR - 2 * R; return nstrips * ((H + cr - 1) / cr); };
    int S = blur_tuning().dma_steps;
    if (S == 0) {
        S = 5;
        if (nwg(5) > slots) {
            int fit = 0;
            for (int q = 6; q <= 10; q++) if (nwg(q) <= slots) { fit = q; break; }
            S = fit ? fit : 7;
        } else {
            for (; S > 2; S--) { const int cr = S * BR - 2 * R; if (cr >= BR && nwg(S) >= 384) break; }
        }
    }
    int cr = S * BR - 2 * R;
    if (cr < BR / 2) cr = BR / 2;
    if (cr > H) cr = H;
    chunk_rows = cr;
    nchunks = (H + cr - 1) / cr;
}

template <int R, bool LEVEL0, int NBUF>
void launch_dma(BlurArgs& a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    constexpr int RING = dma_ring(R);
    int nchunks;
    chunking_dma(a.W, a.H, R, dma_wg_per_cu<R, NBUF, RING>() * device_cus(), a.chunk_rows, nchunks);
    const dim3 grid(a.nstrips * nchunks), block(NT);
    if (ev0 != nullptr || ev1 != nullptr) hipExtLaunchKernelGGL((k_blur_dma<R, LEVEL0, NBUF, RING>), grid, block, 0, s, ev0, ev1, 0, a);
    else                                  hipLaunchKernelGGL((k_blur_dma<R, LEVEL0, NBUF, RING>), grid, block, 0, s, a);
}

template <int R>
hipError_t launch_blur_r(const float* src, float* dst, int W, int H, int pitch, const PsxTaps& taps,
                         float* half_dst, int half_pitch, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    BlurArgs a;
    PsxBlurJob j;
    j.src = src; j.dst = dst; j.half_dst = half_dst; j.W = W; j.H = H; j.pitch = pitch; j.half_pitch = half_pitch;
    j.taps = taps; j.span = R + 1;
    const dim3 grid(fill_job<R>(a, j)), block(NT);
    if constexpr (R <= 13) {
        const int dma = blur_tuning().dma;
        if (dma == 1) { launch_dma<R, false, 1>(a, s, ev0, ev1); return hipGetLastError(); }
        if (dma == 2) { launch_dma<R, false, 2>(a, s, ev0, ev1); return hipGetLastError(); }
        if (dma == 3) { launch_dma<R, false, 3>(a, s, ev0, ev1); return hipGetLastError(); }
    }
    const bool ext = ev0 != nullptr || ev1 != nullptr;       // kernel begin / end timestamps of THIS dispatch (what rocprofv3 --kernel-trace reports)
    if (blur_tuning().defer) {
        // POPSIFT_BLUR_LDS_PAD (measurement switch): extra dynamic LDS per workgroup, i.e. fewer resident k_blur
        // workgroups per CU, leaving room for another stream's kernels on the same CUs
        static const unsigned pad = [] { const char* e = getenv("POPSIFT_BLUR_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
        if (ext) hipExtLaunchKernelGGL((k_blur<R, false, true>), grid, block, pad, s, ev0, ev1, 0, a);
        else     hipLaunchKernelGGL((k_blur<R, false, true>), grid, block, pad, s, a);
    } else {
        if (ext) hipExtLaunchKernelGGL((k_blur<R, false, false>), grid, block, 0, s, ev0, ev1, 0, a);
        else     hipLaunchKernelGGL((k_blur<R, false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

// POPSIFT_LEVEL0_FUSED=0 keeps k_upscale + k_blur<R, true> for every configuration
inline bool level0_fused_enabled()
{
    static const bool on = [] { const char* e = getenv("POPSIFT_LEVEL0_FUSED"); return !(e != nullptr && e[0] == '0'); }();
    return on;
}

template <int R>
hipError_t launch_level0_r(const PsxLevel0Args& h, hipStream_t s)
{
    if constexpr (R <= 8) {
        // the default x2 upsampling: W = 2w, H = 2h exactly, with the sampling shift of a SiftMode (1.0 PopSift / VLFeat,
        // 0.5 OpenCV).  The kernel's staging assumes what holds for exactly these: the 8 texel columns of 4 adjacent
        // outputs lie within 4 consecutive texels and a 32-row step touches at most NTX texel rows (checked for both
        // shifts incl. the al = 1 knife edge); any other shift or scale keeps k_upscale, which tests the condition itself
        if (level0_fused_enabled() && h.W == 2 * h.w && h.H == 2 * h.h && h.w >= 4 && (h.shift == 1.0f || h.shift == 0.5f)) {
            L0Args f;
            f.img = h.img; f.w = h.w; f.h = h.h; f.dst = h.dst; f.W = h.W; f.H = h.H; f.pitch = h.pitch; f.shift = h.shift;
            f.nstrips = (h.W + TW - 1) / TW;
            int nchunks;
            chunking(h.W, h.H, R, f.chunk_rows, nchunks);
            f.taps = h.taps_h; f.taps_v = h.taps_v;
            const dim3 grid(f.nstrips * nchunks), block(NT);
            // POPSIFT_LEVEL0_X2=0: round 3's k_level0_fused (general weights) instead of the x2-specialised kernel
            static const bool x2 = [] { const char* e = getenv("POPSIFT_LEVEL0_X2"); return !(e != nullptr && e[0] == '0'); }();
            if constexpr (R == 8) {
                if (h.v_ifilter != nullptr) {
                    // VLFeat_Relative: the interpolated vertical pass; np pairs run on the next instantiation (zero-weight pairs)
                    const int np = (h.v_ispan - 1) / 2;
                    f.vg0 = h.v_ifilter[0];
                    static const int force = [] { const char* e = getenv("POPSIFT_INTERP_LITERAL"); return e != nullptr && e[0] == '1' ? 1 : 0; }();
                    f.vforce = force;
                    for (int p = 0; p < 4; p++) {
                        const int offset = 2 * p + 1;
                        const float u = p < np ? h.v_ifilter[offset] : 0.0f;
                        f.vmul[p] = p < np ? h.v_ifilter[offset + 1] : 0.0f;
                        f.voff[p] = offset + (1.0f - u);
                    }
                    const bool s1 = h.shift == 1.0f;
                    if (np <= 3) {
                        if (h.is_float) { if (s1) hipLaunchKernelGGL((k_level0_x2<R, true, true, 3>), grid, block, 0, s, f);  else hipLaunchKernelGGL((k_level0_x2<R, true, false, 3>), grid, block, 0, s, f); }
                        else            { if (s1) hipLaunchKernelGGL((k_level0_x2<R, false, true, 3>), grid, block, 0, s, f); else hipLaunchKernelGGL((k_level0_x2<R, false, false, 3>), grid, block, 0, s, f); }
                    } else {
                        if (h.is_float) { if (s1) hipLaunchKernelGGL((k_level0_x2<R, true, true, 4>), grid, block, 0, s, f);  else hipLaunchKernelGGL((k_level0_x2<R, true, false, 4>), grid, block, 0, s, f); }
                        else            { if (s1) hipLaunchKernelGGL((k_level0_x2<R, false, true, 4>), grid, block, 0, s, f); else hipLaunchKernelGGL((k_level0_x2<R, false, false, 4>), grid, block, 0, s, f); }
                    }
                    return hipGetLastError();
                }
            }
            if (h.v_ifilter != nullptr) return hipErrorNotSupported;
            if (x2) {
                const bool s1 = h.shift == 1.0f;
                if (h.is_float) { if (s1) hipLaunchKernelGGL((k_level0_x2<R, true, true>), grid, block, 0, s, f);  else hipLaunchKernelGGL((k_level0_x2<R, true, false>), grid, block, 0, s, f); }
                else            { if (s1) hipLaunchKernelGGL((k_level0_x2<R, false, true>), grid, block, 0, s, f); else hipLaunchKernelGGL((k_level0_x2<R, false, false>), grid, block, 0, s, f); }
                return hipGetLastError();
            }
            if (h.is_float) hipLaunchKernelGGL((k_level0_fused<R, true>), grid, block, 0, s, f);
            else            hipLaunchKernelGGL((k_level0_fused<R, false>), grid, block, 0, s, f);
            return hipGetLastError();
        }
    }
    if (h.v_ifilter != nullptr) return hipErrorNotSupported;     // the interpolated vertical pass exists in the x2 kernel only
    const int pad = PSX_LEVEL0_PAD;
    const int wr = ((h.W + TW - 1) / TW) * TW;           // every strip reads full-width source rows
    UpArgs u;
    u.img = h.img; u.w = h.w; u.h = h.h; u.is_float = h.is_float;
    u.dst = h.tmp; u.W = h.W; u.H = h.H; u.pitch = h.tmp_pitch; u.pad = pad; u.ncol4 = (wr + 2 * pad) / 4;
    u.shift = h.shift;
    hipLaunchKernelGGL(k_upscale, dim3((u.ncol4 + 63) / 64, (h.H + 4 * UP_ROWS - 1) / (4 * UP_ROWS)), dim3(256), 0, s, u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    BlurArgs a;
    a.src = h.tmp; a.dst = h.dst; a.half_dst = nullptr;
    a.W = h.W; a.H = h.H; a.pitch = h.pitch; a.half_pitch = 0;
    a.src_pitch = h.tmp_pitch; a.src_xoff = pad; a.src_width = wr + 2 * pad;
    a.nstrips = (h.W + TW - 1) / TW;
    int nchunks;
    chunking(h.W, h.H, R, a.chunk_rows, nchunks);
    a.taps = h.taps_h; a.taps_v = h.taps_v;
#ifdef PSX_PHASE_TIMING
    a.dbg = 0;
#endif
    if constexpr (R <= 8) {
        const int dma = blur_tuning().dma;
        if (dma == 2) { launch_dma<R, true, 2>(a, s, nullptr, nullptr); return hipGetLastError(); }
        if (dma == 3) { launch_dma<R, true, 3>(a, s, nullptr, nullptr); return hipGetLastError(); }
    }
    hipLaunchKernelGGL((k_blur<R, true>), dim3(a.nstrips * nchunks), dim3(NT), 0, s, a);
    return hipGetLastError();
}

} // namespace

// span = one-sided tap count including the centre (GaussTable::span); radius R = span-1.
// Kernels are instantiated for a set of radii; a smaller radius runs on the next larger
// instantiation with zero weights, which is bit-exact (fma(x, 0, acc) == acc).
hipError_t psx_launch_blur(const float* src, float* dst, int W, int H, int pitch, const PsxTaps& taps,
                           int span, float* half_dst, int half_pitch, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    const int R = span - 1;
    if (R <= 5)  return launch_blur_r<5>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s, ev0, ev1);
    if (R <= 7)  return launch_blur_r<7>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s, ev0, ev1);
    if (R <= 8)  return launch_blur_r<8>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s, ev0, ev1);
    if (R <= 10) return launch_blur_r<10>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s, ev0, ev1);
    if (R <= 13) return launch_blur_r<13>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s, ev0, ev1);
    if (R <= 16) return launch_blur_r<16>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s, ev0, ev1);
    if (R <= 22) return launch_blur_r<22>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s, ev0, ev1);
    if (R <= 30) return launch_blur_r<30>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s, ev0, ev1);
    return hipErrorInvalidValue;
}

// Pairing rule of the diagonal schedule: the level launches of two planes share ONE launch when both chunk grids fit one
// round of resident workgroups
bool psx_blur_pair_ok(int W1, int H1, int W2, int H2, int span, int resident_marching)
{
    return psx_blur_grid(W1, H1, span) + psx_blur_grid(W2, H2, span) <= resident_marching;
}

// workgroups a blur of a W x H plane is launched with
int psx_blur_grid(int W, int H, int span)
{
    int cr, nchunks;
    chunking(W, H, span - 1, cr, nchunks);
    return ((W + TW - 1) / TW) * nchunks;
}

// two independent blurs in one launch; the kernel is instantiated for the larger radius (zero taps for the other)
hipError_t psx_launch_blur2(const PsxBlurJob& a, const PsxBlurJob& b, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    const int R = (a.span > b.span ? a.span : b.span) - 1;
    if (R <= 5)  return launch_blur2_r<5>(a, b, s, ev0, ev1);
    if (R <= 7)  return launch_blur2_r<7>(a, b, s, ev0, ev1);
    if (R <= 8)  return launch_blur2_r<8>(a, b, s, ev0, ev1);
    if (R <= 10) return launch_blur2_r<10>(a, b, s, ev0, ev1);
    if (R <= 13) return launch_blur2_r<13>(a, b, s, ev0, ev1);
    if (R <= 16) return launch_blur2_r<16>(a, b, s, ev0, ev1);
    if (R <= 22) return launch_blur2_r<22>(a, b, s, ev0, ev1);
    if (R <= 30) return launch_blur2_r<30>(a, b, s, ev0, ev1);
    return hipErrorInvalidValue;
}

// ---- k_pyramid_flow: host side ------------------------------------------------------------------------------------
// marching steps per work item of octave o (0 = the chunking of the launch-per-level schedule)
static int flow_steps(int o)
{
    // read per plan (psx_resize), not once per process: tests switch it between contexts
    std::vector<int> tab;
    const char* e = getenv("POPSIFT_FLOW_STEPS");
    std::string t = e ? e : "3,2,1";
    size_t p = 0;
    while (p <= t.size()) {
        const size_t q = t.find(',', p);
        tab.push_back(atoi(t.substr(p, q == std::string::npos ? std::string::npos : q - p).c_str()));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    if (tab.empty()) tab.push_back(0);
    return tab[(size_t)std::min<int>(o, (int)tab.size() - 1)];
}

bool psx_flow_plan(const PsxParams& P, const float* inc_filter, const int* inc_span, int first_octave,
                   int resident_blocks, int order, PsxFlowPlan* out)
{
    const int L = P.L, nlev = L - 1;
    const int noct = P.num_octaves - first_octave;
    if (noct <= 0 || nlev <= 0) return false;
    int rsel[PSX_GAUSS_LEVELS];
    for (int l = 1; l < L; l++) {
        const int R = inc_span[l] - 1;
        int i = 0;
        while (i < FLOW_NR && flow_radius(i) < R) i++;
        if (i == FLOW_NR) return false;                       // a radius the flow kernel is not instantiated for
        rsel[l] = i;
    }
    const int njobs = noct * nlev;
    std::vector<PsxFlowJob> jobs((size_t)njobs);
    int ncnt = 0;
    size_t nitems = 0;
    for (int o = first_octave; o < P.num_octaves; o++)
        for (int l = 1; l < L; l++) {
            PsxFlowJob& j = jobs[(size_t)(o - first_octave) * nlev + (l - 1)];
            memset(&j, 0, sizeof(j));
            const PsxOctave& oc = P.oct[o];
            j.src = oc.data + (size_t)(l - 1) * oc.plane;
            j.dst = oc.data + (size_t)l * oc.plane;
            if (l == L - 3 && o + 1 < P.num_octaves) { j.half_dst = P.oct[o + 1].data; j.half_pitch = P.oct[o + 1].pitch; }
            j.W = oc.w; j.H = oc.h; j.pitch = oc.pitch;
            j.nstrips = (oc.w + TW - 1) / TW;
            j.rsel = rsel[l];
            chunking(oc.w, oc.h, flow_radius(j.rsel), j.chunk_rows, j.nchunks);
            // The octaves that cannot fill the chip are a CHAIN of dependent levels (o+1 starts when level L-3 of o is
            // done): what counts there is the latency of one item, i.e. its number of marching steps, not the warm-up rows
            // it recomputes.  POPSIFT_FLOW_STEPS="s0,s1,s2,.." overrides the steps per item of octave 0, 1, 2, .. (0 = keep)
            {
                const int so = flow_steps(o - 0);
                if (so > 0) {
                    int cr = so * BR - 2 * flow_radius(j.rsel);
                    if (cr < 4) cr = 4;
                    if (cr > oc.h) cr = oc.h;
                    j.chunk_rows = cr; j.nchunks = (oc.h + cr - 1) / cr;
                }
            }
            j.cnt_off = ncnt; ncnt += j.nchunks;
            j.dep_cnt_off = -1; j.dep_need = 0;
            j.octave = o; j.level = l;
            for (int k = 0; k < PSX_GAUSS_ALIGN; k++) j.taps.g[k] = inc_filter[l * PSX_GAUSS_ALIGN + k];
            nitems += (size_t)j.nstrips * j.nchunks;
            if (j.nstrips > 65535 || j.nchunks > 65535) return false;
        }
    if (ncnt > PSX_FLOW_MAX_COUNTERS || njobs > 65535 || nitems > (size_t)1 << 30) return false;

    // dependencies at (job, chunk) granularity, and each node's earliest start: in hops (order 0) or in estimated
    // item durations (order 1), for the ticket order
    struct Node { int c0, c1, dep; float t0, dur; };
    std::vector<std::vector<Node>> node((size_t)njobs);
    for (int ji = 0; ji < njobs; ji++) {
        PsxFlowJob& j = jobs[(size_t)ji];
        const int o = j.octave, l = j.level;
        const int Rt = flow_radius(j.rsel);
        int dep = -1, scale = 1;
        if (l >= 2) dep = ji - 1;
        else if (o > first_octave) { dep = (o - 1 - first_octave) * nlev + (L - 3 - 1); scale = 2; }
        if (dep >= 0) { j.dep_cnt_off = jobs[(size_t)dep].cnt_off; j.dep_need = jobs[(size_t)dep].nstrips; }
        node[(size_t)ji].resize((size_t)j.nchunks);
        for (int c = 0; c < j.nchunks; c++) {
            Node& n = node[(size_t)ji][(size_t)c];
            const int Y0 = c * j.chunk_rows, Y1 = std::min(Y0 + j.chunk_rows, j.H);
            const int nsteps = (Y1 - Y0 + 2 * Rt + BR - 1) / BR;
            // every row the workgroup LOADS (the staged rows of all its steps), not only the rows its outputs need: a
            // line must never enter a cache before it is complete
            const int lo = std::max(Y0 - Rt, 0), hi = std::min(Y0 - Rt + nsteps * BR - 1, j.H - 1);
            n.dep = dep; n.c0 = n.c1 = 0; n.t0 = 0.0f;
            n.dur = order == 1 ? (float)nsteps * (1.0f + (float)Rt / 16.0f) + 1.0f : 1.0f;
            if (dep >= 0) {
                const PsxFlowJob& d = jobs[(size_t)dep];
                n.c0 = std::min(lo * scale, d.H - 1) / d.chunk_rows;
                n.c1 = std::min(hi * scale, d.H - 1) / d.chunk_rows;
                if (n.c1 - n.c0 + 1 > 64) return false;
                for (int q = n.c0; q <= n.c1; q++) {
                    const Node& dn = node[(size_t)dep][(size_t)q];
                    n.t0 = std::max(n.t0, dn.t0 + dn.dur);
                }
            }
        }
    }
    struct Key { float t; int octave, job, chunk; };
    std::vector<Key> groups;
    if (order == 2) {
        // Ticket order = a simulated list schedule: `grid` processors, every item of a node (job, chunk) takes dur
        // (a per-step cost model calibrated on MI355X traces), a node is ready when all of its producer nodes have
        // finished, and among the ready nodes the one with the longest chain of dependents behind it goes first -- the
        // small octaves (a chain of ~14 dependent levels) then run as soon as their rows exist instead of queueing behind
        // octave 0's bulk.  The order of the simulated starts is topological (a consumer starts after its producers end).
        int P_ = std::min(resident_blocks, (int)std::min<size_t>(nitems, 1u << 20));
        if (const char* e = getenv("POPSIFT_FLOW_GRID")) { const int v = atoi(e); if (v >= PSX_FLOW_SHARDS) P_ = v; }
        std::vector<int> base((size_t)njobs + 1, 0);
        for (int ji = 0; ji < njobs; ji++) base[(size_t)ji + 1] = base[(size_t)ji] + jobs[(size_t)ji].nchunks;
        const int nn = base[(size_t)njobs];
        std::vector<std::vector<int>> rev((size_t)nn);
        std::vector<int> pending((size_t)nn, 0), left((size_t)nn, 0);
        std::vector<float> dur((size_t)nn), bl((size_t)nn, 0.0f), ready((size_t)nn, 0.0f), fin((size_t)nn, 0.0f);
        for (int ji = 0; ji < njobs; ji++)
            for (int c = 0; c < jobs[(size_t)ji].nchunks; c++) {
                const Node& n = node[(size_t)ji][(size_t)c];
                const int id = base[(size_t)ji] + c;
                const PsxFlowJob& j = jobs[(size_t)ji];
                const int Rt = flow_radius(j.rsel);
                const int Y0 = c * j.chunk_rows, Y1 = std::min(Y0 + j.chunk_rows, j.H);
                const int nsteps = (Y1 - Y0 + 2 * Rt + BR - 1) / BR;
                dur[(size_t)id] = 1.6f + (1.7f + 0.02f * (float)Rt) * (float)nsteps;       // us, 4 workgroups per CU
                left[(size_t)id] = j.nstrips;
                if (n.dep >= 0)
                    for (int q = n.c0; q <= n.c1; q++) { rev[(size_t)(base[(size_t)n.dep] + q)].push_back(id); pending[(size_t)id]++; }
            }
        for (int id = nn - 1; id >= 0; id--) {          // producers have smaller ids than their consumers
            float m = 0.0f;
            for (int d : rev[(size_t)id]) m = std::max(m, bl[(size_t)d]);
            bl[(size_t)id] = dur[(size_t)id] + m;
        }
        typedef std::pair<float, int> FI;
        std::priority_queue<FI, std::vector<FI>, std::greater<FI>> procs, future;      // (free time, proc) / (ready time, node)
        std::priority_queue<FI> avail;                                               // (bottom level, node)
        for (int p = 0; p < P_; p++) procs.push(FI(0.0f, p));
        for (int id = 0; id < nn; id++) if (pending[(size_t)id] == 0) future.push(FI(0.0f, id));
        size_t done = 0;
        std::vector<int> job_of((size_t)nn), chunk_of((size_t)nn);
        for (int ji = 0; ji < njobs; ji++) for (int c = 0; c < jobs[(size_t)ji].nchunks; c++) { job_of[(size_t)(base[(size_t)ji] + c)] = ji; chunk_of[(size_t)(base[(size_t)ji] + c)] = c; }
        std::vector<char> emitted((size_t)nn, 0);
        while (done < nitems) {
            FI pr = procs.top(); procs.pop();
            float tau = pr.first;
            while (!future.empty() && future.top().first <= tau) { avail.push(FI(bl[(size_t)future.top().second], future.top().second)); future.pop(); }
            if (avail.empty()) {
                if (future.empty()) return false;            // cannot happen: the graph is acyclic
                tau = future.top().first;
                while (!future.empty() && future.top().first <= tau) { avail.push(FI(bl[(size_t)future.top().second], future.top().second)); future.pop(); }
            }
            const int id = avail.top().second;
            // the whole node goes out as one group of consecutive tickets (its strips are spread over the classes
            // below); its items start on the processors that free up next
            if (!emitted[(size_t)id]) { emitted[(size_t)id] = 1; groups.push_back(Key{tau, jobs[(size_t)job_of[(size_t)id]].octave, job_of[(size_t)id], chunk_of[(size_t)id]}); }
            const float f = tau + dur[(size_t)id];
            fin[(size_t)id] = std::max(fin[(size_t)id], f);
            procs.push(FI(f, pr.second));
            done++;
            if (--left[(size_t)id] == 0) {
                avail.pop();
                for (int d : rev[(size_t)id]) {
                    ready[(size_t)d] = std::max(ready[(size_t)d], fin[(size_t)id]);
                    if (--pending[(size_t)d] == 0) future.push(FI(ready[(size_t)d], d));
                }
            }
        }
    } else {
    for (int ji = 0; ji < njobs; ji++)
        for (int c = 0; c < jobs[(size_t)ji].nchunks; c++)
            groups.push_back(Key{node[(size_t)ji][(size_t)c].t0, jobs[(size_t)ji].octave, ji, c});
    // earliest first; among equals the deeper octave first (its chain of dependent levels is the longer one)
    std::stable_sort(groups.begin(), groups.end(), [](const Key& a, const Key& b) {
        if (a.t != b.t) return a.t < b.t;
        if (a.octave != b.octave) return a.octave > b.octave;
        if (a.job != b.job) return a.job < b.job;
        return a.chunk < b.chunk;
    });
    }
    std::vector<PsxFlowItem> items(nitems);
    size_t base = 0;
    for (const Key& g : groups) {
        const PsxFlowJob& j = jobs[(size_t)g.job];
        const Node& n = node[(size_t)g.job][(size_t)g.chunk];
        // ticket base + p belongs to class (base + p) & 7, which is (observed) the XCD the workgroup runs on: give every
        // class a contiguous run of strips, so that neighbouring strips -- they share halo columns -- meet in one L2
        int per[PSX_FLOW_SHARDS] = {0}, start[PSX_FLOW_SHARDS], used[PSX_FLOW_SHARDS] = {0};
        for (int p = 0; p < j.nstrips; p++) per[(base + (size_t)p) & (PSX_FLOW_SHARDS - 1)]++;
        int acc = 0;
        for (int q = 0; q < PSX_FLOW_SHARDS; q++) { start[q] = acc; acc += per[q]; }
        for (int p = 0; p < j.nstrips; p++) {
            const int q = (int)((base + (size_t)p) & (PSX_FLOW_SHARDS - 1));
            PsxFlowItem& it = items[base + (size_t)p];
            it.job = (unsigned short)g.job; it.strip = (unsigned short)(start[q] + used[q]++); it.chunk = (unsigned short)g.chunk;
            it.dep_c0 = (unsigned short)n.c0; it.dep_c1 = (unsigned short)n.c1; it.pad0 = it.pad1 = it.pad2 = 0;
        }
        base += (size_t)j.nstrips;
    }
    out->njobs = njobs; out->nitems = (int)nitems; out->ncounters = ncnt;
    int grid = std::min(resident_blocks, (int)((nitems + PSX_FLOW_SHARDS - 1) / PSX_FLOW_SHARDS) * PSX_FLOW_SHARDS);
    if (const char* e = getenv("POPSIFT_FLOW_GRID")) { const int v = atoi(e); if (v >= PSX_FLOW_SHARDS) grid = v; }
    out->grid = (grid / PSX_FLOW_SHARDS) * PSX_FLOW_SHARDS;
    out->jobs = static_cast<PsxFlowJob*>(malloc(sizeof(PsxFlowJob) * (size_t)njobs));
    out->items = static_cast<PsxFlowItem*>(malloc(sizeof(PsxFlowItem) * nitems));
    if (!out->jobs || !out->items) { free(out->jobs); free(out->items); out->jobs = nullptr; out->items = nullptr; return false; }
    memcpy(out->jobs, jobs.data(), sizeof(PsxFlowJob) * (size_t)njobs);
    memcpy(out->items, items.data(), sizeof(PsxFlowItem) * nitems);
    return true;
}

// Host-only self check of a plan (no device needed; tests/test_capi_cpu.py): every (job, strip, chunk) exactly once, every
// item behind ALL items of the chunks it waits for (the ticket order is topological: the deadlock-freedom argument of
// k_pyramid_flow rests on it), the waited-for chunks cover every source row the workgroup loads.  Returns the number of
// items, or a negative code naming the violated invariant.  stats (optional, 4 ints): jobs, counters, grid, largest wait list.
extern "C" int psx_flow_selfcheck(int w0, int h0, int num_octaves, int levels, const int* spans, int first_octave,
                                  int resident_blocks, int order, int* stats)
{
    PsxParams P;
    memset(&P, 0, sizeof(P));
    P.num_octaves = num_octaves; P.levels = levels; P.L = levels + 3;
    if (num_octaves < 1 || num_octaves > PSX_MAX_OCTAVES || P.L > PSX_GAUSS_LEVELS) return -1;
    size_t off = 4096;
    int w = w0, h = h0;
    for (int o = 0; o < num_octaves; o++) {
        PsxOctave& oc = P.oct[o];
        oc.w = w; oc.h = h; oc.pitch = (w + 63) & ~63; oc.plane = (size_t)oc.pitch * h;
        oc.data = reinterpret_cast<float*>(off * 4);
        off += oc.plane * P.L;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    std::vector<float> filt((size_t)PSX_GAUSS_LEVELS * PSX_GAUSS_ALIGN, 0.0f);
    PsxFlowPlan plan{};
    if (!psx_flow_plan(P, filt.data(), spans, first_octave, resident_blocks, order, &plan)) return -2;
    int rc = plan.nitems, maxwait = 0;
    std::vector<std::vector<int>> last_pos((size_t)plan.njobs), seen((size_t)plan.njobs);
    for (int j = 0; j < plan.njobs; j++) {
        last_pos[(size_t)j].assign((size_t)plan.jobs[j].nchunks, -1);
        seen[(size_t)j].assign((size_t)plan.jobs[j].nchunks * plan.jobs[j].nstrips, 0);
    }
    for (int p = 0; p < plan.nitems && rc > 0; p++) {
        const PsxFlowItem& it = plan.items[p];
        if (it.job >= plan.njobs) { rc = -3; break; }
        const PsxFlowJob& j = plan.jobs[it.job];
        if (it.strip >= j.nstrips || it.chunk >= j.nchunks) { rc = -3; break; }
        int& s = seen[it.job][(size_t)it.chunk * j.nstrips + it.strip];
        if (s++) { rc = -4; break; }
        last_pos[it.job][it.chunk] = p;                       // positions are visited in increasing order
    }
    for (int j = 0; j < plan.njobs && rc > 0; j++)
        for (int v : seen[(size_t)j]) if (v != 1) { rc = -4; break; }
    const int nlev = P.L - 1;
    for (int p = 0; p < plan.nitems && rc > 0; p++) {
        const PsxFlowItem& it = plan.items[p];
        const PsxFlowJob& j = plan.jobs[it.job];
        const int Rt = flow_radius(j.rsel);
        if (spans[j.level] - 1 > Rt) { rc = -5; break; }
        const int Y0 = it.chunk * j.chunk_rows, Y1 = std::min(Y0 + j.chunk_rows, j.H);
        const int nsteps = (Y1 - Y0 + 2 * Rt + BR - 1) / BR;
        const int lo = std::max(Y0 - Rt, 0), hi = std::min(Y0 - Rt + nsteps * BR - 1, j.H - 1);
        int dep = -1, scale = 1;
        if (j.level >= 2) dep = it.job - 1;
        else if (j.octave > first_octave) { dep = (j.octave - 1 - first_octave) * nlev + (P.L - 3 - 1); scale = 2; }
        if ((dep >= 0) != (j.dep_cnt_off >= 0)) { rc = -6; break; }
        if (dep < 0) continue;
        const PsxFlowJob& d = plan.jobs[dep];
        if (j.dep_cnt_off != d.cnt_off || j.dep_need != d.nstrips || j.src != d.dst + 0) {
            // the source plane of a level-1 job of octave o > first is the half plane of (o - 1, L - 3), not its dst
            if (!(scale == 2 && j.dep_cnt_off == d.cnt_off && j.dep_need == d.nstrips && j.src == d.half_dst)) { rc = -6; break; }
        }
        // rows lo..hi of the source = rows scale*lo .. scale*hi of the producing job's OUTPUT
        const int need0 = std::min(lo * scale, d.H - 1) / d.chunk_rows, need1 = std::min(hi * scale, d.H - 1) / d.chunk_rows;
        if (it.dep_c0 > need0 || it.dep_c1 < need1 || it.dep_c1 >= d.nchunks) { rc = -7; break; }
        maxwait = std::max(maxwait, it.dep_c1 - it.dep_c0 + 1);
        for (int c = it.dep_c0; c <= it.dep_c1; c++)
            if (last_pos[(size_t)dep][(size_t)c] >= p) { rc = -8; break; }     // a producer sits behind its consumer
    }
    if (plan.grid < PSX_FLOW_SHARDS || plan.grid % PSX_FLOW_SHARDS != 0) rc = rc > 0 ? -9 : rc;
    if (stats) { stats[0] = plan.njobs; stats[1] = plan.ncounters; stats[2] = plan.grid; stats[3] = maxwait; }
    free(plan.jobs); free(plan.items);
    return rc;
}

hipError_t psx_launch_flow(const PsxFlowJob* d_jobs, const PsxFlowItem* d_items, int nitems, int* d_state, int* d_err,
                           int grid, int ldmode, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, long long* trace)
{
    const dim3 g(grid), b(NT);
    const bool ext = ev0 != nullptr || ev1 != nullptr;
    // measurement switch (results are wrong with it): 1 = no dependency waits, 2 = no arithmetic (tickets, waits and publishes only)
    static const int dbg = [] { const char* e = getenv("POPSIFT_FLOW_DEBUG"); return e ? atoi(e) : 0; }();
    if (ldmode == 2) {
        if (ext) hipExtLaunchKernelGGL((k_pyramid_flow<2>), g, b, 0, s, ev0, ev1, 0, d_jobs, d_items, nitems, d_state, d_err, dbg, trace);
        else     hipLaunchKernelGGL((k_pyramid_flow<2>), g, b, 0, s, d_jobs, d_items, nitems, d_state, d_err, dbg, trace);
    } else {
        if (ext) hipExtLaunchKernelGGL((k_pyramid_flow<1>), g, b, 0, s, ev0, ev1, 0, d_jobs, d_items, nitems, d_state, d_err, dbg, trace);
        else     hipLaunchKernelGGL((k_pyramid_flow<1>), g, b, 0, s, d_jobs, d_items, nitems, d_state, d_err, dbg, trace);
    }
    return hipGetLastError();
}

// The U(x-k) / U(x+k) form of the level-0 kernels equals the reference's per-tap coordinates (x + shift)/W -+ k/W
// (s_pyramid_build_ra.cu:35-50) only while both forms land on the same 1/256 sub-texel.  For a power-of-two ratio
// between image and octave the sub-texel positions are multiples of 2^-(n+1), far from the rounding boundaries at
// (m + 0.5)/256, and the one or two ulps between the two forms cannot matter; for any other ratio (fractional
// setDownsampling values, ScaleDirect octaves of an odd-sized image) a few coordinates per row sit on a boundary and
// the two forms differ by one 1/256 step -- those run the literal kernels of pyramid_alt.hip.
static bool pow2_ratio(int big, int small)
{
    for (int n = 0; n <= 4; n++) if ((small << n) == big) return true;
    return false;
}
bool psx_level0_exact(int w, int h, int W, int H)
{
    return (W >= w ? pow2_ratio(W, w) : pow2_ratio(w, W)) && (H >= h ? pow2_ratio(H, h) : pow2_ratio(h, H));
}

// level 0 with the interpolated vertical pass of GaussMode VLFeat_Relative (a.v_ifilter / a.v_ispan): covered by the x2 kernel for
// up to 4 tap pairs and a "dd" radius up to 8
bool psx_level0_interp_ok(const PsxLevel0Args& a)
{
    static const bool off = [] { const char* e = getenv("POPSIFT_INTERP_FUSED"); return e != nullptr && e[0] == '0'; }();
    return !off && level0_fused_enabled() && a.W == 2 * a.w && a.H == 2 * a.h && a.w >= 4 && (a.shift == 1.0f || a.shift == 0.5f) &&
           (a.v_ispan - 1) / 2 <= 4 && a.span_h - 1 <= 8 && psx_level0_exact(a.w, a.h, a.W, a.H);
}

hipError_t psx_launch_level0(const PsxLevel0Args& a, hipStream_t s)
{
    if (a.v_ifilter != nullptr) return psx_level0_interp_ok(a) ? launch_level0_r<8>(a, s) : hipErrorNotSupported;
    if (!psx_level0_exact(a.w, a.h, a.W, a.H)) return psx_launch_level0_literal(a, s);
    const int R = (a.span_h > a.span_v ? a.span_h : a.span_v) - 1;
    if (R <= 5)  return launch_level0_r<5>(a, s);
    if (R <= 8)  return launch_level0_r<8>(a, s);
    if (R <= 16) return launch_level0_r<16>(a, s);
    if (R <= 30) return launch_level0_r<30>(a, s);
    return hipErrorInvalidValue;
}

hipError_t psx_launch_dog(const float* a, const float* b, float* d, int W, int H, int pitch, hipStream_t s)
{
    hipLaunchKernelGGL(k_dog, dim3((W + 255) / 256, H), dim3(256), 0, s, a, b, d, W, H, pitch);
    return hipGetLastError();
}
