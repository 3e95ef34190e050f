// pyramid_tile.hip -- k_blur_tile: several consecutive blur levels of an octave in one launch, on LDS-resident tiles
// (blur_tile_core.h holds the phases and the reasoning; this file is the kernel around them and its launcher).
//
// Replaces, for the octaves that cannot fill the chip, the per-level launches of the default build_pyramid branch
// (s_pyramid_build.cu:547-575: horiz + vert per level and octave, get_by_2_pick_every_second per octave): a launch runs
// the tiles of up to PSX_TILE_JOBS jobs, a job = (octave, consecutive levels).  Planes stay bit-identical: the
// arithmetic is blur_arith.h's, shared with k_blur.
#include "psx_internal.h"
#include "blur_tile_core.h"

#include <hip/hip_ext.h>

namespace {

// Barrier between two phases that hand data over through LDS only: wait for this wave's LDS operations, not for its
// global stores (__syncthreads() waits for vmcnt(0) as well, i.e. for the write-through plane stores of the vertical
// pass to reach memory -- microseconds per level that nobody needs: the planes are read by LATER launches).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef PSX_PHASE_TIMING
// measurement build (tools/build_phase_lib.sh, tools/tile_phase.py): clock64 at the phase boundaries of every workgroup
__device__ long long* g_tile_dbg = nullptr;
// record slot = an atomic ticket (word 0 of the buffer), so that every workgroup of every launch of a frame keeps its stamps
#define TSTAMP(i) do { if (threadIdx.x == 0 && g_tile_dbg) g_tile_dbg[16 + (size_t)tslot * 16 + (i)] = clock64(); } while (0)
#else
#define TSTAMP(i)
#endif

template <int NT>
__device__ __forceinline__ void tile_run(const PsxTileJob* __restrict__ jobs, const int njobs, float* const s_tile)
{
#ifdef PSX_PHASE_TIMING
    int tslot = 0;
    if (threadIdx.x == 0 && g_tile_dbg) tslot = (int)atomicAdd((unsigned long long*)g_tile_dbg, 1ull) & 4095;
#endif
    TSTAMP(0);
    const int tid = threadIdx.x;
    const int lid = psx_xcd_remap(blockIdx.x, gridDim.x);
    int ji = 0;
    for (int q = 1; q < njobs; q++) if (lid >= jobs[q].h.block0) ji = q;
    const PsxTileJob* __restrict__ jb = jobs + ji;
    const PsxTileHdr h = jb->h;                           // workgroup uniform: scalar registers
    const int t = lid - h.block0;
    const int ty = t / h.tiles_x, tx = t - ty * h.tiles_x;
    const int X0 = tx * h.TX, Y0 = ty * h.TY;
    float* const P = s_tile;
    float* const Q = s_tile + h.NR * h.SP;

    TSTAMP(1);
#ifdef PSX_PHASE_TIMING
    if (threadIdx.x == 0 && g_tile_dbg) g_tile_dbg[16 + (size_t)tslot * 16 + 15] = (long long)h.nlev | ((long long)gridDim.x << 8) | ((long long)h.TX << 32) | ((long long)h.TY << 40);
#endif
    tile_load<NT>(h, X0, Y0, P, tid);
    TSTAMP(2);
    // any cell of P outside the plane: the levels' regions must be re-clamped between levels
    const bool edge = X0 - h.OX < 0 || X0 + h.TX + h.OX > h.W || Y0 - h.OY < 0 || Y0 + h.TY + h.OY > h.H;
    lds_barrier();
    TSTAMP(3);
    for (int l = 0; l < h.nlev; l++) {
        const PsxTileLevel lv = jb->lev[l];
        PsxTaps tp;
#pragma unroll
        for (int i = 0; i < 16; i++) tp.g[i] = jb->taps[l].g[i];      // radii <= 13
        float* const gdst = jb->dst[l];
        float* const ghalf = l == h.half_lev ? h.half_dst : nullptr;
        const bool keep = l + 1 < h.nlev;
        // the thread index is laundered per pass: otherwise the lane geometry of all ten pass bodies is hoisted out of
        // the level loop and spilled (44 bytes of scratch per lane)
        int th = tid, tv = tid;
        asm volatile("" : "+v"(th));
        switch (lv.rsel) {
            case 0:  tile_hpass<psx_tile_radius(0), NT>(h, lv, tp, P, Q, th); break;
            case 1:  tile_hpass<psx_tile_radius(1), NT>(h, lv, tp, P, Q, th); break;
            case 2:  tile_hpass<psx_tile_radius(2), NT>(h, lv, tp, P, Q, th); break;
            case 3:  tile_hpass<psx_tile_radius(3), NT>(h, lv, tp, P, Q, th); break;
            default: tile_hpass<psx_tile_radius(4), NT>(h, lv, tp, P, Q, th); break;
        }
        TSTAMP(4 + 3 * l);
        lds_barrier();
        TSTAMP(5 + 3 * l);
        asm volatile("" : "+v"(tv));
        switch (lv.rsel) {
            case 0:  tile_vpass<psx_tile_radius(0), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, tv); break;
            case 1:  tile_vpass<psx_tile_radius(1), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, tv); break;
            case 2:  tile_vpass<psx_tile_radius(2), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, tv); break;
            case 3:  tile_vpass<psx_tile_radius(3), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, tv); break;
            default: tile_vpass<psx_tile_radius(4), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, tv); break;
        }
        TSTAMP(6 + 3 * l);
        if (keep) {
            lds_barrier();
            if (edge) {
                tile_fixup<NT>(h, lv, P, X0, Y0, tid);
                lds_barrier();
            }
        }
    }
}

// LDSF floats of static LDS: 80 KB (two workgroups of 512 threads per CU) or the CU's whole 160 KB
template <int NT, int LDSF, int MINB>
__global__ __launch_bounds__(NT, MINB) void k_blur_tile(const PsxTileJob* __restrict__ jobs, int njobs)
{
    __shared__ __attribute__((aligned(16))) float s_tile[LDSF];
    tile_run<NT>(jobs, njobs, s_tile);
}

constexpr int LDSF_HALF = 80 * 1024 / 4, LDSF_FULL = 160 * 1024 / 4;

template <int NT, int LDSF, int MINB>
hipError_t launch(const PsxTileJob* d_jobs, int njobs, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    if (ev0 != nullptr || ev1 != nullptr) hipExtLaunchKernelGGL((k_blur_tile<NT, LDSF, MINB>), dim3(grid), dim3(NT), 0, s, ev0, ev1, 0, d_jobs, njobs);
    else                                  hipLaunchKernelGGL((k_blur_tile<NT, LDSF, MINB>), dim3(grid), dim3(NT), 0, s, d_jobs, njobs);
    return hipGetLastError();
}

} // namespace

#ifdef PSX_PHASE_TIMING
extern "C" void psx_debug_set_tile_buffer(long long* d) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_dbg), &d, sizeof(d)); }
#endif

hipError_t psx_launch_blur_tile(const PsxTileJob* d_jobs, int njobs, int grid, size_t lds_bytes, int nt, hipStream_t s,
                                hipEvent_t ev0, hipEvent_t ev1)
{
    if (njobs < 1 || grid < 1 || lds_bytes > (size_t)LDSF_FULL * 4) return hipErrorInvalidValue;
    if (nt >= 1024) return launch<1024, LDSF_FULL, 4>(d_jobs, njobs, grid, s, ev0, ev1);
    if (lds_bytes <= (size_t)LDSF_HALF * 4) return launch<512, LDSF_HALF, 4>(d_jobs, njobs, grid, s, ev0, ev1);
    return launch<512, LDSF_FULL, 2>(d_jobs, njobs, grid, s, ev0, ev1);
}
