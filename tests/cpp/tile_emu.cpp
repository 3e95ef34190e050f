// tile_emu.cpp -- the multi-level tile kernel (popsift_amd/csrc/hip/blur_tile_core.h) run on the CPU.
//
// TEST INFRASTRUCTURE.  The phase functions of k_blur_tile are compiled here for the host (PSX_TILE_EMU) and run the
// way the GPU runs them, only serially: for every tile, for every phase, for every "thread".  P and Q start as NaN, the
// destination planes as NaN: a read of a cell nobody wrote, or a pixel nobody stored, shows up as a mismatch against the
// oracle's planes (tests/test_tile_emu_cpu.py).  What this cannot see: races between threads of one phase (there are none
// by construction: the phases write disjoint cells and read only cells written in earlier phases) and the device-only
// macros (address spaces, write-through stores).
//
// Build: clang++ -O1 -std=c++17 -ffp-contract=off -DPSX_TILE_EMU -shared -fPIC (tests/test_tile_emu_cpu.py does it).
#include "blur_tile_core.h"

#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace {

template <int NT>
void run_tile(const PsxTileJob& job, int tile, std::vector<float>& lds)
{
    const PsxTileHdr& h = job.h;
    const int ty = tile / h.tiles_x, tx = tile - ty * h.tiles_x;
    const int X0 = tx * h.TX, Y0 = ty * h.TY;
    const float nan = std::numeric_limits<float>::quiet_NaN();
    for (float& v : lds) v = nan;
    float* P = lds.data();
    float* Q = lds.data() + (size_t)h.NR * h.SP;
    for (int t = 0; t < NT; t++) tile_load<NT>(h, X0, Y0, P, t);
    const bool edge = X0 - h.OX < 0 || X0 + h.TX + h.OX > h.W || Y0 - h.OY < 0 || Y0 + h.TY + h.OY > h.H;
    for (int l = 0; l < h.nlev; l++) {
        const PsxTileLevel lv = job.lev[l];
        const PsxTaps tp = job.taps[l];
        float* gdst = job.dst[l];
        float* ghalf = l == h.half_lev ? h.half_dst : nullptr;
        const bool keep = l + 1 < h.nlev;
        for (int t = 0; t < NT; t++)
            switch (lv.rsel) {
                case 0:  tile_hpass<psx_tile_radius(0), NT>(h, lv, tp, P, Q, t); break;
                case 1:  tile_hpass<psx_tile_radius(1), NT>(h, lv, tp, P, Q, t); break;
                case 2:  tile_hpass<psx_tile_radius(2), NT>(h, lv, tp, P, Q, t); break;
                case 3:  tile_hpass<psx_tile_radius(3), NT>(h, lv, tp, P, Q, t); break;
                default: tile_hpass<psx_tile_radius(4), NT>(h, lv, tp, P, Q, t); break;
            }
        for (int t = 0; t < NT; t++)
            switch (lv.rsel) {
                case 0:  tile_vpass<psx_tile_radius(0), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, t); break;
                case 1:  tile_vpass<psx_tile_radius(1), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, t); break;
                case 2:  tile_vpass<psx_tile_radius(2), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, t); break;
                case 3:  tile_vpass<psx_tile_radius(3), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, t); break;
                default: tile_vpass<psx_tile_radius(4), NT>(h, lv, tp, Q, P, gdst, ghalf, X0, Y0, keep, t); break;
            }
        if (keep && edge)
            for (int t = 0; t < NT; t++) tile_fixup<NT>(h, lv, P, X0, Y0, t);
    }
}

} // namespace

// src: plane of the level in front of the first fused one (pitch floats per row, 64-byte aligned); spans / taps: nlev
// entries (span = radius + 1; 32 taps each); dst: nlev planes of pitch * H floats, back to back; half: the decimated
// output of level half_lev (or null).  info (optional, 8 ints): OX OY NC NR SP tiles lds_bytes QOFF.
// Returns 0, or -1 when the plan rejects the job (the kernel would not be used for it).
extern "C" int tile_emu_run(const float* src, int W, int H, int pitch, int nlev, const int* spans, const float* taps,
                            int TX, int TY, int nt, float* dst, float* half, int half_lev, int half_pitch, int* info)
{
    PsxTileJob job;
    memset(&job, 0, sizeof(job));
    int radii[PSX_TILE_MAXLEV];
    if (nlev < 1 || nlev > PSX_TILE_MAXLEV) return -1;
    for (int l = 0; l < nlev; l++) radii[l] = spans[l] - 1;
    const size_t bytes = psx_tile_plan_job(job, W, H, pitch, nlev, radii, TX, TY);
    if (bytes == 0) return -1;
    const int rows_per_pass = nt >> job.h.lpr_shift;
    if ((nt != 512 && nt != 1024) || job.h.NR > (nt >= 1024 ? 8 : 16) * rows_per_pass) return -1;
    job.h.src = src;
    job.h.half_dst = half; job.h.half_pitch = half_pitch; job.h.half_lev = half ? half_lev : -1;
    job.h.block0 = 0;
    for (int l = 0; l < nlev; l++) {
        job.dst[l] = dst + (size_t)l * pitch * H;
        for (int k = 0; k < PSX_GAUSS_ALIGN; k++) job.taps[l].g[k] = taps[l * PSX_GAUSS_ALIGN + k];
    }
    if (info) {
        info[0] = job.h.OX; info[1] = job.h.OY; info[2] = job.h.NC; info[3] = job.h.NR; info[4] = job.h.SP;
        info[5] = job.h.tiles_x * job.h.tiles_y; info[6] = (int)bytes; info[7] = job.h.QOFF;
    }
    // 16-byte aligned LDS image (the phases use aligned vector accesses)
    std::vector<float> lds(bytes / 4 + 4);
    const int ntiles = job.h.tiles_x * job.h.tiles_y;
    for (int t = 0; t < ntiles; t++) {
        if (nt == 512) run_tile<512>(job, t, lds); else run_tile<1024>(job, t, lds);
    }
    return 0;
}
