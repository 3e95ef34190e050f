// psx_internal.h -- device-side data layout shared by the HIP translation units.
//
// Design: no textures, no surfaces, no global __device__/__constant__ symbols.
// Every Pyramid is a heap context; kernels receive a pointer to its PsxParams block (read
// through scalar loads) and to its PsxCounters block.  Gaussian planes are plain pitched
// float32 rows in HBM; texture clamp addressing is done by clamping integer coordinates.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "popsift_hip.h"

#define PSX_WAVE 64
#define PSX_CAND_SUB 64

struct PsxOctave {
    float*   data;     // L planes, plane l at data + l*plane
    int      w, h;
    int      pitch;    // floats per row (multiple of 64)
    int      pad;
    size_t   plane;    // floats per plane = pitch*h
};

// Parameter block, uploaded whenever config or dimensions change (replaces d_consts,
// sift_constants.h:58-69, and the per-octave texture objects, sift_octave.h:25-173).
struct PsxParams {
    int   num_octaves;
    int   L;               // Gaussian levels per octave = levels+3
    int   levels;
    int   sift_mode;
    int   norm_mode;
    int   norm_multi;
    int   max_extrema;     // per octave
    int   ext_capacity;    // entries in extrema / features arrays
    int   ori_capacity;    // entries in desc / feat_to_ext arrays
    int   grid_size;
    int   up_fac;          // int(upscale_factor), sift_pyramid.cu:250
    float sigma0;
    float sigma_k;         // 2^(1/levels), sift_constants.cu:27
    float threshold;       // Config::getPeakThreshold()
    float edge_limit;
    float pad0;
    PsxOctave oct[PSX_MAX_OCTAVES];
    float w_grid_div[PSX_MAX_OCTAVES];   // sift_octave.cu:40-41
    float h_grid_div[PSX_MAX_OCTAVES];
    psx_iext* iext[PSX_MAX_OCTAVES];     // max_extrema entries each (dobuf.i_ext_dat)
    int*      iext_off[PSX_MAX_OCTAVES]; // dobuf.i_ext_off
    // extremum candidates awaiting refinement (y<<32 | z<<24 | x): per octave PSX_CAND_SUB sub-lists of
    // cand_capacity entries, each with its own counter on its own 128-byte line -- thousands of tiles
    // appending through ONE counter serialise on that address in L2
    unsigned long long* cand[PSX_MAX_OCTAVES];
    int*      cand_ct;                   // [num_octaves][PSX_CAND_SUB][32]
    int       cand_capacity;             // entries per sub-list
    int       pad1;
    psx_extremum* extrema;               // dobuf.extrema
    psx_feature*  features;              // dobuf.features
    float*        desc;                  // dbuf.desc, 128 floats each
    int*          feat_to_ext;           // dobuf.feat_to_ext_map
    int*          ext_nori;              // num_ori per extremum (SoA copy for the scan)
};

// Zero-copy export targets in mapped host memory (nullptr when detached).  Passed to the scan and descriptor
// kernels BY VALUE as a kernel argument: attaching a fresh result buffer per frame then costs no HIP call at all
// (a parameter-block update over PCIe per frame measured 0.9 ms of host time per worker).
struct PsxExport {
    psx_feature*  features;
    float*        desc;
    int*          counts;                // [0]=ext_total [1]=ori_total [2]=ori_raw, pinned host memory
    int           feat_capacity;
    int           desc_capacity;
};

// ExtremaCounters (sift_pyramid.h:21-33), kept in device memory of the context.
struct PsxCounters {
    int ext_ct[PSX_MAX_OCTAVES];   // raw atomic counters (may exceed max_extrema)
    int ext_ps[PSX_MAX_OCTAVES + 1];
    int ext_total;
    int ori_total;
    int ori_raw;                   // orientations before the clamp to ori_capacity
    int flow_error;                // k_pyramid_flow: a dependency wait ran into its bound (the frame is invalid)
    int iext_ct[PSX_MAX_OCTAVES];  // initial extrema per octave before the grid filter (set by k_gf_apply)
};

struct PsxTaps { float g[PSX_GAUSS_ALIGN]; };

// ---- host-side launch helpers implemented in the .hip files ---------------------------------
// columns of the resampled input kept on each side of the plane (>= the largest filter halo, 32)
#define PSX_LEVEL0_PAD 32
struct PsxLevel0Args {
    const void* img; int w, h, is_float;
    float* dst; int W, H, pitch;
    float* tmp; int tmp_pitch;    // resampled input: H rows of roundup(W,64) + 2*PSX_LEVEL0_PAD floats
    float shift;
    PsxTaps taps_h; int span_h;   // dd table, octave 0
    PsxTaps taps_v; int span_v;   // inc table, level 0
    // GaussMode VLFeat_Relative: the interpolated table of level 0 (i_filter row, its odd span) -- the vertical pass is
    // absoluteSourceInterpolated::vert instead of absoluteSource::vert; nullptr otherwise
    const float* v_ifilter = nullptr; int v_ispan = 0;
};
bool psx_level0_interp_ok(const PsxLevel0Args& a);

hipError_t psx_launch_level0(const PsxLevel0Args& a, hipStream_t s);
// per-tap texture coordinates exactly as the reference forms them (pyramid_alt.hip): what psx_launch_level0 runs when
// the image / octave ratio is not a power of two (psx_level0_exact)
hipError_t psx_launch_level0_literal(const PsxLevel0Args& a, hipStream_t s);
bool psx_level0_exact(int w, int h, int W, int H);
hipError_t psx_launch_blur(const float* src, float* dst, int W, int H, int pitch,
                           const PsxTaps& taps, int span,
                           float* half_dst, int half_pitch, hipStream_t s,
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// one plane-to-plane blur (levels >= 1); psx_launch_blur2 runs two independent ones in a single launch
struct PsxBlurJob {
    const float* src; float* dst; float* half_dst;
    int W, H, pitch, half_pitch;
    PsxTaps taps; int span;
};
int psx_blur_grid(int W, int H, int span);
bool psx_blur_pair_ok(int W1, int H1, int W2, int H2, int span, int resident_marching);
hipError_t psx_launch_blur2(const PsxBlurJob& a, const PsxBlurJob& b, hipStream_t s,
                            hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// ---- k_pyramid_flow: every plane-to-plane blur of a frame in ONE launch (pyramid.hip) ----------------------------------
// Work items = (job = (octave, level), 64-column strip, chunk of rows); a persistent grid takes tickets in a host-computed
// topological order; an item may start when the chunks of the source plane that hold its input rows are complete (one
// counter per (job, chunk), bumped by each strip's workgroup after it has drained its write-through stores).
struct PsxFlowJob {                 // 256 bytes, device resident, read with scalar loads
    const float* src; float* dst; float* half_dst;
    int W, H, pitch, half_pitch;
    int nstrips, chunk_rows, nchunks, rsel;     // rsel: index into the radii the kernel is instantiated for
    int cnt_off;                    // first chunk counter of this job
    int dep_cnt_off;                // first chunk counter of the job that writes src (-1: src is complete before the launch)
    int dep_need;                   // counter value that means "chunk complete" (= strips of the producing job)
    int octave, level;
    int pad[13];
    PsxTaps taps;
};
static_assert(sizeof(PsxFlowJob) == 256, "PsxFlowJob layout");
struct PsxFlowItem { unsigned short job, strip, chunk, dep_c0, dep_c1, pad0, pad1, pad2; };   // 16 bytes
#define PSX_FLOW_SHARDS 8           // ticket counters (one per workgroup class blockIdx & 7), each on its own 128-byte line
#define PSX_FLOW_HEAD_INTS (PSX_FLOW_SHARDS * 32)
#define PSX_FLOW_MAX_COUNTERS 4096
#define PSX_FLOW_CNT_STRIDE 32      // ints between two chunk counters: one counter per 128-byte line
struct PsxFlowPlan {
    int njobs, nitems, ncounters, grid;
    PsxFlowJob*  jobs;              // host arrays (malloc), owned by the caller
    PsxFlowItem* items;
};
// jobs: the blur levels 1..L-1 of the octaves first_octave..; false when the configuration is outside what the kernel is
// instantiated for (a radius above 13, too many chunk counters): the caller keeps the launch-per-level schedule
bool psx_flow_plan(const PsxParams& P, const float* inc_filter, const int* inc_span, int first_octave,
                   int resident_blocks, int order, PsxFlowPlan* out);
// state: PSX_FLOW_HEAD_INTS ticket words + ncounters * PSX_FLOW_CNT_STRIDE ints of chunk counters, zeroed before the launch; err: set to 1 by a workgroup
// whose dependency wait ran into its bound (never in a correct run)
hipError_t psx_launch_flow(const PsxFlowJob* d_jobs, const PsxFlowItem* d_items, int nitems, int* d_state, int* d_err,
                           int grid, int ldmode, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                           long long* trace = nullptr);

// every non-default branch of Pyramid::build_pyramid (pyramid_alt.hip)
struct PsxAltArgs {
    const PsxParams* hp;
    const void* img; int w, h, is_float;
    int gauss_mode, scaling_mode, sift_mode;
    float upscale_factor;
    const float *inc_filter, *inc_ifilter, *dd_filter, *abs0_filter, *absN_filter;      // host tables
    const int *inc_span, *inc_ispan, *dd_span, *abs0_span;
    float* up; int up_pitch;     // scratch of psx_launch_level0 (the resampled, padded input: PsxLevel0Args::tmp)
    float* intm;                 // scratch: one plane of octave 0 (pitch x height)
    float* vbuf; int vbuf_pitch; // scratch of the fixed-span modes: pitch + 2*7 columns
    hipError_t (*after_octave)(void* user, int octave);   // e.g. launch the octave's extrema scan
    void* user;
    // blur probe (psx_enable_blur_probe): begin / end timestamps of the fixed-span modes' octave-0 launch; *probe_hit is set
    // when that launch was the one-kernel octave of pyramid_fixed.hip
    hipEvent_t probe_ev0 = nullptr, probe_ev1 = nullptr;
    int* probe_hit = nullptr;
};
hipError_t psx_launch_pyramid_alt(const PsxAltArgs& a, hipStream_t s);
// GaussMode Fixed9 / Fixed15: every derived level of an octave in one launch (pyramid_fixed.hip)
struct PsxFixedOctaveArgs {
    const void* src;            // level 0 of the octave (from_input == 0) or the input image
    int src_w, src_h, is_float; // image size / type (from_input != 0)
    int from_input;             // octave 0: levels 0..5 from the image (x2 upsampling only: psx_fixed_octave0_ok)
    int shift;                  // 4 (Fixed9) or 7 (Fixed15)
    int nlev;                   // levels written: 6 from the image, 5 from a plane
    float* dst;                 // plane of the first level written
    size_t plane;               // floats between two levels
    float* half_dst;            // level 0 of the next octave (every second row / column of level L - 3), or nullptr
    int half_pitch, half_level; // half_level: index of that level among the levels written
    int W, H, pitch;
    float scale;
    const float* taps;          // host table: nlev rows of PSX_GAUSS_ALIGN floats
    hipEvent_t ev0, ev1;        // begin / end timestamps of the dispatch, or nullptr
};
// GaussMode VLFeat_Relative: one fused H + V launch per level (pyramid_interp.hip); fi / ispan: the level's row of the
// interpolated table; psx_blur_interp_ok: the pair count the kernel is instantiated for
bool psx_blur_interp_ok(int ispan);
struct PsxInterpJob {
    const float* src; float* dst; float* half_dst;     // half_dst: level 0 of the next octave (level L - 3 only), or nullptr
    int W, H, pitch, half_pitch;
    const float* fi; int ispan;                         // the level's row of the interpolated table (host), its odd span
};
hipError_t psx_launch_blur_interp(const PsxInterpJob& j, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t psx_launch_blur_interp2(const PsxInterpJob& a, const PsxInterpJob& b, hipStream_t s);
int psx_blur_interp_grid(int W, int H, int ispan);
bool psx_blur_interp_pair_ok(int W1, int H1, int ispan1, int W2, int H2, int ispan2);
bool psx_fixed_octave0_ok(int w, int h, int W, int H);
bool psx_fixed_octave_enabled();
hipError_t psx_launch_fixed_octave(const PsxFixedOctaveArgs& a, hipStream_t s);
hipError_t psx_launch_dog(const float* a, const float* b, float* d, int W, int H, int pitch, hipStream_t s);
// k_extrema scans the tiles of up to PSX_EXT_BATCH octaves in one launch
#define PSX_EXT_BATCH 4
struct PsxExtBatch { int n; int octave[PSX_EXT_BATCH]; int tiles_x[PSX_EXT_BATCH]; int tile_end[PSX_EXT_BATCH]; };
hipError_t psx_launch_extrema_batch(const PsxParams* d_params, const PsxParams& h_params, PsxCounters* d_cnt,
                                    const int* octaves, int n, hipStream_t s);
int psx_extrema_tiles(const PsxParams& h_params, int octave);
hipError_t psx_launch_extrema(const PsxParams* d_params, const PsxParams& h_params, PsxCounters* d_cnt,
                              int octave, hipStream_t s);
hipError_t psx_launch_refine(const PsxParams* d_params, const PsxParams& h_params, PsxCounters* d_cnt, hipStream_t s);
// grid filter (gridfilter.hip)
size_t     psx_gridfilter_scratch_ints(int grid_size);
hipError_t psx_gridfilter_sort_bytes(int total, size_t* bytes);
hipError_t psx_launch_gridfilter(const PsxParams* d_params, PsxCounters* d_cnt, int mode, int total,
                                 int filter_max, unsigned long long* keys_in, unsigned long long* keys_out,
                                 unsigned* vals_in, unsigned* vals_out, void* temp, size_t temp_bytes,
                                 int* scratch, hipStream_t s);
hipError_t psx_launch_orientation(const PsxParams* d_params, PsxCounters* d_cnt, hipStream_t s);
hipError_t psx_launch_scan(const PsxParams* d_params, PsxCounters* d_cnt, const PsxExport& x, hipStream_t s);
hipError_t psx_launch_feature_ptrs(const psx_feature* in, psx_feature_dev* out, int n, float* desc_base, int num_desc,
                                   hipStream_t s);
hipError_t psx_launch_descriptors(const PsxParams* d_params, const PsxCounters* d_cnt, const PsxExport& x, int cus, hipStream_t s);
hipError_t psx_launch_descriptors_alt(const PsxParams* d_params, const PsxCounters* d_cnt, int desc_mode, const PsxExport& x, int cus, hipStream_t s);

// ---- multi-level tile kernel (pyramid_tile.hip, blur_tile_core.h): several consecutive levels of an octave per launch ----
struct PsxTileJob;
// nt: 512 or 1024 threads per workgroup; lds_bytes: the largest LDS need among the jobs (<= 80 KB: two workgroups per CU at nt = 512)
hipError_t psx_launch_blur_tile(const PsxTileJob* d_jobs, int njobs, int grid, size_t lds_bytes, int nt, hipStream_t s,
                                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- small device helpers --------------------------------------------------------------------
__device__ __forceinline__ int psx_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// Blocks with equal (blockIdx % 8) run on the same XCD and share its L2 (MI355X_MICROARCH.md, "Workgroup dispatch"); give
// them contiguous logical ids so that neighbouring strips / tiles, which share halo pixels, hit the same L2.  Bijective
// for any grid size.  Speed only.
__device__ __forceinline__ int psx_xcd_remap(int b, int n)
{
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
