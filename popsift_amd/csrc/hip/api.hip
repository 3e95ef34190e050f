// api.hip -- implementation of the C-ABI declared in include/popsift_hip.h.
//
// One psx_ctx == one Pyramid of the reference (sift_pyramid.h:53-163) plus the state the
// reference keeps in global __device__/__constant__/thread_local symbols (gauss tables
// gauss_filter.cu:18-21, constants sift_constants.cu:19-20, counters and buffers
// sift_pyramid.cu:41-49).  Because nothing is global, any number of contexts can live on one
// device; the batch dispatcher uses that to keep several frames in flight per GPU.
//
// Stream model: every context owns one HIP stream; a frame is one stream-ordered chain
//   memset(counters) -> k_upscale -> k_blur<R,true> -> per octave { k_blur x (L-1) -> k_extrema } -> k_refine
//   -> k_orientation -> k_scan -> k_descriptors
// with no host synchronisation inside the chain (the reference has four blocking counter
// round-trips and four device-wide syncs per image).
#include "psx_internal.h"
#include "blur_tile_core.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

inline int imin(int a, int b) { return a < b ? a : b; }
inline int imax(int a, int b) { return a > b ? a : b; }

// ---- Gauss tables: restatement of gauss_filter.cu:127-371 (host arithmetic) -----------------
int vlfeat_span(float sigma) { return imin((int)(ceilf(4.0f * sigma) + 1), PSX_GAUSS_ALIGN - 1); }
int opencv_span(float sigma)
{
    int span = (int)(roundf(2.0f * 4.0f * sigma + 1.0f)) | 1;
    span >>= 1;
    span += 1;
    return imin(span, PSX_GAUSS_ALIGN - 1);
}
int get_span(int mode, float sigma)
{
    switch (mode) {
    case PSX_GAUSS_VLFEAT_RELATIVE_ALL:
    case PSX_GAUSS_VLFEAT_COMPUTE: return vlfeat_span(sigma);
    case PSX_GAUSS_VLFEAT_RELATIVE: { int s = vlfeat_span(sigma); if ((s & 1) == 0) s += 1; return s; }
    case PSX_GAUSS_OPENCV_COMPUTE: return opencv_span(sigma);
    case PSX_GAUSS_FIXED9: return 5;
    case PSX_GAUSS_FIXED15: return 8;
    default: return -1;
    }
}
void blur_table(int mode, int nlev, const float* sigma, int* span, float* filter)
{
    for (int level = 0; level < nlev; level++)
        span[level] = imin(get_span(mode, sigma[level]), PSX_GAUSS_ALIGN - 1);
    for (int level = 0; level < nlev; level++) {
        const float sig = sigma[level];
        const int spn = span[level];
        float* f = filter + level * PSX_GAUSS_ALIGN;
        double sum = 1.0;
        f[0] = 1.0f;
        for (int x = 1; x < spn; x++) {
            const float val = (float)std::exp(-0.5 * (std::pow(double(x) / sig, 2.0)));
            f[x] = val;
            sum += 2.0f * val;
        }
        for (int x = 0; x < spn; x++) f[x] = (float)(f[x] / sum);
        for (int x = spn; x < PSX_GAUSS_ALIGN; x++) f[x] = 0.0f;
    }
}

} // namespace

constexpr size_t CNT_BLOCK = 512;      // bytes reserved for PsxCounters in front of the flow state and the candidate counters
constexpr size_t FLOW_MAX_BYTES = sizeof(int) * (size_t)(PSX_FLOW_HEAD_INTS + PSX_FLOW_MAX_COUNTERS * PSX_FLOW_CNT_STRIDE);
constexpr size_t CAND_CT_BYTES = sizeof(int) * (size_t)PSX_MAX_OCTAVES * PSX_CAND_SUB * 32;

struct psx_ctx {
    int         device = 0;
    psx_config  cfg{};
    std::string err;
    hipStream_t stream = nullptr;

    float inc_filter[PSX_GAUSS_LEVELS * PSX_GAUSS_ALIGN];
    int   inc_span[PSX_GAUSS_LEVELS];
    float inc_sigma[PSX_GAUSS_LEVELS];
    float dd_filter[PSX_MAX_OCTAVES * PSX_GAUSS_ALIGN];
    int   dd_span[PSX_MAX_OCTAVES];
    float dd_sigma[PSX_MAX_OCTAVES];
    // tables of the alternative pyramid modes (gauss_filter.cu:188-214, 373-410)
    float abs0_filter[PSX_GAUSS_LEVELS * PSX_GAUSS_ALIGN]; int abs0_span[PSX_GAUSS_LEVELS];
    float absN_filter[PSX_GAUSS_LEVELS * PSX_GAUSS_ALIGN]; int absN_span[PSX_GAUSS_LEVELS];
    float inc_ifilter[PSX_GAUSS_LEVELS * PSX_GAUSS_ALIGN]; int inc_ispan[PSX_GAUSS_LEVELS];
    bool  alt_pyramid = false;         // any branch of build_pyramid other than the default one
    float* d_intm = nullptr;           size_t intm_cap = 0;     // scratch planes of the alternative branches
    float* d_vbuf = nullptr;           size_t vbuf_cap = 0;

    int in_w = 0, in_h = 0;
    int octaves_resolved = -1;         // sticky auto-octave value (popsift.cpp:118-122)
    PsxParams    hp{};
    PsxParams*   d_params = nullptr;
    PsxParams*   h_params_pin = nullptr;   // pinned mirror: source of the asynchronous parameter updates
    PsxCounters* d_cnt = nullptr;
    PsxCounters* h_cnt = nullptr;      // pinned
    bool counts_valid = false;
    bool counts_partial = false;       // only ext_total / ori_total valid (export fast path)

    void*       d_input_own = nullptr; size_t input_cap = 0;
    // pageable caller memory is staged through a pinned buffer: hipMemcpyAsync from pageable memory took
    // 4.9 ms for a 2 MB frame (measured), a host memcpy + DMA from pinned memory takes ~0.1 ms
    void*       h_stage = nullptr;     size_t stage_cap = 0;
    hipEvent_t  ev_upload = nullptr;   // the DMA out of h_stage has finished
    const void* d_input = nullptr;     int input_is_float = 0;

    float* d_pyr = nullptr;            size_t pyr_cap = 0;      // floats
    float* d_up = nullptr;             size_t up_cap = 0;       // resampled input of octave 0 (floats)
    int up_pitch = 0;
    psx_iext* d_iext = nullptr;        size_t iext_cap = 0;
    int* d_iext_off = nullptr;         size_t iext_off_cap = 0;
    unsigned long long* d_cand = nullptr; size_t cand_cap = 0;
    int* d_cand_ct = nullptr;          // behind d_cnt in the same allocation
    psx_extremum* d_extrema = nullptr; size_t extrema_cap = 0;
    psx_feature* d_features = nullptr; size_t features_cap = 0;
    float* d_desc = nullptr;           size_t desc_cap = 0;       // floats
    int* d_feat_to_ext = nullptr;      size_t f2e_cap = 0;
    int* d_ext_nori = nullptr;         size_t nori_cap = 0;

    // grid filter scratch (allocated on first use)
    unsigned long long* d_gf_keys = nullptr; size_t gf_keys_cap = 0;    // 2 x total
    unsigned* d_gf_vals = nullptr;           size_t gf_vals_cap = 0;    // 2 x total
    unsigned char* d_gf_temp = nullptr;      size_t gf_temp_cap = 0;
    int* d_gf_scratch = nullptr;             size_t gf_scratch_cap = 0;
    bool filtered = false;             // the grid filter ran on the current frame
    bool null_dl_done = false;         // PSX_NULL_DEVICE_WORK=2: this context's one real download has happened
    bool interleave = false;           // psx_extract: launch an octave's extrema scan right behind its last blur level
    // psx_extract's whole launch chain captured as a hipGraph; valid for one (input pointer, type, size);
    // everything else the kernels read is device resident (PsxParams) or constant per context (taps)
    hipGraphExec_t graph = nullptr;
    const void* graph_input = nullptr; int graph_is_float = 0, graph_w = 0, graph_h = 0;
    bool graph_off = true;             // enabled with POPSIFT_HIP_GRAPH=1; switched off again if a capture fails
    bool ext_launched = false;         // ... which has happened for the current frame

    // zero-copy export
    psx_feature* x_host_feat = nullptr; float* x_host_desc = nullptr;
    psx_feature* x_dev_feat = nullptr;  float* x_dev_desc = nullptr;
    int x_feat_cap = 0, x_desc_cap = 0;
    bool x_registered_feat = false, x_registered_desc = false;
    int* h_xcnt = nullptr;             // pinned [4]: ext_total, ori_total, ori_raw
    // the export targets the frame in flight was LAUNCHED with (psx_orientation): the attach calls may change the
    // targets of the next frame before this one's counters and results have been fetched
    PsxExport fx{};
    psx_feature* fx_host_feat = nullptr; float* fx_host_desc = nullptr;
    bool fx_on = false;

    // MEASUREMENT switch PSX_NULL_DEVICE_WORK (bench.py host_ceiling): 1 = after a context's first frame psx_extract launches
    // nothing -- uploads, the counter read-back and the result downloads still run, on the first frame's results (same
    // counts, same bytes): what the HOST path and PCIe sustain without the kernels; 2 = the DMAs are skipped as well: the
    // host software alone (threads, queues, pools, the per-keypoint record loop).  Results are stale by construction.
    int  null_work = 0;
    bool null_primed = false;
    bool timers = false;
    bool blocking_wait = false;        // psx_set_wait_mode: sleep on an event instead of spinning in hipStreamSynchronize
    hipEvent_t ev_wait = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    // in-pipeline timing of the octave-0 separable-Gaussian launches (psx_enable_blur_probe)
    bool blur_probe = false;
    hipEvent_t ev_blur[2 * PSX_GAUSS_LEVELS] = {};     // [2l], [2l+1]: begin / end of the level-(l+1) kernel
    hipEvent_t ev_x[4] = {};           // the probe also brackets octave 0's level 0 [0,1] and its extrema scan [2,3] (stream events)
    bool probe_ext0 = false;           // octave 0's extrema scan was a launch of its own in the last extraction
    int  blur_probe_n = 0;             // levels timed in the last extraction
    double blur_probe_bytes = 0.0;     // algorithmic bytes per timed launch (8 B per pixel of every plane the launch blurs), averaged
    // k_pyramid_flow (POPSIFT_FLOW: 0 = one launch per level -- the default: the one-launch kernel measured 172 us against
    // 190 us of launches for a single 1080p frame but -10 % throughput with several frames in flight, DESIGN.md 3.1b --,
    // 1 = every blur level of the frame in one launch, 2 = octave 0 by launches, octaves >= 1 in one launch)
    int  flow_mode = 0, flow_ld = 2, flow_order = 0;
    bool flow_on = false;              // a plan exists for the current size
    int  flow_first = 0, flow_nitems = 0, flow_grid = 0, flow_ncnt = 0, flow_njobs = 0;
    size_t flow_bytes = 0;             // ticket words + chunk counters, between PsxCounters and the candidate counters
    double flow_algo_bytes = 0.0;      // algorithmic bytes of the launch (8 B per pixel and blurred plane + 4 B per decimated pixel)
    PsxFlowJob*  d_flow_jobs = nullptr;  size_t flow_jobs_cap = 0;
    PsxFlowItem* d_flow_items = nullptr; size_t flow_items_cap = 0;
    long long*   d_flow_trace = nullptr;     // psx_flow_trace only
    // k_blur_tile (pyramid_tile.hip): the octaves that cannot fill the chip run several levels per launch on LDS-resident
    // tiles -- levels 1..L-3 (+ the decimation) of octave o together with levels L-2..L-1 of octave o-1 in ONE launch.
    // OPT-IN (POPSIFT_TILE=1; default: one launch per level, the diagonal schedule): bit-exact, five pyramid launches
    // for octaves 1-4 instead of thirteen, but measured slower on MI355X (profiles/r05_tile_schedule_ab.txt: single frame
    // 0.467 vs 0.436 ms, 8-context throughput -11 %): a tile is a serial chain of passes whose instruction overhead per
    // 8 x 8 block (addresses, predicates, stores) is ~2x its filter arithmetic, and the halo work is 1.65x.
    // POPSIFT_TILE_MAXPX: largest plane (pixels) that takes the tile kernel; POPSIFT_TILE_TY / POPSIFT_TILE_NT: tile rows
    // (32 / 64) and threads per workgroup (512 / 1024); POPSIFT_TILE_SMALL=0: no 32 x 32 tiles for the tiny octaves
    int  tile_mode = 0, tile_ty = 64, tile_nt = 1024;
    long long tile_maxpx = 3ll << 20;
    bool tile_on = false;              // a tile schedule exists for the current size
    int  tile_first = 0;               // first octave on the tile kernel; the octaves in front keep one launch per level
    struct TileLaunch { int job0, njobs, grid; size_t lds; };
    std::vector<TileLaunch> tile_launches;
    PsxTileJob* d_tile_jobs = nullptr; size_t tile_jobs_cap = 0;
    int  resident_blocks = 1024;       // 4 x compute units
    bool batch_octaves = true;         // diagonal schedule: two octaves' levels in one launch (POPSIFT_BATCH_OCTAVES=0: one plane per launch)
};

namespace {

PsxExport export_of(const psx_ctx* c)
{
    PsxExport x;
    x.features = c->x_dev_feat; x.desc = c->x_dev_desc;
    x.counts = (c->x_dev_feat || c->x_dev_desc) ? c->h_xcnt : nullptr;
    x.feat_capacity = c->x_feat_cap; x.desc_capacity = c->x_desc_cap;
    return x;
}
inline bool exporting(const psx_ctx* c) { return c->x_dev_feat != nullptr || c->x_dev_desc != nullptr; }
inline void snapshot_export(psx_ctx* c)
{
    c->fx = export_of(c);
    c->fx_host_feat = c->x_host_feat; c->fx_host_desc = c->x_host_desc;
    c->fx_on = exporting(c);
}

int fail(psx_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define PSX_HIP(call)                                                                           \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            char buf__[512];                                                                    \
            snprintf(buf__, sizeof(buf__), "%s:%d\n    %s failed: %s", __FILE__, __LINE__, #call, \
                     hipGetErrorString(e__));                                                   \
            return fail(ctx, PSX_ERR_HIP, buf__);                                               \
        }                                                                                       \
    } while (0)

int compute_tables(const psx_config* cfg, float* inc_filter, int* inc_span, float* inc_sigma,
                   float* dd_filter, int* dd_span, float* dd_sigma, std::string* why)
{
    const float sigma0 = cfg->sigma;
    const int levels = cfg->levels;
    if (sigma0 > 2.0) { if (why) *why = "ERROR:  Sigma > 2.0 is not supported."; return PSX_ERR_INVALID; }
    if (levels + 3 > PSX_GAUSS_LEVELS) {
        if (why) *why = "ERROR:  More than 12 levels not supported.";
        return PSX_ERR_INVALID;
    }
    if (get_span(cfg->gauss_mode, 1.0f) < 0) {
        if (why) *why = "ERROR: The mode for computing Gauss filter scan is invalid";
        return PSX_ERR_INVALID;
    }
    memset(inc_filter, 0, sizeof(float) * PSX_GAUSS_LEVELS * PSX_GAUSS_ALIGN);
    memset(inc_sigma, 0, sizeof(float) * PSX_GAUSS_LEVELS);
    memset(dd_filter, 0, sizeof(float) * PSX_MAX_OCTAVES * PSX_GAUSS_ALIGN);
    const int stages = levels + 3;
    const float initial_blur = cfg->assume_initial_blur
                             ? cfg->initial_blur * powf(2.0f, cfg->upscale_factor) : 0.0f;
    inc_sigma[0] = cfg->assume_initial_blur
                 ? sqrtf(fabsf(sigma0 * sigma0 - initial_blur * initial_blur)) : sigma0;
    for (int lvl = 1; lvl < stages; lvl++) {
        const float sigmaP = sigma0 * powf(2.0f, (float)(lvl - 1) / (float)levels);
        const float sigmaS = sigma0 * powf(2.0f, (float)(lvl) / (float)levels);
        inc_sigma[lvl] = sqrtf(sigmaS * sigmaS - sigmaP * sigmaP);
    }
    blur_table(cfg->gauss_mode, PSX_GAUSS_LEVELS, inc_sigma, inc_span, inc_filter);
    for (int oct = 0; oct < PSX_MAX_OCTAVES; oct++) {
        const float oct_sigma = scalbnf(sigma0, oct);
        const float b = sqrtf(fabsf(oct_sigma * oct_sigma - initial_blur * initial_blur));
        dd_sigma[oct] = scalbnf(b, -oct);
    }
    blur_table(cfg->gauss_mode, PSX_MAX_OCTAVES, dd_sigma, dd_span, dd_filter);
    return PSX_OK;
}

// abs_o0, abs_oN and the interpolated (ratio, multiplier) form of the inc table
void compute_alt_tables(psx_ctx* n, float* abs0_sigma = nullptr, float* absN_sigma = nullptr)
{
    const psx_config& c = n->cfg;
    const float sigma0 = c.sigma;
    const int levels = c.levels, stages = levels + 3;
    const float initial_blur = c.assume_initial_blur ? c.initial_blur * powf(2.0f, c.upscale_factor) : 0.0f;
    float s0[PSX_GAUSS_LEVELS] = {0}, sN[PSX_GAUSS_LEVELS] = {0};
    for (int lvl = 0; lvl < stages; lvl++) {
        const float sigmaS = sigma0 * powf(2.0f, (float)(lvl) / (float)levels);
        s0[lvl] = sqrtf(fabsf(sigmaS * sigmaS - initial_blur * initial_blur));
        if (lvl > 0) sN[lvl] = sqrtf(sigmaS * sigmaS - sigma0 * sigma0);
    }
    blur_table(c.gauss_mode, PSX_GAUSS_LEVELS, s0, n->abs0_span, n->abs0_filter);
    blur_table(c.gauss_mode, PSX_GAUSS_LEVELS, sN, n->absN_span, n->absN_filter);
    if (abs0_sigma) memcpy(abs0_sigma, s0, sizeof(s0));
    if (absN_sigma) memcpy(absN_sigma, sN, sizeof(sN));
    for (int level = 0; level < PSX_GAUSS_LEVELS; level++) {       // GaussTable::transformBlurTable
        int isp = n->inc_span[level];
        if (!(isp & 1)) isp += 1;
        n->inc_ispan[level] = isp;
        const float* f = n->inc_filter + level * PSX_GAUSS_ALIGN;
        float* fi = n->inc_ifilter + level * PSX_GAUSS_ALIGN;
        for (int x = 0; x < PSX_GAUSS_ALIGN; x++) fi[x] = 0.0f;
        for (int x = 1; x < isp; x += 2) {
            const float a = f[x], b = f[x + 1];
            fi[x] = a / (a + b);
            fi[x + 1] = a + b;
        }
        fi[0] = f[0];
    }
}

PsxTaps taps_from(const float* row)
{
    PsxTaps t;
    for (int i = 0; i < PSX_GAUSS_ALIGN; i++) t.g[i] = row[i];
    return t;
}

template <class T>
int grow(psx_ctx* ctx, T** ptr, size_t* cap, size_t need)
{
    if (need <= *cap && *ptr) return PSX_OK;
    if (*ptr) { PSX_HIP(hipFree(*ptr)); *ptr = nullptr; *cap = 0; }
    PSX_HIP(hipMalloc(reinterpret_cast<void**>(ptr), need * sizeof(T)));
    *cap = need;
    return PSX_OK;
}

} // namespace

// ---- k_blur_tile: the schedule of a frame size (psx_resize) --------------------------------------------------------------
// The blur levels 1..L-1 of every octave from tile_first on are cut into groups of consecutive levels, each one job of the
// tile kernel: a group ends at level L-3 (the level that feeds the next octave: the next octave must not wait for more) and
// wherever the plan of blur_tile_core.h says "does not fit" (LDS, Q's width).  A job runs in the launch slot behind its
// producer -- the previous group of its octave, or the group of the octave above that ends at level L-3 -- and all jobs of
// a slot share one launch, the deepest octave (the head of the dependency chain) first.  Default configuration: slot s =
// {levels 1..3 of octave first + s, levels 4..5 of octave first + s - 1}.
struct TileSched { std::vector<PsxTileJob> jobs; std::vector<psx_ctx::TileLaunch> launches; std::vector<int> octave, l0, slot; int first = 0; };
// false: nothing to run on the tile kernel (no octave small enough, or a level it is not built for)
static bool tile_schedule(const PsxParams& P, const int* inc_span, const float* inc_filter, int tile_ty, int tile_nt,
                          long long maxpx, TileSched& out)
{
    static const bool small_tiles = [] { const char* e = getenv("POPSIFT_TILE_SMALL"); return !(e != nullptr && e[0] == '0'); }();
    const int L = P.L, D = L - 3;
    int first = 0;
    while (first < P.num_octaves && (long long)P.oct[first].w * P.oct[first].h > maxpx) first++;
    if (first >= P.num_octaves || D < 1) return false;
    struct Group { int octave, l0, nlev, slot; PsxTileJob job; size_t lds; };
    std::vector<Group> groups;
    int feeder_slot = -1;                              // slot of the group that writes level 0 of the octave being planned
    const int rows_cap = (tile_nt >= 1024 ? 8 : 16);
    for (int o = first; o < P.num_octaves; o++) {
        const PsxOctave& oc = P.oct[o];
        // an octave that cannot give every CU a 64-column tile takes 32 x 32 tiles: a tile is a serial chain of passes,
        // what counts for such an octave is the length of that chain, not the halo work it repeats
        int tx = 64, ty = tile_ty;
        if (small_tiles && ((oc.w + 63) / 64) * ((oc.h + ty - 1) / ty) < 200) { tx = 32; ty = 32; }
        int l = 1, prev_slot = feeder_slot, next_feeder = -1;
        while (l < L) {
            const int lmax = l <= D ? D : L - 1;       // a group does not cross level L-3
            int n = lmax - l + 1;
            if (n > PSX_TILE_MAXLEV) n = PSX_TILE_MAXLEV;
            Group g{};
            for (; n >= 1; n--) {
                int radii[PSX_TILE_MAXLEV];
                for (int k = 0; k < n; k++) radii[k] = inc_span[l + k] - 1;
                memset(&g.job, 0, sizeof(g.job));
                g.lds = psx_tile_plan_job(g.job, oc.w, oc.h, oc.pitch, n, radii, tx, ty);
                if (g.lds != 0 && g.job.h.NR <= rows_cap * (tile_nt >> g.job.h.lpr_shift)) break;
            }
            if (n < 1) return false;                   // a level the tile kernel is not built for: keep the launches
            g.octave = o; g.l0 = l; g.nlev = n; g.slot = prev_slot + 1;
            PsxTileHdr& h = g.job.h;
            h.src = oc.data + (size_t)(l - 1) * oc.plane;
            h.half_dst = nullptr; h.half_pitch = 0; h.half_lev = -1;
            for (int k = 0; k < n; k++) {
                g.job.dst[k] = oc.data + (size_t)(l + k) * oc.plane;
                for (int q = 0; q < PSX_GAUSS_ALIGN; q++) g.job.taps[k].g[q] = inc_filter[(l + k) * PSX_GAUSS_ALIGN + q];
            }
            if (l <= D && l + n - 1 == D) {
                next_feeder = g.slot;
                if (o + 1 < P.num_octaves) { h.half_dst = P.oct[o + 1].data; h.half_pitch = P.oct[o + 1].pitch; h.half_lev = D - l; }
            }
            prev_slot = g.slot;
            groups.push_back(g);
            l += n;
        }
        feeder_slot = next_feeder;
    }
    int nslots = 0;
    for (const Group& g : groups) if (g.slot + 1 > nslots) nslots = g.slot + 1;
    for (int s = 0; s < nslots; s++) {
        psx_ctx::TileLaunch ln{(int)out.jobs.size(), 0, 0, 0};
        for (int o = P.num_octaves - 1; o >= first; o--)
            for (const Group& g : groups)
                if (g.slot == s && g.octave == o) {
                    PsxTileJob j = g.job;
                    j.h.block0 = ln.grid;
                    ln.grid += j.h.tiles_x * j.h.tiles_y;
                    if (g.lds > ln.lds) ln.lds = g.lds;
                    ln.njobs++;
                    out.jobs.push_back(j);
                    out.octave.push_back(g.octave); out.l0.push_back(g.l0); out.slot.push_back(s);
                }
        if (ln.njobs > 0) out.launches.push_back(ln);
    }
    out.first = first;
    return !out.jobs.empty();
}

static int plan_tiles(psx_ctx* ctx)
{
    ctx->tile_on = false;
    ctx->tile_launches.clear();
    if (ctx->tile_mode == 0 || ctx->alt_pyramid || ctx->flow_on) return PSX_OK;
    TileSched sc;
    if (!tile_schedule(ctx->hp, ctx->inc_span, ctx->inc_filter, ctx->tile_ty, ctx->tile_nt, ctx->tile_maxpx, sc)) return PSX_OK;
    int rc = grow(ctx, &ctx->d_tile_jobs, &ctx->tile_jobs_cap, sc.jobs.size());
    if (rc != PSX_OK) return rc;
    PSX_HIP(hipMemcpy(ctx->d_tile_jobs, sc.jobs.data(), sizeof(PsxTileJob) * sc.jobs.size(), hipMemcpyHostToDevice));
    ctx->tile_launches = sc.launches;
    ctx->tile_on = true;
    ctx->tile_first = sc.first;
    return PSX_OK;
}

// Host-only self check of a tile schedule (no device needed; tests/test_capi_cpu.py): every level 1..L-1 of every octave
// from the first tiled one on is produced by exactly one job, a job sits in a later launch than the job that writes its
// source plane, the decimated plane of an octave is written by the job that ends at level L-3, workgroup ranges of a
// launch are contiguous.  Returns the number of jobs (0: the kernel would not be used), or a negative code naming the
// violated invariant.  stats (optional, 4 ints): first tiled octave, launches, largest LDS need, largest grid.
extern "C" int psx_tile_selfcheck(int w0, int h0, int num_octaves, int levels, const int* spans, int tile_ty, int tile_nt,
                                  long long maxpx, int* stats)
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
    TileSched sc;
    if (!tile_schedule(P, spans, filt.data(), tile_ty, tile_nt, maxpx, sc)) return 0;
    const int L = P.L, D = L - 3;
    std::vector<int> writer((size_t)num_octaves * L, -1), wslot((size_t)num_octaves * L, -1);
    for (size_t q = 0; q < sc.jobs.size(); q++) {
        const PsxTileJob& j = sc.jobs[q];
        const int o = sc.octave[q], l0 = sc.l0[q];
        if (o < sc.first || o >= num_octaves || l0 < 1 || l0 + j.h.nlev > L) return -2;
        if (j.h.src != P.oct[o].data + (size_t)(l0 - 1) * P.oct[o].plane || j.h.W != P.oct[o].w || j.h.H != P.oct[o].h) return -3;
        for (int k = 0; k < j.h.nlev; k++) {
            if (j.dst[k] != P.oct[o].data + (size_t)(l0 + k) * P.oct[o].plane) return -3;
            if (spans[l0 + k] - 1 > psx_tile_radius(j.lev[k].rsel)) return -4;
            int& wr = writer[(size_t)o * L + l0 + k];
            if (wr != -1) return -5;
            wr = (int)q; wslot[(size_t)o * L + l0 + k] = sc.slot[q];
        }
        const bool feeds = l0 <= D && l0 + j.h.nlev - 1 == D && o + 1 < num_octaves;
        if (feeds != (j.h.half_dst != nullptr)) return -6;
        if (feeds) {
            if (j.h.half_dst != P.oct[o + 1].data || j.h.half_lev != D - l0 || j.h.half_pitch != P.oct[o + 1].pitch) return -6;
            writer[(size_t)(o + 1) * L] = (int)q; wslot[(size_t)(o + 1) * L] = sc.slot[q];
        }
    }
    for (int o = sc.first; o < num_octaves; o++)
        for (int l = 1; l < L; l++) if (writer[(size_t)o * L + l] < 0) return -7;
    for (size_t q = 0; q < sc.jobs.size(); q++) {
        const int o = sc.octave[q], l0 = sc.l0[q];
        const int src_slot = (o == sc.first && l0 == 1) ? -1 : wslot[(size_t)o * L + l0 - 1];
        if (!(o == sc.first && l0 == 1) && (src_slot < 0 || src_slot >= sc.slot[q])) return -8;   // the producer is not in an earlier launch
    }
    size_t maxlds = 0; int maxgrid = 0;
    for (const psx_ctx::TileLaunch& ln : sc.launches) {
        int at = 0;
        for (int q = 0; q < ln.njobs; q++) {
            const PsxTileJob& j = sc.jobs[(size_t)ln.job0 + q];
            if (j.h.block0 != at) return -9;
            at += j.h.tiles_x * j.h.tiles_y;
            if (sc.slot[(size_t)ln.job0 + q] != sc.slot[(size_t)ln.job0]) return -9;
        }
        if (at != ln.grid || ln.lds == 0 || ln.lds > PSX_TILE_LDS_MAX) return -9;
        if (ln.lds > maxlds) maxlds = ln.lds;
        if (ln.grid > maxgrid) maxgrid = ln.grid;
    }
    if (stats) { stats[0] = sc.first; stats[1] = (int)sc.launches.size(); stats[2] = (int)maxlds; stats[3] = maxgrid; }
    return (int)sc.jobs.size();
}

extern "C" {

const char* psx_version(void) { return "popsift-mi355x 0.1 (gfx950, HIP)"; }

int psx_config_default(psx_config* c)
{
    if (!c) return PSX_ERR_INVALID;
    memset(c, 0, sizeof(*c));
    c->octaves = -1;
    c->levels = 3;
    c->sigma = 1.6f;
    c->edge_limit = 10.0f;
    c->threshold = (float)0.04;
    c->upscale_factor = 1.0f;
    c->gauss_mode = PSX_GAUSS_VLFEAT_COMPUTE;
    c->sift_mode = PSX_MODE_POPSIFT;
    c->scaling_mode = PSX_SCALE_DEFAULT;
    c->desc_mode = PSX_DESC_LOOP;
    c->norm_mode = PSX_NORM_ROOTSIFT;
    c->norm_multi = 0;
    c->max_extrema = 100000;
    c->assume_initial_blur = 1;
    c->initial_blur = 0.5f;
    c->filter_max_extrema = -1;
    c->filter_grid_size = 2;
    c->grid_filter_mode = PSX_FILTER_RANDOM;
    return PSX_OK;
}

float psx_peak_threshold(const psx_config* c) { return c->threshold * 0.5f * 255.0f / c->levels; }

int psx_gauss_tables(const psx_config* cfg, float* inc_filter, int* inc_span, float* inc_sigma,
                     float* dd_filter, int* dd_span, float* dd_sigma)
{
    if (!cfg || !inc_filter || !inc_span || !inc_sigma || !dd_filter || !dd_span || !dd_sigma)
        return PSX_ERR_INVALID;
    return compute_tables(cfg, inc_filter, inc_span, inc_sigma, dd_filter, dd_span, dd_sigma, nullptr);
}

// Config::setPrintGaussTables: what init_filter prints (gauss_filter.cu:146-161) and what its device-side
// print_gauss_filter_symbol<<<1,1>>>(10) prints (gauss_filter.cu:24-120), from the host tables.
int psx_print_gauss_tables(const psx_config* cfg, int columns)
{
    if (!cfg) return PSX_ERR_INVALID;
    std::unique_ptr<psx_ctx> n(new psx_ctx);
    n->cfg = *cfg;
    const int rc = compute_tables(cfg, n->inc_filter, n->inc_span, n->inc_sigma, n->dd_filter, n->dd_span, n->dd_sigma, nullptr);
    if (rc != PSX_OK) return rc;
    float s0[PSX_GAUSS_LEVELS], sN[PSX_GAUSS_LEVELS];
    compute_alt_tables(n.get(), s0, sN);
    printf("\n"
           "Upscaling factor: %f (i.e. original image is scaled by a factor of %f)\n"
           "\n"
           "Sigma computations\n"
           "    Initial sigma is %f\n"
           "    Input blurriness is assumed to be %f (scaled to %f)\n",
           cfg->upscale_factor, pow(2.0f, cfg->upscale_factor), cfg->sigma, cfg->initial_blur,
           cfg->initial_blur * pow(2.0f, cfg->upscale_factor));
    const int stages = cfg->levels + 3;
    auto table = [&](int rows, const int* span, const float* sigma, const float* filter, bool split_sigma) {
        for (int lvl = 0; lvl < rows; lvl++) {
            if (split_sigma) { printf("      %d %d ", lvl, span[lvl] + span[lvl] - 1); printf("%2.6f: ", sigma[lvl]); }
            else             printf("      %d %d %2.6f: ", lvl, span[lvl] + span[lvl] - 1, sigma[lvl]);
            const int m = span[lvl] < columns ? span[lvl] : columns;
            for (int x = 0; x < m; x++) printf("%0.8f ", filter[lvl * PSX_GAUSS_ALIGN + x]);
            printf(m < span[lvl] ? "...\n" : "\n");
        }
    };
    printf("\nGauss tables\n      level span sigma : center value -> edge value\n    relative sigma\n");
    table(stages, n->inc_span, n->inc_sigma, n->inc_filter, true);
    printf("\n");
    printf("\nGauss tables for hardware interpolation\n"
           "      level span sigma : center value -> ( interpolation value, multiplier ) [one edge value] \n");
    table(stages, n->inc_ispan, n->inc_sigma, n->inc_ifilter, true);
    printf("\n");
    printf("\nGauss tables\n      level span sigma : center value -> edge value\n"
           "      absolute filters octave 0 (compute level 0, all other levels directly from level 0)\n");
    table(stages, n->abs0_span, s0, n->abs0_filter, false);
    printf("\n      absolute filters other octaves\n      (level 0 via downscaling, all other levels directly from level 0)\n");
    table(stages, n->absN_span, sN, n->absN_filter, false);
    printf("\n");
    printf("    level 0-filters for direct downscaling\n");
    table(PSX_MAX_OCTAVES, n->dd_span, n->dd_sigma, n->dd_filter, false);
    printf("\n");
    fflush(stdout);
    return PSX_OK;
}

const char* psx_last_error(const psx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int psx_create(int device, const psx_config* cfg, psx_ctx** out)
{
    psx_ctx* ctx = nullptr;   // PSX_HIP reports into g_create_error while ctx == nullptr
    if (!cfg || !out) return fail(nullptr, PSX_ERR_INVALID, "psx_create: null argument");
    *out = nullptr;
    psx_config c = *cfg;
    c.levels = imax(2, c.levels);                      // popsift.cpp:86
    if (c.gauss_mode < PSX_GAUSS_VLFEAT_COMPUTE || c.gauss_mode > PSX_GAUSS_FIXED15)
        return fail(nullptr, PSX_ERR_INVALID, "ERROR: The mode for computing Gauss filter scan is invalid");   // gauss_filter.cu:292-296
    if (c.scaling_mode != PSX_SCALE_DEFAULT && c.scaling_mode != PSX_SCALE_DIRECT)
        return fail(nullptr, PSX_ERR_INVALID, "invalid scaling mode");
    if (c.desc_mode < PSX_DESC_LOOP || c.desc_mode > PSX_DESC_NOTILE)
        return fail(nullptr, PSX_ERR_INVALID, "not yet");   // sift_desc.cu:80-82
    // make_octave exists for levels = 3 only (s_pyramid_fixed.cu:270-292)
    if ((c.gauss_mode == PSX_GAUSS_FIXED9 || c.gauss_mode == PSX_GAUSS_FIXED15) && c.levels != 3)
        return fail(nullptr, PSX_ERR_INVALID, "Unsupported number of levels for making all octaves at once");
    if (c.sift_mode != PSX_MODE_POPSIFT && c.sift_mode != PSX_MODE_OPENCV && c.sift_mode != PSX_MODE_VLFEAT)
        return fail(nullptr, PSX_ERR_INVALID, "invalid sift mode");
    if (c.max_extrema <= 0 || c.filter_grid_size <= 0)
        return fail(nullptr, PSX_ERR_INVALID, "invalid max_extrema / filter_grid_size");
    if (c.filter_max_extrema > 0 && c.filter_grid_size > 64)
        return fail(nullptr, PSX_ERR_INVALID, "filter_grid_size > 64 not supported by the HIP grid filter");
    if (c.grid_filter_mode != PSX_FILTER_RANDOM && c.grid_filter_mode != PSX_FILTER_LARGEST_FIRST &&
        c.grid_filter_mode != PSX_FILTER_SMALLEST_FIRST)
        return fail(nullptr, PSX_ERR_INVALID, "invalid grid_filter_mode");

    PSX_HIP(hipSetDevice(device));
    psx_ctx* n = new (std::nothrow) psx_ctx();
    if (!n) return fail(nullptr, PSX_ERR_NOMEM, "out of host memory");
    n->device = device;
    n->cfg = c;
    // opt-in: measured on MI355X / ROCm 7.2 the replayed graph is not faster than the 36 stream launches
    // (single frame 0.63 vs 0.63 ms, throughput equal): kernel-to-kernel dependencies cost the same either way
    { const char* g = getenv("POPSIFT_HIP_GRAPH"); n->graph_off = !(g != nullptr && g[0] == '1'); }
    { const char* g = getenv("POPSIFT_BATCH_OCTAVES"); n->batch_octaves = !(g != nullptr && g[0] == '0'); }
    { const char* g = getenv("PSX_NULL_DEVICE_WORK"); if (g != nullptr && (g[0] == '1' || g[0] == '2')) n->null_work = g[0] - '0'; }
    { const char* g = getenv("POPSIFT_TILE"); if (g != nullptr && (g[0] == '0' || g[0] == '1')) n->tile_mode = g[0] - '0'; }
    { const char* g = getenv("POPSIFT_TILE_TY"); if (g != nullptr) { const int v = atoi(g); if (v >= 8 && v <= 128 && (v & 3) == 0) n->tile_ty = v; } }
    { const char* g = getenv("POPSIFT_TILE_NT"); if (g != nullptr) { const int v = atoi(g); if (v == 512 || v == 1024) n->tile_nt = v; } }
    { const char* g = getenv("POPSIFT_TILE_MAXPX"); if (g != nullptr) { const long long v = atoll(g); if (v >= 0) n->tile_maxpx = v; } }
    std::string why;
    int rc = compute_tables(&n->cfg, n->inc_filter, n->inc_span, n->inc_sigma, n->dd_filter, n->dd_span,
                            n->dd_sigma, &why);
    if (rc != PSX_OK) { delete n; return fail(nullptr, rc, why); }
    compute_alt_tables(n);
    n->alt_pyramid = !(c.scaling_mode == PSX_SCALE_DEFAULT &&
                       (c.gauss_mode == PSX_GAUSS_VLFEAT_COMPUTE || c.gauss_mode == PSX_GAUSS_OPENCV_COMPUTE));
    ctx = nullptr;
#define PSX_HIPC(call)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            std::string m__ = std::string(#call) + " failed: " + hipGetErrorString(e__);         \
            psx_destroy(n);                                                                     \
            return fail(nullptr, PSX_ERR_HIP, m__);                                             \
        }                                                                                       \
    } while (0)
    {
        // POPSIFT_CU_PARTITIONS=P (measurement switch, default off): the contexts of a process take turns over P partitions of
        // the chip, each context's stream masked to its partition.  POPSIFT_CU_PARTITION_MODE: 0 = by XCD (mask bit b belongs to
        // XCD b % 8: partition = a set of whole XCDs with their own L2s), 1 = a slice of the CUs of every XCD.
        static const int parts = [] { const char* e = getenv("POPSIFT_CU_PARTITIONS"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 8 ? v : 0; }();
        static const int pmode = [] { const char* e = getenv("POPSIFT_CU_PARTITION_MODE"); return e ? atoi(e) : 0; }();
        static std::atomic<int> next_part{0};
        int cus = 0;
        if (parts > 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus >= 64) {
            const int part = next_part.fetch_add(1) % parts;
            uint32_t mask[16] = {0};
            const int nbits = cus < 512 ? cus : 512;
            for (int b = 0; b < nbits; b++) {
                const int xcd = b % 8, cu = b / 8;
                const bool mine = pmode == 0 ? (xcd * parts / 8 == part) : (cu % parts == part);
                if (mine) mask[b / 32] |= 1u << (b % 32);
            }
            PSX_HIPC(hipExtStreamCreateWithCUMask(&n->stream, (uint32_t)((nbits + 31) / 32), mask));
        } else
            PSX_HIPC(hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking));
    }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) n->resident_blocks = 4 * cus; }
    PSX_HIPC(hipMalloc(reinterpret_cast<void**>(&n->d_params), sizeof(PsxParams)));
    PSX_HIPC(hipHostMalloc(reinterpret_cast<void**>(&n->h_params_pin), sizeof(PsxParams), hipHostMallocDefault));
    // frame counters and the candidate sub-list counters in ONE allocation: one fill kernel clears both per frame
    static_assert(sizeof(PsxCounters) <= CNT_BLOCK, "PsxCounters outgrew its slot");
    // layout: [PsxCounters | flow state (flow_bytes, set per size) | candidate counters]; the per-frame fill covers the
    // used prefix of it
    PSX_HIPC(hipMalloc(reinterpret_cast<void**>(&n->d_cnt), CNT_BLOCK + FLOW_MAX_BYTES + CAND_CT_BYTES));
    n->d_cand_ct = reinterpret_cast<int*>(reinterpret_cast<char*>(n->d_cnt) + CNT_BLOCK);
    {
        const char* g = getenv("POPSIFT_FLOW");
        if (g != nullptr && g[0] >= '0' && g[0] <= '2' && g[1] == 0) n->flow_mode = g[0] - '0';
        else if (g != nullptr && g[0] != 0) {          // a mislabelled A/B run is worse than no run
            psx_destroy(n);
            return fail(nullptr, PSX_ERR_INVALID, std::string("POPSIFT_FLOW=") + g + ": valid values are 0 (one launch per level), 1 (every level in one launch), 2 (octave 0 by launches)");
        }
    }
    { const char* g = getenv("POPSIFT_FLOW_LD"); if (g != nullptr && (g[0] == '1' || g[0] == '2')) n->flow_ld = g[0] - '0'; }
    { const char* g = getenv("POPSIFT_FLOW_ORDER"); if (g != nullptr && g[0] >= '0' && g[0] <= '2') n->flow_order = g[0] - '0'; }
    PSX_HIPC(hipHostMalloc(reinterpret_cast<void**>(&n->h_cnt), sizeof(PsxCounters), hipHostMallocDefault));
    PSX_HIPC(hipMemset(n->d_cnt, 0, CNT_BLOCK + FLOW_MAX_BYTES + CAND_CT_BYTES));
    PSX_HIPC(hipHostMalloc(reinterpret_cast<void**>(&n->h_xcnt), 4 * sizeof(int), hipHostMallocDefault));
    n->h_xcnt[0] = n->h_xcnt[1] = n->h_xcnt[2] = n->h_xcnt[3] = 0;
    for (int i = 0; i < 5; i++) PSX_HIPC(hipEventCreate(&n->ev[i]));
    PSX_HIPC(hipEventCreate(&n->ev_t0));
    PSX_HIPC(hipEventCreate(&n->ev_t1));
#undef PSX_HIPC
    *out = n;
    return PSX_OK;
}

int psx_destroy(psx_ctx* ctx)
{
    if (!ctx) return PSX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_params); (void)hipFree(ctx->d_cnt);
    if (ctx->h_params_pin) (void)hipHostFree(ctx->h_params_pin);
    if (ctx->h_cnt) (void)hipHostFree(ctx->h_cnt);
    if (ctx->x_registered_feat) (void)hipHostUnregister(ctx->x_host_feat);
    if (ctx->x_registered_desc) (void)hipHostUnregister(ctx->x_host_desc);
    if (ctx->h_xcnt) (void)hipHostFree(ctx->h_xcnt);
    (void)hipFree(ctx->d_input_own); (void)hipFree(ctx->d_pyr); (void)hipFree(ctx->d_up);
    (void)hipFree(ctx->d_intm); (void)hipFree(ctx->d_vbuf);
    (void)hipFree(ctx->d_flow_jobs); (void)hipFree(ctx->d_flow_items);
    (void)hipFree(ctx->d_tile_jobs);
    (void)hipFree(ctx->d_gf_keys); (void)hipFree(ctx->d_gf_vals); (void)hipFree(ctx->d_gf_temp);
    (void)hipFree(ctx->d_gf_scratch);
    (void)hipFree(ctx->d_iext); (void)hipFree(ctx->d_iext_off); (void)hipFree(ctx->d_cand);
    (void)hipFree(ctx->d_extrema); (void)hipFree(ctx->d_features);
    (void)hipFree(ctx->d_desc); (void)hipFree(ctx->d_feat_to_ext); (void)hipFree(ctx->d_ext_nori);
    for (int i = 0; i < 5; i++) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
    if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
    for (int i = 0; i < 2 * PSX_GAUSS_LEVELS; i++) if (ctx->ev_blur[i]) (void)hipEventDestroy(ctx->ev_blur[i]);
    for (int i = 0; i < 4; i++) if (ctx->ev_x[i]) (void)hipEventDestroy(ctx->ev_x[i]);
    if (ctx->graph) (void)hipGraphExecDestroy(ctx->graph);
    if (ctx->ev_wait) (void)hipEventDestroy(ctx->ev_wait);
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PSX_OK;
}

int psx_resize(psx_ctx* ctx, int w, int h)
{
    if (!ctx) return PSX_ERR_INVALID;
    if (w <= 0 || h <= 0) return fail(ctx, PSX_ERR_INVALID, "psx_resize: non-positive image size");
    if (w == ctx->in_w && h == ctx->in_h && ctx->d_pyr) return PSX_OK;
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->graph) { (void)hipGraphExecDestroy(ctx->graph); ctx->graph = nullptr; }

    const psx_config& c = ctx->cfg;
    // PopSift::private_apply_scale_factor, popsift.cpp:109-126
    const float scaleFactor = 1.0f / powf(2.0f, -c.upscale_factor);
    if (ctx->octaves_resolved < 0) {
        if (c.octaves < 0)
            ctx->octaves_resolved = imax((int)(floorf(logf((float)imin(w, h)) / logf(2.0f)) - 3.0f + scaleFactor), 1);
        else
            ctx->octaves_resolved = c.octaves;
        ctx->octaves_resolved = imin(imax(ctx->octaves_resolved, 1), PSX_MAX_OCTAVES);
    }
    int ow = (int)ceilf(w * scaleFactor);
    int oh = (int)ceilf(h * scaleFactor);
    if (ow <= 0 || oh <= 0) return fail(ctx, PSX_ERR_INVALID, "psx_resize: scaled image is empty");
    // kernels address a plane (and the padded resampled input) with 32-bit byte offsets from its base
    if (((size_t)((ow + 63) & ~63) + 2 * PSX_LEVEL0_PAD) * (size_t)oh * sizeof(float) >= ((size_t)1 << 32))
        return fail(ctx, PSX_ERR_INVALID, "psx_resize: octave 0 plane of 4 GiB or more is not supported");
    // extremum candidates are packed as y << 32 | z << 24 | x
    if (ow >= (1 << 24)) return fail(ctx, PSX_ERR_INVALID, "psx_resize: octave 0 wider than 2^24 - 1 columns is not supported");

    PsxParams& P = ctx->hp;
    memset(&P, 0, sizeof(P));
    P.num_octaves = ctx->octaves_resolved;
    P.levels = c.levels;
    P.L = c.levels + 3;
    P.sift_mode = c.sift_mode;
    P.norm_mode = c.norm_mode;
    P.norm_multi = c.norm_multi;
    P.max_extrema = c.max_extrema;
    P.grid_size = c.filter_grid_size;
    P.up_fac = (int)c.upscale_factor;
    P.sigma0 = c.sigma;
    P.sigma_k = powf(2.0f, 1.0f / c.levels);            // sift_constants.cu:27
    P.threshold = psx_peak_threshold(&c);
    P.edge_limit = c.edge_limit;

    size_t total = 0;
    size_t offs[PSX_MAX_OCTAVES];
    for (int o = 0; o < P.num_octaves; o++) {
        PsxOctave& oc = P.oct[o];
        oc.w = ow; oc.h = oh;
        oc.pitch = (ow + 63) & ~63;
        oc.plane = (size_t)oc.pitch * oh;
        offs[o] = total;
        total += oc.plane * P.L;
        P.w_grid_div[o] = float(ow) / c.filter_grid_size;   // sift_octave.cu:40-41
        P.h_grid_div[o] = float(oh) / c.filter_grid_size;
        ow = (int)ceilf(ow / 2.0f);                          // sift_pyramid.cu:132-133
        oh = (int)ceilf(oh / 2.0f);
    }
    total += 64;   // slack: vector loads never run past the last plane
    int rc;
    if ((rc = grow(ctx, &ctx->d_pyr, &ctx->pyr_cap, total)) != PSX_OK) return rc;
    ctx->up_pitch = ((P.oct[0].w + 63) / 64) * 64 + 2 * PSX_LEVEL0_PAD;
    if ((rc = grow(ctx, &ctx->d_up, &ctx->up_cap, (size_t)ctx->up_pitch * P.oct[0].h)) != PSX_OK) return rc;
    for (int o = 0; o < P.num_octaves; o++) P.oct[o].data = ctx->d_pyr + offs[o];
    if (ctx->alt_pyramid) {
        if ((rc = grow(ctx, &ctx->d_intm, &ctx->intm_cap, P.oct[0].plane + 64)) != PSX_OK) return rc;
        if ((rc = grow(ctx, &ctx->d_vbuf, &ctx->vbuf_cap, (size_t)(P.oct[0].pitch + 64) * P.oct[0].h)) != PSX_OK) return rc;
    }

    // Extrema buffers are sized for the worst case (max_extrema per octave, 100 B per entry); the descriptor
    // buffers start at the reference's size, max(2 max_extrema, 1.25 max_extrema) entries of 512 B
    // (sift_pyramid.cu:186-209), never shrink, and grow after the counter read-back when a frame needed more
    // (regrow_descriptors below; the reference reallocates in Pyramid::reallocExtrema the same way).
    const size_t iext_need = (size_t)P.num_octaves * c.max_extrema;
    const size_t ori_floor = (size_t)imax(2 * c.max_extrema, c.max_extrema + c.max_extrema / 4);
    const size_t ori_need = ctx->desc_cap / 128 > ori_floor ? ctx->desc_cap / 128 : ori_floor;
    if ((rc = grow(ctx, &ctx->d_iext, &ctx->iext_cap, iext_need)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_iext_off, &ctx->iext_off_cap, iext_need)) != PSX_OK) return rc;
    // candidates before refinement: ~1.3x the survivors on natural images; room for 4x the cap per octave,
    // split over PSX_CAND_SUB sub-lists; a candidate that finds its sub-list full is refined in place
    // k_pyramid_flow: work list of this size (default pyramid modes only), and where the candidate counters start behind
    // its ticket words and chunk counters
    ctx->flow_on = false; ctx->flow_bytes = 0;
    if (!ctx->alt_pyramid && ctx->flow_mode != 0) {
        PsxFlowPlan plan{};
        const int first = ctx->flow_mode == 2 ? 1 : 0;
        if (first < P.num_octaves && psx_flow_plan(P, ctx->inc_filter, ctx->inc_span, first, ctx->resident_blocks, ctx->flow_order, &plan)) {
            bool ok = grow(ctx, &ctx->d_flow_jobs, &ctx->flow_jobs_cap, (size_t)plan.njobs) == PSX_OK &&
                      grow(ctx, &ctx->d_flow_items, &ctx->flow_items_cap, (size_t)plan.nitems) == PSX_OK;
            ok = ok && hipMemcpy(ctx->d_flow_jobs, plan.jobs, sizeof(PsxFlowJob) * (size_t)plan.njobs, hipMemcpyHostToDevice) == hipSuccess &&
                       hipMemcpy(ctx->d_flow_items, plan.items, sizeof(PsxFlowItem) * (size_t)plan.nitems, hipMemcpyHostToDevice) == hipSuccess;
            if (ok) {
                ctx->flow_on = true; ctx->flow_first = first; ctx->flow_nitems = plan.nitems; ctx->flow_grid = plan.grid;
                ctx->flow_ncnt = plan.ncounters; ctx->flow_njobs = plan.njobs;
                ctx->flow_bytes = sizeof(int) * ((size_t)PSX_FLOW_HEAD_INTS + (size_t)plan.ncounters * PSX_FLOW_CNT_STRIDE);
                double by = 0.0;
                for (int q = 0; q < plan.njobs; q++) {
                    by += 8.0 * (double)plan.jobs[q].W * plan.jobs[q].H;
                    if (plan.jobs[q].half_dst) by += 4.0 * (double)((plan.jobs[q].W + 1) / 2) * ((plan.jobs[q].H + 1) / 2);
                }
                ctx->flow_algo_bytes = by;
            }
            free(plan.jobs); free(plan.items);
            if (!ok) return fail(ctx, PSX_ERR_HIP, "psx_resize: could not upload the pyramid work list");
        }
    }
    { const int trc = plan_tiles(ctx); if (trc != PSX_OK) return trc; }
    ctx->d_cand_ct = reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->d_cnt) + CNT_BLOCK + ctx->flow_bytes);
    P.cand_capacity = (4 * c.max_extrema + PSX_CAND_SUB - 1) / PSX_CAND_SUB;
    if ((rc = grow(ctx, &ctx->d_cand, &ctx->cand_cap, (size_t)P.num_octaves * PSX_CAND_SUB * P.cand_capacity)) != PSX_OK) return rc;
    P.cand_ct = ctx->d_cand_ct;
    if ((rc = grow(ctx, &ctx->d_extrema, &ctx->extrema_cap, iext_need)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_features, &ctx->features_cap, iext_need)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_desc, &ctx->desc_cap, ori_need * 128)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_feat_to_ext, &ctx->f2e_cap, ori_need)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_ext_nori, &ctx->nori_cap, iext_need + 64)) != PSX_OK) return rc;
    for (int o = 0; o < P.num_octaves; o++) {
        P.iext[o] = ctx->d_iext + (size_t)o * c.max_extrema;
        P.iext_off[o] = ctx->d_iext_off + (size_t)o * c.max_extrema;
        P.cand[o] = ctx->d_cand + (size_t)o * PSX_CAND_SUB * P.cand_capacity;
    }
    P.ext_capacity = (int)iext_need;
    P.ori_capacity = (int)ori_need;
    P.extrema = ctx->d_extrema;
    P.features = ctx->d_features;
    P.desc = ctx->d_desc;
    P.feat_to_ext = ctx->d_feat_to_ext;
    P.ext_nori = ctx->d_ext_nori;

    PSX_HIP(hipMemcpy(ctx->d_params, &P, sizeof(P), hipMemcpyHostToDevice));
    ctx->in_w = w; ctx->in_h = h;
    ctx->counts_valid = false;
    return PSX_OK;
}

int psx_num_octaves(const psx_ctx* ctx) { return ctx ? ctx->hp.num_octaves : 0; }
int psx_num_levels(const psx_ctx* ctx) { return ctx ? ctx->hp.L : 0; }
int psx_octave_dims(const psx_ctx* ctx, int o, int* w, int* h)
{
    if (!ctx || o < 0 || o >= ctx->hp.num_octaves) return PSX_ERR_INVALID;
    if (w) *w = ctx->hp.oct[o].w;
    if (h) *h = ctx->hp.oct[o].h;
    return PSX_OK;
}

static int upload_common(psx_ctx* ctx, const void* host, int w, int h, int is_float, bool known_pinned = false)
{
    if (!ctx || !host) return PSX_ERR_INVALID;
    int rc = psx_resize(ctx, w, h);
    if (rc != PSX_OK) return rc;
    PSX_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)w * h * (is_float ? 4 : 1);
    if (bytes > ctx->input_cap) {
        PSX_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_input_own) PSX_HIP(hipFree(ctx->d_input_own));
        ctx->d_input_own = nullptr; ctx->input_cap = 0;
        PSX_HIP(hipMalloc(&ctx->d_input_own, bytes + 64));
        ctx->input_cap = bytes;
    }
    const void* src = host;
    hipPointerAttribute_t attr;
    // hipPointerGetAttributes costs ~0.2 ms per call in a process with many mappings: callers that KNOW their
    // buffer is pinned (psx_host_alloc) say so
    if (!known_pinned && (hipPointerGetAttributes(&attr, host) != hipSuccess || attr.type == hipMemoryTypeUnregistered)) {
        (void)hipGetLastError();
        if (!ctx->ev_upload) PSX_HIP(hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
        else PSX_HIP(hipEventSynchronize(ctx->ev_upload));            // previous DMA out of the staging buffer
        if (bytes > ctx->stage_cap) {
            if (ctx->h_stage) PSX_HIP(hipHostFree(ctx->h_stage));
            ctx->h_stage = nullptr; ctx->stage_cap = 0;
            PSX_HIP(hipHostMalloc(&ctx->h_stage, bytes, hipHostMallocDefault));
            ctx->stage_cap = bytes;
        }
        memcpy(ctx->h_stage, host, bytes);
        src = ctx->h_stage;
    }
    // DMA engine, not a kernel: a copy kernel that reads the mapped host image over PCIe itself was measured
    // (GPU-initiated reads are slow: +0.8 ms per frame in flight, -8 % end-to-end throughput)
    if (!(ctx->null_work == 2 && ctx->null_primed))        // PSX_NULL_DEVICE_WORK=2: host software only, no DMA
        PSX_HIP(hipMemcpyAsync(ctx->d_input_own, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (src == ctx->h_stage) PSX_HIP(hipEventRecord(ctx->ev_upload, ctx->stream));
    ctx->d_input = ctx->d_input_own;
    ctx->input_is_float = is_float;
    return PSX_OK;
}

int psx_upload_u8(psx_ctx* ctx, const uint8_t* host, int w, int h) { return upload_common(ctx, host, w, h, 0); }
int psx_upload_f32(psx_ctx* ctx, const float* host, int w, int h) { return upload_common(ctx, host, w, h, 1); }
int psx_upload_pinned(psx_ctx* ctx, const void* pinned_host, int w, int h, int is_float)
{
    return upload_common(ctx, pinned_host, w, h, is_float ? 1 : 0, true);
}

int psx_set_input_dev(psx_ctx* ctx, const void* dev_ptr, int w, int h, int is_float)
{
    if (!ctx || !dev_ptr) return PSX_ERR_INVALID;
    int rc = psx_resize(ctx, w, h);
    if (rc != PSX_OK) return rc;
    ctx->d_input = dev_ptr;
    ctx->input_is_float = is_float ? 1 : 0;
    return PSX_OK;
}

static PsxBlurJob blur_job(const psx_ctx* ctx, int o, int level)
{
    const PsxParams& P = ctx->hp;
    const PsxOctave& oc = P.oct[o];
    PsxBlurJob j;
    j.src = oc.data + (size_t)(level - 1) * oc.plane;
    j.dst = oc.data + (size_t)level * oc.plane;
    j.half_dst = nullptr; j.half_pitch = 0;
    if (level == P.L - 3 && o + 1 < P.num_octaves) {      // PREV_LEVEL 3, s_pyramid_build.cu:21,228
        j.half_dst = P.oct[o + 1].data;
        j.half_pitch = P.oct[o + 1].pitch;
    }
    j.W = oc.w; j.H = oc.h; j.pitch = oc.pitch;
    j.taps = taps_from(ctx->inc_filter + level * PSX_GAUSS_ALIGN);
    j.span = ctx->inc_span[level];
    return j;
}

static int launch_blur_level(psx_ctx* ctx, int o, int level, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr)
{
    const PsxBlurJob j = blur_job(ctx, o, level);
    PSX_HIP(psx_launch_blur(j.src, j.dst, j.W, j.H, j.pitch, j.taps, j.span, j.half_dst, j.half_pitch, ctx->stream, ev0, ev1));
    return PSX_OK;
}

// extrema scans of a set of octaves, PSX_EXT_BATCH per launch
static int launch_extrema_set(psx_ctx* ctx, const int* octaves, int n)
{
    for (int i = 0; i < n; i += PSX_EXT_BATCH) {
        const int m = n - i < PSX_EXT_BATCH ? n - i : PSX_EXT_BATCH;
        PSX_HIP(psx_launch_extrema_batch(ctx->d_params, ctx->hp, ctx->d_cnt, octaves + i, m, ctx->stream));
    }
    return PSX_OK;
}

int psx_build_pyramid(psx_ctx* ctx)
{
    if (!ctx) return PSX_ERR_INVALID;
    if (!ctx->d_input || !ctx->d_pyr) return fail(ctx, PSX_ERR_STATE, "psx_build_pyramid: no input image");
    PSX_HIP(hipSetDevice(ctx->device));
    const PsxParams& P = ctx->hp;
    const psx_config& c = ctx->cfg;
    ctx->counts_valid = false;
    if (ctx->timers) PSX_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    // Pyramid::reset_extrema_mgmt, sift_pyramid.cu:364-371
    PSX_HIP(hipMemsetAsync(ctx->d_cnt, 0, CNT_BLOCK + ctx->flow_bytes + sizeof(int) * (size_t)P.num_octaves * PSX_CAND_SUB * 32, ctx->stream));

    if (ctx->alt_pyramid) {
        PsxAltArgs q;
        q.hp = &ctx->hp;
        q.img = ctx->d_input; q.w = ctx->in_w; q.h = ctx->in_h; q.is_float = ctx->input_is_float;
        q.gauss_mode = c.gauss_mode; q.scaling_mode = c.scaling_mode; q.sift_mode = c.sift_mode;
        q.upscale_factor = c.upscale_factor;
        q.inc_filter = ctx->inc_filter; q.inc_ifilter = ctx->inc_ifilter; q.dd_filter = ctx->dd_filter;
        q.abs0_filter = ctx->abs0_filter; q.absN_filter = ctx->absN_filter;
        q.inc_span = ctx->inc_span; q.inc_ispan = ctx->inc_ispan; q.dd_span = ctx->dd_span; q.abs0_span = ctx->abs0_span;
        q.up = ctx->d_up; q.up_pitch = ctx->up_pitch;
        q.intm = ctx->d_intm; q.vbuf = ctx->d_vbuf; q.vbuf_pitch = P.oct[0].pitch + 64;
        q.user = ctx;
        q.after_octave = ctx->interleave ? +[](void* u, int o) -> hipError_t {
            psx_ctx* cx = static_cast<psx_ctx*>(u);
            return psx_launch_extrema(cx->d_params, cx->hp, cx->d_cnt, o, cx->stream);
        } : nullptr;
        ctx->ext_launched = false;
        int probe_hit = 0;
        if (ctx->blur_probe) { q.probe_ev0 = ctx->ev_blur[0]; q.probe_ev1 = ctx->ev_blur[1]; q.probe_hit = &probe_hit; ctx->blur_probe_n = 0; }
        const hipError_t e = psx_launch_pyramid_alt(q, ctx->stream);
        if (probe_hit) {
            // Fixed9 / Fixed15: the one-kernel octave 0 (six planes written from the input image: 24 B per pixel + the image)
            ctx->blur_probe_n = 1;
            ctx->blur_probe_bytes = 24.0 * (double)P.oct[0].w * P.oct[0].h + (double)ctx->in_w * ctx->in_h * (ctx->input_is_float ? 4 : 1);
        }
        if (e == hipErrorInvalidValue) return fail(ctx, PSX_ERR_INVALID, "Unsupported number of levels for making all octaves at once");
        PSX_HIP(e);
        ctx->ext_launched = ctx->interleave;
        if (ctx->timers) PSX_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
        return PSX_OK;
    }

    PsxLevel0Args a;
    a.img = ctx->d_input; a.w = ctx->in_w; a.h = ctx->in_h; a.is_float = ctx->input_is_float;
    a.dst = P.oct[0].data; a.W = P.oct[0].w; a.H = P.oct[0].h; a.pitch = P.oct[0].pitch;
    a.tmp = ctx->d_up; a.tmp_pitch = ctx->up_pitch;
    a.shift = 0.5f;                                                    // s_pyramid_build.cu:109-114
    if (c.sift_mode == PSX_MODE_POPSIFT || c.sift_mode == PSX_MODE_VLFEAT)
        a.shift = 0.5f * powf(2.0f, c.upscale_factor - 0);
    a.taps_h = taps_from(ctx->dd_filter); a.span_h = ctx->dd_span[0];
    a.taps_v = taps_from(ctx->inc_filter); a.span_v = ctx->inc_span[0];
    if (ctx->blur_probe) PSX_HIP(hipEventRecord(ctx->ev_x[0], ctx->stream));
    PSX_HIP(psx_launch_level0(a, ctx->stream));
    if (ctx->blur_probe) PSX_HIP(hipEventRecord(ctx->ev_x[1], ctx->stream));

    ctx->ext_launched = false;
    const bool probe = ctx->blur_probe;
    if (ctx->flow_on) {
        // k_pyramid_flow: octave 0's levels by launches first when the plan starts at octave 1 (POPSIFT_FLOW=2), then every
        // remaining blur level of the frame in ONE launch with device-side dependencies; the extrema scans follow in
        // psx_find_extrema
        for (int level = 1; ctx->flow_first > 0 && level < P.L; level++) {
            hipEvent_t e0 = probe ? ctx->ev_blur[2 * (level - 1)] : nullptr, e1 = probe ? ctx->ev_blur[2 * (level - 1) + 1] : nullptr;
            const int rc = launch_blur_level(ctx, 0, level, e0, e1);
            if (rc != PSX_OK) return rc;
        }
        const bool pf = probe && ctx->flow_first == 0;
        int* state = reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->d_cnt) + CNT_BLOCK);
        PSX_HIP(psx_launch_flow(ctx->d_flow_jobs, ctx->d_flow_items, ctx->flow_nitems, state, &ctx->d_cnt->flow_error,
                                ctx->flow_grid, ctx->flow_ld, ctx->stream, pf ? ctx->ev_blur[0] : nullptr, pf ? ctx->ev_blur[1] : nullptr,
                                ctx->d_flow_trace));
        if (probe) {
            if (pf) { ctx->blur_probe_n = 1; ctx->blur_probe_bytes = ctx->flow_algo_bytes; }
            else    { ctx->blur_probe_n = P.L - 1; ctx->blur_probe_bytes = 8.0 * (double)P.oct[0].w * P.oct[0].h; }
        }
        if (ctx->timers) PSX_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
        return PSX_OK;
    }
    if (ctx->tile_on && !(probe && ctx->tile_first == 0)) {
        // octaves in front of tile_first: one launch per level, the octave's extrema scan right behind its last level;
        // then the tile launches (several levels of up to two octaves each); then the scans of the small octaves
        double probe_bytes = 0.0;
        int deferred[PSX_MAX_OCTAVES], ndef = 0;
        auto scan_after = [&](int o) -> int {
            if (!ctx->interleave) return PSX_OK;
            if (psx_extrema_tiles(P, o) >= ctx->resident_blocks) {
                const bool px = probe && o == 0;
                if (px) PSX_HIP(hipEventRecord(ctx->ev_x[2], ctx->stream));
                PSX_HIP(psx_launch_extrema(ctx->d_params, ctx->hp, ctx->d_cnt, o, ctx->stream));
                if (px) { PSX_HIP(hipEventRecord(ctx->ev_x[3], ctx->stream)); ctx->probe_ext0 = true; }
            } else
                deferred[ndef++] = o;
            return PSX_OK;
        };
        for (int o = 0; o < ctx->tile_first; o++) {
            for (int level = 1; level < P.L; level++) {
                const bool pl = probe && o == 0;
                hipEvent_t e0 = pl ? ctx->ev_blur[2 * (level - 1)] : nullptr, e1 = pl ? ctx->ev_blur[2 * (level - 1) + 1] : nullptr;
                const int rc = launch_blur_level(ctx, o, level, e0, e1);
                if (rc != PSX_OK) return rc;
                if (pl) probe_bytes += 8.0 * (double)P.oct[o].w * P.oct[o].h;
            }
            const int rc = scan_after(o);
            if (rc != PSX_OK) return rc;
        }
        for (const psx_ctx::TileLaunch& ln : ctx->tile_launches)
            PSX_HIP(psx_launch_blur_tile(ctx->d_tile_jobs + ln.job0, ln.njobs, ln.grid, ln.lds, ctx->tile_nt, ctx->stream));
        for (int o = ctx->tile_first; o < P.num_octaves; o++) { const int rc = scan_after(o); if (rc != PSX_OK) return rc; }
        if (ndef > 0) { const int rc = launch_extrema_set(ctx, deferred, ndef); if (rc != PSX_OK) return rc; }
        if (probe) { ctx->blur_probe_n = P.L - 1; ctx->blur_probe_bytes = probe_bytes / (P.L - 1); }
        ctx->ext_launched = ctx->interleave;
        if (ctx->timers) PSX_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
        return PSX_OK;
    }
    // Diagonal schedule.  Level l of octave o only needs level l-1 of the same octave, and level 0 of octave o+1
    // is written by the launch of level D = L-3 of octave o; so octave o+1 may start D launch slots after octave
    // o, and (o+1, l) then shares ONE launch with (o, l+D) -- if the two planes fit one round of resident
    // workgroups (4 per CU).  The small octaves, chains of ~5 us launches on their own, ride along with the octave
    // above.  Where the pair does not fit (octave 0 / 1 of a 1080p frame) the smaller plane would only queue behind
    // a full chip, run on the larger radius' kernel for nothing and push the larger octave's planes out of L2
    // between its own levels: there octave o+1 starts after octave o's last level.
    const int D = P.L - 3;
    int t0[PSX_MAX_OCTAVES];                       // level l of octave o runs in launch slot t0[o] + l
    t0[0] = 0;
    for (int o = 0; o + 1 < P.num_octaves; o++) {
        const int span = ctx->inc_span[P.L - 1];
        const bool fits = ctx->batch_octaves &&
            psx_blur_pair_ok(P.oct[o].w, P.oct[o].h, P.oct[o + 1].w, P.oct[o + 1].h, span, ctx->resident_blocks);
        t0[o + 1] = t0[o] + (fits ? D : P.L - 1);
    }
    const int T = t0[P.num_octaves - 1] + P.L - 1;
    double probe_bytes = 0.0;
    int deferred[PSX_MAX_OCTAVES], ndef = 0;
    for (int t = 1; t <= T; t++) {
        int jo[4], nj = 0;
        for (int o = 0; o < P.num_octaves && nj < 4; o++) {
            const int l = t - t0[o];
            if (l >= 1 && l <= P.L - 1) jo[nj++] = o;
        }
        for (int q = 0; q < nj; q += 2) {
            const int o = jo[q], level = t - t0[o];
            const bool pl = probe && o == 0;
            hipEvent_t e0 = pl ? ctx->ev_blur[2 * (level - 1)] : nullptr, e1 = pl ? ctx->ev_blur[2 * (level - 1) + 1] : nullptr;
            // only when both fit into one round of resident workgroups (4 per CU): behind a launch that fills the
            // chip the second plane would just queue, and it would run on the larger radius' kernel for nothing
            bool pair = q + 1 < nj && ctx->batch_octaves;
            if (pair) {
                const int o2 = jo[q + 1], level2 = t - t0[o2];
                const int span = ctx->inc_span[level] > ctx->inc_span[level2] ? ctx->inc_span[level] : ctx->inc_span[level2];
                pair = psx_blur_pair_ok(P.oct[o].w, P.oct[o].h, P.oct[o2].w, P.oct[o2].h, span, ctx->resident_blocks);
            }
            if (pair) {
                const int o2 = jo[q + 1], level2 = t - t0[o2];
                PSX_HIP(psx_launch_blur2(blur_job(ctx, o, level), blur_job(ctx, o2, level2), ctx->stream, e0, e1));
                if (pl) probe_bytes += 8.0 * ((double)P.oct[o].w * P.oct[o].h + (double)P.oct[o2].w * P.oct[o2].h);
            } else {
                int rc = launch_blur_level(ctx, o, level, e0, e1);
                if (rc != PSX_OK) return rc;
                if (pl) probe_bytes += 8.0 * (double)P.oct[o].w * P.oct[o].h;
                if (q + 1 < nj) {
                    rc = launch_blur_level(ctx, jo[q + 1], t - t0[jo[q + 1]]);
                    if (rc != PSX_OK) return rc;
                }
            }
        }
        // a large octave's scan right behind its last level: its planes are as cache-resident now as they will ever
        // be.  The small octaves' scans (a few hundred tiles each, latency-bound launches) wait and share one launch.
        for (int q = 0; q < nj; q++)
            if (t - t0[jo[q]] == P.L - 1 && ctx->interleave) {
                if (psx_extrema_tiles(P, jo[q]) >= ctx->resident_blocks) {
                    const bool px = probe && jo[q] == 0;
                    if (px) PSX_HIP(hipEventRecord(ctx->ev_x[2], ctx->stream));
                    PSX_HIP(psx_launch_extrema(ctx->d_params, ctx->hp, ctx->d_cnt, jo[q], ctx->stream));
                    if (px) { PSX_HIP(hipEventRecord(ctx->ev_x[3], ctx->stream)); ctx->probe_ext0 = true; }
                } else
                    deferred[ndef++] = jo[q];
            }
    }
    if (ndef > 0) { int rc = launch_extrema_set(ctx, deferred, ndef); if (rc != PSX_OK) return rc; }
    if (probe) { ctx->blur_probe_n = P.L - 1; ctx->blur_probe_bytes = probe_bytes / (P.L - 1); }
    ctx->ext_launched = ctx->interleave;
    if (ctx->timers) PSX_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    return PSX_OK;
}

static int wait_stream(psx_ctx* ctx)
{
    if (ctx->blocking_wait) {
        // The C++ pipeline's workers wait here for most of a frame's life.  hipEventSynchronize on a hipEventBlockingSync
        // event still burnt the core on this stack (POPSIFT_PROFILE: 2.4 ms of thread CPU time per 0.35 ms frame, 8 workers
        // 86 % busy -- 7 cores per GPU, 54 on an 8-GPU host); asking the event and sleeping in between costs a few
        // microseconds of CPU per frame and, with several frames in flight per worker pool, no throughput.
        // POPSIFT_WAIT_SLEEP_US (default 40; 0 = the runtime's blocking wait).
        static const int sleep_us = [] { const char* e = getenv("POPSIFT_WAIT_SLEEP_US"); const int v = e ? atoi(e) : 40; return v < 0 ? 0 : v; }();
        if (!ctx->ev_wait) PSX_HIP(hipEventCreateWithFlags(&ctx->ev_wait, hipEventBlockingSync | hipEventDisableTiming));
        PSX_HIP(hipEventRecord(ctx->ev_wait, ctx->stream));
        if (sleep_us == 0) { PSX_HIP(hipEventSynchronize(ctx->ev_wait)); return PSX_OK; }
        for (;;) {
            const hipError_t q = hipEventQuery(ctx->ev_wait);
            if (q == hipSuccess) return PSX_OK;
            if (q != hipErrorNotReady) PSX_HIP(q);
            struct timespec ts = {0, (long)sleep_us * 1000L};
            nanosleep(&ts, nullptr);
        }
    }
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    return PSX_OK;
}

// Pyramid::extrema_filter_grid (s_filtergrid.cu:113-325), gated as in s_orientation.cu:378-383.
// Like the reference, the host reads the per-octave counts once to decide and to size the sort.
static int grid_filter(psx_ctx* ctx)
{
    PSX_HIP(hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, sizeof(PsxCounters), hipMemcpyDeviceToHost, ctx->stream));
    { int wrc = wait_stream(ctx); if (wrc != PSX_OK) return wrc; }
    int total = 0;
    for (int o = 0; o < ctx->hp.num_octaves; o++) total += imin(ctx->h_cnt->ext_ct[o], ctx->cfg.max_extrema);
    const int fmax = ctx->cfg.filter_max_extrema;
    if (!((int)(fmax * 1.1) < total)) return PSX_OK;

    int rc;
    size_t temp_bytes = 0;
    PSX_HIP(psx_gridfilter_sort_bytes(total, &temp_bytes));
    if ((rc = grow(ctx, &ctx->d_gf_keys, &ctx->gf_keys_cap, 2 * (size_t)total)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_gf_vals, &ctx->gf_vals_cap, 2 * (size_t)total)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_gf_temp, &ctx->gf_temp_cap, temp_bytes + 256)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_gf_scratch, &ctx->gf_scratch_cap,
                   psx_gridfilter_scratch_ints(ctx->cfg.filter_grid_size))) != PSX_OK) return rc;
    PSX_HIP(psx_launch_gridfilter(ctx->d_params, ctx->d_cnt, ctx->cfg.grid_filter_mode, total, fmax,
                                  ctx->d_gf_keys, ctx->d_gf_keys + total, ctx->d_gf_vals,
                                  ctx->d_gf_vals + total, ctx->d_gf_temp, temp_bytes, ctx->d_gf_scratch,
                                  ctx->stream));
    ctx->filtered = true;
    ctx->counts_valid = false;
    return PSX_OK;
}

int psx_find_extrema(psx_ctx* ctx)
{
    if (!ctx) return PSX_ERR_INVALID;
    if (!ctx->d_pyr) return fail(ctx, PSX_ERR_STATE, "psx_find_extrema: no pyramid");
    PSX_HIP(hipSetDevice(ctx->device));
    if (!ctx->ext_launched) {
        int small[PSX_MAX_OCTAVES], ns = 0;
        for (int o = 0; o < ctx->hp.num_octaves; o++) {
            if (psx_extrema_tiles(ctx->hp, o) >= ctx->resident_blocks) {
                const bool px = ctx->blur_probe && o == 0;
                if (px) PSX_HIP(hipEventRecord(ctx->ev_x[2], ctx->stream));
                PSX_HIP(psx_launch_extrema(ctx->d_params, ctx->hp, ctx->d_cnt, o, ctx->stream));
                if (px) { PSX_HIP(hipEventRecord(ctx->ev_x[3], ctx->stream)); ctx->probe_ext0 = true; }
            } else
                small[ns++] = o;
        }
        if (ns > 0) { int rc = launch_extrema_set(ctx, small, ns); if (rc != PSX_OK) return rc; }
    }
    ctx->ext_launched = false;
    PSX_HIP(psx_launch_refine(ctx->d_params, ctx->hp, ctx->d_cnt, ctx->stream));
    ctx->filtered = false;
    if (ctx->timers) PSX_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    return PSX_OK;
}

int psx_orientation(psx_ctx* ctx)
{
    if (!ctx) return PSX_ERR_INVALID;
    if (!ctx->d_pyr) return fail(ctx, PSX_ERR_STATE, "psx_orientation: no pyramid");
    PSX_HIP(hipSetDevice(ctx->device));
    if (ctx->cfg.filter_max_extrema > 0 && !ctx->filtered) {
        int rc = grid_filter(ctx);
        if (rc != PSX_OK) return rc;
    }
    PSX_HIP(psx_launch_orientation(ctx->d_params, ctx->d_cnt, ctx->stream));
    snapshot_export(ctx);
    PSX_HIP(psx_launch_scan(ctx->d_params, ctx->d_cnt, ctx->fx, ctx->stream));
    if (ctx->timers) PSX_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
    return PSX_OK;
}

int psx_descriptors(psx_ctx* ctx)
{
    if (!ctx) return PSX_ERR_INVALID;
    if (!ctx->d_pyr) return fail(ctx, PSX_ERR_STATE, "psx_descriptors: no pyramid");
    PSX_HIP(hipSetDevice(ctx->device));
    if (ctx->cfg.desc_mode == PSX_DESC_LOOP)
        PSX_HIP(psx_launch_descriptors(ctx->d_params, ctx->d_cnt, ctx->fx, ctx->resident_blocks / 4, ctx->stream));
    else
        PSX_HIP(psx_launch_descriptors_alt(ctx->d_params, ctx->d_cnt, ctx->cfg.desc_mode, ctx->fx, ctx->resident_blocks / 4, ctx->stream));
    if (ctx->timers) PSX_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
    return PSX_OK;
}

static int extract_chain(psx_ctx* ctx)
{
    int rc;
    ctx->interleave = !ctx->timers;        // per-stage timers need the stages back to back
    rc = psx_build_pyramid(ctx);
    ctx->interleave = false;
    if (rc != PSX_OK) return rc;
    if ((rc = psx_find_extrema(ctx)) != PSX_OK) return rc;
    if ((rc = psx_orientation(ctx)) != PSX_OK) return rc;
    return psx_descriptors(ctx);
}

static void drop_graph(psx_ctx* ctx)
{
    if (ctx->graph) { (void)hipGraphExecDestroy(ctx->graph); ctx->graph = nullptr; }
}

int psx_extract(psx_ctx* ctx)
{
    if (!ctx) return PSX_ERR_INVALID;
    if (!ctx->d_input || !ctx->d_pyr) return fail(ctx, PSX_ERR_STATE, "psx_extract: no input image");
    if (ctx->null_work != 0) {
        if (ctx->null_primed) {                    // measurement: no kernels; the first frame's results stand in
            if (ctx->null_work == 1) ctx->counts_valid = false;      // mode 1 reads the counters back per frame, like a real frame
            snapshot_export(ctx);
            return PSX_OK;
        }
        ctx->null_primed = true;
    }
    // Optionally replay the 36-launch chain as one hipGraph (POPSIFT_HIP_GRAPH=1).  Not with the grid filter
    // (it reads counters on the host in mid-chain) and not with the per-stage timers.
    const bool use_graph = !ctx->graph_off && !ctx->timers && !ctx->blur_probe && ctx->cfg.filter_max_extrema <= 0;
    if (!use_graph) return extract_chain(ctx);
    PSX_HIP(hipSetDevice(ctx->device));
    if (ctx->graph && (ctx->graph_input != ctx->d_input || ctx->graph_is_float != ctx->input_is_float ||
                       ctx->graph_w != ctx->in_w || ctx->graph_h != ctx->in_h))
        drop_graph(ctx);
    if (!ctx->graph) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            ctx->graph_off = true;
            return extract_chain(ctx);
        }
        const int rc = extract_chain(ctx);
        const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
        if (rc != PSX_OK || e != hipSuccess || g == nullptr ||
            hipGraphInstantiate(&ctx->graph, g, nullptr, nullptr, 0) != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            ctx->graph = nullptr;
            ctx->graph_off = true;
            (void)hipGetLastError();
            return rc != PSX_OK ? rc : extract_chain(ctx);
        }
        (void)hipGraphDestroy(g);
        ctx->graph_input = ctx->d_input; ctx->graph_is_float = ctx->input_is_float;
        ctx->graph_w = ctx->in_w; ctx->graph_h = ctx->in_h;
    }
    ctx->counts_valid = false;
    ctx->filtered = false;
    snapshot_export(ctx);             // the captured kernels carry the targets attached at capture time; attach drops the graph
    PSX_HIP(hipGraphLaunch(ctx->graph, ctx->stream));
    return PSX_OK;
}

int psx_sync(psx_ctx* ctx)
{
    if (!ctx) return PSX_ERR_INVALID;
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    return PSX_OK;
}

static int fetch_counts(psx_ctx* ctx);
static int fetch_counts_full(psx_ctx* ctx)
{
    if (ctx->counts_valid && !ctx->counts_partial) return PSX_OK;
    { int rc0 = fetch_counts(ctx); if (rc0 != PSX_OK) return rc0; }     // grows the descriptor buffers if needed
    if (!ctx->counts_partial) return PSX_OK;
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, sizeof(PsxCounters), hipMemcpyDeviceToHost, ctx->stream));
    { int wrc = wait_stream(ctx); if (wrc != PSX_OK) return wrc; }
    ctx->counts_valid = true;
    ctx->counts_partial = false;
    return PSX_OK;
}

// A frame produced more orientations than the descriptor buffers hold (they start at the reference's
// 2 x max_extrema entries): grow them and redo the two stages that depend on the capacity -- the
// orientation scan and the descriptors; the oriented extrema are still in place.  Pyramid::reallocExtrema,
// sift_pyramid.cu:186-209.
static int regrow_descriptors(psx_ctx* ctx, int ori_raw)
{
    PsxParams& P = ctx->hp;
    const size_t need = (size_t)ori_raw + (size_t)ori_raw / 4 + 1024;
    int rc;
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    if ((rc = grow(ctx, &ctx->d_desc, &ctx->desc_cap, need * 128)) != PSX_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_feat_to_ext, &ctx->f2e_cap, need)) != PSX_OK) return rc;
    P.desc = ctx->d_desc;
    P.feat_to_ext = ctx->d_feat_to_ext;
    P.ori_capacity = (int)need;
    *ctx->h_params_pin = P;
    PSX_HIP(hipMemcpyAsync(ctx->d_params, ctx->h_params_pin, sizeof(P), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->graph) { (void)hipGraphExecDestroy(ctx->graph); ctx->graph = nullptr; }
    // same targets as the launch being repeated
    PSX_HIP(psx_launch_scan(ctx->d_params, ctx->d_cnt, ctx->fx, ctx->stream));
    if (ctx->cfg.desc_mode == PSX_DESC_LOOP)
        PSX_HIP(psx_launch_descriptors(ctx->d_params, ctx->d_cnt, ctx->fx, ctx->resident_blocks / 4, ctx->stream));
    else
        PSX_HIP(psx_launch_descriptors_alt(ctx->d_params, ctx->d_cnt, ctx->cfg.desc_mode, ctx->fx, ctx->resident_blocks / 4, ctx->stream));
    return PSX_OK;
}

static int fetch_counts(psx_ctx* ctx)
{
    if (ctx->counts_valid) return PSX_OK;
    PSX_HIP(hipSetDevice(ctx->device));
    for (int attempt = 0; attempt < 2; attempt++) {
        int raw;
        if (ctx->fx_on) {
            // the frame was launched with export targets: its scan kernel deposited the counters in pinned memory
            { int wrc = wait_stream(ctx); if (wrc != PSX_OK) return wrc; }
            ctx->h_cnt->ext_total = ctx->h_xcnt[0];
            ctx->h_cnt->ori_total = ctx->h_xcnt[1];
            raw = ctx->h_xcnt[2];
            ctx->h_cnt->flow_error = ctx->h_xcnt[3];
            ctx->counts_partial = true;
        } else {
            PSX_HIP(hipMemcpyAsync(ctx->h_cnt, ctx->d_cnt, sizeof(PsxCounters), hipMemcpyDeviceToHost, ctx->stream));
            { int wrc = wait_stream(ctx); if (wrc != PSX_OK) return wrc; }
            raw = ctx->h_cnt->ori_raw;
            ctx->counts_partial = false;
        }
        // k_pyramid_flow gave up on a dependency wait (never in a correct run): the planes, hence the features, are invalid
        if (ctx->h_cnt->flow_error != 0)
            return fail(ctx, PSX_ERR_STATE, "the pyramid kernel ran into the bound of a device-side dependency wait: the frame is invalid");
        if (raw <= ctx->hp.ori_capacity || attempt == 1) break;
        int rc = regrow_descriptors(ctx, raw);
        if (rc != PSX_OK) return rc;
    }
    ctx->counts_valid = true;
    return PSX_OK;
}

int psx_counts(psx_ctx* ctx, int* num_features, int* num_descriptors)
{
    if (!ctx) return PSX_ERR_INVALID;
    int rc = fetch_counts(ctx);
    if (rc != PSX_OK) return rc;
    if (num_features) *num_features = ctx->h_cnt->ext_total;
    if (num_descriptors) *num_descriptors = ctx->h_cnt->ori_total;
    return PSX_OK;
}

int psx_download(psx_ctx* ctx, psx_feature* features, int feature_capacity, float* descriptors,
                 int descriptor_capacity)
{
    if (!ctx) return PSX_ERR_INVALID;
    int rc = fetch_counts(ctx);
    if (rc != PSX_OK) return rc;
    const int ne = ctx->h_cnt->ext_total, no = ctx->h_cnt->ori_total;
    if (ne > feature_capacity || no > descriptor_capacity)
        return fail(ctx, PSX_ERR_INVALID, "psx_download: output capacity too small");
    if (ne > 0 && !features) return fail(ctx, PSX_ERR_INVALID, "psx_download: null feature buffer");
    if (no > 0 && !descriptors) return fail(ctx, PSX_ERR_INVALID, "psx_download: null descriptor buffer");
    const bool feat_exported = (ctx->fx_on && features == ctx->fx_host_feat && ne <= ctx->fx.feat_capacity);
    const bool desc_exported = (ctx->fx_on && descriptors == ctx->fx_host_desc && no <= ctx->fx.desc_capacity);
    // PSX_NULL_DEVICE_WORK=2: one real download per CONTEXT keeps its host records well formed (a process-wide flag was a
    // data race between the workers and left every context but the first with uninitialised host buffers)
    if (ctx->null_work == 2 && ctx->null_primed && ctx->null_dl_done) return PSX_OK;
    if (ctx->null_work == 2 && ctx->null_primed) ctx->null_dl_done = true;
    if (ne > 0 && !feat_exported)
        PSX_HIP(hipMemcpyAsync(features, ctx->d_features, (size_t)ne * sizeof(psx_feature),
                               hipMemcpyDeviceToHost, ctx->stream));
    if (no > 0 && !desc_exported)
        PSX_HIP(hipMemcpyAsync(descriptors, ctx->d_desc, (size_t)no * 128 * sizeof(float),
                               hipMemcpyDeviceToHost, ctx->stream));
    // sleep on an event when the context is in blocking mode (the C++ pipeline's workers): hipStreamSynchronize spins, and
    // the ~0.3 ms of a frame's result DMA -- several ms when replicas share a GPU -- were a busy core per worker
    return wait_stream(ctx);
}

static int map_host(psx_ctx* ctx, void* host, size_t bytes, void** dev, bool* registered)
{
    *registered = false;
    *dev = nullptr;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, host);
    if (e != hipSuccess || attr.type == hipMemoryTypeUnregistered) {
        (void)hipGetLastError();
        PSX_HIP(hipHostRegister(host, bytes, hipHostRegisterMapped));
        *registered = true;
    }
    PSX_HIP(hipHostGetDevicePointer(dev, host, 0));
    return PSX_OK;
}

int psx_attach_export(psx_ctx* ctx, psx_feature* host_features, int feature_capacity,
                      float* host_descriptors, int descriptor_capacity)
{
    if (!ctx) return PSX_ERR_INVALID;
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->x_registered_feat) { (void)hipHostUnregister(ctx->x_host_feat); ctx->x_registered_feat = false; }
    if (ctx->x_registered_desc) { (void)hipHostUnregister(ctx->x_host_desc); ctx->x_registered_desc = false; }
    ctx->x_host_feat = nullptr; ctx->x_host_desc = nullptr;
    ctx->x_dev_feat = nullptr; ctx->x_dev_desc = nullptr;
    ctx->x_feat_cap = ctx->x_desc_cap = 0;
    if (host_features && feature_capacity > 0) {
        void* d = nullptr;
        int rc = map_host(ctx, host_features, (size_t)feature_capacity * sizeof(psx_feature), &d, &ctx->x_registered_feat);
        if (rc != PSX_OK) return rc;
        ctx->x_host_feat = host_features; ctx->x_dev_feat = static_cast<psx_feature*>(d);
        ctx->x_feat_cap = feature_capacity;
    }
    if (host_descriptors && descriptor_capacity > 0) {
        void* d = nullptr;
        int rc = map_host(ctx, host_descriptors, (size_t)descriptor_capacity * 128 * sizeof(float), &d, &ctx->x_registered_desc);
        if (rc != PSX_OK) return rc;
        ctx->x_host_desc = host_descriptors; ctx->x_dev_desc = static_cast<float*>(d);
        ctx->x_desc_cap = descriptor_capacity;
    }
    if (ctx->graph) { (void)hipGraphExecDestroy(ctx->graph); ctx->graph = nullptr; }      // the targets are kernel arguments
    return PSX_OK;
}

int psx_attach_export_mapped(psx_ctx* ctx, psx_feature* host_features, int feature_capacity,
                             float* host_descriptors, int descriptor_capacity)
{
    if (!ctx) return PSX_ERR_INVALID;
    // Normally no HIP call: the targets travel as kernel arguments of the NEXT extraction; the frame in flight (if
    // any) keeps the targets it was launched with, and its counters / results are still fetched from those
    // (psx_ctx::fx).  Only buffers an earlier psx_attach_export had to register are a reason to wait: the frame
    // in flight may still be storing into them.
    if (ctx->x_registered_feat || ctx->x_registered_desc) {
        PSX_HIP(hipSetDevice(ctx->device));
        PSX_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (ctx->x_registered_feat) { (void)hipHostUnregister(ctx->x_host_feat); ctx->x_registered_feat = false; }
    if (ctx->x_registered_desc) { (void)hipHostUnregister(ctx->x_host_desc); ctx->x_registered_desc = false; }
    // psx_host_alloc memory: mapped, and its device address is its host address (checked at allocation)
    ctx->x_host_feat = ctx->x_dev_feat = (host_features && feature_capacity > 0) ? host_features : nullptr;
    ctx->x_host_desc = ctx->x_dev_desc = (host_descriptors && descriptor_capacity > 0) ? host_descriptors : nullptr;
    ctx->x_feat_cap = ctx->x_dev_feat ? feature_capacity : 0;
    ctx->x_desc_cap = ctx->x_dev_desc ? descriptor_capacity : 0;
    if (ctx->graph) { (void)hipGraphExecDestroy(ctx->graph); ctx->graph = nullptr; }
    return PSX_OK;
}

int psx_device_results(psx_ctx* ctx, const psx_feature** d_features, const float** d_descriptors,
                       const int** d_feat_to_ext)
{
    if (!ctx) return PSX_ERR_INVALID;
    if (d_features) *d_features = ctx->d_features;
    if (d_descriptors) *d_descriptors = ctx->d_desc;
    if (d_feat_to_ext) *d_feat_to_ext = ctx->d_feat_to_ext;
    return PSX_OK;
}

int psx_host_alloc(size_t bytes, void** out)
{
    psx_ctx* ctx = nullptr;
    if (!out) return PSX_ERR_INVALID;
    PSX_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocPortable));
    // psx_upload_pinned / psx_attach_export_mapped rely on "device address == host address" for this memory
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, *out, 0) != hipSuccess || dev != *out) {
        (void)hipHostFree(*out); *out = nullptr;
        return fail(nullptr, PSX_ERR_HIP, "psx_host_alloc: mapped host memory has a different device address on this system");
    }
    return PSX_OK;
}

int psx_host_alloc_near(int device, size_t bytes, void** out)
{
    psx_ctx* ctx = nullptr;
    if (device >= 0) PSX_HIP(hipSetDevice(device));
    return psx_host_alloc(bytes, out);
}

int psx_host_free(void* ptr)
{
    psx_ctx* ctx = nullptr;
    if (!ptr) return PSX_OK;
    PSX_HIP(hipHostFree(ptr));
    return PSX_OK;
}

int psx_dev_alloc(int device, size_t bytes, void** out)
{
    psx_ctx* ctx = nullptr;
    if (!out) return PSX_ERR_INVALID;
    PSX_HIP(hipSetDevice(device));
    PSX_HIP(hipMalloc(out, bytes ? bytes : 1));
    return PSX_OK;
}

int psx_dev_free(int device, void* ptr)
{
    psx_ctx* ctx = nullptr;
    if (!ptr) return PSX_OK;
    PSX_HIP(hipSetDevice(device));
    PSX_HIP(hipFree(ptr));
    return PSX_OK;
}

int psx_dev_read(int device, void* host_dst, const void* dev_src, size_t bytes)
{
    psx_ctx* ctx = nullptr;
    if (bytes == 0) return PSX_OK;
    if (!host_dst || !dev_src) return PSX_ERR_INVALID;
    PSX_HIP(hipSetDevice(device));
    PSX_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return PSX_OK;
}

int psx_dev_write(int device, void* dev_dst, const void* host_src, size_t bytes)
{
    psx_ctx* ctx = nullptr;
    if (bytes == 0) return PSX_OK;
    if (!dev_dst || !host_src) return PSX_ERR_INVALID;
    PSX_HIP(hipSetDevice(device));
    PSX_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return PSX_OK;
}

int psx_clone_results(psx_ctx* ctx, void* d_features, void* d_descriptors, int* d_reverse_map)
{
    if (!ctx) return PSX_ERR_INVALID;
    int rc = fetch_counts(ctx);
    if (rc != PSX_OK) return rc;
    const int ne = ctx->h_cnt->ext_total, no = ctx->h_cnt->ori_total;
    if (ne > 0 && d_features)
        PSX_HIP(psx_launch_feature_ptrs(ctx->d_features, static_cast<psx_feature_dev*>(d_features), ne,
                                        static_cast<float*>(d_descriptors), d_descriptors ? no : 0, ctx->stream));
    if (no > 0 && d_descriptors)
        PSX_HIP(hipMemcpyAsync(d_descriptors, ctx->d_desc, (size_t)no * 128 * sizeof(float),
                               hipMemcpyDeviceToDevice, ctx->stream));
    if (no > 0 && d_reverse_map)
        PSX_HIP(hipMemcpyAsync(d_reverse_map, ctx->d_feat_to_ext, (size_t)no * sizeof(int),
                               hipMemcpyDeviceToDevice, ctx->stream));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    return PSX_OK;
}

int psx_device_count(int* count)
{
    psx_ctx* ctx = nullptr;
    if (!count) return PSX_ERR_INVALID;
    *count = 0;
    PSX_HIP(hipGetDeviceCount(count));
    return PSX_OK;
}

int psx_device_info(int device, char* name, int name_len, size_t* total_mem, int* compute_units, int* clock_khz)
{
    psx_ctx* ctx = nullptr;
    hipDeviceProp_t p;
    PSX_HIP(hipGetDeviceProperties(&p, device));
    if (name && name_len > 0) { strncpy(name, p.name, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (total_mem) *total_mem = p.totalGlobalMem;
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    return PSX_OK;
}

int psx_device_pci(int device, char* bus_id, int len)
{
    psx_ctx* ctx = nullptr;
    if (!bus_id || len < 16) return PSX_ERR_INVALID;
    bus_id[0] = 0;
    PSX_HIP(hipDeviceGetPCIBusId(bus_id, len, device));
    return PSX_OK;
}

int psx_dump_plane(psx_ctx* ctx, int kind, int octave, int level, float* host_out)
{
    if (!ctx || !host_out) return PSX_ERR_INVALID;
    const PsxParams& P = ctx->hp;
    if (!ctx->d_pyr || octave < 0 || octave >= P.num_octaves) return fail(ctx, PSX_ERR_INVALID, "bad octave");
    const int nl = (kind == PSX_PLANE_DOG) ? P.L - 1 : P.L;
    if (level < 0 || level >= nl) return fail(ctx, PSX_ERR_INVALID, "bad level");
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    const PsxOctave& oc = P.oct[octave];
    const float* src = oc.data + (size_t)level * oc.plane;
    if (kind == PSX_PLANE_GAUSS) {
        PSX_HIP(hipMemcpy2D(host_out, (size_t)oc.w * 4, src, (size_t)oc.pitch * 4, (size_t)oc.w * 4, oc.h,
                            hipMemcpyDeviceToHost));
    } else {
        // make_dog (s_pyramid_build.cu:74-92) into a scratch plane
        float* tmp = nullptr;
        PSX_HIP(hipMalloc(reinterpret_cast<void**>(&tmp), oc.plane * sizeof(float)));
        hipError_t e = psx_launch_dog(src, src + oc.plane, tmp, oc.w, oc.h, oc.pitch, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpy2D(host_out, (size_t)oc.w * 4, tmp, (size_t)oc.pitch * 4, (size_t)oc.w * 4, oc.h,
                            hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        PSX_HIP(e);
    }
    return PSX_OK;
}

int psx_dump_iext(psx_ctx* ctx, int octave, psx_iext* host_out, int capacity, int* count)
{
    if (!ctx || octave < 0 || octave >= ctx->hp.num_octaves) return PSX_ERR_INVALID;
    if (ctx->counts_partial) { ctx->counts_valid = false; }
    int rc = fetch_counts_full(ctx);
    if (rc != PSX_OK) return rc;
    int n = imin(ctx->filtered ? ctx->h_cnt->iext_ct[octave] : ctx->h_cnt->ext_ct[octave], ctx->cfg.max_extrema);
    if (count) *count = n;
    if (host_out) {
        n = imin(n, capacity);
        if (n > 0) PSX_HIP(hipMemcpy(host_out, ctx->hp.iext[octave], (size_t)n * sizeof(psx_iext), hipMemcpyDeviceToHost));
    }
    return PSX_OK;
}

int psx_dump_extrema(psx_ctx* ctx, psx_extremum* host_out, int capacity, int* count)
{
    if (!ctx) return PSX_ERR_INVALID;
    int rc = fetch_counts(ctx);
    if (rc != PSX_OK) return rc;
    int n = ctx->h_cnt->ext_total;
    if (count) *count = n;
    if (host_out) {
        n = imin(n, capacity);
        if (n > 0) PSX_HIP(hipMemcpy(host_out, ctx->d_extrema, (size_t)n * sizeof(psx_extremum), hipMemcpyDeviceToHost));
    }
    return PSX_OK;
}

int psx_set_wait_mode(psx_ctx* ctx, int blocking)
{
    if (!ctx) return PSX_ERR_INVALID;
    ctx->blocking_wait = blocking != 0;
    return PSX_OK;
}

int psx_enable_timers(psx_ctx* ctx, int on)
{
    if (!ctx) return PSX_ERR_INVALID;
    ctx->timers = on != 0;
    return PSX_OK;
}

int psx_stage_times(psx_ctx* ctx, float ms[4])
{
    if (!ctx || !ms) return PSX_ERR_INVALID;
    if (!ctx->timers) return fail(ctx, PSX_ERR_STATE, "timers are not enabled");
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; i++) PSX_HIP(hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]));
    return PSX_OK;
}

int psx_time_blur(psx_ctx* ctx, int octave, int level, int reps, float* avg_ms, double* bytes)
{
    if (!ctx || !avg_ms) return PSX_ERR_INVALID;
    const PsxParams& P = ctx->hp;
    if (!ctx->d_pyr || octave < 0 || octave >= P.num_octaves || level < 1 || level >= P.L || reps < 1)
        return fail(ctx, PSX_ERR_INVALID, "psx_time_blur: bad arguments");
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipEventRecord(ctx->ev_t0, ctx->stream));
    for (int r = 0; r < reps; r++) {
        int rc = launch_blur_level(ctx, octave, level);
        if (rc != PSX_OK) return rc;
    }
    PSX_HIP(hipEventRecord(ctx->ev_t1, ctx->stream));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.0f;
    PSX_HIP(hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1));
    *avg_ms = ms / reps;
    if (bytes) *bytes = 8.0 * (double)P.oct[octave].w * (double)P.oct[octave].h;
    return PSX_OK;
}

#ifdef PSX_PHASE_TIMING
int psx_debug_launch_extrema(psx_ctx* ctx, int o)
{
    PSX_HIP(psx_launch_extrema(ctx->d_params, ctx->hp, ctx->d_cnt, o, ctx->stream));
    return PSX_OK;
}
#endif


int psx_enable_blur_probe(psx_ctx* ctx, int on)
{
    if (!ctx) return PSX_ERR_INVALID;
    PSX_HIP(hipSetDevice(ctx->device));
    if (on)
        for (int i = 0; i < 2 * PSX_GAUSS_LEVELS; i++)
            if (!ctx->ev_blur[i]) PSX_HIP(hipEventCreate(&ctx->ev_blur[i]));
    if (on)
        for (int i = 0; i < 4; i++)
            if (!ctx->ev_x[i]) PSX_HIP(hipEventCreate(&ctx->ev_x[i]));
    ctx->blur_probe = on != 0;
    ctx->probe_ext0 = false;
    ctx->blur_probe_n = 0;
    return PSX_OK;
}

int psx_blur_probe_times(psx_ctx* ctx, float* ms, int capacity, int* n, double* bytes_per_launch)
{
    if (!ctx || !ms || !n) return PSX_ERR_INVALID;
    if (!ctx->blur_probe) return fail(ctx, PSX_ERR_STATE, "the blur probe is not enabled");
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    *n = ctx->blur_probe_n;
    for (int i = 0; i < ctx->blur_probe_n && i < capacity; i++)
        PSX_HIP(hipEventElapsedTime(&ms[i], ctx->ev_blur[2 * i], ctx->ev_blur[2 * i + 1]));
    if (bytes_per_launch) *bytes_per_launch = ctx->blur_probe_bytes;
    return PSX_OK;
}

// Measurement: one pyramid build with k_pyramid_flow recording, per work item, the 100 MHz wall clock at dequeue /
// dependencies met / arithmetic done / published, a word (octave << 40 | level << 32 | chunk << 16 | strip) and the
// workgroup: 6 int64 per item, in ticket order.  *nitems = items of the plan (0: the flow kernel is not in use).
int psx_flow_trace(psx_ctx* ctx, long long* host_out, int capacity_items, int* nitems)
{
    if (!ctx || !nitems) return PSX_ERR_INVALID;
    *nitems = ctx->flow_on ? ctx->flow_nitems : 0;
    if (!ctx->flow_on || host_out == nullptr || capacity_items < ctx->flow_nitems) return PSX_OK;
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    const size_t bytes = sizeof(long long) * 6 * (size_t)ctx->flow_nitems;
    PSX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_flow_trace), bytes));
    (void)hipMemset(ctx->d_flow_trace, 0, bytes);
    int rc = psx_build_pyramid(ctx);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(host_out, ctx->d_flow_trace, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(ctx->d_flow_trace);
    ctx->d_flow_trace = nullptr;
    if (rc != PSX_OK) return rc;
    PSX_HIP(e);
    return PSX_OK;
}

// The probe's two other HBM-bound kernels of octave 0, timed in the pipeline with stream events around the launch (they
// include the gap in front of the kernel): level 0 (k_level0_fused, or k_upscale + k_blur<R, true>) and the extrema scan.
// bytes: the algorithmic figures of SURVEY.md 8d (4 B per octave-0 pixel + the input; 24 B per octave-0 pixel).
// extrema_ms = 0 when octave 0's scan shared a launch with other octaves (small images).
int psx_probe_extra_times(psx_ctx* ctx, float* level0_ms, double* level0_bytes, float* extrema_ms, double* extrema_bytes)
{
    if (!ctx || !level0_ms || !extrema_ms) return PSX_ERR_INVALID;
    if (!ctx->blur_probe) return fail(ctx, PSX_ERR_STATE, "the blur probe is not enabled");
    PSX_HIP(hipSetDevice(ctx->device));
    PSX_HIP(hipStreamSynchronize(ctx->stream));
    const PsxParams& P = ctx->hp;
    *level0_ms = 0.0f; *extrema_ms = 0.0f;
    if (!ctx->alt_pyramid) PSX_HIP(hipEventElapsedTime(level0_ms, ctx->ev_x[0], ctx->ev_x[1]));
    if (ctx->probe_ext0) PSX_HIP(hipEventElapsedTime(extrema_ms, ctx->ev_x[2], ctx->ev_x[3]));
    const double n0 = (double)P.oct[0].w * P.oct[0].h;
    if (level0_bytes) *level0_bytes = 4.0 * n0 + (double)ctx->in_w * ctx->in_h * (ctx->input_is_float ? 4.0 : 1.0);
    if (extrema_bytes) *extrema_bytes = 4.0 * (double)P.L * n0;
    return PSX_OK;
}

void* psx_stream(psx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

} // extern "C"
