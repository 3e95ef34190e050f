/*
 * include/popsift_hip.h -- C-ABI of the MI355X-native SIFT extraction hot path.
 *
 * This is the drop-in boundary: the C++14 host library (popsift_amd/csrc/host, classes
 * PopSift / SiftJob / popsift::Config / popsift::FeaturesHost with the reference's
 * signatures) sits above it, the hand-written HIP kernels for gfx950 sit below it.
 * Plain pointers and sizes only; no HIP, torch or C++ types; every function returns an
 * int status (PSX_OK == 0) and never throws.  psx_last_error() gives the message the C++
 * layer turns into std::runtime_error, the reference's error convention
 * (common/debug_macros.h:122-127 POP_FATAL).
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/src/popsift).
 */
#ifndef POPSIFT_HIP_H
#define POPSIFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSX_OK                 0
#define PSX_ERR_INVALID       -1   /* bad argument / unsupported mode          */
#define PSX_ERR_HIP           -2   /* a HIP runtime call failed                */
#define PSX_ERR_NOMEM         -3   /* host or device allocation failed         */
#define PSX_ERR_STATE         -4   /* call sequence error (e.g. no image set)  */

#define PSX_MAX_OCTAVES       20   /* sift_conf.h:12  MAX_OCTAVES              */
#define PSX_GAUSS_ALIGN       32   /* sift_constants.h:37 GAUSS_ALIGN          */
#define PSX_GAUSS_LEVELS      12   /* sift_constants.h:38 GAUSS_LEVELS         */
#define PSX_ORI_MAX            4   /* sift_constants.h:55 ORIENTATION_MAX_COUNT*/

/* popsift::Config::GaussMode, sift_conf.h:38-46 */
enum { PSX_GAUSS_VLFEAT_COMPUTE = 0, PSX_GAUSS_VLFEAT_RELATIVE = 1, PSX_GAUSS_VLFEAT_RELATIVE_ALL = 2,
       PSX_GAUSS_OPENCV_COMPUTE = 3, PSX_GAUSS_FIXED9 = 4, PSX_GAUSS_FIXED15 = 5 };
/* popsift::Config::SiftMode, sift_conf.h:51-61 */
enum { PSX_MODE_POPSIFT = 0, PSX_MODE_OPENCV = 1, PSX_MODE_VLFEAT = 2 };
/* popsift::Config::ScalingMode, sift_conf.h:75-80 */
enum { PSX_SCALE_DIRECT = 0, PSX_SCALE_DEFAULT = 1 };
/* popsift::Config::DescMode, sift_conf.h:85-97 */
enum { PSX_DESC_LOOP = 0, PSX_DESC_ILOOP = 1, PSX_DESC_GRID = 2, PSX_DESC_IGRID = 3, PSX_DESC_NOTILE = 4 };
/* popsift::Config::NormMode, sift_conf.h:102-108 */
enum { PSX_NORM_ROOTSIFT = 0, PSX_NORM_CLASSIC = 1 };
/* popsift::Config::GridFilterMode, sift_conf.h:118-125 */
enum { PSX_FILTER_RANDOM = 0, PSX_FILTER_LARGEST_FIRST = 1, PSX_FILTER_SMALLEST_FIRST = 2 };

/* POD image of popsift::Config (sift_conf.h:29-409; defaults sift_conf.cu:18-41). */
typedef struct psx_config {
    int   octaves;             /* -1: auto = max(floor(log2(min(w,h))) - 3 + 2^up, 1), popsift.cpp:118-122 */
    int   levels;              /* 3 */
    float sigma;               /* 1.6 */
    float edge_limit;          /* 10 */
    float threshold;           /* 0.04 */
    float upscale_factor;      /* 1.0 (Config::setDownsampling stores -v) */
    int   gauss_mode;          /* PSX_GAUSS_VLFEAT_COMPUTE */
    int   sift_mode;           /* PSX_MODE_POPSIFT */
    int   scaling_mode;        /* PSX_SCALE_DEFAULT */
    int   desc_mode;           /* PSX_DESC_LOOP */
    int   norm_mode;           /* PSX_NORM_ROOTSIFT */
    int   norm_multi;          /* 0 */
    int   max_extrema;         /* 100000 per octave */
    int   assume_initial_blur; /* 1 */
    float initial_blur;        /* 0.5 */
    int   filter_max_extrema;  /* -1 (grid filter off) */
    int   filter_grid_size;    /* 2 */
    int   grid_filter_mode;    /* PSX_FILTER_RANDOM */
} psx_config;

/* One keypoint as handed back to the host: popsift::Feature (features.h:23-37) with the
 * four Descriptor* replaced by indices into the descriptor array (-1 == nullptr); the C++
 * layer turns them into pointers into its own FeaturesHost storage, which removes the
 * reference's device-side host-pointer arithmetic (sift_pyramid.cu:242-280). */
typedef struct psx_feature {
    int   debug_octave;
    float xpos;
    float ypos;
    float sigma;
    int   num_ori;
    float orientation[PSX_ORI_MAX];
    int   desc_idx[PSX_ORI_MAX];
} psx_feature;

/* popsift::Feature itself (features.h:23-37) as laid out by the x86-64 and gfx950 ABIs (72 bytes), with
 * DEVICE descriptor pointers: what FeaturesDev::getFeatures() points at (features.h:104-122). */
typedef struct psx_feature_dev {
    int    debug_octave;
    float  xpos;
    float  ypos;
    float  sigma;
    int    num_ori;
    float  orientation[PSX_ORI_MAX];
    int    pad;
    float* desc[PSX_ORI_MAX];
} psx_feature_dev;

/* popsift::InitialExtremum (sift_extremum.h:25-39) as stored by the extrema kernel. */
typedef struct psx_iext {
    float xpos;
    float ypos;
    int   lpos;
    float sigma;
    int   cell;
    int   ignore;
} psx_iext;

/* popsift::Extremum (sift_extremum.h:47-63). */
typedef struct psx_extremum {
    float xpos;
    float ypos;
    int   lpos;
    float sigma;
    int   octave;
    int   num_ori;
    int   idx_ori;
    float orientation[PSX_ORI_MAX];
} psx_extremum;

typedef struct psx_ctx psx_ctx;   /* one Pyramid + its buffers + one HIP stream */

/* ---- configuration -------------------------------------------------------------------- */

/* Config::Config() defaults, sift_conf.cu:18-41 (without its cudaGetDevice side effect). */
int psx_config_default(psx_config* cfg);

/* Config::getPeakThreshold(), sift_conf.cu:276-279. */
float psx_peak_threshold(const psx_config* cfg);

/* init_filter() host part (gauss_filter.cu:127-237): fills the "inc" and "dd" tables.
 * inc_filter: PSX_GAUSS_LEVELS*PSX_GAUSS_ALIGN floats, dd_filter: PSX_MAX_OCTAVES*PSX_GAUSS_ALIGN.
 * Error cases of gauss_filter.cu:131-144 (sigma > 2, too many levels) return PSX_ERR_INVALID. */
int psx_gauss_tables(const psx_config* cfg, float* inc_filter, int* inc_span, float* inc_sigma,
                     float* dd_filter, int* dd_span, float* dd_sigma);
/* Config::setPrintGaussTables / --print-gauss-tables: prints to stdout what the reference's init_filter prints
 * (gauss_filter.cu:146-161) and what print_gauss_filter_symbol<<<1,1>>>(columns) prints from the device copy of the
 * tables (gauss_filter.cu:24-120, 247-256; the reference passes columns = 10): the inc, interpolated inc, abs_o0,
 * abs_oN and dd tables, same layout and number formats, computed on the host. */
int psx_print_gauss_tables(const psx_config* cfg, int columns);


/* ---- context -------------------------------------------------------------------------- */

/* Replaces PopSift::PopSift + applyConfiguration (popsift.cpp:25-48, 91-107): selects the
 * device, uploads Gauss tables / constants into the context (no global symbols), creates the
 * stream.  Modes outside the default branch of build_pyramid (s_pyramid_build.cu:547-575) and
 * DescMode != Loop return PSX_ERR_INVALID ("not yet", sift_desc.cu:80-82). */
int psx_create(int device, const psx_config* cfg, psx_ctx** out);
int psx_destroy(psx_ctx* ctx);
const char* psx_last_error(const psx_ctx* ctx);   /* ctx may be NULL: last create() error */

/* Replaces PopSift::private_init / Pyramid::Pyramid / Pyramid::resetDimensions
 * (popsift.cpp:128-144, sift_pyramid.cu:108-177): (re)allocates pyramid planes for an input of
 * w x h pixels.  Buffers only ever grow. */
int psx_resize(psx_ctx* ctx, int w, int h);

/* Octave geometry after psx_resize: popsift.cpp:124-125 and sift_pyramid.cu:129-134. */
int psx_num_octaves(const psx_ctx* ctx);
int psx_num_levels(const psx_ctx* ctx);           /* levels + 3 */
int psx_octave_dims(const psx_ctx* ctx, int octave, int* w, int* h);

/* ---- input ---------------------------------------------------------------------------- */

/* Replaces Image::load / ImageFloat::load (s_image.cu:69-77, 193-201): host -> device copy of
 * a tightly packed w*h plane, stream ordered.  Calls psx_resize when dimensions change. */
int psx_upload_u8(psx_ctx* ctx, const uint8_t* host, int w, int h);
int psx_upload_f32(psx_ctx* ctx, const float* host, int w, int h);

/* The same for an image that already lies in memory from psx_host_alloc (pinned): no pointer query, no staging
 * copy; the DMA reads the buffer directly, so it must stay untouched until the extraction has been waited for. */
int psx_upload_pinned(psx_ctx* ctx, const void* pinned_host, int w, int h, int is_float);

/* Input already resident in HBM (tight w*h plane).  The pointer must stay valid until the
 * extraction that uses it has finished.  No copy is made. */
int psx_set_input_dev(psx_ctx* ctx, const void* dev_ptr, int w, int h, int is_float);

/* ---- stages (all asynchronous on the context's stream) ---------------------------------- */

/* Pyramid::step1 -> build_pyramid (s_pyramid_build.cu:459-594, default branch). */
int psx_build_pyramid(psx_ctx* ctx);
/* Pyramid::find_extrema (s_extrema.cu:560-640). */
int psx_find_extrema(psx_ctx* ctx);
/* Pyramid::orientation incl. ori_prefix_sum (s_orientation.cu:364-441) and, when
 * filter_max_extrema > 0, extrema_filter_grid (s_filtergrid.cu:113-325) on the device.  As in the
 * reference the filter reads the extrema counts on the host first (one stream sync per frame); it
 * runs only if int(filter_max_extrema*1.1) < number of extrema (s_orientation.cu:378-383).
 * LargestScaleFirst / SmallestScaleFirst results are deterministic; RandomScale keeps the first
 * extrema of each cell in buffer (atomicAdd arrival) order, as the reference does. */
int psx_orientation(psx_ctx* ctx);
/* Pyramid::descriptors incl. normalize_histogram (sift_desc.cu:55-110) and prep_features
 * (sift_pyramid.cu:250-280). */
int psx_descriptors(psx_ctx* ctx);
/* step1 + step2 in one call (popsift.cpp:321-324). */
int psx_extract(psx_ctx* ctx);

/* How psx_counts / psx_sync-like waits of this context wait for the GPU: 0 (default) = hipStreamSynchronize
 * (lowest latency, the calling thread may spin), 1 = sleep on a blocking event (for hosts that run one
 * thread per context). */
int psx_set_wait_mode(psx_ctx* ctx, int blocking);

/* Waits for everything queued on the context's stream. */
int psx_sync(psx_ctx* ctx);

/* ---- results -------------------------------------------------------------------------- */

/* Pyramid::readDescCountersFromDevice (sift_pyramid.cu:373-381): synchronises, returns the
 * number of keypoints and descriptors of the last extraction. */
int psx_counts(psx_ctx* ctx, int* num_features, int* num_descriptors);

/* Pyramid::get_descriptors (sift_pyramid.cu:282-322): copies num_features psx_feature records
 * and num_descriptors*128 floats to host memory (capacities in elements); synchronises. */
int psx_download(psx_ctx* ctx, psx_feature* features, int feature_capacity,
                 float* descriptors, int descriptor_capacity);

/* Zero-copy export.  Replaces the per-image cudaHostRegister + 2x cudaMemcpyAsync + unregister of
 * Pyramid::get_descriptors / FeaturesHost::pin (sift_pyramid.cu:300-318, features.cu:86-111):
 * the caller attaches two host buffers (any host memory; they are registered once with the HIP
 * runtime, or pass memory that is already pinned) and the descriptor kernel streams every Feature
 * record and every normalised descriptor straight into them over PCIe while it computes, the scan
 * kernel deposits the two counters.  After psx_sync()/psx_counts() the results are in the buffers;
 * no copy is queued and the host never waits for a transfer.  Capacities are in elements;
 * results beyond a capacity are dropped from the export (the device copy stays complete).
 * Pass NULL/0 to detach. */
int psx_attach_export(psx_ctx* ctx, psx_feature* host_features, int feature_capacity,
                      float* host_descriptors, int descriptor_capacity);

/* psx_attach_export for buffers from psx_host_alloc (already pinned and GPU-mapped): no registration and no
 * pointer queries -- cheap enough to call once per frame when every result keeps its own descriptor buffer. */
int psx_attach_export_mapped(psx_ctx* ctx, psx_feature* host_features, int feature_capacity,
                             float* host_descriptors, int descriptor_capacity);

/* Device-resident results (FeaturesDev, features.h:104-122): pointers valid until the next
 * extraction on this context. */
int psx_device_results(psx_ctx* ctx, const psx_feature** d_features, const float** d_descriptors,
                       const int** d_feat_to_ext);

/* FeaturesDev support (MatchingMode, popsift.cpp:346-383, sift_pyramid.cu:324-362): device
 * buffers owned by the caller and a device-to-device clone of the last results: descriptors,
 * descriptor->extremum map, and the features as psx_feature_dev records (= popsift::Feature, 72 bytes)
 * whose desc[] point INTO d_descriptors, as Pyramid::clone_device_descriptors + prep_features leave them. */
/* pinned, GPU-mapped host memory (hipHostMalloc): cheap targets for psx_attach_export and sources for
 * psx_upload_*.  Allocation is slow (pool the buffers). */
int psx_host_alloc(size_t bytes, void** out);
/* the same with `device` made current on the calling thread first (device < 0: as psx_host_alloc): the HIP runtime
 * places pinned pages near the calling thread's current device or CPUs; a pool that serves device d allocates here. */
int psx_host_alloc_near(int device, size_t bytes, void** out);
int psx_host_free(void* ptr);
int psx_dev_alloc(int device, size_t bytes, void** out);
int psx_dev_free(int device, void* ptr);
/* synchronous device -> host copy of a buffer obtained from psx_dev_alloc / psx_clone_results */
int psx_dev_read(int device, void* host_dst, const void* dev_src, size_t bytes);
/* synchronous host -> device copy into such a buffer */
int psx_dev_write(int device, void* dev_dst, const void* host_src, size_t bytes);
int psx_clone_results(psx_ctx* ctx, void* d_features, void* d_descriptors, int* d_reverse_map);

/* FeaturesDev::match (features.cu:160-304, compute_distance / l2_in_t0): brute-force 2-nearest
 * neighbours of every left descriptor among the right ones, squared L2 distance evaluated with the
 * reference's operation tree, right side scanned in index order with strict '<' (ties keep the earlier
 * index).  d_left / d_right: DEVICE pointers to l_len / r_len descriptors of 128 floats (e.g.
 * psx_device_results / psx_clone_results).  host_match[3*i..3*i+2] = {best, second, accept} with
 * accept = (d_best / d_second < 0.8f), the int3 match_matrix of the reference; host_dist (may be NULL)
 * [2*i..2*i+1] = the two squared distances (what show_distance prints).  Synchronous. */
int psx_match(int device, const float* d_left, int l_len, const float* d_right, int r_len,
              int* host_match, float* host_dist);
/* Frees the calling thread's matcher scratch (a private stream and device buffers that psx_match keeps between
 * calls); optional -- the scratch is also released when the thread exits. */
int psx_match_release(void);

/* device_prop_t (common/device_prop.h:23-108): enumeration only; there are no texture limits. */
int psx_device_count(int* count);
int psx_device_info(int device, char* name, int name_len, size_t* total_mem, int* compute_units, int* clock_khz);
/* PCI bus id ("0000:c1:00.0") of a device: lets the host library place its worker threads on the CPUs
 * local to the GPU (sysfs local_cpulist) when it drives one PopSift per GPU (popsift.h:158,166-168). */
int psx_device_pci(int device, char* bus_id, int len);

/* ---- introspection for parity tests (Octave::download_and_save_array, sift_octave.cu:111-188) */

#define PSX_PLANE_GAUSS 0
#define PSX_PLANE_DOG   1
/* Copies one W*H float plane (tight) of the current pyramid to host memory; synchronises. */
int psx_dump_plane(psx_ctx* ctx, int kind, int octave, int level, float* host_out);
/* Initial extrema of one octave (i_ext_dat, sift_pyramid.h:44-48); returns the count. */
int psx_dump_iext(psx_ctx* ctx, int octave, psx_iext* host_out, int capacity, int* count);
/* Oriented extrema (dobuf.extrema) in octave-major order. */
int psx_dump_extrema(psx_ctx* ctx, psx_extremum* host_out, int capacity, int* count);

/* ---- measurement ---------------------------------------------------------------------- */

/* Stage timers: when enabled, HIP events bracket each stage group on the context's stream;
 * psx_stage_times synchronises and returns milliseconds of the last extraction
 * [0]=pyramid [1]=extrema [2]=orientation+scan [3]=descriptors+features. */
int psx_enable_timers(psx_ctx* ctx, int on);
int psx_stage_times(psx_ctx* ctx, float ms[4]);

/* Times `reps` launches of the separable-Gaussian kernel of (octave, level>=1) with HIP events
 * on the context's stream and returns the average duration in ms plus the algorithmic bytes
 * of one launch (8 bytes per pixel: plane read once, written once). */
int psx_time_blur(psx_ctx* ctx, int octave, int level, int reps, float* avg_ms, double* bytes);

/* With the blur probe enabled: in-pipeline durations (stream events around the launch) of octave 0's level 0 and of its
 * extrema scan in the last extraction, and their algorithmic bytes (4 B per octave-0 pixel + the input image;
 * 4 B x (levels + 3) planes per octave-0 pixel read by the scan).  extrema_ms = 0 when the scan shared a launch. */
int psx_probe_extra_times(psx_ctx* ctx, float* level0_ms, double* level0_bytes, float* extrema_ms, double* extrema_bytes);

/* Diagnostic (tests): `rounds` hand-overs of one plain device-memory word from a kernel on one stream -- busy with
 * pcie_words_per_thread stores per thread into mapped host memory -- to a kernel on a second stream behind an event, and
 * back.  *stale = number of rounds in which the reader saw another round's value (0 on a stack where stream order across
 * an event carries plain stores between kernels; the assumption every multi-stream use of this library rests on). */
int psx_debug_cross_stream(int device, int rounds, int pcie_words_per_thread, int* stale);

/* Measurement: one pyramid build with the whole-pyramid kernel (k_pyramid_flow) recording, per work item and in ticket
 * order, six int64: the 100 MHz wall clock at dequeue / dependencies met / arithmetic done / published, a word
 * (octave << 40 | level << 32 | chunk << 16 | strip) and the workgroup index.  *nitems = items of the current plan
 * (0 when the launch-per-level schedule is in use); call with host_out = NULL to ask for the size. */
int psx_flow_trace(psx_ctx* ctx, long long* host_out, int capacity_items, int* nitems);

/* In-pipeline timing of the dominant kernel: when enabled, every separable-Gaussian launch of octave 0
 * (levels 1..L-1) INSIDE psx_extract / psx_build_pyramid carries a begin and an end event of its own
 * dispatch (hipExtLaunchKernel: the kernel's start / end timestamps, the quantity rocprofv3 --kernel-trace
 * reports); the launches run back to back with their real producers and consumers, not replayed in isolation.
 * psx_blur_probe_times synchronises and returns the n = L-1 durations (ms) of the last extraction plus the
 * algorithmic bytes of one launch, averaged over the n launches (8 B per pixel of every plane a launch blurs:
 * octave 0, and from level 4 on also the level of octave 1 that shares the launch). */
int psx_enable_blur_probe(psx_ctx* ctx, int on);
int psx_blur_probe_times(psx_ctx* ctx, float* ms, int capacity, int* n, double* bytes_per_launch);

/* Measured HBM roofline: a hand-written 16 B-per-lane streaming copy kernel over `bytes`
 * (0 = 1 GiB) read + `bytes` written, `reps` timed launches after warm-up, HIP events on its own stream;
 * the better of a plain and a non-temporal variant.  avg_ms per launch, bytes_moved = 2 * bytes. */
int psx_copy_bench(int device, size_t bytes, int reps, float* avg_ms, double* bytes_moved);

/* The HIP stream of the context as an opaque handle (hipStream_t), for callers that need to
 * order their own work (e.g. a torch tensor producer) against it. */
void* psx_stream(psx_ctx* ctx);

/* Library version / build info string. */
const char* psx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* POPSIFT_HIP_H */
