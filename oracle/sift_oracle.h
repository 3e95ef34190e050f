/*
 * oracle/sift_oracle.h -- CPU restatement of the PopSift extraction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (popsift_amd/, include/) may
 * include, link or call this.  Allowed users: tests/, __graft_entry__.smoke(),
 * and bench.py -- its cpu_baseline leg (the thing timed there is the oracle itself, as the
 * CPU baseline) and its parity_checked step (the checker of 4 timed frames, outside every
 * timed region).
 *
 * The oracle restates, op by op, the arithmetic of the reference's default path
 * (reference = alicevision/popsift, paths relative to /root/reference/src/popsift):
 *   Gauss tables            gauss_filter.cu:127-371
 *   octave-0 level-0 pass   s_pyramid_build_ra.cu:17-55, s_pyramid_build.cu:96-126
 *   H / V blur              s_pyramid_build_aa.cu:17-86
 *   2x decimation           s_pyramid_build.cu:50-71
 *   DoG                     s_pyramid_build.cu:74-92
 *   extrema + refinement    s_extrema.cu:56-503, s_solve.h:25-86
 *   orientation             s_orientation.cu:39-259, s_gradiant.h:56-69
 *   orientation scan        s_orientation.cu:320-362
 *   descriptor (loop)       s_desc_loop.cu:19-139
 *   normalisation           s_desc_norm_rs.h:42-77, s_desc_norm_l2.h:86-135
 *   output mapping          sift_pyramid.cu:250-280
 *   grid filter             s_filtergrid.cu:36-325
 *   2-NN matcher            features.cu:160-225 (osift_match)
 *   alternative pyramids    s_pyramid_build.cu:478-546, s_pyramid_build_ai.cu:17-132, s_pyramid_build_ra.cu:57-132,
 *                           s_pyramid_build_aa.cu:88-188, s_pyramid_fixed.cu:24-298 (VLFeat_Relative,
 *                           VLFeat_Relative_All, Fixed9 / Fixed15, ScaleDirect)
 *   alternative descriptors s_desc_iloop.cu:19-149, s_desc_grid.cu:19-145, s_desc_igrid.cu:19-108,
 *                           s_desc_notile.cu:31-128, s_gradiant.h:56-88, sift_constants.cu:34-47
 *
 * PARITY PIN STATUS: PINNED.  The reference ships no golden vectors of its own (its goldens are
 * an external reference.tgz fetched by wget, testScripts/downloadOxfordDataset.sh.in:4-9), so
 * the restatement is pinned against OUTPUTS OF THE REFERENCE ITSELF RUN HERE: its sources are
 * compiled for the CPU through the CUDA-emulation shim in oracle/ref_shim (recipe: `make -C
 * oracle ref`, output oracle/_ref/libpopsift_ref.so) and compared with this file
 *   - live in tests/test_ref_shim_cpu.py (Gauss tables and every pyramid plane bit-identical,
 *     feature sets identical within 2e-5 px, descriptors within 2e-6), and
 *   - through the committed fixtures tests/golden/ref_*.npz (tests/golden/make_golden.py).
 * The grid filter is pinned the same way: s_filtergrid.cu itself is compiled against a serial
 * stand-in for the Thrust calls it makes (oracle/ref_shim/thrust_shim.h; sort_by_key = stable sort,
 * which is what Thrust's radix / merge sorts are) -- all three sort modes, live and as fixtures.
 * The 2-NN matcher (osift_match) is pinned against FeaturesDev::match of the reference (features.cu on the
 * shim; its printed result is parsed): indices and accept flags equal, distances to the printed precision.
 * See DESIGN.md "Oracle".
 */
#ifndef SIFT_ORACLE_H
#define SIFT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSIFT_MAX_OCTAVES 20
#define OSIFT_GAUSS_ALIGN 32
#define OSIFT_GAUSS_LEVELS 12
#define OSIFT_ORI_MAX 4

/* enum values mirror popsift::Config (sift_conf.h:38-136) */
enum { OSIFT_GAUSS_VLFEAT_COMPUTE = 0, OSIFT_GAUSS_VLFEAT_RELATIVE = 1,
       OSIFT_GAUSS_VLFEAT_RELATIVE_ALL = 2, OSIFT_GAUSS_OPENCV_COMPUTE = 3,
       OSIFT_GAUSS_FIXED9 = 4, OSIFT_GAUSS_FIXED15 = 5 };
enum { OSIFT_MODE_POPSIFT = 0, OSIFT_MODE_OPENCV = 1, OSIFT_MODE_VLFEAT = 2 };
enum { OSIFT_NORM_ROOTSIFT = 0, OSIFT_NORM_CLASSIC = 1 };
enum { OSIFT_FILTER_RANDOM = 0, OSIFT_FILTER_LARGEST_FIRST = 1, OSIFT_FILTER_SMALLEST_FIRST = 2 };
enum { OSIFT_SCALE_DIRECT = 0, OSIFT_SCALE_DEFAULT = 1 };
enum { OSIFT_DESC_LOOP = 0, OSIFT_DESC_ILOOP = 1, OSIFT_DESC_GRID = 2, OSIFT_DESC_IGRID = 3, OSIFT_DESC_NOTILE = 4 };

typedef struct osift_config {
    int   octaves;            /* -1 = auto (popsift.cpp:118-122) */
    int   levels;             /* 3 */
    float sigma;              /* 1.6 */
    float edge_limit;         /* 10 */
    float threshold;          /* 0.04 */
    float upscale_factor;     /* 1.0 (= -downsampling) */
    int   gauss_mode;         /* VLFeat_Compute */
    int   sift_mode;          /* PopSift */
    int   norm_mode;          /* RootSift */
    int   norm_multi;         /* 0 */
    int   max_extrema;        /* 100000 per octave */
    int   assume_initial_blur;/* 1 */
    float initial_blur;       /* 0.5 */
    int   filter_max_extrema; /* -1 */
    int   filter_grid_size;   /* 2 */
    int   grid_filter_mode;   /* RandomScale */
    int   literal_tex;        /* oracle-only, octave 0 level 0: 1 = per-tap texture coordinates exactly as
                                 s_pyramid_build_ra.cu:35-50; 2 = upsampled-row form; 0 = the
                                 upsampled-row form where it is the same bits (power-of-two scale), literal otherwise */
    int   scaling_mode;       /* ScaleDefault (sift_conf.h:75-80) */
    int   desc_mode;          /* Loop (sift_conf.h:85-97) */
} osift_config;

/* sift_extremum.h:25-39 (fields used on the path) */
typedef struct osift_iext {
    float xpos, ypos;
    int   lpos;
    float sigma;
    int   cell;
    int   ignore;
} osift_iext;

/* sift_extremum.h:47-63 */
typedef struct osift_ext {
    float xpos, ypos;
    int   lpos;
    float sigma;
    int   octave;
    int   num_ori;
    int   idx_ori;
    float orientation[OSIFT_ORI_MAX];
} osift_ext;

/* features.h:23-37, with descriptor pointers replaced by indices (-1 = nullptr) */
typedef struct osift_feature {
    int   debug_octave;
    float xpos, ypos, sigma;
    int   num_ori;
    float orientation[OSIFT_ORI_MAX];
    int   desc_idx[OSIFT_ORI_MAX];
} osift_feature;

typedef struct osift_tables {
    float inc_filter[OSIFT_GAUSS_LEVELS * OSIFT_GAUSS_ALIGN];
    float inc_sigma[OSIFT_GAUSS_LEVELS];
    int   inc_span[OSIFT_GAUSS_LEVELS];
    float dd_filter[OSIFT_MAX_OCTAVES * OSIFT_GAUSS_ALIGN];
    float dd_sigma[OSIFT_MAX_OCTAVES];
    int   dd_span[OSIFT_MAX_OCTAVES];
    /* tables of the alternative pyramid modes (gauss_filter.cu:188-214, 373-410) */
    float abs0_filter[OSIFT_GAUSS_LEVELS * OSIFT_GAUSS_ALIGN];   /* abs_o0: octave 0 directly from the input */
    float abs0_sigma[OSIFT_GAUSS_LEVELS];
    int   abs0_span[OSIFT_GAUSS_LEVELS];
    float absN_filter[OSIFT_GAUSS_LEVELS * OSIFT_GAUSS_ALIGN];   /* abs_oN: levels >= 1 from level 0 of their octave */
    float absN_sigma[OSIFT_GAUSS_LEVELS];
    int   absN_span[OSIFT_GAUSS_LEVELS];
    float inc_ifilter[OSIFT_GAUSS_LEVELS * OSIFT_GAUSS_ALIGN];   /* inc.i_filter: (ratio, multiplier) pairs */
    int   inc_ispan[OSIFT_GAUSS_LEVELS];
} osift_tables;

typedef struct osift_result osift_result;

void  osift_config_default(osift_config* c);
float osift_peak_threshold(const osift_config* c);               /* sift_conf.cu:276-279 */
int   osift_gauss_tables(const osift_config* c, osift_tables* t);/* gauss_filter.cu:127-257 */

/* Full run. img is w*h bytes (is_float=0, 0..255) or w*h floats (is_float=1, [0,1)). */
osift_result* osift_run(const osift_config* c, const void* img, int w, int h, int is_float);
/* Pyramid only (no extrema/orientation/descriptors); for stage-level parity tests. */
osift_result* osift_run_pyramid(const osift_config* c, const void* img, int w, int h, int is_float);
void  osift_free(osift_result* r);

int   osift_num_octaves(const osift_result* r);
int   osift_num_levels(const osift_result* r);        /* levels+3 Gaussian levels */
int   osift_octave_width(const osift_result* r, int o);
int   osift_octave_height(const osift_result* r, int o);
const float* osift_gauss_plane(const osift_result* r, int o, int l);  /* W*H floats, tight */
const float* osift_dog_plane(const osift_result* r, int o, int l);

int   osift_iext_count(const osift_result* r, int o);
const osift_iext* osift_get_iext(const osift_result* r, int o);

int   osift_ext_total(const osift_result* r);
int   osift_ori_total(const osift_result* r);
const osift_ext*     osift_extrema(const osift_result* r);
const osift_feature* osift_features(const osift_result* r);
const float*         osift_descriptors(const osift_result* r);   /* ori_total * 128 */
const int*           osift_feat_to_ext(const osift_result* r);   /* ori_total */

/* Descriptor stage alone on r's pyramid and config for caller-supplied oriented extrema (same layout as the
 * device's psx_extremum): descriptor k of extremum i is written to out + (ext[i].idx_ori + k) * 128 when that index
 * is below n_desc.  For stage-level parity under identical keypoint / orientation bits. */
int   osift_describe(const osift_result* r, const osift_ext* ext, int n_ext, int n_desc, float* out);

/* set number of OpenMP threads used by osift_run (0 = library default) */
void  osift_set_threads(int n);

/* FeaturesDev::match -> compute_distance (features.cu:160-225): for each of the nl left descriptors
 * (128 floats) the two right descriptors with the smallest squared L2 distance, scanned in index order
 * with strict '<'.  out3[3i..] = {best, second, accept = d1/d2 < 0.8f}; dist2[2i..] = {d1, d2} (may be NULL).
 * The distance follows l2_in_t0's operation tree: per float4 t a partial
 * fma(w,w, fma(z,z, fma(x,x, y*y))) (the left-to-right contraction of x*x + y*y + z*z + w*w), then the
 * shuffle_down tree 16/8/4/2/1 as seen by lane 0. */
void osift_match(const float* l, int nl, const float* r, int nr, int* out3, float* dist2);

#ifdef __cplusplus
}
#endif
#endif
