/*
 * oracle/sift_oracle.c -- CPU restatement of the PopSift extraction path (default branch).
 *
 * TEST INFRASTRUCTURE ONLY (see sift_oracle.h).  Every function cites the reference
 * file:line (relative to /root/reference/src/popsift) whose arithmetic it restates.
 *
 * Floating point rules of the restatement (shared with the HIP kernels):
 *   - all arithmetic is IEEE binary32, round-to-nearest, NO implicit contraction
 *     (build with -ffp-contract=off);
 *   - where the reference source has the shape  acc += a * b  inside a filter loop
 *     (nvcc contracts it, -fmad=true is nvcc's default) we write fmaf() explicitly;
 *   - CUDA fast intrinsics (__expf, __sincosf, __fdividef, __frcp_rn, __fsqrt_rn,
 *     __frsqrt_rn) are restated with the correctly rounded libm / IEEE operation;
 *   - __fmul_ru / __fmaf_ru (s_desc_loop.cu:108,118-119) are restated with the MXCSR
 *     rounding mode set to round-up.
 */
#define _GNU_SOURCE
#include "sift_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <xmmintrin.h>
#include "cuda_model.h"     /* -DOSIFT_CUDA_MODEL: sensitivity builds; otherwise every CM_* macro is the plain operation */
#ifdef _OPENMP
#include <omp.h>
#endif

/* sift_constants.h:21-33: M_PI and M_PI2 are *float* constants in device code */
static const float PI_F  = 3.14159265358979323846f;
static const float PI2_F = 2.0f * 3.14159265358979323846f;
#define M_4RPI_F (4.0f / PI_F)
#define ORI_NBINS 36
#define ORI_WINFACTOR 1.5f
#define DESC_MAGNIFY 3.0f
#define DESC_BINS 8

static int g_threads = 0;
void osift_set_threads(int n) { g_threads = n; }

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------- */
/* Config (sift_conf.cu:18-41, 276-279)                                      */
/* ------------------------------------------------------------------------- */
void osift_config_default(osift_config* c)
{
    memset(c, 0, sizeof(*c));
    c->octaves = -1;
    c->levels = 3;
    c->sigma = 1.6f;
    c->edge_limit = 10.0f;
    c->threshold = (float)0.04;
    c->upscale_factor = 1.0f;
    c->gauss_mode = OSIFT_GAUSS_VLFEAT_COMPUTE;
    c->sift_mode = OSIFT_MODE_POPSIFT;
    c->norm_mode = OSIFT_NORM_ROOTSIFT;
    c->norm_multi = 0;
    c->max_extrema = 100000;
    c->assume_initial_blur = 1;
    c->initial_blur = 0.5f;
    c->filter_max_extrema = -1;
    c->scaling_mode = OSIFT_SCALE_DEFAULT;
    c->desc_mode = OSIFT_DESC_LOOP;
    c->filter_grid_size = 2;
    c->grid_filter_mode = OSIFT_FILTER_RANDOM;
    c->literal_tex = 0;
}

float osift_peak_threshold(const osift_config* c)
{
    /* sift_conf.cu:278: ( _threshold * 0.5f * 255.0f / levels ) */
    return c->threshold * 0.5f * 255.0f / c->levels;
}

/* ------------------------------------------------------------------------- */
/* Gauss tables (gauss_filter.cu:127-257, 269-371)                           */
/* ------------------------------------------------------------------------- */
static int vlfeat_span(float sigma)
{
    /* gauss_filter.cu:301-307 */
    return imin((int)(ceilf(4.0f * sigma) + 1), OSIFT_GAUSS_ALIGN - 1);
}
static int opencv_span(float sigma)
{
    /* gauss_filter.cu:321-327 */
    int span = (int)(roundf(2.0f * 4.0f * sigma + 1.0f)) | 1;
    span >>= 1;
    span += 1;
    return imin(span, OSIFT_GAUSS_ALIGN - 1);
}
static int get_span(int gauss_mode, float sigma)
{
    /* gauss_filter.cu:269-298 */
    switch (gauss_mode) {
    case OSIFT_GAUSS_VLFEAT_RELATIVE_ALL:
    case OSIFT_GAUSS_VLFEAT_COMPUTE: return vlfeat_span(sigma);
    case OSIFT_GAUSS_VLFEAT_RELATIVE: { int s = vlfeat_span(sigma); if ((s & 1) == 0) s += 1; return s; }
    case OSIFT_GAUSS_OPENCV_COMPUTE: return opencv_span(sigma);
    case OSIFT_GAUSS_FIXED9: return 5;
    case OSIFT_GAUSS_FIXED15: return 8;
    default: return -1;
    }
}

static void compute_blur_table(int gauss_mode, int nlev, const float* sigma, int* span, float* filter)
{
    /* gauss_filter.cu:341-371 */
    for (int level = 0; level < nlev; level++)
        span[level] = imin(get_span(gauss_mode, sigma[level]), OSIFT_GAUSS_ALIGN - 1);
    for (int level = 0; level < nlev; level++) {
        const float sig = sigma[level];
        const int   spn = span[level];
        float*      f   = filter + level * OSIFT_GAUSS_ALIGN;
        double sum = 1.0;
        f[0] = 1.0f;
        for (int x = 1; x < spn; x++) {
            const float val = (float)exp(-0.5 * (pow((double)x / sig, 2.0)));
            f[x] = val;
            sum += 2.0f * val;
        }
        for (int x = 0; x < spn; x++) f[x] = (float)(f[x] / sum);
        for (int x = spn; x < OSIFT_GAUSS_ALIGN; x++) f[x] = 0;
    }
}

int osift_gauss_tables(const osift_config* c, osift_tables* t)
{
    /* gauss_filter.cu:127-237 */
    const float sigma0 = c->sigma;
    const int   levels = c->levels;
    if (sigma0 > 2.0) return -1;                   /* :131-137 */
    if (levels > OSIFT_GAUSS_LEVELS) return -2;    /* :138-144 */
    if (get_span(c->gauss_mode, 1.0f) < 0) return -3;
    memset(t, 0, sizeof(*t));
    const int stages = levels + 3;
    if (stages > OSIFT_GAUSS_LEVELS) return -2;
    const float initial_blur = c->assume_initial_blur
                             ? c->initial_blur * powf(2.0f, c->upscale_factor) : 0.0f;   /* :168-170 */
    t->inc_sigma[0] = c->assume_initial_blur
                    ? sqrtf(fabsf(sigma0 * sigma0 - initial_blur * initial_blur)) : sigma0; /* :176-178 */
    for (int lvl = 1; lvl < stages; lvl++) {
        const float sigmaP = sigma0 * powf(2.0f, (float)(lvl - 1) / (float)levels);
        const float sigmaS = sigma0 * powf(2.0f, (float)(lvl) / (float)levels);
        t->inc_sigma[lvl] = sqrtf(sigmaS * sigmaS - sigmaP * sigmaP);                    /* :181-185 */
    }
    compute_blur_table(c->gauss_mode, OSIFT_GAUSS_LEVELS, t->inc_sigma, t->inc_span, t->inc_filter);
    for (int oct = 0; oct < OSIFT_MAX_OCTAVES; oct++) {
        float oct_sigma = scalbnf(sigma0, oct);                                          /* :227-235 */
        float b = sqrtf(fabsf(oct_sigma * oct_sigma - initial_blur * initial_blur));
        t->dd_sigma[oct] = scalbnf(b, -oct);
    }
    compute_blur_table(c->gauss_mode, OSIFT_MAX_OCTAVES, t->dd_sigma, t->dd_span, t->dd_filter);

    /* abs_o0 (:188-197): octave 0 directly from the input image */
    for (int lvl = 0; lvl < stages; lvl++) {
        const float sigmaS = sigma0 * powf(2.0f, (float)(lvl) / (float)levels);
        t->abs0_sigma[lvl] = sqrtf(fabsf(sigmaS * sigmaS - initial_blur * initial_blur));
    }
    compute_blur_table(c->gauss_mode, OSIFT_GAUSS_LEVELS, t->abs0_sigma, t->abs0_span, t->abs0_filter);
    /* abs_oN (:199-214): levels >= 1 directly from level 0 of their octave */
    t->absN_sigma[0] = 0;
    for (int lvl = 1; lvl < stages; lvl++) {
        const float sigmaP = sigma0;
        const float sigmaS = sigma0 * powf(2.0f, (float)(lvl) / (float)levels);
        t->absN_sigma[lvl] = sqrtf(sigmaS * sigmaS - sigmaP * sigmaP);
    }
    compute_blur_table(c->gauss_mode, OSIFT_GAUSS_LEVELS, t->absN_sigma, t->absN_span, t->absN_filter);
    /* transformBlurTable (:373-410) of the inc table: tap pairs (x, x+1) become a ratio u = a/(a+b) (odd index)
     * and a multiplier v = a+b (even index) for one hardware-interpolated fetch per pair */
    for (int level = 0; level < OSIFT_GAUSS_LEVELS; level++) {
        int isp = t->inc_span[level];
        if (!(isp & 1)) isp += 1;
        t->inc_ispan[level] = isp;
        const float* f = t->inc_filter + level * OSIFT_GAUSS_ALIGN;
        float* fi = t->inc_ifilter + level * OSIFT_GAUSS_ALIGN;
        for (int x = 1; x < isp; x += 2) {
            const float a = f[x], b = f[x + 1];
            fi[x] = a / (a + b);
            fi[x + 1] = a + b;
        }
        fi[0] = f[0];
        for (int x = isp; x < OSIFT_GAUSS_ALIGN; x++) fi[x] = 0;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Result container                                                          */
/* ------------------------------------------------------------------------- */
struct osift_result {
    osift_config cfg;
    osift_tables tab;
    int   num_octaves;
    int   L;                         /* levels + 3 */
    int   W[OSIFT_MAX_OCTAVES], H[OSIFT_MAX_OCTAVES];
    float* data[OSIFT_MAX_OCTAVES];  /* L planes of W*H */
    float* dog[OSIFT_MAX_OCTAVES];   /* L-1 planes */
    osift_iext* iext[OSIFT_MAX_OCTAVES];
    int*   iext_off[OSIFT_MAX_OCTAVES];   /* indices of non-ignored initial extrema (i_ext_off) */
    int   iext_ct[OSIFT_MAX_OCTAVES];     /* found (capped) */
    int   ext_ct[OSIFT_MAX_OCTAVES];      /* after grid filter */
    int   ext_total, ori_total;
    osift_ext*     ext;
    osift_feature* feat;
    float*         desc;
    int*           feat_to_ext;
};

int osift_num_octaves(const osift_result* r) { return r->num_octaves; }
int osift_num_levels(const osift_result* r) { return r->L; }
int osift_octave_width(const osift_result* r, int o) { return r->W[o]; }
int osift_octave_height(const osift_result* r, int o) { return r->H[o]; }
const float* osift_gauss_plane(const osift_result* r, int o, int l)
{ return r->data[o] + (size_t)l * r->W[o] * r->H[o]; }
const float* osift_dog_plane(const osift_result* r, int o, int l)
{ return r->dog[o] + (size_t)l * r->W[o] * r->H[o]; }
int osift_iext_count(const osift_result* r, int o) { return r->iext_ct[o]; }
const osift_iext* osift_get_iext(const osift_result* r, int o) { return r->iext[o]; }
int osift_ext_total(const osift_result* r) { return r->ext_total; }
int osift_ori_total(const osift_result* r) { return r->ori_total; }
const osift_ext* osift_extrema(const osift_result* r) { return r->ext; }
const osift_feature* osift_features(const osift_result* r) { return r->feat; }
const float* osift_descriptors(const osift_result* r) { return r->desc; }
const int* osift_feat_to_ext(const osift_result* r) { return r->feat_to_ext; }

void osift_free(osift_result* r)
{
    if (!r) return;
    for (int o = 0; o < OSIFT_MAX_OCTAVES; o++) {
        free(r->data[o]); free(r->dog[o]); free(r->iext[o]); free(r->iext_off[o]);
    }
    free(r->ext); free(r->feat); free(r->desc); free(r->feat_to_ext);
    free(r);
}

/* ------------------------------------------------------------------------- */
/* Input image as a CUDA texture (s_image.cu:138-167): normalised coordinates, */
/* clamp addressing, bilinear filtering, u8 read as v/255 (NormalizedFloat).  */
/* Filtering model: CUDA C Programming Guide, "Texture Fetching / Linear      */
/* Filtering": xB = x - 0.5, i = floor(xB), alpha = frac(xB) held in 1.8       */
/* fixed point (8 fractional bits).                                          */
/* ------------------------------------------------------------------------- */
typedef struct { const void* px; int w, h, is_float; } tex_in;

static inline float texel(const tex_in* t, int i, int j)
{
    i = clampi(i, 0, t->w - 1);
    j = clampi(j, 0, t->h - 1);
    if (t->is_float) return ((const float*)t->px)[(size_t)j * t->w + i];
    return (float)((const uint8_t*)t->px)[(size_t)j * t->w + i] / 255.0f;
}

static inline void tex_axis(float cn, int size, int* i0, float* a)
{
    const float tcoord = cn * (float)size;
    const float tb = tcoord - 0.5f;
    const float fl = floorf(tb);
    float al = tb - fl;
    al = rintf(al * 256.0f) * (1.0f / 256.0f);
    *i0 = (int)fl;
    *a = al;
}

static inline float lerpf(float p, float q, float a)
{
    return fmaf(a, q, (1.0f - a) * p);
}

static float tex2d_norm(const tex_in* t, float un, float vn)
{
    int i0, j0; float a, b;
    tex_axis(un, t->w, &i0, &a);
    tex_axis(vn, t->h, &j0, &b);
    const float r0 = lerpf(texel(t, i0, j0),     texel(t, i0 + 1, j0),     a);
    const float r1 = lerpf(texel(t, i0, j0 + 1), texel(t, i0 + 1, j0 + 1), a);
    return lerpf(r0, r1, b);
}

/* ------------------------------------------------------------------------- */
/* Pyramid                                                                   */
/* ------------------------------------------------------------------------- */

/* a == b << n or b == a << n for a small n: the sub-texel positions (x +- k + shift)/W * w - 0.5 are then multiples of
 * 2^-(n+1), far from the rounding boundaries of the 1.8 fixed-point filter weight */
static int pow2_ratio(int a, int b)
{
    const int big = a > b ? a : b, small = a > b ? b : a;
    for (int n = 0; n <= 4; n++) if ((small << n) == big) return 1;
    return 0;
}

/* s_pyramid_build_ra.cu:17-55 (normalizedSource::horiz) followed by
 * s_pyramid_build_aa.cu:52-86 (absoluteSource::vert, level 0). */
static void octave0_level0(osift_result* r, const tex_in* t)
{
    const osift_config* c = &r->cfg;
    const int W = r->W[0], H = r->H[0];
    const int   span   = r->tab.dd_span[0];
    const float* filter = r->tab.dd_filter;           /* octave 0 row */
    /* s_pyramid_build.cu:109-114 */
    float shift = 0.5f;
    if (c->sift_mode == OSIFT_MODE_POPSIFT || c->sift_mode == OSIFT_MODE_VLFEAT)
        shift = 0.5f * powf(2.0f, c->upscale_factor - 0);

    float* intm = (float*)malloc(sizeof(float) * (size_t)W * H);
    /* literal_tex 0: the cheaper upsampled-row form where it is provably the same bits (power-of-two ratio between
     * image and octave), the literal per-tap coordinates everywhere else; 1: always literal; 2: always the
     * upsampled-row form (tests/test_oracle_cpu.py shows where the two part) */
    const int literal = c->literal_tex == 1 || (c->literal_tex == 0 && !(pow2_ratio(W, t->w) && pow2_ratio(H, t->h)));
    if (literal) {
        /* per-tap texture coordinates exactly as the reference computes them */
        #pragma omp parallel for schedule(static)
        for (int y = 0; y < H; y++) {
            const float read_y = ((float)y + shift) / H;
            for (int x = 0; x < W; x++) {
                const float read_x = ((float)x + shift) / W;
                float out = 0.0f;
                for (int offset = span; offset > 0; offset--) {
                    const float g = filter[offset];
                    const float offrel = (float)offset / W;
                    const float v1 = tex2d_norm(t, read_x - offrel, read_y);
                    const float v2 = tex2d_norm(t, read_x + offrel, read_y);
                    out = fmaf(v1 + v2, g, out);
                }
                out = fmaf(tex2d_norm(t, read_x, read_y), filter[0], out);
                intm[(size_t)y * W + x] = out * 255.0f;
            }
        }
    } else {
        /* upsampled-row form: tap k of output x reads U(x-k) / U(x+k), where U(X) is the
         * texture fetch at the coordinate of output column X.  Identical to the literal form
         * whenever (x +- k + shift)/W * w lands on the same 1/256 sub-texel, which holds for
         * every power-of-two scale factor (DESIGN.md "octave 0"). */
        const int R = span;  /* filter[span] == 0, taps 1..span-1 contribute */
        #pragma omp parallel
        {
            float* U = (float*)malloc(sizeof(float) * (size_t)(W + 2 * R));
            #pragma omp for schedule(static)
            for (int y = 0; y < H; y++) {
                const float read_y = ((float)y + shift) / H;
                for (int X = -R; X < W + R; X++)
                    U[X + R] = tex2d_norm(t, ((float)X + shift) / W, read_y);
                for (int x = 0; x < W; x++) {
                    float out = 0.0f;
                    for (int offset = span; offset > 0; offset--)
                        out = fmaf(U[x - offset + R] + U[x + offset + R], filter[offset], out);
                    out = fmaf(U[x + R], filter[0], out);
                    intm[(size_t)y * W + x] = out * 255.0f;
                }
            }
            free(U);
        }
    }
    /* vertical pass, inc table level 0 */
    const int   vspan = r->tab.inc_span[0];
    const float* vf   = r->tab.inc_filter;
    float* dst = r->data[0];
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            float out = 0.0f;
            for (int offset = vspan; offset > 0; offset--) {
                const float g = vf[offset];
                out = fmaf(intm[(size_t)clampi(y - offset, 0, H - 1) * W + x], g, out);
                out = fmaf(intm[(size_t)clampi(y + offset, 0, H - 1) * W + x], g, out);
            }
            out = fmaf(intm[(size_t)y * W + x], vf[0], out);
            dst[(size_t)y * W + x] = out;
        }
    }
    free(intm);
}

/* s_pyramid_build_aa.cu:17-50 (absoluteSource::horiz) then :52-86 (vert), level l from l-1 */
static void blur_level(osift_result* r, int o, int level)
{
    const int W = r->W[o], H = r->H[o];
    const int   span = r->tab.inc_span[level];
    const float* f   = r->tab.inc_filter + level * OSIFT_GAUSS_ALIGN;
    const float* src = r->data[o] + (size_t)(level - 1) * W * H;
    float*       dst = r->data[o] + (size_t)level * W * H;
    float* intm = (float*)malloc(sizeof(float) * (size_t)W * H);
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const float* row = src + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            /* centre first, then the (zero-weight) outermost pair, then inwards */
            float out = 0.0f;
            out = fmaf(row[x], f[0], out);
            out = fmaf(row[clampi(x - span, 0, W - 1)] + row[clampi(x + span, 0, W - 1)], f[span], out);
            for (int offset = span - 1; offset > 0; offset--) {
                const float D = row[clampi(x - offset, 0, W - 1)];
                const float E = row[clampi(x + offset, 0, W - 1)];
                out = fmaf(D + E, f[offset], out);
            }
            intm[(size_t)y * W + x] = out;
        }
    }
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            float out = 0.0f;
            for (int offset = span; offset > 0; offset--) {
                const float g = f[offset];
                out = fmaf(intm[(size_t)clampi(y - offset, 0, H - 1) * W + x], g, out);
                out = fmaf(intm[(size_t)clampi(y + offset, 0, H - 1) * W + x], g, out);
            }
            out = fmaf(intm[(size_t)y * W + x], f[0], out);
            dst[(size_t)y * W + x] = out;
        }
    }
    free(intm);
}

/* s_pyramid_build.cu:50-71 (get_by_2_pick_every_second), src level = L-3 (PREV_LEVEL 3, :228) */
static void downscale(osift_result* r, int o)
{
    const int W = r->W[o], H = r->H[o];
    const int sw = r->W[o - 1], sh = r->H[o - 1];
    const float* src = r->data[o - 1] + (size_t)(r->L - 3) * sw * sh;
    float* dst = r->data[o];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int rx = imin(imax(x << 1, 0), sw - 1);   /* common/clamp.h:17-23 */
            const int ry = imin(imax(y << 1, 0), sh - 1);
            dst[(size_t)y * W + x] = src[(size_t)ry * sw + rx];
        }
}

/* s_pyramid_build.cu:74-92 (make_dog) */
static void make_dog(osift_result* r, int o)
{
    const size_t n = (size_t)r->W[o] * r->H[o];
    for (int l = 0; l < r->L - 1; l++) {
        const float* a = r->data[o] + l * n;
        const float* b = r->data[o] + (l + 1) * n;
        float* d = r->dog[o] + l * n;
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) d[i] = b[i] - a[i];
    }
}

/* ------------------------------------------------------------------------- */
/* Alternative pyramid modes (s_pyramid_build.cu:478-546)                     */
/* ------------------------------------------------------------------------- */

/* readTex (assist.h:68-77) on a LINEAR-filtered layered texture of one plane: unnormalised coordinates,
 * clamp addressing, filtering model as for the input image (1.8 fixed-point weights).  x, y are the
 * arguments of readTex, which adds 0.5 to both. */
static float plane_linear(const float* p, int W, int H, float x, float y)
{
    const float xs = x + 0.5f, ys = y + 0.5f;
    const float xb = xs - 0.5f, yb = ys - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    float a = xb - fx, b = yb - fy;
    a = rintf(a * 256.0f) * (1.0f / 256.0f);
    b = rintf(b * 256.0f) * (1.0f / 256.0f);
    const int i = (int)fx, j = (int)fy;
    const int i0 = clampi(i, 0, W - 1), i1 = clampi(i + 1, 0, W - 1);
    const int j0 = clampi(j, 0, H - 1), j1 = clampi(j + 1, 0, H - 1);
    const float r0 = lerpf(p[(size_t)j0 * W + i0], p[(size_t)j0 * W + i1], a);
    const float r1 = lerpf(p[(size_t)j1 * W + i0], p[(size_t)j1 * W + i1], a);
    return lerpf(r0, r1, b);
}

/* normalizedSource::horiz / horiz_level / horiz_all (s_pyramid_build_ra.cu:17-132): one level of octave o
 * filtered horizontally straight from the input image, per-tap texture coordinates, result * 255 */
static void h_from_input(const tex_in* t, int W, int H, const float* filter, int span, float shift, float* intm)
{
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const float read_y = ((float)y + shift) / H;
        for (int x = 0; x < W; x++) {
            const float read_x = ((float)x + shift) / W;
            float out = 0.0f;
            for (int offset = span; offset > 0; offset--) {
                const float g = filter[offset];
                const float offrel = (float)offset / W;
                const float v1 = tex2d_norm(t, read_x - offrel, read_y);
                const float v2 = tex2d_norm(t, read_x + offrel, read_y);
                out = fmaf(v1 + v2, g, out);
            }
            out = fmaf(tex2d_norm(t, read_x, read_y), filter[0], out);
            intm[(size_t)y * W + x] = out * 255.0f;
        }
    }
}

/* absoluteSource::vert / vert_abs0 (s_pyramid_build_aa.cu:52-122) */
static void v_plain(const float* intm, float* dst, int W, int H, const float* f, int span)
{
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float out = 0.0f;
            for (int offset = span; offset > 0; offset--) {
                const float g = f[offset];
                out = fmaf(intm[(size_t)clampi(y - offset, 0, H - 1) * W + x], g, out);
                out = fmaf(intm[(size_t)clampi(y + offset, 0, H - 1) * W + x], g, out);
            }
            out = fmaf(intm[(size_t)y * W + x], f[0], out);
            dst[(size_t)y * W + x] = out;
        }
}

/* absoluteSourceInterpolated::horiz / vert (s_pyramid_build_ai.cu:17-69): tap pairs through one linear fetch */
static void hv_interp(const float* src, float* dst, int W, int H, const float* fi, int ispan, int vertical)
{
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float out = 0.0f;
            for (int offset = 1; offset <= ispan; offset += 2) {
                const float u = fi[offset];
                const float off = offset + (1.0f - u);
                float val;
                if (vertical) val = plane_linear(src, W, H, (float)x, (float)y - off) + plane_linear(src, W, H, (float)x, (float)y + off);
                else          val = plane_linear(src, W, H, (float)x - off, (float)y) + plane_linear(src, W, H, (float)x + off, (float)y);
                const float v = fi[offset + 1];
                out = fmaf(val, v, out);
            }
            out = fmaf(plane_linear(src, W, H, (float)x, (float)y), fi[0], out);
            dst[(size_t)y * W + x] = out;
        }
}

/* absoluteSource::horiz (s_pyramid_build_aa.cu:17-50) */
static void h_plain(const float* src, float* intm, int W, int H, const float* f, int span)
{
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const float* row = src + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float out = 0.0f;
            out = fmaf(row[x], f[0], out);
            out = fmaf(row[clampi(x - span, 0, W - 1)] + row[clampi(x + span, 0, W - 1)], f[span], out);
            for (int offset = span - 1; offset > 0; offset--)
                out = fmaf(row[clampi(x - offset, 0, W - 1)] + row[clampi(x + offset, 0, W - 1)], f[offset], out);
            intm[(size_t)y * W + x] = out;
        }
    }
}

/* fixedSpan::relativeTexAddress::octave_fixed (s_pyramid_fixed.cu:120-190): octave 0, every level straight
 * from the input image with the first SHIFT+1 taps of abs_o0; vertical first, then horizontal (the warp
 * shuffles of octave_fixed_horiz see lane N's vertical value for column idx-SHIFT and hand lane N the sum
 * centred on column idx) */
static void fixed_octave0(osift_result* r, const tex_in* t, int SHIFT)
{
    const osift_config* c = &r->cfg;
    const int W = r->W[0], H = r->H[0];
    const float tshift = 0.5f * powf(2.0f, c->upscale_factor);
    const float mul_w = 1.0f / (float)W, mul_h = 1.0f / (float)H;          /* __frcp_rn */
    float* vbuf = (float*)malloc(sizeof(float) * (size_t)(W + 2 * SHIFT) * H);
    for (int level = 0; level < r->L; level++) {
        const float* f = r->tab.abs0_filter + level * OSIFT_GAUSS_ALIGN;
        float* dst = r->data[0] + (size_t)level * W * H;
        const int VW = W + 2 * SHIFT;
        #pragma omp parallel for schedule(static)
        for (int y = 0; y < H; y++)
            for (int cx = -SHIFT; cx < W + SHIFT; cx++) {
                const float xpos = ((float)cx + tshift) * mul_w;
                const float ypos = ((float)y + tshift) * mul_h;
                float val = tex2d_norm(t, xpos, ypos);
                float fval = val * f[0];
                for (int i = 1; i <= SHIFT; i++) {
                    /* "ypos - i * mul_h" in device code: nvcc contracts it into one fma (-fmad=true), and so does the
                     * reference built for the CPU (oracle/Makefile REF_CXX); two roundings instead of one move a tap across
                     * a 1/256 sub-texel boundary once in a while when H is not a power-of-two multiple of the image
                     * height (found by tools/ref_fuzz.py: one row of a 193 x 191 plane at upscale 0.5) */
                    val  = tex2d_norm(t, xpos, fmaf(-(float)i, mul_h, ypos));
                    val += tex2d_norm(t, xpos, fmaf((float)i, mul_h, ypos));
                    fval = fmaf(val, f[i], fval);
                }
                vbuf[(size_t)y * VW + cx + SHIFT] = fval;
            }
        #pragma omp parallel for schedule(static)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float* v = vbuf + (size_t)y * VW + x + SHIFT;
                float out = v[0] * f[0];
                for (int i = 1; i <= SHIFT; i++) out = fmaf(v[-i] + v[i], f[i], out);
                dst[(size_t)y * W + x] = out * 255.0f;
            }
    }
    free(vbuf);
}

/* fixedSpan::absoluteTexAddress::octave_fixed (s_pyramid_fixed.cu:48-116): levels 1..L-1 of an octave from its
 * level 0 with the first SHIFT+1 taps of abs_oN */
static void fixed_octaveN(osift_result* r, int o, int SHIFT)
{
    const int W = r->W[o], H = r->H[o];
    const float* src = r->data[o];
    const int VW = W + 2 * SHIFT;
    float* vbuf = (float*)malloc(sizeof(float) * (size_t)VW * H);
    for (int level = 1; level < r->L; level++) {
        const float* f = r->tab.absN_filter + level * OSIFT_GAUSS_ALIGN;
        float* dst = r->data[o] + (size_t)level * W * H;
        #pragma omp parallel for schedule(static)
        for (int y = 0; y < H; y++)
            for (int cx = -SHIFT; cx < W + SHIFT; cx++) {
                const int xc = clampi(cx, 0, W - 1);
                float val = src[(size_t)y * W + xc];
                float fval = val * f[0];
                for (int i = 1; i <= SHIFT; i++) {
                    val = src[(size_t)clampi(y - i, 0, H - 1) * W + xc] + src[(size_t)clampi(y + i, 0, H - 1) * W + xc];
                    fval = fmaf(val, f[i], fval);
                }
                vbuf[(size_t)y * VW + cx + SHIFT] = fval;
            }
        #pragma omp parallel for schedule(static)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float* v = vbuf + (size_t)y * VW + x + SHIFT;
                float out = v[0] * f[0];
                for (int i = 1; i <= SHIFT; i++) out = fmaf(v[-i] + v[i], f[i], out);
                dst[(size_t)y * W + x] = out;
            }
    }
    free(vbuf);
}

/* Pyramid::build_pyramid (s_pyramid_build.cu:459-594): all branches.  Returns 0, or -1 for the combinations
 * the reference rejects with POP_FATAL (Fixed9 / Fixed15 with levels != 3, s_pyramid_fixed.cu:270-292). */
static int build_pyramid(osift_result* r, const tex_in* t)
{
    const osift_config* c = &r->cfg;
    const int gm = c->gauss_mode;
    const int fixed = (gm == OSIFT_GAUSS_FIXED9 || gm == OSIFT_GAUSS_FIXED15);
    const int direct = (c->scaling_mode == OSIFT_SCALE_DIRECT);
    const int SHIFT = (gm == OSIFT_GAUSS_FIXED9) ? 4 : 7;
    if (fixed && r->L != 6) return -1;
    if (!fixed && !direct && (gm == OSIFT_GAUSS_VLFEAT_COMPUTE || gm == OSIFT_GAUSS_OPENCV_COMPUTE)) {
        /* default branch, s_pyramid_build.cu:547-586 */
        for (int o = 0; o < r->num_octaves; o++) {
            if (o == 0) octave0_level0(r, t);
            else downscale(r, o);
            for (int level = 1; level < r->L; level++) blur_level(r, o, level);
        }
        return 0;
    }
    const osift_tables* T = &r->tab;
    for (int o = 0; o < r->num_octaves; o++) {
        const int W = r->W[o], H = r->H[o];
        const size_t n = (size_t)W * H;
        float* intm = (float*)malloc(sizeof(float) * n);
        /* horiz_from_input_image (s_pyramid_build.cu:96-126) */
        float shift = 0.5f;
        if (o == 0 && (c->sift_mode == OSIFT_MODE_POPSIFT || c->sift_mode == OSIFT_MODE_VLFEAT))
            shift = 0.5f * powf(2.0f, c->upscale_factor - o);
        if (fixed) {
            if (o == 0) fixed_octave0(r, t, SHIFT);
            else {
                if (direct) {
                    h_from_input(t, W, H, T->dd_filter + o * OSIFT_GAUSS_ALIGN, T->dd_span[o], shift, intm);
                    v_plain(intm, r->data[o], W, H, T->inc_filter, T->inc_span[0]);
                } else downscale(r, o);
                fixed_octaveN(r, o, SHIFT);
            }
        } else if (direct) {
            const int interp = (gm == OSIFT_GAUSS_VLFEAT_RELATIVE);
            for (int level = 0; level < r->L; level++) {
                float* dst = r->data[o] + level * n;
                if (level == 0) h_from_input(t, W, H, T->dd_filter + o * OSIFT_GAUSS_ALIGN, T->dd_span[o], shift, intm);
                else if (interp) hv_interp(r->data[o] + (level - 1) * n, intm, W, H, T->inc_ifilter + level * OSIFT_GAUSS_ALIGN, T->inc_ispan[level], 0);
                else h_plain(r->data[o] + (level - 1) * n, intm, W, H, T->inc_filter + level * OSIFT_GAUSS_ALIGN, T->inc_span[level]);
                if (interp) hv_interp(intm, dst, W, H, T->inc_ifilter + level * OSIFT_GAUSS_ALIGN, T->inc_ispan[level], 1);
                else v_plain(intm, dst, W, H, T->inc_filter + level * OSIFT_GAUSS_ALIGN, T->inc_span[level]);
            }
        } else if (gm == OSIFT_GAUSS_VLFEAT_RELATIVE) {
            for (int level = 0; level < r->L; level++) {
                float* dst = r->data[o] + level * n;
                const float* fi = T->inc_ifilter + level * OSIFT_GAUSS_ALIGN;
                if (level == 0) {
                    if (o == 0) {
                        h_from_input(t, W, H, T->dd_filter, T->dd_span[0], shift, intm);
                        hv_interp(intm, dst, W, H, fi, T->inc_ispan[0], 1);
                    } else downscale(r, o);
                } else {
                    hv_interp(r->data[o] + (level - 1) * n, intm, W, H, fi, T->inc_ispan[level], 0);
                    hv_interp(intm, dst, W, H, fi, T->inc_ispan[level], 1);
                }
            }
        } else if (o == 0 && gm == OSIFT_GAUSS_VLFEAT_RELATIVE_ALL) {
            /* horiz_all_from_input_image + vert_all_from_interm(NotInterpolated_FromFirst) */
            for (int level = 0; level < r->L; level++) {
                h_from_input(t, W, H, T->abs0_filter + level * OSIFT_GAUSS_ALIGN, T->abs0_span[level], shift, intm);
                v_plain(intm, r->data[0] + level * n, W, H, T->abs0_filter + level * OSIFT_GAUSS_ALIGN, T->abs0_span[level]);
            }
        } else {
            if (o == 0) octave0_level0(r, t);
            else downscale(r, o);
            for (int level = 1; level < r->L; level++) blur_level(r, o, level);
        }
        free(intm);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Extrema (s_extrema.cu:56-503, s_solve.h:25-86)                             */
/* ------------------------------------------------------------------------- */
typedef struct { const float* dog; int W, H, NL; } dogtex;

/* point texture, clamp in x,y (sift_octave.cu:233-236); layer index clamped (PTX tex.a2d) */
static inline float rdog(const dogtex* t, int x, int y, int z)
{
    x = clampi(x, 0, t->W - 1);
    y = clampi(y, 0, t->H - 1);
    z = clampi(z, 0, t->NL - 1);
    return t->dog[((size_t)z * t->H + y) * t->W + x];
}

/* s_extrema.cu:56-120: strict maximum or strict minimum over the 26 neighbours */
static int is_extremum(const dogtex* t, int x, int y, int z)
{
    const float val = rdog(t, x, y, z);
    int gt = 1, lt = 1;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (dx == 0 && dy == 0 && dz == 0) continue;
                const float f = rdog(t, x + dx, y + dy, z + dz);
                gt &= (val > f);
                lt &= (val < f);
            }
    return gt || lt;
}

/* s_solve.h:25-86 */
static int solve3(float i[3][3], float b[3])
{
    float det0b = -i[1][2] * i[1][2];
    float det0a =  i[1][1] * i[2][2];
    float det0  = det0b + det0a;
    float det1b = -i[0][1] * i[2][2];
    float det1a =  i[1][2] * i[0][2];
    float det1  = det1b + det1a;
    float det2b = -i[1][1] * i[0][2];
    float det2a =  i[0][1] * i[1][2];
    float det2  = det2b + det2a;
    float det3b = -i[0][2] * i[0][2];
    float det3a =  i[0][0] * i[2][2];
    float det3  = det3b + det3a;
    float det4b = -i[0][0] * i[1][2];
    float det4a =  i[0][1] * i[0][2];
    float det4  = det4b + det4a;
    float det5b = -i[0][1] * i[0][1];
    float det5a =  i[0][0] * i[1][1];
    float det5  = det5b + det5a;

    float det;
    det  = (i[0][0] * det0);
    det += (i[0][1] * det1);
    det += (i[0][2] * det2);
    if (det == 0) return 0;
    float rsd = 1.0f / det;   /* __frcp_rn */

    i[0][0] = det0 * rsd;
    i[1][0] = det1 * rsd;
    i[2][0] = det2 * rsd;
    i[1][1] = det3 * rsd;
    i[1][2] = det4 * rsd;
    i[2][2] = det5 * rsd;
    i[0][1] = i[1][0];
    i[0][2] = i[2][0];
    i[2][1] = i[1][2];

    float vout[3] = {0, 0, 0};
    for (int y = 0; y < 3; y++) {
        vout[y] += (i[y][0] * b[0]);
        vout[y] += (i[y][1] * b[1]);
        vout[y] += (i[y][2] * b[2]);
    }
    b[0] = vout[0]; b[1] = vout[1]; b[2] = vout[2];
    return 1;
}

/* ModeFunctions<mode>::refine, s_extrema.cu:155-284. returns -1 fail, 0 continue, 1 done */
static int refine_step(int mode, const float d[3], int n[3], int width, int height, int maxlevel, int last_it)
{
    if (mode == OSIFT_MODE_OPENCV) {
        const float tx = fabsf(d[0]), ty = fabsf(d[1]), tz = fabsf(d[2]);
        if (tx < 0.5f && ty < 0.5f && tz < 0.5f) return 1;
        n[0] += (int)roundf(d[0]);
        n[1] += (int)roundf(d[1]);
        n[2] += (int)roundf(d[2]);
        return (n[0] < 5 || n[0] >= width - 5 || n[1] < 5 || n[1] >= height - 5 ||
                n[2] < 1 || n[2] > maxlevel - 2) ? -1 : 0;
    }
    if (last_it) return 0;
    int tx = ((d[0] >= 0.6f && n[0] < width - 2) ? 1 : 0) + ((d[0] <= -0.6f && n[0] > 1) ? -1 : 0);
    int ty = ((d[1] >= 0.6f && n[1] < height - 2) ? 1 : 0) + ((d[1] <= -0.6f && n[1] > 1) ? -1 : 0);
    int tz = 0;
    if (mode == OSIFT_MODE_POPSIFT)
        tz = ((d[2] >= 0.6f && n[2] < maxlevel - 1) ? 1 : 0) + ((d[2] <= -0.6f && n[2] > 1) ? -1 : 0);
    if (tx == 0 && ty == 0 && tz == 0) return 1;
    n[0] += tx; n[1] += ty; n[2] += tz;
    return 0;
}

/* find_extrema_in_dog_sub, s_extrema.cu:298-503 */
static int find_extremum_at(const osift_config* c, const dogtex* t, int x, int y, int level,
                            float thr, int maxlevel, float wdiv, float hdiv, osift_iext* ec)
{
    const int width = t->W, height = t->H;
    const int mode = c->sift_mode;
    if (mode == OSIFT_MODE_OPENCV)
        if (x < 5 || y < 5 || x >= width - 5 || y >= height - 5) return 0;

    const float val = rdog(t, x, y, level);
    /* first_contrast_ok :149-153, :200-204, :252-256 */
    if (mode == OSIFT_MODE_OPENCV) { if (!(fabsf(val) >= floorf(thr))) return 0; }
    else if (mode == OSIFT_MODE_VLFEAT) { if (!(fabsf(val) >= 0.8f * 2.0f * thr)) return 0; }
    else { if (!(fabsf(val) >= 1.6f * thr)) return 0; }

    if (!is_extremum(t, x, y, level)) return 0;

    float D[3], DD[3], DX[3], d[3] = {0, 0, 0};
    const float v = val;
    int n[3] = {x, y, level};
    int iter = 0;
    const int MAX_ITERATIONS = 5;
    do {
        iter++;
        const float x2y1z1 = rdog(t, n[0] + 1, n[1], n[2]);
        const float x0y1z1 = rdog(t, n[0] - 1, n[1], n[2]);
        const float x1y2z1 = rdog(t, n[0], n[1] + 1, n[2]);
        const float x1y0z1 = rdog(t, n[0], n[1] - 1, n[2]);
        const float x1y1z2 = rdog(t, n[0], n[1], n[2] + 1);
        const float x1y1z0 = rdog(t, n[0], n[1], n[2] - 1);
        D[0] = scalbnf(x2y1z1 - x0y1z1, -1);
        D[1] = scalbnf(x1y2z1 - x1y0z1, -1);
        D[2] = scalbnf(x1y1z2 - x1y1z0, -1);

        const float x1y1z1 = rdog(t, n[0], n[1], n[2]);
        DD[0] = x2y1z1 + x0y1z1 - scalbnf(x1y1z1, 1);
        DD[1] = x1y2z1 + x1y0z1 - scalbnf(x1y1z1, 1);
        DD[2] = x1y1z2 + x1y1z0 - scalbnf(x1y1z1, 1);

        const float x0y0z1 = rdog(t, n[0] - 1, n[1] - 1, n[2]);
        const float x0y1z0 = rdog(t, n[0] - 1, n[1], n[2] - 1);
        const float x0y1z2 = rdog(t, n[0] - 1, n[1], n[2] + 1);
        const float x0y2z1 = rdog(t, n[0] - 1, n[1] + 1, n[2]);
        const float x1y0z0 = rdog(t, n[0], n[1] - 1, n[2] - 1);
        const float x1y0z2 = rdog(t, n[0], n[1] - 1, n[2] + 1);
        const float x1y2z0 = rdog(t, n[0], n[1] + 1, n[2] - 1);
        const float x1y2z2 = rdog(t, n[0], n[1] + 1, n[2] + 1);
        const float x2y0z1 = rdog(t, n[0] + 1, n[1] - 1, n[2]);
        const float x2y1z0 = rdog(t, n[0] + 1, n[1], n[2] - 1);
        const float x2y1z2 = rdog(t, n[0] + 1, n[1], n[2] + 1);
        const float x2y2z1 = rdog(t, n[0] + 1, n[1] + 1, n[2]);
        DX[0] = scalbnf(x2y2z1 + x0y0z1 - x0y2z1 - x2y0z1, -2);
        DX[1] = scalbnf(x2y1z2 + x0y1z0 - x0y1z2 - x2y1z0, -2);
        DX[2] = scalbnf(x1y2z2 + x1y0z0 - x1y2z0 - x1y0z2, -2);

        float b[3];
        float A[3][3];
        A[0][0] = DD[0];
        A[1][1] = DD[1];
        A[2][2] = DD[2];
        A[1][0] = A[0][1] = DX[0];
        A[2][0] = A[0][2] = DX[1];
        A[2][1] = A[1][2] = DX[2];
        b[0] = -D[0]; b[1] = -D[1]; b[2] = -D[2];

        if (!solve3(A, b)) { d[0] = d[1] = d[2] = 0; break; }
        d[0] = b[0]; d[1] = b[1]; d[2] = b[2];

        const int retval = refine_step(mode, d, n, width, height, maxlevel, iter == MAX_ITERATIONS);
        if (retval == -1) return 0;
        else if (retval == 1) break;
    } while (iter < MAX_ITERATIONS);

    if (iter >= MAX_ITERATIONS && mode == OSIFT_MODE_OPENCV) return 0;            /* :447-452 */
    if (mode == OSIFT_MODE_POPSIFT || mode == OSIFT_MODE_VLFEAT)
        if (d[0] >= 1.5f || d[1] >= 1.5f || d[2] >= 1.5f) return 0;              /* :454-459 */

    const float xn = n[0] + d[0];
    const float yn = n[1] + d[1];
    const float sn = n[2] + d[2];

    if (mode != OSIFT_MODE_OPENCV) {                                             /* verify :234-245, :286-297 */
        if (xn < 0.0f || xn > width - 1.0f || yn < 0.0f || yn > height - 1.0f ||
            sn < 0.0f || sn > maxlevel) return 0;
    }

    const float contr   = v + scalbnf(D[0] * d[0] + D[1] * d[1] + D[2] * d[2], -1);
    const float tr      = DD[0] + DD[1];
    const float det     = DD[0] * DD[1] - DX[0] * DX[0];
    const float edgeval = tr * tr / det;

    if (det <= 0.0f) return 0;
    if (fabsf(contr) < scalbnf(thr, 1)) return 0;
    if (edgeval >= (c->edge_limit + 1.0f) * (c->edge_limit + 1.0f) / c->edge_limit) return 0;

    const float sigma_k = powf(2.0f, 1.0f / c->levels);                           /* sift_constants.cu:27 */
    ec->xpos  = xn;
    ec->ypos  = yn;
    ec->lpos  = (int)roundf(sn);
    ec->sigma = c->sigma * powf(sigma_k, sn);
    ec->cell  = (int)(floorf(yn / hdiv) * c->filter_grid_size + floorf(xn / wdiv));
    ec->ignore = 0;
    return 1;
}

typedef struct { osift_iext e; int64_t key; } keyed_iext;
static int cmp_keyed(const void* a, const void* b)
{
    int64_t ka = ((const keyed_iext*)a)->key, kb = ((const keyed_iext*)b)->key;
    return (ka > kb) - (ka < kb);
}

/* Pyramid::find_extrema + find_extrema_in_dog, s_extrema.cu:506-640.
 * The reference's output order is the order of atomicAdd (nondeterministic);
 * the oracle emits (level, y, x) raster order. */
static void find_extrema(osift_result* r, int o)
{
    const osift_config* c = &r->cfg;
    const int W = r->W[o], H = r->H[o];
    dogtex t = { r->dog[o], W, H, r->L - 1 };
    const float thr = osift_peak_threshold(c);
    const int maxlevel = r->L - 1;
    const float wdiv = (float)W / c->filter_grid_size;      /* sift_octave.cu:40-41 */
    const float hdiv = (float)H / c->filter_grid_size;

    keyed_iext* all = NULL; size_t nall = 0, cap = 0;
    #pragma omp parallel
    {
        keyed_iext* loc = NULL; size_t nl = 0, cl = 0;
        #pragma omp for schedule(dynamic, 8) collapse(2)
        for (int level = 1; level <= r->L - 3; level++)
            for (int y = 1; y <= H - 2; y++)
                for (int x = 1; x <= W - 2; x++) {
                    osift_iext ec;
                    if (find_extremum_at(c, &t, x, y, level, thr, maxlevel, wdiv, hdiv, &ec)) {
                        if (nl == cl) { cl = cl ? cl * 2 : 256; loc = (keyed_iext*)realloc(loc, cl * sizeof(*loc)); }
                        loc[nl].e = ec;
                        loc[nl].key = ((int64_t)level * H + y) * W + x;
                        nl++;
                    }
                }
        #pragma omp critical
        {
            if (nall + nl > cap) { cap = (nall + nl) * 2 + 16; all = (keyed_iext*)realloc(all, cap * sizeof(*all)); }
            if (nl) memcpy(all + nall, loc, nl * sizeof(*loc));
            nall += nl;
        }
        free(loc);
    }
    if (nall) qsort(all, nall, sizeof(*all), cmp_keyed);
    int n = (int)nall;
    if (n > c->max_extrema) n = c->max_extrema;              /* :541, :553 */
    r->iext[o] = (osift_iext*)malloc(sizeof(osift_iext) * (size_t)(n > 0 ? n : 1));
    r->iext_off[o] = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) { r->iext[o][i] = all[i].e; r->iext_off[o][i] = i; }
    r->iext_ct[o] = n;
    r->ext_ct[o] = n;
    free(all);
}

/* ------------------------------------------------------------------------- */
/* Grid filter (s_filtergrid.cu:113-325), host arithmetic restated            */
/* ------------------------------------------------------------------------- */
typedef struct { int cell; float scale; int octave; int idx; int seq; } gf_item;
static int gf_cmp_dec(const void* a, const void* b)
{
    const gf_item* l = (const gf_item*)a; const gf_item* r = (const gf_item*)b;
    if (l->cell != r->cell) return l->cell < r->cell ? -1 : 1;
    if (l->scale != r->scale) return l->scale > r->scale ? -1 : 1;
    return (l->seq > r->seq) - (l->seq < r->seq);
}
static int gf_cmp_inc(const void* a, const void* b)
{
    const gf_item* l = (const gf_item*)a; const gf_item* r = (const gf_item*)b;
    if (l->cell != r->cell) return l->cell < r->cell ? -1 : 1;
    if (l->scale != r->scale) return l->scale < r->scale ? -1 : 1;
    return (l->seq > r->seq) - (l->seq < r->seq);
}
static int gf_cmp_cell(const void* a, const void* b)
{
    const gf_item* l = (const gf_item*)a; const gf_item* r = (const gf_item*)b;
    if (l->cell != r->cell) return l->cell < r->cell ? -1 : 1;
    return (l->seq > r->seq) - (l->seq < r->seq);
}
typedef struct { int count; int perm; } gf_cc;
static int gf_cmp_cc(const void* a, const void* b)
{
    const gf_cc* l = (const gf_cc*)a; const gf_cc* r = (const gf_cc*)b;
    if (l->count != r->count) return l->count < r->count ? -1 : 1;
    return (l->perm > r->perm) - (l->perm < r->perm);
}

static int grid_filter(osift_result* r, int ext_total)
{
    const osift_config* c = &r->cfg;
    const int slots = c->filter_grid_size;
    const int n = slots * slots;
    gf_item* it = (gf_item*)malloc(sizeof(gf_item) * (size_t)ext_total);
    int sum = 0;
    for (int o = 0; o < r->num_octaves; o++)
        for (int i = 0; i < r->iext_ct[o]; i++) {
            it[sum].cell = r->iext[o][i].cell;
            it[sum].scale = r->iext[o][i].sigma * powf(2.0f, (float)o);   /* :68 */
            it[sum].octave = o; it[sum].idx = i; it[sum].seq = sum;
            sum++;
        }
    if (c->grid_filter_mode == OSIFT_FILTER_LARGEST_FIRST) qsort(it, sum, sizeof(*it), gf_cmp_dec);
    else if (c->grid_filter_mode == OSIFT_FILTER_SMALLEST_FIRST) qsort(it, sum, sizeof(*it), gf_cmp_inc);
    else qsort(it, sum, sizeof(*it), gf_cmp_cell);

    /* reduce_by_key (:191-194): one count per *distinct* cell value, compacted to the front */
    int* cell_counts = (int*)calloc((size_t)n + 1, sizeof(int));
    int nruns = 0;
    for (int i = 0; i < sum; ) {
        int j = i; while (j < sum && it[j].cell == it[i].cell) j++;
        if (nruns < n) cell_counts[nruns] = j - i;
        nruns++; i = j;
    }
    int* h_counts = (int*)malloc(sizeof(int) * n);
    int* h_offsets = (int*)malloc(sizeof(int) * n);
    int* h_limits = (int*)malloc(sizeof(int) * n);
    memcpy(h_counts, cell_counts, sizeof(int) * n);
    { int acc = 0; for (int i = 0; i < n; i++) { h_offsets[i] = acc; acc += h_counts[i]; h_limits[i] = acc; } }
    gf_cc* cc = (gf_cc*)malloc(sizeof(gf_cc) * n);
    for (int i = 0; i < n; i++) { cc[i].count = h_counts[i]; cc[i].perm = i; }
    qsort(cc, n, sizeof(*cc), gf_cmp_cc);
    int ct = 0;
    { int acc = 0;
      for (int i = 0; i < n; i++) {
          acc += cc[i].count;
          int sumup = cc[i].count * (n - 1 - i) + acc;
          if (sumup > c->filter_max_extrema) ct++;
      } }
    if (ct > 0) {
        int tail = 0; for (int i = n - ct; i < n; i++) tail += cc[i].count;
        float tailaverage = (float)tail / ct;
        int newlimit = (int)ceilf(tailaverage - (ext_total - c->filter_max_extrema) / ct);
        for (int i = 0; i < n; i++) if (cc[i].count > newlimit) cc[i].count = newlimit;
    }
    for (int i = 0; i < n; i++) h_counts[cc[i].perm] = cc[i].count;
    for (int i = 0; i < n; i++) {
        int from = h_offsets[i] + h_counts[i];
        int to = h_limits[i];
        for (int k = from; k < to; k++) r->iext[it[k].octave][it[k].idx].ignore = 1;
    }
    int ret = 0;
    for (int o = 0; o < r->num_octaves; o++) {
        int k = 0;
        for (int i = 0; i < r->iext_ct[o]; i++)
            if (!r->iext[o][i].ignore) r->iext_off[o][k++] = i;
        r->ext_ct[o] = k;
        ret += k;
    }
    free(it); free(cell_counts); free(h_counts); free(h_offsets); free(h_limits); free(cc);
    return ret;
}

/* ------------------------------------------------------------------------- */
/* Orientation (s_orientation.cu:75-259)                                     */
/* ------------------------------------------------------------------------- */
static inline float rdata(const float* plane, int W, int H, int x, int y)
{
    x = clampi(x, 0, W - 1); y = clampi(y, 0, H - 1);
    return plane[(size_t)y * W + x];
}

/* atan2f of the reference's device code (CUDA libdevice, <= 2 ulp): the angle picks the orientation-histogram bin
 * through roundf(36 (theta + pi) / 2 pi), and on exactly diagonal gradients (binary images, symmetric patterns) the
 * bin hangs on theta's last ulp.  All sides -- this file, the HIP kernels' exact path, the shim's stand-in --
 * therefore use the correctly rounded value: atan2 evaluated in double, rounded once. */
#ifdef OSIFT_PLAIN_LIBM
/* sensitivity build only (make plain; tests/test_oracle_cpu.py): glibc's own atan2f, which differs from the single-rounded
 * value in the last ulp for some arguments -- how many orientation bins hang on that ulp is measured, not defined away */
static inline float atan2f_1r(float y, float x) { return atan2f(y, x); }
#else
static inline float atan2f_1r(float y, float x) { return CM_ATAN2F((float)atan2((double)y, (double)x)); }
#endif

/* s_gradiant.h:56-69 (texture variant) */
static inline void get_gradiant(float* grad, float* theta, int x, int y, const float* plane, int W, int H)
{
    float dx = rdata(plane, W, H, x + 1, y) - rdata(plane, W, H, x - 1, y);
    float dy = rdata(plane, W, H, x, y + 1) - rdata(plane, W, H, x, y - 1);
    *grad  = CM_HYPOTF(dx, dy);
    *theta = atan2f_1r(dy, dx);
}

/* common/warp_bitonic_sort.h:17-79, lane-synchronous emulation of Warp32<float>::sort64 */
static void bitonic_shiftit(const float* arr, int* idx, int shift, int direction, int increasing)
{
    int nidx[32];
    for (int l = 0; l < 32; l++) {
        const float my_val = arr[idx[l]];
        const float other_val = arr[idx[l ^ (1 << shift)]];
        const int reverse = (l & (1 << direction)) != 0;
        const int id_less = ((l & (1 << shift)) == 0);
        const int my_more = id_less ? (my_val > other_val) : (my_val < other_val);
        const int must_swap = !(my_more ^ reverse ^ increasing);
        nidx[l] = must_swap ? idx[l ^ (1 << shift)] : idx[l];
    }
    memcpy(idx, nidx, sizeof(nidx));
}
static void bitonic_sort64(const float* arr, int* ix, int* iy)
{
    for (int outer = 0; outer < 5; outer++)
        for (int inner = outer; inner >= 0; inner--) {
            bitonic_shiftit(arr, ix, inner, outer + 1, 0);
            bitonic_shiftit(arr, iy, inner, outer + 1, 1);
        }
    for (int l = 0; l < 32; l++)
        if (arr[ix[l]] < arr[iy[l]]) { int m = iy[l]; iy[l] = ix[l]; ix[l] = m; }
    for (int outer = 0; outer < 5; outer++)
        for (int inner = outer; inner >= 0; inner--) {
            bitonic_shiftit(arr, ix, inner, outer + 1, 0);
            bitonic_shiftit(arr, iy, inner, outer + 1, 0);
        }
}

static void orientation_one(const osift_result* r, int o, const osift_iext* iext, osift_ext* ext)
{
    const int w = r->W[o], h = r->H[o];
    float hist[64], sm_hist[64 + 1], refined_angle[64], yval[64];
    for (int i = 0; i < 64; i++) { hist[i] = 0.0f; sm_hist[i] = 0.0f; }
    sm_hist[64] = 0.0f;

    const float x = iext->xpos, y = iext->ypos;
    const int   level = iext->lpos;
    const float sig = iext->sigma;
    const float* plane = r->data[o] + (size_t)clampi(level, 0, r->L - 1) * w * h;

    const float sigw = ORI_WINFACTOR * sig;
    const int   rad  = (int)roundf((3.0f * sigw));
    const float factor = CM_FDIVIDEF(-0.5f, (sigw * sigw));   /* __fdividef */
    const int   sq_thres = rad * rad;

    int xmin = imax(1,     (int)roundf(x) - rad);
    int xmax = imin(w - 2, (int)roundf(x) + rad);
    int ymin = imax(1,     (int)roundf(y) - rad);
    int ymax = imin(h - 2, (int)roundf(y) + rad);
    int wx = xmax - xmin + 1;
    int hy = ymax - ymin + 1;
    int loops = wx * hy;
    if (wx <= 0 || hy <= 0) loops = 0;

    for (int i = 0; i < loops; i++) {
        int yy = i / wx + ymin;
        int xx = i % wx + xmin;
        float grad, theta;
        get_gradiant(&grad, &theta, xx, yy, plane, w, h);
        float dx = xx - x;
        float dy = yy - y;
        int sq_dist = (int)(dx * dx + dy * dy);
        if (sq_dist <= sq_thres) {
            float weight = grad * CM_EXPF(sq_dist * factor);
            int bidx = (int)roundf(CM_FDIVIDEF((float)ORI_NBINS * (theta + PI_F), PI2_F));   /* __fdividef, s_orientation.cu:148 */
            bidx = (bidx == ORI_NBINS) ? 0 : bidx;
            hist[bidx] += weight;   /* reference: shared-memory float atomicAdd, order unspecified */
        }
    }

    /* WITH_VLFEAT_SMOOTHING :163-180; bins >= 36 are scratch lanes and never feed bins < 36 */
    for (int it = 0; it < 3; it++) {
        for (int bin = 0; bin < ORI_NBINS; bin++) {
            const int prev = (bin == 0) ? ORI_NBINS - 1 : bin - 1;
            const int next = (bin == ORI_NBINS - 1) ? 0 : bin + 1;
            sm_hist[bin] = (hist[prev] + hist[bin] + hist[next]) / 3.0f;
        }
        for (int bin = 0; bin < ORI_NBINS; bin++) {
            const int prev = (bin == 0) ? ORI_NBINS - 1 : bin - 1;
            const int next = (bin == ORI_NBINS - 1) ? 0 : bin + 1;
            hist[bin] = (sm_hist[prev] + sm_hist[bin] + sm_hist[next]) / 3.0f;
        }
    }
    for (int bin = 0; bin < ORI_NBINS; bin++) sm_hist[bin] = hist[bin];

    for (int bin = 0; bin < 64; bin++) {
        int predicate = 0;
        float num = 0.0f, denB = 1.0f;
        int prev = 0;
        if (bin < ORI_NBINS) {
            prev = bin == 0 ? ORI_NBINS - 1 : bin - 1;
            const int next = bin == ORI_NBINS - 1 ? 0 : bin + 1;
            predicate = (sm_hist[bin] > fmaxf(sm_hist[prev], sm_hist[next]));
            if (predicate) {
                num  = 3.0f * sm_hist[prev] - 4.0f * sm_hist[bin] + 1.0f * sm_hist[next];
                denB = 2.0f * (sm_hist[prev] - 2.0f * sm_hist[bin] + sm_hist[next]);
            }
        }
        const float newbin = CM_FDIVIDEF(num, denB);         /* __fdividef */
        predicate = (predicate && newbin >= 0.0f && newbin <= 2.0f);
        refined_angle[bin] = predicate ? prev + newbin : -1;
        yval[bin] = predicate ? -(num * num) / (4.0f * denB) + sm_hist[prev] : -INFINITY;
    }

    int ix[32], iy[32];
    for (int l = 0; l < 32; l++) { ix[l] = l; iy[l] = l + 32; }
    bitonic_sort64(yval, ix, iy);

    const float yval_ref = 0.8f * yval[ix[0]];
    int angles = 0;
    for (int l = 0; l < OSIFT_ORI_MAX; l++) {
        const float best_val = yval[ix[l]];
        const int valid = (best_val >= yval_ref);
        ext->orientation[l] = 0.0f;
        if (valid) {
            float chosen_bin = refined_angle[ix[l]];
            if (chosen_bin >= ORI_NBINS) chosen_bin -= ORI_NBINS;
            float th = fmaf(PI2_F * chosen_bin, 1.0f / ORI_NBINS, -PI_F);
            ext->orientation[l] = th;
            angles++;
        }
    }
    ext->xpos = iext->xpos;
    ext->ypos = iext->ypos;
    ext->lpos = iext->lpos;
    ext->sigma = iext->sigma;
    ext->octave = o;
    ext->num_ori = angles;
    ext->idx_ori = 0;
}

/* ------------------------------------------------------------------------- */
/* Descriptor, "loop" mode (s_desc_loop.cu:19-139)                            */
/* ------------------------------------------------------------------------- */
static inline __attribute__((always_inline)) void fp_dep3(float* a, float* b, float* c)
{ __asm__ volatile("" : "+x"(*a), "+x"(*b), "+x"(*c)); }
static inline __attribute__((always_inline)) void fp_dep1(float* a)
{ __asm__ volatile("" : "+x"(*a)); }

/* th*M_4RPI rounded up, then the two round-up FMAs (:108-119) */
static __attribute__((noinline)) void desc_bin_accum(float th, float wgt, float* dpt)
{
    float k = M_4RPI_F;
    unsigned int csr = _mm_getcsr();
    _mm_setcsr((csr & ~_MM_ROUND_MASK) | _MM_ROUND_UP);
    fp_dep3(&th, &k, &wgt);
    float tth = th * k;
    fp_dep1(&tth);
    _mm_setcsr(csr);
    fp_dep1(&tth);
    const int   fo0  = (int)floorf(tth);
    const float do0  = tth - fo0;
    float wgt1 = 1.0f - do0;
    float wgt2 = do0;
    int fo = fo0 % DESC_BINS;
    float a0 = dpt[fo], a1 = dpt[fo + 1];
    fp_dep3(&wgt1, &wgt2, &wgt);
    _mm_setcsr((csr & ~_MM_ROUND_MASK) | _MM_ROUND_UP);
    fp_dep3(&wgt1, &wgt, &a0);
    fp_dep3(&wgt2, &wgt, &a1);
    float r0 = fmaf(wgt1, wgt, a0);
    float r1 = fmaf(wgt2, wgt, a1);
    fp_dep1(&r0); fp_dep1(&r1);
    _mm_setcsr(csr);
    fp_dep1(&r0); fp_dep1(&r1);
    dpt[fo] = r0;
    dpt[fo + 1] = r1;
}

static void descriptor_one(const osift_result* r, const osift_ext* ext, float ang, float* features)
{
    const int o = ext->octave;
    const int width = r->W[o], height = r->H[o];
    const float x = ext->xpos, y = ext->ypos;
    const int level = ext->lpos;
    const float sig = ext->sigma;
    const float SBP = fabsf(DESC_MAGNIFY * sig);
    const float* plane = r->data[o] + (size_t)clampi(level, 0, r->L - 1) * width * height;
    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (SBP == 0) return;

    const float cos_t = CM_FAST_COS(cosf(ang));      /* __sincosf */
    const float sin_t = CM_FAST_SIN(sinf(ang));
    const float csbp  = cos_t * SBP;
    const float ssbp  = sin_t * SBP;
    const float crsbp = cos_t / SBP;
    const float srsbp = sin_t / SBP;

    for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++) {
        const int tile = (((iy << 2) + ix) << 3);
        const float offx = ix - 1.5f, offy = iy - 1.5f;
        const float ptx = fmaf(csbp, offx, fmaf(-ssbp, offy, x));
        const float pty = fmaf(csbp, offy, fmaf( ssbp, offx, y));
        const float bsz = fabsf(csbp) + fabsf(ssbp);
        const int xmin = imax(1,          (int)floorf(ptx - bsz));
        const int ymin = imax(1,          (int)floorf(pty - bsz));
        const int xmax = imin(width - 2,  (int)floorf(ptx + bsz));
        const int ymax = imin(height - 2, (int)floorf(pty + bsz));
        const int wx = xmax - xmin + 1;
        const int hy = ymax - ymin + 1;
        int loops = wx * hy;
        if (wx <= 0 || hy <= 0) loops = 0;

        float dpt[32][9];
        memset(dpt, 0, sizeof(dpt));
        /* lane l visits i = l, l+32, ... (:79) */
        for (int i = 0; i < loops; i++) {
            const int lane = i & 31;
            const int ii = i / wx + ymin;
            const int jj = i % wx + xmin;
            const float dx = jj - ptx, dy = ii - pty;
            const float nx = fmaf(crsbp, dx,  srsbp * dy);
            const float ny = fmaf(crsbp, dy, -srsbp * dx);
            const float nnx = fabsf(nx), nny = fabsf(ny);
            if (nnx < 1.0f && nny < 1.0f) {
                float mod, th;
                get_gradiant(&mod, &th, jj, ii, plane, width, height);
                const float dnx = nx + offx, dny = ny + offy;
                const float ww = CM_FAST_EXPF(-scalbnf(dnx * dnx + dny * dny, -3));   /* __expf */
                const float wx_ = 1.0f - nnx, wy_ = 1.0f - nny;
                const float wgt = ww * wx_ * wy_ * mod;
                th -= ang;
                th += (th <  0.0f  ? PI2_F : 0.0f);
                th -= (th >= PI2_F ? PI2_F : 0.0f);
                desc_bin_accum(th, wgt, dpt[lane]);
            }
        }
        for (int l = 0; l < 32; l++) dpt[l][0] += dpt[l][8];
        /* shuffle_down tree 16,8,4,2,1 (:127-134); lanes >= 32-delta read their own value */
        for (int b = 0; b < 8; b++) {
            float v[32];
            for (int l = 0; l < 32; l++) v[l] = dpt[l][b];
            for (int delta = 16; delta >= 1; delta >>= 1) {
                float nv[32];
                for (int l = 0; l < 32; l++) nv[l] = v[l] + (l + delta < 32 ? v[l + delta] : v[l]);
                memcpy(v, nv, sizeof(v));
            }
            features[tile + b] = v[0];
        }
    }
}

/* s_desc_norm_rs.h:42-77 / s_desc_norm_l2.h:86-135 (non-normf branch) */
static float warp_sum32(const float* lane_vals)
{
    float v[32];
    memcpy(v, lane_vals, sizeof(v));
    for (int delta = 16; delta >= 1; delta >>= 1) {
        float nv[32];
        for (int l = 0; l < 32; l++) nv[l] = v[l] + (l + delta < 32 ? v[l + delta] : v[l]);
        memcpy(v, nv, sizeof(v));
    }
    return v[0];
}

static void normalize_desc(const osift_config* c, float* d)
{
    float lane[32];
    if (c->norm_mode == OSIFT_NORM_ROOTSIFT) {
        for (int l = 0; l < 32; l++) lane[l] = d[4 * l] + d[4 * l + 1] + d[4 * l + 2] + d[4 * l + 3];
        const float sum = warp_sum32(lane);
        for (int i = 0; i < 128; i++) d[i] = scalbnf(sqrtf(CM_FDIVIDEF(d[i], sum)), c->norm_multi);   /* __fsqrt_rn(__fdividef) */
    } else {
        for (int l = 0; l < 32; l++)
            lane[l] = d[4 * l] * d[4 * l] + d[4 * l + 1] * d[4 * l + 1]
                    + d[4 * l + 2] * d[4 * l + 2] + d[4 * l + 3] * d[4 * l + 3];
        float norm = sqrtf(warp_sum32(lane));
        for (int i = 0; i < 128; i++) d[i] = fminf(d[i], 0.2f * norm);
        for (int l = 0; l < 32; l++)
            lane[l] = d[4 * l] * d[4 * l] + d[4 * l + 1] * d[4 * l + 1]
                    + d[4 * l + 2] * d[4 * l + 2] + d[4 * l + 3] * d[4 * l + 3];
        norm = warp_sum32(lane);
        norm = CM_FRSQRT_RN(norm);           /* __frsqrt_rn */
        norm = scalbnf(norm, c->norm_multi);
        for (int i = 0; i < 128; i++) d[i] = d[i] * norm;
    }
}

/* ------------------------------------------------------------------------- */
/* Alternative descriptor modes                                              */
/* ------------------------------------------------------------------------- */

/* get_gradiant on a linear texture with a rotated stencil (s_gradiant.h:72-88): the gradient in the
 * keypoint's frame, so the angle needs no "- ang" afterwards */
static inline void get_gradiant_rot(float* grad, float* theta, float x, float y, float cos_t, float sin_t,
                                    const float* plane, int W, int H)
{
    const float dx = plane_linear(plane, W, H, x + cos_t, y + sin_t) - plane_linear(plane, W, H, x - cos_t, y - sin_t);
    const float dy = plane_linear(plane, W, H, x - sin_t, y + cos_t) - plane_linear(plane, W, H, x + sin_t, y - cos_t);
    *grad = CM_HYPOTF(dx, dy);
    *theta = atan2f_1r(dy, dx);
}

/* point-texture gradient at float coordinates that are integers after rounding (s_gradiant.h:56-69 called
 * with float arguments from s_desc_grid.cu:78: the int parameters truncate them) */
static inline void get_gradiant_pt(float* grad, float* theta, int x, int y, const float* plane, int W, int H)
{
    const float dx = rdata(plane, W, H, x + 1, y) - rdata(plane, W, H, x - 1, y);
    const float dy = rdata(plane, W, H, x, y + 1) - rdata(plane, W, H, x, y - 1);
    *grad = CM_HYPOTF(dx, dy);
    *theta = atan2f_1r(dy, dx);
}

/* the shuffle_down tree of a group of `width` lanes as lane 0 sees it (lanes beyond the group read their own value) */
static float group_sum(const float* lane_vals, int width)
{
    float v[32];
    memcpy(v, lane_vals, sizeof(float) * (size_t)width);
    for (int delta = width / 2; delta >= 1; delta >>= 1) {
        float nv[32];
        for (int l = 0; l < width; l++) nv[l] = v[l] + (l + delta < width ? v[l + delta] : v[l]);
        memcpy(v, nv, sizeof(float) * (size_t)width);
    }
    return v[0];
}

/* ext_desc_iloop (s_desc_iloop.cu:19-130): per tile a 32 x 32 grid of sample points in the tile's own
 * 2x2-SBP support, gradients by bilinear interpolation in the rotated frame */
static void descriptor_iloop(const osift_result* r, const osift_ext* ext, float ang, float* features)
{
    const int o = ext->octave;
    const int W = r->W[o], H = r->H[o];
    const float x = ext->xpos, y = ext->ypos;
    const float SBP = fabsf(DESC_MAGNIFY * ext->sigma);
    const float* plane = r->data[o] + (size_t)clampi(ext->lpos, 0, r->L - 1) * W * H;
    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (SBP == 0) return;
    const float cos_t = CM_FAST_COS(cosf(ang)), sin_t = CM_FAST_SIN(sinf(ang));          /* __sincosf */
    const float csbp = cos_t * SBP, ssbp = sin_t * SBP;
    for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++) {
        const int tile = (((iy << 2) + ix) << 3);
        const float offx = ix - 1.5f, offy = iy - 1.5f;
        const float ptx = fmaf(csbp, offx, -ssbp * offy);
        const float pty = fmaf(csbp, offy,  ssbp * offx);
        const float bsz = fabsf(cos_t) + fabsf(sin_t);
        float dpt[32][9];
        memset(dpt, 0, sizeof(dpt));
        for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            const float dx = (-bsz + j * bsz / 16.0f);
            const float dy = (-bsz + i * bsz / 16.0f);
            const float nx = fmaf(cos_t, dx,  sin_t * dy);
            const float ny = fmaf(cos_t, dy, -sin_t * dx);
            const float nnx = fabsf(nx), nny = fabsf(ny);
            if (nnx < 1.0f && nny < 1.0f) {
                const float jj = x + ptx + dx * SBP;
                const float ii = y + pty + dy * SBP;
                float mod, th;
                get_gradiant_rot(&mod, &th, jj, ii, cos_t, sin_t, plane, W, H);
                const float dnx = nx + offx, dny = ny + offy;
                const float ww = CM_FAST_EXPF(-scalbnf(dnx * dnx + dny * dny, -3));   /* __expf */
                const float wgt = ww * (1.0f - nnx) * (1.0f - nny) * mod;
                th += (th <  0.0f  ? PI2_F : 0.0f);
                th -= (th >= PI2_F ? PI2_F : 0.0f);
                desc_bin_accum(th, wgt, dpt[j]);
            }
        }
        for (int l = 0; l < 32; l++) dpt[l][0] += dpt[l][8];
        for (int b = 0; b < 8; b++) {
            float v[32];
            for (int l = 0; l < 32; l++) v[l] = dpt[l][b];
            features[tile + b] = group_sum(v, 32);
        }
    }
}

/* ext_desc_grid (s_desc_grid.cu:19-124): per tile a 16 x 16 grid of sample points snapped to pixel centres */
static void descriptor_grid(const osift_result* r, const osift_ext* ext, float ang, float* features)
{
    const int o = ext->octave;
    const int W = r->W[o], H = r->H[o];
    const float x = ext->xpos, y = ext->ypos;
    const float SBP = fabsf(DESC_MAGNIFY * ext->sigma);
    const float* plane = r->data[o] + (size_t)clampi(ext->lpos, 0, r->L - 1) * W * H;
    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (SBP == 0) return;
    /* __sincosf in the reference.  The sample positions of this mode go through (int)(pt + (round(pt + pix) - pt)),
     * which flips to the neighbouring pixel on the last bit of pt, i.e. of sin / cos: all three sides (this file, the
     * HIP kernel, the shim's __sincosf stand-in) therefore evaluate them in double and round ONCE, so that equal
     * orientation bits give equal sample positions. */
    const float cos_t = CM_FAST_COS((float)cos((double)ang)), sin_t = CM_FAST_SIN((float)sin((double)ang));
    const float csbp = cos_t * SBP, ssbp = sin_t * SBP;
    for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++) {
        const int tile = (((iy << 2) + ix) << 3);
        const float offx = ix - 1.5f, offy = iy - 1.5f;
        const float ptx = fmaf(csbp, offx, fmaf(-ssbp, offy, x));
        const float pty = fmaf(csbp, offy, fmaf( ssbp, offx, y));
        const float ldx = -cos_t + sin_t, ldy = -cos_t - sin_t;           /* lft_dn */
        const float rsx = cos_t / 8.0f, rsy = sin_t / 8.0f;               /* rgt_stp */
        const float usx = -sin_t / 8.0f, usy = cos_t / 8.0f;              /* up__stp */
        float dpt[16][9];
        memset(dpt, 0, sizeof(dpt));
        for (int xd = 0; xd < 16; xd++)
        for (int yd = 0; yd < 16; yd++) {
            /* float2 pixo = lft_dn + (xd+0.5f) * rgt_stp + (yd+0.5f) * up__stp; pix = pixo * SBP;
             * pix = round(pt + pix) - pt  -- as the device compiler contracts it (nvcc -fmad=true) */
            float pox = fmaf(yd + 0.5f, usx, fmaf(xd + 0.5f, rsx, ldx));
            float poy = fmaf(yd + 0.5f, usy, fmaf(xd + 0.5f, rsy, ldy));
            float pix_x = roundf(fmaf(pox, SBP, ptx)) - ptx;
            float pix_y = roundf(fmaf(poy, SBP, pty)) - pty;
            pox = pix_x / SBP; poy = pix_y / SBP;
            float mod, th;
            get_gradiant_pt(&mod, &th, (int)(ptx + pix_x), (int)(pty + pix_y), plane, W, H);
            const float npx = fmaf(cos_t, pox,  sin_t * poy);
            const float npy = fmaf(cos_t, poy, -sin_t * pox);
            const float dnx = npx + offx, dny = npy + offy;
            const float ww = CM_EXPF(-scalbnf(dnx * dnx + dny * dny, -3));    /* expf, s_desc_grid.cu:82 */
            const float wx = 1.0f - fabsf(npx), wy = 1.0f - fabsf(npy);
            if (wx < 0.0f || wy < 0.0f) continue;
            const float wgt = ww * wx * wy * mod;
            th -= ang;
            th += (th <  0.0f  ? PI2_F : 0.0f);
            th -= (th >= PI2_F ? PI2_F : 0.0f);
            desc_bin_accum(th, wgt, dpt[xd]);
        }
        for (int l = 0; l < 16; l++) dpt[l][0] += dpt[l][8];
        for (int b = 0; b < 8; b++) {
            float v[16];
            for (int l = 0; l < 16; l++) v[l] = dpt[l][b];
            features[tile + b] = group_sum(v, 16);
        }
    }
}

/* sift_constants.cu:34-47: desc_gauss[40][40], desc_tile[16] */
static float g_desc_gauss[40][40];
static float g_desc_tile[16];
static int   g_desc_tables_done = 0;
static void init_desc_tables(void)
{
    if (g_desc_tables_done) return;
    const float dn_step = 1.0f / 8.0f;
    const float dn_base = 0.5f * dn_step - 20.0f * dn_step;
    for (int yy = 0; yy < 40; yy++)
        for (int xx = 0; xx < 40; xx++) {
            const float dnx = dn_base + xx * dn_step;
            const float dny = dn_base + yy * dn_step;
            g_desc_gauss[yy][xx] = expf(-scalbnf(dnx * dnx + dny * dny, -3));
        }
    for (int i = 0; i < 16; i++) {
        const float nx = -1.0f + 1.0f / 16.0f + i * 1.0f / 8.0f;
        g_desc_tile[i] = 1.0f - fabsf(nx);
    }
    g_desc_tables_done = 1;
}

/* th * M_4RPI rounded up (__fmul_ru, s_desc_igrid.cu:48) */
static __attribute__((noinline)) float mul_ru(float a, float b)
{
    unsigned int csr = _mm_getcsr();
    fp_dep1(&a); fp_dep1(&b);
    _mm_setcsr((csr & ~_MM_ROUND_MASK) | _MM_ROUND_UP);
    fp_dep1(&a); fp_dep1(&b);
    float t = a * b;
    fp_dep1(&t);
    _mm_setcsr(csr);
    fp_dep1(&t);
    return t;
}

/* ext_desc_igrid (s_desc_igrid.cu:19-72): per tile 16 x 16 fixed sample points in the keypoint frame, bilinear
 * gradients, tabulated Gaussian and tile weights; xor-tree reduction over 16 lanes */
static void descriptor_igrid(const osift_result* r, const osift_ext* ext, float ang, float* features)
{
    const int o = ext->octave;
    const int W = r->W[o], H = r->H[o];
    const float x = ext->xpos, y = ext->ypos;
    const float* plane = r->data[o] + (size_t)clampi(ext->lpos, 0, r->L - 1) * W * H;
    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (ext->sigma == 0) return;
    const float SBP = fabsf(DESC_MAGNIFY * ext->sigma);
    const float cos_t = CM_FAST_COS(cosf(ang)), sin_t = CM_FAST_SIN(sinf(ang));
    init_desc_tables();
    for (int iy = 0; iy < 4; iy++)
    for (int ix = 0; ix < 4; ix++) {
        const int tile = (((iy << 2) + ix) << 3);
        float dpt[16][8];
        memset(dpt, 0, sizeof(dpt));
        for (int xd = 0; xd < 16; xd++)
        for (int yd = 0; yd < 16; yd++) {
            const float stepx = ix - 2.5f + 1.0f / 16.0f + xd / 8.0f;
            const float stepy = iy - 2.5f + 1.0f / 16.0f + yd / 8.0f;
            const float ptx = fmaf(cos_t, stepx, -sin_t * stepy);
            const float pty = fmaf(cos_t, stepy,  sin_t * stepx);
            float mod, th;
            get_gradiant_rot(&mod, &th, fmaf(ptx, SBP, x), fmaf(pty, SBP, y), cos_t, sin_t, plane, W, H);
            th += (th <  0.0f  ? PI2_F : 0.0f);
            th -= (th >= PI2_F ? PI2_F : 0.0f);
            const float ww = g_desc_gauss[iy * 8 + yd][ix * 8 + xd];
            const float wgt = ww * g_desc_tile[xd] * g_desc_tile[yd] * mod;
            const float tth = mul_ru(th, M_4RPI_F);
            const int   fo  = (int)floorf(tth);
            const float do0 = tth - fo;
            const int fo1 = (fo + 1) & 7, fo0 = fo & 7;
            dpt[xd][fo1] = fmaf(wgt, do0, dpt[xd][fo1]);
            dpt[xd][fo0] = fmaf(wgt, 1.0f - do0, dpt[xd][fo0]);
        }
        for (int b = 0; b < 8; b++) {
            /* shuffle_xor 1, 2, 4, 8 over 16 lanes */
            float v[16];
            for (int l = 0; l < 16; l++) v[l] = dpt[l][b];
            for (int m = 1; m <= 8; m <<= 1) {
                float nv[16];
                for (int l = 0; l < 16; l++) nv[l] = v[l] + v[l ^ m];
                memcpy(v, nv, sizeof(v));
            }
            features[tile + b] = v[b];       /* lane threadIdx.x = b writes features[tile + b] */
        }
    }
}

/* ext_desc_notile (s_desc_notile.cu:31-95): one 40 x 40 grid of sample points for the whole window; the thread
 * (tx 0..31, ty 0..3) owns columns tx and tx+8 and rows 8 ty .. 8 ty + 15; 8-lane shuffle tree */
static void descriptor_notile(const osift_result* r, const osift_ext* ext, float ang, float* features)
{
    const int o = ext->octave;
    const int W = r->W[o], H = r->H[o];
    const float x = ext->xpos, y = ext->ypos;
    const float* plane = r->data[o] + (size_t)clampi(ext->lpos, 0, r->L - 1) * W * H;
    for (int i = 0; i < 128; i++) features[i] = 0.0f;
    if (ext->sigma == 0) return;
    const float SBP = fabsf(DESC_MAGNIFY * ext->sigma);
    const float cos_t = CM_FAST_COS(cosf(ang)), sin_t = CM_FAST_SIN(sinf(ang));
    const float stepbase = -2.5f + 1.0f / 16.0f;
    init_desc_tables();
    for (int out_y = 0; out_y < 4; out_y++) {
        float dpt[32][8];
        memset(dpt, 0, sizeof(dpt));
        for (int tx = 0; tx < 32; tx++) {
            const int in_x = tx & 7;
            for (int xoff = 0; xoff < 2; xoff++) {
                const int xd = (xoff << 3) + in_x;
                const int newx = (xoff << 3) + tx;
                for (int yoff = 0; yoff < 2; yoff++)
                for (int in_y = 0; in_y < 8; in_y++) {
                    const int yd = (yoff << 3) + in_y;
                    const int newy = (out_y << 3) + yd;
                    const float wgt = g_desc_tile[xd] * g_desc_tile[yd];
                    const float stepx = stepbase + scalbnf((float)newx, -3);
                    const float stepy = stepbase + scalbnf((float)newy, -3);
                    const float ptx = fmaf(cos_t, stepx, -sin_t * stepy);
                    const float pty = fmaf(cos_t, stepy,  sin_t * stepx);
                    float mod, th;
                    get_gradiant_rot(&mod, &th, fmaf(ptx, SBP, x), fmaf(pty, SBP, y), cos_t, sin_t, plane, W, H);
                    th += (th < 0.0f ? PI2_F : 0.0f);
                    const float tth = th * M_4RPI_F;
                    const int   fo  = (int)floorf(th * M_4RPI_F);
                    const float do0 = tth - fo;
                    const int fo0 = fo & 7, fo1 = (fo0 + 1) & 7;
                    const float ww = g_desc_gauss[newy][newx] * mod;
                    const float ow0 = (1.0f - do0) * ww, ow1 = do0 * ww;
                    dpt[tx][fo0] = fmaf(wgt, ow0, dpt[tx][fo0]);
                    dpt[tx][fo1] = fmaf(wgt, ow1, dpt[tx][fo1]);
                }
            }
        }
        for (int g = 0; g < 4; g++)
            for (int b = 0; b < 8; b++) {
                float v[8];
                for (int l = 0; l < 8; l++) v[l] = dpt[g * 8 + l][b];
                /* features[out_y * 32 + tx] = dpt[in_x] after the tree has moved lane 0's sum to all 8 lanes */
                features[out_y * 32 + g * 8 + b] = group_sum(v, 8);
            }
    }
}

/* ------------------------------------------------------------------------- */
/* Driver (popsift.cpp:109-144, sift_pyramid.cu:108-134,227-240,250-322)      */
/* ------------------------------------------------------------------------- */
static osift_result* run_impl(const osift_config* cin, const void* img, int w, int h, int is_float, int full)
{
#ifdef _OPENMP
    if (g_threads > 0) omp_set_num_threads(g_threads);
#endif
    osift_result* r = (osift_result*)calloc(1, sizeof(*r));
    r->cfg = *cin;
    osift_config* c = &r->cfg;
    c->levels = imax(2, c->levels);                              /* popsift.cpp:86 */
    if (osift_gauss_tables(c, &r->tab) != 0) { free(r); return NULL; }

    /* popsift.cpp:109-126 */
    const float scaleFactor = 1.0f / powf(2.0f, -c->upscale_factor);
    if (c->octaves < 0) {
        int oct = imax((int)(floorf(logf((float)imin(w, h)) / logf(2.0f)) - 3.0f + scaleFactor), 1);
        c->octaves = oct;
    }
    int ow = (int)ceilf(w * scaleFactor);
    int oh = (int)ceilf(h * scaleFactor);
    r->num_octaves = imin(c->octaves, OSIFT_MAX_OCTAVES);
    r->L = c->levels + 3;
    for (int o = 0; o < r->num_octaves; o++) {
        r->W[o] = ow; r->H[o] = oh;
        r->data[o] = (float*)malloc(sizeof(float) * (size_t)ow * oh * r->L);
        r->dog[o]  = (float*)malloc(sizeof(float) * (size_t)ow * oh * (r->L - 1));
        ow = (int)ceilf(ow / 2.0f);                              /* sift_pyramid.cu:132-133 */
        oh = (int)ceilf(oh / 2.0f);
    }

    tex_in t = { img, w, h, is_float };
    if (build_pyramid(r, &t) != 0) { osift_free(r); return NULL; }
    for (int o = 0; o < r->num_octaves; o++) make_dog(r, o);
    if (!full) return r;

    /* step2: find_extrema, orientation, descriptors (sift_pyramid.cu:233-240) */
    int ext_total = 0;
    for (int o = 0; o < r->num_octaves; o++) { find_extrema(r, o); ext_total += r->iext_ct[o]; }

    /* s_orientation.cu:378-383 */
    if (c->filter_max_extrema > 0 && (int)(c->filter_max_extrema * 1.1) < ext_total)
        ext_total = grid_filter(r, ext_total);

    r->ext_total = ext_total;
    r->ext = (osift_ext*)calloc((size_t)(ext_total > 0 ? ext_total : 1), sizeof(osift_ext));
    {
        int base = 0;
        for (int o = 0; o < r->num_octaves; o++) {
            const int n = r->ext_ct[o];
            #pragma omp parallel for schedule(dynamic, 16)
            for (int i = 0; i < n; i++)
                orientation_one(r, o, &r->iext[o][r->iext_off[o][i]], &r->ext[base + i]);
            base += n;
        }
    }
    /* ori_prefix_sum, s_orientation.cu:320-362 */
    const int max_orientations = c->max_extrema + c->max_extrema / 4;
    int ori_allocated = imax(2 * c->max_extrema, max_orientations);            /* sift_pyramid.cu:154-159 */
    /* Pyramid::reallocExtrema (sift_pyramid.cu:179-209): more extrema than the initial max_extrema entries =>
     * extrema buffers grow to the count rounded up to 1024 and the descriptor buffers to twice that.
     * Orientations beyond the capacity are dropped here (the reference would write past its buffer). */
    if (ext_total > c->max_extrema) ori_allocated = imax(ori_allocated, 2 * ((ext_total + 1024) & ~1023));
    int total_ori = 0;
    for (int i = 0; i < ext_total; i++) { r->ext[i].idx_ori = total_ori; total_ori += r->ext[i].num_ori; }
    if (total_ori > ori_allocated) total_ori = ori_allocated;
    r->ori_total = total_ori;
    r->feat_to_ext = (int*)malloc(sizeof(int) * (size_t)(total_ori > 0 ? total_ori : 1));
    for (int i = 0; i < ext_total; i++)
        for (int k = 0; k < r->ext[i].num_ori; k++)
            if (r->ext[i].idx_ori + k < total_ori) r->feat_to_ext[r->ext[i].idx_ori + k] = i;

    r->desc = (float*)calloc((size_t)(total_ori > 0 ? total_ori : 1) * 128, sizeof(float));
    init_desc_tables();
    #pragma omp parallel for schedule(dynamic, 8)
    for (int j = 0; j < total_ori; j++) {
        const osift_ext* e = &r->ext[r->feat_to_ext[j]];
        const int ori_num = j - e->idx_ori;
        float* dj = r->desc + (size_t)j * 128;
        switch (c->desc_mode) {                                  /* sift_desc.cu:66-83 */
        case OSIFT_DESC_ILOOP:  descriptor_iloop(r, e, e->orientation[ori_num], dj); break;
        case OSIFT_DESC_GRID:   descriptor_grid(r, e, e->orientation[ori_num], dj); break;
        case OSIFT_DESC_IGRID:  descriptor_igrid(r, e, e->orientation[ori_num], dj); break;
        case OSIFT_DESC_NOTILE: descriptor_notile(r, e, e->orientation[ori_num], dj); break;
        default:                descriptor_one(r, e, e->orientation[ori_num], dj); break;
        }
        normalize_desc(c, dj);
    }

    /* prep_features, sift_pyramid.cu:250-280 */
    const int up_fac = (int)c->upscale_factor;
    r->feat = (osift_feature*)calloc((size_t)(ext_total > 0 ? ext_total : 1), sizeof(osift_feature));
    for (int i = 0; i < ext_total; i++) {
        const osift_ext* e = &r->ext[i];
        osift_feature* f = &r->feat[i];
        const float s = powf(2.0f, (float)(e->octave - up_fac));
        f->debug_octave = e->octave;
        f->xpos = e->xpos * s;
        f->ypos = e->ypos * s;
        f->sigma = e->sigma * s;
        f->num_ori = e->num_ori;
        int k;
        for (k = 0; k < e->num_ori; k++) {
            f->desc_idx[k] = (e->idx_ori + k < total_ori) ? e->idx_ori + k : -1;
            f->orientation[k] = e->orientation[k];
        }
        for (; k < OSIFT_ORI_MAX; k++) { f->desc_idx[k] = -1; f->orientation[k] = 0; }
    }
    return r;
}

osift_result* osift_run(const osift_config* c, const void* img, int w, int h, int is_float)
{ return run_impl(c, img, w, h, is_float, 1); }

/* Descriptor stage alone (sift_desc.cu:66-83 + normalisation) on r's pyramid for CALLER-SUPPLIED oriented extrema
 * (e.g. the ones a device run produced, bit for bit): descriptor k of extremum i goes to out + (idx_ori + k) * 128.
 * Lets a test compare the descriptor arithmetic under identical keypoint and orientation bits. */
int osift_describe(const osift_result* r, const osift_ext* ext, int n_ext, int n_desc, float* out)
{
    const osift_config* c = &r->cfg;
    init_desc_tables();
    #pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n_ext; i++) {
        const osift_ext* e = &ext[i];
        if (e->octave < 0 || e->octave >= r->num_octaves) continue;
        for (int k = 0; k < e->num_ori && k < OSIFT_ORI_MAX; k++) {
            const int j = e->idx_ori + k;
            if (j < 0 || j >= n_desc) continue;
            float* dj = out + (size_t)j * 128;
            switch (c->desc_mode) {
            case OSIFT_DESC_ILOOP:  descriptor_iloop(r, e, e->orientation[k], dj); break;
            case OSIFT_DESC_GRID:   descriptor_grid(r, e, e->orientation[k], dj); break;
            case OSIFT_DESC_IGRID:  descriptor_igrid(r, e, e->orientation[k], dj); break;
            case OSIFT_DESC_NOTILE: descriptor_notile(r, e, e->orientation[k], dj); break;
            default:                descriptor_one(r, e, e->orientation[k], dj); break;
            }
            normalize_desc(c, dj);
        }
    }
    return 0;
}
osift_result* osift_run_pyramid(const osift_config* c, const void* img, int w, int h, int is_float)
{ return run_impl(c, img, w, h, is_float, 0); }

/* ------------------------------------------------------------------------- */
/* 2-NN matcher (features.cu:160-225)                                         */
/* ------------------------------------------------------------------------- */
static float match_l2(const float* l, const float* r)
{
    float p[32];
    for (int t = 0; t < 32; t++) {                 /* l2_in_t0, one CUDA thread each (:160-176) */
        const float x = l[4 * t + 0] - r[4 * t + 0];
        const float y = l[4 * t + 1] - r[4 * t + 1];
        const float z = l[4 * t + 2] - r[4 * t + 2];
        const float w = l[4 * t + 3] - r[4 * t + 3];
        float s = y * y;
        s = fmaf(x, x, s);
        s = fmaf(z, z, s);
        s = fmaf(w, w, s);
        p[t] = s;
    }
    /* shuffle_down 16, 8, 4, 2, 1 (:177-181): value of lane 0 */
    for (int off = 16; off >= 1; off >>= 1)
        for (int t = 0; t < off; t++) p[t] = p[t] + p[t + off];
    return p[0];
}

void osift_match(const float* l, int nl, const float* r, int nr, int* out3, float* dist2)
{
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < nl; i++) {
        float v1 = INFINITY, v2 = INFINITY;        /* compute_distance, :183-223 */
        int i1 = 0, i2 = 0;
        for (int j = 0; j < nr; j++) {
            const float res = match_l2(l + (size_t)i * 128, r + (size_t)j * 128);
            if (res < v1) { v2 = v1; i2 = i1; v1 = res; i1 = j; }
            else if (res < v2) { v2 = res; i2 = j; }
        }
        out3[3 * i + 0] = i1; out3[3 * i + 1] = i2; out3[3 * i + 2] = (v1 / v2 < 0.8f) ? 1 : 0;
        if (dist2) { dist2[2 * i + 0] = v1; dist2[2 * i + 1] = v2; }
    }
}
