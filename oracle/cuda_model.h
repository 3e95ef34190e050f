/*
 * oracle/cuda_model.h -- error model of CUDA's device math functions, for SENSITIVITY builds only.
 *
 * TEST INFRASTRUCTURE.  The oracle (and the CPU build of the reference's own sources, oracle/ref_shim) restate CUDA's
 * approximate device functions with correctly rounded stand-ins: the bits a real NVIDIA run produces are defined by
 * approximations whose error the CUDA C++ Programming Guide only BOUNDS ("Mathematical Functions" appendix):
 *     expf 2 ulp, hypotf 3 ulp, atan2f 3 ulp (libdevice, full range);
 *     __fdividef(x, y) 2 ulp for 2^-126 <= |y| <= 2^126;   __expf(x) 2 + floor(|1.173 x|) ulp;
 *     __sinf / __cosf / __sincosf absolute error 2^-21.41 on [-pi, pi];
 *     __frsqrt_rn / __fsqrt_rn / __frcp_rn, x / y, sqrtf (nvcc defaults -prec-div / -prec-sqrt): correctly rounded.
 * With -DOSIFT_CUDA_MODEL every such call site moves its correctly rounded result by an error inside that bound, chosen
 * by OSIFT_CUDA_MODEL (environment, read once):
 *     plus        every result at + the bound          minus       every result at - the bound
 *     rand:<seed> an error drawn per call from the uniform distribution over [-bound, +bound], keyed by the operand
 *                 bits, the call site and the seed (deterministic; different seeds = different plausible GPUs)
 *     (unset)     no error: the build then equals the standard one
 * tests/test_oracle_cpu.py::test_sensitivity_to_cuda_fast_math_models runs the standard fixtures through these models and
 * states what a real GPU run may differ by (tests/golden/README.md holds the counts).  Nothing here is ever shipped.
 */
#ifndef OSIFT_CUDA_MODEL_H
#define OSIFT_CUDA_MODEL_H

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef OSIFT_CUDA_MODEL

static int cm_mode_ = -1;          /* 0 off, 1 plus, 2 minus, 3 rand */
static uint32_t cm_seed_ = 0;
static inline int cm_mode(void)
{
    if (cm_mode_ < 0) {
        const char* e = getenv("OSIFT_CUDA_MODEL");
        int m = 0;
        if (e && !strcmp(e, "plus")) m = 1;
        else if (e && !strcmp(e, "minus")) m = 2;
        else if (e && !strncmp(e, "rand", 4)) { m = 3; cm_seed_ = e[4] == ':' ? (uint32_t)strtoul(e + 5, NULL, 10) : 1u; }
        cm_mode_ = m;
    }
    return cm_mode_;
}
static inline uint32_t cm_hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
/* signed unit in [-1, 1] for this call */
static inline float cm_unit(float v, uint32_t site)
{
    const int m = cm_mode();
    if (m == 1) return 1.0f;
    if (m == 2) return -1.0f;
    uint32_t b; memcpy(&b, &v, 4);
    const uint32_t h = cm_hash(b ^ cm_hash(site * 0x9e3779b9u ^ cm_seed_));
    return (float)((double)h / 2147483647.5 - 1.0);
}
/* v moved by up to maxulp units in the last place (of v) */
static inline float cm_ulp(float v, float maxulp, uint32_t site)
{
    if (cm_mode() == 0 || !isfinite(v) || fabsf(v) < 1.17549435e-38f) return v;
    const float u = cm_unit(v, site);
    const int k = (int)lrintf(u * maxulp);
    int32_t b; memcpy(&b, &v, 4);
    b += (v > 0.0f) ? k : -k;                      /* sign-magnitude: + k raises the value of a positive float */
    float r; memcpy(&r, &b, 4);
    return isfinite(r) ? r : v;
}
/* v moved by up to abserr */
static inline float cm_abs(float v, float abserr, uint32_t site)
{
    if (cm_mode() == 0 || !isfinite(v)) return v;
    return v + cm_unit(v, site) * abserr;
}
#define CM_ULP(v, n, site) cm_ulp((v), (n), (site))
#define CM_ABS(v, e, site) cm_abs((v), (e), (site))

#else
#define CM_ULP(v, n, site) (v)
#define CM_ABS(v, e, site) (v)
#endif

/* the call sites, by the CUDA function they stand for */
#define CM_EXPF(x)          CM_ULP(expf(x), 2.0f, 1u)                                         /* expf             */
#define CM_FAST_EXPF(x)     CM_ULP(expf(x), 2.0f + floorf(fabsf(1.173f * (x))), 2u)           /* __expf           */
#define CM_HYPOTF(a, b)     CM_ULP(hypotf((a), (b)), 3.0f, 3u)                                /* hypotf           */
#define CM_ATAN2F(v)        CM_ULP((v), 3.0f, 4u)                                             /* atan2f (on the stand-in's value) */
#define CM_FDIVIDEF(a, b)   CM_ULP((a) / (b), 2.0f, 5u)                                       /* __fdividef       */
#define CM_FAST_SIN(v)      CM_ABS((v), 3.5908e-7f, 6u)                                       /* __sincosf: 2^-21.41 */
#define CM_FAST_COS(v)      CM_ABS((v), 3.5908e-7f, 7u)
/* __frsqrt_rn is correctly rounded: under the model it is the single rounding of 1/sqrt(x), not 1.0f / sqrtf(x) */
#ifdef OSIFT_CUDA_MODEL
#define CM_FRSQRT_RN(x)     (cm_mode() ? (float)(1.0 / sqrt((double)(x))) : 1.0f / sqrtf(x))
#else
#define CM_FRSQRT_RN(x)     (1.0f / sqrtf(x))
#endif

#endif
