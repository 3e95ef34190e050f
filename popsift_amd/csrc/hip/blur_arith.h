// blur_arith.h -- the arithmetic of the separable Gaussian, shared by every kernel that blurs a plane
// (pyramid.hip: marching strips; pyramid_tile.hip: multi-level tiles) and by the host emulation of the tile
// kernel (tests/cpp/tile_emu.cpp, built with PSX_TILE_EMU: the same functions compiled for the CPU).
//
// The operation order is the reference's, written with explicit fma (every user is compiled with
// -ffp-contract=off), so planes are bit-identical to oracle/sift_oracle.c:
//   H (s_pyramid_build_aa.cu:17-50): centre, then pairs (x-k)+(x+k) from k=span-1 down to 1
//   V (s_pyramid_build_aa.cu:52-86): k=span-1..1: acc+=T[y-k]*g; acc+=T[y+k]*g; then centre
//   level 0 H (s_pyramid_build_ra.cu:17-55): pairs outermost-in, then centre, then *255
#pragma once

#include "popsift_hip.h"

#ifdef PSX_TILE_EMU
#include <cmath>
#define PSX_DEV static inline
struct PsxTaps { float g[PSX_GAUSS_ALIGN]; };
#else
#include "psx_internal.h"
#define PSX_DEV __device__ __forceinline__
#endif

// the same chain on two adjacent columns at once (v_pk_fma_f32): v[j] = (T[.][c], T[.][c+1])
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

PSX_DEV v2f pk_fma(v2f a, float g, v2f c)
{
    return __builtin_elementwise_fma(a, (v2f){g, g}, c);
}

// k-major forms of the two filters: the per-output chains are the reference's, but the independent
// chains advance together so that dependent v_pk_fma_f32 never issue back to back
template <int R, int HALO, bool LEVEL0>
PSX_DEV void hfilter8_km(const float* win, const PsxTaps& tp, float* out)
{
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = LEVEL0 ? 0.0f : fmaf(win[HALO + i], tp.g[0], 0.0f);
#pragma unroll
    for (int k = R; k >= 1; k--) {
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = fmaf(win[HALO + i - k] + win[HALO + i + k], tp.g[k], out[i]);
    }
    if (LEVEL0) {
#pragma unroll
        for (int i = 0; i < 8; i++) { out[i] = fmaf(win[HALO + i], tp.g[0], out[i]); out[i] = out[i] * 255.0f; }
    }
}
template <int R>
PSX_DEV void vfilter2x4_km(const v2f* v, const PsxTaps& tp, v2f* o)
{
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (v2f){0.0f, 0.0f};
#pragma unroll
    for (int k = R; k >= 1; k--) {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = pk_fma(v[R + i - k], tp.g[k], o[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = pk_fma(v[R + i + k], tp.g[k], o[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = pk_fma(v[R + i], tp.g[0], o[i]);
}
