// pyramid.hip -- Gaussian scale-space construction for gfx950 (MI355X).
//
// Replaces (behaviour, not code) the reference's default build_pyramid branch
// (s_pyramid_build.cu:547-575):
//   normalizedSource::horiz + absoluteSource::vert (level 0 of octave 0)   -> k_level0
//   absoluteSource::horiz + absoluteSource::vert   (levels 1..L-1)         -> k_blur
//   get_by_2_pick_every_second                                             -> fused into k_blur
//                                                                             (k_downscale standalone)
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

#include <type_traits>

namespace {

constexpr int TW = 64;    // strip width (columns per workgroup)
constexpr int BR = 32;    // rows per marching step
constexpr int NT = 256;   // threads per workgroup

// Blocks with equal (blockIdx % 8) run on the same XCD and share its L2 (MI355X_MICROARCH.md,
// "Workgroup dispatch"); give them contiguous logical ids so that neighbouring strips, which
// share halo columns, hit the same L2.  Bijective for any grid size.  Speed only.
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, k = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct BlurArgs {
    const float* src;
    float*       dst;
    float*       half_dst;      // next octave level 0 (pick every second), or nullptr
    int W, H, pitch, half_pitch;
    int nstrips, chunk_rows;
    PsxTaps taps;
};

template <int R>
struct Geom {
    static constexpr int HALO = (R + 3) & ~3;
    static constexpr int SW   = TW + 2 * HALO;     // staged row width (floats)
    static constexpr int SW4  = SW / 4;
    static constexpr int NLD  = (BR * SW4 + NT - 1) / NT;
    static constexpr int RING = (BR + 2 * R <= 64) ? 64 : 128;
    // LDS row stride of the staged rows: 96 or 160 floats, i.e. == 128 B (mod 256 B).  Together
    // with the XOR swizzle below this makes the horizontal pass's ds_read_b128 conflict-free
    // (checked against the gfx950 b128 lane groups, MI355X_MICROARCH.md "LDS").
    static constexpr int SWA  = ((SW + 63) / 64) * 64 + 32;
};

// Staged rows are stored in 16-byte chunks; chunk c of row r lives at chunk (c ^ (r & 1)).
__device__ __forceinline__ int stage_chunk(int row, int c4) { return c4 ^ (row & 1); }

// Ring rows (64 H-filtered values) are stored permuted so that the horizontal pass writes two
// conflict-free 128-byte runs per row: value of column col sits at ring_pos(col).
__device__ __forceinline__ int ring_col_of_pos(int p) { return ((p & 31) >> 2) * 8 + (p & 3) + ((p >> 5) << 2); }

// horizontal filter of 8 adjacent outputs from a register window; win[HALO+i] is the centre
// of output i.
template <int R, int HALO, bool LEVEL0>
__device__ __forceinline__ void hfilter8(const float* win, const PsxTaps& tp, float* out)
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = HALO + i;
        float o = 0.0f;
        if (!LEVEL0) o = fmaf(win[c], tp.g[0], o);
#pragma unroll
        for (int k = R; k >= 1; k--) o = fmaf(win[c - k] + win[c + k], tp.g[k], o);
        if (LEVEL0) { o = fmaf(win[c], tp.g[0], o); o = o * 255.0f; }
        out[i] = o;
    }
}

// vertical filter: v[j] = T[r_out0 - R + j]; output i is centred on v[R + i]
template <int R>
__device__ __forceinline__ float vfilter(const float* v, int i, const PsxTaps& tp)
{
    float o = 0.0f;
#pragma unroll
    for (int k = R; k >= 1; k--) {
        o = fmaf(v[R + i - k], tp.g[k], o);
        o = fmaf(v[R + i + k], tp.g[k], o);
    }
    o = fmaf(v[R + i], tp.g[0], o);
    return o;
}

// the same chain on two adjacent columns at once (v_pk_fma_f32): v[j] = (T[.][c], T[.][c+1])
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ v2f pk_fma(v2f a, float g, v2f c)
{
    return __builtin_elementwise_fma(a, (v2f){g, g}, c);
}
template <int R>
__device__ __forceinline__ v2f vfilter2(const v2f* v, int i, const PsxTaps& tp)
{
    v2f o = {0.0f, 0.0f};
#pragma unroll
    for (int k = R; k >= 1; k--) {
        o = pk_fma(v[R + i - k], tp.g[k], o);
        o = pk_fma(v[R + i + k], tp.g[k], o);
    }
    o = pk_fma(v[R + i], tp.g[0], o);
    return o;
}

#ifdef PSX_PHASE_TIMING
__device__ long long* g_blur_dbg = nullptr;
extern "C" void psx_debug_set_blur_buffer(long long* d) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_blur_dbg), &d, sizeof(d)); }
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

// k-major forms of the two filters: the per-output chains are the reference's, but the independent
// chains advance together so that dependent v_pk_fma_f32 never issue back to back
template <int R, int HALO>
__device__ __forceinline__ void hfilter8_km(const float* win, const PsxTaps& tp, float* out)
{
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = fmaf(win[HALO + i], tp.g[0], 0.0f);
#pragma unroll
    for (int k = R; k >= 1; k--) {
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = fmaf(win[HALO + i - k] + win[HALO + i + k], tp.g[k], out[i]);
    }
}
template <int R>
__device__ __forceinline__ void vfilter2x4_km(const v2f* v, const PsxTaps& tp, v2f* o)
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

template <int R>
__global__ __launch_bounds__(NT, (R <= 13) ? 4 : ((R <= 22) ? 2 : 1)) void k_blur(BlurArgs a)
{
    using G = Geom2<R>;
    constexpr int HALO = G::HALO, SW4 = G::SW4, NLD = G::NLD, SWA = G::SWA, RING = G::RING;
    constexpr int VWIN = G::VWIN, MIRROR = G::MIRROR, RS = G::RS;
    constexpr bool LAST_PARTIAL = (BR * SW4) % NT != 0;     // only the last staging slot can be empty
    __shared__ __attribute__((aligned(16))) float s_stage[BR * SWA];
    __shared__ __attribute__((aligned(16))) float s_ring[(RING + MIRROR) * RS];

    const int t     = threadIdx.x;
    const int lid   = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = lid % a.nstrips;
    const int chunk = lid / a.nstrips;
    const int x0    = strip * TW;
    const int Y0    = chunk * a.chunk_rows;
    const int Y1    = min(Y0 + a.chunk_rows, a.H);
    const int nsteps = (Y1 - Y0 + 2 * R + BR - 1) / BR;
    const bool interior = (x0 - HALO >= 0) && (x0 + TW + HALO <= a.W);     // workgroup uniform

    // ---- staging geometry of this thread (step invariant) ----
    int st_row[NLD], st_x[NLD], st_lds[NLD];
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int idx = t + j * NT;
        const int row = idx / SW4, c4 = idx - row * SW4;
        st_row[j] = row;
        st_x[j]   = x0 - HALO + c4 * 4;
        st_lds[j] = row * SWA + c4 * 4;
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
        auto issue = [&](int k) {
            const int ybase = Y0 - R + k * BR;
#pragma unroll
            for (int j = 0; j < NLD; j++) {
                if (j < NLD - 1 || last_on) {
                    const int y = psx_clampi(ybase + st_row[j], 0, a.H - 1);
                    if (INTERIOR) {
                        const unsigned off = (unsigned)(y * a.pitch + st_x[j]) * 4u;
                        pre[j] = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(a.src) + off);
                    } else {
                        const float* rp = a.src + (size_t)y * a.pitch;
                        const int x = st_x[j];
                        pre[j].x = rp[psx_clampi(x + 0, 0, a.W - 1)];
                        pre[j].y = rp[psx_clampi(x + 1, 0, a.W - 1)];
                        pre[j].z = rp[psx_clampi(x + 2, 0, a.W - 1)];
                        pre[j].w = rp[psx_clampi(x + 3, 0, a.W - 1)];
                    }
                }
            }
        };
        auto commit = [&]() {
#pragma unroll
            for (int j = 0; j < NLD; j++)
                if (j < NLD - 1 || last_on) *reinterpret_cast<v4f*>(&s_stage[st_lds[j]]) = pre[j];
        };

        issue(0);
        for (int k = 0; k < nsteps; k++) {
            commit();
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
                    // odd-aligned pair with ds_read2_b32, at a quarter of the b128 rate and with conflicts)
                    const v4f v = ((const volatile LDS_AS v4f*)h_src)[q];
                    win[4 * q + 0] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
                }
                float out[8];
                hfilter8_km<R, HALO>(win, a.taps, out);
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
                    vfilter2x4_km<R>(v, a.taps, o);
                    // pin the four results here: otherwise each chain is sunk into its own predicated
                    // store block and runs alone, dependent v_pk_fma_f32 back to back
                    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
                    // uniform row base of this step; thread offsets are step invariant
                    char* drow = reinterpret_cast<char*>(a.dst + (ptrdiff_t)(Y0 - 2 * R + k * BR) * a.pitch);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int r_out = r_out0 + i;
                        if (r_out >= Y0 && r_out < Y1 && v_xok) {
                            char* di = drow + (size_t)i * a.pitch * 4 + v_doff;
                            if (v_pair) *reinterpret_cast<v2f*>(di) = o[i]; else *reinterpret_cast<float*>(di) = o[i].x;
                            // get_by_2_pick_every_second: rows and columns 0,2,4,.. (v_x is even)
                            if (a.half_dst != nullptr && (r_out & 1) == 0)
                                a.half_dst[(size_t)(r_out >> 1) * a.half_pitch + (v_x >> 1)] = o[i].x;
                        }
                    }
                }
            }
            BSTAMP(4);
        }
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
// Octave 0, level 0: resample the input image (software model of the reference's normalised,
// clamped, bilinear texture, s_image.cu:138-167) + horizontal "dd" filter + x255 + vertical "inc"
// filter.  U(X, y) = tex2D at the coordinate of output column X; tap k of output x reads U(x-k),
// U(x+k)  (DESIGN.md "octave 0").
// ---------------------------------------------------------------------------------------------
struct Level0Dev {
    const void* img; int w, h, is_float;
    float* dst; int W, H, pitch;
    float shift;
    int nstrips, chunk_rows;
    PsxTaps taps_h, taps_v;
};

__device__ __forceinline__ float l0_texel(const Level0Dev& a, int i, int j)
{
    i = psx_clampi(i, 0, a.w - 1);
    j = psx_clampi(j, 0, a.h - 1);
    if (a.is_float) return static_cast<const float*>(a.img)[(size_t)j * a.w + i];
    return (float)static_cast<const uint8_t*>(a.img)[(size_t)j * a.w + i] / 255.0f;
}
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

template <int R>
__global__ __launch_bounds__(NT) void k_level0(Level0Dev a)
{
    using G = Geom<R>;
    constexpr int HALO = G::HALO, SW = G::SW, RING = G::RING, SWA = G::SWA;
    __shared__ __attribute__((aligned(16))) float s_stage[BR * SWA];
    __shared__ __attribute__((aligned(16))) float s_ring[RING * TW];
    __shared__ float s_lut[256];      // u8 texel -> v/255 (cudaReadModeNormalizedFloat)
    __shared__ int   s_ci[SW];        // per staged column: left texel index and 1.8 weight
    __shared__ float s_ca[SW];
    __shared__ int   s_rj[2 * BR];    // per staged row (double buffered)
    __shared__ float s_rb[2 * BR];
    constexpr int TEXW = 56, TEXH = 24;    // input-texel footprint of one step (covers upscale >= 1; else direct loads)
    __shared__ float s_tex[TEXH * TEXW];

    const int t     = threadIdx.x;
    const int lid   = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = lid % a.nstrips;
    const int chunk = lid / a.nstrips;
    const int x0    = strip * TW;
    const int Y0    = chunk * a.chunk_rows;
    const int Y1    = min(Y0 + a.chunk_rows, a.H);
    const int nsteps = (Y1 - Y0 + 2 * R + BR - 1) / BR;

    s_lut[t] = (float)t / 255.0f;
    if (t < SW) {
        int i0; float al;
        l0_axis(((float)(x0 - HALO + t) + a.shift) / (float)a.W, a.w, i0, al);
        s_ci[t] = i0; s_ca[t] = al;
    }
    // row tables are double buffered: the table of step k+1 is built during step k so that the
    // texel loads of step k+1 can be issued a whole step ahead (software prefetch into registers)
    auto build_rows = [&](int k) {
        if (t < BR) {
            const int y = psx_clampi(Y0 - R + k * BR + t, 0, a.H - 1);
            int j0; float be;
            l0_axis(((float)y + a.shift) / (float)a.H, a.h, j0, be);
            s_rj[(k & 1) * BR + t] = j0; s_rb[(k & 1) * BR + t] = be;
        }
    };
    constexpr int NTEX = (TEXH * 64 + NT - 1) / NT;          // texel loads per thread (64-wide rows)
    float treg[NTEX];
    int jlo = 0, ilo = 0, nrows = 0, ncols = 0;
    bool via_lds = false;
    auto footprint = [&](int k) {
        const int* rj = &s_rj[(k & 1) * BR];
        jlo = psx_clampi(rj[0], 0, a.h - 1);
        const int jhi = psx_clampi(rj[BR - 1] + 1, 0, a.h - 1);
        ilo = psx_clampi(s_ci[0], 0, a.w - 1);
        const int ihi = psx_clampi(s_ci[SW - 1] + 1, 0, a.w - 1);
        nrows = jhi - jlo + 1; ncols = ihi - ilo + 1;
        via_lds = (nrows <= TEXH && ncols <= TEXW);             // uniform over the workgroup
    };
    auto issue_texels = [&]() {
        if (!via_lds) return;
        const int ic = t & 63;
#pragma unroll
        for (int q = 0; q < NTEX; q++) {
            const int jr = (t >> 6) + q * (NT / 64);
            treg[q] = 0.0f;
            if (jr < nrows && ic < ncols) {
                const size_t g = (size_t)(jlo + jr) * a.w + (ilo + ic);
                treg[q] = a.is_float ? static_cast<const float*>(a.img)[g]
                                     : (float)static_cast<const uint8_t*>(a.img)[g];    // LUT applied at commit
            }
        }
    };

#ifdef PSX_PHASE_TIMING
    long long tacc[5] = {0, 0, 0, 0, 0};
    long long tprev = clock64();
    const long long tstart = tprev;
#endif
    build_rows(0);
    __syncthreads();
    footprint(0);
    issue_texels();

    for (int k = 0; k < nsteps; k++) {
        // commit the prefetched texels of this step (previous U staging has passed two barriers)
        if (via_lds) {
            const int ic = t & 63;
#pragma unroll
            for (int q = 0; q < NTEX; q++) {
                const int jr = (t >> 6) + q * (NT / 64);
                if (jr < nrows && ic < ncols)
                    s_tex[jr * TEXW + ic] = a.is_float ? treg[q] : s_lut[(int)treg[q]];
            }
        }
        const int cjlo = jlo, cilo = ilo;
        const bool cvia = via_lds;
        if (k + 1 < nsteps) build_rows(k + 1);
        BSTAMP(0);
        __syncthreads();   // texels + next row table visible; previous H pass finished reading s_stage
        BSTAMP(1);
        if (k + 1 < nsteps) { footprint(k + 1); issue_texels(); }
        const int* rjk = &s_rj[(k & 1) * BR];
        const float* rbk = &s_rb[(k & 1) * BR];
        // U(X, y) = lerp_y( lerp_x(T[j0]), lerp_x(T[j0+1]) ): same operations, same order as the texture
        // model of the oracle
        // All LDS / global reads of a thread's elements are issued before the first lerp: the loops are
        // fully unrolled with a static trip count, and the (workgroup-uniform) source selection is
        // hoisted OUTSIDE them -- a per-element select makes hipcc branch and wait per element.
        {
            constexpr int NPT = (BR * SW + NT - 1) / NT;
            auto stage_elems = [&](auto mode) {
                constexpr int MODE = decltype(mode)::value;     // 0: LDS texels, 1: float image, 2: u8 image
                float t00[NPT], t10[NPT], t01[NPT], t11[NPT], al[NPT], be[NPT];
#pragma unroll
                for (int q = 0; q < NPT; q++) {
                    const int idx = min(t + q * NT, BR * SW - 1);
                    const int row = idx / SW, c = idx - row * SW;
                    const int i0 = s_ci[c], j0 = rjk[row];
                    al[q] = s_ca[c]; be[q] = rbk[row];
                    const int ia = psx_clampi(i0, 0, a.w - 1), ib = psx_clampi(i0 + 1, 0, a.w - 1);
                    const int ja = psx_clampi(j0, 0, a.h - 1), jb = psx_clampi(j0 + 1, 0, a.h - 1);
                    if (MODE == 0) {
                        const float* ra = &s_tex[(ja - cjlo) * TEXW - cilo];
                        const float* rb = &s_tex[(jb - cjlo) * TEXW - cilo];
                        t00[q] = ra[ia]; t10[q] = ra[ib]; t01[q] = rb[ia]; t11[q] = rb[ib];
                    } else if (MODE == 1) {
                        const float* f = static_cast<const float*>(a.img);
                        t00[q] = f[(size_t)ja * a.w + ia]; t10[q] = f[(size_t)ja * a.w + ib];
                        t01[q] = f[(size_t)jb * a.w + ia]; t11[q] = f[(size_t)jb * a.w + ib];
                    } else {
                        const uint8_t* b = static_cast<const uint8_t*>(a.img);
                        t00[q] = (float)b[(size_t)ja * a.w + ia]; t10[q] = (float)b[(size_t)ja * a.w + ib];
                        t01[q] = (float)b[(size_t)jb * a.w + ia]; t11[q] = (float)b[(size_t)jb * a.w + ib];
                    }
                }
#pragma unroll
                for (int q = 0; q < NPT; q++) {
                    const int idx = t + q * NT;
                    if (idx < BR * SW) {
                        const int row = idx / SW, c = idx - row * SW;
                        float p00 = t00[q], p10 = t10[q], p01 = t01[q], p11 = t11[q];
                        if (MODE == 2) { p00 = s_lut[(int)p00]; p10 = s_lut[(int)p10]; p01 = s_lut[(int)p01]; p11 = s_lut[(int)p11]; }
                        const float r0 = l0_lerp(p00, p10, al[q]);
                        const float r1 = l0_lerp(p01, p11, al[q]);
                        s_stage[row * SWA + stage_chunk(row, c >> 2) * 4 + (c & 3)] = l0_lerp(r0, r1, be[q]);
                    }
                }
            };
            if (cvia) stage_elems(std::integral_constant<int, 0>{});
            else if (a.is_float) stage_elems(std::integral_constant<int, 1>{});
            else stage_elems(std::integral_constant<int, 2>{});
        }
        BSTAMP(2);
        __syncthreads();
        {
            const int row = t >> 3, seg = t & 7;
            float win[8 + 2 * HALO];
            const float4* sp = reinterpret_cast<const float4*>(&s_stage[row * SWA]);
#pragma unroll
            for (int q = 0; q < (8 + 2 * HALO) / 4; q++) {
                const float4 v = sp[stage_chunk(row, seg * 2 + q)];
                win[4 * q + 0] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
            }
            float out[8];
            hfilter8<R, HALO, true>(win, a.taps_h, out);
            const int slot = (k * BR + row) & (RING - 1);
            float4* rp = reinterpret_cast<float4*>(&s_ring[slot * TW + seg * 4]);
            rp[0] = make_float4(out[0], out[1], out[2], out[3]);
            rp[8] = make_float4(out[4], out[5], out[6], out[7]);
        }
        BSTAMP(3);
        __syncthreads();
        {
            const int pos = t & (TW - 1), rg = t >> 6;
            const int col = ring_col_of_pos(pos);
            const int rel0 = k * BR - 2 * R + rg * 8;
            const int r_out0 = Y0 + rel0;
            if (r_out0 + 7 >= Y0 && r_out0 < Y1) {
                float v[8 + 2 * R];
#pragma unroll
                for (int j = 0; j < 8 + 2 * R; j++) v[j] = s_ring[((rel0 + j) & (RING - 1)) * TW + pos];
                const int x = x0 + col;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int r_out = r_out0 + i;
                    const float o = vfilter<R>(v, i, a.taps_v);
                    if (r_out >= Y0 && r_out < Y1 && x < a.W) a.dst[(size_t)r_out * a.pitch + x] = o;
                }
            }
        }
        BSTAMP(4);
    }
#ifdef PSX_PHASE_TIMING
    if (threadIdx.x == 0 && g_blur_dbg) {
        long long* d_ = g_blur_dbg + 1100 * 8;     // level-0 region of the debug buffer
        for (int q = 0; q < 5; q++) d_[blockIdx.x * 8 + q] = tacc[q];
        d_[blockIdx.x * 8 + 5] = clock64() - tstart; d_[blockIdx.x * 8 + 6] = nsteps;
    }
#endif
}

// get_by_2_pick_every_second (s_pyramid_build.cu:50-71), used only when the fused path is off
__global__ void k_downscale(const float* src, int sw, int sh, int spitch, float* dst, int W, int H, int pitch)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W || y >= H) return;
    const int rx = min(x << 1, sw - 1);
    const int ry = min(y << 1, sh - 1);
    dst[(size_t)y * pitch + x] = src[(size_t)ry * spitch + rx];
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

inline void chunking(int W, int H, int R, int& chunk_rows, int& nchunks)
{
    // S marching steps per chunk: the 2R warm-up rows cost ~2R/(S*BR) extra horizontal work, but a
    // chunk is a serial chain of S steps.  Large planes take S=5; small octaves trade efficiency
    // for more, shorter workgroups (they are latency bound, not bandwidth bound).
    const int nstrips = (W + TW - 1) / TW;
    int S = 5;
    for (; S > 2; S--) {
        const int cr = S * BR - 2 * R;
        if (cr >= BR && nstrips * ((H + cr - 1) / cr) >= 384) break;
    }
    int cr = S * BR - 2 * R;
    if (cr < BR / 2) cr = BR / 2;
    if (cr > H) cr = H;
    chunk_rows = cr;
    nchunks = (H + cr - 1) / cr;
}

template <int R>
hipError_t launch_blur_r(const float* src, float* dst, int W, int H, int pitch, const PsxTaps& taps,
                         float* half_dst, int half_pitch, hipStream_t s)
{
    BlurArgs a;
    a.src = src; a.dst = dst; a.half_dst = half_dst;
    a.W = W; a.H = H; a.pitch = pitch; a.half_pitch = half_pitch;
    a.nstrips = (W + TW - 1) / TW;
    int nchunks;
    chunking(W, H, R, a.chunk_rows, nchunks);
    a.taps = taps;
    hipLaunchKernelGGL(k_blur<R>, dim3(a.nstrips * nchunks), dim3(NT), 0, s, a);
    return hipGetLastError();
}

template <int R>
hipError_t launch_level0_r(const PsxLevel0Args& h, hipStream_t s)
{
    Level0Dev a;
    a.img = h.img; a.w = h.w; a.h = h.h; a.is_float = h.is_float;
    a.dst = h.dst; a.W = h.W; a.H = h.H; a.pitch = h.pitch;
    a.shift = h.shift;
    a.nstrips = (h.W + TW - 1) / TW;
    int nchunks;
    chunking(h.W, h.H, R, a.chunk_rows, nchunks);
    a.taps_h = h.taps_h; a.taps_v = h.taps_v;
    hipLaunchKernelGGL(k_level0<R>, dim3(a.nstrips * nchunks), dim3(NT), 0, s, a);
    return hipGetLastError();
}

} // namespace

// span = one-sided tap count including the centre (GaussTable::span); radius R = span-1.
// Kernels are instantiated for a set of radii; a smaller radius runs on the next larger
// instantiation with zero weights, which is bit-exact (fma(x, 0, acc) == acc).
hipError_t psx_launch_blur(const float* src, float* dst, int W, int H, int pitch, const PsxTaps& taps,
                           int span, float* half_dst, int half_pitch, hipStream_t s)
{
    const int R = span - 1;
    if (R <= 5)  return launch_blur_r<5>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s);
    if (R <= 7)  return launch_blur_r<7>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s);
    if (R <= 8)  return launch_blur_r<8>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s);
    if (R <= 10) return launch_blur_r<10>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s);
    if (R <= 13) return launch_blur_r<13>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s);
    if (R <= 16) return launch_blur_r<16>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s);
    if (R <= 22) return launch_blur_r<22>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s);
    if (R <= 30) return launch_blur_r<30>(src, dst, W, H, pitch, taps, half_dst, half_pitch, s);
    return hipErrorInvalidValue;
}

hipError_t psx_launch_level0(const PsxLevel0Args& a, hipStream_t s)
{
    const int R = (a.span_h > a.span_v ? a.span_h : a.span_v) - 1;
    if (R <= 5)  return launch_level0_r<5>(a, s);
    if (R <= 8)  return launch_level0_r<8>(a, s);
    if (R <= 16) return launch_level0_r<16>(a, s);
    if (R <= 30) return launch_level0_r<30>(a, s);
    return hipErrorInvalidValue;
}

hipError_t psx_launch_downscale(const float* src, int sw, int sh, int spitch,
                                float* dst, int W, int H, int pitch, hipStream_t s)
{
    hipLaunchKernelGGL(k_downscale, dim3((W + 255) / 256, H), dim3(256), 0, s, src, sw, sh, spitch, dst, W, H, pitch);
    return hipGetLastError();
}

hipError_t psx_launch_dog(const float* a, const float* b, float* d, int W, int H, int pitch, hipStream_t s)
{
    hipLaunchKernelGGL(k_dog, dim3((W + 255) / 256, H), dim3(256), 0, s, a, b, d, W, H, pitch);
    return hipGetLastError();
}
