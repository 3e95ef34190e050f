// pyramid_alt.hip -- the non-default branches of Pyramid::build_pyramid (s_pyramid_build.cu:478-546):
//   GaussMode VLFeat_Relative      tap pairs through one linear-filtered fetch  (s_pyramid_build_ai.cu:17-69)
//   GaussMode VLFeat_Relative_All  octave 0: every level straight from the input (s_pyramid_build_ra.cu:92-132,
//                                  s_pyramid_build_aa.cu:124-186)
//   GaussMode Fixed9 / Fixed15     9- / 15-tap octaves, levels from level 0     (s_pyramid_fixed.cu:24-298)
//   ScalingMode ScaleDirect        level 0 of EVERY octave straight from the input (s_pyramid_build_ra.cu:17-55)
//
// These modes exist for API completeness of setGaussMode / setScalingMode; each gives a different numerical
// result and has its own branch in the CPU restatement (oracle/sift_oracle.c build_pyramid), pinned against the
// reference's own kernels.  The kernels of this file are written for exactness, not for the roofline: one thread per
// output pixel, separate horizontal / vertical launches through an intermediate plane, the reference's operation order
// with explicit fmaf (this file is compiled with -ffp-contract=off).  The texture unit the reference relies on
// (normalised / unnormalised coordinates, clamp addressing, linear filtering with 1.8 fixed-point weights) is
// software here, as in pyramid.hip.
// Round 4: where a branch is the arithmetic of the default pyramid with other tables -- every level of
// VLFeat_Relative_All's octave 0, level 0 of every ScaleDirect octave, the levels >= 1 of both -- it runs on
// pyramid.hip's kernels (psx_launch_level0 / psx_launch_blur); k_alt_h_input + k_alt_v_plain remain what
// psx_launch_level0 itself falls back to when the image / octave ratio is not a power of two
// (psx_launch_level0_literal below: the per-tap texture coordinates as the reference forms them).
#include "psx_internal.h"

namespace {

constexpr int ANT = 256;

struct AltImg { const void* px; int w, h, is_float; };

// ---- the input image as a normalised, clamped, linearly filtered texture (s_image.cu:138-167) ----
__device__ __forceinline__ float a_texel(const AltImg& t, int i, int j)
{
    i = psx_clampi(i, 0, t.w - 1);
    j = psx_clampi(j, 0, t.h - 1);
    if (t.is_float) return static_cast<const float*>(t.px)[(size_t)j * t.w + i];
    // q / 255 correctly rounded without the IEEE division sequence (~10 instructions, four times per bilinear fetch, 9-15
    // fetches per output in the fixed-span modes): q * RN(1/255) plus one Newton correction is fl(q / 255) for all 256
    // inputs -- the same three instructions as l0_unorm8 in pyramid.hip (tests/test_gpu_parity.py::test_u8_normalisation_exact
    // walks all values through that one; the mode tests hold this one to the oracle's planes bit for bit)
    const float f = (float)static_cast<const uint8_t*>(t.px)[(size_t)j * t.w + i];
    const float c = 1.0f / 255.0f;
    const float r = f * c;
    return fmaf(fmaf(-255.0f, r, f), c, r);
}
__device__ __forceinline__ void a_axis(float cn, int size, int& i0, float& a)
{
    const float tb = cn * (float)size - 0.5f;
    const float fl = floorf(tb);
    a = rintf((tb - fl) * 256.0f) * (1.0f / 256.0f);       // 1.8 fixed-point filter weight
    i0 = (int)fl;
}
__device__ __forceinline__ float a_lerp(float p, float q, float a) { return fmaf(a, q, (1.0f - a) * p); }
__device__ __forceinline__ float tex2d_norm(const AltImg& t, float un, float vn)
{
    int i0, j0; float a, b;
    a_axis(un, t.w, i0, a);
    a_axis(vn, t.h, j0, b);
    const float r0 = a_lerp(a_texel(t, i0, j0),     a_texel(t, i0 + 1, j0),     a);
    const float r1 = a_lerp(a_texel(t, i0, j0 + 1), a_texel(t, i0 + 1, j0 + 1), a);
    return a_lerp(r0, r1, b);
}

// ---- a Gaussian plane as an unnormalised, clamped, linearly filtered texture (readTex, assist.h:68-77): plane_linear_1d
// below; the interpolating descriptor modes have their own 2-D form in orient_desc.hip ----

// normalizedSource::horiz / horiz_level / horiz_all (s_pyramid_build_ra.cu:17-132)
__global__ __launch_bounds__(ANT) void k_alt_h_input(AltImg t, float* intm, int W, int H, int pitch, PsxTaps f, int span, float shift)
{
    const int x = blockIdx.x * ANT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float read_x = ((float)x + shift) / W;
    const float read_y = ((float)y + shift) / H;
    // float(offset) / W per tap (the reference divides): q = k * RN(1 / W), one Newton correction -- equal to the IEEE
    // quotient for every k <= 32 and every W < 2^17 (checked exhaustively on the CPU), three instructions instead of ten
    const float fW = (float)W, rW = 1.0f / fW;
    const bool small_w = W < (1 << 17);
    float out = 0.0f;
    for (int offset = span; offset > 0; offset--) {
        const float fo = (float)offset, q0 = fo * rW;
        const float offrel = small_w ? fmaf(fmaf(-fW, q0, fo), rW, q0) : fo / fW;
        const float v1 = tex2d_norm(t, read_x - offrel, read_y);
        const float v2 = tex2d_norm(t, read_x + offrel, read_y);
        out = fmaf(v1 + v2, f.g[offset], out);
    }
    out = fmaf(tex2d_norm(t, read_x, read_y), f.g[0], out);
    intm[(size_t)y * pitch + x] = out * 255.0f;
}

// absoluteSource::vert / vert_abs0 (s_pyramid_build_aa.cu:52-122)
__global__ __launch_bounds__(ANT) void k_alt_v_plain(const float* intm, float* dst, int W, int H, int pitch, PsxTaps f, int span)
{
    const int x = blockIdx.x * ANT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    float out = 0.0f;
    for (int offset = span; offset > 0; offset--) {
        const float g = f.g[offset];
        out = fmaf(intm[(size_t)psx_clampi(y - offset, 0, H - 1) * pitch + x], g, out);
        out = fmaf(intm[(size_t)psx_clampi(y + offset, 0, H - 1) * pitch + x], g, out);
    }
    out = fmaf(intm[(size_t)y * pitch + x], f.g[0], out);
    dst[(size_t)y * pitch + x] = out;
}

// absoluteSource::horiz (s_pyramid_build_aa.cu:17-50): centre, the (zero-weight) outermost pair, then inwards
__global__ __launch_bounds__(ANT) void k_alt_h_plain(const float* src, float* intm, int W, int H, int pitch, PsxTaps f, int span)
{
    const int x = blockIdx.x * ANT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float* row = src + (size_t)y * pitch;
    float out = 0.0f;
    out = fmaf(row[x], f.g[0], out);
    out = fmaf(row[psx_clampi(x - span, 0, W - 1)] + row[psx_clampi(x + span, 0, W - 1)], f.g[span], out);
    for (int offset = span - 1; offset > 0; offset--)
        out = fmaf(row[psx_clampi(x - offset, 0, W - 1)] + row[psx_clampi(x + offset, 0, W - 1)], f.g[offset], out);
    intm[(size_t)y * pitch + x] = out;
}

// plane_linear with one of the two coordinates an integer pixel index (xi, yi inside the plane) and the other one, t,
// fractional.  On the integer axis the 1.8 weight is exactly 0 and lerp(p, q, 0) = p + 0 * q = p for the finite,
// non-negative values of a Gaussian plane (a zero keeps its sign only if q's sign agrees: planes of images in [0, 1] never
// hold a negative zero), so the neighbour on that axis is neither loaded nor multiplied: half the loads, a third of the lerps.
template <bool VERTICAL>
__device__ __forceinline__ float plane_linear_1d(const float* p, int W, int H, int pitch, int xi, int yi, float t)
{
    const float ts = t + 0.5f, tb = ts - 0.5f;                 // readTex adds 0.5, the texture unit takes it off again
    const float ft = floorf(tb);
    const float w = rintf((tb - ft) * 256.0f) * (1.0f / 256.0f);
    const int k = (int)ft;
    if (VERTICAL) {
        const int j0 = psx_clampi(k, 0, H - 1), j1 = psx_clampi(k + 1, 0, H - 1);
        return a_lerp(p[(size_t)j0 * pitch + xi], p[(size_t)j1 * pitch + xi], w);
    }
    const int i0 = psx_clampi(k, 0, W - 1), i1 = psx_clampi(k + 1, 0, W - 1);
    return a_lerp(p[(size_t)yi * pitch + i0], p[(size_t)yi * pitch + i1], w);
}

// absoluteSourceInterpolated::horiz / vert (s_pyramid_build_ai.cu:17-69)
template <bool VERTICAL>
__global__ __launch_bounds__(ANT) void k_alt_interp(const float* src, float* dst, int W, int H, int pitch, PsxTaps fi, int ispan)
{
    const int x = blockIdx.x * ANT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    float out = 0.0f;
    const float c = VERTICAL ? (float)y : (float)x;             // the coordinate the taps move along
    for (int offset = 1; offset <= ispan; offset += 2) {
        const float u = fi.g[offset];
        const float off = offset + (1.0f - u);
        const float val = plane_linear_1d<VERTICAL>(src, W, H, pitch, x, y, c - off) + plane_linear_1d<VERTICAL>(src, W, H, pitch, x, y, c + off);
        out = fmaf(val, fi.g[offset + 1], out);
    }
    out = fmaf(src[(size_t)y * pitch + x], fi.g[0], out);       // both coordinates integer: the pixel itself
    dst[(size_t)y * pitch + x] = out;
}

// fixedSpan::relativeTexAddress::octave_fixed_vert (s_pyramid_fixed.cu:123-140): column cx - SHIFT of the vbuf.
// The 2 SHIFT + 1 fetches of an output share their column (texel columns i0 / i0 + 1 and weight a: formed once) and walk up
// the texel rows: they are taken in ascending order, the x-lerp of a texel row is kept while consecutive fetches need it
// (at x2 upsampling a fetch advances half a texel row: 6 row lerps instead of 18 for 9 taps), and the weighted sum then runs
// over the stored values in the reference's order (centre, then the pairs -i / +i) -- the same values, the same sum.
template <int SHIFT>
__global__ __launch_bounds__(ANT) void k_fixed_v_input(AltImg t, float* vbuf, int W, int H, int vpitch, PsxTaps f, float tshift)
{
    const int cx = blockIdx.x * ANT + threadIdx.x - SHIFT, y = blockIdx.y;
    if (cx >= W + SHIFT) return;
    const float mul_w = 1.0f / (float)W, mul_h = 1.0f / (float)H;        // __frcp_rn
    const float xpos = ((float)cx + tshift) * mul_w;
    const float ypos = ((float)y + tshift) * mul_h;
    int i0; float a;
    a_axis(xpos, t.w, i0, a);
    const int ia = psx_clampi(i0, 0, t.w - 1), ib = psx_clampi(i0 + 1, 0, t.w - 1);
    auto row = [&](int j) { return a_lerp(a_texel(t, ia, j), a_texel(t, ib, j), a); };      // j already clamped
    float val[2 * SHIFT + 1];
    int cj0 = -1, cj1 = -1;                                                 // texel rows whose x-lerp is held in r0 / r1
    float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
    for (int k = -SHIFT; k <= SHIFT; k++) {
        // ypos -+ i * mul_h is ONE fma in the reference's device code (nvcc contracts it; this file is built -ffp-contract=off)
        const float vn = k == 0 ? ypos : fmaf((float)k, mul_h, ypos);
        int j0; float b;
        a_axis(vn, t.h, j0, b);
        // every lane of the block has the same y: scalar row indices, scalar branches
        const int ja = __builtin_amdgcn_readfirstlane(psx_clampi(j0, 0, t.h - 1));
        const int jb = __builtin_amdgcn_readfirstlane(psx_clampi(j0 + 1, 0, t.h - 1));
        const float ra = ja == cj0 ? r0 : (ja == cj1 ? r1 : row(ja));
        const float rb = jb == ja ? ra : (jb == cj1 ? r1 : (jb == cj0 ? r0 : row(jb)));
        cj0 = ja; cj1 = jb; r0 = ra; r1 = rb;
        val[k + SHIFT] = a_lerp(ra, rb, b);
    }
    float fval = val[SHIFT] * f.g[0];
#pragma unroll
    for (int i = 1; i <= SHIFT; i++) fval = fmaf(val[SHIFT - i] + val[SHIFT + i], f.g[i], fval);
    vbuf[(size_t)y * vpitch + cx + SHIFT] = fval;
}

// fixedSpan::absoluteTexAddress::octave_fixed_vert (s_pyramid_fixed.cu:51-70)
__global__ __launch_bounds__(ANT) void k_fixed_v_plane(const float* src, int pitch, float* vbuf, int W, int H, int vpitch, PsxTaps f, int SHIFT)
{
    const int cx = blockIdx.x * ANT + threadIdx.x - SHIFT, y = blockIdx.y;
    if (cx >= W + SHIFT) return;
    const int xc = psx_clampi(cx, 0, W - 1);
    float val = src[(size_t)y * pitch + xc];
    float fval = val * f.g[0];
    for (int i = 1; i <= SHIFT; i++) {
        val = src[(size_t)psx_clampi(y - i, 0, H - 1) * pitch + xc] + src[(size_t)psx_clampi(y + i, 0, H - 1) * pitch + xc];
        fval = fmaf(val, f.g[i], fval);
    }
    vbuf[(size_t)y * vpitch + cx + SHIFT] = fval;
}

// octave_fixed_horiz (s_pyramid_fixed.cu:24-44) as lane N of the warp sees it after shuffle_down(out, SHIFT)
__global__ __launch_bounds__(ANT) void k_fixed_h(const float* vbuf, int vpitch, float* dst, int W, int H, int pitch, PsxTaps f, int SHIFT, float scale)
{
    const int x = blockIdx.x * ANT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float* v = vbuf + (size_t)y * vpitch + x + SHIFT;
    float out = v[0] * f.g[0];
    for (int i = 1; i <= SHIFT; i++) out = fmaf(v[-i] + v[i], f.g[i], out);
    dst[(size_t)y * pitch + x] = (scale != 1.0f) ? out * scale : out;
}

// get_by_2_pick_every_second (s_pyramid_build.cu:50-71)
__global__ __launch_bounds__(ANT) void k_alt_downscale(const float* src, int sw, int sh, int spitch, float* dst, int W, int H, int pitch)
{
    const int x = blockIdx.x * ANT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    dst[(size_t)y * pitch + x] = src[(size_t)min(y << 1, sh - 1) * spitch + min(x << 1, sw - 1)];
}

inline dim3 grid_for(int W, int H) { return dim3((W + ANT - 1) / ANT, H); }

} // namespace

// normalizedSource::horiz + absoluteSource::vert with the tap coordinates formed as the reference forms them,
// (x + shift)/W -+ k/W: level 0 from the input image when the image / octave ratio is not a power of two
// (psx_level0_exact in pyramid.hip says why the fast kernels are not bit-exact there).  a.tmp holds the horizontal pass.
hipError_t psx_launch_level0_literal(const PsxLevel0Args& a, hipStream_t s)
{
    if (a.tmp_pitch < a.pitch) return hipErrorInvalidValue;
    const AltImg img{a.img, a.w, a.h, a.is_float};
    const dim3 g = grid_for(a.W, a.H), b(ANT);
    hipLaunchKernelGGL(k_alt_h_input, g, b, 0, s, img, a.tmp, a.W, a.H, a.pitch, a.taps_h, a.span_h, a.shift);
    hipLaunchKernelGGL(k_alt_v_plain, g, b, 0, s, a.tmp, a.dst, a.W, a.H, a.pitch, a.taps_v, a.span_v);
    return hipGetLastError();
}

// GaussMode VLFeat_Relative with the default scaling, every level on the fused kernels (pyramid_interp.hip): the diagonal
// schedule of the default pyramid (psx_build_pyramid).  Level l of octave o needs level l - 1 of its octave, and level 0 of octave
// o + 1 is written by the launch of level D = L - 3 of octave o; so (o, l) runs in launch slot l + D o, and the jobs of a slot -- two
// for the default 3 levels per octave -- share ONE launch when both grids fit one round of resident workgroups.  The small
// octaves, chains of ~8 us launches on their own, ride along with the octave above: 17 launches instead of 25 at 5 octaves.
static hipError_t relative_diagonal(const PsxAltArgs& a, hipStream_t s)
{
    const PsxParams& P = *a.hp;
    const int L = P.L, D = L - 3, noct = P.num_octaves;
    auto job = [&](int o, int l) {
        const PsxOctave& oc = P.oct[o];
        PsxInterpJob j;
        j.src = oc.data + (size_t)(l - 1) * oc.plane; j.dst = oc.data + (size_t)l * oc.plane;
        const bool feeds = l == D && o + 1 < noct;
        j.half_dst = feeds ? P.oct[o + 1].data : nullptr; j.half_pitch = feeds ? P.oct[o + 1].pitch : 0;
        j.W = oc.w; j.H = oc.h; j.pitch = oc.pitch;
        j.fi = a.inc_ifilter + l * PSX_GAUSS_ALIGN; j.ispan = a.inc_ispan[l];
        return j;
    };
    for (int t = 1; t <= (L - 1) + D * (noct - 1); t++) {
        int os[PSX_MAX_OCTAVES], n = 0;
        for (int o = noct - 1; o >= 0; o--) {           // the deeper octave (the head of the dependency chain) first
            const int l = t - D * o;
            if (l >= 1 && l <= L - 1) os[n++] = o;
        }
        for (int i = 0; i < n;) {
            const int o1 = os[i], l1 = t - D * o1;
            hipError_t e;
            if (i + 1 < n) {
                const int o2 = os[i + 1], l2 = t - D * o2;
                if (psx_blur_interp_pair_ok(P.oct[o1].w, P.oct[o1].h, a.inc_ispan[l1], P.oct[o2].w, P.oct[o2].h, a.inc_ispan[l2])) {
                    e = psx_launch_blur_interp2(job(o1, l1), job(o2, l2), s);
                    if (e != hipSuccess) return e;
                    i += 2;
                    continue;
                }
            }
            e = psx_launch_blur_interp(job(o1, l1), s);
            if (e != hipSuccess) return e;
            i++;
        }
        for (int i = 0; i < n; i++)
            if (t - D * os[i] == L - 1 && a.after_octave) { const hipError_t e = a.after_octave(a.user, os[i]); if (e != hipSuccess) return e; }
    }
    return hipSuccess;
}

// Pyramid::build_pyramid for every mode combination outside the default branch; mirrors build_pyramid() of
// oracle/sift_oracle.c statement by statement.  Returns hipErrorInvalidValue for Fixed9 / Fixed15 with levels != 3
// (the reference: POP_FATAL "Unsupported number of levels for making all octaves at once").
hipError_t psx_launch_pyramid_alt(const PsxAltArgs& a, hipStream_t s)
{
    const PsxParams& P = *a.hp;
    const int gm = a.gauss_mode;
    const bool fixed = (gm == PSX_GAUSS_FIXED9 || gm == PSX_GAUSS_FIXED15);
    const bool direct = (a.scaling_mode == PSX_SCALE_DIRECT);
    const int SHIFT = (gm == PSX_GAUSS_FIXED9) ? 4 : 7;
    if (fixed && P.L != 6) return hipErrorInvalidValue;
    const AltImg img{a.img, a.w, a.h, a.is_float};
    auto taps = [](const float* row) { PsxTaps t; for (int i = 0; i < PSX_GAUSS_ALIGN; i++) t.g[i] = row[i]; return t; };
    auto inc = [&](int l) { return taps(a.inc_filter + l * PSX_GAUSS_ALIGN); };
    auto inci = [&](int l) { return taps(a.inc_ifilter + l * PSX_GAUSS_ALIGN); };
    bool next_l0_done = false;          // level 0 of this octave was written by the previous octave's fused launch
    for (int o = 0; o < P.num_octaves; o++) {
        const PsxOctave& oc = P.oct[o];
        const int W = oc.w, H = oc.h, pitch = oc.pitch;
        const dim3 g = grid_for(W, H), b(ANT);
        auto plane = [&](int l) { return oc.data + (size_t)l * oc.plane; };
        float shift = 0.5f;
        if (o == 0 && (a.sift_mode == PSX_MODE_POPSIFT || a.sift_mode == PSX_MODE_VLFEAT))
            shift = 0.5f * powf(2.0f, a.upscale_factor - o);
        auto downscale = [&]() {
            const PsxOctave& po = P.oct[o - 1];
            hipLaunchKernelGGL(k_alt_downscale, g, b, 0, s, po.data + (size_t)(P.L - 3) * po.plane, po.w, po.h, po.pitch,
                               plane(0), W, H, pitch);
        };
        // normalizedSource::horiz (taps th) + absoluteSource::vert (taps tv) from the input image into `dst`: exactly what
        // level 0 of the default pyramid is, so it goes through psx_launch_level0: the default path's kernels (k_level0_x2
        // at x2 with a radius <= 8, k_upscale + k_blur<R, true> otherwise) when the image / octave ratio is a power of two
        // (bit-identical planes, ~10x faster on octave 0), k_alt_h_input + k_alt_v_plain when it is not
        auto level_from_input = [&](float* dst, const PsxTaps& th, int sh, const PsxTaps& tv, int sv) -> hipError_t {
            PsxLevel0Args l0;
            l0.img = a.img; l0.w = a.w; l0.h = a.h; l0.is_float = a.is_float;
            l0.dst = dst; l0.W = W; l0.H = H; l0.pitch = pitch;
            l0.tmp = a.up; l0.tmp_pitch = a.up_pitch;
            l0.shift = shift;
            l0.taps_h = th; l0.span_h = sh; l0.taps_v = tv; l0.span_v = sv;
            return psx_launch_level0(l0, s);
        };
        auto fixed_levels = [&](int first, bool from_input) {
            const int vpitch = a.vbuf_pitch;
            const dim3 gv((W + 2 * SHIFT + ANT - 1) / ANT, H);
            for (int level = first; level < P.L; level++) {
                const PsxTaps f = taps((from_input ? a.abs0_filter : a.absN_filter) + level * PSX_GAUSS_ALIGN);
                if (from_input) {
                    const float tshift = 0.5f * powf(2.0f, a.upscale_factor);
                    if (SHIFT == 4) hipLaunchKernelGGL(k_fixed_v_input<4>, gv, b, 0, s, img, a.vbuf, W, H, vpitch, f, tshift);
                    else            hipLaunchKernelGGL(k_fixed_v_input<7>, gv, b, 0, s, img, a.vbuf, W, H, vpitch, f, tshift);
                } else {
                    hipLaunchKernelGGL(k_fixed_v_plane, gv, b, 0, s, plane(0), pitch, a.vbuf, W, H, vpitch, f, SHIFT);
                }
                hipLaunchKernelGGL(k_fixed_h, g, b, 0, s, a.vbuf, vpitch, plane(level), W, H, pitch, f, SHIFT, from_input ? 255.0f : 1.0f);
            }
        };
        if (fixed) {
            // one launch per octave (pyramid_fixed.hip: level 0 read once, every derived level and the next octave's level 0
            // written from LDS) wherever that kernel applies; the per-level kernels above otherwise
            const bool fused = psx_fixed_octave_enabled();
            PsxFixedOctaveArgs fo;
            fo.src_w = a.w; fo.src_h = a.h; fo.is_float = a.is_float;
            fo.shift = SHIFT; fo.plane = oc.plane; fo.W = W; fo.H = H; fo.pitch = pitch;
            fo.half_dst = nullptr; fo.half_pitch = 0;
            if (!direct && o + 1 < P.num_octaves) { fo.half_dst = P.oct[o + 1].data; fo.half_pitch = P.oct[o + 1].pitch; }
            fo.ev0 = fo.ev1 = nullptr;
            if (o == 0) {
                if (fused && psx_fixed_octave0_ok(a.w, a.h, W, H) && a.upscale_factor == 1.0f) {
                    fo.src = a.img; fo.from_input = 1; fo.nlev = 6; fo.dst = plane(0); fo.half_level = P.L - 3;
                    fo.scale = 255.0f; fo.taps = a.abs0_filter;
                    fo.ev0 = a.probe_ev0; fo.ev1 = a.probe_ev1;
                    if (a.probe_hit) *a.probe_hit = 1;
                    const hipError_t e2 = psx_launch_fixed_octave(fo, s);
                    fo.ev0 = fo.ev1 = nullptr;
                    if (e2 != hipSuccess) return e2;
                    next_l0_done = fo.half_dst != nullptr;
                } else { fixed_levels(0, true); next_l0_done = false; }
            } else {
                if (direct) {
                    hipLaunchKernelGGL(k_alt_h_input, g, b, 0, s, img, a.intm, W, H, pitch, taps(a.dd_filter + o * PSX_GAUSS_ALIGN), a.dd_span[o], shift);
                    hipLaunchKernelGGL(k_alt_v_plain, g, b, 0, s, a.intm, plane(0), W, H, pitch, inc(0), a.inc_span[0]);
                } else if (!next_l0_done) downscale();
                if (fused) {
                    fo.src = plane(0); fo.from_input = 0; fo.nlev = 5; fo.dst = plane(1); fo.half_level = P.L - 3 - 1;
                    fo.scale = 1.0f; fo.taps = a.absN_filter + PSX_GAUSS_ALIGN;
                    const hipError_t e2 = psx_launch_fixed_octave(fo, s);
                    if (e2 != hipSuccess) return e2;
                    next_l0_done = fo.half_dst != nullptr;
                } else { fixed_levels(1, false); next_l0_done = false; }
            }
        } else if (direct) {
            const bool interp = (gm == PSX_GAUSS_VLFEAT_RELATIVE);
            for (int level = 0; level < P.L; level++) {
                if (!interp) {
                    // ScaleDirect with the plain tables: level 0 of every octave straight from the input image, the other
                    // levels absoluteSource::horiz + vert -- the default path's fused kernels
                    const hipError_t e2 = level == 0 ? level_from_input(plane(0), taps(a.dd_filter + o * PSX_GAUSS_ALIGN), a.dd_span[o], inc(0), a.inc_span[0])
                                                     : psx_launch_blur(plane(level - 1), plane(level), W, H, pitch, inc(level), a.inc_span[level], nullptr, 0, s);
                    if (e2 != hipSuccess) return e2;
                    continue;
                }
                if (level > 0 && psx_blur_interp_ok(a.inc_ispan[level])) {
                    // the fused kernel of pyramid_interp.hip (no decimation: ScaleDirect takes level 0 of every octave from the input)
                    PsxInterpJob ij;
                    ij.src = plane(level - 1); ij.dst = plane(level); ij.W = W; ij.H = H; ij.pitch = pitch;
                    ij.half_dst = nullptr; ij.half_pitch = 0;
                    ij.fi = a.inc_ifilter + level * PSX_GAUSS_ALIGN; ij.ispan = a.inc_ispan[level];
                    const hipError_t e2 = psx_launch_blur_interp(ij, s);
                    if (e2 != hipSuccess) return e2;
                    continue;
                }
                if (level == 0) hipLaunchKernelGGL(k_alt_h_input, g, b, 0, s, img, a.intm, W, H, pitch, taps(a.dd_filter + o * PSX_GAUSS_ALIGN), a.dd_span[o], shift);
                else hipLaunchKernelGGL(k_alt_interp<false>, g, b, 0, s, plane(level - 1), a.intm, W, H, pitch, inci(level), a.inc_ispan[level]);
                hipLaunchKernelGGL(k_alt_interp<true>, g, b, 0, s, a.intm, plane(level), W, H, pitch, inci(level), a.inc_ispan[level]);
            }
        } else if (gm == PSX_GAUSS_VLFEAT_RELATIVE) {
            bool all_fused = P.L >= 4;
            for (int level = 1; level < P.L; level++) all_fused = all_fused && psx_blur_interp_ok(a.inc_ispan[level]);
            static const bool diag = [] { const char* e = getenv("POPSIFT_INTERP_DIAGONAL"); return !(e != nullptr && e[0] == '0'); }();
            for (int level = 0; level < P.L; level++) {
                if (level == 1 && o == 0 && all_fused && diag) {
                    // every remaining level of the frame (and the extrema scans behind the octaves' last levels): diagonal schedule
                    return relative_diagonal(a, s);
                }
                if (level == 0) {
                    if (o == 0) {
                        // the x2 level-0 kernel of pyramid.hip with the interpolated vertical pass where it applies
                        PsxLevel0Args l0;
                        l0.img = a.img; l0.w = a.w; l0.h = a.h; l0.is_float = a.is_float;
                        l0.dst = plane(0); l0.W = W; l0.H = H; l0.pitch = pitch;
                        l0.tmp = a.up; l0.tmp_pitch = a.up_pitch; l0.shift = shift;
                        l0.taps_h = taps(a.dd_filter); l0.span_h = a.dd_span[0]; l0.taps_v = inc(0); l0.span_v = a.inc_span[0];
                        l0.v_ifilter = a.inc_ifilter; l0.v_ispan = a.inc_ispan[0];
                        if (psx_level0_interp_ok(l0)) {
                            const hipError_t e2 = psx_launch_level0(l0, s);
                            if (e2 != hipSuccess) return e2;
                        } else {
                            hipLaunchKernelGGL(k_alt_h_input, g, b, 0, s, img, a.intm, W, H, pitch, taps(a.dd_filter), a.dd_span[0], shift);
                            hipLaunchKernelGGL(k_alt_interp<true>, g, b, 0, s, a.intm, plane(0), W, H, pitch, inci(0), a.inc_ispan[0]);
                        }
                    } else if (!next_l0_done) downscale();
                    next_l0_done = false;
                } else if (psx_blur_interp_ok(a.inc_ispan[level])) {
                    // one fused launch per level (pyramid_interp.hip); level L - 3 also writes level 0 of the next octave
                    const bool feeds = level == P.L - 3 && o + 1 < P.num_octaves;
                    PsxInterpJob ij;
                    ij.src = plane(level - 1); ij.dst = plane(level); ij.W = W; ij.H = H; ij.pitch = pitch;
                    ij.half_dst = feeds ? P.oct[o + 1].data : nullptr; ij.half_pitch = feeds ? P.oct[o + 1].pitch : 0;
                    ij.fi = a.inc_ifilter + level * PSX_GAUSS_ALIGN; ij.ispan = a.inc_ispan[level];
                    const hipError_t e2 = psx_launch_blur_interp(ij, s);
                    if (e2 != hipSuccess) return e2;
                    if (feeds) next_l0_done = true;
                } else {
                    hipLaunchKernelGGL(k_alt_interp<false>, g, b, 0, s, plane(level - 1), a.intm, W, H, pitch, inci(level), a.inc_ispan[level]);
                    hipLaunchKernelGGL(k_alt_interp<true>, g, b, 0, s, a.intm, plane(level), W, H, pitch, inci(level), a.inc_ispan[level]);
                }
            }
        } else if (o == 0 && gm == PSX_GAUSS_VLFEAT_RELATIVE_ALL) {
            for (int level = 0; level < P.L; level++) {
                // every level of octave 0 from the input image with its absolute sigma: level-0 kernels, the level's table
                const PsxTaps f = taps(a.abs0_filter + level * PSX_GAUSS_ALIGN);
                const hipError_t e2 = level_from_input(plane(level), f, a.abs0_span[level], f, a.abs0_span[level]);
                if (e2 != hipSuccess) return e2;
            }
        } else {
            // the default arithmetic for this octave (VLFeat_Relative_All beyond octave 0): the fused kernels of pyramid.hip
            if (o == 0) return hipErrorInvalidValue;       // not reached: the default branch is psx_build_pyramid's own
            downscale();
            for (int level = 1; level < P.L; level++) {
                hipError_t e = psx_launch_blur(plane(level - 1), plane(level), W, H, pitch, inc(level), a.inc_span[level], nullptr, 0, s);
                if (e != hipSuccess) return e;
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (a.after_octave) { e = a.after_octave(a.user, o); if (e != hipSuccess) return e; }
    }
    return hipSuccess;
}
