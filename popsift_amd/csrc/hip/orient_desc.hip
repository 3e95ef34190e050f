// orient_desc.hip -- orientation assignment, orientation scan + feature preparation, and the
// 4x4x8 "loop" descriptor with fused normalisation, for gfx950.
//
// Replaces (behaviour):
//   ori_par            s_orientation.cu:75-259   -> k_orientation   (one wave64 per extremum)
//   ori_prefix_sum     s_orientation.cu:320-362  -> k_scan          (+ prep_features, sift_pyramid.cu:250-280)
//   ext_desc_loop      s_desc_loop.cu:19-158     -> k_descriptors   (one wave64 per descriptor)
//   normalize_histogram s_desc_normalize.h:14-33 -> fused into k_descriptors
//
// wave64 re-design notes:
//   * one launch covers all octaves: each wave looks its extremum up through the per-octave
//     counters, so the host never reads a counter between stages (the reference does four blocking
//     symbol copies and three device-wide syncs per image);
//   * 36-bin histogram: 8 LDS copies per wave (lane & 7) absorb ds_add_f32 conflicts, reduced in
//     fixed order; smoothing and the parabola fit run on lanes 0..35 with __shfl;
//   * the 64-candidate bitonic sort becomes four rounds of wave-max selection;
//   * descriptor: every pixel of the rotated 5x5-SBP window is visited ONCE (gradient, hypot,
//     atan2 once) and scattered into the <= 4 tiles whose |n| < 1 test passes, using the
//     reference's per-tile arithmetic; the reference visits it from each of 16 tile scans.
#include "psx_internal.h"

namespace {

constexpr int NT = 256;
constexpr int WPB = NT / PSX_WAVE;       // waves per block
constexpr int ORI_NBINS = 36;
constexpr int HCOPIES = 8;
// sift_constants.h:21-33: float constants
constexpr float PI_F  = 3.14159265358979323846f;
constexpr float PI2_F = 2.0f * 3.14159265358979323846f;
constexpr float M_4RPI_F = 4.0f / PI_F;
constexpr float ORI_WINFACTOR = 1.5f;
constexpr float DESC_MAGNIFY = 3.0f;

__device__ __forceinline__ void wave_fence()
{
    // LDS traffic of one wave is executed in order; this only stops compiler reordering
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Fixed-point accumulation.  Measured on gfx950 (tools/ubench/lds_atomic.hip): ds_add_f32 is
// serialised per lane (~190 clk per wave-instruction), ds_add_u64 runs ~29x faster.  Histogram
// weights are non-negative, so they are accumulated as unsigned 32.32 fixed point with
// ds_add_u64: integer addition is associative, which also makes the histograms independent of
// the order in which lanes and waves arrive (the reference's shared-memory float atomicAdd,
// s_orientation.cu:159, is order dependent).
typedef unsigned long long fix64;
// A single contribution is grad * weight <= |gradient| <= 255*sqrt(2) < 512, so 9.23 fixed point fits
// one v_cvt_u32_f32; the 64-bit accumulator leaves 2^32 of headroom for the sum.
constexpr float FIX_SCALE = 8388608.0f;           // 2^23
__device__ __forceinline__ fix64 to_fix(float w) { return (fix64)(unsigned)(w * FIX_SCALE); }
__device__ __forceinline__ float from_fix(fix64 v) { return __ull2float_rn(v) * (1.0f / FIX_SCALE); }

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// clamped per-octave extrema count (find_extrema_in_dog's atomicMin, s_extrema.cu:553)
__device__ __forceinline__ int ext_count(const PsxParams* P, const PsxCounters* cnt, int o)
{
    return min(cnt->ext_ct[o], P->max_extrema);
}

// ---------------------------------------------------------------------------------------------
// Orientation
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_orientation(const PsxParams* __restrict__ P, const PsxCounters* cnt)
{
    __shared__ fix64 s_hist[WPB][HCOPIES * ORI_NBINS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    fix64* hist = s_hist[wave];

    int total = 0;
    for (int o = 0; o < P->num_octaves; o++) total += ext_count(P, cnt, o);
    if (total > P->ext_capacity) total = P->ext_capacity;

    const int nwaves = gridDim.x * WPB;
    for (int e = blockIdx.x * WPB + wave; e < total; e += nwaves) {
        int o = 0, base = 0;
        for (;;) {
            const int c = ext_count(P, cnt, o);
            if (e < base + c) break;
            base += c; o++;
        }
        const PsxOctave oc = P->oct[o];
        const int w = oc.w, h = oc.h;
        const psx_iext ie = P->iext[o][P->iext_off[o][e - base]];

        for (int i = lane; i < HCOPIES * ORI_NBINS; i += PSX_WAVE) hist[i] = 0ull;
        wave_fence();

        const float x = ie.xpos, y = ie.ypos;
        const int   level = psx_clampi(ie.lpos, 0, P->L - 1);
        const float sig = ie.sigma;
        const float* plane = oc.data + (size_t)level * oc.plane;

        const float sigw = ORI_WINFACTOR * sig;
        const int   rad  = (int)roundf((3.0f * sigw));
        const float factor = -0.5f / (sigw * sigw);
        const int   sq_thres = rad * rad;

        const int xmin = max(1,     (int)roundf(x) - rad);
        const int xmax = min(w - 2, (int)roundf(x) + rad);
        const int ymin = max(1,     (int)roundf(y) - rad);
        const int ymax = min(h - 2, (int)roundf(y) + rad);
        const int wx = xmax - xmin + 1;
        const int hy = ymax - ymin + 1;
        const int loops = (wx > 0 && hy > 0) ? wx * hy : 0;

        fix64* myhist = hist + (lane & (HCOPIES - 1)) * ORI_NBINS;
        const float rcp_wx = 1.0f / (float)max(wx, 1);
        for (int i = lane; i < loops; i += PSX_WAVE) {
            // i / wx without integer division: (i+0.5)/wx is >= 0.5/wx away from an integer
            const int q = (int)(((float)i + 0.5f) * rcp_wx);
            const int yy = q + ymin;
            const int xx = i - q * wx + xmin;
            const float* p = plane + (size_t)yy * oc.pitch + xx;
            const float gdx = p[1] - p[-1];
            const float gdy = p[oc.pitch] - p[-oc.pitch];
            const float grad  = hypotf(gdx, gdy);
            const float theta = atan2f(gdy, gdx);
            const float dx = xx - x;
            const float dy = yy - y;
            const int sq_dist = (int)(dx * dx + dy * dy);
            if (sq_dist <= sq_thres) {
                const float weight = grad * expf(sq_dist * factor);
                int bidx = (int)roundf((float)ORI_NBINS * (theta + PI_F) / PI2_F);
                bidx = (bidx == ORI_NBINS) ? 0 : bidx;
                atomicAdd(&myhist[bidx], to_fix(weight));
            }
        }
        wave_fence();

        // bins live on lanes 0..35; lanes >= 36 mirror the reference's scratch bins 36..63
        const bool isbin = lane < ORI_NBINS;
        float hval = 0.0f;
        if (isbin) {
            fix64 hsum = 0ull;
#pragma unroll
            for (int c = 0; c < HCOPIES; c++) hsum += hist[c * ORI_NBINS + lane];
            hval = from_fix(hsum);
        }
        const int prev_l = isbin ? (lane == 0 ? ORI_NBINS - 1 : lane - 1) : lane;
        const int next_l = isbin ? (lane == ORI_NBINS - 1 ? 0 : lane + 1) : lane;
#pragma unroll
        for (int it = 0; it < 6; it++) {   // 3 x (hist->sm_hist->hist), s_orientation.cu:166-174
            const float pv = __shfl(hval, prev_l);
            const float nv = __shfl(hval, next_l);
            hval = (pv + hval + nv) / 3.0f;
        }
        const float hp = __shfl(hval, prev_l);
        const float hn = __shfl(hval, next_l);
        bool predicate = isbin && (hval > fmaxf(hp, hn));
        const float num  = predicate ? 3.0f * hp - 4.0f * hval + 1.0f * hn : 0.0f;
        const float denB = predicate ? 2.0f * (hp - 2.0f * hval + hn) : 1.0f;
        const float newbin = num / denB;
        predicate = predicate && newbin >= 0.0f && newbin <= 2.0f;
        const float refined = predicate ? (float)prev_l + newbin : -1.0f;
        const float yval    = predicate ? -(num * num) / (4.0f * denB) + hp : -INFINITY;

        // top-4 by value (BitonicSort::Warp32::sort64 + lanes 0..3, s_orientation.cu:224-247)
        bool alive = true;
        float sel_val[PSX_ORI_MAX], sel_bin[PSX_ORI_MAX];
#pragma unroll
        for (int k = 0; k < PSX_ORI_MAX; k++) {
            const float m = wave_max(alive ? yval : -INFINITY);
            const unsigned long long cand = __ballot(alive && yval == m);
            const int pick = cand ? (__ffsll((long long)cand) - 1) : 0;
            sel_val[k] = m;
            sel_bin[k] = __shfl(refined, pick);
            if (lane == pick) alive = false;
        }
        if (lane == 0) {
            psx_extremum ex;
            const float yval_ref = 0.8f * sel_val[0];
            int angles = 0;
#pragma unroll
            for (int k = 0; k < PSX_ORI_MAX; k++) {
                ex.orientation[k] = 0.0f;
                if (sel_val[k] >= yval_ref) {
                    float chosen_bin = sel_bin[k];
                    if (chosen_bin >= ORI_NBINS) chosen_bin -= ORI_NBINS;
                    ex.orientation[k] = fmaf(PI2_F * chosen_bin, 1.0f / ORI_NBINS, -PI_F);
                    angles++;
                }
            }
            ex.xpos = ie.xpos; ex.ypos = ie.ypos; ex.lpos = ie.lpos; ex.sigma = ie.sigma;
            ex.octave = o; ex.num_ori = angles; ex.idx_ori = 0;
            P->extrema[e] = ex;
            P->ext_nori[e] = angles;
        }
        wave_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// Exclusive scan of num_ori over all extrema (octave-major) -> idx_ori, feat_to_ext map, counters.
// One 1024-thread workgroup; every thread owns a contiguous run of K = ceil(total/1024) extrema
// (sequential partial sum), one block-level scan of the 1024 partial sums (wave __shfl_up + 16
// wave totals), then a second sequential pass writes the results.  Replaces the reference's
// 32x32-thread ExclusivePrefixSum::Block (excl_blk_prefix_sum.h:34-145).
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_NT = 1024;

__device__ __forceinline__ void write_feature(const PsxParams* P, int i, const psx_extremum& ex, int excl)
{
    // prep_features, sift_pyramid.cu:250-280 (descriptor pointers become indices, -1 == nullptr)
    psx_feature f;
    const float s = ldexpf(1.0f, ex.octave - P->up_fac);
    f.debug_octave = ex.octave;
    f.xpos = ex.xpos * s;
    f.ypos = ex.ypos * s;
    f.sigma = ex.sigma * s;
    f.num_ori = ex.num_ori;
#pragma unroll
    for (int k = 0; k < PSX_ORI_MAX; k++) {
        const bool on = k < ex.num_ori;
        f.orientation[k] = on ? ex.orientation[k] : 0.0f;
        f.desc_idx[k] = (on && excl + k < P->ori_capacity) ? excl + k : -1;
    }
    P->features[i] = f;
    if (P->x_features != nullptr && i < P->x_feat_capacity) P->x_features[i] = f;
}

__global__ __launch_bounds__(SCAN_NT) void k_scan(const PsxParams* __restrict__ P, PsxCounters* cnt)
{
    __shared__ int s_wsum[SCAN_NT / PSX_WAVE];
    __shared__ int s_total;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    if (t == 0) {
        int ps = 0;
        for (int o = 0; o < PSX_MAX_OCTAVES; o++) {
            cnt->ext_ps[o] = ps;
            ps += (o < P->num_octaves) ? ext_count(P, cnt, o) : 0;
        }
        cnt->ext_ps[PSX_MAX_OCTAVES] = ps;
        s_total = min(ps, P->ext_capacity);
    }
    __syncthreads();
    const int total = s_total;
    const int cap = P->ori_capacity;
    const int* nori = P->ext_nori;
    // every thread owns SCAN_K consecutive extrema per pass; all 4 int4 loads are in flight at once
    constexpr int SCAN_K = 16;
    int carry = 0;
    for (int base = 0; base < total; base += SCAN_NT * SCAN_K) {
        const int i0 = base + t * SCAN_K;
        int nv[SCAN_K];
#pragma unroll
        for (int q = 0; q < SCAN_K / 4; q++) {
            int4 v4 = make_int4(0, 0, 0, 0);
            if (i0 + 4 * q < total) v4 = reinterpret_cast<const int4*>(nori + i0)[q];   // buffer padded to x16
            nv[4 * q + 0] = v4.x; nv[4 * q + 1] = v4.y; nv[4 * q + 2] = v4.z; nv[4 * q + 3] = v4.w;
        }
        int local = 0;
#pragma unroll
        for (int k = 0; k < SCAN_K; k++) { if (i0 + k >= total) nv[k] = 0; local += nv[k]; }

        int v = local;
#pragma unroll
        for (int off = 1; off < PSX_WAVE; off <<= 1) {
            const int u = __shfl_up(v, off);
            if (lane >= off) v += u;
        }
        __syncthreads();                      // s_wsum free (previous pass finished reading)
        if (lane == PSX_WAVE - 1) s_wsum[wave] = v;
        __syncthreads();
        if (wave == 0) {
            int ws = (lane < SCAN_NT / PSX_WAVE) ? s_wsum[lane] : 0;
#pragma unroll
            for (int off = 1; off < SCAN_NT / PSX_WAVE; off <<= 1) {
                const int u = __shfl_up(ws, off);
                if (lane >= off) ws += u;
            }
            if (lane < SCAN_NT / PSX_WAVE) s_wsum[lane] = ws;   // inclusive over waves
        }
        __syncthreads();
        int excl = carry + v - local + (wave > 0 ? s_wsum[wave - 1] : 0);
        carry += s_wsum[SCAN_NT / PSX_WAVE - 1];
#pragma unroll
        for (int k = 0; k < SCAN_K; k++) {
            const int i = i0 + k;
            if (i < total) {
                const int n = nv[k];
                P->extrema[i].idx_ori = excl;
                for (int q = 0; q < n; q++)
                    if (excl + q < cap) P->feat_to_ext[excl + q] = i;
                if (excl >= cap) {                 // no descriptor wave will visit this extremum
                    psx_extremum ex = P->extrema[i];
                    ex.idx_ori = excl;
                    write_feature(P, i, ex, excl);
                }
                excl += n;
            }
        }
    }
    const int grand = carry;
    __syncthreads();
    if (t == 0) {
        const int ori_total = min(grand, cap);
        cnt->ext_total = total;
        cnt->ori_total = ori_total;
        if (P->x_counts != nullptr) { P->x_counts[0] = total; P->x_counts[1] = ori_total; }
        // per-octave orientation counts (dct.ori_ct / ori_ps, s_orientation.cu:340-360)
        for (int o = 0; o < PSX_MAX_OCTAVES; o++) {
            const int fe = cnt->ext_ps[o];
            cnt->ori_ps[o] = (fe < total) ? min(P->extrema[fe].idx_ori, ori_total) : ori_total;
        }
        cnt->ori_ps[PSX_MAX_OCTAVES] = ori_total;
        for (int o = 0; o < PSX_MAX_OCTAVES; o++) cnt->ori_ct[o] = cnt->ori_ps[o + 1] - cnt->ori_ps[o];
    }
}

// ---------------------------------------------------------------------------------------------
// Descriptor ("loop" mode) + normalisation + Feature record
//
// One wave64 per descriptor.  The wave walks the bounding box of the rotated 5x5-SBP window in
// 16x4-pixel tiles (lane = (x&15, y&3)); tiles with no pixel inside the window are skipped with
// one ballot.  Every pixel inside is visited once: gradient magnitude / angle once, Gaussian
// weight once, then the classic trilinear scatter into <= 2x2 spatial tiles x 2 orientation bins
// with ds_add_f32.  This is the same sum the reference forms by scanning the window from each of
// its 16 tile warps (s_desc_loop.cu:60-124): a pixel contributes to tile (ix,iy) with weight
// (1-|u-ix|)(1-|v-iy|) iff |u-ix|<1 and |v-iy|<1, where (u,v) are the pixel's coordinates in
// tile units; only rounding differs (<= 1e-6 relative per sample; the test tolerance on the
// normalised descriptor is 1e-3).  The reference's fast intrinsics (__expf, __sincosf,
// __fdividef) are matched with gfx950 fast paths here: v_exp_f32, v_rcp_f32, v_sqrt_f32 and a
// degree-13 odd minimax polynomial for atan (max error 3.3e-7 rad).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mn * __frcp_rn(fmaxf(mx, 1e-30f));
    const float s = a * a;
    float r = 0.006811792496591806f;
    r = fmaf(r, s, -0.0336042195558548f);
    r = fmaf(r, s, 0.07962366938591003f);
    r = fmaf(r, s, -0.1323334127664566f);
    r = fmaf(r, s, 0.19807815551757812f);
    r = fmaf(r, s, -0.3331736922264099f);
    r = fmaf(r, s, 0.9999961256980896f);
    r = r * a;
    if (ay > ax) r = 1.57079632679489662f - r;
    if (x < 0.0f) r = PI_F - r;
    return (y < 0.0f) ? -r : r;
}

__global__ __launch_bounds__(NT) void k_descriptors(const PsxParams* __restrict__ P, const PsxCounters* cnt)
{
    // 4 private copies of the 128-bin histogram per wave (lane column & 3): neighbouring pixels of
    // a row fall into the same tile and orientation bin, and same-address atomics serialise.
    // Copy stride 129 entries puts equal bins of different copies on different LDS banks.
    constexpr int DCOPIES = 4, DSTRIDE = 129;
    __shared__ fix64 s_desc[WPB][DCOPIES * DSTRIDE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    fix64* acc = s_desc[wave];
    const int lx = lane & 15, ly = lane >> 4;
    fix64* myacc = acc + (lane & (DCOPIES - 1)) * DSTRIDE;

    const int total = cnt->ori_total;
    const int nwaves = gridDim.x * WPB;
    for (int j = blockIdx.x * WPB + wave; j < total; j += nwaves) {
        const int ext_idx = P->feat_to_ext[j];
        const psx_extremum ex = P->extrema[ext_idx];
        const int ori_num = psx_clampi(j - ex.idx_ori, 0, PSX_ORI_MAX - 1);
        const float ang = ex.orientation[ori_num];
        const PsxOctave oc = P->oct[ex.octave];
        const int width = oc.w, height = oc.h;

        if (ori_num == 0 && lane == 0) write_feature(P, ext_idx, ex, ex.idx_ori);

        for (int i = lane; i < DCOPIES * DSTRIDE; i += PSX_WAVE) acc[i] = 0ull;
        wave_fence();

        const float x = ex.xpos, y = ex.ypos;
        const int   level = psx_clampi(ex.lpos, 0, P->L - 1);
        const float SBP = fabsf(DESC_MAGNIFY * ex.sigma);
        const float* plane = oc.data + (size_t)level * oc.plane;

        if (SBP != 0.0f) {
            const float cos_t = cosf(ang);
            const float sin_t = sinf(ang);
            const float csbp  = cos_t * SBP;
            const float ssbp  = sin_t * SBP;
            const float crsbp = cos_t / SBP;
            const float srsbp = sin_t / SBP;
            const float bsz   = fabsf(csbp) + fabsf(ssbp);

            // union of the 16 tile bounding boxes (s_desc_loop.cu:66-70): extremes at the corner tiles
            int xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -0x7fffffff, ymax = -0x7fffffff;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float ox = (c & 1) ? 1.5f : -1.5f;
                const float oy = (c & 2) ? 1.5f : -1.5f;
                const float ptx = fmaf(csbp, ox, fmaf(-ssbp, oy, x));
                const float pty = fmaf(csbp, oy, fmaf( ssbp, ox, y));
                xmin = min(xmin, (int)floorf(ptx - bsz));
                ymin = min(ymin, (int)floorf(pty - bsz));
                xmax = max(xmax, (int)floorf(ptx + bsz));
                ymax = max(ymax, (int)floorf(pty + bsz));
            }
            xmin = max(1, xmin); ymin = max(1, ymin);
            xmax = min(width - 2, xmax); ymax = min(height - 2, ymax);

            for (int ty = ymin; ty <= ymax; ty += 4) {
                const int ii = ty + ly;
                const float dyk = ii - y;
                const float ub = fmaf(srsbp, dyk, 1.5f);      // u = crsbp*dx + srsbp*dy + 1.5
                const float vb = fmaf(crsbp, dyk, 1.5f);      // v = crsbp*dy - srsbp*dx + 1.5
                const float* prow = plane + (size_t)ii * oc.pitch;
                for (int tx = xmin; tx <= xmax; tx += 16) {
                    const int jj = tx + lx;
                    const float dxk = jj - x;
                    const float u = fmaf(crsbp, dxk, ub);
                    const float v = fmaf(-srsbp, dxk, vb);
                    const bool in = (ii <= ymax) && (jj <= xmax) &&
                                    (u > -1.0f) && (u < 4.0f) && (v > -1.0f) && (v < 4.0f);
                    if (__ballot(in) == 0ull) continue;
                    if (in) {
                        const float* p = prow + jj;
                        const float gdx = p[1] - p[-1];
                        const float gdy = p[oc.pitch] - p[-oc.pitch];
                        const float mod = __fsqrt_rn(fmaf(gdx, gdx, gdy * gdy));
                        float th = fast_atan2(gdy, gdx) - ang;
                        th += (th <  0.0f  ? PI2_F : 0.0f);
                        th -= (th >= PI2_F ? PI2_F : 0.0f);
                        const float tth  = th * M_4RPI_F;
                        const float ffo  = floorf(tth);
                        const int   fo0  = (int)ffo;
                        const float wgt2 = tth - ffo;
                        const float wgt1 = 1.0f - wgt2;
                        const int   fo   = fo0 & 7;
                        const int   fo1  = (fo0 + 1) & 7;

                        const float un = u - 1.5f, vn = v - 1.5f;
                        const float ww = __expf(-0.125f * fmaf(un, un, vn * vn)) * mod;
                        const float fu = floorf(u), fv = floorf(v);
                        const int   ix0 = (int)fu, iy0 = (int)fv;
                        const float ax1 = u - fu, ay1 = v - fv;      // weight of tile ix0+1 / iy0+1
                        const float ax0 = 1.0f - ax1, ay0 = 1.0f - ay1;
#pragma unroll
                        for (int dy = 0; dy < 2; dy++) {
                            const int iy = iy0 + dy;
                            if (iy < 0 || iy > 3) continue;
                            const float wy = (dy ? ay1 : ay0) * ww;
#pragma unroll
                            for (int dx = 0; dx < 2; dx++) {
                                const int ix = ix0 + dx;
                                if (ix < 0 || ix > 3) continue;
                                const float wgt = wy * (dx ? ax1 : ax0);
                                fix64* tb = myacc + ((iy << 2) + ix) * 8;
                                atomicAdd(&tb[fo],  to_fix(wgt1 * wgt));
                                atomicAdd(&tb[fo1], to_fix(wgt2 * wgt));
                            }
                        }
                    }
                }
            }
        }
        wave_fence();

        // normalize_histogram (s_desc_norm_rs.h:42-77 / s_desc_norm_l2.h:86-135); lane owns 2 bins
        fix64 sa = 0ull, sb = 0ull;
#pragma unroll
        for (int c = 0; c < DCOPIES; c++) {
            sa += acc[c * DSTRIDE + 2 * lane];
            sb += acc[c * DSTRIDE + 2 * lane + 1];
        }
        float a = from_fix(sa), b = from_fix(sb);
        if (P->norm_mode == PSX_NORM_ROOTSIFT) {
            const float sum = wave_sum(a + b);
            a = ldexpf(sqrtf(a / sum), P->norm_multi);
            b = ldexpf(sqrtf(b / sum), P->norm_multi);
        } else {
            float norm = sqrtf(wave_sum(a * a + b * b));
            a = fminf(a, 0.2f * norm);
            b = fminf(b, 0.2f * norm);
            norm = wave_sum(a * a + b * b);
            norm = 1.0f / sqrtf(norm);
            norm = ldexpf(norm, P->norm_multi);
            a = a * norm;
            b = b * norm;
        }
        reinterpret_cast<float2*>(P->desc + (size_t)j * 128)[lane] = make_float2(a, b);
        if (P->x_desc != nullptr && j < P->x_desc_capacity)
            reinterpret_cast<float2*>(P->x_desc + (size_t)j * 128)[lane] = make_float2(a, b);
        wave_fence();
    }
}

} // namespace

hipError_t psx_launch_orientation(const PsxParams* d_params, PsxCounters* d_cnt, hipStream_t s)
{
    hipLaunchKernelGGL(k_orientation, dim3(2048), dim3(NT), 0, s, d_params, d_cnt);
    return hipGetLastError();
}

hipError_t psx_launch_scan(const PsxParams* d_params, PsxCounters* d_cnt, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(SCAN_NT), 0, s, d_params, d_cnt);
    return hipGetLastError();
}

hipError_t psx_launch_descriptors(const PsxParams* d_params, const PsxCounters* d_cnt, hipStream_t s)
{
    hipLaunchKernelGGL(k_descriptors, dim3(2048), dim3(NT), 0, s, d_params, d_cnt);
    return hipGetLastError();
}
