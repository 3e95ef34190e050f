// match.hip -- brute-force 2-nearest-neighbour descriptor matcher of MatchingMode.
//
// Reference: FeaturesDev::match -> compute_distance / l2_in_t0 (features.cu:160-225): for every left
// descriptor the two right descriptors with the smallest squared L2 distance, scanned in index order
// with strict '<' (ties keep the earlier index), accept = d1 / d2 < 0.8.  The reference runs one
// 32-thread block per left descriptor and walks the right side serially with a __syncthreads per pair.
//
// The RESULT is never taken from a GEMM form (|a|^2 + |b|^2 - 2ab cancels catastrophically for near
// matches, and the indices are integer output that has to agree with the reference): every pair that can
// matter is evaluated with the reference's own operation tree, so the distances are bit-identical to
// oracle/sift_oracle.c.  Round 5: an f16-MFMA form of that expression with a PROVEN error margin only DISCARDS
// pairs (k_match_mfma below); rounds 1-4 evaluated every pair exactly (k_match_partial, still the path for small
// sets, for POPSIFT_MATCH_MFMA=0 and whenever the prefilter cannot bound its candidates).  The tree:
//   lane t of the reference's warp: q = l[4t..4t+3] - r[4t..4t+3];  p_t = q.x*q.x + q.y*q.y + q.z*q.z + q.w*q.w
//   (contracted left to right: fma(q.w,q.w, fma(q.z,q.z, fma(q.x,q.x, q.y*q.y))))
//   then the shuffle_down tree 16, 8, 4, 2, 1:  a_i = p_i + p_{i+16}, b_i = a_i + a_{i+8}, ... (lane 0's value)
//
// Layout here: one LANE owns one left descriptor (128 floats = 64 packed pairs in VGPRs); a wave walks a
// chunk of the right side, whose values are wave uniform (scalar loads) and enter the packed arithmetic
// as SGPR operands: v_pk_add_f32 (difference), v_pk_mul/fma_f32 (two partial sums per instruction).
// No LDS, no shuffles.  The right side is split into chunks so that ~18k x 18k descriptors fill the chip;
// k_match_merge combines the per-chunk top-2 with the (distance, index) order that the sequential scan
// of the reference produces.
#include "psx_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

struct Top2 { float d1, d2; int i1, i2; };

// two smallest under (distance, index) lexicographic order == result of the reference's sequential scan
__device__ __forceinline__ void top2_insert(Top2& t, float d, int i)
{
    if (d < t.d1) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = i; }
    else if (d < t.d2) { t.d2 = d; t.i2 = i; }
}

__device__ __forceinline__ v2f pk_fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// Descriptors are permuted once so that the two partial sums computed together sit in one register /
// SGPR pair: pair k = c*16 + m (component c of float4 m and of float4 m+16) = (v[4m+c], v[4(m+16)+c]).
__device__ __forceinline__ int perm_src(int k, int half) { return 4 * ((k & 15) + 16 * half) + (k >> 4); }

__global__ void k_match_permute(const float* __restrict__ src, int n, float* __restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;           // one output pair per thread
    if (i >= n * 64) return;
    const int d = i >> 6, k = i & 63;
    const float* s = src + (size_t)d * 128;
    reinterpret_cast<float2*>(dst)[i] = make_float2(s[perm_src(k, 0)], s[perm_src(k, 1)]);
}

// distance of this lane's left descriptor (64 permuted pairs) to the wave-uniform, permuted right
// descriptor r: P_m = (p_m, p_{m+16}) for m = 0..15, two partial sums per packed instruction
__device__ __forceinline__ float l2_tree(const v2f* l, const float* __restrict__ r)
{
    float a[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const v2f rx = {r[2 * m], r[2 * m + 1]},           ry = {r[32 + 2 * m], r[32 + 2 * m + 1]};
        const v2f rz = {r[64 + 2 * m], r[64 + 2 * m + 1]}, rw = {r[96 + 2 * m], r[96 + 2 * m + 1]};
        const v2f x = l[m] - rx, y = l[16 + m] - ry, z = l[32 + m] - rz, w = l[48 + m] - rw;
        v2f p = y * y;
        p = pk_fma2(x, x, p);
        p = pk_fma2(z, z, p);
        p = pk_fma2(w, w, p);
        a[m] = p.x + p.y;                                  // a_m = p_m + p_{m+16}
    }
    float b[8], c[4], d[2];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = a[i] + a[i + 8];
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = b[i] + b[i + 4];
#pragma unroll
    for (int i = 0; i < 2; i++) d[i] = c[i] + c[i + 2];
    return d[0] + d[1];
}

// grid (ceil(l_len/64), nchunks), 64 threads.  left: original layout; right_p: permuted copy.
// partial[(chunk * l_len + left)] = top-2 within the chunk.
__global__ __launch_bounds__(64) void k_match_partial(const float* __restrict__ left, int l_len,
                                                      const float* __restrict__ right_p, int r_len,
                                                      int chunk_len, Top2* __restrict__ partial)
{
    const int li = blockIdx.x * 64 + threadIdx.x;
    const int lq = min(li, l_len - 1);                       // idle lanes repeat the last descriptor
    v2f l[64];
    {
        const float* lp = left + (size_t)lq * 128;
#pragma unroll
        for (int k = 0; k < 64; k++) l[k] = (v2f){lp[perm_src(k, 0)], lp[perm_src(k, 1)]};
    }
    const int r0 = blockIdx.y * chunk_len;
    const int r1 = min(r0 + chunk_len, r_len);
    Top2 t = {INFINITY, INFINITY, 0, 0};
    for (int i = r0; i < r1; i++) {
        const float d = l2_tree(l, right_p + (size_t)i * 128);
        top2_insert(t, d, i);
    }
    if (li < l_len) partial[(size_t)blockIdx.y * l_len + li] = t;
}

// out[3*l + 0..2] = (best, second, accept) (int3 match_matrix, features.cu:218-222), dist[2*l + 0..1]
__global__ void k_match_merge(const Top2* __restrict__ partial, int l_len, int nchunks, int r_len,
                              int* __restrict__ out, float* __restrict__ dist)
{
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= l_len) return;
    Top2 t = {INFINITY, INFINITY, 0, 0};
    // chunks are visited in index order and a chunk's own entries are already in (distance, index)
    // order, so strict '<' insertion reproduces the sequential scan: ties keep the smaller index
    for (int c = 0; c < nchunks; c++) {
        const Top2 p = partial[(size_t)c * l_len + li];
        if (p.d1 < t.d1) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = p.d1; t.i1 = p.i1; }
        else if (p.d1 < t.d2) { t.d2 = p.d1; t.i2 = p.i1; }
        if (p.d2 < t.d1) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = p.d2; t.i1 = p.i2; }
        else if (p.d2 < t.d2) { t.d2 = p.d2; t.i2 = p.i2; }
    }
    const bool accept = (t.d1 / t.d2 < 0.8f);
    out[3 * li + 0] = t.i1; out[3 * li + 1] = t.i2; out[3 * li + 2] = accept ? 1 : 0;
    if (dist) { dist[2 * li + 0] = t.d1; dist[2 * li + 1] = t.d2; }
}


// ---------------------------------------------------------------------------------------------------------------------
// MFMA prefilter (round 5).  The exact scan above costs 165 lane instructions per pair and runs at ~1/3 of the VALU issue
// peak.  The RESULT must stay the reference's -- integer indices, bit-identical distances -- so nothing approximate may
// decide anything; but an approximate distance with a PROVEN error bound can discard almost every pair:
//   s(l, r) = |l|^2 + |r|^2 - 2 <f16(l), f16(r)>          (v_mfma_f32_32x32x16_f16, f32 accumulation)
//   |s - d| <= E(l, r),   d = the reference's float distance of the pair:
//       f16 keeps 11 significant bits (round to nearest even): an element >= 2^-14 in magnitude has relative error <= u = 2^-11,
//       a smaller one absolute error <= 2^-25, so |<f16 l, f16 r> - <l, r>| <= 2.002 u |l| |r| + 2^-25 sqrt(128) (|l| + |r|)
//       (Cauchy-Schwarz); the f32 accumulation of 128 exact products, the f32 norms and the rounding of the reference's own
//       operation tree add <= 2e-5 (|l|^2 + |r|^2) (round 6: the accumulation STARTS at |r|^2 / fneg2 -- an exact scaling of the
//       norm -- instead of adding it last: 129 terms instead of 128 in the same f32 chain, (129 / 2^24) (|l| |r| + |r|^2 / 2) <
//       8e-6 (|l|^2 + |r|^2), inside the same term).  With Rmax = max |r|:
//       E_l := 0.00197 |l| Rmax + 2e-7 M (|l| + Rmax) + 4e-5 (|l|^2 + Rmax^2)  (bf16, 8 bits, had 0.0157: on random descriptors,
//       whose distances concentrate, hundreds of neighbours fell inside the margin).
//       f16's RANGE is taken out of the picture by scaling both sides with a power of two 2^k (exact) before the conversion,
//       k = floor(log2(16384 / M)), M = the largest norm of either side: no element exceeds 16384 (f16 max 65504), and an
//       element that falls below 2^-14 after scaling -- where f16 loses relative precision, or the matrix unit may flush
//       it to zero -- is below M 2^-27, which is the "2e-7 M (|l| + Rmax)" term (with either behaviour).  -2 / 4^k rides on the
//       epilogue's fma.  Norms that are zero, infinite or NaN switch the prefilter off (overflow flag -> exact scan).
// If s2 is the second smallest s of a left descriptor, its true second-best distance is <= s2 + E_l (two pairs have
// d <= s + E_l <= s2 + E_l), and every pair with d <= that has s <= s2 + 2 E_l.  So the set {r : s <= s2 + 2 E_l} contains
// every pair that can be best or second best INCLUDING all ties, and the reference's scan restricted to it (same operation
// tree, (distance, index) order) returns the reference's answer.  A running s2 (per half wave, per chunk of the right
// side, started from a seeding pass over the first 2048 right descriptors) is >= the final one: the candidate set only
// grows.  A lane appends its candidates to ITS segment of the left descriptor's list (no atomics); a segment that overflows
// (thousands of near-equal neighbours: duplicates, constant descriptors) sends the whole call through the exact scan of
// every pair.  k_match_exact evaluates the candidates in the reference's own 32-threads-per-pair shape, one wave per left
// descriptor.  18 432 x 18 432 unit-norm descriptors: 26 candidates per left descriptor on average, 3.8 ms -> 0.34 ms of
// kernel time (profiles/r05_match_prefilter.txt).
// MFMA operand layout: only "lane l supplies row / column l & 31 and the k-slots of half l >> 5" is used -- the SAME 16
// bytes of a descriptor go into the same operand slot on both sides, so whatever k the hardware assigns to a slot, the
// products pair up (the sum over k is what matters); C/D: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_SEGS = 32;                  // candidate segments per left descriptor: one per (chunk of the right side, half wave) ...
constexpr int MF_SEGCAP = 32;                // ... of this many slots: private to one lane of one workgroup, so appending needs no atomic
constexpr int MF_CAP = MF_SEGS * MF_SEGCAP;  // (a returned global atomic per candidate made the scan wait ~1 us per tile and column set)
constexpr int MF_TILE = 32;                  // right descriptors per MFMA tile
constexpr int MF_SUB = 2;                    // MFMA tiles per staged block (= per barrier)
constexpr int MF_ROW = 272;                  // bytes per staged right descriptor: 256 + 16 (rows 4 banks apart: conflict-free b128 reads)
constexpr int MF_SEED = 2048;                // right descriptors of the seeding pass ...
constexpr int MF_SEEDCH = 8;                 // ... in this many chunks (workgroups per 256 left descriptors)
constexpr int MF_MAXSLOTS = 64;              // the largest squared norm of the right side is kept in 64 slots (same-address atomics serialise)

__device__ __forceinline__ unsigned short to_f16(float f)
{
    const _Float16 h = (_Float16)f;                                                // v_cvt_f16_f32: round to nearest even
    unsigned short b; __builtin_memcpy(&b, &h, 2);
    return b;
}

// squared norm of every descriptor and the largest one (bits of a non-negative float, 64 slots: same-address atomics serialise)
// (both sides in one launch: blocks [0, blocks_a) take the first set, the others the second)
__global__ void k_match_norms(const float* __restrict__ src_a, int n_a, float* __restrict__ norm2_a, unsigned* __restrict__ maxbits_a, int blocks_a,
                              const float* __restrict__ src_b, int n_b, float* __restrict__ norm2_b, unsigned* __restrict__ maxbits_b)
{
    const bool second = (int)blockIdx.x >= blocks_a;
    const float* src = second ? src_b : src_a;
    const int n = second ? n_b : n_a;
    float* norm2 = second ? norm2_b : norm2_a;
    unsigned* maxbits = second ? maxbits_b : maxbits_a;
    const int bid = second ? (int)blockIdx.x - blocks_a : (int)blockIdx.x;
    const int nblk = second ? (int)gridDim.x - blocks_a : blocks_a;
    const int q = threadIdx.x & 31;
    // 32 threads per descriptor (4 elements each), a block's 8 groups walk the set with the stride of all groups of this side
    float mx = 0.0f;
    for (int d = bid * 8 + (threadIdx.x >> 5); d < n; d += nblk * 8) {
        const float4 v = reinterpret_cast<const float4*>(src + (size_t)d * 128)[q];
        float ss = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
        for (int k = 16; k >= 1; k >>= 1) ss += __shfl_xor(ss, k, 32);
        if (q == 0) norm2[d] = ss;
        // a NaN norm has the bit pattern of a huge unsigned: it wins the maximum and switches the prefilter off (k_match_cvt)
        mx = __uint_as_float(max(__float_as_uint(mx), __float_as_uint(ss < 0.0f ? 0.0f : ss)));
    }
    __shared__ unsigned s_max;
    if (threadIdx.x == 0) s_max = 0u;
    __syncthreads();
    if (q == 0) atomicMax(&s_max, __float_as_uint(mx));
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&maxbits[bid & (MF_MAXSLOTS - 1)], s_max);
}

// par[0] = 2^k, par[1] = -2 / 4^k, par[2] = Rmax^2, par[3] = M; *bad = 1 when the prefilter cannot be used
__device__ __forceinline__ void match_scale(float l2, float r2, float* par, int* bad)
{
    const float m2 = fmaxf(l2, r2);
    // finite, positive, and far enough from both ends of the float range for the squares below
    if (!(l2 == l2) || !(r2 == r2) || !(m2 > 1e-30f) || !(m2 < 1e30f)) { *bad = 1; par[0] = 1.0f; par[1] = -2.0f; par[2] = 0.0f; par[3] = 0.0f; return; }
    const float M = sqrtf(m2);
    const int k = (int)floorf(log2f(16384.0f / M));
    *bad = 0;
    par[0] = ldexpf(1.0f, k);
    par[1] = ldexpf(-2.0f, -2 * k);
    par[2] = r2;
    par[3] = M;
}

// f16 copy of every descriptor of both sides, scaled by 2^k (blocks [0, blocks_a): the first set).  Every block derives the
// scale from the 2 x 64 maximum slots itself (a 1-thread kernel in between cost a 5 us launch); block 0 publishes the
// parameters and the flag for the kernels behind it.  a = right, b = left.
__global__ void k_match_cvt(const float* __restrict__ src_a, int n_a, unsigned short* __restrict__ dst_a, int blocks_a,
                            const float* __restrict__ src_b, int n_b, unsigned short* __restrict__ dst_b,
                            const unsigned* __restrict__ lmax, const unsigned* __restrict__ rmax, float* __restrict__ par, int* __restrict__ flag)
{
    __shared__ float s_par[4];
    if (threadIdx.x < 64) {
        unsigned lb = lmax[threadIdx.x], rb = rmax[threadIdx.x];
        for (int k = 32; k >= 1; k >>= 1) { lb = max(lb, (unsigned)__shfl_xor((int)lb, k)); rb = max(rb, (unsigned)__shfl_xor((int)rb, k)); }
        if (threadIdx.x == 0) {
            float pp[4]; int bad;
            match_scale(__uint_as_float(lb), __uint_as_float(rb), pp, &bad);
            s_par[0] = pp[0];
            if (blockIdx.x == 0) { par[0] = pp[0]; par[1] = pp[1]; par[2] = pp[2]; par[3] = pp[3]; *flag = bad; }   // the call's first write of the flag
        }
    }
    __syncthreads();
    const bool second = (int)blockIdx.x >= blocks_a;
    const float* src = second ? src_b : src_a;
    unsigned short* dst = second ? dst_b : dst_a;
    const int n = second ? n_b : n_a;
    const int g = ((int)blockIdx.x - (second ? blocks_a : 0)) * blockDim.x + threadIdx.x;          // one thread per 4 elements
    if (g >= n * 32) return;
    const float sc = s_par[0];
    const float4 v = reinterpret_cast<const float4*>(src)[g];
    ushort4 o; o.x = to_f16(v.x * sc); o.y = to_f16(v.y * sc); o.z = to_f16(v.z * sc); o.w = to_f16(v.w * sc);
    reinterpret_cast<ushort4*>(dst)[g] = o;
}

// grid (ceil(l_len / 256), nchunks), 256 threads: wave w owns left descriptors [256 bx + 64 w, + 64) as two B fragment
// sets (32 columns each); the block stages tiles of 32 right descriptors in LDS (double buffered) and every wave runs
// 2 x 8 MFMAs per tile, then the epilogue on its 2 x 16 accumulators per lane.
//   SEED = true : the first MF_SEED right descriptors in MF_SEEDCH chunks; no candidates, the result is every chunk's (smallest,
//                 second smallest) s' per left descriptor: the second smallest of their union is an upper bound of the final
//                 one and lets every chunk of the real pass start tight (an empty running minimum makes the first rows of
//                 every chunk candidates: ~26 per chunk and descriptor)
//   SEED = false: every chunk of the right side; running minima start at seed[l]; candidates appended to the left's list
template <bool SEED>
__global__ __launch_bounds__(256, 2) void k_match_mfma(const unsigned short* __restrict__ lh, const float* __restrict__ ln2, int l_len,
                                                       const unsigned short* __restrict__ rh, const float* __restrict__ rn2, int r_len,
                                                       int chunk_len, const float* __restrict__ par, float* __restrict__ seed,
                                                       int* __restrict__ cand_ct, int* __restrict__ cand)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_r[2][MF_SUB * MF_TILE * MF_ROW];
    __shared__ __attribute__((aligned(16))) float s_n[2][MF_SUB * MF_TILE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int r0 = blockIdx.y * chunk_len, r1 = min(r0 + chunk_len, r_len);
    const int ntiles = (r1 - r0 + MF_SUB * MF_TILE - 1) / (MF_SUB * MF_TILE);      // staged blocks of MF_SUB tiles
    const float fneg2 = par[1], rmax2 = par[2], bigM = par[3];
    const float rmax = sqrtf(rmax2);
    // Round 6: the epilogue works on D = s' / fneg2 = <f16 l, f16 r> + |r|^2 / fneg2 instead of s' = |r|^2 + fneg2 <l, r>.  fneg2 = -2 / 4^k is
    // a negative power of two times 2: dividing by it is EXACT and reverses the order, so "s' <= threshold" is "D >= threshold /
    // fneg2" with the same bits on both sides; the norm term enters as the INITIAL VALUE of the MFMA accumulator (the
    // staged norms are pre-scaled), which removes the fma per pair, and the test of a whole tile column is one maximum.
    const float inv = 1.0f / fneg2;

    // ---- this wave's left descriptors: B operands (8 K-steps x 2 column sets), norms, margins ----
    f16x8 bfrag[2][8];
    float twoE[2], m1[2], m2[2];
    int lidx[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int li = blockIdx.x * 256 + wave * 64 + c * 32 + col;
        lidx[c] = li;
        const int lq = min(li, l_len - 1);
        const unsigned short* lp = lh + (size_t)lq * 128 + half * 8;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) bfrag[c][kk] = *reinterpret_cast<const f16x8*>(lp + kk * 16);
        const float nl = ln2[lq];
        const float E = 0.00197f * sqrtf(nl) * rmax + 2e-7f * bigM * (sqrtf(nl) + rmax) + 4e-5f * (nl + rmax2);
        twoE[c] = 2.0f * E * -inv;                      // the margin in D units (-inv > 0, a power of two: exact)
        // running LARGEST / second largest of D (= smallest / second smallest of s' = |r|^2 - 2 <l, r>; |l|^2 is the same for every
        // pair of a column).  The real pass starts both at the sample's second smallest s': "two values <= S exist" is all a
        // threshold needs.
        if (SEED) m1[c] = m2[c] = -INFINITY;
        else {
            // second smallest of the seeding pass' MF_SEEDCH (smallest, second smallest) pairs
            float a1 = INFINITY, a2 = INFINITY;
#pragma unroll
            for (int q = 0; q < 2 * MF_SEEDCH; q++) {
                const float v = seed[((size_t)(q >> 1) * l_len + lq) * 2 + (q & 1)];
                a2 = fminf(a2, fmaxf(a1, v)); a1 = fminf(a1, v);
            }
            m1[c] = m2[c] = a2 * inv;
        }
    }
    int cnt[2] = {0, 0};                         // candidates of this lane's segment = (left descriptor, chunk, half wave)
    const int seg = blockIdx.y * 2 + half;

    // ---- staging of a block of MF_SUB right tiles: 64 descriptors x 256 bytes = 1024 16-byte pieces, four per thread; norms by 64
    // threads.  Two steps, so that the global loads of block k + 2 are in flight while block k is computed: issue() = loads into
    // registers; commit() = registers into the LDS buffer of block k + 1 (free since the barrier that ended iteration k - 1).
    // One barrier per TWO tiles (round 6: the waves of a workgroup take different times per tile -- half of them run the
    // candidate path -- and met at a barrier after every 16 MFMAs) ----
    // (named register sets and macros, not an array captured by lambdas: the array stayed in scratch memory, and the
    // scratch store behind each global load waited for it -- the loads were not in flight at all)
    static_assert(MF_SUB == 2, "four staging loads per thread");
    uint4 pre0 = make_uint4(0u, 0u, 0u, 0u), pre1 = make_uint4(0u, 0u, 0u, 0u), pre2 = make_uint4(0u, 0u, 0u, 0u), pre3 = make_uint4(0u, 0u, 0u, 0u);
    float pre_n = INFINITY;
    const int st_row0 = t >> 4, st_c16 = t & 15;             // rows st_row0 + 16 i, i = 0..3
#define MF_LOAD_(i_) (*reinterpret_cast<const uint4*>(rh + (size_t)min(base_ + st_row0 + 16 * (i_), r_len - 1) * 128 + st_c16 * 8))
#define MF_ISSUE(tile_) do { \
        const int base_ = r0 + (tile_) * (MF_SUB * MF_TILE); \
        pre0 = MF_LOAD_(0); pre1 = MF_LOAD_(1); pre2 = MF_LOAD_(2); pre3 = MF_LOAD_(3); \
        if (t < MF_SUB * MF_TILE) pre_n = (base_ + t < r1) ? rn2[min(base_ + t, r_len - 1)] : INFINITY;     /* rows beyond the chunk never win */ \
    } while (0)
#define MF_COMMIT(buf_) do { \
        *reinterpret_cast<uint4*>(&s_r[buf_][(st_row0 +  0) * MF_ROW + st_c16 * 16]) = pre0; \
        *reinterpret_cast<uint4*>(&s_r[buf_][(st_row0 + 16) * MF_ROW + st_c16 * 16]) = pre1; \
        *reinterpret_cast<uint4*>(&s_r[buf_][(st_row0 + 32) * MF_ROW + st_c16 * 16]) = pre2; \
        *reinterpret_cast<uint4*>(&s_r[buf_][(st_row0 + 48) * MF_ROW + st_c16 * 16]) = pre3; \
        if (t < MF_SUB * MF_TILE) s_n[buf_][t] = pre_n * inv;      /* +inf (rows beyond the chunk) becomes -inf: never a maximum */ \
    } while (0)

    if (ntiles > 0) { MF_ISSUE(0); MF_COMMIT(0); }
    if (ntiles > 1) MF_ISSUE(1);
    __syncthreads();
    for (int tile = 0; tile < ntiles; tile++) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) MF_COMMIT(buf ^ 1);       // loaded during the previous iteration
        if (tile + 2 < ntiles) MF_ISSUE(tile + 2);
#pragma unroll
        for (int sub = 0; sub < MF_SUB; sub++) {
        // accumulators start at |r|^2 / fneg2 of their row (C/D layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
        f32x16 acc[2];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 nr = *reinterpret_cast<const f32x4*>(&s_n[buf][sub * MF_TILE + 8 * g + 4 * half]);
#pragma unroll
            for (int e = 0; e < 4; e++) { acc[0][4 * g + e] = nr[e]; acc[1][4 * g + e] = nr[e]; }
        }
        const unsigned char* rp = &s_r[buf][(sub * MF_TILE + col) * MF_ROW + half * 16];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const f16x8 a = *reinterpret_cast<const f16x8*>(rp + kk * 32);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfrag[0][kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfrag[1][kk], acc[1], 0, 0, 0);
        }
        const int base = r0 + (tile * MF_SUB + sub) * MF_TILE;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            // The whole tile column against the threshold the column had BEFORE this tile (a superset test): one maximum of the
            // lane's 16 values (v_max3_f32) and one compare -- the running maxima are updated only from values that pass it (a
            // value below the threshold is below the second largest: it cannot change either of the two).  Rounds 5's form
            // spent an fma, a compare and two v_med3 on every pair: 212 VALU instructions per 16 MFMAs.
            if (SEED) {
                // the seeding pass wants the two largest of everything it sees: no threshold, every value updates them
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    m2[c] = __builtin_amdgcn_fmed3f(m1[c], m2[c], acc[c][reg]);
                    m1[c] = fmaxf(m1[c], acc[c][reg]);
                }
                continue;
            }
            const float thr = m2[c] - twoE[c];
            // maxima of the four register groups (4 values each), then of the lane's 16 values
            float gm[4];
#pragma unroll
            for (int g = 0; g < 4; g++) gm[g] = fmaxf(fmaxf(fmaxf(acc[c][4 * g], acc[c][4 * g + 1]), acc[c][4 * g + 2]), acc[c][4 * g + 3]);
            const float mx = fmaxf(fmaxf(fmaxf(gm[0], gm[1]), gm[2]), gm[3]);
            const bool any = mx >= thr;
            m2[c] = __builtin_amdgcn_fmed3f(m1[c], m2[c], mx);
            m1[c] = fmaxf(m1[c], mx);
            if (__ballot(any) != 0ull) {
                if (any && lidx[c] < l_len) {
                    // which of the 16 values pass: one bit per value (bit 15 - reg); a group of four is looked at only when some lane's
                    // group maximum passes (one wave-uniform branch per group instead of a compare per value)
                    unsigned hits = 0u;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        if (__ballot(gm[g] >= thr) != 0ull) {
#pragma unroll
                            for (int e = 0; e < 4; e++) hits |= (acc[c][4 * g + e] >= thr ? 1u : 0u) << (15 - (4 * g + e));
                        }
                    }
                    while (hits != 0u) {
                        const int b = 31 - __builtin_clz(hits);
                        hits &= ~(1u << b);
                        const int reg = 15 - b;
                        const int ridx = base + 8 * (reg >> 2) + 4 * half + (reg & 3);
                        if (ridx < r1) {
                            if (cnt[c] < MF_SEGCAP) cand[((size_t)lidx[c] * MF_SEGS + seg) * MF_SEGCAP + cnt[c]] = ridx;
                            cnt[c]++;
                        }
                    }
                }
            }
        }
        }   // sub
        __syncthreads();
    }
    if (!SEED) {
#pragma unroll
        for (int c = 0; c < 2; c++)
            if (lidx[c] < l_len) cand_ct[(size_t)lidx[c] * MF_SEGS + seg] = cnt[c];
    }
    if (SEED) {
        // the two half waves saw different rows of the same columns: second smallest of the union
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float o1 = __shfl_xor(m1[c], 32), o2 = __shfl_xor(m2[c], 32);
            const float d1 = fmaxf(m1[c], o1), d2 = fmaxf(fminf(m1[c], o1), fmaxf(m2[c], o2));     // largest, second largest D
            if (half == 0 && lidx[c] < l_len) {
                float* sd = seed + ((size_t)blockIdx.y * l_len + lidx[c]) * 2;
                sd[0] = d1 * fneg2; sd[1] = d2 * fneg2;                                           // back to s' (exact)
            }
        }
    }
}

#undef MF_ISSUE
#undef MF_COMMIT
#undef MF_LOAD_

// (distance, index) lexicographic order: what the reference's sequential scan with strict '<' yields
__device__ __forceinline__ bool lex_less(float d, int i, float e, int j) { return d < e || (d == e && i < j); }
__device__ __forceinline__ void top2_insert_lex(Top2& t, float d, int i)
{
    if (!(d == d)) return;                                   // NaN never passes the reference's '<'
    if (lex_less(d, i, t.d1, t.i1)) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = i; }
    else if (lex_less(d, i, t.d2, t.i2)) { t.d2 = d; t.i2 = i; }
}

// Exact distances of the candidates, one wave per left descriptor, one HALF WAVE per pair -- the reference's own shape
// (l2_in_t0, features.cu:160-189: 32 threads, thread t takes floats 4t..4t+3, then shuffle_down 16, 8, 4, 2, 1): one
// coalesced 512-byte read per pair instead of 64 scattered ones per lane, 10 registers instead of 256.  Lane t of a half:
//   q = l4 - r4;  p_t = fma(q.w, q.w, fma(q.z, q.z, fma(q.x, q.x, q.y * q.y)));  then v += shfl_down(v, 16 / 8 / 4 / 2 / 1):
// lane 0 ends with ((((p0 + p16) + (p8 + p24)) + ...)), the tree of l2_tree above.  `left` / `right` in ORIGINAL layout.
__global__ __launch_bounds__(256) void k_match_exact(const float* __restrict__ left, int l_len, const float* __restrict__ right, int r_len,
                                                     int* cand_ct, const int* __restrict__ cand,
                                                     int* __restrict__ out, float* __restrict__ dist, int* __restrict__ flag,
                                                     unsigned* __restrict__ maxslots, int tidy)
{
    // the last kernel of a call leaves the prefilter's counters as the next call needs them (two memsets per call, one of them
    // 4.7 MB, were ~10 us of a 0.25 ms call): the 2 x MF_MAXSLOTS maximum slots here, every candidate count below once it is read
    if (tidy && blockIdx.x == 0 && threadIdx.x < 2 * MF_MAXSLOTS) maxslots[threadIdx.x] = 0u;
    // launched behind the prefilter without waiting for its verdict: a raised flag (a candidate segment overflowed, norms the
    // margin cannot bound) means the host will run the exact scan of every pair instead
    if (*flag != 0) return;
    const int lane = threadIdx.x & 63, tl = lane & 31, half = lane >> 5;
    const int li = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (li >= l_len) return;
    const float4 l4 = reinterpret_cast<const float4*>(left + (size_t)li * 128)[tl];
    // the left descriptor's segments, compacted into one list in LDS: lane s < MF_SEGS owns segment s
    __shared__ int s_list[4][MF_CAP];
    int* cl = s_list[threadIdx.x >> 6];
    const int rawct = lane < MF_SEGS ? cand_ct[(size_t)li * MF_SEGS + lane] : 0;
    if (tidy && lane < MF_SEGS && rawct != 0) cand_ct[(size_t)li * MF_SEGS + lane] = 0;
    // a segment that overflowed (the prefilter counted more candidates than it could store): the whole call goes through the
    // exact scan of every pair -- the host reads the flag with the results (a separate kernel for this test cost a 5 us launch)
    if (rawct > MF_SEGCAP) *flag = 1;
    int myct = min(rawct, MF_SEGCAP);
    int off = myct;                                          // inclusive prefix sum over the 64 lanes
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) { const int o = __shfl_up(off, k); if (lane >= k) off += o; }
    const int n = __shfl(off, 63);
    off -= myct;
    for (int k = 0; k < myct; k++) cl[off + k] = cand[((size_t)li * MF_SEGS + lane) * MF_SEGCAP + k];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the wave reads what its own lanes wrote
    __builtin_amdgcn_wave_barrier();
    // the sentinel index is larger than any real one: an empty slot loses every tie
    Top2 t = {INFINITY, INFINITY, 0x7fffffff, 0x7fffffff};
    // 2 x EX_K candidates per round (EX_K per half wave): all 512-byte reads of a half are in flight before the first reduction
    // (one at a time, every round waited for its own L2 round trip: 47 us for ~20 candidates per descriptor; two: 37 us)
    constexpr int EX_K = 2;                                 // 4 measured the same (37 us): past two, the insert chain is the limit
    for (int c0 = 0; c0 < n; c0 += 2 * EX_K) {
        int ri[EX_K];
        float4 r4[EX_K];
#pragma unroll
        for (int k = 0; k < EX_K; ++k) {
            const int c = c0 + 2 * k + half;
            ri[k] = c < n ? cl[c] : -1;
            r4[k] = reinterpret_cast<const float4*>(right + (size_t)(ri[k] < 0 ? 0 : ri[k]) * 128)[tl];
        }
#pragma unroll
        for (int k = 0; k < EX_K; ++k) {
            const float qx = l4.x - r4[k].x, qy = l4.y - r4[k].y, qz = l4.z - r4[k].z, qw = l4.w - r4[k].w;
            float v = fmaf(qw, qw, fmaf(qz, qz, fmaf(qx, qx, qy * qy)));
            v += __shfl_down(v, 16, 32); v += __shfl_down(v, 8, 32); v += __shfl_down(v, 4, 32);
            v += __shfl_down(v, 2, 32);  v += __shfl_down(v, 1, 32);
            const float d = __shfl(v, 0, 32);                // the half's distance, in all of its lanes
            if (ri[k] >= 0) top2_insert_lex(t, d, ri[k]);
        }
    }
    {   // the two halves hold disjoint candidate subsets: merge
        const float od1 = __shfl_xor(t.d1, 32), od2 = __shfl_xor(t.d2, 32);
        const int oi1 = __shfl_xor(t.i1, 32), oi2 = __shfl_xor(t.i2, 32);
        top2_insert_lex(t, od1, oi1);
        top2_insert_lex(t, od2, oi2);
    }
    if (lane == 0) {
        // the reference starts from (inf, inf, index 0, index 0) and never replaces an entry by an equal one
        const int i1 = t.i1 == 0x7fffffff ? 0 : t.i1, i2 = t.i2 == 0x7fffffff ? 0 : t.i2;
        const bool accept = (t.d1 / t.d2 < 0.8f);
        out[3 * li + 0] = i1; out[3 * li + 1] = i2; out[3 * li + 2] = accept ? 1 : 0;
        if (dist) { dist[2 * li + 0] = t.d1; dist[2 * li + 1] = t.d2; }
    }
}

} // namespace

// Scratch of one calling thread: a private non-blocking stream (no null-stream launch, so nothing else on
// the device is synchronised) and buffers that only ever grow.  Freed when the thread exits.
namespace {
struct MatchScratch {
    int device = -1;
    hipStream_t stream = nullptr;
    void* buf[10] = {};
    size_t cap[10] = {};
    void* hpin = nullptr;                // pinned staging of the results: a DMA into pageable caller memory goes through the
    size_t hpin_cap = 0;                 // runtime's own bounce buffers, ~0.5 ms per call on this stack
    bool tidy = false;                   // the prefilter's counters are zero (left so by the previous call's last kernel)
    void release()
    {
        if (device < 0) return;
        int cur = -1;                                        // the caller's current device is left as it was
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        (void)hipSetDevice(device);
        for (int i = 0; i < 10; i++) { (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0; }
        tidy = false;
        if (hpin) (void)hipHostFree(hpin);
        hpin = nullptr; hpin_cap = 0;
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr; device = -1;
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    bool need_pinned(size_t bytes)
    {
        if (bytes <= hpin_cap && hpin) return true;
        if (hpin) (void)hipHostFree(hpin);
        hpin = nullptr; hpin_cap = 0;
        if (hipHostMalloc(&hpin, bytes, hipHostMallocDefault) != hipSuccess) return false;
        hpin_cap = bytes;
        return true;
    }
    bool need(int i, size_t bytes)
    {
        if (bytes <= cap[i] && buf[i]) return true;
        (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0;
        tidy = false;
        if (hipMalloc(&buf[i], bytes) != hipSuccess) return false;
        cap[i] = bytes;
        return true;
    }
    // Thread exit.  For the main thread that is process teardown, where the HIP runtime may already be gone: ask it
    // first (a finalised runtime answers with an error) and then leave the buffers to the process exit.
    ~MatchScratch() { int n = 0; if (device >= 0 && hipGetDeviceCount(&n) == hipSuccess && n > 0) release(); }
};
thread_local MatchScratch t_scratch;
} // namespace

// frees the calling thread's matcher scratch (stream + up to 10 device buffers); an explicit user call -- PopSift::uninit
// does NOT call it: the scratch belongs to the thread, not to one PopSift object (another replica may be using it)
extern "C" int psx_match_release(void)
{
    t_scratch.release();
    return PSX_OK;
}

extern "C" int psx_match(int device, const float* d_left, int l_len, const float* d_right, int r_len,
                         int* host_match, float* host_dist)
{
    if (l_len < 0 || r_len < 0 || (l_len > 0 && (!d_left || !host_match)) || (r_len > 0 && !d_right))
        return PSX_ERR_INVALID;
    if (l_len == 0) return PSX_OK;
    if (hipSetDevice(device) != hipSuccess) return PSX_ERR_HIP;
    MatchScratch& sc = t_scratch;
    if (sc.device != device) {
        sc.release();
        if (hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking) != hipSuccess) return PSX_ERR_HIP;
        sc.device = device;
    }
    // enough (left group, chunk) waves to fill the chip: 256 CUs x 4 SIMDs x 2 waves
    const int lgroups = (l_len + 63) / 64;
    int nchunks = (2048 + lgroups - 1) / lgroups;
    if (nchunks > (r_len + 63) / 64) nchunks = (r_len + 63) / 64;      // at least 64 right descriptors per chunk
    if (nchunks < 1) nchunks = 1;
    const int chunk_len = r_len > 0 ? (r_len + nchunks - 1) / nchunks : 1;
    if (r_len > 0) nchunks = (r_len + chunk_len - 1) / chunk_len;

    if (!sc.need(0, sizeof(Top2) * (size_t)nchunks * l_len) ||
        !sc.need(1, sizeof(float) * 128 * (size_t)(r_len > 0 ? r_len : 1)) ||
        !sc.need(2, (sizeof(int) * 3 + sizeof(float) * 2) * (size_t)l_len + 64))
        return PSX_ERR_NOMEM;
    Top2* d_partial = static_cast<Top2*>(sc.buf[0]);
    float* d_rperm = static_cast<float*>(sc.buf[1]);
    int* d_out = static_cast<int*>(sc.buf[2]);
    // matches, distances and the prefilter's flag in ONE buffer: one copy back per call (three cost ~5 us each)
    float* d_dist = reinterpret_cast<float*>(d_out + 3 * (size_t)l_len);
    int* d_flag = reinterpret_cast<int*>(d_dist + 2 * (size_t)l_len);
    hipStream_t st = sc.stream;
    // POPSIFT_MATCH_MFMA=0: the exact scan of every pair (rounds 1-4); default: MFMA prefilter + exact evaluation of the
    // candidates (identical results by construction; used from 2 * MF_SEED = 4096 right / 256 left descriptors on)
    static const bool use_mfma = [] { const char* e = getenv("POPSIFT_MATCH_MFMA"); return !(e != nullptr && e[0] == '0'); }();
    bool exact_scan = true;
    int* d_flag_used = nullptr;
    int* d_cct_used = nullptr;
    // (a failed allocation of the prefilter's scratch -- 4 KB of candidate slots per left descriptor -- leaves the exact scan)
    if (use_mfma && r_len >= 2 * MF_SEED && l_len >= 256 &&
        sc.need(4, sizeof(unsigned short) * 128 * (size_t)l_len) && sc.need(5, sizeof(unsigned short) * 128 * (size_t)r_len) &&
        sc.need(6, sizeof(float) * (1 + 2 * MF_SEEDCH) * (size_t)l_len) && sc.need(7, sizeof(float) * (size_t)r_len + 1024) &&
        sc.need(8, sizeof(int) * (size_t)MF_SEGS * l_len) && sc.need(9, sizeof(int) * (size_t)MF_CAP * l_len)) {
        unsigned short* d_lf16 = static_cast<unsigned short*>(sc.buf[4]);
        unsigned short* d_rf16 = static_cast<unsigned short*>(sc.buf[5]);
        float* d_ln2 = static_cast<float*>(sc.buf[6]);
        float* d_seed = d_ln2 + l_len;
        // in front of the right norms, at an address that does not move with r_len (the slots are zeroed by the previous call):
        // 2 x MF_MAXSLOTS maximum slots, 4 parameters
        unsigned* d_rmax = static_cast<unsigned*>(sc.buf[7]);
        unsigned* d_lmax = d_rmax + MF_MAXSLOTS;
        float* d_par = reinterpret_cast<float*>(d_lmax + MF_MAXSLOTS);
        float* d_rn2 = reinterpret_cast<float*>(d_rmax + 256);
        int* d_cct = static_cast<int*>(sc.buf[8]);
        int* d_cand = static_cast<int*>(sc.buf[9]);
        // workgroups the prefilter kernel keeps resident (the occupancy the runtime computes from its registers and LDS: 3 per CU)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        static const int per_cu = [] {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_match_mfma<false>, 256, 0) != hipSuccess || n < 1) n = 2;
            if (const char* e = getenv("POPSIFT_MATCH_WGS_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 8) n = v; }
            return n;
        }();
        const int resident = per_cu * cus;
        const int lblocks = (l_len + 255) / 256;
        // seeding pass: the first ~MF_SEED right descriptors in at most MF_SEEDCH chunks, ONE round of resident workgroups
        // (8 chunks x 72 left blocks were 576 workgroups on 512 slots: a second, nearly empty round); the slots of the
        // chunks that do not run hold +inf
        int nseed = resident / lblocks;
        if (nseed > MF_SEEDCH) nseed = MF_SEEDCH;
        if (nseed < 1) nseed = 1;
        const int seedlen = (((MF_SEED + nseed - 1) / nseed + MF_TILE - 1) / MF_TILE) * MF_TILE;
        static const bool stats = getenv("POPSIFT_MATCH_STATS") != nullptr;       // measurement: candidates per left descriptor
        // the candidate counts and the maximum slots: zeroed here after an allocation or a call that did not finish the usual
        // way, otherwise left at zero by the previous call's k_match_exact
        if (!sc.tidy && (hipMemsetAsync(d_cct, 0, sc.cap[8], st) != hipSuccess ||
                         hipMemsetAsync(d_rmax, 0, sizeof(unsigned) * 2 * MF_MAXSLOTS, st) != hipSuccess))
            return PSX_ERR_HIP;
        sc.tidy = false;                                       // until this call's results are back with the flag down
        if ((nseed < MF_SEEDCH && hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_seed), 0x7f800000, 2 * (size_t)MF_SEEDCH * l_len, st) != hipSuccess))
            return PSX_ERR_HIP;
        const int rblk = (r_len * 32 + 255) / 256, lblk = (l_len * 32 + 255) / 256;
        // norms: a few descriptors per 32-thread group (one per group made 9 216 workgroups of one load each: 18 us for 19 MB)
        const int rnb = rblk < 4 * cus ? rblk : 4 * cus, lnb = lblk < 4 * cus ? lblk : 4 * cus;
        hipLaunchKernelGGL(k_match_norms, dim3(rnb + lnb), dim3(256), 0, st, d_right, r_len, d_rn2, d_rmax, rnb, d_left, l_len, d_ln2, d_lmax);
        hipLaunchKernelGGL(k_match_cvt, dim3(rblk + lblk), dim3(256), 0, st, d_right, r_len, d_rf16, rblk, d_left, l_len, d_lf16, d_lmax, d_rmax, d_par, d_flag);
        hipLaunchKernelGGL((k_match_mfma<true>), dim3(lblocks, nseed), dim3(256), 0, st, d_lf16, d_ln2, l_len, d_rf16, d_rn2, r_len,
                           seedlen, d_par, d_seed, d_cct, d_cand);
        // every chunk of the right side: FULL rounds of resident workgroups (15 chunks x 72 left blocks = 1080 workgroups on 768 slots
        // ran 1.4 rounds, i.e. the time of two), whole tiles per chunk
        // POPSIFT_MATCH_ROUNDS: rounds of resident workgroups the chunking aims at (measurement switch)
        static const int rounds = [] { const char* e = getenv("POPSIFT_MATCH_ROUNDS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 4 ? v : 1; }();
        int mchunks = (rounds * resident) / lblocks;
        if (mchunks > (r_len + 8 * MF_TILE - 1) / (8 * MF_TILE)) mchunks = (r_len + 8 * MF_TILE - 1) / (8 * MF_TILE);
        if (mchunks > MF_SEGS / 2) mchunks = MF_SEGS / 2;          // one candidate segment per (chunk, half wave)
        if (mchunks < 1) mchunks = 1;
        int mlen = (r_len + mchunks - 1) / mchunks;
        mlen = ((mlen + MF_TILE - 1) / MF_TILE) * MF_TILE;
        mchunks = (r_len + mlen - 1) / mlen;
        hipLaunchKernelGGL((k_match_mfma<false>), dim3(lblocks, mchunks), dim3(256), 0, st, d_lf16, d_ln2, l_len, d_rf16, d_rn2, r_len, mlen,
                           d_par, d_seed, d_cct, d_cand);
        // the exact evaluation of the candidates goes out at once (it also raises the flag for an overflowed candidate segment); the
        // flag comes back with the results (one synchronisation per call instead of two)
        hipLaunchKernelGGL(k_match_exact, dim3((l_len + 3) / 4), dim3(256), 0, st, d_left, l_len, d_right, r_len, d_cct, d_cand,
                           d_out, d_dist, d_flag, d_rmax, stats ? 0 : 1);
        exact_scan = false;
        d_flag_used = d_flag; d_cct_used = d_cct;
    }
    const size_t mb = sizeof(int) * 3 * (size_t)l_len, db = sizeof(float) * 2 * (size_t)l_len;
    if (!sc.need_pinned(mb + db + 64)) return PSX_ERR_NOMEM;
    char* hp = static_cast<char*>(sc.hpin);
    int* h_flagp = reinterpret_cast<int*>(hp + mb + db);
    *h_flagp = 0;
    auto fetch = [&]() -> bool {
        return hipGetLastError() == hipSuccess &&
               hipMemcpyAsync(hp, d_out, mb + db + (d_flag_used ? sizeof(int) : 0), hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipStreamSynchronize(st) == hipSuccess;
    };
    if (!exact_scan) {
        if (!fetch()) return PSX_ERR_HIP;
        const int h_flag = *h_flagp;
        static const bool stats = getenv("POPSIFT_MATCH_STATS") != nullptr;
        if (stats) {
            std::vector<int> h((size_t)l_len * MF_SEGS);
            if (hipMemcpy(h.data(), d_cct_used, sizeof(int) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
                long long sum = 0; int mx = 0, mxl = 0;
                for (int i = 0; i < l_len; i++) { int t = 0; for (int q = 0; q < MF_SEGS; q++) { const int v = h[(size_t)i * MF_SEGS + q]; t += v; if (v > mx) mx = v; } sum += t; if (t > mxl) mxl = t; }
                fprintf(stderr, "psx_match prefilter: %d x %d, candidates per left descriptor: mean %.1f, max %d; fullest segment %d of %d%s\n", l_len, r_len,
                        (double)sum / l_len, mxl, mx, MF_SEGCAP, h_flag ? " -> exact scan of every pair" : "");
            }
        }
        // a list overflowed (thousands of near-equal neighbours) or the norms are out of the margin's reach: the exact scan below
        if (h_flag != 0) { exact_scan = true; d_flag_used = nullptr; }
        else if (!stats) sc.tidy = true;
    }
    if (exact_scan) {
    if (r_len > 0)
        hipLaunchKernelGGL(k_match_permute, dim3((r_len * 64 + 255) / 256), dim3(256), 0, st, d_right, r_len, d_rperm);
    hipLaunchKernelGGL(k_match_partial, dim3(lgroups, nchunks), dim3(64), 0, st, d_left, l_len, d_rperm, r_len,
                       chunk_len, d_partial);
    hipLaunchKernelGGL(k_match_merge, dim3((l_len + 255) / 256), dim3(256), 0, st, d_partial, l_len, nchunks,
                       r_len, d_out, d_dist);
    if (!fetch()) return PSX_ERR_HIP;
    }
    memcpy(host_match, hp, mb);
    if (host_dist) memcpy(host_dist, hp + mb, db);
    return PSX_OK;
}
