// match.hip -- brute-force 2-nearest-neighbour descriptor matcher of MatchingMode.
//
// Reference: FeaturesDev::match -> compute_distance / l2_in_t0 (features.cu:160-225): for every left
// descriptor the two right descriptors with the smallest squared L2 distance, scanned in index order
// with strict '<' (ties keep the earlier index), accept = d1 / d2 < 0.8.  The reference runs one
// 32-thread block per left descriptor and walks the right side serially with a __syncthreads per pair.
//
// The distance is NOT reshaped into a GEMM (|a|^2 + |b|^2 - 2ab cancels catastrophically for near
// matches, and the indices are integer output that has to agree with the reference): every pair is
// evaluated with the reference's own operation tree, so the distances are bit-identical to
// oracle/sift_oracle.c:
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

} // namespace

// Scratch of one calling thread: a private non-blocking stream (no null-stream launch, so nothing else on
// the device is synchronised) and buffers that only ever grow.  Freed when the thread exits.
namespace {
struct MatchScratch {
    int device = -1;
    hipStream_t stream = nullptr;
    void* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    void release()
    {
        if (device < 0) return;
        int cur = -1;                                        // the caller's current device is left as it was
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        (void)hipSetDevice(device);
        for (int i = 0; i < 4; i++) { (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0; }
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr; device = -1;
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    bool need(int i, size_t bytes)
    {
        if (bytes <= cap[i] && buf[i]) return true;
        (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0;
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

// frees the calling thread's matcher scratch (stream + up to 4 device buffers); an explicit user call -- PopSift::uninit
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
        !sc.need(2, sizeof(int) * 3 * (size_t)l_len) || !sc.need(3, sizeof(float) * 2 * (size_t)l_len))
        return PSX_ERR_NOMEM;
    Top2* d_partial = static_cast<Top2*>(sc.buf[0]);
    float* d_rperm = static_cast<float*>(sc.buf[1]);
    int* d_out = static_cast<int*>(sc.buf[2]);
    float* d_dist = static_cast<float*>(sc.buf[3]);
    hipStream_t st = sc.stream;
    if (r_len > 0)
        hipLaunchKernelGGL(k_match_permute, dim3((r_len * 64 + 255) / 256), dim3(256), 0, st, d_right, r_len, d_rperm);
    hipLaunchKernelGGL(k_match_partial, dim3(lgroups, nchunks), dim3(64), 0, st, d_left, l_len, d_rperm, r_len,
                       chunk_len, d_partial);
    hipLaunchKernelGGL(k_match_merge, dim3((l_len + 255) / 256), dim3(256), 0, st, d_partial, l_len, nchunks,
                       r_len, d_out, d_dist);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(host_match, d_out, sizeof(int) * 3 * (size_t)l_len, hipMemcpyDeviceToHost, st) != hipSuccess ||
        (host_dist && hipMemcpyAsync(host_dist, d_dist, sizeof(float) * 2 * (size_t)l_len, hipMemcpyDeviceToHost, st) != hipSuccess) ||
        hipStreamSynchronize(st) != hipSuccess)
        return PSX_ERR_HIP;
    return PSX_OK;
}
