// util.hip -- small device utilities behind the C-ABI that are not part of the extraction chain:
//   * psx_copy_bench: the measured HBM roofline (">= 1 GiB float4 copy kernel"), the
//     denominator bench.py reports next to the 8 TB/s peak;
//   * k_feature_ptrs: psx_feature (descriptor indices) -> popsift::Feature records with DEVICE
//     descriptor pointers for FeaturesDev (features.h:104-122; the reference's prep_features writes such
//     pointers directly, sift_pyramid.cu:242-280).
#include "psx_internal.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// Streaming copy, 16 B per lane, ONE access per thread, one workgroup = one contiguous 4 KiB block, as many
// workgroups as blocks.  Measured on MI355X (tools/ubench/copy_variants.hip, 1 GiB + 1 GiB): this shape reaches
// 6.2 TB/s; grid-stride loops over a few thousand persistent workgroups (what a "tuned" copy usually looks
// like) stay at 4.5-5.0 TB/s, hipMemcpyAsync at 4.8 TB/s -- concurrent workgroups then touch addresses far
// apart instead of neighbouring DRAM pages.
template <bool NT>
__global__ __launch_bounds__(256) void k_copy16(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    else    dst[i] = src[i];
}

__global__ void k_feature_ptrs(const psx_feature* __restrict__ in, psx_feature_dev* __restrict__ out, int n,
                               float* desc_base, int num_desc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const psx_feature f = in[i];
    psx_feature_dev o;
    o.debug_octave = f.debug_octave;
    o.xpos = f.xpos; o.ypos = f.ypos; o.sigma = f.sigma;
    o.num_ori = f.num_ori;
    o.pad = 0;
#pragma unroll
    for (int k = 0; k < PSX_ORI_MAX; k++) {
        o.orientation[k] = f.orientation[k];
        o.desc[k] = (f.desc_idx[k] >= 0 && f.desc_idx[k] < num_desc) ? desc_base + (size_t)f.desc_idx[k] * 128 : nullptr;
    }
    out[i] = o;
}

} // namespace

hipError_t psx_launch_feature_ptrs(const psx_feature* in, psx_feature_dev* out, int n, float* desc_base, int num_desc,
                                   hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_feature_ptrs, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n, desc_base, num_desc);
    return hipGetLastError();
}

extern "C" int psx_copy_bench(int device, size_t bytes, int reps, float* avg_ms, double* bytes_moved)
{
    if (!avg_ms || reps < 1) return PSX_ERR_INVALID;
    if (bytes == 0) bytes = (size_t)1 << 30;                 // 1 GiB read + 1 GiB written: far beyond the 256 MiB Infinity Cache
    bytes &= ~(size_t)15;
    if (hipSetDevice(device) != hipSuccess) return PSX_ERR_HIP;
    void *src = nullptr, *dst = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PSX_OK;
    float best = 0.0f;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) { rc = PSX_ERR_NOMEM; goto done; }
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess || hipMemsetAsync(src, 0x3c, bytes, st) != hipSuccess) { rc = PSX_ERR_HIP; goto done; }
    for (int variant = 0; variant < 2; variant++) {
        const size_t n4 = bytes / 16;
        const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
        for (int r = -2; r < reps; r++) {                    // two untimed warm-up launches
            if (r == 0 && hipEventRecord(e0, st) != hipSuccess) { rc = PSX_ERR_HIP; goto done; }
            if (variant == 0) hipLaunchKernelGGL((k_copy16<false>), grid, block, 0, st, (const v4f*)src, (v4f*)dst, n4);
            else              hipLaunchKernelGGL((k_copy16<true>), grid, block, 0, st, (const v4f*)src, (v4f*)dst, n4);
        }
        float ms = 0.0f;
        if (hipEventRecord(e1, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
            hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) { rc = PSX_ERR_HIP; goto done; }
        ms /= reps;
        if (variant == 0 || ms < best) best = ms;
    }
    *avg_ms = best;
    if (bytes_moved) *bytes_moved = 2.0 * (double)bytes;
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(src); (void)hipFree(dst);
    return rc;
}
