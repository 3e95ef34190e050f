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

// ---- cross-stream hand-over check (tests/test_gpu_api.py) ------------------------------------------------------------
// Round 3's split-frame experiment (part 1 of a frame's orientation scan on stream B, part 2 on stream A behind an event,
// both with zero-copy export) saw a result that carried only part 2's count and blamed "the cross-stream wait under load
// with PCIe-bound stores".  This is that hand-over reduced to its mechanism: a kernel on stream A that is busy storing
// into mapped host memory (slow PCIe stores) and whose LAST workgroup writes a word of device memory with a plain store;
// an event; a one-workgroup kernel on stream B behind hipStreamWaitEvent that reads the word with a plain load and reports
// it; an event back.  If every report is the value of its round, stream order across an event carries plain device-memory
// stores between kernels on this stack (kernel-end release, kernel-start acquire), and the experiment's failure was a
// missing edge in ITS launch order, not a scope problem of the export stores.
namespace {
__global__ void k_xs_producer(int* __restrict__ word, int* __restrict__ ticket, volatile int* __restrict__ host_sink, int words_per_thread, int round)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < words_per_thread; i++) host_sink[(size_t)i * gridDim.x * blockDim.x + gid] = round + i;   // PCIe-bound
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int t = atomicAdd(ticket, 1);
        if (t == (int)gridDim.x - 1) { *ticket = 0; *word = round; }                  // the last workgroup: a plain store
    }
}
__global__ void k_xs_consumer(const int* __restrict__ word, int* __restrict__ out, int round)
{
    if (threadIdx.x == 0) out[round] = *word;                                          // a plain load
}
} // namespace

extern "C" int psx_debug_cross_stream(int device, int rounds, int pcie_words_per_thread, int* stale)
{
    if (!stale || rounds < 1 || pcie_words_per_thread < 0) return PSX_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return PSX_ERR_HIP;
    *stale = -1;
    const int grid = 64, block = 256;
    int *word = nullptr, *ticket = nullptr, *out = nullptr, *sink = nullptr, *h_out = nullptr;
    hipStream_t sa = nullptr, sb = nullptr;
    hipEvent_t ea = nullptr, eb = nullptr;
    int rc = PSX_ERR_HIP;
    const size_t sink_words = (size_t)(pcie_words_per_thread > 0 ? pcie_words_per_thread : 1) * grid * block;
    if (hipMalloc(reinterpret_cast<void**>(&word), 256) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&ticket), 256) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&out), sizeof(int) * (size_t)rounds) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&sink), sizeof(int) * sink_words, hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h_out), sizeof(int) * (size_t)rounds, hipHostMallocDefault) != hipSuccess) goto done;
    if (hipMemset(word, 0xff, 256) != hipSuccess || hipMemset(ticket, 0, 256) != hipSuccess || hipMemset(out, 0xff, sizeof(int) * (size_t)rounds) != hipSuccess) goto done;
    if (hipStreamCreateWithFlags(&sa, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&sb, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) goto done;
    for (int r = 0; r < rounds; r++) {
        if (r > 0 && hipStreamWaitEvent(sa, eb, 0) != hipSuccess) goto done;          // stream A continues behind stream B's reader
        hipLaunchKernelGGL(k_xs_producer, dim3(grid), dim3(block), 0, sa, word, ticket, sink, pcie_words_per_thread, r);
        if (hipEventRecord(ea, sa) != hipSuccess || hipStreamWaitEvent(sb, ea, 0) != hipSuccess) goto done;
        hipLaunchKernelGGL(k_xs_consumer, dim3(1), dim3(64), 0, sb, word, out, r);
        if (hipEventRecord(eb, sb) != hipSuccess) goto done;
    }
    if (hipStreamSynchronize(sa) != hipSuccess || hipStreamSynchronize(sb) != hipSuccess || hipGetLastError() != hipSuccess) goto done;
    if (hipMemcpy(h_out, out, sizeof(int) * (size_t)rounds, hipMemcpyDeviceToHost) != hipSuccess) goto done;
    *stale = 0;
    for (int r = 0; r < rounds; r++) if (h_out[r] != r) (*stale)++;
    rc = PSX_OK;
done:
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    if (sa) (void)hipStreamDestroy(sa);
    if (sb) (void)hipStreamDestroy(sb);
    (void)hipFree(word); (void)hipFree(ticket); (void)hipFree(out);
    if (sink) (void)hipHostFree(sink);
    if (h_out) (void)hipHostFree(h_out);
    return rc;
}
