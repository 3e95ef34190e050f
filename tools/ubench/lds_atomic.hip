// micro-benchmark: LDS atomic throughput on gfx950 (ds_add_f32 vs ds_add_u32 vs plain RMW)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride_mode)
{
    __shared__ float sf[4096];
    __shared__ unsigned su[4096];
    __shared__ unsigned long long sl[4096];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 4096; i += 256) { sf[i] = 0.f; su[i] = 0u; sl[i] = 0ull; }
    __syncthreads();
    int base = wave * 1024;
    int idx;
    if (stride_mode == 0) idx = lane;              // all distinct, conflict-free banks
    else if (stride_mode == 1) idx = lane & 15;    // 4 lanes share an address
    else if (stride_mode == 2) idx = 0;            // all lanes same address
    else idx = (lane & 15) * 129 + (lane >> 4);    // padded copies
    long long t0 = clock64();
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        const int a = base + ((idx + (i & 7) * 64) & 1023);
        if (MODE == 0) atomicAdd(&sf[a], 1.0f);
        else if (MODE == 1) atomicAdd(&su[a], 1u);
        else if (MODE == 4) atomicAdd(&sl[a], (unsigned long long)(lane * 1.5f * 4294967296.0f));
        else if (MODE == 2) { sf[a] += 1.0f; }     // non-atomic RMW (race, timing only)
        else { acc += sf[a]; }                      // read only
    }
    __syncthreads();
    long long t1 = clock64();
    if (t == 0) out[blockIdx.x] = (float)(t1 - t0) / iters;
    if (MODE == 3 && acc == 12345.f) out[0] = acc;
}

int main()
{
    float* d; hipMalloc(&d, 4096 * sizeof(float));
    const int iters = 4096;
    const char* names[] = {"ds_add_f32", "ds_add_u32", "plain RMW", "ds_read", "ds_add_u64"};
    const char* sm[] = {"distinct", "4-share", "all-same", "padded"};
    for (int nb = 1024; nb <= 1024; nb *= 32)
    for (int mode = 0; mode < 5; mode++)
        for (int s = 0; s < 4; s++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, d, iters, s);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, d, iters, s);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, d, iters, s);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(256), 0, 0, d, iters, s);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, d, iters, s);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<float> h(nb); hipMemcpy(h.data(), d, nb * sizeof(float), hipMemcpyDeviceToHost);
            printf("blocks %4d %-11s %-9s: %.1f clk/iter per WG (4 waves)  %.3f ms\n", nb, names[mode], sm[s], h[0], ms);
        }
    return 0;
}
