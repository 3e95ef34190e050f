// dispatch_ramp.hip -- how long does the chip take to START N workgroups?  (tools/ubench, measurement only)
// A kernel whose workgroups do nothing (one clock read, one conditional store that never happens) is all dispatch: its
// duration over N workgroups is the ramp every real launch of that shape pays before its last workgroup starts.
// Variants: threads per workgroup (256 / 512 / 1024), static LDS per workgroup (0 / 32 KB: the LDS allocation is part of the
// dispatch), a body of T microseconds of s_sleep per workgroup (ramp + life + drain of a one-round grid).
// Timing: hipExtLaunchKernelGGL start / stop events (begin / end timestamps of the dispatch itself), best and median of 20.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
template <int LDSB>
__global__ void k_ramp(int* sink, int spin)
{
    __shared__ int s[LDSB > 0 ? LDSB / 4 : 1];
    if (LDSB > 0 && threadIdx.x == 0) s[blockIdx.x & 63] = spin;
    const long long t0 = wall_clock64();
    while (spin > 0 && wall_clock64() - t0 < (long long)spin) __builtin_amdgcn_s_sleep(8);
    if (sink != nullptr && blockIdx.x == 0x7fffffff) sink[threadIdx.x] = s[0];
}
template <int LDSB>
static void run(int wgs, int nt, int spin, int* d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> t;
    for (int i = 0; i < 24; i++) {
        hipExtLaunchKernelGGL((k_ramp<LDSB>), dim3(wgs), dim3(nt), 0, 0, e0, e1, 0, d, spin);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 4) t.push_back(ms * 1000.0f);
    }
    std::sort(t.begin(), t.end());
    printf("lds %5d B  threads %4d  workgroups %5d  body %5.1f us : min %6.2f  median %6.2f us\n", LDSB, nt, wgs, spin / 100.0, t[0], t[t.size() / 2]);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}
int main()
{
    int* d; (void)hipMalloc(&d, 4096);
    const int grids[] = {256, 512, 900, 1024, 2048, 4096, 8192};
    for (int spin : {0, 1000}) {           // wall_clock64 ticks at 100 MHz: 1000 ticks = 10 us
        for (int g : grids) run<0>(g, 256, spin, d);
        for (int g : grids) run<32768>(g, 256, spin, d);
        for (int g : {128, 256, 450, 512, 1024, 2048}) run<32768>(g, 512, spin, d);
        for (int g : {64, 128, 225, 256, 512, 1024}) run<65536>(g, 1024, spin, d);
    }
    (void)hipFree(d);
    return 0;
}
