// gridfilter.hip -- spatial thinning of the initial extrema before orientation assignment.
//
// Reference: Pyramid::extrema_filter_grid, s_filtergrid.cu:113-325, called from
// Pyramid::orientation (s_orientation.cu:378-383) when FilterMaxExtrema > 0 and the frame found
// more than 1.1x that many extrema.  The reference strings ~20 thrust calls, two device-wide syncs
// and a host round trip for the per-cell limits.  Here:
//   k_gf_keys   one 64-bit key per initial extremum, (cell << 32) | order-bits(scale), value = its
//               sequence number (octave-major, index order)
//   rocprim     one stable radix sort of the pairs (thrust::sort_by_key's role)
//   k_gf_apply  ONE workgroup: run lengths per cell by binary search on the sorted keys, the
//               clamp limit (the arithmetic the reference does on the host, :203-270), ignore
//               flags, and the stable per-octave compaction of i_ext_off (:272-317)
// Only the total count is read by the host (it sizes the sort), as in the reference.
#include "psx_internal.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int GF_NT = 1024;

__device__ __forceinline__ int gf_count(const PsxParams* P, const PsxCounters* cnt, int o)
{
    return min(cnt->ext_ct[o], P->max_extrema);
}

__global__ __launch_bounds__(256) void k_gf_keys(const PsxParams* __restrict__ P, const PsxCounters* cnt,
                                                 int mode, int total,
                                                 unsigned long long* __restrict__ keys,
                                                 unsigned* __restrict__ vals)
{
    const int max_cell = P->grid_size * P->grid_size + P->grid_size;
    for (int seq = blockIdx.x * blockDim.x + threadIdx.x; seq < total; seq += gridDim.x * blockDim.x) {
        int o = 0, base = 0;
        for (;;) {
            const int c = gf_count(P, cnt, o);
            if (seq < base + c || o + 1 >= P->num_octaves) break;
            base += c; o++;
        }
        const psx_iext e = P->iext[o][seq - base];
        // FunctionExtractCell, s_filtergrid.cu:56-69
        const float scale = e.sigma * powf(2.0f, (float)o);
        const unsigned bits = __float_as_uint(scale);          // scale > 0: bit order == value order
        unsigned order = 0u;                                   // RandomScale: cell only (:177-186)
        if (mode == PSX_FILTER_LARGEST_FIRST) order = ~bits;   // FunctionSort_IncCell_DecScale, :36-44
        else if (mode == PSX_FILTER_SMALLEST_FIRST) order = bits;
        const unsigned cell = (unsigned)psx_clampi(e.cell, 0, max_cell);
        keys[seq] = ((unsigned long long)cell << 32) | order;
        vals[seq] = (unsigned)seq;
    }
}

// first sorted position whose cell is >= c
__device__ int gf_lower_bound(const unsigned long long* keys, int total, unsigned c)
{
    int lo = 0, hi = total;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((unsigned)(keys[mid] >> 32) < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(GF_NT) void k_gf_apply(const PsxParams* __restrict__ P, PsxCounters* cnt,
                                                    int total, int filter_max,
                                                    const unsigned long long* __restrict__ keys,
                                                    const unsigned* __restrict__ vals,
                                                    int* __restrict__ scratch)
{
    // scratch (global, 4*(n + grid + 2) ints): cell starts, then per run: count, offset, sorted count
    const int t = threadIdx.x;
    const int n = P->grid_size * P->grid_size;
    const int ncell = n + P->grid_size + 1;          // cell values 0 .. ncell-1 can occur
    int* cstart  = scratch;                           // [ncell + 1]
    int* rcount  = cstart + ncell + 1;                // [n] count of run r (reduce_by_key output, :191-194)
    int* roffset = rcount + n;                        // [n]
    int* rsorted = roffset + n;                       // [n] counts in increasing order
    __shared__ int s_newlimit;
    __shared__ int s_wsum[GF_NT / PSX_WAVE];
    __shared__ int s_base;

    for (int c = t; c <= ncell; c += GF_NT) cstart[c] = (c == ncell) ? total : gf_lower_bound(keys, total, (unsigned)c);
    __syncthreads();

    // runs = non-empty cells in increasing cell order; only the first n are kept (cell_counts has n slots)
    if (t == 0) {
        int r = 0, acc = 0;
        for (int c = 0; c < ncell && r < n; c++) {
            const int k = cstart[c + 1] - cstart[c];
            if (k > 0) { rcount[r] = k; roffset[r] = acc; acc += k; r++; }
        }
        for (; r < n; r++) { rcount[r] = 0; roffset[r] = acc; }
    }
    __syncthreads();

    // counts in increasing order (thrust::sort_by_key on the host vectors, :216-217), by ranking
    for (int i = t; i < n; i += GF_NT) {
        const int ci = rcount[i];
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const int cj = rcount[j];
            rank += (cj < ci || (cj == ci && j < i)) ? 1 : 0;
        }
        rsorted[rank] = ci;
    }
    __syncthreads();

    if (t == 0) {
        // :219-262
        int ct = 0, acc = 0;
        for (int i = 0; i < n; i++) {
            acc += rsorted[i];
            const int sumup = rsorted[i] * (n - 1 - i) + acc;
            if (sumup > filter_max) ct++;
        }
        int newlimit = 0x7fffffff;
        if (ct > 0) {
            int tail = 0;
            for (int i = n - ct; i < n; i++) tail += rsorted[i];
            const float tailaverage = (float)tail / (float)ct;
            newlimit = (int)ceilf(tailaverage - (float)((total - filter_max) / ct));
        }
        s_newlimit = newlimit;
    }
    __syncthreads();
    const int newlimit = s_newlimit;

    // FunctionDisableExtremum over [offset + clamped count, limit) of every run, :264-279
    for (int r = 0; r < n; r++) {
        const int k = rcount[r];
        const int from = roffset[r] + min(k, newlimit);
        const int to = roffset[r] + k;
        for (int p = from + t; p < to; p += GF_NT) {
            const int seq = (int)vals[p];
            int o = 0, base = 0;
            for (;;) {
                const int c = gf_count(P, cnt, o);
                if (seq < base + c || o + 1 >= P->num_octaves) break;
                base += c; o++;
            }
            P->iext[o][seq - base].ignore = 1;
        }
    }
    __threadfence();
    __syncthreads();

    // i_ext_off[o] = indices of the surviving extrema in increasing order (copy_if, :303-308)
    const int lane = t & (PSX_WAVE - 1), wave = t >> 6;
    for (int o = 0; o < P->num_octaves; o++) {
        const int count = gf_count(P, cnt, o);
        const psx_iext* ie = P->iext[o];
        int* off = P->iext_off[o];
        if (t == 0) s_base = 0;
        __syncthreads();
        for (int start = 0; start < count; start += GF_NT) {
            const int i = start + t;
            const bool keep = (i < count) && (ie[i].ignore == 0);
            const unsigned long long m = __ballot(keep);
            if (lane == 0) s_wsum[wave] = __popcll(m);
            __syncthreads();
            int pre = s_base;
            for (int w = 0; w < wave; w++) pre += s_wsum[w];
            if (keep) off[pre + __popcll(m & ((1ull << lane) - 1ull))] = i;
            __syncthreads();
            if (t == 0) {
                int s = 0;
                for (int w = 0; w < GF_NT / PSX_WAVE; w++) s += s_wsum[w];
                s_base += s;
            }
            __syncthreads();
        }
        if (t == 0) {
            cnt->iext_ct[o] = count;        // what psx_dump_iext reports
            cnt->ext_ct[o] = s_base;        // hct.ext_ct[o] = reduce(grid), :310
        }
        __syncthreads();
    }
}

} // namespace

size_t psx_gridfilter_scratch_ints(int grid_size)
{
    const size_t n = (size_t)grid_size * grid_size;
    return 4 * (n + (size_t)grid_size + 2);
}

hipError_t psx_gridfilter_sort_bytes(int total, size_t* bytes)
{
    unsigned long long* k = nullptr; unsigned* v = nullptr;
    return rocprim::radix_sort_pairs(nullptr, *bytes, k, k, v, v, (size_t)total, 0, 64, (hipStream_t)0);
}

hipError_t psx_launch_gridfilter(const PsxParams* d_params, PsxCounters* d_cnt, int mode, int total,
                                 int filter_max, unsigned long long* keys_in, unsigned long long* keys_out,
                                 unsigned* vals_in, unsigned* vals_out, void* temp, size_t temp_bytes,
                                 int* scratch, hipStream_t s)
{
    if (total <= 0) return hipSuccess;
    const int blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_gf_keys, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, s,
                       d_params, d_cnt, mode, total, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                             (size_t)total, 0, 64, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gf_apply, dim3(1), dim3(GF_NT), 0, s,
                       d_params, d_cnt, total, filter_max, keys_out, vals_out, scratch);
    return hipGetLastError();
}
