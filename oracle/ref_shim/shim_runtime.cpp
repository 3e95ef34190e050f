// oracle/ref_shim/shim_runtime.cpp -- CPU implementation of the CUDA emulation declared in
// ref_shim/cuda_runtime.h.  TEST INFRASTRUCTURE ONLY (see the header).
#include "cuda_runtime.h"

#include <ucontext.h>
#include <vector>
#include <string>
#include <set>

namespace shim {

thread_local uint3 t_threadIdx = {0, 0, 0}, t_blockIdx = {0, 0, 0};
thread_local dim3 t_blockDim(1, 1, 1), t_gridDim(1, 1, 1);

// ------------------------------------------------------------------------------------------------
// Fibers: every CUDA thread of the running block is a ucontext fiber of the launching OS thread.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr size_t STACK_BYTES = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    uint3 tid;
    int lin = 0;
};

struct WarpState {
    int alive = 0;          // fibers of this warp that have not returned
    int arrived = 0;
    unsigned long long gen = 0;
    unsigned long long vals[32];
    unsigned char present[32];
    unsigned long long res[2][32];
    unsigned char res_present[2][32];
};

struct BlockState {
    std::vector<Fiber> fibers;
    std::vector<WarpState> warps;
    int alive = 0;
    int bar_arrived = 0;
    unsigned long long bar_gen = 0;
    ucontext_t sched;
    int current = -1;
    const std::function<void()>* body = nullptr;
};

thread_local BlockState* g_blk = nullptr;
thread_local std::vector<char*> g_stack_pool;

void yield_to_scheduler()
{
    BlockState* b = g_blk;
    Fiber& f = b->fibers[b->current];
    swapcontext(&f.ctx, &b->sched);
}

void complete_warp_op(WarpState& w)
{
    const int slot = (int)(w.gen & 1);
    memcpy(w.res[slot], w.vals, sizeof(w.vals));
    memcpy(w.res_present[slot], w.present, sizeof(w.present));
    memset(w.present, 0, sizeof(w.present));
    w.arrived = 0;
    w.gen++;
}

void fiber_entry()
{
    BlockState* b = g_blk;
    Fiber& f = b->fibers[b->current];
    (*b->body)();
    f.done = true;
    // leaving threads no longer take part in collectives
    WarpState& w = b->warps[f.lin / 32];
    w.alive--;
    b->alive--;
    if (w.alive > 0 && w.arrived == w.alive) complete_warp_op(w);
    if (b->alive > 0 && b->bar_arrived == b->alive) { b->bar_arrived = 0; b->bar_gen++; }
    swapcontext(&f.ctx, &b->sched);
}

// kernels without any block- or warp-level collective: plain loops, no fibers
bool is_simple_kernel(const std::string& n)
{
    static const char* simple[] = {
        "normalizedSource::horiz", "absoluteSource::vert", "make_dog", "get_by_2_pick_every_second",
        "get_by_2_interpolate", "prep_features", "print_gauss_filter_symbol",
    };
    for (const char* s : simple) {
        const std::string ss(s);
        // match "…::horiz" exactly, not "…::horiz_all" etc.
        size_t p = n.find(ss);
        if (p != std::string::npos) {
            const size_t e = p + ss.size();
            if (e == n.size() || !(isalnum((unsigned char)n[e]) || n[e] == '_')) return true;
        }
    }
    return false;
}

} // namespace

void launch(const char* name, dim3 grid, dim3 block, const std::function<void()>& body)
{
    const int nthreads = (int)(block.x * block.y * block.z);
    const bool simple = is_simple_kernel(name);
    t_gridDim = grid;
    t_blockDim = block;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        t_blockIdx = uint3{bx, by, bz};
        if (simple) {
            for (unsigned tz = 0; tz < block.z; tz++)
            for (unsigned ty = 0; ty < block.y; ty++)
            for (unsigned tx = 0; tx < block.x; tx++) {
                t_threadIdx = uint3{tx, ty, tz};
                body();
            }
            continue;
        }
        BlockState blk;
        blk.body = &body;
        blk.fibers.resize(nthreads);
        blk.warps.resize((nthreads + 31) / 32);
        for (auto& w : blk.warps) { memset(w.present, 0, sizeof(w.present)); }
        blk.alive = nthreads;
        g_blk = &blk;
        int lin = 0;
        for (unsigned tz = 0; tz < block.z; tz++)
        for (unsigned ty = 0; ty < block.y; ty++)
        for (unsigned tx = 0; tx < block.x; tx++, lin++) {
            Fiber& f = blk.fibers[lin];
            f.tid = uint3{tx, ty, tz};
            f.lin = lin;
            blk.warps[lin / 32].alive++;
            if (!g_stack_pool.empty()) { f.stack = g_stack_pool.back(); g_stack_pool.pop_back(); }
            else f.stack = (char*)malloc(STACK_BYTES);
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK_BYTES;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        }
        // round-robin until every fiber has returned
        int remaining = nthreads;
        while (remaining > 0) {
            int progressed = 0;
            for (int i = 0; i < nthreads; i++) {
                Fiber& f = blk.fibers[i];
                if (f.done) continue;
                blk.current = i;
                t_threadIdx = f.tid;
                swapcontext(&blk.sched, &f.ctx);
                if (f.done) { remaining--; }
                progressed++;
            }
            if (progressed == 0) break;
        }
        for (auto& f : blk.fibers) g_stack_pool.push_back(f.stack);
        g_blk = nullptr;
    }
}

void syncthreads()
{
    BlockState* b = g_blk;
    if (!b) return;                       // simple kernel: nothing to wait for
    const unsigned long long my_gen = b->bar_gen;
    b->bar_arrived++;
    if (b->bar_arrived == b->alive) { b->bar_arrived = 0; b->bar_gen++; return; }
    while (b->bar_gen == my_gen) yield_to_scheduler();
}

// returns, for the calling lane, the snapshot of all lanes' values of this collective
static const WarpState& warp_collective(unsigned long long v, int& lane_out, int& slot_out)
{
    BlockState* b = g_blk;
    if (!b) { fprintf(stderr, "shim: warp collective in a kernel registered as simple\n"); abort(); }
    Fiber& f = b->fibers[b->current];
    WarpState& w = b->warps[f.lin / 32];
    const int lane = f.lin & 31;
    w.vals[lane] = v;
    w.present[lane] = 1;
    w.arrived++;
    const unsigned long long my_gen = w.gen;
    if (w.arrived == w.alive) complete_warp_op(w);
    else while (w.gen == my_gen) yield_to_scheduler();
    lane_out = lane;
    slot_out = (int)(my_gen & 1);
    return w;
}

unsigned long long warp_exchange(unsigned long long v, int mode, int arg, int width)
{
    int lane, slot;
    const WarpState& w = warp_collective(v, lane, slot);
    if (width <= 0 || width > 32) width = 32;
    const int base = lane & ~(width - 1), rel = lane & (width - 1);
    int src;
    switch (mode) {
    case 0: src = base + (arg & (width - 1)); break;                         // shfl (idx)
    case 1: src = (rel - arg >= 0) ? lane - arg : lane; break;               // up
    case 2: src = (rel + arg < width) ? lane + arg : lane; break;            // down
    default: { const int t = rel ^ arg; src = (t < width) ? base + t : lane; break; }   // xor
    }
    if (src < 0 || src > 31 || !w.res_present[slot][src]) src = lane;       // inactive source: own value
    return w.res[slot][src];
}

unsigned warp_ballot(int pred)
{
    int lane, slot;
    const WarpState& w = warp_collective(pred ? 1ull : 0ull, lane, slot);
    unsigned m = 0;
    for (int l = 0; l < 32; l++) if (w.res_present[slot][l] && w.res[slot][l]) m |= (1u << l);
    return m;
}

// ------------------------------------------------------------------------------------------------
// Textures
// ------------------------------------------------------------------------------------------------
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float texel(const TexObj* t, int i, int j, int layer)
{
    if (t->is_array) {
        const cudaArray* a = t->arr;
        i = clampi(i, 0, (int)a->w - 1);
        j = clampi(j, 0, (int)a->h - 1);
        layer = clampi(layer, 0, (int)a->depth - 1);
        return a->data[((size_t)layer * a->h + j) * a->w + i];
    }
    i = clampi(i, 0, (int)t->lw - 1);
    j = clampi(j, 0, (int)t->lh - 1);
    const char* row = (const char*)t->lin + (size_t)j * t->lpitch;
    if (t->is_float_elem) return ((const float*)row)[i];
    const float v = (float)((const unsigned char*)row)[i];
    return t->norm_float ? v / 255.0f : v;
}

float tex_fetch2d(const TexObj* t, float x, float y, int layer)
{
    const int W = t->is_array ? (int)t->arr->w : (int)t->lw;
    const int H = t->is_array ? (int)t->arr->h : (int)t->lh;
    if (t->normalized) { x = x * (float)W; y = y * (float)H; }
    if (!t->linear) {
        // point sampling: texel floor(x), floor(y)
        return texel(t, (int)floorf(x), (int)floorf(y), layer);
    }
    // linear filtering (CUDA C Programming Guide, "Linear Filtering"): xB = x - 0.5, i = floor(xB),
    // alpha = frac(xB) in 9-bit fixed point with 8 bits of fractional value
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    float a = xb - fx, b = yb - fy;
    a = rintf(a * 256.0f) * (1.0f / 256.0f);
    b = rintf(b * 256.0f) * (1.0f / 256.0f);
    const int i = (int)fx, j = (int)fy;
    const float t00 = texel(t, i, j, layer), t10 = texel(t, i + 1, j, layer);
    const float t01 = texel(t, i, j + 1, layer), t11 = texel(t, i + 1, j + 1, layer);
    const float r0 = fmaf(a, t10, (1.0f - a) * t00);
    const float r1 = fmaf(a, t11, (1.0f - a) * t01);
    return fmaf(b, r1, (1.0f - b) * r0);
}

} // namespace shim

// ------------------------------------------------------------------------------------------------
// Runtime API: host memory for everything, synchronous execution
// ------------------------------------------------------------------------------------------------
cudaError_t cudaGetDevice(int* d) { *d = 0; return cudaSuccess; }
cudaError_t cudaSetDevice(int) { return cudaSuccess; }
cudaError_t cudaGetDeviceCount(int* n) { *n = 1; return cudaSuccess; }
cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int)
{
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "CUDA emulation on CPU (oracle/ref_shim)");
    p->totalGlobalMem = (size_t)64 << 30;
    p->sharedMemPerBlock = 48 << 10; p->warpSize = 32; p->maxThreadsPerBlock = 1024;
    p->maxThreadsDim[0] = p->maxThreadsDim[1] = 1024; p->maxThreadsDim[2] = 64;
    p->maxGridSize[0] = 2147483647; p->maxGridSize[1] = p->maxGridSize[2] = 65535;
    p->major = 9; p->minor = 9; p->multiProcessorCount = 1; p->maxThreadsPerMultiProcessor = 2048;
    p->maxTexture2D[0] = p->maxTexture2D[1] = 131072;
    p->maxTexture2DLinear[0] = p->maxTexture2DLinear[1] = 131072; p->maxTexture2DLinear[2] = 1 << 30;
    p->maxTexture2DLayered[0] = p->maxTexture2DLayered[1] = 32768; p->maxTexture2DLayered[2] = 2048;
    p->maxSurface2DLayered[0] = p->maxSurface2DLayered[1] = 32768; p->maxSurface2DLayered[2] = 2048;
    p->unifiedAddressing = 1; p->canMapHostMemory = 1; p->concurrentKernels = 1;
    return cudaSuccess;
}
cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
cudaError_t cudaDeviceReset() { return cudaSuccess; }
cudaError_t cudaGetLastError() { return cudaSuccess; }
const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "emulation error"; }
cudaError_t cudaMalloc(void** p, size_t sz) { *p = calloc(1, sz ? sz : 1); return *p ? cudaSuccess : cudaErrorMemoryAllocation; }
cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
cudaError_t cudaMallocHost(void** p, size_t sz) { *p = calloc(1, sz ? sz : 1); return *p ? cudaSuccess : cudaErrorMemoryAllocation; }
cudaError_t cudaFreeHost(void* p) { free(p); return cudaSuccess; }
cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t wbytes, size_t h)
{
    *pitch = (wbytes + 511) & ~(size_t)511;
    *p = calloc(1, *pitch * (h ? h : 1));
    return *p ? cudaSuccess : cudaErrorMemoryAllocation;
}
cudaError_t cudaMemset(void* p, int v, size_t n) { memset(p, v, n); return cudaSuccess; }
cudaError_t cudaMemsetAsync(void* p, int v, size_t n, cudaStream_t) { memset(p, v, n); return cudaSuccess; }
cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memmove(d, s, n); return cudaSuccess; }
cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind, cudaStream_t) { memmove(d, s, n); return cudaSuccess; }
cudaError_t cudaMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t wbytes, size_t h, cudaMemcpyKind)
{
    for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, wbytes);
    return cudaSuccess;
}
cudaError_t cudaMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t wbytes, size_t h, cudaMemcpyKind k, cudaStream_t)
{ return cudaMemcpy2D(d, dp, s, sp, wbytes, h, k); }
cudaError_t cudaMemcpy3D(const cudaMemcpy3DParms* p)
{
    // only array -> host pitched pointer is used by the reference (debug dumps)
    if (p->srcArray && p->dstPtr.ptr) {
        const cudaArray* a = p->srcArray;
        for (size_t z = 0; z < p->extent.depth; z++)
            for (size_t y = 0; y < p->extent.height; y++)
                memcpy((char*)p->dstPtr.ptr + (z * p->dstPtr.ysize + y) * p->dstPtr.pitch,
                       a->data + ((z + p->srcPos.z) * a->h + y + p->srcPos.y) * a->w + p->srcPos.x,
                       p->extent.width * sizeof(float));
        return cudaSuccess;
    }
    return cudaErrorInvalidValue;
}
cudaError_t cudaHostRegister(void*, size_t, unsigned) { return cudaSuccess; }
cudaError_t cudaHostUnregister(void*) { return cudaSuccess; }
cudaError_t cudaStreamCreate(cudaStream_t* s) { *s = malloc(1); return cudaSuccess; }
cudaError_t cudaStreamDestroy(cudaStream_t s) { free(s); return cudaSuccess; }
cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
cudaError_t cudaStreamWaitEvent(cudaStream_t, cudaEvent_t, unsigned) { return cudaSuccess; }
cudaError_t cudaEventCreate(cudaEvent_t* e) { *e = malloc(1); return cudaSuccess; }
cudaError_t cudaEventDestroy(cudaEvent_t e) { free(e); return cudaSuccess; }
cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t) { return cudaSuccess; }
cudaError_t cudaEventSynchronize(cudaEvent_t) { return cudaSuccess; }
cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t, cudaEvent_t) { *ms = 0.0f; return cudaSuccess; }

cudaError_t cudaMalloc3DArray(cudaArray_t* a, const cudaChannelFormatDesc*, cudaExtent e, unsigned)
{
    cudaArray* arr = new cudaArray;
    arr->w = e.width; arr->h = e.height; arr->depth = e.depth ? e.depth : 1;
    arr->data = (float*)calloc(arr->w * arr->h * arr->depth, sizeof(float));
    *a = arr;
    return arr->data ? cudaSuccess : cudaErrorMemoryAllocation;
}
cudaError_t cudaFreeArray(cudaArray_t a) { if (a) { free(a->data); delete a; } return cudaSuccess; }

cudaError_t cudaCreateTextureObject(cudaTextureObject_t* t, const cudaResourceDesc* r, const cudaTextureDesc* d, const cudaResourceViewDesc*)
{
    shim::TexObj* o = new shim::TexObj();
    memset(o, 0, sizeof(*o));
    if (r->resType == cudaResourceTypeArray) { o->is_array = true; o->arr = r->res.array.array; }
    else if (r->resType == cudaResourceTypePitch2D) {
        o->is_array = false;
        o->lin = r->res.pitch2D.devPtr; o->lw = r->res.pitch2D.width; o->lh = r->res.pitch2D.height;
        o->lpitch = r->res.pitch2D.pitchInBytes;
        o->is_float_elem = (r->res.pitch2D.desc.f == cudaChannelFormatKindFloat);
        o->elem_bytes = r->res.pitch2D.desc.x / 8;
    } else { delete o; return cudaErrorInvalidValue; }
    o->normalized = d->normalizedCoords != 0;
    o->linear = d->filterMode == cudaFilterModeLinear;
    o->norm_float = d->readMode == cudaReadModeNormalizedFloat;
    *t = (cudaTextureObject_t)(uintptr_t)o;
    return cudaSuccess;
}
cudaError_t cudaDestroyTextureObject(cudaTextureObject_t t) { delete reinterpret_cast<shim::TexObj*>((uintptr_t)t); return cudaSuccess; }
cudaError_t cudaCreateSurfaceObject(cudaSurfaceObject_t* s, const cudaResourceDesc* r)
{
    shim::TexObj* o = new shim::TexObj();
    memset(o, 0, sizeof(*o));
    o->is_array = true; o->arr = r->res.array.array;
    *s = (cudaSurfaceObject_t)(uintptr_t)o;
    return cudaSuccess;
}
cudaError_t cudaDestroySurfaceObject(cudaSurfaceObject_t s) { delete reinterpret_cast<shim::TexObj*>((uintptr_t)s); return cudaSuccess; }
