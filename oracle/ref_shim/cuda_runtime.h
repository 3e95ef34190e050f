// oracle/ref_shim/cuda_runtime.h -- a CUDA *emulation* for the CPU, just large enough to compile
// and run the reference's own sources (/root/reference/src/popsift) with g++.
//
// TEST INFRASTRUCTURE ONLY.  This is how the oracle is pinned: the reference ships no golden
// vectors, so its real code is executed here on the CPU and compared with oracle/sift_oracle.c.
//
// What is emulated
//   * kernel launches  k<<<grid,block,shm,stream>>>(args)  are rewritten by ref_shim/prep.py into
//     SHIM_LAUNCH("k", grid, block, [&]{ k(args); });  every CUDA thread of a block runs as a
//     ucontext fiber inside the launching OS thread, blocks run one after another;
//   * __syncthreads, __shfl*, __ballot, __any, __all: cooperative barriers between fibers of a
//     block / 32-lane warp (lane = linear thread id % 32, CUDA's linearisation x + y*Dx + z*Dx*Dy);
//   * __shared__  -> static storage (one block at a time), __device__/__constant__ -> globals,
//     cudaMemcpyTo/FromSymbol -> memcpy;
//   * cudaArray (layered), surface writes, texture objects: point / linear filtering, clamp
//     addressing, normalised coordinates and normalised-float reads as documented in the CUDA C
//     Programming Guide ("Texture Fetching": xB = x - 0.5, 1.8 fixed-point weights);
//   * fast-math intrinsics: correctly rounded stand-ins (__expf -> expf ...); __fmul_ru/__fmaf_ru
//     honour round-up through the MXCSR;
//   * streams, events: everything is synchronous.
#pragma once

#include <algorithm>
#include <cfenv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <xmmintrin.h>

// ---- qualifiers ------------------------------------------------------------------------------
#define __host__
#define __device__
#define __global__
#define __constant__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __align__(n) alignas(n)

// ---- vector types ----------------------------------------------------------------------------
struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
    dim3(uint3 v) : x(v.x), y(v.y), z(v.z) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct uchar2 { unsigned char x, y; };
struct uchar3 { unsigned char x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct ushort2 { unsigned short x, y; };
struct short2 { short x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

namespace shim {
extern thread_local uint3 t_threadIdx, t_blockIdx;
extern thread_local dim3 t_blockDim, t_gridDim;
}
#define threadIdx (shim::t_threadIdx)
#define blockIdx  (shim::t_blockIdx)
#define blockDim  (shim::t_blockDim)
#define gridDim   (shim::t_gridDim)
static const int warpSize = 32;

// ---- runtime API types -----------------------------------------------------------------------
typedef int cudaError_t;
enum { cudaSuccess = 0, cudaErrorInvalidValue = 1, cudaErrorMemoryAllocation = 2 };
typedef void* cudaStream_t;
typedef void* cudaEvent_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice, cudaMemcpyDefault };
enum cudaChannelFormatKind { cudaChannelFormatKindSigned, cudaChannelFormatKindUnsigned, cudaChannelFormatKindFloat, cudaChannelFormatKindNone };
struct cudaChannelFormatDesc { int x, y, z, w; cudaChannelFormatKind f; };
struct cudaExtent { size_t width, height, depth; };
struct cudaPos { size_t x, y, z; };
struct cudaPitchedPtr { void* ptr; size_t pitch, xsize, ysize; };
struct cudaArray { float* data; size_t w, h, depth; };
typedef cudaArray* cudaArray_t;
enum { cudaArrayDefault = 0, cudaArrayLayered = 1, cudaArraySurfaceLoadStore = 2 };
enum cudaResourceType { cudaResourceTypeArray, cudaResourceTypeMipmappedArray, cudaResourceTypeLinear, cudaResourceTypePitch2D };
struct cudaResourceDesc {
    cudaResourceType resType;
    struct {
        struct { cudaArray_t array; } array;
        struct { void* devPtr; cudaChannelFormatDesc desc; size_t sizeInBytes; } linear;
        struct { void* devPtr; cudaChannelFormatDesc desc; size_t width, height, pitchInBytes; } pitch2D;
    } res;
};
enum cudaTextureAddressMode { cudaAddressModeWrap, cudaAddressModeClamp, cudaAddressModeMirror, cudaAddressModeBorder };
enum cudaTextureFilterMode { cudaFilterModePoint, cudaFilterModeLinear };
enum cudaTextureReadMode { cudaReadModeElementType, cudaReadModeNormalizedFloat };
struct cudaTextureDesc {
    cudaTextureAddressMode addressMode[3];
    cudaTextureFilterMode filterMode;
    cudaTextureReadMode readMode;
    int sRGB; float borderColor[4]; int normalizedCoords; unsigned maxAnisotropy;
    cudaTextureFilterMode mipmapFilterMode; float mipmapLevelBias, minMipmapLevelClamp, maxMipmapLevelClamp;
};
struct cudaResourceViewDesc;
typedef unsigned long long cudaTextureObject_t;
typedef unsigned long long cudaSurfaceObject_t;
enum cudaSurfaceBoundaryMode { cudaBoundaryModeZero, cudaBoundaryModeClamp, cudaBoundaryModeTrap };
struct cudaMemcpy3DParms {
    cudaArray_t srcArray; cudaPos srcPos; cudaPitchedPtr srcPtr;
    cudaArray_t dstArray; cudaPos dstPos; cudaPitchedPtr dstPtr;
    cudaExtent extent; cudaMemcpyKind kind;
};
struct cudaDeviceProp {
    char name[256]; size_t totalGlobalMem, sharedMemPerBlock; int regsPerBlock, warpSize; size_t memPitch;
    int maxThreadsPerBlock; int maxThreadsDim[3]; int maxGridSize[3]; int clockRate; size_t totalConstMem;
    int major, minor; size_t textureAlignment, texturePitchAlignment; int deviceOverlap, multiProcessorCount;
    int kernelExecTimeoutEnabled, integrated, canMapHostMemory, computeMode;
    int maxTexture1D, maxTexture1DMipmap, maxTexture1DLinear; int maxTexture2D[2], maxTexture2DMipmap[2], maxTexture2DLinear[3];
    int maxTexture2DGather[2]; int maxTexture3D[3], maxTexture3DAlt[3]; int maxTextureCubemap;
    int maxTexture1DLayered[2]; int maxTexture2DLayered[3]; int maxTextureCubemapLayered[2];
    int maxSurface1D; int maxSurface2D[2]; int maxSurface3D[3]; int maxSurface1DLayered[2]; int maxSurface2DLayered[3];
    int maxSurfaceCubemap; int maxSurfaceCubemapLayered[2]; size_t surfaceAlignment;
    int concurrentKernels, ECCEnabled, pciBusID, pciDeviceID, pciDomainID, tccDriver, asyncEngineCount, unifiedAddressing;
    int memoryClockRate, memoryBusWidth, l2CacheSize, maxThreadsPerMultiProcessor, streamPrioritiesSupported;
    int globalL1CacheSupported, localL1CacheSupported; size_t sharedMemPerMultiprocessor; int regsPerMultiprocessor;
    int managedMemory, isMultiGpuBoard, multiGpuBoardGroupID;
};

static inline cudaExtent make_cudaExtent(size_t w, size_t h, size_t d) { return cudaExtent{w, h, d}; }
static inline cudaPos make_cudaPos(size_t x, size_t y, size_t z) { return cudaPos{x, y, z}; }
static inline cudaPitchedPtr make_cudaPitchedPtr(void* p, size_t pitch, size_t xs, size_t ys) { return cudaPitchedPtr{p, pitch, xs, ys}; }
static inline cudaChannelFormatDesc cudaCreateChannelDesc(int x, int y, int z, int w, cudaChannelFormatKind f) { return cudaChannelFormatDesc{x, y, z, w, f}; }

// ---- runtime API (implemented in shim_runtime.cpp) ----------------------------------------------
cudaError_t cudaGetDevice(int* d);
cudaError_t cudaSetDevice(int d);
cudaError_t cudaGetDeviceCount(int* n);
cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int d);
cudaError_t cudaDeviceSynchronize();
cudaError_t cudaDeviceReset();
cudaError_t cudaGetLastError();
const char* cudaGetErrorString(cudaError_t e);
cudaError_t cudaMalloc(void** p, size_t sz);
template <class T> static inline cudaError_t cudaMalloc(T** p, size_t sz) { return cudaMalloc((void**)p, sz); }
cudaError_t cudaFree(void* p);
cudaError_t cudaMallocHost(void** p, size_t sz);
template <class T> static inline cudaError_t cudaMallocHost(T** p, size_t sz) { return cudaMallocHost((void**)p, sz); }
cudaError_t cudaFreeHost(void* p);
cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t wbytes, size_t h);
cudaError_t cudaMemset(void* p, int v, size_t n);
cudaError_t cudaMemsetAsync(void* p, int v, size_t n, cudaStream_t s = nullptr);
cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind k);
cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind k, cudaStream_t st = nullptr);
cudaError_t cudaMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t wbytes, size_t h, cudaMemcpyKind k);
cudaError_t cudaMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t wbytes, size_t h, cudaMemcpyKind k, cudaStream_t st = nullptr);
cudaError_t cudaMemcpy3D(const cudaMemcpy3DParms* p);
cudaError_t cudaHostRegister(void* p, size_t n, unsigned flags);
cudaError_t cudaHostUnregister(void* p);
cudaError_t cudaStreamCreate(cudaStream_t* s);
cudaError_t cudaStreamDestroy(cudaStream_t s);
cudaError_t cudaStreamSynchronize(cudaStream_t s);
cudaError_t cudaStreamWaitEvent(cudaStream_t s, cudaEvent_t e, unsigned flags);
cudaError_t cudaEventCreate(cudaEvent_t* e);
cudaError_t cudaEventDestroy(cudaEvent_t e);
cudaError_t cudaEventRecord(cudaEvent_t e, cudaStream_t s = nullptr);
cudaError_t cudaEventSynchronize(cudaEvent_t e);
cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t a, cudaEvent_t b);
cudaError_t cudaMalloc3DArray(cudaArray_t* a, const cudaChannelFormatDesc* d, cudaExtent e, unsigned flags = 0);
cudaError_t cudaFreeArray(cudaArray_t a);
cudaError_t cudaCreateTextureObject(cudaTextureObject_t* t, const cudaResourceDesc* r, const cudaTextureDesc* d, const cudaResourceViewDesc* v);
cudaError_t cudaDestroyTextureObject(cudaTextureObject_t t);
cudaError_t cudaCreateSurfaceObject(cudaSurfaceObject_t* s, const cudaResourceDesc* r);
cudaError_t cudaDestroySurfaceObject(cudaSurfaceObject_t s);

// symbols are ordinary globals
#define cudaMemcpyToSymbol(sym, src, size, ...)            (memcpy((void*)&(sym), (src), (size)), cudaSuccess)
#define cudaMemcpyToSymbolAsync(sym, src, size, ...)       (memcpy((void*)&(sym), (src), (size)), cudaSuccess)
#define cudaMemcpyFromSymbol(dst, sym, size, ...)          (memcpy((dst), (const void*)&(sym), (size)), cudaSuccess)
#define cudaMemcpyFromSymbolAsync(dst, sym, size, ...)     (memcpy((dst), (const void*)&(sym), (size)), cudaSuccess)

// ---- texture / surface emulation ----------------------------------------------------------------
namespace shim {
struct TexObj {
    bool is_array; cudaArray* arr;                      // layered float array
    const void* lin; size_t lw, lh, lpitch; int elem_bytes; bool is_float_elem;   // pitch2D
    bool normalized, linear, norm_float;
};
float tex_fetch2d(const TexObj* t, float x, float y, int layer);
void launch(const char* name, dim3 grid, dim3 block, const std::function<void()>& body);
void syncthreads();
unsigned long long warp_exchange(unsigned long long v, int mode, int arg, int width);
unsigned warp_ballot(int pred);
}
#define SHIM_LAUNCH(name, grid, block, ...) shim::launch(name, dim3(grid), dim3(block), __VA_ARGS__)

template <class T> static inline T tex2D(cudaTextureObject_t t, float x, float y)
{ return (T)shim::tex_fetch2d(reinterpret_cast<const shim::TexObj*>(t), x, y, 0); }
template <class T> static inline T tex2DLayered(cudaTextureObject_t t, float x, float y, int layer)
{ return (T)shim::tex_fetch2d(reinterpret_cast<const shim::TexObj*>(t), x, y, layer); }
template <class T> static inline void surf2DLayeredwrite(T val, cudaSurfaceObject_t s, int xbytes, int y, int layer, cudaSurfaceBoundaryMode = cudaBoundaryModeTrap)
{
    const shim::TexObj* o = reinterpret_cast<const shim::TexObj*>(s);
    const int x = xbytes / (int)sizeof(T);
    if (x < 0 || y < 0 || layer < 0 || (size_t)x >= o->arr->w || (size_t)y >= o->arr->h || (size_t)layer >= o->arr->depth) return;
    o->arr->data[((size_t)layer * o->arr->h + y) * o->arr->w + x] = (float)val;
}

// ---- device intrinsics --------------------------------------------------------------------------
static inline void __syncthreads() { shim::syncthreads(); }
template <class T> static inline T shim_xchg(T v, int mode, int arg, int width)
{
    static_assert(sizeof(T) <= 8, "shuffle of <= 8 bytes");
    unsigned long long raw = 0; memcpy(&raw, &v, sizeof(T));
    raw = shim::warp_exchange(raw, mode, arg, width);
    T out; memcpy(&out, &raw, sizeof(T)); return out;
}
template <class T> static inline T __shfl(T v, int src, int width = 32)        { return shim_xchg(v, 0, src, width); }
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 32)  { return shim_xchg(v, 1, (int)d, width); }
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 32){ return shim_xchg(v, 2, (int)d, width); }
template <class T> static inline T __shfl_xor(T v, int m, int width = 32)      { return shim_xchg(v, 3, m, width); }
static inline unsigned __ballot(int pred) { return shim::warp_ballot(pred); }
static inline int __any(int pred) { return shim::warp_ballot(pred) != 0u; }
static inline int __all(int pred) { return shim::warp_ballot(!pred) == 0u; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }

// -DOSIFT_CUDA_MODEL (make -C oracle ref_model; device translation units only): every stand-in below moves its correctly
// rounded result by an error inside the bound the CUDA Programming Guide documents for the function it stands for
// (oracle/cuda_model.h, selected at run time by the environment variable OSIFT_CUDA_MODEL); without the define the CM_* macros are
// the plain operations
#include "../cuda_model.h"
// glibc's <math.h> declares __expf / __sincosf as its own internal entry points: rename ours
static inline float shim_expf_(float x) { return CM_FAST_EXPF(x); }
// evaluated in double and rounded once: the grid descriptor's pixel snapping (s_desc_grid.cu:72-78) flips on the last
// bit of sin / cos, so the stand-in must not depend on libm's float sinf / cosf rounding (oracle/sift_oracle.c does the same)
static inline void shim_sincosf_(float a, float* s, float* c) { *s = CM_FAST_SIN((float)sin((double)a)); *c = CM_FAST_COS((float)cos((double)a)); }
#define __expf shim_expf_
#define __sincosf shim_sincosf_
// CUDA's atan2f (<= 2 ulp, unspecified beyond that) picks the orientation-histogram bin through
// roundf(36 (theta + pi) / 2 pi) (s_orientation.cu:148-151): on exactly diagonal gradients (binary images, symmetric
// patterns) the bin hangs on theta's last ulp.  The stand-in is the correctly rounded value -- atan2 in double,
// rounded once -- which the oracle and the HIP kernel use too, so that all three sides take the same bin.
static inline float shim_atan2f_(float y, float x) { return CM_ATAN2F((float)atan2((double)y, (double)x)); }
#define atan2f shim_atan2f_
#ifdef OSIFT_CUDA_MODEL
// device expf / hypotf (libdevice: 2 / 3 ulp); the standard build leaves them to libm
static inline float shim_model_expf_(float x) { return CM_EXPF(x); }
static inline float shim_model_hypotf_(float a, float b) { return CM_HYPOTF(a, b); }
#define expf shim_model_expf_
#define hypotf shim_model_hypotf_
#endif
static inline float __fdividef(float a, float b) { return CM_FDIVIDEF(a, b); }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __fsqrt_rz(float a) { return sqrtf(a); }
static inline float __frsqrt_rn(float a) { return CM_FRSQRT_RN(a); }
static __attribute__((noinline)) float __fmul_ru(float a, float b)
{
    volatile float va = a, vb = b;
    const unsigned csr = _mm_getcsr();
    _mm_setcsr((csr & ~_MM_ROUND_MASK) | _MM_ROUND_UP);
    volatile float r = va * vb;
    _mm_setcsr(csr);
    return r;
}
static __attribute__((noinline)) float __fmaf_ru(float a, float b, float c)
{
    volatile float va = a, vb = b, vc = c;
    const unsigned csr = _mm_getcsr();
    _mm_setcsr((csr & ~_MM_ROUND_MASK) | _MM_ROUND_UP);
    volatile float r = __builtin_fmaf(va, vb, vc);
    _mm_setcsr(csr);
    return r;
}
using std::min;
using std::max;
static inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }
static inline int max(int a, unsigned b) { return a > (int)b ? a : (int)b; }
static inline int min(unsigned a, int b) { return (int)a < b ? (int)a : b; }
static inline int max(unsigned a, int b) { return (int)a > b ? (int)a : b; }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
