// tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded emulator of the subset of HIP that trust4_amd/csrc uses, so that the very
// same kernel sources can be compiled with g++ and executed (slowly) on the CPU inside the
// `-m "not gpu"` test-suite: every lane of a workgroup is a fiber (hand-rolled x86-64 context switch), wave intrinsics and
// __syncthreads() are rendezvous points. It exists because the development container has no GPU;
// it is NOT a fallback: the product library (trust4_amd/libt4hip.so) is built by hipcc from the
// same sources and never links this header. Built only by tests/hipemu/build_emu.py.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <chrono>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local   /* one workgroup at a time per host worker thread */
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
static inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = {x, y}; return r; }
static inline int4 make_int4(int x, int y, int z, int w) { int4 r = {x, y, z, w}; return r; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef struct hipemu_stream *hipStream_t;
typedef struct hipemu_event { std::chrono::steady_clock::time_point t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; size_t totalGlobalMem; };

namespace hipemu {
struct Tid { unsigned x, y, z; };
extern thread_local Tid cur_tid, cur_bid;
extern thread_local dim3 cur_bdim, cur_gdim;
extern thread_local uint64_t xchg[1024];
extern thread_local int nthreads, nalive;
void yield_lane(void *site = nullptr);
void block_barrier(void *site);
void wave_rendezvous(void *site);
void run_grid(dim3 grid, dim3 block, const std::function<void()> &body);
}  // namespace hipemu

#define threadIdx (hipemu::cur_tid)
#define blockIdx (hipemu::cur_bid)
#define blockDim (hipemu::cur_bdim)
#define gridDim (hipemu::cur_gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::run_grid(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::block_barrier(__builtin_return_address(0)); }

// ---- wave intrinsics (wave == the 64 consecutive threads the caller belongs to)
static inline int hipemu_lane() { return hipemu::cur_tid.x & 63; }
static inline int hipemu_wbase() { return hipemu::cur_tid.x & ~63u; }
template <class T> static inline T hipemu_shfl_any(T v, int src) {
  static_assert(sizeof(T) <= 8, "shfl size");
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  hipemu::xchg[hipemu::cur_tid.x] = raw;
  hipemu::wave_rendezvous(nullptr);
  uint64_t got = hipemu::xchg[hipemu_wbase() + (src & 63)];
  hipemu::wave_rendezvous(nullptr);
  T r; memcpy(&r, &got, sizeof(T));
  return r;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) { (void)width; return hipemu_shfl_any(v, src); }
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  (void)width; int l = hipemu_lane(); return hipemu_shfl_any(v, l >= (int)d ? l - (int)d : l);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  (void)width; int l = hipemu_lane(); return hipemu_shfl_any(v, l + (int)d < 64 ? l + (int)d : l);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) { (void)width; return hipemu_shfl_any(v, hipemu_lane() ^ m); }
// wave-level ordering of LDS traffic: a rendezvous of the wave's fibers
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_rendezvous(__builtin_return_address(0)); }
// DPP wave shifts by one lane (the only dpp_ctrl values the kernels use)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
  if (old != src) {   // edge lanes receive `old`
    const int l = hipemu_lane();
    if (ctrl == 0x111) { int v = hipemu_shfl_any(src, (l & 15) ? l - 1 : l); return (l & 15) ? v : old; }
    if (ctrl == 0x101) { int v = hipemu_shfl_any(src, (l & 15) != 15 ? l + 1 : l); return (l & 15) != 15 ? v : old; }
    if (ctrl == 0x138) { int v = __shfl_up(src, 1); return l ? v : old; }
    if (ctrl == 0x130) { int v = __shfl_down(src, 1); return l != 63 ? v : old; }
  }
  if (ctrl == 0x138) return __shfl_up(src, 1);
  if (ctrl == 0x130) return __shfl_down(src, 1);
  if (ctrl == 0x111) { int l = hipemu_lane(); return hipemu_shfl_any(src, (l & 15) ? l - 1 : l); }        // row_shr:1
  if (ctrl == 0x101) { int l = hipemu_lane(); return hipemu_shfl_any(src, (l & 15) != 15 ? l + 1 : l); }  // row_shl:1
  fprintf(stderr, "hipemu: dpp_ctrl 0x%x not emulated\n", ctrl); abort();
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return hipemu_shfl_any(v, 0); }   // called with all lanes active
static inline unsigned long long __ballot(int pred) {
  hipemu::xchg[hipemu::cur_tid.x] = pred ? 1 : 0;
  hipemu::wave_rendezvous(__builtin_return_address(0));
  unsigned long long m = 0; int b = hipemu_wbase();
  for (int i = 0; i < 64 && b + i < hipemu::nthreads; ++i) if (hipemu::xchg[b + i]) m |= 1ull << i;
  hipemu::wave_rendezvous(nullptr);
  return m;
}
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { return __ballot(!p) == 0; }
static inline unsigned long long __brevll(unsigned long long x) { unsigned long long r = 0; for (int i = 0; i < 64; ++i) if (x >> i & 1) r |= 1ull << (63 - i); return r; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
static inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }

// workgroups of one launch run on several host threads: real atomics (LDS atomics pay for it too, harmlessly)
template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicCAS(T *p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }
template <class T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() {}
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- host API
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  memset(p, 0, sizeof(*p)); p->multiProcessorCount = 8; strcpy(p->name, "hipemu"); strcpy(p->gcnArchName, "emu");
  p->totalGlobalMem = 1ull << 34; return hipSuccess;
}
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) { *p = (T *)calloc(1, n ? n : 1); return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   /* launches run to completion at once */
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
#define hipHostMallocMapped 2
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
