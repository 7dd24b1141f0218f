// oracle/emul/sluamd_rt.h -- TEST INFRASTRUCTURE.  Host stand-in for the handful of HIP runtime calls the library's HOST
// files make (memory, streams, events), so that the same planning / orchestration sources (superlu_dist_amd/csrc/
// sluamd_{host,plan,factor,api,comm,symb}.cpp) can be linked against the serial CPU engine (engine_cpu.cpp) into
// oracle/libsluamd_emul.so.  "Device" memory is host memory, streams execute immediately.  The product library never
// includes this header (its include path finds superlu_dist_amd/csrc/sluamd_rt.h = <hip/hip_runtime.h>).
#pragma once
#include <chrono>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
inline const char *hipGetErrorString(hipError_t) { return "emulated HIP error"; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
typedef struct emul_stream_s *hipStream_t;
struct emul_event_s { std::chrono::steady_clock::time_point t; };
typedef emul_event_s *hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct int2 { int x, y; };
inline int2 make_int2(int x, int y) { int2 v = {x, y}; return v; }
struct int4 { int x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { int4 v = {x, y, z, w}; return v; }

inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 1; return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned, const unsigned *) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emul_event_s(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
