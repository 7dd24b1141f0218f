// oracle/emul/sluamd_rt.h -- TEST INFRASTRUCTURE.  Host stand-in for the handful of HIP runtime calls the library's HOST
// files make (memory, streams, events), so that the same planning / orchestration sources (superlu_dist_amd/csrc/
// sluamd_{host,plan,factor,api,comm,symb}.cpp) can be linked against the serial CPU engine (engine_cpu.cpp) into
// oracle/libsluamd_emul.so.  "Device" memory is host memory.  Streams (emul_rt.cpp) either execute immediately (default) or
// -- sluamd_emul_sched(mode, seed) -- queue their work and run it in a seeded adversarial order that honours only what HIP
// guarantees (stream order, event waits, the synchronising calls): a dependency the drivers forgot to express then shows up
// as a wrong result on CPU.  The product library never includes this header (its include path finds
// superlu_dist_amd/csrc/sluamd_rt.h = <hip/hip_runtime.h>).
#pragma once
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
inline const char *hipGetErrorString(hipError_t) { return "emulated HIP error"; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
struct emul_stream_s;
typedef emul_stream_s *hipStream_t;
struct emul_event_s;
typedef emul_event_s *hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct int2 { int x, y; };
inline int2 make_int2(int x, int y) { int2 v = {x, y}; return v; }
struct int4 { int x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { int4 v = {x, y, z, w}; return v; }

// a "kernel launch" of the CPU engine: runs now (immediate mode) or when the scheduler gets to it
void emul_enqueue(hipStream_t s, std::function<void()> f);
// ... that may run only once ready() holds (a condition established by another rank's thread: the emulated stream-ordered transport)
void emul_enqueue_when(hipStream_t s, std::function<bool()> ready, std::function<void()> f);
// adversarial modes: a permutation seed for the work units INSIDE one launch (0 in immediate mode = issue order)
unsigned emul_launch_seed();

hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t *s);
struct hipDeviceProp_t { int multiProcessorCount; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 1; return hipSuccess; }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned, const unsigned *);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int);
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned);
hipError_t hipEventCreate(hipEvent_t *e);
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipEventDestroy(hipEvent_t e);
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = *tot = (size_t) 1 << 40; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceGetPCIBusId(char *buf, int len, int) { if (len > 0) buf[0] = 0; return hipErrorUnknown; }     // (no device behind the emulation)
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipDeviceSynchronize();
