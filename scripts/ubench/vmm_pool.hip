// Does a pool of physical chunks behind a virtual-address reservation avoid what large hipMalloc calls pay on these boxes (device memory cleared when it is handed
// out again, ~35 GB/s: NOTEBOOK.md section 5, round 5)?  hipcc --offload-arch=gfx950 -O2 vmm_pool.hip -o vmm_pool && ./vmm_pool [GB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <thread>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(double *p, size_t n, double v) { size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; for (; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = v; }
int main(int argc, char **argv)
{
    const size_t GB = argc > 1 ? atoi(argv[1]) : 60;
    const size_t bytes = GB << 30;
    CK(hipSetDevice(0));
    CK(hipFree(0));
    for (int rep = 0; rep < 3; ++rep) {
        double t = now(); void *p = nullptr; CK(hipMalloc(&p, bytes)); double t1 = now();
        fill<<<4096, 256>>>((double *) p, bytes / 8, 1.0); CK(hipDeviceSynchronize()); double t2 = now();
        CK(hipFree(p)); double t3 = now();
        printf("hipMalloc %zu GB: %.3f s, fill %.3f s, free %.3f s\n", GB, t1 - t, t2 - t1, t3 - t2);
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t chunk = ((size_t) 2 << 30);
    printf("granularity %zu, chunk %zu\n", gran, chunk);
    const size_t nch = bytes / chunk;
    std::vector<hipMemGenericAllocationHandle_t> hs(nch);
    for (int rep = 0; rep < 3; ++rep) {
        double t = now();
        if (rep == 0) for (size_t i = 0; i < nch; ++i) CK(hipMemCreate(&hs[i], chunk, &prop, 0));
        double t1 = now();
        void *va = nullptr; CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
        for (size_t i = 0; i < nch; ++i) CK(hipMemMap((char *) va + i * chunk, chunk, 0, hs[i], 0));
        hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, bytes, &ad, 1));
        double t2 = now();
        fill<<<4096, 256>>>((double *) va, bytes / 8, 2.0); CK(hipDeviceSynchronize()); double t3 = now();
        CK(hipMemUnmap(va, bytes)); CK(hipMemAddressFree(va, bytes)); double t4 = now();
        printf("VMM rep %d: create %.3f s, reserve+map+access %.3f s, fill %.3f s, unmap %.3f s\n", rep, t1 - t, t2 - t1, t3 - t2, t4 - t3);
    }
    for (auto h : hs) CK(hipMemRelease(h));
    // do the chunks' first allocations (the driver's clearing of re-used memory) run in parallel when several host threads create them?
    for (int nth : {1, 4, 16}) {
        std::vector<hipMemGenericAllocationHandle_t> h2(nch);
        std::vector<std::thread> th;
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        double t = now();
        for (int q = 0; q < nth; ++q) th.emplace_back([&] { hipSetDevice(0); for (;;) { size_t i = next.fetch_add(1); if (i >= nch) return; if (hipMemCreate(&h2[i], chunk, &prop, 0) != hipSuccess) bad++; } });
        for (auto &x : th) x.join();
        double t1 = now();
        printf("hipMemCreate of %zu chunks from %d threads: %.3f s (%d failed)\n", nch, nth, t1 - t, (int) bad);
        for (size_t i = 0; i < nch; ++i) hipMemRelease(h2[i]);
        // dirty the memory again through a plain allocation so that the next round starts from re-used pages
        void *p = nullptr; if (hipMalloc(&p, bytes) == hipSuccess) { fill<<<4096, 256>>>((double *) p, bytes / 8, 3.0); hipDeviceSynchronize(); hipFree(p); }
    }
    { double t = now(); void *p = nullptr; CK(hipMalloc(&p, bytes)); double t1 = now(); CK(hipFree(p)); printf("hipMalloc after release: %.3f s\n", t1 - t); }
    return 0;
}
