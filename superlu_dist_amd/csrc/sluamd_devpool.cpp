// sluamd_devpool.cpp -- the value arenas of single-rank handles on a process-level pool of PHYSICAL device chunks behind virtual-address reservations
// (hipMemCreate / hipMemAddressReserve / hipMemMap), so that the second and later handles of a process do not pay for what large hipMalloc calls cost on
// these boxes: device memory is cleared when it is handed out again (profiles/r06_ubench_vmm_pool.txt: 200 GB hipMalloc 6.0-7.9 s EVERY time; the same
// bytes as chunks pay once, re-mapping recycled chunks costs 1-4 ms).  A destroyed handle's chunks go back to the pool, not to the driver; the next arena
// maps them under a fresh reservation and creates only what is missing.  VERDICT r5 item 5.
//
// Scope: arenas of 1 x 1 x 1 handles only (grid handles keep plain hipMalloc: their ranges are RCCL send / receive buffers, and whether RCCL's peer paths
// take mapped memory could not be tested on a multi-GPU box).  Everything falls back to hipMalloc when a VMM call fails (SLUAMD_NO_DEVPOOL=1 forces that).
// The pool holds memory the process does not use: sluamd_device_pool_trim() returns it to the driver, and the library trims by itself and retries when one
// of its own allocations fails.  The CPU test build has its own trivial version (oracle/emul/emul_rt.cpp).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include "sluamd_internal.h"

namespace sluamd {
namespace {
constexpr size_t CHUNK = (size_t) 1 << 30;
struct Live { size_t bytes; int device; std::vector<hipMemGenericAllocationHandle_t> chunks; };
std::mutex g_mu;
std::map<int, std::vector<hipMemGenericAllocationHandle_t>> g_free;     // per device: chunks no arena maps
std::map<void *, Live> g_live;
bool pool_off() { static const bool off = getenv("SLUAMD_NO_DEVPOOL") != nullptr; return off; }
}  // namespace

int devpool_alloc(void **p, size_t bytes, int device)
{
    *p = nullptr;
    if (pool_off() || bytes < CHUNK) return hipMalloc(p, std::max<size_t>(bytes, 1)) == hipSuccess ? 0 : SLUAMD_ENOMEM;      // (small arenas: nothing to win)
    const size_t nch = (bytes + CHUNK - 1) / CHUNK;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &fr = g_free[device];
        while (chunks.size() < nch && !fr.empty()) { chunks.push_back(fr.back()); fr.pop_back(); }
    }
    auto give_back = [&] { std::lock_guard<std::mutex> lk(g_mu); auto &fr = g_free[device]; fr.insert(fr.end(), chunks.begin(), chunks.end()); chunks.clear(); };
    bool ok = true;
    while (ok && chunks.size() < nch) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, CHUNK, &prop, 0) != hipSuccess) { (void) hipGetLastError(); ok = false; break; }
        chunks.push_back(h);
    }
    void *va = nullptr;
    if (ok && hipMemAddressReserve(&va, nch * CHUNK, 0, nullptr, 0) != hipSuccess) { (void) hipGetLastError(); ok = false; va = nullptr; }
    size_t mapped = 0;
    if (ok)
        for (; mapped < nch; ++mapped)
            if (hipMemMap((char *) va + mapped * CHUNK, CHUNK, 0, chunks[mapped], 0) != hipSuccess) { (void) hipGetLastError(); ok = false; break; }
    if (ok) {
        hipMemAccessDesc ad = {};
        ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(va, nch * CHUNK, &ad, 1) != hipSuccess) { (void) hipGetLastError(); ok = false; }
    }
    if (!ok) {
        if (va) { if (mapped) hipMemUnmap(va, mapped * CHUNK); hipMemAddressFree(va, nch * CHUNK); }
        give_back();
        devpool_trim(device);         // whatever the pool holds goes back to the driver before the plain allocation is tried
        return hipMalloc(p, bytes) == hipSuccess ? 0 : SLUAMD_ENOMEM;
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_live[va] = Live{nch * CHUNK, device, std::move(chunks)};
    }
    *p = va;
    return 0;
}

void devpool_free(void *p)
{
    if (!p) return;
    Live l;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) { hipFree(p); return; }      // a plain allocation (pool off / small / fallback)
        l = std::move(it->second);
        g_live.erase(it);
    }
    hipDeviceSynchronize();
    hipMemUnmap(p, l.bytes);
    hipMemAddressFree(p, l.bytes);
    std::lock_guard<std::mutex> lk(g_mu);
    auto &fr = g_free[l.device];
    fr.insert(fr.end(), l.chunks.begin(), l.chunks.end());
}

size_t devpool_cached_bytes(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    size_t n = 0;
    for (auto &kv : g_free) if (device < 0 || kv.first == device) n += kv.second.size();
    return n * CHUNK;
}

void devpool_trim(int device)
{
    std::vector<hipMemGenericAllocationHandle_t> rel;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto &kv : g_free) if (device < 0 || kv.first == device) { rel.insert(rel.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
    }
    for (auto h : rel) hipMemRelease(h);
}

}  // namespace sluamd
