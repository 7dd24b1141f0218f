// Host allocations of THIS library on transparent huge pages.
//
// Handle creation builds ~1.5 GB of host tables at 100^3 (index images, block / tile tables, tile lists, pair maps, scatter positions); every one above glibc's
// mmap threshold is a fresh mapping whose first touch takes a 4 KB page fault per page -- a third of the planner's time on the MI355X hosts
// (profiles/r05_setup.txt: setup 0.82 -> 0.63 s with the process-wide glibc.malloc.hugetlb=1 tunable).  A library cannot set a tunable of its host process, so
// it asks for the same thing for its own blocks: operator new of this shared object (made LOCAL by the link's version script, sluamd.map: the replacement binds
// the library's own calls only, nobody else's) returns blocks of 4 MB and more 2 MB-aligned and madvise(MADV_HUGEPAGE)d before their first touch; everything comes from malloc's
// arena and goes back through free(), so a block may cross into code that uses the default operators.  Kernels with THP "never" ignore the advice;
// SLUAMD_NO_THP=1 turns it off.
//
// CONSTRAINT: blocks cross between this object's inline code and out-of-line libstdc++ code (std::string, std::thread state, exceptions), which
// uses the PROCESS-global operators -- so the process-global operator new / delete must be malloc / free compatible (glibc's, tcmalloc's, jemalloc's and
// mimalloc's replacements of malloc all are; a host application that replaces operator new with an allocator whose blocks free() cannot take must build this
// library with -DSLUAMD_NO_LOCAL_NEW, which compiles this file to nothing).  The replacements must stay LOCAL: the Makefile fails the link when one of them
// shows up in the dynamic symbol table (no GNU --version-script), and tests/test_abi.py checks the shipped object.  Over-aligned (align_val_t) requests are
// not routed here: nothing in the library makes them, the global operators serve them and their deletes.
#ifndef SLUAMD_NO_LOCAL_NEW
#include <cstdlib>
#include <new>
#include <sys/mman.h>

namespace {
constexpr std::size_t kHuge = std::size_t(2) << 20, kBig = std::size_t(4) << 20;
inline bool thp_on() { static const bool on = getenv("SLUAMD_NO_THP") == nullptr; return on; }
inline void *host_alloc(std::size_t n) noexcept
{
    if (n >= kBig && thp_on()) {
        const std::size_t r = (n + kHuge - 1) & ~(kHuge - 1);
        if (void *p = aligned_alloc(kHuge, r)) { madvise(p, r, MADV_HUGEPAGE); return p; }
    }
    return malloc(n ? n : 1);
}
}  // namespace

#define SLUAMD_LOCAL
SLUAMD_LOCAL void *operator new(std::size_t n) { if (void *p = host_alloc(n)) return p; throw std::bad_alloc(); }
SLUAMD_LOCAL void *operator new[](std::size_t n) { if (void *p = host_alloc(n)) return p; throw std::bad_alloc(); }
SLUAMD_LOCAL void *operator new(std::size_t n, const std::nothrow_t &) noexcept { return host_alloc(n); }
SLUAMD_LOCAL void *operator new[](std::size_t n, const std::nothrow_t &) noexcept { return host_alloc(n); }
SLUAMD_LOCAL void operator delete(void *p) noexcept { free(p); }
SLUAMD_LOCAL void operator delete[](void *p) noexcept { free(p); }
SLUAMD_LOCAL void operator delete(void *p, std::size_t) noexcept { free(p); }
SLUAMD_LOCAL void operator delete[](void *p, std::size_t) noexcept { free(p); }
SLUAMD_LOCAL void operator delete(void *p, const std::nothrow_t &) noexcept { free(p); }
SLUAMD_LOCAL void operator delete[](void *p, const std::nothrow_t &) noexcept { free(p); }
#endif
