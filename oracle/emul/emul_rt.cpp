// oracle/emul/emul_rt.cpp -- TEST INFRASTRUCTURE: the emulated HIP runtime behind oracle/emul/sluamd_rt.h.
//
// Mode 0 (default): every stream operation executes at the call, in issue order -- what the CPU test build always did.
// Modes 1-3 (sluamd_emul_sched): operations are queued per stream and executed only when a synchronising call needs them, in an
// order chosen by a seeded scheduler that honours exactly what HIP guarantees:
//   * operations of one stream run in issue order;
//   * hipStreamWaitEvent(s, e) holds s until the work recorded by the LAST hipEventRecord(e) before the wait has run;
//   * hipStreamSynchronize / hipEventSynchronize / hipDeviceSynchronize / hipFree return after the work they name;
//   * synchronous hipMemcpy / hipMemset order themselves against the null stream and the BLOCKING streams only
//     (hipStreamNonBlocking streams are not implicitly synchronised);
//   * hipMemcpyAsync from pageable host memory reads its source at the call, from pinned memory when it runs; to pageable host
//     memory it completes at the call.
// Everything else is free: mode 1 picks a random runnable stream at every step, mode 2 runs only what the synchronising call
// transitively needs (other streams stay behind as long as possible), mode 3 runs every other runnable stream before the one
// being waited for; the work units inside one launch (Schur tiles, update units) run in a shuffled order.  A missing event wait or a host read without synchronisation in the drivers (look-ahead schedule of
// pdgstrf3d, panel exchanges, sweeps) becomes a wrong result under one of these orders -- tests/test_stream_order.py.
#include <atomic>
#include <cstdio>
#include <deque>
#include <map>
#include <sys/mman.h>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <vector>
#include "sluamd_rt.h"

struct emul_op {
    std::function<void()> f;            // null: a wait
    emul_stream_s *wait_on = nullptr;
    unsigned long long wait_seq = 0;
    unsigned long long issue = 0;       // global issue index (statistics)
    std::function<bool()> ready;        // optional: an EXTERNAL condition (a message of another rank's thread: comm_emul_stream.cpp)
};
struct emul_stream_s {
    std::deque<emul_op> q;
    unsigned long long issued = 0, done = 0;
    bool nonblocking = false;
    int owner = 0;                      // the thread (= rank of an in-process grid) that created it: "the device" of that rank
};
struct emul_event_s {
    emul_stream_s *st = nullptr;        // stream of the last record (deferred modes)
    unsigned long long seq = 0;         // ... complete once st->done >= seq
    std::shared_ptr<std::chrono::steady_clock::time_point> t = std::make_shared<std::chrono::steady_clock::time_point>();
};

namespace {
std::recursive_mutex g_mu;
std::vector<emul_stream_s *> g_streams;            // streams are never freed: queued waits keep pointers to them
// Ranks of an in-process grid are threads; on a real node each rank has its own device, so "the device" a synchronising call waits
// for (hipDeviceSynchronize, hipFree, the null stream) is the set of streams the CALLING thread created -- never another rank's
std::atomic<int> g_next_tid{1};
thread_local int t_tid = 0;
thread_local emul_stream_s *t_null = nullptr;
int my_tid() { if (!t_tid) t_tid = g_next_tid++; return t_tid; }
std::map<const char *, size_t> g_pinned;
int g_mode = 0;
std::mt19937 g_rng(1);
unsigned long long g_issue = 0, g_max_run = 0, g_reordered = 0, g_run = 0;

// SLUAMD_EMUL_SCHED="mode,seed" in the environment: the whole process starts under that schedule (e.g. the complete CPU test suite)
struct EnvInit {
    EnvInit()
    {
        if (const char *v = std::getenv("SLUAMD_EMUL_SCHED")) {
            int m = 0; unsigned sd = 1;
            if (std::sscanf(v, "%d,%u", &m, &sd) >= 1) { g_mode = m; g_rng.seed(sd ? sd : 1u); }
        }
    }
} g_env_init;

bool runnable(const emul_op &o) { return (!o.wait_on || o.wait_on->done >= o.wait_seq) && (!o.ready || o.ready()); }

// every entry point holds g_mu through a Guard; a thread that has to wait for ANOTHER thread (a message, see `ready`) lets go of
// all its recursion levels for a moment
thread_local int t_depth = 0;
struct Guard {
    Guard() { g_mu.lock(); ++t_depth; }
    ~Guard() { --t_depth; g_mu.unlock(); }
};
void yield_all()
{
    const int d = t_depth;
    for (int i = 0; i < d; ++i) g_mu.unlock();
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    for (int i = 0; i < d; ++i) g_mu.lock();
}

void step(emul_stream_s *st)
{
    emul_op o = std::move(st->q.front());
    st->q.pop_front();
    if (o.f) {
        if (o.issue < g_max_run) ++g_reordered; else g_max_run = o.issue;
        ++g_run;
        o.f();
    }
    ++st->done;
}

// run queued work until pred() holds; `target` = the stream the caller waits for (may be null)
template <class Pred>
void flush(emul_stream_s *target, Pred pred)
{
    long spins = 0;
    while (!pred()) {
        std::vector<emul_stream_s *> cand;
        bool pending = false;
        for (auto *st : g_streams)
            if (!st->q.empty()) { pending = true; if (runnable(st->q.front())) cand.push_back(st); }
        if (cand.empty()) {
            if (!pending) break;
            bool external = false;
            for (auto *st : g_streams)
                if (!st->q.empty() && st->q.front().ready && (!st->q.front().wait_on || st->q.front().wait_on->done >= st->q.front().wait_seq)) external = true;
            if (!external || ++spins > 3000000) { std::fprintf(stderr, "emulated HIP runtime: streams wait for each other (deadlock)\n"); std::abort(); }
            yield_all();      // another rank's thread has to publish / take a message first
            continue;
        }
        emul_stream_s *pick = nullptr;
        if (g_mode == 2 && target) {          // only what the caller needs: follow the chain of unsatisfied waits
            emul_stream_s *cur = target;
            for (int hop = 0; hop < 64 && cur && !cur->q.empty() && !runnable(cur->q.front()); ++hop) cur = cur->q.front().wait_on;
            if (cur && !cur->q.empty() && runnable(cur->q.front())) pick = cur;
        } else if (g_mode == 3 && target) {   // everybody else first
            std::vector<emul_stream_s *> others;
            for (auto *st : cand) if (st != target) others.push_back(st);
            if (!others.empty()) pick = others[g_rng() % others.size()];
        }
        if (!pick) pick = cand[g_rng() % cand.size()];
        step(pick);
    }
}
bool all_empty() { const int me = my_tid(); for (auto *st : g_streams) if (st->owner == me && !st->q.empty()) return false; return true; }
bool everything_empty() { for (auto *st : g_streams) if (!st->q.empty()) return false; return true; }
bool blocking_empty() { const int me = my_tid(); for (auto *st : g_streams) if (st->owner == me && !st->nonblocking && !st->q.empty()) return false; return true; }
bool pinned(const void *p)
{
    auto it = g_pinned.upper_bound((const char *) p);
    if (it == g_pinned.begin()) return false;
    --it;
    return (const char *) p < it->first + it->second;
}
hipStream_t new_stream(bool nonblocking)
{
    Guard lk;
    auto *s = new emul_stream_s();
    s->nonblocking = nonblocking;
    s->owner = my_tid();
    g_streams.push_back(s);
    return s;
}
emul_stream_s *null_stream()      // the calling thread's legacy default stream (call with the lock held)
{
    if (!t_null) { t_null = new emul_stream_s(); t_null->owner = my_tid(); g_streams.push_back(t_null); }
    return t_null;
}
}  // namespace

void emul_enqueue(hipStream_t s, std::function<void()> f)
{
    Guard lk;
    if (g_mode == 0) { f(); return; }
    emul_stream_s *st = s ? s : null_stream();
    if (st == t_null) flush(nullptr, blocking_empty);     // legacy null stream: after everything queued on the blocking streams
    emul_op o; o.f = std::move(f); o.issue = ++g_issue;
    st->q.push_back(std::move(o)); ++st->issued;
    if (st == t_null) flush(st, [&] { return st->q.empty(); });
}

unsigned emul_launch_seed()
{
    Guard lk;
    return g_mode == 0 ? 0u : (unsigned) (g_rng() | 1u);
}

// an operation that may run only once ready() holds -- a condition another rank's thread establishes (message passing of the emulated
// stream-ordered transport).  Immediate mode: the calling thread waits here, without the lock.
void emul_enqueue_when(hipStream_t s, std::function<bool()> ready, std::function<void()> f)
{
    Guard lk;
    if (g_mode == 0) {
        long spins = 0;
        while (!ready()) { if (++spins > 3000000) { std::fprintf(stderr, "emulated HIP runtime: message never arrived (deadlock)\n"); std::abort(); } yield_all(); }
        f();
        return;
    }
    emul_stream_s *st = s ? s : null_stream();
    if (st == t_null) flush(nullptr, blocking_empty);
    emul_op o; o.f = std::move(f); o.ready = std::move(ready); o.issue = ++g_issue;
    st->q.push_back(std::move(o)); ++st->issued;
    if (st == t_null) flush(st, [&] { return st->q.empty(); });
}

extern "C" void sluamd_emul_sched(int mode, unsigned seed)
{
    Guard lk;
    flush(nullptr, everything_empty);
    g_mode = mode; g_rng.seed(seed ? seed : 1u);
    g_reordered = 0; g_run = 0; g_max_run = g_issue;
}
// operations run so far under the current schedule / how many of them ran after an operation issued later
extern "C" void sluamd_emul_sched_stats(unsigned long long *run, unsigned long long *reordered)
{
    Guard lk;
    *run = g_run; *reordered = g_reordered;
}

// SLUAMD_EMUL_LAZY_ZERO=1 (planning / footprint runs of problems whose factors exceed this host's memory, scripts/setup_breakdown.py): allocations of
// >= 256 MiB come from calloc -- untouched zero pages -- and the first whole-buffer hipMemset(0) of such a buffer is skipped, so that creating a handle
// never touches the value arena.  Nothing else changes; a factorisation would still commit every page it writes.
// (round 6: the untouched zero pages come from mmap(MAP_NORESERVE) -- calloc of an arena larger than RAM + swap is refused under the kernel's heuristic overcommit)
static std::map<const char *, size_t> g_fresh_zero, g_mapped;
static bool lazy_zero() { static const bool on = getenv("SLUAMD_EMUL_LAZY_ZERO") != nullptr; return on; }
hipError_t hipMalloc(void **p, size_t n)
{
    if (lazy_zero() && n >= ((size_t) 256 << 20)) {
        void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        *p = m == MAP_FAILED ? nullptr : m;
        if (*p) { Guard lk; g_fresh_zero[(const char *) *p] = n; g_mapped[(const char *) *p] = n; }
    } else *p = std::malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p)
{
    hipDeviceSynchronize();
    size_t mapped = 0;
    { Guard lk; g_fresh_zero.erase((const char *) p); auto it = g_mapped.find((const char *) p); if (it != g_mapped.end()) { mapped = it->second; g_mapped.erase(it); } }
    if (mapped) munmap(p, mapped); else std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
    Guard lk;
    *p = std::malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    g_pinned[(const char *) *p] = n ? n : 1;
    return hipSuccess;
}
hipError_t hipHostFree(void *p)
{
    hipDeviceSynchronize();
    Guard lk;
    g_pinned.erase((const char *) p);
    std::free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
    Guard lk;
    flush(nullptr, blocking_empty);
    if (n) std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n)
{
    Guard lk;
    flush(nullptr, blocking_empty);
    if (v == 0 && !g_fresh_zero.empty()) {
        auto it = g_fresh_zero.find((const char *) d);
        if (it != g_fresh_zero.end()) { const bool whole = it->second <= n; g_fresh_zero.erase(it); if (whole) return hipSuccess; }
    }
    if (n) std::memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st)
{
    Guard lk;
    if (g_mode == 0 || !n) { if (n) std::memmove(d, s, n); return hipSuccess; }
    if ((k == hipMemcpyHostToDevice && !pinned(s)) || k == hipMemcpyHostToHost) {
        if (k == hipMemcpyHostToHost) { hipStreamSynchronize(st); std::memmove(d, s, n); return hipSuccess; }
        auto buf = std::make_shared<std::vector<char>>((const char *) s, (const char *) s + n);   // pageable source: staged at the call
        emul_enqueue(st, [d, buf, n] { std::memcpy(d, buf->data(), n); });
        return hipSuccess;
    }
    if (k == hipMemcpyDeviceToHost && !pinned(d)) { hipStreamSynchronize(st); std::memmove(d, s, n); return hipSuccess; }   // pageable destination
    emul_enqueue(st, [d, s, n] { std::memmove(d, s, n); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st)
{
    if (n) emul_enqueue(st, [d, v, n] { std::memset(d, v, n); });
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s) { *s = new_stream(false); return hipSuccess; }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned, const unsigned *) { *s = new_stream(false); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int) { *s = new_stream((flags & hipStreamNonBlocking) != 0); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s)
{
    Guard lk;
    emul_stream_s *st = s ? s : null_stream();
    if (st == t_null) flush(nullptr, blocking_empty);
    else flush(st, [&] { return st->q.empty(); });
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) { return hipStreamSynchronize(s); }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    Guard lk;
    emul_stream_s *st = s ? s : null_stream();
    if (g_mode == 0 || !e->st || e->st == st || e->st->done >= e->seq) return hipSuccess;   // never recorded / same stream / already complete
    emul_op o; o.wait_on = e->st; o.wait_seq = e->seq;
    st->q.push_back(std::move(o)); ++st->issued;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new emul_event_s(); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    Guard lk;
    auto tp = e->t;
    if (g_mode == 0) { *tp = std::chrono::steady_clock::now(); e->st = nullptr; return hipSuccess; }
    emul_stream_s *st = s ? s : null_stream();
    emul_enqueue(st, [tp] { *tp = std::chrono::steady_clock::now(); });
    e->st = st; e->seq = st->issued;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    Guard lk;
    if (e->st) flush(e->st, [&] { return e->st->done >= e->seq; });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    hipEventSynchronize(a); hipEventSynchronize(b);
    *ms = std::chrono::duration<float, std::milli>(*b->t - *a->t).count();
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipDeviceSynchronize()
{
    Guard lk;
    flush(nullptr, all_empty);
    return hipSuccess;
}

// the product's pool of physical device chunks (sluamd_devpool.cpp) has nothing to pool on the host: plain allocations
namespace sluamd {
int devpool_alloc(void **p, size_t bytes, int) { return hipMalloc(p, bytes) == hipSuccess ? 0 : -4; }
void devpool_free(void *p) { if (p) hipFree(p); }
void devpool_trim(int) {}
size_t devpool_cached_bytes(int) { return 0; }
}  // namespace sluamd
