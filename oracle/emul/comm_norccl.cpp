// oracle/emul/comm_norccl.cpp -- TEST INFRASTRUCTURE: the CPU test build has no RCCL.  What stands behind
// sluamd_comm_create_rccl here is an in-process STREAM-ORDERED transport with RCCL's contract as the library uses it
// (sluamd_comm_rccl.cpp): end(s) only QUEUES the grouped sends / receives on stream s and returns; a send completes when the
// peer's matching receive has copied the data, a receive when the data has arrived; messages between one pair of ranks match in
// order, sizes must agree.  Ranks are threads of one process that were given the same unique id.  With it the drivers run the
// way they do over RCCL -- the host never waits for an exchange, only stream order and events protect the staging buffers, the
// scratch copies of received panels and the ancestor reduction -- under the adversarial scheduler of emul_rt.cpp
// (tests/test_stream_order.py).  The host-staged transports (LocalComm, CallbackComm) synchronise the stream at every exchange and
// would hide a missing dependency there.
#include <atomic>
#include <cstring>
#include <map>
#include <string>
#include <unistd.h>
#include "sluamd_comm.h"

namespace sluamd {

namespace {
struct SMsg { const void *ptr; int64_t bytes; std::atomic<bool> taken{false}; };
struct SWorld {
    int size = 0;
    std::vector<std::deque<std::shared_ptr<SMsg>>> box;   // [src * size + dst]; touched only by stream operations (under the runtime's lock)
    std::mutex mu; std::condition_variable cv;            // min-all-reduce (host side)
    int red_count = 0, red_gen = 0;
    std::vector<int> red_val, red_out;
};
std::mutex g_worlds_mu;
std::map<std::string, std::shared_ptr<SWorld>> g_worlds;
std::atomic<unsigned> g_next_id{1};

struct EmulStreamComm : Comm {
    std::shared_ptr<SWorld> w;
    int me = 0;
    struct Op { void *d; int64_t bytes; int peer; bool is_recv; };
    std::vector<Op> ops;
    bool stream_ordered() const override { return true; }
    int begin() override { ops.clear(); return 0; }
    int send(const void *dbuf, int64_t bytes, int dst) override { ops.push_back({const_cast<void *>(dbuf), bytes, dst, false}); return 0; }
    int recv(void *dbuf, int64_t bytes, int src) override { ops.push_back({dbuf, bytes, src, true}); return 0; }
    int end(hipStream_t s) override
    {
        // one group: every send is published first, then the receives wait for their messages, then the sends wait to be taken --
        // two ranks that send to each other in the same group cannot block each other (ncclGroupStart / ncclGroupEnd)
        std::vector<std::shared_ptr<SMsg>> mine;
        auto world = w;
        const int P = w->size, self = me;
        for (auto &o : ops) {
            if (o.is_recv || !o.bytes) continue;
            auto m = std::make_shared<SMsg>();
            m->ptr = o.d; m->bytes = o.bytes;
            mine.push_back(m);
            const size_t key = (size_t) self * P + o.peer;
            emul_enqueue(s, [world, key, m] { world->box[key].push_back(m); });
        }
        for (auto &o : ops) {
            if (!o.is_recv || !o.bytes) continue;
            const size_t key = (size_t) o.peer * P + self;
            void *dst = o.d; const int64_t bytes = o.bytes;
            emul_enqueue_when(s, [world, key] { return !world->box[key].empty(); }, [world, key, dst, bytes] {
                auto m = world->box[key].front();
                world->box[key].pop_front();
                if (m->bytes != bytes) { std::fprintf(stderr, "emulated stream-ordered transport: message size mismatch (%lld sent, %lld expected)\n", (long long) m->bytes, (long long) bytes); std::abort(); }
                std::memcpy(dst, m->ptr, (size_t) bytes);
                m->taken = true;
            });
        }
        for (auto &m : mine) emul_enqueue_when(s, [m] { return m->taken.load(); }, [] {});
        ops.clear();
        return 0;
    }
    int allreduce_min(int *v, int n, hipStream_t s) override
    {
        HIPCHK(hipStreamSynchronize(s));     // RcclComm: the collective runs on s and the host waits for s alone
        std::unique_lock<std::mutex> lk(w->mu);
        const int gen = w->red_gen;
        if (w->red_count == 0) w->red_val.assign(v, v + n);
        else for (int i = 0; i < n; ++i) w->red_val[i] = std::min(w->red_val[i], v[i]);
        if (++w->red_count == w->size) { w->red_out = w->red_val; w->red_count = 0; ++w->red_gen; w->cv.notify_all(); }
        else w->cv.wait(lk, [&] { return w->red_gen != gen; });
        for (int i = 0; i < n; ++i) v[i] = w->red_out[i];
        return 0;
    }
};
}  // namespace

int rccl_unique_id(void *id128)
{
    std::memset(id128, 0, SLUAMD_UNIQUE_ID_BYTES);
    const unsigned id = g_next_id++;
    std::memcpy(id128, "emul-stream-world", 17);
    std::memcpy(static_cast<char *>(id128) + 32, &id, sizeof(id));
    const long pid = (long) getpid();                  // the world lives in THIS process: ranks are its threads
    std::memcpy(static_cast<char *>(id128) + 48, &pid, sizeof(pid));
    return 0;
}

Comm *make_rccl_comm(const void *id128, const Grid &g, int)
{
    const std::string key(static_cast<const char *>(id128), SLUAMD_UNIQUE_ID_BYTES);
    if (key.compare(0, 17, "emul-stream-world") != 0) { set_error("the CPU test build has no RCCL transport (unknown unique id)"); return nullptr; }
    long pid = 0;
    std::memcpy(&pid, static_cast<const char *>(id128) + 48, sizeof(pid));
    if (pid != (long) getpid()) { set_error("the CPU test build has no RCCL transport: its stream-ordered stand-in connects threads of ONE process"); return nullptr; }
    auto *c = new EmulStreamComm();
    c->grid = g;
    c->me = g.rank();
    std::lock_guard<std::mutex> lk(g_worlds_mu);
    auto &w = g_worlds[key];
    if (!w) { w = std::make_shared<SWorld>(); w->size = g.size(); w->box.resize((size_t) g.size() * g.size()); }
    if (w->size != g.size()) { set_error("ranks of one unique id disagree on the grid size"); delete c; return nullptr; }
    c->w = w;
    return c;
}

}  // namespace sluamd
