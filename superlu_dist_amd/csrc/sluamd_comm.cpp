// sluamd_comm.cpp -- host-staged and in-process transports of sluamd_comm.h + the sluamd_comm_* C ABI.
#include <cstring>
#include "sluamd_comm.h"

namespace sluamd {

// ---- default host-buffer exchange: stage through device memory and reuse the device point-to-point path ----
void Comm::hfree_()
{
    for (auto &o : hops_) if (o.d) hipFree(o.d);
    hops_.clear();
}
int Comm::hbegin() { hfree_(); return begin(); }
int Comm::hsend(const void *buf, int64_t bytes, int dst)
{
    void *d = nullptr;
    if (hipMalloc(&d, (size_t) std::max<int64_t>(bytes, 8)) != hipSuccess) { hfree_(); set_error("hipMalloc of a host-exchange staging buffer failed"); return SLUAMD_ENOMEM; }
    hops_.push_back({const_cast<void *>(buf), d, bytes, false});
    if (hipMemcpy(d, buf, (size_t) bytes, hipMemcpyHostToDevice) != hipSuccess) { hfree_(); set_error("hipMemcpy to a host-exchange staging buffer failed"); return SLUAMD_EHIP; }
    const int rc = send(d, bytes, dst);
    if (rc) hfree_();
    return rc;
}
int Comm::hrecv(void *buf, int64_t bytes, int src)
{
    void *d = nullptr;
    if (hipMalloc(&d, (size_t) std::max<int64_t>(bytes, 8)) != hipSuccess) { hfree_(); set_error("hipMalloc of a host-exchange staging buffer failed"); return SLUAMD_ENOMEM; }
    hops_.push_back({buf, d, bytes, true});
    const int rc = recv(d, bytes, src);
    if (rc) hfree_();
    return rc;
}
int Comm::hend()
{
    // the staging copies above were synchronous; the exchange itself runs on a stream of its own so that nothing here depends on
    // (or waits for) the null stream
    hipStream_t hs = nullptr;
    if (hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, 0) != hipSuccess) { hfree_(); set_error("hipStreamCreate failed"); return SLUAMD_EHIP; }
    int rc = end(hs);
    if (!rc && hipStreamSynchronize(hs) != hipSuccess) { set_error("hipStreamSynchronize failed in a host-buffer exchange"); rc = SLUAMD_EHIP; }
    if (!rc)
        for (auto &o : hops_)
            if (o.is_recv && o.bytes && hipMemcpy(o.h, o.d, (size_t) o.bytes, hipMemcpyDeviceToHost) != hipSuccess) { set_error("hipMemcpy from a host-exchange staging buffer failed"); rc = SLUAMD_EHIP; break; }
    hipStreamDestroy(hs);
    hfree_();
    return rc;
}

// ---- CallbackComm ----------------------------------------------------------------------------------------------
CallbackComm::~CallbackComm() { if (stage) hipHostFree(stage); }
int CallbackComm::begin() { ops.clear(); return 0; }
int CallbackComm::send(const void *dbuf, int64_t bytes, int dst) { ops.push_back({const_cast<void *>(dbuf), bytes, dst, false, 0}); return 0; }
int CallbackComm::recv(void *dbuf, int64_t bytes, int src) { ops.push_back({dbuf, bytes, src, true, 0}); return 0; }
int CallbackComm::end(hipStream_t s)
{
    struct Clear { std::vector<Op> &o; ~Clear() { o.clear(); } } clear_on_exit{ops};   // also on the error returns: no stale operations in the next group
    size_t need = 0;
    for (auto &o : ops) { o.stage_off = need; need += ((size_t) o.bytes + 63) & ~(size_t) 63; }
    if (need > stage_cap) {
        if (stage) hipHostFree(stage);
        stage = nullptr; stage_cap = 0;
        HIPCHK(hipHostMalloc((void **) &stage, need, hipHostMallocDefault));
        stage_cap = need;
    }
    HIPCHK(hipStreamSynchronize(s));   // the data to send is produced by kernels queued on s
    for (auto &o : ops)
        if (!o.is_recv && o.bytes) HIPCHK(hipMemcpyAsync(stage + o.stage_off, o.d, (size_t) o.bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (auto &o : ops) {
        if (!o.bytes) continue;
        const int rc = o.is_recv ? cb.irecv(cb.ctx, stage + o.stage_off, o.bytes, o.peer) : cb.isend(cb.ctx, stage + o.stage_off, o.bytes, o.peer);
        if (rc) { set_error("comm callback isend/irecv failed"); return SLUAMD_EINVAL; }
    }
    if (cb.waitall(cb.ctx)) { set_error("comm callback waitall failed"); return SLUAMD_EINVAL; }
    for (auto &o : ops)
        if (o.is_recv && o.bytes) HIPCHK(hipMemcpyAsync(o.d, stage + o.stage_off, (size_t) o.bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}
int CallbackComm::allreduce_min(int *v, int n, hipStream_t s)
{
    HIPCHK(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) {
        int32_t x = v[i];
        if (cb.allreduce_min_i32(cb.ctx, &x)) { set_error("comm callback allreduce_min failed"); return SLUAMD_EINVAL; }
        v[i] = x;
    }
    return 0;
}
int CallbackComm::hsend(const void *buf, int64_t bytes, int dst)
{
    if (bytes && cb.isend(cb.ctx, buf, bytes, dst)) { set_error("comm callback isend failed"); return SLUAMD_EINVAL; }
    return 0;
}
int CallbackComm::hrecv(void *buf, int64_t bytes, int src)
{
    if (bytes && cb.irecv(cb.ctx, buf, bytes, src)) { set_error("comm callback irecv failed"); return SLUAMD_EINVAL; }
    return 0;
}
int CallbackComm::hend()
{
    if (cb.waitall(cb.ctx)) { set_error("comm callback waitall failed"); return SLUAMD_EINVAL; }
    return 0;
}

// ---- LocalComm -------------------------------------------------------------------------------------------------
int LocalComm::begin() { ops.clear(); return 0; }
int LocalComm::send(const void *dbuf, int64_t bytes, int dst) { ops.push_back({const_cast<void *>(dbuf), bytes, dst, false, false}); return 0; }
int LocalComm::recv(void *dbuf, int64_t bytes, int src) { ops.push_back({dbuf, bytes, src, true, false}); return 0; }
int LocalComm::hsend(const void *buf, int64_t bytes, int dst) { ops.push_back({const_cast<void *>(buf), bytes, dst, false, true}); return 0; }
int LocalComm::hrecv(void *buf, int64_t bytes, int src) { ops.push_back({buf, bytes, src, true, true}); return 0; }
int LocalComm::end(hipStream_t s) { return run(s, true); }
int LocalComm::hend() { return run(nullptr, false); }

void LocalComm::poison()
{
    { std::lock_guard<std::mutex> lk(w->mu); w->poisoned = true; }
    w->cv.notify_all();
}

int LocalComm::run(hipStream_t s, bool sync_stream)
{
    // every exit clears the group; a failing rank poisons the world so that its peers return an error instead of waiting for it
    struct Clear { std::vector<Op> &o; ~Clear() { o.clear(); } } clear_on_exit{ops};
    auto fail = [&](int rc, const std::string &msg) { if (!msg.empty()) set_error(msg); poison(); return rc; };
    if (sync_stream && hipStreamSynchronize(s) != hipSuccess) return fail(SLUAMD_EHIP, "hipStreamSynchronize failed before an exchange");   // my send buffers are final, my receive buffers are no longer read
    const int P = w->size;
    std::vector<std::shared_ptr<LocalWorld::Msg>> mine;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->poisoned) { set_error("LocalComm: a peer rank failed inside an exchange"); return SLUAMD_EINVAL; }
        for (auto &o : ops)
            if (!o.is_recv) {
                auto m = std::make_shared<LocalWorld::Msg>();
                m->ptr = o.p; m->bytes = o.bytes; m->host = o.host;
                w->box[(size_t) me * P + o.peer].push_back(m);
                mine.push_back(m);
            }
    }
    w->cv.notify_all();
    for (auto &o : ops) {
        if (!o.is_recv) continue;
        std::shared_ptr<LocalWorld::Msg> m;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            auto &q = w->box[(size_t) o.peer * P + me];
            w->cv.wait(lk, [&] { return !q.empty() || w->poisoned; });
            if (q.empty()) { set_error("LocalComm: a peer rank failed inside an exchange"); return SLUAMD_EINVAL; }
            m = q.front(); q.pop_front();
        }
        int rc = 0; std::string msg;
        if (m->bytes != o.bytes) { rc = SLUAMD_EINVAL; msg = "LocalComm: message size mismatch (" + std::to_string(m->bytes) + " sent, " + std::to_string(o.bytes) + " expected)"; }
        else if (o.bytes) {
            if (o.host) std::memcpy(o.p, m->ptr, (size_t) o.bytes);
            else {   // on the caller's stream (a device-to-device hipMemcpy on the null stream is neither host-synchronous nor
                     // ordered with non-blocking streams), completed before the sender is released
                if (hipMemcpyAsync(o.p, m->ptr, (size_t) o.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = SLUAMD_EHIP; msg = "LocalComm: device copy of a message failed"; }
            }
        }
        {
            std::lock_guard<std::mutex> lk(w->mu);
            m->taken = true;     // also after a failure: the sender must not wait for a message nobody will copy
        }
        w->cv.notify_all();
        if (rc) return fail(rc, msg);
    }
    {   // my sends may be reused once their receivers copied them
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { if (w->poisoned) return true; for (auto &m : mine) if (!m->taken) return false; return true; });
        for (auto &m : mine) if (!m->taken) { set_error("LocalComm: a peer rank failed inside an exchange"); return SLUAMD_EINVAL; }
    }
    return 0;
}

int LocalComm::allreduce_min(int *v, int n, hipStream_t s)
{
    if (hipStreamSynchronize(s) != hipSuccess) { set_error("hipStreamSynchronize failed"); poison(); return SLUAMD_EHIP; }
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->poisoned) { set_error("LocalComm: a peer rank failed"); return SLUAMD_EINVAL; }
    const int gen = w->red_gen;
    if (w->red_count == 0) w->red_val.assign(v, v + n);
    else for (int i = 0; i < n && i < (int) w->red_val.size(); ++i) w->red_val[i] = std::min(w->red_val[i], v[i]);
    if (++w->red_count == w->size) {
        w->red_out = w->red_val; w->red_count = 0; ++w->red_gen;
        w->cv.notify_all();
    } else {
        w->cv.wait(lk, [&] { return w->red_gen != gen || w->poisoned; });
        if (w->red_gen == gen) { set_error("LocalComm: a peer rank failed"); return SLUAMD_EINVAL; }
    }
    for (int i = 0; i < n && i < (int) w->red_out.size(); ++i) v[i] = w->red_out[i];
    return 0;
}

}  // namespace sluamd

using namespace sluamd;

extern "C" {

static int grid_ok(int nprow, int npcol, int npdep, int r, int c, int z)
{
    if (nprow < 1 || npcol < 1 || npdep < 1 || (npdep & (npdep - 1)) || r < 0 || r >= nprow || c < 0 || c >= npcol || z < 0 || z >= npdep) {
        set_error("bad process grid (npdep must be a power of two, coordinates inside the grid)");
        return 0;
    }
    return 1;
}

int sluamd_comm_create_callbacks(sluamd_comm_t *out, const sluamd_comm_callbacks_t *cb, int nprow, int npcol, int npdep, int myrow, int mycol,
                                 int myz)
{
    if (!out || !cb || !cb->isend || !cb->irecv || !cb->waitall || !cb->allreduce_min_i32) { set_error("null comm callbacks"); return SLUAMD_EINVAL; }
    if (!grid_ok(nprow, npcol, npdep, myrow, mycol, myz)) return SLUAMD_EINVAL;
    auto *c = new CallbackComm();
    c->cb = *cb;
    c->grid = Grid{nprow, npcol, npdep, myrow, mycol, myz};
    *out = new sluamd_comm_s{c};
    return 0;
}

int sluamd_comm_create_local(sluamd_comm_t *comms, int nprow, int npcol, int npdep)
{
    if (!comms || !grid_ok(nprow, npcol, npdep, 0, 0, 0)) { if (!comms) set_error("null comm array"); return SLUAMD_EINVAL; }
    auto w = std::make_shared<LocalWorld>();
    const int P = nprow * npcol * npdep;
    w->size = P;
    w->box.resize((size_t) P * P);
    for (int z = 0; z < npdep; ++z)
        for (int r = 0; r < nprow; ++r)
            for (int c = 0; c < npcol; ++c) {
                auto *lc = new LocalComm();
                lc->w = w;
                lc->grid = Grid{nprow, npcol, npdep, r, c, z};
                lc->me = lc->grid.rank();
                comms[lc->me] = new sluamd_comm_s{lc};
            }
    return 0;
}

int sluamd_comm_rccl_unique_id(void *id128)
{
    if (!id128) { set_error("null id buffer"); return SLUAMD_EINVAL; }
    return rccl_unique_id(id128);
}

int sluamd_comm_create_rccl(sluamd_comm_t *out, const void *id128, int nprow, int npcol, int npdep, int myrow, int mycol, int myz, int device)
{
    if (!out || !id128) { set_error("null argument"); return SLUAMD_EINVAL; }
    if (!grid_ok(nprow, npcol, npdep, myrow, mycol, myz)) return SLUAMD_EINVAL;
    Comm *c = make_rccl_comm(id128, Grid{nprow, npcol, npdep, myrow, mycol, myz}, device);
    if (!c) return SLUAMD_EHIP;
    *out = new sluamd_comm_s{c};
    return 0;
}

// Transport self-test: every rank sends `bytes` bytes of a rank-specific pattern to the NEXT world rank (itself on a one-rank
// world) and receives from the previous one, as ONE group queued on a stream of its own between two device fills -- the
// stream-ordered contract the drivers rely on (sluamd_factor.cpp: exchange(), xseg_exchange(), reduce_ancestors()) -- then the
// host-buffer group and the min-all-reduce.  Collective.
int sluamd_comm_selftest(sluamd_comm_t comm, int64_t bytes)
{
    if (!comm || bytes < 8) { set_error("sluamd_comm_selftest: null communicator or fewer than 8 bytes"); return SLUAMD_EINVAL; }
    Comm *c = comm->c;
    const int P = c->grid.size(), me = c->grid.rank(), nxt = (me + 1) % P, prv = (me + P - 1) % P;
    const size_t n = (size_t) bytes;
    unsigned char *d_src = nullptr, *d_dst = nullptr;
    hipStream_t s = nullptr;
    std::vector<unsigned char> h(n), hsrc(n, (unsigned char) (0x50 + me)), hdst(n, 0);
    int rc = 0;
    auto done = [&](int r, const char *msg) { if (msg) set_error(msg); if (d_src) hipFree(d_src); if (d_dst) hipFree(d_dst); if (s) hipStreamDestroy(s); return r; };
    if (hipMalloc((void **) &d_src, n) != hipSuccess || hipMalloc((void **) &d_dst, n) != hipSuccess) return done(SLUAMD_ENOMEM, "sluamd_comm_selftest: hipMalloc failed");
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0) != hipSuccess) return done(SLUAMD_EHIP, "sluamd_comm_selftest: stream creation failed");
    // (1) device group, stream-ordered: fill(src) -> exchange -> nothing else; the receive buffer is cleared on the same stream first
    if (hipMemsetAsync(d_src, 0xA0 + me, n, s) != hipSuccess || hipMemsetAsync(d_dst, 0, n, s) != hipSuccess) return done(SLUAMD_EHIP, "sluamd_comm_selftest: memset failed");
    if ((rc = c->begin()) || (rc = c->send(d_src, bytes, nxt)) || (rc = c->recv(d_dst, bytes, prv)) || (rc = c->end(s))) return done(rc, nullptr);
    if (hipMemcpyAsync(h.data(), d_dst, n, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return done(SLUAMD_EHIP, "sluamd_comm_selftest: copy back failed");
    for (size_t i = 0; i < n; ++i) if (h[i] != (unsigned char) (0xA0 + prv)) return done(SLUAMD_EINVAL, "sluamd_comm_selftest: device exchange delivered wrong data");
    // (2) an empty group and a zero-byte message are legal
    if ((rc = c->begin()) || (rc = c->end(s))) return done(rc, nullptr);
    if ((rc = c->begin()) || (rc = c->send(d_src, 0, nxt)) || (rc = c->recv(d_dst, 0, prv)) || (rc = c->end(s))) return done(rc, nullptr);
    // (3) host-buffer group (creation-time structure exchange)
    if ((rc = c->hbegin()) || (rc = c->hsend(hsrc.data(), bytes, nxt)) || (rc = c->hrecv(hdst.data(), bytes, prv)) || (rc = c->hend())) return done(rc, nullptr);
    for (size_t i = 0; i < n; ++i) if (hdst[i] != (unsigned char) (0x50 + prv)) return done(SLUAMD_EINVAL, "sluamd_comm_selftest: host exchange delivered wrong data");
    // (4) min-all-reduce of two values, ordered on the stream
    int v[2] = {100 + me, -me};
    if ((rc = c->allreduce_min(v, 2, s))) return done(rc, nullptr);
    if (v[0] != 100 || v[1] != -(P - 1)) return done(SLUAMD_EINVAL, "sluamd_comm_selftest: allreduce_min gave a wrong result");
    return done(0, nullptr);
}

int sluamd_comm_rank(sluamd_comm_t c) { return c ? c->c->grid.rank() : -1; }
int sluamd_comm_size(sluamd_comm_t c) { return c ? c->c->grid.size() : 0; }

void sluamd_comm_destroy(sluamd_comm_t c)
{
    if (!c) return;
    delete c->c;
    delete c;
}

}  // extern "C"
