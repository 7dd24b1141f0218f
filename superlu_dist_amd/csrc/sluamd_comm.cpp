// sluamd_comm.cpp -- host-staged and in-process transports of sluamd_comm.h + the sluamd_comm_* C ABI.
#include <cstring>
#include "sluamd_comm.h"

namespace sluamd {

// ---- default host-buffer exchange: stage through device memory and reuse the device point-to-point path ----
int Comm::hbegin() { hops_.clear(); return begin(); }
int Comm::hsend(const void *buf, int64_t bytes, int dst)
{
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t) std::max<int64_t>(bytes, 8)));
    HIPCHK(hipMemcpy(d, buf, (size_t) bytes, hipMemcpyHostToDevice));
    hops_.push_back({const_cast<void *>(buf), d, bytes, false});
    return send(d, bytes, dst);
}
int Comm::hrecv(void *buf, int64_t bytes, int src)
{
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t) std::max<int64_t>(bytes, 8)));
    hops_.push_back({buf, d, bytes, true});
    return recv(d, bytes, src);
}
int Comm::hend()
{
    int rc = end(nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    for (auto &o : hops_) {
        if (o.is_recv) HIPCHK(hipMemcpy(o.h, o.d, (size_t) o.bytes, hipMemcpyDeviceToHost));
        hipFree(o.d);
    }
    hops_.clear();
    return 0;
}

// ---- CallbackComm ----------------------------------------------------------------------------------------------
CallbackComm::~CallbackComm() { if (stage) hipHostFree(stage); }
int CallbackComm::begin() { ops.clear(); return 0; }
int CallbackComm::send(const void *dbuf, int64_t bytes, int dst) { ops.push_back({const_cast<void *>(dbuf), bytes, dst, false, 0}); return 0; }
int CallbackComm::recv(void *dbuf, int64_t bytes, int src) { ops.push_back({dbuf, bytes, src, true, 0}); return 0; }
int CallbackComm::end(hipStream_t s)
{
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
    ops.clear();
    return 0;
}
int CallbackComm::allreduce_min(int *v)
{
    int32_t x = *v;
    if (cb.allreduce_min_i32(cb.ctx, &x)) { set_error("comm callback allreduce_min failed"); return SLUAMD_EINVAL; }
    *v = x;
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

int LocalComm::run(hipStream_t s, bool sync_stream)
{
    if (sync_stream) HIPCHK(hipStreamSynchronize(s));   // my send buffers are final, my receive buffers are no longer read
    const int P = w->size;
    std::vector<std::shared_ptr<LocalWorld::Msg>> mine;
    {
        std::lock_guard<std::mutex> lk(w->mu);
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
            w->cv.wait(lk, [&] { return !q.empty(); });
            m = q.front(); q.pop_front();
        }
        if (m->bytes != o.bytes) { set_error("LocalComm: message size mismatch (" + std::to_string(m->bytes) + " sent, " + std::to_string(o.bytes) + " expected)"); return SLUAMD_EINVAL; }
        if (o.bytes) {
            if (o.host) std::memcpy(o.p, m->ptr, (size_t) o.bytes);
            else {   // on the caller's stream (a device-to-device hipMemcpy on the null stream is neither host-synchronous nor
                     // ordered with non-blocking streams), completed before the sender is released
                HIPCHK(hipMemcpyAsync(o.p, m->ptr, (size_t) o.bytes, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipStreamSynchronize(s));
            }
        }
        {
            std::lock_guard<std::mutex> lk(w->mu);
            m->taken = true;
        }
        w->cv.notify_all();
    }
    {   // my sends may be reused once their receivers copied them
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { for (auto &m : mine) if (!m->taken) return false; return true; });
    }
    ops.clear();
    return 0;
}

int LocalComm::allreduce_min(int *v)
{
    std::unique_lock<std::mutex> lk(w->mu);
    const int gen = w->red_gen;
    if (w->red_count == 0) w->red_val = *v; else w->red_val = std::min(w->red_val, *v);
    if (++w->red_count == w->size) {
        w->red_out = w->red_val; w->red_count = 0; ++w->red_gen;
        w->cv.notify_all();
    } else {
        w->cv.wait(lk, [&] { return w->red_gen != gen; });
    }
    *v = w->red_out;
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

int sluamd_comm_rank(sluamd_comm_t c) { return c ? c->c->grid.rank() : -1; }
int sluamd_comm_size(sluamd_comm_t c) { return c ? c->c->grid.size() : 0; }

void sluamd_comm_destroy(sluamd_comm_t c)
{
    if (!c) return;
    delete c->c;
    delete c;
}

}  // extern "C"
