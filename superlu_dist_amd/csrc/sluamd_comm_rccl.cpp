// sluamd_comm_rccl.cpp -- RCCL transport: one rank per GPU, ncclSend / ncclRecv grouped per exchange phase and queued on
// the caller's HIP stream (so an exchange overlaps with the Schur tiles running on the other stream), ncclAllReduce(min)
// for info.  Replaces the MPI panel exchange of the reference (dIBcast_LPanel / dIBcast_UPanel / dDiagFactIBCast,
// dcommunication_aux.c:32-200; dzSendLPanel / dzRecvLPanel, pd3dcomm.c:189-331) over xGMI.
#include <rccl/rccl.h>
#include <cstring>
#include "sluamd_comm.h"

namespace sluamd {

#define NCCLCHK(expr)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            set_error(std::string(#expr) + " failed: " + ncclGetErrorString(r_));              \
            return SLUAMD_EHIP;                                                                \
        }                                                                                      \
    } while (0)

struct RcclComm : Comm {
    ncclComm_t nc = nullptr;
    int device = 0;
    static constexpr int RED_MAX = 16;
    int *d_red = nullptr, *h_red = nullptr;   // device / pinned host image of the values of allreduce_min
    struct Op { void *d; int64_t bytes; int peer; bool is_recv; };
    std::vector<Op> ops;
    ~RcclComm() override
    {
        if (d_red) hipFree(d_red);
        if (h_red) hipHostFree(h_red);
        if (nc) ncclCommDestroy(nc);
    }
    bool stream_ordered() const override { return true; }
    int begin() override { ops.clear(); return 0; }
    int send(const void *dbuf, int64_t bytes, int dst) override { ops.push_back({const_cast<void *>(dbuf), bytes, dst, false}); return 0; }
    int recv(void *dbuf, int64_t bytes, int src) override { ops.push_back({dbuf, bytes, src, true}); return 0; }
    int end(hipStream_t s) override
    {
        struct Clear { std::vector<Op> &o; ~Clear() { o.clear(); } } clear_on_exit{ops};   // on every exit: no stale operations in the next group
        bool any = false;
        for (auto &o : ops) any |= o.bytes > 0;
        if (!any) return 0;
        NCCLCHK(ncclGroupStart());
        // an error inside the group must still close it: a return between ncclGroupStart and ncclGroupEnd would leave this thread's
        // group depth unbalanced and every later RCCL call silently deferred
        ncclResult_t first = ncclSuccess;
        const char *what = "";
        for (auto &o : ops) {
            if (!o.bytes || first != ncclSuccess) continue;
            // payloads are whole doubles except the creation-time index exchange: count in bytes
            first = o.is_recv ? ncclRecv(o.d, (size_t) o.bytes, ncclChar, o.peer, nc, s) : ncclSend(o.d, (size_t) o.bytes, ncclChar, o.peer, nc, s);
            what = o.is_recv ? "ncclRecv" : "ncclSend";
        }
        const ncclResult_t ge = ncclGroupEnd();
        if (first != ncclSuccess) { set_error(std::string(what) + " failed: " + ncclGetErrorString(first)); return SLUAMD_EHIP; }
        if (ge != ncclSuccess) { set_error(std::string("ncclGroupEnd failed: ") + ncclGetErrorString(ge)); return SLUAMD_EHIP; }
        return 0;
    }
    int allreduce_min(int *v, int n, hipStream_t s) override
    {
        // on the caller's stream: ordered behind the factorisation's last kernels without a device-wide wait and without touching
        // the null stream (which would serialise against the library's blocking streams)
        if (n > RED_MAX) { set_error("allreduce_min: too many values"); return SLUAMD_EINVAL; }
        std::memcpy(h_red, v, sizeof(int) * (size_t) n);
        HIPCHK(hipMemcpyAsync(d_red, h_red, sizeof(int) * (size_t) n, hipMemcpyHostToDevice, s));
        NCCLCHK(ncclAllReduce(d_red, d_red, (size_t) n, ncclInt32, ncclMin, nc, s));
        HIPCHK(hipMemcpyAsync(h_red, d_red, sizeof(int) * (size_t) n, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::memcpy(v, h_red, sizeof(int) * (size_t) n);
        return 0;
    }
};

int rccl_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) <= SLUAMD_UNIQUE_ID_BYTES, "unique id buffer too small");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    std::memset(id128, 0, SLUAMD_UNIQUE_ID_BYTES);
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

Comm *make_rccl_comm(const void *id128, const Grid &g, int device)
{
    auto *c = new RcclComm();
    c->grid = g;
    auto fail = [&](const std::string &m) -> Comm * { set_error(m); delete c; return nullptr; };
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    if (hipGetDevice(&c->device) != hipSuccess) return fail("no HIP device");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->nc, g.size(), id, g.rank());
    if (r != ncclSuccess) return fail(std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r));
    if (hipMalloc((void **) &c->d_red, sizeof(int) * RcclComm::RED_MAX) != hipSuccess) return fail("hipMalloc failed");
    if (hipHostMalloc((void **) &c->h_red, sizeof(int) * RcclComm::RED_MAX, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc failed");
    return c;
}

}  // namespace sluamd
