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
    int *d_red = nullptr;
    struct Op { void *d; int64_t bytes; int peer; bool is_recv; };
    std::vector<Op> ops;
    ~RcclComm() override
    {
        if (d_red) hipFree(d_red);
        if (nc) ncclCommDestroy(nc);
    }
    bool stream_ordered() const override { return true; }
    int begin() override { ops.clear(); return 0; }
    int send(const void *dbuf, int64_t bytes, int dst) override { ops.push_back({const_cast<void *>(dbuf), bytes, dst, false}); return 0; }
    int recv(void *dbuf, int64_t bytes, int src) override { ops.push_back({dbuf, bytes, src, true}); return 0; }
    int end(hipStream_t s) override
    {
        bool any = false;
        for (auto &o : ops) any |= o.bytes > 0;
        if (!any) { ops.clear(); return 0; }
        NCCLCHK(ncclGroupStart());
        for (auto &o : ops) {
            if (!o.bytes) continue;
            // payloads are whole doubles except the creation-time index exchange: count in bytes
            if (o.is_recv) NCCLCHK(ncclRecv(o.d, (size_t) o.bytes, ncclChar, o.peer, nc, s));
            else NCCLCHK(ncclSend(o.d, (size_t) o.bytes, ncclChar, o.peer, nc, s));
        }
        NCCLCHK(ncclGroupEnd());
        ops.clear();
        return 0;
    }
    int allreduce_min(int *v) override
    {
        HIPCHK(hipMemcpy(d_red, v, sizeof(int), hipMemcpyHostToDevice));
        NCCLCHK(ncclAllReduce(d_red, d_red, 1, ncclInt32, ncclMin, nc, nullptr));
        HIPCHK(hipStreamSynchronize(nullptr));
        HIPCHK(hipMemcpy(v, d_red, sizeof(int), hipMemcpyDeviceToHost));
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
    if (hipMalloc((void **) &c->d_red, sizeof(int)) != hipSuccess) return fail("hipMalloc failed");
    return c;
}

}  // namespace sluamd
