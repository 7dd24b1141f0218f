// sluamd_comm.h -- point-to-point transport between the ranks of the 3D process grid (internal interface behind
// sluamd_comm_t).  The hot path only needs grouped sends / receives of contiguous device ranges (every panel
// "broadcast" of the reference -- dIBcast_LPanel / dIBcast_UPanel, dDiagFactIBCast, dcommunication_aux.c:32-200 -- has
// at most Pc - 1 or Pr - 1 receivers, and xGMI is a point-to-point fabric, so the root sends to each peer directly)
// plus one integer min-all-reduce (info, pdgstrf3d.c:388-392).
//
// Backends:
//   RcclComm      (sluamd_comm_rccl.cpp) ncclSend / ncclRecv / ncclAllReduce called directly, stream-ordered, one rank per GPU
//   CallbackComm  host-buffer isend / irecv / waitall supplied by the application (MPI in the reference-side binding,
//                 gloo in the CPU tests); device ranges are staged through pinned host memory
//   LocalComm     ranks = threads of one process (any number per device): device-to-device copies through a mailbox
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>
#include "sluamd_internal.h"

namespace sluamd {

struct Comm {
    Grid grid;
    virtual ~Comm() {}
    // grouped point-to-point on DEVICE buffers.  After end(s) the operations are ordered on stream s (RcclComm) or
    // complete (host-staged backends, which synchronise s first).  Messages between one pair of ranks match in order.
    virtual int begin() = 0;
    virtual int send(const void *dbuf, int64_t bytes, int dst) = 0;
    virtual int recv(void *dbuf, int64_t bytes, int src) = 0;
    virtual int end(hipStream_t s) = 0;
    // element-wise minimum of v[0..n) over the whole grid, host values in and out.  Ordered behind everything queued on
    // stream s and complete on return (RcclComm runs the collective ON s: no null-stream work, no device-wide wait)
    virtual int allreduce_min(int *v, int n, hipStream_t s) = 0;
    // error on one rank: release the peers that wait for it (in-process transports), so that a failure surfaces as an error
    // code everywhere instead of a hang
    virtual void poison() {}
    virtual bool stream_ordered() const { return false; }
    // grouped point-to-point on HOST buffers (creation-time structure exchange); default: staged through device memory
    virtual int hbegin();
    virtual int hsend(const void *buf, int64_t bytes, int dst);
    virtual int hrecv(void *buf, int64_t bytes, int src);
    virtual int hend();

protected:
    struct HOp { void *h; void *d; int64_t bytes; bool is_recv; };
    std::vector<HOp> hops_;
    void hfree_();   // release the device staging of a (possibly failed) host-buffer group
};

// ---- application-supplied transport ----
struct CallbackComm : Comm {
    sluamd_comm_callbacks_t cb{};
    struct Op { void *d; int64_t bytes; int peer; bool is_recv; size_t stage_off; };
    std::vector<Op> ops;
    char *stage = nullptr; size_t stage_cap = 0;
    ~CallbackComm() override;
    int begin() override;
    int send(const void *dbuf, int64_t bytes, int dst) override;
    int recv(void *dbuf, int64_t bytes, int src) override;
    int end(hipStream_t s) override;
    int allreduce_min(int *v, int n, hipStream_t s) override;
    int hbegin() override { return 0; }
    int hsend(const void *buf, int64_t bytes, int dst) override;
    int hrecv(void *buf, int64_t bytes, int src) override;
    int hend() override;
};

// ---- in-process world ----
struct LocalWorld {
    int size = 0;
    std::mutex mu;
    std::condition_variable cv;
    struct Msg { const void *ptr; int64_t bytes; bool host; bool taken = false; };
    std::vector<std::deque<std::shared_ptr<Msg>>> box;   // [src * size + dst]
    // min-all-reduce
    int red_count = 0, red_gen = 0;
    std::vector<int> red_val, red_out;
    bool poisoned = false;   // a rank failed inside an exchange: every waiter returns SLUAMD_EINVAL
};
struct LocalComm : Comm {
    std::shared_ptr<LocalWorld> w;
    int me = 0;
    struct Op { void *p; int64_t bytes; int peer; bool is_recv; bool host; };
    std::vector<Op> ops;
    int begin() override;
    int send(const void *dbuf, int64_t bytes, int dst) override;
    int recv(void *dbuf, int64_t bytes, int src) override;
    int end(hipStream_t s) override;
    int allreduce_min(int *v, int n, hipStream_t s) override;
    void poison() override;
    int hbegin() override { return 0; }
    int hsend(const void *buf, int64_t bytes, int dst) override;
    int hrecv(void *buf, int64_t bytes, int src) override;
    int hend() override;
private:
    int run(hipStream_t s, bool sync_stream);
};

Comm *make_rccl_comm(const void *id128, const Grid &g, int device);   // sluamd_comm_rccl.cpp
int rccl_unique_id(void *id128);

}  // namespace sluamd

struct sluamd_comm_s { sluamd::Comm *c; };
