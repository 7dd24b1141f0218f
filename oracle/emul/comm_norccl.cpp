// oracle/emul/comm_norccl.cpp -- TEST INFRASTRUCTURE: the CPU test build has no RCCL transport
#include "sluamd_comm.h"
namespace sluamd {
Comm *make_rccl_comm(const void *, const Grid &, int) { set_error("the CPU test build has no RCCL transport"); return nullptr; }
int rccl_unique_id(void *) { set_error("the CPU test build has no RCCL transport"); return SLUAMD_ENODEVICE; }
}
