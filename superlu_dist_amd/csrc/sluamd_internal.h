// Internal types shared by the host planner (sluamd_core.hip) and the symbolic producer
// (sluamd_symb.cpp).  Not part of the C ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "superlu_dist_amd.h"

namespace sluamd {

constexpr int BC_HEADER = 2;      // reference superlu_defs.h:169
constexpr int LB_DESCRIPTOR = 2;  // :170
constexpr int BR_HEADER = 3;      // :190
constexpr int UB_DESCRIPTOR = 2;  // :191

void set_error(const std::string &msg);

// Host copy of one rank's L/U *structure* (index arrays only), flattened over GLOBAL supernode ids.
// Round 1 supports 1 x 1 x Pz grids, so every supernode's panel/row is local (possibly as a zero replica).
struct HostStruct {
    int64_t n = 0;
    int nsupers = 0;
    std::vector<int> xsup;                 // [nsupers+1]
    std::vector<int64_t> lidx_off, uidx_off;  // [nsupers+1] into lidx/uidx
    std::vector<int64_t> lval_off, uval_off;  // [nsupers+1] into the value arena halves
    std::vector<int> lidx, uidx;
    int64_t nnzL = 0, nnzU = 0;
    std::vector<uint8_t> present;          // [nsupers] 0 = panel/row not stored on this rank (other Z layer's forest)
};

// Symbolic object behind sluamd_symb_t
struct Symb {
    HostStruct hs;
    std::vector<int> perm_c_final;  // perm_c_out
    std::vector<double> lval, uval; // optional host values (sluamd_ddistribute_host)
    std::vector<int *> lptr, uptr;  // pointer views for sluamd_symb_view
    std::vector<double *> lvptr, uvptr;
    std::vector<int> supno;         // [n] column -> supernode
    std::vector<int64_t> srow_off;  // [nsupers+1] off-diagonal structure (sorted global rows > last column)
    std::vector<int> srows;
    double flops = 0;
};

// positions of A's entries inside the value arena: out_pos[e] (into L arena if is_u[e]==0 else U arena)
// `hs` supplies the value offsets (it may be a local subset of sy.hs); `owned` (may be null) filters the entries
// by destination supernode: entries whose destination is not owned get pos = -1.
void compute_scatter_positions(const Symb &sy, const HostStruct &hs, int64_t n, const int *rowptr, const int *colind,
                               const int *perm_c_final, const uint8_t *owned, std::vector<int64_t> &pos,
                               std::vector<uint8_t> &is_u);
// tree (heap-numbered, root 0) of every supernode for a 1 x 1 x npdep grid
void partition_forests(const Symb &sy, int npdep, std::vector<int> &sn_tree);

}  // namespace sluamd
