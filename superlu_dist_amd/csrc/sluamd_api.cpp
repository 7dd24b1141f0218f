// sluamd_api.cpp -- the C ABI of include/superlu_dist_amd.h (handle life cycle, value upload / download, factor / solve /
// refinement entry points).  No CPU fallback: every entry point fails when no HIP device is present.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "sluamd_comm.h"
#include "sluamd_plan.h"

using namespace sluamd;

namespace sluamd {

// ---- bounded pinned staging: own slot values <-> caller's panel / skyline arrays -------------------------------------
static int ensure_pinned(Handle *H)
{
    if (H->h_pinned) return 0;
    size_t want = (size_t) 64 << 20;
    if (const char *v = getenv("SLUAMD_PINNED_BYTES")) want = std::max<size_t>(256, (size_t) atoll(v) & ~(size_t) 15);   // test knob: slots split across many flushes
    HIPCHK(hipHostMalloc(&H->h_pinned, want, hipHostMallocDefault));
    H->pinned_bytes = want;
    return 0;
}

// dir = 0: host -> device (upload), 1: device -> host.  Own slots are contiguous in the arena in own_*_order, so the
// copies go through the pinned buffer in runs of whole slots (slots larger than the buffer are split).
static int copy_values(Handle *H, const sluamd_dLUview_t *lu, int dir)
{
    const HostStruct &hs = H->hs;
    const Grid &g = H->grid;
    const size_t esz = H->z ? 16 : 8;
    int rc = ensure_pinned(H);
    if (rc) return rc;
    char *pin = reinterpret_cast<char *>(H->h_pinned);
    const size_t cap = H->pinned_bytes;
    char *dv = reinterpret_cast<char *>(H->d_val);
    // a piece = bytes of the staging buffer <-> host memory: a contiguous run (perm == null), or elements [e0, e0 + bytes / esz) of an L panel in the handle's
    // row order, whose rows the caller keeps in another order inside its blocks (HostStruct::lrow_perm): element (r, c) of the panel <-> host[perm[r] + c * lda]
    struct Piece { char *host; size_t bytes; const int *perm; int64_t e0; int lda; };
    std::vector<Piece> pieces;
    size_t fill = 0; int64_t dev_byte = -1;   // byte offset in the arena of the first staged byte
    auto move_piece = [&](const Piece &p, char *stage_ptr) {
        if (!p.perm) { if (dir == 0) std::memcpy(stage_ptr, p.host, p.bytes); else std::memcpy(p.host, stage_ptr, p.bytes); return; }
        // column by column, a tight gather / scatter over the rows of the piece (8-byte words: one per double, two per doublecomplex)
        const int64_t ne = (int64_t) (p.bytes / esz);
        const int w = (int) (esz / 8);
        int64_t c = p.e0 / p.lda; int r = (int) (p.e0 - c * p.lda);
        uint64_t *st8 = reinterpret_cast<uint64_t *>(stage_ptr);
        for (int64_t i = 0; i < ne;) {
            const int nr = (int) std::min<int64_t>(p.lda - r, ne - i);
            uint64_t *col = reinterpret_cast<uint64_t *>(p.host) + (size_t) c * p.lda * w;
            const int *pm = p.perm + r;
            uint64_t *sp = st8 + (size_t) i * w;
            if (w == 1) { if (dir == 0) for (int q = 0; q < nr; ++q) sp[q] = col[pm[q]]; else for (int q = 0; q < nr; ++q) col[pm[q]] = sp[q]; }
            else if (dir == 0) for (int q = 0; q < nr; ++q) { sp[2 * q] = col[2 * (size_t) pm[q]]; sp[2 * q + 1] = col[2 * (size_t) pm[q] + 1]; }
            else for (int q = 0; q < nr; ++q) { col[2 * (size_t) pm[q]] = sp[2 * q]; col[2 * (size_t) pm[q] + 1] = sp[2 * q + 1]; }
            i += nr; r = 0; ++c;
        }
    };
    auto flush = [&]() -> int {
        if (!fill) return 0;
        if (dir == 0) {
            size_t o = 0;
            for (auto &p : pieces) { move_piece(p, pin + o); o += p.bytes; }
            HIPCHK(hipMemcpy(dv + dev_byte, pin, fill, hipMemcpyHostToDevice));
        } else {
            HIPCHK(hipMemcpy(pin, dv + dev_byte, fill, hipMemcpyDeviceToHost));
            size_t o = 0;
            for (auto &p : pieces) { move_piece(p, pin + o); o += p.bytes; }
        }
        pieces.clear(); fill = 0; dev_byte = -1;
        return 0;
    };
    // one contiguous host range -> the next bytes of the arena, through the pinned buffer
    auto stage = [&](char *hp, size_t total, int64_t dev_elem_off, const int *perm = nullptr, int lda = 0) -> int {
        size_t done = 0;
        while (done < total) {
            const int64_t byte_off = dev_elem_off * (int64_t) esz + (int64_t) done;
            if (fill && (dev_byte + (int64_t) fill != byte_off || fill == cap)) { int rc2 = flush(); if (rc2) return rc2; }
            if (!fill) dev_byte = byte_off;
            const size_t n = std::min(total - done, cap - fill);      // cap and done are multiples of the element size
            if (perm) pieces.push_back({hp, n, perm, (int64_t) (done / esz), lda});
            else pieces.push_back({hp + done, n, nullptr, 0, 0});
            fill += n; done += n;
        }
        return 0;
    };
    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<int> &order = pass == 0 ? H->own_l_order : H->own_u_order;
        for (int k : order) {
            const int64_t len = pass == 0 ? hs.lval_len[k] : hs.uval_len[k];
            if (!len) continue;
            const int64_t off = pass == 0 ? hs.lval_off[k] : hs.uval_off[k];
            if (H->split.active) {   // refined wide supernodes: the slot's values are gathered from column pieces of the caller's panels
                int64_t o = off;
                for (const auto &pc : (pass == 0 ? H->split.lsrc[k] : H->split.usrc[k])) {
                    char *base = reinterpret_cast<char *>(pc.arr == 0 ? (void *) lu->Lnzval_bc_ptr[pc.ok / g.Pc] : (void *) lu->Unzval_br_ptr[pc.ok / g.Pr]);
                    if (!base) { set_error("value array missing for a stored panel"); return SLUAMD_ESTRUCT; }
                    if ((rc = stage(base + (size_t) pc.hoff * esz, (size_t) pc.len * esz, o))) return rc;
                    o += pc.len;
                }
                if (o - off != len) { set_error("internal: refined slot size mismatch"); return SLUAMD_ESTRUCT; }
                continue;
            }
            char *hp = reinterpret_cast<char *>(pass == 0 ? (void *) lu->Lnzval_bc_ptr[k / g.Pc] : (void *) lu->Unzval_br_ptr[k / g.Pr]);
            if (!hp) { set_error("value array missing for a stored panel"); return SLUAMD_ESTRUCT; }
            const std::vector<int> *pm = (pass == 0 && (size_t) k < hs.lrow_perm.size() && !hs.lrow_perm[k].empty()) ? &hs.lrow_perm[k] : nullptr;
            if (pm) {
                const int lda = (int) pm->size();
                if (!lda || len % lda) { set_error("internal: row permutation of an L panel does not match its slot"); return SLUAMD_ESTRUCT; }
                if ((rc = stage(hp, (size_t) len * esz, off, pm->data(), lda))) return rc;
            } else if ((rc = stage(hp, (size_t) len * esz, off))) return rc;
        }
        if ((rc = flush())) return rc;
    }
    return 0;
}

static int timed_copy(Handle *H, const sluamd_dLUview_t *lu, int dir)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    int rc = copy_values(H, lu, dir);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    (dir == 0 ? H->st.t_h2d_ms : H->st.t_d2h_ms) = ms;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}

static int create_from_view(sluamd_handle_t *out, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests,
                            const sluamd_options_t *opt, bool z, Comm *comm)
{
    if (!out) { set_error("null handle pointer"); return SLUAMD_EINVAL; }
    *out = nullptr;
    sluamd_options_t o;
    if (opt) o = *opt; else sluamd_default_options(&o);
    int rc = check_device(o.device);
    if (rc) return rc;
    auto *hh = new sluamd_lu_handle_s();
    Handle *H = &hh->H;
    H->opt = o;
    H->z = z;
    H->comm = comm;
    H->env.info_last = o.info_rule == SLUAMD_INFO_REFERENCE;
    read_env(H->env);
    auto fail = [&](int code) { sluamd_dDestroyLUHandle(hh); return code; };
    if (hipGetDevice(&H->device) != hipSuccess) { set_error("hipGetDevice failed"); return fail(SLUAMD_EHIP); }
    SlotInput in;
    HostTables t;
    if ((rc = slots_from_view(*H, lu, forests, comm, in))) return fail(rc);
    if ((rc = plan_and_upload(H, in, t))) return fail(rc);
    if ((rc = timed_copy(H, lu, 0))) return fail(rc);
    *out = hh;
    return 0;
}

// positions of A's entries inside the value arena of this rank's store (device-side pddistribute3d): -1 = not stored
// here (other process row / column, other layer's forest, or a replicated ancestor whose A entries live on the first
// layer of its group -- dinit3DLUstructForest's rule, pd3dcomm.c:334-800)
static int scatter_positions(const Handle &H, const Symb &sy, const HostTables &t, const int *rowptr, const int *colind, const int *perm,
                             std::vector<int64_t> &pos)
{
    const HostStruct &hs = H.hs;
    const Grid &g = H.grid;
    const int64_t n = hs.n;
    const int ns = hs.nsupers;
    std::vector<uint8_t> mine(ns, 0);   // A's entries destined to supernode k's panel / row are kept on this layer
    for (size_t zl = 0; zl < H.forest_nodes.size(); ++zl)
        if (g.z % (1 << zl) == 0) for (int k : H.forest_nodes[zl]) mine[k] = 1;
    pos.assign((size_t) rowptr[n], -1);
    // supernode of a column in the handle's (possibly refined: H.split) partition
    auto snode = [&](int col) { return H.split.active ? (int) (std::upper_bound(hs.xsup.begin(), hs.xsup.end(), col) - hs.xsup.begin()) - 1 : sy.supno[col]; };
    // every entry of A finds its arena position on its own: rows in dynamic chunks over the planner's threads
    std::atomic<int> err{0};     // 1: outside L, 2: outside U, 3: above the skyline
    parallel_chunks(n, 8192, [&](int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; ++i)
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int pi = perm[i], pj = perm[colind[e]];
            const int s = snode(pj);
            if (pi >= hs.xsup[s]) {       // L(:, s), row pi
                if (!mine[s] || g.kcol(s) != g.c) continue;
                const int ib = snode(pi);
                if (g.krow(ib) != g.r) continue;
                const int o = t.sn_lb_off[s], nb = t.sn_nlb[s];
                const int *dir = t.lbs_gid.data() + o;
                const int *f = std::lower_bound(dir, dir + nb, ib);
                if (f == dir + nb || *f != ib) { err = 1; return; }
                const int b = o + t.lbs_idx[o + (int) (f - dir)];
                const int *rows = hs.lidx.data() + hs.lidx_off[s] + t.lb_lptr[b];
                const int *fr = std::lower_bound(rows, rows + t.lb_nbrow[b], pi);
                if (fr == rows + t.lb_nbrow[b] || *fr != pi) { fr = std::find(rows, rows + t.lb_nbrow[b], pi); if (fr == rows + t.lb_nbrow[b]) { err = 1; return; } }
                pos[e] = hs.lval_off[s] + t.lb_rowoff[b] + (fr - rows) + (int64_t) (pj - hs.xsup[s]) * t.sn_nsupr[s];
            } else {                       // U(r, s), r = supernode of row pi
                const int r = snode(pi);
                if (!mine[r] || g.krow(r) != g.r || g.kcol(s) != g.c) continue;
                const int o = t.sn_ub_off[r], nb = t.sn_nub[r];
                const int *dir = t.ub_gid.data() + o;
                const int *f = std::lower_bound(dir, dir + nb, s);
                if (f == dir + nb || *f != s) { err = 2; return; }
                const int64_t ip = hs.uidx_off[r] + t.ub_iukp[o + (int) (f - dir)] + (pj - hs.xsup[s]);
                const int fst = hs.uidx[ip];
                if (pi < fst) { err = 3; return; }
                pos[e] = hs.uval_off[r] + t.ucolptr[ip] + (pi - fst);
            }
        }
    });
    if (err) { set_error(err == 1 ? "A entry outside the symbolic structure of L" : err == 2 ? "A entry outside the symbolic structure of U" : "A entry above the skyline of U"); return SLUAMD_ESTRUCT; }
    return 0;
}

static int create_from_symb(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr, const sluamd_int_t *colind, const double *nzval,
                            const sluamd_int_t *perm_c_final, const sluamd_options_t *opt, const Grid &g, const int32_t *sn_tree, Comm *comm,
                            bool z)
{
    if (!out || !s || !rowptr || !colind || !nzval || !perm_c_final) { set_error("null argument"); return SLUAMD_EINVAL; }
    *out = nullptr;
    if (g.size() > 1 && !comm) { set_error("a process grid with more than one rank needs a communicator"); return SLUAMD_EINVAL; }
    sluamd_options_t o;
    if (opt) o = *opt; else sluamd_default_options(&o);
    int rc = check_device(o.device);
    if (rc) return rc;
    Symb *sy = reinterpret_cast<Symb *>(s);
    auto *hh = new sluamd_lu_handle_s();
    Handle *H = &hh->H;
    H->opt = o;
    H->z = z;
    H->comm = comm;
    H->env.info_last = o.info_rule == SLUAMD_INFO_REFERENCE;
    read_env(H->env);
    auto fail = [&](int code) { sluamd_dDestroyLUHandle(hh); return code; };
    if (hipGetDevice(&H->device) != hipSuccess) { set_error("hipGetDevice failed"); return fail(SLUAMD_EHIP); }
    SlotInput in;
    HostTables t;
    H->setup.start();
    if ((rc = slots_from_symb(*H, *sy, g, sn_tree, in))) return fail(rc);
    H->setup.lap("slots_from_symbolic");
    if ((rc = plan_and_upload(H, in, t))) return fail(rc);
    {   // device-side distribution of A's values (the arena is zero-filled)
        std::vector<int64_t> pos;
        if ((rc = scatter_positions(*H, *sy, t, rowptr, colind, perm_c_final, pos))) return fail(rc);
        H->setup.lap("scatter_positions");
        const int w = z ? 2 : 1;   // doubles per value
        std::vector<int64_t> pos2; std::vector<double> val2;
        pos2.reserve(pos.size()); val2.reserve(pos.size() * w);
        for (size_t e = 0; e < pos.size(); ++e)
            if (pos[e] >= 0) { pos2.push_back(pos[e]); for (int q = 0; q < w; ++q) val2.push_back(nzval[e * w + q]); }
        const int64_t nnz = (int64_t) pos2.size();
        const size_t esz = z ? 16 : 8;
        if (hipMalloc((void **) &H->d_apos, sizeof(int64_t) * std::max<int64_t>(nnz, 1)) != hipSuccess ||
            hipMalloc((void **) &H->d_aval, esz * std::max<int64_t>(nnz, 1)) != hipSuccess) { set_error("hipMalloc failed"); return fail(SLUAMD_ENOMEM); }
        H->a_nnz = nnz;
        if (nnz) {
            if (hipMemcpy(H->d_apos, pos2.data(), sizeof(int64_t) * nnz, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(H->d_aval, val2.data(), esz * nnz, hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return fail(SLUAMD_EHIP); }
            if (z) eng::zscatter_values(H->stream, H->d_val, H->d_apos, H->d_aval, nnz);
            else eng::scatter_values(H->stream, H->d_val, H->d_apos, H->d_aval, nnz);
        }
        if (hipStreamSynchronize(H->stream) != hipSuccess) { set_error("distribution kernel failed"); return fail(SLUAMD_EHIP); }
        H->setup.lap("values_upload_scatter");
    }
    *out = hh;
    return 0;
}

}  // namespace sluamd

extern "C" {

const char *sluamd_last_error(void) { return get_error().c_str(); }

int sluamd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t sluamd_device_pool_trim(int dev)
{
    const int64_t held = (int64_t) devpool_cached_bytes(dev);
    devpool_trim(dev);
    return held;
}

int sluamd_device_pci_bus_id(int dev, char *buf, int len)
{
    if (!buf || len < 16) { set_error("sluamd_device_pci_bus_id: buffer of at least 16 bytes"); return SLUAMD_EINVAL; }
    HIPCHK(hipDeviceGetPCIBusId(buf, len, dev));
    return 0;
}

void sluamd_default_options(sluamd_options_t *opt)
{
    std::memset(opt, 0, sizeof(*opt));
    opt->device = -1;
}

int sluamd_dCreateLUHandle(sluamd_handle_t *out, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests, const sluamd_options_t *opt)
{
    return create_from_view(out, lu, forests, opt, false, nullptr);
}

int sluamd_dCreateLUHandleGrid(sluamd_handle_t *out, const sluamd_dLUview_t *lu, const sluamd_forest_view_t *forests, const sluamd_options_t *opt,
                               sluamd_comm_t comm)
{
    if (!comm) { set_error("null communicator"); return SLUAMD_EINVAL; }
    return create_from_view(out, lu, forests, opt, false, comm->c);
}

// complex16 twin: zCreateLUgpuHandle (SRC/include/superlu_upacked.h, z section)
int sluamd_zCreateLUHandle(sluamd_handle_t *out, const sluamd_zLUview_t *lu, const sluamd_forest_view_t *forests, const sluamd_options_t *opt)
{
    return create_from_view(out, reinterpret_cast<const sluamd_dLUview_t *>(lu), forests, opt, true, nullptr);
}

// complex16 on Z layers: 1 x 1 x npdep grids (the Z ancestor reduction and the distributed solve run on pairs of doubles)
int sluamd_zCreateLUHandleGrid(sluamd_handle_t *out, const sluamd_zLUview_t *lu, const sluamd_forest_view_t *forests, const sluamd_options_t *opt,
                               sluamd_comm_t comm)
{
    if (!comm) { set_error("null communicator"); return SLUAMD_EINVAL; }

    return create_from_view(out, reinterpret_cast<const sluamd_dLUview_t *>(lu), forests, opt, true, comm->c);
}

int sluamd_dSetValues(sluamd_handle_t h, const sluamd_dLUview_t *lu)
{
    if (!h || !lu) { set_error("null argument"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(h->H.device));
    h->H.dinv_ready = false; h->H.inv_ready = false; h->H.factored = false;
    return timed_copy(&h->H, lu, 0);
}

int sluamd_zSetValues(sluamd_handle_t h, const sluamd_zLUview_t *lu)
{
    if (!h || !lu || !h->H.z) { set_error("not a complex16 handle"); return SLUAMD_EINVAL; }
    return sluamd_dSetValues(h, reinterpret_cast<const sluamd_dLUview_t *>(lu));
}

int sluamd_pdgstrf3d(sluamd_handle_t h, double thresh, int *info)
{
    if (!h) { set_error("null handle"); return SLUAMD_EINVAL; }
    if (h->H.z) { set_error("complex16 handle: call sluamd_pzgstrf3d"); return SLUAMD_EINVAL; }
    return run_factor(&h->H, thresh, info);
}

int sluamd_pzgstrf3d(sluamd_handle_t h, double thresh, int *info)
{
    if (!h || !h->H.z) { set_error("not a complex16 handle"); return SLUAMD_EINVAL; }
    return run_factor(&h->H, thresh, info);
}

int sluamd_factor_info(sluamd_handle_t h, int *info, int *tiny)
{
    if (!h) return SLUAMD_EINVAL;
    int res[8];
    HIPCHK(hipMemcpy(res, h->H.d_info, sizeof(res), hipMemcpyDeviceToHost));
    if (info) *info = h->H.env.info_last ? res[4] : ((res[0] == 0x7fffffff) ? 0 : res[0]);
    if (tiny) *tiny = res[1];
    return 0;
}

int sluamd_dGetDiagInv(sluamd_handle_t h, int32_t k, double *Linv, double *Uinv)
{
    if (!h || h->H.z || !Linv || !Uinv) { set_error("bad sluamd_dGetDiagInv arguments"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    if (H->split.active) { set_error("sluamd_dGetDiagInv: the handle refined supernodes wider than 256 columns into pieces; their inverses are per piece"); return SLUAMD_EINVAL; }
    if (!H->factored) { set_error("sluamd_dGetDiagInv: no factorisation has run on the handle's current values (the inverses are those of the FACTORED diagonal blocks)"); return SLUAMD_EINVAL; }
    if (H->grid.Pr * H->grid.Pc > 1) { set_error("sluamd_dGetDiagInv: XY layers keep the inverses of their peers' blocks in a per-level scratch; use a 1 x 1 layer"); return SLUAMD_EINVAL; }
    if (k < 0 || k >= H->hs.nsupers || !(H->h_flags[k] & SNF_OWN_DIAG)) { set_error("sluamd_dGetDiagInv: this rank does not own the diagonal block of that supernode"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(H->device));
    int rc = ensure_inv(H);       // (computed by the factorisation already unless the panel solves ran as substitutions)
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(H->stream));
    const int64_t ns = H->hs.xsup[k + 1] - H->hs.xsup[k];
    std::vector<int64_t> off(1);
    HIPCHK(hipMemcpy(off.data(), H->T.sn_inv + k, sizeof(int64_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(Linv, H->T.inv + off[0], sizeof(double) * (size_t) (ns * ns), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(Uinv, H->T.inv + off[0] + ns * ns, sizeof(double) * (size_t) (ns * ns), hipMemcpyDeviceToHost));
    return 0;
}

int sluamd_dCopyLU2Host(sluamd_handle_t h, const sluamd_dLUview_t *lu)
{
    if (!h || !lu) { set_error("null argument"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(h->H.device));
    HIPCHK(hipStreamSynchronize(h->H.stream));
    return timed_copy(&h->H, lu, 1);
}

int sluamd_zCopyLU2Host(sluamd_handle_t h, const sluamd_zLUview_t *lu)
{
    if (!h || !h->H.z) { set_error("not a complex16 handle"); return SLUAMD_EINVAL; }
    return sluamd_dCopyLU2Host(h, reinterpret_cast<const sluamd_dLUview_t *>(lu));
}

// x: n x nrhs doublecomplex, column-major, host memory; overwritten by the solution of L U y = x
int sluamd_pzgstrs3d(sluamd_handle_t h, sluamd_doublecomplex *x, int64_t ldx, int32_t nrhs)
{
    if (!h || !h->H.z || !x || nrhs < 0 || ldx < h->H.hs.n) { set_error("bad complex solve arguments"); return SLUAMD_EINVAL; }
    if (nrhs == 0) return 0;
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    const int64_t need = 2 * ldx * nrhs;   // in doubles
    if (need > H->x_cap) {
        if (H->d_x) hipFree(H->d_x);
        H->d_x = nullptr; H->x_cap = 0;
        HIPCHK(hipMalloc((void **) &H->d_x, sizeof(double) * need));
        H->x_cap = need;
    }
    HIPCHK(hipMemcpy(H->d_x, x, sizeof(double) * need, hipMemcpyHostToDevice));
    hipStream_t s = H->stream;
    HIPCHK(hipEventRecord(H->ev0, s));
    {   // the same driver as the double path: level sweeps per Z level, Z exchanges / gather on grids (x seen as 2 n doubles there)
        const int rc = run_solve_dev(H, H->d_x, ldx, nrhs);
        if (rc) return rc;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(H->ev1, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev0, H->ev1));
    H->st.t_solve_ms = ms;
    HIPCHK(hipMemcpy(x, H->d_x, sizeof(double) * need, hipMemcpyDeviceToHost));
    return 0;
}

int sluamd_pdgstrs3d_dev(sluamd_handle_t h, double *d_x, int64_t ldx, int32_t nrhs)
{
    if (!h || !d_x || nrhs < 0 || ldx < h->H.hs.n) { set_error("bad solve arguments"); return SLUAMD_EINVAL; }
    if (h->H.z) { set_error("complex16 handle: call sluamd_pzgstrs3d"); return SLUAMD_EINVAL; }
    if (nrhs == 0) return 0;
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    HIPCHK(hipEventRecord(H->ev0, H->stream));
    int rc = run_solve_dev(H, d_x, ldx, nrhs);
    if (rc) return rc;
    HIPCHK(hipEventRecord(H->ev1, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev0, H->ev1));
    H->st.t_solve_ms = ms;
    return 0;
}

// B: host, this rank's m_loc rows (vs doubles per value), leading dimension ldb in values
static int solve_dist_host(sluamd_handle_t h, double *B, int64_t ldb, int32_t nrhs, int64_t m_loc, int64_t fst_row, const sluamd_int_t *perm_in,
                           const sluamd_int_t *perm_out, bool z)
{
    if (!h || nrhs < 0 || m_loc < 0 || fst_row < 0 || fst_row + m_loc > h->H.hs.n || (m_loc > 0 && (!B || ldb < m_loc))) { set_error("bad distributed-solve arguments"); return SLUAMD_EINVAL; }
    if (h->H.z != z) { set_error(z ? "double handle: call sluamd_pdgstrs3d_dist" : "complex16 handle: call sluamd_pzgstrs3d_dist"); return SLUAMD_EINVAL; }
    if (nrhs == 0) return 0;
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    const int vs = z ? 2 : 1;
    const int64_t ldd = std::max<int64_t>(m_loc, 1);
    const int64_t need = ldd * nrhs * vs;
    if (need > H->bloc_cap) {
        if (H->d_bloc) hipFree(H->d_bloc);
        H->d_bloc = nullptr; H->bloc_cap = 0;
        HIPCHK(hipMalloc((void **) &H->d_bloc, sizeof(double) * (size_t) need));
        H->bloc_cap = need;
    }
    for (int q = 0; q < nrhs && m_loc; ++q) HIPCHK(hipMemcpy(H->d_bloc + (size_t) q * ldd * vs, B + (size_t) q * ldb * vs, sizeof(double) * (size_t) m_loc * vs, hipMemcpyHostToDevice));
    HIPCHK(hipEventRecord(H->ev0, H->stream));
    int rc = run_solve_dist(H, H->d_bloc, ldd, nrhs, m_loc, fst_row, perm_in, perm_out);
    if (rc) return rc;
    HIPCHK(hipEventRecord(H->ev1, H->stream));
    HIPCHK(hipStreamSynchronize(H->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, H->ev0, H->ev1));
    H->st.t_solve_ms = ms;
    for (int q = 0; q < nrhs && m_loc; ++q) HIPCHK(hipMemcpy(B + (size_t) q * ldb * vs, H->d_bloc + (size_t) q * ldd * vs, sizeof(double) * (size_t) m_loc * vs, hipMemcpyDeviceToHost));
    return 0;
}

int sluamd_pdgstrs3d_dist(sluamd_handle_t h, double *B, int64_t ldb, int32_t nrhs, int64_t m_loc, int64_t fst_row, const sluamd_int_t *perm_in,
                          const sluamd_int_t *perm_out)
{
    return solve_dist_host(h, B, ldb, nrhs, m_loc, fst_row, perm_in, perm_out, false);
}

int sluamd_pzgstrs3d_dist(sluamd_handle_t h, sluamd_doublecomplex *B, int64_t ldb, int32_t nrhs, int64_t m_loc, int64_t fst_row,
                          const sluamd_int_t *perm_in, const sluamd_int_t *perm_out)
{
    return solve_dist_host(h, reinterpret_cast<double *>(B), ldb, nrhs, m_loc, fst_row, perm_in, perm_out, true);
}

int sluamd_pdgstrs3d(sluamd_handle_t h, double *x, int64_t ldx, int32_t nrhs)
{
    if (!h || !x || nrhs < 0 || ldx < h->H.hs.n) { set_error("bad solve arguments"); return SLUAMD_EINVAL; }
    if (nrhs == 0) return 0;
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    const int64_t need = ldx * nrhs;
    if (need > H->x_cap) {
        if (H->d_x) hipFree(H->d_x);
        H->d_x = nullptr; H->x_cap = 0;
        HIPCHK(hipMalloc((void **) &H->d_x, sizeof(double) * need));
        H->x_cap = need;
    }
    HIPCHK(hipMemcpy(H->d_x, x, sizeof(double) * need, hipMemcpyHostToDevice));
    int rc = sluamd_pdgstrs3d_dev(h, H->d_x, ldx, nrhs);
    if (rc) return rc;
    HIPCHK(hipMemcpy(x, H->d_x, sizeof(double) * need, hipMemcpyDeviceToHost));
    return 0;
}

// ---- iterative refinement: pdgsrfs3d (SRC/double/pdgsrfs.c:345-510), SURVEY 8(f)-2 ----
static void free_rfs(Handle *H)
{
    void **ps[] = {(void **) &H->d_rfs_rp, (void **) &H->d_rfs_ci, (void **) &H->d_rfs_pc, (void **) &H->d_rfs_av, (void **) &H->d_rfs_work, (void **) &H->d_rfs_s};
    for (void **p : ps) { if (*p) hipFree(*p); *p = nullptr; }
    H->rfs_nnz = 0;
}

int sluamd_dAttachMatrix(sluamd_handle_t h, sluamd_int_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind, const double *nzval,
                         const sluamd_int_t *perm_c)
{
    if (!h || !rowptr || !colind || !nzval || !perm_c || n != h->H.hs.n) { set_error("bad matrix arguments"); return SLUAMD_EINVAL; }
    if (h->H.z) { set_error("iterative refinement is double precision only"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    const int64_t nnz = rowptr[n];
    free_rfs(H);
    auto bail = [&](const char *what) { set_error(std::string(what) + " failed in sluamd_dAttachMatrix"); free_rfs(H); return SLUAMD_EHIP; };
    if (hipMalloc((void **) &H->d_rfs_rp, sizeof(int) * (n + 1)) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void **) &H->d_rfs_ci, sizeof(int) * std::max<int64_t>(nnz, 1)) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void **) &H->d_rfs_av, sizeof(double) * std::max<int64_t>(nnz, 1)) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void **) &H->d_rfs_pc, sizeof(int) * n) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void **) &H->d_rfs_work, sizeof(double) * 3 * (size_t) n) != hipSuccess) return bail("hipMalloc");   // r_perm | b | x
    if (hipMalloc((void **) &H->d_rfs_s, sizeof(unsigned long long)) != hipSuccess) return bail("hipMalloc");
    if (hipMemcpy(H->d_rfs_rp, rowptr, sizeof(int) * (n + 1), hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy");
    if (hipMemcpy(H->d_rfs_ci, colind, sizeof(int) * nnz, hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy");
    if (hipMemcpy(H->d_rfs_av, nzval, sizeof(double) * nnz, hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy");
    if (hipMemcpy(H->d_rfs_pc, perm_c, sizeof(int) * n, hipMemcpyHostToDevice) != hipSuccess) return bail("hipMemcpy");
    H->rfs_nnz = nnz;
    return 0;
}

// d_B, d_X: device-resident, original ordering, column-major; X holds the initial solution and is refined in place
int sluamd_pdgsrfs3d_dev(sluamd_handle_t h, const double *d_B, int64_t ldb, double *d_X, int64_t ldx, int32_t nrhs, double *berr, int32_t *steps)
{
    if (!h || !d_B || !d_X || !berr || nrhs < 0 || ldb < h->H.hs.n || ldx < h->H.hs.n) { set_error("bad refinement arguments"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    if (!H->d_rfs_rp) { set_error("no matrix attached: call sluamd_dAttachMatrix first"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(H->device));
    const int n = (int) H->hs.n;
    const int ITMAX = 20;                                   // pdgsrfs.c:371
    const double eps = 0x1p-53, safmin = 2.2250738585072014e-308;
    const double safe1 = (double) (n + 1) * safmin, safe2 = safe1 / eps;
    double *r_perm = H->d_rfs_work;
    hipStream_t s = H->stream;
    int count = 0;
    for (int j = 0; j < nrhs; ++j) {
        const double *Bc = d_B + (size_t) j * ldb;
        double *Xc = d_X + (size_t) j * ldx;
        double lstres = 3.0;
        count = 0;
        for (;;) {
            HIPCHK(hipMemsetAsync(H->d_rfs_s, 0, sizeof(unsigned long long), s));
            eng::rfs_residual(s, n, H->d_rfs_rp, H->d_rfs_ci, H->d_rfs_av, Xc, Bc, H->d_rfs_pc, r_perm, H->d_rfs_s, safe1, safe2);
            double sv = 0.0;
            HIPCHK(hipMemcpyAsync(&sv, H->d_rfs_s, sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            berr[j] = sv;
            if (sv > eps && sv * 2 <= lstres && count < ITMAX) {
                int rc = run_solve_dev(H, r_perm, n, 1);
                if (rc) return rc;
                eng::rfs_update(s, n, H->d_rfs_pc, r_perm, Xc);
                lstres = sv;
                ++count;
            } else break;
        }
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (steps) *steps = count;
    return 0;
}

int sluamd_pdgsrfs3d(sluamd_handle_t h, const double *B, int64_t ldb, double *X, int64_t ldx, int32_t nrhs, double *berr, int32_t *steps)
{
    if (!h || !B || !X || !berr || nrhs < 0 || ldb < h->H.hs.n || ldx < h->H.hs.n) { set_error("bad refinement arguments"); return SLUAMD_EINVAL; }
    if (nrhs == 0) { if (steps) *steps = 0; return 0; }
    Handle *H = &h->H;
    if (!H->d_rfs_rp) { set_error("no matrix attached: call sluamd_dAttachMatrix first"); return SLUAMD_EINVAL; }
    HIPCHK(hipSetDevice(H->device));
    const int64_t n = H->hs.n;
    double *d_b = H->d_rfs_work + n, *d_x = H->d_rfs_work + 2 * (size_t) n;
    int last = 0;
    for (int j = 0; j < nrhs; ++j) {   // one column at a time through the two resident work vectors
        HIPCHK(hipMemcpy(d_b, B + (size_t) j * ldb, sizeof(double) * n, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_x, X + (size_t) j * ldx, sizeof(double) * n, hipMemcpyHostToDevice));
        int rc = sluamd_pdgsrfs3d_dev(h, d_b, n, d_x, n, 1, berr + j, &last);
        if (rc) return rc;
        HIPCHK(hipMemcpy(X + (size_t) j * ldx, d_x, sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    if (steps) *steps = last;
    return 0;
}

void sluamd_dDestroyLUHandle(sluamd_handle_t h)
{
    if (!h) return;
    Handle *H = &h->H;
    hipSetDevice(H->device);
    if (H->stream) hipStreamSynchronize(H->stream);
    if (H->pstream) hipStreamSynchronize(H->pstream);
    if (H->ustream) hipStreamSynchronize(H->ustream);
    if (H->u2stream) hipStreamSynchronize(H->u2stream);
    if (H->rstream) hipStreamSynchronize(H->rstream);
    for (void *p : H->d_misc) hipFree(p);
    if (H->d_val) devpool_free(H->d_val);       // (plain allocations are recognised and freed as such)
    if (H->d_info) hipFree(H->d_info);
    if (H->d_x) hipFree(H->d_x);
    if (H->d_xtmp) hipFree(H->d_xtmp);
    if (H->d_w) hipFree(H->d_w);
    if (H->d_apos) hipFree(H->d_apos);
    if (H->d_aval) hipFree(H->d_aval);
    if (H->h_pinned) hipHostFree(H->h_pinned);
    if (H->d_bloc) hipFree(H->d_bloc);
    for (void *q : H->dist.bufs) hipFree(q);
    free_rfs(H);
    for (auto &e : H->ev_schur) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : H->ev_panel) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : H->ev_xchg) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto &e : H->ev_red) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (H->ev0) hipEventDestroy(H->ev0);
    if (H->ev1) hipEventDestroy(H->ev1);
    for (auto e : H->ev_pool) hipEventDestroy(e);
    if (H->pstream) hipStreamDestroy(H->pstream);
    if (H->ustream) hipStreamDestroy(H->ustream);
    if (H->u2stream) hipStreamDestroy(H->u2stream);
    if (H->rstream) hipStreamDestroy(H->rstream);
    if (H->gstream) hipStreamDestroy(H->gstream);
    for (auto e : H->red_pool) hipEventDestroy(e);
    if (H->red_all) hipEventDestroy(H->red_all);
    if (H->stream) hipStreamDestroy(H->stream);
    delete h;
}

int sluamd_setup_times(sluamd_handle_t h, char *buf, int32_t cap)
{
    if (!h || !buf || cap < 1) return SLUAMD_EINVAL;
    std::string s;
    for (auto &l : h->H.setup.laps) { char t[96]; snprintf(t, sizeof t, "%s%s=%.6f", s.empty() ? "" : ";", l.first.c_str(), l.second); s += t; }
    if ((int) s.size() + 1 > cap) return SLUAMD_EINVAL;
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

// One row of SLUAMD_PLAN_COLS doubles per (Z level, DAG level) of THIS rank's schedule -- what the level costs this rank, from the plan alone:
//   0 Z level, 1 DAG level, 2 supernodes, 3 widest supernode, 4 exact-segment Schur flops of this rank's tiles, 5 its planned Schur tile executions,
//   6 diagonal LU + panel-solve flops it owns, 7/8 bytes / messages it SENDS in exchange phase 1 (factored diagonal blocks), 9/10 bytes / messages it receives there,
//   11/12 bytes / messages sent in phase 2 (L panels along the process row, U panels down the process column), 13/14 received there,
//   15 bytes this rank sends (+) or receives (-) in the Z reduction of the ancestors that follows the Z level (on the row of the Z level's LAST DAG level; 0 elsewhere),
//   16/17 bytes to / from the busiest single peer in phase 1, 18/19 in phase 2
int sluamd_plan_table(sluamd_handle_t h, double *buf, int64_t cap_rows, int64_t *rows)
{
    if (!h || !rows) { set_error("null argument"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    const int64_t esz = H->z ? 16 : 8;
    int64_t n = 0;
    for (size_t zl = 0; zl < H->sched.size(); ++zl) {
        const LevelSched &S = H->sched[zl];
        for (int l = 0; l < S.nlevels; ++l, ++n) {
            if (!buf || n >= cap_rows) continue;
            double *r = buf + SLUAMD_PLAN_COLS * n;
            for (int c = 0; c < SLUAMD_PLAN_COLS; ++c) r[c] = 0.0;
            r[0] = (double) zl; r[1] = l; r[2] = S.lvl_off[l + 1] - S.lvl_off[l]; r[3] = S.max_nsupc.empty() ? 0 : S.max_nsupc[l];
            r[4] = l < (int) S.lvl_flops_schur.size() ? S.lvl_flops_schur[l] : 0.0;
            if (!S.u_off.empty()) r[5] = (double) (S.u_off[8 * (l + 1)] - S.u_off[8 * l]);
            r[6] = l < (int) S.lvl_flops_panel.size() ? S.lvl_flops_panel[l] : 0.0;
            auto tally = [&](const std::vector<std::vector<XMsg>> &v, double &bytes, double &msgs) {
                if (l < (int) v.size()) for (const XMsg &m : v[l]) { bytes += (double) (esz * m.len); msgs += 1.0; }
            };
            tally(S.x_diag_send, r[7], r[8]); tally(S.x_diag_recv, r[9], r[10]);
            tally(S.x_panel_send, r[11], r[12]); tally(S.x_panel_recv, r[13], r[14]);
            // 16..19: the most bytes one PEER gets from / sends to this rank in phase 1 / phase 2 (xGMI is point to point: a phase lasts what its busiest link takes)
            auto busiest = [&](const std::vector<std::vector<XMsg>> &v) {
                double mx = 0.0;
                if (l < (int) v.size()) for (const XMsg &m : v[l]) { double b = 0.0; for (const XMsg &q : v[l]) if (q.peer == m.peer) b += (double) (esz * q.len); mx = std::max(mx, b); }
                return mx;
            };
            r[16] = busiest(S.x_diag_send); r[17] = busiest(S.x_diag_recv); r[18] = busiest(S.x_panel_send); r[19] = busiest(S.x_panel_recv);
            if (l == S.nlevels - 1 && H->grid.Pz > 1 && zl + 1 < H->forest_nodes.size()) {
                // dreduceAllAncestors3d after Z level zl (pd3dcomm.c:1046-1081): layer z + 2^zl sends its copies of every ancestor forest to layer z (z % 2^(zl+1) == 0)
                const int step = 1 << zl, z = H->grid.z;
                const bool recv = z % (2 * step) == 0 && z + step < H->grid.Pz, send = z % (2 * step) == step;
                if (recv || send) {
                    double b = 0.0;
                    for (size_t a = zl + 1; a < H->forest_nodes.size(); ++a)
                        for (int k : H->forest_nodes[a]) {
                            if (H->grid.kcol(k) == H->grid.c) b += (double) (esz * H->hs.lval_len[k]);
                            if (H->grid.krow(k) == H->grid.r) b += (double) (esz * H->hs.uval_len[k]);
                        }
                    r[15] = send ? b : -b;
                }
            }
        }
    }
    *rows = n;
    return 0;
}

int sluamd_get_stats(sluamd_handle_t h, sluamd_stats_t *out)
{
    if (!h || !out) return SLUAMD_EINVAL;
    *out = h->H.st;
    return 0;
}

int sluamd_dCreateLUHandleFromSymb(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                                   const double *nzval, const sluamd_int_t *perm_c_final, const sluamd_options_t *opt)
{
    return create_from_symb(out, s, rowptr, colind, nzval, perm_c_final, opt, Grid{}, nullptr, nullptr, false);
}

// complex16 values (nzval = doublecomplex[nnz] aligned with colind), 1x1x1 grid
int sluamd_zCreateLUHandleFromSymb(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                                   const sluamd_doublecomplex *nzval, const sluamd_int_t *perm_c_final, const sluamd_options_t *opt)
{
    return create_from_symb(out, s, rowptr, colind, reinterpret_cast<const double *>(nzval), perm_c_final, opt, Grid{}, nullptr, nullptr, true);
}

int sluamd_dCreateLUHandleFromSymbGrid(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                                       const double *nzval, const sluamd_int_t *perm_c_final, const sluamd_options_t *opt,
                                       const int32_t *sn_tree, sluamd_comm_t comm)
{
    if (!comm) { set_error("null communicator"); return SLUAMD_EINVAL; }
    return create_from_symb(out, s, rowptr, colind, nzval, perm_c_final, opt, comm->c->grid, sn_tree, comm->c, false);
}

// complex16 values on a 1 x 1 x npdep grid
int sluamd_zCreateLUHandleFromSymbGrid(sluamd_handle_t *out, sluamd_symb_t s, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                                       const sluamd_doublecomplex *nzval, const sluamd_int_t *perm_c_final, const sluamd_options_t *opt,
                                       const int32_t *sn_tree, sluamd_comm_t comm)
{
    if (!comm) { set_error("null communicator"); return SLUAMD_EINVAL; }

    return create_from_symb(out, s, rowptr, colind, reinterpret_cast<const double *>(nzval), perm_c_final, opt, comm->c->grid, sn_tree, comm->c, true);
}

// Device-side re-distribution of A's values into the resident store (handles made by sluamd_dCreateLUHandleFromSymb*):
// zero-fill + scatter, asynchronous on the handle's stream.
int sluamd_dResetValues(sluamd_handle_t h)
{
    if (!h || !h->H.d_apos) { set_error("handle has no device-side copy of A"); return SLUAMD_EINVAL; }
    Handle *H = &h->H;
    HIPCHK(hipSetDevice(H->device));
    H->dinv_ready = false; H->inv_ready = false; H->factored = false;
    HIPCHK(hipMemsetAsync(H->d_val, 0, (H->z ? 16 : 8) * (size_t) H->own_len, H->stream));
    if (H->a_nnz && !H->z) eng::scatter_values(H->stream, H->d_val, H->d_apos, H->d_aval, H->a_nnz);
    if (H->a_nnz && H->z) eng::zscatter_values(H->stream, H->d_val, H->d_apos, H->d_aval, H->a_nnz);
    HIPCHK(hipGetLastError());
    return 0;
}

int sluamd_device_synchronize(void)
{
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

int sluamd_set_profile(sluamd_handle_t h, int on)
{
    if (!h) return SLUAMD_EINVAL;
    h->H.opt.verbose = on ? 2 : 0;
    if (!on) h->H.profile = h->H.env.profile;      // at once, not at the next factorisation: the solves after a profiled factorisation ran their serial (unjoined) form until then
    return 0;
}

// debug hook (not in the public header): accumulated phase timers of k_diag_lu2 [A, B, C, D+store, E] in shader clock ticks

// test hook: MFMA fp64 fragment layout check
int sluamd_mfma_selftest(const double *A, const double *B, double *D)
{
    int rc = check_device(-1);
    if (rc) return rc;
    return eng::mfma_selftest(A, B, D);
}

}  // extern "C"
