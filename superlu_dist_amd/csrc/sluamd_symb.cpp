// sluamd_symb.cpp -- host-side producer of the L/U store for the hot path when the reference's
// pre-processing is not linked (built only as far as the path needs):
// symmetric-pattern supernodal symbolic factorisation + distribution for a 1x1 process layer.
//
// Effect mirrors symbfact_dist (SRC/prec-independent/symbfact.c) + pddistribute3d
// (SRC/double/pddistribute3d.c:1357): same store formats (superlu_defs.h:156-198), relaxed supernodes
// like relax_snode (SRC/prec-independent/symbfact.c, sp_ienv_dist(2)), supernode width capped like
// sp_ienv_dist(3).  The algorithm is our own: elimination tree of A+A^T (Liu), postorder composed into
// perm_c (as sp_colorder does), supernodal structure by child-structure union.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <functional>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include "sluamd_internal.h"
#include "sluamd_plan.h"

using namespace sluamd;

namespace {

struct Graph {           // permuted symmetric pattern, strictly-lower and strictly-upper adjacency
    std::vector<int64_t> lo_off, up_off;
    std::vector<int> lo, up;   // lo: rows > col ; up: rows < col
};

void build_graph(int64_t n, const int *rowptr, const int *colind, const int *perm, Graph &g, bool want_lo = true, bool want_up = true)
{
    // (want_lo / want_up: the elimination tree reads only the upper lists, the structure pass only the lower ones -- each caller builds the half it uses)
    // two passes over the rows of A on the planner's threads: degrees (relaxed atomic increments), serial prefix sums, fill (atomic slot claims).
    // The order of the entries inside an adjacency list therefore varies from run to run; every consumer is order-independent (Liu's elimination
    // tree is unique, the structure pass sorts what it collects).
    std::vector<int64_t> cl(n + 1, 0), cu(n + 1, 0);
    parallel_chunks(n, 16384, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i)
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
                const int a = perm[i], b = perm[colind[e]];
                if (a == b) continue;
                const int lo = std::min(a, b), hi = std::max(a, b);
                if (want_lo) __atomic_fetch_add(&cl[lo + 1], 1, __ATOMIC_RELAXED);
                if (want_up) __atomic_fetch_add(&cu[hi + 1], 1, __ATOMIC_RELAXED);
            }
    });
    for (int64_t i = 0; i < n; ++i) { cl[i + 1] += cl[i]; cu[i + 1] += cu[i]; }
    g.lo_off = cl; g.up_off = cu;
    g.lo.resize(cl[n]); g.up.resize(cu[n]);
    std::vector<int64_t> pl(cl.begin(), cl.end() - 1), pu(cu.begin(), cu.end() - 1);
    parallel_chunks(n, 16384, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i)
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
                const int a = perm[i], b = perm[colind[e]];
                if (a == b) continue;
                const int lo = std::min(a, b), hi = std::max(a, b);
                if (want_lo) g.lo[__atomic_fetch_add(&pl[lo], 1, __ATOMIC_RELAXED)] = hi;
                if (want_up) g.up[__atomic_fetch_add(&pu[hi], 1, __ATOMIC_RELAXED)] = lo;
            }
    });
}

// perm_c -> validated permutation composed with a postorder of the elimination tree of Pc (A + A^T) Pc^T (what sp_colorder does for the
// reference, sp_colorder.c:137-166: sp_symetree_dist + postorder), the tree in the final labels, and the symmetric adjacency in the final labels
static int order_and_etree(int64_t n, const int *rowptr, const int *colind, const int *perm_c, std::vector<int> &perm, std::vector<int> &parent,
                           Graph &g, int *perm_c_out, const std::function<void(const char *)> &lap)
{
    perm.resize(n);
    if (perm_c) std::copy(perm_c, perm_c + n, perm.begin()); else std::iota(perm.begin(), perm.end(), 0);
    {   // validate permutation
        std::vector<char> seen(n, 0);
        for (int64_t i = 0; i < n; ++i) { if (perm[i] < 0 || perm[i] >= n || seen[perm[i]]) { set_error("perm_c is not a permutation"); return SLUAMD_EINVAL; } seen[perm[i]] = 1; }
    }
    build_graph(n, rowptr, colind, perm.data(), g, false, true);      // upper lists: the elimination tree
    lap("graph");
    // ---- elimination tree (Liu, path compression) ----
    parent.assign(n, -1);
    std::vector<int> anc(n, -1);
    for (int j = 0; j < n; ++j)
        for (int64_t e = g.up_off[j]; e < g.up_off[j + 1]; ++e) {
            int i = g.up[e];
            while (i != -1 && i < j) {
                int nx = anc[i];
                anc[i] = j;
                if (nx == -1) parent[i] = j;
                i = nx;
            }
        }
    // ---- postorder (children in increasing order, iterative DFS) ----
    std::vector<int> head(n, -1), next(n, -1), post(n), newlab(n);
    for (int j = (int) n - 1; j >= 0; --j) if (parent[j] != -1) { next[j] = head[parent[j]]; head[parent[j]] = j; }
    {
        int k = 0;
        std::vector<int> stack;
        for (int r = 0; r < n; ++r) {
            if (parent[r] != -1) continue;
            stack.push_back(r);
            while (!stack.empty()) {
                int v = stack.back();
                int c = head[v];
                if (c == -1) { post[k] = v; newlab[v] = k++; stack.pop_back(); }
                else { head[v] = next[c]; stack.push_back(c); }
            }
        }
    }
    // compose postorder into the permutation and relabel
    for (int64_t i = 0; i < n; ++i) perm[i] = newlab[perm[i]];
    if (perm_c_out) std::copy(perm.begin(), perm.end(), perm_c_out);
    {
        std::vector<int> p2(n, -1);
        for (int j = 0; j < n; ++j) if (parent[j] != -1) p2[newlab[j]] = newlab[parent[j]];
        parent.swap(p2);
    }
    lap("etree + postorder");
    build_graph(n, rowptr, colind, perm.data(), g, true, false);  // lower adjacency in final labels: the structure pass
    lap("graph (final labels)");
    return 0;
}

}  // namespace

extern "C" {

int64_t sluamd_poisson3d(int32_t nx, int32_t ny, int32_t nz, sluamd_int_t *rowptr, sluamd_int_t *colind, double *nzval)
{
    if (nx < 1 || ny < 1 || nz < 1 || !rowptr || !colind || !nzval) { set_error("bad sluamd_poisson3d arguments"); return SLUAMD_EINVAL; }
    const int64_t n = (int64_t) nx * ny * nz;
    const int64_t nnz = 7 * n - 2 * ((int64_t) nx * ny + (int64_t) ny * nz + (int64_t) nx * nz);
    if (n > 0x7fffffff || nnz > 0x7fffffff) { set_error("sluamd_poisson3d: the operator does not fit 32-bit CSR indices"); return SLUAMD_EINVAL; }
    // entries of row (i, j, k): the plane i holds ny nz rows of the same count pattern, so the row pointer of a plane start is a closed form
    // and planes are written independently
    std::vector<int64_t> plane_off(nx + 1, 0);
    for (int i = 0; i < nx; ++i) {
        const int64_t full = (int64_t) ny * nz * (5 + (i > 0) + (i < nx - 1));      // 1 + four in-plane neighbours + the planes before / after
        plane_off[i + 1] = plane_off[i] + full - 2 * (int64_t) nz - 2 * (int64_t) ny;     // in-plane neighbours missing on the plane's four edges
    }
    sluamd::parallel_chunks(nx, 1, [&](int64_t i0, int64_t i1) {
        for (int i = (int) i0; i < (int) i1; ++i) {
            int64_t p = plane_off[i];
            for (int j = 0; j < ny; ++j)
                for (int k = 0; k < nz; ++k) {
                    const int64_t r = ((int64_t) i * ny + j) * nz + k;
                    rowptr[r] = (sluamd_int_t) p;
                    if (i > 0) { colind[p] = (sluamd_int_t) (r - (int64_t) ny * nz); nzval[p++] = -1.0; }
                    if (j > 0) { colind[p] = (sluamd_int_t) (r - nz); nzval[p++] = -1.0; }
                    if (k > 0) { colind[p] = (sluamd_int_t) (r - 1); nzval[p++] = -1.0; }
                    colind[p] = (sluamd_int_t) r; nzval[p++] = 6.0;
                    if (k < nz - 1) { colind[p] = (sluamd_int_t) (r + 1); nzval[p++] = -1.0; }
                    if (j < ny - 1) { colind[p] = (sluamd_int_t) (r + nz); nzval[p++] = -1.0; }
                    if (i < nx - 1) { colind[p] = (sluamd_int_t) (r + (int64_t) ny * nz); nzval[p++] = -1.0; }
                }
        }
    });
    rowptr[n] = (sluamd_int_t) nnz;
    return plane_off[nx] == nnz ? nnz : (int64_t) SLUAMD_ESTRUCT;
}

int sluamd_dsymbfact(sluamd_symb_t *out, int64_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                     const sluamd_int_t *perm_c, int32_t relax, int32_t maxsup, sluamd_int_t *perm_c_out)
{
    if (!out || n <= 0 || !rowptr || !colind) { set_error("bad symbfact arguments"); return SLUAMD_EINVAL; }
    if (maxsup < 1) maxsup = 256;
    if (maxsup > 512) maxsup = 512;  // MAX_SUPER_SIZE, superlu_defs.h:154
    if (relax < 1) relax = 1;
    if (relax > maxsup) relax = maxsup;
    double amalg_frac = 0.05;  // tolerated explicit-zero fraction when amalgamating along etree chains
    if (const char *e = getenv("SLUAMD_AMALG_FRAC")) amalg_frac = atof(e);
    std::vector<int> perm, parent;
    Graph g;
    static const bool timing = getenv("SLUAMD_SYMB_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[sluamd_dsymbfact] %-28s %.3f s\n", what, t - t_prev); t_prev = t; } };
    if (int rc = order_and_etree(n, rowptr, colind, perm_c, perm, parent, g, perm_c_out, lap)) return rc;
    // ---- subtree sizes, child counts, relaxed subtree roots ----
    std::vector<int> sz(n, 1), nchild(n, 0);
    for (int j = 0; j < n; ++j) if (parent[j] != -1) { sz[parent[j]] += sz[j]; nchild[parent[j]]++; }
    // length of the etree chain starting at each column (upper bound of what one supernode chain can absorb)
    std::vector<int> chain(n, 1);
    for (int j = (int) n - 2; j >= 0; --j) if (parent[j] == j + 1) chain[j] = chain[j + 1] + 1;
    std::vector<int> relax_end(n, -1);  // relax_end[a] = b if [a,b] is a relaxed subtree
    for (int j = 0; j < n; ++j)
        if (sz[j] <= relax && (parent[j] == -1 || sz[parent[j]] > relax)) relax_end[j - sz[j] + 1] = j;
    // ---- supernodal structure ----
    // One supernode ("unit") at a time in column order: its row structure is the union of the adjacency of its columns and of the row structures of the
    // units pending on them (children in the supernodal tree).  In parallel: the etree is postordered, so a subtree is a contiguous column range that needs
    // nothing from outside it -- disjoint subtrees of at most `cut` columns are tasks for the worker threads; what is left (the supernode that holds a task's
    // root column -- it may continue into the parent, which waits for OTHER subtrees -- and everything above the cut) runs serially afterwards with the task
    // units as its children.  The result is the serial one (order of discovery only feeds sorted lists and counts), whatever the number of threads.
    auto *sy = new Symb();
    HostStruct &hs = sy->hs;
    hs.n = n;
    sy->supno.assign(n, -1);
    struct Unit { int first, last, ctx; int64_t off; int len; };
    struct Ctx { std::vector<int> rows; std::vector<Unit> units; std::vector<std::pair<int, int>> pend_out; int stop = 0; };      // pend_out: (local unit, column it pends on) beyond the task
    static const bool equal_split = getenv("SLUAMD_SYMB_EQUAL_SPLIT") != nullptr;   // development: round-2 rule (equal pieces)
    const bool run_merge = getenv("SLUAMD_SYMB_NO_RUN_MERGE") == nullptr;           // tests: sort every unit's rows instead of merging the child runs
    // grows the unit that starts at column a.  children(c, f): f(rows, len) for every unit pending on column c; lim: last column the unit may examine
    // (task: its root; returns -1 when the unit reaches it -- not this task's to finish).  Appends the sorted rows > b to `out`, returns b.
    auto grow = [&](int a, int lim, int stamp, std::vector<int> &mark, std::vector<int> &cur, std::vector<int> &extra, auto &&children, std::vector<int> &out) -> int {
        int b;
        const bool bounded = lim < (int) n;
        cur.clear();
        int nrun = 0;
        size_t run_off[10] = {0};
        auto add_adj = [&](int c) {
            for (int64_t e = g.lo_off[c]; e < g.lo_off[c + 1]; ++e) { int r = g.lo[e]; if (mark[r] != stamp) { mark[r] = stamp; cur.push_back(r); } }
        };
        if (relax_end[a] >= 0) {
            b = relax_end[a];
            if (bounded && b >= lim) return -1;
            for (int c = a; c <= b; ++c) add_adj(c);  // full subtree: no child unit lies outside [a,b]
        } else {
            b = a;
            if (bounded && a >= lim) return -1;
            add_adj(a);
            // the rows taken from one child unit stay ascending (a subsequence of its sorted list): remembered as runs, merged instead of sorted below
            nrun = 0; run_off[0] = cur.size();
            children(a, [&](const int *rows, int len) {
                for (int i = 0; i < len; ++i) { const int r = rows[i]; if (mark[r] != stamp) { mark[r] = stamp; cur.push_back(r); } }
                if (nrun < 8) run_off[++nrun] = cur.size(); else nrun = 9;
            });
            // extend the supernode by column c = b+1 (its etree parent) while struct(c) ~= struct(b) \ {c}.
            // Non-fundamental supernodes are allowed (other children of c may hang anywhere), and so is a small
            // amount of explicit-zero padding (relaxed amalgamation): where two ND separators meet, every
            // separator column brings ONE private row of the ancestor separator, which would otherwise shatter
            // the separator into singleton supernodes.
            double zacc = 0;
            // split long chains (separators) into pieces of exactly maxsup columns (full 128-wide Schur tiles and K chunks
            // when maxsup is a multiple of 128); the tail of a chain is never a sliver: a remainder below maxsup / 2 is
            // merged with the piece before it and that is halved (to a multiple of 16)
            const int rem = chain[a];
            int cap = maxsup;
            if (equal_split) { const int pieces = (rem + maxsup - 1) / maxsup; cap = (rem + pieces - 1) / pieces; } else
            if (rem <= maxsup) cap = rem;
            else if (rem < 2 * maxsup && rem - maxsup < maxsup / 2) cap = std::min(maxsup, ((rem + 1) / 2 + 15) & ~15);
            while (b + 1 < n && (b - a + 1) < cap && parent[b] == b + 1 && relax_end[b + 1] < 0) {
                const int c = b + 1;
                if (bounded && c >= lim) return -1;
                extra.clear();
                for (int64_t e = g.lo_off[c]; e < g.lo_off[c + 1]; ++e) { int r = g.lo[e]; if (mark[r] != stamp) { mark[r] = stamp; extra.push_back(r); } }
                children(c, [&](const int *rows, int len) { for (int i = 0; i < len; ++i) { const int r = rows[i]; if (r > c && mark[r] != stamp) { mark[r] = stamp; extra.push_back(r); } } });
                const double cols = c - a, rows = (double) cur.size();
                const double znew = zacc + (double) extra.size() * cols;
                const bool ok = extra.empty() ||
                                ((double) extra.size() <= amalg_frac * rows + 2.0 && znew <= 2.0 * amalg_frac * rows * (cols + 1) + 16.0);
                if (!ok) { for (int r : extra) mark[r] = -1; break; }
                zacc = znew;
                cur.insert(cur.end(), extra.begin(), extra.end());
                b = c;
            }
        }
        // finalize: rows > b, sorted.  A separator piece takes tens of thousands of rows from ONE or two children (the previous piece of its chain, the two
        // subtrees below a separator's first column) and a handful from its own columns: sort the handfuls, merge the runs -- the sort of every piece was the
        // serial share of this pass at 150^3
        if (run_merge && nrun >= 1 && nrun <= 8 && cur.size() > 2048) {
            std::sort(cur.begin(), cur.begin() + run_off[0]);
            std::sort(cur.begin() + run_off[nrun], cur.end());
            for (int q = 0; q < nrun; ++q) std::inplace_merge(cur.begin(), cur.begin() + run_off[q], cur.begin() + run_off[q + 1]);
            std::inplace_merge(cur.begin(), cur.begin() + run_off[nrun], cur.end());
            const size_t skip = std::upper_bound(cur.begin(), cur.end(), b) - cur.begin();      // the unit's own columns a + 1 .. b
            out.insert(out.end(), cur.begin() + skip, cur.end());
            return b;
        }
        size_t w = 0;
        for (size_t i = 0; i < cur.size(); ++i) if (cur[i] > b) cur[w++] = cur[i];
        cur.resize(w);
        std::sort(cur.begin(), cur.end());
        out.insert(out.end(), cur.begin(), cur.end());
        return b;
    };
    // tasks: the maximal subtrees of at most `cut` columns (roots in column order)
    const int nthr = plan_threads();
    int cut = (int) std::max<int64_t>(2048, n / (8 * (int64_t) std::max(nthr, 1))), tmin = 256;
    bool tasks_on = nthr > 1 && n >= 50000;
    if (const char *e = getenv("SLUAMD_SYMB_CUT")) { cut = atoi(e); tmin = 1; tasks_on = cut > 0; }      // tests: the task path on small structures
    std::vector<int> troot;
    if (tasks_on)
        for (int jj = 0; jj < n; ++jj)
            if (sz[jj] <= cut && sz[jj] >= tmin && (parent[jj] == -1 || sz[parent[jj]] > cut)
                && (parent[jj] == -1 || sz[parent[jj]] > relax))      // never strictly INSIDE a relaxed subtree (cut < relax): its unit [a, b] is one task's or the serial pass's, whole
                troot.push_back(jj);
    std::vector<Ctx> ctx(troot.size() + 1);
    std::vector<int> skip_to(n, -1);      // first column of a task's range -> first column the task left
    // one mark / cur / extra set per WORKER, handed from chunk to chunk with its stamp (a fresh n-sized mark array per task cost O(n * #tasks) writes: a forest of
    // thousands of 256..cut-column subtrees ran slower threaded than serial)
    struct Scratch { std::vector<int> mark, cur, extra; int stamp = 0; };
    std::vector<std::unique_ptr<Scratch>> pool;
    std::mutex pool_mu;
    parallel_chunks((int64_t) troot.size(), 1, [&](int64_t t0, int64_t t1) {
        std::unique_ptr<Scratch> sc;
        { std::lock_guard<std::mutex> g(pool_mu); if (!pool.empty()) { sc = std::move(pool.back()); pool.pop_back(); } }
        if (!sc) { sc.reset(new Scratch()); sc->mark.assign(n, -1); }
        std::vector<int> &mark = sc->mark, &cur = sc->cur, &extra = sc->extra;
        int &stamp = sc->stamp;
        struct Back { std::unique_ptr<Scratch> &sc; std::vector<std::unique_ptr<Scratch>> &pool; std::mutex &mu;
                      ~Back() { std::lock_guard<std::mutex> g(mu); pool.push_back(std::move(sc)); } } back{sc, pool, pool_mu};
        for (int64_t t = t0; t < t1; ++t) {
            Ctx &cx = ctx[t];
            const int root = troot[t], lo = root - sz[root] + 1, len = root - lo + 1;
            std::vector<int> pend_head(len, -1), pend_next;
            auto children = [&](int c, auto &&f) {
                for (int cu = pend_head[c - lo]; cu != -1; cu = pend_next[cu]) f(cx.rows.data() + cx.units[cu].off, cx.units[cu].len);
            };
            int j = lo;
            while (j <= root) {
                const int64_t off = (int64_t) cx.rows.size();
                const int b = grow(j, root, stamp++, mark, cur, extra, children, cx.rows);
                if (b < 0) break;                 // the unit that reaches the root: left to the serial pass
                const int u = (int) cx.units.size();
                cx.units.push_back(Unit{j, b, (int) t, off, (int) (cx.rows.size() - off)});
                pend_next.push_back(-1);
                const int pb = parent[b];
                if (pb != -1) {
                    if (pb <= root) { pend_next[u] = pend_head[pb - lo]; pend_head[pb - lo] = u; }
                    else cx.pend_out.push_back({u, pb});
                }
                j = b + 1;
            }
            cx.stop = j;
            // units pending on columns the task did not finish go to the serial pass too
            for (int c = std::max(j, lo); c <= root; ++c) for (int cu = pend_head[c - lo]; cu != -1; cu = pend_next[cu]) cx.pend_out.push_back({cu, c});
            if (j > lo) skip_to[lo] = j;
        }
    });
    lap("structure: subtree tasks");
    // the serial pass over what is left, the task units as children
    std::vector<Unit> units;
    for (size_t t = 0; t < troot.size(); ++t) units.insert(units.end(), ctx[t].units.begin(), ctx[t].units.end());
    {
        Ctx &cs = ctx.back();
        const int sctx = (int) troot.size();
        std::vector<int> pend_head(n, -1), pend_next(units.size(), -1);
        size_t base = 0;
        for (size_t t = 0; t < troot.size(); ++t) {
            for (const auto &po : ctx[t].pend_out) { const int u = (int) base + po.first; pend_next[u] = pend_head[po.second]; pend_head[po.second] = u; }
            base += ctx[t].units.size();
        }
        auto children = [&](int c, auto &&f) {
            for (int cu = pend_head[c]; cu != -1; cu = pend_next[cu]) { const Unit &q = units[cu]; f(ctx[q.ctx].rows.data() + q.off, q.len); }
        };
        std::vector<int> mark(n, -1), cur, extra;
        int stamp = 0, j = 0;
        int64_t dbg_units = 0, dbg_cols = 0, dbg_rows = 0;
        while (j < n) {
            if (skip_to[j] > j) { j = skip_to[j]; continue; }
            const int64_t off = (int64_t) cs.rows.size();
            const int b = grow(j, (int) n, stamp++, mark, cur, extra, children, cs.rows);
            const int u = (int) units.size();
            units.push_back(Unit{j, b, sctx, off, (int) (cs.rows.size() - off)});
            pend_next.push_back(-1);
            if (parent[b] != -1) { pend_next[u] = pend_head[parent[b]]; pend_head[parent[b]] = u; }
            dbg_units++; dbg_cols += b - j + 1; dbg_rows += (int64_t) (cs.rows.size() - off);
            j = b + 1;
        }
        if (timing) fprintf(stderr, "[sluamd_dsymbfact] serial pass: %lld units, %lld columns, %lld rows kept; %zu task units\n", (long long) dbg_units, (long long) dbg_cols, (long long) dbg_rows, units.size() - (size_t) dbg_units);
    }
    lap("structure: serial top");
    // units in column order; rows gathered into one array
    std::vector<int> uord(units.size());
    std::iota(uord.begin(), uord.end(), 0);
    std::sort(uord.begin(), uord.end(), [&](int x, int y) { return units[x].first < units[y].first; });
    std::vector<int> ufirst(units.size()), ulast(units.size());
    sy->srow_off.assign(units.size() + 1, 0);
    for (size_t i = 0; i < uord.size(); ++i) { const Unit &q = units[uord[i]]; ufirst[i] = q.first; ulast[i] = q.last; sy->srow_off[i + 1] = sy->srow_off[i] + q.len; }
    sy->srows.resize(sy->srow_off[units.size()]);
    parallel_chunks((int64_t) uord.size(), 64, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) {
            const Unit &q = units[uord[i]];
            std::copy(ctx[q.ctx].rows.data() + q.off, ctx[q.ctx].rows.data() + q.off + q.len, sy->srows.data() + sy->srow_off[i]);
            for (int c = q.first; c <= q.last; ++c) sy->supno[c] = (int) i;
        }
    });
    std::vector<Ctx>().swap(ctx);
    lap("structure: gather");
    const int ns = (int) ufirst.size();
    hs.nsupers = ns;
    hs.xsup.resize(ns + 1);
    for (int k = 0; k < ns; ++k) hs.xsup[k] = ufirst[k];
    hs.xsup[ns] = (int) n;
    // ---- index arrays in the reference formats ----
    hs.lidx_off.assign(ns + 1, 0); hs.uidx_off.assign(ns + 1, 0); hs.lval_off.assign(ns + 1, 0); hs.uval_off.assign(ns + 1, 0);
    // Two passes over the supernodes, both embarrassingly parallel (every supernode's arrays depend on its own row structure only):
    // sizes -> serial prefix sums -> fill.  Worker threads over contiguous supernode ranges (SLUAMD_SYMB_THREADS, default: the
    // hardware's, at most 16): at 200^3 this phase is half of the symbolic factorisation.
    auto parallel_for = [&](int count, const std::function<void(int, int)> &body) {      // dynamic chunks: the supernodes of the top separators carry most of the entries
        parallel_chunks(count, 64, [&](int64_t b, int64_t e) { body((int) b, (int) e); });
    };
    std::vector<int64_t> sz_lidx(ns), sz_lval(ns), sz_uidx(ns), sz_uval(ns);
    std::vector<double> fl(ns);
    parallel_for(ns, [&](int k0, int k1) {
        for (int k = k0; k < k1; ++k) {
            const int nsupc = hs.xsup[k + 1] - hs.xsup[k];
            const int64_t s0 = sy->srow_off[k], s1 = sy->srow_off[k + 1];
            int nblk = 0, ucols = 0; int64_t ulen = 0;
            for (int64_t e = s0; e < s1;) {
                const int gb = sy->supno[sy->srows[e]];
                int64_t f = e;
                while (f < s1 && sy->supno[sy->srows[f]] == gb) ++f;
                ++nblk; ulen += UB_DESCRIPTOR + (hs.xsup[gb + 1] - hs.xsup[gb]); ucols += (int) (f - e);
                e = f;
            }
            const int64_t r = s1 - s0;
            sz_lidx[k] = BC_HEADER + (int64_t) (nblk + 1) * LB_DESCRIPTOR + nsupc + r;
            sz_lval[k] = (int64_t) (nsupc + r) * nsupc;
            sz_uidx[k] = nblk ? BR_HEADER + ulen : 0;
            sz_uval[k] = (int64_t) ucols * nsupc;
            fl[k] = (2.0 / 3.0) * nsupc * (double) nsupc * nsupc + 2.0 * (double) nsupc * nsupc * r + 2.0 * (double) nsupc * r * r;
        }
    });
    double flops = 0;
    for (int k = 0; k < ns; ++k) {
        hs.lidx_off[k + 1] = hs.lidx_off[k] + sz_lidx[k]; hs.lval_off[k + 1] = hs.lval_off[k] + sz_lval[k];
        hs.uidx_off[k + 1] = hs.uidx_off[k] + sz_uidx[k]; hs.uval_off[k + 1] = hs.uval_off[k] + sz_uval[k];
        flops += fl[k];
    }
    sy->flops = flops;
    hs.nnzL = hs.lval_off[ns]; hs.nnzU = hs.uval_off[ns];
    hs.lidx.resize(hs.lidx_off[ns]); hs.uidx.resize(hs.uidx_off[ns]);
    parallel_for(ns, [&](int k0, int k1) {
        for (int k = k0; k < k1; ++k) {
            const int nsupc = hs.xsup[k + 1] - hs.xsup[k], klst = hs.xsup[k + 1];
            const int64_t s0 = sy->srow_off[k], s1 = sy->srow_off[k + 1];
            int *li = hs.lidx.data() + hs.lidx_off[k];
            int p = BC_HEADER, nblk = 1;
            li[1] = (int) (nsupc + (s1 - s0));
            li[p] = k; li[p + 1] = nsupc;
            for (int i = 0; i < nsupc; ++i) li[p + LB_DESCRIPTOR + i] = hs.xsup[k] + i;
            p += LB_DESCRIPTOR + nsupc;
            int *ui = (s1 > s0) ? hs.uidx.data() + hs.uidx_off[k] : nullptr;
            int q = BR_HEADER, nub = 0;
            for (int64_t e = s0; e < s1;) {
                const int gb = sy->supno[sy->srows[e]];
                int64_t f = e;
                while (f < s1 && sy->supno[sy->srows[f]] == gb) ++f;
                li[p] = gb; li[p + 1] = (int) (f - e);
                for (int64_t t = e; t < f; ++t) li[p + LB_DESCRIPTOR + (t - e)] = sy->srows[t];
                p += LB_DESCRIPTOR + (int) (f - e); ++nblk;
                const int nsj = hs.xsup[gb + 1] - hs.xsup[gb];
                ui[q] = gb; ui[q + 1] = (int) (f - e) * nsupc;
                for (int c = 0; c < nsj; ++c) ui[q + UB_DESCRIPTOR + c] = klst;            // empty segment
                for (int64_t t = e; t < f; ++t) ui[q + UB_DESCRIPTOR + (sy->srows[t] - hs.xsup[gb])] = hs.xsup[k];  // full segment
                q += UB_DESCRIPTOR + nsj; ++nub;
                e = f;
            }
            li[0] = nblk;
            if (ui) { ui[0] = nub; ui[1] = (int) (hs.uval_off[k + 1] - hs.uval_off[k]); ui[2] = q; }
        }
    });
    lap("index arrays");
    sy->perm_c_final = perm;
    hs.present.assign(ns, 1);
    *out = reinterpret_cast<sluamd_symb_t>(sy);
    return 0;
}

// ---- unsymmetric-pattern symbolic factorisation with the reference's supernode rules ----------------
// What symbfact() computes for the reference (SRC/prec-independent/symbfact.c:83-200): the EXACT structure of L and U of
// A1 = Pc A Pc^T under elimination without pivoting -- struct((L + U)(:, j)) = the rows reachable from struct(A1(:, j)) through the
// columns of L to the left of j -- L by columns and U as per-column segments inside supernodes (first nonzero row of the segment; the
// rest of the supernode's rows below it are nonzero because the diagonal block of L is dense), with the supernode partition of
//   relax_snode (symbfact.c:221-265): maximal etree subtrees walked up from a leaf while the parent has < relax descendants;
//   column_dfs's boundary test (:598-672): column j continues the supernode of j - 1 iff struct(L(:, j)) is a subset of what column
//     j - 1 marked, |struct(L(:, j))| = |struct(L(:, j - 1))| - 1 (T2_SUPER) and the supernode has fewer than maxsup columns.
// The algorithm below is our own statement of that reach: supernode-level traversal over sorted row lists, symmetric pruning
// (Eisenstat-Liu: once L(j, s) and U(s, j) are both nonzero the traversal of supernode s may stop at row j, a PREFIX of its sorted
// list), no depth-first bookkeeping (the order of a column's segments is not part of the store).  The result is the reference's
// stored structure for the same perm_c / relax / maxsup: xsup, the row sets of every L block, every Ufstnz entry -- pinned by
// tests/test_symbolic_parity.py against the structures recorded from the real symbfact (tests/golden/*.npz).
int sluamd_dsymbfact_unsym(sluamd_symb_t *out, int64_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                           const sluamd_int_t *perm_c, int32_t relax, int32_t maxsup, sluamd_int_t *perm_c_out)
{
    if (!out || n <= 0 || !rowptr || !colind) { set_error("bad symbfact arguments"); return SLUAMD_EINVAL; }
    if (maxsup < 1) maxsup = 256;
    if (maxsup > 512) maxsup = 512;  // MAX_SUPER_SIZE, superlu_defs.h:154
    if (relax < 1) relax = 1;
    // relax_snode (symbfact.c:221-265) can build relaxed supernodes of up to `relax` columns whatever maxsup says; the handle's refinement takes
    // supernodes of <= 512 columns: reject instead of producing a structure that cannot be factored
    if (relax > 512) { set_error("sluamd_dsymbfact_unsym: relax > 512 (MAX_SUPER_SIZE) is not supported"); return SLUAMD_EINVAL; }
    std::vector<int> perm, parent;
    Graph g;
    static const bool timing = getenv("SLUAMD_SYMB_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[sluamd_dsymbfact_unsym] %-28s %.3f s\n", what, t - t_prev); t_prev = t; } };
    if (int rc = order_and_etree(n, rowptr, colind, perm_c, perm, parent, g, perm_c_out, lap)) return rc;
    g = Graph();
    // columns of A1 (unsymmetric pattern, final labels)
    std::vector<int64_t> acol(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) acol[perm[colind[e]] + 1]++;
    for (int64_t j = 0; j < n; ++j) acol[j + 1] += acol[j];
    std::vector<int> arow(acol[n]);
    {
        std::vector<int64_t> fillp(acol.begin(), acol.end() - 1);
        for (int64_t i = 0; i < n; ++i) for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) arow[fillp[perm[colind[e]]]++] = perm[i];
    }
    // relaxed supernodes (relax_snode)
    std::vector<int> desc(n + 1, 0), relax_end(n, -1);
    for (int j = 0; j < n; ++j) if (parent[j] != -1) desc[parent[j]] += desc[j] + 1;
    for (int j = 0; j < n;) {
        int p = parent[j];
        const int f = j;
        while (p != -1 && desc[p] < relax) { j = p; p = parent[j]; }
        relax_end[f] = j;
        ++j;
        while (j < n && desc[j] != 0) ++j;
    }
    auto *sy = new Symb();
    HostStruct &hs = sy->hs;
    hs.n = n;
    sy->supno.assign(n, -1);
    std::vector<std::vector<int>> lrows;          // per supernode: sorted struct of its first column (relaxed: union), diagonal rows included
    std::vector<int> sfirst;                      // first column of every supernode
    std::vector<int> prune_end;                   // traversal bound inside lrows[s] (symmetric pruning)
    struct USeg { int col, fnz; };
    std::vector<std::vector<USeg>> useg;          // per supernode (block row): segments of the columns to its right, ascending column
    std::vector<int> mark(n, -1), segfnz, seglist, Lj, stack;
    int prev_size = 0;                            // |struct(L(:, j - 1))| as the boundary test counts it
    std::vector<uint8_t> relaxed;                 // per supernode: made by relax_snode
    auto new_snode = [&](int first) { sfirst.push_back(first); lrows.emplace_back(); prune_end.push_back(0); useg.emplace_back(); segfnz.push_back(-1); relaxed.push_back(0); return (int) sfirst.size() - 1; };
    for (int j = 0; j < n;) {
        if (relax_end[j] >= 0) {                  // a relaxed supernode [j, k]: union of the columns' structures (snode_dfs, :283-377)
            const int k = relax_end[j], s = new_snode(j);
            relaxed[s] = 1;
            std::vector<int> &R = lrows[s];
            for (int c = j; c <= k; ++c)
                for (int64_t e = acol[c]; e < acol[c + 1]; ++e) { const int r = arow[e]; if (mark[r] != k) { mark[r] = k; R.push_back(r); } }
            std::sort(R.begin(), R.end());
            if (R.empty() || R[0] < j) { delete sy; set_error("relaxed supernode with an entry above its diagonal block (perm_c is not an etree postorder)"); return SLUAMD_ESTRUCT; }
            for (int c = j; c <= k; ++c) {
                sy->supno[c] = s;
                if (!std::binary_search(R.begin(), R.end(), c)) { delete sy; set_error("zero diagonal in the symbolic factorisation"); return SLUAMD_ESTRUCT; }
            }
            prune_end[s] = (int) R.size();
            prev_size = (int) R.size();
            j = k + 1;
            continue;
        }
        // ---- one column: reach of struct(A1(:, j)) ----
        const int cur = sfirst.empty() ? -1 : (int) sfirst.size() - 1;      // supernode of column j - 1
        Lj.clear(); seglist.clear(); stack.clear();
        bool subset = true;
        auto visit = [&](int r) {
            const int old = mark[r];
            if (old == j) return;
            if (r >= j) {
                mark[r] = j;
                Lj.push_back(r);
                if (old != j - 1) subset = false;
            } else {
                const int s = sy->supno[r];
                // a relaxed supernode is entered through the copy of its WHOLE row list (diagonal rows included, snode_dfs :353-366), whose
                // already eliminated rows lower the first nonzero to the supernode's first column: its segments are always full height
                const int r0 = relaxed[s] ? sfirst[s] : r;
                if (segfnz[s] < 0) { segfnz[s] = r0; seglist.push_back(s); stack.push_back(s); }
                else if (r0 < segfnz[s]) segfnz[s] = r0;
            }
        };
        for (int64_t e = acol[j]; e < acol[j + 1]; ++e) {
            visit(arow[e]);
            while (!stack.empty()) {
                const int s = stack.back(); stack.pop_back();
                const std::vector<int> &R = lrows[s];
                // rows of s beyond its own columns: closed supernodes start behind their diagonal rows, the one still growing at row j
                const int last = (s == cur) ? j - 1 : ((s + 1 < (int) sfirst.size()) ? sfirst[s + 1] - 1 : j - 1);
                const int b = (int) (std::upper_bound(R.begin(), R.end(), last) - R.begin());
                const int e2 = (s == cur) ? (int) R.size() : prune_end[s];
                for (int q = b; q < e2; ++q) visit(R[q]);
            }
        }
        if (std::find(Lj.begin(), Lj.end(), j) == Lj.end()) { delete sy; set_error("zero diagonal in the symbolic factorisation"); return SLUAMD_ESTRUCT; }
        // ---- supernode boundary (column_dfs :598-672) ----
        bool cont = cur >= 0 && subset && (int) Lj.size() == prev_size - 1 && (j - sfirst[cur]) < maxsup;
        int sj;
        if (cont) sj = cur;
        else {
            sj = new_snode(j);
            std::sort(Lj.begin(), Lj.end());
            lrows[sj] = Lj;
            prune_end[sj] = (int) Lj.size();
        }
        sy->supno[j] = sj;
        prev_size = (int) Lj.size();
        // ---- U segments + symmetric pruning ----
        for (int s : seglist) {
            if (s != sj) {
                useg[s].push_back({j, segfnz[s]});
                const std::vector<int> &R = lrows[s];
                const int pe = prune_end[s];
                const int q = (int) (std::upper_bound(R.begin(), R.begin() + pe, j) - R.begin());
                if (q > 0 && R[q - 1] == j && q < pe) prune_end[s] = q;       // L(j, s) != 0 and U(s, j) != 0: nothing beyond row j needs s any more
            }
            segfnz[s] = -1;
        }
        ++j;
    }
    lap("reach + supernodes");
    const int ns = (int) sfirst.size();
    hs.nsupers = ns;
    hs.xsup.assign(sfirst.begin(), sfirst.end());
    hs.xsup.push_back((int) n);
    // ---- stores in the reference formats ----
    hs.lidx_off.assign(ns + 1, 0); hs.uidx_off.assign(ns + 1, 0); hs.lval_off.assign(ns + 1, 0); hs.uval_off.assign(ns + 1, 0);
    sy->srow_off.assign(1, 0);
    sy->ucol_off.assign(1, 0);
    sy->sn_parent.assign(ns, -1);
    double flops = 0;
    for (int k = 0; k < ns; ++k) {
        const int nsupc = hs.xsup[k + 1] - hs.xsup[k], klst = hs.xsup[k + 1];
        const std::vector<int> &R = lrows[k];
        if ((int) R.size() < nsupc || R[nsupc - 1] != klst - 1) { delete sy; set_error("supernode without its full diagonal block"); return SLUAMD_ESTRUCT; }
        const int pl = parent[klst - 1];
        sy->sn_parent[k] = pl >= 0 ? sy->supno[pl] : -1;
        // L: blocks by supernode of the row (ascending, rows sorted), diagonal block first
        std::vector<int> li(BC_HEADER, 0);
        int nblk = 0;
        for (size_t e = 0; e < R.size();) {
            const int gb = sy->supno[R[e]];
            size_t f = e;
            while (f < R.size() && sy->supno[R[f]] == gb) ++f;
            li.push_back(gb); li.push_back((int) (f - e));
            li.insert(li.end(), R.begin() + e, R.begin() + f);
            ++nblk; e = f;
        }
        li[0] = nblk; li[1] = (int) R.size();
        hs.lidx.insert(hs.lidx.end(), li.begin(), li.end());
        hs.lidx_off[k + 1] = (int64_t) hs.lidx.size();
        hs.lval_off[k + 1] = hs.lval_off[k] + (int64_t) R.size() * nsupc;
        sy->srows.insert(sy->srows.end(), R.begin() + nsupc, R.end());
        sy->srow_off.push_back((int64_t) sy->srows.size());
        // U: blocks by supernode of the column, a first-nonzero row per column (klst: empty)
        const std::vector<USeg> &U = useg[k];
        int64_t unz = 0; double segsq = 0, segsum = 0;
        if (!U.empty()) {
            std::vector<int> ui(BR_HEADER, 0);
            int nub = 0;
            for (size_t e = 0; e < U.size();) {
                const int jb = sy->supno[U[e].col], nsj = hs.xsup[jb + 1] - hs.xsup[jb];
                size_t f = e;
                const size_t h0 = ui.size();
                ui.push_back(jb); ui.push_back(0);
                ui.resize(ui.size() + nsj, klst);
                int bn = 0;
                while (f < U.size() && sy->supno[U[f].col] == jb) {
                    ui[h0 + UB_DESCRIPTOR + (U[f].col - hs.xsup[jb])] = U[f].fnz;
                    sy->ucol_col.push_back(U[f].col); sy->ucol_fnz.push_back(U[f].fnz); sy->ucol_voff.push_back(unz + bn);
                    const int seg = klst - U[f].fnz;
                    bn += seg; segsq += (double) seg * seg; segsum += seg;
                    ++f;
                }
                ui[h0 + 1] = bn; unz += bn; ++nub; e = f;
            }
            if (unz > 0x7fffffffLL || ui.size() > 0x7fffffffULL) {   // the reference's 32-bit header words (Ufstnz[1], [2]); one block row above 2^31 values needs _LONGINT
                set_error("sluamd_dsymbfact_unsym: a U block row holds more than 2^31 - 1 values"); delete sy; return SLUAMD_ESTRUCT;
            }
            ui[0] = nub; ui[1] = (int) unz; ui[2] = (int) ui.size();
            hs.uidx.insert(hs.uidx.end(), ui.begin(), ui.end());
        }
        hs.uidx_off[k + 1] = (int64_t) hs.uidx.size();
        hs.uval_off[k + 1] = hs.uval_off[k] + unz;
        sy->ucol_off.push_back((int64_t) sy->ucol_col.size());
        const double r = (double) R.size() - nsupc;
        flops += (2.0 / 3.0) * nsupc * (double) nsupc * nsupc + (double) nsupc * nsupc * r + segsq + 2.0 * r * segsum;
    }
    sy->flops = flops;
    hs.nnzL = hs.lval_off[ns]; hs.nnzU = hs.uval_off[ns];
    lap("index arrays");
    sy->perm_c_final = perm;
    hs.present.assign(ns, 1);
    *out = reinterpret_cast<sluamd_symb_t>(sy);
    return 0;
}

int sluamd_symb_info(sluamd_symb_t s, int32_t *nsupers, int64_t *nnzL, int64_t *nnzU, int64_t *lidx_len,
                     int64_t *uidx_len, double *flops)
{
    if (!s) return SLUAMD_EINVAL;
    Symb *sy = reinterpret_cast<Symb *>(s);
    if (nsupers) *nsupers = sy->hs.nsupers;
    if (nnzL) *nnzL = sy->hs.nnzL;
    if (nnzU) *nnzU = sy->hs.nnzU;
    if (lidx_len) *lidx_len = (int64_t) sy->hs.lidx.size();
    if (uidx_len) *uidx_len = (int64_t) sy->hs.uidx.size();
    if (flops) *flops = sy->flops;
    return 0;
}

int sluamd_symb_view(sluamd_symb_t s, sluamd_dLUview_t *v)
{
    if (!s || !v) return SLUAMD_EINVAL;
    Symb *sy = reinterpret_cast<Symb *>(s);
    HostStruct &hs = sy->hs;
    const int ns = hs.nsupers;
    if (sy->lval.size() != (size_t) hs.nnzL) sy->lval.assign(hs.nnzL, 0.0);
    if (sy->uval.size() != (size_t) hs.nnzU) sy->uval.assign(hs.nnzU, 0.0);
    sy->lptr.resize(ns); sy->uptr.resize(ns); sy->lvptr.resize(ns); sy->uvptr.resize(ns);
    for (int k = 0; k < ns; ++k) {
        sy->lptr[k] = hs.lidx.data() + hs.lidx_off[k];
        sy->lvptr[k] = sy->lval.data() + hs.lval_off[k];
        const bool hasu = hs.uidx_off[k + 1] > hs.uidx_off[k];
        sy->uptr[k] = hasu ? hs.uidx.data() + hs.uidx_off[k] : nullptr;
        sy->uvptr[k] = hasu ? sy->uval.data() + hs.uval_off[k] : nullptr;
    }
    std::memset(v, 0, sizeof(*v));
    v->n = hs.n; v->nsupers = ns; v->xsup = hs.xsup.data();
    v->nprow = v->npcol = v->npdep = 1;
    v->Lrowind_bc_ptr = sy->lptr.data(); v->Lnzval_bc_ptr = sy->lvptr.data();
    v->Ufstnz_br_ptr = sy->uptr.data(); v->Unzval_br_ptr = sy->uvptr.data();
    return 0;
}

int sluamd_ddistribute_host(sluamd_symb_t s, const sluamd_int_t *rowptr, const sluamd_int_t *colind,
                            const double *nzval, const sluamd_int_t *perm_c_final)
{
    if (!s) return SLUAMD_EINVAL;
    Symb *sy = reinterpret_cast<Symb *>(s);
    HostStruct &hs = sy->hs;
    sy->lval.assign(hs.nnzL, 0.0); sy->uval.assign(hs.nnzU, 0.0);
    std::vector<int64_t> pos; std::vector<uint8_t> isu;
    compute_scatter_positions(*sy, hs, hs.n, rowptr, colind, perm_c_final ? perm_c_final : sy->perm_c_final.data(), nullptr, pos, isu);
    for (size_t e = 0; e < pos.size(); ++e) {
        if (pos[e] < 0) { set_error("A entry outside the symbolic structure"); return SLUAMD_ESTRUCT; }
        (isu[e] ? sy->uval : sy->lval)[pos[e]] = nzval[e];
    }
    return 0;
}

/* copy the store out as flat arrays (any pointer may be NULL) -- test / baseline harness helper */
int sluamd_symb_export(sluamd_symb_t s, sluamd_int_t *xsup, int64_t *lidx_off, sluamd_int_t *lidx, int64_t *lval_off,
                       double *lval, int64_t *uidx_off, sluamd_int_t *uidx, int64_t *uval_off, double *uval)
{
    if (!s) return SLUAMD_EINVAL;
    Symb *sy = reinterpret_cast<Symb *>(s);
    const HostStruct &hs = sy->hs;
    const size_t ns1 = (size_t) hs.nsupers + 1;
    if (xsup) std::memcpy(xsup, hs.xsup.data(), ns1 * sizeof(int));
    if (lidx_off) std::memcpy(lidx_off, hs.lidx_off.data(), ns1 * sizeof(int64_t));
    if (lval_off) std::memcpy(lval_off, hs.lval_off.data(), ns1 * sizeof(int64_t));
    if (uidx_off) std::memcpy(uidx_off, hs.uidx_off.data(), ns1 * sizeof(int64_t));
    if (uval_off) std::memcpy(uval_off, hs.uval_off.data(), ns1 * sizeof(int64_t));
    if (lidx) std::memcpy(lidx, hs.lidx.data(), hs.lidx.size() * sizeof(int));
    if (uidx) std::memcpy(uidx, hs.uidx.data(), hs.uidx.size() * sizeof(int));
    if (lval) { if (sy->lval.size() != (size_t) hs.nnzL) { set_error("no host values: call sluamd_ddistribute_host first"); return SLUAMD_EINVAL; } std::memcpy(lval, sy->lval.data(), hs.nnzL * sizeof(double)); }
    if (uval) { if (sy->uval.size() != (size_t) hs.nnzU) { set_error("no host values: call sluamd_ddistribute_host first"); return SLUAMD_EINVAL; } std::memcpy(uval, sy->uval.data(), hs.nnzU * sizeof(double)); }
    return 0;
}

void sluamd_symb_free(sluamd_symb_t s) { delete reinterpret_cast<Symb *>(s); }

}  // extern "C"

namespace sluamd {

void compute_scatter_positions(const Symb &sy, const HostStruct &hs, int64_t n, const int *rowptr, const int *colind,
                               const int *perm, const uint8_t *owned, std::vector<int64_t> &pos, std::vector<uint8_t> &is_u)
{
    const int64_t nnz = rowptr[n];
    pos.resize(nnz); is_u.resize(nnz);
    auto rank_in = [&](int k, int row) -> int64_t {
        const int *b = sy.srows.data() + sy.srow_off[k], *e = sy.srows.data() + sy.srow_off[k + 1];
        return std::lower_bound(b, e, row) - b;
    };
    for (int64_t i = 0; i < n; ++i)
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int pi = perm[i], pj = perm[colind[e]];
            const int s = sy.supno[pj];
            const int nsupc = hs.xsup[s + 1] - hs.xsup[s];
            const int dest = (pi >= hs.xsup[s]) ? s : sy.supno[pi];
            if (owned && !owned[dest]) { pos[e] = -1; is_u[e] = 0; continue; }
            if (pi >= hs.xsup[s]) {
                const int nsupr = nsupc + (int) (sy.srow_off[s + 1] - sy.srow_off[s]);
                const int64_t lr = (pi < hs.xsup[s + 1]) ? (pi - hs.xsup[s]) : nsupc + rank_in(s, pi);
                pos[e] = hs.lval_off[s] + lr + (int64_t) (pj - hs.xsup[s]) * nsupr;
                is_u[e] = 0;
            } else if (!sy.ucol_off.empty()) {      // unsymmetric structure (sluamd_dsymbfact_unsym): skyline segment of column pj in block row r
                const int r = sy.supno[pi];
                const int *b = sy.ucol_col.data() + sy.ucol_off[r], *en = sy.ucol_col.data() + sy.ucol_off[r + 1];
                const int *f = std::lower_bound(b, en, pj);
                const int64_t q = sy.ucol_off[r] + (f - b);
                if (f == en || *f != pj || pi < sy.ucol_fnz[q]) { pos[e] = -1; is_u[e] = 1; continue; }     // outside the symbolic structure: reported by the caller
                pos[e] = hs.uval_off[r] + sy.ucol_voff[q] + (pi - sy.ucol_fnz[q]);
                is_u[e] = 1;
            } else {
                const int r = sy.supno[pi];
                const int nr = hs.xsup[r + 1] - hs.xsup[r];
                pos[e] = hs.uval_off[r] + rank_in(r, pj) * nr + (pi - hs.xsup[r]);
                is_u[e] = 1;
            }
        }
}


// Elimination-forest partition for a 1 x 1 x npdep grid (what getForests does for the reference,
// SRC/prec-independent/supernodalForest.c: greedy load balance "GD"; tree numbering = heap order as
// getGridTrees, supernodal_etree.c:840-851): tree 0 = common ancestors, trees 2t+1 / 2t+2 = the two halves below.
void partition_forests(const Symb &sy, int npdep, std::vector<int> &sn_tree)
{
    const HostStruct &hs = sy.hs;
    const int ns = hs.nsupers;
    int maxLvl = 1;
    while ((1 << (maxLvl - 1)) < npdep) ++maxLvl;
    sn_tree.assign(ns, 0);
    std::vector<int> parent(ns, -1);
    std::vector<double> w(ns, 0.0);
    std::vector<std::vector<int>> child(ns);
    for (int k = 0; k < ns; ++k) {
        const int64_t r = sy.srow_off[k + 1] - sy.srow_off[k];
        const double s = hs.xsup[k + 1] - hs.xsup[k];
        w[k] += (2.0 / 3.0) * s * s * s + 2.0 * s * s * r + 2.0 * s * (double) r * r;
        if (!sy.sn_parent.empty()) { if (sy.sn_parent[k] >= 0) { parent[k] = sy.sn_parent[k]; child[parent[k]].push_back(k); } }   // unsymmetric structure: the etree of A + A^T
        else if (r > 0) { parent[k] = sy.supno[sy.srows[sy.srow_off[k]]]; child[parent[k]].push_back(k); }
    }
    for (int k = 0; k < ns; ++k) if (parent[k] >= 0) w[parent[k]] += w[k];   // subtree weights (postorder: k < parent)
    struct Job { std::vector<int> roots; int tree, depth; };
    std::vector<Job> stack;
    Job top; top.tree = 0; top.depth = 0;
    for (int k = 0; k < ns; ++k) if (parent[k] < 0) top.roots.push_back(k);
    stack.push_back(top);
    std::vector<int> st;
    while (!stack.empty()) {
        Job j = stack.back(); stack.pop_back();
        if (j.depth == maxLvl - 1) {   // leaf tree: whole subtrees
            st = j.roots;
            while (!st.empty()) { int v = st.back(); st.pop_back(); sn_tree[v] = j.tree; for (int c : child[v]) st.push_back(c); }
            continue;
        }
        std::vector<int> R = j.roots;
        while (R.size() == 1 && !child[R[0]].empty()) { sn_tree[R[0]] = j.tree; R = child[R[0]]; }   // peel the common chain
        std::sort(R.begin(), R.end(), [&](int a, int b) { return w[a] > w[b]; });
        Job a, b; a.tree = 2 * j.tree + 1; b.tree = 2 * j.tree + 2; a.depth = b.depth = j.depth + 1;
        double wa = 0, wb = 0;
        for (int r : R) { if (wa <= wb) { a.roots.push_back(r); wa += w[r]; } else { b.roots.push_back(r); wb += w[r]; } }
        stack.push_back(a); stack.push_back(b);
    }
}

}  // namespace sluamd

extern "C" int sluamd_symb_partition(sluamd_symb_t s, int32_t npdep, int32_t *sn_tree)
{
    if (!s || !sn_tree || npdep < 1 || (npdep & (npdep - 1))) { sluamd::set_error("npdep must be a power of two"); return SLUAMD_EINVAL; }
    std::vector<int> t;
    sluamd::partition_forests(*reinterpret_cast<sluamd::Symb *>(s), npdep, t);
    std::copy(t.begin(), t.end(), sn_tree);
    return 0;
}

// Storage of the factors per rank of an nprow x npcol x npdep grid, from the symbolic structure alone (what
// sluamd_dCreateLUHandleFromSymbGrid will allocate for the own slots): L blocks live at (block row % nprow, k % npcol), U blocks at
// (k % nprow, block column % npcol); a supernode of a forest at depth d of the forest tree is stored by every layer below that
// forest -- factored on the first, zero-initialised replicas on the others (dinit3DLUstructForest, pd3dcomm.c:334-800) -- which is
// what the Z ancestor reduction sums (pd3dcomm.c:1046-1081).
extern "C" int sluamd_symb_grid_footprint(sluamd_symb_t sp, int32_t nprow, int32_t npcol, int32_t npdep, const int32_t *sn_tree,
                                          int64_t *values, int64_t *replicated, int64_t *index_entries)
{
    using namespace sluamd;
    if (!sp || nprow < 1 || npcol < 1 || npdep < 1 || (npdep & (npdep - 1)) || (npdep > 1 && !sn_tree) || !values) { set_error("bad footprint arguments"); return SLUAMD_EINVAL; }
    const Symb &sy = *reinterpret_cast<Symb *>(sp);
    const HostStruct &hs = sy.hs;
    const int P = nprow * npcol * npdep;
    int L = 1;
    while ((1 << (L - 1)) < npdep) ++L;
    std::fill(values, values + P, 0);
    if (replicated) std::fill(replicated, replicated + P, 0);
    if (index_entries) std::fill(index_entries, index_entries + P, 0);
    for (int k = 0; k < hs.nsupers; ++k) {
        const int f = npdep > 1 ? sn_tree[k] : 0;
        int d = 0;
        while ((1 << (d + 1)) - 1 <= f) ++d;                       // depth of forest f in the heap-ordered forest tree
        if (d >= L) { set_error("sn_tree holds a forest id outside the tree of this npdep"); return SLUAMD_EINVAL; }
        const int span = 1 << (L - 1 - d), z0 = (f - ((1 << d) - 1)) * span;     // layers z0 .. z0 + span - 1; z0 factors it
        const int ns = hs.xsup[k + 1] - hs.xsup[k];
        const int *li = hs.lidx.data() + hs.lidx_off[k];
        if (hs.lidx_off[k + 1] - hs.lidx_off[k] >= BC_HEADER) {
            int p = BC_HEADER;
            for (int b = 0; b < li[0]; ++b) {
                const int gid = li[p], nbrow = li[p + 1];
                for (int z = z0; z < z0 + span && z < npdep; ++z) {
                    const int w = (z * nprow + gid % nprow) * npcol + k % npcol;
                    values[w] += (int64_t) nbrow * ns;
                    if (replicated && z != z0) replicated[w] += (int64_t) nbrow * ns;
                    if (index_entries) index_entries[w] += LB_DESCRIPTOR + nbrow;
                }
                p += LB_DESCRIPTOR + nbrow;
            }
        }
        if (hs.uidx_off[k + 1] - hs.uidx_off[k] >= BR_HEADER) {
            const int *ui = hs.uidx.data() + hs.uidx_off[k];
            int p = BR_HEADER;
            for (int b = 0; b < ui[0]; ++b) {
                const int jb = ui[p], nnzb = ui[p + 1], nsj = hs.xsup[jb + 1] - hs.xsup[jb];
                for (int z = z0; z < z0 + span && z < npdep; ++z) {
                    const int w = (z * nprow + k % nprow) * npcol + jb % npcol;
                    values[w] += nnzb;
                    if (replicated && z != z0) replicated[w] += nnzb;
                    if (index_entries) index_entries[w] += UB_DESCRIPTOR + nsj;
                }
                p += UB_DESCRIPTOR + nsj;
            }
        }
    }
    return 0;
}

// ---- fill-reducing ordering for matrices without geometry (what get_perm_c / METIS do for the reference,
// SRC/prec-independent/get_perm_c.c, get_perm_c_parmetis.c).  Nested dissection of the pattern of A + A^T by BFS level structures
// (George): connected component -> pseudo-peripheral root (repeated BFS) -> the level that halves the component, thinned to the
// vertices that really touch the far side, is the separator (numbered last) -> recurse on both sides; components of at most `leaf`
// vertices are numbered as they are.  Our own algorithm, O(nnz log n); the elimination-tree postorder of sluamd_dsymbfact then
// makes the supernodes.
extern "C" int sluamd_order_nd(int64_t n, const sluamd_int_t *rowptr, const sluamd_int_t *colind, int32_t leaf, sluamd_int_t *perm_c)
{
    using namespace sluamd;
    if (n <= 0 || !rowptr || !colind || !perm_c) { set_error("bad ordering arguments"); return SLUAMD_EINVAL; }
    if (leaf < 1) leaf = 64;
    // symmetric adjacency without the diagonal
    std::vector<int64_t> off(n + 1, 0);
    for (int64_t i = 0; i < n; ++i)
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int j = colind[e];
            if (j < 0 || j >= n) { set_error("column index out of range"); return SLUAMD_EINVAL; }
            if (j != i) { off[i + 1]++; off[j + 1]++; }
        }
    for (int64_t i = 0; i < n; ++i) off[i + 1] += off[i];
    std::vector<int> adj((size_t) off[n]);
    {
        std::vector<int64_t> pos(off.begin(), off.end() - 1);
        for (int64_t i = 0; i < n; ++i)
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) { const int j = colind[e]; if (j != i) { adj[pos[i]++] = j; adj[pos[j]++] = (int) i; } }
    }   // (duplicate edges of a symmetric input are harmless to BFS)
    // Recursive bisection as independent jobs.  A job = a vertex set + the END of its label range: labels are handed out from the end (separators last), the set's
    // separator takes the top of the range, the far half the labels below it, the near half the rest -- exactly what the depth-first order of a single stack gives,
    // so the permutation does not depend on how many threads run the jobs (plan_threads(); sets of <= 4096 vertices stay on the thread that produced them).
    // Shared arrays are touched per vertex by the job that owns the vertex; a BFS reads `part` of foreign neighbours, whose value is never this job's id.
    std::vector<int> part(n, 0);            // id of the vertex set a vertex currently belongs to; -1 = numbered
    // `part` is read by concurrently running jobs (the BFS of one job looks at neighbours another job owns and may be writing): relaxed atomic accesses, so that the
    // benign race is a defined one (free on x86).  `level` is only ever touched after part[w] == id, i.e. by the owner.
    auto pget = [&](int v) { return __atomic_load_n(&part[v], __ATOMIC_RELAXED); };
    auto pset = [&](int v, int x) { __atomic_store_n(&part[v], x, __ATOMIC_RELAXED); };
    std::vector<int> level(n, -1);
    struct Job { std::vector<int> verts; int64_t end; };
    std::atomic<int> next_part{1};
    std::atomic<int64_t> numbered{0};
    auto process = [&](Job &job, std::vector<int> &queue, std::vector<Job> &out) {
        std::vector<int> &V = job.verts;
        if (V.empty()) return;
        const int id = next_part.fetch_add(1);
        auto number = [&](const std::vector<int> &vs, int64_t end) { int64_t nx = end; for (auto it = vs.rbegin(); it != vs.rend(); ++it) { perm_c[*it] = (int) --nx; pset(*it, -1); } numbered.fetch_add((int64_t) vs.size()); };
        // BFS inside the set `id` from `root`; fills queue (visit order) and level[]; returns the number of levels
        auto bfs = [&](int root) {
            queue.clear(); queue.push_back(root); level[root] = 0;
            int nl = 1;
            for (size_t h = 0; h < queue.size(); ++h) {
                const int v = queue[h];
                for (int64_t e = off[v]; e < off[v + 1]; ++e) {
                    const int w = adj[e];
                    if (pget(w) == id && level[w] < 0) { level[w] = level[v] + 1; nl = level[w] + 1; queue.push_back(w); }
                }
            }
            return nl;
        };
        for (int v : V) pset(v, id);
        if ((int) V.size() <= leaf) { number(V, job.end); return; }
        // disconnected set: label ALL its connected components in one sweep (one BFS each, every vertex visited once -- peeling one
        // component per iteration and rescanning the rest costs O(|V| * #components) on block-diagonal inputs); they are
        // dissected -- and numbered -- in discovery order
        int nl = bfs(V[0]);
        if (queue.size() < V.size()) {
            int64_t end = job.end;
            { Job c; c.verts = queue; c.end = end; end -= (int64_t) c.verts.size(); out.push_back(std::move(c)); }
            for (int v : V) if (level[v] < 0) { bfs(v); Job c; c.verts = queue; c.end = end; end -= (int64_t) c.verts.size(); out.push_back(std::move(c)); }
            for (int v : V) level[v] = -1;
            return;
        }
        // pseudo-peripheral root: restart from a vertex of the last level while the level structure gets deeper
        for (int it = 0; it < 4; ++it) {
            int far = queue.back(), best = (int) (off[far + 1] - off[far]);
            for (size_t q = queue.size(); q-- > 0 && level[queue[q]] == nl - 1;) { const int v = queue[q], dg = (int) (off[v + 1] - off[v]); if (dg < best) { best = dg; far = v; } }
            for (int v : queue) level[v] = -1;
            const int nl2 = bfs(far);
            if (nl2 <= nl) { nl = nl2; break; }
            nl = nl2;
        }
        if (nl < 3) { for (int v : queue) level[v] = -1; number(V, job.end); return; }     // (near-)clique: nothing to dissect
        // the level that halves the component; among the levels around it the smallest one
        std::vector<int64_t> cnt(nl, 0);
        for (int v : queue) cnt[level[v]]++;
        int64_t run = 0; int mid = 1;
        for (int l = 0; l < nl; ++l) { run += cnt[l]; if (2 * run >= (int64_t) V.size()) { mid = l; break; } }
        mid = std::max(1, std::min(nl - 2, mid));
        int m = mid;
        for (int l = std::max(1, mid - 1); l <= std::min(nl - 2, mid + 1); ++l) if (cnt[l] < cnt[m]) m = l;
        Job lo, hi; std::vector<int> sep;
        for (int v : queue) {
            if (level[v] < m) lo.verts.push_back(v);
            else if (level[v] > m) hi.verts.push_back(v);
            else {   // thinning: a separator-level vertex without a neighbour in level m + 1 belongs to the near side
                bool touches = false;
                for (int64_t e = off[v]; e < off[v + 1] && !touches; ++e) touches = pget(adj[e]) == id && level[adj[e]] == m + 1;
                if (touches) sep.push_back(v); else lo.verts.push_back(v);
            }
        }
        for (int v : queue) level[v] = -1;
        number(sep, job.end);
        hi.end = job.end - (int64_t) sep.size();
        lo.end = hi.end - (int64_t) hi.verts.size();
        out.push_back(std::move(hi)); out.push_back(std::move(lo));
    };
    {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<Job> shared;
        int active = 0;
        { Job all; all.verts.resize(n); std::iota(all.verts.begin(), all.verts.end(), 0); all.end = n; shared.push_back(std::move(all)); }
        const int nthr = std::max(1, plan_threads());
        auto worker = [&]() {
            std::vector<int> queue; queue.reserve(1024);
            std::vector<Job> local, out;
            for (;;) {
                Job job;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !shared.empty() || active == 0; });
                    if (shared.empty()) { cv.notify_all(); return; }
                    job = std::move(shared.back()); shared.pop_back(); ++active;
                }
                local.clear(); local.push_back(std::move(job));
                while (!local.empty()) {
                    Job j = std::move(local.back()); local.pop_back();
                    out.clear();
                    process(j, queue, out);
                    bool gave = false;
                    for (Job &c : out) {
                        if (nthr > 1 && c.verts.size() > 4096) { std::lock_guard<std::mutex> lk(mu); shared.push_back(std::move(c)); gave = true; }
                        else local.push_back(std::move(c));
                    }
                    if (gave) cv.notify_all();
                }
                { std::lock_guard<std::mutex> lk(mu); --active; }
                cv.notify_all();
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthr; ++t) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    }
    const int64_t next = n - numbered.load();
    if (next != 0) { set_error("ordering did not number every vertex"); return SLUAMD_ESTRUCT; }
    return 0;
}
