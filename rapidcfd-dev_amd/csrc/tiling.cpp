// tiling.cpp -- see tiling.hpp.  Host code, runs once per mesh (the reference
// also builds its addressing tables and its GAMG agglomeration once per mesh
// and caches them: lduAddressing.C:169-400, GAMGAgglomeration.C:132-182).
#include "tiling.hpp"
#include "host_parallel.hpp"
#include "host_match.hpp"

#include <algorithm>
#include <cstring>
#include <cmath>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <numeric>
#include <thread>
#include <utility>
#ifdef MI_TIMING
#include <chrono>
#include <cstdio>
#define MI_T(label) do { auto now__ = std::chrono::steady_clock::now(); fprintf(stderr, "[tiling] %-28s %.3f s\n", label, std::chrono::duration<double>(now__ - t__).count()); t__ = now__; } while (0)
#else
#define MI_T(label) do {} while (0)
#endif

namespace mi {

namespace {

// One level of the multilevel clustering graph (CSR, undirected, both directions stored).
struct Graph {
    int32_t n = 0;
    Table<int64_t> xadj;
    std::vector<int32_t, NoInitAlloc<int32_t>> adj;
    std::vector<int32_t, NoInitAlloc<int32_t>> ew;   // edge weight = number of mesh faces between the clusters
    Table<int32_t> vw;   // cells in cluster
    Table<int32_t> vinc; // face incidences of the cluster's cells (internal counted twice)
    Table<int32_t> vint; // faces internal to the cluster
};

struct ClusterGraph {   // one level of the clustering graph for greedy_match_parallel (host_match.hpp)
    typedef int32_t Weight;
    static int32_t none() { return 0; }
    int32_t n; bool forward;
    const Graph* g; int32_t cellCap, slotCap;
    int64_t begin(int32_t v) const { return g->xadj[(size_t)v]; }
    int64_t end(int32_t v) const { return g->xadj[(size_t)v + 1]; }
    int32_t other(int32_t, int64_t e) const { return g->adj[(size_t)e]; }
    bool better(int32_t v, int64_t e, int32_t u, int32_t best) const
    {
        if (g->vw[(size_t)v] + g->vw[(size_t)u] > cellCap) return false;
        if (g->vinc[(size_t)v] + g->vinc[(size_t)u] - (g->vint[(size_t)v] + g->vint[(size_t)u] + g->ew[(size_t)e]) > slotCap) return false;
        return g->ew[(size_t)e] > best;
    }
    int32_t weight(int32_t, int64_t e) const { return g->ew[(size_t)e]; }
};

// Heavy-edge matching with size caps; returns number of coarse vertices and cmap.
// Large levels: the same decisions taken by the host threads (host_match.hpp); coarse vertex ids are the ranks of the vertices
// whose turn made a pair or that stayed alone, in index order.
int32_t match_level(const Graph& g, int32_t cellCap, int32_t slotCap, Table<int32_t>& cmap, Table<int32_t>& match)
{
    const int32_t n = g.n;
    match.resize((size_t)n);
    cmap.resize((size_t)n);
    if (host_threads() > 1 && n >= (1 << 15) && env_int_host("MI_MATCH_PARALLEL", 0) != 0) {
        ClusterGraph cg{n, true, &g, cellCap, slotCap};
        Table<uint8_t> proposer;
        greedy_match_parallel(cg, match, proposer);
        Table<int32_t> rank((size_t)n);
        parallel_for(n, 1 << 18, [&](int64_t v) { rank[(size_t)v] = (proposer[(size_t)v] || match[(size_t)v] == (int32_t)v) ? 1 : 0; });
        parallel_inclusive_scan(rank.data(), (int64_t)n);
        parallel_for(n, 1 << 18, [&](int64_t v) {
            const int32_t lead = (proposer[(size_t)v] || match[(size_t)v] == (int32_t)v) ? (int32_t)v : match[(size_t)v];
            cmap[(size_t)v] = rank[(size_t)lead] - 1;
        });
        return n > 0 ? rank[(size_t)n - 1] : 0;
    }
    // The sequential loop.  A vertex whose turn has passed is matched (with a partner or with itself), so only LATER neighbours
    // can be free: the earlier part of an adjacency list is skipped without a look at its state.  (A variant with the per-vertex
    // data packed into 16-byte records and the coarse ids from a prefix count measured 20 % SLOWER on the GPU box's host:
    // profiles/r04_p_startup_timing.md.)
    std::fill(match.begin(), match.end(), -1);
    int32_t nc = 0;
    for (int32_t v = 0; v < n; ++v) {
        if (match[v] >= 0) continue;
        int32_t best = -1, bw = 0;
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
            const int32_t u = g.adj[e];
            if (u <= v || match[u] >= 0) continue;
            if (g.vw[v] + g.vw[u] > cellCap) continue;
            const int32_t slots = g.vinc[v] + g.vinc[u] - (g.vint[v] + g.vint[u] + g.ew[e]);
            if (slots > slotCap) continue;
            if (g.ew[e] > bw) { bw = g.ew[e]; best = u; }
        }
        if (best >= 0) { match[v] = best; match[best] = v; cmap[v] = cmap[best] = nc++; }
        else           { match[v] = v; cmap[v] = nc++; }
    }
    return nc;
}

void coarsen(const Graph& g, const Table<int32_t>& cmap, const Table<int32_t>& match, int32_t nc, Graph& c)
{
    c.n = nc;
    c.vw.resize(nc); c.vinc.resize(nc); c.vint.resize(nc);
    // members of each coarse vertex (1 or 2): the vertex that was visited first (the smaller one) and its partner
    Table<int32_t> first(nc, -1), second(nc, -1);
    parallel_for(g.n, 1 << 16, [&](int64_t v) {
        const int32_t u = match[(size_t)v];
        if (u < (int32_t)v) return;
        const int32_t cv = cmap[(size_t)v];
        first[(size_t)cv] = (int32_t)v;
        c.vw[(size_t)cv] = g.vw[(size_t)v]; c.vinc[(size_t)cv] = g.vinc[(size_t)v]; c.vint[(size_t)cv] = g.vint[(size_t)v];
        if (u != (int32_t)v) { second[(size_t)cv] = u; c.vw[(size_t)cv] += g.vw[(size_t)u]; c.vinc[(size_t)cv] += g.vinc[(size_t)u]; c.vint[(size_t)cv] += g.vint[(size_t)u]; }
    });
    // Row of a coarse vertex = the neighbours of its members, mapped, merged (weights added) and sorted by id (deterministic
    // tie-breaking in match_level).  Rows are independent: built twice under OpenMP (count, then fill), merged through a small
    // per-thread sort instead of a global scratch table, so the result does not depend on the number of threads.
#ifdef MI_TIMING
    auto t__ = std::chrono::steady_clock::now();
#endif
    c.xadj.assign((size_t)nc + 1, 0);
    Table<int32_t> vintAdd((size_t)nc, 0);
    auto build_row = [&](int32_t cv, Table<std::pair<int32_t, int32_t>>& tmp, int32_t& internal) {
        tmp.clear(); internal = 0;
        for (int k = 0; k < 2; ++k) {
            const int32_t v = k ? second[cv] : first[cv];
            if (v < 0) continue;
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                const int32_t cu = cmap[g.adj[e]];
                if (cu == cv) { if (k == 0) internal += g.ew[e]; continue; } // edge inside the pair
                tmp.push_back({cu, g.ew[e]});
            }
        }
        std::sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) { return a.first < b.first; });
        size_t w = 0;
        for (size_t i = 0; i < tmp.size(); ++i) {
            if (w > 0 && tmp[w - 1].first == tmp[i].first) tmp[w - 1].second += tmp[i].second;
            else tmp[w++] = tmp[i];
        }
        tmp.resize(w);
    };
    parallel_blocks(nc, 16384, [&](int64_t b, int64_t e, int) {
        Table<std::pair<int32_t, int32_t>> tmp;
        int32_t internal;
        for (int32_t cv = (int32_t)b; cv < (int32_t)e; ++cv) { build_row(cv, tmp, internal); c.xadj[(size_t)cv + 1] = (int64_t)tmp.size(); vintAdd[(size_t)cv] = internal; }
    });
    MI_T("    coarsen: count");
    parallel_for(nc, 1 << 18, [&](int64_t cv) { c.vint[(size_t)cv] += vintAdd[(size_t)cv]; });
    parallel_inclusive_scan(c.xadj.data() + 1, (int64_t)nc);
    c.adj.resize((size_t)c.xadj[nc]); c.ew.resize((size_t)c.xadj[nc]);
    MI_T("    coarsen: prefix+alloc");
    parallel_blocks(nc, 16384, [&](int64_t b, int64_t e, int) {
        Table<std::pair<int32_t, int32_t>> tmp;
        int32_t internal;
        for (int32_t cv = (int32_t)b; cv < (int32_t)e; ++cv) {
            build_row(cv, tmp, internal);
            int64_t at = c.xadj[cv];
            for (const auto& q : tmp) { c.adj[(size_t)at] = q.first; c.ew[(size_t)at] = q.second; ++at; }
        }
    });
}

// Reverse Cuthill-McKee ordering of the cell graph (new -> old): components started from their lowest-degree cell, neighbours
// appended in order of increasing degree (ties by cell index: deterministic).  Host, sequential, only for meshes whose
// numbering has no locality.
void cuthill_mckee(int32_t n, const Table<int32_t>& ownStart, const Table<int32_t>& neiStart, const IndexList& ownFaces,
                   const IndexList& neiFaces, const int32_t* lower, const int32_t* upper, Table<int32_t>& order)
{
    Table<int32_t> deg((size_t)n);
    int32_t maxDeg = 0;
    for (int32_t c = 0; c < n; ++c) { deg[c] = (ownStart[(size_t)c + 1] - ownStart[c]) + (neiStart[(size_t)c + 1] - neiStart[c]); maxDeg = std::max(maxDeg, deg[c]); }
    Table<int32_t> byDeg((size_t)n); // cells by (degree, index): counting sort
    {
        Table<int32_t> cnt((size_t)maxDeg + 2, 0);
        for (int32_t c = 0; c < n; ++c) cnt[(size_t)deg[c] + 1]++;
        for (int32_t d = 0; d <= maxDeg; ++d) cnt[(size_t)d + 1] += cnt[d];
        for (int32_t c = 0; c < n; ++c) byDeg[(size_t)cnt[deg[c]]++] = c;
    }
    Table<char> seen((size_t)n, 0);
    order.clear(); order.reserve((size_t)n);
    Table<int32_t> nb;
    for (int32_t s = 0; s < n; ++s) {
        const int32_t start = byDeg[s];
        if (seen[start]) continue;
        seen[start] = 1; order.push_back(start);
        for (size_t head = order.size() - 1; head < order.size(); ++head) {
            const int32_t v = order[head];
            nb.clear();
            for (int32_t j = ownStart[v]; j < ownStart[(size_t)v + 1]; ++j) { const int32_t u = upper[ownFaces[j]]; if (!seen[u]) { seen[u] = 1; nb.push_back(u); } }
            for (int32_t j = neiStart[v]; j < neiStart[(size_t)v + 1]; ++j) { const int32_t u = lower[neiFaces[j]]; if (!seen[u]) { seen[u] = 1; nb.push_back(u); } }
            std::sort(nb.begin(), nb.end(), [&](int32_t a, int32_t b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    std::reverse(order.begin(), order.end());
}
} // namespace

std::string build_tile_layout(int32_t nCells, int32_t nFaces, const int32_t* lower,
                              const int32_t* upper, int32_t nPatches,
                              const int32_t* patchSizes, const int32_t* const* patchFaceCells,
                              const TileParams& prm, TileLayout& L, const int32_t* const* patchNbrCells)
{
    if (nCells <= 0) return "n_cells must be positive";
    if (prm.slotCap > 32766) return "slotCap exceeds the 15-bit slot field";
#ifdef MI_TIMING
    auto t__ = std::chrono::steady_clock::now();
#endif
    std::atomic<bool> lowerUnsorted{false};   // upper-triangular order (lduAddressing) has the owners ascending: their face lists are ranges
    {
        std::atomic<bool> bad{false};
        parallel_blocks(nFaces, 1 << 18, [&](int64_t b, int64_t e, int) {
            bool any = false, unsorted = false;
            for (int64_t f = b; f < e; ++f) { any |= lower[f] < 0 || upper[f] >= nCells || lower[f] >= upper[f]; unsorted |= f > 0 && lower[f] < lower[f - 1]; }
            if (any) bad = true;
            if (unsorted) lowerUnsorted = true;
        });
        if (bad) return "addressing must satisfy 0 <= lowerAddr[f] < upperAddr[f] < nCells";
    }
    L = TileLayout();
    L.nCells = nCells; L.nFaces = nFaces; L.nPatches = nPatches;
    L.patchOffset.assign((size_t)nPatches + 1, 0);
    for (int32_t p = 0; p < nPatches; ++p) L.patchOffset[(size_t)p + 1] = L.patchOffset[p] + patchSizes[p];
    L.nExt = L.patchOffset[nPatches];

    // ---- per-cell face lists in the reference's row order -------------------
    // owner side: faces with lower==c ascending (ownerStartAddr); neighbour side:
    // faces with upper==c in losort order (stable sort by upper) -- lduAddressing.C:169-344
    // (threaded: buckets filled through atomic cursors, then sorted -- ascending face id inside a cell as the stable passes give)
    Table<int32_t> ownStart, neiStart;
    IndexList ownFaces, neiFaces, ownPos((size_t)nFaces); // ownPos: rank of f among its owner's faces
    if (lowerUnsorted) bucket_items(nFaces, nCells, [&](int64_t f) { return lower[f]; }, ownStart, ownFaces);
    else {
        ownStart.resize((size_t)nCells + 1); ownFaces.resize((size_t)nFaces);
        parallel_for(nFaces, 1 << 18, [&](int64_t f) {
            ownFaces[(size_t)f] = (int32_t)f;
            for (int32_t c = f > 0 ? lower[f - 1] + 1 : 0; c <= lower[f]; ++c) ownStart[(size_t)c] = (int32_t)f;   // (empty unless f opens a new owner)
        });
        for (int32_t c = nFaces > 0 ? lower[nFaces - 1] + 1 : 0; c <= nCells; ++c) ownStart[(size_t)c] = nFaces;
    }
    bucket_items(nFaces, nCells, [&](int64_t f) { return upper[f]; }, neiStart, neiFaces);
    parallel_for(nCells, 1 << 16, [&](int64_t c) { for (int32_t j = ownStart[(size_t)c]; j < ownStart[(size_t)c + 1]; ++j) ownPos[(size_t)ownFaces[(size_t)j]] = j - ownStart[(size_t)c]; });
    // patch faces per cell (patch order, then face order)
    Table<int32_t> pfStart((size_t)nCells + 1, 0), pfList((size_t)L.nExt);
    for (int32_t p = 0; p < nPatches; ++p)
        for (int32_t i = 0; i < patchSizes[p]; ++i) {
            const int32_t c = patchFaceCells[p][i];
            if (c < 0 || c >= nCells) return "patch faceCells out of range";
            pfStart[(size_t)c + 1]++;
        }
    for (int32_t c = 0; c < nCells; ++c) pfStart[(size_t)c + 1] += pfStart[c];
    {
        Table<int32_t> cp(pfStart.begin(), pfStart.end() - 1);
        for (int32_t p = 0; p < nPatches; ++p)
            for (int32_t i = 0; i < patchSizes[p]; ++i)
                pfList[(size_t)cp[patchFaceCells[p][i]]++] = L.patchOffset[p] + i;
    }

    // local coupled patches (cyclic): caller cell on the other side of every interface face, or -1 (ext region)
    Table<int32_t> ifaceLocalNbr((size_t)L.nExt, -1);
    if (patchNbrCells)
        for (int32_t p = 0; p < nPatches; ++p)
            if (patchNbrCells[p])
                for (int32_t i = 0; i < patchSizes[p]; ++i) {
                    const int32_t c = patchNbrCells[p][i];
                    if (c < 0 || c >= nCells) return "cyclic patch neighbour cells out of range";
                    ifaceLocalNbr[(size_t)L.patchOffset[p] + i] = c;
                }

    MI_T("face lists");
    // ---- multilevel heavy-edge clustering ------------------------------------
    Table<int32_t> part((size_t)nCells);
    std::iota(part.begin(), part.end(), 0);
    int32_t nClusters = nCells;
    Table<int32_t> cmOrder, cmRank; // Cuthill-McKee ordering (new -> old) and its inverse, when the clustering runs on it
    if (prm.givenPart) {
        // given partition: checked against the caps (cells; slots = face incidences + patch faces - faces inside the tile)
        const int32_t nP = prm.nGivenParts;
        if (nP <= 0) return "given partition: no tiles";
        Table<int32_t> cells((size_t)nP, 0), inc((size_t)nP, 0), internal((size_t)nP, 0);
        std::atomic<bool> bad{false};
        parallel_for(nCells, 1 << 16, [&](int64_t c) {
            const int32_t t = prm.givenPart[c];
            if (t < 0 || t >= nP) { bad = true; return; }
            part[(size_t)c] = t;
            atomic_add_i32(&cells[(size_t)t], 1);
            atomic_add_i32(&inc[(size_t)t], (ownStart[(size_t)c + 1] - ownStart[(size_t)c]) + (neiStart[(size_t)c + 1] - neiStart[(size_t)c]) + (pfStart[(size_t)c + 1] - pfStart[(size_t)c]));
        });
        if (bad) return "given partition: tile id out of range";
        parallel_for(nFaces, 1 << 16, [&](int64_t f) { const int32_t t = part[(size_t)lower[f]]; if (t == part[(size_t)upper[f]]) atomic_add_i32(&internal[(size_t)t], 1); });
        for (int32_t t = 0; t < nP; ++t) {
            if (cells[(size_t)t] == 0) return "given partition: empty tile";
            if (cells[(size_t)t] > prm.tileCells || cells[(size_t)t] > 65535 - 4096 || inc[(size_t)t] - internal[(size_t)t] > prm.slotCap) return "a given tile exceeds the tile caps (cells or coefficient slots)";
        }
        nClusters = nP;
    } else if (prm.keepOrder) {
        // ordered layout: tiles are ranges of the caller's numbering (given, or cut greedily under the caps)
        if (prm.givenTileStart) {
            if (prm.nGivenTiles <= 0 || prm.givenTileStart[0] != 0 || prm.givenTileStart[prm.nGivenTiles] != nCells) return "tile_cell_start must run from 0 to n_cells";
            for (int32_t t = 0; t < prm.nGivenTiles; ++t) {
                const int32_t a = prm.givenTileStart[t], b = prm.givenTileStart[(size_t)t + 1];
                if (b <= a) return "tile_cell_start must be strictly increasing";
                int64_t inc = 0, internal = 0;
                for (int32_t c = a; c < b; ++c) {
                    part[c] = t;
                    inc += (ownStart[(size_t)c + 1] - ownStart[c]) + (neiStart[(size_t)c + 1] - neiStart[c]) + (pfStart[(size_t)c + 1] - pfStart[c]);
                    for (int32_t j = ownStart[c]; j < ownStart[(size_t)c + 1]; ++j) if (upper[ownFaces[j]] < b) ++internal; // owner < neighbour, both in [a, b)
                }
                if (b - a > 65535 - 4096 || inc - internal > prm.slotCap) return "a given tile exceeds the tile caps (cells or coefficient slots)";
            }
            nClusters = prm.nGivenTiles;
        } else {
            int32_t t = 0, cells = 0; int64_t slots = 0;
            int32_t a = 0;
            for (int32_t c = 0; c < nCells; ++c) {
                int64_t add = (ownStart[(size_t)c + 1] - ownStart[c]) + (pfStart[(size_t)c + 1] - pfStart[c]);
                for (int32_t j = neiStart[c]; j < neiStart[(size_t)c + 1]; ++j) if (lower[neiFaces[j]] < a) ++add;   // faces to cells of earlier tiles are cut: one more slot here
                if (add > prm.slotCap) return "a single cell has more faces than a tile can hold";
                if (cells > 0 && (cells + 1 > prm.tileCells || slots + add > prm.slotCap)) {
                    ++t; cells = 0; slots = 0; a = c;
                    add = (ownStart[(size_t)c + 1] - ownStart[c]) + (pfStart[(size_t)c + 1] - pfStart[c]) + (neiStart[(size_t)c + 1] - neiStart[c]);
                }
                part[c] = t; ++cells; slots += add;
            }
            nClusters = t + 1;
        }
    } else {
        // numbering without locality?  then the clustering (which visits vertices in index order) runs on a Cuthill-McKee
        // ordering: vertex v of the graph = cell cmOrder[v]
        bool reorder = prm.reorder > 0;
        if (prm.reorder < 0 && nCells > 2 * prm.tileCells && nFaces > 0) {
            double sum = 0;
            for (int32_t f = 0; f < nFaces; ++f) sum += (double)(upper[f] - lower[f]);
            reorder = sum / nFaces > 4.0 * std::pow((double)nCells, 2.0 / 3.0);
        }
        if (reorder) {
            cuthill_mckee(nCells, ownStart, neiStart, ownFaces, neiFaces, lower, upper, cmOrder);
            cmRank.resize((size_t)nCells);
            for (int32_t v = 0; v < nCells; ++v) cmRank[(size_t)cmOrder[v]] = v;
            MI_T("Cuthill-McKee");
        }
        Graph g;
        g.n = nCells;
        g.xadj.assign((size_t)nCells + 1, 0);
        for (int32_t v = 0; v < nCells; ++v) {
            const int32_t c = reorder ? cmOrder[v] : v;
            g.xadj[(size_t)v + 1] = g.xadj[v] + (ownStart[(size_t)c + 1] - ownStart[c]) + (neiStart[(size_t)c + 1] - neiStart[c]);
        }
        g.adj.resize((size_t)2 * nFaces); g.ew.resize((size_t)2 * nFaces);
        g.vw.assign(nCells, 1); g.vint.assign(nCells, 0); g.vinc.resize(nCells);
        std::atomic<bool> tooMany{false};
        parallel_blocks(nCells, 65536, [&](int64_t b, int64_t e, int) {
            for (int32_t v = (int32_t)b; v < (int32_t)e; ++v) {
                const int32_t c = reorder ? cmOrder[v] : v;
                int64_t k = g.xadj[v];
                for (int32_t j = ownStart[c]; j < ownStart[(size_t)c + 1]; ++j) { g.ew[(size_t)k] = 1; g.adj[(size_t)k++] = upper[ownFaces[j]]; }
                for (int32_t j = neiStart[c]; j < neiStart[(size_t)c + 1]; ++j) { g.ew[(size_t)k] = 1; g.adj[(size_t)k++] = lower[neiFaces[j]]; }
                if (reorder) { // as the row of a mesh numbered that way would read: larger-numbered neighbours ascending, then the smaller ones
                    int32_t* a0 = g.adj.data() + g.xadj[v]; int32_t* a1 = g.adj.data() + g.xadj[(size_t)v + 1];
                    for (int32_t* q = a0; q < a1; ++q) *q = cmRank[(size_t)*q];
                    int32_t* mid = std::partition(a0, a1, [&](int32_t u) { return u > v; });
                    std::sort(a0, mid); std::sort(mid, a1);
                }
                g.vinc[v] = (int32_t)(g.xadj[(size_t)v + 1] - g.xadj[v]) + (pfStart[(size_t)c + 1] - pfStart[c]);
                if (g.vinc[v] > prm.slotCap) tooMany = true;
            }
        });
        if (tooMany) return "a single cell has more faces than a tile can hold";
        if (reorder) for (int32_t c = 0; c < nCells; ++c) part[c] = cmRank[(size_t)c];
        // multi-edges (two faces between the same cell pair) are legal in LDU addressing; merge them
        Table<int32_t> cmap, partner;
        for (int level = 0; level < 64; ++level) {
            const int32_t nc = match_level(g, prm.tileCells, prm.slotCap, cmap, partner);
            MI_T("  match");
            if (nc == g.n) break;
            parallel_for(nCells, 1 << 18, [&](int64_t c) { part[(size_t)c] = cmap[(size_t)part[(size_t)c]]; });
            MI_T("  part update");
            Graph cg;
            coarsen(g, cmap, partner, nc, cg);
            MI_T("  coarsen");
            g = std::move(cg);
            nClusters = nc;
        }
    }

    MI_T("clustering");
    // ---- order tiles by their smallest caller cell, cells inside a tile ascending ----
    const int32_t nT = nClusters;
    L.nTiles = nT;
    const bool byRank = !cmRank.empty(); // tiles and the cells inside them follow the ordering the clustering ran on
    {
        Table<int32_t> tileMin((size_t)nT, INT32_MAX);   // smallest vertex of every tile (a minimum: any order of visits)
        parallel_for(nCells, 1 << 16, [&](int64_t v) { atomic_min_i32(&tileMin[(size_t)part[(size_t)(byRank ? cmOrder[(size_t)v] : (int32_t)v)]], (int32_t)v); });
        Table<int32_t> order((size_t)nT);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return tileMin[a] < tileMin[b]; });
        Table<int32_t> rank((size_t)nT);
        for (int32_t i = 0; i < nT; ++i) rank[order[i]] = i;
        parallel_for(nCells, 1 << 18, [&](int64_t c) { part[(size_t)c] = rank[(size_t)part[(size_t)c]]; });
    }
    L.e2c.resize(nCells); L.c2e.resize(nCells);
    {
        // vertices of every tile, ascending (bucket_items: the stable count / fill passes, threaded); engine cell e = position
        IndexList members;
        bucket_items(nCells, nT, [&](int64_t v) { return part[(size_t)(byRank ? cmOrder[(size_t)v] : (int32_t)v)]; }, L.tileCellStart, members);
        parallel_for(nCells, 1 << 16, [&](int64_t e) { const int32_t v = members[(size_t)e]; const int32_t c = byRank ? cmOrder[(size_t)v] : v; L.e2c[(size_t)e] = c; L.c2e[(size_t)c] = (int32_t)e; });
    }

    MI_T("renumbering");
    // ---- slots, halos, row entries --------------------------------------------
    // Slot order of a tile: for every local row its owner faces (ascending face id) as one run, then for every halo
    // cell the faces it owns whose neighbour is a local row, then the interface slots, then the zero slot.  Every
    // interface face has a halo entry of its own (remote: its ext cell; cyclic: a private reference to the partner
    // cell), so that slotBase[other] names its slot.
    L.tileSlotStart.assign((size_t)nT + 1, 0);
    L.tileIfaceSlot0.assign((size_t)nT, 0);
    L.tileHaloStart.assign((size_t)nT + 1, 0);
    L.tileSliceStart.assign((size_t)nT + 1, 0);
    L.sliceEntryStart.clear(); L.sliceEntryStart.push_back(0);
    L.sliceEntryStart16.clear(); L.sliceEntryStart16.push_back(0);
    L.tileSbStart.assign(1, 0);
    L.slotFace.clear(); L.haloCell.clear(); L.entries.clear(); L.entries16.clear(); L.slotBase.clear();
    L.compact = prm.compact;
    L.extSlot.assign((size_t)L.nExt, -1);
    L.faceSlot.assign((size_t)nFaces, -1);
    L.patchFaceCellsE.resize((size_t)L.nExt);
    for (int32_t p = 0; p < nPatches; ++p)
        for (int32_t i = 0; i < patchSizes[p]; ++i)
            L.patchFaceCellsE[(size_t)L.patchOffset[p] + i] = L.c2e[patchFaceCells[p][i]];

    struct RowEnt { int32_t other, slot, k; bool rule, lowerSide; };
    struct Pending { int32_t ent, halo, id; };
    // Every tile is built on its own (OpenMP over tiles: nothing a tile writes depends on another tile) into local tables;
    // a sequential pass then lays them end to end.  A cut face gets its faceSlot from the tile of its OWNER cell (both
    // tiles hold the coefficient), so the result does not depend on the order tiles are visited in.
    // (their storage: the building thread's BumpArena, csrc/host_tables.hpp -- reserved to the sizes known before a tile is filled)
    struct TileOut {
        ArenaVec<int32_t> slotFace, haloCell, sliceEntryStart, sliceEntryStart16;   // slice starts relative to the tile's first entry
        ArenaVec<uint32_t> entries, entries16;
        ArenaVec<uint16_t> slotBase;
        ArenaVec<std::pair<int32_t, int32_t>> extSlot, faceSlot;                    // (ext / face id, local slot)
        int32_t ifaceSlot0 = 0, nSlots = 0, nHalo = 0, nc = 0;
        bool boundary = false, fits16 = true;
        std::string err;
    };
    Table<TileOut> outs((size_t)nT);
    const bool wantCompact = prm.compact;
    Table<std::unique_ptr<BumpArena>> arenas((size_t)host_threads());
    for (auto& a : arenas) { a.reset(new BumpArena()); a->chunk = std::min<size_t>((size_t)16 << 20, std::max<size_t>((size_t)64 << 10, (size_t)96 * (size_t)nCells / (size_t)host_threads())); }
    parallel_blocks(nT, 8, [&](int64_t tBegin, int64_t tEnd, int worker) {
    struct UseArena { explicit UseArena(BumpArena* a) { tile_arena_of_this_thread() = a; } ~UseArena() { tile_arena_of_this_thread() = nullptr; } } useArena(arenas[(size_t)worker].get());
    // per-block scratch: halo lookup by engine cell (open addressing, rebuilt per tile)
    Table<RowEnt> rowEnt;          // entries of the rows of the current tile, row-major
    Table<int32_t> rowEntStart, sbLocal, haloCnt, sbHalo;
    Table<Pending> cutList, ifaceList;
    Table<int32_t> hKey, hVal;      // hash table
    auto hash_reset = [&](size_t want) { size_t cap = 64; while (cap < 2 * want) cap <<= 1; hKey.assign(cap, -1); hVal.assign(cap, 0); };
    for (int32_t t = (int32_t)tBegin; t < (int32_t)tEnd; ++t) {
        TileOut& O = outs[(size_t)t];
        const int32_t cs = L.tileCellStart[t], ce = L.tileCellStart[(size_t)t + 1], nc = ce - cs;
        O.nc = nc;
        int32_t nHalo = 0;
        bool boundary = false, fits16 = true;
        rowEnt.clear(); rowEntStart.assign(1, 0);
        cutList.clear(); ifaceList.clear(); haloCnt.clear();

        sbLocal.assign((size_t)nc + 1, 0);
        int64_t incid = 0;
        for (int32_t e = cs; e < ce; ++e) {
            const int32_t c = L.e2c[e];
            sbLocal[(size_t)(e - cs) + 1] = sbLocal[e - cs] + (ownStart[(size_t)c + 1] - ownStart[c]);
            incid += (ownStart[(size_t)c + 1] - ownStart[c]) + (neiStart[(size_t)c + 1] - neiStart[c]) + (pfStart[(size_t)c + 1] - pfStart[c]);
        }
        const int32_t nLocal = sbLocal[nc];
        {   // sizes known now: the slots are at most one per incidence (+ the zero slot, + the pad), the entries exactly the slices' widths
            O.slotFace.reserve((size_t)incid + 3); O.faceSlot.reserve((size_t)nLocal); O.haloCell.reserve((size_t)incid / 8 + 32);
            size_t nEnt = 0, nEnt16 = 0, nPf = 0;
            const int32_t nSl0 = (nc + 63) / 64;
            for (int32_t sl = 0; sl < nSl0; ++sl) {
                int32_t width = 0;
                for (int32_t e = cs + sl * 64; e < std::min(ce, cs + sl * 64 + 64); ++e) {
                    const int32_t c = L.e2c[e];
                    width = std::max(width, (ownStart[(size_t)c + 1] - ownStart[c]) + (neiStart[(size_t)c + 1] - neiStart[c]) + (pfStart[(size_t)c + 1] - pfStart[c]));
                    nPf += (size_t)(pfStart[(size_t)c + 1] - pfStart[c]);
                }
                nEnt += (size_t)width * 64; nEnt16 += (size_t)((width + 1) / 2) * 64;
            }
            O.entries.reserve(nEnt); O.extSlot.reserve(nPf);
            if (wantCompact) O.entries16.reserve(nEnt16);
            O.sliceEntryStart.reserve((size_t)nSl0 + 1); O.sliceEntryStart16.reserve((size_t)nSl0 + 1);
        }
        hash_reset((size_t)incid + 8);
        const size_t hMask = hKey.size() - 1;
        auto halo_of = [&](int32_t engineCell) -> int32_t {
            size_t h = ((size_t)(uint32_t)engineCell * 2654435761u) & hMask;
            while (hKey[h] != -1 && hKey[h] != engineCell) h = (h + 1) & hMask;
            if (hKey[h] == -1) { hKey[h] = engineCell; hVal[h] = nHalo++; O.haloCell.push_back(engineCell); haloCnt.push_back(0); }
            return nc + hVal[h];
        };
        auto place = [&](int32_t slot, int32_t what) {
            if (O.slotFace.size() <= (size_t)slot) O.slotFace.resize((size_t)slot + 1, -1);
            O.slotFace[(size_t)slot] = what;
        };
        for (int32_t e = cs; e < ce; ++e) {
            const int32_t c = L.e2c[e], i = e - cs;
            // owner side, ascending face id: row uses upper[f], other = upper cell
            for (int32_t j = ownStart[c]; j < ownStart[(size_t)c + 1]; ++j) {
                const int32_t f = ownFaces[j], o = upper[f];
                const int32_t slot = sbLocal[i] + (j - ownStart[c]);
                place(slot, f);
                int32_t other;
                if (part[o] == t) other = L.c2e[o] - cs;
                else other = halo_of(L.c2e[o]);
                O.faceSlot.push_back({f, slot});       // the owner's tile names the face's slot
                rowEnt.push_back({other, slot, 0, false, false});
            }
            // neighbour side, losort order: row uses lower[f], other = lower cell (the owner of the face)
            for (int32_t j = neiStart[c]; j < neiStart[(size_t)c + 1]; ++j) {
                const int32_t f = neiFaces[j], o = lower[f];
                if (part[o] == t) {
                    const int32_t ol = L.c2e[o] - cs, k = ownPos[f];
                    rowEnt.push_back({ol, sbLocal[ol] + k, k, true, true});
                } else {
                    const int32_t h = halo_of(L.c2e[o]) - nc, k = haloCnt[h]++;
                    cutList.push_back({(int32_t)rowEnt.size(), h, f});
                    rowEnt.push_back({nc + h, -1, k, true, true});
                }
            }
            // coupled interfaces in patch order; their slots follow all face slots of the tile so that a kernel
            // can tell them apart with one compare: slot >= tileIfaceSlot0[t]
            for (int32_t j = pfStart[c]; j < pfStart[(size_t)c + 1]; ++j) {
                const int32_t x = pfList[j];
                const int32_t nb = ifaceLocalNbr[x];
                int32_t h;
                if (nb < 0) { h = halo_of(nCells + x) - nc; boundary = true; }          // remote: value arrives in the ext region
                else { h = nHalo++; O.haloCell.push_back(L.c2e[nb]); haloCnt.push_back(0); } // cyclic partner (any tile): private halo entry
                ifaceList.push_back({(int32_t)rowEnt.size(), h, x});
                rowEnt.push_back({nc + h, -1, 0, true, false});
            }
            rowEntStart.push_back((int32_t)rowEnt.size());
        }
        sbHalo.assign((size_t)nHalo + 1, 0);
        for (int32_t h = 0; h < nHalo; ++h) sbHalo[(size_t)h + 1] = sbHalo[h] + haloCnt[h];
        const int32_t nFaceSlots = nLocal + sbHalo[nHalo];
        for (int32_t h = 0; h <= nHalo; ++h) sbHalo[h] += nLocal;
        for (const Pending& q : cutList) {
            RowEnt& re = rowEnt[(size_t)q.ent];
            re.slot = sbHalo[q.halo] + re.k;
            place(re.slot, q.id);
        }
        O.ifaceSlot0 = nFaceSlots;
        int32_t nSlots = nFaceSlots;
        for (const Pending& q : ifaceList) {
            const int32_t slot = nSlots++;
            rowEnt[(size_t)q.ent].slot = slot;
            place(slot, -(2 + q.id));
            O.extSlot.push_back({q.id, slot});
            sbHalo[q.halo] = slot;
        }
        if (nSlots > 32766 || nc + nHalo > 65535) { O.err = "tile exceeds the entry field widths"; continue; }
        if (nc + nHalo + 1 > 4096) fits16 = false;
        for (const RowEnt& re : rowEnt) if (re.rule && re.k > 7) { fits16 = false; break; }
        // zero slot + pad the segment to an even length (16-byte aligned double2 loads)
        place(nSlots, -1); // the zero slot, local index nSlots
        if ((O.slotFace.size() & 1u) != 0) O.slotFace.push_back(-1);
        // slot bases: local rows, halo cells, the pad cell (-> zero slot)
        if (wantCompact) {
            O.slotBase.reserve((size_t)nc + (size_t)nHalo + 2);
            for (int32_t i = 0; i < nc; ++i) O.slotBase.push_back((uint16_t)sbLocal[i]);
            for (int32_t h = 0; h < nHalo; ++h) O.slotBase.push_back((uint16_t)sbHalo[h]);
            O.slotBase.push_back((uint16_t)nSlots);
            if ((O.slotBase.size() & 1u) != 0) O.slotBase.push_back(0);
        }
        // slices of 64 rows, column-major, padded with {zero slot, other 0} / {pad cell}
        const uint32_t padEnt = ((uint32_t)nSlots << 16);
        const uint32_t pad16 = (uint32_t)((nc + nHalo) & 0xFFF) | 0x8000u;
        const int32_t nSl = (nc + 63) / 64;
        O.sliceEntryStart.assign(1, 0); O.sliceEntryStart16.assign(1, 0);
        for (int32_t s = 0; s < nSl; ++s) {
            const int32_t r0 = s * 64, r1 = std::min(nc, r0 + 64);
            int32_t width = 0;
            for (int32_t r = r0; r < r1; ++r) width = std::max(width, rowEntStart[(size_t)r + 1] - rowEntStart[r]);
            const size_t base = O.entries.size();
            O.entries.resize(base + (size_t)width * 64, padEnt);
            const int32_t width2 = (width + 1) / 2;
            const size_t base16 = O.entries16.size();
            if (wantCompact && fits16) O.entries16.resize(base16 + (size_t)width2 * 64, pad16 | (pad16 << 16));
            for (int32_t r = r0; r < r1; ++r) {
                const int32_t len = rowEntStart[(size_t)r + 1] - rowEntStart[r];
                for (int32_t j = 0; j < len; ++j) {
                    const RowEnt& re = rowEnt[(size_t)rowEntStart[r] + j];
                    O.entries[base + (size_t)j * 64 + (r - r0)] = (uint32_t)re.other | ((uint32_t)re.slot << 16) | (re.lowerSide ? 0x80000000u : 0u);
                    if (wantCompact && fits16) {
                        const uint32_t e16 = (uint32_t)(re.other & 0xFFF) | ((uint32_t)(re.k & 7) << 12) | (re.rule ? 0x8000u : 0u);
                        uint32_t& w = O.entries16[base16 + (size_t)(j >> 1) * 64 + (r - r0)];
                        w = (j & 1) ? ((w & 0x0000FFFFu) | (e16 << 16)) : ((w & 0xFFFF0000u) | e16);
                    }
                }
            }
            O.sliceEntryStart.push_back((int32_t)O.entries.size());
            O.sliceEntryStart16.push_back((int32_t)O.entries16.size());
        }
        O.nSlots = nSlots; O.nHalo = nHalo; O.boundary = boundary; O.fits16 = fits16;
    }
    });
    MI_T("  tiles (threads)");
    // ---- lay the tiles end to end ----------------------------------------------------------------------------------
    for (int32_t t = 0; t < nT; ++t) {
        if (!outs[(size_t)t].err.empty()) return outs[(size_t)t].err;
        if (!outs[(size_t)t].fits16) L.compact = false;
    }
    {
        size_t nSlotTot = 0, nEntTot = 0, nEnt16Tot = 0, nHaloTot = 0, nSbTot = 0, nSlTot = 0;
        for (const TileOut& O : outs) { nSlotTot += O.slotFace.size(); nEntTot += O.entries.size(); nEnt16Tot += O.entries16.size(); nHaloTot += O.haloCell.size(); nSbTot += O.slotBase.size(); nSlTot += O.sliceEntryStart.size() - 1; }
        if (nEntTot > (size_t)INT32_MAX - 4096 || nSlotTot > (size_t)INT32_MAX - 4096) return "mesh too large for 32-bit layout offsets";
        L.slotFace.resize(nSlotTot); L.entries.resize(nEntTot); L.haloCell.resize(nHaloTot);
        if (L.compact) { L.entries16.resize(nEnt16Tot); L.slotBase.resize(nSbTot); }
        (void)nSlTot;
    }
    {
        // offsets of every tile (sequential: nT additions), then the copies and the face / ext slot scatters by the host threads
        Table<size_t> sAt((size_t)nT + 1, 0), eAt((size_t)nT + 1, 0), e16At((size_t)nT + 1, 0), hAt((size_t)nT + 1, 0), sbAt((size_t)nT + 1, 0), slAt((size_t)nT + 1, 0);
        for (int32_t t = 0; t < nT; ++t) {
            const TileOut& O = outs[(size_t)t];
            sAt[(size_t)t + 1] = sAt[(size_t)t] + O.slotFace.size(); eAt[(size_t)t + 1] = eAt[(size_t)t] + O.entries.size(); hAt[(size_t)t + 1] = hAt[(size_t)t] + O.haloCell.size();
            e16At[(size_t)t + 1] = e16At[(size_t)t] + (L.compact ? O.entries16.size() : 0); sbAt[(size_t)t + 1] = sbAt[(size_t)t] + (L.compact ? O.slotBase.size() : 0);
            slAt[(size_t)t + 1] = slAt[(size_t)t] + (O.sliceEntryStart.size() - 1);
            L.tileIfaceSlot0[t] = O.ifaceSlot0;
            L.tileSlotStart[(size_t)t + 1] = (int32_t)sAt[(size_t)t + 1];
            L.tileHaloStart[(size_t)t + 1] = (int32_t)hAt[(size_t)t + 1];
            L.tileSbStart.push_back((int32_t)(sbAt[(size_t)t + 1] / 2));
            L.tileSliceStart[(size_t)t + 1] = (int32_t)slAt[(size_t)t + 1];
            L.maxCells = std::max(L.maxCells, O.nc);
            L.maxSlots = std::max(L.maxSlots, O.nSlots + 2);
            L.maxHalo = std::max(L.maxHalo, O.nHalo);
            (O.boundary ? L.boundaryTiles : L.interiorTiles).push_back(t);
        }
        L.sliceEntryStart.resize(slAt[(size_t)nT] + 1); L.sliceEntryStart16.resize(slAt[(size_t)nT] + 1);
        L.sliceEntryStart[0] = 0; L.sliceEntryStart16[0] = 0;
        parallel_blocks(nT, 4, [&](int64_t t0, int64_t t1, int) {
            for (int64_t t = t0; t < t1; ++t) {
                TileOut& O = outs[(size_t)t];
                std::copy(O.slotFace.begin(), O.slotFace.end(), L.slotFace.begin() + (std::ptrdiff_t)sAt[(size_t)t]);
                std::copy(O.entries.begin(), O.entries.end(), L.entries.begin() + (std::ptrdiff_t)eAt[(size_t)t]);
                std::copy(O.haloCell.begin(), O.haloCell.end(), L.haloCell.begin() + (std::ptrdiff_t)hAt[(size_t)t]);
                if (L.compact) {
                    std::copy(O.entries16.begin(), O.entries16.end(), L.entries16.begin() + (std::ptrdiff_t)e16At[(size_t)t]);
                    std::copy(O.slotBase.begin(), O.slotBase.end(), L.slotBase.begin() + (std::ptrdiff_t)sbAt[(size_t)t]);
                }
                for (size_t k = 1; k < O.sliceEntryStart.size(); ++k) {
                    L.sliceEntryStart[slAt[(size_t)t] + k] = (int32_t)(eAt[(size_t)t] + (size_t)O.sliceEntryStart[k]);
                    L.sliceEntryStart16[slAt[(size_t)t] + k] = (int32_t)(e16At[(size_t)t] + (size_t)(L.compact ? O.sliceEntryStart16[k] : 0));
                }
                for (const auto& q : O.extSlot) L.extSlot[(size_t)q.first] = (int32_t)(sAt[(size_t)t] + (size_t)q.second);
                for (const auto& q : O.faceSlot) L.faceSlot[(size_t)q.first] = (int32_t)(sAt[(size_t)t] + (size_t)q.second);
            }
        });
    }
    free_in_background(outs, arenas, ownStart, neiStart, ownFaces, neiFaces, ownPos, pfStart, pfList, part, ifaceLocalNbr);
    MI_T("slots / halos / entries");
    if (!L.compact) { Table<uint32_t>().swap(L.entries16); Table<int32_t>().swap(L.sliceEntryStart16); }
    L.nSlices = L.tileSliceStart[nT];
    L.totalSlots = (int64_t)L.slotFace.size();
    return std::string();
}

std::string inherit_tiles(int32_t nFine, const int32_t* restrictMap, const int32_t* fineTileOfCell, int32_t nFineTiles,
                          int32_t nCoarse, int32_t nCoarseFaces, const int32_t* cLower, const int32_t* cUpper,
                          int32_t nPatches, const int32_t* patchSizes, const int32_t* const* patchFaceCells,
                          int32_t cellCap, int32_t slotCap, Table<int32_t>& part, int32_t& nParts)
{
    if (nFine <= 0 || nCoarse <= 0 || nFineTiles <= 0) return "inherit_tiles: bad argument";
    const int32_t nT = nFineTiles;
    // a coarse cell goes where its first (smallest) child is
    Table<int32_t> firstChild((size_t)nCoarse, INT32_MAX);
    parallel_for(nFine, 1 << 16, [&](int64_t fc) { atomic_min_i32(&firstChild[(size_t)restrictMap[fc]], (int32_t)fc); });
    part.resize((size_t)nCoarse);
    Table<int32_t> vw((size_t)nT, 0), vinc((size_t)nT, 0), vint((size_t)nT, 0);
    std::atomic<bool> bad{false};
    parallel_for(nCoarse, 1 << 16, [&](int64_t c) {
        if (firstChild[(size_t)c] == INT32_MAX) { bad = true; return; }
        const int32_t t = fineTileOfCell[firstChild[(size_t)c]];
        if (t < 0 || t >= nT) { bad = true; return; }
        part[(size_t)c] = t;
        atomic_add_i32(&vw[(size_t)t], 1);
    });
    if (bad) return "inherit_tiles: a coarse cell without children, or a fine tile id out of range";
    for (int32_t p = 0; p < nPatches; ++p) for (int32_t i = 0; i < patchSizes[p]; ++i) ++vinc[(size_t)part[(size_t)patchFaceCells[p][i]]];
    // tile graph: faces between two tiles counted per unordered pair (thread-local tables, merged)
    struct EdgeMap { Table<uint64_t> key; Table<int32_t> val; size_t mask = 0, used = 0;
        void init(size_t cap) { size_t c = 64; while (c < cap) c <<= 1; key.assign(c, ~0ull); val.assign(c, 0); mask = c - 1; used = 0; }
        void grow() { EdgeMap g; g.init(key.size() * 2); for (size_t i = 0; i < key.size(); ++i) if (key[i] != ~0ull) g.add(key[i], val[i]); *this = std::move(g); }
        void add(uint64_t k, int32_t v) { if (2 * (used + 1) > key.size()) grow(); size_t h = (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20) & mask; while (key[h] != ~0ull && key[h] != k) h = (h + 1) & mask; if (key[h] == ~0ull) { key[h] = k; ++used; } val[h] += v; } };
    EdgeMap all;
    all.init(4096);
    std::mutex mu;
    parallel_blocks(nCoarseFaces, 1 << 18, [&](int64_t b, int64_t e, int) {
        EdgeMap local;
        local.init(1024);
        for (int64_t f = b; f < e; ++f) {
            const int32_t a = part[(size_t)cLower[f]], c2 = part[(size_t)cUpper[f]];
            atomic_add_i32(&vinc[(size_t)a], 1); atomic_add_i32(&vinc[(size_t)c2], 1);
            if (a == c2) atomic_add_i32(&vint[(size_t)a], 1);
            else local.add(((uint64_t)(uint32_t)std::min(a, c2) << 32) | (uint32_t)std::max(a, c2), 1);
        }
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < local.key.size(); ++i) if (local.key[i] != ~0ull) all.add(local.key[i], local.val[i]);
    });
    // sorted edge list (tile pairs ascending): deterministic whatever order the blocks were merged in
    Table<std::pair<uint64_t, int32_t>> edges;
    edges.reserve(all.used);
    for (size_t i = 0; i < all.key.size(); ++i) if (all.key[i] != ~0ull) edges.push_back({all.key[i], all.val[i]});
    std::sort(edges.begin(), edges.end());
    // rounds of pairwise merges: tiles in index order take the free neighbour across the heaviest common boundary that fits
    Table<int32_t> rep((size_t)nT);
    std::iota(rep.begin(), rep.end(), 0);
    for (int round = 0; round < 4; ++round) {
        Table<std::pair<uint64_t, int32_t>> cur;
        cur.reserve(edges.size());
        for (const auto& q : edges) {
            const int32_t a = rep[(size_t)(q.first >> 32)], b = rep[(size_t)(uint32_t)q.first];
            if (a != b) cur.push_back({((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b), q.second});
        }
        std::sort(cur.begin(), cur.end());
        size_t w = 0;
        for (size_t i = 0; i < cur.size(); ++i) { if (w > 0 && cur[w - 1].first == cur[i].first) cur[w - 1].second += cur[i].second; else cur[w++] = cur[i]; }
        cur.resize(w);
        Table<int32_t> start((size_t)nT + 1, 0);
        for (const auto& q : cur) { ++start[(size_t)(q.first >> 32) + 1]; ++start[(size_t)(uint32_t)q.first + 1]; }
        for (int32_t t = 0; t < nT; ++t) start[(size_t)t + 1] += start[(size_t)t];
        Table<int32_t> nbr((size_t)start[(size_t)nT]), wgt((size_t)start[(size_t)nT]), fill(start.begin(), start.end() - 1);
        for (const auto& q : cur) {   // (ascending pairs: every tile's list comes out ascending by neighbour)
            const int32_t a = (int32_t)(q.first >> 32), b = (int32_t)(uint32_t)q.first;
            nbr[(size_t)fill[(size_t)a]] = b; wgt[(size_t)fill[(size_t)a]++] = q.second;
            nbr[(size_t)fill[(size_t)b]] = a; wgt[(size_t)fill[(size_t)b]++] = q.second;
        }
        Table<char> used((size_t)nT, 0);
        Table<int32_t> into((size_t)nT, -1);
        int merged = 0;
        for (int32_t t = 0; t < nT; ++t) {
            if (rep[(size_t)t] != t || used[(size_t)t] || vw[(size_t)t] == 0) continue;
            int32_t best = -1, bw = 0;
            for (int32_t j = start[(size_t)t]; j < start[(size_t)t + 1]; ++j) {
                const int32_t u = nbr[(size_t)j];
                if (used[(size_t)u] || vw[(size_t)t] + vw[(size_t)u] > cellCap) continue;
                if (vinc[(size_t)t] + vinc[(size_t)u] - (vint[(size_t)t] + vint[(size_t)u] + wgt[(size_t)j]) > slotCap) continue;
                if (wgt[(size_t)j] > bw) { bw = wgt[(size_t)j]; best = u; }
            }
            if (best < 0) continue;
            used[(size_t)t] = used[(size_t)best] = 1; ++merged;
            vw[(size_t)t] += vw[(size_t)best]; vinc[(size_t)t] += vinc[(size_t)best]; vint[(size_t)t] += vint[(size_t)best] + bw; vw[(size_t)best] = 0;
            into[(size_t)best] = t;
        }
        if (!merged) break;
        for (int32_t k = 0; k < nT; ++k) { const int32_t r = rep[(size_t)k]; if (into[(size_t)r] >= 0) rep[(size_t)k] = into[(size_t)r]; }
    }
    // compact ids (tiles that hold cells), caps
    Table<int32_t> id((size_t)nT, -1);
    nParts = 0;
    for (int32_t t = 0; t < nT; ++t) if (rep[(size_t)t] == t && vw[(size_t)t] > 0) {
        if (vw[(size_t)t] > cellCap || vinc[(size_t)t] - vint[(size_t)t] > slotCap) return "inherit_tiles: an inherited tile exceeds the caps";
        id[(size_t)t] = nParts++;
    }
    parallel_for(nCoarse, 1 << 18, [&](int64_t c) { part[(size_t)c] = id[(size_t)rep[(size_t)part[(size_t)c]]]; });
    return std::string();
}

} // namespace mi
