// host_match.hpp -- the greedy matching both one-time host builds run (pair agglomeration of the GAMG hierarchy,
// pairGAMGAgglomerate.C:204-313; heavy-edge clustering of the tile layout), by the host threads, with the result of the
// sequential loop.
//
// The sequential loop visits the vertices in order; a vertex that is still free picks, among its neighbours that are free at
// that moment, the one across the heaviest eligible edge (strict '>': the first maximum in adjacency order) and both leave the
// pool; a free vertex without such a neighbour leaves the pool alone.  Two facts make it a parallel algorithm:
//   * a vertex is never free any more once its turn has passed, so the candidates of v are LATER neighbours only, and v's own
//     state at its turn is known as soon as its EARLIER neighbours have had theirs;
//   * whether a later neighbour u is still free at v's turn is known as soon as every neighbour of u that comes before v has
//     had its turn (only they could have taken u).
// So: a vertex becomes ACTIVE when its last earlier neighbour is resolved (a counter per vertex); an active free vertex decides
// as soon as its best candidate is certain, otherwise it stays active for the next round.  Every decision is the one the
// sequential loop takes -- rounds, threads and the order inside a round do not enter it.  The vertex with the smallest turn
// among the unresolved ones can always decide, so the rounds end; a mesh whose numbering chains the decisions (a rod of cells)
// degenerates into rounds of a few vertices, which one thread then runs without any synchronisation.
#pragma once
#include "host_parallel.hpp"

#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

namespace mi {

// spinning barrier for a team of threads that meets a few hundred times within milliseconds
struct TeamBarrier {
    explicit TeamBarrier(int n) : nThreads(n) {}
    void wait() { wait([] {}); }
    template <class F>   // the last thread to arrive runs onLast() before the others are released
    void wait(F onLast)
    {
        const unsigned gen = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) == nThreads - 1) {
            onLast();
            arrived.store(0, std::memory_order_relaxed);
            generation.store(gen + 1, std::memory_order_release);
            return;
        }
        for (int spins = 0; generation.load(std::memory_order_acquire) == gen; ++spins) {
            if (spins < 2000) __builtin_ia32_pause(); else std::this_thread::yield();
        }
    }
    const int nThreads;
    std::atomic<int> arrived{0};
    std::atomic<unsigned> generation{0};
};

// G provides:  int32_t n;  bool forward;
//              int64_t begin(v), end(v)        adjacency range of v (both directions of every edge are listed)
//              int32_t other(v, e)             the vertex across entry e of v's list
//              bool    better(v, e, u, best)   is the edge eligible and heavier ('>') than `best` (a G::Weight, starts at G::none())
//              G::Weight weight(v, e)
// Result: mate[v] = partner of v, or v itself when v left the pool alone; proposer[v] = 1 for the vertex of a pair whose turn
// made the pair (the one visited first).
template <class G>
void greedy_match_parallel(const G& g, Table<int32_t>& mate, Table<uint8_t>& proposer)
{
    const int32_t n = g.n;
    const bool fwd = g.forward;
    auto before = [fwd](int32_t a, int32_t b) { return fwd ? a < b : a > b; };
    mate.assign((size_t)n, -1);
    proposer.assign((size_t)n, 0);
    Table<int32_t> pending((size_t)n);   // earlier neighbours that have not had their turn yet
    Table<uint8_t> done((size_t)n, 0);
    // work lists: the round's list is ONE array (threads take blocks of it through a cursor); what a round activates goes to the
    // activating thread's own list (no shared counter in the hot path) and is gathered into the array between the rounds
    const int nt = host_threads();
    Table<int32_t> cur((size_t)n);
    Table<Table<int32_t>> mine((size_t)nt);
    Table<int64_t> offset((size_t)nt + 1, 0);
    std::atomic<int64_t> cursor{0};
    int64_t nCur = 0;
    int32_t* M = mate.data();
    uint8_t* D = done.data();
    {
        std::atomic<int64_t> k0{0};
        parallel_blocks(n, 1 << 15, [&](int64_t b, int64_t e, int) {
            for (int32_t v = (int32_t)b; v < (int32_t)e; ++v) {
                int32_t k = 0;
                for (int64_t j = g.begin(v); j < g.end(v); ++j) if (before(g.other(v, j), v)) ++k;
                pending[(size_t)v] = k;
                if (k == 0) cur[(size_t)k0.fetch_add(1, std::memory_order_relaxed)] = v;
            }
        });
        nCur = k0.load();
    }
    auto resolve = [&](int32_t x, Table<int32_t>& out) {   // x has had its turn (or was taken): its later neighbours lose one pending vertex
        __atomic_store_n(D + x, (uint8_t)1, __ATOMIC_RELEASE);
        for (int64_t j = g.begin(x); j < g.end(x); ++j) {
            const int32_t y = g.other(x, j);
            if (before(x, y) && __atomic_fetch_sub(&pending[(size_t)y], 1, __ATOMIC_ACQ_REL) == 1) out.push_back(y);
        }
    };
    auto process = [&](int32_t v, Table<int32_t>& out) {
        if (__atomic_load_n(M + v, __ATOMIC_ACQUIRE) >= 0) return;   // taken before its turn: resolved by the vertex that took it
        typename G::Weight best = G::none();
        int32_t pick = -1;
        for (int64_t j = g.begin(v); j < g.end(v); ++j) {
            const int32_t u = g.other(v, j);
            if (!before(v, u) || !g.better(v, j, u, best)) continue;
            if (__atomic_load_n(M + u, __ATOMIC_ACQUIRE) >= 0) continue;   // taken (by a vertex before v: a later one waits for v)
            for (int64_t k = g.begin(u); k < g.end(u); ++k) {
                const int32_t x = g.other(u, k);
                if (x != v && before(x, v) && !__atomic_load_n(D + x, __ATOMIC_ACQUIRE)) { out.push_back(v); return; }   // x could still take u: not certain yet
            }
            if (__atomic_load_n(M + u, __ATOMIC_ACQUIRE) >= 0) continue;
            pick = u; best = g.weight(v, j);
        }
        if (pick >= 0) {
            __atomic_store_n(M + pick, v, __ATOMIC_RELEASE);
            __atomic_store_n(M + v, pick, __ATOMIC_RELEASE);
            proposer[(size_t)v] = 1;
            resolve(v, out); resolve(pick, out);
        } else {
            __atomic_store_n(M + v, v, __ATOMIC_RELEASE);
            resolve(v, out);
        }
    };
    const int64_t small = 16 * (int64_t)nt;   // rounds this short are run by one thread, back to back, without barriers
    TeamBarrier bar(nt);
    bool finished = false;
    std::atomic<int64_t> total{0};
    // between two rounds (run by the last thread to finish its copy): the next list is complete; short rounds are run here
    auto between = [&] {
        nCur = total.load(std::memory_order_relaxed);
        Table<int32_t> a, b;
        while (nCur > 0 && nCur <= small) {
            a.assign(cur.begin(), cur.begin() + nCur);
            while (!a.empty() && (int64_t)a.size() <= small) { b.clear(); for (int32_t v : a) process(v, b); a.swap(b); }
            nCur = (int64_t)a.size();
            std::copy(a.begin(), a.end(), cur.begin());
        }
        finished = nCur == 0;
        total.store(0, std::memory_order_relaxed);
        cursor.store(0, std::memory_order_relaxed);
    };
    total.store(nCur);
    between();
    auto team = [&](int tid) {
        Table<int32_t>& out = mine[(size_t)tid];
        while (!finished) {
            const int64_t m = nCur;
            for (;;) {
                const int64_t b = cursor.fetch_add(256, std::memory_order_relaxed);
                if (b >= m) break;
                for (int64_t i = b, e = std::min(m, b + 256); i < e; ++i) process(cur[(size_t)i], out);
            }
            bar.wait();                                   // every decision of the round is made: the list may be overwritten
            if (!out.empty()) {
                std::copy(out.begin(), out.end(), cur.begin() + total.fetch_add((int64_t)out.size(), std::memory_order_relaxed));
                out.clear();
            }
            bar.wait(between);                            // the next round's list is published
        }
    };
    Table<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(team, t);
    team(0);
    for (auto& t : pool) t.join();
}

} // namespace mi
