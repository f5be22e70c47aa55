// host_parallel.hpp -- threads for the one-time host builds (tile layout, GAMG coarse addressing).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <memory>
#include <tuple>
#include <utility>
#include <vector>

#include "host_tables.hpp"

namespace mi {
// Host threads for the one-time layout build: plain std::thread workers pulling blocks of `grain` indices from an atomic
// counter (no OpenMP runtime to clash with the caller's).  Everything built under it is independent per index, so the
// layout does not depend on the number of threads (MI_HOST_THREADS; default: the hardware's, at most 64 -- divided by the
// number of ranks the launcher started on this node, so that eight ranks building their sub-domains at the same time do not
// put 8 x 64 threads on the host: LOCAL_WORLD_SIZE of torchrun, the local sizes of Open MPI / MVAPICH / Slurm).
inline int host_threads()
{
    static int n = [] {
        const char* e = getenv("MI_HOST_THREADS");
        if (e && *e) { const int v = atoi(e); return v < 1 ? 1 : (v > 64 ? 64 : v); }
        int v = (int)std::thread::hardware_concurrency();
        for (const char* name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MV2_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE"}) {
            const char* l = getenv(name);
            if (l && atoi(l) > 1) { v /= atoi(l); break; }
        }
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    return n;
}
inline int env_int_host(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
// The workers are created once and shared by everything that runs at the same time (the hierarchy build of round 3 lays out
// level l on other threads while level l + 1 is matched: with a team of threads per call the box ran 3 x 64 threads on 64
// cores, and a 64-thread spawn per pass cost more than the pass): a call publishes its block counter, works on it itself,
// and whichever workers are idle join in; it returns when the blocks are done and the last helper has left.
class HostPool {
public:
    struct Job {
        void (*run)(void*, int64_t, int64_t, int);
        void* ctx;
        int64_t n, grain;
        int maxHelpers;
        std::atomic<int64_t> next{0};
        std::atomic<int> helpers{0}, active{0};
    };
    static HostPool& get() { static HostPool* p = new HostPool(host_threads() - 1); return *p; }   // (never destroyed: the workers sleep until the process ends)
    static void work(Job& j, int w)
    {
        for (;;) { const int64_t b = j.next.fetch_add(j.grain, std::memory_order_relaxed); if (b >= j.n) break; j.run(j.ctx, b, std::min(j.n, b + j.grain), w); }
    }
    void run(Job& j)
    {
        { std::lock_guard<std::mutex> lk(mu_); jobs_.push_back(&j); }
        cv_.notify_all();
        work(j, 0);
        { std::lock_guard<std::mutex> lk(mu_); jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &j)); }
        for (int spins = 0; j.active.load(std::memory_order_acquire) != 0; ++spins) { if (spins < 4000) __builtin_ia32_pause(); else std::this_thread::yield(); }
    }
private:
    explicit HostPool(int nWorkers) { for (int i = 0; i < nWorkers; ++i) std::thread([this] { loop(); }).detach(); }
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            Job* j = nullptr;
            for (Job* q : jobs_) if (q->next.load(std::memory_order_relaxed) < q->n && q->helpers.load(std::memory_order_relaxed) < q->maxHelpers) { j = q; break; }
            if (!j) { cv_.wait(lk); continue; }
            const int w = 1 + j->helpers.fetch_add(1, std::memory_order_relaxed);
            j->active.fetch_add(1, std::memory_order_relaxed);
            lk.unlock();
            work(*j, w);
            j->active.fetch_sub(1, std::memory_order_release);   // (the caller may return now: j is not touched again)
            lk.lock();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Job*> jobs_;
};
template <class F>   // fn(begin, end, worker); worker < the number of threads that take part in THIS call
void parallel_blocks(int64_t n, int64_t grain, F fn)
{
    const int nt = (int)std::min<int64_t>(host_threads(), (n + grain - 1) / grain);
    if (nt <= 1) { if (n > 0) fn((int64_t)0, n, 0); return; }
    HostPool::Job j;
    j.run = [](void* c, int64_t b, int64_t e, int w) { (*static_cast<F*>(c))(b, e, w); };
    j.ctx = &fn; j.n = n; j.grain = grain; j.maxHelpers = nt - 1;
    HostPool::get().run(j);
}

// Returning hundreds of MB to the system (munmap of the big one-time tables) takes tens of milliseconds: the containers are moved
// into a heap object that a detached thread destroys, the caller goes on.
template <class... V>
void free_in_background(V&... v)
{
    auto* bag = new std::tuple<V...>(std::move(v)...);
    std::thread([bag] { delete bag; }).detach();
}

// ---- building blocks of the threaded host builds (round 4).  Every one of them returns what its sequential form returns,
// whatever the number of threads: counts are sums, bucket lists are sorted after the (unordered) atomic fill.
template <class F>   // fn(i) for i in [0, n)
void parallel_for(int64_t n, int64_t grain, F fn)
{
    parallel_blocks(n, grain, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; ++i) fn(i); });
}
inline int32_t atomic_add_i32(int32_t* p, int32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline void atomic_min_i32(int32_t* p, int32_t v)
{
    int32_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
// v[i] <- v[0] + ... + v[i] (inclusive, in place); two passes over blocks of 1 << 16
template <class T>
void parallel_inclusive_scan(T* v, int64_t n)
{
    const int64_t B = 1 << 16, nb = (n + B - 1) / B;
    if (nb <= 2 || host_threads() == 1) { for (int64_t i = 1; i < n; ++i) v[i] += v[i - 1]; return; }
    std::vector<T> tot((size_t)nb);
    parallel_for(nb, 1, [&](int64_t k) { T s = 0; const int64_t e = std::min(n, (k + 1) * B); for (int64_t i = k * B; i < e; ++i) s += v[i]; tot[(size_t)k] = s; });
    T run = 0;
    for (int64_t k = 0; k < nb; ++k) { const T t = tot[(size_t)k]; tot[(size_t)k] = run; run += t; }
    parallel_for(nb, 1, [&](int64_t k) { T s = tot[(size_t)k]; const int64_t e = std::min(n, (k + 1) * B); for (int64_t i = k * B; i < e; ++i) { s += v[i]; v[i] = s; } });
}
// Stable counting sort of the items 0 .. n-1 by key(i) in [0, nBuckets) (key < 0: item left out): start[b] .. start[b+1] lists
// the items of bucket b in ASCENDING item order -- what the sequential count / prefix / fill passes produce.  Threads fill
// the buckets through atomic cursors in whatever order they run, then every bucket is sorted (they are short: the faces of a
// cell, the children of a coarse cell).
// (List: Table<int32_t>, or IndexList whose resize() does not zero what the fill pass writes anyway)
template <class Key, class List>
void bucket_items(int64_t n, int32_t nBuckets, Key key, Table<int32_t>& start, List& list)
{
    start.assign((size_t)nBuckets + 1, 0);
    if (host_threads() == 1 || n < (1 << 16)) {
        for (int64_t i = 0; i < n; ++i) { const int32_t k = key(i); if (k >= 0) ++start[(size_t)k + 1]; }
        for (int32_t b = 0; b < nBuckets; ++b) start[(size_t)b + 1] += start[(size_t)b];
        list.resize((size_t)start[(size_t)nBuckets]);
        Table<int32_t> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < n; ++i) { const int32_t k = key(i); if (k >= 0) list[(size_t)fill[(size_t)k]++] = (int32_t)i; }
        return;
    }
    // few buckets (tiles): one histogram per chunk of items, every chunk then fills its own sub-ranges in item order -- stable
    // without atomics or sorting
    const int64_t nChunks = std::min<int64_t>(4 * (int64_t)host_threads(), (n + (1 << 16) - 1) >> 16);
    if ((int64_t)nBuckets * nChunks <= 2 * n) {
        const int64_t per = (n + nChunks - 1) / nChunks;
        Table<int32_t> hist((size_t)(nBuckets * nChunks), 0);
        parallel_for(nChunks, 1, [&](int64_t c) {
            int32_t* h = hist.data() + (size_t)c * (size_t)nBuckets;
            for (int64_t i = c * per, e = std::min(n, (c + 1) * per); i < e; ++i) { const int32_t k = key(i); if (k >= 0) ++h[k]; }
        });
        parallel_for(nBuckets, 1 << 10, [&](int64_t b) {   // per bucket: chunk offsets relative to the bucket's start, and its size
            int32_t run = 0;
            for (int64_t c = 0; c < nChunks; ++c) { int32_t& h = hist[(size_t)c * (size_t)nBuckets + (size_t)b]; const int32_t t = h; h = run; run += t; }
            start[(size_t)b + 1] = run;
        });
        parallel_inclusive_scan(start.data() + 1, (int64_t)nBuckets);
        list.resize((size_t)start[(size_t)nBuckets]);
        parallel_for(nChunks, 1, [&](int64_t c) {
            int32_t* h = hist.data() + (size_t)c * (size_t)nBuckets;
            for (int64_t i = c * per, e = std::min(n, (c + 1) * per); i < e; ++i) { const int32_t k = key(i); if (k >= 0) list[(size_t)(start[(size_t)k] + h[k]++)] = (int32_t)i; }
        });
        return;
    }
    int32_t* cnt = start.data() + 1;
    parallel_for(n, 1 << 16, [&](int64_t i) { const int32_t k = key(i); if (k >= 0) atomic_add_i32(cnt + k, 1); });
    parallel_inclusive_scan(cnt, (int64_t)nBuckets);
    list.resize((size_t)start[(size_t)nBuckets]);
    IndexList fill((size_t)nBuckets);
    parallel_for(nBuckets, 1 << 18, [&](int64_t b) { fill[(size_t)b] = start[(size_t)b]; });
    parallel_for(n, 1 << 16, [&](int64_t i) { const int32_t k = key(i); if (k >= 0) list[(size_t)atomic_add_i32(fill.data() + k, 1)] = (int32_t)i; });
    parallel_blocks(nBuckets, 1 << 14, [&](int64_t b0, int64_t b1, int) {
        for (int64_t b = b0; b < b1; ++b) {
            int32_t* lo = list.data() + start[(size_t)b]; int32_t* hi = list.data() + start[(size_t)b + 1];
            if (hi - lo > 1 && !std::is_sorted(lo, hi)) std::sort(lo, hi);
        }
    });
}

} // namespace mi
