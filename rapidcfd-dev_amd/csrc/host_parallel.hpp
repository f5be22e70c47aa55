// host_parallel.hpp -- threads for the one-time host builds (tile layout, GAMG coarse addressing).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace mi {
// Host threads for the one-time layout build: plain std::thread workers pulling blocks of `grain` indices from an atomic
// counter (no OpenMP runtime to clash with the caller's).  Everything built under it is independent per index, so the
// layout does not depend on the number of threads (MI_HOST_THREADS; default: the hardware's, at most 64).
inline int host_threads()
{
    static int n = [] {
        const char* e = getenv("MI_HOST_THREADS");
        int v = (e && *e) ? atoi(e) : (int)std::thread::hardware_concurrency();
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    return n;
}
template <class F>   // fn(begin, end, worker)
void parallel_blocks(int64_t n, int64_t grain, F fn)
{
    const int nt = (int)std::min<int64_t>(host_threads(), (n + grain - 1) / grain);
    if (nt <= 1) { if (n > 0) fn((int64_t)0, n, 0); return; }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;
    auto work = [&](int w) { for (;;) { const int64_t b = next.fetch_add(grain); if (b >= n) break; fn(b, std::min(n, b + grain), w); } };
    for (int w = 1; w < nt; ++w) pool.emplace_back(work, w);
    work(0);
    for (auto& t : pool) t.join();
}

} // namespace mi
