// Host-side containers of the one-time builds (tile layout, GAMG hierarchy): std::vector over allocators that place large blocks for the
// host's transparent huge pages.
#pragma once
#include <sys/mman.h>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <new>
#include <utility>
#include <vector>

namespace mi {
// Large tables (>= 4 MiB) are placed on 2 MiB boundaries and advised to use transparent huge pages (round 6): a 10 M-cell layout first-touches
// ~1 GB of fresh pages, a quarter of a million 4 KiB faults on the threads that fill the tables -- 0.4 s of the first mi_addr_create of a
// process on the GPU box's host (profiles/r06_startup_timing.md).  No effect where THP is off; the tables' contents are what they were.
inline void* host_table_alloc(size_t bytes)
{
    static const bool thp = [] { const char* e = std::getenv("MI_HOST_THP"); return !e || e[0] != '0'; }();   // MI_HOST_THP=0: plain malloc (A/B)
    if (thp && bytes >= ((size_t)4 << 20)) {
        void* p = nullptr;
        if (posix_memalign(&p, (size_t)2 << 20, bytes) == 0 && p) { (void)madvise(p, bytes, MADV_HUGEPAGE); return p; }
    }
    void* p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(host_table_alloc(n * sizeof(T))); }
    void deallocate(T* p, size_t) { std::free(p); }
    template <class U, class... A> void construct(U* p, A&&... a) { if (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...); }
};
// value-initialising twin of NoInitAlloc for the tables that rely on std::vector's zero fill: same placement of large blocks
template <class T>
struct TableAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef TableAlloc<U> other; };
    TableAlloc() = default;
    template <class U> TableAlloc(const TableAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(host_table_alloc(n * sizeof(T))); }
    void deallocate(T* p, size_t) { std::free(p); }
};
template <class T> using Table = std::vector<T, TableAlloc<T>>;
typedef std::vector<int32_t, NoInitAlloc<int32_t>> IndexList;
// Bump arena of the per-tile tables of build_tile_layout (round 6).  The tiles' outputs are ~10^5 small vectors filled by all host threads at
// once: out of malloc they grow every thread's arena page by page (mprotect + 4 KiB faults under one address-space lock -- 0.3 s of the
// FIRST layout of a process at 10 M cells, nothing later when the arenas are warm).  Here a worker carves them out of chunks of its own
// (large blocks: huge pages), nothing is returned piecewise, and the chunks are dropped together once the tables have been laid end to end.
struct BumpArena {
    char *cur = nullptr, *end = nullptr;
    size_t chunk = (size_t)1 << 20;
    std::vector<void*> chunks;
    BumpArena() = default;
    BumpArena(const BumpArena&) = delete;
    BumpArena& operator=(const BumpArena&) = delete;
    ~BumpArena() { for (void* p : chunks) std::free(p); }
    void* take(size_t bytes)
    {
        bytes = (bytes + 15) & ~(size_t)15;
        if (cur == nullptr || bytes > (size_t)(end - cur)) {
            const size_t sz = bytes > chunk ? bytes : chunk;
            void* p = host_table_alloc(sz);
            chunks.push_back(p);
            cur = static_cast<char*>(p); end = cur + sz;
        }
        void* r = cur;
        cur += bytes;
        return r;
    }
};
inline BumpArena*& tile_arena_of_this_thread() { static thread_local BumpArena* a = nullptr; return a; }
template <class T>
struct ArenaAlloc {
    typedef T value_type;
    ArenaAlloc() = default;
    template <class U> ArenaAlloc(const ArenaAlloc<U>&) {}
    T* allocate(size_t n) { BumpArena* a = tile_arena_of_this_thread(); if (!a) throw std::bad_alloc(); return static_cast<T*>(a->take(n * sizeof(T))); }
    void deallocate(T*, size_t) {}
    template <class U> bool operator==(const ArenaAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const ArenaAlloc<U>&) const { return false; }
};
template <class T> using ArenaVec = std::vector<T, ArenaAlloc<T>>;
} // namespace mi
