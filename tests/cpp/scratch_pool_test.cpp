// CPU-side unit test of the scratch pool's bookkeeping (optrace_amd/csrc/ot_scratch.hpp) with malloc as the allocator.
// Built and run by tests/test_scratch_pool.py (g++ -pthread); prints "ok <checks>" or aborts with the failed condition.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <thread>
#include <vector>

#include "ot_scratch.hpp"

static std::atomic<long> g_live{0}, g_allocs{0}, g_frees{0}, g_syncs{0};
static std::atomic<size_t> g_fail_above{SIZE_MAX};  // allocations larger than this fail ("out of memory")
static std::mutex g_set_mutex;
static std::set<void*> g_blocks;

static void* t_alloc(size_t n) {
    if (n > g_fail_above.load()) return nullptr;
    void* p = std::malloc(n ? n : 1);
    std::lock_guard<std::mutex> l(g_set_mutex);
    g_blocks.insert(p);
    g_live++;
    g_allocs++;
    return p;
}
static void t_release(void* p) {
    std::lock_guard<std::mutex> l(g_set_mutex);
    if (!g_blocks.erase(p)) {
        std::fprintf(stderr, "release of an unknown block\n");
        std::abort();
    }
    std::free(p);
    g_live--;
    g_frees++;
}
static void t_sync() { g_syncs++; }

static int g_checks = 0;
#define CHECK(c)                                                            \
    do {                                                                    \
        g_checks++;                                                         \
        if (!(c)) {                                                         \
            std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); \
            std::abort();                                                   \
        }                                                                   \
    } while (0)

using namespace ot_scratch;

int main() {
    void* const S1 = (void*)0x10;
    void* const S2 = (void*)0x20;
    size_t kept;
    int nb, busy;
    {
        Pool pool({t_alloc, t_release, t_sync}, 1 << 20);
        // a block serves call after call; a larger request replaces it
        char* p1;
        {
            Lease a(pool, 0, 1, S1, 1000);
            CHECK(a && a.bytes() >= 1000);
            p1 = a.p();
            std::memset(p1, 7, 1000);
        }
        {
            Lease a(pool, 0, 1, S1, 900);
            CHECK(a.p() == p1);  // reused
        }
        {
            Lease a(pool, 0, 1, S1, 5000);
            CHECK(a && a.bytes() >= 5000);
            pool.stats(&kept, &nb, &busy);
            CHECK(nb == 1 && busy == 1);  // the small one was replaced, not kept beside
        }
        // other stream / purpose / device: their own blocks
        {
            Lease a(pool, 0, 1, S1, 100), b(pool, 0, 1, S2, 100), c(pool, 0, 2, S1, 100), d(pool, 1, 1, S1, 100);
            std::set<char*> ps{a.p(), b.p(), c.p(), d.p()};
            CHECK(ps.size() == 4);
        }
        // a leased block is never handed out twice: the same key again gets a second block
        {
            Lease a(pool, 0, 1, S1, 100);
            Lease b(pool, 0, 1, S1, 100);
            CHECK(a.p() != b.p());
            // trim frees the idle blocks only
            const long syncs = g_syncs;
            const size_t left = pool.trim();
            CHECK(g_syncs == syncs + 1);
            pool.stats(&kept, &nb, &busy);
            CHECK(nb == 2 && busy == 2 && kept == left && left == a.bytes() + b.bytes());
            std::memset(a.p(), 1, 100);  // still ours
            std::memset(b.p(), 2, 100);
            // a lease moved into a long-lived object (an open automatic image) survives the scope and a trim
            Lease keep = std::move(a);
            CHECK(!a && keep);
            pool.trim();
            pool.stats(&kept, &nb, &busy);
            CHECK(nb == 2 && busy == 2);
        }
        pool.stats(&kept, &nb, &busy);
        CHECK(busy == 0 && nb == 2);
        pool.trim();
        pool.stats(&kept, &nb, &busy);
        CHECK(nb == 0 && kept == 0 && g_live == 0);

        // the cap: idle blocks go, least recently used first
        pool.set_cap(10000);
        { Lease a(pool, 0, 1, S1, 4000); }  // kept 4500
        { Lease b(pool, 0, 2, S1, 4000); }  // kept 9000
        { Lease a(pool, 0, 1, S1, 10); }    // touches purpose 1: purpose 2 is now the older one
        {
            Lease c(pool, 0, 3, S1, 4000);  // 13500 > cap: purpose 2 goes
            pool.stats(&kept, &nb, &busy);
            CHECK(nb == 2 && kept <= 10000);
            Lease a(pool, 0, 1, S1, 10);
            CHECK(a.bytes() >= 4000);  // purpose 1 survived
        }
        // a block on lease is not evicted, whatever the cap
        pool.set_cap(1);
        {
            Lease a(pool, 0, 1, S1, 10);
            Lease big(pool, 0, 4, S1, 3000);
            CHECK(a && big);
            std::memset(a.p(), 3, 10);
        }
        pool.set_cap(1 << 20);
        pool.trim();

        // out of memory: everything idle is freed and the exact size is tried
        { Lease a(pool, 0, 1, S1, 50000); }
        g_fail_above = 60000;
        {
            Lease b(pool, 0, 2, S1, 56000);  // 56000 + 1/8 = 63000 fails, exact 56000 fits once the idle block is gone
            CHECK(b && b.bytes() == 56000);
            pool.stats(&kept, &nb, &busy);
            CHECK(nb == 1);
            Lease c(pool, 0, 3, S1, 70000);  // cannot fit at all
            CHECK(!c && c.p() == nullptr);
        }
        g_fail_above = SIZE_MAX;
        pool.trim();
        CHECK(g_live == 0);

        // many threads, few keys, trims in between: no block is ever shared while on lease, nothing is freed under a lease
        std::atomic<bool> bad{false};
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        for (int t = 0; t < 8; t++)
            th.emplace_back([&, t] {
                unsigned x = 1234567u * (t + 1);
                for (int i = 0; i < 4000; i++) {
                    x = x * 1664525u + 1013904223u;
                    const size_t n = 64 + (x >> 8) % 4000;
                    Lease l(pool, 0, (int)((x >> 4) % 3), (x & 1) ? S1 : S2, n);
                    if (!l) { bad = true; return; }
                    std::memset(l.p(), t + 1, n);  // (a double hand-out or a free under us shows as a mismatch below or in ASan)
                    for (size_t k = 0; k < n; k += 97)
                        if (l.p()[k] != (char)(t + 1)) bad = true;
                }
            });
        std::thread trimmer([&] {
            while (!stop) {
                pool.trim();
                std::this_thread::yield();
            }
        });
        for (auto& t : th) t.join();
        stop = true;
        trimmer.join();
        CHECK(!bad);
        pool.stats(&kept, &nb, &busy);
        CHECK(busy == 0);
        pool.trim();
        pool.stats(&kept, &nb, &busy);
        CHECK(nb == 0 && kept == 0 && g_live == 0 && g_allocs == g_frees);
    }
    std::printf("ok %d\n", g_checks);
    return 0;
}
