// Scratch of the binning paths (hit records, chunk indices, slabs: up to ~25 B per ray and image) -- the bookkeeping.
//
// One pool per process.  A block belongs to (device, stream, purpose); launches of one stream run in order, so the block of a
// purpose can serve call after call without a free in between (it used to come from the stream-ordered pool: 0.2 ms per call,
// and stalls of 7-58 ms while the driver digested a large free, profiles/r3/readback_after_free.txt).  Rules:
//   * a call LEASES its block from acquire() until its last launch is enqueued (`Lease`, RAII); an image with an automatic
//     extent keeps the lease from ot_detector_image_auto_begin to _finish / _cancel.  A leased block is never handed to
//     another caller and never freed by trim(): a second request for the same key while the first is leased gets a block of
//     its own (two threads on one stream, two open automatic images).
//   * blocks are not tied to the thread that created them: a thread or a torch stream that has gone leaves blocks any later
//     caller on that stream reuses, and trim() or the cap reclaims.
//   * the pool keeps at most `cap` bytes: before it grows beyond that it frees idle blocks, least recently used first; when
//     the allocation itself fails it frees every idle block and tries once more with the exact size.
//   * trim() waits for the device and frees every IDLE block (torch's allocator cannot see this memory: the Python wrapper
//     calls it when a torch allocation runs out of memory, and retries).
// The allocator is a triple of function pointers so that tests/test_scratch_pool.py can drive the same code with malloc on a
// machine without a GPU (tests/cpp/scratch_pool_test.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <list>
#include <mutex>

namespace ot_scratch {

struct Allocator {
    void* (*alloc)(size_t bytes);  // nullptr: out of memory
    void (*release)(void* p);      // waits for device work that may still use p (hipFree does)
    void (*sync)();                // wait for the device (trim)
};

struct Block {
    int dev, purpose;
    void* stream;
    char* p;
    size_t bytes;
    bool busy;
    uint64_t used;  // pool clock at the last release
};

class Pool {
public:
    Pool(Allocator a, size_t cap_bytes) : a_(a), cap_(cap_bytes) {}
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;

    // -> a leased block of at least `bytes`, or nullptr (out of memory)
    Block* acquire(int dev, int purpose, void* stream, size_t bytes) {
        std::lock_guard<std::mutex> lock(m_);
        Block* fit = nullptr;   // smallest idle block of this key that is large enough
        Block* grow = nullptr;  // else: the largest idle block of this key, to be replaced
        for (auto& b : blocks_) {
            if (b.busy || b.dev != dev || b.purpose != purpose || b.stream != stream) continue;
            if (b.bytes >= bytes) {
                if (!fit || b.bytes < fit->bytes) fit = &b;
            } else if (!grow || b.bytes > grow->bytes) {
                grow = &b;
            }
        }
        if (fit) {
            fit->busy = true;
            return fit;
        }
        if (grow) {  // (release waits for the work that may still use the old block)
            drop(grow);
            grow = nullptr;
        }
        const size_t want = bytes + bytes / 8;  // a little room: chunks of slightly different size do not reallocate
        evict_for(want);
        char* p = (char*)a_.alloc(want);
        size_t got = want;
        if (!p) {  // everything idle goes, then the exact size
            evict_for(SIZE_MAX);
            p = (char*)a_.alloc(bytes);
            got = bytes;
        }
        if (!p) return nullptr;
        blocks_.push_back(Block{dev, purpose, stream, p, got, true, ++clock_});
        kept_ += got;
        return &blocks_.back();
    }

    void release(Block* b) {
        if (!b) return;
        std::lock_guard<std::mutex> lock(m_);
        b->busy = false;
        b->used = ++clock_;
    }

    // frees every idle block; -> bytes still kept (leased blocks)
    size_t trim() {
        a_.sync();
        std::lock_guard<std::mutex> lock(m_);
        evict_for(SIZE_MAX);
        return kept_;
    }

    void set_cap(size_t cap_bytes) {
        std::lock_guard<std::mutex> lock(m_);
        cap_ = cap_bytes;
    }

    void stats(size_t* kept_bytes, int* n_blocks, int* n_busy) {
        std::lock_guard<std::mutex> lock(m_);
        int busy = 0;
        for (auto& b : blocks_) busy += b.busy;
        if (kept_bytes) *kept_bytes = kept_;
        if (n_blocks) *n_blocks = (int)blocks_.size();
        if (n_busy) *n_busy = busy;
    }

private:
    void drop(Block* b) {
        a_.release(b->p);
        kept_ -= b->bytes;
        for (auto it = blocks_.begin(); it != blocks_.end(); ++it)
            if (&*it == b) {
                blocks_.erase(it);
                return;
            }
    }

    // idle blocks go, least recently used first, until `want` more bytes fit under the cap (SIZE_MAX: all of them)
    void evict_for(size_t want) {
        for (;;) {
            if (want != SIZE_MAX && (kept_ + want <= cap_ || kept_ == 0)) return;
            Block* lru = nullptr;
            for (auto& b : blocks_)
                if (!b.busy && (!lru || b.used < lru->used)) lru = &b;
            if (!lru) return;
            drop(lru);
        }
    }

    Allocator a_;
    size_t cap_;
    std::mutex m_;
    std::list<Block> blocks_;  // (stable addresses: leases point into it)
    size_t kept_ = 0;
    uint64_t clock_ = 0;
};

// a block for the duration of a scope (or of an object that the lease is moved into)
class Lease {
public:
    Lease() = default;
    Lease(Pool& pool, int dev, int purpose, void* stream, size_t bytes) : pool_(&pool), b_(pool.acquire(dev, purpose, stream, bytes)) {}
    Lease(Lease&& o) noexcept : pool_(o.pool_), b_(o.b_) { o.b_ = nullptr; }
    Lease& operator=(Lease&& o) noexcept {
        if (this != &o) {
            reset();
            pool_ = o.pool_;
            b_ = o.b_;
            o.b_ = nullptr;
        }
        return *this;
    }
    Lease(const Lease&) = delete;
    Lease& operator=(const Lease&) = delete;
    ~Lease() { reset(); }
    void reset() {
        if (b_) pool_->release(b_);
        b_ = nullptr;
    }
    char* p() const { return b_ ? b_->p : nullptr; }
    size_t bytes() const { return b_ ? b_->bytes : 0; }
    explicit operator bool() const { return b_ != nullptr; }

private:
    Pool* pool_ = nullptr;
    Block* b_ = nullptr;
};

}  // namespace ot_scratch
