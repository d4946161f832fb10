// vstab_hostlogic.hpp -- host-side bookkeeping of the pipeline that does not need a device: the parser of the tracker's result
// records and the import cache of DMA-BUF objects.  Kept apart from vstab_pipeline.cpp so that the CPU test suite (and its
// sanitizer build, tools/run_sanitized_tests.sh) can drive them with hand-made buffers through the vstabx_* test hooks.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

namespace vstab {

// ---- tracker result records (k_lk_track, make_record) -----------------------------------------------------------------------
// One record per feature slot and frame pair: four dwords {x bits, seq, y bits, seq << 2 | status} = two naturally aligned
// 8-byte granules, each carrying its own tag -- nothing promises that the GPU's 16-byte store lands in host memory as one write.
// status: 0 lost in this frame pair, 1 tracked, 2 lost in an earlier pair of the chain (not part of this pair's point list),
// 3 the slot's chain input did not carry its parent's tag (bookkeeping error).
inline bool lk_record_ready(const volatile uint32_t *rec, int i, uint32_t seq) {
    return __atomic_load_n(&rec[4 * i + 1], __ATOMIC_ACQUIRE) == seq && (__atomic_load_n(&rec[4 * i + 3], __ATOMIC_ACQUIRE) & ~3u) == (seq << 2);
}
enum LkParse { LK_PARSE_OK = 0, LK_PARSE_NOT_READY = 1, LK_PARSE_BAD_CHAIN = -1, LK_PARSE_COUNT_MISMATCH = -2 };
// Decode records [first, n) of one frame pair into the compacted point list (appending).  Stops at the first record whose tags are
// not both `seq` (*next = its index: the caller waits and calls again from there).  After the last record the number of entries
// must be expect_n, the number of points that went into the pair.
inline LkParse lk_parse_records(const volatile uint32_t *rec, int first, int n, uint32_t seq, size_t expect_n, std::vector<float> &xy,
                                std::vector<uint8_t> &status, int *next) {
    for (int i = first; i < n; i++) {
        if (!lk_record_ready(rec, i, seq)) {
            *next = i;
            return LK_PARSE_NOT_READY;
        }
        const uint32_t x = rec[4 * i], y = rec[4 * i + 2], s = rec[4 * i + 3] & 3u;
        if (s == 3u) return LK_PARSE_BAD_CHAIN;
        if (s == 2u) continue;
        float fx, fy;
        std::memcpy(&fx, &x, 4), std::memcpy(&fy, &y, 4);
        xy.push_back(fx), xy.push_back(fy), status.push_back((uint8_t)s);
    }
    *next = n;
    return status.size() == expect_n ? LK_PARSE_OK : LK_PARSE_COUNT_MISMATCH;
}

// ---- DMA-BUF import cache ----------------------------------------------------------------------------------------------------
// Objects are recognised by the inode of their descriptor (decoders hand the same pool of surfaces round and round, under fds
// that are closed and reused) plus their size.  At most `cap` objects stay mapped; beyond that the least recently used one is
// unmapped -- but never one used within the last `window` lookups: a frame stays in the pipeline that long (read-ahead +
// look-ahead queue + warp) and may be read in place, so the cache grows past its cap rather than unmap it.
template <typename Handle>
class DmaBufCache {
  public:
    struct Entry {
        unsigned long long ino;
        size_t size;
        Handle handle;
        uint8_t *base;
        long last_use;
    };
    int cap = 256;
    long clock = 0, imports = 0, evictions = 0;
    // the mapped base of object (ino, size): cached, or imported with `import` (-> false on failure; nothing is cached then).
    // `destroy` unmaps an evicted object.
    bool lookup(unsigned long long ino, size_t size, long window, const std::function<bool(Handle &, uint8_t *&)> &import,
                const std::function<void(Handle &)> &destroy, uint8_t *&base) {
        Entry *hit = nullptr;
        for (auto &e : entries_)
            if (e.ino == ino && e.size == size) hit = &e;
        if (!hit) {
            if ((long)entries_.size() >= (long)cap) {
                size_t old = 0;
                for (size_t i = 1; i < entries_.size(); i++)
                    if (entries_[i].last_use < entries_[old].last_use) old = i;
                if (clock - entries_[old].last_use > window) {
                    destroy(entries_[old].handle);
                    entries_.erase(entries_.begin() + (long)old);
                    evictions++;
                }
            }
            Entry e{ino, size, Handle(), nullptr, 0};
            if (!import(e.handle, e.base)) return false;
            entries_.push_back(e);
            hit = &entries_.back();
            imports++;
        }
        hit->last_use = ++clock;
        base = hit->base;
        return true;
    }
    void clear(const std::function<void(Handle &)> &destroy) {
        for (auto &e : entries_) destroy(e.handle);
        entries_.clear();
    }
    size_t size() const { return entries_.size(); }

  private:
    std::vector<Entry> entries_;
};

}  // namespace vstab
