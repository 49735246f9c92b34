// Byte arena with id -> range bookkeeping, a first-fit free list and dirty-range tracking.
//
// This is what lets a world SVO replace one chunk's bytes without rewriting the rest and lets the
// uploader copy only what changed. Behaviour follows the reference's `RangeBuffer`
// (src/world/hds/internal.rs:163-277): insert = drop old range for the id, reuse the FIRST free range
// that is large enough (splitting it), else append; both `free_ranges` and `updated_ranges` are kept
// sorted by start and merged when they touch or overlap (internal.rs:252-272).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <unordered_map>
#include <vector>

namespace vx {

// memcpy for the writers' large ranges (write_to / write_changes_to into the mapped buffer: the reference's one memcpy, esvo.rs:291-339, csvo.rs:262-312): a whole
// depth-14 ESVO world is 6.7 GB, 0.8 s of a first Svo::update on one thread -- from 32 MiB on the range is cut into pieces for up to eight threads.
inline void copy_bytes(uint8_t* dst, const uint8_t* src, size_t n) {
    constexpr size_t kPiece = size_t(32) << 20;
    if (n == 0) return;  // (an empty range's pointers may be null: not memcpy's business)
    if (n < 2 * kPiece) {
        std::memcpy(dst, src, n);
        return;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t threads = std::min<size_t>(std::min<size_t>(8, hw ? hw : 1), n / kPiece);
    const size_t share = ((n + threads - 1) / threads + 4095) & ~size_t(4095);  // (rounded UP first: the shares must cover n -- host_kats, copy_bytes_in_pieces)
    std::vector<std::thread> pool;
    for (size_t t = 1; t < threads && t * share < n; ++t)
        pool.emplace_back([=] { std::memcpy(dst + t * share, src + t * share, std::min(share, n - t * share)); });
    std::memcpy(dst, src, std::min(share, n));
    for (std::thread& t : pool) t.join();
}

struct Range {
    size_t start = 0;
    size_t length = 0;
    bool operator==(const Range& o) const { return start == o.start && length == o.length; }
};

class RangeBuffer {
public:
    std::vector<uint8_t> bytes;
    std::vector<Range> free_ranges;
    std::vector<Range> updated_ranges;
    std::unordered_map<uint64_t, Range> octant_to_range;

    RangeBuffer() = default;
    explicit RangeBuffer(size_t initial_capacity) : bytes(initial_capacity, 0) {
        if (initial_capacity > 0) free_ranges.push_back({0, initial_capacity});
    }

    // internal.rs:196-201 (the reference frees `capacity()` bytes; the arena here is kept fully sized)
    void clear() {
        bytes.resize(bytes.capacity());
        free_ranges.clear();
        free_ranges.push_back({0, bytes.size()});
        updated_ranges.clear();
        octant_to_range.clear();
    }

    // Copies `len` bytes in and returns their start offset (internal.rs:204-237).
    size_t insert(uint64_t id, const uint8_t* buf, size_t len) {
        remove(id);

        size_t ptr = bytes.size();
        auto it = std::find_if(free_ranges.begin(), free_ranges.end(), [len](const Range& r) { return len <= r.length; });
        if (it != free_ranges.end()) {
            ptr = it->start;
            if (len < it->length) {
                it->start += len;
                it->length -= len;
            } else {
                free_ranges.erase(it);
            }
            if (len) std::memcpy(bytes.data() + ptr, buf, len);
        } else {
            bytes.insert(bytes.end(), buf, buf + len);
        }

        octant_to_range[id] = Range{ptr, len};
        updated_ranges.push_back({ptr, len});
        merge_ranges(updated_ranges);
        return ptr;
    }

    // internal.rs:240-249
    void remove(uint64_t id) {
        auto it = octant_to_range.find(id);
        if (it == octant_to_range.end()) return;
        free_ranges.push_back(it->second);
        octant_to_range.erase(it);
        merge_ranges(free_ranges);
    }

    // internal.rs:252-272
    static void merge_ranges(std::vector<Range>& ranges) {
        std::sort(ranges.begin(), ranges.end(), [](const Range& a, const Range& b) { return a.start < b.start; });
        size_t w = 0;  // last kept element
        for (size_t i = 1; i < ranges.size(); ++i) {
            Range& lhs = ranges[w];
            const Range rhs = ranges[i];
            if (rhs.start <= lhs.start + lhs.length) {
                const size_t diff = lhs.start + lhs.length - rhs.start;
                if (rhs.length > diff) lhs.length += rhs.length - diff;
            } else {
                ranges[++w] = rhs;
            }
        }
        if (!ranges.empty()) ranges.resize(w + 1);
    }

    size_t size_in_bytes() const { return bytes.size(); }
};

}  // namespace vx
