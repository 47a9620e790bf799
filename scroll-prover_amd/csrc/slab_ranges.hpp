// slab_ranges.hpp -- the bookkeeping behind mi355_buf_alloc's slabs (lib_core.hip "Slabs"): which address ranges of which slab are free.  Plain C++, no HIP: lib_core.hip owns the
// device memory and the events, this file only answers "where does a block of `want` bytes go" and "which slabs are whole again", so the same code runs under the host-side model
// check in tests/cpp/test_halo2_mirror.cpp (--host-only: random alloc / free sequences against a byte map).
//
// Rules: a free range never spans two slabs (slabs are separate hipMalloc regions even when their addresses happen to touch); adjacent free ranges of one slab are always merged, so a
// slab whose blocks have all come back is exactly one range [base, base + bytes) and can be returned to HIP; carve is best fit (the smallest range that holds the request) and takes
// the front of the range, so blocks of one size pack from the slab's start and the tail stays one large range for the next layer's larger blocks.
// Segregation by size (round 6): a slab made for ONE request larger than the shared slab size is `dedicated`; a request below `small_limit` is never carved out of a dedicated
// slab (a 4 KiB staging block cut from the front of a freed n x 32-byte polynomial slab would keep that slab from ever being whole again: mi355_buf_trim and the out-of-memory
// retry could not return it, and the next polynomial of that size would need a fresh hipMalloc next to it).  Small blocks live in shared slabs only.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <vector>

namespace mi355zk {

struct SlabRanges {
  struct Slab { uintptr_t base; size_t bytes; bool dedicated = false; };
  size_t small_limit = 0;                    // requests below this never go into a dedicated slab (0: no segregation)
  std::vector<Slab> slabs;
  std::map<uintptr_t, size_t> free_ranges;   // start -> length, by address

  const Slab *slab_of(uintptr_t p) const { for (const auto &s : slabs) if (p >= s.base && p < s.base + s.bytes) return &s; return nullptr; }
  // a new slab, entirely free
  void add_slab(uintptr_t base, size_t bytes, bool dedicated = false) { slabs.push_back({base, bytes, dedicated}); free_ranges[base] = bytes; }
  // [p, p + len) comes back; it must lie inside one slab and must not overlap a free range (the caller hands back exactly what carve returned)
  void insert(uintptr_t p, size_t len) {
    const Slab *s = slab_of(p);
    auto nx = free_ranges.lower_bound(p);
    if (s && nx != free_ranges.end() && p + len == nx->first && nx->first < s->base + s->bytes) { len += nx->second; nx = free_ranges.erase(nx); }
    if (s && nx != free_ranges.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == p && pv->first >= s->base) { pv->second += len; return; } }
    free_ranges[p] = len;
  }
  // best fit; 0 = no range holds `want`
  uintptr_t carve(size_t want) {
    auto best = free_ranges.end();
    const bool small = want < small_limit;
    for (auto it = free_ranges.begin(); it != free_ranges.end(); ++it) {
      if (it->second < want || (best != free_ranges.end() && it->second >= best->second)) continue;
      if (small) { const Slab *s = slab_of(it->first); if (s && s->dedicated) continue; }
      best = it;
    }
    if (best == free_ranges.end()) return 0;
    const uintptr_t p = best->first; const size_t len = best->second; free_ranges.erase(best);
    if (len > want) free_ranges[p + want] = len - want;
    return p;
  }
  // slabs that are one free range again leave the bookkeeping; their bases are appended to `out`; returns the bytes they held
  size_t take_whole_slabs(std::vector<uintptr_t> &out) {
    size_t bytes = 0;
    for (auto it = slabs.begin(); it != slabs.end();) {
      auto f = free_ranges.find(it->base);
      if (f != free_ranges.end() && f->second == it->bytes) { free_ranges.erase(f); out.push_back(it->base); bytes += it->bytes; it = slabs.erase(it); } else ++it;
    }
    return bytes;
  }
  size_t free_bytes() const { size_t b = 0; for (const auto &kv : free_ranges) b += kv.second; return b; }
  void clear() { slabs.clear(); free_ranges.clear(); }
};

}  // namespace mi355zk
