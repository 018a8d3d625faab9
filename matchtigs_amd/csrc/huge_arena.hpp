// huge_arena.hpp -- large host mappings a graph keeps between calls. The reference-order Euler walk needs 256 bytes of records
// per node plus three entry arrays: mapping, faulting in and unmapping those 26 GB on every call cost ~1 s of an 11-s step on the
// 2^27 bench graph. A HostGraph owns one arena; its mappings are reused by the next call on the same graph and released with it
// (matchtigs_compute_tigs frees its graph, so a one-shot caller never keeps anything).
#pragma once
#include <sys/mman.h>

#include <cstddef>
#include <mutex>
#include <vector>

namespace mtg {

struct HugeArena {
    struct Slot {
        void *p;
        size_t bytes;
        bool busy;
        bool pinned = false;  // registered with the HIP runtime (page-locked) by the device finish: unregistered before it is unmapped
        unsigned uses = 0;    // times the mapping was handed out
    };
    // set by the HIP translation unit that pins mappings (hipHostUnregister); host-only builds never pin
    static void (*&unpin_hook())(void *) {
        static void (*hook)(void *) = nullptr;
        return hook;
    }
    static void unmap_slot(const Slot &s) {
        if (s.pinned && unpin_hook()) unpin_hook()(s.p);
        munmap(s.p, s.bytes);
    }
    std::vector<Slot> slots;
    std::mutex mu;
    int node = -1;  // NUMA node the mappings were placed on (the first user's): later users run there, next to the memory
    static constexpr size_t MAX_SLOTS = 8;

    HugeArena() = default;
    HugeArena(const HugeArena &) {}  // a copied graph starts with an empty arena
    HugeArena &operator=(const HugeArena &) { return *this; }
    ~HugeArena() { release(); }

    // a free mapping of at least `bytes` and at most 25 % more, or nullptr (content: whatever the last user left)
    void *take(size_t bytes) {
        std::lock_guard<std::mutex> l(mu);
        for (Slot &s : slots)
            if (!s.busy && s.bytes >= bytes && s.bytes <= bytes + bytes / 4) {
                s.busy = true;
                s.uses++;
                return s.p;
            }
        return nullptr;
    }
    // the arena takes over a mapping that is in use by the caller; false = no room (the caller unmaps it itself when done)
    bool adopt(void *p, size_t bytes) {
        std::lock_guard<std::mutex> l(mu);
        if (slots.size() >= MAX_SLOTS) {  // drop a free one of another size
            bool dropped = false;
            for (size_t i = 0; i < slots.size() && !dropped; i++)
                if (!slots[i].busy) {
                    unmap_slot(slots[i]);
                    slots.erase(slots.begin() + (long)i);
                    dropped = true;
                }
            if (!dropped) return false;
        }
        slots.push_back(Slot{p, bytes, true, false, 1});
        return true;
    }
    // (uses, pinned) of the mapping at p; uses == 0: not an arena mapping
    unsigned uses_of(void *p, bool *pinned = nullptr) {
        std::lock_guard<std::mutex> l(mu);
        for (Slot &s : slots)
            if (s.p == p) {
                if (pinned) *pinned = s.pinned;
                return s.uses;
            }
        return 0;
    }
    void set_pinned(void *p) {
        std::lock_guard<std::mutex> l(mu);
        for (Slot &s : slots)
            if (s.p == p) s.pinned = true;
    }
    bool give_back(void *p) {
        std::lock_guard<std::mutex> l(mu);
        for (Slot &s : slots)
            if (s.p == p) {
                s.busy = false;
                return true;
            }
        return false;
    }
    // page-locked mappings are given back to the runtime (so that release() makes no HIP call: it may run on a thread of its own)
    void unpin_all() {
        std::lock_guard<std::mutex> l(mu);
        for (Slot &s : slots)
            if (s.pinned && unpin_hook()) {
                unpin_hook()(s.p);
                s.pinned = false;
            }
    }
    void release() {
        std::lock_guard<std::mutex> l(mu);
        for (Slot &s : slots) unmap_slot(s);
        slots.clear();
    }
};

}  // namespace mtg
