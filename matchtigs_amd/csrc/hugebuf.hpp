// hugebuf.hpp -- big randomly-accessed host arrays on transparent huge pages, placed on the caller's NUMA node.
// THP is in `madvise` mode on the target hosts: with 4 KB pages every random access is also a TLB miss. The target
// hosts are 2-socket machines: pages first-touched by helper threads on the other socket would double the latency of
// the sequential walk that later chases through them, so buffers are bound (preferred) to the creating thread's node.
#pragma once

#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstddef>
#include <cstdio>
#include <cstring>

#include "host_graph.hpp"

namespace mtg {

inline int current_numa_node() {
    unsigned cpu = 0, node = 0;
    if (syscall(SYS_getcpu, &cpu, &node, nullptr) != 0) return -1;
    return (int)node;
}

// Pins the calling thread to the CPUs of its current NUMA node (never beyond its current mask) for the lifetime of the guard and
// restores the old mask. A host program that manages affinities itself turns this off with MTG_NO_NUMA_PIN=1 (speed only: results
// do not depend on it).
struct NumaPin {
    cpu_set_t old_mask;
    bool active = false;
    int node = -1;
    // preferred >= 0: that node (where an arena's memory lives), else the node the thread is running on
    explicit NumaPin(int preferred = -1) {
        if (std::getenv("MTG_NO_NUMA_PIN")) return;
        node = preferred >= 0 ? preferred : current_numa_node();
        if (node < 0 || sched_getaffinity(0, sizeof old_mask, &old_mask) != 0) return;
        char path[128];
        std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        FILE *f = std::fopen(path, "r");
        if (!f) return;
        char buf[4096];
        const bool ok = std::fgets(buf, sizeof buf, f) != nullptr;
        std::fclose(f);
        if (!ok) return;
        cpu_set_t m;
        CPU_ZERO(&m);
        int n_set = 0;
        for (char *p = buf; *p;) {  // "0-63,128-191"
            char *end;
            long a = std::strtol(p, &end, 10);
            if (end == p) break;
            long b = a;
            if (*end == '-') { p = end + 1; b = std::strtol(p, &end, 10); }
            for (long c = a; c <= b && c < CPU_SETSIZE; c++)
                if (CPU_ISSET(c, &old_mask)) { CPU_SET(c, &m); n_set++; }
            p = (*end == ',') ? end + 1 : end;
            if (*end != ',') break;
        }
        if (n_set > 0 && sched_setaffinity(0, sizeof m, &m) == 0) active = true;
    }
    ~NumaPin() {
        if (active) sched_setaffinity(0, sizeof old_mask, &old_mask);
    }
    NumaPin(const NumaPin &) = delete;
    NumaPin &operator=(const NumaPin &) = delete;
};

template <typename T>
struct HugeBuf {
    T *p = nullptr;
    size_t n = 0, bytes = 0;
    HugeArena *arena = nullptr;  // non-null: the mapping belongs to this arena (taken from it or adopted by it) and goes back there
    // With an arena the content is NOT zero when the mapping is a reused one: callers write before they read.
    explicit HugeBuf(size_t count, HugeArena *from = nullptr) : n(count) {
        bytes = ((count * sizeof(T) + (2u << 20) - 1) / (2u << 20)) * (2u << 20);
        if (bytes == 0) bytes = 2u << 20;
        if (from) {
            if (void *m = from->take(bytes)) {
                p = static_cast<T *>(m);
                arena = from;
                return;
            }
        }
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) MTG_DIE("out of memory (%zu bytes)", bytes);
        madvise(m, bytes, MADV_HUGEPAGE);
        const int node = (from && from->node >= 0) ? from->node : current_numa_node();  // (an arena keeps all its mappings on one node)
        if (from && from->node < 0) from->node = node;
        if (node >= 0 && node < 64) {  // MPOL_PREFERRED = 1: allocate on this node whoever touches the page first
            unsigned long mask = 1ul << node;
            (void)syscall(SYS_mbind, m, bytes, 1 /*MPOL_PREFERRED*/, &mask, sizeof(mask) * 8, 0);
        }
        p = static_cast<T *>(m);
        if (from && from->adopt(m, bytes)) arena = from;
    }
    ~HugeBuf() {
        if (!p) return;
        if (arena && arena->give_back(p)) return;
        munmap(p, bytes);
    }
    HugeBuf(const HugeBuf &) = delete;
    HugeBuf &operator=(const HugeBuf &) = delete;
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

}  // namespace mtg
