// device_sssp.hip -- the bounded many-to-many search of the device stage (MI355X / gfx950): cooperative levels, the table-free enumeration
// level with its post-pass, the dense level, the level cascade and its counters. The other translation units of the stage:
// device_build.hip (device graph), device_classify.hip, device_replay.hip (claim loop), device_pairs.hip (one-shot and multi-GPU drivers).
//
// Replaces, for ALL sources at once, what the reference does one source at a time on the CPU:
//   * node classification                      /root/reference/src/implementation/greedytigs/mod.rs:222-255
//   * Dijkstra::shortest_path_lens (bounded)   call site greedytigs/mod.rs:324-335 (traitgraph-algo 8.1.2)
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   NodeBlock[V] 64-byte family block per node: <=4 inline (neighbour, clamped weight) pairs, degree, in-node flag,
//                its children's in-node flags and, while they fit, its children's out-edges. One aligned gather
//                fetches everything a path enumeration needs about a node AND its embedded children (the
//                cooperative levels read the first 32 bytes only). Nodes with more than 4 out-edges (never the
//                case in a de Bruijn graph) spill to a CSR side array. Built on the GPU from the edge arrays.
//   out_nodes[S] ascending source list (u32), mult[V] (i32) from classification.
//   pool[]       candidate keys (distance << 32 | node), per source contiguous and ascending.
//
// SSSP stage (integer, gather/latency bound, no MFMA) = a cascade of levels; a source a
// level cannot finish is appended to a device list by that kernel and re-run from scratch by the next:
//   level 0   sssp_enum_kernel: one LANE per source enumerates the bounded paths depth-first with a private LDS stack (a
//             (k-1)-ball of a unitig graph is almost a tree, so no visited table is needed); one 64-byte block gather per step,
//             self-refilling lanes, no barrier; results written straight to memory; fix_compact_kernel + sort_lists_kernel put
//             the lists that are not yet in Dijkstra order in order. Sources beyond its budgets go to level 1+.
//   level 1+  sssp_kernel: a workgroup takes a batch of BSRC sources and runs all their bounded searches
//             together as ONE label-correcting wavefront over a shared LDS open-addressing table keyed by
//             (local source, node) -> tentative distance (64-bit entries, ds_cmpst_b64 / ds_min_u64) with
//             one append-only frontier log. Work is balanced over frontier items, not over sources.
//             Distances are exact when the log drains (non-negative weights), so the result equals
//             Dijkstra's regardless of relaxation order. Emission ranks each source's targets by
//             (distance, node) in LDS and writes them contiguously. The last level keeps table and log
//             in a global workspace.
// Claim loop (greedytigs/mod.rs:301-523) on the device: replay_kernels.inc.
#include "device_internal.hpp"

namespace mtg {

__global__ __launch_bounds__(64) void active_range_kernel(const uint32_t *act_index, const unsigned long long *n_act_total, uint64_t src_begin,
                                                          uint64_t src_end, unsigned long long *act_begin, unsigned long long *act_count) {
    const int lane = threadIdx.x;
    const unsigned long long n = *n_act_total;
    auto lower_bound = [&](uint64_t key) -> unsigned long long {  // first i in [0, n] with act_index[i] >= key
        unsigned long long lo = 0, hi = n;
        while (lo < hi) {
            const unsigned long long chunk = (hi - lo + 63) / 64;
            const unsigned long long idx = lo + (unsigned long long)lane * chunk;
            const bool ge = idx >= hi || (uint64_t)act_index[idx] >= key;  // (monotone over the lanes)
            const unsigned long long m = __ballot(ge);
            const int first = m ? __builtin_ctzll(m) : 64;
            if (first == 0) break;  // act_index[lo] >= key
            const unsigned long long new_lo = lo + (unsigned long long)(first - 1) * chunk + 1;
            const unsigned long long at_first = lo + (unsigned long long)first * chunk;
            hi = (first < 64 && at_first < hi) ? at_first : hi;
            lo = new_lo;
        }
        return lo;
    };
    const unsigned long long b = lower_bound(src_begin), e = lower_bound(src_end);
    if (lane == 0) { *act_begin = b; *act_count = e - b; }
}

template <bool GLOBAL_WS>
struct Mem {
    // LDS: workgroup scope suffices. Global workspace: agent scope so that loads bypass the CU's L1
    // (atomics execute in L2 and do not refresh an L1-resident line).
    static constexpr int SCOPE = GLOBAL_WS ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP;
    template <typename T> static __device__ __forceinline__ T ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
    template <typename T> static __device__ __forceinline__ void st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ unsigned long long cas(unsigned long long *p, unsigned long long expect, unsigned long long v) {
        __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, SCOPE);
        return expect;  // previous value
    }
    static __device__ __forceinline__ unsigned long long fmin(unsigned long long *p, unsigned long long v) {
        return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, SCOPE);
    }
    static __device__ __forceinline__ void fand(unsigned long long *p, unsigned long long v) {
        __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, SCOPE);
    }
};

template <int LOGH>
__device__ __forceinline__ uint32_t tbl_hash(uint32_t src, uint32_t node) {
    const uint32_t x = node * 0x9E3779B1u + src * 0x85EBCA77u;
    return (x ^ (x >> 15)) * 0x2C1B3C6Du >> (32 - LOGH);
}

// returns > 0 improved (2 = new entry, 1 = existing entry; slot set), 0 not improved, -1 table full
template <int LOGH, bool GLOBAL_WS>
__device__ __forceinline__ int tbl_relax(unsigned long long *table, uint32_t src, uint32_t node, uint32_t dist, uint32_t &slot) {
    constexpr uint32_t H = 1u << LOGH;
    constexpr int MAX_PROBE = H < 256 ? (int)H : 256;
    const unsigned long long key = ((unsigned long long)src << ENT_SRC_SHIFT) | ((unsigned long long)node << ENT_NODE_SHIFT);
    const unsigned long long val = key | ((unsigned long long)dist << 1) | 1ull;
    uint32_t h = tbl_hash<LOGH>(src, node);
    for (int probe = 0; probe < MAX_PROBE; probe++) {
        unsigned long long cur = Mem<GLOBAL_WS>::ld(&table[h]);
        if (cur == TBL_EMPTY) {
            cur = Mem<GLOBAL_WS>::cas(&table[h], TBL_EMPTY, val);
            if (cur == TBL_EMPTY) { slot = h; return 2; }
        }
        if ((cur >> ENT_NODE_SHIFT) == (key >> ENT_NODE_SHIFT)) {
            if (((cur >> 1) & ENT_DIST_MASK) <= dist) return 0;
            const unsigned long long old = Mem<GLOBAL_WS>::fmin(&table[h], val);
            slot = h;
            return (((old >> 1) & ENT_DIST_MASK) > dist) ? 1 : 0;
        }
        h = (h + 1) & (H - 1);
    }
    return -1;
}

// ---- SSSP kernel, version 2 ------------------------------------------------------------------
// Changes against the first correct kernel (profiles/r01_baseline_*): no global atomic per batch any more
// (batches are strided statically over the persistent grid; pool space is taken in block-local chunks;
// counters are flushed once per block), the frontier is ONE append-only log so that emission and the
// table clean-up walk only the touched slots instead of scanning/clearing the whole table per batch.
constexpr unsigned long long POOL_CHUNK = 2048;  // keys per block-local pool chunk

template <int BLOCK, int LOGH, int QCAP, int SCAP, int BSRC, bool COUNT, bool GLOBAL_WS>
struct SsspLds {
    static constexpr uint32_t H = 1u << LOGH;
    unsigned long long table[GLOBAL_WS ? 1 : H];
    unsigned long long stage[GLOBAL_WS ? 1 : SCAP];  // emission staging: keys per source segment
    uint32_t log[GLOBAL_WS ? 1 : QCAP];              // every push of the batch, in push order (rounds are ranges)
    uint16_t stage_src[GLOBAL_WS ? 1 : SCAP];
    uint32_t srcnode[BSRC];
    uint32_t cnt[BSRC];
    uint32_t off[BSRC];
    uint32_t fill[BSRC];
    uint32_t tail;   // next free log position (may run past QCAP -> overflow)
    uint32_t end;    // snapshot of tail taken between two barriers
    uint32_t ovf;
    uint32_t total;  // keys emitted by this batch
    unsigned long long base;                    // pool position of this batch's keys
    unsigned long long chunk_next, chunk_end;   // block-local pool chunk
    unsigned long long bt_settled, bt_relaxed, bt_attempts;              // per batch (COUNT)
    unsigned long long st_settled, st_relaxed, st_attempts, st_emitted;  // per block (COUNT)
    unsigned long long st_pushes, st_max_log, st_max_ent;
};

template <int BLOCK, int LOGH, int QCAP, int SCAP, int BSRC, bool COUNT, bool GLOBAL_WS>
__global__ __launch_bounds__(BLOCK) void sssp_kernel(SsspArgs a) {
    static_assert(BSRC <= 512 && BSRC <= BLOCK && BLOCK >= 64, "BSRC/BLOCK limits");
    static_assert(LOGH <= 22, "slot must fit the log item");
    constexpr uint32_t H = 1u << LOGH;
    constexpr int HINT_BITS = 32 - LOGH;  // distance (hint) bits of a log item
    constexpr uint32_t HINT_MASK = (1u << HINT_BITS) - 1u;
    // With >= 16 hint bits the hint IS the distance (k-1 <= 65535): the push that carries an entry's final
    // distance is unique, so the log doubles as the list of live table slots.
    constexpr bool LOG_EMIT = HINT_BITS >= 16;
    using M = Mem<GLOBAL_WS>;
    __shared__ SsspLds<BLOCK, LOGH, QCAP, SCAP, BSRC, COUNT, GLOBAL_WS> s;

    unsigned long long *table, *stage;
    uint32_t *log;
    uint16_t *stage_src;
    if constexpr (GLOBAL_WS) {
        unsigned long long *base = a.ws + (uint64_t)blockIdx.x * a.ws_stride;
        table = base;
        stage = base + H;
        log = reinterpret_cast<uint32_t *>(base + H + SCAP);
        stage_src = reinterpret_cast<uint16_t *>(base + H + SCAP + (QCAP + 1) / 2);
    } else {
        table = s.table;
        stage = s.stage;
        log = s.log;
        stage_src = s.stage_src;
    }

    const int tid = threadIdx.x;
    const uint64_t n_batches = (a.n_items + BSRC - 1) / BSRC;

    for (uint32_t i = tid; i < H; i += BLOCK) M::st(&table[i], TBL_EMPTY);
    if (tid == 0) {
        s.chunk_next = 0; s.chunk_end = 0;
        s.st_settled = 0; s.st_relaxed = 0; s.st_attempts = 0; s.st_emitted = 0;
        s.st_pushes = 0; s.st_max_log = 0; s.st_max_ent = 0;
    }
    __syncthreads();

    for (uint64_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        const uint64_t item0 = batch * BSRC;
        const int nsrc = (int)min((uint64_t)BSRC, a.n_items - item0);
        // ---- init (the table is clean here) ----
        if (tid < BSRC) { s.cnt[tid] = 0; s.fill[tid] = 0; }
        if (tid == 0) { s.tail = (uint32_t)nsrc; s.ovf = 0; s.bt_settled = 0; s.bt_relaxed = 0; s.bt_attempts = 0; }
        __syncthreads();
        uint32_t begin = 0;
        if (tid < nsrc) {
            const uint64_t abs_idx = a.src_index ? a.src_index[item0 + tid] : a.src_begin + item0 + tid;
            const uint32_t node = a.sources[abs_idx];
            s.srcnode[tid] = node;
            uint32_t slot = 0;
            const int r = tbl_relax<LOGH, GLOBAL_WS>(table, (uint32_t)tid, node, 0u, slot);
            if (r < 0) s.ovf = 1;
            M::st(&log[tid], (slot << HINT_BITS) | 0u);
        }
        __syncthreads();
        if (tid == 0) s.end = s.ovf ? 0u : (uint32_t)nsrc;  // loop bounds are only ever published between two barriers
        __syncthreads();

        // ---- label-correcting rounds: round r processes log[begin, end), pushes append at tail ----
        uint32_t end = s.end;
        for (int round = 0; begin < end; round++) {  // uniform: `end` is a snapshot published by thread 0
            // one lane per (frontier item, inline edge j): the four lanes of an item read the same 32-byte record
            // (one memory request) and each relaxes one edge, instead of one lane running four divergent table updates
            for (uint32_t w = begin * 4 + tid; w < end * 4; w += BLOCK) {
                const uint32_t i = w >> 2, j = w & 3u;
                const uint32_t item = M::ld(&log[i]);
                const uint32_t slot = item >> HINT_BITS;
                const unsigned long long e = M::ld(&table[slot]);
                const uint32_t d = (uint32_t)((e >> 1) & ENT_DIST_MASK);
                if ((d & HINT_MASK) != (item & HINT_MASK)) continue;  // superseded by a shorter distance
                const uint32_t node = (uint32_t)(e >> ENT_NODE_SHIFT);
                const uint32_t src = (uint32_t)(e >> ENT_SRC_SHIFT);
                const uint32_t *rw = reinterpret_cast<const uint32_t *>(a.recs + node);
                const uint32_t meta = rw[6];  // deg | flags << 8
                const uint32_t nb_j = rw[j];  // issued together with meta: one memory latency per round, not two
                const uint32_t w_pair = rw[4 + (j >> 1)];
                const uint32_t flags = (meta >> 8) & 0xFFu;
                if (j == 0 && (flags & F_TARGET)) M::fand(&table[slot], ~1ull);  // node property: confirmed in-node
                uint32_t pushed_ovf = 0;
                auto relax = [&](uint32_t nb, uint32_t wt, uint32_t wl) {  // wl = weight + lower bound of nb (= wt without one)
                    const uint32_t nd = d + wt;
                    if (nd > a.K1) return;
                    if (a.prune && d + wl > a.K1) return;  // no in-node behind nb within the bound (build_lb_kernel)
                    uint32_t nslot = 0;
                    const int r = tbl_relax<LOGH, GLOBAL_WS>(table, src, nb, nd, nslot);
                    if (r > 0) {
                        const uint32_t pos = atomicAdd(&s.tail, 1u);
                        if (pos < (uint32_t)QCAP) M::st(&log[pos], (nslot << HINT_BITS) | (nd & HINT_MASK));
                        else pushed_ovf = 1;
                    } else if (r < 0) pushed_ovf = 1;
                };
                if (!(flags & F_EXT)) {
                    const uint32_t deg = meta & 0xFFu;
                    const uint32_t wslot = (w_pair >> ((j & 1u) * 16)) & 0xFFFFu;
                    // (8:8 format: the high byte is weight + lb+(child), what the child's block is needed for beyond the child itself -- an
                    // in-node child, cmeta bit j, is needed for itself: this kernel emits a node from its own entry)
                    if (j < deg) relax(nb_j, wslot & a.wmask, a.wmask == 0xFFu ? (((meta >> (16 + j)) & 1u) ? (wslot & 0xFFu) : wslot >> 8) : wslot);
                    if constexpr (COUNT) { if (j == 0) atomicAdd(&s.bt_attempts, (unsigned long long)deg); }
                } else if (j == 0) {  // spilled adjacency (more than 4 out-edges): one lane walks the list
                    const uint64_t eb = ((uint64_t)rw[1] << 32) | nb_j;  // j == 0: nb_j is word 0
                    const uint32_t deg = rw[2];
                    for (uint32_t q = 0; q < deg; q++) relax(a.ext_col[eb + q], a.ext_w[eb + q], a.ext_w[eb + q]);
                    if constexpr (COUNT) atomicAdd(&s.bt_attempts, (unsigned long long)deg);
                }
                if (pushed_ovf) s.ovf = 1;
            }
            __syncthreads();
            if (tid == 0) {
                if (round > (1 << 20)) s.ovf = 1;
                s.end = s.ovf ? end : min(s.tail, (uint32_t)QCAP);  // on overflow: no progress -> the loop ends
            }
            __syncthreads();
            begin = end;
            end = s.end;
        }
        __syncthreads();
        const uint32_t n_log = min(s.tail, (uint32_t)QCAP);

        // visits every live table entry exactly once: f(entry)
        auto for_each_entry = [&](auto &&f) {
            if constexpr (LOG_EMIT) {
                for (uint32_t i = tid; i < n_log; i += BLOCK) {
                    const uint32_t item = M::ld(&log[i]);
                    const unsigned long long e = M::ld(&table[item >> HINT_BITS]);
                    if (((uint32_t)((e >> 1) & ENT_DIST_MASK) & HINT_MASK) != (item & HINT_MASK)) continue;  // not the final push of this slot
                    f(e);
                }
            } else {
                for (uint32_t i = tid; i < H; i += BLOCK) {
                    const unsigned long long e = M::ld(&table[i]);
                    if (e != TBL_EMPTY) f(e);
                }
            }
        };

        // ---- emission ----
        if (!s.ovf) {
            for_each_entry([&](unsigned long long e) {  // pass 1: per-source counts
                const uint32_t node = (uint32_t)(e >> ENT_NODE_SHIFT);
                const uint32_t src = (uint32_t)(e >> ENT_SRC_SHIFT);
                if constexpr (COUNT) {
                    const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs + node);
                    const uint4 lo = rp[0];
                    const uint4 hi = rp[1];
                    const uint32_t flags = (hi.z >> 8) & 0xFFu;
                    // the units of the pruned search: every entry is a settled node (its distance is final and used), but an in-node
                    // with nothing beyond it within the bound is not EXPANDED -- the enumeration level records it from its parent's
                    // block and never gathers its own, so its out-edges are not relaxed (here it has an entry because this kernel
                    // emits a node from its own entry). lb+(node) = the smallest "what the child is needed for" of its own slots.
                    bool gathered = true;
                    if (a.prune && !(flags & F_EXT) && node != s.srcnode[src]) {
                        const uint32_t ws[4] = {hi.x & 0xFFFFu, hi.x >> 16, hi.y & 0xFFFFu, hi.y >> 16};
                        uint32_t beyond = 255u;
                        for (uint32_t j = 0; j < (hi.z & 0xFFu); j++) beyond = min(beyond, ((hi.z >> (16 + j)) & 1u) ? (ws[j] & 0xFFu) : (ws[j] >> 8));
                        gathered = (uint32_t)((e >> 1) & ENT_DIST_MASK) + beyond <= a.K1;
                    }
                    atomicAdd(&s.bt_settled, 1ull);
                    if (gathered) atomicAdd(&s.bt_relaxed, (unsigned long long)((flags & F_EXT) ? lo.z : (hi.z & 0xFFu)));
                }
                if (!(e & 1ull) && node != s.srcnode[src]) atomicAdd(&s.cnt[src], 1u);
            });
            __syncthreads();
            if (tid < 64) {  // exclusive scan of cnt[0..nsrc) by the first wave, then pool space for the batch
                uint32_t running = 0;
                for (int base = 0; base < BSRC; base += 64) {
                    const int i = base + tid;
                    const uint32_t v = i < nsrc ? s.cnt[i] : 0;
                    uint32_t incl = v;
                    for (int dd = 1; dd < 64; dd <<= 1) {
                        const uint32_t t = __shfl_up(incl, dd);
                        if (tid >= dd) incl += t;
                    }
                    if (i < BSRC) s.off[i] = running + incl - v;
                    running += __shfl(incl, 63);
                }
                if (tid == 0) {
                    s.total = running;
                    // staging too small -> larger level. The ranking below is quadratic in a source's candidates: fine up to the
                    // few thousand the LDS levels can stage, not for the millions the global-workspace level could -- beyond
                    // 32768 (10^9 comparisons, a few ms) that level hands the source to the dense level, which sorts (n log n).
                    if (running > (uint32_t)SCAP || (GLOBAL_WS && running > 32768u)) s.ovf = 1;
                    else if constexpr (!COUNT) {
                        if (s.chunk_next + running > s.chunk_end) {
                            const unsigned long long grab = running > POOL_CHUNK ? (unsigned long long)running : POOL_CHUNK;
                            s.chunk_next = atomicAdd(&a.counters[C_POOL], grab);
                            s.chunk_end = s.chunk_next + grab;
                        }
                        s.base = s.chunk_next;
                        s.chunk_next += running;
                    } else s.base = 0;
                }
            }
            __syncthreads();
        }
        if (!s.ovf) {
            if constexpr (!COUNT) {
                for_each_entry([&](unsigned long long e) {  // pass 2: stage keys per source segment (unsorted)
                    if (e & 1ull) return;
                    const uint32_t node = (uint32_t)(e >> ENT_NODE_SHIFT);
                    const uint32_t src = (uint32_t)(e >> ENT_SRC_SHIFT);
                    if (node == s.srcnode[src]) return;
                    const uint32_t pos = s.off[src] + atomicAdd(&s.fill[src], 1u);
                    M::st(&stage[pos], (((e >> 1) & ENT_DIST_MASK) << 32) | (unsigned long long)node);
                    M::st(&stage_src[pos], (uint16_t)src);
                });
                __syncthreads();
                // pass 3: rank within the source's segment (= sort by (distance, node)), write contiguous
                const uint32_t total = s.total;
                const unsigned long long base = s.base;
                for (uint32_t i = tid; i < total; i += BLOCK) {
                    const uint32_t src = M::ld(&stage_src[i]);
                    const unsigned long long key = M::ld(&stage[i]);
                    const uint32_t lo_ = s.off[src], hi_ = lo_ + s.cnt[src];
                    uint32_t rank = 0;
                    for (uint32_t j = lo_; j < hi_; j++) rank += mtg_policy_pops_before(M::ld(&stage[j]), key) ? 1u : 0u;  // policy P1 (mtg_policy.h)
                    const unsigned long long dst = base + lo_ + rank;
                    if (dst < a.pool_cap) a.pool[dst] = key;
                }
            }
            if (tid < nsrc) {
                const uint64_t abs_idx = a.src_index ? a.src_index[item0 + tid] : a.src_begin + item0 + tid;
                const uint64_t o = abs_idx - a.src_begin;
                a.cand_start[o] = s.base + s.off[tid];
                a.cand_count[o] = s.cnt[tid];
            }
            if constexpr (COUNT) {
                if (tid == 0) {
                    s.st_settled += s.bt_settled; s.st_relaxed += s.bt_relaxed; s.st_attempts += s.bt_attempts;
                    s.st_emitted += s.total;
                    s.st_pushes += n_log;
                    if (n_log > s.st_max_log) s.st_max_log = n_log;
                    if (s.bt_settled > s.st_max_ent) s.st_max_ent = s.bt_settled;
                }
            }
        } else {
            if (tid == 0) s.base = atomicAdd(&a.counters[C_OVERFLOW], (unsigned long long)nsrc);  // slot in the overflow list
            __syncthreads();
            if (tid < nsrc) {
                const uint64_t abs_idx = a.src_index ? a.src_index[item0 + tid] : a.src_begin + item0 + tid;
                a.cand_count[abs_idx - a.src_begin] = CAND_OVERFLOW;
                a.ovf_list[s.base + tid] = (uint32_t)abs_idx;
            }
        }
        __syncthreads();

        // ---- clean the table for the next batch ----
        if (LOG_EMIT && !s.ovf) {
            for (uint32_t i = tid; i < n_log; i += BLOCK) M::st(&table[M::ld(&log[i]) >> HINT_BITS], TBL_EMPTY);
        } else {  // overflowed batches may have table entries that never reached the log
            for (uint32_t i = tid; i < H; i += BLOCK) M::st(&table[i], TBL_EMPTY);
        }
        __syncthreads();
    }

    if constexpr (COUNT) {
        if (tid == 0) {
            atomicAdd(&a.counters[C_SETTLED], s.st_settled);
            atomicAdd(&a.counters[C_RELAXED], s.st_relaxed);
            atomicAdd(&a.counters[C_ATTEMPTS], s.st_attempts);
            atomicAdd(&a.counters[C_EMITTED], s.st_emitted);
            atomicAdd(&a.counters[C_PUSHES], s.st_pushes);
            atomicMax(&a.counters[C_MAX_LOG], s.st_max_log);
            atomicMax(&a.counters[C_MAX_ENT], s.st_max_ent);
        }
    }
}

// Wave-level helpers shared by the lane-per-source level

// Overflowed sources of a wave are buffered in LDS and appended to the global overflow list 64 at a time, so the
// list cursor sees one atomic per 64 sources instead of one per overflow event.
struct WaveOvfBuf {
    uint32_t buf[128];
};
__device__ __forceinline__ void wave_ovf_push(WaveOvfBuf &w, uint32_t &nbuf, bool ovf, uint32_t abs_idx, const SsspArgs &a, int lane) {
    const unsigned long long m = __ballot(ovf);
    if (!m) return;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (ovf) w.buf[nbuf + rank] = abs_idx;
    nbuf += (uint32_t)__popcll(m);
    if (nbuf >= 64) {
        unsigned long long p0 = 0;
        if (lane == 0) p0 = atomicAdd(&a.counters[C_OVERFLOW], 64ull);
        p0 = __shfl(p0, 0);
        a.ovf_list[p0 + lane] = w.buf[lane];
        const uint32_t rest = w.buf[64 + lane];
        w.buf[lane] = rest;
        nbuf -= 64;
    }
}
__device__ __forceinline__ void wave_ovf_flush(WaveOvfBuf &w, uint32_t nbuf, const SsspArgs &a, int lane) {
    if (!nbuf) return;
    unsigned long long p0 = 0;
    if (lane == 0) p0 = atomicAdd(&a.counters[C_OVERFLOW], (unsigned long long)nbuf);
    p0 = __shfl(p0, 0);
    if ((uint32_t)lane < nbuf) a.ovf_list[p0 + lane] = w.buf[lane];
}

// ------------------------------------------------------------------------------------------------
// Lane-per-source kernel WITHOUT a table (level 0 of the default plan)
//
// A (k-1)-ball of a unitig graph is almost a tree (bench graph: 1.0004 path enumerations per settled node), so the search needs
// no visited set at all: every lane enumerates the bounded PATHS from its source depth-first with a small private stack in LDS --
// gather the node's 64-byte family block (the node, its children's in-node flags and the embedded children's out-edges), remember
// the in-nodes among the node and its children, push the grandchildren (and not-embedded children) whose distance stays <= k-1,
// pop the next one. No select-min, no find, no insert. A node reached along two paths is expanded twice (bounded: every edge
// weighs >= 1 and the path length is capped at k-1), its target hits are de-duplicated (minimum distance) and sorted by
// (distance, node) -- lists of up to four in registers when the source finishes, longer ones by the post-pass below --, which
// makes the output identical to the Dijkstra order. A source whose enumeration exceeds the step budget or its LDS space is handed
// to the cooperative cascade, which is exact for any ball.
//
// What bounds it, as it stands at the end of round 6: the search itself runs at the ceiling of dependent gathers on the
// table (2^27: 0.75 ms alone = 45 G gathers/s of 44-48); the rest of the kernel's time is the results' way out (keys, the lists'
// places, the post-pass work list), whose cost follows the number of distinct lines its store instructions touch. The vector pipes
// are a third busy; 2, 3 or 4 workgroups of four waves per CU measured 1.82 / 1.70 / 1.72 ms for the stage: three it is -- by the
// kernel's 133 registers and by its LDS (12 KB per wave: home blocks 4 KB, 25 extension blocks of 24 entries 5 KB, the keys' ring 1 KB,
// the record table of four chunks of sources 2 KB), which three workgroups per CU let it have.
// History: round 2 wrote the step branch-free (every potential push / hit an UNCONDITIONAL LDS store to the lane's next free slot,
// 4 + 4 slots per lane in a slot-major first tier + one 16-entry extension block) because the version before it was bound by
// instruction issue; round 6's counters showed the opposite regime and replaced that form (see the template's comment below).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t ENUM_POP_BUDGET = 256;
constexpr uint32_t ENUM_FIX_CHUNK = 512;    // post-pass work-list slots a wave takes per global atomic (unused ones hold FIX_NONE)
constexpr uint32_t FIX_CLASS_TAG = 0xFFFFFF00u;  // slot 0 of a work-list chunk: FIX_CLASS_TAG | class (0: <= 8 keys, 1: <= 16, 2: <= 32)
constexpr uint32_t FIX_NONE = 0xFFFFFFFFu;
#ifndef MTG_ENUM_HOME
#define MTG_ENUM_HOME 8
#define MTG_ENUM_NB 25  // (25 blocks of 16 entries: 62 187 sources of the 2^27 bench graph outgrow the level, every one of them because its block is
#define MTG_ENUM_BE 24  //  full -- 17 blocks of 24 entries: 19 388, half of them because the pool is empty -- 25 of 24: 9 798, cascade 0.20 -> 0.05 ms)
#endif
#ifndef MTG_ENUM_TSLOTS
#define MTG_ENUM_TSLOTS 4  // chunks of sources whose records wait in the wave's table (a power of two)
#endif
#ifndef MTG_ENUM_KRING
#define MTG_ENUM_KRING 128  // keys of the wave's output ring (a power of two)
#endif
constexpr int ENUM_BE = MTG_ENUM_BE;        // entries per extension block
// LDS words between the starts of two extension blocks: one more than a block holds. With a stride of
// 16 eight-byte entries (128 bytes = all 32 banks once) the same row of every block falls on the same bank pair, and lanes that work
// in different blocks -- the common case -- collide on every access (3.6 conflict cycles per LDS instruction in the round-2 PMC
// pass); an odd stride walks the rows of consecutive blocks through the banks.
constexpr int ENUM_BS = MTG_ENUM_BE + 1;

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {  // value of the first lane, known uniform to the compiler
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

#ifdef MTG_ENUM_STATS
__device__ unsigned long long g_enum_prof[16384][3];  // development build: per wave [start, end (100-MHz ticks), steps]
#endif
#ifndef MTG_ENUM_WAVES_PER_SIMD
#define MTG_ENUM_WAVES_PER_SIMD 1  // (experiments: 5 makes the compiler keep the registers within a fifth wave per SIMD)
#endif
// W8: the blocks are in the 8:8 format (k <= 255). PRUNE (needs W8): successors are tested against distance + weight + lower bound,
// and the sources come from the launch's list of sources that can reach an in-node at all (act_index / act_node, counters[C_ACTIVE]).
//
// Where a lane's entries live (round 6; before: S1 stack rows and H1 hit rows per lane in a slot-major first tier, the rest in an
// extension block, every one of a step's 21 potential stores unconditional -- to the next free slot, counted or not -- and computing
// its word from its row: tier test, two address forms, a select, the byte offset: 8 vector instructions per store in a loop whose
// time is its vector instructions). Now every lane has a HOME BLOCK of HOME entries (stack rows from its bottom, hits from its top,
// like an extension block) and at most one extension block; a step's stores all go to ONE of them -- the home block while the step's
// entries fit it, the extension block from the step on that outgrows it -- so that a store is `under the lanes' condition: write at
// the pointer, move the pointer by one word': one vector instruction beside the write. The stack is then split at row s0 (rows below
// it in the home block, rows from it in the extension block; a step that starts below s0 lowers it: the rows above are dead), the
// hits at row h0 (fixed when the extension block arrives); only the pop and the reads of a finished list select between the two.
template <int WPB, int HOME, int NB, bool QUAD, bool W8, bool PRUNE>
__global__ __launch_bounds__(WPB * 64, MTG_ENUM_WAVES_PER_SIMD) void sssp_enum_kernel(SsspArgs a) {
    static_assert(W8 || !PRUNE, "the lower bounds live in the 8:8 format");
    static_assert(NB >= 1 && NB <= 64 && HOME >= 4, "pool free mask is one 64-bit word; lists of up to four are read from fixed rows");
    constexpr int BE = ENUM_BE, BS = ENUM_BS;
    static_assert(BS % 2 == 1, "odd block stride");
    constexpr uint32_t T1 = 64u * HOME;                         // words of the home blocks
    constexpr uint32_t KRING = MTG_ENUM_KRING;                  // keys on their way to the pool
    static_assert((KRING & (KRING - 1)) == 0 && KRING >= 64 && KRING <= ENUM_POOL_CHUNK, "the ring is indexed by the chunk offset modulo its size");
    constexpr uint32_t TSLOTS = MTG_ENUM_TSLOTS;
    static_assert(TSLOTS >= 2 && (TSLOTS & (TSLOTS - 1)) == 0, "the table is indexed by the chunk number modulo its size");
    constexpr uint32_t WAVE_WORDS = T1 + (uint32_t)NB * BS + KRING + TSLOTS * 64u;  // + pool + output ring + the records of TSLOTS chunks of sources
    constexpr uint32_t IDLE_DIST = 0xFFFF0000u;                 // distance of a lane without a source: nothing is within the bound from there
    // stack entry: node | (distance | own-flag-still-open << 16) << 32; hit entry = candidate key: node | distance << 32
    __shared__ unsigned long long s_mem[WPB][WAVE_WORDS];
    __shared__ uint32_t s_cnt[WPB];
    __shared__ WaveOvfBuf s_ovf[WPB];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t K1 = a.K1;
    // every LDS position below is a BYTE offset into s_mem (the array's own address is the instruction's immediate offset)
    auto lds = [&](uint32_t off) -> unsigned long long & {
        return *reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(&s_mem[0][0]) + off);
    };
    const uint32_t wave_off = (uint32_t)__builtin_amdgcn_readfirstlane(wv) * WAVE_WORDS * 8u;  // (known uniform to the compiler: the pool's free mask stays scalar)
    // home blocks: row-major over the wave (word r of lane l at r * 64 + l: whatever rows the lanes work in, they never share a bank --
    // lane-major blocks measured the loss: random rows, three or four lanes per bank pair); extension blocks: BS consecutive words
    constexpr uint32_t HOME_SHIFT = 9, EXT_SHIFT = 3;          // log2 of the bytes from one word of a block to the next
    constexpr uint32_t HOME_STEP = 1u << HOME_SHIFT, EXT_STEP = 1u << EXT_SHIFT;
    const uint32_t home_b = wave_off + (uint32_t)lane * 8u, home_t = home_b + (HOME - 1) * HOME_STEP;  // first stack word / first hit word
    const uint32_t pool_off = wave_off + T1 * 8u;
    const uint32_t ring_off = pool_off + (uint32_t)NB * (BS * 8u);
    const uint32_t tab_off = ring_off + KRING * 8u;  // records: [chunk parity][position in the chunk]

    unsigned long long free_mask = NB == 64 ? ~0ull : ((1ull << (NB & 63)) - 1ull);  // wave-uniform: free extension blocks
    unsigned long long pool_base = 0;                                                // wave-uniform
    // Keys leave through a ring in LDS: a finished list is copied to the ring at its offset in the wave's pool chunk, and the wave writes
    // the ring to the pool in whole rows of 64 keys (512 bytes, one instruction, every line written once and in full) -- before, every
    // finished lane stored its own keys (4 + the longest list's length store instructions per step, each a few partial lines).
    uint32_t staged = ENUM_POOL_CHUNK, flushed = ENUM_POOL_CHUNK;  // (no chunk yet) wave-uniform chunk offsets: keys below `staged' are in the ring or the pool, below `flushed' in the pool
    unsigned long long fix_next[3] = {0, 0, 0}, fix_end[3] = {0, 0, 0};             // wave-uniform: the wave's open work-list chunk per length class
    uint32_t fix_total[3] = {0, 0, 0};                                              // wave-uniform: entries appended per class
    uint32_t n_overflow = 0;                                                         // wave-uniform
    if (lane == 0) s_cnt[wv] = ENUM_POOL_CHUNK;  // position inside the wave's pool chunk (no chunk yet)
#ifdef MTG_ENUM_STATS
    uint32_t st_starved = 0, st_full = 0, st_budget = 0, st_steps = 0, st_lanes = 0;
    const unsigned long long st_t0 = wall_clock64();
#endif

    // ---- sources: chunks of 64 in a STATIC stride (chunk c belongs to wave c mod n_waves), held in registers: `ids` is the chunk
    // being handed out, `ahead` the wave's next one (loaded a whole chunk before it is needed). A lane that needs a source gets
    // the next unused one by a cross-lane permute: no atomic, no memory round trip, no LDS. ----
    const unsigned long long n_items = PRUNE ? a.counters[C_ACTIVE] : a.n_items;
    const unsigned long long act_first = PRUNE ? a.counters[C_ACT_BEGIN] : 0ull;
    const unsigned long long n_waves = (unsigned long long)gridDim.x * WPB;
    unsigned long long next_chunk = (unsigned long long)blockIdx.x * WPB + wv;  // chunk that `ahead` will hold (wave-uniform)
    uint32_t cur_base = 0, cur_len = 0, cur_pos = 0, nxt_base = 0, nxt_len = 0;  // wave-uniform
    uint32_t ids = 0, ahead = 0;
    uint32_t ids_item = 0, ahead_item = 0;  // PRUNE: the sources' indices relative to src_begin (without a list they are cur_base + position)
    // (measured and dropped, twice: the last quarter of the chunks handed out by a global counter, requested two chunks ahead so that the
    // atomic is never waited for -- round 3: 7 % slower at 2^27 and no better lane utilisation at 2^24. Round 6, because the waves of the
    // static deal end 15 % apart (2^27: the first after 994 us, the median after 1166, the last after 1340): the last quarter / eighth /
    // sixteenth of the chunks in PAIRS from one cursor -- a same-address atomic is served every ~7 ns, one per chunk would be 126 M/s --,
    // the first pair asked for one chunk before a wave's last own one: + 0.10 / + 0.07 / + 0.06 ms on the stage)
    auto prefetch_chunk = [&]() {
        const unsigned long long lo = next_chunk * 64;
        nxt_base = (uint32_t)lo;
        nxt_len = lo >= n_items ? 0u : (n_items - lo < 64 ? (uint32_t)(n_items - lo) : 64u);
        if constexpr (PRUNE) {
            ahead = (uint32_t)lane < nxt_len ? a.act_node[act_first + lo + lane] : 0u;
            ahead_item = (uint32_t)lane < nxt_len ? (uint32_t)(a.act_index[act_first + lo + lane] - a.src_begin) : 0u;
        } else {
            ahead = (uint32_t)lane < nxt_len ? a.sources[a.src_begin + lo + lane] : 0u;
        }
        next_chunk += n_waves;
    };
    prefetch_chunk();
    ids = ahead; ids_item = ahead_item; cur_base = nxt_base; cur_len = nxt_len;
    prefetch_chunk();
    bool exhausted = cur_len == 0;
    // (the chunk in `ids' is number cur_seq of the wave; a lane remembers its source's chunk and position there, `tag' = chunk << 6 |
    // position: where its record waits, see below. When `ids' moves on, the records of the chunk before the one that leaves `ids' are
    // due: `evict_*' says so to the end of the step)
    uint32_t cur_seq = 0;                                // wave-uniform
    uint32_t hist_items[TSLOTS - 1], evict_items = 0;    // per lane = per position: the sources (indices relative to src_begin) of chunks cur_seq - 1, - 2, ... / of the chunk that is due
#pragma unroll
    for (uint32_t q = 0; q + 1 < TSLOTS; q++) hist_items[q] = 0;
    bool evict_pending = false;                          // wave-uniform
    auto take_source = [&](bool want, uint32_t &new_item, uint32_t &new_src, uint32_t &new_tag) -> bool {
        if (exhausted) return false;
        const unsigned long long need = __ballot(want);
        const uint32_t idx = cur_pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        const bool second = idx >= 64u;  // (only a full chunk has a successor: cur_len < 64 means nxt_len == 0)
        const bool got = want && (second ? idx - 64u < nxt_len : idx < cur_len);
        const int sel = (int)((idx & 63u) << 2);
        const uint32_t from_cur = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)ids), from_nxt = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)ahead);
        new_src = second ? from_nxt : from_cur;
        if constexpr (PRUNE) {
            const uint32_t item_cur = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)ids_item), item_nxt = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)ahead_item);
            new_item = second ? item_nxt : item_cur;
        } else {
            new_item = second ? nxt_base + idx - 64u : cur_base + idx;
        }
        new_tag = ((cur_seq + (second ? 1u : 0u)) << 6) | (idx & 63u);
        cur_pos += (uint32_t)__popcll(need);
        if (cur_pos >= 64u) {
            cur_pos -= 64u;
            evict_items = hist_items[TSLOTS - 2]; evict_pending = true;  // (the oldest chunk of the table: its place goes to chunk cur_seq + 1)
#pragma unroll
            for (uint32_t q = TSLOTS - 2; q > 0; q--) hist_items[q] = hist_items[q - 1];
            hist_items[0] = PRUNE ? ids_item : cur_base + (uint32_t)lane;
            cur_seq++;
            ids = ahead; ids_item = ahead_item; cur_base = nxt_base; cur_len = nxt_len;
            prefetch_chunk();  // in flight while the chunk that just became current is handed out
        }
        exhausted = cur_pos >= cur_len;
        return got;
    };

    // ---- per-lane search state ----
    bool active = false;
    uint32_t sp = 0, nhit = 0;        // stack entries / hits of the lane's source
    uint32_t s0 = 0, h0 = 0;          // first stack row / first hit of the lane's store block (0 while that is its home block)
    uint32_t cb = home_b, ct = home_t;  // the lane's store block: its first stack word and its first hit word (LDS byte offsets) ...
    uint32_t step = HOME_STEP, shift = HOME_SHIFT;  // ... and the bytes between two of its words
    uint32_t pops = 0, src_node = 0, item = 0, tag = 0;
    bool src_tgt = false;             // the source is an in-node itself: its own hits are taken out when it finishes (see hv[0])
    uint32_t cur_node = 0, cur_dist = IDLE_DIST;  // the node whose block is in b0..b3
    bool cur_chk = false;                         // its own in-node flag was already evaluated from its parent's block
    // The gather. QUAD = false: every lane loads the four quarters of its own block (four requests per lane to the same line).
    // QUAD = true: four lanes load one block together -- in load l member q of a group fetches quarter q of the block of the group's
    // member l -- and transpose the group's 4 x 4 quarters in registers when the data is needed. An instruction then touches
    // 16 lines instead of 64. Measured (tools/gather_bench_tlb.hip): with a 5.7-GB table (the 2^27 graph) dependent random
    // 64-byte gathers run at 18.8 G/s with four requests per lane and at 44 G/s either way of making it one request per lane
    // and line -- beyond ~4 GB every lane-request pays an address translation; below 3 GB both forms reach 51-55 G/s.
    uint4 b0 = {0, 0, 0, 0}, b1 = b0, b2 = b0, b3 = b0;  // as loaded; after arrive_block(): the block of cur_node
    // The four lanes of a gather group are the lanes of one COLUMN of the wave seen as 4 rows of 16 (lane, lane ^ 16, lane ^ 32,
    // lane ^ 48), not four neighbours: gfx950 exchanges the odd rows of one register with the even rows of another
    // (v_permlane16_swap) and the upper half of one with the lower half of another (v_permlane32_swap) in ONE instruction, so the
    // 4 x 4 transposition of the group's quarters is 16 instructions where the neighbour form needed 128 (two stages of DPP
    // quad_perm moves + selects per word: a fifth of the level's instructions, in a loop bound by instruction issue). The memory
    // system does not care which four lanes share a line (tools/gather_bench_rows.hip: 48 G gathers/s either way at 5.7 GB).
    auto swap_halves = [](uint32_t &x, uint32_t &y) {  // x[lanes 32..63] <-> y[lanes 0..31]
        const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
        x = r[0]; y = r[1];
    };
    auto swap_rows = [](uint32_t &x, uint32_t &y) {  // x[rows 1, 3] <-> y[rows 0, 2]
        const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
        x = r[0]; y = r[1];
    };
    // (row r of the wave, register j) <- (row j, register r), within every column
    auto transpose_rows = [&](uint32_t &x0, uint32_t &x1, uint32_t &x2, uint32_t &x3) {
        swap_halves(x0, x2); swap_halves(x1, x3);
        swap_rows(x0, x1); swap_rows(x2, x3);
    };
    auto load_block = [&](bool need, uint32_t node) {
        if constexpr (!QUAD) {
            if (need) {
                const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs + node);
                b0 = rp[0]; b1 = rp[1]; b2 = rp[2]; b3 = rp[3];
            }
        } else {
            // (a lane without a next node asks for block 0: its group's loads are unconditional, what arrives is never looked at --
            // the distance of an idle lane puts everything beyond the bound)
            const uint32_t want = need ? node : 0u;
            const uint32_t q = (uint32_t)lane >> 4;
            uint32_t n0 = want, n1 = want, n2 = want, n3 = want;  // n_j <- what row j of this column wants
            transpose_rows(n0, n1, n2, n3);
            // (non-temporal loads for the gathers: + 3 % on the stage)
            b0 = reinterpret_cast<const uint4 *>(a.recs + n0)[q];
            b1 = reinterpret_cast<const uint4 *>(a.recs + n1)[q];
            b2 = reinterpret_cast<const uint4 *>(a.recs + n2)[q];
            b3 = reinterpret_cast<const uint4 *>(a.recs + n3)[q];
        }
    };
    auto arrive_block = [&]() {  // QUAD: b_j of row r is quarter r of the block of row j; afterwards b_q of row r is quarter q of its own
        if constexpr (QUAD) {
            transpose_rows(b0.x, b1.x, b2.x, b3.x);
            transpose_rows(b0.y, b1.y, b2.y, b3.y);
            transpose_rows(b0.z, b1.z, b2.z, b3.z);
            transpose_rows(b0.w, b1.w, b2.w, b3.w);
        }
    };
    auto ring = [&](uint32_t chunk_offset) -> unsigned long long & { return lds(ring_off + ((chunk_offset & (KRING - 1u)) << 3)); };
    auto flush_keys = [&](uint32_t lo, uint32_t hi) {  // chunk offsets [lo, hi) from the ring to the pool
        for (uint32_t j = lo + (uint32_t)lane; j < hi; j += 64u) {
            const unsigned long long pos = pool_base + j;
#ifndef MTG_EXP_NO_POOL_STORES  // (timing experiments only)
            // (results leave with non-temporal stores: nothing in this kernel reads them again, and they do not push the blocks' lines out of
            // the caches -- keys: - 0.025 ms on the stage at 2^27, the lists' places: - 0.027 ms)
            if (pos < a.pool_cap) __builtin_nontemporal_store(ring(j), &a.pool[pos]);  // (pool too small: the host retries with a larger one)
#endif
        }
    };
    // A finished source's (start, count) -- and its entry in the post-pass work list -- wait in LDS as a record: start << 9 |
    // needs-the-post-pass << 8 | count (0xFF: handed to the cascade), in a table [chunk parity][position in the chunk] that holds the
    // wave's current chunk of sources and the TSLOTS - 1 before. When the wave moves on to its next chunk, the records of the oldest of them
    // leave TOGETHER, lane = position = the sources' order: 64 consecutive searched sources span ~220 source indices, i.e. ~14 lines of
    // cand_count and ~28 of cand_start -- where every record stored on its own touches two lines of its own. A source that outlives
    // TSLOTS chunks stores its record itself.
    // What the result stores cost the level (round 6, G-csr 2^27, under the profiler): 0.55 of 1.30 ms -- the search alone runs in
    // 0.75 ms; keys 0.18, (start, count) 0.25-0.31, work list 0.06-0.10. What did NOT change that: issuing the stores after the next
    // step's wait instead of behind the gather; dense arrays over the searched sources plus an expansion pass; bursts of 96 records
    // in finishing order or sorted by source (a 64-lane network); runs of consecutive chunks per wave (the lines of a run then meet
    // in L2). What did: fewer store instructions (the keys' ring) and fewer distinct LINES per store instruction (made-up addresses,
    // 36 lines per 64 records: -0.12 ms, 12 lines: -0.21 ms) -- which is what the table is for. tools/kstats_multi.sh, -DMTG_EXP_*.
    auto table = [&](uint32_t t) -> unsigned long long & { return lds(tab_off + ((t & (TSLOTS * 64u - 1u)) << 3)); };  // (t = tag: chunk << 6 | position)
#pragma unroll
    for (uint32_t q = 0; q < TSLOTS; q++) table(q * 64u + (uint32_t)lane) = 0ull;
    auto emit_records = [&](unsigned long long v, uint32_t r_item) {  // v != 0: this lane has a record, of source r_item
        const bool have = v != 0ull;
        const uint32_t r_c = (uint32_t)v & 0xFFu;
        const bool r_fix = ((uint32_t)v & 0x100u) != 0u;
#ifndef MTG_EXP_NO_START_COUNT
        if (have && r_c != 0xFFu) {
            __builtin_nontemporal_store((unsigned long long)(v >> 9), &a.cand_start[r_item]);
            __builtin_nontemporal_store(r_c, &a.cand_count[r_item]);
        } else if (have) a.cand_count[r_item] = CAND_OVERFLOW;
#endif
        // post-pass work list: a wave fills one chunk per length class at a time (a wave of the post-pass then sorts lists of
        // similar length with a network of that size)
        auto append_fix = [&](bool f, int cls) {
            const unsigned long long fm = __ballot(f);
            if (!fm) return;
            const uint32_t nf = (uint32_t)__popcll(fm);
            if (fix_next[cls] + nf > fix_end[cls]) {  // the rest of the old chunk is marked unused
                for (unsigned long long t = fix_next[cls] + lane; t < fix_end[cls]; t += 64) a.fix_list[t] = FIX_NONE;
                unsigned long long f0 = 0;
                if (lane == 0) {
                    f0 = atomicAdd(&a.counters[C_FIX], (unsigned long long)ENUM_FIX_CHUNK);
                    a.fix_list[f0] = FIX_CLASS_TAG | (uint32_t)cls;
                }
                fix_next[cls] = uniform_u64(f0) + 1;
                fix_end[cls] = fix_next[cls] - 1 + ENUM_FIX_CHUNK;
            }
            if (f) {
                const unsigned long long slot = fix_next[cls] + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                a.fix_list[slot] = r_item;  // (read again by the post-pass right after the kernel: ordinary stores; non-temporal ones measured the same)
                a.fix_val[slot] = ((v >> 9) << 8) | r_c;
            }
            fix_next[cls] += nf;
            fix_total[cls] += nf;
        };
#if !defined(MTG_EXP_NO_START_COUNT) && !defined(MTG_EXP_NO_FIX)  // (without the lists' places the post-pass must get no work)
        const bool f = have && r_fix && r_c != 0xFFu;
        append_fix(f && r_c <= 8, 0);
        append_fix(f && r_c > 8 && r_c <= 16, 1);
        append_fix(f && r_c > 16, 2);
#else
        (void)append_fix; (void)r_fix;
#endif
    };
    auto flush_chunk_records = [&](uint32_t chunk, uint32_t items) {  // the records of one chunk: lane = position
        const uint32_t t = (chunk << 6) | (uint32_t)lane;
        const unsigned long long v = table(t);
        if (!__any(v != 0ull)) return;
        table(t) = 0ull;
        emit_records(v, items);
    };
    // hit r of the lane's source: in the home block below h0, in the store block from there
    auto hit_at = [&](uint32_t r) -> uint32_t { return r >= h0 ? ct - ((r - h0) << shift) : home_t - (r << HOME_SHIFT); };
    {
        uint32_t ni = 0, ns = 0;
        uint32_t nt = 0;
        if (take_source(true, ni, ns, nt)) {
            item = ni; src_node = ns; cur_node = ns; cur_dist = 0; tag = nt;
            active = true;
        }
        load_block(active, cur_node);
    }
    while (__any(active)) {
        // (the quad form runs the first half of a step -- transpose, decode, stores, pop, until its gather has left -- at a raised wave
        // priority: 2.5 % faster at 2^27; the per-lane form measured 10 % slower with it at 2^24)
        if constexpr (QUAD) __builtin_amdgcn_s_setprio(2);
        arrive_block();
        // ---- the block that arrived: which hits and successors count ----
        const uint32_t u = cur_node, d = cur_dist;
        const uint32_t meta = b1.z;
        const uint32_t nb[4] = {b0.x, b0.y, b0.z, b0.w};
        const uint32_t gn[GSLOTS] = {b1.w, b2.x, b2.y, b2.z, b2.w, b3.x};
        // weight of the edge to a child / of the path to a grandchild, and (PRUNE) that weight + the lower bound of what lies beyond it
        uint32_t wc[4], wg[GSLOTS], lc[4], lg[GSLOTS];
        {
            const uint32_t cs[4] = {b1.x & 0xFFFFu, b1.x >> 16, b1.y & 0xFFFFu, b1.y >> 16};
            const uint32_t gs[GSLOTS] = {b3.y & 0xFFFFu, b3.y >> 16, b3.z & 0xFFFFu, b3.z >> 16, b3.w & 0xFFFFu, b3.w >> 16};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                wc[j] = W8 ? (cs[j] & 0xFFu) : cs[j];
                lc[j] = PRUNE ? (cs[j] >> 8) : wc[j];
            }
#pragma unroll
            for (int t = 0; t < GSLOTS; t++) {
                wg[t] = W8 ? (gs[t] & 0xFFu) : gs[t];
                lg[t] = PRUNE ? (gs[t] >> 8) : wg[t];
            }
        }
        // what is left of the bound at this node (an idle lane: nothing; an unused slot holds 0xFFFF or 0xFF: beyond any bound)
        const int32_t left = active ? (int32_t)(K1 - d) : -1;
        const bool is_ext = active && (meta & ((uint32_t)F_EXT << 8));
        // PRUNE: a grandchild that is an in-node (cmeta bit 8 + t) is recorded from THIS block, like the children, and pushed -- with its
        // own flag done -- only if something lies beyond it (the high bytes carry weight + lb+): a quarter of the node visits of the
        // bench graph are in-nodes with nothing behind them within the bound, and their blocks are never gathered.
        // forbid_source_target (greedytigs/mod.rs:329) is NOT tested here: a hit names the source itself only if the source is an
        // in-node (then hv[0] is true at its root, distance 0), and such a source strikes its own node from its list when it finishes.
        // What a step costs is its VECTOR instructions (round 6, counters: 4 waves per SIMD keep its vector pipe busy all the time; the
        // kernel's time followed their number, not the scalar instructions'): a condition is one compare of a byte against `left' and a
        // bit test, a store happens under the lanes' condition (write, move the pointer: scalar mask handling + one vector instruction).
        bool hv[5], hg[GSLOTS], pv[4 + GSLOTS];
        hv[0] = active & !cur_chk & ((meta & ((uint32_t)F_TARGET << 8)) != 0u);
        src_tgt |= hv[0] & (d == 0u);
        uint32_t nh_f = nhit + (hv[0] ? 1u : 0u), sp_f = sp;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            hv[1 + j] = ((int32_t)wc[j] <= left) & ((meta & (0x10000u << j)) != 0u);
            pv[j] = ((int32_t)lc[j] <= left) & ((meta & (0x100000u << j)) != 0u);
            nh_f += hv[1 + j] ? 1u : 0u;
            sp_f += pv[j] ? 1u : 0u;
        }
#pragma unroll
        for (int t = 0; t < GSLOTS; t++) {
            pv[4 + t] = (int32_t)lg[t] <= left;
            sp_f += pv[4 + t] ? 1u : 0u;
            hg[t] = PRUNE & ((int32_t)wg[t] <= left) & ((meta & (0x1000000u << t)) != 0u);
            nh_f += hg[t] ? 1u : 0u;
        }
        if (__any(is_ext)) {  // spilled adjacency (more than 4 out-edges; never in a de Bruijn graph): count first
            if (is_ext) {
                const uint64_t ext_begin = ((uint64_t)b0.y << 32) | b0.x;
                for (uint32_t j = 0; j < b0.z; j++) sp_f += d + a.ext_w[ext_begin + j] <= K1 ? 1u : 0u;
            }
        }
        // ---- an extension block for the lanes whose entries outgrow their home block in this step ----
        const bool need_blk = cb == home_b && sp_f + nh_f > (uint32_t)HOME;
        unsigned long long nm = __ballot(need_blk);
        uint32_t blk_new = 0;
        while (nm && free_mask) {
            const int l = __builtin_ctzll(nm);
            const int bi = __builtin_ctzll(free_mask);
            nm &= nm - 1;
            free_mask &= free_mask - 1;
            blk_new = lane == l ? pool_off + (uint32_t)bi * (BS * 8u) : blk_new;
        }
        const bool got_blk = need_blk && blk_new != 0u;  // (pool_off > 0: the home blocks come first)
        cb = got_blk ? blk_new : cb;
        ct = got_blk ? blk_new + (BE - 1) * EXT_STEP : ct;
        step = got_blk ? EXT_STEP : step;
        shift = got_blk ? EXT_SHIFT : shift;
        s0 = got_blk ? sp : (s0 < sp ? s0 : sp);  // (a step that starts below the split lowers it: the rows above are dead)
        h0 = got_blk ? nhit : h0;
        pops += active ? 1u : 0u;
        // (a lane the pool had no block for, or whose block is full, hands its source to the cascade)
        const bool starved = need_blk && !got_blk;
        const bool ovf = active && (starved || (sp_f - s0) + (nh_f - h0) > (uint32_t)BE || pops > ENUM_POP_BUDGET);
#ifdef MTG_ENUM_STATS  // development build: why sources leave this level, how many steps run, how full the lanes are
        st_starved += (uint32_t)__popcll(__ballot(active && starved));
        st_full += (uint32_t)__popcll(__ballot(active && !starved && (sp_f - s0) + (nh_f - h0) > (uint32_t)BE));
        st_budget += (uint32_t)__popcll(__ballot(active && pops > ENUM_POP_BUDGET));
        st_steps += 1;
        st_lanes += (uint32_t)__popcll(__ballot(active));
#endif

        // ---- hits and successors into LDS (a lane that leaves the level stores nothing) ----
        uint32_t spt = cb + ((sp - s0) << shift);    // the next free stack word ...
        uint32_t hpt = ct - ((nhit - h0) << shift);  // ... and the next free hit word of the store block
        auto put = [&](uint32_t at, uint32_t node, uint32_t hi) {  // (two 32-bit halves from wherever they are: no register pairing)
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(&s_mem[0][0]) + at) = node;
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(&s_mem[0][0]) + at + 4u) = hi;
        };
        if (!ovf) {
            if (hv[0]) { put(hpt, u, d); hpt -= step; }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (hv[1 + j]) { put(hpt, nb[j], d + wc[j]); hpt -= step; }
            if constexpr (PRUNE) {
#pragma unroll
                for (int t = 0; t < GSLOTS; t++)
                    if (hg[t]) { put(hpt, gn[t], d + wg[t]); hpt -= step; }
            }
            if (__any(is_ext)) {
                if (is_ext) {
                    const uint64_t ext_begin = ((uint64_t)b0.y << 32) | b0.x;
                    for (uint32_t j = 0; j < b0.z; j++) {
                        const uint32_t nd = d + a.ext_w[ext_begin + j];
                        if (nd <= K1) { put(spt, a.ext_col[ext_begin + j], nd | 0x10000u); spt += step; }  // (its own flag is still open)
                    }
                }
            }
#pragma unroll
            for (int t = GSLOTS - 1; t >= 0; t--)
                if (pv[4 + t]) { put(spt, gn[t], (d + wg[t]) | (PRUNE ? 0u : 0x10000u)); spt += step; }  // (PRUNE: its flag was evaluated above)
#pragma unroll
            for (int j = 3; j >= 0; j--)  // (children come off the stack before grandchildren: nearer nodes first)
                if (pv[j]) { put(spt, nb[j], d + wc[j]); spt += step; }
        }
        const bool go_on = active && !ovf && sp_f > 0;  // the next node comes off the stack
        // (the newest row is in the store block if that holds any row at all, else it is the home block's last one)
        const unsigned long long top = lds(sp_f > s0 ? cb + ((sp_f - s0 - 1u) << shift) : home_b + (((sp_f > 1u ? sp_f : 1u) - 1u) << HOME_SHIFT));
        sp = sp_f - (go_on ? 1u : 0u);
        nhit = nh_f;
        const bool fin = active && !ovf && !go_on;  // the source is finished

        // ---- lanes without a next node take a new source; every lane's next gather leaves now ----
        uint32_t new_item = 0, new_src = 0, new_tag = 0;
        const bool got_new = take_source(!go_on, new_item, new_src, new_tag);
        const uint32_t nx_node = go_on ? (uint32_t)top : new_src;
        load_block(go_on || got_new, nx_node);
        if constexpr (QUAD) __builtin_amdgcn_s_setprio(0);

        // ---- finished / overflowed sources write their result ----
        if (evict_pending) {  // the wave has moved on to chunk cur_seq: the records of chunk cur_seq - 2 leave, its part of the table is the new chunk's
            flush_chunk_records(cur_seq, evict_items);  // (chunk cur_seq - TSLOTS: the same part of the table)
            evict_pending = false;
        }
        const unsigned long long donemask = __ballot(fin || ovf);
        if (donemask) {
            uint32_t c = fin ? nhit : 0u;
            if (__any(fin && src_tgt)) {  // a source that is an in-node strikes itself from its list (forbid_source_target)
                if (fin && src_tgt) {
                    uint32_t w = 0;
                    for (uint32_t r = 0; r < c; r++) {
                        const unsigned long long key = lds(hit_at(r));
                        if ((uint32_t)key != src_node) { lds(hit_at(w)) = key; w++; }
                    }
                    c = w;
                }
            }
            // lists of up to four keys are put in Dijkstra order in registers; longer ones (and repeated nodes) go to the post-pass
            unsigned long long k0 = ~0ull, k1 = ~0ull, k2 = ~0ull, k3 = ~0ull;
            if (c > 0) k0 = lds(hit_at(0u));
            if (c > 1) k1 = lds(hit_at(1u));
            if (c > 2) k2 = lds(hit_at(2u));
            if (c > 3) k3 = lds(hit_at(3u));
            bool fix = c > 4;
            if (__any(c >= 2 && c <= 4)) {  // (a longer list is sorted as a whole by the post-pass: no need to run the network for it alone)
                auto cswap = [](unsigned long long &x, unsigned long long &y) {  // pop order: policy P1 (mtg_policy.h)
                    const bool xf = mtg_policy_pops_before(x, y);
                    const unsigned long long lo = xf ? x : y, hi = xf ? y : x;
                    x = lo; y = hi;
                };
                cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);  // (longer lists are sorted again anyway)
                const uint32_t n0 = (uint32_t)k0, n1 = (uint32_t)k1, n2 = (uint32_t)k2, n3 = (uint32_t)k3;
                const bool dup = (c >= 2 && c <= 4) && (n0 == n1 || (c > 2 && (n0 == n2 || n1 == n2)) || (c > 3 && (n0 == n3 || n1 == n3 || n2 == n3)));
                if (dup && c == 2) c = 1;  // the same node along two paths: the shorter distance stays
                else fix |= dup;
            }
            uint32_t off = c ? atomicAdd(&s_cnt[wv], c) : 0u;  // any order: (start, count) index the content
            // (read as it is in LDS now: lane 0's own view of the counter is its reset below, not the other lanes' additions)
            auto chunk_fill = [&]() -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile uint32_t *>(&s_cnt[wv])); };
            uint32_t total = chunk_fill();
            if (total > ENUM_POOL_CHUNK) {  // chunk used up: this step's lists go to a new one (what the old one still has in the ring leaves now)
                flush_keys(flushed, staged);
                unsigned long long p0 = 0;
                if (lane == 0) {
                    p0 = atomicAdd(&a.counters[C_POOL], (unsigned long long)ENUM_POOL_CHUNK);
                    s_cnt[wv] = 0;
                }
                pool_base = uniform_u64(p0);
                off = c ? atomicAdd(&s_cnt[wv], c) : 0u;
                total = chunk_fill();
                staged = 0; flushed = 0;
            }
            // this step's keys are the chunk offsets [staged, total)
            if (total - flushed > KRING) { flush_keys(flushed, staged); flushed = staged; }  // (room for them: the ring's partial row leaves early)
            const unsigned long long pos = pool_base + off;
#ifdef MTG_DBG_RING_OFF
            if (false) {
#else
            if (total - flushed <= KRING) {
#endif
                if (c > 0) ring(off) = k0;
                if (c > 1) ring(off + 1u) = k1;
                if (c > 2) ring(off + 2u) = k2;
                if (c > 3) ring(off + 3u) = k3;
                for (uint32_t r = 4; __any(r < c); r++)
                    if (r < c) ring(off + r) = lds(hit_at(r));
                staged = total;
#ifdef MTG_DBG_RING_ALL
                const uint32_t rows = staged;
#else
                const uint32_t rows = staged & ~63u;
#endif
                if (rows > flushed) { flush_keys(flushed, rows); flushed = rows; }
            } else {  // more keys in one step than the ring holds (at most 64 lists of HOME + BE keys): straight to the pool
                const bool room = pos + c <= a.pool_cap;
#ifndef MTG_EXP_NO_POOL_STORES
                if (c > 0 && room) a.pool[pos] = k0;
                if (c > 1 && room) a.pool[pos + 1] = k1;
                if (c > 2 && room) a.pool[pos + 2] = k2;
                if (c > 3 && room) a.pool[pos + 3] = k3;
                for (uint32_t r = 4; __any(r < c); r++)
                    if (r < c && room) a.pool[pos + r] = lds(hit_at(r));
#endif
                staged = total; flushed = total;
            }
            // (7 of 10 sources have no candidate: their counts are zeroed by one streaming pass before the launch, their starts are
            // never read: no record for them)
            {  // the list's place (or: handed to the cascade) as a record in its chunk's part of the table; a straggler stores it now
                const bool rec = (fin && c) || ovf;
                const unsigned long long v = rec ? (ovf ? 0xFFull : ((pos << 9) | (fix ? 0x100ull : 0ull) | c)) : 0ull;
                const bool resident = (tag >> 6) + TSLOTS > cur_seq;  // (one of the chunks cur_seq - TSLOTS + 1 .. cur_seq)
#ifdef MTG_ENUM_STATS
                st_budget += (uint32_t)__popcll(__ballot(rec && !resident));  // (development build: counted with the step budget's)
#endif
                if (rec && resident) table(tag) = v;
                if (__any(rec && !resident)) emit_records(resident ? 0ull : v, item);
            }
            unsigned long long rel = __ballot((fin || ovf) && cb != home_b);  // extension blocks go back to the pool
            while (rel) {
                const int l = __builtin_ctzll(rel);
                rel &= rel - 1;
                free_mask |= 1ull << ((((uint32_t)__builtin_amdgcn_readlane((int)cb, l) - pool_off) / (uint32_t)(BS * 8)) & 63u);
            }
            wave_ovf_push(s_ovf[wv], n_overflow, ovf, (uint32_t)(a.src_begin + item), a, lane);
        }

        // ---- commit: every lane moves to its next block ----
        if (go_on) {
            cur_node = nx_node;
            cur_dist = (uint32_t)(top >> 32) & 0xFFFFu;
            cur_chk = ((uint32_t)(top >> 32) & 0x10000u) == 0u;
        } else {
            sp = 0; nhit = 0; pops = 0; s0 = 0; h0 = 0;
            cb = home_b; ct = home_t; step = HOME_STEP; shift = HOME_SHIFT;
            cur_chk = false;
            src_tgt = false;
            active = got_new;
            item = new_item; src_node = new_src; cur_node = new_src; tag = new_tag;
            cur_dist = got_new ? 0u : IDLE_DIST;
        }
    }
    flush_keys(flushed, staged);
    if (evict_pending) flush_chunk_records(cur_seq, evict_items);
#pragma unroll
    for (uint32_t q = 0; q + 1 < TSLOTS; q++) flush_chunk_records(cur_seq - 1u - q, hist_items[q]);  // (chunk numbers modulo TSLOTS: before the first chunks these parts are empty)
    flush_chunk_records(cur_seq, PRUNE ? ids_item : cur_base + (uint32_t)lane);
    for (int cls = 0; cls < 3; cls++) {
        for (unsigned long long t = fix_next[cls] + lane; t < fix_end[cls]; t += 64) a.fix_list[t] = FIX_NONE;
        if (lane == 0 && fix_total[cls]) atomicAdd(&a.counters[C_FIX_CLASS0 + cls], (unsigned long long)fix_total[cls]);
    }
    wave_ovf_flush(s_ovf[wv], n_overflow, a, lane);
#ifdef MTG_ENUM_STATS
    if (lane == 0) {
        const uint32_t wid = blockIdx.x * WPB + wv;
        if (wid < 16384) { g_enum_prof[wid][0] = st_t0; g_enum_prof[wid][1] = wall_clock64(); g_enum_prof[wid][2] = st_steps; }
        atomicAdd(&a.counters[C_SETTLED], (unsigned long long)st_steps);
        atomicAdd(&a.counters[C_RELAXED], (unsigned long long)st_lanes);
        atomicAdd(&a.counters[C_EMITTED], (unsigned long long)st_starved);
        atomicAdd(&a.counters[C_ATTEMPTS], (unsigned long long)st_full);
        atomicAdd(&a.counters[C_PUSHES], (unsigned long long)st_budget);
    }
#endif
}

// Post-pass of the enumeration level: a source's hits arrive in discovery order and may name a node more than once (one
// hit per path). Keep the smallest distance per node and order by (distance, node) -- what Dijkstra's pop order gives
// (SURVEY App. A.1). The level itself puts lists of up to four keys in order; what it leaves here are the longer lists (5.8 % of
// the bench graph's sources have 5-8 candidates, 2.8 % 9-16, 0.4 % more: together 58 % of all keys) and the rare short list with a
// repeated node. What was measured on the way (2^27 bench graph): one thread per list with an insertion sort in LDS
// (round 2) 0.61 ms -- dependent LDS round trips and divergent loop control; one thread per list with the keys in registers and a
// sorting network 0.15 + 0.23 ms for the lists of <= 8 / <= 16 keys, bound by the number of memory requests (every lane reads and
// writes its own list), and a 32-slot network is 40 KB of straight-line code whose first pass runs at the latency of
// instruction-cache misses (0.26 ms for a handful of lists); the form below 0.13 + 0.12 + 0.03 ms.
// The level's work list is chunked (slot 0 of a chunk names its length class, unused slots hold FIX_NONE); this pass makes it dense:
// class 0 first, then class 1, then class 2 (the level counted the entries per class), so that the sorting kernels below run over
// plain ranges.
__global__ __launch_bounds__(256) void fix_compact_kernel(const uint32_t *fix_list, const unsigned long long *fix_val, unsigned long long *counters,
                                                          uint32_t *dense, unsigned long long *dense_val) {
    __shared__ uint32_t s_cnt[3], s_off[3];
    __shared__ unsigned long long s_base[3];
    const unsigned long long n_slots = counters[C_FIX];
    const unsigned long long off[3] = {0, counters[C_FIX_CLASS0], counters[C_FIX_CLASS0] + counters[C_FIX_CLASS0 + 1]};
    // a workgroup takes a contiguous, chunk-aligned slice of the slots: it counts its entries per class, reserves its dense ranges with
    // ONE global atomic per class (same-address atomics cost ~7 ns each: one per wave made this pass 0.64 ms at 2^27), then copies
    const unsigned long long n_chunks = (n_slots + ENUM_FIX_CHUNK - 1) / ENUM_FIX_CHUNK;
    const unsigned long long per_block = (n_chunks + gridDim.x - 1) / gridDim.x * ENUM_FIX_CHUNK;
    const unsigned long long lo = (unsigned long long)blockIdx.x * per_block, hi = lo + per_block < n_slots ? lo + per_block : n_slots;
    if (threadIdx.x < 3) { s_cnt[threadIdx.x] = 0; s_off[threadIdx.x] = 0; }
    __syncthreads();
    for (unsigned long long s0 = lo; s0 < hi; s0 += 256) {  // (a wave stays inside one chunk: one class)
        const unsigned long long sl = s0 + threadIdx.x;
        const uint32_t e = sl < hi && (sl % ENUM_FIX_CHUNK) != 0 ? fix_list[sl] : FIX_NONE;
        const unsigned long long m = __ballot(e != FIX_NONE);
        if (m && (threadIdx.x & 63) == 0) atomicAdd(&s_cnt[fix_list[sl / ENUM_FIX_CHUNK * ENUM_FIX_CHUNK] & 3u], (uint32_t)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = off[threadIdx.x] + atomicAdd(&counters[C_FIX_CURSOR0 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
    for (unsigned long long s0 = lo; s0 < hi; s0 += 256) {
        const unsigned long long sl = s0 + threadIdx.x;
        const uint32_t e = sl < hi && (sl % ENUM_FIX_CHUNK) != 0 ? fix_list[sl] : FIX_NONE;
        const unsigned long long m = __ballot(e != FIX_NONE);
        if (!m) continue;
        const uint32_t cls = fix_list[sl / ENUM_FIX_CHUNK * ENUM_FIX_CHUNK] & 3u;
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&s_off[cls], (uint32_t)__popcll(m));
        base = __shfl(base, 0);
        if (e != FIX_NONE) {  // (the list's place, start << 8 | count, travels with its entry)
            const unsigned long long to = s_base[cls] + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            dense[to] = e;
            dense_val[to] = fix_val[sl];
        }
    }
}

// Lane-parallel form: G lanes per list, one key per lane -- coalesced reads and writes (a list is one or two memory requests instead
// of one per key or key pair), bitonic sort across the lanes, then each key looks at the keys before it for its node.
template <int G, int CLS, int BLOCK>
__device__ __forceinline__ void sort_lists_class(unsigned long long *pool, uint64_t pool_cap, uint32_t *cand_count, const uint32_t *dense,
                                                 const unsigned long long *dense_val, const unsigned long long *counters, uint32_t block,
                                                 uint32_t n_blocks) {
    constexpr uint32_t LPB = BLOCK / G;  // lists per workgroup pass
    unsigned long long begin = 0;
    for (int k = 0; k < CLS; k++) begin += counters[C_FIX_CLASS0 + k];
    const unsigned long long n = counters[C_FIX_CLASS0 + CLS];
    const uint32_t g = threadIdx.x % G;
    // (the list's place comes with its entry: no look-up at the source's index -- those were two of the pass's three scattered reads
    // per list: 0.30 -> 0.23 ms at 2^27, the compaction 0.045 -> 0.065 ms. Measured and dropped: no compaction at all, a workgroup
    // per work-list chunk -- most chunks are the half-filled last ones of their wave and class: 0.42 ms)
    // A pass is a chain of two dependent reads (the place, then the keys there) before its shuffles: the place of the pass after the
    // next and the keys of the next pass are requested before this pass's keys are sorted (0.246 -> 0.217 ms at 2^27; three passes
    // ahead: the same, six: 0.234).
    const unsigned long long stride = (unsigned long long)n_blocks * LPB, first = (unsigned long long)block * LPB + threadIdx.x / G;
    auto place_of = [&](unsigned long long l) -> unsigned long long { return l < n ? dense_val[begin + l] : 0ull; };
    auto count_of = [&](unsigned long long place) -> uint32_t {
        const uint32_t c = (uint32_t)place & 0xFFu;
        return (c < 2 || c > (uint32_t)G || (place >> 8) + c > pool_cap) ? 0u : c;  // (pool too small: the host retries with a larger one)
    };
    unsigned long long place = place_of(first), place_next = place_of(first + stride);
    unsigned long long key = g < count_of(place) ? pool[(place >> 8) + g] : ~0ull;
    for (unsigned long long l0 = (unsigned long long)block * LPB; l0 < n; l0 += stride) {
        const unsigned long long l = l0 + threadIdx.x / G;
        const uint32_t c = count_of(place);
        const unsigned long long st = place >> 8;
        // (the requests of the passes to come)
        const unsigned long long place_after = place_of(l + 2 * stride);
        const unsigned long long key_next = g < count_of(place_next) ? pool[(place_next >> 8) + g] : ~0ull;
#pragma unroll
        for (int k = 2; k <= G; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const unsigned long long other = __shfl_xor(key, j);
                const bool take_min = ((g & (uint32_t)k) == 0) == ((g & (uint32_t)j) == 0);
                const bool kf = mtg_policy_pops_before(key, other);  // pop order: policy P1 (mtg_policy.h)
                const unsigned long long lo = kf ? key : other, hi = kf ? other : key;
                key = take_min ? lo : hi;
            }
        }
        const uint32_t node = (uint32_t)key;
        bool dup = false;  // a node named twice: only its first (= smallest distance) occurrence stays
#pragma unroll
        for (int dlt = 1; dlt < G; dlt++) {
            const uint32_t before = __shfl_up(node, dlt, G);
            dup |= g >= (uint32_t)dlt && g < c && before == node;
        }
        const unsigned long long dm = __ballot(dup);
        if (dm) {  // (rare)
            const uint32_t lane = threadIdx.x & 63u;
            const unsigned long long group = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << (lane - g);
            const uint32_t before = (uint32_t)__popcll(dm & group & ((1ull << lane) - 1ull));
            if (g < c && !dup) pool[st + g - before] = key;
            if (g == 0 && (dm & group)) cand_count[dense[begin + l]] = c - (uint32_t)__popcll(dm & group);
        } else if (g < c) pool[st + g] = key;  // (non-temporal: the same)
        place = place_next; place_next = place_after; key = key_next;
    }
}
// the three length classes in ONE launch (a third of the grid each: their lists are disjoint, and three launches in a row spent more
// on their gaps and tails than on sorting; the grid split in proportion to the classes' workgroup passes instead: 0.246 -> 0.259 ms)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void sort_lists_kernel(unsigned long long *pool, uint64_t pool_cap, uint32_t *cand_count, const uint32_t *dense,
                                                           const unsigned long long *dense_val, const unsigned long long *counters) {
    const uint32_t per = gridDim.x / 3u, cls = blockIdx.x / per, block = blockIdx.x % per;
    if (cls == 0) sort_lists_class<8, 0, BLOCK>(pool, pool_cap, cand_count, dense, dense_val, counters, block, per);
    else if (cls == 1) sort_lists_class<16, 1, BLOCK>(pool, pool_cap, cand_count, dense, dense_val, counters, block, per);
    else if (cls == 2) sort_lists_class<32, 2, BLOCK>(pool, pool_cap, cand_count, dense, dense_val, counters, block, per);
}


typedef void (*sssp_fn)(SsspArgs);
struct LevelCfg {
    sssp_fn fn;
    sssp_fn fn_count;
    int block;
    int bsrc;
    int logh;
    int qcap;
    int scap;
    bool global_ws;
    std::string name() const {
        char b[96];
        std::snprintf(b, sizeof b, "sssp_kernel<%d,%d,%d,%d,%d%s>", block, logh, qcap, scap, bsrc, global_ws ? ",global" : "");
        return b;
    }
};

template <int BLOCK, int LOGH, int QCAP, int SCAP, int BSRC, bool GLOBAL_WS>
static LevelCfg make_cfg() {
    return LevelCfg{sssp_kernel<BLOCK, LOGH, QCAP, SCAP, BSRC, false, GLOBAL_WS>,
                    sssp_kernel<BLOCK, LOGH, QCAP, SCAP, BSRC, true, GLOBAL_WS>, BLOCK, BSRC, LOGH, QCAP, SCAP, GLOBAL_WS};
}

constexpr int ENUM_WPB = 4, ENUM_HOME = MTG_ENUM_HOME, ENUM_NB = MTG_ENUM_NB;  // 50 KB of LDS per workgroup: 3 workgroups = 12 waves per CU
constexpr int ENUM_MAX_HITS = ENUM_HOME + ENUM_BE;  // longest list the level can emit
static std::string enum_level_name(bool quad, bool prune) {
    char b[160];
    std::snprintf(b, sizeof b, "%ssssp_enum_kernel<%d,%d,%d,%s%s> + fix_compact_kernel + sort_lists_kernel", prune ? "active_range_kernel + " : "",
                  ENUM_WPB, ENUM_HOME, ENUM_NB, quad ? "quad" : "lane", prune ? ",pruned" : "");
    return b;
}

// Cascade of cooperative levels: 32 sources per workgroup, then 8, then 1 (128 KB LDS table), then a 32 MB
// global-memory table. A level re-runs the sources whose batch overflowed the previous level's tables.
static const int N_COOP_LEVELS = 5;
static LevelCfg coop_level(int i, bool after_enum) {
    switch (i) {  //                    BLOCK LOGH  QCAP  SCAP BSRC
        // 25 KB of LDS per workgroup -> 6 workgroups per CU: the level is bound by rounds x gather latency per batch
        // (4 us per round, ~10 rounds), so concurrency per CU is what counts; its heavy batches overflow the 1024-item
        // log early and are re-run by the next level
        case 0: return make_cfg<256, 11, 1024, 512, 32, false>();
        case 1:
            // what the enumeration level hands on are its ~1 % heaviest sources: 8 per workgroup in 29 KB of LDS (5 workgroups
            // per CU) measured 0.18 ms against 0.27 ms for 16 per workgroup in 58 KB
            if (after_enum) return make_cfg<256, 11, 2048, 512, 8, false>();
            return make_cfg<256, 12, 4096, 1024, 16, false>();
        case 2: return make_cfg<256, 13, 8192, 2048, 8, false>();
        case 3: return make_cfg<256, 14, 4096, 1024, 1, false>();
        default: return make_cfg<256, 22, 1 << 22, 1 << 21, 1, true>();
    }
}

float elapsed_ms(Device *d) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, d->ev0, d->ev1));
    return ms;
}

static bool enum_uses_quad_gathers(const Device *d) {
    // beyond ~3 GB of family blocks every lane-request pays an address translation: quad-cooperative gathers (see the kernel)
    return (d->plan & 3) == 2 || ((d->plan & 3) == 0 && d->V * sizeof(NodeBlock) > (3ull << 30));
}
static bool enum_prunes(const Device *d) { return d->w8 && !(d->plan & 4) && (d->plan & 3) != 1; }  // (plan 1 = the plain cascade: full balls)

// the searched sources of [src_begin, src_begin + n): the part of the classification's list (d_act_index / d_act_node) inside the range --
// first entry and number in counters[C_ACT_BEGIN] / counters[C_ACTIVE]; zeroes cand_count (a search only stores the count of a non-empty list)
static void launch_active_list(Device *d, hipStream_t st, const SsspArgs &args, uint64_t n) {
    hipLaunchKernelGGL(active_range_kernel, dim3(1), dim3(64), 0, st, d->d_act_index, d->d_act_total, args.src_begin, args.src_begin + n,
                       &args.counters[C_ACT_BEGIN], &args.counters[C_ACTIVE]);
    HIP_CHECK(hipMemsetAsync(args.cand_count, 0, n * sizeof(uint32_t), st));
    // (measured and dropped: zeroing cand_start too, so that the scattered 8-byte stores of the lists' starts land in lines a streaming pass has
    // just written -- + 0.05 ms, the pass itself)
    HIP_CHECK(hipGetLastError());
}

static void launch_enum(Device *d, hipStream_t st, SsspArgs args) {
    if (args.n_items == 0) return;
    const bool quad = enum_uses_quad_gathers(d), prune = enum_prunes(d);
    sssp_fn fn;
    if (prune) fn = quad ? sssp_enum_kernel<ENUM_WPB, ENUM_HOME, ENUM_NB, true, true, true> : sssp_enum_kernel<ENUM_WPB, ENUM_HOME, ENUM_NB, false, true, true>;
    else if (d->w8) fn = quad ? sssp_enum_kernel<ENUM_WPB, ENUM_HOME, ENUM_NB, true, true, false> : sssp_enum_kernel<ENUM_WPB, ENUM_HOME, ENUM_NB, false, true, false>;
    else fn = quad ? sssp_enum_kernel<ENUM_WPB, ENUM_HOME, ENUM_NB, true, false, false> : sssp_enum_kernel<ENUM_WPB, ENUM_HOME, ENUM_NB, false, false, false>;
    int occ = 1;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, ENUM_WPB * 64, 0));
    if (occ < 1) occ = 1;
    static const bool debug = std::getenv("MTG_DEBUG") != nullptr;
    if (debug) std::fprintf(stderr, "[mtg] enumeration level: %d workgroups of %d waves per CU\n", occ, ENUM_WPB);
    // (with pruning the number of searched sources stays on the GPU: the grid is sized for all of them, waves without a chunk leave at once)
    const uint64_t waves_needed = (args.n_items + 63) / 64;
    // (always the full grid: 8 waves per CU with twice the chunks per wave measured 20 % slower at 2^24, 65 % slower at 2^22)
    uint64_t grid = std::min<uint64_t>((uint64_t)d->n_cu * (uint64_t)occ, (waves_needed + ENUM_WPB - 1) / ENUM_WPB);
    grid = std::max<uint64_t>(grid, 1);
    // (plan + 8, tests: one workgroup -- its four waves take chunk after chunk of sources, so that a few thousand sources exercise what a wave
    // does between chunks: the record table's turn-over and its stragglers, the key ring across pool chunks)
    if (d->plan & 8) grid = 1;
    HIP_CHECK(hipEventRecord(d->ev0, st));
    if (prune) {
        launch_active_list(d, st, args, args.n_items);
        args.act_index = d->d_act_index;
        args.act_node = d->d_act_node;
    } else {
        HIP_CHECK(hipMemsetAsync(args.cand_count, 0, args.n_items * sizeof(uint32_t), st));  // (part of the level: see the kernel's result stores)
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(ENUM_WPB * 64), 0, st, args);
    HIP_CHECK(hipGetLastError());
    const unsigned post_grid = (unsigned)std::min<uint64_t>((args.n_items + 255) / 256 + 1, (uint64_t)d->n_cu * 8);
    static_assert(ENUM_MAX_HITS <= 32, "the post-pass sorts up to 32 keys");
    hipLaunchKernelGGL(fix_compact_kernel, dim3(d->n_cu * 4), dim3(256), 0, st, args.fix_list, args.fix_val, args.counters, d->d_fix_dense, d->d_fix_dense_val);
    hipLaunchKernelGGL((sort_lists_kernel<256>), dim3(3 * post_grid), dim3(256), 0, st, args.pool, args.pool_cap, args.cand_count, d->d_fix_dense,
                       d->d_fix_dense_val, args.counters);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(d->ev1, st));
}

static void launch_level(Device *d, hipStream_t st, const LevelCfg &cfg, bool count, SsspArgs args) {
    if (args.n_items == 0) return;
    const uint64_t n_batches = (args.n_items + cfg.bsrc - 1) / cfg.bsrc;
    sssp_fn fn = count ? cfg.fn_count : cfg.fn;
    int occ = 1;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, cfg.block, 0));
    if (occ < 1) occ = 1;
    uint64_t grid = (uint64_t)d->n_cu * (uint64_t)occ;
    unsigned long long *ws = nullptr;
    uint64_t ws_stride = 0;
    if (cfg.global_ws) {
        grid = std::min<uint64_t>(grid, 64);
        const uint64_t H = 1ull << cfg.logh;
        // table[H] u64 | stage[SCAP] u64 | log[QCAP] u32 | stage_src[SCAP] u16
        ws_stride = H + (uint64_t)cfg.scap + ((uint64_t)cfg.qcap + 1) / 2 + ((uint64_t)cfg.scap + 3) / 4;
        grid = std::min<uint64_t>(grid, n_batches);
        hu::device_malloc(&ws, grid * ws_stride * 8);
    }
    grid = std::max<uint64_t>(1, std::min<uint64_t>(grid, n_batches));
    args.ws = ws;
    args.ws_stride = ws_stride;
    HIP_CHECK(hipEventRecord(d->ev0, st));
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(cfg.block), 0, st, args);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(d->ev1, st));
    if (ws) {
        HIP_CHECK(hipStreamSynchronize(st));
        hu::device_free(ws);
    }
}

// ------------------------------------------------------------------------------------------------
// Last level: no limit on the ball. One source at a time over DENSE arrays in HBM (a 32-bit tentative distance and a round
// stamp per node): frontier rounds of label-correcting relaxations (atomicMin; a node joins the next frontier once per
// round), until the frontier is empty; then one pass over the nodes collects the in-nodes within the bound. O(V) per source
// on top of the ball: only sources that overflowed the 2^22-entry global-workspace level land here (none on any genome graph;
// the reference's search has no such limit either, greedytigs/mod.rs:548-551), and the price buys "no abort on legal input".
// ------------------------------------------------------------------------------------------------
__global__ void dense_relax_kernel(const NodeBlock *recs, const uint32_t *ext_col, const uint16_t *ext_w, uint32_t K1, uint32_t wmask, const uint32_t *frontier,
                                   uint64_t n_front, uint32_t round, uint32_t *dist, uint32_t *stamp, uint32_t *next, unsigned long long *n_next) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_front) return;
    const uint32_t u = frontier[i];
    const uint32_t du = __hip_atomic_load(&dist[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const NodeBlock &r = recs[u];
    auto relax = [&](uint32_t v, uint32_t w) {
        const uint64_t nd = (uint64_t)du + w;
        if (nd > K1) return;
        const uint32_t old = atomicMin(&dist[v], (uint32_t)nd);
        if ((uint32_t)nd < old && atomicExch(&stamp[v], round) != round) next[atomicAdd(n_next, 1ull)] = v;
    };
    if (r.flags & F_EXT) {
        const uint64_t b = (uint64_t)r.nbr[0] | ((uint64_t)r.nbr[1] << 32);
        for (uint32_t e = 0; e < r.nbr[2]; e++) relax(ext_col[b + e], ext_w[b + e]);
    } else {
        for (uint32_t e = 0; e < r.deg; e++) relax(r.nbr[e], r.w[e] & wmask);
    }
}
__global__ void dense_collect_kernel(const NodeBlock *recs, uint64_t n_nodes, uint32_t source, uint32_t K1, const uint32_t *dist,
                                     unsigned long long *keys, unsigned long long *n_keys) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes || v == source) return;  // forbid_source_target
    const uint32_t d = dist[v];
    if (d <= K1 && (recs[v].flags & F_TARGET)) keys[atomicAdd(n_keys, 1ull)] = ((unsigned long long)d << 32) | v;
}
__global__ void dense_set_kernel(uint32_t *dist, uint32_t *frontier, uint32_t source) {
    dist[source] = 0;
    frontier[0] = source;
}
__global__ void dense_result_kernel(unsigned long long *cand_start, uint32_t *cand_count, uint64_t slot, unsigned long long start, uint32_t count) {
    cand_start[slot] = start;
    cand_count[slot] = count;
}

// Sources [0, n_src) of `list` (absolute source indices). Keys go to the pool at the cursor C_POOL like every other level's;
// their order (distance, node) -- all distinct -- is established by a sort of the collected keys on the host.
static void run_dense_level(Device *d, hipStream_t st, const SsspArgs &a, const uint32_t *d_list, uint64_t n_src) {
    const uint64_t V = d->V;
    std::vector<uint32_t> list(n_src);
    HIP_CHECK(hipMemcpyAsync(list.data(), d_list, n_src * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    std::sort(list.begin(), list.end());
    uint32_t *d_dist = nullptr, *d_stamp = nullptr, *d_front[2] = {nullptr, nullptr};
    unsigned long long *d_keys = nullptr, *d_n = nullptr;
    hu::device_malloc(&d_dist, V * 4);
    hu::device_malloc(&d_stamp, V * 4);
    hu::device_malloc(&d_front[0], V * 4);
    hu::device_malloc(&d_front[1], V * 4);
    hu::device_malloc(&d_keys, V * 8);
    hu::device_malloc(&d_n, 16);
    std::vector<unsigned long long> keys;
    std::vector<uint32_t> sources(1);
    for (const uint32_t idx : list) {
        HIP_CHECK(hipMemcpyAsync(sources.data(), d->d_out_nodes + idx, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const uint32_t s = sources[0];
        HIP_CHECK(hipMemsetAsync(d_dist, 0xFF, V * 4, st));
        HIP_CHECK(hipMemsetAsync(d_stamp, 0xFF, V * 4, st));
        hipLaunchKernelGGL(dense_set_kernel, dim3(1), dim3(1), 0, st, d_dist, d_front[0], s);
        uint64_t n_front = 1;
        int cur = 0;
        for (uint32_t round = 0; n_front; round++) {
            if (round == 0xFFFFFFFEu) MTG_DIE("dense search level: round counter exhausted");
            HIP_CHECK(hipMemsetAsync(d_n, 0, 8, st));
            hipLaunchKernelGGL(dense_relax_kernel, dim3((unsigned)((n_front + 255) / 256)), dim3(256), 0, st, a.recs, a.ext_col, a.ext_w, a.K1, a.wmask,
                               d_front[cur], n_front, round, d_dist, d_stamp, d_front[cur ^ 1], d_n);
            HIP_CHECK(hipGetLastError());
            unsigned long long h = 0;
            HIP_CHECK(hipMemcpyAsync(&h, d_n, 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            n_front = h;
            cur ^= 1;
        }
        HIP_CHECK(hipMemsetAsync(d_n, 0, 8, st));
        hipLaunchKernelGGL(dense_collect_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, a.recs, V, s, a.K1, d_dist, d_keys, d_n);
        HIP_CHECK(hipGetLastError());
        unsigned long long n_keys = 0;
        HIP_CHECK(hipMemcpyAsync(&n_keys, d_n, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (n_keys > 0xFFFFFFFFull) MTG_DIE("dense search level: a candidate list beyond 2^32 entries");
        keys.resize(n_keys);
        if (n_keys) HIP_CHECK(hipMemcpyAsync(keys.data(), d_keys, n_keys * 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::sort(keys.begin(), keys.end(), [](unsigned long long x, unsigned long long y) { return mtg_policy_pops_before(x, y) != 0; });  // policy P1
        const unsigned long long start = d->h_counters[C_POOL];  // host mirror of the pool cursor (read_counters ran after the last level)
        d->h_counters[C_POOL] += n_keys;
        if (start + n_keys <= a.pool_cap && n_keys) HIP_CHECK(hipMemcpyAsync(a.pool + start, keys.data(), n_keys * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(dense_result_kernel, dim3(1), dim3(1), 0, st, a.cand_start, a.cand_count, (uint64_t)idx - a.src_begin, start, (uint32_t)n_keys);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
    }
    HIP_CHECK(hipMemcpyAsync(&d->d_counters[C_POOL], &d->h_counters[C_POOL], 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (void *p : {(void *)d_dist, (void *)d_stamp, (void *)d_front[0], (void *)d_front[1], (void *)d_keys, (void *)d_n}) hu::device_free(p);
}

void read_counters(Device *d, hipStream_t st) {
    HIP_CHECK(hipMemcpyAsync(d->h_counters, d->d_counters, C_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
}

// runs level 0 over [src_begin, src_end) and larger levels over whatever overflowed
int run_levels(Device *d, hipStream_t st, int count_mode, uint64_t src_begin, uint64_t src_end, unsigned long long *d_pool,
                      uint64_t pool_cap, unsigned long long *d_cand_start, uint32_t *d_cand_count, uint64_t *pool_needed,
                      mtg_sssp_stats *stats) {
    if (!d->d_recs) MTG_DIE("this device copy was built for one search (mtg_compute_tigs_cfg) and has given its search arrays back");
    if (!d->classified) MTG_DIE("mtg_sssp_candidates: call mtg_classify first");
    if (src_end > d->n_sources || src_begin > src_end) MTG_DIE("mtg_sssp_candidates: source range out of bounds");
    const uint64_t n = src_end - src_begin;
    const bool count = count_mode != 0;   // 1 = unit counters (cooperative plan), 2 = per-query performance data (one source per workgroup)
    const int first_coop = count_mode == 2 ? 3 : 0;
    HIP_CHECK(hipMemsetAsync(d->d_counters, 0, C_COUNT * sizeof(unsigned long long), st));
    SsspArgs a{};
    a.recs = d->d_recs; a.ext_col = d->d_ext_col; a.ext_w = d->d_ext_w;
    a.sources = d->d_out_nodes; a.src_index = nullptr; a.n_items = n; a.src_begin = src_begin;
    a.K1 = d->K1; a.pool = d_pool; a.pool_cap = pool_cap; a.cand_start = d_cand_start; a.cand_count = d_cand_count;
    a.counters = d->d_counters;
    a.wmask = d->w8 ? 0xFFu : 0xFFFFu;
    // count_mode 3 = the units of the pruned search (what the default plan really visits): the counting kernels over the searched sources only
    const bool prune_count = count_mode == 3;
    a.prune = (prune_count || !count) && enum_prunes(d) ? 1u : 0u;
    if (prune_count && !enum_prunes(d)) MTG_DIE("mtg_sssp_count_visited: this device graph / plan does not prune (k > 255 or plan + 4)");
    if (d->ovf_cap < n) {  // two overflow lists of up to n source indices each + the post-pass work list
        for (int i = 0; i < 2; i++) {
            if (d->d_ovf[i]) hu::device_free(d->d_ovf[i]);
            hu::device_malloc(&d->d_ovf[i], std::max<uint64_t>(n, 1) * sizeof(uint32_t));
        }
        if (d->d_fix) hu::device_free(d->d_fix);

        // (a wave abandons a work-list chunk with fewer than 64 free slots: < 1/7 of every chunk incl. its tag) + three open chunks per wave
        const uint64_t fix_slots = std::max<uint64_t>(n, 1) * 5 / 4 + (uint64_t)d->n_cu * 32 * 3 * ENUM_FIX_CHUNK;
        hu::device_malloc(&d->d_fix, fix_slots * sizeof(uint32_t));
        if (d->d_fix_dense) hu::device_free(d->d_fix_dense);
        hu::device_malloc(&d->d_fix_dense, std::max<uint64_t>(n, 1) * sizeof(uint32_t));
        if (d->d_fix_val) hu::device_free(d->d_fix_val);
        hu::device_malloc(&d->d_fix_val, fix_slots * sizeof(unsigned long long));
        if (d->d_fix_dense_val) hu::device_free(d->d_fix_dense_val);
        hu::device_malloc(&d->d_fix_dense_val, std::max<uint64_t>(n, 1) * sizeof(unsigned long long));
        d->ovf_cap = n;
    }
    a.ovf_list = d->d_ovf[0];
    a.fix_list = d->d_fix;
    a.fix_val = d->d_fix_val;
    double total_ms = 0.0;
    // the counting instantiations (untimed instrumentation) exist for the cooperative kernel only: it counts DISTINCT
    // settled nodes, an enumeration counts path steps
    const bool use_enum = (d->plan & 3) != 1 && !count && d->K1 < 0x8000u;  // (the enumeration level keeps 15-bit distances on its stack)
    const bool dense_only = d->K1 >= (1u << 21);  // (the cooperative levels keep 21-bit distances in their table entries)
    if (dense_only) {  // every source straight to the dense level: correct for any bound, O(V) per source
        if (count) MTG_DIE("the counting kernels need k - 1 < 2^21");
        std::vector<uint32_t> all(n);
        for (uint64_t i = 0; i < n; i++) all[i] = (uint32_t)(src_begin + i);
        if (n) HIP_CHECK(hipMemcpyAsync(d->d_ovf[0], all.data(), n * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        d->h_counters[C_POOL] = 0;
        const auto t0 = std::chrono::steady_clock::now();
        if (n) run_dense_level(d, st, a, d->d_ovf[0], n);
        d->last_kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        d->last_n_levels = 1;
        d->last_level_ms[0] = d->last_kernel_ms;
        d->last_level_sources[0] = n;
        d->last_level_name[0] = "dense_relax_kernel rounds + dense_collect_kernel (one source at a time)";
        const bool small = d->h_counters[C_POOL] > pool_cap;
        if (pool_needed) *pool_needed = d->h_counters[C_POOL];
        return small ? 1 : 0;
    }
    uint64_t n_first = n;  // sources the first level really launches over
    if (use_enum) launch_enum(d, st, a);
    else if (prune_count) {
        if (n) {
            launch_active_list(d, st, a, n);
            read_counters(d, st);
            n_first = d->h_counters[C_ACTIVE];
            SsspArgs b = a;
            b.src_index = d->d_act_index + d->h_counters[C_ACT_BEGIN];
            b.n_items = n_first;
            launch_level(d, st, coop_level(first_coop, false), count, b);
        }
    } else launch_level(d, st, coop_level(first_coop, false), count, a);
    read_counters(d, st);
    if (use_enum && enum_prunes(d)) d->last_active_sources = d->h_counters[C_ACTIVE];
    d->last_n_levels = 0;
    if (n && n_first) {
        total_ms += elapsed_ms(d);
        d->last_level_ms[0] = elapsed_ms(d); d->last_level_sources[0] = n; d->last_n_levels = 1;
        d->last_level_name[0] = use_enum ? enum_level_name(enum_uses_quad_gathers(d), enum_prunes(d)) : coop_level(first_coop, false).name();
    }
    static const bool debug = std::getenv("MTG_DEBUG") != nullptr;
    if (debug && n && n_first) std::fprintf(stderr, "[mtg] level0 (%s): %llu sources (%llu searched), %.3f ms, %llu overflowed, cum settled %llu\n", use_enum ? "enum" : "coop level 0",
                                 (unsigned long long)n, (unsigned long long)(use_enum && enum_prunes(d) ? d->h_counters[C_ACTIVE] : n_first), elapsed_ms(d),
                                 (unsigned long long)d->h_counters[C_OVERFLOW], (unsigned long long)d->h_counters[C_SETTLED]);
#ifdef MTG_ENUM_STATS
    if (use_enum && n) std::fprintf(stderr, "[mtg] enum stats: wave steps %llu, lane steps %llu (%.1f of 64), starved %llu, block full %llu, step budget + records stored by stragglers %llu, fix cursor %llu, pool cursor %llu\n",
                                    d->h_counters[C_SETTLED], d->h_counters[C_RELAXED], (double)d->h_counters[C_RELAXED] / (double)std::max<unsigned long long>(d->h_counters[C_SETTLED], 1),
                                    d->h_counters[C_EMITTED], d->h_counters[C_ATTEMPTS], d->h_counters[C_PUSHES], d->h_counters[C_FIX], d->h_counters[C_POOL]);
    if (use_enum && n) {  // when the waves end, and how many steps they ran
        static std::vector<unsigned long long> hp(16384 * 3);
        HIP_CHECK(hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_enum_prof), hp.size() * 8));
        std::vector<double> end_us, life_us, steps, start_us;
        unsigned long long t_min = ~0ull;
        for (int w = 0; w < 16384; w++) if (hp[3 * w + 1]) t_min = std::min(t_min, hp[3 * w]);
        for (int w = 0; w < 16384; w++)
            if (hp[3 * w + 1]) { start_us.push_back((hp[3 * w] - t_min) * 0.01); end_us.push_back((hp[3 * w + 1] - t_min) * 0.01); life_us.push_back((hp[3 * w + 1] - hp[3 * w]) * 0.01); steps.push_back((double)hp[3 * w + 2]); }
        auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(q * (v.size() - 1))]; };
        std::fprintf(stderr, "[mtg] enum waves: %zu; end of wave (us after the first start) min %.0f p10 %.0f median %.0f p90 %.0f max %.0f; steps per wave min %.0f median %.0f max %.0f; us per step median %.3f\n",
                     end_us.size(), pct(end_us, 0), pct(end_us, 0.1), pct(end_us, 0.5), pct(end_us, 0.9), pct(end_us, 1), pct(steps, 0), pct(steps, 0.5), pct(steps, 1),
                     pct(life_us, 0.5) / std::max(1.0, pct(steps, 0.5)));
        std::fprintf(stderr, "[mtg] enum waves: start (us after the first) median %.0f p90 %.0f p99 %.0f max %.0f; life median %.0f p90 %.0f max %.0f us\n",
                     pct(start_us, 0.5), pct(start_us, 0.9), pct(start_us, 0.99), pct(start_us, 1), pct(life_us, 0.5), pct(life_us, 0.9), pct(life_us, 1));
        {  // by XCD (workgroup b runs on XCD b mod 8; wave w belongs to workgroup w / ENUM_WPB): do some XCDs end later than others?
            std::vector<double> xe[8], xs[8];
            for (int w = 0; w < 16384; w++)
                if (hp[3 * w + 1]) { const int x = (w / ENUM_WPB) % 8; xe[x].push_back((hp[3 * w + 1] - t_min) * 0.01); xs[x].push_back((hp[3 * w + 1] - hp[3 * w]) * 0.01 / std::max<double>(1.0, (double)hp[3 * w + 2])); }
            char line[512]; int o = 0;
            for (int x = 0; x < 8; x++) o += std::snprintf(line + o, sizeof line - o, " %d: %.0f / %.0f us, %.2f us per step;", x, pct(xe[x], 0.5), pct(xe[x], 1.0), pct(xs[x], 0.5));
            std::fprintf(stderr, "[mtg] enum waves by XCD (median / last end):%s\n", line);
        }
        std::vector<unsigned long long> zero(16384 * 3, 0);
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_enum_prof), zero.data(), zero.size() * 8));
    }
#endif
    const uint64_t total_overflow = d->h_counters[C_OVERFLOW];
    // remaining levels over whatever overflowed the previous one; each launch appends the sources it could not finish
    // to the other of two ping-pong lists
    int cur_list = 0;
    // (what the enumeration level hands on are the heavy balls: batches of 32 of them do not fit the first cooperative level)
    for (int li = first_coop + 1; li < N_COOP_LEVELS && d->h_counters[C_OVERFLOW] > 0; li++) {
        const LevelCfg next = coop_level(li, use_enum);
        const uint64_t n_ovf = d->h_counters[C_OVERFLOW];
        HIP_CHECK(hipMemsetAsync(&d->d_counters[C_OVERFLOW], 0, sizeof(unsigned long long), st));
        SsspArgs b = a;
        b.src_index = d->d_ovf[cur_list];
        b.ovf_list = d->d_ovf[cur_list ^ 1];
        cur_list ^= 1;
        b.n_items = n_ovf;
        launch_level(d, st, next, count, b);
        read_counters(d, st);
        total_ms += elapsed_ms(d);
        if (d->last_n_levels < 8) {
            d->last_level_ms[d->last_n_levels] = elapsed_ms(d);
            d->last_level_sources[d->last_n_levels] = n_ovf;
            d->last_level_name[d->last_n_levels] = next.name();
            d->last_n_levels++;
        }
        if (debug) std::fprintf(stderr, "[mtg] level %d (coop, %d src/block): %llu sources, %.3f ms, %llu overflowed, cum settled %llu\n", li,
                                next.bsrc, (unsigned long long)n_ovf, elapsed_ms(d), (unsigned long long)d->h_counters[C_OVERFLOW], (unsigned long long)d->h_counters[C_SETTLED]);
    }
    if (d->h_counters[C_OVERFLOW] > 0) {  // balls beyond the 2^22 entries of the global-workspace level: the dense level has no limit
        const uint64_t n_ovf = d->h_counters[C_OVERFLOW];
        if (count) MTG_DIE("the counting kernels cannot follow %llu source(s) beyond the last cooperative level", (unsigned long long)n_ovf);
        const auto t0 = std::chrono::steady_clock::now();
        run_dense_level(d, st, a, d->d_ovf[cur_list], n_ovf);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        total_ms += ms;
        if (d->last_n_levels < 8) {
            d->last_level_ms[d->last_n_levels] = ms;
            d->last_level_sources[d->last_n_levels] = n_ovf;
            d->last_level_name[d->last_n_levels] = "dense_relax_kernel rounds + dense_collect_kernel (one source at a time)";
            d->last_n_levels++;
        }
        if (debug) std::fprintf(stderr, "[mtg] dense level: %llu sources, %.3f ms\n", (unsigned long long)n_ovf, ms);
        d->h_counters[C_OVERFLOW] = 0;
    }
    if (!count) d->last_kernel_ms = total_ms;
    if (stats) {
        stats->sources = n;
        stats->settled_nodes = d->h_counters[C_SETTLED];
        stats->relaxed_edges = d->h_counters[C_RELAXED];
        stats->emitted = d->h_counters[C_EMITTED];
        stats->relax_attempts = d->h_counters[C_ATTEMPTS];
        stats->overflow_sources = total_overflow;
    }
    // C_POOL is the pool cursor: keys plus the unused tails of block-local chunks (positions are launch-dependent,
    // (start,count) index the content). A retry may chunk differently, hence the slack.
    const bool too_small = !count && d->h_counters[C_POOL] > pool_cap;
    if (pool_needed) *pool_needed = d->h_counters[C_POOL] + (too_small ? (uint64_t)d->n_cu * 16 * POOL_CHUNK : 0);
    return too_small ? 1 : 0;
}

int device_sssp(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, uint64_t *d_pool, uint64_t pool_cap,
                uint64_t *d_cand_start, uint32_t *d_cand_count, uint64_t *pool_needed) {
    HIP_CHECK(hipSetDevice(d->dev));
    return run_levels(d, (hipStream_t)stream, 0, src_begin, src_end, (unsigned long long *)d_pool, pool_cap,
                      (unsigned long long *)d_cand_start, d_cand_count, pool_needed, nullptr);
}

void device_sssp_count(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats) {
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t n = src_end - src_begin;
    unsigned long long *d_start = nullptr;
    uint32_t *d_count = nullptr;
    hu::device_malloc(&d_start, std::max<uint64_t>(n, 1) * 8);
    hu::device_malloc(&d_count, std::max<uint64_t>(n, 1) * 4);
    const double keep_ms = d->last_kernel_ms;
    run_levels(d, (hipStream_t)stream, 1, src_begin, src_end, nullptr, 0, d_start, d_count, nullptr, stats);
    d->last_kernel_ms = keep_ms;
    hu::device_free(d_start);
    hu::device_free(d_count);
}

// the same counters for the search the default plan really runs: only the sources that can reach an in-node, successors pruned by
// their lower bounds (settled = distinct (source, node) pairs visited, relaxed = their out-edges, emitted = the same candidates)
void device_sssp_count_visited(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats) {
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t n = src_end - src_begin;
    unsigned long long *d_start = nullptr;
    uint32_t *d_count = nullptr;
    hu::device_malloc(&d_start, std::max<uint64_t>(n, 1) * 8);
    hu::device_malloc(&d_count, std::max<uint64_t>(n, 1) * 4);
    const double keep_ms = d->last_kernel_ms;
    run_levels(d, (hipStream_t)stream, 3, src_begin, src_end, nullptr, 0, d_start, d_count, nullptr, stats);
    if (stats) stats->sources = d->h_counters[C_ACTIVE];
    d->last_kernel_ms = keep_ms;
    hu::device_free(d_start);
    hu::device_free(d_count);
}
bool device_prunes(const Device *d) { return enum_prunes(d); }
uint64_t device_last_active_sources(const Device *d) { return d->last_active_sources; }

// greedytigs/mod.rs:647-673 counters in the engine's terms (see mtg_dijkstra_performance_data): one source per workgroup
void device_performance_data(Device *d, void *stream, mtg_dijkstra_performance_data *out) {
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t n = d->n_sources;
    unsigned long long *d_start = nullptr;
    uint32_t *d_count = nullptr;
    hu::device_malloc(&d_start, std::max<uint64_t>(n, 1) * 8);
    hu::device_malloc(&d_count, std::max<uint64_t>(n, 1) * 4);
    mtg_sssp_stats st{};
    run_levels(d, (hipStream_t)stream, 2, 0, n, nullptr, 0, d_start, d_count, nullptr, &st);
    hu::device_free(d_start);
    hu::device_free(d_count);
    out->dijkstras = n;
    out->iterations = st.settled_nodes;
    out->heap_pushes = d->h_counters[C_PUSHES];
    out->unnecessary_heap_elements = d->h_counters[C_PUSHES] - st.settled_nodes;
    out->max_max_heap_size = d->h_counters[C_MAX_LOG];
    out->max_max_distance_array_size = d->h_counters[C_MAX_ENT];
    out->sum_max_heap_size = d->h_counters[C_PUSHES];
    out->sum_max_distance_array_size = st.settled_nodes;
}

double device_last_kernel_ms(const Device *d) { return d->last_kernel_ms; }
const char *device_last_level_name(const Device *d, int level) {
    return level >= 0 && level < d->last_n_levels ? d->last_level_name[level].c_str() : "";
}
int device_last_levels(const Device *d, double *ms, uint64_t *sources, int cap) {
    const int n = d->last_n_levels < cap ? d->last_n_levels : cap;
    for (int i = 0; i < n; i++) { ms[i] = d->last_level_ms[i]; sources[i] = d->last_level_sources[i]; }
    return n;
}


void device_warm_sssp_unit(hipFuncAttributes *a) { (void)hipFuncGetAttributes(a, reinterpret_cast<const void *>(fix_compact_kernel)); }

}  // namespace mtg
