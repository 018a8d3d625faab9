// cut_first_device.hip -- the tigs of an Eulerised bigraph straight from the pairing, without closed walks ("cut first").
//
// What the reference does after Eulerisation (greedytigs/mod.rs:722-789, eulertigs/mod.rs:119-186): one closed walk per connected
// component, rotated to its longest dummy, cut at every breaking edge (weight k) -- the tigs are the stretches between consecutive
// breaking edges. The device-order decomposition of euler_device.hip builds those closed walks in full (trail labels, hooking rounds,
// rotation, ranking: two thirds of the all-GPU step) only for the cutter to chop them at 23.7 M breaking edges two kernels later.
// Here the walks are never built. After the mirror-symmetric pairing `succ` (euler_device.hip, step 2) the darts form closed TRAILS;
// a trail that contains a breaking dart needs no label, no hooking, no rotation and no ranking:
//   * its tigs are the succ-chains between consecutive breaking darts -- one walker per breaking dart b follows succ from succ[b]
//     to the next breaking dart (mean 3-4 darts on the bench graph);
//   * trails come in disjoint mirror pairs (succ commutes with mirroring), and so do the stretches: the stretch after b1 that ends
//     before b2 is the mirror image of the stretch after b2^1 that ends before b1^1 -- the one with b1 < b2^1 is emitted;
//   * the number of tigs and their cumulative length do not depend on how the trails of a component would have been joined into one
//     closed walk: joining never creates or removes a breaking dart, it only re-pairs stretch ends (SURVEY 8a invariance note).
// What remains are the trails WITHOUT a breaking dart (a self-loop paired with itself, a two-cycle, a balanced component): the first
// pass marks every dart it walks, the unmarked non-breaking darts are exactly those trails, and each is spliced into a trail it
// touches -- the succ words of one in-dart of either trail swap at a shared node, and mirror-symmetrically at the mirror node --
// which lengthens one stretch and changes nothing else; a component none of whose trails has a breaking dart becomes one tig, cut at
// its longest matched dummy (greedytigs/mod.rs:737-788). They are few (a handful on the bench graphs), so the splicing is a small
// sequential job on the host over records the GPU gathers for exactly those darts (splice_breaking_free, below); a graph where they
// are many (more than a 64th of the darts) goes through the closed walks of euler_device.hip instead, as before.
//
// Integer gather / scatter work, HBM-bound (random 4-byte gathers from the 0.75-GB successor array): no MFMA. Deterministic: the tigs
// are ordered by their walker's breaking dart, nothing depends on thread timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "device.hpp"
#include "finish_device.hpp"
#include "hip_util.hpp"

namespace mtg {

using namespace hu;

namespace {

constexpr uint32_t MARK = 0x80000000u;  // bit 31 of a successor word: the first pass has walked this dart (dart ids stay below 2^31 here)

// ---- pass 1: one walker per breaking dart -- length of the stretch behind it, the breaking dart that ends it, marks ------------
// (a walker reads succ[b] of consecutive b coalesced; every further step is a dependent random gather.)
// Marks: what has to be found afterwards are the closed trails NO walker passes. Marking every walked dart costs a store per step
// (measured: 9.4 ms for the pass at 2^27 against 3-4 without). It is enough that every such trail PAIR keeps one dart that would
// have been marked had a walker passed: the smallest dart m of a trail pair is even (m odd would have m ^ 1 = m - 1 in the mirror
// trail) and not larger than its predecessor or its successor on its trail -- so only EVEN darts that are LOCAL MINIMA of the walk
// (x <= previous dart, x <= next dart; a breaking dart counts as larger than anything) get the mark, a sixth of the darts, and the
// search for unwalked trails looks at exactly the darts with that property (unwalked_rep_kernel). The mark is a plain store of the word
// just read with bit 31 set: every dart is walked by one walker only, and nobody waits for the store.
__device__ __forceinline__ bool is_rep_candidate(uint32_t x, uint32_t prev, uint32_t next) { return !(x & 1u) && x <= prev && x <= next; }
// Output per walker: the length of its stretch if the stretch is emitted, else 0, and the emit flag. The stretch behind breaking dart b that
// ends before breaking dart e is emitted iff it is not empty and b < (e ^ 1) (its mirror image lies behind e ^ 1 and ends before b ^ 1).
__global__ __launch_bounds__(EB) void stretch_measure_kernel(uint32_t *succ, uint64_t first_brk, uint64_t n_brk, uint32_t *keep_len, uint32_t *is_tig,
                                                            unsigned long long *walked_blocks, uint32_t *error) {
    const uint64_t i = gid();
    uint32_t len = 0;
    if (i < n_brk) {
        const uint32_t b = (uint32_t)(first_brk + i);
        uint32_t prev = b, x = succ[b] & ~MARK;
        while (x < first_brk) {
            const uint32_t s = succ[x];
            if (s & MARK) { atomicOr(error, 2u); break; }  // walked twice: succ is not a permutation
            if (is_rep_candidate(x, prev, s)) succ[x] = s | MARK;
            prev = x;
            x = s;
            if (++len == 0xFFFFFFFFu) { atomicOr(error, 4u); break; }
        }
        const bool emit = len != 0 && b < (x ^ 1u);
        keep_len[i] = emit ? len : 0u;
        is_tig[i] = emit ? 1u : 0u;
        if (x == (b ^ 1u)) atomicOr(error, 8u);  // a stretch that is its own mirror image: the pairing's trails would not be disjoint from their mirrors
    }
    // darts walked, per block (summed by sum_blocks_kernel: 740 K waves adding to one word cost more than the walk itself)
    __shared__ unsigned long long wave_sum[EB / 64];
    unsigned long long t = len;
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = 0;
        for (int w = 0; w < EB / 64; w++) b += wave_sum[w];
        walked_blocks[blockIdx.x] = b;
    }
}
__global__ __launch_bounds__(1024) void sum_blocks_kernel(const unsigned long long *v, uint64_t n, unsigned long long *out) {
    __shared__ unsigned long long part[1024 / 64];
    unsigned long long t = 0;
    for (uint64_t i = threadIdx.x; i < n; i += 1024) t += v[i];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = 0;
        for (int w = 0; w < 1024 / 64; w++) b += part[w];
        *out = b;
    }
}
// ---- pass 2: the emitted stretches are walked once more and written to their places ------------------------------------------------
__global__ __launch_bounds__(EB) void stretch_write_kernel(const uint32_t *succ, uint64_t first_brk, uint64_t n_brk, const uint32_t *keep_len,
                                                          const uint32_t *edge_off, const uint32_t *tig_idx, uint32_t *tig_edges, uint32_t *tig_limits) {
    const uint64_t i = gid();
    if (i >= n_brk) return;
    const uint32_t n = keep_len[i];
    if (!n) return;
    const uint32_t o = edge_off[i];
    uint32_t x = succ[first_brk + i] & ~MARK;
    for (uint32_t j = 0; j < n; j++) {
        tig_edges[o + j] = x;
        x = succ[x] & ~MARK;
    }
    tig_limits[tig_idx[i]] = o + n;
}
// ---- the trails no walker has passed ------------------------------------------------------------------------------------------------
// A streaming pass over the even non-breaking darts: x with the mark property (see stretch_measure_kernel; pred(x) = succ[x ^ 1] ^ 1 is
// the neighbouring word) but without a mark lies on a trail without a breaking dart. Its thread walks that trail once: the trail
// pair's representative is the smallest even dart of the trail and its mirror, min over the trail's darts y of (y & ~1); the thread
// that IS the representative reports the pair -- count only (list == null), or every dart of both trails. A trail longer than
// `max_len` raises bit 32 of *error: such a graph goes through the closed walks instead.
__global__ __launch_bounds__(EB) void unwalked_rep_kernel(const uint32_t *succ, uint64_t first_brk, uint32_t max_len, uint32_t *cursor, uint32_t cap, uint32_t *list,
                                                         uint32_t *error) {
    // (eight successor words = four even darts with their mirrors per thread: a streaming pass at a few bytes per thread runs at a
    // quarter of the bandwidth)
    const uint64_t x0 = gid() * 8;
    if (x0 >= first_brk) return;
    uint32_t w[8];
    if (x0 + 8 <= first_brk) {
        const uint4 a = reinterpret_cast<const uint4 *>(succ)[x0 / 4], b = reinterpret_cast<const uint4 *>(succ)[x0 / 4 + 1];
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else {
        for (int t = 0; t < 8; t++) w[t] = x0 + t < first_brk ? succ[x0 + t] : MARK;
    }
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
        const uint32_t x = (uint32_t)x0 + t;
        if ((w[t] & MARK) || x0 + t >= first_brk) continue;
        const uint32_t next = w[t], prev = (w[t + 1] & ~MARK) ^ 1u;
        if (!(x <= prev && x <= next)) continue;
        // (an unmarked candidate: on an unwalked trail -- a walked one would have been marked)
        uint32_t len = 0, lo = x;
        bool bad = false;
        for (uint32_t y = x;;) {
            lo = min(lo, y & ~1u);
            len++;
            y = succ[y] & ~MARK;
            if (y == x) break;
            if (y >= first_brk || len > max_len) { atomicOr(error, y >= first_brk ? 64u : 32u); bad = true; break; }
        }
        if (bad || lo != x) continue;
        const uint32_t at = atomicAdd(cursor, 2u * len);
        if (!list) continue;
        uint32_t o = at;
        for (uint32_t y = x, j = 0; j < len; j++) {
            if (o + 1 < cap) { list[o] = y; list[o + 1] = y ^ 1u; }
            o += 2;
            y = succ[y] & ~MARK;
        }
    }
}

// ---- records of the darts on trails without a breaking dart, for the host's splicing ------------------------------------------------
__global__ __launch_bounds__(EB) void unwalked_head_kernel(const uint32_t *list, uint32_t n, const uint32_t *succ, const uint32_t *from, const uint32_t *row,
                                                          uint32_t *out_succ, uint32_t *out_node, uint32_t *out_deg) {
    const uint64_t i = gid();
    if (i >= n) return;
    const uint32_t x = list[i], v = from[x];
    out_succ[i] = succ[x] & ~MARK;
    out_node[i] = v;
    out_deg[i] = row[v + 1] - row[v];
}
// the out-darts of the dart's from-node, each with its current predecessor: pred(o) = succ[o ^ 1] ^ 1 (the pairing commutes with mirroring)
__global__ __launch_bounds__(EB) void unwalked_adj_kernel(const uint32_t *node, const uint32_t *off, uint32_t n, const uint32_t *succ, const uint32_t *row,
                                                         const uint32_t *adj, uint32_t *out_dart, uint32_t *out_pred) {
    const uint64_t i = gid();
    if (i >= n) return;
    const uint32_t v = node[i], lo = row[v], d = row[v + 1] - lo, o0 = off[i];
    for (uint32_t j = 0; j < d; j++) {
        const uint32_t o = adj[lo + j];
        out_dart[o0 + j] = o;
        out_pred[o0 + j] = (succ[o ^ 1u] & ~MARK) ^ 1u;
    }
}
__global__ __launch_bounds__(EB) void dummy_weight_kernel(const uint32_t *darts, uint32_t n, uint32_t E0, const uint32_t *pair_w, uint32_t *out) {
    const uint64_t i = gid();
    if (i < n) out[i] = pair_w[(darts[i] - E0) >> 1];
}
__global__ __launch_bounds__(EB) void succ_patch_kernel(uint32_t *succ, const uint32_t *dart, const uint32_t *value, uint32_t n) {
    const uint64_t i = gid();
    if (i < n) succ[dart[i]] = (succ[dart[i]] & MARK) | value[i];
}
// a stretch that took a spliced trail in: from any dart on it back to its walker (pred(x) = succ[x ^ 1] ^ 1), then measured again.
// Several darts of one stretch arrive at the same walker and write the same values.
__global__ __launch_bounds__(EB) void stretch_fix_kernel(uint32_t *succ, const uint32_t *darts, uint32_t n, uint64_t first_brk, uint32_t *keep_len, uint32_t *is_tig,
                                                        uint32_t *error) {
    const uint64_t i = gid();
    if (i >= n) return;
    uint32_t y = darts[i];
    for (uint64_t guard = 0; y < first_brk; guard++) {
        y = (succ[y ^ 1u] & ~MARK) ^ 1u;
        if (guard > 0xFFFFFFFFull) { atomicOr(error, 16u); return; }
    }
    uint32_t x = succ[y] & ~MARK, len = 0;
    while (x < first_brk) {
        const uint32_t s = succ[x];
        succ[x] = s | MARK;  // (idempotent: the same walker may be measured by several threads)
        x = s & ~MARK;
        if (++len == 0xFFFFFFFFu) { atomicOr(error, 4u); return; }
    }
    const bool emit = len != 0 && y < (x ^ 1u);
    keep_len[y - first_brk] = emit ? len : 0u;
    is_tig[y - first_brk] = emit ? 1u : 0u;
    if (x == (y ^ 1u)) atomicOr(error, 8u);
}

// ---- the splicing itself: sequential, over the few darts concerned (see the head of this file) ---------------------------------------
struct SpliceResult {
    std::vector<uint32_t> patch_dart, patch_value;  // successor words to rewrite on the device
    std::vector<uint32_t> touched;                  // darts on stretches that took a trail in
    std::vector<uint32_t> cyc_edges, cyc_limits;    // tigs of the components none of whose trails has a breaking dart
    uint64_t spliced = 0, cyc_dropped = 0;
};
// fd: the darts on trails without a breaking dart; fs[i] = succ[fd[i]]; (ro[i], ro[i + 1]) = range of the out-darts od[] of from[fd[i]]
// with their predecessors op[]. dummy_weights(darts) = the weights of those matched dummy darts.
template <typename W>
SpliceResult splice_breaking_free(const std::vector<uint32_t> &fd, const std::vector<uint32_t> &fs, const std::vector<uint32_t> &ro, const std::vector<uint32_t> &od,
                                  const std::vector<uint32_t> &op, uint64_t E0, uint64_t first_brk, W &&dummy_weights) {
    SpliceResult r;
    const uint32_t n = (uint32_t)fd.size();
    std::unordered_map<uint32_t, uint32_t> idx;  // dart -> position in fd
    idx.reserve(n * 2);
    for (uint32_t i = 0; i < n; i++) idx.emplace(fd[i], i);
    std::unordered_map<uint32_t, uint32_t> succ_ov, pred;  // rewritten successor words; current predecessor of every out-dart of a node concerned
    pred.reserve(od.size() * 2);
    for (size_t j = 0; j < od.size(); j++) pred[od[j]] = op[j];
    auto succ_of = [&](uint32_t d) -> uint32_t {
        auto o = succ_ov.find(d);
        if (o != succ_ov.end()) return o->second;
        auto it = idx.find(d);
        if (it == idx.end()) MTG_DIE("device_cut_first: internal error (successor of dart %u is not among the records)", d);
        return fs[it->second];
    };
    // trails and trail pairs (a trail and its mirror trail): classes of a union-find over the trails; class 0 = "has a breaking dart"
    std::vector<uint32_t> trail_of(n, 0xFFFFFFFFu);
    uint32_t n_trails = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (trail_of[i] != 0xFFFFFFFFu) continue;
        for (uint32_t x = fd[i];;) {
            auto it = idx.find(x);
            if (it == idx.end()) MTG_DIE("device_cut_first: internal error (an unmarked trail leaves the unmarked darts at %u)", x);
            if (trail_of[it->second] != 0xFFFFFFFFu) break;
            trail_of[it->second] = n_trails;
            x = fs[it->second];
        }
        n_trails++;
    }
    std::vector<uint32_t> parent(n_trails + 1);  // class ids are trail + 1; 0 = B
    for (uint32_t t = 0; t <= n_trails; t++) parent[t] = t;
    auto find = [&](uint32_t c) { while (parent[c] != c) { parent[c] = parent[parent[c]]; c = parent[c]; } return c; };
    auto unite = [&](uint32_t a, uint32_t b) { a = find(a); b = find(b); if (a == b) return; if (a < b) parent[b] = a; else parent[a] = b; };  // (0 always wins)
    for (uint32_t i = 0; i < n; i++) {
        auto m = idx.find(fd[i] ^ 1u);
        if (m == idx.end()) MTG_DIE("device_cut_first: internal error (the mirror of unmarked dart %u is marked)", fd[i]);
        unite(trail_of[i] + 1, trail_of[m->second] + 1);
    }
    auto class_of = [&](uint32_t d) -> uint32_t {
        auto it = idx.find(d);
        return it == idx.end() ? 0u : find(trail_of[it->second] + 1);
    };
    for (bool changed = true; changed;) {
        changed = false;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t x = fd[i], c = class_of(x);
            if (c == 0) continue;
            for (uint32_t j = ro[i]; j < ro[i + 1]; j++) {
                const uint32_t o = od[j];
                if (o == x) continue;
                const uint32_t c2 = class_of(o);
                if (c2 == c) continue;
                // passages (e_f -> x) and (e_o -> o) at the shared node swap their out-darts; their mirror passages (x^1 -> e_f^1),
                // (o^1 -> e_o^1) at the mirror node swap accordingly
                const uint32_t e_f = pred.at(x), e_o = pred.at(o);
                succ_ov[e_f] = o;
                succ_ov[e_o] = x;
                succ_ov[o ^ 1u] = e_f ^ 1u;
                succ_ov[x ^ 1u] = e_o ^ 1u;
                pred[o] = e_f;
                pred[x] = e_o;
                pred[e_f ^ 1u] = o ^ 1u;
                pred[e_o ^ 1u] = x ^ 1u;
                const bool attaches = find(c2) == 0;  // (c itself has no breaking dart: classes that have one are skipped above)
                unite(c, c2);
                // the stretch that took the trail in, and its mirror stretch: one dart on either (the device walks back to their walkers).
                // A merge of two classes without breaking darts reports nothing: if the merged class is attached later, that event does.
                if (attaches) { r.touched.push_back(e_o); r.touched.push_back(o ^ 1u); }
                r.spliced++;
                changed = true;
                break;
            }
        }
    }
    for (auto &kv : succ_ov) { r.patch_dart.push_back(kv.first); r.patch_value.push_back(kv.second); }
    // components without any breaking dart: one closed trail pair each by now; the trail with the class's smallest dart becomes one tig,
    // rotated to its first strictly-longest dummy, which is dropped (greedytigs/mod.rs:737-788)
    std::vector<std::pair<uint32_t, uint32_t>> by_class;  // (class, dart) of the darts that stay without a breaking dart
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = class_of(fd[i]);
        if (c) by_class.emplace_back(c, fd[i]);
    }
    std::sort(by_class.begin(), by_class.end());
    // (the weights of their matched dummies, fetched in one go: a graph may have many small components that the matching balanced completely)
    std::unordered_map<uint32_t, uint32_t> w_of;
    {
        std::vector<uint32_t> dd;
        for (auto &cd : by_class) if (cd.second >= E0) dd.push_back(cd.second);
        const std::vector<uint32_t> ww = dummy_weights(dd);
        for (size_t q = 0; q < dd.size(); q++) w_of.emplace(dd[q], ww[q]);
    }
    for (size_t g0 = 0; g0 < by_class.size();) {
        size_t g1 = g0;
        while (g1 < by_class.size() && by_class[g1].first == by_class[g0].first) g1++;
        const uint32_t lo = by_class[g0].second;  // the class's smallest dart
        std::vector<uint32_t> cyc;
        for (uint32_t x = lo;;) {
            cyc.push_back(x);
            x = succ_of(x);
            if (x == lo) break;
            if (cyc.size() > g1 - g0) MTG_DIE("device_cut_first: internal error (a trail without a breaking dart does not close)");
        }
        if (cyc.size() * 2 != g1 - g0) MTG_DIE("device_cut_first: internal error (a component without a breaking dart kept %zu darts in a trail pair of %zu)", g1 - g0, cyc.size() * 2);
        uint64_t best_w = 0;
        size_t rot = 0;
        for (size_t q = 0; q < cyc.size(); q++)
            if (cyc[q] >= E0) {
                const uint64_t w = w_of.at(cyc[q]);
                if (w > best_w) { best_w = w; rot = q; }
            }
        std::rotate(cyc.begin(), cyc.begin() + (long)rot, cyc.end());
        size_t a = 0, b = cyc.size();
        if (a < b && cyc[a] >= E0) { a++; r.cyc_dropped++; }      // a dummy at index 0 is cut (:767-769)
        if (a < b && cyc[b - 1] >= E0) { b--; r.cyc_dropped++; }  // a trailing dummy is dropped (:779-788)
        for (size_t q = a; q < b; q++) {
            if (cyc[q] >= first_brk) MTG_DIE("device_cut_first: internal error (a breaking dart on a trail without one)");
            r.cyc_edges.push_back(cyc[q]);
        }
        if (b > a) r.cyc_limits.push_back((uint32_t)r.cyc_edges.size());
        g0 = g1;
    }
    return r;
}

}  // namespace

// Tigs of the Eulerian bigraph from[E] / mirror[V] (darts [E0, first_brk) = matched pairs, [first_brk, E) = breaking darts) into
// b_te (u32[n_kept]: dart ids, tig after tig) and b_tl (u32[n_tigs]: exclusive ends). Returns false -- nothing written -- when the
// graph has to go through the closed walks instead (2^31 darts or more; too many darts on trails without a breaking dart).
bool device_cut_first(hipStream_t st, const uint32_t *d_from, const uint32_t *d_mirror, uint64_t E, uint64_t V, uint64_t E0, uint64_t first_brk,
                      const uint32_t *d_pw, const uint32_t *d_row0, const uint32_t *d_adj0, const ZipBuckets *zip, Buf &b_te, Buf &b_tl,
                      uint64_t *n_kept_out, uint64_t *n_tigs_out, CutFirstStats *stats) {
    if (E == 0 || (E & 1) || E >= 0x80000000ull || first_brk > E || ((E - first_brk) & 1)) return false;
    const uint64_t n_brk = E - first_brk;
    static const bool dbg = std::getenv("MTG_DEBUG") != nullptr;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        HIP_CHECK(hipStreamSynchronize(st));
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mtg] cut first:   %-32s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t_lap).count());
        t_lap = n;
    };
    Buf b_row, b_adj, b_succ, b_small, b_bsum;
    uint32_t *d_row = b_row.alloc<uint32_t>(st, V + 1), *d_adj = b_adj.alloc<uint32_t>(st, E), *d_succ = b_succ.alloc<uint32_t>(st, E);
    unsigned long long *d_small = b_small.alloc<unsigned long long>(st, 8);  // [0] darts walked, [1] lo: error, [2] lo: unwalked darts / list cursor, [3..4] scan totals
    uint32_t *d_bsum = b_bsum.alloc<uint32_t>(st, scan_blocks(std::max<uint64_t>(n_brk, 1)) + 2);
    HIP_CHECK(hipMemsetAsync(d_small, 0, 64, st));
    uint32_t *d_error = reinterpret_cast<uint32_t *>(d_small + 1), *d_unwalked = reinterpret_cast<uint32_t *>(d_small + 2);
    lap("allocations");
    if (d_row0) device_build_buckets_merged(st, d_from, d_mirror, E0, E, V, d_row0, d_adj0, d_row, d_adj, zip);
    else device_build_buckets(st, d_from, E, V, d_row, d_adj, nullptr, d_succ);  // (the successor array is free until the pairing: scratch)
    lap("buckets");
    device_pairing(st, d_mirror, V, d_row, d_adj, d_succ, d_error);
    lap("pairing");
    Buf b_keep, b_flag, b_off;
    uint32_t *d_keep = b_keep.alloc<uint32_t>(st, std::max<uint64_t>(n_brk, 1)), *d_flag = b_flag.alloc<uint32_t>(st, std::max<uint64_t>(n_brk, 1));
    Buf b_wsum;
    unsigned long long *d_wsum = b_wsum.alloc<unsigned long long>(st, std::max<uint64_t>(grid_for(n_brk), 1));
    if (n_brk) {
        stretch_measure_kernel<<<grid_for(n_brk), EB, 0, st>>>(d_succ, first_brk, n_brk, d_keep, d_flag, d_wsum, d_error);
        sum_blocks_kernel<<<1, 1024, 0, st>>>(d_wsum, grid_for(n_brk), d_small);
    }
    unsigned long long h_small[8];
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if ((uint32_t)h_small[1] & 1u) MTG_DIE("device_cut_first: the graph is not Eulerian (greedytigs/mod.rs:708)");
    if ((uint32_t)h_small[1]) MTG_DIE("device_cut_first: internal error %u (the successor array is not a mirror-symmetric permutation)", (uint32_t)h_small[1]);
    lap("pass 1: stretch lengths + marks");
    const uint64_t unwalked = first_brk - h_small[0];  // non-breaking darts no walker passed
    if (stats) { stats->breaking_free_darts = unwalked; stats->stretches = n_brk; }
    SpliceResult sp;
    if (unwalked) {
        // the trails without a breaking dart: spliced in on the host when they are few, else the closed walks of euler_device.hip
        if (dbg) std::fprintf(stderr, "[mtg] cut first: %llu of %llu darts lie on trails without a breaking dart\n", (unsigned long long)unwalked, (unsigned long long)E);
        if (unwalked > std::max<uint64_t>(1u << 16, E / 64) || unwalked > (4u << 20)) return false;
        const uint32_t nf = (uint32_t)unwalked;
        Buf b_list, b_hs, b_hv, b_hd;
        uint32_t *d_list = b_list.alloc<uint32_t>(st, nf), *d_hs = b_hs.alloc<uint32_t>(st, nf), *d_hv = b_hv.alloc<uint32_t>(st, nf), *d_hd = b_hd.alloc<uint32_t>(st, nf);
        unwalked_rep_kernel<<<grid_for((first_brk + 7) / 8), EB, 0, st>>>(d_succ, first_brk, 1u << 16, d_unwalked, nf, d_list, d_error);
        std::vector<uint32_t> fd(nf), fs(nf), fv(nf), ro(nf + 1);
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if ((uint32_t)h_small[1] & 32u) {  // a trail without a breaking dart longer than one thread should walk
            if (dbg) std::fprintf(stderr, "[mtg] cut first: a trail without a breaking dart has more than 65536 darts\n");
            return false;
        }
        if ((uint32_t)h_small[1]) MTG_DIE("device_cut_first: internal error %u while listing the unwalked trails", (uint32_t)h_small[1]);
        if ((uint32_t)h_small[2] != nf) MTG_DIE("device_cut_first: internal error (%u darts on the unwalked trails found, %u counted)", (uint32_t)h_small[2], nf);
        // (the list comes in atomic order: sorted, so that everything that follows is a function of the graph alone)
        HIP_CHECK(hipMemcpyAsync(fd.data(), d_list, (uint64_t)nf * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::sort(fd.begin(), fd.end());
        HIP_CHECK(hipMemcpyAsync(d_list, fd.data(), (uint64_t)nf * 4, hipMemcpyHostToDevice, st));
        unwalked_head_kernel<<<grid_for(nf), EB, 0, st>>>(d_list, nf, d_succ, d_from, d_row, d_hs, d_hv, d_hd);
        HIP_CHECK(hipMemcpyAsync(fs.data(), d_hs, (uint64_t)nf * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(fv.data(), d_hd, (uint64_t)nf * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        uint64_t tot = 0;
        for (uint32_t i = 0; i < nf; i++) { ro[i] = (uint32_t)tot; tot += fv[i]; }
        if (tot >= 0xFFFFFFFFull) return false;
        ro[nf] = (uint32_t)tot;
        Buf b_ro, b_od, b_op;
        uint32_t *d_ro = b_ro.alloc<uint32_t>(st, nf + 1), *d_od = b_od.alloc<uint32_t>(st, std::max<uint64_t>(tot, 1)), *d_op = b_op.alloc<uint32_t>(st, std::max<uint64_t>(tot, 1));
        HIP_CHECK(hipMemcpyAsync(d_ro, ro.data(), (uint64_t)(nf + 1) * 4, hipMemcpyHostToDevice, st));
        unwalked_adj_kernel<<<grid_for(nf), EB, 0, st>>>(d_hv, d_ro, nf, d_succ, d_row, d_adj, d_od, d_op);
        std::vector<uint32_t> od(tot), op(tot);
        if (tot) {
            HIP_CHECK(hipMemcpyAsync(od.data(), d_od, tot * 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipMemcpyAsync(op.data(), d_op, tot * 4, hipMemcpyDeviceToHost, st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
        auto dummy_weights = [&](const std::vector<uint32_t> &darts) -> std::vector<uint32_t> {  // (the matched dummies of components without any breaking dart)
            std::vector<uint32_t> w(darts.size());
            if (darts.empty()) return w;
            Buf b_d, b_w;
            uint32_t *d_d = b_d.alloc<uint32_t>(st, darts.size()), *d_w = b_w.alloc<uint32_t>(st, darts.size());
            HIP_CHECK(hipMemcpyAsync(d_d, darts.data(), darts.size() * 4, hipMemcpyHostToDevice, st));
            dummy_weight_kernel<<<grid_for(darts.size()), EB, 0, st>>>(d_d, (uint32_t)darts.size(), (uint32_t)E0, d_pw, d_w);
            HIP_CHECK(hipMemcpyAsync(w.data(), d_w, darts.size() * 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            return w;
        };
        sp = splice_breaking_free(fd, fs, ro, od, op, E0, first_brk, dummy_weights);
        if (!sp.patch_dart.empty()) {
            const uint32_t np = (uint32_t)sp.patch_dart.size(), nt = (uint32_t)sp.touched.size();
            Buf b_pd, b_pv, b_td;
            uint32_t *d_pd = b_pd.alloc<uint32_t>(st, np), *d_pv = b_pv.alloc<uint32_t>(st, np), *d_td = b_td.alloc<uint32_t>(st, std::max<uint32_t>(nt, 1));
            HIP_CHECK(hipMemcpyAsync(d_pd, sp.patch_dart.data(), (uint64_t)np * 4, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(d_pv, sp.patch_value.data(), (uint64_t)np * 4, hipMemcpyHostToDevice, st));
            if (nt) HIP_CHECK(hipMemcpyAsync(d_td, sp.touched.data(), (uint64_t)nt * 4, hipMemcpyHostToDevice, st));
            succ_patch_kernel<<<grid_for(np), EB, 0, st>>>(d_succ, d_pd, d_pv, np);
            if (nt) stretch_fix_kernel<<<grid_for(nt), EB, 0, st>>>(d_succ, d_td, nt, first_brk, d_keep, d_flag, d_error);
            HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if ((uint32_t)h_small[1]) MTG_DIE("device_cut_first: internal error %u after the splices", (uint32_t)h_small[1]);
        }
        if (stats) { stats->spliced_trails = sp.spliced; stats->cyclic_tigs = sp.cyc_limits.size(); }
        lap("trails without a breaking dart spliced in");
    }
    b_row.release();
    b_adj.release();
    uint32_t *d_off = b_off.alloc<uint32_t>(st, std::max<uint64_t>(n_brk, 1));
    uint32_t *d_tot = reinterpret_cast<uint32_t *>(d_small + 3);
    // (exclusive scans: the emitted lengths -> edge offsets; the flags, in place -> tig indices)
    scan_u32<uint32_t>(st, d_keep, n_brk, d_off, d_bsum, d_tot);
    scan_u32<uint32_t>(st, d_flag, n_brk, d_flag, d_bsum, d_tot + 2);
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const uint64_t n_kept_s = (uint32_t)h_small[3], n_tigs_s = (uint32_t)h_small[4];  // from the stretches; the cyclic tigs of the host follow them
    const uint64_t n_kept = n_kept_s + sp.cyc_edges.size(), n_tigs = n_tigs_s + sp.cyc_limits.size();
    if ((n_kept + sp.cyc_dropped) * 2 != first_brk)
        MTG_DIE("device_cut_first: internal error (the tigs hold %llu of %llu biedges)", (unsigned long long)(n_kept + sp.cyc_dropped), (unsigned long long)(first_brk / 2));
    lap("selection + scans");
    uint32_t *d_te = b_te.alloc<uint32_t>(st, std::max<uint64_t>(n_kept, 1)), *d_tl = b_tl.alloc<uint32_t>(st, std::max<uint64_t>(n_tigs, 1));
    if (n_brk) stretch_write_kernel<<<grid_for(n_brk), EB, 0, st>>>(d_succ, first_brk, n_brk, d_keep, d_off, d_flag, d_te, d_tl);
    HIP_CHECK(hipGetLastError());
    if (!sp.cyc_limits.empty()) {
        for (uint32_t &l : sp.cyc_limits) l += (uint32_t)n_kept_s;
        HIP_CHECK(hipMemcpyAsync(d_te + n_kept_s, sp.cyc_edges.data(), sp.cyc_edges.size() * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(d_tl + n_tigs_s, sp.cyc_limits.data(), sp.cyc_limits.size() * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    lap("pass 2: tigs written");
    *n_kept_out = n_kept;
    *n_tigs_out = n_tigs;
    return true;
}

}  // namespace mtg
