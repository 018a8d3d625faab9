// device_build.hip -- the device graph: family blocks built ON the GPU from the uploaded edge arrays, the goal-directed lower bounds
// (k <= 255), the library's device memory (arena reservations), creation and release of a Device.
// Part of the device stage; shared types: device_internal.hpp.
#include "device_internal.hpp"

namespace mtg {

// ------------------------------------------------------------------------------------------------
// Device graph build: original edge arrays -> family blocks (+ CSR spill for nodes with more than 4 out-edges)
// Out-edges are kept in edge-id order per node (slots are claimed in any order, then sorted by edge id), so the content is
// independent of thread timing.
// ------------------------------------------------------------------------------------------------
// (the count and the edge's slot at its from-node come from the same atomic: `rank` is read back, coalesced, by build_fill_kernel --
// one random atomic per edge instead of two: 18.5 -> see DESIGN 4.1)
__global__ void build_count_kernel(const uint32_t *e_from, uint64_t n_edges, uint32_t *odeg, uint32_t *rank) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) rank[e] = atomicAdd(&odeg[e_from[e]], 1u);
}
__global__ void build_ext_need_kernel(const uint32_t *odeg, uint64_t n_nodes, uint32_t *ext_need) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_nodes) ext_need[n] = odeg[n] > 4 ? odeg[n] : 0u;
}
// every edge leaves its EDGE ID in the slot of its from-node the counting pass gave it (inline slot or spill position)
__global__ void build_fill_kernel(const uint32_t *e_from, uint64_t n_edges, const uint32_t *odeg, const unsigned long long *ext_off,
                                  const uint32_t *rank, NodeBlock *blocks, uint32_t *ext_col) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const uint32_t f = e_from[e];
    const uint32_t slot = rank[e];
    if (odeg[f] > 4) ext_col[ext_off[f] + slot] = (uint32_t)e;
    else blocks[f].nbr[slot] = (uint32_t)e;
}
// per node: edge ids ascending -> (neighbour, clamped weight); degree, own class flag. The head of edge e is mirror(from(e ^ 1)) (the
// mirror edge runs mirror(to) -> mirror(from), clib.rs:244-248) and both edges of unitig u = e >> 1 weigh w_unitig[u], so neither a
// head array nor per-edge weights are uploaded. adj0 (optional): the node's out-edges in ascending id at row0[n] -- the buckets of
// the original darts the finishing stages keep with the graph (finish_device.hip), a by-product of the sort here.
__global__ void build_nodes_kernel(uint64_t n_nodes, const uint32_t *odeg, const uint32_t *mirror, const unsigned long long *ext_off,
                                   const uint32_t *e_from, const uint16_t *w_unitig, NodeBlock *blocks, uint32_t *ext_col, uint16_t *ext_w,
                                   const uint32_t *row0, uint32_t *adj0) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t dg = odeg[n];
    const uint32_t m = mirror[n];
    const NodeClass c = classify_node(dg, m == n ? 0u : odeg[m], m == n);
    NodeBlock b;
    uint32_t *bw = reinterpret_cast<uint32_t *>(&b);
#pragma unroll
    for (int i = 0; i < 16; i++) bw[i] = 0;
    b.flags = c.cls & F_TARGET;
    b.cmeta = 0x00F0;  // no child embedded (build_children_kernel fills this in)
    for (int j = 0; j < 4; j++) b.w[j] = 0xFFFFu;
    for (int t = 0; t < GSLOTS; t++) b.gw[t] = 0xFFFFu;
    if (dg <= 4) {
        uint32_t ids[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (uint32_t j = 0; j < dg; j++) ids[j] = blocks[n].nbr[j];
        auto cswap = [&](int x, int y) { if (ids[x] > ids[y]) { const uint32_t t = ids[x]; ids[x] = ids[y]; ids[y] = t; } };
        cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
        for (uint32_t j = 0; j < dg; j++) { b.nbr[j] = mirror[e_from[ids[j] ^ 1u]]; b.w[j] = w_unitig[ids[j] >> 1]; }
        if (adj0)
            for (uint32_t j = 0; j < dg; j++) adj0[row0[n] + j] = ids[j];
        b.deg = (uint8_t)dg;
    } else {
        const unsigned long long off = ext_off[n];
        for (uint32_t i = 1; i < dg; i++) {  // insertion sort of the spill segment by edge id
            const uint32_t key = ext_col[off + i];
            uint32_t q = i;
            while (q > 0 && ext_col[off + q - 1] > key) { ext_col[off + q] = ext_col[off + q - 1]; q--; }
            ext_col[off + q] = key;
        }
        for (uint32_t i = 0; i < dg; i++) {
            const uint32_t e = ext_col[off + i];
            if (adj0) adj0[row0[n] + i] = e;
            ext_col[off + i] = mirror[e_from[e ^ 1u]];
            ext_w[off + i] = w_unitig[e >> 1];
        }
        b.flags |= F_EXT;
        b.nbr[0] = (uint32_t)(off & 0xFFFFFFFFull);
        b.nbr[1] = (uint32_t)(off >> 32);
        b.nbr[2] = dg;
    }
    blocks[n] = b;
}
// ------------------------------------------------------------------------------------------------
// Goal-directed lower bounds (k <= 255): lb(v) = distance from v to the nearest initial in-node, 0 for an in-node, "infinite" beyond
// k - 1. A function of the graph alone, like the in-node flags. A search at node u with distance d only ever needs the successor v
// over an edge of weight w if d + w + lb(v) <= k - 1: every in-node behind v is at least that far from the source, so dropping v
// can never drop a candidate -- the lists stay exactly Dijkstra's (greedytigs/mod.rs:324-335; the reference truncates its search
// too, by target_amount). On the bench graph 7 of 10 sources have no in-node within the bound at all and the remaining searches
// visit a third of their balls.
// dist(v -> t) in G = dist(mirror t -> mirror v) in G (every edge has its mirror edge with the same weight), so ONE bounded
// multi-source search from the mirrors of the in-nodes gives D(x) = lb(mirror x). Dial's buckets without queues: weights are >= 1,
// so the nodes with D == r are final in round r; round r relaxes their out-edges with atomicMin. k - 1 rounds of a streaming pass
// over D plus one 32-byte gather per reached node, once per device graph.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t LB_INF = 0xFFFFFFFFu;
__global__ void lb_init_kernel(const uint32_t *odeg, const uint32_t *mirror, uint64_t n_nodes, uint32_t *D) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t m = mirror[n];
    const NodeClass c = classify_node(odeg[m], m == n ? 0u : odeg[n], m == n);  // class of mirror(n)
    D[n] = (c.cls & F_TARGET) ? 0u : LB_INF;
}
// (first halves still hold plain 16-bit weights here: build_lb_kernel rewrites them afterwards)
__global__ void lb_round_kernel(const NodeBlock *blocks, const uint32_t *ext_col, const uint16_t *ext_w, uint64_t n_nodes, uint32_t r, uint32_t K1,
                                uint32_t *D) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes || D[n] != r) return;
    const uint4 *rp = reinterpret_cast<const uint4 *>(blocks + n);
    const uint4 lo = rp[0], hi = rp[1];
    const uint32_t nb[4] = {lo.x, lo.y, lo.z, lo.w};
    const uint32_t wt[4] = {hi.x & 0xFFFFu, hi.x >> 16, hi.y & 0xFFFFu, hi.y >> 16};
    if ((hi.z >> 8) & F_EXT) {
        const uint64_t b = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
        for (uint32_t e = 0; e < lo.z; e++) {
            const uint32_t nd = r + ext_w[b + e];
            if (nd <= K1) atomicMin(&D[ext_col[b + e]], nd);
        }
    } else {
        const uint32_t dg = hi.z & 0xFFu;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t nd = r + wt[j];
            if (j < dg && nd <= K1) atomicMin(&D[nb[j]], nd);
        }
    }
}
__global__ void lb_mirror_kernel(const uint32_t *mirror, const uint32_t *D, uint64_t n_nodes, uint8_t *lb8) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t v = D[mirror[n]];
    lb8[n] = (uint8_t)(v < 255u ? v : 255u);
}
// Pass 1: lbx[n] = lb(n) | lb+(n) << 8, where lb+(n) = min over the out-edges n -> c of weight + lb(c) is the distance from n to the
// nearest in-node BEYOND n (for a node that is not an in-node itself lb+ = lb; for an in-node lb = 0 and lb+ says what a search that
// has recorded n still needs n's block for). bit 31 of odeg[n] (ODEG_REACH) is set iff lb+(n) <= k - 1: a source without that has an empty candidate list and
// is never searched. (First halves still hold plain 16-bit weights here.)
__global__ void build_lbx_kernel(uint64_t n_nodes, const NodeBlock *blocks, const uint32_t *ext_col, const uint16_t *ext_w, const uint8_t *lb8, uint32_t K1,
                                 uint16_t *lbx, uint32_t *odeg) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t *me = reinterpret_cast<const uint32_t *>(blocks + n);
    const uint32_t meta = me[6];
    uint32_t best = 255u;
    if ((meta >> 8) & F_EXT) {
        const uint64_t b = (uint64_t)me[0] | ((uint64_t)me[1] << 32);
        for (uint32_t e = 0; e < me[2]; e++) best = min(best, (uint32_t)ext_w[b + e] + lb8[ext_col[b + e]]);
    } else {
        const uint32_t dg = meta & 0xFFu;
        for (uint32_t j = 0; j < dg; j++) {
            const uint32_t w = (me[4 + (j >> 1)] >> ((j & 1u) * 16)) & 0xFFFFu;  // <= k <= 255
            best = min(best, w + lb8[me[j]]);
        }
    }
    lbx[n] = (uint16_t)(lb8[n] | (best << 8));
    if (best <= K1) odeg[n] |= ODEG_REACH;  // (read by the classification with the degree; every other reader of odeg runs before this kernel or masks)
}
// second half of every block: the children's in-node flags and, while they fit, the children's out-edges.
// W8 = false: plain 16-bit weights (k > 255, or a device graph built without lower bounds).
// W8 = true: the 8:8 format, written in the same pass (round 5: one kernel where build_lb_kernel rewrote the first halves and this
// kernel then read the children's rewritten halves). Own first half: low byte the weight, high byte weight + lb+(child) -- what a
// search needs the CHILD'S BLOCK for; whether the child is an in-node itself (lb = 0) travels in cmeta bit j, so the parent's step
// records that candidate and the child's gather only happens when something lies beyond it ("leaf" in-nodes, a quarter of all
// visits on the bench graph, cost no gather). Second half: path weight | path weight + lb+(grandchild), both saturated at 255 (> any
// bound of this format), and the grandchild's in-node flag in cmeta bit 8 + t. lb and lb+ of children and grandchildren come from
// lbx[] (build_lbx_kernel), never from another node's block: while this kernel runs a node's first half is read by its parents'
// threads and rewritten by its own, and either version decodes to the same neighbours, degree, flags and LOW bytes (a plain
// 16-bit weight is <= k <= 255, an unused slot is 0xFFFF in both formats; the three words are written whole). A spilled adjacency
// keeps plain weights (no pruning behind such a node).
template <bool W8>
__global__ void build_children_kernel(uint64_t n_nodes, NodeBlock *blocks, const uint16_t *lbx) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t *me = reinterpret_cast<const uint32_t *>(blocks + n);
    const uint32_t meta = me[6];
    if ((meta >> 8) & F_EXT) return;
    const uint32_t dg = meta & 0xFFu;
    const uint32_t nb[4] = {me[0], me[1], me[2], me[3]};
    const uint32_t w45[2] = {me[4], me[5]};
    uint32_t cmeta = 0, used = 0;
    uint32_t gn[GSLOTS];
    uint32_t gwt[GSLOTS];
    uint32_t slot[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
#pragma unroll
    for (int i = 0; i < GSLOTS; i++) { gn[i] = 0; gwt[i] = 0xFFFFu; }
    bool open = true;  // children are embedded in order while they fit
    for (uint32_t j = 0; j < 4; j++) {
        if (j >= dg) continue;
        const uint32_t *ch = reinterpret_cast<const uint32_t *>(blocks + nb[j]);  // first half only
        const uint32_t cmt = ch[6];
        const uint32_t cflags = (cmt >> 8) & 0xFFu, cdeg = cmt & 0xFFu;
        const uint32_t sj = (w45[j >> 1] >> ((j & 1u) * 16)) & 0xFFFFu;
        const uint32_t wj = W8 ? (sj & 0xFFu) : sj;
        if (cflags & F_TARGET) cmeta |= 1u << j;
        if constexpr (W8) slot[j] = (min(255u, wj + ((uint32_t)lbx[nb[j]] >> 8)) << 8) | wj;
        if (open && !(cflags & F_EXT) && used + cdeg <= (uint32_t)GSLOTS) {
            for (uint32_t t = 0; t < cdeg; t++) {
                const uint32_t gc = ch[t];
                gn[used + t] = gc;
                const uint32_t st = (ch[4 + (t >> 1)] >> ((t & 1u) * 16)) & 0xFFFFu;
                if constexpr (W8) {
                    const uint32_t x = lbx[gc], wt = st & 0xFFu;
                    gwt[used + t] = min(255u, wj + wt) | (min(255u, wj + wt + (x >> 8)) << 8);
                    if ((x & 0xFFu) == 0) cmeta |= 1u << (8 + used + t);  // the grandchild is an in-node
                } else {
                    const uint32_t sum = wj + st;
                    gwt[used + t] = sum < 0xFFFFu ? sum : 0xFFFFu;
                }
            }
            used += cdeg;
        } else {
            open = false;
            cmeta |= 16u << j;
        }
    }
    // The WHOLE block is stored, the unchanged neighbour words included (four 16-byte stores per thread): a line that leaves the L2
    // partly written is a masked write, which the ECC-protected HBM turns into a read-modify-write (round 5; the same finding as the
    // claim replay's records, replay_kernels.inc). Readers of this node's first half see old or new words, which decode alike (above).
    uint32_t o[16];
    o[0] = nb[0]; o[1] = nb[1]; o[2] = nb[2]; o[3] = nb[3];
    o[4] = W8 ? (slot[0] | (slot[1] << 16)) : w45[0];
    o[5] = W8 ? (slot[2] | (slot[3] << 16)) : w45[1];
    o[6] = (meta & 0xFFFFu) | (cmeta << 16);
#pragma unroll
    for (int i = 0; i < GSLOTS; i++) o[7 + i] = gn[i];
#pragma unroll
    for (int i = 0; i < GSLOTS / 2; i++) o[13 + i] = gwt[2 * i] | (gwt[2 * i + 1] << 16);
    uint4 *out = reinterpret_cast<uint4 *>(blocks + n);
#pragma unroll
    for (int q = 0; q < 4; q++) out[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

// ---- public (device.hpp) ----
int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Bytes of device memory a whole call on a graph of V nodes and E original edges has in use at its peak (device graph with the build's
// scratch, or the device graph beside the search and replay arrays, or the finish's dart arrays): the size of the arena chunk a
// first call reserves. Calibrated on G-csr graphs from 2^20 to 2^30 edges (MTG_DEBUG prints the arena's peak at the end of a call); an
// estimate that is too small costs a second chunk, one that is too large memory the driver has to map for nothing.
size_t device_call_bytes_estimate(uint64_t V, uint64_t E, uint64_t k) {
    (void)k;
    return (size_t)(V * 106 + E * 14) + (64u << 20);
}

// Kernel code objects load lazily, at the first launch of a kernel of their translation unit (3-10 ms each on a cold process); asking
// for a kernel's attributes loads them without launching anything.
// The first COOPERATIVE launch of a process costs 6 ms more than any later one (measured on the claim replay's rounds kernel, 10.7
// against 4.8 ms): an empty one is issued here, on the helper thread, on the stream the replay will use.
__global__ void coop_warm_kernel(unsigned) {}
void device_warm_device_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(build_count_kernel));
    device_warm_classify_unit(&a);  // (one kernel per translation unit of the device stage: their code objects load)
    device_warm_sssp_unit(&a);
    device_warm_replay_unit(&a);
    device_warm_pairs_unit(&a);
    unsigned zero = 0;
    void *args[] = {&zero};
    (void)hipLaunchCooperativeKernel(reinterpret_cast<const void *>(coop_warm_kernel), dim3(1), dim3(64), args, 0, nullptr);
}
void device_warm_finish_kernels();  // finish_device.hip
void device_warm_euler_kernels();   // euler_device.hip

void device_arena_stats(int device_id, uint64_t out[4]) {
    hu::DeviceArena &a = hu::device_arena(device_id);
    std::lock_guard<std::mutex> lock(a.m);
    out[0] = a.chunk_bytes; out[1] = a.live_bytes; out[2] = a.peak_bytes; out[3] = a.n_chunk_allocs;
}
void device_arena_reset_peak(int device_id) {
    hu::DeviceArena &a = hu::device_arena(device_id);
    std::lock_guard<std::mutex> lock(a.m);
    a.peak_bytes = a.live_bytes;
}
static std::atomic<int> g_default_device{0};
void device_set_default(int device_id) { g_default_device.store(device_id); }
int device_get_default() { return g_default_device.load(); }

// Called when a host graph of V nodes and E edges comes into being (graph_build.cpp): on a helper thread, beside the host's own work,
// the HIP runtime starts, the code objects load, and the arena of the default device gets its chunk for the call that will follow --
// ONE hipMalloc, whose cost (nothing to 60 ms per GB depending on the box) no stage then waits for. Small graphs reserve nothing.
// A host-only constructor does not know which GPU the computation will name, so what it reserves is provisional: the device is
// remembered, and a call that computes elsewhere gives the chunk back when it ends (device_drop_foreign_reservation);
// mtg_set_reserve_ahead(0) turns the whole thing off for callers that want a constructor without GPU side effects. The helper threads
// are joinable: each new reservation joins the ones that are through, and the library's teardown joins the rest.
static std::atomic<int> g_reserve_ahead{1};
static std::atomic<int> g_reserved_device{-1};  // device of the last provisional reservation no call has confirmed yet
void device_set_reserve_ahead(int on) { g_reserve_ahead.store(on ? 1 : 0); }
namespace {
struct ReserveThreads {
    std::mutex m;
    std::vector<std::pair<std::thread, std::shared_future<void>>> th;
    void reap(bool all) {
        std::vector<std::thread> done;
        {
            std::lock_guard<std::mutex> lock(m);
            for (size_t i = 0; i < th.size();) {
                if (all || th[i].second.wait_for(std::chrono::seconds(0)) == std::future_status::ready) {
                    done.push_back(std::move(th[i].first));
                    th.erase(th.begin() + (long)i);
                } else i++;
            }
        }
        for (std::thread &t : done) if (t.joinable()) t.join();
    }
    ~ReserveThreads() { reap(true); }  // library teardown: no helper thread is left inside HIP when the process goes on to exit
};
ReserveThreads &reserve_threads() {
    static ReserveThreads r;
    return r;
}
}  // namespace
void device_reserve_async(uint64_t V, uint64_t E, int device_id) {
    const size_t bytes = device_call_bytes_estimate(V, E, 31);
    if (bytes < (256u << 20)) return;
    const bool provisional = device_id < 0;
    if (provisional && !g_reserve_ahead.load()) return;
    const int dev = device_id >= 0 ? device_id : device_get_default();
    std::promise<void> done;
    std::shared_future<void> fut = done.get_future().share();
    {
        hu::DeviceArena &arena = hu::device_arena(dev);
        std::lock_guard<std::mutex> lock(arena.m);
        // (one at a time; a reservation nobody had to wait for is over by now)
        if (arena.pending.valid() && arena.pending.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return;
        arena.pending = fut;
    }
    if (provisional) g_reserved_device.store(dev);
    ReserveThreads &rt = reserve_threads();
    rt.reap(false);
    std::thread t([dev, bytes](std::promise<void> p) {
        if (device_count() > dev && hipSetDevice(dev) == hipSuccess) {
            hu::device_arena(dev).reserve(bytes);
            (void)hu::finish_stream(dev);
            {   // the pinned ring of the sliced transfers (4 x 16 MB of page-locked memory: 10 ms the first upload would pay)
                hu::TransferRing &r = hu::transfer_ring(dev);
                std::lock_guard<std::mutex> lock(r.m);
                r.ready();
            }
            device_warm_device_kernels();
            device_warm_finish_kernels();
            device_warm_euler_kernels();
        }
        (void)hipGetLastError();
        p.set_value();
    }, std::move(done));
    std::lock_guard<std::mutex> lock(rt.m);
    rt.th.emplace_back(std::move(t), fut);
}
// A call that computed on `used` (n of them): a provisional reservation that sits on another device goes back to the driver.
void device_drop_foreign_reservation(const int *used, int n) {
    const int dev = g_reserved_device.exchange(-1);
    if (dev < 0) return;
    for (int i = 0; i < n; i++) if (used[i] == dev) return;
    hu::DeviceArena &arena = hu::device_arena(dev);
    {
        std::unique_lock<std::mutex> lock(arena.m);
        if (arena.pending.valid()) {
            std::shared_future<void> f = arena.pending;
            arena.pending = std::shared_future<void>();
            lock.unlock();
            f.wait();
        }
    }
    arena.release_free_chunks(true);
}

// The goal-directed lower bounds of a device graph whose blocks hold plain weights (k <= 255): k - 1 rounds over a 32-bit distance
// array (lb), one pass for lb+ and the reach flags, and the blocks rewritten into the 8:8 format. A function of the graph alone;
// what it costs (HIP events: device_lower_bounds_ms) is the price of the pruned search, paid once per device graph -- a caller that
// searches once is better off without (mtg_compute_tigs_cfg builds none: 4.5 ms of full-ball search against 2 + 12 ms).
static void build_lower_bounds(Device *d, hipStream_t st) {
    const uint64_t V = d->V;
    if (!V || d->w8) return;
    const unsigned vb = (unsigned)((V + 255) / 256);
    uint32_t *d_D = nullptr;
    uint8_t *d_lb8 = nullptr;
    uint16_t *d_lbx = nullptr;
    hu::device_malloc(&d_D, V * 4);
    hu::device_malloc(&d_lb8, V);
    hu::device_malloc(&d_lbx, V * 2);
    HIP_CHECK(hipEventRecord(d->ev0, st));
    hipLaunchKernelGGL(lb_init_kernel, dim3(vb), dim3(256), 0, st, d->d_odeg, d->d_mirror, V, d_D);
    for (uint32_t r = 0; r < d->K1; r++)
        hipLaunchKernelGGL(lb_round_kernel, dim3(vb), dim3(256), 0, st, d->d_recs, d->d_ext_col, d->d_ext_w, V, r, d->K1, d_D);
    hipLaunchKernelGGL(lb_mirror_kernel, dim3(vb), dim3(256), 0, st, d->d_mirror, d_D, V, d_lb8);
    hipLaunchKernelGGL(build_lbx_kernel, dim3(vb), dim3(256), 0, st, V, d->d_recs, d->d_ext_col, d->d_ext_w, d_lb8, d->K1, d_lbx, d->d_odeg);
    hipLaunchKernelGGL(build_children_kernel<true>, dim3(vb), dim3(256), 0, st, V, d->d_recs, (const uint16_t *)d_lbx);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(d->ev1, st));
    HIP_CHECK(hipStreamSynchronize(st));
    d->lower_bounds_ms = elapsed_ms(d);
    d->w8 = true;
    hu::device_free(d_D);
    hu::device_free(d_lb8);
    hu::device_free(d_lbx);
}
// ... for a device graph that was built without them (a caller that turns out to iterate): the searchable-source list of the
// classification changes with the reach flags, so a classification that was there is repeated.
void device_build_lower_bounds(Device *d, void *stream) {
    HIP_CHECK(hipSetDevice(d->dev));
    if (d->k > 255 || d->w8) return;
    if (!d->d_recs) MTG_DIE("mtg_device_build_lower_bounds: this device copy has given its search arrays back");
    build_lower_bounds(d, (hipStream_t)stream);
    if (d->classified) (void)device_classify(d, stream);
}
double device_lower_bounds_ms(const Device *d) { return d->lower_bounds_ms; }
bool device_has_lower_bounds(const Device *d) { return d->w8; }

Device *device_create(const HostGraph &g, uint64_t k, int device_id, bool lower_bounds) {
    if (k < 1) MTG_DIE("k must be >= 1");
    if (k > 0xFFFFFFFFull) MTG_DIE("k = %llu is not supported by the device stage (32-bit distances)", (unsigned long long)k);
    if (k > 65535) {
        // Edge weights live in 16 bits, clamped to min(w, k) (an edge of weight >= k can never lie on a path within k - 1). With
        // k beyond 16 bits the clamp no longer fits, so every weight itself has to: true for any unitig set (a weight is a
        // number of k-mers of a unitig). Distances beyond 15 / 21 bits skip the enumeration / cooperative levels (run_levels).
        parallel_ranges(g.n_original_edges / 2, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t u = lo; u < hi; u++)
                if (g.w_biedge[u] > 65534)
                    MTG_DIE("k = %llu with a unitig of %llu k-mers: beyond k = 65535 the device stage needs every weight below 65535",
                            (unsigned long long)k, (unsigned long long)g.w_biedge[u]);
        });
    }
    // weights: 16 bits per UNITIG, clamped to min(w, k) (both edges of a unitig weigh the same, host_graph.hpp); a unitig without
    // k-mers aborts here -- the bounded search, its lower bounds and the (distance, node) pop order of the reference's heap all need
    // weights >= 1 (the reference computes weight = len + 1 - k >= 1, bin.rs:369-376)
    const uint64_t U = g.n_original_edges / 2;
    PodVec<uint16_t> wclamp(U);
    parallel_ranges(U, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t u = lo; u < hi; u++) {
            const uint64_t w = g.w_biedge[u];
            if (w == 0)
                MTG_DIE("unitig %llu has weight 0: the bounded search needs weights >= 1 (the reference computes "
                        "weight = len + 1 - k >= 1, bin.rs:369-376)", (unsigned long long)u);
            wclamp[u] = (uint16_t)std::min<uint64_t>(w, k);
        }
    });
    if (device_count() <= device_id) MTG_DIE("no MI355X/HIP device %d available; libmatchtigs has no CPU path", device_id);
    HIP_CHECK(hipSetDevice(device_id));
    Device *d = new Device();
    d->dev = device_id;
    d->k = k;
    if (MTG_POLICY_BOUND_EXCLUSIVE && k < 2) MTG_DIE("k must be >= 2 under the exclusive-bound policy");
    d->K1 = (uint32_t)mtg_policy_search_bound(k);  // the largest distance a target is found at: policy P2 (mtg_policy.h)
    d->V = g.node_count();
    d->E0 = g.n_original_edges;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    d->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    // The device graph is built ON the GPU from the ORIGINAL edges only (the search runs before any dummy edge exists, :678):
    // the host uploads from / to / clamped weight / mirror, the build kernels make the family blocks.
    const uint64_t V = d->V, E = g.n_original_edges;
    struct {
        const bool on = std::getenv("MTG_DEBUG") != nullptr;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        void lap(const char *what) {
            if (!on) return;
            (void)hipDeviceSynchronize();
            const auto n = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[mtg] device_create: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
            t = n;
        }
    } dl;
    dl.lap("clamped weights (host)");
    hipStream_t st = nullptr;
    // One chunk of the device's arena for the whole call unless an earlier call (or the reservation the graph's construction started
    // on a helper thread, device_reserve_async) left one: every allocation below and in the stages that follow is a range of it.
    const size_t call_bytes = device_call_bytes_estimate(V, E, k);
    uint32_t *d_from = nullptr, *d_fill = nullptr, *d_need = nullptr, *d_row0 = nullptr, *d_adj0 = nullptr;
    uint16_t *d_w = nullptr;
    unsigned long long *d_ext_off = nullptr;
    // the long-lived arrays first (they collect at the front of the chunk), the build's scratch after them
    hu::device_malloc(&d->d_recs, std::max<uint64_t>(V, 1) * sizeof(NodeBlock), call_bytes);
    hu::device_malloc(&d->d_odeg, std::max<uint64_t>(V, 1) * 4);
    hu::device_malloc(&d->d_mirror, std::max<uint64_t>(V, 1) * 4);
    // (class bytes, multiplicities, out-node list: taken by the first classification, once the build's scratch arrays are back)
    d->n_cls_blocks = (V + CLS_NODES - 1) / CLS_NODES;
    hu::device_malloc(&d->d_block_counts, std::max<uint64_t>(d->n_cls_blocks, 1) * 2 * 4);  // source counts | positive multiplicities per block
    hu::device_malloc(&d->d_act_blocks, std::max<uint64_t>(d->n_cls_blocks, 1) * 4);
    hu::device_malloc(&d->d_act_total, 8);
    hu::device_malloc(&d->d_counters, C_COUNT * sizeof(unsigned long long));
    hu::device_malloc(&d_from, std::max<uint64_t>(E, 1) * 4);
    // the buckets of the original darts by from-node (row0[V + 1], adj0[E]: what the finishing stages keep with the graph) fall out
    // of build_nodes_kernel's sort; kept while they are small next to the stand-ins of BASELINE configs[4] (finish_device.hip)
    const bool want_buckets = E && (V + 1 + E) * 4 <= (8ull << 30) && !g.device_cache && !(hu::finish_tuning().flags.load() & hu::FT_NO_EDGE_CACHE);
    if (want_buckets) {
        hu::device_malloc(&d_row0, (V + 1) * 4);
        hu::device_malloc(&d_adj0, E * 4);
    }
    hu::device_malloc(&d_w, std::max<uint64_t>(U, 1) * 2);
    hu::device_malloc(&d_fill, std::max<uint64_t>(E, 1) * 4);  // rank of every edge among the out-edges of its from-node (arrival order of the counting pass)
    hu::device_malloc(&d_need, std::max<uint64_t>(V, 1) * 4);
    hu::device_malloc(&d_ext_off, std::max<uint64_t>(V, 1) * 8);
    HIP_CHECK(hipHostMalloc(&d->h_counters, C_COUNT * sizeof(unsigned long long)));
    HIP_CHECK(hipEventCreate(&d->ev0));
    HIP_CHECK(hipEventCreate(&d->ev1));
    for (auto &e : d->ev_r) HIP_CHECK(hipEventCreate(&e));
    dl.lap("allocations");
    // (through the pinned ring: host threads fill the next slice while one crosses PCIe -- the runtime stages a pageable upload on one
    // thread. from + mirror + 2 bytes per unitig: the heads are mirror(from(e ^ 1)), 6.3 B per edge + 4 B per node in all)
    // `from` first: the counting pass and the scans over its degrees (12 ms of kernels at 2^27) need nothing else and run on a stream
    // of their own (non-blocking: `st` is the legacy default stream, which every ordinary stream waits for) while the weights and the
    // mirror array (0.49 GB, 10 ms of PCIe) come up on `st`.
    if (E) hu::upload_sliced(d_from, g.e_from.data(), E * 4, st, device_id);
    dl.lap("upload of from");
    const unsigned eb = (unsigned)((E + 255) / 256), vb = (unsigned)((V + 255) / 256);
    uint64_t ext_total = 0;
    uint32_t *d_bs = nullptr;
    hipStream_t bs = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&bs, hipStreamNonBlocking));
    HIP_CHECK(hipMemsetAsync(d->d_odeg, 0, std::max<uint64_t>(V, 1) * 4, bs));
    if (V) {
        if (E) hipLaunchKernelGGL(build_count_kernel, dim3(eb), dim3(256), 0, bs, d_from, E, d->d_odeg, d_fill);
        hipLaunchKernelGGL(build_ext_need_kernel, dim3(vb), dim3(256), 0, bs, d->d_odeg, V, d_need);
        HIP_CHECK(hipGetLastError());
        scan_u32(d, bs, d->replay, d_need, V, d_ext_off, &d->d_counters[C_OVF_LIST]);
        if (want_buckets) {  // row0 = exclusive scan of the out-degrees
            hu::device_malloc(&d_bs, (hu::scan_blocks(V) + 2) * 4);
            hu::scan_u32<uint32_t>(bs, d->d_odeg, V, d_row0, d_bs, d_row0 + V);
        }
    }
    if (E) hu::upload_sliced(d_w, wclamp.data(), U * 2, st, device_id);
    if (V) hu::upload_sliced(d->d_mirror, g.mirror.data(), V * 4, st, device_id);  // (both calls return when the data is there)
    if (V) {
        read_counters(d, bs);
        if (d_bs) hu::device_free(d_bs);
        ext_total = d->h_counters[C_OVF_LIST];
    }
    HIP_CHECK(hipStreamSynchronize(bs));
    HIP_CHECK(hipStreamDestroy(bs));
    dl.lap("count + scans beside the uploads of weights and mirror");
    d->ext_n = ext_total;
    hu::device_malloc(&d->d_ext_col, std::max<uint64_t>(ext_total, 1) * 4);
    hu::device_malloc(&d->d_ext_w, std::max<uint64_t>(ext_total, 1) * 2);
    if (V) {
        if (E) hipLaunchKernelGGL(build_fill_kernel, dim3(eb), dim3(256), 0, st, d_from, E, d->d_odeg, d_ext_off, d_fill, d->d_recs, d->d_ext_col);
        hipLaunchKernelGGL(build_nodes_kernel, dim3(vb), dim3(256), 0, st, V, d->d_odeg, d->d_mirror, d_ext_off, d_from, d_w, d->d_recs,
                           d->d_ext_col, d->d_ext_w, d_row0, d_adj0);
        d->w8 = false;
        if (lower_bounds && k <= 255) build_lower_bounds(d, st);  // (writes the second halves too)
        else hipLaunchKernelGGL(build_children_kernel<false>, dim3(vb), dim3(256), 0, st, V, d->d_recs, (const uint16_t *)nullptr);
        HIP_CHECK(hipGetLastError());
    }
    dl.lap("build kernels (+ lower bounds)");
    // the finishing stages on this GPU start from the same arrays: from, mirror and the buckets stay with the graph
    uint32_t *d_mirror_copy = nullptr;
    hu::device_malloc(&d_mirror_copy, std::max<uint64_t>(V, 1) * 4);
    if (V) HIP_CHECK(hipMemcpyAsync(d_mirror_copy, d->d_mirror, V * 4, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    hu::edge_cache_put(g, device_id, d_from, d_mirror_copy);
    if (want_buckets) hu::edge_cache_set_buckets(g, device_id, d_row0, d_adj0);
    for (void *p : {(void *)d_w, (void *)d_fill, (void *)d_need, (void *)d_ext_off}) hu::device_free(p);
    d->graph_bytes = V * sizeof(NodeBlock) + ext_total * 6 + V * 13;
    dl.lap("edge cache + frees");
    return d;
}

// What the stages of a STEP take from the arena beside a device graph that stays (classification, search lists, claim-replay records,
// the finish's dart arrays at their peak), calibrated on G-csr 2^24 / 2^27 / 2^30 (bench.py full_size.arena: 1.77 / 13.4 / 106.5 GB):
// 60 bytes per node + 62 per original edge, + 8 %.
size_t device_step_work_bytes_estimate(uint64_t V, uint64_t E) { return (size_t)((V * 64 + E * 62) / 100 * 108) + (64u << 20); }
// mtg_device_create_opts(MTG_DEVICE_RESERVE_WORK): a caller that will step through the stages with this device graph (classify /
// search / replay / finish, again and again) takes that memory NOW, as ONE chunk, unless the arena has it free already -- the first
// step then makes no driver call (five otherwise, each of which can stall for a second on this pool: DESIGN 9). Asked for
// explicitly, so the whole-call limit of the implicit reservations does not apply; what must remain is room for the other
// allocators of the process (a sixth of the device), else nothing is reserved and the arrays come piece by piece as before.
void device_reserve_step_work(Device *d) {
    HIP_CHECK(hipSetDevice(d->dev));
    const size_t want = device_step_work_bytes_estimate(d->V, d->E0);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    hu::device_arena(d->dev).ensure_free(want, /*explicit_request=*/true, free_b > total_b / 6 ? free_b - total_b / 6 : 0);
}

void device_free(Device *d) {
    if (!d) return;
    (void)hipSetDevice(d->dev);
    void *bufs[] = {d->d_recs, d->d_odeg, d->d_cls, d->d_ext_col, d->d_ext_w, d->d_mult, d->d_mirror, d->d_out_nodes, d->d_block_counts, d->d_counters,
                    d->d_act_index, d->d_act_node, d->d_act_blocks, d->d_act_total};
    for (void *b : bufs) hu::device_free(b);
    for (int i = 0; i < 2; i++) hu::device_free(d->d_ovf[i]);
    hu::device_free(d->d_fix);
    hu::device_free(d->d_fix_dense);
    hu::device_free(d->d_fix_val);
    hu::device_free(d->d_fix_dense_val);
    ReplayWork &w = d->replay;
    void *rb[] = {w.state, w.resv[0], w.resv[1], w.touch, w.src_mirror, w.dense, w.claims, w.pending[0], w.pending[1], w.spill,
                  w.final_off, w.block_sums, w.ctl, w.out};
    for (void *b : rb) hu::device_free(b);
    (void)hipHostFree(w.h_ctl);
    (void)hipHostFree(w.h_out);
    (void)hipHostFree(d->h_counters);
    (void)hipEventDestroy(d->ev0);
    (void)hipEventDestroy(d->ev1);
    for (auto e : d->ev_r) (void)hipEventDestroy(e);
    delete d;
}

uint64_t device_graph_bytes(const Device *d) { return d->graph_bytes; }

}  // namespace mtg
