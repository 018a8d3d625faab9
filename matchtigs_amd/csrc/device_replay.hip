// device_replay.hip -- the claim loop of greedytigs/mod.rs:301-523 on the GPU: kernels in replay_kernels.inc (one cooperative launch,
// deterministic reservations), host driver, pair compaction and download. Part of the device stage; shared types:
// device_internal.hpp.
#include "device_internal.hpp"

namespace mtg {

#include "replay_kernels.inc"

// ------------------------------------------------------------------------------------------------
// GPU claim replay (replay_kernels.inc): host driver
// ------------------------------------------------------------------------------------------------
template <typename In>
static void scan_values(hipStream_t st, ReplayWork &w, In in, uint64_t n, unsigned long long *out, unsigned long long *d_total) {
    const uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nb > w.cap_blocks) {
        if (w.block_sums) hu::device_free(w.block_sums);
        hu::device_malloc(&w.block_sums, nb * 8);
        w.cap_blocks = nb;
    }
    hipLaunchKernelGGL(scan_reduce_kernel<In>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, n, w.block_sums);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, w.block_sums, nb, d_total);
    hipLaunchKernelGGL(scan_apply_kernel<In>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, n, w.block_sums, out);
    HIP_CHECK(hipGetLastError());
}
void scan_u32(Device *d, hipStream_t st, ReplayWork &w, const uint32_t *in, uint64_t n, unsigned long long *out,
              unsigned long long *d_total) {
    scan_values(st, w, ScanInU32{in}, n, out, d_total);
    (void)d;
}

// Claims for ALL n_sources classified sources (candidate arrays indexed by absolute source index, device pointers).
// Returns the number of pairs; *pairs_out (host, malloc'd) holds them in the reference's push order.
uint64_t device_replay(Device *d, void *stream, uint64_t n_sources, const uint64_t *d_cand_start, const uint32_t *d_cand_count,
                       const uint64_t *d_pool, mtg_pair **pairs_out, int *rounds_out) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    if (!d->classified || n_sources != d->n_sources) MTG_DIE("mtg_replay_claims_device: classify first; n_sources must be all sources");
    ReplayWork &w = d->replay;  // buffers are re-used across calls on this device
    const uint64_t V = d->V, S = n_sources;
    const auto t_replay_begin = std::chrono::steady_clock::now();
    d->last_wall_s[1] = d->last_wall_s[2] = 0;
    struct {  // MTG_DEBUG=1: host-side wall clock of the call's segments
        const bool on = std::getenv("MTG_DEBUG") != nullptr;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        void lap(const char *what) {
            if (!on) return;
            const auto n = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[mtg] replay: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
            t = n;
        }
    } rt;
    if (S == 0) {
        if (pairs_out) *pairs_out = (mtg_pair *)std::malloc(sizeof(mtg_pair));
        d->last_replay_rounds = 0;
        d->last_n_pairs = 0;
        if (rounds_out) *rounds_out = 0;
        return 0;
    }
    if (V > w.cap_v) {
        if (w.state) hu::device_free(w.state);
        for (int i = 0; i < 2; i++) if (w.resv[i]) { hu::device_free(w.resv[i]); w.resv[i] = nullptr; }
#if MTG_REPLAY_RECORDS
        hu::device_malloc(&w.state, ((V + 1) / 2) * 64);  // one 64-byte record per pair of numeric neighbours: states + both reservation words
#else
        hu::device_malloc(&w.state, (V + 2) * 8);  // (states are read in aligned pairs: the last pair may reach one word beyond V)
        hu::device_malloc(&w.resv[0], V * 8);
        hu::device_malloc(&w.resv[1], V * 8);
#endif
        w.cap_v = V;
        w.tag_base = 0xFFFFFFFFu;  // forces the clear below
    }
    if (S > w.cap_s) {
        if (w.touch) {
            hu::device_free(w.touch); hu::device_free(w.src_mirror); hu::device_free(w.claims);
            hu::device_free(w.pending[0]); hu::device_free(w.pending[1]); hu::device_free(w.final_off);
        }
        hu::device_malloc(&w.touch, S * sizeof(Touch));
        hu::device_malloc(&w.src_mirror, S * 4);
        hu::device_malloc(&w.claims, S * 8);
        hu::device_malloc(&w.pending[0], S * 4);
        hu::device_malloc(&w.pending[1], S * 4);
        hu::device_malloc(&w.final_off, S * 8);
        w.cap_s = S;
    }
    const uint64_t spill_need = std::max<uint64_t>(d->total_demand, 1);  // a source emits at most its demand (classification)
    if (spill_need > w.cap_spill) {
        if (w.spill) hu::device_free(w.spill);
        hu::device_malloc(&w.spill, spill_need * 4);
        w.cap_spill = spill_need;
    }
    if (!w.ctl) {
        hu::device_malloc(&w.ctl, RC_COUNT * 8);
        HIP_CHECK(hipHostMalloc(&w.h_ctl, RC_COUNT * 8));
    }
    // reservation tags decrease with every round of every call, so the two reservation arrays are never cleared; only when
    // the 32-bit tag space is used up (or the arrays are new)
#if MTG_REPLAY_RECORDS
    w.tag_base = 0;  // (the records are written whole below, reservation words included)
#endif
    if ((uint64_t)w.tag_base + REPLAY_MAX_ROUNDS + 128 >= 0xFFFFFFF0ull) {
#if MTG_REPLAY_RECORDS
#else
        HIP_CHECK(hipMemsetAsync(w.resv[0], 0xFF, V * 8, st));
        HIP_CHECK(hipMemsetAsync(w.resv[1], 0xFF, V * 8, st));
#endif
        w.tag_base = 0;
    }
    HIP_CHECK(hipEventRecord(d->ev_r[0], st));
    // working copy of the classification state; per-source outputs start at "nothing claimed"
    {
#if MTG_REPLAY_RECORDS
        const uint64_t n_quarters = ((V + 1) / 2) * 4;
        hipLaunchKernelGGL(replay_record_init_kernel, dim3((unsigned)((n_quarters + 255) / 256)), dim3(256), 0, st, d->d_mirror, d->d_mult, d->d_cls, V, w.state);
#else
        ReplayArgs ia{};
        ia.state = w.state; ia.resv[0] = w.resv[0]; ia.resv[1] = w.resv[1];
        hipLaunchKernelGGL(replay_state_init_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, d->d_mirror, d->d_mult, d->d_cls, V, ia);
#endif
        HIP_CHECK(hipGetLastError());
    }
    // the sources that have candidates, in order, with the static words of their admission (Dense)
    uint64_t n_dense = 0;
    {
        static_assert(DENSE_BLOCK == SCAN_BLOCK, "the dense fill uses the scan's block offsets");
        const uint64_t nb = (S + SCAN_BLOCK - 1) / SCAN_BLOCK;
        if (nb > w.cap_blocks) {
            if (w.block_sums) hu::device_free(w.block_sums);
            hu::device_malloc(&w.block_sums, nb * 8);
            w.cap_blocks = nb;
        }
        unsigned long long *total = &d->d_counters[C_OVF_LIST];
        hipLaunchKernelGGL(scan_reduce_kernel<ScanInHasCand>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, ScanInHasCand{d_cand_count}, S, w.block_sums);
        hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, w.block_sums, nb, total);
        HIP_CHECK(hipGetLastError());
        read_counters(d, st);
        n_dense = d->h_counters[C_OVF_LIST];
        if (n_dense > w.cap_dense) {
            if (w.dense) hu::device_free(w.dense);
            hu::device_malloc(&w.dense, n_dense * sizeof(Dense));
            w.cap_dense = n_dense;
        }
        hipLaunchKernelGGL(replay_dense_fill_kernel, dim3((unsigned)nb), dim3(DENSE_BLOCK), 0, st, d->d_out_nodes, d_cand_count,
                           (const unsigned long long *)d_cand_start, (const unsigned long long *)d_pool, d->d_mirror, S, w.block_sums, w.dense, w.src_mirror, w.claims);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipMemsetAsync(w.ctl, 0, RC_COUNT * 8, st));

    ReplayArgs a{};
    a.out_nodes = d->d_out_nodes; a.state = w.state;
    a.cand_start = (const unsigned long long *)d_cand_start; a.cand_count = d_cand_count; a.pool = (const unsigned long long *)d_pool;
    a.resv[0] = w.resv[0]; a.resv[1] = w.resv[1]; a.touch = w.touch; a.src_mirror = w.src_mirror; a.claims = w.claims;
    a.dense = w.dense; a.n_dense = n_dense; a.spill = w.spill; a.pending[0] = w.pending[0]; a.pending[1] = w.pending[1]; a.ctl = w.ctl; a.n_sources = S;
    a.tag_base = w.tag_base; a.max_rounds = REPLAY_MAX_ROUNDS;

    // index-ordered admission windows over the dense list: ~32 K listed sources each, between 4 and 48 of them (a round costs a
    // grid barrier plus one dependent-access chain, so tiny windows are latency bound; huge ones bring the waiting visits back)
    {
        // Round 5: 36 windows instead of 48, the second half of them twice as large -- the late windows meet mostly dead candidates (their
        // retries fall from a third of a window to nothing), so they can be larger and the rounds fewer: 50 -> 39 rounds, rounds kernel
        // 4.41 -> 4.14 ms at 2^27, 1.58 -> 1.46 ms at 2^24 (tools/replay_window_sweep.py; fewer windows of ONE size lose more to
        // retries than they save in rounds: 36 equal windows 4.65 ms)
        uint64_t n_win = std::max<uint64_t>(4, std::min<uint64_t>(36, (n_dense + (1u << 15) - 1) >> 15));
        // (tuning only -- the pair list does not depend on any of it: bits 0-15 of tune_windows = number of windows, bits 16-23 = the
        // sixteenth of them from which they grow, bits 24-31 = by which factor)
        uint64_t grow_16th = n_win >= 8 ? 8 : 16, grow_mul = n_win >= 8 ? 2 : 1;
        if (d->tune_windows & 0xFFFF) { n_win = d->tune_windows & 0xFFFF; grow_16th = 16; grow_mul = 1; }
        if ((d->tune_windows >> 16) & 0xFFFF) { grow_16th = (d->tune_windows >> 16) & 0xFF; grow_mul = std::max<uint64_t>(1, (d->tune_windows >> 24) & 0xFF); }
        const uint64_t g = std::min<uint64_t>(n_win, n_win * grow_16th / 16);  // windows of the base size
        // g windows of `window` sources, the other n_win - g of grow_mul times as many
        a.window = std::max<uint64_t>((n_dense + g + (n_win - g) * grow_mul - 1) / std::max<uint64_t>(g + (n_win - g) * grow_mul, 1), 256);
        a.grow_from = g;
        a.grow_mul = grow_mul;
        {
            const uint64_t base_cover = g * a.window;
            a.n_windows = n_dense <= base_cover ? (n_dense + a.window - 1) / a.window
                                                : g + (n_dense - base_cover + a.window * grow_mul - 1) / (a.window * grow_mul);
        }
        // Bit 32 of the tuning word lets the windows ADAPT on top of this list (replay_kernels.inc: while a round is short and the share
        // of checks sent on does not grow, every third round doubles the window, up to a sixth of the list). Measured, not the default:
        // how often sources block each other is a property of the graph -- on a real compacted de Bruijn graph (100 Mbp) a fifth of
        // the checks are retried whatever the window and eight windows beat thirty-six by 2.4 x, on the G-csr graphs retries explode
        // with the window (8 windows at 2^24: 4.5 x the visits) and do so several rounds AFTER the doubling that caused them, so a
        // controller that serves the first (1.68 -> 1.44 ms) costs the second (2^22: 1.08 -> 1.64 ms). tools/replay_window_sweep.py.
        a.window_cap = (d->tune_windows >> 32) & 1u ? std::max<uint64_t>(a.window, n_dense / 6) : 0;
    }
    const uint64_t widest = a.window_cap ? a.window_cap : a.window * a.grow_mul;
    // one cooperative launch for all rounds: the runtime refuses a grid that cannot be co-resident, so the grid barrier cannot
    // deadlock. Workgroups of 1024 when a round's tiles (a window to admit, about as many sources to check) fill the device, of 256
    // when they do not (replay_kernels.inc); as many workgroups as a round has tiles, at least 64, at most what is co-resident.
    if (w.grid == 0) {
        int coop = 0;
        HIP_CHECK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, d->dev));
        if (!coop) MTG_DIE("device %d does not support cooperative launches (needed by the claim replay's grid barrier)", d->dev);
        int occ = 0, occ_small = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, replay_rounds_kernel<REPLAY_BLOCK>, REPLAY_BLOCK, 0));
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_small, replay_rounds_kernel<REPLAY_BLOCK_SMALL>, REPLAY_BLOCK_SMALL, 0));
        if (occ < 1 || occ_small < 1) MTG_DIE("replay_rounds_kernel does not fit a compute unit");
        w.grid = (unsigned)d->n_cu * (unsigned)std::min(occ, 2);
        w.grid_small = (unsigned)d->n_cu * (unsigned)std::min(occ_small, 2);
    }
    // (by the WIDEST window since round 5: a real de Bruijn graph of 100 Mbp -- 4.6 M listed sources, windows of 86 K and 172 K -- ran
    // its rounds kernel in 2.60 ms with workgroups of 256 and runs it in 1.68 ms with workgroups of 1024: a third as many arrivals at
    // the grid barrier and tile counters; the G-csr graph of 2^24, windows of 19 K and 39 K, stays with 256)
    bool small = 2 * ((widest + REPLAY_BLOCK - 1) / REPLAY_BLOCK) < (uint64_t)d->n_cu;
    if (d->tune_block) small = d->tune_block < REPLAY_BLOCK;  // (tuning only)
    const uint64_t block = small ? REPLAY_BLOCK_SMALL : REPLAY_BLOCK;
    unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(small ? w.grid_small : w.grid), (n_dense + block - 1) / block,
                                                                       std::max<uint64_t>(64, 2 * (((a.window_cap ? widest : a.window) + block - 1) / block))}));
    if (d->tune_grid) grid = std::max(1u, std::min(grid, (unsigned)d->tune_grid));  // (tuning only)
    // one workgroup in role_mod admits, the others check (the longer chain). Measured with the per-XCD barrier: 2^27 (the grid fills
    // the device) role_mod 2 / 4 / 6 / 8 = 6.6 / 6.7 / 7.2 / 8.2 ms; 2^24 (64 workgroups) 2.7 / 2.2 / 2.4 / 2.8 ms
    a.role_mod = (!small && grid >= w.grid) ? 2u : 4u;
    if (d->tune_role_mod) a.role_mod = (uint32_t)std::max(1, d->tune_role_mod);  // (tuning only)
    a.plain_barrier = d->tune_plain_barrier ? 1u : 0u;
    void *kargs[] = {&a};
    rt.lap("buffers + launches");
    HIP_CHECK(hipEventRecord(d->ev_r[1], st));
    HIP_CHECK(hipLaunchCooperativeKernel(small ? reinterpret_cast<void *>(replay_rounds_kernel<REPLAY_BLOCK_SMALL>) : reinterpret_cast<void *>(replay_rounds_kernel<REPLAY_BLOCK>),
                                         dim3(grid), dim3((unsigned)block), kargs, 0, st));
    HIP_CHECK(hipEventRecord(d->ev_r[2], st));
    HIP_CHECK(hipMemcpyAsync(w.h_ctl, w.ctl, RC_COUNT * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    rt.lap("state init + rounds kernel");
#ifdef MTG_REPLAY_PROF
    {
        std::vector<unsigned long long> hp((size_t)grid * 64 * 4);
        HIP_CHECK(hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost));
        const int nr = std::min<int>(64, (int)w.h_ctl[RC_ROUNDS]);
        for (int r = 0; r < nr; r += (r < 8 ? 1 : 8)) {
            for (int role = 0; role < 2; role++) {
                double sw = 0, mw = 0, sf = 0, mf = 0, sb = 0, mb = 0; int n = 0;
                for (unsigned b = role; b < grid; b += 2) {
                    const unsigned long long *t = &hp[((size_t)b * 64 + r) * 4];
                    const double wk = (t[1] - t[0]) * 0.01, fl = (t[2] - t[1]) * 0.01, ba = (t[3] - t[2]) * 0.01;
                    sw += wk; mw = std::max(mw, wk); sf += fl; mf = std::max(mf, fl); sb += ba; mb = std::max(mb, ba); n++;
                }
                if (n) std::fprintf(stderr, "[mtg] replay prof: round %2d %s: work mean %.1f max %.1f us, flush mean %.1f max %.1f, barrier wait mean %.1f max %.1f\n",
                                    r, role ? "check" : "admit", sw / n, mw, sf / n, mf, sb / n, mb);
            }
        }
    }
#endif
    if (w.h_ctl[RC_ABORT]) MTG_DIE("claim replay: a workgroup never reached the grid barrier (watchdog)");
    const int rounds = (int)w.h_ctl[RC_ROUNDS];
    w.tag_base += (uint32_t)rounds + 2;
    d->last_replay_visits = 0;
    for (int r = 0; r < rounds && r < RC_TRACE_ROUNDS; r++) d->last_replay_visits += w.h_ctl[RC_TRACE + 2 * r];
    static const bool replay_debug = std::getenv("MTG_DEBUG") != nullptr;
    if (replay_debug) {
        std::fprintf(stderr, "[mtg] replay: %llu listed sources, windows from %llu (cap %llu), widest admitted %llu; %llu checks sent on\n", (unsigned long long)n_dense,
                     (unsigned long long)a.window, (unsigned long long)a.window_cap, (unsigned long long)w.h_ctl[RC_WIDEST], (unsigned long long)w.h_ctl[RC_RETRIED]);
        std::fprintf(stderr, "[mtg] replay: %d rounds, %llu left; per round (pending, us since kernel start):", rounds, (unsigned long long)w.h_ctl[RC_LEFT]);
        for (int r = 0; r < rounds && r < RC_TRACE_ROUNDS; r++)
            std::fprintf(stderr, " (%llu, %.0f)", (unsigned long long)w.h_ctl[RC_TRACE + 2 * r], (double)(w.h_ctl[RC_TRACE + 2 * r + 1] - w.h_ctl[RC_LEFT_PAR + 2]) * 0.01);
        std::fprintf(stderr, "\n");
    }
    uint64_t n_left = w.h_ctl[RC_LEFT];
    if (n_left > 0) {  // very long priority chain: finish the rest in order on one GPU thread
        uint32_t *lst = w.pending[w.h_ctl[RC_LEFT_PAR] & 1];
        std::vector<uint32_t> rest(n_left);
        HIP_CHECK(hipMemcpyAsync(rest.data(), lst, n_left * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::sort(rest.begin(), rest.end());
        HIP_CHECK(hipMemcpyAsync(lst, rest.data(), n_left * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(replay_tail_kernel, dim3(1), dim3(64), 0, st, a, lst, n_left);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
    }
    d->last_replay_rounds = rounds;
    if (rounds_out) *rounds_out = rounds;

    // compaction in source order
    unsigned long long *cnt = &d->d_counters[C_OVF_LIST];
    uint64_t n_pairs = 0;
    if (n_dense) {  // (over the sources with candidates only: nobody else can claim)
        scan_values(st, w, ScanInClaims{w.claims, w.dense}, n_dense, w.final_off, cnt);
        read_counters(d, st);
        n_pairs = d->h_counters[C_OVF_LIST];
    }
    rt.lap("tail + scan");
    d->last_n_pairs = n_pairs;
    if (n_pairs > w.cap_out || !w.out) {
        if (w.out) hu::device_free(w.out);
        hu::device_malloc(&w.out, std::max<uint64_t>(n_pairs, 1) * sizeof(mtg_pair));
        w.cap_out = std::max<uint64_t>(n_pairs, 1);
    }
    if (n_pairs) {
        hipLaunchKernelGGL(replay_compact_kernel, dim3((unsigned)((n_dense + 255) / 256)), dim3(256), 0, st, a, w.final_off, w.out);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipEventRecord(d->ev_r[3], st));
    HIP_CHECK(hipStreamSynchronize(st));
    {
        float a_ms = 0, b_ms = 0;
        HIP_CHECK(hipEventElapsedTime(&a_ms, d->ev_r[1], d->ev_r[2]));
        HIP_CHECK(hipEventElapsedTime(&b_ms, d->ev_r[0], d->ev_r[3]));
        d->last_replay_kernel_ms = a_ms;
        d->last_replay_gpu_ms = b_ms;
    }
    const auto t_replay_done = std::chrono::steady_clock::now();
    d->last_wall_s[1] = std::chrono::duration<double>(t_replay_done - t_replay_begin).count();
    d->last_wall_s[2] = 0;
    if (!pairs_out) {  // the pairs stay in HBM for a finish on this GPU (device_resident_pairs / device_take_pairs)
        rt.lap("compact (pairs stay on the GPU)");
        return n_pairs;
    }
    mtg_pair *host = (mtg_pair *)big_malloc(std::max<uint64_t>(n_pairs, 1) * sizeof(mtg_pair));  // (the caller frees it with free())
    if (!host) MTG_DIE("out of memory");
    if (n_pairs) {
        if (n_pairs < (1u << 18)) {
            HIP_CHECK(hipMemcpyAsync(host, w.out, n_pairs * sizeof(mtg_pair), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
        } else {
            // through a pinned staging buffer kept with the device, in slices: while slice i+1 crosses PCIe, a few host threads
            // copy slice i into the caller's (pageable, freshly allocated) array
            if (n_pairs > w.cap_h_out) {
                if (w.h_out) HIP_CHECK(hipHostFree(w.h_out));
                w.cap_h_out = n_pairs + n_pairs / 4;
                // (coherent, the default: host threads read each slice right after hipEventSynchronize on an event recorded behind
                // its copy; non-coherent host memory is only guaranteed visible after a stream / device synchronize)
                HIP_CHECK(hipHostMalloc(&w.h_out, w.cap_h_out * sizeof(mtg_pair), hipHostMallocDefault));
            }
            const uint64_t n_slices = std::min<uint64_t>(8, (n_pairs + (1u << 18) - 1) >> 18);
            const uint64_t slice = (n_pairs + n_slices - 1) / n_slices;
            std::vector<hipEvent_t> ev(n_slices);
            for (uint64_t i = 0; i < n_slices; i++) {
                const uint64_t lo = i * slice, n = std::min(slice, n_pairs - lo);
                HIP_CHECK(hipMemcpyAsync(w.h_out + lo, w.out + lo, n * sizeof(mtg_pair), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventReleaseToSystem));
                HIP_CHECK(hipEventRecord(ev[i], st));
            }
            const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({8, (uint64_t)std::thread::hardware_concurrency(), n_pairs >> 18}));
            double dbg_wait = 0, dbg_copy = 0;
            auto worker = [&](unsigned t) {
                for (uint64_t i = 0; i < n_slices; i++) {
                    const uint64_t lo = i * slice, n = std::min(slice, n_pairs - lo);
                    const uint64_t a0 = lo + n * t / T, a1 = lo + n * (t + 1) / T;
                    const auto t0 = std::chrono::steady_clock::now();
                    HIP_CHECK(hipEventSynchronize(ev[i]));
                    const auto t1 = std::chrono::steady_clock::now();
                    std::memcpy(host + a0, w.h_out + a0, (a1 - a0) * sizeof(mtg_pair));
                    if (t == 0) {
                        dbg_wait += std::chrono::duration<double, std::milli>(t1 - t0).count();
                        dbg_copy += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
                    }
                }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; t++) th.emplace_back(worker, t);
            worker(0);
            for (auto &x : th) x.join();
            for (uint64_t i = 0; i < n_slices; i++) HIP_CHECK(hipEventDestroy(ev[i]));
            if (rt.on) std::fprintf(stderr, "[mtg] replay: pair download: %llu slices, %u host threads; thread 0 waited %.3f ms for copies, copied for %.3f ms\n",
                                    (unsigned long long)n_slices, T, dbg_wait, dbg_copy);
        }
    }
    rt.lap("compact + pair download");
    d->last_wall_s[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_replay_done).count();
    *pairs_out = host;
    return n_pairs;
}

int device_set_plan(Device *d, int plan) {
    if (plan >= 0 && plan <= 15) d->plan = (plan & 3) == 1 ? 1 : plan;  // (+ 8: the enumeration level on ONE workgroup -- tests)
    return d->plan;
}
int device_last_replay_rounds(const Device *d) { return d->last_replay_rounds; }
void device_set_replay_tuning(Device *d, uint64_t windows, int block, int grid, int role_mod, int plain_barrier) {
    d->tune_windows = windows; d->tune_block = block; d->tune_grid = grid; d->tune_role_mod = role_mod; d->tune_plain_barrier = plain_barrier != 0;
}
void device_last_pairs_wall_s(const Device *d, double out[3]) { for (int i = 0; i < 3; i++) out[i] = d->last_wall_s[i]; }
void device_last_replay_ms(const Device *d, double out[2]) { out[0] = d->last_replay_kernel_ms; out[1] = d->last_replay_gpu_ms; }
int device_id_of(const Device *d) { return d->dev; }
// was d built from g (same node and edge counts) for bound k - 1? What the resident pairs of d may be finished on.
bool device_matches(const Device *d, const HostGraph &g, uint64_t k) { return d->V == g.node_count() && d->E0 == g.n_original_edges && d->k == k; }
// the pairs of the last claim replay as they lie in HBM (valid until the next replay on this device)
const mtg_pair *device_resident_pairs(const Device *d, uint64_t *n_out) {
    if (n_out) *n_out = d->last_n_pairs;
    return d->last_n_pairs ? d->replay.out : nullptr;
}
// the same, handed over: the caller owns the device array now (device_free_array) -- lets the device graph go before the finish starts
mtg_pair *device_take_pairs(Device *d, uint64_t *n_out) {
    if (n_out) *n_out = d->last_n_pairs;
    mtg_pair *p = d->replay.out;
    d->replay.out = nullptr;
    d->replay.cap_out = 0;
    d->last_n_pairs = 0;
    return p;
}
void device_free_array(int device_id, void *p) {
    hu::device_free_on(device_id, p);
}
// host copy of the resident pairs (malloc'd)
uint64_t device_download_pairs(Device *d, mtg_pair **pairs_out) {
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t n = d->last_n_pairs;
    mtg_pair *host = (mtg_pair *)big_malloc(std::max<uint64_t>(n, 1) * sizeof(mtg_pair));
    if (!host) MTG_DIE("out of memory");
    if (n) HIP_CHECK(hipMemcpy(host, d->replay.out, n * sizeof(mtg_pair), hipMemcpyDeviceToHost));
    *pairs_out = host;
    return n;
}
uint64_t device_last_replay_visits(const Device *d) { return d->last_replay_visits; }

// one-shot path: SSSP candidates for all sources into engine-owned device buffers (pool grown on demand), then the
// claim replay on the GPU; only the matched pairs travel to the host.
// A device copy that is searched once (the consuming call, clib.rs:291): the family blocks (64 of the ~100 bytes per node a call holds
// at its peak) and the search's lists are dead once the candidates exist, and the claim replay's arrays take their place in the arena

void device_warm_replay_unit(hipFuncAttributes *a) { (void)hipFuncGetAttributes(a, reinterpret_cast<const void *>(replay_record_init_kernel)); }

}  // namespace mtg
