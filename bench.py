#!/usr/bin/env python3
"""bench.py -- greedy-matchtigs hot path on N MI355X GPUs (contract: see the task statement / DESIGN.md 'Measurement').

One "step" = one full pass of the hot path over one synthetic unitig graph whose device copy is already
resident in HBM: node classification (GPU) -> bounded many-to-many SSSP candidate generation for this
rank's block of sources (GPU) -> [N>1: one all-gather of candidate lists over RCCL] -> greedy claim
replay -> dummy insertion + Eulerisation -> Euler bicycle decomposition -> cut into tigs (host, rank 0)
-> graph reset (drop the dummy edges again). Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 3 --warmup 1
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(stats: dict) -> int:
    """5 B per relaxed edge + 12 B per settled node + 12 B per emitted candidate."""
    return 5 * stats["relaxed_edges"] + 12 * stats["settled_nodes"] + 12 * stats["emitted"]


def workload_key(args, world: int) -> str:
    if args.workload == "g_seq":
        return f"g_seq:length={args.genome_length}:k={args.k}:seed={args.seed}:plan={args.plan}:gpus={world}"
    return f"g_csr:log2_edges={args.log2_edges}:k={args.k}:seed={args.seed}:plan={args.plan}:gpus={world}"


def traffic_bytes(args, world: int, level_names: list):
    """(HBM bytes per SSSP stage, note): --traffic-bytes, else the committed PMC measurement (separate rocprofv3 --pmc passes of
    this very command, profiles/traffic.json) for exactly this workload -- but only while the kernels it was measured on are
    the kernels that ran (the entry's `levels` = mtg_last_sssp_level_name of the profiled run); a stale entry gives null."""
    if args.traffic_bytes is not None:
        return args.traffic_bytes, "--traffic-bytes"
    try:
        entry = json.loads((ROOT / "profiles" / "traffic.json").read_text()).get(workload_key(args, world))
    except (OSError, ValueError):
        return None, "profiles/traffic.json unreadable"
    if not entry:
        return None, "no PMC pass committed for this workload"
    if entry.get("levels") != level_names:
        return None, f"stale: PMC pass was taken on {entry.get('levels')}, this run used {level_names}"
    return entry.get("traffic_bytes"), entry.get("source")


def size_label(log2_edges: int) -> str:
    """BASELINE.json config the G-csr size stands in for (real genomes are not shippable)."""
    if log2_edges <= 21:
        return "E. coli-like"
    if log2_edges <= 25:
        return "C. elegans-like"
    if log2_edges <= 29:
        return "human-like (inside SURVEY 8's estimate of a human genome's unitig count, U = 10^7..10^8; the nominal size of SURVEY 8d is 2^30)"
    if log2_edges == 30:
        return "human nominal (SURVEY 8d: 2^30)"
    return "661k-pangenome-like (SURVEY 8d: 2^31)"


def stage_models(V: int, E0: int, P: int, N: int, E: int, S: int, n_dense: int, visits: int, kept: int, tigs: int, A: int = 0) -> dict:
    """Algorithmic bytes per stage of ONE step = the sum over the stage's kernels of the arrays each must read and write once
    (a gathered array counts once per gathering pass; DESIGN.md 3 derives every term). V nodes, E0 original darts, P matched
    pairs, N Euleriser units, E darts after the finish, D = E - E0 dummy darts, n = E / 2 biedges, M = E / 64 splitters."""
    D, n = E - E0, E // 2
    M = E // 64 + 1
    import math
    wy = max(1, math.ceil(math.log2(2 * M)))
    # the dummy darts that are bucketed with atomics (pairs, self-mirror phase, tail: D - N) + the arithmetic buckets of the Euleriser's
    # regular steps (N darts: per node the counters, their prefixes and the mirror's, 24 V read + 4 V written + a scan of 12 V, read
    # again by the merge) + one streaming merge with the kept buckets of the original darts
    Dg = max(D - N, 0)
    buckets = 88 * V + 44 * Dg + 4 * N + 4 * E0 + 4 * E  # (32 V + 44 D + 4 E0 + 4 E while all dummy darts were bucketed with atomics)
    return {
        # classify (odeg, mirror, reach -> mult, cls) + compaction (cls, reach -> out_nodes and the A sources the SSSP stage searches)
        "classify": 19 * V + 4 * S + 8 * A,
        # state copy 17 V; dense list + claims words + pair-count scan 48 S; admission 187 B per listed source; 67.2 B per check
        # visit; compaction 32 B per pair
        "replay": 17 * V + 48 * S + 187 * n_dense + int(67.2 * visits) + 32 * P,
        "insert_eulerise": 48 * V + 8 * E0 + 68 * P + 38 * N + 12 * D,
        # (since the arithmetic splitters and the recorded sequence: root pass 4 E + bitmap E / 8, measuring walk 4 E gathered + 4 E
        # recorded, copy 2 E + 2 E -- 16.1 E where flags, scan, compaction and the two walks were 32.4 E)
        "decomposition": buckets + 48 * V + int(88.2 * E) + (100 + 16 * wy) * M,
        "records": buckets + 732 * V + 12 * E,
        "cut": 12 * n + 4 * P + 4 * kept + 4 * tigs,                 # three passes over the closed walks (rotation, count, emit) + the tigs
        # device order since round 6 (cut_first_device.hip): the tigs straight from the pairing, no closed walks. Bd = breaking darts =
        # walkers. Buckets as above; pairing reads row, mirror, adj and writes succ (12 V + 8 E); pass 1 reads every successor word once
        # (4 E: the breaking darts' coalesced, the others gathered), marks a sixth of them, writes length + end per walker (8 Bd); selection
        # + two scans 40 Bd; pass 2 reads (length, offset, tig index) per walker and gathers the kept darts again, writes tigs and ends
        "tigs_from_pairing": buckets + 12 * V + 8 * E + 4 * E + 4 * E // 6 + 8 * (D - 2 * P) + 40 * (D - 2 * P) + 12 * (D - 2 * P) + 8 * kept + 4 * tigs,
    }


def stage_minimum_models(V: int, E0: int, P: int, N: int, E: int, S: int, C: int, kept: int, tigs: int) -> dict:
    """Bytes a stage cannot avoid whatever its pass structure: its inputs read once and its outputs written once, per unit of the
    PROBLEM (not of the implementation: stage_models() above moves when kernels are re-cut, this does not, so `frac_of_minimum` is
    comparable across rounds). V nodes, E0 original darts, P pairs, N Euleriser units, E darts after the finish, S sources, C
    candidates in the lists the search emitted, kept = tig edges, tigs = tig count.
      replay           candidates (key 8 B + source 4 B) + per source its multiplicity and mirror (8 B) -> pairs (12 B)
      insert_eulerise  pairs (12 B) + out-degree and mirror per node (8 B) -> two darts per pair and per unit (from, to, weight: 12 B each)
      decomposition    darts (from, to: 8 B) -> the closed walks (4 B per biedge) and their limits
      records          darts (from, to: 8 B) -> one adjacency entry per dart (4 B) and one offset per node (4 B): what a walk needs at least
      cut              closed walks (4 B per biedge) + the weight of every edge on them (2 B) -> tig edges (4 B) + limits (8 B per tig)
      tigs_from_pairing  (device order, round 6: decomposition and cut as one stage) darts (from, to: 8 B) -> tig edges (4 B) + limits (8 B per tig)"""
    n = E // 2
    return {
        "replay": 12 * C + 8 * S + 12 * P,
        "insert_eulerise": 12 * P + 8 * V + 24 * P + 24 * N,
        "decomposition": 8 * E + 4 * n,
        "records": 8 * E + 4 * E + 4 * V,
        "cut": 6 * n + 4 * kept + 8 * tigs,
        "tigs_from_pairing": 8 * E + 4 * kept + 8 * tigs,   # darts (from, to) -> tig edges + their ends: decomposition and cut in one
    }


def stage_traffic(args, world: int, mode: str):
    """PMC traffic per stage (profiles/stage_traffic.json, tools/stage_traffic.py over separate rocprofv3 --pmc passes of this very
    command) for this workload and Euler mode, or {}."""
    try:
        allj = json.loads((ROOT / "profiles" / "stage_traffic.json").read_text())
    except (OSError, ValueError):
        return {}
    return (allj.get(f"{workload_key(args, world)}:{mode}") or {}).get("stages", {})


def host_cores() -> int:
    """CPU cores this process may really use: the cgroup quota if there is one (16 on a 1-GPU box of the pool), else the count."""
    n = os.cpu_count() or 1
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)  # (the second reference-order finish of a graph page-locks its record arena: ~0.5 s, once)
    ap.add_argument("--workload", choices=["g_csr", "g_seq"], default="g_csr",
                    help="g_csr = random bidirected unitig graph of 2^x edges (SURVEY 8d; the default, human-like at 2^27); "
                         "g_seq = REAL compacted de Bruijn graph of a random genome with haplotype bubbles (--genome-length)")
    ap.add_argument("--log2-edges", type=int, default=27, help="G-csr: |E| ~ 2^x directed unitig edges (whole graph)")
    ap.add_argument("--genome-length", type=int, default=4_600_000, help="G-seq: genome length (4 haplotypes, 2 %% substitutions)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--extra-seeds", default="2,3", help="G-csr, N = 1: further seeds whose SSSP stage is timed after the main run (SURVEY 8d: seeds {1,2,3})")
    ap.add_argument("--plan", type=int, default=0, help="SSSP level plan: 0 = default (enumeration level + cascade), 1 = cascade only")
    ap.add_argument("--device-mode-steps", type=int, default=None,
                    help="timed steps of the second mode (device Euler decomposition) after the headline region, reported as "
                         "the device_mode block (default: as many as --steps)")
    ap.add_argument("--no-cold-steps", dest="cold_steps", action="store_false", help="skip the two cold (first-call) steps")
    ap.add_argument("--full-size-log2", type=int, default=30,
                    help="after the timed regions: ONE device-mode step at this G-csr size (SURVEY 8d's nominal human size) when the box "
                         "has the memory (>= 170 GB of HBM and >= 140 GB of host memory free); 0 = skip")
    ap.add_argument("--no-one-shot", dest="one_shot", action="store_false",
                    help="skip the one_shot block (the consuming one-shot call, host arrays in -> clib.rs arrays out, in a child process of its own per Euler mode)")
    ap.add_argument("--keep-awake", action="store_true", help="measurement: a trivial kernel every 2 ms while the host walks (mtg_set_finish_tuning flag 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--traffic-bytes", type=float, default=None, help="HBM bytes per launch from a separate --pmc pass")
    ap.add_argument("--host-replay", action="store_true", help="run the claim loop on the host instead of the GPU (A/B)")
    ap.add_argument("--host-finish", action="store_true", help="run insertion / Euleriser / cut on the host instead of the GPU (A/B)")
    ap.add_argument("--euler", choices=["host", "device"], default="host",
                    help="host = Euler walk in the reference's order (default, bit-exact tigs); device = parallel Euler "
                         "bicycles on the GPU (SURVEY 8 f-3: valid walks, same #tigs/cumulative length, different order)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for functional checks)")
    ap.add_argument("--single-device", action="store_true",
                    help="functional check of the N>1 code path on a 1-GPU box: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the matchtigs_amd hot path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    from matchtigs_amd import api, synth, torch_glue
    from matchtigs_amd import distributed as mdist

    if args.keep_awake:
        api.set_finish_tuning(keep_awake=True)
    api.set_default_device(local_rank)  # (the GPU whose memory a graph's construction reserves ahead of the call that follows)
    k = args.k
    # ---- the consuming one-shot call (clib.rs:280-291: every real call of the reference is a first call), each Euler mode in a
    # process of its own, before this process takes any device memory ----
    one_shot = run_one_shot(args) if (world == 1 and args.one_shot and args.workload == "g_csr") else None
    clib_route = run_clib_route(args) if (world == 1 and args.one_shot and args.workload == "g_csr") else None
    # fixed total graph (strong scaling): the same unitig graph on every rank, sources block-partitioned.
    # G-csr is generated ON each rank's GPU (csrc/synth_device.hip, the twin of synth.g_csr): seconds, no numpy argsort.
    t_gen = time.perf_counter()
    if args.workload == "g_seq":
        # (the torch form of the generator, on this rank's GPU: the same graph as synth.g_seq_arrays -- tests hold them equal -- in
        # seconds instead of minutes at 10^8)
        ua = synth.g_seq_arrays_torch(args.genome_length, seed=args.seed, k=k, haplotypes=4, sub_rate=0.02, device=f"cuda:{local_rank}")
        torch.cuda.empty_cache()
        graph = api.Bigraph.from_unitig_links_arrays(ua.weights, ua.links)
        workload = (f"G-seq REAL compacted de Bruijn graph: random genome L={args.genome_length}, 4 haplotypes, 2% substitutions, "
                    f"k={k}, {ua.n_unitigs} unitigs, {len(ua.kmers)} k-mers (clib.rs graph construction), seed={args.seed}")
        del ua
    else:
        graph = synth.g_csr_device(int((1 << args.log2_edges) / 1.5 / 2), seed=args.seed, k=k, device_id=local_rank)
        workload = (f"G-csr random bidirected de Bruijn-like unitig graph ({size_label(args.log2_edges)}, 2^{args.log2_edges} nominal "
                    f"edges), k={k}, seed={args.seed}, generated on the GPU")
    n_nodes, n_edges = graph.node_count(), graph.edge_count()
    t_graph = time.perf_counter() - t_gen
    dev = api.DeviceGraph(graph, k, local_rank, reserve_work=True)  # H2D: inputs resident in HBM before any timed region
    dev.set_plan(args.plan)
    if rank != 0:  # only rank 0 finishes: the other ranks keep the device copy and give the host graph back
        dev.graph = None
        graph = None
    euler_mode = api.EulerMode.Device if args.euler == "device" else api.EulerMode.HostReferenceOrder
    finish_stage = api.FinishStage.Host if args.host_finish else api.FinishStage.Auto
    t_gen = time.perf_counter() - t_gen
    stream = torch_glue.current_stream_ptr()

    bufs = None
    result_info: dict = {}
    count_ref = [None]  # the candidate counts of the last step (number of sources with candidates, for the replay's byte model)

    def sync_barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # static source partition: contiguous blocks of equal estimated work (1 + out-degree of the source node),
    # computed on the GPU (mtg_partition_sources); classification is a function of the graph, so the split is computed once,
    # before the timed region, and every rank gets the same cuts
    ranges_by_work = None
    if world > 1:
        dev.classify(stream)
        cuts = api.partition_sources(dev, world)
        ranges_by_work = [(cuts[r], cuts[r + 1]) for r in range(world)]

    stage_ms: dict[str, dict[str, list]] = {"host": {}, "device": {}}  # per Euler mode: stage -> GPU ms (HIP events) of every timed step

    def step(mode, acc, kernel_ms=None, level_ms=None, stages=None, tig_download_ms=None):
        nonlocal bufs
        ph = {}
        t0 = time.perf_counter()
        S = dev.classify(stream)
        ranges = ranges_by_work if ranges_by_work is not None else mdist.partition_sources(S, world)
        lo, hi = ranges[rank]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ph["classify"] = t1 - t0
        if bufs is None or bufs.n != hi - lo:
            bufs = torch_glue.CandidateBuffers(hi - lo, max(1024, 4 * (hi - lo)))
        bufs = torch_glue.run_sssp(dev, lo, hi, bufs)
        t2 = time.perf_counter()
        ph["sssp"] = t2 - t1
        if kernel_ms is not None:
            kernel_ms.append(dev.last_sssp_kernel_ms())
            level_ms.append(dev.last_sssp_levels())
        if world > 1:
            start_all, count_all, pool_all = mdist.allgather_candidates(bufs.start, bufs.count, bufs.pool, bufs.used, ranges)
            torch.cuda.synchronize()
        else:
            start_all, count_all, pool_all = bufs.start[: bufs.n], bufs.count[: bufs.n], bufs.pool[: bufs.used]
        t3 = time.perf_counter()
        ph["allgather"] = t3 - t2
        if rank == 0:
            if args.host_replay:
                cs, cc, po = mdist.to_numpy_u(start_all, count_all, pool_all)
                count_ref[0] = count_all
                on, mu, li = dev.classify_download(stream)
                t4 = time.perf_counter()
                ph["download"] = t4 - t3
                pairs = graph.replay_claims(on, mu, li, cs, cc, po)
            resident = not args.host_replay and not args.host_finish
            if not args.host_replay:  # claim loop on the GPU (deterministic reservations)
                t4 = time.perf_counter()
                ph["download"] = 0.0
                if resident:  # ... whose pairs stay in HBM for the finish on the same GPU (as in mtg_compute_tigs_cfg)
                    n_pairs = dev.replay_claims_resident(start_all.data_ptr(), count_all.data_ptr(), pool_all.data_ptr(), stream)
                else:
                    pairs = dev.replay_claims_device(start_all.data_ptr(), count_all.data_ptr(), pool_all.data_ptr(), stream)
                    n_pairs = len(pairs)
                po = pool_all
                count_ref[0] = count_all
                result_info["replay_rounds"] = dev.last_replay_rounds()
                result_info["replay_visits"] = dev.last_replay_visits()
            else:
                n_pairs = len(pairs)
            t5 = time.perf_counter()
            ph["replay"] = t5 - t4
            if resident:
                # (the tigs of a finish on the GPU stay in HBM, like the pairs before them: a caller asks for counts, flattens into
                # clib.rs arrays or spells on the GPU; what bringing them to the host as walks costs is `tig_download_ms`, measured
                # once after the timed region)
                tigs = api.finish_greedytigs_resident(graph, dev, k, mode, local_rank, finish_stage)
                n_tigs, n_tig_edges = tigs.count(), tigs.total_edges()
                if tig_download_ms is not None and not tig_download_ms:
                    t_dl = time.perf_counter()
                    tigs.arrays()
                    tig_download_ms.append((time.perf_counter() - t_dl) * 1e3)
                del tigs
            else:
                tigs_lim, tigs_edges = api.finish_greedytigs_np(graph, pairs, k, mode, local_rank, finish_stage)
                n_tigs, n_tig_edges = len(tigs_lim), len(tigs_edges)
                del tigs_lim, tigs_edges
            t6 = time.perf_counter()
            ph["finish"] = t6 - t5
            hp = api.last_phase_seconds()
            # insertion + Euleriser (on the GPU unless --host-finish), Euler bicycles (reference-order host walk over GPU-built
            # records, or the device decomposition), rotate + cut (GPU) + tig download
            ph["insert_eulerise"], ph["euler"], ph["cut"] = hp["eulerise"], hp["euler"], hp["cut"]
            if mode == api.EulerMode.Device:
                ph["euler_device_kernels"] = api.last_euler_kernel_ms() * 1e-3
            if stages is not None and not args.host_finish and not args.host_replay:
                fs = api.last_finish_device_stage_ms()
                rp = dev.last_replay_ms()
                for kk, v in (("replay_rounds_kernel", rp["rounds_kernel_ms"]), ("replay", rp["gpu_ms"]), ("insert_eulerise", fs["insert_eulerise_ms"]),
                              ("records", fs["records_ms"]), ("decomposition", fs["decomposition_ms"]), ("cut", fs["cut_ms"])):
                    stages.setdefault(kk, []).append(v)
                result_info.update(darts=fs["darts"], units=fs["units"])
            result_info.update(S=int(S), pairs=int(n_pairs), tigs=int(n_tigs), tig_edges=int(n_tig_edges),
                               pool_words=int(len(po)), graph_edges_after=int(graph.edge_count()))
            graph.reset()
            ph["reset"] = time.perf_counter() - t6
        for kk, v in ph.items():
            acc[kk] = acc.get(kk, 0.0) + v

    def timed_region(mode, n_warm, n_steps, kernel_ms=None, level_ms=None):
        acc: dict[str, float] = {}
        for _ in range(n_warm):
            step(mode, {})
        sync_barrier()
        stages = stage_ms["device" if mode == api.EulerMode.Device else "host"]
        t_begin = time.perf_counter()
        for _ in range(n_steps):
            step(mode, acc, kernel_ms, level_ms, stages)
        sync_barrier()
        elapsed = time.perf_counter() - t_begin
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed / max(n_steps, 1) * 1e3, acc

    # ---- cold steps: the FIRST step on a fresh graph, as a one-shot caller sees it (the reference is only ever used that way):
    # nothing kept from an earlier call -- the library's device block cache is released, the graph has no device buckets, no
    # page-locked arena, the candidate buffers are new. Each mode gets a graph of its own; the device-graph build is reported beside.
    cold = None
    if world == 1 and args.cold_steps and args.workload == "g_csr":
        cold = {}
        for name, mode in (("device", api.EulerMode.Device), ("host", euler_mode)):
            if name == "host" and args.euler == "device":
                continue
            t0c = time.perf_counter()
            g2 = synth.g_csr_device(int((1 << args.log2_edges) / 1.5 / 2), seed=args.seed, k=k, device_id=local_rank)
            t1c = time.perf_counter()
            # (a caller's first and only search: the device graph without the goal-directed lower bounds, as mtg_compute_tigs_cfg builds it)
            d2 = api.DeviceGraph(g2, k, local_rank, lower_bounds=False, reserve_work=True)
            d2.set_plan(args.plan)
            torch.cuda.synchronize()
            t2c = time.perf_counter()
            keep = (graph, dev, bufs)
            graph, dev, bufs = g2, d2, None
            phc: dict[str, float] = {}
            step(mode, phc)
            torch.cuda.synchronize()
            t3c = time.perf_counter()
            cold[name] = {"step_ms": round((t3c - t2c) * 1e3, 2), "device_graph_build_ms": round((t2c - t1c) * 1e3, 2),
                          "generate_ms": round((t1c - t0c) * 1e3, 2), "phases_ms": {kk: round(v * 1e3, 2) for kk, v in phc.items()},
                          "sssp_stage_ms": round(d2.last_sssp_kernel_ms(), 4),
                          "sssp_levels": [{"kernel": x["kernel"], "ms": round(x["ms"], 4), "sources": x["sources"]} for x in d2.last_sssp_levels()]}
            graph, dev, bufs = keep
            del g2, d2
            torch.cuda.empty_cache()
        cold["note"] = ("first step on a fresh graph and a fresh device graph (beside the main one): its device arrays are new ranges of the "
                        "library's arena (new chunks from the driver where the arena has no room), host result arrays and the walk's arena are mapped "
                        "for the first time. The arena is NOT released first: a release would put its frees and the fresh allocations that follow -- driver "
                        "calls, which sporadically stall for seconds on this pool (tools/alloc_probe.hip, DESIGN 9) -- into the measured step; the "
                        "one_shot block is the consuming call in a process of its own")

    # ---- second mode, first class: the same step with the parallel Euler decomposition on the GPU, in its own timed region.
    # It runs BEFORE the headline region: the two modes use different sets of device work arrays, and the runtime's stream-ordered
    # pool serves a mode best right after that mode's own warm-up step. ----
    device_mode = None
    dm_steps = args.steps if args.device_mode_steps is None else args.device_mode_steps
    if args.euler == "host" and dm_steps > 0:
        dm_kernel_ms: list[float] = []
        dm_ms, dm_acc = timed_region(api.EulerMode.Device, 1, dm_steps, dm_kernel_ms, [])
        tig_dl: list[float] = []
        step(api.EulerMode.Device, {}, tig_download_ms=tig_dl)  # one more step on every rank (it gathers), untimed: what bringing its tigs to the host as walks costs
        sync_barrier()
        device_mode = {"steps": dm_steps, "ms_per_step": round(dm_ms, 3),
                       "phases_ms": {kk: round(v / dm_steps * 1e3, 3) for kk, v in dm_acc.items()},
                       # (the SSSP stage by the same HIP events as `roofline', inside THESE steps: they follow one another on the GPU, where a
                       # headline step in the reference's order starts after seconds of host walk with an idle GPU)
                       "sssp_stage_ms": round(float(np.mean(dm_kernel_ms)), 4) if dm_kernel_ms else None,
                       "tigs": result_info.get("tigs"),
                       "tig_download_ms": round(tig_dl[0], 3) if tig_dl else None,
                       "ms_per_step_with_tigs_on_host": round(dm_ms + tig_dl[0], 3) if tig_dl else None,
                       "note": "every stage of the step on the GPU; tig order differs from the reference's, tig count and cumulative length are equal (DESIGN 4.7 / docs/history). "
                               "The step ends with the tigs in HBM (edge ids + exclusive ends, as the cutter wrote them): a caller takes counts, flattens "
                               "into clib.rs arrays (one_shot), spells on the GPU, or asks for the walks on the host -- tig_download_ms, outside the step "
                               "since round 5 (rounds 1-4 ended every step with that copy into pageable memory: +8 ms at 2^27)"}

    # ---- the headline: K timed steps in the selected mode (default: the reference's walk order, bit-exact tigs) ----
    kernel_ms: list[float] = []
    level_ms: list[list[dict]] = []
    ms_per_step, phases_acc = timed_region(euler_mode, args.warmup, args.steps, kernel_ms, level_ms)
    # what delivering the step's tigs to the host as walks costs on top (the reference's compute_tigs always ends with them in host
    # memory; rounds 1-4 ended every timed step with that copy): the same arrays in either Euler mode, so the device-mode measurement
    # is reused when there is one, else one more untimed step measures it
    if device_mode is not None:  # (the same branch on every rank: the extra step below gathers and ends in a barrier)
        tig_download_ms = device_mode["tig_download_ms"]  # (measured on rank 0, which finishes; None elsewhere)
    elif not args.host_replay and not args.host_finish:
        tig_dl2: list[float] = []
        step(euler_mode, {}, tig_download_ms=tig_dl2)
        sync_barrier()
        tig_download_ms = round(tig_dl2[0], 3) if tig_dl2 else None
    else:
        tig_download_ms = 0.0  # (a host finish ends with the tigs on the host anyway)

    # ---- units of work (untimed counting kernel over this rank's block) ----
    S = dev.n_sources
    lo, hi = (ranges_by_work if ranges_by_work is not None else mdist.partition_sources(S, world))[rank]
    stats = dev.sssp_count(lo, hi, stream)
    # ... and what the goal-directed search really visits (pruned by the nearest-in-node lower bounds): the bytes the stage needs
    visited = dev.sssp_count_visited(lo, hi, stream) if dev.prunes() else None
    local_kernel_ms = float(np.mean(kernel_ms)) if kernel_ms else 0.0
    vz = visited or {"relaxed_edges": 0, "settled_nodes": 0, "emitted": 0, "sources": 0}
    tot = torch.tensor([stats["relaxed_edges"], stats["settled_nodes"], stats["emitted"], stats["relax_attempts"],
                        stats["overflow_sources"], vz["relaxed_edges"], vz["settled_nodes"], vz["emitted"], vz["sources"]], dtype=torch.int64, device="cuda")
    # per-N scaling figures of the stages that DO shard (the whole step is bound by rank 0's finish): max over ranks
    kmax = torch.tensor([local_kernel_ms, phases_acc.get("sssp", 0.0) / max(args.steps, 1) * 1e3,
                         phases_acc.get("allgather", 0.0) / max(args.steps, 1) * 1e3], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    tot = [int(x) for x in tot.tolist()]
    kmax = [float(x) for x in kmax.tolist()]
    total_stats = dict(relaxed_edges=tot[0], settled_nodes=tot[1], emitted=tot[2], relax_attempts=tot[3], overflow_sources=tot[4])
    total_visited = dict(relaxed_edges=tot[5], settled_nodes=tot[6], emitted=tot[7], searched_sources=tot[8]) if visited else None

    device_graph_bytes = dev.graph_bytes()
    if rank == 0:
        # roofline of the dominant GPU stage of the sharded part (the SSSP stage = its level kernels) on THIS rank's launch.
        # Bytes: SURVEY 8d's per-unit figures x the units the launch really processes -- with the goal-directed pruning that is the
        # visited set (mtg_sssp_count_visited), not the full balls; the full-ball figure is kept beside it as the throughput in the
        # reference's units (a full-ball Dijkstra's edges per second).
        alg_bytes_full = algorithmic_bytes(stats)
        alg_bytes = algorithmic_bytes(visited) if visited else alg_bytes_full
        achieved = alg_bytes / (local_kernel_ms * 1e-3) / 1e9 if local_kernel_ms > 0 else 0.0
        kernels = []
        if level_ms:
            for li in range(max(len(x) for x in level_ms)):
                vals = [x[li] for x in level_ms if li < len(x)]
                kernels.append({"kernel": vals[0].get("kernel", f"level{li}"),
                                "avg_launch_ms": round(float(np.mean([v["ms"] for v in vals])), 4),
                                "sources": int(vals[0]["sources"])})
        traffic, traffic_note = traffic_bytes(args, world, [kk["kernel"] for kk in kernels])
        roofline = {
            "bound": "hbm", "kernel": "SSSP stage = sum of its level kernels (see 'kernels')", "kernels": kernels,
            "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
            "traffic": traffic, "traffic_note": traffic_note,
            "bytes_note": ("algorithmic bytes = 5 B x relaxed edges + 12 B x settled nodes + 12 B x candidates of the search the launch runs: "
                           "the goal-directed (pruned) search when the device graph carries lower bounds (units in 'visited_per_step': settled = "
                           "nodes whose distance the search determines, relaxed = out-edges of the nodes it expands -- an in-node with nothing "
                           "beyond it within the bound is settled from its parent's block and not expanded), "
                           "else full balls; 'full_ball_equivalent' prices the same time against the full-ball units of 'units_per_step'"),
            "gather_ceiling_note": "dependent random 64-B block gathers saturate at ~54 G/s below 3 GB of blocks and at ~44 G/s (one request "
                                   "per lane and line; 19 G/s with four) at 5.7 GB, tools/gather_bench_tlb.hip",
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(local_kernel_ms, 4),
            "full_ball_equivalent": {"algorithmic_bytes_per_launch": alg_bytes_full,
                                     "achieved": round(alg_bytes_full / (local_kernel_ms * 1e-3) / 1e9, 3) if local_kernel_ms > 0 else 0.0,
                                     "frac": round(alg_bytes_full / (local_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if local_kernel_ms > 0 else 0.0},
            "kernel_sssp_edges_per_s": round(stats["relaxed_edges"] / (local_kernel_ms * 1e-3), 1) if local_kernel_ms > 0 else 0.0,
            "work_efficiency_attempts_per_edge": round(stats["relax_attempts"] / max(stats["relaxed_edges"], 1), 4),
        }
        # what the pruning costs: the lower bounds are computed once per device graph (HIP events inside the engine); a caller that
        # searches a graph ONCE (the reference's only calling convention) runs the full-ball search without them instead -- the stage
        # time of the cold step's device graph, built the way mtg_compute_tigs_cfg builds it
        pre_ms = dev.lower_bounds_ms()
        roofline["precompute_ms"] = round(pre_ms, 4)
        cold_dev = (cold or {}).get("device") or (cold or {}).get("host")
        if cold_dev:
            os_ms = cold_dev["sssp_stage_ms"]
            roofline["one_shot"] = {"stage_ms": os_ms, "precompute_ms": 0.0, "kernels": cold_dev["sssp_levels"],
                                    "algorithmic_bytes_per_launch": alg_bytes_full,
                                    "frac": round(alg_bytes_full / (os_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if os_ms > 0 else None,
                                    "note": "first search on a fresh device graph without lower bounds (full balls, cold caches): what one call pays"}
            gain = os_ms - local_kernel_ms
            roofline["pruned_search_pays_off_after_steps"] = (int(np.ceil(pre_ms / gain)) if gain > 0 and pre_ms > 0 else None)
        # ---- the other GPU stages of the step, each against the HBM roofline: GPU time from HIP events inside the engine (on the
        # stream the kernels run on), algorithmic bytes from stage_models(), counter traffic from profiles/stage_traffic.json ----
        # (the number of candidates = the sum of the lists' lengths; the pool holds them in per-wave chunks with unused tails, whose
        # number depends on how the sources are split over ranks)
        if count_ref[0] is not None:
            result_info["candidates"] = int(count_ref[0].to(torch.int64).sum().item())
        roofline_stages = None
        if world == 1 and (stage_ms["host"] or stage_ms["device"]):
            count_last = count_ref[0]
            n_dense = int((count_last > 0).sum().item()) if count_last is not None else 0
            models = stage_models(n_nodes, n_edges, result_info.get("pairs", 0), result_info.get("units", 0), result_info.get("darts", 0),
                                  result_info.get("S", 0), n_dense, result_info.get("replay_visits") or 0, result_info.get("tig_edges", 0),
                                  result_info.get("tigs", 0), (total_visited or {}).get("searched_sources", 0))
            names = {"replay": "claim replay: replay_state_init + dense list + replay_rounds_kernel + pair-count scan + compaction",
                     "insert_eulerise": "matched-pair darts + Euleriser: degree / need / 3 scans / expand / zip_check / zip_emit / head kernels",
                     "decomposition": "Euler decomposition (device mode): bucket merge, pairing, union-find over biedges, hooking, recording splitter walks, ranking, copy",
                     "records": "walk records (reference-order mode): bucket merge + lean_build + wide_build kernels (the records' download is not in it)",
                     "cut": "rotate + cut: cycle heads / rotation / rotate_cycles / cut_flags / 3 scans / cut_write (the tig download is not in it)",
                     "tigs_from_pairing": "tigs straight from the pairing (device order, cut_first_device.hip; takes the place of decomposition + cut): bucket merge, "
                                          "pairing, stretch_measure (one walker per breaking dart), selection + 2 scans, stretch_write; trails without a breaking "
                                          "dart spliced in on the host"}
            minimum = stage_minimum_models(n_nodes, n_edges, result_info.get("pairs", 0), result_info.get("units", 0), result_info.get("darts", 0),
                                           result_info.get("S", 0), int(total_stats.get("emitted", 0)), result_info.get("tig_edges", 0),
                                           result_info.get("tigs", 0))
            roofline_stages = []
            for mode_name in ("device", "host"):
                tr = stage_traffic(args, world, mode_name)
                # (device order: when the cutter had nothing to do -- no closed walks were built, cut_first_device.hip made the tigs
                # -- decomposition + cut are ONE stage with a byte model of its own)
                cut_vals = [v for v in stage_ms[mode_name].get("cut", []) if v > 0]
                cut_first = mode_name == "device" and bool(stage_ms[mode_name].get("decomposition")) and (not cut_vals or float(np.mean(cut_vals)) < 0.05)
                for st_name in ("replay", "insert_eulerise", "decomposition", "records", "cut"):
                    vals = [v for v in stage_ms[mode_name].get(st_name, []) if v > 0]
                    if not vals or (cut_first and st_name == "cut"):
                        continue
                    ms = float(np.mean(vals))
                    if cut_first and st_name == "decomposition":
                        st_name = "tigs_from_pairing"
                    b = models[st_name]
                    entry = {"stage": st_name, "euler_mode": mode_name, "kernels": names[st_name], "bound": "hbm", "avg_launch_ms": round(ms, 4),
                             "algorithmic_bytes": int(b), "achieved": round(b / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "traffic": (tr.get(st_name) or {}).get("traffic_bytes"),
                             # (inputs once + outputs once, independent of the pass structure: stage_minimum_models)
                             "minimum_bytes": int(minimum[st_name]),
                             "frac_of_minimum": round(minimum[st_name] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                    if st_name == "replay":
                        rk = [v for v in stage_ms[mode_name].get("replay_rounds_kernel", []) if v > 0]
                        entry["rounds_kernel_ms"] = round(float(np.mean(rk)), 4) if rk else None
                    roofline_stages.append(entry)
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            ex = graph.export()
            bg = synth.Bigraph(ex["mirror"], ex["edge_from"], ex["edge_to"], ex["edge_weight"], k)
            del ex
            cpu_baseline = run_cpu_baseline(bg, k, args.cpu_baseline_seconds, dev, stream, total_stats["relaxed_edges"])
            del bg
            gpu_stage_s = sum(phases_acc.get(kk, 0.0) for kk in ("classify", "sssp", "allgather", "download", "replay")) / max(args.steps, 1)
            cpu_baseline["gpu_same_stage_edges_per_s"] = round(total_stats["relaxed_edges"] / max(gpu_stage_s, 1e-9), 1)
            if "stage" in cpu_baseline and args.workload == "g_csr" and args.log2_edges > 24:
                # the WHOLE path on the CPU next to the whole step on the GPU: the oracle's literal Hierholzer does not finish the bench
                # graph in the budget, so both run on a bounded sample of the same workload family (G-csr 2^24, same generator and seed)
                cpu_baseline["whole_path_sample"] = whole_path_sample(args, k, local_rank)
        extra_seeds = None
        if world == 1 and args.workload == "g_csr" and args.extra_seeds:
            extra_seeds = sssp_stage_of_seeds(args, [int(x) for x in args.extra_seeds.split(",") if x.strip()], local_rank)
        # ---- one step at SURVEY 8d's nominal human size, after everything else (the 2^27 graph is given back first) ----
        full_size = None
        if world == 1 and args.workload == "g_csr" and args.full_size_log2 and args.full_size_log2 > args.log2_edges >= 26:  # (the headline configuration only)
            del dev, graph, bufs
            dev = graph = bufs = None
            count_ref[0] = None
            torch.cuda.empty_cache()
            api.release_device_memory(local_rank)
            time.sleep(1.0)  # (let the driver settle after ~30 GB of frees: DESIGN 9)
            full_size = full_size_step(args, k, local_rank)
        value = total_stats["relaxed_edges"] / (ms_per_step * 1e-3)
        out = {
            "metric": "greedy-matchtigs SSSP edges/s, full-ball-equivalent edges (the edges a full-ball Dijkstra of every source examines -- the "
                      "pruned search examines fewer for the same lists, 'visited_per_step') / whole hot-path step: classify+SSSP+claim+Euler+cut",
            "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "wall_clock_s": round(ms_per_step / 1e3, 4), "higher_is_better": True,
            # (the timed step ends with the tigs in HBM; the variant that ends with them on the host as walks, comparable with rounds 1-4
            # and with a CPU compute_tigs: ms_per_step + the measured copy)
            "tig_download_ms": tig_download_ms,
            "ms_per_step_with_tigs_on_host": None if tig_download_ms is None else round(ms_per_step + tig_download_ms, 3),
            "scaling": "strong", "vs_baseline": None, "dtype": "u32/u64 integer", "data": "synthetic",
            "config": {"workload": f"{workload}; |V|={n_nodes}, |E|={n_edges}; sources block-partitioned over ranks",
                       "V": n_nodes, "E": n_edges, "k": k, "sources": result_info.get("S"),
                       "pairs": result_info.get("pairs"), "tigs": result_info.get("tigs"),
                       "candidates": result_info.get("candidates"), "parallelism": f"sources/{world}"},
            "units_per_step": total_stats,
            "visited_per_step": total_visited,
            "phases_ms": {kk: round(v / max(args.steps, 1) * 1e3, 3) for kk, v in phases_acc.items()},
            "euler_mode": args.euler,
            "device_mode": device_mode,
            "euler_device_ms_per_step": None if device_mode is None else device_mode["ms_per_step"],
            # the stages that shard over ranks, max over ranks (the whole-step curve is flat by design: rank 0 finishes alone)
            "scaling_stages_ms": {"sssp_kernels": round(kmax[0], 4), "sssp_stage": round(kmax[1], 4), "allgather": round(kmax[2], 4)},
            "level0_finish_rate": (round(1.0 - kernels[1]["sources"] / max(kernels[0]["sources"], 1), 6) if len(kernels) > 1 else 1.0) if kernels else None,
            "setup_s": round(t_gen, 2), "setup_graph_s": round(t_graph, 2),
            "device_graph_bytes": device_graph_bytes,
            "roofline": roofline,
            "roofline_stages": roofline_stages,
            "one_shot": one_shot,
            "clib_route": clib_route,
            "cold_step_ms": cold,
            "full_size": full_size,
            # claim replay (one cooperative kernel): cost model = source visits x (32-B touch record + 8-B state and 8-B reservation
            # word for the source's own binode and for each of its ~1.2 live candidates) + 64-bit atomics of the failed visits
            "replay": {"rounds": result_info.get("replay_rounds"), "source_visits": result_info.get("replay_visits"),
                       "model_bytes": None if result_info.get("replay_visits") is None else int(result_info["replay_visits"] * (32 + 16 + 1.2 * 16)),
                       "ms": round(phases_acc.get("replay", 0.0) / max(args.steps, 1) * 1e3, 3)},
            "cpu_baseline": cpu_baseline,
            "vs_cpu_baseline_same_stage": None if not cpu_baseline else round(cpu_baseline["gpu_same_stage_edges_per_s"] / max(cpu_baseline["value"], 1e-9), 2),
            "seeds": extra_seeds,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_one_shot(args) -> dict:
    """tools/one_shot.py as a child process per Euler mode: host edge arrays -> mtg_graph_from_edges -> mtg_compute_tigs_clib (what
    matchtigs_compute_tigs runs) -> clib.rs output arrays, twice per process. `first_call_of_the_process` pays for the HIP runtime's
    queues, the transfer ring and the kernels' code objects on top; `later_call` is a first call on a fresh graph in a process that has
    called before. The library's device memory is released before each call; the generator that makes the input arrays is not timed."""
    import subprocess

    out = {"note": "host edge arrays -> mtg_graph_from_edges -> mtg_compute_tigs_clib (= matchtigs_compute_tigs after its configuration) -> "
                   "clib.rs output arrays; total_s = graph_build_s + compute_tigs_clib_s (graph_free_s, the consumed handle's memory going back, beside it); "
                   "phases_s of the compute call: device_build = upload + device graph, sssp = the search, replay = the claim replay, download = the matched "
                   "pairs to the host (0: they stay in HBM for the finish), eulerise = "
                   "matched-pair darts + Euleriser, euler = Euler bicycles, cut = rotate + cut + flattened tigs into the caller's arrays"}
    for mode in ("device", "host"):
        if mode == "host" and args.euler == "device":
            continue
        cmd = [sys.executable, str(ROOT / "tools" / "one_shot.py"), "--log2-edges", str(args.log2_edges), "--k", str(args.k),
               "--seed", str(args.seed), "--euler", mode, "--calls", "2"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
            if r.returncode != 0 or len(lines) < 2:
                out[mode] = {"error": f"rc={r.returncode}: {r.stderr[-300:]}"}
                continue
            keep = ("graph_build_s", "compute_tigs_clib_s", "graph_free_s", "total_s", "tigs", "phases_s")
            out[mode] = {"first_call_of_the_process": {kk: lines[0][kk] for kk in keep},
                         "later_call": {kk: lines[1][kk] for kk in keep}, "total_s": lines[1]["total_s"],
                         "same_result_both_calls": lines[0]["checksum"] == lines[1]["checksum"] and lines[0]["tigs"] == lines[1]["tigs"]}
        except (subprocess.TimeoutExpired, OSError, ValueError) as e:
            out[mode] = {"error": repr(e)[:300]}
    return out


def run_clib_route(args) -> dict:
    """tools/clib_route_timing.py as a child process: the reference's own way in (clib.rs:94-410) -- initialise, one merge_nodes per
    link, build_graph, compute into clib.rs-sized arrays -- on a unitig graph with a compacted de Bruijn graph's degree structure,
    8 M unitigs unless the bench graph is smaller (the link list is made with numpy: 2^27's 65 M unitigs would take minutes)."""
    import subprocess

    unitigs = min(8_000_000, max(1 << (args.log2_edges - 1), 1024))
    out = {"note": "mtg_graph_builder_new / _merge_links (= matchtigs_merge_nodes from one C loop) / _build + mtg_compute_tigs_clib on "
                   "synth.dbg_like_links; the builder is host code (union-find in the order of the calls: sequential by the API's shape); "
                   "second repetition of the process reported"}
    for mode in ("device", "host"):
        if mode == "host" and args.euler == "device":
            continue
        cmd = [sys.executable, str(ROOT / "tools" / "clib_route_timing.py"), "--unitigs", str(unitigs), "--k", str(args.k), "--reps", "2",
               "--compute", mode]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
            if r.returncode != 0 or len(lines) < 2:
                out[mode] = {"error": f"rc={r.returncode}: {r.stderr[-300:]}"}
                continue
            out[mode] = {kk: vv for kk, vv in lines[1].items() if kk not in ("rep", "euler_mode")}
        except (subprocess.TimeoutExpired, OSError, ValueError) as e:
            out[mode] = {"error": repr(e)[:300]}
    return out


def host_free_bytes() -> int:
    """Host memory this process can still take: MemAvailable, and what the cgroup leaves."""
    avail = 1 << 62
    try:
        for line in Path("/proc/meminfo").read_text().splitlines():
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) << 10
    except (OSError, ValueError):
        pass
    try:
        lim = Path("/sys/fs/cgroup/memory.max").read_text().strip()
        cur = int(Path("/sys/fs/cgroup/memory.current").read_text())
        if lim != "max":
            avail = min(avail, int(lim) - cur)
    except (OSError, ValueError):
        pass
    return avail


def full_size_step(args, k: int, device_id: int):
    """ONE device-mode step of the whole path at G-csr 2^full_size_log2 (2^30 = "human-like" nominal), cold, on a graph
    generated on the GPU -- only when the box has the memory (170 GB of HBM, 140 GB of host memory free at 2^30)."""
    from matchtigs_amd import api, synth, torch_glue

    lg = args.full_size_log2
    scale = 1 << max(lg - 30, 0)
    free_hbm, _ = torch.cuda.mem_get_info()
    free_host = host_free_bytes()
    need_hbm, need_host = 170 * scale << 30, 140 * scale << 30
    if free_hbm < need_hbm or free_host < need_host:
        return {"log2_edges": lg, "skipped": f"needs {need_hbm >> 30} GB of HBM and {need_host >> 30} GB of host memory free; "
                                             f"this box has {free_hbm >> 30} / {free_host >> 30} GB"}
    stream = torch_glue.current_stream_ptr()
    t0 = time.perf_counter()
    g = synth.g_csr_device(int((1 << lg) / 1.5 / 2), seed=args.seed, k=k, device_id=device_id)
    t1 = time.perf_counter()
    d = api.DeviceGraph(g, k, device_id, reserve_work=True)
    d.set_plan(args.plan)
    # the caller's own buffers exist before its call, like the device graph: the candidate pool + (start, count) are torch tensors (a
    # caller's, so that they can be all-gathered), 11 GB at 2^30, whose first allocation by torch's allocator costs 0.24 s there; their
    # size comes from the source count, so the classification runs once ahead (and again inside the timed step)
    S0 = d.classify(stream)
    bufs_fs = [torch_glue.CandidateBuffers(S0, max(1024, 4 * S0))]  # (they stay for the second step)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    arena0 = api.device_arena_stats(device_id, reset_peak=True)

    def one_step():
        ta = time.perf_counter()
        S_ = d.classify(stream)
        b = bufs_fs[0] = torch_glue.run_sssp(d, 0, S_, bufs_fs[0])
        st_ms, lv = d.last_sssp_kernel_ms(), d.last_sssp_levels()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        npairs = d.replay_claims_resident(b.start.data_ptr(), b.count.data_ptr(), b.pool.data_ptr(), stream)
        rp_ = d.last_replay_ms()
        tc = time.perf_counter()
        lim, ed = api.finish_greedytigs_resident_np(g, d, k, api.EulerMode.Device, device_id, api.FinishStage.Auto)
        td = time.perf_counter()
        fs_, hp_ = api.last_finish_device_stage_ms(), api.last_phase_seconds()
        res = dict(S=S_, stage_ms=st_ms, levels=lv, n_pairs=npairs, rp=rp_, fs=fs_, hp=hp_, n_tigs=int(len(lim)), n_tig_edges=int(len(ed)),
                   e_after=int(g.edge_count()), t=(ta, tb, tc, td))
        del lim, ed
        return res

    r1 = one_step()
    arena1 = api.device_arena_stats(device_id)
    g.reset()
    r2 = one_step()  # the same step again: work arrays, result arrays and the graph's device cache exist now
    S, stage_ms, levels, n_pairs, rp, fs, hp = r1["S"], r1["stage_ms"], r1["levels"], r1["n_pairs"], r1["rp"], r1["fs"], r1["hp"]
    n_tigs, n_tig_edges, e_after = r1["n_tigs"], r1["n_tig_edges"], r1["e_after"]
    _, t3, t4, t5 = r1["t"]
    w0, w1, w2, w3 = r2["t"]
    warm = {"ms_step": round((w3 - w0) * 1e3, 1),
            "phases_ms": {"classify_sssp": round((w1 - w0) * 1e3, 2), "replay": round((w2 - w1) * 1e3, 2), "finish": round((w3 - w2) * 1e3, 1),
                          "insert_eulerise": round(r2["hp"]["eulerise"] * 1e3, 1), "euler": round(r2["hp"]["euler"] * 1e3, 1),
                          "cut": round(r2["hp"]["cut"] * 1e3, 1)},
            "sssp_stage_ms": round(r2["stage_ms"], 3), "replay_rounds_kernel_ms": round(r2["rp"]["rounds_kernel_ms"], 3),
            "decomposition_gpu_ms": round(r2["fs"]["decomposition_ms"], 2), "insert_eulerise_gpu_ms": round(r2["fs"]["insert_eulerise_ms"], 2),
            "cut_gpu_ms": round(r2["fs"]["cut_ms"], 2), "tigs": r2["n_tigs"]}
    vis = d.sssp_count_visited(0, S, stream) if d.prunes() else None
    full = d.sssp_count(0, S, stream)
    bytes_v = algorithmic_bytes(vis) if vis else algorithmic_bytes(full)
    out = {"log2_edges": lg, "size_label": size_label(lg), "V": int(g.node_count()), "E": int(g.original_edge_count()),
           "sources": int(S), "pairs": int(n_pairs), "tigs": n_tigs, "tig_edges": n_tig_edges, "darts_after_finish": e_after,
           "ms_step": round((t5 - t2) * 1e3, 1), "generate_s": round(t1 - t0, 2), "device_graph_build_s": round(t2 - t1, 2),
           "phases_ms": {"classify_sssp": round((t3 - t2) * 1e3, 2), "replay": round((t4 - t3) * 1e3, 2), "finish": round((t5 - t4) * 1e3, 1),
                         "insert_eulerise": round(hp["eulerise"] * 1e3, 1), "euler": round(hp["euler"] * 1e3, 1), "cut": round(hp["cut"] * 1e3, 1)},
           "sssp_stage_ms": round(stage_ms, 3), "sssp_levels": [{"kernel": x["kernel"], "ms": round(x["ms"], 3), "sources": x["sources"]} for x in levels],
           "frac": round(bytes_v / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if stage_ms > 0 else None,
           "frac_full_ball_equivalent": round(algorithmic_bytes(full) / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if stage_ms > 0 else None,
           "units_full_ball": full, "units_visited": vis,
           "replay_rounds_kernel_ms": round(rp["rounds_kernel_ms"], 3), "decomposition_gpu_ms": round(fs["decomposition_ms"], 2),
           "insert_eulerise_gpu_ms": round(fs["insert_eulerise_ms"], 2), "cut_gpu_ms": round(fs["cut_ms"], 2),
           "second_step": warm,
           # the library's device memory (hip_util.hpp: DeviceArena) before the first step and after it: what the step's work arrays took
           # beside the device graph, and how many driver allocations the step itself made (0 = everything was reserved ahead)
           "arena": {"live_before_gb": round(arena0["live_bytes"] / 1e9, 2), "chunks_before_gb": round(arena0["chunk_bytes"] / 1e9, 2),
                     "peak_in_step_gb": round(arena1["peak_bytes"] / 1e9, 2), "chunks_after_gb": round(arena1["chunk_bytes"] / 1e9, 2),
                     "driver_allocations_in_step": arena1["driver_allocations"] - arena0["driver_allocations"]},
           "note": "one cold device-mode step (every stage on the GPU, pairs resident, tigs downloaded) and the same step once more (second_step: "
                   "the library's work arrays, the result arrays and the graph's device cache exist by then); the device graph -- created with "
                   "MTG_DEVICE_RESERVE_WORK, so that the step's work arrays are ranges of memory the arena already holds -- and the caller's "
                   "candidate buffers exist before the clock starts, like the headline's"}
    bufs_fs[0] = None
    del d, g
    torch.cuda.empty_cache()
    api.release_device_memory(device_id)
    return out


def sssp_stage_of_seeds(args, seeds, device_id) -> dict:
    """SURVEY 8d names seeds {1, 2, 3}: the SSSP stage (HIP events, sum of the level kernels) on the same workload from other seeds."""
    from matchtigs_amd import api, synth, torch_glue

    out = {}
    for seed in seeds:
        if seed == args.seed:
            continue
        g = synth.g_csr_device(int((1 << args.log2_edges) / 1.5 / 2), seed=seed, k=args.k, device_id=device_id)
        d = api.DeviceGraph(g, args.k, device_id)
        d.set_plan(args.plan)
        S = d.classify(torch_glue.current_stream_ptr())
        b = torch_glue.run_sssp(d, 0, S)
        ms = []
        for _ in range(3):
            b = torch_glue.run_sssp(d, 0, S, b)
            ms.append(d.last_sssp_kernel_ms())
        st = d.sssp_count(0, S, torch_glue.current_stream_ptr())
        vis = d.sssp_count_visited(0, S, torch_glue.current_stream_ptr()) if d.prunes() else st
        out[str(seed)] = {"V": g.node_count(), "E": g.edge_count(), "sources": int(S), "sssp_stage_ms": round(float(np.mean(ms)), 4),
                          "frac": round(algorithmic_bytes(vis) / (float(np.mean(ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                          "frac_full_ball_equivalent": round(algorithmic_bytes(st) / (float(np.mean(ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}
        del b, d, g
        torch.cuda.empty_cache()
    return out


def run_cpu_baseline(bg, k: int, budget_s: float, dev, stream, full_ball_edges_all: int) -> dict:
    """Oracle (C restatement of the reference's 1-thread CPU path) timed on the same workload, bounded to ~budget_s:
    the WHOLE path (claim loop + Eulerisation + Euler decomposition + cut) when the graph is small enough to finish in
    the budget, else the Dijkstra + claim phase on a prefix of the sources. `value` is in the unit of the headline value:
    FULL-BALL SSSP edges of the sampled sources (counted by the GPU's counting kernel -- the work the sample stands for)
    per CPU second, so value(GPU) / value(CPU) is a ratio of seconds for the same work. The CPU's own truncated searches
    examine fewer edges (reference semantics: stop after demand + 1 hits); that figure is reported beside it."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib

    og = oracle_lib.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    t0 = time.perf_counter()
    _, st = og.greedy_pairs_np(k, 50000)
    dt = max(time.perf_counter() - t0, 1e-6)
    n_sources = len(og.classify()[0])
    est_whole = dt * n_sources / 50000 * 3.0  # the Euler stages cost about twice the claim loop
    if est_whole <= 2.5 * budget_s:
        stages, st = og.whole_path_timed(k)
        total = sum(stages.values())
        return {
            "value": round(full_ball_edges_all / total, 1), "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": f"the same graph, whole path on 1 core (oracle/mtg_oracle.c: reference-style truncated Dijkstra + claim "
                      f"loop, Eulerisation, literal Hierholzer, cut), {total:.1f} s; value = the graph's {full_ball_edges_all} "
                      f"full-ball SSSP edges (the unit of the headline value) / those seconds",
            "seconds": round(total, 2), "stage_seconds": {kk: round(v, 3) for kk, v in stages.items()},
            "cpu_examined_edges": st["relaxed_edges"],
            "cpu_examined_edges_per_s_dijkstra_phase": round(st["relaxed_edges"] / stages["dijkstra_claim"], 1),
            "settled_nodes": st["settled_nodes"], "queries": st["queries"], "pairs": st["pairs"], "tigs": st["tigs"],
            "multi_thread": run_cpu_baseline_mt(bg, k, stages),
        }
    n1 = int(min(n_sources, 400000))          # second calibration sample: the first 50 000 sources run on cold caches
    t0 = time.perf_counter()
    og.greedy_pairs_np(k, n1)
    rate = n1 / max(time.perf_counter() - t0, 1e-6)
    n = int(min(n_sources, max(50000, rate * budget_s * 0.5)))  # half of the budget for 1 core, half for all cores
    t0 = time.perf_counter()
    _, st = og.greedy_pairs_np(k, n)
    dt = time.perf_counter() - t0
    full_ball_prefix = dev.sssp_count(0, n, stream)["relaxed_edges"]
    # the reference's multi-threaded path (greedytigs/mod.rs:528-644: worker threads over chunks of out_nodes, per-node locks) on
    # the cores this box gives the process, over a prefix sized to the other half of the budget
    cores = host_cores()
    n_mt = int(min(n_sources, max(50000, rate * budget_s * 0.5 * min(cores, 16) * 0.5)))
    t0 = time.perf_counter()
    pairs_mt, st_mt = og.greedy_pairs_np(k, n_mt, threads=cores)
    dt_mt = time.perf_counter() - t0
    full_ball_mt = dev.sssp_count(0, n_mt, stream)["relaxed_edges"]
    return {
        "value": round(full_ball_prefix / dt, 1), "unit": "edges/s", "cores": 1, "kind": "port",
        "sample": f"Dijkstra + claim stage only (greedytigs/mod.rs:276-526, oracle og_greedy_pairs_prefix) on the first {n} of "
                  f"{n_sources} sources of the same graph, {dt:.1f} s on 1 core; value = the {full_ball_prefix} full-ball SSSP edges "
                  f"of those sources / those seconds; compare with gpu_same_stage_edges_per_s (the GPU's classify + SSSP + claim "
                  f"stages over ALL sources), not with the whole-step headline value",
        "stage": "dijkstra_claim", "sources": n, "sources_per_s": round(n / dt, 1), "seconds": round(dt, 2),
        "whole_stage_seconds_estimate": round(dt * n_sources / n, 1),
        "cpu_examined_edges": st["relaxed_edges"], "cpu_examined_edges_per_s": round(st["relaxed_edges"] / dt, 1),
        "settled_nodes": st["settled_nodes"], "queries": st["queries"],
        "multi_thread": {
            "value": round(full_ball_mt / dt_mt, 1), "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"the same stage with the reference's worker-thread scheme (oracle og_greedy_pairs_mt) on {cores} threads = the "
                      f"cores this box gives the process, first {n_mt} of {n_sources} sources, {dt_mt:.1f} s; value = their {full_ball_mt} "
                      f"full-ball SSSP edges / those seconds",
            "sources": n_mt, "seconds": round(dt_mt, 2), "sources_per_s": round(n_mt / dt_mt, 1),
            "whole_stage_seconds_estimate": round(dt_mt * n_sources / n_mt, 1),
            "cpu_examined_edges_per_s": round(st_mt["relaxed_edges"] / dt_mt, 1), "pairs": int(len(pairs_mt)),
        },
    }


def whole_path_sample(args, k: int, device_id: int) -> dict:
    """Whole path (Dijkstra + claims, Eulerisation, Euler walk, cut) by the oracle on 1 core, and the whole step of this engine in both
    Euler modes, on the SAME G-csr 2^24 graph (1/8 of the bench workload's edges; ~10 s of CPU work)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    from matchtigs_amd import api, synth, torch_glue

    lg = 24
    g = synth.g_csr_device(int((1 << lg) / 1.5 / 2), seed=args.seed, k=k, device_id=device_id)
    d = api.DeviceGraph(g, k, device_id)
    d.set_plan(args.plan)
    stream = torch_glue.current_stream_ptr()
    gpu_ms = {}
    for name, mode in (("device", api.EulerMode.Device), ("host", api.EulerMode.HostReferenceOrder)):
        best = None
        for _ in range(2):  # (the second run is the warm one)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            S = d.classify(stream)
            b = torch_glue.run_sssp(d, 0, S)
            d.replay_claims_resident(b.start.data_ptr(), b.count.data_ptr(), b.pool.data_ptr(), stream)
            lim, ed = api.finish_greedytigs_resident_np(g, d, k, mode, device_id, api.FinishStage.Auto)
            n_tigs = int(len(lim))
            del lim, ed
            g.reset()
            best = (time.perf_counter() - t0) * 1e3
        gpu_ms[name] = round(best, 2)
    full = d.sssp_count(0, S, stream)["relaxed_edges"]
    ex = g.export()
    og = oracle_lib.OracleGraph.from_arrays(ex["mirror"], ex["edge_from"], ex["edge_to"], ex["edge_weight"])
    del ex
    stages, st = og.whole_path_timed(k)
    total = sum(stages.values())
    del d, g
    torch.cuda.empty_cache()
    return {"workload": f"G-csr 2^{lg} (the bench generator and seed, 1/{1 << (args.log2_edges - lg)} of its edges)", "cores": 1, "kind": "port",
            "cpu_whole_path_seconds": round(total, 2), "cpu_stage_seconds": {kk: round(v, 3) for kk, v in stages.items()},
            "cpu_value_edges_per_s": round(full / total, 1), "cpu_tigs": st["tigs"], "gpu_tigs": n_tigs,
            "gpu_step_ms_reference_order": gpu_ms["host"], "gpu_step_ms_device_mode": gpu_ms["device"],
            "gpu_value_edges_per_s_reference_order": round(full / (gpu_ms["host"] * 1e-3), 1),
            "note": "same graph, same unit (its full-ball SSSP edges / whole-path seconds); the GPU figures include every download to the host"}


def run_cpu_baseline_mt(bg, k: int, stages_1core: dict) -> dict:
    """The Dijkstra + claim stage again with the reference's worker threads on all host cores (oracle og_greedy_pairs_mt;
    timing-dependent result like the reference with -t > 1). The later stages are sequential in the reference too, so the
    whole-path estimate reuses their 1-core times."""
    import oracle_lib

    cores = host_cores()
    og = oracle_lib.OracleGraph.from_arrays(bg.mirror, bg.edge_from, bg.edge_to, bg.edge_weight)
    t0 = time.perf_counter()
    pairs, st = og.greedy_pairs_np(k, threads=cores)
    dt = time.perf_counter() - t0
    rest = sum(v for kk, v in stages_1core.items() if kk != "dijkstra_claim")
    return {"cores": cores, "dijkstra_claim_seconds": round(dt, 3), "whole_path_seconds_estimate": round(dt + rest, 2),
            "cpu_examined_edges_per_s_dijkstra_phase": round(st["relaxed_edges"] / dt, 1), "pairs": int(len(pairs))}


if __name__ == "__main__":
    main()
