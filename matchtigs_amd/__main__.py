"""`python -m matchtigs_amd` -- the reference CLI's flag surface for the path this engine serves.

Flag names, defaults and exclusivity rules follow /root/reference/src/bin.rs:56-205, 850-862 for the subset that maps onto
the engine (the rest of the reference CLI -- GFA/plain-FASTA input, pathtigs, optimal matchtigs, bitvectors -- is out of
scope). All work happens inside libmatchtigs.so; this file only parses flags and prints the reference's
closing log line (bin.rs:1209-1211).
"""
from __future__ import annotations

import argparse
import sys
import time


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="matchtigs_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("--bcalm-in", help="bcalm2/GGCAT unitig fasta (optionally .gz); requires -k (bin.rs:76-83)")
    ap.add_argument("--gfa-in", help="(not served by this engine)")
    ap.add_argument("--fa-in", help="(not served by this engine)")
    ap.add_argument("-k", type=int, help="k-mer size used to build the de Bruijn graph (bin.rs:139-141)")
    ap.add_argument("-t", "--threads", type=int, default=1, help="accepted; results always equal the 1-thread order (bin.rs:148-149)")
    ap.add_argument("--greedytigs-fa-out", help="write greedy matchtigs as fasta (.gz => gzip) (bin.rs:107-109)")
    ap.add_argument("--eulertigs-fa-out", help="write eulertigs as fasta (bin.rs:99-101)")
    ap.add_argument("--greedytigs-gfa-out", help="write greedy matchtigs as GFA (bin.rs:107-109, 667-818)")
    ap.add_argument("--eulertigs-gfa-out", help="write eulertigs as GFA (bin.rs:97-99)")
    ap.add_argument("--greedytigs-duplication-bitvector-out",
                    help="per greedy matchtig a line of 1 (original k-mer) / 0 (duplicate) characters (bin.rs:129-132)")
    ap.add_argument("--matchtigs-fa-out", help="write optimal matchtigs as fasta; needs the external matcher (bin.rs:123-125)")
    ap.add_argument("--matchtigs-gfa-out", help="write optimal matchtigs as GFA (bin.rs:117-119)")
    ap.add_argument("--matchtigs-duplication-bitvector-out", help="duplication bitvector of the optimal matchtigs")
    ap.add_argument("--blossom5-command", default="blossom5", help="the command used to run blossom5 (bin.rs:151-153)")
    ap.add_argument("--compression-level", type=int, default=6, help="0-9 (bin.rs:203-218)")
    ap.add_argument("--device", type=int, default=0, help="GPU ordinal (not in the reference)")
    args = ap.parse_args(argv)

    n_inputs = sum(x is not None for x in (args.bcalm_in, args.gfa_in, args.fa_in))
    if n_inputs == 0:  # bin.rs:855-858
        ap.error("Missing input argument. Specify exactly least one of --fa-in, --gfa-in or --bcalm-in")
    if n_inputs > 1:  # bin.rs:860-862
        ap.error("Too many input arguments. Specify exactly least one of --fa-in, --gfa-in or --bcalm-in")
    if args.bcalm_in is None:
        ap.error("only --bcalm-in is served by the MI355X engine (SURVEY.md 8 f-2)")
    if args.k is None:
        ap.error("--bcalm-in requires -k")
    if not 0 <= args.compression_level <= 9:
        ap.error("compression level must be in 0..9")
    if args.matchtigs_duplication_bitvector_out and not (args.matchtigs_fa_out or args.matchtigs_gfa_out):
        ap.error("--matchtigs-duplication-bitvector-out needs --matchtigs-fa-out or --matchtigs-gfa-out (bin.rs:955-957)")
    if not (args.greedytigs_fa_out or args.eulertigs_fa_out or args.greedytigs_gfa_out or args.eulertigs_gfa_out
            or args.greedytigs_duplication_bitvector_out or args.matchtigs_fa_out or args.matchtigs_gfa_out):
        ap.error("nothing to do: give --greedytigs-fa-out / --greedytigs-gfa-out and/or --eulertigs-fa-out / --eulertigs-gfa-out")

    from . import api

    t0 = time.perf_counter()
    graph, store = api.read_bcalm2(args.bcalm_in, args.k)
    print(f"Loaded {len(store)} unitigs: {graph.node_count()} nodes, {graph.edge_count()} edges in {time.perf_counter() - t0:.1f}s",
          file=sys.stderr)
    for name, alg, out, gfa, dup in (("matchtigs", 4, args.matchtigs_fa_out, args.matchtigs_gfa_out,
                                      args.matchtigs_duplication_bitvector_out),
                                     ("eulertigs", 3, args.eulertigs_fa_out, args.eulertigs_gfa_out, None),
                                     ("greedytigs", 5, args.greedytigs_fa_out, args.greedytigs_gfa_out,
                                      args.greedytigs_duplication_bitvector_out)):
        if not (out or gfa or dup):
            continue
        cfg = None
        if alg == 4:  # bin.rs:1141-1151: the matching files live next to the first matchtigs output; the matcher is looked up on PATH
            import shutil

            matcher = shutil.which(args.blossom5_command) or args.blossom5_command
            cfg = api.MatchtigAlgorithmConfiguration(args.threads, args.k, out or gfa, matcher, device_id=args.device)
        r = api.compute_tigs_to_fasta_file(graph, store, alg, args.k, out, args.compression_level, args.device, gfa_path=gfa,
                                           duplication_bitvector_path=dup, configuration=cfg)
        graph.reset()  # the reference clones the graph per algorithm (bin.rs:1069)
        print(f"Computing {name} took {r['compute_s']:.1f}s and writing took {r['write_s']:.1f}s "
              f"({r['tigs']} tigs, {r['fasta_bytes']} fasta bytes)", file=sys.stderr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
