// dump_fixtures.rs -- run INSIDE the reference crate (algbio/matchtigs 2.1.9) by someone who has cargo; see README.md beside this file.
//
// WRITTEN BLIND: this file has never been compiled (the image this repository is built in has no Rust toolchain and none of the
// crates; tools/ref_fixtures/README.md). It uses nothing but std and the five `extern "C"` functions of the reference's own
// src/clib.rs (signatures copied from clib.rs:90, :97, :135-141, :180-183, :280-290), so what can be wrong is syntax, not an API.
//
// What it does: reads tests/golden/ref_inputs.txt of the matchtigs_amd repository (unitig weights + links, the input form of
// clib.rs:124-259), and for every case and each of the tig algorithms 1 (unitigs), 3 (eulertigs) and 5 (greedy matchtigs; clib.rs:350-391
// as implemented) builds the graph through the C-ABI, computes the tigs with one thread and prints the three output arrays
// (clib.rs:393-407) as one JSON object per line. The output, saved as tests/golden/ref_outputs.jsonl, is what
// tests/test_ref_fixtures.py compares the CPU oracle and the HIP path against, byte for byte: the first reference-PRODUCED
// known answers this path would have (the reference's own test, src/implementation/mod.rs:762-785, asserts nothing).
//
//   cp dump_fixtures.rs <reference>/examples/ && cd <reference>
//   cargo run --release --example dump_fixtures -- <matchtigs_amd>/tests/golden/ref_inputs.txt > ref_outputs.jsonl
//
// Input format (one case after the other; names contain no blanks):
//   case <name> k <k> unitigs <n> links <m>
//   w <n unitig weights: k-mers per unitig>
//   l <unitig_a> <strand_a: 0|1> <unitig_b> <strand_b: 0|1>          (m lines, in the order they are passed to matchtigs_merge_nodes)
use std::ffi::CString;
use std::io::{BufRead, BufReader, Write};

use libmatchtigs::clib::{
    matchtigs_build_graph, matchtigs_compute_tigs, matchtigs_initialise, matchtigs_initialise_graph,
    matchtigs_merge_nodes,
};

struct Case {
    name: String,
    k: usize,
    weights: Vec<usize>,
    links: Vec<(usize, bool, usize, bool)>,
}

fn parse(path: &str) -> Vec<Case> {
    let file = std::fs::File::open(path).expect("cannot open the input file");
    let mut cases: Vec<Case> = Vec::new();
    for line in BufReader::new(file).lines() {
        let line = line.expect("read error");
        let f: Vec<&str> = line.split_whitespace().collect();
        if f.is_empty() || f[0].starts_with('#') {
            continue;
        }
        match f[0] {
            "case" => {
                // case <name> k <k> unitigs <n> links <m>
                assert!(f.len() == 8 && f[2] == "k" && f[4] == "unitigs" && f[6] == "links");
                cases.push(Case {
                    name: f[1].to_string(),
                    k: f[3].parse().unwrap(),
                    weights: Vec::with_capacity(f[5].parse().unwrap()),
                    links: Vec::with_capacity(f[7].parse().unwrap()),
                });
            }
            "w" => {
                let c = cases.last_mut().expect("w before case");
                for x in &f[1..] {
                    c.weights.push(x.parse().unwrap());
                }
            }
            "l" => {
                assert!(f.len() == 5);
                let c = cases.last_mut().expect("l before case");
                c.links.push((
                    f[1].parse().unwrap(),
                    f[2] == "1",
                    f[3].parse().unwrap(),
                    f[4] == "1",
                ));
            }
            other => panic!("unknown line tag {}", other),
        }
    }
    cases
}

fn json_array<T: std::fmt::Display>(v: &[T]) -> String {
    let parts: Vec<String> = v.iter().map(|x| x.to_string()).collect();
    format!("[{}]", parts.join(","))
}

fn main() {
    let path = std::env::args().nth(1).expect("usage: dump_fixtures <ref_inputs.txt>");
    let cases = parse(&path);
    matchtigs_initialise(); // clib.rs:87-92: once
    let empty = CString::new("").unwrap(); // both strings must be non-null even when unused (clib.rs:299-330)
    let stdout = std::io::stdout();
    let mut out = stdout.lock();
    for case in &cases {
        for &algorithm in &[1usize, 3, 5] {
            let unitig_amount = case.weights.len();
            // clib.rs:94-102, :124-170, :172-259 -- the handle is consumed by matchtigs_compute_tigs (clib.rs:291)
            let data = matchtigs_initialise_graph(unitig_amount);
            for &(ua, sa, ub, sb) in &case.links {
                unsafe { matchtigs_merge_nodes(data, ua, sa, ub, sb) };
            }
            unsafe { matchtigs_build_graph(data, case.weights.as_ptr()) };
            // output arrays sized as clib.rs:332-348: 2 * edge_count, 2 * edge_count, edge_count with edge_count = 2 * unitig_amount
            let edge_count = 2 * unitig_amount;
            let mut edge_out: Vec<isize> = vec![0; 2 * edge_count.max(1)];
            let mut insert_out: Vec<usize> = vec![0; 2 * edge_count.max(1)];
            let mut out_limits: Vec<usize> = vec![0; edge_count.max(1)];
            let n_tigs = unsafe {
                matchtigs_compute_tigs(
                    data,
                    algorithm,
                    1, // threads: the one deterministic order the reference has
                    case.k,
                    empty.as_ptr(),
                    empty.as_ptr(),
                    edge_out.as_mut_ptr(),
                    insert_out.as_mut_ptr(),
                    out_limits.as_mut_ptr(),
                )
            };
            let n_edges = if n_tigs == 0 { 0 } else { out_limits[n_tigs - 1] };
            writeln!(
                out,
                "{{\"case\":\"{}\",\"k\":{},\"algorithm\":{},\"tigs\":{},\"edge_out\":{},\"insert_out\":{},\"out_limits\":{}}}",
                case.name,
                case.k,
                algorithm,
                n_tigs,
                json_array(&edge_out[..n_edges]),
                json_array(&insert_out[..n_edges]),
                json_array(&out_limits[..n_tigs])
            )
            .unwrap();
        }
    }
}
