"""A plain C program as the caller of the reference's C-ABI (include/matchtigs.h == src/clib.rs:87-410): compiled with gcc, linked
against libmatchtigs.so, no Python, ctypes or torch in the process. It is what GGCAT-style C/C++ callers of the Rust dylib do
(README.md:11-13 of the reference); its output arrays must equal the oracle's restatement of clib.rs:393-407."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

C_MAIN = r"""
#include <stdio.h>
#include <stdlib.h>
#include "matchtigs.h"
%(data)s
int main(void) {
    matchtigs_initialise();
    for (int alg = 0; alg < 3; alg++) {
        static const size_t algs[3] = {5, 3, 1};
        MatchtigsData *d = matchtigs_initialise_graph(N_UNITIGS);
        for (size_t i = 0; i < N_LINKS; i++) matchtigs_merge_nodes(d, links[i][0], links[i][1] != 0, links[i][2], links[i][3] != 0);
        matchtigs_build_graph(d, weights);
        ptrdiff_t *edge_out = malloc(sizeof(ptrdiff_t) * 2 * N_UNITIGS);
        size_t *insert_out = malloc(sizeof(size_t) * 2 * N_UNITIGS), *limits = malloc(sizeof(size_t) * N_UNITIGS);
        size_t n = matchtigs_compute_tigs(d, algs[alg], 1, K, "", "", edge_out, insert_out, limits);  /* consumes d */
        printf("ALG %%zu N %%zu\n", algs[alg], n);
        size_t total = n ? limits[n - 1] : 0;
        printf("L");
        for (size_t i = 0; i < n; i++) printf(" %%zu", limits[i]);
        printf("\nE");
        for (size_t i = 0; i < total; i++) printf(" %%td", edge_out[i]);
        printf("\nI");
        for (size_t i = 0; i < total; i++) printf(" %%zu", insert_out[i]);
        printf("\n");
        free(edge_out); free(insert_out); free(limits);
    }
    return 0;
}
"""


def test_c_program_through_the_clib_abi(product_lib, oracle, tmp_path):
    if product_lib.mtg_device_count() < 1:
        pytest.fail("needs a GPU")
    from matchtigs_amd import _lib, synth

    k = 15
    ug = synth.g_seq(6000, seed=4, k=k, haplotypes=3, sub_rate=0.03)
    weights, links = [int(w) for w in ug.weights], [(int(a), int(bool(b)), int(c), int(bool(d))) for a, b, c, d in ug.links]
    data = (f"#define N_UNITIGS {len(weights)}\n#define N_LINKS {len(links)}\n#define K {k}\n"
            f"static const size_t weights[] = {{{', '.join(map(str, weights))}}};\n"
            f"static const size_t links[][4] = {{{', '.join('{%d,%d,%d,%d}' % l for l in links)}}};\n")
    src = tmp_path / "caller.c"
    src.write_text(C_MAIN % {"data": data})
    exe = tmp_path / "caller"
    libdir = _lib.LIB_PATH.parent
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe),
                    "-L", str(libdir), "-lmatchtigs", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[INFO] Logging initialised successfully" in r.stderr   # clib.rs:87-92
    blocks = r.stdout.strip().split("ALG ")[1:]
    assert len(blocks) == 3
    for blk in blocks:
        lines = blk.splitlines()
        alg, n = int(lines[0].split()[0]), int(lines[0].split()[2])
        lim = [int(x) for x in lines[1].split()[1:]]
        eo = [int(x) for x in lines[2].split()[1:]]
        io = [int(x) for x in lines[3].split()[1:]]
        og = oracle.OracleGraph.from_unitig_links(ug.weights, ug.links)
        wn, we, wi, wl = og.clib_compute_tigs(alg, k)
        assert n == wn and lim == list(wl) and eo == list(we) and io == list(wi), alg
