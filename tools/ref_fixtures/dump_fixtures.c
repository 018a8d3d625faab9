/* dump_fixtures.c -- the C twin of dump_fixtures.rs: the same program against the same five extern "C" functions (include/matchtigs.h ==
 * src/clib.rs:90, 97, 135, 180, 280), so it links against EITHER library -- the reference's own dylib or this repository's drop-in
 * libmatchtigs.so -- and prints the same JSON lines. Two uses:
 *   (1) tests/test_ref_fixtures.py builds it against this repository's library on the GPU box and holds its output to the CPU oracle
 *       over every case of tests/golden/ref_inputs.txt: the input format, the JSON format and the consumer are exercised end to end,
 *       and the drop-in boundary is driven by a plain C caller over 219 graphs x 3 algorithms;
 *   (2) someone with the reference's dylib but no wish to touch its Cargo project can link this file against it instead of running the
 *       Rust example:   gcc -std=c99 -O1 dump_fixtures.c -I <include dir with matchtigs.h> -L <reference>/target/release -llibmatchtigs
 *   usage: dump_fixtures <ref_inputs.txt>  > ref_outputs.jsonl
 * Input format: see dump_fixtures.rs. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "matchtigs.h"

typedef struct { size_t ua, ub; int sa, sb; } link_t;

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: dump_fixtures <ref_inputs.txt>\n"); return 2; }
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    matchtigs_initialise(); /* clib.rs:87-92: once */
    size_t cap = 1 << 20;
    char *line = malloc(cap);
    char name[256] = "";
    size_t k = 0, n = 0, m = 0, nw = 0, nl = 0;
    size_t *weights = NULL;
    link_t *links = NULL;
    int have_case = 0, eof = 0;
    while (!eof) {
        size_t len = 0;
        int c;
        while ((c = fgetc(f)) != EOF && c != '\n') {  /* (weight lines of the larger cases are long) */
            if (len + 2 > cap) { cap *= 2; line = realloc(line, cap); }
            line[len++] = (char)c;
        }
        line[len] = 0;
        if (c == EOF) eof = 1;
        int is_case = strncmp(line, "case ", 5) == 0;
        if ((is_case || eof) && have_case) {  /* the case that just ended */
            if (nw != n || nl != m) { fprintf(stderr, "case %s: %zu weights of %zu, %zu links of %zu\n", name, nw, n, nl, m); return 1; }
            static const size_t algorithms[3] = {1, 3, 5};
            for (int ai = 0; ai < 3; ai++) {
                /* clib.rs:94-102, :124-170, :172-259 -- the handle is consumed by matchtigs_compute_tigs (clib.rs:291) */
                MatchtigsData *d = matchtigs_initialise_graph(n);
                for (size_t i = 0; i < m; i++) matchtigs_merge_nodes(d, links[i].ua, links[i].sa != 0, links[i].ub, links[i].sb != 0);
                matchtigs_build_graph(d, weights);
                /* output arrays sized as clib.rs:332-348: 2 * edge_count, 2 * edge_count, edge_count with edge_count = 2 * unitig_amount */
                const size_t ec = 2 * n > 0 ? 2 * n : 1;
                ptrdiff_t *edge_out = calloc(2 * ec, sizeof(ptrdiff_t));
                size_t *insert_out = calloc(2 * ec, sizeof(size_t)), *limits = calloc(ec, sizeof(size_t));
                const size_t nt = matchtigs_compute_tigs(d, algorithms[ai], 1, k, "", "", edge_out, insert_out, limits);
                const size_t ne = nt ? limits[nt - 1] : 0;
                printf("{\"case\":\"%s\",\"k\":%zu,\"algorithm\":%zu,\"tigs\":%zu,\"edge_out\":[", name, k, algorithms[ai], nt);
                for (size_t i = 0; i < ne; i++) printf(i ? ",%td" : "%td", edge_out[i]);
                printf("],\"insert_out\":[");
                for (size_t i = 0; i < ne; i++) printf(i ? ",%zu" : "%zu", insert_out[i]);
                printf("],\"out_limits\":[");
                for (size_t i = 0; i < nt; i++) printf(i ? ",%zu" : "%zu", limits[i]);
                printf("]}\n");
                free(edge_out); free(insert_out); free(limits);
            }
        }
        if (is_case) {
            if (sscanf(line, "case %255s k %zu unitigs %zu links %zu", name, &k, &n, &m) != 4) { fprintf(stderr, "bad case line: %s\n", line); return 1; }
            free(weights); free(links);
            weights = malloc((n ? n : 1) * sizeof(size_t));
            links = malloc((m ? m : 1) * sizeof(link_t));
            nw = nl = 0;
            have_case = 1;
        } else if (line[0] == 'w' && line[1] == ' ') {
            char *p = line + 2;
            while (*p) {
                char *e;
                const unsigned long long v = strtoull(p, &e, 10);
                if (e == p) break;
                if (nw < n) weights[nw] = (size_t)v;
                nw++;
                p = e;
            }
        } else if (line[0] == 'l' && line[1] == ' ') {
            link_t l;
            if (sscanf(line + 2, "%zu %d %zu %d", &l.ua, &l.sa, &l.ub, &l.sb) != 4) { fprintf(stderr, "bad link line: %s\n", line); return 1; }
            if (nl < m) links[nl] = l;
            nl++;
        }
    }
    free(weights); free(links); free(line);
    fclose(f);
    return 0;
}
