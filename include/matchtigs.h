/*
 * matchtigs.h -- the drop-in C-ABI of libmatchtigs (MI355X-native build).
 *
 * These five entry points are exactly what the reference exports from src/clib.rs (crate
 * `matchtigs` 2.1.9, built as `dylib` named libmatchtigs, Cargo.toml:16-19). A program that links
 * the Rust libmatchtigs.so (e.g. GGCAT, README.md:11-13) links this library unchanged.
 *
 * Types: Rust `usize` = size_t, `isize` = ptrdiff_t, `bool` = 1 byte holding 0/1.
 * Errors: the reference has no error codes -- assert!/panic!/unwrap abort the process across
 * the FFI. This library does the same: it prints "libmatchtigs: <reason>" to stderr and abort()s
 * on null pointers, an unknown algorithm id, a broken node pairing / mirror property, a graph that
 * cannot be made Eulerian, a missing GPU, or any HIP error. There is NO CPU fallback.
 *
 * Algorithms served on the MI355X path: 1 (unitigs), 3 (eulertigs), 5 (greedy matchtigs) and 4
 * (optimal matchtigs: the matching instance is built from the GPU's candidate lists and the matcher named by
 * `matcher_path` -- blossom5 or compatible, found through PATH like the reference's Command::new -- runs as a
 * child process, exactly as in the reference). Id 2 (pathtigs) is outside this engine's scope and aborts with
 * a message saying so.
 */
#ifndef MATCHTIGS_H
#define MATCHTIGS_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque graph-under-construction handle. Replaces `pub struct MatchtigsData` (src/clib.rs:40-43). */
typedef struct MatchtigsData MatchtigsData;

/* Replaces `matchtigs_initialise` (src/clib.rs:87-92): initialise logging at Info level.
 * Call once before anything else. */
void matchtigs_initialise(void);

/* Replaces `matchtigs_initialise_graph` (src/clib.rs:94-102): allocate the builder with a
 * union-find over 4 * unitig_amount unitig-end slots. The caller owns the pointer until it is
 * handed to matchtigs_compute_tigs, which frees it. */
MatchtigsData *matchtigs_initialise_graph(size_t unitig_amount);

/* Replaces `matchtigs_merge_nodes` (src/clib.rs:124-170): one de Bruijn link unitig_a -> unitig_b;
 * strand true = forward, false = reverse complement. Slots per unitig u: forward-in 4u,
 * backward-out 4u+1, forward-out 4u+2, backward-in 4u+3 (src/clib.rs:104-122); unions
 * (out_a, in_b) and (mirror_in_a, mirror_out_b). */
void matchtigs_merge_nodes(MatchtigsData *matchtigs_data, size_t unitig_a, bool strand_a,
                           size_t unitig_b, bool strand_b);

/* Replaces `matchtigs_build_graph` (src/clib.rs:172-259): unitig_weights[unitig_amount] = k-mers
 * per unitig (borrowed). Node id = rank of the union-find representative among the sorted distinct
 * representatives; unitig u becomes edge 2u (n1 -> n2, forwards) and edge 2u+1
 * (mirror_n2 -> mirror_n1, backwards). Aborts if node pairing or the edge mirror property fails. */
void matchtigs_build_graph(MatchtigsData *matchtigs_data, const size_t *unitig_weights);

/* Replaces `matchtigs_compute_tigs` (src/clib.rs:261-410). Consumes and frees matchtigs_data.
 * tig_algorithm as IMPLEMENTED by the reference (src/clib.rs:350-391; its doc comment at :264 swaps
 * 4 and 5): 1 unitigs, 2 pathtigs, 3 eulertigs, 4 optimal matchtigs, 5 greedy matchtigs.
 * matching_file_prefix / matcher_path must be non-null C strings even when unused (:299-330).
 * Outputs are caller-allocated: tigs_edge_out[2*E], tigs_insert_out[2*E], tigs_out_limits[E] with
 * E = 2 * unitig_amount (:332-348). Per walk edge: tigs_edge_out = +-unitig_id (negative = reverse
 * complement; unitig 0 cannot carry a sign, :397-398; dummy edges report unitig id 0),
 * tigs_insert_out = 0 for a unitig edge, else the dummy edge's weight (:399-403);
 * tigs_out_limits[i] = exclusive end of tig i (:405-406). Returns the number of tigs.
 * `threads` is accepted for signature compatibility; the result always equals the reference's
 * 1-thread execution order (the only deterministic one). */
size_t matchtigs_compute_tigs(MatchtigsData *matchtigs_data, size_t tig_algorithm, size_t threads,
                              size_t k, const char *matching_file_prefix, const char *matcher_path,
                              ptrdiff_t *tigs_edge_out, size_t *tigs_insert_out,
                              size_t *tigs_out_limits);

#ifdef __cplusplus
}
#endif
#endif /* MATCHTIGS_H */
