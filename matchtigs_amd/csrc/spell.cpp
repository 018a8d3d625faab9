// spell.cpp -- tig spelling: walks -> FASTA text, /root/reference/src/bin.rs:466-606, and GFA text,
// bin.rs:667-818 (same spelling; a header line "H\tKL:Z:{k}" -- or the input file's header -- and records
// "S\t{i+1}\t{sequence}\n" instead of ">{i+1}\n{sequence}\n").
//
// Per walk i: header ">{i+1}\n" (:492); the first edge's full sequence, reverse-complemented for a backwards edge
// (:497-501, :269-285); every following ORIGINAL edge contributes its sequence minus the overlap with what is already
// written: offset = k-1 after an original edge, k-1-weight after a dummy edge (:533-537); a forwards edge appends
// seq[offset..] (:539-566), a backwards edge appends revcomp(seq[0..len-offset]) (:567-596); dummy edges emit
// nothing (:519-531); "\n" closes the record (:601).
//
// Two passes so the output buffer is allocated once and records can be written independently (the second pass is
// embarrassingly parallel over walks; it is a plain loop here and the shape a GPU kernel would take).
#include <cstring>
#include <string>
#include <vector>

#include "host_graph.hpp"

namespace mtg {

static inline char complement(char c) {
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'T': return 'A';
        case 'a': return 't';
        case 'c': return 'g';
        case 'g': return 'c';
        case 't': return 'a';
        default: return 'N';
    }
}

static inline unsigned decimal_digits(uint64_t v) {
    unsigned d = 1;
    while (v >= 10) { v /= 10; d++; }
    return d;
}

// Returns the number of bytes; *out_buf is malloc'd (caller frees with mtg_free).
uint64_t write_walks_text(const HostGraph &g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                          const char *seqs, const uint64_t *seq_off, bool gfa, const char *gfa_header, char **out_buf) {
    if (k < 1) MTG_DIE("k must be >= 1");
    std::string head;  // bin.rs:688-693
    if (gfa) head = (gfa_header ? std::string(gfa_header) : "H\tKL:Z:" + std::to_string(k)) + "\n";
    const uint64_t prefix_len = gfa ? 2 : 1, after_number = 1;  // "S\t" / ">", then the number, then "\t" / "\n"
    const uint64_t n_orig = g.n_original_edges;
    auto seq_len = [&](uint32_t e) -> uint64_t { return seq_off[g.unitig(e) + 1] - seq_off[g.unitig(e)]; };
    // pass 1: record offsets
    std::vector<uint64_t> rec_off(n_walks + 1, 0);
    rec_off[0] = head.size();
    uint64_t begin = 0;
    for (uint64_t i = 0; i < n_walks; i++) {
        const uint64_t end = limits[i];
        if (end <= begin) MTG_DIE("empty walk %llu", (unsigned long long)i);
        uint64_t len = prefix_len + decimal_digits(i + 1) + after_number;
        uint32_t prev = edges[begin];
        if (prev >= n_orig) MTG_DIE("walk %llu starts with a dummy edge (bin.rs:489)", (unsigned long long)i);
        len += seq_len(prev);
        for (uint64_t j = begin + 1; j < end; j++) {
            const uint32_t cur = edges[j];
            if (cur >= n_orig) { prev = cur; continue; }
            const uint64_t offset = prev < n_orig ? k - 1 : k - 1 - g.weight(prev);
            const uint64_t sl = seq_len(cur);
            if (offset > sl) MTG_DIE("overlap %llu longer than unitig %llu", (unsigned long long)offset, (unsigned long long)g.unitig(cur));
            len += sl - offset;
            prev = cur;
        }
        len += 1;  // "\n"
        rec_off[i + 1] = rec_off[i] + len;
        begin = end;
    }
    const uint64_t total = rec_off[n_walks];
    char *out = static_cast<char *>(std::malloc(total + 1));
    if (!out) MTG_DIE("out of memory (%llu bytes)", (unsigned long long)total);
    std::memcpy(out, head.data(), head.size());
    // pass 2: spell
    auto put_edge = [&](char *dst, uint32_t e, uint64_t offset) -> char * {
        const char *s = seqs + seq_off[g.unitig(e)];
        const uint64_t sl = seq_len(e);
        const uint64_t n = sl - offset;
        if (g.forwards(e)) {
            std::memcpy(dst, s + offset, n);
        } else {
            for (uint64_t i = 0; i < n; i++) dst[i] = complement(s[n - 1 - i]);
        }
        return dst + n;
    };
    begin = 0;
    for (uint64_t i = 0; i < n_walks; i++) {
        const uint64_t end = limits[i];
        char *p = out + rec_off[i];
        if (gfa) { *p++ = 'S'; *p++ = '\t'; }  // bin.rs:704
        else *p++ = '>';                       // bin.rs:492
        {
            char num[24];
            unsigned d = decimal_digits(i + 1);
            uint64_t v = i + 1;
            for (unsigned q = d; q-- > 0;) { num[q] = (char)('0' + v % 10); v /= 10; }
            std::memcpy(p, num, d);
            p += d;
        }
        *p++ = gfa ? '\t' : '\n';
        uint32_t prev = edges[begin];
        p = put_edge(p, prev, 0);
        for (uint64_t j = begin + 1; j < end; j++) {
            const uint32_t cur = edges[j];
            if (cur >= n_orig) { prev = cur; continue; }
            const uint64_t offset = prev < n_orig ? k - 1 : k - 1 - g.weight(prev);
            p = put_edge(p, cur, offset);
            prev = cur;
        }
        *p++ = '\n';
        if ((uint64_t)(p - out) != rec_off[i + 1]) MTG_DIE("internal error: record length mismatch");
        begin = end;
    }
    out[total] = '\0';
    *out_buf = out;
    return total;
}

// implementation/mod.rs:668-702: per walk one line with, for every edge, `weight` characters: '1' for an original edge
// (its k-mers are new), '0' for a dummy edge (its k-mers repeat ones spelled elsewhere).
uint64_t write_duplication_bitvector(const HostGraph &g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges,
                                     char **out_buf) {
    std::vector<uint64_t> rec_off(n_walks + 1, 0);
    uint64_t begin = 0;
    for (uint64_t i = 0; i < n_walks; i++) {
        const uint64_t end = limits[i];
        if (end <= begin) MTG_DIE("Found empty walk when writing duplication bitvector (implementation/mod.rs:686)");
        uint64_t len = 1;  // "\n"
        for (uint64_t j = begin; j < end; j++) len += g.weight(edges[j]);
        rec_off[i + 1] = rec_off[i] + len;
        begin = end;
    }
    const uint64_t total = rec_off[n_walks];
    char *out = static_cast<char *>(std::malloc(total + 1));
    if (!out) MTG_DIE("out of memory (%llu bytes)", (unsigned long long)total);
    begin = 0;
    for (uint64_t i = 0; i < n_walks; i++) {
        char *p = out + rec_off[i];
        for (uint64_t j = begin; j < limits[i]; j++) {
            const uint32_t e = edges[j];
            std::memset(p, g.is_dummy(e) ? '0' : '1', g.weight(e));
            p += g.weight(e);
        }
        *p = '\n';
        begin = limits[i];
    }
    out[total] = '\0';
    *out_buf = out;
    return total;
}

uint64_t write_walks_fasta(const HostGraph &g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                           const char *seqs, const uint64_t *seq_off, char **out_buf) {
    return write_walks_text(g, n_walks, limits, edges, k, seqs, seq_off, false, nullptr, out_buf);
}

}  // namespace mtg
