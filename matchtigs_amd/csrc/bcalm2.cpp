// bcalm2.cpp -- SURVEY.md 8 f-2: BCALM2 / GGCAT unitig FASTA reader -> edge-centric bigraph, and FASTA file output.
//
// Replaces, for the `--bcalm-in X -k K` input route (/root/reference/src/bin.rs:902-912), the third-party reader
// genome_graph::io::bcalm2::read_bigraph_from_bcalm2_as_edge_centric (genome-graph 11.0.0, source absent). The graph is
// built the way the reference's own in-tree specification of that construction does it: src/clib.rs:135-248 -- every
// `L:<strand>:<id>:<strand>` annotation is one matchtigs_merge_nodes call, node ids come from the union-find, unitig u
// becomes edges 2u / 2u+1, weight = len + 1 - k (bin.rs:357-379).
// PARITY NOTE (unpinned): the Rust reader's node numbering (hence the greedy order on real data) cannot be checked here.
//
// Format: `>id LN:i:<len> KC:i:<n> km:f:<x> L:+:<id>:- ...` then the sequence (one or more lines). `.gz` inputs are
// inflated with zlib (gzopen also reads plain files). Only ACGT are representable in the reference's 2-bit store.
#include <zlib.h>

#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_graph.hpp"

namespace mtg {

struct Link {
    uint64_t from_record;
    uint64_t to_id;
    bool from_strand, to_strand;
};

static bool read_line(gzFile f, std::string &line) {
    line.clear();
    char buf[1 << 16];
    for (;;) {
        if (!gzgets(f, buf, sizeof buf)) return !line.empty();
        const size_t n = std::strlen(buf);
        line.append(buf, n);
        if (n && buf[n - 1] == '\n') break;
    }
    while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
    return true;
}

HostGraph *read_bcalm2(const char *path, uint64_t k, UnitigStore **store_out) {
    if (!path || !store_out) MTG_DIE("mtg_read_bcalm2: null argument");
    if (k < 2) MTG_DIE("mtg_read_bcalm2: k must be >= 2");
    gzFile f = gzopen(path, "rb");
    if (!f) MTG_DIE("cannot open %s", path);
    gzbuffer(f, 1 << 20);
    UnitigStore *st = new UnitigStore();
    st->off.push_back(0);
    std::vector<uint64_t> ids;
    std::vector<Link> links;
    std::string line;
    bool have_record = false;
    while (read_line(f, line)) {
        if (line.empty()) continue;
        if (line[0] == '>') {
            if (have_record) st->off.push_back(st->data.size());
            have_record = true;
            const uint64_t rec = ids.size();
            // id
            size_t p = 1;
            while (p < line.size() && line[p] == ' ') p++;
            char *end = nullptr;
            const uint64_t id = std::strtoull(line.c_str() + p, &end, 10);
            if (end == line.c_str() + p) MTG_DIE("%s: record %llu has no numeric id", path, (unsigned long long)rec);
            ids.push_back(id);
            // links
            size_t q = (size_t)(end - line.c_str());
            while ((q = line.find("L:", q)) != std::string::npos) {
                if (q > 0 && line[q - 1] != ' ' && line[q - 1] != '\t') { q += 2; continue; }
                // L:<+|->:<id>:<+|->
                if (q + 3 >= line.size() || (line[q + 2] != '+' && line[q + 2] != '-') || line[q + 3] != ':')
                    MTG_DIE("%s: malformed link annotation in record %llu", path, (unsigned long long)rec);
                const bool s1 = line[q + 2] == '+';
                char *e2 = nullptr;
                const uint64_t to = std::strtoull(line.c_str() + q + 4, &e2, 10);
                if (e2 == line.c_str() + q + 4 || *e2 != ':' || (e2[1] != '+' && e2[1] != '-'))
                    MTG_DIE("%s: malformed link annotation in record %llu", path, (unsigned long long)rec);
                links.push_back(Link{rec, to, s1, e2[1] == '+'});
                q = (size_t)(e2 - line.c_str()) + 2;
            }
        } else {
            if (!have_record) MTG_DIE("%s: sequence data before the first header", path);
            for (char &c : line) {
                if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
                if (c != 'A' && c != 'C' && c != 'G' && c != 'T') MTG_DIE("%s: character '%c' is not in the DNA alphabet", path, c);
            }
            st->data += line;
        }
    }
    gzclose(f);
    if (have_record) st->off.push_back(st->data.size());
    const uint64_t U = ids.size();
    // ids are 0..U-1 in order in BCALM2 output; accept any distinct ids
    bool sequential = true;
    for (uint64_t i = 0; i < U; i++) sequential &= ids[i] == i;
    std::unordered_map<uint64_t, uint64_t> id_map;
    if (!sequential) {
        id_map.reserve(U * 2);
        for (uint64_t i = 0; i < U; i++)
            if (!id_map.emplace(ids[i], i).second) MTG_DIE("%s: duplicate unitig id %llu", path, (unsigned long long)ids[i]);
    }
    std::vector<uint64_t> weights(U);
    for (uint64_t u = 0; u < U; u++) {
        const uint64_t len = st->off[u + 1] - st->off[u];
        if (len < k) MTG_DIE("%s: unitig %llu has length %llu < k = %llu", path, (unsigned long long)ids[u], (unsigned long long)len, (unsigned long long)k);
        weights[u] = len + 1 - k;  // bin.rs:369
    }
    HostGraph *g = builder_new(U);
    for (const Link &l : links) {
        uint64_t to;
        if (sequential) {
            if (l.to_id >= U) MTG_DIE("%s: link to unknown unitig %llu", path, (unsigned long long)l.to_id);
            to = l.to_id;
        } else {
            auto it = id_map.find(l.to_id);
            if (it == id_map.end()) MTG_DIE("%s: link to unknown unitig %llu", path, (unsigned long long)l.to_id);
            to = it->second;
        }
        builder_merge(g, l.from_record, l.from_strand, to, l.to_strand);
    }
    builder_build(g, weights.data());
    *store_out = st;
    return g;
}

// Writes FASTA text to `path`; a ".gz" suffix selects gzip at `compression_level` (bin.rs:203, :442-446: default 6).
void write_file(const char *path, const char *data, uint64_t len, int compression_level) {
    const size_t pl = std::strlen(path);
    if (pl > 3 && std::strcmp(path + pl - 3, ".gz") == 0) {
        char mode[8];
        std::snprintf(mode, sizeof mode, "wb%d", compression_level < 0 ? 6 : (compression_level > 9 ? 9 : compression_level));
        gzFile f = gzopen(path, mode);
        if (!f) MTG_DIE("cannot create %s", path);
        uint64_t done = 0;
        while (done < len) {
            const unsigned chunk = (unsigned)std::min<uint64_t>(len - done, 1u << 30);
            if (gzwrite(f, data + done, chunk) != (int)chunk) MTG_DIE("write error on %s", path);
            done += chunk;
        }
        gzclose(f);
    } else {
        FILE *f = std::fopen(path, "wb");
        if (!f) MTG_DIE("cannot create %s", path);
        if (len && std::fwrite(data, 1, len, f) != len) MTG_DIE("write error on %s", path);
        std::fclose(f);
    }
}

}  // namespace mtg
