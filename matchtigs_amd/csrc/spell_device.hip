// spell_device.hip -- tig spelling on the GPU (SURVEY.md 8 f-1 "2-bit packed store + GPU spelling"): walks -> FASTA / GFA text.
//
// Same rules as spell.cpp (/root/reference/src/bin.rs:466-606 and 667-818): per walk i a header (">{i+1}\n" or "S\t{i+1}\t"),
// the first edge's full sequence (reverse-complemented for a backwards edge, :497-501), every following ORIGINAL edge minus its
// overlap with what is already written (offset k-1 after an original edge, k-1-weight after a dummy edge, :533-537; a backwards
// edge appends revcomp(seq[0..len-offset]), :567-596), nothing for dummy edges (:519-531), "\n" at the end (:601).
//
// Layout: the unitig store is packed to 2 bits per base on the device (only ACGT is representable, like the reference's
// DnaAlphabet store; anything else aborts). Every walk position p (one edge of one walk) owns an output region
//   [header bytes if p starts a walk (incl. the "\n" that closes the previous walk)] [its characters]
// whose start is an exclusive prefix sum; the writing kernel is OUTPUT-centric: a workgroup takes 256 consecutive positions, its
// threads sweep the bytes of their joint region in order (coalesced stores), each byte finding its position by a binary search
// over the 257 region starts held in LDS. Streaming, HBM-bound: output bytes written once, packed bases read once.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <string>

#include "device.hpp"
#include "hip_util.hpp"

namespace mtg {

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) MTG_DIE("HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__, __LINE__, #expr); \
    } while (0)

namespace {

constexpr int SP_BLOCK = 256;

__device__ __forceinline__ uint32_t base_code(unsigned char c) {  // A C G T (either case) -> 0..3, anything else -> 4
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

// 16 bases per 32-bit word, base b at bits [2b, 2b+2)
__global__ void pack_kernel(const char *ascii, uint64_t n_bases, uint32_t *packed, unsigned long long *bad) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t b0 = w * 16;
    if (b0 >= n_bases) return;
    uint32_t v = 0;
    for (int i = 0; i < 16 && b0 + i < n_bases; i++) {
        const uint32_t c = base_code((unsigned char)ascii[b0 + i]);
        if (c > 3) { atomicMin(bad, (unsigned long long)(b0 + i)); continue; }
        v |= c << (2 * i);
    }
    packed[w] = v;
}

__device__ __forceinline__ uint32_t digits_of(uint64_t v) {
    uint32_t d = 1;
    while (v >= 10) { v /= 10; d++; }
    return d;
}

struct SpellArgs {
    const uint32_t *edges;        // [P] walk edges
    const uint32_t *walk_start;   // [P + 1] walk index + 1 at the first position of a walk (and n_walks + 1 at position P), else 0
    // (original edge e is unitig e >> 1, forwards iff e is even: host_graph.hpp)
    const uint32_t *dummy_w;      // [n_dummy / 2] weight of the dummy biedge (n_orig + 2 i, n_orig + 2 i + 1)
    const unsigned long long *seq_off;  // [U + 1] base offsets
    const uint32_t *packed;
    uint64_t n_pos;               // P
    uint64_t n_orig;
    uint32_t k;
    uint32_t rec_prefix;          // 1 (">") or 2 ("S\t")
    unsigned char sep;            // '\n' after the number (FASTA) or '\t' (GFA)
    uint64_t head_bytes;          // bytes of the file header line in front of everything (GFA)
};

// bytes position p contributes: [header] + [characters]; also reports the split
__device__ __forceinline__ void position_extent(const SpellArgs &a, uint64_t p, uint32_t &hdr, uint64_t &chars, uint64_t &offset) {
    hdr = 0; chars = 0; offset = 0;
    const uint32_t ws = a.walk_start[p];
    if (ws) hdr = (ws > 1 ? 1u : 0u) + (p < a.n_pos ? a.rec_prefix + digits_of(ws) + 1u : 0u);  // "\n" of the previous walk + header
    if (p >= a.n_pos) return;
    const uint32_t e = a.edges[p];
    if (e >= a.n_orig) return;  // dummy edges emit nothing
    const uint32_t u = e >> 1;
    const uint64_t sl = a.seq_off[u + 1] - a.seq_off[u];
    if (!ws) {
        const uint32_t prev = a.edges[p - 1];
        offset = prev < a.n_orig ? a.k - 1 : a.k - 1 - a.dummy_w[(prev - a.n_orig) >> 1];
    }
    chars = sl > offset ? sl - offset : 0;
}

__global__ void extent_kernel(SpellArgs a, uint32_t *ext_lo, uint32_t *ext_hi) {  // 64-bit extents as two u32 planes for the u32 scan
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > a.n_pos) return;
    uint32_t hdr; uint64_t chars, off;
    position_extent(a, p, hdr, chars, off);
    const uint64_t t = hdr + chars;
    ext_lo[p] = (uint32_t)t;
    ext_hi[p] = (uint32_t)(t >> 32);
}

// exclusive scan of u64 values given as (lo, hi) planes: block sums then carry; 1024 values per block
__global__ __launch_bounds__(1024) void scan64_reduce_kernel(const uint32_t *lo, const uint32_t *hi, uint64_t n, unsigned long long *block_sums) {
    __shared__ unsigned long long ws[16];
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    unsigned long long v = i < n ? ((unsigned long long)hi[i] << 32) | lo[i] : 0ull;
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < 16; w++) s += ws[w];
        block_sums[blockIdx.x] = s;
    }
}
__global__ __launch_bounds__(1024) void scan64_sums_kernel(unsigned long long *block_sums, uint64_t n_blocks, unsigned long long *total, unsigned long long base) {
    __shared__ unsigned long long wt[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = base;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint64_t b = 0; b < n_blocks; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const unsigned long long v = i < n_blocks ? block_sums[i] : 0;
        unsigned long long incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wt[wv] = incl;
        __syncthreads();
        unsigned long long woff = 0;
        for (int j = 0; j < wv; j++) woff += wt[j];
        const unsigned long long c = carry;
        if (i < n_blocks) block_sums[i] = c + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(1024) void scan64_apply_kernel(const uint32_t *lo, const uint32_t *hi, uint64_t n, const unsigned long long *block_off,
                                                            unsigned long long *out) {
    __shared__ unsigned long long wt[16];
    const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long v = i < n ? ((unsigned long long)hi[i] << 32) | lo[i] : 0ull;
    unsigned long long incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wt[wv] = incl;
    __syncthreads();
    unsigned long long woff = 0;
    for (int j = 0; j < wv; j++) woff += wt[j];
    if (i < n) out[i] = block_off[blockIdx.x] + woff + incl - v;
}

// a workgroup writes the joint region of SP_BLOCK consecutive positions, byte by byte in output order
__global__ __launch_bounds__(SP_BLOCK) void spell_kernel(SpellArgs a, const unsigned long long *start, unsigned long long total, char *out) {
    __shared__ unsigned long long s_start[SP_BLOCK + 1];
    const uint64_t p0 = (uint64_t)blockIdx.x * SP_BLOCK;
    const uint64_t n_here = (a.n_pos + 1 - p0) < (uint64_t)SP_BLOCK ? (a.n_pos + 1 - p0) : (uint64_t)SP_BLOCK;
    for (uint32_t t = threadIdx.x; t <= (uint32_t)n_here; t += SP_BLOCK) s_start[t] = (p0 + t <= a.n_pos) ? start[p0 + t] : total;
    __syncthreads();
    const unsigned long long B0 = s_start[0], B1 = s_start[n_here];
    for (unsigned long long b = B0 + threadIdx.x; b < B1; b += SP_BLOCK) {
        uint32_t lo = 0, hi = (uint32_t)n_here;  // largest t with s_start[t] <= b
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_start[mid] <= b) lo = mid; else hi = mid;
        }
        const uint64_t p = p0 + lo;
        uint32_t hdr; uint64_t chars, off;
        position_extent(a, p, hdr, chars, off);
        uint64_t r = b - s_start[lo];
        char c;
        if (r < hdr) {  // ["\n" of the previous walk] [">" | "S\t"] [digits of walk number] ["\n" | "\t"]
            const uint32_t ws = a.walk_start[p];
            if (ws > 1) {
                if (r == 0) { out[b] = '\n'; continue; }
                r--;
            }
            if (r < a.rec_prefix) c = a.rec_prefix == 1 ? '>' : (r == 0 ? 'S' : '\t');
            else {
                r -= a.rec_prefix;
                const uint32_t nd = digits_of(ws);
                if (r == nd) c = (char)a.sep;
                else {
                    uint64_t v = ws;
                    for (uint32_t q = nd - 1 - (uint32_t)r; q > 0; q--) v /= 10;
                    c = (char)('0' + v % 10);
                }
            }
        } else {
            const uint64_t ci = r - hdr;
            const uint32_t e = a.edges[p];
            const uint32_t u = e >> 1;
            const bool fwd = !(e & 1u);
            const uint64_t bi = a.seq_off[u] + (fwd ? off + ci : chars - 1 - ci);
            uint32_t code = (a.packed[bi >> 4] >> (2 * (bi & 15))) & 3u;
            if (!fwd) code = 3u - code;
            c = "ACGT"[code];
        }
        out[b] = c;
    }
}

__global__ void mark_starts_kernel(const unsigned long long *limits, uint64_t n_walks, uint64_t n_pos, uint32_t *walk_start) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_walks) return;
    const unsigned long long at = i == 0 ? 0ull : limits[i - 1];
    if (i == n_walks) { walk_start[n_pos] = (uint32_t)(n_walks + 1); return; }
    walk_start[at] = (uint32_t)(i + 1);
}

// the same for tigs that are still on the device (32-bit exclusive ends, as the cutter wrote them), with the input checks of the
// host path made here: err[0] = 1 + index of the first empty walk / of the first walk that starts with a dummy edge (atomicMin)
__global__ void mark_starts32_kernel(const uint32_t *limits, const uint32_t *edges, uint64_t n_walks, uint64_t n_pos, uint32_t n_orig,
                                     uint32_t *walk_start, unsigned long long *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_walks) return;
    const uint32_t at = i == 0 ? 0u : limits[i - 1];
    if (i == n_walks) { walk_start[n_pos] = (uint32_t)(n_walks + 1); return; }
    if (limits[i] <= at) { atomicMin(&err[0], (unsigned long long)i); return; }
    if (edges[at] >= n_orig) atomicMin(&err[1], (unsigned long long)i);
    walk_start[at] = (uint32_t)(i + 1);
}

}  // namespace

// Returns the number of bytes; *out_buf is malloc'd (caller frees with mtg_free). kernel_ms_out / bytes_out: the spelling kernel's
// HIP-event time and the HBM bytes it moves (output written once + packed bases read once + per-position metadata).
// (resident: the tigs as a finish on the same GPU left them in HBM -- `limits` / `edges` are then not read, nothing is uploaded for them)
uint64_t device_write_walks_text(const HostGraph &g, uint64_t n_walks, const uint64_t *limits, const uint32_t *edges, uint64_t k,
                                 const char *seqs, const uint64_t *seq_off, bool gfa, const char *gfa_header, int device_id,
                                 char **out_buf, double *kernel_ms_out, uint64_t *bytes_out, const ResidentTigs *resident) {
    if (resident && resident->device != device_id) resident = nullptr;
    if (resident) n_walks = resident->n_tigs;
    if (k < 1) MTG_DIE("k must be >= 1");
    if (device_count() <= device_id) MTG_DIE("no HIP device %d for the GPU tig spelling", device_id);
    if (n_walks >= 0xFFFFFFFEull) MTG_DIE("too many walks for the device spelling");
    HIP_CHECK(hipSetDevice(device_id));
    std::string head;
    if (gfa) head = (gfa_header ? std::string(gfa_header) : "H\tKL:Z:" + std::to_string(k)) + "\n";
    const uint64_t P = resident ? resident->n_edges : (n_walks ? limits[n_walks - 1] : 0);
    const uint64_t n_orig = g.n_original_edges, n_dummy = g.edge_count() - n_orig, U = n_orig / 2;
    const uint64_t n_bases = seq_off[U];
    uint64_t begin = 0;
    for (uint64_t i = 0; i < n_walks && !resident; i++) {  // the same input checks as the host path
        if (limits[i] <= begin) MTG_DIE("empty walk %llu", (unsigned long long)i);
        if (edges[begin] >= n_orig) MTG_DIE("walk %llu starts with a dummy edge (bin.rs:489)", (unsigned long long)i);
        begin = limits[i];
    }
    hipStream_t st = nullptr;
    char *d_ascii = nullptr, *d_out = nullptr;
    uint32_t *d_packed = nullptr, *d_edges = nullptr, *d_ws = nullptr, *d_unitig = nullptr, *d_dw = nullptr, *d_lo = nullptr, *d_hi = nullptr;
    uint8_t *d_fwd = nullptr;
    unsigned long long *d_seq_off = nullptr, *d_limits = nullptr, *d_start = nullptr, *d_bsum = nullptr, *d_tot = nullptr;
    const uint64_t n_words = (n_bases + 15) / 16;
    hu::device_malloc(&d_ascii, std::max<uint64_t>(n_bases, 1));
    hu::device_malloc(&d_packed, std::max<uint64_t>(n_words, 1) * 4);
    hu::device_malloc(&d_tot, 16);
    HIP_CHECK(hipMemcpyAsync(d_ascii, seqs, n_bases, hipMemcpyHostToDevice, st));
    const unsigned long long none = ~0ull;
    HIP_CHECK(hipMemcpyAsync(d_tot + 1, &none, 8, hipMemcpyHostToDevice, st));
    if (n_words) hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, d_ascii, n_bases, d_packed, d_tot + 1);
    HIP_CHECK(hipGetLastError());
    // per-edge tables the positions index (u32 / u8 instead of the host graph's 64-bit fields)
    PodVec<uint32_t> h_dw(std::max<uint64_t>(n_dummy / 2, 1));
    for (uint64_t i = 0; i < n_dummy / 2; i++) h_dw[i] = (uint32_t)std::min<uint64_t>(g.w_biedge[n_orig / 2 + i], 0xFFFFFFFFull);
    hu::device_malloc(&d_dw, std::max<uint64_t>(n_dummy / 2, 1) * 4);
    hu::device_malloc(&d_seq_off, (U + 1) * 8);
    if (!resident) {
        hu::device_malloc(&d_edges, std::max<uint64_t>(P, 1) * 4);
        hu::device_malloc(&d_limits, std::max<uint64_t>(n_walks, 1) * 8);
    }
    unsigned long long *d_err = nullptr;
    hu::device_malloc(&d_err, 16);
    HIP_CHECK(hipMemsetAsync(d_err, 0xFF, 16, st));
    hu::device_malloc(&d_ws, (P + 1) * 4);
    hu::device_malloc(&d_lo, (P + 1) * 4);
    hu::device_malloc(&d_hi, (P + 1) * 4);
    hu::device_malloc(&d_start, (P + 2) * 8);
    const uint64_t nb = (P + 1 + 1023) / 1024;
    hu::device_malloc(&d_bsum, nb * 8);
    if (n_dummy) HIP_CHECK(hipMemcpyAsync(d_dw, h_dw.data(), n_dummy / 2 * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(d_seq_off, seq_off, (U + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(d_ws, 0, (P + 1) * 4, st));
    if (resident) {
        hipLaunchKernelGGL(mark_starts32_kernel, dim3((unsigned)((n_walks + 1 + 255) / 256)), dim3(256), 0, st, resident->d_limits, resident->d_edges, n_walks, P,
                           (uint32_t)n_orig, d_ws, d_err);
    } else {
        if (P) HIP_CHECK(hipMemcpyAsync(d_edges, edges, P * 4, hipMemcpyHostToDevice, st));
        if (n_walks) HIP_CHECK(hipMemcpyAsync(d_limits, limits, n_walks * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(mark_starts_kernel, dim3((unsigned)((n_walks + 1 + 255) / 256)), dim3(256), 0, st, d_limits, n_walks, P, d_ws);
    }
    SpellArgs a{};
    a.edges = resident ? resident->d_edges : d_edges; a.walk_start = d_ws; a.dummy_w = d_dw; a.seq_off = d_seq_off; a.packed = d_packed;
    a.n_pos = P; a.n_orig = n_orig; a.k = (uint32_t)k; a.rec_prefix = gfa ? 2 : 1; a.sep = gfa ? '\t' : '\n'; a.head_bytes = head.size();
    hipLaunchKernelGGL(extent_kernel, dim3((unsigned)((P + 1 + 255) / 256)), dim3(256), 0, st, a, d_lo, d_hi);
    hipLaunchKernelGGL(scan64_reduce_kernel, dim3((unsigned)nb), dim3(1024), 0, st, d_lo, d_hi, P + 1, d_bsum);
    hipLaunchKernelGGL(scan64_sums_kernel, dim3(1), dim3(1024), 0, st, d_bsum, nb, d_tot, (unsigned long long)head.size());
    hipLaunchKernelGGL(scan64_apply_kernel, dim3((unsigned)nb), dim3(1024), 0, st, d_lo, d_hi, P + 1, d_bsum, d_start);
    HIP_CHECK(hipGetLastError());
    unsigned long long h_tot[2] = {0, 0}, h_err[2] = {~0ull, ~0ull};
    HIP_CHECK(hipMemcpyAsync(h_tot, d_tot, 16, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    hu::device_free(d_err);
    if (h_err[0] != ~0ull && h_err[0] < h_err[1]) MTG_DIE("empty walk %llu", h_err[0]);  // (the first offending walk, as the host loop reports it)
    if (h_err[1] != ~0ull) MTG_DIE("walk %llu starts with a dummy edge (bin.rs:489)", h_err[1]);
    if (h_tot[1] != none) MTG_DIE("unitig sequences: character at offset %llu is not in the DNA alphabet (ACGT)", h_tot[1]);
    const uint64_t total = n_walks ? h_tot[0] : head.size();
    char *out = static_cast<char *>(std::malloc(total + 1));
    if (!out) MTG_DIE("out of memory (%llu bytes)", (unsigned long long)total);
    hu::device_free(d_ascii);
    d_ascii = nullptr;
    double ms = 0.0;
    if (n_walks) {
        hu::device_malloc(&d_out, total);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(spell_kernel, dim3((unsigned)((P + 1 + SP_BLOCK - 1) / SP_BLOCK)), dim3(SP_BLOCK), 0, st, a, d_start, (unsigned long long)total, d_out);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(e1, st));
        if (head.size()) HIP_CHECK(hipMemcpyAsync(d_out, head.data(), head.size(), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(out, d_out, total, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        float f = 0.f;
        HIP_CHECK(hipEventElapsedTime(&f, e0, e1));
        ms = f;
        HIP_CHECK(hipEventDestroy(e0));
        HIP_CHECK(hipEventDestroy(e1));
    } else if (head.size()) std::memcpy(out, head.data(), head.size());
    out[total] = '\0';
    if (kernel_ms_out) *kernel_ms_out = ms;
    if (bytes_out) *bytes_out = total + n_bases / 4 + (P + 1) * (8 + 4 + 4);
    for (void *p : {(void *)d_out, (void *)d_packed, (void *)d_edges, (void *)d_ws, (void *)d_unitig, (void *)d_dw, (void *)d_lo, (void *)d_hi,
                    (void *)d_fwd, (void *)d_seq_off, (void *)d_limits, (void *)d_start, (void *)d_bsum, (void *)d_tot})
        hu::device_free(p);
    *out_buf = out;
    return total;
}

}  // namespace mtg
