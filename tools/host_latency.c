// host_latency.c -- the floor of the reference-order Euler walk: one dependent random load per hop through a table of RECORD-byte
// records (a random cyclic permutation, so every hop is a miss), on transparent huge pages like the walk's arena.
// usage: host_latency [table GiB = 3] [record bytes = 32] [hops = 20e6] [lines touched per hop = 1]   (DESIGN.md 4.3 quotes its
// output for 2.9 GB / 32 B = the lean records of the 2^27 graph and 23 GB / 256 B = the wide ones; with 4 lines touched per hop
// the successor index is the sum of a word from every line of the record, like a walk step that needs the whole record)
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <time.h>

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rng(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 3.0;
    const size_t rec = argc > 2 ? (size_t)atol(argv[2]) : 32;
    const size_t hops = argc > 3 ? (size_t)atof(argv[3]) : 20000000;
    const size_t lines = argc > 4 ? (size_t)atol(argv[4]) : 1;
    const size_t n = (size_t)(gib * (double)(1ull << 30)) / rec;
    const size_t bytes = n * rec;
    char *t = mmap(0, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (t == MAP_FAILED) { perror("mmap"); return 1; }
    madvise(t, bytes, MADV_HUGEPAGE);
    // Sattolo: one cycle through all n records; record i holds the index of its successor in its first 8 bytes
    uint64_t *perm = malloc(n * 8);
    for (size_t i = 0; i < n; i++) perm[i] = i;
    for (size_t i = n - 1; i > 0; i--) { size_t j = rng() % i; uint64_t x = perm[i]; perm[i] = perm[j]; perm[j] = x; }
    for (size_t i = 0; i < n; i++) {  // the successor is split over the touched lines: every line must arrive before the next hop
        for (size_t l = 1; l < lines && l * 64 < rec; l++) *(uint64_t *)(t + i * rec + l * 64) = l;
        uint64_t rest = perm[i];
        for (size_t l = 1; l < lines && l * 64 < rec; l++) rest -= l;
        *(uint64_t *)(t + i * rec) = rest;
    }
    free(perm);
    struct timespec t0, t1;
    uint64_t idx = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t s = 0; s < hops; s++) {
        uint64_t nx = *(volatile uint64_t *)(t + idx * rec);
        for (size_t l = 1; l < lines && l * 64 < rec; l++) nx += *(volatile uint64_t *)(t + idx * rec + l * 64);
        idx = nx;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double dt = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    printf("dependent random load, %.1f GiB of %zu-byte records, %zu line(s) per hop: %.1f ns per hop (end %llu)\n", gib, rec, lines, dt / hops * 1e9, (unsigned long long)idx);
    return 0;
}
