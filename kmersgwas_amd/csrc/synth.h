// synth.h — counter-based synthetic .table rows, shared by the device generator
// kernel and its bit-identical host twin.
//   kmer(r)   = r + 1                      (strictly ascending, < 2^62)
//   q(r)      in [5, 250]; per-row presence frequency q/256
//   bit(r, c) ~ Bernoulli(q/256), independent, from splitmix64 keyed by (seed, r, word, digit)
// Rows with q <= 12 or q >= 244 fail a 5 % minor-allele filter, so ~6 % of rows exercise the
// MAC predicate.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KGWAS_HD __host__ __device__
#else
#define KGWAS_HD
#endif

namespace kgwas {

KGWAS_HD inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

KGWAS_HD inline uint32_t synth_q(uint64_t seed, uint64_t r) {
    return 5u + (uint32_t)(splitmix64(seed ^ (r * 0xD1B54A32D192ED03ULL + 0x2545F4914F6CDD1DULL)) % 246u);
}

// word index w: 0 = k-mer id, 1..W_f = presence/absence words.
KGWAS_HD inline uint64_t synth_word(uint64_t seed, uint64_t r, uint32_t w, uint64_t n_acc) {
    if (w == 0) return r + 1;
    const uint32_t q = synth_q(seed, r);
    const uint64_t h = splitmix64(seed + r * 0x9E3779B97F4A7C15ULL);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < 8; i++) {
        uint64_t rnd = splitmix64(h ^ ((uint64_t)(w * 8u + i + 1u) * 0xC2B2AE3D27D4EB4FULL));
        acc = ((q >> i) & 1u) ? (acc | rnd) : (acc & rnd);
    }
    const uint64_t W_f = (n_acc + 63) / 64;
    if (w == W_f && (n_acc & 63)) acc &= ((1ULL << (n_acc & 63)) - 1ULL);
    return acc;
}

}  // namespace kgwas
