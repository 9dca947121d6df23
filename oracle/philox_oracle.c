/* TEST INFRASTRUCTURE ONLY: plain-C restatement of the counter-based noise generator of the library
 * (tgru_uniform in include/tgru.h), which stands where the reference draws `torch.rand_like(x)`
 * (outdated_files/TimeGANv1.py:762,793).  The generator is Philox4x32-10 (Salmon, Moraes, Dror, Shaw:
 * "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 library's philox4x32 with ten rounds);
 * pinned by that library's published known-answer vectors in tests/test_oracle_cpu.py.
 * Element e of stream s under seed k:  word (e mod 4) of philox(counter = {e/4 low, e/4 high, s low, s high},
 * key = {k low, k high}), mapped to [0, 1) as (word >> 8) * 2^-24. */
#include <stdint.h>

void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void philox_uniform_f32(float *out, int64_t n, uint64_t seed, uint64_t stream, uint64_t first) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t e = first + (uint64_t)i, q = e >> 2;
        const uint32_t ctr[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
        uint32_t w[4];
        philox4x32_10(ctr, key, w);
        out[i] = (float)(w[e & 3] >> 8) * (1.0f / 16777216.0f);
    }
}
