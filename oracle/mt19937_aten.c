/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the random-number path Marius's link-prediction hot path consumes.
 * The arithmetic lives in a third-party dependency that is absent from /root/reference:
 * libtorch (reference pins only `torch>=1.7.1`, setup.cfg:39; this container has 2.10.0+rocm7.0).
 * Reference call sites that draw from this stream (CPU generator seeded once by
 * torch::manual_seed, src/cpp/src/marius.cpp:47):
 *   - DataLoader::setActiveEdges  -> torch::randperm(E)        src/cpp/src/data/dataloader.cpp:176-182
 *   - CorruptNodeNegativeSampler::getNegatives -> torch::randint(num_nodes,{n_uni})
 *                                                               src/cpp/src/data/samplers/negative.cpp:340-357
 *   - batch_sample -> torch::randint(0, B, {n_deg})             src/cpp/src/data/samplers/negative.cpp:7-19
 *
 * Published algorithm restated here (ATen):
 *   at::mt19937 (ATen/core/MT19937RNGEngine.h): standard MT19937, seeded by init_genrand(seed & 0xffffffff),
 *     first output after one full twist.
 *   CPUGeneratorImpl::random()   = engine()                 (32 bit)
 *   CPUGeneratorImpl::random64() = (engine() << 32) | engine()
 *   uniform_int_from_to_distribution (ATen/core/DistributionsHelper.h:38-58):
 *     range >= 2^28 -> random64() % range + base, else random() % range + base; drawn serially.
 *   randperm_cpu (ATen/native/TensorFactories.cpp): r[i]=i; for i in [0,n-1): z = random() % (n-i); swap(r[i], r[i+z]).
 *
 * Pinned against torch.randint / torch.randperm of this container in tests/test_oracle_rng.py and against
 * the committed vectors in tests/golden/rng_*.json (generator script: tests/golden/make_golden.py).
 */
#include <stdint.h>
#include <stddef.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t state[MT_N];
    int32_t next;   /* index of next untempered word; == MT_N means "twist before next draw" */
} mt_state;

void mt_seed(mt_state* s, uint64_t seed) {
    s->state[0] = (uint32_t)(seed & 0xffffffffu);
    for (int j = 1; j < MT_N; j++) {
        s->state[j] = 1812433253u * (s->state[j - 1] ^ (s->state[j - 1] >> 30)) + (uint32_t)j;
    }
    s->next = MT_N;
}

static void mt_twist(mt_state* s) {
    uint32_t* p = s->state;
    for (int i = 0; i < MT_N; i++) {
        uint32_t y = (p[i] & 0x80000000u) | (p[(i + 1) % MT_N] & 0x7fffffffu);
        uint32_t v = p[(i + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        p[i] = v;
    }
    s->next = 0;
}

uint32_t mt_random(mt_state* s) {
    if (s->next >= MT_N) mt_twist(s);
    uint32_t y = s->state[s->next++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

uint64_t mt_random64(mt_state* s) {
    uint32_t hi = mt_random(s);
    uint32_t lo = mt_random(s);
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

/* raw 32-bit outputs */
void mt_fill_raw(mt_state* s, uint32_t* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) out[i] = mt_random(s);
}

/* torch.randint(low, low+range, {n}) on the CPU generator */
void mt_randint(mt_state* s, int64_t* out, int64_t n, uint64_t range, int64_t base) {
    if (range >= (1ull << 28)) {
        for (int64_t i = 0; i < n; i++) out[i] = (int64_t)(mt_random64(s) % range) + base;
    } else {
        for (int64_t i = 0; i < n; i++) out[i] = (int64_t)((uint64_t)mt_random(s) % range) + base;
    }
}

/* torch.randperm(n) on the CPU generator (ATen/native/TensorFactories.cpp, randperm_cpu): Fisher-Yates with 32-bit draws for
 * n < 2^32 / 20; above that the "inside-out" variant with random64() draws (r[i] = r[z]; r[z] = i, z = random64() % (i + 1)). */
void mt_randperm(mt_state* s, int64_t* out, int64_t n) {
    if ((uint64_t)n >= (0xffffffffull / 20ull)) {
        for (int64_t i = 0; i < n; i++) {
            int64_t z = (int64_t)(mt_random64(s) % (uint64_t)(i + 1));
            out[i] = out[z];
            out[z] = i;
        }
        return;
    }
    for (int64_t i = 0; i < n; i++) out[i] = i;
    for (int64_t i = 0; i < n - 1; i++) {
        int64_t z = (int64_t)((uint64_t)mt_random(s) % (uint64_t)(n - i));
        int64_t sav = out[i];
        out[i] = out[z + i];
        out[z + i] = sav;
    }
}

/*
 * CorruptNodeNegativeSampler::getNegatives (negative.cpp:328-366) for one call:
 * per chunk: n_uni uniform ids (drawn FIRST), then n_deg edge positions (batch_sample);
 * output row = cat({deg_sample, uniform}); deg_pos (C x n_deg) receives the sampled edge positions.
 * edges: [B,3] (or [B,2] when ncols==2) int64; inverse -> take column 0 (src) else last column (dst).
 */
void oracle_get_negatives(mt_state* s, const int64_t* edges, int64_t B, int ncols, int inverse,
                          int64_t num_nodes, int num_chunks, int num_negatives, float degree_fraction,
                          int64_t* out_ids, int64_t* deg_pos) {
    int n_deg = (int)(num_negatives * degree_fraction);
    int n_uni = num_negatives - n_deg;
    for (int c = 0; c < num_chunks; c++) {
        int64_t* row = out_ids + (int64_t)c * num_negatives;
        mt_randint(s, row + n_deg, n_uni, (uint64_t)num_nodes, 0);
        if (degree_fraction > 0) {
            int64_t* pos = deg_pos + (int64_t)c * n_deg;
            mt_randint(s, pos, n_deg, (uint64_t)B, 0);
            for (int k = 0; k < n_deg; k++) {
                const int64_t* e = edges + pos[k] * ncols;
                row[k] = inverse ? e[0] : e[ncols - 1];
            }
        }
    }
}
