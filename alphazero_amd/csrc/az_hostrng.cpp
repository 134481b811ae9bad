// Host-side random streams for lockstep self-play: one legacy-NumPy-compatible MT19937 stream per
// game, so that game g draws exactly what `np.random.seed(s_g)` followed by the upstream calls
// would draw: np.random.dirichlet([alpha]*k) at the root (mcts.py:122) and the uniform behind
// np.random.choice(len(pi), p=pi) (mcts.py:281).
//
// Restates NumPy's published legacy algorithms (numpy/random/mtrand.pyx RandomState.dirichlet and
// .choice, numpy/random/src/legacy/legacy-distributions.c legacy_standard_gamma /
// legacy_standard_exponential, numpy/random/src/mt19937/mt19937.c); libm pow/log as NumPy uses.
#include <math.h>
#include <stdint.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/azamd.h"

namespace az {
int fail(int code, const std::string& msg);
}
using az::fail;

namespace {
struct Mt {
    uint32_t key[624];
    int pos;
};

void mt_seed(Mt& s, uint32_t seed) {  // mt19937_seed: Knuth initialisation
    for (int i = 0; i < 624; ++i) {
        s.key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    s.pos = 624;
}

void mt_gen(Mt& s) {
    const uint32_t N = 624, M = 397, MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
    uint32_t y;
    uint32_t i;
    for (i = 0; i < N - M; ++i) {
        y = (s.key[i] & UPPER) | (s.key[i + 1] & LOWER);
        s.key[i] = s.key[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; i < N - 1; ++i) {
        y = (s.key[i] & UPPER) | (s.key[i + 1] & LOWER);
        s.key[i] = s.key[i + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (s.key[N - 1] & UPPER) | (s.key[0] & LOWER);
    s.key[N - 1] = s.key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    s.pos = 0;
}

inline uint32_t mt_next32(Mt& s) {
    if (s.pos == 624) mt_gen(s);
    uint32_t y = s.key[s.pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

inline double mt_double(Mt& s) {  // 53-bit double in [0, 1)
    const int32_t a = (int32_t)(mt_next32(s) >> 5), b = (int32_t)(mt_next32(s) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

inline double legacy_exponential(Mt& s) { return -log(1.0 - mt_double(s)); }

double legacy_gauss(Mt& s, bool& has, double& saved) {
    if (has) {
        has = false;
        const double t = saved;
        saved = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_double(s) - 1.0;
        x2 = 2.0 * mt_double(s) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    saved = f * x1;
    has = true;
    return f * x2;
}

struct Stream {
    Mt mt;
    bool has_gauss = false;
    double gauss = 0.0;
};

double legacy_gamma(Stream& st, double shape) {
    Mt& s = st.mt;
    if (shape == 1.0) return legacy_exponential(s);
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
        for (;;) {
            const double U = mt_double(s), V = legacy_exponential(s);
            if (U <= 1.0 - shape) {
                const double X = pow(U, 1. / shape);
                if (X <= V) return X;
            } else {
                const double Y = -log((1 - U) / shape), X = pow(1.0 - shape + shape * Y, 1. / shape);
                if (X <= (V + Y)) return X;
            }
        }
    }
    const double b = shape - 1. / 3., c = 1. / sqrt(9 * b);
    for (;;) {
        double X, V;
        do {
            X = legacy_gauss(s, st.has_gauss, st.gauss);
            V = 1.0 + c * X;
        } while (V <= 0.0);
        V = V * V * V;
        const double U = mt_double(s);
        if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return (b * V);
        if (log(U) < 0.5 * X * X + b * (1. - V + log(V))) return (b * V);
    }
}
}  // namespace

struct az_rng {
    std::vector<Stream> s;
};

extern "C" {

int az_rng_create(int32_t count, az_rng** out) {
    if (count < 1 || !out) return fail(AZ_ERR_ARG, "az_rng_create: bad argument");
    az_rng* r = new az_rng();
    r->s.resize((size_t)count);
    for (auto& st : r->s) mt_seed(st.mt, 0);
    *out = r;
    return AZ_OK;
}

int az_rng_destroy(az_rng* r) {
    delete r;
    return AZ_OK;
}

int az_rng_seed(az_rng* r, int32_t index, uint32_t seed) {
    if (!r || index < 0 || (size_t)index >= r->s.size()) return fail(AZ_ERR_ARG, "az_rng_seed: bad index");
    mt_seed(r->s[(size_t)index].mt, seed);
    r->s[(size_t)index].has_gauss = false;
    r->s[(size_t)index].gauss = 0.0;
    return AZ_OK;
}

int az_rng_dirichlet(az_rng* r, const int32_t* counts, double alpha, double* out, int32_t stride, int32_t threads) {
    if (!r || !counts || !out || stride < 1) return fail(AZ_ERR_ARG, "az_rng_dirichlet: bad argument");
    const int G = (int)r->s.size();
    for (int g = 0; g < G; ++g)
        if (counts[g] < 0 || counts[g] > stride) return fail(AZ_ERR_ARG, "az_rng_dirichlet: count exceeds stride");
    auto work = [&](int lo, int hi) {
        for (int g = lo; g < hi; ++g) {
            const int k = counts[g];
            if (k == 0) continue;
            double* v = out + (size_t)g * stride;
            double acc = 0.0;
            for (int j = 0; j < k; ++j) {
                v[j] = legacy_gamma(r->s[(size_t)g], alpha);
                acc = acc + v[j];
            }
            const double inv = 1 / acc;
            for (int j = 0; j < k; ++j) v[j] = v[j] * inv;
        }
    };
    const int T = threads < 1 ? 1 : (threads > G ? G : threads);
    if (T == 1) {
        work(0, G);
    } else {
        std::vector<std::thread> pool;
        const int per = (G + T - 1) / T;
        for (int t = 0; t < T; ++t) {
            const int lo = t * per, hi = lo + per < G ? lo + per : G;
            if (lo < hi) pool.emplace_back(work, lo, hi);
        }
        for (auto& th : pool) th.join();
    }
    return AZ_OK;
}

int az_rng_random_sample(az_rng* r, const uint8_t* mask, double* out) {
    if (!r || !out) return fail(AZ_ERR_ARG, "az_rng_random_sample: bad argument");
    const size_t G = r->s.size();
    for (size_t g = 0; g < G; ++g)
        if (!mask || mask[g]) out[g] = mt_double(r->s[g].mt);
    return AZ_OK;
}

}  // extern "C"
