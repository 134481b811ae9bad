// Wave-per-board Go rules for gfx950 (wave64). One 64-lane wavefront owns one board; lane l owns
// the points l, l+64, ... (PPL points per lane). Board colours, group labels and per-group
// accumulators live in LDS; cross-lane decisions use wave votes.
//
// Semantics follow go_zero/go.py: groups/liberties (go.py:90-106), capture of dead neighbouring
// groups (go.py:108-120), suicide and situational superko (go.py:122-143), area scoring
// (go.py:208-249). The superko key is a 64-bit Zobrist hash of (board, side to move) instead of
// the blake2b digest of go.py:53-59; set membership is what the rules use, so results agree unless
// two different positions collide in 64 bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace az {

constexpr int kZobSide = 2 * 361;  // index of the side-to-move keys in the Zobrist table

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t zobrist_key(int index) {
    return splitmix64(0x5A0B1D5EED5EEDULL + (uint64_t)(index + 1) * 0x9E3779B97F4A7C15ULL);
}

// Ordering point for a workgroup that is one wave (every kernel that uses this file): LDS accesses of one wave execute
// in issue order, so lanes see each other's earlier LDS writes once the compiler keeps the order; unlike __syncthreads
// this does not wait for outstanding global loads and stores, which may then overlap the board analysis.
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Per-lane geometry: owned points and their four neighbours (-1 = off board / unused slot).
template <int PPL>
struct Geo {
    int p[PPL], up[PPL], dn[PPL], lf[PPL], rt[PPL];
};

template <int PPL>
__device__ inline void geo_init(Geo<PPL>& g, int n, int lane) {
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        int p = lane + 64 * k;
        if (p < n * n) {
            int r = p / n, c = p - r * n;
            g.p[k] = p;
            g.up[k] = r > 0 ? p - n : -1;
            g.dn[k] = r < n - 1 ? p + n : -1;
            g.lf[k] = c > 0 ? p - 1 : -1;
            g.rt[k] = c < n - 1 ? p + 1 : -1;
        } else {
            g.p[k] = g.up[k] = g.dn[k] = g.lf[k] = g.rt[k] = -1;
        }
    }
}

// LDS work area of one board.
template <int PPL>
struct BoardLds {
    static constexpr int NPAD = 64 * PPL;
    int8_t col[NPAD];     // colours
    uint16_t lab[NPAD];   // group label = smallest point index of the connected same-colour region
    uint32_t lib[NPAD];   // at a stone root: number of liberties; at an empty root: border colour mask
    uint32_t gsz[NPAD];   // at a root: number of points in the region
    uint64_t gz[NPAD];    // at a stone root: XOR of the Zobrist keys of its stones
};

// Connected-component labelling by min-label propagation with pointer jumping, then per-group
// liberties / sizes / Zobrist sums through LDS atomics. EMPTIES also labels empty regions and
// records which colours border them (needed only for scoring).
template <int PPL, bool EMPTIES>
__device__ inline void analyze(const Geo<PPL>& g, BoardLds<PPL>& b, const uint64_t* __restrict__ zob) {
    int c[PPL];
    unsigned same[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int p = g.p[k];
        c[k] = 0;
        if (p >= 0) {
            c[k] = b.col[p];
            b.lab[p] = (uint16_t)p;
            b.lib[p] = 0;
            b.gsz[p] = 0;
            b.gz[p] = 0;
        }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        same[k] = 0;
        if (g.p[k] >= 0 && (EMPTIES || c[k] != 0)) {
            if (g.up[k] >= 0 && b.col[g.up[k]] == c[k]) same[k] |= 1u;
            if (g.dn[k] >= 0 && b.col[g.dn[k]] == c[k]) same[k] |= 2u;
            if (g.lf[k] >= 0 && b.col[g.lf[k]] == c[k]) same[k] |= 4u;
            if (g.rt[k] >= 0 && b.col[g.rt[k]] == c[k]) same[k] |= 8u;
        }
    }
    for (;;) {
        bool changed = false;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            if (same[k]) {
                const int p = g.p[k];
                unsigned l = b.lab[p], m = l;
                if (same[k] & 1u) m = min(m, (unsigned)b.lab[g.up[k]]);
                if (same[k] & 2u) m = min(m, (unsigned)b.lab[g.dn[k]]);
                if (same[k] & 4u) m = min(m, (unsigned)b.lab[g.lf[k]]);
                if (same[k] & 8u) m = min(m, (unsigned)b.lab[g.rt[k]]);
                m = min(m, (unsigned)b.lab[m]);
                if (m < l) {
                    b.lab[p] = (uint16_t)m;
                    changed = true;
                }
            }
        }
        wave_sync();
        if (!__any(changed)) break;
    }
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int p = g.p[k];
        if (p < 0) continue;
        if (c[k] != 0) {
            const int r = b.lab[p];
            atomicAdd(&b.gsz[r], 1u);
            atomicXor((unsigned long long*)&b.gz[r], (unsigned long long)zob[2 * p + (c[k] < 0)]);
        } else {
            int r0 = -1, r1 = -1, r2 = -1, r3 = -1;
            unsigned border = 0;
            if (g.up[k] >= 0 && b.col[g.up[k]] != 0) { r0 = b.lab[g.up[k]]; border |= b.col[g.up[k]] > 0 ? 1u : 2u; }
            if (g.dn[k] >= 0 && b.col[g.dn[k]] != 0) { r1 = b.lab[g.dn[k]]; border |= b.col[g.dn[k]] > 0 ? 1u : 2u; }
            if (g.lf[k] >= 0 && b.col[g.lf[k]] != 0) { r2 = b.lab[g.lf[k]]; border |= b.col[g.lf[k]] > 0 ? 1u : 2u; }
            if (g.rt[k] >= 0 && b.col[g.rt[k]] != 0) { r3 = b.lab[g.rt[k]]; border |= b.col[g.rt[k]] > 0 ? 1u : 2u; }
            if (r1 == r0) r1 = -1;
            if (r2 == r0 || r2 == r1) r2 = -1;
            if (r3 == r0 || r3 == r1 || r3 == r2) r3 = -1;
            if (r0 >= 0) atomicAdd(&b.lib[r0], 1u);
            if (r1 >= 0) atomicAdd(&b.lib[r1], 1u);
            if (r2 >= 0) atomicAdd(&b.lib[r2], 1u);
            if (r3 >= 0) atomicAdd(&b.lib[r3], 1u);
            if (EMPTIES) {
                const int r = b.lab[p];
                atomicAdd(&b.gsz[r], 1u);
                if (border) atomicOr(&b.lib[r], border);
            }
        }
    }
    wave_sync();
}

// Place a stone of `color` at p on an analysed board: remove the opponent groups whose only liberty
// was p, update the board hash. Returns the number of captured stones (same in every lane). The
// labels are stale afterwards. The move must be legal.
template <int PPL>
__device__ inline int place_stone(const Geo<PPL>& g, BoardLds<PPL>& b, const uint64_t* __restrict__ zob, int n,
                                  int lane, int p, int color, uint64_t& hash) {
    const int r = p / n, c = p - r * n;
    const int q0 = r + 1 < n ? p + n : -1, q1 = r > 0 ? p - n : -1, q2 = c + 1 < n ? p + 1 : -1,
              q3 = c > 0 ? p - 1 : -1;
    int c0 = -1, c1 = -1, c2 = -1, c3 = -1, captured = 0;
    uint64_t hx = 0;
    if (q0 >= 0 && b.col[q0] == -color) { int rr = b.lab[q0]; if (b.lib[rr] == 1) { c0 = rr; captured += b.gsz[rr]; hx ^= b.gz[rr]; } }
    if (q1 >= 0 && b.col[q1] == -color) { int rr = b.lab[q1]; if (b.lib[rr] == 1 && rr != c0) { c1 = rr; captured += b.gsz[rr]; hx ^= b.gz[rr]; } }
    if (q2 >= 0 && b.col[q2] == -color) { int rr = b.lab[q2]; if (b.lib[rr] == 1 && rr != c0 && rr != c1) { c2 = rr; captured += b.gsz[rr]; hx ^= b.gz[rr]; } }
    if (q3 >= 0 && b.col[q3] == -color) { int rr = b.lab[q3]; if (b.lib[rr] == 1 && rr != c0 && rr != c1 && rr != c2) { c3 = rr; captured += b.gsz[rr]; hx ^= b.gz[rr]; } }
    wave_sync();
    if (captured) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int pt = g.p[k];
            if (pt >= 0 && b.col[pt] == -color) {
                const int l = b.lab[pt];
                if (l == c0 || l == c1 || l == c2 || l == c3) b.col[pt] = 0;
            }
        }
    }
    if (lane == (p & 63)) b.col[p] = (int8_t)color;
    hash ^= zob[2 * p + (color < 0)] ^ hx;
    wave_sync();
    return captured;
}

// Legality of every owned point for `color` to move on an analysed board (go.py:122-143).
// seen[0..ns) holds the situational keys of all earlier positions. legal[k] refers to g.p[k].
template <int PPL>
__device__ inline void legal_points(const Geo<PPL>& g, const BoardLds<PPL>& b, const uint64_t* __restrict__ zob,
                                    int color, uint64_t hash, const uint64_t* seen, int ns, bool legal[PPL]) {
    uint64_t key[PPL];
    const uint64_t side_after = zob[kZobSide + (color > 0 ? 1 : 0)];  // the opponent moves next
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int p = g.p[k];
        legal[k] = false;
        key[k] = 0;
        if (p < 0 || b.col[p] != 0) continue;
        bool haslib = false;
        uint64_t hx = 0;
        int c0 = -1, c1 = -1, c2 = -1;
        const int q[4] = {g.up[k], g.dn[k], g.lf[k], g.rt[k]};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (q[d] < 0) continue;
            const int v = b.col[q[d]];
            if (v == 0) { haslib = true; continue; }
            const int rr = b.lab[q[d]];
            const unsigned L = b.lib[rr];
            if (v == color) {
                if (L >= 2) haslib = true;
            } else if (L == 1) {  // p is that group's last liberty: it is captured
                haslib = true;
                if (rr != c0 && rr != c1 && rr != c2) {
                    hx ^= b.gz[rr];
                    if (c0 < 0) c0 = rr; else if (c1 < 0) c1 = rr; else c2 = rr;
                }
            }
        }
        legal[k] = haslib;  // false = suicide
        key[k] = hash ^ zob[2 * p + (color < 0)] ^ hx ^ side_after;
    }
    // situational superko: every lane compares its candidate keys with every seen key (the same address in all lanes:
    // LDS broadcasts); four keys per trip so that the reads of a trip are in flight together
    int i = 0;
    for (; i + 4 <= ns; i += 4) {
        const uint64_t s0 = seen[i], s1 = seen[i + 1], s2 = seen[i + 2], s3 = seen[i + 3];
#pragma unroll
        for (int k = 0; k < PPL; ++k)
            if (key[k] == s0 || key[k] == s1 || key[k] == s2 || key[k] == s3) legal[k] = false;
    }
    for (; i < ns; ++i) {
        const uint64_t s = seen[i];
#pragma unroll
        for (int k = 0; k < PPL; ++k)
            if (key[k] == s) legal[k] = false;
    }
}

// Area score of a board analysed with EMPTIES (go.py:208-249): stones + empty regions bordered by
// exactly one colour. Wave-wide sums returned in every lane.
template <int PPL>
__device__ inline void area_score(const Geo<PPL>& g, const BoardLds<PPL>& b, int& black, int& white) {
    int bs = 0, ws = 0;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int p = g.p[k];
        if (p < 0) continue;
        const int v = b.col[p];
        if (v > 0) bs += 1;
        else if (v < 0) ws += 1;
        else if (b.lab[p] == p) {
            const unsigned m = b.lib[p] & 3u;
            if (m == 1u) bs += (int)b.gsz[p];
            else if (m == 2u) ws += (int)b.gsz[p];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bs += __shfl_xor(bs, o);
        ws += __shfl_xor(ws, o);
    }
    black = bs;
    white = ws;
}

}  // namespace az
