// libazamd engine: G lockstep Go games and their PUCT search trees resident in HBM, one wavefront
// per game (or per pending leaf) for every kernel. See include/azamd.h for the C ABI.
//
// Search semantics follow go_zero/mcts.py line by line in float64: PUCT score and first-strict-max
// tie-break (mcts.py:97-116), virtual loss on the whole path (mcts.py:196-199), masked softmax
// (mcts.py:72-89, including numpy's pairwise summation order), root noise (mcts.py:121-124), top-k
// with forced pass (mcts.py:126-132), expand-once for repeated leaves (mcts.py:159-161), revert +
// backup in slot order (mcts.py:205-218). This file is compiled with -ffp-contract=off so every
// product and sum is rounded separately, as in the Python code.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/azamd.h"
#include "az_board.cuh"
#include "az_net.h"

namespace az {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(x)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess)                                                                   \
            return fail(AZ_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));            \
    } while (0)

enum Phase { PH_IDLE = 0, PH_ROOT = 1, PH_SELECTED = 2, PH_NEED_SELECT = 3, PH_DONE = 4 };
enum { ST_SIMS = 0, ST_ROWS, ST_ROWS_RAW, ST_DEPTH_SUM, ST_MAX_DEPTH, ST_MAX_NODES, ST_TERMINAL, ST_BATCHES };

// Everything a kernel needs, passed by value.
struct Dev {
    int n, np, A, H, C, G, sims, B, K, dedup;
    int npad, seen_cap, max_nodes, max_depth, max_rows, tab_len;
    double komi, eps, one_minus_eps, vl;
    const uint64_t* zob;
    const double* pbc;  // log((N + c_base + 1) / c_base) + c_init, N = parent visits (host libm)
    const double* sqn;  // sqrt(N + 1)
    // root states
    int8_t* board;   // [G][npad]
    int8_t* hist;    // [G][H][npad], newest first
    uint64_t* seen;  // [G][seen_cap]
    uint64_t* hash;  // [G] board-only Zobrist hash
    int* sc;         // [G][8] to_play, passes, move_count, caps_b, caps_w, last_move, hist_count, n_seen
    // trees
    double* prior; double* vsum; int* visits; int16_t* action; int* child_off; int16_t* n_child;  // [G][max_nodes]
    int* n_nodes;     // [G]
    uint8_t* active;  // [G]
    // current batch
    int* path;        // [G][B][max_depth] node indices, path[0] = 0 (root)
    int* path_len;    // [G][B]
    int* leaf_ref;    // [G][B] -1 = terminal, else slot whose row carries this slot's evaluation
    double* term_value;  // [G][B]
    int* n_unique;    // [G]
    int* uniq_slot;   // [G][B]
    int* brec;        // [G][4 + 4 B] batch record of a game's descents (select_body)
    int* row_off;     // [G+1]
    int* row_game; int* row_slot;  // [max_rows]
    uint8_t* row_legal;  // [max_rows][A]
    int* ctl;         // [8] rows to evaluate, error code, rows to gather (>= rows to evaluate: roots with carried-over logits come last), -,
                      //     [4], [5]: row counters of the even / odd batch rounds of az_mcts_run_rounds
    long long* stats; // [8], filled by k_reduce_stats
    // Root-evaluation reuse (rc_cap > 0): the root of the next search is the child played after this one, and that child was
    // evaluated as a leaf of this search; its logits are kept so that a batch-invariant evaluator need not see it again.
    int rc_cap;            // root children covered per game (0: off)
    float* rc_logits;      // [G][rc_cap][A] logits of the root children expanded in the current search
    uint8_t* rc_valid;     // [G][rc_cap]
    float* root_logits;    // [G][A] logits carried over to the next root
    uint8_t* root_cached;  // [G] root_logits holds the evaluation of the current root state
    uint8_t* tree_live;    // [G] the tree in memory was searched from the state before the next move
    long long* gstats; // [G][8] per-game counters of the current search (no cross-workgroup atomics on the hot kernels)
    int* root_count;   // [G] legal actions (pass included) of each root whose row the last root-phase leaf gather wrote
    int* ctl_mirror;   // null, or [4]: the root-phase row scan also leaves its counts here (a self-play lane's record)
};

// -DAZ_ESTAMP (diagnostic builds, tools/estamp_probe.py): lane 0 of every wave of the search kernels leaves the 100 MHz
// real-time counter at phase boundaries; az_debug_estamps copies the table out. The regular build has no stamps.
#ifdef AZ_ESTAMP
__device__ unsigned long long g_estamp[4096 * 16];
#define ESTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_estamp[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ESTAMP_CYCLES(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_estamp[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// what each of the first 16 descents of the last batch did: path length | terminal << 8 | replayed << 9 | level loads << 12 |
// nodes picked from memory << 16 | picks the guess did not settle << 24 | float64 picks << 28
__device__ unsigned int g_einfo[4096 * 16];
#define EINFO(slot, v) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && (slot) < 16) g_einfo[blockIdx.x * 16 + (slot)] = (v); } while (0)
#else
#define EINFO(slot, v) do {} while (0)
#define ESTAMP(i) do {} while (0)
#define ESTAMP_CYCLES(i) do {} while (0)
#endif

__device__ inline int sc_idx(int g, int f) { return g * 8 + f; }
enum { SC_TP = 0, SC_PASSES, SC_MOVES, SC_CAPB, SC_CAPW, SC_LAST, SC_HIST, SC_SEEN };

// ------------------------------------------------------------------------------------------------
// A game state being replayed inside one wave: board in LDS, scalars in registers (wave-uniform).
template <int PPL>
struct Sim {
    uint64_t hash;
    int tp, passes, moves, capb, capw, last, ns;
};

template <int PPL>
__device__ inline void sim_load_root(const Dev& d, int g, int lane, const Geo<PPL>& geo, BoardLds<PPL>& b,
                                     uint64_t* seen_lds, Sim<PPL>& s) {
#pragma unroll
    for (int k = 0; k < PPL; ++k)
        if (geo.p[k] >= 0) b.col[geo.p[k]] = d.board[(size_t)g * d.npad + geo.p[k]];
    s.tp = d.sc[sc_idx(g, SC_TP)];
    s.passes = d.sc[sc_idx(g, SC_PASSES)];
    s.moves = d.sc[sc_idx(g, SC_MOVES)];
    s.capb = d.sc[sc_idx(g, SC_CAPB)];
    s.capw = d.sc[sc_idx(g, SC_CAPW)];
    s.last = d.sc[sc_idx(g, SC_LAST)];
    s.ns = d.sc[sc_idx(g, SC_SEEN)];
    s.hash = d.hash[g];
    for (int i = lane; i < s.ns; i += 64) seen_lds[i] = d.seen[(size_t)g * d.seen_cap + i];
    wave_sync();
}

template <int PPL>
__device__ inline bool sim_terminal(const Dev& d, const Sim<PPL>& s) {
    return s.passes >= 2 || s.moves >= 2 * d.np;  // go.py:205-206
}

// go.py:159-203 on the replayed state. `hist_lds` (may be null) holds H boards newest first.
template <int PPL>
__device__ inline void sim_play(const Dev& d, int lane, const Geo<PPL>& geo, BoardLds<PPL>& b, uint64_t* seen_lds,
                                int8_t* hist_lds, Sim<PPL>& s, int action) {
    if (action == d.np) {
        s.passes += 1;
        s.last = -1;
    } else {
        analyze<PPL, false>(geo, b, d.zob);
        const int cap = place_stone<PPL>(geo, b, d.zob, d.n, lane, action, s.tp, s.hash);
        if (s.tp > 0) s.capb += cap; else s.capw += cap;
        s.passes = 0;
        s.last = action;
    }
    s.tp = -s.tp;
    s.moves += 1;
    if (lane == 0) seen_lds[s.ns] = s.hash ^ d.zob[kZobSide + (s.tp < 0)];
    s.ns += 1;
    if (hist_lds) {
        for (int i = d.H - 1; i >= 1; --i) {
#pragma unroll
            for (int k = 0; k < PPL; ++k)
                if (geo.p[k] >= 0) hist_lds[i * BoardLds<PPL>::NPAD + geo.p[k]] = hist_lds[(i - 1) * BoardLds<PPL>::NPAD + geo.p[k]];
        }
#pragma unroll
        for (int k = 0; k < PPL; ++k)
            if (geo.p[k] >= 0) hist_lds[geo.p[k]] = b.col[geo.p[k]];
    }
    wave_sync();
}

// go.py:251-263 for a terminal replayed state.
template <int PPL>
__device__ inline double sim_outcome(const Dev& d, const Geo<PPL>& geo, BoardLds<PPL>& b, const Sim<PPL>& s,
                                     double* black_out = nullptr, double* white_out = nullptr) {
    analyze<PPL, true>(geo, b, d.zob);
    int bs, ws;
    area_score<PPL>(geo, b, bs, ws);
    const double black = (double)bs, white = (double)ws + d.komi;
    if (black_out) { *black_out = black; *white_out = white; }
    const int w = black > white ? 1 : (white > black ? -1 : 0);
    return w == 0 ? 0.0 : (w == s.tp ? 1.0 : -1.0);
}

// ------------------------------------------------------------------------------------------------
// Root-state kernels
template <int PPL>
__global__ __launch_bounds__(64) void k_reset(Dev d, const uint8_t* mask) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (mask && !mask[g]) return;
    if (d.rc_cap && lane == 0) { d.root_cached[g] = 0; d.tree_live[g] = 0; }
    for (int i = lane; i < d.npad; i += 64) d.board[(size_t)g * d.npad + i] = 0;
    for (int i = lane; i < d.H * d.npad; i += 64) d.hist[(size_t)g * d.H * d.npad + i] = 0;
    if (lane == 0) {
        int* sc = d.sc + g * 8;
        sc[SC_TP] = 1; sc[SC_PASSES] = 0; sc[SC_MOVES] = 0; sc[SC_CAPB] = 0; sc[SC_CAPW] = 0;
        sc[SC_LAST] = -1; sc[SC_HIST] = 1; sc[SC_SEEN] = 1;
        d.hash[g] = 0;
        d.seen[(size_t)g * d.seen_cap] = d.zob[kZobSide + 0];  // empty board, Black to move
    }
}

template <int PPL>
__global__ __launch_bounds__(64) void k_play(Dev d, const int* actions, int* status) {
    __shared__ BoardLds<PPL> b;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint64_t* seen_lds = (uint64_t*)dyn;
    const int g = blockIdx.x, lane = threadIdx.x;
    const int a = actions[g];
    if (a < 0) { if (status && lane == 0) status[g] = 0; return; }
    Geo<PPL> geo;
    geo_init<PPL>(geo, d.n, lane);
    Sim<PPL> s;
    sim_load_root<PPL>(d, g, lane, geo, b, seen_lds, s);
    bool ok = a <= d.np && s.ns < d.seen_cap;
    if (ok && a < d.np) {  // go.py:179-180: re-check legality
        analyze<PPL, false>(geo, b, d.zob);
        bool legal[PPL];
        legal_points<PPL>(geo, b, d.zob, s.tp, s.hash, seen_lds, s.ns, legal);
        bool mine = false;
#pragma unroll
        for (int k = 0; k < PPL; ++k)
            if (geo.p[k] == a) mine = legal[k];
        ok = __any(mine);
    }
    if (!ok) { if (status && lane == 0) status[g] = AZ_ERR_ILLEGAL; return; }
    const int ns0 = s.ns;
    sim_play<PPL>(d, lane, geo, b, seen_lds, nullptr, s, a);
    // write back
    int8_t* hist = d.hist + (size_t)g * d.H * d.npad;
    for (int i = d.H - 1; i >= 1; --i)
#pragma unroll
        for (int k = 0; k < PPL; ++k)
            if (geo.p[k] >= 0) hist[i * d.npad + geo.p[k]] = hist[(i - 1) * d.npad + geo.p[k]];
#pragma unroll
    for (int k = 0; k < PPL; ++k)
        if (geo.p[k] >= 0) {
            const int8_t v = b.col[geo.p[k]];
            d.board[(size_t)g * d.npad + geo.p[k]] = v;
            hist[geo.p[k]] = v;
        }
    if (lane == 0) {
        int* sc = d.sc + g * 8;
        // the seen list stands for a set (go.py:173/:199): skip keys already present
        const uint64_t key = seen_lds[ns0];
        bool dup = false;
        for (int i = 0; i < ns0; ++i) dup |= seen_lds[i] == key;
        if (!dup) { d.seen[(size_t)g * d.seen_cap + ns0] = key; sc[SC_SEEN] = ns0 + 1; }
        sc[SC_TP] = s.tp; sc[SC_PASSES] = s.passes; sc[SC_MOVES] = s.moves; sc[SC_CAPB] = s.capb;
        sc[SC_CAPW] = s.capw; sc[SC_LAST] = s.last;
        sc[SC_HIST] = min(sc[SC_HIST] + 1, d.H);
        d.hash[g] = s.hash;
        if (status) status[g] = 0;
    }
    if (d.rc_cap) {   // wave-uniform
        const size_t nb = (size_t)g * d.max_nodes;
        int ci = -1;
        if (d.tree_live[g]) {
            const int first = d.child_off[nb], nc = min((int)d.n_child[nb], d.rc_cap);
            for (int c0 = 0; c0 < nc; c0 += 64) {
                const int c = c0 + lane;
                const bool hit = c < nc && d.action[nb + first + c] == a && d.rc_valid[(size_t)g * d.rc_cap + c];
                const unsigned long long m = __ballot(hit);
                if (m) { ci = c0 + __ffsll((long long)m) - 1; break; }
            }
        }
        if (ci >= 0) {
            const float* src = d.rc_logits + ((size_t)g * d.rc_cap + ci) * d.A;
            for (int i = lane; i < d.A; i += 64) d.root_logits[(size_t)g * d.A + i] = src[i];
        }
        __syncthreads();
        if (lane == 0) { d.root_cached[g] = ci >= 0 ? 1 : 0; d.tree_live[g] = 0; }
    }
}

// Feature planes + legal mask of one replayed state (go.py:265-304). mask_out may be null.
template <int PPL>
__device__ inline int emit_features(const Dev& d, int lane, const Geo<PPL>& geo, BoardLds<PPL>& b,
                                    const uint64_t* seen_lds, const int8_t* hist, int hist_stride, int hist_count,
                                    const Sim<PPL>& s, void* out_row, int layout, uint8_t* mask_out) {
    analyze<PPL, false>(geo, b, d.zob);
    bool legal[PPL];
    legal_points<PPL>(geo, b, d.zob, s.tp, s.hash, seen_lds, s.ns, legal);
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < PPL; ++k) cnt += legal[k] ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (mask_out) {
#pragma unroll
        for (int k = 0; k < PPL; ++k)
            if (geo.p[k] >= 0) mask_out[geo.p[k]] = legal[k] ? 1 : 0;
        if (lane == 0) mask_out[d.np] = 1;
    }
    if (out_row) {
        const double max_pts = (double)(d.np > 1 ? d.np : 1);
        const int own = s.tp > 0 ? s.capb : s.capw, opp = s.tp > 0 ? s.capw : s.capb;
        const float f_own = (float)((double)own / max_pts), f_opp = (float)((double)opp / max_pts);
        const float f_tp = s.tp > 0 ? 1.0f : 0.0f;
        const int take = hist_count < d.H ? hist_count : d.H;
        if (layout == AZ_FEAT_F32_NCHW) {
            float* o = (float*)out_row;
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                const int p = geo.p[k];
                if (p < 0) continue;
                for (int i = 0; i < d.H; ++i) {
                    const int v = i < take ? hist[i * hist_stride + p] : 0;
                    o[i * d.np + p] = (v != 0 && v == s.tp) ? 1.0f : 0.0f;
                    o[(d.H + i) * d.np + p] = (v != 0 && v == -s.tp) ? 1.0f : 0.0f;
                }
                o[(2 * d.H) * d.np + p] = f_tp;
                o[(2 * d.H + 1) * d.np + p] = legal[k] ? 1.0f : 0.0f;
                o[(2 * d.H + 2) * d.np + p] = p == s.last ? 1.0f : 0.0f;
                o[(2 * d.H + 3) * d.np + p] = f_own;
                o[(2 * d.H + 4) * d.np + p] = f_opp;
            }
        } else {  // bf16 [np][cpad], channel-minor: a point's channels are assembled in registers, 16 bytes per store
            const int cpad = d.C <= 32 ? 32 : 64;
            const uint32_t one = 0x3F80u;
            auto bf = [](float f) -> uint32_t {  // round to nearest even
                uint32_t u = __float_as_uint(f);
                return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
            };
            const uint32_t w_own = bf(f_own), w_opp = bf(f_opp), w_tp = s.tp > 0 ? one : 0u;
            const int H = d.H;
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                const int p = geo.p[k];
                if (p < 0) continue;
                const uint32_t w_legal = legal[k] ? one : 0u, w_last = p == s.last ? one : 0u;
                auto chan = [&](int c) -> uint32_t {   // plane c of point p (go.py:270-302)
                    if (c < 2 * H) {
                        const int i = c < H ? c : c - H, want = c < H ? s.tp : -s.tp;
                        const int v = i < take ? hist[i * hist_stride + p] : 0;
                        return v == want ? one : 0u;   // want is +-1, so v != 0 is implied
                    }
                    c -= 2 * H;
                    return c == 0 ? w_tp : c == 1 ? w_legal : c == 2 ? w_last : c == 3 ? w_own : c == 4 ? w_opp : 0u;
                };
                uint4* op = (uint4*)((uint16_t*)out_row + (size_t)p * cpad);
                for (int q = 0; q < cpad / 8; ++q) {
                    const int c0 = 8 * q;
                    uint4 v;
                    v.x = chan(c0) | (chan(c0 + 1) << 16);
                    v.y = chan(c0 + 2) | (chan(c0 + 3) << 16);
                    v.z = chan(c0 + 4) | (chan(c0 + 5) << 16);
                    v.w = chan(c0 + 6) | (chan(c0 + 7) << 16);
                    op[q] = v;
                }
            }
        }
    }
    return cnt + 1;  // + pass
}

template <int PPL>
__global__ __launch_bounds__(64) void k_root_features(Dev d, void* out, int layout, uint8_t* mask_out) {
    __shared__ BoardLds<PPL> b;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint64_t* seen_lds = (uint64_t*)dyn;
    const int g = blockIdx.x, lane = threadIdx.x;
    Geo<PPL> geo;
    geo_init<PPL>(geo, d.n, lane);
    Sim<PPL> s;
    sim_load_root<PPL>(d, g, lane, geo, b, seen_lds, s);
    const size_t row_bytes = layout == AZ_FEAT_F32_NCHW ? (size_t)d.C * d.np * 4 : (size_t)d.np * (d.C <= 32 ? 32 : 64) * 2;
    emit_features<PPL>(d, lane, geo, b, seen_lds, d.hist + (size_t)g * d.H * d.npad, d.npad, d.sc[sc_idx(g, SC_HIST)], s,
                       out ? (unsigned char*)out + (size_t)g * row_bytes : nullptr, layout,
                       mask_out ? mask_out + (size_t)g * d.A : nullptr);
}

template <int PPL>
__global__ __launch_bounds__(64) void k_status(Dev d, double* score, int* info) {
    __shared__ BoardLds<PPL> b;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint64_t* seen_lds = (uint64_t*)dyn;
    const int g = blockIdx.x, lane = threadIdx.x;
    Geo<PPL> geo;
    geo_init<PPL>(geo, d.n, lane);
    Sim<PPL> s;
    sim_load_root<PPL>(d, g, lane, geo, b, seen_lds, s);
    double bl, wh;
    sim_outcome<PPL>(d, geo, b, s, &bl, &wh);
    if (lane == 0) {
        if (score) { score[2 * g] = bl; score[2 * g + 1] = wh; }
        if (info) {
            int* o = info + g * 8;
            o[0] = s.tp; o[1] = s.passes; o[2] = s.moves; o[3] = s.capb; o[4] = s.capw; o[5] = s.last;
            o[6] = sim_terminal<PPL>(d, s) ? 1 : 0;
            o[7] = bl > wh ? 1 : (wh > bl ? -1 : 0);
        }
    }
}

__global__ void k_export(Dev d, int8_t* board, int8_t* hist, int* scalars, uint64_t* seen) {
    const int g = blockIdx.x;
    for (int i = threadIdx.x; i < d.np; i += blockDim.x) board[(size_t)g * d.np + i] = d.board[(size_t)g * d.npad + i];
    for (int i = threadIdx.x; i < d.H * d.np; i += blockDim.x)
        hist[(size_t)g * d.H * d.np + i] = d.hist[((size_t)g * d.H + i / d.np) * d.npad + i % d.np];
    for (int i = threadIdx.x; i < 8; i += blockDim.x) scalars[g * 8 + i] = d.sc[g * 8 + i];
    const int ns = d.sc[sc_idx(g, SC_SEEN)];
    for (int i = threadIdx.x; i < d.seen_cap; i += blockDim.x)
        seen[(size_t)g * d.seen_cap + i] = i < ns ? d.seen[(size_t)g * d.seen_cap + i] : 0;
}

__global__ void k_import(Dev d, const int8_t* board, const int8_t* hist, const int* scalars, const uint64_t* seen) {
    const int g = blockIdx.x;
    __shared__ unsigned long long hsh;
    if (threadIdx.x == 0) hsh = 0;
    __syncthreads();
    unsigned long long h = 0;
    for (int i = threadIdx.x; i < d.npad; i += blockDim.x) {
        const int8_t v = i < d.np ? board[(size_t)g * d.np + i] : 0;
        d.board[(size_t)g * d.npad + i] = v;
        if (v != 0) h ^= d.zob[2 * i + (v < 0)];
    }
    atomicXor(&hsh, h);
    for (int i = threadIdx.x; i < d.H * d.npad; i += blockDim.x) {
        const int hi = i / d.npad, p = i % d.npad;
        d.hist[(size_t)g * d.H * d.npad + i] = p < d.np ? hist[((size_t)g * d.H + hi) * d.np + p] : 0;
    }
    for (int i = threadIdx.x; i < 8; i += blockDim.x) d.sc[g * 8 + i] = scalars[g * 8 + i];
    for (int i = threadIdx.x; i < d.seen_cap; i += blockDim.x) d.seen[(size_t)g * d.seen_cap + i] = seen[(size_t)g * d.seen_cap + i];
    __syncthreads();
    if (threadIdx.x == 0) {
        d.hash[g] = hsh;
        if (d.rc_cap) { d.root_cached[g] = 0; d.tree_live[g] = 0; }
    }
}

// ------------------------------------------------------------------------------------------------
// Search kernels
__global__ __launch_bounds__(64) void k_begin(Dev d, const uint8_t* enable) {
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g == 0) d.ctl[1] = 0;
    if (g >= d.G) return;
    for (int i = 0; i < 8; ++i) d.gstats[(size_t)g * 8 + i] = 0;
    const int* sc = d.sc + g * 8;
    const bool terminal = sc[SC_PASSES] >= 2 || sc[SC_MOVES] >= 2 * d.np;
    const bool act = !terminal && (!enable || enable[g]);
    d.active[g] = act ? 1 : 0;
    const size_t base = (size_t)g * d.max_nodes;
    d.prior[base] = 1.0; d.vsum[base] = 0.0; d.visits[base] = 0; d.action[base] = -1;  // mcts.py:171
    d.child_off[base] = 0; d.n_child[base] = 0;
    d.n_nodes[g] = 1;
    d.n_unique[g] = act ? 1 : 0;  // the root evaluation is the first pending row
    if (d.rc_cap) {
        if (act) {
            for (int i = 0; i < d.rc_cap; ++i) d.rc_valid[(size_t)g * d.rc_cap + i] = 0;
            d.tree_live[g] = 1;
        }
    }
}

// Exclusive scan of n_unique over games -> row offsets, row->(game, slot) tables, row count. One workgroup of 16 waves:
// a thread sums its share of consecutive games, a wave scans its 64 sums by lane shifts, the 16 wave totals are scanned by
// every thread from LDS (two barriers in all).
__device__ inline int block_exclusive_scan_1024(int v, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    __syncthreads();                 // a previous use of wave_tot has been read
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(1024) void k_scan_rows(Dev d, int root_phase) {
    __shared__ int wave_tot[16];
    const int t = threadIdx.x;
    const int per = (d.G + 1023) / 1024;
    const int lo = min(t * per, d.G), hi = min(lo + per, d.G);
    const bool split = root_phase && d.rc_cap;   // rows of roots whose logits were carried over go last
    int sum = 0, late = 0;
    for (int g = lo; g < hi; ++g) {
        if (split && d.root_cached[g]) late += d.n_unique[g];
        else sum += d.n_unique[g];
    }
    int n_eval;
    int off = block_exclusive_scan_1024(sum, wave_tot, n_eval);
    for (int g = lo; g < hi; ++g) {
        if (split && d.root_cached[g]) continue;
        d.row_off[g] = off;
        const int nu = d.n_unique[g];
        for (int u = 0; u < nu; ++u) {
            d.row_game[off + u] = g;
            d.row_slot[off + u] = root_phase ? -1 : d.uniq_slot[g * d.B + u];
        }
        off += nu;
    }
    int n_all = n_eval;
    if (split) {
        int n_late;
        off = n_eval + block_exclusive_scan_1024(late, wave_tot, n_late);
        for (int g = lo; g < hi; ++g) {
            if (!d.root_cached[g]) continue;
            d.row_off[g] = off;
            for (int u = 0; u < d.n_unique[g]; ++u) { d.row_game[off + u] = g; d.row_slot[off + u] = -1; }
            off += d.n_unique[g];
        }
        n_all = n_eval + n_late;
    }
    if (t == 1023) {
        d.row_off[d.G] = n_all; d.ctl[0] = n_eval; d.ctl[2] = n_all;
        if (root_phase && d.ctl_mirror) { d.ctl_mirror[0] = n_eval; d.ctl_mirror[2] = n_all; }
    }
}

// (score, index) arg-max over the wave: largest score, smallest index among equals (idx must be unique per lane, or
// 0x7fffffff with score -inf). The maximum is reduced with DPP row shifts / row broadcasts (register-to-register, no LDS
// round trip per step as a lane shuffle has): after the four row_shr steps lane 15 of every row of 16 holds its row's
// maximum, the two row_bcast steps fold the rows into lane 63. The index follows from a ballot of the lanes that hold
// the maximum. NaN scores never win (they compare false), as in a sequential scan that starts at -inf.
__device__ inline double wave_max_f64(double v) {
    auto step = [&](auto ctrl, auto row_mask) {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const int ninf_hi = (int)0xFFF00000;   // lanes a step does not reach combine with -inf
        const int olo = __builtin_amdgcn_update_dpp(0, lo, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
        const int ohi = __builtin_amdgcn_update_dpp(ninf_hi, hi, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
        const double o = __hiloint2double(ohi, olo);
        if (o > v) v = o;
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});   // row_shr:1
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});   // row_shr:2
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});   // row_shr:4
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});   // row_shr:8
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1 and 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ inline double readlane_f64(double v, int lane_index) {   // lane_index must be wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane_index), __builtin_amdgcn_readlane(__double2loint(v), lane_index));
}
__device__ inline void wave_argmax(double& s, int& idx) {
    const double m = wave_max_f64(s);
    // every lane's candidate is its own index here (one child per lane and trip), or the lane's best over several trips:
    // among the lanes that hold the maximum the smallest index wins
    int cand = (s == m && idx != 0x7fffffff) ? idx : 0x7fffffff;
    cand = min(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x111, 0xf, 0xf, false));
    cand = min(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x112, 0xf, 0xf, false));
    cand = min(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x114, 0xf, 0xf, false));
    cand = min(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x118, 0xf, 0xf, false));
    cand = min(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x142, 0xa, 0xf, false));
    cand = min(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x143, 0xc, 0xf, false));
    s = m; idx = __builtin_amdgcn_readlane(cand, 63);
}

// Wave maximum of floats (no NaNs among them, or a NaN may win) through the order-preserving integer key of a float
// (sign-magnitude to two's complement), so that every step is one v_max_i32 with a DPP operand: lanes a step does not
// reach keep their value.
__device__ inline int float_key(float f) {
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ inline float key_float(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
__device__ inline float wave_max_f32(float f) {
    int v = float_key(f);
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return key_float(__builtin_amdgcn_readlane(v, 63));
}

// Actions (and root children) per lane in the search kernels of a board class: boards up to 11x11 have at most 122 actions
// (2 per lane), 12x12 and 13x13 at most 170 (3), boards up to 19x19 at most 362 (6).
template <int PPL> struct Apl { static constexpr int v = PPL <= 2 ? 2 : PPL; };

// The root's and its children's statistics in the wave's registers for a whole batch (child i: lane i & 63, slot i >> 6):
// every descent starts at the root and, with virtual loss, most end at one of its children.
template <int CPL>
struct RootCache {
    int rnc, rfirst, r_vis;
    double r_vsum;
    double prior[CPL], vsum[CPL];
    int vis[CPL], nch[CPL], act[CPL];
    int cf[CPL];   // child_off of the child: index of its first child, or the outcome code of a terminal one (term_code)
    __device__ bool held() const { return rnc > 0 && rnc <= 64 * CPL; }
};

template <int CPL>
__device__ inline void rc_load(const Dev& d, size_t nb, int lane, int rnc, int rfirst, RootCache<CPL>& c) {
    c.rnc = rnc; c.rfirst = rfirst;
    c.r_vis = d.visits[nb]; c.r_vsum = d.vsum[nb];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int i = lane + 64 * k;
        c.prior[k] = 0.0; c.vsum[k] = 0.0; c.vis[k] = 0; c.nch[k] = 0; c.act[k] = 0; c.cf[k] = 0;
        if (c.held() && i < rnc) {
            const size_t x = nb + rfirst + i;
            c.prior[k] = d.prior[x]; c.vsum[k] = d.vsum[x]; c.vis[k] = d.visits[x]; c.nch[k] = d.n_child[x]; c.act[k] = d.action[x];
            c.cf[k] = d.child_off[x];
        }
    }
}

template <int CPL>
__device__ inline void rc_store(const Dev& d, size_t nb, int lane, const RootCache<CPL>& c) {
    if (!c.held()) return;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int i = lane + 64 * k;
        if (i < c.rnc) { d.visits[nb + c.rfirst + i] = c.vis[k]; d.vsum[nb + c.rfirst + i] = c.vsum[k]; }
    }
    if (lane == 0) { d.visits[nb] = c.r_vis; d.vsum[nb] = c.r_vsum; }
}

// A terminal node never gets children, so its child_off field is free: once its outcome (go.py:251-263: +1, -1 or 0 from the
// side to move) has been computed by replaying the path, it is kept there as -2 - (outcome + 1), and later descents that
// end in the node read it instead of replaying.
__device__ inline int term_code(double outcome) { return -2 - ((int)outcome + 1); }
__device__ inline bool is_term_code(int child_off) { return child_off <= -2; }
__device__ inline double term_outcome(int child_off) { return (double)(-child_off - 3); }

// arr[idx >> 6] of lane idx & 63; idx must be wave-uniform
template <int CPL>
__device__ inline int rc_read(const int (&arr)[CPL], int idx) {
    int v = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
        if (k == (idx >> 6)) v = __builtin_amdgcn_readlane(arr[k], idx & 63);
    return v;
}

// The batch record a game's descents leave for the expansion that follows the network (int32 [4 + 4 B] per game, read
// back with one load per lane): [0] distinct leaves, [1] -, [2] children of the root, [3] index of its first child, then
// per slot s: path length | (slot whose row evaluates the leaf + 1) << 16 (0: terminal), the root child on the path (-1:
// the path ends at the root), the leaf node; then the slots of the distinct leaves in row order.
__device__ inline int brec_words(const Dev& d) { return 4 + 4 * d.B; }

// A child's number of children (10 bits) and action (10 bits) in one word.
__device__ inline int pack_child(int nch, int act) { return nch | (act << 10); }

// The children of one tree node in the wave's registers for the descents of a batch (child i: lane i & 63, slot i >> 6).
// With virtual loss the descents of a batch mostly walk the same path, so a level loaded by one descent serves the next
// ones: they pick among its children, and apply their virtual loss to the pick, without a memory access. The node's own
// statistics are an entry of the level above.
template <int CPL>
struct Level {
    int node, n, first;       // whose children (-1: none), how many, index of the first
    int pv0;                  // visit count of the node that lane 0's exploration-table entries belong to
    int guess;                // the child picked last (with virtual loss, usually picked again)
    double t_pbc, t_sqn;      // lane i: exploration-table entries for pv0 + i visits of the node (pv0 without virtual loss)
    float t_pbc32, t_sqn32;
    double prior[CPL], vsum[CPL];
    float pr32[CPL], vs32[CPL];
    int vis[CPL], pk[CPL], cf[CPL];   // cf: child_off of the child (first grandchild, or the outcome code of a terminal child)
};

// The wave-uniform fields of the levels below the root are kept in vector registers (COLD): scalar registers are short in
// these kernels, and a level's fields are only needed by a descent that reaches it.
__device__ __forceinline__ int to_vgpr(int x) { asm volatile("" : "+v"(x)); return x; }
template <bool COLD> __device__ __forceinline__ int cold_set(int x) { return COLD ? to_vgpr(x) : x; }
template <bool COLD> __device__ __forceinline__ int cold_get(int x) { return COLD ? __builtin_amdgcn_readfirstlane(x) : x; }

template <int CPL, bool COLD>
__device__ __forceinline__ void level_table(const Dev& d, int lane, Level<CPL>& lv, int pv) {
    const int t_idx = min(pv + (d.vl > 0.0 ? lane : 0), d.tab_len - 1);
    lv.pv0 = cold_set<COLD>(pv);
    lv.t_pbc = d.pbc[t_idx]; lv.t_sqn = d.sqn[t_idx];
    lv.t_pbc32 = (float)lv.t_pbc; lv.t_sqn32 = (float)lv.t_sqn;
}

template <int CPL>
__device__ __forceinline__ void level_flush(const Dev& d, size_t nb, int lane, Level<CPL>& lv) {
    if (cold_get<true>(lv.node) >= 0) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int i = lane + 64 * k;
            if (i < lv.n) { d.visits[nb + lv.first + i] = lv.vis[k]; d.vsum[nb + lv.first + i] = lv.vsum[k]; }
        }
    }
}

// pv: the node's visit count now.
template <int CPL>
__device__ __forceinline__ void level_load(const Dev& d, size_t nb, int lane, int node, int first, int n, int pv, Level<CPL>& lv) {
    lv.node = to_vgpr(node); lv.first = to_vgpr(first); lv.n = to_vgpr(n); lv.guess = to_vgpr(-1);
    level_table<CPL, true>(d, lane, lv, pv);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int i = lane + 64 * k;
        lv.prior[k] = 0.0; lv.vsum[k] = 0.0; lv.vis[k] = 0; lv.pk[k] = 0; lv.cf[k] = 0;
        if (i < n) {
            const size_t x = nb + first + i;
            lv.prior[k] = d.prior[x]; lv.vsum[k] = d.vsum[x]; lv.vis[k] = d.visits[x]; lv.cf[k] = d.child_off[x];
            lv.pk[k] = pack_child(d.n_child[x], d.action[x]);
        }
        lv.pr32[k] = (float)lv.prior[k]; lv.vs32[k] = (float)lv.vsum[k];
    }
}

// arr[idx >> 6] of lane idx & 63; idx must be wave-uniform
template <int CPL>
__device__ __forceinline__ int lane_pick(const int (&arr)[CPL], int lane, int idx) {
    int v = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) v = lane + 64 * k == idx ? arr[k] : v;
    return __builtin_amdgcn_readlane(v, idx & 63);
}

// mcts.py:97-116 over the children held in `lv`; pv: the node's visit count now. Returns the child (wave-uniform) or -1.
// Pre-selection in float with error intervals: a child's float score sc and a bound e on its distance from the float64
// score of mcts.py:104-108 (at most eleven roundings of 2^-23 relative each on the exploration and the value term: below
// 1.4e-6 (|u| + |q|); e takes 4e-6). Only children with sc + e >= max(sc - e) can hold the float64 maximum; when exactly one
// does, it is the child the float64 arg-max returns, and the float64 scores are not needed. Before the wave-wide maximum
// the child picked last is tried alone against the others' upper bounds. Ties and near-ties take the float64 arg-max.
template <int CPL, bool COLD>
__device__ __forceinline__ int level_pick(const Dev& d, int lane, Level<CPL>& lv, int pv, int& dbg) {
    int tslot = cold_get<COLD>(pv - lv.pv0);
    const int guess = cold_get<COLD>(lv.guess);
    if (d.vl > 0.0 ? (tslot < 0 || tslot >= 64) : tslot != 0) { level_table<CPL, COLD>(d, lane, lv, pv); tslot = 0; }
    {
        const float pbc32 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv.t_pbc32), tslot));
        const float sq32 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv.t_sqn32), tslot));
        float hi[CPL], lo_k[CPL], lo = -INFINITY;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const bool valid = lane + 64 * k < lv.n;
            const int vc = lv.vis[k];
            const float fvc = (float)vc;
            const float u = pbc32 * (sq32 * __builtin_amdgcn_rcpf(fvc + 1.0f)) * lv.pr32[k];
            const float q = vc == 0 ? 0.0f : lv.vs32[k] * __builtin_amdgcn_rcpf(fvc);
            const float sc = u - q, e = 4e-6f * (fabsf(u) + fabsf(q)) + 1e-30f;
            hi[k] = valid ? sc + e : -INFINITY;
            lo_k[k] = valid ? sc - e : -INFINITY;
            lo = fmaxf(lo, lo_k[k]);
        }
        float m_lo = -INFINITY;
        int total = 0, ci = 0;
        if (guess >= 0) {
            float lo_g = -INFINITY;
#pragma unroll
            for (int k = 0; k < CPL; ++k) lo_g = lane + 64 * k == guess ? lo_k[k] : lo_g;
            m_lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo_g), guess & 63));
#pragma unroll
            for (int k = 0; k < CPL; ++k) total += __popcll(__ballot(hi[k] >= m_lo));
            ci = guess;
        }
        if (total != 1) {
            dbg += 1;            // the guess did not settle it
            m_lo = wave_max_f32(lo);
            total = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const unsigned long long bal = __ballot(hi[k] >= m_lo);
                total += __popcll(bal);
                if (bal) ci = 64 * k + __ffsll((long long)bal) - 1;
            }
        }
        if (total == 1 && m_lo > -1e17f) {   // (scores at or below -1e18 never win: mcts.py:100, :109)
            lv.guess = cold_set<COLD>(ci);
            return ci;
        }
    }
    dbg += 16;               // float64 scores needed
    const double pbc = readlane_f64(lv.t_pbc, tslot), sq = readlane_f64(lv.t_sqn, tslot);
    double best = -INFINITY;
    int best_i = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int i = lane + 64 * k;
        if (i < lv.n) {
            int vc = lv.vis[k];
            asm volatile("" : "+v"(vc));   // keeps these divisions here: hoisted, they would run in front of every pick
            const double pb = pbc * (sq / ((double)vc + 1.0));
            const double prior_score = pb * lv.prior[k];
            const double value_score = -(vc == 0 ? 0.0 : lv.vsum[k] / (double)vc);
            const double score = prior_score + value_score;
            if (score > -1e18 && (score > best)) { best = score; best_i = i; }
        }
    }
    wave_argmax(best, best_i);
    if (best_i == 0x7fffffff) return -1;   // mcts.py:114-115
    lv.guess = cold_set<COLD>(best_i);
    return best_i;
}

// Levels below the root that a wave keeps in registers (the root's children always are): leaves down to depth
// DeepLevels + 1 are reached without memory accesses once their path has been walked in the batch. The "wide" kernels keep
// none and fit four waves on a SIMD (128 registers): with more games than the chip holds waves of the regular kernels
// (two per SIMD), what counts is how many waves run at once, not how long one takes. Same results either way.
#ifdef AZ_DEEP_LEVELS   // diagnostic builds
template <int CPL> struct DeepLevels { static constexpr int v = AZ_DEEP_LEVELS; };
#else
template <int CPL> struct DeepLevels { static constexpr int v = CPL <= 2 ? 3 : (CPL == 3 ? 2 : 1); };
#endif

// mcts.py:181-199: batch_n descents for game blockIdx.x. `rc` holds the root's statistics (rc_load) and is left updated,
// not stored. rows_ctr: null (the row scan kernel assigns the rows), or the counter this batch's rows are allotted from.
template <int PPL, int DC>
__device__ __forceinline__ void select_body(const Dev& d, int batch_n, RootCache<Apl<PPL>::v>& rc, int root_passes, int root_moves,
                                            double t_pbc, double t_sqn, int t_base, int* rows_ctr) {
    constexpr int CPL = Apl<PPL>::v;
    __shared__ BoardLds<PPL> b;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint64_t* seen_lds = (uint64_t*)dyn;
    double* tc_val = (double*)(seen_lds + d.seen_cap);   // outcomes of the terminal leaves met in this batch ...
    int* path_lds = (int*)(tc_val + d.B);            // the current descent's nodes
    int* uniq_lds = path_lds + d.max_depth;          // (slot, leaf node) of the batch's distinct leaves so far
    // Per-descent results wait in registers / LDS and go to global memory after the last descent: stores and loads share
    // one counter, so a store in flight would stall the next descent at its first wait for a load (about 2 us each).
    constexpr int PSTAGE = 8;                        // path nodes staged per descent; a longer path is written at once
    int* st_len = uniq_lds + 2 * d.B;                // [B]
    int* st_ref = st_len + d.B;                      // [B]
    int* st_n1 = st_ref + d.B;                       // [B] the root child on the path
    int* st_leaf = st_n1 + d.B;                      // [B] the leaf node
    int* st_path = st_leaf + d.B;                    // [B][PSTAGE]
    int* tc_node = st_path + d.B * PSTAGE;           // ... and their nodes: with virtual loss the descents of a batch mostly end
    int n_tc = 0;                                    // in the same leaf, and a terminal one is replayed and scored once
    const int g = blockIdx.x, lane = threadIdx.x;

    const size_t nb = (size_t)g * d.max_nodes;
    double* prior = d.prior + nb; double* vsum = d.vsum + nb; int* visits = d.visits + nb;
    const int16_t* action = d.action + nb; const int* child_off = d.child_off + nb; const int16_t* n_child = d.n_child + nb;
    int n_unique = 0, depth_sum = 0, max_depth = 0, n_term = 0;

    // The descent itself needs no board: children only exist for legal moves, and whether a node is terminal
    // follows from the pass count and the move count (go.py:205-206). The moves are replayed on an LDS board
    // only for a terminal leaf whose outcome is not known yet (the leaf gather replays the others).
    const int rnc = rc.rnc, rfirst = rc.rfirst;
    const bool cached = rc.held();
    // The root's children as level 0. Lane i holds the exploration-table entries of t_base + i visits, fetched before the
    // root's statistics were known; t_base is the visit count the caller expects (the simulations finished so far).
    Level<CPL> lv[DC + 1];
    {
        Level<CPL>& L0 = lv[0];
        L0.node = 0; L0.n = cached ? rnc : 0; L0.first = rfirst; L0.pv0 = t_base; L0.guess = -1;
        L0.t_pbc = t_pbc; L0.t_sqn = t_sqn; L0.t_pbc32 = (float)t_pbc; L0.t_sqn32 = (float)t_sqn;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            L0.prior[k] = rc.prior[k]; L0.vsum[k] = rc.vsum[k]; L0.vis[k] = rc.vis[k]; L0.cf[k] = rc.cf[k];
            L0.pk[k] = pack_child(rc.nch[k], rc.act[k]);
            L0.pr32[k] = (float)rc.prior[k]; L0.vs32[k] = (float)rc.vsum[k];
        }
    }
#pragma unroll
    for (int L = 1; L <= DC; ++L) { lv[L].node = to_vgpr(-1); lv[L].n = to_vgpr(0); lv[L].first = to_vgpr(0); lv[L].guess = to_vgpr(-1); }

    int last_node = -1, last_ref = -1;   // the distinct leaf found last (most descents of a batch end in it again)
    // Per-descent results: lane s keeps those of slot s in registers (slots from 64 on: LDS) until they are written out
    int my_len = 0, my_ref = 0, my_n1 = 0, my_leaf = 0;
    auto put_record = [&](int slot, int len, int ref, int n1, int leaf) {   // slot is wave-uniform
        if (slot < 64) {
            const bool me = lane == slot;
            my_len = me ? len : my_len; my_ref = me ? ref : my_ref; my_n1 = me ? n1 : my_n1; my_leaf = me ? leaf : my_leaf;
        } else if (lane == 0) {
            st_len[slot] = len; st_ref[slot] = ref; st_n1[slot] = n1; st_leaf[slot] = leaf;
        }
    };
#ifdef AZ_ESTAMP_FINE   // stamps 12-15: inside the fourth descent
#define FINE(i) do { if (slot == 3) ESTAMP_CYCLES(i); } while (0)
#else
#define FINE(i) do {} while (0)
#endif
    const bool root_open = cached && rnc > 0 && root_passes < 2 && root_moves < 2 * d.np && d.max_depth > 1;
    int slot = 0;
    while (slot < batch_n) {
        // The usual descent ends in a child of the root that has no children: a loop of its own for those, which holds
        // nothing of the deeper levels' state. It leaves the loop with the root's pick made when a descent goes on.
        int ci0 = -1, word0 = 0, cf0 = 0;
        bool failed = false;
        int dbg_loads = 0, dbg_replay = 0, dbg_pick = 0;   // diagnostic builds only
        (void)dbg_loads; (void)dbg_replay;
        for (; slot < batch_n; ++slot) {
#ifdef AZ_ESTAMP_FINE
            if (slot < 4) ESTAMP_CYCLES(8 + slot);
#else
            if (slot < 8) ESTAMP_CYCLES(8 + slot);
#endif
            ci0 = -1;
            dbg_pick = 0;
            if (!root_open) break;
            Level<CPL>& lvl = lv[0];
            const int ci = level_pick<CPL, false>(d, lane, lvl, rc.r_vis, dbg_pick);
            if (ci < 0) { failed = true; break; }
            ci0 = ci;
            word0 = lane_pick<CPL>(lvl.pk, lane, ci);
            cf0 = lane_pick<CPL>(lvl.cf, lane, ci);
            if ((word0 & 1023) != 0) break;
            const int node = rfirst + ci;
            const int passes1 = ((word0 >> 10) & 1023) == d.np ? root_passes + 1 : 0;
            const bool terminal1 = passes1 >= 2 || root_moves + 1 >= 2 * d.np;
            if (terminal1 && !is_term_code(cf0)) break;   // its outcome needs a replay
            int ref = slot;
            if (terminal1) {   // mcts.py:191-192, outcome known from an earlier descent
                ref = -1;
                n_term += 1;
                if (lane == 0) d.term_value[g * d.B + slot] = term_outcome(cf0);
            } else {
                if (d.dedup) {
                    if (node == last_node) ref = last_ref;
                    else
                        for (int u = 0; u < n_unique; ++u)
                            if (uniq_lds[2 * u + 1] == node) { ref = uniq_lds[2 * u]; break; }
                }
                if (ref == slot) {
                    if (lane == 0) { uniq_lds[2 * n_unique] = slot; uniq_lds[2 * n_unique + 1] = node; }
                    n_unique += 1;
                }
                last_node = node; last_ref = ref;
            }
            put_record(slot, 2, ref, node, node);
            EINFO(slot, (unsigned)(2 | (terminal1 ? 256 : 0) | (dbg_pick << 24)));
            depth_sum += 1;
            max_depth = max(max_depth, 1);
            if (d.vl > 0.0) {  // mcts.py:196-199
                rc.r_vis += 1;
                rc.r_vsum -= d.vl;
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    if (lane + 64 * k == ci) { lvl.vis[k] += 1; lvl.vsum[k] -= d.vl; lvl.vs32[k] = (float)lvl.vsum[k]; }
            }
            wave_sync();
        }
        if (slot >= batch_n) break;
        // ---- the descent of `slot` goes deeper (or ends in a way the loop above does not handle)
        int* path = d.path + ((size_t)g * d.B + slot) * d.max_depth;
        int node = 0, len = 1, nc = rnc, first = rfirst;
        int pv = cached ? rc.r_vis : -1;     // the current node's visit count (-1: in memory)
        int node_cf = 0;                     // child_off of the current node when it was picked from a register level
        int passes = root_passes, moves = root_moves;
        int pick[DC + 1];                    // the child picked at each register level (-1: level not reached)
        bool walking = ci0 >= 0;
        if (lane == 0) path_lds[0] = 0;
        pick[0] = ci0;
        if (ci0 >= 0) {
            node_cf = cf0;
            pv = lane_pick<CPL>(lv[0].vis, lane, ci0);
            node = rfirst + ci0;
            nc = word0 & 1023;
            first = cf0;
            passes = ((word0 >> 10) & 1023) == d.np ? passes + 1 : 0;
            moves += 1;
            if (lane == 0) path_lds[1] = node;
            len = 2;
        }
#pragma unroll
        for (int L = 1; L <= DC; ++L) {
            pick[L] = -1;
            if (!walking) continue;
            if (nc == 0 || passes >= 2 || moves >= 2 * d.np || len >= d.max_depth) { walking = false; continue; }
            Level<CPL>& lvl = lv[L];
            if (cold_get<true>(lvl.node) != node) {
                level_flush<CPL>(d, nb, lane, lvl);
                level_load<CPL>(d, nb, lane, node, first, nc, pv, lvl);
                dbg_loads += 1;
            }
            const int ci = level_pick<CPL, true>(d, lane, lvl, pv, dbg_pick);
            if (ci < 0) { failed = true; walking = false; continue; }
            pick[L] = ci;
            const int word = lane_pick<CPL>(lvl.pk, lane, ci);
            node_cf = lane_pick<CPL>(lvl.cf, lane, ci);
            pv = lane_pick<CPL>(lvl.vis, lane, ci);
            node = cold_get<true>(lvl.first) + ci;
            nc = word & 1023;
            first = node_cf;
            const int act = (word >> 10) & 1023;
            passes = act == d.np ? passes + 1 : 0;
            moves += 1;
            if (lane == 0) path_lds[len] = node;
            len += 1;
        }
        FINE(12);
        const int reg_len = len;             // path nodes whose statistics are in registers
        if (!failed && (walking || !cached)) {
            // below the register levels (or a root whose children they cannot hold): children read from memory
            while (true) {
                if (nc == 0 || passes >= 2 || moves >= 2 * d.np || len >= d.max_depth) break;
                int pvv = pv >= 0 ? pv : visits[node];
                pvv = pvv < d.tab_len ? pvv : d.tab_len - 1;
                const double pbc = d.pbc[pvv], sq = d.sqn[pvv];
                double best = -INFINITY;
                int best_i = 0x7fffffff;
                for (int i0 = 0; i0 < nc; i0 += 64) {
                    const int i = i0 + lane;
                    if (i < nc) {
                        const int vc = visits[first + i];
                        double pb = pbc * (sq / ((double)vc + 1.0));
                        const double prior_score = pb * prior[first + i];
                        const double value_score = -(vc == 0 ? 0.0 : vsum[first + i] / (double)vc);
                        const double score = prior_score + value_score;
                        if (score > -1e18 && (score > best)) { best = score; best_i = i; }
                    }
                }
                wave_argmax(best, best_i);
                if (best_i == 0x7fffffff) { failed = true; break; }  // mcts.py:114-115
                node = first + best_i;   // best_i is wave-uniform (wave_argmax returns it from one lane)
                const int act = (int)action[node];
                nc = (int)n_child[node];
                node_cf = child_off[node];
                first = node_cf;
                pv = -1;
                passes = act == d.np ? passes + 1 : 0;
                moves += 1;
                if (lane == 0) path_lds[len] = node;
                len += 1;
            }
        }
        if (failed) { if (lane == 0) atomicMin(&d.ctl[1], (int)AZ_ERR_NOCHILD); }
        wave_sync();
        if (len == 2) {}   // root -> child: written out from the record
        else if (len <= PSTAGE) { if (lane < len) st_path[slot * PSTAGE + lane] = path_lds[lane]; }
        else for (int i = lane; i < len; i += 64) path[i] = path_lds[i];
        const bool terminal = passes >= 2 || moves >= 2 * d.np;
        const int gs = g * d.B + slot;
        int ref;
        if (terminal) {  // mcts.py:191-192
            double v;
            if (len > 1 && is_term_code(node_cf)) {   // scored by an earlier descent (of this batch or another)
                v = term_outcome(node_cf);
            } else {
                int hit = -1;
                for (int u = 0; u < n_tc; ++u)
                    if (tc_node[u] == node) { hit = u; break; }
                if (hit >= 0) {
                    v = tc_val[hit];
                } else {
                    Geo<PPL> geo;
                    geo_init<PPL>(geo, d.n, lane);
                    Sim<PPL> s;
                    __syncthreads();
                    sim_load_root<PPL>(d, g, lane, geo, b, seen_lds, s);
                    for (int i = 1; i < len; ++i) sim_play<PPL>(d, lane, geo, b, seen_lds, nullptr, s, (int)action[path_lds[i]]);
                    v = sim_outcome<PPL>(d, geo, b, s);
                    if (lane == 0) { tc_node[n_tc] = node; tc_val[n_tc] = v; if (len > 1) d.child_off[nb + node] = term_code(v); }
                    n_tc += 1;
                    dbg_replay = 1;
                    // the level that holds the node learns the outcome too
#pragma unroll
                    for (int L = 0; L <= DC; ++L) {
                        if (L + 2 == len && pick[L] >= 0) {
#pragma unroll
                            for (int k = 0; k < CPL; ++k)
                                if (lane + 64 * k == pick[L]) lv[L].cf[k] = term_code(v);
                        }
                    }
                }
            }
            if (lane == 0) d.term_value[gs] = v;
            ref = -1;
            n_term += 1;
        } else {
            ref = slot;
            if (d.dedup) {
                if (node == last_node) ref = last_ref;
                else
                    for (int u = 0; u < n_unique; ++u)
                        if (uniq_lds[2 * u + 1] == node) { ref = uniq_lds[2 * u]; break; }
            }
            if (ref == slot) {
                if (lane == 0) {
                    uniq_lds[2 * n_unique] = slot;
                    uniq_lds[2 * n_unique + 1] = node;
                }
                n_unique += 1;
            }
            last_node = node; last_ref = ref;
        }
        FINE(13);
        EINFO(slot, (unsigned)(len | (terminal ? 256 : 0) | (dbg_replay << 9) | (dbg_loads << 12) | ((len - reg_len) << 16) | (dbg_pick << 24)));
        const int n1 = len > 1 ? (pick[0] >= 0 ? rfirst + pick[0] : path_lds[1]) : -1;   // the root child on the path
        put_record(slot, len, ref, n1, node);
        depth_sum += len - 1;
        max_depth = max(max_depth, len - 1);
        wave_sync();
        FINE(14);
        if (d.vl > 0.0) {  // mcts.py:196-199
            if (cached) {
                rc.r_vis += 1;
                rc.r_vsum -= d.vl;
#pragma unroll
                for (int L = 0; L <= DC; ++L) {
                    if (pick[L] >= 0) {
                        Level<CPL>& lvl = lv[L];
#pragma unroll
                        for (int k = 0; k < CPL; ++k)
                            if (lane + 64 * k == pick[L]) { lvl.vis[k] += 1; lvl.vsum[k] -= d.vl; lvl.vs32[k] = (float)lvl.vsum[k]; }
                    }
                }
            }
            for (int i = (cached ? reg_len : 0) + lane; i < len; i += 64) {
                const int pn = path_lds[i];
                visits[pn] += 1;
                vsum[pn] -= d.vl;
            }
        }
        wave_sync();
        FINE(15);
        slot += 1;
    }
#undef FINE
    ESTAMP(5);
    // the levels go back to memory (the backup that follows the network updates deeper nodes there), the root's to `rc`
#pragma unroll
    for (int L = 1; L <= DC; ++L) level_flush<CPL>(d, nb, lane, lv[L]);
#pragma unroll
    for (int k = 0; k < CPL; ++k) { rc.vis[k] = lv[0].vis[k]; rc.vsum[k] = lv[0].vsum[k]; rc.cf[k] = lv[0].cf[k]; }
    // this game's rows: allotted from the batch's counter (their order among the games does not matter); asked for first,
    // so that the answer travels while the record is written
    int row0 = 0;
    if (rows_ctr && lane == 0 && n_unique > 0) row0 = atomicAdd(rows_ctr, n_unique);
    // the batch's results: the record for the expansion, the path lengths and slot tables the leaf gather reads
    int* brec = d.brec + (size_t)g * brec_words(d);
    for (int sl = lane; sl < batch_n; sl += 64) {
        const int len = sl < 64 ? my_len : st_len[sl], ref = sl < 64 ? my_ref : st_ref[sl];
        const int n1 = sl < 64 ? my_n1 : st_n1[sl], leaf = sl < 64 ? my_leaf : st_leaf[sl];
        d.path_len[g * d.B + sl] = len;
        d.leaf_ref[g * d.B + sl] = ref;
        brec[4 + sl] = len | ((ref + 1) << 16);
        brec[4 + d.B + sl] = n1;
        brec[4 + 2 * d.B + sl] = leaf;
        int* path = d.path + ((size_t)g * d.B + sl) * d.max_depth;
        if (len == 2) { path[0] = 0; path[1] = leaf; }
        else if (len <= PSTAGE)
            for (int i = 0; i < len; ++i) path[i] = st_path[sl * PSTAGE + i];
    }
    for (int u = lane; u < n_unique; u += 64) { d.uniq_slot[g * d.B + u] = uniq_lds[2 * u]; brec[4 + 3 * d.B + u] = uniq_lds[2 * u]; }
    if (lane == 0) {
        brec[0] = n_unique; brec[1] = 0; brec[2] = rnc; brec[3] = rfirst;
        d.n_unique[g] = n_unique;
        long long* gs = d.gstats + (size_t)g * 8;
        gs[ST_DEPTH_SUM] += depth_sum;
        gs[ST_MAX_DEPTH] = max(gs[ST_MAX_DEPTH], (long long)max_depth);
        gs[ST_TERMINAL] += n_term;
        gs[ST_SIMS] += batch_n;
        gs[ST_ROWS_RAW] += batch_n - n_term;
        gs[ST_ROWS] += n_unique;
    }
    ESTAMP(6);
    if (rows_ctr) {
        row0 = __builtin_amdgcn_readfirstlane(row0);
        for (int u = lane; u < n_unique; u += 64) { d.row_game[row0 + u] = g; d.row_slot[row0 + u] = uniq_lds[2 * u]; }
        if (lane == 0) d.row_off[g] = row0;
    }
}

// The first batch of a search (the later ones run behind the previous batch's expansion in k_expand_select).
template <int PPL, int DC>
__device__ __forceinline__ void select_kernel(const Dev& d, int batch_n, int* rows_ctr, int visits_hint) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!d.active[g]) { if (lane == 0) { d.n_unique[g] = 0; d.brec[(size_t)g * brec_words(d)] = 0; } return; }
    const int t_idx = min(visits_hint + (d.vl > 0.0 ? lane : 0), d.tab_len - 1);
    const double t_pbc = d.pbc[t_idx], t_sqn = d.sqn[t_idx];
    const int passes = d.sc[sc_idx(g, SC_PASSES)], moves = d.sc[sc_idx(g, SC_MOVES)];
    const size_t nb = (size_t)g * d.max_nodes;
    RootCache<Apl<PPL>::v> rc;
    rc_load<Apl<PPL>::v>(d, nb, lane, d.n_child[nb], d.child_off[nb], rc);
    select_body<PPL, DC>(d, batch_n, rc, passes, moves, t_pbc, t_sqn, visits_hint, rows_ctr);
    rc_store<Apl<PPL>::v>(d, nb, lane, rc);
}
template <int PPL>
__global__ __launch_bounds__(64) void k_select(Dev d, int batch_n, int* rows_ctr, int visits_hint) {
    select_kernel<PPL, DeepLevels<Apl<PPL>::v>::v>(d, batch_n, rows_ctr, visits_hint);
}
template <int PPL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_select_wide(Dev d, int batch_n, int* rows_ctr, int visits_hint) {
    select_kernel<PPL, 0>(d, batch_n, rows_ctr, visits_hint);
}

// Leaf gather: one wave per pending row replays the row's path from its game's root and writes the
// feature planes (dense, row order) and the legal mask.
// n_rows: the device-side row count; zero_ctr (may be null): a counter block 0 clears for the batch after next.
template <int PPL>
__global__ __launch_bounds__(64) void k_leaf(Dev d, void* out, int layout, const int* n_rows, int* zero_ctr) {
    __shared__ BoardLds<PPL> b;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint64_t* seen_lds = (uint64_t*)dyn;
    int8_t* hist_lds = (int8_t*)(dyn + (size_t)d.seen_cap * 8);
    const int r = blockIdx.x, lane = threadIdx.x;
    if (zero_ctr && r == 0 && lane == 0) *zero_ctr = 0;
    if (r >= *n_rows) return;
    const int g = d.row_game[r], slot = d.row_slot[r];
    Geo<PPL> geo;
    geo_init<PPL>(geo, d.n, lane);
    Sim<PPL> s;
    sim_load_root<PPL>(d, g, lane, geo, b, seen_lds, s);
    {   // the game's H history boards are contiguous ([H][npad], npad = NPAD, a multiple of 64): 16 bytes per load
        const uint4* src = (const uint4*)(d.hist + (size_t)g * d.H * d.npad);
        for (int i = lane; i < d.H * BoardLds<PPL>::NPAD / 16; i += 64) ((uint4*)hist_lds)[i] = src[i];
    }
    int hist_count = d.sc[sc_idx(g, SC_HIST)];
    wave_sync();
    if (slot >= 0) {
        const int len = d.path_len[g * d.B + slot];
        const int* path = d.path + ((size_t)g * d.B + slot) * d.max_depth;
        const int16_t* action = d.action + (size_t)g * d.max_nodes;
        for (int i = 1; i < len; ++i) {
            sim_play<PPL>(d, lane, geo, b, seen_lds, hist_lds, s, (int)action[path[i]]);
            hist_count = min(hist_count + 1, d.H);
        }
    }
    const size_t row_bytes = layout == AZ_FEAT_F32_NCHW ? (size_t)d.C * d.np * 4 : (size_t)d.np * (d.C <= 32 ? 32 : 64) * 2;
    const int n_legal = emit_features<PPL>(d, lane, geo, b, seen_lds, hist_lds, BoardLds<PPL>::NPAD, hist_count, s,
                                           (unsigned char*)out + (size_t)r * row_bytes, layout, d.row_legal + (size_t)r * d.A);
    if (slot < 0 && lane == 0) d.root_count[g] = n_legal;
}

// numpy pairwise summation of a contiguous float64 vector (ndarray.sum(), mcts.py:80), n <= 504 (two levels of halving; the
// engine has n <= 362, checked in az_engine_create), by the whole wave with
// numpy's own addition tree. numpy splits n > 128 in halves (the first rounded down to a multiple of 8) until a piece has
// at most 128 elements, and sums a piece with eight interleaved accumulators r_j = a[j] + a[8+j] + a[16+j] + ... that
// are combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then adds the piece's last n%8 elements one by one. Here lane
// 8b+j owns accumulator j of piece b (at most 4 pieces), the combination is the xor-1/2/4 butterfly (the
// same tree: float addition commutes), and the pieces are added in numpy's order. Every lane returns the sum.
__device__ inline double pw_sum_wave(const double* a, int n, int lane) {
    if (n < 8) {   // wave-uniform
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    // piece boundaries: start[0..np], np in {1, 2, 3, 4}
    int start[5], np_ = 1;
    start[0] = 0; start[1] = n;
    if (n > 128) {
        int n2 = n / 2; n2 -= n2 % 8;
        int cuts[3], nc = 0;
        if (n2 > 128) { int h = n2 / 2; h -= h % 8; cuts[nc++] = h; }
        const int first_half_cuts = nc;
        cuts[nc++] = n2;
        if (n - n2 > 128) { int h = (n - n2) / 2; h -= h % 8; cuts[nc++] = n2 + h; }
        for (int k = 0; k < nc; ++k) start[k + 1] = cuts[k];
        start[nc + 1] = n;
        np_ = nc + 1;
        (void)first_half_cuts;
    }
    const int b = lane >> 3, j = lane & 7;
    double r = 0.0;
    int len = 0, s0 = 0;
    if (b < np_) {
        s0 = start[b]; len = start[b + 1] - s0;
        if (len >= 8) {
            r = a[s0 + j];
            const int lim = len - (len % 8);
            for (int i = 8; i < lim; i += 8) r += a[s0 + i + j];
        }
    }
    r += __shfl_xor(r, 1);
    r += __shfl_xor(r, 2);
    r += __shfl_xor(r, 4);
    if (b < np_) {
        if (len < 8) { r = 0.0; for (int i = 0; i < len; ++i) r += a[s0 + i]; }   // a short piece is summed one by one
        else for (int i = len - (len % 8); i < len; ++i) r += a[s0 + i];
    }
    // r of lane 8b = sum of piece b; numpy adds (piece0 + piece1) + (piece2 + piece3) when both halves were split,
    // (piece0 + piece1) + piece2 or piece0 + (piece1 + piece2) when one was, piece0 + piece1 when none was
    const double p0 = __shfl(r, 0), p1 = __shfl(r, 8), p2 = __shfl(r, 16), p3 = __shfl(r, 24);
    if (np_ == 1) return p0;
    if (np_ == 2) return p0 + p1;
    if (np_ == 4) return (p0 + p1) + (p2 + p3);
    int n2 = n / 2; n2 -= n2 % 8;
    return n2 > 128 ? (p0 + p1) + p2 : p0 + (p1 + p2);
}

// mcts.py:66-91 + :118-137 for one node: masked softmax (float64), optional noise, top-k, children.
// p_lds: double[2*A] work arrays (probabilities, ranking keys); runs on the whole wave. PPA: actions per lane, A <= 64 PPA.
// Returns the number of children the node received, or -1 when it had children already or the node pool is full.
// first_out (may be null): index of the first child created.
template <int PPA>
__device__ inline int expand_node(const Dev& d, int g, int lane, int node, const float* logits, const uint8_t* legal,
                                  const double* noise, double* p_lds, double* priors_out, int* first_out = nullptr) {
    const int A = d.A;
    double* q_lds = p_lds + A;
    bool lg_[PPA];
    double mx = -INFINITY;
    int nl = 0;
#pragma unroll
    for (int k = 0; k < PPA; ++k) {
        const int a = lane + 64 * k;
        lg_[k] = a < A && legal[a];
        nl += __popcll(__ballot(lg_[k]));
        if (a < A) {
            const double m = (double)logits[a] + (lg_[k] ? 0.0 : -1e9);
            p_lds[a] = m;
            mx = fmax(mx, m);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
#pragma unroll
    for (int k = 0; k < PPA; ++k) {
        const int a = lane + 64 * k;
        if (a < A) p_lds[a] = exp(p_lds[a] - mx);
    }
    __syncthreads();
    const double sum = pw_sum_wave(p_lds, A, lane);
    double pr[PPA];
#pragma unroll
    for (int k = 0; k < PPA; ++k) {
        const int a = lane + 64 * k;
        pr[k] = 0.0;
        if (a < A) pr[k] = sum <= 0 ? (lg_[k] ? 1.0 / (double)nl : 0.0) : p_lds[a] / sum;
    }
    if (noise) {  // mcts.py:121-124: noise[i] belongs to the i-th legal action
        int base = 0;
#pragma unroll
        for (int k = 0; k < PPA; ++k) {
            const unsigned long long bal = __ballot(lg_[k]);
            if (lg_[k]) {
                const int i = base + __popcll(bal & ((1ull << lane) - 1ull));
                pr[k] = d.one_minus_eps * pr[k] + d.eps * noise[i];
            }
            base += __popcll(bal);
        }
    }
    if (priors_out)
        for (int a = lane; a < A; a += 64) priors_out[a] = 0.0;
    const bool already = d.n_child[(size_t)g * d.max_nodes + node] > 0;  // mcts.py:160
    // child order: legal order, or rank by (prior desc, action asc) cut to K with the pass forced in
    int cnt = nl;
    const bool topk = d.K > 0 && nl > d.K;
    if (topk) cnt = d.K;
    int first = 0;
    if (!already) {
        if (lane == 0) {
            first = d.n_nodes[g];
            if (first + cnt > d.max_nodes) { atomicMin(&d.ctl[1], (int)AZ_ERR_CAPACITY); first = -1; }
            else d.n_nodes[g] = first + cnt;
        }
        first = __shfl(first, 0);
        if (first < 0) return -1;
    }
    int pos[PPA];
    if (!topk) {
        int base = 0;
#pragma unroll
        for (int k = 0; k < PPA; ++k) {
            const unsigned long long bal = __ballot(lg_[k]);
            pos[k] = lg_[k] ? base + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
            base += __popcll(bal);
        }
    } else if (A > 128) {
        // Large action sets (13x13, 19x19): the K kept children are extracted one by one, best first - K wave arg-max
        // rounds over each lane's best remaining action (prior desc, action asc: the arg-max returns the smallest index
        // among equal maxima), round r giving slot r. O(K * PPA) per lane instead of O(A * PPA) for counting ranks.
        unsigned taken = 0;
        bool pass_in = false;
        int last_idx = -1;
#pragma unroll
        for (int k = 0; k < PPA; ++k) pos[k] = -1;
        for (int r = 0; r < cnt; ++r) {
            double best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < PPA; ++k)
                if (lg_[k] && !((taken >> k) & 1u) && pr[k] > best) { best = pr[k]; bi = lane + 64 * k; }
            wave_argmax(best, bi);          // bi: the action extracted in this round (wave-uniform)
            if (bi == 0x7fffffff) break;    // cannot happen: nl > K legal actions remain
            if ((bi & 63) == lane) {
                const int k = bi >> 6;
#pragma unroll
                for (int kk = 0; kk < PPA; ++kk)
                    if (kk == k) { pos[kk] = r; taken |= 1u << kk; }
            }
            pass_in |= bi == A - 1;
            last_idx = bi;
        }
        // the pass (action A-1) is forced in: if it was not among the K extracted it takes the last slot and the action
        // extracted last is dropped
        if (!pass_in) {
#pragma unroll
            for (int k = 0; k < PPA; ++k) {
                const int a = lane + 64 * k;
                if (a == last_idx) pos[k] = -1;
                if (a == A - 1 && lg_[k]) pos[k] = cnt - 1;
            }
        }
    } else {
        // ranking keys: the prior of a legal action, -1 for an illegal one (priors are >= 0, so those never count)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PPA; ++k) {
            const int a = lane + 64 * k;
            if (a < A) q_lds[a] = lg_[k] ? pr[k] : -1.0;
        }
        __syncthreads();
        int rank[PPA];
#pragma unroll
        for (int k = 0; k < PPA; ++k) rank[k] = 0;
        const int kmax = (A + 63) / 64;   // wave-uniform
        for (int x = 0; x < A; ++x) {     // q_lds[x]: one broadcast read serves every lane and every k
            const double px = q_lds[x];
#pragma unroll
            for (int k = 0; k < PPA; ++k)
                if (k < kmax) rank[k] += (px > pr[k] || (px == pr[k] && x < lane + 64 * k)) ? 1 : 0;
        }
        // the pass (action A-1) is forced in: if it ranks outside the K kept, it takes the last slot and the action that
        // ranked K-1 is dropped. prank = legal stone moves with prior >= the pass's = the pass's rank among the stones
        const double pp = q_lds[A - 1];
        int prank = 0;
#pragma unroll
        for (int k = 0; k < PPA; ++k) {
            const int a = lane + 64 * k;
            prank += __popcll(__ballot(lg_[k] && a < A - 1 && pr[k] >= pp));
        }
#pragma unroll
        for (int k = 0; k < PPA; ++k) {
            const int a = lane + 64 * k;
            int ps = -1;
            if (lg_[k]) {
                ps = rank[k];
                if (a == A - 1 && rank[k] >= cnt) ps = cnt - 1;
                else if (rank[k] >= cnt) ps = -1;
                else if (rank[k] == cnt - 1 && a != A - 1 && prank >= cnt) ps = -1;
            }
            pos[k] = ps;
        }
    }
    const size_t nbase = (size_t)g * d.max_nodes;
#pragma unroll
    for (int k = 0; k < PPA; ++k) {
        const int a = lane + 64 * k;
        if (pos[k] >= 0) {
            if (priors_out) priors_out[a] = pr[k];
            if (!already) {
                const size_t c = nbase + first + pos[k];
                d.prior[c] = pr[k]; d.vsum[c] = 0.0; d.visits[c] = 0; d.action[c] = (int16_t)a;
                d.child_off[c] = 0; d.n_child[c] = 0;
            }
        }
    }
    if (!already && lane == 0) {
        d.child_off[nbase + node] = first;
        d.n_child[nbase + node] = (int16_t)cnt;
        d.gstats[(size_t)g * 8 + ST_MAX_NODES] = max(d.gstats[(size_t)g * 8 + ST_MAX_NODES], (long long)(first + cnt));
    }
    __threadfence_block();
    __syncthreads();
    if (first_out) *first_out = first;
    return already ? -1 : cnt;
}

// Root expansion (mcts.py:172): one wave per game.
// `enable` (may be null): games switched off after their root row was gathered leave the search here.
template <int PPL>
__global__ __launch_bounds__(64) void k_expand_root(Dev d, const float* logits, const double* noise, double* priors_out, const uint8_t* enable) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    double* p_lds = (double*)dyn;
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g == 0 && lane == 0) { d.ctl[4] = 0; d.ctl[5] = 0; }   // the row counters of the batch rounds (az_mcts_run_rounds)
    if (lane < 8) d.gstats[(size_t)g * 8 + lane] = 0;   // a search's counters start here (k_begin clears them too)
    __syncthreads();
    if (!d.active[g]) return;
    if (enable && !enable[g]) {
        __syncthreads();
        if (lane == 0) d.active[g] = 0;
        return;
    }
    const int r = d.row_off[g];
    const float* lg = (d.rc_cap && d.root_cached[g]) ? d.root_logits + (size_t)g * d.A : logits + (size_t)r * d.A;
    expand_node<Apl<PPL>::v>(d, g, lane, 0, lg, d.row_legal + (size_t)r * d.A,
                             noise ? noise + (size_t)g * d.A : nullptr, p_lds, priors_out ? priors_out + (size_t)r * d.A : nullptr);
}

// Expand the batch's leaves, then revert virtual loss and back up in slot order (mcts.py:203-218). Loads the root's
// statistics into `rc` and leaves them updated there, not stored.
template <int PPL, bool STAMPED = false>
__device__ __forceinline__ void expand_backup_body(const Dev& d, const float* logits, const float* values, int batch_n, double* priors_out,
                                                   RootCache<Apl<PPL>::v>& rc) {
    constexpr int CPL = Apl<PPL>::v;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    double* p_lds = (double*)dyn;
    double* s_val = p_lds + 2 * d.A;          // [B] value each slot backs up
    int* s_rec = (int*)(s_val + d.B);         // [4 + 4 B] the batch record the descents left
    const int g = blockIdx.x, lane = threadIdx.x, B = d.B;
    const size_t nb = (size_t)g * d.max_nodes;
    {
        const int* brec = d.brec + (size_t)g * brec_words(d);
        for (int i = lane; i < brec_words(d); i += 64) s_rec[i] = brec[i];
    }
    const int r0 = d.row_off[g];
    __syncthreads();
    if (STAMPED) ESTAMP(1);
    const int nu = s_rec[0];
    const int* r_len = s_rec + 4; const int* r_n1 = r_len + B; const int* r_leaf = r_n1 + B; const int* r_uniq = r_leaf + B;
    rc_load<CPL>(d, nb, lane, s_rec[2], s_rec[3], rc);
    // the value every slot backs up: its leaf's evaluation, or the outcome of a terminal leaf
    for (int s = lane; s < batch_n; s += 64) {
        const int ref = (r_len[s] >> 16) - 1;
        double value;
        if (ref < 0) value = d.term_value[g * B + s];
        else {
            int u = 0;
            for (; u < nu; ++u) if (r_uniq[u] == ref) break;
            value = (double)values[r0 + u];
        }
        s_val[s] = value;
    }
    if (STAMPED) ESTAMP(2);
    const bool cached = rc.held();
    const int rfirst = rc.rfirst;
    for (int u = 0; u < nu; ++u) {
        const int slot = r_uniq[u];
        const int plen = r_len[slot] & 0xffff, node = r_leaf[slot];
        int first_new = 0;
        const int made = expand_node<CPL>(d, g, lane, node, logits + (size_t)(r0 + u) * d.A, d.row_legal + (size_t)(r0 + u) * d.A, nullptr, p_lds,
                                          priors_out ? priors_out + (size_t)(r0 + u) * d.A : nullptr, &first_new);
        if (plen == 2) {   // a child of the root
            const int ci = node - rfirst;
            if (made >= 0 && cached) {
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    if (lane + 64 * k == ci) { rc.nch[k] = made; rc.cf[k] = first_new; }
            }
            if (d.rc_cap && ci >= 0 && ci < d.rc_cap) {   // a candidate for the next root: keep its evaluation
                float* dst = d.rc_logits + ((size_t)g * d.rc_cap + ci) * d.A;
                for (int i = lane; i < d.A; i += 64) dst[i] = logits[(size_t)(r0 + u) * d.A + i];
                if (lane == 0) d.rc_valid[(size_t)g * d.rc_cap + ci] = 1;
            }
        }
    }
    __syncthreads();
    if (STAMPED) ESTAMP(3);
    double* vsum = d.vsum + nb;
    int* visits = d.visits + nb;
    const bool vl_on = d.vl > 0.0;
    // Nodes below the root's children are updated in memory. When the batch's deep path nodes fit the LDS table below, every
    // distinct node is loaded once, takes the updates of all slots that pass through it in slot order (the order and the
    // operations of mcts.py:205-218 per node; different nodes do not interact) and is stored once; else slot by slot.
    int* e_node = (int*)p_lds;                      // the expansion's scratch is free now: [2 A] doubles
    const int ecap = (2 * d.A * 8 - 8) / 12;
    double* e_val = (double*)(e_node + ((ecap + 1) & ~1));
    int my_cnt = 0, my_off = 0, n_deep = 0;
    bool table = cached && batch_n <= 64;
    if (table) {
        if (lane < batch_n) my_cnt = max(0, (r_len[lane] & 0xffff) - 2);
        int incl = my_cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        my_off = incl - my_cnt;
        n_deep = __builtin_amdgcn_readlane(incl, 63);
        table = n_deep <= ecap;
    }
    for (int slot = 0; slot < batch_n; ++slot) {
        const double value = s_val[slot];
        const int len = r_len[slot] & 0xffff;
        if (cached) {
            {   // root (i = 0)
                const double v = ((len - 1) & 1) ? -value : value;
                int vc = rc.r_vis; double vs = rc.r_vsum;
                if (vl_on) { vc -= 1; vs += d.vl; }
                rc.r_vis = vc + 1; rc.r_vsum = vs + v;
            }
            if (len > 1) {   // the root child (i = 1)
                const int ci = r_n1[slot] - rfirst;
                const double v = ((len - 2) & 1) ? -value : value;
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                    if (lane + 64 * k == ci) {
                        int vc = rc.vis[k]; double vs = rc.vsum[k];
                        if (vl_on) { vc -= 1; vs += d.vl; }
                        rc.vis[k] = vc + 1; rc.vsum[k] = vs + v;
                    }
            }
        }
        if (!table && (!cached || len > 2)) {
            const int* path = d.path + ((size_t)g * d.B + slot) * d.max_depth;
            for (int i = (cached ? 2 : 0) + lane; i < len; i += 64) {
                const int pn = path[i];
                int vc = visits[pn];
                double vs = vsum[pn];
                if (vl_on) { vc -= 1; vs += d.vl; }
                const double v = ((len - 1 - i) & 1) ? -value : value;
                visits[pn] = vc + 1;
                vsum[pn] = vs + v;
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    if (table && n_deep > 0) {
        if (my_cnt > 0) {   // lane = slot: its path nodes from index 2 on, with the value each takes
            const int len = r_len[lane] & 0xffff;
            const double value = s_val[lane];
            const int* path = d.path + ((size_t)g * d.B + lane) * d.max_depth;
            for (int i = 2; i < len; ++i) {
                e_node[my_off + i - 2] = path[i];
                e_val[my_off + i - 2] = ((len - 1 - i) & 1) ? -value : value;
            }
        }
        __syncthreads();
        for (int e0 = 0; e0 < n_deep; e0 += 64) {
            const int e = e0 + lane;
            if (e < n_deep) {
                const int pn = e_node[e];
                bool first = true;
                for (int e2 = 0; e2 < e; ++e2) first = first && e_node[e2] != pn;
                if (first) {   // the node's first appearance: this lane applies all of them
                    int vc = visits[pn];
                    double vs = vsum[pn];
                    for (int e2 = e; e2 < n_deep; ++e2) {
                        if (e_node[e2] == pn) {
                            if (vl_on) { vc -= 1; vs += d.vl; }
                            vc += 1;
                            vs += e_val[e2];
                        }
                    }
                    visits[pn] = vc;
                    vsum[pn] = vs;
                }
            }
        }
    }
}

template <int PPL>
__global__ __launch_bounds__(64) void k_expand_backup(Dev d, const float* logits, const float* values, int batch_n, double* priors_out) {
    if (!d.active[blockIdx.x]) return;
    RootCache<Apl<PPL>::v> rc;
    expand_backup_body<PPL>(d, logits, values, batch_n, priors_out, rc);
    rc_store<Apl<PPL>::v>(d, (size_t)blockIdx.x * d.max_nodes, threadIdx.x, rc);
}

// Expansion + backup of one batch followed by the descents of the next, for the same game in the same wave: between the
// two nothing crosses games (the leaf gather and the network sit between a select and its expand), so a search enqueued
// as a whole saves a launch per round, the root's statistics stay in registers from the backup through the descents, and
// the descents' rows are allotted from a counter (rows_ctr) instead of a scan over the games.
template <int PPL, int DC>
__device__ __forceinline__ void expand_select_kernel(const Dev& d, const float* logits, const float* values, int batch_prev, double* priors_out,
                                                     int batch_next, int* rows_ctr, int visits_hint) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!d.active[g]) { if (lane == 0) d.n_unique[g] = 0; return; }
    // independent of the expansion: the exploration-table entries of the coming descents, the root's pass / move counters
    const int t_idx = min(visits_hint + (d.vl > 0.0 ? lane : 0), d.tab_len - 1);
    const double t_pbc = d.pbc[t_idx], t_sqn = d.sqn[t_idx];
    const int passes = d.sc[sc_idx(g, SC_PASSES)], moves = d.sc[sc_idx(g, SC_MOVES)];
    RootCache<Apl<PPL>::v> rc;
    ESTAMP(0);
    expand_backup_body<PPL, true>(d, logits, values, batch_prev, priors_out, rc);
    __threadfence_block();
    __syncthreads();   // deeper statistics written by the backup are read by the descents; the two phases share the LDS
    ESTAMP(4);
    select_body<PPL, DC>(d, batch_next, rc, passes, moves, t_pbc, t_sqn, visits_hint, rows_ctr);
    rc_store<Apl<PPL>::v>(d, (size_t)g * d.max_nodes, lane, rc);
    ESTAMP(7);
}
template <int PPL>
__global__ __launch_bounds__(64) void k_expand_select(Dev d, const float* logits, const float* values, int batch_prev, double* priors_out, int batch_next,
                                                      int* rows_ctr, int visits_hint) {
    expand_select_kernel<PPL, DeepLevels<Apl<PPL>::v>::v>(d, logits, values, batch_prev, priors_out, batch_next, rows_ctr, visits_hint);
}
template <int PPL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_expand_select_wide(Dev d, const float* logits, const float* values, int batch_prev,
                                                                                                    double* priors_out, int batch_next, int* rows_ctr, int visits_hint) {
    expand_select_kernel<PPL, 0>(d, logits, values, batch_prev, priors_out, batch_next, rows_ctr, visits_hint);
}

__global__ __launch_bounds__(64) void k_root_counts(Dev d, int* counts) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (!d.active[g]) { if (lane == 0) counts[g] = 0; return; }
    const uint8_t* legal = d.row_legal + (size_t)d.row_off[g] * d.A;
    int c = 0;
    for (int a = lane; a < d.A; a += 64) c += legal[a];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) counts[g] = c;
}

__global__ __launch_bounds__(64) void k_result(Dev d, int* visits_out, double* root_value) {
    const int g = blockIdx.x, lane = threadIdx.x;
    for (int a = lane; a < d.A; a += 64) visits_out[(size_t)g * d.A + a] = 0;
    if (!d.active[g]) { if (lane == 0) root_value[g] = 0.0; return; }
    __syncthreads();
    const size_t nb = (size_t)g * d.max_nodes;
    const int first = d.child_off[nb], nc = d.n_child[nb];
    for (int i = lane; i < nc; i += 64) visits_out[(size_t)g * d.A + d.action[nb + first + i]] = d.visits[nb + first + i];
    if (lane == 0) root_value[g] = d.visits[nb] == 0 ? 0.0 : d.vsum[nb] / (double)d.visits[nb];  // mcts.py:24-27
}

// Sum (or maximum, for the two high-water marks) of the per-game counters.
__global__ __launch_bounds__(256) void k_reduce_stats(Dev d) {
    __shared__ long long part[256][8];
    long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = threadIdx.x; g < d.G; g += 256) {
        const long long* gs = d.gstats + (size_t)g * 8;
        for (int i = 0; i < 8; ++i) acc[i] = (i == ST_MAX_DEPTH || i == ST_MAX_NODES) ? max(acc[i], gs[i]) : acc[i] + gs[i];
    }
    for (int i = 0; i < 8; ++i) part[threadIdx.x][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 8) {
        const int i = threadIdx.x;
        long long r = 0;
        for (int t = 0; t < 256; ++t) r = (i == ST_MAX_DEPTH || i == ST_MAX_NODES) ? max(r, part[t][i]) : r + part[t][i];
        d.stats[i] = r;
    }
}

}  // namespace az

// =================================================================================================
// Host side
using namespace az;

struct az_engine {
    az_config cfg;
    Dev d;
    int ppl;
    int simds = 1024;               // SIMDs of the device (4 per CU)
    int phase = PH_IDLE;
    int sims_done = 0, batch_n = 0;
    int pending_rows = 0;
    int last_rows = -1;             // rows the last az_mcts_expand produced priors for (-1: not known on the host)
    int gather_rows = 0;            // rows az_mcts_gather fills (pending_rows + roots whose logits were carried over)
    bool rows_known = false;
    bool async_mode = false;        // never synchronise inside a search: grids sized by upper bounds
    int* ctl_host = nullptr;        // pinned
    long long* stats_host = nullptr;
    double* last_priors = nullptr;  // [max_rows][A]
    const uint8_t* root_enable = nullptr;   // device mask a self-play lane hands to the root expansion (null: every active game)
    std::vector<void*> allocs;
    // az_mcts_run_rounds: the launches of all remaining batch rounds of a search, captured once and replayed per move
    hipGraphExec_t rounds_exec = nullptr;
    const void* rounds_key[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // net, features, logits, values, (layout, first batch), net generation
};

#define DISPATCH_PPL(e, KERNEL, grid, shmem, stream, ...)                                              \
    do {                                                                                               \
        switch ((e)->ppl) {                                                                            \
            case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(grid), dim3(64), shmem, (hipStream_t)stream, __VA_ARGS__); break; \
            case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(grid), dim3(64), shmem, (hipStream_t)stream, __VA_ARGS__); break; \
            case 3: hipLaunchKernelGGL((KERNEL<3>), dim3(grid), dim3(64), shmem, (hipStream_t)stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL((KERNEL<6>), dim3(grid), dim3(64), shmem, (hipStream_t)stream, __VA_ARGS__); break; \
        }                                                                                              \
    } while (0)

template <typename T>
static int dev_alloc(az_engine* e, T** p, size_t count) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    e->allocs.push_back(q);
    *p = (T*)q;
    return AZ_OK;
}
#define ALLOC(field, count)                                                \
    do {                                                                   \
        int rc_ = dev_alloc(e, &e->d.field, (size_t)(count));              \
        if (rc_ != AZ_OK) { az_engine_destroy(e); return rc_; }            \
    } while (0)

extern "C" {

const char* az_last_error(void) { return g_err.c_str(); }
uint64_t az_zobrist_key(int32_t index) { return zobrist_key(index); }

int az_engine_create(const az_config* c, az_engine** out) {
    if (!c || !out) return fail(AZ_ERR_ARG, "null argument");
    if (c->board_size < 2 || c->board_size > 19) return fail(AZ_ERR_ARG, "board_size must be 2..19");
    if (c->history_len < 1 || c->history_len > 16) return fail(AZ_ERR_ARG, "history_len must be 1..16");
    if (c->games < 1) return fail(AZ_ERR_ARG, "games must be >= 1");
    if (c->simulations < 0) return fail(AZ_ERR_ARG, "simulations must be >= 0");
    HIP_TRY(hipSetDevice(c->device));
    az_engine* e = new az_engine();
    e->cfg = *c;
    Dev& d = e->d;
    memset(&d, 0, sizeof(d));
    d.n = c->board_size; d.np = d.n * d.n; d.A = d.np + 1; d.H = c->history_len; d.C = 2 * d.H + 5; d.G = c->games;
    d.sims = c->simulations;
    d.B = std::max(1, c->inference_batch);          // mcts.py:53
    d.K = std::max(0, c->topk_expand);              // mcts.py:54
    d.vl = std::max(0.0, c->virtual_loss);          // mcts.py:55
    d.dedup = c->dedup_leaves ? 1 : 0;
    d.komi = c->komi; d.eps = c->dirichlet_eps; d.one_minus_eps = 1 - c->dirichlet_eps;
    e->ppl = d.np <= 64 ? 1 : d.np <= 128 ? 2 : d.np <= 192 ? 3 : 6;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0) e->simds = 4 * cus;
    }
    {   // what the search kernels are laid out for: actions per lane (Apl), the pairwise-sum pieces, the packed child word
        const int apl = e->ppl <= 2 ? 2 : e->ppl;
        if (d.A > 64 * apl || d.A > 504 || d.A > 1023) { delete e; return fail(AZ_ERR_ARG, "action count above the search kernels' layout"); }
    }
    d.npad = 64 * e->ppl;
    d.seen_cap = 2 * d.np + 2;
    const int kc = d.K > 0 ? std::min(d.K, d.A) : d.A;
    d.max_nodes = 1 + (d.sims + 1) * kc;
    d.max_depth = std::min(d.sims, 2 * d.np) + 2;
    d.max_rows = d.G * d.B;
    d.tab_len = d.sims + d.B + 2;

    ALLOC(board, (size_t)d.G * d.npad); ALLOC(hist, (size_t)d.G * d.H * d.npad);
    ALLOC(seen, (size_t)d.G * d.seen_cap); ALLOC(hash, d.G); ALLOC(sc, (size_t)d.G * 8);
    const size_t nn = (size_t)d.G * d.max_nodes;
    ALLOC(prior, nn); ALLOC(vsum, nn); ALLOC(visits, nn); ALLOC(action, nn); ALLOC(child_off, nn); ALLOC(n_child, nn);
    ALLOC(n_nodes, d.G); ALLOC(active, d.G);
    ALLOC(path, (size_t)d.G * d.B * d.max_depth); ALLOC(path_len, (size_t)d.G * d.B); ALLOC(leaf_ref, (size_t)d.G * d.B);
    ALLOC(term_value, (size_t)d.G * d.B); ALLOC(n_unique, d.G); ALLOC(uniq_slot, (size_t)d.G * d.B);
    ALLOC(row_off, d.G + 1); ALLOC(row_game, d.max_rows); ALLOC(row_slot, d.max_rows);
    ALLOC(row_legal, (size_t)d.max_rows * d.A); ALLOC(ctl, 8); ALLOC(stats, 8); ALLOC(gstats, (size_t)d.G * 8);
    ALLOC(root_count, d.G); ALLOC(brec, (size_t)d.G * (4 + 4 * d.B));
    {
        int rc = dev_alloc(e, &e->last_priors, (size_t)d.max_rows * d.A);
        if (rc != AZ_OK) { az_engine_destroy(e); return rc; }
    }
    // tables (host libm, like math.log / math.sqrt in mcts.py:104-105)
    std::vector<uint64_t> zob(kZobSide + 2);
    for (int i = 0; i < kZobSide + 2; ++i) zob[i] = zobrist_key(i);
    std::vector<double> pbc(d.tab_len), sqn(d.tab_len);
    for (int N = 0; N < d.tab_len; ++N) {
        pbc[N] = log(((double)N + c->c_base + 1.0) / c->c_base) + c->c_init;
        sqn[N] = sqrt((double)N + 1.0);
    }
    uint64_t* zdev; double *pdev, *sdev;
    if (dev_alloc(e, &zdev, zob.size()) || dev_alloc(e, &pdev, pbc.size()) || dev_alloc(e, &sdev, sqn.size())) {
        az_engine_destroy(e);
        return AZ_ERR_HIP;
    }
    // from here on a failing HIP call must release the engine before returning
#define HIP_TRY_E(x)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            az_engine_destroy(e);                                                               \
            return fail(AZ_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));            \
        }                                                                                       \
    } while (0)
    HIP_TRY_E(hipMemcpy(zdev, zob.data(), zob.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(pdev, pbc.data(), pbc.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(sdev, sqn.data(), sqn.size() * 8, hipMemcpyHostToDevice));
    d.zob = zdev; d.pbc = pdev; d.sqn = sdev;
    HIP_TRY_E(hipHostMalloc((void**)&e->ctl_host, 4 * sizeof(int)));
    HIP_TRY_E(hipHostMalloc((void**)&e->stats_host, 8 * sizeof(long long)));
    HIP_TRY_E(hipMemset(d.ctl, 0, 8 * sizeof(int)));
    HIP_TRY_E(hipMemset(d.brec, 0, (size_t)d.G * (4 + 4 * d.B) * sizeof(int)));
    HIP_TRY_E(hipMemset(d.stats, 0, 8 * sizeof(long long)));
    HIP_TRY_E(hipMemset(d.gstats, 0, (size_t)d.G * 8 * sizeof(long long)));
    int rc = az_rules_reset(e, nullptr, nullptr);
    if (rc != AZ_OK) { az_engine_destroy(e); return rc; }
    HIP_TRY_E(hipStreamSynchronize(nullptr));
#undef HIP_TRY_E
    *out = e;
    return AZ_OK;
}

int az_engine_destroy(az_engine* e) {
    if (!e) return AZ_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    if (e->rounds_exec) (void)hipGraphExecDestroy(e->rounds_exec);
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->ctl_host) (void)hipHostFree(e->ctl_host);
    if (e->stats_host) (void)hipHostFree(e->stats_host);
    delete e;
    return AZ_OK;
}

int az_engine_action_size(const az_engine* e) { return e->d.A; }
int az_engine_feature_planes(const az_engine* e) { return e->d.C; }
int az_engine_seen_capacity(const az_engine* e) { return e->d.seen_cap; }
int az_engine_max_rows(const az_engine* e) { return std::max(e->d.max_rows, e->d.G); }

static size_t seen_bytes(const az_engine* e) { return (size_t)e->d.seen_cap * 8; }
static int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(AZ_ERR_HIP, std::string(what) + ": " + hipGetErrorString(err));
    return AZ_OK;
}

int az_rules_reset(az_engine* e, const uint8_t* mask, void* stream) {
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DISPATCH_PPL(e, k_reset, e->d.G, 0, stream, e->d, mask);
    e->phase = PH_IDLE;
    return check_launch("k_reset");
}

int az_rules_play(az_engine* e, const int32_t* actions, int32_t* status, void* stream) {
    if (!e || !actions) return fail(AZ_ERR_ARG, "null argument");
    DISPATCH_PPL(e, k_play, e->d.G, seen_bytes(e), stream, e->d, actions, status);
    e->phase = PH_IDLE;
    return check_launch("k_play");
}

int az_rules_legal_mask(az_engine* e, uint8_t* mask, void* stream) {
    if (!e || !mask) return fail(AZ_ERR_ARG, "null argument");
    DISPATCH_PPL(e, k_root_features, e->d.G, seen_bytes(e), stream, e->d, (void*)nullptr, 0, mask);
    return check_launch("k_root_features");
}

int az_rules_features(az_engine* e, void* out, int32_t layout, void* stream) {
    if (!e || !out) return fail(AZ_ERR_ARG, "null argument");
    if (layout != AZ_FEAT_F32_NCHW && layout != AZ_FEAT_BF16_NHWC) return fail(AZ_ERR_ARG, "bad layout");
    DISPATCH_PPL(e, k_root_features, e->d.G, seen_bytes(e), stream, e->d, out, layout, (uint8_t*)nullptr);
    return check_launch("k_root_features");
}

int az_rules_status(az_engine* e, double* score, int32_t* info, void* stream) {
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    DISPATCH_PPL(e, k_status, e->d.G, seen_bytes(e), stream, e->d, score, info);
    return check_launch("k_status");
}

int az_rules_export(az_engine* e, int8_t* board, int8_t* hist, int32_t* scalars, uint64_t* seen, void* stream) {
    if (!e || !board || !hist || !scalars || !seen) return fail(AZ_ERR_ARG, "null argument");
    hipLaunchKernelGGL(k_export, dim3(e->d.G), dim3(128), 0, (hipStream_t)stream, e->d, board, hist, scalars, seen);
    return check_launch("k_export");
}

int az_rules_import(az_engine* e, const int8_t* board, const int8_t* hist, const int32_t* scalars, const uint64_t* seen,
                    void* stream) {
    if (!e || !board || !hist || !scalars || !seen) return fail(AZ_ERR_ARG, "null argument");
    hipLaunchKernelGGL(k_import, dim3(e->d.G), dim3(128), 0, (hipStream_t)stream, e->d, board, hist, scalars, seen);
    e->phase = PH_IDLE;
    return check_launch("k_import");
}

int az_mcts_begin(az_engine* e, const uint8_t* enable, void* stream) {
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    hipLaunchKernelGGL(k_begin, dim3((e->d.G + 63) / 64), dim3(64), 0, (hipStream_t)stream, e->d, enable);
    hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->d, 1);
    e->phase = PH_ROOT;
    e->sims_done = 0;
    e->rows_known = false;
    return check_launch("k_begin");
}

// dynamic LDS of the descents: seen keys, then path, uniq pairs, staged len / ref / root child / leaf / path[8], terminal-leaf
// cache (node, value)
// More games than the chip holds waves of the regular descent kernels (two per SIMD): the four-per-SIMD kernels.
// (not for boards above 13x13, whose six children per lane do not fit 128 registers)
static bool wide_search(const az_engine* e) { return e->d.G > 2 * e->simds && e->ppl <= 3; }
static size_t select_shmem(const az_engine* e) {
    return seen_bytes(e) + ((size_t)e->d.max_depth + (2 + 4 + 8 + 1 + 2) * (size_t)e->d.B) * 4;
}
// dynamic LDS of the expansion + backup: probabilities and ranking keys, per-slot values, the batch record
static size_t expand_shmem(const az_engine* e) { return (size_t)e->d.A * 16 + (size_t)e->d.B * 24 + 64; }

int az_mcts_select(az_engine* e, void* stream) {
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    if (e->phase != PH_NEED_SELECT) return fail(AZ_ERR_STATE, "az_mcts_select: no batch to select (expand the pending rows first)");
    e->batch_n = std::min(e->d.B, e->d.sims - e->sims_done);
    if (wide_search(e)) DISPATCH_PPL(e, k_select_wide, e->d.G, select_shmem(e), stream, e->d, e->batch_n, (int*)nullptr, e->sims_done);
    else DISPATCH_PPL(e, k_select, e->d.G, select_shmem(e), stream, e->d, e->batch_n, (int*)nullptr, e->sims_done);
    hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->d, 0);
    e->phase = PH_SELECTED;
    e->rows_known = false;
    return check_launch("k_select");
}

int az_mcts_pending(az_engine* e, int32_t* rows_out, void* stream) {
    if (!e || !rows_out) return fail(AZ_ERR_ARG, "null argument");
    if (e->phase != PH_ROOT && e->phase != PH_SELECTED) { *rows_out = 0; return AZ_OK; }
    HIP_TRY(hipMemcpyAsync(e->ctl_host, e->d.ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    e->pending_rows = e->ctl_host[0];
    e->gather_rows = e->ctl_host[2];
    e->rows_known = true;
    *rows_out = e->pending_rows;
    if (e->ctl_host[1] == AZ_ERR_NOCHILD) return fail(AZ_ERR_NOCHILD, "No child selected in MCTS");
    if (e->ctl_host[1] == AZ_ERR_CAPACITY) return fail(AZ_ERR_CAPACITY, "search tree node capacity exceeded");
    return AZ_OK;
}

static int need_rows(az_engine* e, void* stream);

int az_mcts_gather_rows(az_engine* e, int32_t* rows_out, void* stream) {
    if (!e || !rows_out) return fail(AZ_ERR_ARG, "null argument");
    if (e->phase != PH_ROOT && e->phase != PH_SELECTED) { *rows_out = 0; return AZ_OK; }
    int rc = need_rows(e, stream);
    if (rc != AZ_OK) return rc;
    *rows_out = e->gather_rows;
    return AZ_OK;
}

int az_mcts_set_root_reuse(az_engine* e, int32_t enable) {
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    Dev& d = e->d;
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    // kernels take the device struct by value: a graph captured with the old one must not be replayed
    if (e->rounds_exec) { (void)hipGraphExecDestroy(e->rounds_exec); e->rounds_exec = nullptr; }
    if (!enable) { d.rc_cap = 0; return AZ_OK; }
    const int cap = d.K > 0 ? std::min(d.K, d.A) : d.A;
    if ((size_t)d.G * cap * d.A * 4 > ((size_t)1 << 30)) return fail(AZ_ERR_CAPACITY, "root reuse: logits store above 1 GiB");
    if (!d.rc_logits) {
        if (dev_alloc(e, &d.rc_logits, (size_t)d.G * cap * d.A) || dev_alloc(e, &d.rc_valid, (size_t)d.G * cap) ||
            dev_alloc(e, &d.root_logits, (size_t)d.G * d.A) || dev_alloc(e, &d.root_cached, (size_t)d.G) || dev_alloc(e, &d.tree_live, (size_t)d.G))
            return AZ_ERR_HIP;
    }
    HIP_TRY(hipMemset(d.root_cached, 0, d.G));
    HIP_TRY(hipMemset(d.tree_live, 0, d.G));
    HIP_TRY(hipMemset(d.rc_valid, 0, (size_t)d.G * cap));
    d.rc_cap = cap;
    return AZ_OK;
}

int az_mcts_set_async(az_engine* e, int32_t enable) {
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    e->async_mode = enable != 0;
    return AZ_OK;
}

int az_mcts_rows_device(az_engine* e, const int32_t** rows_dev_out) {
    if (!e || !rows_dev_out) return fail(AZ_ERR_ARG, "null argument");
    *rows_dev_out = e->d.ctl;
    return AZ_OK;
}

static int need_rows(az_engine* e, void* stream) {
    if (e->rows_known) return AZ_OK;
    int32_t r;
    return az_mcts_pending(e, &r, stream);
}

int az_mcts_gather(az_engine* e, void* out, int32_t layout, void* stream) {
    if (!e || !out) return fail(AZ_ERR_ARG, "null argument");
    if (e->phase != PH_ROOT && e->phase != PH_SELECTED) return fail(AZ_ERR_STATE, "az_mcts_gather: no pending rows");
    if (layout != AZ_FEAT_F32_NCHW && layout != AZ_FEAT_BF16_NHWC) return fail(AZ_ERR_ARG, "bad layout");
    int grid;
    if (e->async_mode && !e->rows_known) {
        grid = e->phase == PH_ROOT ? e->d.G : e->d.max_rows;   // surplus waves see r >= rows and leave
    } else {
        int rc = need_rows(e, stream);
        if (rc != AZ_OK) return rc;
        grid = e->gather_rows;
    }
    if (grid == 0) return AZ_OK;
    const size_t shmem = seen_bytes(e) + (size_t)e->d.H * e->d.npad;
    DISPATCH_PPL(e, k_leaf, grid, shmem, stream, e->d, out, layout, (const int*)(e->d.ctl + 2), (int*)nullptr);
    return check_launch("k_leaf");
}

int az_mcts_row_games(az_engine* e, int32_t* row_game, void* stream) {
    if (!e || !row_game) return fail(AZ_ERR_ARG, "null argument");
    int rc = need_rows(e, stream);
    if (rc != AZ_OK) return rc;
    HIP_TRY(hipMemcpyAsync(row_game, e->d.row_game, (size_t)e->pending_rows * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AZ_OK;
}

int az_mcts_expand(az_engine* e, const float* logits, const float* values, const double* noise, void* stream) {
    if (!e) return fail(AZ_ERR_ARG, "null engine");
    const size_t shmem = expand_shmem(e);
    if (e->phase == PH_ROOT) {
        if (!logits && !(e->d.rc_cap && e->rows_known && e->pending_rows == 0)) return fail(AZ_ERR_ARG, "null logits");
        DISPATCH_PPL(e, k_expand_root, e->d.G, shmem, stream, e->d, logits, noise, e->last_priors, e->root_enable);
    } else if (e->phase == PH_SELECTED) {
        if (!e->async_mode) {
            int rc = need_rows(e, stream);
            if (rc != AZ_OK) return rc;
            if (e->pending_rows > 0 && (!logits || !values)) return fail(AZ_ERR_ARG, "null logits/values");
        } else if (!logits || !values) {
            return fail(AZ_ERR_ARG, "null logits/values");
        }
        DISPATCH_PPL(e, k_expand_backup, e->d.G, shmem, stream, e->d, logits, values, e->batch_n, e->last_priors);
        e->sims_done += e->batch_n;
    } else {
        return fail(AZ_ERR_STATE, "az_mcts_expand: nothing pending");
    }
    int rc = check_launch("k_expand");
    if (rc != AZ_OK) return rc;
    e->last_rows = e->rows_known ? e->gather_rows : -1;
    e->rows_known = false;
    if (e->sims_done >= e->d.sims) { e->phase = PH_DONE; return 1; }
    e->phase = PH_NEED_SELECT;
    return 0;
}

// Enqueue the batch rounds from the engine's state (az_mcts_select .. az_mcts_expand until the simulations are used up).
// Round r: the descents of batch r (alone for the first, behind the expansion of batch r-1 afterwards) allot their rows from
// counter r & 1 (d.ctl[4 + (r & 1)]), the leaf gather and the network read that counter, and the gather clears the other one
// for round r+1 (whose readers, the gather and the network of round r-1, are done by then).
static int enqueue_rounds(az_engine* e, az_net* net, void* feat, int32_t layout, float* logits, float* values, void* stream) {
    if (e->phase != PH_NEED_SELECT) return fail(AZ_ERR_STATE, "no batch to select (expand the pending rows first)");
    const int32_t max_rows = std::max(e->d.max_rows, e->d.G);
    const size_t leaf_shmem = seen_bytes(e) + (size_t)e->d.H * e->d.npad;
    int parity = 0;
    e->batch_n = std::min(e->d.B, e->d.sims - e->sims_done);
    if (wide_search(e)) DISPATCH_PPL(e, k_select_wide, e->d.G, select_shmem(e), stream, e->d, e->batch_n, e->d.ctl + 4 + parity, e->sims_done);
    else DISPATCH_PPL(e, k_select, e->d.G, select_shmem(e), stream, e->d, e->batch_n, e->d.ctl + 4 + parity, e->sims_done);
    e->phase = PH_SELECTED;
    e->rows_known = false;
    for (;;) {
        DISPATCH_PPL(e, k_leaf, e->d.max_rows, leaf_shmem, stream, e->d, feat, layout, (const int*)(e->d.ctl + 4 + parity), e->d.ctl + 4 + (parity ^ 1));
        int rc = check_launch("k_leaf");
        if (rc != AZ_OK) return rc;
        rc = az_net_forward_indirect(net, feat, e->d.ctl + 4 + parity, max_rows, logits, values, stream);
        if (rc != AZ_OK) return rc;
        const int left = e->d.sims - (e->sims_done + e->batch_n);
        if (left <= 0) {   // the last batch: expansion and backup only
            rc = az_mcts_expand(e, logits, values, nullptr, stream);
            return rc < 0 ? rc : AZ_OK;
        }
        // expansion + backup of this batch and the descents of the next in one launch
        const int batch_next = std::min(e->d.B, left);
        const size_t shmem = std::max(expand_shmem(e), select_shmem(e));
        parity ^= 1;
        if (wide_search(e))
            DISPATCH_PPL(e, k_expand_select_wide, e->d.G, shmem, stream, e->d, (const float*)logits, (const float*)values, e->batch_n, e->last_priors,
                         batch_next, e->d.ctl + 4 + parity, e->sims_done + e->batch_n);
        else
            DISPATCH_PPL(e, k_expand_select, e->d.G, shmem, stream, e->d, (const float*)logits, (const float*)values, e->batch_n, e->last_priors,
                         batch_next, e->d.ctl + 4 + parity, e->sims_done + e->batch_n);
        rc = check_launch("k_expand_select");
        if (rc != AZ_OK) return rc;
        e->sims_done += e->batch_n;
        e->batch_n = batch_next;
        e->phase = PH_SELECTED;
        e->rows_known = false;
        e->last_rows = -1;
    }
}

int az_mcts_run_rounds(az_engine* e, az_net* net, void* feat, int32_t layout, float* logits, float* values, int32_t use_graph, void* stream) {
    if (!e || !net || !feat || !logits || !values) return fail(AZ_ERR_ARG, "null argument");
    if (!e->async_mode) return fail(AZ_ERR_STATE, "az_mcts_run_rounds needs asynchronous mode (az_mcts_set_async)");
    if (e->phase == PH_DONE) return AZ_OK;
    if (e->phase != PH_NEED_SELECT) return fail(AZ_ERR_STATE, "az_mcts_run_rounds: expand the root rows first");
    const int32_t max_rows = std::max(e->d.max_rows, e->d.G);
    // the legacy default stream cannot be captured
    if (!use_graph || !stream || !az_net_capturable(net, max_rows)) return enqueue_rounds(e, net, feat, layout, logits, values, stream);
    // One graph per (network, buffers, layout, simulations already done): the launch arguments of a move's rounds never
    // change - row counts live on the device - so later moves replay it.
    // ... and state of the network (a reload frees the weights the captured launches point at, a new workgroup limit or
    // scratch changes their arguments): net_generation
    const void* key[6] = {net, feat, logits, values, (const void*)(uintptr_t)(((uint64_t)(uint32_t)layout << 32) | (uint32_t)e->sims_done),
                          (const void*)(uintptr_t)net_generation(net)};
    if (!e->rounds_exec || memcmp(key, e->rounds_key, sizeof(key)) != 0) {
        if (e->rounds_exec) { (void)hipGraphExecDestroy(e->rounds_exec); e->rounds_exec = nullptr; }
        const int phase0 = e->phase, sims0 = e->sims_done;
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();   // this stream cannot be captured: enqueue the rounds one by one
            return enqueue_rounds(e, net, feat, layout, logits, values, stream);
        }
        const int rc = enqueue_rounds(e, net, feat, layout, logits, values, stream);
        const hipError_t ec = hipStreamEndCapture((hipStream_t)stream, &graph);
        e->phase = phase0; e->sims_done = sims0; e->rows_known = false;   // nothing has run yet
        if (rc != AZ_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ec != hipSuccess || !graph) return fail(AZ_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
        const hipError_t ei = hipGraphInstantiate(&e->rounds_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { e->rounds_exec = nullptr; return fail(AZ_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
        memcpy(e->rounds_key, key, sizeof(key));
    }
    HIP_TRY(hipGraphLaunch(e->rounds_exec, (hipStream_t)stream));
    { const int rem = e->d.sims - e->sims_done; e->batch_n = rem % e->d.B ? rem % e->d.B : e->d.B; }   // the last round's
    e->sims_done = e->d.sims;
    e->phase = PH_DONE;
    e->rows_known = false;
    e->last_rows = -1;
    return 1;   // replayed from the graph
}

int az_mcts_root_legal_counts(az_engine* e, int32_t* counts, void* stream) {
    if (!e || !counts) return fail(AZ_ERR_ARG, "null argument");
    if (e->phase != PH_ROOT) return fail(AZ_ERR_STATE, "root legal counts are available between gather and expand of the root rows");
    hipLaunchKernelGGL(k_root_counts, dim3(e->d.G), dim3(64), 0, (hipStream_t)stream, e->d, counts);
    return check_launch("k_root_counts");
}

int az_mcts_result(az_engine* e, int32_t* visits, double* root_value, void* stream) {
    if (!e || !visits || !root_value) return fail(AZ_ERR_ARG, "null argument");
    hipLaunchKernelGGL(k_result, dim3(e->d.G), dim3(64), 0, (hipStream_t)stream, e->d, visits, root_value);
    return check_launch("k_result");
}

int az_mcts_last_priors(az_engine* e, double* priors, void* stream) {
    if (!e || !priors) return fail(AZ_ERR_ARG, "null argument");
    if (e->last_rows < 0) return fail(AZ_ERR_STATE, "az_mcts_last_priors: the row count of the last expansion was never read (asynchronous mode)");
    HIP_TRY(hipMemcpyAsync(priors, e->last_priors, (size_t)e->last_rows * e->d.A * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AZ_OK;
}

int az_mcts_stats(az_engine* e, int64_t* out, void* stream) {
    if (!e || !out) return fail(AZ_ERR_ARG, "null argument");
    hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(256), 0, (hipStream_t)stream, e->d);
    HIP_TRY(hipMemcpyAsync(e->stats_host, e->d.stats, 8 * sizeof(long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (e->async_mode) HIP_TRY(hipMemcpyAsync(e->ctl_host, e->d.ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < 8; ++i) out[i] = e->stats_host[i];
    if (e->async_mode) {   // failures of the batches that were enqueued without a round trip surface here
        if (e->ctl_host[1] == AZ_ERR_NOCHILD) return fail(AZ_ERR_NOCHILD, "No child selected in MCTS");
        if (e->ctl_host[1] == AZ_ERR_CAPACITY) return fail(AZ_ERR_CAPACITY, "search tree node capacity exceeded");
    }
    return AZ_OK;
}

}  // extern "C"

#ifdef AZ_ESTAMP
extern "C" int az_debug_estamps(unsigned long long* out, int words) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_estamp), (size_t)words * 8));
    return AZ_OK;
}
extern "C" int az_debug_einfo(unsigned int* out, int words) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_einfo), (size_t)words * 4));
    return AZ_OK;
}
#endif

#include "az_lane.cuh"
