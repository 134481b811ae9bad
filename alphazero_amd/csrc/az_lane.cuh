// Self-play lane: one move of every lockstep game per enqueue, with one host round trip per move.
// Included at the end of az_engine.hip (it uses the engine's device state and kernels).
//
// What upstream does per game and move in Python (go_zero/train.py:322-343 around MCTS.run, mcts.py:227-284) runs here
// as one wave per game in k_move_tail: visit counts -> policy target with the move's temperature (mcts.py:270-279), the
// sampled action (np.random.choice given its uniform draw, mcts.py:281), difficulty (train.py:336-337), the resignation
// test (train.py:339-341), the sample's feature planes (go.py:265-304, bit-packed), GoState.play (go.py:159-203), the
// terminal test and winner (go.py:205-263), a fresh game in a slot whose game ended, and the root node of the next search.
// The host keeps what must stay on the host for bit-equal games: the per-game legacy-NumPy random streams (root Dirichlet
// noise and the uniform behind the move choice, drawn in upstream's order) and the table of visits**(1/T), which the
// caller fills with NumPy's own pow.
//
// A move is: one host-to-device copy (slot flags, move indices, uniforms, noise) -> root expansion -> the batch rounds ->
// k_move_tail -> row scan, leaf gather and network evaluation of the next roots -> one device-to-host copy (per-slot
// record, sample, legal-move counts of the next roots). The host waits once, for that copy.
#pragma once
#include <chrono>

namespace az {

enum { REC_ACTION = 0, REC_RESIGNED, REC_STATUS, REC_OVER, REC_WINNER, REC_TO_PLAY, REC_SIMS, REC_ROWS, REC_ROWS_RAW,
       REC_DEPTH_SUM, REC_MAX_DEPTH, REC_MAX_NODES, REC_TERMINAL, REC_ERR, REC_ON, REC_WORDS = 16 };

struct MoveDev {
    // host -> device, once per move
    const uint8_t* slot_on;   // [G] the slot holds a running game
    const int* move_idx;      // [G] moves played so far in the slot's game (train.py:320, :343)
    const double* u;          // [G] the uniform draw behind np.random.choice (mcts.py:281)
    const double* noise;      // [G][A] Dirichlet sample in the first n_legal entries (mcts.py:122)
    const int* flags;         // [4]: [0] != 0: a slot whose game ends starts a fresh game (else it goes idle)
    // device -> host, once per move
    int* rec;                 // [G][REC_WORDS]
    double* recd;             // [G][2] root value, difficulty
    float* pi32;              // [G][A]
    uint8_t* bits;            // [G][row_bytes] the 2H+3 binary planes, numpy.packbits order
    float* caps;              // [G][2] the two capture planes' values
    const double* powtab;     // [3][pow_len]: visits ** (1 / T) for the three temperatures
    int pow_len;
    double temp[3];
    int temp_moves, temp_mid_moves;
    double resign_threshold;
    int resign_on, resign_min_move;
    int row_bytes, collect;
};

// dynamic LDS of k_move_tail: seen keys, probabilities, cumulative sums, visit counts, packed-plane words
static size_t tail_shmem(const Dev& d, int row_bytes) {
    return (size_t)d.seen_cap * 8 + (size_t)d.A * 16 + (size_t)(d.A + 1) / 2 * 8 + (size_t)(row_bytes + 7) / 8 * 8 + 16;
}

template <int PPL>
__global__ __launch_bounds__(64) void k_move_tail(Dev d, MoveDev m) {
    __shared__ BoardLds<PPL> b;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    uint64_t* seen_lds = (uint64_t*)dyn;
    double* p_lds = (double*)(seen_lds + d.seen_cap);
    double* cdf = p_lds + d.A;
    int* vis = (int*)(cdf + d.A);
    uint32_t* bitw = (uint32_t*)(vis + 2 * ((d.A + 1) / 2));
    const int g = blockIdx.x, lane = threadIdx.x, A = d.A;
    int* rec = m.rec + (size_t)g * REC_WORDS;
    const size_t nb = (size_t)g * d.max_nodes;
    int err = 0;
    if (g == 0) {   // sticky failure code of this move's rounds (a select without a child, a full node pool)
        err = d.ctl[1];
        __syncthreads();
        if (lane == 0) d.ctl[1] = 0;
    }
    if (!m.slot_on[g]) {   // an empty slot: no record, no search
        if (lane < REC_WORDS) rec[lane] = lane == REC_ACTION ? -1 : (lane == REC_ERR ? err : 0);
        if (lane == 0) { d.active[g] = 0; d.n_unique[g] = 0; }
        return;
    }
    Geo<PPL> geo;
    geo_init<PPL>(geo, d.n, lane);
    Sim<PPL> s;
    sim_load_root<PPL>(d, g, lane, geo, b, seen_lds, s);
    const int tp0 = s.tp;

    // ---- visit counts and root value (mcts.py:222-225)
    for (int a = lane; a < A; a += 64) vis[a] = 0;
    wave_sync();
    const bool searched = d.active[g] != 0;
    {
        const int first = d.child_off[nb], nc = searched ? (int)d.n_child[nb] : 0;
        for (int i = lane; i < nc; i += 64) vis[d.action[nb + first + i]] = d.visits[nb + first + i];
    }
    const int rvis = d.visits[nb];
    const double rv = (!searched || rvis == 0) ? 0.0 : d.vsum[nb] / (double)rvis;
    wave_sync();

    // ---- policy target (mcts.py:270-279)
    const int mi = m.move_idx[g];
    const int ti = mi < m.temp_moves ? 0 : (mi < m.temp_mid_moves ? 1 : 2);
    if (m.temp[ti] <= 1e-2) {   // one-hot at the first maximum
        int bv = -1, bi = 0x7fffffff;
        for (int a = lane; a < A; a += 64)
            if (vis[a] > bv) { bv = vis[a]; bi = a; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(bv, o), oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        for (int a = lane; a < A; a += 64) p_lds[a] = a == bi ? 1.0 : 0.0;
        wave_sync();
    } else {
        const double* tab = m.powtab + (size_t)ti * m.pow_len;
        for (int a = lane; a < A; a += 64) p_lds[a] = tab[min(vis[a], m.pow_len - 1)];
        __syncthreads();
        const double sum = pw_sum_wave(p_lds, A, lane);   // ndarray.sum(): numpy's pairwise order
        __syncthreads();
        for (int a = lane; a < A; a += 64) p_lds[a] = sum <= 0 ? 1.0 / (double)A : p_lds[a] / sum;
        wave_sync();
    }
    // float32 copy and difficulty = max(0.01, 1 - max(pi32)) (train.py:336-337)
    float mx = 0.0f;
    for (int a = lane; a < A; a += 64) {
        const float f = (float)p_lds[a];
        if (m.collect) m.pi32[(size_t)g * A + a] = f;
        mx = fmaxf(mx, f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const double diff = fmax(0.01, 1.0 - (double)mx);
    // RandomState.choice(A, p=pi) given its uniform draw: cdf = cumsum(pi) (sequential); cdf /= cdf[-1];
    // index = searchsorted(cdf, u, side='right') = number of entries <= u
    if (lane == 0) {
        double c = 0.0;
        for (int a = 0; a < A; ++a) { c = c + p_lds[a]; cdf[a] = c; }
    }
    wave_sync();
    int action;
    {
        const double tot = cdf[A - 1], u = m.u[g];
        int cnt = 0;
        for (int a = lane; a < A; a += 64) cnt += (cdf[a] / tot <= u) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        action = cnt;
    }
    // ---- resignation (train.py:339-341) ends the game before the move is played
    int resigned = 0;
    if (m.resign_on && fabs(rv) >= m.resign_threshold && mi >= m.resign_min_move) resigned = rv < 0 ? -tp0 : tp0;
    if (!searched) action = -1;

    // ---- legality of the root's points: the sample's legal plane and the re-check of go.py:179-180
    analyze<PPL, false>(geo, b, d.zob);
    bool legal[PPL];
    legal_points<PPL>(geo, b, d.zob, s.tp, s.hash, seen_lds, s.ns, legal);
    if (m.collect) {   // to_tensor() of the searched state (go.py:265-304), binary planes packed most significant bit first
        const int words = (m.row_bytes + 3) / 4, H = d.H, np_ = d.np;
        for (int i = lane; i < words; i += 64) bitw[i] = 0;
        wave_sync();
        auto set_bit = [&](int plane, int p) {
            const int idx = plane * np_ + p;
            atomicOr(&bitw[idx >> 5], 1u << (8 * ((idx >> 3) & 3) + 7 - (idx & 7)));
        };
        const int8_t* hist = d.hist + (size_t)g * d.H * d.npad;
        const int take = min(d.sc[sc_idx(g, SC_HIST)], H);
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int p = geo.p[k];
            if (p < 0) continue;
            for (int i = 0; i < take; ++i) {
                const int v = hist[i * d.npad + p];
                if (v == s.tp) set_bit(i, p);
                else if (v == -s.tp) set_bit(H + i, p);
            }
            if (s.tp > 0) set_bit(2 * H, p);
            if (legal[k]) set_bit(2 * H + 1, p);
            if (p == s.last) set_bit(2 * H + 2, p);
        }
        wave_sync();
        uint8_t* dst = m.bits + (size_t)g * m.row_bytes;
        for (int i = lane; i < m.row_bytes; i += 64) dst[i] = (uint8_t)(bitw[i >> 2] >> (8 * (i & 3)));
        if (lane == 0) {
            const double max_pts = (double)(np_ > 1 ? np_ : 1);
            const int own = s.tp > 0 ? s.capb : s.capw, opp = s.tp > 0 ? s.capw : s.capb;
            m.caps[2 * g] = (float)((double)own / max_pts);
            m.caps[2 * g + 1] = (float)((double)opp / max_pts);
        }
    }

    // ---- GoState.play (go.py:159-203)
    int status = searched ? 0 : (int)AZ_ERR_STATE;
    const int a = resigned ? -1 : action;
    bool played = false;
    if (a >= 0) {
        bool ok = a <= d.np && s.ns < d.seen_cap;
        if (ok && a < d.np) {
            bool mine = false;
#pragma unroll
            for (int k = 0; k < PPL; ++k)
                if (geo.p[k] == a) mine = legal[k];
            ok = __any(mine);
        }
        if (!ok) status = AZ_ERR_ILLEGAL;
        else {
            const int ns0 = s.ns;
            sim_play<PPL>(d, lane, geo, b, seen_lds, nullptr, s, a);
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
                const uint64_t key = seen_lds[ns0];   // the seen list stands for a set (go.py:173/:199)
                bool dup = false;
                for (int i = 0; i < ns0; ++i) dup |= seen_lds[i] == key;
                if (!dup) { d.seen[(size_t)g * d.seen_cap + ns0] = key; sc[SC_SEEN] = ns0 + 1; }
                sc[SC_TP] = s.tp; sc[SC_PASSES] = s.passes; sc[SC_MOVES] = s.moves; sc[SC_CAPB] = s.capb;
                sc[SC_CAPW] = s.capw; sc[SC_LAST] = s.last;
                sc[SC_HIST] = min(sc[SC_HIST] + 1, d.H);
                d.hash[g] = s.hash;
            }
            played = true;
            if (d.rc_cap) {   // the played child's evaluation becomes the next root's (see k_play)
                int ci = -1;
                if (d.tree_live[g]) {
                    const int first = d.child_off[nb], nc = min((int)d.n_child[nb], d.rc_cap);
                    for (int c0 = 0; c0 < nc; c0 += 64) {
                        const int c = c0 + lane;
                        const bool hit = c < nc && d.action[nb + first + c] == a && d.rc_valid[(size_t)g * d.rc_cap + c];
                        const unsigned long long mm = __ballot(hit);
                        if (mm) { ci = c0 + __ffsll((long long)mm) - 1; break; }
                    }
                }
                if (ci >= 0) {
                    const float* src = d.rc_logits + ((size_t)g * d.rc_cap + ci) * d.A;
                    for (int i = lane; i < d.A; i += 64) d.root_logits[(size_t)g * d.A + i] = src[i];
                }
                __syncthreads();
                if (lane == 0) d.root_cached[g] = ci >= 0 ? 1 : 0;
            }
        }
    }
    // ---- is_terminal / winner of the new state (go.py:205-263)
    int over = resigned != 0, winner = resigned;
    if (played && sim_terminal<PPL>(d, s)) {
        double bl, wh;
        sim_outcome<PPL>(d, geo, b, s, &bl, &wh);
        over = 1;
        winner = bl > wh ? 1 : (wh > bl ? -1 : 0);
    }
    __threadfence_block();
    __syncthreads();
    // ---- a finished game leaves its slot to a fresh one (GoState.new_game, go.py:61-74); the host deals it an id or
    // switches the slot off before the next move
    const bool refill = m.flags[0] != 0;
    if (over && refill) {
        for (int i = lane; i < d.npad; i += 64) d.board[(size_t)g * d.npad + i] = 0;
        for (int i = lane; i < d.H * d.npad; i += 64) d.hist[(size_t)g * d.H * d.npad + i] = 0;
        if (lane == 0) {
            int* sc = d.sc + g * 8;
            sc[SC_TP] = 1; sc[SC_PASSES] = 0; sc[SC_MOVES] = 0; sc[SC_CAPB] = 0; sc[SC_CAPW] = 0;
            sc[SC_LAST] = -1; sc[SC_HIST] = 1; sc[SC_SEEN] = 1;
            d.hash[g] = 0;
            d.seen[(size_t)g * d.seen_cap] = d.zob[kZobSide + 0];
            if (d.rc_cap) d.root_cached[g] = 0;
        }
    }
    // ---- record for the host, then the root of the next search (mcts.py:171; k_begin for this game)
    if (lane == 0) {
        const long long* gs = d.gstats + (size_t)g * 8;
        rec[REC_ACTION] = action; rec[REC_RESIGNED] = resigned; rec[REC_STATUS] = status; rec[REC_OVER] = over;
        rec[REC_WINNER] = winner; rec[REC_TO_PLAY] = tp0; rec[REC_SIMS] = (int)gs[ST_SIMS]; rec[REC_ROWS] = (int)gs[ST_ROWS];
        rec[REC_ROWS_RAW] = (int)gs[ST_ROWS_RAW]; rec[REC_DEPTH_SUM] = (int)gs[ST_DEPTH_SUM]; rec[REC_MAX_DEPTH] = (int)gs[ST_MAX_DEPTH];
        rec[REC_MAX_NODES] = (int)gs[ST_MAX_NODES]; rec[REC_TERMINAL] = (int)gs[ST_TERMINAL]; rec[REC_ERR] = err; rec[REC_ON] = 1;
        rec[REC_WORDS - 1] = 0;
        m.recd[2 * g] = rv; m.recd[2 * g + 1] = diff;
    }
    __threadfence_block();
    __syncthreads();
    // (the per-game counters stay until the next root expansion clears them: az_mcts_stats still reports this search)
    if ((over && !refill) || status != 0) {   // the slot goes idle: no root for the next search
        if (lane == 0) { d.active[g] = 0; d.n_unique[g] = 0; }
        return;
    }
    if (d.rc_cap)
        for (int i = lane; i < d.rc_cap; i += 64) d.rc_valid[(size_t)g * d.rc_cap + i] = 0;
    if (lane == 0) {
        d.active[g] = 1;
        d.prior[nb] = 1.0; d.vsum[nb] = 0.0; d.visits[nb] = 0; d.action[nb] = -1;
        d.child_off[nb] = 0; d.n_child[nb] = 0;
        d.n_nodes[g] = 1;
        d.n_unique[g] = 1;
        if (d.rc_cap) d.tree_live[g] = 1;
    }
}

}  // namespace az

// =================================================================================================
struct az_lane {
    az_engine* e = nullptr;
    az_net* net = nullptr;
    az_rng* rng = nullptr;
    az_lane_config cfg;
    MoveDev m;
    void* feat = nullptr;
    int32_t layout = 0;
    float* logits = nullptr;
    float* values = nullptr;
    unsigned char *in_host = nullptr, *in_dev = nullptr, *out_host = nullptr, *out_dev = nullptr;
    size_t in_bytes = 0, out_bytes = 0;
    // offsets into the blocks
    size_t in_flags = 0, in_on = 0, in_mi = 0, in_u = 0, in_noise = 0;
    size_t out_rec = 0, out_recd = 0, out_cnt = 0, out_ctl = 0, out_pi = 0, out_bits = 0, out_caps = 0;
    int64_t root_rows = 0;       // network rows of the roots of the move in flight
    double t_rng = 0.0;          // host seconds spent drawing that move's random numbers
    double* powtab_dev = nullptr;
    int* old_root_count = nullptr;
    hipEvent_t done = nullptr;
    enum { IDLE = 0, PREPARING, READY, MOVING } state = IDLE;   // READY: root rows evaluated and their record is on the host
    int row_bytes = 0;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Row scan, leaf gather and network evaluation of the roots the engine's per-game state announces (n_unique = 1 for the
// games that search), then the copy that brings the move's record and the roots' legal-move counts to the host.
static int lane_enqueue_roots(az_lane* L, void* stream) {
    az_engine* e = L->e;
    hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, (hipStream_t)stream, e->d, 1);
    e->phase = PH_ROOT;
    e->sims_done = 0;
    e->rows_known = false;
    int rc = az_mcts_gather(e, L->feat, L->layout, stream);
    if (rc != AZ_OK) return rc;
    const int32_t max_rows = std::max(e->d.max_rows, e->d.G);
    rc = az_net_forward_indirect(L->net, L->feat, e->d.ctl, max_rows, L->logits, L->values, stream);
    if (rc != AZ_OK) return rc;
    HIP_TRY(hipMemcpyAsync(L->out_host, L->out_dev, L->out_bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipEventRecord(L->done, (hipStream_t)stream));
    return AZ_OK;
}

// One wave that waits on the 100 MHz real-time counter; `cap` bounds the loop whatever the counter does.
__global__ __launch_bounds__(64) void k_wait(unsigned long long ticks, int cap, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && n < cap) {
        __builtin_amdgcn_s_sleep(32);
        n += 1;
    }
    if (sink && threadIdx.x == 0 && n < 0) *sink = n;
}

extern "C" {

int az_streams_overlap(void* stream_a, void* stream_b, int32_t microseconds, int32_t device, int32_t* overlap) {
    if (!overlap) return fail(AZ_ERR_ARG, "az_streams_overlap: null argument");
    HIP_TRY(hipSetDevice(device));
    const int us = microseconds < 20 ? 20 : (microseconds > 2000 ? 2000 : microseconds);
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    const unsigned long long ticks = (unsigned long long)us * 100ull;
    const int cap = us * 40;      // a pass of the loop takes well over 25 ns
    auto run = [&](bool both, double& seconds) -> hipError_t {
        hipError_t e;
        if ((e = hipStreamSynchronize(a)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(b)) != hipSuccess) return e;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, a, ticks, cap, (int*)nullptr);
        if (both) hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, b, ticks, cap, (int*)nullptr);
        if ((e = hipStreamSynchronize(a)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(b)) != hipSuccess) return e;
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return hipGetLastError();
    };
    double one = 1e9, two = 1e9, t = 0.0;
    HIP_TRY(run(false, t));       // first launch of the kernel: not timed
    for (int rep = 0; rep < 3; ++rep) {
        HIP_TRY(run(false, t)); one = std::min(one, t);
        HIP_TRY(run(true, t)); two = std::min(two, t);
    }
    *overlap = two < 1.5 * one ? 1 : 0;
    return AZ_OK;
}

int az_lane_create(az_engine* e, az_net* net, az_rng* rng, const az_lane_config* cfg, const double* pow_tables, int32_t pow_len,
                   void* feat_dev, int32_t layout, float* logits_dev, float* values_dev, az_lane** out) {
    if (!e || !net || !rng || !cfg || !pow_tables || !feat_dev || !logits_dev || !values_dev || !out) return fail(AZ_ERR_ARG, "null argument");
    if (!e->async_mode) return fail(AZ_ERR_STATE, "az_lane_create needs asynchronous mode (az_mcts_set_async)");
    if (pow_len < e->d.sims + 1) return fail(AZ_ERR_ARG, "az_lane_create: the power tables must cover 0..simulations visits");
    if (az_net_indirect_passes(net, std::max(e->d.max_rows, e->d.G)) != 1)
        return fail(AZ_ERR_STATE, "az_lane_create: the network cannot evaluate the row bound in one pass (az_net_reserve first)");
    HIP_TRY(hipSetDevice(e->cfg.device));
    az_lane* L = new az_lane();
    L->e = e; L->net = net; L->rng = rng; L->cfg = *cfg;
    L->feat = feat_dev; L->layout = layout; L->logits = logits_dev; L->values = values_dev;
    const Dev& d = e->d;
    const size_t G = d.G, A = d.A;
    L->row_bytes = ((2 * d.H + 3) * d.np + 7) / 8;
    size_t o = 0;
    L->in_flags = o; o = align_up(o + 16, 16);
    L->in_on = o; o = align_up(o + G, 16);
    L->in_mi = o; o = align_up(o + G * 4, 16);
    L->in_u = o; o = align_up(o + G * 8, 16);
    L->in_noise = o; o = align_up(o + G * A * 8, 16);
    L->in_bytes = o;
    o = 0;
    L->out_rec = o; o = align_up(o + G * REC_WORDS * 4, 16);
    L->out_recd = o; o = align_up(o + G * 16, 16);
    L->out_cnt = o; o = align_up(o + G * 4, 16);
    L->out_ctl = o; o = align_up(o + 16, 16);
    if (cfg->collect_samples) {
        L->out_pi = o; o = align_up(o + G * A * 4, 16);
        L->out_bits = o; o = align_up(o + G * L->row_bytes, 16);
        L->out_caps = o; o = align_up(o + G * 8, 16);
    }
    L->out_bytes = o;
#define LANE_TRY(x)                                                                             \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            az_lane_destroy(L);                                                                 \
            return fail(AZ_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));            \
        }                                                                                       \
    } while (0)
    LANE_TRY(hipHostMalloc((void**)&L->in_host, L->in_bytes));
    LANE_TRY(hipHostMalloc((void**)&L->out_host, L->out_bytes));
    LANE_TRY(hipMalloc((void**)&L->in_dev, L->in_bytes));
    LANE_TRY(hipMalloc((void**)&L->out_dev, L->out_bytes));
    LANE_TRY(hipMalloc((void**)&L->powtab_dev, (size_t)3 * pow_len * 8));
    LANE_TRY(hipMemcpy(L->powtab_dev, pow_tables, (size_t)3 * pow_len * 8, hipMemcpyHostToDevice));
    LANE_TRY(hipMemset(L->in_dev, 0, L->in_bytes));
    LANE_TRY(hipMemset(L->out_dev, 0, L->out_bytes));
    LANE_TRY(hipEventCreateWithFlags(&L->done, hipEventDisableTiming));
#undef LANE_TRY
    memset(L->in_host, 0, L->in_bytes);
    memset(L->out_host, 0, L->out_bytes);
    MoveDev& m = L->m;
    memset(&m, 0, sizeof(m));
    m.flags = (const int*)(L->in_dev + L->in_flags);
    m.slot_on = L->in_dev + L->in_on; m.move_idx = (const int*)(L->in_dev + L->in_mi);
    m.u = (const double*)(L->in_dev + L->in_u); m.noise = (const double*)(L->in_dev + L->in_noise);
    m.rec = (int*)(L->out_dev + L->out_rec); m.recd = (double*)(L->out_dev + L->out_recd);
    if (cfg->collect_samples) {
        m.pi32 = (float*)(L->out_dev + L->out_pi); m.bits = L->out_dev + L->out_bits; m.caps = (float*)(L->out_dev + L->out_caps);
    }
    m.powtab = L->powtab_dev; m.pow_len = pow_len;
    m.temp[0] = cfg->temp_start; m.temp[1] = cfg->temp_mid; m.temp[2] = cfg->temp_end;
    m.temp_moves = cfg->temp_moves; m.temp_mid_moves = cfg->temp_mid_moves;
    m.resign_threshold = cfg->resign_threshold; m.resign_on = cfg->disable_resign ? 0 : 1; m.resign_min_move = cfg->resign_min_move;
    m.row_bytes = L->row_bytes; m.collect = cfg->collect_samples ? 1 : 0;
    // the leaf gather of root rows leaves each root's number of legal actions where the record copy picks it up
    L->old_root_count = e->d.root_count;
    e->d.root_count = (int*)(L->out_dev + L->out_cnt);
    e->d.ctl_mirror = (int*)(L->out_dev + L->out_ctl);
    if (e->rounds_exec) { (void)hipGraphExecDestroy(e->rounds_exec); e->rounds_exec = nullptr; }   // captured with the old device struct
    *out = L;
    return AZ_OK;
}

int az_lane_destroy(az_lane* L) {
    if (!L) return AZ_OK;
    (void)hipSetDevice(L->e->cfg.device);
    (void)hipDeviceSynchronize();
    if (L->e->d.root_count == (int*)(L->out_dev + L->out_cnt)) {
        L->e->d.root_count = L->old_root_count;
        L->e->d.ctl_mirror = nullptr;
        if (L->e->rounds_exec) { (void)hipGraphExecDestroy(L->e->rounds_exec); L->e->rounds_exec = nullptr; }
    }
    if (L->done) (void)hipEventDestroy(L->done);
    if (L->in_host) (void)hipHostFree(L->in_host);
    if (L->out_host) (void)hipHostFree(L->out_host);
    if (L->in_dev) (void)hipFree(L->in_dev);
    if (L->out_dev) (void)hipFree(L->out_dev);
    if (L->powtab_dev) (void)hipFree(L->powtab_dev);
    delete L;
    return AZ_OK;
}

int az_lane_slots(az_lane* L, uint8_t** slot_on_host, int32_t** move_idx_host) {
    if (!L || !slot_on_host || !move_idx_host) return fail(AZ_ERR_ARG, "null argument");
    *slot_on_host = L->in_host + L->in_on;
    *move_idx_host = (int32_t*)(L->in_host + L->in_mi);
    return AZ_OK;
}

int az_lane_prepare(az_lane* L, void* stream) {
    if (!L) return fail(AZ_ERR_ARG, "null lane");
    if (L->state == az_lane::MOVING) return fail(AZ_ERR_STATE, "az_lane_prepare: a move is in flight (az_lane_finish first)");
    az_engine* e = L->e;
    HIP_TRY(hipMemcpyAsync(L->in_dev + L->in_on, L->in_host + L->in_on, (size_t)e->d.G, hipMemcpyHostToDevice, (hipStream_t)stream));
    hipLaunchKernelGGL(k_begin, dim3((e->d.G + 63) / 64), dim3(64), 0, (hipStream_t)stream, e->d, L->m.slot_on);
    int rc = lane_enqueue_roots(L, stream);
    if (rc != AZ_OK) return rc;
    rc = check_launch("az_lane_prepare");
    if (rc != AZ_OK) return rc;
    L->state = az_lane::PREPARING;
    return AZ_OK;
}

int az_lane_launch(az_lane* L, int32_t refill, void* stream) {
    if (!L) return fail(AZ_ERR_ARG, "null lane");
    if (L->state != az_lane::PREPARING && L->state != az_lane::READY)
        return fail(AZ_ERR_STATE, "az_lane_launch: no prepared roots (az_lane_prepare, or az_lane_finish of the previous move, comes first)");
    az_engine* e = L->e;
    const Dev& d = e->d;
    if (L->state == az_lane::PREPARING) HIP_TRY(hipEventSynchronize(L->done));
    L->state = az_lane::READY;
    // the game's draws in upstream's order: the root's Dirichlet sample (mcts.py:122), then the uniform of the move choice (mcts.py:281)
    const uint8_t* on = L->in_host + L->in_on;
    int32_t* counts = (int32_t*)(L->out_host + L->out_cnt);
    for (int g = 0; g < d.G; ++g) {
        if (!on[g]) counts[g] = 0;
        else if (counts[g] <= 0 || counts[g] > d.A) return fail(AZ_ERR_STATE, "az_lane_launch: a slot was switched on after its lane's roots were prepared (az_lane_prepare first)");
    }
    L->root_rows = ((const int32_t*)(L->out_host + L->out_ctl))[0];   // rows the network evaluated for these roots
    ((int32_t*)(L->in_host + L->in_flags))[0] = refill ? 1 : 0;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = az_rng_dirichlet(L->rng, counts, L->cfg.dirichlet_alpha, (double*)(L->in_host + L->in_noise), d.A, L->cfg.rng_threads);
    if (rc != AZ_OK) return rc;
    rc = az_rng_random_sample(L->rng, on, (double*)(L->in_host + L->in_u));
    if (rc != AZ_OK) return rc;
    L->t_rng = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(hipMemcpyAsync(L->in_dev, L->in_host, L->in_bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    e->phase = PH_ROOT;
    e->rows_known = false;
    e->root_enable = L->m.slot_on;     // a slot the caller switched off since its root row was gathered sits the search out
    rc = az_mcts_expand(e, L->logits, L->values, L->m.noise, stream);
    e->root_enable = nullptr;
    if (rc < 0) return rc;
    int replayed = 0;
    if (rc == 0) {
        rc = az_mcts_run_rounds(e, L->net, L->feat, L->layout, L->logits, L->values, L->cfg.use_graph, stream);
        if (rc < 0) return rc;
        replayed = rc;
    }
    DISPATCH_PPL(e, k_move_tail, d.G, tail_shmem(d, L->row_bytes), stream, e->d, L->m);
    rc = lane_enqueue_roots(L, stream);
    if (rc != AZ_OK) return rc;
    rc = check_launch("az_lane_launch");
    if (rc != AZ_OK) return rc;
    L->state = az_lane::MOVING;
    return replayed;
}

int az_lane_finish(az_lane* L, az_lane_result* r) {
    if (!L || !r) return fail(AZ_ERR_ARG, "null argument");
    if (L->state != az_lane::MOVING) return fail(AZ_ERR_STATE, "az_lane_finish: no move in flight");
    HIP_TRY(hipEventSynchronize(L->done));
    L->state = az_lane::READY;
    const Dev& d = L->e->d;
    const int G = d.G, A = d.A;
    const int* rec = (const int*)(L->out_host + L->out_rec);
    const double* recd = (const double*)(L->out_host + L->out_recd);
    int32_t* mi = (int32_t*)(L->in_host + L->in_mi);
    for (int i = 0; i < 8; ++i) r->counters[i] = 0;
    r->counters[ST_ROWS] = r->counters[ST_ROWS_RAW] = L->root_rows;
    r->host_seconds_rng = L->t_rng;
    r->n_over = 0;
    const int err = rec[REC_ERR];
    if (err == AZ_ERR_NOCHILD) return fail(AZ_ERR_NOCHILD, "No child selected in MCTS");
    if (err == AZ_ERR_CAPACITY) return fail(AZ_ERR_CAPACITY, "search tree node capacity exceeded");
    for (int g = 0; g < G; ++g) {
        const int* q = rec + (size_t)g * REC_WORDS;
        if (r->to_play) r->to_play[g] = q[REC_TO_PLAY];
        if (r->actions) r->actions[g] = q[REC_ON] ? q[REC_ACTION] : -1;
        if (r->root_value) r->root_value[g] = q[REC_ON] ? recd[2 * g] : 0.0;
        if (r->difficulty) r->difficulty[g] = q[REC_ON] ? recd[2 * g + 1] : 0.0;
        if (!q[REC_ON]) continue;
        if (q[REC_STATUS] == AZ_ERR_ILLEGAL) {
            char buf[96];
            snprintf(buf, sizeof(buf), "Illegal move: (%d, %d)", q[REC_ACTION] / d.n, q[REC_ACTION] % d.n);
            return fail(AZ_ERR_ILLEGAL, buf);
        }
        if (q[REC_STATUS] != 0) return fail(AZ_ERR_STATE, "az_lane: a running game's root was not searched");
        r->counters[ST_SIMS] += q[REC_SIMS]; r->counters[ST_ROWS] += q[REC_ROWS]; r->counters[ST_ROWS_RAW] += q[REC_ROWS_RAW];
        r->counters[ST_DEPTH_SUM] += q[REC_DEPTH_SUM]; r->counters[ST_TERMINAL] += q[REC_TERMINAL];
        r->counters[ST_MAX_DEPTH] = std::max<int64_t>(r->counters[ST_MAX_DEPTH], q[REC_MAX_DEPTH]);
        r->counters[ST_MAX_NODES] = std::max<int64_t>(r->counters[ST_MAX_NODES], q[REC_MAX_NODES]);
        r->counters[ST_BATCHES] += 1;   // moves searched
        if (q[REC_OVER]) {
            if (r->over_slot) r->over_slot[r->n_over] = g;
            if (r->over_winner) r->over_winner[r->n_over] = q[REC_WINNER];
            r->n_over += 1;
            mi[g] = 0;
        } else {
            mi[g] += 1;
        }
    }
    if (L->cfg.collect_samples) {
        if (r->pi32) memcpy(r->pi32, L->out_host + L->out_pi, (size_t)G * A * 4);
        if (r->bits) memcpy(r->bits, L->out_host + L->out_bits, (size_t)G * L->row_bytes);
        if (r->caps) memcpy(r->caps, L->out_host + L->out_caps, (size_t)G * 8);
    }
    return AZ_OK;
}

}  // extern "C"
