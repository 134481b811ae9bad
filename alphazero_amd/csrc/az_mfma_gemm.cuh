// Shared pieces of the bf16 MFMA network kernels for gfx950: fragment types, the LDS weight-tile pipeline and
// the implicit-GEMM convolution loops used by the fused tower (az_net_mfma.hip) and by the layer-by-layer path
// (az_net_layers.hip), plus the host-side weight packing that matches the fragment layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace az {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
// Makes a value opaque to the optimiser at this point, so address arithmetic derived from it is
// recomputed where it is used instead of being hoisted out of the layer loop and spilled.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint16_t to_bf16(float a) {
    __bf16 v = (__bf16)a;
    return __builtin_bit_cast(uint16_t, v);
}

// Asynchronous global -> LDS copy of one packed weight tile (nvec 16-byte vectors, a multiple of 64):
// every wave-instruction moves 1 KiB (64 lanes x 16 B) straight into LDS, no staging registers.
template <int NW>
__device__ __forceinline__ void stage_tile(const uint4* __restrict__ src, char* dst, int nvec, int lane, int wave) {
    const int nchunks = nvec >> 6;
    for (int c = wave; c < nchunks; c += NW) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
    }
}

// Per-lane coordinates inside the workgroup tile. Every phase of the kernel re-derives them from an
// opaque copy of threadIdx.x, so nothing computed from them is hoisted across the big layer loop
// (the optimiser otherwise precomputes hundreds of addresses there and spills them).
// A wave owns 16*MI positions (rows of the activation image) and 16*NI output channels. In every
// 16x16 MFMA tile the lane holds position lr and the four channels 4*kq..4*kq+3 of the tile; tiles
// 2p and 2p+1 are packed so that together they give the lane 8 consecutive channels (see pack_conv).
#define AZ_LANE_GEO(MI_, NI_, WN_)                                                                \
    const int t_ = opaque((int)threadIdx.x);                                                      \
    const int lane = t_ & 63, wave = t_ >> 6, wm = wave / (WN_), wn = wave - wm * (WN_);          \
    const int lr = lane & 15, kq = lane >> 4;                                                     \
    const int row_base = wm * (16 * (MI_)), col0 = wn * ((NI_) * 16);                             \
    (void)row_base; (void)col0; (void)lr; (void)kq; (void)wave

struct TileGeo {   // wave-uniform facts about the workgroup's tile
    int row0;      // GLOBAL_ROWS only: index of the tile's first row in the activation array
    int n, np, vrows;
    unsigned recip_np, recip_n;
};

// LDS byte offsets (from `act`) of the lane's activation fragments for k-step (tap, cc): position
// row + tap shift, 8 channels at 32*cc + 8*kq. Taps that leave the board, rows past the valid ones and
// padding k-steps (tap >= ntaps) read the zero row.
template <int MI>
__device__ __forceinline__ void frag_offsets(int (&out)[MI], const int (&ry)[MI], const int (&rx)[MI], const int (&roff)[MI], int tap, int cc,
                                             int ntaps, bool three, int n, int in_stride, int zoff) {
    const int dy = three ? tap / 3 - 1 : 0, dx = three ? tap - (tap / 3) * 3 - 1 : 0;
    const int shift = (dy * n + dx) * in_stride + cc * 64;
    const bool tap_ok = tap < ntaps;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int yy = ry[mi] + dy, xx = rx[mi] + dx;
        const bool ok = tap_ok & ((unsigned)yy < (unsigned)n) & ((unsigned)xx < (unsigned)n);
        out[mi] = (ok ? roff[mi] + shift : zoff + cc * 64);
    }
}

// Implicit GEMM acc[mi][ni] += W(16 channels x K) * X(K x 16 positions) over `nksteps` (even) k-steps of
// 32 input channels, `kpt` k-steps per 3x3 tap. Weight tiles (two k-steps each) are streamed from `wg`
// through the two LDS buffers at bbuf. The workgroup barrier of a tile sits between its two k-steps:
// by then every wave holds the tile's last fragments in registers, so the buffer can be refilled (two
// tiles ahead) and the MFMAs of the second k-step run while the first fragments of the next tile are
// fetched -- the matrix pipe never waits for LDS at a tile boundary.
template <int MI, int NI, int WM, int WN, int NCOLS, bool SAME_TAP>
__device__ __forceinline__ void gemm_taps(f32x4 (&acc)[MI][NI], const uint4* __restrict__ wg, int nksteps, int kpt, int ntaps, bool three,
                                          const char* act, int in_stride, const char* zrow, char* bbuf, const TileGeo& g) {
    constexpr int NW = WM * WN;
    constexpr int N = NCOLS;
    constexpr int tile_bytes = 64 * N * 2, nvec = tile_bytes / 16, kstep_bytes = tile_bytes / 2;
    constexpr int NCHUNK = nvec / 64;                 // 1 KiB pieces per weight tile
    constexpr int PW = (NCHUNK + NW - 1) / NW;        // pieces per wave
    AZ_LANE_GEO(MI, NI, WN);
    const int ntiles = nksteps >> 1;
    stage_tile<NW>(wg, bbuf, nvec, lane, wave);
    if (ntiles > 1) stage_tile<NW>(wg + nvec, bbuf + tile_bytes, nvec, lane, wave);
    // Tiles 2.. go through registers: fetched two tiles ahead, written to LDS one tile ahead.
    constexpr bool RAGGED = (NCHUNK % NW) != 0;       // the last piece round does not cover every wave
    uint4 stg[PW];
    const uint4* wsrc = wg + lane + wave * 64;        // this lane's vector of piece `wave` of tile 0
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        stg[i] = make_uint4(0, 0, 0, 0);
        if (ntiles > 2 && (!RAGGED || wave + i * NW < NCHUNK)) stg[i] = wsrc[(size_t)2 * nvec + i * NW * 64];
    }
    int ry[MI], rx[MI], roff[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = row_base + mi * 16 + lr;
        const int b = (row * g.recip_np) >> 16, p = row - b * g.np;
        ry[mi] = (p * g.recip_n) >> 16;
        rx[mi] = p - ry[mi] * g.n;
        if (row >= g.vrows) ry[mi] = -64;   // fails every bounds check
        roff[mi] = row * in_stride + kq * 16;
    }
    const int zoff = (int)(zrow - act) + kq * 16;
    const int woff = ((wn * NI) * 64 + lane) * 16;   // this wave's first channel tile inside a k-step image
    const int soff = (wave * 64 + lane) * 16;        // where this lane's staged vector of piece `wave` goes
    int tap = 0, cc = 0;   // cursor: the next k-step whose offsets are to be computed
    int ap0[MI], ap1[MI];
    frag_offsets<MI>(ap0, ry, rx, roff, tap, cc, ntaps, three, g.n, in_stride, zoff);
    if (SAME_TAP) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ap1[mi] = ap0[mi] + 64;
        cc += 2;
    } else {
        if (++cc == kpt) { cc = 0; ++tap; }
        frag_offsets<MI>(ap1, ry, rx, roff, tap, cc, ntaps, three, g.n, in_stride, zoff);
        ++cc;
    }
    if (cc >= kpt) { cc = 0; ++tap; }
    // tiles 0 and 1 have landed; the PW register fetches issued after them may still fly (the counter retires in order)
    if (!RAGGED && ntiles > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 a0[MI], b0[NI], a1[MI], b1[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a0[mi] = *(const bf16x8*)(act + ap0[mi]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b0[ni] = *(const bf16x8*)(bbuf + woff + ni * 1024);
    for (int t = 0; t + 1 < ntiles; ++t) {
        char* X = bbuf + (t & 1) * tile_bytes;
        const char* Y = bbuf + ((t + 1) & 1) * tile_bytes;
        // ---- first k-step: fetch the second k-step's fragments meanwhile
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a1[mi] = *(const bf16x8*)(act + ap1[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b1[ni] = *(const bf16x8*)(X + kstep_bytes + woff + ni * 1024);
        frag_offsets<MI>(ap0, ry, rx, roff, tap, cc, ntaps, three, g.n, in_stride, zoff);   // next tile
        if (SAME_TAP) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) ap1[mi] = ap0[mi] + 64;
            cc += 2;
        } else {
            if (++cc == kpt) { cc = 0; ++tap; }
            frag_offsets<MI>(ap1, ry, rx, roff, tap, cc, ntaps, three, g.n, in_stride, zoff);
            ++cc;
        }
        if (cc >= kpt) { cc = 0; ++tap; }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI * NI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < MI + NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // ---- tile boundary: X is free once every wave has its fragments; tile t+1 is complete in Y
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntiles) {   // tile t+2 (fetched a tile ago) -> X; start fetching tile t+3
#pragma unroll
            for (int i = 0; i < PW; ++i)
                if (!RAGGED || wave + i * NW < NCHUNK) *(uint4*)(X + soff + i * NW * 1024) = stg[i];
            if (t + 3 < ntiles) {
#pragma unroll
                for (int i = 0; i < PW; ++i)
                    if (!RAGGED || wave + i * NW < NCHUNK) stg[i] = wsrc[(size_t)(t + 3) * nvec + i * NW * 64];
            }
        }
        // ---- second k-step: fetch the first fragments of the next tile meanwhile
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a0[mi] = *(const bf16x8*)(act + ap0[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b0[ni] = *(const bf16x8*)(Y + woff + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI * NI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < MI + NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
    {   // last tile
        const char* X = bbuf + ((ntiles - 1) & 1) * tile_bytes;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a1[mi] = *(const bf16x8*)(act + ap1[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b1[ni] = *(const bf16x8*)(X + kstep_bytes + woff + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI * NI; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < MI + NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
    }
}

// The same GEMM for layers whose taps span an even number KPT of k-steps (every layer with C >= 64): the
// tile loop is unrolled over the KPT/2 tiles of a tap, so the channel offsets are instruction immediates
// and the only per-tap vector work is one select per fragment row between the shifted position and the
// zero row (from a per-lane 9-bit mask of the taps that stay on the board).
// GLOBAL_ROWS: the activations are an array of g.vrows rows in global memory (any number of boards) instead of
// the workgroup's LDS image; the tile's rows are g.row0 .. g.row0+255 and the fragment reads become global loads.
template <int MI, int NI, int WM, int WN, int NCOLS, int KPT, bool GLOBAL_ROWS = false>
__device__ __forceinline__ void gemm_conv(f32x4 (&acc)[MI][NI], const uint4* __restrict__ wg, int ntaps, bool three, const char* act,
                                          int in_stride, const char* zrow, char* bbuf, const TileGeo& g) {
    constexpr int NW = WM * WN;
    constexpr int TPT = KPT / 2;                      // tiles per tap
    constexpr int tile_bytes = 64 * NCOLS * 2, nvec = tile_bytes / 16, kstep_bytes = tile_bytes / 2;
    constexpr int NCHUNK = nvec / 64;                 // 1 KiB pieces per weight tile
    constexpr int PW = (NCHUNK + NW - 1) / NW;        // pieces per wave
    constexpr bool RAGGED = (NCHUNK % NW) != 0;       // the last piece round does not cover every wave
    static_assert(KPT % 2 == 0, "taps must cover whole tiles");
    AZ_LANE_GEO(MI, NI, WN);
    const int ntiles = ntaps * TPT;
    stage_tile<NW>(wg, bbuf, nvec, lane, wave);
    if (ntiles > 1) stage_tile<NW>(wg + nvec, bbuf + tile_bytes, nvec, lane, wave);
    // Tiles 2.. go through registers: fetched two tiles ahead, written to LDS one tile ahead.
    uint4 stg[PW];
    const uint4* wsrc = wg + lane + wave * 64;        // this lane's vector of piece `wave` of tile 0
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        stg[i] = make_uint4(0, 0, 0, 0);
        if (ntiles > 2 && (!RAGGED || wave + i * NW < NCHUNK)) stg[i] = wsrc[(size_t)2 * nvec + i * NW * 64];
    }
    int roff[MI], vmask[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        int row = row_base + mi * 16 + lr, p, y;
        if (GLOBAL_ROWS) {
            row += g.row0;
            p = row % g.np;
            y = p / g.n;
        } else {
            p = row - (int)((row * g.recip_np) >> 16) * g.np;
            y = (p * g.recip_n) >> 16;
        }
        const int x = p - y * g.n;
        int m = 0;
        if (row < g.vrows) {
            if (three) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                    if ((unsigned)yy < (unsigned)g.n && (unsigned)xx < (unsigned)g.n) m |= 1 << tp;
                }
            } else {
                m = 1;
            }
        }
        vmask[mi] = m;
        roff[mi] = row * in_stride + kq * 16;
    }
    const int zbase = (int)(zrow - act);
    const int woff = ((wn * NI) * 64 + lane) * 16;   // this wave's first channel tile inside a k-step image
    const int soff = (wave * 64 + lane) * 16;        // where this lane's staged vector of piece `wave` goes
    int apc[MI], apn[MI];                            // fragment offsets of the current / next tap (channel offset 0)
    auto tap_offsets = [&](int (&out)[MI], int tap) {
        const int dy = three ? tap / 3 - 1 : 0, dx = three ? tap - (tap / 3) * 3 - 1 : 0;
        const int shift = (dy * g.n + dx) * in_stride;
        const int bit = tap < ntaps ? 1 << tap : 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // A lane whose tap leaves the board reads zeros from the spot of the zero region that has the banks of
            // the address it would have read (the region is 256-byte aligned), so the redirection adds no conflict.
            const int reg = roff[mi] + shift;
            out[mi] = (vmask[mi] & bit) ? reg : zbase + (reg & 255);
        }
    };
    tap_offsets(apc, 0);
    // tiles 0 and 1 have landed; the PW register fetches issued after them may still fly (the counter retires in order)
    if (!RAGGED && ntiles > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 a0[MI], b0[NI], a1[MI], b1[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a0[mi] = *(const bf16x8*)(act + apc[mi]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b0[ni] = *(const bf16x8*)(bbuf + woff + ni * 1024);
    int t = 0;
    for (int tap = 0; tap < ntaps; ++tap) {
        tap_offsets(apn, tap + 1);
#pragma unroll
        for (int u = 0; u < TPT; ++u, ++t) {
            const int par = (t & 1) * tile_bytes;
            char* X = bbuf + par;
            const char* Y = bbuf + (tile_bytes - par);
            const char* xb = X + woff;
            const char* yb = Y + woff;
            // ---- first k-step: fetch the second k-step's fragments meanwhile
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a1[mi] = *(const bf16x8*)(act + apc[mi] + (u * 128 + 64));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b1[ni] = *(const bf16x8*)(xb + kstep_bytes + ni * 1024);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < (GLOBAL_ROWS ? 0 : MI * NI); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < MI + NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // ---- tile boundary: X is free once every wave has its fragments; tile t+1 is complete in Y
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            // tile t+2 (fetched a tile ago) -> X, then fetch tile t+3 into the same registers; both are spread over
            // the MFMAs below (issued together, 24 KiB of LDS stores and of fetches queue up in the CU's store and
            // vector-memory paths and hold the in-order instruction streams behind them). Past the last tiles the
            // last tile is fetched and stored again: X is dead by then, the copies are never read.
            {
                const uint4* src = wsrc + (size_t)min(t + 3, ntiles - 1) * nvec;
#pragma unroll
                for (int i = 0; i < PW; ++i)
                    if (!RAGGED || wave + i * NW < NCHUNK) {
                        *(uint4*)(X + soff + i * NW * 1024) = stg[i];
                        stg[i] = src[i * NW * 64];
                    }
            }
            // ---- second k-step: fetch the first fragments of the next tile meanwhile (after the last tile: of
            // the zero row and a stale buffer, never used)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a0[mi] = u + 1 < TPT ? *(const bf16x8*)(act + apc[mi] + (u + 1) * 128) : *(const bf16x8*)(act + apn[mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b0[ni] = *(const bf16x8*)(yb + ni * 1024);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
            // the PW global fetches are spread over the MFMAs: issued together they would queue up in the
            // CU's vector-memory path and hold the in-order instruction streams behind them
#pragma unroll
            for (int i = 0; i < (GLOBAL_ROWS ? 0 : MI * NI); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < MI + NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i % (MI * NI / PW) == 1 && i / (MI * NI / PW) < PW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (i % (MI * NI / PW) == MI * NI / PW / 2 + 1 && i / (MI * NI / PW) < PW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) apc[mi] = apn[mi];
    }
}

template <int V> struct RingBase { static constexpr int value = V; };

// The same convolution with the weight fragments read straight from the packed global image into registers (the
// image is stored in fragment order, so a fragment is one coalesced 1 KiB load; the waves that share output channels
// hit the same lines in the CU's vector cache). Nothing is staged through LDS and the activations are read-only
// during the loop, so there is no workgroup barrier inside: every wave runs its MFMAs at its own pace, with the
// weight fragments of the next DEPTH k-steps in flight.
template <int MI, int NI, int WM, int WN, int NCOLS, int KPT, int DEPTH>
__device__ __forceinline__ void gemm_conv_gw(f32x4 (&acc)[MI][NI], const uint4* __restrict__ wg, int ntaps, bool three, const char* act,
                                             int in_stride, const char* zrow, const TileGeo& g) {
    AZ_LANE_GEO(MI, NI, WN);
    const int nks = ntaps * KPT;
    int roff[MI], vmask[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = row_base + mi * 16 + lr;
        const int p = row - (int)((row * g.recip_np) >> 16) * g.np;
        const int y = (p * g.recip_n) >> 16, x = p - y * g.n;
        int m = 0;
        if (row < g.vrows) {
            if (three) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                    if ((unsigned)yy < (unsigned)g.n && (unsigned)xx < (unsigned)g.n) m |= 1 << tp;
                }
            } else {
                m = 1;
            }
        }
        vmask[mi] = m;
        roff[mi] = row * in_stride + kq * 16;
    }
    const int zbase = (int)(zrow - act);
    // k-step s, tile ni of this wave: byte (s * (NCOLS/16) + wn * NI + ni) * 1024 + lane * 16 of the packed image, read
    // through a buffer descriptor: per-lane offset fixed for the whole loop, k-step offset in a scalar register, tile
    // offset an immediate, so a k-step's loads need no vector address arithmetic
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, nks * (NCOLS / 16) * 1024, 0x00020000);
    const int wvoff = ((wn * NI) * 64 + lane) * 16;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    auto wfrag = [&](int s, int ni) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + ni * 1024, min(s, nks - 1) * (NCOLS / 16) * 1024, 0);
        return __builtin_bit_cast(bf16x8, v);
    };
    auto tap_offsets = [&](int (&out)[MI], int tap) {
        const int dy = three ? tap / 3 - 1 : 0, dx = three ? tap - (tap / 3) * 3 - 1 : 0;
        const int shift = (dy * g.n + dx) * in_stride;
        const int bit = tap < ntaps ? 1 << tap : 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int reg = roff[mi] + shift;
            out[mi] = (vmask[mi] & bit) ? reg : zbase + (reg & 255);
        }
    };
    int apc[MI], apn[MI];
    tap_offsets(apc, 0);
    // One set of activation fragments, refilled as it drains: the NI MFMAs that consume fragment mi are followed at once by
    // the LDS read of the next k-step's fragment mi into the same registers, so every read has (MI-1)*NI MFMAs of cover
    // and the registers a second set would take stay free. The issue order is pinned (NI MFMAs, one LDS read, and one of
    // the k-step's NI weight loads after each of the first NI groups): left to itself the scheduler sinks the reads to
    // the end of the k-step and the first MFMA of the next one waits for LDS.
    bf16x8 a[MI], b[DEPTH + 1][NI];
    constexpr int NB = DEPTH + 1;
    // the ring of weight fragments keeps static indices when its period divides a tap's k-steps, or spans exactly two taps
    // (64 channels: two k-steps per tap under a ring of four); the taps then go in pairs, and the odd last one starts at 0 again
    static_assert(KPT % 2 == 0 && (KPT % NB == 0 || NB == 2 * KPT), "the fragment ring must keep static indices across taps");
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[d][ni] = wfrag(d, ni);
    __syncthreads();   // the activation image is complete
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *(const bf16x8*)(act + apc[mi]);
    int s = 0;
    auto one_tap = [&](int tap, auto base_c) {
        constexpr int BASE = decltype(base_c)::value;   // ring position of the tap's first k-step
        tap_offsets(apn, tap + 1);
#pragma unroll
        for (int u = 0; u < KPT; ++u, ++s) {
            const int bc = (BASE + u) % NB, bn = (BASE + u + DEPTH) % NB;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[bn][ni] = wfrag(s + DEPTH, ni);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[bc][ni], a[mi], acc[mi][ni], 0, 0, 0);
                a[mi] = u + 1 < KPT ? *(const bf16x8*)(act + apc[mi] + (u + 1) * 64) : *(const bf16x8*)(act + apn[mi]);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (mi < NI) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the loads of later k-steps from being hoisted here (registers)
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) apc[mi] = apn[mi];
    };
    if constexpr (KPT % NB == 0) {
        for (int tap = 0; tap < ntaps; ++tap) one_tap(tap, RingBase<0>{});
    } else {
        int tap = 0;
        for (; tap + 2 <= ntaps; tap += 2) {
            one_tap(tap, RingBase<0>{});
            one_tap(tap + 1, RingBase<KPT>{});
        }
        if (tap < ntaps) one_tap(tap, RingBase<0>{});
    }
}

// gemm_conv_gw for a wave that owns ALL rows of the workgroup's image (row tiles 0..MI-1; the waves split channels only)
// and must live in few registers: MI may be large (11 row tiles of a two-board image) next to two co-resident workgroups
// per compute unit. Differences: the activation fragments sit in a ring of AR registers sets, fragment f+AR being read
// right after the MFMAs that consumed fragment f (AR-1 groups of NI MFMAs cover each read); where a row tile's fragments
// of a tap start comes from a table in LDS, `taptab[tap][row]` (build_tap_table: offset / 16 of the tap's source row, or
// of the bank-preserving spot in the zero region when the tap falls off the board), read into the tile's offset register
// as soon as its last fragment of the previous tap has been requested - one 2-byte LDS read and one shift-add per row
// tile and tap instead of a bounds test, two address computations and a select.
template <int MI, int NI, int WN, int NCOLS, int KPT, int DEPTH, int AR>
__device__ __forceinline__ void gemm_conv_ring(f32x4 (&acc)[MI][NI], const uint4* __restrict__ wg, int ntaps, const char* act,
                                               const uint16_t* taptab) {
    AZ_LANE_GEO(MI, NI, WN);
    static_assert((KPT * MI) % AR == 0 && KPT % (DEPTH + 1) == 0 && AR < MI, "the fragment rings must keep static indices across taps");
    constexpr int NF = KPT * MI;          // fragments (groups of NI MFMAs) per tap
    constexpr int ROWS = MI * 16;
    const int nks = ntaps * KPT;
    const int kq16 = kq * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, nks * (NCOLS / 16) * 1024, 0x00020000);
    const int wvoff = ((wn * NI) * 64 + lane) * 16;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    auto wfrag = [&](int s, int ni) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + ni * 1024, min(s, nks - 1) * (NCOLS / 16) * 1024, 0);
        return __builtin_bit_cast(bf16x8, v);
    };
    // apc[mi]: before a tap's first fragment of row tile mi is requested, the table entry; from then on the byte offset
    int apc[MI];
    const uint16_t* tb = taptab + lr;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) apc[mi] = tb[mi * 16];
    bf16x8 a[AR], b[DEPTH + 1][NI];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[d][ni] = wfrag(d, ni);
    __syncthreads();   // the activation image is complete
#pragma unroll
    for (int f = 0; f < AR; ++f) {
        apc[f] = (apc[f] << 4) + kq16;
        a[f] = *(const bf16x8*)(act + apc[f]);
    }
    int s = 0;
    for (int tap = 0; tap < ntaps; ++tap) {
        tb += (tap + 1 < ntaps) ? ROWS : 0;   // the next tap's table row (the last tap reads its own again: harmless)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            constexpr int NB = DEPTH + 1;
            const int u = f / MI, mi = f - u * MI;
            if (mi == 0) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[(u + DEPTH) % NB][ni] = wfrag(s + u + DEPTH, ni);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u % NB][ni], a[f % AR], acc[mi][ni], 0, 0, 0);
            const int f2 = f + AR;
            bool table_read = false;
            if (f2 < NF) {
                const int m2 = f2 % MI, u2 = f2 / MI;
                if (u2 == 0) apc[m2] = (apc[m2] << 4) + kq16;   // row tiles AR..: first fragment of this tap
                a[f % AR] = *(const bf16x8*)(act + apc[m2] + u2 * 64);
                if (u2 == KPT - 1) { apc[m2] = tb[m2 * 16]; table_read = true; }   // the tile's last fragment of this tap
            } else {   // the next tap's first fragments (row tiles 0..AR-1)
                const int m2 = f2 - NF;
                apc[m2] = (apc[m2] << 4) + kq16;
                a[f % AR] = *(const bf16x8*)(act + apc[m2]);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
            if (table_read) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (mi < NI) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (mi == MI - 1) __builtin_amdgcn_sched_barrier(0);
        }
        s += KPT;
    }
}

// The table gemm_conv_ring reads: for every tap of a 3x3 convolution and every row of a workgroup's activation image
// (`rows` rows of `stride` bytes, the first g.vrows of them valid; zbase = byte offset of the zero region, a multiple of
// 256), where the tap's source row starts, in units of 16 bytes. A tap that falls off the board (or a row past the valid
// ones) points into the zero region at the spot with the same low 8 address bits, so that the fragment reads keep their
// LDS bank pattern. All threads of the workgroup take part; the caller synchronises before the first use.
__device__ __forceinline__ void build_tap_table(uint16_t* taptab, int rows, int stride, int zbase, const TileGeo& g, int tid, int nthreads) {
    for (int i = tid; i < 9 * rows; i += nthreads) {
        const int tap = i / rows, row = i - tap * rows;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int p = row - (int)((row * g.recip_np) >> 16) * g.np;
        const int y = (p * g.recip_n) >> 16, x = p - y * g.n;
        const int src = (row + dy * g.n + dx) * stride;
        const bool ok = row < g.vrows && (unsigned)(y + dy) < (unsigned)g.n && (unsigned)(x + dx) < (unsigned)g.n;
        taptab[i] = (uint16_t)((ok ? src : zbase + (src & 255)) >> 4);
    }
}

// A short convolution (the stem: few k-steps) with the weight fragments read from the packed global image, one k-step
// ahead, and nothing staged through LDS: `nksteps` k-steps of 32 input channels, `kpt` per tap. For kernels that have no
// LDS to spare for weight staging buffers; its share of a tower's time is a few per cent, so it is not pipelined further.
template <int MI, int NI, int WM, int WN, int NCOLS>
__device__ __forceinline__ void gemm_taps_gw(f32x4 (&acc)[MI][NI], const uint4* __restrict__ wg, int nksteps, int kpt, int ntaps, bool three,
                                             const char* act, int in_stride, const char* zrow, const TileGeo& g) {
    AZ_LANE_GEO(MI, NI, WN);
    int ry[MI], rx[MI], roff[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = row_base + mi * 16 + lr;
        const int b = (row * g.recip_np) >> 16, p = row - b * g.np;
        ry[mi] = (p * g.recip_n) >> 16;
        rx[mi] = p - ry[mi] * g.n;
        if (row >= g.vrows) ry[mi] = -64;   // fails every bounds check
        roff[mi] = row * in_stride + kq * 16;
    }
    const int zoff = (int)(zrow - act) + kq * 16;
    const uint4* wl = wg + (wn * NI) * 64 + lane;   // k-step s, tile ni: wl[(s * (NCOLS / 16) + ni) * 64]
    bf16x8 b[2][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[0][ni] = __builtin_bit_cast(bf16x8, wl[ni * 64]);
    __syncthreads();   // the activation image is complete
    int tap = 0, cc = 0;
    for (int s = 0; s < nksteps; s += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two k-steps per trip keep the fragment ring's indices static
            const int sn = min(s + h + 1, nksteps - 1);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[1 - h][ni] = __builtin_bit_cast(bf16x8, wl[((size_t)sn * (NCOLS / 16) + ni) * 64]);
            int ap[MI];
            frag_offsets<MI>(ap, ry, rx, roff, tap, cc, ntaps, three, g.n, in_stride, zoff);
            if (++cc == kpt) { cc = 0; ++tap; }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const bf16x8 a = *(const bf16x8*)(act + ap[mi]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[h][ni], a, acc[mi][ni], 0, 0, 0);
            }
        }
    }
}

template <int MI, int NI>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MI][NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
}

static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Pack folded weights w[cout][cin][taps] into the LDS image the weight fragments are read from, k-step after
// k-step (32 input channels of one tap each; the count is padded to an even number with zeros):
// [k-step][channel tile][lane][8 consecutive input channels]. Lane l of a fragment holds output channel row
// l & 15 of the tile and input channels 8*(l >> 4).. of the k-step. Tile rows are assigned to output
// channels so that the MFMA result gives a lane 8 consecutive channels over the tile pair (2p, 2p+1):
// row r of tile 2p+e is channel 32*p + 8*(r >> 2) + 4*e + (r & 3).
// channel_major: the k-steps run over 64-channel chunks first and over the taps inside a chunk (k-step
// (chunk*taps + tap)*2 + half) instead of tap after tap; cin_pad must then be a multiple of 64.
// slab_tiles (odd): the tiles are dealt to waves in runs of slab_tiles, and the last tile of a run is not paired:
// its rows are output channels in natural order.
static inline std::vector<uint16_t> pack_conv(const std::vector<float>& w, int cout, int cin, int taps, int cin_pad, int ncols,
                                              bool channel_major = false, int slab_tiles = 0) {
    const int kpt = cin_pad / 32, nks = (taps * kpt + 1) & ~1, ntile = ncols / 16;
    std::vector<uint16_t> out((size_t)nks * ntile * 64 * 8, 0);
    for (int s = 0; s < nks; ++s) {
        int tap = s / kpt, c0 = (s % kpt) * 32;
        if (channel_major) {
            const int tile = s >> 1;
            tap = tile % taps;
            c0 = (tile / taps) * 64 + (s & 1) * 32;
        }
        if (tap >= taps) continue;
        for (int nt = 0; nt < ntile; ++nt)
            for (int l = 0; l < 64; ++l) {
                const int r = l & 15;
                int co = 32 * (nt >> 1) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3);
                if (slab_tiles & 1) {
                    const int slab = nt / slab_tiles, e = nt - slab * slab_tiles, base = slab * slab_tiles * 16;
                    co = e == slab_tiles - 1 ? base + 16 * e + r : base + 32 * (e >> 1) + 8 * (r >> 2) + 4 * (e & 1) + (r & 3);
                }
                if (co >= cout) continue;
                for (int j = 0; j < 8; ++j) {
                    const int ci = c0 + 8 * (l >> 4) + j;
                    if (ci < cin) out[(((size_t)s * ntile + nt) * 64 + l) * 8 + j] = f2bf(w[((size_t)co * cin + ci) * taps + tap]);
                }
            }
    }
    return out;
}

}  // namespace az
