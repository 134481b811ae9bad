// bf16 MFMA inference of the policy/value tower (model.py:43-97, eval mode) layer by layer, for the shapes
// whose activations do not fit one workgroup's LDS (boards above 9x9 other than 11x11 to 13x13 with 192 channels; 256 channels): every convolution is one
// launch, an implicit GEMM over all boards of a chunk on v_mfma_f32_16x16x32_bf16 with the weight tiles
// double-buffered in LDS like in the fused tower (az_mfma_gemm.cuh). The 3x3 convolutions (k_conv3_slab) stage
// the bf16 [position][channel] activations through LDS slabs; the 1x1 head (k_conv_layer) reads them from L2. The
// squeeze-excite gate (pooled in position order, so a board's result does not depend on its neighbours in the
// batch) and the residual add are one bandwidth-bound kernel between the convolutions (one workgroup per board).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "az_net.h"
#include "az_mfma_gemm.cuh"

namespace az {
int fail(int code, const std::string& msg);

struct ConvArgs {
    const char* in;        // [rows][in_stride bytes] bf16 activations, followed by a zero region at zero_off
    int in_stride, zero_off;
    const uint4* w;        // packed weight tiles, one run of w_block_vecs vectors per block of NCOLS output channels
    size_t w_block_vecs;
    const float* bias;     // [cout]
    char* out;             // [rows][out_stride bytes] bf16
    int out_stride, rows, n, np, ntaps, relu;
    int col_blocks;        // k_conv3_slab: blocks of NCOLS output channels (1-D grid, see slab_block)
    const int* rows_dev;   // if set: the batch's board count lives here; this launch covers boards board_off .. of it,
    int board_off;         // at most rows / np of them (rows is then the upper bound the grid was sized for)
};

// Rows this launch really has: all `rows` of them, or what the device-side board count leaves for its chunk.
__device__ __forceinline__ int live_rows(int rows, int np, const int* rows_dev, int board_off) {
    if (!rows_dev) return rows;
    return max(0, min(rows, (*rows_dev - board_off) * np));
}

// Every kernel of this file walks its tiles (or boards, or vectors) with a stride of the grid size: the grids are capped
// (kPersistGrid), so a launch sized for an upper bound far above the device-side count costs no empty workgroups (a
// workgroup that only reads the count and leaves still takes the dispatcher about 7 ns: 30 000 of them per launch made a
// forward five times slower).
constexpr int kPersistGrid = 2048;

// One 256-row tile (row tile bx, block by of NCOLS output channels) of a convolution whose activation fragments come
// straight from global memory (the 1x1 head; 3x3 shapes without a slab kernel). a.rows: the live rows.
template <int KPT, int NCOLS, int WM, int WN>
__device__ __forceinline__ void conv_layer_tile(const ConvArgs& a, int bx, int by, char* smem) {
    constexpr int MI = 256 / (WM * 16), NI = NCOLS / (WN * 16);
    static_assert(NI % 2 == 0, "channel tiles are stored in pairs");
    TileGeo g;
    g.row0 = bx * 256; g.n = a.n; g.np = a.np; g.vrows = a.rows; g.recip_np = 0; g.recip_n = 0;
    f32x4 acc[MI][NI];
    zero_acc<MI, NI>(acc);
    gemm_conv<MI, NI, WM, WN, NCOLS, KPT, true>(acc, a.w + (size_t)by * a.w_block_vecs, a.ntaps, a.ntaps == 9, a.in, a.in_stride,
                                                  a.in + a.zero_off, smem, g);
    AZ_LANE_GEO(MI, NI, WN);
    const int cbase = by * NCOLS + col0 + 8 * kq;
#pragma unroll
    for (int p = 0; p < NI / 2; ++p) {
        const float4 x0 = *(const float4*)(a.bias + cbase + 32 * p), x1 = *(const float4*)(a.bias + cbase + 32 * p + 4);
        const float bv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = g.row0 + row_base + mi * 16 + lr;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mi][2 * p][r] + bv[r]; v[4 + r] = acc[mi][2 * p + 1][r] + bv[4 + r]; }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (row < a.rows)
                *(uint4*)(a.out + (size_t)row * a.out_stride + (size_t)(cbase + 32 * p) * 2) =
                    make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        }
    }
}

template <int KPT, int NCOLS, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void k_conv_layer(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    a.rows = live_rows(a.rows, a.np, a.rows_dev, a.board_off);
    for (int bx = blockIdx.x; bx * 256 < a.rows; bx += gridDim.x) {
        if (bx != (int)blockIdx.x) __syncthreads();   // the previous tile's last weight fragments have been read
        conv_layer_tile<KPT, NCOLS, WM, WN>(a, bx, blockIdx.y, smem);
    }
}

// 3x3 convolution with the activations staged through LDS: the workgroup's 256 rows plus a halo of n+1 rows
// on either side, 64 input channels at a time ("slab", row stride 160 B: conflict-free 16-byte fragment reads),
// double-buffered; the nine taps of a channel chunk read the same slab at shifted rows, so every activation
// byte is fetched from L2 once per chunk instead of nine times. K order: chunk-major (pack_conv channel_major).
// Slab kc+1 is fetched global -> registers -> LDS one 16-byte piece per thread and tile while chunk kc computes.
constexpr int kSlabStride = 160, kMaxHalo = 20, kSlabRows = 256 + 2 * kMaxHalo, kSlabBytes = kSlabRows * kSlabStride;
constexpr int kSlabPieces = (kSlabRows * 8 + 511) / 512;   // 16-byte pieces per thread of a 512-thread workgroup

template <int NCOLS, int WM, int WN>
__device__ __forceinline__ void conv3_slab_tile(const ConvArgs& a, int bx, int by, char* smem) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int MI = 256 / (WM * 16), NI = NCOLS / (WN * 16);
    constexpr int tile_bytes = 64 * NCOLS * 2, nvec = tile_bytes / 16, kstep_bytes = tile_bytes / 2;
    constexpr int NCHUNK = nvec / 64, PW = (NCHUNK + NW - 1) / NW;
    constexpr bool RAGGED = (NCHUNK % NW) != 0;
    static_assert(NI % 2 == 0 && NT == 512, "tile pairs; slab pieces are dealt to 512 threads");
    char* slab = smem;                         // two slabs
    char* zreg = smem + 2 * kSlabBytes;        // 1 KiB of zeros, 256-byte aligned (kSlabBytes is a multiple of 256)
    char* bbuf = zreg + 1024;                  // two weight tiles
    AZ_LANE_GEO(MI, NI, WN);
    const int n = a.n, np = a.np, halo = n + 1, srows = 256 + 2 * halo;
    const int row0 = bx * 256, C2 = a.in_stride, nkc = a.in_stride / 128, ntiles = nkc * 9;
    const uint4* wg = a.w + (size_t)by * a.w_block_vecs;
    for (int i = t_; i < 256; i += NT) ((uint32_t*)zreg)[i] = 0;
    stage_tile<NW>(wg, bbuf, nvec, lane, wave);
    stage_tile<NW>(wg + nvec, bbuf + tile_bytes, nvec, lane, wave);
    uint4 stg[PW];
    const uint4* wsrc = wg + lane + wave * 64;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        stg[i] = make_uint4(0, 0, 0, 0);
        if (!RAGGED || wave + i * NW < NCHUNK) stg[i] = wsrc[(size_t)2 * nvec + i * NW * 64];
    }
    // this thread's slab pieces: piece j = t_ + i*NT -> slab row j >> 3, 16-byte part j & 7 (recomputed where used:
    // holding the offsets would cost registers the accumulators need)
    auto piece = [&](int i, int& src, int& dst) {
        const int j = opaque(t_) + i * NT, srow = j >> 3, part = j & 7, grow = row0 - halo + srow;
        dst = srow < srows ? srow * kSlabStride + part * 16 : -1;
        src = (srow < srows && grow >= 0 && grow < a.rows) ? grow * C2 + part * 16 : -1;   // -1: outside the array, zeros
    };
#pragma unroll
    for (int i = 0; i < kSlabPieces; ++i) {   // slab 0
        int src, dst;
        piece(i, src, dst);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (src >= 0) v = *(const uint4*)(a.in + src);
        if (dst >= 0) *(uint4*)(slab + dst) = v;
    }
    uint4 sp = make_uint4(0, 0, 0, 0);   // the piece of the next slab in flight
    // fragment geometry
    int roff[MI], vmask[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int lrow = row_base + mi * 16 + lr, row = row0 + lrow;
        const int p = row % np, y = p / n, x = p - y * n;
        int m = 0;
        if (row < a.rows) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                if ((unsigned)yy < (unsigned)n && (unsigned)xx < (unsigned)n) m |= 1 << tp;
            }
        }
        vmask[mi] = m;
        roff[mi] = (lrow + halo) * kSlabStride + kq * 16;
    }
    const int zbase = 2 * kSlabBytes;   // offset of the zero region from `slab`
    const int woff = ((wn * NI) * 64 + lane) * 16, soff = (wave * 64 + lane) * 16;
    auto tap_offsets = [&](int (&out)[MI], int tap, int sbase) {
        const int shift = ((tap / 3 - 1) * n + (tap - (tap / 3) * 3 - 1)) * kSlabStride + sbase;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int reg = roff[mi] + shift;
            out[mi] = (vmask[mi] >> tap & 1) ? reg : zbase + (reg & 255);
        }
    };
    f32x4 acc[MI][NI];
    zero_acc<MI, NI>(acc);
    int apc[MI], apn[MI];
    tap_offsets(apc, 0, 0);
    if (ntiles > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 a0[MI], b0[NI], a1[MI], b1[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a0[mi] = *(const bf16x8*)(slab + apc[mi]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b0[ni] = *(const bf16x8*)(bbuf + woff + ni * 1024);
    int t = 0;
    for (int kc = 0; kc < nkc; ++kc) {
        const int sb_cur = (kc & 1) * kSlabBytes, sb_nxt = kSlabBytes - sb_cur;
        const bool more = kc + 1 < nkc;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++t) {
            const int par = (t & 1) * tile_bytes;
            char* X = bbuf + par;
            const char* Y = bbuf + (tile_bytes - par);
            // fragment offsets of the next tile: next tap in this slab, or tap 0 of the next chunk's slab
            if (tap < 8) tap_offsets(apn, tap + 1, sb_cur);
            else tap_offsets(apn, 0, sb_nxt);
            // ---- first k-step
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a1[mi] = *(const bf16x8*)(slab + apc[mi] + 64);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b1[ni] = *(const bf16x8*)(X + kstep_bytes + woff + ni * 1024);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            {   // weight tile t+2 -> X, fetch tile t+3
                const uint4* src = wsrc + (size_t)min(t + 3, ntiles - 1) * nvec;
#pragma unroll
                for (int i = 0; i < PW; ++i)
                    if (!RAGGED || wave + i * NW < NCHUNK) {
                        *(uint4*)(X + soff + i * NW * 1024) = stg[i];
                        stg[i] = src[i * NW * 64];
                    }
            }
            // next slab: piece `tap` is fetched during tile `tap` and stored during tile tap+1 (tiles 0..kSlabPieces)
            if (more) {
                int src, dst;
                if (tap >= 1 && tap <= kSlabPieces) {
                    piece(tap - 1, src, dst);
                    if (dst >= 0) *(uint4*)(slab + sb_nxt + dst) = sp;
                }
                if (tap < kSlabPieces) {
                    piece(tap, src, dst);
                    sp = make_uint4(0, 0, 0, 0);
                    if (src >= 0) sp = *(const uint4*)(a.in + src + (kc + 1) * 128);
                }
            }
            // ---- second k-step
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a0[mi] = *(const bf16x8*)(slab + apn[mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b0[ni] = *(const bf16x8*)(Y + woff + ni * 1024);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) apc[mi] = apn[mi];
        }
    }
    // epilogue: bias, ReLU, bf16 rows to global memory
    const int cbase = by * NCOLS + col0 + 8 * kq;
#pragma unroll
    for (int p = 0; p < NI / 2; ++p) {
        const float4 x0 = *(const float4*)(a.bias + cbase + 32 * p), x1 = *(const float4*)(a.bias + cbase + 32 * p + 4);
        const float bv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = row0 + row_base + mi * 16 + lr;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mi][2 * p][r] + bv[r]; v[4 + r] = acc[mi][2 * p + 1][r] + bv[4 + r]; }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (row < a.rows)
                *(uint4*)(a.out + (size_t)row * a.out_stride + (size_t)(cbase + 32 * p) * 2) =
                    make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        }
    }
}

template <int NCOLS, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void k_conv3_slab(ConvArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    a.rows = live_rows(a.rows, a.np, a.rows_dev, a.board_off);
    // Workgroups are dealt round-robin over the 8 XCDs, so ids 8 apart share an L2: the column blocks of one row tile get
    // (virtual) ids 8 apart and fetch the tile's activations into one L2 instead of one L2 each. The grid size is a
    // multiple of 8 * col_blocks, so a workgroup's later tiles keep that property.
    const int per_group = 8 * a.col_blocks, total = ((a.rows + 255) / 256 + 7) / 8 * per_group;
    bool first = true;
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        const int grp = v / per_group, within = v - grp * per_group;
        const int bx = grp * 8 + (within & 7), by = within >> 3;
        if (bx * 256 >= a.rows) continue;   // padding of the last group of 8 row tiles (whole workgroup)
        if (!first) __syncthreads();        // the previous tile's last LDS reads are done
        first = false;
        conv3_slab_tile<NCOLS, WM, WN>(a, bx, by, smem);
    }
}

// The 3x3 convolution for C output channels as 256-thread workgroups, two per compute unit: 96 rows x all C output
// channels each (4 waves = 4 channel quarters, every wave owns all 96 rows: 6 x C/64 accumulator tiles), so that one
// workgroup's barriers, slab refills and epilogue run under the other's MFMAs. The activations go through the same
// 64-channel slabs as in k_conv3_slab (the workgroup's rows + a halo of n+1 rows either side, double-buffered, a zero
// region behind each slab); the weight fragments come straight from the packed global image (fragment order, chunk-major:
// one coalesced 1 KiB load per fragment, two k-steps ahead), so LDS serves activation fragments only: one 1 KiB LDS read
// per 4 MFMAs and wave, half of what the 64 x 64 wave tiles with LDS-staged weights of k_conv3_slab need. Where a row
// tile's fragments of a tap start comes from a table in LDS built once per workgroup (offset / 16 of the tap's source row
// in the slab, or of the bank-preserving spot in the zero region when the tap falls off the board). K order and hence
// every output bit equal k_conv3_slab's. Measured at 19x19, 256 channels, 20 blocks: 96 rows with two k-steps of weights
// in flight 7.90 ms per 512 boards and 4.14 ms per 255; 128 rows with one k-step in flight (all the registers allow) 7.9-8.1
// and 4.64 ms: the smaller tiles fill the last round of workgroups better.
#ifndef AZ_DUO_NB
#define AZ_DUO_NB 3   // weight fragment sets in the ring: the k-step in use + (AZ_DUO_NB - 1) in flight
#endif
template <int ROWS>
struct DuoGeo {   // LDS layout of a workgroup of ROWS rows
    static constexpr int kRows = ROWS, kSlabRows = ROWS + 2 * kMaxHalo, kSlabData = kSlabRows * kSlabStride;
    static constexpr int kSlabBytes = kSlabData + 512;             // + zero region (kSlabData is a multiple of 256)
    static constexpr int kPieces = (kSlabRows * 8 + 255) / 256;   // 16-byte slab pieces per thread
    static constexpr int kTab = 2 * kSlabBytes;                   // the tap table, then a 16-byte trash slot per thread for
    static constexpr int kTrash = kTab + 9 * ROWS * 2;            // slab pieces past the slab
    static constexpr int kLds = kTrash + 256 * 16;
    static_assert(kSlabData % 256 == 0 && kTrash % 16 == 0 && kPieces <= 7, "alignment; a slab piece is fetched at tap i and stored at tap i + 2");
};
// Rows per workgroup by channel count: 256 channels (4 accumulator tiles per row tile and wave) 96, which leaves the
// registers for two k-steps of weights in flight; 192 channels (3 tiles, the third unpaired: pack_conv slab_tiles) 128.
template <int C> struct DuoRows { static constexpr int value = C == 256 ? 96 : 128; };

// Waits for this wave's LDS traffic only (the weight fragments in flight stay in flight), then the workgroup barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int C>
__device__ __forceinline__ void conv3_duo_tile(const ConvArgs& a, int bx, char* smem) {
    typedef DuoGeo<DuoRows<C>::value> G;
    constexpr int kDuoRows = G::kRows, kDuoSlabData = G::kSlabData, kDuoSlabBytes = G::kSlabBytes, kDuoPieces = G::kPieces, kDuoTab = G::kTab,
                  kDuoTrash = G::kTrash;
    constexpr int MI = kDuoRows / 16, NI = C / 64, WN = 4, NT = 256, KPT = 2, AR = 4, NF = KPT * MI, NB = AZ_DUO_NB, KS = 9 * KPT;
    static_assert(C % 64 == 0 && NF % AR == 0 && KS % NB == 0, "whole tiles per wave; static ring indices across taps and chunks");
    char* slab = smem;
    uint16_t* taptab = (uint16_t*)(smem + kDuoTab);
    AZ_LANE_GEO(MI, NI, WN);
    const int n = a.n, np = a.np, halo = n + 1, srows = kDuoRows + 2 * halo;
    const int row0 = bx * kDuoRows;
    const int C2 = a.in_stride, nkc = a.in_stride / 128, nks = nkc * KS;
    for (int i = t_; i < 2 * 128; i += NT) ((uint32_t*)(slab + kDuoSlabData + (i >> 7) * kDuoSlabBytes))[i & 127] = 0;
    for (int i = t_; i < 9 * kDuoRows; i += NT) {
        const int tap = i / kDuoRows, r = i - tap * kDuoRows, row = row0 + r;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int p = row % np, y = p / n, x = p - y * n;
        const int src = (r + halo + dy * n + dx) * kSlabStride;
        const bool ok = row < a.rows && (unsigned)(y + dy) < (unsigned)n && (unsigned)(x + dx) < (unsigned)n;
        taptab[i] = (uint16_t)((ok ? src : kDuoSlabData + (src & 255)) >> 4);
    }
    // this thread's slab pieces: piece j = t_ + i*NT -> slab row j >> 3, 16-byte part j & 7. src: the piece's byte offset
    // in the input array for chunk 0, or kOob for rows outside the array (a buffer load from there gives zeros); dst:
    // where it goes in a slab (a piece past the slab: the thread's trash slot). No branches: the staging code sits
    // between the MFMAs of a tap's first k-step.
    constexpr int kOob = (int)0x80000000;
    auto piece = [&](int i, int& src, int& dst) {
        const int j = opaque(t_) + i * NT, srow = j >> 3, part = j & 7, grow = row0 - halo + srow;
        dst = srow < srows ? srow * kSlabStride + part * 16 : kDuoTrash + (j & (NT - 1)) * 16;
        src = (srow < srows && grow >= 0 && grow < a.rows) ? grow * C2 + part * 16 : kOob;
    };
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (unsigned)a.rows * (unsigned)C2, 0x00020000);
    // the fetch of piece i of chunk kcn (zeros when !live; src == kOob stays out of bounds: the array is far below 2 GiB)
    auto get = [&](int i, int kcn, bool live) {
        int src, dst;
        piece(i, src, dst);
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(irs, live ? src + kcn * 128 : kOob, 0, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    };
#pragma unroll
    for (int i = 0; i < kDuoPieces; ++i) {   // slab 0
        int src, dst;
        piece(i, src, dst);
        *(uint4*)(slab + dst) = get(i, 0, true);
    }
    // the two pieces of the next slab in flight: fetched two taps before they are stored
    uint4 sp[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    // k-step s, tile ni of this wave: byte (s * (C/16) + wn * NI + ni) * 1024 + lane * 16 of the packed image
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, nks * (C / 16) * 1024, 0x00020000);
    const int wvoff = ((wn * NI) * 64 + lane) * 16;
    auto wfrag = [&](int s, int ni) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + ni * 1024, min(s, nks - 1) * (C / 16) * 1024, 0);
        return __builtin_bit_cast(bf16x8, v);
    };
    const int kq16 = kq * 16;
    const uint16_t* tb = taptab + lr;
    f32x4 acc[MI][NI];
    zero_acc<MI, NI>(acc);
    bf16x8 af[AR], b[NB][NI];
#pragma unroll
    for (int d = 0; d < NB - 1; ++d)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[d][ni] = wfrag(d, ni);
    int s = 0;
#pragma nounroll
    for (int kc = 0; kc < nkc; ++kc) {
        const char* act = slab + (kc & 1) * kDuoSlabBytes;
        char* nxt = slab + ((kc + 1) & 1) * kDuoSlabBytes;
        const bool more = kc + 1 < nkc;
        lds_barrier();   // slab kc (and, the first time, the table and the zero regions) is complete; nobody reads the other slab any more
        int apc[MI];     // before a tap's first fragment of row tile mi is requested, the table entry; from then on the byte offset
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) apc[mi] = tb[mi * 16];
#pragma unroll
        for (int f = 0; f < AR; ++f) {
            apc[f] = (apc[f] << 4) + kq16;
            af[f] = *(const bf16x8*)(act + apc[f]);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // next slab: piece `tap` is fetched during tap `tap` (after MFMA group MI-2 of the first k-step) and stored
            // during tap + 2 (after group 0); after the last chunk: zeros into the slab nobody reads any more. Fenced where
            // they stand, so that the compiler neither bunches them up nor stretches their registers over the tap.
            const bool putting = tap >= 2 && tap <= kDuoPieces + 1, getting = tap < kDuoPieces;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int u = f / MI, mi = f - u * MI, ks = tap * KPT + u;
                if (mi == 0) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) b[(ks + NB - 1) % NB][ni] = wfrag(s + ks + NB - 1, ni);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks % NB][ni], af[f % AR], acc[mi][ni], 0, 0, 0);
                const int f2 = f + AR;
                int ds_reads = 0;
                if (f2 < NF) {
                    const int m2 = f2 % MI, u2 = f2 / MI;
                    if (u2 == 0) apc[m2] = (apc[m2] << 4) + kq16;   // row tiles AR..: first fragment of this tap
                    af[f % AR] = *(const bf16x8*)(act + apc[m2] + u2 * 64);
                    ds_reads = 1;
                    if (u2 == KPT - 1 && tap < 8) { apc[m2] = tb[(tap + 1) * kDuoRows + m2 * 16]; ds_reads = 2; }   // the tile's last fragment of this tap
                } else if (tap < 8) {   // the next tap's first fragments (row tiles 0..AR-1)
                    const int m2 = f2 - NF;
                    apc[m2] = (apc[m2] << 4) + kq16;
                    af[f % AR] = *(const bf16x8*)(act + apc[m2]);
                    ds_reads = 1;
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
                if (ds_reads == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (ds_reads == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (mi < NI) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (u == 0 && ((mi == 0 && putting) || (mi == MI - 2 && getting))) {
                    if (mi == 0 && putting) {
                        int src, dst;
                        piece(tap - 2, src, dst);
                        *(uint4*)(nxt + dst) = sp[tap & 1];
                    }
                    if (mi == MI - 2 && getting) sp[tap & 1] = get(tap, kc + 1, more);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (mi == MI - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        s += KS;
    }
    // epilogue: bias, ReLU, bf16 rows to global memory
    const int cbase = col0 + 8 * kq;
#pragma unroll
    for (int p = 0; p < NI / 2; ++p) {
        const float4 x0 = *(const float4*)(a.bias + cbase + 32 * p), x1 = *(const float4*)(a.bias + cbase + 32 * p + 4);
        const float bv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = row0 + mi * 16 + lr;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mi][2 * p][r] + bv[r]; v[4 + r] = acc[mi][2 * p + 1][r] + bv[4 + r]; }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (row < a.rows)
                *(uint4*)(a.out + (size_t)row * a.out_stride + (size_t)(cbase + 32 * p) * 2) =
                    make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
        }
    }
    if constexpr (NI & 1) {   // the unpaired last tile: 4 consecutive channels per lane (natural row order), 8-byte stores
        const int ch4 = col0 + 32 * (NI / 2) + 4 * kq;
        const float4 x0 = *(const float4*)(a.bias + ch4);
        const float bv[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = row0 + mi * 16 + lr;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mi][NI - 1][r] + bv[r]; if (a.relu) v[r] = fmaxf(v[r], 0.f); }
            if (row < a.rows) *(uint2*)(a.out + (size_t)row * a.out_stride + (size_t)ch4 * 2) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
        }
    }
}

template <int C>
__global__ __launch_bounds__(256, 2) void k_conv3_duo(ConvArgs a) {
    constexpr int kDuoRows = DuoRows<C>::value;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    a.rows = live_rows(a.rows, a.np, a.rows_dev, a.board_off);
    // consecutive row tiles share halo rows: give them to one XCD (workgroup ids are dealt round-robin over the 8 XCDs and
    // the grid size is a multiple of 8, so a workgroup's tiles v, v + grid, ... all have v & 7 == its XCD)
    const int per_xcd = ((a.rows + kDuoRows - 1) / kDuoRows + 7) >> 3;
    bool first = true;
    for (int v = blockIdx.x; v < 8 * per_xcd; v += gridDim.x) {
        const int bx = (v & 7) * per_xcd + (v >> 3);
        if (bx * kDuoRows >= a.rows) continue;   // whole workgroup
        if (!first) __syncthreads();             // the previous tile's last LDS reads are done
        first = false;
        conv3_duo_tile<C>(a, bx, smem);
    }
}

// [rows][32] bf16 feature rows -> [rows][64] (upper half zero): the stem then runs through the same tile loop.
__global__ __launch_bounds__(256) void k_pad_features(const uint4* __restrict__ in, uint4* __restrict__ out, int rows, int np,
                                                      const int* rows_dev, int board_off) {
    rows = live_rows(rows, np, rows_dev, board_off);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * 8; i += (size_t)gridDim.x * 256) {   // 16-byte pieces, 8 per output row
        const size_t row = i >> 3;
        const int part = (int)(i & 7);
        out[i] = part < 4 ? in[row * 4 + part] : make_uint4(0, 0, 0, 0);
    }
}

// Squeeze-excite gate of every board (model.py:9-22): mean over the board's positions, two small fully connected
// layers in float32, sigmoid. One 256-thread workgroup per board. Pooling: C/8 channel groups x 256/(C/8) position
// phases; a thread adds every phases-th position of its 8 channels (16-byte loads, eight independent sums), then the
// phases are added in index order. The order depends on the position within the board only, so a board's gate does not
// depend on where the board sits in the batch.
// The board's gate stays in LDS and the same workgroup then applies it: x <- relu(y * gate + x) (model.py:38-40) for
// the board's np x C values, 8 channels per thread and trip; y was just read for the pooling and comes from the L2.
__global__ __launch_bounds__(256) void k_se_residual_layer(const uint16_t* __restrict__ y, int np, int C, int hid, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                           uint16_t* __restrict__ x, int boards, const int* rows_dev, int board_off) {
    extern __shared__ float sh[];   // partial [phases][C], pooled [C], hidden [hid], gate [C]
    if (rows_dev) boards = min(boards, *rows_dev - board_off);
    const int groups = C / 8, phases = 256 / groups;   // C in {64,128,192,256}: 8..32 groups; 192 -> 24 groups, 10 phases
    float* partial = sh;
    float* pooled = partial + phases * C;
    float* hidden = pooled + C;
    float* gate = hidden + ((hid + 3) & ~3);
    const int t = threadIdx.x;
    const int grp = t % groups, ph = t / groups;
    for (int b = blockIdx.x; b < boards; b += gridDim.x) {
    if (b != (int)blockIdx.x) __syncthreads();   // the previous board's gate has been read
    if (ph < phases) {
        const uint4* src = (const uint4*)(y + (size_t)b * np * C) + grp;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int p = ph;
        for (; p + 3 * phases < np; p += 4 * phases) {   // four loads in flight, added in position order
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = src[(size_t)(p + k * phases) * groups];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[0] += bf16_lo(v[k].x); s[1] += bf16_hi(v[k].x); s[2] += bf16_lo(v[k].y); s[3] += bf16_hi(v[k].y);
                s[4] += bf16_lo(v[k].z); s[5] += bf16_hi(v[k].z); s[6] += bf16_lo(v[k].w); s[7] += bf16_hi(v[k].w);
            }
        }
        for (; p < np; p += phases) {
            const uint4 v = src[(size_t)p * groups];
            s[0] += bf16_lo(v.x); s[1] += bf16_hi(v.x); s[2] += bf16_lo(v.y); s[3] += bf16_hi(v.y);
            s[4] += bf16_lo(v.z); s[5] += bf16_hi(v.z); s[6] += bf16_lo(v.w); s[7] += bf16_hi(v.w);
        }
        float* dst = partial + ph * C + grp * 8;
        *(float4*)dst = make_float4(s[0], s[1], s[2], s[3]);
        *(float4*)(dst + 4) = make_float4(s[4], s[5], s[6], s[7]);
    }
    __syncthreads();
    const float inv = 1.0f / (float)np;
    for (int ch = t; ch < C; ch += 256) {
        float a = 0.f;
        for (int k = 0; k < phases; ++k) a += partial[k * C + ch];
        pooled[ch] = a * inv;
    }
    __syncthreads();
    {   // fc1 + relu: 8 lanes per hidden unit, each a contiguous run of C/8 channels, fixed reduction tree
        const int sub = t & 7, run = C / 8;
        for (int j = t >> 3; j < hid; j += 32) {
            const float* wr = w1 + (size_t)j * C + sub * run;
            const float* pr = pooled + sub * run;
            float a = 0.f;
            for (int c = 0; c < run; c += 4) {
                const float4 w = *(const float4*)(wr + c);
                a = fmaf(w.x, pr[c], a); a = fmaf(w.y, pr[c + 1], a); a = fmaf(w.z, pr[c + 2], a); a = fmaf(w.w, pr[c + 3], a);
            }
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            a += __shfl_xor(a, 4);
            if (sub == 0) hidden[j] = fmaxf(a + b1[j], 0.f);
        }
    }
    __syncthreads();
    for (int ch = t; ch < C; ch += 256) {   // fc2 + sigmoid (hid is a multiple of 4: max(8, C/8))
        const float* wr = w2 + (size_t)ch * hid;
        float a = b2[ch];
        for (int j = 0; j < hid; j += 4) {
            const float4 w = *(const float4*)(wr + j);
            a = fmaf(w.x, hidden[j], a); a = fmaf(w.y, hidden[j + 1], a); a = fmaf(w.z, hidden[j + 2], a); a = fmaf(w.w, hidden[j + 3], a);
        }
        gate[ch] = 1.f / (1.f + __expf(-a));
    }
    __syncthreads();
    const uint4* yb = (const uint4*)(y + (size_t)b * np * C);
    uint4* xb = (uint4*)(x + (size_t)b * np * C);
    const int total = np * groups;   // 16-byte vectors of the board; thread t keeps channel group t % groups on every trip
    float gv[8];
    {
        const float4 g0 = *(const float4*)(gate + grp * 8), g1 = *(const float4*)(gate + grp * 8 + 4);
        gv[0] = g0.x; gv[1] = g0.y; gv[2] = g0.z; gv[3] = g0.w; gv[4] = g1.x; gv[5] = g1.y; gv[6] = g1.z; gv[7] = g1.w;
    }
    const int step = phases * groups;   // threads that take part (a multiple of groups, so the channel group is fixed)
    auto apply = [&](const uint4& yv, const uint4& xv) {
        const uint32_t yu[4] = {yv.x, yv.y, yv.z, yv.w}, xu[4] = {xv.x, xv.y, xv.z, xv.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack_bf16(fmaxf(bf16_lo(yu[k]) * gv[2 * k] + bf16_lo(xu[k]), 0.f), fmaxf(bf16_hi(yu[k]) * gv[2 * k + 1] + bf16_hi(xu[k]), 0.f));
        return make_uint4(o[0], o[1], o[2], o[3]);
    };
    if (ph < phases) {
        int i = t;
        for (; i + 3 * step < total; i += 4 * step) {
            uint4 yv[4], xv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { yv[k] = yb[i + k * step]; xv[k] = xb[i + k * step]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) xb[i + k * step] = apply(yv[k], xv[k]);
        }
        for (; i < total; i += step) xb[i] = apply(yb[i], xb[i]);
    }
    }
}

// x <- relu(y * gate + x) (model.py:38-40), 8 channels per thread; gate == nullptr: no squeeze-excite.
__global__ __launch_bounds__(256) void k_scale_residual(const uint4* __restrict__ y, const float* __restrict__ gate, uint4* __restrict__ x,
                                                        int rows, int np, int C, const int* rows_dev, int board_off) {
    rows = live_rows(rows, np, rows_dev, board_off);
    const int per_row = C / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * per_row; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / per_row), c8 = (int)(i - (size_t)row * per_row) * 8;
    const uint4 yv = y[i], xv = x[i];
    const uint32_t yu[4] = {yv.x, yv.y, yv.z, yv.w}, xu[4] = {xv.x, xv.y, xv.z, xv.w};
    float gv[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (gate) {
        const float* gp = gate + (size_t)(row / np) * C + c8;
        const float4 g0 = *(const float4*)gp, g1 = *(const float4*)(gp + 4);
        gv[0] = g0.x; gv[1] = g0.y; gv[2] = g0.z; gv[3] = g0.w; gv[4] = g1.x; gv[5] = g1.y; gv[6] = g1.z; gv[7] = g1.w;
    }
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = pack_bf16(fmaxf(bf16_lo(yu[k]) * gv[2 * k] + bf16_lo(xu[k]), 0.f), fmaxf(bf16_hi(yu[k]) * gv[2 * k + 1] + bf16_hi(xu[k]), 0.f));
    x[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------------------ host
struct PackedConv {
    uint4* w = nullptr;
    float* b = nullptr;
    size_t block_vecs = 0;
    int cin_pad = 0, cout = 0, ncols = 0, taps = 0;
    bool slab = false;   // 3x3 with activations staged through LDS (weights packed chunk-major)
    bool duo = false;    // packed for k_conv3_duo with an odd tile count per wave (the wave's last tile unpaired)
};

struct LayerNet {
    az_net_config cfg;
    int np, q, chunk_boards, ncols;   // ncols: output channels per workgroup of the tower convolutions
    int cap_boards = 0;               // boards the activation arrays hold (>= chunk_boards, see layers_reserve)
    std::vector<void*> allocs;
    PackedConv stem, head;
    std::vector<PackedConv> convs;     // 2 per block
    std::vector<float*> se;            // per block: w1, b1, w2, b2 (float32), empty without squeeze-excite
    char *f64 = nullptr, *x = nullptr, *h = nullptr, *y = nullptr;   // chunk activations, each followed by a zero region
    size_t zero_off_f = 0, zero_off_c = 0;
};

static const int kZeroTail = 4096;

template <typename T>
static int dev_up(LayerNet* m, const T* src, size_t count, T** out) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return fail(AZ_ERR_HIP, "hipMalloc (network weights)");
    m->allocs.push_back(p);
    if (count && hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return fail(AZ_ERR_HIP, "hipMemcpy (network weights)");
    *out = (T*)p;
    return AZ_OK;
}

// Folded weights w[cout][cin][taps] -> one packed run per block of `ncols` output channels.
static int pack_layer(LayerNet* m, const std::vector<float>& w, const std::vector<float>& b, int cout, int cin, int taps, int cin_pad,
                      int ncols, PackedConv* out) {
    const bool slab = taps == 9 && cin_pad % 64 == 0;
    std::vector<uint16_t> all;
    size_t block = 0;
    for (int c0 = 0; c0 < cout; c0 += ncols) {
        const int nc = std::min(ncols, cout - c0);
        std::vector<float> sub(w.begin() + (size_t)c0 * cin * taps, w.begin() + (size_t)(c0 + nc) * cin * taps);
        // 192 output channels in one block: k_conv3_duo, 3 tiles per wave, the third with its rows in natural channel order
        const bool duo3 = slab && ncols == 192 && cout == 192;
        out->duo = duo3;
        std::vector<uint16_t> pk = pack_conv(sub, nc, cin, taps, cin_pad, ncols, slab, duo3 ? 3 : 0);
        block = pk.size() * 2 / 16;
        all.insert(all.end(), pk.begin(), pk.end());
    }
    std::vector<float> bias((size_t)((cout + ncols - 1) / ncols) * ncols, 0.f);
    std::copy(b.begin(), b.end(), bias.begin());
    uint16_t* wd;
    int rc = dev_up(m, all.data(), all.size(), &wd);
    if (rc != AZ_OK) return rc;
    rc = dev_up(m, bias.data(), bias.size(), &out->b);
    if (rc != AZ_OK) return rc;
    out->w = (uint4*)wd; out->block_vecs = block; out->cin_pad = cin_pad; out->cout = cout; out->ncols = ncols; out->taps = taps;
    out->slab = slab;
    return AZ_OK;
}

bool layers_supported(const az_net_config& c) {
    const bool cok = c.channels == 64 || c.channels == 128 || c.channels == 192 || c.channels == 256;
    return cok && c.board_size >= 2 && c.board_size <= 19 && c.in_planes <= 32;
}

void layers_destroy(LayerNet* m) {
    if (!m) return;
    (void)hipSetDevice(m->cfg.device);
    (void)hipDeviceSynchronize();
    for (void* p : m->allocs) (void)hipFree(p);
    delete m;
}

int layers_chunk_boards(const LayerNet* m) { return m->chunk_boards; }
int layers_cap_boards(const LayerNet* m) { return m->cap_boards; }

// Activation arrays for passes of up to `boards` boards (at most what 31-bit byte offsets reach; never shrinks). A direct
// forward still runs in chunks of chunk_boards; an indirect one covers its upper bound in passes of cap_boards, so a
// caller that reserves its bound gets one pass. The arrays in use stay allocated until the network is destroyed (a launch
// in flight may still read them).
int layers_reserve(LayerNet* m, int boards) {
    const int C = m->cfg.channels, np = m->np;
    const int limit = (int)((((size_t)1 << 31) - ((size_t)1 << 20)) / ((size_t)np * C * 2)) - 1;   // 32-bit byte offsets, halo rows included
    boards = std::min(std::max(boards, m->chunk_boards), std::max(limit, m->chunk_boards));
    if (boards <= m->cap_boards) return AZ_OK;
    if (hipSetDevice(m->cfg.device) != hipSuccess) return fail(AZ_ERR_HIP, "hipSetDevice");
    const size_t rows = (size_t)boards * np;
    const size_t zf = (rows * 128 + 255) / 256 * 256, zc = (rows * (size_t)C * 2 + 255) / 256 * 256;
    char* fresh[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i) {
        const size_t bytes = (i == 0 ? zf : zc) + kZeroTail;
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            for (int k = 0; k < i; ++k) (void)hipFree(fresh[k]);
            return fail(AZ_ERR_HIP, "hipMalloc (layer activations)");
        }
        fresh[i] = (char*)p;
        if (hipMemset(p, 0, bytes) != hipSuccess) {
            for (int k = 0; k <= i; ++k) (void)hipFree(fresh[k]);
            return fail(AZ_ERR_HIP, "hipMemset (layer activations)");
        }
    }
    for (int i = 0; i < 4; ++i) m->allocs.push_back(fresh[i]);
    m->f64 = fresh[0]; m->x = fresh[1]; m->h = fresh[2]; m->y = fresh[3];
    m->zero_off_f = zf; m->zero_off_c = zc;
    m->cap_boards = boards;
    return AZ_OK;
}

int layers_create(const az_net_config& c, const NetLayout& L, const float* Pm, LayerNet** out) {
    if (!layers_supported(c)) return fail(AZ_ERR_ARG, "bf16 layer path: unsupported network shape");
    if (hipSetDevice(c.device) != hipSuccess) return fail(AZ_ERR_HIP, "hipSetDevice");
    LayerNet* m = new LayerNet();
    m->cfg = c;
    const int C = c.channels, np = c.board_size * c.board_size, q = C / 4;
    m->np = np; m->q = q;
    m->ncols = C;   // 256 channels: k_conv3_duo (one block of all channels); below: k_conv3_slab<C>
    // boards per chunk: activation arrays of at most ~96 MiB each (measured at 19x19 256 channels: 32 MiB -6 %, 96-128 MiB +2 % over 64)
    m->chunk_boards = std::max(16, (int)(((size_t)96 << 20) / ((size_t)np * C * 2)));
    if (const char* e = getenv("AZ_LAYER_CHUNK_BOARDS")) {   // diagnostics: another chunk size (byte offsets stay below 2 GiB)
        const int v = atoi(e), cap = (int)((((size_t)1 << 31) - ((size_t)1 << 20)) / ((size_t)np * C * 2)) - 1;
        if (v >= 16) m->chunk_boards = std::min(v, cap);
    }
    std::vector<float> w, b;
    int rc;
#define TRY(x) if ((rc = (x)) != AZ_OK) { layers_destroy(m); return rc; }
    fold_bn(Pm, L.stem_w, L.stem_bn, C, (int64_t)c.in_planes * 9, w, b);
    TRY(pack_layer(m, w, b, C, c.in_planes, 9, 64, m->ncols, &m->stem));
    for (int i = 0; i < c.blocks; ++i) {
        const BlockOff& bo = L.blocks[i];
        PackedConv pc;
        fold_bn(Pm, bo.conv1, bo.bn1, C, (int64_t)C * 9, w, b);
        TRY(pack_layer(m, w, b, C, C, 9, C, m->ncols, &pc));
        m->convs.push_back(pc);
        fold_bn(Pm, bo.conv2, bo.bn2, C, (int64_t)C * 9, w, b);
        TRY(pack_layer(m, w, b, C, C, 9, C, m->ncols, &pc));
        m->convs.push_back(pc);
        if (c.use_se) {
            const int hid = c.se_hidden;
            float* d;
            TRY(dev_up(m, Pm + bo.se_w1, (size_t)hid * C, &d)); m->se.push_back(d);
            TRY(dev_up(m, Pm + bo.se_b1, (size_t)hid, &d)); m->se.push_back(d);
            TRY(dev_up(m, Pm + bo.se_w2, (size_t)C * hid, &d)); m->se.push_back(d);
            TRY(dev_up(m, Pm + bo.se_b2, (size_t)C, &d)); m->se.push_back(d);
        }
    }
    {   // fused head 1x1: channels [0,q) policy_head.0, [q,2q) value_head.0
        std::vector<float> wp, bp, wv, bv, wh((size_t)2 * q * C), bh((size_t)2 * q);
        fold_bn(Pm, L.ph0_w, L.ph0_bn, q, C, wp, bp);
        fold_bn(Pm, L.vh0_w, L.vh0_bn, q, C, wv, bv);
        std::copy(wp.begin(), wp.end(), wh.begin());
        std::copy(wv.begin(), wv.end(), wh.begin() + (size_t)q * C);
        std::copy(bp.begin(), bp.end(), bh.begin());
        std::copy(bv.begin(), bv.end(), bh.begin() + q);
        TRY(pack_layer(m, wh, bh, 2 * q, C, 1, C, 2 * q, &m->head));
    }
    TRY(layers_reserve(m, m->chunk_boards));
#undef TRY
    *out = m;
    return AZ_OK;
}

template <int KPT, int NCOLS, int WM, int WN>
static void launch_one(const ConvArgs& a, int cout, hipStream_t st) {
    const dim3 grid(std::min((a.rows + 255) / 256, kPersistGrid), (cout + NCOLS - 1) / NCOLS);
    hipLaunchKernelGGL((k_conv_layer<KPT, NCOLS, WM, WN>), grid, dim3(64 * WM * WN), (size_t)2 * 64 * NCOLS * 2, st, a);
}

template <int NCOLS>
static int launch_slab(const ConvArgs& a, int cout, hipStream_t st) {
    const size_t lds = (size_t)2 * kSlabBytes + 1024 + (size_t)2 * 64 * NCOLS * 2;
    static bool opted = false;   // the large dynamic LDS size is opted into once per kernel
    if (!opted) {
        if (hipFuncSetAttribute((const void*)k_conv3_slab<NCOLS, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(AZ_ERR_HIP, "hipFuncSetAttribute (k_conv3_slab)");
        opted = true;
    }
    ConvArgs b = a;
    b.col_blocks = (cout + NCOLS - 1) / NCOLS;
    const int row_tiles = (a.rows + 255) / 256;
    const dim3 grid((unsigned)std::min((row_tiles + 7) / 8 * 8 * b.col_blocks, kPersistGrid));   // both multiples of 8 * col_blocks
    hipLaunchKernelGGL((k_conv3_slab<NCOLS, 4, 2>), grid, dim3(512), lds, st, b);
    return AZ_OK;
}

template <int C>
static int launch_duo(const ConvArgs& a, hipStream_t st) {
    constexpr int kDuoRows = DuoRows<C>::value;
    const int row_tiles = (a.rows + kDuoRows - 1) / kDuoRows;
    hipLaunchKernelGGL((k_conv3_duo<C>), dim3((unsigned)std::min((row_tiles + 7) / 8 * 8, kPersistGrid)), dim3(256), (size_t)DuoGeo<kDuoRows>::kLds, st, a);
    return AZ_OK;
}

// Tower convolutions: 4x2 waves of 64 positions x NCOLS/2 channels. Heads: 8x1 waves of 32 positions x all channels.
static int launch_conv(const PackedConv& pc, const char* in, int zero_off, char* out, int out_stride, int rows, int n, int relu, bool head,
                       hipStream_t st, const int* rows_dev, int board_off) {
    ConvArgs a;
    a.rows_dev = rows_dev; a.board_off = board_off;
    a.in = in; a.in_stride = pc.cin_pad * 2; a.zero_off = zero_off; a.w = pc.w; a.w_block_vecs = pc.block_vecs; a.bias = pc.b;
    a.out = out; a.out_stride = out_stride; a.rows = rows; a.n = n; a.np = n * n; a.ntaps = pc.taps; a.relu = relu; a.col_blocks = 1;
    const int key = (pc.cin_pad / 32) * 1000 + pc.ncols;
    if (pc.slab && n + 1 <= kMaxHalo) {
        if (pc.ncols == 256 && pc.cout == 256) return launch_duo<256>(a, st);
        if (pc.ncols == 192 && pc.cout == 192 && pc.duo) return launch_duo<192>(a, st);
        switch (pc.ncols) {
            case 64: return launch_slab<64>(a, pc.cout, st);
            case 128: return launch_slab<128>(a, pc.cout, st);
            default: return fail(AZ_ERR_ARG, "bf16 layer path: no slab kernel for this shape");
        }
    }
    if (!head) {
        switch (key) {
            case 2064: launch_one<2, 64, 4, 2>(a, pc.cout, st); break;     // stem or 64-channel tower
            case 2128: launch_one<2, 128, 4, 2>(a, pc.cout, st); break;    // stem of the 128- and 256-channel towers
            case 2192: launch_one<2, 192, 4, 2>(a, pc.cout, st); break;    // stem of the 192-channel tower
            case 4128: launch_one<4, 128, 4, 2>(a, pc.cout, st); break;
            case 6192: launch_one<6, 192, 4, 2>(a, pc.cout, st); break;
            case 8128: launch_one<8, 128, 4, 2>(a, pc.cout, st); break;
            default: return fail(AZ_ERR_ARG, "bf16 layer path: no convolution kernel for this shape");
        }
    } else {
        switch (key) {
            case 2032: launch_one<2, 32, 8, 1>(a, pc.cout, st); break;
            case 4064: launch_one<4, 64, 8, 1>(a, pc.cout, st); break;
            case 6096: launch_one<6, 96, 8, 1>(a, pc.cout, st); break;
            case 8128: launch_one<8, 128, 8, 1>(a, pc.cout, st); break;
            default: return fail(AZ_ERR_ARG, "bf16 layer path: no head kernel for this shape");
        }
    }
    return AZ_OK;
}

// Boards [0, boards) of `feat` ([boards][np][32] bf16) -> hq [boards][np][2q] bf16; boards <= chunk_boards.
// rows_dev != null: `feat` is the chunk that starts at board `board_off` of a batch whose board count is *rows_dev on the
// device; the grids are sized for `boards` and the workgroups past the live boards of the chunk leave at once.
int layers_tower(LayerNet* m, const uint16_t* feat, int boards, uint16_t* hq, hipStream_t st, const int* rows_dev, int board_off) {
    const az_net_config& c = m->cfg;
    const int C = c.channels, n = c.board_size, np = m->np, rows = boards * np;
    if (boards <= 0) return AZ_OK;
    if (boards > m->cap_boards) return fail(AZ_ERR_ARG, "layers_tower: more boards than the activation arrays hold");
    hipLaunchKernelGGL(k_pad_features, dim3(std::min((rows * 8 + 255) / 256, kPersistGrid)), dim3(256), 0, st, (const uint4*)feat, (uint4*)m->f64, rows, np, rows_dev, board_off);
    int rc = launch_conv(m->stem, m->f64, (int)m->zero_off_f, m->x, C * 2, rows, n, 1, false, st, rows_dev, board_off);
    if (rc != AZ_OK) return rc;
    const int per_row = C / 8;
    const dim3 egrid((unsigned)std::min<size_t>(((size_t)rows * per_row + 255) / 256, 2 * kPersistGrid));
    for (int i = 0; i < c.blocks; ++i) {
        rc = launch_conv(m->convs[2 * i], m->x, (int)m->zero_off_c, m->h, C * 2, rows, n, 1, false, st, rows_dev, board_off);
        if (rc != AZ_OK) return rc;
        rc = launch_conv(m->convs[2 * i + 1], m->h, (int)m->zero_off_c, m->y, C * 2, rows, n, 0, false, st, rows_dev, board_off);
        if (rc != AZ_OK) return rc;
        if (c.use_se)
            hipLaunchKernelGGL(k_se_residual_layer, dim3(std::min(boards, kPersistGrid)), dim3(256), ((size_t)(256 / (C / 8)) * C + 2 * C + ((c.se_hidden + 3) & ~3)) * 4, st,
                               (const uint16_t*)m->y, np, C, c.se_hidden, m->se[4 * i], m->se[4 * i + 1], m->se[4 * i + 2], m->se[4 * i + 3], (uint16_t*)m->x, boards, rows_dev,
                               board_off);
        else
            hipLaunchKernelGGL(k_scale_residual, egrid, dim3(256), 0, st, (const uint4*)m->y, (const float*)nullptr, (uint4*)m->x, rows, np, C, rows_dev, board_off);
    }
    rc = launch_conv(m->head, m->x, (int)m->zero_off_c, (char*)hq, 2 * m->q * 2, rows, n, 1, true, st, rows_dev, board_off);
    if (rc != AZ_OK) return rc;
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(AZ_ERR_HIP, std::string("bf16 layer path: ") + hipGetErrorString(err));
    return AZ_OK;
}

}  // namespace az
