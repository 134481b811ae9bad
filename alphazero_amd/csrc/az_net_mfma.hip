// bf16 MFMA inference of the policy/value network (model.py:43-97, eval mode) for gfx950.
//
// k_tower: ONE launch runs the stem, every residual block (with squeeze-excite) and the two 1x1
// head convolutions. A 512-thread workgroup owns TB whole boards (TB*n*n <= 256 positions, 3 boards
// on 9x9); their activations stay in LDS ([position][channel] bf16) from the first layer to the last
// and the residual stream stays in registers, so the only HBM/L2 traffic of the tower is the weight
// stream. Each 3x3 convolution is an implicit GEMM: M = 256 positions, N = C output channels,
// K = 9*C, computed with v_mfma_f32_16x16x32_bf16 by 8 waves (each wave 64 positions x C/2 channels, or
// 32 x C). The weights are the MFMA's row operand, so a lane ends up with consecutive output channels
// of one position and writes them back with 16-byte LDS stores. Board edges are handled by pointing
// out-of-board taps at a zero row in LDS. Weights are packed at load time into the exact LDS image the
// fragments are read from, tile after tile, and streamed global -> LDS (global_load_lds, no staging
// registers) ahead of the MFMAs.
// The SE pooling adds a board's positions in index order (one thread per board and channel), so the
// output of a position does not depend on which other boards share its workgroup or launch.
//
// k_policy_tail / k_value_tail finish the heads (tiny next to the tower).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "az_net.h"
#include "az_mfma_gemm.cuh"

// k-steps of weight fragments in flight in the kernels that read them from global memory (the 192-channel ones: 2, or 5 in
// the 96-row 256-thread form; see their instantiations)
#ifndef AZ_SE_LDS96
#define AZ_SE_LDS96 1     // 0: the 96-row 256-thread form reads its squeeze-excite weights from global memory (A/B builds)
#endif
#ifndef AZ_GW128
#define AZ_GW128 1       // 128 channels, 256- and 384-row kernels (3 measured the same on one box: 916 / 916 TFLOP/s at 4096 boards of 9x9)
#endif
#ifndef AZ_GW128S
#define AZ_GW128S 3      // 128 channels, 96-row variant: a lone workgroup per compute unit (one board 0.271 -> 0.182 ms, 256 boards 0.286 -> 0.202)
#endif
#ifndef AZ_GW128D
#define AZ_GW128D 3      // 128 channels, two 176-row workgroups per compute unit
#endif
#ifndef AZ_GW64S
#define AZ_GW64S 3       // 64 channels, 96-row variant (256 boards 0.061 -> 0.058 ms)
#endif
#ifndef AZ_RING
#define AZ_RING 6        // activation-fragment ring of the two-workgroups-per-CU tower (gemm_conv_ring)
#endif

namespace az {
int fail(int code, const std::string& msg);


struct TowerParams {
    int n, np, TB, rows, blocks, use_se, hid, cin_pad, q;
    unsigned recip_np;           // floor(row / np) == (row * recip_np) >> 16 for every row of a workgroup (checked at creation)
    unsigned recip_n;            // floor(p / n) == (p * recip_n) >> 16 for p < 512
    const uint16_t* feat;        // [rows][np][cin_pad] bf16
    const uint4* w_stem;         // packed tiles
    const float* b_stem;         // [C]
    const uint4* w_tower;        // [blocks][2] packed convs
    const float* b_tower;        // [blocks][2][C]
    const uint4* se_blob;        // [blocks] packed {w1 bf16 [hid][C], w2^T bf16 [hid][C], b1 f32 [hid], b2 f32 [C]}, padded to 1 KiB
    int se_vecs;                 // 16-byte vectors per block in se_blob (multiple of 64)
    const uint4* w_head;         // packed 1x1, NH columns (policy q | value q | zero pad)
    const float* b_head;         // [NH]
    uint16_t* hq;                // out: [rows][np][2q] bf16, relu(bn(conv1x1)) of both heads
    uint4* xres;                 // residual stream scratch: [gridDim.x][MI*NI*2][threads] 16-byte vectors, private per thread
    int items;                   // board groups (of TB boards) to process; workgroups stride over them
    const int* rows_dev;         // if set, the row count is read here (rows is then an upper bound)
    int gate_mode, gate_rows;    // with rows_dev: 1 = this launch works only if the count is <= gate_rows, 2 = only if it is above
                                 // (two launches, one per kernel variant; the one the count does not select leaves at once)
    unsigned long long* stamps;  // diagnostic build (-DAZ_STAMP) only: per-workgroup phase cycle sums
    int lds_bytes;               // diagnostic build (-DAZ_POISON) only: dynamic LDS of the launch, filled with a NaN pattern first
    int* owner;                  // diagnostic build (-DAZ_POISON) only: [grid + 1] who holds each residual-scratch slot; [grid] counts clashes
};

#ifdef AZ_POISON
static int* g_owner_dev = nullptr;   // owner table of the last tower launch (its last word counts slot clashes)
static int g_owner_grid = 0;
#endif
#ifdef AZ_CLOCK
static unsigned long long* g_clock_dev = nullptr;   // [grid][2] of the last tower launch: shader cycles, 100 MHz ticks per workgroup
static int g_clock_grid = 0;
#endif
#ifdef AZ_STAMP
#define STAMP(slot)                                                                    \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        unsigned long long t_;                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        stamp_acc[slot] += t_ - stamp_last;                                            \
        stamp_last = t_;                                                               \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif


// act <- bf16(f(acc + bias)) with f = relu or identity (bias may be null); optionally park the packed values
// in xres (the residual stream: each thread's private slice). One 16-byte LDS store per tile pair.
template <int MI, int NI, int WM, int WN, int STRIDE, bool RELU, bool KEEP>
__device__ __forceinline__ void store_act(const f32x4 (&acc)[MI][NI], const float* __restrict__ bias, char* act, uint4* xres_wg) {
    constexpr int NT = 64 * WM * WN;
    AZ_LANE_GEO(MI, NI, WN);
    __syncthreads();  // every wave has finished reading act
    char* wbase = act + (row_base + lr) * STRIDE + (col0 + 8 * kq) * 2;
    uint4* xr = xres_wg + t_;
#pragma unroll
    for (int p = 0; p < NI / 2; ++p) {
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const float4 x0 = *(const float4*)(bias + col0 + 32 * p + 8 * kq), x1 = *(const float4*)(bias + col0 + 32 * p + 8 * kq + 4);
            bv[0] = x0.x; bv[1] = x0.y; bv[2] = x0.z; bv[3] = x0.w; bv[4] = x1.x; bv[5] = x1.y; bv[6] = x1.z; bv[7] = x1.w;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mi][2 * p][r] + bv[r]; v[4 + r] = acc[mi][2 * p + 1][r] + bv[4 + r]; }
            if (RELU) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            const uint4 pk = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
            *(uint4*)(wbase + mi * 16 * STRIDE + p * 64) = pk;
            if (KEEP) xr[(size_t)(mi * (NI / 2) + p) * NT] = pk;
        }
    }
    if constexpr (NI & 1) {   // the unpaired last tile: 4 consecutive channels per lane (natural row order), 8-byte stores
        constexpr int e = NI - 1, P2 = NI / 2;
        const int ch4 = col0 + 32 * P2 + 4 * kq;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) { const float4 x0 = *(const float4*)(bias + ch4); bv[0] = x0.x; bv[1] = x0.y; bv[2] = x0.z; bv[3] = x0.w; }
        uint2 hold = make_uint2(0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mi][e][r] + bv[r]; if (RELU) v[r] = fmaxf(v[r], 0.f); }
            const uint2 pk = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
            *(uint2*)(act + (row_base + mi * 16 + lr) * STRIDE + ch4 * 2) = pk;
            if (KEEP) {   // two tile rows share one 16-byte slot of the residual scratch (an odd last row has one to itself)
                if (mi & 1) xr[(size_t)(MI * P2 + mi / 2) * NT] = make_uint4(hold.x, hold.y, pk.x, pk.y);
                else if (mi == MI - 1) xr[(size_t)(MI * P2 + mi / 2) * NT] = make_uint4(pk.x, pk.y, 0, 0);
                else hold = pk;
            }
        }
    }
    __syncthreads();
}

// Squeeze-excite gate of one block (model.py:9-22) applied to acc in place. The bf16 image of y is
// in act; pooled/hidden/scale and the block's SE weights live in `scratch` (the idle weight buffers).
// BLOB_GLOBAL: the block's squeeze-excite weights are read where they lie in global memory instead of being staged in LDS
// (for workgroups whose LDS has no room for a staging area). Staging them into the activation image once the pooling
// has read it - the image is dead from then until the residual store - was measured 1-2 % slower than reading them from
// global memory: the extra barrier and the exposed load latency cost more than the faster fc2 gains while the compute
// unit's other workgroup keeps the matrix pipes busy anyway.
template <int MI, int NI, int WM, int WN, int C, int STRIDE, bool BLOB_GLOBAL = false>
__device__ __forceinline__ void se_gate(f32x4 (&acc)[MI][NI], const char* act, float* scratch, const uint4* __restrict__ blob, int se_vecs,
                                        int hid, int TB, int nb, const TileGeo& g
#ifdef AZ_STAMP
                                        , unsigned long long* stamp_acc, unsigned long long& stamp_last
#endif
                                        ) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    AZ_LANE_GEO(MI, NI, WN);
    const int NP = g.np;
    float* pooled = scratch;                 // [TB][C]
    float* sc = pooled + TB * C;             // [TB][C], 16-byte aligned (C is a multiple of 4)
    float* hidden = sc + TB * C;             // [TB][hid]
    const char* sew = (const char*)blob;     // this block's SE weights
    if constexpr (!BLOB_GLOBAL) {
        char* stage = (char*)(hidden + TB * hid);
        stage += (16 - ((size_t)(stage - (char*)scratch) & 15)) & 15;
        stage_tile<NW>(blob, stage, se_vecs, lane, wave);
        sew = stage;
    }
    const uint16_t* w1 = (const uint16_t*)sew;   // [hid][C] bf16
    const uint16_t* w2t = w1 + hid * C;          // [hid][C] bf16 (fc2 weight transposed)
    const float* b1 = (const float*)(w2t + hid * C);
    const float* b2 = b1 + hid;
    // Pooling: one thread per (board, channel pair) adds the board's positions in index order, so the
    // pooled value is bit-identical whichever slot of whichever workgroup the board occupies.
    // (Splitting a board's positions over four lanes with 16-byte reads and a fixed combining tree, and fc2's hidden units
    // over four lanes likewise, was measured 1-2 % slower end to end in the two-per-CU kernel: these phases run in the
    // shadow of the compute unit's other workgroup, and the wider versions only add LDS and issue pressure.)
    const float inv_np = 1.0f / (float)NP;
    for (int i = t_; i < nb * (C / 2); i += NT) {
        const int b = i / (C / 2), cp = i - b * (C / 2);
        const char* src = act + (size_t)b * NP * STRIDE + cp * 4;
        float s0 = 0.f, s1 = 0.f;
        int p = 0;
        for (; p + 9 <= NP; p += 9) {   // loads batched nine at a time, adds in position order
            uint32_t v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = *(const uint32_t*)(src + (p + k) * STRIDE);
#pragma unroll
            for (int k = 0; k < 9; ++k) { s0 += bf16_lo(v[k]); s1 += bf16_hi(v[k]); }
        }
        for (; p < NP; ++p) {
            const uint32_t v = *(const uint32_t*)(src + p * STRIDE);
            s0 += bf16_lo(v); s1 += bf16_hi(v);
        }
        pooled[b * C + 2 * cp] = s0 * inv_np;
        pooled[b * C + 2 * cp + 1] = s1 * inv_np;
    }
    STAMP(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // SE weights have landed
    __syncthreads();
    STAMP(9);
    {   // fc1 + relu: 8 lanes per output, fixed reduction tree
        const int sub = t_ & 7;
        for (int o = t_ >> 3; o < nb * hid; o += NT / 8) {
            const int b = o / hid, j = o - b * hid;
            float s = 0.f;
            if constexpr ((C / 8) % 8 == 0) {   // a contiguous run of C/8 channels per lane, 16-byte LDS reads
                const uint16_t* wr = w1 + j * C + sub * (C / 8);
                const float* pr = pooled + b * C + sub * (C / 8);
#pragma unroll
                for (int c = 0; c < C / 8; c += 8) {
                    const uint4 w = *(const uint4*)(wr + c);
                    const float4 p0 = *(const float4*)(pr + c), p1 = *(const float4*)(pr + c + 4);
                    s = fmaf(bf16_lo(w.x), p0.x, s); s = fmaf(bf16_hi(w.x), p0.y, s);
                    s = fmaf(bf16_lo(w.y), p0.z, s); s = fmaf(bf16_hi(w.y), p0.w, s);
                    s = fmaf(bf16_lo(w.z), p1.x, s); s = fmaf(bf16_hi(w.z), p1.y, s);
                    s = fmaf(bf16_lo(w.w), p1.z, s); s = fmaf(bf16_hi(w.w), p1.w, s);
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < C; ch += 8) s = fmaf(__uint_as_float((uint32_t)w1[j * C + ch + sub] << 16), pooled[b * C + ch + sub], s);
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if (sub == 0) hidden[o] = fmaxf(s + b1[j], 0.f);
        }
    }
    __syncthreads();
    STAMP(10);
    if constexpr (BLOB_GLOBAL) {
        // fc2 + sigmoid with the weights in global memory: a thread takes 8 consecutive channels of one board and reads
        // a 16-byte vector per hidden unit, four of them in flight (the sums keep their order in j)
        for (int i = t_; i < nb * (C / 8); i += NT) {
            const int b = i / (C / 8), c8 = (i - b * (C / 8)) * 8;
            float s[8];
            {
                const float4 x0 = *(const float4*)(b2 + c8), x1 = *(const float4*)(b2 + c8 + 4);
                s[0] = x0.x; s[1] = x0.y; s[2] = x0.z; s[3] = x0.w; s[4] = x1.x; s[5] = x1.y; s[6] = x1.z; s[7] = x1.w;
            }
            const uint16_t* wr = w2t + c8;
            int j = 0;
            for (; j + 4 <= hid; j += 4) {
                uint4 w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = *(const uint4*)(wr + (size_t)(j + k) * C);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float h = hidden[b * hid + j + k];
                    s[0] = fmaf(bf16_lo(w[k].x), h, s[0]); s[1] = fmaf(bf16_hi(w[k].x), h, s[1]);
                    s[2] = fmaf(bf16_lo(w[k].y), h, s[2]); s[3] = fmaf(bf16_hi(w[k].y), h, s[3]);
                    s[4] = fmaf(bf16_lo(w[k].z), h, s[4]); s[5] = fmaf(bf16_hi(w[k].z), h, s[5]);
                    s[6] = fmaf(bf16_lo(w[k].w), h, s[6]); s[7] = fmaf(bf16_hi(w[k].w), h, s[7]);
                }
            }
            for (; j < hid; ++j) {
                const uint4 w = *(const uint4*)(wr + (size_t)j * C);
                const float h = hidden[b * hid + j];
                s[0] = fmaf(bf16_lo(w.x), h, s[0]); s[1] = fmaf(bf16_hi(w.x), h, s[1]);
                s[2] = fmaf(bf16_lo(w.y), h, s[2]); s[3] = fmaf(bf16_hi(w.y), h, s[3]);
                s[4] = fmaf(bf16_lo(w.z), h, s[4]); s[5] = fmaf(bf16_hi(w.z), h, s[5]);
                s[6] = fmaf(bf16_lo(w.w), h, s[6]); s[7] = fmaf(bf16_hi(w.w), h, s[7]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) sc[b * C + c8 + e] = 1.f / (1.f + __expf(-s[e]));
        }
    } else
    for (int i = t_; i < nb * C; i += NT) {   // fc2 + sigmoid
        const int b = i / C, ch = i - b * C;
        float s = b2[ch];
        int j = 0;
        for (; j + 8 <= hid; j += 8) {
            float w[8], hv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { w[k] = __uint_as_float((uint32_t)w2t[(j + k) * C + ch] << 16); hv[k] = hidden[b * hid + j + k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) s = fmaf(w[k], hv[k], s);
        }
        for (; j < hid; ++j) s = fmaf(__uint_as_float((uint32_t)w2t[j * C + ch] << 16), hidden[b * hid + j], s);
        sc[i] = 1.f / (1.f + __expf(-s));
    }
    __syncthreads();
    STAMP(11);
    // every lane owns one position per tile row, so the board (and its gate vector) is per lane and tile row
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = row_base + mi * 16 + lr;
        const int b = min((int)((row * g.recip_np) >> 16), nb - 1);   // rows past the valid ones are never read back
        const float* sb = sc + b * C + col0 + 8 * kq;
#pragma unroll
        for (int p = 0; p < NI / 2; ++p) {
            const float4 s0 = *(const float4*)(sb + 32 * p), s1 = *(const float4*)(sb + 32 * p + 4);
            acc[mi][2 * p][0] *= s0.x; acc[mi][2 * p][1] *= s0.y; acc[mi][2 * p][2] *= s0.z; acc[mi][2 * p][3] *= s0.w;
            acc[mi][2 * p + 1][0] *= s1.x; acc[mi][2 * p + 1][1] *= s1.y; acc[mi][2 * p + 1][2] *= s1.z; acc[mi][2 * p + 1][3] *= s1.w;
        }
        if constexpr (NI & 1) {
            const float4 s0 = *(const float4*)(sc + b * C + col0 + 32 * (NI / 2) + 4 * kq);
            acc[mi][NI - 1][0] *= s0.x; acc[mi][NI - 1][1] *= s0.y; acc[mi][NI - 1][2] *= s0.z; acc[mi][NI - 1][3] *= s0.w;
        }
    }
}

// x = relu(y + x) (model.py:40): act <- bf16(x); the new residual goes back to xres when `keep`.
template <int MI, int NI, int WM, int WN, int STRIDE>
__device__ __forceinline__ void residual_store(const f32x4 (&acc)[MI][NI], const uint4 (&resv)[(MI * NI + 1) / 2], char* act, uint4* xres_wg, bool keep) {
    constexpr int NT = 64 * WM * WN;
    AZ_LANE_GEO(MI, NI, WN);
    __syncthreads();
    char* wbase = act + (row_base + lr) * STRIDE + (col0 + 8 * kq) * 2;
    uint4* xr = xres_wg + t_;
#pragma unroll
    for (int p = 0; p < NI / 2; ++p) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const uint4 o = resv[mi * (NI / 2) + p];
            const uint32_t old[4] = {o.x, o.y, o.z, o.w};
            uint32_t pk[4];
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                pk[r >> 1] = pack_bf16(fmaxf(acc[mi][2 * p][r] + bf16_lo(old[r >> 1]), 0.f), fmaxf(acc[mi][2 * p][r + 1] + bf16_hi(old[r >> 1]), 0.f));
                pk[2 + (r >> 1)] = pack_bf16(fmaxf(acc[mi][2 * p + 1][r] + bf16_lo(old[2 + (r >> 1)]), 0.f),
                                             fmaxf(acc[mi][2 * p + 1][r + 1] + bf16_hi(old[2 + (r >> 1)]), 0.f));
            }
            const uint4 v = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            *(uint4*)(wbase + mi * 16 * STRIDE + p * 64) = v;
            if (keep) xr[(size_t)(mi * (NI / 2) + p) * NT] = v;
        }
    }
    if constexpr (NI & 1) {
        constexpr int e = NI - 1, P2 = NI / 2;
        const int ch4 = col0 + 32 * P2 + 4 * kq;
        uint2 hold = make_uint2(0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const uint4 o = resv[MI * P2 + mi / 2];
            const uint32_t o0 = (mi & 1) ? o.z : o.x, o1 = (mi & 1) ? o.w : o.y;
            const uint2 pk = make_uint2(pack_bf16(fmaxf(acc[mi][e][0] + bf16_lo(o0), 0.f), fmaxf(acc[mi][e][1] + bf16_hi(o0), 0.f)),
                                        pack_bf16(fmaxf(acc[mi][e][2] + bf16_lo(o1), 0.f), fmaxf(acc[mi][e][3] + bf16_hi(o1), 0.f)));
            *(uint2*)(act + (row_base + mi * 16 + lr) * STRIDE + ch4 * 2) = pk;
            if (keep) {
                if (mi & 1) xr[(size_t)(MI * P2 + mi / 2) * NT] = make_uint4(hold.x, hold.y, pk.x, pk.y);
                else if (mi == MI - 1) xr[(size_t)(MI * P2 + mi / 2) * NT] = make_uint4(pk.x, pk.y, 0, 0);
                else hold = pk;
            }
        }
    }
    __syncthreads();
}

// C channels; NW = WM x WN waves of 64 positions-by-C/WN channels; NH = padded channel count of the fused
// head 1x1 convolution, which all NW waves split by positions only. One workgroup per compute unit
// (LDS-bound), striding over the board groups.
// GW > 0: the residual blocks' convolutions read their weight fragments straight from global memory, GW k-steps
// ahead (gemm_conv_gw: no LDS staging, no barrier inside the loop).
// ROWS: positions a workgroup holds (256: TB = 256/np boards; 96: the low-latency variant for small batches, one 9x9 board
// per workgroup, so that a handful of positions spreads over as many compute units instead of queueing on one).
// LEAN: no weight staging buffers and no staged squeeze-excite weights in LDS (stem and head read their weight fragments
// from global memory too), which leaves room for OCC / ((WM*WN)/4) workgroups per compute unit: two 256-thread workgroups
// of two boards each run their phases independently, so one's epilogues overlap the other's convolutions.
template <int C, int WM, int WN, int NH, int GW = 0, int ROWS = 256, bool LEAN = false, int OCC = (WM * WN) / 4>
__global__ __launch_bounds__(64 * WM * WN, OCC) void k_tower(TowerParams P) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int MI = ROWS / (WM * 16), NI = C / (WN * 16);
    constexpr int MIH = (ROWS + NW * 16 - 1) / (NW * 16), NIH = NH / 16;   // head: the waves split the row tiles; waves past ROWS compute zeros
    constexpr int STRIDE = 2 * C + 32;
    // The 96-row LEAN form has LDS to spare and mostly runs alone on its compute unit (few positions): its squeeze-excite
    // weights are staged in LDS like the 512-thread kernels' (reading them from global memory, which the 176-row form hides
    // behind the compute unit's other workgroup, cost 9 % of a lone workgroup's cycles in fc2 alone).
    constexpr bool SE_LDS = LEAN && ROWS == 96 && AZ_SE_LDS96;
    constexpr int RING = (MI > AZ_RING && ((C / 32) * MI) % AZ_RING == 0) ? AZ_RING : 4;   // activation-fragment ring of gemm_conv_ring: fewer than the row tiles a wave owns
    constexpr int NV = MI * (NI / 2) + ((NI & 1) ? (MI + 1) / 2 : 0);  // 16-byte vectors of packed bf16 residual per thread
    static_assert(ROWS % (WM * 16) == 0 && NIH % 2 == 0, "whole row tiles per wave; head channel tiles are stored in pairs");
    static_assert(NW * 16 * MIH * (2 * NH + 16) <= ROWS * STRIDE, "the head activations of every wave's row tiles must fit the activation image");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* act = smem;
    constexpr int ZBYTES = ((256 + 2 * C + 16 + 127) / 128) * 128;   // zero region, see gemm_conv
    char* zrow = smem + ((ROWS * STRIDE + 255) / 256) * 256;
    char* bbuf = zrow + ZBYTES;
    float* scratch = (float*)bbuf;  // epilogue scratch shares the weight staging buffers
    // LEAN: the tap table of gemm_conv_ring follows the squeeze-excite scratch (tower_lds_bytes)
    uint16_t* taptab = (uint16_t*)(bbuf + ((((size_t)2 * P.TB * C + (size_t)P.TB * P.hid) * 4 + 16 + 15) & ~(size_t)15) + (SE_LDS ? (size_t)P.se_vecs * 16 : 0));
    (void)taptab;
    const int tid = threadIdx.x;
    uint4* xres_wg = P.xres + (size_t)blockIdx.x * NV * NT;
#ifdef AZ_POISON
    // Diagnostic build: whatever a previous workgroup left in this LDS allocation is replaced by NaN bit patterns, and the
    // workgroup's slot of the residual scratch is claimed, so that (1) outputs that depended on uninitialised LDS or scratch
    // would change against the regular build (which half of a compute unit's LDS a workgroup gets depends on timing when
    // two are resident), and (2) two workgroups on one slot would be counted.
    for (int i = tid; i < P.lds_bytes / 4; i += NT) ((uint32_t*)smem)[i] = 0xFFFFFFFFu;
    if (tid == 0 && P.owner && atomicExch(&P.owner[blockIdx.x], (int)blockIdx.x + 1) != 0) atomicAdd(&P.owner[gridDim.x], 1);
    __syncthreads();
#endif
#ifdef AZ_STAMP
    unsigned long long stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
    stamp_acc[12] = stamp_last;   // absolute start
#endif
#ifdef AZ_CLOCK   // diagnostic build: shader cycles and 100 MHz real-time ticks over the workgroup's life -> the in-kernel clock
    unsigned long long ck_cycles = 0, ck_ticks = 0;
    if (tid == 0) {
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck_cycles), "=s"(ck_ticks)::"memory");
    }
#endif
    for (int i = tid; i < ZBYTES / 4; i += NT) ((uint32_t*)zrow)[i] = 0;
    // Two workgroups per compute unit: the issue arbiter prefers the older wave of a SIMD, so in a launch that fills the
    // chip exactly the workgroup that arrived first on a compute unit runs ahead (6.3 M against 8.1 M cycles for the same four
    // items). Alternating s_setprio between the two (by convolution parity, by clock parity, younger first) evens that out
    // and shortens such a launch by 1-3 %, but costs 3-4 % of the throughput with several launches in flight: left to age.
    TileGeo g;
    g.row0 = 0; g.n = P.n; g.np = P.np; g.recip_np = P.recip_np; g.recip_n = P.recip_n;
    const int rows = P.rows_dev ? min(*P.rows_dev, P.rows) : P.rows;
    if ((P.gate_mode == 1 && rows > P.gate_rows) || (P.gate_mode == 2 && rows <= P.gate_rows)) return;   // the whole grid alike
    const int items = (rows + P.TB - 1) / P.TB;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int board0 = item * P.TB;
        const int nb = min(P.TB, rows - board0);
        g.vrows = nb * P.np;  // valid rows of this workgroup
        __syncthreads();      // the previous item's last LDS reads are done
        if constexpr (LEAN && WM == 1) build_tap_table(taptab, ROWS, STRIDE, (int)(zrow - act), g, tid, NT);

        // ---- stage 0: feature tile -> act (row stride 2*cin_pad+32)
        const int in_stride0 = 2 * P.cin_pad + 32;
        {
            const int ppr = P.cin_pad * 2 / 16;  // 16-byte pieces per row
            const uint4* src = (const uint4*)(P.feat + (size_t)board0 * P.np * P.cin_pad);
            for (int i = tid; i < ROWS * ppr; i += NT) {
                const int row = i / ppr, part = i - row * ppr;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < g.vrows) v = src[i];
                *(uint4*)(act + row * in_stride0 + part * 16) = v;
            }
        }
        __syncthreads();

        f32x4 acc[MI][NI];
        // ---- stem (model.py:61-65)
        zero_acc<MI, NI>(acc);
        {
            const int kpt = P.cin_pad / 32, nks = (9 * kpt + 1) & ~1;
            if constexpr (LEAN) gemm_taps_gw<MI, NI, WM, WN, C>(acc, P.w_stem, nks, kpt, 9, true, act, in_stride0, zrow, g);
            else gemm_taps<MI, NI, WM, WN, C, false>(acc, P.w_stem, nks, kpt, 9, true, act, in_stride0, zrow, bbuf, g);
        }
        STAMP(0);
        store_act<MI, NI, WM, WN, STRIDE, true, true>(acc, P.b_stem, act, xres_wg);
        STAMP(1);

        // ---- residual tower (model.py:25-40)
        constexpr int KPT = C / 32, NKS = (9 * KPT + 1) & ~1;
        constexpr bool SAME = (KPT % 2) == 0;
        const size_t conv_vecs = (size_t)NKS * 32 * C * 2 / 16;
        for (int blk = 0; blk < P.blocks; ++blk) {
            zero_acc<MI, NI>(acc);
            if constexpr (LEAN && WM == 1) gemm_conv_ring<MI, NI, WN, C, KPT, GW, RING>(acc, P.w_tower + (size_t)(2 * blk) * conv_vecs, 9, act, taptab);
            else if constexpr (SAME && GW > 0) gemm_conv_gw<MI, NI, WM, WN, C, KPT, GW>(acc, P.w_tower + (size_t)(2 * blk) * conv_vecs, 9, true, act, STRIDE, zrow, g);
            else if constexpr (SAME) gemm_conv<MI, NI, WM, WN, C, KPT>(acc, P.w_tower + (size_t)(2 * blk) * conv_vecs, 9, true, act, STRIDE, zrow, bbuf, g);
            else gemm_taps<MI, NI, WM, WN, C, false>(acc, P.w_tower + (size_t)(2 * blk) * conv_vecs, NKS, KPT, 9, true, act, STRIDE, zrow, bbuf, g);
            STAMP(2);
            store_act<MI, NI, WM, WN, STRIDE, true, false>(acc, P.b_tower + (size_t)(2 * blk) * C, act, xres_wg);
            STAMP(3);
            zero_acc<MI, NI>(acc);
            if constexpr (LEAN && WM == 1) gemm_conv_ring<MI, NI, WN, C, KPT, GW, RING>(acc, P.w_tower + (size_t)(2 * blk + 1) * conv_vecs, 9, act, taptab);
            else if constexpr (SAME && GW > 0) gemm_conv_gw<MI, NI, WM, WN, C, KPT, GW>(acc, P.w_tower + (size_t)(2 * blk + 1) * conv_vecs, 9, true, act, STRIDE, zrow, g);
            else if constexpr (SAME) gemm_conv<MI, NI, WM, WN, C, KPT>(acc, P.w_tower + (size_t)(2 * blk + 1) * conv_vecs, 9, true, act, STRIDE, zrow, bbuf, g);
            else gemm_taps<MI, NI, WM, WN, C, false>(acc, P.w_tower + (size_t)(2 * blk + 1) * conv_vecs, NKS, KPT, 9, true, act, STRIDE, zrow, bbuf, g);
            STAMP(2);
            uint4 resv[NV];   // the block input, fetched now, consumed after the SE gate
            {
                const uint4* xr = xres_wg + opaque(tid);
#pragma unroll
                for (int v = 0; v < NV; ++v) resv[v] = xr[(size_t)v * NT];
            }
            {   // y = acc + bias (kept in registers in float32)
                AZ_LANE_GEO(MI, NI, WN);
                const float* bias2 = P.b_tower + (size_t)(2 * blk + 1) * C + col0 + 8 * kq;
#pragma unroll
                for (int p = 0; p < NI / 2; ++p) {
                    const float4 x0 = *(const float4*)(bias2 + 32 * p), x1 = *(const float4*)(bias2 + 32 * p + 4);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        acc[mi][2 * p][0] += x0.x; acc[mi][2 * p][1] += x0.y; acc[mi][2 * p][2] += x0.z; acc[mi][2 * p][3] += x0.w;
                        acc[mi][2 * p + 1][0] += x1.x; acc[mi][2 * p + 1][1] += x1.y; acc[mi][2 * p + 1][2] += x1.z; acc[mi][2 * p + 1][3] += x1.w;
                    }
                }
                if constexpr (NI & 1) {
                    const float4 x0 = *(const float4*)(P.b_tower + (size_t)(2 * blk + 1) * C + col0 + 32 * (NI / 2) + 4 * kq);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) { acc[mi][NI - 1][0] += x0.x; acc[mi][NI - 1][1] += x0.y; acc[mi][NI - 1][2] += x0.z; acc[mi][NI - 1][3] += x0.w; }
                }
            }
            if (P.use_se) {
                store_act<MI, NI, WM, WN, STRIDE, false, false>(acc, nullptr, act, xres_wg);   // bf16 image of y for the pooling
                STAMP(7);
#ifdef AZ_STAMP
                se_gate<MI, NI, WM, WN, C, STRIDE, LEAN && !SE_LDS>(acc, act, scratch, P.se_blob + (size_t)blk * P.se_vecs, P.se_vecs, P.hid, P.TB, nb, g, stamp_acc, stamp_last);
#else
                se_gate<MI, NI, WM, WN, C, STRIDE, LEAN && !SE_LDS>(acc, act, scratch, P.se_blob + (size_t)blk * P.se_vecs, P.se_vecs, P.hid, P.TB, nb, g);
#endif
            }
            STAMP(4);
            residual_store<MI, NI, WM, WN, STRIDE>(acc, resv, act, xres_wg, blk + 1 < P.blocks);
            STAMP(5);
        }

        // ---- both 1x1 head convolutions as one GEMM (model.py:68-71, :78-81); the waves split positions only
        {
            f32x4 hacc[MIH][NIH];
            zero_acc<MIH, NIH>(hacc);
            constexpr int NKH = (KPT + 1) & ~1;
            if constexpr (LEAN && SAME) gemm_conv_gw<MIH, NIH, NW, 1, NH, KPT, 1>(hacc, P.w_head, 1, false, act, STRIDE, zrow, g);
            else if constexpr (SAME) gemm_conv<MIH, NIH, NW, 1, NH, KPT>(hacc, P.w_head, 1, false, act, STRIDE, zrow, bbuf, g);
            else gemm_taps<MIH, NIH, NW, 1, NH, false>(hacc, P.w_head, NKH, KPT, 1, false, act, STRIDE, zrow, bbuf, g);
            // relu(acc + bias) -> LDS as [row][2q] bf16, then one coalesced copy: the workgroup's slice of hq is contiguous
            AZ_LANE_GEO(MIH, NIH, 1);
            const int q2 = 2 * P.q, hstride = q2 * 2 + 16;
            __syncthreads();  // every wave has finished reading act
#pragma unroll
            for (int p = 0; p < NIH / 2; ++p) {
                const int ch8 = 32 * p + 8 * kq;
                if (ch8 >= q2) continue;
                const float4 x0 = *(const float4*)(P.b_head + ch8), x1 = *(const float4*)(P.b_head + ch8 + 4);
                const float bv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int mi = 0; mi < MIH; ++mi) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = fmaxf(hacc[mi][2 * p][r] + bv[r], 0.f); v[4 + r] = fmaxf(hacc[mi][2 * p + 1][r] + bv[4 + r], 0.f); }
                    *(uint4*)(act + (row_base + mi * 16 + lr) * hstride + ch8 * 2) =
                        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
                }
            }
            __syncthreads();
            const int ppr = q2 * 2 / 16;  // 16-byte pieces per row (q2 is a multiple of 8)
            uint4* dst = (uint4*)(P.hq + (size_t)board0 * P.np * q2);
            for (int i = t_; i < g.vrows * ppr; i += NT) {
                const int row = i / ppr, part = i - row * ppr;
                dst[i] = *(const uint4*)(act + row * hstride + part * 16);
            }
        }
        STAMP(6);
    }  // item loop
#ifdef AZ_POISON
    if (tid == 0 && P.owner && atomicExch(&P.owner[blockIdx.x], 0) != (int)blockIdx.x + 1) atomicAdd(&P.owner[gridDim.x], 1);
#endif
#ifdef AZ_CLOCK
    if (tid == 0 && P.stamps) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        P.stamps[(size_t)blockIdx.x * 2] = c1 - ck_cycles;
        P.stamps[(size_t)blockIdx.x * 2 + 1] = r1 - ck_ticks;
    }
#endif
#ifdef AZ_STAMP
    stamp_acc[13] = stamp_last;   // absolute end
    {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamp_acc[14] = hw; stamp_acc[15] = xcc;
    }
    if ((tid & 63) == 0 && P.stamps)   // one record per wave
        for (int i = 0; i < 16; ++i) P.stamps[((size_t)blockIdx.x * NW + (tid >> 6)) * 16 + i] = stamp_acc[i];
#endif
}

// Policy tail (model.py:72-76, :93-94): 1x1 conv q->2 + BN + ReLU, flatten channel-major, linear.
// RB boards per workgroup so that the linear layer's weights are fetched once per RB boards; every
// board's sums run in the same order whichever workgroup and slot it is in.
template <int RB>
__device__ __forceinline__ void policy_tail_body(int bx, const uint16_t* __restrict__ hq, const float* __restrict__ w3, const float* __restrict__ b3,
                                                 const float* __restrict__ wfcT, const float* __restrict__ bfc, float* __restrict__ logits,
                                                 int rows, int np, int q, const int* rows_dev, int row_off) {
    extern __shared__ float flat[];  // [RB][2*np]
    if (rows_dev) rows = min(rows, *rows_dev - row_off);   // this launch covers rows row_off .. of the batch
    const int r0 = bx * RB, A = np + 1, q2 = 2 * q, F = 2 * np;
    const int nbk = min(RB, rows - r0);
    if (nbk <= 0) return;
    for (int i = threadIdx.x; i < RB * F; i += blockDim.x) {
        const int b = i / F, k = i - b * F, c2 = k / np, pos = k - c2 * np;
        float s = 0.f;
        if (b < nbk) {
            const uint16_t* src = hq + ((size_t)(r0 + b) * np + pos) * q2;
            s = b3[c2];
            for (int kk = 0; kk < q; kk += 8) {
                const uint4 v = *(const uint4*)(src + kk);
                const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = fmaf(w3[c2 * q + kk + 2 * e], bf16_lo(u[e]), s);
                    s = fmaf(w3[c2 * q + kk + 2 * e + 1], bf16_hi(u[e]), s);
                }
            }
            s = fmaxf(s, 0.f);
        }
        flat[i] = s;
    }
    __syncthreads();
    constexpr int RH = RB / 2;   // boards per thread
    for (int o = threadIdx.x; o < 2 * A; o += blockDim.x) {
        const int hsel = o / A, a = o - hsel * A;
        float acc[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) acc[r] = bfc[a];
        const float* fl = flat + hsel * RH * F;
        int i = 0;
        for (; i + 8 <= F; i += 8) {   // eight weight loads in flight; the sums keep their order
            float w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = wfcT[(size_t)(i + k) * A + a];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int r = 0; r < RH; ++r) acc[r] = fmaf(w[k], fl[r * F + i + k], acc[r]);
        }
        for (; i < F; ++i) {
            const float w = wfcT[(size_t)i * A + a];
#pragma unroll
            for (int r = 0; r < RH; ++r) acc[r] = fmaf(w, fl[r * F + i], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < RH; ++r)
            if (hsel * RH + r < nbk) logits[(size_t)(r0 + hsel * RH + r) * A + a] = acc[r];
    }
}

template <int RB>
__global__ __launch_bounds__(256) void k_policy_tail(const uint16_t* __restrict__ hq, const float* __restrict__ w3, const float* __restrict__ b3,
                                                     const float* __restrict__ wfcT, const float* __restrict__ bfc, float* __restrict__ logits,
                                                     int rows, int np, int q, const int* rows_dev, int row_off) {
    policy_tail_body<RB>(blockIdx.x, hq, w3, b3, wfcT, bfc, logits, rows, np, q, rows_dev, row_off);
}

// Value tail (model.py:82-86): flatten channel-major, linear + ReLU, linear + tanh. RB boards per block.
template <int RB>
__global__ __launch_bounds__(256) void k_value_tail(const uint16_t* __restrict__ hq, const uint16_t* __restrict__ w4T, const float* __restrict__ b4,
                                                    const float* __restrict__ w6, const float* __restrict__ b6, float* __restrict__ values,
                                                    int rows, int np, int q, int vh, const int* rows_dev, int row_off) {
    extern __shared__ float v0[];  // [RB][q*np] then [RB][vh]
    if (rows_dev) rows = min(rows, *rows_dev - row_off);
    const int r0 = blockIdx.x * RB, nbk = min(RB, rows - r0), K = q * np, q2 = 2 * q;
    if (nbk <= 0) return;
    float* hid = v0 + RB * K;
    for (int i = threadIdx.x; i < RB * K; i += blockDim.x) {
        const int b = i / K, k = i - b * K, ch = k / np, pos = k - ch * np;
        v0[i] = b < nbk ? __uint_as_float((uint32_t)hq[((size_t)(r0 + b) * np + pos) * q2 + q + ch] << 16) : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < vh; j += blockDim.x) {
        float acc[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) acc[b] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float w = __uint_as_float((uint32_t)w4T[(size_t)k * vh + j] << 16);
#pragma unroll
            for (int b = 0; b < RB; ++b) acc[b] = fmaf(w, v0[b * K + k], acc[b]);
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) hid[b * vh + j] = fmaxf(acc[b] + b4[j], 0.f);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < nbk; b += 4) {
        float s = 0.f;
        for (int j = lane; j < vh; j += 64) s = fmaf(w6[j], hid[b * vh + j], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) values[r0 + b] = tanhf(s + b6[0]);
    }
}

// Value tail on the matrix cores: hidden = relu(W4 * flatten(v) + b4) as a [16 boards] x [K = np*q] x [vh] GEMM per
// workgroup (4 waves, each NTW tiles of 16 hidden units), operands straight from global memory into
// registers, four k-steps in flight. K is walked position-major (k' = pos*q + ch, the order the head
// activations are stored in); the weights are packed to match (pack_value_fc). value = tanh(w6 . hidden + b6)
// with a fixed reduction tree, so a board's value does not depend on its row or on the other rows.
template <int NTW>
__device__ __forceinline__ void value_tail_body(int bx, int by, const uint16_t* __restrict__ hq, const uint4* __restrict__ w4p, const float* __restrict__ b4p,
                                                const float* __restrict__ w6p, const float* __restrict__ b6, float* __restrict__ values,
                                                int rows, int np, int q, int nks, int ntiles, unsigned recip_qc, const int* rows_dev, int row_off, float* part_out, int part_stride) {
    __shared__ float red[4][16];
    if (rows_dev) rows = min(rows, *rows_dev - row_off);
    const int r0 = bx * 16;
    if (r0 >= rows) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    const int row = min(r0 + li, rows - 1);
    const int qc = q >> 3, nfrag = np * qc;
    const uint16_t* abase = hq + (size_t)row * np * 2 * q + q;   // the value half of this board's head activations
    const int t0 = by * NTW;                                     // split launch: this workgroup's share of the unit tiles
    const uint4* bbase = w4p + (size_t)wave * 64 + lane + (size_t)t0 * 256;   // tile `wave + 4 t0` of k-step 0; tiles wave + 4(t0 + t) follow at +256t vectors
    const size_t kstride = (size_t)ntiles * 64;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto load_a = [&](int s) -> uint4 {
        const int f = 4 * s + kq;
        if (f >= nfrag) return zero4;
        const int pos = (int)(((unsigned)f * recip_qc) >> 16), chunk = f - pos * qc;
        return *(const uint4*)(abase + (size_t)pos * 2 * q + chunk * 8);
    };
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 a[4], b[4][NTW];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        a[d] = zero4;
#pragma unroll
        for (int t = 0; t < NTW; ++t) b[d][t] = zero4;
        if (d < nks) {
            a[d] = load_a(d);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                if (wave + 4 * (t0 + t) < ntiles) b[d][t] = bbase[(size_t)d * kstride + t * 256];
        }
    }
    for (int s = 0; s < nks; s += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (s + d < nks) {
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[d]), __builtin_bit_cast(bf16x8, b[d][t]), acc[t], 0, 0, 0);
                if (s + d + 4 < nks) {
                    a[d] = load_a(s + d + 4);
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        if (wave + 4 * (t0 + t) < ntiles) b[d][t] = bbase[(size_t)(s + d + 4) * kstride + t * 256];
                }
            }
        }
    }
    // lane: hidden unit (wave + 4t)*16 + li of boards 4*kq .. 4*kq+3
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int j = (wave + 4 * (t0 + t)) * 16 + li;
        const bool live = wave + 4 * (t0 + t) < ntiles;
        const float bb = live ? b4p[j] : 0.f, ww = live ? w6p[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[r] = fmaf(ww, fmaxf(acc[t][r] + bb, 0.f), part[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[r] += __shfl_xor(part[r], 1);
        part[r] += __shfl_xor(part[r], 2);
        part[r] += __shfl_xor(part[r], 4);
        part[r] += __shfl_xor(part[r], 8);
    }
    if (li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * kq + r] = part[r];
    }
    __syncthreads();
    if (threadIdx.x < 16 && r0 + (int)threadIdx.x < rows) {
        const int i = threadIdx.x;
        const float sum = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        if (part_out) part_out[(size_t)(r0 + i) * part_stride + by] = sum;   // split launch: k_value_finish adds the shares
        else values[r0 + i] = tanhf(sum + b6[0]);
    }
}

template <int NTW>
__global__ __launch_bounds__(256) void k_value_tail_mfma(const uint16_t* __restrict__ hq, const uint4* __restrict__ w4p, const float* __restrict__ b4p,
                                                         const float* __restrict__ w6p, const float* __restrict__ b6, float* __restrict__ values,
                                                         int rows, int np, int q, int nks, int ntiles, unsigned recip_qc, const int* rows_dev, int row_off, float* part_out, int part_stride) {
    // units of 16 boards with a stride of the grid, which run_tails caps for a launch over an upper bound
    const int live = rows_dev ? min(rows, *rows_dev - row_off) : rows;
    for (int bx = blockIdx.x; bx * 16 < live; bx += gridDim.x) {
        if (bx != (int)blockIdx.x) __syncthreads();   // `red` of the previous unit has been read
        value_tail_body<NTW>(bx, blockIdx.y, hq, w4p, b4p, w6p, b6, values, rows, np, q, nks, ntiles, recip_qc, rows_dev, row_off, part_out, part_stride);
    }
}

// Both tails in one launch (boards up to 9x9, un-split value tail). The two read the same head activations and are independent, so they share the machine instead of running
// one after the other; each computes exactly what its own kernel computes.
struct TailArgs {
    const uint16_t* hq; const float *w3, *b3, *wfcT, *bfc; float* logits;
    const uint4* w4p; const float *b4p, *w6p, *b6; float* values;
    int rows, np, q, nks, ntiles; unsigned recip_qc; const int* rows_dev; int row_off;
};
template <int NTW>
__global__ __launch_bounds__(256) void k_tails(TailArgs a) {
    // workgroups come in threes per 16 boards - two policy workgroups (8 boards each) and one value workgroup - so that
    // the live workgroups of both kinds are dispatched first when the grid covers an upper bound of the row count
    const int unit = blockIdx.x / 3, role = blockIdx.x - 3 * unit;
    if (role < 2)
        policy_tail_body<8>(2 * unit + role, a.hq, a.w3, a.b3, a.wfcT, a.bfc, a.logits, a.rows, a.np, a.q, a.rows_dev, a.row_off);
    else
        value_tail_body<NTW>(unit, 0, a.hq, a.w4p, a.b4p, a.w6p, a.b6, a.values, a.rows, a.np, a.q, a.nks, a.ntiles, a.recip_qc,
                             a.rows_dev, a.row_off, nullptr, 0);
}

// Second half of a split value tail: value = tanh(sum of the workgroups' shares in index order + b6).
__global__ __launch_bounds__(256) void k_value_finish(const float* __restrict__ part, int shares, const float* __restrict__ b6, float* __restrict__ values,
                                                      int rows, const int* rows_dev, int row_off) {
    if (rows_dev) rows = min(rows, *rows_dev - row_off);
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int k = 0; k < shares; ++k) s += part[(size_t)r * shares + k];
    values[r] = tanhf(s + b6[0]);
}

// ------------------------------------------------------------------------------------------------ host
struct MfmaNet {
    az_net_config cfg;
    int TB, cin_pad, NH, q;
    std::vector<void*> allocs;
    TowerParams P;
    // 64 channels: the 96-row variant is a 2x4-wave kernel reading weight fragments from global memory, while the main kernel
    // (4x2 waves, weights through LDS) pairs its channel tiles differently: convolution images of its own
    const uint4 *w_stem_small = nullptr, *w_tower_small = nullptr;
    float *w3, *b3, *wfcT, *bfc, *b4, *w6, *b6;
    uint16_t* w4T;
    uint16_t* w4p = nullptr;   // value fc1 packed for k_value_tail_mfma (null: hidden size too large, scalar tail)
    float *b4p = nullptr, *w6p = nullptr;
    int v_nks = 0, v_ntiles = 0;
    unsigned v_recip = 0;
    uint16_t* hq = nullptr;
    int hq_rows = 0;
    float* vpart = nullptr;       // split value tail: [chunk rows][shares] partial sums, allocated with hq (layer path, large boards)
    uint4* xres = nullptr;   // residual-stream scratch for max_grid workgroups
    int max_grid = 0;
    int cus = 256;                // compute units of the device
    std::vector<void*> retired;   // scratch buffers that were replaced by larger ones: launches already enqueued may still use
                                  // them, so they are only freed with the network (sizes grow geometrically: bounded waste)
    int TB_duo = 0;               // boards per workgroup of the two-workgroups-per-CU variant (176 rows, 256 threads; 0: none)
    bool duo_above_one_round = false;   // the two-per-CU kernel only where the 256-row kernel would need a second round of workgroups (128 channels)
    int small_rounds = 2;         // rounds of workgroups up to which the 96-row variant is chosen (AZ_TOWER_SMALL_ROUNDS overrides)
    bool small_lean = false;      // the 96-row variant as 256-thread workgroups, two per compute unit (192 channels)
    int TB_small = 0;             // boards per workgroup of the 96-row low-latency variant (0: this shape has none)
    bool device_choice = true;    // indirect forwards under a moderate bound launch the 96-row variant too, gated on the count
    int rows_main = 256;          // rows of the main kernel's workgroup: 256, or 384 for one board of 17x17 to 19x19 (128 channels)
    int split = 0;                // 0: 4x2 waves, weights through LDS; 1: 2x4 waves, weight fragments from global memory (C = 192)
    LayerNet* layers = nullptr;   // set instead of the fused tower's weights when the shape needs the layer-by-layer path
    // optional timing of the tower kernel with HIP events on the launch stream
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t ev_used = 0;
    long long prof_rows = 0;
};

static int static_scratch(int device, unsigned char** out);   // defined beside az_net_forward_static below

template <typename T>
static int up(MfmaNet* m, const std::vector<T>& v, T** out) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) return fail(AZ_ERR_HIP, "hipMalloc (network weights)");
    m->allocs.push_back(p);
    if (hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return fail(AZ_ERR_HIP, "hipMemcpy (network weights)");
    *out = (T*)p;
    return AZ_OK;
}

// Whether whole boards fill the 176 rows of a two-per-CU workgroup well enough: 90 % where the 256-row kernel is the
// alternative (9x9: 162 rows, 5x5: 175); 65 % for boards of more than 88 points, whose alternative is the layer-by-layer
// path (measured at 192 channels x 12 blocks, 1024 boards, fused against layer path: 13x13 1 255 / 793 TFLOP/s at 96 %,
// 12x12 1 095 / 899 at 82 %, 11x11 957 / 890 at 69 %, 10x10 817 / 807 at 57 %).
static bool duo_fill_ok(int np) { return np <= 176 && (176 / np) * np * 100 >= 176 * (np > 88 ? 65 : 90); }

static bool fused_supported(const az_net_config& c) {
    const bool cok = c.channels == 32 || c.channels == 64 || c.channels == 128 || c.channels == 192;
    // above 9x9: 192 channels where one board fills the 176 rows of a two-per-CU workgroup well enough (11x11 to 13x13)
    // 17x17 to 19x19: 128 channels, one board in a 384-row workgroup (its activation image, 108 KB, still fits the LDS)
    const int np = c.board_size * c.board_size;
    const bool bok = c.board_size <= 9 || ((c.channels == 192 || c.channels == 128) && duo_fill_ok(np) && !getenv("AZ_NO_FUSED_13")) ||
                     (c.channels == 128 && np > 256 && np <= 384 && np * 4 >= 384 * 3 && !getenv("AZ_NO_FUSED_19"));
    // 14x14 to 16x16: 128 or 192 channels, one board in a 256-row workgroup of the 2x4-wave kernel (at least three quarters of its rows)
    const bool bok16 = (c.channels == 128 || c.channels == 192) && np > 176 && np <= 256 && np * 4 >= 256 * 3 && !getenv("AZ_NO_FUSED_16");
    return cok && (bok || bok16) && c.board_size >= 2 && c.in_planes <= 64;
}
bool mfma_supported(const az_net_config& c) { return (fused_supported(c) || layers_supported(c)) && c.value_hidden >= 1; }
bool mfma_is_fused(const MfmaNet* m) { return m->layers == nullptr; }

// Residual scratch for `workgroups` resident workgroups of the 256-row kernel; the two-per-CU variant runs twice as many
// workgroups of 17 x 256 vectors each out of the same buffer.
static size_t xres_bytes(int workgroups, int C, bool duo, int rows_main = 256) {
    const size_t per = (size_t)rows_main * C * 2, per_duo = (size_t)2 * 17 * 256 * 16;
    return (size_t)workgroups * std::max(per, duo ? per_duo : (size_t)0);
}

static size_t se_blob_bytes(int C, int hid) {  // per block: w1, w2^T (bf16), b1, b2 (f32), padded to 1 KiB
    const size_t raw = (size_t)2 * hid * C * 2 + (size_t)(hid + C) * 4;
    return (raw + 1023) / 1024 * 1024;
}

// LDS: activations, zero row, then a region shared by the two weight staging buffers and the
// squeeze-excite scratch (pooled [TB][C], hidden [TB][hid], scale [TB][C],
// the block's SE weights).
static size_t tower_lds_bytes(int C, int NH, int TB, int hid, int rows = 256, bool lean = false) {
    const size_t stride = 2 * C + 32;
    const size_t tile = lean ? 0 : (size_t)64 * std::max(C, NH) * 2;
    const size_t se = lean ? ((((size_t)2 * TB * C + (size_t)TB * hid) * 4 + 16 + 15) & ~(size_t)15) + (size_t)9 * rows * 2   // + tap table
                                 + (rows == 96 ? se_blob_bytes(C, hid) : 0)   // the 96-row form stages the squeeze-excite weights
                           : ((size_t)2 * TB * C + (size_t)TB * hid) * 4 + 16 + se_blob_bytes(C, hid);
    const size_t zbytes = ((256 + 2 * (size_t)C + 16 + 127) / 128) * 128;
    return (((size_t)rows * stride + 255) / 256) * 256 + zbytes + std::max(2 * tile, se);
}

int mfma_create(const az_net_config& c, const NetLayout& L, const float* Pm, MfmaNet** out) {
    if (!mfma_supported(c)) return fail(AZ_ERR_ARG, "bf16 MFMA path: unsupported network shape");
    if (hipSetDevice(c.device) != hipSuccess) return fail(AZ_ERR_HIP, "hipSetDevice");
    MfmaNet* m = new MfmaNet();
    m->cfg = c;
    const int C = c.channels, np = c.board_size * c.board_size, q = C / 4, A = np + 1;
    m->rows_main = np > 256 ? 384 : 256;
    m->TB = m->rows_main / np;
    m->cin_pad = c.in_planes <= 32 ? 32 : 64;
    m->q = q;
    m->NH = std::max(32, 2 * q);   // policy q + value q channels, in whole tile pairs
    const bool fused = fused_supported(c);
    m->split = C >= 128 ? 1 : 0;   // measured: 192 ch +3 %, 128 ch +5 %, 64 ch -3 %
    if (const char* sp = getenv("AZ_TOWER_SPLIT")) m->split = (C >= 64 && atoi(sp) == 1) ? 1 : 0;   // 0 selects the LDS-staged variant
    if (m->rows_main == 384) m->split = 1;   // the 384-row kernel exists in the global-weights form only
    // the 96-row variant shares the 2x4-wave kernels' weight images: 128 / 192 channels in their default (global-weights)
    // form, boards of at most 96 points; AZ_TOWER_SMALL=0 switches it off
    // 64 channels have it too (their main kernel is the LDS-staged one, so the variant gets weight images of its own)
    m->TB_small = (fused && ((m->split == 1 && (C == 128 || C == 192)) || C == 64) && np <= 96) ? 96 / np : 0;
    if (const char* sm = getenv("AZ_TOWER_SMALL")) if (atoi(sm) == 0) m->TB_small = 0;
    // two 256-thread workgroups of two boards per compute unit (192 channels, global-weights form)
    // where whole boards fill at least 90 % of its 176 rows (duo_fill_ok); AZ_TOWER_DUO=0 switches it off
    // 128 channels on boards up to 9x9 have the two-per-CU kernel too, from the row count on where the 256-row kernel needs
    // a second round of workgroups (same box, 9x9 x 10 blocks, TFLOP/s against the 256-row kernel: 4096 boards 1 125 / 922,
    // 1024 boards 979 / 698, 768 boards 840 / 836, 600 boards 700 / 750). AZ_TOWER_DUO128=0: off, =2: at every row count.
    const int duo128_mode = getenv("AZ_TOWER_DUO128") ? atoi(getenv("AZ_TOWER_DUO128")) : 1;
    const bool duo128 = C == 128 && (duo128_mode != 0 || c.board_size > 9);   // 11x11 to 13x13: one board per workgroup, as with 192 channels
    m->duo_above_one_round = duo128 && duo128_mode != 2 && c.board_size <= 9;
    m->TB_duo = (fused && m->split == 1 && (C == 192 || duo128) && duo_fill_ok(np)) ? 176 / np : 0;
    if (const char* du = getenv("AZ_TOWER_DUO")) if (atoi(du) == 0) m->TB_duo = 0;
    m->small_lean = m->TB_small > 0 && (m->TB_duo > 0 || (C == 128 && m->split == 1)) && np <= 96;
    if (const char* sl = getenv("AZ_TOWER_SMALL_LEAN")) if (atoi(sl) == 0) m->small_lean = false;
    m->small_rounds = (C == 64 || (C == 192 && m->TB_duo > 0)) ? 1 : 2;
    if (const char* sr = getenv("AZ_TOWER_SMALL_ROUNDS")) m->small_rounds = std::max(1, atoi(sr));
    const int tiles_per_wave = C / 64;              // 2x4 waves: C/4 channels each
    const int slab_tiles = (m->split == 1 && (tiles_per_wave & 1)) ? tiles_per_wave : 0;   // channel tiles per wave when that count is odd
    const bool small_images = m->TB_small > 0 && m->split == 0;   // (only C == 64) pack the 2x4-wave images beside the main ones
    if (fused && tower_lds_bytes(C, m->NH, m->TB, c.se_hidden, m->rows_main) > 160 * 1024) { delete m; return fail(AZ_ERR_ARG, "bf16 MFMA path: LDS budget exceeded"); }
    std::vector<float> w, b;
    TowerParams& P = m->P;
    memset(&P, 0, sizeof(P));
    if (!fused) {
        int rcl = layers_create(c, L, Pm, &m->layers);
        if (rcl != AZ_OK) { delete m; return rcl; }
    }
    P.n = c.board_size; P.np = np; P.TB = m->TB; P.blocks = c.blocks; P.use_se = c.use_se; P.hid = c.se_hidden;
    P.cin_pad = m->cin_pad; P.q = q;
    P.recip_np = (65536 + np - 1) / np;
    P.recip_n = (65536 + c.board_size - 1) / c.board_size;
    for (int row = 0; fused && row < m->rows_main; ++row)   // every row index a workgroup of any variant can hold
        if ((((unsigned)row * P.recip_np) >> 16 != (unsigned)(row / np)) || (((unsigned)row * P.recip_n) >> 16 != (unsigned)(row / c.board_size))) {
            delete m;
            return fail(AZ_ERR_ARG, "reciprocal division check failed");
        }
    int rc;
#define TRY(x) if ((rc = (x)) != AZ_OK) { mfma_destroy(m); return rc; }
    if (fused) {   // stem
        fold_bn(Pm, L.stem_w, L.stem_bn, C, (int64_t)c.in_planes * 9, w, b);
        std::vector<uint16_t> pk = pack_conv(w, C, c.in_planes, 9, m->cin_pad, C, false, slab_tiles);
        uint16_t* d; float* bd;
        TRY(up(m, pk, &d)); TRY(up(m, b, &bd));
        P.w_stem = (const uint4*)d; P.b_stem = bd;
        if (small_images) {
            pk = pack_conv(w, C, c.in_planes, 9, m->cin_pad, C, false, tiles_per_wave);
            TRY(up(m, pk, &d)); m->w_stem_small = (const uint4*)d;
        }
    }
    if (fused) {   // tower
        std::vector<uint16_t> all, all_small;
        std::vector<float> ball;
        const size_t blob_bytes = se_blob_bytes(C, c.se_hidden);
        std::vector<uint8_t> blob(c.use_se ? blob_bytes * c.blocks : 0, 0);
        for (int i = 0; i < c.blocks; ++i) {
            const BlockOff& bo = L.blocks[i];
            fold_bn(Pm, bo.conv1, bo.bn1, C, (int64_t)C * 9, w, b);
            std::vector<uint16_t> pk = pack_conv(w, C, C, 9, C, C, false, slab_tiles);
            all.insert(all.end(), pk.begin(), pk.end()); ball.insert(ball.end(), b.begin(), b.end());
            if (small_images) { pk = pack_conv(w, C, C, 9, C, C, false, tiles_per_wave); all_small.insert(all_small.end(), pk.begin(), pk.end()); }
            fold_bn(Pm, bo.conv2, bo.bn2, C, (int64_t)C * 9, w, b);
            pk = pack_conv(w, C, C, 9, C, C, false, slab_tiles);
            all.insert(all.end(), pk.begin(), pk.end()); ball.insert(ball.end(), b.begin(), b.end());
            if (small_images) { pk = pack_conv(w, C, C, 9, C, C, false, tiles_per_wave); all_small.insert(all_small.end(), pk.begin(), pk.end()); }
            if (c.use_se) {
                const int hid = c.se_hidden;
                uint8_t* dst = blob.data() + blob_bytes * i;
                uint16_t* w1 = (uint16_t*)dst;
                uint16_t* w2t = w1 + (size_t)hid * C;
                float* b1 = (float*)(w2t + (size_t)hid * C);
                float* b2 = b1 + hid;
                for (int j = 0; j < hid; ++j)
                    for (int ch = 0; ch < C; ++ch) {
                        w1[(size_t)j * C + ch] = f2bf(Pm[bo.se_w1 + (int64_t)j * C + ch]);
                        w2t[(size_t)j * C + ch] = f2bf(Pm[bo.se_w2 + (int64_t)ch * hid + j]);
                    }
                memcpy(b1, Pm + bo.se_b1, sizeof(float) * hid);
                memcpy(b2, Pm + bo.se_b2, sizeof(float) * C);
            }
        }
        uint16_t* d; float* f; uint8_t* bl;
        TRY(up(m, all, &d)); P.w_tower = (const uint4*)d;
        if (small_images) { TRY(up(m, all_small, &d)); m->w_tower_small = (const uint4*)d; }
        TRY(up(m, ball, &f)); P.b_tower = f;
        TRY(up(m, blob, &bl)); P.se_blob = (const uint4*)bl; P.se_vecs = (int)(blob_bytes / 16);
    }
    if (fused) {   // fused head 1x1: columns [0,q) policy_head.0, [q,2q) value_head.0
        std::vector<float> wp, bp, wv, bv, wh((size_t)2 * q * C), bh(m->NH, 0.f);
        fold_bn(Pm, L.ph0_w, L.ph0_bn, q, C, wp, bp);
        fold_bn(Pm, L.vh0_w, L.vh0_bn, q, C, wv, bv);
        std::copy(wp.begin(), wp.end(), wh.begin());
        std::copy(wv.begin(), wv.end(), wh.begin() + (size_t)q * C);
        std::copy(bp.begin(), bp.end(), bh.begin());
        std::copy(bv.begin(), bv.end(), bh.begin() + q);
        std::vector<uint16_t> pk = pack_conv(wh, 2 * q, C, 1, C, m->NH);
        uint16_t* d; float* f;
        TRY(up(m, pk, &d)); P.w_head = (const uint4*)d;
        TRY(up(m, bh, &f)); P.b_head = f;
    }
    {   // tails
        fold_bn(Pm, L.ph3_w, L.ph3_bn, 2, q, w, b);
        TRY(up(m, w, &m->w3)); TRY(up(m, b, &m->b3));
        std::vector<float> t((size_t)2 * np * A);
        for (int a = 0; a < A; ++a)
            for (int i = 0; i < 2 * np; ++i) t[(size_t)i * A + a] = Pm[L.pfc_w + (int64_t)a * 2 * np + i];
        TRY(up(m, t, &m->wfcT));
        TRY(up(m, std::vector<float>(Pm + L.pfc_b, Pm + L.pfc_b + A), &m->bfc));
        const int vh = c.value_hidden, K = q * np;
        std::vector<uint16_t> t4((size_t)K * vh);
        for (int j = 0; j < vh; ++j)
            for (int k = 0; k < K; ++k) t4[(size_t)k * vh + j] = f2bf(Pm[L.vh4_w + (int64_t)j * K + k]);
        TRY(up(m, t4, &m->w4T));
        TRY(up(m, std::vector<float>(Pm + L.vh4_b, Pm + L.vh4_b + vh), &m->b4));
        TRY(up(m, std::vector<float>(Pm + L.vh6_w, Pm + L.vh6_w + vh), &m->w6));
        TRY(up(m, std::vector<float>(Pm + L.vh6_b, Pm + L.vh6_b + 1), &m->b6));
        if (vh <= 256) {
            // [k-step][unit tile][lane][8]: lane l holds unit tile*16 + (l & 15), k' = 32*s + 8*(l >> 4) + e with
            // k' = pos*q + ch standing for column ch*np + pos of value_head.4 (flatten is channel-major, model.py:82)
            const int ntiles = (vh + 15) / 16, nks = (K + 31) / 32, qc = q / 8;
            std::vector<uint16_t> pk((size_t)nks * ntiles * 64 * 8, 0);
            for (int s2 = 0; s2 < nks; ++s2)
                for (int t = 0; t < ntiles; ++t)
                    for (int l = 0; l < 64; ++l) {
                        const int j = t * 16 + (l & 15);
                        if (j >= vh) continue;
                        for (int e = 0; e < 8; ++e) {
                            const int kp = 32 * s2 + 8 * (l >> 4) + e;
                            if (kp >= K) continue;
                            const int pos = kp / q, ch = kp % q;
                            pk[(((size_t)s2 * ntiles + t) * 64 + l) * 8 + e] = f2bf(Pm[L.vh4_w + (int64_t)j * K + (int64_t)ch * np + pos]);
                        }
                    }
            std::vector<float> b4p((size_t)ntiles * 16, 0.f), w6p((size_t)ntiles * 16, 0.f);
            for (int j = 0; j < vh; ++j) { b4p[j] = Pm[L.vh4_b + j]; w6p[j] = Pm[L.vh6_w + j]; }
            TRY(up(m, pk, &m->w4p)); TRY(up(m, b4p, &m->b4p)); TRY(up(m, w6p, &m->w6p));
            m->v_nks = nks; m->v_ntiles = ntiles;
            m->v_recip = (65536 + qc - 1) / qc;
            for (int f = 0; f < np * qc + 8; ++f)
                if ((((unsigned)f * m->v_recip) >> 16) != (unsigned)(f / qc)) { mfma_destroy(m); return fail(AZ_ERR_ARG, "reciprocal division check failed (value tail)"); }
        }
    }
#undef TRY
    // opt in to the large dynamic LDS size once
    const size_t lds = fused ? tower_lds_bytes(C, m->NH, m->TB, c.se_hidden, m->rows_main) : 0;
    hipError_t e = hipSuccess;
    switch (fused ? C : 0) {
        case 32: e = hipFuncSetAttribute((const void*)k_tower<32, 8, 1, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); break;
        case 64:
            e = hipFuncSetAttribute((const void*)k_tower<64, 4, 2, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_tower<64, 2, 4, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess && m->TB_small)
                e = hipFuncSetAttribute((const void*)k_tower<64, 2, 4, 32, AZ_GW64S, 96>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tower_lds_bytes(C, m->NH, m->TB_small, c.se_hidden, 96));
            break;
        case 128:
            e = hipFuncSetAttribute((const void*)k_tower<128, 4, 2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_tower<128, 2, 4, 64, AZ_GW128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess && m->TB_small)
                e = hipFuncSetAttribute((const void*)k_tower<128, 2, 4, 64, AZ_GW128S, 96>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tower_lds_bytes(C, m->NH, m->TB_small, c.se_hidden, 96));
            if (e == hipSuccess && m->small_lean)
                e = hipFuncSetAttribute((const void*)k_tower<128, 1, 4, 64, AZ_GW128S, 96, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tower_lds_bytes(C, m->NH, m->TB_small, c.se_hidden, 96, true));
            if (e == hipSuccess && m->TB_duo)
                e = hipFuncSetAttribute((const void*)k_tower<128, 1, 4, 64, AZ_GW128D, 176, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tower_lds_bytes(C, m->NH, m->TB_duo, c.se_hidden, 176, true));
            if (e == hipSuccess && m->rows_main == 384)
                e = hipFuncSetAttribute((const void*)k_tower<128, 2, 4, 64, AZ_GW128, 384>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            break;
        case 192:
            e = hipFuncSetAttribute((const void*)k_tower<192, 4, 2, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_tower<192, 2, 4, 96, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess && m->TB_small)
                e = hipFuncSetAttribute((const void*)k_tower<192, 2, 4, 96, 2, 96>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tower_lds_bytes(C, m->NH, m->TB_small, c.se_hidden, 96));
            if (e == hipSuccess && m->TB_duo)
                e = hipFuncSetAttribute((const void*)k_tower<192, 1, 4, 96, 2, 176, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tower_lds_bytes(C, m->NH, m->TB_duo, c.se_hidden, 176, true));
            if (e == hipSuccess && m->small_lean)
                e = hipFuncSetAttribute((const void*)k_tower<192, 1, 4, 96, 5, 96, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)tower_lds_bytes(C, m->NH, m->TB_small, c.se_hidden, 96, true));
            break;
    }
    if (e != hipSuccess) { mfma_destroy(m); return fail(AZ_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); }
    if (!m->w4p) {   // scalar value tail (hidden size above 256): its input image must fit LDS
        const size_t vt = ((size_t)4 * q * np + 4 * c.value_hidden) * 4;
        e = vt <= 160 * 1024 ? hipFuncSetAttribute((const void*)k_value_tail<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vt) : hipErrorInvalidValue;
        if (e != hipSuccess) { mfma_destroy(m); return fail(AZ_ERR_ARG, "bf16 MFMA path: value head too large (value_hidden > 256 on this board size)"); }
    }
    if (fused) {   // one workgroup per CU (LDS-bound), each striding over the board groups
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c.device) != hipSuccess) { mfma_destroy(m); return fail(AZ_ERR_HIP, "hipGetDeviceProperties"); }
        m->max_grid = std::max(1, prop.multiProcessorCount);
        m->cus = m->max_grid;
        if (hipMalloc((void**)&m->xres, xres_bytes(m->max_grid, C, m->TB_duo > 0, m->rows_main)) != hipSuccess) { mfma_destroy(m); return fail(AZ_ERR_HIP, "hipMalloc (residual scratch)"); }
        // The address of the library's static scratch array on this device is looked up here, at set-up, so that
        // az_net_forward_static itself (possibly under graph capture) makes no symbol query. A failure is not an error
        // of creation: the entry point repeats the lookup and reports it.
        if (c.board_size <= 9) { unsigned char* unused = nullptr; (void)static_scratch(c.device, &unused); }
    } else {   // layer path: head activations (and value-tail shares) of one chunk, sized once
        int rcr = mfma_reserve(m, layers_chunk_boards(m->layers));
        if (rcr != AZ_OK) { mfma_destroy(m); return rcr; }
    }
    *out = m;
    return AZ_OK;
}

// Scratch for forwards of up to `rows` positions, so that later forwards of that size allocate nothing. Never waits for
// the device and never frees a buffer a launch in flight may use (see MfmaNet::retired).
int mfma_reserve(MfmaNet* m, int rows) {
    if (rows < 1) return fail(AZ_ERR_ARG, "az_net_reserve: rows must be >= 1");
    if (hipSetDevice(m->cfg.device) != hipSuccess) return fail(AZ_ERR_HIP, "hipSetDevice");
    const int np = m->cfg.board_size * m->cfg.board_size;
    if (m->layers) {   // a pass holds at most what the activation arrays can (larger batches run pass by pass)
        int rcl = layers_reserve(m->layers, rows);
        if (rcl != AZ_OK) return rcl;
        rows = std::min(rows, layers_cap_boards(m->layers));
    }
    if (rows <= m->hq_rows) return AZ_OK;
    uint16_t* h = nullptr;
    if (hipMalloc((void**)&h, (size_t)rows * np * 2 * m->q * 2) != hipSuccess) return fail(AZ_ERR_HIP, "hipMalloc (head scratch)");
    float* vp = nullptr;
    if (np > 81 && m->w4p && m->v_ntiles > 4) {   // split value tail (run_tails)
        if (hipMalloc((void**)&vp, (size_t)(rows + 16) * ((m->v_ntiles + 3) / 4) * sizeof(float)) != hipSuccess) {
            (void)hipFree(h);
            return fail(AZ_ERR_HIP, "hipMalloc (value tail scratch)");
        }
    }
    if (m->hq) m->retired.push_back(m->hq);
    if (vp && m->vpart) m->retired.push_back(m->vpart);
    m->hq = h;
    if (vp) m->vpart = vp;
    m->hq_rows = rows;
    return AZ_OK;
}

// One reference event per device: launch spans of different networks and streams share its time line.
static hipEvent_t g_base_event[64];
static bool g_base_set[64];

static int base_event(int device, hipEvent_t* out) {
    if (device < 0 || device >= 64) return fail(AZ_ERR_ARG, "device index out of range");
    if (!g_base_set[device]) {
        if (hipEventCreate(&g_base_event[device]) != hipSuccess || hipEventRecord(g_base_event[device], nullptr) != hipSuccess ||
            hipEventSynchronize(g_base_event[device]) != hipSuccess)
            return fail(AZ_ERR_HIP, "profiling base event");
        g_base_set[device] = true;
    }
    *out = g_base_event[device];
    return AZ_OK;
}

void mfma_set_device_choice(MfmaNet* m, bool on) { m->device_choice = on; }

int mfma_set_grid_limit(MfmaNet* m, int workgroups) {
    if (workgroups < 1) return fail(AZ_ERR_ARG, "workgroup limit must be >= 1");
    if (m->layers) return AZ_OK;   // the layer path has no resident workgroups
    if (workgroups == m->max_grid) return AZ_OK;
    if (hipSetDevice(m->cfg.device) != hipSuccess) return fail(AZ_ERR_HIP, "hipSetDevice");
    uint4* x = nullptr;
    if (hipMalloc((void**)&x, xres_bytes(workgroups, m->cfg.channels, m->TB_duo > 0, m->rows_main)) != hipSuccess) return fail(AZ_ERR_HIP, "hipMalloc (residual scratch)");
    if (m->xres) m->retired.push_back(m->xres);
    m->xres = x;
    m->max_grid = workgroups;
    return AZ_OK;
}

int mfma_profile_spans(MfmaNet* m, double* spans, long long cap, long long* count) {
    hipEvent_t base;
    int rc = base_event(m->cfg.device, &base);
    if (rc != AZ_OK) return rc;
    long long n = 0;
    for (size_t i = 0; i < m->ev_used && n < cap; ++i, ++n) {
        if (hipEventSynchronize(m->events[i].second) != hipSuccess) return fail(AZ_ERR_HIP, "hipEventSynchronize");
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, base, m->events[i].first) != hipSuccess || hipEventElapsedTime(&b, base, m->events[i].second) != hipSuccess)
            return fail(AZ_ERR_HIP, "hipEventElapsedTime");
        spans[2 * n] = a;
        spans[2 * n + 1] = b;
    }
    *count = (long long)m->ev_used;
    return AZ_OK;
}

void mfma_set_profiling(MfmaNet* m, bool on) {
    if (on) { hipEvent_t b; (void)base_event(m->cfg.device, &b); }
    m->profiling = on;
    m->ev_used = 0;
    m->prof_rows = 0;
}

int mfma_profile(MfmaNet* m, double* tower_ms, long long* launches, long long* rows) {
    double total = 0.0;
    for (size_t i = 0; i < m->ev_used; ++i) {
        if (hipEventSynchronize(m->events[i].second) != hipSuccess) return fail(AZ_ERR_HIP, "hipEventSynchronize");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->events[i].first, m->events[i].second) != hipSuccess) return fail(AZ_ERR_HIP, "hipEventElapsedTime");
        total += ms;
    }
    *tower_ms = total;
    *launches = (long long)m->ev_used;
    *rows = m->prof_rows;
    m->ev_used = 0;
    m->prof_rows = 0;
    return AZ_OK;
}

void mfma_destroy(MfmaNet* m) {
    if (!m) return;
    (void)hipSetDevice(m->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto& e : m->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->hq) (void)hipFree(m->hq);
    if (m->vpart) (void)hipFree(m->vpart);
    if (m->xres) (void)hipFree(m->xres);
    for (void* p : m->retired) (void)hipFree(p);
    if (m->layers) layers_destroy(m->layers);
    delete m;
}

// Policy and value tails over the head activations hq [rows][np][2q].
static void run_tails(MfmaNet* m, const uint16_t* hq_in, int rows, const int* rows_dev, int row_off, float* logits, float* values, hipStream_t st) {
    const az_net_config& c = m->cfg;
    const int np = c.board_size * c.board_size, q = m->q;
    // boards per workgroup by board size only (a board's sums run in the same order either way): large boards come in
    // chunks of a few hundred rows, which 8 per workgroup would spread over too few CUs
    if (np <= 81 && m->w4p && !(m->vpart && (m->v_ntiles + 3) / 4 > 1)) {   // both tails in one launch
        TailArgs a{hq_in, m->w3, m->b3, m->wfcT, m->bfc, logits, (const uint4*)m->w4p, m->b4p, m->w6p, m->b6, values,
                   rows, np, q, m->v_nks, m->v_ntiles, m->v_recip, rows_dev, row_off};
        const dim3 grid(3 * ((rows + 15) / 16));
        const size_t lds = (size_t)8 * 2 * np * 4;
        switch ((m->v_ntiles + 3) / 4) {
            case 1: hipLaunchKernelGGL((k_tails<1>), grid, dim3(256), lds, st, a); break;
            case 2: hipLaunchKernelGGL((k_tails<2>), grid, dim3(256), lds, st, a); break;
            case 3: hipLaunchKernelGGL((k_tails<3>), grid, dim3(256), lds, st, a); break;
            default: hipLaunchKernelGGL((k_tails<4>), grid, dim3(256), lds, st, a); break;
        }
        return;
    }
    if (np <= 81)
        hipLaunchKernelGGL((k_policy_tail<8>), dim3((rows + 7) / 8), dim3(256), (size_t)8 * 2 * np * 4, st, hq_in, m->w3, m->b3, m->wfcT, m->bfc, logits, rows,
                           np, q, rows_dev, row_off);
    else
        hipLaunchKernelGGL((k_policy_tail<2>), dim3((rows + 1) / 2), dim3(256), (size_t)2 * 2 * np * 4, st, hq_in, m->w3, m->b3, m->wfcT, m->bfc, logits, rows,
                           np, q, rows_dev, row_off);
    if (m->w4p) {
        // over an upper bound: few workgroups, each walking the live units with the grid's stride. With one workgroup per unit
        // of the bound (512 x 4 for 8192 boards, 32 x 4 of them live) the live ones, first in line, were packed onto a few
        // compute units: 414 us instead of 127 us per launch
        const dim3 vg(rows_dev ? std::min((rows + 15) / 16, 64) : (rows + 15) / 16);
        const uint4* wp = (const uint4*)m->w4p;
        const int shares = (m->v_ntiles + 3) / 4;
        if (m->vpart && shares > 1) {
            // large boards (long K, chunks of a few hundred rows): the unit tiles are dealt to `shares` workgroups per 16
            // boards, one tile per wave, and a second kernel adds the shares in index order. Chosen by shape only.
            hipLaunchKernelGGL((k_value_tail_mfma<1>), dim3(vg.x, shares), dim3(256), 0, st, hq_in, wp, m->b4p, m->w6p, m->b6, values, rows, np, q, m->v_nks,
                               m->v_ntiles, m->v_recip, rows_dev, row_off, m->vpart, shares);
            hipLaunchKernelGGL(k_value_finish, dim3((rows + 255) / 256), dim3(256), 0, st, (const float*)m->vpart, shares, m->b6, values, rows, rows_dev, row_off);
            return;
        }
        switch (shares) {
            case 1: hipLaunchKernelGGL((k_value_tail_mfma<1>), vg, dim3(256), 0, st, hq_in, wp, m->b4p, m->w6p, m->b6, values, rows, np, q, m->v_nks, m->v_ntiles, m->v_recip, rows_dev, row_off, (float*)nullptr, 0); break;
            case 2: hipLaunchKernelGGL((k_value_tail_mfma<2>), vg, dim3(256), 0, st, hq_in, wp, m->b4p, m->w6p, m->b6, values, rows, np, q, m->v_nks, m->v_ntiles, m->v_recip, rows_dev, row_off, (float*)nullptr, 0); break;
            case 3: hipLaunchKernelGGL((k_value_tail_mfma<3>), vg, dim3(256), 0, st, hq_in, wp, m->b4p, m->w6p, m->b6, values, rows, np, q, m->v_nks, m->v_ntiles, m->v_recip, rows_dev, row_off, (float*)nullptr, 0); break;
            default: hipLaunchKernelGGL((k_value_tail_mfma<4>), vg, dim3(256), 0, st, hq_in, wp, m->b4p, m->w6p, m->b6, values, rows, np, q, m->v_nks, m->v_ntiles, m->v_recip, rows_dev, row_off, (float*)nullptr, 0); break;
        }
    } else {
        const size_t vt = ((size_t)4 * q * np + 4 * c.value_hidden) * 4;
        hipLaunchKernelGGL((k_value_tail<4>), dim3((rows + 3) / 4), dim3(256), vt, st, hq_in, m->w4T, m->b4, m->w6, m->b6, values, rows, np, q,
                           c.value_hidden, rows_dev, row_off);
    }
}

// A timed launch's event pair (null when profiling is off).
static int prof_begin(MfmaNet* m, long long rows, hipStream_t st, std::pair<hipEvent_t, hipEvent_t>** out) {
    *out = nullptr;
    if (!m->profiling) return AZ_OK;
    if (m->ev_used == m->events.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return fail(AZ_ERR_HIP, "hipEventCreate");
        m->events.emplace_back(a, b);
    }
    *out = &m->events[m->ev_used++];
    m->prof_rows += rows;
    (void)hipEventRecord((*out)->first, st);
    return AZ_OK;
}

// Fused tower + tails with the given scratch (head activations for `rows` rows, residual stream for `grid_cap` workgroups).
static int fused_forward(MfmaNet* m, const void* feat, int rows, const int* rows_dev, float* logits, float* values, hipStream_t st,
                         uint16_t* hq, uint4* xres, int grid_cap, bool timed, bool static_scratch_call) {
    // static_scratch_call: the scratch is the library's static array, laid out for the 256-row kernel and kStaticGrid workgroups
    const az_net_config& c = m->cfg;
    const int C = c.channels;
    TowerParams P;
    // Few positions: the 96-row variant gives every TB_small boards a workgroup of their own, so they spread over the
    // compute units instead of queueing three boards deep on a few. It wins while its workgroups fit two rounds
    // (a round takes about 0.4 of a 256-row round); `rows` is the upper bound when the count lives on the device.
    // 64 channels: one round only (measured at 9x9, 4 blocks: 64 against 83 us up to 256 boards, 104 against 87 us at 512);
    // one round also where the alternative is the two-per-CU kernel (9x9 192ch x 12, lockstep moves of 512 games with about
    // one row per game: 12.0 ms with that kernel against 14.9 ms with two rounds of the 96-row variant; 256 games: 11.0 / 8.6).
    // 192 channels: the variant runs as 256-thread workgroups that read every weight fragment once per compute unit (9x9 x 12
    // blocks, forward of 256 boards: 0.418 ms against 0.450 ms with 512 threads; no difference at 64 boards; two of them per
    // compute unit for 257 to 512 boards are no faster than the two-per-CU main kernel: 0.709 against 0.675 ms at 512).
    // It keeps 5 k-steps of weight fragments in flight instead of 2 (a lone wave per SIMD has no other wave to cover the
    // L2 latency: forward of 1 / 40 / 96 / 160 / 256 boards 0.347 / 0.350 / 0.353 / 0.364 / 0.41 -> 0.320 / 0.330 / 0.342 / 0.360 / 0.41 ms).
    const bool lean_small = m->small_lean && !static_scratch_call;
    const int small_rounds = m->small_rounds;
    const int small_rows = m->TB_small > 0 ? small_rounds * m->cus * m->TB_small : 0;   // most rows the 96-row variant is chosen for
    const bool small = rows <= small_rows;
    // Count on the device and a bound above that limit (a search that evaluates a repeated leaf once has about one row per
    // game under a bound of inference-batch rows per game): both variants are launched and the count picks the one that
    // works; the other one's workgroups read the count and leave (a few microseconds). Not for bounds so large that
    // few rows are implausible (32 times the limit: inference batches of up to 32 rows per game).
    const bool both = !small && m->device_choice && rows_dev != nullptr && small_rows > 0 && rows <= 32 * (long long)small_rows && !static_scratch_call;
#if defined(AZ_STAMP) || defined(AZ_CLOCK)
    static unsigned long long* stamp_dev = nullptr;
    static int stamp_grid = 0;
#endif
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if (timed) {
        int rcp = prof_begin(m, rows_dev ? 0 : rows, st, &ev);   // indirect launches: the caller knows the row total
        if (rcp != AZ_OK) return rcp;
    }
    bool duo = false;
    int grid = 0;
    for (int pass = both ? 0 : 1; pass < 2; ++pass) {
        const bool sm = both ? pass == 0 : small;
        duo = !sm && m->TB_duo > 0 && (!m->duo_above_one_round || rows > m->TB * m->cus) && !static_scratch_call;
        const int TB = sm ? m->TB_small : duo ? m->TB_duo : m->TB;
        P = m->P;
        P.rows = rows; P.feat = (const uint16_t*)feat; P.hq = hq; P.stamps = nullptr; P.xres = xres; P.rows_dev = rows_dev;
        P.TB = TB;
        if (sm && m->w_tower_small) { P.w_stem = m->w_stem_small; P.w_tower = m->w_tower_small; }
        P.items = (rows + TB - 1) / TB;
        if (both) { P.gate_mode = sm ? 1 : 2; P.gate_rows = small_rows; }
        const bool lean = duo || (sm && lean_small);
        grid = std::min(P.items, lean ? 2 * grid_cap : grid_cap);
#if defined(AZ_STAMP) || defined(AZ_CLOCK)
        if (grid > stamp_grid) {
            if (stamp_dev) (void)hipFree(stamp_dev);
            (void)hipMalloc((void**)&stamp_dev, (size_t)grid * 8 * 128);
            stamp_grid = grid;
        }
        P.stamps = stamp_dev;
#endif
#ifdef AZ_CLOCK
        g_clock_dev = stamp_dev; g_clock_grid = grid;
#endif
        const size_t lds = tower_lds_bytes(C, m->NH, TB, c.se_hidden, sm ? 96 : duo ? 176 : m->rows_main, lean);
#ifdef AZ_POISON
        {
            static int* owner_dev = nullptr;
            static int owner_n = 0;
            if (grid + 1 > owner_n) {
                if (owner_dev) (void)hipFree(owner_dev);
                (void)hipMalloc((void**)&owner_dev, (size_t)(grid + 1) * 4);
                owner_n = grid + 1;
                (void)hipMemset(owner_dev, 0, (size_t)owner_n * 4);
            }
            (void)hipMemsetAsync(owner_dev, 0, (size_t)(grid + 1) * 4, st);
            P.owner = owner_dev;
            P.lds_bytes = (int)lds;
            g_owner_dev = owner_dev; g_owner_grid = grid;
            if (!static_scratch_call) {   // residual scratch and head activations start as NaN patterns
                (void)hipMemsetAsync(xres, 0xFF, xres_bytes(grid_cap, C, m->TB_duo > 0, m->rows_main), st);
                (void)hipMemsetAsync(hq, 0xFF, (size_t)rows * c.board_size * c.board_size * 2 * m->q * 2, st);
            }
        }
#endif
        switch (C) {
            case 32: hipLaunchKernelGGL((k_tower<32, 8, 1, 32>), dim3(grid), dim3(512), lds, st, P); break;
            case 64:
                if (sm) hipLaunchKernelGGL((k_tower<64, 2, 4, 32, AZ_GW64S, 96>), dim3(grid), dim3(512), lds, st, P);
                else if (m->split == 1) hipLaunchKernelGGL((k_tower<64, 2, 4, 32, 1>), dim3(grid), dim3(512), lds, st, P);
                else hipLaunchKernelGGL((k_tower<64, 4, 2, 32>), dim3(grid), dim3(512), lds, st, P);
                break;
            case 128:
                if (sm && lean_small) hipLaunchKernelGGL((k_tower<128, 1, 4, 64, AZ_GW128S, 96, true, 2>), dim3(grid), dim3(256), lds, st, P);
                else if (sm) hipLaunchKernelGGL((k_tower<128, 2, 4, 64, AZ_GW128S, 96>), dim3(grid), dim3(512), lds, st, P);
                else if (duo) hipLaunchKernelGGL((k_tower<128, 1, 4, 64, AZ_GW128D, 176, true, 2>), dim3(grid), dim3(256), lds, st, P);
                else if (m->rows_main == 384) hipLaunchKernelGGL((k_tower<128, 2, 4, 64, AZ_GW128, 384>), dim3(grid), dim3(512), lds, st, P);
                else if (m->split == 1) hipLaunchKernelGGL((k_tower<128, 2, 4, 64, AZ_GW128>), dim3(grid), dim3(512), lds, st, P);
                else hipLaunchKernelGGL((k_tower<128, 4, 2, 64>), dim3(grid), dim3(512), lds, st, P);
                break;
            case 192:
                if (sm && lean_small) hipLaunchKernelGGL((k_tower<192, 1, 4, 96, 5, 96, true, 2>), dim3(grid), dim3(256), lds, st, P);
                else if (sm) hipLaunchKernelGGL((k_tower<192, 2, 4, 96, 2, 96>), dim3(grid), dim3(512), lds, st, P);
                else if (duo) hipLaunchKernelGGL((k_tower<192, 1, 4, 96, 2, 176, true, 2>), dim3(grid), dim3(256), lds, st, P);
                else if (m->split == 1) hipLaunchKernelGGL((k_tower<192, 2, 4, 96, 2>), dim3(grid), dim3(512), lds, st, P);
                else hipLaunchKernelGGL((k_tower<192, 4, 2, 96>), dim3(grid), dim3(512), lds, st, P);
                break;
            default: return fail(AZ_ERR_ARG, "unsupported channel count");
        }
    }
    if (ev) (void)hipEventRecord(ev->second, st);
    run_tails(m, hq, rows, rows_dev, 0, logits, values, st);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(AZ_ERR_HIP, std::string("bf16 forward: ") + hipGetErrorString(err));
#ifdef AZ_STAMP
    {   // diagnostic build: print where a workgroup's cycles go (mean over workgroups)
        const int nwv = (duo || (small && lean_small)) ? 4 : 8;   // waves per workgroup
        std::vector<unsigned long long> hst((size_t)grid * nwv * 16);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hst.data(), P.stamps, hst.size() * 8, hipMemcpyDeviceToHost);
        double sum[16] = {0}, conv_w[8] = {0};
        for (int g = 0; g < grid; ++g) {
            for (int i = 0; i < 16; ++i) sum[i] += (double)hst[(size_t)g * nwv * 16 + i];          // wave 0's view
            for (int w = 0; w < nwv; ++w) conv_w[w] += (double)hst[((size_t)g * nwv + w) * 16 + 2];
        }
        fprintf(stderr, "[AZ_STAMP] conv gemm cycles per wave:");
        for (int w = 0; w < nwv; ++w) fprintf(stderr, " %.0f", conv_w[w] / grid);
        fprintf(stderr, "\n");
        const char* nm[16] = {"load+stem gemm", "stem epilogue", "conv gemm (2/blk)", "conv1 epilogue", "SE scale", "residual epilogue", "head gemm+store",
                              "resv issue+bias+y image", "SE pooling", "SE wait+sync", "SE fc1", "SE fc2", "-", "-", "-", "-"};
        double tot = 0; for (int i = 0; i < 12; ++i) tot += sum[i];
        fprintf(stderr, "[AZ_STAMP] rows=%d grid=%d mean cycles/WG=%.0f\n", rows, grid, tot / grid);
        for (int i = 0; i < 12; ++i) fprintf(stderr, "  %-20s %10.0f  %5.1f%%\n", nm[i], sum[i] / grid, 100.0 * sum[i] / tot);
        {   // spread over workgroups: busy cycles, start and end relative to the first start
            unsigned long long t0 = ~0ull, t1 = 0, smax = 0;
            double bmin = 1e30, bmax = 0;
            for (int g = 0; g < grid; ++g) {
                const unsigned long long* r = &hst[(size_t)g * nwv * 16];
                double b = 0; for (int i = 0; i < 12; ++i) b += (double)r[i];
                bmin = std::min(bmin, b); bmax = std::max(bmax, b);
                t0 = std::min(t0, r[12]); t1 = std::max(t1, r[13]);
            }
            for (int g = 0; g < grid; ++g) smax = std::max(smax, hst[(size_t)g * nwv * 16 + 12] - t0);
            fprintf(stderr, "[AZ_STAMP] busy cycles/WG min %.0f max %.0f; first start to last end %llu; latest start +%llu\n", bmin, bmax, t1 - t0, smax);
            int hist[10] = {0};
            for (int g = 0; g < grid; ++g) { const unsigned long long e = hst[(size_t)g * nwv * 16 + 13] - t0; hist[std::min(9, (int)(e * 10 / (t1 - t0 + 1)))]++; }
            if (const char* dump = getenv("AZ_STAMP_DUMP")) {   // raw per-wave records: wg wave hw_id xcc_id start end busy conv
                FILE* f = fopen(dump, "w");
                for (int g = 0; f && g < grid; ++g)
                    for (int w = 0; w < nwv; ++w) {
                        const unsigned long long* r = &hst[((size_t)g * nwv + w) * 16];
                        double b = 0; for (int i = 0; i < 12; ++i) b += (double)r[i];
                        fprintf(f, "%d %d %llu %llu %llu %llu %.0f %llu\n", g, w, r[14], r[15], r[12], r[13], b, r[2]);
                    }
                if (f) fclose(f);
            }
            fprintf(stderr, "[AZ_STAMP] end-time deciles:"); for (int i = 0; i < 10; ++i) fprintf(stderr, " %d", hist[i]); fprintf(stderr, "\n");
        }
    }
#endif
    return AZ_OK;
}

// Launch sequences an indirect forward over `max_rows` rows makes: 1 for the fused tower; for the layer path one per
// activation-array capacity of the upper bound (az_net_reserve(max_rows) grows the arrays, up to 2 GiB each, so that it is 1;
// each pass past the live rows costs its launches of workgroups that leave at once).
int mfma_indirect_passes(const MfmaNet* m, int max_rows) {
    if (!m->layers) return 1;
    const int cap = std::max(1, std::min(layers_cap_boards(m->layers), m->hq_rows));
    return (max_rows + cap - 1) / cap;
}

// Whether an indirect forward over `max_rows` rows consists of kernel launches only, with nothing that a stream capture
// could not record: one pass, scratch already reserved (no allocation), launch timing off (its events are per launch).
bool mfma_capturable(const MfmaNet* m, int max_rows) {
    return !m->profiling && mfma_indirect_passes(m, max_rows) == 1 && max_rows <= m->hq_rows;
}

// rows_dev == null: `rows` positions. rows_dev != null: *rows_dev positions (read on the device, at most `rows`).
int mfma_forward(MfmaNet* m, const void* feat, int rows, const int* rows_dev, float* logits, float* values, hipStream_t st) {
    const az_net_config& c = m->cfg;
    const int np = c.board_size * c.board_size;
    if (m->layers) {
        const int chunk = layers_chunk_boards(m->layers), A = np + 1;   // hq / vpart hold one pass (mfma_reserve)
        std::pair<hipEvent_t, hipEvent_t>* evl = nullptr;
        int rcp = prof_begin(m, rows_dev ? 0 : rows, st, &evl);
        if (rcp != AZ_OK) return rcp;
        // direct: even shares, no small last chunk. indirect: whole chunks over the upper bound; the workgroups of a chunk
        // past the device-side count leave at once, so a batch that fills one chunk pays for the others' empty launches only
        // (a pass never exceeds what the head scratch holds: a failed allocation in mfma_reserve may have left the activation
        // arrays larger than it)
        const int nchunks = (rows + chunk - 1) / chunk;
        const int per = std::min(m->hq_rows, rows_dev ? layers_cap_boards(m->layers) : (rows + nchunks - 1) / nchunks);
        if (per < 1) return fail(AZ_ERR_STATE, "bf16 forward (layers): no head scratch");
        for (int off = 0; off < rows; off += per) {
            const int nb = std::min(per, rows - off);
            int rc = layers_tower(m->layers, (const uint16_t*)feat + (size_t)off * np * 32, nb, m->hq, st, rows_dev, off);
            if (rc != AZ_OK) return rc;
            run_tails(m, m->hq, nb, rows_dev, off, logits + (size_t)off * A, values + off, st);
        }
        if (evl) (void)hipEventRecord(evl->second, st);
        hipError_t errl = hipGetLastError();
        if (errl != hipSuccess) return fail(AZ_ERR_HIP, std::string("bf16 forward (layers): ") + hipGetErrorString(errl));
        return AZ_OK;
    }
    if (rows > m->hq_rows) {   // first use, or a larger batch than any before: az_net_reserve avoids this in a steady state
        int rc = mfma_reserve(m, rows + rows / 4 + 16);
        if (rc != AZ_OK) return rc;
    }
    return fused_forward(m, feat, rows, rows_dev, logits, values, st, m->hq, m->xres, m->max_grid, true, false);
}

// ---- the entry point that allocates nothing: scratch in a device array of the library itself -----------------------
// One array per GPU (every GPU the library runs on holds its copy until the process ends), shared by every network and
// every call on that GPU: calls that overlap on one GPU race. Layout: residual-stream scratch for kStaticGrid workgroups,
// then the head activations.
constexpr size_t kStaticBytes = (size_t)64 << 20;
constexpr int kStaticGrid = 256;
static __device__ __align__(256) unsigned char scratch[kStaticBytes];
static unsigned char* g_static_ptr[64];   // per device: the array's address on that device

static int static_scratch(int device, unsigned char** out) {
    if (device < 0 || device >= 64) return fail(AZ_ERR_ARG, "device index out of range");
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!g_static_ptr[device]) {
        if (hipSetDevice(device) != hipSuccess) return fail(AZ_ERR_HIP, "hipSetDevice");
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(AZ_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_static_ptr[device] = (unsigned char*)p;
    }
    *out = g_static_ptr[device];
    return AZ_OK;
}

static size_t static_xres_bytes(const MfmaNet* m) { return (size_t)kStaticGrid * 256 * (size_t)m->cfg.channels * 2; }

long long mfma_static_max_rows(const MfmaNet* m) {
    if (m->layers || m->cfg.board_size > 9) return 0;   // larger boards: other workgroup sizes and a value tail with scratch of its own
    const size_t np = (size_t)m->cfg.board_size * m->cfg.board_size;
    return (long long)((kStaticBytes - static_xres_bytes(m)) / (np * 2 * m->q * 2));
}

int mfma_forward_static(MfmaNet* m, const void* feat, int rows, float* logits, float* values, hipStream_t st) {
    if (m->layers || m->cfg.board_size > 9) return fail(AZ_ERR_ARG, "az_net_forward_static: fused bf16 networks on boards up to 9x9 only");
    const long long cap = mfma_static_max_rows(m);
    if (rows > cap)
        return fail(AZ_ERR_CAPACITY, "az_net_forward_static: " + std::to_string(rows) + " rows exceed the static scratch (at most " +
                                         std::to_string(cap) + " rows for this network shape)");
    unsigned char* base = nullptr;
    int rc = static_scratch(m->cfg.device, &base);
    if (rc != AZ_OK) return rc;
    return fused_forward(m, feat, rows, nullptr, logits, values, st, (uint16_t*)(base + static_xres_bytes(m)), (uint4*)base, kStaticGrid, false, true);
}

}  // namespace az

#ifdef AZ_CLOCK
// Diagnostic build only: per-workgroup in-kernel clock (MHz) of the last tower launch, ascending. Returns the workgroup count.
extern "C" int az_debug_tower_clock(double* mhz_out, int capacity) {
    if (!az::g_clock_dev || az::g_clock_grid <= 0) return 0;
    std::vector<unsigned long long> h((size_t)az::g_clock_grid * 2);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(h.data(), az::g_clock_dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    std::vector<double> mhz;
    for (int g = 0; g < az::g_clock_grid; ++g)
        if (h[2 * g + 1] > 0) mhz.push_back((double)h[2 * g] / (double)h[2 * g + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const int n = std::min((int)mhz.size(), capacity);
    for (int i = 0; i < n; ++i) mhz_out[i] = mhz[(size_t)i * mhz.size() / std::max(1, n)];
    return (int)mhz.size();
}
#endif

#ifdef AZ_POISON
// Diagnostic build only: residual-scratch slots that two workgroups of the last tower launch held at the same time.
extern "C" int az_debug_slot_clashes(void) {
    if (!az::g_owner_dev) return -1;
    int v = -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(&v, az::g_owner_dev + az::g_owner_grid, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}
#endif
