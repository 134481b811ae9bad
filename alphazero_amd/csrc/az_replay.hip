// Training-batch assembly on the device (go_zero/train.py:147-176, :187-207): gather sampled positions
// from the replay store and apply one of the 8 dihedral board symmetries to the feature planes and to
// the policy target. The store keeps what to_tensor() produces in compact form: the 2H+3 binary planes
// bit-packed (row-major, most significant bit first, as numpy.packbits does), the two capture planes
// as one float each, the policy as float32. HBM-bound: 1 wave per sample, coalesced float4-free
// stores of C*n*n + A floats, reads of ~(C*n*n)/8 + 4*A bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/azamd.h"

namespace az {
int fail(int code, const std::string& msg);

// perm[sym][dst] = source point of destination point dst under symmetry sym (built on the host by
// transforming an index grid with the same numpy calls the upstream code uses).
__global__ __launch_bounds__(256) void k_replay_gather(const uint8_t* __restrict__ bits, const float* __restrict__ caps,
                                                       const float* __restrict__ pi, const int16_t* __restrict__ perm,
                                                       const int32_t* __restrict__ idx, const int32_t* __restrict__ sym, int np,
                                                       int nbin, int row_bytes, float* __restrict__ out_states,
                                                       float* __restrict__ out_pi) {
    const int s = blockIdx.x;
    const size_t src = (size_t)idx[s];
    const int16_t* pm = perm + (size_t)sym[s] * np;
    const uint8_t* b = bits + src * row_bytes;
    const int C = nbin + 2, A = np + 1;
    float* os = out_states + (size_t)s * C * np;
    for (int i = threadIdx.x; i < nbin * np; i += blockDim.x) {
        const int c = i / np, p = i - c * np;
        const int bit = c * np + pm[p];
        os[i] = (b[bit >> 3] >> (7 - (bit & 7))) & 1 ? 1.0f : 0.0f;
    }
    for (int i = threadIdx.x; i < 2 * np; i += blockDim.x) os[nbin * np + i] = caps[src * 2 + i / np];
    float* op = out_pi + (size_t)s * A;
    const float* ps = pi + src * A;
    for (int i = threadIdx.x; i < A; i += blockDim.x) op[i] = i < np ? ps[pm[i]] : ps[np];
}
// to_tensor() planes (float32 [count][nbin+2][np]) -> the store's compact form: one thread per packed byte,
// plus the two capture-plane values (their first element) per position.
__global__ __launch_bounds__(256) void k_pack_planes(const float* __restrict__ planes, int count, int np, int nbin, int row_bytes,
                                                     uint8_t* __restrict__ bits, float* __restrict__ caps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count * row_bytes) return;
    const int s = i / row_bytes, byte = i - s * row_bytes;
    const float* src = planes + (size_t)s * (nbin + 2) * np;
    unsigned v = 0;
    for (int k = 0; k < 8; ++k) {
        const int bit = byte * 8 + k;
        if (bit < nbin * np && src[bit] > 0.5f) v |= 0x80u >> k;
    }
    bits[i] = (uint8_t)v;
    if (byte < 2) caps[s * 2 + byte] = src[(size_t)(nbin + byte) * np];
}
}  // namespace az

using namespace az;

extern "C" int az_pack_planes(const float* planes, int32_t count, int32_t board_size, int32_t history_len, int32_t row_bytes, uint8_t* bits,
                              float* caps, void* stream) {
    if (!planes || !bits || !caps) return fail(AZ_ERR_ARG, "null argument");
    if (board_size < 2 || board_size > 19 || history_len < 1 || history_len > 16) return fail(AZ_ERR_ARG, "bad board_size / history_len");
    const int np = board_size * board_size, nbin = 2 * history_len + 3;
    if (row_bytes < (nbin * np + 7) / 8 || row_bytes < 2) return fail(AZ_ERR_ARG, "row_bytes too small for the packed planes");
    if (count <= 0) return AZ_OK;
    hipLaunchKernelGGL(k_pack_planes, dim3((count * row_bytes + 255) / 256), dim3(256), 0, (hipStream_t)stream, planes, count, np, nbin, row_bytes,
                       bits, caps);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(AZ_ERR_HIP, std::string("k_pack_planes: ") + hipGetErrorString(err));
    return AZ_OK;
}

extern "C" int az_replay_gather(const uint8_t* bits, const float* caps, const float* pi, const int16_t* perm, const int32_t* idx,
                                const int32_t* sym, int32_t count, int32_t board_size, int32_t history_len, int32_t row_bytes,
                                float* out_states, float* out_pi, void* stream) {
    if (!bits || !caps || !pi || !perm || !idx || !sym || !out_states || !out_pi) return fail(AZ_ERR_ARG, "null argument");
    if (board_size < 2 || board_size > 19 || history_len < 1 || history_len > 16) return fail(AZ_ERR_ARG, "bad board_size / history_len");
    const int np = board_size * board_size, nbin = 2 * history_len + 3;
    if (row_bytes < (nbin * np + 7) / 8) return fail(AZ_ERR_ARG, "row_bytes too small for the packed planes");
    if (count <= 0) return AZ_OK;
    hipLaunchKernelGGL(k_replay_gather, dim3(count), dim3(256), 0, (hipStream_t)stream, bits, caps, pi, perm, idx, sym, np, nbin, row_bytes,
                       out_states, out_pi);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(AZ_ERR_HIP, std::string("k_replay_gather: ") + hipGetErrorString(err));
    return AZ_OK;
}
