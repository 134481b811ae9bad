// Shared declarations of the network code (parameter blob layout, bf16 MFMA path entry points).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/azamd.h"

namespace az {

struct BnOff { int64_t w, b, mean, var; };
struct BlockOff { int64_t conv1; BnOff bn1; int64_t conv2; BnOff bn2; int64_t se_w1, se_b1, se_w2, se_b2; };
// Offsets (in float32 words) into the flat parameter blob: the state_dict of model.py:61-87 in
// registration order, every tensor flattened row-major, num_batches_tracked entries left out.
struct NetLayout {
    int64_t stem_w; BnOff stem_bn;
    std::vector<BlockOff> blocks;
    int64_t ph0_w; BnOff ph0_bn; int64_t ph3_w; BnOff ph3_bn; int64_t pfc_w, pfc_b;
    int64_t vh0_w; BnOff vh0_bn; int64_t vh4_w, vh4_b, vh6_w, vh6_b;
    int64_t total;
};
void net_layout(const az_net_config& c, NetLayout& L);
void fold_bn(const float* P, int64_t w_off, const BnOff& bn, int cout, int64_t per_out, std::vector<float>& w,
             std::vector<float>& b);

// bf16 layer-by-layer tower for shapes the fused kernel does not hold in LDS (az_net_layers.hip)
struct LayerNet;
bool layers_supported(const az_net_config& c);
int layers_create(const az_net_config& c, const NetLayout& L, const float* params, LayerNet** out);
void layers_destroy(LayerNet* m);
int layers_chunk_boards(const LayerNet* m);
int layers_cap_boards(const LayerNet* m);
int layers_reserve(LayerNet* m, int boards);
int layers_tower(LayerNet* m, const uint16_t* feat, int boards, uint16_t* hq, hipStream_t st, const int* rows_dev = nullptr, int board_off = 0);

// Changes whenever something a captured launch sequence of the network depends on changes: weights and scratch pointers
// (az_net_load, az_net_reserve), launch shape (az_net_set_tower_workgroups, az_net_set_device_choice). A HIP graph recorded
// over az_net_forward_indirect is stale once this differs from its value at capture.
unsigned long long net_generation(const az_net* net);

// bf16 MFMA path (az_net_mfma.hip)
struct MfmaNet;
bool mfma_supported(const az_net_config& c);
bool mfma_is_fused(const MfmaNet* m);
int mfma_create(const az_net_config& c, const NetLayout& L, const float* params, MfmaNet** out);
void mfma_destroy(MfmaNet* m);
void mfma_set_profiling(MfmaNet* m, bool on);
int mfma_profile(MfmaNet* m, double* tower_ms, long long* launches, long long* rows);
int mfma_profile_spans(MfmaNet* m, double* spans, long long cap, long long* count);
int mfma_set_grid_limit(MfmaNet* m, int workgroups);
void mfma_set_device_choice(MfmaNet* m, bool on);
int mfma_forward(MfmaNet* m, const void* feat_bf16_nhwc, int rows, const int* rows_dev, float* logits, float* values, hipStream_t st);
int mfma_reserve(MfmaNet* m, int rows);
int mfma_forward_static(MfmaNet* m, const void* feat_bf16_nhwc, int rows, float* logits, float* values, hipStream_t st);
long long mfma_static_max_rows(const MfmaNet* m);
int mfma_indirect_passes(const MfmaNet* m, int max_rows);
bool mfma_capturable(const MfmaNet* m, int max_rows);

}  // namespace az
