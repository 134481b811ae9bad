// Policy/value network inference for libazamd (model.py:43-97 AlphaZeroNet.forward in eval mode).
//
// Two precisions behind one C ABI:
//   AZ_NET_F32  - plain float32 kernels, NCHW input, any board size / width. Used for the 1e-5
//                 parity check against the PyTorch module and as the fallback shape path.
//   AZ_NET_BF16 - the production path: bf16 MFMA kernels in az_net_mfma.hip.
// BatchNorm (eval mode, eps 1e-5) is folded into the preceding bias-free convolution at load time.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/azamd.h"
#include "az_net.h"

namespace az {
extern thread_local std::string g_err;
int fail(int code, const std::string& msg);
}  // namespace az
using namespace az;

#define HIP_TRY(x)                                                                   \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) return fail(AZ_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

// ---- float32 reference kernels ---------------------------------------------------------------
// out[r][co][p] = act(bias[co] + sum_{ci,tap} w[co][ci][tap] * in[r][ci][p+tap])   (ks = 1 or 3)
__global__ void k_conv_f32(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                           float* __restrict__ out, int rows, int cin, int cout, int n, int ks, int relu) {
    const int np = n * n;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * cout * np) return;
    const int p = (int)(idx % np), co = (int)((idx / np) % cout), r = (int)(idx / ((long long)np * cout));
    const int y = p / n, x = p % n, half = ks / 2;
    const float* inr = in + (size_t)r * cin * np;
    const float* wr = w + (size_t)co * cin * ks * ks;
    float acc = 0.f;
    for (int ci = 0; ci < cin; ++ci)
        for (int dy = 0; dy < ks; ++dy) {
            const int yy = y + dy - half;
            if (yy < 0 || yy >= n) continue;
            for (int dx = 0; dx < ks; ++dx) {
                const int xx = x + dx - half;
                if (xx < 0 || xx >= n) continue;
                acc = fmaf(wr[(ci * ks + dy) * ks + dx], inr[ci * np + yy * n + xx], acc);
            }
        }
    acc += bias[co];
    out[idx] = relu ? fmaxf(acc, 0.f) : acc;
}

// scale[r][c] = sigmoid(W2 relu(W1 mean_p(y[r][:,p]) + b1) + b2)   (model.py:9-22); one block per row
__global__ void k_se_f32(const float* __restrict__ y, const float* __restrict__ w1, const float* __restrict__ b1,
                         const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ scale, int c, int hid,
                         int np) {
    extern __shared__ float sm[];
    float* pooled = sm;
    float* hidden = sm + c;
    const int r = blockIdx.x;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        float s = 0.f;
        for (int p = 0; p < np; ++p) s += y[((size_t)r * c + ch) * np + p];
        pooled[ch] = s / (float)np;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < hid; j += blockDim.x) {
        float s = b1[j];
        for (int ch = 0; ch < c; ++ch) s = fmaf(w1[j * c + ch], pooled[ch], s);
        hidden[j] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        float s = b2[ch];
        for (int j = 0; j < hid; ++j) s = fmaf(w2[ch * hid + j], hidden[j], s);
        scale[(size_t)r * c + ch] = 1.f / (1.f + expf(-s));
    }
}

// x = relu(y * scale + x)   (model.py:39-40); scale may be null (no SE)
__global__ void k_residual_f32(float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ scale, long long total,
                               int np) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float s = scale ? scale[idx / np] : 1.f;
    x[idx] = fmaxf(fmaf(y[idx], s, x[idx]), 0.f);
}

// out[r][j] = act(b[j] + sum_i w[j][i] in[r][i]); act: 0 none, 1 relu, 2 tanh
__global__ void k_linear_f32(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                             float* __restrict__ out, int rows, int nin, int nout, int act) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * nout) return;
    const int j = (int)(idx % nout), r = (int)(idx / nout);
    float s = 0.f;
    for (int i = 0; i < nin; ++i) s = fmaf(w[(size_t)j * nin + i], in[(size_t)r * nin + i], s);
    s += b[j];
    out[idx] = act == 1 ? fmaxf(s, 0.f) : act == 2 ? tanhf(s) : s;
}

inline unsigned nblk(long long total, int bs) { return (unsigned)((total + bs - 1) / bs); }

}  // namespace

// ---- parameter blob layout (state_dict order of model.py:61-87, num_batches_tracked left out) ------
namespace az {

void net_layout(const az_net_config& c, NetLayout& L) {
    const int C = c.channels, Cin = c.in_planes, np = c.board_size * c.board_size, A = np + 1, q = C / 4;
    int64_t off = 0;
    auto take = [&](int64_t cnt) { int64_t o = off; off += cnt; return o; };
    auto bn = [&](BnOff& b, int ch) { b.w = take(ch); b.b = take(ch); b.mean = take(ch); b.var = take(ch); };
    L.stem_w = take((int64_t)C * Cin * 9); bn(L.stem_bn, C);
    L.blocks.resize(c.blocks);
    for (auto& b : L.blocks) {
        b.conv1 = take((int64_t)C * C * 9); bn(b.bn1, C);
        b.conv2 = take((int64_t)C * C * 9); bn(b.bn2, C);
        if (c.use_se) {
            b.se_w1 = take((int64_t)c.se_hidden * C); b.se_b1 = take(c.se_hidden);
            b.se_w2 = take((int64_t)C * c.se_hidden); b.se_b2 = take(C);
        }
    }
    L.ph0_w = take((int64_t)q * C); bn(L.ph0_bn, q);
    L.ph3_w = take((int64_t)2 * q); bn(L.ph3_bn, 2);
    L.pfc_w = take((int64_t)A * 2 * np); L.pfc_b = take(A);
    L.vh0_w = take((int64_t)q * C); bn(L.vh0_bn, q);
    L.vh4_w = take((int64_t)c.value_hidden * q * np); L.vh4_b = take(c.value_hidden);
    L.vh6_w = take(c.value_hidden); L.vh6_b = take(1);
    L.total = off;
}

// Fold eval-mode BatchNorm into a bias-free conv: w' = w * g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps).
void fold_bn(const float* P, int64_t w_off, const BnOff& bn, int cout, int64_t per_out, std::vector<float>& w, std::vector<float>& b) {
    w.resize((size_t)cout * per_out);
    b.resize(cout);
    for (int co = 0; co < cout; ++co) {
        const double s = (double)P[bn.w + co] / sqrt((double)P[bn.var + co] + 1e-5);
        for (int64_t i = 0; i < per_out; ++i) w[(size_t)co * per_out + i] = (float)((double)P[w_off + co * per_out + i] * s);
        b[co] = (float)((double)P[bn.b + co] - (double)P[bn.mean + co] * s);
    }
}

}  // namespace az

struct F32Conv { float* w = nullptr; float* b = nullptr; };
struct F32Block { F32Conv c1, c2; float *se_w1 = nullptr, *se_b1 = nullptr, *se_w2 = nullptr, *se_b2 = nullptr; };

struct az_net {
    az_net_config cfg;
    NetLayout L;
    bool loaded = false;
    std::vector<void*> allocs;
    // f32 path
    F32Conv stem, ph0, ph3, vh0;
    std::vector<F32Block> blocks;
    float *pfc_w = nullptr, *pfc_b = nullptr, *vh4_w = nullptr, *vh4_b = nullptr, *vh6_w = nullptr, *vh6_b = nullptr;
    // scratch (grown on demand)
    int scratch_rows = 0;
    float *x = nullptr, *h = nullptr, *y = nullptr, *scale = nullptr, *p0 = nullptr, *p1 = nullptr, *v0 = nullptr, *v1 = nullptr;
    // bf16 MFMA path
    MfmaNet* mfma = nullptr;
    unsigned long long generation = 1;   // see net_generation
};

namespace az {
unsigned long long net_generation(const az_net* net) { return net ? net->generation : 0; }
}

static int upload(az_net* net, const std::vector<float>& v, float** out) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(float)));
    net->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = (float*)p;
    return AZ_OK;
}
static int upload_raw(az_net* net, const float* src, int64_t cnt, float** out) {
    return upload(net, std::vector<float>(src, src + cnt), out);
}

extern "C" {

int az_net_create(const az_net_config* c, az_net** out) {
    if (!c || !out) return fail(AZ_ERR_ARG, "null argument");
    if (c->board_size < 2 || c->board_size > 19 || c->channels < 4 || c->channels % 4 || c->blocks < 0 || c->in_planes < 1 ||
        c->value_hidden < 1 || (c->use_se && c->se_hidden < 1))
        return fail(AZ_ERR_ARG, "unsupported network configuration");
    if (c->precision != AZ_NET_BF16 && c->precision != AZ_NET_F32) return fail(AZ_ERR_ARG, "bad precision");
    if (c->precision == AZ_NET_BF16 && !mfma_supported(*c))
        return fail(AZ_ERR_ARG, "bf16 MFMA path: unsupported shape (channels 32/64/128/192 on boards up to 9x9, 64/128/192/256 with at most 32 input planes on boards up to 19x19)");
    az_net* net = new az_net();
    net->cfg = *c;
    net_layout(*c, net->L);
    net->blocks.resize(c->blocks);
    *out = net;
    return AZ_OK;
}

int az_net_destroy(az_net* net) {
    if (!net) return AZ_OK;
    (void)hipSetDevice(net->cfg.device);
    (void)hipDeviceSynchronize();
    for (void* p : net->allocs) (void)hipFree(p);
    for (float* p : {net->x, net->h, net->y, net->scale, net->p0, net->p1, net->v0, net->v1})
        if (p) (void)hipFree(p);
    if (net->mfma) mfma_destroy(net->mfma);
    delete net;
    return AZ_OK;
}

int64_t az_net_param_count(const az_net* net) { return net ? net->L.total : 0; }

int az_net_load(az_net* net, const float* P, int64_t count) {
    if (!net || !P) return fail(AZ_ERR_ARG, "null argument");
    if (count != net->L.total) return fail(AZ_ERR_ARG, "parameter blob has the wrong length");
    HIP_TRY(hipSetDevice(net->cfg.device));
    net->generation += 1;
    const az_net_config& c = net->cfg;
    const NetLayout& L = net->L;
    const int C = c.channels, np = c.board_size * c.board_size, q = C / 4;
    if (c.precision == AZ_NET_BF16) {
        if (net->mfma) { mfma_destroy(net->mfma); net->mfma = nullptr; }
        int rc = mfma_create(c, L, P, &net->mfma);
        if (rc != AZ_OK) return rc;
        net->loaded = true;
        return AZ_OK;
    }
    for (void* p : net->allocs) (void)hipFree(p);
    net->allocs.clear();
    std::vector<float> w, b;
    int rc;
#define UP(vec, dst) if ((rc = upload(net, vec, &(dst))) != AZ_OK) return rc
    fold_bn(P, L.stem_w, L.stem_bn, C, (int64_t)c.in_planes * 9, w, b); UP(w, net->stem.w); UP(b, net->stem.b);
    for (int i = 0; i < c.blocks; ++i) {
        const BlockOff& bo = L.blocks[i];
        F32Block& fb = net->blocks[i];
        fold_bn(P, bo.conv1, bo.bn1, C, (int64_t)C * 9, w, b); UP(w, fb.c1.w); UP(b, fb.c1.b);
        fold_bn(P, bo.conv2, bo.bn2, C, (int64_t)C * 9, w, b); UP(w, fb.c2.w); UP(b, fb.c2.b);
        if (c.use_se) {
            if ((rc = upload_raw(net, P + bo.se_w1, (int64_t)c.se_hidden * C, &fb.se_w1))) return rc;
            if ((rc = upload_raw(net, P + bo.se_b1, c.se_hidden, &fb.se_b1))) return rc;
            if ((rc = upload_raw(net, P + bo.se_w2, (int64_t)C * c.se_hidden, &fb.se_w2))) return rc;
            if ((rc = upload_raw(net, P + bo.se_b2, C, &fb.se_b2))) return rc;
        }
    }
    fold_bn(P, L.ph0_w, L.ph0_bn, q, C, w, b); UP(w, net->ph0.w); UP(b, net->ph0.b);
    fold_bn(P, L.ph3_w, L.ph3_bn, 2, q, w, b); UP(w, net->ph3.w); UP(b, net->ph3.b);
    fold_bn(P, L.vh0_w, L.vh0_bn, q, C, w, b); UP(w, net->vh0.w); UP(b, net->vh0.b);
#undef UP
    if ((rc = upload_raw(net, P + L.pfc_w, (int64_t)(np + 1) * 2 * np, &net->pfc_w))) return rc;
    if ((rc = upload_raw(net, P + L.pfc_b, np + 1, &net->pfc_b))) return rc;
    if ((rc = upload_raw(net, P + L.vh4_w, (int64_t)c.value_hidden * q * np, &net->vh4_w))) return rc;
    if ((rc = upload_raw(net, P + L.vh4_b, c.value_hidden, &net->vh4_b))) return rc;
    if ((rc = upload_raw(net, P + L.vh6_w, c.value_hidden, &net->vh6_w))) return rc;
    if ((rc = upload_raw(net, P + L.vh6_b, 1, &net->vh6_b))) return rc;
    net->loaded = true;
    return AZ_OK;
}

static int ensure_scratch(az_net* net, int rows) {
    if (rows <= net->scratch_rows) return AZ_OK;
    const az_net_config& c = net->cfg;
    const size_t np = (size_t)c.board_size * c.board_size, C = c.channels, q = C / 4;
    HIP_TRY(hipDeviceSynchronize());
    net->scratch_rows = 0;   // nothing usable until every buffer below exists
    for (float** p : {&net->x, &net->h, &net->y, &net->scale, &net->p0, &net->p1, &net->v0, &net->v1}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    const int cap = rows + rows / 2 + 8;
    HIP_TRY(hipMalloc((void**)&net->x, cap * C * np * 4)); HIP_TRY(hipMalloc((void**)&net->h, cap * C * np * 4));
    HIP_TRY(hipMalloc((void**)&net->y, cap * C * np * 4)); HIP_TRY(hipMalloc((void**)&net->scale, cap * C * 4));
    HIP_TRY(hipMalloc((void**)&net->p0, cap * q * np * 4)); HIP_TRY(hipMalloc((void**)&net->p1, cap * 2 * np * 4));
    HIP_TRY(hipMalloc((void**)&net->v0, cap * q * np * 4)); HIP_TRY(hipMalloc((void**)&net->v1, (size_t)cap * c.value_hidden * 4));
    net->scratch_rows = cap;
    return AZ_OK;
}

int az_net_is_fused(az_net* net, int32_t* fused) {
    if (!net || !fused) return fail(AZ_ERR_ARG, "null argument");
    *fused = (net->cfg.precision == AZ_NET_BF16 && net->mfma && mfma_is_fused(net->mfma)) ? 1 : 0;
    return AZ_OK;
}

int az_net_profile_spans(az_net* net, double* spans_ms, int64_t capacity, int64_t* count) {
    if (!net || !count || (capacity > 0 && !spans_ms)) return fail(AZ_ERR_ARG, "null argument");
    if (net->cfg.precision != AZ_NET_BF16 || !net->mfma) return fail(AZ_ERR_STATE, "az_net_profile_spans: bf16 network not loaded");
    long long n = 0;
    int rc = mfma_profile_spans(net->mfma, spans_ms, capacity, &n);
    *count = n;
    return rc;
}

int az_net_set_tower_workgroups(az_net* net, int32_t workgroups) {
    if (!net) return fail(AZ_ERR_ARG, "null network");
    if (net->cfg.precision != AZ_NET_BF16 || !net->mfma) return fail(AZ_ERR_STATE, "az_net_set_tower_workgroups: bf16 network not loaded");
    net->generation += 1;
    return mfma_set_grid_limit(net->mfma, workgroups);
}

int az_net_set_device_choice(az_net* net, int32_t on) {
    if (!net) return fail(AZ_ERR_ARG, "null network");
    if (net->cfg.precision != AZ_NET_BF16 || !net->mfma) return fail(AZ_ERR_STATE, "az_net_set_device_choice: bf16 network not loaded");
    net->generation += 1;
    mfma_set_device_choice(net->mfma, on != 0);
    return AZ_OK;
}

int az_net_profile(az_net* net, int32_t enable, double* tower_ms, int64_t* launches, int64_t* rows) {
    if (!net) return fail(AZ_ERR_ARG, "null network");
    if (net->cfg.precision != AZ_NET_BF16 || !net->mfma) return fail(AZ_ERR_STATE, "az_net_profile: bf16 network not loaded");
    if (tower_ms && launches && rows) {
        long long l = 0, r = 0;
        int rc = mfma_profile(net->mfma, tower_ms, &l, &r);
        if (rc != AZ_OK) return rc;
        *launches = l;
        *rows = r;
    }
    mfma_set_profiling(net->mfma, enable != 0);
    return AZ_OK;
}

int az_net_forward_indirect(az_net* net, const void* feat, const int32_t* rows_dev, int32_t max_rows, float* logits, float* values,
                            void* stream_) {
    if (!net || !feat || !rows_dev || !logits || !values) return fail(AZ_ERR_ARG, "null argument");
    if (!net->loaded) return fail(AZ_ERR_STATE, "az_net_forward_indirect before az_net_load");
    if (net->cfg.precision != AZ_NET_BF16) return fail(AZ_ERR_ARG, "az_net_forward_indirect: bf16 networks only");
    if (max_rows <= 0) return AZ_OK;
    return mfma_forward(net->mfma, feat, max_rows, rows_dev, logits, values, (hipStream_t)stream_);
}

int az_net_reserve(az_net* net, int32_t rows) {
    if (!net) return fail(AZ_ERR_ARG, "null network");
    if (!net->loaded) return fail(AZ_ERR_STATE, "az_net_reserve before az_net_load");
    if (rows < 1) return fail(AZ_ERR_ARG, "az_net_reserve: rows must be >= 1");
    net->generation += 1;
    if (net->cfg.precision == AZ_NET_BF16) return mfma_reserve(net->mfma, rows);
    HIP_TRY(hipSetDevice(net->cfg.device));
    return ensure_scratch(net, rows);
}

int az_net_forward_static(az_net* net, const void* feat, int32_t rows, float* logits, float* values, void* stream_) {
    if (!net || !feat || !logits || !values) return fail(AZ_ERR_ARG, "null argument");
    if (!net->loaded) return fail(AZ_ERR_STATE, "az_net_forward_static before az_net_load");
    if (net->cfg.precision != AZ_NET_BF16) return fail(AZ_ERR_ARG, "az_net_forward_static: bf16 networks only");
    if (rows <= 0) return AZ_OK;
    return mfma_forward_static(net->mfma, feat, rows, logits, values, (hipStream_t)stream_);
}

int az_net_indirect_passes(const az_net* net, int32_t max_rows) {
    if (!net || !net->loaded || net->cfg.precision != AZ_NET_BF16 || max_rows < 1) return 0;
    return mfma_indirect_passes(net->mfma, max_rows);
}

int az_net_capturable(const az_net* net, int32_t max_rows) {
    if (!net || !net->loaded || net->cfg.precision != AZ_NET_BF16 || max_rows < 1) return 0;
    return mfma_capturable(net->mfma, max_rows) ? 1 : 0;
}

int64_t az_net_static_max_rows(const az_net* net) {
    if (!net || !net->loaded || net->cfg.precision != AZ_NET_BF16) return 0;
    return mfma_static_max_rows(net->mfma);
}

int az_net_forward(az_net* net, const void* feat, int32_t rows, float* logits, float* values, void* stream_) {
    if (!net || !feat || !logits || !values) return fail(AZ_ERR_ARG, "null argument");
    if (!net->loaded) return fail(AZ_ERR_STATE, "az_net_forward before az_net_load");
    if (rows <= 0) return AZ_OK;
    hipStream_t st = (hipStream_t)stream_;
    const az_net_config& c = net->cfg;
    if (c.precision == AZ_NET_BF16) return mfma_forward(net->mfma, feat, rows, nullptr, logits, values, st);
    int rc = ensure_scratch(net, rows);
    if (rc != AZ_OK) return rc;
    const int n = c.board_size, np = n * n, C = c.channels, q = C / 4, A = np + 1, T = 256;
    const long long tot = (long long)rows * C * np;
    k_conv_f32<<<nblk(tot, T), T, 0, st>>>((const float*)feat, net->stem.w, net->stem.b, net->x, rows, c.in_planes, C, n, 3, 1);
    for (int i = 0; i < c.blocks; ++i) {
        F32Block& b = net->blocks[i];
        k_conv_f32<<<nblk(tot, T), T, 0, st>>>(net->x, b.c1.w, b.c1.b, net->h, rows, C, C, n, 3, 1);
        k_conv_f32<<<nblk(tot, T), T, 0, st>>>(net->h, b.c2.w, b.c2.b, net->y, rows, C, C, n, 3, 0);
        if (c.use_se)
            k_se_f32<<<rows, 256, (C + c.se_hidden) * sizeof(float), st>>>(net->y, b.se_w1, b.se_b1, b.se_w2, b.se_b2, net->scale, C,
                                                                          c.se_hidden, np);
        k_residual_f32<<<nblk(tot, T), T, 0, st>>>(net->x, net->y, c.use_se ? net->scale : nullptr, tot, np);
    }
    k_conv_f32<<<nblk((long long)rows * q * np, T), T, 0, st>>>(net->x, net->ph0.w, net->ph0.b, net->p0, rows, C, q, n, 1, 1);
    k_conv_f32<<<nblk((long long)rows * 2 * np, T), T, 0, st>>>(net->p0, net->ph3.w, net->ph3.b, net->p1, rows, q, 2, n, 1, 1);
    k_linear_f32<<<nblk((long long)rows * A, T), T, 0, st>>>(net->p1, net->pfc_w, net->pfc_b, logits, rows, 2 * np, A, 0);
    k_conv_f32<<<nblk((long long)rows * q * np, T), T, 0, st>>>(net->x, net->vh0.w, net->vh0.b, net->v0, rows, C, q, n, 1, 1);
    k_linear_f32<<<nblk((long long)rows * c.value_hidden, T), T, 0, st>>>(net->v0, net->vh4_w, net->vh4_b, net->v1, rows, q * np,
                                                                         c.value_hidden, 1);
    k_linear_f32<<<nblk(rows, T), T, 0, st>>>(net->v1, net->vh6_w, net->vh6_b, values, rows, c.value_hidden, 1, 2);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(AZ_ERR_HIP, std::string("net f32 forward: ") + hipGetErrorString(err));
    return AZ_OK;
}

}  // extern "C"
