// Persistent XCD-team cINN chain (i2v_flow_chain.hip): arguments and launcher.
#pragma once
#include "i2v_common.h"

namespace i2v {

constexpr int FLOW_CHAIN_SYNC_INTS = 32;  // [0] abort flag, [8..15] team leaders' XCC ids, [16] pass epoch (persists)

struct FlowChainArgs {
    float* x;              // [B][64] state, in place
    float* logdet;         // [B] or null (forward only)
    const float* pre;      // [B][pre_stride]: embedding part + bias of every first layer (flow_pre_kernel)
    long pre_stride;       // S * 1024
    const float* W0x;      // [S][8][1024][4]
    const float* Wmid;     // [S][2][1024][512]
    const float* bmid;     // [S][2][1024]
    const float* W3T;      // [S][512][64]
    const float* b3;       // [S][64]
    const float* an_loc;   // [n_flows][64]
    const float* an_scale;
    const float* an_logdet;  // [n_flows] sum log|scale| (device)
    const int* shuf_f;     // [n_flows][64]
    const int* shuf_b;
    float* exch;           // team exchange buffers: flow_chain_exchange_floats() floats
    int* sync;             // FLOW_CHAIN_SYNC_INTS ints; the caller zeroes them (and the exchange buffer) once per workspace
    unsigned long long cond_mask;  // bit `step`: the first layer of that half-step sees only the embedding (mode 'cond')
    int B, n_flows, reverse;
    int use_an, use_act, use_shuf;
};

size_t flow_chain_lds_bytes();
size_t flow_chain_exchange_floats();
// hidden_dim 512, hidden_depth 2, in_channels 64, batch <= 64 (8 teams x 8 samples)
int flow_chain_launch(const FlowChainArgs& a, hipStream_t st);

}  // namespace i2v
