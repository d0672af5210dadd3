// cINN coupling chain on the fp32 matrix cores (i2v_flow_tile.hip): packed-weight container, workspace layout, launcher.
//
// Everything on the chain is a 16 x 16 tile of v_mfma_f32_16x16x4_f32 (exact fp32, = an fmaf chain): 16 output rows of a
// Linear layer x 16 samples.  A tile is stored as the accumulator fragment itself -- 64 lanes x 4 floats = 1 KB contiguous,
// lane l = (q = l >> 4, n = l & 15), register r: row 16*tile + 4*q + r, sample n -- which is ALSO the B-operand fragment of
// the next layer (k = 16*tile + 4*q + j for MFMA step j): a layer's output tile is consumed by the next layer with one
// coalesced 16-byte load per lane and no transposition anywhere on the chain.
#pragma once
#include "i2v_common.h"

namespace i2v {

// caller tensors of the current pass, read through one level of indirection so that a captured graph stays valid when
// the caller passes different tensors (updated by a one-thread kernel only when a pointer changed)
struct FlowIo {
    const float* xin;     // [B][64]
    const float* embed;   // [B][E]
    float* xout;          // [B][64]
    float* logdet_out;    // [B] or null
};

struct FlowTilePack {
    int H = 0, HB = 0, NRT = 0, KE16 = 0, depth = 0, S = 0, E = 0;
    DevBuf WT;   // [S][depth][NRT][HB][256]   hidden layers, A fragments: lane (q, m), j -> W[16 rt + m][16 k16 + 4 q + j]
    DevBuf W3P;  // [S][NRT][2][256]           last layer: lane (q, m), j -> W3[net][c = 16 cb + m][k = 16 rtn + 4 q + j], rt = net*HB + rtn
    DevBuf W0T;  // [S][NRT][2][256]           state part of the first layer (K = 32)
    DevBuf W0E;  // [S][NRT][KE16][256]        embedding part of the first layer (K = E, zero padded to 16 KE16)
    DevBuf io;   // one FlowIo
    FlowIo io_host{};  // what the device copy holds
    bool ok = false;
    // fp16-operand mode (BASELINE configs[4] "fp16 MFMA conditioning GEMM"; i2v_flow_cfg.linear_f16): every Linear of the s- / t-nets
    // runs v_mfma_f32_16x16x16_f16 -- weights rounded to fp16 once at load (the same fragments with 8 instead of 16 bytes per lane:
    // half the streamed parameter bytes), activations rounded to fp16 per layer in registers, fp32 accumulation, bias / LeakyReLU /
    // coupling / log-det in fp32.  One MFMA per 16 x 16 x 16 block instead of four.
    bool f16 = false;
    // A/B switches, read ONCE when the weights are packed (i2v_flow_load), never on a launch path: I2V_FLOW_NS = sample tiles per
    // hidden-layer workgroup (0: by batch), I2V_FLOW_FOLD = 0 | 1 forces the unfolded / folded chain (-1: by batch)
    int force_ns = 0, force_fold = -1;
};

// geometry the tile chain covers (every shipped config: 64 channels, hidden 512, depth 2); anything else runs the generic
// vector-ALU launch chain of i2v_flow.hip
inline bool flow_tile_geometry_ok(int in_channels, int H, int depth, int E) {
    return in_channels == 64 && H >= 128 && H <= 512 && H % 128 == 0 && depth >= 1 && E >= 1 && E <= 128;
}

// host weights in the layouts i2v_flow_load builds: W0 [S][2H][32 + E], Wmid [S][depth][2H][H], W3T [S][H][64]
int flow_tile_pack(FlowTilePack& p, int S, int H, int depth, int E, const float* W0, const float* Wmid, const float* W3T, bool f16 = false);

struct FlowTileWs {
    size_t x, x2, logdet, pre, hA, hB, P, P2, total;   // P / P2: partial products of the final Linear, alternating per half-step
};
FlowTileWs flow_tile_ws(const FlowTilePack& p, int B);

struct FlowTileChain {
    const FlowTilePack* pack;
    const float* b0;     // [S][2H]
    const float* bmid;   // [S][depth][2H]
    const float* b3;     // [S][64]
    const float* an_loc; const float* an_scale;   // [n_flows][64]
    const float* an_logdet_host;                  // [n_flows] (host)
    const int* shuf_f; const int* shuf_b;         // [n_flows][64]
    const int* step_cond;                         // [S] (host): first layer sees only the embedding
    int n_flows, use_an, use_act, use_shuf;
};

// enqueues the whole pass (pre-GEMM, 1 + S tail launches, S * depth hidden launches) on `st`; reads the caller's tensors
// through pack->io (see flow_tile_set_io)
int flow_tile_enqueue(const FlowTileChain& c, bool reverse, char* ws, int B, hipStream_t st);
// makes the device-side FlowIo match (launches a one-thread kernel on `st` only when something changed)
int flow_tile_set_io(FlowTilePack& p, const FlowIo& io, hipStream_t st);

}  // namespace i2v
