// cINN flow on gfx950: ConditionalFlow forward / inverse (reference: stage2_cINN/modules/flow_blocks.py).
//
// The flow is a FLAT vector flow on z in R^{B x 64}: 20 blocks x 2 coupling half-steps, each half-step
// evaluating an s-net and a t-net (4 Linear layers each, hidden 512).  Per pass all 47-50 M parameters are
// read exactly once (189-200 MB fp32): the pass is bound by weight streaming and by the 160-deep chain of
// dependent layers, not by arithmetic (DESIGN.md "cINN pass").
//
// Data layout in HBM
//   x state      [B][64]      one wavefront (64 lanes) owns one sample: lane == channel
//   activations  [2H][B]      row n = net*H + j (net 0 = s, 1 = t), batch contiguous -> lane == sample,
//                             every load/store of the N-split layer kernels is a coalesced 256-B row segment
//   weights      torch layout [N][K] rows (K contiguous): a workgroup's rows are wave-uniform -> scalar loads;
//                the s- and t-net of a half-step are stored back to back so one launch covers both.
//   W3T          [H][64]      last layer transposed: lane == (net, channel), coalesced
//
//   W0x / W0e    state part [S][8][2H][4] and embedding part [R/64][E/4][64][4] of the first layers: every wave load 1 KB
//
// Two implementations of the launch chain, both replayed from one hipGraph per direction:
//   * i2v_flow_tile.hip (default for the shipped geometry: 64 channels, hidden 128..512, depth >= 1): every Linear on
//     16 x 16 tiles of v_mfma_f32_16x16x4_f32, tile-major activations, last layer fused into the last hidden layer;
//   * the generic vector-ALU kernels below (any hidden_dim that is a multiple of 64, depth 0; I2V_FLOW_TILE=0 selects them
//     for A/B measurements).
// Generic kernels (121 launches per pass)
//   flow_pre_kernel    : embedding part of all 80 first layers at once, off the dependent chain, sample-major output
//   flow_hidden_kernel : (i2v_linear.h) hidden Linear + LeakyReLU, s- and t-net in one launch; rows split over
//                        workgroups, K over the waves of a workgroup (LDS reduce), float4 = 4 samples per lane
//   flow_tail_kernel   : one workgroup per sample: last Linear (H -> 32, s and t), affine coupling
//                        x*exp(s)+t / (x-t)*exp(-s) (flow_blocks.py:91,103), log-det = sum_c s by a wavefront
//                        shuffle reduction (:93), then the elementwise ops between two half-steps (Shuffle gather
//                        as a lane permute :152-154, ActNorm modules.py:80/100, InvLeakyRelu :180-187, half swap),
//                        then the NEXT half-step's first Linear (K = the 32 state channels just produced).
//   (flow_linear_kernel in i2v_linear.h serves the stand-alone MLP / Linear entry points.)
#include "i2v_common.h"
#include "i2v_linear.h"
#include "i2v_flow_tile.h"

#include <cmath>
#include <cstdlib>
#include <memory>

namespace i2v {

struct TailArgs {
    const float* h;    // [2H][B] output of the last hidden layer, or null (no coupling in this launch)
    const float* W3T;  // [H][64]
    const float* b3;   // [64]
    float* x;          // [B][64] state, updated in place
    float* logdet;     // [B] or null
    int H, B, Bp;  // Bp: row stride of h
    int reverse;
    // elementwise ops applied after the coupling, before the next half-step's first layer
    const int* shuf;       // [64] gather indices or null
    const float* an_loc;   // [64] or null
    const float* an_scale; // [64]
    float an_logdet;       // sum log|scale| of that ActNorm (forward only)
    int do_lrelu;
    int do_swap;
    // first layer of the NEXT half-step, fused in (its K = 32 inputs are the state channels this launch just produced):
    // h0[n][b] = lrelu_0.01(pre[n][b] + sum_k W0x[n][k] * x[b][k])
    int l1;            // 0: none, 1: K = 32, 2: K = 0 (mode 'cond': the first layer sees only the embedding)
    int N2;            // rows of that layer (2H <= 64 * TAIL_WAVES)
    const float* W0x;  // [N2][32] state part of the first layer's weights
    const float* pre;  // [B][pre_stride] embedding part + bias (transposed pre-GEMM output), offset to this half-step's rows
    long pre_stride;
    float* h0;         // [N2][Bp]
};

constexpr int TAIL_WAVES = 16;

__global__ __launch_bounds__(64 * TAIL_WAVES) void flow_tail_kernel(TailArgs a) {
    __shared__ float part[TAIL_WAVES][64];
    __shared__ __attribute__((aligned(16))) float xs[64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // sample of this workgroup: workgroup i runs on XCD i % 8; give every XCD a CONTIGUOUS run of B/8 samples so that its
    // 4-byte stores into the [row][sample] first-layer output fill whole 32-byte sectors of a row instead of every 8th word
    // (measured: 2.1 MB of partial-line write-backs per launch for 0.26 MB of output with the identity mapping)
    const int b = (a.B & 7) == 0 ? (int)(blockIdx.x & 7) * (a.B >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    // everything needed later is requested before the reduction so the latencies overlap: wave 0's state / per-channel
    // parameters, and every thread's row of the next first layer
    float x = 0.f, b3 = 0.f, anl = 0.f, ans = 1.f;
    int sidx = lane;
    if (w == 0) {
        x = a.x[(long)b * 64 + lane];
        if (a.h) b3 = a.b3[lane];
        if (a.an_loc) { anl = a.an_loc[lane]; ans = a.an_scale[lane]; }
        if (a.shuf) sidx = a.shuf[lane];
    }
    const int n1 = threadIdx.x;
    const bool l1 = a.l1 != 0 && n1 < a.N2;
    float4 w0[8];
    float pre = 0.f;
    {
        // W0x is stored [q = k / 4][n][4]: for a fixed q the 64 lanes of a wave read 1 KB contiguously
        const float* wr = a.W0x + (l1 && a.l1 == 1 ? (long)n1 * 4 : 0);  // (clamped: unconditional loads stay in registers)
#pragma unroll
        for (int q = 0; q < 8; ++q)
            w0[q] = a.l1 ? *reinterpret_cast<const float4*>(wr + (long)q * a.N2 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (l1) pre = a.pre[(long)b * a.pre_stride + n1];
    }
    if (a.h) {
        // last Linear: lane = (net, c); the waves split K = H, all of a wave's loads are in flight together
        const int net = lane >> 5;
        constexpr int KCMAX = LIN_MAXK / TAIL_WAVES;
        const int kc = a.H / TAIL_WAVES;
        const float* hp = a.h + ((long)net * a.H + (long)w * kc) * a.Bp + b;
        const float* wp = a.W3T + (long)w * kc * 64 + lane;
        float hv[KCMAX], wv[KCMAX];
#pragma unroll
        for (int k = 0; k < KCMAX; ++k) {
            hv[k] = k < kc ? hp[(long)k * a.Bp] : 0.f;
            wv[k] = k < kc ? wp[k * 64] : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KCMAX; ++k) acc = fmaf(hv[k], wv[k], acc);
        part[w][lane] = acc;
    }
    __syncthreads();
    if (w == 0) {
        if (a.h) {
            float st = b3;
#pragma unroll
            for (int q = 0; q < TAIL_WAVES; ++q) st += part[q][lane];
            // lanes 0..31 hold s[c], lanes 32..63 hold t[c]; the transformed half is x[32..63]
            const float s = __shfl(st, lane & 31);
            if (lane >= 32) x = a.reverse ? (x - st) * expf(-s) : fmaf(x, expf(s), st);
            if (a.logdet && !a.reverse) {
                float r = lane < 32 ? st : 0.f;  // log-det of the coupling: sum over the 32 channels of s
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) r += __shfl_xor(r, off);
                if (lane == 0) a.logdet[b] += r;
            }
        }
        if (!a.reverse) {
            if (a.shuf) x = __shfl(x, sidx);                    // Shuffle.forward: x[:, idx]
            if (a.an_loc) {                                     // ActNorm.forward: scale * (x + loc)
                x = ans * (x + anl);
                if (a.logdet && lane == 0) a.logdet[b] += a.an_logdet;
            }
            if (a.do_lrelu) x = x * (x >= 0.f ? 1.0f : 0.9f);   // InvLeakyRelu.forward (log-det reported as 0, quirk Q2)
        } else {
            if (a.do_lrelu) x = x / (x >= 0.f ? 1.0f : 0.9f);   // InvLeakyRelu.reverse
            if (a.an_loc) x = x / ans - anl;                    // ActNorm.reverse
            if (a.shuf) x = __shfl(x, sidx);                    // Shuffle.reverse: x[:, argsort(idx)]
        }
        if (a.do_swap) x = __shfl(x, lane ^ 32);                // chunk / cat[::-1], flow_blocks.py:86,99
        a.x[(long)b * 64 + lane] = x;
        xs[lane] = x;
    }
    if (!a.l1) return;  // (uniform)
    __syncthreads();
    if (l1) {
        // next half-step's first Linear (modules.py:17: nn.LeakyReLU() slope 0.01): thread = output row, K = the 32 state
        // channels just written (broadcast LDS reads); one launch of the chain saved per half-step
        float acc = pre;
        if (a.l1 == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 xv = *reinterpret_cast<const float4*>(&xs[4 * q]);
                acc = fmaf(w0[q].x, xv.x, acc); acc = fmaf(w0[q].y, xv.y, acc);
                acc = fmaf(w0[q].z, xv.z, acc); acc = fmaf(w0[q].w, xv.w, acc);
            }
        }
        a.h0[(long)n1 * a.Bp + b] = acc >= 0.f ? acc : 0.01f * acc;
    }
}

// Embedding part of ALL first layers of a pass, off the dependent chain: preT[b][r] = b0[r] + sum_k W0[r][32 + k] * embed[b][k]
// for the R = S * 2H rows, written sample-major so that the sample-per-workgroup tail kernel reads its rows contiguously.
// lane = row: a wavefront keeps 64 weight rows in registers (W0e is stored [r / 64][k / 4][r % 64][4]: every load of a
// wave is 1 KB contiguous) and sweeps the samples, whose embeddings are broadcast from LDS; stores are coalesced.
constexpr int PRE_EMAX = 128;  // embedding_dim <= 128 (64 BAIR, 128 others, 94 endpoint control)
constexpr int PRE_SC = 16;     // samples per workgroup (blockIdx.y): enough waves to fill the chip at B = 64
__global__ __launch_bounds__(256) void flow_pre_kernel(const float* __restrict__ W0e, const float* __restrict__ b0,
                                                       const float* __restrict__ embed, float* __restrict__ preT, int R, int E,
                                                       int Epad, int B) {
    __shared__ __attribute__((aligned(16))) float es[PRE_SC * PRE_EMAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int blk = blockIdx.x * 4 + w;  // 64-row block of this wave
    const int r = blk * 64 + lane;
    const bool rok = r < R;
    const int nq = Epad >> 2;
    const int bb = blockIdx.y * PRE_SC, nb = min(PRE_SC, B - bb);
    float4 wv[PRE_EMAX / 4];
#pragma unroll
    for (int q = 0; q < PRE_EMAX / 4; ++q)
        wv[q] = (q < nq && blk * 64 < R) ? *reinterpret_cast<const float4*>(W0e + (((long)blk * nq + q) * 64 + lane) * 4)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bias = rok ? b0[r] : 0.f;
    for (int i = threadIdx.x; i < PRE_SC * Epad; i += 256) {
        const int b = i / Epad, k = i - b * Epad;
        es[i] = (b < nb && k < E) ? embed[(long)(bb + b) * E + k] : 0.f;
    }
    __syncthreads();
    for (int b = 0; b < PRE_SC; b += 4) {  // four samples at a time: independent accumulation chains
        float a0 = bias, a1 = bias, a2 = bias, a3 = bias;
        const float* e0 = &es[b * Epad];
#pragma unroll
        for (int q = 0; q < PRE_EMAX / 4; ++q) {
            if (q < nq) {  // uniform
                const float4 x0 = *reinterpret_cast<const float4*>(e0 + 4 * q);
                const float4 x1 = *reinterpret_cast<const float4*>(e0 + Epad + 4 * q);
                const float4 x2 = *reinterpret_cast<const float4*>(e0 + 2 * Epad + 4 * q);
                const float4 x3 = *reinterpret_cast<const float4*>(e0 + 3 * Epad + 4 * q);
                a0 = fmaf(wv[q].x, x0.x, a0); a1 = fmaf(wv[q].x, x1.x, a1); a2 = fmaf(wv[q].x, x2.x, a2); a3 = fmaf(wv[q].x, x3.x, a3);
                a0 = fmaf(wv[q].y, x0.y, a0); a1 = fmaf(wv[q].y, x1.y, a1); a2 = fmaf(wv[q].y, x2.y, a2); a3 = fmaf(wv[q].y, x3.y, a3);
                a0 = fmaf(wv[q].z, x0.z, a0); a1 = fmaf(wv[q].z, x1.z, a1); a2 = fmaf(wv[q].z, x2.z, a2); a3 = fmaf(wv[q].z, x3.z, a3);
                a0 = fmaf(wv[q].w, x0.w, a0); a1 = fmaf(wv[q].w, x1.w, a1); a2 = fmaf(wv[q].w, x2.w, a2); a3 = fmaf(wv[q].w, x3.w, a3);
            }
        }
        if (rok) {
            if (b + 0 < nb) preT[(long)(bb + b + 0) * R + r] = a0;
            if (b + 1 < nb) preT[(long)(bb + b + 1) * R + r] = a1;
            if (b + 2 < nb) preT[(long)(bb + b + 2) * R + r] = a2;
            if (b + 3 < nb) preT[(long)(bb + b + 3) * R + r] = a3;
        }
    }
}

// ingest / egress between caller tensors and the workspace-resident state (a hipMemcpyAsync costs ~12 us each here)
__global__ void flow_copy2_kernel(const float* __restrict__ a, float* __restrict__ da, int na, const float* __restrict__ b,
                                  float* __restrict__ db, int nb) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += gridDim.x * blockDim.x) {
        if (i < na) da[i] = a[i];
        else db[i - na] = b[i - na];
    }
}

}  // namespace i2v

using namespace i2v;

struct i2v_flow {
    i2v_flow_cfg cfg;
    int device = 0;
    bool loaded = false;
    int S = 0;      // half-steps = 2 * n_flows
    int H = 0, E = 0, ld0 = 0, depth = 0;
    DevBuf W0, W0x, W0e, b0, Wmid, bmid, W3T, b3, an_loc, an_scale, shuf_f, shuf_b;  // W0x: [S][8][2H][4] state part of W0
    FlowTilePack tile;               // matrix-core tile chain (i2v_flow_tile.hip); tile.ok: packed and selected
    bool tile_wanted = true;         // env I2V_FLOW_TILE=0: generic vector-ALU chain
    std::vector<float> an_logdet;
    std::vector<int> step_cond;  // 1: first layer sees only the embedding (mode 'cond')
    size_t param_bytes = 0;
    // graph cache: one instantiated chain per direction (motion transfer alternates forward and inverse passes)
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    int g_B[2] = {-1, -1};
    void* g_ws[2] = {nullptr, nullptr};
    // One handle = one FlowIo block + (normally) one workspace: passes on a handle are serialised.  A call that arrives on
    // another stream than the previous one (LatentPrefetcher's side stream next to the main stream) first waits for the
    // previous pass (event recorded behind every pass), so that its set_io kernel / state buffers cannot overtake it.
    StreamOrder order;   // (capture-aware: see i2v_common.h)
    void drop_graphs() {
        for (auto& g : gexec) {
            if (g) (void)hipGraphExecDestroy(g);
            g = nullptr;
        }
    }

    ~i2v_flow() {
        drop_graphs();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
    }
};

namespace {

struct WsLayout {
    size_t x, embed, logdet, preT, hA, hB, total;
};

WsLayout ws_layout(const i2v_flow* f, int B) {
    WsLayout L;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n, 256); return r; };
    L.x = take((size_t)B * 64 * 4);
    L.embed = take((size_t)B * f->E * 4);
    L.logdet = take((size_t)B * 4);
    L.preT = take((size_t)f->S * 2 * f->H * B * 4);
    const size_t Bp = (size_t)(B + 63) / 64 * 64;  // hidden activations are [2H][Bp]
    L.hA = take((size_t)2 * f->H * Bp * 4);
    L.hB = take((size_t)2 * f->H * Bp * 4);
    L.total = o;
    if (f->tile.ok) L.total = std::max(L.total, flow_tile_ws(f->tile, B).total);  // either chain may use the buffer
    return L;
}

// The launch chain of one pass on `st`, reading/writing only workspace memory.
int enqueue_chain(i2v_flow* f, bool reverse, char* ws, int B, hipStream_t st) {
    const WsLayout L = ws_layout(f, B);
    float* x = reinterpret_cast<float*>(ws + L.x);
    const float* embed = reinterpret_cast<const float*>(ws + L.embed);
    float* logdet = reinterpret_cast<float*>(ws + L.logdet);
    float* preT = reinterpret_cast<float*>(ws + L.preT);
    float* hA = reinterpret_cast<float*>(ws + L.hA);
    float* hB = reinterpret_cast<float*>(ws + L.hB);
    const int H = f->H, N2 = 2 * f->H, S = f->S;
    const int Bp = (B + 63) / 64 * 64;
    const bool act = f->cfg.activation != 0, an = !f->cfg.skip_actnorm, sh = !f->cfg.skip_shuffle;

    if (!reverse) I2V_HIP_CHECK(hipMemsetAsync(logdet, 0, (size_t)B * 4, st));
    {   // embedding part of every first layer, all half-steps at once: preT[b][s][n] = b0 + W0[:, 32:] . embed
        const int R = S * N2, Epad = (f->E + 3) / 4 * 4;
        hipLaunchKernelGGL(flow_pre_kernel, dim3((R + 255) / 256, (B + PRE_SC - 1) / PRE_SC), dim3(256), 0, st, f->W0e.as<float>(), f->b0.as<float>(), embed,
                           preT, R, f->E, Epad, B);
        I2V_HIP_CHECK(hipGetLastError());
    }
    // next_step: half-step whose first layer is evaluated at the end of this launch (-1: none)
    auto tail = [&](const float* h, int step, int shuf_block, int an_block, bool lrelu, bool swap, int next_step) -> int {
        TailArgs t{};
        if (next_step >= 0) {
            t.l1 = f->step_cond[next_step] ? 2 : 1;
            t.N2 = N2;
            t.W0x = f->W0x.as<float>() + (size_t)next_step * N2 * 32;
            t.pre = preT + (size_t)next_step * N2;
            t.pre_stride = (long)S * N2;
            t.h0 = hA;  // (may alias h: a workgroup only touches column b of either, reads before it writes)
        }
        t.h = h;
        t.W3T = h ? f->W3T.as<float>() + (size_t)step * H * 64 : nullptr;
        t.b3 = h ? f->b3.as<float>() + (size_t)step * 64 : nullptr;
        t.x = x;
        t.logdet = reverse ? nullptr : logdet;
        t.H = H;
        t.B = B;
        t.Bp = Bp;
        t.reverse = reverse ? 1 : 0;
        t.shuf = shuf_block >= 0 ? (reverse ? f->shuf_b.as<int>() : f->shuf_f.as<int>()) + shuf_block * 64 : nullptr;
        t.an_loc = an_block >= 0 ? f->an_loc.as<float>() + an_block * 64 : nullptr;
        t.an_scale = an_block >= 0 ? f->an_scale.as<float>() + an_block * 64 : nullptr;
        t.an_logdet = an_block >= 0 ? f->an_logdet[an_block] : 0.f;
        t.do_lrelu = lrelu ? 1 : 0;
        t.do_swap = swap ? 1 : 0;
        hipLaunchKernelGGL(flow_tail_kernel, dim3(B), dim3(64 * TAIL_WAVES), 0, st, t);
        I2V_HIP_CHECK(hipGetLastError());
        return I2V_OK;
    };
    const int nf = f->cfg.n_flows;
    // ops in front of the first half-step
    int rc;
    auto step_of = [&](int it) {  // forward visits (fl, i) = (0,0),(0,1),(1,0)...; reverse visits (nf-1,1),(nf-1,0),(nf-2,1)...
        const int fl = reverse ? nf - 1 - it / 2 : it / 2;
        const int i = reverse ? 1 - it % 2 : it % 2;
        return fl * 2 + i;
    };
    if (!reverse) rc = tail(nullptr, 0, -1, an ? 0 : -1, act, false, step_of(0));
    else rc = tail(nullptr, 0, sh ? nf - 1 : -1, -1, false, false, step_of(0));
    if (rc) return rc;

    for (int it = 0; it < S; ++it) {
        // forward visits (fl, i) = (0,0),(0,1),(1,0)...; reverse visits (nf-1,1),(nf-1,0),(nf-2,1)...
        const int fl = reverse ? nf - 1 - it / 2 : it / 2;
        const int i = reverse ? 1 - it % 2 : it % 2;
        const int step = fl * 2 + i;
        // layer 0 (K = 32 state channels + the precomputed embedding part as a per-(n,b) bias) was evaluated into hA by the
        // previous launch of the chain (flow_tail_kernel)
        float* cur = hA;
        float* nxt = hB;
        for (int d = 0; d < f->depth; ++d) {
            HidArgs m{};
            m.W = f->Wmid.as<float>() + ((size_t)step * f->depth + d) * N2 * H;
            m.ldw = H;
            m.K = H;
            m.in = cur;
            m.in_group_stride = (long)H * Bp;
            m.group_rows = H;
            m.bias = f->bmid.as<float>() + ((size_t)step * f->depth + d) * N2;
            m.out = nxt;
            m.N = N2;
            m.Bp = Bp;
            m.slope = 0.01f;
            if ((rc = launch_hidden(m, st))) return rc;
            std::swap(cur, nxt);
        }
        // last layer + coupling + the ops up to the next half-step's first layer
        int shuf_block = -1, an_block = -1;
        bool lrelu = false, swap = false;
        if (!reverse) {
            if (i == 0) swap = true;  // before half-step 1: cat(chunk[::-1])
            else {
                if (sh) shuf_block = fl;
                if (fl + 1 < nf) { if (an) an_block = fl + 1; lrelu = act; }
            }
        } else {
            if (i == 1) swap = true;  // before half-step 0 (flow_blocks.py:98-99)
            else {
                lrelu = act;
                if (an) an_block = fl;
                if (fl - 1 >= 0 && sh) shuf_block = fl - 1;
            }
        }
        if ((rc = tail(cur, step, shuf_block, an_block, lrelu, swap, it + 1 < S ? step_of(it + 1) : -1))) return rc;
    }
    return I2V_OK;
}

// Matrix-core tile chain: the caller's tensors are read / written by the chain's own first / last launches through the
// handle's FlowIo block, so the replayed graph needs no copy kernels around it.
int run_pass_tile(i2v_flow* f, bool reverse, const float* xin, const float* embed, float* xout, float* logdet, char* ws, int B,
                  hipStream_t st) {
    FlowTileChain c{};
    c.pack = &f->tile;
    c.b0 = f->b0.as<float>(); c.bmid = f->bmid.as<float>(); c.b3 = f->b3.as<float>();
    c.an_loc = f->an_loc.as<float>(); c.an_scale = f->an_scale.as<float>(); c.an_logdet_host = f->an_logdet.data();
    c.shuf_f = f->shuf_f.as<int>(); c.shuf_b = f->shuf_b.as<int>(); c.step_cond = f->step_cond.data();
    c.n_flows = f->cfg.n_flows;
    c.use_an = !f->cfg.skip_actnorm; c.use_act = f->cfg.activation != 0; c.use_shuf = !f->cfg.skip_shuffle;
    int rc = flow_tile_set_io(f->tile, FlowIo{xin, embed, xout, reverse ? nullptr : logdet}, st);
    if (rc) return rc;
    if (!f->cfg.use_graph) return flow_tile_enqueue(c, reverse, ws, B, st);
    const int d = reverse ? 1 : 0;
    if (!(f->gexec[d] && f->g_B[d] == B && f->g_ws[d] == ws)) {
        if (f->gexec[d]) { (void)hipGraphExecDestroy(f->gexec[d]); f->gexec[d] = nullptr; }
        if (!f->cap_stream) I2V_HIP_CHECK(hipStreamCreateWithFlags(&f->cap_stream, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        I2V_HIP_CHECK(hipStreamBeginCapture(f->cap_stream, hipStreamCaptureModeThreadLocal));
        rc = flow_tile_enqueue(c, reverse, ws, B, f->cap_stream);
        hipError_t e = hipStreamEndCapture(f->cap_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        I2V_HIP_CHECK(e);
        e = hipGraphInstantiate(&f->gexec[d], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        I2V_HIP_CHECK(e);
        f->g_B[d] = B;
        f->g_ws[d] = ws;
    }
    I2V_HIP_CHECK(hipGraphLaunch(f->gexec[d], st));
    return I2V_OK;
}

int run_pass(i2v_flow* f, bool reverse, const float* xin, const float* embed, float* xout, float* logdet,
             void* workspace, size_t workspace_bytes, int B, hipStream_t st) {
    I2V_REQUIRE(f && f->loaded, I2V_E_STATE, "i2v_flow: weights not loaded");
    I2V_REQUIRE(B > 0 && xin && embed && xout && workspace, I2V_E_INVALID, "i2v_flow: null argument or batch <= 0");
    const WsLayout L = ws_layout(f, B);
    I2V_REQUIRE(workspace_bytes >= L.total, I2V_E_WORKSPACE, "i2v_flow: workspace %zu < required %zu",
                workspace_bytes, L.total);
    char* ws = static_cast<char*>(workspace);
    if (int rco = f->order.entry(st)) return rco;
    StreamOrderMark mark{&f->order, st};   // records the end of this pass on its stream (also on the error paths)
    if (f->tile.ok) return run_pass_tile(f, reverse, xin, embed, xout, logdet, ws, B, st);
    {
        const int na = B * 64, nb = B * f->E;
        hipLaunchKernelGGL(flow_copy2_kernel, dim3((na + nb + 255) / 256), dim3(256), 0, st, xin,
                           reinterpret_cast<float*>(ws + L.x), na, embed, reinterpret_cast<float*>(ws + L.embed), nb);
        I2V_HIP_CHECK(hipGetLastError());
    }
    if (f->cfg.use_graph) {
        const int d = reverse ? 1 : 0;
        if (!(f->gexec[d] && f->g_B[d] == B && f->g_ws[d] == workspace)) {
            if (f->gexec[d]) { (void)hipGraphExecDestroy(f->gexec[d]); f->gexec[d] = nullptr; }
            if (!f->cap_stream) I2V_HIP_CHECK(hipStreamCreateWithFlags(&f->cap_stream, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            I2V_HIP_CHECK(hipStreamBeginCapture(f->cap_stream, hipStreamCaptureModeThreadLocal));
            int rc = enqueue_chain(f, reverse, ws, B, f->cap_stream);
            hipError_t e = hipStreamEndCapture(f->cap_stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            I2V_HIP_CHECK(e);
            e = hipGraphInstantiate(&f->gexec[d], graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            I2V_HIP_CHECK(e);
            f->g_B[d] = B;
            f->g_ws[d] = workspace;
        }
        I2V_HIP_CHECK(hipGraphLaunch(f->gexec[d], st));
    } else {
        int rc = enqueue_chain(f, reverse, ws, B, st);
        if (rc) return rc;
    }
    {
        const int na = B * 64, nb = logdet ? B : 0;
        hipLaunchKernelGGL(flow_copy2_kernel, dim3((na + nb + 255) / 256), dim3(256), 0, st,
                           reinterpret_cast<const float*>(ws + L.x), xout, na, reinterpret_cast<const float*>(ws + L.logdet),
                           logdet, nb);
        I2V_HIP_CHECK(hipGetLastError());
    }
    return I2V_OK;
}

}  // namespace

extern "C" {

int i2v_flow_create(const i2v_flow_cfg* cfg, i2v_flow** out) {
    I2V_REQUIRE(cfg && out, I2V_E_INVALID, "i2v_flow_create: null argument");
    I2V_REQUIRE(cfg->in_channels == 64, I2V_E_INVALID,
                "i2v_flow_create: in_channels must be 64 (lane <-> channel mapping), got %d", cfg->in_channels);
    I2V_REQUIRE(cfg->hidden_dim >= 64 && cfg->hidden_dim % 64 == 0 && cfg->hidden_dim <= LIN_MAXK, I2V_E_INVALID,
                "i2v_flow_create: hidden_dim must be a multiple of 64 in [64, %d], got %d", LIN_MAXK, cfg->hidden_dim);
    I2V_REQUIRE(cfg->embedding_dim > 0 && cfg->embedding_dim <= PRE_EMAX && cfg->hidden_depth >= 0 && cfg->n_flows > 0, I2V_E_INVALID,
                "i2v_flow_create: bad embedding_dim/hidden_depth/n_flows");
    int ndev = 0;
    I2V_HIP_CHECK(hipGetDeviceCount(&ndev));
    I2V_REQUIRE(ndev > 0, I2V_E_HIP, "i2v_flow_create: no HIP device");
    auto f = std::make_unique<i2v_flow>();
    f->cfg = *cfg;
    I2V_HIP_CHECK(hipGetDevice(&f->device));
    f->S = 2 * cfg->n_flows;
    f->H = cfg->hidden_dim;
    f->E = cfg->embedding_dim;
    f->ld0 = 32 + cfg->embedding_dim;
    f->depth = cfg->hidden_depth;
    if (const char* e = std::getenv("I2V_FLOW_TILE")) f->tile_wanted = std::atoi(e) != 0;  // 0: generic vector-ALU chain
    *out = f.release();
    return I2V_OK;
}

void i2v_flow_destroy(i2v_flow* f) { delete f; }

int i2v_flow_load(i2v_flow* f, const i2v_tensor* tensors, int32_t n_tensors) {
    if (f) I2V_REQUIRE_DEVICE(f->device, "i2v_flow_load");
    I2V_REQUIRE(f && tensors && n_tensors > 0, I2V_E_INVALID, "i2v_flow_load: null argument");
    StateDict sd(tensors, n_tensors);
    const int nf = f->cfg.n_flows, H = f->H, E = f->E, ld0 = f->ld0, D = f->depth, S = f->S, N2 = 2 * H;
    std::vector<float> W0((size_t)S * N2 * ld0, 0.f), b0((size_t)S * N2), Wmid((size_t)S * D * N2 * H),
        bmid((size_t)S * D * N2), W3T((size_t)S * H * 64), b3((size_t)S * 64), loc((size_t)nf * 64, 0.f),
        scale((size_t)nf * 64, 1.f);
    std::vector<int> sf((size_t)nf * 64), sb((size_t)nf * 64);
    f->an_logdet.assign(nf, 0.f);
    f->step_cond.assign(S, 0);
    size_t pbytes = 0, wfloats = 0;
    for (int fl = 0; fl < nf; ++fl) {
        const std::string p = "sub_layers." + std::to_string(fl) + ".";
        const bool cond = f->cfg.control == 2 || (f->cfg.control == 1 && fl % 4 != 0);  // flow_blocks.py:24
        const int dim = cond ? E : 32 + E;
        if (!f->cfg.skip_actnorm) {
            const float* l = sd.f32(p + "norm_layer.loc", 64);
            const float* s = sd.f32(p + "norm_layer.scale", 64);
            if (!l || !s) return I2V_E_MISSING;
            double ld = 0.0;
            for (int c = 0; c < 64; ++c) {
                loc[fl * 64 + c] = l[c];
                scale[fl * 64 + c] = s[c];
                ld += std::log(std::fabs((double)s[c]));  // modules.py:86-87 with H = W = 1
            }
            f->an_logdet[fl] = (float)ld;
            pbytes += 2 * 64 * 4;
        }
        if (!f->cfg.skip_shuffle) {
            const int64_t* a = sd.i64(p + "shuffle.forward_shuffle_idx", 64);
            const int64_t* b = sd.i64(p + "shuffle.backward_shuffle_idx", 64);
            if (!a || !b) return I2V_E_MISSING;
            for (int c = 0; c < 64; ++c) {
                I2V_REQUIRE(a[c] >= 0 && a[c] < 64 && b[c] >= 0 && b[c] < 64, I2V_E_INVALID,
                            "i2v_flow_load: shuffle index out of range in block %d", fl);
                sf[fl * 64 + c] = (int)a[c];
                sb[fl * 64 + c] = (int)b[c];
            }
            pbytes += 2 * 64 * 8;
        }
        for (int i = 0; i < 2; ++i) {
            const int step = fl * 2 + i;
            f->step_cond[step] = cond ? 1 : 0;
            for (int net = 0; net < 2; ++net) {
                const std::string q = p + "coupling." + (net == 0 ? "s." : "t.") + std::to_string(i) + ".main.";
                const float* w = sd.f32(q + "0.weight", (int64_t)H * dim);
                const float* bb = sd.f32(q + "0.bias", H);
                if (!w || !bb) return I2V_E_MISSING;
                for (int n = 0; n < H; ++n) {
                    float* dst = &W0[((size_t)step * N2 + (size_t)net * H + n) * ld0];
                    if (cond) std::memcpy(dst + 32, w + (size_t)n * dim, (size_t)E * 4);
                    else std::memcpy(dst, w + (size_t)n * dim, (size_t)dim * 4);
                    b0[(size_t)step * N2 + net * H + n] = bb[n];
                }
                for (int d = 0; d < D; ++d) {
                    const std::string li = std::to_string(2 * (d + 1));
                    const float* wm = sd.f32(q + li + ".weight", (int64_t)H * H);
                    const float* bm = sd.f32(q + li + ".bias", H);
                    if (!wm || !bm) return I2V_E_MISSING;
                    std::memcpy(&Wmid[(((size_t)step * D + d) * N2 + (size_t)net * H) * H], wm, (size_t)H * H * 4);
                    std::memcpy(&bmid[((size_t)step * D + d) * N2 + (size_t)net * H], bm, (size_t)H * 4);
                }
                const std::string ll = std::to_string(2 * (D + 1));
                const float* w3 = sd.f32(q + ll + ".weight", (int64_t)32 * H);
                const float* bb3 = sd.f32(q + ll + ".bias", 32);
                if (!w3 || !bb3) return I2V_E_MISSING;
                for (int c = 0; c < 32; ++c) {
                    for (int k = 0; k < H; ++k) W3T[((size_t)step * H + k) * 64 + net * 32 + c] = w3[(size_t)c * H + k];
                    b3[(size_t)step * 64 + net * 32 + c] = bb3[c];
                }
                pbytes += ((size_t)H * dim + H + (size_t)D * ((size_t)H * H + H) + (size_t)32 * H + 32) * 4;
                wfloats += (size_t)H * dim + (size_t)D * H * H + (size_t)32 * H;   // the weight matrices alone
            }
        }
    }
    std::vector<float> W0x((size_t)S * N2 * 32);
    for (int st = 0; st < S; ++st)
        for (int n = 0; n < N2; ++n)
            for (int k = 0; k < 32; ++k)
                W0x[(((size_t)st * 8 + k / 4) * N2 + n) * 4 + k % 4] = W0[((size_t)st * N2 + n) * ld0 + k];
    // embedding part of W0 for flow_pre_kernel: [r / 64][k / 4][r % 64][4], k zero-padded to a multiple of 4
    const int Epad = (E + 3) / 4 * 4;
    const size_t R = (size_t)S * N2, Rblk = (R + 63) / 64;
    std::vector<float> W0e(Rblk * Epad * 64, 0.f);
    for (size_t r = 0; r < R; ++r)
        for (int k = 0; k < E; ++k) W0e[(((r / 64) * (Epad / 4) + k / 4) * 64 + r % 64) * 4 + k % 4] = W0[r * ld0 + 32 + k];
    int rc;
    if ((rc = f->W0.upload(W0.data(), W0.size() * 4))) return rc;
    if ((rc = f->W0x.upload(W0x.data(), W0x.size() * 4))) return rc;
    if ((rc = f->W0e.upload(W0e.data(), W0e.size() * 4))) return rc;
    if ((rc = f->b0.upload(b0.data(), b0.size() * 4))) return rc;
    if ((rc = f->Wmid.upload(Wmid.data(), Wmid.size() * 4))) return rc;
    if ((rc = f->bmid.upload(bmid.data(), bmid.size() * 4))) return rc;
    if ((rc = f->W3T.upload(W3T.data(), W3T.size() * 4))) return rc;
    if ((rc = f->b3.upload(b3.data(), b3.size() * 4))) return rc;
    if ((rc = f->an_loc.upload(loc.data(), loc.size() * 4))) return rc;
    if ((rc = f->an_scale.upload(scale.data(), scale.size() * 4))) return rc;
    if ((rc = f->shuf_f.upload(sf.data(), sf.size() * 4))) return rc;
    if ((rc = f->shuf_b.upload(sb.data(), sb.size() * 4))) return rc;
    f->tile.ok = false;
    I2V_REQUIRE(!f->cfg.linear_f16 || (f->tile_wanted && flow_tile_geometry_ok(f->cfg.in_channels, H, D, E)), I2V_E_INVALID,
                "i2v_flow_load: linear_f16 needs the matrix-core tile chain (64 channels, hidden 128..512 in steps of 128, depth >= 1, E <= 128)");
    if (f->tile_wanted && flow_tile_geometry_ok(f->cfg.in_channels, H, D, E)) {
        if ((rc = flow_tile_pack(f->tile, S, H, D, E, W0.data(), Wmid.data(), W3T.data(), f->cfg.linear_f16 != 0))) return rc;
    }
    if (f->cfg.linear_f16) {   // the weight matrices stream as fp16 (biases, ActNorm, Shuffle stay 4-byte)
        pbytes -= wfloats * 2;
    }
    f->param_bytes = pbytes;
    f->loaded = true;
    f->drop_graphs();
    return I2V_OK;
}

size_t i2v_flow_workspace_bytes(const i2v_flow* f, int32_t batch) {
    if (!f || batch <= 0) return 0;
    return ws_layout(f, batch).total;
}

size_t i2v_flow_param_bytes(const i2v_flow* f) { return f ? f->param_bytes : 0; }

int i2v_flow_forward(i2v_flow* f, const float* x, const float* embed, float* zt, float* logdet, void* workspace,
                     size_t workspace_bytes, int32_t batch, void* stream) {
    I2V_REQUIRE(f && logdet, I2V_E_INVALID, "i2v_flow_forward: null argument");
    I2V_REQUIRE_DEVICE(f->device, "i2v_flow_forward");
    return run_pass(f, false, x, embed, zt, logdet, workspace, workspace_bytes, batch, static_cast<hipStream_t>(stream));
}

int i2v_flow_inverse(i2v_flow* f, const float* residual, const float* embed, float* z, void* workspace,
                     size_t workspace_bytes, int32_t batch, void* stream) {
    I2V_REQUIRE(f, I2V_E_INVALID, "i2v_flow_inverse: null handle");
    I2V_REQUIRE_DEVICE(f->device, "i2v_flow_inverse");
    return run_pass(f, true, residual, embed, z, nullptr, workspace, workspace_bytes, batch,
                    static_cast<hipStream_t>(stream));
}

}  // extern "C"
