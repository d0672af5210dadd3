// cINN pass as ONE persistent launch: XCD-local teams instead of a 121-launch chain.
//
// The launch chain of i2v_flow.hip pays, per dependent layer, a kernel boundary plus one cold memory round trip, and every
// XCD's L2 re-fetches each layer's whole activation matrix (541 MB of HBM traffic per pass for 189 MB of parameters).  Here
// the grid is 8 teams x 32 workgroups, one workgroup per CU; team = blockIdx.x & 7 (the workgroups an XCD receives), and
// a team owns a contiguous run of <= 8 samples for the WHOLE pass: nothing crosses an XCD except the parameters, which
// every team streams through its own L2 (the 189 MB fit the 256 MB Infinity Cache, so HBM sees them about once).
//
// Per half-step (s- and t-net evaluated together, rows n = net * 512 + j):
//   first layer  h0 = lrelu(pre + W0x . x)        each workgroup: its 32 rows, K = 32, from the state it holds itself
//   hidden 1, 2  h  = lrelu(b + W . h)            its 32 rows x 8 samples, K = 512: lane = 8-wide k slice, wave = 4 rows;
//                                                 the 64 KB weight slice goes straight into registers, one layer ahead
//   last layer                                    K-split: the workgroup multiplies its own 32 hidden units with its 32 x 32
//                                                 slice of W3 and publishes a partial [32 out][8 samples]; everybody sums
//   coupling + ActNorm + InvLeakyRelu + Shuffle   redundantly in every workgroup (wave = sample, lane = channel)
// i.e. three exchanges per half-step (h0, hidden 1, the partials), 120 per pass -- as many as the chain had launches.
//
// Exchange inside a team: DATA-TAGGED GRANULES, no barrier and no flag.  Every published float travels as one naturally
// aligned 8-byte {value, tag} store (plain: the line stays in the XCD's L2), tag = pass epoch * 256 + exchange number; a
// consumer reads the granules it needs with `sc1` 8-byte loads (L2-served, never a stale L1 line) and simply retries until
// every tag matches.  A workgroup cannot run more than one exchange ahead of the slowest member (it needs everybody's
// data), and the three buffers (h0, h1, partials) are each reused only every third exchange, so a granule is never
// overwritten before its readers are done.  (A first version with one L2 counter barrier per exchange -- store, wait for
// the acknowledgement, atomic add, poll, then load -- measured 1016 us per pass, slower than the 844 us launch chain: five
// serialised L2 round trips per exchange.)  Same-XCD placement is what makes the plain stores visible: every workgroup
// compares its HW_REG_XCC_ID with the team leader's and raises the abort flag otherwise, all spins are bounded, and the
// host falls back to the launch chain for good when the flag is set (MI355X_MICROARCH.md: placement is observed, not
// promised).
#include "i2v_flow_chain.h"

namespace i2v {

constexpr int FC_BPT = 8;          // samples per team
constexpr int FC_SLOTS = 32;       // workgroups per team
constexpr int FC_ROWS = 32;        // rows of a layer per workgroup (2 nets x 512 / 32)
constexpr int FC_RED = 68;         // padded lane count of the k-split reduction buffer (conflict-free float4 reads)
constexpr int FC_SPIN_LIMIT = 1 << 22;

struct FcLds {
    float act[512 * FC_BPT];                 // input activations of this workgroup's net, [k][sample]
    float red[8][32][FC_RED];                // per wave: 32 partial values x 64 k-slices
    float parts[FC_SLOTS][256];              // last layer: everybody's partials [slot][out * 8 + sample]
    float w3s[32][32];                       // W3 slice [k][out]
    float w0s[FC_ROWS][32];                  // next first layer, state part [row][k]
    float h2s[FC_ROWS][FC_BPT];              // this workgroup's slice of the last hidden layer
    float xs[FC_BPT][64];                    // state (all samples of the team)
    float stl[64][FC_BPT];                   // s | t of the coupling, [net * 32 + c][sample]
};

typedef unsigned long long granule_t;  // {float value (low dword), int tag (high dword)}

__device__ __forceinline__ granule_t ld_granule(const granule_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dwordx2 sc1: served by L2
}
__device__ __forceinline__ void st_granule(granule_t* p, float v, int tag) {
    *p = (granule_t)(unsigned)__float_as_int(v) | ((granule_t)(unsigned)tag << 32);  // one 8-byte store
}

// Gathers N granules per thread (element tid + 512 u of `src`) with tag `tag` into dst (LDS, floats): retry until all match.
template <int N>
__device__ __forceinline__ void gather_granules(const granule_t* src, float* dst, int tag, int* abort_flag) {
    const int tid = threadIdx.x;
    granule_t g[N];
    int spins = 0;
    while (true) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < N; ++u) g[u] = ld_granule(src + tid + 512 * u);
#pragma unroll
        for (int u = 0; u < N; ++u) ok &= (int)(g[u] >> 32) == tag;
        if (ok) break;
        if (++spins > FC_SPIN_LIMIT) {  // a producer never delivered (not resident?): give up, the host falls back
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if ((spins & 1023) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int u = 0; u < N; ++u) dst[tid + 512 * u] = __int_as_float((int)(unsigned)g[u]);
}

struct W8 { float4 r[4][2]; };  // hidden-layer weights of one thread: 4 rows x 8 consecutive k

__device__ __forceinline__ void load_w8(W8& w, const float* Wl, int row0, int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float* p = Wl + (long)(row0 + j) * 512 + k0;
        w.r[j][0] = *reinterpret_cast<const float4*>(p);
        w.r[j][1] = *reinterpret_cast<const float4*>(p + 4);
    }
}

__global__ __launch_bounds__(512, 2) void flow_chain_kernel(FlowChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    FcLds& L = *reinterpret_cast<FcLds*>(smem_);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bpt = (a.B + 7) >> 3;                 // samples per team (<= FC_BPT)
    const int b_lo = team * bpt;
    const int nb = min(bpt, a.B - b_lo);
    if (nb <= 0) return;                             // (whole team: no barrier is ever entered)
    int* abort_flag = a.sync;
    int* team_xcc = a.sync + 8 + team;
    const int tagbase = a.sync[16] << 8;             // pass epoch (bumped by one thread at the end of every pass)
    int ph = 0;                                      // exchange number inside the pass

    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15;
    if (slot == 0 && tid == 0) __hip_atomic_store(team_xcc, xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    const int net = slot >> 4, r0 = (slot & 15) * FC_ROWS;  // this workgroup's rows: n = net * 512 + r0 + j
    const int nrow0 = net * 512 + r0;
    granule_t* ex_h0 = reinterpret_cast<granule_t*>(a.exch) + (long)team * (3 * 1024 * FC_BPT);  // [1024][8] each
    granule_t* ex_h1 = ex_h0 + 1024 * FC_BPT;
    granule_t* ex_p = ex_h1 + 1024 * FC_BPT;                                                      // [32 slots][256]
    const int nf = a.n_flows, S = 2 * nf;
    auto step_of = [&](int it) {
        const int fl = a.reverse ? nf - 1 - it / 2 : it / 2;
        const int i = a.reverse ? 1 - it % 2 : it % 2;
        return fl * 2 + i;
    };

    // ---- state: wave = sample, lane = channel (registers) + LDS mirror for the first-layer GEMM
    float x = (w < nb) ? a.x[(long)(b_lo + w) * 64 + lane] : 0.f;
    float logdet = 0.f;
    auto elementwise = [&](int shuf_block, int an_block, bool lrelu, bool swap) {
        if (!a.reverse) {
            if (shuf_block >= 0) x = __shfl(x, a.shuf_f[shuf_block * 64 + lane]);
            if (an_block >= 0) {
                x = a.an_scale[an_block * 64 + lane] * (x + a.an_loc[an_block * 64 + lane]);
                logdet += a.an_logdet[an_block];
            }
            if (lrelu) x = x * (x >= 0.f ? 1.0f : 0.9f);
        } else {
            if (lrelu) x = x / (x >= 0.f ? 1.0f : 0.9f);
            if (an_block >= 0) x = x / a.an_scale[an_block * 64 + lane] - a.an_loc[an_block * 64 + lane];
            if (shuf_block >= 0) x = __shfl(x, a.shuf_b[shuf_block * 64 + lane]);
        }
        if (swap) x = __shfl(x, lane ^ 32);
        L.xs[w][lane] = x;
    };
    // first layer of half-step `step` for this workgroup's 32 rows: publishes h0
    auto first_layer = [&](int step, float pre_v, float4 w0v, int tag) {
        const bool cond = (a.cond_mask >> step) & 1;
        if (tid < 256) *reinterpret_cast<float4*>(&L.w0s[tid >> 3][(tid & 7) * 4]) = w0v;
        __syncthreads();  // xs, w0s
        if (tid < 256) {
            const int j = tid >> 3, s = tid & 7;
            float acc = pre_v;
            if (!cond) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 wv = *reinterpret_cast<const float4*>(&L.w0s[j][4 * q]);
                    const float4 xv = *reinterpret_cast<const float4*>(&L.xs[s][4 * q]);
                    acc = fmaf(wv.x, xv.x, acc); acc = fmaf(wv.y, xv.y, acc);
                    acc = fmaf(wv.z, xv.z, acc); acc = fmaf(wv.w, xv.w, acc);
                }
            }
            st_granule(&ex_h0[(nrow0 + j) * FC_BPT + s], acc >= 0.f ? acc : 0.01f * acc, tag);
        }
    };
    // requests (independent of the chain) for the first layer of `step`: its embedding part and its state weights
    auto request_first = [&](int step, float& pre_v, float4& w0v) {
        pre_v = 0.f;
        w0v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 256) {
            const int j = tid >> 3, s = tid & 7, q = tid & 7;
            if (s < nb) pre_v = a.pre[(long)(b_lo + s) * a.pre_stride + (long)step * 1024 + nrow0 + j];
            w0v = *reinterpret_cast<const float4*>(a.W0x + (((long)step * 8 + q) * 1024 + nrow0 + j) * 4);
        }
    };

    if (!a.reverse) elementwise(-1, a.use_an ? 0 : -1, a.use_act != 0, false);
    else elementwise(a.use_shuf ? nf - 1 : -1, -1, false, false);
    W8 wA, wB;
    const int hrow0 = nrow0 + 4 * w, k0 = 8 * lane;
    {
        float pre_v; float4 w0v;
        const int st0 = step_of(0);
        request_first(st0, pre_v, w0v);
        load_w8(wA, a.Wmid + ((long)st0 * 2 + 0) * 1024 * 512, hrow0, k0);
        first_layer(st0, pre_v, w0v, tagbase + (++ph));
    }

    // one hidden layer: activations from `src` (team exchange buffer), weights `wc`; result of lanes < 32: value (j, s)
    // (`prefetch` issues the loads that do not depend on the chain once the activations have ARRIVED: VMEM returns in order,
    // so a weight fetch queued in front of a poll would delay it by its own latency)
    auto hidden = [&](const granule_t* src, int tag, const W8& wc, const float* bias_l, auto&& prefetch) -> float {
        gather_granules<8>(src + (long)net * 512 * FC_BPT, L.act, tag, abort_flag);
        prefetch();
        __syncthreads();
        float acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[j][s] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float4 a0 = *reinterpret_cast<const float4*>(&L.act[(k0 + kk) * FC_BPT]);
            const float4 a1 = *reinterpret_cast<const float4*>(&L.act[(k0 + kk) * FC_BPT + 4]);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 wv = wc.r[j][kk >> 2];
                const float ws = (kk & 3) == 0 ? wv.x : (kk & 3) == 1 ? wv.y : (kk & 3) == 2 ? wv.z : wv.w;
#pragma unroll
                for (int s = 0; s < 8; ++s) acc[j][s] = fmaf(ws, av[s], acc[j][s]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 8; ++s) L.red[w][j * 8 + s][lane] = acc[j][s];
        __syncthreads();
        const int v = lane & 31, hf = lane >> 5;
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 p = *reinterpret_cast<const float4*>(&L.red[w][v][hf * 32 + 4 * q]);
            sum += (p.x + p.y) + (p.z + p.w);
        }
        sum += __shfl_xor(sum, 32);
        sum += bias_l[hrow0 + (v >> 3)];
        return sum >= 0.f ? sum : 0.01f * sum;
    };

    for (int it = 0; it < S; ++it) {
        const int step = step_of(it);
        const int fl = step >> 1, i = step & 1;
        const bool more = it + 1 < S;
        const int nstep = more ? step_of(it + 1) : step;
        // hidden layer 1; behind its activation loads: the requests that do not depend on the chain (next first layer,
        // last-layer slice, the second hidden layer's weights)
        float pre_v = 0.f, b3v = 0.f;
        float4 w0v = make_float4(0.f, 0.f, 0.f, 0.f), w3v = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            const float hv = hidden(ex_h0, tagbase + ph, wA, a.bmid + ((long)step * 2 + 0) * 1024, [&]() {
                request_first(nstep, pre_v, w0v);
                if (tid < 256)
                    w3v = *reinterpret_cast<const float4*>(a.W3T + ((long)step * 512 + r0 + (tid >> 3)) * 64 + net * 32 + (tid & 7) * 4);
                b3v = a.b3[step * 64 + lane];
                load_w8(wB, a.Wmid + ((long)step * 2 + 1) * 1024 * 512, hrow0, k0);
            });
            ++ph;
            if (lane < 32) st_granule(&ex_h1[(hrow0 + (lane >> 3)) * FC_BPT + (lane & 7)], hv, tagbase + ph);
        }
        // hidden layer 2 (kept local) + this workgroup's K-slice of the last layer
        {
            const float hv = hidden(ex_h1, tagbase + ph, wB, a.bmid + ((long)step * 2 + 1) * 1024, [&]() {
                load_w8(wA, a.Wmid + ((long)nstep * 2 + 0) * 1024 * 512, hrow0, k0);  // next half-step's first hidden layer
            });
            if (lane < 32) L.h2s[4 * w + (lane >> 3)][lane & 7] = hv;
            if (tid < 256) *reinterpret_cast<float4*>(&L.w3s[tid >> 3][(tid & 7) * 4]) = w3v;
        }
        __syncthreads();
        if (tid < 256) {
            const int o = tid >> 3, s = tid & 7;
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) p = fmaf(L.h2s[k][s], L.w3s[k][o], p);
            st_granule(&ex_p[slot * 256 + tid], p, tagbase + ph + 1);
        }
        ++ph;
        gather_granules<16>(ex_p, &L.parts[0][0], tagbase + ph, abort_flag);
        __syncthreads();
        {   // s | t = b3 + sum of the 16 partials of the net: thread = (c2 = net' * 32 + out, sample)
            const int c2 = tid >> 3, s = tid & 7, np = c2 >> 5, o = c2 & 31;
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += L.parts[np * 16 + q][o * 8 + s];
            L.stl[c2][s] = sum;
        }
        __syncthreads();
        {   // affine coupling on the second half of the state (flow_blocks.py:91,103); wave = sample, lane = channel
            const float sv = L.stl[lane & 31][w] + __shfl(b3v, lane & 31);
            const float tv = L.stl[32 + (lane & 31)][w] + __shfl(b3v, 32 + (lane & 31));
            if (lane >= 32) x = a.reverse ? (x - tv) * expf(-sv) : fmaf(x, expf(sv), tv);
            if (!a.reverse) {
                float r = lane < 32 ? sv : 0.f;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) r += __shfl_xor(r, off);
                logdet += r;
            }
            int shuf_block = -1, an_block = -1;
            bool lrelu = false, swap = false;
            if (!a.reverse) {
                if (i == 0) swap = true;
                else {
                    if (a.use_shuf) shuf_block = fl;
                    if (fl + 1 < nf) { if (a.use_an) an_block = fl + 1; lrelu = a.use_act != 0; }
                }
            } else {
                if (i == 1) swap = true;
                else {
                    lrelu = a.use_act != 0;
                    if (a.use_an) an_block = fl;
                    if (fl - 1 >= 0 && a.use_shuf) shuf_block = fl - 1;
                }
            }
            elementwise(shuf_block, an_block, lrelu, swap);
        }
        if (more) first_layer(nstep, pre_v, w0v, tagbase + (++ph));
    }
    if (tid == 0) {
        const int lead = __hip_atomic_load(team_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lead != xcc + 1) __hip_atomic_store(abort_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // team spans XCDs
        if (blockIdx.x == 0) __hip_atomic_fetch_add(a.sync + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next pass's epoch
    }
    if (slot == 0 && w < nb) {
        a.x[(long)(b_lo + w) * 64 + lane] = x;
        if (a.logdet && lane == 0) a.logdet[b_lo + w] = logdet;
    }
}

size_t flow_chain_lds_bytes() { return sizeof(FcLds); }
size_t flow_chain_exchange_floats() { return (size_t)8 * 3 * 1024 * FC_BPT * 2; }  // 8-byte granules

int flow_chain_launch(const FlowChainArgs& a, hipStream_t st) {
    I2V_REQUIRE(a.B >= 1 && a.B <= 8 * FC_BPT, I2V_E_INVALID, "flow chain: batch %d", a.B);
    static bool attr[I2V_MAX_DEV] = {};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(flow_chain_kernel), 160 * 1024, attr)) return rc;
    // abort flag and team leaders' XCC ids start from zero; the epoch word (sync[16]) persists across passes
    I2V_HIP_CHECK(hipMemsetAsync(a.sync, 0, 16 * sizeof(int), st));
    hipLaunchKernelGGL(flow_chain_kernel, dim3(8 * FC_SLOTS), dim3(512), sizeof(FcLds), st, a);
    I2V_HIP_CHECK(hipGetLastError());
    return I2V_OK;
}

}  // namespace i2v
