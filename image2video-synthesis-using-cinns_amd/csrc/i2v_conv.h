// Implicit-GEMM convolution on gfx950 matrix cores (shared declarations).
#pragma once
#include "i2v_common.h"

namespace i2v {

constexpr int CONV_BM = 128;   // output positions per workgroup
constexpr int CONV_KC = 16;    // input channels per K chunk
constexpr int CONV_LDS_STRIDE = 20;  // floats per staged row of 16 channels (+4 pad: conflict-free ds_read_b128)

enum ConvEpilogue : int {
    EPI_NONE = 0,
    EPI_LRELU = 1,   // leaky_relu(0.2) on the result (decoder.py:51-52,117)
    EPI_FRAMES = 2,  // tanh + store as [B][T][3][H][W] (decoder.py:118-120)
    EPI_HL16 = 4,    // store in the split-fp16 operand format (input of i2v_conv16.hip) instead of fp32
};

// Weights packed for the kernel: [tap][chunk][CoutPad][16] floats (zero padded).
struct ConvWeights {
    DevBuf w;
    DevBuf bias;      // [Cout] (empty = no bias)
    int Cin = 0, Cout = 0, CoutPad = 0, nchunk = 0;
    int KT = 1, KH = 1, KW = 1;
    // w_src: torch layout [Cout][Cin][KT][KH][KW]; scale multiplies every weight (1/sigma); cin_pad_to: the
    // activation tensor's channel count when it is wider than Cin (zero channels appended).
    int pack(const float* w_src, const float* bias_src, int cout, int cin, int kt, int kh, int kw, double scale);
};

struct ConvArgs {
    const float* in;    // channels-last [B][T][H][W][CinAct]
    const float* wp;
    const float* bias;
    const float* res;   // channels-last [B][T/rt][H/rs][W/rs][Cout] or null
    const float* coef;  // optional per-(b,c) affine (A,B) pairs applied to the input on load: x*A + B (1x1x1 convs only)
    float* out;
    int B, T, H, W;
    int CinAct;         // channel stride of `in` (>= Cin of the weights)
    int Cout, CoutPad, nchunk;
    int KT, KH, KW;
    int sS, sT;               // spatial / temporal stride (1 or 2; embedder and motion encoder); T, H, W are OUTPUT
                              // dims, the input is [B][T*sT][H*sS][W*sS][CinAct]
    int TB, TT, TH, TW;       // output brick handled by one workgroup (TB*TT*TH*TW == CONV_BM)
    int nbB, nbT, nbH, nbW;   // bricks per dimension
    int rt, rs;               // nearest-upsample factors applied when reading `res`
    int epi;
    long frames_bstride;      // EPI_FRAMES: floats between the samples of `out` (>= T * Cout * H * W)
};

// Chooses the brick and the tile variant and enqueues the kernel.
int conv_forward(const ConvWeights& wts, const float* in, int cin_act, float* out, const float* res, int rt, int rs,
                 int B, int T, int H, int W, int epi, hipStream_t st, const float* coef = nullptr, int stride = 1,
                 int stride_t = 1, long frames_bstride = 0);

// ---- exact-fp32 mode: 3x3x3 conv with a Winograd F(4,3) transform along W (i2v_wino32.hip): six 3x3x1 plane convs on the kernel above
struct Wino4F32Weights {
    ConvWeights u[6];   // U_x = (G g)_x per (kt, kh), fp64 -> fp32 once at load
    DevBuf bias;
    int Cin = 0, Cout = 0;
    // w_src: torch layout [Cout][Cin][3][3][3]; scale multiplies every weight (1 / sigma)
    int pack(const float* w_src, const float* bias_src, int cout, int cin, double scale);
};
bool wino4f32_supported(int cout, int cin, int T, int H, int W);
// V: [6][B][T][H][W/4][C] fp32 = B^T d of act((x A + B) gamma' + beta) read through the nearest up-sampling map (ut, us)
int modulate_wino4_f32(const float* x, const float* coef, const float* gb, float* V, int B, int T, int H, int W, int C, int ut, int us,
                       int lrelu, hipStream_t st);
// M: scratch [6][B][T][H][W/4][Cout]; out: channels-last fp32 [B][T][H][W][Cout]; res as conv_forward
int wino4f32_forward(const Wino4F32Weights& wts, const float* V, float* M, float* out, const float* res, int rt, int rs, int B, int T, int H,
                     int W, int epi, hipStream_t st);

// ---- 1x1(x1) convs / Linear layers as a plain pipelined GEMM (i2v_pointwise.hip); conv_forward dispatches to it
bool pointwise_supported(const ConvWeights& wts, const float* res, int rt, int rs, int epi, int stride, int stride_t);
// M = rows (B*T*H*W), P = positions per sample (row / P selects the coef row)
int pointwise_forward(const ConvWeights& wts, const float* in, int cin_act, float* out, const float* res, long M, long P,
                      int epi, hipStream_t st, const float* coef);

// ---- split-fp16 path (i2v_conv16.hip): operands carried as (fp16 hi, fp16 lo = x - hi) pairs, 3 fp16 MFMAs per product
struct Conv16Weights {
    DevBuf w;      // [tap][chunk32][CoutPad][4 groups x (8 hi | 8 lo) fp16] = 128 B per (n, chunk)
    DevBuf bias;
    int Cin = 0, Cout = 0, CoutPad = 0, nchunk = 0;
    int KT = 1, KH = 1, KW = 1;
    int wexp = 0;  // weights are stored multiplied by 2^wexp (undone in the epilogue)
    bool tdup = false;   // packed by pack_tdup: two parity sets of a 2x3x3 kernel
    long set_bytes = 0;  // bytes of one parity set
    int pack(const float* w_src, const float* bias_src, int cout, int cin, int kt, int kh, int kw, double scale);
    // 3x3x3 conv whose input is a x2 nearest up-sampling IN TIME of a half-rate tensor (frames 2i and 2i+1 identical):
    // packs the equivalent pair of 2-tap temporal kernels; conv16_forward then reads the half-rate tensor directly.
    int pack_tdup(const float* w_src, const float* bias_src, int cout, int cin, double scale);
};

// in_hl16: channels-last activations in the hl16 format (4 bytes per element, Cin % 8 == 0); out: fp32 channels-last.
// T,H,W = OUTPUT geometry (for pack_tdup weights the input tensor has T/2 frames).
// range_flag (optional, with EPI_HL16): device int set to 1 when a stored value does not fit the fp16 hi part
// splitk_ws (optional): scratch of splitk_ws_floats floats; a launch too small to fill the chip (and without fused
// statistics) then splits its K chunks over up to 8 workgroups per tile and sums the partials in a second pass
int conv16_forward(const Conv16Weights& wts, const void* in_hl16, float* out, const float* res, int rt, int rs, int B, int T,
                   int H, int W, int epi, hipStream_t st, double* stats = nullptr, int* range_flag = nullptr,
                   float* splitk_ws = nullptr, size_t splitk_ws_floats = 0);
// 1x1x1 conv / Linear on split-fp16 operands (i2v_pointwise.hip): in fp32 [M][Cin] (split on the fly, optional per-(sample,
// channel) affine `coef` [M / P][Cin][2] folded in), weights = Conv16Weights packed with kt = kh = kw = 1, out fp32 [M][Cout]
// (transposed: [Cout][M], no residual / affine).
int pointwise16_forward(const Conv16Weights& wts, const float* in, float* out, const float* res, long M, long P, int epi,
                        hipStream_t st, const float* coef = nullptr, int* range_flag = nullptr, bool transposed = false);
// K-split factor conv16_forward uses when it is given scratch (a function of the layer geometry only)
int conv16_splitk_factor(long pos_per_sample, int nchunk);
// true when conv16_forward can accumulate per-(sample, channel) sum / sum-of-squares of its output in the epilogue
bool conv16_can_fuse_stats(int T, int H, int W);

// ---- split-fp16 path with a Winograd F(2,3) transform along W (i2v_conv16w.hip): 1.5x fewer MFMAs than conv16_forward.
// Input: the TRANSFORMED activations V = B^T d in hl16 format, [B][T][Cin/16][4][H][W/2][16 channels] (written by the
// producer kernel: per output pair (2j, 2j+1) and channel V0 = d0-d2, V1 = d1+d2, V2 = d2-d1, V3 = d1-d3,
// d_k = a[.., 2j-1+k] zero padded; 16 channels = 2 groups x (8 fp16 hi | 8 fp16 lo) = 64 bytes).
struct Wino16Weights {
    DevBuf w;      // U = G g: [parity][tap (kt,kh)][chunk16][4][CoutPad][2 groups x (8 hi | 8 lo) fp16 = 64 B]
    DevBuf bias;
    int Cin = 0, Cout = 0, CoutPad = 0, nchunk = 0;
    int KT = 3;    // temporal taps (3, 1, or 2 for the temporal-duplication pair)
    int wexp = 0;
    bool tdup = false;
    long set_bytes = 0;
    // w_src: torch layout [Cout][Cin][kt][3][3]
    int pack(const float* w_src, const float* bias_src, int cout, int cin, int kt, double scale);
    int pack_tdup(const float* w_src, const float* bias_src, int cout, int cin, double scale);  // from a 3x3x3 kernel
};
// T = frames of the tensor V was built from (half the output frames for pack_tdup weights)
// KT: temporal taps of the packed weights (3; 2 = temporal-duplication pair; 1 = single time slice).  False = use the direct kernel.
bool wino16_supported(int cout, int cin, int T, int H, int W, int KT = 3);
// T,H,W = OUTPUT geometry; epi: EPI_NONE or EPI_LRELU; stats as conv16_forward
int wino16_forward(const Wino16Weights& wts, const void* v_hl16, float* out, const float* res, int rt, int rs, int B, int T,
                   int H, int W, int epi, hipStream_t st, double* stats = nullptr);

// ---- split-fp16 Winograd F(4,3) along W (i2v_conv16w4.hip): 6 GEMMs per 4 output positions (0.75x the MFMAs of F(2,3)).
// Input: V = B^T d in hl16 format, [B][T][Cin/16][6][H][W/4][16 channels = 64 B] (modulate_wino4_kernel).
struct Wino4Weights {
    DevBuf w;      // U = G g: [parity][tap (kt,kh)][chunk16][6][CoutPad/32][hi | lo][64 lanes][16 B]
    DevBuf bias;
    int Cin = 0, Cout = 0, CoutPad = 0, nchunk = 0;
    int KT = 3;    // temporal taps: 3, or 2 for the temporal-duplication pair
    int wexp = 0;
    bool tdup = false;
    long set_bytes = 0;
    int pack(const float* w_src, const float* bias_src, int cout, int cin, double scale, int kt = 3);   // w_src [Cout][Cin][kt][3][3], kt = 3 or 1
    int pack_tdup(const float* w_src, const float* bias_src, int cout, int cin, double scale);  // from a 3x3x3 kernel
};
// T = frames of the tensor V was built from (half the output frames for pack_tdup weights); false = use another kernel
bool wino4_supported(int cout, int cin, int T, int H, int W, int KT);
int wino4_forward(const Wino4Weights& wts, const void* v_hl16, float* out, const float* res, int rt, int rs, int B, int T,
                  int H, int W, int epi, hipStream_t st, double* stats = nullptr);

// The same conv with the operand GENERATED in the kernel (conv_wino4g_f16x3_kernel: 8 MFMA waves + 4 producer waves per workgroup):
// x = the conv's fp32 input BEFORE normalisation / activation, channels-last [B][T][H / us][W / us][Cin]; coef = per-(b,c) (A, B)
// pairs; gb = SPADE's gamma' | beta maps [B][H][W][2 Cin] (with us = 2) or null (ADAIN, us = 1); the operand is lrelu((x A + B) gamma'
// + beta) read through the nearest up-sampling map -- what modulate_wino4_kernel would have written, bit for bit.  32 output channels,
// 3x3x3, no temporal up-sampling in front (g_4 of the 128 x 128 configs).
bool wino4g_supported(int cout, int cin, int T, int H, int W, int us);
int wino4g_forward(const Wino4Weights& wts, const float* x, const float* coef, const float* gb, int us, float* out, const float* res, int rt,
                   int rs, int B, int T, int H, int W, int epi, hipStream_t st, double* stats = nullptr, int* range_flag = nullptr,
                   int* umax = nullptr);

// ---- helpers implemented in i2v_dec.hip, shared with the embedder (i2v_embed.hip)
// per-(b,c) sum / sum of squares (fp64) of a channels-last tensor [B][P][C]
int stats_forward(const float* x, double* sums, int B, long P, int C, hipStream_t st);
// (sum, sumsq) -> per-(b,c) (A, B) pairs with norm(x) == x*A + B (biased variance, eps 1e-5)
// gw/gb: optional per-channel affine weight / bias folded into the pairs (GroupNorm(affine=True))
int coef_forward(const double* sums, float* coef, int B, int C, int groups, double count, hipStream_t st,
                 const float* gw = nullptr, const float* gb = nullptr);
// out = act(x * A + B (+ res)) on a channels-last [B][P][C] tensor; coef index = b * cstride + c (i2v_embed.hip)
// out_hl16 (optional): the same result in the split-fp16 operand format (input of conv16_forward); out may then be null
int norm_act_forward(const float* x, const float* coef, long cstride, const float* res, float* out, int B, long P, int C, bool relu,
                     hipStream_t st, void* out_hl16 = nullptr);
// bilinear (align_corners=True) NCHW [B,3,Hi,Wi] -> channels-last [B][Ho][Wo][16] (channels 3..15 zero)
int resize_forward(const float* img, float* out, int B, int Hi, int Wi, int Ho, int Wo, hipStream_t st);

// ---- conv_img (i2v_convimg.hip): Conv3d(nf -> 3) + tanh on the vector ALU, exact fp32
struct ConvImgWeights {
    DevBuf w, bias;  // [chunk16][tap][16][4] floats, bias[3]
    int Cin = 0, nchunk = 0;
    int pack(const float* w_src, const float* bias_src, int cin);
};
bool conv_img_supported(int T, int H, int W, int C);
// ---- conv_img as a fused matrix-core kernel (per temporal tap a 32 x Cin split-fp16 GEMM over the halo positions of an 8 x 32
// brick into LDS, then a per-position gather): Cin in {16, 32, 48, 64}, H % 8 == 0, W % 32 == 0
struct ConvImgMfmaWeights {
    DevBuf w, bias;  // [dt][Cin/16][hi | lo][64 lanes][8 halfs], bias[3]
    int Cin = 0;
    int pack(const float* w_src, const float* bias_src, int cin);
};
bool conv_img_mfma_supported(int T, int H, int W, int C);
int conv_img_mfma_forward(const ConvImgMfmaWeights& wts, const float* in, float* out, int B, int T, int H, int W, hipStream_t st,
                          int* range_flag = nullptr, long out_bstride = 0);
// in: fp32 channels-last [B][T][H][W][Cin]; out: frames [B][T][3][H][W] with tanh applied
int conv_img_forward(const ConvImgWeights& wts, const float* in, float* out, int B, int T, int H, int W, hipStream_t st, long out_bstride = 0);

}  // namespace i2v
