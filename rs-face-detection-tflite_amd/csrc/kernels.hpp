// kernels.hpp — launch descriptors shared between the planner (host) and the HIP kernels (kernels.hip).
// Activations are f32 NHWC; every tensor is addressed as base + frame * frame_stride (+ pixel * C + c), so that
// RESHAPE is a view and CONCATENATION is done by producers writing straight into the concatenated buffer.
#pragma once

#include <cstddef>
#include <cstdint>

namespace mi {

enum ActKind : int { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_PRELU = 3 };
// How the skip connection of a BlazeBlock is read inside the fused epilogue:
//   RES_DIRECT  : skip[c] = res[pixel][c] for c < res_C, 0 above (channel PAD fused)
//   RES_MAXPOOL : skip[c] = max over the 2x2 stride-2 window of the (2Ho x 2Wo) source, c < res_C (MAX_POOL_2D fused)
//   RES_UP2X    : skip[c] = bilinear x2 upsample (half_pixel_centers) of a (Ho/2 x Wo/2) source (RESIZE_BILINEAR fused)
enum ResMode : int { RES_NONE = 0, RES_DIRECT = 1, RES_MAXPOOL = 2, RES_UP2X = 3 };

struct Epilogue {
    const float* bias = nullptr;   // [Co] or null
    const float* alpha = nullptr;  // PReLU slopes [Co]
    const float* res = nullptr;    // skip-connection source
    long res_fs = 0;               // frame stride of res (floats)
    int res_mode = RES_NONE;
    int res_C = 0;                 // channels present in res (pixel stride of res)
    int res_W = 0, res_H = 0;      // source spatial size (RES_MAXPOOL / RES_UP2X)
    int act = ACT_NONE;
    int res_after = 0;             // 1: out = act(conv + bias) + skip (a convolution with a fused activation followed by ADD: the lateral
                                   // connections of full_range's decoder), 0: out = act(conv + bias + skip)
};

struct ConvArgs {
    const float* in = nullptr;
    // u8 input (the stem of a detector fed with 8UC3 frames, transform.rs:292-301 folded into the tile load): in_u8 != null replaces `in`;
    // a value v becomes u8_lut[v] = (f32)((f64)v * (max - min) / 255 + min), the 256 results computed on the host with that expression
    const uint8_t* in_u8 = nullptr;
    const float* u8_lut = nullptr;
    long u8_frame_bytes = 0;
    int u8_row_bytes = 0;
    const float* w = nullptr;   // [KH][KW][C][Cop]   (Cop = Co rounded up to 4, zero padded)
    float* out = nullptr;
    long in_fs = 0, out_fs = 0;
    int B = 0, H = 0, W = 0, C = 0, Ho = 0, Wo = 0, Co = 0, Cop = 0;
    int KH = 1, KW = 1, sh = 1, sw = 1, pt = 0, pl = 0;
    Epilogue ep;
    int no_mfma = 0;            // 1: the 5 x 5 stem stays on stem_conv_kernel (engine option "stem_mfma" = 0)
};

// Convolution whose window is the whole frame (VALID, H == KH, W == KW: the 3x3-stride-3 mesh head 32 -> 1404, the 2x2 iris
// heads 128 -> 213 / 15): per frame a matrix-vector product over K = KH*KW*C contiguous floats, so over a batch a plain
// GEMM  out[B][N] = x[B][K] * W[N][K]^T + bias  whose weights are read once per 32 frames instead of once per frame.
struct HeadGemmArgs {
    const float* in = nullptr;   // frame b: K contiguous floats at in + b * in_fs
    const float* w = nullptr;    // [N][K] (TFLite OHWI as stored)
    const float* bias = nullptr; // [N] or null
    const float* alpha = nullptr;
    float* out = nullptr;        // frame b: N floats at out + b * out_fs
    long in_fs = 0, out_fs = 0;
    int B = 0, K = 0, N = 0;
    int act = ACT_NONE;
};

struct DwArgs {
    const float* in = nullptr;
    const float* w = nullptr;   // [3][3][C]
    float* out = nullptr;
    long in_fs = 0, out_fs = 0;
    int B = 0, H = 0, W = 0, C = 0, Ho = 0, Wo = 0;
    int KH = 3, KW = 3, sh = 1, sw = 1, pt = 0, pl = 0;
    Epilogue ep;
};

// Fused BlazeBlock: DW3x3(+bias) -> PW1x1 (MFMA f32) -> +bias +skip -> activation, one kernel, DW result never
// leaves the CU.  w_dw [3][3][C]; w_pw packed for the MFMA B/A operand by pack_pw_weights().
struct BlockArgs {
    const float* in = nullptr;
    const float* w_dw = nullptr;
    const float* b_dw = nullptr;
    const float* w_pw = nullptr;  // [Cop][Cp] row-major (out channel major), zero padded: Cp = C up to 4, Cop = Co up to 16/32
    const float* w_strip = nullptr;  // strip_pack_consts() blob when the shape qualifies for strip_kernels.hip, else null
    const float* w_mwalk = nullptr;  // mwalk_pack_consts() blob when the shape qualifies for mwalk_kernels.hip, else null
    float* out = nullptr;
    long in_fs = 0, out_fs = 0;
    int B = 0, H = 0, W = 0, C = 0, Ho = 0, Wo = 0, Co = 0;
    int sh = 1, sw = 1, pt = 0, pl = 0;
    int has_dw = 1;               // 0: plain pointwise conv (no depthwise stage)
    int pipe_rows = 0;            // row pipelines: rows per step (0 = automatic: 2 when the height is even; 1 forces strip_pipe_kernel)
    int pipe_band = 0;            // row pipelines: rows per band (0 = automatic: about one resident set of workgroups over the chip)
    Epilogue ep;
};

// Frame-resident chain of same-shape stride-1 BlazeBlocks (chain_kernels.hip): x <- act(PW(DW(x)) + b + [x]) per block.
constexpr int kMaxChain = 8;
struct ChainBlock {
    const float* w_dw = nullptr;   // [3][3][C]
    const float* b_dw = nullptr;   // [C] or null
    const float* w_pw = nullptr;   // A-fragment order (see block_weight_dims)
    const float* bias = nullptr;   // [C] or null
    const float* alpha = nullptr;  // PReLU slopes [C]
    int act = ACT_NONE;
    int has_res = 0;               // skip connection = the block's own input
};
// Optional stride-2 BlazeBlock run by the chain's launch in front of (`pre`) or behind (`post`) the resident blocks:
// DW3x3 s2 (TF SAME on an even size: taps at rows 2oy..2oy+2) -> PW -> [+ 2x2 max-pool of its input, zero channel-padded] -> act.
//   pre : reads its (2H x 2W x Cin) input from global memory, leaves its H x W x C output in the LDS tile instead of HBM
//   post: reads the chain's final H x W x C frame from the tile, writes its (H/2 x W/2 x Co) output to global memory
struct ChainEdge {
    int on = 0;
    ChainBlock blk;                // weights in the block kernel's packing; blk.has_res = max-pool skip
    const float* in = nullptr;     // pre: input tensor
    long in_fs = 0;
    int Cin = 0;                   // pre: input channels (the input is 2H x 2W x Cin)
    float* out = nullptr;          // post: output tensor
    long out_fs = 0;
    int Co = 0;                    // post: output channels (<= 32 * ceil(C / 32))
};
// Output heads run by the chain's launch: up to two 1x1 convolutions without activation on the same resident tensor (the SSD
// regressor and classifier heads), contracted as ONE pointwise product with their rows stacked (A rows [0, Co_a) then [Co_a,
// Co_a + Co_b)); each writes its own tensor.  src: 0 = the chain's final frame (H x W x C, the LDS tile), 1 = the output of
// `post` (H/2 x W/2 x post.Co, kept in LDS for this purpose).
struct ChainHead {
    int on = 0, src = 0;
    const float* w_pw = nullptr;   // stacked rows in A-fragment order (block_weight_dims(C_src, Co_a + Co_b))
    const float* bias = nullptr;   // [Co_a + Co_b] stacked, or null
    int Co_a = 0, Co_b = 0;        // Co_a % 4 == 0; Co_b may be ragged (2, 6 classifier channels), 0 = no second head
    float* out_a = nullptr;
    float* out_b = nullptr;
    long out_a_fs = 0, out_b_fs = 0;
};
constexpr int kChainHeads = 2;
struct ChainArgs {
    const float* in = nullptr;
    float* out = nullptr;
    long in_fs = 0, out_fs = 0;
    int B = 0, H = 0, W = 0, C = 0, nblocks = 0;
    ChainBlock blocks[kMaxChain];
    ChainEdge pre, post;
    ChainHead heads[kChainHeads];
    int write_out = 1;             // 0: nobody but `post` / the heads reads the chain's own output: it is not written
};

struct EltArgs {  // ADD / activation / MAX_POOL / channel PAD / RESIZE / DEPTH_TO_SPACE fallbacks (un-fused graphs)
    const float* a = nullptr;
    const float* b = nullptr;
    const float* alpha = nullptr;
    float* out = nullptr;
    long a_fs = 0, b_fs = 0, out_fs = 0;
    int B = 0, H = 0, W = 0, C = 0, Ho = 0, Wo = 0, Co = 0;
    int act = ACT_NONE;
    int p0 = 0, p1 = 0, p2 = 0, p3 = 0;  // op-specific ints (pool filter/stride, pad offsets, block size, flags)
};

struct RectD {  // layout-compatible with mi_rect (include/mi_face.h) / Rect (types.rs:24-36)
    double x_center, y_center, width, height, rotation;
    int normalized;
    int pad_;
};

struct PostArgs {  // fused SSD decode + sigmoid + threshold + weighted NMS + letterbox removal, one workgroup/frame
    const float* raw_boxes = nullptr;   // [B][N][16]
    const float* raw_scores = nullptr;  // [B][N]
    const float* anchors = nullptr;     // [N][2]
    const double* padding = nullptr;    // [B][4] or null
    float* out = nullptr;               // [B][cap][17]
    int* counts = nullptr;              // [B]; -1 where the reference's letterbox-scale assert would fire
    int B = 0, N = 0, cap = 0;
    float scale = 1.f;
    // the batched pipeline (lib.rs:24-40 per frame): slots behind the frame's last detection are written as zeros (no memset in front of the
    // launch), and the ROI of faces[0] (face_detection_to_roi on a picture of image_w x image_h) comes out of the same launch
    int zero_rest = 0;
    RectD* face_rois = nullptr;         // [B] or null
    int* face_valid = nullptr;          // [B]
    int image_w = 0, image_h = 0;
};

struct ProjArgs {  // project_landmarks, one thread per landmark
    const float* raw = nullptr;   // [B][raw_fs] (first n*3 floats used)
    long raw_fs = 0;
    const RectD* roi = nullptr;   // [B] or null
    const int* image_size = nullptr;  // [B][2] (w,h) or null
    const float* flag = nullptr;      // optional face-flag logits [B][flag_fs]
    long flag_fs = 0;
    int* present = nullptr;           // [B] sigmoid(flag) > 0.5  (face_landmark.rs:292-296)
    float* raw_flag_out = nullptr;    // [B] or null
    const double* padding = nullptr;  // [B][4] or null
    const int* flip = nullptr;        // [B] or null
    const int* gate = nullptr;        // [B] or null: 0 = item has no ROI (no face upstream) -> zeros, present = 0
    float* out = nullptr;             // [B][out_fs] (first n*3 floats written)
    long out_fs = 0;                  // floats between items (0 = n*3)
    int B = 0, n = 0, tensor_w = 1, tensor_h = 1;
    // optional second list of landmarks of the same items (same ROI / padding / flip; no flag): raw2 [B][raw2_fs] -> out2 [B][out2_fs]
    const float* raw2 = nullptr;
    long raw2_fs = 0;
    float* out2 = nullptr;
    long out2_fs = 0;   // floats between items (0 = n2*3)
    int n2 = 0;
};


// Frame-resident stage programs (resident_kernels.hip, fuse level 5): ONE workgroup per frame runs a list of fused stages
// with the activations kept in LDS.  Descriptors are pointer-free (they live in device memory, written once per plan):
// a global tensor is named by (base index, offsets) and resolved against the ResBases passed with each launch.
struct ResRef {
    int base = -1;        // index into ResBases (-1: unused)
    int pad_ = 0;
    long root_off = 0;    // floats, multiplied by ResBases::scale[base] (arena tensors are laid out per chunk capacity)
    long inner = 0;       // floats
    long fs = 0;          // frame stride (floats)
};
enum ResKind : int { RES_STAGE_LOAD = 0, RES_STAGE_GATHER = 1, RES_STAGE_DW = 2 };
struct ResStage {
    int kind = RES_STAGE_GATHER;
    int dw_pg = 0;                     // DW: 32-pixel groups per batch (depthwise result of a batch -> LDS scratch -> pointwise)
    // source: LDS tensor (src_off >= 0; pixel (y,x) at src_off + ((y+b)*(W+2b) + x+b)*PS) or dense NHWC in global memory
    int src_off = -1, src_H = 0, src_W = 0, src_C = 0, src_PS = 0, src_b = 0;
    ResRef src_g;
    int KH = 1, KW = 1, S = 1, pt = 0, pl = 0;
    int Kv = 0;                        // contraction length: KH*KW*C (GATHER) or C (DW)
    int Ho = 0, Wo = 0, Co = 0;
    int dst_off = -1, dst_PS = 0, dst_b = 0;
    int zero_dst = 0;                  // floats to clear at dst_off before the stage runs
    ResRef dst_g;                      // optional global copy of the output, dense [Ho][Wo][Co]
    int res_mode = RES_NONE, res_C = 0, res_H = 0, res_W = 0;
    int res_off = -1, res_PS = 0, res_b = 0;   // skip source in LDS, else res_g
    ResRef res_g;
    int act = ACT_NONE;
    int dw_off = -1;                   // DW: LDS floats, scratch [dw_pg*32][roundup8(C) + 4]
    // Row bands (ResLaunch::bands > 1): a workgroup owns output rows [r0, r0 + band_rows) of an image of band_H rows.
    //   role 1 (pointwise GATHER from global): produces rows [r0 - 1, r1 + 1) clipped to the image into a bordered LDS tensor
    //           of band_rows interior rows (the halo rows of interior bands are recomputed, rows outside the image stay zero)
    //   role 2 (DW from that tensor): produces rows [r0, r1) into global memory (skip read from global memory)
    int band_role = 0, band_rows = 0, band_H = 0;
    // Pointwise GATHER from global memory with Kv % 16 == 0: the contraction runs in blocks of kblk = 16 channels that each wave
    // stages through a private LDS slab at dw_off (coalesced 64 B per pixel instead of 16 B pieces 128 B apart); the weights are
    // packed for that order (channel of (k-half h, chunk j, e) = 16*(j/2) + 8h + 4*(j%2) + e).  0 = classic order.
    int kblk = 0;
    // float offsets into the weights blob: pointwise / k x k weights in A-fragment order; the stage's small constants, padded and
    // ready to be copied to LDS: [9][Cp] depthwise taps + [Cp] depthwise bias (DW only), [Cop] bias, [Cop] negative-side slopes
    long w_pw = -1, cblob = -1;
};
constexpr int kResBases = 8;
struct ResBases {
    float* p[kResBases];
    long scale[kResBases];   // multiplier of ResRef::root_off
    int frame0[kResBases];   // first frame of this launch inside the tensor
    const float* weights;
};
struct ResLaunch {
    const ResStage* prog = nullptr;  // device memory
    int nstages = 0, B = 0;          // B = frames
    int bands = 1;                   // workgroups per frame (row bands)
    int const_off = 0;               // LDS floats: start of the per-stage constants area (two halves of const_floats each)
    int const_floats = 0;
    int lds_bytes = 0;
    ResBases bases;
};
int launch_resident(const ResLaunch& a, void* stream);
int resident_const_floats(const ResStage& st);  // LDS floats the stage's constants need
constexpr int kResSlabFloats = 8 * 32 * 20;        // LDS floats of the 8 waves' staging slabs (kblk stages)
constexpr int kResConstMax = 5120;               // most constants a stage may have (10 per thread, prefetched in registers)

// Stage programs with several frames per workgroup (tail_kernels.hip): the same three stage kinds on v_mfma_f32_16x16x4_f32, every
// LDS tensor a dense [G * H * W][C + 4] array without borders (per-frame offsets, multiplied by G at run time).
enum TailKind : int { TAIL_LOAD = 0, TAIL_GATHER = 1, TAIL_DW = 2 };
struct TailStage {
    int kind = TAIL_GATHER;
    int K = 1, S = 1, pt = 0, pl = 0;  // GATHER: K x K window with stride K (1 or 2); DW: 3 x 3, stride S, TF SAME pads before the first row / column
    int src_off = -1;                  // LDS floats per frame (-1: global memory, LOAD and DW only)
    int src_H = 0, src_W = 0, src_C = 0;
    ResRef src_g;
    int Kv = 0;                        // contraction length: K * K * src_C (GATHER), src_C (DW)
    int Ho = 0, Wo = 0, Co = 0;
    int dst_off = -1;                  // LDS floats per frame (-1: none)
    ResRef dst_g;                      // optional global copy, dense [Ho][Wo][Co] (base < 0: none)
    int res_mode = RES_NONE, res_C = 0, res_W = 0;
    int res_off = -1;                  // skip source in LDS (floats per frame), else res_g
    ResRef res_g;
    int act = ACT_NONE;
    int scr_off = -1;                  // DW: the depthwise scratch [Ho * Wo][Kv + 4], LDS floats per frame
    int pool_off = -1;                 // DW stride 2 whose skip is the 2x2 max-pool of its own source (pads 0): the depthwise phase leaves the pooled
                                       // pixels here ([Ho * Wo][Kv + 4]: taps (0..1, 0..1) are the pool window) and the epilogue reads them as a direct skip
    unsigned mHW = 0, mW = 0;          // tail_magic(Ho * Wo), tail_magic(Wo); LOAD: tail_magic(H * W * C / 4), tail_magic(C / 4)
    unsigned mHWp = 0, mWp = 0;        // DW: tail_magic(Ho * ceil(Wo / 2)), tail_magic(ceil(Wo / 2)) (the depthwise phase works on pixel pairs)
    long w_a = -1;                     // float offsets into the weights blob: A operands [tile][Kv / 16][lane][4] ...
    long w_c = -1;                     // ... and [bias 16 nct][slope 16 nct]([9][Kv] depthwise taps, [Kv] depthwise bias)
};
struct TailLaunch {
    const TailStage* prog = nullptr;   // device memory
    int nstages = 0, B = 0;
    int G = 1;                         // frames per workgroup
    int frame_floats = 0;              // LDS floats per frame (activations + depthwise scratch)
    int variant = 0;                   // 0: chosen per launch; 1: constants a stage ahead (256 registers, one workgroup per CU); 2: 128 registers, two per CU
    ResBases bases;
};
int launch_tail(const TailLaunch& a, void* stream);
bool tail_stage_ok(int kind, int K, int S, int src_C, int Kv, int Co, bool dst_lds);
unsigned tail_magic(int d);

// ---- launchers (kernels.hip). All enqueue on `stream` and return hipError_t as int (0 = success).
int launch_conv(const ConvArgs& a, void* stream);
const char* conv_kernel_label(const ConvArgs& a);
bool conv_takes_u8(const ConvArgs& a);  // the specialised stem kernel: the only convolution with a u8 input form
int launch_dw(const DwArgs& a, void* stream);
int launch_block(const BlockArgs& a, void* stream);
bool block_kernel_supports(const BlockArgs& a);
const char* block_kernel_label(const BlockArgs& a, char* buf, size_t cap);  // instantiation name as rocprofv3 prints it
// strip_kernels.hip: register-resident variant for stride-1 blocks with C = Co in {16, 24, 32}
int launch_strip(const BlockArgs& a, void* stream);
bool strip_kernel_supports(const BlockArgs& a);
bool strip_shape_ok(int C, int Co);
int strip_consts_floats(int C);
void strip_pack_consts(int C, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst);
const char* strip_kernel_label(const BlockArgs& a, char* buf, size_t cap);
// mstrip_kernels.hip: stride-1 blocks with C = Co = 48 on 32-pixel-wide frames: depthwise stage in the MFMA operand layout, pointwise
// conv on v_mfma_f32_16x16x4_f32 with resident weights (constants blob through BlockArgs::w_strip)
// mwalk_kernels.hip: the same scheme for other widths / channel counts (face mesh 48x48x32); constants in BlockArgs::w_mwalk
int launch_mwalk(const BlockArgs& a, void* stream);
bool mwalk_kernel_supports(const BlockArgs& a);
bool mwalk_shape_ok(int W, int C, int Co);
int mwalk_consts_floats(int W, int C, int Co);
void mwalk_pack_consts(int W, int C, int Co, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst);
const char* mwalk_kernel_label(const BlockArgs& a, char* buf, size_t cap);
// ms2_kernels.hip: stride-2 blocks in the same scheme (two input rows per step, max-pool skip from the row images); constants in BlockArgs::w_mwalk
int launch_ms2(const BlockArgs& a, void* stream);
bool ms2_kernel_supports(const BlockArgs& a);
bool ms2_shape_ok(int W, int C, int Co, bool skip);   // W: input width
int ms2_consts_floats(int W, int C, int Co, bool skip);
void ms2_pack_consts(int W, int C, int Co, bool skip, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst);
const char* ms2_kernel_label(const BlockArgs& a, char* buf, size_t cap);
int launch_mstrip(const BlockArgs& a, void* stream);
bool mstrip_kernel_supports(const BlockArgs& a);
bool mstrip_shape_ok(int C, int Co);
int mstrip_consts_floats(int C);
void mstrip_pack_consts(int C, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst);
const char* mstrip_kernel_label(const BlockArgs& a, char* buf, size_t cap);
// a run of 2..8 such blocks (each reading its predecessor's output, 32 x 32 frames) as ONE launch: a workgroup per frame, a barrier between blocks
bool mstrip_chain_supports(const BlockArgs* blocks, int n);
int launch_mstrip_chain(const BlockArgs* blocks, int n, void* stream);
// row-pipelined chain of 2..4 strip-eligible blocks (blocks[k+1].in == blocks[k].out, which never reaches HBM)
bool strip_pipe_supports(const BlockArgs* blocks, int n);
bool strip_pipe_shape_ok(int C, int W);  // host-only shape tests for the planner
bool strip_tail_shape_ok(int C, int Co, int H, int W);
int strip_consts_s2_floats(int C, int Co);
void strip_pack_consts_s2(int C, int Co, const float* w_dw, const float* b_dw, const float* w_pw, const float* bias, const float* alpha, int act, float* dst);
int launch_strip_pipe(const BlockArgs* blocks, int n, void* stream);
const char* strip_pipe_label(const BlockArgs* blocks, int n, char* buf, size_t cap);
int strip_pipe_rows_per_step(int H, int hint = 0);
int launch_chain(const ChainArgs& a, void* stream);

// ---- bottleneck blocks with the wide tensor in registers (bneck_kernels.hip):
//   r = act1(W1 . x + b1) (C -> Cm);  y = act2(W2 . (DW3x3(r) + b_dw) + b2 + x) (Cm -> C)
struct BneckBlock {
    const float* w1 = nullptr;      // [Cm/32 tiles][C/8 chunks][64 lanes][4]: lane (row r, half h), e -> W1[32 t + r][32 (j/4) + 8 (j%4) + 4 h + e]
    const float* w2 = nullptr;      // block kernel's A-fragment packing of W2 [C][Cm]
    const float* consts = nullptr;  // bneck_const_floats(): [b1 Cm][slope1 Cm][dw taps 9 x Cm][b_dw Cm][b2 C][slope2 C] (slope: PReLU alpha, 0 = ReLU, 1 = none)
    float hi1 = 0.f, hi2 = 0.f;     // upper clamps (6 for ReLU6, +inf otherwise)
    const float* mconsts = nullptr; // mbneck_pack_consts() blob when the pair has an mdblock_kernels.hip form (else null)
    int act1 = ACT_RELU, act2 = ACT_RELU;
};
constexpr int kMaxBneck = 6;
struct BneckArgs {
    const float* in = nullptr;
    float* out = nullptr;
    long in_fs = 0, out_fs = 0;
    int B = 0, H = 0, W = 0, C = 0, Cm = 0, nblocks = 0;
    int bands = 1;                  // workgroups per frame: 1 = the frame (<= 256 pixels) is resident, a run of blocks per launch;
                                    // > 1: row bands of ceil(H / bands) rows (<= 256 pixels each), one block per launch
    BneckBlock blocks[kMaxBneck];
};
// ---- runs of expand / contract BlazeBlocks on tiny frames as one frame-resident launch (xc_kernels.hip):
//   even stages: wide = act(W . (DW3x3(narrow) + b_dw) + b + skip);  odd stages: narrow = act(W . (DW3x3(wide) + b_dw) + b)
struct XcStage {
    const float* cblob = nullptr;   // xc_const_floats(C, Co) floats, 16-byte aligned: depthwise taps [9][Cp] (zeros: none), depthwise bias [Cp], pointwise bias [Cop]
                                    // (Cp = C rounded up to 8, Cop = Co rounded up to 32): copied to LDS as they are
    int has_dw = 1;                 // 0: the stage is pointwise only
    const float* w_pw = nullptr;    // block kernel's A-fragment packing of W [Co][C]
    int C = 0, Co = 0;
    int act = ACT_RELU;             // ACT_NONE / ACT_RELU / ACT_RELU6
    int skip = 0;                   // 0: none; 1: the stage's own input (channels >= C: zero); 2: 2x2 max-pool of `res` ([2H][2W][res_C], channels >= res_C: zero);
                                    // 3: the output of the expand stage two stages back (same shape; updated in place)
    const float* res = nullptr;
    long res_fs = 0;
    int res_C = 0, res_W = 0;
};
constexpr int kMaxXc = 8;
struct XcArgs {
    const float* in = nullptr;      // [B][H][W][st[0].C]
    float* out = nullptr;           // [B][H][W][st[nstages - 1].Co]
    long in_fs = 0, out_fs = 0;
    int B = 0, H = 0, W = 0, nstages = 0;
    XcStage st[kMaxXc];
};
bool xc_kernel_supports(const XcArgs& a);
int xc_const_floats(int C, int Co);
int launch_xc(const XcArgs& a, void* stream);
// ---- full_range's double BlazeBlock as one launch (dblock_kernels.hip):
//   a = act1(W1 . (DW3x3(x) + b_dw1) + b1) (C -> Cm);  y = act2(W2 . (DW3x3(a) + b_dw2) + b2 + pad(x)) (Cm -> Co >= C)
struct DblockArgs {
    const float* in = nullptr;
    float* out = nullptr;
    long in_fs = 0, out_fs = 0;
    int B = 0, H = 0, W = 0, C = 0, Cm = 0;
    int Co = 0;                     // output channels >= C (the skip is x zero-padded to Co channels, as where full_range widens: 48 -> 16 -> 64)
    int skip1 = 0;                  // a pair of plain BlazeBlocks instead: stage 1 adds x as well (Cm == C) ...
    int skip2_from_a = 0;           // ... and stage 2's skip is a, the first block's output (Co == Cm)
    const float* consts = nullptr;  // dblock_const_floats(): [dw1 taps 9 x C][b_dw1 C][b1 32 MTA][slope1 32 MTA][dw2 taps 9 x Cmp][b_dw2 Cmp][b2 32 MT][slope2 32 MT]
    const float* w1 = nullptr;      // block kernel's A-fragment packing of W1 [Cm][C] (MTA = ceil(Cm / 32) tiles)
    const float* w2 = nullptr;      // ... of W2 [C][Cm] (MT tiles, contraction padded to Cmp = roundup8(Cm))
    float hi1 = 0.f, hi2 = 0.f;
    const float* mconsts = nullptr; // mdblock_pack_consts() blob when the layer has an mdblock_kernels.hip form (else null)
    int act1 = ACT_RELU, act2 = ACT_RELU;
    // mdblock_kernels.hip, pair form behind the network's first convolution (MD::STEM, round 6): that convolution runs inside the launch — `in` is never read
    const float* stem_in = nullptr;      // [B][2 H][2 W][3] f32
    long stem_in_fs = 0;
    const float* stem_consts = nullptr;  // mdblock_pack_stem()
    float stem_hi = 0.f;                 // upper clamp of its activation
    int band_rows = 0;                   // mdblock_kernels.hip: rows per band (0: chosen by the launcher)
};
bool dblock_kernel_supports(const DblockArgs& a);
// mdblock_kernels.hip: the double blocks of the 96- and 48-pixel-wide layers with both pointwise convs as 16x16x4 MFMAs in the
// operand layout (row-walking waves, taps in registers)
// (pair: two plain BlazeBlocks in a row, each adding its own input, instead of the double block)
bool mdblock_shape_ok(int W, int C, int Cm, int Co, bool pair = false);
int mdblock_consts_floats(int W, int C, int Cm, int Co, bool pair = false);
void mdblock_pack_consts(int W, int C, int Cm, int Co, const float* w_dw1, const float* b_dw1, const float* w_pw1, const float* b1, const float* alpha1, int act1,
                         const float* w_dw2, const float* b_dw2, const float* w_pw2, const float* b2, const float* alpha2, int act2, float* dst, bool pair = false);
bool mdblock_kernel_supports(const DblockArgs& a);
bool mdblock_stem_shape_ok(int H, int W, int C, int KH, int KW, int sh, int sw, int Ho, int Wo, int Co);
int mdblock_stem_consts_floats();
void mdblock_pack_stem(const float* w, const float* bias, const float* alpha, int act, float* dst);
int launch_mdblock(const DblockArgs& a, void* stream);
int dblock_const_floats(int C, int Cm, int Co);
int launch_dblock(const DblockArgs& a, void* stream);

bool bneck_kernel_supports(const BneckArgs& a);
// mdblock_kernels.hip, mbneck_kernel: the 32-pixel-wide bottleneck pairs (64 -> 32 -> 64) in the operand layout
bool mbneck_shape_ok(int W, int C, int Cm);
int mbneck_consts_floats(int W, int C, int Cm);
void mbneck_pack_consts(int W, int C, int Cm, const float* w_pw1, const float* b1, const float* alpha1, int act1, const float* w_dw2, const float* b_dw2,
                        const float* w_pw2, const float* b2, const float* alpha2, int act2, float* dst);
bool mbneck_kernel_supports(const BneckArgs& a);
int launch_mbneck(const BneckArgs& a, void* stream);
int bneck_const_floats(int C, int Cm);
int launch_bneck(const BneckArgs& a, void* stream);
int launch_head_gemm(const HeadGemmArgs& a, void* stream);
bool head_gemm_supports(int K, int N);
bool chain_kernel_supports(const ChainArgs& a);
int launch_add(const EltArgs& a, void* stream);
int launch_act(const EltArgs& a, void* stream);
int launch_maxpool(const EltArgs& a, void* stream);
int launch_padc(const EltArgs& a, void* stream);
int launch_resize2x(const EltArgs& a, void* stream);
int launch_depth_to_space(const EltArgs& a, void* stream);
int launch_copy_strided(const EltArgs& a, void* stream);
int launch_postprocess(const PostArgs& a, void* stream);
int launch_project(const ProjArgs& a, void* stream);
int launch_spin(long long ticks, int max_iters, int* sink, void* stream);   // one idle wave for `ticks` of the 100 MHz wall clock (bounded)

// ---- bandnet_kernels.hip: a whole BlazeBlock network behind its first convolution as ONE launch, for the handful of frames of a
// single-image call (face_detection.rs:205: one Mat per call).  NW workgroups per frame; every stage (one BlazeBlock or one 1x1
// convolution) cuts its output into row bands, one per active workgroup, which keeps its band in LDS for the next stage; only the
// rows a neighbour needs (the 3x3 halo) travel through global memory, as 16-byte packets {value, tag, value, tag} (sc0 sc1: the
// workgroups of a frame sit on all eight XCDs, whose L2s are not coherent with each other) that the reader polls until both tags are
// the producer stage's — no flag round trip, no wait for the stores, no launch boundary and no grid-wide barrier between blocks.
enum BandKind : int { BAND_BLOCK = 0, BAND_PW = 1 };
constexpr int kBandBases = 8;
constexpr int kBandTiles = 4;
struct alignas(16) BandStage {
    int kind = BAND_BLOCK;
    int S = 1;                   // BLOCK: DW3x3 stride (1: pad 1; 2: TF SAME on an even size, taps at 2o .. 2o+2); PW with S == 2: a 2x2 stride-2
                                 // convolution (C % 32 == 0; contraction over (ky, kx, c), weights [Co][2][2][C] as TFLite stores them)
    int H = 0, W = 0, C = 0;     // input  (C % 4 == 0, C <= 128)
    int Ho = 0, Wo = 0, Co = 0;  // output (Co <= 128)
    int R = 1;                   // output rows per band
    int wshift = 0;              // workgroup w runs the stage when its low wshift bits are 0; it owns band w >> wshift (the owner of output row r
                                 // also owns input row S r: its part of the input is still in LDS)
    int woff = 0;                // ... or, for the second branch behind a fork, when they are woff (< 1 << wshift): workgroups the first branch leaves idle
    int nbands = 0;
    int dep = -1;                // stage that produces the input (-1: in global memory, complete before the launch)
    int Rin = 0;                 // input rows of this band that the workgroup owns (dep >= 0: the producer's R), from row S r0 on
    int src_tile = 0;            // LDS tile (0 .. kBandTiles - 1) that holds the input band: rows [S r0 - 1, S r0 + Rin + 2) at tile rows 0 .., one zero pixel left and right
    int dst_tile = -1;           // LDS tile the output band is left in for the stages that read it (-1: nobody does)
    int pub_lo = 0, pub_hi = 0;  // the first pub_lo and last pub_hi rows of the band also go to the packet buffer (other workgroups read them)
    int src_base = 0, dst_base = -1;  // BandLaunch::base index of the input (dep < 0) / of a plain copy of the output (-1: none; graph outputs, and tensors
                                 // the launches behind the band program read)
    int res_mode = RES_NONE;     // RES_DIRECT: the input itself (S == 1), RES_MAXPOOL: 2x2 max of the input (S == 2); channels >= C: zero (Co >= C)
    int res_dep = -2;            // (host only) stage whose output the skip is when it is not the stage's input (-1: the program's input, -2: none)
    int res_tile = -1;           // -1: the skip is the stage's own input; else RES_DIRECT from the tensor in this LDS tile (same rows, Co channels:
                                 // the iris network's bottlenecks add the tensor in front of their 1x1 reduction), or RES_MAXPOOL of the C-channel
                                 // tensor of twice the size in this tile (the input of the 2x2 convolution in front of this block)
    int cross = 0;               // (host only) the stage reads a tensor of the other branch's workgroups: Rin = 0, every row from the packets
    int src_lds = 0, dst_lds = 0, res_lds = 0;   // LDS floats in front of the input / output / skip tile (multiples of 16; placed by liveness in build_bandnet)
    int pre = 0;                 // a stride-2 BLOCK whose 3x3 window starts at row / column 2r - 1 (an explicit zero pad of one pixel in front, VALID behind it:
                                 // full_range_sparse); the WIDE instantiation only
    int dst_h3 = 0;              // the output tile has R + 3 rows (a stride-2 block reads the tensor: two rows below the band), else R + 2
    int far_src = 0;             // (host only) the stage reads its input — rows this workgroup wrote itself, many stages ago — back from the launch's workspace instead
                                 // of keeping the tile alive (full_range's lateral convolutions read trunk tensors 10 - 30 stages old); packed with dep = -1
    int far_copy = 0;            // (host only) ... and the producer writes that copy (dst_base = 0: the workspace)
    long res_ll = -1;            // RES_UP2X (a 1x1 stage whose skip, added behind the activation, is the bilinear x2 up-sampling of the output of stage
    int res_stage = 0;           // res_stage, half its size): that tensor's packet buffer
    int res_c = 0;               // channels of the skip tensor in res_tile (<= Co: the rest of the skip is the zero pad of a widening block)
    int act = ACT_NONE;
    long src_off = 0, dst_off = 0;   // floats from the base to frame 0 of the tensor
    long src_fs = 0, dst_fs = 0;     // floats between frames
    long src_ll = -1, dst_ll = -1;   // packet buffers of the input / output tensor inside base[0] (floats from the frame's workspace; 2 x the tensor)
    long w_a = 0;                // BandLaunch::consts offset: A operands of v_mfma_f32_16x16x4_f32, per 16-channel output tile
                                 //   [C/16 chunks][64 lanes][4] then (C % 16 == 8) [64 lanes][2]: lane (kq = l / 16, m = l % 16) holds W[16 t + m][16 j + 4 kq + e]
                                 //   (8-chunk: W[16 t + m][16 (C/16) + 2 kq + e]); then (C % 8 == 4) [64 lanes]: W[16 t + m][C - 4 + kq]
    long w_c = 0;                // [bias 16 nct][slope 16 nct] then BLOCK: [taps 9 C][depthwise bias C]
    int c_floats = 0;
    int wpc_shift = 0;           // log2 of the waves per 16-channel output tile (8 / nct rounded down to a power of two)
    int per_ct = 0;              // floats of A operands per output tile
    unsigned mC4 = 0, mWo = 0, mrowq = 0;   // magic numbers ceil(2^32 / d) of C / 4, Wo, W C / 4 (0: d = 1): n / d = umulhi(n, m) for n < 2^16
};
// A stage as the kernel reads it: 20 dwords (one LDS read per lane, one v_readlane per dword; the full descriptor costs twice that)
constexpr int kBandPackedWords = 20;
struct alignas(16) BandPacked {
    unsigned w[kBandPackedWords];
};
// false: a field does not fit its bits (offsets are 32-bit, shapes 16-bit)
bool bandnet_pack(const BandStage& st, BandPacked* out);
struct BandLaunch {
    const BandPacked* prog = nullptr;
    int nstages = 0, NW = 0, F = 0, lds_bytes = 0;
    int tiles_floats = 0;           // LDS floats of the tiles (each stage names its own: BandStage::src_lds ...)
    int cv2 = 0;                    // the program has 2x2 stride-2 convolution stages (the kernel instantiation with their code)
    int wide = 0;                   // ... stages of more than 128 input or output channels (the WIDE instantiation)
    int xb = 0;                     // ... a BLOCK stage that takes all of its input rows from the packets (Rin = 0: zeroes its tile's border pixels)
    int dw_floats = 0;              // LDS: [tiles][depthwise result][small constants][program]
    long ws_frame_floats = 0;       // base[0]: floats between the frames' workspaces
    float* base[kBandBases] = {};   // 0: the launch's own workspace (packet buffers), 1: the first stage's input, 2..: graph outputs / tensors later launches read
    const float* consts = nullptr;
    unsigned* sync = nullptr;       // [0] generation (tags are 64 x generation + stage + 1), [1] workgroups finished, [2] somebody has given up (cleared by the last workgroup)
    int* fail = nullptr;            // set to 1 when a wait ran out of iterations (host-visible): the results of that launch are void
    int absent_mod = 0;             // test hook: every absent_mod-th workgroup leaves at once without publishing (a workgroup that is not resident)
    unsigned long long* stamps = nullptr;  // diagnostic builds only (MI_BAND_STAMPS)
};
int launch_bandnet(const BandLaunch& a, void* stream);
// LDS floats of a band of R rows of a W x C tensor (with its halo rows and border pixels) / of a stage's depthwise result / small constants
int bandnet_tile_floats(int R, int W, int C, int halo);
int bandnet_dw_floats(const BandStage& st);
int bandnet_const_floats(const BandStage& st);
int bandnet_lds_bytes(int tiles_floats, int dw_floats, int nstages);

// Pointwise weight packing for the fused block kernel: [Co][C] (TFLite [O,1,1,I]) -> [Cop][Cp] zero padded.
void block_weight_dims(int C, int Co, int* Cp, int* Cop);

}  // namespace mi
