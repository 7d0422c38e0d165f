// plan.hpp — lowering of a parsed .tflite graph to a list of fused kernel launches (host side).
//
// The reference executes these graphs op by op inside TensorFlow-Lite (`interpreter.invoke()`,
// /root/reference/src/face_detection_lite/face_detection.rs:235).  Here the graph is rewritten once, at load:
//   level 0  one launch per builtin op (RESHAPE = view, CONCATENATION = producers write into the joined buffer)
//   level 1  ADD / RELU / PRELU / MAX_POOL_2D-skip / channel-PAD-skip / RESIZE_BILINEAR-skip folded into the
//            producing convolution's epilogue
//   level 2  DEPTHWISE_CONV_2D -> CONV_2D 1x1 (-> skip -> activation) fused into one BlazeBlock kernel
//   level 3  runs of same-shape stride-1 BlazeBlocks whose frame fits in LDS fused into one frame-resident chain kernel
//   level 4  runs of narrow (C <= 24, W <= 128) stride-1 BlazeBlocks cut into row-pipelined chains of up to 4 blocks: one
//            launch, the intermediate rows handed from block to block through LDS (strip_kernels.hip)
//   level 5  the small-spatial parts (outputs of <= 256 pixels) become frame-resident stage programs: independent branches are
//            made contiguous, then maximal runs of blocks / pointwise convs / k x k stride-k convs whose live activations
//            fit in LDS are one launch each (resident_kernels.hip; round 5: tail_kernels.hip where its stage forms exist, which
//            also takes the face mesh's frame-resident chains with their stride-2 neighbours)
#pragma once

#include <string>
#include <vector>

#include "kernels.hpp"
#include "tflite_graph.hpp"

namespace mi {

struct Node {
    enum Kind { Conv, Dw, Block, Add, Act, MaxPool, Pad, Reshape, Concat, Resize, DepthToSpace, Chain, Resident } kind = Conv;
    std::vector<Node> members;  // Chain: the fused BlazeBlocks, in order; Resident: the fused nodes (chains expanded), in order
    // Resident: the stage program.  ResStage geometry / LDS placement is final; global references and weight offsets are
    // filled in by the engine from (src_t, dst_t, res_t) and the member's constants.
    struct Stage {
        ResStage st;
        TailStage tst;                         // Node::tail: the stage in tail_kernels.hip's form (st is unused then)
        int member = -1;                       // index into members (-1: LOAD stage)
        int src_t = -1, dst_t = -1, res_t = -1;  // tensors behind the global references (-1: none)
    };
    std::vector<Stage> stages;
    int res_const_off = 0, res_const_floats = 0, res_lds_bytes = 0;
    int res_bands = 1;           // Resident: workgroups per frame (row bands; 1 = the whole frame is resident)
    bool tail = false;           // Resident: the stage program runs on tail_kernels.hip (several frames per workgroup, 16x16x4 MFMAs)
    int tail_frame_floats = 0;   // ... LDS floats one frame needs (activations + depthwise scratch, placed by liveness)
    int tail_min_px = 0;         // ... fewest output pixels per frame among its stages (the engine picks the frames per workgroup)
    bool dblock = false;         // Resident without stages: members are the two blocks of a full_range double block (dblock_kernels.hip)
    bool xc = false;             // Resident without stages: members alternate expand / contract blocks on a tiny frame (xc_kernels.hip)
    bool bneck = false;          // Resident without stages: members are (pointwise C -> Cm, depthwise block Cm -> C + skip) pairs run by
                                 // bneck_kernels.hip with the C-channel tensor in registers
    // frame-resident Chain: output heads run by its launch.  head_nodes = the absorbed 1x1 convolutions; a pair = up to two of them
    // on the same tensor (src 0: the chain's final frame, 1: chain_post's output), contracted as one stacked product
    struct HeadPair { int src = 0, a = -1, b = -1; };
    std::vector<Node> head_nodes;
    std::vector<HeadPair> head_pairs;
    bool chain_pre = false, chain_post = false;  // frame-resident Chain: members.front() / members.back() is the stride-2 block before / after the resident blocks
    bool gemm_head = false;      // Conv whose window is the whole frame: runs as a GEMM over the batch (head_gemm_kernel)
    std::vector<int> extra_out;  // Resident: further tensors the launch writes to global memory (besides `out`)
    std::vector<int> in;   // activation inputs (tensor ids)
    int out = -1;
    bool dead = false;
    // convolution parameters (Conv / Dw / Block: the depthwise stage)
    int w = -1, b = -1;
    int KH = 1, KW = 1, sh = 1, sw = 1;
    Padding padding = Padding::Same;
    int ept = -1, epl = -1;        // explicit zero pad before the first row / column (a spatial PAD op folded into this node;
                                   // padding == Valid then means "valid on the padded input"); -1: none
    // Block: pointwise stage
    int w2 = -1, b2 = -1;
    // fused epilogue
    int act = ACT_NONE;
    int alpha = -1;
    int res = -1;
    int res_mode = RES_NONE;
    bool res_after = false;        // the skip is added behind the activation (Epilogue::res_after): block / generic convolution kernels only
    // misc
    int pads = -1;                 // PAD: paddings tensor
    int axis = 0;                  // CONCAT
    int filter_h = 1, filter_w = 1;
    int block_size = 1;
    bool half_pixel = false, align_corners = false;
    std::vector<int> src_ops;      // indices of the .tflite operators folded into this node (for describe())
};

struct Storage {
    int root = -1;        // tensor id that owns the buffer
    long offset = 0;      // floats from the root's frame start
    long frame_stride = 0;
};

struct Plan {
    Graph graph;
    std::vector<Node> nodes;            // live nodes in execution order
    std::vector<int> branch;            // per node: -1 = trunk; >= 0 = independent chain behind the plan's last fork (0 stays on the trunk stream)
    std::vector<Storage> storage;       // per tensor
    std::vector<long> root_offset;      // per tensor (valid for roots): float offset inside the arena, per frame-slot
    std::vector<long> root_elems;       // per root: floats per frame
    long arena_floats_per_frame = 0;
    int fuse_level = 5;
    double bytes_per_frame = 0, macs_per_frame = 0;
    std::string describe() const;
};

// pipe_max: most blocks one row-pipelined chain may hold (level 4; 2..4, below 2 disables them)
// res_budget_bytes: LDS a frame-resident stage program may use (level 5)
// tail: stage programs take tail_kernels.hip's form where it exists (false: the round-4 plan)
Plan build_plan(Graph g, int fuse_level, int pipe_max = 4, int res_budget_bytes = 156 * 1024, bool tail = true);

}  // namespace mi
