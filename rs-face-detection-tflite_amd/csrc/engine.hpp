// engine.hpp — device-side model instance: weights in HBM, activation arena, launch plan, hipGraph replay.
//
// Replaces the interpreter the reference rebuilds on every call (`InterpreterBuilder::build` + `allocate_tensors`
// + `invoke`, /root/reference/src/face_detection_lite/face_detection.rs:207-235): here lowering, weight upload and
// arena planning happen once per handle and `run` only enqueues kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "plan.hpp"

namespace mi {

void hip_check(hipError_t e, const char* what);

class Model {
   public:
    Model(const uint8_t* bytes, size_t n, int device);
    ~Model();
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;

    int device() const { return device_; }
    const Graph& graph() const { return plan_.graph; }
    const Plan& plan() const { return plan_; }
    std::vector<int> input_dims() const { return plan_.graph.tensors[plan_.graph.inputs[0]].shape; }
    int num_outputs() const { return static_cast<int>(plan_.graph.outputs.size()); }
    const std::vector<int>& output_dims(int i) const { return plan_.graph.tensors[plan_.graph.outputs.at(i)].shape; }
    size_t output_elems(int i) const { return plan_.graph.tensors[plan_.graph.outputs.at(i)].elems(); }
    size_t input_elems() const { return plan_.graph.tensors[plan_.graph.inputs[0]].elems(); }

    // Enqueue the network for `batch` frames. `in` is a DEVICE pointer [batch, H, W, C]. Results land in the
    // model-owned device buffers returned by output_device(i) ([batch, output_elems(i)]).
    // one_shot: the caller is a single-image entry — it synchronises `stream` before anything else uses this model and then asks
    // band_failed(); such calls may run on the single-launch plan (bandnet_kernels.hip)
    void run_device(const float* in, int batch, hipStream_t stream, bool one_shot = false);
    // true when the last single-launch run gave up waiting (its results are void; the flag is cleared): repeat the call with one_shot = false
    bool band_failed();
    int get_option(const std::string& key) const;
    // workgroups a one_shot run of `batch` frames would occupy, one per CU for the whole launch (0: it would not use the single-launch plan)
    int band_workgroups(int batch);
    // Full boundary call (host or device user buffers).
    void run(const float* in, int batch, float* const* outs, int mem, hipStream_t stream);
    float* output_device(int i) const { return d_out_.at(i); }
    hipStream_t stream() const { return stream_; }
    // u8 frames as the graph input (8UC3 rows of row_bytes bytes, frames frame_bytes apart, exactly the input size of the graph):
    // possible when the graph input is read by one node only and that node is the specialised stem convolution, which then
    // normalises a byte through `lut` (device, 256 floats) while it fills its tile.  run_device with bytes instead of floats.
    bool takes_u8_input();
    void run_device_u8(const uint8_t* frames, long frame_bytes, int row_bytes, const float* lut, int batch, hipStream_t stream);

    // Per-launch timing with HIP events on the stream the kernels are launched on (eager launches, whole batch in one
    // chunk layout as configured). Returns one record per launch: kernel label, avg milliseconds, algorithmic bytes
    // and MACs for `batch` frames.
    struct LaunchStat {
        std::string kernel, detail;
        double ms = 0, bytes = 0, macs = 0;
    };
    std::vector<LaunchStat> profile(const float* in, int batch, int reps, hipStream_t stream);
    void set_option(const std::string& key, int value);
    size_t debug_tensor(int tensor, int frame, float* dst, size_t cap);
    std::string describe() const { return plan_.describe(); }

   private:
    void rebuild();                       // (re)lower + upload weights for the current options
    void build_bandnet();                 // the single-launch plan of the same graph, when its operators have band stages
    void build_bandnet_try(bool conv2_ok);
    void free_bandnet();
    bool band_usable(int batch) const;
    void band_before_launch(hipStream_t s);
    void ensure_capacity(int batch);
    void enqueue_chunk(const float* in, int chunk_start, int frames, hipStream_t s, std::vector<hipEvent_t>* marks = nullptr,
                       std::vector<std::string>* labels = nullptr);
    std::string node_label(const Node& n) const;
    void enqueue_all(const float* in, int batch, hipStream_t s);
    const float* tensor_ptr(int t, const float* in, int chunk_start, long* frame_stride) const;
    float* tensor_ptr_mut(int t, int chunk_start, long* frame_stride) const;
    void invalidate_graphs();

    int device_ = 0;
    std::vector<uint8_t> blob_;
    Plan plan_;
    int fuse_level_ = 5, res_budget_ = 156 * 1024, chunk_ = 0, use_graph_ = 1, reuse_ = 1, lanes_ = 1, pipe_max_ = 4, strip_ = 1, pipe_rows_ = 0, pipe_band_ = 0, fork_ = 1, head_streams_opt_ = 1, stem_fuse_ = 1, pair_fuse_ = 1, mdb_band_ = 0, stem_mfma_ = 1;
    int arena_lane_ = 0;                    // arena region the chunk being enqueued writes to
    std::vector<hipStream_t> side_streams_;  // lanes 1.. run on their own streams (forked/joined with events)
    std::vector<hipEvent_t> lane_events_;
    // output heads (nodes that only feed graph outputs, after the last trunk node) run on side streams beside the trunk
    std::vector<int> head_slot_;          // per node: side stream of a forked head, -1 otherwise
    std::vector<int> head_wait_;          // per forked head: node whose completion it waits for (-1: start of the plan)
    std::vector<char> event_after_;       // per node: a forked head waits for it
    std::vector<hipStream_t> head_streams_;
    std::vector<hipEvent_t> head_events_; // one per node (lazily created) + one per side stream for the join
    bool dirty_ = true;

    float* d_weights_ = nullptr;
    std::vector<long> node_w_, node_b_, node_w2_, node_b2_, node_alpha_;  // float offsets into d_weights_ (-1 none)
    std::vector<long> node_pair_;   // constants of the pair launch this Block node and the next one share (mdblock_pack_consts, pair form), -1: no such form
    std::vector<long> node_stem_;   // the first convolution's constants for the launch it shares with the block pair behind it (mdblock_pack_stem), -1: no such form
    std::vector<long> node_mwalk_;  // mwalk-kernel constants of a Block node (mwalk_pack_consts), -1 when the shape does not qualify
    std::vector<long> node_strip_;  // strip-kernel constants of a Block node (strip_pack_consts), -1 when the shape does not qualify
    struct MemberOff { long w = -1, b = -1, w2 = -1, b2 = -1, alpha = -1, strip = -1; };
    int profile_inner_ = 1;  // executions of a launch between its two profiling marks (profile() only)
    std::vector<std::vector<MemberOff>> chain_off_;  // per node: offsets of each chain member's constants
    std::vector<std::vector<MemberOff>> chain_head_off_;  // per node, per head pair: stacked weights (w2) and bias (b2)

    ResStage* d_programs_ = nullptr;        // stage programs of the Resident nodes (device memory)
    TailStage* d_tail_programs_ = nullptr;  // ... of those that run on tail_kernels.hip (Node::tail; node_prog_ indexes this array then)
    std::vector<std::vector<long>> tail_wa_, tail_wc_;  // per tail node, per stage: A operands / small constants (-1: LOAD)
    int mchain_ = 1;                        // option "mchain"
    int tail_ = 1;                          // option "tail"
    int tail_pre_ = 0;                      // option "tail_pre"
    int tail_g_ = 0;                        // option "tail_g": frames per workgroup of the tail programs (0 = chosen per launch)
    std::vector<std::vector<long>> res_wblk_;   // per Resident node, per stage: K-blocked weight packing (-1: classic order)
    std::vector<std::vector<long>> res_cblob_;  // per Resident node, per stage: offset of its packed constants (-1: LOAD)
    std::vector<long> node_prog_;           // per node: first stage in d_programs_ (-1 none)
    float* d_arena_ = nullptr;
    size_t arena_floats_ = 0;
    int chunk_cap_ = 0;      // frames per chunk the arena is laid out for
    int batch_cap_ = 0;      // frames the output buffers hold
    std::vector<float*> d_out_;
    float* d_in_stage_ = nullptr;
    size_t in_stage_floats_ = 0;
    int last_chunk_frames_ = 0;

    // single-launch plan for one_shot runs (bandnet_kernels.hip): the first convolution as in the batched plan, everything behind it ONE launch
    int band_ = 1;                  // option "band": 0 never, 1 one_shot runs, 2 every run of few enough frames (tests, profiling)
    int band_nw_ = 128;             // option "band_nw": most workgroups per frame
    int band_nw_used_ = 0;          // ... of the program that was built
    bool band_ready_ = false;       // the graph has a single-launch form
    bool band_use_ = false;         // the run being enqueued takes it
    bool band_ran_ = false;         // the last run_device took it
    bool band_test_fail_ = false;   // option "band_test_fail"
    int band_test_absent_ = 0;      // option "band_test_absent"
    int band_fail_streak_ = 0;      // single launches in a row that gave up; at 3 the handle stops using the plan (band_ = 0)
    bool band_disabled_ = false;    // ... which it did
    unsigned band_gen_ = 0;         // band launches since the workspace was last cleared (tags wrap at 2^26: band_before_launch)
    bool band_gen_force_ = false;   // option "band_test_gen"
    int band_wraps_ = 0;            // times the workspace was cleared for that reason
    size_t band_ws_bytes_ = 0;
    int band_first_ = 0;            // plan_ node the band launch stands for (with every node behind it that band_node_runs_ does not name)
    int band_stem_out_ = -1;        // tensor the first convolution writes = the band program's input
    int band_nstages_ = 0, band_lds_bytes_ = 0, band_max_frames_ = 0, band_dw_floats_ = 0;
    int band_tiles_floats_ = 0;     // LDS floats of the program's tiles
    long band_ws_frame_floats_ = 0;
    struct BandExt { int out_k = -1, tensor = -1; };   // BandLaunch::base[2 + j]: graph output out_k, or the arena storage of `tensor` (read by a launch behind the band program)
    std::vector<BandExt> band_ext_;
    std::vector<char> band_node_runs_;   // per plan_ node from band_first_ on: 1 = it runs as its own launch behind the band launch (the program stops in front of it)
    bool band_fork_ = true, band_cv2_ = false, band_xb_ = false, band_saw_conv2_ = false, band_wide_ = false, band_wide_ok_ = true;
    BandPacked* d_band_prog_ = nullptr;
    float* d_band_consts_ = nullptr;
    float* d_band_ws_ = nullptr;
    unsigned* d_band_sync_ = nullptr;
    int* h_band_fail_ = nullptr;    // pinned + mapped: the kernel raises it, band_failed() reads it without a copy
    int* d_band_fail_ = nullptr;

    hipStream_t stream_ = nullptr;
    int small_chain_ = 16;          // option "small_chain" (measured on BackCamera: block by block wins up to ~24 frames, tools/small_batch_probe.py)
    float* d_small_ = nullptr;      // [2][small_chain_][largest chain frame] ping-pong scratch
    size_t small_floats_ = 0;
    struct GraphKey {
        const void* in;
        int batch;
        long u8_frame_bytes;  // 0: f32 input
        int u8_row_bytes;
        int band = 0;         // 1: the single-launch plan
        bool operator<(const GraphKey& o) const {
            if (in != o.in) return in < o.in;
            if (batch != o.batch) return batch < o.batch;
            if (u8_frame_bytes != o.u8_frame_bytes) return u8_frame_bytes < o.u8_frame_bytes;
            if (u8_row_bytes != o.u8_row_bytes) return u8_row_bytes < o.u8_row_bytes;
            return band < o.band;
        }
    };
    struct U8Input {  // set by run_device_u8 for the duration of the enqueue
        const uint8_t* frames = nullptr;
        const float* lut = nullptr;
        long frame_bytes = 0;
        int row_bytes = 0;
    } u8_;
    void run_graph_or_eager(const float* in, int batch, hipStream_t s, const GraphKey& key);
    std::map<GraphKey, hipGraphExec_t> graphs_;
};

}  // namespace mi
