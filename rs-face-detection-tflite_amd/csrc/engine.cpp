// engine.cpp — see engine.hpp.
#include "engine.hpp"
#include "launch.hpp"

#include <mutex>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>

namespace mi {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

namespace {
constexpr int kHeadStreams = 4;  // most side streams for output heads that run beside the trunk (option "heads")

void same_pad(int in, int k, int stride, int out, int* before) {
    int total = std::max(0, (out - 1) * stride + k - in);
    *before = total / 2;
}
long align_up(long v, long a) { return (v + a - 1) / a * a; }
}  // namespace

Model::Model(const uint8_t* bytes, size_t n, int device) : device_(device), blob_(bytes, bytes + n) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw std::runtime_error("no HIP device visible: the MI355X kernels cannot run (no CPU fallback exists)");
    if (device < 0 || device >= count) throw std::runtime_error("device ordinal out of range");
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hipDeviceProp_t prop;
    hip_check(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        throw std::runtime_error(std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName);
    hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    rebuild();
}

Model::~Model() {
    hipSetDevice(device_);
    invalidate_graphs();
    if (d_weights_) hipFree(d_weights_);
    if (d_programs_) hipFree(d_programs_);
    if (d_tail_programs_) hipFree(d_tail_programs_);
    if (d_arena_) hipFree(d_arena_);
    if (d_small_) hipFree(d_small_);
    if (d_in_stage_) hipFree(d_in_stage_);
    free_bandnet();
    for (float* p : d_out_)
        if (p) hipFree(p);
    for (hipStream_t st : side_streams_) hipStreamDestroy(st);
    for (hipEvent_t ev : lane_events_) hipEventDestroy(ev);
    for (hipStream_t st : head_streams_) hipStreamDestroy(st);
    for (hipEvent_t ev : head_events_) if (ev) hipEventDestroy(ev);
    if (stream_) hipStreamDestroy(stream_);
}

void Model::invalidate_graphs() {
    // a replay may still be running on a caller's stream (asynchronous entry points): let the device drain first
    if (!graphs_.empty()) (void)hipDeviceSynchronize();
    for (auto& kv : graphs_) hipGraphExecDestroy(kv.second);
    graphs_.clear();
}

void Model::set_option(const std::string& key, int value) {
    if (key == "chunk") chunk_ = std::max(0, value);
    else if (key == "graph") use_graph_ = value != 0;
    else if (key == "fuse") { fuse_level_ = std::min(5, std::max(0, value)); dirty_ = true; }
    else if (key == "res_budget") { res_budget_ = std::min(156, std::max(16, value)) * 1024; dirty_ = true; }  // LDS (KiB) a stage program may use
    else if (key == "pipe") { pipe_max_ = std::min(4, std::max(0, value)); dirty_ = true; }   // blocks per row-pipelined chain (level 4)
    else if (key == "small_chain") { small_chain_ = std::max(0, std::min(value, 64)); if (d_small_) { hipFree(d_small_); d_small_ = nullptr; small_floats_ = 0; } invalidate_graphs(); }  // frames up to which a row-pipelined chain runs one launch per block (0: never)
    else if (key == "pipe_rows") { pipe_rows_ = (value == 1 || value == 2 || value == 4) ? value : 0; }     // 1: one row per pipeline step (strip_pipe_kernel), 2: two rows, packed-FMA pointwise convs (strip_pipe2_kernel), 4: one row, MFMA pointwise convs (strip_pipe1m_kernel)
    else if (key == "pipe_band") { pipe_band_ = std::max(0, std::min(value, 4096)); }   // rows per band of the row pipelines (0: automatic)
    else if (key == "strip") { strip_ = value != 0; }
    else if (key == "stem_fuse") { stem_fuse_ = value != 0; }   // 0: the first convolution keeps its own launch in front of the face mesh's block pair (mdblock_kernel<stem+pair>, round 6)
    else if (key == "stem_mfma") { stem_mfma_ = value != 0; }   // 0: the detectors' 5x5 first convolution stays on stem_conv_kernel (packed VALU FMAs) at every batch size; the results are bit-identical
    else if (key == "pair_fuse") { pair_fuse_ = value != 0; }   // 0: two plain blocks in a row keep a launch each (the face mesh's 48x48x32 blocks: mwalk_kernel) instead of one mdblock_kernel<pair>
    else if (key == "mdb_band") { mdb_band_ = std::max(0, std::min(value, 4096)); }   // rows per band of the mdblock_kernel launches (0: chosen per launch)
    else if (key == "mchain") { mchain_ = value != 0; }   // 0: the 32x32x48 blocks run one launch each (mstrip_kernel) instead of one launch per run
    else if (key == "tail") { tail_ = value != 0; dirty_ = true; }   // 0: no stage program runs on tail_kernels.hip (the round-4 plan)
    else if (key == "tail_pre") { tail_pre_ = std::max(0, std::min(value, 2)); }   // tail programs: 0 = chosen per launch, 1 = constants a stage ahead (one workgroup per CU), 2 = 128 registers (two per CU)
    else if (key == "tail_g") { tail_g_ = std::max(0, std::min(value, 64)); }   // frames per workgroup of the tail stage programs (0: chosen per launch)
    else if (key == "band") { band_ = std::max(0, std::min(value, 2)); dirty_ = true; }   // single-launch plan: 0 never, 1 one_shot runs (the single-image entries), 2 every run of few enough frames
    else if (key == "band_test_fail") { band_test_fail_ = value != 0; }   // test hook: the next single-launch run reports that it gave up
    else if (key == "band_test_absent") { band_test_absent_ = std::max(0, std::min(value, 256)); }   // test hook: every k-th workgroup of a single-launch run leaves at once without publishing (a workgroup that is not resident; 0: off)
    else if (key == "band_test_gen") { band_gen_ = static_cast<unsigned>(std::max(0, value)); band_gen_force_ = true; }   // test hook: the host's count of band launches (the packet tags wrap at 2^26: see run_device)
    else if (key == "band_wide") { band_wide_ok_ = value != 0; dirty_ = true; }   // single-launch plan: stages of more than 128 channels (0: the program ends in front of the first one)
    else if (key == "band_fork") { band_fork_ = value != 0; dirty_ = true; }   // single-launch plan: the second branch behind a fork on the idle workgroups
    else if (key == "band_nw") { band_nw_ = std::max(8, std::min(value, 256)); dirty_ = true; }   // its workgroups per frame
    else if (key == "fork") { fork_ = value != 0; }
    else if (key == "heads") { head_streams_opt_ = std::min(kHeadStreams, std::max(1, value)); dirty_ = true; }  // side streams the output heads are spread over                                             // 0: output heads stay on the trunk's stream                                          // 0: LDS-ring block kernel for every block
    else if (key == "reuse") { reuse_ = value != 0; dirty_ = true; }
    else if (key == "lanes") lanes_ = std::min(4, std::max(1, value));
    else throw std::runtime_error("unknown option '" + key + "'");
    invalidate_graphs();
    chunk_cap_ = 0;  // arena is re-laid out on the next run
}

int Model::get_option(const std::string& key) const {
    if (key == "chunk") return chunk_;
    if (key == "graph") return use_graph_;
    if (key == "fuse") return fuse_level_;
    if (key == "res_budget") return res_budget_ / 1024;
    if (key == "pipe") return pipe_max_;
    if (key == "small_chain") return small_chain_;
    if (key == "pipe_rows") return pipe_rows_;
    if (key == "pipe_band") return pipe_band_;
    if (key == "strip") return strip_;
    if (key == "stem_fuse") return stem_fuse_;
    if (key == "stem_mfma") return stem_mfma_;
    if (key == "pair_fuse") return pair_fuse_;
    if (key == "mdb_band") return mdb_band_;
    if (key == "mchain") return mchain_;
    if (key == "tail") return tail_;
    if (key == "tail_pre") return tail_pre_;
    if (key == "tail_g") return tail_g_;
    if (key == "band") return band_;
    if (key == "band_fork") return band_fork_;
    if (key == "band_wide") return band_wide_ok_;
    if (key == "band_nw") return band_nw_;
    if (key == "band_fail_streak") return band_fail_streak_;
    if (key == "band_wraps") return band_wraps_;
    if (key == "fork") return fork_;
    if (key == "heads") return head_streams_opt_;
    if (key == "reuse") return reuse_;
    if (key == "lanes") return lanes_;
    throw std::runtime_error("unknown option '" + key + "'");
}

void Model::rebuild() {
    if (d_small_) { hipFree(d_small_); d_small_ = nullptr; small_floats_ = 0; }  // sized for the plan it was allocated under
    hip_check(hipSetDevice(device_), "hipSetDevice");
    invalidate_graphs();
    plan_ = build_plan(parse_tflite(blob_.data(), blob_.size()), fuse_level_, pipe_max_, res_budget_, tail_ != 0);
    const Graph& g = plan_.graph;
    if (!reuse_) {  // debugging layout: every tensor keeps its own slot
        long off = 0;
        for (size_t t = 0; t < g.tensors.size(); t++)
            if (plan_.root_offset[t] >= 0) { plan_.root_offset[t] = off; off += plan_.root_elems[t]; }
        plan_.arena_floats_per_frame = off;
    }
    // ---- pack every constant the launches need into one device blob
    std::vector<float> host;
    auto put = [&](const std::vector<float>& v) {
        long off = align_up(static_cast<long>(host.size()), 64);
        host.resize(static_cast<size_t>(off) + v.size(), 0.f);
        std::copy(v.begin(), v.end(), host.begin() + off);
        return off;
    };
    const size_t NN = plan_.nodes.size();
    node_w_.assign(NN, -1); node_b_.assign(NN, -1); node_w2_.assign(NN, -1); node_b2_.assign(NN, -1); node_alpha_.assign(NN, -1);
    chain_off_.assign(NN, {});
    chain_head_off_.assign(NN, {});
    node_strip_.assign(NN, -1);
    node_mwalk_.assign(NN, -1);
    node_stem_.assign(NN, -1);
    node_pair_.assign(NN, -1);
    res_cblob_.assign(NN, {});
    res_wblk_.assign(NN, {});
    tail_wa_.assign(NN, {});
    tail_wc_.assign(NN, {});
    // pointwise weights [O][1][1][I] -> MFMA A-fragment order [tile][k-chunk][lane][4]:
    // lane l = (row m = l & 31, k-half h = l >> 5) holds W[tile*32 + m][h*Cp/2 + 4*chunk + e], zero padded
    auto pack_pw = [&](int wt, int kblk = 0) {
        const auto& ws = g.tensors[wt].shape;
        const auto& src = g.tensors[wt].f32;
        // [O][KH][KW][I] read as [O][KH*KW*I]: the k x k stride-k convolutions of the stage programs contract over the
        // "virtual channels" (tap, channel) in exactly this order
        int O = ws[0], I = ws[1] * ws[2] * ws[3], Cp, Cop;
        block_weight_dims(I, O, &Cp, &Cop);
        const int Ch = Cp / 2, MT = Cop / 32;
        std::vector<float> r(static_cast<size_t>(Cop) * Cp, 0.f);
        for (int mt = 0; mt < MT; mt++)
            for (int j = 0; j < Ch / 4; j++)
                for (int l = 0; l < 64; l++)
                    for (int e2 = 0; e2 < 4; e2++) {
                        int o = mt * 32 + (l & 31), c = (l >> 5) * Ch + 4 * j + e2;
                        if (kblk == 16) c = 16 * (j / 2) + 8 * (l >> 5) + 4 * (j % 2) + e2;  // K-blocked order of the LDS-staged pointwise stages
                        if (o < O && c < I) r[((static_cast<size_t>(mt) * (Ch / 4) + j) * 64 + l) * 4 + e2] = src[static_cast<size_t>(o) * I + c];
                    }
        return put(r);
    };
    for (size_t i = 0; i < NN; i++) {
        const Node& n = plan_.nodes[i];
        if (n.kind == Node::Chain) {
            for (const Node& m : n.members) {
                MemberOff mo;
                mo.w = put(g.tensors[m.w].f32);
                if (m.b >= 0) mo.b = put(g.tensors[m.b].f32);
                mo.w2 = pack_pw(m.w2);
                if (m.b2 >= 0) mo.b2 = put(g.tensors[m.b2].f32);
                if (m.alpha >= 0) mo.alpha = put(g.tensors[m.alpha].f32);
                const auto& ws = g.tensors[m.w2].shape;  // [O][1][1][I]
                if (n.chain_pre || n.chain_post) {
                    // frame-resident chain with stride-2 edge stages: block-kernel packing only
                } else if (m.sh == 2) {  // stride-2 tail of a row pipeline
                    std::vector<float> sc(static_cast<size_t>(strip_consts_s2_floats(ws[3], ws[0])));
                    strip_pack_consts_s2(ws[3], ws[0], g.tensors[m.w].f32.data(), m.b >= 0 ? g.tensors[m.b].f32.data() : nullptr, g.tensors[m.w2].f32.data(),
                                         m.b2 >= 0 ? g.tensors[m.b2].f32.data() : nullptr, m.alpha >= 0 ? g.tensors[m.alpha].f32.data() : nullptr, m.act, sc.data());
                    mo.strip = put(sc);
                } else if (strip_shape_ok(ws[3], ws[0])) {
                    std::vector<float> sc(static_cast<size_t>(strip_consts_floats(ws[3])));
                    strip_pack_consts(ws[3], g.tensors[m.w].f32.data(), m.b >= 0 ? g.tensors[m.b].f32.data() : nullptr, g.tensors[m.w2].f32.data(),
                                      m.b2 >= 0 ? g.tensors[m.b2].f32.data() : nullptr, m.alpha >= 0 ? g.tensors[m.alpha].f32.data() : nullptr, m.act, sc.data());
                    mo.strip = put(sc);
                }
                chain_off_[i].push_back(mo);
            }
            // a row-pipelined pair of plain BlazeBlocks on a wide layer also gets the constants of the operand-layout kernel
            // (mdblock_kernels.hip, pair form; which of the two runs is decided per launch)
            if (n.members.size() == 2 && !n.chain_pre && !n.chain_post && n.head_pairs.empty()) {
                const Node &pa = n.members[0], &pb = n.members[1];
                const auto& sx = g.tensors[pa.in[0]].shape;
                auto plain = [&](const Node& m) { return m.w >= 0 && m.sh == 1 && m.sw == 1 && m.res == m.in[0] && m.res_mode == RES_DIRECT && !m.res_after && m.padding == Padding::Same; };
                const int C = sx[3], Cm = g.tensors[pa.out].shape[3], Co = g.tensors[pb.out].shape[3];
                if (sx.size() == 4 && plain(pa) && plain(pb) && pb.in[0] == pa.out && mdblock_shape_ok(sx[2], C, Cm, Co, true)) {
                    std::vector<float> mc(static_cast<size_t>(mdblock_consts_floats(sx[2], C, Cm, Co, true)));
                    auto ptr = [&](int t) { return t >= 0 ? g.tensors[t].f32.data() : nullptr; };
                    mdblock_pack_consts(sx[2], C, Cm, Co, ptr(pa.w), ptr(pa.b), ptr(pa.w2), ptr(pa.b2), pa.act == ACT_PRELU ? ptr(pa.alpha) : nullptr, pa.act,
                                        ptr(pb.w), ptr(pb.b), ptr(pb.w2), ptr(pb.b2), pb.act == ACT_PRELU ? ptr(pb.alpha) : nullptr, pb.act, mc.data(), true);
                    node_mwalk_[i] = put(mc);   // (the slot of the Block nodes' mwalk constants: a Chain node has none of its own)
                }
            }
            // output heads of the launch: the weights of a pair stacked [Co_a + Co_b][C] in A-fragment order, the biases stacked
            for (const Node::HeadPair& hp : n.head_pairs) {
                const Node* hn[2] = {&n.head_nodes[static_cast<size_t>(hp.a)], hp.b >= 0 ? &n.head_nodes[static_cast<size_t>(hp.b)] : nullptr};
                auto wt = [&](const Node& m) { return m.kind == Node::Conv ? m.w : m.w2; };
                auto bt = [&](const Node& m) { return m.kind == Node::Conv ? m.b : m.b2; };
                const int I = g.tensors[wt(*hn[0])].shape[3];
                std::vector<float> rows, bias;
                for (const Node* m : hn) {
                    if (!m) continue;
                    const auto& w = g.tensors[wt(*m)];
                    if (w.shape[3] != I || w.shape[1] != 1 || w.shape[2] != 1) throw std::runtime_error("engine: output heads of a pair differ in their input width");
                    rows.insert(rows.end(), w.f32.begin(), w.f32.end());
                    if (bt(*m) >= 0) bias.insert(bias.end(), g.tensors[bt(*m)].f32.begin(), g.tensors[bt(*m)].f32.end());
                    else bias.insert(bias.end(), static_cast<size_t>(w.shape[0]), 0.f);
                }
                const int O = static_cast<int>(rows.size()) / I, MT = (O + 31) / 32, Ch = I / 2;
                std::vector<float> r(static_cast<size_t>(MT) * 32 * I, 0.f);
                for (int mt = 0; mt < MT; mt++)
                    for (int j = 0; j < Ch / 4; j++)
                        for (int l = 0; l < 64; l++)
                            for (int e2 = 0; e2 < 4; e2++) {
                                const int o = mt * 32 + (l & 31), c = (l >> 5) * Ch + 4 * j + e2;
                                if (o < O) r[((static_cast<size_t>(mt) * (Ch / 4) + j) * 64 + l) * 4 + e2] = rows[static_cast<size_t>(o) * I + c];
                            }
                bias.resize(static_cast<size_t>(MT) * 32, 0.f);
                MemberOff mo;
                mo.w2 = put(r);
                mo.b2 = put(bias);
                chain_head_off_[i].push_back(mo);
            }
            continue;
        }
        if (n.kind == Node::Resident && n.xc) {
            // expand / contract runs (xc_kernels.hip): per member the depthwise taps and biases as stored, the pointwise matrix in the block kernel's packing
            for (const Node& m : n.members) {
                MemberOff mo;
                mo.w2 = pack_pw(m.w2);
                // the stage's small constants as the kernel copies them to LDS: taps [9][Cp], depthwise bias [Cp], pointwise bias [Cop]
                const int C = g.tensors[m.in[0]].shape[3], Co = g.tensors[m.out].shape[3], Cp = (C + 7) & ~7;
                std::vector<float> cb(static_cast<size_t>(xc_const_floats(C, Co)), 0.f);
                if (m.w >= 0)
                    for (int t = 0; t < 9; t++)
                        for (int c = 0; c < C; c++) cb[static_cast<size_t>(t) * Cp + c] = g.tensors[m.w].f32[static_cast<size_t>(t) * C + c];
                if (m.b >= 0)
                    for (int c = 0; c < C; c++) cb[static_cast<size_t>(9) * Cp + c] = g.tensors[m.b].f32[static_cast<size_t>(c)];
                if (m.b2 >= 0)
                    for (int c = 0; c < Co; c++) cb[static_cast<size_t>(10) * Cp + c] = g.tensors[m.b2].f32[static_cast<size_t>(c)];
                mo.strip = put(cb);
                chain_off_[i].push_back(mo);
            }
            res_wblk_[i].clear();
            res_cblob_[i].clear();
            continue;
        }
        if (n.kind == Node::Resident && n.dblock) {
            // double block (dblock_kernels.hip): both pointwise matrices in the block kernel's packing, one blob of small constants
            const Node &pa = n.members[0], &pb = n.members[1];
            const int C = g.tensors[pa.in[0]].shape[3], Cm = g.tensors[pa.out].shape[3], Co = g.tensors[pb.out].shape[3], Cmp = (Cm + 7) & ~7, MT = (Co + 31) / 32, MTA = (Cm + 31) / 32;
            MemberOff ma, mb;
            ma.w2 = pack_pw(pa.w2);
            mb.w2 = pack_pw(pb.w2);
            std::vector<float> cb(static_cast<size_t>(dblock_const_floats(C, Cm, Co)), 0.f);
            auto slope = [&](const Node& m, int c) { return m.act == ACT_PRELU ? g.tensors[m.alpha].f32[static_cast<size_t>(c)] : (m.act == ACT_NONE ? 1.f : 0.f); };
            size_t o = 0;
            for (int tap = 0; tap < 9; tap++)
                for (int c = 0; c < C; c++) cb[o + static_cast<size_t>(tap) * C + c] = g.tensors[pa.w].f32[static_cast<size_t>(tap) * C + c];
            o += static_cast<size_t>(9) * C;
            for (int c = 0; c < C; c++) cb[o + c] = pa.b >= 0 ? g.tensors[pa.b].f32[static_cast<size_t>(c)] : 0.f;
            o += static_cast<size_t>(C);
            for (int c = 0; c < Cm; c++) { cb[o + c] = pa.b2 >= 0 ? g.tensors[pa.b2].f32[static_cast<size_t>(c)] : 0.f; cb[o + static_cast<size_t>(32) * MTA + c] = slope(pa, c); }
            o += static_cast<size_t>(64) * MTA;
            for (int tap = 0; tap < 9; tap++)
                for (int c = 0; c < Cm; c++) cb[o + static_cast<size_t>(tap) * Cmp + c] = g.tensors[pb.w].f32[static_cast<size_t>(tap) * Cm + c];
            o += static_cast<size_t>(9) * Cmp;
            for (int c = 0; c < Cm; c++) cb[o + c] = pb.b >= 0 ? g.tensors[pb.b].f32[static_cast<size_t>(c)] : 0.f;
            o += static_cast<size_t>(Cmp);
            for (int c = 0; c < Co; c++) { cb[o + c] = pb.b2 >= 0 ? g.tensors[pb.b2].f32[static_cast<size_t>(c)] : 0.f; cb[o + static_cast<size_t>(32) * MT + c] = slope(pb, c); }
            ma.strip = put(cb);
            // the wide layers also get the constants of the operand-layout kernel (mdblock_kernels.hip); which of the two runs is decided per launch
            const int Wd = g.tensors[pa.in[0]].shape[2];
            if (mdblock_shape_ok(Wd, C, Cm, Co) && pa.res < 0 && pb.res == pa.in[0]) {
                std::vector<float> mc(static_cast<size_t>(mdblock_consts_floats(Wd, C, Cm, Co)));
                auto ptr = [&](int t) { return t >= 0 ? g.tensors[t].f32.data() : nullptr; };
                mdblock_pack_consts(Wd, C, Cm, Co, ptr(pa.w), ptr(pa.b), ptr(pa.w2), ptr(pa.b2), pa.act == ACT_PRELU ? ptr(pa.alpha) : nullptr, pa.act,
                                    ptr(pb.w), ptr(pb.b), ptr(pb.w2), ptr(pb.b2), pb.act == ACT_PRELU ? ptr(pb.alpha) : nullptr, pb.act, mc.data());
                mb.strip = put(mc);
            }
            chain_off_[i].push_back(ma);
            chain_off_[i].push_back(mb);
            res_wblk_[i].clear();
            res_cblob_[i].clear();
            continue;
        }
        if (n.kind == Node::Resident && n.bneck) {
            // bottleneck pairs (bneck_kernels.hip): per pair the first pointwise matrix with its contraction index in the MFMA result
            // order, the second in the block kernel's order, and one blob of small constants
            for (size_t k = 0; k + 1 < n.members.size(); k += 2) {
                const Node &pa = n.members[k], &pb = n.members[k + 1];
                const auto& w1s = g.tensors[pa.w2].shape;  // [Cm][1][1][C]
                const int Cm = w1s[0], C = w1s[3];
                const auto& w1 = g.tensors[pa.w2].f32;
                std::vector<float> r(static_cast<size_t>(Cm) * C, 0.f);
                const int NCH1 = C / 8;
                for (int t = 0; t < Cm / 32; t++)
                    for (int j = 0; j < NCH1; j++)
                        for (int l = 0; l < 64; l++)
                            for (int e = 0; e < 4; e++)
                                r[((static_cast<size_t>(t) * NCH1 + j) * 64 + l) * 4 + e] = w1[static_cast<size_t>(32 * t + (l & 31)) * C + 32 * (j / 4) + 8 * (j % 4) + 4 * (l >> 5) + e];
                MemberOff ma, mb;
                ma.w2 = put(r);
                mb.w2 = pack_pw(pb.w2);
                std::vector<float> cb(static_cast<size_t>(bneck_const_floats(C, Cm)), 0.f);
                auto slope = [&](const Node& m, int c) { return m.act == ACT_PRELU ? g.tensors[m.alpha].f32[static_cast<size_t>(c)] : (m.act == ACT_NONE ? 1.f : 0.f); };
                for (int c = 0; c < Cm; c++) {
                    cb[static_cast<size_t>(c)] = pa.b2 >= 0 ? g.tensors[pa.b2].f32[static_cast<size_t>(c)] : 0.f;
                    cb[static_cast<size_t>(Cm + c)] = slope(pa, c);
                    for (int tap = 0; tap < 9; tap++) cb[static_cast<size_t>(2 * Cm + tap * Cm + c)] = g.tensors[pb.w].f32[static_cast<size_t>(tap) * Cm + c];
                    cb[static_cast<size_t>(11 * Cm + c)] = pb.b >= 0 ? g.tensors[pb.b].f32[static_cast<size_t>(c)] : 0.f;
                }
                for (int c = 0; c < C; c++) {
                    cb[static_cast<size_t>(12 * Cm + c)] = pb.b2 >= 0 ? g.tensors[pb.b2].f32[static_cast<size_t>(c)] : 0.f;
                    cb[static_cast<size_t>(12 * Cm + C + c)] = slope(pb, c);
                }
                ma.strip = put(cb);
                // 32-pixel-wide pairs also get the constants of the operand-layout kernel (mdblock_kernels.hip, mbneck_kernel)
                const int Wd = g.tensors[pa.in[0]].shape[2];
                if (mbneck_shape_ok(Wd, C, Cm)) {
                    std::vector<float> mc(static_cast<size_t>(mbneck_consts_floats(Wd, C, Cm)));
                    auto ptr = [&](int t) { return t >= 0 ? g.tensors[t].f32.data() : nullptr; };
                    mbneck_pack_consts(Wd, C, Cm, ptr(pa.w2), ptr(pa.b2), pa.act == ACT_PRELU ? ptr(pa.alpha) : nullptr, pa.act, ptr(pb.w), ptr(pb.b), ptr(pb.w2), ptr(pb.b2),
                                       pb.act == ACT_PRELU ? ptr(pb.alpha) : nullptr, pb.act, mc.data());
                    mb.strip = put(mc);
                }
                chain_off_[i].push_back(ma);
                chain_off_[i].push_back(mb);
            }
            res_wblk_[i].clear();
            res_cblob_[i].clear();
            continue;
        }
        if (n.kind == Node::Resident && n.tail) {
            // tail_kernels.hip: per stage the A operands of v_mfma_f32_16x16x4_f32 — lane (k-quarter kq = l / 16, row m = l % 16) holds, for
            // float4 step i, W[16 tile + m][kq * Kv / 4 + 4 i .. + 3] ([O][KH][KW][I] read as [O][Kv]) — and the small constants:
            // [bias][slope] padded to whole tiles, then for depthwise stages the taps [9][Kv] and the depthwise bias [Kv]
            tail_wa_[i].assign(n.stages.size(), -1);
            tail_wc_[i].assign(n.stages.size(), -1);
            for (size_t k = 0; k < n.stages.size(); k++) {
                const Node::Stage& sg = n.stages[k];
                const TailStage& st = sg.tst;
                if (st.kind == TAIL_LOAD) continue;
                const Node& m = n.members[static_cast<size_t>(sg.member)];
                const auto& wsrc = g.tensors[m.kind == Node::Conv ? m.w : m.w2].f32;
                const int O = st.Co, Kv = st.Kv, nct = (O + 15) / 16, n4 = Kv / 16, K4 = Kv / 4;
                if (wsrc.size() != static_cast<size_t>(O) * Kv) throw std::runtime_error("engine: tail stage weights do not match its shape");
                std::vector<float> r(static_cast<size_t>(nct) * n4 * 256, 0.f);
                for (int ct = 0; ct < nct; ct++)
                    for (int q = 0; q < n4; q++)
                        for (int l = 0; l < 64; l++)
                            for (int e = 0; e < 4; e++) {
                                const int o = 16 * ct + (l & 15), c = (l >> 4) * K4 + 4 * q + e;
                                if (o < O) r[((static_cast<size_t>(ct) * n4 + q) * 64 + l) * 4 + e] = wsrc[static_cast<size_t>(o) * Kv + c];
                            }
                tail_wa_[i][k] = put(r);
                const int bias_t = m.kind == Node::Conv ? m.b : m.b2;
                std::vector<float> cb(static_cast<size_t>(32 * nct) + (st.kind == TAIL_DW ? static_cast<size_t>(10) * Kv : 0), 0.f);
                for (int c = 0; c < O; c++) {
                    cb[static_cast<size_t>(c)] = bias_t >= 0 ? g.tensors[bias_t].f32[static_cast<size_t>(c)] : 0.f;
                    cb[static_cast<size_t>(16 * nct + c)] = m.act == ACT_PRELU ? g.tensors[m.alpha].f32[static_cast<size_t>(c)] : (m.act == ACT_NONE ? 1.f : 0.f);
                }
                if (st.kind == TAIL_DW) {
                    const auto& wd = g.tensors[m.w].f32;  // [3][3][C]
                    for (int t = 0; t < 9 * Kv; t++) cb[static_cast<size_t>(32 * nct + t)] = wd[static_cast<size_t>(t)];
                    if (m.b >= 0)
                        for (int c = 0; c < Kv; c++) cb[static_cast<size_t>(32 * nct + 9 * Kv + c)] = g.tensors[m.b].f32[static_cast<size_t>(c)];
                }
                tail_wc_[i][k] = put(cb);
            }
            continue;
        }
        if (n.kind == Node::Resident) {
            for (const Node& m : n.members) {
                MemberOff mo;
                // pointwise / k x k stride-k weights in A-fragment order (the k x k ones over the virtual channels)
                mo.w2 = pack_pw(m.kind == Node::Conv ? m.w : m.w2);
                chain_off_[i].push_back(mo);
            }
            // per stage: the small constants, padded the way the kernel copies them to LDS
            res_wblk_[i].assign(n.stages.size(), -1);
            for (size_t k = 0; k < n.stages.size(); k++)
                if (n.stages[k].st.kblk) {
                    const Node& m = n.members[static_cast<size_t>(n.stages[k].member)];
                    res_wblk_[i][k] = pack_pw(m.kind == Node::Conv ? m.w : m.w2, n.stages[k].st.kblk);
                }
            res_cblob_[i].assign(n.stages.size(), -1);
            for (size_t k = 0; k < n.stages.size(); k++) {
                const Node::Stage& sg = n.stages[k];
                if (sg.st.kind == RES_STAGE_LOAD) continue;
                const Node& m = n.members[static_cast<size_t>(sg.member)];
                const ResStage& st = sg.st;
                const int Cp = (st.Kv + 7) & ~7, Cop = (st.Co + 31) / 32 * 32;
                std::vector<float> cb(static_cast<size_t>(resident_const_floats(st)), 0.f);
                size_t o = 0;
                const int bias_t = m.kind == Node::Conv ? m.b : m.b2;
                if (st.kind == RES_STAGE_DW) {
                    const auto& wd = g.tensors[m.w].f32;  // [3][3][C]
                    for (int tap = 0; tap < 9; tap++)
                        for (int c = 0; c < st.Kv; c++) cb[static_cast<size_t>(tap) * Cp + c] = wd[static_cast<size_t>(tap) * st.Kv + c];
                    if (m.b >= 0)
                        for (int c = 0; c < st.Kv; c++) cb[static_cast<size_t>(9) * Cp + c] = g.tensors[m.b].f32[c];
                    o = static_cast<size_t>(10) * Cp;
                }
                for (int c = 0; c < st.Co; c++) {
                    cb[o + c] = bias_t >= 0 ? g.tensors[bias_t].f32[c] : 0.f;
                    cb[o + Cop + c] = m.act == ACT_PRELU ? g.tensors[m.alpha].f32[c] : (m.act == ACT_NONE ? 1.f : 0.f);
                }
                res_cblob_[i][k] = put(cb);
            }
            continue;
        }
        if (n.b >= 0) node_b_[i] = put(g.tensors[n.b].f32);
        if (n.b2 >= 0) node_b2_[i] = put(g.tensors[n.b2].f32);
        if (n.alpha >= 0) node_alpha_[i] = put(g.tensors[n.alpha].f32);
        if (n.kind == Node::Conv && n.gemm_head) {
            node_w_[i] = put(g.tensors[n.w].f32);  // [O][KH*KW*I] as stored: the GEMM's W[N][K]
        } else if (n.kind == Node::Conv) {
            const auto& ws = g.tensors[n.w].shape;  // [O][KH][KW][I]
            const auto& src = g.tensors[n.w].f32;
            int O = ws[0], KH = ws[1], KW = ws[2], I = ws[3], Cop = (O + 3) & ~3;
            std::vector<float> r(static_cast<size_t>(KH) * KW * I * Cop, 0.f);
            for (int o = 0; o < O; o++)
                for (int ky = 0; ky < KH; ky++)
                    for (int kx = 0; kx < KW; kx++)
                        for (int c = 0; c < I; c++)
                            r[((static_cast<size_t>(ky) * KW + kx) * I + c) * Cop + o] = src[((static_cast<size_t>(o) * KH + ky) * KW + kx) * I + c];
            node_w_[i] = put(r);
            // the face mesh's first convolution can run inside the launch of the block pair behind it (mdblock_kernels.hip, MD::STEM)
            const auto& sxi = g.tensors[n.in[0]].shape;
            const auto& sxo = g.tensors[n.out].shape;
            if (sxi.size() == 4 && sxo.size() == 4 && n.padding == Padding::Same && n.ept < 0 && n.res < 0 && n.in[0] == g.inputs[0] &&
                (n.act == ACT_NONE || n.act == ACT_RELU || n.act == ACT_RELU6 || n.act == ACT_PRELU) &&
                mdblock_stem_shape_ok(sxi[1], sxi[2], sxi[3], n.KH, n.KW, n.sh, n.sw, sxo[1], sxo[2], sxo[3]) && !(n.act == ACT_PRELU && n.alpha < 0) &&
                [&] {   // its output is read by the next node alone (the pair: as input and as the first block's skip) and is no graph output
                    int readers = 0;
                    for (const Node& m : plan_.nodes) {
                        for (int t : m.in) readers += t == n.out ? 1 : 0;
                        if (m.res == n.out) readers++;
                    }
                    bool is_out = false;
                    for (int t : g.outputs) is_out = is_out || plan_.storage[t].root == plan_.storage[n.out].root;
                    return readers == 1 && !is_out && i + 1 < NN && plan_.nodes[i + 1].kind == Node::Chain && plan_.nodes[i + 1].members.size() == 2 &&
                           plan_.nodes[i + 1].in.size() >= 1 && plan_.nodes[i + 1].in[0] == n.out;
                }()) {
                std::vector<float> sc(static_cast<size_t>(mdblock_stem_consts_floats()));
                mdblock_pack_stem(src.data(), n.b >= 0 ? g.tensors[n.b].f32.data() : nullptr, n.act == ACT_PRELU && n.alpha >= 0 ? g.tensors[n.alpha].f32.data() : nullptr, n.act, sc.data());
                node_stem_[i] = put(sc);
            }
        } else if (n.kind == Node::Dw) {
            node_w_[i] = put(g.tensors[n.w].f32);
        } else if (n.kind == Node::Block) {
            if (n.w >= 0) node_w_[i] = put(g.tensors[n.w].f32);
            node_w2_[i] = pack_pw(n.w2);
            const auto& ws = g.tensors[n.w2].shape;  // [O][1][1][I]
            if (n.w >= 0 && n.sh == 1 && n.sw == 1 && n.padding == Padding::Same && strip_shape_ok(ws[3], ws[0])) {
                std::vector<float> sc(static_cast<size_t>(strip_consts_floats(ws[3])));
                strip_pack_consts(ws[3], g.tensors[n.w].f32.data(), n.b >= 0 ? g.tensors[n.b].f32.data() : nullptr, g.tensors[n.w2].f32.data(),
                                  n.b2 >= 0 ? g.tensors[n.b2].f32.data() : nullptr, n.alpha >= 0 ? g.tensors[n.alpha].f32.data() : nullptr, n.act, sc.data());
                node_strip_[i] = put(sc);
            } else if (n.w >= 0 && n.sh == 1 && n.sw == 1 && n.padding == Padding::Same && mstrip_shape_ok(ws[3], ws[0]) && g.tensors[n.out].shape[2] == 32) {
                std::vector<float> sc(static_cast<size_t>(mstrip_consts_floats(ws[3])));
                mstrip_pack_consts(ws[3], g.tensors[n.w].f32.data(), n.b >= 0 ? g.tensors[n.b].f32.data() : nullptr, g.tensors[n.w2].f32.data(),
                                   n.b2 >= 0 ? g.tensors[n.b2].f32.data() : nullptr, n.alpha >= 0 ? g.tensors[n.alpha].f32.data() : nullptr, n.act, sc.data());
                node_strip_[i] = put(sc);
            }
            const int Wn = g.tensors[n.out].shape.size() == 4 ? g.tensors[n.out].shape[2] : 0;
            if (n.w >= 0 && n.sh == 1 && n.sw == 1 && n.padding == Padding::Same && mwalk_shape_ok(Wn, ws[3], ws[0])) {
                std::vector<float> sc(static_cast<size_t>(mwalk_consts_floats(Wn, ws[3], ws[0])));
                mwalk_pack_consts(Wn, ws[3], ws[0], g.tensors[n.w].f32.data(), n.b >= 0 ? g.tensors[n.b].f32.data() : nullptr, g.tensors[n.w2].f32.data(),
                                  n.b2 >= 0 ? g.tensors[n.b2].f32.data() : nullptr, n.alpha >= 0 ? g.tensors[n.alpha].f32.data() : nullptr, n.act, sc.data());
                node_mwalk_[i] = put(sc);
            }
            // stride-2 blocks with an operand-layout form (ms2_kernels.hip; the same slot: a node is one or the other)
            const auto& sin = g.tensors[n.in[0]].shape;
            const bool pool_skip = n.res >= 0 && n.res == n.in[0] && n.res_mode == RES_MAXPOOL && !n.res_after;
            if (n.w >= 0 && n.sh == 2 && n.sw == 2 && n.padding == Padding::Same && n.ept < 0 && sin.size() == 4 && (n.res < 0 || pool_skip) &&
                ms2_shape_ok(sin[2], ws[3], ws[0], pool_skip)) {
                std::vector<float> sc(static_cast<size_t>(ms2_consts_floats(sin[2], ws[3], ws[0], pool_skip)));
                ms2_pack_consts(sin[2], ws[3], ws[0], pool_skip, g.tensors[n.w].f32.data(), n.b >= 0 ? g.tensors[n.b].f32.data() : nullptr, g.tensors[n.w2].f32.data(),
                                n.b2 >= 0 ? g.tensors[n.b2].f32.data() : nullptr, n.alpha >= 0 ? g.tensors[n.alpha].f32.data() : nullptr, n.act, sc.data());
                node_mwalk_[i] = put(sc);
            }
        }
    }
    // two plain BlazeBlocks in a row on a layer mdblock_kernels.hip has a pair form for (the face mesh's 48x48x32 blocks, round 6): the constants of the pair
    // launch, kept at the first node; which form runs is decided per launch
    for (size_t i = 0; i + 1 < NN; i++) {
        const Node &pa = plan_.nodes[i], &pb = plan_.nodes[i + 1];
        auto plain = [&](const Node& m) {
            return m.kind == Node::Block && m.w >= 0 && m.in.size() == 1 && m.sh == 1 && m.sw == 1 && m.res == m.in[0] && m.res_mode == RES_DIRECT && !m.res_after &&
                   m.padding == Padding::Same && m.ept < 0 && (m.act != ACT_PRELU || m.alpha >= 0);
        };
        if (!plain(pa) || !plain(pb) || pb.in[0] != pa.out) continue;
        const auto& sx = g.tensors[pa.in[0]].shape;
        if (sx.size() != 4 || g.tensors[pa.out].shape != sx || g.tensors[pb.out].shape != sx || !mdblock_shape_ok(sx[2], sx[3], sx[3], sx[3], true)) continue;
        // the tensor between the two is never written by that launch: nobody else may read it
        bool only_reader = true;
        for (size_t j = 0; j < NN && only_reader; j++) {
            if (j == i + 1) continue;
            for (int t : plan_.nodes[j].in) only_reader = only_reader && t != pa.out;
            only_reader = only_reader && plan_.nodes[j].res != pa.out;
        }
        for (int t : g.outputs) only_reader = only_reader && plan_.storage[t].root != plan_.storage[pa.out].root;
        if (!only_reader) continue;
        std::vector<float> mc(static_cast<size_t>(mdblock_consts_floats(sx[2], sx[3], sx[3], sx[3], true)));
        auto ptr = [&](int t) { return t >= 0 ? g.tensors[t].f32.data() : nullptr; };
        mdblock_pack_consts(sx[2], sx[3], sx[3], sx[3], ptr(pa.w), ptr(pa.b), ptr(pa.w2), ptr(pa.b2), pa.act == ACT_PRELU ? ptr(pa.alpha) : nullptr, pa.act,
                            ptr(pb.w), ptr(pb.b), ptr(pb.w2), ptr(pb.b2), pb.act == ACT_PRELU ? ptr(pb.alpha) : nullptr, pb.act, mc.data(), true);
        node_pair_[i] = put(mc);
    }
    host.resize(host.size() + 4096, 0.f);  // slack: the stage programs' A-fragment prefetch walks up to 8 KiB past a tile's last chunk
    if (d_weights_) hip_check(hipFree(d_weights_), "hipFree");
    d_weights_ = nullptr;
    hip_check(hipMalloc(reinterpret_cast<void**>(&d_weights_), std::max<size_t>(host.size(), 64) * sizeof(float)), "hipMalloc weights");
    hip_check(hipMemcpy(d_weights_, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice), "upload weights");
    // ---- stage programs of the resident launches: pointer-free descriptors, resolved against ResBases at launch
    {
        std::vector<ResStage> progs;
        node_prog_.assign(NN, -1);
        auto ref = [&](int t) {
            ResRef r;
            const Storage& st = plan_.storage[t];
            r.inner = st.offset;
            r.fs = st.frame_stride;
            if (st.root == plan_.storage[g.inputs[0]].root) { r.base = 1; return r; }
            for (size_t k = 0; k < g.outputs.size(); k++)
                if (plan_.storage[g.outputs[k]].root == st.root) {
                    if (2 + k >= static_cast<size_t>(kResBases)) throw std::runtime_error("stage program: too many graph outputs");
                    r.base = 2 + static_cast<int>(k);
                    return r;
                }
            if (plan_.root_offset[st.root] < 0) throw std::runtime_error("stage program: tensor has no storage");
            r.base = 0;
            r.root_off = plan_.root_offset[st.root];
            return r;
        };
        std::vector<TailStage> tprogs;
        for (size_t i = 0; i < NN; i++) {
            const Node& n = plan_.nodes[i];
            if (n.kind != Node::Resident || !n.tail) continue;
            node_prog_[i] = static_cast<long>(tprogs.size());
            for (size_t k = 0; k < n.stages.size(); k++) {
                const Node::Stage& sg = n.stages[k];
                TailStage st = sg.tst;
                if (sg.src_t >= 0) st.src_g = ref(sg.src_t);
                if (st.kind != TAIL_LOAD) {
                    if (sg.dst_t >= 0) st.dst_g = ref(sg.dst_t);
                    if (sg.res_t >= 0) st.res_g = ref(sg.res_t);
                    st.w_a = tail_wa_[i][k];
                    st.w_c = tail_wc_[i][k];
                }
                tprogs.push_back(st);
            }
        }
        if (d_tail_programs_) hip_check(hipFree(d_tail_programs_), "hipFree");
        d_tail_programs_ = nullptr;
        if (!tprogs.empty()) {
            hip_check(hipMalloc(reinterpret_cast<void**>(&d_tail_programs_), tprogs.size() * sizeof(TailStage)), "hipMalloc programs");
            hip_check(hipMemcpy(d_tail_programs_, tprogs.data(), tprogs.size() * sizeof(TailStage), hipMemcpyHostToDevice), "upload programs");
        }
        for (size_t i = 0; i < NN; i++) {
            const Node& n = plan_.nodes[i];
            if (n.kind != Node::Resident || n.tail) continue;
            node_prog_[i] = static_cast<long>(progs.size());
            for (const Node::Stage& sg : n.stages) {
                ResStage st = sg.st;
                if (sg.src_t >= 0) st.src_g = ref(sg.src_t);
                if (st.kind != RES_STAGE_LOAD) {
                    if (sg.dst_t >= 0) st.dst_g = ref(sg.dst_t);
                    if (sg.res_t >= 0) st.res_g = ref(sg.res_t);
                    const size_t k = static_cast<size_t>(&sg - n.stages.data());
                    st.w_pw = st.kblk ? res_wblk_[i][k] : chain_off_[i][static_cast<size_t>(sg.member)].w2;
                    st.cblob = res_cblob_[i][k];
                }
                progs.push_back(st);
            }
        }
        if (d_programs_) hip_check(hipFree(d_programs_), "hipFree");
        d_programs_ = nullptr;
        if (!progs.empty()) {
            hip_check(hipMalloc(reinterpret_cast<void**>(&d_programs_), progs.size() * sizeof(ResStage)), "hipMalloc programs");
            hip_check(hipMemcpy(d_programs_, progs.data(), progs.size() * sizeof(ResStage), hipMemcpyHostToDevice), "upload programs");
        }
    }
    // ---- output heads that may run beside the trunk: compute nodes whose output lives in a graph-output buffer and is
    // read by no other launch, and that come after the last trunk node in plan order (so no later trunk launch can
    // re-use arena memory they still read: the arena's liveness analysis follows plan order)
    {
        const size_t N = plan_.nodes.size();
        head_slot_.assign(N, -1); head_wait_.assign(N, -1); event_after_.assign(N, 0);
        auto is_view = [&](const Node& n) { return n.kind == Node::Reshape || n.kind == Node::Concat; };
        auto root = [&](int t) { return plan_.storage[t].root; };
        std::vector<int> out_roots;
        for (int o : g.outputs) out_roots.push_back(root(o));
        std::vector<char> head(N, 0);
        int last_trunk = -1;
        for (size_t i = 0; i < N; i++) {
            const Node& n = plan_.nodes[i];
            if (is_view(n)) continue;
            bool feeds_output = std::find(out_roots.begin(), out_roots.end(), root(n.out)) != out_roots.end();
            for (size_t j = 0; j < N && feeds_output; j++) {
                const Node& m = plan_.nodes[j];
                if (is_view(m)) continue;
                for (int x : m.in) if (x == n.out) feeds_output = false;
                if (m.res == n.out) feeds_output = false;
            }
            if (n.kind == Node::Resident && (n.in.size() != 1 || !n.extra_out.empty())) feeds_output = false;  // several inputs / outputs: stays on the trunk
            head[i] = feeds_output;
            if (!feeds_output) last_trunk = static_cast<int>(i);
        }
        std::vector<int> slot_of_producer(N + 1, -1);
        int slots = 0;
        for (size_t i = 0; i < N; i++) {
            const Node& n = plan_.nodes[i];
            if (!head[i] || static_cast<int>(i) < last_trunk || is_view(n)) continue;
            int prod = -1;  // the launch that produces (the buffer of) its input
            for (size_t j = 0; j < i; j++) {
                if (is_view(plan_.nodes[j])) continue;
                bool makes = root(plan_.nodes[j].out) == root(n.in[0]);
                for (int t : plan_.nodes[j].extra_out) makes |= root(t) == root(n.in[0]);
                if (makes) prod = static_cast<int>(j);
            }
            if (n.res >= 0 && n.res != n.in[0]) continue;  // two producers: keep it on the trunk
            // heads are spread round robin over `heads` side streams (option, default 1).  They are independent of each other, but
            // on BackCamera (four small heads behind the last trunk launch) every extra parallel branch of the replay graph cost
            // more than it hid: 1.675 ms per step with 1 stream, 1.69 / 1.68 / 1.72 with 2 / 3 / 4
            (void)slot_of_producer;
            head_slot_[i] = slots++ % std::max(1, std::min(head_streams_opt_, kHeadStreams));
            head_wait_[i] = prod;
            if (prod >= 0) event_after_[prod] = 1;
        }
        // tail branches (Plan::branch): chain 0 stays on the trunk stream, every other chain runs on a side stream behind the launch
        // that produced its newest input (the trunk runs in plan order, so the older inputs are done by then; launches of the same
        // chain share a stream).  The arena keeps everything the branches touch allocated to the end of the plan.
        static const bool no_branches = getenv("MI_NO_BRANCHES") != nullptr;  // development aid
        for (size_t i = 0; i < N && !no_branches && plan_.branch.size() == N; i++) {
            const Node& n = plan_.nodes[i];
            if (plan_.branch[i] < 1 || is_view(n)) continue;
            int prod = -1;
            std::vector<int> srcs = n.in;
            if (n.res >= 0) srcs.push_back(n.res);
            for (int t : srcs)
                for (size_t j = 0; j < i; j++) {
                    if (is_view(plan_.nodes[j])) continue;
                    bool makes = root(plan_.nodes[j].out) == root(t);
                    for (int x : plan_.nodes[j].extra_out) makes |= root(x) == root(t);
                    if (makes) prod = std::max(prod, static_cast<int>(j));
                }
            head_slot_[i] = (plan_.branch[i] - 1) % std::max(1, std::min(head_streams_opt_, kHeadStreams));
            head_wait_[i] = prod;
            if (prod >= 0) event_after_[static_cast<size_t>(prod)] = 1;
        }
    }
    build_bandnet();
    dirty_ = false;
    chunk_cap_ = 0;
}

void Model::free_bandnet() {
    if (d_band_prog_) hipFree(d_band_prog_);
    if (d_band_consts_) hipFree(d_band_consts_);
    if (d_band_ws_) hipFree(d_band_ws_);
    if (d_band_sync_) hipFree(d_band_sync_);
    if (h_band_fail_) hipHostFree(h_band_fail_);
    d_band_prog_ = nullptr; d_band_consts_ = nullptr; d_band_ws_ = nullptr; d_band_sync_ = nullptr; h_band_fail_ = nullptr; d_band_fail_ = nullptr;
    band_ready_ = false;
}

// The single-launch plan (bandnet_kernels.hip).  It is made from the level-2 lowering of the same graph — one node per BlazeBlock /
// convolution — when that is: a first convolution (which keeps its launch of the batched plan), then nothing but 3x3 BlazeBlocks whose
// skip is their own input (or, without a stride, another tensor of the program: the iris network's bottlenecks), pointwise blocks,
// 1x1 convolutions and 2x2 stride-2 convolutions, each reading the tensor of an earlier one.  The program stops in front of the first node
// that is none of these (the face mesh's and the iris network's two whole-frame convolutions): the plan nodes from there on keep their
// launches of the batched plan, behind the band launch, and the tensors they read are written to their arena storage by the band
// program — provided no launch of the batched plan straddles the cut.  Graphs whose FIRST block is already something else (full_range's
// double blocks) leave band_ready_ false and the handle on the batched plan.
// MI_BAND_DEBUG=1 names the line at which a graph was found to have no single-launch form
#define BAND_GIVE_UP do { if (std::getenv("MI_BAND_DEBUG")) std::fprintf(stderr, "bandnet: no single-launch plan (engine.cpp:%d)\n", __LINE__); return; } while (0)
// (round 5) 2x2 stride-2 convolutions and the blocks behind them, whose skip is the 2x2 max of the convolution's input,
// are stages too: the whole iris network but its two whole-frame heads is one program (iris_landmark.rs:203).  Should a graph with such
// nodes have no program with them, it is lowered once more with the program ending in front of the first of them (the earlier form).
void Model::build_bandnet() {
    band_saw_conv2_ = false;
    build_bandnet_try(true);
    if (!band_ready_ && band_saw_conv2_) build_bandnet_try(false);
}

void Model::build_bandnet_try(bool conv2_ok) {
    free_bandnet();
    if (!band_ || fuse_level_ < 2) BAND_GIVE_UP;
    const Plan p2 = build_plan(parse_tflite(blob_.data(), blob_.size()), 2);
    const Graph& g = p2.graph;
    auto is_view = [](const Node& n) { return n.kind == Node::Reshape || n.kind == Node::Concat; };
    // the first launch of both plans must be the same convolution
    size_t i5 = 0, i2 = 0;
    while (i5 < plan_.nodes.size() && is_view(plan_.nodes[i5])) i5++;
    while (i2 < p2.nodes.size() && is_view(p2.nodes[i2])) i2++;
    if (i5 >= plan_.nodes.size() || i2 >= p2.nodes.size()) BAND_GIVE_UP;
    const Node &stem5 = plan_.nodes[i5], &stem2 = p2.nodes[i2];
    if (stem5.kind != Node::Conv || stem2.kind != Node::Conv || stem5.out != stem2.out || stem5.gemm_head) BAND_GIVE_UP;
    band_stem_out_ = stem2.out;
    if (plan_.storage[band_stem_out_].root != band_stem_out_ || plan_.storage[band_stem_out_].offset != 0) BAND_GIVE_UP;
    band_first_ = static_cast<int>(i5) + 1;
    while (band_first_ < static_cast<int>(plan_.nodes.size()) && is_view(plan_.nodes[static_cast<size_t>(band_first_)])) band_first_++;
    if (band_first_ >= static_cast<int>(plan_.nodes.size())) BAND_GIVE_UP;
    // workgroups per frame: one per row of the first tensor, at most band_nw_ (the bands of the later, smaller tensors are one row of every
    // 2nd, 4th ... workgroup)
    const auto& stem_shape = g.tensors[band_stem_out_].shape;
    if (stem_shape.size() != 4) BAND_GIVE_UP;
    int NW = band_nw_;
    while (NW > 1 && (stem_shape[1] % NW) && (NW % stem_shape[1])) NW--;
    if (stem_shape[1] < NW) NW = stem_shape[1];
    band_nw_used_ = NW;
    band_max_frames_ = std::max(0, device_cu_count() / NW);
    if (band_max_frames_ < 1) BAND_GIVE_UP;

    std::vector<BandStage> prog;
    std::vector<const Node*> nodes;
    std::vector<float> consts;
    std::vector<int> producer(g.tensors.size(), -1);
    std::vector<int> out_root;
    for (int o : g.outputs) out_root.push_back(p2.storage[o].root);
    band_ext_.clear();
    auto ext_slot = [&](int out_k, int tensor) {   // BandLaunch::base index of a graph output / an arena tensor (-1: no slot left)
        for (size_t j = 0; j < band_ext_.size(); j++)
            if (band_ext_[j].out_k == out_k && band_ext_[j].tensor == tensor) return 2 + static_cast<int>(j);
        if (2 + band_ext_.size() >= static_cast<size_t>(kBandBases)) return -1;
        BandExt e; e.out_k = out_k; e.tensor = tensor;
        band_ext_.push_back(e);
        return 1 + static_cast<int>(band_ext_.size());
    };
    std::vector<char> band_op(g.ops.size(), 0);   // .tflite operators the band program computes
    size_t cut = p2.nodes.size();                 // first p2 node the program does not take
    auto put = [&](const std::vector<float>& v) {
        const long off = align_up(static_cast<long>(consts.size()), 64);
        consts.resize(static_cast<size_t>(off) + v.size(), 0.f);
        std::copy(v.begin(), v.end(), consts.begin() + off);
        return off;
    };
    // ---- pass 1: one stage per node, its shape, bands, source and constants
    for (size_t i = i2 + 1; i < p2.nodes.size(); i++) {
        const Node& n = p2.nodes[i];
        if (is_view(n)) continue;
        const bool pw_block = n.kind == Node::Block && n.w < 0;
        const bool dw_block = n.kind == Node::Block && n.w >= 0;
        const bool conv1 = n.kind == Node::Conv && n.KH == 1 && n.KW == 1 && n.sh == 1 && n.sw == 1 && !n.gemm_head;
        const bool conv2 = conv2_ok && n.kind == Node::Conv && n.KH == 2 && n.KW == 2 && n.sh == 2 && n.sw == 2 && !n.gemm_head && n.in.size() == 1 &&
                           g.tensors[n.in[0]].shape.size() == 4 && g.tensors[n.in[0]].shape[1] % 2 == 0 && g.tensors[n.in[0]].shape[2] % 2 == 0 &&
                           g.tensors[n.in[0]].shape[3] % 32 == 0 && n.in[0] != band_stem_out_;
        if (std::getenv("MI_BAND_DEBUG"))
            std::fprintf(stderr, "bandnet: node %zu kind %d K %dx%d s %d in %zu res %d mode %d after %d ept %d act %d gemm %d\n", i, static_cast<int>(n.kind), n.KH, n.KW, n.sh, n.in.size(), n.res, n.res_mode,
                         n.res_after ? 1 : 0, n.ept, n.act, n.gemm_head ? 1 : 0);
        if (!pw_block && !dw_block && !conv1 && !conv2) { cut = i; break; }
        if (conv2) band_saw_conv2_ = true;
        // full_range's lateral convolutions — a 1x1 convolution with a fused activation, then ADD with the bilinearly up-sampled coarser map: the skip
        // joins BEHIND the activation — are stages of the WIDE instantiation when the coarse tensor is the program's (round 6); any other such node ends the program
        const bool up2x = (conv1 || pw_block) && n.res_after && n.res >= 0 && n.res_mode == RES_UP2X && band_wide_ok_ && n.in.size() == 1 && n.ept < 0 &&
                          producer[static_cast<size_t>(n.res)] >= 0 && g.tensors[n.res].shape.size() == 4;
        // (full_range_sparse pads its stride-2 blocks explicitly, one pixel in front: the same stage with its window one row / column earlier — round 6)
        const bool pre = dw_block && n.ept == 1 && n.epl == 1 && n.sh == 2 && n.sw == 2 && n.padding == Padding::Valid && n.res < 0 && band_wide_ok_;
        if (n.in.size() != 1 || (n.ept >= 0 && !pre) || (n.res_after && !up2x)) {
            if (prog.empty()) BAND_GIVE_UP;
            cut = i;
            break;
        }
        const auto& si = g.tensors[n.in[0]].shape;
        const auto& so = g.tensors[n.out].shape;
        if (si.size() != 4 || so.size() != 4) BAND_GIVE_UP;
        BandStage st;
        st.kind = dw_block ? BAND_BLOCK : BAND_PW;
        st.H = si[1]; st.W = si[2]; st.C = si[3]; st.Ho = so[1]; st.Wo = so[2]; st.Co = so[3];
        const bool wide_ok = band_wide_ok_ && !conv2 && st.C <= 384 && st.Co <= 384 && (st.C <= 128 || st.C % 16 == 0) && (st.Co <= 128 || st.Co % 16 == 0);   // (the WIDE instantiation: round 6)
        if (st.C % 4 || st.C < 8 || st.Co < 1 || ((st.C > 128 || st.Co > 128) && !wide_ok)) {
            // channel counts the kernel does not take (a wave keeps ONE 16-channel output tile and at most eight 16-value chunks of A operands): the
            // program ends in front of this node when it has stages already (full_range: the trunk down to 12x12x36; round 6), else there is none
            if (prog.empty()) BAND_GIVE_UP;
            cut = i;
            break;
        }
        st.S = 1;
        if (dw_block) {
            if (n.KH != 3 || n.KW != 3 || n.sh != n.sw || (n.sh != 1 && n.sh != 2) || (n.padding != Padding::Same && !pre)) BAND_GIVE_UP;
            st.S = n.sh;
            st.pre = pre ? 1 : 0;
            if (st.S == 2 && ((st.H & 1) || (st.W & 1))) BAND_GIVE_UP;
            if (st.Ho != st.H / st.S || st.Wo != st.W / st.S) BAND_GIVE_UP;
        } else if (conv2) {
            st.S = 2;   // (even sizes: SAME and VALID are the same window)
            if (st.Ho != st.H / 2 || st.Wo != st.W / 2 || n.res >= 0) BAND_GIVE_UP;
        } else if (st.Ho != st.H || st.Wo != st.W) {
            BAND_GIVE_UP;
        }
        st.act = n.act;
        if (n.act != ACT_NONE && n.act != ACT_RELU && n.act != ACT_RELU6 && n.act != ACT_PRELU) BAND_GIVE_UP;
        st.res_mode = RES_NONE;
        if (up2x) {
            const auto& sr = g.tensors[n.res].shape;
            const BandStage& cd = prog[static_cast<size_t>(producer[static_cast<size_t>(n.res)])];
            if (sr[1] * 2 != st.Ho || sr[2] * 2 != st.Wo || sr[3] != st.Co || (st.Co & 3) || cd.R != 1 || (st.Wo / 2) * (st.Co / 4) > 512) { if (prog.empty()) BAND_GIVE_UP; cut = i; break; }
            st.res_mode = RES_UP2X;
            st.res_dep = producer[static_cast<size_t>(n.res)];   // (its rows come through the packets, not from a tile: pass 2)
            st.res_c = st.Co;
        } else if (n.res >= 0) {
            if (!dw_block) BAND_GIVE_UP;
            if (n.res != n.in[0] && n.res_mode == RES_MAXPOOL) {
                // the skip is the 2x2 max of the tensor the 2x2 convolution in front of this block read: the rows 2r, 2r + 1 of it that the
                // owner of output row r needs are in the LDS tile that convolution read them from
                const auto& sr = g.tensors[n.res].shape;
                const int d = producer[static_cast<size_t>(n.in[0])];
                // (round 6: ... or the stride-2 BLOCK in front — full_range's down-sampling pairs: DW s2 + PW reduce, then DW + PW expand + 2x2 max of the
                // pair's input, zero-padded from its res_c channels to Co)
                if (st.S != 1 || sr.size() != 4 || sr[1] != 2 * st.Ho || sr[2] != 2 * st.Wo || sr[3] > st.Co || (sr[3] & 3) || d < 0) BAND_GIVE_UP;
                const BandStage& cv = prog[static_cast<size_t>(d)];
                if (cv.S != 2 || cv.pre || cv.dep < 0 || cv.dep != producer[static_cast<size_t>(n.res)]) BAND_GIVE_UP;
                st.res_dep = cv.dep;
                st.res_mode = RES_MAXPOOL;
                st.res_c = sr[3];
            }
            else if (n.res != n.in[0]) {
                // the skip is another tensor of the program, with the output's shape (its rows then have the output's owners)
                const auto& sr = g.tensors[n.res].shape;
                if (n.res_mode != RES_DIRECT || st.S != 1 || sr.size() != 4 || sr[1] != st.Ho || sr[2] != st.Wo || sr[3] > st.Co || (sr[3] & 3)) BAND_GIVE_UP;   // (fewer channels than Co: zero-padded)
                st.res_c = sr[3];
                if (n.res == band_stem_out_) st.res_dep = -1;
                else if (producer[static_cast<size_t>(n.res)] >= 0) st.res_dep = producer[static_cast<size_t>(n.res)];
                else BAND_GIVE_UP;
                st.res_mode = RES_DIRECT;
            }
            else if (n.res_mode == RES_DIRECT && st.S == 1 && st.Co >= st.C) st.res_mode = RES_DIRECT;   // (Co > C: the skip is zero-padded to Co channels)
            else if (n.res_mode == RES_MAXPOOL && st.S == 2 && st.Co >= st.C) st.res_mode = RES_MAXPOOL;
            else BAND_GIVE_UP;
        }
        // bands: whole rows per workgroup while there are at least NW rows, one row for every (NW / rows)-th workgroup below that
        auto log2_exact = [](int v) { int k = 0; while ((1 << k) < v) k++; return (1 << k) == v ? k : -1; };
        if (st.Ho >= NW) {
            if (st.Ho % NW) BAND_GIVE_UP;
            st.R = st.Ho / NW; st.wshift = 0; st.nbands = NW;
        } else {
            if (NW % st.Ho || log2_exact(NW / st.Ho) < 0) BAND_GIVE_UP;
            st.R = 1; st.wshift = log2_exact(NW / st.Ho); st.nbands = st.Ho;
        }
        if (n.in[0] == band_stem_out_) {
            st.src_base = 1; st.src_off = 0; st.dep = -1; st.Rin = 0;
            st.src_fs = plan_.storage[band_stem_out_].frame_stride;
            if (st.src_fs & 3) BAND_GIVE_UP;
        } else {
            const int d = producer[static_cast<size_t>(n.in[0])];
            if (d < 0) BAND_GIVE_UP;
            const BandStage& pd = prog[static_cast<size_t>(d)];
            st.dep = d;
            st.Rin = pd.R;
            // the owner of output row r must own input row S r, and the rows it lacks must be at most one above and two below its own
            for (int b = 0; b < st.nbands; b++) {
                const int r0 = b * st.R, nro = std::min(st.Ho, r0 + st.R) - r0, p0 = st.S * r0;
                if (((p0 / pd.R) << pd.wshift) != (b << st.wshift) || p0 % pd.R) BAND_GIVE_UP;
                const int rin = std::min(pd.R, st.H - p0);
                const int yb = dw_block ? (st.S == 1 ? p0 + nro + 1 : p0 + 2 * nro + 1 - st.pre) : (conv2 ? p0 + 2 * nro : p0 + nro);
                const int below = yb - (p0 + rin);
                if (below < 0 || below > 2) BAND_GIVE_UP;
                if (((dw_block && (st.S == 1 || st.pre) ? 1 : 0) + below) * st.W * (st.C / 4) > 4 * 512) BAND_GIVE_UP;   // the halo rows: four 16-byte elements per lane
            }
        }
        // a plain copy of the output where it is a graph output (through the reshape / concatenation views behind it)
        const Storage& so_st = p2.storage[n.out];
        for (size_t k = 0; k < out_root.size(); k++)
            if (out_root[k] == so_st.root) {
                st.dst_base = ext_slot(static_cast<int>(k), -1); st.dst_off = so_st.offset; st.dst_fs = so_st.frame_stride;
                if (st.dst_base < 0) BAND_GIVE_UP;
            }
        if (st.dst_base < 0 && so_st.root != n.out) BAND_GIVE_UP;
        if (st.dst_base < 0 && st.Co % 4) BAND_GIVE_UP;
        if ((st.dst_off & 3) || (st.dst_fs & 3)) {
            if (st.Co % 4 == 0) BAND_GIVE_UP;   // 16-byte stores need the alignment; the ragged heads store floats
        }
        // constants
        const int wt = n.kind == Node::Conv ? n.w : n.w2, bt = n.kind == Node::Conv ? n.b : n.b2;
        const auto& wsrc = g.tensors[wt].f32;
        const int C = conv2 ? 4 * st.C : st.C;   // (the contraction length: [Co][2][2][C] read as [Co][4 C])
        const int Co = st.Co, nct = (Co + 15) / 16, n16 = C / 16, has8 = (C / 8) & 1, has4 = (C / 4) & 1, per_ct = n16 * 256 + has8 * 128 + has4 * 64;
        if (wsrc.size() != static_cast<size_t>(Co) * C) BAND_GIVE_UP;
        std::vector<float> A(static_cast<size_t>(nct) * per_ct, 0.f);
        for (int ct = 0; ct < nct; ct++)
            for (int l = 0; l < 64; l++) {
                const int o = 16 * ct + (l & 15), kq = l >> 4;
                if (o >= Co) continue;
                for (int j = 0; j < n16; j++)
                    for (int e = 0; e < 4; e++) A[static_cast<size_t>(ct) * per_ct + (j * 64 + l) * 4 + e] = wsrc[static_cast<size_t>(o) * C + 16 * j + 4 * kq + e];
                if (has8)
                    for (int e = 0; e < 2; e++) A[static_cast<size_t>(ct) * per_ct + n16 * 256 + l * 2 + e] = wsrc[static_cast<size_t>(o) * C + 16 * n16 + 2 * kq + e];
                if (has4) A[static_cast<size_t>(ct) * per_ct + n16 * 256 + has8 * 128 + l] = wsrc[static_cast<size_t>(o) * C + 16 * n16 + 8 * has8 + kq];
            }
        st.c_floats = bandnet_const_floats(st);   // (what the kernel stages in LDS: without the depthwise taps where they do not fit — a wide stage reads them from L2)
        st.per_ct = per_ct;
        if (nct > 24 || (nct > 8 && !wide_ok)) BAND_GIVE_UP;
        st.wpc_shift = nct == 1 ? 3 : (nct == 2 ? 2 : (nct <= 4 ? 1 : 0));
        std::vector<float> cb(static_cast<size_t>(32 * nct + (dw_block ? 10 * C : 0)), 0.f);
        for (int c = 0; c < Co; c++) {
            cb[static_cast<size_t>(c)] = bt >= 0 ? g.tensors[bt].f32[static_cast<size_t>(c)] : 0.f;
            cb[static_cast<size_t>(16 * nct + c)] = n.act == ACT_PRELU ? g.tensors[n.alpha].f32[static_cast<size_t>(c)] : (n.act == ACT_NONE ? 1.f : 0.f);
        }
        if (dw_block) {
            const auto& wd = g.tensors[n.w].f32;  // [1][3][3][C]
            if (wd.size() != static_cast<size_t>(9) * C) BAND_GIVE_UP;
            for (int t = 0; t < 9 * C; t++) cb[static_cast<size_t>(32 * nct + t)] = wd[static_cast<size_t>(t)];
            if (n.b >= 0)
                for (int c = 0; c < C; c++) cb[static_cast<size_t>(32 * nct + 9 * C + c)] = g.tensors[n.b].f32[static_cast<size_t>(c)];
        }
        st.w_a = put(A);
        st.w_c = put(cb);
        auto magic = [](int d) { return d <= 1 ? 0u : static_cast<unsigned>((0x100000000ull + static_cast<unsigned long long>(d) - 1) / static_cast<unsigned long long>(d)); };
        st.mC4 = magic(st.C / 4); st.mWo = magic(st.Wo); st.mrowq = magic(st.W * (st.C / 4));
        if (static_cast<long>(st.R + 3) * st.W * (st.C / 4) >= 65536 || st.R * st.Wo * std::max(st.C, st.Co) / 4 >= 65536) BAND_GIVE_UP;   // the magic divisions' range
        producer[static_cast<size_t>(n.out)] = static_cast<int>(prog.size());
        prog.push_back(st);
        nodes.push_back(&n);
        for (int o : n.src_ops) band_op[static_cast<size_t>(o)] = 1;
    }
    if (prog.empty() || prog.size() > 63) BAND_GIVE_UP;
    // ---- the cut: a plan node behind the first convolution either lies wholly inside the band program (the band launch stands for it) or wholly
    // behind it (it keeps its launch); a tensor such a launch reads from the program is written to its arena storage by the producing stage
    band_node_runs_.assign(plan_.nodes.size(), 0);
    if (g.ops.size() != plan_.graph.ops.size() || g.tensors.size() > plan_.storage.size()) BAND_GIVE_UP;
    {
        std::vector<int> op_producer(g.tensors.size(), -1);
        for (size_t o = 0; o < g.ops.size(); o++)
            for (int t : g.ops[o].outputs)
                if (t >= 0) op_producer[static_cast<size_t>(t)] = static_cast<int>(o);
        std::function<void(const Node&, std::vector<int>&)> reads = [&](const Node& n, std::vector<int>& v) {
            for (int t : n.in) v.push_back(t);
            if (n.res >= 0) v.push_back(n.res);
            for (const Node& m : n.members) reads(m, v);
            for (const Node& m : n.head_nodes) reads(m, v);
            for (const Node::Stage& sg : n.stages) { if (sg.src_t >= 0) v.push_back(sg.src_t); if (sg.res_t >= 0) v.push_back(sg.res_t); }
        };
        bool any_band = false;
        for (size_t i = static_cast<size_t>(band_first_); i < plan_.nodes.size(); i++) {
            const Node& n = plan_.nodes[i];
            if (is_view(n)) continue;
            size_t inside = 0;
            for (int o : n.src_ops) inside += band_op[static_cast<size_t>(o)] ? 1 : 0;
            if (n.src_ops.empty() || (inside != 0 && inside != n.src_ops.size())) BAND_GIVE_UP;   // a launch of the batched plan straddles the cut
            if (inside) { any_band = true; continue; }
            band_node_runs_[i] = 1;
            std::vector<int> rd;
            reads(n, rd);
            for (int t : rd) {
                const int o = op_producer[static_cast<size_t>(t)];
                if (o < 0 || !band_op[static_cast<size_t>(o)]) continue;   // a constant, the graph input, or a tensor of another launch behind the cut
                const int d = producer[static_cast<size_t>(t)];
                if (d < 0) BAND_GIVE_UP;   // a tensor inside one of the program's blocks
                const Storage& sp = plan_.storage[static_cast<size_t>(t)];
                if (sp.root < 0 || (sp.offset & 3) || (sp.frame_stride & 3)) BAND_GIVE_UP;
                bool is_out = false;
                for (int go : g.outputs) is_out = is_out || plan_.storage[static_cast<size_t>(go)].root == sp.root;
                if (is_out) { if (prog[static_cast<size_t>(d)].dst_base < 0) BAND_GIVE_UP; continue; }   // (already written where the launch reads it)
                if (plan_.root_offset[static_cast<size_t>(sp.root)] < 0 || (plan_.root_offset[static_cast<size_t>(sp.root)] & 3)) BAND_GIVE_UP;
                BandStage& pd = prog[static_cast<size_t>(d)];
                if (pd.dst_base >= 2 && band_ext_[static_cast<size_t>(pd.dst_base - 2)].tensor == t) continue;   // (a second reader of the same tensor)
                if (pd.dst_base >= 0 || (pd.Co & 3)) BAND_GIVE_UP;
                pd.dst_base = ext_slot(-1, t); pd.dst_off = 0; pd.dst_fs = sp.frame_stride;
                if (pd.dst_base < 0) BAND_GIVE_UP;
            }
        }
        if (!any_band || band_node_runs_[static_cast<size_t>(band_first_)]) BAND_GIVE_UP;
        // the band launch writes its tensors EARLIER than the batched plan's launches would have: a launch that keeps its place in front of the
        // last node the program stands for may only write graph outputs (an arena slot it writes might be one the program's tensors live in)
        size_t last_inside = 0;
        for (size_t i = static_cast<size_t>(band_first_); i < plan_.nodes.size(); i++)
            if (!is_view(plan_.nodes[i]) && !band_node_runs_[i]) last_inside = i;
        for (size_t i = static_cast<size_t>(band_first_); i < last_inside; i++) {
            if (!band_node_runs_[i]) continue;
            std::vector<int> outs = plan_.nodes[i].extra_out;
            outs.push_back(plan_.nodes[i].out);
            for (int t : outs) {
                bool is_out = false;
                for (int go : g.outputs) is_out = is_out || plan_.storage[static_cast<size_t>(go)].root == plan_.storage[static_cast<size_t>(t)].root;
                if (!is_out) BAND_GIVE_UP;
            }
        }
        // the nodes in front of the cut in the level-2 lowering must be exactly the program (nothing the batched plan computes is skipped)
        for (size_t i = cut; i < p2.nodes.size(); i++)
            for (int o : p2.nodes[i].src_ops)
                if (band_op[static_cast<size_t>(o)]) BAND_GIVE_UP;
    }
    // every graph output must be written, whole, by the band program or by a launch behind it (the first convolution writes none of them)
    for (size_t k = 0; k < out_root.size(); k++) {
        size_t written = 0;
        for (const BandStage& st : prog)
            if (st.dst_base >= 2 && band_ext_[static_cast<size_t>(st.dst_base - 2)].out_k == static_cast<int>(k)) written += static_cast<size_t>(st.Ho) * st.Wo * st.Co;
        bool later = false;
        for (size_t i = static_cast<size_t>(band_first_); i < plan_.nodes.size(); i++)
            if (band_node_runs_[i] && plan_.storage[static_cast<size_t>(plan_.nodes[i].out)].root == out_root[k]) later = true;
        if (!later && written != g.tensors[g.outputs[k]].elems()) BAND_GIVE_UP;
    }
    // ---- a program with 2x2 convolutions (the iris network: two branches of 21 stages behind its 8x8 fork) runs branch by branch, not in the
    // graph's interleaved order: a stage goes behind the newest tensor it can read, so that one branch's skip / middle / output tensors and the
    // fork tensor the other branch still waits for are all that is alive — four LDS tiles
    {
        const int N0 = static_cast<int>(prog.size());
        bool any_cv2 = false;
        for (const BandStage& st : prog) any_cv2 = any_cv2 || (st.kind == BAND_PW && st.S == 2);
        if (any_cv2) {
            std::vector<int> pos(static_cast<size_t>(N0), -1), order;
            for (int step = 0; step < N0; step++) {
                int best = -1, best_pos = -2;
                for (int k = 0; k < N0; k++) {
                    const BandStage& st = prog[static_cast<size_t>(k)];
                    if (pos[static_cast<size_t>(k)] >= 0) continue;
                    if (st.dep >= 0 && pos[static_cast<size_t>(st.dep)] < 0) continue;
                    if (st.res_dep >= 0 && pos[static_cast<size_t>(st.res_dep)] < 0) continue;
                    const int dp = st.dep >= 0 ? pos[static_cast<size_t>(st.dep)] : -1;
                    if (dp > best_pos) { best = k; best_pos = dp; }
                }
                if (best < 0) BAND_GIVE_UP;
                pos[static_cast<size_t>(best)] = step;
                order.push_back(best);
            }
            std::vector<BandStage> re;
            for (int k = 0; k < N0; k++) {
                BandStage st = prog[static_cast<size_t>(order[static_cast<size_t>(k)])];
                if (st.dep >= 0) st.dep = pos[static_cast<size_t>(st.dep)];
                if (st.res_dep >= 0) st.res_dep = pos[static_cast<size_t>(st.res_dep)];
                re.push_back(st);
            }
            prog.swap(re);
            for (int& pr : producer)
                if (pr >= 0) pr = pos[static_cast<size_t>(pr)];
        }
        // ... and (any program: the face mesh's two branches behind its 6x6 tensor as well) the second branch behind a fork goes to the workgroups
        // the first one leaves idle (BandStage::woff): where a tensor of one-row bands on every 2nd / 4th ... workgroup is the input of two stages, the
        // later one and everything behind it are run by the workgroups half a group further on, at the same time as the first branch; its first
        // stage takes its whole input from the packet buffer
        {
            if (band_fork_) {
                for (int k = 0; k < N0; k++) {
                    const BandStage& fk = prog[static_cast<size_t>(k)];
                    std::vector<int> readers;
                    for (int j = 0; j < N0; j++)
                        if (prog[static_cast<size_t>(j)].dep == k) readers.push_back(j);
                    if (readers.size() != 2 || fk.R != 1 || fk.wshift < 1 || fk.woff != 0) continue;
                    bool leaf = false;   // (a reader nobody reads is an output head: those have their own rule below)
                    for (int r : readers) {
                        bool is_read = false;
                        for (int j = 0; j < N0; j++) is_read = is_read || prog[static_cast<size_t>(j)].dep == r || prog[static_cast<size_t>(j)].res_dep == r;
                        leaf = leaf || !is_read;
                    }
                    if (leaf) continue;
                    const int woff = 1 << (fk.wshift - 1);
                    std::vector<char> inB(static_cast<size_t>(N0), 0);
                    inB[static_cast<size_t>(readers[1])] = 1;
                    for (int j = readers[1] + 1; j < N0; j++)
                        if (prog[static_cast<size_t>(j)].dep >= 0 && inB[static_cast<size_t>(prog[static_cast<size_t>(j)].dep)]) inB[static_cast<size_t>(j)] = 1;
                    bool ok = true;
                    for (int j = 0; j < N0 && ok; j++) {
                        const BandStage& st = prog[static_cast<size_t>(j)];
                        if (inB[static_cast<size_t>(j)]) {
                            ok = st.wshift >= fk.wshift && st.woff == 0 && (st.res_dep < 0 || st.res_dep == k || inB[static_cast<size_t>(st.res_dep)]);
                        } else if (st.res_dep >= 0 && inB[static_cast<size_t>(st.res_dep)]) {
                            ok = false;
                        }
                    }
                    const BandStage& root = prog[static_cast<size_t>(readers[1])];
                    const int rows = root.kind == BAND_BLOCK ? (root.S == 1 ? root.R + 2 : 2 * root.R + 1) : (root.S == 2 ? 2 * root.R : root.R);
                    if (!ok || rows * root.W * (root.C / 4) > 4 * 512) continue;
                    for (int j = 0; j < N0; j++)
                        if (inB[static_cast<size_t>(j)]) prog[static_cast<size_t>(j)].woff = woff;
                    prog[static_cast<size_t>(readers[1])].Rin = 0;
                    prog[static_cast<size_t>(readers[1])].cross = 1;
                }
            }
        }
    }
    // ---- the output heads (1x1 stages nobody reads: the SSD heads of the detectors) of a tensor of one-row bands go to workgroups the trunk
    // leaves idle there (BandStage::woff, as the iris network's second branch): they take their input rows from the packet buffer and run beside
    // the trunk's next stages instead of in front of them; the two heads of one tensor on two different sets of idle workgroups where there are two
    if (band_fork_) {
        const int N0 = static_cast<int>(prog.size());
        std::vector<char> read(static_cast<size_t>(N0), 0);
        for (const BandStage& st : prog) {
            if (st.dep >= 0) read[static_cast<size_t>(st.dep)] = 1;
            if (st.res_dep >= 0) read[static_cast<size_t>(st.res_dep)] = 1;
        }
        std::vector<int> moved(static_cast<size_t>(N0), 0);   // heads of a tensor already moved
        for (int k = 0; k < N0; k++) {
            BandStage& st = prog[static_cast<size_t>(k)];
            if (read[static_cast<size_t>(k)] || st.kind != BAND_PW || st.S != 1 || st.dep < 0 || st.woff != 0 || st.cross || st.res_mode != RES_NONE) continue;
            const BandStage& pd = prog[static_cast<size_t>(st.dep)];
            if (pd.R != 1 || pd.wshift < 1 || pd.woff != 0 || st.R != 1 || st.wshift != pd.wshift || st.nbands != pd.nbands) continue;
            if (st.W * (st.C / 4) > 4 * 512) continue;   // its one input row: four 16-byte elements per lane
            const int j = moved[static_cast<size_t>(st.dep)]++;
            int woff = 1 << (pd.wshift - 1);
            if ((j & 1) && pd.wshift >= 2) woff += 1 << (pd.wshift - 2);
            st.woff = woff;
            st.Rin = 0;
            st.cross = 1;
        }
    }
    // ---- the output heads (stages nobody reads) move up behind the first other reader of their input: the two LDS tiles hold a tensor
    // only until the trunk has moved on twice, and a head costs its workgroups two microseconds wherever it stands
    {
        const int N0 = static_cast<int>(prog.size());
        std::vector<char> read(static_cast<size_t>(N0), 0), placed(static_cast<size_t>(N0), 0);
        for (const BandStage& st : prog) {
            if (st.dep >= 0) read[static_cast<size_t>(st.dep)] = 1;
            if (st.res_dep >= 0) read[static_cast<size_t>(st.res_dep)] = 1;
        }
        std::vector<int> order;
        for (int k = 0; k < N0; k++) {
            if (placed[static_cast<size_t>(k)]) continue;
            const bool head = !read[static_cast<size_t>(k)] && prog[static_cast<size_t>(k)].kind == BAND_PW;
            if (head) continue;   // placed behind a sibling, or at the end
            order.push_back(k);
            placed[static_cast<size_t>(k)] = 1;
            for (int j = 0; j < N0; j++)
                if (!placed[static_cast<size_t>(j)] && !read[static_cast<size_t>(j)] && prog[static_cast<size_t>(j)].kind == BAND_PW &&
                    prog[static_cast<size_t>(j)].dep == prog[static_cast<size_t>(k)].dep && prog[static_cast<size_t>(j)].dep >= 0) {
                    order.push_back(j);
                    placed[static_cast<size_t>(j)] = 1;
                }
        }
        for (int k = 0; k < N0; k++)
            if (!placed[static_cast<size_t>(k)]) order.push_back(k);
        std::vector<int> new_index(static_cast<size_t>(N0), -1);
        for (int k = 0; k < N0; k++) new_index[static_cast<size_t>(order[static_cast<size_t>(k)])] = k;
        std::vector<BandStage> re;
        for (int k = 0; k < N0; k++) {
            BandStage st = prog[static_cast<size_t>(order[static_cast<size_t>(k)])];
            if (st.dep >= 0) {
                st.dep = new_index[static_cast<size_t>(st.dep)];
                if (st.dep >= k) BAND_GIVE_UP;   // (cannot happen: a head only moves down to behind a reader of its own input)
            }
            if (st.res_dep >= 0) {
                st.res_dep = new_index[static_cast<size_t>(st.res_dep)];
                if (st.res_dep >= k) BAND_GIVE_UP;
            }
            re.push_back(st);
        }
        prog.swap(re);
    }
    // ---- pass 2: who reads what -> LDS tiles (placed by liveness; a straight chain uses two in turn), packet buffers for the rows other
    // workgroups read
    const int NS = static_cast<int>(prog.size());
    std::vector<int> last_reader(static_cast<size_t>(NS), -1);
    int input_last_reader = -1;   // ... of the program's input
    for (int k = 0; k < NS; k++) {
        const BandStage& st = prog[static_cast<size_t>(k)];
        if (st.dep >= 0) {
            BandStage& pd = prog[static_cast<size_t>(st.dep)];
            // a lateral convolution reads a trunk tensor that is 10 - 30 stages old: its rows are this workgroup's own, so the producer also writes them to
            // the launch's workspace and this stage reads them back from there (like the program's input) — the tile does not have to stay alive
            BandStage& me = prog[static_cast<size_t>(k)];
            if (st.res_mode == RES_UP2X && st.kind == BAND_PW && st.S == 1 && !st.cross && k - st.dep > 2 && pd.R == st.R && pd.wshift == st.wshift && pd.woff == st.woff &&
                pd.nbands == st.nbands && (pd.dst_base < 0 || pd.far_copy) && (pd.Co & 3) == 0) {
                me.far_src = 1;
                pd.far_copy = 1;
                pd.dst_base = 0;
            } else {
            last_reader[static_cast<size_t>(st.dep)] = k;
            const bool cv2_halo = st.kind == BAND_PW && st.S == 2 && pd.R < 2 * st.R;   // its row 2r + 1 is the next workgroup's
            if (st.cross) pd.pub_lo = 1;   // (one-row bands: all of the tensor)
            if ((st.kind == BAND_BLOCK || cv2_halo) && pd.nbands > 1) {
                pd.pub_lo = 1;
                if (st.kind == BAND_BLOCK && (st.S == 1 || st.pre) && pd.R > 1) pd.pub_hi = 1;
            }
            }
        } else {
            input_last_reader = k;
        }
        if (st.res_dep >= 0) last_reader[static_cast<size_t>(st.res_dep)] = k;
        if (st.res_dep == -1) input_last_reader = k;
        if (st.res_mode == RES_UP2X) {
            if (st.res_dep < 0) BAND_GIVE_UP;
            prog[static_cast<size_t>(st.res_dep)].pub_lo = 1;   // (one-row bands: all of the coarse tensor travels)
        }
    }
    long ws = 0;
    int dw_floats = 0;
    // LDS tiles, placed by liveness (round 6: an interval allocator — before, up to four equal tiles; full_range's tensors go from 56 KB for a band of
    // 96x96x32 to 14 KB for one of 96x96x8, and its decoder keeps three 31 KB tensors of 48x48x48 alive): a tensor gets the lowest gap that holds it
    // when it is produced and keeps it until its last reader has run.  Rows of a tile: the band's own + one above + one below, + one more below where a
    // stride-2 BLOCK reads the tensor.
    std::vector<char> read_s2(static_cast<size_t>(NS), 0);
    for (const BandStage& st : prog)
        if (st.kind == BAND_BLOCK && st.S == 2 && !st.pre && st.dep >= 0) read_s2[static_cast<size_t>(st.dep)] = 1;
    struct Alloc { int off, size, stage; };   // stage: producer (-1: the program's input, -3: a far input held for one stage)
    auto dead_at = [&](const Alloc& al, int k) { return al.stage == -3 || (al.stage == -1 ? input_last_reader <= k : last_reader[static_cast<size_t>(al.stage)] <= k); };
    // The places are found first, for the smallest arena that takes them: a tensor goes to the lowest or to the highest gap that holds it, whichever leaves
    // the larger free block (first fit from the bottom alone puts full_range's second 96x96x32 tensor in the middle of the arena, and the third, 56 KB,
    // behind it: 155 KB for 113 KB of live tensors)
    std::vector<int> in_off(static_cast<size_t>(NS), -1), out_off(static_cast<size_t>(NS), -1);
    auto layout = [&](int S, int policy) {
        std::vector<Alloc> lv;
        auto put = [&](int need, int stage) {
            need = static_cast<int>(align_up(std::max(need, 16), 16));
            std::sort(lv.begin(), lv.end(), [](const Alloc& x, const Alloc& y) { return x.off < y.off; });
            std::vector<std::pair<int, int>> gaps;   // [begin, end)
            int at = 0;
            for (const Alloc& al : lv) { if (al.off > at) gaps.push_back({at, al.off}); at = std::max(at, al.off + al.size); }
            if (S > at) gaps.push_back({at, S});
            int best = -1, best_left = -1;
            for (const auto& gp : gaps) {
                if (gp.second - gp.first < need) continue;
                for (int end = 0; end < 2; end++) {
                    const int off = end ? gp.second - need : gp.first;
                    int left = 0;   // the largest free block that remains
                    for (const auto& g2 : gaps) {
                        if (&g2 != &gp) left = std::max(left, g2.second - g2.first);
                        else left = std::max(left, std::max(off - g2.first, g2.second - (off + need)));
                    }
                    // policy 0: whichever end leaves the larger free block; 1 / 2: wide tensors (>= 32 KB) at the bottom and narrow ones at the top, or the
                    // other way round (then the first / last gap that fits)
                    const bool wide_t = need >= 8192;
                    const int score = policy == 0 ? left : ((policy == 1) == wide_t ? (end ? -1 : S - off) : (end ? off : -1));
                    if (score > best_left) { best_left = score; best = off; }
                }
            }
            if (best >= 0) lv.push_back(Alloc{best, need, stage});
            return best;
        };
        for (int k = 0; k < NS; k++) {
            const BandStage& st = prog[static_cast<size_t>(k)];
            const int keep_src = st.dep >= 0 && !st.far_src ? st.dep : (st.dep < 0 ? -1 : -4);
            const int keep_res = st.res_mode != RES_UP2X && st.res_dep >= -1 ? st.res_dep : -4;
            std::vector<Alloc> kept;
            for (const Alloc& al : lv)
                if (!dead_at(al, k) || al.stage == keep_src || al.stage == keep_res) kept.push_back(al);
            lv.swap(kept);
            in_off[static_cast<size_t>(k)] = out_off[static_cast<size_t>(k)] = -1;
            if (st.dep < 0) {
                const int rows = st.kind == BAND_BLOCK ? (st.S == 1 ? st.R + 2 : 2 * st.R + 2) : st.R + 1;
                if ((in_off[static_cast<size_t>(k)] = put(rows * (st.W + 2) * (st.C + 4), -1)) < 0) return false;
            } else if (st.far_src) {
                if ((in_off[static_cast<size_t>(k)] = put((st.R + 1) * (st.W + 2) * (st.C + 4), -3)) < 0) return false;
            }
            if (last_reader[static_cast<size_t>(k)] >= 0)
                if ((out_off[static_cast<size_t>(k)] = put(bandnet_tile_floats(st.R, st.Wo, st.Co, 2 + (read_s2[static_cast<size_t>(k)] ? 1 : 0)), k)) < 0) return false;
        }
        return true;
    };
    {
        int S = 4096;
        bool ok = false;
        for (; S <= 40960 && !ok; S += ok ? 0 : 64)   // (floats: 16 .. 160 KB in steps of 256 bytes; three placement policies each)
            for (int policy = 0; policy < 3 && !ok; policy++) ok = layout(S, policy);
        if (!ok) BAND_GIVE_UP;
    }
    std::vector<Alloc> live;
    int tiles_floats = 0;
    auto place = [&](int off, int need, int stage) {   // (the place found above)
        need = static_cast<int>(align_up(std::max(need, 16), 16));
        live.push_back(Alloc{off, need, stage});
        tiles_floats = std::max(tiles_floats, off + need);
        return off;
    };
    auto where = [&](int stage) {
        for (const Alloc& al : live)
            if (al.stage == stage) return al.off;
        return -1;
    };
    for (int k = 0; k < NS; k++) {
        BandStage& st = prog[static_cast<size_t>(k)];
        // what nobody reads any more is free — but for what THIS stage reads (its last reader may be this very stage)
        {
            const int keep_src = st.dep >= 0 && !st.far_src ? st.dep : (st.dep < 0 ? -1 : -4);
            const int keep_res = st.res_mode != RES_UP2X && st.res_dep >= -1 ? st.res_dep : -4;
            std::vector<Alloc> kept;
            for (const Alloc& al : live)
                if (!dead_at(al, k) || al.stage == keep_src || al.stage == keep_res) kept.push_back(al);
            live.swap(kept);
        }
        if (st.dep < 0) {
            // the program's input comes from global memory into a tile of its own: only its first reader may be such a stage (a skip may read it there later)
            if (where(-1) >= 0) BAND_GIVE_UP;
            const int rows = st.kind == BAND_BLOCK ? (st.S == 1 ? st.R + 2 : 2 * st.R + 2) : st.R + 1;
            st.src_lds = place(in_off[static_cast<size_t>(k)], rows * (st.W + 2) * (st.C + 4), -1);
            st.src_tile = 0;
        } else if (st.far_src) {
            // its own rows come back from the workspace into a free place (held for this stage only)
            st.src_lds = place(in_off[static_cast<size_t>(k)], (st.R + 1) * (st.W + 2) * (st.C + 4), -3);
            st.src_tile = 0;
        } else {
            st.src_lds = where(st.dep);
            if (st.src_lds < 0) BAND_GIVE_UP;   // its input is no longer in LDS
            st.src_tile = 0;
            st.src_ll = prog[static_cast<size_t>(st.dep)].dst_ll;
            const BandStage& pd = prog[static_cast<size_t>(st.dep)];
            const bool cv2_halo = st.kind == BAND_PW && st.S == 2 && pd.R < 2 * st.R;
            if (((st.kind == BAND_BLOCK || cv2_halo) && st.nbands > 1 && st.src_ll < 0) || (st.cross && st.src_ll < 0)) BAND_GIVE_UP;
        }
        if (st.res_mode == RES_UP2X) {
            const BandStage& cd = prog[static_cast<size_t>(st.res_dep)];
            if (cd.dst_ll < 0 || cd.R != 1 || cd.Ho * 2 != st.Ho || cd.Wo * 2 != st.Wo || cd.Co != st.Co || st.R != 1) BAND_GIVE_UP;
            st.res_ll = cd.dst_ll;
            st.res_stage = st.res_dep;
            dw_floats = std::max(dw_floats, 2 * cd.Wo * (st.Co + 4));   // its two rows land in the depthwise area (a 1x1 stage does not use it)
        } else if (st.res_dep >= -1) {
            st.res_lds = where(st.res_dep);
            if (st.res_lds < 0) BAND_GIVE_UP;
            st.res_tile = 1;   // (a flag now: the skip comes from another tile, at res_lds)
            // the skip is read at the output's pixel positions: its band must have the output's rows (same shape, same owners)
            if (st.res_mode == RES_MAXPOOL) {
                // ... or, the 2x2 max of the input of the 2x2 convolution / stride-2 block in front: that stage, run by the same workgroups on the same
                // bands, left rows 2 r0 .. 2 r0 + 2 R - 1 of the tensor in the tile it read them from
                if (st.dep < 0 || st.res_dep < 0) BAND_GIVE_UP;
                const BandStage& cv = prog[static_cast<size_t>(st.dep)];
                const BandStage& rd = prog[static_cast<size_t>(st.res_dep)];
                if (cv.S != 2 || cv.dep != st.res_dep || cv.src_lds != st.res_lds || cv.far_src || cv.R != st.R || cv.wshift != st.wshift || cv.nbands != st.nbands)
                    BAND_GIVE_UP;
                if (rd.Ho != 2 * st.Ho || rd.Wo != 2 * st.Wo || rd.Co != st.res_c) BAND_GIVE_UP;
            } else if (st.res_dep >= 0) {
                const BandStage& rd = prog[static_cast<size_t>(st.res_dep)];
                if (rd.Ho != st.Ho || rd.Wo != st.Wo || rd.Co != st.res_c || rd.R != st.R || rd.wshift != st.wshift) BAND_GIVE_UP;
            } else {
                const BandStage& first = prog[0];   // (the stage that loaded the program's input: own rows at tile rows 1 ..)
                if (first.dep >= 0 || first.H != st.Ho || first.W != st.Wo || first.C != st.res_c || first.R != st.R || first.wshift != st.wshift || first.S != 1) BAND_GIVE_UP;
            }
        }
        if (last_reader[static_cast<size_t>(k)] >= 0) {
            st.dst_h3 = read_s2[static_cast<size_t>(k)] ? 1 : 0;
            st.dst_lds = place(out_off[static_cast<size_t>(k)], bandnet_tile_floats(st.R, st.Wo, st.Co, 2 + st.dst_h3), k);
            st.dst_tile = 0;   // (a flag now: the output stays in LDS, at dst_lds)
            if (st.pub_lo || st.pub_hi) {
                st.dst_ll = ws;
                ws += align_up(2 * static_cast<long>(st.Ho) * st.Wo * st.Co, 64);
            }
        }
        if (st.far_copy) {   // the plain copy a lateral convolution reads back (frame stride = the workspace's: patched below)
            st.dst_off = ws;
            ws += align_up(static_cast<long>(st.Ho) * st.Wo * st.Co, 64);
        }
        if (st.far_src) {
            const BandStage& pd = prog[static_cast<size_t>(st.dep)];
            if (!pd.far_copy || pd.dst_base != 0) BAND_GIVE_UP;
            st.src_base = 0;
            st.src_off = pd.dst_off;
        }
        dw_floats = std::max(dw_floats, bandnet_dw_floats(st));
    }
    dw_floats = static_cast<int>(align_up(dw_floats, 4));
    tiles_floats = static_cast<int>(align_up(tiles_floats, 16));
    band_tiles_floats_ = tiles_floats;
    band_dw_floats_ = dw_floats;
    band_cv2_ = false;
    for (const BandStage& st : prog) band_cv2_ = band_cv2_ || (st.kind == BAND_PW && st.S == 2);
    band_xb_ = false;
    for (const BandStage& st : prog) band_xb_ = band_xb_ || (st.kind == BAND_BLOCK && st.cross);
    if (band_cv2_ && band_xb_) BAND_GIVE_UP;   // (no kernel instantiation for both: the iris network's second branch starts with a 1x1 stage)
    band_wide_ = false;
    for (const BandStage& st : prog) band_wide_ = band_wide_ || st.C > 128 || st.Co > 128 || st.res_mode == RES_UP2X || st.pre;
    if (band_wide_ && (band_cv2_ || band_xb_)) BAND_GIVE_UP;   // (likewise)
    band_lds_bytes_ = bandnet_lds_bytes(tiles_floats, dw_floats, NS);
    if (std::getenv("MI_BAND_DEBUG")) {
        std::fprintf(stderr, "bandnet: %d stages, NW %d, LDS %d B = tiles %d + depthwise %d + constants + program\n", NS, NW, band_lds_bytes_, tiles_floats * 4, dw_floats * 4);
        for (int k = 0; k < NS; k++) {
            const BandStage& st = prog[static_cast<size_t>(k)];
            std::fprintf(stderr, "  stage %2d %s S%d %dx%dx%d -> %dx%dx%d R %d wshift %d src %d dst %d res %d (dep %d, mode %d, c %d) last reader %d\n", k, st.kind == BAND_BLOCK ? "block" : "pw   ", st.S, st.H, st.W, st.C,
                         st.Ho, st.Wo, st.Co, st.R, st.wshift, st.src_lds * 4, st.dst_tile >= 0 ? st.dst_lds * 4 : -1, st.res_tile >= 0 ? st.res_lds * 4 : -1, st.res_dep, st.res_mode, st.res_c, last_reader[static_cast<size_t>(k)]);
        }
    }
    if (band_lds_bytes_ > 160 * 1024) BAND_GIVE_UP;
    band_ws_frame_floats_ = std::max<long>(ws, 64);
    band_nstages_ = NS;
    consts.resize(consts.size() + 64, 0.f);
    std::vector<BandPacked> packed(prog.size());
    for (size_t k = 0; k < prog.size(); k++) {
        BandStage q = prog[k];
        if (q.far_copy) q.dst_fs = band_ws_frame_floats_;
        if (q.far_src) { q.src_fs = band_ws_frame_floats_; q.dep = -1; q.Rin = 0; }   // (the kernel's "input in plain memory" path; the stage order keeps the host's dep)
        if (!bandnet_pack(q, &packed[k])) BAND_GIVE_UP;
    }
    hip_check(hipMalloc(reinterpret_cast<void**>(&d_band_prog_), packed.size() * sizeof(BandPacked)), "hipMalloc band program");
    hip_check(hipMemcpy(d_band_prog_, packed.data(), packed.size() * sizeof(BandPacked), hipMemcpyHostToDevice), "upload band program");
    hip_check(hipMalloc(reinterpret_cast<void**>(&d_band_consts_), consts.size() * sizeof(float)), "hipMalloc band constants");
    hip_check(hipMemcpy(d_band_consts_, consts.data(), consts.size() * sizeof(float), hipMemcpyHostToDevice), "upload band constants");
    const size_t ws_bytes = static_cast<size_t>(band_ws_frame_floats_) * band_max_frames_ * sizeof(float);
    band_ws_bytes_ = ws_bytes;
    hip_check(hipMalloc(reinterpret_cast<void**>(&d_band_ws_), ws_bytes), "hipMalloc band workspace");
    hip_check(hipMemset(d_band_ws_, 0, ws_bytes), "hipMemset");   // no packet carries a tag yet (tags start at 1)
    hip_check(hipMalloc(reinterpret_cast<void**>(&d_band_sync_), 64 * sizeof(unsigned)), "hipMalloc band generation");
    hip_check(hipMemset(d_band_sync_, 0, 64 * sizeof(unsigned)), "hipMemset");
    hip_check(hipHostMalloc(reinterpret_cast<void**>(&h_band_fail_), 64, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
    *h_band_fail_ = 0;
    hip_check(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_band_fail_), h_band_fail_, 0), "hipHostGetDevicePointer");
    band_ready_ = true;
}

#undef BAND_GIVE_UP

void Model::ensure_capacity(int batch) {
    int chunk = chunk_ > 0 ? std::min(chunk_, batch) : batch;
    if (lanes_ > 1) chunk = std::min(chunk, (batch + lanes_ - 1) / lanes_);  // each lane owns one arena region
    if (chunk > chunk_cap_ || static_cast<size_t>(plan_.arena_floats_per_frame) * chunk * lanes_ > arena_floats_) {
        invalidate_graphs();
        if (d_arena_) hip_check(hipFree(d_arena_), "hipFree arena");
        d_arena_ = nullptr;
        chunk = std::max(chunk, chunk_cap_);
        arena_floats_ = static_cast<size_t>(plan_.arena_floats_per_frame) * chunk * lanes_;
        hip_check(hipMalloc(reinterpret_cast<void**>(&d_arena_), std::max<size_t>(arena_floats_, 64) * sizeof(float)), "hipMalloc arena");
        chunk_cap_ = chunk;
    }
    if (!d_small_ && small_chain_ > 0) {  // ping-pong scratch of the small-batch form of the row-pipelined chains (fixed size: replay graphs hold the pointer)
        size_t fmax = 0;
        for (const Node& n : plan_.nodes)
            if (n.kind == Node::Chain) {
                const auto& sh = plan_.graph.tensors[n.in[0]].shape;
                if (sh.size() == 4 && sh[1] * sh[2] > 256) fmax = std::max(fmax, static_cast<size_t>(sh[1]) * sh[2] * sh[3]);
            }
        if (fmax) {
            small_floats_ = 2 * static_cast<size_t>(small_chain_) * fmax;
            hip_check(hipMalloc(reinterpret_cast<void**>(&d_small_), small_floats_ * sizeof(float)), "hipMalloc small-batch scratch");
        }
    }
    if (batch > batch_cap_) {
        invalidate_graphs();
        for (float* p : d_out_)
            if (p) hip_check(hipFree(p), "hipFree out");
        d_out_.assign(num_outputs(), nullptr);
        for (int k = 0; k < num_outputs(); k++)
            hip_check(hipMalloc(reinterpret_cast<void**>(&d_out_[k]), output_elems(k) * sizeof(float) * batch), "hipMalloc outputs");
        batch_cap_ = batch;
    }
}

const float* Model::tensor_ptr(int t, const float* in, int chunk_start, long* fs) const {
    const Graph& g = plan_.graph;
    const Storage& s = plan_.storage[t];
    *fs = s.frame_stride;
    if (s.root == plan_.storage[g.inputs[0]].root) return in + static_cast<long>(chunk_start) * s.frame_stride + s.offset;
    return tensor_ptr_mut(t, chunk_start, fs);
}

float* Model::tensor_ptr_mut(int t, int chunk_start, long* fs) const {
    const Graph& g = plan_.graph;
    const Storage& s = plan_.storage[t];
    *fs = s.frame_stride;
    for (size_t k = 0; k < g.outputs.size(); k++)
        if (plan_.storage[g.outputs[k]].root == s.root) return d_out_[k] + static_cast<long>(chunk_start) * s.frame_stride + s.offset;
    if (s.root == plan_.storage[g.inputs[0]].root) throw std::runtime_error("plan writes into the graph input");
    long off = plan_.root_offset[s.root];
    if (off < 0) throw std::runtime_error("tensor has no storage");
    return d_arena_ + static_cast<size_t>(plan_.arena_floats_per_frame) * chunk_cap_ * arena_lane_ + off * chunk_cap_ + s.offset;
}

std::string Model::node_label(const Node& n) const {
    const Graph& g = plan_.graph;
    switch (n.kind) {
        case Node::Conv: return n.gemm_head ? "head_gemm_kernel" : "conv_generic_kernel";
        case Node::Dw: return "dw_kernel";
        case Node::Block: {
            const int Co = g.tensors[n.out].shape.back();
            const int MT = (Co + 31) / 32, MTG = std::min(4, MT), PG = MT <= 2 ? 2 : 1;
            return "block_kernel<" + std::to_string(MTG) + "," + std::to_string(n.w >= 0 ? n.sh : 1) + "," + (n.w >= 0 ? "3" : "1") + "," + std::to_string(PG) + ">";
        }
        case Node::Chain: {
            const auto& so = g.tensors[n.out].shape;
            if (n.chain_pre || n.chain_post) {
                const auto& sm = g.tensors[n.members[n.chain_pre ? 1 : 0].in[0]].shape;
                return "chain_kernel<" + std::to_string((sm.back() + 31) / 32) + ">";
            }
            if (g.tensors[n.in[0]].shape[1] * g.tensors[n.in[0]].shape[2] <= 256) return "chain_kernel<" + std::to_string((so.back() + 31) / 32) + ">";
            const auto& sin = g.tensors[n.in[0]].shape;
            const int nh2 = n.members.back().sh == 2 ? so.back() / sin.back() : 0;
            const int rps = strip_pipe_rows_per_step(sin[1], pipe_rows_);
            return std::string(rps == 4 ? "strip_pipe1m_kernel<" : (rps == 3 ? "strip_pipe2m_kernel<" : (rps == 2 ? "strip_pipe2_kernel<" : "strip_pipe_kernel<"))) + std::to_string(sin.back() / 4) + "," + std::to_string(n.members.size()) + "," + (n.members[0].act == ACT_RELU ? "1" : "0") + "," + std::to_string(nh2) + ">";
        }
        case Node::Resident: return n.xc ? "xc_kernel" : (n.dblock ? "dblock_kernel" : (n.bneck ? "bneck_kernel" : (n.tail ? "tail_kernel" : "resident_kernel")));
        case Node::Add: return "add_kernel";
        case Node::Act: return "act_kernel";
        case Node::MaxPool: return "maxpool_kernel";
        case Node::Pad: return "pad_kernel";
        case Node::Resize: return "resize_kernel";
        case Node::DepthToSpace: return "d2s_kernel";
        default: return "view";
    }
}

std::vector<Model::LaunchStat> Model::profile(const float* in, int batch, int reps, hipStream_t stream) {
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (dirty_) rebuild();
    const int saved_chunk = chunk_;
    ensure_capacity(batch);
    hipStream_t s = stream ? stream : stream_;
    const Graph& g = plan_.graph;
    std::vector<LaunchStat> stats;
    for (const Node& n : plan_.nodes) {
        if (n.kind == Node::Reshape || n.kind == Node::Concat) continue;
        LaunchStat st;
        st.kernel = node_label(n);
        auto elems = [&](int t) { return t >= 0 ? static_cast<double>(g.tensors[t].elems()) : 0.0; };
        double per_frame = 4 * elems(n.out);
        for (int t : n.in) per_frame += 4 * elems(t);
        if (n.res >= 0 && !(n.kind == Node::Block && n.res == n.in[0])) per_frame += 4 * elems(n.res);
        double weights = 0;
        for (int c : {n.w, n.b, n.w2, n.b2, n.alpha}) weights += 4 * elems(c);
        st.bytes = per_frame * batch + weights;
        const auto& si = g.tensors[n.in[0]].shape;
        const auto& so = g.tensors[n.out].shape;
        if (n.kind == Node::Conv) st.macs = elems(n.out) * n.KH * n.KW * si.back() * batch;
        if (n.kind == Node::Dw) st.macs = elems(n.out) * n.KH * n.KW * batch;
        if (n.kind == Node::Block) st.macs = elems(n.out) / so.back() * si.back() * ((n.w >= 0 ? 9 : 0) + so.back()) * batch;
        if (n.kind == Node::Resident) {
            st.macs = 0;
            for (int t : n.extra_out) st.bytes += 4 * elems(t) * batch;
            for (const Node& m : n.members) {
                const auto& mo = g.tensors[m.out].shape;
                const int Cm = g.tensors[m.in[0]].shape[3];
                if (m.kind == Node::Conv) st.macs += elems(m.out) * m.KH * m.KW * Cm * batch;
                else st.macs += static_cast<double>(mo[1]) * mo[2] * Cm * ((m.w >= 0 ? 9 : 0) + mo[3]) * batch;
                for (int c : {m.w, m.b, m.w2, m.b2, m.alpha}) st.bytes += 4 * elems(c);
            }
        }
        if (n.kind == Node::Chain) {
            st.macs = 0;
            for (int t : n.extra_out) st.bytes += 4 * elems(t) * batch;
            for (const Node& m : n.members) {
                const auto& mo = g.tensors[m.out].shape;
                st.macs += static_cast<double>(mo[1]) * mo[2] * g.tensors[m.in[0]].shape[3] * (9 + mo[3]) * batch;
            }
            for (const Node& m : n.members)
                for (int c : {m.w, m.b, m.w2, m.b2, m.alpha}) st.bytes += 4 * elems(c);
            for (const Node& m : n.head_nodes) {
                st.macs += elems(m.out) * g.tensors[m.in[0]].shape[3] * batch;
                for (int c : {m.w, m.b, m.w2, m.b2}) st.bytes += 4 * elems(c);
            }
        }
        std::string d;
        for (size_t k = 1; k < si.size(); k++) d += (k > 1 ? "x" : "") + std::to_string(si[k]);
        d += "->";
        for (size_t k = 1; k < so.size(); k++) d += (k > 1 ? "x" : "") + std::to_string(so[k]);
        st.detail = d;
        stats.push_back(st);
    }
    const size_t nl = stats.size();
    const int nchunks = (batch + chunk_cap_ - 1) / chunk_cap_;
    std::vector<hipEvent_t> marks;
    // one untimed pass first: the first launch of a kernel symbol in a process loads its code object (about a millisecond,
    // which averaged over a few reps made the first layer of every kernel type look 3x slower than its twins)
    profile_inner_ = 4;
    band_use_ = band_ == 2 && band_usable(batch);
    for (int r = -1; r < reps; r++) {
        marks.clear();
        std::vector<std::string> labels;
        for (int start = 0; start < batch; start += chunk_cap_) enqueue_chunk(in, start, std::min(chunk_cap_, batch - start), s, &marks, &labels);
        for (size_t i = 0; i < nl && i < labels.size(); i++) stats[i].kernel = labels[i];
        hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        if (marks.size() != static_cast<size_t>(nchunks) * (nl + 1)) throw std::runtime_error("internal: profile marks mismatch");
        for (int c = 0; c < nchunks; c++)
            for (size_t i = 0; i < nl; i++) {
                float ms = 0;
                hip_check(hipEventElapsedTime(&ms, marks[c * (nl + 1) + i], marks[c * (nl + 1) + i + 1]), "hipEventElapsedTime");
                if (r >= 0) stats[i].ms += ms / reps / profile_inner_;
            }
        for (hipEvent_t ev : marks) hipEventDestroy(ev);
    }
    profile_inner_ = 1;
    band_use_ = false;
    (void)saved_chunk;
    // plan nodes that ran inside the launch before them (a run of blocks on mstrip_chain_kernel): their work belongs to that launch
    for (size_t i = 1; i < stats.size();) {
        if (stats[i].kernel == "(inside the band launch)") {   // its work belongs to the band launch, which need not be the entry in front of it
            for (size_t j = 0; j < i; j++)
                if (stats[j].kernel == "bandnet_kernel") {
                    stats[j].ms += stats[i].ms; stats[j].bytes += stats[i].bytes; stats[j].macs += stats[i].macs;
                    const size_t arrow = stats[i].detail.find("->"), parrow = stats[j].detail.find("->");
                    if (arrow != std::string::npos && parrow != std::string::npos) stats[j].detail = stats[j].detail.substr(0, parrow) + stats[i].detail.substr(arrow);
                }
            stats.erase(stats.begin() + static_cast<long>(i));
        } else if (stats[i].kernel == "(fused into previous launch)") {
            stats[i - 1].ms += stats[i].ms; stats[i - 1].bytes += stats[i].bytes; stats[i - 1].macs += stats[i].macs;
            const size_t arrow = stats[i].detail.find("->");
            const size_t parrow = stats[i - 1].detail.find("->");
            if (arrow != std::string::npos && parrow != std::string::npos) stats[i - 1].detail = stats[i - 1].detail.substr(0, parrow) + stats[i].detail.substr(arrow);
            stats.erase(stats.begin() + static_cast<long>(i));
        } else {
            i++;
        }
    }
    return stats;
}

void Model::enqueue_chunk(const float* in, int chunk_start, int F, hipStream_t s, std::vector<hipEvent_t>* marks,
                          std::vector<std::string>* labels) {
    const Graph& g = plan_.graph;
    auto mark = [&] {
        if (!marks) return;
        hipEvent_t ev;
        hip_check(hipEventCreate(&ev), "hipEventCreate");
        hip_check(hipEventRecord(ev, s), "hipEventRecord");  // profiling marks: eager launches only
        marks->push_back(ev);
    };
    mark();
    // heads beside the trunk (not while profiling: the per-launch events there assume one stream)
    // (with several lanes the chunks already overlap; the head streams and node events are one set per model, and sharing
    // them between concurrently captured lanes crashes hipGraph capture)
    const bool fork = fork_ && !marks && lanes_ == 1;
    hipStream_t const trunk = s;
    unsigned used_heads = 0;
    auto node_event = [&](size_t k) {
        if (head_events_.size() < plan_.nodes.size() + 2 + kHeadStreams) head_events_.resize(plan_.nodes.size() + 2 + kHeadStreams, nullptr);
        if (!head_events_[k]) hip_check(hipEventCreateWithFlags(&head_events_[k], hipEventDisableTiming), "hipEventCreate");
        return head_events_[k];
    };
    int fused_behind = 0;   // plan nodes that ran inside the launch just made (a run of 32x32x48 blocks on mstrip_chain_kernel)
    const bool band = band_use_ && chunk_start == 0;
    for (size_t i = 0; i < plan_.nodes.size(); i++) {
        const Node& n = plan_.nodes[i];
        if (n.kind == Node::Reshape || n.kind == Node::Concat) continue;  // views
        if (band && static_cast<int>(i) >= band_first_ && !band_node_runs_[i]) {
            // the single-launch plan: everything behind the first convolution is this one launch (bandnet_kernels.hip), but for the nodes behind
            // the program's end (band_node_runs_), which keep their launches
            if (static_cast<int>(i) > band_first_) {
                if (labels) labels->push_back("(inside the band launch)");
                mark();
                continue;
            }
            if (labels) labels->push_back("bandnet_kernel");
            BandLaunch a;
            a.prog = d_band_prog_; a.nstages = band_nstages_; a.NW = band_nw_used_; a.F = F; a.lds_bytes = band_lds_bytes_;
            a.tiles_floats = band_tiles_floats_;
            a.dw_floats = band_dw_floats_; a.ws_frame_floats = band_ws_frame_floats_;
            long fs = 0;
            a.base[0] = d_band_ws_;
            a.base[1] = const_cast<float*>(tensor_ptr(band_stem_out_, in, chunk_start, &fs));
            a.cv2 = band_cv2_ ? 1 : 0;
            a.xb = band_xb_ ? 1 : 0;
            a.wide = band_wide_ ? 1 : 0;
            for (size_t k = 0; k < band_ext_.size(); k++) {
                long efs = 0;
                a.base[2 + k] = band_ext_[k].out_k >= 0 ? d_out_[static_cast<size_t>(band_ext_[k].out_k)] : tensor_ptr_mut(band_ext_[k].tensor, chunk_start, &efs);
            }
            a.consts = d_band_consts_; a.sync = d_band_sync_; a.fail = d_band_fail_;
            a.absent_mod = band_test_absent_;
            int rc = 0;
            for (int rep_ = 0; rep_ < (marks ? profile_inner_ : 1) && rc == 0; rep_++) rc = launch_bandnet(a, trunk);
            if (rc != 0) throw std::runtime_error(std::string("kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
            // launches behind the program that run beside the trunk wait for the node that made their input: every such node inside the
            // program is this launch
            for (size_t j = i; fork && j < plan_.nodes.size(); j++)
                if (!band_node_runs_[j] && event_after_[j]) hip_check(record_event(node_event(j), trunk), "hipEventRecord");
            mark();
            continue;
        }
        if (fused_behind > 0) {
            fused_behind--;
            if (labels) labels->push_back("(fused into previous launch)");
            if (fork && event_after_[i]) hip_check(record_event(node_event(i), trunk), "hipEventRecord");
            mark();
            continue;
        }
        s = trunk;
        // (a whole-frame convolution right behind the band launch stays on the trunk: the face mesh's two heads are 9 + 11 us, a side stream's
        // events cost more than they hide — FaceLandmark::infer 208 us forked, 190 us in line)
        const bool in_line = band && n.gemm_head && head_wait_[i] >= 0 && !band_node_runs_[static_cast<size_t>(head_wait_[i])];
        if (fork && head_slot_[i] >= 0 && !in_line) {
            while (static_cast<int>(head_streams_.size()) <= head_slot_[i]) {
                hipStream_t st;
                hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
                head_streams_.push_back(st);
            }
            s = head_streams_[head_slot_[i]];
            if (head_wait_[i] >= 0) {
                hip_check(wait_event(s, node_event(static_cast<size_t>(head_wait_[i]))), "hipStreamWaitEvent");
            } else {  // reads the graph input: order it behind whatever the trunk stream was doing before this plan
                hip_check(record_event(node_event(plan_.nodes.size()), trunk), "hipEventRecord");
                hip_check(wait_event(s, node_event(plan_.nodes.size())), "hipStreamWaitEvent");
            }
            used_heads |= 1u << head_slot_[i];
        }
        const auto& si = g.tensors[n.in[0]].shape;
        const auto& so = g.tensors[n.out].shape;
        auto dim = [](const std::vector<int>& v, size_t d) { return d < v.size() ? v[d] : 1; };
        Epilogue ep;
        ep.bias = (n.kind == Node::Block ? node_b2_[i] : node_b_[i]) >= 0 ? d_weights_ + (n.kind == Node::Block ? node_b2_[i] : node_b_[i]) : nullptr;
        ep.alpha = node_alpha_[i] >= 0 ? d_weights_ + node_alpha_[i] : nullptr;
        ep.act = n.act;
        if (n.res >= 0) {
            const auto& sr = g.tensors[n.res].shape;
            ep.res = tensor_ptr(n.res, in, chunk_start, &ep.res_fs);
            ep.res_mode = n.res_mode;
            ep.res_after = n.res_after ? 1 : 0;
            ep.res_C = sr.back();
            ep.res_H = dim(sr, 1);
            ep.res_W = dim(sr, 2);
        }
        int rc = 0;
        long in_fs = 0, out_fs = 0;
        if (labels) labels->push_back(node_label(n));
        const float* ip = tensor_ptr(n.in[0], in, chunk_start, &in_fs);
        float* op = tensor_ptr_mut(n.out, chunk_start, &out_fs);
        // profiling: the launch is repeated between its two marks (a launch never reads what it writes), so that the event bubble
        // between marks is shared by profile_inner_ executions and the per-launch figure approaches rocprofv3's kernel duration
        for (int rep_ = 0; rep_ < (marks ? profile_inner_ : 1) && rc == 0; rep_++)
        switch (n.kind) {
            case Node::Conv: {
                if (n.gemm_head) {
                    HeadGemmArgs h;
                    h.in = ip; h.out = op; h.in_fs = in_fs; h.out_fs = out_fs;
                    h.w = d_weights_ + node_w_[i];
                    h.bias = ep.bias; h.alpha = ep.alpha; h.act = ep.act;
                    h.B = F; h.K = si[1] * si[2] * si[3]; h.N = so[3];
                    rc = launch_head_gemm(h, s);
                    if (labels) labels->back() = F <= 4 ? "head_dot_kernel" : "head_gemm_kernel";   // (launch_head_gemm: a handful of frames take the dot-product form)
                    break;
                }
                // the first convolution inside the launch of the pair of BlazeBlocks behind it (f32 pictures, from 32 frames on: mdblock_kernels.hip, MD::STEM)
                if (strip_ && stem_fuse_ && node_stem_[i] >= 0 && !u8_.frames) {   // (node_stem_: the graph's side of the conditions, checked when the constants were packed)
                    const size_t j = i + 1;
                    if (node_mwalk_[j] >= 0 && !event_after_[i] && head_slot_[j] < 0) {
                        const Node& c = plan_.nodes[j];
                        const auto& co = g.tensors[c.out].shape;
                        DblockArgs d;
                        long ofs = 0;
                        d.out = tensor_ptr_mut(c.out, chunk_start, &ofs); d.out_fs = ofs;
                        d.B = F; d.H = so[1]; d.W = so[2]; d.C = so[3]; d.Cm = g.tensors[c.members[0].out].shape[3]; d.Co = co[3];
                        d.hi1 = c.members[0].act == ACT_RELU6 ? 6.f : INFINITY;
                        d.hi2 = c.members[1].act == ACT_RELU6 ? 6.f : INFINITY;
                        d.skip1 = 1; d.skip2_from_a = 1;
                        d.act1 = c.members[0].act; d.act2 = c.members[1].act;
                        d.mconsts = d_weights_ + node_mwalk_[j];
                        d.stem_in = ip; d.stem_in_fs = in_fs; d.stem_consts = d_weights_ + node_stem_[i];
                        d.stem_hi = n.act == ACT_RELU6 ? 6.f : INFINITY;
                        d.band_rows = mdb_band_;
                        if (mdblock_kernel_supports(d)) {
                            if (labels) labels->back() = "mdblock_kernel<stem+pair>";
                            rc = launch_mdblock(d, s);
                            fused_behind = 1;
                            break;
                        }
                    }
                }
                ConvArgs a;
                a.in = ip; a.out = op; a.in_fs = in_fs; a.out_fs = out_fs;
                a.w = d_weights_ + node_w_[i];
                a.B = F; a.H = si[1]; a.W = si[2]; a.C = si[3]; a.Ho = so[1]; a.Wo = so[2]; a.Co = so[3]; a.Cop = (a.Co + 3) & ~3;
                a.KH = n.KH; a.KW = n.KW; a.sh = n.sh; a.sw = n.sw;
                if (n.padding == Padding::Same) { same_pad(a.H, a.KH, a.sh, a.Ho, &a.pt); same_pad(a.W, a.KW, a.sw, a.Wo, &a.pl); }
                if (n.ept >= 0) { a.pt = n.ept; a.pl = n.epl; }
                a.ep = ep;
                a.no_mfma = stem_mfma_ ? 0 : 1;
                if (u8_.frames && plan_.storage[n.in[0]].root == plan_.storage[g.inputs[0]].root) {
                    if (!conv_takes_u8(a)) throw std::runtime_error("plan: this graph's first convolution has no u8 input form");
                    a.in_u8 = u8_.frames + static_cast<long>(chunk_start) * u8_.frame_bytes;
                    a.u8_lut = u8_.lut; a.u8_frame_bytes = u8_.frame_bytes; a.u8_row_bytes = u8_.row_bytes;
                }
                if (labels) labels->back() = conv_kernel_label(a);
                rc = launch_conv(a, s);
                break;
            }
            case Node::Dw: {
                DwArgs a;
                a.in = ip; a.out = op; a.in_fs = in_fs; a.out_fs = out_fs;
                a.w = d_weights_ + node_w_[i];
                a.B = F; a.H = si[1]; a.W = si[2]; a.C = si[3]; a.Ho = so[1]; a.Wo = so[2];
                a.KH = n.KH; a.KW = n.KW; a.sh = n.sh; a.sw = n.sw;
                if (n.padding == Padding::Same) { same_pad(a.H, a.KH, a.sh, a.Ho, &a.pt); same_pad(a.W, a.KW, a.sw, a.Wo, &a.pl); }
                if (n.ept >= 0) { a.pt = n.ept; a.pl = n.epl; }
                a.ep = ep;
                rc = launch_dw(a, s);
                break;
            }
            case Node::Resident: {
                if (n.xc) {
                    XcArgs a;
                    a.in = ip; a.in_fs = in_fs; a.out = op; a.out_fs = out_fs;
                    a.B = F; a.H = si[1]; a.W = si[2]; a.nstages = static_cast<int>(n.members.size());
                    for (size_t k = 0; k < n.members.size(); k++) {
                        const Node& m = n.members[k];
                        const MemberOff& mo = chain_off_[i][k];
                        XcStage& st = a.st[k];
                        st.cblob = d_weights_ + mo.strip;
                        st.has_dw = m.w >= 0;
                        st.w_pw = d_weights_ + mo.w2;
                        st.C = g.tensors[m.in[0]].shape[3]; st.Co = g.tensors[m.out].shape[3]; st.act = m.act;
                        st.skip = m.res < 0 ? 0 : (m.res == m.in[0] ? 1 : (k >= 2 && m.res == n.members[k - 2].out ? 3 : 2));
                        if (st.skip == 2) {
                            st.res = tensor_ptr(m.res, in, chunk_start, &st.res_fs);
                            st.res_C = g.tensors[m.res].shape[3]; st.res_W = g.tensors[m.res].shape[2];
                        }
                    }
                    rc = launch_xc(a, s);
                    break;
                }
                if (n.dblock) {
                    DblockArgs a;
                    a.in = ip; a.in_fs = in_fs; a.out = op; a.out_fs = out_fs;
                    a.B = F; a.H = si[1]; a.W = si[2]; a.C = si[3]; a.Cm = g.tensors[n.members[0].out].shape[3]; a.Co = so[3];
                    a.w1 = d_weights_ + chain_off_[i][0].w2;
                    a.w2 = d_weights_ + chain_off_[i][1].w2;
                    a.consts = d_weights_ + chain_off_[i][0].strip;
                    a.hi1 = n.members[0].act == ACT_RELU6 ? 6.f : INFINITY;
                    a.hi2 = n.members[1].act == ACT_RELU6 ? 6.f : INFINITY;
                    a.skip1 = n.members[0].res >= 0;                       // two plain BlazeBlocks: each adds its own input
                    a.skip2_from_a = n.members[1].res == n.members[0].out;
                    a.act1 = n.members[0].act; a.act2 = n.members[1].act;
                    if (chain_off_[i][1].strip >= 0) a.mconsts = d_weights_ + chain_off_[i][1].strip;
                    if (strip_ && mdblock_kernel_supports(a)) {
                        if (labels) labels->back() = "mdblock_kernel";
                        rc = launch_mdblock(a, s);
                        break;
                    }
                    rc = launch_dblock(a, s);
                    break;
                }
                if (n.bneck) {
                    BneckArgs a;
                    a.in = ip; a.in_fs = in_fs; a.out = op; a.out_fs = out_fs;
                    a.B = F; a.H = si[1]; a.W = si[2]; a.C = si[3]; a.Cm = g.tensors[n.members[0].out].shape[3];
                    a.nblocks = static_cast<int>(n.members.size() / 2);
                    a.bands = n.res_bands;
                    for (int k = 0; k < a.nblocks; k++) {
                        const MemberOff &ma = chain_off_[i][static_cast<size_t>(2 * k)], &mb = chain_off_[i][static_cast<size_t>(2 * k + 1)];
                        a.blocks[k].w1 = d_weights_ + ma.w2;
                        a.blocks[k].w2 = d_weights_ + mb.w2;
                        a.blocks[k].consts = d_weights_ + ma.strip;
                        a.blocks[k].hi1 = n.members[static_cast<size_t>(2 * k)].act == ACT_RELU6 ? 6.f : INFINITY;
                        a.blocks[k].hi2 = n.members[static_cast<size_t>(2 * k + 1)].act == ACT_RELU6 ? 6.f : INFINITY;
                        a.blocks[k].act1 = n.members[static_cast<size_t>(2 * k)].act;
                        a.blocks[k].act2 = n.members[static_cast<size_t>(2 * k + 1)].act;
                        if (mb.strip >= 0) a.blocks[k].mconsts = d_weights_ + mb.strip;
                    }
                    if (strip_ && mbneck_kernel_supports(a)) {
                        if (labels) labels->back() = "mbneck_kernel";
                        rc = launch_mbneck(a, s);
                        break;
                    }
                    rc = launch_bneck(a, s);
                    break;
                }
                if (n.tail) {
                    TailLaunch a;
                    a.prog = d_tail_programs_ + node_prog_[i];
                    a.nstages = static_cast<int>(n.stages.size());
                    a.B = F;
                    a.frame_floats = n.tail_frame_floats;
                    a.variant = tail_pre_;
                    // frames per workgroup: as many as keep every CU busy (a workgroup's stage costs the same few thousand cycles of
                    // latency whether its pixel tiles are full or not), within what the CU's LDS holds
                    const int gmax = std::max(1, (160 * 1024 - 1024) / (4 * n.tail_frame_floats));
                    a.G = tail_g_ > 0 ? std::min(tail_g_, gmax) : std::max(1, std::min(gmax, F / device_cu_count()));
                    for (int k = 0; k < kResBases; k++) { a.bases.p[k] = nullptr; a.bases.scale[k] = 0; a.bases.frame0[k] = 0; }
                    a.bases.p[0] = d_arena_ + static_cast<size_t>(plan_.arena_floats_per_frame) * chunk_cap_ * arena_lane_;
                    a.bases.scale[0] = chunk_cap_;
                    a.bases.p[1] = const_cast<float*>(in);
                    a.bases.frame0[1] = chunk_start;
                    for (int k = 0; k < num_outputs() && 2 + k < kResBases; k++) { a.bases.p[2 + k] = d_out_[k]; a.bases.frame0[2 + k] = chunk_start; }
                    a.bases.weights = d_weights_;
                    rc = launch_tail(a, s);
                    break;
                }
                ResLaunch a;
                a.prog = d_programs_ + node_prog_[i];
                a.nstages = static_cast<int>(n.stages.size());
                a.B = F;
                a.bands = n.res_bands;
                a.const_off = n.res_const_off;
                a.const_floats = n.res_const_floats;
                a.lds_bytes = n.res_lds_bytes;
                for (int k = 0; k < kResBases; k++) { a.bases.p[k] = nullptr; a.bases.scale[k] = 0; a.bases.frame0[k] = 0; }
                a.bases.p[0] = d_arena_ + static_cast<size_t>(plan_.arena_floats_per_frame) * chunk_cap_ * arena_lane_;
                a.bases.scale[0] = chunk_cap_;
                a.bases.p[1] = const_cast<float*>(in);
                a.bases.frame0[1] = chunk_start;
                for (int k = 0; k < num_outputs() && 2 + k < kResBases; k++) { a.bases.p[2 + k] = d_out_[k]; a.bases.frame0[2 + k] = chunk_start; }
                a.bases.weights = d_weights_;
                rc = launch_resident(a, s);
                break;
            }
            case Node::Chain: {
                ChainArgs a;
                auto fill = [&](ChainBlock& cb, size_t k) {
                    const MemberOff& mo = chain_off_[i][k];
                    cb.w_dw = d_weights_ + mo.w;
                    cb.b_dw = mo.b >= 0 ? d_weights_ + mo.b : nullptr;
                    cb.w_pw = d_weights_ + mo.w2;
                    cb.bias = mo.b2 >= 0 ? d_weights_ + mo.b2 : nullptr;
                    cb.alpha = mo.alpha >= 0 ? d_weights_ + mo.alpha : nullptr;
                    cb.act = n.members[k].act;
                    cb.has_res = n.members[k].res >= 0;
                };
                if (n.chain_pre || n.chain_post || !n.head_pairs.empty()) {  // frame-resident chain with stride-2 blocks around it and / or output heads in the same launch
                    const size_t k0 = n.chain_pre ? 1 : 0, k1 = n.members.size() - (n.chain_post ? 1 : 0);
                    const auto& sm = g.tensors[n.members[k0].in[0]].shape;  // the resident frame
                    a.B = F; a.H = sm[1]; a.W = sm[2]; a.C = sm[3]; a.nblocks = static_cast<int>(k1 - k0);
                    for (size_t k = k0; k < k1; k++) fill(a.blocks[k - k0], k);
                    const int t_main = n.members[k1 - 1].out;   // the chain's own output tensor
                    a.write_out = n.out == t_main || std::find(n.extra_out.begin(), n.extra_out.end(), t_main) != n.extra_out.end();
                    if (a.write_out) { a.out = tensor_ptr_mut(t_main, chunk_start, &a.out_fs); } else { a.out = op; a.out_fs = out_fs; }
                    if (n.chain_pre) {
                        a.pre.on = 1; fill(a.pre.blk, 0);
                        a.pre.in = ip; a.pre.in_fs = in_fs; a.pre.Cin = si[3];
                        a.in = ip; a.in_fs = in_fs;
                    } else {
                        a.in = ip; a.in_fs = in_fs;
                    }
                    if (n.chain_post) {
                        const int t_post = n.members.back().out;
                        a.post.on = 1; fill(a.post.blk, n.members.size() - 1);
                        a.post.out = tensor_ptr_mut(t_post, chunk_start, &a.post.out_fs);
                        a.post.Co = g.tensors[t_post].shape[3];
                    }
                    for (size_t k = 0; k < n.head_pairs.size(); k++) {
                        const Node::HeadPair& hp = n.head_pairs[k];
                        ChainHead& H = a.heads[hp.src];
                        H.on = 1; H.src = hp.src;
                        H.w_pw = d_weights_ + chain_head_off_[i][k].w2;
                        H.bias = d_weights_ + chain_head_off_[i][k].b2;
                        const int ta = n.head_nodes[static_cast<size_t>(hp.a)].out;
                        H.Co_a = g.tensors[ta].shape.back();
                        H.out_a = tensor_ptr_mut(ta, chunk_start, &H.out_a_fs);
                        if (hp.b >= 0) {
                            const int tb = n.head_nodes[static_cast<size_t>(hp.b)].out;
                            H.Co_b = g.tensors[tb].shape.back();
                            H.out_b = tensor_ptr_mut(tb, chunk_start, &H.out_b_fs);
                        }
                    }
                    if (!chain_kernel_supports(a)) throw std::runtime_error("chain node with edge stages without a kernel");
                    rc = launch_chain(a, s);
                    break;
                }
                a.in = ip; a.out = op; a.in_fs = in_fs; a.out_fs = out_fs;
                a.B = F; a.H = si[1]; a.W = si[2]; a.C = si[3]; a.nblocks = static_cast<int>(n.members.size());
                if (a.nblocks <= kMaxChain && a.H * a.W <= 256 && chain_kernel_supports(a)) {  // frame-resident in LDS
                    for (size_t k = 0; k < n.members.size(); k++) fill(a.blocks[k], k);
                    rc = launch_chain(a, s);
                    break;
                }
                if (strip_ && node_mwalk_[i] >= 0) {   // a pair of plain BlazeBlocks with an operand-layout form
                    DblockArgs d;
                    d.in = ip; d.in_fs = in_fs; d.out = op; d.out_fs = out_fs;
                    d.B = F; d.H = si[1]; d.W = si[2]; d.C = si[3]; d.Cm = g.tensors[n.members[0].out].shape[3]; d.Co = so[3];
                    d.hi1 = n.members[0].act == ACT_RELU6 ? 6.f : INFINITY;
                    d.hi2 = n.members[1].act == ACT_RELU6 ? 6.f : INFINITY;
                    d.skip1 = 1; d.skip2_from_a = 1;
                    d.act1 = n.members[0].act; d.act2 = n.members[1].act;
                    d.mconsts = d_weights_ + node_mwalk_[i];
                    d.band_rows = mdb_band_;
                    if (mdblock_kernel_supports(d)) {
                        if (labels) labels->back() = "mdblock_kernel<pair>";
                        rc = launch_mdblock(d, s);
                        break;
                    }
                }
                // row-pipelined group of strip blocks: only the first input and the last output exist in memory
                std::vector<BlockArgs> blk(n.members.size());
                for (size_t k = 0; k < n.members.size(); k++) {
                    const MemberOff& mo = chain_off_[i][k];
                    const Node& m = n.members[k];
                    BlockArgs& b = blk[k];
                    b.in = ip; b.out = op; b.in_fs = in_fs; b.out_fs = out_fs;
                    b.has_dw = 1;
                    b.pipe_rows = pipe_rows_;
                    b.pipe_band = pipe_band_;
                    b.w_dw = d_weights_ + mo.w;
                    b.b_dw = mo.b >= 0 ? d_weights_ + mo.b : nullptr;
                    b.w_pw = d_weights_ + mo.w2;
                    b.w_strip = mo.strip >= 0 ? d_weights_ + mo.strip : nullptr;
                    b.B = F; b.H = si[1]; b.W = si[2]; b.C = si[3]; b.Ho = si[1]; b.Wo = si[2]; b.Co = si[3];
                    b.sh = b.sw = 1; b.pt = b.pl = 1;
                    b.ep.bias = mo.b2 >= 0 ? d_weights_ + mo.b2 : nullptr;
                    b.ep.alpha = mo.alpha >= 0 ? d_weights_ + mo.alpha : nullptr;
                    b.ep.act = m.act;
                    if (m.res >= 0) { b.ep.res = b.in; b.ep.res_fs = b.in_fs; b.ep.res_C = b.C; b.ep.res_mode = RES_DIRECT; }
                    if (m.sh == 2) {  // stride-2 tail: halves the resolution, 2x2 max-pool skip from its (never materialised) input
                        b.sh = b.sw = 2; b.pt = b.pl = 0;
                        b.Ho = so[1]; b.Wo = so[2]; b.Co = so[3];
                        if (m.res >= 0) { b.ep.res_mode = RES_MAXPOOL; b.ep.res_H = b.H; b.ep.res_W = b.W; }
                    }
                }
                // Small batches: a row pipeline is a chain of 2S + rows/2 dependent steps of ~5 us whatever the batch (88 us per launch for ONE
                // BackCamera frame, four such launches of its 0.6 ms), while one strip-kernel launch per block is hundreds of independent
                // waves (~5 us).  Below `small_chain_` frames the members run one launch each, ping-ponging through a per-handle scratch.
                if (small_chain_ > 0 && F <= small_chain_ && lanes_ == 1 && d_small_) {
                    const int nb = static_cast<int>(blk.size());
                    const long fsz = static_cast<long>(si[1]) * si[2] * si[3];
                    float* T[2] = {d_small_, d_small_ + static_cast<size_t>(small_chain_) * fsz};
                    std::vector<BlockArgs> sb = blk;
                    const float* cur = ip;
                    long cur_fs = in_fs;
                    bool ok = static_cast<size_t>(2 * small_chain_) * fsz <= small_floats_;
                    for (int k = 0; k < nb && ok; k++) {
                        BlockArgs& b = sb[static_cast<size_t>(k)];
                        b.in = cur; b.in_fs = cur_fs;
                        if (b.ep.res_mode != RES_NONE) { b.ep.res = cur; b.ep.res_fs = cur_fs; }
                        if (k == nb - 1) { b.out = op; b.out_fs = out_fs; } else { b.out = T[k & 1]; b.out_fs = fsz; }
                        ok = b.sh == 1 ? strip_kernel_supports(b) : block_kernel_supports(b);   // (the stride-2 member too, or the run fails where the row pipeline below would have taken it)
                        cur = b.out; cur_fs = b.out_fs;
                    }
                    if (ok) {
                        if (labels) {
                            char name[48], buf[112];
                            const bool s2 = sb[static_cast<size_t>(nb - 1)].sh == 2;
                            snprintf(buf, sizeof buf, "%s x%d%s (small batch)", strip_kernel_label(sb[0], name, sizeof name), s2 ? nb - 1 : nb, s2 ? " + block_kernel" : "");
                            labels->back() = buf;
                        }
                        for (int k = 0; k < nb && rc == 0; k++) rc = sb[static_cast<size_t>(k)].sh == 1 ? launch_strip(sb[static_cast<size_t>(k)], s) : launch_block(sb[static_cast<size_t>(k)], s);
                        break;
                    }
                }
                if (!strip_pipe_supports(blk.data(), static_cast<int>(blk.size()))) throw std::runtime_error("chain node without a kernel");
                if (labels) { char buf[96]; labels->back() = strip_pipe_label(blk.data(), static_cast<int>(blk.size()), buf, sizeof buf); }
                rc = launch_strip_pipe(blk.data(), static_cast<int>(blk.size()), s);
                break;
            }
            case Node::Block: {
                BlockArgs a;
                a.in = ip; a.out = op; a.in_fs = in_fs; a.out_fs = out_fs;
                a.has_dw = n.w >= 0;
                a.w_dw = a.has_dw ? d_weights_ + node_w_[i] : nullptr;
                a.b_dw = node_b_[i] >= 0 ? d_weights_ + node_b_[i] : nullptr;
                a.w_pw = d_weights_ + node_w2_[i];
                a.B = F; a.H = si[1]; a.W = si[2]; a.C = si[3]; a.Ho = so[1]; a.Wo = so[2]; a.Co = so[3];
                a.sh = n.sh; a.sw = n.sw;
                if (a.has_dw && n.padding == Padding::Same) { same_pad(a.H, 3, a.sh, a.Ho, &a.pt); same_pad(a.W, 3, a.sw, a.Wo, &a.pl); }
                if (a.has_dw && n.ept >= 0) { a.pt = n.ept; a.pl = n.epl; }
                a.ep = ep;
                a.w_strip = node_strip_[i] >= 0 ? d_weights_ + node_strip_[i] : nullptr;
                a.w_mwalk = node_mwalk_[i] >= 0 ? d_weights_ + node_mwalk_[i] : nullptr;
                // this block and the next one as ONE launch (mdblock_kernel, pair form): the tensor between them is neither written nor read
                if (strip_ && pair_fuse_ && node_pair_[i] >= 0 && i + 1 < plan_.nodes.size() && !event_after_[i] && head_slot_[i + 1] < 0 && head_slot_[i] < 0) {
                    const Node& nb = plan_.nodes[i + 1];
                    {
                        DblockArgs d;
                        long ofs = 0;
                        d.in = ip; d.in_fs = in_fs;
                        d.out = tensor_ptr_mut(nb.out, chunk_start, &ofs); d.out_fs = ofs;
                        d.B = F; d.H = si[1]; d.W = si[2]; d.C = si[3]; d.Cm = si[3]; d.Co = si[3];
                        d.hi1 = n.act == ACT_RELU6 ? 6.f : INFINITY;
                        d.hi2 = nb.act == ACT_RELU6 ? 6.f : INFINITY;
                        d.skip1 = 1; d.skip2_from_a = 1;
                        d.act1 = n.act; d.act2 = nb.act;
                        d.mconsts = d_weights_ + node_pair_[i];
                        d.band_rows = mdb_band_;
                        // (the launch reads its input while it writes its output: the arena keeps the two apart — plan.cpp, liveness — and this checks it)
                        const bool apart = d.out + d.out_fs * F <= d.in || d.in + d.in_fs * F <= d.out;
                        if (apart && mdblock_kernel_supports(d)) {
                            if (labels) labels->back() = "mdblock_kernel<pair>";
                            rc = launch_mdblock(d, s);
                            fused_behind = 1;
                            break;
                        }
                    }
                }
                if (strip_ && ms2_kernel_supports(a)) {
                    if (labels) { char buf[96]; labels->back() = ms2_kernel_label(a, buf, sizeof buf); }
                    rc = launch_ms2(a, s);
                    break;
                }
                if (strip_ && mwalk_kernel_supports(a)) {
                    if (labels) { char buf[96]; labels->back() = mwalk_kernel_label(a, buf, sizeof buf); }
                    rc = launch_mwalk(a, s);
                    break;
                }
                const bool strip = strip_ && strip_kernel_supports(a);
                const bool mstrip = strip_ && !strip && mstrip_kernel_supports(a);
                if (mstrip && mchain_ && lanes_ == 1) {
                    // the blocks behind this one that the same kernel takes, each reading its predecessor's output: ONE launch for the run
                    // (every intermediate tensor keeps its arena slot; a workgroup per frame walks through the blocks)
                    std::vector<BlockArgs> run{a};
                    for (size_t j = i + 1; j < plan_.nodes.size() && run.size() < 8; j++) {
                        const Node& m = plan_.nodes[j];
                        if (m.kind != Node::Block || m.w < 0 || m.in.size() != 1 || m.in[0] != plan_.nodes[j - 1].out || node_strip_[j] < 0 || head_slot_[j] >= 0) break;
                        const auto& mi_ = g.tensors[m.in[0]].shape;
                        const auto& mo_ = g.tensors[m.out].shape;
                        if (mi_.size() != 4 || mo_ != mi_) break;
                        BlockArgs bb;
                        bb.in = tensor_ptr(m.in[0], in, chunk_start, &bb.in_fs);
                        bb.out = tensor_ptr_mut(m.out, chunk_start, &bb.out_fs);
                        bb.has_dw = 1;
                        bb.w_dw = d_weights_ + node_w_[j];
                        bb.b_dw = node_b_[j] >= 0 ? d_weights_ + node_b_[j] : nullptr;
                        bb.w_pw = d_weights_ + node_w2_[j];
                        bb.w_strip = d_weights_ + node_strip_[j];
                        bb.B = F; bb.H = mi_[1]; bb.W = mi_[2]; bb.C = mi_[3]; bb.Ho = mo_[1]; bb.Wo = mo_[2]; bb.Co = mo_[3];
                        bb.sh = m.sh; bb.sw = m.sw;
                        if (m.padding == Padding::Same) { same_pad(bb.H, 3, bb.sh, bb.Ho, &bb.pt); same_pad(bb.W, 3, bb.sw, bb.Wo, &bb.pl); }
                        if (m.ept >= 0) break;
                        bb.ep.bias = node_b2_[j] >= 0 ? d_weights_ + node_b2_[j] : nullptr;
                        bb.ep.alpha = node_alpha_[j] >= 0 ? d_weights_ + node_alpha_[j] : nullptr;
                        bb.ep.act = m.act;
                        if (m.res >= 0) {
                            if (m.res != m.in[0] || m.res_mode != RES_DIRECT || m.res_after) break;
                            bb.ep.res = bb.in; bb.ep.res_fs = bb.in_fs; bb.ep.res_mode = RES_DIRECT; bb.ep.res_C = bb.C;
                            bb.ep.res_H = bb.H; bb.ep.res_W = bb.W;
                        }
                        run.push_back(bb);
                        if (!mstrip_chain_supports(run.data(), static_cast<int>(run.size()))) { run.pop_back(); break; }
                    }
                    if (run.size() >= 2) {
                        if (labels) { char buf[96]; snprintf(buf, sizeof buf, "mstrip_chain_kernel<%d,%d>", a.C / 4, a.ep.act == ACT_RELU ? 1 : 0); labels->back() = buf; }
                        rc = launch_mstrip_chain(run.data(), static_cast<int>(run.size()), s);
                        fused_behind = static_cast<int>(run.size()) - 1;
                        break;
                    }
                }
                if (labels) { char buf[96]; labels->back() = strip ? strip_kernel_label(a, buf, sizeof buf) : (mstrip ? mstrip_kernel_label(a, buf, sizeof buf) : block_kernel_label(a, buf, sizeof buf)); }
                rc = strip ? launch_strip(a, s) : (mstrip ? launch_mstrip(a, s) : launch_block(a, s));
                break;
            }
            default: {
                EltArgs a;
                a.a = ip; a.a_fs = in_fs; a.out = op; a.out_fs = out_fs; a.alpha = ep.alpha; a.act = n.act;
                a.B = F; a.H = dim(si, 1); a.W = dim(si, 2); a.C = si.back();
                a.Ho = dim(so, 1); a.Wo = dim(so, 2); a.Co = so.back();
                if (si.size() != 4) { a.H = 1; a.W = 1; a.C = static_cast<int>(g.tensors[n.in[0]].elems()); a.Ho = a.Wo = 1; a.Co = a.C; }
                if (n.kind == Node::Add) {
                    a.b = tensor_ptr(n.in[1], in, chunk_start, &a.b_fs);
                    rc = launch_add(a, s);
                } else if (n.kind == Node::Act) {
                    rc = launch_act(a, s);
                } else if (n.kind == Node::MaxPool) {
                    a.p0 = n.filter_h; a.p1 = n.filter_w; a.p2 = n.sh; a.p3 = n.sw;
                    rc = launch_maxpool(a, s);
                } else if (n.kind == Node::Pad) {
                    const auto& pv = g.tensors[n.pads].i32;
                    if (pv[0] != 0 || pv[1] != 0) throw std::runtime_error("PAD on the batch axis unsupported");
                    a.p0 = pv[2]; a.p1 = pv[4]; a.p2 = pv[6];
                    rc = launch_padc(a, s);
                } else if (n.kind == Node::Resize) {
                    a.p0 = n.half_pixel; a.p1 = n.align_corners;
                    rc = launch_resize2x(a, s);
                } else if (n.kind == Node::DepthToSpace) {
                    a.p0 = n.block_size;
                    rc = launch_depth_to_space(a, s);
                } else {
                    throw std::runtime_error("internal: unhandled node kind");
                }
            }
        }
        if (rc != 0) throw std::runtime_error(std::string("kernel launch failed: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        if (fork && event_after_[i]) hip_check(record_event(node_event(i), s), "hipEventRecord");
        mark();
    }
    s = trunk;
    for (int k = 0; k < kHeadStreams; k++)  // join: the trunk stream continues (post-processing, the next chunk) after every head
        if (used_heads & (1u << k)) {
            hip_check(record_event(node_event(plan_.nodes.size() + 1 + k), head_streams_[static_cast<size_t>(k)]), "hipEventRecord");
            hip_check(wait_event(trunk, node_event(plan_.nodes.size() + 1 + k)), "hipStreamWaitEvent");
        }
    last_chunk_frames_ = F;
}

void Model::enqueue_all(const float* in, int batch, hipStream_t s) {
    if (lanes_ > 1 && batch > chunk_cap_) {
        // Independent frame ranges on concurrent streams: while one lane is in its small, latency-bound tail layers the
        // other lanes' large layers keep the CUs busy.  Fork/join with events (captured as parallel graph branches).
        while (static_cast<int>(side_streams_.size()) < lanes_ - 1) {
            hipStream_t st;
            hip_check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
            side_streams_.push_back(st);
        }
        while (static_cast<int>(lane_events_.size()) < lanes_) {
            hipEvent_t ev;
            hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
            lane_events_.push_back(ev);
        }
        hip_check(record_event(lane_events_[0], s), "hipEventRecord");
        int lane = 0;
        for (int start = 0; start < batch; start += chunk_cap_, lane = (lane + 1) % lanes_) {
            hipStream_t ls = lane == 0 ? s : side_streams_[lane - 1];
            if (lane > 0 && start < chunk_cap_ * lanes_) hip_check(wait_event(ls, lane_events_[0]), "hipStreamWaitEvent");
            arena_lane_ = lane;
            enqueue_chunk(in, start, std::min(chunk_cap_, batch - start), ls);
        }
        arena_lane_ = 0;
        for (int l = 1; l < lanes_; l++) {
            hip_check(record_event(lane_events_[l], side_streams_[l - 1]), "hipEventRecord");
            hip_check(wait_event(s, lane_events_[l]), "hipStreamWaitEvent");
        }
        return;
    }
    int chunk = chunk_cap_;
    for (int start = 0; start < batch; start += chunk) enqueue_chunk(in, start, std::min(chunk, batch - start), s);
}

bool Model::takes_u8_input() {
    if (dirty_) rebuild();
    const Graph& g = plan_.graph;
    const int root = plan_.storage[g.inputs[0]].root;
    int readers = 0;
    bool stem = false;
    for (const Node& n : plan_.nodes) {
        if (n.kind == Node::Reshape || n.kind == Node::Concat) continue;
        bool reads = false;
        for (int t : n.in) reads |= t >= 0 && plan_.storage[t].root == root;
        if (n.res >= 0 && plan_.storage[n.res].root == root) reads = true;
        if (!reads) continue;
        readers++;
        if (n.kind == Node::Conv && !n.gemm_head && n.in[0] >= 0 && plan_.storage[n.in[0]].root == root) {
            const auto& si = g.tensors[n.in[0]].shape;
            const auto& so = g.tensors[n.out].shape;
            ConvArgs a;
            a.out = reinterpret_cast<float*>(uintptr_t{256}); a.out_fs = 4;  // aligned placeholders: only the shape tests matter here
            a.C = si[3]; a.Co = so[3]; a.KH = n.KH; a.KW = n.KW; a.sh = n.sh; a.sw = n.sw;
            static const float some_bias = 0.f;
            a.ep.bias = &some_bias;
            a.ep.res_mode = n.res >= 0 ? n.res_mode : RES_NONE;
            stem = conv_takes_u8(a) && node_b_[&n - plan_.nodes.data()] >= 0;
        }
    }
    return readers == 1 && stem;
}

void Model::run_device_u8(const uint8_t* frames, long frame_bytes, int row_bytes, const float* lut, int batch, hipStream_t stream) {
    if (batch <= 0) throw std::runtime_error("batch must be positive");
    if (!frames || !lut) throw std::runtime_error("null tensor pointer");
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (!takes_u8_input()) throw std::runtime_error("plan: this graph has no u8 input form");
    ensure_capacity(batch);
    u8_.frames = frames; u8_.lut = lut; u8_.frame_bytes = frame_bytes; u8_.row_bytes = row_bytes;
    try {
        // the graph input tensor itself is never read: `in` only has to be a stable non-null key
        run_graph_or_eager(reinterpret_cast<const float*>(frames), batch, stream ? stream : stream_, GraphKey{frames, batch, frame_bytes, row_bytes});
    } catch (...) {
        u8_ = U8Input{};
        throw;
    }
    u8_ = U8Input{};
}

void Model::run_device(const float* in, int batch, hipStream_t stream, bool one_shot) {
    if (batch <= 0) throw std::runtime_error("batch must be positive");
    hip_check(hipSetDevice(device_), "hipSetDevice");
    if (dirty_) rebuild();
    ensure_capacity(batch);
    band_use_ = band_usable(batch) && (band_ == 2 || one_shot);
    band_ran_ = band_use_;
    GraphKey key{in, batch, 0, 0};
    key.band = band_use_;
    hipStream_t s = stream ? stream : stream_;
    if (band_use_) band_before_launch(s);
    try {
        run_graph_or_eager(in, batch, s, key);
    } catch (...) {
        band_use_ = false;
        throw;
    }
    const bool checked_here = band_use_ && !one_shot;
    band_use_ = false;
    if (checked_here) {
        // option band = 2 (tests, profiling, bench --opt): nobody behind this call asks band_failed(), so the run is checked here — a launch that gave
        // up leaves void results: wait for it and repeat the run on the batched plan
        hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        if (band_failed()) {
            key.band = false;
            run_graph_or_eager(in, batch, s, key);
        }
    }
}

// In front of every single-launch run: the packet tags are 64 x generation + stage + 1 in 32 bits, and the packet slots of frames a launch does not
// have keep their old tags — after 2^26 launches of a handle (hours of single-image calls) a stale packet could carry the tag a reader waits for.
// Well before that the workspace and the generation are cleared (behind everything the stream holds).
void Model::band_before_launch(hipStream_t s) {
    constexpr unsigned kWrap = (1u << 26) - 4096u;   // (Model::profile repeats a launch a few times per call)
    band_gen_ += static_cast<unsigned>(std::max(1, profile_inner_));
    if (band_gen_ < kWrap || !d_band_ws_) return;
    hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    hip_check(hipMemsetAsync(d_band_ws_, 0, band_ws_bytes_, s), "hipMemsetAsync band workspace");
    if (!band_gen_force_) hip_check(hipMemsetAsync(d_band_sync_, 0, 64 * sizeof(unsigned), s), "hipMemsetAsync band generation");
    else {   // test hook: the device's generation is moved to the same count, so that the real tags do wrap in the launches that follow
        const unsigned g[4] = {kWrap + 4096u - 2u, 0u, 0u, 0u};
        hip_check(hipMemcpyAsync(d_band_sync_, g, sizeof(g), hipMemcpyHostToDevice, s), "band generation");
        band_gen_force_ = false;
    }
    hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    band_gen_ = 0;
    band_wraps_++;
}

bool Model::band_usable(int batch) const {
    return band_ready_ && band_ > 0 && lanes_ == 1 && batch <= band_max_frames_ && batch <= chunk_cap_;
}

int Model::band_workgroups(int batch) {
    if (dirty_) rebuild();
    return band_ready_ && band_ > 0 && lanes_ == 1 && batch <= band_max_frames_ ? batch * band_nw_used_ : 0;
}

bool Model::band_failed() {
    if (band_test_fail_ && band_ran_) { band_test_fail_ = false; band_ran_ = false; return true; }
    const bool ran = band_ran_;
    band_ran_ = false;
    if (!h_band_fail_ || !*h_band_fail_) {
        if (ran) band_fail_streak_ = 0;
        return false;
    }
    *h_band_fail_ = 0;
    // a handle whose single launches keep giving up (other processes' kernels on the CUs, a CU mask) stops trying: every failed call costs the
    // kernel's bounded wait on top of the batched plan it then runs anyway
    if (++band_fail_streak_ >= 3 && band_ > 0) {
        band_ = 0;
        band_disabled_ = true;
        invalidate_graphs();
    }
    return true;
}

void Model::run_graph_or_eager(const float* in, int batch, hipStream_t s, const GraphKey& key) {
    // A single-launch run that is two launches in all (first convolution + band program) goes out eagerly: replaying a two-node graph costs
    // more than it saves (tools/probes/graph_probe.py: FaceDetection::infer BackCamera 252 -> 242 us, Short 170 -> 160; with the face mesh's
    // two more launches behind the program the graph wins again, 191 against 200 us)
    bool eager = !use_graph_;
    if (band_use_ && !eager) {
        int launches = 1;
        for (size_t i = 0; i < plan_.nodes.size(); i++) {
            const Node& n = plan_.nodes[i];
            if (n.kind == Node::Reshape || n.kind == Node::Concat) continue;
            if (static_cast<int>(i) < band_first_ || band_node_runs_[i]) launches++;
        }
        eager = launches <= 2;
    }
    if (eager) {
        enqueue_all(in, batch, s);
        return;
    }
    auto it = graphs_.find(key);
    if (it == graphs_.end()) {
        if (graphs_.size() > 16) invalidate_graphs();
        // The plan becomes a hipGraph built node by node (launch.hpp) — no stream capture, which is process-wide runtime
        // state that other host threads break (and are broken by).
        GraphRecorder rec;
        hip_check(hipGraphCreate(&rec.graph, 0), "hipGraphCreate");
        current_recorder() = &rec;
        try {
            enqueue_all(in, batch, s);
        } catch (...) {
            current_recorder() = nullptr;
            hipGraphDestroy(rec.graph);
            throw;
        }
        current_recorder() = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipGraphInstantiate(&exec, rec.graph, nullptr, nullptr, 0);
        hipGraphDestroy(rec.graph);
        hip_check(e, "hipGraphInstantiate");
        it = graphs_.emplace(key, exec).first;
    }
    hip_check(hipGraphLaunch(it->second, s), "hipGraphLaunch");
}

void Model::run(const float* in, int batch, float* const* outs, int mem, hipStream_t stream) {
    if (!in || !outs) throw std::runtime_error("null tensor pointer");
    hip_check(hipSetDevice(device_), "hipSetDevice");
    hipStream_t s = stream ? stream : stream_;
    const float* din = in;
    if (mem == 0) {
        size_t need = input_elems() * static_cast<size_t>(batch);
        if (need > in_stage_floats_) {
            invalidate_graphs();
            if (d_in_stage_) hip_check(hipFree(d_in_stage_), "hipFree stage");
            d_in_stage_ = nullptr;
            hip_check(hipMalloc(reinterpret_cast<void**>(&d_in_stage_), need * sizeof(float)), "hipMalloc stage");
            in_stage_floats_ = need;
        }
        hip_check(hipMemcpyAsync(d_in_stage_, in, need * sizeof(float), hipMemcpyHostToDevice, s), "H2D input");
        din = d_in_stage_;
    }
    run_device(din, batch, s);
    for (int k = 0; k < num_outputs(); k++) {
        if (!outs[k]) continue;
        hip_check(hipMemcpyAsync(outs[k], d_out_[k], output_elems(k) * sizeof(float) * batch,
                                 mem == 0 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s), "copy outputs");
    }
    if (mem == 0 || !stream) hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
}

size_t Model::debug_tensor(int tensor, int frame, float* dst, size_t cap) {
    const Graph& g = plan_.graph;
    if (tensor < 0 || tensor >= static_cast<int>(g.tensors.size())) throw std::runtime_error("tensor index out of range");
    if (frame < 0 || frame >= last_chunk_frames_) throw std::runtime_error("frame outside the last chunk");
    const Storage& st = plan_.storage[tensor];
    bool produced = false;
    for (const Node& n : plan_.nodes) produced |= (n.out == tensor);
    if (!produced) throw std::runtime_error("tensor was fused away (or is an input/constant)");
    long fs;
    float* p = tensor_ptr_mut(tensor, 0, &fs);
    size_t n = std::min(cap, g.tensors[tensor].elems());
    hip_check(hipDeviceSynchronize(), "sync");
    hip_check(hipMemcpy(dst, p + static_cast<long>(frame) * fs, n * sizeof(float), hipMemcpyDeviceToHost), "D2H debug");
    (void)st;
    return n;
}

}  // namespace mi
