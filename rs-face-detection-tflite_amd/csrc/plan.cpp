// plan.cpp — graph rewriting + activation-arena planning (see plan.hpp).
#include "plan.hpp"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <sstream>
#include <set>
#include <functional>
#include <stdexcept>

namespace mi {
namespace {

int act_of(FusedAct a) {
    switch (a) {
        case FusedAct::None: return ACT_NONE;
        case FusedAct::Relu: return ACT_RELU;
        case FusedAct::Relu6: return ACT_RELU6;
    }
    throw std::runtime_error("plan: unsupported fused activation");
}

struct Rewriter {
    Graph& g;
    std::vector<Node>& nodes;

    std::vector<int> consumers(int t) const {
        std::vector<int> c;
        for (size_t i = 0; i < nodes.size(); i++) {
            if (nodes[i].dead) continue;
            for (int x : nodes[i].in)
                if (x == t) c.push_back(static_cast<int>(i));
            if (nodes[i].res == t) c.push_back(static_cast<int>(i));
        }
        return c;
    }
    int producer(int t) const {
        for (size_t i = 0; i < nodes.size(); i++)
            if (!nodes[i].dead && nodes[i].out == t) return static_cast<int>(i);
        return -1;
    }
    bool is_graph_output(int t) const { return std::find(g.outputs.begin(), g.outputs.end(), t) != g.outputs.end(); }
    const std::vector<int>& shape(int t) const { return g.tensors[t].shape; }

    // Fold ADD / activation that exclusively consume node i's output into node i; the fused node takes the
    // position of the last folded op so that every operand (e.g. the skip tensor) is already available there.
    void fuse_epilogues() {
        for (size_t start = 0; start < nodes.size(); start++) {
            for (size_t i = start;;) {
                Node& n = nodes[i];
                if (n.dead || !(n.kind == Node::Conv || n.kind == Node::Dw)) break;
                if (is_graph_output(n.out)) break;
                std::vector<int> cons = consumers(n.out);
                if (cons.size() != 1) break;
                int ci = cons[0];
                if (ci <= static_cast<int>(i)) break;
                Node& c = nodes[ci];
                Node fused = n;
                if (c.kind == Node::Add && n.res < 0 && n.act == ACT_NONE && c.in.size() == 2) {
                    int other = c.in[0] == n.out ? c.in[1] : c.in[0];
                    if (other == n.out || g.tensors[other].is_const) break;
                    if (shape(other) != shape(n.out)) break;
                    fused.res = other;
                    fused.res_mode = RES_DIRECT;
                    fused.act = c.act;
                } else if (c.kind == Node::Add && n.kind == Node::Conv && n.res < 0 && n.act != ACT_NONE && c.act == ACT_NONE && c.in.size() == 2) {
                    // a convolution with a FUSED activation, then ADD: the skip joins behind the activation
                    int other = c.in[0] == n.out ? c.in[1] : c.in[0];
                    if (other == n.out || g.tensors[other].is_const) break;
                    if (shape(other) != shape(n.out)) break;
                    fused.res = other;
                    fused.res_mode = RES_DIRECT;
                    fused.res_after = true;
                } else if (c.kind == Node::Act && n.act == ACT_NONE) {
                    fused.act = c.act;
                    fused.alpha = c.alpha;
                } else {
                    break;
                }
                fused.out = c.out;
                fused.src_ops.insert(fused.src_ops.end(), c.src_ops.begin(), c.src_ops.end());
                n.dead = true;
                nodes[ci] = fused;
                i = static_cast<size_t>(ci);  // continue folding from the new position
                if (fused.res_after) break;   // nothing folds behind a post-activation skip
            }
        }
        // Skip-path simplification: res <- PAD(channels) <- MAX_POOL 2x2/2 | RESIZE x2, each with a single consumer.
        for (size_t i = 0; i < nodes.size(); i++) {
            Node& n = nodes[i];
            if (n.dead || n.res < 0 || n.res_mode != RES_DIRECT) continue;
            int p = producer(n.res);
            if (p >= 0 && nodes[p].kind == Node::Pad && consumers(n.res).size() == 1 && !is_graph_output(n.res)) {
                const Node& pd = nodes[p];
                const auto& pv = g.tensors[pd.pads].i32;
                const auto& si = shape(pd.in[0]);
                bool channel_only = pv.size() == 8 && pv[0] == 0 && pv[1] == 0 && pv[2] == 0 && pv[3] == 0 && pv[4] == 0 &&
                                    pv[5] == 0 && pv[6] == 0 && si.size() == 4;
                if (channel_only) {
                    n.res = pd.in[0];
                    n.src_ops.insert(n.src_ops.end(), pd.src_ops.begin(), pd.src_ops.end());
                    nodes[p].dead = true;
                    p = producer(n.res);
                }
            }
            if (p >= 0 && consumers(n.res).size() == 1 && !is_graph_output(n.res)) {
                const Node& q = nodes[p];
                const auto& si = shape(q.in.empty() ? n.res : q.in[0]);
                const auto& so = shape(n.res);
                if (q.kind == Node::MaxPool && q.filter_h == 2 && q.filter_w == 2 && q.sh == 2 && q.sw == 2 && q.act == ACT_NONE &&
                    si.size() == 4 && si[1] == 2 * so[1] && si[2] == 2 * so[2]) {
                    n.res = q.in[0];
                    n.res_mode = RES_MAXPOOL;
                    n.src_ops.insert(n.src_ops.end(), q.src_ops.begin(), q.src_ops.end());
                    nodes[p].dead = true;
                } else if (q.kind == Node::Resize && q.half_pixel && !q.align_corners && si.size() == 4 && so[1] == 2 * si[1] &&
                           so[2] == 2 * si[2]) {
                    n.res = q.in[0];
                    n.res_mode = RES_UP2X;
                    n.src_ops.insert(n.src_ops.end(), q.src_ops.begin(), q.src_ops.end());
                    nodes[p].dead = true;
                }
            }
        }
    }

    // Spatial PAD (zeros) in front of VALID convolutions — how the MLIR-converted full_range_sparse graph writes every
    // convolution — folds into its consumers: as SAME when the amounts are TF's SAME amounts for that consumer (3x3 stride 1 with
    // 1/1/1/1), as explicit pads before the first row / column otherwise (the kernels bound-check the far side).
    void fold_spatial_pads() {
        for (size_t i = 0; i < nodes.size(); i++) {
            Node& pd = nodes[i];
            if (pd.dead || pd.kind != Node::Pad || is_graph_output(pd.out)) continue;
            const auto& pv = g.tensors[pd.pads].i32;
            const auto& si = shape(pd.in[0]);
            if (pv.size() != 8 || si.size() != 4 || pv[0] || pv[1] || pv[6] || pv[7]) continue;   // batch / channel pads: not this rule
            const int pt = pv[2], pb = pv[3], pl = pv[4], pr = pv[5];
            if (pt < 0 || pb < 0 || pl < 0 || pr < 0 || pt > 1 || pl > 1) continue;             // the kernels keep a one-pixel border
            std::vector<int> cons = consumers(pd.out);
            if (cons.empty()) continue;
            bool ok = true;
            for (int ci : cons) {
                const Node& c = nodes[ci];
                ok &= (c.kind == Node::Conv || c.kind == Node::Dw) && c.padding == Padding::Valid && c.in.size() == 1 && c.in[0] == pd.out && c.res != pd.out && c.ept < 0;
            }
            if (!ok) continue;
            for (int ci : cons) {
                Node& c = nodes[ci];
                const auto& so = shape(c.out);
                int bt, bl, oh, ow;
                same_pad_of(si[1], c.KH, c.sh, bt, oh);
                same_pad_of(si[2], c.KW, c.sw, bl, ow);
                const int tot_h = std::max(0, (oh - 1) * c.sh + c.KH - si[1]), tot_w = std::max(0, (ow - 1) * c.sw + c.KW - si[2]);
                c.in[0] = pd.in[0];
                if (oh == so[1] && ow == so[2] && bt == pt && bl == pl && tot_h - bt == pb && tot_w - bl == pr) {
                    c.padding = Padding::Same;
                } else {
                    c.ept = pt; c.epl = pl;
                }
                c.src_ops.insert(c.src_ops.begin(), pd.src_ops.begin(), pd.src_ops.end());
            }
            // the PAD op is executed by its consumers; with several consumers the op index is listed once (the first)
            for (size_t k = 1; k < cons.size(); k++) {
                Node& c = nodes[cons[k]];
                c.src_ops.erase(c.src_ops.begin(), c.src_ops.begin() + static_cast<long>(pd.src_ops.size()));
            }
            pd.dead = true;
        }
    }
    static void same_pad_of(int in, int k, int stride, int& before, int& out) {
        out = (in + stride - 1) / stride;
        const int total = std::max(0, (out - 1) * stride + k - in);
        before = total / 2;
    }

    // Would the fused MFMA kernel take this node? (shape-level check with placeholder, 16-byte aligned pointers)
    bool block_supported(const Node& f) const {
        const auto& si = shape(f.in[0]);
        const auto& so = shape(f.out);
        if (si.size() != 4 || so.size() != 4) return false;
        BlockArgs a;
        a.in = reinterpret_cast<const float*>(0x1000);
        a.out = reinterpret_cast<float*>(0x2000);
        a.in_fs = static_cast<long>(g.tensors[f.in[0]].elems());
        a.out_fs = static_cast<long>(g.tensors[f.out].elems());
        a.B = 1; a.H = si[1]; a.W = si[2]; a.C = si[3]; a.Ho = so[1]; a.Wo = so[2]; a.Co = so[3];
        a.sh = f.sh; a.sw = f.sw;
        a.has_dw = f.w >= 0;
        if (a.has_dw && f.padding == Padding::Same) {
            a.pt = std::max(0, (a.Ho - 1) * a.sh + 3 - a.H) / 2;
            a.pl = std::max(0, (a.Wo - 1) * a.sw + 3 - a.W) / 2;
        }
        if (a.has_dw && f.ept >= 0) { a.pt = f.ept; a.pl = f.epl; }
        if (f.res >= 0) {
            a.ep.res = f.res == f.in[0] ? a.in : reinterpret_cast<const float*>(0x3000);
            a.ep.res_fs = static_cast<long>(g.tensors[f.res].elems());
            a.ep.res_mode = f.res_mode;
            a.ep.res_C = shape(f.res).back();
        }
        return block_kernel_supports(a);
    }

    // DEPTHWISE 3x3 -> CONV 1x1 (stride 1) with the depthwise result consumed only by that conv.
    void fuse_blocks() {
        for (size_t i = 0; i < nodes.size(); i++) {
            Node& d = nodes[i];
            if (d.dead || d.kind != Node::Dw || d.act != ACT_NONE || d.res >= 0 || is_graph_output(d.out)) continue;
            if (d.KH != 3 || d.KW != 3) continue;
            std::vector<int> cons = consumers(d.out);
            if (cons.size() != 1) continue;
            Node& c = nodes[cons[0]];
            if (c.kind != Node::Conv || c.KH != 1 || c.KW != 1 || c.sh != 1 || c.sw != 1 || c.in[0] != d.out) continue;
            if (c.res == d.out) continue;
            Node f = c;
            f.kind = Node::Block;
            f.in = d.in;
            f.w2 = c.w; f.b2 = c.b;
            f.w = d.w; f.b = d.b;
            f.KH = d.KH; f.KW = d.KW; f.sh = d.sh; f.sw = d.sw; f.padding = d.padding; f.ept = d.ept; f.epl = d.epl;
            f.src_ops = d.src_ops;
            f.src_ops.insert(f.src_ops.end(), c.src_ops.begin(), c.src_ops.end());
            if (!block_supported(f)) continue;
            d.dead = true;
            c = f;
        }
        // remaining plain 1x1 stride-1 convolutions also run on the MFMA kernel (no depthwise stage: w = -1)
        for (Node& c : nodes) {
            if (c.dead || c.kind != Node::Conv || c.KH != 1 || c.KW != 1 || c.sh != 1 || c.sw != 1) continue;
            Node f = c;
            f.kind = Node::Block;
            f.w2 = c.w; f.b2 = c.b;
            f.w = -1; f.b = -1;
            if (!block_supported(f)) continue;
            c = f;
        }
    }
};

void same_pad(int in, int k, int stride, int& before, int& out) {
    out = (in + stride - 1) / stride;
    int total = std::max(0, (out - 1) * stride + k - in);
    before = total / 2;
}


// Level 5, step 1: make independent branches contiguous.  TFLite serialises the two heads of the face-mesh / iris graphs
// op by op in alternation; following each producer by its (ready) consumer keeps a branch together so that it can become one
// stage program.  Only the order among independent nodes changes.
void reorder_branches(std::vector<Node>& ns) {
    const int N = static_cast<int>(ns.size());
    std::map<int, int> producer;
    for (int i = 0; i < N; i++) producer[ns[i].out] = i;
    std::vector<char> emitted(N, 0);
    std::vector<int> order;
    auto ready = [&](int k) {
        auto ok = [&](int t) {
            auto it = producer.find(t);
            return t < 0 || it == producer.end() || emitted[it->second];
        };
        for (int t : ns[k].in)
            if (!ok(t)) return false;
        return ok(ns[k].res);
    };
    auto reads = [&](const Node& n, int t) { return std::find(n.in.begin(), n.in.end(), t) != n.in.end() || n.res == t; };
    for (int i = 0; i < N; i++) {
        if (emitted[i]) continue;
        int cur = i;
        emitted[cur] = 1;
        order.push_back(cur);
        for (;;) {
            int next = -1;
            for (int k = cur + 1; k < N && next < 0; k++)
                if (!emitted[k] && reads(ns[k], ns[cur].out) && ready(k)) next = k;
            if (next < 0) break;
            emitted[next] = 1;
            order.push_back(next);
            cur = next;
        }
    }
    std::vector<Node> out;
    out.reserve(ns.size());
    for (int k : order) out.push_back(std::move(ns[k]));
    ns = std::move(out);
}

// Level 5, step 2: one candidate group = plan nodes [i, j].  Builds the stage list and places the in-group activations in
// LDS (first fit by liveness; a block whose skip is its own LDS-resident, now dead, same-shape input tensor is updated in
// place).  Returns false when a node has no stage form or the program does not fit in `budget` bytes of LDS.
bool build_resident(const Graph& g, const std::vector<Node>& ns, size_t i, size_t j, int budget, Node* out) {
    std::vector<Node> M;
    for (size_t k = i; k <= j; k++) {
        const Node& n = ns[k];
        if (n.kind == Node::Chain) {
            const auto& si = g.tensors[n.in[0]].shape;
            if (si[1] * si[2] > 256 || n.chain_pre || n.chain_post || !n.head_pairs.empty()) return false;  // row-pipelined chains and chains with stride-2 edge stages / output heads stay what they are
            for (const Node& m : n.members) M.push_back(m);
        } else if (n.kind == Node::Conv && n.gemm_head) {
            return false;  // a whole-frame convolution is a GEMM over the batch: its weights are read once per 32 frames there, once per frame here
        } else if (n.kind == Node::Block || n.kind == Node::Conv) {
            M.push_back(n);
        } else {
            return false;
        }
    }
    auto shp = [&](int t) -> const std::vector<int>& { return g.tensors[t].shape; };
    std::map<int, int> made;  // tensor -> member that produces it
    for (size_t m = 0; m < M.size(); m++) made[M[m].out] = static_cast<int>(m);
    auto external = [&](int t) {
        if (std::find(g.outputs.begin(), g.outputs.end(), t) != g.outputs.end()) return true;
        for (size_t k = 0; k < ns.size(); k++) {
            if (k >= i && k <= j) continue;
            if (std::find(ns[k].in.begin(), ns[k].in.end(), t) != ns[k].in.end() || ns[k].res == t) return true;
        }
        return false;
    };
    struct Lt { int H, W, C, b = 0, def = -1, last = -1, off = -1; bool in_lds = false; };
    std::map<int, Lt> lt;
    auto geom = [&](int t) -> Lt& {
        auto it = lt.find(t);
        if (it == lt.end()) {
            Lt x;
            x.H = shp(t)[1]; x.W = shp(t)[2]; x.C = shp(t)[3];
            it = lt.emplace(t, x).first;
        }
        return it->second;
    };
    // ---- pass 1: stages with tensor ids
    std::vector<Node::Stage> S;
    std::vector<int> ext_in, ext_out;
    auto add_ext_in = [&](int t) { if (std::find(ext_in.begin(), ext_in.end(), t) == ext_in.end()) ext_in.push_back(t); };
    for (size_t m = 0; m < M.size(); m++) {
        const Node& n = M[m];
        if (n.in.size() != 1 || n.res_after) return false;
        const int src = n.in[0];
        if (shp(src).size() != 4 || shp(n.out).size() != 4) return false;
        const int H = shp(src)[1], W = shp(src)[2], C = shp(src)[3];
        const int Ho = shp(n.out)[1], Wo = shp(n.out)[2], Co = shp(n.out)[3];
        if (C % 4 || Ho * Wo > 256 || H * W > 1024) return false;
        Node::Stage sg;
        ResStage& st = sg.st;
        sg.member = static_cast<int>(m);
        st.src_H = H; st.src_W = W; st.src_C = C;
        st.Ho = Ho; st.Wo = Wo; st.Co = Co;
        st.act = n.act;
        const bool src_inside = made.count(src) && made[src] < static_cast<int>(m);
        if (n.kind == Node::Block && n.w >= 0) {
            if (n.KH != 3 || n.KW != 3 || n.sh != n.sw || (n.sh != 1 && n.sh != 2) || n.padding != Padding::Same) return false;
            st.kind = RES_STAGE_DW;
            st.KH = st.KW = 3; st.S = n.sh; st.Kv = C;
            st.pt = std::max(0, (Ho - 1) * n.sh + 3 - H) / 2;
            st.pl = std::max(0, (Wo - 1) * n.sw + 3 - W) / 2;
            if ((Ho - 1) * n.sh + 2 - st.pt > H || (Wo - 1) * n.sw + 2 - st.pl > W) return false;  // one border pixel covers the overhang
            if (!src_inside && !lt.count(src)) {  // bring the frame into LDS first
                Node::Stage ld;
                ld.st.kind = RES_STAGE_LOAD;
                ld.st.src_H = H; ld.st.src_W = W; ld.st.src_C = C;
                ld.src_t = src;
                ld.dst_t = src;
                Lt& x = geom(src);
                x.in_lds = true; x.def = static_cast<int>(S.size());
                S.push_back(ld);
                add_ext_in(src);
            }
            Lt& x = geom(src);
            if (!x.in_lds) return false;
            x.b = 1;
        } else {
            int K = 1, sh = 1;
            if (n.kind == Node::Conv) {
                if (n.KH != n.KW || n.res >= 0) return false;
                if (n.ept >= 0 || n.epl >= 0) return false;  // a folded spatial PAD: the k x k stride-k gather has no border
                K = n.KH; sh = n.sh;
                const bool whole = H == K && W == K;  // the window is the frame: one output pixel whatever the stride
                if (!whole && (n.sh != K || n.sw != K || H % K || W % K)) return false;
                if (Ho != H / K || Wo != W / K) return false;
                (void)sh;
            } else if (n.kind != Node::Block) {
                return false;
            }
            st.kind = RES_STAGE_GATHER;
            st.KH = st.KW = K; st.S = K; st.Kv = K * K * C;
            if (src_inside || lt.count(src)) {
                if (!geom(src).in_lds) return false;
            } else {
                sg.src_t = src;  // gathered straight from global memory
                add_ext_in(src);
            }
        }
        if (lt.count(src) && lt[src].in_lds) lt[src].last = static_cast<int>(S.size());
        if (n.res >= 0) {
            const auto& sr = shp(n.res);
            if (sr.size() != 4 || sr[3] % 4 || sr[3] > Co) return false;
            if (n.res_mode == RES_DIRECT) { if (sr[1] != Ho || sr[2] != Wo) return false; }
            else if (n.res_mode == RES_MAXPOOL) { if (sr[1] != 2 * Ho || sr[2] != 2 * Wo) return false; }
            else return false;
            st.res_mode = n.res_mode; st.res_C = sr[3]; st.res_H = sr[1]; st.res_W = sr[2];
            if (lt.count(n.res) && lt[n.res].in_lds) {
                lt[n.res].last = static_cast<int>(S.size());
            } else if (made.count(n.res) && made[n.res] < static_cast<int>(m)) {
                return false;  // produced inside but not kept: cannot happen (marked below), defensive
            } else {
                sg.res_t = n.res;
                add_ext_in(n.res);
            }
        }
        // where the output goes
        bool used_inside = false;
        for (size_t m2 = m + 1; m2 < M.size(); m2++)
            used_inside |= std::find(M[m2].in.begin(), M[m2].in.end(), n.out) != M[m2].in.end() || M[m2].res == n.out;
        const bool ext = external(n.out);
        if (!used_inside && !ext) return false;
        if (used_inside) {
            if (Co % 4) return false;
            Lt& y = geom(n.out);
            y.in_lds = true; y.def = static_cast<int>(S.size());
        }
        if (ext) { sg.dst_t = n.out; ext_out.push_back(n.out); }
        S.push_back(sg);
    }
    if (ext_out.empty() || ext_in.size() + ext_out.size() > 12) return false;
    // ---- pass 2: LDS placement in stage order
    struct Live { int off, size, t; };
    std::vector<Live> live;
    struct Clean { int off, size, H, W, C, b; };
    std::vector<Clean> clean;
    int high = 0, const_max = 0;
    auto size_of = [&](const Lt& x) { return (x.H + 2 * x.b) * (x.W + 2 * x.b) * (x.C + 4); };
    for (size_t k = 0; k < S.size(); k++) {
        Node::Stage& sg = S[k];
        ResStage& st = sg.st;
        const int src = sg.member >= 0 ? M[sg.member].in[0] : sg.src_t;
        const int dst = sg.member >= 0 ? M[sg.member].out : sg.dst_t;
        const int res = sg.member >= 0 ? M[sg.member].res : -1;
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return lt[l.t].last < static_cast<int>(k); }), live.end());
        if (st.kind != RES_STAGE_LOAD && lt.count(src) && lt[src].in_lds) {
            const Lt& x = lt[src];
            st.src_off = x.off; st.src_PS = x.C + 4; st.src_b = x.b;
        }
        if (res >= 0 && lt.count(res) && lt[res].in_lds) {
            const Lt& x = lt[res];
            st.res_off = x.off; st.res_PS = x.C + 4; st.res_b = x.b;
        }
        if (lt.count(dst) && lt[dst].in_lds && lt[dst].def == static_cast<int>(k)) {
            Lt& y = lt[dst];
            if (y.last < static_cast<int>(k)) y.last = static_cast<int>(k);
            const int size = size_of(y);
            bool placed = false;
            if (st.kind != RES_STAGE_LOAD && res >= 0 && res != src && st.res_off >= 0 && st.res_mode == RES_DIRECT) {
                const Lt& r = lt[res];
                if (r.last == static_cast<int>(k) && r.H == y.H && r.W == y.W && r.C == y.C && r.b == y.b) {  // in place on the skip tensor
                    y.off = r.off;
                    for (Live& l : live) if (l.t == res) l.t = dst;
                    placed = true;
                }
            }
            if (!placed) {
                std::sort(live.begin(), live.end(), [](const Live& a, const Live& b) { return a.off < b.off; });
                int off = 0;
                for (const Live& l : live) {
                    if (off + size <= l.off) break;
                    off = std::max(off, l.off + l.size);
                }
                y.off = off;
                live.push_back({off, size, dst});
                high = std::max(high, off + size);
                bool same = false;
                for (const Clean& c : clean) same |= c.off == off && c.size == size && c.H == y.H && c.W == y.W && c.C == y.C && c.b == y.b;
                if (!same) {
                    clean.erase(std::remove_if(clean.begin(), clean.end(), [&](const Clean& c) { return c.off < off + size && off < c.off + c.size; }), clean.end());
                    clean.push_back({off, size, y.H, y.W, y.C, y.b});
                    st.zero_dst = size;
                }
            }
            st.dst_off = y.off; st.dst_PS = y.C + 4; st.dst_b = y.b;
        }
        const_max = std::max(const_max, resident_const_floats(st));
    }
    // LDS layout: [activations | constants, two halves (stage s computes from one while s+1's land in the other) | depthwise scratch]
    if (const_max > kResConstMax) return false;
    // Depthwise batches: enough pixel groups per batch to give each of the 8 waves a unit — unless smaller batches let two
    // workgroups share a CU (half of its 160 KB each), which hides more latency than full batches do.
    const int base_floats = ((high + 3) & ~3) + 2 * const_max;
    auto scratch_for = [&](int cap_pg) {
        int mx = 0;
        for (Node::Stage& sg : S) {
            ResStage& st = sg.st;
            if (st.kind != RES_STAGE_DW) continue;
            const int MT = (st.Co + 31) / 32, PGn = (st.Ho * st.Wo + 31) / 32, Cp = (st.Kv + 7) & ~7;
            st.dw_pg = std::max(1, std::min(std::min(PGn, cap_pg), (8 + MT - 1) / MT));
            mx = std::max(mx, st.dw_pg * 32 * (Cp + 4));
        }
        return mx;
    };
    // pointwise stages that read global memory stage their source through per-wave LDS slabs (K-blocked contraction order)
    bool slabs = false;
    for (Node::Stage& sg : S) {
        ResStage& st = sg.st;
        if (st.kind == RES_STAGE_GATHER && st.src_off < 0 && st.KH == 1 && st.Kv % 16 == 0) { st.kblk = 16; slabs = true; }
    }
    const int half_cu = 80 * 1024 / 4;
    int scratch_max = scratch_for(8);
    for (int cap = 4; cap >= 1 && base_floats + scratch_max > half_cu && base_floats + scratch_for(1) <= half_cu; cap /= 2) scratch_max = scratch_for(cap);
    if (base_floats + scratch_max > half_cu) scratch_max = scratch_for(8);
    if (slabs && base_floats + std::max(scratch_max, kResSlabFloats) > half_cu && base_floats + scratch_max <= half_cu) {
        slabs = false;  // the slabs would cost the second workgroup per CU: keep the direct gather here
        for (Node::Stage& sg : S) sg.st.kblk = 0;
    }
    if (slabs) scratch_max = std::max(scratch_max, kResSlabFloats);
    Node r;
    r.kind = Node::Resident;
    r.res_const_off = (high + 3) & ~3;
    r.res_const_floats = const_max;
    const int dw_off = r.res_const_off + 2 * const_max;
    for (Node::Stage& sg : S)
        if (sg.st.kind == RES_STAGE_DW || sg.st.kblk) sg.st.dw_off = dw_off;
    r.res_lds_bytes = (dw_off + scratch_max) * 4;
    if (r.res_lds_bytes > budget) return false;
    r.members = std::move(M);
    r.stages = std::move(S);
    r.in = ext_in;
    r.out = ext_out.back();
    ext_out.pop_back();
    r.extra_out = ext_out;
    for (size_t k = i; k <= j; k++) r.src_ops.insert(r.src_ops.end(), ns[k].src_ops.begin(), ns[k].src_ops.end());
    *out = std::move(r);
    return true;
}


// Level 5, round 5: the same candidate group [i, j] as a stage program of tail_kernels.hip (several frames per workgroup, every LDS
// tensor a dense borderless [pixels][C + 4] array, 16x16x4 MFMAs).  Also takes the frame-resident chains with their stride-2
// neighbours (the face mesh's 24x24 -> 12x12 x2 -> 6x6 and 6x6 x3 -> 3x3 launches): a BlazeBlock updates its input tensor in place, the
// stride-2 block in front reads its taps from global memory.  Placement by liveness, the depthwise scratch included.
bool build_tail(const Graph& g, const std::vector<Node>& ns, size_t i, size_t j, Node* out) {
    std::vector<Node> M;
    for (size_t k = i; k <= j; k++) {
        const Node& n = ns[k];
        if (n.kind == Node::Chain) {
            if (!n.head_pairs.empty()) return false;  // chains with output heads stay on chain_kernels.hip
            const auto& sr = g.tensors[n.members[n.chain_pre ? 1 : 0].in[0]].shape;  // the resident frame
            if (sr.size() != 4 || sr[1] * sr[2] > 256) return false;                  // row-pipelined chains
            for (const Node& m : n.members) M.push_back(m);
        } else if (n.kind == Node::Conv && n.gemm_head) {
            return false;
        } else if (n.kind == Node::Block || n.kind == Node::Conv) {
            M.push_back(n);
        } else {
            return false;
        }
    }
    auto shp = [&](int t) -> const std::vector<int>& { return g.tensors[t].shape; };
    std::map<int, int> made;
    for (size_t m = 0; m < M.size(); m++) made[M[m].out] = static_cast<int>(m);
    auto external = [&](int t) {
        if (std::find(g.outputs.begin(), g.outputs.end(), t) != g.outputs.end()) return true;
        for (size_t k = 0; k < ns.size(); k++) {
            if (k >= i && k <= j) continue;
            if (std::find(ns[k].in.begin(), ns[k].in.end(), t) != ns[k].in.end() || ns[k].res == t) return true;
        }
        return false;
    };
    struct Lt { int H = 0, W = 0, C = 0, def = -1, last = -1, off = -1; bool in_lds = false; };
    std::map<int, Lt> lt;
    auto geom = [&](int t) -> Lt& {
        auto it = lt.find(t);
        if (it == lt.end()) {
            Lt x;
            x.H = shp(t)[1]; x.W = shp(t)[2]; x.C = shp(t)[3];
            it = lt.emplace(t, x).first;
        }
        return it->second;
    };
    std::vector<Node::Stage> S;
    std::vector<int> ext_in, ext_out;
    auto add_ext_in = [&](int t) { if (std::find(ext_in.begin(), ext_in.end(), t) == ext_in.end()) ext_in.push_back(t); };
    auto in_lds = [&](int t) { return lt.count(t) && lt[t].in_lds; };
    auto load = [&](int t) {   // bring an external frame into LDS
        const auto& s = shp(t);
        if (s[3] % 4 || static_cast<long>(s[1]) * s[2] * (s[3] + 4) * 4 > 150 * 1024) return false;
        Node::Stage ld;
        ld.tst.kind = TAIL_LOAD;
        ld.tst.src_H = s[1]; ld.tst.src_W = s[2]; ld.tst.src_C = s[3];
        ld.tst.mHW = tail_magic(s[1] * s[2] * (s[3] / 4)); ld.tst.mW = tail_magic(s[3] / 4);
        ld.src_t = t; ld.dst_t = t;
        Lt& x = geom(t);
        x.in_lds = true; x.def = static_cast<int>(S.size());
        S.push_back(ld);
        add_ext_in(t);
        return true;
    };
    int min_px = 1 << 30;
    for (size_t m = 0; m < M.size(); m++) {
        const Node& n = M[m];
        if (n.in.size() != 1 || n.res_after || n.ept >= 0 || n.epl >= 0) return false;
        const int src = n.in[0];
        if (shp(src).size() != 4 || shp(n.out).size() != 4) return false;
        const int H = shp(src)[1], W = shp(src)[2], C = shp(src)[3];
        const int Ho = shp(n.out)[1], Wo = shp(n.out)[2], Co = shp(n.out)[3];
        if (C % 4 || Ho * Wo > 256 || H * W > 1024) return false;
        Node::Stage sg;
        TailStage& st = sg.tst;
        sg.member = static_cast<int>(m);
        st.src_H = H; st.src_W = W; st.src_C = C;
        st.Ho = Ho; st.Wo = Wo; st.Co = Co;
        st.mHW = tail_magic(Ho * Wo); st.mW = tail_magic(Wo);
        st.act = n.act;
        const bool src_inside = made.count(src) && made[src] < static_cast<int>(m);
        if (src_inside && !in_lds(src)) return false;
        // where the output goes
        bool used_inside = false;
        for (size_t m2 = m + 1; m2 < M.size(); m2++)
            used_inside |= std::find(M[m2].in.begin(), M[m2].in.end(), n.out) != M[m2].in.end() || M[m2].res == n.out;
        const bool ext = external(n.out);
        if (!used_inside && !ext) return false;
        if (n.kind == Node::Block && n.w >= 0) {
            if (n.KH != 3 || n.KW != 3 || n.sh != n.sw || (n.sh != 1 && n.sh != 2) || n.padding != Padding::Same) return false;
            st.kind = TAIL_DW;
            st.K = 3; st.S = n.sh; st.Kv = C;
            st.mHWp = tail_magic(Ho * ((Wo + 1) / 2)); st.mWp = tail_magic((Wo + 1) / 2);
            st.pt = std::max(0, (Ho - 1) * n.sh + 3 - H) / 2;
            st.pl = std::max(0, (Wo - 1) * n.sw + 3 - W) / 2;
            if (!tail_stage_ok(TAIL_DW, 3, st.S, C, C, Co, used_inside)) return false;
            if (!in_lds(src)) {
                // a stride-1 block brings its frame in (its skip is the frame, and it updates it in place); a stride-2 block reads the
                // nine taps of a pixel straight from global memory (its input is four times the frame and often does not fit)
                if (n.sh == 1) { if (!load(src)) return false; }
                else { sg.src_t = src; add_ext_in(src); }
            }
        } else {
            int K = 1;
            if (n.kind == Node::Conv) {
                if (n.KH != n.KW || n.res >= 0) return false;
                K = n.KH;
                if (K != 1 && (K != 2 || n.sh != 2 || n.sw != 2 || H % 2 || W % 2)) return false;
                if (K == 1 && (n.sh != 1 || n.sw != 1)) return false;
                if (Ho != H / K || Wo != W / K) return false;
            } else if (n.kind != Node::Block) {
                return false;
            }
            st.kind = TAIL_GATHER;
            st.K = K; st.S = K; st.Kv = K * K * C;
            if (!tail_stage_ok(TAIL_GATHER, K, K, C, st.Kv, Co, used_inside)) return false;
            if (!in_lds(src) && !load(src)) return false;
        }
        if (in_lds(src)) lt[src].last = static_cast<int>(S.size());
        if (n.res >= 0) {
            const auto& sr = shp(n.res);
            if (sr.size() != 4 || sr[3] % 4 || sr[3] > Co) return false;
            if (n.res_mode == RES_DIRECT) { if (sr[1] != Ho || sr[2] != Wo) return false; }
            else if (n.res_mode == RES_MAXPOOL) { if (sr[1] != 2 * Ho || sr[2] != 2 * Wo) return false; }
            else return false;
            st.res_mode = n.res_mode; st.res_C = sr[3]; st.res_W = sr[2];
            if (in_lds(n.res)) {
                lt[n.res].last = static_cast<int>(S.size());
            } else if (made.count(n.res) && made[n.res] < static_cast<int>(m)) {
                return false;
            } else {
                sg.res_t = n.res;
                add_ext_in(n.res);
            }
        }
        if (used_inside) {
            Lt& y = geom(n.out);
            y.in_lds = true; y.def = static_cast<int>(S.size());
        }
        if (ext) { sg.dst_t = n.out; ext_out.push_back(n.out); }
        min_px = std::min(min_px, Ho * Wo);
        S.push_back(sg);
    }
    if (ext_out.empty() || ext_in.size() + ext_out.size() > 12) return false;
    // ---- LDS placement in stage order (floats per frame; the kernel multiplies every offset by the frames per workgroup)
    struct Live { int off, size, t; };   // t < 0: a depthwise scratch
    std::vector<Live> live;
    int high = 0;
    auto size_of = [&](const Lt& x) { return x.H * x.W * (x.C + 4); };
    auto place = [&](int size, int t) {
        std::sort(live.begin(), live.end(), [](const Live& a, const Live& b) { return a.off < b.off; });
        int off = 0;
        for (const Live& l : live) {
            if (off + size <= l.off) break;
            off = std::max(off, l.off + l.size);
        }
        live.push_back({off, size, t});
        high = std::max(high, off + size);
        return off;
    };
    for (size_t k = 0; k < S.size(); k++) {
        Node::Stage& sg = S[k];
        TailStage& st = sg.tst;
        const int src = sg.member >= 0 ? M[static_cast<size_t>(sg.member)].in[0] : sg.src_t;
        const int dst = sg.member >= 0 ? M[static_cast<size_t>(sg.member)].out : sg.dst_t;
        const int res = sg.member >= 0 ? M[static_cast<size_t>(sg.member)].res : -1;
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return l.t < 0 || lt[l.t].last < static_cast<int>(k); }), live.end());
        if (st.kind != TAIL_LOAD && in_lds(src)) st.src_off = lt[src].off;
        if (st.kind != TAIL_LOAD && res >= 0 && in_lds(res)) st.res_off = lt[res].off;
        // (the output first: it outlives the stage's scratch, which then fills the space above it)
        if (in_lds(dst) && lt[dst].def == static_cast<int>(k)) {
            Lt& y = lt[dst];
            if (y.last < static_cast<int>(k)) y.last = static_cast<int>(k);
            bool placed = false;
            // in place on the skip tensor when it dies here: the lane that stores an output quad is the one that read the skip quad at
            // that address.  A depthwise stage may do so even when the skip is its own source (the taps went into the scratch before
            // the stage's barrier); a pointwise stage reads its source across lanes while others store.
            if (st.kind != TAIL_LOAD && res >= 0 && st.res_off >= 0 && st.res_mode == RES_DIRECT && (res != src || st.kind == TAIL_DW)) {
                const Lt& r = lt[res];
                if (r.last == static_cast<int>(k) && r.H == y.H && r.W == y.W && r.C == y.C) {
                    y.off = r.off;
                    for (Live& l : live) if (l.t == res) l.t = dst;
                    placed = true;
                }
            }
            if (!placed) y.off = place(size_of(y), dst);
            st.dst_off = y.off;
        }
        if (st.kind == TAIL_DW) {
            st.scr_off = place(st.Ho * st.Wo * (st.Kv + 4), -1);
            if (st.S == 2 && st.res_mode == RES_MAXPOOL && res == src && st.pt == 0 && st.pl == 0 && st.res_C == st.Kv) {
                const int before = high;
                st.pool_off = place(st.Ho * st.Wo * (st.Kv + 4), -1);
                if ((static_cast<long>(high) + 260) * 4 > 160 * 1024) { st.pool_off = -1; live.pop_back(); high = before; }   // no room: the epilogue pools from the source
            }
        }
    }
    const int frame_floats = (high + 3) & ~3;
    if ((static_cast<long>(frame_floats) + 256) * 4 > 160 * 1024) return false;
    Node r;
    r.kind = Node::Resident;
    r.tail = true;
    r.tail_frame_floats = frame_floats;
    r.tail_min_px = min_px;
    r.res_lds_bytes = (frame_floats + 256) * 4;
    r.members = std::move(M);
    r.stages = std::move(S);
    r.in = ext_in;
    r.out = ext_out.back();
    ext_out.pop_back();
    r.extra_out = ext_out;
    for (size_t k = i; k <= j; k++) r.src_ops.insert(r.src_ops.end(), ns[k].src_ops.begin(), ns[k].src_ops.end());
    *out = std::move(r);
    return true;
}

// Level 5, large frames: the bottleneck pair  r = act(PW(x));  y = act(PW(DW3x3(r)) + b + x')  (iris 32x32: 64 -> 32 -> 64
// channels) whose frame does not fit in LDS runs as ONE launch cut into row bands: a workgroup recomputes the one-row halo of
// r for its band, keeps r in LDS and reads x / x' and writes y in global memory — r never reaches HBM.
bool build_banded_bottleneck(const Graph& g, const std::vector<Node>& ns, size_t i, Node* out) {
    if (i + 1 >= ns.size()) return false;
    const Node &a = ns[i], &b = ns[i + 1];
    if (a.kind != Node::Block || a.w >= 0 || a.res >= 0 || a.in.size() != 1) return false;
    if (b.kind != Node::Block || b.w < 0 || b.in.size() != 1 || b.in[0] != a.out) return false;
    if (b.KH != 3 || b.KW != 3 || b.sh != 1 || b.sw != 1 || b.padding != Padding::Same) return false;
    if (b.res < 0 || b.res_mode != RES_DIRECT || b.res == b.in[0] || b.res_after) return false;
    if (std::find(g.outputs.begin(), g.outputs.end(), a.out) != g.outputs.end()) return false;
    for (size_t k = 0; k < ns.size(); k++) {
        if (k == i + 1) continue;
        if (std::find(ns[k].in.begin(), ns[k].in.end(), a.out) != ns[k].in.end() || ns[k].res == a.out) return false;  // r has one reader
    }
    const auto &sx = g.tensors[a.in[0]].shape, &sr = g.tensors[a.out].shape, &sy = g.tensors[b.out].shape, &ss = g.tensors[b.res].shape;
    if (sx.size() != 4 || sr.size() != 4 || sy.size() != 4 || ss.size() != 4) return false;
    const int H = sx[1], W = sx[2], Cx = sx[3], Cr = sr[3], Cy = sy[3];
    if (sr[1] != H || sr[2] != W || sy[1] != H || sy[2] != W || ss[1] != H || ss[2] != W) return false;
    if (Cx % 4 || Cr % 4 || Cy % 4 || ss[3] % 4 || ss[3] > Cy || H * W <= 256 || W > 64) return false;
    Node::Stage p, c;
    ResStage& ps = p.st;
    ps.kind = RES_STAGE_GATHER; ps.src_H = H; ps.src_W = W; ps.src_C = Cx; ps.KH = ps.KW = 1; ps.S = 1; ps.Kv = Cx;
    ps.Wo = W; ps.Co = Cr; ps.act = a.act; ps.band_role = 1; ps.band_H = H;
    p.member = 0; p.src_t = a.in[0];
    ResStage& cs = c.st;
    cs.kind = RES_STAGE_DW; cs.src_W = W; cs.src_C = Cr; cs.src_PS = Cr + 4; cs.src_b = 1; cs.src_off = 0;
    cs.KH = cs.KW = 3; cs.S = 1; cs.pt = cs.pl = 1; cs.Kv = Cr; cs.Wo = W; cs.Co = Cy; cs.act = b.act;
    cs.res_mode = RES_DIRECT; cs.res_C = ss[3]; cs.res_H = H; cs.res_W = W; cs.band_role = 2; cs.band_H = H;
    c.member = 1; c.dst_t = b.out; c.res_t = b.res;
    const int MT = (Cy + 31) / 32, Cp = (Cr + 7) & ~7;
    const int const_max = std::max(resident_const_floats(ps), resident_const_floats(cs));
    if (const_max > kResConstMax) return false;
    // rows per band: as many as keep two workgroups on a CU (half of the 160 KB each)
    const int slab = Cx % 16 == 0 ? kResSlabFloats : 0;  // the pointwise stage stages x through per-wave LDS slabs
    int R = 0, total = 0, dw_pg = 1;
    for (int r = std::min(H, 32); r >= 2; r--) {
        const int buf = (r + 2) * (W + 2) * (Cr + 4);
        const int PGn = (r * W + 31) / 32;
        const int pg = std::max(1, std::min(PGn, (8 + MT - 1) / MT));
        const int t = ((buf + 3) & ~3) + 2 * const_max + std::max(pg * 32 * (Cp + 4), slab);
        if (t * 4 <= 80 * 1024 - 512) { R = r; total = t; dw_pg = pg; break; }
    }
    if (R < 2) return false;
    R = (H + ((H + R - 1) / R) - 1) / ((H + R - 1) / R);  // same band count, even bands
    const int buf = (R + 2) * (W + 2) * (Cr + 4);
    {
        const int PGn = (R * W + 31) / 32;
        dw_pg = std::max(1, std::min(PGn, (8 + MT - 1) / MT));
        total = ((buf + 3) & ~3) + 2 * const_max + std::max(dw_pg * 32 * (Cp + 4), slab);
    }
    if (slab) { ps.kblk = 16; }
    ps.Ho = R + 2; ps.band_rows = R; ps.dst_off = 0; ps.dst_PS = Cr + 4; ps.dst_b = 1; ps.zero_dst = buf;
    cs.src_H = R; cs.Ho = R; cs.band_rows = R; cs.dw_pg = dw_pg;
    Node r;
    r.kind = Node::Resident;
    r.res_const_off = (buf + 3) & ~3;
    r.res_const_floats = const_max;
    cs.dw_off = r.res_const_off + 2 * const_max;
    ps.dw_off = cs.dw_off;
    r.res_lds_bytes = total * 4;
    r.res_bands = (H + R - 1) / R;
    r.members = {a, b};
    r.stages = {p, c};
    r.in = {a.in[0]};
    if (b.res != a.in[0]) r.in.push_back(b.res);
    r.out = b.out;
    r.src_ops = a.src_ops;
    r.src_ops.insert(r.src_ops.end(), b.src_ops.begin(), b.src_ops.end());
    *out = std::move(r);
    return true;
}

// Level 5: runs of bottleneck pairs  r = act(PW(x));  y = act(PW(DW3x3(r)) + b + x)  on 64 / 128 channels (the iris network) for the
// register-resident kernel (bneck_kernels.hip): frames of <= 256 pixels take the whole run in one launch, larger frames one pair per
// launch in row bands of 256 pixels.  Returns the number of plan nodes consumed (0: no match).
size_t build_bneck(const Graph& g, const std::vector<Node>& ns, size_t i, Node* out) {
    auto pair_at = [&](size_t k, int x_t) {
        if (k + 1 >= ns.size()) return false;
        const Node &a = ns[k], &b = ns[k + 1];
        if (a.kind != Node::Block || a.w >= 0 || a.res >= 0 || a.in.size() != 1 || a.in[0] != x_t) return false;
        if (b.kind != Node::Block || b.w < 0 || b.in.size() != 1 || b.in[0] != a.out) return false;
        if (b.KH != 3 || b.KW != 3 || b.sh != 1 || b.sw != 1 || b.padding != Padding::Same || b.ept >= 0 || b.epl >= 0) return false;
        if (b.res != x_t || b.res_mode != RES_DIRECT || b.res_after) return false;
        if (std::find(g.outputs.begin(), g.outputs.end(), a.out) != g.outputs.end()) return false;
        for (size_t q = 0; q < ns.size(); q++) {
            if (q == k + 1) continue;
            if (std::find(ns[q].in.begin(), ns[q].in.end(), a.out) != ns[q].in.end() || ns[q].res == a.out) return false;  // r has one reader
        }
        const auto &sx = g.tensors[x_t].shape, &sr = g.tensors[a.out].shape, &sy = g.tensors[b.out].shape;
        if (sx.size() != 4 || sr.size() != 4 || sy.size() != 4 || sy != sx || sr[1] != sx[1] || sr[2] != sx[2]) return false;
        return true;
    };
    if (i >= ns.size() || ns[i].in.size() != 1) return 0;
    const int x0 = ns[i].in[0];
    if (!pair_at(i, x0)) return 0;
    const auto& sx = g.tensors[x0].shape;
    const int H = sx[1], W = sx[2], C = sx[3], Cm = g.tensors[ns[i].out].shape[3];
    BneckArgs ba;
    ba.in = reinterpret_cast<const float*>(0x1000); ba.out = reinterpret_cast<float*>(0x2000);
    ba.in_fs = ba.out_fs = static_cast<long>(g.tensors[x0].elems());
    ba.B = 1; ba.H = H; ba.W = W; ba.C = C; ba.Cm = Cm;
    for (BneckBlock& bk : ba.blocks) { bk.w1 = bk.w2 = bk.consts = reinterpret_cast<const float*>(0x3000); }
    size_t npairs = 1;
    if (H * W <= 256) {
        if (H * W < 128) return 0;  // a quarter of the waves would have pixels: the stage programs do these
        // the whole run: the next pair reads this pair's output, which nobody else may need before the run ends ... unless it is written
        int x = ns[i + 1].out;
        while (npairs < static_cast<size_t>(kMaxBneck) && pair_at(i + 2 * npairs, x)) {
            bool other = std::find(g.outputs.begin(), g.outputs.end(), x) != g.outputs.end();
            for (size_t q = 0; q < ns.size() && !other; q++) {
                if (q == i + 2 * npairs || q == i + 2 * npairs + 1) continue;
                other = std::find(ns[q].in.begin(), ns[q].in.end(), x) != ns[q].in.end() || ns[q].res == x;
            }
            if (other) break;  // an intermediate frame that somebody else reads ends the run
            x = ns[i + 2 * npairs + 1].out;
            npairs++;
        }
        ba.bands = 1;
    } else {
        if (W > 256 || 256 / W < 2) return 0;
        const int R = 256 / W;
        ba.bands = (H + R - 1) / R;
    }
    ba.nblocks = static_cast<int>(npairs);
    if (!bneck_kernel_supports(ba)) return 0;
    Node r;
    r.kind = Node::Resident;
    r.bneck = true;
    r.res_bands = ba.bands;
    r.in = {x0};
    r.out = ns[i + 2 * npairs - 1].out;
    for (size_t k = i; k < i + 2 * npairs; k++) {
        r.members.push_back(ns[k]);
        r.src_ops.insert(r.src_ops.end(), ns[k].src_ops.begin(), ns[k].src_ops.end());
    }
    *out = std::move(r);
    return 2 * npairs;
}

// Level 5: full_range's double block  a = act(PW(DW3x3(x)));  y = act(PW(DW3x3(a)) + x)  (two stride-1 BlazeBlocks, the skip around
// both) as one launch of dblock_kernels.hip.  Returns the number of plan nodes consumed (0: no match).
size_t build_dblock(const Graph& g, const std::vector<Node>& ns, size_t i, Node* out) {
    if (i + 1 >= ns.size()) return 0;
    const Node &a = ns[i], &b = ns[i + 1];
    auto dw_block = [](const Node& n) {
        return n.kind == Node::Block && n.w >= 0 && n.KH == 3 && n.KW == 3 && n.sh == 1 && n.sw == 1 && n.padding == Padding::Same && n.ept < 0 && n.epl < 0 && n.in.size() == 1;
    };
    if (!dw_block(a) || !dw_block(b) || b.in[0] != a.out) return 0;
    // either the double block (no skip on the first half, the second half's skip is x) or two plain BlazeBlocks in a row (each adds its own input)
    const bool blaze_pair = a.res == a.in[0] && a.res_mode == RES_DIRECT && !a.res_after && b.res == a.out && b.res_mode == RES_DIRECT && !b.res_after;
    const bool dbl = a.res < 0 && b.res == a.in[0] && b.res_mode == RES_DIRECT && !b.res_after;
    // (the pair form is measured slower than two block-kernel launches — BackCamera 32x32x48: 0.107 against 2 x 0.040 ms, both stages are
    // heavy and the recomputed halo rows cost more than the saved round trip — and stays off unless asked for)
    static const bool pairs_on = getenv("MI_BLAZE_PAIRS") != nullptr;  // development aid
    if (!dbl && !(blaze_pair && pairs_on)) return 0;
    if (std::find(g.outputs.begin(), g.outputs.end(), a.out) != g.outputs.end()) return 0;
    for (size_t q = 0; q < ns.size(); q++) {
        if (q == i + 1) continue;
        if (std::find(ns[q].in.begin(), ns[q].in.end(), a.out) != ns[q].in.end() || ns[q].res == a.out) return 0;  // a has one reader
    }
    const auto &sx = g.tensors[a.in[0]].shape, &sa = g.tensors[a.out].shape, &sy = g.tensors[b.out].shape;
    if (sx.size() != 4 || sa.size() != 4 || sy.size() != 4 || sy[1] != sx[1] || sy[2] != sx[2] || sy[3] < sx[3] || sa[1] != sx[1] || sa[2] != sx[2]) return 0;
    if (sx[1] * sx[2] <= 256) return 0;  // small frames: the per-block launches with their tiles spread over the chip do better
    DblockArgs da;
    da.in = reinterpret_cast<const float*>(0x1000); da.out = reinterpret_cast<float*>(0x2000);
    da.in_fs = da.out_fs = static_cast<long>(g.tensors[a.in[0]].elems());
    da.B = 1; da.H = sx[1]; da.W = sx[2]; da.C = sx[3]; da.Cm = sa[3]; da.Co = sy[3];
    da.skip1 = blaze_pair; da.skip2_from_a = blaze_pair;
    da.out_fs = static_cast<long>(g.tensors[b.out].elems());
    da.consts = da.w1 = da.w2 = reinterpret_cast<const float*>(0x3000);
    if (!dblock_kernel_supports(da)) return 0;
    Node r;
    r.kind = Node::Resident;
    r.dblock = true;
    r.in = {a.in[0]};
    r.out = b.out;
    r.members = {a, b};
    r.src_ops = a.src_ops;
    r.src_ops.insert(r.src_ops.end(), b.src_ops.begin(), b.src_ops.end());
    *out = std::move(r);
    return 2;
}

// Level 5: runs of stride-1 BlazeBlocks on tiny frames (<= 64 pixels) whose channel count alternates narrow -> wide -> narrow ...
// (full_range's 6x6x96 -> 384 -> 96 pairs) as one frame-resident launch of xc_kernels.hip.  Returns the number of plan nodes consumed (0: no match).
size_t build_xc(const Graph& g, const std::vector<Node>& ns, size_t i, Node* out) {
    static const bool off = getenv("MI_NO_XC") != nullptr;  // development aid
    if (off || i + 1 >= ns.size()) return 0;
    auto plain = [](const Node& n) {
        return n.kind == Node::Block && n.w >= 0 && n.KH == 3 && n.KW == 3 && n.sh == 1 && n.sw == 1 && n.padding == Padding::Same && n.ept < 0 && n.epl < 0 &&
               n.in.size() == 1 && !n.res_after && n.act != ACT_PRELU;
    };
    // (a contract stage may be pointwise only: the last block of full_range's 6x6 run)
    auto pointwise = [](const Node& n) { return n.kind == Node::Block && n.w < 0 && n.sh == 1 && n.sw == 1 && n.in.size() == 1 && !n.res_after && n.act != ACT_PRELU; };
    if (!plain(ns[i])) return 0;
    const auto& s0 = g.tensors[ns[i].in[0]].shape;
    if (s0.size() != 4 || s0[1] * s0[2] > 64) return 0;
    // skip of an expand stage: 0 none, 1 its own input, 2 the max-pooled previous resolution, 3 the wide tensor of the pair before (full_range's
    // double block: wide -> narrow -> wide with the skip around both), -1 something else
    auto skip_of = [&](size_t n) {
        const Node& m = ns[i + n];
        if (m.res < 0) return 0;
        const auto &sr = g.tensors[m.res].shape, &so = g.tensors[m.out].shape;
        if (sr.size() != 4 || sr[3] % 4 || sr[3] > so[3]) return -1;
        if (m.res == m.in[0]) return m.res_mode == RES_DIRECT ? 1 : -1;
        if (n >= 2 && m.res == ns[i + n - 2].out) return (m.res_mode == RES_DIRECT && sr == so) ? 3 : -1;
        return (m.res_mode == RES_MAXPOOL && sr[1] == 2 * s0[1] && sr[2] == 2 * s0[2]) ? 2 : -1;
    };
    size_t n = 0;
    while (n < static_cast<size_t>(kMaxXc) && i + n < ns.size()) {
        const Node& m = ns[i + n];
        if (!plain(m) && !((n & 1) && pointwise(m))) break;
        if (n > 0 && m.in[0] != ns[i + n - 1].out) break;
        const auto &si = g.tensors[m.in[0]].shape, &so = g.tensors[m.out].shape;
        if (si.size() != 4 || so.size() != 4 || so[1] != s0[1] || so[2] != s0[2] || si[1] != s0[1] || si[2] != s0[2]) break;
        if (n & 1) {   // contract: back to the narrow width, no skip
            if (m.res >= 0 || so[3] != s0[3]) break;
        } else {       // expand: from the narrow width
            if (si[3] != s0[3] || so[3] <= si[3]) break;
            if (n >= 2 && so[3] != g.tensors[ns[i].out].shape[3]) break;
            if (skip_of(n) < 0) break;
        }
        n++;
    }
    // whole pairs whose intermediate tensors nobody outside the run reads (and that are no graph outputs)
    auto internal_ok = [&](size_t len) {
        for (size_t k = 0; k + 1 < len; k++) {
            const int t = ns[i + k].out;
            if (std::find(g.outputs.begin(), g.outputs.end(), t) != g.outputs.end()) return false;
            for (size_t q = 0; q < ns.size(); q++) {
                if (q >= i && q < i + len) continue;
                if (std::find(ns[q].in.begin(), ns[q].in.end(), t) != ns[q].in.end() || ns[q].res == t) return false;
            }
        }
        return true;
    };
    n &= ~static_cast<size_t>(1);
    while (n >= 2 && !internal_ok(n)) n -= 2;
    if (n < 2) return 0;
    XcArgs xa;
    xa.in = reinterpret_cast<const float*>(0x1000); xa.out = reinterpret_cast<float*>(0x2000);
    xa.in_fs = xa.out_fs = static_cast<long>(g.tensors[ns[i].in[0]].elems());
    xa.B = 1; xa.H = s0[1]; xa.W = s0[2]; xa.nstages = static_cast<int>(n);
    std::vector<int> extra_in;
    for (size_t k = 0; k < n; k++) {
        const Node& m = ns[i + k];
        XcStage& st = xa.st[k];
        st.w_pw = st.cblob = reinterpret_cast<const float*>(0x3000);
        st.has_dw = m.w >= 0;
        st.C = g.tensors[m.in[0]].shape[3]; st.Co = g.tensors[m.out].shape[3]; st.act = m.act;
        st.skip = (k & 1) ? 0 : skip_of(k);
        if (st.skip == 2) {
            st.res = reinterpret_cast<const float*>(0x4000); st.res_fs = static_cast<long>(g.tensors[m.res].elems());
            st.res_C = g.tensors[m.res].shape[3]; st.res_W = g.tensors[m.res].shape[2];
            if (std::find(extra_in.begin(), extra_in.end(), m.res) == extra_in.end()) extra_in.push_back(m.res);
        }
    }
    if (!xc_kernel_supports(xa)) return 0;
    Node r;
    r.kind = Node::Resident;
    r.xc = true;
    r.in = {ns[i].in[0]};
    for (int t : extra_in) r.in.push_back(t);
    r.out = ns[i + n - 1].out;
    for (size_t k = i; k < i + n; k++) {
        r.members.push_back(ns[k]);
        r.src_ops.insert(r.src_ops.end(), ns[k].src_ops.begin(), ns[k].src_ops.end());
    }
    *out = std::move(r);
    return n;
}

// Which nodes read tensor t (as input or skip).
std::vector<int> readers_of(const std::vector<Node>& ns, int t) {
    std::vector<int> r;
    for (size_t k = 0; k < ns.size(); k++)
        if (std::find(ns[k].in.begin(), ns[k].in.end(), t) != ns[k].in.end() || ns[k].res == t) r.push_back(static_cast<int>(k));
    return r;
}

// dep[b][a]: node b needs node a's result, directly or through other nodes (plan order = a topological order)
std::vector<std::vector<char>> dependence(const std::vector<Node>& ns) {
    const size_t N = ns.size();
    if (N > 2048) return {};   // (an N x N table: graphs of that size — none of the reference's has 200 operators — do without the fork analysis)
    std::map<int, int> producer;
    for (size_t i = 0; i < N; i++) {
        producer[ns[i].out] = static_cast<int>(i);
        for (int t : ns[i].extra_out) producer[t] = static_cast<int>(i);
    }
    std::vector<std::vector<char>> dep(N, std::vector<char>(N, 0));
    for (size_t b = 0; b < N; b++) {
        std::vector<int> srcs = ns[b].in;
        if (ns[b].res >= 0) srcs.push_back(ns[b].res);
        for (int t : srcs) {
            auto it = producer.find(t);
            if (it == producer.end() || it->second >= static_cast<int>(b)) continue;
            const size_t a = static_cast<size_t>(it->second);
            dep[b][a] = 1;
            for (size_t k = 0; k < N; k++) dep[b][k] |= dep[a][k];
        }
    }
    return dep;
}

// A fork: node j's output is read by two nodes neither of which needs the other — the start of independent branches (the two heads of
// the iris / face mesh networks).  A stage program that ran on past the fork would chain one branch behind the common part and leave
// the other waiting for the whole launch; ending the launch at the fork lets the engine run the branches side by side.
bool is_fork(const std::vector<Node>& ns, const std::vector<std::vector<char>>& dep, size_t j) {
    if (dep.size() != ns.size()) return false;
    std::vector<int> rd = readers_of(ns, ns[j].out);
    for (int t : ns[j].extra_out)   // a launch with several outputs: their readers part just the same
        for (int k : readers_of(ns, t))
            if (std::find(rd.begin(), rd.end(), k) == rd.end()) rd.push_back(k);
    for (size_t x = 0; x < rd.size(); x++)
        for (size_t y = x + 1; y < rd.size(); y++)
            if (!dep[static_cast<size_t>(rd[y])][static_cast<size_t>(rd[x])] && !dep[static_cast<size_t>(rd[x])][static_cast<size_t>(rd[y])]) return true;
    return false;
}

std::vector<Node> group_resident(const Graph& g, const std::vector<Node>& ns, int budget, bool tail) {
    std::vector<Node> outv;
    static const bool no_bneck = getenv("MI_NO_BNECK") != nullptr;  // development aid
    static const bool no_dblock = getenv("MI_NO_DBLOCK") != nullptr;  // development aid
    static const bool no_fork_split = getenv("MI_NO_FORK_SPLIT") != nullptr;  // development aid
    const bool no_tail = !tail || getenv("MI_NO_TAIL") != nullptr;     // option "tail" 0 / development aid: the round-4 plan (resident_kernel / chain_kernel for every small-spatial part)
    static const bool tail_split = getenv("MI_TAIL_SPLIT") != nullptr;  // development aid
    const std::vector<std::vector<char>> dep = dependence(ns);
    for (size_t i = 0; i < ns.size();) {
        {
            Node xn;
            const size_t used = build_xc(g, ns, i, &xn);
            if (used) {
                outv.push_back(std::move(xn));
                i += used;
                continue;
            }
        }
        {
            Node db;
            const size_t used = no_dblock ? 0 : build_dblock(g, ns, i, &db);
            if (used) {
                outv.push_back(std::move(db));
                i += used;
                continue;
            }
        }
        {
            Node bn;
            const size_t used = no_bneck ? 0 : build_bneck(g, ns, i, &bn);
            if (used) {
                outv.push_back(std::move(bn));
                i += used;
                continue;
            }
        }
        Node best;
        size_t best_j = 0;
        bool have = false;
        // the longest run from i in either form; the several-frames-per-workgroup form (tail_kernels.hip) wins when it reaches at
        // least as far (it also takes frame-resident chains with stride-2 neighbours, which the classic form leaves alone)
        for (int form = no_tail ? 1 : 0; form < 2; form++) {
            Node fbest;
            size_t fj = 0;
            bool fhave = false;
            for (size_t j = i; j < ns.size(); j++) {
                Node cand;
                {   // a stage program does not run on into an expand / contract run: that one has its own launch (xc_kernels.hip)
                    Node xn;
                    if (j > i && build_xc(g, ns, j, &xn)) break;
                }
                if (!(form == 0 ? build_tail(g, ns, i, j, &cand) : build_resident(g, ns, i, j, budget, &cand))) break;  // a longer run only needs more
                fbest = std::move(cand);
                fj = j;
                fhave = true;
                if (!no_fork_split && j > i && is_fork(ns, dep, j)) break;  // the launch ends where the branches part (a fork at the very start is the previous launch's business)
                if (form == 0 && tail_split && j + 1 < ns.size() && ns[j].kind == Node::Chain && ns[j].chain_post) break;  // development aid: a launch per resolution
            }
            if (fhave && (!have || fj > best_j)) { best = std::move(fbest); best_j = fj; have = true; }
        }
        // worth a launch of its own: at least two fused nodes, or a k x k convolution (otherwise the generic direct conv)
        bool take = have && (best.members.size() >= 2 || (best.members[0].kind == Node::Conv && best.members[0].KH > 1));
        Node banded;
        if (take) {
            outv.push_back(std::move(best));
            i = best_j + 1;
        } else if (build_banded_bottleneck(g, ns, i, &banded)) {
            outv.push_back(std::move(banded));
            i += 2;
        } else {
            outv.push_back(ns[i]);
            i++;
        }
    }
    return outv;
}

}  // namespace

namespace {
Plan build_plan_impl(Graph graph, int fuse_level, int pipe_max_opt, int res_budget_bytes, bool fuse_heads, bool tail);

// The chain's output heads and the batch-GEMM heads are chosen before storage exists (placeholder pointers, 16-byte aligned); the
// kernels need the real views 16-byte aligned with frame strides that are multiples of 4 floats.  A head whose slice of a
// concatenated output starts at an odd offset (e.g. 15 x 15 anchors) fails that: such a graph is lowered again without them.
bool head_views_aligned(const Plan& plan) {
    auto aligned = [&](int t) {
        if (t < 0) return true;
        const Storage& s = plan.storage[t];
        return (s.offset & 3) == 0 && (s.frame_stride & 3) == 0;
    };
    for (const Node& n : plan.nodes) {
        // chain heads: the head with Co % 4 == 0 of a pair is stored as float4 (chain_kernel_supports: out_a, out_a_fs); its partner
        // (the classifier, 2 / 6 channels) is stored float by float and may start anywhere
        if (n.kind == Node::Chain)
            for (const Node::HeadPair& hp : n.head_pairs)
                if (!aligned(n.head_nodes[static_cast<size_t>(hp.a)].out)) return false;
        // batch GEMM heads read their input as float4 (launch_head_gemm: in, in_fs); the output is stored float by float (the face
        // flag is one float per frame)
        if (n.kind == Node::Conv && n.gemm_head && !aligned(n.in[0])) return false;
    }
    return true;
}
}  // namespace

Plan build_plan(Graph graph, int fuse_level, int pipe_max_opt, int res_budget_bytes, bool tail) {
    Plan plan = build_plan_impl(graph, fuse_level, pipe_max_opt, res_budget_bytes, true, tail);
    if (head_views_aligned(plan)) return plan;
    return build_plan_impl(std::move(graph), fuse_level, pipe_max_opt, res_budget_bytes, false, tail);
}

namespace {
// Activation tensors whose channel count is not a multiple of 4 (front / short-range: 36 -> 42 -> 48) keep their layers off every
// fused kernel (float4 pixels everywhere).  Such a tensor — together with everything tied to it by channel-preserving operators
// (depthwise conv, ADD, RELU / PRELU, MAX_POOL) — is widened to the next multiple of 4 with channels that are exactly zero: the
// producing convolution gets zero filter rows and biases, depthwise stages zero taps and biases, PRELU a zero slope, consuming
// convolutions zero filter columns, and channel PADs on either side (zeros appended at the high end) append
// or remove correspondingly fewer / more.  Every real channel computes what it computed before, bit for bit.  Classes that touch a
// graph input / output or an operator not listed here are left alone.
void pad_odd_channels(Graph& g, std::vector<double>* logical_elems, std::vector<int>* logical_C) {
    const int NT = static_cast<int>(g.tensors.size());
    logical_elems->resize(NT);
    logical_C->resize(NT);
    for (int t = 0; t < NT; t++) {
        (*logical_elems)[t] = static_cast<double>(g.tensors[t].elems());
        (*logical_C)[t] = g.tensors[t].shape.empty() ? 0 : g.tensors[t].shape.back();
    }
    // Untrusted graphs reach this pass before the lowering has checked them: anything that is not the plain, consistent form
    // handled below leaves the graph as it is (and to the lowering's own error messages).
    for (const OpInfo& op : g.ops) {
        if (op.inputs.empty() || op.outputs.empty()) return;
        auto cst = [&](int t, size_t rank) {
            return t >= 0 && g.tensors[t].is_const && g.tensors[t].shape.size() == rank && g.tensors[t].f32.size() == g.tensors[t].elems() && g.tensors[t].elems() > 0;
        };
        auto r4 = [&](int t) { return t >= 0 && !g.tensors[t].is_const && g.tensors[t].shape.size() == 4 && g.tensors[t].shape.back() > 0; };
        const int in = op.inputs[0], out = op.outputs[0];
        if (op.op == BuiltinOp::Conv2D || op.op == BuiltinOp::DepthwiseConv2D) {
            if (op.inputs.size() < 2 || !r4(in) || !r4(out) || !cst(op.inputs[1], 4)) return;
            const auto& ws = g.tensors[op.inputs[1]].shape;
            const int Ci = g.tensors[in].shape.back(), Co = g.tensors[out].shape.back();
            if (op.op == BuiltinOp::Conv2D ? (ws[0] != Co || ws[3] != Ci) : (ws[0] != 1 || ws[3] != Ci || Ci != Co)) return;
            if (op.inputs.size() > 2 && op.inputs[2] >= 0 && !(cst(op.inputs[2], 1) && g.tensors[op.inputs[2]].shape[0] == Co)) return;
        } else if (op.op == BuiltinOp::Prelu) {
            if (op.inputs.size() < 2 || !r4(in) || !r4(out)) return;
            const TensorInfo& al = g.tensors[op.inputs[1]];
            if (!al.is_const || al.shape.empty() || al.f32.size() != al.elems() || static_cast<int>(al.elems()) != g.tensors[in].shape.back() ||
                al.shape.back() != g.tensors[in].shape.back()) return;
        } else if (op.op == BuiltinOp::Pad) {
            if (op.inputs.size() < 2 || op.inputs[1] < 0 || !g.tensors[op.inputs[1]].is_const) return;
        } else if (op.op == BuiltinOp::Add) {
            if (op.inputs.size() < 2 || op.inputs[1] < 0) return;
        }
    }
    for (int t = 0; t < NT; t++)
        if (!g.tensors[t].is_const && g.tensors[t].shape.empty()) return;
    std::vector<int> parent(NT);
    for (int t = 0; t < NT; t++) parent[t] = t;
    std::function<int(int)> find = [&](int t) { return parent[t] == t ? t : parent[t] = find(parent[t]); };
    auto unite = [&](int a, int b) { if (a >= 0 && b >= 0) parent[find(a)] = find(b); };
    auto act = [&](int t) { return t >= 0 && t < NT && !g.tensors[t].is_const; };
    for (const OpInfo& op : g.ops) {
        switch (op.op) {
            case BuiltinOp::DepthwiseConv2D: case BuiltinOp::Relu: case BuiltinOp::Prelu: case BuiltinOp::MaxPool2D:
                unite(op.inputs.at(0), op.outputs.at(0));
                break;
            case BuiltinOp::Add:
                unite(op.inputs.at(0), op.outputs.at(0));
                if (act(op.inputs.at(1))) unite(op.inputs.at(1), op.outputs.at(0));
                break;
            default: break;
        }
    }
    std::vector<char> bad(NT, 0);  // per class root
    auto mark = [&](int t) { if (act(t)) bad[find(t)] = 1; };
    for (int t : g.inputs) mark(t);
    for (int t : g.outputs) mark(t);
    for (int t = 0; t < NT; t++)
        if (act(t) && g.tensors[t].shape.size() != 4) mark(t);
    for (const OpInfo& op : g.ops) {
        const bool conv = op.op == BuiltinOp::Conv2D;
        switch (op.op) {
            case BuiltinOp::Conv2D: case BuiltinOp::DepthwiseConv2D: case BuiltinOp::Relu: case BuiltinOp::MaxPool2D:
                break;
            case BuiltinOp::Prelu:
                if (!g.tensors[op.inputs.at(1)].is_const) mark(op.inputs[0]);
                break;
            case BuiltinOp::Add:
                if (!act(op.inputs.at(1))) mark(op.inputs[0]);  // constant addend: not handled
                if (!act(op.inputs.at(0))) { mark(op.inputs[1]); mark(op.outputs.at(0)); }  // ... on either side
                break;
            case BuiltinOp::Pad: {
                const auto& pd = g.tensors[op.inputs.at(1)].i32;
                bool channel_only = pd.size() == 8;
                for (size_t k = 0; k < 7 && channel_only; k++) channel_only = pd[k] == 0;
                if (!channel_only) { mark(op.inputs[0]); mark(op.outputs[0]); }
                break;
            }
            default:  // RESHAPE, CONCATENATION, RESIZE_BILINEAR, DEPTH_TO_SPACE ...: the channel count is part of their meaning
                for (int t : op.inputs) mark(t);
                for (int t : op.outputs) mark(t);
                break;
        }
        (void)conv;
    }
    // Growth groups: classes tied by the channel PAD on the skip of a stride-1 block (the front / short-range graph widens 32 -> 36 -> 42
    // and 48 -> 56 -> ... -> 88 that way, a few channels per block).  Widened to ONE width, such a run becomes a run of same-shape
    // blocks with plain skips — the shape the fast kernels exist for (BackCamera's: 48 channels on 32-pixel-wide frames, 96 on
    // frames of at most 256 pixels) — and its PADs disappear.  The price is arithmetic on zeros; the gain is kernels and launches
    // (front: five 16x16 blocks of 48..88 channels -> one frame-resident chain).  Only PADs that read a block's own input count: the
    // stride-2 blocks' PADs (behind the max-pool) separate the groups.
    std::vector<int> gparent(NT);
    for (int t = 0; t < NT; t++) gparent[t] = t;
    std::function<int(int)> gfind = [&](int t) { return gparent[t] == t ? t : gparent[t] = gfind(gparent[t]); };
    std::vector<char> from_pool(NT, 0);
    for (const OpInfo& op : g.ops)
        if (op.op == BuiltinOp::MaxPool2D) from_pool[op.outputs[0]] = 1;
    static const bool no_groups = getenv("MI_NO_WIDEN_GROUPS") != nullptr;  // tuning aid
    struct GroupInfo { int maxC = 0, H = 0, W = 0, classes = 0; bool ok = true; };
    std::map<int, GroupInfo> groups;  // by group root (a class root)
    if (!no_groups) {
        for (const OpInfo& op : g.ops)
            if (op.op == BuiltinOp::Pad) {
                const int in = op.inputs.at(0), out = op.outputs.at(0);
                if (!act(in) || from_pool[in] || bad[find(in)] || bad[find(out)]) continue;
                gparent[gfind(find(in))] = gfind(find(out));
            }
        std::map<int, std::set<int>> members;
        for (const OpInfo& op : g.ops)
            if (op.op == BuiltinOp::Pad) {
                const int in = op.inputs.at(0), out = op.outputs.at(0);
                if (!act(in) || from_pool[in] || bad[find(in)] || bad[find(out)]) continue;
                GroupInfo& gi = groups[gfind(find(out))];
                const auto& sh = g.tensors[out].shape;
                if (gi.H == 0) { gi.H = sh[1]; gi.W = sh[2]; }
                gi.ok &= gi.H == sh[1] && gi.W == sh[2];
                gi.maxC = std::max({gi.maxC, g.tensors[in].shape.back(), sh.back()});
                members[gfind(find(out))].insert(find(in));
                members[gfind(find(out))].insert(find(out));
            }
        for (auto& kv : groups) kv.second.classes = static_cast<int>(members[kv.first].size());
    }
    auto group_width = [&](int t) {  // 0: the tensor's class is in no group that gets one width
        auto it = groups.find(gfind(find(t)));
        if (it == groups.end() || !it->second.ok || it->second.classes < 2) return 0;
        const GroupInfo& gi = it->second;
        if (gi.H * gi.W <= 256 && ((gi.maxC + 31) & ~31) <= 128) return (gi.maxC + 31) & ~31;  // frame-resident chains: whole 32-channel tiles
        if (gi.W == 32 && gi.H * gi.W <= 1024 && gi.maxC > 32 && gi.maxC <= 48) return 48;     // mstrip_kernel
        return 0;
    };
    auto padded = [&](int t) {  // new channel count of activation tensor t (unchanged when its class is left alone)
        const int C = g.tensors[t].shape.back();
        if (!(act(t) && g.tensors[t].shape.size() == 4 && !bad[find(t)])) return C;
        const int gw = group_width(t);
        if (gw >= C) return gw;
        return (C & 3) ? (C + 3) & ~3 : C;
    };
    // a channel PAD that consumes a widened tensor must have room for the extra channels.  Marking a class changes what its
    // neighbours (and its growth group) are padded to, so the check runs to a fixpoint (a single pass could leave an
    // earlier PAD with a negative amount on crafted graphs)
    // (a pass counts as a change only when a class really flips — a PAD whose data input is a constant can never be marked
    // (mark() skips constants), and with a shrinking output it kept the loop spinning on crafted graphs; every flip is permanent and
    // there are at most NT classes, so the loop ends)
    for (bool changed = true; changed;) {
        changed = false;
        for (const OpInfo& op : g.ops)
            if (op.op == BuiltinOp::Pad) {
                const int in = op.inputs.at(0), out = op.outputs.at(0);
                if (padded(out) - padded(in) >= 0) continue;
                auto flip = [&](int t) {
                    if (!act(t) || bad[find(t)]) return false;
                    bad[find(t)] = 1;
                    return true;
                };
                const bool a = flip(in), b = flip(out);
                changed |= a || b;
            }
    }
    std::vector<int> newC(NT);
    bool any = false;
    for (int t = 0; t < NT; t++) {
        newC[t] = act(t) ? padded(t) : 0;
        any |= act(t) && newC[t] != g.tensors[t].shape.back();
    }
    if (!any) return;
    // last line of defence before anything is rewritten: the new widths must be consistent operator by operator, or the graph stays
    // exactly as it was stored
    for (const OpInfo& op : g.ops) {
        const int in = op.inputs.at(0), out = op.outputs.at(0);
        switch (op.op) {
            case BuiltinOp::Pad:
                if (act(in) && newC[out] - newC[in] < 0) return;
                break;
            case BuiltinOp::DepthwiseConv2D: case BuiltinOp::Relu: case BuiltinOp::Prelu: case BuiltinOp::MaxPool2D:
                if (act(in) && newC[in] != newC[out]) return;
                break;
            case BuiltinOp::Add:
                if ((act(in) && newC[in] != newC[out]) || (act(op.inputs.at(1)) && newC[op.inputs[1]] != newC[out])) return;
                break;
            default: break;
        }
    }
    auto clone_const = [&](int t) {  // constants may be shared between operators: every change goes to a private copy
        g.tensors.push_back(g.tensors[t]);
        logical_elems->push_back((*logical_elems)[t]);
        logical_C->push_back((*logical_C)[t]);
        return static_cast<int>(g.tensors.size()) - 1;
    };
    auto widen_last = [&](int t, int Cn) {  // [..., C] -> [..., Cn], zeros appended (filters [*,kh,kw,I], taps [1,kh,kw,C], bias / alpha [C])
        TensorInfo& ti = g.tensors[t];
        const int C = ti.shape.back();
        const size_t rows = ti.elems() / static_cast<size_t>(C);
        std::vector<float> w(rows * Cn, 0.f);
        for (size_t r = 0; r < rows; r++)
            for (int c = 0; c < C; c++) w[r * Cn + c] = ti.f32[r * C + c];
        ti.f32 = std::move(w);
        ti.shape.back() = Cn;
    };
    auto widen_first = [&](int t, int On) {  // [O, ...] -> [On, ...], zero filters appended
        TensorInfo& ti = g.tensors[t];
        const size_t per = ti.elems() / static_cast<size_t>(ti.shape[0]);
        ti.f32.resize(per * On, 0.f);
        ti.shape[0] = On;
    };
    for (OpInfo& op : g.ops) {
        const int in = op.inputs.at(0), out = op.outputs.at(0);
        const int Ci = act(in) ? g.tensors[in].shape.back() : 0, Co = g.tensors[out].shape.back();
        const int Cin = act(in) ? newC[in] : 0, Con = newC[out];
        if (op.op == BuiltinOp::Conv2D) {
            if (Cin != Ci || Con != Co) {
                op.inputs[1] = clone_const(op.inputs[1]);
                if (Cin != Ci) widen_last(op.inputs[1], Cin);
                if (Con != Co) widen_first(op.inputs[1], Con);
            }
            if (Con != Co && op.inputs.size() > 2 && op.inputs[2] >= 0) { op.inputs[2] = clone_const(op.inputs[2]); widen_last(op.inputs[2], Con); }
        } else if (op.op == BuiltinOp::DepthwiseConv2D && Con != Co) {
            op.inputs[1] = clone_const(op.inputs[1]);
            widen_last(op.inputs[1], Con);
            if (op.inputs.size() > 2 && op.inputs[2] >= 0) { op.inputs[2] = clone_const(op.inputs[2]); widen_last(op.inputs[2], Con); }
        } else if (op.op == BuiltinOp::Prelu && Con != Co) {
            op.inputs[1] = clone_const(op.inputs[1]);
            widen_last(op.inputs[1], Con);
        } else if (op.op == BuiltinOp::Pad && (Cin != Ci || Con != Co)) {
            op.inputs[1] = clone_const(op.inputs[1]);
            g.tensors[op.inputs[1]].i32[7] = Con - Cin;
        }
    }
    for (int t = 0; t < NT; t++)
        if (act(t)) g.tensors[t].shape.back() = newC[t];
    // channel PADs that no longer append anything: their output is their input
    std::vector<OpInfo> kept;
    std::map<int, int> alias;
    auto resolve = [&](int t) { for (auto it = alias.find(t); it != alias.end(); it = alias.find(t)) t = it->second; return t; };
    for (OpInfo& op : g.ops) {
        for (int& t : op.inputs) t = t >= 0 ? resolve(t) : t;
        if (op.op == BuiltinOp::Pad && act(op.inputs[0]) && g.tensors[op.inputs[1]].i32.size() == 8 && g.tensors[op.inputs[1]].i32[7] == 0 &&
            g.tensors[op.inputs[0]].shape == g.tensors[op.outputs[0]].shape) {
            alias[op.outputs[0]] = op.inputs[0];
            continue;
        }
        kept.push_back(std::move(op));
    }
    g.ops = std::move(kept);
    for (int& t : g.outputs) t = resolve(t);
}

Plan build_plan_impl(Graph graph, int fuse_level, int pipe_max_opt, int res_budget_bytes, bool fuse_heads, bool tail) {
    Plan plan;
    plan.graph = std::move(graph);
    plan.fuse_level = fuse_level;
    // (untrusted bytes, mi_*_create_from_bytes: the reader has checked every tensor index; the lowering below reads shape.back() of the tensors an
    // operator touches — a crafted tensor of rank 0 made that a read in front of an empty vector, found under AddressSanitizer)
    for (const OpInfo& op : plan.graph.ops) {
        if (op.outputs.empty()) throw std::runtime_error("plan: operator without an output");
        for (int t : op.outputs)
            if (t < 0 || plan.graph.tensors[static_cast<size_t>(t)].shape.empty()) throw std::runtime_error("plan: operator output without a shape");
        for (int t : op.inputs)
            if (t >= 0 && plan.graph.tensors[static_cast<size_t>(t)].shape.empty()) throw std::runtime_error("plan: operator input without a shape");
    }
    for (int t : plan.graph.inputs)
        if (t < 0 || plan.graph.tensors[static_cast<size_t>(t)].shape.empty()) throw std::runtime_error("plan: graph input without a shape");
    for (int t : plan.graph.outputs)
        if (t < 0 || plan.graph.tensors[static_cast<size_t>(t)].shape.empty()) throw std::runtime_error("plan: graph output without a shape");
    // algorithmic sizes (traffic / MAC figures of the plan) are those of the graph as stored, whatever padding the lowering adds
    std::vector<double> logical_elems;
    std::vector<int> logical_C;
    if (fuse_level >= 2) pad_odd_channels(plan.graph, &logical_elems, &logical_C);  // (the op-by-op levels take any channel count)
    Graph& g = plan.graph;
    std::vector<Node> nodes;
    for (size_t oi = 0; oi < g.ops.size(); oi++) {
        const OpInfo& op = g.ops[oi];
        Node n;
        n.out = op.outputs[0];
        n.src_ops = {static_cast<int>(oi)};
        auto need_const = [&](int t, const char* what) {
            if (t < 0 || !g.tensors[t].is_const) throw std::runtime_error(std::string("plan: ") + what + " must be a constant tensor");
        };
        switch (op.op) {
            case BuiltinOp::Conv2D:
            case BuiltinOp::DepthwiseConv2D: {
                n.kind = op.op == BuiltinOp::Conv2D ? Node::Conv : Node::Dw;
                n.in = {op.inputs.at(0)};
                n.w = op.inputs.at(1);
                need_const(n.w, "convolution filter");
                n.b = op.inputs.size() > 2 ? op.inputs[2] : -1;
                if (n.b >= 0) need_const(n.b, "convolution bias");
                const auto& ws = g.tensors[n.w].shape;
                if (ws.size() != 4) throw std::runtime_error("plan: convolution filter must be rank 4");
                n.KH = ws[1]; n.KW = ws[2];
                n.sh = op.stride_h; n.sw = op.stride_w;
                n.padding = op.padding;
                n.act = act_of(op.act);
                if (n.kind == Node::Dw && (op.depth_multiplier != 1 || ws[0] != 1 || ws[3] != g.tensors[n.in[0]].shape.at(3)))
                    throw std::runtime_error("plan: depthwise conv with depth_multiplier != 1 unsupported");
                if (n.kind == Node::Conv && ws[3] != g.tensors[n.in[0]].shape.at(3))
                    throw std::runtime_error("plan: convolution channel mismatch");
                break;
            }
            case BuiltinOp::Add:
                n.kind = Node::Add;
                n.in = {op.inputs.at(0), op.inputs.at(1)};
                n.act = act_of(op.act);
                if (g.tensors[n.in[0]].shape != g.tensors[n.in[1]].shape) throw std::runtime_error("plan: broadcasting ADD unsupported");
                break;
            case BuiltinOp::Relu:
                n.kind = Node::Act; n.in = {op.inputs.at(0)}; n.act = ACT_RELU;
                break;
            case BuiltinOp::Prelu:
                n.kind = Node::Act; n.in = {op.inputs.at(0)}; n.act = ACT_PRELU; n.alpha = op.inputs.at(1);
                need_const(n.alpha, "PRELU alpha");
                if (static_cast<int>(g.tensors[n.alpha].elems()) != g.tensors[n.in[0]].shape.back())
                    throw std::runtime_error("plan: PRELU alpha must be per-channel");
                break;
            case BuiltinOp::MaxPool2D:
                n.kind = Node::MaxPool; n.in = {op.inputs.at(0)};
                n.filter_h = op.filter_h; n.filter_w = op.filter_w; n.sh = op.stride_h; n.sw = op.stride_w;
                n.padding = op.padding; n.act = act_of(op.act);
                break;
            case BuiltinOp::Pad:
                n.kind = Node::Pad; n.in = {op.inputs.at(0)}; n.pads = op.inputs.at(1);
                need_const(n.pads, "PAD paddings");
                if (g.tensors[n.pads].i32.size() != 8) throw std::runtime_error("plan: PAD expects rank-4 paddings");
                break;
            case BuiltinOp::Reshape:
                n.kind = Node::Reshape; n.in = {op.inputs.at(0)};
                break;
            case BuiltinOp::Concatenation:
                n.kind = Node::Concat; n.in = op.inputs; n.axis = op.axis;
                if (op.act != FusedAct::None) throw std::runtime_error("plan: CONCATENATION with activation unsupported");
                break;
            case BuiltinOp::ResizeBilinear:
                n.kind = Node::Resize; n.in = {op.inputs.at(0)};
                n.half_pixel = op.half_pixel_centers; n.align_corners = op.align_corners;
                break;
            case BuiltinOp::DepthToSpace:
                n.kind = Node::DepthToSpace; n.in = {op.inputs.at(0)}; n.block_size = op.block_size;
                break;
            default:
                throw std::runtime_error("plan: builtin operator " + std::to_string(op.raw_code) + " unsupported");
        }
        for (int t : n.in)
            if (t < 0 || g.tensors[t].is_const) throw std::runtime_error("plan: constant activation inputs unsupported");
        nodes.push_back(std::move(n));
    }

    Rewriter rw{g, nodes};
    if (fuse_level >= 1) rw.fold_spatial_pads();
    if (fuse_level >= 1) rw.fuse_epilogues();
    if (fuse_level >= 2) rw.fuse_blocks();
    for (auto& n : nodes)
        if (!n.dead) plan.nodes.push_back(n);

    // ---- level 3: frame-resident chains (consecutive same-shape stride-1 blocks, each feeding only the next one)
    if (fuse_level >= 3) {
        auto uses = [&](int t) {
            int c = 0;
            for (const Node& n : plan.nodes) {
                for (int x : n.in) c += x == t;
                c += n.res == t;
            }
            return c;
        };
        // a block that can sit inside a chain: stride 1, same shape in and out, skip = its own input (or none)
        auto chainable = [&](const Node& n) {
            if (n.kind != Node::Block || n.w < 0 || n.sh != 1 || n.sw != 1 || n.padding != Padding::Same) return false;
            const auto& si = g.tensors[n.in[0]].shape;
            const auto& so = g.tensors[n.out].shape;
            if (si.size() != 4 || si != so) return false;
            if (n.res >= 0 && !(n.res == n.in[0] && n.res_mode == RES_DIRECT && !n.res_after)) return false;
            return true;
        };
        auto links = [&](size_t j) {  // node j+1 continues the chain that node j is in
            return j + 1 < plan.nodes.size() && chainable(plan.nodes[j + 1]) && plan.nodes[j + 1].in[0] == plan.nodes[j].out &&
                   g.tensors[plan.nodes[j + 1].in[0]].shape == g.tensors[plan.nodes[j].in[0]].shape &&
                   (plan.nodes[j + 1].act == ACT_RELU) == (plan.nodes[j].act == ACT_RELU) &&
                   uses(plan.nodes[j].out) == (plan.nodes[j + 1].res >= 0 ? 2 : 1) &&
                   std::find(g.outputs.begin(), g.outputs.end(), plan.nodes[j].out) == g.outputs.end();
        };
        auto make_chain = [&](size_t i, size_t j) {
            Node c;
            c.kind = Node::Chain;
            c.in = {plan.nodes[i].in[0]};
            c.out = plan.nodes[j].out;
            for (size_t k = i; k <= j; k++) {
                c.members.push_back(plan.nodes[k]);
                c.src_ops.insert(c.src_ops.end(), plan.nodes[k].src_ops.begin(), plan.nodes[k].src_ops.end());
            }
            return c;
        };
        // The stride-2 BlazeBlocks on either side of a frame-resident chain join its launch (chain_kernels.hip, ChainEdge): the
        // one in front (DW3x3 s2 -> PW -> + 2x2 max-pool skip -> act; the producer of the chain's input, read by nobody else) gathers
        // its taps from global memory and leaves its output in the LDS tile; the one behind reads the tile and writes global memory.
        // `next` = index in plan.nodes of the node after the chain (advanced past an absorbed block).
        auto edge_block = [&](const Node& e) {
            return e.kind == Node::Block && !e.res_after && e.w >= 0 && e.KH == 3 && e.KW == 3 && e.sh == 2 && e.sw == 2 && e.padding == Padding::Same &&
                   (e.res < 0 || (e.res == e.in[0] && e.res_mode == RES_MAXPOOL));
        };
        auto is_output = [&](int t) { return std::find(g.outputs.begin(), g.outputs.end(), t) != g.outputs.end(); };
        std::vector<char> absorbed(plan.nodes.size(), 0);
        // 1x1 convolutions without activation or skip whose result only reaches graph outputs through views (RESHAPE /
        // CONCATENATION): the SSD heads.  Those that read tensor t, at plan positions >= from.
        auto view_only = [&](int t) {
            for (int depth = 0; depth < 4; depth++) {
                if (is_output(t)) return true;
                int next_t = -1, n_cons = 0;
                for (const Node& m : plan.nodes) {
                    const bool reads = std::find(m.in.begin(), m.in.end(), t) != m.in.end() || m.res == t;
                    if (!reads) continue;
                    n_cons++;
                    if (m.kind != Node::Reshape && m.kind != Node::Concat) return false;
                    next_t = m.out;
                }
                if (n_cons != 1) return false;
                t = next_t;
            }
            return false;
        };
        auto heads_on = [&](int t, size_t from) {
            std::vector<size_t> hs;
            for (size_t k = from; k < plan.nodes.size(); k++) {
                const Node& m = plan.nodes[k];
                if (absorbed[k] || m.in.size() != 1 || m.in[0] != t || m.res >= 0 || m.act != ACT_NONE) continue;
                const bool pw = (m.kind == Node::Conv && m.KH == 1 && m.KW == 1 && m.sh == 1 && m.sw == 1 && !m.gemm_head) || (m.kind == Node::Block && m.w < 0);
                if (pw && view_only(m.out)) hs.push_back(k);
            }
            return hs;
        };
        auto attach_edges = [&](Node& c, std::vector<Node>& done, size_t& next, ChainArgs ca) {
            const auto& sc = g.tensors[c.in[0]].shape;  // H x W x C of the resident frame
            const int tin = c.in[0], tout = c.out;
            if (!done.empty() && (sc[1] % 1) == 0) {
                const Node& e = done.back();
                const auto& se = g.tensors[e.in.empty() ? tin : e.in[0]].shape;
                const int users = c.members.front().res == tin ? 2 : 1;
                if (edge_block(e) && e.out == tin && uses(tin) == users && !is_output(tin) && se.size() == 4 && se[1] == 2 * sc[1] && se[2] == 2 * sc[2] &&
                    se[3] % 8 == 0 && se[3] <= sc[3]) {
                    ca.pre.on = 1; ca.pre.Cin = se[3]; ca.pre.in = reinterpret_cast<const float*>(0x3000); ca.pre.in_fs = static_cast<long>(g.tensors[e.in[0]].elems());
                    if (chain_kernel_supports(ca)) {
                        c.members.insert(c.members.begin(), e);
                        c.src_ops.insert(c.src_ops.begin(), e.src_ops.begin(), e.src_ops.end());
                        c.in = {e.in[0]};
                        c.chain_pre = true;
                        done.pop_back();
                    } else {
                        ca.pre.on = 0;
                    }
                }
            }
            if (next < plan.nodes.size()) {
                const Node& e = plan.nodes[next];
                const auto& so = g.tensors[e.out].shape;
                if (edge_block(e) && e.in[0] == tout && so.size() == 4 && sc[1] % 2 == 0 && sc[2] % 2 == 0 && so[1] * 2 == sc[1] && so[2] * 2 == sc[2]) {
                    ca.post.on = 1; ca.post.Co = so[3]; ca.post.out = reinterpret_cast<float*>(0x4000); ca.post.out_fs = static_cast<long>(g.tensors[e.out].elems());
                    if (chain_kernel_supports(ca)) {
                        const bool others = uses(tout) > (e.res == tout ? 2 : 1) || is_output(tout);
                        c.members.push_back(e);
                        c.src_ops.insert(c.src_ops.end(), e.src_ops.begin(), e.src_ops.end());
                        c.chain_post = true;
                        if (others) c.extra_out = {e.out};   // the chain's own output is still read (output heads): both are written
                        else c.out = e.out;
                        next++;
                    }
                }
            }
            // output heads on the chain's final frame (src 0) and on the block behind it (src 1)
            const int t_post = c.chain_post ? c.members.back().out : -1;
            for (int src = 0; src < 2; src++) {
                const int t = src == 0 ? tout : t_post;
                if (t < 0) continue;
                std::vector<size_t> hs = heads_on(t, next);
                if (!fuse_heads || hs.empty() || hs.size() > 2) continue;
                // a = the head whose channel count is a multiple of 4 (the regressors), b = the other one
                auto co = [&](size_t k) { return g.tensors[plan.nodes[k].out].shape.back(); };
                size_t ka = hs[0], kb = hs.size() > 1 ? hs[1] : static_cast<size_t>(-1);
                if (hs.size() > 1 && (co(ka) % 4 != 0 || (co(kb) % 4 == 0 && co(kb) > co(ka)))) std::swap(ka, kb);
                if (co(ka) % 4 != 0) continue;
                ChainArgs cb = ca;
                ChainHead& H = cb.heads[src];
                H.on = 1; H.src = src; H.Co_a = co(ka); H.Co_b = kb != static_cast<size_t>(-1) ? co(kb) : 0;
                H.w_pw = reinterpret_cast<const float*>(0x5000); H.out_a = reinterpret_cast<float*>(0x6000); H.out_b = H.Co_b ? reinterpret_cast<float*>(0x7000) : nullptr;
                H.out_a_fs = H.out_b_fs = 64;
                if (!chain_kernel_supports(cb)) continue;
                ca = cb;
                Node::HeadPair hp;
                hp.src = src;
                hp.a = static_cast<int>(c.head_nodes.size());
                c.head_nodes.push_back(plan.nodes[ka]);
                if (H.Co_b) { hp.b = static_cast<int>(c.head_nodes.size()); c.head_nodes.push_back(plan.nodes[kb]); }
                c.head_pairs.push_back(hp);
                for (size_t k : {ka, kb}) {
                    if (k == static_cast<size_t>(-1)) continue;
                    absorbed[k] = 1;
                    c.extra_out.push_back(plan.nodes[k].out);
                    c.src_ops.insert(c.src_ops.end(), plan.nodes[k].src_ops.begin(), plan.nodes[k].src_ops.end());
                }
            }
            // the chain's final frame needs no memory when every reader runs inside this launch
            if (c.chain_post && c.out == tout && !is_output(tout)) {
                bool read = false;
                for (size_t k = next; k < plan.nodes.size(); k++) {
                    const Node& m = plan.nodes[k];
                    read |= !absorbed[k] && (std::find(m.in.begin(), m.in.end(), tout) != m.in.end() || m.res == tout);
                }
                if (!read) {
                    c.out = t_post;
                    c.extra_out.erase(std::find(c.extra_out.begin(), c.extra_out.end(), t_post));
                }
            }
        };
        const int pipe_max = fuse_level >= 4 ? pipe_max_opt : 0;
        std::vector<Node> fusedv;
        for (size_t i = 0; i < plan.nodes.size();) {
            if (absorbed[i]) { i++; continue; }  // an output head that runs inside an earlier chain's launch
            if (chainable(plan.nodes[i])) {
                const auto& si = g.tensors[plan.nodes[i].in[0]].shape;
                size_t j = i;
                while (links(j)) j++;
                const int run = static_cast<int>(j - i + 1);
                if (run >= 2) {
                    // (a) the whole frame fits in LDS: frame-resident chain kernel
                    ChainArgs ca;
                    ca.in = reinterpret_cast<const float*>(0x1000); ca.out = reinterpret_cast<float*>(0x2000);
                    ca.in_fs = ca.out_fs = static_cast<long>(g.tensors[plan.nodes[i].in[0]].elems());
                    ca.B = 1; ca.H = si[1]; ca.W = si[2]; ca.C = si[3];
                    ca.nblocks = std::min(run, kMaxChain);
                    if (si[1] * si[2] <= 256 && si[3] % 8 == 0 && chain_kernel_supports(ca)) {
                        Node c = make_chain(i, i + ca.nblocks - 1);
                        i += ca.nblocks;
                        if (fuse_level >= 5) attach_edges(c, fusedv, i, ca);
                        fusedv.push_back(std::move(c));
                        continue;
                    }
                }
                // (b) narrow layers: row-pipelined groups of up to 4 blocks, the intermediate rows handed over in LDS; the
                // stride-2 block that follows the run (max-pool skip from its own input) can be the last stage of the last group
                if (pipe_max >= 2 && strip_pipe_shape_ok(si[3], si[2])) {
                    bool tail = false;
                    if (j + 1 < plan.nodes.size()) {
                        const Node& t = plan.nodes[j + 1];
                        tail = t.kind == Node::Block && t.w >= 0 && t.sh == 2 && t.sw == 2 && t.padding == Padding::Same && t.in[0] == plan.nodes[j].out &&
                               t.res == t.in[0] && t.res_mode == RES_MAXPOOL && t.act == ACT_RELU && plan.nodes[i].act == ACT_RELU &&
                               uses(plan.nodes[j].out) == 2 && std::find(g.outputs.begin(), g.outputs.end(), plan.nodes[j].out) == g.outputs.end() &&
                               strip_tail_shape_ok(si[3], g.tensors[t.out].shape[3], si[1], si[2]);
                    }
                    int left = run + (tail ? 1 : 0);
                    if (left >= 2) {
                        while (left >= 2) {
                            int take = std::min(left, std::min(pipe_max, 4));
                            if (left - take == 1 && take > 2) take--;  // never leave a single block behind a full group
                            fusedv.push_back(make_chain(i, i + take - 1));
                            i += take;
                            left -= take;
                        }
                        continue;  // a leftover single block is handled by the next iteration as a plain node
                    }
                }
            }
            fusedv.push_back(plan.nodes[i]);
            i++;
        }
        plan.nodes = std::move(fusedv);
    }

    // ---- whole-frame convolutions (VALID, window = frame, one output pixel: the mesh and iris output heads) as batch GEMMs
    if (fuse_level >= 2) {
        for (Node& n : plan.nodes) {
            if (n.kind != Node::Conv || n.res >= 0 || n.in.size() != 1) continue;
            const auto& si = g.tensors[n.in[0]].shape;
            const auto& so = g.tensors[n.out].shape;
            if (si.size() != 4 || so.size() != 4 || si[1] != n.KH || si[2] != n.KW || so[1] != 1 || so[2] != 1) continue;
            if (n.padding != Padding::Valid && !(n.KH == 1 && n.KW == 1)) continue;
            if (si[1] * si[2] < 2) continue;  // 1x1 frames: pointwise, the stage programs / block kernels have them
            n.gemm_head = fuse_heads && head_gemm_supports(n.KH * n.KW * si[3], so[3]);
        }
    }

    // ---- level 5: frame-resident stage programs
    if (fuse_level >= 5) {
        reorder_branches(plan.nodes);
        plan.nodes = group_resident(g, plan.nodes, res_budget_bytes, tail);
    }
    // ---- tail branches: behind the LAST fork of the plan the remaining launches fall into chains that do not need each other (the
    // output heads of the iris / face mesh networks: 16 + 16 stage-program nodes).  branch[i] = chain of node i (0 = stays on the trunk
    // stream, >= 1 = may run on a side stream), -1 for everything up to the fork.  A node that needs two different chains joins them: no
    // branches then.
    plan.branch.assign(plan.nodes.size(), -1);
    if (fuse_level >= 5) {
        const std::vector<std::vector<char>> dep = dependence(plan.nodes);
        auto is_view = [](const Node& n) { return n.kind == Node::Reshape || n.kind == Node::Concat; };
        int fork_at = -1;
        if (dep.size() == plan.nodes.size())
        for (size_t j = 0; j < plan.nodes.size(); j++)
            if (!is_view(plan.nodes[j]) && is_fork(plan.nodes, dep, j)) fork_at = static_cast<int>(j);
        if (fork_at >= 0) {
            std::vector<int> br(plan.nodes.size(), -1);
            int next = 0;
            bool ok = true;
            for (size_t i = static_cast<size_t>(fork_at) + 1; i < plan.nodes.size() && ok; i++) {
                int mine = -1;
                for (size_t a = static_cast<size_t>(fork_at) + 1; a < i; a++)
                    if (dep[i][a] && br[a] >= 0) {
                        if (mine >= 0 && mine != br[a]) ok = false;
                        mine = br[a];
                    }
                if (is_view(plan.nodes[i]) && mine < 0) continue;  // a view of a tensor from before the fork
                br[i] = mine >= 0 ? mine : next++;
            }
            if (ok && next >= 2) plan.branch = br;
        }
    }

    // ---- storage: RESHAPE = view of its input; CONCATENATION inputs live inside the joined buffer.
    const int NT = static_cast<int>(g.tensors.size());
    plan.storage.resize(NT);
    for (int t = 0; t < NT; t++) plan.storage[t] = Storage{t, 0, static_cast<long>(g.tensors[t].elems())};
    for (int i = static_cast<int>(plan.nodes.size()) - 1; i >= 0; i--) {
        const Node& n = plan.nodes[i];
        if (n.kind == Node::Reshape) {
            plan.storage[n.in[0]] = plan.storage[n.out];
        } else if (n.kind == Node::Concat) {
            const auto& so = g.tensors[n.out].shape;
            int axis = n.axis < 0 ? n.axis + static_cast<int>(so.size()) : n.axis;
            // (a crafted graph's axis may lie outside the output's shape: so[d] below must not be read beyond it — found by the mutation test
            // under AddressSanitizer, a read of four bytes behind the shape vector)
            if (axis < 1 || axis >= static_cast<int>(so.size())) throw std::runtime_error("plan: CONCATENATION axis outside the output's shape");
            long outer = 1;
            for (int d = 1; d < axis; d++) outer *= so[d];
            if (outer != 1) throw std::runtime_error("plan: CONCATENATION must join the first non-batch axis");
            long off = 0;
            for (int t : n.in) {
                if (plan.storage[t].root != t) throw std::runtime_error("plan: tensor feeds two concatenations/views");
                plan.storage[t] = Storage{plan.storage[n.out].root, plan.storage[n.out].offset + off, plan.storage[n.out].frame_stride};
                off += static_cast<long>(g.tensors[t].elems());
            }
        }
    }
    // views created above must be propagated forward to tensors whose storage pointed at an intermediate root
    for (int pass = 0; pass < 4; pass++)
        for (int t = 0; t < NT; t++) {
            Storage& s = plan.storage[t];
            const Storage& r = plan.storage[s.root];
            if (r.root != s.root) s = Storage{r.root, r.offset + s.offset, r.frame_stride};
        }

    // ---- liveness + first-fit arena (offsets in floats per frame; scaled by the chunk size at run time)
    const int NN = static_cast<int>(plan.nodes.size());
    std::vector<int> first(NT, -2), last(NT, -2);
    auto touch = [&](int t, int step) {
        if (t < 0 || g.tensors[t].is_const) return;
        int r = plan.storage[t].root;
        if (first[r] == -2 || step < first[r]) first[r] = step;
        if (step > last[r]) last[r] = step;
    };
    touch(g.inputs[0], -1);
    for (int i = 0; i < NN; i++) {
        for (int t : plan.nodes[i].in) touch(t, i);
        touch(plan.nodes[i].res, i);
        touch(plan.nodes[i].out, i);
        for (int t : plan.nodes[i].extra_out) touch(t, i);
    }
    for (int t : g.outputs) touch(t, NN);
    // two plain BlazeBlocks of one shape in a row may run as ONE launch (engine.cpp: mdblock_kernel, pair form) that reads the first one's input while it
    // writes the second one's output: that input stays allocated one node longer
    for (int i = 0; i + 1 < NN; i++) {
        const Node &pa = plan.nodes[i], &pb = plan.nodes[i + 1];
        if (pa.kind != Node::Block || pb.kind != Node::Block || pa.w < 0 || pb.w < 0 || pa.in.size() != 1 || pb.in.size() != 1 || pb.in[0] != pa.out) continue;
        if (pa.sh != 1 || pb.sh != 1 || g.tensors[pa.in[0]].shape != g.tensors[pb.out].shape) continue;
        touch(pa.in[0], i + 1);
    }
    // tail branches may run side by side in any interleaving: everything they read or write stays allocated to the end of the plan
    for (int i = 0; i < NN; i++) {
        if (plan.branch[static_cast<size_t>(i)] < 0) continue;
        for (int t : plan.nodes[i].in) touch(t, NN);
        touch(plan.nodes[i].res, NN);
        touch(plan.nodes[i].out, NN);
        for (int t : plan.nodes[i].extra_out) touch(t, NN);
    }
    plan.root_offset.assign(NT, -1);
    plan.root_elems.assign(NT, 0);
    struct Live { long off, size; int until; };
    std::vector<Live> live;
    std::vector<int> order;
    for (int t = 0; t < NT; t++)
        if (first[t] != -2 && plan.storage[t].root == t && !g.tensors[t].is_const) order.push_back(t);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
    long high = 0;
    for (int t : order) {
        long size = (plan.storage[t].frame_stride + 63) & ~63L;  // 256-byte granules
        // buffers whose last reader is an earlier step than this tensor's first writer can be recycled
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return l.until < first[t]; }), live.end());
        std::sort(live.begin(), live.end(), [](const Live& a, const Live& b) { return a.off < b.off; });
        long off = 0;
        for (const Live& l : live) {
            if (off + size <= l.off) break;
            off = std::max(off, l.off + l.size);
        }
        plan.root_offset[t] = off;
        plan.root_elems[t] = size;
        live.push_back({off, size, last[t]});
        high = std::max(high, off + size);
    }
    plan.arena_floats_per_frame = high;

    // ---- algorithmic traffic / MACs of the plan as launched (per frame)
    double bytes = 0, macs = 0;
    for (const Node& n : plan.nodes) {
        auto elems = [&](int t) { return t < 0 ? 0.0 : (static_cast<size_t>(t) < logical_elems.size() ? logical_elems[t] : static_cast<double>(g.tensors[t].elems())); };
        auto LC = [&](int t) { return static_cast<size_t>(t) < logical_C.size() ? logical_C[t] : g.tensors[t].shape.back(); };
        if (n.kind == Node::Reshape || n.kind == Node::Concat) continue;
        for (int t : n.in) bytes += 4 * elems(t);
        bytes += 4 * elems(n.out);
        for (int t : n.extra_out) bytes += 4 * elems(t);
        if (n.kind == Node::Resident) {
            for (const Node& m : n.members) {
                for (int c : {m.w, m.b, m.w2, m.b2, m.alpha}) bytes += 4 * elems(c);
                const auto& so2 = g.tensors[m.out].shape;
                const int Cm = LC(m.in[0]);
                if (m.kind == Node::Conv) macs += elems(m.out) * m.KH * m.KW * Cm;
                else macs += static_cast<double>(so2[1]) * so2[2] * Cm * ((m.w >= 0 ? 9 : 0) + LC(m.out));
            }
            continue;
        }
        if (n.kind == Node::Chain) {
            for (const Node& m : n.members) {
                for (int c : {m.w, m.b, m.w2, m.b2, m.alpha}) bytes += 4 * elems(c);
                const auto& so2 = g.tensors[m.out].shape;
                macs += static_cast<double>(so2[1]) * so2[2] * LC(m.in[0]) * (9 + LC(m.out));
            }
            for (const Node& m : n.head_nodes) {
                for (int c : {m.w, m.b, m.w2, m.b2}) bytes += 4 * elems(c);
                macs += elems(m.out) * LC(m.in[0]);
            }
            continue;
        }
        if (n.res >= 0 && !(n.kind == Node::Block && n.res == n.in[0])) bytes += 4 * elems(n.res);
        for (int c : {n.w, n.b, n.w2, n.b2, n.alpha}) bytes += 4 * elems(c);
        const auto& so = g.tensors[n.out].shape;
        if (n.kind == Node::Conv) macs += elems(n.out) * n.KH * n.KW * LC(n.in[0]);
        if (n.kind == Node::Dw) macs += elems(n.out) * n.KH * n.KW;
        if (n.kind == Node::Block) {
            int C = LC(n.in[0]);
            macs += static_cast<double>(so[1]) * so[2] * C * ((n.w >= 0 ? n.KH * n.KW : 0) + LC(n.out));
        }
    }
    plan.bytes_per_frame = bytes;
    plan.macs_per_frame = macs;
    (void)same_pad;
    return plan;
}
}  // namespace

std::string Plan::describe() const {
    static const char* kinds[] = {"conv", "dw", "block", "add", "act", "maxpool", "pad", "reshape", "concat", "resize", "d2s", "chain", "resident"};
    static const char* acts[] = {"", "+relu", "+relu6", "+prelu"};
    static const char* res[] = {"", "+skip", "+skip(maxpool)", "+skip(up2x)"};
    std::ostringstream os;
    os << "# fuse_level=" << fuse_level << " launches=" << nodes.size() << " arena_floats_per_frame=" << arena_floats_per_frame
       << " bytes_per_frame=" << static_cast<long long>(bytes_per_frame) << " macs_per_frame=" << static_cast<long long>(macs_per_frame) << "\n";
    for (const Node& n : nodes) {
        const auto& si = graph.tensors[n.in[0]].shape;
        const auto& so = graph.tensors[n.out].shape;
        os << kinds[n.kind] << acts[n.act] << res[n.res_mode] << " t" << n.in[0] << "[";
        for (size_t d = 1; d < si.size(); d++) os << (d > 1 ? "x" : "") << si[d];
        os << "] -> t" << n.out << "[";
        for (size_t d = 1; d < so.size(); d++) os << (d > 1 ? "x" : "") << so[d];
        os << "]";
        if (n.kind == Node::Conv || n.kind == Node::Dw || (n.kind == Node::Block && n.w >= 0)) os << " k" << n.KH << "x" << n.KW << " s" << n.sh;
        if (n.kind == Node::Conv && n.gemm_head) os << " whole-frame window: GEMM over the batch";
        if (n.ept >= 0) os << " pad" << n.ept << "/" << n.epl;
        if (n.kind == Node::Block && n.w < 0) os << " pointwise";
        if (n.kind == Node::Chain)
            if (n.chain_pre || n.chain_post)
                os << " x" << n.members.size() << " blocks, frame resident in LDS" << (n.chain_pre ? ", stride-2 block in front" : "") << (n.chain_post ? ", stride-2 block behind" : "")
                   << (n.head_nodes.empty() ? "" : ", " + std::to_string(n.head_nodes.size()) + " output heads");
            else
                os << " x" << n.members.size() << " blocks, " << (si[1] * si[2] <= 256 ? "frame resident in LDS" : "row-pipelined through LDS")
                   << (n.members.back().sh == 2 ? " (stride-2 tail)" : "") << (n.head_nodes.empty() ? "" : ", " + std::to_string(n.head_nodes.size()) + " output heads");
        if (n.kind == Node::Resident && n.dblock) {
            os << (n.members[0].res >= 0 ? " two BlazeBlocks (" : " double block (") << graph.tensors[n.members[0].in[0]].shape[3] << " -> " << graph.tensors[n.members[0].out].shape[3] << " -> " << graph.tensors[n.members[1].out].shape[3]
               << " channels), walking row bands, narrow tensor in LDS";
        } else if (n.kind == Node::Resident && n.xc) {
            os << " x" << n.members.size() / 2 << " expand / contract pairs (" << graph.tensors[n.members[0].in[0]].shape[3] << " <-> " << graph.tensors[n.members[0].out].shape[3]
               << " channels), frame resident, depthwise stages on the fly";
        } else if (n.kind == Node::Resident && n.bneck) {
            os << " x" << n.members.size() / 2 << " bottleneck blocks, wide tensor in registers, " << (n.res_bands > 1 ? std::to_string(n.res_bands) + " row bands" : std::string("frame resident"));
        } else if (n.kind == Node::Resident && n.tail) {
            os << " x" << n.members.size() << " nodes in " << n.stages.size() << " stages, several frames per workgroup, " << n.tail_frame_floats * 4 << " B LDS per frame";
            for (int t : n.extra_out) os << " +t" << t;
        } else if (n.kind == Node::Resident) {
            os << " x" << n.members.size() << " nodes in " << n.stages.size() << " stages, " << (n.res_bands > 1 ? "row-band resident" : "frame resident") << ", " << n.res_lds_bytes << " B LDS";
            if (n.res_bands > 1) os << ", " << n.res_bands << " bands of " << n.stages[0].st.band_rows << " rows";
            for (int t : n.extra_out) os << " +t" << t;
        }
        if (n.kind == Node::Chain)
            for (int t : n.extra_out) os << " +t" << t;
        os << " ops{";
        for (size_t k = 0; k < n.src_ops.size(); k++) os << (k ? "," : "") << n.src_ops[k];
        os << "}\n";
    }
    return os.str();
}

}  // namespace mi
