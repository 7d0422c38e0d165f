// tflite_graph.cpp — TFL3 flatbuffer reader (see tflite_graph.hpp).  Field numbers: TFLite schema v3
// (Model: 1 operator_codes, 2 subgraphs, 3 description, 4 buffers; SubGraph: 0 tensors, 1 inputs, 2 outputs,
// 3 operators; Tensor: 0 shape, 1 type, 2 buffer, 3 name, 6 sparsity; Operator: 0 opcode_index, 1 inputs, 2 outputs,
// 4 builtin_options; OperatorCode: 0 deprecated_builtin_code, 3 builtin_code).
#include "tflite_graph.hpp"

#include <cstring>
#include <stdexcept>

namespace mi {
namespace {

class View {
   public:
    View(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    template <typename T>
    T at(size_t off) const {
        if (off > n_ || n_ - off < sizeof(T)) throw std::runtime_error("tflite: read past end of buffer");  // no off + size: it may wrap
        T v;
        std::memcpy(&v, p_ + off, sizeof(T));
        return v;
    }
    const uint8_t* ptr(size_t off, size_t len) const {
        if (off > n_ || n_ - off < len) throw std::runtime_error("tflite: span past end of buffer");
        return p_ + off;
    }
    size_t size() const { return n_; }

   private:
    const uint8_t* p_;
    size_t n_;
};

struct Vec {
    size_t start = 0;
    uint32_t len = 0;
};

class Table {
   public:
    Table(const View* v, size_t pos) : v_(v), pos_(pos) {
        int32_t soff = v_->at<int32_t>(pos_);
        vt_ = static_cast<size_t>(static_cast<int64_t>(pos_) - soff);
        vsize_ = v_->at<uint16_t>(vt_);
    }
    size_t field(int k) const {
        size_t slot = 4 + 2 * static_cast<size_t>(k);
        if (slot + 2 > vsize_) return 0;
        uint16_t off = v_->at<uint16_t>(vt_ + slot);
        return off ? pos_ + off : 0;
    }
    template <typename T>
    T scalar(int k, T def) const {
        size_t f = field(k);
        return f ? v_->at<T>(f) : def;
    }
    Vec vec(int k) const {
        size_t f = field(k);
        if (!f) return {};
        size_t v = f + v_->at<uint32_t>(f);
        const uint32_t len = v_->at<uint32_t>(v);
        if (len > v_->size()) throw std::runtime_error("tflite: vector longer than the file");  // bounds every allocation sized by it
        return {v + 4, len};
    }
    Table table_at(const Vec& vv, uint32_t i) const {
        size_t e = vv.start + 4 * static_cast<size_t>(i);
        return Table(v_, e + v_->at<uint32_t>(e));
    }
    Table sub(int k) const {
        size_t f = field(k);
        if (!f) throw std::runtime_error("tflite: missing table field");
        return Table(v_, f + v_->at<uint32_t>(f));
    }
    bool has(int k) const { return field(k) != 0; }
    std::vector<int> ints(int k) const {
        Vec vv = vec(k);
        v_->ptr(vv.start, 4 * static_cast<size_t>(vv.len));  // the whole vector lies inside the file
        std::vector<int> out(vv.len);
        for (uint32_t i = 0; i < vv.len; i++) out[i] = v_->at<int32_t>(vv.start + 4 * static_cast<size_t>(i));
        return out;
    }
    std::string str(int k) const {
        Vec vv = vec(k);
        if (!vv.len) return {};
        return std::string(reinterpret_cast<const char*>(v_->ptr(vv.start, vv.len)), vv.len);
    }
    const View* view() const { return v_; }

   private:
    const View* v_;
    size_t pos_, vt_ = 0;
    uint16_t vsize_ = 0;
};

float half_to_float(uint16_t h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
    int exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal half -> normal float
            int shift = 0;
            while (!(man & 0x400u)) {
                man <<= 1;
                ++shift;
            }
            bits = sign | static_cast<uint32_t>(127 - 14 - shift) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | man << 13;
    } else {
        bits = sign | static_cast<uint32_t>(exp + 112) << 23 | man << 13;
    }
    float f;
    std::memcpy(&f, &bits, sizeof f);
    return f;
}

// ---- sparse constants (DENSIFY): TFLite sparsity format, traversal order + per-level dense/CSR metadata.
struct Level {
    int format = 0, dense_size = 0;
    std::vector<int64_t> segments, indices;
};

std::vector<int64_t> read_index_vector(const Table& dm, int type_field, int value_field) {
    std::vector<int64_t> out;
    int type = dm.scalar<uint8_t>(type_field, 0);
    if (!type) return out;
    Table t = dm.sub(value_field);
    Vec vv = t.vec(0);
    const View* v = t.view();
    v->ptr(vv.start, static_cast<size_t>(vv.len) * (type == 1 ? 4 : type == 2 ? 2 : 1));
    out.resize(vv.len);
    for (uint32_t i = 0; i < vv.len; i++) {
        switch (type) {
            case 1: out[i] = v->at<int32_t>(vv.start + 4 * static_cast<size_t>(i)); break;
            case 2: out[i] = v->at<uint16_t>(vv.start + 2 * static_cast<size_t>(i)); break;
            case 3: out[i] = v->at<uint8_t>(vv.start + i); break;
            default: throw std::runtime_error("tflite: unknown sparse index vector type");
        }
    }
    return out;
}

template <typename Fetch>
void densify(const Table& sp, const std::vector<int>& shape, Fetch fetch, std::vector<float>& dense) {
    std::vector<int> order = sp.ints(0), block_map = sp.ints(1);
    Vec dmv = sp.vec(2);
    std::vector<Level> levels(dmv.len);
    for (uint32_t i = 0; i < dmv.len; i++) {
        Table dm = sp.table_at(dmv, i);
        levels[i].format = dm.scalar<int8_t>(0, 0);
        levels[i].dense_size = dm.scalar<int32_t>(1, 0);
        levels[i].segments = read_index_vector(dm, 2, 3);
        levels[i].indices = read_index_vector(dm, 4, 5);
    }
    const int nd = static_cast<int>(shape.size());
    const int nl = static_cast<int>(order.size());
    // Everything below indexes with file contents: validate them first (a malformed blob must end in MI_EMODEL, not in an
    // out-of-bounds access). traversal_order is a permutation of [0, nl); nl = nd + number of block dimensions; every block
    // dimension maps to a real dimension; dense sizes are positive; CSR levels carry both of their vectors.
    const int nb = static_cast<int>(block_map.size());
    if (nd < 1 || static_cast<int>(levels.size()) != nl || nl != nd + nb) throw std::runtime_error("tflite: sparse metadata mismatch");
    {
        std::vector<char> seen(nl, 0);
        for (int v : order) {
            if (v < 0 || v >= nl || seen[v]) throw std::runtime_error("tflite: sparse traversal order is not a permutation");
            seen[v] = 1;
        }
        for (int v : block_map)
            if (v < 0 || v >= nd) throw std::runtime_error("tflite: sparse block map out of range");
        for (const Level& L : levels) {
            if (L.format == 0 && L.dense_size <= 0) throw std::runtime_error("tflite: sparse dense level without a size");
            if (L.format != 0 && L.format != 1) throw std::runtime_error("tflite: unknown sparse dimension format");
            if (L.format == 1 && L.segments.empty()) throw std::runtime_error("tflite: sparse CSR level without segments");
        }
        for (int d : shape)
            if (d <= 0) throw std::runtime_error("tflite: sparse tensor with an empty shape");
    }
    std::vector<int> block_size;
    for (int b = 0; b < nb; b++) {
        if (levels[nd + b].dense_size <= 0) throw std::runtime_error("tflite: sparse block without a size");
        block_size.push_back(levels[nd + b].dense_size);
    }
    std::vector<int> idx(nl, 0);
    // iterative DFS over the levels
    struct Frame {
        int level;
        int64_t prev, cur, end;
    };
    std::vector<Frame> stack;
    auto open_level = [&](int level, int64_t prev) {
        const Level& L = levels[level];
        if (L.format == 0) stack.push_back({level, prev, 0, L.dense_size});
        else {
            if (prev < 0 || static_cast<size_t>(prev) + 1 >= L.segments.size()) throw std::runtime_error("tflite: sparse segment index out of range");
            const int64_t lo = L.segments[prev], hi = L.segments[prev + 1];
            if (lo < 0 || hi < lo || static_cast<size_t>(hi) > L.indices.size()) throw std::runtime_error("tflite: sparse segment out of range");
            stack.push_back({level, prev, lo, hi});
        }
    };
    open_level(0, 0);
    while (!stack.empty()) {
        Frame& f = stack.back();
        if (f.cur >= f.end) {
            stack.pop_back();
            continue;
        }
        const Level& L = levels[f.level];
        int64_t child;
        if (L.format == 0) {
            idx[f.level] = static_cast<int>(f.cur);
            child = f.prev * L.dense_size + f.cur;
        } else {
            const int64_t at = L.indices.at(f.cur);
            if (at < 0 || at > 0x3fffffff) throw std::runtime_error("tflite: sparse index out of range");
            idx[f.level] = static_cast<int>(at);
            child = f.cur;
        }
        int level = f.level;
        f.cur++;
        if (level + 1 == nl) {
            std::vector<int> coord(nl);
            for (int lv = 0; lv < nl; lv++) coord[order[lv]] = idx[lv];
            std::vector<int64_t> orig(coord.begin(), coord.begin() + nd);
            for (size_t b = 0; b < block_map.size(); b++)
                orig[block_map[b]] = orig[block_map[b]] * block_size[b] + coord[nd + static_cast<int>(b)];
            size_t lin = 0;
            for (int d = 0; d < nd; d++) {
                if (orig[d] < 0 || orig[d] >= shape[d]) throw std::runtime_error("tflite: sparse index out of range");
                lin = lin * static_cast<size_t>(shape[d]) + static_cast<size_t>(orig[d]);
            }
            if (lin >= dense.size()) throw std::runtime_error("tflite: sparse index out of range");
            dense[lin] = fetch(static_cast<size_t>(child));
        } else {
            open_level(level + 1, child);
        }
    }
}

}  // namespace

Graph parse_tflite(const uint8_t* data, size_t size) {
    if (size < 8 || std::memcmp(data + 4, "TFL3", 4) != 0) throw std::runtime_error("tflite: missing TFL3 identifier");
    View view(data, size);
    Table model(&view, view.at<uint32_t>(0));

    std::vector<int> codes;
    {
        Vec v = model.vec(1);
        for (uint32_t i = 0; i < v.len; i++) {
            Table oc = model.table_at(v, i);
            int a = oc.scalar<int8_t>(0, 0), b = oc.scalar<int32_t>(3, 0);
            codes.push_back(a > b ? a : b);
        }
    }
    Vec buffers = model.vec(4);
    Vec subgraphs = model.vec(2);
    if (subgraphs.len < 1) throw std::runtime_error("tflite: no subgraph");
    Table sg = model.table_at(subgraphs, 0);

    Graph g;
    g.description = model.str(3);
    g.inputs = sg.ints(1);
    g.outputs = sg.ints(2);

    Vec tv = sg.vec(0);
    g.tensors.resize(tv.len);
    for (uint32_t i = 0; i < tv.len; i++) {
        Table t = sg.table_at(tv, i);
        TensorInfo& ti = g.tensors[i];
        ti.shape = t.ints(0);
        ti.dtype = t.scalar<int8_t>(1, 0);
        ti.name = t.str(3);
        uint32_t bidx = t.scalar<uint32_t>(2, 0);
        if (bidx >= buffers.len) continue;
        Vec payload = model.table_at(buffers, bidx).vec(0);
        if (!payload.len) continue;
        const uint8_t* raw = view.ptr(payload.start, payload.len);
        // element count of a constant: non-negative dimensions, no overflow, and a size a real model can have (the largest
        // constant in the seven shipped graphs has 0.6 M elements) — a mutated shape must not drive a multi-GB allocation
        size_t n = 1;
        for (int d : ti.shape) {
            if (d < 0 || (d > 0 && n > (size_t(1) << 24) / static_cast<size_t>(d))) throw std::runtime_error("tflite: constant tensor shape out of range");
            n *= static_cast<size_t>(d);
        }
        ti.is_const = true;
        auto f32_at = [&](size_t k) {
            float f;
            if ((k + 1) * 4 > payload.len) throw std::runtime_error("tflite: constant shorter than its shape");
            std::memcpy(&f, raw + 4 * k, 4);
            return f;
        };
        auto f16_at = [&](size_t k) {
            uint16_t h;
            if ((k + 1) * 2 > payload.len) throw std::runtime_error("tflite: constant shorter than its shape");
            std::memcpy(&h, raw + 2 * k, 2);
            return half_to_float(h);
        };
        if (t.has(6)) {  // sparse
            ti.f32.assign(n, 0.f);
            Table sp = t.sub(6);
            if (ti.dtype == 0) densify(sp, ti.shape, f32_at, ti.f32);
            else if (ti.dtype == 1) densify(sp, ti.shape, f16_at, ti.f32);
            else throw std::runtime_error("tflite: unsupported sparse tensor type");
        } else if (ti.dtype == 0) {
            if (n * 4 > payload.len) throw std::runtime_error("tflite: constant shorter than its shape");
            ti.f32.resize(n);
            for (size_t k = 0; k < n; k++) ti.f32[k] = f32_at(k);
        } else if (ti.dtype == 1) {
            if (n * 2 > payload.len) throw std::runtime_error("tflite: constant shorter than its shape");
            ti.f32.resize(n);
            for (size_t k = 0; k < n; k++) ti.f32[k] = f16_at(k);
        } else if (ti.dtype == 2) {
            if (n * 4 > payload.len) throw std::runtime_error("tflite: constant shorter than its shape");
            ti.i32.resize(n);
            std::memcpy(ti.i32.data(), raw, n * 4);
        } else {
            ti.is_const = false;  // unsupported constant type: only an error if an op actually needs it
        }
    }

    Vec ov = sg.vec(3);
    for (uint32_t i = 0; i < ov.len; i++) {
        Table o = sg.table_at(ov, i);
        OpInfo op;
        uint32_t ci = o.scalar<uint32_t>(0, 0);
        if (ci >= codes.size()) throw std::runtime_error("tflite: opcode index out of range");
        op.raw_code = codes[ci];
        op.op = static_cast<BuiltinOp>(op.raw_code);
        op.inputs = o.ints(1);
        op.outputs = o.ints(2);
        for (size_t k = 0; k < op.inputs.size(); k++) {
            // -1 = optional input absent: only the bias slot of a convolution is optional in the operators taken here; a -1
            // anywhere else (an activation, a filter, PAD's paddings ...) is a malformed model, refused before any g.tensors[t]
            int t = op.inputs[k];
            bool optional_slot = k >= 2 && (op.op == BuiltinOp::Conv2D || op.op == BuiltinOp::DepthwiseConv2D);
            if (t < (optional_slot ? -1 : 0) || t >= static_cast<int>(g.tensors.size())) throw std::runtime_error("tflite: tensor index out of range");
        }
        for (int t : op.outputs)
            if (t < 0 || t >= static_cast<int>(g.tensors.size())) throw std::runtime_error("tflite: tensor index out of range");
        if (op.outputs.empty()) throw std::runtime_error("tflite: operator without outputs");
        if (o.has(4)) {
            Table opt = o.sub(4);
            switch (op.op) {
                case BuiltinOp::Conv2D:
                    op.padding = static_cast<Padding>(opt.scalar<int8_t>(0, 0));
                    op.stride_w = opt.scalar<int32_t>(1, 1);
                    op.stride_h = opt.scalar<int32_t>(2, 1);
                    op.act = static_cast<FusedAct>(opt.scalar<int8_t>(3, 0));
                    if (opt.scalar<int32_t>(4, 1) != 1 || opt.scalar<int32_t>(5, 1) != 1)
                        throw std::runtime_error("tflite: dilated convolution unsupported");
                    break;
                case BuiltinOp::DepthwiseConv2D:
                    op.padding = static_cast<Padding>(opt.scalar<int8_t>(0, 0));
                    op.stride_w = opt.scalar<int32_t>(1, 1);
                    op.stride_h = opt.scalar<int32_t>(2, 1);
                    op.depth_multiplier = opt.scalar<int32_t>(3, 1);
                    op.act = static_cast<FusedAct>(opt.scalar<int8_t>(4, 0));
                    if (opt.scalar<int32_t>(5, 1) != 1 || opt.scalar<int32_t>(6, 1) != 1)
                        throw std::runtime_error("tflite: dilated convolution unsupported");
                    break;
                case BuiltinOp::MaxPool2D:
                    op.padding = static_cast<Padding>(opt.scalar<int8_t>(0, 0));
                    op.stride_w = opt.scalar<int32_t>(1, 1);
                    op.stride_h = opt.scalar<int32_t>(2, 1);
                    op.filter_w = opt.scalar<int32_t>(3, 1);
                    op.filter_h = opt.scalar<int32_t>(4, 1);
                    op.act = static_cast<FusedAct>(opt.scalar<int8_t>(5, 0));
                    break;
                case BuiltinOp::Add: op.act = static_cast<FusedAct>(opt.scalar<int8_t>(0, 0)); break;
                case BuiltinOp::Concatenation:
                    op.axis = opt.scalar<int32_t>(0, 0);
                    op.act = static_cast<FusedAct>(opt.scalar<int8_t>(1, 0));
                    break;
                case BuiltinOp::ResizeBilinear:
                    op.align_corners = opt.scalar<uint8_t>(2, 0) != 0;
                    op.half_pixel_centers = opt.scalar<uint8_t>(3, 0) != 0;
                    break;
                case BuiltinOp::DepthToSpace: op.block_size = opt.scalar<int32_t>(0, 1); break;
                default: break;
            }
        }
        // Fold DEQUANTIZE / DENSIFY of constants: the output tensor simply becomes the (already f32, dense) constant.
        if (op.op == BuiltinOp::Dequantize || op.op == BuiltinOp::Densify) {
            const TensorInfo& src = g.tensors.at(op.inputs.at(0));
            if (!src.is_const || src.f32.empty())
                throw std::runtime_error("tflite: DEQUANTIZE/DENSIFY of a non-constant tensor is unsupported");
            TensorInfo& dst = g.tensors[op.outputs[0]];
            dst.is_const = true;
            dst.f32 = src.f32;
            dst.dtype = 0;
            continue;
        }
        g.ops.push_back(std::move(op));
    }
    if (g.inputs.size() != 1) throw std::runtime_error("tflite: expected exactly one graph input");
    if (g.tensors.at(g.inputs[0]).shape.size() != 4) throw std::runtime_error("tflite: graph input must be NHWC rank 4");
    return g;
}

}  // namespace mi
