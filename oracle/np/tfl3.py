"""TEST INFRASTRUCTURE — dependency-free TFL3 (.tflite flatbuffer) reader in pure Python.

Part of the *second, independent* CPU evaluation used only to cross-check the C oracle
(oracle/c) before goldens are frozen.  Nothing in the product path imports it.

The reference never parses these files itself: it hands them to the third-party `tflite` crate
0.9.8 (`FlatBufferModel::build_from_file`, /root/reference/src/face_detection_lite/face_detection.rs:188,
face_landmark.rs:216, iris_landmark.rs:150).  The field indices below follow the published TFLite
schema v3 (schema.fbs) — see SURVEY.md Appendix B, verified there against all seven shipped graphs.

Pin: the reference's tests hold no numeric assertions for this path; the C oracle this module cross-checks is
pinned by the reference's rendered PNGs on man.jpg, pixel for pixel (tests/test_pins.py, oracle/c/oracle.h).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

# builtin operator codes that appear in the seven graphs
ADD, CONCATENATION, CONV_2D, DEPTHWISE_CONV_2D, DEPTH_TO_SPACE, DEQUANTIZE = 0, 2, 3, 4, 5, 6
MAX_POOL_2D, RELU, RESHAPE, RESIZE_BILINEAR, PAD, PRELU, DENSIFY = 17, 19, 22, 23, 34, 54, 124
OP_NAMES = {
    ADD: "ADD", CONCATENATION: "CONCATENATION", CONV_2D: "CONV_2D", DEPTHWISE_CONV_2D: "DEPTHWISE_CONV_2D",
    DEPTH_TO_SPACE: "DEPTH_TO_SPACE", DEQUANTIZE: "DEQUANTIZE", MAX_POOL_2D: "MAX_POOL_2D", RELU: "RELU",
    RESHAPE: "RESHAPE", RESIZE_BILINEAR: "RESIZE_BILINEAR", PAD: "PAD", PRELU: "PRELU", DENSIFY: "DENSIFY",
}
TENSOR_TYPES = {0: np.float32, 1: np.float16, 2: np.int32, 3: np.uint8, 9: np.int8}


class _FB:
    """Minimal flatbuffer table walker."""

    def __init__(self, buf: bytes):
        self.b = buf

    def u8(self, o):
        return self.b[o]

    def i8(self, o):
        return struct.unpack_from("<b", self.b, o)[0]

    def u16(self, o):
        return struct.unpack_from("<H", self.b, o)[0]

    def i32(self, o):
        return struct.unpack_from("<i", self.b, o)[0]

    def u32(self, o):
        return struct.unpack_from("<I", self.b, o)[0]

    def indirect(self, o):
        return o + self.u32(o)

    def field(self, table, k):
        """Absolute offset of field k of `table`, or 0 when absent."""
        vt = table - self.i32(table)
        vsize = self.u16(vt)
        slot = 4 + 2 * k
        if slot >= vsize:
            return 0
        off = self.u16(vt + slot)
        return table + off if off else 0

    def vec(self, table, k):
        """(start, length) of vector field k, or (0, 0)."""
        f = self.field(table, k)
        if not f:
            return 0, 0
        v = self.indirect(f)
        return v + 4, self.u32(v)

    def vec_i32(self, table, k):
        s, n = self.vec(table, k)
        return list(struct.unpack_from("<%di" % n, self.b, s)) if n else []

    def vec_tables(self, table, k):
        s, n = self.vec(table, k)
        return [self.indirect(s + 4 * i) for i in range(n)]

    def string(self, table, k):
        s, n = self.vec(table, k)
        return self.b[s:s + n].decode("utf-8", "replace") if s else ""

    def scalar(self, table, k, kind, default=0):
        f = self.field(table, k)
        if not f:
            return default
        return getattr(self, kind)(f)


@dataclass
class Tensor:
    index: int
    shape: list
    dtype: type
    buffer: int
    name: str
    data: np.ndarray | None = None      # constant payload (raw dtype), None for activations
    sparsity: dict | None = None


@dataclass
class Op:
    index: int
    code: int
    inputs: list
    outputs: list
    opts: dict = field(default_factory=dict)

    @property
    def name(self):
        return OP_NAMES.get(self.code, "OP%d" % self.code)


@dataclass
class Graph:
    description: str
    tensors: list
    ops: list
    inputs: list
    outputs: list


def _parse_sparsity(fb: _FB, t):
    """SparsityParameters: 0 traversal_order[i32], 1 block_map[i32], 2 dim_metadata[DimensionMetadata].
    DimensionMetadata: 0 format (0 dense, 1 sparse CSR), 1 dense_size, 2 array_segments_type, 3 array_segments,
    4 array_indices_type, 5 array_indices.  Index vectors are unions Int32Vector/Uint16Vector/Uint8Vector
    (types 1/2/3), each a table with field 0 = values."""
    sp = {"traversal_order": fb.vec_i32(t, 0), "block_map": fb.vec_i32(t, 1), "dims": []}
    for dm in fb.vec_tables(t, 2):
        d = {"format": fb.scalar(dm, 0, "i8"), "dense_size": fb.scalar(dm, 1, "i32")}
        for nm, kt, kv in (("segments", 2, 3), ("indices", 4, 5)):
            ty = fb.scalar(dm, kt, "u8")
            arr = None
            if ty:
                tbl = fb.indirect(fb.field(dm, kv))
                s, n = fb.vec(tbl, 0)
                dt = {1: "<i4", 2: "<u2", 3: "u1"}[ty]
                arr = np.frombuffer(fb.b, dtype=dt, count=n, offset=s).astype(np.int64)
            d[nm] = arr
        sp["dims"].append(d)
    return sp


def load(path_or_bytes) -> Graph:
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    buf = bytes(buf)
    if len(buf) < 8 or buf[4:8] != b"TFL3":
        raise ValueError("not a TFL3 flatbuffer")
    fb = _FB(buf)
    model = fb.indirect(0)
    opcodes = []
    for oc in fb.vec_tables(model, 1):
        opcodes.append(max(fb.scalar(oc, 0, "i8"), fb.scalar(oc, 3, "i32")))
    buffers = []
    for bt in fb.vec_tables(model, 4):
        s, n = fb.vec(bt, 0)
        buffers.append((s, n))
    sg = fb.vec_tables(model, 2)[0]
    tensors = []
    for i, tt in enumerate(fb.vec_tables(sg, 0)):
        shape = fb.vec_i32(tt, 0)
        ty = fb.scalar(tt, 1, "i8")
        bidx = fb.scalar(tt, 2, "u32")
        dt = TENSOR_TYPES[ty]
        t = Tensor(i, shape, dt, bidx, fb.string(tt, 3))
        s, n = buffers[bidx]
        if n:
            t.data = np.frombuffer(buf, dtype=np.dtype(dt).newbyteorder("<"), count=n // np.dtype(dt).itemsize,
                                   offset=s)
        spf = fb.field(tt, 6)
        if spf:
            t.sparsity = _parse_sparsity(fb, fb.indirect(spf))
        tensors.append(t)
    ops = []
    for i, ot in enumerate(fb.vec_tables(sg, 3)):
        code = opcodes[fb.scalar(ot, 0, "u32")]
        op = Op(i, code, fb.vec_i32(ot, 1), fb.vec_i32(ot, 2))
        of = fb.field(ot, 4)
        o = fb.indirect(of) if of else 0
        if o:
            if code == CONV_2D:
                op.opts = dict(padding=fb.scalar(o, 0, "i8"), stride_w=fb.scalar(o, 1, "i32"),
                               stride_h=fb.scalar(o, 2, "i32"), act=fb.scalar(o, 3, "i8"),
                               dil_w=fb.scalar(o, 4, "i32", 1), dil_h=fb.scalar(o, 5, "i32", 1))
            elif code == DEPTHWISE_CONV_2D:
                op.opts = dict(padding=fb.scalar(o, 0, "i8"), stride_w=fb.scalar(o, 1, "i32"),
                               stride_h=fb.scalar(o, 2, "i32"), depth_multiplier=fb.scalar(o, 3, "i32"),
                               act=fb.scalar(o, 4, "i8"), dil_w=fb.scalar(o, 5, "i32", 1),
                               dil_h=fb.scalar(o, 6, "i32", 1))
            elif code == MAX_POOL_2D:
                op.opts = dict(padding=fb.scalar(o, 0, "i8"), stride_w=fb.scalar(o, 1, "i32"),
                               stride_h=fb.scalar(o, 2, "i32"), filter_w=fb.scalar(o, 3, "i32"),
                               filter_h=fb.scalar(o, 4, "i32"), act=fb.scalar(o, 5, "i8"))
            elif code == ADD:
                op.opts = dict(act=fb.scalar(o, 0, "i8"))
            elif code == CONCATENATION:
                op.opts = dict(axis=fb.scalar(o, 0, "i32"), act=fb.scalar(o, 1, "i8"))
            elif code == RESHAPE:
                op.opts = dict(new_shape=fb.vec_i32(o, 0))
            elif code == RESIZE_BILINEAR:
                op.opts = dict(align_corners=fb.scalar(o, 2, "u8"), half_pixel_centers=fb.scalar(o, 3, "u8"))
            elif code == DEPTH_TO_SPACE:
                op.opts = dict(block_size=fb.scalar(o, 0, "i32"))
        ops.append(op)
    return Graph(fb.string(model, 3), tensors, ops, fb.vec_i32(sg, 1), fb.vec_i32(sg, 2))


def dump(g: Graph) -> str:
    lines = ["# %s  inputs=%s outputs=%s" % (g.description, g.inputs, g.outputs)]
    for op in g.ops:
        def tdesc(i):
            if i < 0:
                return "-"
            t = g.tensors[i]
            c = "c" if t.data is not None else ""
            return "%d%s%s" % (i, c, t.shape)
        lines.append("%3d %-18s in=[%s] out=[%s] %s" % (
            op.index, op.name, ", ".join(tdesc(i) for i in op.inputs), ", ".join(tdesc(i) for i in op.outputs),
            " ".join("%s=%s" % kv for kv in op.opts.items())))
    return "\n".join(lines)


if __name__ == "__main__":
    import sys
    print(dump(load(sys.argv[1])))
