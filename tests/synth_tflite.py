"""Synthetic .tflite graphs for the parity tests (test infrastructure).

The six shipped graphs fix every layer shape the kernels meet in the product; these graphs use the same operator chains
(CONV_2D / DEPTHWISE_CONV_2D / ADD / PRELU / RELU / MAX_POOL_2D / PAD / RESHAPE / CONCATENATION, the op set of SURVEY.md Appendix A)
on OTHER shapes — ragged row bands, odd channel counts, frames with a partial last pixel group — so that the lowering and the
kernels are checked for what they claim to support, not only for the shapes of the reference's models.

A minimal TFL3 flatbuffer writer (the subset of the schema the reference's graphs use): tables are written parent first, the
children behind them, offsets patched afterwards (flatbuffer offsets point towards higher addresses).
"""
import struct

import numpy as np

# BuiltinOperator codes (tensorflow/lite/schema/schema.fbs)
ADD, CONCATENATION, CONV_2D, DEPTHWISE_CONV_2D, MAX_POOL_2D, RELU, RESHAPE, PAD, PRELU = 0, 2, 3, 4, 17, 19, 22, 34, 54
# BuiltinOptions union tags
OPT_CONV, OPT_DW, OPT_POOL, OPT_RESHAPE, OPT_CONCAT, OPT_ADD, OPT_PAD = 1, 2, 5, 17, 10, 11, 22
SAME, VALID = 0, 1
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 3


class _FB:
    """Flatbuffer writer.  A value is ('i8' | 'u8' | 'i32' | 'u32', number), ('str', text), ('bytes', b), ('ints', [..]),
    ('table', [field or None, ...]) or ('tables', [[fields], ...]); None = field absent."""

    def __init__(self):
        self.b = bytearray()

    def _align(self, n=4):
        while len(self.b) % n:
            self.b.append(0)

    def _write(self, val):
        """Writes one out-of-line object at the end of the buffer, returns its position."""
        kind, v = val
        self._align(4)
        if kind == "str":
            raw = v.encode()
            pos = len(self.b)
            self.b += struct.pack("<I", len(raw)) + raw + b"\0"
            return pos
        if kind == "bytes":
            self._align(16)
            # the length word sits right in front of 16-byte aligned data
            while (len(self.b) + 4) % 16:
                self.b.append(0)
            pos = len(self.b)
            self.b += struct.pack("<I", len(v)) + bytes(v)
            return pos
        if kind == "ints":
            pos = len(self.b)
            self.b += struct.pack("<I", len(v)) + b"".join(struct.pack("<i", int(x)) for x in v)
            return pos
        if kind == "tables":
            pos = len(self.b)
            self.b += struct.pack("<I", len(v)) + b"\0" * (4 * len(v))
            for i, fields in enumerate(v):
                slot = pos + 4 + 4 * i
                t = self._write(("table", fields))
                struct.pack_into("<I", self.b, slot, t - slot)
            return pos
        if kind == "table":
            sizes = {"i8": 1, "u8": 1, "i32": 4, "u32": 4}
            # vtable first (it may sit anywhere), then the table: soffset, inline fields
            offs, cur = [], 4
            for f in v:
                if f is None:
                    offs.append(0)
                    continue
                sz = sizes.get(f[0], 4)
                cur = (cur + sz - 1) // sz * sz
                offs.append(cur)
                cur += sz
            tsize = (cur + 3) // 4 * 4
            vt = len(self.b)
            self.b += struct.pack("<HH", 4 + 2 * len(v), tsize) + b"".join(struct.pack("<H", o) for o in offs)
            self._align(4)
            pos = len(self.b)
            self.b += b"\0" * tsize
            struct.pack_into("<i", self.b, pos, pos - vt)
            pending = []
            for f, o in zip(v, offs):
                if f is None:
                    continue
                if f[0] in sizes:
                    struct.pack_into({"i8": "<b", "u8": "<B", "i32": "<i", "u32": "<I"}[f[0]], self.b, pos + o, f[1])
                else:
                    pending.append((pos + o, f))
            for slot, f in pending:
                t = self._write(f)
                struct.pack_into("<I", self.b, slot, t - slot)
            return pos
        raise ValueError(kind)

    def finish(self, root_fields):
        self.b = bytearray(8)
        self.b[4:8] = b"TFL3"
        root = self._write(("table", root_fields))
        struct.pack_into("<I", self.b, 0, root)
        return bytes(self.b)


class GraphBuilder:
    """Builds a graph op by op; tensors are NHWC float32, constants come from a seeded generator."""

    def __init__(self, seed, in_shape):
        self.rng = np.random.default_rng(seed)
        self.tensors = []   # (shape, buffer index, name)
        self.buffers = [b""]
        self.ops = []       # (code, inputs, outputs, option tag, option fields)
        self.input = self._act(in_shape, "input")
        self.outputs = []

    # ---- tensors
    def _act(self, shape, name):
        self.tensors.append((list(shape), 0, name))
        return len(self.tensors) - 1

    def const(self, arr, name="c", dtype=0):
        a = np.ascontiguousarray(arr, dtype=np.float32 if dtype == 0 else np.int32)
        self.buffers.append(a.tobytes())
        self.tensors.append((list(a.shape), len(self.buffers) - 1, name, dtype))
        return len(self.tensors) - 1

    def shape(self, t):
        return self.tensors[t][0]

    def _w(self, shape, fan_in):
        return (self.rng.standard_normal(shape) * (0.9 / np.sqrt(fan_in))).astype(np.float32)

    # ---- operators (the reference graphs' chains)
    def conv(self, x, co, k=1, stride=1, padding=SAME, act=ACT_NONE):
        n, h, w, c = self.shape(x)
        wt = self.const(self._w((co, k, k, c), k * k * c), "w")
        bt = self.const(self.rng.standard_normal(co).astype(np.float32) * 0.1, "b")
        if padding == SAME:
            ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
        else:
            ho, wo = (h - k) // stride + 1, (w - k) // stride + 1
        y = self._act([n, ho, wo, co], "conv")
        self.ops.append((CONV_2D, [x, wt, bt], [y], OPT_CONV, [("i8", padding), ("i32", stride), ("i32", stride), ("i8", act)]))
        return y

    def dw(self, x, k=3, stride=1, padding=SAME):
        n, h, w, c = self.shape(x)
        wt = self.const(self._w((1, k, k, c), k * k), "wd")
        bt = self.const(self.rng.standard_normal(c).astype(np.float32) * 0.1, "bd")
        ho, wo = ((h + stride - 1) // stride, (w + stride - 1) // stride) if padding == SAME else ((h - k) // stride + 1, (w - k) // stride + 1)
        y = self._act([n, ho, wo, c], "dw")
        self.ops.append((DEPTHWISE_CONV_2D, [x, wt, bt], [y], OPT_DW, [("i8", padding), ("i32", stride), ("i32", stride), ("i32", 1), ("i8", ACT_NONE)]))
        return y

    def add(self, a, b):
        y = self._act(self.shape(a), "add")
        self.ops.append((ADD, [a, b], [y], OPT_ADD, [("i8", ACT_NONE)]))
        return y

    def relu(self, x):
        y = self._act(self.shape(x), "relu")
        self.ops.append((RELU, [x], [y], 0, None))
        return y

    def prelu(self, x):
        c = self.shape(x)[3]
        al = self.const((self.rng.random((1, 1, c)) * 0.5).astype(np.float32), "alpha")
        y = self._act(self.shape(x), "prelu")
        self.ops.append((PRELU, [x, al], [y], 0, None))
        return y

    def maxpool(self, x, k=2):
        n, h, w, c = self.shape(x)
        y = self._act([n, h // k, w // k, c], "pool")
        self.ops.append((MAX_POOL_2D, [x], [y], OPT_POOL, [("i8", VALID), ("i32", k), ("i32", k), ("i32", k), ("i32", k), ("i8", ACT_NONE)]))
        return y

    def pad_channels(self, x, extra):
        n, h, w, c = self.shape(x)
        p = self.const(np.array([[0, 0], [0, 0], [0, 0], [0, extra]]), "pads", dtype=2)
        y = self._act([n, h, w, c + extra], "pad")
        self.ops.append((PAD, [x, p], [y], OPT_PAD, []))
        return y

    def pad_spatial(self, x, top, bottom, left, right):
        n, h, w, c = self.shape(x)
        p = self.const(np.array([[0, 0], [top, bottom], [left, right], [0, 0]]), "pads", dtype=2)
        y = self._act([n, h + top + bottom, w + left + right, c], "pad")
        self.ops.append((PAD, [x, p], [y], OPT_PAD, []))
        return y

    def reshape(self, x, shape):
        s = self.const(np.array(shape), "shape", dtype=2)
        y = self._act(list(shape), "reshape")
        self.ops.append((RESHAPE, [x, s], [y], OPT_RESHAPE, [("ints", list(shape))]))
        return y

    def concat(self, xs, axis):
        shp = list(self.shape(xs[0]))
        shp[axis] = sum(self.shape(t)[axis] for t in xs)
        y = self._act(shp, "concat")
        self.ops.append((CONCATENATION, list(xs), [y], OPT_CONCAT, [("i32", axis), ("i8", ACT_NONE)]))
        return y

    # ---- the networks' building blocks
    def blaze_block(self, x, co=None, stride=1, act="relu"):
        """DW3x3 -> PW1x1 -> + skip (2x2 max-pool and zero channel pad when it down-samples / widens) -> activation."""
        c = self.shape(x)[3]
        co = co or c
        y = self.conv(self.dw(x, 3, stride), co)
        skip = x
        if stride == 2:
            skip = self.maxpool(skip)
        if co > c:
            skip = self.pad_channels(skip, co - c)
        y = self.add(y, skip)
        return self.relu(y) if act == "relu" else self.prelu(y)

    def bottleneck(self, x, cm):
        """iris: PW C -> cm + PReLU; DW3x3; PW cm -> C; + x; PReLU."""
        c = self.shape(x)[3]
        r = self.prelu(self.conv(x, cm))
        y = self.conv(self.dw(r), c)
        return self.prelu(self.add(y, x))

    def double_block(self, x, cm):
        """full_range: (DW3x3 -> PW C -> cm, ReLU), (DW3x3 -> PW cm -> C) + x, ReLU."""
        c = self.shape(x)[3]
        a = self.relu(self.conv(self.dw(x), cm))
        y = self.conv(self.dw(a), c)
        return self.relu(self.add(y, x))

    def double_block_widen(self, x, cm, co):
        """full_range where it widens without down-sampling: (DW3x3 -> PW C -> cm, ReLU), (DW3x3 -> PW cm -> co) + pad(x), ReLU."""
        c = self.shape(x)[3]
        a = self.relu(self.conv(self.dw(x), cm))
        y = self.conv(self.dw(a), co)
        return self.relu(self.add(y, self.pad_channels(x, co - c)))

    def finish(self):
        codes = sorted({op[0] for op in self.ops})
        fb = _FB()
        tensors = []
        for t in self.tensors:
            shape, buf, name = t[0], t[1], t[2]
            dtype = t[3] if len(t) > 3 else 0
            tensors.append([("ints", shape), ("i8", dtype), ("u32", buf), ("str", name)])
        ops = []
        for code, ins, outs, tag, opt in self.ops:
            fields = [("u32", codes.index(code)), ("ints", ins), ("ints", outs)]
            if opt is not None:
                fields += [("u8", tag), ("table", opt)]
            ops.append(fields)
        subgraph = [("tables", tensors), ("ints", [self.input]), ("ints", self.outputs), ("tables", ops), ("str", "main")]
        model = [("u32", 3),
                 ("tables", [[("i8", min(c, 127)), None, None, ("i32", c)] for c in codes]),
                 ("tables", [subgraph]),
                 ("str", "synthetic parity graph"),
                 ("tables", [[("bytes", b)] if b else [] for b in self.buffers])]
        return fb.finish(model)


def iris_like(seed, h, w, c=64, cm=32, pairs=2, down=True):
    """stem 3x3 s2 -> `pairs` bottlenecks at (h/2 x w/2 x c) [-> stride-2 block to 2c -> one bottleneck] -> 2x2-window head."""
    g = GraphBuilder(seed, [1, h, w, 3])
    x = g.prelu(g.conv(g.input, c, 3, 2))
    for _ in range(pairs):
        x = g.bottleneck(x, cm)
    if down:
        x = g.blaze_block(x, 2 * c, 2, act="prelu")
        x = g.bottleneck(x, cm * 2)
    g.outputs = [x]
    return g.finish()


def back_like(seed, size, c0=24, n1=3, n2=3, coarse_first=False):
    """stem 5x5 s2 -> n1 BlazeBlocks(c0) -> s2 block to 2 c0 -> n2 blocks -> s2 block to 4 c0 -> 3 blocks -> s2 block -> 2 blocks; SSD-style
    heads on the last two resolutions, reshaped and concatenated like the detectors' outputs."""
    g = GraphBuilder(seed, [1, size, size, 3])
    x = g.relu(g.conv(g.input, c0, 5, 2))
    for _ in range(n1):
        x = g.blaze_block(x)
    x = g.blaze_block(x, 2 * c0, 2)
    for _ in range(n2):
        x = g.blaze_block(x)
    x = g.blaze_block(x, 4 * c0, 2)
    for _ in range(3):
        x = g.blaze_block(x)
    a = x
    x = g.blaze_block(x, 4 * c0, 2)
    for _ in range(2):
        x = g.blaze_block(x)
    b = x
    regs, clss = [], []
    for t, anchors in (((b, 6), (a, 2)) if coarse_first else ((a, 2), (b, 6))):   # coarse_first: concatenated in the other order
        n, h, w, _ = g.shape(t)
        clss.append(g.reshape(g.conv(t, anchors), [1, h * w * anchors, 1]))
        regs.append(g.reshape(g.conv(t, anchors * 16), [1, h * w * anchors, 16]))
    g.outputs = [g.concat(regs, 1), g.concat(clss, 1)]
    return g.finish()


def front_like(seed, size, act="relu", stem_act="relu"):
    """The front / short-range detector's habit of growing by channel PADs on the skip, with widths that are not multiples of 4:
    24 -> 30 -> 42 at size/2, stride 2 to 54 (max-pool + PAD skip), 54 -> 58 -> 58, 1x1 head.  The lowering widens the 30 / 42 /
    54 / 58-channel tensors to 32 / 44 / 56 / 60 with zero channels (plan.cpp pad_odd_channels) so the fused kernels take them."""
    g = GraphBuilder(seed, [1, size, size, 3])
    x = g.conv(g.input, 24, 5, 2)
    x = g.relu(x) if stem_act == "relu" else (g.prelu(x) if stem_act == "prelu" else x)
    x = g.blaze_block(x, 30, act=act)
    x = g.blaze_block(x, 42, act=act)
    x = g.blaze_block(x, 54, 2, act=act)
    x = g.blaze_block(x, 58, act=act)
    x = g.blaze_block(x, act=act)
    n, h, w, _ = g.shape(x)
    g.outputs = [g.reshape(g.conv(x, 6), [1, h * w * 6, 1]), g.reshape(g.conv(x, 32), [1, h * w * 2, 16])]
    return g.finish()


def full_widen(seed, h, w, c=24, cm=8, co=40):
    """double blocks on a non-square frame whose row bands end ragged, one of them widening (skip zero-padded to co channels)."""
    g = GraphBuilder(seed, [1, h, w, 3])
    x = g.relu(g.conv(g.input, c, 3, 2))
    x = g.double_block(x, cm)
    x = g.double_block_widen(x, cm + 4, co)
    x = g.double_block(x, 12)
    g.outputs = [g.conv(x, 5)]
    return g.finish()


def full_like(seed, size, c=32, cm=12):
    """stem 3x3 s2 -> double blocks at (size/2)^2 x c -> stride-2 double block to 2c -> double blocks -> pointwise head."""
    g = GraphBuilder(seed, [1, size, size, 3])
    x = g.relu(g.conv(g.input, c, 3, 2))
    x = g.double_block(x, cm)
    x = g.double_block(x, cm)
    a = g.relu(g.conv(g.dw(x, 3, 2), cm + 4))
    y = g.conv(g.dw(a), 2 * c)
    x = g.relu(g.add(y, g.pad_channels(g.maxpool(x), c)))
    x = g.double_block(x, cm * 2)
    g.outputs = [g.conv(x, 6)]
    return g.finish()


def sparse_like(seed, h, w, c=32, cm=12):
    """full_range_sparse's down-sampling: the stride-2 depthwise stage reads an explicitly padded input (PAD of one pixel all round, VALID behind it:
    its window starts at row / column 2r - 1, not 2r as SAME's does), no skip around the pair."""
    g = GraphBuilder(seed, [1, h, w, 3])
    x = g.relu(g.conv(g.input, c, 3, 2))
    x = g.double_block(x, cm)
    a = g.relu(g.conv(g.dw(g.pad_spatial(x, 1, 1, 1, 1), 3, 2, VALID), cm + 4))
    x = g.relu(g.conv(g.dw(a), 2 * c))
    x = g.double_block(x, cm * 2)
    a = g.relu(g.conv(g.dw(g.pad_spatial(x, 1, 1, 1, 1), 3, 2, VALID), 2 * cm))
    x = g.relu(g.conv(g.dw(a), 3 * c))
    g.outputs = [g.conv(x, 6)]
    return g.finish()


def full_tail(seed, h, w, cn=32, cw=128, cp=64, pairs=2, pointwise_last=True):
    """full_range's coarsest resolution: a narrow tensor from a stride-2 block, an expand block whose skip is the max-pooled, channel-padded
    previous resolution, `pairs` double blocks (wide -> narrow -> wide, skip around both), a contract block (pointwise only, as in the
    reference graph, or with its depthwise stage) — the run xc_kernels.hip takes as one launch when the frame has <= 64 pixels."""
    g = GraphBuilder(seed, [1, h, w, 3])
    p = g.relu(g.conv(g.input, cp, 3, 2))                      # previous resolution: (h/2) x (w/2) x cp
    x = g.relu(g.conv(g.dw(p, 3, 2), cn))                      # narrow, (h/4) x (w/4) x cn
    y = g.conv(g.dw(x), cw)
    x = g.relu(g.add(y, g.pad_channels(g.maxpool(p), cw - cp)))
    for _ in range(pairs):
        x = g.double_block(x, cn)
    x = g.relu(g.conv(x if pointwise_last else g.dw(x), cn))
    g.outputs = [g.conv(x, 7)]
    return g.finish()


def mesh_like(seed, size, c0=16, act="prelu"):
    """face-mesh style: stem 3x3 s2 + PReLU -> 2 blocks(c0) -> s2 to 2c0 -> 2 blocks -> s2 to 4c0 -> 2 blocks -> s2 to 8c0 -> 3 blocks -> s2 -> 2 blocks;
    two branches of small-spatial work ending in whole-frame convolutions (the GEMM heads)."""
    g = GraphBuilder(seed, [1, size, size, 3])
    x = g.conv(g.input, c0, 3, 2)
    x = g.prelu(x) if act == "prelu" else g.relu(x)
    for mult, nb in ((1, 2), (2, 2), (4, 2), (8, 3)):
        if mult > 1:
            x = g.blaze_block(x, mult * c0, 2, act=act)
        for _ in range(nb):
            x = g.blaze_block(x, act=act)
    x = g.blaze_block(x, 8 * c0, 2, act=act)
    for _ in range(2):
        x = g.blaze_block(x, act=act)
    n, h, w, c = g.shape(x)
    a = g.conv(x, 32)
    a = g.prelu(a) if act == "prelu" else g.relu(a)
    a = g.blaze_block(a, act=act)
    out_a = g.conv(a, 47, h, 1, VALID)          # window = frame: [1,1,1,47]
    b = g.blaze_block(x, act=act)
    b = g.conv(b, 32)
    b = g.prelu(b) if act == "prelu" else g.relu(b)
    out_b = g.conv(b, 1, h, 1, VALID)
    g.outputs = [out_a, out_b]
    return g.finish()


CASES = {
    # name: (builder, input H, input W): what it aims at
    "iris_32x20_ragged_bands": (lambda: iris_like(11, 40, 64, 64, 32, 2), 40, 64),        # 20 x 32 frame: bands of 8, 8, 4 rows
    "iris_16x16_c64_run": (lambda: iris_like(12, 32, 32, 64, 32, 3), 32, 32),             # 64-channel frame-resident run of 3 pairs
    "iris_12x20_c128": (lambda: iris_like(13, 24, 40, 128, 64, 2, down=False), 24, 40),   # 240 pixels: partial last pixel group
    "iris_24x24_fallback": (lambda: iris_like(14, 48, 48, 64, 32, 1), 48, 48),            # band rows do not tile 32-pixel groups: stage programs
    # round 5 (tail_kernels.hip): 256-channel stages = 16 output tiles (two per wave) and a 1x1 contraction over 256 values (two k-blocks
    # streamed through the A registers), on 6 x 10 frames (pixel tiles that straddle frames when several frames share a workgroup)
    "iris_6x10_c256_tail": (lambda: iris_like(15, 12, 20, 256, 64, 2, down=False), 12, 20),
    "iris_4x4_c128_down_tail": (lambda: iris_like(16, 8, 8, 128, 32, 1), 8, 8),           # 4 x 4 x 128 -> stride-2 block to 2 x 2 x 256 -> bottleneck 256 <-> 64
    "back_96": (lambda: back_like(21, 96), 96, 96),                                        # chains at 12x12x96 / 6x6x96 with edges and heads
    "back_160_c16": (lambda: back_like(22, 160, 16, 2, 4), 160, 160),                      # 16-channel pipelines, 20x20 / 10x10 chains
    "back_48_odd_head_slices": (lambda: back_like(24, 48, coarse_first=True), 48, 48),     # 3x3 heads first: the 6x6 classifier's slice of the concatenation starts at float 54 (not 16-byte aligned: stored float by float by the chain's fused heads)
    "front_64_odd_widths": (lambda: front_like(25, 64), 64, 64),                            # 30 / 42 / 54 / 58 channels: zero-padded to multiples of 4 at lowering
    "front_48_odd_widths_prelu": (lambda: front_like(26, 48, "prelu"), 48, 48),            # the same with PReLU slopes to pad
    # round 6: a 5x5 first convolution of 64 output pixels per row (stem_mfma_kernel takes it) with a PReLU / with no activation behind it — the shipped detectors' is ReLU
    "front_128_prelu_stem": (lambda: front_like(27, 128, "prelu", "prelu"), 128, 128),
    "front_128_linear_stem": (lambda: front_like(28, 128, "relu", "none"), 128, 128),
    "back_128_c32": (lambda: back_like(23, 128, 32, 2, 2), 128, 128),                      # 32 / 64 / 128 channels: strip kernel, 4-tile chains with heads
    "mesh_160": (lambda: mesh_like(41, 160), 160, 160),                                    # 80x80x16 pipelines ... 5x5x128 chains, stage programs, GEMM heads
    "mesh_96_c24": (lambda: mesh_like(42, 96, 24), 96, 96),
    "mesh_192_relu": (lambda: mesh_like(43, 192, 16, "relu"), 192, 192),                   # round 6: the face mesh's shapes with ReLU everywhere (mdblock_kernel<stem+pair>'s ReLU instantiation, from 32 frames on)                                # 24-channel PReLU pipelines, 3x3 frames
    "full_widen_70x44": (lambda: full_widen(33, 70, 44), 70, 44),                           # 35 x 22 frames: ragged bands, widening double block
    # (wide tensors too large for the stage programs' depthwise scratch: these runs go to xc_kernel)
    "full_tail_6x6": (lambda: full_tail(34, 24, 24, 96, 384, 256, 2), 24, 24),                  # 6 x 6: expand (max-pool skip) + 2 double blocks + pointwise contract, full_range's widths
    "full_tail_8x8_dw_last": (lambda: full_tail(35, 32, 32, 64, 224, 40, 3, False), 32, 32),   # 8 x 8 = two full pixel groups, 64 <-> 224 channels (7 tiles), depthwise contract at the end
    "full_tail_5x7": (lambda: full_tail(36, 20, 28, 48, 328, 24, 1), 20, 28),                  # 35 pixels (partial second group), 328 wide channels = 10.25 tiles, 6 k-chunks per stage-1 wave
    "full_64": (lambda: full_like(31, 64), 64, 64),                                        # double blocks, odd middle widths
    "full_80_c48": (lambda: full_like(32, 80, 48, 20), 80, 80),
    # round 6: explicitly padded stride-2 blocks (full_range_sparse), one-row bands and — 256 rows on 128 workgroups — two-row bands whose LAST row travels
    "sparse_64": (lambda: sparse_like(37, 64, 64), 64, 64),
    "sparse_512x48_two_row_bands": (lambda: sparse_like(38, 512, 48, 24, 8), 512, 48),
}
