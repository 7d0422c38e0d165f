"""TEST INFRASTRUCTURE — second, independent CPU evaluation of the frozen MediaPipe graphs (torch-CPU).

Used only to cross-check the C oracle (oracle/c/interp.c) before goldens are frozen (SURVEY.md §8c:
"two independent evaluations must agree").  Arithmetic is delegated to torch's CPU convolution (oneDNN), i.e. a
different code path and a different summation order from the C oracle's plain loops.

What it stands in for: the third-party TensorFlow-Lite runtime behind the `tflite` crate 0.9.8
(`interpreter.invoke()`, /root/reference/src/face_detection_lite/face_detection.rs:235, face_landmark.rs:265,
iris_landmark.rs:203).  The runtime's source is not under /root/reference; op semantics follow the published TFLite
builtin float kernels.

Pin: the reference's tests hold no numeric assertions for this path; the C oracle this module cross-checks is
pinned by the reference's rendered PNGs on man.jpg, pixel for pixel (tests/test_pins.py, oracle/c/oracle.h).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import tfl3


def same_pads(in_size, k, stride):
    """TF SAME: out = ceil(in/stride); total = max(0,(out-1)*stride + k - in); before = total//2 (Appendix C.1)."""
    out = -(-in_size // stride)
    total = max(0, (out - 1) * stride + k - in_size)
    return total // 2, total - total // 2


def _act(x, code):
    if code == 0:
        return x
    if code == 1:
        return torch.relu(x)
    if code == 3:
        return torch.clamp(x, 0.0, 6.0)
    raise NotImplementedError("fused activation %d" % code)


def densify(t: tfl3.Tensor) -> np.ndarray:
    """TFLite sparse tensor → dense (format converter, published algorithm: traversal order over
    (possibly block-expanded) dims, dim_metadata per traversal level: dense size or CSR segments/indices)."""
    sp = t.sparsity
    shape = list(t.shape)
    order = sp["traversal_order"]
    block_map = sp["block_map"]
    dims = sp["dims"]
    nd = len(shape)
    # expanded shape: original dims (block-reduced) followed by block dims
    block_size = [d["dense_size"] for d in dims[nd:]] if len(dims) > nd else []
    values = np.asarray(t.data)
    dense = np.zeros(shape, dtype=values.dtype)
    lvl_n = len(order)
    idx = [0] * lvl_n
    pos = [0]

    def emit(vidx):
        # map expanded coordinates (in traversal order) back to original coordinates
        coord = [0] * lvl_n
        for lv in range(lvl_n):
            coord[order[lv]] = idx[lv]
        orig = coord[:nd]
        for bi, od in enumerate(block_map):
            orig[od] = orig[od] * block_size[bi] + coord[nd + bi]
        dense[tuple(orig)] = values[vidx]

    def rec(level, prev):
        if level == lvl_n:
            emit(prev)
            return
        d = dims[level]
        if d["format"] == 0:
            n = d["dense_size"]
            for i in range(n):
                idx[level] = i
                rec(level + 1, prev * n + i)
        else:
            seg, ind = d["segments"], d["indices"]
            for p in range(int(seg[prev]), int(seg[prev + 1])):
                idx[level] = int(ind[p])
                rec(level + 1, p)

    rec(0, 0)
    return dense


def run(g: tfl3.Graph, x: np.ndarray, keep_all=False, num_threads=None):
    """x: float32 [B,H,W,C] NHWC. Returns list of output arrays (graph output order) or dict of all tensors."""
    if num_threads:
        torch.set_num_threads(num_threads)
    vals = {}

    def const(i):
        t = g.tensors[i]
        a = np.asarray(t.data).reshape(t.shape if t.shape else ())
        return a

    def get(i):
        if i in vals:
            return vals[i]
        t = g.tensors[i]
        if t.data is None:
            raise KeyError("tensor %d has no value" % i)
        a = const(i)
        if a.dtype == np.float16:
            a = a.astype(np.float32)
        v = torch.from_numpy(np.ascontiguousarray(a))
        vals[i] = v
        return v

    vals[g.inputs[0]] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    B = x.shape[0]
    with torch.no_grad():
        for op in g.ops:
            c = op.code
            o = op.outputs[0]
            if c == tfl3.DEQUANTIZE:
                src = vals[op.inputs[0]].numpy() if op.inputs[0] in vals else const(op.inputs[0])
                vals[o] = torch.from_numpy(np.ascontiguousarray(src).astype(np.float32))
            elif c == tfl3.DENSIFY:
                vals[o] = torch.from_numpy(densify(g.tensors[op.inputs[0]]))
            elif c in (tfl3.CONV_2D, tfl3.DEPTHWISE_CONV_2D):
                a = get(op.inputs[0]).permute(0, 3, 1, 2)
                w = get(op.inputs[1]).to(torch.float32)
                bias = get(op.inputs[2]).to(torch.float32) if len(op.inputs) > 2 and op.inputs[2] >= 0 else None
                sh, sw = op.opts["stride_h"], op.opts["stride_w"]
                if c == tfl3.CONV_2D:
                    kh, kw = w.shape[1], w.shape[2]
                    wt = w.permute(0, 3, 1, 2).contiguous()
                    groups = 1
                else:
                    kh, kw = w.shape[1], w.shape[2]
                    wt = w.permute(3, 0, 1, 2).contiguous()
                    groups = wt.shape[0]
                    assert op.opts["depth_multiplier"] == 1
                if op.opts["padding"] == 0:
                    pt, pb = same_pads(a.shape[2], kh, sh)
                    pl, pr = same_pads(a.shape[3], kw, sw)
                    a = F.pad(a, (pl, pr, pt, pb))
                y = F.conv2d(a, wt, bias, stride=(sh, sw), groups=groups)
                vals[o] = _act(y, op.opts["act"]).permute(0, 2, 3, 1).contiguous()
            elif c == tfl3.ADD:
                vals[o] = _act(get(op.inputs[0]) + get(op.inputs[1]), op.opts.get("act", 0))
            elif c == tfl3.RELU:
                vals[o] = torch.relu(get(op.inputs[0]))
            elif c == tfl3.PRELU:
                a = get(op.inputs[0])
                alpha = get(op.inputs[1])
                vals[o] = torch.where(a >= 0, a, a * alpha)
            elif c == tfl3.MAX_POOL_2D:
                a = get(op.inputs[0]).permute(0, 3, 1, 2)
                fh, fw = op.opts["filter_h"], op.opts["filter_w"]
                sh, sw = op.opts["stride_h"], op.opts["stride_w"]
                if op.opts["padding"] == 0:
                    pt, pb = same_pads(a.shape[2], fh, sh)
                    pl, pr = same_pads(a.shape[3], fw, sw)
                    a = F.pad(a, (pl, pr, pt, pb), value=float("-inf"))
                vals[o] = F.max_pool2d(a, (fh, fw), (sh, sw)).permute(0, 2, 3, 1).contiguous()
            elif c == tfl3.PAD:
                a = get(op.inputs[0])
                p = const(op.inputs[1]).reshape(-1, 2)
                flat = []
                for d in range(a.dim() - 1, -1, -1):
                    flat += [int(p[d, 0]), int(p[d, 1])]
                vals[o] = F.pad(a, flat)
            elif c == tfl3.RESHAPE:
                a = get(op.inputs[0])
                ns = list(g.tensors[o].shape)
                ns[0] = B
                vals[o] = a.reshape(ns)
            elif c == tfl3.CONCATENATION:
                vals[o] = torch.cat([get(i) for i in op.inputs], dim=op.opts["axis"])
            elif c == tfl3.RESIZE_BILINEAR:
                a = get(op.inputs[0]).permute(0, 3, 1, 2)
                size = [int(v) for v in const(op.inputs[1]).reshape(-1)]
                assert op.opts["half_pixel_centers"] and not op.opts["align_corners"]
                vals[o] = F.interpolate(a, size=size, mode="bilinear", align_corners=False).permute(
                    0, 2, 3, 1).contiguous()
            elif c == tfl3.DEPTH_TO_SPACE:
                a = get(op.inputs[0])
                bs = op.opts["block_size"]
                b_, h, w, ch = a.shape
                a = a.reshape(b_, h, w, bs, bs, ch // (bs * bs)).permute(0, 1, 3, 2, 4, 5)
                vals[o] = a.reshape(b_, h * bs, w * bs, ch // (bs * bs)).contiguous()
            else:
                raise NotImplementedError(op.name)
    if keep_all:
        return {k: v.numpy() for k, v in vals.items()}
    return [vals[i].numpy() for i in g.outputs]
