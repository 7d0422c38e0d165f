"""CPU-side checks of the product library: it loads, exports every symbol include/mi_face.h declares, lowers every
shipped graph without a GPU, and refuses to run without one (no silent CPU fallback)."""
import ctypes
import os
import re

import numpy as np

import pytest

from conftest import MODEL_FILES, ROOT, model_path


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "mi_face.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(mi):
    L = ctypes.CDLL(mi.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), "libmiface.so does not export %s" % n
    assert set(mi.EXPORTS) == set(names)


@pytest.mark.parametrize("name", list(MODEL_FILES))
def test_lowering_without_gpu(mi, name):
    blob = open(model_path(name), "rb").read()
    p0, p1, p2 = (mi.plan_describe(blob, lvl) for lvl in (0, 1, 2))
    p3 = mi.plan_describe(blob, 3)
    assert int(re.search(r"launches=(\d+)", p3).group(1)) <= int(re.search(r"launches=(\d+)", p2).group(1))
    p4 = mi.plan_describe(blob, 4)
    assert int(re.search(r"launches=(\d+)", p4).group(1)) <= int(re.search(r"launches=(\d+)", p3).group(1))
    n0, n1, n2 = (int(re.search(r"launches=(\d+)", p).group(1)) for p in (p0, p1, p2))
    assert n0 > n1 > n2
    assert "block" in p2 and "block" not in p1
    macs = [int(re.search(r"macs_per_frame=(\d+)", p).group(1)) for p in (p0, p1, p2)]
    assert macs[0] == macs[1] == macs[2]


@pytest.mark.parametrize("name", list(MODEL_FILES))
def test_stage_program_lowering_covers_every_operator_once(mi, name):
    """Level 5 reorders independent branches and regroups nodes into frame-resident stage programs: every .tflite operator
    must still be executed exactly once, the arithmetic (MACs) must not change, and the launch count must not grow."""
    blob = open(model_path(name), "rb").read()
    p4, p5 = mi.plan_describe(blob, 4), mi.plan_describe(blob, 5)
    ops = lambda p: sorted(int(x) for grp in re.findall(r"ops\{([0-9,]*)\}", p) for x in grp.split(",") if x)
    assert ops(p4) == ops(p5)
    assert len(set(ops(p5))) == len(ops(p5))
    get = lambda p, key: int(re.search(key + r"=(\d+)", p).group(1))
    assert get(p5, "macs_per_frame") == get(p4, "macs_per_frame")
    assert get(p5, "launches") <= get(p4, "launches")
    for lds in re.findall(r"frame resident, (\d+) B LDS", p5):
        assert int(lds) <= 160 * 1024
    if name in ("iris", "landmark"):
        assert get(p5, "launches") + 6 <= get(p4, "launches")   # the small-spatial tails collapse (the whole-frame heads are GEMM launches of their own)
        assert "resident" in p5


def test_back_plan_matches_survey_numbers(mi):
    p = mi.plan_describe(open(model_path("back"), "rb").read(), 0)
    assert "macs_per_frame=188749824" in p        # SURVEY.md §8d, back 256^2


def test_malformed_model_is_an_error_not_a_crash(mi):
    with pytest.raises(mi.MiError):
        mi.plan_describe(b"\x00" * 64, 2)
    blob = open(model_path("front"), "rb").read()
    with pytest.raises(mi.MiError):
        mi.plan_describe(blob[: len(blob) // 3], 2)


def test_no_cpu_fallback(mi):
    if mi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mi.MiError) as e:
        mi.FaceDetection(mi.FaceDetectionModel.BackCamera)
    assert e.value.code == -4


def test_batched_u8_entry_points_refuse_bad_arguments_without_gpu(mi):
    """mi_fd_infer_images / mi_fd_submit_images / mi_fd_collect / mi_host_alloc: null handles and pointers are MI_EINVAL with a message,
    never a crash — checked before any device work, so it runs without a GPU."""
    import ctypes as C
    L = mi.lib()
    buf = (C.c_uint8 * 64)()
    out = (C.c_float * 17)()
    cnt = C.c_int()
    assert L.mi_fd_infer_images(None, buf, 1, 2, 2, 6, None, out, 1, C.byref(cnt), 0, None) == -1
    assert b"null" in L.mi_last_error()
    assert L.mi_fd_submit_images(None, 0, buf, 1, 2, 2, 6, 1) == -1
    assert L.mi_fd_collect(None, 0, out, C.byref(cnt)) == -1
    p = C.c_void_p()
    assert L.mi_host_alloc(0, C.byref(p)) == -1 and not p.value
    assert L.mi_host_alloc(16, None) == -1
    L.mi_host_free(None)            # a no-op
    # round 5: the batched u8 entries of the mesh and the iris network
    lm = (C.c_float * (3 * 468))()
    present = C.c_int()
    assert L.mi_fl_infer_images(None, buf, 1, 2, 2, 6, None, 1, lm, C.byref(present), None, 0, None) == -1
    assert b"null" in L.mi_last_error()
    assert L.mi_iris_infer_images(None, buf, 1, 2, 2, 6, None, None, 1, lm, lm, 0, None) == -1
    assert b"null" in L.mi_last_error()


def test_header_is_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: include/mi_face.h must compile as C99 (no C++-isms), alone and from a C translation unit
    that takes the address of every declared function with its declared type."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "mi_face.h")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    src = tmp_path / "use.c"
    names = _declared_functions()
    src.write_text('#include "mi_face.h"\n#include <stdio.h>\nint main(void) {\n  const void *f[] = {%s};\n'
                   '  mi_detection d; mi_rect r; mi_landmark l;\n'
                   '  printf("%%d %%d %%d %%d\\n", (int)(sizeof f / sizeof f[0]), (int)sizeof d, (int)sizeof r, (int)sizeof l);\n  return 0;\n}\n'
                   % ", ".join("(const void *)%s" % n for n in names))
    exe = str(tmp_path / "use")
    lib = os.path.dirname(mi_lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe, "-L", lib,
                           "-lmiface", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    n, sd, sr, sl = (int(v) for v in out.stdout.split())
    assert n == len(names) and (sd, sr, sl) == (68, 48, 24)      # types.rs layouts: 17 f32; 5 f64 + flag (padded); 3 f64


def mi_lib_path():
    import rs_face_detection_tflite_amd as m
    return m.LIB_PATH


def _c_prototypes():
    """name -> number of parameters, from include/mi_face.h"""
    src = open(os.path.join(ROOT, "include", "mi_face.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(mi_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return protos


def test_rust_shim_declarations_match_the_header():
    """bindings/rust is source only (no rustc in the image).  What can be checked without a compiler: every `extern "C"` item
    of src/ffi.rs names a function the header declares, with the same number of parameters, and the #[repr(C)] structs list
    the header's fields in the header's order."""
    protos = _c_prototypes()
    ffi = open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read()
    block = ffi[ffi.index('extern "C" {'):]
    decls = re.findall(r"pub fn (mi_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S)
    assert len(decls) >= 24
    for name, args in decls:
        assert name in protos, "ffi.rs declares %s, which mi_face.h does not" % name
        n = len([a for a in args.split(",") if ":" in a])
        assert n == protos[name], "%s: %d parameters in ffi.rs, %d in mi_face.h" % (name, n, protos[name])
    for needed in ("mi_fd_create", "mi_fd_infer_image", "mi_fl_create", "mi_fl_infer_image", "mi_iris_create", "mi_iris_infer_image",
                   "mi_face_detection_to_roi", "mi_iris_roi_from_face_landmarks", "mi_last_error"):
        assert needed in dict(decls)
    fields = lambda body: re.findall(r"pub (\w+):", body)
    struct = lambda name: ffi[ffi.index("pub struct %s {" % name):].split("}")[0]
    assert fields(struct("mi_detection")) == ["data", "score"]
    assert fields(struct("mi_rect")) == ["x_center", "y_center", "width", "height", "rotation", "normalized"]
    assert fields(struct("mi_landmark")) == ["x", "y", "z"]
    # every module lib.rs names exists, and braces / parentheses balance in every source file
    src_dir = os.path.join(ROOT, "bindings", "rust", "src")
    lib_rs = open(os.path.join(src_dir, "lib.rs")).read()
    for mod in re.findall(r"pub mod (\w+);", lib_rs):
        assert os.path.exists(os.path.join(src_dir, mod + ".rs")), mod
    for f in os.listdir(src_dir):
        text = re.sub(r"//.*", "", open(os.path.join(src_dir, f)).read())
        text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', text)
        for a, b in ("{}", "()", "[]"):
            assert text.count(a) == text.count(b), (f, a)


def test_absent_operand_is_refused_not_dereferenced(mi):
    """-1 ("optional input absent") is legal only in a convolution's bias slot.  A model with -1 as an activation,
    filter, skip, slope or paddings operand must end in MI_EMODEL from the parser — the lowering indexes g.tensors[operand]."""
    import synth_tflite as st

    def graph(mutate=None):
        gb = st.GraphBuilder(3, [1, 16, 16, 8])
        x = gb.blaze_block(gb.input, 8)
        x = gb.prelu(gb.conv(x, 8))
        x = gb.blaze_block(x, 16, stride=2)       # max-pool + channel PAD on the skip
        gb.outputs = [x]
        ops = list(gb.ops)
        if mutate:
            mutate(gb.ops)
        return gb.finish(), ops

    blob, ops = graph()
    mi.plan_describe(blob, 5)                     # the unmodified graph lowers
    refused = accepted = 0
    for i, op in enumerate(ops):
        for slot in range(len(op[1])):
            def mutate(ops_, i=i, slot=slot):
                code, ins, outs, tag, opt = ops_[i]
                ops_[i] = (code, [(-1 if k == slot else t) for k, t in enumerate(ins)], outs, tag, opt)
            bias_slot = slot == 2 and op[0] in (st.CONV_2D, st.DEPTHWISE_CONV_2D)
            try:
                mi.plan_describe(graph(mutate)[0], 5)
                assert bias_slot, "operand %d of op %d (builtin %d) absent, yet the model was accepted" % (slot, i, op[0])
                accepted += 1
            except mi.MiError:
                assert not bias_slot
                refused += 1
    assert refused >= len(ops) and accepted >= 4


def test_pad_with_constant_input_and_shrinking_output_terminates(tmp_path):
    """the widening pass (plan.cpp pad_odd_channels) checks its channel PADs to a fixpoint; a PAD whose DATA input is a
    constant tensor can never be marked (constants are no activation classes), and when its output is narrower than its input the
    loop used to report a change on every pass — a crafted model hung mi_*_create_from_bytes.  The lowering runs in a child process
    under a timeout: it must come back (accepting or refusing the graph), with or without an odd-width neighbour that makes the pass
    do real work."""
    import subprocess, sys
    import synth_tflite as st

    for variant in range(3):
        gb = st.GraphBuilder(7, [1, 8, 8, 8])
        x = gb.conv(gb.input, 6 if variant else 8)            # 6 channels: the pass has something to widen
        x = gb.relu(x)
        c = gb.const(np.zeros((1, 8, 8, 12), np.float32), "const_in")
        pads = gb.const(np.array([[0, 0], [0, 0], [0, 0], [0, -4 if variant < 2 else 0]]), "pads", dtype=2)
        y = gb._act([1, 8, 8, 8 if variant < 2 else 12], "pad_of_const")
        gb.ops.append((st.PAD, [c, pads], [y], st.OPT_PAD, []))
        x8 = gb.conv(x, y and gb.shape(y)[3])
        z = gb.add(x8, y)
        gb.outputs = [gb.relu(z)]
        path = tmp_path / ("pad_const_%d.tflite" % variant)
        path.write_bytes(gb.finish())
        code = ("import sys; sys.path.insert(0, %r); import rs_face_detection_tflite_amd as mi\n"
                "try:\n    mi.plan_describe(open(%r, 'rb').read(), 5); print('accepted')\n"
                "except mi.MiError as e:\n    print('refused')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(path)))
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
        assert p.returncode == 0 and p.stdout.strip() in ("accepted", "refused"), (variant, p.returncode, p.stdout, p.stderr[-400:])
