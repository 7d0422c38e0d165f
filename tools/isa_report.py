#!/usr/bin/env python3
"""Per-kernel register / spill / scratch report from the gfx950 ISA of one .hip source (cross-compiles, no GPU needed).

usage: tools/isa_report.py csrc/strip_kernels.hip [substring] [--asm out.s] [-- extra hipcc flags]
Prints one line per kernel: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes, LDS bytes.
Used by tests/test_isa_checks.py (kernels that must not spill) and by hand while tuning."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rs-face-detection-tflite_amd", "csrc")


def compile_asm(src, out, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-S", "--cuda-device-only", "-x", "hip", src, "-o", out, *extra]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n") if p.returncode == 0 else names


def kernels(asm_path):
    """[{name, vgpr, agpr, sgpr, vspill, sspill, scratch, lds}] from the .amdgpu_metadata block."""
    txt = open(asm_path).read()
    meta = txt[txt.index("amdhsa.kernels:"):]
    out = []
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        out.append(dict(name=g("name").group(1), vgpr=int(g("vgpr_count").group(1)), agpr=int(g("agpr_count").group(1)),
                        sgpr=int(g("sgpr_count").group(1)), vspill=int(g("vgpr_spill_count").group(1)),
                        sspill=int(g("sgpr_spill_count").group(1)), scratch=int(g("private_segment_fixed_size").group(1)),
                        lds=int(g("group_segment_fixed_size").group(1))))
    for k, d in zip(out, demangle([k["name"] for k in out])):
        k["pretty"] = re.sub(r"mi::\(anonymous namespace\)::|void |\(.*\)$", "", d)
        # instructions that touch the private segment inside this kernel's text (a reserved segment nobody accesses costs nothing at run time)
        m = re.search(r"^%s:\n(.*?)^\s*s_endpgm" % re.escape(k["name"]), txt, flags=re.S | re.M)
        body = m.group(1) if m else ""
        k["scratch_ops"] = len(re.findall(r"^\s*(?:scratch_(?:load|store)\w*|buffer_(?:load|store)\w* [^\n]*\bs\[0:3\])", body, flags=re.M))
    return out


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    asm = None
    if "--asm" in args:
        i = args.index("--asm"); asm = args[i + 1]; del args[i:i + 2]
    src = args[0] if os.path.exists(args[0]) else os.path.join(CSRC, args[0])
    sub = args[1] if len(args) > 1 else ""
    tmp = asm or tempfile.mktemp(suffix=".s")
    compile_asm(src, tmp, extra)
    for k in kernels(tmp):
        if sub in k["pretty"]:
            print("%-58s vgpr %3d agpr %3d sgpr %3d  spill v %3d s %3d  scratch %4d  lds %6d" %
                  (k["pretty"], k["vgpr"], k["agpr"], k["sgpr"], k["vspill"], k["sspill"], k["scratch"], k["lds"]))
    if not asm: os.unlink(tmp)


if __name__ == "__main__":
    main()
