"""Build libvspbfr_hip with conv_bf16_rv.hip's DEVICE code taken from a (hand-edited) assembly file -- the bisection of the packed-fp32
miscompare (review r5 item 4b): the SLP-vectorised assembly of the kernel is the failing program; variants of it that differ in a
few instructions are assembled, linked and bundled exactly as hipcc does it (device .s -> .o -> lld -> offload bundle -> host object) and
linked with the other objects of the production build into build/rvasm/<name>.so.

  rv_asm_build.py emit <out.s> [extra hipcc flags]     the device assembly of conv_bf16_rv.hip (SLP on unless -fno-slp-vectorize is given)
  rv_asm_build.py link <in.s> <name>                   build/rvasm/<name>.so with that device code
"""
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vspbfr_amd", "csrc", "conv_bf16_rv.hip")
OUT = os.path.join(ROOT, "build", "rvasm")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-DVSP_BUILT_WITHOUT_SLP"]
os.makedirs(OUT, exist_ok=True)


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(" ".join(cmd)[:300] + "\n" + r.stderr[-3000:])
        raise SystemExit(r.returncode)
    return r


def emit(out, extra):
    run([HIPCC] + FLAGS + extra + ["-S", "--cuda-device-only", SRC, "-o", out])


def link(asm, name):
    r = subprocess.run([HIPCC] + FLAGS + ["-c", SRC, "-o", os.path.join(OUT, name + ".o"), "-###"], capture_output=True, text=True)
    cmds = [shlex.split(l.strip()) for l in r.stderr.splitlines() if l.strip().startswith('"')]
    assert len(cmds) == 4, len(cmds)
    dev_cc1, lld, bundler, host_cc1 = cmds
    dev_o, dev_out, fb = (os.path.join(OUT, name + s) for s in (".dev.o", ".dev.out", ".hipfb"))
    clang = dev_cc1[0]
    run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", dev_o])
    old_o = [a for a in lld if a.startswith("/tmp/") and a.endswith(".o")][0]
    old_out = [a for a in lld if a.startswith("/tmp/") and a.endswith(".out")][0]
    run([dev_o if a == old_o else dev_out if a == old_out else a for a in lld])
    old_fb = [a for a in host_cc1 if a.startswith("/tmp/") and a.endswith(".hipfb")][0]
    run([("-input=" + dev_out) if a == "-input=" + old_out else ("-output=" + fb) if a == "-output=" + old_fb else a for a in bundler])
    run([fb if a == old_fb else a for a in host_cc1])
    objs = [os.path.join(ROOT, "build", "csrc", f) for f in sorted(os.listdir(os.path.join(ROOT, "build", "csrc")))
            if f.endswith(".o") and f != "conv_bf16_rv.o"]
    so = os.path.join(OUT, name + ".so")
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [os.path.join(OUT, name + ".o")])
    print(so)


if __name__ == "__main__":
    if sys.argv[1] == "emit":
        emit(sys.argv[2], sys.argv[3:])
    else:
        link(sys.argv[2], sys.argv[3])
