#!/usr/bin/env python3
"""Build an A/B variant of libmarius_hip.so with extra -D flags into build_abl/<name>/ (git-ignored, travels with gpurun):

    python tools/build_variant.py seg16 -DMARIUS_SEG_BATCH=16
    LD_LIBRARY_PATH=$PWD/build_abl/seg16 MARIUS_HIP_LIB=$PWD/build_abl/seg16/libmarius_hip.so python bench.py ...

(the host module finds libmarius_hip.so through LD_LIBRARY_PATH before its RUNPATH, the ctypes view through MARIUS_HIP_LIB; same C-ABI)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, defs = sys.argv[1], sys.argv[2:]
kdir = os.path.join(ROOT, "marius_amd", "csrc", "kernels")
out = os.path.join(ROOT, "build_abl", name)
os.makedirs(out, exist_ok=True)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + kdir] + defs
procs, objs = [], []
for f in sorted(os.listdir(kdir)):
    if f.endswith(".hip"):
        o = os.path.join(out, f[:-4] + ".o")
        objs.append(o)
        # only the files that mention one of the macros need a rebuild; the others are taken from the default build
        text = open(os.path.join(kdir, f)).read()
        macros = [d[2:].split("=")[0] for d in defs if d.startswith("-D")]
        default_o = os.path.join(ROOT, "marius_amd", "lib", "obj", f[:-4] + ".o")
        if not any(m in text for m in macros) and os.path.exists(default_o):
            objs[-1] = default_o
            continue
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(kdir, f), "-o", o]))
for p in procs:
    if p.wait() != 0:
        sys.exit("compile failed")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libmarius_hip.so")] + objs)
print("built", os.path.join(out, "libmarius_hip.so"))
