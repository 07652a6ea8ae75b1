"""TEST INFRASTRUCTURE: builds tests/emul/_build/libneighbor_emul.so = marius_amd/csrc/kernels/neighbor.hip compiled by g++ against the shim
tests/emul/common.h (HIP's execution model on host threads).  The kernel file is copied next to the harness so that its `#include "common.h"`
resolves to the shim, not to the hipcc header of the same name.  python tests/emul/build_emul.py"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def build(force=False, sanitize=False):
    """sanitize: -fsanitize=address,undefined (load it into a python started with LD_PRELOAD=libasan.so: tests/test_neighbor_emul_cpu.py does)"""
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(ROOT, "marius_amd", "csrc", "kernels", "neighbor.hip")
    lib = os.path.join(out_dir, "libneighbor_emul_asan.so" if sanitize else "libneighbor_emul.so")
    deps = [src, os.path.join(HERE, "common.h"), os.path.join(HERE, "neighbor_emul.cpp"), os.path.join(ROOT, "include", "marius_hip.h")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    shutil.copyfile(src, os.path.join(out_dir, "neighbor.hip.inc"))
    shutil.copyfile(os.path.join(HERE, "neighbor_emul.cpp"), os.path.join(out_dir, "neighbor_emul.cpp"))
    cmd = ["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wno-unknown-pragmas",
           # -Bsymbolic: the emulated library defines the SAME symbols as libmarius_hip.so (entry points, and the kernels' names, which are host-side launch
           # stubs there); a process that has the HIP library loaded globally (marius_amd.host() does) must not get them interposed into this one
           "-Wl,-Bsymbolic", "-I" + HERE, "-I" + os.path.join(ROOT, "include"),
           os.path.join(out_dir, "neighbor_emul.cpp"), "-o", lib] + (["-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if sanitize else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulation build failed:\n" + r.stdout)
    return lib


if __name__ == "__main__":
    print(build(force=True))
    print(build(force=True, sanitize=True))
