"""Build the native pieces in-tree (no JIT cache): `python -m marius_amd.build [--host] [--oracle]`.

  libmarius_hip.so   HIP kernels + C-ABI (include/marius_hip.h), hipcc --offload-arch=gfx950
  _marius_host*.so   C++ host classes on libtorch mirroring the reference's operator API (optional, --host)
  oracle/_build, oracle/_ref   test infrastructure (optional, --oracle)
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "marius_amd")
KERNEL_DIR = os.path.join(PKG, "csrc", "kernels")
HOST_DIR = os.path.join(PKG, "csrc", "host")
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(PKG, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_kernels(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(KERNEL_DIR, f) for f in os.listdir(KERNEL_DIR) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "marius_hip.h"))
    srcs = sorted(f for f in os.listdir(KERNEL_DIR) if f.endswith(".hip"))
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + KERNEL_DIR]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(KERNEL_DIR, s)
        obj = os.path.join(OBJ_DIR, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])
    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    lib = os.path.join(LIB_DIR, "libmarius_hip.so")
    if force or jobs or _newer(objs, lib):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    if verbose:
        print("built", lib, "(%d objects recompiled)" % len(jobs))
    return lib


def build_host(force=False, verbose=True):
    """C++ host layer (libtorch) + pybind11 module `_marius_host`."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce

    srcs = sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".cpp"))
    if not srcs:
        return None
    hdrs = [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".h")] + [os.path.join(ROOT, "include", "marius_hip.h")]
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    target = os.path.join(LIB_DIR, "_marius_host" + ext)
    os.makedirs(OBJ_DIR, exist_ok=True)
    tdir = os.path.dirname(torch.__file__)
    inc = ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROOT, "include"),
                                                   "-I" + HOST_DIR, "-I/opt/rocm/include"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cflags = ["-O2", "-std=c++17", "-fPIC", "-fopenmp", "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi, "-DTORCH_EXTENSION_NAME=_marius_host",
              "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-Wno-deprecated-declarations", "-Wno-attributes"]
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(OBJ_DIR, "host_" + os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + hdrs, obj):
            jobs.append(["g++"] + cflags + inc + ["-c", src, "-o", obj])
    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    if force or jobs or _newer(objs, target):
        libs = ["-L" + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-Wl,-rpath," + os.path.join(tdir, "lib"),
                "-L" + LIB_DIR, "-lmarius_hip", "-Wl,-rpath,$ORIGIN", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
        if os.path.exists(os.path.join(tdir, "lib", "libtorch_hip.so")):
            libs += ["-ltorch_hip", "-lc10_hip"]
        _run(["g++", "-shared", "-fopenmp", "-o", target] + objs + libs)
    if verbose:
        print("built", target, "(%d objects recompiled)" % len(jobs))
    return target


def build_oracle(verbose=True):
    out = _run(["make", "-C", os.path.join(ROOT, "oracle")])
    if verbose:
        print(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--host", action="store_true")
    ap.add_argument("--oracle", action="store_true")
    a = ap.parse_args()
    build_kernels(force=a.force)
    if a.host:
        build_host(force=a.force)
    if a.oracle:
        build_oracle()


if __name__ == "__main__":
    sys.exit(main())
