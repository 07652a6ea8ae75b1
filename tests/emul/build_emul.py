"""TEST INFRASTRUCTURE: the CPU build of kernel files.  tests/emul/_build/libkernels_emul.so = the kernel files named in FILES, each copied next to
the harness with its `kernel<<<grid, block, shmem, stream>>>(args);` launches rewritten to `emul::launch(grid, block, [&] { kernel(args); });`,
compiled by g++ against the shim tests/emul/common.h (HIP's execution model on host threads; the copies' `#include "common.h"` resolves to the
shim, not to the hipcc header of the same name).  The product sources are not touched and nothing under marius_amd/ can reach this library.
python tests/emul/build_emul.py"""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
KDIR = os.path.join(ROOT, "marius_amd", "csrc", "kernels")
FILES = ["neighbor.hip", "rows.hip", "rng.hip", "encoder.hip", "segreduce.hip", "sort_unique.hip", "exchange.hip", "lp_decoder.hip", "eval_filter.hip"]
HEADERS = ["seg_plan.h", "lp_common.h"]  # kernel-side headers the files include by name: copied next to them (their own `#include "common.h"` then finds the shim)


def _match(s, i, open_ch, close_ch):
    """index just past the bracket that closes the one at s[i]"""
    depth = 0
    while True:
        if s[i] == open_ch:
            depth += 1
        elif s[i] == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1


def _split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    parts.append(cur.strip())
    return parts


def transform(src):
    """rewrite every kernel launch of a .hip source for the emulation build; dynamic LDS (`extern __shared__ T name[];`) becomes a pointer into the
    launch's buffer; clang's ext_vector_type becomes the shim's emul::vec"""
    src = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];", r"\1* \2 = reinterpret_cast<\1*>(emul::g_dyn_smem);", src)
    src = re.sub(r"typedef\s+(\w+)\s+(\w+)\s+__attribute__\(\(ext_vector_type\((\d+)\)\)\);", r"typedef emul::vec<\1, \3> \2;", src)
    out, i = "", 0
    while True:
        j = src.find("<<<", i)
        if j < 0:
            return out + src[i:]
        # kernel name: identifier, optionally followed by <template arguments>, right in front of the chevrons
        k = j
        if src[k - 1] == ">":
            depth, k = 0, k - 1
            while True:
                if src[k] == ">":
                    depth += 1
                elif src[k] == "<":
                    depth -= 1
                    if depth == 0:
                        break
                k -= 1
        m = re.search(r"[A-Za-z_]\w*$", src[:k])
        name = src[m.start():j]
        e = src.index(">>>", j)
        cfg = _split_top(src[j + 3:e])
        grid, block, shmem = cfg[0], cfg[1], (cfg[2] if len(cfg) > 2 else "0")
        a0 = e + 3
        while src[a0] in " \n":
            a0 += 1
        assert src[a0] == "(", src[j - 40:j + 80]
        a1 = _match(src, a0, "(", ")")
        assert src[a1] == ";", src[a1 - 60:a1 + 5]
        # the arguments are evaluated ONCE, by the launching thread, and copied — as a kernel launch does (a `tickets.fetch_add(1)` among them must not run per work-item)
        out += src[i:m.start()] + ("{ auto emul_args_ = std::make_tuple%s; emul::launch(dim3(%s), dim3(%s), [&] { std::apply([](auto&... a) { %s(a...); }, emul_args_); }, (size_t)(%s)); }"
                                   % (src[a0:a1], grid, block, name, shmem))
        i = a1 + 1


def build(force=False, sanitize=False):
    """sanitize: -fsanitize=address,undefined (load it into a python started with LD_PRELOAD=libasan.so: tests/test_kernels_emul_cpu.py does)"""
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libkernels_emul_asan.so" if sanitize else "libkernels_emul.so")
    deps = [os.path.join(KDIR, f) for f in FILES + HEADERS] + [os.path.join(HERE, "common.h"), os.path.join(HERE, "kernels_emul.cpp"), os.path.join(ROOT, "include", "marius_hip.h"),
                                                    os.path.abspath(__file__)]
    if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    for h in HEADERS:
        with open(os.path.join(KDIR, h)) as fh, open(os.path.join(out_dir, h), "w") as oh:
            oh.write(transform(fh.read()))
    for f in FILES:
        with open(os.path.join(KDIR, f)) as fh:
            text = transform(fh.read())
        with open(os.path.join(out_dir, f + ".inc"), "w") as fh:
            fh.write(text)
    with open(os.path.join(HERE, "kernels_emul.cpp")) as fh:
        harness = fh.read()
    with open(os.path.join(out_dir, "kernels_emul.cpp"), "w") as fh:
        fh.write(harness + "".join('#include "%s.inc"\n' % f for f in FILES))
    cmd = ["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wno-unknown-pragmas",
           # -Bsymbolic: the emulated library defines the SAME symbols as libmarius_hip.so (entry points, and the kernels' names, which are host-side launch
           # stubs there); a process that has the HIP library loaded globally (marius_amd.host() does) must not get them interposed into this one
           "-Wl,-Bsymbolic", "-I" + HERE, "-I" + os.path.join(ROOT, "include"), os.path.join(out_dir, "kernels_emul.cpp"), "-o", lib]
    cmd += ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if sanitize else []
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulation build failed:\n" + r.stdout[-6000:])
    return lib


if __name__ == "__main__":
    print(build(force=True))
    print(build(force=True, sanitize=True))
