"""Drop-in boundary, source level: a translation unit written against the signatures of the reference's own headers must
compile against the host layer's header (`tests/boundary/reference_style_usage.cpp`; syntax + semantic analysis only, nothing is run)."""
import os
import subprocess
import sysconfig

import torch
from torch.utils import cpp_extension as ce

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_style_user_code_compiles_against_the_host_header():
    src = os.path.join(ROOT, "tests", "boundary", "reference_style_usage.cpp")
    inc = ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROOT, "include"),
                                                   "-I" + os.path.join(ROOT, "marius_amd", "csrc", "host"), "-I/opt/rocm/include"]
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-fopenmp", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", "-Wno-deprecated-declarations"] + inc + [src]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
