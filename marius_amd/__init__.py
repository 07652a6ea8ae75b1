"""marius_amd — MI355X-native link-prediction training hot path behind Marius's operator API.

  marius_amd.hip        ctypes binding of the C-ABI (libmarius_hip.so, include/marius_hip.h)
  marius_amd.host()     the C++ host classes on libtorch (Storage, CorruptNodeNegativeSampler, DistMult/ComplEx/TransE, Model,
                        DataLoader, SynchronousTrainer, ...) as a pybind11 module
  marius_amd.config     YAML configuration surface (reference: src/python/tools/configuration/marius_config.py defaults)
  marius_amd.marius_train   `marius_train <config.yaml>` entry point
There is no CPU fallback: a missing native library raises.
"""
import importlib.util
import os
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_host = None


def host():
    """Import marius_amd/lib/_marius_host*.so (built by `python -m marius_amd.build --host`)."""
    global _host
    if _host is None:
        import ctypes

        import torch  # noqa: F401  (libtorch must be loaded first)

        from . import hip

        hip.lib()
        # the host module's marius_* symbols bind to the first RTLD_GLOBAL copy: it must be the one hip.lib() opened (MARIUS_HIP_LIB experiment builds)
        ctypes.CDLL(os.environ.get("MARIUS_HIP_LIB", hip.LIB_PATH), mode=ctypes.RTLD_GLOBAL)
        path = os.path.join(_HERE, "lib", "_marius_host" + sysconfig.get_config_var("EXT_SUFFIX"))
        if not os.path.exists(path):
            raise ImportError("%s not found — build it with `python -m marius_amd.build --host`" % path)
        spec = importlib.util.spec_from_file_location("_marius_host", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _host = mod
    return _host
