"""Loader / builder of the C++ autograd node of the fused step (csrc/torch_ext/step_ext.cpp).

The node is host plumbing around the same C-ABI calls the Python autograd Function makes
(gcm_dense_step_fwd / gcm_dense_step_bwd in libgcm_hip.so); it exists because the per-step loop of
the reference's callers is host-bound.  Built in-tree by `python __graft_entry__.py`
(`build()` below); when the built module is absent the Python Function in _ops.py is used - the
same kernels, more interpreter time per step."""
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_DIR = os.path.join(_HERE, "_lib")
_EXT_DIR = os.path.join(_LIB_DIR, "ext")
_NAME = "gcm_torch_ext"
_SO = os.path.join(_EXT_DIR, _NAME + ".so")
_SRC = os.path.join(os.path.dirname(_HERE), "csrc", "torch_ext", "step_ext.cpp")
_INCLUDE = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include")

_mod = None
_tried = False


def build(verbose=False):
    """Compile the extension into gcm/_lib/ext/ (host C++ only; links libgcm_hip.so)."""
    import torch
    from torch.utils import cpp_extension
    from . import _hip
    _hip.lib()   # libgcm_hip.so must exist (and is then already mapped when the module loads)
    os.makedirs(_EXT_DIR, exist_ok=True)
    global _mod, _tried
    _mod = cpp_extension.load(
        name=_NAME, sources=[_SRC],
        # host C++ only; the ROCm include path is for c10/hip (current stream / device of the
        # process), whose library torch has loaded already
        extra_include_paths=[_INCLUDE, os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")],
        extra_cflags=["-O2", "-std=c++17", "-Wno-deprecated-declarations"]
        + (["-DGCM_HOST_PROF"] if os.environ.get("GCM_HOST_PROF") == "1" else []),   # (tools/hosttime.py: segments of RowsFast.step)
        extra_ldflags=[f"-L{_LIB_DIR}", "-lgcm_hip", "-Wl,-rpath,'$$ORIGIN/..'",
                       f"-L{os.path.join(os.path.dirname(torch.__file__), 'lib')}", "-lc10_hip",
                       "-ltorch_python"],
        build_directory=_EXT_DIR, verbose=verbose)
    _tried = True
    return _mod


def module():
    """The built extension module, or None when it has not been built."""
    global _mod, _tried
    if _mod is None and not _tried:
        _tried = True
        if os.environ.get("GCM_NO_TORCH_EXT") != "1" and os.path.exists(_SO):
            import torch  # noqa: F401  (libtorch must be loaded first)
            from . import _hip
            _hip.lib()
            spec = importlib.util.spec_from_file_location(_NAME, _SO)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _mod = mod
    return _mod
