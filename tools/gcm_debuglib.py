"""ctypes binding of libgcm_hip_debug.so (include/gcm_hip_debug.h): the measurement aids of bench.py and tools/.
NOT part of the product: nothing under graph-conv-memory_amd/gcm loads this library, and the product library
(libgcm_hip.so) exports none of these entry points."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_debug.so")
_P, _I = ctypes.c_void_p, ctypes.c_int
PROTOTYPES = {
    "gcm_debug_time_next_launch": (_I, [_P, _P]),
    "gcm_debug_time_rows_rollout": (_I, [_P] * 5 + [_I, _P, _I, _I, _I] + [_P] * 4 + [_I] * 6 + [_P]),
    "gcm_debug_time_cached_rollout": (_I, [_P] * 5 + [_I, _P, _P, _I, _I, _I] + [_P] * 7 + [_I] * 6 + [_P]),
    "gcm_debug_empty_graph_cadence": (_I, [_I, _I, _I, _I, ctypes.POINTER(ctypes.c_float)]),
    "gcm_debug_empty_launch_duration": (_I, [_I, _I, _I, ctypes.POINTER(ctypes.c_float),
                                              ctypes.POINTER(ctypes.c_float)]),
    # product entry points the aids are used with (the same kernels, built from the same sources)
    "gcm_dense_rows_cached_weight_image": (_I, [_P, _P, _I, _I, _I, _P]),
}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: `make -C graph-conv-memory_amd/csrc debug`")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def launch_floor(grid=256, block=64, nodes=128, replays=50):
    """(us per node of a replayed HIP graph of `nodes` empty kernels, mean begin->end duration of one empty dispatch,
    begin-to-begin cadence of back-to-back empty launches without a graph) on this box."""
    a, d, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
    rc = lib().gcm_debug_empty_graph_cadence(nodes, grid, block, replays, ctypes.byref(a))
    assert rc == 0, rc
    rc = lib().gcm_debug_empty_launch_duration(nodes, grid, block, ctypes.byref(d), ctypes.byref(c))
    assert rc == 0, rc
    return a.value, d.value, c.value
