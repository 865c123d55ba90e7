"""ctypes binding of libtsidb.so (include/tsidb.h).  There is no CPU fallback: if the HIP library
is missing or a call fails, this raises."""
import ctypes as C
from pathlib import Path

_HERE = Path(__file__).parent
import os

# TSIDB_LIB_PATH selects a diagnostic build (tools/stamp_profile.py); the product default is in-tree
LIB_PATH = Path(os.environ.get("TSIDB_LIB_PATH", _HERE / "libtsidb.so"))

SYMBOLS = ["tsidb_dims", "tsidb_create", "tsidb_destroy", "tsidb_last_error", "tsidb_set_params", "tsidb_set_refs", "tsidb_reset",
           "tsidb_tick", "tsidb_sim", "tsidb_step", "tsidb_rbd_terms", "tsidb_lds_bytes", "tsidb_walk_update", "tsidb_set_env_params", "tsidb_set_cop_ref",
           "tsidb_reset_done", "tsidb_set_posture_bias", "tsidb_walk_plan", "tsidb_set_option", "tsidb_tick_walk", "tsidb_sim_batch", "tsidb_stream_create", "tsidb_stream_destroy", "tsidb_get_option"]

_libs = {}


class TsidbError(RuntimeError):
    pass


class WalkArgs(C.Structure):
    """tsidb_walk_args (include/tsidb.h): tsidb_walk_update's arguments as one block"""
    _fields_ = [("coef", C.c_void_p), ("side", C.c_void_p), ("nsteps", C.c_void_p), ("rest", C.c_void_p), ("com", C.c_void_p),
                ("K", C.c_int), ("t", C.c_double), ("step_duration", C.c_double), ("t_start", C.c_double), ("omega", C.c_double),
                ("com_z0", C.c_double), ("com_drop", C.c_double), ("frames", C.c_void_p), ("t_offset", C.c_void_p),
                ("ncon", C.c_void_p), ("con_pairs", C.c_void_p), ("td_latch", C.c_void_p), ("td_fraction", C.c_double),
                ("t_device", C.c_void_p)]


def dims9(L):
    """the nine ints a library was built for, in the order of the blob's model_dims section:
    NJ, NQ, NV, NA, sim bodies, has_sim, collision geoms, contact dimension, damped joints"""
    out = (C.c_int * 9)()
    L.tsidb_dims(out)
    return tuple(out)


def dims(L):
    """(NJ, NQ, NV, NA, sim bodies, has_sim) of the robot a loaded library was built for"""
    return dims9(L)[:6]


def load_for(model_dims):
    """The library built for a blob's robot (its model_dims section): one libtsidb*.so per robot sits next to this file
    (libtsidb.so = the v1 robot, libtsidb_v0.so = robot/v0).  TSIDB_LIB_PATH (diagnostic builds) is tried first."""
    want = tuple(int(x) for x in model_dims)[:9]   # all nine: two builds may differ in geoms / condim / damping only
    cands = [LIB_PATH] + sorted(p for p in _HERE.glob("libtsidb*.so") if p != LIB_PATH)
    for p in cands:
        if p.exists():
            L = load(p)
            if dims9(L)[:len(want)] == want:
                return L
    raise TsidbError(f"no libtsidb*.so in {_HERE} is built for a robot with dimensions {want}: compile the blob's topology "
                     "header into a library (python -c 'import __graft_entry__ as g; g.build()')")


def load(path=None):
    """Load a libtsidb*.so once (default: libtsidb.so, the v1 robot); raise loudly when it has not been built."""
    path = Path(path) if path is not None else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not path.exists():
        raise TsidbError(f"{path} is missing: build the HIP extension first "
                         "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU path")
    # torch first: its bundled HIP runtime must be the one this process initialises - the library is handed torch's device
    # pointers and streams, and loaded before torch it would bind /opt/rocm's libamdhip64 instead (a second runtime in one
    # process: hipGetDeviceCount then finds no device)
    import torch  # noqa: F401
    L = C.CDLL(str(path))
    vp, i32p = C.c_void_p, C.c_void_p
    L.tsidb_create.argtypes = [vp, C.c_size_t, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tsidb_destroy.argtypes = [vp]
    L.tsidb_last_error.argtypes = [vp]
    L.tsidb_last_error.restype = C.c_char_p
    L.tsidb_set_params.argtypes = [vp, vp, C.c_int]
    L.tsidb_set_refs.argtypes = [vp] * 7
    L.tsidb_reset.argtypes = [vp, i32p, C.c_int, vp, vp, vp, vp, vp, vp]
    L.tsidb_tick.argtypes = [vp] * 8 + [C.c_int] + [vp] * 3
    L.tsidb_sim.argtypes = [vp] * 11
    L.tsidb_step.argtypes = [vp] * 11 + [C.c_int] + [vp] * 4 + [C.c_int, vp]
    L.tsidb_rbd_terms.argtypes = [vp] * 10
    L.tsidb_lds_bytes.argtypes = [C.c_int, C.c_int]
    L.tsidb_set_env_params.argtypes = [vp, vp, vp]
    L.tsidb_set_cop_ref.argtypes = [vp, vp]
    L.tsidb_walk_update.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int] + [C.c_double] * 6 + [vp, vp, vp, vp, vp, C.c_double, vp, vp]
    L.tsidb_dims.argtypes = [C.POINTER(C.c_int)]
    L.tsidb_reset_done.argtypes = [vp, vp, C.c_int] + [vp] * 7
    L.tsidb_set_posture_bias.argtypes = [vp, vp]
    L.tsidb_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.tsidb_sim_batch.argtypes = [vp, C.c_int] + [vp] * 11
    L.tsidb_tick_walk.argtypes = [vp, vp] + [vp] * 7 + [C.c_int] + [vp] * 5
    L.tsidb_walk_plan.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int] + [vp] * 9 + \
                                 [C.c_double, vp, vp]
    for s in SYMBOLS:
        if s != "tsidb_last_error":
            getattr(L, s).restype = C.c_int
    _libs[path] = L
    return L


def check(L, handle, rc, what):
    if rc != 0:
        msg = L.tsidb_last_error(handle)
        raise TsidbError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
