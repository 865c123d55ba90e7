"""Host-side logic that needs no GPU: the RobotConfig mirror, parameter packing, the C-ABI library
(loads, exports every symbol include/tsidb.h declares - no compute calls), loud failure off-GPU."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

# every attribute of the reference's ctrl/conf.py:8-75 with its value
REFERENCE_CONF = dict(
    robot_path="./robot/v1", root_urdf="./robot/v1/urdf", urdf="./robot/v1/urdf/robot_mod.urdf",
    pin_urdf="./robot/v1/urdf/robot_mod.urdf", mjcf="./robot/v1/mujoco/scene.xml", srdf="./robot/v1/urdf/robot.srdf",
    lf_fixed_joint="left_sole_joint_fixed", rf_fixed_joint="right_sole_joint_fixed", dt=0.002, step_height=0.2,
    step_width=0.2, step_length=0.3, step_duration=0.5, rise_ratio=0.5, lxn=0.055, lyn=0.0275, lxp=0.055, lyp=0.0275,
    lz=0.0, mu=0.5, fMin=10.0, fMax=1000.0, w_contact=-1.0, w_forceRef=1e-5, kp_contact=10.0, w_foot=1e-1, kp_foot=10.0,
    w_com=1e-1, kp_com=10.0, w_posture=1e-1, kp_posture=10.0, tau_max_scaling=5.0, v_max_scaling=10.0,
    w_torque_bounds=1e-2, w_joint_bounds=1e-2, visualizer=None)


def test_robot_config_surface():
    from tsid_control_amd import RobotConfig
    for k, v in REFERENCE_CONF.items():
        assert getattr(RobotConfig, k) == v, k
    assert np.array_equal(RobotConfig.contactNormal, [0, 0, 1]) and np.array_equal(RobotConfig.masks_posture, np.ones(20))
    assert RobotConfig.gain_vector.tolist() == [100, 100, 10, 5, 5, 1, 1, 1, 10, 10, 10, 10, 5, 5, 1, 1, 1, 10, 10, 10]


def test_param_packing(blob):
    from tsid_control_amd import RobotConfig
    from tsid_control_amd import params as P
    p = P.pack_params(RobotConfig(), blob.effort_limit, blob.velocity_limit)
    assert p.shape == (P.P_COUNT,) and p[P.P_DT] == 0.002 and p[P.P_KD_COM] == 2 * np.sqrt(10.0)
    assert np.all(p[P.P_TAU_MAX:P.P_TAU_MAX + 20] == 50.0) and np.all(p[P.P_V_MAX:P.P_V_MAX + 20] == 100.0)
    cp = p[P.P_CPOINTS:P.P_CPOINTS + 12].reshape(4, 3)   # WalkController.py:55-57
    assert cp.tolist() == [[-0.055, -0.0275, -0.0], [-0.055, 0.0275, -0.0], [0.055, -0.0275, -0.0], [0.055, 0.0275, -0.0]]
    assert p[P.P_KP_POSTURE] == 1000.0 and abs(p[P.P_KD_POSTURE + 5] - 2 * np.sqrt(10.0)) < 1e-15
    soft = RobotConfig(); soft.w_contact = 1.0
    with pytest.raises(NotImplementedError):
        P.pack_params(soft, blob.effort_limit, blob.velocity_limit)


def test_param_indices_agree_with_headers():
    """include/tsidb.h, oracle/oracle.h and params.py must describe one layout."""
    from tsid_control_amd import params as P
    h = (ROOT / "include" / "tsidb.h").read_text()
    assert "TSIDB_P_COUNT = 128" in h and P.P_COUNT == 128
    assert P.P_NORMAL == 16 and P.P_CPOINTS == 19 and P.P_KP_POSTURE == 31 and P.P_MAX_ITER == 111 and P.P_SIM_ENABLED == 112


def test_cabi_library_exports_declared_symbols():
    lib = ROOT / "tsid_control_amd" / "libtsidb.so"
    assert lib.exists(), "libtsidb.so not built: python -c 'import __graft_entry__ as g; g.build()'"
    header = (ROOT / "include" / "tsidb.h").read_text()
    declared = sorted(set(re.findall(r"\b(tsidb_[a-z_]+)\s*\(", header)))
    assert len(declared) >= 11
    L = ctypes.CDLL(str(lib))
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/tsidb.h but not exported"
    from tsid_control_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared
    # pure-host query, no GPU involved
    _lib.load()
    assert 0 < L.tsidb_lds_bytes(1, 0) < L.tsidb_lds_bytes(0, 0) <= 65536
    assert 0 < L.tsidb_lds_bytes(0, 1) <= 65536


def test_no_cpu_fallback():
    """Off-GPU the product path must fail loudly, never compute."""
    import torch
    from tsid_control_amd import RobotConfig, WalkController, _lib
    with pytest.raises(_lib.TsidbError):
        WalkController(RobotConfig(), num_envs=2, device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            WalkController(RobotConfig(), num_envs=2, device="cuda:0")


def test_product_never_imports_oracle():
    for f in (ROOT / "tsid_control_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".hpp", ".h"):
            txt = f.read_text()
            assert "oracle" not in txt.replace("no oracle", "").lower() or f.name == "model_compiler.py", f


def test_truncated_model_blob_is_rejected_on_the_host(blob):
    """tsidb_create validates the section table before any HIP call: a truncated or corrupt blob gives an
    error message, never a read past the buffer (no GPU needed - the check precedes device selection)."""
    from tsid_control_amd import _lib
    from tsid_control_amd.params import P_COUNT
    L = _lib.load()
    p = np.zeros(P_COUNT)
    raw = blob.raw
    for cut in (20, 16 + 40 * 3, len(raw) // 2, len(raw) - 8):
        h = ctypes.c_void_p()
        rc = L.tsidb_create(raw[:cut], cut, p.ctypes.data_as(ctypes.c_void_p), P_COUNT, 4, 0, 0, ctypes.byref(h))
        assert rc != 0 and b"model blob" in L.tsidb_last_error(h), cut
        L.tsidb_destroy(h)


def test_bench_contract_and_execution_options():
    """bench.py's command line (the driver's contract: --gpus / --steps / --warmup, defaults that finish in minutes) and the
    execution options that must not change results: every secondary run names every attribute run_workload reads, the
    option numbers of include/tsidb.h are the ones the facade passes, RobotConfig carries their switches."""
    import sys
    import importlib
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
        assert (a.gpus, a.steps, a.warmup, a.envs, a.dtype, a.workload) == (1, 5000, 20, 4096, "f64", "walk")
        assert a.device_plan is False and a.no_overlap is False and a.closed_loop is False
        sys.argv = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--device-plan"]
        b = bench.parse()
        assert (b.steps, b.warmup, b.device_plan) == (20, 5, True)
    finally:
        sys.argv = argv
    assert bench.TICK_WORDS == 347 and bench.SIM_WORDS == 185 and bench.HBM_PEAK_GBS == 8000.0
    hdr = (ROOT / "include" / "tsidb.h").read_text()
    m = re.search(r"enum \{ TSIDB_OPT_SIM_WAVES = (\d+), TSIDB_OPT_LDS_PAD = (\d+), TSIDB_OPT_CU_SPLIT = (\d+), TSIDB_OPT_SIM_PACK = (\d+), TSIDB_OPT_QP_FAST_EQ = (\d+) \}", hdr)
    assert m and [int(g) for g in m.groups()] == [1, 2, 3, 4, 5]
    wc_src = (ROOT / "tsid_control_amd" / "walk_controller.py").read_text()
    assert "tsidb_set_option(self._h, 1, sw)" in wc_src and "tsidb_set_option(self._h, 4, sp)" in wc_src and "tsidb_set_option(self._h, 5, fe)" in wc_src
    from tsid_control_amd import RobotConfig
    assert RobotConfig.sim_pack == -1 and RobotConfig.qp_fast_equalities == -1 and RobotConfig.sim_waves == 0 and RobotConfig.pipeline_sim_batch == 0
    # the flat compute-roofline scalars come from the committed counter passes
    import json
    t = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    for k in ("k_tick", "k_sim"):
        v = t["valu"][k]
        assert 0.3 < 2.0 * v["issue_floor_cycles_per_env"] / (4.0 * v["wave_cycles_per_env"]) < 0.7
        assert 100.0 < v["launch_us_back_to_back"] < 200.0 and 0.8 < t[k]["bytes_per_launch"] / (4096 * 8 * (347 if k == "k_tick" else 185)) < 1.4
