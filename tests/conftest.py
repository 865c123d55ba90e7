import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def blob():
    from tsid_control_amd.model import ModelBlob
    return ModelBlob()


@pytest.fixture(scope="session")
def oracle(blob):
    from oracle.oracle import Oracle, build
    build()
    return Oracle(blob.raw)


@pytest.fixture(scope="session")
def params(blob):
    from tsid_control_amd.conf import RobotConfig
    from tsid_control_amd.params import pack_params
    return pack_params(RobotConfig(), blob.effort_limit, blob.velocity_limit)


def se3vec(oMf12):
    """oracle frame placement (R row-major 9, p 3) -> tsid SE3ToVector layout (p 3, R col-major 9)."""
    R = np.asarray(oMf12[:9]).reshape(3, 3)
    return np.concatenate([oMf12[9:], R.T.reshape(-1)])


@pytest.fixture(scope="session")
def standing(blob, oracle):
    """Reset state as WalkController.__init__ leaves it (WalkController.py:22-26,72-79,81,122,151,164)."""
    q = blob.q0
    v = np.zeros(26)
    t = oracle.terms(q, v)
    q[2] -= t["oMf"][0][11]
    t = oracle.terms(q, v)
    foot_ref = np.zeros((2, 24))
    foot_ref[:, 3] = foot_ref[:, 7] = foot_ref[:, 11] = 1.0
    return dict(q=q, v=v, terms=t, contact_ref=np.stack([se3vec(t["oMf"][0]), se3vec(t["oMf"][1])]),
                foot_ref=foot_ref, com_ref=np.concatenate([t["com"], np.zeros(6)]), posture_ref=q[7:].copy(),
                cop_frames=t["oMf"].copy())


def oracle_state(n, standing):
    """Env-major float64 state arrays for or_env_step_batch, every env at the reset state."""
    from oracle.oracle import new_state
    st = new_state(n)
    st["q"][:] = standing["q"]
    st["qpos"][:] = standing["q"]  # main.py:64
    st["com_ref"][:] = standing["com_ref"]
    st["posture_ref"][:] = standing["posture_ref"]
    st["foot_ref"][:] = standing["foot_ref"]
    st["contact_ref"][:] = standing["contact_ref"]
    st["cop_frames"][:] = standing["cop_frames"]
    return st
