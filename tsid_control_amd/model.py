"""Reader for the compiled model blob (written by model_compiler.py; layout in include/tsidb_model.h)."""
import struct
from pathlib import Path

import numpy as np

DEFAULT_BLOB = Path(__file__).parent / "assets" / "op3_v1.tsidb"


class ModelBlob:
    """Raw bytes + named sections as numpy arrays (views; read-only)."""

    def __init__(self, path=None):
        self.path = Path(path) if path else DEFAULT_BLOB
        self.raw = self.path.read_bytes()
        if self.raw[:8] != b"TSIDBM01":
            raise ValueError(f"{self.path}: not a TSIDB model blob")
        n, _ = struct.unpack_from("<II", self.raw, 8)
        self.sections = {}
        for i in range(n):
            off = 16 + 40 * i
            name = self.raw[off:off + 24].split(b"\0")[0].decode()
            code, cnt, o = struct.unpack_from("<IIQ", self.raw, off + 24)
            dt = np.dtype("<f8") if code == 0 else np.dtype("<i4")
            self.sections[name] = np.frombuffer(self.raw, dtype=dt, count=cnt, offset=o)

    def __getitem__(self, k):
        return self.sections[k]

    @property
    def q0(self):
        return self["pin_q0"].copy()

    @property
    def effort_limit(self):
        return self["pin_effort"].copy()

    @property
    def velocity_limit(self):
        return self["pin_velocity"].copy()
