#!/bin/bash
# product libraries (v1, v0) and the stamps diagnostic build, in parallel
cd "$(dirname "$0")/.."
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -disable-machine-licm -ffp-contract=on"
S=tsid_control_amd/csrc/tsidb_api.hip
mkdir -p tools/_diag
($H $@ -o tsid_control_amd/libtsidb.so $S 2>&1 | grep -E "error|Error") &
($H $@ -DTSIDB_STAMPS -o tools/_diag/libtsidb_stamps.so $S 2>&1 | grep -E "error|Error") &
($H $@ '-DTSIDB_TOPOLOGY_HEADER="tsidb_topology_v0.hpp"' -o tsid_control_amd/libtsidb_v0.so $S 2>&1 | grep -E "error|Error") &
wait
ls -la tsid_control_amd/*.so tools/_diag/*.so
