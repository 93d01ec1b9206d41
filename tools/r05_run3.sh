#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q -s -k "default or sweep or forced_kernels" 2>&1 | grep "default path\|one ulp\|passed\|failed\|Error\|assert") > $OUT/pytest_new.txt
bash tools/corr_energy_probes.sh run $OUT 3 4 5
timeout 600 python3 tools/knn_gate_table.py > $OUT/knn_gate_table.txt 2> $OUT/knn_gate_table.err
ls -la $OUT
