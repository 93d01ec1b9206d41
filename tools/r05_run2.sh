#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q -s -k "default or sweep" 2>&1 | grep -v "Warn\|amdgpu.ids\|MIOpen" | tail -40) > $OUT/pytest_new.txt
(timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Warn\|amdgpu.ids\|MIOpen" | tail -8) > $OUT/pytest_gpu.txt
timeout 300 python3 tools/aten_gpu_census.py hotpath > $OUT/aten_gpu_census_hotpath.txt 2> $OUT/aten_gpu_census_hotpath.err
timeout 300 python3 tools/aten_gpu_census.py forward > $OUT/aten_gpu_census_forward.txt 2> $OUT/aten_gpu_census_forward.err
for v in corrp1 corrp2; do
  RPE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_exp/librpeflow_$v.so timeout 300 python3 tools/corr_clock.py --out $OUT/corr_clock_$v.json > $OUT/corr_clock_$v.log 2>&1
done
timeout 300 python3 tools/corr_clock.py --out $OUT/corr_clock_b.json > $OUT/corr_clock_b.log 2>&1
ls -la $OUT
