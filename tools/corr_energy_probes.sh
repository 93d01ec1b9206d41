#!/bin/bash
# Diagnostic builds of corr_mfma_dma_kernel (csrc/correlation.hip, RPE_CORR_PROBE) through tools/corr_clock.py: time, engine clock
# over the very launches (clock stamps), cycles and socket power per operand kind.
#   1: a third of the matrix work (WRONG results)   2: one B-operand LDS read per step instead of three (WRONG results)
#   3 / 4 / 5: 4 / 5 / 6 ring slots instead of 3 (correct results; checked against the parity tests first)
# Build here (build container): tools/corr_energy_probes.sh build ; run on the GPU box: tools/corr_energy_probes.sh run <outdir> [probes...]
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
E=$ROOT/tools/_exp
mkdir -p $E
if [ "$1" = "build" ]; then
  cd $ROOT && python -m rpeflow_amd.build > /dev/null
  for v in 1 2 3 4 5; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -Wno-unused-result -DRPE_CORR_PROBE=$v \
      -I include -I rpeflow_amd/csrc -c rpeflow_amd/csrc/correlation.hip -o /tmp/corr_probe_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $E/librpeflow_corrp$v.so $(ls rpeflow_amd/csrc/build/*.o | grep -v /correlation.o) /tmp/corr_probe_$v.o
  done
  ls -la $E
  exit 0
fi
OUT=$2; shift 2
mkdir -p $OUT
cd $ROOT
for v in "$@"; do
  if [ $v -ge 3 ]; then
    RPE_HIP_LIB=$E/librpeflow_corrp$v.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "correlation" 2>&1 | tail -2 > $OUT/corr_probe${v}_parity.txt
  fi
  RPE_ALLOW_DIAGNOSTIC_LIB=1 RPE_HIP_LIB=$E/librpeflow_corrp$v.so timeout 300 python3 tools/corr_clock.py --out $OUT/corr_clock_probe$v.json > $OUT/corr_clock_probe$v.log 2>&1
done
timeout 300 python3 tools/corr_clock.py --out $OUT/corr_clock_shipped.json > $OUT/corr_clock_shipped.log 2>&1
