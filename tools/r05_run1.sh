#!/bin/bash
# round 5, first GPU call: the parity suite with the new default-path tests, the clock experiment, LDS counters of the correlation kernel
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q -s -k "default or sweep or stress" 2>&1 | grep -v "Warn\|amdgpu.ids\|MIOpen" | tail -40) > $OUT/pytest_new.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Warn\|amdgpu.ids\|MIOpen" | tail -8) > $OUT/pytest_gpu.txt
timeout 300 python3 tools/corr_clock.py --out $OUT/corr_clock.json > $OUT/corr_clock.log 2>&1
cd /tmp
rocprofv3-avail list 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $OUT/counters_avail.txt
for pass in d e f; do
  case $pass in
    d) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS";;
    e) C="SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM";;
    f) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES";;
  esac
  rm -rf /tmp/pmc_corr_$pass
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_corr_$pass -- python3 $GRAFT_REPO_ROOT/tools/prof_corr.py 0 6 > /tmp/pmc_corr_$pass.log 2>&1
  tail -3 /tmp/pmc_corr_$pass.log
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py corr_mfma_dma_kernel $OUT/corr_lds_pmc.json "correlation2d 1x256x544x960 md=4 fp32 (tools/prof_corr.py 0 6): LDS / VMEM / busy counters" /tmp/pmc_corr_d /tmp/pmc_corr_e /tmp/pmc_corr_f
ls -la $OUT
