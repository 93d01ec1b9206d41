#!/bin/bash
# Round profiles on the GPU box: kernel stats of the default bench command, of one forward, the stream timeline, and the
# PMC passes (separate runs, counters only with --kernel-trace) for the kernels DESIGN.md quotes.  Output: gpurun_out/$1/
set -x
R=${1:-r02}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R/prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# 1. the default bench command, kernel stats
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/p1.err
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv
# 2. one eager forward: launches / step and per-kernel totals
rm -rf /tmp/p2; rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -- python3 $GRAFT_REPO_ROOT/tools/prof_forward.py 3 > /tmp/p2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_window.py $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) 3 > $OUT/forward_kernel_stats.txt
# 3. stream timeline of the replayed graph
python3 $GRAFT_REPO_ROOT/tools/stamp_timeline.py > $OUT/forward_stream_timeline.txt 2>/dev/null
# 4. PMC passes
for op in pointconv knn16 knn3 corr3d; do
  for pass in a b c; do
    case $pass in
      a) C="FETCH_SIZE";;
      b) C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";;
      c) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE";;
    esac
    rm -rf /tmp/pmc_${op}_$pass
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${op}_$pass -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py $op 6 > /tmp/pmc_${op}_$pass.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py pointconv_fused_kernel $OUT/pointconv_pmc.json "PointConvNoSampling 195->128, B=4, N=4096 (tools/prof_ops.py pointconv 6)" /tmp/pmc_pointconv_a /tmp/pmc_pointconv_b /tmp/pmc_pointconv_c
python3 tools/pmc_summary.py "knn_mfma_kernel" $OUT/knn16_pmc.json "k_nearest_neighbor 3-D k=16, B=4, 8192 -> 4096 (tools/prof_ops.py knn16 6)" /tmp/pmc_knn16_a /tmp/pmc_knn16_b /tmp/pmc_knn16_c
python3 tools/pmc_summary.py "knn_mfma_kernel" $OUT/knn3_pmc.json "k_nearest_neighbor 3-D k=3, B=4, 4096 -> 4096 (tools/prof_ops.py knn3 6)" /tmp/pmc_knn3_a /tmp/pmc_knn3_b /tmp/pmc_knn3_c
python3 tools/pmc_summary.py corr3d_cost_kernel $OUT/corr3d_cost_pmc.json "Correlation3D cost kernel, B=4, N=4096, C=32 (tools/prof_ops.py corr3d 6)" /tmp/pmc_corr3d_a /tmp/pmc_corr3d_b /tmp/pmc_corr3d_c
# 5. corr microbench PMC (kernel of roofline_corr)
for pass in a b c; do
  case $pass in
    a) C="FETCH_SIZE";;
    b) C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";;
    c) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE";;
  esac
  rm -rf /tmp/pmc_corr_$pass
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_corr_$pass -- python3 $GRAFT_REPO_ROOT/tools/prof_corr.py 0 6 > /tmp/pmc_corr_$pass.log 2>&1)
done
python3 tools/pmc_summary.py corr_mfma_dma_kernel $OUT/corr_microbench_pmc.json "correlation2d 1x256x544x960 md=4 fp32 (tools/prof_corr.py 0 6)" /tmp/pmc_corr_a /tmp/pmc_corr_b /tmp/pmc_corr_c
# 6. the hot-path sequence alone, eager, kernel stats; and the FPS PMC passes
(cd /tmp && rm -rf /tmp/p6 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p6 -- python3 $GRAFT_REPO_ROOT/bench.py --workload hotpath --eager --steps 10 --warmup 2 --no-cpu-baseline --no-corr-microbench > /tmp/p6.log 2>&1)
cp $(find /tmp/p6 -name "*kernel_stats.csv" | head -1) $OUT/hotpath_kernel_stats.csv
# 7. the bench lines themselves (default and dsec), un-profiled
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --config dsec > $OUT/bench_dsec.json 2> $OUT/bench_dsec.err
ls -la $OUT
