#!/bin/bash
# Round-4 profiles on the GPU box (see profiles/README.md): the bench lines, kernel stats of the default bench command and of one
# forward, the stream timeline, PMC passes (separate runs, counters only with --kernel-trace), the 8-rank host rehearsal.
set -x
R=${1:-r04}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R/prof
mkdir -p $OUT
export TMPDIR=/tmp
# 0. the bench lines themselves, un-profiled (first process on this box)
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --workload eval --steps 128 --no-cpu-baseline --no-corr-microbench > $OUT/bench_eval.json 2> $OUT/bench_eval.err
python3 bench.py --config dsec > $OUT/bench_dsec.json 2> $OUT/bench_dsec.err
python3 bench.py --workload eval --steps 128 --no-cpu-baseline --no-corr-microbench --eval-raw-events 300000 > $OUT/bench_eval_raw_events.json 2> $OUT/bench_eval_raw_events.err
cd /tmp
# 1. the default bench command, kernel stats
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --eval-batches 0 --backend none > $OUT/bench_under_rocprof.json 2> /tmp/p1.err
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv
# 2. one eager forward: launches / step and per-kernel totals
rm -rf /tmp/p2; rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -- python3 $GRAFT_REPO_ROOT/tools/prof_forward.py 3 > /tmp/p2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_window.py $(find /tmp/p2 -name "*kernel_trace.csv" | head -1) 3 > $OUT/forward_kernel_stats.txt
# 3. stream timeline of the replayed graph
python3 $GRAFT_REPO_ROOT/tools/stamp_timeline.py > $OUT/forward_stream_timeline.txt 2>/dev/null
# 4. PMC passes
pmc() {  # op-tag command...
  tag=$1; shift
  for pass in a b c; do
    case $pass in
      a) C="FETCH_SIZE";;
      b) C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";;
      c) C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE";;
    esac
    rm -rf /tmp/pmc_${tag}_$pass
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$pass -- "$@" > /tmp/pmc_${tag}_$pass.log 2>&1
  done
}
pmc knn16 python3 $GRAFT_REPO_ROOT/tools/prof_ops.py knn16 6
pmc knn2d python3 $GRAFT_REPO_ROOT/tools/prof_ops.py knn2d 6
pmc pointconv python3 $GRAFT_REPO_ROOT/tools/prof_ops.py pointconv 6
pmc corr python3 $GRAFT_REPO_ROOT/tools/prof_corr.py 0 6
cd $GRAFT_REPO_ROOT
S="python3 tools/pmc_summary.py"
$S knn_mfma_kernel $OUT/knn16_pmc.json "k_nearest_neighbor 3-D k=16, B=8, 8192 -> 4096: the sweep kernel (tools/prof_ops.py knn16 6)" /tmp/pmc_knn16_a /tmp/pmc_knn16_b /tmp/pmc_knn16_c
$S knn_tie_replay_kernel $OUT/knn16_replay_pmc.json "the same search: its tied rows' second launch (tools/prof_ops.py knn16 6)" /tmp/pmc_knn16_a /tmp/pmc_knn16_b /tmp/pmc_knn16_c
$S nearest2d_search_kernel $OUT/knn2d_search_pmc.json "k_nearest_neighbor 2-D k=1, B=8, 4096 points, 144x240 raster queries: the search kernel (tools/prof_ops.py knn2d 6)" /tmp/pmc_knn2d_a /tmp/pmc_knn2d_b /tmp/pmc_knn2d_c
$S nearest2d_build_kernel $OUT/knn2d_build_pmc.json "the same search: binning the 8 clouds (tools/prof_ops.py knn2d 6)" /tmp/pmc_knn2d_a /tmp/pmc_knn2d_b /tmp/pmc_knn2d_c
$S pointconv_fused_kernel $OUT/pointconv_pmc.json "PointConvNoSampling 195->128, B=4, N=4096 (tools/prof_ops.py pointconv 6)" /tmp/pmc_pointconv_a /tmp/pmc_pointconv_b /tmp/pmc_pointconv_c
$S corr_mfma_dma_kernel $OUT/corr_microbench_pmc.json "correlation2d 1x256x544x960 md=4 fp32 (tools/prof_corr.py 0 6)" /tmp/pmc_corr_a /tmp/pmc_corr_b /tmp/pmc_corr_c
for pass in a b c; do
  case $pass in
    a) C="FETCH_SIZE";;
    b) C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";;
    c) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE";;
  esac
  rm -rf /tmp/pmc_fps_$pass
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_fps_$pass -- python3 $GRAFT_REPO_ROOT/tools/prof_fps.py 5 > /tmp/pmc_fps_$pass.log 2>&1)
done
$S fps_pruned2_kernel $OUT/fps_pmc.json "furthest_point_sampling 8 x 8192 -> 4096 (tools/prof_fps.py 5)" /tmp/pmc_fps_a /tmp/pmc_fps_b /tmp/pmc_fps_c
# 5. the nearest-point searches of the five levels: binned against the sweeps, kernel times
(cd /tmp && rm -rf /tmp/p5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- python3 $GRAFT_REPO_ROOT/tools/knn2d_bench.py > $OUT/knn2d_bench.txt 2>/dev/null)
cp $(find /tmp/p5 -name "*kernel_stats.csv" | head -1) $OUT/knn2d_kernel_stats.csv
# 6. the hot-path sequence alone, eager, kernel stats
(cd /tmp && rm -rf /tmp/p6 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p6 -- python3 $GRAFT_REPO_ROOT/bench.py --workload hotpath --eager --steps 10 --warmup 2 --no-cpu-baseline --no-corr-microbench --backend none > /tmp/p6.log 2>&1)
cp $(find /tmp/p6 -name "*kernel_stats.csv" | head -1) $OUT/hotpath_kernel_stats.csv
# 7. the 8-rank host side
for w in 1 2; do timeout 300 python3 tools/host_rehearsal.py --ranks 8 --batches 384 --workers $w 2>/dev/null | grep "^{" > $OUT/host_rehearsal_w$w.json; done
timeout 300 python3 tools/host_rehearsal.py --ranks 8 --batches 384 2>/dev/null | grep "^{" > $OUT/host_rehearsal_default.json
ls -la $OUT
